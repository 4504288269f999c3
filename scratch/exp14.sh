#!/bin/bash
# final numbers of the round-3 build: kernel stats, driver-flag bench, S1, tumor/normal
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
STATS=1 PMC=0 SQ=${SQ:-1} BENCH=1 EXTRA=1 bash scratch/make_profiles_r03.sh
