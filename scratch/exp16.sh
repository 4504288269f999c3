#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
TESTS="${TESTS:-trio_in_blocks or table_counts or tumor}" ENVS="A=1" S1=0 bash scratch/exp4.sh
bash scratch/exp15.sh | grep -A12 "k_msp_leaf"
