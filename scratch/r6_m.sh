#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=32000000; G=100000000
D=/dev/shm/rfx_cli_scale; mkdir -p $D; O=gpurun_out/cli_trace6; mkdir -p $O
BIN=rufus_amd/bin
$BIN/rfx_synth_fastq $G 0 100 12345 0 $PAIRS $D/reads.fq || exit 1
run() {
  s=$(date +%s.%N)
  RFX_CLI_TRACE=1 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t $T -o $D/out.Jhash -C $D/reads.fq 2> $O/trace.x
  e=$(date +%s.%N)
  python3 - <<PY
import re
t=open("$O/trace.x").read()
def at(s):
    m=re.search(r"\[rfx\s+([0-9.]+) s\] "+s, t); return float(m.group(1)) if m else float('nan')
a=at("count: text arenas open") if "text arenas" in t else at("count: staging blocks pinned")
print("FAILED RUN:\n"+t[-1500:] if "nan" in str(at("count: input parsed")) else "", end="")
print("\n".join(l for l in t.splitlines() if "text fed" in l or "unmapped" in l or "arenas open" in l))
print("%-34s ingest %.3f s  (open %.3f, parsed %.3f, device done %.3f, closed %.3f)" % ("$1", at("count: input parsed")-a, a, at("count: input parsed"), at("count: finished on the device"), at("count: output closed")), re.findall(r"text route, (.*)", t))
PY
}
for i in 1 2 3 4 5 6 7 8 9 10; do
RFX_DEVICE_PARSE=1 T=64 run "text pread $i"
done
RFX_DEVICE_PARSE=1 RFX_HOST_THREADS=5 T=64 run "text pread 5 thr"
RFX_HOST_THREADS=5 RFX_HOST_PARSE=1 T=64 run "host parse 5 thr"
T=5 run "auto -t 5"
T=64 run "auto -t 64"
rm -rf $D
