#!/bin/bash
# End-to-end rate of the drop-in `jellyfish count` on a big FASTQ in tmpfs (>= 50 M reads/s
# on 64 M reads).  usage: cli_scale.sh [pairs=32000000] [genome=320000000] [threads=64]
cd "$GRAFT_REPO_ROOT" || cd "$(dirname "$0")/.." || exit 1
PAIRS=${1:-32000000}; G=${2:-320000000}; T=${3:-64}
D=/dev/shm/rfx_cli_scale; mkdir -p $D
BIN=rufus_amd/bin
t0=$(date +%s.%N)
$BIN/rfx_synth_fastq $G 0 100 12345 0 $PAIRS $D/reads.fq || exit 1
t1=$(date +%s.%N)
ls -la $D/reads.fq
for mode in eager defer eager128 eager200; do
  [ $mode = defer ] && export RFX_COUNT_DEFER=1 || export RFX_COUNT_DEFER=0
  TT=$T; [ $mode = eager128 ] && TT=128; [ $mode = eager200 ] && TT=200
  s=$(date +%s.%N)
  $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t $TT -o $D/out.$mode.Jhash -C --timing $D/timing.$mode $D/reads.fq || exit 1
  e=$(date +%s.%N)
  python3 - <<PY
reads = 2 * $PAIRS
dt = $e - $s
print("cli_count mode=$mode threads=$TT reads=%d wall=%.2fs rate=%.1f M reads/s  (generate %.1fs)" % (reads, dt, reads / dt / 1e6, $t1 - $t0))
print(open("$D/timing.$mode").read().replace("\n", "  "))
PY
done
cmp $D/out.eager.Jhash $D/out.defer.Jhash > /dev/null 2>&1; python3 - <<PY
def payload(p):
    import hashlib
    f = open(p, "rb"); n = int(f.read(9)); f.seek(9 + n)
    h = hashlib.sha256()
    while True:
        b = f.read(1 << 26)
        if not b: break
        h.update(b)
    return h.hexdigest()
a, b = payload("$D/out.eager.Jhash"), payload("$D/out.defer.Jhash")
print("payload sha256 eager == defer:", a == b, a[:16])
PY
# pipe route
mkfifo $D/pipe.fq 2>/dev/null
cat $D/reads.fq > $D/pipe.fq &
s=$(date +%s.%N)
$BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t $T -o $D/out.pipe.Jhash -C $D/pipe.fq || exit 1
e=$(date +%s.%N)
python3 -c "print('cli_count mode=pipe rate=%.1f M reads/s' % (2*$PAIRS/($e-$s)/1e6))"
rm -rf $D
