#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/tn_trace; rm -rf $O; mkdir -p $O
RFX_STAGE_DEBUG=1 timeout 1500 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python bench.py --inner --workload tn --steps 1 --warmup 1 --no-check --no-cpu-baseline --no-end-to-end > $O/log 2>$O/err
grep "rfx stage" $O/err | sort | uniq -c | head
python3 - <<PY
import csv,re,collections,glob
f=glob.glob("$O/t/**/t_kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
d=collections.defaultdict(list)
for r in rows:
    m=re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:20]
    d[k].append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k in ("k_surv_place","k_msp_leaf"):
    v=sorted(d[k]); print(k, len(v), "total ms", sum(x[1] for x in v)/1e6)
    print([round(x[1]/1e3) for x in v])
PY
rm -rf $O/t
