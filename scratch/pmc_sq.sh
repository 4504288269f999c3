cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o s -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1
python - <<'PY'
import csv,collections,re,glob
f=glob.glob('gpurun_out/pmc_sq/**/s_counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_[a-z0-9_]+)",r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
for k,v in agg.items():
    if k in("k_msp_leaf","k_msp_part1","k_part2","k_surv_sort"): print(k,{a:f"{b:.3g}" for a,b in v.items()})
PY
