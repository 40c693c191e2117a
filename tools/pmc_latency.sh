#!/bin/bash
# average L2->fabric read latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) and credit stalls, slice 128 vs 64
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_latency; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for SC in 128 64; do
  i=0
  for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_BUSY_sum TCC_CYCLE_sum" "TCC_REQ_sum TCC_TAG_STALL_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/s${SC}_p$i -o b -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --slice-cols $SC > $OUT/s${SC}_p$i.log 2>&1
  done
done
python - <<PY
import pandas as pd, glob, json
res={}
for sc in (128,64):
    r={}
    for f in sorted(glob.glob("$OUT/s%d_p*/b_counter_collection.csv"%sc)):
        df=pd.read_csv(f); k=df[df.Kernel_Name.str.contains("spmm_hops")]
        for name,g in k.groupby("Counter_Name"): r[name]=float(g.Counter_Value.mean())
    if "TCC_EA0_RDREQ_LEVEL_sum" in r: r["avg_read_latency_cycles"]=r["TCC_EA0_RDREQ_LEVEL_sum"]/r["TCC_EA0_RDREQ_sum"]
    res["slice_%d"%sc]=r
print(json.dumps(res, indent=1)); json.dump(res, open("$OUT/summary.json","w"), indent=1)
PY
