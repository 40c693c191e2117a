#!/bin/bash
# extra PMC passes for the dominant kernel (each pass its own run; PMC never combined with other trace domains)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcx_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/p$i" -o b -- python "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 "$@" > "$OUT/p$i.log" 2>&1
done
python - <<PY
import pandas as pd, glob, json
res={}
for f in sorted(glob.glob("$OUT/p*/b_counter_collection.csv")):
    df=pd.read_csv(f); k=df[df.Kernel_Name.str.contains("spmm_hops")]
    for name,g in k.groupby("Counter_Name"): res[name]=float(g.Counter_Value.mean())
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/summary.json","w"), indent=1)
PY
