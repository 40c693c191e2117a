#!/bin/bash
# tools/ab_output_stores.sh <outfile> -- (GPU box) why the forward's second output stream is expensive once X is DRAM-resident:
#  (1) interleaved A/B of build/ab/lib_{base,plainstore,defer}.so (non-temporal vs plain stores; a row's two hop pieces
#      stored back to back) on products and products_x6, forward + adjoint;
#  (2) TCC write / read request counters of the forward and the adjoint launch on products_x6 (separate --pmc passes).
OUT=$1; : > $OUT
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for round in ${ROUNDS:-1 2}; do for lib in ${LIBS:-base plainstore defer}; do for args in "" "--shape products_x6 --steps 4 --warmup 1"; do
  echo -n "round=$round $lib [$args] " >> $OUT
  H2GCN_HIP_LIBRARY=$ROOT/build/ab/lib_$lib.so timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --no-hbm-leg --steps 10 --warmup 3 $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), 'adjoint', round(d['adjoint']['kernel_ms'],3), round(d['adjoint']['frac'],4))" >> $OUT 2>&1
done; done; done
export TMPDIR=/tmp
for SET in "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_DRAM_CREDIT_STALL" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ_LEVEL"; do
  D=$ROOT/gpurun_out/pmc_stores_$(echo $SET | tr ' ' '_' | cut -c1-60)
  (cd /tmp && rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$D" -o t -- python "$ROOT/bench.py" --shape products_x6 --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-traffic --no-hbm-leg > "$D.log" 2>&1)
  python - "$D" "$SET" >> $OUT <<'PY'
import sys, glob
import pandas as pd
d, names = sys.argv[1], sys.argv[2].split()
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("counters", names, ": no output"); sys.exit(0)
df = pd.read_csv(f[0])
df = df[df.Kernel_Name.str.contains("spmm_hops_kernel")]
df["mode"] = df.Kernel_Name.map(lambda k: "adjoint" if k.split("spmm_hops_kernel<")[1].split(",")[3].strip() == "true" else "forward")
for (mode, name), g in df.groupby(["mode", "Counter_Name"]):
    per = g.groupby("Dispatch_Id").Counter_Value.sum()
    print(f"products_x6 {mode:8s} {name:36s} mean per launch {per.mean():.4e}  (launches {len(per)})")
PY
done
cat $OUT
