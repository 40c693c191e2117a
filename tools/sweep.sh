#!/bin/bash
# tools/sweep.sh <outfile> "<bench args>" ["<bench args>" ...] -- one condensed bench line per argument set (GPU box):
# schedule, forward / adjoint kernel time and roofline fraction.  H2GCN_HIP_LIBRARY may point at an alternative build.
#   tools/sweep.sh gpurun_out/s.txt "" "--d 100" "--shape arxiv --d 1433" "--shape lowdeg"
OUT=$1; shift; mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for ARGS in "$@"; do
  echo "## $ARGS" >> "$OUT"
  timeout 900 python bench.py --no-cpu-baseline --no-probe --no-traffic --no-hbm-leg --steps 8 --warmup 2 $ARGS 2>>"$OUT.err" | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('adjoint',{})
print(json.dumps({'sched':d['config'].get('schedule'),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4),'adjoint_ms':round(a.get('kernel_ms',0),3),'adjoint_frac':round(a.get('frac',0),4)}))" >> "$OUT" 2>&1
done
cat "$OUT"
