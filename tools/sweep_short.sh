#!/bin/bash
# tools/sweep_short.sh <outfile> -- short-segment regimes x segment-walk variants (GPU box)
OUT=$1; : > $OUT
run() { echo "## $*" >> $OUT; timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('adjoint',{})
print(json.dumps({'walk':d['config']['schedule']['segment_walk'],'kernel_ms':round(r['kernel_ms'],4),'frac':round(r['frac'],4),'adjoint_ms':round(a.get('kernel_ms',0),4),'adjoint_frac':round(a.get('frac',0),4)}))" >> $OUT 2>&1; }
for shape in lowdeg hbm16m arxiv; do for v in 3 2 5 0; do run --shape $shape --variant $v; done; done
run --shape lowdeg --d 64 --variant 3
run --shape lowdeg --d 64 --variant 5
run --shape lowdeg --d 256 --variant 3
run --shape lowdeg --d 256 --variant 5
run --variant 5
run --variant 0
cat $OUT
