#!/bin/bash
# tools/sweep_r02.sh <outfile> -- one condensed bench line per regime (GPU box): forward + adjoint fractions
OUT=$1; : > $OUT
run() { echo "## $*" >> $OUT; timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('adjoint',{})
print(json.dumps({'edges_per_s':d['value'],'kernel_ms':r['kernel_ms'],'frac':round(r['frac'],4),'adjoint_ms':a.get('kernel_ms'),'adjoint_frac':a.get('frac')}))" >> $OUT 2>&1; }
run --d 64
run --d 100
run --d 128
run --d 132
run --d 200
run --d 256
run --d 448
run --d 512
run --shape arxiv
run --shape lowdeg
cat $OUT
