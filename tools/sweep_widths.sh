#!/bin/bash
# tools/sweep_widths.sh <outfile> -- odd widths x slice widths (GPU box)
OUT=$1; : > $OUT
run() { echo "## $*" >> $OUT; timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --no-adjoint --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'kernel_ms':r['kernel_ms'],'frac':round(r['frac'],4)}))" >> $OUT 2>&1; }
for d in 100 132 200 300; do for sc in 64 128 256; do run --d $d --slice-cols $sc; done; done
run --d 256 --variant 4
run --d 512 --variant 4
run --d 512 --slice-cols 128
cat $OUT
