#!/bin/bash
OUT=$1; : > $OUT
run() { echo "## $*" >> $OUT; timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('adjoint',{})
print(json.dumps({'sched':d['config']['schedule'],'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4),'adjoint_frac':round(a.get('frac',0),4)}))" >> $OUT 2>&1; }
for d in 100 132 200 300; do run --d $d; run --d $d --variant 4; done
cat $OUT
