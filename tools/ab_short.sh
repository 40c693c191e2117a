#!/bin/bash
# tools/ab_short.sh <outfile>: interleaved A/B of library builds (build/ab/lib_*.so) on the short-row shapes
OUT=$1; : > $OUT
for round in 1 2; do for lib in before auto; do for args in "--shape lowdeg" "--shape lowdeg --d 64" "--shape hbm16m" "--shape arxiv"; do
echo -n "round=$round $lib [$args] " >> $OUT
H2GCN_HIP_LIBRARY=$PWD/build/ab/lib_$lib.so timeout 300 python bench.py --no-cpu-baseline --no-probe --no-traffic --steps 10 --warmup 3 $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), 'adjoint', round(d['adjoint']['kernel_ms'],3), round(d['adjoint']['frac'],4))" >> $OUT 2>&1
done; done; done; cat $OUT
