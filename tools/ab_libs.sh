#!/bin/bash
# tools/ab_libs.sh <outfile> "<lib names>" -- interleaved A/B of build/ab/lib_<name>.so over a few shapes (2 rounds)
OUT=$1; LIBS=$2; : > $OUT
for round in 1 2; do for lib in $LIBS; do for args in "" "--d 64" "--d 256" "--shape products_x6 --steps 4 --warmup 1"; do
echo -n "round=$round $lib [$args] " >> $OUT
H2GCN_HIP_LIBRARY=$PWD/build/ab/lib_$lib.so timeout 400 python bench.py --no-cpu-baseline --no-probe --no-traffic --steps 10 --warmup 3 $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), 'adjoint', round(d['adjoint']['kernel_ms'],3), round(d['adjoint']['frac'],4))" >> $OUT 2>&1
done; done; done; cat $OUT
