#!/bin/bash
# tools/ab.sh <outfile>: A/B of launch-bounds builds (build/ab/lib_mw*.so), interleaved, 2 rounds
OUT=$1; : > $OUT
for round in 1 2; do
for mw in 4 6 7 8; do
for sc in 64 128; do
echo "## mw=$mw slice=$sc round=$round" >> $OUT
H2GCN_HIP_LIBRARY=$PWD/build/ab/lib_mw$mw.so timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --slice-cols $sc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])" >> $OUT 2>&1
done; done; done
cat $OUT
