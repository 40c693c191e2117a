#!/bin/bash
# tools/ab2.sh <outfile> libA libB ... : interleaved A/B of library builds on the products shape (3 rounds)
OUT=$1; shift; : > $OUT
for round in 1 2 3; do for lib in "$@"; do
echo -n "$lib round=$round " >> $OUT
H2GCN_HIP_LIBRARY=$PWD/build/ab/$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])" >> $OUT 2>&1
done; done; cat $OUT
