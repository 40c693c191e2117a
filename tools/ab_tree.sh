#!/bin/bash
# tools/ab_tree.sh <outfile> [bench args]: interleaved A/B of an older source tree (build/ab/old, exported with git archive
# and built in place) against the working tree, same box, 3 rounds
OUT=$1; shift; : > $OUT
for round in 1 2 3; do for tree in build/ab/old .; do
echo -n "$tree round=$round " >> $OUT
(cd $tree && H2GCN_BENCH_CHILD=1 timeout 300 python bench.py --no-cpu-baseline --no-probe --no-adjoint --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])") >> $OUT 2>&1
done; done; cat $OUT
