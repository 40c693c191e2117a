#!/bin/bash
# tools/ab_short_list.sh <outfile> -- (GPU box) knobs of the binned short-segment walk on the short-row shapes, same box,
# interleaved: prefetch depth (build/ab/lib_pd{1,3}.so vs the default 2), entries per wave, hop-major vs chunk-major list
# workgroups, and the round-3 in-tile short-row mode (build/ab/lib_r03.so) as the reference point.
OUT=$1; : > $OUT
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), 'adjoint', round(d['adjoint']['kernel_ms'],3), round(d['adjoint']['frac'],4))"; }
for round in 1 2; do
 for shape in "--shape lowdeg" "--shape lowdeg --d 64" "--shape hbm16m" "--shape arxiv"; do
  for cfg in "base" "base H2GCN_SHORT_HOP_MAJOR=1" "base H2GCN_SHORT_PER_WAVE=32" "base H2GCN_SHORT_PER_WAVE=16" "pd1" "pd3" "r03"; do
    set -- $cfg; lib=$1; envs=$2
    echo -n "round=$round [$shape] $cfg : " >> $OUT
    env $envs H2GCN_HIP_LIBRARY=$ROOT/build/ab/lib_$lib.so timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --no-hbm-leg --steps 10 --warmup 3 $shape 2>/dev/null | tail -1 | line >> $OUT 2>&1
  done
 done
done
cat $OUT
