#!/bin/bash
# tools/profile_regimes.sh -- rocprofv3 kernel-trace + PMC passes for every kernel regime (GPU box).
# Output: gpurun_out/prof_<tag>/ per regime; condense with tools/summarize_profile.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() { TAG=$1; shift; echo "== $TAG: $*"; tools/profile_bench.sh "$TAG" "$@" > /dev/null 2>&1; tail -1 gpurun_out/prof_$TAG/bench_trace.log | cut -c1-300; }
run r02_products_d128
run r02_arxiv_d128 --shape arxiv
run r02_lowdeg_d128 --shape lowdeg
run r02_products_d64 --d 64
run r02_products_d100 --d 100
run r02_products_d256 --d 256
run r02_hbm16m_d128 --shape hbm16m
run r02_products_d200 --d 200
run r02_products_d512 --d 512
run r02_products_x6_d128 --shape products_x6 --steps 5 --warmup 2
