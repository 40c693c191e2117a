#!/bin/bash
# tools/profile_regimes.sh [round tag, default r05] -- rocprofv3 kernel-trace + PMC passes for every kernel regime (GPU box).
# Output: gpurun_out/prof_<tag>_<regime>/ per regime, condensed into profiles/<tag>_<regime>_summary.json + _kernel_stats.csv.
set -u
R=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() { TAG=$1; shift; echo "== $TAG: $*"; tools/profile_bench.sh "$TAG" "$@" > /dev/null 2>&1; python tools/summarize_profile.py gpurun_out/prof_$TAG $TAG; }
run ${R}_products_d128
run ${R}_arxiv_d128 --shape arxiv
run ${R}_lowdeg_d128 --shape lowdeg
run ${R}_lowdeg_d64 --shape lowdeg --d 64
run ${R}_products_d64 --d 64
run ${R}_products_d100 --d 100
run ${R}_products_d130 --d 130
run ${R}_products_d200 --d 200
run ${R}_products_d256 --d 256
run ${R}_products_d512 --d 512
run ${R}_arxiv_d1433 --shape arxiv --d 1433
run ${R}_arxiv_d3703 --shape arxiv --d 3703
run ${R}_hbm16m_d128 --shape hbm16m
run ${R}_products_x6_d128 --shape products_x6 --steps 12 --warmup 2
# mixed segment classes in one launch (round 4: CSR-adaptive dispatch by class, binned short-segment list)
run ${R}_h2gcn_like_d128 --shape h2gcn_like
run ${R}_products_tail_d128 --shape products_tail
run ${R}_h2gcn_like_d64 --shape h2gcn_like --d 64
