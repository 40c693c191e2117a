#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3: one kernel-trace/stats pass, then one PMC pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; PMC never combined with other trace domains).
# usage: tools/profile_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --no-probe --no-traffic --no-hbm-leg "$@" > "$OUT/bench_trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --no-probe --no-traffic --no-hbm-leg --no-adjoint --steps 3 --warmup 1 "$@" > "$OUT/bench_pmc_$C.log" 2>&1
done
# calibration of the read counter on a known byte count: random 512 B row gathers (tools/gather_probe); PROBE=1 only
[ "${PROBE:-0}" = 1 ] && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_probe" -o probe -- "$ROOT/tools/gather_probe" > "$OUT/probe_pmc.log" 2>&1
find "$OUT" -name "*.csv" | head -50
