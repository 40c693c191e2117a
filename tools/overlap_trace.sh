#!/bin/bash
# tools/overlap_trace.sh <exchange> <chunks> -- two ranks of bench.py on ONE GPU (products shape), rank 0 under rocprofv3
# --kernel-trace: shows from timestamps that the pulls of chunk c+1 run while the SpMM of chunk c does.
# Output: gpurun_out/overlap_<exchange>_<chunks>/ + a summary on stdout (tools/overlap_summary.py).
EX=${1:-ipc_kernel}; CH=${2:-4}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/overlap_${EX}_${CH//+/_}
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
PORT=$((20000 + RANDOM % 20000))
COMMON="MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT WORLD_SIZE=2 H2GCN_DIST_BACKEND=gloo H2GCN_SHARE_GPU=1"
ARGS="--gpus 2 --no-cpu-baseline --no-probe --no-traffic --no-adjoint --steps 6 --warmup 2 --exchange $EX --chunks $CH"
env $COMMON RANK=1 LOCAL_RANK=1 python "$ROOT/bench.py" $ARGS > "$OUT/rank1.log" 2>&1 &
P1=$!
env $COMMON RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o r0 -- python "$ROOT/bench.py" $ARGS > "$OUT/rank0.log" 2>&1
wait $P1
python "$ROOT/tools/overlap_summary.py" "$OUT"
