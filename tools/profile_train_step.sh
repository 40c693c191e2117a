#!/bin/bash
# tools/profile_train_step.sh <tag> [hidden] -- one full-batch H2GCN-2 training step at the products shape under rocprofv3
# (kernel trace + stats); prints the per-step time of every kernel family.  Output: gpurun_out/prof_<tag>/
set -u
TAG=$1; HIDDEN=${2:-64}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o step -- python "$ROOT/tools/epoch_products.py" "$HIDDEN" > "$OUT/step.log" 2>&1
python - "$OUT" <<'PY'
import sys, pandas as pd
from pathlib import Path
out = Path(sys.argv[1])
print((out / "step.log").read_text().strip().splitlines()[-1])
ks = pd.read_csv(next((out / "trace").rglob("*kernel_stats.csv")))
steps = 7
ks["ms_per_step"] = ks.TotalDurationNs / steps / 1e6
ks["short"] = ks.Name.str.slice(0, 110)
print(ks[["short", "Calls", "ms_per_step"]].head(24).to_string(index=False))
spmm = ks[ks.Name.str.contains("spmm_hops_kernel")].ms_per_step.sum()
print(f"per step: hop kernels {spmm:.2f} ms, everything else {ks.ms_per_step.sum() - spmm:.2f} ms (sum over kernels {ks.ms_per_step.sum():.2f} ms)")
PY
