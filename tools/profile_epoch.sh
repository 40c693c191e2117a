#!/bin/bash
# tools/profile_epoch.sh <tag> [epochs] [extra run_experiments flags...] -- full-batch H2GCN-2 epochs (training step + evaluation) at the
# products shape through the reference-style entry point under rocprofv3 (kernel trace + stats); prints the per-epoch time of every
# kernel family (the one-off operand construction / data generation kernels are listed separately).  Output: gpurun_out/prof_<tag>/
set -u
TAG=$1; EPOCHS=${2:-12}; shift; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT/h2gcn_amd"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o epoch -- python run_experiments.py H2GCN synthetic --shape products --epochs "$EPOCHS" --no_feature_normalize --early_stopping 0 "$@" > "$OUT/epoch.log" 2>&1
python - "$OUT" "$EPOCHS" <<'PY'
import sys, pandas as pd
from pathlib import Path
out, epochs = Path(sys.argv[1]), int(sys.argv[2])
print([l for l in (out / "epoch.log").read_text().splitlines() if "Epoch loop" in l][-1])
ks = pd.read_csv(next((out / "trace").rglob("*kernel_stats.csv")))
per_epoch = ks[ks.Calls >= epochs].copy()          # launched at least once per epoch
per_epoch["ms_per_epoch"] = per_epoch.TotalDurationNs / epochs / 1e6
per_epoch["calls_per_epoch"] = per_epoch.Calls / epochs
per_epoch["kernel"] = per_epoch.Name.str.slice(0, 104)
pd.set_option("display.width", 250)
print(per_epoch[["kernel", "calls_per_epoch", "ms_per_epoch"]].head(22).to_string(index=False))
spmm = per_epoch[per_epoch.Name.str.contains("spmm_hops_kernel")].ms_per_epoch.sum()
print(f"per epoch: hop kernels {spmm:.2f} ms, all other per-epoch kernels {per_epoch.ms_per_epoch.sum() - spmm:.2f} ms; one-off kernels (setup) {ks[ks.Calls < epochs].TotalDurationNs.sum() / 1e6:.1f} ms in total")
PY
