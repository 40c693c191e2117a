#!/bin/bash
# tools/profile_cora_epoch.sh <tag> [epochs] -- the reference's own workload (Cora, full-batch H2GCN-2, entry point, replayed
# hipGraphs) under rocprofv3 --kernel-trace --stats: per-epoch count and GPU time of every kernel.  Output: gpurun_out/prof_<tag>/
set -u
TAG=$1; EPOCHS=${2:-1000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o epoch -- python "$ROOT/tools/epoch_time.py" --epochs "$EPOCHS" > "$OUT/epoch.log" 2>&1
python - "$OUT" "$EPOCHS" <<'PY'
import sys, pandas as pd
from pathlib import Path
out, epochs = Path(sys.argv[1]), int(sys.argv[2])
print((out / "epoch.log").read_text().strip().splitlines()[-1][:60])
ks = pd.read_csv(next((out / "trace").rglob("*kernel_stats.csv")))
per = ks[ks.Calls >= epochs * 0.9].copy()
per["us_per_epoch"] = per.TotalDurationNs / epochs / 1e3
per["calls_per_epoch"] = per.Calls / epochs
per["avg_us"] = per.AverageNs / 1e3
per["kernel"] = per.Name.str.slice(0, 110)
pd.set_option("display.width", 250)
print(per[["kernel", "calls_per_epoch", "avg_us", "us_per_epoch"]].to_string(index=False))
print(f"per epoch: {per.calls_per_epoch.sum():.1f} kernel launches, {per.us_per_epoch.sum():.1f} us of kernel time (sum of durations)")
PY
