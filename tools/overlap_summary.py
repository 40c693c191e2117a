"""Summarise tools/overlap_trace.sh: from rank 0's rocprofv3 kernel trace, how much of the exchange kernels' time ran
concurrently with SpMM kernels of the same process (stream-level overlap the pipelining relies on)."""
import json
import sys
from pathlib import Path

import pandas as pd

out = Path(sys.argv[1])
f = next(out.glob("trace/**/r0_kernel_trace.csv"))
df = pd.read_csv(f)
spmm = df[df.Kernel_Name.str.contains("spmm_hops_kernel")].sort_values("Start_Timestamp")
xchg = df[df.Kernel_Name.str.contains("pull_kernel|stage_kernel|wait_kernel|signal_kernel")]
# last 6 steps only (timed region): take the SpMM launches of the last 6 * C
line = [l for l in (out / "rank0.log").read_text().splitlines() if l.startswith("{")][-1]
cfg = json.loads(line)
C = len(str(cfg["config"]["feature_chunks"]).split("+")) if "+" in str(cfg["config"]["feature_chunks"]) else int(cfg["config"]["feature_chunks"])
spmm = spmm.tail(6 * C)
t0, t1 = spmm.Start_Timestamp.min(), spmm.End_Timestamp.max()
x = xchg[(xchg.End_Timestamp > t0) & (xchg.Start_Timestamp < t1)]
iv = sorted(zip(spmm.Start_Timestamp, spmm.End_Timestamp))


def overlap(a, b):
    tot = 0
    for s, e in iv:
        tot += max(0, min(e, b) - max(s, a))
    return tot


res = {}
for name, g in x.groupby(x.Kernel_Name.str.extract(r"(pull_kernel|stage_kernel|wait_kernel|signal_kernel)")[0]):
    dur = (g.End_Timestamp - g.Start_Timestamp).sum()
    ov = sum(overlap(a, b) for a, b in zip(g.Start_Timestamp, g.End_Timestamp))
    res[name] = {"launches": int(len(g)), "total_ms": dur / 1e6, "ms_overlapped_with_spmm": ov / 1e6, "fraction_overlapped": float(ov / dur) if dur else None}
summary = {"exchange": cfg["config"]["diagnostics"].get("exchange"), "feature_chunks": cfg["config"]["feature_chunks"],
           "ms_per_step": cfg["ms_per_step"], "spmm_ms_per_step": float((spmm.End_Timestamp - spmm.Start_Timestamp).sum() / 6e6),
           "window_ms": (t1 - t0) / 1e6, "exchange_kernels_in_window": res,
           "note": "two ranks share ONE GPU: both ranks' SpMMs and copies compete for the same HBM, so overlap shows the stream choreography, not a speed-up"}
print(json.dumps(summary, indent=1))
(out / "overlap_summary.json").write_text(json.dumps(summary, indent=1))
