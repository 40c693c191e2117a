#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>/) into profiles/<name>_*:
kernel-stats head, PMC means, corrected traffic (2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 on gfx950, see
MI355X_MICROARCH.md 'HBM'), and the FETCH_SIZE calibration from the gather probe."""
import json
import sys
from pathlib import Path

import pandas as pd


def main(src, name, workload, key, b_alg):
    src, dst = Path(src), Path("profiles")
    dst.mkdir(exist_ok=True)
    ks = pd.read_csv(src / "trace/bench_kernel_stats.csv")
    ks.head(12).to_csv(dst / f"{name}_kernel_stats.csv", index=False)
    k = ks[ks.Name.str.contains("spmm_hops")].iloc[0]
    out = {"command": f"tools/profile_bench.sh ({src.name}): rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE, --pmc WRITE_SIZE",
           "workload": workload, "kernel": k.Name, "calls": int(k.Calls), "avg_ns": float(k.AverageNs),
           "min_ns": int(k.MinNs), "max_ns": int(k.MaxNs), "pct_of_gpu_time": float(k.Percentage)}
    pm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        df = pd.read_csv(src / f"pmc_{c}/bench_counter_collection.csv")
        kk = df[df.Kernel_Name.str.contains("spmm_hops")]
        pm[c] = {"mean_KiB": float(kk.Counter_Value.mean()), "launches": int(len(kk)), "grid": int(kk.Grid_Size.iloc[0]),
                 "workgroup": int(kk.Workgroup_Size.iloc[0]), "lds_bytes": int(kk.LDS_Block_Size.iloc[0])}
    out["pmc"] = pm
    probe = src / "pmc_probe/probe_counter_collection.csv"
    if probe.exists():
        df = pd.read_csv(probe)
        g = df[df.Kernel_Name.str.contains("gather_kernel")].sort_values("Dispatch_Id")
        v = g.Counter_Value.values.reshape(-1, 7).mean(1)
        known = [(1 << 18) * 512 * lpr * 16 for _ in range(6) for lpr in (8, 16, 32, 64)]
        r = [float(a * 1024 / b) for a, b in zip(v, known)]
        out["fetch_size_calibration"] = {"ratio_FETCHx1024_over_known_bytes_by_table_MiB":
                                         {str(s): r[i * 4:(i + 1) * 4] for i, s in enumerate([16, 128, 512, 1229, 4096, 8192])}}
    fetch = 2 * pm["FETCH_SIZE"]["mean_KiB"] * 1024
    write = pm["WRITE_SIZE"]["mean_KiB"] * 1024
    out["traffic_bytes_per_launch"] = fetch + write
    out["traffic_breakdown"] = {"read_bytes_(2x FETCH_SIZE x1024)": fetch, "write_bytes_(WRITE_SIZE x1024)": write}
    out["algorithmic_bytes_per_launch"] = b_alg
    out["traffic_over_algorithmic"] = (fetch + write) / b_alg
    out["achieved_GBps_from_rocprof_avg"] = b_alg / (float(k.AverageNs) * 1e-9) / 1e9
    json.dump(out, open(dst / f"{name}_summary.json", "w"), indent=1)
    tfile = dst / "pmc_traffic.json"
    table = json.loads(tfile.read_text()) if tfile.exists() else {}
    table[key] = {"bytes_per_launch": fetch + write, "source": f"profiles/{name}_summary.json"}
    tfile.write_text(json.dumps(table, indent=1))
    print(name, "avg_ms", out["avg_ns"] / 1e6, "traffic/alg", out["traffic_over_algorithmic"], "achieved GB/s", out["achieved_GBps_from_rocprof_avg"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]))
