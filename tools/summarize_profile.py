#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>/) into profiles/<name>_*:
kernel-stats head, the forward and the adjoint launch of h2gcn::spmm_hops_kernel (rocprofv3 average duration, the
roofline fraction recomputed from it), PMC means and the corrected traffic (2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 on
gfx950, see MI355X_MICROARCH.md 'HBM'), and -- when the run included it -- the FETCH_SIZE calibration from the gather
probe.  Everything about the workload is read from the bench JSON line the profiled command printed.

    python tools/summarize_profile.py gpurun_out/prof_r02_products_d256 r02_products_d256
"""
import json
import re
import sys
from pathlib import Path

import pandas as pd

KERNEL = re.compile(r"spmm_hops_kernel<(\d+), (\d+), (true|false), (true|false), (true|false), (true|false)(?:, (?:true|false|\d+))*>")  # VEC, LPR, EXACT, SUM, OFF32, PIPE[, SHORT, EPI]
PEAK = 8000.0


def bench_line(log: Path):
    for line in reversed(log.read_text().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    raise SystemExit(f"no bench JSON line in {log}")


def split_kernels(df, col):
    fwd, adj = [], []
    for _, r in df.iterrows():
        m = KERNEL.search(r[col])
        if m:
            (adj if m.group(4) == "true" else fwd).append(r)
    return fwd, adj


def main(src, name):
    src, dst = Path(src), Path("profiles")
    dst.mkdir(exist_ok=True)
    line = bench_line(src / "bench_trace.log")
    ks = pd.read_csv(src / "trace/bench_kernel_stats.csv")
    ks.head(12).to_csv(dst / f"{name}_kernel_stats.csv", index=False)
    fwd, adj = split_kernels(ks, "Name")
    if not fwd:
        raise SystemExit("forward kernel not found in the trace")
    b_alg = line["roofline"]["algorithmic_bytes_per_launch"]
    out = {"command": f"tools/profile_bench.sh {src.name.replace('prof_', '')}: rocprofv3 --kernel-trace --stats -- python bench.py ...; "
                      "then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes",
           "workload": line["config"]["workload"], "schedule": line["config"].get("schedule"),
           "bench_line_of_the_traced_run": {k: line[k] for k in ("value", "ms_per_step")} | {
               "kernel_ms_hip_events": line["roofline"]["kernel_ms"], "frac_hip_events": line["roofline"]["frac"]},
           "forward": [], "algorithmic_bytes_per_launch": b_alg}
    for k in fwd:
        out["forward"].append({"kernel": k.Name, "calls": int(k.Calls), "avg_ns": float(k.AverageNs), "min_ns": int(k.MinNs),
                               "max_ns": int(k.MaxNs), "pct_of_gpu_time": float(k.Percentage)})
    # one forward step may be several launches (column chunks): time per step = total forward time / steps
    steps = line["steps"] + max(line["warmup"], 1)
    fwd_ns_per_step = sum(float(k.TotalDurationNs) for k in fwd) / steps
    rp = ks[ks.Name.str.contains("repack_slice_major_kernel")]
    if len(rp):  # the slice-major scratch copy is part of the forward launch
        out["scratch_copy"] = {"kernel": rp.Name.iloc[0], "calls": int(rp.Calls.iloc[0]), "avg_ns": float(rp.AverageNs.iloc[0])}
        fwd_ns_per_step += float(rp.TotalDurationNs.iloc[0]) / steps
    out["forward_ns_per_step_from_rocprof"] = fwd_ns_per_step
    # per-dispatch durations (kernel trace): the MEDIAN launch next to rocprofv3's average -- on DRAM-resident operands the first
    # launches after the allocations run several per cent slower than the steady state and pull a short run's average up
    trace = src / "trace/bench_kernel_trace.csv"
    if trace.exists():
        kt = pd.read_csv(trace, usecols=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        kt = kt[kt.Kernel_Name.map(lambda k: bool(KERNEL.search(k)))]
        kt["ns"] = kt.End_Timestamp - kt.Start_Timestamp
        kt["sum_mode"] = kt.Kernel_Name.map(lambda k: KERNEL.search(k).group(4) == "true")
        for label, sel in (("forward", kt[~kt.sum_mode]), ("adjoint", kt[kt.sum_mode])):
            if len(sel):
                out[f"{label}_launch_ns"] = {"launches": int(len(sel)), "median": float(sel.ns.median()), "min": int(sel.ns.min()),
                                             "max": int(sel.ns.max()), "first": int(sel.ns.iloc[0])}
        launches_per_step_ = max(1, round(sum(int(k.Calls) for k in fwd) / steps))
        if "forward_launch_ns" in out:
            med_step = out["forward_launch_ns"]["median"] * launches_per_step_ + (float(rp.TotalDurationNs.iloc[0]) / steps if len(rp) else 0.0)
            out["frac_from_rocprof_median"] = b_alg / (med_step * 1e-9) / 1e9 / PEAK
    out["achieved_GBps_from_rocprof_avg"] = b_alg / (fwd_ns_per_step * 1e-9) / 1e9
    out["frac_from_rocprof_avg"] = out["achieved_GBps_from_rocprof_avg"] / PEAK
    if adj and "adjoint" in line:
        b_adj = line["adjoint"]["algorithmic_bytes_per_launch"]
        k = adj[0]
        out["adjoint"] = {"kernel": k.Name, "calls": int(k.Calls), "avg_ns": float(k.AverageNs),
                          "algorithmic_bytes_per_launch": b_adj,
                          "achieved_GBps_from_rocprof_avg": b_adj / (float(k.AverageNs) * 1e-9) / 1e9,
                          "frac_from_rocprof_avg": b_adj / (float(k.AverageNs) * 1e-9) / 1e9 / PEAK,
                          "frac_hip_events": line["adjoint"]["frac"]}
        if "adjoint_launch_ns" in out:
            out["adjoint"]["frac_from_rocprof_median"] = b_adj / (out["adjoint_launch_ns"]["median"] * 1e-9) / 1e9 / PEAK
    pm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = src / f"pmc_{c}/bench_counter_collection.csv"
        if not f.exists():
            continue
        df = pd.read_csv(f)
        rows_f, rows_a = split_kernels(df, "Kernel_Name")
        if rows_f:
            kk = pd.DataFrame(rows_f)
            per_launch = kk.groupby("Dispatch_Id").Counter_Value.sum()   # counters may be split over XCD rows
            pm[c] = {"mean_KiB_per_launch": float(per_launch.mean()), "launches": int(len(per_launch)),
                     "grid": int(kk.Grid_Size.iloc[0]), "workgroup": int(kk.Workgroup_Size.iloc[0])}
        if rows_a:
            ka = pd.DataFrame(rows_a)
            pm.setdefault("adjoint", {})[c] = float(ka.groupby("Dispatch_Id").Counter_Value.sum().mean())
    out["pmc"] = pm
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        launches_per_step = max(1, round(sum(int(k.Calls) for k in fwd) / steps))
        fetch = 2 * pm["FETCH_SIZE"]["mean_KiB_per_launch"] * 1024 * launches_per_step
        write = pm["WRITE_SIZE"]["mean_KiB_per_launch"] * 1024 * launches_per_step
        out["traffic_bytes_per_step"] = fetch + write
        out["traffic_breakdown"] = {"read_bytes_(2x FETCH_SIZE x1024)": fetch, "write_bytes_(WRITE_SIZE x1024)": write,
                                    "note": "FETCH_SIZE counts L2 misses, i.e. bytes served by the Infinity Cache AND by HBM"}
        out["traffic_over_algorithmic"] = (fetch + write) / b_alg
        tfile = dst / "pmc_traffic.json"
        table = json.loads(tfile.read_text()) if tfile.exists() else {}
        cfg = line["config"]
        shape = {"BASELINE configs[3]": "products", "BASELINE configs[2]": "arxiv"}.get(cfg["workload"].split(":")[0])
        if shape is None:
            m = re.match(r"\(not a BASELINE config\) (\w+)", cfg["workload"])
            shape = m.group(1) if m else "unknown"
        key = f"{shape}|d={cfg['d']}|chunks={cfg['feature_chunks']}|slice={cfg['slice_cols']}|gpus={line['n_gpus']}"
        table[key] = {"bytes_per_launch": fetch + write, "source": f"profiles/{name}_summary.json"}
        tfile.write_text(json.dumps(table, indent=1))
    probe = src / "pmc_probe/probe_counter_collection.csv"
    if probe.exists():
        df = pd.read_csv(probe)
        g = df[df.Kernel_Name.str.contains("gather_kernel")].sort_values("Dispatch_Id")
        v = g.Counter_Value.values.reshape(-1, 7).mean(1)
        known = [(1 << 18) * 512 * lpr * 16 for _ in range(6) for lpr in (8, 16, 32, 64)]
        r = [float(a * 1024 / b) for a, b in zip(v, known)]
        out["fetch_size_calibration"] = {"ratio_FETCHx1024_over_known_bytes_by_table_MiB":
                                         {str(s): r[i * 4:(i + 1) * 4] for i, s in enumerate([16, 128, 512, 1229, 4096, 8192])}}
    for k in ("gather_ceiling_GBps", "peak_achievable"):
        if k in line["roofline"]:
            out[k] = line["roofline"][k]
    json.dump(out, open(dst / f"{name}_summary.json", "w"), indent=1)
    print(name, "fwd ms/step", fwd_ns_per_step / 1e6, "frac", round(out["frac_from_rocprof_avg"], 4),
          "traffic/alg", out.get("traffic_over_algorithmic"), "adjoint frac", out.get("adjoint", {}).get("frac_from_rocprof_avg"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
