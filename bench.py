#!/usr/bin/env python3
"""bench.py -- aggregated edges/s of the fused 1+2-hop SpMM (BASELINE.json metric) on N GPUs of one node.

    python bench.py                                  # N=1, products shape (configs[3]), 20 steps
    python bench.py --gpus N --steps K --warmup W    # launches its N ranks itself (re-exec under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W       # configs[4]: same graph row-partitioned, strong scaling

One "step" = one pass of the hot path over the whole graph: [all-gather of the row-sharded embedding over
RCCL when N > 1] + one fused 1+2-hop SpMM launch per rank (GCNLayer.call, reference
h2gcn/models/_layers.py:78-81).  Inputs are synthetic (h2gcn_amd/synth.py), generated on the device and
resident in HBM before the timed region.  Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the
algorithmic-bytes model behind `roofline`.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

#: order-independent fingerprint of Y (sum of the fp32 bit patterns as int64) per (shape, d), computed by the ORACLE on the CPU:
#: `python -m oracle.fullsize <shape>` rebuilds every row of the operands on the host and runs oracle_spmm_tree_f32_mt (the
#: library's documented summation tree in plain C) -- profiles/r06_oracle_checksums_of_the_bench_shapes.txt; the arxiv entry is
#: recomputed in the CPU suite, arxiv and products are asserted element by element on the GPU box
#: (tests/test_fullsize_parity_gpu.py).  The per-row tree is canonical, so EVERY schedule -- any rank count, exchange, chunking,
#: slice width -- must reproduce it bit for bit (SURVEY.md 8(e) "Determinism"): `checksum_matches_n1` is a comparison with the
#: oracle, not with an earlier run of the HIP path.
N1_CHECKSUMS = {("products", 128): -26948829970322352, ("products", 64): -13319257904282617, ("arxiv", 128): -1390319019045034,
                ("h2gcn_like", 128): -20343064339982979, ("products_tail", 128): -25034865256373447,
                ("lowdeg", 128): -57806506298044835, ("hbm16m", 128): -132311004237956237, ("products_x6", 128): -179026745709730822}

#: ... and of the adjoint dX = sum_k A_k^T W[:, k, :] with W = synth_features(2 d, seed 77) (`python -m oracle.fullsize --adjoint <shape>`:
#: host-built transpose, the documented tree in plain C); single-GPU lines report `adjoint.checksum_matches_oracle`
N1_ADJOINT_CHECKSUMS = {("arxiv", 128): -621262802795563, ("products", 128): -12144664314681507, ("lowdeg", 128): -25634735670479872,
                        ("h2gcn_like", 128): -8224933467863159, ("products_tail", 128): -10435508044493920}

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable copy)


def peak_source(pr):
    """"confirm on the box".  `roofline.peak` stays the guide's spec constant (what every fraction of every round
    is quoted against); next to it goes what THIS box reports about its memory system -- hipDeviceProp_t's memory clock and bus
    width, read by tools/gather_probe -- and the rate they imply: bus_width / 8 x clock x transfers per reported clock.  The
    runtime reports the HBM command clock; HBM3 / HBM3E move 4 transfers per pin per such clock (MI300X: 1300 MHz x 8192 bit x 4
    = 5.3 TB/s = its spec), so that is the factor used; the 2x reading is carried too."""
    clk, width = (pr or {}).get("memory_clock_khz"), (pr or {}).get("memory_bus_width_bits")
    if not clk or not width:
        return {"peak_source": "spec constant (MI355X_MICROARCH.md: HBM3E 8.0 TB/s); the box reported no memory clock / bus width",
                "peak_confirmed_on_box": False}
    per_clock = clk * 1e3 * width / 8 / 1e9
    derived = 4 * per_clock
    ok = abs(derived / HBM_PEAK_GBPS - 1.0) <= 0.05
    return {"peak_source": f"spec constant (MI355X_MICROARCH.md: HBM3E 8.0 TB/s); this box ({pr.get('device_name')}, {pr.get('gcn_arch')}) reports "
                           f"memory clock {clk / 1e3:.0f} MHz x bus {width} bit -> {derived:.0f} GB/s at 4 transfers/clock "
                           f"({2 * per_clock:.0f} at 2): " + ("confirms the constant" if ok else "does NOT match the constant within 5 %"),
            "peak_confirmed_on_box": ok, "peak_box_memory_clock_khz": clk, "peak_box_bus_width_bits": width, "peak_box_derived_GBps": derived}


def algorithmic_bytes(nnz_list, n_rows_out, d, n_hops):
    """per edge one column id + one value + one gathered feature row (no-reuse model), per hop
    one int64 row-pointer sweep, plus one write of Y[n_rows, H, d]."""
    return sum(z * (4 + 4 + 4 * d) + (n_rows_out + 1) * 8 for z in nnz_list) + n_rows_out * n_hops * d * 4


def compulsory_bytes(nnz_list, n_rows_out, n_cols, d, n_hops):
    """Lower bound with perfect reuse of X (SURVEY.md §8d `B_min`): every index/value once, X once, Y once."""
    return sum(z * 8 + (n_rows_out + 1) * 8 for z in nnz_list) + n_cols * d * 4 + n_rows_out * n_hops * d * 4


def pmc_traffic(shape, d, chunks, slice_cols, world):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    see profiles/*_summary.json); bench.py cannot collect counters itself, so this is null for any
    configuration that has not been profiled."""
    try:
        table = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    except Exception:
        return None
    key = f"{shape}|d={d}|chunks={chunks}|slice={slice_cols or 'auto'}|gpus={world}"
    return table.get(key)


def cpu_baseline(plan_csr, x_full, d, target_seconds=12.0):
    """Reported baseline (NOT the target): the plain-C oracle port of the reference's CPU arithmetic, 1 thread
    (TF's CPU SparseTensorDenseMatMul is single-threaded), on a bounded row sample of the SAME operands."""
    from oracle import gcn_layer as og

    rowptrs = [c[0] for c in plan_csr]
    n_rows = rowptrs[0].numel() - 1
    x_cpu = x_full.cpu().numpy()

    def sample(n_take):
        parts = []
        edges = 0
        for rp, ci, va in plan_csr:
            hi = int(rp[n_take])
            parts.append((rp[: n_take + 1].cpu().numpy(), ci[:hi].cpu().numpy(), va[:hi].cpu().numpy()))
            edges += hi
        return parts, edges

    # calibrate on ~1e6 edges, then size the sample for ~target_seconds
    avg_deg = sum(int(rp[-1]) for rp in rowptrs) / max(n_rows, 1)
    n0 = max(1, min(n_rows, int(1e6 / max(avg_deg, 1))))
    parts, edges = sample(n0)
    t = time.perf_counter()
    og.gcn_layer_c(parts, x_cpu)
    dt = max(time.perf_counter() - t, 1e-6)
    rate = edges / dt
    n1 = max(n0, min(n_rows, int(n0 * target_seconds / dt)))
    parts, edges = sample(n1)
    t = time.perf_counter()
    og.gcn_layer_c(parts, x_cpu)
    dt = time.perf_counter() - t
    res = {
        "value": edges / dt, "unit": "edges/s", "cores": 1, "kind": "port",
        "sample": f"rows [0,{n1}) of the same operands: {edges} aggregated edges, d={d}, {dt:.1f} s, "
                  f"oracle/spmm_oracle.c (plain C, -O2, 1 thread; host has {os.cpu_count()} cores)",
        "calibration_edges_per_s": rate,
    }
    # the same C loop with the output rows spread over all host cores (OpenMP; identical bits) -- an upper bound on what
    # a parallelised build of the reference's CPU kernel could do on this host
    try:
        # threads the OpenMP runtime will really use: OMP_NUM_THREADS when set (torch.distributed.run exports 1 per rank;
        # bench_supervisor widens that for its workers), else every core this process may run on
        threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or len(os.sched_getaffinity(0))
        og.gcn_layer_c(parts, x_cpu, threads=True)  # warm-up (thread pool)
        t = time.perf_counter()
        og.gcn_layer_c(parts, x_cpu, threads=True)
        dt3 = time.perf_counter() - t
        res["port_all_cores"] = {"value": edges / dt3, "unit": "edges/s", "cores": threads, "seconds": dt3,
                                 "what": "oracle_gcn_layer_f32_mt (OpenMP over output rows), same sample"}
    except Exception as e:  # noqa: BLE001
        res["port_all_cores"] = {"value": None, "error": str(e)}
    # second reported baseline (SURVEY.md §8d ii): torch's CPU CSR @ dense on all host threads, same sample
    try:
        xt = torch.from_numpy(x_cpu)
        mats = [torch.sparse_csr_tensor(torch.from_numpy(p[0]), torch.from_numpy(p[1].astype(np.int64)),
                                        torch.from_numpy(p[2]), size=(n1, x_cpu.shape[0])) for p in parts]
        for m in mats:
            m @ xt  # warm-up
        t = time.perf_counter()
        for m in mats:
            m @ xt
        dt2 = time.perf_counter() - t
        res["torch_cpu_all_threads"] = {"value": edges / dt2, "unit": "edges/s", "threads": torch.get_num_threads(),
                                        "seconds": dt2, "what": "torch.sparse_csr_tensor @ dense, same sample"}
    except Exception as e:
        res["torch_cpu_all_threads"] = {"value": None, "error": str(e)}
    return res


def probe_ceilings(table_mib, row_bytes):
    """Live ceilings of THIS box from tools/gather_probe (built by __graft_entry__.build(); micro-benchmark, not
    product): random `row_bytes`-row gathers out of a `table_mib` MiB table (what the SpMM's gather stream can
    reach at its working-set size) and a plain stream copy (the achievable HBM rate)."""
    import subprocess
    exe = ROOT / "tools" / "gather_probe"
    if not exe.exists():
        return {"error": "tools/gather_probe not built"}
    try:
        out = subprocess.run([str(exe), "--point", str(int(table_mib)), str(int(row_bytes))], capture_output=True,
                             text=True, timeout=120).stdout
        return json.loads(out.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 -- a report, never a reason to lose the GPU number
        return {"error": f"{type(e).__name__}: {e}"}


def measure_traffic_live(a, timeout_s=240):
    """HBM-side bytes per forward launch measured on THIS box: two short child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel-trace only -- MI355X_MICROARCH.md 'HBM');
    traffic = 2 x FETCH_SIZE x 1024 (gfx950 counts half of a wide coalesced read) + WRITE_SIZE x 1024, averaged over
    the forward launches.  Returns None when rocprofv3 is unavailable or the pass fails (the line then falls back to
    the committed offline profile and says so)."""
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists() or os.environ.get("H2GCN_BENCH_CHILD") == "1":
        return None
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) or k in ("HSA_TOOLS_LIB", "LD_PRELOAD") for k in os.environ):
        return None  # this process is itself being profiled / instrumented: do not nest a profiler under it
    pat = re.compile(r"spmm_hops_kernel<\d+, \d+, (?:true|false), false")   # forward launches (SUM = false)
    child = [sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-probe",
             "--no-adjoint", "--no-traffic", "--shape", a.shape, "--variant", str(a.variant)]
    for flag, val in (("--d", a.d), ("--slice-cols", a.slice_cols), ("--long-row-threshold", a.long_row_threshold),
                      ("--rows-per-wave", a.rows_per_wave)):
        if val:
            child += [flag, str(val)]
    if a.chunks:
        child += ["--chunks", a.chunks]
    out = {}
    try:
        import pandas as pd
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                env = dict(os.environ, TMPDIR="/tmp", H2GCN_BENCH_CHILD="1")
                r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", td, "-o", "t", "--"] + child,
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
                files = list(Path(td).rglob("*counter_collection.csv"))
                if r.returncode != 0 or not files:
                    return None
                df = pd.read_csv(files[0])
                df = df[df.Kernel_Name.map(lambda k: bool(pat.search(k)))]
                if df.empty:
                    return None
                out[counter] = float(df.groupby("Dispatch_Id").Counter_Value.sum().mean())
    except Exception:  # noqa: BLE001 -- a report, never a reason to lose the GPU number
        return None
    launches_per_step = len(a.chunks.split("+")) if "+" in a.chunks else int(a.chunks or 1)
    return {"bytes_per_launch": (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024 * launches_per_step,
            "read_bytes": 2 * out["FETCH_SIZE"] * 1024 * launches_per_step, "write_bytes": out["WRITE_SIZE"] * 1024 * launches_per_step,
            "source": "LIVE: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this command on this box "
                      "(2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 per step; FETCH_SIZE counts L2 misses, i.e. Infinity-Cache hits too)"}


def secondary_leg(shape, steps=200, warmup=20, timeout_s=300):
    """BASELINE configs[2] (the arxiv shape: an Infinity-Cache-resident 0.2 ms launch) next to the headline in the SAME default
    line: a child run of this script after the timed region -- kernel time from the hip events around each launch, the PMC traffic
    from its own two rocprofv3 passes.  Reported under `secondary` (+ scalars under `roofline`); never part of `value`."""
    import subprocess

    child = [sys.executable, str(ROOT / "bench.py"), "--shape", shape, "--steps", str(steps), "--warmup", str(warmup),
             "--no-cpu-baseline", "--no-adjoint", "--no-hbm-leg", "--no-probe", "--no-secondary"]
    try:
        env = {k: v for k, v in os.environ.items() if k != "H2GCN_BENCH_CHILD"}     # the child collects its own PMC passes
        r = subprocess.run(child, env=env, capture_output=True, text=True, timeout=timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"child exited with {r.returncode}: {r.stderr[-300:]}"}
        c = json.loads(line[-1])
        rf = c["roofline"]
        return {"workload": c["config"]["workload"], "steps": c["steps"], "warmup": c["warmup"], "ms_per_step": c["ms_per_step"],
                "edges_per_s": c["value"], "kernel_ms": rf["kernel_ms"], "kernel_ms_median": rf["kernel_ms_median"],
                "achieved_GBps": rf["achieved"], "frac": rf["frac"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                "traffic": rf.get("traffic"), "traffic_over_algorithmic": rf.get("traffic_over_algorithmic"),
                "traffic_source": rf.get("traffic_source"), "segment_walk": c["config"]["schedule"].get("segment_walk"),
                "checksum_matches_n1": c["config"].get("checksum_matches_n1"),
                "source": f"LIVE child run of `bench.py --shape {shape} --steps {steps} --warmup {warmup}` on this box after the timed region"}
    except Exception as e:  # noqa: BLE001 -- a report, never a reason to lose the GPU number
        return {"error": f"{type(e).__name__}: {e}"}


HBM_LEG_STEPS = 12   # timed launches of the HBM-resident leg (after 2 warm-ups)


def hbm_resident_leg(timeout_s=600):
    """BASELINE's metric says "achieved HBM GB/s": on the headline shape X (1.23 GB; 614 MB per column slice) is only
    2.4-4.8x the 256 MiB Infinity Cache, so part of the delivered gather bytes are cache hits -- and rocprofv3 on
    gfx950 has no counter on the DRAM side of that cache (TCC_EA0_RDREQ_DRAM counts requests *destined* for local
    memory, Infinity-Cache hits included; no MALL / HBM-channel counter is exposed: `rocprofv3 --list-avail`).  So the
    line carries a second, short, labelled leg instead: the SAME kernel on the SAME degree distribution with |V| scaled
    until X (8.2 GB) and every column slice of it are >= 16x the cache (shape `products_x6`: |V| = 16M, 8e8 nonzeros per
    hop), where delivered bytes ~= DRAM bytes.  Runs as a child process of this script after the headline has been
    timed (never inside its timed region): 2 warm-up + HBM_LEG_STEPS timed launches, figures from the median launch."""
    import subprocess

    child = [sys.executable, str(ROOT / "bench.py"), "--shape", "products_x6", "--steps", str(HBM_LEG_STEPS), "--warmup", "2",
             "--no-cpu-baseline", "--no-adjoint", "--no-traffic", "--no-hbm-leg"]
    try:
        r = subprocess.run(child, env=dict(os.environ, H2GCN_BENCH_CHILD="1"), capture_output=True, text=True, timeout=timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"child exited with {r.returncode}: {r.stderr[-300:]}"}
        c = json.loads(line[-1])
        rf = c["roofline"]
        # the figure is the MEDIAN launch of >= 10 (the DRAM-side rate moves by a few per cent from launch to launch and from
        # box to box; a 3-launch mean, as rounds 2-4 reported, cannot tell a regression from the box)
        b_alg = rf["algorithmic_bytes_per_launch"]
        med = rf["kernel_ms_median"]
        achieved = b_alg / (med * 1e-3) / 1e9
        ceil = rf.get("gather_ceiling_GBps")
        return {"workload": c["config"]["workload"], "X_bytes": c["config"]["n_rows"] * c["config"]["d"] * 4,
                "steps": c["steps"], "warmup": c["warmup"],
                "kernel_ms": med, "kernel_ms_median": med, "kernel_ms_min": rf["kernel_ms_min"], "kernel_ms_max": rf["kernel_ms_max"],
                "kernel_ms_mean": rf["kernel_ms"],
                "edges_per_s": sum(c["config"]["nnz_per_hop"]) / (med * 1e-3), "achieved_GBps": achieved, "frac": achieved / HBM_PEAK_GBPS,
                "frac_range": [b_alg / (rf["kernel_ms_max"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, b_alg / (rf["kernel_ms_min"] * 1e-3) / 1e9 / HBM_PEAK_GBPS],
                "gather_ceiling_GBps": ceil, "stream_read_GBps": (rf.get("ceilings") or {}).get("stream_read_GBps"),
                "achieved_over_gather_ceiling": None if not ceil else achieved / ceil,
                "checksum_matches_oracle": c["config"].get("checksum_matches_n1"),   # Y of the 16M-row launch == the CPU oracle's, bit for bit
                "source": f"LIVE child run of `bench.py --shape products_x6 --steps {HBM_LEG_STEPS} --warmup 2` on this box, same kernel and "
                          "schedule rule; figures from the MEDIAN launch; X and each of its column slices are >= 16x the Infinity "
                          "Cache, so this rate is DRAM-side"}
    except Exception as e:  # noqa: BLE001 -- a report, never a reason to lose the GPU number
        return {"error": f"{type(e).__name__}: {e}"}


def dry_exchange(a, world, rank, device, backend, to_stderr=False, rccl_log_dir=None):
    """`--dry-exchange` (N > 1): FIRST-CONTACT check of every exchange form on this node before any big allocation --
    1 MiB shards, a handful of rounds each, per-candidate GB/s of landed bytes, which candidates fail and why.  Meant to
    be the first thing run on a multi-GPU box: it exercises the RCCL high-priority stream, the grouped send/recv form
    and the IPC handle exchange with real peers, and the table carries what RCCL chose (channels, transports, algorithm /
    protocol picks: condensed from its per-process debug file, `rccl`)."""
    from h2gcn_amd import HopPlan, synth
    from h2gcn_amd.partition import PipelinedHopAggregation, RowPartition

    d = a.d or 128
    per = max(1, (1 << 20) // (4 * d))
    n = per * world
    part = RowPartition.equal(n, world)
    r0, r1 = part.rows(rank)
    degs = [synth.synth_degrees(n, 8 * n, s, n) for s in (synth.SEED_A1, synth.SEED_A2)]
    csr = [synth.synth_hop_rows(degs[k], n, (synth.SEED_A1, synth.SEED_A2)[k], r0, r1, device) for k in range(2)]
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
    x_local = synth.synth_features(d, synth.SEED_X, r0, r1, device)
    hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    table, rejected = {}, {}

    def all_ok(flag):
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0)

    fams = exchange_families(a)
    if to_stderr and not a.exchange:
        # inside a measuring run the table is a quick smoke test of RCCL and of the IPC bootstrap BEFORE the big allocations: the
        # two safest forms only.  The others (copy-engine pulls, grouped send/recv) meet the node inside the calibration, which
        # runs safest-first -- if one of them takes the job down there, every safer candidate is already timed and on record
        fams = [f for f in fams if EXCHANGE_ORDER.get(f, 9) < 2] or fams[:1]
    for ex in fams:
        for spec in ([a.chunks] if a.chunks else ["1", "2"]):
            key = f"{ex}/{spec}"
            if ex == "ipc_engine" and world - 1 > hwq - 2:
                rejected[key] = f"copy-engine pulls park {world - 1} spin-wait kernels on hardware queues; GPU_MAX_HW_QUEUES = {hwq} leaves {hwq - 2}"
                continue
            if to_stderr:   # (part of a measuring run: on record which form is in flight, should it take the job down)
                progress({"starting": key, "stage": "first contact"}, rank)
            cand, why = None, None
            try:
                cand = PipelinedHopAggregation(plan, n, d, parse_chunks(spec, d), device, exchange=ex, partition=part)
            except Exception as e:  # noqa: BLE001
                why = f"construct: {type(e).__name__}: {e}"
            if not all_ok(cand is not None):
                rejected[key] = why or "construction failed on another rank"
                if cand is not None and cand.ipc is not None:
                    cand.ipc._destroy_local()
                if to_stderr:
                    progress({"finished": key, "stage": "first contact"}, rank)
                continue
            try:
                cand(x_local)                      # stages the shard and runs one full step (exchange + SpMM)
                torch.cuda.synchronize()
                cand.check()
                dist.barrier()
                t = time.perf_counter()
                for _ in range(10):
                    cand.exchange_only()
                torch.cuda.synchronize()
                dist.barrier()
                ms = (time.perf_counter() - t) / 10 * 1e3
                cand.check()
            except Exception as e:  # noqa: BLE001
                why = f"run: {type(e).__name__}: {e}"
            if not all_ok(why is None):
                rejected[key] = why or "failed on another rank"
            else:
                landed = (world - 1) * per * d * 4
                table[key] = {"ms_per_exchange": ms, "landed_GBps_per_rank": landed / (ms * 1e-3) / 1e9}
            try:
                cand.close()
            except Exception:  # noqa: BLE001
                pass
            if to_stderr:
                progress({"finished": key, "stage": "first contact"}, rank)
    report = {"dry_exchange": table, "rejected": rejected, "n_gpus": world, "shard_bytes": per * d * 4,
              "dist_backend": backend, "GPU_MAX_HW_QUEUES": hwq,
              "note": "1 MiB shards: latency-dominated rates, a smoke test of every exchange form -- not a bandwidth figure"}
    if backend == "nccl" and rank == 0 and rccl_log_dir:
        from h2gcn_amd.partition import summarize_rccl_log
        report["rccl"] = summarize_rccl_log(rccl_log_dir)
    if to_stderr:
        progress(report, rank)
    elif rank == 0:
        print(json.dumps(report), flush=True)
    return report


USAGE_ERROR = 64   # exit code of a command line that cannot run under any schedule (bench_supervisor does not retry it)


def fail_line(a, why, rank=0, code=2):
    """A bench run that cannot start still prints ONE JSON line (rank 0) -- with "error" and a null value -- and exits
    non-zero, so that whoever parses the output sees why instead of a traceback."""
    if rank == 0:
        print(json.dumps({"metric": "aggregated edges/sec (1+2-hop SpMM)", "value": None, "unit": "edges/s", "n_gpus": a.gpus,
                          "steps": a.steps, "warmup": a.warmup, "error": why}), flush=True)
    raise SystemExit(code)


def progress(obj, rank=0):
    """Put a finished piece of work on record AT ONCE (rank 0): a JSON line on stderr and, under bench_supervisor, in the
    attempt's progress file -- if a later stage takes the job down, what was measured survives in the supervisor's line."""
    if rank != 0:
        return
    obj = dict(obj, attempt=int(os.environ.get("H2GCN_BENCH_ATTEMPT", "0")))
    txt = json.dumps(obj)
    print(txt, file=sys.stderr, flush=True)
    path = os.environ.get("H2GCN_BENCH_PROGRESS")
    if path:
        try:
            with open(path, "a") as f:
                f.write(txt + "\n")
        except OSError:
            pass


def injected_failure(stage, rank):
    """Failure injection for the supervisor tests: H2GCN_BENCH_ABORT_RANK / H2GCN_BENCH_HANG_RANK = <rank> makes that rank
    die with SIGABRT (what the ProcessGroupNCCL watchdog does) / stop responding at `stage`, in the attempts listed in
    H2GCN_BENCH_FAIL_ATTEMPTS (default "0": the first attempt only)."""
    attempts = os.environ.get("H2GCN_BENCH_FAIL_ATTEMPTS", "0").split(",")
    if os.environ.get("H2GCN_BENCH_ATTEMPT", "0") not in attempts or stage != os.environ.get("H2GCN_BENCH_FAIL_STAGE", "calibration"):
        return
    if os.environ.get("H2GCN_BENCH_ABORT_RANK") == str(rank):
        os.abort()
    if os.environ.get("H2GCN_BENCH_HANG_RANK") == str(rank):
        time.sleep(1e6)


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a torch.distributed.run environment: re-exec this command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (one rank
    per GPU, RCCL) and return its exit code.  With fewer than N GPUs visible (and not in the shared-GPU test mode) nothing
    is launched: one JSON line with "error", exit code 2."""
    import socket
    import subprocess

    share = os.environ.get("H2GCN_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < (1 if share else n_ranks):
        ap = argparse.Namespace(gpus=n_ranks, steps=None, warmup=None)
        try:
            fail_line(ap, f"--gpus {n_ranks} but only {n_dev} GPU(s) are visible to this process")
        except SystemExit as e:
            return e.code
    from bench_supervisor import _free_port
    port = _free_port()          # (below the kernel's ephemeral range: see there)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL / the IPC exchange need on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    # every rank torch.distributed.run starts is a bench_supervisor (retry ladder, one line from rank 0's supervisor); this
    # parent is the last line of defence: if the launcher itself dies without a line, the parent prints the error line
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        print(lines[-1], flush=True)
        return r.returncode
    print(json.dumps({"metric": "aggregated edges/sec (1+2-hop SpMM)", "value": None, "unit": "edges/s", "n_gpus": n_ranks,
                      "error": f"torch.distributed.run exited with {r.returncode} without a result line",
                      "stdout_tail": r.stdout[-500:]}), flush=True)
    return r.returncode or 1


EXCHANGE_ORDER = {"allgather": 0, "ipc_kernel": 1, "ipc_engine": 2, "p2p": 3}   # safest first (see main)


def exchange_families(a):
    """The exchange forms of this run, SAFEST FIRST: --exchange, else $H2GCN_BENCH_EXCHANGES, else all four; minus what a retry of
    the supervisor's ladder excludes ($H2GCN_BENCH_EXCLUDE_EXCHANGES: the form that was in flight when an attempt died)."""
    fams = [a.exchange] if a.exchange else os.environ.get("H2GCN_BENCH_EXCHANGES", "allgather,p2p,ipc_engine,ipc_kernel").split(",")
    gone = set(filter(None, os.environ.get("H2GCN_BENCH_EXCLUDE_EXCHANGES", "").split(",")))
    return sorted((f for f in fams if f and f not in gone), key=lambda f: EXCHANGE_ORDER.get(f, 9))


def parse_chunks(spec, d):
    """'2' -> 2 equal chunks; '32+32+64' -> explicit widths."""
    spec = str(spec)
    if "+" in spec:
        return [int(w) for w in spec.split("+")]
    return int(spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="products", choices=sorted(__import__("h2gcn_amd.synth", fromlist=["SHAPES"]).SHAPES))
    ap.add_argument("--d", type=int, default=0, help="feature width (default: the shape's, 128)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--long-row-threshold", type=int, default=0)
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--slice-cols", type=int, default=0, help="feature columns per slice (0 = library heuristic)")
    ap.add_argument("--chunks", default="",
                    help="feature chunks of the pipelined exchange/SpMM: a count ('2') or explicit widths ('32+32+64'). "
                         "Default: 1 on one GPU; on several, the fastest candidate as timed during warm-up")
    ap.add_argument("--exchange", default="", help="allgather | p2p | ipc_engine | ipc_kernel (default: calibrate)")
    ap.add_argument("--no-adjoint", action="store_true", help="skip the (untimed, secondary) adjoint launch figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the live gather/copy ceiling probe")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--pad-rows", action="store_true",
                    help="give X a row stride padded to 32 floats (every row starts on a 128-byte line), as the model's concat buffer has")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the BASELINE configs[2] (arxiv shape) leg that the default products line carries under `secondary` "
                         "(--no-hbm-leg skips it too: no child legs at all)")
    ap.add_argument("--no-hbm-leg", action="store_true",
                    help="skip the HBM-resident leg (products_x6, 12 steps, child process) the default N=1 products line carries")
    ap.add_argument("--dry-exchange", action="store_true",
                    help="N > 1: only smoke-test every exchange form with 1 MiB shards and print a table (no big allocation)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (same thing the documented
        # `python -m torch.distributed.run ... bench.py --gpus N` form does) and pass rank 0's line through
        raise SystemExit(self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        fail_line(a, f"WORLD_SIZE={world} but --gpus {a.gpus}", rank, code=USAGE_ERROR)
    if world > 1 and os.environ.get("H2GCN_BENCH_WORKER") != "1" and not a.dry_exchange:
        # a rank the launcher started (torch.distributed.run -- the driver's form -- or self_launch): only a supervisor.  The
        # real rank runs as its child; a rank that aborts or hangs costs an attempt, not the line (bench_supervisor.py)
        import bench_supervisor
        raise SystemExit(bench_supervisor.supervise(sys.argv[1:], rank, world))
    if os.environ.get("H2GCN_BENCH_FORCE_EXCHANGE"):     # a fallback attempt of the supervisor's ladder overrides the command line
        a.exchange = os.environ["H2GCN_BENCH_FORCE_EXCHANGE"]
    if os.environ.get("H2GCN_BENCH_FORCE_CHUNKS"):
        a.chunks = os.environ["H2GCN_BENCH_FORCE_CHUNKS"]
    if world > 1:
        # the exchange pipeline drives up to world + 1 streams (main, exchange / one per peer); HIP multiplexes streams
        # onto GPU_MAX_HW_QUEUES hardware queues (default 4) -- give every stream its own, before the runtime starts
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(max(world + 2, 4), 12)))
    if not torch.cuda.is_available():
        fail_line(a, "bench.py needs a GPU: h2gcn_amd has no CPU fallback", rank, code=USAGE_ERROR)
    if os.environ.get("H2GCN_SHARE_GPU") != "1" and local_rank >= torch.cuda.device_count():
        fail_line(a, f"rank {rank} has LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible", 0, code=USAGE_ERROR)
    if os.environ.get("H2GCN_SHARE_GPU") == "1":  # test mode: several ranks on one GPU (gloo, or RCCL with a host id per rank)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend, rccl_log_dir = None, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("H2GCN_DIST_BACKEND", "nccl")  # "gloo" only for the shared-GPU test mode
        # a failed rank must not hang its peers forever.  60 s: no collective of this run takes more than a few seconds (RCCL's
        # communicator set-up on 8 GPUs: tens of seconds at most), and every hung exchange form costs this long plus the abort
        timeout_s = float(os.environ.get("H2GCN_DIST_TIMEOUT_S", "60"))
        if backend == "nccl":
            from h2gcn_amd.partition import enable_rccl_debug_log, init_rccl_process_group
            # a collective that exceeds timeout_s: the watchdog tears the process down (SIGABRT) -- the supervisor's ladder
            # takes it from there.  (TORCH_NCCL_BLOCKING_WAIT would turn the time-out into a Python exception, but it also
            # makes every collective block the launching thread, which serialises the exchange/SpMM pipeline: not used.)
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            # ... and promptly: by default the watchdog spends two more minutes collecting a flight-recorder dump between the
            # time-out and the abort (measured: timeout + 120 s); nobody reads that dump here
            os.environ.setdefault("TORCH_NCCL_DUMP_ON_TIMEOUT", "0")
            os.environ.setdefault("TORCH_NCCL_WAIT_TIMEOUT_DUMP_MILSEC", "2000")
            os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "0")
            os.environ.setdefault("TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC", str(int(timeout_s) + 60))   # a watchdog that hangs itself
            # what RCCL chose (channels, transports, algorithm/protocol) goes to a file per process, summarised into the line
            import tempfile
            rccl_log_dir = os.environ.get("H2GCN_BENCH_SCRATCH") or tempfile.mkdtemp(prefix="h2gcn_rccl_", dir="/tmp")
            # (per-collective TUNING lines only in the stand-alone first-contact run: they are file I/O on the launching thread)
            enable_rccl_debug_log(rccl_log_dir, tuning=True if a.dry_exchange else None)
            init_rccl_process_group(device, timeout_s)
        else:
            import datetime
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s))

    from h2gcn_amd import HopPlan, synth
    from h2gcn_amd.partition import PipelinedHopAggregation, block_bounds

    if a.dry_exchange:
        if world < 2:
            raise SystemExit("--dry-exchange needs N > 1 ranks")   # (a usage error, not a measurement: plain message)
        dry_exchange(a, world, rank, device, backend, rccl_log_dir=rccl_log_dir)
        dist.destroy_process_group()
        return
    first_contact = None
    if world > 1 and os.environ.get("H2GCN_BENCH_SKIP_DRY") != "1":
        # first contact with the node BEFORE any big allocation: every exchange form with 1 MiB shards (seconds); the table
        # goes to stderr right away -- if a later stage dies, what worked and what did not is already on record -- and
        # into the line's diagnostics
        try:
            first_contact = dry_exchange(a, world, rank, device, backend, to_stderr=True)
        except Exception as e:  # noqa: BLE001 -- a report, never a reason to lose the measurement
            first_contact = {"error": f"{type(e).__name__}: {e}"}
            progress({"dry_exchange_failed": first_contact["error"]}, rank)
    cfg = synth.SHAPES[a.shape]
    n, d = cfg["n"], (a.d or cfg["d"])
    seeds = (synth.SEED_A1, synth.SEED_A2)
    r0, r1 = block_bounds(n, world, rank)
    degs = synth.hop_degrees(cfg, seeds)
    csr = [synth.synth_hop_rows(degs[k], n, seeds[k], r0, r1, device) for k in range(2)]
    torch.cuda.synchronize()
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n,
                   variant=a.variant, long_row_threshold=a.long_row_threshold, rows_per_wave=a.rows_per_wave,
                   slice_cols=a.slice_cols, build_transpose=not a.no_adjoint)
    x_local = synth.synth_features(d, synth.SEED_X, r0, r1, device)
    if a.pad_rows:
        padded = torch.zeros((r1 - r0, (d + 31) // 32 * 32), dtype=torch.float32, device=device)
        padded[:, :d] = x_local
        x_local = padded[:, :d]
    y = torch.empty((r1 - r0, 2, d), dtype=torch.float32, device=device)
    nnz_local = plan.nnz
    nnz_t = torch.tensor(nnz_local, dtype=torch.int64, device=device)
    if world > 1:
        dist.all_reduce(nnz_t)
    nnz_global = [int(v) for v in nnz_t.tolist()]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_ms(fn, reps):
        """max over ranks of the mean wall time of `fn` (barrier + device sync on both sides)."""
        barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        v = torch.tensor([(time.perf_counter() - t) / reps * 1e3], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return float(v.item())

    def all_ok(flag):
        """True iff `flag` holds on every rank (a candidate must be valid everywhere before it is used)."""
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0)

    # Exchange schedule.  On one GPU: a single fused launch.  On several: the exchange is pipelined against the SpMM
    # in feature chunks; which exchange (one ncclAllGather / P-1 grouped ncclSend+ncclRecv / IPC copy-engine pulls /
    # IPC copy-kernel pulls) and how many chunks pay off depends on the node's RCCL/xGMI rates, so -- unless forced
    # by --exchange/--chunks -- the candidates are timed during warm-up and the fastest is used for the measured
    # steps.  Construction is collectively safe (every rank reports, all agree before a candidate runs); the
    # calibration table and the comm-only / compute-only times are reported as diagnostics.
    rejected, diagnostics = {}, {}
    if world == 1:
        chunks = parse_chunks(a.chunks, d) if a.chunks else 1
        layer = PipelinedHopAggregation(plan, n, d, chunks, device)
        exchange = None
    else:
        exchanges = exchange_families(a)
        # every chunk width builds the canonical summation tree, so the checksum of Y is the same for every candidate and
        # equal to the 1-GPU line's; 32-column chunks (a short exposed head of the exchange) run as masked 64-column slices
        chunk_specs = [a.chunks] if a.chunks else os.environ.get("H2GCN_BENCH_CHUNK_SPECS", "1,2,4,32+32+64").split(",")
        # SAFEST FIRST: one ncclAllGather per chunk and the library's own copy-kernel pulls are timed before anything whose
        # first contact with a second device could take the job down (grouped send/recv last); whatever is on record when
        # a later candidate dies is what the supervisor's line carries
        order = EXCHANGE_ORDER
        pairs = sorted(((ex, spec) for ex in exchanges for spec in chunk_specs),
                       key=lambda es: (0 if es[1] == "2" and order.get(es[0], 9) < 2 else 1, order.get(es[0], 9), chunk_specs.index(es[1])))
        calib_budget = float(os.environ.get("H2GCN_BENCH_CALIBRATION_BUDGET_S", "180"))
        calib_t0 = time.perf_counter()
        # What an exchange must deliver is known exactly: the counter-based generator yields any rows of X on any rank.  Every
        # candidate is checked with TWO different inputs (the second exchange must not serve the first one's bytes again)
        # before it is timed, and its Y against the single-GPU checksum: a fast schedule that moves wrong bytes is rejected.
        x_alt_local = synth.synth_features(d, synth.SEED_X + 1, r0, r1, device)
        want_ck = [N1_CHECKSUMS.get((a.shape, d))]

        def exchange_is_exact(cand):
            """None, or why this rank rejects the candidate.  Every rank takes the same path through the one collective.  The
            reference rows are regenerated one peer block at a time (the generator yields any rows): nothing of the size of X is
            held besides the exchange's own buffers -- row partitioning is there to divide that memory, not to triple it."""
            bad = None
            try:
                for seed, xl in ((synth.SEED_X + 1, x_alt_local), (synth.SEED_X, x_local)):
                    cand(xl, out=y)
                    torch.cuda.synchronize()
                    for q in range(world):
                        q0, q1 = block_bounds(n, world, q)
                        want_blk = synth.synth_features(d, seed, q0, q1, device)
                        for c in range(cand.C):
                            got = cand.full[c][q0:q1]
                            want = want_blk[:, cand.offsets[c]:cand.offsets[c] + cand.widths[c]]
                            if cand.halo is None:
                                if bad is None and not torch.equal(got, want):
                                    bad = f"chunk {c}: rank {q}'s block of the gathered embedding differs from X (input seed {seed})"
                                continue
                            # halo pulls: only the rows this rank's hop matrices name arrive (all of its own block)
                            rows = torch.arange(0, q1 - q0, device=device) if q == rank else cand.halo[q].to(torch.int64)
                            if bad is None and rows.numel() and not torch.equal(got[rows], want[rows]):
                                bad = f"chunk {c}: named rows of rank {q}'s block differ from X (input seed {seed})"
                        del want_blk
                if cand.ipc is not None:
                    cand.ipc.check()
            except Exception as e:  # noqa: BLE001
                bad = f"warm-up: {type(e).__name__}: {e}"
            ck_ = torch.stack([y.view(torch.int32).to(torch.int64).sum(), torch.tensor(0 if bad is None else 1, dtype=torch.int64, device=device)])
            dist.all_reduce(ck_)
            total, n_bad = (int(v) for v in ck_.tolist())
            if n_bad == 0 and want_ck[0] is None:
                want_ck[0] = total                 # shape without a recorded checksum: the first exact exchange sets it
            elif bad is None and n_bad == 0 and total != want_ck[0]:
                bad = f"Y checksum {total} != {want_ck[0]} (the single-GPU result)"
            return bad

        cands = {}
        # (a retry of the supervisor's ladder: candidates an earlier attempt already timed -- all but its fastest -- are not
        # timed again; their figures travel in the line)
        skip_keys = set(filter(None, os.environ.get("H2GCN_BENCH_SKIP_CANDIDATES", "").split(",")))
        if os.environ.get("H2GCN_BENCH_EARLIER_TIMINGS"):
            diagnostics["calibration_ms_per_step_in_an_earlier_attempt"] = json.loads(os.environ["H2GCN_BENCH_EARLIER_TIMINGS"])
        for ex, spec in pairs:
            if f"{ex}/{spec}" in skip_keys:
                continue
            widths = parse_chunks(spec, d)
            if isinstance(widths, int) and (d % widths or (widths > 1 and d // widths < 32)):
                continue
            if isinstance(widths, list) and sum(widths) != d:
                continue
            key = f"{ex}/{spec}"
            if cands and not all_ok(time.perf_counter() - calib_t0 < calib_budget):
                rejected[key] = f"calibration budget ({calib_budget:.0f} s) used up"
                continue
            hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
            if ex == "ipc_engine" and world - 1 > hwq - 2:   # one spin-wait kernel per peer is parked on a hardware queue
                rejected[key] = f"copy-engine pulls need {world - 1} hardware queues besides the main and exchange ones; GPU_MAX_HW_QUEUES = {hwq}"
                continue
            cand, why = None, None
            progress({"starting": key, "stage": "calibration"}, rank)   # which form is in flight, should it take the job down
            if key == os.environ.get("H2GCN_BENCH_FAIL_IN_CANDIDATE"):
                injected_failure("candidate", rank)
            try:
                cand = PipelinedHopAggregation(plan, n, d, widths, device, exchange=ex)
            except Exception as e:  # noqa: BLE001 -- unavailable on this node/backend: not a candidate
                why = f"construct: {type(e).__name__}: {e}"
            if not all_ok(cand is not None):
                rejected[key] = why or "construction failed on another rank"
                if cand is not None:
                    cand.ipc = None if cand.ipc is None else cand.ipc._destroy_local()
                progress({"calibration": key, "rejected": rejected[key], "n_gpus": world}, rank)
                progress({"finished": key, "stage": "calibration"}, rank)
                continue
            why = exchange_is_exact(cand)
            if not all_ok(why is None):
                rejected[key] = why or "warm-up failed on another rank"
                progress({"calibration": key, "rejected": rejected[key], "n_gpus": world}, rank)
                try:
                    cand.close()       # collective: every rank is here
                except Exception:  # noqa: BLE001
                    pass
                progress({"finished": key, "stage": "calibration"}, rank)
                continue
            cands[key] = (timed_ms(lambda: cand(x_local, out=y), 3), cand, ex, spec)
            # on record at once: if a later candidate takes the job down, what was measured survives (stderr + supervisor)
            progress({"calibration": key, "ms_per_step": cands[key][0], "n_gpus": world, "edges_per_s": sum(nnz_global) / (cands[key][0] * 1e-3),
                      "note": "3-step calibration timing, not the K-step measurement"}, rank)
            progress({"finished": key, "stage": "calibration"}, rank)
            if len(cands) == 1:
                injected_failure("calibration", rank)
        del x_alt_local
        if not cands:
            fail_line(a, f"no exchange schedule works on this node: {rejected}", rank)
        best = min(cands, key=lambda k: cands[k][0])
        _, layer, exchange, spec = cands[best]
        chunks = parse_chunks(spec, d)
        diagnostics["exchange"] = exchange
        diagnostics["calibration_ms_per_step"] = {k: v[0] for k, v in cands.items()}
        diagnostics["rejected"] = rejected
        diagnostics["first_contact_dry_exchange"] = first_contact
        # comm-only and compute-only times of the chosen schedule (not part of the metric)
        injected_failure("exchange_only", rank)
        diagnostics["exchange_only_ms"] = timed_ms(layer.exchange_only, 3)
        diagnostics["spmm_only_ms"] = timed_ms(
            lambda: [plan.spmm(layer.full[c][: layer.n_src], out=y[:, :, layer.offsets[c]:layer.offsets[c] + layer.widths[c]])
                     for c in range(layer.C)], 3)
        for k, v in cands.items():  # collective: release the exported buffers of the schedules not chosen
            if v[1] is not layer:
                v[1].close()
        del cands
    for _ in range(max(a.warmup, 1)):
        layer(x_local, out=y)
    torch.cuda.synchronize()
    layer.kernel_events = []   # HIP events around every SpMM launch, on the stream it is launched on

    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        layer(x_local, out=y)      # [staging + exchange of the chunks, overlapped] + `chunks` fused SpMM launches
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if layer.ipc is not None:
        layer.ipc.check()
    if world > 1:
        # the measurement exists from here on: on record before anything else (diagnostics, CPU legs, tear-down) can fail
        progress({"measured": {"metric": "aggregated edges/sec (1+2-hop SpMM)", "value": sum(nnz_global) * a.steps / elapsed, "unit": "edges/s",
                               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
                               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                               "config": {"workload": f"{a.shape} shape row-partitioned over {world} GPUs: synthetic CSR |V|={n}, "
                                                      f"nnz(A1)={nnz_global[0]}, nnz(A2)={nnz_global[1]}, d={d}",
                                          "n_rows": n, "nnz_per_hop": nnz_global, "d": d, "dist_backend": backend,
                                          "parallelism": f"row-partition x{world}, exchange of X per step: {exchange}", "feature_chunks": spec}}}, rank)
        injected_failure("after_timed", rank)
        if backend == "nccl" and rank == 0:
            from h2gcn_amd.partition import summarize_rccl_log
            try:
                diagnostics["rccl"] = summarize_rccl_log(rccl_log_dir) or ["(RCCL wrote no debug file: NCCL_DEBUG_FILE unsupported or overridden)"]
            except Exception as e:  # noqa: BLE001 -- a report
                diagnostics["rccl"] = [f"summary failed: {type(e).__name__}: {e}"]
    # per-step kernel time = sum over the step's `chunks` launches (one launch when chunks == 1); mean for the
    # roofline (comparable with rocprofv3's average), median as SURVEY 8(d) defines t_fused
    per_launch = np.array([s_.elapsed_time(e_) for s_, e_ in layer.kernel_events], dtype=np.float64)
    per_step = per_launch.reshape(a.steps, layer.C).sum(1)
    kern_ms, kern_ms_median = float(per_step.mean()), float(np.median(per_step))
    kern_ms_min, kern_ms_max_step = float(per_step.min()), float(per_step.max())
    kt = torch.tensor([kern_ms], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
    kern_ms_max = float(kt.item())
    per_rank = None
    if world > 1:
        mine = torch.tensor([kern_ms, float(sum(nnz_local)), float(algorithmic_bytes(nnz_local, r1 - r0, d, 2))], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": q, "kernel_ms": float(v[0]), "edges": int(v[1]), "algorithmic_bytes": int(v[2]), "achieved": float(v[2]) / (float(v[0]) * 1e-3) / 1e9,
                     "frac": float(v[2]) / (float(v[0]) * 1e-3) / 1e9 / HBM_PEAK_GBPS} for q, v in enumerate(t_.tolist() for t_ in every)]
    # order-independent fingerprint of the result (same schedule => same bits for any number of ranks)
    ck = y.view(torch.int32).to(torch.int64).sum()
    if world > 1:
        dist.all_reduce(ck)
    y_checksum = int(ck.item())

    edges = sum(nnz_global)
    b_alg = algorithmic_bytes(nnz_local, r1 - r0, d, 2)
    b_min = compulsory_bytes(nnz_local, r1 - r0, n, d, 2)
    achieved = b_alg / (kern_ms * 1e-3) / 1e9
    chunk_label = chunks if isinstance(chunks, int) else "+".join(str(w) for w in chunks)
    traffic = pmc_traffic(a.shape, d, chunk_label, a.slice_cols, world)
    out = {
        "metric": "aggregated edges/sec (1+2-hop SpMM)",
        "value": edges * a.steps / elapsed,
        "unit": "edges/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ({"products": "BASELINE configs[3]", "arxiv": "BASELINE configs[2]"}.get(a.shape, f"(not a BASELINE config) {a.shape}"))
                        + (f"/configs[4] row-partitioned over {world} GPUs" if world > 1 else "")
                        + f": synthetic CSR |V|={n}, nnz(A1)={nnz_global[0]}, nnz(A2)={nnz_global[1]}, d={d}, "
                          "row-normalised values 1/deg, 2-hop CSR supplied (not derived)",
            "n_rows": n, "nnz_per_hop": nnz_global, "d": d,
            "parallelism": f"row-partition x{world}" + (f", exchange of X per step: {exchange}" if world > 1 else ""),
            "kernel_variant": a.variant, "feature_chunks": chunk_label, "dist_backend": backend,
            "diagnostics": diagnostics, "slice_cols": a.slice_cols or "auto", "y_checksum": y_checksum,
            "schedule": plan.schedule(layer.widths[0], ld_src=x_local.stride(0) if world == 1 else None),
            "t_fused_median_ms": kern_ms_median,
            "edges_per_s_from_median_kernel_time": sum(nnz_local) / (kern_ms_median * 1e-3),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None if traffic is None else traffic["bytes_per_launch"],
            "traffic_source": None if traffic is None else
                f"offline rocprofv3 PMC passes of the same command ({traffic.get('source')}); NOT collected by this run",
            "achieved_is": "ALGORITHMIC (no-reuse) bytes per launch / mean kernel time: it counts every gathered feature "
                           "row as memory traffic, so L2 and Infinity-Cache (MALL) hits are included -- it is the "
                           "delivered gather rate, not a DRAM-pin rate; compare with gather_ceiling_GBps",
            "kernel": "h2gcn::spmm_hops_kernel (fused 1+2-hop)"
                      + (f", {layer.C} launches over column chunks {layer.widths}" if layer.C > 1 else ""),
            "kernel_ms": kern_ms, "kernel_ms_median": kern_ms_median, "kernel_ms_min": kern_ms_min, "kernel_ms_max": kern_ms_max_step,
            "kernel_ms_max_over_ranks": kern_ms_max,
            "algorithmic_bytes_per_launch": b_alg,
            "compulsory_bytes_per_launch": b_min,
            "compulsory_frac": b_min / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        },
    }
    if per_rank is not None:
        # `achieved` / `frac` above are rank 0's shard kernel (its algorithmic bytes / its mean launch time); every rank's own
        # figures, and the whole-job delivered rate over the step time (exchange included):
        out["roofline"]["per_rank"] = per_rank
        out["roofline"]["whole_job_GBps_over_step_time"] = sum(r_["algorithmic_bytes"] for r_ in per_rank) / (elapsed / a.steps) / 1e9
    want = N1_CHECKSUMS.get((a.shape, d))
    out["config"]["n1_checksum_expected"] = want
    out["config"]["checksum_matches_n1"] = None if want is None else (y_checksum == want)
    out["config"]["checksum_source"] = "CPU oracle (oracle/fullsize.py: every row of the operands rebuilt on the host, documented summation tree in plain C)"
    if not a.no_adjoint:
        # backward launch dX = sum_k A_k^T dY[:, k, :] on the local shard: half of every training step
        # (reference h2gcn/models/H2GCN.py:66-74); secondary figure, not part of the metric
        dy = synth.synth_features(2 * d, 77, r0, r1, device).view(r1 - r0, 2, d)
        for _ in range(3):
            plan.spmm_t(dy)
        evs = []
        for _ in range(10):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(); plan.spmm_t(dy); e_.record()
            evs.append((s_, e_))
        torch.cuda.synchronize()
        adj = np.array([s_.elapsed_time(e_) for s_, e_ in evs])
        b_adj = sum(z * (4 + 4 + 4 * d) + (n + 1) * 8 for z in nnz_local) + n * d * 4
        adj_ck = None
        if world == 1:      # (row-partitioned: dX is a per-rank partial before the reduce-scatter -- not comparable)
            adj_ck = int(plan.spmm_t(dy).view(torch.int32).to(torch.int64).sum().item())
        want_adj = N1_ADJOINT_CHECKSUMS.get((a.shape, d))
        out["adjoint"] = {"checksum": adj_ck, "checksum_expected_from_oracle": want_adj,
                          "checksum_matches_oracle": None if (adj_ck is None or want_adj is None) else adj_ck == want_adj,
                          "kernel": "h2gcn::spmm_hops_kernel SUM mode on plan-owned A_k^T (local shard)",
                          "kernel_ms": float(adj.mean()), "kernel_ms_median": float(np.median(adj)),
                          "algorithmic_bytes_per_launch": b_adj,
                          "achieved_GBps": b_adj / (adj.mean() * 1e-3) / 1e9,
                          "frac": b_adj / (adj.mean() * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                          "edges_per_s": sum(nnz_local) / (adj.mean() * 1e-3)}
        del dy
    if rank == 0 and world == 1 and not a.no_traffic:
        # release the big operands of this process first? no: the child generates its own; 288 GB hold both
        live = measure_traffic_live(a)
        if live is not None:
            out["roofline"]["traffic"] = live["bytes_per_launch"]
            out["roofline"]["traffic_source"] = live["source"]
            out["roofline"]["traffic_read_bytes"], out["roofline"]["traffic_write_bytes"] = live["read_bytes"], live["write_bytes"]
            out["roofline"]["traffic_over_algorithmic"] = live["bytes_per_launch"] / b_alg
    if rank == 0 and world == 1 and a.shape == "products" and not a.no_hbm_leg and os.environ.get("H2GCN_BENCH_CHILD") != "1":
        del y
        torch.cuda.empty_cache()
        leg = hbm_resident_leg()
        out["roofline"]["hbm_resident"] = leg
        # top-level scalars (a reader that flattens nested objects keeps the DRAM-side evidence with its sample size)
        out["roofline"]["hbm_resident_frac"] = leg.get("frac")
        out["roofline"]["hbm_resident_kernel_ms"] = leg.get("kernel_ms_median")
        out["roofline"]["hbm_resident_kernel_ms_min"] = leg.get("kernel_ms_min")
        out["roofline"]["hbm_resident_kernel_ms_max"] = leg.get("kernel_ms_max")
        out["roofline"]["hbm_resident_steps"] = leg.get("steps")
        out["roofline"]["hbm_resident_checksum_matches_oracle"] = leg.get("checksum_matches_oracle")
    if (rank == 0 and world == 1 and a.shape == "products" and not (a.no_secondary or a.no_hbm_leg) and os.environ.get("H2GCN_BENCH_CHILD") != "1"
            and not any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ)):
        leg2 = secondary_leg("arxiv")
        out["secondary"] = {"arxiv_d128": leg2}
        for k_ in ("kernel_ms", "frac", "traffic_over_algorithmic"):     # scalars a flattening reader keeps
            out["roofline"][f"arxiv_d128_{k_}"] = leg2.get(k_)
    if world > 1:
        layer.close()          # collective: release the exchange buffers before rank 0 spends its seconds on the CPU legs
    if rank == 0 and not a.no_probe:
        # live ceilings of this box at the kernel's gather working set (one column slice of X)
        slice_cols = min(plan.schedule(d)["slice_cols"], d)
        pr = probe_ceilings(n * slice_cols * 4 / 2**20, slice_cols * 4)
        out["roofline"]["gather_ceiling_GBps"] = pr.get("gather_GBps")
        out["roofline"]["peak_achievable"] = max(pr.get("copy_GBps") or 0.0, pr.get("stream_read_GBps") or 0.0) or None
        out["roofline"]["ceilings"] = pr
        out["roofline"].update(peak_source(pr))
        if pr.get("gather_GBps"):
            out["roofline"]["achieved_over_gather_ceiling"] = achieved / pr["gather_GBps"]
    if rank == 0 and world > 1 and out["roofline"]["traffic"] is None:
        out["roofline"]["traffic_source"] = "N > 1: PMC passes are collected on the single-GPU line only (same kernel, same per-row work)"
    if rank == 0 and not a.no_cpu_baseline:
        try:
            # N > 1: rank 0's row block of the operands against the FULL embedding (regenerated: the same counter-based
            # generator, any rows), timed after the measured region like the N = 1 leg
            x_src = x_local if world == 1 else synth.synth_features(d, synth.SEED_X, 0, n, torch.device("cpu"))
            out["cpu_baseline"] = cpu_baseline(csr, x_src, d, a.cpu_seconds)
            if world > 1:
                out["cpu_baseline"]["sample"] += f" [rank 0's row block of the {world}-way partition]"
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    elif rank == 0:
        out["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 1, "kind": "port", "sample": "skipped (--no-cpu-baseline)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        # The other ranks wait here while rank 0 times the CPU legs (tens of seconds) -- on the process group's STORE, not in a
        # collective: a barrier would sit under the watchdog's collective time-out, and a slow host must not turn the CPU
        # baseline of rank 0 into an abort of its peers after the measurement is done.
        try:
            import datetime
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("h2gcn_bench/line_printed", "1")
                for q in range(1, world):      # bounded: the store lives in this process and must outlive the peers' look at it
                    try:
                        store.wait([f"h2gcn_bench/seen{q}"], datetime.timedelta(seconds=5))
                    except Exception:  # noqa: BLE001 -- a peer that is gone already does not need the store
                        break
            else:
                store.wait(["h2gcn_bench/line_printed"], datetime.timedelta(seconds=1800))
                store.set(f"h2gcn_bench/seen{rank}", "1")
        except Exception as e:  # noqa: BLE001 -- the measurement is printed; NO collective from here (rank 0 may be gone already)
            print(f"rank {rank}: final store hand-shake: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 -- whatever takes the run down, whoever parses stdout gets ONE line saying so
        if os.environ.get("RANK", "0") == "0":
            print(json.dumps({"metric": "aggregated edges/sec (1+2-hop SpMM)", "value": None, "unit": "edges/s",
                              "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "error": f"{type(e).__name__}: {e}"[:2000]}), flush=True)
        raise
