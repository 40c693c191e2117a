#!/usr/bin/env python3
"""tools/ab_dram_regimes.py -- interleaved same-box, same-process A/B of several builds of libh2gcn_hip.so on the DRAM-side
regimes (is the drift of the HBM-resident figure a regression or the box?).

    tools/ab_dram_regimes.py --libs r03=build/ab/lib_r03.so r04=build/ab/lib_r04.so r05=build/ab/lib_r05.so \
        --shapes products_x6 hbm16m lowdeg --rounds 5 --launches 10 > profiles/r05_ab_r03_vs_r04_dram_regimes.txt

Every library is dlopen'ed into THIS process (its own ctypes handle, its own plan) and driven through the C ABI on the SAME
operands at the SAME addresses: one synthetic graph per shape (h2gcn_amd/synth.py, generated once), one X, one dY, one Y /
dX.  Per round the libraries take turns (A B C, A B C, ...), forward then adjoint, 2 warm-up + `launches` timed launches
each, HIP events on the launch stream; the figure per (library, round) is the MEDIAN launch.  Micro-benchmark, not product.
"""
import argparse
import ctypes as C
import json
import statistics
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from h2gcn_amd import synth  # noqa: E402
from h2gcn_amd._capi import LaunchOpts, PlanOpts  # noqa: E402  (struct layouts only; no library is loaded through _capi)

PLAN_BUILD_TRANSPOSE = 0x1


class Lib:
    def __init__(self, name, path):
        self.name, self.path = name, str(path)
        L = self.L = C.CDLL(self.path)
        L.h2gcn_abi_version.restype = C.c_int
        L.h2gcn_last_error.restype = C.c_char_p
        L.h2gcn_plan_create.restype = C.c_int
        L.h2gcn_plan_create.argtypes = [C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(PlanOpts), C.c_void_p, C.POINTER(C.c_void_p)]
        L.h2gcn_plan_destroy.restype = None
        L.h2gcn_plan_destroy.argtypes = [C.c_void_p]
        L.h2gcn_spmm_workspace_bytes.restype = C.c_size_t
        L.h2gcn_spmm_workspace_bytes.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int32]
        L.h2gcn_spmm_hops_opts_f32.restype = C.c_int
        L.h2gcn_spmm_hops_opts_f32.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64,
                                               C.POINTER(LaunchOpts), C.c_void_p]
        L.h2gcn_spmm_hops_T_opts_f32.restype = C.c_int
        L.h2gcn_spmm_hops_T_opts_f32.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                                 C.POINTER(LaunchOpts), C.c_void_p]
        self.abi = L.h2gcn_abi_version()
        self.plan = C.c_void_p()
        self.ws = {}

    def check(self, st):
        if st < 0:
            raise RuntimeError(f"{self.name}: {self.L.h2gcn_last_error().decode()} (status {st})")

    def create(self, csr, n_cols, transpose=True, variant=0, rows_per_wave=0, slice_cols=0):
        n_rows = csr[0][0].numel() - 1
        arr = C.c_void_p * len(csr)
        opts = PlanOpts(struct_size=C.sizeof(PlanOpts), flags=PLAN_BUILD_TRANSPOSE if transpose else 0, variant=variant,
                        rows_per_wave=rows_per_wave, slice_cols=slice_cols)
        stream = torch.cuda.current_stream().cuda_stream
        self.check(self.L.h2gcn_plan_create(len(csr), n_rows, n_cols, arr(*[c[0].data_ptr() for c in csr]), arr(*[c[1].data_ptr() for c in csr]),
                                            arr(*[c[2].data_ptr() for c in csr]), C.byref(opts), C.c_void_p(stream), C.byref(self.plan)))

    def destroy(self):
        if self.plan.value:
            self.L.h2gcn_plan_destroy(self.plan)
            self.plan = C.c_void_p()
        self.ws = {}

    def _opts(self, adjoint, src, ld_row, ld_hop, d, mask=0):
        """Same scratch rule as HopPlan.spmm / spmm_t (the library asks for a slice-major copy of some sources)."""
        key = (adjoint, d, mask)
        if key not in self.ws:
            nbytes = int(self.L.h2gcn_spmm_workspace_bytes(self.plan, mask, adjoint, C.c_void_p(src.data_ptr()), ld_row, ld_hop, d))
            self.ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=src.device) if nbytes else None
        ws = self.ws[key]
        if ws is None:
            return None
        return LaunchOpts(struct_size=C.sizeof(LaunchOpts), flags=0, workspace=ws.data_ptr(), workspace_bytes=ws.numel(), bias=None)

    def forward(self, x, y, mask=0):
        d = x.shape[1]
        o = self._opts(0, x, x.stride(0), 0, d, mask)
        self.check(self.L.h2gcn_spmm_hops_opts_f32(self.plan, mask, C.c_void_p(x.data_ptr()), x.stride(0), d, C.c_void_p(y.data_ptr()), y.stride(0),
                                                   y.stride(1), C.byref(o) if o is not None else None,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def adjoint(self, dy, dx, mask=0):
        d = dy.shape[2]
        o = self._opts(1, dy, dy.stride(0), dy.stride(1), d, mask)
        self.check(self.L.h2gcn_spmm_hops_T_opts_f32(self.plan, mask, C.c_void_p(dy.data_ptr()), dy.stride(0), dy.stride(1), d, C.c_void_p(dx.data_ptr()),
                                                     dx.stride(0), C.byref(o) if o is not None else None,
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def timed(fn, warm, launches):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(launches):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    return [s.elapsed_time(e) for s, e in evs]


def forward_vs_adjoint(a):
    """--matrix (round 6): why is the DRAM-resident forward 15 points below its own adjoint?  One process, one set
    of operands, every configuration a (library, plan options, direction, hop mask) tuple taking turns per round:
      F2 / F1  forward, both hops / hop 0 only          A2 / A1  adjoint, both hops / hop 0 only
      *_nostore  the same launches of a build that computes every sum but does not issue the output stream (-DH2GCN_AB_NO_STORES)
      F2 v2 / rpwN   forward with the cross-segment index prefetch (variant 2) / N rows per wave
      F2 / A2 of builds with other launch bounds / load-batch depths (w7b8, w6b4, w7b4 = waves per SIMD, deepest batch)
    Rates are quoted on each launch's OWN algorithmic bytes (one-hop launches: that hop's nonzeros + its share of the output)."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    libs = dict(spec.split("=", 1) for spec in a.libs)
    shape = a.shapes[0]
    cfg = synth.SHAPES[shape]
    n, d = cfg["n"], cfg["d"]
    seeds = (synth.SEED_A1, synth.SEED_A2)
    degs = synth.hop_degrees(cfg, seeds)
    csr = [synth.synth_hop_rows(degs[k], n, seeds[k], 0, n, dev) for k in range(2)]
    x = synth.synth_features(d, synth.SEED_X, 0, n, dev)
    dy = synth.synth_features(2 * d, 77, 0, n, dev).view(n, 2, d)
    y = torch.empty((n, 2, d), dtype=torch.float32, device=dev)
    y1 = torch.empty((n, 1, d), dtype=torch.float32, device=dev)
    dx = torch.empty((n, d), dtype=torch.float32, device=dev)
    nnz = [int(c[0][-1]) for c in csr]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    def b_alg(hops, out_hops):
        return sum(nnz[k] * (8 + 4 * d) + (n + 1) * 8 for k in hops) + n * out_hops * d * 4

    plans = {}

    def plan(lib, transpose, **kw):
        key = (lib, transpose, tuple(sorted(kw.items())))
        if key not in plans:
            lb = Lib(lib, libs[lib])
            lb.create(csr, n, transpose=transpose, **kw)
            plans[key] = lb
        return plans[key]

    base = a.baseline or next(iter(libs))
    configs = []   # (label, callable, algorithmic bytes)
    pb = plan(base, True)
    configs.append(("F2 forward, hops 0+1", lambda: pb.forward(x, y), b_alg((0, 1), 2)))
    configs.append(("F1 forward, hop 0 only", lambda: pb.forward(x, y1, mask=1), b_alg((0,), 1)))
    configs.append(("A2 adjoint, hops 0+1", lambda: pb.adjoint(dy, dx), b_alg((0, 1), 1)))
    configs.append(("A1 adjoint, hop 0 only", lambda: pb.adjoint(dy, dx, mask=1), b_alg((0,), 1)))
    if "nostore" in libs:
        pn = plan("nostore", True)
        configs.append(("F2 nostore (sums computed, no output stream)", lambda: pn.forward(x, y), b_alg((0, 1), 0)))
        configs.append(("F1 nostore", lambda: pn.forward(x, y1, mask=1), b_alg((0,), 0)))
        configs.append(("A2 nostore", lambda: pn.adjoint(dy, dx), b_alg((0, 1), 0)))
    for label, kw in () if a.lite else (("F2 variant 2 (index prefetch across segments)", dict(variant=2)), ("F2 rows_per_wave 1", dict(rows_per_wave=1)),
                      ("F2 rows_per_wave 2", dict(rows_per_wave=2)), ("F2 rows_per_wave 7", dict(rows_per_wave=7)),
                      ("F2 slice 128", dict(slice_cols=128))):
        pv = plan(base, False, **kw)
        configs.append((label, (lambda q: (lambda: q.forward(x, y)))(pv), b_alg((0, 1), 2)))
    if a.transposed:
        # the same launches on the TRANSPOSED hop matrices: rows of A^T have Poisson(50) lengths (A's columns are uniform) and the
        # gathered rows have A's Pareto row degrees as their popularity -- the forward on A^T reads like the adjoint on A and vice versa
        def transpose(c):
            rp, ci, va = c
            counts = rp[1:] - rp[:-1]
            rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), counts)
            key = ci.to(torch.int64) * n + rows
            del rows
            key, order = torch.sort(key)
            col_t = (key % n).to(torch.int32)
            row_t = torch.div(key, n, rounding_mode="floor")
            del key
            rp_t = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            rp_t[1:] = torch.cumsum(torch.bincount(row_t, minlength=n), 0)
            del row_t
            va_t = va[order].contiguous()
            return rp_t, col_t.contiguous(), va_t
        csr_t = [transpose(c) for c in csr]
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        lt = Lib(base + "_T", libs[base])
        lt.create(csr_t, n, transpose=True)
        plans[("T",)] = lt
        nnz_t = [int(c[0][-1]) for c in csr_t]
        assert nnz_t == nnz
        configs.append(("F2 forward on A^T (Poisson rows, Pareto-popular gathers)", lambda: lt.forward(x, y), b_alg((0, 1), 2)))
        configs.append(("A2 adjoint on A^T (= SUM walk over A's Pareto rows)", lambda: lt.adjoint(dy, dx), b_alg((0, 1), 1)))
        if "nostore" in libs:
            ltn = Lib("nostore_T", libs["nostore"])
            ltn.create(csr_t, n, transpose=True)
            plans[("Tn",)] = ltn
            configs.append(("F2 nostore on A^T", lambda: ltn.forward(x, y), b_alg((0, 1), 0)))
            configs.append(("A2 nostore on A^T", lambda: ltn.adjoint(dy, dx), b_alg((0, 1), 0)))
    for name in libs:
        if name in (base, "nostore"):
            continue
        pl = plan(name, True)
        configs.append((f"F2 build {name}", (lambda q: (lambda: q.forward(x, y)))(pl), b_alg((0, 1), 2)))
        configs.append((f"A2 build {name}", (lambda q: (lambda: q.adjoint(dy, dx)))(pl), b_alg((0, 1), 1)))
    torch.cuda.synchronize()
    print(f"# {shape}: |V| = {n}, nnz = {nnz}, d = {d}; X = {n * d * 4 / 1e9:.2f} GB; device {torch.cuda.get_device_name(0)}")
    print(f"# libraries: {libs}; baseline build: {base}")
    print(f"# {a.rounds} rounds x [2 warm-up + {a.launches} timed launches] per configuration, configurations take turns; median launch per round")
    per = {c[0]: [] for c in configs}
    for r in range(a.rounds):
        for label, fn, _ in configs:
            per[label].append(statistics.median(timed(fn, 2, a.launches)))
    out = {}
    for label, _, b in configs:
        v = per[label]
        med = statistics.median(v)
        out[label] = {"median_ms": med, "min_ms": min(v), "max_ms": max(v), "algorithmic_GB": b / 1e9, "GBps": b / med / 1e6, "frac": b / med / 1e6 / 8000}
        print(f"{label:<52} {med:9.3f} ms  [{min(v):8.3f} .. {max(v):8.3f}]   {b / 1e9:8.2f} GB  {b / med / 1e6:7.0f} GB/s  {b / med / 1e6 / 8000:.3f} of 8 TB/s")
    print("\n" + json.dumps({"matrix": out}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True, help="name=path ...")
    ap.add_argument("--shapes", nargs="+", default=["products_x6", "hbm16m", "lowdeg"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--baseline", default=None, help="library the relative differences are quoted against (default: the second)")
    ap.add_argument("--transposed", action="store_true", help="--matrix: also run forward / adjoint on the transposed hop matrices")
    ap.add_argument("--lite", action="store_true", help="--matrix without the plan-option configurations (builds only)")
    ap.add_argument("--matrix", action="store_true", help="forward-vs-adjoint matrix on the first of --shapes (see forward_vs_adjoint)")
    a = ap.parse_args()
    if a.matrix:
        return forward_vs_adjoint(a)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    libs = [Lib(*spec.split("=", 1)) for spec in a.libs]
    base = a.baseline or libs[min(1, len(libs) - 1)].name
    print(f"# interleaved A/B on one box, one process, same operands; libraries: " + ", ".join(f"{lb.name} (ABI {lb.abi}, {lb.path})" for lb in libs))
    print(f"# {a.rounds} rounds x [2 warm-up + {a.launches} timed launches], median launch per round; device: {torch.cuda.get_device_name(0)}")
    summary = {}
    for shape in a.shapes:
        cfg = synth.SHAPES[shape]
        n, d = cfg["n"], cfg["d"]
        seeds = (synth.SEED_A1, synth.SEED_A2)
        degs = synth.hop_degrees(cfg, seeds)
        csr = [synth.synth_hop_rows(degs[k], n, seeds[k], 0, n, dev) for k in range(2)]
        x = synth.synth_features(d, synth.SEED_X, 0, n, dev)
        dy = synth.synth_features(2 * d, 77, 0, n, dev).view(n, 2, d)
        y = torch.empty((n, 2, d), dtype=torch.float32, device=dev)
        dx = torch.empty((n, d), dtype=torch.float32, device=dev)
        nnz = [int(c[0][-1]) for c in csr]
        b_fwd = sum(z * (8 + 4 * d) + (n + 1) * 8 for z in nnz) + n * 2 * d * 4
        b_adj = sum(z * (8 + 4 * d) + (n + 1) * 8 for z in nnz) + n * d * 4
        torch.cuda.synchronize()
        torch.cuda.empty_cache()     # the libraries allocate with hipMalloc: give back what the generator's temporaries left cached
        for lb in libs:
            lb.create(csr, n)
        torch.cuda.synchronize()
        print(f"\n## {shape}: |V| = {n}, nnz = {nnz}, d = {d}; X = {n * d * 4 / 1e9:.2f} GB")
        # the bits must agree before the times are compared
        sums = {}
        for lb in libs:
            lb.forward(x, y)
            lb.adjoint(dy, dx)
            torch.cuda.synchronize()
            sums[lb.name] = (int(y.view(torch.int32).to(torch.int64).sum()), int(dx.view(torch.int32).to(torch.int64).sum()))
        print("checksums (Y, dX) identical across libraries:", len(set(sums.values())) == 1, sums[libs[0].name])
        per = {lb.name: {"fwd": [], "adj": []} for lb in libs}
        for r in range(a.rounds):
            for lb in libs:
                f = statistics.median(timed(lambda: lb.forward(x, y), 2, a.launches))
                t = statistics.median(timed(lambda: lb.adjoint(dy, dx), 2, a.launches))
                per[lb.name]["fwd"].append(f)
                per[lb.name]["adj"].append(t)
                print(f"round {r + 1} {lb.name:>5}  fwd {f:9.3f} ms ({b_fwd / f / 1e6 / 8000:.3f} of 8 TB/s)   adj {t:9.3f} ms ({b_adj / t / 1e6 / 8000:.3f})")
        summary[shape] = {}
        for direction, b in (("fwd", b_fwd), ("adj", b_adj)):
            ref = statistics.median(per[base][direction])
            for lb in libs:
                v = per[lb.name][direction]
                med = statistics.median(v)
                summary[shape][f"{lb.name}/{direction}"] = {"median_ms": med, "min_ms": min(v), "max_ms": max(v), "frac": b / med / 1e6 / 8000,
                                                           "vs_" + base: med / ref - 1.0}
                print(f"{shape:>12} {direction} {lb.name:>5}: median of rounds {med:9.3f} ms  [{min(v):.3f} .. {max(v):.3f}]  frac {b / med / 1e6 / 8000:.3f}  "
                      f"{(med / ref - 1.0) * 100:+.2f} % vs {base}")
        for lb in libs:
            lb.destroy()
        del csr, x, dy, y, dx
        torch.cuda.empty_cache()
    print("\n" + json.dumps({"summary": summary}))


if __name__ == "__main__":
    main()
