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

    def create(self, csr, n_cols):
        n_rows = csr[0][0].numel() - 1
        arr = C.c_void_p * len(csr)
        opts = PlanOpts(struct_size=C.sizeof(PlanOpts), flags=PLAN_BUILD_TRANSPOSE)
        stream = torch.cuda.current_stream().cuda_stream
        self.check(self.L.h2gcn_plan_create(len(csr), n_rows, n_cols, arr(*[c[0].data_ptr() for c in csr]), arr(*[c[1].data_ptr() for c in csr]),
                                            arr(*[c[2].data_ptr() for c in csr]), C.byref(opts), C.c_void_p(stream), C.byref(self.plan)))

    def destroy(self):
        if self.plan.value:
            self.L.h2gcn_plan_destroy(self.plan)
            self.plan = C.c_void_p()
        self.ws = {}

    def _opts(self, adjoint, src, ld_row, ld_hop, d):
        """Same scratch rule as HopPlan.spmm / spmm_t (the library asks for a slice-major copy of some sources)."""
        key = (adjoint, d)
        if key not in self.ws:
            nbytes = int(self.L.h2gcn_spmm_workspace_bytes(self.plan, 0, adjoint, C.c_void_p(src.data_ptr()), ld_row, ld_hop, d))
            self.ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=src.device) if nbytes else None
        ws = self.ws[key]
        if ws is None:
            return None
        return LaunchOpts(struct_size=C.sizeof(LaunchOpts), flags=0, workspace=ws.data_ptr(), workspace_bytes=ws.numel(), bias=None)

    def forward(self, x, y):
        d = x.shape[1]
        o = self._opts(0, x, x.stride(0), 0, d)
        self.check(self.L.h2gcn_spmm_hops_opts_f32(self.plan, 0, C.c_void_p(x.data_ptr()), x.stride(0), d, C.c_void_p(y.data_ptr()), y.stride(0),
                                                   y.stride(1), C.byref(o) if o is not None else None,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def adjoint(self, dy, dx):
        d = dy.shape[2]
        o = self._opts(1, dy, dy.stride(0), dy.stride(1), d)
        self.check(self.L.h2gcn_spmm_hops_T_opts_f32(self.plan, 0, C.c_void_p(dy.data_ptr()), dy.stride(0), dy.stride(1), d, C.c_void_p(dx.data_ptr()),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True, help="name=path ...")
    ap.add_argument("--shapes", nargs="+", default=["products_x6", "hbm16m", "lowdeg"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--baseline", default=None, help="library the relative differences are quoted against (default: the second)")
    a = ap.parse_args()
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
