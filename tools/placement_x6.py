#!/usr/bin/env python3
"""tools/placement_x6.py -- within ONE process: does the DRAM-resident launch time follow the ADDRESSES of its operands?
profiles/r05_products_x6_process_to_process.txt shows +-6 % between processes and +-0.1 % within one.  Here the plan (CSR,
lists) is built once; X and Y are then re-allocated several times at shifted addresses (a spacer allocation of varying size in
front, allocator cache emptied in between) and the launch is timed each time; then the CSR itself is re-generated too.
Micro-benchmark, not product."""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd import HopPlan, synth  # noqa: E402


def timed(fn, n=8):
    fn(); fn()
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    return statistics.median(s.elapsed_time(e) for s, e in ev)


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "products_x6"
    dev = torch.device("cuda", 0)
    cfg = synth.SHAPES[shape]
    n, d = cfg["n"], cfg["d"]
    seeds = (synth.SEED_A1, synth.SEED_A2)
    for rebuild in range(2):
        degs = synth.hop_degrees(cfg, seeds)
        spacer0 = torch.empty((1 << 20) * (17 + 301 * rebuild), dtype=torch.uint8, device=dev)     # shifts where the CSR lands
        csr = [synth.synth_hop_rows(degs[k], n, seeds[k], 0, n, dev) for k in range(2)]
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
        print(f"== CSR build {rebuild}: colidx at {[hex(c[1].data_ptr()) for c in csr]}")
        for trial in range(5):
            spacer = torch.empty((1 << 20) * (3 + 997 * trial), dtype=torch.uint8, device=dev)
            x = synth.synth_features(d, synth.SEED_X, 0, n, dev)
            y = torch.empty((n, 2, d), dtype=torch.float32, device=dev)
            ms = timed(lambda: plan.spmm(x, out=y))
            print(f"  trial {trial}: X at {hex(x.data_ptr())}  Y at {hex(y.data_ptr())}  median launch {ms:8.2f} ms")
            del x, y, spacer
            torch.cuda.synchronize(); torch.cuda.empty_cache()
        del plan, csr, spacer0
        torch.cuda.synchronize(); torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
