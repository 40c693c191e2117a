#!/usr/bin/env python3
"""tools/placement_x6_contig.py -- experiment: products_x6 forward launch with X and Y in (a) ordinary torch allocations,
(b) physically contiguous VRAM (hipDeviceMallocContiguous through a torch MemPool), re-allocated alternately in one process."""
import statistics, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from h2gcn_amd import HopPlan, synth  # noqa: E402

def timed(fn, n=8):
    fn(); fn()
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    return statistics.median(s.elapsed_time(e) for s, e in ev)

dev = torch.device("cuda", 0)
cfg = synth.SHAPES[sys.argv[1] if len(sys.argv) > 1 else "products_x6"]
which = sys.argv[2] if len(sys.argv) > 2 else "xy"          # which operands go to the contiguous pool: x, y or xy
n, d = cfg["n"], cfg["d"]
seeds = (synth.SEED_A1, synth.SEED_A2)
degs = synth.hop_degrees(cfg, seeds)
csr = [synth.synth_hop_rows(degs[k], n, seeds[k], 0, n, dev) for k in range(2)]
torch.cuda.synchronize(); torch.cuda.empty_cache()
plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
alloc = torch.cuda.memory.CUDAPluggableAllocator(str(ROOT / "build" / "contig_alloc.so"), "contig_alloc", "contig_free")
for trial in range(10):
    contig = trial % 2 == 1
    pool = torch.cuda.MemPool(alloc.allocator()) if contig else None
    spacer = torch.empty((1 << 20) * (3 + 499 * trial), dtype=torch.uint8, device=dev)
    def make(kind):
        if kind == "x":
            src = synth.synth_features(d, synth.SEED_X, 0, n, dev)
            if contig and "x" in which:
                with torch.cuda.use_mem_pool(pool):
                    t = torch.empty_like(src)
                t.copy_(src); del src
                return t
            return src
        if contig and "y" in which:
            with torch.cuda.use_mem_pool(pool):
                return torch.empty((n, 2, d), dtype=torch.float32, device=dev)
        return torch.empty((n, 2, d), dtype=torch.float32, device=dev)
    x, y = make("x"), make("y")
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    ms = timed(lambda: plan.spmm(x, out=y))
    print(f"trial {trial}: {'CONTIGUOUS ' + which if contig else 'default      '}  median launch {ms:8.2f} ms", flush=True)
    del x, y, spacer, pool
    torch.cuda.synchronize(); torch.cuda.empty_cache()
