"""Per-rank SpMM time of a row-partitioned run, emulated on ONE GPU: rank 0's row block of the products-shape graph
(1/P of the rows, global column space) against the full X, for the chunkings the exchange pipeline can use.
Gives the compute side of the N-GPU time model (the exchange side needs the real node)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd import HopPlan, synth
from h2gcn_amd.partition import block_bounds
cfg = synth.SHAPES["products"]; n, d = cfg["n"], 128
dev = torch.device("cuda:0")
degs = [synth.synth_degrees(n, cfg["nnz_per_hop"], s, n) for s in (123, 124)]
x = synth.synth_features(d, 125, 0, n, dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for P in (1, 2, 4, 8):
    r0, r1 = block_bounds(n, P, 0)
    csr = [synth.synth_hop_rows(degs[k], n, (123, 124)[k], r0, r1, dev) for k in range(2)]
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
    y = torch.empty((r1 - r0, 2, d), device=dev)
    res = {}
    for C in (1, 2, 4, 8):
        dc = d // C
        xb = [x[:, c * dc:(c + 1) * dc].contiguous() for c in range(C)]   # what the all-gather delivers
        res[C] = t(lambda: [plan.spmm(xb[c], out=y[:, :, c * dc:(c + 1) * dc]) for c in range(C)])
    print(f"P={P}: rows/rank {r1 - r0}, per-rank SpMM ms by chunks {res};  ideal 1/P of 17.45 = {17.45 / P:.2f}")
    del plan, csr
