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
    for spec in ((32, 32, 64), (64, 32, 32)):          # the explicit chunk orders of bench.py's calibration
        offs = [sum(spec[:i]) for i in range(len(spec))]
        xb = [x[:, o:o + w].contiguous() for o, w in zip(offs, spec)]
        res["+".join(map(str, spec))] = t(lambda: [plan.spmm(xb[c], out=y[:, :, offs[c]:offs[c] + spec[c]]) for c in range(len(spec))])
    print(f"P={P}: rows/rank {r1 - r0}, per-rank SpMM ms by chunks { {k: round(v, 3) for k, v in res.items()} };  ideal 1/P of the P=1 time = {res[1] if P == 1 else base / P:.2f}")
    if P == 1:
        base = res[1]
    del plan, csr
