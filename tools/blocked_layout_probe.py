"""Experiment: d=256 features stored slice-major (4 contiguous [N,64] blocks) vs row-major with strided slices."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd import HopPlan, synth
cfg = synth.SHAPES["products"]; n = cfg["n"]; d = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
degs = [synth.synth_degrees(n, cfg["nnz_per_hop"], s, n) for s in (123, 124)]
csr = [synth.synth_hop_rows(degs[k], n, (123, 124)[k], 0, n, dev) for k in range(2)]
plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
x = synth.synth_features(d, 125, 0, n, dev)
y = torch.empty((n, 2, d), device=dev)
nb = d // 64
xb = x.view(n, nb, 64).permute(1, 0, 2).contiguous()
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("row-major, in-launch slices (auto):", t(lambda: plan.spmm(x, out=y)))
print("row-major, strided views, %d launches:" % nb, t(lambda: [plan.spmm(x[:, b*64:(b+1)*64], out=y[:, :, b*64:(b+1)*64]) for b in range(nb)]))
print("slice-major contiguous blocks, %d launches:" % nb, t(lambda: [plan.spmm(xb[b], out=y[:, :, b*64:(b+1)*64]) for b in range(nb)]))
yb = torch.empty((nb, n, 2, 64), device=dev)
print("slice-major in AND out:", t(lambda: [plan.spmm(xb[b], out=yb[b]) for b in range(nb)]))
