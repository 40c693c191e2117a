"""Experiment: gather source = column window of a wider row-major buffer (concat-free propagation) vs contiguous."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd import HopPlan, synth
cfg = synth.SHAPES["products"]; n = cfg["n"]
dev = torch.device("cuda:0")
degs = [synth.synth_degrees(n, cfg["nnz_per_hop"], s, n) for s in (123, 124)]
csr = [synth.synth_hop_rows(degs[k], n, (123, 124)[k], 0, n, dev) for k in range(2)]
plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for d, W, off in ((128, 448, 320), (128, 256, 0), (128, 384, 128), (256, 256, 0), (256, 288, 0), (256, 272, 0), (256, 260, 0), (128, 132, 0)):
    buf = torch.rand((n, W), device=dev)
    xc = buf[:, off:off + d].contiguous()
    y = torch.empty((n, 2, d), device=dev)
    ybuf = torch.empty((n, W if W >= 2 * d else 2 * d), device=dev)
    print(f"d={d} src window of [N,{W}] at col {off}: strided {t(lambda: plan.spmm(buf[:, off:off + d], out=y)):.2f} ms   contiguous {t(lambda: plan.spmm(xc, out=y)):.2f} ms")
