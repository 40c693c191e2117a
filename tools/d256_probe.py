"""Experiment (GPU box): where does d = 256 lose its roofline fraction?  Products shape, forward launch.
Row-major X with ld = 256 (1 KiB stride) vs padded ld = 288 / 320, every slice width, vs slice-major storage
(n_slices contiguous [N, w] blocks, the repack timed separately)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd import HopPlan, synth  # noqa: E402

cfg = synth.SHAPES["products"]
n = cfg["n"]
d = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
degs = [synth.synth_degrees(n, cfg["nnz_per_hop"], s, n) for s in (123, 124)]
csr = [synth.synth_hop_rows(degs[k], n, (123, 124)[k], 0, n, dev) for k in range(2)]
nnz = sum(int(c[1].numel()) for c in csr)
b_alg = nnz * (8 + 4 * d) + 2 * (n + 1) * 8 + n * 2 * d * 4


def t(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def report(label, ms):
    print(f"{label:70s} {ms:8.2f} ms  frac {b_alg / (ms * 1e-3) / 8e12:.3f}", flush=True)


x = synth.synth_features(d, 125, 0, n, dev)
y = torch.empty((n, 2, d), device=dev)
for sc in (0, 64, 128, 256):
    if sc and d % sc:
        continue
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n, slice_cols=sc)
    report(f"row-major ld={d}, slice_cols={sc or 'auto'}", t(lambda: plan.spmm(x, out=y)))
    for ld in (d + 32, d + 64):
        big = torch.empty((n, ld), device=dev)
        big[:, :d].copy_(x)
        xv = big[:, :d]
        report(f"row-major ld={ld} (padded), slice_cols={sc or 'auto'}", t(lambda: plan.spmm(xv, out=y)))
        del big, xv
    # padded output as well (Y row stride 2*d*4 bytes is a power of two too)
    ybig = torch.empty((n, 2, d + 32), device=dev)
    report(f"row-major ld={d}, padded Y (ld_hop {d + 32}), slice_cols={sc or 'auto'}", t(lambda: plan.spmm(x, out=ybig[:, :, :d])))
    del ybig
    del plan

plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
for w in (64, 128):
    nb = d // w
    xb = torch.empty((nb, n, w), device=dev)
    report(f"repack row-major -> slice-major [{nb}][N][{w}] (torch copy)", t(lambda: xb.copy_(x.view(n, nb, w).permute(1, 0, 2))))
    report(f"slice-major blocks of {w}, {nb} launches, row-major Y",
           t(lambda: [plan.spmm(xb[b], out=y[:, :, b * w:(b + 1) * w]) for b in range(nb)]))
    yb = torch.empty((nb, n, 2, w), device=dev)
    report(f"slice-major blocks of {w}, {nb} launches, slice-major Y", t(lambda: [plan.spmm(xb[b], out=yb[b]) for b in range(nb)]))
    del xb, yb
