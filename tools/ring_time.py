"""Timing of the device-side exact-k-hop ring construction (csrc/rings.hip) -- SURVEY.md 8(f) rank 4.
  * the syn-products fixture graph (n = 10 000, 2.8M-nonzero 2-hop ring), next to scipy's host SpGEMM path;
  * a 1M-node synthetic graph (symmetrised, mean degree ~8, degrees clipped so that the 2-hop ring fits).
Run on the GPU box; wrap in rocprofv3 --kernel-trace --stats for the per-kernel lines."""
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from h2gcn_amd import operands as po, synth  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, out


def report(name, rp, ci, n, host_adj=None):
    a = (rp, ci)
    t2, rings = timed(lambda: po.exact_hop_rings_device(rp, ci, n, 2))
    nnz = [int(r[1].numel()) for r in rings]
    tn, vals = timed(lambda: po.normalize_pattern_device(rings[2], n, po.SYM_NORMALIZED))
    # bytes the ring kernels must move at least: the A rows read during the expansion (once per pass) + the output
    cand = int((rp[1:] - rp[:-1])[ci.long()].sum())          # sum over edges (i,j) of deg(j) = candidates of ring 2
    print(f"{name}: n={n} nnz(A)={nnz[1]} candidates={cand} nnz(ring2)={nnz[2]}  rings(1..2) {t2*1e3:.2f} ms  "
          f"normalise ring2 {tn*1e3:.2f} ms  -> {cand*2/t2/1e9:.2f} G candidate-marks/s (two passes)", flush=True)
    if host_adj is not None:
        t = time.perf_counter()
        po.build_adj_norm_hops(host_adj, ("1", "2"), po.SYM_NORMALIZED)
        print(f"{name}: host scipy SpGEMM path (reference's way) {1e3*(time.perf_counter()-t):.1f} ms", flush=True)
    del a


from conftest import load_syn_products_golden  # noqa: E402

a, _, _ = load_syn_products_golden()
adj = po.remove_self_loops(a)
adj.sort_indices()
report("syn-products", torch.from_numpy(adj.indptr.astype(np.int64)).to(dev), torch.from_numpy(adj.indices.astype(np.int32)).to(dev),
       adj.shape[0], host_adj=adj)

# 1M nodes: directed synthetic rows, symmetrised on the device
n = 1_000_000
deg = synth.synth_degrees(n, 4 * n, 11, n, clip=40.0)
rp, ci, _ = synth.synth_hop_rows(deg, n, 11, 0, n, dev)
rows = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
keys = torch.unique(torch.cat([rows * n + ci.long(), ci.long() * n + rows]))
keys = keys[(keys // n) != (keys % n)]
r2 = keys // n
rp2 = torch.zeros(n + 1, dtype=torch.int64, device=dev)
rp2[1:] = torch.cumsum(torch.bincount(r2, minlength=n), 0)
ci2 = (keys % n).to(torch.int32)
del rows, keys, r2
report("synthetic-1M", rp2, ci2, n)
