"""H2GCN-2 propagation [r2 | r0 | r1] at the products shape: concat-free (fused_propagation) vs layer by layer with
stack/flatten/concat copies (what the reference's interpreter does)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd import HopPlan, GCNLayer, synth
from h2gcn_amd.layers import fused_propagation
cfg = synth.SHAPES["products"]; n = cfg["n"]
dev = torch.device("cuda:0")
degs = [synth.synth_degrees(n, cfg["nnz_per_hop"], s, n) for s in (123, 124)]
csr = [synth.synth_hop_rows(degs[k], n, (123, 124)[k], 0, n, dev) for k in range(2)]
plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
r0 = synth.synth_features(64, 125, 0, n, dev)
layer = GCNLayer()
def generic():
    r1 = layer(plan, r0).flatten(1)
    r2 = layer(plan, r1).flatten(1)
    return torch.cat([torch.cat([r2, r0], 1), r1], 1)     # C1 then C2, as the interpreter executes them
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
a, b = fused_propagation(plan, r0, 2), generic()
print("equal:", torch.equal(a, b))
del a, b
print(f"concat-free {t(lambda: fused_propagation(plan, r0, 2)):.2f} ms   layer-by-layer {t(generic):.2f} ms")
