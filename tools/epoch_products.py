"""One full-batch H2GCN-2 training step at the products shape on ONE GPU (synthetic labels/features): forward
(dense embedding -> concat-free propagation -> classifier), masked CE + L2, backward (adjoint SpMMs), Adam."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd import HopPlan, synth
from h2gcn_amd.models import parse_network_setup
from h2gcn_amd.models.H2GCN import H2GCN, make_optimizer
cfg = synth.SHAPES["products"]; n = cfg["n"]; F, C = 100, 47
HIDDEN = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
dev = torch.device("cuda:0")
degs = [synth.synth_degrees(n, cfg["nnz_per_hop"], s, n) for s in (123, 124)]
csr = [synth.synth_hop_rows(degs[k], n, (123, 124)[k], 0, n, dev) for k in range(2)]
t0 = time.perf_counter()
plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n, build_transpose=True)
torch.cuda.synchronize(); print(f"plan with device-built transposes: {time.perf_counter() - t0:.2f} s")
feats = synth.synth_features(F, 5, 0, n, dev)
labels = torch.nn.functional.one_hot(torch.randint(0, C, (n,), device=dev), C).float()
mask = torch.rand(n, device=dev) < 0.1
model = H2GCN(parse_network_setup(f"M{HIDDEN}-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO", C), input_dim=F, n_hops=2, sparse_input=False,
              l2_regularize_weight=5e-4, fused_classifier="--stock-classifier" not in sys.argv).to(dev)
opt = make_optimizer("adam", model.parameters(), 0.01)
def step():
    model.train(); opt.zero_grad(set_to_none=True)
    loss = model.loss(model(None, feats, plan), labels, mask); loss.backward(); opt.step(); return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
edges = sum(plan.nnz)
print(f"hidden {HIDDEN}: train step {dt:.1f} ms  (loss {l.item():.4f}); 2 G-layers fwd + 2 adjoints = {4 * edges} edge visits -> {4 * edges / dt / 1e6:.2f}e9 edges/s; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
