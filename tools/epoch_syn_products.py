"""One full-batch H2GCN-2 training step + evaluation on the syn-products fixture graph (BASELINE configs[1]: |V| = 10 000, the
reference generator's graph with h = 0.2; exact-2-hop ring built on the device; d = 100 synthetic class-conditional features,
hidden 64, no feature normalisation as in experiments/h2gcn/configs/syn-products/h2gcn.json): ms per epoch, eager loop.
usage: python tools/epoch_syn_products.py [epochs]      (H2GCN_VARIANT / H2GCN_ROWS_PER_WAVE ... select the walk, see tools/README.md)"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_syn_products_golden
from h2gcn_amd import HopPlan, operands
from h2gcn_amd.models import parse_network_setup
from h2gcn_amd.models.H2GCN import H2GCN, make_optimizer
dev = torch.device("cuda:0")
a, labels, _ = load_syn_products_golden()
n, C, F = a.shape[0], int(labels.max()) + 1, 100
rps, cis, vas, _ = operands.build_adj_norm_hops_device(operands.remove_self_loops(a), ("1", "2"), operands.SYM_NORMALIZED, dev)
plan = HopPlan(rps, cis, vas, n, build_transpose=True)
rng = np.random.default_rng(0)
feats = torch.from_numpy((rng.normal(size=(C, F))[labels] + rng.normal(size=(n, F))).astype(np.float32)).to(dev)
y = torch.nn.functional.one_hot(torch.from_numpy(labels.astype(np.int64)), C).float().to(dev)
mask = torch.from_numpy(rng.random(n) < 0.25).to(dev)
model = H2GCN(parse_network_setup("M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO", C), input_dim=F, n_hops=2, sparse_input=False, l2_regularize_weight=5e-4).to(dev)
opt = make_optimizer("adam", model.parameters(), 0.01)
opt.set_l2([l.kernel for l in model.regularized], model.l2)
def epoch():
    model.train(); opt.zero_grad(set_to_none=True)
    loss = model.data_loss(model(None, feats, plan), y, mask); loss.backward(); opt.step()
    model.eval()
    with torch.no_grad():
        model(None, feats, plan)
    return loss
for _ in range(5): epoch()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(E): l = epoch()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / E * 1e3
cls = plan.segment_classes(64)
print(f"syn-products fixture (n = {n}, nnz {plan.nnz}): {dt:.3f} ms per eager epoch (train step + evaluation); walk {plan.schedule(64)['segment_walk']!r}, "
      f"classes {[h['segments'] for h in cls['per_hop']]}, loss {l.item():.4f}")
