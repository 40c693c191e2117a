"""Fused dropout + output Dense (csrc/classifier.hip) vs the stock torch pair at the products shape: [N, 7*hidden] x [7*hidden, 47],
forward + backward, hip-event times.  usage: python tools/classifier_time.py [hidden ...]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd.layers import DropoutDense

dev = torch.device("cuda:0")
n, c = 2_400_000, 47


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for hidden in [int(a) for a in sys.argv[1:]] or [64]:
    k = 7 * hidden
    x = torch.randn((n, k), device=dev, requires_grad=True)
    wgt = torch.randn((n, c), device=dev)
    fused = DropoutDense(k, c, True, 0.5).to(dev).train()
    lin_w = fused.kernel.detach().clone().requires_grad_(True)
    lin_b = fused.bias.detach().clone().requires_grad_(True)

    def step_fused():
        x.grad = None
        (fused(x) * wgt).sum().backward()

    def step_stock():
        x.grad = None
        z = torch.nn.functional.dropout(x, 0.5, True) @ lin_w + lin_b
        (z * wgt).sum().backward()

    def fwd_fused():
        with torch.no_grad():
            fused(x)

    def fwd_stock():
        with torch.no_grad():
            torch.nn.functional.dropout(x, 0.5, True) @ lin_w + lin_b

    gb = n * k * 4 / 1e9
    tf_, ts_ = timed(fwd_fused), timed(fwd_stock)
    bf_, bs_ = timed(step_fused), timed(step_stock)
    print(f"hidden {hidden}: X = [{n}, {k}] ({gb:.2f} GB), C = {c}")
    print(f"  forward         fused {tf_:7.3f} ms ({gb / tf_ * 1e3:6.0f} GB/s of X)   stock dropout + matmul {ts_:7.3f} ms")
    print(f"  forward+backward fused {bf_:7.3f} ms                          stock {bs_:7.3f} ms   (incl. the loss stand-in: z*w, sum)")
    del x, wgt
    torch.cuda.empty_cache()
