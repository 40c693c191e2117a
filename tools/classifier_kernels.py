"""Per-entry-point times of the dropout + dense kernels (csrc/classifier.hip) at the products shape, through the C ABI:
forward, backward-data only, backward-weights only.  usage: python tools/classifier_kernels.py [K] [C] [N] [keep_prob]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from h2gcn_amd import _capi

k = int(sys.argv[1]) if len(sys.argv) > 1 else 448
c = int(sys.argv[2]) if len(sys.argv) > 2 else 47
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2_400_000
keep = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
dev = torch.device("cuda:0")
lib = _capi.lib()
import os
if os.environ.get("H2GCN_CLS_SMALL_ROWS") is not None:      # rows at or below which the small-operand kernels serve the call (0: never)
    lib.h2gcn_dropout_dense_small_rows(int(os.environ["H2GCN_CLS_SMALL_ROWS"]))
x = torch.randn((n, k), device=dev)
w = torch.randn((k, c), device=dev) * 0.05
b = torch.randn((c,), device=dev)
g = torch.randn((n, c), device=dev)
z = torch.empty((n, c), device=dev)
dx = torch.empty((n, k), device=dev)
dw = torch.empty((k, c), device=dev)
ws = torch.empty(int(lib.h2gcn_dropout_dense_workspace_bytes(n, k, c)), dtype=torch.uint8, device=dev)
step = torch.zeros((), dtype=torch.int64, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def fwd():
    _capi.check(lib.h2gcn_dropout_dense_f32(P(x), k, n, k, P(w), c, P(b), keep, 7, P(step), P(z), c, P(ws), ws.numel(), stream))


def bwd(want_dx, want_dw):
    _capi.check(lib.h2gcn_dropout_dense_backward_f32(P(x), k, n, k, P(w), c, P(g), c, keep, 7, P(step), P(dx) if want_dx else None, k,
                                                     P(dw) if want_dw else None, P(ws), ws.numel(), stream))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


gb = n * k * 4 / 1e9
flop = 2.0 * n * k * ((c + 15) // 16 * 16)
for name, fn in (("forward", fwd), ("backward dX", lambda: bwd(True, False)), ("backward dW (+ reduction)", lambda: bwd(False, True))):
    t = timed(fn, reps=200 if n <= 200_000 else 10)
    print(f"N={n} K={k} C={c} keep={keep}  {name:26s} {t:7.3f} ms   {gb / t * 1e3:6.0f} GB/s of the [N, K] operand   {flop / t / 1e9:6.1f} TFLOP/s fp32 MFMA")
