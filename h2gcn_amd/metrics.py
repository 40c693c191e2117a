"""Masked softmax cross-entropy / masked accuracy on the library's one-pass kernels (``csrc/metrics.hip``).

Reference: ``h2gcn/models/_metrics.py:8-25`` -- ``masked_softmax_cross_entropy`` and ``masked_accuracy`` weight the per-node
cross-entropy / argmax agreement with ``mask / mean(mask)`` and take the mean, i.e. a weighted sum with the row weights
``mask / sum(mask)``.  ``train_step`` needs one such loss (and its gradient), ``test_step`` three accuracies and two losses of the
same logits (``h2gcn/models/H2GCN.py:66-74, 77-107``).  Here a *set* is ``(labels [N, C], row weights [N])``; all sets of a
call share one read of the logits.  GPU only, fp32, ``C <= 64``: anything else is the caller's business (``models/_metrics.py``
keeps the plain torch expressions for those cases); the HIP library is required -- there is no CPU path in this module.

Divergence on non-finite logits (documented, deliberate): the reference multiplies every row's term by its mask weight, so a
NaN / Inf logit in a row the mask EXCLUDES still poisons the result (``0 * NaN``); the kernels skip rows of zero weight, so only
rows a mask includes can make its loss / accuracy non-finite.
"""
import ctypes as C
from typing import Sequence, Tuple

import torch

from . import _capi


def supported(preds: torch.Tensor) -> bool:
    return preds.is_cuda and preds.dtype == torch.float32 and preds.dim() == 2 and 1 <= preds.shape[1] <= 64 and preds.stride(1) == 1


def _rows(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if t.dim() == 2 and t.stride(1) != 1:
        t = t.contiguous()
    if t.dim() == 1 and t.stride(0) != 1:
        t = t.contiguous()
    return t


def masked_metrics(preds: torch.Tensor, labels: Sequence[torch.Tensor], weights: Sequence[torch.Tensor],
                   want_accuracy: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(loss [M], acc [M])`` for the M sets ``(labels[m], weights[m])``; no gradient (see :func:`masked_cross_entropy`)."""
    if not supported(preds):
        raise ValueError("masked_metrics: logits must be a CUDA fp32 [N, C <= 64] tensor with unit column stride")
    m = len(labels)
    if m != len(weights) or not 1 <= m <= _capi.METRICS_MAX_SETS:
        raise ValueError(f"masked_metrics: 1..{_capi.METRICS_MAX_SETS} (labels, weights) sets expected, got {len(labels)} / {len(weights)}")
    n, c = preds.shape
    ys = [_rows(y, "labels") for y in labels]
    ws = [_rows(w, "weights") for w in weights]
    for y, w in zip(ys, ws):
        if y.shape != (n, c) or w.shape != (n,) or y.device != preds.device or w.device != preds.device:
            raise ValueError(f"masked_metrics: labels {tuple(y.shape)} / weights {tuple(w.shape)} do not match logits {tuple(preds.shape)}")
    lib = _capi.lib()
    out = torch.empty((2, m), dtype=torch.float32, device=preds.device)
    ws_buf = torch.empty(max(8, int(lib.h2gcn_masked_metrics_workspace_bytes(n))), dtype=torch.uint8, device=preds.device)
    y_ptrs = (C.c_void_p * m)(*[y.data_ptr() for y in ys])
    ldys = (C.c_int64 * m)(*[y.stride(0) for y in ys])
    w_ptrs = (C.c_void_p * m)(*[w.data_ptr() for w in ws])
    with torch.cuda.device(preds.device):
        stream = torch.cuda.current_stream(preds.device).cuda_stream
        _capi.check(lib.h2gcn_masked_metrics_f32(
            C.c_void_p(preds.data_ptr()), preds.stride(0), n, c, m, y_ptrs, ldys, w_ptrs, C.c_void_p(out[0].data_ptr()),
            C.c_void_p(out[1].data_ptr()) if want_accuracy else None, C.c_void_p(ws_buf.data_ptr()), ws_buf.numel(), C.c_void_p(stream)))
    return out[0], out[1]


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, labels, weights):
        loss, _ = masked_metrics(preds, [labels], [weights], want_accuracy=False)
        ctx.save_for_backward(preds, _rows(labels, "labels"), _rows(weights, "weights"))
        return loss[0]

    @staticmethod
    def backward(ctx, grad):
        preds, labels, weights = ctx.saved_tensors
        n, c = preds.shape
        g = grad.to(torch.float32).reshape(1).contiguous()
        dz = torch.empty((n, c), dtype=torch.float32, device=preds.device)
        with torch.cuda.device(preds.device):
            stream = torch.cuda.current_stream(preds.device).cuda_stream
            _capi.check(_capi.lib().h2gcn_masked_ce_backward_f32(
                C.c_void_p(preds.data_ptr()), preds.stride(0), n, c, C.c_void_p(labels.data_ptr()), labels.stride(0),
                C.c_void_p(weights.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(dz.data_ptr()), dz.stride(0), C.c_void_p(stream)))
        return dz, None, None


def masked_cross_entropy(preds: torch.Tensor, labels: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """``sum_n weights[n] * CE(preds[n], labels[n])`` as a scalar, differentiable with respect to ``preds``."""
    if not supported(preds):
        raise ValueError("masked_cross_entropy: logits must be a CUDA fp32 [N, C <= 64] tensor with unit column stride")
    if preds.requires_grad and torch.is_grad_enabled():
        return _MaskedCE.apply(preds, labels, weights)
    return masked_metrics(preds, [labels], [weights], want_accuracy=False)[0][0]
