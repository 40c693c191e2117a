"""Masked loss / accuracy (reference ``h2gcn/models/_metrics.py:8-25``): the mask is normalised to sum 1, so both
are means over the masked nodes; labels are one-hot rows (all-zero rows contribute zero loss).  On the GPU (fp32 logits with
at most 64 classes) both run on the library's one-pass kernels (``h2gcn_amd/metrics.py``); other inputs take the plain
torch expressions below."""
import torch

from .. import metrics as _fused


def _weights(mask: torch.Tensor) -> torch.Tensor:
    m = mask.to(torch.float32)
    return m / m.sum()


def masked_softmax_cross_entropy(preds: torch.Tensor, labels: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    if _fused.supported(preds):
        return _fused.masked_cross_entropy(preds, labels, _weights(mask))
    loss = -(labels * torch.log_softmax(preds, dim=1)).sum(dim=1)
    return (loss * _weights(mask)).sum()


def masked_accuracy(preds: torch.Tensor, labels: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    if _fused.supported(preds):
        return _fused.masked_metrics(preds, [labels], [_weights(mask)])[1][0]
    correct = (preds.argmax(dim=1) == labels.argmax(dim=1)).to(torch.float32)
    return (correct * _weights(mask)).sum()
