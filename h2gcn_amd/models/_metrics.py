"""Masked loss / accuracy (reference ``h2gcn/models/_metrics.py:8-25``): the mask is normalised to sum 1, so both
are means over the masked nodes; labels are one-hot rows (all-zero rows contribute zero loss)."""
import torch


def masked_softmax_cross_entropy(preds: torch.Tensor, labels: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    loss = -(labels * torch.log_softmax(preds, dim=1)).sum(dim=1)
    m = mask.to(torch.float32)
    return (loss * (m / m.sum())).sum()


def masked_accuracy(preds: torch.Tensor, labels: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    correct = (preds.argmax(dim=1) == labels.argmax(dim=1)).to(torch.float32)
    m = mask.to(torch.float32)
    return (correct * (m / m.sum())).sum()
