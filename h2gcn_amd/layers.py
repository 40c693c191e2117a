"""GCNLayer -- torch mirror of the reference's hop-aggregation layer, running on libh2gcn_hip.so.

Reference: ``GCNLayer`` in ``h2gcn/models/_layers.py:54-81``: ``SIGNATURE = ["adjhops", "inputs"]``, stateless,
``layer(adjhops, inputs) -> stack([A_k @ inputs for k in hops], axis=-2)`` of shape ``[N, H_sel, d]``; invoked
by the model interpreter as ``layer(adjhops, inputs)`` (``h2gcn/models/H2GCN.py:318-319``).  Differences by
design: ``adjhops`` is a :class:`h2gcn_amd.hops.HopPlan` (the hop list as one device object) and all selected
hops are aggregated by ONE fused kernel launch that writes the stacked layout directly, so the ``tf.stack`` copy
(and the ``nnz*d > 2**31`` column split of ``_layers.py:65-74``) has no counterpart.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from .hops import HopPlan


class _HopSpMM(torch.autograd.Function):
    """forward: fused multi-hop SpMM; backward: adjoint SpMM on the plan's transposed operands (the gradient
    TF registers for SparseTensorDenseMatMul wrt its dense input; the gradient wrt the adjacency values, which
    TF also computes and the reference never uses, is not produced)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, plan: HopPlan, hops):
        ctx.plan = plan
        ctx.hops = hops
        return plan.spmm(x, hops=hops)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        return ctx.plan.spmm_t(grad_out, hops=ctx.hops), None, None


def hop_spmm(adjhops: HopPlan, inputs: torch.Tensor, hops: Optional[Iterable[int]] = None) -> torch.Tensor:
    """Functional form of :class:`GCNLayer`: ``[n_cols, d] -> [n_rows, H_sel, d]``, differentiable wrt ``inputs``."""
    if not isinstance(adjhops, HopPlan):
        raise TypeError(f"adjhops must be a HopPlan, got {type(adjhops).__name__}")
    sel = None if hops is None else tuple(sorted(set(int(h) for h in hops)))
    if inputs.requires_grad and torch.is_grad_enabled():
        return _HopSpMM.apply(inputs, adjhops, sel)
    return adjhops.spmm(inputs, hops=sel)


class GCNLayer(torch.nn.Module):
    """``GCNLayer(hops=None)(adjhops, inputs) -> [N, H_sel, d]`` (reference ``_layers.py:54-81``).

    ``hops``: optional set of hop indices to keep (the ``G0`` / ``G0_1`` forms of the network-setup DSL,
    ``h2gcn/models/__init__.py:88-95``); ``None`` keeps every hop of ``adjhops``.  Unknown indices are ignored
    like the reference's ``if ind in self.hops`` filter does -- unless nothing is left, which raises.
    """

    SIGNATURE = ["adjhops", "inputs"]

    def __init__(self, hops=None):
        super().__init__()
        self.hops = None if hops is None else set(int(h) for h in hops)

    def forward(self, adjhops: HopPlan, inputs: torch.Tensor) -> torch.Tensor:
        sel = None
        if self.hops is not None:
            sel = [h for h in range(adjhops.n_hops) if h in self.hops]
            if not sel:
                raise ValueError(f"GCNLayer(hops={sorted(self.hops)}) selects none of the {adjhops.n_hops} hops")
        return hop_spmm(adjhops, inputs, sel)

    def extra_repr(self) -> str:
        return f"hops={None if self.hops is None else sorted(self.hops)}"
