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

import os

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
        if hasattr(adjhops, "aggregate"):  # partition.ShardedHops: all-gather + local SpMM (+ reduce-scatter backward)
            return adjhops.aggregate(inputs, None if self.hops is None else sorted(self.hops))
        if self.hops is not None:
            sel = [h for h in range(adjhops.n_hops) if h in self.hops]
            if not sel:
                raise ValueError(f"GCNLayer(hops={sorted(self.hops)}) selects none of the {adjhops.n_hops} hops")
        return hop_spmm(adjhops, inputs, sel)

    def extra_repr(self) -> str:
        return f"hops={None if self.hops is None else sorted(self.hops)}"


class SparseDropout(torch.nn.Module):
    """Dropout on the VALUES of the sparse feature operand (reference ``SparseDropout``, ``_layers.py:7-19``: mask =
    ``floor(keep_prob + U[0,1))`` per stored value, survivors divided by ``keep_prob``; reached when ``D`` precedes the
    first dense layer, ``H2GCN.py:250-257``).  The operand is a 1-hop :class:`HopPlan`; its pattern is static, so the
    layer writes the masked values into one persistent buffer (dropped entries as explicit zeros -- the same product as
    ``tf.sparse.retain``) and points the plan at it (``HopPlan.set_values``, which also refreshes the transposed
    operand the kernel gradient needs).

    Eval mode -- a DOCUMENTED DIVERGENCE from the reference: its ``SparseDropout.call(self, input)`` takes no
    ``training`` argument and ``H2GCN.call`` invokes ``layer(inputs)`` (``H2GCN.py:323``), so Keras never switches it
    off: the reference also drops sparse feature values during evaluation (its dense ``Dropout`` layers are switched
    off).  That looks unintended; here the layer is inactive in eval mode by default, like every other dropout.
    ``at_eval=True`` (CLI ``--sparse_dropout_at_eval``) reproduces the reference's behaviour.

    The plan is shared (``tensors["features"]``): :meth:`restore` puts the original values back; the training step calls
    it once the backward pass -- which still needs the dropped operand for ``dW = X_drop^T g`` -- is done, so nobody
    observes dropped values between steps."""

    def __init__(self, drop_prob: float, at_eval: bool = False):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.at_eval = bool(at_eval)
        self._buf = None
        self._plan = None

    def restore(self) -> None:
        """Point the plan back at its original values (no-op if they are in place)."""
        plan = self._plan
        if plan is None:
            return
        orig = getattr(plan, "_values_before_dropout", None)
        if orig is not None and plan.vals[0] is not orig:
            plan.set_values(0, orig)

    def forward(self, plan: HopPlan) -> HopPlan:
        if not isinstance(plan, HopPlan) or plan.n_hops != 1:
            raise TypeError("SparseDropout expects the sparse feature operand as a 1-hop HopPlan")
        self._plan = plan
        orig = getattr(plan, "_values_before_dropout", None)
        if not (self.training or self.at_eval) or self.drop_prob <= 0.0:
            self.restore()
            return plan
        if orig is None:
            orig = plan._values_before_dropout = plan.vals[0]
        if self._buf is None or self._buf.shape != orig.shape or self._buf.device != orig.device:
            self._buf = torch.empty_like(orig)
        keep = 1.0 - self.drop_prob
        mask = torch.floor(torch.rand_like(orig) + keep)
        torch.mul(orig, mask / keep, out=self._buf)
        plan.set_values(0, self._buf)
        return plan


class _SparseDenseFused(torch.autograd.Function):
    """``act(X_sp @ W + b)`` in ONE launch: bias and ReLU are the store epilogue of the hop kernel (reference
    ``SparseDense.call``, ``_layers.py:45-52``).  Backward: ``g = dY * (Y > 0)``; ``dW = X_sp^T g`` (adjoint launch),
    ``db = sum_rows g``."""

    @staticmethod
    def forward(ctx, kernel, bias, plan, relu):
        y = plan.spmm(kernel, bias=bias, relu=relu)[:, 0, :]
        ctx.plan, ctx.relu, ctx.has_bias = plan, relu, bias is not None
        ctx.save_for_backward(y if relu else torch.empty(0, device=y.device))
        return y

    @staticmethod
    def backward(ctx, grad):
        (y,) = ctx.saved_tensors
        g = grad * (y > 0) if ctx.relu else grad
        d_kernel = ctx.plan.spmm_t(g.contiguous().unsqueeze(1))
        return d_kernel, (g.sum(0) if ctx.has_bias else None), None, None


class SparseDense(torch.nn.Module):
    """Sparse features x dense kernel (reference ``SparseDense``, ``h2gcn/models/_layers.py:22-52``): the feature
    embedding ``X_sp[N, F] @ W[F, units]`` (+ bias, + activation).  The sparse operand is a 1-hop
    :class:`HopPlan` holding the feature matrix in CSR, so the product runs on the same HIP kernel as the hop
    aggregation; the kernel gradient ``X_sp^T @ dY`` is the plan's adjoint launch."""

    def __init__(self, input_dim: int, output_dim: int, use_bias: bool = False, activation=None):
        super().__init__()
        self.kernel = torch.nn.Parameter(torch.empty(input_dim, output_dim))
        torch.nn.init.xavier_uniform_(self.kernel)  # keras default: glorot_uniform
        self.bias = torch.nn.Parameter(torch.zeros(output_dim)) if use_bias else None
        self.activation = activation

    def forward(self, inputs: HopPlan) -> torch.Tensor:
        if not isinstance(inputs, HopPlan) or inputs.n_hops != 1:
            raise TypeError("SparseDense expects the sparse feature operand as a 1-hop HopPlan")
        if inputs.n_cols != self.kernel.shape[0]:
            raise ValueError(f"features have {inputs.n_cols} columns, kernel has {self.kernel.shape[0]} rows")
        relu = self.activation in ("relu", torch.relu, torch.nn.functional.relu) or isinstance(self.activation, torch.nn.ReLU)
        if self.bias is not None or relu:   # bias / ReLU fused into the store of the sparse product
            if inputs.has_transpose or not (self.kernel.requires_grad and torch.is_grad_enabled()):
                return _SparseDenseFused.apply(self.kernel, self.bias, inputs, relu)
        out = hop_spmm(inputs, self.kernel)[:, 0, :]
        if self.bias is not None:
            out = out + self.bias
        if self.activation is not None:
            out = torch.relu(out) if relu else self.activation(out)
        return out


class ConcatLayer(torch.nn.Module):
    """``concat([inputs] + [tagged[name] for name in tags])`` on the last axis (reference ``ConcatLayer``,
    ``_layers.py:83-96``).  Tagged outputs are taken in the order they were produced (the reference iterates the
    kwargs dict, i.e. insertion order), not in the order the tags are listed."""

    def __init__(self, tags, axis: int = -1, addInputs: bool = True):
        super().__init__()
        self.tags = list(tags)
        self.axis = axis
        self.addInputs = addInputs

    def forward(self, *args, **tagged) -> torch.Tensor:
        selected = [v for name, v in tagged.items() if name in self.tags]
        return torch.cat((list(args) if self.addInputs else []) + selected, dim=self.axis)


class SliceLayer(torch.nn.Module):
    """Column slice of the input or of a tagged output (reference ``SliceLayer``, ``_layers.py:107-116``)."""

    def __init__(self, loadTag, sliceObj, **_):
        super().__init__()
        self.tag = loadTag
        self.sliceObj = sliceObj

    def forward(self, inputs, **tagged):
        if self.tag:
            inputs = tagged[self.tag]
        return inputs[:, self.sliceObj]


class _FusedPropagation(torch.autograd.Function):
    """K rounds of hop aggregation written straight into the final concat buffer.

    H2GCN-K's representation is ``[r_K | r_0 | r_1 | ... | r_{K-1}]`` with ``r_k = flatten(GCNLayer(r_{k-1}))``
    (reference: ``G``/``V`` layers ``h2gcn/models/H2GCN.py:266-272,318-319`` followed by ``C<tag>`` concats,
    ``_layers.py:90-96``; concat order = running input first, then tags in production order).  The reference
    materialises every ``r_k`` (``tf.stack``), flattens, then copies everything twice more through ``ConcatV2``.
    Here one ``[N, W]`` buffer is allocated; each round's fused SpMM reads ``r_{k-1}`` from its column slot
    (row stride ``W``) and writes ``r_k`` into its own slot through the kernel's output strides -- no stack, no
    flatten, no concat copy.  Backward walks the rounds in reverse with the adjoint launch, reading the incoming
    gradient slots in place.
    """

    @staticmethod
    def forward(ctx, r0: torch.Tensor, plan: HopPlan, rounds: int, out: Optional[torch.Tensor] = None, reuse: bool = False,
                private_grad: bool = False):
        """``private_grad``: the caller guarantees that the gradient tensor this node's backward receives is a temporary
        nobody else reads (see :func:`fused_propagation`).  ``out``: a caller-owned ``[N, W]`` contiguous buffer to fill instead of a fresh one.  ``reuse``: ``out`` ALREADY
        holds the propagation of this very ``r0`` (see :func:`fused_propagation`) -- nothing is computed, the buffer only
        enters the autograd graph (the backward needs none of the forward's values: the rounds are linear)."""
        n, w0 = r0.shape
        H = plan.n_hops
        widths = [w0 * H ** k for k in range(rounds + 1)]
        total = sum(widths)
        # column offsets: r_K first, then r_0 .. r_{K-1}
        off = [0] * (rounds + 1)
        off[rounds] = 0
        pos = widths[rounds]
        for k in range(rounds):
            off[k] = pos
            pos += widths[k]
        if out is None:
            buf = concat_buffer(n, total, r0.device)
        else:
            if out.shape != (n, total) or out.dtype != torch.float32 or out.device != r0.device or not out.is_contiguous():
                raise ValueError(f"fused_propagation: out must be a contiguous float32 [{n}, {total}] tensor on {r0.device}")
            buf = out.view(n, total)   # a new tensor object on the caller's storage: autograd marks IT as this node's output
        if not reuse:
            buf[:, off[0]:off[0] + w0].copy_(r0)
            for k in range(1, rounds + 1):
                src = buf[:, off[k - 1]:off[k - 1] + widths[k - 1]]
                dst = buf[:, off[k]:off[k] + widths[k]].unflatten(1, (H, widths[k - 1]))
                plan.spmm(src, out=dst)
        ctx.plan, ctx.rounds, ctx.widths, ctx.off = plan, rounds, widths, off
        ctx.private_grad = bool(private_grad)
        return buf

    @staticmethod
    def backward(ctx, grad: torch.Tensor):
        plan, K, widths, off = ctx.plan, ctx.rounds, ctx.widths, ctx.off
        H = plan.n_hops
        g_k = grad[:, off[K]:off[K] + widths[K]]  # d r_K: a view, read in place by the adjoint launch
        # The adjoint of round k is ADDED to the slot of r_{k-1} inside the incoming gradient itself (the library's
        # accumulate flag: the `+=` rides on the adjoint's store) -- but ONLY when the caller has vouched that this tensor is a
        # temporary of its own (`private_grad`: models.H2GCN sets it when the buffer's sole consumer is a layer whose
        # backward allocates its input gradient, DropoutDense / Dense / Dropout).  Autograd itself gives no such guarantee
        # (a user-supplied `out.backward(G)`, a hook or `retain_grad` on the buffer, a gradient shared with another node),
        # so the default is a fresh tensor per round plus one `+=` pass.  Also needed: an ordinary dense gradient.
        dense = (ctx.private_grad and grad.is_contiguous() and isinstance(plan, HopPlan)
                 and os.environ.get("H2GCN_BACKWARD_IN_PLACE", "1") != "0")
        for k in range(K, 0, -1):
            slot = grad[:, off[k - 1]:off[k - 1] + widths[k - 1]]
            # (not where the plain launch would run in the in-tile short-row mode -- short-throughout operands, where that mode is
            # worth more than the saved pass; list-driven / wave-walk launches lose nothing on the accumulating tile walk)
            in_place = dense and plan.schedule(widths[k - 1], ld_src=grad.stride(0), adjoint=True)["segment_walk"] != "lane group per segment (short rows)"
            if in_place:
                g_k = plan.spmm_t(g_k.unflatten(1, (H, widths[k - 1])), out=slot, accumulate=True)
            else:
                g_prev = plan.spmm_t(g_k.unflatten(1, (H, widths[k - 1])))
                g_prev += slot
                g_k = g_prev
        return g_k, None, None, None, None, None


def concat_buffer(n_rows: int, width: int, device) -> torch.Tensor:
    """The ``[n_rows, width]`` fp32 concat buffer, contiguous.  (Padding its row stride to a cache line was tried in round
    3 and dropped: the hop launches gain < 2 % -- the slots inside a row still start off-line, so they take the
    scratch-copy schedule either way -- while the stock dropout / classifier kernels that consume the buffer fall off
    their vectorised paths on a non-contiguous view: +2.8 ms per products-scale step at ``--hidden 100``,
    ``profiles/r03_train_step_hidden100.txt``.)"""
    return torch.empty((n_rows, width), dtype=torch.float32, device=device)


def fused_propagation(plan: HopPlan, r0: torch.Tensor, rounds: int, out: Optional[torch.Tensor] = None,
                      reuse: bool = False, private_grad: bool = False) -> torch.Tensor:
    """``[r_K | r_0 | ... | r_{K-1}]`` for ``rounds = K`` aggregation rounds, without intermediate copies.

    ``out`` / ``reuse``: the propagation is a deterministic function of ``(plan, r0)``, and an epoch of the reference evaluates
    the model right after every update (``run_experiments.py:44-61``: ``train_step`` then ``test_step``) -- so the next
    epoch's training forward recomputes exactly the buffer the evaluation has just produced whenever nothing stochastic
    precedes the propagation (H2GCN's default setup: the only dropout sits behind it).  A caller that knows this
    (``models.H2GCN``) lets the evaluation fill a persistent buffer (``out=``) and hands the same buffer to the training
    forward with ``reuse=True``: same bits, one propagation per epoch instead of two.

    ``private_grad``: promise that the gradient arriving for the returned buffer is a temporary no one else holds; the
    backward then accumulates the rounds' adjoints into its slots in place (bit-identical, one pass and one tensor less per
    round).  Leave it off when the buffer's gradient may be observed (hooks, ``retain_grad``, an explicit ``backward(G)``)."""
    if rounds < 1:
        raise ValueError("rounds must be >= 1")
    if r0.dim() != 2 or r0.shape[0] != plan.n_cols or plan.n_rows != plan.n_cols:
        raise ValueError(f"r0 must be [{plan.n_cols}, d] and the hop matrices square")
    if reuse and out is None:
        raise ValueError("fused_propagation: reuse=True needs the buffer that holds the propagation (out=)")
    if r0.requires_grad and torch.is_grad_enabled():
        return _FusedPropagation.apply(r0, plan, rounds, out, reuse, private_grad)
    return _FusedPropagation.forward(_NoCtx(), r0, plan, rounds, out, reuse)


class _NoCtx:
    pass


# ---------------------------------------------------------------------------------------------------------------------
# Dropout + output Dense in one pass over the concat buffer (csrc/classifier.hip)
# ---------------------------------------------------------------------------------------------------------------------
def _dd_workspace(n: int, k: int, c: int, device) -> torch.Tensor:
    from . import _capi
    nbytes = int(_capi.lib().h2gcn_dropout_dense_workspace_bytes(n, k, c))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


class _DropoutDenseFn(torch.autograd.Function):
    """``Z = (X .* M / keep) @ W + b`` and its gradients on the library's fp32-MFMA kernels; the mask ``M`` is a counter-based
    function of (seed, step, row, column), recomputed in the backward kernels instead of being stored."""

    @staticmethod
    def forward(ctx, x, kernel, bias, keep_prob, seed, step_dev):
        import ctypes as C

        from . import _capi
        n, k = x.shape
        c = kernel.shape[1]
        w = kernel.contiguous()
        z = torch.empty((n, c), dtype=torch.float32, device=x.device)
        ws = _dd_workspace(n, k, c, x.device)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _capi.check(_capi.lib().h2gcn_dropout_dense_f32(
                C.c_void_p(x.data_ptr()), x.stride(0), n, k, C.c_void_p(w.data_ptr()), c,
                C.c_void_p(bias.data_ptr()) if bias is not None else None, float(keep_prob), int(seed),
                C.c_void_p(step_dev.data_ptr()) if step_dev is not None else None, C.c_void_p(z.data_ptr()), z.stride(0),
                C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(stream)))
        ctx.save_for_backward(x, w, step_dev if step_dev is not None else torch.empty(0, device=x.device))
        ctx.keep_prob, ctx.seed, ctx.has_bias, ctx.has_step = float(keep_prob), int(seed), bias is not None, step_dev is not None
        return z

    @staticmethod
    def backward(ctx, g):
        import ctypes as C

        from . import _capi
        x, w, step = ctx.saved_tensors
        n, k = x.shape
        c = w.shape[1]
        g = g.contiguous()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty((n, k), dtype=torch.float32, device=x.device) if need_dx else None
        dw = torch.empty((k, c), dtype=torch.float32, device=x.device) if need_dw else None
        if need_dx or need_dw:
            ws = _dd_workspace(n, k, c, x.device)
            with torch.cuda.device(x.device):
                stream = torch.cuda.current_stream(x.device).cuda_stream
                _capi.check(_capi.lib().h2gcn_dropout_dense_backward_f32(
                    C.c_void_p(x.data_ptr()), x.stride(0), n, k, C.c_void_p(w.data_ptr()), c, C.c_void_p(g.data_ptr()), g.stride(0),
                    ctx.keep_prob, ctx.seed, C.c_void_p(step.data_ptr()) if ctx.has_step else None,
                    C.c_void_p(dx.data_ptr()) if need_dx else None, dx.stride(0) if need_dx else k,
                    C.c_void_p(dw.data_ptr()) if need_dw else None, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(stream)))
        db = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None, None


class DropoutDense(torch.nn.Module):
    """keras ``Dropout(rate)`` followed by ``Dense(units)`` -- the ``D0.5-MO`` tail of the network setup (reference
    ``h2gcn/models/H2GCN.py:235-257``, called in order at ``:308-325``) -- as ONE pass over the ``[N, K]`` input per
    direction: the dropout mask is drawn from a counter-based generator inside the fp32-MFMA product kernels of
    ``csrc/classifier.hip`` (forward, ``dX``, ``dW``) instead of being materialised by a pass of its own.  Same parameters
    as the unfused pair (``kernel [K, units]``, optional ``bias``); in evaluation the mask is off and the layer is the plain
    product.  Inputs the kernels do not cover (CPU tensors, ``units > 64``, non-fp32) take the stock two-op path.

    The mask stream differs from torch's (and from TensorFlow's, which nothing can reproduce): one mask per training
    forward, keyed by ``(seed, step)``; ``step`` lives in a device counter bumped by a stream-ordered op, so a replayed
    hipGraph draws a fresh mask every epoch."""

    def __init__(self, input_dim: int, units: int, use_bias: bool, drop_prob: float, seed: Optional[int] = None):
        super().__init__()
        self.kernel = torch.nn.Parameter(torch.empty(input_dim, units))
        torch.nn.init.xavier_uniform_(self.kernel)
        self.bias = torch.nn.Parameter(torch.zeros(units)) if use_bias else None
        self.drop_prob = float(drop_prob)
        if seed is None:
            # the mask is a pure function of (seed, step, row, column group) and every layer's step counter advances in
            # lock step: without a per-layer salt two DropoutDense layers of one model would draw IDENTICAL masks.  The
            # salt is drawn from torch's default generator right after the kernel's initialisation, like one more weight:
            # a pure function of torch.manual_seed and the construction order of THIS model -- never of what else the
            # process built before (repeated runs of run_experiments, one test after another).  H2GCN passes
            # seed = initial_seed + layer index explicitly.
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.seed = int(seed) & 0x7FFFFFFFFFFFFFFF
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # row-partitioned runs: the kernels index the mask by LOCAL row, so every rank gets its own stream (otherwise row i
            # of every shard would be dropped identically)
            self.seed = (self.seed ^ (torch.distributed.get_rank() * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF
        self.register_buffer("_step", torch.zeros(1, dtype=torch.int64), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        training = self.training and self.drop_prob > 0.0
        fused_ok = (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.kernel.shape[1] <= 64
                    and (x.shape[1] <= 1 or x.stride(1) == 1) and x.stride(0) >= x.shape[1])
        if not fused_ok:
            if training:
                x = torch.nn.functional.dropout(x, self.drop_prob, True)
            y = x @ self.kernel
            return y if self.bias is None else y + self.bias
        step = None
        if training:
            self._step += 1                      # stream-ordered: a captured graph bumps it on every replay
            step = self._step.clone()            # the value this forward (and its backward) uses
        return _DropoutDenseFn.apply(x, self.kernel, self.bias, 1.0 - self.drop_prob if training else 1.0, self.seed, step)

    def extra_repr(self) -> str:
        return f"in={self.kernel.shape[0]}, units={self.kernel.shape[1]}, bias={self.bias is not None}, drop={self.drop_prob}"
