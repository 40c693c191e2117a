"""The reference's optimizer step on the library's one-launch kernel (``csrc/optimizer.hip``).

Reference: ``keras.optimizers.get(optimizer).from_config({"lr": lr})`` + ``optimizer.apply_gradients``
(``h2gcn/models/H2GCN.py:62-63, 73``), i.e. Keras Adam with its defaults (beta_1 0.9, beta_2 0.999, epsilon 1e-7).  Keras /
TensorFlow add epsilon to the UNCORRECTED ``sqrt(v)``::

    alpha = lr * sqrt(1 - beta_2**t) / (1 - beta_1**t)
    m += (g - m) * (1 - beta_1);  v += (g*g - v) * (1 - beta_2);  param -= (m * alpha) / (sqrt(v) + epsilon)

whereas ``torch.optim.Adam`` adds it to the bias-corrected one -- the same update only when epsilon is negligible.  This class
keeps the reference's form.  GPU fp32 parameters: ONE kernel launch per step for all tensors, the step counter in device memory
(bumped by a stream-ordered op, so the step is capturable into a hipGraph).  Other parameters (CPU tensors in the CPU test
suite): the same formula as torch expressions.
"""
import ctypes as C

import torch

from . import _capi


class KerasAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 0.001, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7):
        if lr < 0 or not 0 <= beta_1 < 1 or not 0 <= beta_2 < 1 or epsilon < 0:
            raise ValueError(f"KerasAdam: lr {lr}, beta_1 {beta_1}, beta_2 {beta_2}, epsilon {epsilon}")
        super().__init__(params, dict(lr=lr, beta_1=beta_1, beta_2=beta_2, epsilon=epsilon))
        self._zero_grads = {}
        self._l2 = {}   # param (tensors hash by identity, like self.state's keys) -> coefficient of a keras-style l2 regulariser folded into the step (see set_l2)

    def set_l2(self, params, coefficient: float) -> None:
        """Fold the gradient of ``coefficient * sum(p ** 2)`` (keras ``regularizers.l2``, reference ``H2GCN.py:239-240,247-248``)
        of every parameter in ``params`` into the step: the update uses ``p.grad + 2 * coefficient * p`` -- the very value
        autograd would have accumulated had the penalty been part of the loss (same roundings), without the pow / sum / mul /
        add launches per kernel.  The caller then keeps the penalty OUT of the autograd graph (``l2_penalty`` gives its value)."""
        if coefficient < 0:
            raise ValueError(f"KerasAdam.set_l2: coefficient {coefficient}")
        for p in params:
            self._l2[p] = float(coefficient)
            if p.is_cuda and _capi.has("h2gcn_l2_penalty_workspace_bytes"):
                # the step closures report the penalty's VALUE through l2_penalty(): its scratch must exist (and be zero) BEFORE
                # any hipGraph capture -- a buffer first allocated inside a capture lives in the graph's private pool, its zero-fill
                # recorded but not yet executed
                _penalty_workspace(p.device)

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def _zero_grad_of(self, p):
        """The all-zero data gradient of a regularised kernel that did not take part in the loss: one buffer per parameter, made
        once (nothing writes to it), not one allocation per step.  Kept outside ``self.state`` (not optimizer state)."""
        z = self._zero_grads.get(p)
        if z is None:
            z = self._zero_grads[p] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return z

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, b1, b2, eps = group["lr"], group["beta_1"], group["beta_2"], group["epsilon"]
            # a kernel registered with set_l2 whose data gradient is absent (it did not take part in this loss) still owes the
            # penalty's 2*l2*w -- what autograd would have produced had the penalty been part of the loss: step it on zeros
            # (a FROZEN kernel -- requires_grad False -- is left alone: frozen means no update, penalty included)
            ps = [p for p in group["params"] if p.grad is not None or (p.requires_grad and self._l2.get(p, 0.0))]
            if not ps:
                continue
            # one step counter per group, on the device of its first parameter (host copy for the CPU formula)
            if "step_dev" not in group:
                group["step_dev"] = torch.zeros(1, dtype=torch.int64, device=ps[0].device)
            group["step_dev"] += 1
            fast = [p for p in ps if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.device == group["step_dev"].device]
            slow = [p for p in ps if not any(p is q for q in fast)]
            if fast:
                grads = [self._zero_grad_of(p) if p.grad is None else
                         (p.grad if p.grad.is_contiguous() and p.grad.dtype == torch.float32 else p.grad.to(torch.float32).contiguous()) for p in fast]
                states = [self._state(p) for p in fast]
                n = len(fast)
                arr = C.c_void_p * n
                l2s = [self._l2.get(p, 0.0) for p in fast]
                with torch.cuda.device(fast[0].device):
                    stream = torch.cuda.current_stream(fast[0].device).cuda_stream
                    common = (n, arr(*[p.data_ptr() for p in fast]), arr(*[g.data_ptr() for g in grads]),
                              arr(*[s["m"].data_ptr() for s in states]), arr(*[s["v"].data_ptr() for s in states]),
                              (C.c_int64 * n)(*[p.numel() for p in fast]))
                    tail = (lr, b1, b2, eps, C.c_void_p(group["step_dev"].data_ptr()), 0, C.c_void_p(stream))
                    if any(l2s):
                        _capi.check(_capi.lib().h2gcn_adam_keras_l2_f32(*common, (C.c_float * n)(*l2s), *tail))
                    else:
                        _capi.check(_capi.lib().h2gcn_adam_keras_f32(*common, *tail))
                # the kernel wrote the parameters through raw pointers: tell autograd's version counters, so that anything
                # keyed by `p._version` (models.H2GCN's propagation reuse) sees the update like after any in-place torch op
                torch._C._increment_version(fast)
            if slow and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("KerasAdam: parameters outside the one-launch kernel (CPU / non-fp32 / non-contiguous) cannot be "
                                   "stepped inside a hipGraph capture")
            t = int(group["step_dev"].item()) if slow else 0   # ONE counter: the device one (a replayed graph advances it too)
            for p in slow:
                st = self._state(p)
                one, tb1, tb2 = (torch.tensor(x, dtype=torch.float32) for x in (1.0, b1, b2))   # fp32 like the kernel
                alpha = (torch.tensor(lr, dtype=torch.float32) * torch.sqrt(one - tb2 ** t) / (one - tb1 ** t)).item()
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                if self._l2.get(p, 0.0):
                    g = g + p * (2.0 * self._l2[p])
                st["m"].add_((g - st["m"]) * (one - tb1).item())
                st["v"].add_((g * g - st["v"]) * (one - tb2).item())
                p.sub_((st["m"] * alpha) / (st["v"].sqrt() + eps))
        return loss


    def load_state_dict(self, state_dict) -> None:
        """Restore INTO the existing tensors: a captured training hipGraph holds the addresses of ``m``, ``v`` and the step
        counter, so a restore (``BestSnapshot.restore``) must not replace them with fresh allocations the graph knows nothing
        about."""
        old_state = {p: dict(st) for p, st in self.state.items()}
        old_steps = [g.get("step_dev") for g in self.param_groups]
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for p, st in self.state.items():
                for name in ("m", "v"):
                    kept = old_state.get(p, {}).get(name)
                    if kept is not None and name in st and kept is not st[name] and kept.shape == st[name].shape:
                        kept.copy_(st[name])
                        st[name] = kept
            for g, kept in zip(self.param_groups, old_steps):
                new = g.get("step_dev")
                if kept is not None and new is not None and kept is not new:
                    kept.copy_(new.to(kept.device))
                    g["step_dev"] = kept
                g.pop("step_host", None)   # (state dicts written before the single-counter change)


_PENALTY_WS = {}


def _penalty_workspace(dev) -> torch.Tensor:
    """Scratch of ``h2gcn_l2_penalty_f32`` (partials + a ticket the kernel re-arms itself): ONE zero-initialised buffer per device
    (a captured step and the eager steps around it must use the same one, and a capture runs on a stream of its own -- so not per
    stream: calls on one device have to be stream-ordered with respect to each other, which the step closures are).  It must come
    into being OUTSIDE a hipGraph capture -- ``KerasAdam.set_l2`` sees to that: created inside one it would live in the graph's
    private pool with its zero-fill only recorded, and an eager call before the first replay would read a garbage ticket and
    never write its result."""
    dev = torch.device(dev)
    ws = _PENALTY_WS.get(dev)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("l2_penalty: first use on this device inside a hipGraph capture -- call it (or KerasAdam.set_l2) "
                               "once eagerly first: its workspace cannot be created inside a capture")
        with torch.cuda.device(dev):
            ws = torch.zeros(int(_capi.lib().h2gcn_l2_penalty_workspace_bytes()) // 8 + 1, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)       # the fill has executed before anything, on any stream, can use the buffer
        _PENALTY_WS[dev] = ws
    return ws


def l2_penalty(params, coefficients) -> torch.Tensor:
    """``sum_k coefficients[k] * sum(params[k] ** 2)`` as a 0-dim tensor WITHOUT a gradient: the value of the keras l2 penalty a
    step reports next to its cross-entropy (reference ``H2GCN.py:363-367``), one kernel launch for all tensors
    (``h2gcn_l2_penalty_f32``: fp64 inside a tensor, fp32 across tensors).  GPU fp32 contiguous tensors only.  The kernel's scratch
    (partials + a ticket it re-arms itself) is one persistent buffer per device (``_penalty_workspace``: created eagerly by
    ``KerasAdam.set_l2``, never inside a capture), so a replayed step carries no extra fill -- which means calls on ONE device must
    be stream-ordered with respect to each other (they are: the step closures issue everything on the current stream)."""
    params, coefficients = list(params), [float(c) for c in coefficients]
    n = len(params)
    if n == 0:
        return torch.zeros(())
    dev = params[0].device
    if not all(p.is_cuda and p.device == dev and p.dtype == torch.float32 and p.is_contiguous() for p in params) or n > 16:
        raise ValueError("l2_penalty: contiguous fp32 tensors on one GPU (at most 16) expected")
    L = _capi.lib()
    ws = _penalty_workspace(dev)
    out = torch.empty((), dtype=torch.float32, device=dev)
    arr = C.c_void_p * n
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(L.h2gcn_l2_penalty_f32(n, arr(*[p.data_ptr() for p in params]), (C.c_int64 * n)(*[p.numel() for p in params]),
                                           (C.c_float * n)(*coefficients), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                           ws.numel() * 8, C.c_void_p(stream)))
    return out


class KerasRMSprop(torch.optim.Optimizer):
    """``--optimizer rmsprop``: Keras RMSprop with its defaults (rho 0.9, momentum 0, epsilon 1e-7, not centered) as
    TensorFlow's ApplyRMSProp computes it -- epsilon INSIDE the square root::

        ms += (g*g - ms) * (1 - rho);   param -= lr * g / sqrt(ms + epsilon)

    (``torch.optim.RMSprop`` divides by ``sqrt(ms) + eps``: with eps = 1e-7 that is a 3e-4 vs 1e-7 floor on the denominator.)
    Not a hot path (the reference's default is adam): plain torch expressions on any device."""

    def __init__(self, params, lr: float = 0.001, rho: float = 0.9, epsilon: float = 1e-7):
        if lr < 0 or not 0 <= rho < 1 or epsilon < 0:
            raise ValueError(f"KerasRMSprop: lr {lr}, rho {rho}, epsilon {epsilon}")
        super().__init__(params, dict(lr=lr, rho=rho, epsilon=epsilon))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["ms"] = torch.zeros_like(p)
                g = p.grad
                st["ms"].add_((g * g - st["ms"]) * (1 - group["rho"]))
                p.sub_(group["lr"] * g / torch.sqrt(st["ms"] + group["epsilon"]))
        return loss
