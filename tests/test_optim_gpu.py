"""GPU suite: the one-launch Adam kernel with the reference's (Keras / TensorFlow) arithmetic (csrc/optimizer.hip) against its
fp32 restatement (oracle/keras_adam.py).  Tolerance 2e-6 relative: the same fp32 operations in the same order; powf / sqrtf of
the device library may differ from numpy's in the last bit."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import keras_adam as ok

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(shapes, seed):
    rng = np.random.default_rng(seed)
    return [rng.normal(size=s).astype(np.float32) for s in shapes]


def test_steps_match_the_restatement_over_many_tensors():
    from h2gcn_amd.optim import KerasAdam
    shapes = [(1433, 64), (64,), (448, 7), (7,), (1,), (3, 5, 2)] + [(11,)] * 14       # 20 tensors: two launch groups
    host = _params(shapes, 1)
    ps = [torch.nn.Parameter(torch.from_numpy(h.copy()).to(DEV)) for h in host]
    opt = KerasAdam(ps, lr=0.01)
    m = [np.zeros_like(h) for h in host]
    v = [np.zeros_like(h) for h in host]
    rng = np.random.default_rng(2)
    for t in range(1, 13):
        gs = [rng.normal(size=h.shape).astype(np.float32) * np.float32(10.0 ** rng.integers(-7, 1)) for h in host]
        for p, g in zip(ps, gs):
            p.grad = torch.from_numpy(g).to(DEV)
        opt.step()
        for k in range(len(host)):
            host[k], m[k], v[k] = ok.keras_adam_step(host[k], gs[k], m[k], v[k], t, lr=0.01)
    for p, h in zip(ps, host):
        got = p.detach().cpu().numpy()
        assert np.abs(got - h).max() <= 2e-6 * max(1.0, np.abs(h).max()), np.abs(got - h).max()


def test_tiny_gradients_follow_keras_not_torch():
    """|g| = 1e-6: Keras moves the weight by 0.24 lr at step 1 (epsilon on the uncorrected sqrt(v)), torch.optim.Adam by 0.91 lr."""
    from h2gcn_amd.optim import KerasAdam
    w = torch.nn.Parameter(torch.zeros(4, device=DEV))
    w.grad = torch.full((4,), 1e-6, device=DEV)
    KerasAdam([w], lr=0.01).step()
    assert abs(w[0].item() / -0.01 - 0.2403) < 2e-3
    w2 = torch.nn.Parameter(torch.zeros(4, device=DEV))
    w2.grad = torch.full((4,), 1e-6, device=DEV)
    torch.optim.Adam([w2], lr=0.01, eps=1e-7).step()
    assert abs(w2[0].item() / -0.01 - 0.909) < 5e-3


def test_replayed_graph_advances_the_step_counter():
    from h2gcn_amd.optim import KerasAdam
    host = _params([(100, 8), (8,)], 3)
    g_host = _params([(100, 8), (8,)], 4)

    def run(graph):
        ps = [torch.nn.Parameter(torch.from_numpy(h.copy()).to(DEV)) for h in host]
        for p, g in zip(ps, g_host):
            p.grad = torch.from_numpy(g).to(DEV)
        opt = KerasAdam(ps, lr=0.05)
        if not graph:
            for _ in range(5):
                opt.step()
        else:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                opt.step()                                   # warm-up step 1 (allocates the state)
            torch.cuda.current_stream().wait_stream(side)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                opt.step()                                   # captured, not executed: step 2 happens on the first replay ...
            # ... except that the host-side bump inside the capture also ran its `+= 1` node once per replay only
            for _ in range(4):
                cg.replay()
        torch.cuda.synchronize()
        return [p.detach().cpu().numpy() for p in ps], int(opt.param_groups[0]["step_dev"].item())

    eager, t_e = run(False)
    replay, t_r = run(True)
    assert t_e == 5 and t_r == 5
    for a, b in zip(eager, replay):
        assert np.array_equal(a, b)


def test_capi_rejects_bad_arguments():
    from h2gcn_amd import _capi
    L = _capi.lib()
    x = torch.zeros(8, device=DEV)
    arr = (C.c_void_p * 1)(x.data_ptr())
    n = (C.c_int64 * 1)(8)
    call = lambda lr, b1, step: L.h2gcn_adam_keras_f32(1, arr, arr, arr, arr, n, lr, b1, 0.999, 1e-7, None, step, None)
    assert call(0.01, 1.0, 1) == _capi.ERR_INVALID_ARGUMENT and b"beta_1" in L.h2gcn_last_error()
    assert call(0.01, 0.9, 0) == _capi.ERR_INVALID_ARGUMENT and b"1-based" in L.h2gcn_last_error()
    assert call(-1.0, 0.9, 1) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_adam_keras_f32(0, None, None, None, None, None, 0.01, 0.9, 0.999, 1e-7, None, 1, None) == 0


def test_host_step_argument_is_the_device_counter_by_another_route():
    """step_dev = NULL: the 1-based step comes from the host argument -- same update as with the device counter."""
    from h2gcn_amd import _capi
    L = _capi.lib()
    rng = np.random.default_rng(5)
    p0, g = rng.normal(size=300).astype(np.float32), rng.normal(size=300).astype(np.float32)
    outs = []
    for use_dev in (True, False):
        p, m, v = (torch.from_numpy(a.copy()).to(DEV) for a in (p0, np.zeros(300, np.float32), np.zeros(300, np.float32)))
        gt = torch.from_numpy(g).to(DEV)
        step = torch.zeros(1, dtype=torch.int64, device=DEV)
        arr = lambda t: (C.c_void_p * 1)(t.data_ptr())
        for t in range(1, 4):
            step += 1
            _capi.check(L.h2gcn_adam_keras_f32(1, arr(p), arr(gt), arr(m), arr(v), (C.c_int64 * 1)(300), 0.01, 0.9, 0.999, 1e-7,
                                               C.c_void_p(step.data_ptr()) if use_dev else None, t, None))
        torch.cuda.synchronize()
        outs.append(p.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    want, m, v = p0.copy(), np.zeros(300, np.float32), np.zeros(300, np.float32)
    for t in range(1, 4):
        want, m, v = ok.keras_adam_step(want, g, m, v, t, lr=0.01)
    assert np.abs(outs[0] - want).max() <= 2e-6 * np.abs(want).max()


def test_l2_folded_into_the_step_equals_autograd_accumulation_bit_for_bit():
    """KerasAdam.set_l2: the update uses p.grad + 2 * l2 * p -- with the roundings autograd makes when the keras penalty
    l2 * sum(p ** 2) is part of the loss (a multiply, then an add into .grad) -- so keeping the penalty out of the graph changes
    no bit of the trajectory.  l2_penalty: the penalty's value in one launch."""
    from h2gcn_amd.optim import KerasAdam, l2_penalty
    shapes = [(1433, 64), (448, 7), (7,)]
    host = _params(shapes, 5)
    l2 = 5e-4
    a = [torch.nn.Parameter(torch.from_numpy(h.copy()).to(DEV)) for h in host]     # penalty through autograd
    b = [torch.nn.Parameter(torch.from_numpy(h.copy()).to(DEV)) for h in host]     # penalty folded into the step
    opt_a, opt_b = KerasAdam(a, lr=0.01), KerasAdam(b, lr=0.01)
    opt_b.set_l2(b[:2], l2)                                                        # the bias is not regularised
    rng = np.random.default_rng(6)
    for t in range(8):
        targets = [torch.from_numpy(rng.normal(size=h.shape).astype(np.float32)).to(DEV) for h in host]
        for ps, opt, fold in ((a, opt_a, False), (b, opt_b, True)):
            opt.zero_grad(set_to_none=True)
            data = sum(((p - tg) ** 2).mean() for p, tg in zip(ps, targets))
            reg = sum(l2 * (p ** 2).sum() for p in ps[:2])
            (data if fold else data + reg).backward()
            opt.step()
        for pa, pb in zip(a, b):
            assert torch.equal(pa, pb), t
    want = sum(l2 * float((p.detach().double() ** 2).sum()) for p in b[:2])
    got = float(l2_penalty([p.detach() for p in b[:2]], [l2, l2]))
    assert abs(got - want) <= 2e-6 * want
    assert float(l2_penalty([p.detach() for p in b[:2]], [l2, l2])) == got          # the workspace re-arms itself
    # CPU parameters take the formula path with the same fold
    c = [torch.nn.Parameter(torch.from_numpy(host[0].copy()))]
    oc = KerasAdam(c, lr=0.01)
    oc.set_l2(c, l2)
    c[0].grad = torch.zeros_like(c[0])
    oc.step()
    assert not torch.equal(c[0].detach(), torch.from_numpy(host[0]))                 # a zero data gradient still moves a regularised weight
