"""GPU suite: the fused dropout + output-Dense kernels (csrc/classifier.hip, fp32 MFMA) against the CPU restatement
(oracle/classifier.py: the same counter-based mask bit for bit, products in fp64).

Tolerance: the kernels are exact-fp32 multiply-add chains over K (forward, <= 1792 terms), C (dX, <= 64 terms) or all rows
(dW); against the fp64 restatement the error is fp32 round-off of that chain: 2e-6 * sum |terms| is asserted (measured:
~1e-7 * sum)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import classifier as oc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["matrix-core kernels", "small-operand kernels where they apply"])
def _kernel_family(request):
    """Every test runs twice: with the latency-optimised small-operand kernels switched off (h2gcn_dropout_dense_small_rows(0):
    every call on the fp32-MFMA kernels, as in round 3) and with the shipped rule (<= 12288 rows and <= 16 classes: the plain
    kernels).  Same contract, same mask, same tolerance."""
    from h2gcn_amd import _capi
    L = _capi.lib()
    old = L.h2gcn_dropout_dense_small_rows(0 if request.param.startswith("matrix") else 12288)
    yield request.param
    L.h2gcn_dropout_dense_small_rows(old)


def _call_fwd(x, w, b, keep, seed, step):
    from h2gcn_amd import _capi
    L = _capi.lib()
    n, k = x.shape
    c = w.shape[1]
    z = torch.full((n, c + 3), 9.0, device=DEV)
    ws = torch.empty(int(L.h2gcn_dropout_dense_workspace_bytes(n, k, c)), dtype=torch.uint8, device=DEV)
    st = torch.tensor([step], dtype=torch.int64, device=DEV)
    _capi.check(L.h2gcn_dropout_dense_f32(C.c_void_p(x.data_ptr()), x.stride(0), n, k, C.c_void_p(w.data_ptr()), c,
                                          C.c_void_p(b.data_ptr()) if b is not None else None, keep, seed, C.c_void_p(st.data_ptr()),
                                          C.c_void_p(z.data_ptr()), z.stride(0), C.c_void_p(ws.data_ptr()), ws.numel(), None))
    torch.cuda.synchronize()
    assert bool((z[:, c:] == 9.0).all())            # guard columns of a strided output untouched
    return z[:, :c].cpu().numpy()


@pytest.mark.parametrize("n,k,c", [(1, 4, 1), (37, 7, 3), (129, 448, 47), (1000, 700, 10), (513, 896, 64), (300, 130, 17), (4099, 448, 7)])
@pytest.mark.parametrize("keep", [0.5, 1.0, 0.9])
def test_forward_and_backward_match_the_restatement(n, k, c, keep):
    from h2gcn_amd import _capi
    rng = np.random.default_rng(n * 7 + k + c)
    xbuf = torch.zeros((n, k + 5), device=DEV)                  # strided, only 4-byte aligned rows
    x = xbuf[:, 1:1 + k]
    x.copy_(torch.from_numpy(rng.uniform(-1, 1, (n, k)).astype(np.float32)))
    w = torch.from_numpy(rng.uniform(-0.3, 0.3, (k, c)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.uniform(-0.5, 0.5, c).astype(np.float32)).to(DEV)
    g = torch.from_numpy(rng.uniform(-1, 1, (n, c)).astype(np.float32)).to(DEV)
    seed, step = 0x1234_5678_9ABC, 41
    xn, wn, bn, gn = x.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), g.cpu().numpy()
    z = _call_fwd(x, w, b, keep, seed, step)
    want = oc.dropout_dense(xn, wn, bn, keep, seed, step)
    mag = np.abs(np.where(oc.keep_mask(n, k, keep, seed, step), xn / keep, 0)) @ np.abs(wn) + np.abs(bn)
    assert (np.abs(z - want) <= 2e-6 * np.maximum(mag, 1.0)).all(), np.abs(z - want).max()
    if keep < 1.0 and n * k >= 1000:   # a different step draws a different mask
        assert not np.array_equal(z, _call_fwd(x, w, b, keep, seed, step + 1))
    # backward
    L = _capi.lib()
    ws = torch.empty(int(L.h2gcn_dropout_dense_workspace_bytes(n, k, c)), dtype=torch.uint8, device=DEV)
    st = torch.tensor([step], dtype=torch.int64, device=DEV)
    dxbuf = torch.full((n, k + 2), 5.0, device=DEV)
    dw = torch.empty((k, c), device=DEV)
    _capi.check(L.h2gcn_dropout_dense_backward_f32(C.c_void_p(x.data_ptr()), x.stride(0), n, k, C.c_void_p(w.data_ptr()), c,
                                                   C.c_void_p(g.data_ptr()), g.stride(0), keep, seed, C.c_void_p(st.data_ptr()),
                                                   C.c_void_p(dxbuf.data_ptr()), dxbuf.stride(0), C.c_void_p(dw.data_ptr()),
                                                   C.c_void_p(ws.data_ptr()), ws.numel(), None))
    torch.cuda.synchronize()
    dx_w, dw_w, _ = oc.dropout_dense_grad(xn, wn, gn, keep, seed, step)
    assert bool((dxbuf[:, k:] == 5.0).all())
    dx = dxbuf[:, :k].cpu().numpy()
    m = oc.keep_mask(n, k, keep, seed, step)
    assert np.array_equal(dx == 0, ~m | (dx_w == 0))                      # exactly the dropped elements are zero
    assert (np.abs(dx - dx_w) <= 2e-6 * np.maximum((np.abs(gn) @ np.abs(wn).T) / keep, 1.0)).all()
    mag_w = np.abs(np.where(m, xn / keep, 0)).T @ np.abs(gn)
    assert (np.abs(dw.cpu().numpy() - dw_w) <= 2e-6 * np.maximum(mag_w, 1.0)).all(), np.abs(dw.cpu().numpy() - dw_w).max()


def test_module_matches_unfused_pair_and_autograd():
    """DropoutDense: evaluation == plain product; rate 0 == Dense; training keeps ~keep_prob of the inputs, gradients flow
    to input, kernel and bias and agree with autograd through the same mask; every training forward draws a new mask;
    results are bitwise repeatable for a fixed (seed, step)."""
    from h2gcn_amd.layers import DropoutDense

    torch.manual_seed(3)
    n, k, c = 2000, 448, 47
    x = torch.randn((n, k), device=DEV)
    layer = DropoutDense(k, c, use_bias=True, drop_prob=0.5).to(DEV)
    with torch.no_grad():
        layer.bias.uniform_(-0.2, 0.2)
    ref = x @ layer.kernel + layer.bias
    assert (layer.eval()(x) - ref).abs().max().item() <= 2e-5
    layer.train()
    xa = x.clone().requires_grad_(True)
    z1 = layer(xa)
    step_used = int(layer._step.item())
    m = torch.from_numpy(oc.keep_mask(n, k, 0.5, layer.seed, step_used)).to(DEV)
    assert 0.49 < m.float().mean().item() < 0.51
    xb = x.clone().requires_grad_(True)
    k2 = layer.kernel.detach().clone().requires_grad_(True)
    b2 = layer.bias.detach().clone().requires_grad_(True)
    z_ref = (xb * m / 0.5) @ k2 + b2
    assert (z1 - z_ref).abs().max().item() <= 1e-4
    wgt = torch.randn_like(z1)
    (z1 * wgt).sum().backward()
    (z_ref * wgt).sum().backward()
    assert (xa.grad - xb.grad).abs().max().item() <= 1e-4
    assert (layer.kernel.grad - k2.grad).abs().max().item() <= 2e-3 and (layer.kernel.grad - k2.grad).abs().max().item() / k2.grad.abs().max().item() <= 1e-5
    assert (layer.bias.grad - b2.grad).abs().max().item() <= 1e-3
    z2 = layer(x)
    assert not torch.equal(z1.detach(), z2.detach())                       # next step, next mask
    plain = DropoutDense(k, c, use_bias=True, drop_prob=0.0).to(DEV).train()
    plain.load_state_dict(layer.state_dict())
    assert (plain(x) - ref).abs().max().item() <= 2e-5
    # wide output layers and CPU tensors take the stock path
    wide = DropoutDense(16, 100, use_bias=False, drop_prob=0.5).to(DEV).eval()
    assert wide(x[:, :16]).shape == (n, 100)


def test_model_uses_the_fused_classifier_and_trains():
    """`M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO`: the `D0.5` + output dense pair becomes ONE DropoutDense layer (the dropout slot an
    Identity, parameters under the same names); evaluation equals the stock pair with the same weights; a short full-batch
    training run on Cora with the fused layer decreases the loss like the stock pair does (the mask streams differ, so the
    trajectories agree statistically, not bitwise); `fused_classifier=False` keeps the stock layers."""
    from conftest import load_planetoid_golden
    from h2gcn_amd import HopPlan, operands
    from h2gcn_amd.layers import DropoutDense
    from h2gcn_amd.models import parse_network_setup
    from h2gcn_amd.models.H2GCN import H2GCN, make_optimizer

    g = load_planetoid_golden("cora")
    adj = operands.remove_self_loops(g["adj_raw"])
    rp, ci, va, n = operands.build_adj_norm_hops_device(adj, ("1", "2"), "sym", DEV)
    plan = HopPlan(rp, ci, va, n, build_transpose=True)
    feats = HopPlan.from_scipy([g["feat_rownorm"]], DEV, build_transpose=True, keep_permutation=True)
    mask = torch.from_numpy(np.asarray(g["train_mask"], dtype=bool)).to(DEV)
    y = torch.from_numpy(np.asarray(g["y_all"], dtype=np.float32)).to(DEV) * mask[:, None]
    setup = parse_network_setup("M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO", y.shape[1])
    losses = {}
    models = {}
    for fused in (True, False):
        torch.manual_seed(5)
        model = H2GCN(setup, input_dim=feats.n_cols, n_hops=2, sparse_input=True, l2_regularize_weight=5e-4, fused_classifier=fused).to(DEV)
        kinds = [type(m).__name__ for m in model.layer_objs]
        assert ("DropoutDense" in kinds) == fused and ("Dropout" in kinds) != fused and ("Dense" in kinds) != fused
        models[fused] = model
        opt = make_optimizer("adam", model.parameters(), 0.01)
        hist = []
        for _ in range(30):
            model.train()
            opt.zero_grad(set_to_none=True)
            loss = model.loss(model(None, feats, plan), y, mask)
            loss.backward()
            model.restore_sparse_inputs()
            opt.step()
            hist.append(float(loss))
        losses[fused] = hist
    assert losses[True][-1] < 0.6 * losses[True][0] and losses[False][-1] < 0.6 * losses[False][0]
    assert abs(losses[True][-1] - losses[False][-1]) <= 0.25 * losses[False][-1]          # same regime, different masks
    # same weights -> same evaluation outputs
    models[True].load_state_dict(models[False].state_dict())
    with torch.no_grad():
        a = models[True].eval()(None, feats, plan)
        b = models[False].eval()(None, feats, plan)
    assert (a - b).abs().max().item() <= 1e-5
    assert isinstance(next(m for m in models[True].layer_objs if isinstance(m, DropoutDense)), DropoutDense)
