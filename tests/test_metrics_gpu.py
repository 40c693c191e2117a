"""GPU suite: the one-pass masked cross-entropy / accuracy kernels (csrc/metrics.hip) against the CPU restatement of the
reference's metrics (oracle/h2gcn_model.py: `_metrics.py:8-25` in fp64) and against torch autograd of the plain expression.

Tolerance: per-row terms are fp32 (expf / logf of the runtime library), the sums over rows fp64: 2e-6 relative on the losses
(measured ~1e-7); accuracies are sums of the weights of the agreeing rows -- equal to the restatement up to fp32 rounding of
the weights (1e-6) unless two logits of a row tie exactly (the tests draw continuous logits; a tie case is pinned by hand)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import h2gcn_model as om

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(n, c, seed, frac=0.3, empty_label_rows=True):
    rng = np.random.default_rng(seed)
    z = rng.normal(0, 2.0, (n, c)).astype(np.float32)
    cls = rng.integers(0, c, n)
    y = np.zeros((n, c), np.float32)
    y[np.arange(n), cls] = 1.0
    if empty_label_rows and n > 4:
        y[rng.choice(n, max(1, n // 10), replace=False)] = 0.0       # unlabeled nodes: all-zero rows (planetoid test split filler)
    mask = rng.random(n) < frac
    if not mask.any():
        mask[0] = True
    return z, y, mask


@pytest.mark.parametrize("n,c", [(1, 1), (5, 3), (17, 7), (1000, 47), (4099, 64), (100003, 47), (333, 10), (64, 2)])
def test_losses_and_accuracies_match_the_restatement(n, c):
    from h2gcn_amd import metrics
    z, y0, m0 = _case(n, c, 1)
    _, y1, m1 = _case(n, c, 2, frac=0.6)
    _, y2, m2 = _case(n, c, 3, frac=0.05, empty_label_rows=False)
    zt = torch.from_numpy(z).to(DEV)
    ys = [torch.from_numpy(y).to(DEV) for y in (y0, y1, y2)]
    ws = [torch.from_numpy((m / m.sum()).astype(np.float32)).to(DEV) for m in (m0, m1, m2)]
    loss, acc = metrics.masked_metrics(zt, ys, ws)
    loss, acc = loss.cpu().numpy(), acc.cpu().numpy()
    for k, (y, m) in enumerate(((y0, m0), (y1, m1), (y2, m2))):
        ref_l = om.masked_softmax_cross_entropy(z.astype(np.float64), y.astype(np.float64), m)
        ref_a = om.masked_accuracy(z.astype(np.float64), y.astype(np.float64), m)
        assert abs(loss[k] - ref_l) <= 2e-6 * max(1.0, abs(ref_l)), (k, loss[k], ref_l)
        assert abs(acc[k] - ref_a) <= 1e-6 * max(1.0, n ** 0.5), (k, acc[k], ref_a)
    # deterministic: the same call twice is bit-identical
    l2, a2 = metrics.masked_metrics(zt, ys, ws)
    assert torch.equal(l2.cpu(), torch.from_numpy(loss)) and torch.equal(a2.cpu(), torch.from_numpy(acc))


def test_hand_example_and_ties():
    """5 nodes, 3 classes, worked by hand: uniform logits give CE = ln 3; a confident correct row contributes ~0; ties resolve
    to the lowest class on both sides (numpy / torch argmax)."""
    from h2gcn_amd import metrics
    z = np.array([[0, 0, 0], [10, 0, 0], [0, 10, 0], [1, 1, 0], [2, 1, 3]], np.float32)
    y = np.array([[0, 1, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0], [0, 0, 0]], np.float32)
    mask = np.array([1, 1, 1, 1, 0], bool)
    w = (mask / mask.sum()).astype(np.float32)
    loss, acc = metrics.masked_metrics(torch.from_numpy(z).to(DEV), [torch.from_numpy(y).to(DEV)], [torch.from_numpy(w).to(DEV)])
    lse = lambda r: np.log(np.exp(np.asarray(r, np.float64)).sum())
    want = 0.25 * ((lse(z[0]) - 0) + (lse(z[1]) - 10) + (lse(z[2]) - 0) + (lse(z[3]) - 1))
    assert abs(loss.item() - want) < 1e-6 and abs(lse(z[0]) - np.log(3)) < 1e-12
    # rows 0 (tie -> class 0 vs label 1: miss), 1 (hit), 2 (miss), 3 (tie 1,1 -> class 0: hit)
    assert abs(acc.item() - 0.5) < 1e-7
    assert abs(loss.item() - om.masked_softmax_cross_entropy(z.astype(np.float64), y.astype(np.float64), mask)) < 1e-6
    # an all-zero weight vector: both quantities are exactly zero and nothing is read
    l0, a0 = metrics.masked_metrics(torch.from_numpy(z).to(DEV), [torch.from_numpy(y).to(DEV)], [torch.zeros(5, device=DEV)])
    assert l0.item() == 0.0 and a0.item() == 0.0


@pytest.mark.parametrize("n,c", [(5, 3), (1000, 47), (4099, 64), (257, 7)])
def test_gradient_matches_autograd_of_the_plain_expression(n, c):
    from h2gcn_amd import metrics
    z, y, m = _case(n, c, 11)
    w = torch.from_numpy((m / m.sum()).astype(np.float32)).to(DEV)
    yt = torch.from_numpy(y).to(DEV)
    big = torch.zeros((n, c + 5), device=DEV)                        # strided logits: a column slice of a wider buffer
    big[:, :c] = torch.from_numpy(z).to(DEV)
    za = big[:, :c].detach().requires_grad_(True)
    zb = torch.from_numpy(z).to(DEV).double().requires_grad_(True)
    la = metrics.masked_cross_entropy(za, yt, w)
    lb = (-(yt.double() * torch.log_softmax(zb, dim=1)).sum(1) * w.double()).sum()
    (la * 3.5).backward()                                            # a non-unit upstream gradient
    (lb * 3.5).backward()
    assert abs(la.item() - lb.item()) <= 2e-6 * max(1.0, abs(lb.item()))
    err = (za.grad.double() - zb.grad).abs().max().item()
    assert err <= 2e-6 * max(zb.grad.abs().max().item(), 1e-3), err
    assert bool((za.grad[~torch.from_numpy(m).to(DEV)] == 0).all())  # unmasked rows: exact zeros


def test_model_metrics_route_through_the_kernels_and_agree_with_torch():
    from h2gcn_amd.models import _metrics
    z, y, m = _case(3000, 47, 21)
    zt, yt, mt = torch.from_numpy(z).to(DEV), torch.from_numpy(y).to(DEV), torch.from_numpy(m).to(DEV)
    wt = mt.float() / mt.float().sum()
    plain_l = (-(yt * torch.log_softmax(zt, dim=1)).sum(1) * wt).sum().item()
    plain_a = ((zt.argmax(1) == yt.argmax(1)).float() * wt).sum().item()
    assert abs(_metrics.masked_softmax_cross_entropy(zt, yt, mt).item() - plain_l) < 2e-6 * max(1.0, plain_l)
    assert abs(_metrics.masked_accuracy(zt, yt, mt).item() - plain_a) < 1e-5
    # wide heads (> 64 classes) and CPU tensors keep the torch expressions
    zw = torch.randn(50, 100, device=DEV)
    yw = torch.nn.functional.one_hot(torch.randint(0, 100, (50,), device=DEV), 100).float()
    assert torch.isfinite(_metrics.masked_softmax_cross_entropy(zw, yw, torch.ones(50, device=DEV)))
    assert torch.isfinite(_metrics.masked_softmax_cross_entropy(zt.cpu(), yt.cpu(), mt.cpu()))


def test_capi_rejects_bad_arguments():
    from h2gcn_amd import _capi
    L = _capi.lib()
    z = torch.zeros((4, 70), device=DEV)
    w = torch.ones(4, device=DEV)
    out = torch.zeros(2, device=DEV)
    ws = torch.empty(4096, dtype=torch.uint8, device=DEV)
    yp, ld, wp = (C.c_void_p * 1)(z.data_ptr()), (C.c_int64 * 1)(70), (C.c_void_p * 1)(w.data_ptr())
    call = lambda c, n_sets, wsb: L.h2gcn_masked_metrics_f32(C.c_void_p(z.data_ptr()), 70, 4, c, n_sets, yp, ld, wp, C.c_void_p(out.data_ptr()),
                                                             None, C.c_void_p(ws.data_ptr()), wsb, None)
    assert call(65, 1, 4096) == _capi.ERR_INVALID_ARGUMENT and b"C 65" in L.h2gcn_last_error()
    assert call(10, 5, 4096) == _capi.ERR_INVALID_ARGUMENT
    assert call(10, 1, 8) == _capi.ERR_INVALID_ARGUMENT and b"workspace" in L.h2gcn_last_error()
    assert call(10, 1, 4096) == 0
    assert L.h2gcn_masked_ce_backward_f32(C.c_void_p(z.data_ptr()), 5, 4, 10, C.c_void_p(z.data_ptr()), 70, C.c_void_p(w.data_ptr()), None,
                                          C.c_void_p(z.data_ptr()), 70, None) == _capi.ERR_INVALID_ARGUMENT


def test_property_sweep_over_shapes_strides_and_weight_patterns():
    """hypothesis: class counts 1..64, row counts 1..3000, strided logits / labels, 1..4 sets, sparse / dense / all-zero weight
    vectors, soft (non one-hot) label rows, logits drawn from a coarse grid (frequent exact ties) -- losses, accuracies and the
    gradient against the fp64 restatement."""
    import os
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    from h2gcn_amd import metrics

    @settings(max_examples=int(os.environ.get("H2GCN_FUZZ_EXAMPLES", "150")) // 2, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(1, 3000), c=st.integers(1, 64), n_sets=st.integers(1, 4),
           pad_z=st.integers(0, 5), pad_y=st.integers(0, 3), grid=st.booleans(), soft=st.booleans())
    def run(seed, n, c, n_sets, pad_z, pad_y, grid, soft):
        rng = np.random.default_rng(seed)
        z = rng.integers(-3, 4, (n, c)).astype(np.float32) if grid else rng.normal(0, 3, (n, c)).astype(np.float32)
        zt = torch.zeros((n, c + pad_z), device=DEV)
        zt[:, :c] = torch.from_numpy(z).to(DEV)
        ys, ws, ys_h, ws_h = [], [], [], []
        for m in range(n_sets):
            if soft:
                y = rng.random((n, c)).astype(np.float32) * (rng.random((n, 1)) < 0.8)
            else:
                y = np.zeros((n, c), np.float32)
                y[np.arange(n), rng.integers(0, c, n)] = 1.0
                y[rng.random(n) < 0.1] = 0.0
            frac = [0.0, 0.02, 0.5, 1.0][int(rng.integers(0, 4))]
            mask = rng.random(n) < frac
            w = (mask / max(1, mask.sum())).astype(np.float32)
            yt = torch.zeros((n, c + pad_y), device=DEV)
            yt[:, :c] = torch.from_numpy(y).to(DEV)
            ys.append(yt[:, :c]); ws.append(torch.from_numpy(w).to(DEV)); ys_h.append(y); ws_h.append(w)
        loss, acc = metrics.masked_metrics(zt[:, :c], ys, ws)
        loss, acc = loss.cpu().numpy(), acc.cpu().numpy()
        z64 = z.astype(np.float64)
        zs = z64 - z64.max(1, keepdims=True)
        logp = zs - np.log(np.exp(zs).sum(1, keepdims=True))
        for m in range(n_sets):
            y64, w64 = ys_h[m].astype(np.float64), ws_h[m].astype(np.float64)
            ref_l = float((-(y64 * logp).sum(1) * w64).sum())
            ref_a = float(((z64.argmax(1) == y64.argmax(1)) * w64).sum())
            assert abs(loss[m] - ref_l) <= 3e-6 * max(1.0, abs(ref_l)) * max(1.0, float(y64.sum(1).max())), (m, loss[m], ref_l)
            assert abs(acc[m] - ref_a) <= 2e-6, (m, acc[m], ref_a)
        # gradient of set 0
        za = zt[:, :c].detach().requires_grad_(True)
        metrics.masked_cross_entropy(za, ys[0], ws[0]).backward()
        p = np.exp(logp)
        want = ws_h[0].astype(np.float64)[:, None] * (p * ys_h[0].astype(np.float64).sum(1, keepdims=True) - ys_h[0])
        got = za.grad.cpu().numpy()
        # (the entry of the labelled class is w * (p - 1): where p is within a few ulp of 1 -- a confident row, small c -- fp32 can
        # only hold it to ulp(1) * w, whatever the size of the result; hence the absolute term)
        assert np.abs(got - want).max() <= 3e-6 * max(1e-3, np.abs(want).max()) + 2.5e-7 * float(ws_h[0].max()) + 1e-12

    run()
