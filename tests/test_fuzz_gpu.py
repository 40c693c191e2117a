"""GPU suite: property-based sweep (hypothesis) over operand shapes and every schedule knob -- forward and adjoint
against the fp64 oracle, with the tolerance of tests/test_spmm_gpu.py."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import gcn_layer as og

pytestmark = pytest.mark.gpu


def _csr(rng, n_rows, n_cols, mean_deg, heavy, empty_frac):
    deg = rng.poisson(mean_deg, n_rows)
    if heavy:
        deg[rng.integers(0, n_rows)] = min(n_cols, int(rng.integers(200, 900)))
    deg[rng.random(n_rows) < empty_frac] = 0
    deg = np.minimum(deg, n_cols)
    rows = np.repeat(np.arange(n_rows), deg)
    cols = np.concatenate([rng.choice(n_cols, k, replace=False) for k in deg]) if deg.sum() else np.zeros(0, dtype=np.int64)
    vals = rng.uniform(-1, 1, len(rows)).astype(np.float32)
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n_rows, n_cols))
    m.sort_indices()
    return m


@settings(max_examples=int(os.environ.get("H2GCN_FUZZ_EXAMPLES", "150")), deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 2 ** 31 - 1), n_rows=st.integers(1, 700), n_cols=st.integers(1, 700),
       d=st.sampled_from([1, 2, 5, 8, 16, 32, 48, 64, 67, 96, 128, 130, 160, 256, 320]), n_hops=st.integers(1, 8),
       mean_deg=st.sampled_from([0.3, 3.0, 20.0, 70.0]), heavy=st.booleans(),
       threshold=st.sampled_from([0, 4, 64, 300]), rpw=st.sampled_from([0, 1, 3, 7]),
       variant=st.sampled_from([0, 1, 2, 3, 5, 6]), slice_cols=st.sampled_from([0, 64, 128, 256]),
       mask_bits=st.integers(0, 255))
def test_random_operands_and_schedules(seed, n_rows, n_cols, d, n_hops, mean_deg, heavy, threshold, rpw, variant,
                                       slice_cols, mask_bits):
    from h2gcn_amd import HopPlan

    # any number of hop groups up to H2GCN_MAX_HOPS = 8 (the reference takes whatever --adj_nhood lists,
    # h2gcn/datasets/_dataset.py:559-576): above 4 selected hops the adjoint leaves its per-class row lists for the tile walk
    # (kShortSumHops), and (rows_per_wave + 1) * n_hops reaches the 64 row pointers one wave-wide load holds at 7 x 8
    rng = np.random.default_rng(seed)
    hops = [_csr(rng, n_rows, n_cols, mean_deg * (k % 3 + 1), heavy and k == 0, 0.15) for k in range(n_hops)]
    x = rng.uniform(-1, 1, (n_cols, d)).astype(np.float32)
    dev = torch.device("cuda:0")
    plan = HopPlan.from_scipy(hops, dev, build_transpose=True, long_row_threshold=threshold, rows_per_wave=rpw,
                              variant=variant, slice_cols=slice_cols)
    sel = [k for k in range(n_hops) if (mask_bits >> k) & 1] or None
    hsel = hops if sel is None else [hops[k] for k in sel]
    y = plan.spmm(torch.from_numpy(x).to(dev), hops=sel).cpu().numpy()
    want = og.gcn_layer_f64acc(hsel, x)
    mag = og.gcn_layer_f64acc([abs(h) for h in hsel], np.abs(x))
    assert y.shape == want.shape
    assert (np.abs(y - want) <= 1e-5 * np.maximum(1.0, mag)).all()
    # ... and the bits are those of the library's documented summation tree, whatever schedule the draw picked
    assert np.array_equal(y, og.gcn_layer_tree(hsel, x, long_threshold=threshold or 256))
    w = rng.uniform(-1, 1, want.shape).astype(np.float32)
    dx = plan.spmm_t(torch.from_numpy(w).to(dev), hops=sel).cpu().numpy()
    want_t = sum(h.T.astype(np.float64) @ w[:, k, :].astype(np.float64) for k, h in enumerate(hsel))
    mag_t = sum(abs(h).T.astype(np.float64) @ np.abs(w[:, k, :]).astype(np.float64) for k, h in enumerate(hsel))
    assert (np.abs(dx - want_t) <= 1e-5 * np.maximum(1.0, mag_t)).all()
    assert np.array_equal(dx, og.gcn_layer_grad_tree(hsel, w, n_cols, long_threshold=threshold or 256))
    # ... and the accumulating adjoint lands exactly `old + sum` in a strided slot of a wider buffer
    pad = int(rng.integers(0, 4))
    wide = torch.from_numpy(rng.uniform(-1, 1, (n_cols, d + 2 * pad + 1)).astype(np.float32)).to(dev)
    before = wide.clone()
    plan.spmm_t(torch.from_numpy(w).to(dev), hops=sel, out=wide[:, pad:pad + d], accumulate=True)
    assert torch.equal(wide[:, pad:pad + d], before[:, pad:pad + d] + torch.from_numpy(dx).to(dev))
    assert torch.equal(wide[:, :pad], before[:, :pad]) and torch.equal(wide[:, pad + d:], before[:, pad + d:])
