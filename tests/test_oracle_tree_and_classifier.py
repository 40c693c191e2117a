"""CPU suite: the two round-3 restatements in oracle/ that pin BITS of the HIP path -- the canonical summation tree
(oracle_spmm_tree_f32) and the counter-based dropout mask of the classifier kernels (oracle/classifier.py).  They are
checked here against independent formulations, so that the GPU suite's bit-exact comparisons rest on something."""
import numpy as np
import scipy.sparse as sp

from oracle import classifier as oc
from oracle import gcn_layer as og


def _rand_hops(n, seed, dens=(0.03, 0.2)):
    rng = np.random.default_rng(seed)
    hops = []
    for k, d in enumerate(dens):
        m = sp.random(n, n, d, format="csr", random_state=seed + k, dtype=np.float32)
        m.data = rng.uniform(-1, 1, m.nnz).astype(np.float32)
        m.sort_indices()
        hops.append(m)
    return hops


def test_tree_restatement_is_a_regrouping_of_the_reference_sum():
    """Same terms as the reference's sequential loop, only grouped differently: on integer-valued operands (every partial
    sum exact in fp32) the tree, the sequential C port and scipy agree EXACTLY -- short rows, rows cut into 64-neighbour
    chunks over four "waves", forward and adjoint; on real-valued operands both are within fp32 round-off of fp64."""
    n, d = 400, 9
    rng = np.random.default_rng(1)
    hops = _rand_hops(n, 3)
    for h in hops:
        h.data = rng.integers(-3, 4, h.nnz).astype(np.float32)
    hops[1] = sp.csr_matrix(sp.vstack([hops[1][:5], sp.csr_matrix(np.ones((1, n), dtype=np.float32)), hops[1][6:]]))   # a 400-nnz row
    x = rng.integers(-4, 5, (n, d)).astype(np.float32)
    w = rng.integers(-2, 3, (n, 2, d)).astype(np.float32)
    want = og.gcn_layer_c(hops, x)
    want_t = og.gcn_layer_grad_c(hops, w, n)
    for thr in (8, 64, 256, 100000):
        assert np.array_equal(og.gcn_layer_tree(hops, x, long_threshold=thr), want), thr
        assert np.array_equal(og.gcn_layer_grad_tree(hops, w, n, long_threshold=thr), want_t), thr
    # real-valued: round-off only
    hops = _rand_hops(n, 7)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    exact = og.gcn_layer_f64acc(hops, x)
    mag = og.gcn_layer_f64acc([abs(h) for h in hops], np.abs(x))
    for thr in (8, 256):
        t = og.gcn_layer_tree(hops, x, long_threshold=thr)
        assert (np.abs(t - exact) <= 2e-6 * np.maximum(mag, 1.0)).all()
    assert not np.array_equal(og.gcn_layer_tree(hops, x, 8), og.gcn_layer_tree(hops, x, 256))   # the threshold is part of the tree


def test_tree_of_one_row_by_hand():
    """P[j mod 4] accumulation and (P0 + P1) + (P2 + P3), spelled out with numpy float32 FMAs for one 7-nonzero row."""
    a = np.array([0.1, -0.7, 0.3, 0.9, -0.2, 0.5, 0.25], dtype=np.float32)
    xcol = np.array([1.5, -2.25, 0.125, 3.0, -1.0, 0.75, 2.5], dtype=np.float32)
    m = sp.csr_matrix((a, np.arange(7), [0, 7]), shape=(1, 7))
    got = og.gcn_layer_tree([m], xcol.reshape(7, 1))[0, 0, 0]

    def fma(p, q, r):   # one rounding: exact in float64 for fp32 inputs of this size, then rounded once
        return np.float32(np.float64(p) * np.float64(q) + np.float64(r))
    P = [np.float32(0)] * 4
    for j in range(7):
        P[j % 4] = fma(a[j], xcol[j], P[j % 4])
    assert got == np.float32(np.float32(P[0] + P[1]) + np.float32(P[2] + P[3]))


def test_mask_generator_statistics_and_structure():
    for keep in (0.5, 0.9, 0.25):
        m = oc.keep_mask(4000, 448, keep, seed=0xDEADBEEF12345, step=3)
        assert abs(m.mean() - keep) < 2e-3
        assert np.abs(m.mean(0) - keep).max() < 0.04 and np.abs(m.mean(1) - keep).max() < 0.12   # every column / row
        assert abs((m[:, :-1] & m[:, 1:]).mean() - keep * keep) < 3e-3                           # neighbours independent
        assert abs((m[:-1] & m[1:]).mean() - keep * keep) < 3e-3
    a = oc.keep_mask(100, 30, 0.5, 11, 5)
    assert np.array_equal(a, oc.keep_mask(100, 30, 0.5, 11, 5))                  # a pure function of (seed, step, index)
    assert not np.array_equal(a, oc.keep_mask(100, 30, 0.5, 11, 6)) and not np.array_equal(a, oc.keep_mask(100, 30, 0.5, 12, 5))
    assert np.array_equal(oc.keep_mask(40, 30, 0.5, 11, 5, row0=60), a[60:])     # rows are addressed globally
    for keep in (0.5, 0.9):                                                      # 8-bit / 16-bit fields
        s5, s6 = oc.keep_mask(3000, 448, keep, 77, 5), oc.keep_mask(3000, 448, keep, 77, 6)
        assert abs((s5 & s6).mean() - keep * keep) < 3e-3                        # consecutive steps independent
        far = oc.keep_mask(3000, 448, keep, 77, 5, row0=2 ** 33)                 # group ids beyond 32 bits
        assert abs(far.mean() - keep) < 2e-3 and abs((far & s5).mean() - keep * keep) < 3e-3
    assert oc.keep_mask(5, 7, 1.0, 1, 1).all()
    # widths that are not multiples of 4: the row's last group is partial, columns beyond it do not exist
    assert oc.keep_mask(50, 7, 0.5, 1, 1).shape == (50, 7)


def test_dropout_dense_gradients_by_finite_differences():
    rng = np.random.default_rng(0)
    n, k, c = 30, 13, 5
    x, w, b = rng.standard_normal((n, k)), rng.standard_normal((k, c)), rng.standard_normal(c)
    g = rng.standard_normal((n, c))
    seed, step, keep = 99, 4, 0.7
    f = lambda xx, ww: float((oc.dropout_dense(xx, ww, b, keep, seed, step) * g).sum())
    dx, dw, db = oc.dropout_dense_grad(x, w, g, keep, seed, step)
    eps = 1e-6
    for (i, j) in ((0, 0), (7, 12), (29, 5)):
        xp = x.copy(); xp[i, j] += eps
        assert abs((f(xp, w) - f(x, w)) / eps - dx[i, j]) < 1e-4
    for (i, j) in ((0, 0), (12, 4), (5, 2)):
        wp = w.copy(); wp[i, j] += eps
        assert abs((f(x, wp) - f(x, w)) / eps - dw[i, j]) < 1e-4
    assert np.allclose(db, g.sum(0))
    m = oc.keep_mask(n, k, keep, seed, step)
    assert np.array_equal(dx == 0, ~m)


def test_keras_adam_restatement_first_step_by_hand_and_cpu_optimizer_follows_it():
    """Keras / TensorFlow Adam (reference `H2GCN.py:62-63, 73`): epsilon joins the UNCORRECTED sqrt(v).  First step by hand:
    m = 0.1 g, v = 0.001 g^2, alpha = lr sqrt(0.001) / 0.1, update = alpha m / (sqrt(v) + eps) = lr g / (|g| + eps / sqrt(0.001))
    -- so a gradient of 1e-6 moves a weight by lr * 1e-6 / (1e-6 + 3.16e-6) = 0.24 lr, where torch's Adam (epsilon on the
    corrected sqrt(v)) would move it by lr * 1e-6 / (1e-6 + 1e-7) = 0.91 lr."""
    import torch
    from oracle import keras_adam as ok
    from h2gcn_amd.optim import KerasAdam

    g = np.array([1.0, -2.0, 1e-6, 0.0], np.float32)
    p, m, v = ok.keras_adam_step(np.zeros(4, np.float32), g, np.zeros(4), np.zeros(4), 1, lr=0.01)
    want = -0.01 * g.astype(np.float64) / (np.abs(g.astype(np.float64)) + 1e-7 / np.sqrt(0.001))
    assert np.allclose(p, want, rtol=2e-4, atol=1e-9) and abs(p[2] / -0.01 - 0.2403) < 2e-3 and p[3] == 0
    assert np.allclose(m, 0.1 * g, rtol=1e-6) and np.allclose(v, 0.001 * g * g, rtol=3e-5)   # fp32(1) - fp32(0.999) = 0.99998713e-3
    # the CPU branch of the optimizer is the same arithmetic, step after step
    rng = np.random.default_rng(0)
    w0 = rng.normal(size=(5, 3)).astype(np.float32)
    w = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    opt = KerasAdam([w], lr=0.01)
    p, m, v = w0.copy(), np.zeros_like(w0), np.zeros_like(w0)
    for t in range(1, 8):
        gt = (rng.normal(size=w0.shape) * (10.0 ** rng.integers(-6, 1))).astype(np.float32)
        w.grad = torch.from_numpy(gt.copy())
        opt.step()
        p, m, v = ok.keras_adam_step(p, gt, m, v, t, lr=0.01)
        assert np.allclose(w.detach().numpy(), p, rtol=2e-6, atol=3e-7), t      # pow / sqrt of torch vs numpy: last-bit differences
    assert "step_dev" in opt.state_dict()["param_groups"][0]


def test_keras_rmsprop_by_hand():
    """TensorFlow's ApplyRMSProp puts epsilon inside the square root: first step ms = 0.1 g^2, update = lr g / sqrt(0.1 g^2 + eps)."""
    import torch
    from h2gcn_amd.optim import KerasRMSprop
    g = np.array([1.0, -0.5, 1e-5, 0.0])
    w = torch.nn.Parameter(torch.zeros(4, dtype=torch.float64))
    w.grad = torch.from_numpy(g.copy())
    KerasRMSprop([w], lr=0.01).step()
    assert np.allclose(w.detach().numpy(), -0.01 * g / np.sqrt(0.1 * g * g + 1e-7), rtol=1e-12)
    assert abs(w[2].item() / -0.01 - 1e-5 / np.sqrt(1e-11 + 1e-7)) < 1e-9        # 0.0316: the epsilon floor, not 1 / sqrt(0.1) = 3.16


def test_threaded_tree_oracle_has_the_bits_of_the_serial_one_and_the_checksum_helpers_count_right():
    """oracle_spmm_tree_f32_mt (rows spread over OpenMP threads, what the full-size GPU checks run) == oracle_spmm_tree_f32 bit
    for bit, forward and adjoint, long segments included; the whole-array helpers agree with numpy."""
    import scipy.sparse as sp
    from oracle import fullsize as fs
    from oracle import gcn_layer as og

    rng = np.random.default_rng(11)
    n, d = 900, 40
    hops = []
    for k, dens in enumerate((0.01, 0.08)):
        m = sp.random(n, n, dens, format="lil", random_state=k, dtype=np.float32)
        m[5, :700] = 1.0                                   # a long segment (>= 256): the 4-"wave" branch of the tree
        m = sp.csr_matrix(m)
        m.data[:] = rng.uniform(-1, 1, len(m.data)).astype(np.float32)
        m.sort_indices()
        hops.append(m)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    parts = [og._csr_parts(m) for m in hops]
    y0, y1 = og.gcn_layer_tree(hops, x), fs.gcn_layer_tree_mt(parts, x)
    assert np.array_equal(y0, y1)
    assert np.array_equal(og.gcn_layer_c(hops, x), fs.gcn_layer_seq_mt(parts, x))
    dy = rng.uniform(-1, 1, (n, 2, d)).astype(np.float32)
    t_parts = []
    for m in hops:
        t = m.T.tocsr()
        t.sort_indices()
        t_parts.append(og._csr_parts(t))
    assert np.array_equal(og.gcn_layer_grad_tree(hops, dy, n), fs.gcn_layer_grad_tree_mt(t_parts, dy))
    assert fs.bits_checksum(y0) == int(y0.view(np.int32).astype(np.int64).sum())
    assert fs.count_bit_mismatches(y0, y1) == (0, -1)
    y2 = y1.copy()
    y2.reshape(-1)[[77, 5000]] += np.float32(1e-3)
    y2.reshape(-1)[9] = -y2.reshape(-1)[9] if y2.reshape(-1)[9] != 0 else np.float32(-0.0)
    assert fs.count_bit_mismatches(y0, y2) == (3, 9)
    assert abs(fs.max_abs_diff(y0, y2) - float(np.abs(y0.astype(np.float64) - y2).max())) < 1e-12
    y2.reshape(-1)[3] = np.nan
    assert fs.max_abs_diff(y0, y2) == float("inf")


def test_bench_checksum_constants_come_from_the_oracle():
    """bench.py's N1_CHECKSUMS (what `checksum_matches_n1` compares with) is the ORACLE's checksum of Y -- recomputed here on the
    CPU for BASELINE configs[2] (the arxiv shape: a second; the products constant is re-derived the same way by `python -m
    oracle.fullsize products`, ~30 s on 8 cores, and asserted on the GPU box in tests/test_fullsize_parity_gpu.py)."""
    import sys
    from pathlib import Path
    from oracle import fullsize as fs

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench

    ck, nnz = fs.oracle_checksum("arxiv", block_rows=60_000)       # three row blocks: the sum does not depend on the blocking
    assert ck == bench.N1_CHECKSUMS[("arxiv", 128)]
    assert nnz == [1203951, 1205200]
