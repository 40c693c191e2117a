"""CPU suite: the oracle against the golden fixtures (reference outputs) and against itself."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_planetoid_golden, load_syn_products_golden
from oracle import gcn_layer as og
from oracle import operands as oo


def _same_csr(a, b, exact=True):
    a = sp.csr_matrix(a); a.sort_indices()
    b = sp.csr_matrix(b); b.sort_indices()
    assert a.shape == b.shape
    assert np.array_equal(a.indptr, b.indptr)
    assert np.array_equal(a.indices, b.indices)
    if exact:
        assert np.array_equal(a.data.astype(np.float32), b.data.astype(np.float32))
    else:
        np.testing.assert_allclose(a.data, b.data, rtol=1e-6)


@pytest.mark.parametrize("name,expect_nnz", [("cora", [2708, 10556, 86332]), ("citeseer", [3327, 9104, 37826])])
def test_operand_restatement_matches_reference(name, expect_nnz):
    g = load_planetoid_golden(name)
    assert list(g["split_nnz"]) == expect_nnz  # SURVEY.md §8a probe values
    adj = oo.remove_eye(g["adj_raw"])
    _same_csr(adj, g["adj_noeye"])
    splits = oo.nhood_split(adj, 2)
    assert [sp.csr_matrix(s).nnz for s in splits] == expect_nnz
    for ntype, tag in ((oo.SYM, "sym"), (oo.RW, "rw")):
        for k in (1, 2):
            _same_csr(oo.normalize(splits[k], ntype), g[f"hop{k}_{tag}"])  # bit-exact after the fp32 cast
    hops = oo.adj_norm_hops(adj, ("0,1", "2"), oo.SYM)
    _same_csr(hops[0], g["hop01_sym"])
    _same_csr(hops[1], g["hop2_sym"])
    _same_csr(oo.row_normalize_features(g["feat_raw"]), g["feat_rownorm"])


def test_citeseer_has_empty_rows_and_zero_scaling():
    g = load_planetoid_golden("citeseer")
    d1 = np.diff(g["hop1_rw"].indptr)
    d2 = np.diff(g["hop2_rw"].indptr)
    assert (d1 == 0).sum() == 48 and (d2 == 0).sum() == 653  # SURVEY.md §8c probe
    assert np.isfinite(g["hop1_sym"].data).all() and np.isfinite(g["hop2_rw"].data).all()


def test_canonical_csr_order():
    g = load_planetoid_golden("cora")
    ip, ix, da = oo.to_canonical_csr(g["hop2_sym"])
    assert da.dtype == np.float32 and ix.dtype == np.int32 and ip.dtype == np.int64
    for r in (0, 1, 1000, 2707):
        seg = ix[ip[r]:ip[r + 1]]
        assert np.all(np.diff(seg) > 0)
    a = g["hop2_sym"]
    assert abs(a - a.T).max() == 0  # SYM-normalised undirected graph is symmetric


def test_c_oracle_equals_scipy_loop_nest_and_bounds_fp64():
    g = load_planetoid_golden("cora")
    hops = [g["hop1_sym"], g["hop2_sym"]]
    rng = np.random.Generator(np.random.PCG64(123))
    x = rng.uniform(-1, 1, size=(g["n"], 64)).astype(np.float32)
    y_c = og.gcn_layer_c(hops, x)
    y_s = og.gcn_layer_scipy(hops, x)
    assert y_c.shape == (g["n"], 2, 64)
    assert np.array_equal(y_c, y_s)  # same loop nest, same rounding
    y64 = og.gcn_layer_f64acc(hops, x)
    assert np.abs(y_c - y64).max() < 2e-6
    assert np.abs(og.gcn_layer_c(hops, x, fma=True) - y64).max() < 2e-6
    # literal COO loop (stored order) == CSR loop
    import ctypes as C
    a = sp.coo_matrix(hops[1]); order = np.lexsort((a.col, a.row))
    rows = a.row[order].astype(np.int64); cols = a.col[order].astype(np.int64); vals = a.data[order].astype(np.float32)
    out = np.empty((g["n"], 64), dtype=np.float32)
    og._lib().oracle_spmm_coo_f32(C.c_int64(len(vals)), rows.ctypes.data_as(C.c_void_p), cols.ctypes.data_as(C.c_void_p),
                                  vals.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_int64(64), C.c_int64(64),
                                  out.ctypes.data_as(C.c_void_p), C.c_int64(64), C.c_int64(g["n"]))
    assert np.array_equal(out, y_c[:, 1, :])
    # two stacked layers = r1, r2 of H2GCN-2: r2 = layer(flatten(r1))
    r1 = y_c.reshape(g["n"], 128)
    r2 = og.gcn_layer_c(hops, r1).reshape(g["n"], 256)
    r2_64 = og.gcn_layer_f64acc(hops, og.gcn_layer_f64acc(hops, x).reshape(g["n"], 128).astype(np.float32))
    assert np.abs(r2 - r2_64.reshape(g["n"], 256)).max() < 2e-6


def test_grad_oracle_matches_dense_adjoint():
    rng = np.random.default_rng(5)
    hops = [sp.random(40, 30, 0.2, format="csr", random_state=1, dtype=np.float32),
            sp.random(40, 30, 0.1, format="csr", random_state=2, dtype=np.float32)]
    dy = rng.standard_normal((40, 2, 9)).astype(np.float32)
    want = sum(h.toarray().astype(np.float64).T @ dy[:, k, :].astype(np.float64) for k, h in enumerate(hops))
    np.testing.assert_allclose(og.gcn_layer_grad_c(hops, dy, 30), want, atol=1e-5)
    np.testing.assert_allclose(og.gcn_layer_grad_scipy(hops, dy, 30), want, atol=1e-5)


def test_rows_subset_matches_full():
    a, _, h = load_syn_products_golden()
    assert abs(h - 0.2) < 0.01 and a.shape == (10000, 10000)
    adj = oo.remove_eye(a)
    hops = oo.adj_norm_hops(adj, ("1", "2"), oo.SYM)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (10000, 8)).astype(np.float32)
    rows = [0, 17, 9999]
    sub = og.rows_subset([oo.to_canonical_csr(m) for m in hops], x, rows)
    full = og.gcn_layer_f64acc(hops, x)
    np.testing.assert_allclose(sub, full[rows], atol=1e-12)


def test_oracle_reproduces_stored_layer_outputs():
    """tests/golden/cora_layer_outputs.npz (scipy csr @ dense, made by make_golden.py): the C oracle gives the
    same fp32 rows bit-for-bit and the same column sums."""
    from conftest import GOLDEN
    z = np.load(GOLDEN / "cora_layer_outputs.npz")
    g = load_planetoid_golden("cora")
    hops = [g["hop1_sym"], g["hop2_sym"]]
    x = np.random.Generator(np.random.PCG64(123)).uniform(-1, 1, (g["n"], 64)).astype(np.float32)
    r1 = og.gcn_layer_c(hops, x).reshape(g["n"], 128)
    r2 = og.gcn_layer_c(hops, r1).reshape(g["n"], 256)
    assert np.array_equal(r1[z["rows"]], z["r1_rows"]) and np.array_equal(r2[z["rows"]], z["r2_rows"])
    assert np.abs(r1.astype(np.float64).sum(0) - z["r1_colsum64"]).max() < 1e-4
    assert np.abs(r2.astype(np.float64).sum(0) - z["r2_colsum64"]).max() < 1e-4


def test_oracle_agrees_with_an_independent_implementation():
    """Third implementation as a cross-check of the restatement: torch's CPU sparse-CSR matmul (different code
    base, different loop structure) agrees with the C oracle to fp32 rounding on the Cora operands."""
    import torch

    g = load_planetoid_golden("cora")
    x = np.random.Generator(np.random.PCG64(5)).uniform(-1, 1, (g["n"], 32)).astype(np.float32)
    for name in ("hop1_sym", "hop2_rw"):
        m = g[name]
        t = torch.sparse_csr_tensor(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
                                    torch.from_numpy(m.data.astype(np.float32)), size=m.shape)
        got = (t @ torch.from_numpy(x)).numpy()
        assert np.abs(got - og.gcn_layer_c([m], x)[:, 0, :]).max() <= 2e-6


def test_threaded_oracle_layer_has_identical_bits():
    """oracle_gcn_layer_f32_mt (bench.py's all-cores CPU baseline) only spreads output rows over threads."""
    import scipy.sparse as sp

    from oracle import gcn_layer as og

    rng = np.random.default_rng(3)
    hops = [sp.random(3000, 2000, 0.01, format="csr", random_state=k, dtype=np.float32) for k in range(3)]
    x = rng.uniform(-1, 1, (2000, 48)).astype(np.float32)
    assert np.array_equal(og.gcn_layer_c(hops, x), og.gcn_layer_c(hops, x, threads=True))
