"""GPU suite: the HIP hop aggregation (through the C ABI) against the CPU oracle.

Tolerance: BASELINE.json's north star asks for layer outputs within 1e-5 of the reference in fp32.  ATOL below
is that 1e-5 (absolute, on outputs whose magnitude is O(1) or smaller); the only arithmetic difference between
the kernel and the oracle's sequential loop is the association order of the per-row sum (the library's documented
canonical tree: 4 interleaved partials, 4 waves for long rows) and fused multiply-add.  That tree is restated in the
oracle (``og.gcn_layer_tree``), so every kernel variant / slice width / feature chunking is ALSO required to match one
fixed function of the inputs bit for bit (``test_bits_are_the_canonical_tree_*``).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_planetoid_golden, load_syn_products_golden
from oracle import gcn_layer as og
from oracle import operands as oo

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def assert_close(y, hops, x, want=None, atol=ATOL):
    """|y - want| <= atol * max(1, sum_j |a_ij| |x_jc|): 1e-5 absolute on reference-like operands (normalised
    adjacency, |x| <= 1, so the magnitude term is <= 1) and the same RELATIVE accuracy on adversarial operands
    whose rows sum thousands of O(1) terms."""
    if want is None:
        want = og.gcn_layer_f64acc(hops, x)
    mag = og.gcn_layer_f64acc([abs(sp.csr_matrix(h)) for h in hops], np.abs(x))
    bad = np.abs(y - want) > atol * np.maximum(1.0, mag)
    assert not bad.any(), f"{bad.sum()} elements off, worst {np.abs(y - want).max():.3e}"


def dev():
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    return torch.device("cuda:0")


def run_hip(hops, x, **kw):
    from h2gcn_amd import HopPlan

    spmm_kw = {k: kw.pop(k) for k in ("hops_sel",) if k in kw}
    plan = HopPlan.from_scipy(hops, dev(), **kw)
    y = plan.spmm(torch.from_numpy(np.ascontiguousarray(x)).to(dev()), hops=spmm_kw.get("hops_sel"))
    torch.cuda.synchronize()
    return y.cpu().numpy(), plan


def rand_csr(n_rows, n_cols, density, seed, empty_frac=0.0):
    rng = np.random.default_rng(seed)
    m = sp.random(n_rows, n_cols, density, format="csr", random_state=seed, dtype=np.float32)
    m.data = rng.uniform(-1, 1, m.nnz).astype(np.float32)
    if empty_frac:
        keep = rng.random(n_rows) >= empty_frac
        m = sp.diags(keep.astype(np.float32)) @ m
        m.eliminate_zeros()
    m = sp.csr_matrix(m)
    m.sort_indices()
    return m


# ----------------------------------------------------------------------------- golden graphs (configs 1, 2)
@pytest.mark.parametrize("norm", ["sym", "rw"])
@pytest.mark.parametrize("d", [64, 128])
def test_cora_golden_operands(d, norm):
    g = load_planetoid_golden("cora")
    hops = [g[f"hop1_{norm}"], g[f"hop2_{norm}"]]
    x = np.random.Generator(np.random.PCG64(123)).uniform(-1, 1, (g["n"], d)).astype(np.float32)
    y, _ = run_hip(hops, x)
    want = og.gcn_layer_c(hops, x)
    assert y.shape == (g["n"], 2, d)
    assert np.abs(y - want).max() <= ATOL
    assert np.abs(y - og.gcn_layer_f64acc(hops, x)).max() <= ATOL
    # rows with no 2-hop neighbours are exactly zero (TF zero-initialises)
    empty = np.diff(hops[1].indptr) == 0
    assert empty.sum() == 141 and not y[empty, 1, :].any()


def test_cora_two_stacked_layers_r1_r2():
    """r1 = [A1 r0 | A2 r0], r2 = [A1 r1 | A2 r1] of H2GCN-2 (reference H2GCN.py:294-346)."""
    from h2gcn_amd import GCNLayer, HopPlan

    g = load_planetoid_golden("cora")
    hops = [g["hop1_sym"], g["hop2_sym"]]
    r0 = np.abs(np.random.Generator(np.random.PCG64(7)).standard_normal((g["n"], 64))).astype(np.float32)
    plan = HopPlan.from_scipy(hops, dev())
    layer = GCNLayer()
    r1 = layer(plan, torch.from_numpy(r0).to(dev())).flatten(1)
    r2 = layer(plan, r1).flatten(1)
    w1 = og.gcn_layer_c(hops, r0).reshape(g["n"], 128)
    w2 = og.gcn_layer_c(hops, w1).reshape(g["n"], 256)
    assert np.abs(r1.cpu().numpy() - w1).max() <= ATOL
    assert np.abs(r2.cpu().numpy() - w2).max() <= ATOL


def test_cora_stored_layer_outputs():
    """HIP r1 / r2 against the STORED golden rows and fp64 column sums (tests/golden/cora_layer_outputs.npz)."""
    from conftest import GOLDEN
    from h2gcn_amd import GCNLayer, HopPlan

    z = np.load(GOLDEN / "cora_layer_outputs.npz")
    g = load_planetoid_golden("cora")
    plan = HopPlan.from_scipy([g["hop1_sym"], g["hop2_sym"]], dev())
    x = np.random.Generator(np.random.PCG64(123)).uniform(-1, 1, (g["n"], 64)).astype(np.float32)
    r1 = GCNLayer()(plan, torch.from_numpy(x).to(dev())).flatten(1)
    r2 = GCNLayer()(plan, r1).flatten(1)
    assert np.abs(r1.cpu().numpy()[z["rows"]] - z["r1_rows"]).max() <= ATOL
    assert np.abs(r2.cpu().numpy()[z["rows"]] - z["r2_rows"]).max() <= ATOL
    assert np.abs(r1.double().sum(0).cpu().numpy() - z["r1_colsum64"]).max() <= 1e-4
    assert np.abs(r2.double().sum(0).cpu().numpy() - z["r2_colsum64"]).max() <= 1e-4


@pytest.mark.parametrize("d", [64, 128])
def test_syn_products_fixture(d):
    """BASELINE.json configs[1]: syn-products h=0.2, |V|=10k -- graph from the reference generator, exact 2-hop
    ring from the oracle's nhoodSplit restatement, SYM norm, H2GCN-2 widths 64 and 128."""
    a, labels, _ = load_syn_products_golden()
    hops = oo.adj_norm_hops(oo.remove_eye(a), ("1", "2"), oo.SYM)
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((10000, d)) + labels[:, None] * 0.1).astype(np.float32)  # class-conditional
    y, plan = run_hip(hops, x)
    want = og.gcn_layer_c(hops, x)
    assert np.abs(y - want).max() <= ATOL
    assert plan.info(1)["nnz"] == hops[1].nnz


# ----------------------------------------------------------------------------- adversarial shapes
@pytest.mark.parametrize("d", [1, 3, 4, 16, 32, 48, 64, 80, 96, 100, 128, 132, 200, 256, 260, 448])
def test_feature_widths(d):
    hops = [rand_csr(301, 301, 0.05, 1, empty_frac=0.1), rand_csr(301, 301, 0.15, 2, empty_frac=0.3)]
    x = np.random.default_rng(d).uniform(-1, 1, (301, d)).astype(np.float32)
    y, _ = run_hip(hops, x)
    assert_close(y, hops, x)


@pytest.mark.parametrize("d", [20, 36, 100, 132, 200, 300, 452])
def test_partial_last_slice_widths(d):
    """Widths that are multiples of 4 but not of the slice width (--hidden 100 -> d = 100, 200) run through the
    sliced float4 kernels with a masked last slice: every slice width, the pipelined segment walk, a long row, and the
    adjoint, all against the oracle; the neighbouring memory of a strided output must stay untouched."""
    from h2gcn_amd import HopPlan

    n = 700
    hops = [rand_csr(n, n, 0.01, 1, empty_frac=0.1), rand_csr(n, n, 0.06, 2, empty_frac=0.2)]
    hops[0] = sp.csr_matrix(sp.vstack([hops[0][:5], sp.csr_matrix(np.full((1, n), 0.01, dtype=np.float32)), hops[0][6:]]))
    rng = np.random.default_rng(d)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    w = rng.uniform(-1, 1, (n, 2, d)).astype(np.float32)
    want = og.gcn_layer_f64acc(hops, x)
    want_t = og.gcn_layer_grad_c(hops, w, n)
    for sc in (0, 64, 128, 256):
        for variant in (0, 2):
            plan = HopPlan.from_scipy(hops, dev(), slice_cols=sc, variant=variant, long_row_threshold=128, build_transpose=True)
            ybuf = torch.full((n, 2, d + 12), 7.0, device=dev())          # guard columns right of every hop block
            y = plan.spmm(torch.from_numpy(x).to(dev()), out=ybuf[:, :, :d])
            assert_close(y.cpu().numpy(), hops, x, want)
            assert bool((ybuf[:, :, d:] == 7.0).all()), (sc, variant)
            dx = plan.spmm_t(torch.from_numpy(w).to(dev()))
            assert np.abs(dx.cpu().numpy() - want_t).max() <= 2e-5, (sc, variant)


def test_slice_major_scratch_copy_is_bitwise_identical():
    """X with a 1 KiB row stride (contiguous [N, 256]) far beyond the caches: the launch gathers from a slice-major
    scratch copy (h2gcn_spmm_hops_ws_f32); same bits as the plain launch, sampled rows against the oracle."""
    import ctypes as C

    from h2gcn_amd import HopPlan, _capi, synth

    n, d, device = 540_000, 256, dev()
    degs = [synth.synth_degrees(n, 9_500_000, s, n) for s in (5, 6)]
    csr = [synth.synth_hop_rows(degs[k], n, (5, 6)[k], 0, n, device) for k in range(2)]
    # (64-column slices forced: at this size -- X = 553 MB -- the heuristic would not slice, at products scale it does)
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n, slice_cols=64)
    x = synth.synth_features(d, 7, 0, n, device)
    L = _capi.lib()
    xp = C.c_void_p(x.data_ptr())
    assert L.h2gcn_spmm_workspace_bytes(plan._handle, 0, 0, xp, d, 0, d) == n * d * 4          # contiguous: 1 KiB stride
    assert L.h2gcn_spmm_workspace_bytes(plan._handle, 0, 0, xp, d + 32, 0, d) == 0             # padded rows do not alias
    assert L.h2gcn_spmm_workspace_bytes(plan._handle, 0, 0, xp, 64, 0, 64) == 0
    y_ws = plan.spmm(x)
    plan.use_workspace = False
    y_plain = plan.spmm(x)
    assert torch.equal(y_ws, y_plain)
    # a workspace that is too small silently selects the plain launch
    small = torch.empty(1024, dtype=torch.uint8, device=device)
    y3 = torch.empty_like(y_ws)
    opts = _capi.LaunchOpts(struct_size=C.sizeof(_capi.LaunchOpts), flags=0, workspace=small.data_ptr(), workspace_bytes=1024, bias=None)
    _capi.check(L.h2gcn_spmm_hops_opts_f32(plan._handle, 0, C.c_void_p(x.data_ptr()), d, d, C.c_void_p(y3.data_ptr()), 2 * d, d,
                                           C.byref(opts), None))
    torch.cuda.synchronize()
    assert torch.equal(y3, y_plain)
    for r0 in (0, 12345, n - 8):
        parts = [synth.synth_hop_rows_np(degs[k], n, (5, 6)[k], r0, r0 + 8) for k in range(2)]
        cols = np.unique(np.concatenate([q[1] for q in parts]))
        xs = x[torch.from_numpy(cols.astype(np.int64)).to(device)].cpu().numpy()
        remap = {c: i for i, c in enumerate(cols)}
        local = [(q[0], np.array([remap[c] for c in q[1]], dtype=np.int32), q[2]) for q in parts]
        assert np.abs(y_ws[r0:r0 + 8].cpu().numpy() - og.rows_subset(local, xs, list(range(8)))).max() <= ATOL


def test_scratch_copy_for_rows_that_are_not_line_aligned():
    """d = 200 (`--hidden 100`, second round): rows of 800 B are only 16-B aligned.  With scratch the launch gathers
    from line-aligned 64-column blocks (slices 64, 64, 64, 8 -- the last one masked); results equal the plain launch
    (one masked slice of 256) BIT FOR BIT (canonical summation tree) and the oracle on sampled rows; guard columns of a
    strided output untouched."""
    import ctypes as C

    from h2gcn_amd import HopPlan, _capi, synth

    n, d, device = 700_000, 200, dev()
    degs = [synth.synth_degrees(n, 12_500_000, s, n) for s in (15, 16)]
    csr = [synth.synth_hop_rows(degs[k], n, (15, 16)[k], 0, n, device) for k in range(2)]
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
    x = synth.synth_features(d, 17, 0, n, device)
    L = _capi.lib()
    xp = C.c_void_p(x.data_ptr())
    assert L.h2gcn_spmm_workspace_bytes(plan._handle, 0, 0, xp, d, 0, d) == n * 4 * 64 * 4      # 4 blocks of 64 columns
    assert L.h2gcn_spmm_workspace_bytes(plan._handle, 0, 0, xp, 224, 0, d) == 0                 # rows padded to 896 B: aligned
    assert plan.schedule(d, ld_src=224)["slice_cols"] == 64 and not plan.schedule(d, ld_src=224)["scratch_copy"]   # sliced in place
    assert L.h2gcn_spmm_workspace_bytes(plan._handle, 0, 0, C.c_void_p(x.data_ptr() + 16), 224, 0, d) == n * 4 * 64 * 4   # slot not on a line
    sched = plan.schedule(d)
    assert sched["scratch_copy"] and sched["slice_cols"] == 64 and sched["n_slices"] == 4
    ybuf = torch.full((n, 2, d + 8), 3.0, device=device)
    y_ws = plan.spmm(x, out=ybuf[:, :, :d])
    assert bool((ybuf[:, :, d:] == 3.0).all())
    plan.use_workspace = False
    y_plain = plan.spmm(x)
    assert torch.equal(y_ws, y_plain)
    # rows that DO start on cache lines (stride padded to 224 floats) are sliced 64 columns wide in place: same bits
    xpad = torch.zeros((n, 224), device=device)
    xpad[:, :d] = x
    plan.use_workspace = True
    assert torch.equal(plan.spmm(xpad[:, :d]), y_plain)
    del xpad
    for r0 in (0, 4321, n - 8):
        parts = [synth.synth_hop_rows_np(degs[k], n, (15, 16)[k], r0, r0 + 8) for k in range(2)]
        cols = np.unique(np.concatenate([q[1] for q in parts]))
        xs = x[torch.from_numpy(cols.astype(np.int64)).to(device)].cpu().numpy()
        remap = {c: i for i, c in enumerate(cols)}
        local = [(q[0], np.array([remap[c] for c in q[1]], dtype=np.int32), q[2]) for q in parts]
        assert np.abs(y_ws[r0:r0 + 8].cpu().numpy() - og.rows_subset(local, xs, list(range(8)))).max() <= ATOL


def test_schedule_heuristics_are_what_the_documentation_says():
    """h2gcn_plan_schedule: the launch-time decisions for the regimes DESIGN.md describes, queried on plans whose
    COLUMN space and mean degree mimic the big shapes (no big operand is allocated: the decision needs sizes only)."""
    from h2gcn_amd import HopPlan

    def plan(n_rows, n_cols, deg, seed):
        rng = np.random.default_rng(seed)
        rows = np.repeat(np.arange(n_rows), deg)
        cols = rng.integers(0, n_cols, n_rows * deg)
        m = sp.csr_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(n_rows, n_cols))
        m.sum_duplicates()
        m.sort_indices()
        return HopPlan.from_scipy([m, m], dev(), build_transpose=True)

    products_like = plan(2000, 2_400_000, 50, 1)          # X = 1.2 GB at d = 128, mean degree 50
    s = products_like.schedule(128)
    assert (s["slice_cols"], s["n_slices"], s["segment_walk"], s["scratch_copy"]) == (64, 2, "wave per segment", False)
    assert products_like.schedule(64)["slice_cols"] == 64 and products_like.schedule(64)["n_slices"] == 1
    s = products_like.schedule(256)                        # contiguous [N, 256]: 1 KiB stride -> slice-major scratch copy
    assert s["scratch_copy"] is False                      # ... only when every column is gathered often enough (nnz >= 32 N)
    dense_like = plan(160_000, 900_000, 150, 2)            # 48M nonzeros over 900k columns, X = 922 MB at d = 256
    s = dense_like.schedule(256)
    assert s["scratch_copy"] and s["slice_cols"] in (64, 128)
    assert not dense_like.schedule(256, ld_src=288)["scratch_copy"]          # padded rows: no aliasing, no copy
    assert dense_like.schedule(300)["scratch_copy"] and dense_like.schedule(300)["slice_cols"] == 64   # rows not line-aligned
    assert not dense_like.schedule(100)["scratch_copy"] and dense_like.schedule(100)["slice_cols"] == 128
    lowdeg_like = plan(4000, 8_000_000, 4, 3)              # mean degree 4
    s = lowdeg_like.schedule(128)
    assert s["segment_walk"].startswith("lane group per segment") and s["slice_cols"] == 128
    assert lowdeg_like.schedule(128, adjoint=True)["segment_walk"].startswith(("wave per segment", "lane group per segment"))
    s32 = lowdeg_like.schedule(32)                                                      # d < 64: one masked 64-column slice
    assert s32["slice_cols"] == 64 and s32["n_slices"] == 1 and s32["segment_walk"].startswith("lane group per segment")
    small = plan(3000, 3000, 20, 4)
    assert small.schedule(128)["slice_cols"] == 128 and small.schedule(448)["slice_cols"] == 64 and small.schedule(448)["n_slices"] == 7


@pytest.mark.parametrize("d", [64, 100, 7])
def test_fused_bias_relu_epilogue(d):
    """Y = relu(A X + b) in one launch (SparseDense.call: bias, then activation, reference _layers.py:45-52): every
    store path (regular, pipelined, long segment, generic scalar) against the oracle + numpy epilogue, bitwise equal
    to the unfused launch followed by torch ops."""
    from h2gcn_amd import HopPlan

    n = 600
    hops = [rand_csr(n, n, 0.02, 3, empty_frac=0.15)]
    hops[0] = sp.csr_matrix(sp.vstack([hops[0][:7], sp.csr_matrix(np.full((1, n), 0.02, dtype=np.float32)), hops[0][8:]]))
    rng = np.random.default_rng(d)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, d).astype(np.float32)
    want = og.gcn_layer_f64acc(hops, x)
    xt, bt = torch.from_numpy(x).to(dev()), torch.from_numpy(b).to(dev())
    for variant in (0, 2):
        plan = HopPlan.from_scipy(hops, dev(), variant=variant, long_row_threshold=128)
        plain = plan.spmm(xt)
        for bias, relu in ((bt, True), (bt, False), (None, True)):
            y = plan.spmm(xt, bias=bias, relu=relu)
            ref = plain + bias if bias is not None else plain
            ref = torch.relu(ref) if relu else ref
            assert torch.equal(y, ref), (variant, bias is not None, relu)
            w = want + (b if bias is not None else 0)
            w = np.maximum(w, 0) if relu else w
            assert np.abs(y.cpu().numpy() - w).max() <= ATOL
    with pytest.raises(ValueError):
        plan.spmm(xt, bias=bt[:-1])


def test_sparse_dense_fused_gradients():
    """SparseDense with the fused bias/ReLU epilogue: outputs and gradients equal the unfused composition."""
    from h2gcn_amd import HopPlan
    from h2gcn_amd.layers import SparseDense

    feats = rand_csr(500, 300, 0.03, 5, empty_frac=0.05)
    plan = HopPlan.from_scipy([feats], dev(), build_transpose=True)
    torch.manual_seed(0)
    fused = SparseDense(300, 64, use_bias=True, activation="relu").to(dev())
    plain = SparseDense(300, 64, use_bias=True, activation=None).to(dev())
    plain.load_state_dict(fused.state_dict())
    with torch.no_grad():
        fused.bias.uniform_(-0.1, 0.1)
        plain.bias.copy_(fused.bias)
    w = torch.rand((500, 64), device=dev())
    yf = fused(plan)
    yp = torch.relu(plain(plan) if False else (plan.spmm(plain.kernel)[:, 0, :] + plain.bias))
    assert torch.equal(yf, yp)
    (yf * w).sum().backward()
    k = plain.kernel.detach().clone().requires_grad_(True)
    bb = plain.bias.detach().clone().requires_grad_(True)
    dense = torch.from_numpy(feats.toarray()).to(dev())
    (torch.relu(dense @ k + bb) * w).sum().backward()
    assert (fused.kernel.grad - k.grad).abs().max().item() <= 1e-4
    assert (fused.bias.grad - bb.grad).abs().max().item() <= 1e-4


def test_set_values_refreshes_forward_and_adjoint_and_sparse_dropout():
    """h2gcn_plan_set_values (SparseDropout's effect on the feature operand, reference _layers.py:7-19): new values on
    the same pattern, forward AND transposed operand follow; the SparseDropout layer masks ~drop_prob of the stored
    values, rescales the survivors, and restores the original values in eval mode."""
    from h2gcn_amd import HopPlan
    from h2gcn_amd._capi import H2GCNError
    from h2gcn_amd.layers import SparseDropout

    feats = rand_csr(800, 500, 0.02, 9)
    plan = HopPlan.from_scipy([feats], dev(), build_transpose=True, keep_permutation=True)
    x = torch.rand((500, 32), device=dev())
    g = torch.rand((800, 1, 32), device=dev())
    rng = np.random.default_rng(0)
    new = feats.copy()
    new.data = rng.uniform(-1, 1, new.nnz).astype(np.float32)
    orig_vals = plan.vals[0]
    plan.set_values(0, torch.from_numpy(new.data).to(dev()))
    assert np.abs(plan.spmm(x).cpu().numpy() - og.gcn_layer_f64acc([new], x.cpu().numpy())).max() <= ATOL
    assert np.abs(plan.spmm_t(g).cpu().numpy() - og.gcn_layer_grad_c([new], g.cpu().numpy(), 500)).max() <= 2e-5
    plan.set_values(0, orig_vals)
    assert np.abs(plan.spmm_t(g).cpu().numpy() - og.gcn_layer_grad_c([feats], g.cpu().numpy(), 500)).max() <= 2e-5
    no_perm = HopPlan.from_scipy([feats], dev(), build_transpose=True)
    with pytest.raises(H2GCNError):
        no_perm.set_values(0, orig_vals)
    drop = SparseDropout(0.4).train()
    torch.manual_seed(1)
    out = drop(plan)
    v = out.vals[0]
    kept = (v != 0)
    assert 0.5 < kept.float().mean().item() < 0.7
    assert torch.allclose(v[kept], orig_vals[kept] / 0.6)
    drop.restore()                                                                # what the training step does after backward
    assert plan.vals[0] is orig_vals
    drop(plan)
    assert plan.vals[0] is not orig_vals
    assert torch.equal(drop.eval()(plan).vals[0], orig_vals)                      # eval: original operand again
    ref_like = SparseDropout(0.4, at_eval=True).eval()                            # the reference never switches it off
    assert (ref_like(plan).vals[0] == 0).float().mean().item() > 0.3
    ref_like.restore()
    assert np.abs(plan.spmm(x).cpu().numpy() - og.gcn_layer_f64acc([feats], x.cpu().numpy())).max() <= ATOL


@pytest.mark.parametrize("n_rows,n_cols", [(1, 1), (5, 9), (63, 64), (64, 63), (65, 1000), (1000, 17), (4097, 333)])
def test_ragged_shapes_rectangular(n_rows, n_cols):
    hops = [rand_csr(n_rows, n_cols, 0.3, 3), rand_csr(n_rows, n_cols, 0.6, 4, empty_frac=0.2)]
    x = np.random.default_rng(0).uniform(-1, 1, (n_cols, 128)).astype(np.float32)
    y, _ = run_hip(hops, x)
    assert y.shape == (n_rows, 2, 128)
    assert_close(y, hops, x)


def test_all_empty_and_zero_nnz():
    hops = [sp.csr_matrix((50, 50), dtype=np.float32), sp.csr_matrix((50, 50), dtype=np.float32)]
    x = np.ones((50, 128), dtype=np.float32)
    y, _ = run_hip(hops, x)
    assert y.shape == (50, 2, 128) and not y.any()


@pytest.mark.parametrize("d", [64, 128, 100])
@pytest.mark.parametrize("threshold", [0, 8, 64, 70])
def test_long_rows_split_across_the_workgroup(d, threshold):
    """One huge row (5000 nnz), a few medium ones, many short: exercises the LDS-staged long-segment path
    (threshold 0 = library default 256; small thresholds push most rows through it)."""
    rng = np.random.default_rng(9)
    n = 6000
    deg = rng.integers(0, 12, n)
    deg[17] = 5000
    deg[n - 1] = 1500
    deg[100:110] = 65
    rows = np.repeat(np.arange(n), deg)
    cols = np.concatenate([rng.choice(n, k, replace=False) for k in deg])
    a1 = sp.csr_matrix((rng.uniform(-1, 1, len(rows)).astype(np.float32), (rows, cols)), shape=(n, n))
    a1.sort_indices()
    a2 = rand_csr(n, n, 0.002, 5)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    y, plan = run_hip([a1, a2], x, long_row_threshold=threshold)
    assert_close(y, [a1, a2], x)  # row 17 sums 5000 O(1) terms: the tolerance scales with sum |a||x|
    if threshold == 0:
        assert plan.info(0)["n_long_segments"] == 2 and plan.info(1)["n_long_segments"] == 0


@pytest.mark.parametrize("rpw", [1, 2, 3, 7])
def test_rows_per_wave_does_not_change_results(rpw):
    hops = [rand_csr(777, 777, 0.05, 1, empty_frac=0.1), rand_csr(777, 777, 0.1, 2)]
    x = np.random.default_rng(1).uniform(-1, 1, (777, 128)).astype(np.float32)
    y0, _ = run_hip(hops, x)
    y1, _ = run_hip(hops, x, rows_per_wave=rpw)
    assert np.array_equal(y0, y1)  # geometry never changes the arithmetic


@pytest.mark.parametrize("d", [128, 256, 200])
def test_column_slices_do_not_change_results(d):
    """The slice-major schedule (Infinity-Cache-sized column slices inside one launch) re-orders independent output
    columns.  Slice widths 64 / 128 / 256 have different lane geometries (4 / 2 / 1 gathered rows per load) but build
    the SAME canonical summation tree: bitwise equal results, equal to the oracle's restatement of that tree.  Narrower
    lane geometries do not exist any more (ABI 3: the 16 / 32-column kernels were slower than the masked 64-column slice on
    every width and had a different tree): asking for them is an argument error."""
    hops = [rand_csr(900, 900, 0.05, 1, empty_frac=0.1), rand_csr(900, 900, 0.1, 2)]
    hops[1] = sp.csr_matrix(sp.vstack([hops[1][:3], sp.csr_matrix(np.full((1, 900), 0.01, dtype=np.float32)), hops[1][4:]]))
    x = np.random.default_rng(1).uniform(-1, 1, (900, d)).astype(np.float32)
    tree = og.gcn_layer_tree(hops, x, long_threshold=512)
    assert_close(tree, hops, x)
    for sc in (0, 64, 128, 256):
        for rpw in (0, 2):
            y, plan = run_hip(hops, x, slice_cols=sc, long_row_threshold=512, rows_per_wave=rpw)
            assert np.array_equal(y, tree), (sc, rpw, plan.schedule(d))
    from h2gcn_amd._capi import H2GCNError
    for sc in (16, 32, 48):
        with pytest.raises(H2GCNError, match="slice_cols"):
            run_hip(hops, x, slice_cols=sc)
    for dn in (16, 32, 48):    # narrow widths: same tree
        yn, plan = run_hip(hops, x[:, :dn].copy(), long_row_threshold=512)
        assert plan.schedule(dn)["slice_cols"] == 64
        assert np.array_equal(yn, tree[:, :, :dn]), dn


@pytest.mark.parametrize("d", [64, 100, 128, 133, 192, 256, 7, 1, 16, 32, 48])
@pytest.mark.parametrize("thr", [0, 20, 100])
def test_bits_are_the_canonical_tree_for_every_schedule(d, thr):
    """SURVEY.md 8(e) "Determinism": the bits of Y must not depend on how the work was scheduled.  Every segment walk
    (variants 0 / 2 / 3 / 5 / 6), every slice width >= 64, launches with and without scratch (incl. the zero-padded copy of
    odd widths and the generic column-tiled kernel), FEATURE CHUNKS of different widths, row blocks, hop selections and
    the adjoint all reproduce ONE function of the inputs: the canonical tree restated in oracle/spmm_oracle.c."""
    from h2gcn_amd import HopPlan

    rng = np.random.default_rng(d * 7 + thr)
    n = 1500
    hops = []
    for k in range(2):
        deg = np.minimum(rng.poisson(5 * (2 * k + 1), n), n)
        deg[rng.random(n) < 0.1] = 0
        deg[rng.integers(0, n, 4)] = [70, 300, 129, 1000]
        rows = np.repeat(np.arange(n), deg)
        cols = np.concatenate([rng.choice(n, kk, replace=False) for kk in deg])
        m = sp.csr_matrix((rng.uniform(-1, 1, len(rows)).astype(np.float32), (rows, cols)), shape=(n, n))
        m.sort_indices()
        hops.append(m)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    w = rng.uniform(-1, 1, (n, 2, d)).astype(np.float32)
    thr_eff = thr if thr else 256
    tree = og.gcn_layer_tree(hops, x, long_threshold=thr_eff)
    tree_t = og.gcn_layer_grad_tree(hops, w, n, long_threshold=thr_eff)
    assert_close(tree, hops, x)
    xt, wt = torch.from_numpy(x).to(dev()), torch.from_numpy(w).to(dev())
    for variant in (0, 2, 3, 5, 6):
        for sc in (0, 64, 128, 256):
            plan = HopPlan.from_scipy(hops, dev(), build_transpose=True, long_row_threshold=thr, variant=variant, slice_cols=sc)
            for use_ws in (True, False):
                plan.use_workspace = use_ws
                assert np.array_equal(plan.spmm(xt).cpu().numpy(), tree), (variant, sc, use_ws, plan.schedule(d))
                assert np.array_equal(plan.spmm_t(wt).cpu().numpy(), tree_t), (variant, sc, use_ws, "adjoint")
            assert np.array_equal(plan.spmm(xt, hops=[1]).cpu().numpy(), tree[:, 1:]), (variant, sc)
    plan = HopPlan.from_scipy(hops, dev(), long_row_threshold=thr)
    if d >= 32:   # feature chunks of unequal widths (what the multi-GPU pipeline does), written into one output
        for widths in ([16, d - 16], [d - 16, 16], [d // 2, d - d // 2], [d // 4, d // 4, d - 2 * (d // 4)]):
            y = torch.empty((n, 2, d), device=dev())
            c0 = 0
            for wd in widths:
                plan.spmm(xt[:, c0:c0 + wd], out=y[:, :, c0:c0 + wd])
                c0 += wd
            assert np.array_equal(y.cpu().numpy(), tree), widths
    # a row block (what a rank of the row partition computes) gives the same rows
    sub = HopPlan.from_scipy([h[400:900] for h in hops], dev(), long_row_threshold=thr)
    assert np.array_equal(sub.spmm(xt).cpu().numpy(), tree[400:900])


@pytest.mark.parametrize("d", [64, 128, 256])
def test_variant_pipelined_segment_walk_is_bitwise_identical(d):
    """variant=2 prefetches the next segment's index chunk while the current one gathers (the default does so on
    short-segment operands), variant=3 is the plain walk.  Prefetching only re-times loads: same arithmetic, same
    bits -- forward and adjoint, with long segments and empty rows in the mix."""
    from h2gcn_amd import HopPlan

    hops = [rand_csr(1200, 900, 0.02, 1, empty_frac=0.2), rand_csr(1200, 900, 0.15, 2, empty_frac=0.05)]
    hops[0] = sp.csr_matrix(sp.vstack([hops[0][:9], sp.csr_matrix(np.full((1, 900), 0.02, dtype=np.float32)), hops[0][10:]]))
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (900, d)).astype(np.float32)).to(dev())
    w = torch.from_numpy(np.random.default_rng(2).uniform(-1, 1, (1200, 2, d)).astype(np.float32)).to(dev())
    ref = HopPlan.from_scipy(hops, dev(), build_transpose=True, long_row_threshold=128, variant=2)
    for rpw in (0, 1, 7):
        alt = HopPlan.from_scipy(hops, dev(), build_transpose=True, long_row_threshold=128, variant=3, rows_per_wave=rpw)
        assert torch.equal(ref.spmm(x), alt.spmm(x))
        assert torch.equal(ref.spmm_t(w), alt.spmm_t(w))
        assert torch.equal(ref.spmm(x, hops=[1]), alt.spmm(x, hops=[1]))
    assert_close(ref.spmm(x).cpu().numpy(), hops, x.cpu().numpy())


@pytest.mark.parametrize("d", [64, 128, 256, 100])
@pytest.mark.parametrize("mean_deg", [1.5, 6, 14])
def test_short_row_mode_is_bitwise_identical(d, mean_deg):
    """variant=5: rounds of G short segments are served one lane group per segment (G segments of a wave in flight at
    once).  Each group keeps the G partial sums the wave-per-segment walk would have spread over its lane groups and
    combines them in the same tree, so the bits are those of variant 3 -- forward, adjoint, hop selection, every
    rows_per_wave, long rows and empty rows in the mix, thresholds below the group width."""
    from h2gcn_amd import HopPlan

    rng = np.random.default_rng(int(d * 10 + mean_deg))
    n = 3000
    hops = []
    for k in range(2):
        deg = np.minimum(rng.poisson(mean_deg * (k + 1), n), n)
        deg[rng.random(n) < 0.1] = 0
        deg[rng.integers(0, n, 3)] = [40, 200, 17]
        rows = np.repeat(np.arange(n), deg)
        cols = np.concatenate([rng.choice(n, kk, replace=False) for kk in deg])
        m = sp.csr_matrix((rng.uniform(-1, 1, len(rows)).astype(np.float32), (rows, cols)), shape=(n, n))
        m.sort_indices()
        hops.append(m)
    x = torch.from_numpy(rng.uniform(-1, 1, (n, d)).astype(np.float32)).to(dev())
    w = torch.from_numpy(rng.uniform(-1, 1, (n, 2, d)).astype(np.float32)).to(dev())
    for thr in (0, 8, 20):
        ref = HopPlan.from_scipy(hops, dev(), build_transpose=True, long_row_threshold=thr, variant=3)
        for rpw in (0, 1, 3, 7):
            for sc in (0, 64, 128):
                for variant in (5, 6):   # in-tile short-row mode / list-driven by segment class
                    alt = HopPlan.from_scipy(hops, dev(), build_transpose=True, long_row_threshold=thr, variant=variant, rows_per_wave=rpw, slice_cols=sc)
                    base = ref   # one canonical tree: the plain walk at the default slice width is the target for every slice width
                    assert torch.equal(base.spmm(x), alt.spmm(x)), (thr, rpw, sc, variant)
                    assert torch.equal(base.spmm_t(w), alt.spmm_t(w)), (thr, rpw, sc, variant)
                    assert torch.equal(base.spmm(x, hops=[1]), alt.spmm(x, hops=[1]))
                    assert torch.equal(base.spmm_t(w[:, :1].contiguous(), hops=[0]), alt.spmm_t(w[:, :1].contiguous(), hops=[0]))
    assert_close(ref.spmm(x).cpu().numpy(), hops, x.cpu().numpy())
    assert alt.schedule(d)["segment_walk"].startswith(("lane group", "wave per segment"))


def _mixed_hops(rng, n, kinds):
    """Hop matrices with prescribed degree mixes: "sparse" (mean 3, all short), "dense" (mean 60), "mixed" (short rows
    scattered among medium and long ones -- a degree sequence that starts at 1), empty rows everywhere."""
    hops = []
    for kind in kinds:
        if kind == "sparse":
            deg = rng.poisson(3, n)
        elif kind == "dense":
            deg = rng.poisson(60, n)
        else:
            deg = np.floor(np.exp(1.3 * rng.standard_normal(n)) * 12 + 0.5).astype(np.int64)
            deg[rng.integers(0, n, 5)] = [16, 17, 255, 256, 700]
        deg = np.minimum(deg, n)
        deg[rng.random(n) < 0.05] = 0
        rows = np.repeat(np.arange(n), deg)
        cols = np.concatenate([rng.choice(n, kk, replace=False) for kk in deg])
        m = sp.csr_matrix((rng.uniform(-1, 1, len(rows)).astype(np.float32), (rows, cols)), shape=(n, n))
        m.sort_indices()
        hops.append(m)
    return hops


@pytest.mark.parametrize("kinds", [("sparse", "dense"), ("mixed", "mixed"), ("sparse", "mixed", "dense"),
                                   ("sparse", "sparse", "mixed", "sparse", "dense")])
@pytest.mark.parametrize("d", [64, 128])
def test_csr_adaptive_segment_classes_in_one_launch(kinds, d):
    """CSR-adaptive dispatch by segment class (round 4): a launch whose selected hops hold short, medium and long
    segments side by side -- a sparse 1-hop matrix next to a dense 2-hop one (the reference's rings,
    h2gcn/datasets/_dataset.py:138-158), short rows scattered among long ones -- serves the short class from the plan's
    binned list (one lane group per segment), the long class by workgroups and the rest by the wave walk, all in ONE
    launch, and the bits are still the canonical tree's: equal to the plain wave walk (variant 3) and to the oracle's
    restatement, forward / hop subsets / adjoint (the SUM-mode list serves selections of up to 4 hops; 5 fall back)."""
    from h2gcn_amd import HopPlan

    rng = np.random.default_rng(len(kinds) * 100 + d)
    n = 2600   # > 256 listed entries per hop: several list workgroups and a partial last wave
    hops = _mixed_hops(rng, n, kinds)
    H = len(hops)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    w = rng.uniform(-1, 1, (n, H, d)).astype(np.float32)
    xt, wt = torch.from_numpy(x).to(dev()), torch.from_numpy(w).to(dev())
    tree = og.gcn_layer_tree(hops, x)
    tree_t = og.gcn_layer_grad_tree(hops, w, n)
    assert_close(tree, hops, x)
    ref = HopPlan.from_scipy(hops, dev(), build_transpose=True, variant=3)
    for variant in (0, 6, 5):      # default (mean >= 16, many short segments: list-driven), forced lists, forced in-tile mode
        for sc in (0, 64, 128):
            for rpw in (0, 1, 7):
                plan = HopPlan.from_scipy(hops, dev(), build_transpose=True, variant=variant, slice_cols=sc, rows_per_wave=rpw)
                cls = plan.segment_classes(d)
                if variant == 5:
                    assert cls["listed"] == -1 and plan.schedule(d)["segment_walk"] == "lane group per segment (short rows)"
                else:
                    assert cls["listed"] > 0 and cls["walks"]["short"] == "lane group per segment (binned list)", (variant, sc, cls)
                    assert plan.schedule(d)["segment_walk"].startswith("lane group per segment (binned")
                y = plan.spmm(xt)
                assert np.array_equal(y.cpu().numpy(), tree), (variant, sc, rpw)
                assert torch.equal(y, ref.spmm(xt))
                dx = plan.spmm_t(wt)
                assert np.array_equal(dx.cpu().numpy(), tree_t), (variant, sc, rpw, "adjoint")
                sel = [0, H - 1]
                assert torch.equal(plan.spmm(xt, hops=sel), ref.spmm(xt, hops=sel))
                assert torch.equal(plan.spmm_t(wt[:, sel].contiguous(), hops=sel), ref.spmm_t(wt[:, sel].contiguous(), hops=sel))
                assert torch.equal(plan.spmm(xt, hops=[H - 1]), ref.spmm(xt, hops=[H - 1]))
    # the per-hop classes add up, and name the sparse hop as (almost) entirely short
    cls = HopPlan.from_scipy(hops, dev(), build_transpose=True).segment_classes(d)
    for k, (hop, m) in enumerate(zip(cls["per_hop"], hops)):
        lens = np.diff(m.indptr)
        assert hop["segments"] == dict(short=int((lens <= 16).sum()), medium=int(((lens > 16) & (lens < 256)).sum()), long=int((lens >= 256).sum()))
        assert hop["nonzeros"] == dict(short=int(lens[lens <= 16].sum()), medium=int(lens[(lens > 16) & (lens < 256)].sum()),
                                       long=int(lens[lens >= 256].sum()))
    assert cls["listed"] == sum(h["segments"]["short"] for h in cls["per_hop"])
    # odd widths / fused epilogues run on the general-store kernels: no list there, same bits
    xo = xt[:, : d - 3].contiguous()
    plan = HopPlan.from_scipy(hops, dev(), variant=6)
    assert plan.segment_classes(d - 3)["listed"] == 0
    assert np.array_equal(plan.spmm(xo).cpu().numpy(), og.gcn_layer_tree(hops, x[:, : d - 3]))


@pytest.mark.parametrize("kinds", [("sparse", "mixed", "sparse", "sparse", "sparse"),
                                   ("sparse", "mixed", "sparse", "dense", "sparse", "sparse", "mixed", "sparse")])
@pytest.mark.parametrize("d", [64, 100])
def test_five_and_eight_hop_groups(kinds, d):
    """`--adj_nhood` takes any number of hop groups (reference h2gcn/datasets/_dataset.py:559-576); the header admits
    H2GCN_MAX_HOPS = 8.  With 5 and 8 hops, short / medium / long segments side by side: forward, hop subsets, adjoint and the
    accumulating adjoint carry the bits of the canonical tree for the default schedule (0), forced lists (6) and the plain
    wave walk (3).  Above kShortSumHops = 4 selected hops the SUM-mode (adjoint) launch has no row lists and walks the
    tiles (h2gcn_capi.hip, build_class_lists); a 4-hop subset of the same plan gets its lists back.  rows_per_wave = 7 with
    8 hops is the largest (rows_per_wave + 1) * n_hops = 64 row-pointer load one wave holds."""
    from h2gcn_amd import HopPlan

    H = len(kinds)
    rng = np.random.default_rng(1000 * H + d)
    n = 1500
    hops = _mixed_hops(rng, n, kinds)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    w = rng.uniform(-1, 1, (n, H, d)).astype(np.float32)
    xt, wt = torch.from_numpy(x).to(dev()), torch.from_numpy(w).to(dev())
    tree, tree_t = og.gcn_layer_tree(hops, x), og.gcn_layer_grad_tree(hops, w, n)
    assert_close(tree, hops, x)
    want_t = sum(h.T.astype(np.float64) @ w[:, k, :].astype(np.float64) for k, h in enumerate(hops))
    mag_t = sum(abs(h).T.astype(np.float64) @ np.abs(w[:, k, :]).astype(np.float64) for k, h in enumerate(hops))
    assert (np.abs(tree_t - want_t) <= 1e-5 * np.maximum(1.0, mag_t)).all()
    sparse = [k for k, kind in enumerate(kinds) if kind == "sparse"]          # 4 of 5 / 5 of 8 hops: every segment short
    sub4, sub5 = [0, 1, H - 2, H - 1], [0, 1, 2, H - 2, H - 1]
    for variant in (0, 6, 3):
        for rpw in (0, 7):
            plan = HopPlan.from_scipy(hops, dev(), build_transpose=True, variant=variant, rows_per_wave=rpw)
            assert plan.n_hops == H
            y = plan.spmm(xt)
            assert y.shape == (n, H, d) and np.array_equal(y.cpu().numpy(), tree), (variant, rpw)
            dx = plan.spmm_t(wt)
            assert np.array_equal(dx.cpu().numpy(), tree_t), (variant, rpw, "adjoint")
            for sel in (sub4, sub5, sparse[:4], sparse, [H - 1]):
                hs = [hops[k] for k in sel]
                assert np.array_equal(plan.spmm(xt, hops=sel).cpu().numpy(), og.gcn_layer_tree(hs, x)), (variant, rpw, sel)
                ws = wt[:, sel].contiguous()
                assert np.array_equal(plan.spmm_t(ws, hops=sel).cpu().numpy(), og.gcn_layer_grad_tree(hs, w[:, sel], n)), (variant, rpw, sel)
            # accumulate into a strided slot of a wider gradient buffer: exactly old + sum, nothing else touched
            wide = torch.from_numpy(rng.uniform(-1, 1, (n, d + 5)).astype(np.float32)).to(dev())
            before = wide.clone()
            plan.spmm_t(wt, out=wide[:, 2:2 + d], accumulate=True)
            assert torch.equal(wide[:, 2:2 + d], before[:, 2:2 + d] + dx)
            assert torch.equal(wide[:, :2], before[:, :2]) and torch.equal(wide[:, 2 + d:], before[:, 2 + d:])
            if variant == 6 and d % 4 == 0:
                # the forward keeps one list per selected hop at any hop count; the adjoint (one list of the rows that are short
                # in EVERY selected hop) only for selections of at most 4 hops
                assert plan.segment_classes(d)["listed"] > 0
                assert plan.segment_classes(d, hops=sparse[:4], adjoint=True)["listed"] > 0
                if len(sparse) > 4:
                    assert plan.segment_classes(d, hops=sparse, adjoint=True)["listed"] == 0
                assert plan.segment_classes(d, adjoint=True)["listed"] == 0


def test_first_launch_of_a_hop_selection_inside_a_capture_is_refused_with_advice():
    """include/h2gcn_hip.h, "hipGraph capture": the device lists of a hop selection are built by its first launch (allocation
    + synchronous upload); on a capturing stream that launch is refused with a message that names the remedy -- never a
    capture failure deep inside HIP, never a silent fall-back to a slower walk.  After one eager launch the same call captures
    and replays with the eager bits."""
    from h2gcn_amd import HopPlan
    from h2gcn_amd._capi import H2GCNError

    rng = np.random.default_rng(3)
    n, d = 1500, 64
    hops = _mixed_hops(rng, n, ("sparse", "mixed", "dense"))        # "mixed" has segments of 256 and 700 nonzeros: a long list
    plan = HopPlan.from_scipy(hops, dev(), build_transpose=True)
    xt = torch.from_numpy(rng.uniform(-1, 1, (n, d)).astype(np.float32)).to(dev())
    wt = torch.from_numpy(rng.uniform(-1, 1, (n, 1, d)).astype(np.float32)).to(dev())
    y, dx = torch.zeros((n, 1, d), device=dev()), torch.zeros((n, d), device=dev())
    plan.spmm(xt)                                                   # the all-hops selection: prepared at plan creation
    torch.cuda.synchronize()
    errors = {}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for name, fn in (("forward", lambda: plan.spmm(xt, hops=[1], out=y)), ("adjoint", lambda: plan.spmm_t(wt, hops=[1], out=dx))):
            try:
                fn()
            except H2GCNError as e:
                errors[name] = str(e)
    # the forward selection has long segments (a list to build): refused.  The adjoint's A_1^T may have none (the long rows of A_1
    # spread over its columns): then there is nothing to build and the launch is simply captured -- either outcome is legitimate
    assert "forward" in errors and all("captured" in e and "warm-up" in e for e in errors.values()), errors
    want, want_t = plan.spmm(xt, hops=[1]), plan.spmm_t(wt, hops=[1])     # eager once: builds the lists
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        plan.spmm(xt, hops=[1], out=y)
        plan.spmm_t(wt, hops=[1], out=dx)
    y.zero_(); dx.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want) and torch.equal(dx, want_t)
    assert np.array_equal(want.cpu().numpy(), og.gcn_layer_tree([hops[1]], xt.cpu().numpy()))


@pytest.mark.parametrize("n,d", [(300_000, 128), (1_200_000, 64)])
def test_list_driven_launch_at_scale_has_the_bits_of_the_tile_walk(n, d):
    """Mixed segment classes at a size where the list-driven launch uses everything it has -- hundreds of thousands of listed
    entries (16 ... 64 per wave), short- and medium-list workgroups interleaved in groups of 8, a partial last workgroup per
    list, long segments next to them: the WHOLE result (forward, hop subset, adjoint) equals the plain tile walk's bit for bit."""
    from h2gcn_amd import HopPlan, synth

    device = dev()
    cfg = dict(n=n, nnz_per_hop=[6 * n, 40 * n], degrees=[dict(sigma=1.0), dict(sigma=1.3)])
    degs = synth.hop_degrees(cfg, (31, 32))
    csr = [synth.synth_hop_rows(degs[k], n, (31, 32)[k], 0, n, device) for k in range(2)]
    args = ([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
    ref = HopPlan(*args, build_transpose=True, variant=3)
    x = synth.synth_features(d, 33, 0, n, device)
    w = synth.synth_features(2 * d, 34, 0, n, device).view(n, 2, d)
    y_ref, dx_ref = ref.spmm(x), ref.spmm_t(w)
    for variant in (0, 6):
        plan = HopPlan(*args, build_transpose=True, variant=variant)
        cls = plan.segment_classes(d)
        assert cls["listed"] > 0.3 * n and all(h["segments"]["long"] > 0 for h in cls["per_hop"][1:]), cls
        assert torch.equal(plan.spmm(x), y_ref), variant
        assert torch.equal(plan.spmm(x, hops=[0]), ref.spmm(x, hops=[0]))
        assert torch.equal(plan.spmm_t(w), dx_ref), variant     # (the transposed operands have Poisson column counts ~ 6 / 40: the
        #  all-hops adjoint has no all-short rows and stays on the tile walk; the 1-hop one alone is list-driven or in-tile)
        assert torch.equal(plan.spmm_t(w[:, :1].contiguous(), hops=[0]), ref.spmm_t(w[:, :1].contiguous(), hops=[0]))


def test_list_driven_adjoint_at_scale_on_a_symmetric_graph():
    """Real adjacency rings are symmetric, so the transposed operands have the forward's segment classes: a quarter of the rows of
    A_1^T and A_2^T are short in BOTH hops and the SUM-mode launch is list-driven as well (rows whose segments of all selected
    hops are short: one lane group each, partials running on across the hops).  Whole-tensor bit equality with the tile walk."""
    from h2gcn_amd import HopPlan, synth

    device = dev()
    n, d = 200_000, 128
    cfg = dict(n=n, nnz_per_hop=[5 * n, 12 * n], degrees=[dict(sigma=1.0), dict(sigma=1.3)])     # symmetrised: means ~10 and ~24
    degs = synth.hop_degrees(cfg, (41, 42))
    csr = []
    for k in range(2):
        rp, ci, _ = synth.synth_hop_rows(degs[k], n, (41, 42)[k], 0, n, device)
        rows = torch.repeat_interleave(torch.arange(n, device=device), rp[1:] - rp[:-1])
        key = torch.unique(torch.cat([rows * n + ci.long(), ci.long() * n + rows]))        # A + A^T pattern
        r, c = torch.div(key, n, rounding_mode="floor"), (key % n).to(torch.int32)
        cnt = torch.bincount(r, minlength=n)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        rowptr[1:] = torch.cumsum(cnt, 0)
        vals = (1.0 / cnt.to(torch.float32))[r] * (1.0 + (c % 7).to(torch.float32) * 0.125)   # not symmetric in value: A^T != A
        csr.append((rowptr, c.contiguous(), vals.contiguous()))
    args = ([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
    ref = HopPlan(*args, build_transpose=True, variant=3)
    x = synth.synth_features(d, 43, 0, n, device)
    w = synth.synth_features(2 * d, 44, 0, n, device).view(n, 2, d)
    for variant in (0, 6):
        plan = HopPlan(*args, build_transpose=True, variant=variant)
        assert plan.segment_classes(d, adjoint=True)["listed"] > 0.1 * n and plan.segment_classes(d)["listed"] > 0.5 * n
        assert plan.schedule(d, adjoint=True)["segment_walk"].startswith("lane group per segment (binned")
        assert torch.equal(plan.spmm_t(w), ref.spmm_t(w)), variant
        assert torch.equal(plan.spmm(x), ref.spmm(x)), variant
        assert torch.equal(plan.spmm_t(w[:, 1:].contiguous(), hops=[1]), ref.spmm_t(w[:, 1:].contiguous(), hops=[1]))


def test_cora_one_hop_ring_is_served_by_the_binned_list():
    """The reference's own operands are bimodal (exact-1-hop ring: mean 3.9 nonzeros per row, exact-2-hop ring: 31.9): the
    pooled mean (17.9) used to put the WHOLE launch on the wave walk; now A1's segments are listed and served one lane
    group each while A2's longer ones keep the wave walk -- and the result is the stored golden one."""
    from h2gcn_amd import HopPlan

    g = load_planetoid_golden("cora")
    hops = [g["hop1_sym"], g["hop2_sym"]]
    plan = HopPlan.from_scipy(hops, dev(), build_transpose=True)
    for d in (64, 128):
        cls = plan.segment_classes(d)
        a1, a2 = cls["per_hop"]
        assert a1["segments"]["short"] >= 0.97 * g["n"] and a2["segments"]["medium"] > 0.4 * g["n"]
        assert cls["listed"] == a1["segments"]["short"] + a2["segments"]["short"]
        assert cls["walks"] == dict(short="lane group per segment (binned list)", medium="wave per segment",
                                    long="workgroup per segment (4 waves, LDS-staged)")
        x = np.random.default_rng(d).uniform(-1, 1, (g["n"], d)).astype(np.float32)
        y = plan.spmm(torch.from_numpy(x).to(dev())).cpu().numpy()
        assert np.array_equal(y, og.gcn_layer_tree(hops, x))
        assert_close(y, hops, x)


def test_variant_scalar_addressing_matches():
    hops = [rand_csr(500, 500, 0.1, 1), rand_csr(500, 500, 0.2, 2)]
    x = np.random.default_rng(1).uniform(-1, 1, (500, 128)).astype(np.float32)
    y1, _ = run_hip(hops, x, variant=1)
    assert_close(y1, hops, x)


def test_padding_never_touches_x():
    """No nonzero points at column 0, whose feature row is Inf/NaN: results must stay finite -- i.e. padded
    gather slots are predicated off, not multiplied by zero."""
    rng = np.random.default_rng(3)
    n = 400
    m = rand_csr(n, n, 0.08, 6).tolil()
    m[:, 0] = 0
    m = sp.csr_matrix(m)
    m.eliminate_zeros()
    x = rng.uniform(-1, 1, (n, 128)).astype(np.float32)
    x[0, :64] = np.inf
    x[0, 64:] = np.nan
    for d in (128, 100, 64):
        y, _ = run_hip([m, m], x[:, :d].copy())
        assert np.isfinite(y).all()


def test_hop_selection_and_strided_output():
    """GCNLayer(hops={1}) (reference _layers.py:57-59,80-81) and writing into a slice of a wider buffer."""
    from h2gcn_amd import GCNLayer, HopPlan

    hops = [rand_csr(300, 300, 0.05, 1), rand_csr(300, 300, 0.1, 2), rand_csr(300, 300, 0.02, 3)]
    x = np.random.default_rng(1).uniform(-1, 1, (300, 64)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev())
    plan = HopPlan.from_scipy(hops, dev())
    want = og.gcn_layer_f64acc(hops, x)
    y = GCNLayer(hops={1})(plan, xt)
    assert y.shape == (300, 1, 64) and np.abs(y.cpu().numpy()[:, 0] - want[:, 1]).max() <= ATOL
    y = GCNLayer(hops={0, 2, 9})(plan, xt)  # unknown index 9 is ignored like the reference's filter
    assert np.abs(y.cpu().numpy() - want[:, [0, 2]]).max() <= ATOL
    with pytest.raises(ValueError):
        GCNLayer(hops={7})(plan, xt)
    # land the three hops in columns [64, 256) of a [300, 320] concat buffer, strided input
    buf = torch.full((300, 320), -7.0, device=dev())
    xwide = torch.zeros((300, 96), device=dev())
    xwide[:, :64] = xt
    plan.spmm(xwide[:, :64], out=buf[:, 64:256].view(300, 3, 64))
    b = buf.cpu().numpy()
    assert np.abs(b[:, 64:256].reshape(300, 3, 64) - want).max() <= ATOL
    assert (b[:, :64] == -7).all() and (b[:, 256:] == -7).all()


def test_unaligned_operands_and_odd_widths():
    """Views whose base is not 16-byte aligned, odd row strides, d % 4 != 0 (raw feature widths: the reference takes any
    b.shape[1], _layers.py:62-76): gathered IN PLACE by the float4 kernels (16-byte global loads / stores only need dword
    alignment on gfx950; the lane that straddles the end of a row overlaps its neighbour) -- same bits as every other
    schedule (canonical tree), guard columns untouched, with and without scratch, forward, adjoint and fused epilogue."""
    from h2gcn_amd import HopPlan

    hops = [rand_csr(400, 400, 0.05, 1), rand_csr(400, 400, 0.1, 2)]
    hops[0] = sp.csr_matrix(sp.vstack([hops[0][:3], sp.csr_matrix(np.full((1, 400), 0.01, dtype=np.float32)), hops[0][4:]]))
    for d in (128, 133, 1433 // 7, 65, 67, 3):
        plan = HopPlan.from_scipy(hops, dev(), build_transpose=True)
        assert plan.schedule(d, ld_src=d + 3)["slice_cols"] in ((64, 128, 256) if d >= 4 else (d,))
        x = np.random.default_rng(d).uniform(-1, 1, (400, d)).astype(np.float32)
        w = np.random.default_rng(d + 1).uniform(-1, 1, (400, 2, d)).astype(np.float32)
        tree = og.gcn_layer_tree(hops, x)
        tree_t = og.gcn_layer_grad_tree(hops, w, 400)
        assert_close(tree, hops, x)
        xbuf = torch.zeros((400, d + 3), device=dev())
        xbuf[:, 1:1 + d] = torch.from_numpy(x).to(dev())
        xv = xbuf[:, 1:1 + d]
        wbuf = torch.zeros((400, 2 * d + 5), device=dev())
        wv = wbuf[:, 2:2 + 2 * d].view(400, 2, d)
        wv.copy_(torch.from_numpy(w))
        for use_ws in (True, False):
            plan.use_workspace = use_ws
            ybuf = torch.full((400, 2 * d + 3), 5.0, device=dev())
            y = plan.spmm(xv, out=ybuf[:, 2:2 + 2 * d].view(400, 2, d))
            assert np.array_equal(y.cpu().numpy(), tree), (d, use_ws)
            assert bool((ybuf[:, :2] == 5.0).all()) and bool((ybuf[:, 2 + 2 * d:] == 5.0).all())
            assert np.array_equal(plan.spmm_t(wv).cpu().numpy(), tree_t), (d, use_ws)
            b = torch.from_numpy(np.random.default_rng(5).uniform(-0.5, 0.5, d).astype(np.float32)).to(dev())
            yb = plan.spmm(xv, bias=b, relu=True)
            assert torch.equal(yb, torch.relu(torch.from_numpy(tree).to(dev()) + b)), (d, use_ws)
        # the LAST row of X ends exactly at the end of its allocation: the tail lane must not read past it
        xt = torch.from_numpy(x).to(dev())
        assert np.array_equal(plan.spmm(xt).cpu().numpy(), tree)


def test_odd_width_scratch_copy_of_wide_unaligned_rows():
    """d = 203 on an operand far beyond the caches: rows of 812 B are wide and not cache-line aligned, so a launch with
    scratch gathers from the zero-padded slice-major copy (element-wise repack) -- bit-identical to the in-place launch."""
    import ctypes as C

    from h2gcn_amd import HopPlan, _capi, synth

    n, d, device = 700_000, 203, dev()
    degs = [synth.synth_degrees(n, 12_500_000, s, n) for s in (15, 16)]
    csr = [synth.synth_hop_rows(degs[k], n, (15, 16)[k], 0, n, device) for k in range(2)]
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n, build_transpose=True)
    x = synth.synth_features(d, 17, 0, n, device)
    assert _capi.lib().h2gcn_spmm_workspace_bytes(plan._handle, 0, 0, C.c_void_p(x.data_ptr()), d, 0, d) == n * 4 * 64 * 4
    assert plan.schedule(d)["scratch_copy"] and not plan.schedule(d, adjoint=True)["scratch_copy"]
    y_ws = plan.spmm(x)
    plan.use_workspace = False
    assert torch.equal(y_ws, plan.spmm(x))
    g = synth.synth_features(2 * 303, 9, 0, n, device).view(n, 2, 303)      # adjoint: copied beyond 256 columns
    plan.use_workspace = True
    assert plan.schedule(303, adjoint=True)["scratch_copy"]
    dx_ws = plan.spmm_t(g)
    plan.use_workspace = False
    assert torch.equal(dx_ws, plan.spmm_t(g))


def test_c_abi_from_plain_c():
    """build/capi_demo (tools/capi_demo.c, gcc): plan create -> forward -> adjoint -> destroy with hipMalloc'd
    buffers, no Python in the loop."""
    import subprocess
    from pathlib import Path

    exe = Path(__file__).resolve().parents[1] / "build" / "capi_demo"
    if not exe.exists():
        import __graft_entry__ as ge
        exe = ge.build_capi_demo()
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout and "hop_mask" in r.stdout


# ----------------------------------------------------------------------------- adjoint / autograd
@pytest.mark.parametrize("d", [64, 128, 100])
def test_adjoint_matches_oracle(d):
    from h2gcn_amd import HopPlan

    hops = [rand_csr(700, 500, 0.05, 1, empty_frac=0.1), rand_csr(700, 500, 0.1, 2)]
    dy = np.random.default_rng(2).uniform(-1, 1, (700, 2, d)).astype(np.float32)
    plan = HopPlan.from_scipy(hops, dev(), build_transpose=True, long_row_threshold=16)
    dx = plan.spmm_t(torch.from_numpy(dy).to(dev())).cpu().numpy()
    want = og.gcn_layer_grad_c(hops, dy, 500)
    assert dx.shape == (500, d)
    assert np.abs(dx - want).max() <= 2e-5
    dx1 = plan.spmm_t(torch.from_numpy(dy[:, 1:2].copy()).to(dev()), hops=[1]).cpu().numpy()
    assert np.abs(dx1 - og.gcn_layer_grad_c(hops[1:], dy[:, 1:2].copy(), 500)).max() <= 2e-5


@pytest.mark.parametrize("d", [64, 128, 67, 3, 200])
@pytest.mark.parametrize("thr", [16, 100000])
def test_adjoint_accumulates_into_an_existing_gradient(d, thr):
    """H2GCN_LAUNCH_ACCUMULATE: dX += A^T dY lands on top of what dX holds -- each element `old + sum` with `sum` the bits of
    the plain launch -- through strided outputs (a column slot of a wider gradient buffer), long and short rows, odd widths."""
    from h2gcn_amd import HopPlan

    hops = [rand_csr(700, 500, 0.05, 1, empty_frac=0.1), rand_csr(700, 500, 0.1, 2)]
    rng = np.random.default_rng(5)
    dy = torch.from_numpy(rng.uniform(-1, 1, (700, 2, d)).astype(np.float32)).to(dev())
    plan = HopPlan.from_scipy(hops, dev(), build_transpose=True, long_row_threshold=thr)
    plain = plan.spmm_t(dy)
    wide = torch.from_numpy(rng.uniform(-1, 1, (500, d + 9)).astype(np.float32)).to(dev())
    before = wide.clone()
    slot = wide[:, 5:5 + d]
    got = plan.spmm_t(dy, out=slot, accumulate=True)
    assert got.data_ptr() == slot.data_ptr()
    assert torch.equal(slot, before[:, 5:5 + d] + plain)                                # bit for bit
    assert torch.equal(wide[:, :5], before[:, :5]) and torch.equal(wide[:, 5 + d:], before[:, 5 + d:])   # neighbours untouched
    # out= without accumulate overwrites
    plan.spmm_t(dy, out=slot)
    assert torch.equal(slot, plain)
    with pytest.raises(ValueError):
        plan.spmm_t(dy, accumulate=True)
    with pytest.raises(ValueError):
        plan.spmm_t(dy, out=wide[:, :d + 1], accumulate=True)


def test_fused_propagation_backward_in_place_is_bitwise_the_out_of_place_one(monkeypatch):
    """The concat-free propagation's backward adds each round's adjoint into the slot of the incoming gradient (accumulate
    flag) -- same bits as a fresh tensor per round plus a `+=` pass, and the incoming gradient's other columns are left alone."""
    from h2gcn_amd import HopPlan, layers

    n = 900
    hops = [rand_csr(n, n, 0.04, 3, empty_frac=0.05), rand_csr(n, n, 0.08, 4)]
    plan = HopPlan.from_scipy(hops, dev(), build_transpose=True, long_row_threshold=64)
    assert plan.schedule(64, ld_src=448, adjoint=True)["segment_walk"] == "wave per segment"
    rng = np.random.default_rng(9)
    r0 = torch.from_numpy(rng.uniform(-1, 1, (n, 64)).astype(np.float32)).to(dev())
    gout = torch.from_numpy(rng.uniform(-1, 1, (n, 448)).astype(np.float32)).to(dev())
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("H2GCN_BACKWARD_IN_PLACE", mode)
        x = r0.clone().requires_grad_(True)
        buf = layers.fused_propagation(plan, x, 2)
        g = gout.clone()
        buf.backward(g)
        grads[mode] = x.grad.clone()
        if mode == "1":
            assert torch.equal(g[:, :256], gout[:, :256])      # d r_2 is only read
    assert torch.equal(grads["1"], grads["0"])


def test_device_and_host_transposition_agree():
    """The device radix-sort transposition and the host counting sort build the same canonical A^T: the adjoint
    launch gives identical bits on either."""
    from h2gcn_amd import HopPlan

    hops = [rand_csr(1500, 900, 0.03, 1, empty_frac=0.1), rand_csr(1500, 900, 0.08, 2), sp.csr_matrix((1500, 900), dtype=np.float32)]
    dy = torch.from_numpy(np.random.default_rng(2).uniform(-1, 1, (1500, 3, 64)).astype(np.float32)).to(dev())
    a = HopPlan.from_scipy(hops, dev(), build_transpose=True).spmm_t(dy)
    b = HopPlan.from_scipy(hops, dev(), build_transpose=True, host_transpose=True).spmm_t(dy)
    assert torch.equal(a, b)
    assert np.abs(a.cpu().numpy() - og.gcn_layer_grad_c(hops, dy.cpu().numpy(), 900)).max() <= 2e-5


def test_autograd_through_gcn_layer():
    from h2gcn_amd import GCNLayer, HopPlan

    g = load_planetoid_golden("cora")
    hops = [g["hop1_sym"], g["hop2_sym"]]
    plan = HopPlan.from_scipy(hops, dev(), build_transpose=True)
    x = torch.randn(g["n"], 64, device=dev(), requires_grad=True)
    w = torch.randn(g["n"], 2, 64, device=dev())
    (GCNLayer()(plan, x) * w).sum().backward()
    want = og.gcn_layer_grad_c(hops, w.cpu().numpy(), g["n"])
    assert np.abs(x.grad.cpu().numpy() - want).max() <= 2e-5
    # A is symmetric under SYM normalisation on an undirected graph: adjoint == forward applied per hop
    y = plan.spmm(w[:, 0].contiguous())[:, 0] + plan.spmm(w[:, 1].contiguous())[:, 1]
    assert (y - x.grad).abs().max().item() <= 2e-5
    plan_nt = HopPlan.from_scipy(hops, dev())
    x2 = torch.randn(g["n"], 64, device=dev(), requires_grad=True)
    with pytest.raises(ValueError, match="build_transpose"):
        GCNLayer()(plan_nt, x2).sum().backward()


# ----------------------------------------------------------------------------- determinism / partitioning
def test_bitwise_repeatable_and_row_partition_invariant():
    """a row-partitioned run must equal the single-GPU run bit-for-bit."""
    from h2gcn_amd import HopPlan

    hops = [rand_csr(2001, 2001, 0.02, 1, empty_frac=0.05), rand_csr(2001, 2001, 0.05, 2)]
    hops[0] = sp.csr_matrix(sp.vstack([hops[0][:5], sp.csr_matrix(np.ones((1, 2001), dtype=np.float32) / 2001), hops[0][6:]]))
    x = np.random.default_rng(1).uniform(-1, 1, (2001, 128)).astype(np.float32)
    full, _ = run_hip(hops, x, long_row_threshold=256)
    again, _ = run_hip(hops, x, long_row_threshold=256)
    assert np.array_equal(full, again)
    for P in (2, 3, 8):
        bounds = np.linspace(0, 2001, P + 1).astype(int)
        parts = []
        for p in range(P):
            shard = [h[bounds[p]:bounds[p + 1]] for h in hops]
            yp, _ = run_hip(shard, x, long_row_threshold=256)
            parts.append(yp)
        assert np.array_equal(np.concatenate(parts, 0), full)


def test_soak_repeated_launches_are_bitwise_stable():
    """200 back-to-back launches (forward and adjoint, many long segments, concurrent streams) give the same bits
    every time: no race in the LDS-staged long path, no dependence on scheduling."""
    from h2gcn_amd import HopPlan, synth

    device = dev()
    n = 60_000
    degs = [synth.synth_degrees(n, 30 * n, s, n) for s in (5, 6)]
    csr = [synth.synth_hop_rows(degs[k], n, (5, 6)[k], 0, n, device) for k in range(2)]
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n, build_transpose=True,
                   long_row_threshold=48)
    assert plan.info(0)["n_long_segments"] > 1000
    x = synth.synth_features(128, 7, 0, n, device)
    w = synth.synth_features(256, 8, 0, n, device).view(n, 2, 128)
    y0, g0 = plan.spmm(x).clone(), plan.spmm_t(w).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())  # w / x were produced on the main stream
    for i in range(100):
        y = plan.spmm(x)
        with torch.cuda.stream(side):   # same plan, another stream, at the same time
            g = plan.spmm_t(w)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(y, y0) and torch.equal(g, g0), i


# ----------------------------------------------------------------------------- error behaviour
def test_errors_are_python_exceptions_before_or_at_the_c_call():
    from h2gcn_amd import HopPlan, _capi

    d = dev()
    m = rand_csr(10, 10, 0.3, 1)
    plan = HopPlan.from_scipy([m], d)
    with pytest.raises(ValueError):
        plan.spmm(torch.zeros(11, 4, device=d))           # wrong row count
    with pytest.raises(ValueError):
        plan.spmm(torch.zeros(10, 4, device=d, dtype=torch.float64))
    with pytest.raises(ValueError):
        plan.spmm(torch.zeros(10, 4))                      # CPU tensor
    with pytest.raises(ValueError):
        plan.spmm_t(torch.zeros(10, 1, 4, device=d))       # no transpose built
    rp = torch.tensor([0, 2, 1], dtype=torch.int64, device=d)
    ci = torch.tensor([0, 1], dtype=torch.int32, device=d)
    va = torch.ones(2, device=d)
    with pytest.raises(_capi.H2GCNError) as e:
        HopPlan([rp], [ci], [va], 2)
    assert e.value.status == _capi.ERR_BAD_INDEX
    rp = torch.tensor([0, 1, 2], dtype=torch.int64, device=d)
    ci = torch.tensor([0, 5], dtype=torch.int32, device=d)
    with pytest.raises(_capi.H2GCNError) as e:
        HopPlan([rp], [ci], [va], 2)                       # column 5 out of range (TF: InvalidArgument)
    assert e.value.status == _capi.ERR_BAD_INDEX


# ----------------------------------------------------------------------------- BASELINE shapes, size-independent properties
@pytest.mark.parametrize("shape", ["arxiv", "products", "lowdeg", "h2gcn_like", "products_tail"])
def test_baseline_shapes_properties(shape):
    """configs[2] / configs[3] at FULL size (+ the mean-degree-4 stress shape, which runs the in-tile short-row kernels, and the two
    mixed-class shapes, which run list-driven): (1) row-stochastic hops map the all-ones features to ones on every
    non-empty row and zeros on empty rows; (2) linearity; (3) sampled rows against the fp64 oracle, the CPU
    regenerating those rows of the operands independently (counter-based generator)."""
    from h2gcn_amd import HopPlan, synth

    cfg = synth.SHAPES[shape]
    n, d = cfg["n"], cfg["d"]
    device = dev()
    degs = synth.hop_degrees(cfg)
    csr = [synth.synth_hop_rows(degs[k], n, (synth.SEED_A1, synth.SEED_A2)[k], 0, n, device) for k in range(2)]
    plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n)
    nnz = plan.nnz
    want_nnz = cfg["nnz_per_hop"] if isinstance(cfg["nnz_per_hop"], (list, tuple)) else [cfg["nnz_per_hop"]] * 2
    assert all(abs(z - t) < 0.01 * t for z, t in zip(nnz, want_nnz))
    walk = plan.schedule(d)["segment_walk"]
    assert walk.startswith({"lowdeg": "lane group per segment (short rows)", "arxiv": "lane group per segment (short rows)",
                            "h2gcn_like": "lane group per segment (binned", "products_tail": "lane group per segment (binned"}.get(shape, "wave per segment")), walk
    x = synth.synth_features(d, synth.SEED_X, 0, n, device)
    y = plan.spmm(x)
    # (0) every element, through the order-independent fingerprint: the checksum of Y is the CPU ORACLE's for this shape (computed by
    #     `python -m oracle.fullsize`, profiles/r06_oracle_checksums_of_the_bench_shapes.txt; arxiv / products are also compared
    #     element by element in tests/test_fullsize_parity_gpu.py)
    import sys
    from pathlib import Path as _P
    sys.path.insert(0, str(_P(__file__).resolve().parents[1]))
    import bench
    assert int(y.view(torch.int32).to(torch.int64).sum().item()) == bench.N1_CHECKSUMS[(shape, d)]
    # (1) row-stochastic
    ones = torch.ones((n, d), device=device)
    y1 = plan.spmm(ones)
    for k in range(2):
        nonempty = (csr[k][0][1:] - csr[k][0][:-1]) > 0
        assert (y1[nonempty, k] - 1).abs().max().item() <= 1e-4  # sum of deg copies of fl(1/deg), deg up to 17k
        assert not y1[~nonempty, k].any().item()
    # (2) linearity: A(2x + 1) == 2 A x + A 1
    y2 = plan.spmm(2 * x + ones)
    assert (y2 - (2 * y + y1)).abs().max().item() <= 1e-4
    del y1, y2, ones
    # (2b) adjoint identity <A x, w> == <x, A^T w> with the device-built transposed operands
    plan_t = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n, build_transpose=True)
    w = synth.synth_features(2 * d, 77, 0, n, device).view(n, 2, d)
    dx = plan_t.spmm_t(w)
    assert int(dx.view(torch.int32).to(torch.int64).sum().item()) == bench.N1_ADJOINT_CHECKSUMS[(shape, d)]     # the oracle's dX, likewise
    lhs = (y.double() * w.double()).sum().item()
    rhs = (x.double() * dx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs), (y.double().abs() * w.double().abs()).sum().item() * 1e-3)
    del plan_t, w, dx
    # (3) sampled rows vs the oracle on independently regenerated operands
    rng = np.random.default_rng(0)
    blocks = [0, n - 8] + list(rng.integers(0, n - 8, 6))
    longest = int(np.argmax(degs[0]))
    blocks.append(min(longest, n - 8))
    for r0 in blocks:
        r0 = int(r0)
        parts = []
        for k, seed in enumerate((synth.SEED_A1, synth.SEED_A2)):
            rp, ci, va = synth.synth_hop_rows_np(degs[k], n, seed, r0, r0 + 8)
            lo, hi = int(csr[k][0][r0]), int(csr[k][0][r0 + 8])
            assert np.array_equal(ci, csr[k][1][lo:hi].cpu().numpy())  # GPU-built CSR == CPU-built CSR
            parts.append((rp, ci, va))
        cols = np.unique(np.concatenate([p[1] for p in parts]))
        xs = x[torch.from_numpy(cols.astype(np.int64)).to(device)].cpu().numpy()
        remap = {c: i for i, c in enumerate(cols)}
        local = [(p[0], np.array([remap[c] for c in p[1]], dtype=np.int32), p[2]) for p in parts]
        want = og.rows_subset(local, xs, list(range(8)))
        got = y[r0:r0 + 8].cpu().numpy()
        assert np.abs(got - want).max() <= ATOL


# ----------------------------------------------------------------------------- maximum sizes: 64-bit offsets
def test_x_beyond_4gib_uses_64bit_gather_offsets():
    """X of 4.6 GB: byte offsets into the gather source exceed 32 bits, so the launch takes the 64-bit-offset
    kernels (the reference needs a column split at nnz*d > 2^31 on GPU, `_layers.py:65-74`; here nothing splits)."""
    from h2gcn_amd import HopPlan

    device = dev()
    n_cols, d, n_rows = 9_000_000, 128, 4096
    assert n_cols * d * 4 > 2 ** 32
    base = torch.arange(n_cols, device=device, dtype=torch.float32).remainder_(1000.0).mul_(1e-3)
    x = (base[:, None] + torch.arange(d, device=device, dtype=torch.float32)[None, :] * 1e-2).contiguous()
    rng = np.random.default_rng(4)
    deg = rng.integers(0, 40, n_rows)
    deg[7] = 700  # one long segment too
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    cols = np.concatenate([np.sort(rng.choice(n_cols, k, replace=False)) for k in deg]).astype(np.int32)
    cols[-1] = n_cols - 1 if deg[-1] > 0 else cols[-1]  # touch the very last row of X
    vals = rng.uniform(-1, 1, len(cols)).astype(np.float32)
    m = sp.csr_matrix((vals, cols, rowptr), shape=(n_rows, n_cols))
    m.sort_indices()
    plan = HopPlan.from_scipy([m, m[::-1]], device)
    y = plan.spmm(x).cpu().numpy()
    # oracle on the referenced rows of X only
    used = np.unique(m.indices)
    xs = x[torch.from_numpy(used.astype(np.int64)).to(device)].cpu().numpy()
    remap = sp.csr_matrix((m.data, np.searchsorted(used, m.indices), m.indptr), shape=(n_rows, len(used)))
    want = og.gcn_layer_f64acc([remap, remap[::-1]], xs)
    mag = og.gcn_layer_f64acc([abs(remap), abs(remap[::-1])], np.abs(xs))
    assert (np.abs(y - want) <= ATOL * np.maximum(1.0, mag)).all()


def test_more_than_2_31_nonzeros():
    """2.3e9 nonzeros in one hop matrix: row pointers beyond int32 -- the size the reference's GPU path cannot
    index.  Operands are built by formula on the device (18 GB); checks: row-stochastic property on every row,
    and sampled rows (first, last, across the 2^31 boundary) against the fp64 oracle."""
    from h2gcn_amd import HopPlan

    device = dev()
    n_rows, deg, n_cols, d = 36_000, 64_000, 100_000, 32
    nnz = n_rows * deg
    assert nnz > 2 ** 31
    rowptr = torch.arange(n_rows + 1, device=device, dtype=torch.int64) * deg
    # row i holds columns (i * 7 + t * stride_i) mod n_cols for t < deg, made ascending by construction:
    # start_i + t  (a contiguous window, wrapped rows avoided by clamping the start)
    start = (torch.arange(n_rows, device=device, dtype=torch.int64) * 7919) % (n_cols - deg)
    colidx = torch.empty(nnz, dtype=torch.int32, device=device)
    step = 4000
    ar = torch.arange(deg, device=device, dtype=torch.int32)
    for r0 in range(0, n_rows, step):
        r1 = min(r0 + step, n_rows)
        colidx[r0 * deg:r1 * deg] = (start[r0:r1, None].to(torch.int32) + ar[None, :]).reshape(-1)
    vals = torch.full((nnz,), 1.0 / deg, dtype=torch.float32, device=device)
    plan = HopPlan([rowptr], [colidx], [vals], n_cols)
    assert plan.info(0)["nnz"] == nnz and plan.info(0)["n_long_segments"] == n_rows
    ones = torch.ones((n_cols, d), device=device)
    y1 = plan.spmm(ones)
    assert (y1 - 1).abs().max().item() <= 1e-4
    x = torch.rand((n_cols, d), device=device) * 2 - 1
    y = plan.spmm(x)[:, 0]
    csum = torch.cat([torch.zeros((1, d), dtype=torch.float64, device=device), x.double().cumsum(0)])
    boundary = (2 ** 31) // deg
    for i in (0, 1, boundary - 1, boundary, boundary + 1, n_rows - 1):
        s = int(start[i])
        want = (csum[s + deg] - csum[s]) / deg   # mean of a contiguous window of X rows, in fp64
        assert (y[i].double() - want).abs().max().item() <= ATOL
