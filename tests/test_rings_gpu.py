"""GPU suite: exact-k-hop ring construction and normalisation by the HIP kernels of csrc/rings.hip (through the C
ABI) against the host builder / the fixtures produced by the reference's own `nhoodSplit` + `normalize`
(tests/golden/*_operands.npz).  Integer/bit work: everything is compared bit for bit."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_planetoid_golden, load_syn_products_golden
from h2gcn_amd import operands as po

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _same_operands(rps, cis, vas, host):
    assert len(host) == len(rps)
    for k, h in enumerate(host):
        h = sp.csr_matrix(h)
        h.sort_indices()
        assert np.array_equal(rps[k].cpu().numpy(), h.indptr), k
        assert np.array_equal(cis[k].cpu().numpy(), h.indices), k
        assert np.array_equal(vas[k].cpu().numpy(), h.data.astype(np.float32)), k


@pytest.mark.parametrize("name", ["cora", "citeseer"])
def test_planetoid_rings_equal_reference_fixtures(name):
    g = load_planetoid_golden(name)
    adj = po.remove_self_loops(g["adj_raw"])
    for norm in (po.SYM_NORMALIZED, po.RW_NORMALIZED, po.ORDINARY):
        for nh in (("1", "2"), ("0,1", "2"), ("2",), ("0",), ("0,1,2",)):
            rps, cis, vas, n = po.build_adj_norm_hops_device(adj, nh, norm, DEV)
            assert n == g["n"]
            _same_operands(rps, cis, vas, po.build_adj_norm_hops(adj, nh, norm))
    # directly against what the reference's own code produced (fixtures hold its fp32-cast outputs)
    rps, cis, vas, n = po.build_adj_norm_hops_device(adj, ("1", "2"), po.SYM_NORMALIZED, DEV)
    for k, key in enumerate(("hop1_sym", "hop2_sym")):
        want = g[key]
        assert np.array_equal(cis[k].cpu().numpy(), want.indices) and np.array_equal(vas[k].cpu().numpy(), want.data)
    rps, cis, vas, n = po.build_adj_norm_hops_device(adj, ("0,1",), po.SYM_NORMALIZED, DEV)
    assert np.array_equal(vas[0].cpu().numpy(), g["hop01_sym"].data) and np.array_equal(cis[0].cpu().numpy(), g["hop01_sym"].indices)
    nnz = [int(r[1].numel()) for r in po.exact_hop_rings_device(torch.from_numpy(adj.indptr.astype(np.int64)).to(DEV),
                                                                torch.from_numpy(adj.indices.astype(np.int32)).to(DEV), n, 2)]
    assert nnz == list(g["split_nnz"])


def test_saturating_reachability_and_self_loops():
    path = sp.csr_matrix(np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]], dtype=np.float32))
    with pytest.raises(ValueError):
        po.build_adj_norm_hops_device(path, ("3",), device=DEV)
    rp = torch.from_numpy(path.indptr.astype(np.int64)).to(DEV)
    ci = torch.from_numpy(path.indices.astype(np.int32)).to(DEV)
    rings = po.exact_hop_rings_device(rp, ci, 3, 5)
    assert [r[1].cpu().tolist() for r in rings] == [[0, 1, 2], [1, 0, 2, 1], [2, 0]]        # like the reference: list ends early
    # self loops in A do not leak into ring 1 (bin(A + I) - I)
    loops = sp.csr_matrix(np.array([[1, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.float32))
    r1 = po.ring_set_device(3, torch.device(DEV), add=[(torch.from_numpy(loops.indptr.astype(np.int64)).to(DEV),
                                                      torch.from_numpy(loops.indices.astype(np.int32)).to(DEV))], sub_diag=True)
    assert r1[1].cpu().tolist() == [1, 0, 2, 1] and r1[0].cpu().tolist() == [0, 1, 3, 4]


@pytest.mark.parametrize("n,deg,hops", [(1, 0, 2), (64, 3, 3), (5000, 4, 3), (33000, 2.5, 4), (20000, 30, 2), (200000, 3, 2)])
def test_random_graphs_equal_host_spgemm(n, deg, hops):
    rng = np.random.default_rng(n)
    m = int(n * deg / 2)
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    a = sp.csr_matrix((np.ones(2 * m, dtype=np.float32), (np.r_[r, c], np.r_[c, r])), shape=(n, n))
    a = po.remove_self_loops(a)
    a.data[:] = 1
    host = po.exact_hop_rings(a, hops)
    a.sort_indices()
    rings = po.exact_hop_rings_device(torch.from_numpy(a.indptr.astype(np.int64)).to(DEV),
                                      torch.from_numpy(a.indices.astype(np.int32)).to(DEV), n, hops)
    assert len(rings) == len(host)
    for k, h in enumerate(host):
        h = sp.csr_matrix(h)
        h.sort_indices()
        assert np.array_equal(rings[k][0].cpu().numpy(), h.indptr) and np.array_equal(rings[k][1].cpu().numpy(), h.indices), k


def test_wide_graph_with_hubs_mixes_sorted_and_bitmap_rows():
    """n beyond the LDS bitmap: rows with <= 1024 candidates are served by the sorted-candidate kernel (one wave per
    row), hub rows and their neighbours by the bitmap kernel with global level-0 slabs -- in the same result."""
    n = 150_000
    rng = np.random.default_rng(8)
    m = int(1.5 * n)
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    hub_nb = rng.choice(n, 5000, replace=False)
    r = np.r_[r, np.full(5000, 77), np.full(300, 4242)]
    c = np.r_[c, hub_nb, rng.choice(n, 300, replace=False)]
    a = sp.csr_matrix((np.ones(2 * len(r), dtype=np.float32), (np.r_[r, c], np.r_[c, r])), shape=(n, n))
    a = po.remove_self_loops(a)
    a.data[:] = 1
    a.sort_indices()
    host = po.exact_hop_rings(a, 2)
    rings = po.exact_hop_rings_device(torch.from_numpy(a.indptr.astype(np.int64)).to(DEV),
                                      torch.from_numpy(a.indices.astype(np.int32)).to(DEV), n, 2)
    for k in (1, 2):
        h = sp.csr_matrix(host[k])
        h.sort_indices()
        assert np.array_equal(rings[k][0].cpu().numpy(), h.indptr) and np.array_equal(rings[k][1].cpu().numpy(), h.indices), k
    merged = po.ring_set_device(n, torch.device(DEV), add=[rings[1], rings[2]], add_diag=True)
    want = sp.csr_matrix(host[0] + host[1] + host[2])
    want.sort_indices()
    assert np.array_equal(merged[1].cpu().numpy(), want.indices)


def test_syn_products_two_hop_ring():
    a, _, _ = load_syn_products_golden()
    adj = po.remove_self_loops(a)
    rps, cis, vas, n = po.build_adj_norm_hops_device(adj, ["1", "2"], "sym", DEV)
    _same_operands(rps, cis, vas, po.build_adj_norm_hops(adj, ["1", "2"], "sym"))
    assert cis[1].numel() > 2_000_000        # the 2.8M-nonzero ring of SURVEY.md config 2


def test_more_than_2_20_columns_uses_the_global_bitmap():
    """n > 1 Mi columns: level 0 of the bitmap lives in per-workgroup global slabs (level 1 stays in LDS)."""
    n = (1 << 20) + 12345
    rng = np.random.default_rng(3)
    m = 2 * n
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    a = sp.csr_matrix((np.ones(2 * m, dtype=np.float32), (np.r_[r, c], np.r_[c, r])), shape=(n, n))
    a = po.remove_self_loops(a)
    a.data[:] = 1
    a.sort_indices()
    host = po.exact_hop_rings(a, 2)
    rings = po.exact_hop_rings_device(torch.from_numpy(a.indptr.astype(np.int64)).to(DEV),
                                      torch.from_numpy(a.indices.astype(np.int32)).to(DEV), n, 2)
    for k in (1, 2):
        h = sp.csr_matrix(host[k])
        h.sort_indices()
        assert np.array_equal(rings[k][0].cpu().numpy(), h.indptr) and np.array_equal(rings[k][1].cpu().numpy(), h.indices)
    vals = po.normalize_pattern_device(rings[2], n, po.SYM_NORMALIZED).cpu().numpy()
    want = sp.csr_matrix(po.normalize_hop(host[2], po.SYM_NORMALIZED))
    want.sort_indices()
    assert np.array_equal(vals, want.data.astype(np.float32))


def test_set_algebra_entry_points_directly():
    """h2gcn_ring_count / _fill as a general row-set algebra: unions, differences, the diagonal, empty operands,
    n = 0, and the scratch-size check -- against Python sets."""
    import ctypes as C

    from h2gcn_amd import _capi

    dev = torch.device(DEV)
    rng = np.random.default_rng(4)
    n = 257

    def rand_pattern(density):
        m = sp.random(n, n, density, format="csr", random_state=int(rng.integers(1 << 30)), dtype=np.float32)
        m.sort_indices()
        return m, (torch.from_numpy(m.indptr.astype(np.int64)).to(dev), torch.from_numpy(m.indices.astype(np.int32)).to(dev))

    (ma, a), (mf, f), (mb, b), (mc, c) = rand_pattern(0.03), rand_pattern(0.02), rand_pattern(0.05), rand_pattern(0.04)

    def rows(m):
        return [set(m.indices[m.indptr[i]:m.indptr[i + 1]].tolist()) for i in range(n)]

    ra, rf, rb, rc = rows(ma), rows(mf), rows(mb), rows(mc)

    def check(got, want):
        rp, ci = got[0].cpu().numpy(), got[1].cpu().numpy()
        for i in range(n):
            assert ci[rp[i]:rp[i + 1]].tolist() == sorted(want[i]), i

    expand = [set().union(*[ra[j] for j in rf[i]]) if rf[i] else set() for i in range(n)]
    check(po.ring_set_device(n, dev, a=a, frontier=f), expand)
    check(po.ring_set_device(n, dev, a=a, frontier=f, add=[b], add_diag=True, sub=[c]),
          [(expand[i] | rb[i] | {i}) - rc[i] for i in range(n)])
    check(po.ring_set_device(n, dev, add=[b, c]), [rb[i] | rc[i] for i in range(n)])
    check(po.ring_set_device(n, dev, add=[b], sub=[b]), [set() for _ in range(n)])
    check(po.ring_set_device(n, dev, add_diag=True, sub=[c]), [{i} - rc[i] for i in range(n)])
    check(po.ring_set_device(n, dev, add=[b], sub=[c], sub_diag=True), [rb[i] - rc[i] - {i} for i in range(n)])
    empty = po.ring_set_device(0, dev)
    assert empty[0].tolist() == [0] and empty[1].numel() == 0
    # too little scratch is an error, not a crash
    L = _capi.lib()
    rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nnz = C.c_int64()
    tiny = torch.empty(8, dtype=torch.uint8, device=dev)
    st = L.h2gcn_ring_count(n, None, None, None, None, 0, None, None, 1, 0, None, None, 0, C.c_void_p(rp.data_ptr()), C.byref(nnz),
                            C.c_void_p(tiny.data_ptr()), 8, None)
    assert st == _capi.ERR_INVALID_ARGUMENT and b"scratch" in L.h2gcn_last_error()
    # three exact rings of a path graph with a pendant triangle, through the public builder
    edges = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 3)]
    m = sp.lil_matrix((6, 6), dtype=np.float32)
    for u, v in edges:
        m[u, v] = m[v, u] = 1
    m = sp.csr_matrix(m)
    m.sort_indices()
    host = po.exact_hop_rings(m, 3)
    rings = po.exact_hop_rings_device(torch.from_numpy(m.indptr.astype(np.int64)).to(dev), torch.from_numpy(m.indices.astype(np.int32)).to(dev), 6, 3)
    for k in range(4):
        h = sp.csr_matrix(host[k])
        h.sort_indices()
        assert rings[k][1].cpu().tolist() == h.indices.tolist(), k


# ----------------------------------------------------------------------------- row windows (row-partitioned build)
def _sharded_build(adj, nh, norm, world, balance):
    """Every "rank" of a world-way partition in one process: phase A per rank, the all-gather replaced by a closure."""
    from h2gcn_amd.datasets._dataset import sharded_adj_hops
    from h2gcn_amd.partition import RowPartition

    rp, ci, n = po.upload_pattern(adj, DEV)
    max_hop = max(max(g) for g in po.parse_adj_nhood(nh))
    eq = RowPartition.equal(n, world)
    lens = [po.ring_row_lengths_window(rp, ci, n, max_hop, eq.rows(q)) for q in range(world)]
    return [sharded_adj_hops(adj, nh, norm, DEV, q, world, gather=lambda t: lens, balance=balance) for q in range(world)]


@pytest.mark.parametrize("name", ["cora", "citeseer"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_window_rings_equal_the_rows_of_the_full_build(name, world):
    """Each rank of a row partition builds ONLY its rows of every ring (from the whole adjacency pattern and its own rows
    of the lower rings) and normalises them with the all-gathered row lengths: bit-identical to the rows of the
    single-GPU build -- hence to the reference's own outputs (the fixtures) -- for equal and nnz-balanced blocks, SYM /
    RW, merged groups and 3 hops.  citeseer: isolated nodes, empty rows, inf -> 0."""
    g = load_planetoid_golden(name)
    adj = po.remove_self_loops(g["adj_raw"])
    n = g["n"]
    for norm, nh in ((po.SYM_NORMALIZED, ("1", "2")), (po.RW_NORMALIZED, ("0,1", "2")), (po.SYM_NORMALIZED, ("1", "2", "3"))):
        full = po.build_adj_norm_hops(adj, nh, norm)
        for balance in ("nnz", "rows"):
            parts = _sharded_build(adj, nh, norm, world, balance)
            part = parts[0][3]
            assert all(p[3].bounds == part.bounds for p in parts)
            assert part.is_equal == (balance == "rows")
            for q, (rps, cis, vas, _) in enumerate(parts):
                r0, r1 = part.rows(q)
                for k, h in enumerate(full):
                    h = sp.csr_matrix(h)[r0:r1]
                    h.sort_indices()
                    assert np.array_equal(rps[k].cpu().numpy(), h.indptr), (norm, nh, balance, q, k)
                    want_cols = part.to_padded(torch.from_numpy(h.indices.astype(np.int32))).numpy()
                    assert np.array_equal(cis[k].cpu().numpy(), want_cols), (norm, nh, balance, q, k)
                    assert np.array_equal(vas[k].cpu().numpy(), h.data.astype(np.float32)), (norm, nh, balance, q, k)
            if balance == "nnz":
                work = sum(np.diff(sp.csr_matrix(h).indptr) for h in full) + len(full)
                assert part.imbalance(work) <= 1.10 and part.imbalance(work) <= RowPartition_equal_imbalance(n, world, work)


def RowPartition_equal_imbalance(n, world, work):
    from h2gcn_amd.partition import RowPartition

    return RowPartition.equal(n, world).imbalance(work)


def test_sharded_build_memory_and_balance_scale_with_the_partition():
    """A power-law graph with unshuffled hubs (n = 300k): the nnz-balanced 8-way split brings max/mean work per rank from
    ~1.6 down to <= 1.05, and a rank's build peaks at a fraction of the single-GPU build's memory (no whole ring is ever
    materialised: the scaling wall of the reference's host path, _dataset.py:147-157)."""
    from h2gcn_amd.datasets._dataset import sharded_adj_hops
    from h2gcn_amd.partition import RowPartition

    rng = np.random.default_rng(3)
    n = 300_000
    deg = np.minimum((rng.pareto(1.8, n) + 1.0) * 2.0, 2000).astype(np.int64)
    deg = np.sort(deg)[::-1]                                   # hubs first
    r = np.repeat(np.arange(n), deg)
    c = rng.integers(0, n, len(r))
    a = sp.csr_matrix((np.ones(2 * len(r), dtype=np.float32), (np.r_[r, c], np.r_[c, r])), shape=(n, n))
    a = po.remove_self_loops(a)
    a.data[:] = 1
    world, nh = 8, ("1", "2")
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    rps, cis, vas, _ = po.build_adj_norm_hops_device(a, nh, po.SYM_NORMALIZED, DEV)
    torch.cuda.synchronize()
    peak_full = torch.cuda.max_memory_allocated() - base
    work = sum((rp[1:] - rp[:-1]) for rp in rps).cpu().numpy() + len(rps)
    full_rows = [(rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy()) for rp, ci, va in zip(rps, cis, vas)]
    del rps, cis, vas
    torch.cuda.empty_cache()
    rp_d, ci_d, _ = po.upload_pattern(a, DEV)
    eq = RowPartition.equal(n, world)
    lens = [po.ring_row_lengths_window(rp_d, ci_d, n, 2, eq.rows(q)) for q in range(world)]
    del rp_d, ci_d
    peaks = []
    for q in (0, 3, 7):
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        rp, ci, va, part = sharded_adj_hops(a, nh, po.SYM_NORMALIZED, DEV, q, world, gather=lambda t: lens)
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated() - base)
        r0, r1 = part.rows(q)
        for k in range(2):
            lo, hi = full_rows[k][0][r0], full_rows[k][0][r1]
            assert np.array_equal(rp[k].cpu().numpy(), full_rows[k][0][r0:r1 + 1] - lo)
            assert np.array_equal(ci[k].cpu().numpy(), part.to_padded(torch.from_numpy(full_rows[k][1][lo:hi])).numpy())
            assert np.array_equal(va[k].cpu().numpy(), full_rows[k][2][lo:hi])
        del rp, ci, va
    assert part.imbalance(work) <= 1.05 and RowPartition.equal(n, world).imbalance(work) > 1.3, part.imbalance(work)
    print(f"sharded build: peak bytes per rank {peaks} vs single-GPU build {peak_full}; imbalance {part.imbalance(work):.4f} "
          f"(equal rows: {RowPartition.equal(n, world).imbalance(work):.2f})")
    # the ring kernels' transient level-0 scratch depends on n and the CU count only, not on the window: excluded
    from h2gcn_amd import _capi
    scratch = int(_capi.lib().h2gcn_ring_scratch_bytes(n))
    assert max(peaks) - scratch <= 0.35 * (peak_full - scratch), (peaks, peak_full, scratch)   # ~1/8 of the rings + A + counts
