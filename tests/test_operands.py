"""CPU suite: the PRODUCT's host-side operand builder (h2gcn_amd.operands) against the golden fixtures."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_planetoid_golden
from h2gcn_amd import operands as po


def _same(a, b):
    a = sp.csr_matrix(a); a.sort_indices(); a.eliminate_zeros()
    b = sp.csr_matrix(b); b.sort_indices()
    assert a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    assert np.array_equal(a.data.astype(np.float32), b.data.astype(np.float32))  # bit-exact after the fp32 cast


@pytest.mark.parametrize("name", ["cora", "citeseer"])
def test_hops_match_reference(name):
    g = load_planetoid_golden(name)
    adj = po.remove_self_loops(g["adj_raw"])
    _same(adj, g["adj_noeye"])
    rings = po.exact_hop_rings(adj, 2)
    assert [r.nnz for r in rings] == list(g["split_nnz"])
    for norm in (po.SYM_NORMALIZED, po.RW_NORMALIZED):
        hops = po.build_adj_norm_hops(adj, ("1", "2"), norm)
        _same(hops[0], g[f"hop1_{norm}"])
        _same(hops[1], g[f"hop2_{norm}"])
    merged = po.build_adj_norm_hops(adj, ("0,1", "2"))
    _same(merged[0], g["hop01_sym"])
    _same(po.row_normalize_features(g["feat_raw"]), g["feat_rownorm"])


def test_reachability_saturation_and_errors():
    # path graph 0-1-2: 3 hops requested, reachability saturates after 2
    a = sp.csr_matrix(np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]], dtype=np.float32))
    rings = po.exact_hop_rings(a, 5)
    assert len(rings) == 3
    assert rings[2].toarray().tolist() == [[0, 0, 1], [0, 0, 0], [1, 0, 0]]
    with pytest.raises(ValueError):
        po.build_adj_norm_hops(a, ("1", "3"))
    with pytest.raises(ValueError):
        po.exact_hop_rings(sp.csr_matrix((2, 3)), 1)
    assert po.parse_adj_nhood(["0,1", "2"]) == [[0, 1], [2]]


def test_empty_rows_scale_to_zero():
    a = sp.csr_matrix(np.array([[0, 1, 0], [1, 0, 0], [0, 0, 0]], dtype=np.float32))
    for norm in (po.SYM_NORMALIZED, po.RW_NORMALIZED):
        h = po.build_adj_norm_hops(a, ("1",), norm)[0]
        assert np.isfinite(h.data).all() and h[2].nnz == 0


