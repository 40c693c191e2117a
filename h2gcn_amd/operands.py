"""Host-side construction of H2GCN's hop operands: exact-k-hop neighbourhood matrices, normalised.

This is the step that FEEDS the hot path, run once per graph on the host (the reference does the same with
scipy): ``preprocessing_data`` (``h2gcn/models/H2GCN.py:46-54``) -> ``adj_remove_eye`` -> ``getTensors(
getAdjNormHops=args.adj_nhood)`` (``h2gcn/datasets/_dataset.py:559-576``) -> ``TransformSPAdj.nhoodSplit``
(``:138-158``) and ``.normalize`` (``:109-124``).  Results are bit-identical to the reference's after the fp32
cast it applies in ``sparse2Tensor`` (``:528-535``) -- tests/test_operands.py checks that against fixtures made
by running the reference's own code.

Implementation notes (not a transcription): neighbourhood growth works on boolean CSR structure only
(``reach_k = pattern(reach_{k-1} @ (A + I))``), exact-k rings are obtained by pattern difference, and the
normalisation scales the ring's ``data`` array in place (``s[row] * a * s[col]``) instead of two sparse-diagonal
products.  At products scale the exact 2-hop ring is > 1e10 nonzeros and is out of reach for any host SpGEMM --
the large benchmark shapes therefore supply the 2-hop CSR directly (BASELINE.json config 4, "2-hop adj
precomputed").
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import scipy.sparse as sp

SYM_NORMALIZED = "sym"   # D^-1/2 A_k D^-1/2   (reference default, _dataset.py:537-538)
RW_NORMALIZED = "rw"     # D^-1 A_k            ("row-normalised")
ORDINARY = "ordinary"    # A_k unchanged


def remove_self_loops(adj) -> sp.csr_matrix:
    """Drop the diagonal (reference ``removeEye``, _dataset.py:131-136)."""
    a = sp.csr_matrix(adj, copy=True)
    a.setdiag(0)
    a.eliminate_zeros()
    return a


def _pattern(m) -> sp.csr_matrix:
    m = sp.csr_matrix(m)
    m.eliminate_zeros()
    m.sum_duplicates()
    return sp.csr_matrix((np.ones(m.nnz, dtype=np.float64), m.indices, m.indptr), shape=m.shape)


def exact_hop_rings(adj, max_hop: int) -> List[sp.csr_matrix]:
    """``[I, N_1, ..., N_k]`` with ``N_i[u, v] = 1`` iff dist(u, v) == i (reference ``nhoodSplit``).

    Like the reference, growth stops early -- returning a shorter list -- once the reachable set stops
    growing (``_dataset.py:152-153``)."""
    adj = sp.csr_matrix(adj)
    if adj.shape[0] != adj.shape[1]:
        raise ValueError(f"adjacency must be square, got {adj.shape}")
    n = adj.shape[0]
    eye = sp.identity(n, dtype=np.float64, format="csr")
    step = _pattern(adj + eye)
    reach = eye
    rings = [eye]
    reached = 0
    for _ in range(int(max_hop)):
        nxt = _pattern(reach @ step)
        if nxt.nnz == reached:
            break
        reached = nxt.nnz
        ring = nxt - reach          # reach's pattern is a subset of nxt's: entries are exactly 0 or 1
        ring.eliminate_zeros()
        rings.append(sp.csr_matrix(ring))
        reach = nxt
    return rings


def normalize_hop(m, kind: str = SYM_NORMALIZED) -> sp.csr_matrix:
    """SYM: ``s_i * a_ij * s_j`` with ``s = rowsum^-1/2``; RW: ``rowsum_i^-1 * a_ij``; empty rows scale by 0
    (the reference's ``inf -> 0``, ``_dataset.py:115-123``).  Degrees are those of THIS hop matrix."""
    m = sp.csr_matrix(m, dtype=np.float64, copy=True)
    if kind == ORDINARY:
        return m
    deg = np.asarray(m.sum(axis=1)).reshape(-1)
    rows = np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))
    with np.errstate(divide="ignore"):
        if kind == SYM_NORMALIZED:
            s = np.power(deg, -0.5)
            s[np.isinf(s)] = 0.0
            m.data = (s[rows] * m.data) * s[m.indices]
        elif kind == RW_NORMALIZED:
            s = np.power(deg, -1.0)
            s[np.isinf(s)] = 0.0
            m.data = s[rows] * m.data
        else:
            raise ValueError(f"unknown normalisation {kind!r}")
    return m


def parse_adj_nhood(adj_nhood: Sequence[str]) -> List[List[int]]:
    """``["1", "2"]`` -> ``[[1], [2]]``; ``["0,1", "2"]`` -> ``[[0, 1], [2]]`` (``--adj_nhood``, H2GCN.py:17)."""
    return [[int(x) for x in str(g).split(",")] for g in adj_nhood]


def build_adj_norm_hops(adj_no_self_loops, adj_nhood: Sequence[str] = ("1", "2"),
                        norm: str = SYM_NORMALIZED) -> List[sp.csr_matrix]:
    """The ``adj_hops`` operand list: for each ``--adj_nhood`` group, the union of the named exact-hop rings,
    normalised (reference ``getTensors``, getAdjNormHops branch)."""
    groups = parse_adj_nhood(adj_nhood)
    rings = exact_hop_rings(adj_no_self_loops, max(max(g) for g in groups))
    hops = []
    for g in groups:
        missing = [i for i in g if i >= len(rings)]
        if missing:
            raise ValueError(f"hop {missing[0]} requested but the graph's reachability saturates after {len(rings) - 1} hops")
        merged = rings[g[0]]
        for i in g[1:]:
            merged = merged + rings[i]
        hops.append(normalize_hop(merged, norm))
    return hops


def row_normalize_features(features) -> sp.csr_matrix:
    """``diag(rowsum^-1) @ F`` with all-zero rows left zero (reference ``row_normalize_features``, :502-509).

    Computed in the features' own floating dtype, as the reference does (its planetoid loader yields float32
    for cora and float64 for citeseer) -- the fp32 results differ by 1 ulp otherwise."""
    f = sp.csr_matrix(features, copy=True)
    if not np.issubdtype(f.dtype, np.floating):
        f = f.astype(np.float64)
    rs = np.asarray(f.sum(axis=1)).reshape(-1)
    with np.errstate(divide="ignore"):
        inv = np.power(rs, -1.0)
    inv[np.isinf(inv)] = 0.0
    f.data = inv[np.repeat(np.arange(f.shape[0]), np.diff(f.indptr))] * f.data
    return f
