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


# --------------------------------------------------------------------------------------------------------------
# Device-side construction.  Same rings, same values, built with torch ops on the GPU:
# neighbourhood growth is an expand -> sort -> unique over int64 keys (row * n + col), processed in row blocks
# so that the expansion (sum over reached (i, j) of deg(j)) never exceeds a fixed budget.  The reference's host
# SpGEMM `(A + I)^k` (scipy) is the scaling wall of its preprocessing (`_dataset.py:147-157`); this removes it for
# graphs whose exact-k-hop rings fit in HBM.  (At products scale the 2-hop ring itself is > 1e10 nonzeros.)
# --------------------------------------------------------------------------------------------------------------

def _expand_keys(keys, rowptr, colidx, deg, n, budget):
    """keys: sorted unique (i*n + j) of the current reach set.  Returns sorted unique keys of reach @ (A + I)."""
    import torch

    out = []
    cnt_all = deg[keys % n]
    # split into blocks whose expansion stays under `budget`
    csum = torch.cumsum(cnt_all, 0)
    total = int(csum[-1]) if len(keys) else 0
    start = 0
    while start < len(keys):
        base = int(csum[start - 1]) if start > 0 else 0
        stop = int(torch.searchsorted(csum, torch.tensor(base + budget, device=keys.device), right=True))
        stop = max(stop, start + 1)
        k_blk = keys[start:stop]
        cnt = cnt_all[start:stop]
        i = torch.div(k_blk, n, rounding_mode="floor")
        j = k_blk - i * n
        tot = int(cnt.sum())
        if tot > 0:
            src = torch.repeat_interleave(torch.arange(len(k_blk), device=keys.device), cnt)
            first = torch.cumsum(cnt, 0) - cnt
            off = torch.arange(tot, device=keys.device) - first[src]
            nb = colidx[rowptr[j[src]] + off].to(torch.int64)
            out.append(torch.unique(i[src] * n + nb))
        start = stop
    del total
    merged = torch.unique(torch.cat([keys] + out)) if out else keys
    return merged


def exact_hop_rings_device(rowptr, colidx, n: int, max_hop: int, budget: int = 1 << 27):
    """Device version of :func:`exact_hop_rings`: list of sorted int64 key tensors (row * n + col), ring 0 = I."""
    import torch

    dev = rowptr.device
    deg = (rowptr[1:] - rowptr[:-1])
    eye = torch.arange(n, device=dev, dtype=torch.int64) * (n + 1)
    reach = eye
    rings = [eye]
    for _ in range(int(max_hop)):
        nxt = _expand_keys(reach, rowptr, colidx, deg, n, budget)
        if nxt.numel() == reach.numel():
            break
        # ring = nxt \\ reach  (both sorted unique, reach is a subset of nxt)
        pos = torch.searchsorted(reach, nxt)
        pos = pos.clamp(max=reach.numel() - 1)
        rings.append(nxt[reach[pos] != nxt])
        reach = nxt
    return rings


def build_adj_norm_hops_device(adj_no_self_loops, adj_nhood: Sequence[str] = ("1", "2"),
                               norm: str = SYM_NORMALIZED, device="cuda:0", budget: int = 1 << 27):
    """Device-built ``adj_hops`` operands: ``(rowptr_list, colidx_list, vals_list, n)`` of CUDA tensors ready for
    :class:`~h2gcn_amd.hops.HopPlan`.  Bit-identical to :func:`build_adj_norm_hops` + the fp32 cast: the degree
    scalings are computed on the host with the same numpy call as the host path (length-n vectors), the products
    in fp64 on the device (IEEE multiplication rounds identically everywhere)."""
    import torch

    a = sp.csr_matrix(adj_no_self_loops)
    a.sum_duplicates()
    a.sort_indices()
    a.eliminate_zeros()
    n = a.shape[0]
    if a.shape[0] != a.shape[1]:
        raise ValueError(f"adjacency must be square, got {a.shape}")
    rowptr = torch.from_numpy(a.indptr.astype(np.int64)).to(device)
    colidx = torch.from_numpy(a.indices.astype(np.int64)).to(device)
    groups = parse_adj_nhood(adj_nhood)
    rings = exact_hop_rings_device(rowptr, colidx, n, max(max(g) for g in groups), budget)
    rps, cis, vas = [], [], []
    for g in groups:
        missing = [i for i in g if i >= len(rings)]
        if missing:
            raise ValueError(f"hop {missing[0]} requested but the graph's reachability saturates after {len(rings) - 1} hops")
        keys = rings[g[0]] if len(g) == 1 else torch.sort(torch.cat([rings[i] for i in g]))[0]
        rows = torch.div(keys, n, rounding_mode="floor")
        cols = keys - rows * n
        counts = torch.bincount(rows, minlength=n)
        rp = torch.zeros(n + 1, dtype=torch.int64, device=keys.device)
        rp[1:] = torch.cumsum(counts, 0)
        deg_host = counts.cpu().numpy().astype(np.float64)
        with np.errstate(divide="ignore"):
            if norm == SYM_NORMALIZED:
                s = np.power(deg_host, -0.5)
                s[np.isinf(s)] = 0.0
                st = torch.from_numpy(s).to(keys.device)
                vals = (st[rows] * 1.0) * st[cols]
            elif norm == RW_NORMALIZED:
                s = np.power(deg_host, -1.0)
                s[np.isinf(s)] = 0.0
                vals = torch.from_numpy(s).to(keys.device)[rows] * 1.0
            elif norm == ORDINARY:
                vals = torch.ones(len(keys), dtype=torch.float64, device=keys.device)
            else:
                raise ValueError(f"unknown normalisation {norm!r}")
        rps.append(rp)
        cis.append(cols.to(torch.int32).contiguous())
        vas.append(vals.to(torch.float32).contiguous())
    return rps, cis, vas, n
