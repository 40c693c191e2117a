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
# Device-side construction: the same rings and values, built by hand-written HIP kernels
# behind the C ABI (``h2gcn_ring_count`` / ``h2gcn_ring_fill`` / ``h2gcn_hop_normalize``, csrc/rings.hip).  A ring is a
# boolean set expression over CSR patterns evaluated row by row in a two-level LDS bitmap -- the reference's host
# SpGEMM ``(A + I)^k`` (scipy, ``_dataset.py:147-157``) is the scaling wall of its preprocessing; this removes it for
# graphs whose exact-k-hop rings fit in HBM.  (At products scale the 2-hop ring itself is > 1e10 nonzeros.)
# Host involvement: two scalar read-backs per ring (its nonzero count, to allocate it, and its largest row, to size
# the scaling table) -- nothing proportional to the graph crosses PCIe.
# --------------------------------------------------------------------------------------------------------------

def _pattern_args(patterns):
    import ctypes as C

    k = len(patterns)
    arr_t = C.c_void_p * max(k, 1)
    return k, arr_t(*[p[0].data_ptr() for p in patterns]), arr_t(*[p[1].data_ptr() for p in patterns])


def ring_set_device(n: int, device, a=None, frontier=None, add=(), add_diag: bool = False, sub=(), sub_diag: bool = False,
                    rows=None, count_only: bool = False):
    """``out[i] = (U_{j in frontier[i]} a[j]  U  U add[i]  U {i}?) \\ (U sub[i] U {i}?)`` over CSR patterns
    ``(rowptr int64, colidx int32)`` on ``device``; returns the result pattern with ascending columns.

    ``rows = (r0, r1)``: evaluate the window of rows ``[r0, r1)`` only -- ``frontier`` / ``add`` / ``sub`` and the result
    are then CSRs of that window (``r1 - r0 + 1`` local row pointers), ``a`` stays the whole matrix and ``{i}`` is the
    global row id (``h2gcn_ring_count_rows``).  ``count_only``: return ``(rowptr, None)`` without filling the columns."""
    import ctypes as C

    import torch

    from . import _capi

    L = _capi.lib()
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    a_rp, a_ci = a if a is not None else (None, None)
    f_rp, f_ci = frontier if frontier is not None else (None, None)
    n_add, add_rp, add_ci = _pattern_args(list(add))
    n_sub, sub_rp, sub_ci = _pattern_args(list(sub))
    r0, r1 = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    with torch.cuda.device(device):
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        sb = int(L.h2gcn_ring_scratch_bytes(n))
        scratch = torch.empty(sb, dtype=torch.uint8, device=device)
        rowptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=device)
        nnz = C.c_int64()
        common = (n, r0, r1 - r0, ptr(a_rp), ptr(a_ci), ptr(f_rp), ptr(f_ci), n_add, add_rp, add_ci, int(add_diag), n_sub, sub_rp,
                  sub_ci, int(sub_diag))
        _capi.check(L.h2gcn_ring_count_rows(*common, ptr(rowptr), C.byref(nnz), ptr(scratch), sb, stream))
        if count_only:
            return rowptr, None
        colidx = torch.empty(nnz.value, dtype=torch.int32, device=device)
        if nnz.value:
            _capi.check(L.h2gcn_ring_fill_rows(*common, ptr(rowptr), ptr(colidx), ptr(scratch), sb, stream))
    return rowptr, colidx


def identity_pattern(n: int, device):
    import torch

    return (torch.arange(n + 1, dtype=torch.int64, device=device), torch.arange(n, dtype=torch.int32, device=device))


def _window(pattern, r0: int, r1: int):
    """Rows [r0, r1) of a device CSR pattern as a window CSR (local row pointers)."""
    rowptr, colidx = pattern
    lo, hi = int(rowptr[r0]), int(rowptr[r1])
    return (rowptr[r0:r1 + 1] - lo).contiguous(), colidx[lo:hi].contiguous()


def exact_hop_rings_device(rowptr, colidx, n: int, max_hop: int, rows=None, count_last: bool = False):
    """Device version of :func:`exact_hop_rings`: list of CSR patterns ``(rowptr, colidx)``; ring 0 = I.  Like the
    reference (``_dataset.py:152-153``) the list ends early once reachability stops growing.

    ``rows = (r0, r1)``: only that row window of every ring (window CSRs; what one rank of a row partition needs --
    ring k of the window is grown from the whole ``A`` and the window's rows of the lower rings).  The early exit is then
    left to the caller (it is a property of the whole ring).  ``count_last``: the last ring is only counted
    (``(rowptr, None)``) -- enough to learn its row lengths."""
    dev = rowptr.device
    a = (rowptr, colidx)
    r0, r1 = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    eye = (torch_arange(r1 - r0 + 1, dev), torch_arange_i32(r0, r1, dev))
    rings = [eye]
    for k in range(1, int(max_hop) + 1):
        only_count = count_last and k == int(max_hop)
        if k == 1:   # bin(I (A + I)) - I: the off-diagonal pattern of A
            ring = ring_set_device(n, dev, add=[a if rows is None else _window(a, r0, r1)], sub_diag=True, rows=rows,
                                   count_only=only_count)
        else:        # expand the frontier ring_{k-1}, drop everything within distance < k
            ring = ring_set_device(n, dev, a=a, frontier=rings[k - 1], sub=rings[1:k], sub_diag=True, rows=rows,
                                   count_only=only_count)
        if rows is None and not only_count and ring[1].numel() == 0 and k > 1:   # reach did not grow (the reference's
            break                            # edge_sum test; its first ring is always kept: edge_sum starts at 0, :145-153)
        rings.append(ring)
    return rings


def torch_arange(m: int, device):
    import torch

    return torch.arange(m, dtype=torch.int64, device=device)


def torch_arange_i32(r0: int, r1: int, device):
    import torch

    return torch.arange(r0, r1, dtype=torch.int32, device=device)


_S_TABLE_CACHE = {}


def _scaling_table(norm: str, length: int, device):
    """fp64 ``k^-1/2`` (SYM) / ``k^-1`` (RW) for k = 0 .. length-1 with ``inf -> 0`` -- the reference's own numpy
    expression (``np.power(rowsum, -0.5)``, ``_dataset.py:115-123``), evaluated on the possible row sums."""
    import torch

    key = (norm, str(device))
    tab = _S_TABLE_CACHE.get(key)
    if tab is None or tab.numel() < length:
        size = max(length, 1024, 2 * (tab.numel() if tab is not None else 0))
        k = np.arange(size, dtype=np.float64)
        with np.errstate(divide="ignore"):
            s = np.power(k, -0.5 if norm == SYM_NORMALIZED else -1.0)
        s[np.isinf(s)] = 0.0
        tab = torch.from_numpy(s).to(device)
        _S_TABLE_CACHE[key] = tab
    return tab


def normalize_pattern_device(pattern, n: int, norm: str, col_len=None):
    """fp32 values of the normalised hop matrix for a CSR pattern (``h2gcn_hop_normalize_rows``).  ``pattern`` is the
    whole square matrix, or -- with ``col_len`` (int64 ``[n]``: the row lengths of the WHOLE matrix) -- a row window."""
    import ctypes as C

    import torch

    from . import _capi

    rowptr, colidx = pattern
    dev = rowptr.device
    vals = torch.empty(colidx.numel(), dtype=torch.float32, device=dev)
    mode = {ORDINARY: 0, SYM_NORMALIZED: 1, RW_NORMALIZED: 2}.get(norm)
    if mode is None:
        raise ValueError(f"unknown normalisation {norm!r}")
    if colidx.numel() == 0:
        return vals
    tab = None
    if mode != 0:
        max_deg = int(col_len.max()) if col_len is not None else int((rowptr[1:] - rowptr[:-1]).max())
        tab = _scaling_table(norm, max_deg + 1, dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(_capi.lib().h2gcn_hop_normalize_rows(rowptr.numel() - 1, C.c_void_p(rowptr.data_ptr()), C.c_void_p(colidx.data_ptr()), mode,
                                                         C.c_void_p(tab.data_ptr()) if tab is not None else None,
                                                         tab.numel() if tab is not None else 0,
                                                         C.c_void_p(col_len.data_ptr()) if col_len is not None else None,
                                                         C.c_void_p(vals.data_ptr()), stream))
    return vals


def build_adj_norm_hops_device(adj_no_self_loops, adj_nhood: Sequence[str] = ("1", "2"),
                               norm: str = SYM_NORMALIZED, device="cuda:0"):
    """Device-built ``adj_hops`` operands: ``(rowptr_list, colidx_list, vals_list, n)`` of CUDA tensors ready for
    :class:`~h2gcn_amd.hops.HopPlan`.  Bit-identical to :func:`build_adj_norm_hops` + the fp32 cast.  The adjacency
    may be a scipy matrix (uploaded once) or a ``(rowptr int64, colidx int32)`` pair already on the device."""
    import torch

    if isinstance(adj_no_self_loops, tuple):
        rowptr, colidx = adj_no_self_loops
        n = rowptr.numel() - 1
        device = rowptr.device
    else:
        a = sp.csr_matrix(adj_no_self_loops)
        a.sum_duplicates()
        a.sort_indices()
        a.eliminate_zeros()
        n = a.shape[0]
        if a.shape[0] != a.shape[1]:
            raise ValueError(f"adjacency must be square, got {a.shape}")
        rowptr = torch.from_numpy(a.indptr.astype(np.int64)).to(device)
        colidx = torch.from_numpy(a.indices.astype(np.int32)).to(device)
    device = torch.device(device)
    groups = parse_adj_nhood(adj_nhood)
    rings = exact_hop_rings_device(rowptr, colidx, n, max(max(g) for g in groups))
    rps, cis, vas = [], [], []
    for g in groups:
        missing = [i for i in g if i >= len(rings)]
        if missing:
            raise ValueError(f"hop {missing[0]} requested but the graph's reachability saturates after {len(rings) - 1} hops")
        if len(g) == 1:
            pat = rings[g[0]]
        else:  # merged group: union of the (disjoint) member rings, reference ``sum(...)`` (_dataset.py:571-572)
            pat = ring_set_device(n, device, add=[rings[i] for i in g if i != 0], add_diag=0 in g)
        rps.append(pat[0])
        cis.append(pat[1])
        vas.append(normalize_pattern_device(pat, n, norm))
    return rps, cis, vas, n


def upload_pattern(adj_no_self_loops, device):
    """scipy adjacency -> ``(rowptr int64, colidx int32, n)`` on ``device`` (canonical: duplicates summed, sorted)."""
    import torch

    a = sp.csr_matrix(adj_no_self_loops)
    a.sum_duplicates()
    a.sort_indices()
    a.eliminate_zeros()
    if a.shape[0] != a.shape[1]:
        raise ValueError(f"adjacency must be square, got {a.shape}")
    return (torch.from_numpy(a.indptr.astype(np.int64)).to(device), torch.from_numpy(a.indices.astype(np.int32)).to(device),
            a.shape[0])


def ring_row_lengths_window(rowptr, colidx, n: int, max_hop: int, rows):
    """int64 ``[max_hop + 1, r1 - r0]``: lengths of the window's rows of ring 0 .. ring max_hop (phase A of the sharded
    build: every rank counts an equal share of the rows; only the last ring is count-only)."""
    import torch

    rings = exact_hop_rings_device(rowptr, colidx, n, max_hop, rows=rows, count_last=True)
    return torch.stack([r[0][1:] - r[0][:-1] for r in rings])


def build_adj_norm_hops_window(rowptr, colidx, n: int, rows, ring_len, adj_nhood: Sequence[str] = ("1", "2"),
                               norm: str = SYM_NORMALIZED):
    """The rows ``[r0, r1)`` of the ``adj_hops`` operands, built from the whole adjacency pattern and nothing else
    proportional to the whole rings: ``(rowptr_list, colidx_list, vals_list)`` -- window CSRs with GLOBAL column ids.
    ``ring_len`` = int64 ``[max_hop + 1, n]``, the row lengths of every whole ring (all-gathered phase-A counts): SYM
    needs ``s[deg_j]`` of arbitrary columns j, and the reference's early exit (a ring with no entries ends the list,
    ``_dataset.py:152-153``) is a property of the whole ring.  Values are bit-identical to the rows of
    :func:`build_adj_norm_hops_device`."""
    groups = parse_adj_nhood(adj_nhood)
    max_hop = max(max(g) for g in groups)
    total = ring_len.sum(dim=1)
    n_rings = 1
    for k in range(1, max_hop + 1):     # reference: first ring always kept, later ones until reach stops growing
        if k > 1 and int(total[k]) == 0:
            break
        n_rings += 1
    for g in groups:
        missing = [i for i in g if i >= n_rings]
        if missing:
            raise ValueError(f"hop {missing[0]} requested but the graph's reachability saturates after {n_rings - 1} hops")
    dev = rowptr.device
    rings = exact_hop_rings_device(rowptr, colidx, n, max_hop, rows=rows)
    rps, cis, vas = [], [], []
    for g in groups:
        if len(g) == 1:
            pat = rings[g[0]]
        else:
            pat = ring_set_device(n, dev, add=[rings[i] for i in g if i != 0], add_diag=0 in g, rows=rows)
        col_len = ring_len[g].sum(dim=0).contiguous()     # member rings are disjoint: the group's row lengths add up
        rps.append(pat[0])
        cis.append(pat[1])
        vas.append(normalize_pattern_device(pat, n, norm, col_len=col_len))
    return rps, cis, vas
