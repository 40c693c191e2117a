"""CPU restatement (numpy/scipy) of how the reference builds the hop operands.  TEST INFRASTRUCTURE ONLY.

Follows, function by function:
* ``remove_eye`` / ``add_eye``      <- TransformSPAdj.removeEye / addEye, h2gcn/datasets/_dataset.py:126-136
* ``nhood_split``                   <- TransformSPAdj.nhoodSplit, h2gcn/datasets/_dataset.py:138-158
* ``normalize``                     <- TransformSPAdj.normalize,  h2gcn/datasets/_dataset.py:109-124
* ``adj_norm_hops``                 <- PlanetoidData.getTensors (getAdjNormHops branch), _dataset.py:559-576
* ``to_canonical_csr``              <- sparse2Tensor + tf.sparse.reorder, _dataset.py:528-535
* ``row_normalize_features``        <- PlanetoidData.row_normalize_features, _dataset.py:502-509
* ``preprocess``                    <- preprocessing_data, h2gcn/models/H2GCN.py:46-54 (order of operations)
"""
from itertools import chain

import numpy as np
import scipy.sparse as sp

SYM, RW, ORDINARY = "sym", "rw", "ordinary"


def remove_eye(adj):
    a = sp.csr_matrix(adj).tolil(copy=True)
    a.setdiag(0)
    return a.tocsr()


def add_eye(adj):
    a = sp.csr_matrix(adj).tolil(copy=True)
    a.setdiag(1)
    return a.tocsr()


def nhood_split(adj, nhood):
    """[I, N1, N2, ...]: N_i has a 1 at (u, v) iff the shortest-path distance u->v is exactly i.
    Stops early (shorter list) once reachability stops growing, like the reference (:152-153)."""
    adj = sp.csr_matrix(adj)
    assert adj.ndim == 2 and adj.shape[0] == adj.shape[1]
    n = adj.shape[0]
    reach = sp.eye(n)
    out = [reach]
    step = adj + sp.eye(n)
    total = 0
    i = 0
    while i < nhood:
        prev = reach
        reach = reach @ step
        reach = (reach > 0).astype(reach.dtype)
        s = reach.sum()
        if s == total:
            break
        total = s
        i += 1
        out.append(reach - prev)
    return out


def normalize(adj, ntype):
    if ntype == ORDINARY:
        return adj
    deg = np.asarray(adj.sum(axis=1)).reshape(-1)
    with np.errstate(divide="ignore"):
        if ntype == SYM:
            s = np.power(deg, -0.5)
            s[np.isinf(s)] = 0.0
            D = sp.diags(s)
            return D @ adj @ D
        if ntype == RW:
            s = np.power(deg, -1.0)
            s[np.isinf(s)] = 0.0
            return sp.diags(s) @ adj
    raise ValueError(ntype)


def adj_norm_hops(adj_no_eye, adj_nhood=("1", "2"), ntype=SYM):
    """list of normalised hop matrices for --adj_nhood groups such as ["1", "2"] or ["0,1", "2"]."""
    groups = [[int(x) for x in g.split(",")] for g in adj_nhood]
    hop_max = max(chain(*groups))
    splits = nhood_split(adj_no_eye, hop_max)
    merged = [sum(splits[i] for i in g) for g in groups]
    return [normalize(m, ntype) for m in merged]


def to_canonical_csr(m, dtype=np.float32):
    """fp32 cast + row-major ordering (COO -> reorder) expressed as CSR with sorted indices.
    Explicit zeros that scipy keeps (e.g. from `mt - prev_mt`) are kept as the reference keeps them in COO --
    `eliminate_zeros` is applied because scipy's subtraction already drops them and tocoo() carries no
    duplicates here."""
    c = sp.coo_matrix(m)
    order = np.lexsort((c.col, c.row))
    rows, cols, data = c.row[order], c.col[order], c.data[order].astype(dtype)
    indptr = np.zeros(m.shape[0] + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr)
    return indptr, cols.astype(np.int32), data


def row_normalize_features(features):
    features = sp.csr_matrix(features)
    with np.errstate(divide="ignore"):
        inv = np.power(np.asarray(features.sum(1)).reshape(-1), -1.0)
    inv[np.isinf(inv)] = 0.0
    return sp.diags(inv) @ features


def preprocess(adj, features, adj_nhood=("1", "2"), ntype=SYM, feature_normalize=True):
    """feature row-normalisation (unless disabled) -> remove self loops -> normalised hop list."""
    if feature_normalize:
        features = row_normalize_features(features)
    adj = remove_eye(adj)
    return adj, features, adj_norm_hops(adj, adj_nhood, ntype)
