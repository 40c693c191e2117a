"""CPU restatement of GCNLayer (reference h2gcn/models/_layers.py:54-81) and of its gradient.
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Two interchangeable back ends:

* ``*_scipy``: ``csr_matrix @ ndarray`` in fp32 -- scipy's csr_matvecs is the same loop nest as the upstream TF
  CPU kernel (per row, ascending column, zero-initialised, one multiply and one add per term);
* ``*_c``: the plain-C loops of ``spmm_oracle.c`` through ctypes (also the timed ``cpu_baseline`` of bench.py).
"""
import ctypes as C
from pathlib import Path

import numpy as np
import scipy.sparse as sp

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        p = Path(__file__).resolve().parent / "_build" / "liboracle.so"
        if not p.exists():
            raise RuntimeError(f"{p} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = C.CDLL(str(p))
    return _LIB


def _csr_parts(m):
    """(indptr int64, indices int32, data float32) with sorted indices."""
    if isinstance(m, tuple):
        ip, ix, da = m
    else:
        m = sp.csr_matrix(m)
        m.sort_indices()
        ip, ix, da = m.indptr, m.indices, m.data
    return (np.ascontiguousarray(ip, dtype=np.int64), np.ascontiguousarray(ix, dtype=np.int32),
            np.ascontiguousarray(da, dtype=np.float32))


def gcn_layer_scipy(hops, x, dtype=np.float32):
    """[N_rows, H, d] = stack([A_k @ x]) with A_k, x cast to `dtype` (float32 = reference arithmetic,
    float64 = rounding-free bound)."""
    x = np.asarray(x, dtype=dtype)
    outs = []
    for m in hops:
        ip, ix, da = _csr_parts(m)
        a = sp.csr_matrix((da.astype(dtype), ix, ip), shape=(len(ip) - 1, x.shape[0]))
        outs.append(a @ x)
    return np.stack(outs, axis=-2)


def _ptr_array(arrs, ctype):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data_as(C.c_void_p) for a in arrs])


def gcn_layer_c(hops, x, fma=False, threads=False):
    """``threads=True``: the OpenMP variant (rows spread over all host cores, identical bits)."""
    parts = [_csr_parts(m) for m in hops]
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_rows = len(parts[0][0]) - 1
    d = x.shape[1]
    H = len(parts)
    y = np.empty((n_rows, H, d), dtype=np.float32)
    L = _lib()
    if not fma:
        (L.oracle_gcn_layer_f32_mt if threads else L.oracle_gcn_layer_f32)(C.c_int(H), C.c_int64(n_rows), _ptr_array([p[0] for p in parts], None),
                               _ptr_array([p[1] for p in parts], None), _ptr_array([p[2] for p in parts], None),
                               x.ctypes.data_as(C.c_void_p), C.c_int64(x.shape[1]), C.c_int64(d),
                               y.ctypes.data_as(C.c_void_p))
    else:
        for k, (ip, ix, da) in enumerate(parts):
            L.oracle_spmm_csr_f32_fma(C.c_int64(n_rows), ip.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p),
                                      da.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_int64(d),
                                      C.c_int64(d), C.c_void_p(y.ctypes.data + 4 * k * d), C.c_int64(H * d))
    return y


def gcn_layer_f64acc(hops, x):
    """fp64-accumulated result of the fp32 operands, [N_rows, H, d] float64."""
    parts = [_csr_parts(m) for m in hops]
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_rows = len(parts[0][0]) - 1
    d = x.shape[1]
    H = len(parts)
    y = np.empty((n_rows, H, d), dtype=np.float64)
    L = _lib()
    for k, (ip, ix, da) in enumerate(parts):
        L.oracle_spmm_csr_f64acc(C.c_int64(n_rows), ip.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p),
                                 da.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_int64(d),
                                 C.c_int64(d), C.c_void_p(y.ctypes.data + 8 * k * d), C.c_int64(H * d))
    return y


def gcn_layer_grad_scipy(hops, dy, n_cols, dtype=np.float32):
    """dX = sum_k A_k^T @ dY[:, k, :]  ->  [n_cols, d]."""
    dy = np.asarray(dy, dtype=dtype)
    dx = np.zeros((n_cols, dy.shape[2]), dtype=dtype)
    for k, m in enumerate(hops):
        ip, ix, da = _csr_parts(m)
        a = sp.csr_matrix((da.astype(dtype), ix, ip), shape=(len(ip) - 1, n_cols))
        dx = dx + (a.T.tocsr() @ np.ascontiguousarray(dy[:, k, :]))
    return dx


def gcn_layer_grad_c(hops, dy, n_cols):
    parts = [_csr_parts(m) for m in hops]
    dy = np.ascontiguousarray(dy, dtype=np.float32)
    n_rows, H, d = dy.shape
    dx = np.empty((n_cols, d), dtype=np.float32)
    L = _lib()
    L.oracle_gcn_layer_grad_f32(C.c_int(H), C.c_int64(n_rows), C.c_int64(n_cols),
                                _ptr_array([p[0] for p in parts], None), _ptr_array([p[1] for p in parts], None),
                                _ptr_array([p[2] for p in parts], None), dy.ctypes.data_as(C.c_void_p),
                                C.c_int64(d), dx.ctypes.data_as(C.c_void_p))
    return dx


def rows_subset(hops, x, rows, dtype=np.float64):
    """Oracle restricted to a few output rows (cheap at any graph size): [len(rows), H, d]."""
    x = np.asarray(x)
    out = np.zeros((len(rows), len(hops), x.shape[1]), dtype=dtype)
    for k, m in enumerate(hops):
        ip, ix, da = m if isinstance(m, tuple) else _csr_parts(m)
        for r, i in enumerate(rows):
            s, e = int(ip[i]), int(ip[i + 1])
            acc = np.zeros(x.shape[1], dtype=dtype)
            for t in range(s, e):
                acc = acc + dtype(da[t]) * x[ix[t]].astype(dtype)
            out[r, k] = acc
    return out


def gcn_layer_tree(hops, x, long_threshold=256):
    """Kernel-order restatement (see ``oracle_spmm_tree_f32``): the library's documented canonical summation tree,
    bit-exact target for every HIP kernel variant.  [N_rows, H, d] float32."""
    parts = [_csr_parts(m) for m in hops]
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_rows, d, H = len(parts[0][0]) - 1, x.shape[1], len(parts)
    y = np.zeros((n_rows, H, d), dtype=np.float32)
    _lib().oracle_spmm_tree_f32(C.c_int(H), C.c_int64(n_rows), _ptr_array([p[0] for p in parts], None),
                                _ptr_array([p[1] for p in parts], None), _ptr_array([p[2] for p in parts], None),
                                x.ctypes.data_as(C.c_void_p), C.c_int64(d), C.c_int64(0), C.c_int64(d),
                                C.c_int(long_threshold), C.c_int(0), y.ctypes.data_as(C.c_void_p), C.c_int64(H * d), C.c_int64(d))
    return y


def gcn_layer_grad_tree(hops, dy, n_cols, long_threshold=256):
    """Adjoint in the library's documented order: the transposed operands (stable: ascending row order inside each
    output row), partials running on across the hops.  [n_cols, d] float32."""
    dy = np.ascontiguousarray(dy, dtype=np.float32)
    n_rows, H, d = dy.shape
    parts = []
    for m in hops:
        ip, ix, da = _csr_parts(m)
        t = sp.csr_matrix((da, ix, ip), shape=(n_rows, n_cols)).T.tocsr()  # csr -> csc view -> csr: stable counting sort
        t.sort_indices()
        parts.append(_csr_parts(t))
    dx = np.zeros((n_cols, d), dtype=np.float32)
    _lib().oracle_spmm_tree_f32(C.c_int(H), C.c_int64(n_cols), _ptr_array([p[0] for p in parts], None),
                                _ptr_array([p[1] for p in parts], None), _ptr_array([p[2] for p in parts], None),
                                dy.ctypes.data_as(C.c_void_p), C.c_int64(H * d), C.c_int64(d), C.c_int64(d),
                                C.c_int(long_threshold), C.c_int(1), dx.ctypes.data_as(C.c_void_p), C.c_int64(d), C.c_int64(0))
    return dx
