"""Full-size checks of the BASELINE shapes: EVERY row of Y against the oracle, not a sample.
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

The arithmetic is ``spmm_oracle.c``'s (``oracle_gcn_layer_f32_mt``: the reference's sequential fp32 loop, reference
``h2gcn/models/_layers.py:74-81`` -> TF's SparseTensorDenseMatMul; ``oracle_spmm_tree_f32_mt``: the library's documented
summation tree), with the output rows spread over the host's cores -- per-row arithmetic unchanged, identical bits.  The
operands are the synthetic CSR / features of ``h2gcn_amd/synth.py`` rebuilt on the host by the C restatement of that
generator (pinned bit-for-bit against the numpy one in tests/test_synth.py).

``python -m oracle.fullsize [shape ...]`` prints the oracle's order-independent checksum of Y per shape: that is where
``bench.py``'s ``N1_CHECKSUMS`` table comes from (CPU only; products takes about a minute on 8 cores).
"""
import ctypes as C

import numpy as np

from .gcn_layer import _lib, _ptr_array


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def stream_key(seed: int) -> int:
    """h2gcn_amd.synth._stream_key restated (splitmix64 of the seed)."""
    m = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def synth_hop_rows_c(raw_deg, n_cols, seed, r0, r1):
    """(rowptr int64 [r1-r0+1], colidx int32, vals float32) of rows [r0, r1): same bits as synth.synth_hop_rows_np."""
    raw_ptr = np.concatenate([[0], np.cumsum(raw_deg)]).astype(np.int64)
    cap = int(raw_ptr[r1] - raw_ptr[r0])
    rowptr = np.empty(r1 - r0 + 1, dtype=np.int64)
    col = np.empty(max(cap, 1), dtype=np.int32)
    val = np.empty(max(cap, 1), dtype=np.float32)
    f = _lib().oracle_synth_hop_rows
    f.restype = C.c_int64
    nnz = f(_p(raw_ptr), C.c_int64(r0), C.c_int64(r1), C.c_uint64(stream_key(seed)), C.c_int64(n_cols), _p(rowptr), _p(col), _p(val))
    return rowptr, col[:nnz], val[:nnz]


def synth_features_c(d, seed, r0, r1):
    out = np.empty((r1 - r0, d), dtype=np.float32)
    _lib().oracle_synth_features(C.c_int64(d), C.c_uint64(stream_key(seed)), C.c_int64(r0), C.c_int64(r1), _p(out))
    return out


def gcn_layer_tree_mt(parts, x, long_threshold=256):
    """[N_rows, H, d] float32 in the library's documented summation tree, all host cores.  parts: [(rowptr, colidx, vals)]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_rows, d, H = len(parts[0][0]) - 1, x.shape[1], len(parts)
    y = np.empty((n_rows, H, d), dtype=np.float32)
    _lib().oracle_spmm_tree_f32_mt(C.c_int(H), C.c_int64(n_rows), _ptr_array([p[0] for p in parts], None),
                                   _ptr_array([p[1] for p in parts], None), _ptr_array([p[2] for p in parts], None),
                                   _p(x), C.c_int64(d), C.c_int64(0), C.c_int64(d), C.c_int(long_threshold), C.c_int(0),
                                   _p(y), C.c_int64(H * d), C.c_int64(d))
    return y


def gcn_layer_grad_tree_mt(t_parts, dy, long_threshold=256):
    """[n_cols, d] float32: the adjoint in the library's documented order (partials running on across the hops) on the
    TRANSPOSED operands t_parts = [(rowptr, colidx, vals) of A_k^T], dy [n_rows, H, d]; all host cores."""
    dy = np.ascontiguousarray(dy, dtype=np.float32)
    _, H, d = dy.shape
    n_out = len(t_parts[0][0]) - 1
    dx = np.empty((n_out, d), dtype=np.float32)
    _lib().oracle_spmm_tree_f32_mt(C.c_int(H), C.c_int64(n_out), _ptr_array([p[0] for p in t_parts], None),
                                   _ptr_array([p[1] for p in t_parts], None), _ptr_array([p[2] for p in t_parts], None),
                                   _p(dy), C.c_int64(H * d), C.c_int64(d), C.c_int64(d), C.c_int(long_threshold), C.c_int(1),
                                   _p(dx), C.c_int64(d), C.c_int64(0))
    return dx


def gcn_layer_seq_mt(parts, x):
    """[N_rows, H, d] float32 in the REFERENCE's order (sequential fp32 sum per row, ascending columns), all host cores."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_rows, d, H = len(parts[0][0]) - 1, x.shape[1], len(parts)
    y = np.empty((n_rows, H, d), dtype=np.float32)
    _lib().oracle_gcn_layer_f32_mt(C.c_int(H), C.c_int64(n_rows), _ptr_array([p[0] for p in parts], None),
                                   _ptr_array([p[1] for p in parts], None), _ptr_array([p[2] for p in parts], None),
                                   _p(x), C.c_int64(d), C.c_int64(d), _p(y))
    return y


def bits_checksum(y) -> int:
    """Sum of the fp32 bit patterns (int32) in int64: bench.py's order-independent fingerprint of Y."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    f = _lib().oracle_bits_checksum_f32
    f.restype = C.c_int64
    return int(f(_p(y), C.c_int64(y.size)))


def count_bit_mismatches(a, b):
    """(number of elements with different bit patterns, flat index of the first one or -1)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape
    first = C.c_int64(-1)
    f = _lib().oracle_count_bit_mismatches_f32
    f.restype = C.c_int64
    return int(f(_p(a), _p(b), C.c_int64(a.size), C.byref(first))), int(first.value)


def max_abs_diff(a, b) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape
    f = _lib().oracle_max_abs_diff_f32
    f.restype = C.c_double
    return float(f(_p(a), _p(b), C.c_int64(a.size)))


def host_operands(shape, d=None):
    """(parts, x, n, d) of one entry of h2gcn_amd.synth.SHAPES, every row, built on the host."""
    from h2gcn_amd import synth          # the DEFINITION of the shapes (degree sequences, seeds); numpy only, no device work

    cfg = synth.SHAPES[shape]
    n, d = cfg["n"], d or cfg["d"]
    degs = synth.hop_degrees(cfg)
    parts = [synth_hop_rows_c(degs[k], n, (synth.SEED_A1, synth.SEED_A2)[k], 0, n) for k in range(2)]
    return parts, synth_features_c(d, synth.SEED_X, 0, n), n, d


def oracle_checksum(shape, d=None, block_rows=2_000_000):
    """Checksum of the oracle's Y for a whole shape, computed in row blocks (bounded memory: X + one block of A and Y)."""
    from h2gcn_amd import synth

    cfg = synth.SHAPES[shape]
    n, d = cfg["n"], d or cfg["d"]
    degs = synth.hop_degrees(cfg)
    x = synth_features_c(d, synth.SEED_X, 0, n)
    total, nnz = 0, [0, 0]
    for r0 in range(0, n, block_rows):
        r1 = min(n, r0 + block_rows)
        parts = [synth_hop_rows_c(degs[k], n, (synth.SEED_A1, synth.SEED_A2)[k], r0, r1) for k in range(2)]
        total += bits_checksum(gcn_layer_tree_mt(parts, x))
        nnz = [z + len(p[1]) for z, p in zip(nnz, parts)]
    # the device sums in wrapping int64 (torch); so does this
    total = (total + (1 << 63)) % (1 << 64) - (1 << 63)
    return total, nnz


def oracle_adjoint_checksum(shape, d=None, w_seed=77):
    """Checksum of the oracle's dX = sum_k A_k^T W[:, k, :] for a whole shape (W = the counter-based features of seed 77, as bench.py's
    adjoint leg uses them), in the library's documented order on a host-built transpose (scipy: stable counting sort)."""
    import scipy.sparse as sp

    parts, _, n, d = host_operands(shape, d)
    w = synth_features_c(2 * d, w_seed, 0, n).reshape(n, 2, d)
    t_parts = []
    for rp, ci, va in parts:
        t = sp.csr_matrix((va, ci, rp), shape=(n, n)).T.tocsr()
        t.sort_indices()
        t_parts.append((t.indptr.astype(np.int64), t.indices.astype(np.int32), t.data.astype(np.float32)))
    del parts
    return bits_checksum(gcn_layer_grad_tree_mt(t_parts, w))


if __name__ == "__main__":
    import sys
    import time

    args = [a_ for a_ in sys.argv[1:] if a_ != "--adjoint"]
    for name in args or ["arxiv", "products"]:
        shape, _, dd = name.partition(":")
        t = time.time()
        if "--adjoint" in sys.argv:
            print(f'adjoint ("{shape}", {int(dd) if dd else "default d"}): {oracle_adjoint_checksum(shape, int(dd) if dd else None)}   [{time.time() - t:.1f} s]', flush=True)
            continue
        ck, nnz = oracle_checksum(shape, int(dd) if dd else None)
        print(f'("{shape}", {int(dd) if dd else "default d"}): {ck}   nnz per hop {nnz}   [{time.time() - t:.1f} s]', flush=True)
