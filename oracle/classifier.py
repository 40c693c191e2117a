"""CPU restatement of the classifier side of H2GCN -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What it restates: keras ``Dropout(rate)`` followed by the output ``Dense`` (``D0.5-MO`` of the network-setup DSL; reference
``h2gcn/models/H2GCN.py:235-257`` builds the two layers, ``:308-325`` calls them in order):

    Z = (X * M / keep) @ W + b,     M ~ Bernoulli(keep) per element,

and its gradients ``dX = (G @ W^T) * M / keep``, ``dW = (X * M / keep)^T @ G``, ``db = sum_rows G``.  The reference's mask
comes from TensorFlow's stateful RNG, which no other implementation can reproduce, so parity is statistical (keep rate) plus
exact agreement on everything that is a function of the mask.  The HIP kernels draw the mask from a COUNTER-BASED generator
-- a function of (seed, step, row, column) -- so that forward and backward recompute it instead of storing it; that
generator is documented in ``include/h2gcn_hip.h`` and restated here bit for bit (``keep_mask``).
"""
import numpy as np

_M1, _M2, _GOLD = np.uint32(0x7FEB352D), np.uint32(0x846CA68B), np.uint32(0x9E3779B9)


def _mix(h):
    """lowbias32 avalanche round (all arithmetic modulo 2**32)."""
    h = h ^ (h >> np.uint32(16))
    h = h * _M1
    h = h ^ (h >> np.uint32(15))
    h = h * _M2
    return h ^ (h >> np.uint32(16))


def keep_mask(n_rows: int, k: int, keep_prob: float, seed: int, step: int, row0: int = 0):
    """bool [n_rows, k].  One keyed hash per aligned GROUP of four columns of a row -- gid = row * ceil(k/4) + col // 4.
    keep_prob * 65536 a multiple of 256: the four bytes of the hash word are the elements' fields, kept iff byte <
    keep_prob * 256; otherwise a second round yields a second word and the fields are 16 bits wide (kept iff field <
    floor(keep_prob * 65536))."""
    if keep_prob >= 1.0:
        return np.ones((n_rows, k), dtype=bool)
    gpr = (k + 3) // 4
    thr16 = min(int(np.float32(keep_prob).astype(np.float64) * 65536.0), 65536)
    with np.errstate(over="ignore"):
        gid = (np.arange(row0, row0 + n_rows, dtype=np.uint64)[:, None] * np.uint64(gpr) + np.arange(gpr, dtype=np.uint64)[None, :])
        lo, hi = (gid & np.uint64(0xFFFFFFFF)).astype(np.uint32), (gid >> np.uint64(32)).astype(np.uint32)
        key0 = _mix(np.uint32(seed & 0xFFFFFFFF) ^ _mix(np.uint32(step & 0xFFFFFFFF) + _GOLD))
        key1 = _mix(np.uint32((seed >> 32) & 0xFFFFFFFF) ^ np.uint32((step >> 32) & 0xFFFFFFFF) ^ key0 ^ np.uint32(0x85EBCA6B))
        h = lo ^ ((hi << np.uint32(16)) | (hi >> np.uint32(16))) ^ key0
        h = h ^ (h >> np.uint32(16))
        h = h * _M1
        h = h ^ key1
        h = h ^ (h >> np.uint32(15))
        h = h * _M2
        w0 = h ^ (h >> np.uint32(16))
        if thr16 % 256 == 0:
            fields = np.stack([(w0 >> np.uint32(8 * j)) & np.uint32(0xFF) for j in range(4)], axis=-1)
            thr = np.uint32(thr16 // 256)
        else:
            w1 = _mix(w0 ^ np.uint32(0x85EBCA6B))
            fields = np.stack([w0 & np.uint32(0xFFFF), w0 >> np.uint32(16), w1 & np.uint32(0xFFFF), w1 >> np.uint32(16)], axis=-1)
            thr = np.uint32(thr16)
    return (fields < thr).reshape(n_rows, gpr * 4)[:, :k]


def dropout_dense(x, w, b, keep_prob, seed, step, dtype=np.float64):
    m = keep_mask(x.shape[0], x.shape[1], keep_prob, seed, step)
    xd = np.where(m, x.astype(dtype) / dtype(keep_prob), dtype(0))
    z = xd @ w.astype(dtype)
    return z if b is None else z + b.astype(dtype)


def dropout_dense_grad(x, w, g, keep_prob, seed, step, dtype=np.float64):
    """(dX, dW, db) in ``dtype``."""
    m = keep_mask(x.shape[0], x.shape[1], keep_prob, seed, step)
    xd = np.where(m, x.astype(dtype) / dtype(keep_prob), dtype(0))
    dx = np.where(m, (g.astype(dtype) @ w.astype(dtype).T) / dtype(keep_prob), dtype(0))
    return dx, xd.T @ g.astype(dtype), g.astype(dtype).sum(0)
