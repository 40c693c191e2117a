"""Synthetic hop matrices and features of the BASELINE.json shapes, generated on the device.

The reference has no benchmark inputs; its own generator (``experiments/h2gcn/modules/graphgen.py``) is an
O(n^2) preferential-attachment process that cannot reach the ogbn-arxiv / ogbn-products shapes.  This module
defines the synthetic CSR those configs are measured on:

* row degrees: clipped Pareto(alpha = 1.5) (raw sample ``1 + pareto``, clipped at ``clip``), about 1 % of the
  rows emptied, rescaled so that the total is ``nnz_target`` (numpy ``PCG64(seed)``; identical on every rank);
* column ids: counter-based -- edge ``e`` (global running index of the hop) gets ``splitmix64(seed, e) mod
  n_cols``, so ANY row block can be generated independently (row-partitioned multi-GPU runs generate only their
  shard); per row the ids are sorted and de-duplicated (the canonical order ``tf.sparse.reorder`` gives,
  reference ``h2gcn/datasets/_dataset.py:535``), which removes ~1e-5 of the edges;
* values: ``1 / deg(row)`` in fp32 -- row-normalised (``RW_NORMALIZED``, reference ``_dataset.py:119-123``), the
  normalisation BASELINE.json names for the synthetic shapes;
* features: ``U[-1, 1)`` on a 2^-23 grid, counter-based as well.

The numpy and torch implementations are bit-identical (tests/test_synth.py), so the CPU oracle can rebuild any
rows of the operands the GPU generated.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

_GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB


def _to_i64(u: int) -> int:
    u &= (1 << 64) - 1
    return u - (1 << 64) if u >= (1 << 63) else u


def splitmix64_np(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + np.uint64(_GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def _lsr(z: torch.Tensor, k: int) -> torch.Tensor:
    """logical shift right on int64 (torch's >> is arithmetic)."""
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64_torch(x: torch.Tensor) -> torch.Tensor:
    """Same bits as :func:`splitmix64_np`, on int64 tensors (two's-complement wrapping multiply)."""
    z = x + _to_i64(_GAMMA)
    z = (z ^ _lsr(z, 30)) * _to_i64(_M1)
    z = (z ^ _lsr(z, 27)) * _to_i64(_M2)
    return z ^ _lsr(z, 31)


def _stream_key(seed: int) -> int:
    """Per-stream offset so that different seeds give unrelated sequences."""
    return int(splitmix64_np(np.array([seed], dtype=np.uint64))[0])


def synth_degrees(n_rows: int, nnz_target: int, seed: int, n_cols: int, alpha: float = 1.5,
                  empty_frac: float = 0.01, clip: float = 1000.0) -> np.ndarray:
    """Raw (pre-dedup) degree of every row, int64 [n_rows]; sum ~= nnz_target."""
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = np.minimum(1.0 + rng.pareto(alpha, n_rows), clip)
    raw[rng.random(n_rows) < empty_frac] = 0.0
    total = raw.sum()
    if total <= 0 or nnz_target <= 0:
        return np.zeros(n_rows, dtype=np.int64)
    deg = np.floor(raw * (nnz_target / total) + 0.5).astype(np.int64)
    deg[(raw > 0) & (deg < 1)] = 1
    return np.minimum(deg, n_cols)


def synth_degrees_lognormal(n_rows: int, nnz_target: int, seed: int, n_cols: int, sigma: float = 1.35,
                              empty_frac: float = 0.01) -> np.ndarray:
    """Raw degrees with a long LOW-degree body: ``exp(sigma * N(0, 1))`` scaled to the target total, rounded, non-empty rows
    at least 1 -- degrees start at 1 (the rescaled Pareto of :func:`synth_degrees` has a minimum of ~nnz/3n), the bulk of
    the ROWS is short and the bulk of the EDGES sits in medium / long rows, which is how real co-purchase / citation
    degree sequences look (sigma = 1.35 at mean 50: median 20, ~43 % of the rows below 16, ~3 % at 256 or more)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = np.exp(sigma * rng.standard_normal(n_rows))
    raw[rng.random(n_rows) < empty_frac] = 0.0
    total = raw.sum()
    if total <= 0 or nnz_target <= 0:
        return np.zeros(n_rows, dtype=np.int64)
    deg = np.floor(raw * (nnz_target / total) + 0.5).astype(np.int64)
    deg[(raw > 0) & (deg < 1)] = 1
    return np.minimum(deg, n_cols)


def synth_degrees_mix(n_rows: int, seed: int, n_cols: int, mix, jitter: int = 2, empty_frac: float = 0.01) -> np.ndarray:
    """Raw degrees drawn from a discrete mix ``[(fraction, degree), ...]`` (+- ``jitter``), rows in random order: short rows
    scattered among medium ones at a prescribed share (stress shape for the dispatch by segment class)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fr = np.array([f for f, _ in mix], dtype=np.float64)
    which = rng.choice(len(mix), size=n_rows, p=fr / fr.sum())
    deg = np.array([d for _, d in mix], dtype=np.int64)[which] + rng.integers(-jitter, jitter + 1, n_rows)
    deg = np.maximum(deg, 1)
    deg[rng.random(n_rows) < empty_frac] = 0
    return np.minimum(deg, n_cols)


def hop_degrees(cfg: dict, seeds=None) -> list:
    """Raw degree sequences of the hop matrices of one entry of :data:`SHAPES` (identical on every rank)."""
    seeds = seeds or (SEED_A1, SEED_A2)
    nnz = cfg["nnz_per_hop"]
    nnz = list(nnz) if isinstance(nnz, (list, tuple)) else [nnz] * len(seeds)
    spec = cfg.get("degrees")
    out = []
    for k, s in enumerate(seeds):
        if spec is None:
            out.append(synth_degrees(cfg["n"], nnz[k], s, cfg["n"]))
        else:
            sp = spec[k] if isinstance(spec, (list, tuple)) else spec
            if "mix" in sp:
                out.append(synth_degrees_mix(cfg["n"], s, cfg["n"], sp["mix"]))
            else:
                out.append(synth_degrees_lognormal(cfg["n"], nnz[k], s, cfg["n"], sigma=sp["sigma"]))
    return out


def synth_hop_rows_np(raw_deg: np.ndarray, n_cols: int, seed: int, r0: int, r1: int
                      ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """CSR of rows [r0, r1) (local row pointers), numpy: (rowptr int64, colidx int32, vals float32)."""
    raw_ptr = np.concatenate([[0], np.cumsum(raw_deg)]).astype(np.int64)
    e0, e1 = int(raw_ptr[r0]), int(raw_ptr[r1])
    e = np.arange(e0, e1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64_np(e + np.uint64(_stream_key(seed)))
    col = ((h >> np.uint64(1)) % np.uint64(n_cols)).astype(np.int64)
    row = np.repeat(np.arange(r0, r1, dtype=np.int64), raw_deg[r0:r1])
    key = np.unique(row * n_cols + col)
    row_u, col_u = key // n_cols, key % n_cols
    counts = np.bincount(row_u - r0, minlength=r1 - r0).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    with np.errstate(divide="ignore"):
        inv = (np.float32(1.0) / counts.astype(np.float32)).astype(np.float32)
    vals = inv[row_u - r0]
    return rowptr, col_u.astype(np.int32), vals.astype(np.float32)


def synth_hop_rows(raw_deg: np.ndarray, n_cols: int, seed: int, r0: int, r1: int, device
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Same CSR as :func:`synth_hop_rows_np`, built with torch ops on ``device``."""
    raw_ptr = np.concatenate([[0], np.cumsum(raw_deg)]).astype(np.int64)
    e0, e1 = int(raw_ptr[r0]), int(raw_ptr[r1])
    n_local = r1 - r0
    if e1 == e0:
        return (torch.zeros(n_local + 1, dtype=torch.int64, device=device),
                torch.zeros(0, dtype=torch.int32, device=device), torch.zeros(0, dtype=torch.float32, device=device))
    e = torch.arange(e0, e1, dtype=torch.int64, device=device)
    h = splitmix64_torch(e + _to_i64(_stream_key(seed)))
    del e
    col = _lsr(h, 1) % n_cols
    del h
    deg_t = torch.from_numpy(raw_deg[r0:r1].astype(np.int64)).to(device)
    row = torch.repeat_interleave(torch.arange(r0, r1, dtype=torch.int64, device=device), deg_t)
    key = row * n_cols + col
    del row, col
    key = torch.unique(key, sorted=True)
    row_u = torch.div(key, n_cols, rounding_mode="floor")
    col_u = (key - row_u * n_cols).to(torch.int32)
    del key
    row_l = row_u - r0
    del row_u
    counts = torch.bincount(row_l, minlength=n_local)
    rowptr = torch.zeros(n_local + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(counts, 0)
    inv = torch.ones((), dtype=torch.float32, device=device) / counts.to(torch.float32)
    vals = inv[row_l].contiguous()
    return rowptr, col_u.contiguous(), vals


def synth_features_np(d: int, seed: int, r0: int, r1: int) -> np.ndarray:
    idx = np.arange(r0 * d, r1 * d, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64_np(idx + np.uint64(_stream_key(seed)))
    k = (h >> np.uint64(40)).astype(np.int64)  # 24 random bits
    return (k.astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)).reshape(r1 - r0, d)


def synth_features(d: int, seed: int, r0: int, r1: int, device) -> torch.Tensor:
    idx = torch.arange(r0 * d, r1 * d, dtype=torch.int64, device=device)
    h = splitmix64_torch(idx + _to_i64(_stream_key(seed)))
    k = _lsr(h, 40)
    return (k.to(torch.float32) * (2.0 ** -23) - 1.0).reshape(r1 - r0, d)


#: the synthetic shapes BASELINE.json names (config index -> parameters); nnz is PER HOP
SHAPES = {
    "arxiv": dict(n=170_000, nnz_per_hop=1_200_000, d=128),       # configs[2]
    "products": dict(n=2_400_000, nnz_per_hop=120_000_000, d=128),  # configs[3], configs[4]
    # not a BASELINE config: a low-degree stress shape (mean degree ~4, like Cora's 1-hop) for the short-row path
    "lowdeg": dict(n=8_000_000, nnz_per_hop=32_000_000, d=128),
    # not BASELINE configs: gather working sets far beyond the 256 MiB Infinity Cache, to separate what HBM delivers
    # from what the cache adds.  X = 8.2 GB (4.1 GB per 64-column slice):
    "hbm16m": dict(n=16_000_000, nnz_per_hop=120_000_000, d=128),      # products' edge count, mean degree 7.5
    "products_x6": dict(n=16_000_000, nnz_per_hop=800_000_000, d=128),  # products' degree distribution (mean 50)
    # not BASELINE configs: MIXED segment classes in one launch (round 4: binned short-segment list next to the tile walk).
    # The reference's own operands are bimodal by construction -- the exact-2-hop ring is ~8x denser than the 1-hop one
    # (h2gcn/datasets/_dataset.py:138-158; Cora: mean 3.9 / 31.9) -- and real degree sequences start at 1:
    "h2gcn_like": dict(n=2_000_000, nnz_per_hop=[16_000_000, 200_000_000], d=128,
                       degrees=[dict(sigma=1.0), dict(sigma=1.0)]),       # A1 mean 8 (median 5), A2 mean 100 (median 60)
    "products_tail": dict(n=2_400_000, nnz_per_hop=120_000_000, d=128,
                          degrees=dict(sigma=1.35)),                      # products' |V|, |E|; degrees from 1, ~43 % of rows < 16
    # 60 % of the rows with ~10 nonzeros scattered among 40 % with ~30: mean 18 (a "wave per segment" launch by the pooled
    # mean) although a third of the EDGES sits in short segments; X = 4.1 GB
    "bimodal": dict(n=8_000_000, nnz_per_hop=142_500_000, d=128, degrees=dict(mix=[(0.6, 10), (0.4, 30)])),
}
SEED_A1, SEED_A2, SEED_X = 123, 124, 125
