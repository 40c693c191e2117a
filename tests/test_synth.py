"""CPU suite: the synthetic-operand generator (bench inputs) -- numpy and torch paths are bit-identical and any
row block can be generated independently (what row-partitioned ranks rely on)."""
import numpy as np
import torch

from h2gcn_amd import synth


def test_splitmix_bits_agree():
    x = np.arange(0, 5000, dtype=np.uint64) * np.uint64(0x123456789) + np.uint64(2**63 - 5)
    a = synth.splitmix64_np(x)
    b = synth.splitmix64_torch(torch.from_numpy(x.view(np.int64)))
    assert np.array_equal(a.view(np.int64), b.numpy())
    assert int(synth.splitmix64_np(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF  # published splitmix64 KAT


def test_degrees_shape():
    n = 20000
    deg = synth.synth_degrees(n, 1_000_000, 123, n)
    assert abs(int(deg.sum()) - 1_000_000) < 0.01 * 1_000_000
    assert 0.005 < (deg == 0).mean() < 0.02
    assert deg.max() > 20 * np.median(deg[deg > 0])  # heavy tail
    assert np.array_equal(deg, synth.synth_degrees(n, 1_000_000, 123, n))


def test_numpy_torch_identical_and_block_decomposable():
    n = 3000
    deg = synth.synth_degrees(n, 60000, 124, n)
    full = synth.synth_hop_rows_np(deg, n, 124, 0, n)
    t = synth.synth_hop_rows(deg, n, 124, 0, n, "cpu")
    for a, b in zip(full, t):
        assert np.array_equal(a, b.numpy())
    rp, ci, va = full
    assert rp.dtype == np.int64 and ci.dtype == np.int32 and va.dtype == np.float32
    for r in range(0, n, 97):  # sorted, unique, in range, 1/deg values
        seg = ci[rp[r]:rp[r + 1]]
        assert np.all(np.diff(seg) > 0) and (len(seg) == 0 or (seg.min() >= 0 and seg.max() < n))
        if len(seg):
            assert np.all(va[rp[r]:rp[r + 1]] == np.float32(1) / np.float32(len(seg)))
    # a block generated alone equals the slice of the full matrix
    r0, r1 = 700, 1900
    brp, bci, bva = synth.synth_hop_rows_np(deg, n, 124, r0, r1)
    assert np.array_equal(brp, rp[r0:r1 + 1] - rp[r0])
    assert np.array_equal(bci, ci[rp[r0]:rp[r1]]) and np.array_equal(bva, va[rp[r0]:rp[r1]])


def test_features_identical_and_uniform():
    a = synth.synth_features_np(128, 125, 10, 210)
    b = synth.synth_features(128, 125, 10, 210, "cpu").numpy()
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert -1 <= a.min() and a.max() < 1 and abs(a.mean()) < 0.02
    assert np.array_equal(synth.synth_features_np(128, 125, 0, 300)[10:210], a)


def test_mixed_class_shapes_are_what_their_names_say():
    """The round-4 shapes (dispatch by segment class): h2gcn_like = a sparse hop next to a dense one, products_tail = products'
    |V| and |E| with degrees from 1, bimodal = short rows scattered among medium ones.  Degree sequences only (cheap, deterministic)."""
    like = synth.hop_degrees(synth.SHAPES["h2gcn_like"])
    assert abs(int(like[0].sum()) - 16_000_000) < 0.01 * 16_000_000 and abs(int(like[1].sum()) - 200_000_000) < 0.01 * 200_000_000
    assert (like[0] <= 16).mean() > 0.85 and (like[1] <= 16).mean() < 0.15 and like[0][like[0] > 0].min() == 1
    tail = synth.hop_degrees(synth.SHAPES["products_tail"])
    for d in tail:
        assert abs(int(d.sum()) - 120_000_000) < 0.01 * 120_000_000 and d[d > 0].min() == 1
        assert (d < 16).mean() >= 0.40 and d[d <= 16].sum() / d.sum() < 0.1 and (d >= 256).mean() > 0.02
    bi = synth.hop_degrees(synth.SHAPES["bimodal"])[0]
    assert 0.55 < (bi <= 16).mean() < 0.65 and 0.3 < bi[bi <= 16].sum() / bi.sum() < 0.37 and 17.0 < bi.mean() < 18.5
    assert np.array_equal(bi, synth.hop_degrees(synth.SHAPES["bimodal"])[0])           # identical on every rank
    assert np.array_equal(synth.hop_degrees(synth.SHAPES["products"])[0], synth.synth_degrees(2_400_000, 120_000_000, synth.SEED_A1, 2_400_000))


def test_c_restatement_of_the_generator_is_bit_identical():
    """oracle/spmm_oracle.c restates the generator so that the full-size GPU checks can rebuild EVERY row of the BASELINE shapes on
    the host in seconds (tests/test_fullsize_parity_gpu.py); pinned here against the numpy definition: whole matrices, row blocks,
    rows that collide (duplicates removed), empty rows, both degree families, features at arbitrary row windows."""
    from oracle import fullsize as fs

    assert fs.stream_key(synth.SEED_A1) == synth._stream_key(synth.SEED_A1) and fs.stream_key(7) == synth._stream_key(7)
    for n, nnz, seed, family in ((3000, 60000, 124, "pareto"), (500, 40000, 9, "pareto"), (4000, 30000, 5, "lognormal")):
        deg = synth.synth_degrees(n, nnz, seed, n) if family == "pareto" else synth.synth_degrees_lognormal(n, nnz, seed, n)
        assert (deg == 0).any()
        for r0, r1 in ((0, n), (17, n // 2), (n - 5, n), (40, 41)):
            want = synth.synth_hop_rows_np(deg, n, seed, r0, r1)
            got = fs.synth_hop_rows_c(deg, n, seed, r0, r1)
            for a, b in zip(want, got):
                assert a.dtype == b.dtype and np.array_equal(a, b), (n, family, r0, r1)
        assert len(want[1]) <= int(deg[40:41].sum())
    dense = np.full(50, 45, dtype=np.int64)                 # 45 draws out of 50 columns: collisions in every row
    a, b = synth.synth_hop_rows_np(dense, 50, 3, 0, 50), fs.synth_hop_rows_c(dense, 50, 3, 0, 50)
    assert len(a[1]) < 45 * 50 and all(np.array_equal(p, q) for p, q in zip(a, b))
    for d, r0, r1 in ((128, 0, 700), (67, 5, 4000), (1, 10 ** 9, 10 ** 9 + 50)):
        assert np.array_equal(synth.synth_features_np(d, synth.SEED_X, r0, r1), fs.synth_features_c(d, synth.SEED_X, r0, r1))
