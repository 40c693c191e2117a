"""CPU suite: the N > 1 path -- row partitioning + per-layer all-gather -- with world_size-2/3 gloo processes.
The SpMM itself is replaced by the oracle here (no GPU); what is under test is the partition logic and the
collective plumbing that bench.py and the multi-GPU path use."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def test_block_bounds_cover_rows_exactly():
    from h2gcn_amd.partition import block_bounds, rows_per_rank

    for n in (0, 1, 7, 8, 9, 2_400_000, 170_000):
        for P in (1, 2, 3, 4, 8):
            b = [block_bounds(n, P, p) for p in range(P)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(P - 1))
            assert all(0 <= r1 - r0 <= rows_per_rank(n, P) for r0, r1 in b)
    with pytest.raises(ValueError):
        block_bounds(10, 2, 2)


def _free_port():
    """A rendezvous port below the kernel's ephemeral range (see bench_supervisor._free_port)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from bench_supervisor import _free_port as pick
    return pick()


def _worker(rank, world, port, n, d, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scipy.sparse as sp

        from h2gcn_amd import synth
        from h2gcn_amd.partition import EmbeddingAllGather, block_bounds, shard_rows_scipy
        from oracle import gcn_layer as og

        r0, r1 = block_bounds(n, world, rank)
        degs = [synth.synth_degrees(n, 20 * n, s, n) for s in (1, 2)]
        # each rank generates ONLY its own shard of operands and features
        shard = []
        for k, s in enumerate((1, 2)):
            rp, ci, va = synth.synth_hop_rows_np(degs[k], n, s, r0, r1)
            shard.append(sp.csr_matrix((va, ci, rp), shape=(r1 - r0, n)))
        x_local = torch.from_numpy(synth.synth_features_np(d, 3, r0, r1))
        ag = EmbeddingAllGather(n, d, "cpu")
        x_full = ag.gather(x_local)
        assert x_full.shape == (n, d)
        assert np.array_equal(x_full.numpy(), synth.synth_features_np(d, 3, 0, n))  # gathered == global
        y_local = og.gcn_layer_c(shard, x_full.numpy())
        np.save(Path(out_dir) / f"y{rank}.npy", y_local)

        # the pipelined layer (feature chunks, per-chunk all-gather) with the oracle standing in for the HIP plan
        from h2gcn_amd.partition import PipelinedHopAggregation

        class OraclePlan:
            n_rows, n_cols, n_hops = r1 - r0, n, 2

            @staticmethod
            def n_selected(hops):
                return 2 if hops is None else len(hops)

            @staticmethod
            def spmm(x, hops=None, out=None):
                sel = shard if hops is None else [shard[h] for h in hops]
                y = torch.from_numpy(og.gcn_layer_c(sel, x.contiguous().numpy()))
                return y if out is None else out.copy_(y)

            @staticmethod
            def spmm_t(grad, hops=None):
                sel = shard if hops is None else [shard[h] for h in hops]
                return torch.from_numpy(og.gcn_layer_grad_c(sel, grad.contiguous().numpy(), n))

        for chunks in (1, 2, 4, [4, 4, 8]):
            for exchange in ("allgather", "p2p"):
                layer = PipelinedHopAggregation(OraclePlan, n, d, chunks, "cpu", exchange=exchange)
                y_pipe = layer(x_local)
                assert y_pipe.shape == (r1 - r0, 2, d)
                assert np.array_equal(y_pipe.numpy(), y_local), (chunks, exchange)

        # hop filters (GCNLayer(hops=...), reference _layers.py:57-59,80-81) and the concat-free propagation on shards
        from h2gcn_amd.partition import ShardedHops, sharded_hop_spmm

        sh = ShardedHops(OraclePlan, n, "cpu", chunk_cols=4)
        assert np.array_equal(sh.aggregate(x_local, hops=[1]).numpy(), y_local[:, 1:2])
        assert np.array_equal(sh.aggregate(x_local, hops={0, 1, 5}).numpy(), y_local)
        xa = x_local.clone().requires_grad_(True)
        (sh.aggregate(xa, hops=[0]) * 2.0).sum().backward()
        np.save(Path(out_dir) / f"dx_hop0_{rank}.npy", xa.grad.numpy())
        xf = x_local.clone().requires_grad_(True)
        buf = sh.fused_propagation(xf, 2)                                  # [r2 | r0 | r1] of this rank's rows
        xg = x_local.clone().requires_grad_(True)
        r1_ = sh.aggregate(xg).flatten(1)
        r2_ = sh.aggregate(r1_).flatten(1)
        ref = torch.cat([r2_, xg, r1_], 1)
        assert buf.shape == (r1 - r0, 7 * d) and torch.equal(buf, ref)
        wt = torch.from_numpy(synth.synth_features_np(7 * d, 21, r0, r1))
        (buf * wt).sum().backward()
        (ref * wt).sum().backward()
        assert (xf.grad - xg.grad).abs().max().item() <= 1e-5

        # distributed backward: adjoint on the shard + reduce-scatter == rows [r0, r1) of the global adjoint

        w_full = synth.synth_features_np(2 * d, 9, 0, n).reshape(n, 2, d)
        xl = x_local.clone().requires_grad_(True)
        layer = PipelinedHopAggregation(OraclePlan, n, d, 2, "cpu")
        (sharded_hop_spmm(layer, xl) * torch.from_numpy(w_full[r0:r1])).sum().backward()
        np.save(Path(out_dir) / f"dx{rank}.npy", xl.grad.numpy())
        if rank == 0:
            full_ops = []
            for k, s in enumerate((1, 2)):
                rp, ci, va = synth.synth_hop_rows_np(degs[k], n, s, 0, n)
                full_ops.append(sp.csr_matrix((va, ci, rp), shape=(n, n)))
            np.save(Path(out_dir) / "dx_full.npy", og.gcn_layer_grad_c(full_ops, w_full, n))
            np.save(Path(out_dir) / "dx_hop0_full.npy", og.gcn_layer_grad_c(full_ops[:1], np.full((n, 1, d), 2.0, dtype=np.float32), n))
        if rank == 0:  # single-process answer on the unpartitioned operands
            full = []
            for k, s in enumerate((1, 2)):
                rp, ci, va = synth.synth_hop_rows_np(degs[k], n, s, 0, n)
                full.append(sp.csr_matrix((va, ci, rp), shape=(n, n)))
            np.save(Path(out_dir) / "full.npy", og.gcn_layer_c(full, x_full.numpy()))
            assert all(m.shape[0] == block_bounds(n, world, p)[1] - block_bounds(n, world, p)[0]
                       for p in range(world) for m in shard_rows_scipy(full, world, p)[:1])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1000), (3, 1001), (8, 1003)])
def test_row_partition_allgather_equals_single_rank(world, n, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, 16, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"y{r}.npy") for r in range(world)]
    full = np.load(tmp_path / "full.npy")
    assert np.array_equal(np.concatenate(parts, 0), full)  # bit-for-bit: partitioning never changes arithmetic
    dx = np.concatenate([np.load(tmp_path / f"dx{r}.npy") for r in range(world)], 0)
    want = np.load(tmp_path / "dx_full.npy")
    assert dx.shape == want.shape and np.abs(dx - want).max() <= 1e-5  # sum over ranks re-associates the adjoint
    dx0 = np.concatenate([np.load(tmp_path / f"dx_hop0_{r}.npy") for r in range(world)], 0)
    assert np.abs(dx0 - np.load(tmp_path / "dx_hop0_full.npy")).max() <= 1e-5   # hop-filtered backward


# ----------------------------------------------------------------------------- nnz-balanced partition (real graphs)
def _pareto_unshuffled_degrees(n, mean, seed):
    """Power-law degrees in DESCENDING order: the worst case for equal-row blocks (hubs first, as crawled graphs and
    the planetoid files tend to be), SURVEY.md 8(e) "for real graphs use prefix-sum-of-nnz split"."""
    rng = np.random.default_rng(seed)
    d = np.minimum((rng.pareto(1.5, n) + 1.0) * mean / 3.0, n - 1).astype(np.int64)
    return np.sort(d)[::-1].copy()


def test_balanced_partition_on_power_law_rows():
    from h2gcn_amd.partition import RowPartition

    n = 200_000
    work = _pareto_unshuffled_degrees(n, 50, 1) + _pareto_unshuffled_degrees(n, 50, 2) + 2   # two hops + 2 output rows
    for P in (2, 3, 8):
        eq = RowPartition.equal(n, P)
        bal = RowPartition.balanced(work, P)
        assert bal.world == P and bal.n == n and bal.bounds[0] == 0 and bal.bounds[-1] == n
        assert all(b1 >= b0 for b0, b1 in zip(bal.bounds, bal.bounds[1:]))
        assert bal.imbalance(work) <= 1.05, (P, bal.imbalance(work))
        assert eq.imbalance(work) > 1.5                                   # equal rows: the first block holds the hubs
        assert eq.is_equal and not bal.is_equal and bal.per >= -(-n // P)
    # padded row space: rank q's rows land at [q * per, q * per + rows_q); equal blocks: identity
    bal = RowPartition.balanced(work, 8)
    cols = torch.tensor([0, bal.bounds[1] - 1, bal.bounds[1], bal.bounds[5] + 7, n - 1], dtype=torch.int32)
    got = bal.to_padded(cols).tolist()
    want = [0, bal.bounds[1] - 1, bal.per, 5 * bal.per + 7, 7 * bal.per + (n - 1 - bal.bounds[7])]
    assert got == want
    assert RowPartition.equal(n, 8).to_padded(cols) is cols
    # degenerate inputs
    assert RowPartition.balanced(np.zeros(10), 4).is_equal
    assert RowPartition.balanced(np.array([5.0]), 3).bounds[-1] == 1
    one_hub = np.ones(100)
    one_hub[50] = 1000.0
    p = RowPartition.balanced(one_hub, 4)
    assert p.bounds[0] == 0 and p.bounds[-1] == 100 and all(b1 >= b0 for b0, b1 in zip(p.bounds, p.bounds[1:]))
    with pytest.raises(ValueError):
        RowPartition([0, 5, 3])


def _worker_balanced(rank, world, port, n, d, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scipy.sparse as sp

        from h2gcn_amd.partition import PipelinedHopAggregation, RowPartition, ShardedHops, sharded_hop_spmm
        from oracle import gcn_layer as og

        rng = np.random.default_rng(5)
        hops = []
        for k in range(2):
            deg = _pareto_unshuffled_degrees(n, 12, k + 1)
            rows = np.repeat(np.arange(n), deg)
            cols = np.concatenate([rng.choice(n, int(kk), replace=False) for kk in deg])
            m = sp.csr_matrix((rng.uniform(-1, 1, len(rows)).astype(np.float32), (rows, cols)), shape=(n, n))
            m.sort_indices()
            hops.append(m)
        x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
        w = rng.uniform(-1, 1, (n, 2, d)).astype(np.float32)
        work = sum(np.diff(m.indptr) for m in hops) + 2
        part = RowPartition.balanced(work, world)
        assert part.imbalance(work) < RowPartition.equal(n, world).imbalance(work)
        r0, r1 = part.rows(rank)
        n_pad = world * part.per
        shard = []
        for m in hops:   # this rank's rows, column ids moved into the padded row space
            s = m[r0:r1].tocsr()
            ci = part.to_padded(torch.from_numpy(s.indices.astype(np.int32))).numpy()
            s2 = sp.csr_matrix((s.data, ci, s.indptr), shape=(r1 - r0, n_pad))
            s2.sort_indices()
            shard.append(s2)

        class OraclePlan:
            n_rows, n_cols, n_hops = r1 - r0, n_pad, 2

            @staticmethod
            def n_selected(hops_):
                return 2 if hops_ is None else len(hops_)

            @staticmethod
            def spmm(xx, hops=None, out=None):
                sel = shard if hops is None else [shard[h] for h in hops]
                y = torch.from_numpy(og.gcn_layer_c(sel, xx.contiguous().numpy()))
                return y if out is None else out.copy_(y)

            @staticmethod
            def spmm_t(grad, hops=None):
                sel = shard if hops is None else [shard[h] for h in hops]
                return torch.from_numpy(og.gcn_layer_grad_c(sel, grad.contiguous().numpy(), n_pad))

        x_local = torch.from_numpy(x[r0:r1])
        for chunks in (1, 2):
            for exchange in ("allgather", "p2p"):
                layer = PipelinedHopAggregation(OraclePlan, n, d, chunks, "cpu", exchange=exchange, partition=part)
                np.save(Path(out_dir) / f"y_{chunks}_{exchange}_{rank}.npy", layer(x_local).numpy())
        sh = ShardedHops(OraclePlan, n, "cpu", chunk_cols=4, partition=part)
        xl = x_local.clone().requires_grad_(True)
        (sh.aggregate(xl) * torch.from_numpy(w[r0:r1])).sum().backward()
        np.save(Path(out_dir) / f"dx{rank}.npy", xl.grad.numpy())
        xf = x_local.clone().requires_grad_(True)
        buf = sh.fused_propagation(xf, 2)
        np.save(Path(out_dir) / f"buf{rank}.npy", buf.detach().numpy())
        if rank == 0:
            y = og.gcn_layer_c(hops, x)
            np.save(Path(out_dir) / "full.npy", y)
            np.save(Path(out_dir) / "dx_full.npy", og.gcn_layer_grad_c(hops, w, n))
            r1_ = y.reshape(n, 2 * d)
            r2_ = og.gcn_layer_c(hops, r1_).reshape(n, 4 * d)
            np.save(Path(out_dir) / "buf_full.npy", np.concatenate([r2_, x, r1_], 1))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_nnz_balanced_partition_equals_single_rank(world, tmp_path):
    """Unequal row blocks + padded row space through the whole N > 1 path (both RCCL-style exchanges, chunked, autograd,
    concat-free propagation): the result is still the single-rank result bit for bit."""
    n, d = 900, 8
    port = _free_port()
    mp.spawn(_worker_balanced, args=(world, port, n, d, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    for chunks in (1, 2):
        for exchange in ("allgather", "p2p"):
            y = np.concatenate([np.load(tmp_path / f"y_{chunks}_{exchange}_{r}.npy") for r in range(world)], 0)
            assert np.array_equal(y, full), (chunks, exchange)
    dx = np.concatenate([np.load(tmp_path / f"dx{r}.npy") for r in range(world)], 0)
    assert np.abs(dx - np.load(tmp_path / "dx_full.npy")).max() <= 2e-5
    buf = np.concatenate([np.load(tmp_path / f"buf{r}.npy") for r in range(world)], 0)
    assert np.array_equal(buf, np.load(tmp_path / "buf_full.npy"))
