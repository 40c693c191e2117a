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
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
