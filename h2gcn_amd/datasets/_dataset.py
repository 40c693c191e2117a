"""Planetoid-format data (the on-disk format feeding the path) and the tensors the model consumes.

On-disk format (reference ``PlanetoidData.load_data``, ``h2gcn/datasets/_dataset.py:195-305``): seven pickles
``<name>.{x,y,tx,ty,allx,ally,graph}`` (scipy CSR features / one-hot label arrays / dict-of-neighbour-lists) plus
``<name>.test.index`` (one node id per line).  ``allx`` rows come first, then the test rows ``tx`` in the order of
``test.index``; ids missing from the contiguous test range are isolated nodes (citeseer) and become all-zero,
label-less rows that every mask excludes.  Splits: train = the first ``len(y)`` nodes, test = ``test.index``,
validation = the next ``val_size`` nodes after the training block (or everything left if fewer).

``get_tensors`` plays the role of ``getTensors`` (``:537-584``): features become a device-resident sparse operand
(a 1-hop :class:`~h2gcn_amd.hops.HopPlan`, so the ``SparseDense`` embedding runs on the same HIP kernel), the
normalised hop matrices become the ``adj_hops`` :class:`HopPlan`, labels/masks become dense tensors.
"""
from __future__ import annotations

import pickle
import warnings
from pathlib import Path
from typing import Optional, Sequence

import numpy as np
import scipy.sparse as sp

from .. import operands


def _load_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f, encoding="latin1")


def adjacency_from_neighbour_lists(graph: dict) -> sp.csr_matrix:
    """Symmetric binary float32 adjacency over nodes ``0..len(graph)-1`` (a listed self-loop gives a 1 on the
    diagonal), as ``nx.adjacency_matrix(nx.from_dict_of_lists(graph), nodelist=range(len(graph)))`` yields."""
    n = len(graph)
    src = np.fromiter((u for u, nbrs in graph.items() for _ in nbrs), dtype=np.int64)
    dst = np.fromiter((v for nbrs in graph.values() for v in nbrs), dtype=np.int64)
    rows = np.concatenate([src, dst])
    cols = np.concatenate([dst, src])
    a = sp.csr_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(n, n))
    a.data[:] = 1.0  # duplicates were summed by the constructor
    a.sort_indices()
    return a


class PlanetoidData:
    def __init__(self, dataset_str: str, dataset_path, val_size: Optional[int] = None):
        self.dataset_str = dataset_str
        self.dataset_path = str(dataset_path)
        self.val_size = val_size
        self.preprocessed_feature = False
        self._load()

    # ------------------------------------------------------------------ loading
    def _load(self):
        base = Path(self.dataset_path)
        x, y, tx, ty, allx, ally, graph = (_load_pickle(base / f"{self.dataset_str}.{n}")
                                           for n in ("x", "y", "tx", "ty", "allx", "ally", "graph"))
        test_order = [int(line.strip()) for line in open(base / f"{self.dataset_str}.test.index")]
        test_sorted = np.sort(test_order)
        lo, hi = int(test_sorted[0]), int(test_sorted[-1])
        non_valid = set()
        if hi - lo + 1 != len(test_sorted):
            # isolated nodes inside the test range: give them zero feature / label rows
            print(f"Patch for citeseer dataset is applied for dataset {self.dataset_str} at {self.dataset_path}")
            tx_full = sp.lil_matrix((hi - lo + 1, x.shape[1]))
            tx_full[test_sorted - lo, :] = tx
            ty_full = np.zeros((hi - lo + 1, y.shape[1]))
            ty_full[test_sorted - lo, :] = ty
            tx, ty = tx_full, ty_full
            non_valid = set(range(lo, hi + 1)) - set(test_sorted.tolist())
        features = sp.vstack((allx, tx)).tolil()
        features[test_order, :] = features[test_sorted, :]
        labels = np.vstack((ally, ty))
        labels[test_order, :] = labels[test_sorted, :]
        non_valid |= set(np.where(labels.sum(1) == 0)[0].tolist())

        n = labels.shape[0]
        train_mask = np.zeros(n, dtype=bool)
        train_mask[: len(y)] = True
        test_mask = np.zeros(n, dtype=bool)
        test_mask[test_sorted] = True
        val_mask = ~(train_mask | test_mask)
        if self.val_size is not None:
            if val_mask.sum() > self.val_size:
                val_mask = np.zeros(n, dtype=bool)
                val_mask[len(y): len(y) + self.val_size] = True
            else:
                print(f"Val set size set to {val_mask.sum()} due to insufficient samples.")
        for i in non_valid:
            for name, m in (("training", train_mask), ("test", test_mask), ("val", val_mask)):
                if m[i]:
                    warnings.warn(f"Non valid samples detected in {name} set")
                    m[i] = False
                    break
        self.non_valid_samples = non_valid
        self.sparse_adj = adjacency_from_neighbour_lists(graph)
        self.features = sp.csr_matrix(features)
        self.y_all = labels
        self.train_mask, self.val_mask, self.test_mask = train_mask, val_mask, test_mask
        self.y_train, self.y_val, self.y_test = (np.where(m[:, None], labels, 0.0) for m in (train_mask, val_mask, test_mask))

    # ------------------------------------------------------------------ properties the model plugin uses
    @property
    def num_samples(self) -> int:
        return self.y_all.shape[0]

    @property
    def num_labels(self) -> int:
        return self.y_all.shape[1]

    @property
    def labels(self) -> np.ndarray:
        return np.argmax(self.y_all, axis=1)

    def row_normalize_features(self):
        self.features = operands.row_normalize_features(self.features)
        self.preprocessed_feature = True

    def adj_remove_eye(self):
        self.sparse_adj = operands.remove_self_loops(self.sparse_adj)

    def set_identity_features(self):
        self.features = sp.identity(self.num_samples, dtype=np.float32, format="csr")

    # ------------------------------------------------------------------ tensors
    #: features denser than this are handed to the model as a dense matrix: X @ W0 is then GEMM-shaped work for
    #: rocBLAS / hipBLASLt (MFMA), not a gather (syn-products has F = 100 dense features)
    DENSE_FEATURE_THRESHOLD = 0.25

    def _feature_operand(self, features, device, build_transpose):
        import torch

        from ..hops import HopPlan

        f = sp.csr_matrix(features)
        density = f.nnz / max(1, f.shape[0] * f.shape[1])
        if density >= self.DENSE_FEATURE_THRESHOLD:
            return torch.from_numpy(np.asarray(f.todense(), dtype=np.float32)).to(device)
        return HopPlan.from_scipy([f], device, build_transpose=build_transpose, keep_permutation=True)  # SparseDropout

    def get_tensors(self, device, adj_norm_hops: Optional[Sequence[str]] = None, norm: str = operands.SYM_NORMALIZED,
                    build_transpose: bool = True, host_hops: bool = False, shard=None) -> dict:
        """``adj`` / ``features`` / ``adj_hops`` as device operands + dense label/mask tensors (keys as the
        reference's ``tensors`` namespace: ``H2GCN.py:66,77-79``).

        ``shard = (rank, world)``: row-partitioned run -- this rank gets rows ``[r0, r1)`` of the features, of every
        hop matrix (as :class:`~h2gcn_amd.partition.ShardedHops`) and of the labels/masks."""
        import torch

        from ..hops import HopPlan

        if shard is not None:
            return self._get_tensors_sharded(device, adj_norm_hops, norm, shard)
        t = {}
        t["features"] = self._feature_operand(self.features, device, build_transpose)
        t["adj"] = HopPlan.from_scipy([self.sparse_adj], device)
        if adj_norm_hops and host_hops:      # scipy SpGEMM on the host, as the reference does
            hops = operands.build_adj_norm_hops(self.sparse_adj, adj_norm_hops, norm)
            t["adj_hops"] = HopPlan.from_scipy(hops, device, build_transpose=build_transpose)
        elif adj_norm_hops:                  # exact-k-hop rings grown on the GPU (bit-identical operands)
            rp, ci, va, n = operands.build_adj_norm_hops_device(self.sparse_adj, adj_norm_hops, norm, device)
            t["adj_hops"] = HopPlan(rp, ci, va, n, build_transpose=build_transpose)
        else:
            t["adj_hops"] = None
        for name in ("y_all", "y_train", "y_val", "y_test"):
            t[name] = torch.from_numpy(np.asarray(getattr(self, name), dtype=np.float32)).to(device)
        for name in ("train_mask", "val_mask", "test_mask"):
            t[name] = torch.from_numpy(getattr(self, name)).to(device)
        t["labels"] = torch.from_numpy(self.labels).to(device)
        return t


def sharded_adj_hops(adj_no_self_loops, adj_norm_hops, norm, device, rank: int, world: int, gather=None,
                     balance: str = "nnz"):
    """This rank's rows of the ``adj_hops`` operands of a row-partitioned run, built WITHOUT materialising any whole ring:

    phase A  every rank counts the rows of an EQUAL share of the graph (rings below the last one are filled for that
             share, the last one only counted); the per-row ring lengths are all-gathered (``gather``: list of per-rank
             int64 ``[K+1, rows]`` tensors -> the same list on every rank; default ``torch.distributed.all_gather``);
    split    ``balance="nnz"``: prefix-sum-of-work split over rows (work = nonzeros over all hop groups + one unit per
             group for the output row), ``"rows"``: equal blocks;
    phase B  every rank builds its final row window of every ring from the whole adjacency pattern and normalises it
             (SYM needs the row lengths of arbitrary columns: those are the phase-A counts).

    Time and memory per rank are O(rings / P) instead of the O(rings) of building everything and slicing (the scaling
    wall of the reference's host path, ``_dataset.py:147-157``).  Returns ``(rowptr, colidx, vals, partition)`` with the
    column ids already in the partition's padded row space."""
    import torch

    from ..partition import RowPartition

    rp, ci, n = operands.upload_pattern(adj_no_self_loops, device)
    groups = operands.parse_adj_nhood(adj_norm_hops)
    max_hop = max(max(g) for g in groups)
    eq = RowPartition.equal(n, world)
    mine = operands.ring_row_lengths_window(rp, ci, n, max_hop, eq.rows(rank))
    if world > 1:
        if gather is None:
            import torch.distributed as dist

            def gather(t):
                padded = torch.zeros((t.shape[0], eq.per), dtype=t.dtype, device=t.device)
                padded[:, : t.shape[1]] = t
                outs = [torch.empty_like(padded) for _ in range(world)]
                dist.all_gather(outs, padded)
                return [o[:, : eq.rows(q)[1] - eq.rows(q)[0]] for q, o in enumerate(outs)]
        ring_len = torch.cat(list(gather(mine)), dim=1).contiguous()
    else:
        ring_len = mine
    if balance == "nnz" and world > 1:
        work = sum(ring_len[g].sum(dim=0) for g in groups) + len(groups)
        part = RowPartition.balanced(work.cpu().numpy(), world)
    else:
        part = eq
    rps, cis, vas = operands.build_adj_norm_hops_window(rp, ci, n, part.rows(rank), ring_len, adj_norm_hops, norm)
    cis = [part.to_padded(c) for c in cis]
    return rps, cis, vas, part


def _sharded_tensors(self, device, adj_norm_hops, norm, shard):
    import os

    import torch

    from ..hops import HopPlan
    from ..partition import RowPartition, ShardedHops

    rank, world = shard
    n = self.num_samples
    t = {"adj": None}
    if adj_norm_hops:
        rp, ci, va, part = sharded_adj_hops(self.sparse_adj, adj_norm_hops, norm, device, rank, world,
                                            balance=os.environ.get("H2GCN_PARTITION", "nnz"))
        n_src = n if part.is_equal else world * part.per
        plan = HopPlan(rp, ci, va, n_src, build_transpose=True)
        t["adj_hops"] = ShardedHops(plan, n, device, partition=part)
    else:
        part = RowPartition.equal(n, world)
        t["adj_hops"] = None
    r0, r1 = part.rows(rank)
    t["partition"] = part
    t["features"] = self._feature_operand(sp.csr_matrix(self.features)[r0:r1], device, True)
    for name in ("y_all", "y_train", "y_val", "y_test"):
        t[name] = torch.from_numpy(np.asarray(getattr(self, name), dtype=np.float32)[r0:r1]).to(device)
    for name in ("train_mask", "val_mask", "test_mask"):
        t[name] = torch.from_numpy(getattr(self, name)[r0:r1]).to(device)
    t["labels"] = torch.from_numpy(self.labels[r0:r1]).to(device)
    return t


PlanetoidData._get_tensors_sharded = _sharded_tensors


def read_generated_graph(directory, graph_name: str):
    """The graph generator's on-disk pair (reference ``experiments/h2gcn/modules/graphgen.py:37-58``):
    ``<graph_name>.graph`` = pickled ``nx.to_dict_of_lists(G)`` and ``<graph_name>.ally`` = pickled one-hot label
    array ``[n, numClass]`` (class = node colour - 1).  When only ``<graph_name>.gpickle.gz`` exists (``:60-66``, the
    pickled networkx graph with the ``color`` node attribute) it is read instead.  Returns
    ``(adjacency csr float32 [n, n], labels_onehot float [n, C])``."""
    import gzip

    base = Path(directory)
    g_file, y_file, nx_file = (base / f"{graph_name}{ext}" for ext in (".graph", ".ally", ".gpickle.gz"))
    if g_file.exists() and y_file.exists():
        graph = _load_pickle(g_file)
        ally = np.asarray(_load_pickle(y_file))
    elif nx_file.exists():
        with gzip.open(nx_file, "rb") as f:
            G = pickle.load(f)
        nodes = sorted(G.nodes())
        if nodes != list(range(len(nodes))):
            raise ValueError(f"{nx_file}: nodes must be labelled 0..n-1")
        graph = {u: list(G.adj[u]) for u in nodes}
        colours = np.array([G.nodes[u].get("color", 0) for u in nodes], dtype=np.int64)
        ally = np.zeros((len(nodes), int(colours.max())), dtype=np.float64)
        has = colours > 0
        ally[np.nonzero(has)[0], colours[has] - 1] = 1.0
    else:
        raise FileNotFoundError(f"neither {g_file} + {y_file} nor {nx_file} exists")
    if sorted(graph) != list(range(len(graph))):
        raise ValueError(f"{g_file}: nodes must be labelled 0..n-1")
    if ally.ndim != 2 or ally.shape[0] != len(graph):
        raise ValueError(f"{y_file}: label array has shape {ally.shape}, the graph has {len(graph)} nodes")
    return adjacency_from_neighbour_lists({u: graph[u] for u in range(len(graph))}), ally


class GeneratedGraphData(PlanetoidData):
    """A generator graph (``.graph`` + ``.ally``) as a dataset.  The generator stores no features and no splits -- the
    reference attaches ogbn-products features and splits in a later signac stage (``feature_generation.py``), neither
    of which exists offline -- so features are either given (``features``: ``[n, F]`` array / scipy matrix) or drawn
    class-conditionally (unit-variance Gaussians around random class centres, ``feature_seed``), and the split is a
    seeded random ``train_frac / val_frac / rest`` partition of the labelled nodes."""

    def __init__(self, graph_name: str, dataset_path, features=None, feature_dim: int = 100, feature_seed: int = 0,
                 split_seed: int = 0, train_frac: float = 0.25, val_frac: float = 0.25):
        self.dataset_str, self.dataset_path = graph_name, str(dataset_path)
        self.val_size = None
        self.preprocessed_feature = False
        adj, labels = read_generated_graph(dataset_path, graph_name)
        n = adj.shape[0]
        valid = labels.sum(1) > 0
        self.non_valid_samples = set(np.where(~valid)[0].tolist())
        if features is None:
            rng = np.random.default_rng(feature_seed)
            centres = rng.standard_normal((labels.shape[1], feature_dim))
            features = (labels @ centres + rng.standard_normal((n, feature_dim))).astype(np.float32)
        features = sp.csr_matrix(features)
        if features.shape[0] != n:
            raise ValueError(f"features have {features.shape[0]} rows, the graph has {n} nodes")
        order = np.random.default_rng(split_seed).permutation(np.nonzero(valid)[0])
        n_train, n_val = int(round(train_frac * len(order))), int(round(val_frac * len(order)))
        masks = [np.zeros(n, dtype=bool) for _ in range(3)]
        masks[0][order[:n_train]] = True
        masks[1][order[n_train:n_train + n_val]] = True
        masks[2][order[n_train + n_val:]] = True
        self.sparse_adj, self.features, self.y_all = adj, features, labels
        self.train_mask, self.val_mask, self.test_mask = masks
        self.y_train, self.y_val, self.y_test = (np.where(m[:, None], labels, 0.0) for m in masks)


def export_planetoid(path, name, adj, features, labels_onehot, n_train: int, test_ids: Sequence[int],
                     n_allx: Optional[int] = None):
    """Write a graph in the planetoid on-disk format (inverse of the loader; used to round-trip fixtures and to
    hand synthetic graphs to the entry point).  ``test_ids`` are the test nodes, in the order they should appear
    in ``test.index``; nodes inside their contiguous range that are not listed become "isolated" (citeseer case)."""
    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    adj = sp.csr_matrix(adj)
    features = sp.csr_matrix(features)
    labels_onehot = np.asarray(labels_onehot)
    test_sorted = np.sort(np.asarray(test_ids))
    n_allx = int(test_sorted[0]) if n_allx is None else n_allx
    # after loading, row test_ids[k] holds the k-th row of tx  ->  tx row k = features[test_ids[k]] ... but the loader
    # first places tx rows at the SORTED positions and then permutes: features[test_order] = features[test_sorted]
    # so tx (in file order) must be the rows of the sorted ids permuted accordingly: tx[j] = final[test_order[j]]
    # where j indexes sorted position.
    order = np.asarray(test_ids)
    tx = features[order, :]
    ty = labels_onehot[order, :]
    objs = {
        "x": features[:n_train, :], "y": labels_onehot[:n_train, :],
        "allx": features[:n_allx, :], "ally": labels_onehot[:n_allx, :],
        "tx": tx, "ty": ty,
        "graph": {int(i): [int(j) for j in adj.indices[adj.indptr[i]:adj.indptr[i + 1]]] for i in range(adj.shape[0])},
    }
    for k, v in objs.items():
        with open(path / f"{name}.{k}", "wb") as f:
            pickle.dump(v, f)
    (path / f"{name}.test.index").write_text("".join(f"{int(i)}\n" for i in test_ids))
