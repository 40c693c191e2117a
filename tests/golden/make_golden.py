#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by IMPORTING THE REFERENCE'S OWN CODE.

Runs only in the build container (it needs /root/reference, which never travels to the GPU box); the fixtures
it writes are committed, this script is committed as their provenance.  Nothing from the reference is copied:
its modules are imported in place and executed, and only inputs/outputs (data) are stored.

    python tests/golden/make_golden.py            # cora, citeseer, dsl  (seconds)
    python tests/golden/make_golden.py --syn      # + the syn-products-shaped graph (~1-2 min, O(n^2) generator)

Shims needed to import the reference on this image:
* `scipy.sparse.linalg.eigen.arpack` no longer exists      -> fake module exposing eigsh
* `np.bool` was removed from numpy                         -> alias to bool
* `nx.from_dict_of_lists/adjacency_matrix` still exist; `nx.Graph.node`, `G.selfloop_edges` (generator) -> aliases
* `tensorflow` is absent; only the pure-python DSL parser is used from `h2gcn/models/__init__.py`
  -> a dummy `tensorflow` module is placed in sys.modules for that import.

Fixtures
* cora_operands.npz      reference loader + preprocessing on ind.cora: raw adjacency, row-normalised features,
                         labels/masks, and the SYM- and RW-normalised exact-1-hop / exact-2-hop matrices.
* citeseer_operands.npz  same for ind.citeseer (isolated nodes -> rows with no neighbours: the inf->0 branch).
* dsl_parse.json         parse_network_setup() output for the network strings used by the reference configs.
* cora_layer_outputs.npz r1 / r2 of H2GCN-2 on the golden Cora operands (row subset + fp64 column sums), via scipy.
* syn_products.npz       one graph from the reference generator (n=10000, 10 classes, m=6, h=0.2) as CSR.
* generated/syn_small.{graph,ally,gpickle.gz} + syn_small_expected.npz
                         FILES WRITTEN BY THE REFERENCE'S GENERATOR itself (graphgen.py save_graph / save_y /
                         save_nx_graph) for one n=400, 5-class, h=0.3 graph, plus the adjacency the reference's loader
                         helper (`PlanetoidData.graphDict2Adj`) derives from the .graph file.
* glue_cora.npz/.json    the reference's OWN `_layers.py` / `H2GCN.py` / `_metrics.py` executed under a numpy/scipy
                         stand-in for TensorFlow (tests/golden/_tf_standin.py) on the golden Cora operands with
                         regenerable weights (conftest.golden_weight): logits, every tagged activation (row subset +
                         fp64 column sums), loss / accuracy / L2 term.  Pins the interpreter GLUE (stack axis, hop
                         filter, concat order, slices, tag store, bias/activation order) -- not TF's kernels.
"""
import argparse
import json
import sys
import types
from pathlib import Path

import numpy as np
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def _shim_for_dataset_module():
    import scipy.sparse.linalg as spla

    if not hasattr(np, "bool"):
        np.bool = bool  # noqa: NPY001
    for name in ("scipy.sparse.linalg.eigen", "scipy.sparse.linalg.eigen.arpack"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.eigsh = spla.eigsh
            sys.modules[name] = m
    import networkx as nx

    if not hasattr(nx, "from_scipy_sparse_matrix") and hasattr(nx, "from_scipy_sparse_array"):
        nx.from_scipy_sparse_matrix = nx.from_scipy_sparse_array


def _import_reference_dataset():
    _shim_for_dataset_module()
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_dataset", REF / "h2gcn/datasets/_dataset.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _csr_fields(prefix, m, dtype):
    m = sp.csr_matrix(m)
    m.sort_indices()
    return {
        f"{prefix}_indptr": m.indptr.astype(np.int64),
        f"{prefix}_indices": m.indices.astype(np.int32),
        f"{prefix}_data": m.data.astype(dtype),
    }


def planetoid_fixture(ref, name):
    data = ref.PlanetoidData(name, str(REF / "baselines/gcn/gcn/data"), val_size=500)
    out = {}
    raw_adj = data.sparse_adj.copy()
    raw_feat = sp.csr_matrix(data.features)
    out.update(_csr_fields("adj_raw", raw_adj, np.float32))
    out.update(_csr_fields("feat_raw", raw_feat, np.float32))
    # dtype the reference's loader ends up with (float32 for cora; float64 for citeseer, whose isolated-node
    # patch stacks a float64 lil_matrix, _dataset.py:226-242) -- row normalisation happens in THAT precision
    out["feat_raw_dtype"] = np.array(str(raw_feat.dtype))
    out["adj_raw_dtype"] = np.array(str(raw_adj.dtype))
    # --- the reference's preprocessing_data order (h2gcn/models/H2GCN.py:46-54)
    data.row_normalize_features()
    data.adj_remove_eye()
    fr = _csr_fields("feat_rownorm", sp.csr_matrix(data.features), np.float32)
    assert np.array_equal(fr["feat_rownorm_indices"], out["feat_raw_indices"])
    out["feat_rownorm_data"] = fr["feat_rownorm_data"]
    adj = data.sparse_adj
    out.update(_csr_fields("adj_noeye", adj, np.float32))
    T = ref.TransformSPAdj
    splits = T.nhoodSplit(adj, 2)
    out["split_nnz"] = np.array([sp.csr_matrix(s).nnz for s in splits], dtype=np.int64)
    for ntype, tag in ((T.NType.SYM_NORMALIZED, "sym"), (T.NType.RW_NORMALIZED, "rw")):
        for k in (1, 2):
            with np.errstate(divide="ignore"):
                nm = T.normalize(splits[k], ntype)
            # stored with the fp32 cast sparse2Tensor applies (:528-535); RW shares SYM's structure
            f = _csr_fields(f"hop{k}_{tag}", nm, np.float32)
            if tag == "rw":
                assert np.array_equal(f[f"hop{k}_rw_indices"], out[f"hop{k}_sym_indices"])
                f = {f"hop{k}_rw_data": f[f"hop{k}_rw_data"]}
            out.update(f)
    # merged group "0,1" (self + 1-hop) exercises the group-summing glue of getTensors (:560-572)
    with np.errstate(divide="ignore"):
        merged = T.normalize(sum([splits[0], splits[1]]), T.NType.SYM_NORMALIZED)
    out.update(_csr_fields("hop01_sym", merged, np.float32))
    out["y_all"] = np.asarray(data.y_all).astype(np.int8)
    for m in ("train_mask", "val_mask", "test_mask"):
        out[m] = np.asarray(getattr(data, m)).astype(bool)
    out["num_labels"] = np.int64(data.num_labels)
    np.savez_compressed(HERE / f"{name.split('.')[-1]}_operands.npz", **out)
    print(name, "nnz per split", out["split_nnz"], "bytes", (HERE / f"{name.split('.')[-1]}_operands.npz").stat().st_size)


def layer_output_fixture():
    """r1 = [A1 X | A2 X], r2 = [A1 r1 | A2 r1] on the golden Cora SYM operands for
    X = PCG64(123) U(-1,1) [2708, 64] (regenerated by the tests, not stored): fp32 rows of a fixed subset (first,
    last, max-degree row, empty 2-hop rows, a few more) + fp64 column sums of all rows.  Computed with scipy's
    csr @ dense directly (the loop nest TF's CPU kernel has), independently of oracle/."""
    z = np.load(HERE / "cora_operands.npz")
    n = len(z["hop1_sym_indptr"]) - 1
    hops = [sp.csr_matrix((z[f"hop{k}_sym_data"], z[f"hop{k}_sym_indices"], z[f"hop{k}_sym_indptr"]), shape=(n, n)) for k in (1, 2)]
    x = np.random.Generator(np.random.PCG64(123)).uniform(-1, 1, (n, 64)).astype(np.float32)
    r1 = np.concatenate([h.astype(np.float32) @ x for h in hops], 1)
    r2 = np.concatenate([h.astype(np.float32) @ r1 for h in hops], 1)
    r1_64 = np.concatenate([h.astype(np.float64) @ x.astype(np.float64) for h in hops], 1)
    r2_64 = np.concatenate([h.astype(np.float64) @ r1_64 for h in hops], 1)
    deg1, deg2 = np.diff(hops[0].indptr), np.diff(hops[1].indptr)
    rows = sorted(set([0, 1, n - 1, int(deg1.argmax()), int(deg2.argmax())] + np.where(deg2 == 0)[0][:8].tolist()
                      + list(range(100, 2700, 53))))
    np.savez_compressed(HERE / "cora_layer_outputs.npz", rows=np.array(rows), r1_rows=r1[rows], r2_rows=r2[rows],
                        r1_colsum64=r1_64.sum(0), r2_colsum64=r2_64.sum(0))
    print("layer outputs: rows", len(rows))


def dsl_fixture():
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    # `from modules import logger` inside models/__init__.py needs h2gcn/ on the path and a stub logger
    stub_modules = types.ModuleType("modules")
    stub_modules.logger = types.ModuleType("modules.logger")
    sys.modules["modules"] = stub_modules
    sys.modules["modules.logger"] = stub_modules.logger
    import importlib.util

    pkg_dir = REF / "h2gcn/models"
    # _layers.py needs real TF at import; give the package a stub `_layers`
    pkg = types.ModuleType("ref_models")
    pkg.__path__ = [str(pkg_dir)]
    sys.modules["ref_models"] = pkg
    sys.modules["ref_models._layers"] = types.ModuleType("ref_models._layers")
    spec = importlib.util.spec_from_file_location("ref_models", pkg_dir / "__init__.py", submodule_search_locations=[str(pkg_dir)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_models"] = mod
    spec.loader.exec_module(mod)

    strings = [
        "M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO",   # H2GCN-2 (default, h2gcn/models/H2GCN.py:12)
        "M64-R-T1-G-V-C1-D0.5-MO",             # H2GCN-1
        "M64-R-D0.5-MO",                       # MLP
        "I-T1-G-V-C1-M64-R-D0.5-MO",           # experimental variant used in configs
        "F64-R-E-D-FO",                        # bias layers, embedding modifier, default dropout
        "M-R-T1-G0-V-T2-G0_1-V-C1_2-S1_0_32-D-MO",  # hop filters, multi-tag concat, slice
        # every --network_setup that appears in experiments/h2gcn/configs/**/h2gcn.json and mlp.json
        "M64-R-T1-G-V-T2-G-V-C1-C2-MO", "M64-R-T1-G-V-C1-MO", "M64-T1-G-V-T2-G-V-C1-C2-MO",
        "M64-T1-G-V-T2-G-V-C1-C2-D0.5-MO", "M64-T1-G-V-C1-MO", "M64-T1-G-V-C1-D0.5-MO", "M64-R-D-MO", "M64-R-MO",
        "M64-MO", "M64-D-MO",
    ]

    def enc(v):
        if isinstance(v, slice):
            return {"__slice__": [v.start, v.stop, v.step]}
        if isinstance(v, set):
            return {"__set__": sorted(v)}
        return v

    res = {}
    for s in strings:
        parsed = mod.parse_network_setup(s, 7, _dense_units=64, _dropout_rate=0.5, parse_preprocessing=True)
        res[s] = [[t, {k: enc(v) for k, v in conf.items()}] for t, conf in parsed]
    (HERE / "dsl_parse.json").write_text(json.dumps(res, indent=1, sort_keys=True))
    print("dsl strings:", len(res))


GLUE_NETWORKS = [
    "M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO",            # H2GCN-2 (default, h2gcn/models/H2GCN.py:12)
    "M64-R-T1-G-V-C1-D0.5-MO",                      # H2GCN-1
    "M-R-T1-G0-V-T2-G0_1-V-C1_2-S1_0_32-D-MO",      # hop filters, multi-tag concat, slice of a tagged output
    "I-T1-G-V-C1-M64-R-D0.5-MO",                    # identity (dense features) first, aggregation of raw features
    "F64-R-E-D-FO",                                 # bias layers
    "M64-T1-G1-V-T2-G-V-C2_1-MO",                   # single-hop filter, reversed tag list in the concat
]


def glue_fixture():
    """Run the reference's model code (its own source, imported in place) under the TF stand-in."""
    import importlib
    import importlib.util

    sys.path.insert(0, str(HERE))
    sys.path.insert(0, str(HERE.parent))
    import _tf_standin
    from conftest import golden_weight, load_planetoid_golden

    tf = _tf_standin.install()
    for name in ("modules", "modules.logger", "modules.controller", "modules.monitor"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["modules"].logger = sys.modules["modules.logger"]
    sys.modules["modules"].controller = sys.modules["modules.controller"]
    sys.modules["modules"].monitor = sys.modules["modules.monitor"]
    pkg_dir = REF / "h2gcn/models"
    for k in [k for k in sys.modules if k.startswith("ref_glue")]:
        del sys.modules[k]
    spec = importlib.util.spec_from_file_location("ref_glue", pkg_dir / "__init__.py", submodule_search_locations=[str(pkg_dir)])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["ref_glue"] = pkg
    spec.loader.exec_module(pkg)                              # imports the reference's real _layers.py
    H = importlib.import_module("ref_glue.H2GCN")             # the reference's real H2GCN.py (+ _metrics.py)

    g = load_planetoid_golden("cora")
    n = g["n"]
    feats = tf.SparseTensor(g["feat_rownorm"])
    adjhops = [tf.SparseTensor(g["hop1_sym"]), tf.SparseTensor(g["hop2_sym"])]
    labels = (g["y_all"] * g["train_mask"][:, None]).astype(np.float32)
    deg2 = np.diff(g["hop2_sym"].indptr)
    rows = np.array(sorted(set([0, 1, n - 1, int(np.diff(g["hop1_sym"].indptr).argmax()), int(deg2.argmax())]
                               + np.where(deg2 == 0)[0][:6].tolist() + list(range(50, 2700, 97)))))
    out, meta = {"rows": rows}, {"networks": GLUE_NETWORKS, "l2": 5e-4, "entries": []}
    for i, net in enumerate(GLUE_NETWORKS):
        _tf_standin.reset_weights(golden_weight)
        setups = pkg.parse_network_setup(net, 7, _dense_units=64, _dropout_rate=0.5, parse_preprocessing=True)
        model = H.H2GCN(setups, l2_regularize_weight=5e-4)
        acts = {}
        logits = model(None, feats, adjhops, training=False, saveActivations=None)
        # tagged outputs: re-run the reference's loop bookkeeping through its own tagsDict
        tagged = {}
        x = feats
        for ind, layer in enumerate(model.layer_objs):
            if ind in model.concat_inds:
                x = layer(x, **tagged)
            elif ind in model.graph_hops_inds:
                x = layer(adjhops, x)
            else:
                x = layer(x)
            if ind in model.tagsDict:
                tagged[model.tagsDict[ind]] = x
        assert np.array_equal(x, logits)
        loss = float(model._loss(logits, labels, g["train_mask"]))
        acc = float(H.masked_accuracy(logits, labels, g["train_mask"]))
        reg = float(tf.math.add_n(model.losses))
        out[f"n{i}_logits"] = logits.astype(np.float32)
        for tag, v in tagged.items():
            v = np.asarray(v)
            out[f"n{i}_tag{tag}_rows"] = v[rows].astype(np.float32)
            out[f"n{i}_tag{tag}_colsum64"] = v.astype(np.float64).reshape(n, -1).sum(0)
        meta["entries"].append({"network": net, "weights": [[k, list(sh)] for k, sh in _tf_standin._WEIGHT_LOG],
                                "tags": {t: list(np.asarray(v).shape) for t, v in tagged.items()},
                                "loss": loss, "train_acc": acc, "l2_term": reg})
        print("glue:", net, "logits", logits.shape, "loss %.6f acc %.4f" % (loss, acc), "weights", _tf_standin._WEIGHT_LOG)
    np.savez_compressed(HERE / "glue_cora.npz", **out)
    (HERE / "glue_cora.json").write_text(json.dumps(meta, indent=1))
    for name in ("tensorflow", "tensorflow.keras", "tensorflow.sparse"):
        sys.modules.pop(name, None)


def syn_fixture():
    import importlib.util

    import networkx as nx

    # generator shims: Graph.node -> .nodes ; G.selfloop_edges() -> nx.selfloop_edges(G)
    if not hasattr(nx.Graph, "node"):
        nx.Graph.node = property(lambda self: self.nodes)
    if not hasattr(nx.Graph, "selfloop_edges"):
        nx.Graph.selfloop_edges = lambda self, *a, **k: nx.selfloop_edges(self, *a, **k)
    if not hasattr(nx, "write_gpickle"):
        nx.write_gpickle = lambda *a, **k: None
    spec = importlib.util.spec_from_file_location("ref_graphgen", REF / "experiments/h2gcn/modules/graphgen.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--syn", action="store_true")
    a = ap.parse_args()
    ref = _import_reference_dataset()
    planetoid_fixture(ref, "ind.cora")
    planetoid_fixture(ref, "ind.citeseer")
    dsl_fixture()
    layer_output_fixture()
    glue_fixture()
    make_syn_small(syn_fixture(), ref)
    if a.syn:
        make_syn(syn_fixture())


def make_syn(mod):
    """n=10000, 10 equal classes, m=6, m0=60 (= m * numClass, the smallest the generator accepts), h=0.2,
    heteroClsWeight="circularDist", heteroWeightsExponent=1.0 -- the syn-products recipe
    (experiments/h2gcn/run_graph_generation.py:45-49)."""
    seed = 20200
    np.random.seed(seed)                              # get_neighbors draws from the global RNG (graphgen.py:107)
    mod.random_state = np.random.RandomState(seed)    # label shuffle (graphgen.py:148,150)
    n, n_class, m, h = 10000, 10, 6, 0.2
    gen = mod.MixhopGraphGenerator([n // n_class] * n_class, "circularDist", heteroWeightsExponent=1.0)
    G = gen(n, m, m * n_class, h)
    import networkx as nx

    A = sp.csr_matrix(nx.adjacency_matrix(G, nodelist=range(n))).astype(np.float32)
    A.sort_indices()
    labels = np.array([G.nodes[v]["color"] - 1 for v in range(n)], dtype=np.int8)
    src, dst = A.nonzero()
    homophily = float((labels[src] == labels[dst]).mean())
    np.savez_compressed(HERE / "syn_products.npz", indptr=A.indptr.astype(np.int64), indices=A.indices.astype(np.int32),
                        labels=labels, homophily=np.float64(homophily), seed=np.int64(seed))
    print("syn graph: nnz", A.nnz, "edge homophily", homophily)


def make_syn_small(mod, ref):
    """A small graph from the reference generator, WRITTEN TO DISK BY THE GENERATOR'S OWN save_* methods: fixtures for
    the reader of the generator format (graphgen.py:37-66)."""
    import gzip
    import pickle

    import networkx as nx

    out = HERE / "generated"
    out.mkdir(exist_ok=True)
    seed = 4242
    np.random.seed(seed)
    mod.random_state = np.random.RandomState(seed)
    n, n_class, m, h = 400, 5, 3, 0.3
    gen = mod.MixhopGraphGenerator([n // n_class] * n_class, "circularDist", heteroWeightsExponent=1.0)
    G = gen(n, m, m * n_class, h)
    nx.write_gpickle = lambda g, path: pickle.dump(g, gzip.open(path, "wb"))   # removed from networkx 3
    mod.nx.write_gpickle = nx.write_gpickle
    gen.save_graph(G, str(out), "syn_small")
    gen.save_y(G, str(out), "syn_small")
    gen.save_nx_graph(G, str(out), "syn_small")
    graph = pickle.load(open(out / "syn_small.graph", "rb"))
    A = sp.csr_matrix(ref.PlanetoidData.graphDict2Adj(graph)).astype(np.float32)   # the reference's own conversion
    A.sort_indices()
    np.savez_compressed(out / "syn_small_expected.npz", indptr=A.indptr.astype(np.int64), indices=A.indices.astype(np.int32),
                        data=A.data, labels=np.array([G.nodes[v]["color"] - 1 for v in range(n)], dtype=np.int8))
    print("syn_small: nnz", A.nnz)


if __name__ == "__main__":
    main()
