"""CPU suite: the host plumbing either side of the path -- planetoid on-disk format, plugin/argparse contract,
metrics, early stopping, static width tracking of the model interpreter.  No GPU compute."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_planetoid_golden
from h2gcn_amd.datasets._dataset import PlanetoidData, adjacency_from_neighbour_lists, export_planetoid
from h2gcn_amd.models import parse_network_setup
from h2gcn_amd.models._metrics import masked_accuracy, masked_softmax_cross_entropy
from h2gcn_amd.modules import arguments
from h2gcn_amd.modules.controller import SlidingMeanEarlyStopping
from oracle import h2gcn_model as om


def _export_fixture(g, tmp_path, name):
    """Write the golden (reference-loaded) graph back to planetoid files; isolated citeseer nodes (all-zero
    label rows inside the test range) are left out of test.index, as in the original files."""
    n = g["n"]
    n_train = int(g["train_mask"].sum())
    y = g["y_all"].astype(np.float64)
    test_ids = np.where(g["test_mask"])[0]
    rng = np.random.default_rng(0)
    test_ids = rng.permutation(test_ids)  # test.index is not sorted in the real files either
    lo = int(g["test_mask"].nonzero()[0].min())
    isolated = [i for i in range(lo, n) if y[i].sum() == 0]
    if isolated:
        lo = min(lo, min(isolated))
    export_planetoid(tmp_path, name, g["adj_raw"], g["feat_raw"], y, n_train, test_ids.tolist(), n_allx=lo)
    return n_train


@pytest.mark.parametrize("name", ["cora", "citeseer"])
def test_planetoid_round_trip_matches_reference_loader(name, tmp_path):
    g = load_planetoid_golden(name)  # produced by the reference's PlanetoidData
    _export_fixture(g, tmp_path, f"ind.{name}")
    d = PlanetoidData(f"ind.{name}", tmp_path, val_size=500)
    a = sp.csr_matrix(d.sparse_adj); a.sort_indices()
    assert (abs(a - g["adj_raw"])).nnz == 0
    assert abs(sp.csr_matrix(d.features) - g["feat_raw"]).max() == 0
    assert np.array_equal(d.y_all.astype(np.int8), g["y_all"])
    for m in ("train_mask", "val_mask", "test_mask"):
        assert np.array_equal(getattr(d, m), g[m]), m
    assert d.num_labels == g["num_labels"] and d.num_samples == g["n"]
    assert int(d.train_mask.sum()) == {"cora": 140, "citeseer": 120}[name] and int(d.val_mask.sum()) == 500
    # preprocessing order of the model plugin (reference H2GCN.py:46-54) reproduces the golden operands
    d.row_normalize_features()
    d.adj_remove_eye()
    assert abs(sp.csr_matrix(d.features).astype(np.float32) - g["feat_rownorm"]).max() == 0
    assert abs(sp.csr_matrix(d.sparse_adj) - g["adj_noeye"]).nnz == 0


@pytest.mark.skipif(not __import__("pathlib").Path("/root/reference/baselines/gcn/gcn/data/ind.cora.x").exists(),
                    reason="reference data files only exist in the build container")
@pytest.mark.parametrize("name", ["cora", "citeseer"])
def test_loader_on_the_reference_data_files(name):
    g = load_planetoid_golden(name)
    d = PlanetoidData(f"ind.{name}", "/root/reference/baselines/gcn/gcn/data", val_size=500)
    a = sp.csr_matrix(d.sparse_adj); a.sort_indices()
    assert (abs(a - g["adj_raw"])).nnz == 0
    assert abs(sp.csr_matrix(d.features) - g["feat_raw"]).max() == 0
    assert np.array_equal(d.y_all.astype(np.int8), g["y_all"])
    for m in ("train_mask", "val_mask", "test_mask"):
        assert np.array_equal(getattr(d, m), g[m]), m
    assert str(sp.csr_matrix(d.features).dtype) == str(g["feat_raw"].dtype)


def test_adjacency_from_lists_symmetrises_and_dedups():
    a = adjacency_from_neighbour_lists({0: [1, 1, 2], 1: [0], 2: [2]})
    assert a.toarray().tolist() == [[0, 1, 1], [1, 0, 0], [1, 0, 1]] and a.dtype == np.float32


def test_metrics_known_answers():
    preds = np.array([[2.0, 0.0], [0.0, 3.0], [1.0, 1.0], [5.0, -5.0], [0.0, 0.0]])
    labels = np.array([[1, 0], [1, 0], [0, 1], [0, 1], [0, 0]], dtype=np.float64)
    mask = np.array([1, 1, 1, 0, 1], dtype=bool)
    lse = np.log(np.exp(preds).sum(1))
    want = (-(preds[0, 0] - lse[0]) - (preds[1, 0] - lse[1]) - (preds[2, 1] - lse[2]) + 0.0) / 4
    got = masked_softmax_cross_entropy(torch.tensor(preds), torch.tensor(labels), torch.tensor(mask)).item()
    assert abs(got - want) < 1e-12 and abs(om.masked_softmax_cross_entropy(preds, labels, mask) - want) < 1e-12
    acc = masked_accuracy(torch.tensor(preds), torch.tensor(labels), torch.tensor(mask)).item()
    # rows 0 (correct), 1 (wrong), 2 (argmax tie -> class 0, wrong), 4 (all-zero label: argmax 0 == argmax 0)
    assert abs(acc - 0.5) < 1e-12 and abs(om.masked_accuracy(preds, labels, mask) - 0.5) < 1e-12


def test_early_stopping_window():
    s = SlidingMeanEarlyStopping(3)
    assert [s(v) for v in (1.0, 1.0, 1.0)] == [False] * 3
    assert s(0.5) is False      # below the mean: enters the window
    assert s(2.0) is True       # window full and above its mean
    assert SlidingMeanEarlyStopping(0)(123.0) is False


def test_argparse_hooks_run_dataset_first():
    parser = arguments.create_parser()
    order = []
    parser.function_hooks["argparse"].append(lambda a: order.append("model"))
    parser.function_hooks["argparse"].appendleft(lambda a: order.append("dataset"))
    args = arguments.parse_args(parser, [])
    assert order == ["dataset", "model"]
    assert set(args.objects) >= {"pretrain_callbacks", "pre_epoch_callbacks", "post_epoch_callbacks", "post_train_callbacks"}


def test_model_width_tracking_matches_oracle_forward():
    from h2gcn_amd.models.H2GCN import H2GCN

    for text, want in (("M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO", 448), ("M64-R-T1-G-V-C1-D0.5-MO", 192),
                       ("M64-R-D0.5-MO", 64), ("M-R-T1-G0-V-T2-G0_1-V-C1_2-S1_0_32-D-MO", 32)):
        setup = parse_network_setup(text, 7, _dense_units=64, _dropout_rate=0.5)
        m = H2GCN(setup, input_dim=30, n_hops=2)
        assert m.layer_objs[-1].kernel.shape == (want, 7), text
        # numpy interpreter agrees on the width feeding the output layer
        rng = np.random.default_rng(0)
        feats = sp.random(12, 30, 0.3, format="csr", random_state=0)
        hops = [sp.random(12, 12, 0.3, format="csr", random_state=k) for k in (1, 2)]
        weights = [rng.standard_normal(tuple(l.kernel.shape)) for l in m.regularized]
        enc = [[k, {kk: ({"__set__": sorted(v)} if isinstance(v, set) else v) for kk, v in c.items()}] for k, c in setup]
        assert om.forward(enc, feats, hops, weights).shape == (12, 7)


def test_reference_launch_line_flags_are_accepted():
    """The argv the reference's orchestrator builds (experiments/h2gcn/run_hgcn_experiments.py:13-29 +
    configs' model_args) parses: bookkeeping flags outside this build are accepted and ignored."""
    from h2gcn_amd import run_experiments
    from h2gcn_amd.models import H2GCN as plugin
    from h2gcn_amd.datasets import planetoid
    from h2gcn_amd.modules import logger

    parser = run_experiments.build_parser()
    parser.add_argument("model")
    parser.add_argument("datafmt")
    plugin.add_subparser_args(parser)
    planetoid.add_subparser_args(parser)
    logger.add_subparser_args(parser)
    parser.function_hooks["argparse"].clear()  # parse only, no data / model construction
    argv = ["H2GCN", "planetoid", "--dataset_path", "/data/syn", "--dataset", "syn-products-h0.2", "--run_id=7",
            "--use_signac", "--signac_root", "/tmp/ws", "--exp_tags", "a", "b", "--val_size", "500",
            "--network_setup", "M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO", "--adj_nhood", "1", "2",
            "--l2_regularize_weight", "1e-5", "--no_feature_normalize", "--early_stopping", "40", "--grad_monitor",
            "--deg_acc_monitor", "0.5", "--save_activations", "-v"]
    args = arguments.parse_args(parser, argv)
    assert args.dataset == "syn-products-h0.2" and args._dataset_path == "/data/syn" and args.val_size == 500
    assert args.l2_regularize_weight == 1e-5 and args.no_feature_normalize and args.early_stopping == 40
    assert args._signac_root == "/tmp/ws" and args.verbose and args._exp_tags == ["a", "b"]


def test_dense_features_are_not_wrapped_as_sparse_operands(tmp_path):
    """Feature matrices denser than 25 % are kept dense (GEMM-shaped embedding); sparse ones become a 1-hop
    operand -- decided without touching the GPU."""
    from h2gcn_amd.datasets._dataset import PlanetoidData

    f_dense = sp.csr_matrix(np.random.default_rng(0).standard_normal((6, 5)).astype(np.float32))
    dummy = PlanetoidData.__new__(PlanetoidData)
    out = dummy._feature_operand(f_dense, "cpu", True)
    assert isinstance(out, torch.Tensor) and out.shape == (6, 5)
    f_sparse = sp.random(50, 40, 0.05, format="csr", random_state=0, dtype=np.float32)
    with pytest.raises(ValueError, match="GPU"):   # sparse -> HopPlan, which refuses CPU tensors (no fallback)
        dummy._feature_operand(f_sparse, "cpu", True)


# ------------------------------------------------------------------ the generator's on-disk format (graphgen.py:37-66)
def test_reader_of_the_generator_format_matches_reference_conversion(tmp_path):
    """tests/golden/generated/syn_small.{graph,ally,gpickle.gz} were WRITTEN BY THE REFERENCE'S GENERATOR (save_graph /
    save_y / save_nx_graph, imported in place by make_golden.py); syn_small_expected.npz holds the adjacency the
    reference's own graphDict2Adj derives from the .graph file.  Both the pickle pair and the gzip'd networkx pickle
    must give exactly that graph."""
    import shutil

    import numpy as np

    from conftest import GOLDEN
    from h2gcn_amd.datasets._dataset import GeneratedGraphData, read_generated_graph

    z = np.load(GOLDEN / "generated" / "syn_small_expected.npz")
    adj, ally = read_generated_graph(GOLDEN / "generated", "syn_small")
    assert np.array_equal(adj.indptr, z["indptr"]) and np.array_equal(adj.indices, z["indices"]) and np.array_equal(adj.data, z["data"])
    assert ally.shape == (400, 5) and np.array_equal(ally.argmax(1), z["labels"]) and (ally.sum(1) == 1).all()
    # only the networkx pickle present
    shutil.copy(GOLDEN / "generated" / "syn_small.gpickle.gz", tmp_path / "only_nx.gpickle.gz")
    adj2, ally2 = read_generated_graph(tmp_path, "only_nx")
    assert (adj2 != adj).nnz == 0 and np.array_equal(ally2, ally)
    data = GeneratedGraphData("syn_small", GOLDEN / "generated", feature_dim=16, feature_seed=1, split_seed=2)
    assert data.num_samples == 400 and data.num_labels == 5 and data.features.shape == (400, 16)
    masks = data.train_mask.astype(int) + data.val_mask.astype(int) + data.test_mask.astype(int)
    assert (masks == 1).all() and data.train_mask.sum() == 100 and data.val_mask.sum() == 100
    with __import__("pytest").raises(FileNotFoundError):
        read_generated_graph(tmp_path, "missing")


def test_script_launch_from_the_package_directory():
    """The reference is launched as `python run_experiments.py ...` with cwd = its package directory
    (experiments/h2gcn/experiments_workflow.py:301-318); the same line works here."""
    import subprocess
    import sys
    from pathlib import Path

    pkg = Path(__file__).resolve().parents[1] / "h2gcn_amd"
    r = subprocess.run([sys.executable, "run_experiments.py", "H2GCN", "planetoid", "--help"], cwd=pkg, capture_output=True, text=True)
    assert r.returncode == 0 and "--network_setup" in r.stdout and "--dataset_path" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_synthetic_shape_through_the_entry_point(capsys):
    """`run_experiments.py H2GCN synthetic --shape arxiv`: BASELINE configs[2]'s shape through the reference-style entry point
    (operands generated on the device, 2-hop matrix supplied), full-batch H2GCN-2 epochs incl. hipGraph replay."""
    from h2gcn_amd import run_experiments

    args = run_experiments.main(["H2GCN", "synthetic", "--shape", "arxiv", "--epochs", "6", "--no_feature_normalize", "--classes", "40",
                                 "--random_seed", "3"])
    stats = args.objects["epoch_stats"]
    assert np.isfinite(stats["train_loss"]) and 0.0 <= stats["val_acc"] <= 1.0
    assert args.objects["tensors"]["adj_hops"].n_rows == 170_000
    out = capsys.readouterr().out
    assert "synthetic shape arxiv" in out and "Epoch: 0006" in out


def test_bench_without_a_gpu_or_with_too_few_prints_one_error_line():
    """`python bench.py --gpus N` where N GPUs are not there (here: none): ONE JSON line with "error", exit code 2, no traceback --
    what the driver's plain `--gpus N` command gets instead of an argument error.  (The ranks are launched by bench.py itself
    when they are: tests/test_multirank_gpu.py::test_bench_plain_launch_spawns_its_own_ranks.)"""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    ROOT = Path(__file__).resolve().parents[1]
    if torch.cuda.is_available() and torch.cuda.device_count() >= 4:
        pytest.skip("a 4-GPU node would run this")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "H2GCN_SHARE_GPU")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--steps", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "Traceback" not in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 4 and "GPU" in line["error"]
