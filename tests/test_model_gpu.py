"""GPU suite: the caller of the hot path -- H2GCN model interpreter, SparseDense, losses, entry point -- on Cora
(BASELINE.json configs[0], run on the MI355X instead of TF-CPU).  Forward against the numpy oracle interpreter,
gradients against a dense float64 torch-CPU replica, and a short end-to-end training run."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_planetoid_golden
from oracle import h2gcn_model as om

pytestmark = pytest.mark.gpu
H2GCN2 = "M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO"


def _cora_files(tmp_path):
    from test_entrypoints import _export_fixture

    g = load_planetoid_golden("cora")
    _export_fixture(g, tmp_path, "ind.cora")
    return g


def _setup(tmp_path, network=H2GCN2):
    from h2gcn_amd.datasets._dataset import PlanetoidData
    from h2gcn_amd.models import parse_network_setup
    from h2gcn_amd.models.H2GCN import H2GCN

    g = _cora_files(tmp_path)
    data = PlanetoidData("ind.cora", tmp_path, val_size=500)
    data.row_normalize_features()
    data.adj_remove_eye()
    dev = torch.device("cuda:0")
    tensors = data.get_tensors(dev, adj_norm_hops=["1", "2"])
    setup = parse_network_setup(network, data.num_labels, _dense_units=64, _dropout_rate=0.5)
    torch.manual_seed(0)
    model = H2GCN(setup, input_dim=tensors["features"].n_cols, n_hops=2, l2_regularize_weight=5e-4).to(dev)
    return g, data, tensors, setup, model


def _enc(setup):
    return [[k, {kk: ({"__set__": sorted(v)} if isinstance(v, set) else v) for kk, v in c.items()}] for k, c in setup]


@pytest.mark.parametrize("network", [H2GCN2, "M64-R-T1-G-V-C1-D0.5-MO", "M-R-T1-G0-V-T2-G0_1-V-C1_2-S1_0_32-D-MO",
                                     "M64-T1-G-V-T2-G-V-C1-C2-MO", "M64-T1-G-V-C1-D0.5-MO", "M64-R-D-MO", "M64-MO"])
def test_forward_matches_oracle_interpreter(tmp_path, network):
    g, data, tensors, setup, model = _setup(tmp_path, network)
    model.eval()
    tagged = {}
    with torch.no_grad():
        logits = model(tensors["adj"], tensors["features"], tensors["adj_hops"], tagged_out=tagged)
    weights = [l.kernel.detach().cpu().numpy() for l in model.regularized]
    hops = [g["hop1_sym"], g["hop2_sym"]]
    want, want_tagged, _ = om.forward(_enc(setup), g["feat_rownorm"], hops, weights, return_tagged=True)
    assert logits.shape == (g["n"], 7)
    assert np.abs(logits.cpu().numpy() - want).max() <= 1e-5
    for name, v in want_tagged.items():  # r0 ("1") and r1 ("2") of SURVEY.md §3.2
        assert np.abs(tagged[name].cpu().numpy() - v).max() <= 1e-5, name
    # loss / accuracy restatements agree
    y = tensors["y_train"]
    l_gpu = model.loss(logits, y, tensors["train_mask"]).item()
    reg = 5e-4 * sum(float((w ** 2).sum()) for w in weights)
    l_cpu = om.masked_softmax_cross_entropy(want, g["y_all"] * g["train_mask"][:, None], g["train_mask"]) + reg
    assert abs(l_gpu - l_cpu) <= 1e-5


def _glue_entries():
    import json

    from conftest import GOLDEN
    return json.loads((GOLDEN / "glue_cora.json").read_text())["entries"]


@pytest.mark.parametrize("idx", range(6))
def test_forward_matches_reference_glue_fixture(tmp_path, idx):
    """HIP path vs tests/golden/glue_cora.*: logits / tagged activations / loss produced by the REFERENCE'S OWN
    `_layers.py` + `H2GCN.py` + `_metrics.py` (run under a numpy stand-in for TF by tests/golden/make_golden.py) on
    the golden Cora operands with the regenerable golden weights.  Both the concat-free fused propagation and the
    layer-by-layer interpreter are checked."""
    from conftest import GOLDEN, golden_weight

    entry = _glue_entries()[idx]
    z = np.load(GOLDEN / "glue_cora.npz")
    g, data, tensors, setup, model = _setup(tmp_path, entry["network"])
    model.eval()
    params = []
    for layer in model.regularized:
        params.append(layer.kernel)
        if layer.bias is not None:
            params.append(layer.bias)
    assert [list(p.shape) for p in params] == [shape for _, shape in entry["weights"]]
    with torch.no_grad():
        for i, ((kind, shape), p) in enumerate(zip(entry["weights"], params)):
            p.copy_(torch.from_numpy(golden_weight(i, tuple(shape), kind)))
    want = z[f"n{idx}_logits"]
    for fuse in (True, False):
        tagged = {}
        with torch.no_grad():
            logits = model(tensors["adj"], tensors["features"], tensors["adj_hops"], tagged_out=tagged, fuse=fuse)
        assert np.abs(logits.cpu().numpy() - want).max() <= 1e-5, fuse
        assert set(tagged) == set(entry["tags"])
        for tag, shape in entry["tags"].items():
            v = tagged[tag].cpu().numpy()
            assert list(v.shape) == shape, tag
            assert np.abs(v[z["rows"]] - z[f"n{idx}_tag{tag}_rows"]).max() <= 1e-5, tag
            assert np.allclose(v.astype(np.float64).reshape(g["n"], -1).sum(0), z[f"n{idx}_tag{tag}_colsum64"], rtol=0, atol=1e-3)
    loss = model.loss(logits, tensors["y_train"], tensors["train_mask"]).item()
    assert abs(loss - entry["loss"]) <= 1e-5
    from h2gcn_amd.models._metrics import masked_accuracy
    assert abs(masked_accuracy(logits, tensors["y_train"], tensors["train_mask"]).item() - entry["train_acc"]) <= 1e-6


def test_syn_products_h2gcn2_logits(tmp_path):
    """BASELINE.json configs[1]: syn-products-shaped graph (reference generator, n=10k, h=0.2), synthetic
    class-conditional features d=100, `--no_feature_normalize` (experiments/h2gcn/configs/syn-products/h2gcn.json),
    H2GCN-2 forward: r1, r2 (d=64 and d=128 launches) and logits within 1e-5 of the CPU oracle interpreter."""
    from conftest import load_syn_products_golden
    from h2gcn_amd import HopPlan, operands
    from h2gcn_amd.models import parse_network_setup
    from h2gcn_amd.models.H2GCN import H2GCN
    from oracle import operands as oo

    a, labels, _ = load_syn_products_golden()
    rng = np.random.default_rng(5)
    centers = rng.standard_normal((10, 100))
    feats = (centers[labels] + rng.standard_normal((10000, 100))).astype(np.float32)
    dev = torch.device("cuda:0")
    hops = operands.build_adj_norm_hops(operands.remove_self_loops(a), ["1", "2"], "sym")
    plan = HopPlan.from_scipy(hops, dev, build_transpose=True)
    fplan = HopPlan.from_scipy([sp.csr_matrix(feats)], dev, build_transpose=True)
    setup = parse_network_setup(H2GCN2, 10, _dense_units=64, _dropout_rate=0.5)
    torch.manual_seed(1)
    model = H2GCN(setup, input_dim=100, n_hops=2, l2_regularize_weight=5e-4).to(dev).eval()
    tagged = {}
    with torch.no_grad():
        logits = model(None, fplan, plan, tagged_out=tagged)
    weights = [l.kernel.detach().cpu().numpy() for l in model.regularized]
    ohops = oo.adj_norm_hops(oo.remove_eye(a), ("1", "2"), oo.SYM)  # the oracle's own operand construction
    want, want_tagged, trace = om.forward(_enc(setup), sp.csr_matrix(feats), ohops, weights, return_tagged=True)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(logits.cpu().numpy() - want).max() <= 1e-5 * scale
    # r1 (tag "2": the d = 64 launch) and r2 (untagged: the d = 128 launch, the leading 256 columns of the concat [r2|r0|r1])
    # EACH on its own, not only through the logits.  Tolerance: 1e-5 x max(1, max|reference|) -- the features are not
    # row-normalised here (`--no_feature_normalize`), so activations are O(10), not O(1); BASELINE.md section 4 row 2 says so
    assert np.abs(tagged["2"].cpu().numpy() - want_tagged["2"]).max() <= 1e-5 * max(1.0, np.abs(want_tagged["2"]).max())
    kinds = [k for k, _ in setup]
    second_v, last_c = [i for i, k in enumerate(kinds) if k == "V"][1], [i for i, k in enumerate(kinds) if k == "C"][-1]
    r2_want = np.asarray(trace[second_v])
    assert r2_want.shape == (10000, 256) and np.array_equal(np.asarray(trace[last_c])[:, :256], r2_want)
    with torch.no_grad():
        cat = model(None, fplan, plan, return_before=last_c + 1)
    assert cat.shape == (10000, 448)
    r2_got = cat[:, :256].cpu().numpy()
    assert np.abs(r2_got - r2_want).max() <= 1e-5 * max(1.0, np.abs(r2_want).max()), np.abs(r2_got - r2_want).max()
    assert np.abs(cat[:, 320:].cpu().numpy() - want_tagged["2"]).max() <= 1e-5 * max(1.0, np.abs(want_tagged["2"]).max())
    assert plan.nnz == [h.nnz for h in ohops]
    # dense-ish features take the GEMM path (dense operand, rocBLAS) -- same logits
    torch.manual_seed(1)
    dense_model = H2GCN(setup, input_dim=100, n_hops=2, sparse_input=False, l2_regularize_weight=5e-4).to(dev).eval()
    dense_model.load_state_dict(model.state_dict())
    with torch.no_grad():
        logits_dense = dense_model(None, torch.from_numpy(feats).to(dev), plan)
    assert (logits_dense - logits).abs().max().item() <= 1e-4 * scale


def test_device_built_hops_equal_host_built(tmp_path):
    """get_tensors grows the exact-k-hop rings on the GPU by default; the operands equal the scipy-built ones bit
    for bit (so does a forward pass), on Cora and on the 10k-node syn-products graph (2.8M-nonzero 2-hop ring)."""
    from conftest import load_syn_products_golden
    from h2gcn_amd import HopPlan, operands
    from h2gcn_amd.datasets._dataset import PlanetoidData

    _cora_files(tmp_path)
    data = PlanetoidData("ind.cora", tmp_path, val_size=500)
    data.adj_remove_eye()
    dev = torch.device("cuda:0")
    td = data.get_tensors(dev, adj_norm_hops=["1", "2"])
    th = data.get_tensors(dev, adj_norm_hops=["1", "2"], host_hops=True)
    for k in range(2):
        assert torch.equal(td["adj_hops"].rowptr[k], th["adj_hops"].rowptr[k])
        assert torch.equal(td["adj_hops"].colidx[k], th["adj_hops"].colidx[k])
        assert torch.equal(td["adj_hops"].vals[k], th["adj_hops"].vals[k])
    a, _, _ = load_syn_products_golden()
    adj = operands.remove_self_loops(a)
    rp, ci, va, n = operands.build_adj_norm_hops_device(adj, ["1", "2"], "sym", dev)
    host = operands.build_adj_norm_hops(adj, ["1", "2"], "sym")
    for k, h in enumerate(host):
        h = sp.csr_matrix(h); h.sort_indices()
        assert np.array_equal(ci[k].cpu().numpy(), h.indices) and np.array_equal(va[k].cpu().numpy(), h.data.astype(np.float32))
    x = torch.rand((n, 64), device=dev)
    assert torch.equal(HopPlan(rp, ci, va, n).spmm(x), HopPlan.from_scipy(host, dev).spmm(x))


def test_gradients_match_dense_float64_replica(tmp_path):
    g, data, tensors, setup, model = _setup(tmp_path)
    model.eval()  # no dropout: deterministic comparison
    logits = model(tensors["adj"], tensors["features"], tensors["adj_hops"])
    loss = model.loss(logits, tensors["y_train"], tensors["train_mask"])
    loss.backward()
    # dense float64 replica on the CPU
    A1 = torch.from_numpy(g["hop1_sym"].toarray().astype(np.float64))
    A2 = torch.from_numpy(g["hop2_sym"].toarray().astype(np.float64))
    X = torch.from_numpy(g["feat_rownorm"].toarray().astype(np.float64))
    W0 = model.regularized[0].kernel.detach().cpu().double().requires_grad_(True)
    W1 = model.regularized[1].kernel.detach().cpu().double().requires_grad_(True)
    r0 = torch.relu(X @ W0)
    r1 = torch.cat([A1 @ r0, A2 @ r0], 1)
    r2 = torch.cat([A1 @ r1, A2 @ r1], 1)
    z = torch.cat([r2, r0, r1], 1) @ W1
    y = torch.from_numpy((g["y_all"] * g["train_mask"][:, None]).astype(np.float64))
    m = torch.from_numpy(g["train_mask"].astype(np.float64))
    ref = (-(y * torch.log_softmax(z, 1)).sum(1) * (m / m.sum())).sum() + 5e-4 * ((W0 ** 2).sum() + (W1 ** 2).sum())
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5
    for got, want in ((model.regularized[0].kernel.grad, W0.grad), (model.regularized[1].kernel.grad, W1.grad)):
        assert (got.cpu().double() - want).abs().max().item() <= 1e-6 + 1e-4 * want.abs().max().item()


@pytest.mark.parametrize("network", [H2GCN2, "M64-R-T1-G-V-C1-D0.5-MO", "M64-R-T1-G-V-T2-G-V-T3-G-V-C1-C2-C3-MO"])
def test_concat_free_propagation_equals_generic_interpreter(tmp_path, network):
    """hop outputs written straight into the [N, W] concat buffer (no stack / flatten /
    concat copies) give the same bits as the layer-by-layer interpreter, and the same gradients."""
    g, data, tensors, setup, model = _setup(tmp_path, network)
    assert model.fused is not None
    model.eval()
    args = (tensors["adj"], tensors["features"], tensors["adj_hops"])
    tg_f, tg_g = {}, {}
    y_f = model(*args, tagged_out=tg_f)
    gf = torch.autograd.grad(model.loss(y_f, tensors["y_train"], tensors["train_mask"]), list(model.parameters()))
    y_g = model(*args, tagged_out=tg_g, fuse=False)
    gg = torch.autograd.grad(model.loss(y_g, tensors["y_train"], tensors["train_mask"]), list(model.parameters()))
    assert torch.equal(y_f, y_g)
    assert set(tg_f) == set(tg_g) and all(torch.equal(tg_f[k], tg_g[k]) for k in tg_f)
    for a, b in zip(gf, gg):
        assert (a - b).abs().max().item() <= 1e-6 + 1e-5 * b.abs().max().item()
    # the fused buffer really is [r_K | r_0 | ... ]: embeddings before the output layer
    emb = model(*args, return_before=-1)
    assert emb.shape[1] == model.layer_objs[-1].kernel.shape[0]


def test_entry_point_trains_cora(tmp_path, capsys):
    """`run_experiments H2GCN planetoid --dataset ind.cora ...` (reference README usage): loss falls, validation
    accuracy reaches the usual band for H2GCN-2 on Cora (sanity, not a target)."""
    from h2gcn_amd import run_experiments

    _cora_files(tmp_path)
    args = run_experiments.main(["H2GCN", "planetoid", "--dataset", "ind.cora", "--dataset_path", str(tmp_path),
                                 "--epochs", "120", "--random_seed", "123"])
    out = capsys.readouterr().out
    assert "Epoch: 0001" in out and "Best performance:" in out and "===> Dataset loaded: ind.cora" in out
    best = args.objects["best_val_stats"]
    assert best["val_acc"] >= 0.75 and best["test_accuracy"] >= 0.75
    first_loss = float(out.split("Train Loss:")[1].split()[0])
    assert args.objects["epoch_stats"]["train_loss"] < 0.6 * first_loss
    assert args.objects["tensors"]["adj_hops"].nnz == [10556, 86332]


def test_hipgraph_replay_matches_eager_training(tmp_path, capsys):
    """Captured train/test steps (hipGraph replay) follow the eager trajectory: same seed, dropout disabled so the
    comparison is deterministic -> per-epoch statistics agree to rounding."""
    from h2gcn_amd import run_experiments

    _cora_files(tmp_path)
    common = ["H2GCN", "planetoid", "--dataset", "ind.cora", "--dataset_path", str(tmp_path), "--epochs", "25",
              "--random_seed", "7", "--network_setup", "M64-R-T1-G-V-T2-G-V-C1-C2-D0.0-MO"]
    a = run_experiments.main(common)
    stats_graph = dict(a.objects["epoch_stats"])
    assert a.objects["train_step"].__closure__ is not None
    b = run_experiments.main(common + ["--no_hipgraph"])
    stats_eager = dict(b.objects["epoch_stats"])
    for k in ("train_loss", "val_loss", "test_loss", "val_acc", "test_accuracy"):
        assert abs(stats_graph[k] - stats_eager[k]) <= 1e-4, (k, stats_graph[k], stats_eager[k])
    out = capsys.readouterr().out
    assert "capture unavailable" not in out


def test_entry_point_trains_on_a_generator_format_graph(capsys):
    """`run_experiments H2GCN generated ...`: the reference generator's own output files (tests/golden/generated) as
    the dataset -- hop rings built on the device, class-conditional synthetic features (dense -> GEMM embedding),
    `--no_feature_normalize` as in the syn-products configs; the model fits the training nodes."""
    from conftest import GOLDEN
    from h2gcn_amd import run_experiments

    args = run_experiments.main(["H2GCN", "generated", "--dataset", "syn_small", "--dataset_path", str(GOLDEN / "generated"),
                                 "--feature_dim", "32", "--no_feature_normalize", "--epochs", "60", "--random_seed", "3",
                                 "--json_stats"])
    out = capsys.readouterr().out
    assert "===> Dataset loaded: syn_small" in out and '"epoch": 60' in out
    best = args.objects["best_val_stats"]
    assert best["train_acc"] >= 0.9 and best["val_acc"] >= 0.5
    assert args.objects["tensors"]["adj_hops"].n_rows == 400


def test_zero_epochs_reports_the_untrained_model(tmp_path, capsys):
    from h2gcn_amd import run_experiments

    _cora_files(tmp_path)
    args = run_experiments.main(["H2GCN", "planetoid", "--dataset", "ind.cora", "--dataset_path", str(tmp_path), "--epochs", "0"])
    assert args.objects["best_val_stats"]["epoch"] == 0 and "Best performance:" in capsys.readouterr().out


def test_sparse_dropout_network_trains(tmp_path, capsys):
    """`D` before the first dense layer = SparseDropout on the feature operand (reference H2GCN.py:250-257)."""
    from h2gcn_amd import run_experiments
    from h2gcn_amd.layers import SparseDropout

    _cora_files(tmp_path)
    args = run_experiments.main(["H2GCN", "planetoid", "--dataset", "ind.cora", "--dataset_path", str(tmp_path), "--epochs", "40",
                                 "--random_seed", "5", "--network_setup", "D0.3-M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO"])
    model = args.objects["model"]
    assert isinstance(model.layer_objs[0], SparseDropout)
    best = args.objects["best_val_stats"]
    assert best["val_acc"] >= 0.6 and args.objects["epoch_stats"]["train_loss"] < 1.9


def test_citeseer_forward_matches_oracle_and_trains(tmp_path, capsys):
    """Citeseer (reference fixture): 48 isolated nodes -> hop rows with no neighbours (the reference's inf -> 0 branch,
    _dataset.py:115-123) and label-less rows excluded from every mask, float64 feature row-normalisation.  H2GCN-2
    forward (device-built rings) against the numpy oracle interpreter on the golden operands, then a short training
    run through the entry point."""
    from test_entrypoints import _export_fixture
    from h2gcn_amd import run_experiments
    from h2gcn_amd.datasets._dataset import PlanetoidData
    from h2gcn_amd.models import parse_network_setup
    from h2gcn_amd.models.H2GCN import H2GCN

    g = load_planetoid_golden("citeseer")
    _export_fixture(g, tmp_path, "ind.citeseer")
    data = PlanetoidData("ind.citeseer", tmp_path, val_size=500)
    data.row_normalize_features()
    data.adj_remove_eye()
    dev = torch.device("cuda:0")
    tensors = data.get_tensors(dev, adj_norm_hops=["1", "2"])
    assert tensors["adj_hops"].nnz == [9104, 37826]                      # nnz per split [3327, 9104, 37826]
    setup = parse_network_setup(H2GCN2, data.num_labels, _dense_units=64, _dropout_rate=0.5)
    torch.manual_seed(3)
    model = H2GCN(setup, input_dim=tensors["features"].n_cols, n_hops=2, l2_regularize_weight=5e-4).to(dev).eval()
    tagged = {}
    with torch.no_grad():
        logits = model(tensors["adj"], tensors["features"], tensors["adj_hops"], tagged_out=tagged)
    weights = [l.kernel.detach().cpu().numpy() for l in model.regularized]
    want, want_tagged, _ = om.forward(_enc(setup), g["feat_rownorm"], [g["hop1_sym"], g["hop2_sym"]], weights, return_tagged=True)
    assert np.abs(logits.cpu().numpy() - want).max() <= 1e-5
    for name, v in want_tagged.items():
        assert np.abs(tagged[name].cpu().numpy() - v).max() <= 1e-5, name
    empty2 = np.diff(g["hop2_sym"].indptr) == 0
    assert empty2.sum() == 653 and not tagged["2"][torch.from_numpy(empty2).to(dev)][:, 64:].any().item()   # zero rows stay zero
    args = run_experiments.main(["H2GCN", "planetoid", "--dataset", "ind.citeseer", "--dataset_path", str(tmp_path),
                                 "--epochs", "60", "--random_seed", "123"])
    best = args.objects["best_val_stats"]
    assert best["val_acc"] >= 0.55 and np.isfinite(args.objects["epoch_stats"]["train_loss"])


def _train_epochs(model, tensors, n_epochs, pattern="te"):
    """`pattern`: what one epoch does -- "te" = training step + evaluation (the reference's loop), "tt" = two training steps
    in a row before an evaluation.  Returns the parameter values and the logits of the final evaluation."""
    from h2gcn_amd.models.H2GCN import make_optimizer
    opt = make_optimizer("adam", model.parameters(), 0.01)
    args = (tensors["adj"], tensors["features"], tensors["adj_hops"])
    logits = None
    for _ in range(n_epochs):
        for what in pattern:
            if what == "t":
                model.train()
                opt.zero_grad(set_to_none=True)
                model.loss(model(*args), tensors["y_train"], tensors["train_mask"]).backward()
                opt.step()          # (no model.note_update(): an ordinary torch loop -- the reuse key follows p._version)
            else:
                model.eval()
                with torch.no_grad():
                    logits = model(*args)
    return [p.detach().clone() for p in model.parameters()], logits.clone()


@pytest.mark.parametrize("pattern", ["te", "tte", "et", "ett"])
def test_propagation_reuse_is_invisible_in_the_results(tmp_path, monkeypatch, pattern):
    """The training forward adopts the propagation buffer the evaluation has just filled (one propagation per epoch instead
    of two).  Same bits as recomputing it -- parameters after 6 epochs and the final logits -- also when the train / evaluate
    alternation is broken (two training steps in a row: the second one must NOT adopt a stale buffer)."""
    from h2gcn_amd.models.H2GCN import H2GCN

    results = {}
    for reuse in ("1", "0"):
        monkeypatch.setenv("H2GCN_PROPAGATION_REUSE", reuse)
        g, data, tensors, setup, _ = _setup(tmp_path)
        torch.manual_seed(0)
        model = H2GCN(setup, input_dim=tensors["features"].n_cols, n_hops=2, l2_regularize_weight=5e-4).to("cuda:0")
        assert model.reuse_propagation == (reuse == "1")
        for layer in model.layer_objs:          # identical dropout streams in both runs
            if hasattr(layer, "seed"):
                layer.seed = 1234
        results[reuse] = _train_epochs(model, tensors, 6, pattern)
        if reuse == "1" and "te" in pattern * 2:
            assert model._prop_key is not None
    for a, b in zip(results["1"][0], results["0"][0]):
        assert torch.equal(a, b)
    assert torch.equal(results["1"][1], results["0"][1])


def test_fused_propagation_leaves_a_user_supplied_gradient_alone(tmp_path):
    """`out.backward(G)` with a caller-owned G: the propagation's backward must not accumulate into G's slots (it may only do
    that when the model vouches that the gradient is a private temporary, `private_grad=True` -- same bits either way)."""
    from h2gcn_amd import layers as L
    g, data, tensors, setup, model = _setup(tmp_path)
    plan = tensors["adj_hops"]
    torch.manual_seed(3)
    r0 = torch.randn(plan.n_cols, 16, device="cuda:0", requires_grad=True)
    grads = {}
    for private in (False, True):
        r0.grad = None
        out = L.fused_propagation(plan, r0, 2, private_grad=private)
        G = torch.randn(out.shape, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(5))
        keep = G.clone()
        out.backward(G)
        if not private:
            assert torch.equal(G, keep), "the caller's gradient tensor was modified"
        grads[private] = r0.grad.clone()
    assert torch.equal(grads[False], grads[True])


def test_dropout_dense_layers_of_one_model_draw_different_masks(tmp_path):
    from h2gcn_amd import layers as L
    torch.manual_seed(0)
    a, b = L.DropoutDense(64, 8, False, 0.5), L.DropoutDense(64, 8, False, 0.5)
    assert a.seed != b.seed
    g, data, tensors, setup, model = _setup(tmp_path, "M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-M32-D0.5-MO")
    seeds = [l.seed for l in model.layer_objs if isinstance(l, L.DropoutDense)]
    assert len(seeds) == 2 and seeds[0] != seeds[1]


def test_keras_adam_restores_into_its_existing_state_tensors():
    """load_state_dict (BestSnapshot.restore) must copy INTO m / v / the step counter: a captured training hipGraph holds
    their addresses."""
    from h2gcn_amd.optim import KerasAdam
    p = torch.nn.Parameter(torch.ones(8, device="cuda:0"))
    opt = KerasAdam([p], lr=0.1)
    p.grad = torch.ones_like(p)
    v0 = p._version
    opt.step()
    assert p._version > v0                      # the raw-pointer kernel told autograd's version counter
    saved = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict().items()}
    import copy
    saved = copy.deepcopy(opt.state_dict())
    ptrs = (opt.state[p]["m"].data_ptr(), opt.state[p]["v"].data_ptr(), opt.param_groups[0]["step_dev"].data_ptr())
    opt.step(); opt.step()
    opt.load_state_dict(saved)
    assert ptrs == (opt.state[p]["m"].data_ptr(), opt.state[p]["v"].data_ptr(), opt.param_groups[0]["step_dev"].data_ptr())
    assert int(opt.param_groups[0]["step_dev"].item()) == 1
    assert torch.equal(opt.state[p]["m"], saved["state"][0]["m"])


def test_entry_point_with_the_l2_penalty_folded_into_adam_follows_the_autograd_run(tmp_path, monkeypatch):
    """`run_experiments` on Cora: the keras l2 penalty's gradient folded into the optimizer launch + its value from one kernel
    (default) vs the penalty inside the autograd graph (H2GCN_FUSED_L2=0): same parameters -- hence accuracies -- and losses that
    agree to the rounding of the penalty's VALUE (fp64 inside a tensor vs torch's fp32 reduction)."""
    from test_entrypoints import _export_fixture
    from h2gcn_amd import run_experiments
    data_dir = tmp_path / "data"
    _export_fixture(load_planetoid_golden("cora"), data_dir, "ind.cora")
    stats = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("H2GCN_FUSED_L2", fused)
        args = run_experiments.main(["H2GCN", "planetoid", "--dataset", "ind.cora", "--dataset_path", str(data_dir), "--epochs", "25",
                                     "--random_seed", "7", "--network_setup", "M64-R-T1-G-V-T2-G-V-C1-C2-D0.0-MO", "--no_hipgraph"])
        stats[fused] = {k: float(v) for k, v in args.objects["epoch_stats"].items() if k != "monitor"}
    for k in ("train_acc", "val_acc", "test_accuracy", "test_loss"):
        assert stats["1"][k] == stats["0"][k], (k, stats)
    for k in ("train_loss", "val_loss"):
        assert abs(stats["1"][k] - stats["0"][k]) <= 2e-6 * abs(stats["0"][k]), (k, stats)


@pytest.mark.parametrize("name", ["cora", "citeseer"])
@pytest.mark.parametrize("network", ["M-R-T1-G0-V-T2-G0_1-V-C1_2-S1_0_32-D-MO", "D0.5-M64-R-T1-G1-V-C1-D0.5-MO"])
def test_replayed_training_equals_eager_on_hop_filter_and_dropout_networks(tmp_path, name, network):
    """Networks with hop filters (hop-subset launches: their long / segment-class lists are built lazily, during the eager warm-up
    epochs) and with dropout in front of the sparse embedding (SparseDropout refreshes the feature operand's values every step):
    40 epochs replayed as hipGraphs end in exactly the statistics of the eager loop -- on Cora (mixed classes: list-driven
    launches) and citeseer (short throughout, a few thousand rows: list-driven as well)."""
    from test_entrypoints import _export_fixture
    from h2gcn_amd import run_experiments
    data_dir = tmp_path / "data"
    _export_fixture(load_planetoid_golden(name), data_dir, f"ind.{name}")
    stats = {}
    for extra in ([], ["--no_hipgraph"]):
        args = run_experiments.main(["H2GCN", "planetoid", "--dataset", f"ind.{name}", "--dataset_path", str(data_dir), "--epochs", "40",
                                     "--random_seed", "3", "--network_setup", network] + extra)
        stats[bool(extra)] = {k: float(v) for k, v in args.objects["epoch_stats"].items() if k != "monitor"}
    assert stats[False] == stats[True], stats


def test_propagation_reuse_is_off_when_a_dropout_precedes_the_propagation(tmp_path):
    g, data, tensors, setup, model = _setup(tmp_path, "D0.5-M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO")
    assert not model.reuse_propagation
    g, data, tensors, setup, model = _setup(tmp_path, "M64-R-D0.5-T1-G-V-T2-G-V-C1-C2-MO")
    assert not model.reuse_propagation
    g, data, tensors, setup, model = _setup(tmp_path)
    assert model.reuse_propagation
