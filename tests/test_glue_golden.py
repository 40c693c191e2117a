"""CPU suite: the oracle's model interpreter (oracle/h2gcn_model.py) against tests/golden/glue_cora.* -- outputs of
the REFERENCE'S OWN `_layers.py` / `H2GCN.py` / `_metrics.py` (imported in place by tests/golden/make_golden.py and run
under a numpy/scipy stand-in for TensorFlow).  This pins every restated line of the interpreter glue (stack axis, hop
filter, concat order, slices, tag store, bias order, loss/accuracy/L2); TF's SpMM kernel itself stays unpinned (the
stand-in multiplies with scipy).  The GPU suite checks the HIP path against the same fixture."""
import json

import numpy as np
import pytest

from conftest import GOLDEN, golden_weight, load_planetoid_golden
from oracle import h2gcn_model as om

META = json.loads((GOLDEN / "glue_cora.json").read_text())


def glue_weights(entry):
    return [golden_weight(i, tuple(shape), kind) for i, (kind, shape) in enumerate(entry["weights"])]


def _encode(setup):
    out = []
    for kind, conf in setup:
        c = {}
        for k, v in conf.items():
            if isinstance(v, set):
                v = {"__set__": sorted(v)}
            elif isinstance(v, slice):
                v = {"__slice__": [v.start, v.stop, v.step]}
            c[k] = v
        out.append([kind, c])
    return out


@pytest.mark.parametrize("idx", range(len(META["entries"])))
def test_oracle_interpreter_reproduces_reference_glue(idx):
    from h2gcn_amd.models import parse_network_setup

    entry = META["entries"][idx]
    z = np.load(GOLDEN / "glue_cora.npz")
    g = load_planetoid_golden("cora")
    setup = _encode(parse_network_setup(entry["network"], 7, _dense_units=64, _dropout_rate=0.5))
    weights = glue_weights(entry)
    logits, tagged, _ = om.forward(setup, g["feat_rownorm"], [g["hop1_sym"], g["hop2_sym"]], weights, return_tagged=True)
    want = z[f"n{idx}_logits"]
    assert logits.shape == want.shape
    assert np.abs(logits - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    assert set(tagged) == set(entry["tags"])
    for tag, shape in entry["tags"].items():
        v = np.asarray(tagged[tag])
        assert list(v.shape) == shape, tag                        # [N, H_sel, d] before V, [N, w] after
        assert np.abs(v[z["rows"]] - z[f"n{idx}_tag{tag}_rows"]).max() <= 2e-6
        assert np.allclose(v.reshape(g["n"], -1).sum(0), z[f"n{idx}_tag{tag}_colsum64"], rtol=0, atol=2e-4)
    labels = (g["y_all"] * g["train_mask"][:, None]).astype(np.float64)
    reg = META["l2"] * sum(float((w.astype(np.float64) ** 2).sum()) for (kind, _), w in zip(entry["weights"], weights) if kind == "kernel")
    assert abs(reg - entry["l2_term"]) <= 1e-6
    assert abs(om.masked_softmax_cross_entropy(logits, labels, g["train_mask"]) + reg - entry["loss"]) <= 2e-6
    assert abs(om.masked_accuracy(want.astype(np.float64), labels, g["train_mask"]) - entry["train_acc"]) <= 1e-6
