"""CPU suite: the network-setup DSL parser against the reference parser's output (tests/golden/dsl_parse.json)."""
import json

import pytest

from conftest import GOLDEN
from h2gcn_amd.models import Layer, parse_network_setup


def _enc(v):
    if isinstance(v, slice):
        return {"__slice__": [v.start, v.stop, v.step]}
    if isinstance(v, set):
        return {"__set__": sorted(v)}
    return v


def test_matches_reference_parser():
    golden = json.loads((GOLDEN / "dsl_parse.json").read_text())
    assert len(golden) == 16
    for text, want in golden.items():
        got = parse_network_setup(text, 7, _dense_units=64, _dropout_rate=0.5, parse_preprocessing=True)
        assert [[t, {k: _enc(v) for k, v in c.items()}] for t, c in got] == want, text


def test_h2gcn2_default_structure():
    p = parse_network_setup("M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO", 7)
    assert [t for t, _ in p] == [Layer.DENSE, Layer.RELU, Layer.GCN, Layer.VECTORIZE, Layer.GCN, Layer.VECTORIZE,
                                 Layer.CONCAT, Layer.CONCAT, Layer.DROPOUT, Layer.DENSE]
    assert p[1][1]["tag"] == "1" and p[3][1]["tag"] == "2" and p[-1][1]["beginOutput"]


def test_errors():
    with pytest.raises(ValueError):
        parse_network_setup("M64-Q", 7)
    with pytest.raises(AssertionError):
        parse_network_setup("M-R", 7)            # no default width
    with pytest.raises(AssertionError):
        parse_network_setup("M64-E-R-E", 7)      # two embeddings
    assert parse_network_setup("[M64]-R", 3)[0] == (Layer.DENSE, dict(units=64, use_bias=False))
