"""Model plugins + the ``--network_setup`` DSL (mirror of the reference's ``h2gcn/models/__init__.py``).

* plugin discovery: every non-underscore module of this package is a selectable positional ``model``; it must
  expose ``add_subparser_args(parser)`` (reference ``h2gcn/models/__init__.py:16-31``);
* ``Layer``: the layer-kind tokens (``:34-44``);
* ``parse_network_setup``: the dash-separated network description, e.g. H2GCN-2 =
  ``M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO`` (``:47-150``).  Same output structure as the reference -- a list of
  ``(kind, conf)`` -- checked against ``tests/golden/dsl_parse.json`` (produced by the reference's parser).

Grammar: tokens are separated by ``-`` (not inside ``[...]``).  Layer tokens: ``F<n>|F|FO`` dense with bias,
``M<n>|M|MO`` dense without bias (``O`` = output width, marks the start of the output network), ``D<p>|D``
dropout, ``G`` / ``G0_1`` hop aggregation (optional hop filter), ``C<tag>_<tag>`` concat with tagged outputs,
``R`` ReLU, ``V`` flatten, ``I`` sparse->dense identity, ``S<tag>_<start>_<stop>[_<step>]`` column slice,
``X<name>_<conf>`` experimental, ``lambda...``.  Modifier tokens attach to the previous layer: ``E`` embedding,
``L`` supervised, ``T<tag>`` tag.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional, Tuple


class Layer:
    DENSE = "F"
    DROPOUT = "D"
    GCN = "G"
    RELU = "R"
    CONCAT = "C"
    VECTORIZE = "V"
    IDENTITY = "I"
    SLICE = "S"
    EXPERIMENTAL = "X"
    LAMBDA = "lambda"


_SPLIT = re.compile(r"-(?![^\[]*\])")


def _int_or_none(text: str) -> Optional[int]:
    return int(text) if text else None


def parse_network_setup(network_setup_str: str, output_dim: int, _dense_units: Optional[int] = None,
                        _dropout_rate: Optional[float] = None, parse_preprocessing: bool = False
                        ) -> List[Tuple[str, Dict[str, Any]]]:
    conf: List[Tuple[str, Dict[str, Any]]] = []
    embedding_seen = False
    for tok in _SPLIT.split(network_setup_str):
        if tok.startswith("[") and tok.endswith("]"):
            tok = tok[1:-1].strip()
        head, rest = tok[0], tok[1:]
        if tok.startswith("lambda"):
            conf.append((Layer.LAMBDA, {"lambda": tok}))
        elif head in "FM":
            extra: Dict[str, Any] = {}
            if rest == "O":
                units = output_dim
                extra["beginOutput"] = True
            elif rest:
                units = int(rest)
            else:
                if _dense_units is None:
                    raise AssertionError("dense layer without a width needs a default (--hidden)")
                units = _dense_units
            conf.append((Layer.DENSE, dict(units=units, use_bias=(head == "F"), **extra)))
        elif head == "D":
            if rest:
                rate = float(rest)
            else:
                if _dropout_rate is None:
                    raise AssertionError("dropout layer without a rate needs a default (--dropout)")
                rate = _dropout_rate
            conf.append((Layer.DROPOUT, dict(dropout_rate=rate)))
        elif head == "G":
            conf.append((Layer.GCN, dict(hops={int(i) for i in rest.split("_")} if rest else None)))
        elif head == "C":
            conf.append((Layer.CONCAT, dict(tags=rest.split("_"), addInputs=True)))
        elif head == "R":
            conf.append((Layer.RELU, {}))
        elif head == "V":
            conf.append((Layer.VECTORIZE, {}))
        elif head == "I":
            conf.append((Layer.IDENTITY, {}))
        elif head == "S":
            parts = rest.split("_")
            tag = parts[0] or None
            bounds = parts[1:]
            conf.append((Layer.SLICE, dict(loadTag=tag,
                                           sliceObj=slice(*[_int_or_none(b) for b in bounds]) if bounds else slice(None))))
        elif head == "X":
            name, _, xconf = rest.partition("_")
            conf.append((Layer.EXPERIMENTAL, dict(name=name, conf=xconf, output_dim=output_dim)))
        elif head == "E":
            if embedding_seen:
                raise AssertionError("only one layer can be the embedding (E)")
            conf[-1][1]["isEmbedding"] = True
            embedding_seen = True
        elif head == "L":
            conf[-1][1]["supervised"] = True
        elif head == "T":
            conf[-1][1]["tag"] = rest
        else:
            raise ValueError(f"Unknown layer config {tok} in network config {network_setup_str}")
    return conf


def add_subparsers(parser):
    """Positional ``model`` + the chosen model's own flags (reference ``:16-31``)."""
    import sys

    from .._plugins import register_positional

    return register_positional(parser, sys.modules[__name__], "model", "Network model selected for experiment",
                               announce=True)


def toNumpy(x):
    """Tensor -> numpy (reference ``toNumpy``, ``:153-161``); a HopPlan exports its CSR arrays."""
    from ..hops import HopPlan

    if isinstance(x, HopPlan):
        return [{"indptr": rp.cpu().numpy(), "indices": ci.cpu().numpy(), "values": va.cpu().numpy(),
                 "dense_shape": (x.n_rows, x.n_cols)} for rp, ci, va in zip(x.rowptr, x.colidx, x.vals)]
    return x.detach().cpu().numpy()
