"""CPU restatement (numpy, float64 or float32) of the reference's model interpreter for the layer kinds on the
hot path's call chain.  TEST INFRASTRUCTURE ONLY.

Follows ``H2GCN.call`` (reference ``h2gcn/models/H2GCN.py:294-346``): layers are executed in order; ``C`` layers
concatenate ``[inputs] + tagged outputs (in production order)`` (``_layers.py:90-96``); ``G`` layers stack the hop
products on axis -2 (``_layers.py:78-81``); ``V`` flattens; the first dense layer consumes the sparse features
(``SparseDense``, ``_layers.py:45-52``); dropout is the identity at inference; outputs tagged ``T<name>`` are stored
(``:339-341``).  ``layer_setups`` is the parsed network setup (list of ``[kind, conf]``), e.g. an entry of
``tests/golden/dsl_parse.json``; ``weights`` lists the dense kernels (and biases) in layer order.
"""
import numpy as np
import scipy.sparse as sp


def forward(layer_setups, features, hops, weights, dtype=np.float64, return_tagged=False):
    x = sp.csr_matrix(features).astype(dtype)
    sparse_input = True
    hops = [sp.csr_matrix(h).astype(dtype) for h in hops]
    w_iter = iter(weights)
    tagged = {}
    trace = []
    for kind, conf in layer_setups:
        if kind == "F":
            kernel = np.asarray(next(w_iter), dtype=dtype)
            x = (x @ kernel) if sparse_input else (x @ kernel)
            x = np.asarray(x)
            sparse_input = False
            if conf.get("use_bias"):
                x = x + np.asarray(next(w_iter), dtype=dtype)
        elif kind == "R":
            x = np.maximum(x, 0)
        elif kind == "G":
            sel = conf.get("hops")
            if isinstance(sel, dict):
                sel = set(sel["__set__"])
            x = np.stack([h @ x for i, h in enumerate(hops) if sel is None or i in sel], axis=-2)
        elif kind == "V":
            x = x.reshape(x.shape[0], -1)
        elif kind == "C":
            picked = [v for name, v in tagged.items() if name in conf["tags"]]
            x = np.concatenate(([x] if conf.get("addInputs", True) else []) + picked, axis=-1)
        elif kind == "D":
            pass
        elif kind == "I":
            x = np.asarray(x.todense())
            sparse_input = False
        elif kind == "S":
            src = tagged[conf["loadTag"]] if conf["loadTag"] else x
            so = conf["sliceObj"]
            if isinstance(so, dict):
                so = slice(*so["__slice__"])
            x = src[:, so]
        else:
            raise ValueError(kind)
        if "tag" in conf:
            tagged[conf["tag"]] = x
        trace.append(x)
    return (x, tagged, trace) if return_tagged else x


def masked_softmax_cross_entropy(preds, labels_onehot, mask):
    """reference ``h2gcn/models/_metrics.py:8-15``"""
    z = preds - preds.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    loss = -(labels_onehot * logp).sum(1)
    m = mask.astype(preds.dtype)
    return float((loss * (m / m.sum())).sum())


def masked_accuracy(preds, labels_onehot, mask):
    """reference ``h2gcn/models/_metrics.py:17-25``"""
    correct = (preds.argmax(1) == labels_onehot.argmax(1)).astype(preds.dtype)
    m = mask.astype(preds.dtype)
    return float((correct * (m / m.sum())).sum())
