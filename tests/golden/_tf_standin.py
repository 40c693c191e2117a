"""A numpy/scipy stand-in for the handful of `tf.*` symbols the reference's model code touches, so that
tests/golden/make_golden.py can IMPORT AND RUN the reference's own `h2gcn/models/_layers.py`, `H2GCN.py` and
`_metrics.py` in the build container (TensorFlow is absent and uninstallable there).  TEST INFRASTRUCTURE ONLY:
used by make_golden.py, never by the product, never on the GPU box.

What this pins and what it does not
-----------------------------------
It pins the reference's GLUE, executed from the reference's own source text: `GCNLayer.call` (stack axis, hop filter,
`_layers.py:78-81`), `ConcatLayer.call` (inputs first, then tags in production order, `:90-96`), `SliceLayer`,
`SparseDense.call` (bias/activation order), `H2GCN.__init__` (layer table, tag bookkeeping) and `H2GCN.call`
(`H2GCN.py:294-346`), `masked_softmax_cross_entropy` / `masked_accuracy` (`_metrics.py`).
It does NOT pin TensorFlow's own kernels: `sparse_dense_matmul` below is scipy's `csr @ dense` in fp32 (the same
loop nest as TF's CPU kernel as far as anybody can tell without TF), `Dense` is numpy `@`.  The arithmetic of
SURVEY.md row a3 therefore stays "parity unpinned against TensorFlow".

Weights are not random-initialised the Keras way: every `add_weight` draws from conftest.golden_weight(index, shape),
so tests can regenerate them without storing megabytes.
"""
import sys
import types

import numpy as np
import scipy.sparse as sp

_WEIGHT_FN = None      # set by make_golden: (index, shape, kind) -> np.ndarray
_WEIGHT_LOG = []       # (kind, shape) in creation order


def reset_weights(fn):
    global _WEIGHT_FN
    _WEIGHT_FN = fn
    _WEIGHT_LOG.clear()


class SparseTensor:
    """tf.SparseTensor stand-in around a scipy CSR (row-major, sorted = `tf.sparse.reorder` order)."""

    def __init__(self, csr):
        self.csr = sp.csr_matrix(csr).astype(np.float32)
        self.csr.sort_indices()
        self.values = self.csr.data
        self.dense_shape = np.array(self.csr.shape, dtype=np.int64)
        self.shape = self.csr.shape

    @property
    def indices(self):
        coo = self.csr.tocoo()
        return np.stack([coo.row, coo.col], 1).astype(np.int64)


def _function(fn=None, **_kw):
    return fn if fn is not None else (lambda f: f)


class _Layer:
    def __init__(self, *a, **k):
        self._built = False
        self._losses = []
        self.name = type(self).__name__.lower()

    def add_weight(self, name=None, shape=None, regularizer=None, initializer=None, **_k):
        kind = "bias" if name == "bias" else "kernel"
        w = _WEIGHT_FN(len(_WEIGHT_LOG), tuple(int(s) for s in shape), kind)
        _WEIGHT_LOG.append((kind, tuple(int(s) for s in shape)))
        if regularizer is not None:
            self._losses.append((regularizer, w))
        return w

    def build(self, input_shape):
        self._built = True

    def __call__(self, *args, **kwargs):
        if not self._built:
            shape = getattr(args[0], "shape", None) if args else None
            self.build(tuple(shape) if shape is not None else None)
            self._built = True
        return self.call(*args, **kwargs)


class _Dense(_Layer):
    def __init__(self, units, use_bias=True, kernel_regularizer=None, **_k):
        super().__init__()
        self.units, self.use_bias, self.kernel_regularizer = units, use_bias, kernel_regularizer

    def build(self, input_shape):
        self.kernel = self.add_weight(name="kernel", shape=[int(input_shape[-1]), self.units], regularizer=self.kernel_regularizer)
        if self.use_bias:
            self.bias = self.add_weight(name="bias", shape=[self.units])

    def call(self, x):
        y = np.asarray(x, dtype=np.float32) @ self.kernel
        return y + self.bias if self.use_bias else y


class _Dropout(_Layer):
    def __init__(self, rate, **_k):
        super().__init__()
        self.rate = rate

    def call(self, x, training=False):
        return x  # inference


class _ReLU(_Layer):
    def call(self, x):
        return np.maximum(x, np.float32(0))


class _Flatten(_Layer):
    def call(self, x):
        return np.reshape(x, (x.shape[0], -1))


class _Model(_Layer):
    def __call__(self, *args, **kwargs):
        return self.call(*args, **kwargs)

    @property
    def losses(self):
        out = []
        for obj in getattr(self, "layer_objs", []):
            for reg, w in getattr(obj, "_losses", []):
                out.append(reg(w))
        return out


class _L2:
    def __init__(self, l2):
        self.l2 = l2

    def __call__(self, w):
        return np.float32(self.l2) * np.sum(np.square(w), dtype=np.float32)


def _softmax_xent(logits=None, labels=None):
    z = logits - logits.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    return -(labels * logp).sum(1)


def make_module():
    tf = types.ModuleType("tensorflow")
    tf.__standin__ = True
    tf.float32, tf.bool, tf.int64 = np.float32, np.bool_, np.int64
    tf.SparseTensor = SparseTensor
    tf.function = _function
    tf.is_tensor = lambda x: isinstance(x, (np.ndarray, SparseTensor))
    tf.stack = lambda xs, axis=0: np.stack(list(xs), axis=axis)
    tf.concat = lambda xs, axis=0: np.concatenate(list(xs), axis=axis)
    tf.split = lambda x, sizes, axis=0: np.split(x, np.cumsum(sizes)[:-1], axis=axis)
    tf.reduce_sum = lambda x, axis=None: np.sum(x, axis=axis, dtype=np.float32)
    tf.cast = lambda x, dtype=None: np.asarray(x).astype(dtype)
    tf.floor = np.floor
    tf.equal = np.equal
    tf.argmax = lambda x, axis=None: np.argmax(x, axis=axis)
    tf.stop_gradient = lambda x: x
    tf.zeros_initializer = "zeros"

    sparse = types.ModuleType("tensorflow.sparse")
    sparse.SparseTensor = SparseTensor
    sparse.sparse_dense_matmul = lambda a, b: np.asarray(a.csr @ np.asarray(b, dtype=np.float32), dtype=np.float32)
    sparse.to_dense = lambda a: np.asarray(a.csr.todense(), dtype=np.float32)
    sparse.reorder = lambda a: a
    tf.sparse = sparse

    tf.math = types.SimpleNamespace(add_n=lambda xs: np.sum(np.asarray(list(xs), dtype=np.float32), dtype=np.float32))
    tf.nn = types.SimpleNamespace(softmax_cross_entropy_with_logits=_softmax_xent)
    tf.random = types.SimpleNamespace(uniform=lambda shape: np.zeros(shape, dtype=np.float32))
    tf.config = types.SimpleNamespace(experimental=types.SimpleNamespace(list_physical_devices=lambda kind=None: []))
    tf.train = types.SimpleNamespace(Checkpoint=lambda **k: None)

    keras = types.ModuleType("tensorflow.keras")
    keras.Model = _Model
    keras.layers = types.SimpleNamespace(Layer=_Layer, Dense=_Dense, Dropout=_Dropout, ReLU=_ReLU, Flatten=_Flatten)
    keras.regularizers = types.SimpleNamespace(l2=_L2)
    keras.optimizers = types.SimpleNamespace(get=lambda *a, **k: None)
    tf.keras = keras
    return tf


def install():
    tf = make_module()
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.keras"] = tf.keras
    sys.modules["tensorflow.sparse"] = tf.sparse
    return tf
