"""CPU restatement of the reference's optimizer step -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference builds ``keras.optimizers.get("adam").from_config({"lr": lr})`` (``h2gcn/models/H2GCN.py:62-63``) and calls
``optimizer.apply_gradients`` (``:73``).  The arithmetic lives in a third-party dependency that is absent from
``/root/reference`` (TensorFlow >= 2.0, ``README.md:34``, no pinned version): Keras' ``Adam._resource_apply_dense`` hands
``lr``, ``beta_1**t``, ``beta_2**t`` (fp32 tensors) to TensorFlow's ``ApplyAdam`` kernel, whose published update is restated
here in fp32, operation by operation.  Parity unpinned against TensorFlow itself (it cannot be imported here); pinned by the
hand-worked first step in ``tests/test_oracle_tree_and_classifier.py``.
"""
import numpy as np


def keras_adam_step(param, grad, m, v, t, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    """One update, in place on copies: returns ``(param, m, v)`` as fp32 arrays.  ``t`` is the 1-based step."""
    f = np.float32
    p, g, m, v = (np.array(a, dtype=f, copy=True) for a in (param, grad, m, v))
    b1p, b2p = np.power(f(beta_1), f(t), dtype=f), np.power(f(beta_2), f(t), dtype=f)
    alpha = f(lr) * np.sqrt(f(1) - b2p, dtype=f) / (f(1) - b1p)
    m += (g - m) * (f(1) - f(beta_1))
    v += (g * g - v) * (f(1) - f(beta_2))
    p -= (m * alpha) / (np.sqrt(v) + f(epsilon))
    return p, m, v
