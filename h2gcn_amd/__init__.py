"""h2gcn_amd -- MI355X-native hop aggregation for H2GCN (the GCNLayer path of GemsLab/H2GCN).

The package is a thin PyTorch-ROCm front end over ``libh2gcn_hip.so`` (hand-written gfx950 HIP behind the
C ABI of ``include/h2gcn_hip.h``).  It mirrors the reference's operator interface for this one path:

* :class:`h2gcn_amd.layers.GCNLayer` -- ``layer(adjhops, inputs) -> [N, H, d]``
  (reference ``h2gcn/models/_layers.py:54-81``);
* :class:`h2gcn_amd.hops.HopPlan` -- the device-resident ``adj_hops`` operand list
  (reference ``h2gcn/datasets/_dataset.py:559-576``);
* :mod:`h2gcn_amd.operands` -- construction of the normalised exact-k-hop matrices: on the device with the HIP ring
  kernels (``build_adj_norm_hops_device``) or on the host with scipy as the reference does
  (reference ``h2gcn/datasets/_dataset.py:102-158``);
* :mod:`h2gcn_amd.partition` -- row partitioning (equal or nnz-balanced blocks) + embedding exchange (RCCL all-gather or
  the library's IPC pulls) for 1..8 GPUs (new; the reference is single-device);
* :class:`h2gcn_amd.layers.DropoutDense` -- keras ``Dropout`` + output ``Dense`` (``D0.5-MO``) as one pass over the concat
  buffer per direction (reference ``h2gcn/models/H2GCN.py:235-257``);
* :mod:`h2gcn_amd.metrics` -- masked softmax cross-entropy / accuracy in one pass over the logits
  (reference ``h2gcn/models/_metrics.py:8-25``);
* :class:`h2gcn_amd.optim.KerasAdam` -- the reference's optimizer step (Keras / TensorFlow Adam arithmetic) as one launch for
  all parameters (reference ``h2gcn/models/H2GCN.py:62-63, 73``).

Results of the aggregation are bit-reproducible functions of the operands (one canonical summation tree in every kernel):
any slicing, chunking or row partition gives the same bits.

There is no CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""

__version__ = "0.6.0"

from . import _capi  # noqa: F401  (does not load the library until first use)
from .hops import HopPlan  # noqa: F401
from .layers import ConcatLayer, DropoutDense, GCNLayer, SliceLayer, SparseDense, SparseDropout, hop_spmm  # noqa: F401
