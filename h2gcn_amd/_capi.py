"""ctypes binding of libh2gcn_hip.so -- the C ABI declared in include/h2gcn_hip.h.

Nothing here computes: it loads the in-tree shared library, declares the prototypes, and turns negative
status codes into Python exceptions carrying ``h2gcn_last_error()``.  If the library is missing the import of
this module still succeeds (so CPU-only tooling can import the package) but the first call to :func:`lib`
raises -- the product path never falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

ABI_VERSION = 4
MAX_HOPS = 8

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_HIP = -2
ERR_OUT_OF_MEMORY = -3
ERR_BAD_INDEX = -4
ERR_NO_TRANSPOSE = -5
ERR_INTERNAL = -6
ERR_EXCHANGE_TIMEOUT = -7

METRICS_MAX_SETS = 4
XCHG_BLOB_BYTES = 192
XCHG_COPY_ENGINE = 0
XCHG_COPY_KERNEL = 1

PLAN_BUILD_TRANSPOSE = 0x1
PLAN_SKIP_VALIDATION = 0x2
PLAN_HOST_TRANSPOSE = 0x4
PLAN_KEEP_PERMUTATION = 0x8

#: every symbol include/h2gcn_hip.h declares (tests check the built library exports all of them)
EXPORTED_SYMBOLS = (
    "h2gcn_abi_version",
    "h2gcn_last_error",
    "h2gcn_device_count",
    "h2gcn_plan_create",
    "h2gcn_plan_destroy",
    "h2gcn_plan_info",
    "h2gcn_plan_set_values",
    "h2gcn_spmm_hops_f32",
    "h2gcn_spmm_hops_T_f32",
    "h2gcn_plan_schedule",
    "h2gcn_plan_segment_classes",
    "h2gcn_spmm_workspace_bytes",
    "h2gcn_spmm_hops_opts_f32",
    "h2gcn_spmm_hops_T_opts_f32",
    "h2gcn_ring_scratch_bytes",
    "h2gcn_ring_count",
    "h2gcn_ring_fill",
    "h2gcn_ring_count_rows",
    "h2gcn_ring_fill_rows",
    "h2gcn_hop_normalize",
    "h2gcn_hop_normalize_rows",
    "h2gcn_dropout_dense_small_rows",
    "h2gcn_dropout_dense_workspace_bytes",
    "h2gcn_dropout_dense_f32",
    "h2gcn_dropout_dense_backward_f32",
    "h2gcn_masked_metrics_workspace_bytes",
    "h2gcn_masked_metrics_f32",
    "h2gcn_masked_ce_backward_f32",
    "h2gcn_adam_keras_f32",
    "h2gcn_adam_keras_l2_f32",
    "h2gcn_l2_penalty_workspace_bytes",
    "h2gcn_l2_penalty_f32",
    "h2gcn_xchg_create",
    "h2gcn_xchg_export",
    "h2gcn_xchg_connect",
    "h2gcn_xchg_allgather_begin",
    "h2gcn_xchg_allgather_post",
    "h2gcn_xchg_allgather_pull",
    "h2gcn_xchg_allgather_pull_rows",
    "h2gcn_xchg_allgather_end",
    "h2gcn_xchg_reduce_scatter_begin",
    "h2gcn_xchg_reduce_scatter_end",
    "h2gcn_xchg_reset_dependencies",
    "h2gcn_xchg_status",
    "h2gcn_xchg_destroy",
)


class PlanOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("long_row_threshold", C.c_int32),
        ("rows_per_wave", C.c_int32),
        ("variant", C.c_int32),
        ("slice_cols", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


LAUNCH_RELU = 0x1
LAUNCH_ACCUMULATE = 0x2


class LaunchOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
        ("bias", C.c_void_p),
    ]


class H2GCNError(RuntimeError):
    """A call into libh2gcn_hip.so failed; ``status`` is the negative h2gcn_status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libh2gcn_hip: {message} (status {status})")
        self.status = status


def library_path() -> Path:
    override = os.environ.get("H2GCN_HIP_LIBRARY")
    if override:
        return Path(override)
    return Path(__file__).resolve().parent / "csrc" / "libh2gcn_hip.so"


_LIB = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C h2gcn_amd/csrc`.  h2gcn_amd has no CPU fallback."
        )
    L = C.CDLL(str(path))
    L.h2gcn_abi_version.restype = C.c_int
    L.h2gcn_abi_version.argtypes = []
    L.h2gcn_last_error.restype = C.c_char_p
    L.h2gcn_last_error.argtypes = []
    L.h2gcn_device_count.restype = C.c_int
    L.h2gcn_device_count.argtypes = []
    L.h2gcn_plan_create.restype = C.c_int
    L.h2gcn_plan_create.argtypes = [
        C.c_int, C.c_int64, C.c_int64,
        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
        C.POINTER(PlanOpts), C.c_void_p, C.POINTER(C.c_void_p),
    ]
    L.h2gcn_plan_destroy.restype = None
    L.h2gcn_plan_destroy.argtypes = [C.c_void_p]
    L.h2gcn_plan_set_values.restype = C.c_int
    L.h2gcn_plan_set_values.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.h2gcn_plan_info.restype = C.c_int
    L.h2gcn_plan_info.argtypes = [
        C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
        C.POINTER(C.c_int64), C.POINTER(C.c_int32),
    ]
    L.h2gcn_spmm_hops_f32.restype = C.c_int
    L.h2gcn_spmm_hops_f32.argtypes = [
        C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
    ]
    L.h2gcn_spmm_hops_T_f32.restype = C.c_int
    L.h2gcn_spmm_hops_T_f32.argtypes = [
        C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
    ]
    L.h2gcn_plan_schedule.restype = C.c_int
    L.h2gcn_plan_schedule.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int64, C.c_int32] + [C.POINTER(C.c_int32)] * 4
    if hasattr(L, "h2gcn_plan_segment_classes"):   # (absent from pre-ABI-4 builds loaded for A/B runs)
        L.h2gcn_plan_segment_classes.restype = C.c_int
        L.h2gcn_plan_segment_classes.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int64, C.c_int32] + [C.POINTER(C.c_int64)] * 3
    L.h2gcn_spmm_workspace_bytes.restype = C.c_size_t
    L.h2gcn_spmm_workspace_bytes.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int32]
    L.h2gcn_spmm_hops_opts_f32.restype = C.c_int
    L.h2gcn_spmm_hops_opts_f32.argtypes = [
        C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64,
        C.POINTER(LaunchOpts), C.c_void_p,
    ]
    L.h2gcn_spmm_hops_T_opts_f32.restype = C.c_int
    L.h2gcn_spmm_hops_T_opts_f32.argtypes = [
        C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
        C.POINTER(LaunchOpts), C.c_void_p,
    ]
    L.h2gcn_ring_scratch_bytes.restype = C.c_size_t
    L.h2gcn_ring_scratch_bytes.argtypes = [C.c_int64]
    ring_common = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int,
                   C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]
    L.h2gcn_ring_count.restype = C.c_int
    L.h2gcn_ring_count.argtypes = ring_common + [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_size_t, C.c_void_p]
    L.h2gcn_ring_fill.restype = C.c_int
    L.h2gcn_ring_fill.argtypes = ring_common + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    ring_rows = [C.c_int64, C.c_int64, C.c_int64] + ring_common[1:]      # (n, row_begin, n_rows, a_rowptr, ...)
    L.h2gcn_ring_count_rows.restype = C.c_int
    L.h2gcn_ring_count_rows.argtypes = ring_rows + [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_size_t, C.c_void_p]
    L.h2gcn_ring_fill_rows.restype = C.c_int
    L.h2gcn_ring_fill_rows.argtypes = ring_rows + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.h2gcn_hop_normalize_rows.restype = C.c_int
    L.h2gcn_hop_normalize_rows.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.h2gcn_hop_normalize.restype = C.c_int
    L.h2gcn_hop_normalize.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    if hasattr(L, "h2gcn_dropout_dense_small_rows"):
        L.h2gcn_dropout_dense_small_rows.restype = C.c_int64
        L.h2gcn_dropout_dense_small_rows.argtypes = [C.c_int64]
    L.h2gcn_dropout_dense_workspace_bytes.restype = C.c_size_t
    L.h2gcn_dropout_dense_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    L.h2gcn_dropout_dense_f32.restype = C.c_int
    L.h2gcn_dropout_dense_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_float,
                                          C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    L.h2gcn_dropout_dense_backward_f32.restype = C.c_int
    L.h2gcn_dropout_dense_backward_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                                   C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                   C.c_size_t, C.c_void_p]
    L.h2gcn_masked_metrics_workspace_bytes.restype = C.c_size_t
    L.h2gcn_masked_metrics_workspace_bytes.argtypes = [C.c_int64]
    L.h2gcn_masked_metrics_f32.restype = C.c_int
    L.h2gcn_masked_metrics_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.h2gcn_masked_ce_backward_f32.restype = C.c_int
    L.h2gcn_masked_ce_backward_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int64, C.c_void_p]
    L.h2gcn_adam_keras_f32.restype = C.c_int
    L.h2gcn_adam_keras_f32.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_int64), C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    if hasattr(L, "h2gcn_adam_keras_l2_f32"):   # (absent from pre-ABI-4 builds loaded for A/B runs)
        L.h2gcn_adam_keras_l2_f32.restype = C.c_int
        L.h2gcn_adam_keras_l2_f32.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                              C.c_int64, C.c_void_p]
        L.h2gcn_l2_penalty_workspace_bytes.restype = C.c_size_t
        L.h2gcn_l2_penalty_workspace_bytes.argtypes = []
        L.h2gcn_l2_penalty_f32.restype = C.c_int
        L.h2gcn_l2_penalty_f32.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p]
    L.h2gcn_xchg_create.restype = C.c_int
    L.h2gcn_xchg_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.h2gcn_xchg_export.restype = C.c_int
    L.h2gcn_xchg_export.argtypes = [C.c_void_p, C.c_void_p]
    L.h2gcn_xchg_connect.restype = C.c_int
    L.h2gcn_xchg_connect.argtypes = [C.c_void_p, C.c_void_p]
    L.h2gcn_xchg_allgather_begin.restype = C.c_int
    L.h2gcn_xchg_allgather_begin.argtypes = [
        C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
    ]
    L.h2gcn_xchg_allgather_post.restype = C.c_int
    L.h2gcn_xchg_allgather_post.argtypes = L.h2gcn_xchg_allgather_begin.argtypes
    L.h2gcn_xchg_allgather_pull.restype = C.c_int
    L.h2gcn_xchg_allgather_pull.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_void_p]
    if hasattr(L, "h2gcn_xchg_allgather_pull_rows"):
        L.h2gcn_xchg_allgather_pull_rows.restype = C.c_int
        L.h2gcn_xchg_allgather_pull_rows.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.h2gcn_xchg_allgather_end.restype = C.c_int
    L.h2gcn_xchg_allgather_end.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.h2gcn_xchg_reduce_scatter_begin.restype = C.c_int
    L.h2gcn_xchg_reduce_scatter_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.h2gcn_xchg_reduce_scatter_end.restype = C.c_int
    L.h2gcn_xchg_reduce_scatter_end.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.h2gcn_xchg_reset_dependencies.restype = C.c_int
    L.h2gcn_xchg_reset_dependencies.argtypes = [C.c_void_p]
    L.h2gcn_xchg_status.restype = C.c_int
    L.h2gcn_xchg_status.argtypes = [C.c_void_p]
    L.h2gcn_xchg_destroy.restype = None
    L.h2gcn_xchg_destroy.argtypes = [C.c_void_p]
    got = L.h2gcn_abi_version()
    # An explicitly named ABI-3 build (H2GCN_HIP_LIBRARY: the interleaved A/B tools time an older kernel) is accepted.  It lacks
    # what later rounds ADDED -- h2gcn_plan_segment_classes, h2gcn_adam_keras_l2_f32 / h2gcn_l2_penalty_*,
    # h2gcn_xchg_allgather_pull_rows -- and every caller of those asks `has()` first: the front end then keeps the l2 penalty in
    # the autograd graph, pulls whole shards, and HopPlan.segment_classes raises a message instead of an AttributeError.
    if got != ABI_VERSION and not (os.environ.get("H2GCN_HIP_LIBRARY") and got == 3):
        raise RuntimeError(f"{path}: ABI version {got}, this front end expects {ABI_VERSION}")
    _LIB = L
    return L


def has(symbol: str) -> bool:
    """Does the loaded library export ``symbol``?  (False only for an older build named through H2GCN_HIP_LIBRARY.)"""
    return hasattr(lib(), symbol)


def check(status: int) -> None:
    """Raise :class:`H2GCNError` for a negative status."""
    if status < 0:
        msg = lib().h2gcn_last_error()
        raise H2GCNError(status, msg.decode("utf-8", "replace") if msg else "unknown error")
