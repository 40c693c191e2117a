"""CPU suite: the C-ABI library builds, loads, and exports exactly what include/h2gcn_hip.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import re
from pathlib import Path

import pytest

from h2gcn_amd import _capi

ROOT = Path(__file__).resolve().parents[1]


def _declared_functions():
    text = (ROOT / "include" / "h2gcn_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(h2gcn_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(_capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(_capi.library_path()))
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/h2gcn_hip.h but not exported"


def test_abi_version_and_error_channel():
    L = _capi.lib()
    assert L.h2gcn_abi_version() == _capi.ABI_VERSION
    assert isinstance(L.h2gcn_last_error(), bytes)
    # status codes mirror the header
    text = (ROOT / "include" / "h2gcn_hip.h").read_text()
    for name, val in (("H2GCN_ERR_INVALID_ARGUMENT", -1), ("H2GCN_ERR_HIP", -2), ("H2GCN_ERR_OUT_OF_MEMORY", -3),
                      ("H2GCN_ERR_BAD_INDEX", -4), ("H2GCN_ERR_NO_TRANSPOSE", -5)):
        assert re.search(rf"{name}\s*=\s*{val}\b", text)
    assert ctypes.sizeof(_capi.PlanOpts) == 32
    for name, val in (("H2GCN_ERR_INTERNAL", -6), ("H2GCN_ERR_EXCHANGE_TIMEOUT", -7)):
        assert re.search(rf"{name}\s*=\s*{val}\b", text)
    assert ctypes.sizeof(_capi.LaunchOpts) == 32 and _capi.XCHG_BLOB_BYTES == int(re.search(r"H2GCN_XCHG_BLOB_BYTES\s+(\d+)", text).group(1))
    assert int(re.search(r"#define H2GCN_ABI_VERSION\s+(\d+)", text).group(1)) == _capi.ABI_VERSION


def test_null_plan_is_an_error_not_a_crash():
    L = _capi.lib()
    st = L.h2gcn_spmm_hops_f32(None, 0, None, 0, 1, None, 0, 0, None)
    assert st == _capi.ERR_INVALID_ARGUMENT
    assert b"plan is NULL" in L.h2gcn_last_error()
    with pytest.raises(_capi.H2GCNError):
        _capi.check(st)
    L.h2gcn_plan_destroy(None)  # no-op
    # the entry points added in ABI 2 validate their arguments before touching the device as well
    assert L.h2gcn_spmm_hops_opts_f32(None, 0, None, 0, 1, None, 0, 0, None, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_plan_schedule(None, 0, 0, 1, 1, None, None, None, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_plan_set_values(None, 0, None, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_spmm_workspace_bytes(None, 0, 0, None, 1, 0, 1) == 0
    assert L.h2gcn_ring_count(-1, None, None, None, None, 0, None, None, 0, 0, None, None, 0, None, None, None, 0, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_hop_normalize(4, None, None, 1, None, 0, None, None) == _capi.ERR_INVALID_ARGUMENT
    # ABI 3: the adjoint with options, row-window rings, the dropout+dense kernels, exchange capture support
    assert L.h2gcn_spmm_hops_T_opts_f32(None, 0, None, 1, 1, 1, None, 1, None, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_ring_count_rows(8, 4, 8, None, None, None, None, 0, None, None, 0, 0, None, None, 0, None, None, None, 0, None) == _capi.ERR_INVALID_ARGUMENT
    assert b"row window" in L.h2gcn_last_error()
    assert L.h2gcn_hop_normalize_rows(4, None, None, 1, None, 0, None, None, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_dropout_dense_workspace_bytes(1000, 448, 47) > 0 and L.h2gcn_dropout_dense_workspace_bytes(10, 8, 65) == 0
    assert L.h2gcn_dropout_dense_f32(None, 8, 10, 8, None, 65, None, 0.5, 1, None, None, 8, None, 0, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_dropout_dense_f32(None, 8, 10, 8, None, 4, None, 0.0, 1, None, None, 8, None, 0, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_dropout_dense_backward_f32(None, 8, 10, 8, None, 4, None, 4, 0.5, 1, None, None, 8, None, None, 0, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_xchg_reset_dependencies(None) == _capi.ERR_INVALID_ARGUMENT
    assert L.h2gcn_xchg_status(None) == _capi.ERR_INVALID_ARGUMENT and L.h2gcn_xchg_allgather_end(None, 0, None) == _capi.ERR_INVALID_ARGUMENT
    out = ctypes.c_void_p()
    assert L.h2gcn_xchg_create(0, 0, 1, 64, 0, 0, ctypes.byref(out)) == _capi.ERR_INVALID_ARGUMENT and not out.value
    L.h2gcn_xchg_destroy(None)  # no-op


def test_product_path_has_no_cpu_fallback():
    import torch

    from h2gcn_amd import HopPlan

    rp = torch.tensor([0, 1], dtype=torch.int64)
    ci = torch.tensor([0], dtype=torch.int32)
    va = torch.tensor([1.0])
    with pytest.raises(ValueError, match="GPU"):
        HopPlan([rp], [ci], [va], 1)


def test_product_never_imports_the_oracle():
    for p in (ROOT / "h2gcn_amd").rglob("*.py"):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{p} imports the oracle"
    for p in (ROOT / "h2gcn_amd" / "csrc").glob("*"):
        if p.suffix in (".hip", ".h", ".cpp"):
            assert "oracle" not in p.read_text().lower(), f"{p} mentions the oracle"


def test_header_is_plain_c_and_library_links_without_python(tmp_path):
    """tools/capi_demo.c compiles as C99 with gcc against include/h2gcn_hip.h and links libh2gcn_hip.so alone
    (it is executed by the GPU suite)."""
    import __graft_entry__ as ge

    exe = ge.build_capi_demo()
    assert exe.exists()
