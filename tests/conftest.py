import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "time_cap(seconds): wall-clock cap of this test (default 420 s; tests/conftest.py)")
    # property-based sweeps draw the SAME examples on every run (a suite that is green today is green tomorrow); explore with
    # fresh draws by setting H2GCN_FUZZ_RANDOM=1 (and H2GCN_FUZZ_EXAMPLES=<n>, --hypothesis-seed=<s>): round 4 ran 4 500 / 24 000
    # random examples of the SpMM / metrics sweeps that way
    import os
    try:
        from hypothesis import settings
        settings.register_profile("fixed", derandomize=True)
        if os.environ.get("H2GCN_FUZZ_RANDOM") != "1":
            settings.load_profile("fixed")
    except ImportError:
        pass


# Order of the -m gpu suite (the driver runs it with -x): the parity tests proper come FIRST -- the aggregation kernel against the
# oracle, the property sweeps, the ring kernels, the model glue -- and the multi-process tests (several ranks sharing the box's one
# GPU, rendezvous ports, RCCL's watchdog) LAST, so that nothing that can go wrong in a launcher costs a parity test its run.
# Inside test_multirank_gpu.py the row-partition bit-equality tests run before the bench / supervisor failure-injection tests.
_FILE_RANK = ["test_spmm_gpu", "test_fullsize_parity_gpu", "test_fuzz_gpu", "test_rings_gpu", "test_model_gpu", "test_entrypoints",
              "test_optim_gpu", "test_metrics_gpu", "test_classifier_gpu", "test_multirank_gpu"]
_DEFAULT_CAP_S = 420


def _order_key(item):
    stem = Path(str(item.fspath)).stem
    rank = _FILE_RANK.index(stem) if stem in _FILE_RANK else len(_FILE_RANK) - 1.5    # unknown files: just before the multi-process file
    late = 1 if (stem == "test_multirank_gpu" and (item.name.startswith("test_bench_") or item.name.startswith("test_rccl_"))) else 0
    return (rank, late)


def pytest_collection_modifyitems(config, items):
    """Parity first, launchers last (see _FILE_RANK); the sort is stable, so the order inside a file is the file's own."""
    items.sort(key=_order_key)


def _cap_seconds(item):
    m = item.get_closest_marker("time_cap")
    return int(m.args[0]) if m and m.args else _DEFAULT_CAP_S


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """No single test may eat the suite's time limit: a cap per test (SIGALRM, no plug-in needed; `@pytest.mark.time_cap(s)` sets
    another one) turns a hang into ONE failing test instead of a suite that never reports.  Child processes a test started are
    reaped by the test's own `finally` blocks (the exception below unwinds through them)."""
    import signal

    cap = _cap_seconds(item)
    if cap <= 0 or not hasattr(signal, "SIGALRM"):
        yield
        return

    def on_alarm(signum, frame):
        raise TimeoutError(f"{item.nodeid}: exceeded its cap of {cap} s (tests/conftest.py)")

    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(cap)
    try:
        yield
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """CPU-side artefacts the suites need: the oracle's C library and (if absent) the HIP library.
    Building the checker is not using it; on the GPU box both are already in the snapshot."""
    import subprocess

    if not (ROOT / "oracle" / "_build" / "liboracle.so").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    if not (ROOT / "h2gcn_amd" / "csrc" / "libh2gcn_hip.so").exists():
        subprocess.run(["make", "-C", str(ROOT / "h2gcn_amd" / "csrc")], check=True, capture_output=True)
    yield


def load_planetoid_golden(name):
    """dict of scipy CSR matrices / arrays from tests/golden/<name>_operands.npz."""
    import numpy as np
    import scipy.sparse as sp

    z = np.load(GOLDEN / f"{name}_operands.npz")
    n = len(z["adj_raw_indptr"]) - 1

    def csr(prefix, data_key=None, idx_prefix=None, shape=None):
        ip = z[f"{idx_prefix or prefix}_indptr"]
        ix = z[f"{idx_prefix or prefix}_indices"]
        da = z[data_key or f"{prefix}_data"]
        return sp.csr_matrix((da, ix, ip), shape=shape or (n, n))

    nfeat = int(z["feat_raw_indices"].max()) + 1
    out = dict(
        n=n,
        adj_raw=csr("adj_raw"),
        adj_noeye=csr("adj_noeye"),
        feat_raw=csr("feat_raw", shape=(n, nfeat)).astype(str(z["feat_raw_dtype"])),
        feat_rownorm=csr("feat_rownorm", data_key="feat_rownorm_data", idx_prefix="feat_raw", shape=(n, nfeat)),
        hop1_sym=csr("hop1_sym"), hop2_sym=csr("hop2_sym"),
        hop1_rw=csr("hop1_rw", data_key="hop1_rw_data", idx_prefix="hop1_sym"),
        hop2_rw=csr("hop2_rw", data_key="hop2_rw_data", idx_prefix="hop2_sym"),
        hop01_sym=csr("hop01_sym"),
        split_nnz=z["split_nnz"], y_all=z["y_all"], train_mask=z["train_mask"], val_mask=z["val_mask"],
        test_mask=z["test_mask"], num_labels=int(z["num_labels"]),
    )
    return out


def load_syn_products_golden():
    import numpy as np
    import scipy.sparse as sp

    z = np.load(GOLDEN / "syn_products.npz")
    n = len(z["indptr"]) - 1
    a = sp.csr_matrix((np.ones(len(z["indices"]), dtype=np.float32), z["indices"], z["indptr"]), shape=(n, n))
    return a, z["labels"], float(z["homophily"])


def golden_weight(index, shape, kind="kernel"):
    """Weight number `index` (creation order) of the glue fixtures (tests/golden/glue_cora.*): regenerable, so the
    fixture stores shapes only.  Glorot-uniform kernels, small uniform biases, one PCG64 stream per weight."""
    import numpy as np

    rng = np.random.Generator(np.random.PCG64(977 + int(index)))
    if kind == "bias":
        return rng.uniform(-0.1, 0.1, shape).astype(np.float32)
    lim = np.sqrt(6.0 / float(sum(shape)))
    return rng.uniform(-lim, lim, shape).astype(np.float32)
