"""CPU suite: bench.py's byte models and its refusal to run without a GPU (no CPU fallback anywhere)."""
import importlib.util
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_and_compulsory_bytes():
    b = _bench()
    nnz, n, d = [119979897, 119978572], 2_400_000, 128
    # 520 B per aggregated edge at d = 128, + row pointers per hop, + one write of Y
    want = sum(z * 520 + (n + 1) * 8 for z in nnz) + n * 2 * d * 4
    assert b.algorithmic_bytes(nnz, n, d, 2) == want == 127274403896
    assert b.compulsory_bytes(nnz, n, n, d, 2) == sum(z * 8 + (n + 1) * 8 for z in nnz) + n * d * 4 + n * 2 * d * 4
    assert b.HBM_PEAK_GBPS == 8000.0
    assert b.pmc_traffic("products", 128, 1, 0, 1)["bytes_per_launch"] > 1e11   # committed PMC summary (+ its source)
    assert b.pmc_traffic("products", 128, 1, 0, 8) is None           # never profiled -> null, not a guess


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)
