"""Cora epoch time of build/ab/old (an older source tree exported with `git archive`) vs the working tree through the
CLI -- separate processes, clean sys.path.  Per-epoch time = (t(LONG epochs) - t(SHORT epochs)) / (LONG - SHORT), so interpreter
start-up, operand construction and hipGraph capture cancel.  Usage: python tools/cora_epoch_ab.py"""
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_planetoid_golden  # noqa: E402
from test_entrypoints import _export_fixture  # noqa: E402

data = Path(tempfile.mkdtemp())
_export_fixture(load_planetoid_golden("cora"), data, "ind.cora")


def wall(tree, epochs, extra):
    t = time.perf_counter()
    subprocess.run([sys.executable, "-m", "h2gcn_amd.run_experiments", "H2GCN", "planetoid", "--dataset", "ind.cora",
                    "--dataset_path", str(data), "--epochs", str(epochs)] + extra, cwd=tree, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.perf_counter() - t


SHORT, LONG = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 3000)
for rnd in (1, 2):
    for tree in (ROOT / "build" / "ab" / "old", ROOT):
        if not (tree / "h2gcn_amd").exists():
            continue
        for extra in ([], ["--no_hipgraph"]):
            a, b = wall(tree, SHORT, extra), wall(tree, LONG, extra)
            print(f"round {rnd} tree={tree.name:5s} {' '.join(extra) or 'hipGraph replay':16s}: {(b - a) / (LONG - SHORT) * 1e3:.3f} ms/epoch", flush=True)
