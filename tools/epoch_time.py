"""Wall time per epoch of the reference's own workload (Cora, H2GCN-2, full batch) through the entry point."""
import contextlib, io, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from conftest import load_planetoid_golden
from test_entrypoints import _export_fixture
from h2gcn_amd import run_experiments

tmp = Path(tempfile.mkdtemp())
_export_fixture(load_planetoid_golden("cora"), tmp, "ind.cora")
extra = sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    args = run_experiments.main(["H2GCN", "planetoid", "--dataset", "ind.cora", "--dataset_path", str(tmp), "--epochs", "300"] + extra)
print("ms/epoch", 1e3 * args.objects["wall_seconds"] / args.current_epoch, "best", {k: round(v, 4) for k, v in args.objects["best_val_stats"].items() if isinstance(v, float)})
