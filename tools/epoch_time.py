"""Wall time per epoch of the reference's own workload (Cora, H2GCN-2, full batch) through the entry point."""
import contextlib, io, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from conftest import load_planetoid_golden
from test_entrypoints import _export_fixture
from h2gcn_amd import run_experiments

tmp = Path(tempfile.mkdtemp())
extra = sys.argv[1:]
name = "cora"
if "--graph" in extra:          # --graph citeseer: the other planetoid fixture (3 327 nodes; 1-hop mean 2.7, 2-hop 11.4: short throughout)
    i = extra.index("--graph")
    name = extra[i + 1]
    del extra[i:i + 2]
_export_fixture(load_planetoid_golden(name), tmp, f"ind.{name}")
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    args = run_experiments.main(["H2GCN", "planetoid", "--dataset", f"ind.{name}", "--dataset_path", str(tmp), "--epochs", "300"] + extra)
print("ms/epoch", 1e3 * args.objects["wall_seconds"] / args.current_epoch, "best", {k: round(v, 4) for k, v in args.objects["best_val_stats"].items() if isinstance(v, float)})
