"""tools/epoch_time.py for an arbitrary source tree (A/B of the Cora epoch): python tools/epoch_time_tree.py <tree> [args]"""
import contextlib, io, sys, tempfile
from pathlib import Path
tree = Path(sys.argv[1]).resolve()
sys.path.insert(0, str(tree)); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from conftest import load_planetoid_golden
from test_entrypoints import _export_fixture
from h2gcn_amd import run_experiments
tmp = Path(tempfile.mkdtemp()); _export_fixture(load_planetoid_golden("cora"), tmp, "ind.cora")
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    args = run_experiments.main(["H2GCN", "planetoid", "--dataset", "ind.cora", "--dataset_path", str(tmp), "--epochs", "600"] + sys.argv[2:])
print(tree.name, sys.argv[2:], "ms/epoch", 1e3 * args.objects["wall_seconds"] / args.current_epoch)
