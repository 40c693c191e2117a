cd /root/repo
python - <<'PY'
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import load_planetoid_golden
from test_entrypoints import _export_fixture
from pathlib import Path
_export_fixture(load_planetoid_golden("cora"), Path("/tmp/cora_data"), "ind.cora")
PY
mkdir -p gpurun_out/r03; python tools/sharded_epoch_time.py /tmp/cora_data 2>&1 | tee gpurun_out/r03/sharded_epoch_time.txt
