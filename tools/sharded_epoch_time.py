"""Epoch time of ROW-PARTITIONED Cora training (H2GCN-2) with 2 ranks sharing one GPU, eager vs hipGraph replay
(H2GCN_EXCHANGE=ipc_kernel): differential measurement, (t(E2 epochs) - t(E1 epochs)) / (E2 - E1), so start-up cancels.
usage: python tools/sharded_epoch_time.py <dir with ind.cora.* planetoid files>"""
import json
import os
import socket
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.environ["H2GCN_ROOT"])
import torch
from h2gcn_amd import run_experiments
args = run_experiments.main(["H2GCN", "planetoid", "--dataset", "ind.cora", "--dataset_path", os.environ["DATA_DIR"], "--epochs", os.environ["EPOCHS"],
                      "--random_seed", "11", "--network_setup", "M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO"] + os.environ.get("EXTRA", "").split())
torch.cuda.synchronize()
if int(os.environ.get("RANK", "0")) == 0:
    json.dump({"seconds": args.objects["wall_seconds"]}, open(os.environ["OUT_FILE"], "w"))   # the epoch loop only
'''


def run(world, epochs, extra, data_dir, exchange="ipc_kernel"):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "o.json"
        procs = []
        for rank in range(world):
            env = dict(os.environ, H2GCN_ROOT=str(ROOT), DATA_DIR=str(data_dir), OUT_FILE=str(out), EPOCHS=str(epochs), EXTRA=extra,
                       H2GCN_EXCHANGE=exchange)
            if world > 1:
                env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                           H2GCN_DIST_BACKEND="gloo", H2GCN_SHARE_GPU="1")
            procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
        errs = [p.communicate(timeout=1800)[1].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(errs)
        return json.loads(out.read_text())["seconds"]


if __name__ == "__main__":
    data_dir = sys.argv[1]
    e1, e2 = 100, 1100
    run(1, 5, "", data_dir)   # page the image in
    for label, world, extra in (("1 process, replayed", 1, ""), ("1 process, eager", 1, "--no_hipgraph"),
                                ("2 ranks on one GPU, ipc_kernel exchange, eager", 2, "--no_hipgraph"),
                                ("2 ranks on one GPU, ipc_kernel exchange, hipGraph replay", 2, "")):
        t1, t2 = run(world, e1, extra, data_dir), run(world, e2, extra, data_dir)
        print(f"{label}: {(t2 - t1) / (e2 - e1) * 1e3:.3f} ms / epoch   (t({e2}) = {t2:.2f} s, t({e1}) = {t1:.2f} s)")
