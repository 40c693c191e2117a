"""Stress of the IPC exchange protocol (csrc/exchange.hip): WORLD processes on one GPU, hundreds of rounds over several
channels with random per-rank delays (host sleeps and GPU busy-work) so that ranks drift apart by more than one
round; every gathered block is checked.  Usage: python tools/ipc_stress.py <world> <engine|kernel> <rounds>"""
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]

WORKER = r'''
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["H2GCN_ROOT"])
from h2gcn_amd.partition import IpcExchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
mode, rounds = os.environ["MODE"], int(os.environ["ROUNDS"])
per, widths = 4096, [32, 32, 64, 16]
xc = IpcExchange(len(widths), per * max(widths) * 4, dev, mode=mode, timeout_ms=30000)
rng = np.random.default_rng(1000 + rank)
fulls = [torch.empty((world * per, w), device=dev) for w in widths]
src = torch.empty((per, sum(widths)), device=dev)
busy = torch.rand((2048, 2048), device=dev)
bad_t = torch.zeros((), dtype=torch.int64, device=dev)   # checked on the device: the host never waits, ranks may run ahead
for step in range(rounds):
    if rng.random() < 0.3:
        time.sleep(rng.random() * 0.004)                  # host drift
    if rng.random() < 0.3:
        for _ in range(int(rng.integers(1, 6))): busy = (busy @ busy).clamp_(-1, 1)   # device drift
    src.fill_(float(rank * 100000 + step))
    off = 0
    for c, w in enumerate(widths):
        xc.begin(c, src[:, off:off + w], fulls[c], per); off += w
    for c, w in enumerate(widths):
        xc.end(c)
        got = fulls[c].view(world, per * w)
        want = torch.arange(world, device=dev, dtype=torch.float32) * 100000 + step
        bad_t += (got != want[:, None]).any()
torch.cuda.synchronize()
bad = int(bad_t.item())
xc.check()
xc.close()
dist.barrier(); dist.destroy_process_group()
print("STRESS_OK" if bad == 0 else f"STRESS_BAD {bad}")
'''


def main():
    world, mode, rounds = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MODE=mode, ROUNDS=str(rounds), H2GCN_ROOT=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    ok = all(p.returncode == 0 for p in procs) and all("STRESS_OK" in o for o in outs)
    print(f"world={world} mode={mode} rounds={rounds}:", "OK" if ok else "FAILED")
    if not ok:
        print("\n".join(o[-1500:] for o in outs))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
