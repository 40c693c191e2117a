"""Can RCCL run several ranks on ONE GPU?  Its "Duplicate GPU detected" check compares (host hash, PCI bus id); with a different
NCCL_HOSTID per rank the ranks look like different hosts and RCCL connects them through its NET/Socket transport.  Probe:
`python tools/rccl_shared_gpu_probe.py <world>` spawns <world> ranks on device 0 and runs the collectives the row-partitioned
path uses."""
import os, subprocess, socket, sys, time

WORKER = r'''
import os, sys, datetime, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
opts = dist.ProcessGroupNCCL.Options(); opts.is_high_priority_stream = True
dist.init_process_group("nccl", device_id=dev, pg_options=opts, timeout=datetime.timedelta(seconds=120))
per, w = 1000, 64
send = torch.full((per, w), float(rank + 1), device=dev); full = torch.zeros((world * per, w), device=dev)
dist.all_gather_into_tensor(full, send)
torch.cuda.synchronize()
ok = all(bool((full[q * per:(q + 1) * per] == q + 1).all()) for q in range(world))
t = torch.tensor([rank + 1.0], device=dev); dist.all_reduce(t)
ops = []
recv = torch.zeros((world * per, w), device=dev)
for shift in range(1, world):
    ops.append(dist.P2POp(dist.isend, send, (rank + shift) % world)); src = (rank - shift) % world
    ops.append(dist.P2POp(dist.irecv, recv[src * per:(src + 1) * per], src))
for r_ in dist.batch_isend_irecv(ops): r_.wait()
torch.cuda.synchronize()
ok2 = all(bool((recv[q * per:(q + 1) * per] == q + 1).all()) for q in range(world) if q != rank)
out = torch.empty((per, w), device=dev); dist.reduce_scatter_tensor(out, full.clone())
torch.cuda.synchronize()
ok3 = bool((out == world * (rank + 1)).all())
dist.barrier()
print(f"rank {rank}: all_gather {ok} all_reduce {t.item() == world * (world + 1) / 2} p2p {ok2} reduce_scatter {ok3}", flush=True)
dist.destroy_process_group()
'''

def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench_supervisor import _free_port
    port = _free_port()          # below the kernel's ephemeral range (RCCL's own sockets take ephemeral ports)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NCCL_HOSTID=f"h2gcn-fake-host-{r}", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update({k[len("PROBE_"):]: v for k, v in os.environ.items() if k.startswith("PROBE_")})
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    t0 = time.time()
    for r, p in enumerate(procs):
        try:
            o = p.communicate(timeout=300)[0]
        except subprocess.TimeoutExpired:
            p.kill(); o = "TIMEOUT " + p.communicate()[0]
        print(f"--- rank {r} rc {p.returncode}\n" + "\n".join(o.splitlines()[-12:]))
    print("elapsed", time.time() - t0)

if __name__ == "__main__":
    main()
