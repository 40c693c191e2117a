"""RCCL (backend "nccl") at world size 1 on this box's GPU: the process-group options, collectives and IPC set-up the
row-partitioned path uses on a multi-GPU node (run by tests/test_multirank_gpu.py::test_rccl_backend_at_world_size_one)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29555"), RANK="0", WORLD_SIZE="1")
import json, tempfile
from h2gcn_amd.partition import init_rccl_process_group, _all_gather_rows, _all_gather_rows_p2p, _reduce_scatter_rows
from h2gcn_amd.partition import enable_rccl_debug_log, summarize_rccl_log
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
log_dir = tempfile.mkdtemp(prefix="h2gcn_rccl_", dir="/tmp"); enable_rccl_debug_log(log_dir)   # as bench.py does at N > 1
init_rccl_process_group(dev, 120.0)  # with the collective timeout bench.py passes
send = torch.arange(12, dtype=torch.float32, device=dev).view(3, 4); full = torch.zeros_like(send)
_all_gather_rows(full, send); assert torch.equal(full, send)
full.zero_(); _all_gather_rows_p2p(full, send); assert torch.equal(full, send)
out = _reduce_scatter_rows(send.clone(), 3, 0); assert torch.equal(out, send)
s = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(s):
    dist.all_gather_into_tensor(full, send)
send_t, recv_t = torch.ones(8, device=dev), torch.zeros(8, device=dev)     # the grouped send/recv form (to self at world 1)
for w_ in dist.batch_isend_irecv([dist.P2POp(dist.isend, send_t, 0), dist.P2POp(dist.irecv, recv_t, 0)]):
    w_.wait()
assert torch.equal(send_t, recv_t)
rs_in = torch.arange(6, dtype=torch.float32, device=dev); rs_out = torch.zeros(6, device=dev)
dist.reduce_scatter_tensor(rs_out, rs_in); assert torch.equal(rs_out, rs_in)
objs = [None]; dist.all_gather_object(objs, b"blob")            # the bootstrap channel of the IPC exchange, over RCCL
assert objs == [b"blob"]
from h2gcn_amd.partition import IpcExchange
x = IpcExchange(2, 4096, dev); x.close()                          # world 1: create / export / destroy under the nccl backend
torch.cuda.synchronize(); dist.barrier()
summary = summarize_rccl_log(log_dir)
print(json.dumps({"rccl": summary}))          # what bench.py puts under config.diagnostics.rccl (file log only)
if not summary:   # say why: which files exist, what RCCL was told
    import glob
    print("no RCCL log found:", {k: v for k, v in os.environ.items() if k.startswith("NCCL_") or k.startswith("RCCL_")},
          "dir:", os.listdir(log_dir), "tmp:", glob.glob("/tmp/rccl*") + glob.glob("/tmp/*/rccl*"), file=sys.stderr)
print("rccl single-rank ok", dist.get_backend())
dist.destroy_process_group()
