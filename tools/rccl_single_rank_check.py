import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29555", RANK="0", WORLD_SIZE="1")
from h2gcn_amd.partition import init_rccl_process_group, _all_gather_rows, _all_gather_rows_p2p, _reduce_scatter_rows
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
init_rccl_process_group(dev, 120.0)  # with the collective timeout bench.py passes
send = torch.arange(12, dtype=torch.float32, device=dev).view(3, 4); full = torch.zeros_like(send)
_all_gather_rows(full, send); assert torch.equal(full, send)
full.zero_(); _all_gather_rows_p2p(full, send); assert torch.equal(full, send)
out = _reduce_scatter_rows(send.clone(), 3, 0); assert torch.equal(out, send)
s = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(s):
    dist.all_gather_into_tensor(full, send)
objs = [None]; dist.all_gather_object(objs, b"blob")            # the bootstrap channel of the IPC exchange, over RCCL
assert objs == [b"blob"]
from h2gcn_amd.partition import IpcExchange
x = IpcExchange(2, 4096, dev); x.close()                          # world 1: create / export / destroy under the nccl backend
torch.cuda.synchronize(); dist.barrier(); print("rccl single-rank ok", dist.get_backend())
dist.destroy_process_group()
