"""One-rank RCCL sanity check (run on the GPU box): process group init, all_gather_into_tensor on a
side stream, barrier, all_reduce(MAX) — the collectives bench.py uses at N > 1."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
x = torch.arange(8, dtype=torch.float32, device="cuda").reshape(4, 2)
out = torch.empty((4, 2), dtype=torch.float32, device="cuda")
dist.all_gather_into_tensor(out, x)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(out, x) and float(t) == 1.5
print("rccl ok", torch.cuda.nccl.version())
dist.destroy_process_group()
