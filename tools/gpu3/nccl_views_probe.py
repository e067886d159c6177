"""RCCL probe on ONE GPU (world size 1): the calls the fused schedule makes on a real communicator — the list form of
all-to-all with views of bigger buffers as inputs and outputs, with an EMPTY entry, asynchronously, plus an all-reduce and
the single-tensor all-to-all with split sizes. A one-rank communicator exercises RCCL's group launch and PyTorch's argument
checks, not the links."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
blk = torch.randn(4, 1000, 32, device=dev)
cols = torch.zeros(5000, 32, device=dev)
w = dist.all_to_all([cols[100:1100]], [blk[2]], async_op=True)  # views in, views out
w.wait()
assert torch.equal(cols[100:1100], blk[2])
w = dist.all_to_all([cols[0:0]], [blk[1, 0:0]], async_op=True)  # an empty exchange
w.wait()
recv = torch.empty(1000, 32, device=dev)
dist.all_to_all_single(recv, blk[3].contiguous(), [1000], [1000])
assert torch.equal(recv, blk[3])
t = torch.ones(8, dtype=torch.float64, device=dev)
dist.all_reduce(t)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):  # issued from a second stream, as the interleaved eval forwards do
    w = dist.all_to_all([cols[2000:3000]], [blk[0]], async_op=True)
    w.wait()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
assert torch.equal(cols[2000:3000], blk[0])
print("rccl one-rank probe ok:", torch.cuda.get_device_name(0), "backend", dist.get_backend())
dist.destroy_process_group()
