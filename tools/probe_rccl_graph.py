"""Can torch.distributed (RCCL) collectives be captured in a hipGraph and replayed on this stack?
Single-rank communicator on one GPU: exercises the ProcessGroupNCCL capture path (stream fork/join,
watchdog) — the multi-rank kernels differ only inside RCCL."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
grp = dist.new_group(ranks=[0], backend="nccl")
x = torch.randn(256, 3584, dtype=torch.bfloat16, device=dev)
w = torch.randn(3584, 3584, dtype=torch.bfloat16, device=dev) * 0.02
y = torch.empty_like(x)
# warm-up (communicator creation must happen before capture)
for _ in range(3):
    t = torch.nn.functional.linear(x, w)
    dist.all_reduce(t, group=grp)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t = torch.nn.functional.linear(x, w)
    dist.all_reduce(t, group=grp)
    parts = [torch.empty_like(t)]
    dist.all_gather(parts, t, group=grp)
    y.copy_(parts[0])
for i in range(3):
    x.normal_()
    g.replay()
    torch.cuda.synchronize()
    ref = torch.nn.functional.linear(x, w)
    print("replay", i, "max err", (y.float() - ref.float()).abs().max().item())
dist.destroy_process_group()
print("RCCL_GRAPH_CAPTURE_OK")
