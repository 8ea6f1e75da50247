"""Is a periodic ~80 ms stall a property of the box?  Pure torch: 8 ms of library GEMMs per step, wall time per step.

    python tools/scratch/box_stall_probe.py"""
import time

import torch

dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
for mode in ("gemm_only", "gemm+alloc"):
    out = []
    for step in range(60):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(40):
            c = a @ b
            if mode == "gemm+alloc":
                d = torch.empty(24576, 1024, device=dev, dtype=torch.bfloat16)
                d.zero_()
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t) * 1e3, 1))
    print(mode, out, flush=True)
