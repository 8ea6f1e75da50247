"""Host-side cost of the per-launch plumbing in ops.py (no GPU work waited for).

    python tools/scratch/host_overhead_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))
import torch  # noqa: E402

from vllm_omni_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
x = torch.randn(64, 1024, device=dev, dtype=torch.bfloat16)
w = torch.ones(1024, device=dev, dtype=torch.bfloat16)


def per_call(fn, n=3000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    d = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize()
    return d


print(f"torch.cuda.current_stream().cuda_stream : {per_call(lambda: torch.cuda.current_stream().cuda_stream):.2f} us")
print(f"ops._stream()                           : {per_call(ops._stream):.2f} us")
print(f"x.data_ptr()                            : {per_call(x.data_ptr):.2f} us")
print(f"torch.empty_like(x)                     : {per_call(lambda: torch.empty_like(x)):.2f} us")
print(f"ops.rms_norm(x, w, eps)                 : {per_call(lambda: ops.rms_norm(x, w, 1e-6)):.2f} us")
print(f"x.add_(1)                               : {per_call(lambda: x.add_(1)):.2f} us")
