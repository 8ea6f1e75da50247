import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, "vllm-omni_amd")
from vllm_omni_amd import ops
from vllm_omni_amd.tuned import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for m in (32, 64, 256, 512):
    x = torch.randn(m, 1024, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(6144, 1024, device=dev) * 0.03).to(torch.bfloat16)
    a = t(lambda: ops.silu_and_mul(F.linear(x, w)))
    b = t(lambda: ops.linear_silu_mul(x, w)) if ops.linear_supported(m, 3072, 1024, x.dtype) else float("nan")
    wd = (torch.randn(1024, 3072, device=dev) * 0.03).to(torch.bfloat16)
    xd = torch.randn(m, 3072, device=dev, dtype=torch.bfloat16)
    c = t(lambda: F.linear(xd, wd)); d = t(lambda: ops.linear(xd, wd))
    wq = (torch.randn(4096, 1024, device=dev) * 0.03).to(torch.bfloat16)
    e = t(lambda: F.linear(x, wq)); f = t(lambda: ops.linear(x, wq))
    wo = (torch.randn(1024, 2048, device=dev) * 0.03).to(torch.bfloat16)
    xo = torch.randn(m, 2048, device=dev, dtype=torch.bfloat16)
    g = t(lambda: F.linear(xo, wo)); h = t(lambda: ops.linear(xo, wo))
    print(f"M={m}: o lib {g:.1f}, kernel {h:.1f}")
    print(f"M={m}: gate_up+silu lib {a:.1f} us, fused kernel {b:.1f} us | down lib {c:.1f}, kernel {d:.1f} | qkv lib {e:.1f}, kernel {f:.1f}")
