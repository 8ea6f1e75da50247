"""Does a stall appear every N kernel launches?  Batches of 100 launches, wall time per batch (sync per batch).

    python tools/scratch/launch_stall_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))
import torch  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(2048, 1024, device=dev, dtype=torch.bfloat16)
w = torch.ones(1024, device=dev, dtype=torch.bfloat16)
wl = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16)


def run(label, fn, batches=120, sync=True):
    out = []
    torch.cuda.synchronize()
    for b in range(batches):
        t = time.perf_counter()
        for _ in range(100):
            fn()
        if sync:
            torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t) * 1e3, 1))
    torch.cuda.synchronize()
    big = [(i, v) for i, v in enumerate(out) if v > 5 * sorted(out)[len(out) // 2]]
    print(f"{label}: median {sorted(out)[len(out) // 2]} ms per 100 launches; outliers (batch, ms): {big}", flush=True)


run("torch add_", lambda: x.add_(1.0))
run("torch F.linear 2048x1024x1024", lambda: torch.nn.functional.linear(x, wl))
from vllm_omni_amd import ops  # noqa: E402
run("ops.rms_norm", lambda: ops.rms_norm(x, w, 1e-6))
run("ops.rms_norm, no per-batch sync", lambda: ops.rms_norm(x, w, 1e-6), sync=False)
run("torch add_, no per-batch sync", lambda: x.add_(1.0), sync=False)
