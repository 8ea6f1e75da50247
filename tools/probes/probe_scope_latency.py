"""Round-trip latency of dependent loads / acknowledged stores of ONE wave at the scopes the one-launch stack uses.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/scope_latency.hip -o tools/probes/scope_latency.so
    python tools/probes/probe_scope_latency.py
"""
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "scope_latency.so"))
lib.lat_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(2, dtype=torch.int64, device=dev)


def chase(n_words, stride_words=32):
    """a random cycle over the 128-byte lines of the buffer: word i holds the index of the next line's first word"""
    lines = n_words // stride_words
    perm = torch.randperm(lines, device=dev)
    nxt = torch.empty(lines, dtype=torch.int64, device=dev)
    nxt[perm] = perm.roll(-1)
    buf = torch.zeros(n_words, dtype=torch.int32, device=dev)
    buf[::stride_words] = (nxt * stride_words).to(torch.int32)
    buf.nxt = nxt
    return buf


def expected_index(buf, steps, stride_words=32):
    nxt = buf.nxt.cpu().numpy()
    i = 0
    for _ in range(steps):
        i = int(nxt[i])
    return i * stride_words


def run(buf, mode, steps=20000):
    s = torch.cuda.current_stream().cuda_stream
    lib.lat_launch(buf.data_ptr(), steps, mode, out.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.lat_launch(buf.data_ptr(), steps, mode, out.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    # ns per operation: by the kernel's own clock (s_memrealtime, 100 MHz) and by HIP events around the launch
    return out[0].item() * 10.0 / steps, e0.elapsed_time(e1) * 1e6 / steps


small = chase(64 * 1024 // 4 * 4)            # 256 KB: lives in the L2
big = chase(256 * 1024 * 1024 // 4)          # 256 MB
for label, buf, mode in (("dependent loads, 256 KB buffer, plain       ", small, 0),
                         ("dependent loads, 256 KB buffer, sc1 (device)", small, 1),
                         ("dependent loads, 256 MB buffer, plain       ", big, 2),
                         ("store + vmcnt(0), plain                     ", small, 3),
                         ("store + vmcnt(0), sc1 (write-through)       ", small, 4)):
    a, b = run(buf, mode)
    if mode <= 2:
        assert out[1].item() == expected_index(buf, 20000), (out[1].item(), expected_index(buf, 20000))
    print(f"{label}: {a:7.0f} ns by the kernel's clock, {b:7.0f} ns by HIP events around the launch")
