"""Cost of a grid-wide barrier inside one persistent launch, and of streaming a phase's weights between two of them.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/grid_barrier.hip -o tools/probes/grid_barrier.so
    python tools/probes/probe_grid_barrier.py
"""
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "grid_barrier.so"))
lib.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
ctr = torch.zeros(1024, dtype=torch.int32, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
sink = torch.zeros(2, dtype=torch.int32, device=dev)
PH = 35   # one layer-step = 7 phases; 5 layers
w = torch.randint(0, 255, (PH * (8 << 20),), dtype=torch.uint8, device=dev)   # up to 8 MB per phase: 280 MB... MALL-warm when small


def run(bytes_per_phase, nwg, mode, iters=20):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.probe_launch(w.data_ptr(), bytes_per_phase, PH, nwg, mode, ctr.data_ptr(), err.data_ptr(), sink.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.probe_launch(w.data_ptr(), bytes_per_phase, PH, nwg, mode, ctr.data_ptr(), err.data_ptr(), sink.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3 / PH   # us per phase


for nwg in (256, 512):
    for mb in (0, 1, 2, 4, 8):
        b = mb << 20
        t = [run(b, nwg, m) for m in (0, 1, 2, 3)]
        print(f"nwg {nwg} bytes/phase {mb} MB: acq/rel {t[0]:6.2f} us/phase  relaxed {t[1]:6.2f}  no barrier {t[2]:6.2f}  two-level {t[3]:6.2f}"
              f"   (stream rate {b / 1e6 / max(t[2], 1e-9):6.2f} TB/s)  err {int(err.item())} ctr {ctr[0].item()} {ctr[32].item()} {ctr[288].item()}", flush=True)
print("spin-limit trips:", int(err.item()), " workgroups off XCD blockIdx % 8 (all launches):", int(sink[1].item()))
