"""Producer -> consumer hand-off of freshly written rows between two workgroups of one launch (device scope).
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/handoff_latency.hip -o tools/probes/handoff_latency.so
    python tools/probes/probe_handoff_latency.py
"""
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "handoff_latency.so"))
lib.handoff_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
dev = torch.device("cuda:0")
data = torch.zeros(1 << 22, dtype=torch.int32, device=dev)
flag = torch.zeros(128, dtype=torch.int32, device=dev)
t_raise = torch.zeros(2, dtype=torch.int64, device=dev)
out = torch.zeros(4, dtype=torch.int64, device=dev)
rounds = 200
for cons, where in ((1, "another XCD"), (8, "the same XCD")):
    for vec in (1, 4, 16, 64):
        lib.handoff_launch(data.data_ptr(), flag.data_ptr(), t_raise.data_ptr(), vec, cons, rounds, out.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        o = out.tolist()
        kb = vec * 256 * 16 / 1024
        print(f"consumer on {where}, {kb:5.0f} KB: producer store + acknowledge {o[2] * 10 / rounds:6.0f} ns | flag raise -> seen "
              f"{o[1] * 10 / rounds:6.0f} ns | consumer loads {o[0] * 10 / rounds:6.0f} ns | stale values read {o[3]}")
