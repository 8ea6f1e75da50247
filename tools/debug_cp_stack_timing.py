"""Per-phase timing of the code predictor's one-launch stack from inside the kernel (debug build):
    cd vllm-omni_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMI_CP_TIMING -DMI_CP_TIMING_WG=0 -shared cp_stack.hip \
        -o ../build/libcp_stack_timing.so
    python tools/debug_cp_stack_timing.py [rows]
Loads the debug library for mi_cp_stack_decode only; prints work / barrier microseconds of every phase of the first layers."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from vllm_omni_amd import _C, ops  # noqa: E402
import test_gpu_cp_stack as t  # noqa: E402

dbg = ctypes.CDLL(os.path.join(ROOT, "vllm-omni_amd", "build", f"libcp_stack_timing_{os.environ.get('WG', '0')}.so"))
for name in ("mi_cp_stack_decode", "mi_cp_stack_workspace_bytes"):
    f = getattr(dbg, name)
    f.restype, f.argtypes = getattr(_C.lib, name).restype, getattr(_C.lib, name).argtypes
ops.lib = type("L", (), {"__getattr__": lambda self, n: getattr(dbg, n) if n.startswith("mi_cp_stack") else getattr(_C.lib, n)})()

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
lw, fn = t._weights(dev, 5, torch.bfloat16, 3)
kv, views = t._caches(dev, 5, rows, torch.bfloat16, 5)
stack = ops.CpStack([dict(layer, k_cache=views[i][0], v_cache=views[i][1]) for i, layer in enumerate(lw)], fn, t._cos_sin(dev),
                    t.NQ, t.NKV, t.D, t.EPS, t.D ** -0.5, max_rows=rows)
x = torch.randn(rows, t.H, device=dev).to(torch.bfloat16)
ids = torch.arange(1, rows + 1, dtype=torch.int32, device=dev)
out = torch.empty_like(x)
for _ in range(5):
    stack.decode(out, x, ids, 20)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    stack.decode(out, x, ids, 20)
e1.record()
torch.cuda.synchronize()
print(f"rows {rows}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per pass (5 layers), status {stack.status()}")
buf = (ctypes.c_ulonglong * 128)()
assert dbg.mi_debug_cp_ticks(buf) == 0
ts = [buf[i] for i in range(128)]
names = ["N1 norm", "G1 qkv", "A attn", "G2 o", "N2 norm", "G3 gate_up", "G4 down"]
i = 0
for layer in range(5):
    row = []
    for ph in range(7):
        if i + 2 >= 128 or ts[i + 2] == 0:
            break
        row.append(f"{names[ph]} {(ts[i + 1] - ts[i]) / 100:.1f}+{(ts[i + 2] - ts[i + 1]) / 100:.1f}")
        i += 2
    print(f"layer {layer}: " + " | ".join(row))
print(f"total {(ts[i] - ts[0]) / 100:.1f} us over {i // 2} phases (work+barrier wait, workgroup {os.environ.get('WG', '0')})")
