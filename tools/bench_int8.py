"""W8A8 linear (quantise + tiled int8 kernel) against the 16-bit projection it replaces and against the library
int8 contraction, on the S-A layer shapes at batch 256 and the S-B shapes at batch 512 (BASELINE config #5)."""
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vllm-omni_amd"))
from vllm_omni_amd import ops  # noqa: E402


def timed(fn, iters=30):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [("S-A qkv", 256, 4608, 3584), ("S-A o", 256, 3584, 3584), ("S-A gate_up", 256, 37888, 3584),
              ("S-A down", 256, 3584, 18944), ("S-B qkv", 512, 1280, 1024), ("S-B o", 512, 1024, 1024),
              ("S-B gate_up(dense stand-in)", 512, 768, 1024), ("S-B down", 512, 1024, 384),
              # prefill chunks (more rows than one launch of the tiled kernel takes: run in 512-row blocks)
              ("S-B qkv prefill", 2048, 1280, 1024), ("S-A qkv prefill", 2048, 4608, 3584),
              ("S-A down prefill", 2048, 3584, 18944)]
    for name, m, n, k in shapes:
        copies = max(2, int(600e6 // (n * k * 2)))
        copies = min(copies, 16)
        ws16 = [(torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
        q = [ops.dynamic_int8_quant(w) for w in ws16]
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        xq, xs = ops.dynamic_int8_quant(x)
        i = [0]

        def nxt():
            i[0] = (i[0] + 1) % copies
            return i[0]

        arms = {
            "bf16 (product policy)": lambda: (ops.linear(x, ws16[nxt()]) if m <= 256 and ops.linear_supported(m, n, k, x.dtype)
                                              and k >= 8192 else F.linear(x, ws16[nxt()])),
            "int8 quant+tiled": lambda: ops.int8_linear(x, *q[nxt()]),
            "int8 tiled GEMM only": lambda: (ops.int8_linear_q(xq, xs, *q[nxt()], None, torch.bfloat16) if m <= 512
                                             else ops.int8_linear(ops.QuantizedActs(xq, xs, (m,), torch.bfloat16), *q[nxt()])),
            "int8 simple MFMA kernel GEMM only": lambda: ops.int8_gemm_dequant(xq, xs, *q[nxt()], None, torch.bfloat16),
            "int8 library (_int_mm+dequant) GEMM only": lambda: ops.int8_dequant(torch._int_mm(xq, q[nxt()][0].t()), xs,
                                                                                 q[i[0]][1], None, torch.bfloat16),
        }
        res = {a: [] for a in arms}
        for a, fn in arms.items():
            for _ in range(3):
                fn()
        for _ in range(5):
            for a, fn in arms.items():
                res[a].append(timed(fn))
        print(f"--- {name}: m={m} n={n} k={k}")
        for a in arms:
            print(f"   {a:42s} {statistics.median(res[a]):7.1f} us (min {min(res[a]):7.1f})", flush=True)


if __name__ == "__main__":
    main()
