"""Decode-batch linear kernel against the library GEMM (torch F.linear, TunableOp-selected hipBLASLt/rocBLAS) on the
headline talker's four projections.  Prints us per call and the weight-streaming rate."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vllm-omni_amd"))
from vllm_omni_amd import ops  # noqa: E402


def timed(fn, iters=50, warm=10):
    """us per call with the calls REPLAYED FROM A hipGraph (as the product runs them): an eager loop of 10-25 us
    kernels measures the host's launch rate, not the kernel (the library path costs ~15 us of CPU per call)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    n = 16
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    reps = max(2, iters // n)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        g.replay()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / (reps * n) * 1e3


def main():
    from vllm_omni_amd.tuned import enable_tuned_gemms

    print("tuned library selections loaded:", enable_tuned_gemms())
    m = int(os.environ.get("M", 256))
    shapes = [("qkv", 4608, 3584), ("o", 3584, 3584), ("gate_up", 37888, 3584), ("down", 3584, 18944)]
    tp = int(os.environ.get("TP", 1))    # per-rank shard shapes of a tensor-parallel plan (qwen2_old.py:89-102)
    if tp > 1:
        nq, nkv = 28 // tp, max(1, 4 // tp)
        shapes = [("qkv", (nq + 2 * nkv) * 128, 3584), ("o", 3584, nq * 128), ("gate_up", 2 * 18944 // tp, 3584),
                  ("down", 3584, 18944 // tp)]
    g = torch.Generator(device="cuda").manual_seed(0)
    # rotate over several weight copies so that no weight is served from the 256 MB Infinity Cache
    for name, n, k in shapes:
        copies = max(2, int(600e6 // (n * k * 2)))
        ws = [(torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        i = [0]

        def lib():
            i[0] = (i[0] + 1) % copies
            return F.linear(x, ws[i[0]])

        def mine():
            i[0] = (i[0] + 1) % copies
            return ops.linear(x, ws[i[0]])

        err = (ops.linear(x, ws[0]).float() - F.linear(x, ws[0]).float()).abs().max().item()
        t_lib, t_mine = timed(lib), timed(mine)
        gb = n * k * 2 / 1e9
        extra = ""
        if name == "gate_up":
            def fused():
                i[0] = (i[0] + 1) % copies
                return ops.linear_silu_mul(x, ws[i[0]])

            def lib_pair():
                i[0] = (i[0] + 1) % copies
                return ops.silu_and_mul(F.linear(x, ws[i[0]]))

            extra = f"   | + SiLU: library pair {timed(lib_pair):6.1f} us, fused kernel {timed(fused):6.1f} us"
        if name in ("down", "o", "qkv"):
            def partials():
                i[0] = (i[0] + 1) % copies
                return ops.linear_partials(x, ws[i[0]], None)

            if ops.linear_partials(x, ws[0], None) is not None:
                extra = f"   | split-K partials only (reduce rides in the next kernel) {timed(partials):6.1f} us"
        print(f"{name:8s} m={m} n={n} k={k}: library {t_lib:7.1f} us ({gb / t_lib * 1e3:5.2f} TB/s)   "
              f"skinny {t_mine:7.1f} us ({gb / t_mine * 1e3:5.2f} TB/s)   max|diff| {err:.3g}{extra}", flush=True)


def fused_gate_up():
    m = int(os.environ.get("M", 256))
    g = torch.Generator(device="cuda").manual_seed(1)
    n, k = 37888, 3584
    ws = [(torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16) for _ in range(3)]
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    i = [0]

    def lib():
        i[0] = (i[0] + 1) % 3
        return ops.silu_and_mul(F.linear(x, ws[i[0]]))

    def mine():
        i[0] = (i[0] + 1) % 3
        return ops.linear_silu_mul(x, ws[i[0]])

    err = (ops.linear_silu_mul(x, ws[0]).float() - ops.silu_and_mul(F.linear(x, ws[0])).float()).abs().max().item()
    print(f"gate_up + SiluAndMul m={m}: library GEMM + silu kernel {timed(lib):7.1f} us   fused {timed(mine):7.1f} us   "
          f"max|diff| {err:.3g}", flush=True)


if __name__ == "__main__":
    fused_gate_up()
    main()
