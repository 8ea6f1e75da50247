"""Where a 256-row projection spends its time: the kernel with its DMA stubbed (fragment reads + MFMAs only), with
its consumers stubbed (LDS-DMA ring only) and whole — both kernel generations, one process, interleaved rounds.
Also: x with a padded row stride (does the 7168-byte stride of the activation rows camp on L2 channels?)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vllm-omni_amd"))
from vllm_omni_amd import ops  # noqa: E402


def timed(fn, iters=20):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3


def main():
    m = int(os.environ.get("M", 256))
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, n, k, gated in [("gate_up+silu", 37888, 3584, True), ("down", 3584, 18944, False)]:
        copies = max(2, int(600e6 // (n * k * 2)))
        ws = [(torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        xp = torch.zeros(m, k + 64, device="cuda", dtype=torch.bfloat16)
        xp[:, :k] = x
        xpad = xp[:, :k]
        i = [0]

        def run(xx):
            i[0] = (i[0] + 1) % copies
            return ops.linear_silu_mul(xx, ws[i[0]]) if gated else ops.linear(xx, ws[i[0]])

        variants = [("v1 whole", "0", "0", x), ("v2 whole", "1", "0", x), ("v2 no-DMA", "1", "1", x),
                    ("v2 DMA only", "1", "2", x), ("v2 no-DMA, reads only", "1", "17", x),
                    ("v2 no-DMA, MFMAs only", "1", "33", x), ("v2 no-DMA, no stores", "1", "9", x),
                    ("v2 no-DMA, MFMAs only, no stores", "1", "41", x), ("v2 DMA only, no stores", "1", "10", x)]
        res = {v[0]: [] for v in variants}
        for _ in range(5):
            for tag, v2, dbg, xx in variants:
                os.environ["MI_LINEAR_V2"], os.environ["MI_LINEAR_DBG"] = v2, dbg
                run(xx)
                res[tag].append(timed(lambda: run(xx)))
        os.environ["MI_LINEAR_DBG"] = "0"
        print(f"--- {name} m={m} n={n} k={k}")
        for tag, *_ in variants:
            print(f"   {tag:28s} {statistics.median(res[tag]):7.1f} us (min {min(res[tag]):7.1f})", flush=True)


if __name__ == "__main__":
    main()
