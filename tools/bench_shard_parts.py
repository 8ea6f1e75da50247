"""Which side bounds the decode-batch linear kernel at a shard shape: the whole kernel, its LDS-DMA ring alone
(MI_LINEAR_DBG=2: no fragment reads / MFMAs), its consumers alone (MI_LINEAR_DBG=1: no DMA), for chosen tile plans.
  M=128 python tools/bench_shard_parts.py gate_up:4,3 down:4,3,3 down:8,2,4"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vllm-omni_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_shard_gemm import timed  # noqa: E402
from vllm_omni_amd import ops  # noqa: E402


def main():
    m = int(os.environ.get("M", 128))
    tp = int(os.environ.get("TP", 4))
    nq, nkv = 28 // tp, max(1, 4 // tp)
    shapes = {"qkv": ((nq + 2 * nkv) * 128, 3584, False), "o": (3584, nq * 128, False),
              "gate_up": (2 * 18944 // tp, 3584, True), "down": (3584, 18944 // tp, False)}
    g = torch.Generator(device="cuda").manual_seed(0)
    for spec in sys.argv[1:]:
        name, plan = spec.split(":")
        n, k, gated = shapes[name]
        copies = max(2, int(600e6 // (n * k * 2)))
        ws = [(torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        i = [0]

        def nxt():
            i[0] = (i[0] + 1) % copies
            return ws[i[0]]

        os.environ["MI_GATED_PLAN" if gated else "MI_LINEAR_PLAN"] = plan
        out = []
        for dbg in (0, 2, 1, 8, 3):
            os.environ["MI_LINEAR_DBG"] = str(dbg)
            if gated:
                t = timed(lambda: ops.linear_silu_mul(x, nxt()))
            else:
                t = timed(lambda: (ops.linear_partials(x, nxt(), None) or ops.linear(x, nxt())))
            out.append(t)
        os.environ.pop("MI_LINEAR_DBG")
        extra = ""
        if not gated:
            res = []
            for fr in (False, True):
                ops.LINEAR_FUSED_REDUCE = fr
                res.append(timed(lambda: ops.linear(x, nxt())))
            ops.LINEAR_FUSED_REDUCE = True
            extra = f" || with its reduce: separate launch {res[0]:6.1f} us, inside the launch {res[1]:6.1f} us"
        os.environ.pop("MI_GATED_PLAN" if gated else "MI_LINEAR_PLAN")
        print(f"{name:8s} m={m} plan {plan:8s}: whole {out[0]:6.1f} us | DMA ring only {out[1]:6.1f} | consumers only {out[2]:6.1f} "
              f"| no output stores {out[3]:6.1f} | neither (launch + barriers) {out[4]:6.1f}{extra}", flush=True)


if __name__ == "__main__":
    main()
