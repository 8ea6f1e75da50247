"""A/B of the two generations of the decode-batch linear kernel in ONE process, interleaved rounds (programming guide
§5.4 rule 24): MI_LINEAR_V2 is read per call.  Weights rotate over > 256 MB so nothing is served from the Infinity
Cache.  Prints the median and min over rounds per shape."""
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
    m = int(os.environ.get("M", 256))
    var = os.environ.get("AB_VAR", "MI_LINEAR_V2")   # the per-call switch under test: 0 = "v1" column, 1 = "v2" column
    rounds = int(os.environ.get("ROUNDS", 7))
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [("gate_up+silu", 37888, 3584, True), ("down", 3584, 18944, False), ("qkv", 4608, 3584, False),
              ("o", 3584, 3584, False), ("gate_up tp2", 18944, 3584, True), ("down tp2", 3584, 9472, False)]
    for name, n, k, gated in shapes:
        copies = max(2, int(600e6 // (n * k * 2)))
        ws = [(torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        i = [0]

        def run():
            i[0] = (i[0] + 1) % copies
            return ops.linear_silu_mul(x, ws[i[0]]) if gated else ops.linear(x, ws[i[0]])

        def lib():
            i[0] = (i[0] + 1) % copies
            y = F.linear(x, ws[i[0]])
            return ops.silu_and_mul(y) if gated else y

        res = {"v1": [], "v2": [], "lib": []}
        os.environ[var] = "0"
        a = run().float()
        os.environ[var] = "1"
        b = run().float()
        i[0] -= 1
        diff = (a - b).abs().max().item()
        for _ in range(3):
            run()
            lib()
        for _ in range(rounds):
            for tag in ("v1", "v2", "lib"):
                os.environ[var] = "0" if tag == "v1" else "1"
                res[tag].append(timed(lib if tag == "lib" else run))
        gb = n * k * 2 / 1e9
        line = f"{name:14s} m={m} n={n} k={k}:"
        for tag in ("lib", "v1", "v2"):
            med, mn = statistics.median(res[tag]), min(res[tag])
            line += f"  {tag} {med:6.1f} us (min {mn:6.1f}, {gb / med * 1e3:4.2f} TB/s)"
        print(line + f"  max|v1-v2| {diff:.3g}", flush=True)


if __name__ == "__main__":
    main()
