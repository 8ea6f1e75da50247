"""Few launches of the decode-batch linear kernel on the down_proj / gate_up_proj shapes, for rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vllm-omni_amd"))
from vllm_omni_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for n, k in [(3584, 18944), (37888, 3584)]:
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    x = torch.randn(256, k, device="cuda", generator=g).to(torch.bfloat16)
    for _ in range(4):
        y = ops.linear(x, w)
    torch.cuda.synchronize()
    print("done", n, k, float(y.float().abs().mean()), flush=True)
