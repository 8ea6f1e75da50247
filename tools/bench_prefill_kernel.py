"""Micro-benchmark of the paged-attention prefill kernel (one layer): one prompt of --prompt tokens,
processed in chunks of --chunk tokens against the growing paged cache (BASELINE config #4 shape).

Reports achieved TFLOP/s with the causal FLOP count 2*2*d*n_q*sum_i(ctx_i + (chunk_i+1)/2)*chunk_i.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))

import torch  # noqa: E402

from vllm_omni_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=8192)
    ap.add_argument("--chunks", default="8192,4096,2048")
    ap.add_argument("--nq", type=int, default=28)
    ap.add_argument("--nkv", type=int, default=4)
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--segments", type=int, default=0, help="0 automatic (as the runner calls it), 1 never split the context")
    a = ap.parse_args()
    d = a.d
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    nblk = a.prompt // a.bs
    kv = torch.empty(nblk + 1, 2, a.nkv, a.bs, d, dtype=torch.bfloat16, device=dev).normal_()
    kb, vb = kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)
    table = (torch.randperm(nblk, device=dev) + 1).view(1, -1).to(torch.int32)
    q = torch.randn(a.prompt, a.nq * d, dtype=torch.bfloat16, device=dev)
    out = torch.empty_like(q)
    for chunk in [int(c) for c in a.chunks.split(",")]:
        n = a.prompt // chunk
        flops = sum(2 * 2 * d * a.nq * chunk * (i * chunk + (chunk + 1) / 2) for i in range(n))

        # metadata made once: a host->device copy per launch would be timed as kernel time
        sls = [torch.tensor([(i + 1) * chunk], dtype=torch.int32, device=dev) for i in range(n)]
        qsl = torch.tensor([0, chunk], dtype=torch.int32, device=dev)

        def run():
            for i in range(n):
                ops.paged_attn_prefill(out[i * chunk:(i + 1) * chunk], q[i * chunk:(i + 1) * chunk], kb, vb, table, sls[i],
                                       qsl, chunk, d ** -0.5, max_seq_len=(i + 1) * chunk, segments=a.segments)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f"prompt={a.prompt} chunk={chunk:5d} ({n} launches) heads {a.nq}/{a.nkv}: {ms:8.3f} ms/layer  "
              f"{flops / ms / 1e9:7.1f} TFLOP/s ({flops / ms / 1e9 / 2500:.1%} of 2.5 PF dense bf16)")


if __name__ == "__main__":
    main()
