"""Micro-benchmarks of the small kernels around the attention path (HIP events on the launch stream):
device sampler (N2), fused KV pack (N3), W8A8 int8 linear (A13) against the bf16 library GEMM.

    python tools/bench_misc_kernels.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))

import torch  # noqa: E402

from vllm_omni_amd import kv_transfer, ops  # noqa: E402


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    # ---- sampler: the shipped talker stage config, batch 256, vocab 8448
    n, v = 256, 8448
    logits = torch.randn(n, v, device=dev) * 3
    f = lambda x: torch.full((n,), x, device=dev)  # noqa: E731
    seed = torch.arange(n, device=dev, dtype=torch.int64)
    off = torch.zeros(n, device=dev, dtype=torch.int64)
    bm = torch.zeros(n + 1, (v + 31) // 32, dtype=torch.int32, device=dev)
    rows = torch.arange(n, device=dev, dtype=torch.int32)
    us = timeit(lambda: ops.sample_tokens(logits, f(0.9), seed, off, top_k=f(40).int(), top_p=f(0.8), rep_penalty=f(1.05),
                                          seen_bitmap=bm, seen_row=rows))
    print(f"sampler  T=0.9 top_k=40 top_p=0.8 rep=1.05, {n} rows x vocab {v}: {us:7.1f} us")
    us = timeit(lambda: ops.sample_tokens(logits, f(0.0), seed, off))
    print(f"sampler  greedy rows only:                                  {us:7.1f} us")

    def torch_sampler():
        lf = logits / 0.9
        kth = torch.topk(lf, 40, dim=-1).values[:, -1:]
        lf = lf.masked_fill(lf < kth, float("-inf"))
        probs = torch.softmax(lf, dim=-1)
        sp, si = torch.sort(probs, dim=-1, descending=True)
        keep = (sp.cumsum(-1) - sp) < 0.8
        probs = torch.zeros_like(probs).scatter_(-1, si, sp * keep)
        return torch.multinomial(probs / probs.sum(-1, keepdim=True), 1)
    print(f"         torch ops (topk + sort + scatter + multinomial):   {timeit(torch_sampler):7.1f} us")

    # ---- KV pack: one finished request, 28 layers x 4096 tokens x 4 heads x 128, head-major blocks
    nl, h, d, bs, seq = 28, 4, 128, 16, 4096
    nb = seq // bs + 8
    kv = [torch.randn(nb, 2, h, bs, d, device=dev, dtype=torch.bfloat16) for _ in range(nl)]
    caches = [(x[:, 0].permute(0, 2, 1, 3), x[:, 1].permute(0, 2, 1, 3)) for x in kv]
    ids = (torch.randperm(nb - 1)[: seq // bs] + 1).tolist()
    us = timeit(lambda: kv_transfer.pack_request_kv(caches, "r", ids, seq, bs, "bfloat16"), iters=10, warm=2)
    nbytes = nl * 2 * seq * h * d * 2
    print(f"kv pack  {nl} layers x {seq} tokens ({nbytes / 1e6:.0f} MB payload), host wrapper included: {us:7.1f} us "
          f"= {2 * nbytes / us / 1e6:.2f} TB/s read+write")

    def torch_pack():
        idx = torch.tensor(ids, device=dev)
        ks = [c[0].index_select(0, idx).flatten(0, 1)[:seq].contiguous() for c in caches]
        vs = [c[1].index_select(0, idx).flatten(0, 1)[:seq].contiguous() for c in caches]
        return kv_transfer.KVCacheTransferData("r", {"key_cache": ks, "value_cache": vs}, ids, {}).to_gpu_tensor()
    print(f"         torch extract + to_gpu_tensor:                      {timeit(torch_pack, iters=10, warm=2):7.1f} us")

    # ---- int8 W8A8 linear vs bf16 (config #5 family: M = 512; S-A layer shapes)
    for m, n_, k in ((512, 4608, 3584), (512, 3584, 3584), (512, 37888, 3584), (512, 3584, 18944), (256, 37888, 3584)):
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        w = torch.randn(n_, k, device=dev, dtype=torch.bfloat16) * 0.02
        wq, ws = ops.dynamic_int8_quant(w)
        t_bf16 = timeit(lambda: torch.nn.functional.linear(x, w))

        def int8():
            xq, xs = ops.dynamic_int8_quant(x)
            return ops.int8_gemm_dequant(xq, xs, wq, ws, None, torch.bfloat16)
        t_i8 = timeit(int8)
        print(f"linear   M={m:4d} N={n_:6d} K={k:6d}: bf16 library {t_bf16:7.1f} us ({n_ * k * 2 / t_bf16 / 1e6:5.2f} TB/s of weights) | "
              f"int8 W8A8 {t_i8:7.1f} us ({n_ * k / t_i8 / 1e6:5.2f} TB/s)")


if __name__ == "__main__":
    main()
