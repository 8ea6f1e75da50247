"""Small prefill launches on long contexts — the ragged last chunk of a prompt, a TP shard with one KV head — with and
without the context of each query tile split into segments (ops.paged_attn_prefill(..., max_seq_len=, segments=)).
One layer, bf16, block 16, head-major cache; time per launch from HIP events (metadata prepared once)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))
import torch  # noqa: E402

from vllm_omni_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    bs = 16
    for name, nq, nkv, d, ctx, chunk in (("last chunk, 77 tokens on 16k", 28, 4, 128, 16384, 77),
                                          ("last chunk, 300 tokens on 8k", 28, 4, 128, 8192, 300),
                                          ("verify, 8 tokens on 32k", 28, 4, 128, 32768, 8),
                                          ("chunk 2048 on 6k, TP=4 rank (7/1 heads)", 7, 1, 128, 6144, 2048),
                                          ("whole 8k prompt, TP=4 rank (7/1 heads)", 7, 1, 128, 0, 8192),
                                          ("chunk 2048 on 6k, S-B TP=2 rank (8/1 heads, d 64)", 8, 1, 64, 6144, 2048),
                                          ("chunk 2048 on 6k, 28/4 heads", 28, 4, 128, 6144, 2048)):
        total = ctx + chunk
        nblk = (total + bs - 1) // bs
        kv = torch.empty(nblk + 1, 2, nkv, bs, d, dtype=torch.bfloat16, device=dev).normal_()
        kb, vb = kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)
        table = (torch.randperm(nblk, device=dev) + 1).view(1, -1).to(torch.int32)
        q = torch.randn(chunk, nq * d, dtype=torch.bfloat16, device=dev)
        out = torch.empty_like(q)
        sl = torch.tensor([total], dtype=torch.int32, device=dev)
        qsl = torch.tensor([0, chunk], dtype=torch.int32, device=dev)
        res = {}
        for seg in (1, 0):
            f = lambda: ops.paged_attn_prefill(out, q, kb, vb, table, sl, qsl, chunk, d ** -0.5, max_seq_len=total, segments=seg)  # noqa: E731
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[seg] = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:52s}: unsplit {res[1]:8.1f} us   automatic segments {res[0]:8.1f} us   ({res[1] / res[0]:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
