"""Fused sparse-MoE at the Qwen3-Omni talker shape (E 128, top-8, H 1024, I 384, shared 768):
per-kernel and whole-layer time, achieved expert-weight streaming rate, against the per-expert torch loop the
HF module runs.

    python tools/bench_moe.py [--tokens 512,256,64]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vllm_omni_amd import ops  # noqa: E402
from vllm_omni_amd.moe import MoEConfig, Qwen3OmniMoeSparseMoeBlock  # noqa: E402


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
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="512,256,64")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = MoEConfig(shared_expert_intermediate_size=768)
    gen = torch.Generator(device=dev).manual_seed(0)
    blk = Qwen3OmniMoeSparseMoeBlock(cfg, torch.bfloat16, dev, gen)
    e, k, h, i = cfg.num_experts, cfg.num_experts_per_tok, cfg.hidden_size, cfg.moe_intermediate_size
    wbytes = e * (2 * i * h + h * i) * 2
    for t in [int(x) for x in a.tokens.split(",")]:
        x = torch.randn(t, h, device=dev, dtype=torch.bfloat16)
        logits = F.linear(x, blk.gate_weight)
        t_layer = timeit(lambda: blk(x))
        t_fused = timeit(lambda: ops.fused_moe(x, logits, blk.gate_up_proj, blk.down_proj, k, False))
        # the two expert-GEMM kernels x the two tile heights, interleaved rounds in one process
        arms = [("ring", 32), ("ring", 64), ("direct", 32), ("direct", 64)]
        if t * k >= 64 * e:     # prefill-sized: the tall tiles of the ring kernel
            arms = [("ring", 64), ("ring", 128), ("ring", 256), ("direct", 64)]
        res = {a_: [] for a_ in arms}
        for _ in range(4):
            for mode, rpt in arms:
                os.environ["MI_MOE_GEMM"] = mode
                res[(mode, rpt)].append(timeit(lambda: ops.fused_moe(x, logits, blk.gate_up_proj, blk.down_proj, k, False,
                                                                     rows_per_tile=rpt), iters=10, warm=2))
        os.environ.pop("MI_MOE_GEMM", None)
        print("   expert path by GEMM kernel / rows per tile: " + ", ".join(
            f"{m_}/{r_}: {sorted(v)[len(v) // 2]:.1f} us" for (m_, r_), v in res.items()))
        ids, w, _ = ops.moe_route(logits, k, False, round_to=torch.bfloat16)
        hit = int(torch.unique(ids).numel())

        def torch_loop():
            out = torch.zeros_like(x)
            for ex in torch.unique(ids).tolist():
                tok, pos = (ids == ex).nonzero(as_tuple=True)
                cur = x[tok]
                g, u = F.linear(cur, blk.gate_up_proj[ex]).chunk(2, -1)
                y = F.linear(F.silu(g) * u, blk.down_proj[ex]) * w[tok, pos, None].to(x.dtype)
                out.index_add_(0, tok, y)
            return out
        t_loop = timeit(torch_loop, iters=3, warm=1) if t <= 1024 else float("nan")
        print(f"T={t:4d}: {hit} experts hit | expert path (route..combine) {t_fused:7.1f} us = "
              f"{wbytes * hit / e / t_fused / 1e6:5.2f} TB/s of expert weights | whole layer incl. router + shared expert "
              f"{t_layer:7.1f} us | per-expert torch loop {t_loop:8.1f} us")


if __name__ == "__main__":
    main()
