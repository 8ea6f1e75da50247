"""Code predictor per-call time at the real shape (H 1024, 5 layers, 16/8 heads, G = 32):
incremental decode with a private KV cache (this repo) — eager and as one hipGraph — against re-prefilling the
whole buffer at every step with the same kernels (what the reference's wrapper does).

    python tools/bench_code_predictor.py [--batch 256]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))

import torch  # noqa: E402

from vllm_omni_amd.code_predictor import CodePredictorConfig, MI355XCodePredictor  # noqa: E402
from vllm_omni_amd.layers import AttentionMetadata  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-tuned", action="store_true", help="library-default GEMM selections (the runner loads the tuned CSV)")
    ap.add_argument("--skip-reprefill", action="store_true")
    a = ap.parse_args()
    if not a.no_tuned:   # what GPUARModelRunner.load_model does before it builds the predictor
        from vllm_omni_amd.tuned import enable_tuned_gemms

        print("tuned GEMM selections loaded:", enable_tuned_gemms())
    dev = torch.device("cuda:0")
    cfg = CodePredictorConfig()
    b, g, h = a.batch, cfg.num_code_groups, cfg.hidden_size
    cp = MI355XCodePredictor(cfg, torch.bfloat16, dev, max_batch=b, seed=0)
    code0 = torch.randint(0, cfg.vocab_size, (b,), device=dev)
    emb0 = torch.randn(b, h, device=dev, dtype=torch.bfloat16)
    hid = torch.randn(b, h, device=dev, dtype=torch.bfloat16)
    t_eager = timeit(lambda: cp(code0, emb0, hid))
    cp.capture(b)
    t_graph = timeit(lambda: cp(code0, emb0, hid))

    def reprefill():  # the reference's schedule with this repo's kernels: G-1 passes over all G+1 positions
        t = g + 1
        pos = torch.arange(t, device=dev).repeat(b)
        slots = (cp.block_table[:b].to(torch.int64) * cp.BLOCK + torch.arange(t, device=dev)).reshape(-1)
        md = AttentionMetadata(num_actual_tokens=b * t, max_query_len=t,
                               query_start_loc=torch.arange(0, b * t + 1, t, dtype=torch.int32, device=dev),
                               seq_lens=torch.full((b,), t, dtype=torch.int32, device=dev), max_seq_len=t,
                               block_table=cp.block_table[:b], slot_mapping=slots, num_reqs=b, num_decodes=0)
        for step in range(1, g):
            x = cp.proj_buf[:b].reshape(b * t, h).clone()
            full = cp._stack(x, pos, md).view(b, t, h)
            torch.nn.functional.linear(full[:, step], cp.lm_head[step - 1])
    t_re = float("nan") if a.skip_reprefill else timeit(reprefill, iters=3, warm=1)
    print(f"code predictor, batch {b}, G = {g}: incremental eager {t_eager:.2f} ms | one hipGraph {t_graph:.2f} ms | "
          f"re-prefill schedule (transformer passes only) {t_re:.2f} ms")


if __name__ == "__main__":
    main()
