"""Time mi_sample_tokens (one launch) at the code predictor's and the talker's shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))
import torch  # noqa: E402

from vllm_omni_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for rows, vocab, k, p in ((32, 2048, 50, 0.8), (32, 2048, 0, 1.0), (32, 2048, 50, 1.0), (32, 2048, 0, 0.8), (512, 2048, 50, 0.8), (256, 8448, 40, 0.8), (256, 8448, 0, 0.8),
                              (256, 8448, 40, 1.0), (256, 8448, 0, 1.0)):
        logits = torch.randn(rows, vocab, device=dev, dtype=torch.bfloat16) * 3
        temp = torch.full((rows,), 0.9, device=dev)
        topk = torch.full((rows,), k, dtype=torch.int32, device=dev)
        topp = torch.full((rows,), p, device=dev)
        seed = torch.arange(rows, dtype=torch.int64, device=dev)
        off = torch.zeros(rows, dtype=torch.int64, device=dev)
        f = lambda: ops.sample_tokens(logits, temp, seed, off, top_k=topk, top_p=topp)  # noqa: E731
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                f()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"rows {rows:4d} vocab {vocab:5d} top_k {k:3d} top_p {p}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per launch")


if __name__ == "__main__":
    main()
