"""Tune the dense GEMMs of the talker decode step with PyTorch TunableOp (hipBLASLt + rocBLAS
solution search) for every TP shard shape bench.py / the worker can run, on ONE GPU.

    python tools/tune_gemms.py [--config qwen2_5_omni_talker] [--batches 256,128,64] [--out <csv>]

The result CSV is committed under vllm-omni_amd/vllm_omni_amd/tuned/ and loaded (tuning OFF) by
``vllm_omni_amd.tuned.enable_tuned_gemms()``.  The GEMMs themselves stay library GEMMs.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def shapes(cfg, tp):
    from vllm_omni_amd.distributed import partition_heads

    nq, nkv, _ = partition_heads(cfg.num_attention_heads, cfg.num_key_value_heads, tp)
    d, h, i = cfg.head_dim, cfg.hidden_size, cfg.intermediate_size
    moe = []
    if getattr(cfg, "num_experts", 0):   # the sparse-MoE block's dense GEMMs: router, shared expert (experts: in-tree kernels)
        moe.append(("moe.router", cfg.num_experts, h, False))
        sh = getattr(cfg, "shared_expert_intermediate_size", 0) or 0
        if sh:
            moe += [("moe.shared_gate_up", 2 * sh, h, False), ("moe.shared_down", h, sh, False)]
    return moe + [  # (name, N, K, bias)
        ("thinker_to_talker_proj", h, cfg.embedding_size, True),
        ("qkv_proj", (nq + 2 * nkv) * d, h, cfg.attention_bias),
        ("o_proj", h, nq * d, False),
        ("gate_up_proj", 2 * i // tp, h, False),
        ("down_proj", h, i // tp, False),
        ("lm_head", cfg.vocab_size // tp, h, False),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="qwen2_5_omni_talker")
    ap.add_argument("--batches", default="256,128,64")
    ap.add_argument("--tps", default="1,2,4")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tunableop_gfx950.csv"))
    ap.add_argument("--max-ms", type=int, default=2000)
    ap.add_argument("--code-predictor", action="store_true",
                    help="the Qwen3-Omni code predictor's linears (H 1024, 16/8 heads of 128, I 3072, vocab 2048) at --batches "
                         "rows and twice as many (its 2-token prefill) instead of a talker config")
    a = ap.parse_args()
    import torch.cuda.tunable as tn
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json(a.config)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    tn.enable(True)
    tn.tuning_enable(True)
    tn.set_max_tuning_duration(a.max_ms)
    tn.set_filename(a.out)
    dev = torch.device("cuda:0")
    seen = set()
    def predictor_shapes():
        from vllm_omni_amd.code_predictor import CodePredictorConfig

        c = CodePredictorConfig()
        h, d = c.hidden_size, c.head_dim
        return [("cp.qkv_proj", (c.num_attention_heads + 2 * c.num_key_value_heads) * d, h, c.attention_bias),
                ("cp.o_proj", h, c.num_attention_heads * d, False), ("cp.gate_up_proj", 2 * c.intermediate_size, h, False),
                ("cp.down_proj", h, c.intermediate_size, False), ("cp.lm_head", c.vocab_size, h, False)]

    if a.code_predictor:
        a.tps = "1"
        a.batches = ",".join(str(x) for m in a.batches.split(",") for x in (int(m), 2 * int(m)))
    for tp in [int(x) for x in a.tps.split(",")]:
        try:
            shp = predictor_shapes() if a.code_predictor else shapes(cfg, tp)
        except ValueError as e:
            print(f"tp={tp}: skipped ({e})")
            continue
        for m in [int(x) for x in a.batches.split(",")]:
            for name, n, k, bias in shp:
                key = (m, n, k, bias)
                if key in seen:
                    continue
                seen.add(key)
                x = torch.randn(m, k, dtype=torch.bfloat16, device=dev)
                w = torch.randn(n, k, dtype=torch.bfloat16, device=dev) * 0.02
                b = torch.zeros(n, dtype=torch.bfloat16, device=dev) if bias else None
                for _ in range(3):
                    F.linear(x, w, b)
                torch.cuda.synchronize()
                print(f"tuned tp={tp} M={m} {name}: N={n} K={k} bias={bias}", flush=True)
    if hasattr(tn, "write_file"):
        tn.write_file(a.out)  # otherwise TunableOp writes the file when the process exits
    print("wrote", a.out)


if __name__ == "__main__":
    main()
