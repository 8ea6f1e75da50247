"""Generates tests/golden/hf_talker_mini.pt — run once in the BUILD container (needs `transformers`).

The reference's arithmetic lives in un-vendored upstream vllm==0.19.0 and the reference itself cannot
be imported here (ModuleNotFoundError: aenum / vllm — SURVEY.md §8c).  What IS installed is the HF
`transformers` implementation of the very architecture the reference wraps
(vllm_omni/model_executor/models/qwen2_5_omni/qwen2_5_omni_talker.py:6-7 imports its config/modules),
so this script builds `Qwen2_5OmniTalkerModel` FROM A CONFIG OBJECT with seeded random weights (no hub
access), runs it eagerly in fp32 on a 7-token sequence with three DIFFERENT M-RoPE position rows, and
stores inputs, weights (fused names, fp16-exact values) and outputs as a fixture.  The oracle must
reproduce these outputs both as one causal prefill over a shuffled paged cache and token by token.

    python tests/golden/make_golden.py
"""
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from transformers.models.qwen2_5_omni.configuration_qwen2_5_omni import Qwen2_5OmniTalkerConfig
    from transformers.models.qwen2_5_omni.modeling_qwen2_5_omni import Qwen2_5OmniTalkerModel

    torch.manual_seed(1234)
    cfg = Qwen2_5OmniTalkerConfig(
        hidden_size=64, embedding_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
        head_dim=16, intermediate_size=128, vocab_size=64, max_position_embeddings=256, rms_norm_eps=1e-6,
        rope_parameters={"rope_theta": 1000000.0, "rope_type": "default", "mrope_section": [2, 3, 3]},
    )
    cfg._attn_implementation = "eager"
    model = Qwen2_5OmniTalkerModel(cfg).eval().float()
    sd = model.state_dict()
    # make every weight fp16-exact so the fixture can store fp16 without changing the function
    with torch.no_grad():
        for k, v in sd.items():
            if "norm" in k:
                v.copy_((1.0 + 0.1 * torch.randn_like(v)).half().float())
            elif k.endswith("bias"):
                v.copy_((0.1 * torch.randn_like(v)).half().float())
            else:
                v.copy_((0.08 * torch.randn_like(v)).half().float())
    T = 7
    x = (torch.randn(1, T, cfg.hidden_size) * 0.5).half().float()
    pos = torch.stack([torch.arange(T), torch.arange(T) * 2 + 1, torch.tensor([0, 0, 1, 1, 5, 9, 9])]).view(3, 1, T)
    with torch.no_grad():
        out = model(inputs_embeds=x, position_ids=pos, use_cache=False).last_hidden_state[0]

    fused = {"language_model.model.norm.weight": sd["norm.weight"]}
    for i in range(cfg.num_hidden_layers):
        p, q = f"layers.{i}.", f"language_model.model.layers.{i}."
        fused[q + "self_attn.qkv_proj.weight"] = torch.cat(
            [sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]])
        fused[q + "self_attn.qkv_proj.bias"] = torch.cat(
            [sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"]])
        fused[q + "self_attn.o_proj.weight"] = sd[p + "self_attn.o_proj.weight"]
        fused[q + "mlp.gate_up_proj.weight"] = torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]])
        fused[q + "mlp.down_proj.weight"] = sd[p + "mlp.down_proj.weight"]
        fused[q + "input_layernorm.weight"] = sd[p + "input_layernorm.weight"]
        fused[q + "post_attention_layernorm.weight"] = sd[p + "post_attention_layernorm.weight"]
    assert not any(k.endswith("o_proj.bias") for k in sd), "o_proj is bias-free in the reference wiring too"
    fixture = {
        "generator": "tests/golden/make_golden.py; transformers " + __import__("transformers").__version__,
        "config": {"hidden_size": 64, "embedding_size": 64, "num_hidden_layers": 2, "num_attention_heads": 4,
                   "num_key_value_heads": 2, "head_dim": 16, "intermediate_size": 128, "vocab_size": 64,
                   "rms_norm_eps": 1e-6, "rope_theta": 1000000.0, "max_position_embeddings": 256,
                   "mrope_section": [2, 3, 3], "attention_bias": True},
        "weights_fp16": {k: v.half() for k, v in fused.items()},
        "inputs_embeds_fp16": x[0].half(),
        "position_ids": pos[:, 0].clone(),
        "expected_hidden_fp32": out.clone(),
    }
    path = os.path.join(HERE, "hf_talker_mini.pt")
    torch.save(fixture, path)
    print("wrote", path, os.path.getsize(path), "bytes; |out| max", out.abs().max().item())


if __name__ == "__main__":
    main()
