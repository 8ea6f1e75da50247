"""Generates tests/golden/hf_qwen3_talker_mini.pt — run once in the BUILD container (needs `transformers`).

Two more modules the reference wraps whose arithmetic it does not pin itself, built FROM CONFIG OBJECTS with seeded
random weights (no hub access) and run eagerly in fp32:

* `Qwen3OmniMoeTalkerCodePredictorModel` — the transformer inside the code predictor; the reference's
  vllm_omni/model_executor/models/common/qwen3_code_predictor.py states it re-implements exactly this (":46-62
  _RMSNorm matching HuggingFace's implementation exactly", ":72-102 RoPE matching HuggingFace's").  Pins
  oracle/ref_code_predictor.stack.
* `Qwen3OmniMoeTalkerTextSparseMoeBlock` — router (fp32 softmax -> top-k, norm_topk_prob), the expert MLPs, the
  sigmoid-gated shared expert; the talker MoE layer behind qwen3_omni_moe_talker.py:476-513 (upstream
  `SharedFusedMoE`).  Pins oracle/ref_moe.moe_block.

    python tests/golden/make_golden_qwen3.py
"""
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from transformers.models.qwen3_omni_moe import configuration_qwen3_omni_moe as c
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as m

    torch.manual_seed(4321)
    out = {}
    # ---- code predictor stack
    cp = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=32,
              intermediate_size=96, vocab_size=40, num_code_groups=6, rms_norm_eps=1e-6)
    cfg = c.Qwen3OmniMoeTalkerCodePredictorConfig(**cp)
    cfg._attn_implementation = "eager"
    model = m.Qwen3OmniMoeTalkerCodePredictorModel(cfg).eval().float()
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if "norm" in k:
                v.copy_((1.0 + 0.1 * torch.randn_like(v)).half().float())
            elif "weight" in k and v.dim() == 2:
                v.copy_((0.1 * torch.randn_like(v)).half().float())
        x = (torch.randn(3, 7, 64) * 0.5).half().float()
        y = model(inputs_embeds=x, use_cache=False).last_hidden_state
    out["code_predictor"] = {
        "config": dict(cp, rope_theta=10000.0),
        "weights_fp16": {"model." + k: v.half() for k, v in model.state_dict().items()
                         if k.startswith("layers.") or k == "norm.weight"},
        "inputs_embeds_fp16": x.half(), "expected_hidden_fp32": y,
    }
    # ---- sparse MoE block, both router conventions
    for tag, norm_topk in (("moe", False), ("moe_norm_topk", True)):
        mc = dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1, head_dim=32,
                  intermediate_size=96, moe_intermediate_size=32, num_experts=8, num_experts_per_tok=3,
                  shared_expert_intermediate_size=48, norm_topk_prob=norm_topk, vocab_size=40)
        tcfg = c.Qwen3OmniMoeTalkerTextConfig(**mc)
        blk = m.Qwen3OmniMoeTalkerTextSparseMoeBlock(tcfg).eval().float()
        with torch.no_grad():
            for p in blk.parameters():
                p.copy_((0.15 * torch.randn_like(p)).half().float())
            x = (torch.randn(1, 19, 64) * 0.7).half().float()
            y = blk(x)[0] if isinstance(blk(x), tuple) else blk(x)
        out[tag] = {"config": mc, "weights_fp16": {k: v.half() for k, v in blk.state_dict().items()},
                    "x_fp16": x[0].half(), "expected_fp32": y.reshape(19, 64)}
    torch.save(out, os.path.join(HERE, "hf_qwen3_talker_mini.pt"))
    print("wrote", os.path.join(HERE, "hf_qwen3_talker_mini.pt"), {k: list(v.keys()) for k, v in out.items()})


if __name__ == "__main__":
    main()
