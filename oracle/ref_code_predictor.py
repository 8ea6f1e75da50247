"""ORACLE (test infrastructure only — the product never imports this): CPU restatement of the Qwen3-Omni talker
code predictor, vllm_omni/model_executor/models/common/qwen3_code_predictor.py — the module the reference itself
calls "HF-exact": `_RMSNorm` (:46-62, fp32 variance, cast, then weight), `_rotate_half` + fp32 cos/sin (:65-102),
attention = q_proj/k_proj/v_proj -> per-head q_norm/k_norm -> RoPE -> causal SDPA with GQA -> o_proj (:110-173),
SiLU-gated MLP (:181-191), pre-norm residual layer (:199-223), final norm (:270-280), and the wrapper's RE-PREFILL
loop (:503-626): at step s the whole buffer [talker hidden, emb(c0), ..., emb(c_{s-1}), 0, ...] goes through the
stack and position s predicts code s; "stored" sampling = top-k -> top-p -> softmax -> draw.

parity unpinned for the arithmetic (the reference holds dtype/shape tests only,
tests/model_executor/models/qwen3_tts/test_code_predictor_dtype.py:272-301); the draw uses the sampler oracle
(ref_sampler.py).  fp32 throughout; `round_to` puts activations on the 16-bit grid after every op the device
rounds at, as ref_talker does.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import ref_sampler

F32 = torch.float32


def _rt(x, dt):
    return x if dt is None else x.to(dt).to(F32)


def _rms(x, w, eps, rt=None):
    v = x.to(F32).pow(2).mean(-1, keepdim=True)
    return _rt(w.to(F32) * _rt(x.to(F32) * torch.rsqrt(v + eps), rt), rt)


def _rope(x, pos, theta):
    """x [B, heads, T, d]; pos [T]."""
    d = x.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=F32) / d))
    fr = pos.to(F32)[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    return x * cos + torch.cat((-x2, x1), -1) * sin


def stack(cfg, w: dict, x: torch.Tensor, rt=None) -> torch.Tensor:
    """``CodePredictorBaseModel.forward`` on x [B, T, H] with positions 0..T-1, full causal re-prefill."""
    b, t, _ = x.shape
    nq, nkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    pos = torch.arange(t)
    h = x.to(F32)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        r = h
        y = _rms(h, w[p + "input_layernorm.weight"], cfg.rms_norm_eps, rt)
        q = _rt(y @ w[p + "self_attn.q_proj.weight"].to(F32).t(), rt).view(b, t, nq, d)
        k = _rt(y @ w[p + "self_attn.k_proj.weight"].to(F32).t(), rt).view(b, t, nkv, d)
        v = _rt(y @ w[p + "self_attn.v_proj.weight"].to(F32).t(), rt).view(b, t, nkv, d)
        q = _rms(q, w[p + "self_attn.q_norm.weight"], cfg.rms_norm_eps, None).transpose(1, 2)
        k = _rms(k, w[p + "self_attn.k_norm.weight"], cfg.rms_norm_eps, None).transpose(1, 2)
        q, k = _rt(_rope(q, pos, cfg.rope_theta), rt), _rt(_rope(k, pos, cfg.rope_theta), rt)
        k = k.repeat_interleave(nq // nkv, dim=1)
        vv = v.transpose(1, 2).repeat_interleave(nq // nkv, dim=1)
        a = F.scaled_dot_product_attention(q, k, vv, scale=d ** -0.5, is_causal=True)
        a = _rt(a.transpose(1, 2).reshape(b, t, nq * d), rt)
        h = _rt(r + _rt(a @ w[p + "self_attn.o_proj.weight"].to(F32).t(), rt), rt)
        r = h
        y = _rms(h, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps, rt)
        g = _rt(y @ w[p + "mlp.gate_proj.weight"].to(F32).t(), rt)
        u = _rt(y @ w[p + "mlp.up_proj.weight"].to(F32).t(), rt)
        m = _rt(_rt(F.silu(g) * u, rt) @ w[p + "mlp.down_proj.weight"].to(F32).t(), rt)
        h = _rt(r + m, rt)
    return _rms(h, w["model.norm.weight"], cfg.rms_norm_eps, rt)


def random_weights(cfg, seed: int = 0, dtype=torch.bfloat16) -> dict:
    g = torch.Generator().manual_seed(seed)

    def w(*shape, std=0.05, mean=0.0):
        return (torch.randn(*shape, generator=g) * std + mean).to(dtype)

    h, i, d, nq, nkv, v = (cfg.hidden_size, cfg.intermediate_size, cfg.head_dim, cfg.num_attention_heads,
                           cfg.num_key_value_heads, cfg.vocab_size)
    out = {"model.norm.weight": w(h, std=0.02, mean=1.0)}
    for j in range(cfg.num_code_groups - 1):
        out[f"model.codec_embedding.{j}.weight"] = w(v, h, std=1.0)
        out[f"lm_head.{j}.weight"] = w(v, h, std=0.2)
    for j in range(cfg.num_hidden_layers):
        p = f"model.layers.{j}."
        out.update({p + "self_attn.q_proj.weight": w(nq * d, h), p + "self_attn.k_proj.weight": w(nkv * d, h),
                    p + "self_attn.v_proj.weight": w(nkv * d, h), p + "self_attn.o_proj.weight": w(h, nq * d),
                    p + "self_attn.q_norm.weight": w(d, std=0.02, mean=1.0), p + "self_attn.k_norm.weight": w(d, std=0.02, mean=1.0),
                    p + "mlp.gate_proj.weight": w(i, h), p + "mlp.up_proj.weight": w(i, h), p + "mlp.down_proj.weight": w(h, i),
                    p + "input_layernorm.weight": w(h, std=0.02, mean=1.0),
                    p + "post_attention_layernorm.weight": w(h, std=0.02, mean=1.0)})
    return out


def forward(cfg, w: dict, layer0_code, layer0_embed, last_talker_hidden, rt=None, top_k: int = 50, top_p: float = 0.8,
            seed=None, offset=None, do_sample: bool = True, forced_codes: Optional[torch.Tensor] = None):
    """The wrapper's loop with ``return_proj_buf=True`` / "stored" sampling.  ``forced_codes`` [B, G]: teacher
    forcing — use these codes instead of drawing (the per-step logits then stay comparable with a device run that
    drew them).  Returns (codes [B, G, 1], proj_buf [B, G+1, H], logits list of G-1 [B, V])."""
    b = layer0_code.shape[0]
    g, h = cfg.num_code_groups, cfg.hidden_size
    buf = torch.zeros(b, g + 1, h)
    buf[:, 0] = _rt(last_talker_hidden.reshape(b, -1).to(F32), rt)
    buf[:, 1] = _rt(layer0_embed.reshape(b, -1).to(F32), rt)
    codes = torch.zeros(b, g, dtype=torch.int64)
    codes[:, 0] = layer0_code.reshape(b)
    logits_all = []
    for step in range(1, g):
        hid = stack(cfg, w, buf[:, : g + 1], rt)                      # full re-prefill, zeros past the frontier
        logits = _rt(hid[:, step] @ w[f"lm_head.{step - 1}.weight"].to(F32).t(), rt)
        logits_all.append(logits)
        if forced_codes is not None:
            code = forced_codes[:, step]
        elif do_sample:
            ids, _ = ref_sampler.sample(logits.numpy(), [1.0] * b, [top_k] * b, [top_p] * b, None, None, None,
                                        [0] * b if seed is None else seed,
                                        [(0 if offset is None else int(offset[r])) * g + step for r in range(b)])
            code = torch.from_numpy(ids)
        else:
            code = logits.argmax(-1)
        codes[:, step] = code
        buf[:, step + 1] = _rt(w[f"model.codec_embedding.{step - 1}.weight"].to(F32)[code], rt)
    return codes.unsqueeze(-1), buf, logits_all
