"""ORACLE — test infrastructure, NOT product code.

CPU restatement of one talker engine step: projection -> 28x
{RMSNorm, QKV, RoPE, KV write, paged attention, o_proj, RMSNorm, gate_up, SiLU-mul, down} ->
final norm -> logits -> bad_word mask -> greedy argmax.

Layer order / residual flow: vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:146-156
(attention), :205-225 (layer), :300-326 (stack); talker wrapper: qwen2_5_omni_talker.py:123-174;
per-step input: qwen2_5_omni.py:796-804.  Arithmetic in fp32 with the model-dtype rounding points of
a 16-bit checkpoint at every op boundary (``round_to``).  Weight names are the fused names the
product exposes (``named_weights``).  Parity status: see oracle/ref_ops.py header.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ref_ops as R

F32 = torch.float32


def _lin(x, w, b=None, rt=None):
    y = x @ w.to(F32).t()
    if b is not None:
        y = y + b.to(F32)
    return R._rt(y, rt)


def decode_embeds(weights: dict, last_token_ids: torch.Tensor, step_vec: Optional[torch.Tensor], rt=None):
    e = weights["language_model.model.embed_tokens.weight"].to(F32)[last_token_ids]
    return e if step_vec is None else R._rt(step_vec.to(F32) + e, rt)


def talker_step(cfg, weights: dict, inputs_embeds: torch.Tensor, positions: torch.Tensor, kv_caches: Sequence,
                block_table: torch.Tensor, seq_lens: torch.Tensor, query_start_loc: torch.Tensor,
                slot_map: torch.Tensor, cos_sin_cache: torch.Tensor, round_to: Optional[torch.dtype] = None,
                num_layers: Optional[int] = None, quant: Optional[str] = None):
    """Run the stack on ``inputs_embeds`` [T, E] (all requests' scheduled tokens, varlen).

    kv_caches: per layer (key_blocks, value_blocks) CPU views, UPDATED IN PLACE at ``slot_map``.
    Returns (hidden [T, H] after the final norm, as fp32 values on the ``round_to`` grid).
    """
    rt = round_to
    if "thinker_to_talker_proj.weight" in weights:   # Qwen2.5-Omni; the Qwen3-Omni talker has no input projection
        h = _lin(inputs_embeds.to(F32), weights["thinker_to_talker_proj.weight"], weights["thinker_to_talker_proj.bias"], rt)
    else:
        h = inputs_embeds.to(F32)
    return decoder_stack(cfg, weights, h, positions, kv_caches, block_table, seq_lens, query_start_loc, slot_map,
                         cos_sin_cache, round_to, num_layers, quant)


def decoder_stack(cfg, weights: dict, h: torch.Tensor, positions: torch.Tensor, kv_caches: Sequence,
                  block_table: torch.Tensor, seq_lens: torch.Tensor, query_start_loc: torch.Tensor,
                  slot_map: torch.Tensor, cos_sin_cache: torch.Tensor, round_to: Optional[torch.dtype] = None,
                  num_layers: Optional[int] = None, quant: Optional[str] = None):
    """``Qwen2Model.forward`` on already-projected embeddings ``h`` [T, H] (qwen2_old.py:300-326).

    quant="int8": the four linears of every block run the reference's W8A8 scheme (int8_config.py:180-212,
    323-342: per-output-channel weight scale, dynamic per-token activation scale, int32 accumulation)."""
    rt = round_to
    if quant == "int8":
        def _lin(x, w, b, rt_):  # noqa: F811 — shadows the module-level fp32 linear inside this call
            wq, ws = R.int8_weight_quant(w.to(F32))
            return R._rt(R.int8_linear(x, wq, ws, b), rt_)
    elif quant is not None:
        raise ValueError(quant)
    else:
        _lin = globals()["_lin"]
    d, nq, nkv = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
    t = h.shape[0]
    h = h.to(F32)
    residual = None
    nl = cfg.num_hidden_layers if num_layers is None else num_layers
    pos1d = positions if positions.dim() == 1 else positions
    for i in range(nl):
        p = f"language_model.model.layers.{i}."
        if residual is None:
            residual = h
            x = R.rms_norm(h, weights[p + "input_layernorm.weight"], cfg.rms_norm_eps, rt)
        else:
            x, residual = R.fused_add_rms_norm(h, residual, weights[p + "input_layernorm.weight"], cfg.rms_norm_eps, rt)
        qkv = _lin(x, weights[p + "self_attn.qkv_proj.weight"], weights.get(p + "self_attn.qkv_proj.bias"), rt)
        q, k, v = qkv.split([nq * d, nkv * d, nkv * d], dim=-1)
        if cfg.qk_norm:
            q = R.head_rms_norm(q, weights[p + "self_attn.q_norm.weight"], d, cfg.rms_norm_eps, rt)
            k = R.head_rms_norm(k, weights[p + "self_attn.k_norm.weight"], d, cfg.rms_norm_eps, rt)
        q = R.rope_neox(pos1d, q, d, cos_sin_cache, cfg.mrope_section if positions.dim() == 2 else None, rt)
        k = R.rope_neox(pos1d, k, d, cos_sin_cache, cfg.mrope_section if positions.dim() == 2 else None, rt)
        kb, vb = kv_caches[i]
        R.reshape_and_cache(k.view(t, nkv, d), v.view(t, nkv, d), kb, vb, slot_map)
        a = R.paged_attention_prefill(q.view(t, nq, d), kb, vb, block_table, seq_lens, query_start_loc, d ** -0.5)
        a = R._rt(a.reshape(t, nq * d), rt)
        o = _lin(a, weights[p + "self_attn.o_proj.weight"], None, rt)
        x, residual = R.fused_add_rms_norm(o, residual, weights[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps, rt)
        if p + "mlp.gate.weight" in weights:   # Qwen3-Omni talker: sparse-MoE MLP (oracle/ref_moe.py)
            from . import ref_moe

            mw = {k[len(p + "mlp."):]: v for k, v in weights.items() if k.startswith(p + "mlp.")}
            if "shared_expert.gate_up_proj.weight" in mw:
                gu_w = mw.pop("shared_expert.gate_up_proj.weight")
                mw["shared_expert.gate_proj.weight"], mw["shared_expert.up_proj.weight"] = gu_w.chunk(2, 0)
            logits = R._rt(x @ mw["gate.weight"].to(F32).t(), rt)       # the device routes on 16-bit router logits
            pr = torch.softmax(logits, -1)
            rw, re = torch.topk(pr, cfg.num_experts_per_tok, -1)
            if cfg.norm_topk_prob:
                rw = rw / rw.sum(-1, keepdim=True)
            h = ref_moe.moe_block(x, mw, cfg.num_experts_per_tok, cfg.norm_topk_prob, rt, routing=(rw, re))
        else:
            gu = _lin(x, weights[p + "mlp.gate_up_proj.weight"], None, rt)
            h = _lin(R.silu_and_mul(gu, rt), weights[p + "mlp.down_proj.weight"], None, rt)
    hidden, _ = R.fused_add_rms_norm(h, residual, weights["language_model.model.norm.weight"], cfg.rms_norm_eps, rt)
    return hidden


def thinker_to_talker_prefill(cfg, weights: dict, spk_row: torch.Tensor, thinker_prompt_embeds: torch.Tensor,
                              thinker_result: torch.Tensor, round_to=None):
    """The talker's prompt as the reference's preprocess hook builds it on a prefill span
    (``thinker_to_talker_process`` / ``_thinker_to_talker_prefill``, qwen2_5_omni.py:681-764):

        rows = cat(thinker_prompt_embeds,                                     :735-741
                   emb_thinker(speaker) + emb_codec(tts_codec_pad),
                   thinker_result[0] + emb_codec(tts_codec_start))
        ids  = [tts_codec_mask] * (n - 2) + [tts_codec_pad, tts_codec_start]   :744-756  (n = rows)
        thinker_reply_part = thinker_result[1:]                                :716-717

    ``spk_row`` = the speaker token's row of the THINKER's embedding table (``_get_embed_text_spk_token``, :639-642).
    Independent statement of the same construction in the installed HF ``transformers`` (read as text):
    ``modeling_qwen2_5_omni.py:3859-3890`` (generate: inputs = thinker prompt states | text-BOS(speaker) embed | first
    reply row), ``:2271-2277`` (talker forward at prefill: ``inputs_embeds[:, -1] += emb(codec_bos)``,
    ``inputs_embeds[:, -2] += emb(codec_pad)``), ``:2292-2297`` (decode: ``emb(code) + thinker_reply_part[:, :1]``, queue
    advances by one).
    The sums are the reference's tensor adds in the model dtype: exact fp32 sum, one rounding (``round_to``).
    Returns (ids list, rows [P + 2, E] fp32 on the ``round_to`` grid, reply rows [K - 1, E])."""
    emb = weights["language_model.model.embed_tokens.weight"].to(F32)
    pad, bos = emb[cfg.tts_codec_pad_token_id], emb[cfg.tts_codec_start_token_id]
    row_spk = R._rt(spk_row.to(F32).reshape(-1) + pad, round_to)
    row_bos = R._rt(thinker_result[0].to(F32) + bos, round_to)
    e = emb.shape[1]
    rows = torch.cat([thinker_prompt_embeds.to(F32).reshape(-1, e), row_spk.view(1, e), row_bos.view(1, e)])
    n = rows.shape[0]
    ids = ([cfg.tts_codec_mask_token_id] * (n - 2) + [cfg.tts_codec_pad_token_id, cfg.tts_codec_start_token_id])[-n:]
    return ids, rows, thinker_result[1:]


def logits_and_greedy(cfg, weights: dict, hidden: torch.Tensor, suppress_start_id: Optional[int], round_to=None):
    logits = _lin(hidden, weights["language_model.lm_head.weight"], None, round_to)
    logits = R.bad_word_processor(logits, suppress_start_id, cfg.tts_codec_end_token_id, cfg.tts_codec_start_token_id)
    return logits, logits.argmax(-1)


def random_weights(cfg, seed: int = 0, dtype=torch.bfloat16, num_layers: Optional[int] = None) -> dict:
    """N(0, 0.02^2) linears, RMSNorm weight 1 + N(0, 0.01^2), small biases."""
    g = torch.Generator().manual_seed(seed)
    nl = cfg.num_hidden_layers if num_layers is None else num_layers

    def w(*shape, std=0.02, mean=0.0):
        return (torch.randn(*shape, generator=g) * std + mean).to(dtype)

    d, nq, nkv, h, i = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.hidden_size, cfg.intermediate_size
    out = {}
    if getattr(cfg, "input_projection", True):
        out = {"thinker_to_talker_proj.weight": w(h, cfg.embedding_size), "thinker_to_talker_proj.bias": w(h)}
    out.update({
        "language_model.lm_head.weight": w(cfg.vocab_size, h),
        "language_model.model.embed_tokens.weight": w(cfg.vocab_size, cfg.embedding_size),
        "language_model.model.norm.weight": w(h, std=0.01, mean=1.0),
    })
    for li in range(nl):
        p = f"language_model.model.layers.{li}."
        out[p + "self_attn.qkv_proj.weight"] = w((nq + 2 * nkv) * d, h)
        if cfg.attention_bias:
            out[p + "self_attn.qkv_proj.bias"] = w((nq + 2 * nkv) * d)
        out[p + "self_attn.o_proj.weight"] = w(h, nq * d)
        if cfg.qk_norm:
            out[p + "self_attn.q_norm.weight"] = w(d, std=0.01, mean=1.0)
            out[p + "self_attn.k_norm.weight"] = w(d, std=0.01, mean=1.0)
        if getattr(cfg, "num_experts", 0):
            e, mi, si = cfg.num_experts, cfg.moe_intermediate_size, cfg.shared_expert_intermediate_size
            out[p + "mlp.gate.weight"] = w(e, h, std=0.1)
            out[p + "mlp.experts.gate_up_proj"] = w(e, 2 * mi, h, std=0.05)
            out[p + "mlp.experts.down_proj"] = w(e, h, mi, std=0.05)
            if si:
                out[p + "mlp.shared_expert.gate_up_proj.weight"] = w(2 * si, h)
                out[p + "mlp.shared_expert.down_proj.weight"] = w(h, si)
                out[p + "mlp.shared_expert_gate.weight"] = w(1, h, std=0.1)
        else:
            out[p + "mlp.gate_up_proj.weight"] = w(2 * i, h)
            out[p + "mlp.down_proj.weight"] = w(h, i)
        out[p + "input_layernorm.weight"] = w(h, std=0.01, mean=1.0)
        out[p + "post_attention_layernorm.weight"] = w(h, std=0.01, mean=1.0)
    return out
