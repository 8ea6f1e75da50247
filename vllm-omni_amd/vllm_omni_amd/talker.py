"""Host-side mirror of the talker decoder stack (SURVEY §8a A10/A11 wiring; §8c "files a CPU
restatement must follow").

Layer order, residual flow, scale and rotary flags follow
vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:37-69 (MLP), :72-156 (attention),
:159-225 (decoder layer), :300-326 (model loop); the talker wrapper (projection, logits,
``bad_word_processor``) follows qwen2_5_omni_talker.py:72-81,123-174; the per-step input
fix-up follows qwen2_5_omni.py:766-806.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass, field
from typing import Iterable, Optional

import torch
import torch.nn.functional as F

from . import ops
from .distributed import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size
from .layers import (Attention, MergedColumnParallelLinear, QKVParallelLinear, RMSNorm, RowParallelLinear,
                     SiluAndMul, _init_weight, fused_gate_up_silu, get_forward_context, get_rope, shard_column, shard_merged_column,
                     shard_qkv, shard_row, tuned_linear)
from .distributed import tensor_model_parallel_all_gather

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


@dataclass
class TalkerConfig:
    """Shape parameters of a talker decoder.  ``qwen2_5_omni_talker.json`` holds the
    ``transformers 5.15.0`` class defaults of ``Qwen2_5OmniTalkerConfig`` captured offline (the
    released checkpoint's config.json could not be fetched)."""

    name: str = "qwen2_5_omni_talker"
    hidden_size: int = 3584
    embedding_size: int = 3584
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    head_dim: int = 128
    intermediate_size: int = 18944
    vocab_size: int = 8448
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    max_position_embeddings: int = 32768
    mrope_section: Optional[list[int]] = None
    qk_norm: bool = False  # Qwen3 family: per-head RMSNorm on q and k before RoPE
    attention_bias: bool = True
    tts_codec_start_token_id: Optional[int] = 8293
    tts_codec_end_token_id: Optional[int] = 8294
    tts_codec_pad_token_id: Optional[int] = 8292
    tts_codec_mask_token_id: Optional[int] = 8296
    # ids in the THINKER's vocabulary whose thinker-embedding rows the talker's prompt is built from
    # (``Qwen2_5OmniTalkerConfig`` defaults; qwen2_5_omni.py:563-637)
    tts_text_start_token_id: Optional[int] = 151860
    tts_text_end_token_id: Optional[int] = 151861
    tts_text_pad_token_id: Optional[int] = 151859
    # sparse-MoE MLP (Qwen3-Omni talker); num_experts == 0: dense SiLU MLP
    num_experts: int = 0
    num_experts_per_tok: int = 0
    moe_intermediate_size: int = 0
    norm_topk_prob: bool = False
    shared_expert_intermediate_size: int = 0
    # Qwen2.5-Omni: a Linear embedding_size -> hidden_size in front of the decoder (thinker_to_talker_proj,
    # qwen2_5_omni_talker.py:72-75).  Qwen3-Omni: inputs arrive in the talker's hidden space and go straight into the
    # decoder (qwen3_omni_moe_talker.py:286-300), embedding_size == hidden_size
    input_projection: bool = True

    @classmethod
    def from_json(cls, path_or_name: str) -> "TalkerConfig":
        path = path_or_name if os.path.exists(path_or_name) else os.path.join(CONFIG_DIR, path_or_name + ".json")
        with open(path) as f:
            d = json.load(f)
        known = {k: v for k, v in d.items() if k in cls.__dataclass_fields__}
        return cls(**known)

    def num_params(self) -> int:
        h, i, d = self.hidden_size, self.intermediate_size, self.head_dim
        qkv = h * (self.num_attention_heads + 2 * self.num_key_value_heads) * d
        per_layer = qkv + (self.num_attention_heads + 2 * self.num_key_value_heads) * d * self.attention_bias \
            + self.num_attention_heads * d * h + 3 * h * i + 2 * h
        return self.num_hidden_layers * per_layer + h + self.vocab_size * h \
            + self.vocab_size * self.embedding_size + self.embedding_size * h + h


def bad_word_mask_spec(vocab: int, suppress_start_id: Optional[int], end_id: Optional[int],
                       bos_id: Optional[int]) -> tuple[list[tuple[int, int]], int]:
    """Column ranges / single column that ``bad_word_processor`` sets to -1e9.

    Branch-for-branch restatement of
    vllm_omni/model_executor/models/qwen2_5_omni/qwen2_5_omni_talker.py:150-169 as data for the
    fused mask+argmax kernel (``ops.mask_logits_argmax``).
    """
    ranges: list[tuple[int, int]] = []
    if suppress_start_id and suppress_start_id < vocab:
        if end_id is None:
            raise ValueError("config must have tts_codec_end_token_id attribute")
        if suppress_start_id == end_id:
            ranges.append((end_id + 1, vocab))
        elif suppress_start_id < end_id:
            ranges.append((suppress_start_id, end_id))
            ranges.append((end_id + 1, vocab))
        else:
            ranges.append((suppress_start_id, vocab))
    return ranges, (-1 if bos_id is None else int(bos_id))


class Qwen2MLP:
    """gate_up (merged column-parallel) -> SiluAndMul -> down (row-parallel + all-reduce)."""

    def __init__(self, cfg: TalkerConfig, dtype, device, generator=None):
        self.gate_up_proj = MergedColumnParallelLinear(cfg.hidden_size, [cfg.intermediate_size] * 2, bias=False,
                                                       dtype=dtype, device=device, generator=generator)
        self.down_proj = RowParallelLinear(cfg.intermediate_size, cfg.hidden_size, bias=False, dtype=dtype,
                                           device=device, generator=generator)
        self.down_proj._is_down_proj = True   # VLLM_OMNI_AMD_LINEAR=mlp (policy experiments): both MLP GEMMs on the kernel
        self.act_fn = SiluAndMul()

    def __call__(self, x: torch.Tensor, defer_reduce: bool = False):
        """``defer_reduce``: the caller feeds the result to a fused-add RMSNorm, which may then also do down_proj's
        split-K reduce (the result is an ``ops.SplitKPartials`` in that case)."""
        act = fused_gate_up_silu(self.gate_up_proj, x)   # one launch where the streaming kernel wins (layers.py)
        if act is None:
            gate_up, _ = self.gate_up_proj(x)
            act = self.act_fn(gate_up)
        x, _ = self.down_proj(act, defer_reduce)
        return x


class Qwen2Attention:
    def __init__(self, cfg: TalkerConfig, dtype, device, prefix: str, generator=None):
        self.total_num_heads = cfg.num_attention_heads
        self.total_num_kv_heads = cfg.num_key_value_heads
        self.head_dim = cfg.head_dim
        self.qkv_proj = QKVParallelLinear(cfg.hidden_size, self.head_dim, self.total_num_heads,
                                          self.total_num_kv_heads, bias=cfg.attention_bias, dtype=dtype, device=device,
                                          generator=generator)
        self.num_heads = self.qkv_proj.num_heads
        self.num_kv_heads = self.qkv_proj.num_kv_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.o_proj = RowParallelLinear(self.total_num_heads * self.head_dim, cfg.hidden_size, bias=False,
                                        dtype=dtype, device=device, generator=generator)
        self.o_proj._is_o_proj = True   # VLLM_OMNI_AMD_LINEAR=mlp+o (policy experiments)
        rp = {"base": cfg.rope_theta}
        if cfg.mrope_section:
            rp["mrope_section"] = cfg.mrope_section
        self.rotary_pos_emb = get_rope(head_size=self.head_dim, max_position=cfg.max_position_embeddings,
                                       is_neox_style=True, rope_parameters=rp, device=device)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling, num_kv_heads=self.num_kv_heads,
                              prefix=f"{prefix}.attn")
        self.q_norm = self.k_norm = None
        self.norm_eps = cfg.rms_norm_eps
        if cfg.qk_norm:
            self.q_norm = RMSNorm(self.head_dim, cfg.rms_norm_eps, dtype, device)
            self.k_norm = RMSNorm(self.head_dim, cfg.rms_norm_eps, dtype, device)

    def __call__(self, positions: torch.Tensor, hidden_states: torch.Tensor, defer_reduce: bool = False):
        has_cache = get_forward_context().attn_metadata is not None
        # split-K partials go into the RoPE + KV-write launch — unless that launch itself disappears into the attention
        # kernel (pure decode steps): then qkv_proj writes its 16-bit row
        fused = has_cache and self.attn.rope_decode_fusable(self.rotary_pos_emb, positions,
                                                             self.q_norm.weight if self.q_norm else None)
        qkv, _ = self.qkv_proj(hidden_states, "rope" if has_cache and not fused else False)
        if isinstance(qkv, ops.SplitKPartials):   # split-K reduce + bias inside the RoPE + KV-write launch (layers.py)
            attn_output = self.attn.forward_fused_rope_partials(
                self.rotary_pos_emb, positions, qkv,
                q_norm_weight=self.q_norm.weight if self.q_norm else None,
                k_norm_weight=self.k_norm.weight if self.k_norm else None, norm_eps=self.norm_eps)
            output, _ = self.o_proj(attn_output, defer_reduce)
            return output
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        # rotary_pos_emb(positions, q, k) + attn(q, k, v) of qwen2_old.py:153-154, fused:
        # (q/k-norm ->) RoPE -> KV write in one launch, then paged attention.
        attn_output = self.attn.forward_fused_rope(
            self.rotary_pos_emb, positions, q, k, v,
            q_norm_weight=self.q_norm.weight if self.q_norm else None,
            k_norm_weight=self.k_norm.weight if self.k_norm else None, norm_eps=self.norm_eps)
        output, _ = self.o_proj(attn_output, defer_reduce)   # may be split-K partials for the norm behind (layers.py)
        return output


class Qwen2DecoderLayer:
    def __init__(self, cfg: TalkerConfig, dtype, device, prefix: str, generator=None):
        self.self_attn = Qwen2Attention(cfg, dtype, device, f"{prefix}.self_attn", generator)
        if cfg.num_experts:
            # Qwen3-Omni talker: every layer's MLP is the sparse-MoE block (decoder_sparse_step 1,
            # qwen3_omni_moe_talker.py:476-513); experts are not sharded over TP ranks here
            from .moe import MoEConfig, Qwen3OmniMoeSparseMoeBlock

            if get_tensor_model_parallel_world_size() > 1:
                raise NotImplementedError("the MoE talker runs data-parallel only (tp_size 1)")
            self.mlp = Qwen3OmniMoeSparseMoeBlock(
                MoEConfig(cfg.hidden_size, cfg.moe_intermediate_size, cfg.num_experts, cfg.num_experts_per_tok,
                          cfg.norm_topk_prob, cfg.shared_expert_intermediate_size or None), dtype, device, generator)
        else:
            self.mlp = Qwen2MLP(cfg, dtype, device, generator)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype, device)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype, device)

    def __call__(self, positions, hidden_states, residual, defer_reduce: bool = False):
        """``defer_reduce``: the caller promises to hand the returned hidden states to a fused-add RMSNorm (the next
        layer's input_layernorm or the final norm); the dense MLP may then return its down_proj as split-K partials
        for that norm to reduce (``ops.SplitKPartials``) — one launch and one round trip of [T, H] less per layer."""
        # W8A8 linears behind a norm: the norm hands them the quantised activation directly (ops.rms_norm_int8)
        q_attn = getattr(self.self_attn.qkv_proj, "qweight", None) is not None
        q_mlp = isinstance(self.mlp, Qwen2MLP) and getattr(self.mlp.gate_up_proj, "qweight", None) is not None
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states, quant=q_attn)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual, quant=q_attn)
        # o_proj always meets the fused-add post_attention_layernorm: its split-K reduce (if any) may go there
        hidden_states = self.self_attn(positions=positions, hidden_states=hidden_states, defer_reduce=True)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual, quant=q_mlp)
        if defer_reduce and isinstance(self.mlp, Qwen2MLP):
            hidden_states = self.mlp(hidden_states, True)
        else:
            hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class Qwen2Model:
    def __init__(self, cfg: TalkerConfig, dtype, device, prefix: str = "model", generator=None):
        self.config = cfg
        # embedding table: replicated on every TP rank (60 MB at the talker shape)
        self.embed_tokens = _init_weight(cfg.vocab_size, cfg.embedding_size, dtype, device, generator)
        self.layers = [Qwen2DecoderLayer(cfg, dtype, device, f"{prefix}.layers.{i}", generator)
                       for i in range(cfg.num_hidden_layers)]
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype, device)

    def embed_input_ids(self, input_ids: torch.Tensor) -> torch.Tensor:
        return F.embedding(input_ids, self.embed_tokens)

    def __call__(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None) -> torch.Tensor:
        hidden_states = inputs_embeds if inputs_embeds is not None else self.embed_input_ids(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual, True)   # every output meets a fused norm
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class Qwen2ForCausalLM:
    """``Qwen2ForCausalLM_old``: model + vocab-parallel lm_head (qwen2_old.py:380-449)."""

    def __init__(self, cfg: TalkerConfig, dtype, device, generator=None):
        self.config = cfg
        self.model = Qwen2Model(cfg, dtype, device, generator=generator)
        tp = get_tensor_model_parallel_world_size()
        if cfg.vocab_size % tp:
            raise ValueError("vocab_size not divisible by tp_size")
        self.lm_head = _init_weight(cfg.vocab_size // tp, cfg.hidden_size, dtype, device, generator)

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        logits = tuned_linear(hidden_states, self.lm_head)
        return tensor_model_parallel_all_gather(logits)


class Qwen2_5OmniTalkerForConditionalGeneration:
    """Talker = ``thinker_to_talker_proj`` (Linear embedding_size -> hidden) + Qwen2 decoder
    (qwen2_5_omni_talker.py:72-81); ``forward`` consumes ``inputs_embeds`` (:123-148)."""

    # The runner contract of the reference's model classes (gpu_model_runner.py:1264-1309,1045-1070;
    # gpu_ar_model_runner.py:633-650): `preprocess` builds a request's input rows per step, `postprocess` returns
    # per-request updates from its hidden rows, `sample` may replace the runner's sampler.  `batched_preprocess` is this
    # package's addition: the runner's VECTORISED step (device reply pool + one gather-add launch, hipGraph-able)
    # realises this class's `preprocess` for the whole batch; a subclass that changes `preprocess` sets it False and is
    # driven through the per-request hook loop instead (eager).
    has_preprocess = True
    has_postprocess = False
    prefer_model_sampler = False
    batched_preprocess = True

    def __init__(self, cfg: TalkerConfig, dtype=torch.bfloat16, device=None, seed: Optional[int] = 0):
        self.config = cfg
        self.dtype = dtype
        self.device = device
        gen = None
        if seed is not None:
            gen = torch.Generator(device=device)
            # every TP rank draws its own shard; tests that need identical full weights use load_weights()
            gen.manual_seed(seed + 1000 * get_tensor_model_parallel_rank())
        self.thinker_to_talker_proj_weight = _init_weight(cfg.hidden_size, cfg.embedding_size, dtype, device, gen)
        self.thinker_to_talker_proj_bias = torch.zeros(cfg.hidden_size, dtype=dtype, device=device)
        self.language_model = Qwen2ForCausalLM(cfg, dtype, device, gen)
        self.suppress_start_id: Optional[int] = None

    # ---- reference-named surface --------------------------------------------------
    def get_language_model(self):
        return self.language_model

    def embed_input_ids(self, input_ids: torch.Tensor) -> torch.Tensor:
        return self.language_model.model.embed_input_ids(input_ids)

    def set_suppress_start_id(self, start_id: int) -> None:
        self.suppress_start_id = start_id

    def forward(self, input_ids=None, positions=None, intermediate_tensors=None, inputs_embeds=None, **kwargs):
        assert input_ids is not None or inputs_embeds is not None, "input_ids or inputs_embeds must be provided"
        if inputs_embeds is None:
            inputs_embeds = self.embed_input_ids(input_ids)  # profile run
        if self.config.input_projection:
            inputs_embeds = tuned_linear(inputs_embeds, self.thinker_to_talker_proj_weight, self.thinker_to_talker_proj_bias)
        return self.language_model.model(None, positions, intermediate_tensors, inputs_embeds=inputs_embeds)

    __call__ = forward

    def mask_spec(self, vocab: Optional[int] = None):
        return bad_word_mask_spec(vocab or self.config.vocab_size, self.suppress_start_id,
                                  self.config.tts_codec_end_token_id, self.config.tts_codec_start_token_id)

    def bad_word_processor(self, logits: torch.Tensor) -> torch.Tensor:
        ranges, col = self.mask_spec(logits.size(-1))
        ops.mask_logits_argmax(logits.view(-1, logits.size(-1)), ranges, col)
        return logits

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        logits = self.language_model.compute_logits(hidden_states)
        return self.bad_word_processor(logits)

    def compute_logits_and_greedy(self, hidden_states: torch.Tensor, ids: Optional[torch.Tensor] = None):
        """compute_logits + greedy argmax with the mask and argmax fused in one launch."""
        logits = self.language_model.compute_logits(hidden_states)
        ranges, col = self.mask_spec(logits.size(-1))
        ids = ops.mask_logits_argmax(logits, ranges, col, ids)
        return logits, ids

    # ---- prompt construction (the model's ``preprocess`` hook on a prefill span) -------------------------------
    # speaker -> token id in the thinker's vocabulary (qwen2_5_omni.py:570-579)
    TTS_TEXT_SPK_TOKEN_IDS = {"m02": 151870, "Ethan": 151870, "f030": 151872, "Chelsie": 151872,
                              "prefix_caching": 151870}

    def set_thinker_special_embeddings(self, rows) -> None:
        """Rows of the THINKER's token-embedding table the talker's prompt needs: ``{token id: Tensor[embedding_size]}``
        for the speaker tokens and ``tts_text_start_token_id`` (``_init_special_tokens_embeddings``,
        qwen2_5_omni.py:563-637: the reference keeps the whole ``thinker_embedding`` on the talker stage for these few
        lookups; ``load_weights`` extracts them from ``thinker_embedding.weight``)."""
        self.thinker_special_embeddings = {int(k): v.detach().to(self.dtype).to(self.device).reshape(-1)
                                           for k, v in dict(rows).items()}

    def _embed_text_spk_token(self, voice_type: Optional[str]) -> torch.Tensor:
        rows = getattr(self, "thinker_special_embeddings", None)
        if not rows:
            raise RuntimeError("the talker's prompt needs rows of the thinker's embedding table: load a checkpoint with "
                               "thinker_embedding.weight or call set_thinker_special_embeddings()")
        tok = self.TTS_TEXT_SPK_TOKEN_IDS.get(voice_type or "")
        if tok is None or tok not in rows:            # unknown voice: the text BOS row (qwen2_5_omni.py:639-642)
            tok = self.config.tts_text_start_token_id
        return rows[int(tok)]

    def thinker_to_talker_process(self, info: dict, voice_type: Optional[str] = "Chelsie"):
        """What the reference's ``preprocess`` hook builds for a talker request on its prefill span
        (``thinker_to_talker_process`` / ``_thinker_to_talker_prefill``, qwen2_5_omni.py:681-764) from the request's
        ``additional_information`` — ``prompt_embeds`` [P, E] (the thinker's hidden states over ITS prompt),
        ``thinker_result`` [K, E] (over its reply), ``prompt_token_ids``, ``thinker_output_token_ids``:

            prompt rows   = cat(prompt_embeds, emb_thinker(speaker) + emb_codec(pad), thinker_result[0] + emb_codec(bos))
            prompt ids    = [mask] * P + [pad, bos]                       (the codec-side ids of those rows)
            reply rows    = thinker_result[1:]                            (one per decode step, update_dict)

        Returns ``(ids list, rows Tensor[P + 2, E], {"thinker_reply_part": rows})``; the runner consumes the rows as
        the request's prompt embeddings and the reply rows through its device pool."""
        cfg = self.config
        e = cfg.embedding_size
        pe = info.get("prompt_embeds")
        tr = info.get("thinker_result")
        zero = torch.zeros(0, e, dtype=self.dtype, device=self.device)
        pe = pe.to(self.dtype).to(self.device) if isinstance(pe, torch.Tensor) else zero
        tr = tr.to(self.dtype).to(self.device) if isinstance(tr, torch.Tensor) else zero
        if tr.shape[0] == 0:
            raise ValueError("talker request without thinker_result rows: nothing to start the codec stream from")
        emb = self.language_model.model.embed_tokens
        codec = lambda tok: emb[int(tok)].to(self.dtype)   # noqa: E731
        spk = self._embed_text_spk_token(voice_type)
        # sums in the reference's dtype arithmetic: bf16 + bf16 -> bf16 (one rounding of the exact fp32 sum)
        row_spk = (spk.float() + codec(cfg.tts_codec_pad_token_id).float()).to(self.dtype)
        row_bos = (tr[0].float() + codec(cfg.tts_codec_start_token_id).float()).to(self.dtype)
        rows = torch.cat([pe.reshape(-1, e), row_spk.view(1, e), row_bos.view(1, e)], dim=0)
        n = rows.shape[0]
        ids = ([cfg.tts_codec_mask_token_id] * (n - 2) + [cfg.tts_codec_pad_token_id, cfg.tts_codec_start_token_id])[-n:]
        return ids, rows, {"thinker_reply_part": tr[1:].contiguous()}

    def preprocess(self, input_ids: torch.Tensor, input_embeds: Optional[torch.Tensor], **info_dict):
        """The per-request hook as the reference's runner calls it (``talker_preprocess``, qwen2_5_omni.py:653-679) on one
        request's span of the step: a span longer than one token is the prompt — rebuilt from the thinker hand-over in
        ``info_dict`` when there is one (the update carries the reply queue) — and a one-token span is a decode step:
        the head of the request's ``thinker_reply_part`` queue is added to the token's embedding and the update carries
        the rest of the queue (``thinker_to_talker_decode_one_step``, :766-796).
        Returns ``(input_ids, input_embeds, update_dict)``."""
        if input_embeds is None and input_ids is not None:
            input_embeds = self.embed_input_ids(input_ids)
        if input_ids.shape[0] > 1:
            if isinstance(info_dict.get("thinker_result"), torch.Tensor):
                ids, rows, upd = self.thinker_to_talker_process(info_dict, info_dict.get("voice_type", "Chelsie"))
                return torch.as_tensor(ids, dtype=torch.int64, device=rows.device), rows, upd
            return input_ids, input_embeds, {}
        upd = {}
        q = info_dict.get("thinker_reply_part")
        if isinstance(q, torch.Tensor) and q.numel() > 0:
            upd["thinker_reply_part"] = q[1:]
            input_embeds = self.decode_one_step_embeds(input_ids[0:1], q[0:1].to(input_embeds.device))
        return input_ids[0:1], input_embeds[0:1], upd

    def decode_one_step_embeds(self, last_token_ids: torch.Tensor, step_vec: Optional[torch.Tensor]) -> torch.Tensor:
        """Per-step talker input: ``thinker_reply_part[step] + embed(last codec token)``
        (``_thinker_to_talker_decode_one_step``, qwen2_5_omni.py:796-804), for the whole batch."""
        e = self.embed_input_ids(last_token_ids)
        return e if step_vec is None else step_vec.to(e.dtype) + e

    # ---- weights ------------------------------------------------------------------
    def named_weights(self) -> Iterable[tuple[str, torch.Tensor]]:
        if self.config.input_projection:
            yield "thinker_to_talker_proj.weight", self.thinker_to_talker_proj_weight
            yield "thinker_to_talker_proj.bias", self.thinker_to_talker_proj_bias
        lm = self.language_model
        yield "language_model.lm_head.weight", lm.lm_head
        yield "language_model.model.embed_tokens.weight", lm.model.embed_tokens
        yield "language_model.model.norm.weight", lm.model.norm.weight
        for i, layer in enumerate(lm.model.layers):
            p = f"language_model.model.layers.{i}."
            a, m = layer.self_attn, layer.mlp
            yield p + "self_attn.qkv_proj.weight", a.qkv_proj.weight
            if a.qkv_proj.bias is not None:
                yield p + "self_attn.qkv_proj.bias", a.qkv_proj.bias
            yield p + "self_attn.o_proj.weight", a.o_proj.weight
            if a.q_norm is not None:
                yield p + "self_attn.q_norm.weight", a.q_norm.weight
                yield p + "self_attn.k_norm.weight", a.k_norm.weight
            if self.config.num_experts:
                yield p + "mlp.gate.weight", m.gate_weight
                yield p + "mlp.experts.gate_up_proj", m.gate_up_proj
                yield p + "mlp.experts.down_proj", m.down_proj
                if m.shared_gate_up is not None:
                    yield p + "mlp.shared_expert.gate_up_proj.weight", m.shared_gate_up
                    yield p + "mlp.shared_expert.down_proj.weight", m.shared_down
                    yield p + "mlp.shared_expert_gate.weight", m.shared_expert_gate
            else:
                yield p + "mlp.gate_up_proj.weight", m.gate_up_proj.weight
                yield p + "mlp.down_proj.weight", m.down_proj.weight
            yield p + "input_layernorm.weight", layer.input_layernorm.weight
            yield p + "post_attention_layernorm.weight", layer.post_attention_layernorm.weight

    # ---- W8A8 ------------------------------------------------------
    def linear_layers(self):
        """(weight name, layer) of the decoder linears the reference's int8 config quantises — every linear of the
        transformer blocks (DiffusionInt8Config applies to `LinearBase` layers, int8_config.py:136-164); the
        embedding, lm_head and the thinker→talker projection stay 16-bit."""
        for i, layer in enumerate(self.language_model.model.layers):
            p = f"language_model.model.layers.{i}."
            yield p + "self_attn.qkv_proj.weight", layer.self_attn.qkv_proj
            yield p + "self_attn.o_proj.weight", layer.self_attn.o_proj
            if not self.config.num_experts:      # expert weights stay 16-bit
                yield p + "mlp.gate_up_proj.weight", layer.mlp.gate_up_proj
                yield p + "mlp.down_proj.weight", layer.mlp.down_proj

    def quantize_int8(self) -> None:
        from .layers import quantize_int8_

        for _, lin in self.linear_layers():
            if getattr(lin, "qweight", None) is None:
                quantize_int8_(lin)
        self.quantization = "int8"

    # checkpoint prefix -> parameter prefix: ``hf_to_vllm_mapper`` of the reference's talker classes
    # (qwen2_5_omni_talker.py:46-56; Qwen3: qwen3_omni_moe_talker.py:83-98), first match wins
    hf_to_vllm_prefix = (
        ("talker.codec_head.", "language_model.lm_head."),
        ("talker.model.", "language_model.model."),
        ("talker.thinker_to_talker_proj.", "thinker_to_talker_proj."),
        ("talker.", ""),
    )
    skip_prefixes = ("thinker.", "token2wav.", "code2wav.", "code_predictor.", "talker.code_predictor.")
    # Qwen3-Omni talker checkpoints also hold the two resize MLPs that turn THINKER outputs into talker inputs
    # (``text_projection`` / ``hidden_projection``, qwen3_omni_moe_talker.py:93-95,119-120): the reference's talker
    # class owns them, but they run when the talker's prompt embeddings are built (thinker -> talker hand-over), before
    # the decode path this package implements, whose inputs arrive as ``prompt_embeds`` / ``thinker_reply_part``
    # already in the talker's hidden size.  They are skipped BY NAME and recorded in ``skipped_weights``.
    stage_input_prefixes = ("talker.text_projection.", "talker.hidden_projection.", "text_projection.",
                            "hidden_projection.")
    # (fused parameter, checkpoint shard, shard id): ``stacked_params_mapping`` of qwen2_old.py:328-336
    stacked_params_mapping = (
        ("qkv_proj", "q_proj", "q"), ("qkv_proj", "k_proj", "k"), ("qkv_proj", "v_proj", "v"),
        ("gate_up_proj", "gate_proj", 0), ("gate_up_proj", "up_proj", 1),
    )

    def _map_checkpoint_name(self, name: str) -> Optional[str]:
        if name.startswith(self.skip_prefixes) or "rotary_emb.inv_freq" in name:
            return None
        if name.startswith(self.stage_input_prefixes):
            if not hasattr(self, "skipped_weights"):
                self.skipped_weights = []
            self.skipped_weights.append(name)
            return None
        for old, new in self.hf_to_vllm_prefix:
            if name.startswith(old):
                name = new + name[len(old):]
                break
        if name.startswith("codec_head."):          # Qwen3 keeps it apart; here it IS the LM head
            name = "language_model.lm_head." + name[len("codec_head."):]
        return name

    def _shard_rows(self, fused_name: str, shard_id) -> tuple[int, int, int, int]:
        """(dst row offset, rows, src row offset, src rows) of checkpoint shard ``shard_id`` inside this rank's
        fused parameter ``fused_name`` — what upstream's per-parameter ``weight_loader(param, w, shard_id)`` does
        for QKVParallelLinear / MergedColumnParallelLinear (head partition + KV replication, qwen2_old.py:89-102)."""
        from .distributed import kv_head_for_rank, partition_heads

        cfg = self.config
        tp, r = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        d = cfg.head_dim
        if shard_id in ("q", "k", "v"):
            nq, nkv, _ = partition_heads(cfg.num_attention_heads, cfg.num_key_value_heads, tp)
            if shard_id == "q":
                return 0, nq * d, r * nq * d, nq * d
            kvr = kv_head_for_rank(cfg.num_key_value_heads, tp, r)
            dst = nq * d + (0 if shard_id == "k" else nkv * d)
            return dst, nkv * d, kvr.start * d, nkv * d
        # gate (0) / up (1) halves of a merged column-parallel projection
        if ".shared_expert." in fused_name:
            inter, tp_, r_ = cfg.shared_expert_intermediate_size, 1, 0       # MoE weights are replicated
        else:
            inter, tp_, r_ = cfg.intermediate_size, tp, r
        per = inter // tp_
        return int(shard_id) * per, per, r_ * per, per

    def load_weights(self, weights) -> set[str]:
        """``load_weights(Iterable[(checkpoint name, tensor)])`` as the reference's model classes take it
        (qwen2_5_omni_talker.py:183-211 -> qwen2_old.py:328-377): names go through the ``talker.`` prefix mapper,
        separate ``q_proj`` / ``k_proj`` / ``v_proj`` and ``gate_proj`` / ``up_proj`` tensors are written into this
        rank's slice of the fused ``qkv_proj`` / ``gate_up_proj`` parameters, every other tensor is TP-sharded by
        its role; ``thinker.`` / ``token2wav.`` weights and ``rotary_emb.inv_freq`` are skipped.  Per-expert MoE
        tensors (``mlp.experts.<e>.gate_proj|up_proj|down_proj.weight``) fill the stacked expert parameters.
        Tensors are consumed one at a time (a safetensors iterator never holds the whole checkpoint).  A dict of
        UNSHARDED tensors under this package's fused names (``named_weights``) is accepted as well.  On a
        quantised model the 16-bit shard of a touched linear is rebuilt, loaded and quantised again (online
        quantisation, int8_config.py:391-418).  Returns the parameter names that received data; a name that maps
        to no parameter raises ``KeyError`` (the reference's ``params_dict[name]`` does)."""
        import re

        cfg = self.config
        tp, r = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        items = weights.items() if isinstance(weights, dict) else weights
        params = {n: t for n, t in self.named_weights() if t is not None}
        quantised = {n: lin for n, lin in self.linear_layers() if getattr(lin, "qweight", None) is not None}
        requantise = False
        loaded: set[str] = set()
        expert_re = re.compile(r"^(.*\.mlp\.experts)\.(\d+)\.(gate_proj|up_proj|down_proj)\.weight$")

        def param(name: str) -> torch.Tensor:
            nonlocal requantise
            lin = quantised.get(name)
            if lin is not None and lin.qweight is not None:     # first touch: back to a 16-bit shard
                lin.weight = torch.zeros(lin.qweight.shape, dtype=self.dtype, device=lin.qweight.device)
                lin.qweight = lin.weight_scale = None
                params[name] = lin.weight
                requantise = True
            if name not in params:
                raise KeyError(f"checkpoint tensor maps to no parameter of the talker: {name!r}")
            return params[name]

        for raw_name, w in items:
            if raw_name in ("thinker_embedding.weight", "thinker.model.embed_tokens.weight") and w.dim() == 2 \
                    and w.shape[1] == cfg.embedding_size:
                # the few rows of the thinker's embedding table the talker's prompt is built from (speaker tokens, text
                # BOS): qwen2_5_omni.py:563-637 keeps the whole table on the talker stage for them
                want = set(self.TTS_TEXT_SPK_TOKEN_IDS.values()) | {cfg.tts_text_start_token_id}
                self.set_thinker_special_embeddings({t: w[t] for t in want if t is not None and t < w.shape[0]})
                continue
            name = self._map_checkpoint_name(raw_name)
            if name is None:
                continue
            if name in params or name in quantised:             # a full (unsharded) tensor under the parameter's own name
                dst = param(name)
                if ".mlp.experts." in name or ".mlp.shared_expert" in name or name.endswith(".mlp.gate.weight"):
                    pass   # MoE weights are replicated (tp_size 1)
                elif name.endswith("qkv_proj.weight") or name.endswith("qkv_proj.bias"):
                    w = shard_qkv(w, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, tp, r)
                elif name.endswith("gate_up_proj.weight"):
                    w = shard_merged_column(w, [cfg.intermediate_size] * 2, tp, r)
                elif name.endswith("o_proj.weight") or name.endswith("down_proj.weight"):
                    w = shard_row(w, tp, r)
                elif name.endswith("lm_head.weight"):
                    w = shard_column(w, tp, r)
                dst.copy_(w.to(dst.dtype))
                loaded.add(name)
                continue
            m = expert_re.match(name)
            if m:                                                # per-expert checkpoint tensors -> stacked parameters
                base, e, kind = m.group(1), int(m.group(2)), m.group(3)
                if kind == "down_proj":
                    dst = param(base + ".down_proj")
                    dst[e].copy_(w.to(dst.dtype))
                    loaded.add(base + ".down_proj")
                else:
                    dst = param(base + ".gate_up_proj")
                    mi = cfg.moe_intermediate_size
                    dst[e, (mi if kind == "up_proj" else 0):(2 * mi if kind == "up_proj" else mi)].copy_(w.to(dst.dtype))
                    loaded.add(base + ".gate_up_proj")
                continue
            for param_name, shard_name, shard_id in self.stacked_params_mapping:
                if shard_name not in name:
                    continue
                fused = name.replace(shard_name, param_name)
                dst = param(fused)
                d0, rows, s0, srows = self._shard_rows(fused, shard_id)
                dst[d0:d0 + rows].copy_(w[s0:s0 + srows].to(dst.dtype))
                loaded.add(fused)
                break
            else:
                raise KeyError(f"checkpoint tensor maps to no parameter of the talker: {raw_name!r} -> {name!r}")
        if requantise:
            self.quantize_int8()
        return loaded

    def weight_bytes(self) -> int:
        n = sum(w.numel() * w.element_size() for _, w in self.named_weights() if w is not None)
        for _, lin in self.linear_layers():
            if getattr(lin, "qweight", None) is not None:
                n += lin.qweight.numel() + lin.weight_scale.numel() * 4
        return n
