"""Qwen3-Omni talker code predictor on MI355X (the code-predictor half).

Mirrors ``CodePredictorWrapper.forward`` of vllm_omni/model_executor/models/common/qwen3_code_predictor.py:318-626
as ``Qwen3OmniMoeTalkerCodePredictor`` configures it (qwen3_omni_moe_code_predictor_mtp.py:11-31:
``return_proj_buf=True``, ``sampling_mode="stored"``): for every talker token, predict the residual codebooks
1 .. G-1 autoregressively with a 5-layer Qwen3-style transformer over the sequence
``[talker hidden, emb(code_0), emb(code_1), ...]``; returns ``(codes [B, G, 1], proj_buf [B, G+1, H])``.

MI355X-first difference: the reference RE-PREFILLS the whole (growing) sequence at each of the G-1 steps
("no KV cache ... the extra O(T^2) FLOPs are negligible") — that is 31 forward passes over 33 positions for
G = 32.  Attention is causal and earlier positions never change, so the hidden state at the newest position
is identical if the earlier keys and values are simply kept: the predictor here owns a small private paged KV
cache (one 64-token block per sequence and layer) and runs ONE 2-token prefill followed by G-2 single-token
decode steps through the same HIP kernels as the talker (fused q/k-norm + RoPE + KV write, paged attention,
RMSNorm, SiLU-gate) — 1/17 of the reference's token-forwards at G = 32.  Sampling is the device sampler
(top-k -> top-p -> draw, `ops.sample_tokens`), the embedding lookup is on the device too, so a whole call has
no host synchronisation.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Iterable, Optional

import torch
import torch.nn.functional as F

from . import ops
from .layers import AttentionMetadata, RMSNorm, _init_weight, set_forward_context, tuned_linear
from .talker import Qwen2DecoderLayer, TalkerConfig


@dataclass
class CodePredictorConfig:
    """``transformers`` 5.15.0 class defaults of ``Qwen3OmniMoeTalkerCodePredictorConfig``."""

    hidden_size: int = 1024
    num_hidden_layers: int = 5
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 3072
    vocab_size: int = 2048
    num_code_groups: int = 32
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    attention_bias: bool = False

    def layer_config(self) -> TalkerConfig:
        return TalkerConfig(name="code_predictor", hidden_size=self.hidden_size, embedding_size=self.hidden_size,
                            num_hidden_layers=self.num_hidden_layers, num_attention_heads=self.num_attention_heads,
                            num_key_value_heads=self.num_key_value_heads, head_dim=self.head_dim,
                            intermediate_size=self.intermediate_size, vocab_size=self.vocab_size,
                            rms_norm_eps=self.rms_norm_eps, rope_theta=self.rope_theta, max_position_embeddings=128,
                            mrope_section=None, qk_norm=True, attention_bias=self.attention_bias,
                            tts_codec_start_token_id=None, tts_codec_end_token_id=None, tts_codec_pad_token_id=None,
                            tts_codec_mask_token_id=None)


class MI355XCodePredictor:
    BLOCK = 64  # private KV block: holds the whole G + 1 <= 64 token sequence of one talker token

    def __init__(self, cfg: CodePredictorConfig, dtype=torch.bfloat16, device=None, max_batch: int = 256,
                 talker_hidden_size: Optional[int] = None, use_projection: bool = False, seed: Optional[int] = 0):
        if cfg.num_code_groups + 1 > self.BLOCK:
            raise ValueError("num_code_groups + 1 must fit one KV block")
        self.config, self.dtype, self.device, self.max_batch = cfg, dtype, device, max_batch
        self.num_code_groups = g = cfg.num_code_groups
        gen = None
        if seed is not None:
            gen = torch.Generator(device=device)
            gen.manual_seed(seed)
        lc = cfg.layer_config()
        emb_dim = int(talker_hidden_size) if talker_hidden_size is not None else cfg.hidden_size
        self.codec_embedding = [_init_weight(cfg.vocab_size, emb_dim, dtype, device, gen) for _ in range(g - 1)]
        self.layers = [Qwen2DecoderLayer(lc, dtype, device, f"code_predictor.layers.{i}", gen)
                       for i in range(cfg.num_hidden_layers)]
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype, device)
        self.lm_head = [_init_weight(cfg.vocab_size, cfg.hidden_size, dtype, device, gen) for _ in range(g - 1)]
        self.proj_weight = self.proj_bias = None       # small_to_mtp_projection: Identity unless sizes differ
        if use_projection and emb_dim != cfg.hidden_size:
            self.proj_weight = _init_weight(cfg.hidden_size, emb_dim, dtype, device, gen)
            self.proj_bias = torch.zeros(cfg.hidden_size, dtype=dtype, device=device)
        elif emb_dim != cfg.hidden_size:
            raise ValueError("talker_hidden_size != hidden_size needs use_projection=True")
        self._top_k, self._top_p = 50, 0.8             # "stored" sampling defaults (qwen3_code_predictor.py:375-377)

        # private paged KV cache, head-major blocks; block 0 is the null block, sequence i owns block i + 1
        nkv, d, bs, mb = cfg.num_key_value_heads, cfg.head_dim, self.BLOCK, max_batch
        self.kv = [torch.zeros(mb + 1, 2, nkv, bs, d, dtype=dtype, device=device) for _ in range(cfg.num_hidden_layers)]
        for layer, kv in zip(self.layers, self.kv):
            layer.self_attn.attn.bind_kv_cache((kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)))
        i32 = dict(dtype=torch.int32, device=device)
        self.block_table = torch.arange(1, mb + 1, **i32).view(mb, 1)
        blk = torch.arange(1, mb + 1, dtype=torch.int64, device=device) * bs
        # step 0 (prefill of positions 0 and 1, sequence-major) and single-token steps at position p
        self._slots0 = (blk.view(mb, 1) + torch.arange(2, dtype=torch.int64, device=device)).reshape(-1)
        self._pos0 = torch.arange(2, dtype=torch.int64, device=device).repeat(mb)
        self._qsl0 = torch.arange(0, 2 * mb + 1, 2, **i32)
        self._seq0 = torch.full((mb,), 2, **i32)
        self._slots = [blk + p for p in range(g + 1)]
        self._pos = [torch.full((mb,), p, dtype=torch.int64, device=device) for p in range(g + 1)]
        self._seq = [torch.full((mb,), p + 1, **i32) for p in range(g + 1)]
        self._qsl1 = torch.arange(0, mb + 1, **i32)
        self.proj_buf = torch.zeros(mb, g + 1, cfg.hidden_size, dtype=dtype, device=device)
        self.codes = torch.zeros(mb, g, dtype=torch.int64, device=device)
        self._temp = torch.ones(mb, dtype=torch.float32, device=device)
        self._topk = torch.zeros(mb, dtype=torch.int32, device=device)
        self._topp = torch.ones(mb, dtype=torch.float32, device=device)
        # static inputs of the captured call
        self._in_code = torch.zeros(mb, dtype=torch.int64, device=device)
        self._in_embed = torch.zeros(mb, emb_dim, dtype=dtype, device=device)
        self._in_hidden = torch.zeros(mb, emb_dim, dtype=dtype, device=device)
        self._in_seed = torch.zeros(mb, dtype=torch.int64, device=device)
        self._in_offset = torch.zeros(mb, dtype=torch.int64, device=device)
        self._step_in = torch.zeros(mb, cfg.hidden_size, dtype=dtype, device=device)   # the stack's input row of a step
        self._steps = torch.arange(1, g, dtype=torch.int64, device=device).view(g - 1, 1)
        self._graphs: dict = {}
        # The single-token passes as ONE persistent launch each (ops.CpStack, csrc/cp_stack.hip); built at first use.  Opt-in:
        # bit-identical to the chain of launches with the same tile plans, but not faster — measured per call 8.09 ms against
        # 7.43 at batch 32, 14.07 against 12.55 at 512: every phase is still a chain of device-scope round trips (slabs in,
        # rows out, ~1-2 us each), which is what a kernel boundary costs too (profiles/r04za_cp_stack_phase_timing.log).
        self._one_launch = os.environ.get("VLLM_OMNI_AMD_CP_ONE_LAUNCH", "0") == "1"
        self._cp_stack: Optional[ops.CpStack] = None
        self._hid = torch.zeros(mb, cfg.hidden_size, dtype=dtype, device=device)

    # ---- reference-named surface ------------------------------------------------------------------------
    def get_input_embeddings(self) -> list[torch.Tensor]:
        return self.codec_embedding

    def set_sampling_params(self, top_k: int = 50, top_p: float = 0.8) -> None:
        self._top_k, self._top_p = top_k, top_p

    def _project(self, x: torch.Tensor) -> torch.Tensor:
        return x if self.proj_weight is None else F.linear(x, self.proj_weight, self.proj_bias)

    def _one_launch_stack(self) -> Optional["ops.CpStack"]:
        """The layers' weights and private caches behind one persistent launch per single-token pass — when the stack is
        what that kernel was built for: 16-bit un-quantised dense layers, q/k-norm, heads of 128 with two q heads per KV
        head, hidden <= 1024, one rank.  Otherwise (and with VLLM_OMNI_AMD_CP_ONE_LAUNCH=0) the per-layer launches run."""
        if self._cp_stack is not None or not self._one_launch or torch.device(self.device).type != "cuda":
            return self._cp_stack
        self._one_launch = False        # decided once
        cfg = self.config
        a0 = self.layers[0].self_attn
        if (self.dtype not in (torch.bfloat16, torch.float16) or cfg.head_dim != 128 or cfg.hidden_size > 1024
                or cfg.hidden_size % 64 or cfg.intermediate_size % 64 or a0.q_norm is None
                or a0.num_heads != cfg.num_attention_heads or a0.num_heads != 2 * a0.num_kv_heads
                or cfg.num_hidden_layers > 8 or self.num_code_groups > 32):
            return None
        layers = []
        for layer, kv in zip(self.layers, self.kv):
            at, mlp = layer.self_attn, layer.mlp
            lins = (at.qkv_proj, at.o_proj, getattr(mlp, "gate_up_proj", None), getattr(mlp, "down_proj", None))
            if any(x is None or x.weight is None or getattr(x, "qweight", None) is not None or x.bias is not None for x in lins):
                return None
            layers.append(dict(qkv=at.qkv_proj.weight, o=at.o_proj.weight, gate_up=mlp.gate_up_proj.weight,
                               down=mlp.down_proj.weight, input_norm=layer.input_layernorm.weight,
                               post_norm=layer.post_attention_layernorm.weight, q_norm=at.q_norm.weight,
                               k_norm=at.k_norm.weight, k_cache=kv[:, 0].permute(0, 2, 1, 3), v_cache=kv[:, 1].permute(0, 2, 1, 3)))
        self._cp_stack = ops.CpStack(layers, self.norm.weight, a0.rotary_pos_emb.cos_sin_cache, a0.num_heads, a0.num_kv_heads,
                                     cfg.head_dim, cfg.rms_norm_eps, a0.scaling, self.max_batch)
        return self._cp_stack

    def _stack(self, x: torch.Tensor, positions: torch.Tensor, md: AttentionMetadata) -> torch.Tensor:
        residual = None
        h = x
        with set_forward_context(md):
            for layer in self.layers:
                h, residual = layer(positions, h, residual, True)   # every output meets a fused-add norm: down_proj may defer its reduce
        h, _ = self.norm(h, residual)
        return h

    @torch.inference_mode()
    def forward(self, layer0_code: torch.Tensor, layer0_embed: torch.Tensor, last_talker_hidden: torch.Tensor,
                seed: Optional[torch.Tensor] = None, offset: Optional[torch.Tensor] = None, do_sample: bool = True,
                return_logits: bool = False):
        """layer0_code [B, 1] | [B]; layer0_embed [B, 1, E] | [B, E]; last_talker_hidden [B, E].
        ``seed`` / ``offset`` (int64 [B]): Philox key and draw base of each row's G-1 draws (draw s of a row uses
        counter ``offset * G + s``); default: seed 0, offset 0.  ``do_sample=False``: greedy.
        Returns (codes [B, G, 1] int64, proj_buf [B, G+1, H]) — and the per-step logits when asked (tests).
        After ``capture(B)`` a sampling call at that batch size replays ONE hipGraph of the whole call (31 stack
        passes, draws and embedding look-ups at G = 32); its outputs are then static buffers, overwritten by the
        next call."""
        b = int(layer0_code.shape[0])
        if b > self.max_batch:
            raise ValueError(f"batch {b} exceeds max_batch {self.max_batch}")
        entry = self._graphs.get(b) if (do_sample and not return_logits) else None
        if entry is not None:
            g_, outs = entry
            dev = self.device
            self._in_code[:b].copy_(layer0_code.reshape(b))
            self._in_embed[:b].copy_(layer0_embed.reshape(b, -1))
            self._in_hidden[:b].copy_(last_talker_hidden.reshape(b, -1))
            self._in_seed[:b].copy_(seed) if seed is not None else self._in_seed[:b].zero_()
            self._in_offset[:b].copy_(offset) if offset is not None else self._in_offset[:b].zero_()
            g_.replay()
            return outs
        return self._run(b, layer0_code, layer0_embed, last_talker_hidden, seed, offset, do_sample, return_logits)

    @torch.inference_mode()
    def capture(self, b: int) -> None:
        """Capture the whole call for batch size ``b`` (device-only work: no host synchronisation inside).  Always under
        ``inference_mode``, like every capture the runner makes: torch creates the generator's graph-state tensors at the
        first capture of the process, and a capture outside inference mode after one inside it fails on them ("inplace
        update to inference tensor")."""
        if torch.device(self.device).type != "cuda" or b in self._graphs:
            return
        args = (self._in_code[:b], self._in_embed[:b], self._in_hidden[:b], self._in_seed[:b], self._in_offset[:b])
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._run(b, *args, True, False)      # warm-up: library handles, workspaces
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        from .model_runner import no_gc_during_capture

        graph = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(graph):
            outs = self._run(b, *args, True, False)
        self._graphs[b] = (graph, outs)

    def _run(self, b, layer0_code, layer0_embed, last_talker_hidden, seed, offset, do_sample, return_logits):
        g, h = self.num_code_groups, self.config.hidden_size
        dev = self.device
        buf, codes = self.proj_buf, self.codes
        buf[:b, 0] = self._project(last_talker_hidden.reshape(b, -1).to(self.dtype))
        buf[:b, 1] = self._project(layer0_embed.reshape(b, -1).to(self.dtype))
        codes[:b, 0] = layer0_code.reshape(b)
        if seed is None:
            seed = torch.zeros(b, dtype=torch.int64, device=dev)
        base = (torch.zeros(b, dtype=torch.int64, device=dev) if offset is None else offset.to(torch.int64)) * g
        temp = self._temp[:b] if do_sample else torch.zeros(b, dtype=torch.float32, device=dev)
        topk = self._topk[:b].fill_(int(self._top_k))
        topp = self._topp[:b].fill_(float(self._top_p))
        # positions 0 and 1 of every sequence in one varlen prefill; the hidden state at position 1 predicts code 1
        md = AttentionMetadata(num_actual_tokens=2 * b, max_query_len=2, query_start_loc=self._qsl0[:b + 1],
                               seq_lens=self._seq0[:b], max_seq_len=2, block_table=self.block_table[:b],
                               slot_mapping=self._slots0[:2 * b], num_reqs=b, num_decodes=0)
        hid = self._stack(buf[:b, :2].reshape(2 * b, h), self._pos0[:2 * b], md)[1::2]
        all_logits = []
        draws = base.view(1, b) + self._steps          # [G - 1, B] Philox counters of the call's draws: one launch, not G - 1
        fused = self.proj_weight is None and torch.device(dev).type == "cuda"
        for step in range(1, g):
            logits = tuned_linear(hid, self.lm_head[step - 1])
            if return_logits:
                all_logits.append(logits.float())
            # draw, store the code, embed it -> next buffer position (the last one is only returned, never forwarded) and the
            # next step's input row (the stack works in place on its input): ONE launch (ops.sample_tokens_embed)
            x = self._step_in[:b]
            if fused:
                ops.sample_tokens_embed(logits, temp, seed, draws[step - 1], topk, topp, self.codec_embedding[step - 1],
                                        buf[:b, step + 1], x if step < g - 1 else None, codes[:b, step])
            else:
                code = ops.sample_tokens(logits, temp, seed, draws[step - 1], top_k=topk, top_p=topp)
                codes[:b, step] = code
                buf[:b, step + 1] = self._project(F.embedding(code, self.codec_embedding[step - 1]))
                x = buf[:b, step + 1].clone()
            if step < g - 1:
                p = step + 1
                md = AttentionMetadata(num_actual_tokens=b, max_query_len=1, query_start_loc=self._qsl1[:b + 1],
                                       seq_lens=self._seq[p][:b], max_seq_len=p + 1, block_table=self.block_table[:b],
                                       slot_mapping=self._slots[p][:b], num_reqs=b, num_decodes=b, single_block=True,
                                       uniform_position=p)
                one = self._one_launch_stack()
                if one is not None:   # the five layers + final norm of this pass in one launch
                    hid = one.decode(self._hid[:b], x, self.block_table[:b, 0], p)
                else:
                    hid = self._stack(x, self._pos[p][:b], md)
        out = (codes[:b].unsqueeze(-1).clone(), buf[:b].clone())
        return out + (all_logits,) if return_logits else out

    __call__ = forward

    # ---- weights --------------------------------------------------------------------------------------------
    def named_weights(self) -> Iterable[tuple[str, torch.Tensor]]:
        """Names of the reference module tree (unfused q/k/v and gate/up are handled by ``load_weights``)."""
        for i, w in enumerate(self.codec_embedding):
            yield f"model.codec_embedding.{i}.weight", w
        for i, w in enumerate(self.lm_head):
            yield f"lm_head.{i}.weight", w
        yield "model.norm.weight", self.norm.weight
        for i, layer in enumerate(self.layers):
            p = f"model.layers.{i}."
            yield p + "self_attn.qkv_proj.weight", layer.self_attn.qkv_proj.weight
            yield p + "self_attn.o_proj.weight", layer.self_attn.o_proj.weight
            yield p + "self_attn.q_norm.weight", layer.self_attn.q_norm.weight
            yield p + "self_attn.k_norm.weight", layer.self_attn.k_norm.weight
            yield p + "mlp.gate_up_proj.weight", layer.mlp.gate_up_proj.weight
            yield p + "mlp.down_proj.weight", layer.mlp.down_proj.weight
            yield p + "input_layernorm.weight", layer.input_layernorm.weight
            yield p + "post_attention_layernorm.weight", layer.post_attention_layernorm.weight
        if self.proj_weight is not None:
            yield "small_to_mtp_projection.weight", self.proj_weight
            yield "small_to_mtp_projection.bias", self.proj_bias

    def load_weights(self, weights: dict[str, torch.Tensor]) -> set[str]:
        """Reference checkpoint names (separate q_proj / k_proj / v_proj and gate_proj / up_proj,
        qwen3_code_predictor.py:121-141,184-189) are fused into this stack's qkv_proj / gate_up_proj."""
        full = dict(weights)
        loaded: set[str] = set()
        for i in range(self.config.num_hidden_layers):
            p = f"model.layers.{i}."
            qkv = [p + f"self_attn.{n}_proj.weight" for n in "qkv"]
            if all(n in full for n in qkv):
                full[p + "self_attn.qkv_proj.weight"] = torch.cat([full[n] for n in qkv], dim=0)
                loaded.update(qkv)
            gu = [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"]
            if all(n in full for n in gu):
                full[p + "mlp.gate_up_proj.weight"] = torch.cat([full[n] for n in gu], dim=0)
                loaded.update(gu)
        for name, dst in self.named_weights():
            if name in full:
                dst.copy_(full[name].to(dst.dtype))
                if name in weights:
                    loaded.add(name)
        return loaded
