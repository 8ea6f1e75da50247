"""Host-side mirror of the upstream layers the talker is wired from.

Same names, argument meaning and call order as the classes imported at
vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:10-17 (``RMSNorm``, ``SiluAndMul``,
``get_rope``, ``Attention``, ``QKVParallelLinear``, ``MergedColumnParallelLinear``,
``RowParallelLinear``).  Every forward is a call into the HIP kernels (``ops``); the dense GEMMs are
library GEMMs (hipBLASLt through ``torch.nn.functional.linear``; SURVEY A10 "not a named kernel") except where
the decode-batch streaming kernel is faster (``_apply_linear``).  Inference only; parameters are plain tensors.
"""
from __future__ import annotations

import math
import os
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import ops
from .distributed import (get_parallel_state, get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                          tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce,
                          tensor_model_parallel_all_reduce_partials)
from .kv_cache import normalize_layer_kv


# ------------------------------------------------------------------ forward context
@dataclass
class AttentionMetadata:
    """What the attention kernels need for one step (built by the runner each step).

    Mirrors the fields upstream's metadata builders produce and the reference forwards through
    ``set_forward_context(attn_metadata, ..., slot_mapping=...)``
    (vllm_omni/worker/gpu_ar_model_runner.py:493-510).  Requests are ordered decodes-first.
    """

    num_actual_tokens: int
    max_query_len: int
    query_start_loc: torch.Tensor  # int32 [num_reqs + 1], device
    seq_lens: torch.Tensor         # int32 [num_reqs], device (context INCLUDING this step's tokens)
    max_seq_len: int               # host upper bound of seq_lens (sizes the decode launch)
    block_table: torch.Tensor      # int32 [num_reqs, max_blocks], device
    slot_mapping: torch.Tensor     # int64 [num_tokens_padded], device, -1 = padding
    num_reqs: int = 0
    num_decodes: int = 0           # leading requests with query_len == 1
    decode_partition_size: int = 0
    decode_workspace: Optional[torch.Tensor] = None
    # every sequence's whole context lives in block_table[:, 0] (<= 64 tokens) and the step is one token per sequence
    # at `positions`: RoPE + KV append + attention run as ONE launch (ops.rope_cache_attend_block; the code predictor)
    single_block: bool = False
    uniform_position: Optional[int] = None   # with single_block: every sequence's new token sits at this position


@dataclass
class ForwardContext:
    attn_metadata: Optional[AttentionMetadata] = None
    slot_mapping: Optional[torch.Tensor] = None


_FORWARD_CONTEXT: Optional[ForwardContext] = None


def get_forward_context() -> ForwardContext:
    if _FORWARD_CONTEXT is None:
        raise RuntimeError("forward context is not set; wrap the model call in set_forward_context()")
    return _FORWARD_CONTEXT


@contextmanager
def set_forward_context(attn_metadata: Optional[AttentionMetadata], slot_mapping: Optional[torch.Tensor] = None):
    global _FORWARD_CONTEXT
    prev = _FORWARD_CONTEXT
    _FORWARD_CONTEXT = ForwardContext(attn_metadata, slot_mapping if slot_mapping is not None
                                      else (attn_metadata.slot_mapping if attn_metadata else None))
    try:
        yield _FORWARD_CONTEXT
    finally:
        _FORWARD_CONTEXT = prev


# --------------------------------------------------------------------------- norms
class RMSNorm:
    """``RMSNorm(hidden, eps)``: ``norm(x)`` and ``norm(x, residual) -> (normed, new_residual)``
    (qwen2_old.py:202-203,212-216,223).  The fused form works IN PLACE on both arguments."""

    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype=torch.bfloat16, device=None):
        self.weight = torch.ones(hidden_size, dtype=dtype, device=device)
        self.variance_epsilon = eps

    def __call__(self, x, residual: Optional[torch.Tensor] = None, quant: bool = False):
        """``quant``: the caller feeds the result to a W8A8 linear — the normalised row leaves as int8 + scale
        (``ops.QuantizedActs``), the activation quantisation fused into this launch."""
        if isinstance(x, ops.Int8SplitKPartials):   # a W8A8 row-parallel linear: int32 reduce + dequant folded in here
            if residual is not None:
                return ops.fused_add_rms_norm_splitk_int8(x, residual, self.weight, self.variance_epsilon, quant), residual
            x = x.materialize()
        if isinstance(x, ops.SplitKPartials):   # a row-parallel linear left its split-K reduce to this kernel
            if residual is not None and not quant:
                return ops.fused_add_rms_norm_splitk(x, residual, self.weight, self.variance_epsilon), residual
            x = x.materialize()
        if quant and x.is_cuda and x.shape[-1] % 8 == 0 and x.shape[-1] <= 8192:
            q = ops.rms_norm_int8(x, self.weight, self.variance_epsilon, residual)
            return q if residual is None else (q, residual)
        if residual is None:
            return ops.rms_norm(x, self.weight, self.variance_epsilon)
        ops.fused_add_rms_norm(x, residual, self.weight, self.variance_epsilon)
        return x, residual


class SiluAndMul:
    """``out = silu(x[..., :d]) * x[..., d:]`` (qwen2_old.py:63,67)."""

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return ops.silu_and_mul(x)


# ---------------------------------------------------------------------------- rope
def build_cos_sin_cache(rot_dim: int, max_position: int, base: float, device=None,
                        table_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """fp32 [max_position, rot_dim] = cos | sin, inv_freq = base^(-2i/rot_dim) (upstream
    ``get_rope`` table layout).  ``table_dtype`` other than fp32 ROUNDS the table through that
    dtype first (upstream casts its table to the model dtype; the kernel always reads fp32)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, rot_dim, 2, dtype=torch.float32) / rot_dim))
    t = torch.arange(max_position, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    table = torch.cat([freqs.cos(), freqs.sin()], dim=-1)
    if table_dtype != torch.float32:
        table = table.to(table_dtype).to(torch.float32)
    return table.to(device) if device is not None else table


class RotaryEmbedding:
    """Neox-style rotary embedding; ``forward(positions, q, k)`` rotates in place and returns
    ``(q, k)`` like upstream's op (qwen2_old.py:153).  ``positions`` [T] or [3, T] (M-RoPE)."""

    def __init__(self, head_size: int, rotary_dim: int, max_position: int, base: float, is_neox_style: bool = True,
                 mrope_section: Optional[Sequence[int]] = None, device=None):
        if not is_neox_style:
            raise NotImplementedError("only neox (rotate-half) pairing is on the talker path (qwen2_old.py:129)")
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        self.max_position_embeddings = max_position
        self.base = base
        self.mrope_section = list(mrope_section) if mrope_section else None
        self.cos_sin_cache = build_cos_sin_cache(rotary_dim, max_position, base, device)

    def _section(self, positions: torch.Tensor):
        if positions.dim() == 2:
            if self.mrope_section is None:
                raise ValueError("[3, T] positions need rope_parameters['mrope_section']")
            return torch.tensor(self.mrope_section, dtype=torch.int32)
        return None

    def __call__(self, positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor):
        ops.rope_neox_and_cache(q, k, None, positions, self.cos_sin_cache, self.head_size,
                                mrope_section=self._section(positions))
        return q, k

    forward = __call__


def get_rope(head_size: int, max_position: int, is_neox_style: bool = True, rope_parameters: Optional[dict] = None,
             device=None) -> RotaryEmbedding:
    """``get_rope(head_size=, max_position=, is_neox_style=True, rope_parameters={"base": θ, **scaling})``
    as called at qwen2_old.py:126-134.  An ``mrope_section`` entry selects M-RoPE."""
    rp = dict(rope_parameters or {})
    base = float(rp.get("base", rp.get("rope_theta", 10000.0)))
    return RotaryEmbedding(head_size, head_size, max_position, base, is_neox_style, rp.get("mrope_section"), device)


# ------------------------------------------------------------------------- attention
class MI355XAttentionBackend:
    """Backend descriptor, in the terms the reference probes on upstream's ``AttentionBackend``
    (vllm_omni/worker/gpu_ar_model_runner.py:431-435; cache shape = layout A2)."""

    forward_includes_kv_cache_update = True
    accept_output_buffer = True

    @staticmethod
    def get_name() -> str:
        return "MI355X_HIP_ATTN"

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int) -> tuple[int, ...]:
        return (num_blocks, 2, block_size, num_kv_heads, head_size)

    @staticmethod
    def get_supported_head_sizes() -> list[int]:
        return [64, 128]

    @staticmethod
    def get_supported_block_sizes() -> list[int]:
        return [8, 16, 32]


# VLLM_OMNI_AMD_BLOCK_ATTENTION=0: single-block steps take the general pair (RoPE + KV-write launch, paged decode) — A/B
_BLOCK_ATTENTION = os.environ.get("VLLM_OMNI_AMD_BLOCK_ATTENTION", "1") != "0"


class Attention:
    """``Attention(num_heads, head_size, scale, num_kv_heads=...)``; ``forward(q, k, v)`` writes K/V
    into the paged cache at ``slot_mapping`` and attends over ``block_table`` (qwen2_old.py:135-144,154).
    """

    backend = MI355XAttentionBackend

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 prefix: str = ""):
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = scale
        self.num_kv_heads = num_kv_heads or num_heads
        self.layer_name = prefix
        self.kv_cache = None  # bound by the runner (bind_kv_cache)
        self._kb = self._vb = None

    def bind_kv_cache(self, layer_kv) -> None:
        pair = normalize_layer_kv(layer_kv)
        if pair is None:
            raise ValueError(f"{self.layer_name}: unsupported KV cache layout")
        self.kv_cache = layer_kv
        self._kb, self._vb = pair

    def _attend(self, q: torch.Tensor, md: AttentionMetadata, out: torch.Tensor) -> torch.Tensor:
        nd = md.num_decodes
        if nd > 0:
            ops.paged_attn_decode(out[:nd], q[:nd], self._kb, self._vb, md.block_table[:nd], md.seq_lens[:nd],
                                  md.max_seq_len, self.scale, partition_size=md.decode_partition_size,
                                  workspace=md.decode_workspace)
        if nd < md.num_reqs:
            t0 = nd  # decodes come first and have one token each
            ops.paged_attn_prefill(out[t0:md.num_actual_tokens], q[t0:md.num_actual_tokens], self._kb, self._vb,
                                   md.block_table[nd:md.num_reqs], md.seq_lens[nd:md.num_reqs],
                                   md.query_start_loc[nd:md.num_reqs + 1] - nd, md.max_query_len, self.scale,
                                   max_seq_len=md.max_seq_len)   # lets a small launch split its tiles' context
        return out

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, output: Optional[torch.Tensor] = None):
        md = get_forward_context().attn_metadata
        out = torch.empty_like(q) if output is None else output
        if md is None:  # profile / warm-up run without a cache: shape-only
            return out.zero_()
        ops.reshape_and_cache(k, v, self._kb, self._vb, md.slot_mapping)
        return self._attend(q, md, out)

    __call__ = forward

    def rope_decode_fusable(self, rope: RotaryEmbedding, positions: torch.Tensor, q_norm_weight=None) -> bool:
        """Whether this step's RoPE + KV write run inside the paged-attention launch (``ops.rope_cache_paged_attn_decode``):
        a pure-decode step on the production decode kernel, plain positions (q/k-norm rides along), not a single-block sequence
        (those have their own fused launch).  qkv_proj then hands over a 16-bit row, not split-K partials."""
        md = get_forward_context().attn_metadata
        if md is None or md.single_block or positions.dim() != 1:
            return False
        nd = md.num_decodes
        return (nd > 0 and nd == md.num_reqs and md.num_actual_tokens == nd
                and ops.rope_decode_fusable(None, self._kb, positions, rope.cos_sin_cache.shape[1],
                                            md.decode_partition_size, nd, md.max_seq_len))

    def forward_fused_rope(self, rope: RotaryEmbedding, positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
                           v: torch.Tensor, q_norm_weight=None, k_norm_weight=None, norm_eps: float = 1e-6,
                           output: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(q/k-norm ->) RoPE -> KV write in ONE launch, then attention: the same result as
        ``rope(positions, q, k)`` followed by ``forward(q, k, v)`` with one launch fewer per layer."""
        md = get_forward_context().attn_metadata
        out = torch.empty_like(q) if output is None else output
        sec = rope._section(positions)
        if (md is not None and md.single_block and sec is None and _BLOCK_ATTENTION
                and ops.block_attention_supported(self.num_heads, self.num_kv_heads, self.head_size, self._kb.shape[1],
                                                  rope.cos_sin_cache.shape[1])):
            pos = positions if md.uniform_position is None else int(md.uniform_position)
            return ops.rope_cache_attend_block(out, q, k, v, pos, rope.cos_sin_cache, self.head_size, self._kb,
                                               self._vb, md.block_table[:, 0], self.scale, q_norm_weight=q_norm_weight,
                                               k_norm_weight=k_norm_weight, norm_eps=norm_eps,
                                               max_position_bound=md.max_seq_len - 1)
        if md is None:
            ops.rope_neox_and_cache(q, k, None, positions, rope.cos_sin_cache, self.head_size, mrope_section=sec,
                                    q_norm_weight=q_norm_weight, k_norm_weight=k_norm_weight, norm_eps=norm_eps)
            return out.zero_()
        nd = md.num_decodes
        if self.rope_decode_fusable(rope, positions, q_norm_weight):
            # pure decode: RoPE + KV write ride in the attention launch (same bits, one launch less per layer)
            ops.rope_cache_paged_attn_decode(out[:nd], q[:nd], k[:nd], v[:nd], positions[:nd], rope.cos_sin_cache, self._kb,
                                             self._vb, md.slot_mapping[:nd], md.block_table[:nd], md.seq_lens[:nd],
                                             md.max_seq_len, self.scale, partition_size=md.decode_partition_size,
                                             workspace=md.decode_workspace, q_norm_weight=q_norm_weight,
                                             k_norm_weight=k_norm_weight, norm_eps=norm_eps)
            return out
        ops.rope_neox_and_cache(q, k, v, positions, rope.cos_sin_cache, self.head_size, self._kb, self._vb,
                                md.slot_mapping, mrope_section=sec, q_norm_weight=q_norm_weight,
                                k_norm_weight=k_norm_weight, norm_eps=norm_eps)
        return self._attend(q, md, out)

    def forward_fused_rope_partials(self, rope: RotaryEmbedding, positions: torch.Tensor, qkv: "ops.SplitKPartials",
                                    q_norm_weight=None, k_norm_weight=None, norm_eps: float = 1e-6) -> torch.Tensor:
        """``forward_fused_rope`` when qkv_proj handed over split-K partials: their reduction (+ bias) happens inside
        the RoPE + KV-write launch; same result bit for bit."""
        md = get_forward_context().attn_metadata
        if (md.single_block and rope._section(positions) is None and _BLOCK_ATTENTION
                and ops.block_attention_supported(self.num_heads, self.num_kv_heads, self.head_size, self._kb.shape[1],
                                                  rope.cos_sin_cache.shape[1])):
            pos = positions if md.uniform_position is None else int(md.uniform_position)
            return ops.rope_cache_attend_block_splitk(qkv, self.num_heads, self.num_kv_heads, self.head_size, pos,
                                                      rope.cos_sin_cache, self._kb, self._vb, md.block_table[:, 0], self.scale,
                                                      q_norm_weight=q_norm_weight, k_norm_weight=k_norm_weight,
                                                      norm_eps=norm_eps, max_position_bound=md.max_seq_len - 1)
        q = ops.rope_neox_and_cache_splitk(qkv, self.num_heads, self.num_kv_heads, self.head_size, positions,
                                           rope.cos_sin_cache, self._kb, self._vb, md.slot_mapping,
                                           mrope_section=rope._section(positions), q_norm_weight=q_norm_weight,
                                           k_norm_weight=k_norm_weight, norm_eps=norm_eps)
        return self._attend(q, md, torch.empty_like(q))


# --------------------------------------------------------------------------- linears
def _init_weight(out_f: int, in_f: int, dtype, device, generator=None, std: float = 0.02) -> torch.Tensor:
    w = torch.empty(out_f, in_f, dtype=torch.float32, device=device)
    w.normal_(0.0, std, generator=generator)
    return w.to(dtype)


# Which GEMM runs a 16-bit linear layer.  "auto": the hand-written weight-streaming kernel (csrc/skinny_gemm.hip)
# for decode batches on the shapes where it is measured faster than the TunableOp-selected library GEMM — the two MLP
# projections of the headline talker at TP 1 and 2 — and the library everywhere else; "library" /
# "kernel" force one side (the latter wherever the kernel's shape envelope allows); "mlp" forces the kernel for the
# MLP pair only (policy experiments).
_LINEAR_MODE = os.environ.get("VLLM_OMNI_AMD_LINEAR", "auto")
_DEFER_REDUCE = os.environ.get("VLLM_OMNI_AMD_DEFER_REDUCE", "1") != "0"   # split-K reduce inside the next RMSNorm
# ... and qkv_proj's inside the RoPE + KV-write kernel (mi_rope_neox_and_cache_splitk): bit-identical and tested, but
# level with the library GEMM at every batch size measured (15.14 vs 15.15 ms/step at 256 rows, 4.79 vs 4.80 at 32),
# so it is off unless asked for
_DEFER_QKV = os.environ.get("VLLM_OMNI_AMD_DEFER_QKV", "0") == "1"
# with a tuned entry for the shape the table decides (qkv partials into the RoPE launch when measured ahead); "0" pins
# qkv_proj to a 16-bit output
_DEFER_QKV_TUNED = os.environ.get("VLLM_OMNI_AMD_DEFER_QKV", "1") != "0"


def _tuned(m: int, n: int, k: int, gated: bool = False):
    """The measured entry of this shape in ``tuned/linear_plans_gfx950.json`` (tools/tune_linear_plans.py), or None:
    which tile plan the kernel runs there (registered with the library on first use) and what it, its partials-only
    half and the library GEMM take.  Shapes outside the table keep the round-3 rules below."""
    if _LINEAR_MODE != "auto":
        return None
    from .tuned import linear_plans

    linear_plans.load()
    return linear_plans.lookup(m, n, k, gated)


# what folding a split-K reduce into the consumer (fused-add RMSNorm, RoPE + KV write) adds to that launch, in us
_DEFERRED_REDUCE_US = 1.0
_DEFER_TRAFFIC = os.environ.get("VLLM_OMNI_AMD_DEFER_TRAFFIC", "1") != "0"   # A/B: price the slabs' traffic too (predictor shapes)


def _streaming_kernel_wins(m: int, n: int, k: int) -> bool:
    # measured in situ at the headline talker's shapes, whole and TP-sharded, batch 32-256 per rank:
    # ahead of the library at K 18944 and 9472 (TP 1, 2), level at 4736 with 128 rows, behind with 256 (TP 4)
    if m <= 256 and k >= 8192 and k >= 2 * n:
        return True
    # at <= 32 rows the kernel is at or ahead of the library on every projection measured IN A GRAPH — talker qkv 12.7
    # against 13.4 us, o 11.7 / 12.6 (profiles/r03c_skinny_shard_shapes.log), the code predictor's 1-6 M-element
    # linears: 9.16 against 9.70 ms per call at batch 32 (profiles/r03j_predictor_linear_modes.log) — not at 64 rows
    # (o_proj 11.6 against 8.7) and not at 512 (19.5 against 15.9 ms per predictor call)
    if m <= 32 and n * k >= (1 << 20):
        return True
    # the down_proj of a SMALL dense MLP (the code predictor's 1024 x 3072) up to 256 rows: with the fused gate_up in
    # front of it the pair runs 8.67 against 9.53 ms per predictor call at 64 rows, 11.47 / 12.17 at 256, level at 128
    # (profiles/r03j_predictor_linear_modes.log)
    return m <= 256 and n <= 2048 and k <= 4096 and k > 2 * n and n * k >= (1 << 20)


def fused_gate_up_silu(layer, x: torch.Tensor) -> Optional[torch.Tensor]:
    """``act_fn(gate_up_proj(x))`` in one launch (``ops.linear_silu_mul``) when the policy picks the streaming kernel
    for this shape, else None (the caller runs the GEMM and ``SiluAndMul`` apart).  Column-parallel, no bias, 16-bit
    weights only; under TP a rank's weight is [gate shard; up shard], the layout the kernel expects."""
    if isinstance(x, ops.QuantizedActs):
        n2, k = layer.qweight.shape
        if layer.bias is None and n2 % 32 == 0 and ops.int8_linear_supported(min(x.xq.shape[0], ops.INT8_TILE_ROWS), n2 // 2, k):
            return ops.int8_linear_silu_mul(x, layer.qweight, layer.weight_scale)
        return None
    if layer.bias is not None or getattr(layer, "gather_output", False) or not x.is_cuda:
        return None
    if getattr(layer, "qweight", None) is not None:   # W8A8: the gated form of the tiled int8 kernel
        m = x.numel() // x.shape[-1]
        n2, k = layer.qweight.shape
        if (n2 % 32 == 0 and ops.int8_linear_supported(min(m, ops.INT8_TILE_ROWS), n2 // 2, k)
                and os.environ.get("VLLM_OMNI_AMD_INT8", "kernel") == "kernel"):
            return ops.int8_linear_silu_mul(x, layer.qweight, layer.weight_scale)
        return None
    if _LINEAR_MODE == "library":
        return None
    w = layer.weight
    m = x.numel() // x.shape[-1]
    n, k = w.shape
    if n % 32 or not ops.linear_supported(m, n // 2, k, x.dtype):
        return None
    e = _tuned(m, n // 2, k, gated=True)
    if e is not None and e.get("full"):
        return ops.linear_silu_mul(x, w) if e["full"]["us"] < e["library_us"] else None
    if _LINEAR_MODE not in ("kernel", "mlp", "mlp+o") and not _fused_gate_up_wins(m, n // 2, k):
        return None
    return ops.linear_silu_mul(x, w)


def _deferred_kernel_wins(m: int, n: int, k: int) -> bool:
    # square-ish projections (o_proj 3584 x 3584) when the split-K reduce rides in the fused-add RMSNorm behind them:
    # level with the library at 256 rows, 1-1.5 % of the step ahead at 32-128 rows (same measurements)
    # (not small ones: the code predictor's 1024 x 2048 o_proj ran 5 % slower per call on the kernel)
    return m <= 256 and k >= n and n * k >= (1 << 23)


def _deferred_qkv_wins(m: int, n: int, k: int) -> bool:
    # qkv_proj (4608 x 3584) with its split-K reduce (+ bias) inside the fused RoPE + KV-write kernel behind it
    return m <= 256 and n * k >= (1 << 23) and _DEFER_QKV


def _fused_gate_up_wins(m: int, inter: int, k: int) -> bool:
    # same measurements: ahead at I 18944 and 9472 per rank (>= ~120 tiles in one launch), not at 4736
    if m <= 256 and inter >= 8192:
        return True
    # <= 32 rows: the fused launch replaces a GEMM the kernel already wins (see _streaming_kernel_wins) AND the
    # SiluAndMul launch behind it (the code predictor's 6144 x 1024 gate_up_proj at batch 32)
    if os.environ.get("VLLM_OMNI_AMD_FUSE_SMALL_GATE_UP", "1") != "1" or 2 * inter * k < (1 << 20):
        return False
    # ... and, up to 256 rows, the gate_up of a small dense MLP (the code predictor's 6144 x 1024; see _streaming_kernel_wins)
    return m <= 32 or (m <= 256 and k <= 2048 and inter <= 4096)


def _apply_linear(layer, x: torch.Tensor, bias: Optional[torch.Tensor], defer_reduce: bool = False):
    """16-bit GEMM (library or the decode-batch streaming kernel), or the W8A8 path once ``quantize_int8_`` has run
    on the layer.  ``defer_reduce``: when the streaming kernel runs split-K, return its partials
    (``ops.SplitKPartials``) for the fused-add RMSNorm behind the layer to reduce."""
    if getattr(layer, "qweight", None) is not None:
        if defer_reduce is True and _DEFER_REDUCE and os.environ.get("VLLM_OMNI_AMD_INT8", "kernel") == "kernel":
            part = ops.int8_linear_partials(x, layer.qweight, layer.weight_scale, bias)   # None: no split-K plan
            if part is not None:
                return part
        return ops.int8_linear(x, layer.qweight, layer.weight_scale, bias)
    w = layer.weight
    if _LINEAR_MODE != "library" and x.is_cuda:
        m = x.numel() // x.shape[-1]
        n, k = w.shape
        e = _tuned(m, n, k) if ops.linear_supported(m, n, k, x.dtype) else None
        if e is not None and e.get("full"):
            # measured at this shape: the cheapest of {library, kernel + its reduce launch, partials reduced downstream}
            best_direct = min(e["library_us"], e["full"]["us"])
            part = e.get("partials")
            # what the consumer pays for reducing the slabs: a constant for the shard shapes (as measured in situ there); for
            # the code predictor's small layers (tp 0 entries) also the slabs' traffic — at 128 rows x 8 slices they are 4 MB
            # against a 4 MB weight, and the table's own 1 us margin chose the slower route (8.01 -> 8.25 ms per call)
            defer_us = _DEFERRED_REDUCE_US
            if part and e.get("tp") == 0 and _DEFER_TRAFFIC:
                defer_us += int(part["split"]) * m * n * 4 / 3.0e6
            if defer_reduce and _DEFER_REDUCE and part and part["us"] + defer_us < best_direct and (
                    defer_reduce != "rope" or _DEFER_QKV_TUNED):
                pr = ops.linear_partials(x, w, bias)
                if pr is not None:
                    return pr
            return ops.linear(x, w, bias) if e["full"]["us"] < e["library_us"] else F.linear(x, w, bias)
        if ops.linear_supported(m, n, k, x.dtype):
            wins = (_LINEAR_MODE == "kernel" or _streaming_kernel_wins(m, n, k)
                    or (_LINEAR_MODE in ("mlp", "mlp+o") and getattr(layer, "_is_down_proj", False))
                    or (_LINEAR_MODE == "mlp+o" and getattr(layer, "_is_o_proj", False)))
            if defer_reduce and _DEFER_REDUCE and (
                    wins or (_deferred_qkv_wins(m, n, k) if defer_reduce == "rope" else _deferred_kernel_wins(m, n, k))):
                part = ops.linear_partials(x, w, bias)
                if part is not None:
                    return part
            if wins:
                return ops.linear(x, w, bias)
    return F.linear(x, w, bias)


class _Weights:
    """a bare (weight, bias) pair in the shape ``_apply_linear`` reads a layer"""

    __slots__ = ("weight", "bias", "qweight")

    def __init__(self, weight, bias=None):
        self.weight, self.bias, self.qweight = weight, bias, None


def tuned_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear`` for the step's un-sharded extras (``thinker_to_talker_proj``, the vocab-parallel ``lm_head``), routed
    like the decoder's projections: the streaming kernel where the tuned table (or the round-3 rule) has it ahead of
    the library GEMM at this shape."""
    if x.dim() > 2 or not x.is_cuda:
        return F.linear(x, weight, bias)
    return _apply_linear(_Weights(weight, bias), x, bias)


def quantize_int8_(layer) -> None:
    """Online W8A8 quantisation of one linear layer after its weights are loaded — the reference's
    ``Int8OnlineLinearMethod.process_weights_after_loading`` (vllm_omni/quantization/int8_config.py:391-418:
    ``ops.scaled_int8_quant(weight, scale=None)``): int8 weight [N, K] and an fp32 scale per output channel; the
    16-bit weight is dropped.  Activations are quantised per token at run time."""
    w = layer.weight
    layer.qweight, layer.weight_scale = ops.dynamic_int8_quant(w)   # a row of [N, K] is an output channel
    layer.weight = None


class ColumnParallelLinear:
    """y = x W^T (+ b), output features sharded over TP ranks (no communication)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False, dtype=torch.bfloat16, device=None,
                 generator=None, gather_output: bool = False):
        self.tp_size = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        if output_size % self.tp_size:
            raise ValueError("output_size not divisible by tp_size")
        self.input_size, self.output_size = input_size, output_size
        self.output_size_per_partition = output_size // self.tp_size
        self.gather_output = gather_output
        self.weight = _init_weight(self.output_size_per_partition, input_size, dtype, device, generator)
        self.bias = torch.zeros(self.output_size_per_partition, dtype=dtype, device=device) if bias else None

    def __call__(self, x: torch.Tensor, defer_reduce=False):
        """``defer_reduce="rope"`` (qkv_proj): the caller runs the fused RoPE + KV-write kernel next, which can take the
        split-K partials of the streaming kernel directly (``ops.rope_neox_and_cache_splitk``)."""
        y = _apply_linear(self, x, self.bias, defer_reduce if not self.gather_output else False)
        if self.gather_output:
            y = tensor_model_parallel_all_gather(y)
        return y, None


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel projections fused along the output dim (gate_up_proj,
    qwen2_old.py:47-53); each rank holds its shard of EACH sub-projection, concatenated."""

    def __init__(self, input_size: int, output_sizes: Sequence[int], bias: bool = False, **kw):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias=bias, **kw)


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q/k/v projection with head-wise sharding and KV-head replication
    (qwen2_old.py:109-117; partition rules :89-102)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int, total_num_kv_heads: int,
                 bias: bool = True, dtype=torch.bfloat16, device=None, generator=None):
        from .distributed import partition_heads

        self.tp_size = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.num_heads, self.num_kv_heads, self.num_kv_head_replicas = partition_heads(
            total_num_heads, total_num_kv_heads, self.tp_size)
        self.head_size = head_size
        self.input_size = hidden_size
        self.output_size_per_partition = (self.num_heads + 2 * self.num_kv_heads) * head_size
        self.output_size = (total_num_heads + 2 * total_num_kv_heads) * head_size
        self.gather_output = False
        self.weight = _init_weight(self.output_size_per_partition, hidden_size, dtype, device, generator)
        self.bias = torch.zeros(self.output_size_per_partition, dtype=dtype, device=device) if bias else None


class RowParallelLinear:
    """y = sum_ranks(x_shard W_shard^T) (+ b): input features sharded, TP all-reduce on the
    result (o_proj / down_proj, qwen2_old.py:54-60,118-124)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False, dtype=torch.bfloat16, device=None,
                 generator=None, reduce_results: bool = True):
        self.tp_size = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        if input_size % self.tp_size:
            raise ValueError("input_size not divisible by tp_size")
        self.input_size, self.output_size = input_size, output_size
        self.input_size_per_partition = input_size // self.tp_size
        self.reduce_results = reduce_results
        self.weight = _init_weight(output_size, self.input_size_per_partition, dtype, device, generator)
        self.bias = torch.zeros(output_size, dtype=dtype, device=device) if bias else None

    def __call__(self, x: torch.Tensor, defer_reduce: bool = False):
        # bias is added once (rank 0) so the all-reduced sum carries it exactly once
        # defer_reduce (no TP only: an all-reduce needs the 16-bit tensor): the caller hands y to a fused-add RMSNorm
        # TP > 1 with the in-tree all-reduce: its copy-in phase takes the split-K partials directly (same trick)
        custom = self.tp_size > 1 and self.reduce_results and get_parallel_state().custom_ar is not None
        defer = (defer_reduce and self.tp_size == 1) or custom
        y = _apply_linear(self, x, self.bias if self.tp_rank == 0 else None, defer)
        if self.reduce_results and self.tp_size > 1:
            if isinstance(y, ops.Int8SplitKPartials):
                y = tensor_model_parallel_all_reduce(y.materialize())
            elif isinstance(y, ops.SplitKPartials):
                r = tensor_model_parallel_all_reduce_partials(y)
                y = r if r is not None else tensor_model_parallel_all_reduce(y.materialize())
            else:
                y = tensor_model_parallel_all_reduce(y)
        return y, None


# ------------------------------------------------------- weight sharding (pure indexing)
def shard_column(weight: torch.Tensor, tp_size: int, tp_rank: int) -> torch.Tensor:
    per = weight.shape[0] // tp_size
    return weight[tp_rank * per:(tp_rank + 1) * per]


def shard_row(weight: torch.Tensor, tp_size: int, tp_rank: int) -> torch.Tensor:
    per = weight.shape[1] // tp_size
    return weight[:, tp_rank * per:(tp_rank + 1) * per]


def shard_merged_column(weight: torch.Tensor, output_sizes: Sequence[int], tp_size: int, tp_rank: int) -> torch.Tensor:
    parts, off = [], 0
    for sz in output_sizes:
        parts.append(shard_column(weight[off:off + sz], tp_size, tp_rank))
        off += sz
    return torch.cat(parts, dim=0)


def shard_qkv(weight: torch.Tensor, head_size: int, total_heads: int, total_kv_heads: int, tp_size: int,
              tp_rank: int) -> torch.Tensor:
    """Rows of the fused [q | k | v] weight (or bias) that ``tp_rank`` owns; KV heads are replicated
    when ``total_kv_heads < tp_size`` (qwen2_old.py:94-102)."""
    from .distributed import kv_head_for_rank, partition_heads

    nq, _, _ = partition_heads(total_heads, total_kv_heads, tp_size)
    q_end = total_heads * head_size
    k_end = q_end + total_kv_heads * head_size
    q = weight[tp_rank * nq * head_size:(tp_rank + 1) * nq * head_size]
    kvr = kv_head_for_rank(total_kv_heads, tp_size, tp_rank)
    k = weight[q_end + kvr.start * head_size:q_end + kvr.stop * head_size]
    v = weight[k_end + kvr.start * head_size:k_end + kvr.stop * head_size]
    return torch.cat([q, k, v], dim=0)
