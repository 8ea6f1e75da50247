"""ORACLE — test infrastructure, NOT product code.

CPU restatement (torch, fp32 arithmetic) of the reference's decode hot path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``vllm-omni_amd/``) never does.

Pinning status:
  * INDEXING (slots, block tables, layouts, gathers): pinned by the constants of the
    reference's own tests, restated in tests/golden/ (test_prefix_cache.py,
    test_kv_flow.py, test_tp_rank_aware.py).
  * ARITHMETIC (attention, RMSNorm, RoPE, SiLU-mul, int8): **parity unpinned by the
    reference** — the reference has no test or fixture for these and the arithmetic lives in
    un-vendored upstream ``vllm==0.19.0`` (docker/Dockerfile.rocm:1,13), absent here.  The
    restatement follows the reference's call sites and its in-tree HF-exact statement of
    the same math (models/common/qwen3_code_predictor.py:46-223), and is cross-checked
    against the locally installed HF ``transformers`` implementation of the same
    architecture (tests/golden/make_golden.py), whose outputs are committed as fixtures.

Each function cites the reference file:line it follows (paths relative to the reference
root).  All functions take/return CPU tensors and compute in fp32; ``round_to`` arguments
reproduce the documented rounding points of a 16-bit model dtype.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

F32 = torch.float32


def _rt(x: torch.Tensor, dt: Optional[torch.dtype]) -> torch.Tensor:
    """Round through ``dt`` (if given) and come back to fp32."""
    return x if dt is None else x.to(dt).to(F32)


# ----------------------------------------------------------------------------------
# A2/A3/A4 — layouts, slots, KV write, gather
# ----------------------------------------------------------------------------------
def normalize_layer_kv(layer_kv):
    """(key_blocks, value_blocks) from any of the three accepted layouts, else None.

    Follows vllm_omni/distributed/omni_connectors/utils/kv_utils.py:378-450:
    stacked ``[2, nb, ...]`` (K/V at dim 0) is tested FIRST, then ``[nb, 2, ...]``, then a
    2-tuple of tensors; both results must be at least 2-D.
    """
    if isinstance(layer_kv, torch.Tensor):
        if layer_kv.ndim >= 3 and layer_kv.shape[0] == 2:
            kb, vb = layer_kv[0], layer_kv[1]
        elif layer_kv.ndim >= 3 and layer_kv.shape[1] == 2:
            kb, vb = layer_kv[:, 0], layer_kv[:, 1]
        else:
            return None
    elif isinstance(layer_kv, tuple):
        if len(layer_kv) != 2:
            return None
        kb, vb = layer_kv
        if not isinstance(kb, torch.Tensor) or not isinstance(vb, torch.Tensor):
            return None
    else:
        return None
    if kb.ndim < 2 or vb.ndim < 2:
        return None
    return kb, vb


def extract_kv(key_blocks: torch.Tensor, value_blocks: torch.Tensor, block_ids: Sequence[int], seq_len: int):
    """Gather one request's K/V: blocks[ids].flatten(0,1)[:seq_len] -> [seq, heads, dim].

    Follows vllm_omni/distributed/omni_connectors/kv_transfer_manager.py:829-848: ids outside
    ``[0, min(nk, nv) - 1]`` are dropped; truncation only when seq_len < gathered rows.
    Returns None when no id is valid.
    """
    max_block = min(key_blocks.shape[0], value_blocks.shape[0]) - 1
    valid = [b for b in block_ids if 0 <= b <= max_block]
    if not valid:
        return None
    fk = key_blocks[valid].flatten(0, 1)
    fv = value_blocks[valid].flatten(0, 1)
    if seq_len < fk.shape[0]:
        fk, fv = fk[:seq_len], fv[:seq_len]
    return fk.contiguous(), fv.contiguous()


def truncate_block_ids(block_ids: Sequence[int], seq_len: int, block_size: int) -> list[int]:
    """ceil(seq_len / block_size) leading ids (omni_ar_scheduler.py:644-651)."""
    n = (seq_len + block_size - 1) // block_size
    return list(block_ids[:n]) if len(block_ids) > n else list(block_ids)


def slot_mapping(block_table: torch.Tensor, req_indices: torch.Tensor, positions: torch.Tensor,
                 block_size: int) -> torch.Tensor:
    """slot = block_table[req, pos // b] * b + pos % b  (int64).

    The identity the reference's prefix cache relies on (core/prefix_cache.py:16-28,112-133;
    tests/core/test_prefix_cache.py:173-185: slots 8..15 <-> blocks 2,3 at b=4).
    """
    pos = positions.to(torch.int64)
    blk = block_table[req_indices.to(torch.int64), pos // block_size].to(torch.int64)
    return blk * block_size + pos % block_size


def scatter_rows_by_slot(cache: torch.Tensor, slot_map: torch.Tensor, rows: torch.Tensor) -> None:
    """cache.view(nb*b, F)[slot] = rows, skipping negative (padding) slots.

    core/prefix_cache.py:124-133 (``flat_cache[unpadded_slot_mapping] = hidden_states``).
    """
    b = cache.shape[1]
    keep = slot_map >= 0
    s = slot_map[keep].to(torch.int64)
    # row `slot` of the flattened (nb*b, F) view == cache[slot // b, slot % b]; indexing the
    # 2-D way also works on strided K / V views of a stacked cache tensor
    cache[s // b, s % b] = rows[keep].reshape(-1, *cache.shape[2:]).to(cache.dtype)


def reshape_and_cache(k: torch.Tensor, v: torch.Tensor, key_blocks: torch.Tensor, value_blocks: torch.Tensor,
                      slot_map: torch.Tensor) -> None:
    """K/V write half of upstream Attention.forward (qwen2_old.py:154): token t goes to
    ``blocks[slot // b][slot % b]``; slot < 0 is padding."""
    scatter_rows_by_slot(key_blocks, slot_map, k)
    scatter_rows_by_slot(value_blocks, slot_map, v)


def cached_block_ids(block_table_row: torch.Tensor, num_computed: int, block_size: int) -> torch.Tensor:
    """core/prefix_cache.py:255-264: only FULL blocks are cached."""
    return block_table_row[: num_computed // block_size]


# ----------------------------------------------------------------------------------
# A5/A6 — paged attention
# ----------------------------------------------------------------------------------
def _gather_seq(blocks: torch.Tensor, table_row: torch.Tensor, n: int) -> torch.Tensor:
    b = blocks.shape[1]
    nblk = (n + b - 1) // b
    return blocks[table_row[:nblk].to(torch.int64)].flatten(0, 1)[:n].to(F32)  # [n, Hkv, D]


def paged_attention_decode(q: torch.Tensor, key_blocks: torch.Tensor, value_blocks: torch.Tensor,
                           block_table: torch.Tensor, seq_lens: torch.Tensor, scale: float,
                           p_round_to: Optional[torch.dtype] = None) -> torch.Tensor:
    """One query token per sequence against its paged context; fp32 result [S, nq, D].

    o[s,h] = softmax(q[s,h] . K[s, kv(h), :C_s]^T * scale) . V[s, kv(h), :C_s],
    kv(h) = h // (nq / nkv), scale = head_dim**-0.5 (qwen2_old.py:106,135-144,154);
    softmax in fp32 (HF eager statement of the same layer:
    transformers/models/qwen2_5_omni/modeling_qwen2_5_omni.py:540-562).
    ``p_round_to``: round the probabilities to the cache dtype before the PV product, as
    flash-style kernels (incl. the reference's TRITON_ATTN backend) do.
    """
    s, nq, d = q.shape
    nkv = key_blocks.shape[2]
    grp = nq // nkv
    out = torch.zeros(s, nq, d, dtype=F32)
    for i in range(s):
        n = int(seq_lens[i])
        if n == 0:
            continue
        k = _gather_seq(key_blocks, block_table[i], n)  # [n, nkv, d]
        v = _gather_seq(value_blocks, block_table[i], n)
        k = k.repeat_interleave(grp, dim=1)  # [n, nq, d]
        v = v.repeat_interleave(grp, dim=1)
        sc = torch.einsum("hd,nhd->hn", q[i].to(F32), k) * scale
        p = torch.softmax(sc, dim=-1)
        if p_round_to is not None:
            # flash kernels round the UNNORMALISED exp(s - max) and divide by the fp32 sum
            m = sc.max(dim=-1, keepdim=True).values
            e = torch.exp(sc - m)
            out[i] = torch.einsum("hn,nhd->hd", _rt(e, p_round_to), v) / e.sum(-1, keepdim=True)
        else:
            out[i] = torch.einsum("hn,nhd->hd", p, v)
    return out


def paged_attention_prefill(q: torch.Tensor, key_blocks: torch.Tensor, value_blocks: torch.Tensor,
                            block_table: torch.Tensor, seq_lens: torch.Tensor, query_start_loc: torch.Tensor,
                            scale: float) -> torch.Tensor:
    """Varlen causal attention of query chunks against the paged cache.

    Query row i of sequence s (chunk length L_s = qsl[s+1]-qsl[s]) sits at absolute position
    ``seq_lens[s] - L_s + i`` and sees cache positions ``[0, that position]``.  The cache already
    holds the chunk's own K/V (the KV write precedes attention in the layer, qwen2_old.py:154).
    """
    t, nq, d = q.shape
    nkv = key_blocks.shape[2]
    grp = nq // nkv
    out = torch.zeros(t, nq, d, dtype=F32)
    for s in range(seq_lens.shape[0]):
        q0, q1 = int(query_start_loc[s]), int(query_start_loc[s + 1])
        n = int(seq_lens[s])
        ql = q1 - q0
        if ql == 0:
            continue
        k = _gather_seq(key_blocks, block_table[s], n).repeat_interleave(grp, dim=1)
        v = _gather_seq(value_blocks, block_table[s], n).repeat_interleave(grp, dim=1)
        sc = torch.einsum("qhd,nhd->hqn", q[q0:q1].to(F32), k) * scale
        qpos = torch.arange(n - ql, n).view(1, ql, 1)
        kpos = torch.arange(n).view(1, 1, n)
        sc = sc.masked_fill(kpos > qpos, float("-inf"))
        out[q0:q1] = torch.einsum("hqn,nhd->qhd", torch.softmax(sc, dim=-1), v)
    return out


# ----------------------------------------------------------------------------------
# A7 / A7b — RoPE (neox), M-RoPE, per-head q/k norm
# ----------------------------------------------------------------------------------
def build_cos_sin_cache(rot_dim: int, max_position: int, base: float) -> torch.Tensor:
    """fp32 [max_position, rot_dim]: cos | sin, inv_freq = base^(-2i/rot_dim).

    ``get_rope(head_size, max_position, is_neox_style=True, rope_parameters={"base": θ})``
    (qwen2_old.py:126-134); frequency formula as in the reference's HF-exact code
    (models/common/qwen3_code_predictor.py:65-102).
    """
    inv_freq = 1.0 / (base ** (torch.arange(0, rot_dim, 2, dtype=torch.float32) / rot_dim))
    t = torch.arange(max_position, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    return torch.cat([freqs.cos(), freqs.sin()], dim=-1)


def rope_neox(positions: torch.Tensor, x: torch.Tensor, head_dim: int, cos_sin_cache: torch.Tensor,
              mrope_section: Optional[Sequence[int]] = None,
              round_to: Optional[torch.dtype] = None) -> torch.Tensor:
    """Rotate-half RoPE on the first rot_dim dims of every head of x [T, H*head_dim].

    Pairing (i, i + rot_dim/2): out_i = x_i cos - x_{i+r/2} sin; out_{i+r/2} = x_{i+r/2} cos + x_i sin
    (``_rotate_half``, qwen3_code_predictor.py:65-70; HF rotate_half modeling_qwen2_5_omni.py:977-981).
    M-RoPE: positions [3, T]; half-dim index j takes its angle from row 0/1/2 by section
    (HF apply_multimodal_rotary_pos_emb, modeling_qwen2_5_omni.py:1387-1429).
    fp32 arithmetic, one rounding (``round_to``).
    """
    t = x.shape[0]
    rot = cos_sin_cache.shape[1]
    half = rot // 2
    xf = x.to(F32).view(t, -1, head_dim)
    if positions.dim() == 2:
        assert mrope_section is not None and sum(mrope_section) == half
        bounds = [0, mrope_section[0], mrope_section[0] + mrope_section[1], half]
        cos = torch.empty(t, half)
        sin = torch.empty(t, half)
        for r in range(3):
            cs = cos_sin_cache[positions[r].to(torch.int64)]
            cos[:, bounds[r]:bounds[r + 1]] = cs[:, bounds[r]:bounds[r + 1]]
            sin[:, bounds[r]:bounds[r + 1]] = cs[:, half + bounds[r]:half + bounds[r + 1]]
    else:
        cs = cos_sin_cache[positions.to(torch.int64)]
        cos, sin = cs[:, :half], cs[:, half:]
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    x1, x2 = xf[..., :half], xf[..., half:rot]
    o = xf.clone()
    o[..., :half] = x1 * cos - x2 * sin
    o[..., half:rot] = x2 * cos + x1 * sin
    return _rt(o.reshape(t, -1), round_to)


def head_rms_norm(x: torch.Tensor, weight: torch.Tensor, head_dim: int, eps: float,
                  round_to: Optional[torch.dtype] = None) -> torch.Tensor:
    """Per-head RMSNorm over head_dim (Qwen3 q_norm/k_norm, before RoPE)."""
    t = x.shape[0]
    return rms_norm(x.reshape(t, -1, head_dim), weight, eps, round_to).reshape(t, -1)


# ----------------------------------------------------------------------------------
# A8 / A9 — RMSNorm, SiLU-gate
# ----------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, round_to: Optional[torch.dtype] = None) -> torch.Tensor:
    """fp32 normalise -> cast to model dtype -> * weight (then the product is rounded).

    ``_RMSNorm`` in the reference (models/common/qwen3_code_predictor.py:46-62) == HF
    Qwen2RMSNorm (modeling_qwen2_5_omni.py:1084-1098); used at qwen2_old.py:202-203,212-216,223.
    """
    xf = x.to(F32)
    var = xf.pow(2).mean(-1, keepdim=True)
    n = _rt(xf * torch.rsqrt(var + eps), round_to)
    return _rt(n * weight.to(F32), round_to)


def fused_add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                       round_to: Optional[torch.dtype] = None):
    """(normed, new_residual): new_residual = x + residual stored in the model dtype, then
    RMSNorm of it (qwen2_old.py:212-216,223: ``hidden, residual = norm(hidden, residual)``)."""
    z = _rt(x.to(F32) + residual.to(F32), round_to)
    return rms_norm(z, weight, eps, round_to), z


def silu_and_mul(x: torch.Tensor, round_to: Optional[torch.dtype] = None) -> torch.Tensor:
    """out[:, j] = silu(x[:, j]) * x[:, I + j]  (``SiluAndMul`` on gate_up, qwen2_old.py:63,65-69;
    MLP statement qwen3_code_predictor.py:181-191).  silu is rounded before the product, as
    separate ``act_fn(gate) * up`` ops are in a 16-bit model."""
    inter = x.shape[-1] // 2
    g, u = x[..., :inter].to(F32), x[..., inter:].to(F32)
    return _rt(_rt(g * torch.sigmoid(g), round_to) * u, round_to)


# ----------------------------------------------------------------------------------
# A11 — talker logit masking
# ----------------------------------------------------------------------------------
def bad_word_processor(logits: torch.Tensor, suppress_start_id: Optional[int], end_id: Optional[int],
                       bos_id: Optional[int]) -> torch.Tensor:
    """In place; follows qwen2_5_omni_talker.py:150-169 branch by branch."""
    v = logits.size(-1)
    if suppress_start_id and suppress_start_id < v:
        if end_id is None:
            raise ValueError("config must have tts_codec_end_token_id attribute")
        if suppress_start_id == end_id:
            logits[..., end_id + 1: v] = -1e9
        elif suppress_start_id < end_id:
            logits[..., suppress_start_id:end_id] = -1e9
            logits[..., end_id + 1: v] = -1e9
        else:
            logits[..., suppress_start_id:v] = -1e9
    if bos_id is not None:
        logits[..., bos_id] = -1e9
    return logits


# ----------------------------------------------------------------------------------
# A13 — W8A8 int8 scheme
# ----------------------------------------------------------------------------------
def dynamic_int8_quant(x: torch.Tensor):
    """Per-token symmetric: scale = absmax/127, q = clamp(round_half_even(x/scale), -128, 127).

    Scheme of vllm_omni/quantization/int8_config.py:323-342 (dynamic, symmetric, per-token).
    Rounding mode / clamp bound of upstream ``scaled_int8_quant`` are UNPINNED offline
    (SURVEY §8c vii): this restatement uses round-half-even and [-128, 127].
    """
    xf = x.to(F32)
    amax = xf.abs().amax(-1, keepdim=True)
    scale = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
    q = torch.clamp(torch.round(xf / scale), -128, 127).to(torch.int8)
    return q, scale.squeeze(-1)


def int8_weight_quant(w: torch.Tensor):
    """Per-output-channel symmetric weight quant, weight [N, K] -> int8 [N, K], scale fp32 [N]
    (int8_config.py:180-212,391-418)."""
    return dynamic_int8_quant(w)


def int8_linear(x: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None):
    xq, xs = dynamic_int8_quant(x)
    acc = xq.to(torch.int32) @ wq.to(torch.int32).t()  # exact int32
    y = acc.to(F32) * xs.unsqueeze(-1) * w_scale.unsqueeze(0)
    if bias is not None:
        y = y + bias.to(F32)
    return y
