"""Torch-tensor wrappers over the C ABI.  Plumbing only: no arithmetic happens here.

Every function launches on torch's *current* stream (so it is hipGraph-capturable under
``torch.cuda.graph``) and raises on non-GPU tensors — there is no CPU path in the product.
Reference call sites these stand in for are cited in include/vllm_omni_amd.h.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Union

import torch

from . import _C
from ._C import check, lib

_DT = {torch.float16: _C.MI_DTYPE_F16, torch.bfloat16: _C.MI_DTYPE_BF16, torch.float32: _C.MI_DTYPE_F32}


def _dt(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise TypeError(f"unsupported dtype {t}") from None


def _gpu(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "vllm_omni_amd ops run only on the GPU (HIP kernels); got a CPU tensor. "
                "There is no CPU fallback by design."
            )


# the raw handle of torch's current stream without building a torch.cuda.Stream object per launch (an eager
# prefill chunk is ~500 launches; the object form measured 5-6 us of host time per call, this one well under 1)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _cache_strides(kc: torch.Tensor, vc: torch.Tensor):
    """k_cache / v_cache are views shaped [num_blocks, block_size, kv_heads, head_dim]."""
    if kc.dim() != 4 or vc.shape != kc.shape or kc.stride() != vc.stride() or kc.stride(3) != 1:
        raise ValueError("k_cache/v_cache must be [num_blocks, block_size, kv_heads, head_dim] views "
                         "with identical strides and contiguous head_dim")
    return kc.stride(0), kc.stride(1), kc.stride(2)


# --------------------------------------------------------------------------- A4
def reshape_and_cache(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                      slot_mapping: torch.Tensor) -> None:
    """cache[slot] = (k, v)[token] for slot >= 0.  k, v: [T, kv_heads, head_dim] (or flattened)."""
    _gpu(k, v, k_cache, v_cache, slot_mapping)
    nb, bs, h, d = k_cache.shape
    t = k.shape[0]
    if slot_mapping.dtype != torch.int64 or slot_mapping.numel() < t:
        raise ValueError("slot_mapping must be int64 with >= num_tokens entries")
    cb, ct, ch = _cache_strides(k_cache, v_cache)
    check("mi_reshape_and_cache", lib.mi_reshape_and_cache(
        _p(k), _p(v), _p(k_cache), _p(v_cache), _p(slot_mapping), t, h, d, bs,
        k.stride(0), v.stride(0), cb, ct, ch, _dt(k.dtype), _stream()))


# --------------------------------------------------------------------- A7/A7b/A4
def rope_neox_and_cache(q: torch.Tensor, k: torch.Tensor, v: Optional[torch.Tensor],
                        positions: torch.Tensor, cos_sin_cache: torch.Tensor, head_dim: int,
                        k_cache: Optional[torch.Tensor] = None, v_cache: Optional[torch.Tensor] = None,
                        slot_mapping: Optional[torch.Tensor] = None,
                        mrope_section: Optional[torch.Tensor] = None,
                        q_norm_weight: Optional[torch.Tensor] = None,
                        k_norm_weight: Optional[torch.Tensor] = None, norm_eps: float = 1e-6) -> None:
    """In-place (q/k-norm ->) neox RoPE on q [T, nq*d], k [T, nkv*d]; optional KV-cache write.

    positions int64 [T] or [3, T] (then ``mrope_section`` is an int32 *host* tensor of 3 entries);
    cos_sin_cache fp32 [max_position, rot_dim].
    """
    _gpu(q, k, v, positions, cos_sin_cache, k_cache, v_cache, slot_mapping, q_norm_weight, k_norm_weight)
    t = q.shape[0]
    nq = q.shape[-1] // head_dim if q.dim() == 2 else q.shape[1]
    nkv = k.shape[-1] // head_dim if k.dim() == 2 else k.shape[1]
    if positions.dtype != torch.int64 or cos_sin_cache.dtype != torch.float32:
        raise TypeError("positions must be int64 and cos_sin_cache float32")
    max_pos, rot_dim = cos_sin_cache.shape
    sec = None
    row_stride = 0
    if positions.dim() == 2:
        if mrope_section is None:
            raise ValueError("positions is [3, T]: mrope_section required")
        sec = (ctypes.c_int32 * 3)(*[int(x) for x in mrope_section])
        row_stride = positions.stride(0)
        if positions.stride(1) != 1:
            raise ValueError("positions rows must be contiguous")
    bs = cb = ct = ch = 0
    if k_cache is not None:
        cb, ct, ch = _cache_strides(k_cache, v_cache)
        bs = k_cache.shape[1]
        if slot_mapping is None or slot_mapping.dtype != torch.int64:
            raise ValueError("slot_mapping (int64) required with k_cache")
    check("mi_rope_neox_and_cache", lib.mi_rope_neox_and_cache(
        _p(q), _p(k), _p(v), _p(positions), row_stride,
        ctypes.cast(sec, ctypes.c_void_p) if sec is not None else None,
        _p(cos_sin_cache), max_pos, rot_dim, _p(q_norm_weight), _p(k_norm_weight), float(norm_eps),
        _p(k_cache), _p(v_cache), _p(slot_mapping), t, nq, nkv, head_dim, bs,
        q.stride(0), k.stride(0), v.stride(0) if v is not None else 0, cb, ct, ch,
        _dt(q.dtype), _stream()))


def embed_codes(ids: torch.Tensor, table: torch.Tensor, out_a: torch.Tensor, out_b: Optional[torch.Tensor] = None,
                ids_out: Optional[torch.Tensor] = None) -> None:
    """``out_a[t] = out_b[t] = table[ids[t]]``, ``ids_out[t] = ids[t]`` in one launch (``mi_embed_codes``): the code
    predictor's per-step glue.  ``out_*`` [T, E] row views (any row stride), ``ids_out`` int64 [T] (any stride)."""
    _gpu(ids, table, out_a, out_b, ids_out)
    if ids.dtype != torch.int64 or ids.dim() != 1 or not ids.is_contiguous():
        raise TypeError("ids must be contiguous int64 [T]")
    if table.dim() != 2 or not table.is_contiguous() or out_a.dtype != table.dtype or out_a.stride(-1) != 1:
        raise ValueError("table must be contiguous [V, E]; outputs rows of the same dtype")
    if ids_out is not None and (ids_out.dtype != torch.int64 or ids_out.dim() != 1):
        raise TypeError("ids_out must be int64 [T]")
    check("mi_embed_codes", lib.mi_embed_codes(
        _p(out_a), _p(out_b), _p(ids_out), _p(table), _p(ids), ids.shape[0], table.shape[1], table.shape[0],
        out_a.stride(0), out_b.stride(0) if out_b is not None else 0, ids_out.stride(0) if ids_out is not None else 0,
        _stream()))


def block_attention_supported(num_q_heads: int, num_kv_heads: int, head_dim: int, block_size: int, rot_dim: int) -> bool:
    return (head_dim == 128 and rot_dim == 128 and block_size <= 64 and num_q_heads % num_kv_heads == 0
            and num_q_heads // num_kv_heads in (1, 2, 4, 8))


def rope_cache_attend_block(out: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                            positions: Union[torch.Tensor, int], cos_sin_cache: torch.Tensor, head_dim: int,
                            k_cache: torch.Tensor, v_cache: torch.Tensor, block_ids: torch.Tensor, scale: float,
                            q_norm_weight: Optional[torch.Tensor] = None, k_norm_weight: Optional[torch.Tensor] = None,
                            norm_eps: float = 1e-6, max_position_bound: Optional[int] = None) -> torch.Tensor:
    """(q/k-norm ->) RoPE -> K/V append -> attention in ONE launch for sequences that live in one KV block each
    (``mi_rope_cache_attend_block``): row t's new token sits at ``positions[t]`` of block ``block_ids[t]`` and attends
    over tokens 0..positions[t] of it.  ``block_ids`` int32 [T] (any stride: a block-table column).
    ``positions``: int64 [T] on the device with ``max_position_bound`` >= every entry (default block_size - 1), or a
    host int when every row sits at the same position (the code predictor's lock-step sequences)."""
    _gpu(out, q, k, v, cos_sin_cache, k_cache, v_cache, block_ids, q_norm_weight, k_norm_weight)
    if isinstance(positions, int):
        bound, positions = positions, None
    else:
        _gpu(positions)
        if positions.dtype != torch.int64 or positions.dim() != 1:
            raise TypeError("positions must be int64 [T] (or a host int)")
        bound = k_cache.shape[1] - 1 if max_position_bound is None else int(max_position_bound)
    if cos_sin_cache.dtype != torch.float32:
        raise TypeError("cos_sin_cache must be float32")
    if block_ids.dtype != torch.int32 or block_ids.dim() != 1:
        raise TypeError("block_ids must be int32 [T]")
    t = q.shape[0]
    nq, nkv = q.shape[-1] // head_dim, k.shape[-1] // head_dim
    max_pos, rot_dim = cos_sin_cache.shape
    cb, ct, ch = _cache_strides(k_cache, v_cache)
    check("mi_rope_cache_attend_block", lib.mi_rope_cache_attend_block(
        _p(out), _p(q), _p(k), _p(v), _p(positions), bound, _p(cos_sin_cache), max_pos, rot_dim, _p(q_norm_weight),
        _p(k_norm_weight), float(norm_eps), _p(k_cache), _p(v_cache), _p(block_ids), block_ids.stride(0), t, nq, nkv,
        head_dim, k_cache.shape[1], out.stride(0), q.stride(0), k.stride(0), v.stride(0), cb, ct, ch, float(scale),
        _dt(q.dtype), _stream()))
    return out


def rope_cache_attend_block_splitk(part: "SplitKPartials", num_q_heads: int, num_kv_heads: int, head_dim: int,
                                   positions: Union[torch.Tensor, int], cos_sin_cache: torch.Tensor, k_cache: torch.Tensor,
                                   v_cache: torch.Tensor, block_ids: torch.Tensor, scale: float,
                                   q_norm_weight: Optional[torch.Tensor] = None,
                                   k_norm_weight: Optional[torch.Tensor] = None, norm_eps: float = 1e-6,
                                   max_position_bound: Optional[int] = None) -> torch.Tensor:
    """``rope_cache_attend_block`` fed by the split-K partials of ``qkv_proj`` (``linear_partials``): returns the
    attention output [T, nq*d]; bit-identical to ``linear`` + ``rope_cache_attend_block``."""
    _gpu(cos_sin_cache, k_cache, v_cache, block_ids, q_norm_weight, k_norm_weight)
    part._live()
    if part.n != (num_q_heads + 2 * num_kv_heads) * head_dim:
        raise ValueError("partials row is not [q heads | k heads | v heads]")
    if isinstance(positions, int):
        bound, positions = positions, None
    else:
        _gpu(positions)
        if positions.dtype != torch.int64 or positions.dim() != 1:
            raise TypeError("positions must be int64 [T] (or a host int)")
        bound = k_cache.shape[1] - 1 if max_position_bound is None else int(max_position_bound)
    if cos_sin_cache.dtype != torch.float32 or block_ids.dtype != torch.int32 or block_ids.dim() != 1:
        raise TypeError("cos_sin_cache must be float32, block_ids int32 [T]")
    max_pos, rot_dim = cos_sin_cache.shape
    cb, ct, ch = _cache_strides(k_cache, v_cache)
    out = torch.empty(part.m, num_q_heads * head_dim, dtype=part.dtype, device=part.device)
    check("mi_rope_cache_attend_block_splitk", lib.mi_rope_cache_attend_block_splitk(
        _p(out), _p(part.part), part.split, _p(part.bias), _p(positions), bound, _p(cos_sin_cache), max_pos, rot_dim,
        _p(q_norm_weight), _p(k_norm_weight), float(norm_eps), _p(k_cache), _p(v_cache), _p(block_ids), block_ids.stride(0),
        part.m, num_q_heads, num_kv_heads, head_dim, k_cache.shape[1], out.stride(0), cb, ct, ch, float(scale),
        _dt(part.dtype), _stream()))
    return out


class CpStack:
    """Host handle of the code predictor's one-launch layer stack (``mi_cp_stack_decode``, csrc/cp_stack.hip): the
    pointer tables of the layers' weights and private KV caches (the tensors are kept alive here) and the workspace
    that carries the activations between the launch's phases and its grid-barrier state.

    ``layers``: per layer a dict with ``qkv, o, gate_up, down, input_norm, post_norm, q_norm, k_norm`` (device tensors of
    the compute dtype) and ``k_cache, v_cache`` ([num_blocks, block_size, kv_heads, 128] views); ``max_rows``: the largest
    batch a call may bring."""

    _KEYS = ("qkv", "o", "gate_up", "down", "input_norm", "post_norm", "q_norm", "k_norm", "k_cache", "v_cache")

    def __init__(self, layers: Sequence[dict], final_norm: torch.Tensor, cos_sin_cache: torch.Tensor, num_q_heads: int,
                 num_kv_heads: int, head_dim: int, norm_eps: float, scale: float, max_rows: int):
        import ctypes

        first = layers[0]
        _gpu(final_norm, cos_sin_cache, *[t for layer in layers for t in (layer[k] for k in self._KEYS)])
        self.dtype = first["qkv"].dtype
        self.hidden = int(first["qkv"].shape[1])
        self.inter = int(first["down"].shape[1])
        self.nq, self.nkv, self.head_dim = int(num_q_heads), int(num_kv_heads), int(head_dim)
        self.eps, self.scale = float(norm_eps), float(scale)
        self.max_rows = int(max_rows)
        for layer in layers:
            if (tuple(layer["qkv"].shape) != ((self.nq + 2 * self.nkv) * self.head_dim, self.hidden)
                    or tuple(layer["o"].shape) != (self.hidden, self.nq * self.head_dim)
                    or tuple(layer["gate_up"].shape) != (2 * self.inter, self.hidden)
                    or tuple(layer["down"].shape) != (self.hidden, self.inter)):
                raise ValueError("layer weights do not have the stack's shapes")
            for k in self._KEYS[:8]:
                if layer[k].dtype != self.dtype or not layer[k].is_contiguous():
                    raise TypeError(f"{k}: contiguous {self.dtype} expected")
            if layer["k_cache"].stride() != first["k_cache"].stride() or layer["k_cache"].shape != first["k_cache"].shape:
                raise ValueError("the layers' KV caches must share one layout")
        self._cb, self._ct, self._ch = _cache_strides(first["k_cache"], first["v_cache"])
        self.block_size = int(first["k_cache"].shape[1])
        self._keep = (list(layers), final_norm, cos_sin_cache)
        n = len(layers)
        arr = ctypes.c_void_p * n
        self._tables = {k: arr(*[layer[k].data_ptr() for layer in layers]) for k in self._KEYS}
        self.num_layers = n
        self.final_norm, self.cos_sin = final_norm, cos_sin_cache
        nbytes = int(lib.mi_cp_stack_workspace_bytes(self.max_rows, self.hidden, self.nq, self.nkv, self.inter))
        if nbytes <= 0:
            raise ValueError("unsupported code-predictor stack geometry")
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=final_norm.device)
        self._ws_bytes = nbytes
        self._fresh = True

    def decode(self, out: torch.Tensor, x: torch.Tensor, block_ids: torch.Tensor, position: int) -> torch.Tensor:
        """One pass: x [rows, hidden] -> out [rows, hidden] (final-norm output), every row's new token at ``position`` of
        the KV block ``block_ids[row]``."""
        _gpu(out, x, block_ids)
        rows = int(x.shape[0])
        if rows > self.max_rows:
            raise ValueError(f"{rows} rows exceed the workspace's {self.max_rows}")
        if x.dtype != self.dtype or out.dtype != self.dtype or not x.is_contiguous() or not out.is_contiguous():
            raise TypeError("x / out must be contiguous tensors of the stack's dtype")
        if tuple(x.shape) != (rows, self.hidden) or tuple(out.shape) != (rows, self.hidden):
            raise ValueError("x / out must be [rows, hidden]")
        if block_ids.dtype != torch.int32 or block_ids.dim() != 1 or block_ids.shape[0] < rows:
            raise TypeError("block_ids must be int32 [rows]")
        t = self._tables
        check("mi_cp_stack_decode", lib.mi_cp_stack_decode(
            _p(out), _p(x), t["qkv"], t["o"], t["gate_up"], t["down"], t["input_norm"], t["post_norm"], t["q_norm"], t["k_norm"],
            _p(self.final_norm), t["k_cache"], t["v_cache"], _p(block_ids), block_ids.stride(0), _p(self.cos_sin),
            self.cos_sin.shape[0], int(position), rows, self.hidden, self.nq, self.nkv, self.head_dim, self.inter,
            self.num_layers, self.block_size, self._cb, self._ct, self._ch, self.eps, self.scale, _p(self.workspace),
            self._ws_bytes, 0, _dt(self.dtype), _stream()))   # the workspace was zeroed at allocation
        return out

    def status(self) -> int:
        """0 = every grid barrier of every launch so far completed (synchronises)."""
        return int(self.workspace[self._ws_bytes - 256:self._ws_bytes - 252].view(torch.int32).item())


# --------------------------------------------------------------------------- A5
# Default decode mode (see include/vllm_omni_amd.h, flags): True = P may be rounded to the cache dtype
# before P.V, as the reference's TRITON_ATTN kernel does -> matrix-core kernel where built (head_dim 128,
# block >= 16), dot2 VALU kernel elsewhere.  False = exact mode (P kept fp32, VALU kernel).
DECODE_PACKED_PV = True
DECODE_NONTEMPORAL = True   # nt cache policy on the K/V stream: +8 % measured (profiles/)


def _decode_flags(packed_pv: Optional[bool], force_valu: bool) -> int:
    return ((_C.MI_DECODE_PACKED_PV if (DECODE_PACKED_PV if packed_pv is None else packed_pv) else 0)
            | (_C.MI_DECODE_FORCE_VALU if force_valu else 0))


def decode_auto_partition(num_seqs: int, num_kv_heads: int, max_seq_len: int, block_size: int = 16,
                          requested: int = 0, packed_pv: Optional[bool] = None, force_valu: bool = False,
                          head_dim: int = 128) -> int:
    """Partition size a decode call will really use: ``requested``, or the auto choice — the fewest partitions that
    leave >= 512 waves (1024 at head_dim 64), ONE (direct output, no merge launch) once sequences x KV heads reach that;
    any multiple of 32 tokens for the matrix-core kernel, capped at 64 blocks for the VALU kernel (``packed_pv`` False /
    ``force_valu``)."""
    return int(lib.mi_paged_attn_decode_effective_partition(num_seqs, num_kv_heads, max_seq_len, head_dim, block_size,
                                                            requested, _decode_flags(packed_pv, force_valu)))


def decode_workspace_bytes(num_seqs: int, num_q_heads: int, head_dim: int, max_seq_len: int,
                           partition_size: int) -> int:
    return int(lib.mi_paged_attn_decode_workspace_bytes(num_seqs, num_q_heads, head_dim, max_seq_len,
                                                        partition_size))


# Workspaces whose counter region (the first MI_DECODE_COUNTER_BYTES) this module has zeroed: only on those does a
# decode call ask for the fused merge (the kernel leaves the counters at zero, so once is enough).  Keyed by the
# tensor OBJECT (a weak reference): a recycled address of a freed tensor can never pass for a prepared one.
_DECODE_WS_READY: dict = {}   # id(tensor) -> weakref(tensor)   (tensors compare element-wise: no WeakSet)


def _decode_ws_ready(ws: Optional[torch.Tensor]) -> bool:
    r = _DECODE_WS_READY.get(id(ws)) if ws is not None else None
    return r is not None and r() is ws


def new_decode_workspace(nbytes: int, device) -> torch.Tensor:
    """A decode workspace (``decode_workspace_bytes``) with its arrival-counter region zeroed: what
    ``paged_attn_decode`` needs to merge the context partitions inside the attention launch itself."""
    import weakref

    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    ws[:min(ws.numel(), _C.MI_DECODE_COUNTER_BYTES)].zero_()
    key = id(ws)
    _DECODE_WS_READY[key] = weakref.ref(ws, lambda _r, k=key: _DECODE_WS_READY.pop(k, None))
    return ws


# OFF by default: measured SLOWER than the separate merge kernel on every shape tried (profiles/r03c_decode_fused_ab.log:
# 391 against 363 us at B 256 / ctx 4096, 56 against 46 us at the 1/8-size shard shapes) — every wave now ends with a
# store-acknowledge wait and an atomic round trip while it still holds its LDS slot (5 waves per CU), which costs more
# streaming time than the 5-7 us merge launch it saves.  Kept (bit-identical, tested) for shapes with few, long waves.
DECODE_FUSED_MERGE = os.environ.get("VLLM_OMNI_AMD_DECODE_FUSED_MERGE", "0") == "1"
# RoPE + KV write of a pure-decode step inside the attention launch (rope_cache_paged_attn_decode); "0": two launches (A/B)
ROPE_DECODE_FUSION = os.environ.get("VLLM_OMNI_AMD_ROPE_DECODE_FUSION", "1") != "0"


def paged_attn_decode(out: torch.Tensor, q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                      block_table: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int, scale: float,
                      partition_size: int = 0, workspace: Optional[torch.Tensor] = None,
                      packed_pv: Optional[bool] = None, force_valu: bool = False,
                      nontemporal: Optional[bool] = None, rounded_p: bool = False,
                      fused_merge: Optional[bool] = None, _skip_merge: bool = False) -> torch.Tensor:
    """out[s, h] = softmax(q[s,h] . K^T * scale) . V over the paged cache (one query per sequence).

    q, out: [S, nq*d] (or [S, nq, d]); block_table int32 [S, max_blocks]; seq_lens int32 [S].
    ``max_seq_len`` is a host-side upper bound of seq_lens (sizes the launch).
    ``out`` may be fp32 for parity measurement.  ``packed_pv`` (None = module default
    ``DECODE_PACKED_PV``): P.V on 16-bit operands — the MFMA kernel (P carried as a hi + lo pair of
    cache-dtype values, inside the fp32-P tolerance) where it is built, the dot2 VALU form elsewhere;
    ``force_valu`` keeps the VALU kernel; ``rounded_p`` asks the MFMA kernel for the single-rounding
    P of flash kernels (A/B measurements only).  ``fused_merge`` (None = on for workspaces made by
    ``new_decode_workspace``): the context partitions are merged inside the attention launch by the last-arriving
    wave of each (sequence, head group) instead of a second kernel — same bits.  Returns ``out``.
    """
    _gpu(out, q, k_cache, v_cache, block_table, seq_lens, workspace)
    nb, bs, hkv, d = k_cache.shape
    s = q.shape[0]
    nq = q.numel() // (s * d) if s else 0
    if block_table.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise TypeError("block_table and seq_lens must be int32")
    if block_table.stride(1) != 1:
        raise ValueError("block_table rows must be contiguous")
    cb, ct, ch = _cache_strides(k_cache, v_cache)
    if s == 0:
        return out
    partition_size = decode_auto_partition(s, hkv, max_seq_len, bs, partition_size, packed_pv, force_valu, head_dim=d)
    need = decode_workspace_bytes(s, nq, d, max_seq_len, partition_size)
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = new_decode_workspace(need, q.device)
    ready = _decode_ws_ready(workspace)
    if fused_merge is None:
        fused_merge = DECODE_FUSED_MERGE and ready
    elif fused_merge and not ready:
        raise ValueError("fused_merge needs a workspace from ops.new_decode_workspace (zeroed arrival counters)")
    check("mi_paged_attn_decode", lib.mi_paged_attn_decode(
        _p(out), _p(q), _p(k_cache), _p(v_cache), _p(block_table), _p(seq_lens), s, nq, hkv, d, bs,
        block_table.stride(0), int(max_seq_len), float(scale), q.stride(0), out.stride(0), cb, ct, ch,
        int(partition_size), _p(workspace),
        workspace.numel() * workspace.element_size() if workspace is not None else 0,
        _dt(q.dtype), _dt(out.dtype),
        (_C.MI_DECODE_PACKED_PV if (DECODE_PACKED_PV if packed_pv is None else packed_pv) else 0)
        | (_C.MI_DECODE_FORCE_VALU if force_valu else 0)
        | (_C.MI_DECODE_ROUNDED_P if rounded_p else 0)
        | (_C.MI_DECODE_FUSED_MERGE if fused_merge else 0)
        | (_C.MI_DECODE_SKIP_MERGE if _skip_merge else 0)   # bench.py's kernel-only timing; never set by the product
        | (_C.MI_DECODE_NONTEMPORAL if (DECODE_NONTEMPORAL if nontemporal is None else nontemporal) else 0),
        _stream()))
    return out


def rope_decode_fusable(q: torch.Tensor, k_cache: torch.Tensor, positions: torch.Tensor, rot_dim: int,
                        partition_size: int, num_seqs: int, max_seq_len: int) -> bool:
    """Whether ``rope_cache_paged_attn_decode`` covers this call: the production decode kernel (matrix cores, nt
    streams, hi + lo P), full rotary, 1-D positions."""
    nb, bs, hkv, d = k_cache.shape
    if not (DECODE_PACKED_PV and DECODE_NONTEMPORAL and ROPE_DECODE_FUSION) or positions.dim() != 1 or rot_dim != d or bs < 16:
        return False
    ps = decode_auto_partition(num_seqs, hkv, max_seq_len, bs, partition_size, None, False, head_dim=d)
    return ps % 32 == 0


def rope_cache_paged_attn_decode(out: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                                 positions: torch.Tensor, cos_sin_cache: torch.Tensor, k_cache: torch.Tensor,
                                 v_cache: torch.Tensor, slot_mapping: torch.Tensor, block_table: torch.Tensor,
                                 seq_lens: torch.Tensor, max_seq_len: int, scale: float, partition_size: int = 0,
                                 workspace: Optional[torch.Tensor] = None, _skip_merge: bool = False,
                                 q_norm_weight: Optional[torch.Tensor] = None, k_norm_weight: Optional[torch.Tensor] = None,
                                 norm_eps: float = 1e-6) -> torch.Tensor:
    """``rope_neox_and_cache`` + ``paged_attn_decode`` of a pure-decode step in ONE launch (same bits): q is rotated in
    registers (the tensor keeps its un-rotated values), the token's rotated k and its v are written to the cache by the
    wave that owns the sequence's last context partition.  q [S, nq*d], k / v [S, nkv*d]: views of the qkv row."""
    _gpu(out, q, k, v, positions, cos_sin_cache, k_cache, v_cache, slot_mapping, block_table, seq_lens, workspace,
         q_norm_weight, k_norm_weight)
    if (q_norm_weight is None) != (k_norm_weight is None):
        raise ValueError("q_norm_weight and k_norm_weight come together")
    nb, bs, hkv, d = k_cache.shape
    s = q.shape[0]
    if s == 0:
        return out
    nq = q.numel() // (s * d)
    if block_table.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise TypeError("block_table and seq_lens must be int32")
    if positions.dtype != torch.int64 or slot_mapping.dtype != torch.int64 or cos_sin_cache.dtype != torch.float32:
        raise TypeError("positions / slot_mapping must be int64 and cos_sin_cache float32")
    if positions.dim() != 1 or block_table.stride(1) != 1:
        raise ValueError("1-D positions and contiguous block_table rows required")
    max_pos, rot_dim = cos_sin_cache.shape
    cb, ct, ch = _cache_strides(k_cache, v_cache)
    partition_size = decode_auto_partition(s, hkv, max_seq_len, bs, partition_size, None, False, head_dim=d)
    need = decode_workspace_bytes(s, nq, d, max_seq_len, partition_size)
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = new_decode_workspace(need, q.device)
    fused_merge = DECODE_FUSED_MERGE and _decode_ws_ready(workspace)
    check("mi_rope_cache_paged_attn_decode", lib.mi_rope_cache_paged_attn_decode(
        _p(out), _p(q), _p(k), _p(v), _p(positions), _p(cos_sin_cache), max_pos, rot_dim, _p(q_norm_weight),
        _p(k_norm_weight), float(norm_eps), _p(k_cache), _p(v_cache),
        _p(slot_mapping), _p(block_table), _p(seq_lens), s, nq, hkv, d, bs, block_table.stride(0), int(max_seq_len),
        float(scale), q.stride(0), k.stride(0), v.stride(0), out.stride(0), cb, ct, ch, int(partition_size), _p(workspace),
        workspace.numel() * workspace.element_size() if workspace is not None else 0, _dt(q.dtype), _dt(out.dtype),
        _C.MI_DECODE_PACKED_PV | _C.MI_DECODE_NONTEMPORAL | (_C.MI_DECODE_FUSED_MERGE if fused_merge else 0)
        | (_C.MI_DECODE_SKIP_MERGE if _skip_merge else 0), _stream()))
    return out


# --------------------------------------------------------------------------- A6
_PREFILL_WS: dict = {}   # device -> grow-only fp32 workspace of the split-context prefill (never captured in a graph)


def paged_attn_prefill(out: torch.Tensor, q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                       block_table: torch.Tensor, seq_lens: torch.Tensor, query_start_loc: torch.Tensor,
                       max_query_len: int, scale: float, max_seq_len: Optional[int] = None,
                       segments: int = 0) -> torch.Tensor:
    """Varlen causal attention of query chunks against the paged cache (chunked prefill).

    ``max_seq_len`` (host upper bound of ``seq_lens``) lets a launch that would leave most CUs idle — the ragged last
    chunk of a long prompt, a TP shard with one KV head — split every tile's context into segments run by separate
    workgroups and merged afterwards (``mi_paged_attn_prefill_split``).  Without it nothing is split.
    ``segments``: 0 automatic, 1 never, n > 1 force n (tests)."""
    _gpu(out, q, k_cache, v_cache, block_table, seq_lens, query_start_loc)
    nb, bs, hkv, d = k_cache.shape
    t = q.shape[0]
    nq = q.numel() // (t * d) if t else 0
    s = seq_lens.shape[0]
    cb, ct, ch = _cache_strides(k_cache, v_cache)
    if t == 0 or s == 0:
        return out
    ws, ws_bytes, msl = None, 0, 0
    if max_seq_len is not None and segments != 1:
        msl = int(max_seq_len)
        ws_bytes = int(lib.mi_paged_attn_prefill_workspace_bytes(t, s, int(max_query_len), msl, nq, hkv, d, int(segments)))
        if ws_bytes:
            ws = _PREFILL_WS.get(q.device)
            if ws is None or ws.numel() * 4 < ws_bytes:
                ws = _PREFILL_WS[q.device] = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=q.device)
    check("mi_paged_attn_prefill_split", lib.mi_paged_attn_prefill_split(
        _p(out), _p(q), _p(k_cache), _p(v_cache), _p(block_table), _p(seq_lens), _p(query_start_loc),
        s, t, int(max_query_len), nq, hkv, d, bs, block_table.stride(0), float(scale),
        q.stride(0), out.stride(0), cb, ct, ch, _dt(q.dtype), msl, int(segments) if ws is not None else 1,
        _p(ws) if ws is not None else None, ws.numel() * 4 if ws is not None else 0, _stream()))
    return out


# --------------------------------------------------------------------------- A8
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _gpu(x, weight, out)
    x2 = x.view(-1, x.shape[-1])
    if out is None:
        out = torch.empty_like(x)
    o2 = out.view(-1, x.shape[-1])
    check("mi_rms_norm", lib.mi_rms_norm(_p(o2), _p(x2), _p(weight), float(eps), x2.shape[0], x2.shape[1],
                                          x2.stride(0), o2.stride(0), _dt(x.dtype), _stream()))
    return out


def fused_add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> None:
    """In place: residual <- x + residual; x <- rmsnorm(residual) * weight."""
    _gpu(x, residual, weight)
    x2 = x.view(-1, x.shape[-1])
    r2 = residual.view(-1, x.shape[-1])
    check("mi_fused_add_rms_norm", lib.mi_fused_add_rms_norm(
        _p(x2), _p(r2), _p(weight), float(eps), x2.shape[0], x2.shape[1], x2.stride(0), r2.stride(0),
        _dt(x.dtype), _stream()))


# --------------------------------------------------------------------------- A9
def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _gpu(x, out)
    inter = x.shape[-1] // 2
    x2 = x.view(-1, x.shape[-1])
    if out is None:
        out = torch.empty(*x.shape[:-1], inter, dtype=x.dtype, device=x.device)
    o2 = out.view(-1, inter)
    check("mi_silu_and_mul", lib.mi_silu_and_mul(_p(o2), _p(x2), x2.shape[0], inter, x2.stride(0),
                                                  o2.stride(0), _dt(x.dtype), _stream()))
    return out


# -------------------------------------------------------------------------- A11
def mask_logits_argmax(logits: torch.Tensor, ranges: Sequence[tuple[int, int]] = (), col: int = -1,
                       ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Set up to two [lo, hi) column ranges and one column to -1e9 in place; greedy argmax."""
    _gpu(logits, ids)
    if len(ranges) > 2:
        raise ValueError("at most two masked ranges")
    r = list(ranges) + [(0, 0)] * (2 - len(ranges))
    t, v = logits.shape
    if ids is None:
        ids = torch.empty(t, dtype=torch.int64, device=logits.device)
    check("mi_mask_logits_argmax", lib.mi_mask_logits_argmax(
        _p(logits), _p(ids), t, v, logits.stride(0), r[0][0], r[0][1], r[1][0], r[1][1], int(col),
        _dt(logits.dtype), _stream()))
    return ids


def sample_tokens(logits: torch.Tensor, temperature: torch.Tensor, seed: torch.Tensor, offset: torch.Tensor,
                  top_k: Optional[torch.Tensor] = None, top_p: Optional[torch.Tensor] = None,
                  rep_penalty: Optional[torch.Tensor] = None, seen_bitmap: Optional[torch.Tensor] = None,
                  seen_row: Optional[torch.Tensor] = None, update_seen: bool = True,
                  ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Penalty -> temperature -> top-k -> top-p -> exponential-race draw, one launch (N2).

    temperature/top_p/rep_penalty fp32 [rows]; top_k/seen_row int32 [rows]; seed/offset int64 [rows] (bit
    patterns of the uint64 Philox key and draw counter); seen_bitmap int32 [state rows, words]."""
    _gpu(logits, temperature, seed, offset, top_k, top_p, rep_penalty, seen_bitmap, seen_row, ids)
    t, v = logits.shape
    if temperature.dtype != torch.float32 or seed.dtype != torch.int64 or offset.dtype != torch.int64:
        raise TypeError("temperature must be float32, seed/offset int64")
    for name, x, dt in (("top_k", top_k, torch.int32), ("top_p", top_p, torch.float32),
                        ("rep_penalty", rep_penalty, torch.float32), ("seen_bitmap", seen_bitmap, torch.int32),
                        ("seen_row", seen_row, torch.int32)):
        if x is not None and x.dtype != dt:
            raise TypeError(f"{name} must be {dt}")
    if ids is None:
        ids = torch.empty(t, dtype=torch.int64, device=logits.device)
    check("mi_sample_tokens", lib.mi_sample_tokens(
        _p(ids), _p(logits), t, v, logits.stride(0), _dt(logits.dtype), _p(temperature), _p(top_k), _p(top_p),
        _p(rep_penalty), _p(seen_bitmap), _p(seen_row), seen_bitmap.shape[1] if seen_bitmap is not None else 0,
        1 if update_seen else 0, _p(seed), _p(offset), _stream()))
    return ids


def sample_tokens_embed(logits: torch.Tensor, temperature: torch.Tensor, seed: torch.Tensor, offset: torch.Tensor,
                        top_k: Optional[torch.Tensor], top_p: Optional[torch.Tensor], table: torch.Tensor,
                        out_a: torch.Tensor, out_b: Optional[torch.Tensor] = None, ids_out: Optional[torch.Tensor] = None,
                        ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``sample_tokens`` (no repetition penalty) + ``embed_codes`` in one launch: the drawn id's row of ``table`` goes to
    ``out_a[r]`` (and ``out_b[r]``), the id also to ``ids_out[r]`` (any stride).  The code predictor's per-step draw."""
    _gpu(logits, temperature, seed, offset, top_k, top_p, table, out_a, out_b, ids_out, ids)
    t, v = logits.shape
    if temperature.dtype != torch.float32 or seed.dtype != torch.int64 or offset.dtype != torch.int64:
        raise TypeError("temperature must be float32, seed/offset int64")
    if (top_k is not None and top_k.dtype != torch.int32) or (top_p is not None and top_p.dtype != torch.float32):
        raise TypeError("top_k must be int32, top_p float32")
    if table.dim() != 2 or table.shape[0] < v or not table.is_contiguous() or table.element_size() != 2:
        raise ValueError("table must be a contiguous 16-bit [>= vocab, dim] tensor")
    dim = table.shape[1]
    for o in (out_a, out_b):
        if o is not None and (tuple(o.shape) != (t, dim) or o.dtype != table.dtype or o.stride(1) != 1):
            raise ValueError("out_a / out_b must be [rows, dim] of the table's dtype with contiguous rows")
    if ids_out is not None and (ids_out.dtype != torch.int64 or ids_out.dim() != 1 or ids_out.shape[0] != t):
        raise TypeError("ids_out must be int64 [rows]")
    if ids is None:
        ids = torch.empty(t, dtype=torch.int64, device=logits.device)
    check("mi_sample_tokens_embed", lib.mi_sample_tokens_embed(
        _p(ids), _p(logits), t, v, logits.stride(0), _dt(logits.dtype), _p(temperature), _p(top_k), _p(top_p), _p(seed),
        _p(offset), _p(table), dim, _p(out_a), out_a.stride(0), _p(out_b), out_b.stride(0) if out_b is not None else 0,
        _p(ids_out), ids_out.stride(0) if ids_out is not None else 0, _stream()))
    return ids


def talker_decode_embeds(out: torch.Tensor, embed_table: torch.Tensor, token_ids: torch.Tensor,
                         reply_rows: Optional[torch.Tensor] = None, reply_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[t] = embed_table[token_ids[t]] (+ reply_rows[reply_idx[t]] where reply_idx[t] >= 0)."""
    _gpu(out, embed_table, token_ids, reply_rows, reply_idx)
    t = token_ids.shape[0]
    if token_ids.dtype != torch.int64 or (reply_idx is not None and reply_idx.dtype != torch.int64):
        raise TypeError("token_ids / reply_idx must be int64")
    check("mi_talker_decode_embeds", lib.mi_talker_decode_embeds(
        _p(out), _p(embed_table), _p(token_ids), _p(reply_rows), _p(reply_idx), t, embed_table.shape[1],
        embed_table.shape[0], reply_rows.shape[0] if reply_rows is not None else 0, out.stride(0),
        _dt(embed_table.dtype), _stream()))
    return out


# -------------------------------------------------------------------------- A13
def dynamic_int8_quant(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Dynamic symmetric per-row int8 quantisation (int8_config.py:323-342): (int8 [rows, hidden], fp32 scale [rows])."""
    _gpu(x)
    x2 = x.view(-1, x.shape[-1])
    xq = torch.empty(x2.shape, dtype=torch.int8, device=x.device)
    xs = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    rc = lib.mi_int8_quant_rows(_p(xq), _p(xs), _p(x2), x2.shape[0], x2.shape[1], x2.stride(0), _dt(x.dtype), _stream())
    if rc == _C.MI_ERR_UNSUPPORTED:   # odd widths / alignments: the scalar kernel
        rc = lib.mi_dynamic_int8_quant(_p(xq), _p(xs), _p(x2), x2.shape[0], x2.shape[1], x2.stride(0), _dt(x.dtype),
                                       _stream())
    check("mi_dynamic_int8_quant", rc)
    return xq, xs


class QuantizedActs:
    """A W8A8 activation on its way into a linear layer: int8 rows + one fp32 scale per row, produced by the op in
    front of the linear (``rms_norm_int8``) instead of a separate quantisation launch."""

    __slots__ = ("xq", "xs", "lead", "dtype")

    def __init__(self, xq: torch.Tensor, xs: torch.Tensor, lead: tuple, dtype: torch.dtype):
        self.xq, self.xs, self.lead, self.dtype = xq, xs, lead, dtype


def rms_norm_int8(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None) -> QuantizedActs:
    """RMSNorm (``residual`` None) or fused residual-add RMSNorm (``residual`` updated in place; ``x`` untouched) whose
    output leaves as the W8A8 activation — same bits as the norm followed by ``dynamic_int8_quant``, one launch."""
    _gpu(x, weight, residual)
    x2 = x.view(-1, x.shape[-1])
    r2 = residual.view(-1, x.shape[-1]) if residual is not None else None
    xq = torch.empty(x2.shape, dtype=torch.int8, device=x.device)
    xs = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    check("mi_rms_norm_int8", lib.mi_rms_norm_int8(_p(xq), _p(xs), _p(x2), _p(r2), _p(weight), float(eps), x2.shape[0],
                                                   x2.shape[1], x2.stride(0), r2.stride(0) if r2 is not None else 0,
                                                   _dt(x.dtype), _stream()))
    return QuantizedActs(xq, xs, tuple(x.shape[:-1]), x.dtype)


def int8_gemm_dequant(xq: torch.Tensor, x_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor,
                      bias: Optional[torch.Tensor], out_dtype: torch.dtype) -> torch.Tensor:
    _gpu(xq, x_scale, wq, w_scale, bias)
    m, k = xq.shape
    n = wq.shape[0]
    out = torch.empty(m, n, dtype=out_dtype, device=xq.device)
    check("mi_int8_gemm_dequant", lib.mi_int8_gemm_dequant(
        _p(out), _p(xq), _p(x_scale), _p(wq), _p(w_scale), _p(bias), m, n, k, out.stride(0),
        _dt(out_dtype), _stream()))
    return out


def int8_dequant(acc: torch.Tensor, x_scale: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor],
                 out_dtype: torch.dtype) -> torch.Tensor:
    _gpu(acc, x_scale, w_scale, bias)
    m, n = acc.shape
    out = torch.empty(m, n, dtype=out_dtype, device=acc.device)
    check("mi_int8_dequant", lib.mi_int8_dequant(_p(out), _p(acc), _p(x_scale), _p(w_scale), _p(bias), m, n,
                                                  acc.stride(0), out.stride(0), _dt(out_dtype), _stream()))
    return out


_LINEAR_WS: dict = {}
_LINEAR_CNT: dict = {}   # device -> zeroed arrival counters of the in-launch split-K reduce (every launch leaves them zero)
# "1": split-K linears reduce inside their launch (mi_linear_fused_reduce: the last-arriving slice sums the slabs, same
# bits).  Built, tested, and measured SLOWER than the separate reduce launch at the shard shapes (down_proj at 128 rows
# x TP 4: 19.5 against 18.7 us; o_proj 11.1 / 10.7 — the last arriver's write-through, counter round trip and slab
# re-read cost more than the ~3.8 us launch they replace: profiles/r04i_fused_reduce.log), so it is opt-in.
LINEAR_FUSED_REDUCE = os.environ.get("VLLM_OMNI_AMD_LINEAR_FUSED_REDUCE", "0") == "1"
_LINEAR_WS_RETIRED: list = []   # outgrown workspaces stay allocated: a captured hipGraph may still write to them


def linear_supported(m: int, n: int, k: int, dtype: torch.dtype) -> bool:
    """Shape envelope of ``mi_linear`` (decode batches): rows <= 512 (two row blocks of 256 above 256), k a multiple
    of 128, n of 16."""
    return 0 < m <= 512 and k % 128 == 0 and n % 16 == 0 and dtype in (torch.bfloat16, torch.float16)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x W^T (+ b) for a decode batch (<= 512 rows) by the weight-streaming kernel of csrc/skinny_gemm.hip;
    stands in for the library GEMM behind vLLM's ColumnParallelLinear / RowParallelLinear on the decode path."""
    _gpu(x, weight, bias)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    m, k = x2.shape
    n = weight.shape[0]
    if weight.shape[1] != k or not weight.is_contiguous() or weight.dtype != x.dtype:
        raise ValueError("weight must be a contiguous [n, k] tensor of x's dtype")
    out = torch.empty(m, n, dtype=x.dtype, device=x.device)
    need = lib.mi_linear_workspace_bytes(m, n, k)
    ws = _linear_workspace(x.device, need) if need else None
    if need and LINEAR_FUSED_REDUCE:   # split-K: the reduce rides in the launch (last-arriving slice), same bits
        cnt = _LINEAR_CNT.get(x.device)
        if cnt is None:
            cnt = _LINEAR_CNT[x.device] = torch.zeros(_C.MI_LINEAR_COUNTER_BYTES // 4, dtype=torch.int32, device=x.device)
        check("mi_linear_fused_reduce", lib.mi_linear_fused_reduce(
            _p(out), _p(x2), _p(weight), _p(bias), m, n, k, x2.stride(0), out.stride(0), _p(ws), ws.numel(), _p(cnt),
            cnt.numel() * 4, _dt(x.dtype), _stream()))
        return out.view(*x.shape[:-1], n)
    check("mi_linear", lib.mi_linear(_p(out), _p(x2), _p(weight), _p(bias), m, n, k, x2.stride(0), out.stride(0),
                                     _p(ws) if need else None, ws.numel() if need else 0, _dt(x.dtype), _stream()))
    return out.view(*x.shape[:-1], n)


class SplitKPartials:
    """fp32 partials [split, m, n] of a split-K ``linear`` whose reduction has been left to the consumer: the fused
    residual-add + RMSNorm that follows (``fused_add_rms_norm_splitk``) or ``materialize()``.  The partials live in
    the shared workspace: they must be consumed before the next split-K linear on this device (checked)."""

    __slots__ = ("part", "split", "m", "n", "bias", "dtype", "lead", "device", "_gen")

    def __init__(self, part, split, m, n, bias, dtype, lead, device, gen):
        self.part, self.split, self.m, self.n, self.bias, self.dtype = part, split, m, n, bias, dtype
        self.lead, self.device, self._gen = lead, device, gen

    def _live(self) -> None:
        if _LINEAR_GEN.get(self.device) != self._gen:
            raise RuntimeError("split-K partials were overwritten by a later linear before they were reduced")

    def materialize(self) -> torch.Tensor:
        self._live()
        out = torch.empty(self.m, self.n, dtype=self.dtype, device=self.device)
        check("mi_linear_reduce", lib.mi_linear_reduce(_p(out), _p(self.part), self.split, _p(self.bias), self.m, self.n,
                                                       out.stride(0), _dt(self.dtype), _stream()))
        return out.view(*self.lead, self.n)


_LINEAR_GEN: dict = {}


def _linear_workspace(device, need: int) -> torch.Tensor:
    ws = _LINEAR_WS.get(device)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _LINEAR_WS_RETIRED.append(ws)
        ws = _LINEAR_WS[device] = torch.empty(need, dtype=torch.uint8, device=device)
    _LINEAR_GEN[device] = _LINEAR_GEN.get(device, 0) + 1
    return ws


def linear_partials(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> Optional[SplitKPartials]:
    """The streaming kernel of ``linear`` without its reduce launch, or None when this shape's plan has no split-K
    (nothing to defer).  Reduce with ``fused_add_rms_norm_splitk`` or ``.materialize()``."""
    _gpu(x, weight, bias)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    m, k = x2.shape
    n = weight.shape[0]
    if weight.shape[1] != k or not weight.is_contiguous() or weight.dtype != x.dtype:
        raise ValueError("weight must be a contiguous [n, k] tensor of x's dtype")
    need = lib.mi_linear_workspace_bytes(m, n, k)
    if not need:
        return None
    ws = _linear_workspace(x.device, need)
    split = ctypes.c_int32(0)
    check("mi_linear_partials", lib.mi_linear_partials(_p(x2), _p(weight), m, n, k, x2.stride(0), _p(ws), ws.numel(),
                                                       ctypes.byref(split), _dt(x.dtype), _stream()))
    return SplitKPartials(ws, split.value, m, n, bias, x.dtype, tuple(x.shape[:-1]), x.device, _LINEAR_GEN[x.device])


def fused_add_rms_norm_splitk(p: SplitKPartials, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """residual <- rnd(sum of the partials (+ bias)) + residual (in place); returns rmsnorm(residual) * weight — the
    reduce of a split-K ``linear`` folded into the fused-add RMSNorm behind it, bit-identical to the two apart."""
    _gpu(residual, weight)
    p._live()
    r2 = residual.view(-1, residual.shape[-1])
    if r2.shape != (p.m, p.n):
        raise ValueError("residual shape does not match the partials")
    out = torch.empty(p.m, p.n, dtype=p.dtype, device=p.device)
    check("mi_fused_add_rms_norm_splitk", lib.mi_fused_add_rms_norm_splitk(
        _p(out), _p(p.part), p.split, _p(p.bias), _p(r2), _p(weight), float(eps), p.m, p.n, out.stride(0), r2.stride(0),
        _dt(p.dtype), _stream()))
    return out.view(*p.lead, p.n)


def rope_neox_and_cache_splitk(part: SplitKPartials, num_q_heads: int, num_kv_heads: int, head_dim: int,
                               positions: torch.Tensor, cos_sin_cache: torch.Tensor, k_cache: torch.Tensor,
                               v_cache: torch.Tensor, slot_mapping: torch.Tensor,
                               mrope_section: Optional[torch.Tensor] = None,
                               q_norm_weight: Optional[torch.Tensor] = None,
                               k_norm_weight: Optional[torch.Tensor] = None, norm_eps: float = 1e-6) -> torch.Tensor:
    """``rope_neox_and_cache`` fed by the split-K partials of ``qkv_proj`` (``linear_partials``): returns the rotated
    q [T, nq*d]; k and v go to the cache only.  Bit-identical to ``linear`` + ``rope_neox_and_cache``."""
    _gpu(positions, cos_sin_cache, k_cache, v_cache, slot_mapping, q_norm_weight, k_norm_weight)
    part._live()
    if part.n != (num_q_heads + 2 * num_kv_heads) * head_dim:
        raise ValueError("partials row is not [q heads | k heads | v heads]")
    if positions.dtype != torch.int64 or cos_sin_cache.dtype != torch.float32 or slot_mapping.dtype != torch.int64:
        raise TypeError("positions / slot_mapping must be int64 and cos_sin_cache float32")
    max_pos, rot_dim = cos_sin_cache.shape
    sec = None
    row_stride = 0
    if positions.dim() == 2:
        if mrope_section is None:
            raise ValueError("positions is [3, T]: mrope_section required")
        sec = (ctypes.c_int32 * 3)(*[int(x) for x in mrope_section])
        row_stride = positions.stride(0)
        if positions.stride(1) != 1:
            raise ValueError("positions rows must be contiguous")
    cb, ct, ch = _cache_strides(k_cache, v_cache)
    q_out = torch.empty(part.m, num_q_heads * head_dim, dtype=part.dtype, device=part.device)
    check("mi_rope_neox_and_cache_splitk", lib.mi_rope_neox_and_cache_splitk(
        _p(q_out), _p(part.part), part.split, _p(part.bias), _p(positions), row_stride,
        ctypes.cast(sec, ctypes.c_void_p) if sec is not None else None, _p(cos_sin_cache), max_pos, rot_dim,
        _p(q_norm_weight), _p(k_norm_weight), float(norm_eps), _p(k_cache), _p(v_cache), _p(slot_mapping), part.m,
        num_q_heads, num_kv_heads, head_dim, k_cache.shape[1], q_out.stride(0), cb, ct, ch, _dt(part.dtype), _stream()))
    return q_out


def linear_silu_mul(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """silu(x Wg^T) * (x Wu^T) for a decode batch, weight = [gate rows; up rows] — ``gate_up_proj`` and the
    ``SiluAndMul`` behind it (qwen2_old.py:65-68) in one launch; the [m, 2 I] intermediate never reaches memory."""
    _gpu(x, weight)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    m, k = x2.shape
    if weight.shape[0] % 2 or weight.shape[1] != k or not weight.is_contiguous() or weight.dtype != x.dtype:
        raise ValueError("weight must be a contiguous [2 * inter, k] tensor of x's dtype")
    inter = weight.shape[0] // 2
    out = torch.empty(m, inter, dtype=x.dtype, device=x.device)
    check("mi_linear_silu_mul", lib.mi_linear_silu_mul(_p(out), _p(x2), _p(weight), m, inter, k, x2.stride(0),
                                                       out.stride(0), _dt(x.dtype), _stream()))
    return out.view(*x.shape[:-1], inter)



INT8_TILE_ROWS = 512   # rows one launch of the tiled W8A8 kernel takes


def int8_linear_supported(m: int, n: int, k: int) -> bool:
    """Shape envelope of ONE launch of the tiled W8A8 kernel (csrc/skinny_gemm.hip on int8 rows)."""
    return 0 < m <= INT8_TILE_ROWS and k % 128 == 0 and n % 16 == 0 and k > 0


def int8_linear(x: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """W8A8 linear of the reference's scheme (vllm_omni/quantization/int8_config.py:323-342: dynamic symmetric
    per-token activation scale, per-output-channel weight scale): hand-written per-token quantisation kernel, then
    ONE launch of the tiled int8 kernel — LDS-DMA ring, `v_mfma_i32_16x16x64_i8`, dequantisation + bias in the
    epilogue (``mi_int8_linear``; split-K shapes add its small reduce launch).  Prefill chunks (more than 512 rows)
    take the library contraction with the in-tree dequantisation epilogue (``torch._int_mm`` + ``mi_int8_dequant``),
    tiny weights the simple MFMA kernel ``mi_int8_gemm_dequant`` — the measured policy is spelled out below;
    ``VLLM_OMNI_AMD_INT8`` = ``library`` / ``simple`` / ``tiled`` forces one path (cross-checks, A/B runs).  Integer
    sums are exact on every path, so all three agree bit for bit."""
    if isinstance(x, QuantizedActs):
        xq, xs, lead, dtype = x.xq, x.xs, x.lead, x.dtype
    else:
        xq, xs = dynamic_int8_quant(x.reshape(-1, x.shape[-1]))
        lead, dtype = tuple(x.shape[:-1]), x.dtype
    m, k = xq.shape
    n = wq.shape[0]
    mode = os.environ.get("VLLM_OMNI_AMD_INT8", "kernel")
    lib_ok = m > 16 and k % 8 == 0 and n % 8 == 0
    tiled_ok = int8_linear_supported(min(m, INT8_TILE_ROWS), n, k)
    # Policy from profiles/r03_int8_linear_bench.log (all paths give the same bits, so this is speed only):
    #   <= 512 rows: the tiled kernel (24-48 us against 27-86 library, 33-263 simple at the S-A shapes, 256 rows) —
    #      except weights of <= 2 M elements at >= 256 rows (the S-B attention / expert shapes), where the simple MFMA
    #      kernel has less fixed cost (12.7 against 16.6 us);
    #   more rows (prefill chunks): the library contraction + the in-tree dequantisation epilogue (60 / 163 us against
    #      150 / 307 us for the tiled kernel run in 512-row blocks and 252 / 931 us for the simple kernel at 2048 rows);
    #      the tiled kernel in row blocks where the library cannot take the shape; the simple kernel last.
    if mode == "library" and lib_ok:
        y = int8_dequant(torch._int_mm(xq, wq.t()), xs, w_scale, bias, dtype)
    elif mode == "simple":
        y = int8_gemm_dequant(xq, xs, wq, w_scale, bias, dtype)
    elif m <= INT8_TILE_ROWS and tiled_ok and not (m >= 256 and n * k <= (1 << 21) and mode != "tiled"):
        y = int8_linear_q(xq, xs, wq, w_scale, bias, dtype)
    elif m > INT8_TILE_ROWS and lib_ok and mode != "tiled":
        y = int8_dequant(torch._int_mm(xq, wq.t()), xs, w_scale, bias, dtype)
    elif m > INT8_TILE_ROWS and tiled_ok:
        y = torch.empty(m, n, dtype=dtype, device=xq.device)   # rows of a GEMM are independent: same bits as one launch
        for r0 in range(0, m, INT8_TILE_ROWS):
            r1 = min(r0 + INT8_TILE_ROWS, m)
            int8_linear_q(xq[r0:r1], xs[r0:r1], wq, w_scale, bias, dtype, out=y[r0:r1])
    else:
        y = int8_gemm_dequant(xq, xs, wq, w_scale, bias, dtype)
    return y.view(*lead, n)


def int8_linear_q(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor,
                  bias: Optional[torch.Tensor], out_dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The tiled W8A8 kernel on already-quantised activations (xq int8 [m, k], xs fp32 [m])."""
    _gpu(xq, xs, wq, w_scale, bias, out)
    m, k = xq.shape
    n = wq.shape[0]
    if wq.shape[1] != k or not wq.is_contiguous() or xq.stride(1) != 1:
        raise ValueError("wq must be a contiguous int8 [n, k] tensor; xq rows must be contiguous")
    if out is None:
        out = torch.empty(m, n, dtype=out_dtype, device=xq.device)
    elif out.shape != (m, n) or out.dtype != out_dtype or out.stride(1) != 1:
        raise ValueError("out must be an [m, n] tensor of out_dtype with contiguous rows")
    need = lib.mi_int8_linear_workspace_bytes(m, n, k)
    ws = _linear_workspace(xq.device, need) if need else None
    check("mi_int8_linear", lib.mi_int8_linear(_p(out), _p(xq), _p(xs), _p(wq), _p(w_scale), _p(bias), m, n, k,
                                               xq.stride(0), out.stride(0), _p(ws) if need else None,
                                               ws.numel() if need else 0, _dt(out_dtype), _stream()))
    return out


class Int8SplitKPartials:
    """Exact int32 partials [split, m, n] of a split-K W8A8 projection whose reduction + dequantisation has been left
    to the fused-add RMSNorm behind it (``fused_add_rms_norm_splitk_int8``) or ``materialize()``; same workspace
    discipline as ``SplitKPartials``."""

    __slots__ = ("part", "split", "m", "n", "xs", "ws", "bias", "dtype", "lead", "device", "_gen")

    def __init__(self, part, split, m, n, xs, ws, bias, dtype, lead, device, gen):
        self.part, self.split, self.m, self.n, self.xs, self.ws = part, split, m, n, xs, ws
        self.bias, self.dtype, self.lead, self.device, self._gen = bias, dtype, lead, device, gen

    def _live(self) -> None:
        if _LINEAR_GEN.get(self.device) != self._gen:
            raise RuntimeError("split-K partials were overwritten by a later linear before they were reduced")

    def materialize(self) -> torch.Tensor:
        """The 16-bit projection output (what mi_int8_linear would have returned): reduce + dequant on the device."""
        self._live()
        z = torch.zeros(self.m, self.n, dtype=self.dtype, device=self.device)      # residual = 0: x + 0 = x exactly
        one = torch.ones(self.n, dtype=self.dtype, device=self.device)
        check("mi_fused_add_rms_norm_splitk_int8", lib.mi_fused_add_rms_norm_splitk_int8(
            _p(torch.empty_like(z)), None, _p(self.part), self.split, _p(self.xs), _p(self.ws), _p(self.bias), _p(z),
            _p(one), 1e-6, self.m, self.n, z.stride(0), z.stride(0), _dt(self.dtype), _stream()))
        return z.view(*self.lead, self.n)


def int8_linear_partials(x, wq: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor]) -> Optional[Int8SplitKPartials]:
    """The split-K W8A8 kernel without its reduce launch, or None when this shape has no split-K plan (decided BEFORE
    the activation is quantised: the caller's fallback quantises it itself)."""
    n, k = wq.shape
    m = x.xq.shape[0] if isinstance(x, QuantizedActs) else x.numel() // x.shape[-1]
    if not int8_linear_supported(m, n, k):
        return None
    need = lib.mi_int8_linear_workspace_bytes(m, n, k)
    if not need:
        return None
    if isinstance(x, QuantizedActs):
        xq, xs, lead, dtype = x.xq, x.xs, x.lead, x.dtype
    else:
        xq, xs = dynamic_int8_quant(x.reshape(-1, x.shape[-1]))
        lead, dtype = tuple(x.shape[:-1]), x.dtype
    ws = _linear_workspace(xq.device, need)
    split = ctypes.c_int32(0)
    check("mi_int8_linear_partials", lib.mi_int8_linear_partials(_p(xq), _p(wq), m, n, k, xq.stride(0), _p(ws), ws.numel(),
                                                                 ctypes.byref(split), _stream()))
    return Int8SplitKPartials(ws, split.value, m, n, xs, w_scale, bias, dtype, lead, xq.device, _LINEAR_GEN[xq.device])


def fused_add_rms_norm_splitk_int8(p: Int8SplitKPartials, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                                   quant: bool = False):
    """residual <- rnd(dequant(sum of the int32 partials) (+ bias)) + residual (in place); returns
    rmsnorm(residual) * weight — as a 16-bit tensor, or (quant) as the next W8A8 linear's ``QuantizedActs``."""
    _gpu(residual, weight)
    p._live()
    r2 = residual.view(-1, residual.shape[-1])
    if r2.shape != (p.m, p.n):
        raise ValueError("residual shape does not match the partials")
    dev = residual.device
    if quant:
        out = torch.empty(p.m, p.n, dtype=torch.int8, device=dev)
        qs = torch.empty(p.m, dtype=torch.float32, device=dev)
    else:
        out = torch.empty(p.m, p.n, dtype=p.dtype, device=dev)
        qs = None
    check("mi_fused_add_rms_norm_splitk_int8", lib.mi_fused_add_rms_norm_splitk_int8(
        _p(out), _p(qs), _p(p.part), p.split, _p(p.xs), _p(p.ws), _p(p.bias), _p(r2), _p(weight), float(eps), p.m, p.n,
        out.stride(0), r2.stride(0), _dt(p.dtype), _stream()))
    if quant:
        return QuantizedActs(out, qs, p.lead, p.dtype)
    return out.view(*p.lead, p.n)


def int8_linear_silu_mul(x: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor) -> torch.Tensor:
    """``act_fn(gate_up_proj(x))`` with W8A8 weights in one GEMM launch (+ the activation quantisation in front)."""
    if isinstance(x, QuantizedActs):
        xq, xs, lead, dtype = x.xq, x.xs, x.lead, x.dtype
    else:
        xq, xs = dynamic_int8_quant(x.reshape(-1, x.shape[-1]))
        lead, dtype = tuple(x.shape[:-1]), x.dtype
    m, k = xq.shape
    inter = wq.shape[0] // 2
    out = torch.empty(m, inter, dtype=dtype, device=xq.device)
    for r0 in range(0, m, INT8_TILE_ROWS):   # one launch per 512 rows (prefill chunks): rows are independent
        r1 = min(r0 + INT8_TILE_ROWS, m)
        check("mi_int8_linear_silu_mul", lib.mi_int8_linear_silu_mul(
            _p(out[r0:r1]), _p(xq[r0:r1]), _p(xs[r0:r1]), _p(wq), _p(w_scale), r1 - r0, inter, k, xq.stride(0),
            out.stride(0), _dt(dtype), _stream()))
    return out.view(*lead, inter)


# --------------------------------------------------------------------------- N4: fused sparse MoE
def moe_route(router_logits: torch.Tensor, top_k: int, norm_topk_prob: bool, round_to: Optional[torch.dtype] = None,
              want_counts: bool = True):
    """-> (topk_ids int32 [T, k], topk_weights fp32 [T, k], expert_counts int32 [E] or None).  ``want_counts=False``:
    no per-expert counting and no launch to zero the counters (``fused_moe``: the grouping counts the ids itself)."""
    _gpu(router_logits)
    t, e = router_logits.shape
    ids = torch.empty(t, top_k, dtype=torch.int32, device=router_logits.device)
    w = torch.empty(t, top_k, dtype=torch.float32, device=router_logits.device)
    counts = torch.empty(e, dtype=torch.int32, device=router_logits.device) if want_counts else None
    check("mi_moe_route", lib.mi_moe_route(_p(ids), _p(w), _p(counts), _p(router_logits), t, e, router_logits.stride(0),
                                           int(top_k), 1 if norm_topk_prob else 0, _dt(router_logits.dtype),
                                           -1 if round_to is None else _dt(round_to), _stream()))
    return ids, w, counts


def fused_moe(x: torch.Tensor, router_logits: torch.Tensor, w_gate_up: torch.Tensor, w_down: torch.Tensor, top_k: int,
              norm_topk_prob: bool = False, shared_out: Optional[torch.Tensor] = None,
              shared_gate_weight: Optional[torch.Tensor] = None, routing=None,
              rows_per_tile: int = 0) -> torch.Tensor:
    """Sparse-MoE layer output for x [T, H]: route -> align -> gated expert GEMM -> down expert GEMM ->
    weighted combine (+ sigmoid-gated shared expert output computed by the caller).  w_gate_up [E, 2I, H] (gate rows
    then up rows), w_down [E, H, I].  ``routing`` = (ids int32, weights fp32) skips the router (tests)."""
    _gpu(x, router_logits, w_gate_up, w_down, None if callable(shared_out) else shared_out, shared_gate_weight)
    t, h = x.shape
    e, two_i, _ = w_gate_up.shape
    i = two_i // 2
    dev = x.device
    if routing is None:
        ids, w, _ = moe_route(router_logits, top_k, norm_topk_prob, round_to=x.dtype, want_counts=False)
    else:
        ids, w = routing
    counts = None     # mi_moe_align derives the per-expert counts from the ids
    pairs = t * top_k
    rpt = rows_per_tile or int(lib.mi_moe_rows_per_tile(t, top_k, e))
    max_rows = int(lib.mi_moe_max_rows(t, top_k, e, rpt))
    i32 = dict(dtype=torch.int32, device=dev)
    sorted_pairs = torch.empty(max_rows, **i32)
    row_of_pair = torch.empty(pairs, **i32)
    tile_expert = torch.empty(max_rows // rpt, **i32)
    num_tiles = torch.empty(1, **i32)
    check("mi_moe_align", lib.mi_moe_align(_p(sorted_pairs), _p(row_of_pair), _p(tile_expert), _p(num_tiles), _p(ids),
                                           _p(counts), pairs, e, max_rows, rpt, _stream()))
    act = torch.empty(max_rows, i, dtype=x.dtype, device=dev)
    part = torch.empty(max_rows, h, dtype=x.dtype, device=dev)
    dt = _dt(x.dtype)
    check("mi_moe_expert_gemm", lib.mi_moe_expert_gemm(
        _p(act), _p(x), _p(w_gate_up), _p(sorted_pairs), _p(tile_expert), _p(num_tiles), max_rows // rpt, 1, i, h, top_k,
        x.stride(0), act.stride(0), w_gate_up.stride(0), rpt, dt, _stream()))
    check("mi_moe_expert_gemm", lib.mi_moe_expert_gemm(
        _p(part), _p(act), _p(w_down), _p(sorted_pairs), _p(tile_expert), _p(num_tiles), max_rows // rpt, 0, h, i, top_k,
        act.stride(0), part.stride(0), w_down.stride(0), rpt, dt, _stream()))
    out = torch.empty_like(x)
    if callable(shared_out):      # the caller computed it beside this path (another stream): join here
        shared_out = shared_out()
        _gpu(shared_out)
    check("mi_moe_combine", lib.mi_moe_combine(
        _p(out), _p(part), _p(row_of_pair), _p(w), _p(shared_out), _p(x), _p(shared_gate_weight), t, h, top_k,
        part.stride(0), out.stride(0), x.stride(0), shared_out.stride(0) if shared_out is not None else 0, dt, _stream()))
    return out
