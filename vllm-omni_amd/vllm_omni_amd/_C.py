"""ctypes binding of the C ABI (include/vllm_omni_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile``.  If it is
missing this module raises ImportError: the product path never falls back to torch ops.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

# VLLM_OMNI_AMD_LIB: an alternative build of the same ABI (kernel experiments: csrc/Makefile `make variant`)
_LIB_PATH = os.environ.get("VLLM_OMNI_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                                "libvllm_omni_amd.so")

MI_DTYPE_F16, MI_DTYPE_BF16, MI_DTYPE_F32 = 0, 1, 2
MI_OK = 0
MI_ERR_INVALID_ARG, MI_ERR_UNSUPPORTED, MI_ERR_WORKSPACE = -1, -2, -3
MI_ERR_LAUNCH, MI_ERR_EXHAUSTED, MI_ERR_STATE = -4, -5, -6
ABI_VERSION = 3
MI_LINEAR_PLAN_FULL, MI_LINEAR_PLAN_PARTIALS, MI_LINEAR_PLAN_GATED = 0, 1, 2
MI_LINEAR_COUNTER_BYTES = 16 << 10
MI_DECODE_PACKED_PV = 1
MI_DECODE_FORCE_VALU = 2
MI_DECODE_NONTEMPORAL = 4
MI_DECODE_ROUNDED_P = 8
MI_DECODE_FUSED_MERGE = 16
MI_DECODE_SKIP_MERGE = 32
MI_DECODE_COUNTER_BYTES = 1 << 20
MI_ALLREDUCE_MAX_RANKS, MI_ALLREDUCE_HANDLE_BYTES = 8, 64

# name -> (restype, argtypes); mirrors include/vllm_omni_amd.h declaration by declaration
_P = c_void_p
SIGNATURES = {
    "mi_abi_version": (c_int, []),
    "mi_error_string": (c_char_p, [c_int]),
    "mi_reshape_and_cache": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                      c_int64, c_int64, c_int64, c_int64, c_int64, c_int, _P]),
    "mi_rope_neox_and_cache": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, c_int32, c_int32, _P, _P,
                                        c_float, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                        c_int32, c_int64, c_int64, c_int64, c_int64, c_int64,
                                        c_int64, c_int, _P]),
    "mi_rope_neox_and_cache_splitk": (c_int, [_P, _P, c_int32, _P, _P, c_int64, _P, _P, c_int32, c_int32, _P, _P,
                                               c_float, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32,
                                               c_int64, c_int64, c_int64, c_int64, c_int, _P]),
    "mi_rope_cache_attend_block_splitk": (c_int, [_P, _P, c_int32, _P, _P, c_int32, _P, c_int32, c_int32, _P, _P, c_float,
                                                   _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int64,
                                                   c_int64, c_int64, c_int64, c_float, c_int, _P]),
    "mi_cp_stack_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mi_cp_stack_status": (_P, [_P, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mi_cp_stack_decode": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int32, c_int32, c_int32,
                                    c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64,
                                    c_float, c_float, _P, c_int64, c_int32, c_int, _P]),
    "mi_sample_tokens_embed": (c_int, [_P, _P, c_int32, c_int32, c_int64, c_int, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int64, _P,
                                        c_int64, _P, c_int64, _P]),
    "mi_embed_codes": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, _P]),
    "mi_rope_cache_attend_block": (c_int, [_P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, _P, _P, c_float, _P, _P, _P, c_int64,
                                            c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, c_int64,
                                            c_int64, c_int64, c_int64, c_float, c_int, _P]),
    "mi_paged_attn_decode_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mi_paged_attn_decode_auto_partition": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "mi_paged_attn_decode_effective_partition": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mi_paged_attn_decode": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                      c_int32, c_int32, c_int32, c_float, c_int64, c_int64,
                                      c_int64, c_int64, c_int64, c_int32, _P, c_int64, c_int,
                                      c_int, c_int32, _P]),
    "mi_rope_cache_paged_attn_decode": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, c_float, _P, _P, _P, _P, _P, c_int32, c_int32,
                                                 c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int64, c_int64,
                                                 c_int64, c_int64, c_int64, c_int64, c_int64, c_int32, _P, c_int64, c_int,
                                                 c_int, c_int32, _P]),
    "mi_paged_attn_prefill": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, c_int32, c_float,
                                       c_int64, c_int64, c_int64, c_int64, c_int64, c_int, _P]),
    "mi_paged_attn_prefill_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mi_paged_attn_prefill_split": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32,
                                             c_int32, c_int32, c_int32, c_int32, c_int32, c_float,
                                             c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int32, c_int32,
                                             _P, c_int64, _P]),
    "mi_rms_norm": (c_int, [_P, _P, _P, c_float, c_int32, c_int32, c_int64, c_int64, c_int, _P]),
    "mi_fused_add_rms_norm": (c_int, [_P, _P, _P, c_float, c_int32, c_int32, c_int64, c_int64,
                                       c_int, _P]),
    "mi_silu_and_mul": (c_int, [_P, _P, c_int32, c_int32, c_int64, c_int64, c_int, _P]),
    "mi_mask_logits_argmax": (c_int, [_P, _P, c_int32, c_int32, c_int64, c_int32, c_int32,
                                       c_int32, c_int32, c_int32, c_int, _P]),
    "mi_moe_rows_per_tile": (c_int32, [c_int32, c_int32, c_int32]),
    "mi_moe_max_rows": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "mi_moe_route": (c_int, [_P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32, c_int32, c_int, c_int, _P]),
    "mi_moe_align": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32, _P]),
    "mi_moe_expert_gemm": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64,
                                   c_int64, c_int32, c_int, _P]),
    "mi_moe_combine": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, c_int64,
                               c_int, _P]),
    "mi_int8_dequant": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int64, c_int, _P]),
    "mi_linear_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "mi_linear_partials": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int64, _P, c_int64, POINTER(c_int32), c_int, _P]),
    "mi_linear_reduce": (c_int, [_P, _P, c_int32, _P, c_int32, c_int32, c_int64, c_int, _P]),
    "mi_fused_add_rms_norm_splitk": (c_int, [_P, _P, c_int32, _P, _P, _P, c_float, c_int32, c_int32, c_int64, c_int64,
                                             c_int, _P]),
    "mi_linear_silu_mul": (c_int, [_P, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int64, c_int, _P]),
    "mi_linear": (c_int, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int64, _P, c_int64, c_int, _P]),
    "mi_linear_fused_reduce": (c_int, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int64, _P, c_int64, _P, c_int64,
                                        c_int, _P]),
    "mi_linear_plan_set": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mi_linear_plan_clear": (None, []),
    "mi_linear_plan_get": (c_int, [c_int32, c_int32, c_int32, c_int32, POINTER(c_int32)]),
    "mi_kv_pack": (c_int, [_P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                           c_int64, c_int64, c_int64, _P]),
    "mi_sample_tokens": (c_int, [_P, _P, c_int32, c_int32, c_int64, c_int, _P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P]),
    "mi_talker_decode_embeds": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int64,
                                         c_int, _P]),
    "mi_dynamic_int8_quant": (c_int, [_P, _P, _P, c_int32, c_int32, c_int64, c_int, _P]),
    "mi_int8_gemm_dequant": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64,
                                      c_int, _P]),
    "mi_rms_norm_int8": (c_int, [_P, _P, _P, _P, _P, c_float, c_int32, c_int32, c_int64, c_int64, c_int, _P]),
    "mi_int8_quant_rows": (c_int, [_P, _P, _P, c_int32, c_int32, c_int64, c_int, _P]),
    "mi_int8_linear_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "mi_int8_linear": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int64, _P, c_int64, c_int, _P]),
    "mi_int8_linear_partials": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int64, _P, c_int64, POINTER(c_int32), _P]),
    "mi_fused_add_rms_norm_splitk_int8": (c_int, [_P, _P, _P, c_int32, _P, _P, _P, _P, _P, c_float, c_int32, c_int32, c_int64,
                                                  c_int64, c_int, _P]),
    "mi_int8_linear_silu_mul": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int64, c_int, _P]),
    "mi_allreduce_init": (c_int, [POINTER(c_void_p), c_int32, c_int32, c_int64, c_int32, _P]),
    "mi_allreduce_register_buffer": (c_int, [_P, _P]),
    "mi_allreduce_run": (c_int, [_P, _P, _P, c_int64, c_int32, _P, c_int64, c_int, _P]),
    "mi_allreduce_status": (c_int, [_P]),
    "mi_allreduce_poll": (c_int, [_P]),
    "mi_allreduce_destroy": (c_int, [_P]),
    "mi_blockpool_create": (_P, [c_int32]),
    "mi_blockpool_destroy": (None, [_P]),
    "mi_blockpool_num_free": (c_int32, [_P]),
    "mi_blockpool_num_blocks": (c_int32, [_P]),
    "mi_blockpool_alloc": (c_int, [_P, c_int32, POINTER(c_int32)]),
    "mi_blockpool_ref": (c_int, [_P, POINTER(c_int32), c_int32]),
    "mi_blockpool_free": (c_int, [_P, POINTER(c_int32), c_int32]),
    "mi_blockpool_refcount": (c_int32, [_P, c_int32]),
    "mi_blockpool_cache_insert": (c_int, [_P, c_int32, c_uint64, c_uint64]),
    "mi_blockpool_cache_lookup": (c_int32, [_P, POINTER(c_uint64), c_int32, POINTER(c_int32)]),
    "mi_blockpool_num_cached": (c_int32, [_P]),
    "mi_blockpool_reset_cache": (c_int, [_P]),
    "mi_blockpool_take_evicted": (c_int32, [_P, POINTER(c_int32), c_int32]),
    "mi_compute_slot_mapping": (c_int, [_P, _P, c_int32, _P, _P, c_int32, c_int32]),
}


def _load() -> ctypes.CDLL:
    # torch first: its wheel carries its own libamdhip64, and the process must end up with ONE HIP runtime — if this
    # library were loaded before torch it would pull in /opt/rocm's copy, and kernels registered with that runtime
    # fail to launch on torch's streams ("HIP kernel launch failed"; seen with build() + smoke() in one process)
    import torch  # noqa: F401

    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    got = lib.mi_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libvllm_omni_amd.so ABI version {got} != expected {ABI_VERSION}")
    return lib


lib = _load()
LIB_PATH = _LIB_PATH


class MIError(RuntimeError):
    def __init__(self, fn: str, code: int):
        self.code = code
        msg = lib.mi_error_string(code)
        super().__init__(f"{fn} failed: {code} ({msg.decode() if msg else '?'})")


def check(fn: str, code: int) -> None:
    if code != MI_OK:
        raise MIError(fn, code)
