"""CPU: the C-ABI library loads and exports every symbol include/vllm_omni_amd.h declares; host-side
entry points (block pool, slot mapping) behave as the oracle says.  No GPU compute call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import ref_ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "vllm_omni_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from vllm_omni_amd import _C

    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(_C.lib, name), f"{name} declared in the header but not exported"
    assert sorted(_C.SIGNATURES) == declared, "ctypes signature table and header disagree"
    assert _C.lib.mi_abi_version() == _C.ABI_VERSION
    assert _C.lib.mi_error_string(_C.MI_ERR_EXHAUSTED).decode().startswith("KV block pool")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    import sys

    import vllm_omni_amd._C as C

    monkeypatch.setattr(C, "_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        C._load()


def test_ops_reject_cpu_tensors():
    from vllm_omni_amd import ops

    x = torch.randn(2, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rms_norm(x, torch.ones(64, dtype=torch.bfloat16), 1e-6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.silu_and_mul(x)


def test_argument_validation_without_gpu():
    """Invalid-argument paths return before any launch, so they are checkable without a GPU."""
    from vllm_omni_amd import _C

    lib = _C.lib
    assert lib.mi_rms_norm(None, None, None, 1e-6, 4, 64, 64, 64, _C.MI_DTYPE_BF16, None) == _C.MI_ERR_INVALID_ARG
    assert lib.mi_rms_norm(None, None, None, 1e-6, 0, 64, 64, 64, _C.MI_DTYPE_BF16, None) == _C.MI_OK  # empty
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 15) & ~15
    assert lib.mi_paged_attn_decode(p, p, p, p, p, p, 2, 8, 2, 96, 16, 4, 64, 1.0, 768, 768, 1, 1, 1, 0, None, 0,
                                    _C.MI_DTYPE_BF16, _C.MI_DTYPE_BF16, 0, None) == _C.MI_ERR_UNSUPPORTED
    assert lib.mi_paged_attn_decode(p, p, p, p, p, p, 2, 8, 3, 128, 16, 4, 64, 1.0, 1024, 1024, 8, 8, 8, 0, None, 0,
                                    _C.MI_DTYPE_BF16, _C.MI_DTYPE_BF16, 0, None) == _C.MI_ERR_INVALID_ARG
    assert lib.mi_paged_attn_decode(p, p, p, p, p, p, 2, 8, 2, 128, 12, 4, 64, 1.0, 1024, 1024, 8, 8, 8, 0, None, 0,
                                    _C.MI_DTYPE_BF16, _C.MI_DTYPE_BF16, 0, None) == _C.MI_ERR_UNSUPPORTED
    # arrival counters of the fused merge (fixed region at the front) + fp32 (O, m, l) partials
    assert lib.mi_paged_attn_decode_workspace_bytes(256, 28, 128, 4096, 512) == _C.MI_DECODE_COUNTER_BYTES + 256 * 28 * 8 * 130 * 4
    assert lib.mi_paged_attn_decode_workspace_bytes(256, 28, 128, 4096, 4096) == 0
    # the fewest partitions that leave >= 512 waves (ONE at >= 512 sequence x KV-head pairs), 256..8192 tokens, never
    # beyond the (rounded) context
    auto = lambda s, h, c, d=128: lib.mi_paged_attn_decode_auto_partition(s, h, c, d)   # noqa: E731
    assert auto(256, 4, 4096) == 4096 and auto(64, 4, 2048) == 1024 and auto(32, 4, 1024) == 256
    assert auto(1, 4, 8192) == 256 and auto(4, 2, 40) == 64 and auto(256, 1, 4096) == 2048
    assert auto(256, 4, 4112) == 4128 and auto(256, 4, 32768) == 8192 and auto(96, 4, 4096) == 2048
    # head_dim 64: a wave carries half the bytes -> twice the waves
    assert auto(512, 2, 4096, 64) == 4096 and auto(128, 2, 4096, 64) == 1024 and auto(64, 2, 4096, 64) == 512
    eff = lib.mi_paged_attn_decode_effective_partition
    assert eff(256, 4, 4096, 128, 16, 0, _C.MI_DECODE_PACKED_PV) == 4096               # matrix-core kernel: any length
    assert eff(256, 4, 4096, 128, 16, 0, 0) == 1024                                    # VALU kernel: 64-block cap
    assert eff(256, 4, 4096, 128, 16, 0, _C.MI_DECODE_PACKED_PV | _C.MI_DECODE_FORCE_VALU) == 1024
    assert eff(256, 4, 4096, 128, 8, 0, _C.MI_DECODE_PACKED_PV) == 512                 # block 8: VALU kernel


def test_blockpool_fifo_refcount_and_null_block():
    from vllm_omni_amd import _C

    lib = _C.lib
    pool = lib.mi_blockpool_create(6)
    assert pool and lib.mi_blockpool_num_free(pool) == 5 and lib.mi_blockpool_num_blocks(pool) == 6
    ids = (ctypes.c_int32 * 3)()
    assert lib.mi_blockpool_alloc(pool, 3, ids) == _C.MI_OK and list(ids) == [1, 2, 3]   # never the null block 0
    big = (ctypes.c_int32 * 3)()
    assert lib.mi_blockpool_alloc(pool, 3, big) == _C.MI_ERR_EXHAUSTED                   # all-or-nothing
    assert lib.mi_blockpool_num_free(pool) == 2
    two = (ctypes.c_int32 * 1)(2)
    assert lib.mi_blockpool_ref(pool, two, 1) == _C.MI_OK and lib.mi_blockpool_refcount(pool, 2) == 2
    assert lib.mi_blockpool_free(pool, ids, 3) == _C.MI_OK
    assert lib.mi_blockpool_num_free(pool) == 4 and lib.mi_blockpool_refcount(pool, 2) == 1  # still referenced
    nxt = (ctypes.c_int32 * 4)()
    assert lib.mi_blockpool_alloc(pool, 4, nxt) == _C.MI_OK and list(nxt) == [4, 5, 1, 3]   # FIFO: freed go to the tail
    assert lib.mi_blockpool_free(pool, (ctypes.c_int32 * 1)(0), 1) == _C.MI_ERR_STATE       # null block
    assert lib.mi_blockpool_free(pool, (ctypes.c_int32 * 1)(9), 1) == _C.MI_ERR_STATE       # out of range
    assert lib.mi_blockpool_free(pool, two, 1) == _C.MI_OK
    assert lib.mi_blockpool_free(pool, two, 1) == _C.MI_ERR_STATE                           # double free
    assert lib.mi_blockpool_create(1) is None
    lib.mi_blockpool_destroy(pool)


def test_slot_mapping_matches_oracle_bit_exact():
    from vllm_omni_amd import _C

    rng = np.random.default_rng(0)
    for bs in (4, 8, 16, 32):
        nreq, maxb = 7, 9
        table = rng.integers(1, 1000, size=(nreq, maxb), dtype=np.int32)
        t = 61
        req = rng.integers(0, nreq, size=t).astype(np.int32)
        pos = rng.integers(0, maxb * bs, size=t).astype(np.int64)
        out = np.full(t, -7, dtype=np.int64)
        rc = _C.lib.mi_compute_slot_mapping(out.ctypes.data, table.ctypes.data, maxb, req.ctypes.data, pos.ctypes.data, t, bs)
        assert rc == _C.MI_OK
        exp = ref_ops.slot_mapping(torch.from_numpy(table), torch.from_numpy(req), torch.from_numpy(pos), bs)
        assert np.array_equal(out, exp.numpy())
    bad = np.array([maxb * 32], dtype=np.int64)
    assert _C.lib.mi_compute_slot_mapping(out.ctypes.data, table.ctypes.data, maxb, req.ctypes.data, bad.ctypes.data, 1,
                                          32) == _C.MI_ERR_INVALID_ARG


def test_linear_plan_registry_host_side():
    """mi_linear_plan_set / _get / _clear (round 4): a registered plan replaces the heuristic for its row range only; the
    partials entry point falls back to the full plan; malformed and unbuilt plans are refused."""
    from vllm_omni_amd import _C
    from vllm_omni_amd.tuned import linear_plans

    linear_plans.reset()
    lib = _C.lib
    buf = (ctypes.c_int32 * 3)()
    n, k = 3584, 4736
    assert lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_FULL, 128, n, k, buf) == _C.MI_OK
    heuristic = list(buf)
    assert heuristic[0] == 128 and heuristic[2] >= 2           # 128-row image, split-K
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 64, 128, n, k, 64, 96, 3) == _C.MI_OK
    assert lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_FULL, 128, n, k, buf) == _C.MI_OK and list(buf) == [64, 96, 3]
    assert lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_PARTIALS, 100, n, k, buf) == _C.MI_OK and list(buf) == [64, 96, 3]
    assert lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_FULL, 64, n, k, buf) == _C.MI_OK and list(buf) != [64, 96, 3]
    assert lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_FULL, 129, n, k, buf) == _C.MI_OK and list(buf) != [64, 96, 3]
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_PARTIALS, 64, 128, n, k, 128, 128, 8) == _C.MI_OK
    assert lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_PARTIALS, 128, n, k, buf) == _C.MI_OK and list(buf) == [128, 128, 8]
    assert lib.mi_linear_workspace_bytes(128, n, k) == 8 * 128 * n * 4      # one workspace serves both entry points
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_GATED, 0, 128, 4736, 3584, 64, 96, 1) == _C.MI_OK
    assert lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_GATED, 128, 4736, 3584, buf) == _C.MI_OK and list(buf) == [64, 96, 1]
    # refused: unbuilt tile, malformed ranges / splits, a split on the gated form, a partials plan without a split
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 0, 128, n, k, 64, 224, 1) == _C.MI_ERR_UNSUPPORTED
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 0, 128, n, k, 256, 96, 1) == _C.MI_ERR_UNSUPPORTED
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 128, 128, n, k, 64, 96, 1) == _C.MI_ERR_INVALID_ARG
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 0, 128, n, k, 64, 96, 17) == _C.MI_ERR_INVALID_ARG
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_GATED, 0, 128, 4736, 3584, 64, 96, 2) == _C.MI_ERR_INVALID_ARG
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_PARTIALS, 0, 128, n, k, 64, 96, 1) == _C.MI_ERR_INVALID_ARG
    assert lib.mi_linear_plan_set(7, 0, 128, n, k, 64, 96, 1) == _C.MI_ERR_INVALID_ARG
    lib.mi_linear_plan_clear()
    assert lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_FULL, 128, n, k, buf) == _C.MI_OK and list(buf) == heuristic


def test_tuned_linear_plan_table_loads_and_buckets_rows(monkeypatch):
    """vllm_omni_amd/tuned/linear_plans_gfx950.json: every entry registers; an entry tuned at m rows serves the row counts
    above the next lower tuned m; VLLM_OMNI_AMD_LINEAR_PLANS=0 disables the table."""
    import json

    from vllm_omni_amd import _C
    from vllm_omni_amd.tuned import linear_plans

    linear_plans.reset()
    doc = json.load(open(linear_plans.RESULTS))
    assert doc["arch"] == "gfx950" and len(doc["entries"]) >= 40
    assert linear_plans.load() == len({(e["gated"], e["n"], e["k"], e["m"]) for e in doc["entries"]})
    buf = (ctypes.c_int32 * 3)()
    for e in doc["entries"]:
        got = linear_plans.lookup(e["m"], e["n"], e["k"], e["gated"])
        assert got is not None and got["m"] == e["m"]
        purpose = _C.MI_LINEAR_PLAN_GATED if e["gated"] else _C.MI_LINEAR_PLAN_FULL
        assert _C.lib.mi_linear_plan_get(purpose, e["m"], e["n"], e["k"], buf) == _C.MI_OK
        assert list(buf) == [e["full"]["rows"], e["full"]["wrows"], e["full"]["split"]]
    assert linear_plans.lookup(33, 3584, 4736, False)["m"] == 64 and linear_plans.lookup(257, 3584, 4736, False) is None
    assert linear_plans.lookup(128, 3584, 4737, False) is None
    linear_plans.reset()
    monkeypatch.setenv("VLLM_OMNI_AMD_LINEAR_PLANS", "0")
    assert linear_plans.load() == 0 and linear_plans.lookup(128, 3584, 4736, False) is None
    linear_plans.reset()
