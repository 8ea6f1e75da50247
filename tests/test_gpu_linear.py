"""Decode-batch linear kernel (csrc/skinny_gemm.hip) against an fp32 reference of the same op.

The op is floating point: the tolerance is the bf16/f16 output grid plus fp32 summation-order noise — the kernel
accumulates in fp32 in a different order than a library GEMM (k permuted inside 64-wide blocks, split-K partials),
so results are compared to a float64-accumulated reference rounded to the output type, within 1 output ulp + 1e-3 of
the row's magnitude.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # (m, n, k, bias)  — the four projections of the headline talker, then edge shapes
    (256, 4608, 3584, True),     # qkv_proj (bias)
    (256, 3584, 3584, False),    # o_proj
    (256, 37888, 3584, False),   # gate_up_proj: 160-row tiles, last tile hangs over n
    (256, 3584, 18944, False),   # down_proj: split-K
    (1, 3584, 3584, True),
    (17, 1024, 1024, False),     # ragged m
    (64, 2048, 256, True),       # two k blocks
    (100, 48, 128, False),       # one k block, n smaller than a tile
    (128, 6144, 2048, True),
    (255, 160, 384, True),
    (512, 4096, 1024, False),    # two row blocks of 256 (grid.y): the code predictor's projections at batch 512
    (512, 1024, 3072, True),     # ... with split-K
    (300, 1024, 2048, True),     # ragged second row block
    (257, 160, 384, False),
]


def _ref(x, w, b):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    return y


@pytest.mark.parametrize("m,n,k,bias", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_matches_reference(m, n, k, bias, dtype):
    from vllm_omni_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1234 + m + n + k)
    x = torch.randn(m, k, device="cuda", generator=g).to(dtype)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(dtype)
    b = torch.randn(n, device="cuda", generator=g).to(dtype) if bias else None
    assert ops.linear_supported(m, n, k, dtype)
    y = ops.linear(x, w, b)
    assert y.shape == (m, n) and y.dtype == dtype
    ref = _ref(x, w, b)
    err = (y.double() - ref).abs()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    tol = ref.abs() * ulp + 1e-3 * ref.abs().amax(dim=1, keepdim=True).clamp_min(1.0)
    assert bool((err <= tol).all()), f"max err {err.max().item():.4g} (tol {tol.min().item():.4g}..)"


def test_linear_strided_input_and_3d():
    from vllm_omni_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    big = torch.randn(4, 32, 1024, device="cuda", generator=g).to(torch.bfloat16)
    x = big[:, :, :512]            # row stride 1024, 512 columns
    w = (torch.randn(256, 512, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    y = ops.linear(x, w)
    assert y.shape == (4, 32, 256)
    ref = x.double() @ w.double().t()
    assert (y.double() - ref).abs().max().item() < 0.05


def test_linear_deterministic():
    from vllm_omni_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(256, 18944, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(3584, 18944, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    a = ops.linear(x, w)
    b = ops.linear(x, w)
    assert torch.equal(a, b)     # split-K goes through a workspace and an ordered reduce: no atomics


def test_linear_rejects_unsupported():
    from vllm_omni_amd import ops
    x = torch.zeros(600, 256, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(64, 256, device="cuda", dtype=torch.bfloat16)
    assert not ops.linear_supported(600, 64, 256, torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.linear(x, w)
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 256), torch.zeros(64, 256))    # CPU tensors: no fallback


def test_row_parallel_linear_routes_deep_projections_to_the_kernel(monkeypatch):
    """The "auto" policy (layers._streaming_kernel_wins): a decode batch of up to 256 rows through a deep, narrow
    projection (k >= 8192, k >= 2 n — the headline talker's down_proj at TP 1 and 2) runs the streaming kernel; other
    shapes and larger batches stay on the library GEMM.  Either way the layer's result matches an fp32 reference."""
    from vllm_omni_amd import layers, ops
    calls = []
    real = ops.linear
    monkeypatch.setattr(ops, "linear", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(layers, "_LINEAR_MODE", "auto")
    g = torch.Generator(device="cuda").manual_seed(3)
    lin = layers.RowParallelLinear(8192, 1024, bias=False, dtype=torch.bfloat16, device="cuda")
    lin.weight = (torch.randn(1024, 8192, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    for m, expect in [(256, 1), (200, 1), (128, 1), (32, 1), (512, 0)]:
        calls.clear()
        x = torch.randn(m, 8192, device="cuda", generator=g).to(torch.bfloat16)
        y, _ = lin(x)
        assert len(calls) == expect, (m, calls)
        ref = x.double() @ lin.weight.double().t()
        assert (y.double() - ref).abs().max().item() < 0.03
    monkeypatch.setattr(layers, "_LINEAR_MODE", "library")
    calls.clear()
    lin(torch.randn(256, 8192, device="cuda", generator=g).to(torch.bfloat16))
    assert not calls


@pytest.mark.parametrize("m,inter,k", [(256, 18944, 3584), (256, 1024, 512), (100, 80, 128), (17, 4736, 1024),
                                        (64, 96, 256), (128, 9472, 3584), (512, 3072, 1024), (333, 80, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_silu_mul_matches_unfused_pair(m, inter, k, dtype):
    """gate_up_proj + SiluAndMul in one launch against the same two steps done apart with the rounding the unfused
    pair has: GEMM outputs rounded to the 16-bit type, SiLU and product in fp32, one rounding.  Tolerance: one output
    ulp for the final rounding + the effect of a one-ulp difference in either rounded GEMM output (fp32 summation
    order differs from the float64 reference)."""
    from vllm_omni_amd import ops
    g = torch.Generator(device="cuda").manual_seed(99 + m + inter + k)
    x = torch.randn(m, k, device="cuda", generator=g).to(dtype)
    w = (torch.randn(2 * inter, k, device="cuda", generator=g) * k ** -0.5).to(dtype)
    y = ops.linear_silu_mul(x, w)
    assert y.shape == (m, inter) and y.dtype == dtype
    gu = (x.double() @ w.double().t()).to(dtype).float()          # the unfused pair's intermediate
    gate, up = gu[:, :inter], gu[:, inter:]
    ref = (gate / (1.0 + torch.exp(-gate)) * up)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # d(silu(g) u) <= |u| * 1.1 * dg + |silu(g)| * du with dg, du one ulp of g, u
    slack = (up.abs() * 1.1 * gate.abs() + (gate / (1.0 + torch.exp(-gate))).abs() * up.abs()) * ulp
    tol = ref.abs() * ulp + slack + 1e-3
    err = (y.float() - ref).abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.4g}"
    # and against the product's own unfused kernels
    z = ops.silu_and_mul(ops.linear(x, w)) if ops.linear_supported(m, 2 * inter, k, dtype) else None
    if z is not None:
        assert (y.float() - z.float()).abs().max().item() <= float(tol.max())


def test_mlp_uses_the_fused_gate_up_kernel_when_the_policy_says_so(monkeypatch):
    """Qwen2MLP: act_fn(gate_up_proj(x)) runs as one launch for I >= 8192 per rank and for a small dense MLP
    (hidden <= 2048, I <= 4096: the code predictor's) — auto — and as two steps otherwise; both ways equal the two-step
    result to the fused kernel's tolerance."""
    from vllm_omni_amd import layers, ops
    fused = []
    real = ops.linear_silu_mul
    monkeypatch.setattr(ops, "linear_silu_mul", lambda *a, **k: (fused.append(1), real(*a, **k))[1])
    monkeypatch.setattr(layers, "_LINEAR_MODE", "auto")
    g = torch.Generator(device="cuda").manual_seed(5)
    for inter, expect in [(8192, 1), (6144, 0), (3072, 1)]:
        gu = layers.MergedColumnParallelLinear(512, [inter] * 2, bias=False, dtype=torch.bfloat16, device="cuda")
        gu.weight = (torch.randn(2 * inter, 512, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        x = torch.randn(64, 512, device="cuda", generator=g).to(torch.bfloat16)
        fused.clear()
        y = layers.fused_gate_up_silu(gu, x)
        assert len(fused) == expect and (y is None) == (expect == 0)
        if y is not None:
            z = ops.silu_and_mul(torch.nn.functional.linear(x, gu.weight))
            assert (y.float() - z.float()).abs().max().item() <= 0.02 * z.float().abs().max().item() + 1e-3


@pytest.mark.parametrize("m,n,k,bias", [(256, 3584, 18944, False), (32, 3584, 18944, True), (100, 512, 2048, True)])
def test_deferred_splitk_reduce_is_bit_identical(plan_registry, m, n, k, bias):
    """linear_partials + fused_add_rms_norm_splitk == linear + fused_add_rms_norm, bit for bit (same slice order, same
    roundings), for the output and for the updated residual; materialize() == linear.  (Both entry points on the SAME
    tile plan — the built-in one here; a tuned table may give them different split-K factors, and then they differ by
    fp32 summation order.)"""
    from vllm_omni_amd import ops
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16) if bias else None
    res = torch.randn(m, n, device="cuda", generator=g).to(torch.bfloat16)
    nw = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
    y = ops.linear(x, w, b)
    r1 = res.clone()
    y1 = y.clone()
    ops.fused_add_rms_norm(y1, r1, nw, 1e-6)
    part = ops.linear_partials(x, w, b)
    assert part is not None and part.split >= 2
    assert torch.equal(part.materialize(), y)
    r2 = res.clone()
    y2 = ops.fused_add_rms_norm_splitk(part, r2, nw, 1e-6)
    assert torch.equal(y2, y1) and torch.equal(r2, r1)
    # partials are consumed in order: a later split-K linear invalidates them
    ops.linear(x, w, b)
    with pytest.raises(RuntimeError, match="overwritten"):
        part.materialize()


def test_linear_partials_none_without_split(plan_registry):
    from vllm_omni_amd import ops
    x = torch.zeros(256, 3584, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(37888, 3584, device="cuda", dtype=torch.bfloat16)
    assert ops.linear_partials(x, w) is None      # 237 tiles fill the chip: no split-K, nothing to defer


@pytest.mark.parametrize("mrope,qk_norm,bias", [(False, False, True), (True, False, True), (False, True, False)])
def test_rope_and_cache_from_qkv_partials_is_bit_identical(plan_registry, mrope, qk_norm, bias):
    """qkv_proj's split-K partials straight into the fused RoPE + KV-write kernel == linear -> rope_neox_and_cache:
    rotated q, K cache and V cache bit for bit (plain RoPE, M-RoPE, q/k-norm; with and without bias)."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.layers import build_cos_sin_cache
    T, nq, nkv, d, hid, bs, nb = 48, 28, 4, 128, 3584, 16, 12
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(T, hid, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((nq + 2 * nkv) * d, hid, device="cuda", generator=g) * hid ** -0.5).to(torch.bfloat16)
    b = torch.randn((nq + 2 * nkv) * d, device="cuda", generator=g).to(torch.bfloat16) if bias else None
    csc = build_cos_sin_cache(d, 4096, 1e6, device="cuda")
    pos = torch.randint(0, 4096, (3, T) if mrope else (T,), device="cuda", generator=g)
    sec = torch.tensor([16, 24, 24], dtype=torch.int32) if mrope else None
    qn = (1 + 0.1 * torch.randn(d, device="cuda", generator=g)).to(torch.bfloat16) if qk_norm else None
    kn = (1 + 0.1 * torch.randn(d, device="cuda", generator=g)).to(torch.bfloat16) if qk_norm else None
    slots = torch.randperm(nb * bs, device="cuda", generator=g)[:T].to(torch.int64)
    slots[5] = -1                                            # a padding token: no cache write
    caches = []
    for _ in range(2):
        kc = torch.zeros(nb, bs, nkv, d, device="cuda", dtype=torch.bfloat16)
        vc = torch.zeros_like(kc)
        caches.append((kc, vc))
    # the two-launch form
    qkv = ops.linear(x, w, b)
    q, k, v = qkv.split([nq * d, nkv * d, nkv * d], dim=-1)
    ops.rope_neox_and_cache(q, k, v, pos, csc, d, caches[0][0], caches[0][1], slots, mrope_section=sec,
                            q_norm_weight=qn, k_norm_weight=kn)
    # partials straight into the kernel
    part = ops.linear_partials(x, w, b)
    assert part is not None
    q2 = ops.rope_neox_and_cache_splitk(part, nq, nkv, d, pos, csc, caches[1][0], caches[1][1], slots, mrope_section=sec,
                                        q_norm_weight=qn, k_norm_weight=kn)
    assert torch.equal(q2, q.contiguous())
    assert torch.equal(caches[1][0], caches[0][0]) and torch.equal(caches[1][1], caches[0][1])
    assert caches[0][0].abs().sum() > 0


def test_linear_random_shapes_fuzz():
    """Random (m, n, k) across every template variant of the kernel (m-tile counts 4/8/16, 128- and 160-row tiles,
    split-K factors 1-16, tiles hanging over n, ragged m), plain and gated, against float64 references."""
    from vllm_omni_amd import ops
    rng = torch.Generator().manual_seed(2024)
    g = torch.Generator(device="cuda").manual_seed(77)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    for case in range(40):
        m = ri(1, 256)
        k = 64 * ri(1, 40)
        n = 16 * ri(1, 300) if case % 4 else 160 * ri(129, 200)   # every 4th: the 160-row-tile plan
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
        b = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16) if case % 2 else None
        y = ops.linear(x, w, b)
        ref = x.double() @ w.double().t() + (b.double() if b is not None else 0)
        tol = ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().amax(dim=1, keepdim=True).clamp_min(1.0)
        assert bool(((y.double() - ref).abs() <= tol).all()), (case, m, n, k)
        if n % 32 == 0:
            inter = n // 2
            z = ops.linear_silu_mul(x, w)
            gu = (x.double() @ w.double().t()).to(torch.bfloat16).float()
            gt, up = gu[:, :inter], gu[:, inter:]
            act = gt / (1.0 + torch.exp(-gt))
            zr = act * up
            ztol = zr.abs() * 2.0 ** -8 + (up.abs() * 1.1 * gt.abs() + act.abs() * up.abs()) * 2.0 ** -8 + 1e-3
            assert bool(((z.float() - zr).abs() <= ztol).all()), ("gated", case, m, inter, k)


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: registered tile plans (mi_linear_plan_set) — the per-rank shapes of the tensor-parallel plans (qwen2_old.py:
# 89-102: 28 / tp query heads, max(1, 4 / tp) KV heads; 47-60,109-124 the four projections) on every built tile shape.
def _talker_shard_shapes(tp):
    nq, nkv = 28 // tp, max(1, 4 // tp)
    return {"qkv": ((nq + 2 * nkv) * 128, 3584), "o": (3584, nq * 128), "down": (3584, 18944 // tp)}


_PLAIN_PLANS = [(64, w, s) for w in (32, 64, 96, 128, 160, 192, 256) for s in (1, 3)] + \
               [(128, w, s) for w in (32, 64, 96, 128, 160) for s in (1, 4)] + [(256, 128, 2), (256, 160, 1)]


@pytest.mark.parametrize("rows,wrows,split", _PLAIN_PLANS)
def test_registered_plain_plans_match_reference(plan_registry, rows, wrows, split):
    """Every built (rows per block, weight rows per tile, split-K) on a TP-4 shard shape with 128 rows, a ragged row
    count that leaves the last row block partly empty, and a tile count that is not a multiple of 8 (the id -> (item,
    row block) map pads the grid): against the float64 reference; the partials-only half + its reduce == the one call,
    bit for bit."""
    from vllm_omni_amd import ops
    _C = plan_registry
    g = torch.Generator(device="cuda").manual_seed(rows + wrows + split)
    for m, (n, k) in [(128, _talker_shard_shapes(4)["down"]), (100, (1152, 896)), (256, (3584, 896)), (37, (208, 256))]:
        assert _C.lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 0, 512, n, k, rows, wrows, split) == _C.MI_OK
        import ctypes
        buf = (ctypes.c_int32 * 3)()
        assert _C.lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_FULL, m, n, k, buf) == _C.MI_OK
        assert list(buf)[:2] == [rows, wrows] and buf[2] == (split if split <= k // 64 else 1), list(buf)
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
        b = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
        y = ops.linear(x, w, b)
        ref = _ref(x, w, b)
        err = (y.double() - ref).abs()
        tol = ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().amax(dim=1, keepdim=True).clamp_min(1.0)
        assert bool((err <= tol).all()), f"m={m} n={n} k={k}: max err {err.max().item():.4g}"
        part = ops.linear_partials(x, w, b)
        if buf[2] > 1:
            assert part is not None and part.split == buf[2]
            assert torch.equal(part.materialize(), y)
        else:
            assert part is None


@pytest.mark.parametrize("rows,wrows", [(64, w) for w in (32, 64, 96, 128, 160, 192)] + [(128, w) for w in (32, 64, 96, 128, 160)]
                         + [(256, 32), (256, 64), (256, 128), (256, 160)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_registered_gated_plans_match_unfused_pair(plan_registry, rows, wrows, dtype):
    """gate_up_proj + SiluAndMul on every built gated tile (weight rows per tile = 2 x its output columns), TP-4 shard
    shape at 128 and 256 rows, ragged rows and a last tile that hangs over the intermediate size."""
    from vllm_omni_amd import ops
    _C = plan_registry
    g = torch.Generator(device="cuda").manual_seed(rows + wrows)
    for m, inter, k in [(128, 4736, 3584), (256, 4736, 1024), (100, 304, 256)]:
        assert _C.lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_GATED, 0, 512, inter, k, rows, wrows, 1) == _C.MI_OK
        x = torch.randn(m, k, device="cuda", generator=g).to(dtype)
        w = (torch.randn(2 * inter, k, device="cuda", generator=g) * k ** -0.5).to(dtype)
        y = ops.linear_silu_mul(x, w)
        gu = (x.double() @ w.double().t()).to(dtype).float()
        gate, up = gu[:, :inter], gu[:, inter:]
        ref = gate / (1.0 + torch.exp(-gate)) * up
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        slack = (up.abs() * 1.1 * gate.abs() + (gate / (1.0 + torch.exp(-gate))).abs() * up.abs()) * ulp
        err = (y.float() - ref).abs()
        assert bool((err <= ref.abs() * ulp + slack + 1e-3).all()), f"m={m} inter={inter}: max err {err.max().item():.4g}"


def test_plans_of_different_tile_shapes_give_the_same_sums_up_to_order(plan_registry):
    """Two plans of one problem differ only in fp32 summation order (split-K slices): same 16-bit output to one ulp."""
    from vllm_omni_amd import ops
    _C = plan_registry
    g = torch.Generator(device="cuda").manual_seed(11)
    n, k, m = 3584, 4736, 128
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    outs = []
    for rows, wrows, split in [(64, 96, 1), (128, 64, 1), (64, 32, 1)]:
        assert _C.lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 0, 512, n, k, rows, wrows, split) == _C.MI_OK
        outs.append(ops.linear(x, w))
    # no split: every output element is ONE fp32 chain over k in the same (permuted) order, whatever the tile shape
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_tuned_table_routes_the_shard_shapes(plan_registry, monkeypatch):
    """tuned/linear_plans_gfx950.json (tools/tune_linear_plans.py): loading it registers a plan per measured shape, and
    layers routes a TP-4 rank's projections at 128 rows by its timings — the fused gate_up where it is ahead of the
    library pair, o_proj / down_proj on the kernel — with results equal to the reference."""
    from vllm_omni_amd import layers, ops
    from vllm_omni_amd.tuned import linear_plans
    monkeypatch.setattr(layers, "_LINEAR_MODE", "auto")
    assert linear_plans.load() >= 40
    e = linear_plans.lookup(128, 4736, 3584, True)
    assert e is not None and e["m"] == 128 and e["full"]["us"] < e["library_us"]
    assert linear_plans.lookup(100, 3584, 4736, False)["m"] == 128 and linear_plans.lookup(300, 3584, 4736, False) is None
    calls = {"fused": 0, "linear": 0}
    rf, rl = ops.linear_silu_mul, ops.linear
    monkeypatch.setattr(ops, "linear_silu_mul", lambda *a, **k: (calls.__setitem__("fused", calls["fused"] + 1), rf(*a, **k))[1])
    monkeypatch.setattr(ops, "linear", lambda *a, **k: (calls.__setitem__("linear", calls["linear"] + 1), rl(*a, **k))[1])
    g = torch.Generator(device="cuda").manual_seed(2)
    gu = layers.MergedColumnParallelLinear(3584, [4736] * 2, bias=False, dtype=torch.bfloat16, device="cuda")
    gu.weight = (torch.randn(2 * 4736, 3584, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    x = torch.randn(128, 3584, device="cuda", generator=g).to(torch.bfloat16)
    y = layers.fused_gate_up_silu(gu, x)
    assert y is not None and calls["fused"] == 1
    z = ops.silu_and_mul(torch.nn.functional.linear(x, gu.weight))
    assert (y.float() - z.float()).abs().max().item() <= 0.02 * z.float().abs().max().item() + 1e-3
    dn = layers.RowParallelLinear(4736, 3584, bias=False, dtype=torch.bfloat16, device="cuda")
    dn.weight = (torch.randn(3584, 4736, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    xa = torch.randn(128, 4736, device="cuda", generator=g).to(torch.bfloat16)
    yd, _ = dn(xa)
    assert calls["linear"] == 1
    assert (yd.double() - xa.double() @ dn.weight.double().t()).abs().max().item() < 0.05


@pytest.mark.parametrize("rows,wrows,split", [(64, 96, 3), (64, 128, 8), (128, 64, 4), (64, 32, 2), (128, 128, 6), (64, 256, 2)])
def test_in_launch_splitk_reduce_equals_the_reduce_launch_bit_for_bit(plan_registry, monkeypatch, rows, wrows, split):
    """mi_linear_fused_reduce: the last-arriving slice of a (tile, row block) sums the slabs — same slice order, same
    rounding as the separate reduce launch — over repeated launches on one workspace (the counters must come back to
    zero), ragged rows, a tile that hangs over n, with bias, and replayed from a hipGraph."""
    from vllm_omni_amd import ops
    _C = plan_registry
    g = torch.Generator(device="cuda").manual_seed(rows + wrows + split)
    for m, n, k in [(128, 3584, 4736), (100, 1152, 3584), (256, 3584, 896), (37, 208, 1024)]:
        assert _C.lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 0, 512, n, k, rows, wrows, split) == _C.MI_OK
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
        b = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
        monkeypatch.setattr(ops, "LINEAR_FUSED_REDUCE", False)
        two = ops.linear(x, w, b)
        monkeypatch.setattr(ops, "LINEAR_FUSED_REDUCE", True)
        for _ in range(3):
            one = ops.linear(x, w, b)
            assert torch.equal(one, two), (m, n, k)
        assert int(ops._LINEAR_CNT[x.device].abs().sum()) == 0
    # replayed from a graph, inputs changing between replays
    m, n, k = 128, 3584, 4736
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.linear(x, w)
    torch.cuda.current_stream().wait_stream(st)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = ops.linear(x, w)
    for _ in range(3):
        x.copy_(torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16))
        graph.replay()
        monkeypatch.setattr(ops, "LINEAR_FUSED_REDUCE", False)
        ref = ops.linear(x, w)
        monkeypatch.setattr(ops, "LINEAR_FUSED_REDUCE", True)
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
