"""GPU parity, randomised: seeded sweeps over the element-wise / cache-write / GEMM kernels with shapes drawn around their
tile, vector-width and split boundaries — complements the hand-picked shapes of test_gpu_layer_ops.py, test_gpu_linear.py,
test_gpu_int8.py and test_gpu_moe.py.  A few seeds run by default; ``MI_FUZZ_EXTRA=<n>`` adds n more per test (a sweep
to run by hand on a GPU box)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_moe, ref_ops
from test_gpu_layer_ops import ULP, _close16

pytestmark = pytest.mark.gpu

_EXTRA = int(os.environ.get("MI_FUZZ_EXTRA", "0"))
SEEDS = list(range(4)) + list(range(100, 100 + _EXTRA))


def _dtype(rng):
    return torch.bfloat16 if rng.random() < 0.5 else torch.float16


def _close16_rare2(got, exp, dtype, frac_exact=0.99):
    """`_close16` over thousands of shapes: one 16-bit ulp elementwise, except that an element whose FIRST rounding point
    fell on the other side (the device's rsqrt / exp are 1-ulp approximations of torch's) may come out two ulps away after
    the second rounding — at most one element in 10 000, and never more than two ulps."""
    got, exp = got.float(), exp.float()
    torch.testing.assert_close(got, exp, rtol=2 * ULP[dtype], atol=1e-6)
    beyond1 = ((got - exp).abs() > exp.abs() * ULP[dtype] + 1e-6).float().mean().item()
    assert beyond1 <= 1e-4, f"{beyond1:.2e} of the elements are more than one ulp away"
    same = (got == exp).float().mean().item()
    assert same >= frac_exact, f"only {same:.4f} of elements bit-identical"


@pytest.mark.parametrize("seed", SEEDS)
def test_norm_and_activation_sweep(dev, seed):
    from vllm_omni_amd import ops

    rng = np.random.default_rng(7000 + seed)
    g = torch.Generator().manual_seed(seed)
    dtype = _dtype(rng)
    T = int(rng.choice([1, 2, 3, 17, 64, 255, 256, 257, 512, 700]))
    H = int(rng.choice([8, 64, 104, 200, 1000, 1024, 2048, 3584, 4096, 5120, 8192, 8200])) if rng.random() < 0.7 \
        else 8 * int(rng.integers(1, 1100))
    scale = float(rng.choice([1e-3, 1.0, 30.0]))
    x = (torch.randn(T, H, generator=g) * scale).to(dtype)
    r = (torch.randn(T, H, generator=g) * scale).to(dtype)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    eps = float(rng.choice([1e-6, 1e-5]))
    _close16_rare2(ops.rms_norm(x.to(dev), w.to(dev), eps).cpu(), ref_ops.rms_norm(x, w, eps, round_to=dtype), dtype)
    xg, rg = x.to(dev), r.to(dev)
    ops.fused_add_rms_norm(xg, rg, w.to(dev), eps)
    exp_n, exp_r = ref_ops.fused_add_rms_norm(x, r, w, eps, round_to=dtype)
    assert torch.equal(rg.cpu().float(), exp_r)
    _close16_rare2(xg.cpu(), exp_n, dtype)
    inter = H // 2 if H % 16 == 0 else 8 * int(rng.integers(1, 600))
    gu = (torch.randn(T, 2 * inter, generator=g) * float(rng.choice([0.5, 4.0]))).to(dtype)
    _close16_rare2(ops.silu_and_mul(gu.to(dev)).cpu(), ref_ops.silu_and_mul(gu, round_to=dtype), dtype, frac_exact=0.98)


@pytest.mark.parametrize("seed", SEEDS)
def test_rope_and_cache_sweep(dev, seed):
    from vllm_omni_amd import ops

    rng = np.random.default_rng(7100 + seed)
    g = torch.Generator().manual_seed(seed)
    dtype = _dtype(rng)
    d = int(rng.choice([32, 64, 128]))
    nkv = int(rng.choice([1, 2, 4, 8]))
    nq = nkv * int(rng.choice([1, 2, 4, 7, 8]))
    bs = int(rng.choice([4, 8, 16, 32]))
    T = int(rng.choice([1, 2, 9, 63, 64, 65, 256, 300, 512]))
    nb = (T + bs - 1) // bs + 5
    maxpos = 40000
    q = torch.randn(T, nq * d, generator=g).to(dtype)
    k = torch.randn(T, nkv * d, generator=g).to(dtype)
    v = torch.randn(T, nkv * d, generator=g).to(dtype)
    pos = torch.from_numpy(rng.integers(0, maxpos, size=T)).to(torch.int64)
    pos[0] = 0
    pos[-1] = maxpos - 1
    csc = ref_ops.build_cos_sin_cache(d, maxpos, float(rng.choice([1e4, 1e6])))
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
    pad = torch.from_numpy(rng.random(T) < 0.1)
    slots[pad] = -1
    kv = torch.zeros(nb, 2, bs, nkv, d, dtype=dtype)
    kvg = kv.to(dev)
    with_norm = rng.random() < 0.4 and d in (64, 128)   # the fused q/k-norm exists for the head sizes of the Qwen3 family
    qw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype) if with_norm else None
    kw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype) if with_norm else None
    qg, kg = q.to(dev), k.to(dev)
    kwargs = dict(q_norm_weight=qw.to(dev), k_norm_weight=kw.to(dev), norm_eps=1e-6) if with_norm else {}
    ops.rope_neox_and_cache(qg, kg, v.to(dev), pos.to(dev), csc.to(dev), d, kvg[:, 0], kvg[:, 1], slots.to(dev), **kwargs)
    qe = ref_ops.head_rms_norm(q, qw, d, 1e-6, round_to=dtype) if with_norm else q
    ke = ref_ops.head_rms_norm(k, kw, d, 1e-6, round_to=dtype) if with_norm else k
    eq, ek = ref_ops.rope_neox(pos, qe, d, csc, round_to=dtype), ref_ops.rope_neox(pos, ke, d, csc, round_to=dtype)
    if with_norm:
        # two extra rounding points before the rotation (test_qk_norm_rope): 2 ulp of the result, and where the normalised
        # value itself came out one ulp away, that ulp (of the INPUT pair, not of a result that may be their small
        # difference) times |cos| + |sin|; most elements exact
        for gotx, expx, src in ((qg, eq, qe), (kg, ek, ke)):
            torch.testing.assert_close(gotx.cpu().float(), expx, rtol=2 * ULP[dtype],
                                       atol=1.5 * ULP[dtype] * float(src.float().abs().max()))
            assert (gotx.cpu().float() == expx).float().mean() > 0.97
    else:
        _close16_rare2(qg.cpu(), eq, dtype)
        _close16_rare2(kg.cpu(), ek, dtype)
    ref_ops.reshape_and_cache(kg.cpu().view(T, nkv, d), v.view(T, nkv, d), kv[:, 0], kv[:, 1], slots)
    assert torch.equal(kvg.cpu(), kv)          # the cache holds what the kernel wrote back to k, v unchanged, pads skipped


@pytest.mark.parametrize("seed", SEEDS)
def test_linear_sweep(dev, seed):
    """16-bit streaming linear and its fused SiLU form at random shapes of the envelope (rows to 512, k % 128, n % 16)."""
    from vllm_omni_amd import ops

    rng = np.random.default_rng(7200 + seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    dtype = _dtype(rng)
    m = int(rng.choice([1, 16, 17, 64, 65, 128, 129, 255, 256, 257, 300, 511, 512]))
    k = 128 * int(rng.choice([1, 2, 3, 8, 9, 28, 37, 64]))
    n = 16 * int(rng.choice([1, 3, 8, 10, 11, 64, 100, 236, 237, 296]))
    bias = rng.random() < 0.5
    x = torch.randn(m, k, device="cuda", generator=g).to(dtype)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(dtype)
    b = torch.randn(n, device="cuda", generator=g).to(dtype) if bias else None
    assert ops.linear_supported(m, n, k, dtype)
    y = ops.linear(x, w, b)
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    err = (y.double() - ref).abs()
    tol = ref.abs() * ULP[dtype] + 4e-3 * (1 + (k / 1024) ** 0.5) * ULP[dtype] / ULP[torch.bfloat16] + 1e-3
    assert (err <= tol).all(), (m, n, k, bias, dtype, float(err.max()))
    if n % 32 == 0:
        inter = n // 2
        z = ops.linear_silu_mul(x, w)
        gu = (x.double() @ w.double().t()).to(dtype).float()
        gate, up = gu[:, :inter], gu[:, inter:]
        sil = gate / (1.0 + torch.exp(-gate))
        ref2 = sil * up
        slack = (up.abs() * 1.1 * gate.abs() + sil.abs() * up.abs()) * ULP[dtype]
        assert ((z.float() - ref2).abs() <= ref2.abs() * ULP[dtype] + slack + 1e-3).all(), (m, inter, k, dtype)


@pytest.mark.parametrize("seed", SEEDS)
def test_linear_registered_plan_sweep(dev, seed):
    """Round 4: random shapes on a random BUILT tile plan (row block 64 / 128 / 256, 32-256 weight rows per tile, split-K
    1-8) registered through mi_linear_plan_set — plain (with its reduce launch, its partials-only half, and the
    in-launch reduce: all three the same bits) and gated — against float64 references."""
    import ctypes

    from vllm_omni_amd import _C, ops
    from vllm_omni_amd.tuned import linear_plans

    rng = np.random.default_rng(9100 + seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    linear_plans.reset()
    try:
        for _ in range(6):
            dtype = _dtype(rng)
            m = int(rng.choice([1, 31, 64, 65, 100, 128, 129, 200, 256, 300, 512]))
            k = 128 * int(rng.choice([1, 2, 7, 8, 28, 37]))
            n = 32 * int(rng.choice([1, 3, 9, 36, 100, 112, 148]))
            rows = int(rng.choice([64, 128, 256]))
            gated = rng.random() < 0.4
            if gated:
                wrows = 32 * int(rng.choice([1, 2, 3, 4, 5, 6] if rows == 64 else [1, 2, 3, 4, 5] if rows == 128 else [1, 2, 4, 5]))
                split = 1
            else:
                wrows = 32 * int(rng.choice([1, 2, 3, 4, 5, 6, 8] if rows == 64 else [1, 2, 3, 4, 5] if rows == 128 else [4, 5]))
                split = int(rng.choice([1, 2, 3, 4, 6, 8]))
            x = torch.randn(m, k, device="cuda", generator=g).to(dtype)
            if gated:
                inter = n // 2
                assert _C.lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_GATED, 0, 512, inter, k, rows, wrows, 1) == _C.MI_OK
                w = (torch.randn(2 * inter, k, device="cuda", generator=g) * k ** -0.5).to(dtype)
                z = ops.linear_silu_mul(x, w)
                gu = (x.double() @ w.double().t()).to(dtype).float()
                gate, up = gu[:, :inter], gu[:, inter:]
                sil = gate / (1.0 + torch.exp(-gate))
                slack = (up.abs() * 1.1 * gate.abs() + sil.abs() * up.abs()) * ULP[dtype]
                assert ((z.float() - sil * up).abs() <= (sil * up).abs() * ULP[dtype] + slack + 1e-3).all(), (m, inter, k, rows, wrows)
                continue
            assert _C.lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_FULL, 0, 512, n, k, rows, wrows, split) == _C.MI_OK
            buf = (ctypes.c_int32 * 3)()
            assert _C.lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_FULL, m, n, k, buf) == _C.MI_OK and list(buf)[:2] == [rows, wrows]
            w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(dtype)
            b = torch.randn(n, device="cuda", generator=g).to(dtype) if rng.random() < 0.5 else None
            ops.LINEAR_FUSED_REDUCE = False
            y = ops.linear(x, w, b)
            ref = x.double() @ w.double().t() + (b.double() if b is not None else 0.0)
            err = (y.double() - ref).abs()
            tol = ref.abs() * ULP[dtype] + 4e-3 * (1 + (k / 1024) ** 0.5) * ULP[dtype] / ULP[torch.bfloat16] + 1e-3
            assert (err <= tol).all(), (m, n, k, rows, wrows, split, dtype, float(err.max()))
            if buf[2] > 1:
                part = ops.linear_partials(x, w, b)
                assert part is not None and torch.equal(part.materialize(), y)
                ops.LINEAR_FUSED_REDUCE = True
                assert torch.equal(ops.linear(x, w, b), y), (m, n, k, rows, wrows, split)
                ops.LINEAR_FUSED_REDUCE = False
    finally:
        ops.LINEAR_FUSED_REDUCE = os.environ.get("VLLM_OMNI_AMD_LINEAR_FUSED_REDUCE", "0") == "1"
        linear_plans.reset()


@pytest.mark.parametrize("seed", SEEDS)
def test_fused_moe_sweep(dev, seed):
    """The sparse-MoE layer at random token counts (one row block to several tiles per expert, the 1024-thread route
    launch with 1, 2 and 4 tokens per wave), expert counts 64 / 128 / 256 and top-k 1..8, against the oracle evaluated
    with the device's routing."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.moe import MoEConfig, Qwen3OmniMoeSparseMoeBlock

    rng = np.random.default_rng(7300 + seed)
    dtype = _dtype(rng)
    E = int(rng.choice([64, 128, 256]))
    k = int(rng.choice([1, 2, 4, 6, 8]))
    T = int(rng.choice([1, 2, 31, 64, 100, 513, 1030]))
    H = 128 * int(rng.choice([1, 2, 8]))          # the down GEMM's output tile is 128 wide
    I = 64 * int(rng.choice([1, 2, 6]))
    S = None if rng.random() < 0.4 else 64 * int(rng.choice([1, 4]))
    cfg = MoEConfig(hidden_size=H, moe_intermediate_size=I, num_experts=E, num_experts_per_tok=k,
                    norm_topk_prob=bool(rng.random() < 0.5), shared_expert_intermediate_size=S)
    g = torch.Generator().manual_seed(seed)
    r = lambda *shape, std=0.08: (torch.randn(*shape, generator=g) * std).to(dtype)  # noqa: E731
    w = {"gate.weight": r(E, H, std=0.3), "experts.gate_up_proj": r(E, 2 * I, H), "experts.down_proj": r(E, H, I)}
    if S:
        w.update({"shared_expert.gate_proj.weight": r(S, H), "shared_expert.up_proj.weight": r(S, H),
                  "shared_expert.down_proj.weight": r(H, S), "shared_expert_gate.weight": r(1, H, std=0.3)})
    blk = Qwen3OmniMoeSparseMoeBlock(cfg, dtype, dev)
    assert blk.load_weights({n: v.to(dev) for n, v in w.items()}) == set(w.keys())
    x = (torch.randn(T, H, generator=g) * 0.8).to(dtype)
    xg = x.to(dev)
    got = blk(xg).cpu().float()
    logits = torch.nn.functional.linear(xg, blk.gate_weight)
    ids, wts, counts = ops.moe_route(logits, k, cfg.norm_topk_prob, round_to=dtype)
    assert torch.equal(counts.cpu().long(), torch.bincount(ids.cpu().long().reshape(-1), minlength=E))
    assert (ids.cpu().sort(-1).values.diff(dim=-1) != 0).all() if k > 1 else True          # k distinct experts per token
    wf = {n: v.float() for n, v in w.items()}
    ref = ref_moe.moe_block(x.float(), wf, k, cfg.norm_topk_prob, rt=dtype, routing=(wts.cpu(), ids.cpu().long()))
    # as test_fused_moe_matches_oracle: every term is rounded to 16 bits on both sides; atol = one ulp of the largest output
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    atol = float(ref.abs().max()) * ULP[dtype]
    torch.testing.assert_close(got, ref, rtol=tol, atol=atol)


@pytest.mark.parametrize("seed", SEEDS)
def test_int8_linear_sweep(dev, seed):
    """W8A8 tiled kernel at random shapes of its envelope (rows to 512, k % 128, n % 16): int32 sums are exact, so the result
    equals the simple MFMA kernel bit for bit and the exact integer contraction x scales (+ bias) rounded once."""
    from vllm_omni_amd import ops

    rng = np.random.default_rng(7400 + seed)
    g = torch.Generator().manual_seed(seed)
    dtype = _dtype(rng)
    M = int(rng.choice([1, 15, 16, 17, 64, 100, 128, 129, 255, 256, 257, 300, 511, 512]))
    K = 128 * int(rng.choice([1, 2, 3, 8, 9, 16, 28, 37]))
    N = 16 * int(rng.choice([1, 3, 4, 6, 8, 10, 11, 48, 64, 100, 224, 237]))
    bias = rng.random() < 0.5
    xq = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32).to(torch.int8)
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int32).to(torch.int8)
    xs = torch.rand(M, generator=g) * 0.05 + 1e-3
    ws = torch.rand(N, generator=g) * 0.01 + 1e-4
    b = torch.randn(N, generator=g).to(dtype) if bias else None
    dq = lambda t: None if t is None else t.to(dev)  # noqa: E731
    got = ops.int8_linear_q(dq(xq), dq(xs), dq(wq), dq(ws), dq(b), dtype).cpu()
    simple = ops.int8_gemm_dequant(dq(xq), dq(xs), dq(wq), dq(ws), dq(b), dtype).cpu()
    assert torch.equal(got, simple), (M, N, K, bias, dtype, float((got.float() - simple.float()).abs().max()))
    acc = xq.to(torch.int32) @ wq.to(torch.int32).t()
    t = (acc.float() * xs[:, None]).double() * ws[None, :].double() + (b.double() if bias else 0.0)
    exp = t.float().to(dtype)
    torch.testing.assert_close(got.float(), exp.float(), rtol=ULP[dtype], atol=1e-5)
    assert (got == exp).float().mean() > 0.999
