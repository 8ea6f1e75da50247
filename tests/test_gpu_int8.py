"""GPU parity: W8A8 int8 linear vs the oracle.  Quantised values and the int32
accumulation are bit-exact; the dequantised output is compared at 1 ulp of the 16-bit output."""
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,H", [(1, 1024), (37, 1024), (512, 384), (5, 200)])
def test_dynamic_int8_quant_bit_exact(dev, dtype, T, H):
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(T + H)
    x = (torch.randn(T, H, generator=g) * 3).to(dtype)
    x[0, :4] = torch.tensor([0.0, 1e-3, -1e-3, 5e-4]).to(dtype)
    if T > 1:
        x[1] = 0  # all-zero row: scale 1, no NaN
    xq, xs = ops.dynamic_int8_quant(x.to(dev))
    eq, es = ref_ops.dynamic_int8_quant(x)
    torch.testing.assert_close(xs.cpu(), es, rtol=1e-6, atol=0)
    diff = (xq.cpu().int() - eq.int()).abs()
    # x/scale is an exact fp32 division on both sides; allow no more than isolated 1-LSB flips at .5 ties
    assert diff.max() <= 1 and (diff != 0).float().mean() < 1e-3
    assert xq.cpu().int().abs().max() <= 128


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,bias", [(512, 1024, 1024, True), (37, 100, 384, False), (1, 2560, 1024, True),
                                        (70, 65, 64, False)])
def test_int8_gemm_dequant(dev, dtype, M, N, K, bias):
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N)
    xq = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32).to(torch.int8)
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int32).to(torch.int8)
    xs = torch.rand(M, generator=g) * 0.05 + 1e-3
    ws = torch.rand(N, generator=g) * 0.01 + 1e-4
    b = torch.randn(N, generator=g).to(dtype) if bias else None
    got = ops.int8_gemm_dequant(xq.to(dev), xs.to(dev), wq.to(dev), ws.to(dev), b.to(dev) if bias else None, dtype)
    acc = xq.to(torch.int32) @ wq.to(torch.int32).t()          # exact
    exp = acc.float() * xs[:, None] * ws[None, :] + (b.float() if bias else 0.0)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(got.cpu().float(), exp.to(dtype).float(), rtol=ulp, atol=1e-5)
    assert (got.cpu().float() == exp.to(dtype).float()).float().mean() > 0.99


def test_int8_linear_end_to_end(dev):
    """quantise activations -> int8 GEMM -> dequant, vs the oracle's int8_linear and the fp32 product."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 1024, generator=g).to(torch.bfloat16)
    w = (torch.randn(512, 1024, generator=g) * 0.02)
    wq, ws = ref_ops.int8_weight_quant(w)
    xq, xs = ops.dynamic_int8_quant(x.to(dev))
    got = ops.int8_gemm_dequant(xq, xs, wq.to(dev), ws.to(dev), None, torch.bfloat16).cpu().float()
    exp = ref_ops.int8_linear(x, wq, ws)
    torch.testing.assert_close(got, exp, rtol=2 ** -6, atol=2e-2)
    torch.testing.assert_close(got, x.float() @ w.t(), rtol=0.1, atol=0.05)   # the scheme itself is ~1-2 % accurate


def test_int8_rejects_unsupported_k(dev):
    from vllm_omni_amd import _C, ops

    with pytest.raises(_C.MIError) as ei:
        ops.int8_gemm_dequant(torch.zeros(4, 100, dtype=torch.int8, device=dev), torch.ones(4, device=dev),
                              torch.zeros(8, 100, dtype=torch.int8, device=dev), torch.ones(8, device=dev), None,
                              torch.bfloat16)
    assert ei.value.code == _C.MI_ERR_UNSUPPORTED


@pytest.mark.parametrize("M,N,K,bias", [(64, 256, 512, True), (512, 1024, 384, False), (17, 96, 64, True), (3, 64, 128, True)])
def test_int8_linear_library_and_mfma_paths_agree_exactly(dev, M, N, K, bias):
    """ops.int8_linear: hand-written quantisation -> int8 contraction (library for M > 16, MFMA kernel otherwise) ->
    hand-written dequant.  Integer sums are exact, so it equals the all-MFMA path bit for bit and the oracle to
    one 16-bit rounding."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16) if bias else None
    wq, ws = ops.dynamic_int8_quant(w.to(dev))
    got = ops.int8_linear(x.to(dev), wq, ws, b.to(dev) if bias else None)
    xq, xs = ops.dynamic_int8_quant(x.to(dev))
    mfma = ops.int8_gemm_dequant(xq, xs, wq, ws, b.to(dev) if bias else None, torch.bfloat16)
    assert torch.equal(got, mfma)
    eq, es = ref_ops.int8_weight_quant(w.float())
    assert torch.equal(wq.cpu(), eq) and torch.equal(ws.cpu(), es)
    exp = ref_ops.int8_linear(x.float(), eq, es, b)
    torch.testing.assert_close(got.cpu().float(), exp, rtol=1e-2, atol=1e-2)


TILED = [
    # (m, n, k, bias): S-A projections at batch 256 (qkv with bias; down_proj = split-K with exact int32 partials), the
    # S-B layer at config #5's batch 512 (two row blocks per weight tile), ragged edges
    (256, 4608, 3584, True), (256, 3584, 3584, False), (256, 3584, 18944, False), (256, 37888, 3584, False),
    (512, 1280, 1024, True), (512, 1024, 1024, False), (512, 1024, 384, True), (300, 768, 1024, False),
    (1, 64, 128, True), (17, 96, 256, False), (100, 160, 128, True), (129, 2048, 2048, True), (64, 3584, 18944, True),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,bias", TILED)
def test_tiled_int8_kernel_is_exact(dev, M, N, K, bias, dtype):
    """mi_int8_linear (LDS-DMA ring, v_mfma_i32_16x16x64_i8, dequant epilogue; both kernel generations, split-K with
    int32 partials, M up to 512) against the exact integer contraction: int32 sums are exact, so the result must
    equal exact-int32 x scales (+ bias) rounded once — and the simple MFMA kernel bit for bit."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(M * 3 + N + K)
    xq = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32).to(torch.int8)
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int32).to(torch.int8)
    xs = torch.rand(M, generator=g) * 0.05 + 1e-3
    ws = torch.rand(N, generator=g) * 0.01 + 1e-4
    b = torch.randn(N, generator=g).to(dtype) if bias else None
    dq = lambda t: None if t is None else t.to(dev)  # noqa: E731
    got = ops.int8_linear_q(dq(xq), dq(xs), dq(wq), dq(ws), dq(b), dtype).cpu()
    simple = ops.int8_gemm_dequant(dq(xq), dq(xs), dq(wq), dq(ws), dq(b), dtype).cpu()
    assert torch.equal(got, simple), f"max diff {(got.float() - simple.float()).abs().max()}"
    if M * N * K <= 1 << 31:                                   # exact int32 reference on the host where it is cheap
        acc = xq.to(torch.int32) @ wq.to(torch.int32).t()
        # the kernels spell the dequantisation as fma(rnd(acc * xs), ws, b) (csrc/common.h dequant_i32): in float64 the
        # fma is exact up to its single rounding, so this host expression has the same bits except at double-rounding ties
        t = (acc.float() * xs[:, None]).double() * ws[None, :].double() + (b.double() if bias else 0.0)
        exp = t.float().to(dtype)
        ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        torch.testing.assert_close(got.float(), exp.float(), rtol=ulp, atol=1e-5)
        assert (got == exp).float().mean() > 0.999


@pytest.mark.parametrize("m,inter,k", [(256, 18944, 3584), (512, 384, 1024), (100, 80, 128), (200, 9472, 3584), (32, 1024, 512)])
def test_tiled_int8_gated_equals_unfused_pair(dev, m, inter, k):
    """gate_up_proj + SiluAndMul in one W8A8 launch against the int8 GEMM followed by the SiLU-gate kernel: both round
    each dequantised GEMM output to the 16-bit grid first; the stand-alone kernel also rounds silu(g) before the
    product (norm_act.hip), the fused epilogue does not — one extra 16-bit rounding, hence one output ulp + the
    ulp of silu(g) times |u|."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(m + inter)
    x = (torch.randn(m, k, generator=g) * 2).to(torch.bfloat16).to(dev)
    w = (torch.randn(2 * inter, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    wq, ws = ops.dynamic_int8_quant(w)
    fused = ops.int8_linear_silu_mul(x, wq, ws)
    unfused = ops.silu_and_mul(ops.int8_linear(x, wq, ws))
    assert fused.shape == (m, inter)
    gu = ops.int8_linear(x, wq, ws).float()
    gt, up = gu[:, :inter], gu[:, inter:]
    act = gt / (1.0 + torch.exp(-gt))
    tol = (act * up).abs() * 2.0 ** -7 + act.abs() * up.abs() * 2.0 ** -8 + 1e-6
    assert bool(((fused.float() - unfused.float()).abs() <= tol).all())


@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1024), (1300, 3584, 3584), (513, 64, 128)])
def test_int8_prefill_rows_never_fall_to_the_simple_kernel(dev, M, N, K, monkeypatch):
    """More rows than one launch of the tiled kernel takes (every prefill chunk): ops.int8_linear takes the library
    contraction + in-tree dequantisation (the measured policy), or — forced with VLLM_OMNI_AMD_INT8=tiled — the tiled
    kernel over blocks of 512 rows; the gated form runs in row blocks.  Bit-identical to the simple kernel / the
    unfused pair, and never the simple kernel itself."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 2).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    wq, ws = ops.dynamic_int8_quant(w)
    xq, xs = ops.dynamic_int8_quant(x)
    simple = ops.int8_gemm_dequant(xq, xs, wq, ws, b, torch.bfloat16)
    calls = []
    real = ops.int8_linear_q
    monkeypatch.setattr(ops, "int8_linear_q", lambda *a, **k: (calls.append(a[0].shape[0]), real(*a, **k))[1])
    monkeypatch.setattr(ops, "int8_gemm_dequant", lambda *a, **k: pytest.fail("simple kernel used for a prefill chunk"))
    got = ops.int8_linear(x, wq, ws, b)
    assert torch.equal(got, simple) and not calls              # library contraction + mi_int8_dequant
    monkeypatch.setenv("VLLM_OMNI_AMD_INT8", "tiled")
    got = ops.int8_linear(x, wq, ws, b)
    assert torch.equal(got, simple) and sum(calls) == M and max(calls) <= 512
    if N % 32 == 0:
        fused = ops.int8_linear_silu_mul(x, wq, ws)
        one = torch.cat([ops.int8_linear_silu_mul(x[r:r + 512], wq, ws) for r in range(0, M, 512)])
        assert torch.equal(fused, one)


def test_int8_linear_paths_agree(dev, monkeypatch):
    """ops.int8_linear: tiled kernel (default), simple MFMA kernel, library contraction + dequant: identical."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(4)
    x = (torch.randn(512, 1024, generator=g) * 2).to(torch.bfloat16).to(dev)
    w = (torch.randn(1280, 1024, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(1280, generator=g).to(torch.bfloat16).to(dev)
    wq, ws = ops.dynamic_int8_quant(w)
    outs = {}
    for mode in ("kernel", "simple", "library"):
        monkeypatch.setenv("VLLM_OMNI_AMD_INT8", mode)
        outs[mode] = ops.int8_linear(x, wq, ws, b)
    assert torch.equal(outs["kernel"], outs["simple"]) and torch.equal(outs["kernel"], outs["library"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,H,add", [(256, 3584, True), (256, 3584, False), (512, 1024, True), (5, 512, True), (1, 8192, False)])
def test_rms_norm_int8_equals_norm_then_quant(dev, T, H, add, dtype):
    """The activation quantisation fused into the (fused-add) RMSNorm in front of a W8A8 linear: same bits as the
    two launches apart — int8 row, scale, and the residual update."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(T + H)
    x = torch.randn(T, H, generator=g).to(dtype).to(dev)
    r = torch.randn(T, H, generator=g).to(dtype).to(dev) if add else None
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype).to(dev)
    x0 = x.clone()
    r1 = r.clone() if add else None
    q = ops.rms_norm_int8(x, w, 1e-6, r1)
    assert torch.equal(x, x0)                                   # x is not modified by the fused form
    if add:
        xa, ra = x0.clone(), r.clone()
        ops.fused_add_rms_norm(xa, ra, w, 1e-6)
        assert torch.equal(r1, ra)
        y = xa
    else:
        y = ops.rms_norm(x0, w, 1e-6)
    eq, es = ops.dynamic_int8_quant(y)
    assert torch.equal(q.xs, es) and torch.equal(q.xq, eq) and q.lead == (T,) and q.dtype == dtype


@pytest.mark.parametrize("T,H", [(256, 18944), (512, 384), (3, 200), (7, 24576)])
def test_vectorised_quant_equals_the_scalar_kernel(dev, T, H):
    from vllm_omni_amd import _C, ops

    g = torch.Generator().manual_seed(T * 5 + H)
    x = (torch.randn(T, H, generator=g) * 3).to(torch.bfloat16).to(dev)
    xq, xs = ops.dynamic_int8_quant(x)                          # vectorised where the shape allows
    sq = torch.empty_like(xq)
    ss = torch.empty_like(xs)
    _C.check("q", _C.lib.mi_dynamic_int8_quant(sq.data_ptr(), ss.data_ptr(), x.data_ptr(), T, H, x.stride(0),
                                               _C.MI_DTYPE_BF16, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(xq, sq) and torch.equal(xs, ss)


@pytest.mark.parametrize("quant", [False, True])
@pytest.mark.parametrize("m,n,k,bias", [(256, 3584, 18944, False), (256, 3584, 3584, True), (64, 3584, 18944, True)])
def test_int8_splitk_reduce_folded_into_the_norm_is_bit_identical(dev, m, n, k, bias, quant):
    """A split-K W8A8 projection whose int32 reduce + dequantisation rides in the fused-add RMSNorm behind it
    (o_proj / down_proj -> post_attention / next input layernorm), optionally leaving as the next linear's int8
    activation: same bits as mi_int8_linear (with its reduce launch) -> mi_fused_add_rms_norm (-> quantisation)."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) * 2).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(n, generator=g).to(torch.bfloat16).to(dev) if bias else None
    res = torch.randn(m, n, generator=g).to(torch.bfloat16).to(dev)
    nw = (1 + 0.1 * torch.randn(n, generator=g)).to(torch.bfloat16).to(dev)
    wq, ws = ops.dynamic_int8_quant(w)
    xq, xs = ops.dynamic_int8_quant(x)
    # unfused
    y = ops.int8_linear_q(xq, xs, wq, ws, b, torch.bfloat16)
    r_ref = res.clone()
    ops.fused_add_rms_norm(y, r_ref, nw, 1e-6)
    # fused
    part = ops.int8_linear_partials(ops.QuantizedActs(xq, xs, (m,), torch.bfloat16), wq, ws, b)
    assert part is not None and part.split > 1
    assert torch.equal(part.materialize(), ops.int8_linear_q(xq, xs, wq, ws, b, torch.bfloat16))
    part = ops.int8_linear_partials(ops.QuantizedActs(xq, xs, (m,), torch.bfloat16), wq, ws, b)   # (materialize ran a linear)
    r_got = res.clone()
    out = ops.fused_add_rms_norm_splitk_int8(part, r_got, nw, 1e-6, quant=quant)
    assert torch.equal(r_got, r_ref)
    if quant:
        eq, es = ops.dynamic_int8_quant(y)
        assert torch.equal(out.xq, eq) and torch.equal(out.xs, es)
    else:
        assert torch.equal(out, y)
