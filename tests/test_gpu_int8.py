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
