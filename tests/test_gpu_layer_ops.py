"""GPU parity: KV write, RoPE(+q/k-norm)+cache, RMSNorm, SiLU-gate, logit mask/argmax
 vs the CPU oracle, through the C ABI.

Tolerances: pure copies are bit-exact.  Floating-point ops are compared against the oracle
evaluated with the SAME documented rounding points: results may differ by one
16-bit ulp where an fp32 intermediate lands on a rounding boundary (rsqrt/exp approximations
differ in the last fp32 bits), so the bound is rtol = 2^-7 (bf16) / 2^-10 (f16) elementwise,
and additionally >= 99 % of elements must be bit-identical.
"""
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu

ULP = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}


def _close16(got, exp, dtype, frac_exact=0.99):
    """One 16-bit ulp elementwise; at the headline sizes (~1 M elements per call) two, for the handful of elements (7 of
    917 504 measured) where a 1-ulp difference at the op's first rounding point meets the second one."""
    got, exp = got.float(), exp.float()
    ulps = 2 if got.numel() > 500_000 else 1
    torch.testing.assert_close(got, exp, rtol=ulps * ULP[dtype], atol=1e-6)
    same = (got == exp).float().mean().item()
    assert same >= frac_exact, f"only {same:.4f} of elements bit-identical"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,H", [(1, 3584), (7, 3584), (256, 1024), (3, 8192), (5, 200), (2, 8200), (256, 3584), (512, 1024)])
def test_rms_norm(dev, dtype, T, H):
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(T * 31 + H)
    x = (torch.randn(T, H, generator=g) * 3).to(dtype)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    got = ops.rms_norm(x.to(dev), w.to(dev), 1e-6).cpu()
    _close16(got, ref_ops.rms_norm(x, w, 1e-6, round_to=dtype), dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,H", [(1, 3584), (64, 3584), (9, 1024), (3, 104), (256, 3584), (512, 1024)])
def test_fused_add_rms_norm(dev, dtype, T, H):
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(T + H)
    x = torch.randn(T, H, generator=g).to(dtype)
    r = (torch.randn(T, H, generator=g) * 2).to(dtype)
    w = (1 + 0.01 * torch.randn(H, generator=g)).to(dtype)
    xg, rg = x.to(dev), r.to(dev)
    ops.fused_add_rms_norm(xg, rg, w.to(dev), 1e-6)
    exp_n, exp_r = ref_ops.fused_add_rms_norm(x, r, w, 1e-6, round_to=dtype)
    assert torch.equal(rg.cpu().float(), exp_r), "residual = round(x + r) must be bit-exact"
    _close16(xg.cpu(), exp_n, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,I", [(1, 18944), (33, 18944), (4, 384), (2, 100), (256, 18944), (512, 3072)])
def test_silu_and_mul(dev, dtype, T, I):
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(I)
    x = (torch.randn(T, 2 * I, generator=g) * 4).to(dtype)
    x[0, :4] = torch.tensor([30.0, -30.0, 0.0, -0.0]).to(dtype)  # +-large, zeros (SURVEY §8c vi)
    got = ops.silu_and_mul(x.to(dev)).cpu()
    _close16(got, ref_ops.silu_and_mul(x, round_to=dtype), dtype, frac_exact=0.98)


def _cache(nb, bs, h, d, dtype, dev):
    kv = torch.zeros(nb, 2, bs, h, d, dtype=dtype)
    return kv, kv.to(dev)


@pytest.mark.parametrize("d,bs", [(128, 16), (64, 8), (32, 4)])
def test_reshape_and_cache_bit_exact(dev, d, bs):
    from vllm_omni_amd import ops

    T, h, nb = 23, 4, 12
    g = torch.Generator().manual_seed(d)
    k = torch.randn(T, h * d, generator=g).to(torch.bfloat16)
    v = torch.randn(T, h * d, generator=g).to(torch.bfloat16)
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
    slots[[3, 10]] = -1  # padding tokens are skipped
    kv, kvg = _cache(nb, bs, h, d, torch.bfloat16, dev)
    ops.reshape_and_cache(k.to(dev), v.to(dev), kvg[:, 0], kvg[:, 1], slots.to(dev))
    ref_ops.reshape_and_cache(k.view(T, h, d), v.view(T, h, d), kv[:, 0], kv[:, 1], slots)
    assert torch.equal(kvg.cpu(), kv)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nq,nkv,d,base", [(28, 4, 128, 1e6), (16, 2, 64, 1e4), (8, 2, 32, 1e6)])
def test_rope_and_cache(dev, dtype, nq, nkv, d, base):
    from vllm_omni_amd import ops

    T, bs, nb, maxpos = 9, 16, 8, 32768
    g = torch.Generator().manual_seed(nq)
    q = torch.randn(T, nq * d, generator=g).to(dtype)
    k = torch.randn(T, nkv * d, generator=g).to(dtype)
    v = torch.randn(T, nkv * d, generator=g).to(dtype)
    pos = torch.tensor([0, 1, 4095, 32767, 17, 2, 3, 1000, 31], dtype=torch.int64)
    csc = ref_ops.build_cos_sin_cache(d, maxpos, base)
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
    slots[4] = -1
    kv, kvg = _cache(nb, bs, nkv, d, dtype, dev)
    qg, kg = q.to(dev), k.to(dev)
    ops.rope_neox_and_cache(qg, kg, v.to(dev), pos.to(dev), csc.to(dev), d, kvg[:, 0], kvg[:, 1], slots.to(dev))
    eq = ref_ops.rope_neox(pos, q, d, csc, round_to=dtype)
    ek = ref_ops.rope_neox(pos, k, d, csc, round_to=dtype)
    _close16(qg.cpu(), eq, dtype)
    _close16(kg.cpu(), ek, dtype)
    # the cache holds exactly what the kernel wrote back to k, and v unchanged
    ref_ops.reshape_and_cache(kg.cpu().view(T, nkv, d), v.view(T, nkv, d), kv[:, 0], kv[:, 1], slots)
    assert torch.equal(kvg.cpu(), kv)
    # plain RoPE (no cache) gives the same rotation
    q2, k2 = q.to(dev), k.to(dev)
    ops.rope_neox_and_cache(q2, k2, None, pos.to(dev), csc.to(dev), d)
    assert torch.equal(q2, qg) and torch.equal(k2, kg)


@pytest.mark.parametrize("nq,nkv,d,T", [(28, 4, 128, 256), (16, 2, 64, 512)])
def test_rope_and_cache_at_the_decode_batch_sizes(dev, nq, nkv, d, T):
    """The headline step's launch (256 tokens, S-A heads) and config #5's (512 tokens, S-B heads): positions spread over
    the whole table, every token in its own random slot of a 4096-token-per-sequence-sized pool."""
    from vllm_omni_amd import ops

    dtype, bs, maxpos = torch.bfloat16, 16, 32768
    nb = T * 2 + 3
    g = torch.Generator().manual_seed(T)
    q = torch.randn(T, nq * d, generator=g).to(dtype)
    k = torch.randn(T, nkv * d, generator=g).to(dtype)
    v = torch.randn(T, nkv * d, generator=g).to(dtype)
    pos = torch.randint(0, maxpos, (T,), generator=g, dtype=torch.int64)
    pos[0], pos[1] = 4095, 4096
    csc = ref_ops.build_cos_sin_cache(d, maxpos, 1e6)
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
    kv, kvg = _cache(nb, bs, nkv, d, dtype, dev)
    qg, kg = q.to(dev), k.to(dev)
    ops.rope_neox_and_cache(qg, kg, v.to(dev), pos.to(dev), csc.to(dev), d, kvg[:, 0], kvg[:, 1], slots.to(dev))
    _close16(qg.cpu(), ref_ops.rope_neox(pos, q, d, csc, round_to=dtype), dtype)
    _close16(kg.cpu(), ref_ops.rope_neox(pos, k, d, csc, round_to=dtype), dtype)
    ref_ops.reshape_and_cache(kg.cpu().view(T, nkv, d), v.view(T, nkv, d), kv[:, 0], kv[:, 1], slots)
    assert torch.equal(kvg.cpu(), kv)


def test_mrope_sections(dev):
    """[3, T] positions + sections (HF apply_multimodal_rotary_pos_emb rule); equal rows == 1-D RoPE."""
    from vllm_omni_amd import ops

    dtype, nq, nkv, d, T = torch.bfloat16, 8, 2, 32, 6
    g = torch.Generator().manual_seed(2)
    q = torch.randn(T, nq * d, generator=g).to(dtype)
    k = torch.randn(T, nkv * d, generator=g).to(dtype)
    csc = ref_ops.build_cos_sin_cache(d, 512, 1e6)
    pos3 = torch.stack([torch.arange(T), torch.arange(T) * 3 + 1, torch.arange(T) * 7 + 2]).to(torch.int64)
    sec = [4, 6, 6]
    qg, kg = q.to(dev), k.to(dev)
    ops.rope_neox_and_cache(qg, kg, None, pos3.to(dev), csc.to(dev), d, mrope_section=torch.tensor(sec, dtype=torch.int32))
    _close16(qg.cpu(), ref_ops.rope_neox(pos3, q, d, csc, mrope_section=sec, round_to=dtype), dtype)
    _close16(kg.cpu(), ref_ops.rope_neox(pos3, k, d, csc, mrope_section=sec, round_to=dtype), dtype)
    same = torch.arange(T).to(torch.int64).repeat(3, 1)
    q1, k1, q3, k3 = q.to(dev), k.to(dev), q.to(dev), k.to(dev)
    ops.rope_neox_and_cache(q1, k1, None, same[0].contiguous().to(dev), csc.to(dev), d)
    ops.rope_neox_and_cache(q3, k3, None, same.to(dev), csc.to(dev), d, mrope_section=torch.tensor(sec, dtype=torch.int32))
    assert torch.equal(q1, q3) and torch.equal(k1, k3)


@pytest.mark.parametrize("nq,nkv,d", [(16, 2, 64), (4, 1, 128)])
def test_qk_norm_rope(dev, nq, nkv, d):
    """Qwen3 family: per-head RMSNorm on q and k before RoPE."""
    from vllm_omni_amd import ops

    dtype, T = torch.bfloat16, 5
    g = torch.Generator().manual_seed(d)
    q = torch.randn(T, nq * d, generator=g).to(dtype)
    k = torch.randn(T, nkv * d, generator=g).to(dtype)
    qw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
    kw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
    pos = torch.tensor([0, 3, 77, 4000, 12], dtype=torch.int64)
    csc = ref_ops.build_cos_sin_cache(d, 8192, 1e4)
    qg, kg = q.to(dev), k.to(dev)
    ops.rope_neox_and_cache(qg, kg, None, pos.to(dev), csc.to(dev), d, q_norm_weight=qw.to(dev),
                            k_norm_weight=kw.to(dev), norm_eps=1e-6)
    eq = ref_ops.rope_neox(pos, ref_ops.head_rms_norm(q, qw, d, 1e-6, round_to=dtype), d, csc, round_to=dtype)
    ek = ref_ops.rope_neox(pos, ref_ops.head_rms_norm(k, kw, d, 1e-6, round_to=dtype), d, csc, round_to=dtype)
    # two extra rounding points before the rotation: allow 2 ulp, most elements exact
    torch.testing.assert_close(qg.cpu().float(), eq, rtol=2 ** -6, atol=1e-3)
    torch.testing.assert_close(kg.cpu().float(), ek, rtol=2 ** -6, atol=1e-3)
    assert (qg.cpu().float() == eq).float().mean() > 0.97


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_logits_argmax(dev, dtype):
    """bad_word_processor branches (qwen2_5_omni_talker.py:150-169) + greedy argmax, bit-exact."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.talker import bad_word_mask_spec

    V, T = 8448, 6
    g = torch.Generator().manual_seed(4)
    for suppress, end_id, bos in [(8296, 8294, 8293), (8294, 8294, 8293), (8000, 8294, 8293), (None, 8294, 8293),
                                  (9000, 8294, None)]:
        logits = torch.randn(T, V, generator=g).to(dtype)
        logits[2, 8400] = 50.0   # would win if not masked
        exp = ref_ops.bad_word_processor(logits.clone().float(), suppress, end_id, bos)
        ranges, col = bad_word_mask_spec(V, suppress, end_id, bos)
        lg = logits.to(dev)
        ids = ops.mask_logits_argmax(lg, ranges, col).cpu()
        assert torch.equal(lg.cpu().float(), exp.to(dtype).float())
        assert torch.equal(ids, exp.to(dtype).float().argmax(-1))
