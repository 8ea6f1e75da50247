"""GPU: ``mi_rope_cache_attend_block`` — (q/k-norm ->) RoPE -> K/V append -> attention in one launch for sequences
that live in ONE KV block (the Qwen3-Omni code predictor's steps, SURVEY §8f N4;
vllm_omni/model_executor/models/common/qwen3_code_predictor.py:318-626).

Oracle: ``ref_ops.head_rms_norm`` -> ``ref_ops.rope_neox`` -> ``ref_ops.reshape_and_cache`` ->
``ref_ops.paged_attention_decode`` (fp32 softmax and P.V on the 16-bit q / k / v).  Tolerance: north_star's rtol 1e-3
plus the unit roundoff of the 16-bit output (2^-8 bf16 / 2^-11 f16), atol 1e-4 — the bound of the general decode
kernel (tests/test_gpu_attention_decode.py).  The q/k-norm + rotation in front adds two rounding points where an fp32
sum in a different order can move a q or k element by one ulp; rows where that happened are compared against the
oracle attending over the DEVICE's own rotated q / cache rows (the general pair of launches has the same property,
tests/test_gpu_layer_ops.py::test_qk_norm_rope)."""
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu


def _tol(dtype):
    return 1e-3 + (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11), 1e-4


def _setup(dev, dtype, T, nq, nkv, bs, positions, norm, seed):
    d = 128
    g = torch.Generator().manual_seed(seed)
    nb = T + 3
    kc = torch.randn(nb, bs, nkv, d, generator=g).to(dtype)
    vc = torch.randn(nb, bs, nkv, d, generator=g).to(dtype)
    qkv = torch.randn(T, (nq + 2 * nkv) * d, generator=g).to(dtype)
    qw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype) if norm else None
    kw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype) if norm else None
    blocks = torch.randperm(nb - 1, generator=g)[:T].to(torch.int32) + 1     # one block per row, scattered
    pos = torch.as_tensor(positions, dtype=torch.int64)
    csc = ref_ops.build_cos_sin_cache(d, 128, 1e4)
    return d, kc, vc, qkv, qw, kw, blocks, pos, csc


def _oracle(dtype, d, kc, vc, qkv, qw, kw, blocks, pos, csc, nq, nkv, scale):
    q, k, v = qkv.split([nq * d, nkv * d, nkv * d], dim=-1)
    if qw is not None:
        q = ref_ops.head_rms_norm(q, qw, d, 1e-6, round_to=dtype)
        k = ref_ops.head_rms_norm(k, kw, d, 1e-6, round_to=dtype)
    q = ref_ops.rope_neox(pos, q, d, csc, round_to=dtype)
    k = ref_ops.rope_neox(pos, k, d, csc, round_to=dtype)
    kc, vc = kc.clone(), vc.clone()
    bs = kc.shape[1]
    slots = blocks.to(torch.int64) * bs + pos
    ref_ops.reshape_and_cache(k.view(-1, nkv, d).to(dtype), v.reshape(-1, nkv, d), kc, vc, slots)
    out = ref_ops.paged_attention_decode(q.view(-1, nq, d), kc, vc, blocks.view(-1, 1), (pos + 1).to(torch.int32), scale)
    return out.reshape(out.shape[0], -1), q, kc, vc


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nq,nkv", [(16, 8), (4, 4), (8, 2), (8, 1)])
@pytest.mark.parametrize("norm", [True, False])
def test_block_attention_matches_oracle(dev, dtype, nq, nkv, norm):
    from vllm_omni_amd import ops

    bs, T = 64, 11
    positions = [0, 1, 2, 15, 16, 17, 31, 32, 33, 47, 63]           # every tile boundary, first and last slot
    d, kc, vc, qkv, qw, kw, blocks, pos, csc = _setup(dev, dtype, T, nq, nkv, bs, positions, norm, seed=nq * 10 + nkv)
    scale = d ** -0.5
    want, q_ref, kc_ref, vc_ref = _oracle(dtype, d, kc, vc, qkv, qw, kw, blocks, pos, csc, nq, nkv, scale)

    dq = qkv.to(dev)
    q, k, v = dq.split([nq * d, nkv * d, nkv * d], dim=-1)
    dkc, dvc = kc.to(dev), vc.to(dev)
    out = torch.full((T, nq * d), float("nan"), dtype=dtype, device=dev)
    mv = lambda x: None if x is None else x.to(dev)   # noqa: E731
    ops.rope_cache_attend_block(out, q, k, v, pos.to(dev), csc.to(dev), d, dkc, dvc, blocks.to(dev), scale,
                                q_norm_weight=mv(qw), k_norm_weight=mv(kw), norm_eps=1e-6)
    # the inputs are read-only
    assert torch.equal(dq.cpu(), qkv)
    # K/V append: exactly one slot per row changed; V is a copy, K within 2 ulp of the oracle's (mostly equal)
    got_k, got_v = dkc.cpu(), dvc.cpu()
    assert torch.equal(got_v, vc_ref)
    changed = (got_k != kc).any(dim=-1).any(dim=-1)               # [blocks, tokens]
    want_changed = torch.zeros_like(changed)
    want_changed[blocks.long(), pos] = True
    assert torch.equal(changed | want_changed, want_changed)
    # (a one-ulp flip of a normalised value of magnitude 2-4 moves the rotated value by up to 2^-7 whatever ITS size)
    torch.testing.assert_close(got_k.float(), kc_ref.float(), rtol=2 ** -6, atol=2 ** -6 if norm else 1e-3)
    assert (got_k[blocks.long(), pos] == kc_ref[blocks.long(), pos]).float().mean() > (0.97 if norm else 0.999)
    # attention against the oracle over the device's own cache rows (removes the 1-ulp q/k differences of the norm)
    rtol, atol = _tol(dtype)
    if norm:
        again = ref_ops.paged_attention_decode(q_ref.view(-1, nq, d), got_k, got_v, blocks.view(-1, 1),
                                               (pos + 1).to(torch.int32), scale).reshape(T, -1)
        torch.testing.assert_close(out.cpu().float(), again, rtol=3e-2, atol=3e-2)   # q may still differ by an ulp
        assert ((out.cpu().float() - want).abs() <= atol + rtol * want.abs()).float().mean() > 0.97
    else:
        torch.testing.assert_close(out.cpu().float(), want, rtol=rtol, atol=atol)


def test_block_attention_equals_the_general_pair_of_launches(dev):
    """Same inputs through ``rope_neox_and_cache`` + ``paged_attn_decode`` (what the predictor ran before): caches
    end up identical up to the norm's summation order, outputs within the decode tolerance of each other."""
    from vllm_omni_amd import ops

    dtype, nq, nkv, bs, T = torch.bfloat16, 16, 8, 64, 40
    positions = [(7 * i) % 34 for i in range(T)]
    d, kc, vc, qkv, qw, kw, blocks, pos, csc = _setup(dev, dtype, T, nq, nkv, bs, positions, True, seed=5)
    scale = d ** -0.5
    out_a = torch.empty(T, nq * d, dtype=dtype, device=dev)
    ka, va = kc.to(dev), vc.to(dev)
    q, k, v = qkv.to(dev).split([nq * d, nkv * d, nkv * d], dim=-1)
    ops.rope_cache_attend_block(out_a, q, k, v, pos.to(dev), csc.to(dev), d, ka, va, blocks.to(dev), scale,
                                q_norm_weight=qw.to(dev), k_norm_weight=kw.to(dev))
    kb, vb = kc.to(dev), vc.to(dev)
    q2, k2, v2 = [x.clone() for x in qkv.to(dev).split([nq * d, nkv * d, nkv * d], dim=-1)]
    slots = (blocks.to(torch.int64) * bs + pos).to(dev)
    ops.rope_neox_and_cache(q2, k2, v2, pos.to(dev), csc.to(dev), d, kb, vb, slots, q_norm_weight=qw.to(dev),
                            k_norm_weight=kw.to(dev))
    out_b = torch.empty_like(out_a)
    ops.paged_attn_decode(out_b, q2, kb, vb, blocks.to(dev).view(-1, 1), (pos + 1).to(torch.int32).to(dev), 64, scale)
    assert torch.equal(va, vb)
    assert (ka == kb).float().mean() > 0.999
    torch.testing.assert_close(out_a.float(), out_b.float(), rtol=2e-2, atol=2e-2)
    assert (out_a == out_b).float().mean() > 0.9


@pytest.mark.parametrize("p", [0, 1, 5, 16, 17, 31, 32, 33, 48, 63])
def test_block_attention_uniform_position_equals_the_per_row_form(dev, p):
    """``positions`` as a host int (every row at the same position: the code predictor's lock-step sequences, with the
    register tiles sized for exactly that position) == the per-row form with the default bound, bit for bit."""
    from vllm_omni_amd import ops

    dtype, nq, nkv, bs, T = torch.bfloat16, 16, 8, 64, 9
    d, kc, vc, qkv, qw, kw, blocks, pos, csc = _setup(dev, dtype, T, nq, nkv, bs, [p] * T, True, seed=100 + p)
    outs, caches = [], []
    for form in ("rows", "bounded", "uniform"):
        q, k, v = qkv.to(dev).split([nq * d, nkv * d, nkv * d], dim=-1)
        dk, dv = kc.to(dev), vc.to(dev)
        out = torch.empty(T, nq * d, dtype=dtype, device=dev)
        ops.rope_cache_attend_block(out, q, k, v, p if form == "uniform" else pos.to(dev), csc.to(dev), d, dk, dv,
                                    blocks.to(dev), d ** -0.5, q_norm_weight=qw.to(dev), k_norm_weight=kw.to(dev),
                                    max_position_bound=p if form == "bounded" else None)
        outs.append(out.cpu())
        caches.append((dk.cpu(), dv.cpu()))
    for o, (ck, cv) in zip(outs[1:], caches[1:]):
        assert torch.equal(o, outs[0]) and torch.equal(ck, caches[0][0]) and torch.equal(cv, caches[0][1])


@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("p", [0, 9, 16, 33])
def test_block_attention_from_qkv_partials_is_bit_identical(dev, plan_registry, p, bias):
    """qkv_proj's split-K partials straight into the attention launch == ``linear`` then ``rope_cache_attend_block``:
    same output bits, same cache bits (the code predictor's route at <= 32 rows)."""
    from vllm_omni_amd import ops

    dtype, nq, nkv, bs, T, d, H = torch.bfloat16, 16, 8, 64, 24, 128, 1024
    g = torch.Generator().manual_seed(300 + p)
    x = torch.randn(T, H, generator=g).to(dtype).to(dev)
    w = (torch.randn((nq + 2 * nkv) * d, H, generator=g) * H ** -0.5).to(dtype).to(dev)
    b = torch.randn((nq + 2 * nkv) * d, generator=g).to(dtype).to(dev) if bias else None
    _, kc, vc, _, qw, kw, blocks, pos, csc = _setup(dev, dtype, T, nq, nkv, bs, [p] * T, True, seed=400 + p)
    part = ops.linear_partials(x, w, b)
    assert part is not None and part.split > 1
    ka, va = kc.to(dev), vc.to(dev)
    out_a = ops.rope_cache_attend_block_splitk(part, nq, nkv, d, p, csc.to(dev), ka, va, blocks.to(dev), d ** -0.5,
                                               q_norm_weight=qw.to(dev), k_norm_weight=kw.to(dev))
    qkv = ops.linear(x, w, b)
    q, k, v = qkv.split([nq * d, nkv * d, nkv * d], dim=-1)
    kb, vb = kc.to(dev), vc.to(dev)
    out_b = torch.empty(T, nq * d, dtype=dtype, device=dev)
    ops.rope_cache_attend_block(out_b, q, k, v, pos.to(dev), csc.to(dev), d, kb, vb, blocks.to(dev), d ** -0.5,
                                q_norm_weight=qw.to(dev), k_norm_weight=kw.to(dev), max_position_bound=p)
    assert torch.equal(out_a, out_b) and torch.equal(ka, kb) and torch.equal(va, vb)


def test_block_attention_rejects_what_it_does_not_cover(dev):
    from vllm_omni_amd import ops
    from vllm_omni_amd._C import MIError

    assert not ops.block_attention_supported(28, 4, 128, 16, 128)      # 7 q heads per KV head
    assert not ops.block_attention_supported(16, 8, 64, 64, 64)        # head_dim 64
    assert not ops.block_attention_supported(16, 8, 128, 128, 128)     # block longer than 64 tokens
    assert ops.block_attention_supported(16, 8, 128, 64, 128)
    d = 64
    kc = torch.zeros(2, 16, 2, d, dtype=torch.bfloat16, device=dev)
    q = torch.zeros(1, 4 * d, dtype=torch.bfloat16, device=dev)
    k = torch.zeros(1, 2 * d, dtype=torch.bfloat16, device=dev)
    csc = ref_ops.build_cos_sin_cache(d, 32, 1e4).to(dev)
    with pytest.raises(MIError):
        ops.rope_cache_attend_block(torch.empty_like(q), q, k, k.clone(), torch.zeros(1, dtype=torch.int64, device=dev), csc, d,
                                    kc, kc.clone(), torch.ones(1, dtype=torch.int32, device=dev), 1.0)
