"""GPU parity: paged-attention decode vs the CPU oracle, through the C ABI.

Tolerances (north_star: "fp16/bf16 logits within 1e-3 rtol"): the kernel is asked for an
fp32 output so accumulation accuracy is measured without the final 16-bit rounding:
  * exact mode (VALU kernel, P kept fp32) AND fast mode (the production matrix-core kernel: P carried
    as a hi + lo pair of cache-dtype values) vs the fp32 oracle: rtol 1e-3, atol 1e-4 — the north_star bar;
  * packed_valu / rounded_p (A/B forms that round P ONCE to the cache dtype before PV, as flash
    kernels do; not what the product runs where the matrix-core kernel is built) vs the fp32 oracle:
    rtol 1e-3 + atol 6e-3 (bf16) / 1e-3 (f16) — the P rounding itself is worth up to 2^-9 relative per
    term (|out| <= max|V| ~ 4);
  * 16-bit output == round(fp32 output) bit-exactly.
Block-table indexing is checked bit-exactly with integer-valued V (one-hot attention).
"""
import math

import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu


def _mk_case(S, nq, nkv, d, bs, seq_lens, dtype, seed=0, nb_extra=7, layout="nb2"):
    g = torch.Generator().manual_seed(seed)
    max_blocks = max(1, max((n + bs - 1) // bs for n in seq_lens))
    nb = S * max_blocks + nb_extra + 1
    if layout == "nb2":      # [nb, 2, bs, H, d]  (K/V at dim 1)
        kv = torch.randn(nb, 2, bs, nkv, d, generator=g).to(dtype)
        kb, vb = kv[:, 0], kv[:, 1]
    elif layout == "2nb":    # [2, nb, bs, H, d]  (K/V at dim 0)
        kv = torch.randn(2, nb, bs, nkv, d, generator=g).to(dtype)
        kb, vb = kv[0], kv[1]
    else:                    # head-major physical layout, exposed as the same logical view
        kv = torch.randn(nb, 2, nkv, bs, d, generator=g).to(dtype)
        kb, vb = kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)
    perm = torch.randperm(nb - 1, generator=g)[: S * max_blocks] + 1  # block 0 = null block
    table = perm.view(S, max_blocks).to(torch.int32)
    q = torch.randn(S, nq, d, generator=g).to(dtype)
    return kv, kb, vb, table, q, torch.tensor(seq_lens, dtype=torch.int32)


# fast = production: MFMA kernel (split P) where built, else the fp32-P VALU kernel
MODES = ["exact", "packed_valu", "fast", "rounded_p"]


def _tol(mode, dtype):
    """(rtol, atol) vs the fp32 oracle.  exact / fast: the north_star bar.  packed_valu / rounded_p (A/B forms):
    P rounded once to the cache dtype (2^-9 / 2^-12 relative per term, |out| <= max|V| ~ 4)."""
    if mode in ("exact", "fast"):
        return 1e-3, 1e-4
    return (1e-3, 6e-3) if dtype == torch.bfloat16 else (1e-3, 1e-3)


def _run(dev, kv, layout, table, q, seq_lens, out_dtype, partition_size=0, max_seq_len=None, mode="exact"):
    from vllm_omni_amd import ops

    kvg = kv.to(dev)
    if layout == "nb2":
        kb, vb = kvg[:, 0], kvg[:, 1]
    elif layout == "2nb":
        kb, vb = kvg[0], kvg[1]
    else:
        kb, vb = kvg[:, 0].permute(0, 2, 1, 3), kvg[:, 1].permute(0, 2, 1, 3)
    S, nq, d = q.shape
    out = torch.full((S, nq * d), float("nan"), dtype=out_dtype, device=dev)
    msl = int(seq_lens.max()) if max_seq_len is None else max_seq_len
    ops.paged_attn_decode(out, q.view(S, nq * d).to(dev), kb, vb, table.to(dev), seq_lens.to(dev),
                          max(msl, 1), d ** -0.5, partition_size=partition_size, packed_pv=mode != "exact",
                          force_valu=mode == "packed_valu", rounded_p=mode == "rounded_p")
    torch.cuda.synchronize()
    return out.cpu().view(S, nq, d)


CASES = [
    # S, nq, nkv, d, bs, seq_lens                      (golden-vector shapes of SURVEY §8c ii)
    (3, 8, 2, 128, 16, [1, 17, 40]),
    (3, 8, 2, 64, 16, [1, 17, 40]),
    (4, 28, 4, 128, 16, [5, 16, 33, 300]),             # S-A head geometry (G = 7)
    (2, 16, 2, 64, 16, [64, 129]),                     # S-B head geometry (G = 8, d = 64)
    (2, 7, 1, 128, 16, [31, 257]),                     # S-A at TP=4
    (3, 4, 4, 128, 8, [9, 8, 23]),                     # MHA, block 8
    (2, 12, 2, 128, 32, [70, 33]),                     # G = 6, block 32
    (2, 32, 2, 128, 16, [18, 47]),                     # G = 16 -> two waves of 8 heads per KV head
    (2, 5, 1, 64, 8, [3, 41]),                         # G = 5, d = 64, block 8
    (3, 16, 2, 64, 16, [1, 33, 300]),                  # S-B, long enough for several MFMA steps/partitions
    (2, 2, 1, 64, 32, [257, 64]),                      # S-B at TP=8 (replicated KV head), block 32
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_decode_matches_oracle(dev, case, dtype):
    S, nq, nkv, d, bs, lens = case
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, lens, dtype)
    ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
    for mode in MODES:
        got = _run(dev, kv, "nb2", table, q, seq_lens, torch.float32, mode=mode)
        rtol, atol = _tol(mode, dtype)
        torch.testing.assert_close(got, ref, rtol=rtol, atol=atol, msg=lambda m: f"[{mode}] {m}")
        # production output dtype: exactly the rounding of the fp32 result
        got16 = _run(dev, kv, "nb2", table, q, seq_lens, dtype, mode=mode)
        assert torch.equal(got16, got.to(dtype)), mode


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("layout", ["nb2", "2nb", "head_major"])
def test_decode_cache_layouts(dev, layout, mode):
    """The three physical layouts behind the reference's accepted views (kv_utils.py:378-450)."""
    S, nq, nkv, d, bs, lens = 3, 8, 2, 128, 16, [1, 17, 40]
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, lens, torch.bfloat16, seed=3, layout=layout)
    got = _run(dev, kv, layout, table, q, seq_lens, torch.float32, mode=mode)
    ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
    rtol, atol = _tol(mode, torch.bfloat16)
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("partition_size", [16, 32, 64, 256, 1024])
def test_decode_partitions_agree(dev, partition_size, mode):
    """Split-context + merge must not change the answer (beyond fp32 reassociation)."""
    S, nq, nkv, d, bs = 3, 28, 4, 128, 16
    lens = [1, 130, 517]
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, lens, torch.bfloat16, seed=5)
    ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
    rtol, atol = _tol(mode, torch.bfloat16)
    got = _run(dev, kv, "nb2", table, q, seq_lens, torch.float32, partition_size=partition_size, mode=mode)
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)
    if mode == "exact":  # fp32 P: partitions only reassociate fp32 sums
        one = _run(dev, kv, "nb2", table, q, seq_lens, torch.float32, partition_size=1024, mode=mode)
        torch.testing.assert_close(got, one, rtol=1e-4, atol=1e-4)
    # a launch sized for a longer bound (graph replay while sequences grow) gives the same result
    got2 = _run(dev, kv, "nb2", table, q, seq_lens, torch.float32, partition_size=partition_size,
                max_seq_len=2048, mode=mode)
    assert torch.equal(got2, got)


@pytest.mark.parametrize("mode", MODES)
def test_decode_block_table_indexing_bit_exact(dev, mode):
    """One-hot attention over integer-valued V: the output IS the row the block table points at."""
    S, nq, nkv, d, bs = 4, 8, 2, 128, 16
    lens = [16, 17, 63, 100]
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, lens, torch.bfloat16, seed=7)
    kv.zero_()
    g = torch.Generator().manual_seed(11)
    targets = [int(torch.randint(0, n, (1,), generator=g)) for n in lens]
    q.zero_()
    q[..., 0] = 64.0  # score = 64 * k[0] * d^-0.5
    for s, (n, tgt) in enumerate(zip(lens, targets)):
        for t in range(n):
            blk, off = int(table[s, t // bs]), t % bs
            vb[blk, off] = torch.randint(-8, 8, (nkv, d), generator=g).to(torch.bfloat16)
            kb[blk, off, :, 0] = 200.0 if t == tgt else -200.0
    got = _run(dev, kv, "nb2", table, q, seq_lens, torch.bfloat16, mode=mode)
    for s, tgt in enumerate(targets):
        blk, off = int(table[s, tgt // bs]), tgt % bs
        exp = vb[blk, off].repeat_interleave(nq // nkv, dim=0)
        assert torch.equal(got[s], exp), f"seq {s}: wrong row gathered"


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("partition_size", [16, 32])
def test_decode_ignores_garbage_past_seq_len(dev, mode, partition_size):
    """NaN/Inf in cache rows beyond seq_len (same block, or unallocated table entries) must not leak."""
    S, nq, nkv, d, bs = 2, 8, 2, 128, 16
    lens = [5, 37]
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, lens, torch.bfloat16, seed=9)
    ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
    for s, n in enumerate(lens):
        blk = int(table[s, (n - 1) // bs])
        kb[blk, n % bs:] = float("nan")
        vb[blk, n % bs:] = float("inf")
    table[0, 1:] = -12345  # never-allocated entries: must not be dereferenced
    got = _run(dev, kv, "nb2", table, q, seq_lens, torch.float32, partition_size=partition_size, max_seq_len=64,
               mode=mode)
    assert torch.isfinite(got).all()
    rtol, atol = _tol(mode, torch.bfloat16)
    torch.testing.assert_close(got, ref, rtol=rtol, atol=max(atol, 1e-3))


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_decode_empty_and_zero_len(dev, mode):
    from vllm_omni_amd import ops

    S, nq, nkv, d, bs = 3, 8, 2, 128, 16
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, [0, 20, 0], torch.bfloat16)
    for ps in (0, 16, 32):
        got = _run(dev, kv, "nb2", table, q, seq_lens, torch.float32, partition_size=ps, max_seq_len=64, mode=mode)
        assert torch.equal(got[0], torch.zeros(nq, d)) and torch.equal(got[2], torch.zeros(nq, d))
        ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
        rtol, atol = _tol(mode, torch.bfloat16)
        torch.testing.assert_close(got[1], ref[1], rtol=rtol, atol=max(atol, 1e-3))
    # zero sequences: no launch, no error
    e = torch.empty(0, nq * d, dtype=torch.bfloat16, device=dev)
    ops.paged_attn_decode(e, e, kv.to(dev)[:, 0], kv.to(dev)[:, 1], torch.empty(0, 4, dtype=torch.int32, device=dev),
                          torch.empty(0, dtype=torch.int32, device=dev), 16, 1.0)


@pytest.mark.parametrize("mode", MODES)
def test_decode_softmax_extremes(dev, mode):
    """Running-max rescale path: a late, much larger score must take over exactly (rule 26)."""
    S, nq, nkv, d, bs = 1, 8, 2, 128, 16
    n = 200
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, [n], torch.bfloat16, seed=13)
    # spike key at a late token so every earlier tile is rescaled by ~exp(-large)
    for t in (150, 199):
        blk, off = int(table[0, t // bs]), t % bs
        kb[blk, off] = (q[0, ::nq // nkv] * 3.0).to(torch.bfloat16)
    ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
    for ps in (16, 64, 1024):
        got = _run(dev, kv, "nb2", table, q, seq_lens, torch.float32, partition_size=ps, mode=mode)
        assert torch.isfinite(got).all()
        rtol, atol = _tol(mode, torch.bfloat16)
        torch.testing.assert_close(got, ref, rtol=rtol, atol=max(atol, 1e-3))


def test_decode_rejects_bad_arguments(dev):
    from vllm_omni_amd import _C, ops

    S, nq, nkv, d, bs = 2, 8, 2, 128, 16
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, [4, 4], torch.bfloat16)
    kvg = kv.to(dev)
    out = torch.empty(S, nq * d, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError):  # CPU tensors: no fallback
        ops.paged_attn_decode(out.cpu(), q.view(S, -1), kb, vb, table, seq_lens, 4, 1.0)
    bad = torch.randn(4, 2, bs, nkv, 96).to(torch.bfloat16).to(dev)  # head_dim 96 not built
    with pytest.raises(_C.MIError) as ei:
        ops.paged_attn_decode(torch.empty(S, nq * 96, dtype=torch.bfloat16, device=dev),
                              torch.empty(S, nq * 96, dtype=torch.bfloat16, device=dev), bad[:, 0], bad[:, 1],
                              table.to(dev), seq_lens.to(dev), 4, 1.0)
    assert ei.value.code == _C.MI_ERR_UNSUPPORTED
    # workspace too small -> MI_ERR_WORKSPACE, nothing launched
    qg, tg, lg = q.view(S, -1).to(dev), table.to(dev), seq_lens.to(dev)
    code = ops.lib.mi_paged_attn_decode(out.data_ptr(), qg.data_ptr(), kvg[:, 0].data_ptr(), kvg[:, 1].data_ptr(),
                                        tg.data_ptr(), lg.data_ptr(), S, nq, nkv, d, bs, tg.stride(0), 64, 1.0,
                                        nq * d, nq * d, kvg.stride(0), kvg.stride(2), kvg.stride(3), 16, None, 0,
                                        _C.MI_DTYPE_BF16, _C.MI_DTYPE_BF16, 0, None)
    assert code == _C.MI_ERR_WORKSPACE
    # misaligned stride -> MI_ERR_INVALID_ARG
    code = ops.lib.mi_paged_attn_decode(out.data_ptr(), qg.data_ptr(), kvg[:, 0].data_ptr(), kvg[:, 1].data_ptr(),
                                        tg.data_ptr(), lg.data_ptr(), S, nq, nkv, d, bs, tg.stride(0), 4, 1.0,
                                        nq * d + 1, nq * d, kvg.stride(0), kvg.stride(2), kvg.stride(3), 16, None, 0,
                                        _C.MI_DTYPE_BF16, _C.MI_DTYPE_BF16, 0, None)
    assert code == _C.MI_ERR_INVALID_ARG


def test_fused_merge_equals_the_two_kernel_merge_bit_for_bit(dev):
    """The partitions merged inside the attention launch (last-arriving wave of each (sequence, head group), arrival
    counters at the front of the workspace) against decode_reduce_kernel: identical bits — on ONE workspace reused by
    launches of different batch sizes, partition sizes and context bounds (the counters must come back to zero every
    time), with empty sequences in the batch, and replayed from a hipGraph while the sequences grow."""
    from vllm_omni_amd import ops

    nq, nkv, d, bs = 28, 4, 128, 16
    lens_all = [1, 0, 130, 517, 64, 1000, 33, 0, 999, 256]
    kv, kb, vb, table, q, seq_lens = _mk_case(len(lens_all), nq, nkv, d, bs, [max(n, 1) for n in lens_all],
                                              torch.bfloat16, seed=21)
    seq_lens = torch.tensor(lens_all, dtype=torch.int32)
    kvg = kv.to(dev)
    kbg, vbg = kvg[:, 0], kvg[:, 1]
    ws = ops.new_decode_workspace(ops.decode_workspace_bytes(len(lens_all), nq, d, 2048, 32), dev)

    def run(S, part, msl, out_dtype, fused, workspace):
        out = torch.full((S, nq * d), float("nan"), dtype=out_dtype, device=dev)
        ops.paged_attn_decode(out, q[:S].reshape(S, nq * d).to(dev), kbg, vbg, table[:S].to(dev), seq_lens[:S].to(dev),
                              msl, d ** -0.5, partition_size=part, workspace=workspace, fused_merge=fused)
        return out

    for rep in range(3):
        for S, part, msl in ((10, 256, 1024), (3, 32, 2048), (10, 64, 1000), (7, 1024, 2048), (10, 128, 1040)):
            for out_dtype in (torch.float32, torch.bfloat16):
                two = run(S, part, msl, out_dtype, False, None)
                one = run(S, part, msl, out_dtype, True, ws)
                torch.cuda.synchronize()
                assert torch.equal(one, two), (rep, S, part, msl, out_dtype)
                assert not torch.isnan(one.float()).any()
                assert torch.equal(one[1], torch.zeros_like(one[1]))          # seq_len 0: zeros
        counters = ws[:1 << 20].view(torch.int32)
        assert int(counters.abs().sum()) == 0                                  # every launch left them at zero
    with pytest.raises(ValueError, match="new_decode_workspace"):
        run(3, 32, 2048, torch.float32, True, torch.empty(ws.numel(), dtype=torch.uint8, device=dev))
    # hipGraph replay with growing sequences: same launch, new seq_lens
    S, part, msl = 10, 256, 1024
    sl = seq_lens.to(dev)
    tb, qq = table.to(dev), q.reshape(len(lens_all), nq * d).to(dev)
    out = torch.zeros(S, nq * d, dtype=torch.bfloat16, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.paged_attn_decode(out, qq, kbg, vbg, tb, sl, msl, d ** -0.5, partition_size=part, workspace=ws)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.paged_attn_decode(out, qq, kbg, vbg, tb, sl, msl, d ** -0.5, partition_size=part, workspace=ws)
    for step in range(4):
        sl.copy_(torch.tensor([min(n + 7 * step, 1000) if n else 0 for n in lens_all], dtype=torch.int32))
        g.replay()
        torch.cuda.synchronize()
        want = torch.zeros_like(out)
        ops.paged_attn_decode(want, qq, kbg, vbg, tb, sl, msl, d ** -0.5, partition_size=part, fused_merge=False)
        torch.cuda.synchronize()
        assert torch.equal(out, want), step


@pytest.mark.parametrize("bs,d,lens", [(32, 128, [4500, 2047, 2049, 4096]),      # windows of 2048 tokens
                                         (64, 64, [8200, 4095, 4097, 4096]),       # windows of 4096 tokens, head_dim 64
                                         (16, 64, [3000, 1025, 2048, 1024])])
def test_long_partitions_other_block_sizes(dev, bs, d, lens):
    """The window walk at block sizes 32 / 64 (a window is 64 BLOCKS: 2048 / 4096 tokens) and head_dim 64, one partition
    for the whole context and partitions that end inside a window; layouts head-major and token-major."""
    from vllm_omni_amd import ops

    nq, nkv = 8, 2
    for layout in ("head_major", "nb2"):
        kv, kb, vb, table, q, seq_lens = _mk_case(len(lens), nq, nkv, d, bs, lens, torch.bfloat16, seed=41, layout=layout)
        ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
        for part in (0, 64 * bs + 32 * 3, 16384):
            got = _run(dev, kv, layout, table, q, seq_lens, torch.float32, partition_size=part, max_seq_len=max(lens) + 7,
                       mode="fast")
            torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-4, msg=lambda m: f"bs {bs} {layout} partition {part}: {m}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_long_partitions_walk_the_block_id_windows(dev, dtype):
    """Partitions of more than 64 blocks (the matrix-core kernel reloads its block-id window every 64 blocks), up to ONE
    partition for the whole context (direct output, no merge launch — what the auto choice takes once sequences x KV
    heads >= 512): the oracle's bar, and one-hot targets on both sides of every window boundary gather the right row
    bit-exactly."""
    from vllm_omni_amd import ops

    nq, nkv, d, bs = 8, 2, 128, 16
    lens = [4096, 1023, 1025, 2048 + 31, 3000, 64 * 16 * 3 + 1]
    kv, kb, vb, table, q, seq_lens = _mk_case(len(lens), nq, nkv, d, bs, lens, dtype, seed=31)
    ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
    kvg = kv.to(dev)
    S = len(lens)

    def run(part, out_dtype=torch.float32, kbv=None, qq=None):
        out = torch.full((S, nq * d), float("nan"), dtype=out_dtype, device=dev)
        k_, v_ = kbv if kbv is not None else (kvg[:, 0], kvg[:, 1])
        ops.paged_attn_decode(out, (q if qq is None else qq).view(S, nq * d).to(dev), k_, v_, table.to(dev),
                              seq_lens.to(dev), 4096 + 16, d ** -0.5, partition_size=part)
        torch.cuda.synchronize()
        return out.cpu().view(S, nq, d)

    for part in (2048, 4096 + 32, 1024 + 512, 8192):
        got = run(part)
        torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-4, msg=lambda m: f"partition {part}: {m}")
        assert torch.equal(run(part, dtype), got.to(dtype))
    # one-hot attention over integer-valued V: tokens 1023 / 1024 (window 0 | 1), 2047 / 2048, 3071 / 3072, 4095
    targets = [4095, 1022, 1024, 2048, 2047, 3072]
    kd = torch.zeros_like(kv[:, 0]).float()
    vd = torch.zeros_like(kv[:, 1]).float()
    expect = []
    for s, (n, tgt) in enumerate(zip(lens, targets)):
        assert tgt < n
        for t in range(n):
            blk, off = int(table[s, t // bs]), t % bs
            kd[blk, off, :, 0] = 200.0 if t == tgt else -200.0
            vd[blk, off] = float((t * 7) % 251 - 125)
        expect.append(float((tgt * 7) % 251 - 125))
    qq = torch.zeros(S, nq, d)
    qq[:, :, 0] = 1.0
    got = run(8192, dtype, kbv=(kd.to(dtype).to(dev), vd.to(dtype).to(dev)), qq=qq.to(dtype)).float()
    for s, e in enumerate(expect):
        assert torch.equal(got[s], torch.full((nq, d), e)), (s, targets[s])


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: RoPE + KV write inside the decode launch (mi_rope_cache_paged_attn_decode; qwen2_old.py:153-154 as ONE launch)
ROPE_CASES = [
    # S, nq, nkv, d, bs, seq_lens (INCLUDING this step's token), partition_size
    (5, 28, 4, 128, 16, [1, 16, 33, 300, 0], 0),          # S-A geometry; first token of a sequence; an inert padded row
    (4, 7, 1, 128, 16, [32, 64, 1025, 2048], 1024),       # TP-4 shard: the token opens / ends a partition and a 32-token step
    (4, 7, 1, 128, 16, [31, 65, 1024, 2049], 256),        # many partitions: exactly one wave per KV head owns the write
    (3, 16, 2, 64, 16, [1, 33, 300], 0),                  # S-B geometry (d = 64)
    (3, 32, 2, 128, 32, [18, 47, 129], 64),               # G = 16: two waves per KV head, block 32
    (2, 8, 8, 128, 16, [17, 40], 0),                      # MHA
]


@pytest.mark.parametrize("qk_norm", [False, True], ids=["plain", "qk-norm"])
@pytest.mark.parametrize("layout", ["nb2", "head_major"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ROPE_CASES)
def test_rope_fused_decode_equals_the_two_launches_bit_for_bit(dev, case, dtype, layout, qk_norm):
    """q rotated in registers + the token's k / v written by the owning wave + attention == rope_neox_and_cache followed
    by paged_attn_decode: output, K cache and V cache bit for bit (and therefore inside the oracle tolerance the two
    launches are tested at), un-rotated q left alone, rows with slot -1 write nothing."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.layers import build_cos_sin_cache
    S, nq, nkv, d, bs, lens, ps = case
    kv, kb, vb, table, _, seq_lens = _mk_case(S, nq, nkv, d, bs, [max(n, 1) for n in lens], dtype, seed=11, layout=layout)
    seq_lens = torch.tensor(lens, dtype=torch.int32)
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(S, (nq + 2 * nkv) * d, generator=g).to(dtype).to(dev)
    csc = build_cos_sin_cache(d, 8192, 1e6, device=dev)
    pos = torch.tensor([max(n - 1, 0) for n in lens], dtype=torch.int64)
    slots = torch.tensor([int(table[i, (n - 1) // bs]) * bs + (n - 1) % bs if n > 0 else -1 for i, n in enumerate(lens)],
                         dtype=torch.int64)
    msl = max(max(lens), 1)
    # the Qwen3 family's per-head q / k RMSNorm in front of the rotation (modeling_qwen3_omni_moe.py:1451-1456)
    qn = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(dev) if qk_norm else None
    kn = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(dev) if qk_norm else None

    def caches():
        kvg = kv.to(dev).clone()
        if layout == "nb2":
            return kvg, kvg[:, 0], kvg[:, 1]
        return kvg, kvg[:, 0].permute(0, 2, 1, 3), kvg[:, 1].permute(0, 2, 1, 3)

    outs = []
    for fused in (False, True):
        store, kc, vc = caches()
        row = qkv.clone()
        q, k, v = row.split([nq * d, nkv * d, nkv * d], dim=-1)
        out = torch.full((S, nq * d), float("nan"), dtype=dtype, device=dev)
        if fused:
            assert ops.rope_decode_fusable(q, kc, pos, d, ps, S, msl)
            ops.rope_cache_paged_attn_decode(out, q, k, v, pos.to(dev), csc, kc, vc, slots.to(dev), table.to(dev),
                                             seq_lens.to(dev), msl, d ** -0.5, partition_size=ps, q_norm_weight=qn,
                                             k_norm_weight=kn, norm_eps=1e-6)
            assert torch.equal(row, qkv)                      # q / k / v rows are inputs only
        else:
            ops.rope_neox_and_cache(q, k, v, pos.to(dev), csc, d, kc, vc, slots.to(dev), q_norm_weight=qn, k_norm_weight=kn,
                                    norm_eps=1e-6)
            ops.paged_attn_decode(out, q, kc, vc, table.to(dev), seq_lens.to(dev), msl, d ** -0.5, partition_size=ps)
        torch.cuda.synchronize()
        outs.append((out.cpu(), store.cpu()))
    assert torch.equal(outs[0][1], outs[1][1]), "cache contents differ"
    assert not torch.equal(outs[1][1], kv), "nothing was written"
    assert torch.equal(outs[0][0], outs[1][0]), "attention outputs differ"
    if qk_norm:
        return   # the two launches with q/k-norm have their own oracle test (test_gpu_layer_ops.py); equality is the claim here
    # and against the oracle chain on the same inputs
    q0, k0, v0 = qkv.cpu().split([nq * d, nkv * d, nkv * d], dim=-1)
    live = [i for i, n in enumerate(lens) if n > 0]
    table_cpu = ref_ops.build_cos_sin_cache(d, 8192, 1e6)
    qr = ref_ops.rope_neox(pos, q0, d, table_cpu, round_to=dtype).view(S, nq, d)
    kr = ref_ops.rope_neox(pos, k0, d, table_cpu, round_to=dtype).view(S, nkv, d)
    kb2, vb2 = kb.clone(), vb.clone()
    for i in live:
        blk, off = int(slots[i]) // bs, int(slots[i]) % bs
        kb2[blk, off] = kr[i]
        vb2[blk, off] = v0[i].view(nkv, d)
    ref = ref_ops.paged_attention_decode(qr, kb2, vb2, table, seq_lens, d ** -0.5)
    got = outs[1][0].view(S, nq, d).float()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for i in live:
        torch.testing.assert_close(got[i], ref[i], rtol=1e-3 + ulp, atol=1e-4 + ulp * float(ref[i].abs().max()))


def test_rope_fused_decode_replays_from_a_graph_while_sequences_grow(dev):
    """The fused launch inside a hipGraph: each replay appends one token per sequence (slot, position and seq_len are
    device buffers the host rewrites) and must equal the eager two-launch step on a twin cache — eight steps that cross a
    KV-block and a 32-token step boundary."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.layers import build_cos_sin_cache
    S, nq, nkv, d, bs, dtype = 6, 7, 1, 128, 16, torch.bfloat16
    start = [13, 30, 31, 47, 255, 1000]
    g = torch.Generator().manual_seed(3)
    nblk = (max(start) + 16) // bs + 1
    table = (torch.randperm(S * nblk, generator=g) + 1).view(S, nblk).to(torch.int32).to(dev)
    kc_a = torch.randn(S * nblk + 1, bs, nkv, d, generator=g).to(dtype).to(dev)
    vc_a = torch.randn(S * nblk + 1, bs, nkv, d, generator=g).to(dtype).to(dev)
    kc_b, vc_b = kc_a.clone(), vc_a.clone()
    csc = build_cos_sin_cache(d, 4096, 1e6, device=dev)
    qkv = torch.randn(S, (nq + 2) * d, generator=g).to(dtype).to(dev)
    pos = torch.zeros(S, dtype=torch.int64, device=dev)
    slots = torch.zeros(S, dtype=torch.int64, device=dev)
    lens = torch.zeros(S, dtype=torch.int32, device=dev)
    out_g = torch.zeros(S, nq * d, dtype=dtype, device=dev)
    msl = 2048
    q, k, v = qkv.split([nq * d, d, d], dim=-1)

    def set_step(i):
        n = [s + i + 1 for s in start]
        pos.copy_(torch.tensor([x - 1 for x in n]))
        lens.copy_(torch.tensor(n, dtype=torch.int32))
        tb = table.cpu()
        slots.copy_(torch.tensor([int(tb[j, (x - 1) // bs]) * bs + (x - 1) % bs for j, x in enumerate(n)]))

    set_step(0)
    ws = ops.new_decode_workspace(ops.decode_workspace_bytes(S, nq, d, msl, ops.decode_auto_partition(S, nkv, msl, bs)), dev)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.rope_cache_paged_attn_decode(out_g, q, k, v, pos, csc, kc_a, vc_a, slots, table, lens, msl, d ** -0.5, workspace=ws)
    torch.cuda.current_stream().wait_stream(st)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.rope_cache_paged_attn_decode(out_g, q, k, v, pos, csc, kc_a, vc_a, slots, table, lens, msl, d ** -0.5, workspace=ws)
    for i in range(8):
        set_step(i)
        qkv.copy_(torch.randn(S, (nq + 2) * d, generator=g).to(dtype))
        graph.replay()
        row = qkv.clone()
        q2, k2, v2 = row.split([nq * d, d, d], dim=-1)
        ops.rope_neox_and_cache(q2, k2, v2, pos, csc, d, kc_b, vc_b, slots)
        out_e = torch.empty_like(out_g)
        ops.paged_attn_decode(out_e, q2, kc_b, vc_b, table, lens, msl, d ** -0.5)
        torch.cuda.synchronize()
        assert torch.equal(out_g, out_e), f"step {i}"
        assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b), f"step {i}"


def test_rope_fused_decode_refuses_what_it_does_not_cover(dev):
    from vllm_omni_amd import ops
    from vllm_omni_amd._C import MIError
    from vllm_omni_amd.layers import build_cos_sin_cache
    S, nq, nkv, d = 2, 8, 2, 128
    csc = build_cos_sin_cache(d, 64, 1e4, device=dev)
    pos2 = torch.zeros(3, S, dtype=torch.int64, device=dev)
    kc8 = torch.zeros(8, 8, nkv, d, dtype=torch.bfloat16, device=dev)      # block 8: the VALU kernel's shape
    kc16 = torch.zeros(8, 16, nkv, d, dtype=torch.bfloat16, device=dev)
    assert not ops.rope_decode_fusable(None, kc8, pos2[0], d, 0, S, 40)
    assert not ops.rope_decode_fusable(None, kc16, pos2, d, 0, S, 40)       # M-RoPE rows
    assert not ops.rope_decode_fusable(None, kc16, pos2[0], 64, 0, S, 40)   # partial rotary
    assert ops.rope_decode_fusable(None, kc16, pos2[0], d, 0, S, 40)
    q = torch.zeros(S, nq * d, dtype=torch.bfloat16, device=dev)
    kk = torch.zeros(S, nkv * d, dtype=torch.bfloat16, device=dev)
    tb = torch.ones(S, 4, dtype=torch.int32, device=dev)
    sl = torch.ones(S, dtype=torch.int32, device=dev)
    with pytest.raises(MIError):
        ops.rope_cache_paged_attn_decode(torch.empty_like(q), q, kk, kk, pos2[0], csc, kc8, kc8.clone(),
                                         torch.zeros(S, dtype=torch.int64, device=dev), tb, sl, 40, 1.0)
