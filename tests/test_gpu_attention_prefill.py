"""GPU parity: paged-attention prefill / chunked prefill vs the CPU oracle, through
the C ABI.

Tolerance: north_star's bar — rtol 1e-3, atol 1e-4 against the fp32 oracle — on the kernel's fp32 result (P enters the
PV MFMA as a hi + lo pair of cache-dtype values); the op writes its output in the cache dtype, so the comparison adds
exactly that one rounding: the unit roundoff of the output grid, 2^-8 (bf16) / 2^-11 (f16) relative.  ``_tol``.
The single-rounding P of flash kernels (MI_PREFILL_ROUNDED_P=1, A/B only) is checked once at its own, looser bound.
"""
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu


def _case(dev, nq, nkv, bs, ctxs, chunks, dtype, seed=0, d=128, segments=1):  # noqa: D401
    """ctxs[i] tokens already cached + chunks[i] new query tokens per sequence.  ``segments``: 1 = the context of a
    tile is never split; 0 = automatic; n > 1 = forced (ops.paged_attn_prefill)."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(seed)
    S = len(ctxs)
    lens = [c + q for c, q in zip(ctxs, chunks)]
    max_blocks = max(1, max((n + bs - 1) // bs for n in lens))
    nb = S * max_blocks + 5
    kv = torch.randn(nb, 2, bs, nkv, d, generator=g).to(dtype)
    table = (torch.randperm(nb - 1, generator=g)[: S * max_blocks] + 1).view(S, max_blocks).to(torch.int32)
    T = sum(chunks)
    q = torch.randn(T, nq, d, generator=g).to(dtype)
    qsl = torch.tensor([0] + list(torch.tensor(chunks).cumsum(0)), dtype=torch.int32)
    seq_lens = torch.tensor(lens, dtype=torch.int32)
    ref = ref_ops.paged_attention_prefill(q, kv[:, 0], kv[:, 1], table, seq_lens, qsl, d ** -0.5)
    kvg = kv.to(dev)
    out = torch.full((T, nq * d), float("nan"), dtype=dtype, device=dev)
    ops.paged_attn_prefill(out, q.view(T, -1).to(dev), kvg[:, 0], kvg[:, 1], table.to(dev), seq_lens.to(dev),
                           qsl.to(dev), max(chunks), d ** -0.5, max_seq_len=max(lens), segments=segments)
    torch.cuda.synchronize()
    return out.cpu().float().view(T, nq, d), ref


def _tol(dtype):
    """(rtol, atol): 1e-3 / 1e-4 on the fp32 result + the unit roundoff of the 16-bit output grid (2^-8 bf16, 2^-11 f16)."""
    return (1e-3 + (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11), 1e-4)


CASES = [
    # nq, nkv, bs, ctxs, chunks
    (8, 2, 16, [0], [19]),                       # plain prefill, ragged tile
    (8, 2, 16, [19], [13]),                      # chunked prefill with context (SURVEY §8c iii)
    (8, 2, 16, [40, 5, 19], [1, 1, 13]),         # two decodes + one prefill chunk in one call
    (28, 4, 16, [0, 7], [16, 17]),               # S-A head geometry, tile-boundary lengths
    (7, 1, 16, [100], [33]),                     # S-A at TP=4
    (4, 4, 32, [3], [70]),                       # MHA, block 32
    (32, 2, 16, [10], [20]),                     # GQA group 16 -> two workgroups of 8 heads
    (28, 4, 16, [500], [300]),                   # long context, many steps
    (12, 2, 16, [0, 0, 0], [1, 15, 64]),         # G = 6, varlen
]
CASES_D64 = [
    (16, 2, 16, [0], [19]),                      # S-B head geometry (Qwen3-Omni talker: 16/2 heads, d = 64)
    (16, 2, 16, [19, 40], [13, 1]),              # chunk with context + a decode row
    (2, 1, 32, [70], [50]),                      # S-B at TP=8 (KV head replicated), block 32
    (8, 8, 16, [0, 0], [33, 16]),                # MHA
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_prefill_matches_oracle(dev, case, dtype):
    nq, nkv, bs, ctxs, chunks = case
    got, ref = _case(dev, nq, nkv, bs, ctxs, chunks, dtype)
    assert torch.isfinite(got).all()
    rtol, atol = _tol(dtype)
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES_D64)
def test_prefill_head_dim_64(dev, case, dtype):
    nq, nkv, bs, ctxs, chunks = case
    got, ref = _case(dev, nq, nkv, bs, ctxs, chunks, dtype, d=64)
    assert torch.isfinite(got).all()
    rtol, atol = _tol(dtype)
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_prefill_flattened_columns_many_sequences(dev, dtype):
    """A launch with >= 2 workgroups per CU keeps the whole 7-head GQA group in one workgroup, and the 32x32x16 kernel
    then lays its 7 x 36 (head, query) pairs head-major over 8 waves (csrc/paged_attn_prefill.hip): a 32-column tile
    mixes two heads, tiles are 36 tokens.  48 ragged sequences (chunks 37..120 tokens: partial last tiles of every
    length class; contexts 0..200) x 28/4 heads = 768 workgroups, every row compared with the oracle."""
    g = torch.Generator().manual_seed(11)
    n = 48
    ctxs = torch.randint(0, 201, (n,), generator=g).tolist()
    chunks = torch.randint(37, 121, (n,), generator=g).tolist()
    chunks[0], chunks[1], chunks[2] = 120, 72, 73          # the longest (4 tiles), an exact multiple of 36, one past it
    got, ref = _case(dev, 28, 4, 16, ctxs, chunks, dtype, seed=12)
    assert torch.isfinite(got).all()
    rtol, atol = _tol(dtype)
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("nq,nkv", [(10, 2), (12, 2), (6, 2), (5, 1), (3, 1)])
def test_prefill_gqa_groups_that_are_not_powers_of_two(dev, nq, nkv, d):
    """G = 5, 6, 3 on a small grid: the group is split over two workgroups (3 + 2, 3 + 3 heads) or kept (3), and 3
    heads run as 4 waves x 32 columns = 3 x 42 tokens; 80-token chunks leave a 38-token last tile.  Both head sizes of
    the 32x32x16 kernel (128-B rows at head_dim 64: its own K / V images and fragment reads)."""
    got, ref = _case(dev, nq, nkv, 16, [0, 25, 90], [80, 43, 7], torch.bfloat16, seed=13, d=d)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, rtol=_tol(torch.bfloat16)[0], atol=_tol(torch.bfloat16)[1])


SPLIT_CASES = [
    # nq, nkv, d, ctxs, chunks, forced segments
    (28, 4, 128, [1000], [77], 4),                 # a ragged last chunk on a long context (32x32x16 kernel, 36-token tiles)
    (28, 4, 128, [300, 0, 2100], [40, 130, 5], 3),   # sequences of very different lengths: some tiles fit one segment
    (7, 1, 128, [513], [129], 5),                  # one KV head (a TP=4 rank); a segment boundary inside a tile's diagonal
    (8, 8, 128, [250], [20], 2),                   # MHA: one-wave workgroups
    (16, 2, 64, [900], [64], 4),                   # head_dim 64: the 32x32x16 kernel on 128-B rows
    (16, 2, 64, [700, 64], [9, 16], 3),            # chunks <= 16 tokens: 16-token tiles
    (28, 4, 128, [2000], [8], 6),                  # a speculative-verify-sized chunk on a long context (16-token tiles)
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", SPLIT_CASES)
def test_prefill_with_context_segments(dev, case, dtype):
    """The context of every query tile cut into segments run by separate workgroups and merged (fp32 partials + log-sum-exp,
    csrc/paged_attn_prefill.hip): against the oracle at the unsplit kernel's tolerance, and against the unsplit kernel
    itself (the same products, one more fp32 re-association).  Segment boundaries fall inside tiles' causal diagonals,
    so some query columns see no key at all of a later segment."""
    nq, nkv, d, ctxs, chunks, nseg = case
    got, ref = _case(dev, nq, nkv, 16, ctxs, chunks, dtype, seed=31, d=d, segments=nseg)
    one, _ = _case(dev, nq, nkv, 16, ctxs, chunks, dtype, seed=31, d=d, segments=1)
    assert torch.isfinite(got).all()
    rtol, atol = _tol(dtype)
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)
    # split and unsplit round the same fp32 quantity up to re-association: at most one 16-bit ulp apart
    torch.testing.assert_close(got, one, rtol=2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10, atol=1e-3)


def test_prefill_automatic_segments_on_a_small_launch(dev):
    """No forcing: 77 new tokens on 6000 cached ones are 3 tiles x 4 KV heads = 12 workgroups, so the launch policy splits
    the context (and leaves a big launch alone: the workspace query returns 0 for the 8k prompt)."""
    from vllm_omni_amd import _C

    got, ref = _case(dev, 28, 4, 16, [6000], [77], torch.bfloat16, seed=33, segments=0)
    torch.testing.assert_close(got, ref, rtol=_tol(torch.bfloat16)[0], atol=_tol(torch.bfloat16)[1])
    assert _C.lib.mi_paged_attn_prefill_workspace_bytes(77, 1, 77, 6077, 28, 4, 128, 0) > 0
    assert _C.lib.mi_paged_attn_prefill_workspace_bytes(8192, 1, 8192, 8192, 28, 4, 128, 0) == 0
    assert _C.lib.mi_paged_attn_prefill_workspace_bytes(77, 1, 77, 6077, 28, 4, 128, 1) == 0


def test_prefill_segments_with_a_low_context_bound(dev):
    """``max_seq_len`` is a host-side bound; if it is too LOW the last segment still runs to the end of every tile (the
    launch is less balanced, no key is dropped)."""
    from vllm_omni_amd import ops

    nq, nkv, d, bs = 28, 4, 128, 16
    ctx, chunk = 3000, 50
    g = torch.Generator().manual_seed(41)
    total = ctx + chunk
    nblk = (total + bs - 1) // bs
    kv = torch.randn(nblk + 2, 2, bs, nkv, d, generator=g).to(torch.bfloat16)
    table = (torch.randperm(nblk + 1, generator=g)[:nblk] + 1).view(1, -1).to(torch.int32)
    q = torch.randn(chunk, nq, d, generator=g).to(torch.bfloat16)
    sl, qsl = torch.tensor([total], dtype=torch.int32), torch.tensor([0, chunk], dtype=torch.int32)
    ref = ref_ops.paged_attention_prefill(q, kv[:, 0], kv[:, 1], table, sl, qsl, d ** -0.5)
    kvg = kv.to(dev)
    out = torch.full((chunk, nq * d), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.paged_attn_prefill(out, q.view(chunk, -1).to(dev), kvg[:, 0], kvg[:, 1], table.to(dev), sl.to(dev), qsl.to(dev),
                           chunk, d ** -0.5, max_seq_len=1024, segments=4)   # the sequence is three times as long
    torch.testing.assert_close(out.cpu().float().view(chunk, nq, d), ref, rtol=_tol(torch.bfloat16)[0], atol=_tol(torch.bfloat16)[1])


def test_prefill_equals_decode_for_single_token_chunks(dev):
    """q_len == 1 everywhere: the prefill kernel must agree with the decode kernel (same P rounding)."""
    from vllm_omni_amd import ops

    nq, nkv, bs, d = 28, 4, 16, 128
    ctxs, chunks = [31, 64, 200], [1, 1, 1]
    got, ref = _case(dev, nq, nkv, bs, ctxs, chunks, torch.bfloat16, seed=4)
    torch.testing.assert_close(got, ref, rtol=_tol(torch.bfloat16)[0], atol=_tol(torch.bfloat16)[1])


def test_prefill_causality(dev):
    """Changing K/V rows a query may not see must not change that query's output (bit-exact)."""
    from vllm_omni_amd import ops

    nq, nkv, bs, d, L = 8, 2, 16, 128, 40
    g = torch.Generator().manual_seed(7)
    nb = 8
    kv = torch.randn(nb, 2, bs, nkv, d, generator=g).to(torch.bfloat16)
    table = torch.tensor([[3, 1, 5]], dtype=torch.int32)
    q = torch.randn(L, nq * d, generator=g).to(torch.bfloat16)
    args = (table.to(dev), torch.tensor([L], dtype=torch.int32, device=dev),
            torch.tensor([0, L], dtype=torch.int32, device=dev), L, d ** -0.5)
    outs = []
    for poison in (False, True):
        kv2 = kv.clone()
        if poison:  # rows 24.. of the sequence (block 1 offset 8.., block 5) -> only queries >= 24 may change
            kv2[1, :, 8:] = 7.0
            kv2[5] = float("nan")  # beyond L=40 -> rows 40..47 of block 5 and never-read blocks
            kv2[5, :, :8] = 3.0    # rows 32..39
        kvg = kv2.to(dev)
        out = torch.empty(L, nq * d, dtype=torch.bfloat16, device=dev)
        ops.paged_attn_prefill(out, q.to(dev), kvg[:, 0], kvg[:, 1], *args)
        outs.append(out.cpu())
    assert torch.equal(outs[0][:24], outs[1][:24])
    assert not torch.equal(outs[0][24:], outs[1][24:])
    assert torch.isfinite(outs[1].float()).all()


def test_prefill_rejects_unsupported(dev):
    from vllm_omni_amd import _C, ops

    kv = torch.zeros(4, 2, 16, 2, 32, dtype=torch.bfloat16, device=dev)
    q = torch.zeros(5, 8 * 32, dtype=torch.bfloat16, device=dev)
    with pytest.raises(_C.MIError) as ei:
        ops.paged_attn_prefill(torch.empty_like(q), q, kv[:, 0], kv[:, 1], torch.zeros(1, 2, dtype=torch.int32, device=dev),
                               torch.tensor([5], dtype=torch.int32, device=dev),
                               torch.tensor([0, 5], dtype=torch.int32, device=dev), 5, 1.0)
    assert ei.value.code == _C.MI_ERR_UNSUPPORTED   # only head_dim 64 / 128 are built


@pytest.mark.parametrize("d", [128, 64])
def test_prefill_running_max_jumps_late(dev, d):
    """The kernel keeps a per-column REFERENCE maximum and raises it only when a score outgrows it by more than a
    threshold (deferred rescale, csrc/paged_attn_prefill.hip): a rare, data-dependent branch that bounded random data
    never takes (programming guide §5.4 rule 26).  Keys planted late in the context score far above everything before
    them for SOME query rows of a tile (and not for the others), several times in one tile: everything accumulated at
    the old reference must be rescaled exactly once.  Full-tensor comparison with the oracle; also a tile whose FIRST
    step already holds the spike, and growth just under / over the threshold."""
    from vllm_omni_amd import ops

    nq, nkv, bs = 8, 2, 16
    g = torch.Generator().manual_seed(31 + d)
    ctx, chunk = 500, 96
    n = ctx + chunk
    max_blocks = (n + bs - 1) // bs
    nb = max_blocks + 3
    kv = torch.randn(nb, 2, bs, nkv, d, generator=g).to(torch.bfloat16)
    table = (torch.randperm(nb - 1, generator=g)[:max_blocks] + 1).view(1, max_blocks).to(torch.int32)
    q = torch.randn(chunk, nq, d, generator=g).to(torch.bfloat16)

    def plant(tok, qrow, head, gain):
        blk, off = int(table[0, tok // bs]), tok % bs
        kv[blk, 0, off, head // (nq // nkv)] = (q[qrow, head].float() * gain).to(torch.bfloat16)

    plant(3, 5, 0, 2.5)          # inside the first step of the tile: the textbook path
    plant(300, 5, 0, 3.0)        # later and larger: the reference must move (scale ~ 3 |q|^2 d^-0.5 >> threshold)
    plant(450, 5, 0, 4.0)        # and again
    plant(320, 40, 3, 3.0)       # another tile (rows 32-63), another head
    plant(510, 70, 7, 3.5)       # inside the chunk itself (causal region), third tile
    plant(200, 17, 1, 0.33)      # growth of the order of the threshold
    plant(260, 17, 1, 0.36)
    qsl = torch.tensor([0, chunk], dtype=torch.int32)
    seq_lens = torch.tensor([n], dtype=torch.int32)
    ref = ref_ops.paged_attention_prefill(q, kv[:, 0], kv[:, 1], table, seq_lens, qsl, d ** -0.5)
    kvg = kv.to(dev)
    out = torch.full((chunk, nq * d), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.paged_attn_prefill(out, q.view(chunk, -1).to(dev), kvg[:, 0], kvg[:, 1], table.to(dev), seq_lens.to(dev),
                           qsl.to(dev), chunk, d ** -0.5)
    torch.cuda.synchronize()
    got = out.cpu().float().view(chunk, nq, d)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, rtol=_tol(torch.bfloat16)[0], atol=_tol(torch.bfloat16)[1])
    # the planted rows really are dominated by their spike (the branch was exercised, not skipped)
    blk, off = int(table[0, 450 // bs]), 450 % bs
    torch.testing.assert_close(got[5, 0], kv[blk, 1, off, 0].float(), rtol=2e-2, atol=2e-2)


def test_single_rounding_p_is_the_looser_ab_form(dev, monkeypatch):
    """MI_PREFILL_ROUNDED_P=1 (the `dot(P.to(V.dtype), V)` of flash kernels; A/B measurements only) stays inside
    rtol 1e-3 + atol 6e-3 of the fp32 oracle and is measurably further from it than the production hi + lo pair."""
    args = (28, 4, 16, [500], [300], torch.bfloat16)
    exact, ref = _case(dev, *args, seed=3)
    monkeypatch.setenv("MI_PREFILL_ROUNDED_P", "1")
    loose, _ = _case(dev, *args, seed=3)
    torch.testing.assert_close(loose, ref, rtol=1e-3 + 2.0 ** -8, atol=6e-3)
    need = lambda got: float(((got - ref).abs() - (1e-3 + 2.0 ** -8) * ref.abs()).clamp(min=0).max())   # noqa: E731
    assert need(exact) <= 1e-4 < need(loose)
