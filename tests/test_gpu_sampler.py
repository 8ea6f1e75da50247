"""GPU parity: device sampler (mi_sample_tokens) vs oracle/ref_sampler.py through the C ABI.

Integer result (a token id): compared exactly, except on rows the float64 oracle itself flags as a coin flip
in the last ulp — a top-p decision within 1e-5 (relative) of the threshold or an exponential race whose two
best scores are within 1e-4 of each other (the kernel uses fp32 `__expf` / `__logf`).  Such rows must still
land on a token the oracle keeps, and there must be few of them.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_sampler

pytestmark = pytest.mark.gpu


def _run(dev, logits, temperature, top_k, top_p, rep, bitmap, seen_row, seed, offset, update=True):
    from vllm_omni_amd import ops

    t = lambda a, dt: None if a is None else torch.as_tensor(np.asarray(a), dtype=dt).to(dev)  # noqa: E731
    bm = None if bitmap is None else torch.as_tensor(bitmap.astype(np.int64).astype(np.uint32).view(np.int32)).to(dev)
    ids = ops.sample_tokens(logits.to(dev), t(temperature, torch.float32), t(seed, torch.int64), t(offset, torch.int64),
                            top_k=t(top_k, torch.int32), top_p=t(top_p, torch.float32), rep_penalty=t(rep, torch.float32),
                            seen_bitmap=bm, seen_row=t(seen_row, torch.int32), update_seen=update)
    torch.cuda.synchronize()
    return ids.cpu().numpy(), (None if bm is None else bm.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("vocab", [8448, 3072, 1000])
def test_sampler_matches_oracle(dev, vocab, dtype):
    rng = np.random.default_rng(vocab)
    n = 96
    logits = torch.from_numpy((rng.standard_normal((n, vocab)) * 3).astype(np.float32)).to(dtype)
    logits[:, 5:9] = -1e9 if dtype != torch.float16 else float("-inf")       # bad-word mask already applied upstream
    lf = logits.float().numpy()
    temperature = rng.choice([0.0, 0.5, 0.9, 1.3], size=n).astype(np.float32)
    top_k = rng.choice([0, 1, 5, 40, vocab + 3], size=n).astype(np.int32)
    top_p = rng.choice([1.0, 0.95, 0.8, 0.3], size=n).astype(np.float32)
    rep = rng.choice([1.0, 1.05, 1.5], size=n).astype(np.float32)
    words = (vocab + 31) // 32
    bitmap = rng.integers(0, 2**32, size=(n + 3, words), dtype=np.uint64).astype(np.uint32)
    bitmap &= rng.integers(0, 2**32, size=bitmap.shape, dtype=np.uint64).astype(np.uint32)  # ~25 % of tokens seen
    seen_row = rng.permutation(n + 3)[:n].astype(np.int32)
    seed = rng.integers(0, 2**63, size=n, dtype=np.int64)
    offset = rng.integers(0, 2**40, size=n, dtype=np.int64)
    got, bm_after = _run(dev, logits, temperature, top_k, top_p, rep, bitmap, seen_row, seed, offset)
    ref, margins = ref_sampler.sample(lf, temperature, top_k, top_p, rep, bitmap, seen_row, seed, offset)
    flips = 0
    for r in range(n):
        if got[r] == ref[r]:
            continue
        m = margins[r]
        assert m["topp_gap"] < 1e-5 or m["race_ratio"] < 1 + 1e-4, (r, int(got[r]), int(ref[r]), m, float(temperature[r]),
                                                                      int(top_k[r]), float(top_p[r]))
        flips += 1
    assert flips <= 2, flips
    # the drawn token is recorded in the request's seen set, nothing else changes
    expect = bitmap.copy()
    for r in range(n):
        expect[seen_row[r], got[r] >> 5] |= np.uint32(1 << (int(got[r]) & 31))
    assert np.array_equal(bm_after, expect)


_EXTRA = int(os.environ.get("MI_FUZZ_EXTRA", "0"))   # further seeds of the sweep below, run by hand on a GPU box


@pytest.mark.parametrize("seed", list(range(6)) + list(range(100, 100 + _EXTRA)))
def test_sampler_random_sweep(dev, seed):
    """Seeded sweep over what moves the selection code: vocabulary sizes around the wave / block / word boundaries up to
    the kernel's limit, flat to very peaked rows (how many candidates survive the radix passes — the compacted per-wave
    lists or the block-wide fallback), logits on a coarse grid (exact ties at the k-th value and at the top-p cut),
    top-k / top-p pairs incl. the code predictor's 50 / 0.8."""
    rng = np.random.default_rng(4000 + seed)
    vocab = int(rng.choice([33, 64, 65, 255, 1000, 2048, 3072, 4097, 8448, 15744]))
    n = 48
    scale = rng.choice([0.05, 1.0, 3.0, 12.0], size=(n, 1))
    x = rng.standard_normal((n, vocab)) * scale
    grid = rng.random(n) < 0.3
    x[grid] = np.round(x[grid] * 2) / 2                       # exact ties
    logits = torch.from_numpy(x.astype(np.float32))
    if rng.random() < 0.5:
        logits = logits.to(torch.bfloat16)                    # bf16 logits tie often by themselves
    lf = logits.float().numpy()
    temperature = rng.choice([0.0, 0.3, 0.9, 1.0, 2.0], size=n).astype(np.float32)
    top_k = rng.choice([0, 1, 2, 50, 64, 65, 300, vocab - 1, vocab, vocab + 7], size=n).astype(np.int32)
    top_p = rng.choice([1.0, 0.999, 0.8, 0.5, 0.05], size=n).astype(np.float32)
    seed_v = rng.integers(0, 2**63, size=n, dtype=np.int64)
    offset = rng.integers(0, 2**40, size=n, dtype=np.int64)
    got, _ = _run(dev, logits, temperature, top_k, top_p, None, None, None, seed_v, offset)
    ref, margins = ref_sampler.sample(lf, temperature, top_k, top_p, None, None, None, seed_v, offset)
    flips = 0
    for r in range(n):
        if got[r] == ref[r]:
            continue
        m = margins[r]
        assert m["topp_gap"] < 1e-5 or m["race_ratio"] < 1 + 1e-4, (
            seed, vocab, r, int(got[r]), int(ref[r]), m, float(temperature[r]), int(top_k[r]), float(top_p[r]), float(scale[r, 0]),
            bool(grid[r]))
        flips += 1
    assert flips <= 2, flips


def test_sampler_top_k_with_masked_logits_and_crowded_buckets(dev):
    """What the bucketed top-k select (csrc/sampler.hip) must get right besides the sweep above: rows that are mostly -inf
    (a grammar / codec mask: -inf sits in bucket 0 with the smallest finite values, and k may reach into it), rows whose
    k-th value lies in a crowded bucket (> 64 members: the radix select takes over), rows of one repeated finite value."""
    rng = np.random.default_rng(77)
    vocab, n = 2048, 40
    x = rng.standard_normal((n, vocab)).astype(np.float32) * 2
    for r in range(n):
        frac = (0.5, 0.9, 0.97, 0.999)[r % 4]
        x[r, rng.random(vocab) < frac] = -np.inf
    x[5, :] = -np.inf
    x[5, 7] = 1.0                                      # a single finite logit
    x[6, :] = 0.25                                     # every logit the same
    x[7, :1500] = 3.0                                  # a crowded top bucket: k falls among 1500 equal values
    x[8, 100:400] = np.linspace(2.0, 2.001, 300, dtype=np.float32)   # 300 distinct values inside one bucket
    x[8, 0] = 40.0                                     # ... of a wide range
    logits = torch.from_numpy(x)
    temperature = np.full(n, 0.9, np.float32)
    top_k = rng.choice([1, 5, 50, 300, 2047], size=n).astype(np.int32)
    top_k[7], top_k[8] = 50, 150
    top_p = rng.choice([1.0, 0.8], size=n).astype(np.float32)
    seed = np.arange(n, dtype=np.int64) + 5
    offset = np.arange(n, dtype=np.int64) * 3
    got, _ = _run(dev, logits, temperature, top_k, top_p, None, None, None, seed, offset)
    ref, margins = ref_sampler.sample(x, temperature, top_k, top_p, None, None, None, seed, offset)
    for r in range(n):
        if got[r] != ref[r]:
            m = margins[r]
            assert m["topp_gap"] < 1e-5 or m["race_ratio"] < 1 + 1e-4, (r, int(got[r]), int(ref[r]), m)
    assert np.isfinite(x[np.arange(n), got]).all()     # a masked token is never drawn
    assert got[5] == 7


def test_sampler_signed_zeros_tie_and_the_race_tail(dev):
    """Two things the sweep above found.  (1) -0.0 and +0.0 are one value: a top-k whose k-th value is a zero keeps the
    zeros of either sign (they used to get different sort keys).  (2) The exponential race is decided by the draws next
    to u = 1: Philox words whose top 24 bits are 2^24 - 1 .. 2^24 - 3 (found by search for seed 1234: offset, column
    below) must WIN a flat row by a wide margin — computed from a rounded u they scored -inf or were tens of percent off."""
    vocab = 8448
    n = 64
    row = np.full(vocab, -3.0, np.float32)
    row[100:140:2] = 0.0
    row[101:141:2] = -0.0
    logits = torch.from_numpy(np.tile(row, (n, 1)))
    temperature = np.full(n, 0.7, np.float32)
    top_k = np.full(n, 3, np.int32)
    seed = np.full(n, 99, np.int64)
    offset = np.arange(n, dtype=np.int64)
    got, _ = _run(dev, logits, temperature, top_k, None, None, None, None, seed, offset)
    ref, _ = ref_sampler.sample(logits.numpy(), temperature, top_k, None, None, None, None, seed, offset)
    assert np.array_equal(got, ref)
    assert set(got.tolist()) <= set(range(100, 140)) and (got % 2 == 1).any() and (got % 2 == 0).any()

    tail = [(3, 2647), (996, 1849), (1988, 1493), (2655, 3412)]     # (offset, column): 24-bit draw >= 2^24 - 3
    flat = torch.ones(len(tail), vocab)
    offs = np.array([o for o, _ in tail], np.int64)
    seeds = np.full(len(tail), 1234, np.int64)
    got, _ = _run(dev, flat, np.ones(len(tail), np.float32), None, None, None, None, None, seeds, offs)
    ref, margins = ref_sampler.sample(flat.numpy(), np.ones(len(tail), np.float32), None, None, None, None, None, seeds, offs)
    assert ref.tolist() == [c for _, c in tail] and all(m["race_ratio"] > 50 for m in margins)
    assert got.tolist() == ref.tolist()


def test_sampler_draw_frequencies_follow_the_kept_distribution(dev):
    """Size-independent property: over many draw offsets the empirical frequencies of the exponential race
    equal the renormalised top-k/top-p distribution (chi-square, 5 sigma)."""
    vocab, n = 3072, 4096
    rng = np.random.default_rng(7)
    row = (rng.standard_normal(vocab) * 2.5).astype(np.float32)
    logits = torch.from_numpy(np.tile(row, (n, 1)))
    temperature = np.full(n, 0.9, np.float32)
    top_k = np.full(n, 40, np.int32)
    top_p = np.full(n, 0.8, np.float32)
    seed = np.full(n, 1234, np.int64)
    offset = np.arange(n, dtype=np.int64)
    got, _ = _run(dev, logits, temperature, top_k, top_p, None, None, None, seed, offset)
    v = row.astype(np.float64) / 0.9
    kth = np.sort(v)[-40]
    v[v < kth] = -np.inf
    p = np.exp(v - v.max())
    p /= p.sum()
    order = np.argsort(-p)
    before = np.cumsum(p[order]) - p[order]
    kept = order[before < 0.8]
    q = p[kept] / p[kept].sum()
    assert set(np.unique(got)) <= set(kept.tolist())
    counts = np.array([(got == i).sum() for i in kept])
    chi2 = ((counts - n * q) ** 2 / (n * q)).sum()
    dof = len(kept) - 1
    assert chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_sample_tokens_embed_equals_the_two_launches(dev, dtype):
    """The code predictor's per-step draw (``mi_sample_tokens_embed``): ids, both embedding copies and the strided id store
    are what ``sample_tokens`` followed by ``embed_codes`` produce, bit for bit — sampled rows and greedy rows (T = 0)."""
    from vllm_omni_amd import ops

    rows, vocab, dim, g = 37, 2048, 1024, 5
    gen = torch.Generator(device=dev).manual_seed(17)
    logits = (torch.randn(rows, vocab, device=dev, generator=gen) * 3).to(dtype)
    table = torch.randn(vocab, dim, device=dev, generator=gen).to(dtype)
    temp = torch.ones(rows, device=dev)
    temp[::4] = 0.0
    topk = torch.full((rows,), 50, dtype=torch.int32, device=dev)
    topp = torch.full((rows,), 0.8, device=dev)
    seed = torch.arange(rows, dtype=torch.int64, device=dev) + 3
    off = torch.arange(rows, dtype=torch.int64, device=dev) * 7
    want = ops.sample_tokens(logits, temp, seed, off, top_k=topk, top_p=topp)
    buf_a = torch.zeros(rows, g, dim, dtype=dtype, device=dev)
    buf_b = torch.zeros(rows, dim, dtype=dtype, device=dev)
    codes = torch.full((rows, g), -1, dtype=torch.int64, device=dev)
    ops.embed_codes(want, table, buf_a[:, 2], buf_b, codes[:, 3])
    got_a, got_b, got_codes = torch.zeros_like(buf_a), torch.zeros_like(buf_b), torch.full_like(codes, -1)
    got = ops.sample_tokens_embed(logits, temp, seed, off, topk, topp, table, got_a[:, 2], got_b, got_codes[:, 3])
    assert torch.equal(got, want) and torch.equal(got_a, buf_a) and torch.equal(got_b, buf_b) and torch.equal(got_codes, codes)
    assert torch.equal(got_a[:, 2], table[want])
    # without the second copy and the id store
    only_a = torch.zeros_like(buf_b)
    assert torch.equal(ops.sample_tokens_embed(logits, temp, seed, off, topk, topp, table, only_a), want)
    assert torch.equal(only_a, table[want])


def test_sampler_reproducible_and_edge_cases(dev):
    from vllm_omni_amd import _C, ops

    vocab, n = 1000, 8
    rng = np.random.default_rng(3)
    logits = torch.from_numpy(rng.standard_normal((n, vocab)).astype(np.float32))
    temperature = np.full(n, 1.0, np.float32)
    seed = np.arange(n, dtype=np.int64)
    offset = np.zeros(n, np.int64)
    a, _ = _run(dev, logits, temperature, None, None, None, None, None, seed, offset)
    b, _ = _run(dev, logits, temperature, None, None, None, None, None, seed, offset)
    c, _ = _run(dev, logits, temperature, None, None, None, None, None, seed, offset + 1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)            # same (seed, offset) -> same draw
    # top_k = 1 and top_p -> 0 are greedy whatever the noise; T = 0 ignores top-k/top-p
    g = logits.argmax(-1).numpy()
    k1, _ = _run(dev, logits, temperature, np.ones(n, np.int32), None, None, None, None, seed, offset)
    p0, _ = _run(dev, logits, temperature, None, np.full(n, 1e-6, np.float32), None, None, None, seed, offset)
    t0, _ = _run(dev, logits, np.zeros(n, np.float32), np.full(n, 5, np.int32), np.full(n, 0.5, np.float32), None, None,
                 None, seed, offset)
    assert np.array_equal(k1, g) and np.array_equal(p0, g) and np.array_equal(t0, g)
    # zero rows: no launch; oversize vocabulary and CPU tensors are refused, nothing falls back
    empty = ops.sample_tokens(torch.empty(0, vocab, device=dev), torch.empty(0, device=dev),
                              torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.int64, device=dev))
    assert empty.numel() == 0
    big = torch.zeros(1, 20000, device=dev)
    one = lambda dt: torch.zeros(1, dtype=dt, device=dev)  # noqa: E731
    with pytest.raises(_C.MIError) as ei:
        ops.sample_tokens(big, one(torch.float32), one(torch.int64), one(torch.int64))
    assert ei.value.code == _C.MI_ERR_UNSUPPORTED
    with pytest.raises(RuntimeError):
        ops.sample_tokens(torch.zeros(1, 10), torch.zeros(1), torch.zeros(1, dtype=torch.int64),
                          torch.zeros(1, dtype=torch.int64))
