"""GPU parity, randomised: decode and prefill attention against the oracle over seeded random geometries —
head counts and GQA groups, head_dim 64/128, block sizes 16..128, the three cache layouts, both dtypes, ragged
lengths that straddle block / step / partition boundaries, forced partition sizes, host upper bounds larger than
the longest sequence.  Complements the hand-picked cases of test_gpu_attention_{decode,prefill}.py."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops
from test_gpu_attention_decode import _mk_case, _run, _tol

pytestmark = pytest.mark.gpu

# MI_FUZZ_EXTRA=<n>: n further seeds per test with LONGER geometries (contexts of thousands of tokens, chunks of hundreds)
# — a sweep to run by hand on a GPU box (`MI_FUZZ_EXTRA=64 pytest tests/test_gpu_attention_fuzz.py`); 0 by default
_EXTRA = int(os.environ.get("MI_FUZZ_EXTRA", "0"))


def _geometry(rng):
    d = int(rng.choice([64, 128]))
    nkv = int(rng.choice([1, 2, 4]))
    group = int(rng.choice([1, 2, 4, 7, 8, 16]))
    bs = int(rng.choice([16, 32, 64, 128]))
    return d, nkv, group * nkv, bs


@pytest.mark.parametrize("seed", list(range(24)) + list(range(100, 100 + _EXTRA)))
def test_decode_random_geometry(dev, seed):
    rng = np.random.default_rng(1000 + seed)
    d, nkv, nq, bs = _geometry(rng)
    S = int(rng.integers(1, 6))
    edges = [1, bs - 1, bs, bs + 1, 31, 32, 33, 255, 256, 257, 511, 513]
    hi = 700
    if seed >= 100:
        edges += [1023, 1024, 1025, 2047, 2049, 4095, 4097]
        hi = 6000
    lens = [int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(1, hi)) for _ in range(S)]
    dtype = torch.bfloat16 if rng.random() < 0.5 else torch.float16
    layout = str(rng.choice(["nb2", "2nb", "head_major"]))
    kv, kb, vb, table, q, seq_lens = _mk_case(S, nq, nkv, d, bs, lens, dtype, seed=seed, layout=layout)
    ref = ref_ops.paged_attention_decode(q, kb, vb, table, seq_lens, d ** -0.5)
    part = int(rng.choice([0, 0, 32, 64, 128, 256] + ([512, 1024] if seed >= 100 else [])))
    msl = max(lens) + int(rng.choice([0, 0, 1, 100]))
    for mode in ("exact", "fast"):
        got = _run(dev, kv, layout, table, q, seq_lens, torch.float32, partition_size=part, max_seq_len=msl, mode=mode)
        rtol, atol = _tol(mode, dtype)
        torch.testing.assert_close(got, ref, rtol=rtol, atol=atol,
                                   msg=lambda m: f"[{mode}] d={d} nq={nq} nkv={nkv} bs={bs} lens={lens} part={part} {layout}: {m}")


@pytest.mark.parametrize("seed", list(range(16)) + list(range(100, 100 + _EXTRA)))
def test_prefill_random_geometry(dev, seed):
    from vllm_omni_amd import ops

    rng = np.random.default_rng(2000 + seed)
    d, nkv, nq, bs = _geometry(rng)
    S = int(rng.integers(1, 4))
    ctxs = [int(rng.choice([0, 0, 1, bs - 1, bs, 31, 33, 100])) for _ in range(S)]
    chunks = [int(rng.choice([1, 15, 16, 17, 31, 32, 33, 64, 90])) for _ in range(S)]
    if seed >= 100:
        ctxs = [int(rng.choice([0, 1, 127, 128, 129, 511, 1000, 2049, 3000])) if rng.random() < 0.6 else int(rng.integers(0, 3000))
                for _ in range(S)]
        chunks = [int(rng.choice([1, 8, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 400])) if rng.random() < 0.6
                  else int(rng.integers(1, 400)) for _ in range(S)]
    dtype = torch.bfloat16 if rng.random() < 0.5 else torch.float16
    g = torch.Generator().manual_seed(seed)
    lens = [c + q for c, q in zip(ctxs, chunks)]
    max_blocks = max((n + bs - 1) // bs for n in lens)
    nb = S * max_blocks + 4
    head_major = rng.random() < 0.5
    if head_major:
        kv = torch.randn(nb, 2, nkv, bs, d, generator=g).to(dtype)
        kb, vb = kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)
    else:
        kv = torch.randn(nb, 2, bs, nkv, d, generator=g).to(dtype)
        kb, vb = kv[:, 0], kv[:, 1]
    table = (torch.randperm(nb - 1, generator=g)[: S * max_blocks] + 1).view(S, max_blocks).to(torch.int32)
    T = sum(chunks)
    q = torch.randn(T, nq, d, generator=g).to(dtype)
    qsl = torch.tensor([0] + list(np.cumsum(chunks)), dtype=torch.int32)
    seq_lens = torch.tensor(lens, dtype=torch.int32)
    ref = ref_ops.paged_attention_prefill(q, kb, vb, table, seq_lens, qsl, d ** -0.5)
    kvg = kv.to(dev)
    kbg, vbg = (kvg[:, 0].permute(0, 2, 1, 3), kvg[:, 1].permute(0, 2, 1, 3)) if head_major else (kvg[:, 0], kvg[:, 1])
    out = torch.full((T, nq * d), float("nan"), dtype=dtype, device=dev)
    atol = 8e-3 if dtype == torch.bfloat16 else 1e-3
    # once as a whole, once with every tile's context cut into a random number of segments (merged on the device)
    for segments in (1, int(torch.randint(2, 6, (1,), generator=g))):
        out.fill_(float("nan"))
        ops.paged_attn_prefill(out, q.view(T, -1).to(dev), kbg, vbg, table.to(dev), seq_lens.to(dev), qsl.to(dev), max(chunks),
                               d ** -0.5, max_seq_len=max(lens), segments=segments)
        torch.cuda.synchronize()
        got = out.cpu().float().view(T, nq, d)
        assert torch.isfinite(got).all()
        rtol, atol = (1e-3 + (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11), 1e-4)   # fp32 bar + the output rounding
        torch.testing.assert_close(got, ref, rtol=rtol, atol=atol,
                                   msg=lambda m: f"d={d} nq={nq} nkv={nkv} bs={bs} ctxs={ctxs} chunks={chunks} hm={head_major} "
                                                 f"segments={segments}: {m}")
