"""GPU parity at the BASELINE.json configs' OWN sizes (VERDICT r1 "Next round" item 1).

The unit-size tests (test_gpu_attention_*.py) never reach the regime bench.py times: ctx 4096 =
4 context partitions x 64 KV blocks (a full 64-lane block-id window, first_blk in {64,128,192},
``v_readlane`` at lane 63), batch 256.  Here the production (`fast`, matrix-core) decode kernel and
the MFMA prefill kernel run at those sizes and are compared

  * with the CPU oracle (``oracle/ref_ops.py``) on the same inputs — all sequences where the batch is
    small, a sample of sequences / heads where the launch is full size (B = 256);
  * bit-exactly where the domain is integer (one-hot attention through the block table, a target in
    every partition, including the last block of a window);
  * through size-independent properties at full size (partition sizes agree, 16-bit output ==
    round(fp32 output), finite).

Every comparison records its OBSERVED max abs / rel error (``gpurun_out/parity_report.jsonl``; copied
to ``profiles/`` as evidence) and asserts against a bound derived from those measurements rather than
a blanket atol: see ``_BOUNDS``.  Reference shapes: vllm_omni/model_executor/models/qwen2_5_omni/
qwen2_old.py:89-106 (head partition, scale).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu

S_A = dict(nq=28, nkv=4, d=128)     # Qwen2.5-Omni talker (SURVEY §8 shape S-A)
S_B = dict(nq=16, nkv=2, d=64)      # Qwen3-Omni talker (S-B)
BS = 16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(name: str, **kv) -> None:
    """Append one line of observed errors to gpurun_out/parity_report.jsonl (best effort)."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **kv}) + "\n")
    except OSError:
        pass
    print(f"[parity] {name}: " + ", ".join(f"{k}={v}" for k, v in kv.items()))


def _errs(got: torch.Tensor, ref: torch.Tensor, out_ulp: float = 0.0) -> dict:
    """max abs error, max error relative to max|ref| of the row (head), and the worst violation of
    |got - ref| <= atol + 1e-3 |ref| expressed as the atol it would need.  ``out_ulp``: half an ulp of the 16-bit grid
    the op rounded its output to (unit roundoff 2^-8 bf16, 2^-11 f16), added to the 1e-3 — for ops that have no fp32 output."""
    diff = (got.double() - ref.double()).abs()
    need_atol = (diff - (1e-3 + out_ulp) * ref.double().abs()).clamp(min=0).max().item()
    scale = ref.double().abs().amax(-1, keepdim=True).clamp(min=1e-30)
    return {"max_abs": float(diff.max()), "max_rel_to_rowmax": float((diff / scale).max()),
            "atol_needed_at_rtol_1e-3": float(need_atol), "ref_absmax": float(ref.abs().max())}


# Bounds on `atol_needed_at_rtol_1e-3`: north_star's bar (rtol 1e-3, atol 1e-4 against the fp32 oracle) for BOTH
# kernels — the VALU kernel keeps P in fp32, the production matrix-core kernel carries it as hi + lo 16-bit pairs.
# Observed values: profiles/r03_parity_report.jsonl.
_BOUNDS = {
    ("exact", torch.bfloat16): 1e-4, ("exact", torch.float16): 1e-4,      # north_star bar: rtol 1e-3, atol 1e-4
    ("fast", torch.bfloat16): 1e-4, ("fast", torch.float16): 1e-4,        # production kernel (hi + lo P): same bar
}


def _device_cache(dev, nb, nkv, d, dtype, seed):
    """Head-major physical cache [nb, 2, nkv, BS, d] filled N(0,1) on the device; returns the logical
    (key_blocks, value_blocks) views [nb, BS, nkv, d] the kernels and the oracle index."""
    g = torch.Generator(device=dev).manual_seed(seed)
    kv = torch.empty(nb, 2, nkv, BS, d, dtype=dtype, device=dev)
    kv.normal_(0.0, 1.0, generator=g)
    return kv, kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)


def _table(S, lens, seed, max_blocks=None):
    g = torch.Generator().manual_seed(seed)
    mb = max_blocks or max(1, max((n + BS - 1) // BS for n in lens))
    nb = S * mb + 9
    table = (torch.randperm(nb - 1, generator=g)[: S * mb] + 1).view(S, mb).to(torch.int32)
    return table, nb


def _oracle_rows(kb, vb, table, lens, q, rows, p_round_to=None):
    """Oracle for the sequences ``rows``: their blocks are gathered to the host and re-numbered."""
    outs = []
    for s in rows:
        n = int(lens[s])
        nblk = (n + BS - 1) // BS
        ids = table[s, :nblk].to(torch.int64)
        kc, vc = kb[ids.to(kb.device)].cpu(), vb[ids.to(vb.device)].cpu()
        t = torch.arange(nblk, dtype=torch.int32).view(1, -1)
        outs.append(ref_ops.paged_attention_decode(q[s:s + 1].cpu(), kc, vc, t, torch.tensor([n], dtype=torch.int32),
                                                   q.shape[-1] ** -0.5, p_round_to=p_round_to)[0])
    return torch.stack(outs)


def _decode(dev, q, kb, vb, table, lens, max_seq_len, out_dtype, mode="fast", partition_size=0):
    from vllm_omni_amd import ops

    S, nq, d = q.shape
    out = torch.full((S, nq * d), float("nan"), dtype=out_dtype, device=dev)
    ops.paged_attn_decode(out, q.view(S, nq * d), kb, vb, table.to(dev), torch.as_tensor(lens, dtype=torch.int32).to(dev),
                          max_seq_len, d ** -0.5, partition_size=partition_size, packed_pv=mode != "exact")
    torch.cuda.synchronize()
    return out.view(S, nq, d)


# ------------------------------------------------------------------ decode vs oracle, configs #2 / #3 / #5 geometry
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom,ctx", [(S_A, 2048), (S_A, 4096), (S_B, 4096), (S_B, 2048)],
                         ids=["S-A-ctx2048", "S-A-ctx4096", "S-B-ctx4096", "S-B-ctx2048"])
def test_decode_at_baseline_ctx_matches_oracle(dev, geom, ctx, dtype):
    """ctx-1 / ctx / ctx+1 under a ctx+16 launch bound (the bench's max_model_len), plus mid and tiny
    lengths: partitions of 1024 tokens = 64 blocks (the benchmarked regime; ctx+1 opens a 5th partition
    holding one token) and the auto choice for a small batch."""
    nq, nkv, d = geom["nq"], geom["nkv"], geom["d"]
    lens = [ctx - 1, ctx, ctx + 1, ctx // 2, (3 * ctx) // 4 - 7, 1]
    S = len(lens)
    bound = ctx + BS
    table, nb = _table(S, lens, seed=ctx + d, max_blocks=(bound + BS - 1) // BS)
    kv, kb, vb = _device_cache(dev, nb, nkv, d, dtype, seed=1)
    q = torch.randn(S, nq, d, generator=torch.Generator().manual_seed(2)).to(dtype).to(dev)
    rows = list(range(S))
    ref = _oracle_rows(kb, vb, table, lens, q, rows)
    tag = f"decode {nq}/{nkv}/{d} ctx{ctx} {str(dtype)[6:]}"
    for mode, part in (("fast", 1024), ("fast", 0), ("exact", 1024)):
        got = _decode(dev, q, kb, vb, table, lens, bound, torch.float32, mode, part).cpu()
        assert torch.isfinite(got).all()
        e = _errs(got, ref)
        _record(f"{tag} {mode} part={part} vs fp32 oracle", **e)
        assert e["atol_needed_at_rtol_1e-3"] <= _BOUNDS[(mode, dtype)], (mode, part, e)
        got16 = _decode(dev, q, kb, vb, table, lens, bound, dtype, mode, part).cpu()
        assert torch.equal(got16, got.to(dtype)), f"{mode}/{part}: 16-bit output is not round(fp32 output)"


@pytest.mark.parametrize("geom", [S_A, S_B], ids=["S-A", "S-B"])
def test_decode_32k_context_matches_oracle(dev, geom):
    """Far past the benchmarked context: 32768 (+17) tokens = 33 partitions of 1024 (2049 KV blocks per sequence, block
    ids well beyond one 64-lane window, the merge over 33 partials), next to a 1-token and a 5000-token sequence in the
    same launch; production mode against the fp32 oracle and 16-bit == round(fp32)."""
    nq, nkv, d = geom["nq"], geom["nkv"], geom["d"]
    lens = [32768 + 17, 1, 5000, 32768]
    S = len(lens)
    bound = 32768 + 32
    table, nb = _table(S, lens, seed=77 + d, max_blocks=(bound + BS - 1) // BS)
    kv, kb, vb = _device_cache(dev, nb, nkv, d, torch.bfloat16, seed=5)
    q = torch.randn(S, nq, d, generator=torch.Generator().manual_seed(6)).to(torch.bfloat16).to(dev)
    ref = _oracle_rows(kb, vb, table, lens, q, list(range(S)))
    for part in (1024, 0, 512):
        got = _decode(dev, q, kb, vb, table, lens, bound, torch.float32, "fast", part).cpu()
        assert torch.isfinite(got).all()
        e = _errs(got, ref)
        _record(f"decode {nq}/{nkv}/{d} ctx32k bf16 fast part={part} vs fp32 oracle", **e)
        assert e["atol_needed_at_rtol_1e-3"] <= _BOUNDS[("fast", torch.bfloat16)], (part, e)
        got16 = _decode(dev, q, kb, vb, table, lens, bound, torch.bfloat16, "fast", part).cpu()
        assert torch.equal(got16, got.to(torch.bfloat16))


@pytest.mark.parametrize("geom", [S_A, S_B], ids=["S-A", "S-B"])
def test_decode_one_hot_target_in_every_partition_bit_exact(dev, geom):
    """ctx 4096 (+1): a one-hot attention target in EACH of the four 1024-token partitions — first token of a
    window, last token of a window (block 63 of the window = lane 63 of the block-id register) — and in the 5th
    partition a 4097-token sequence opens.  Integer-valued V: the output must equal the targeted row exactly."""
    nq, nkv, d = geom["nq"], geom["nkv"], geom["d"]
    targets = [0, 1023, 1024, 2047, 2048, 3071, 3072, 4095, 4096, 1008 + 5, 4080 + 3]
    lens = [4096] * 8 + [4097, 4096, 4090]
    S = len(lens)
    bound = 4096 + BS
    mb = (bound + BS - 1) // BS
    table, nb = _table(S, lens, seed=77, max_blocks=mb)
    g = torch.Generator().manual_seed(5)
    kcache = torch.zeros(nb, BS, nkv, d, dtype=torch.bfloat16)
    vcache = torch.zeros(nb, BS, nkv, d, dtype=torch.bfloat16)
    expect = []
    for s, (n, tgt) in enumerate(zip(lens, targets)):
        kd = torch.zeros(mb * BS, nkv, d)
        kd[:n, :, 0] = -200.0
        kd[tgt, :, 0] = 200.0
        vd = torch.zeros(mb * BS, nkv, d)
        vd[:n] = torch.randint(-8, 8, (n, nkv, d), generator=g).float()
        kd[n:], vd[n:] = float("nan"), float("inf")       # rows past seq_len in the last block must not leak
        ids = table[s].to(torch.int64)
        nblk = (n + BS - 1) // BS
        kcache[ids[:nblk]] = kd.view(mb, BS, nkv, d)[:nblk].to(torch.bfloat16)
        vcache[ids[:nblk]] = vd.view(mb, BS, nkv, d)[:nblk].to(torch.bfloat16)
        table[s, nblk:] = -7                               # never-allocated entries must not be dereferenced
        expect.append(vd[tgt].repeat_interleave(nq // nkv, dim=0))
    phys = torch.stack([kcache.permute(0, 2, 1, 3), vcache.permute(0, 2, 1, 3)], dim=1).contiguous().to(dev)
    kb, vb = phys[:, 0].permute(0, 2, 1, 3), phys[:, 1].permute(0, 2, 1, 3)
    q = torch.zeros(S, nq, d, dtype=torch.bfloat16, device=dev)
    q[..., 0] = 64.0
    for mode, part in (("fast", 1024), ("fast", 0), ("fast", 256), ("exact", 1024)):
        got = _decode(dev, q, kb, vb, table, lens, bound, torch.bfloat16, mode, part).cpu().float()
        for s, tgt in enumerate(targets):
            assert torch.equal(got[s], expect[s]), f"{mode}/{part}: seq {s} target token {tgt}: wrong row gathered"


# ------------------------------------------------------------------ full-size launches (B = 256 / 64)
@pytest.mark.parametrize("B,ctx", [(256, 4096), (64, 2048)], ids=["config3-B256-ctx4096", "config2-B64-ctx2048"])
def test_decode_full_size_properties_and_sampled_oracle(dev, B, ctx):
    """The launch bench.py times (S-A, head-major cache, non-contiguous table, every sequence at ctx, bound
    ctx + 16): sampled sequences against the oracle; auto partition (1024) == forced 256 / 512 within fp32
    re-association + P-rounding noise; bf16 output == round(fp32 output); everything finite."""
    from vllm_omni_amd import ops

    nq, nkv, d = S_A["nq"], S_A["nkv"], S_A["d"]
    dtype = torch.bfloat16
    lens = [ctx] * B
    bound = ctx + BS
    table, nb = _table(B, lens, seed=3, max_blocks=(bound + BS - 1) // BS)
    kv, kb, vb = _device_cache(dev, nb, nkv, d, dtype, seed=1)
    q = torch.randn(B, nq, d, generator=torch.Generator().manual_seed(4)).to(dtype).to(dev)
    auto = ops.decode_auto_partition(B, nkv, bound, BS)
    got = _decode(dev, q, kb, vb, table, lens, bound, torch.float32, "fast", 0)
    assert torch.isfinite(got).all()
    got16 = _decode(dev, q, kb, vb, table, lens, bound, dtype, "fast", 0)
    assert torch.equal(got16, got.to(dtype))
    rows = [0, B // 3, B // 2 + 1, B - 1]
    ref = _oracle_rows(kb, vb, table, lens, q, rows)
    e = _errs(got[rows].cpu(), ref)
    _record(f"decode S-A B{B} ctx{ctx} bf16 fast auto-part={auto}: 4 sampled sequences vs fp32 oracle", **e)
    assert e["atol_needed_at_rtol_1e-3"] <= _BOUNDS[("fast", dtype)], e
    for part in (256, 512):
        other = _decode(dev, q, kb, vb, table, lens, bound, torch.float32, "fast", part)
        e = _errs(other.cpu(), got.cpu())
        _record(f"decode S-A B{B} ctx{ctx} bf16 fast part={part} vs auto-part={auto}", **e)
        assert e["atol_needed_at_rtol_1e-3"] <= 2 * _BOUNDS[("fast", dtype)], (part, e)
    exact = _decode(dev, q[rows].contiguous(), kb, vb, table[rows], [ctx] * len(rows), bound, torch.float32, "exact", 0)
    e = _errs(exact.cpu(), ref)
    _record(f"decode S-A ctx{ctx} bf16 exact, same 4 sequences vs fp32 oracle", **e)
    assert e["atol_needed_at_rtol_1e-3"] <= _BOUNDS[("exact", dtype)], e


@pytest.mark.parametrize("B,nq,nkv,ctx", [(256, 28, 4, 4096), (128, 7, 1, 4096), (64, 28, 4, 2048)],
                         ids=["config3-tp1", "config3-dp2xtp4-rank", "config2"])
def test_rope_fused_decode_full_size_equals_the_two_launches_and_the_oracle(dev, B, nq, nkv, ctx):
    """Round 4: the launch bench.py times since RoPE + KV write moved inside it (`mi_rope_cache_paged_attn_decode`) at
    the headline shape, at one rank's shape of the 8-GPU plan (7 / 1 heads, four partitions per sequence) and at config
    #2's: output, K cache and V cache bit-identical to `mi_rope_neox_and_cache` + `mi_paged_attn_decode` on a twin cache;
    sampled sequences against the oracle chain (rotate -> write -> attend) at the production bar."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.layers import build_cos_sin_cache

    d, dtype = 128, torch.bfloat16
    lens = [ctx] * B
    bound = ctx + BS
    table, nb = _table(B, lens, seed=13, max_blocks=(bound + BS - 1) // BS)
    kv_a, kb_a, vb_a = _device_cache(dev, nb, nkv, d, dtype, seed=11)
    kv_b = kv_a.clone()
    kb_b, vb_b = kv_b[:, 0].permute(0, 2, 1, 3), kv_b[:, 1].permute(0, 2, 1, 3)
    g = torch.Generator().manual_seed(14)
    qkv = torch.randn(B, (nq + 2 * nkv) * d, generator=g).to(dtype).to(dev)
    csc = build_cos_sin_cache(d, 8192, 1e6, device=dev)
    pos = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)
    slots = (table[:, (ctx - 1) // BS].to(torch.int64) * BS + (ctx - 1) % BS).to(dev)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=dev)
    tb = table.to(dev)
    # the two launches
    row = qkv.clone()
    q, k, v = row.split([nq * d, nkv * d, nkv * d], dim=-1)
    ops.rope_neox_and_cache(q, k, v, pos, csc, d, kb_a, vb_a, slots)
    out_two = torch.empty(B, nq * d, dtype=dtype, device=dev)
    ops.paged_attn_decode(out_two, q, kb_a, vb_a, tb, lens_d, bound, d ** -0.5)
    # the one launch
    q2, k2, v2 = qkv.split([nq * d, nkv * d, nkv * d], dim=-1)
    assert ops.rope_decode_fusable(q2, kb_b, pos, d, 0, B, bound)
    out_one = torch.empty_like(out_two)
    ops.rope_cache_paged_attn_decode(out_one, q2, k2, v2, pos, csc, kb_b, vb_b, slots, tb, lens_d, bound, d ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(kv_a, kv_b), "cache contents differ"
    assert torch.equal(out_one, out_two), "attention outputs differ"
    # sampled sequences against the oracle chain
    rows = [0, B // 3, B - 1]
    table_cpu = ref_ops.build_cos_sin_cache(d, 8192, 1e6)
    q0, k0, v0 = qkv[rows].cpu().split([nq * d, nkv * d, nkv * d], dim=-1)
    p0 = pos[rows].cpu()
    qr = ref_ops.rope_neox(p0, q0, d, table_cpu, round_to=dtype).view(len(rows), nq, d)
    ref = _oracle_rows(kb_b, vb_b, table, lens, torch.zeros(B, nq, d).index_copy_(0, torch.tensor(rows), qr.float()), rows)
    e = _errs(out_one.view(B, nq, d)[rows].float().cpu(), ref, out_ulp=2.0 ** -9)
    _record(f"fused RoPE + decode B{B} {nq}/{nkv} ctx{ctx} bf16: 3 sampled sequences vs the oracle chain (16-bit output)", **e)
    assert e["atol_needed_at_rtol_1e-3"] <= 2 * _BOUNDS[("fast", dtype)], e


@pytest.mark.parametrize("nq,nkv,d,part", [(28, 4, 128, 0), (7, 1, 128, 512), (16, 2, 64, 0)], ids=["S-A", "S-A-tp4-rank", "S-B"])
def test_rope_fused_decode_ragged_batch_equals_the_two_launches(dev, nq, nkv, d, part):
    """The fused RoPE + KV-write + decode launch on a RAGGED full-size batch (256 sequences of 1 .. 4096 tokens, some
    inert rows, every residue of the last token's position mod the KV block and mod the 32-token step, several
    partitions per sequence in the TP-4 geometry): the wave that owns each sequence's last partition differs per row —
    output and both caches bit-identical to the two launches, three times over on the same caches (the step appends)."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.layers import build_cos_sin_cache

    B, dtype = 256, torch.bfloat16
    rng = np.random.default_rng(5)
    lens = [int(x) for x in rng.integers(1, 4094, size=B)]
    for i in range(0, B, 37):
        lens[i] = 0                                     # inert rows of a bucketed graph replay
    for i, n in enumerate([1, 2, 15, 16, 17, 31, 32, 33, 511, 512, 513, 1024, 1025, 2047, 2048, 2049, 4093]):
        lens[3 + 5 * i] = n
    bound = 4112
    table, nb = _table(B, [bound] * B, seed=23, max_blocks=(bound + BS - 1) // BS)
    g = torch.Generator(device=dev).manual_seed(31)
    kv_a = torch.empty(nb, 2, nkv, BS, d, dtype=dtype, device=dev).normal_(0.0, 1.0, generator=g)
    kv_b = kv_a.clone()
    views = lambda kv: (kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3))   # noqa: E731
    (kb_a, vb_a), (kb_b, vb_b) = views(kv_a), views(kv_b)
    csc = build_cos_sin_cache(d, 8192, 1e6, device=dev)
    tb = table.to(dev)
    for step in range(3):
        cur = [n + step if n > 0 else 0 for n in lens]
        pos = torch.tensor([max(n - 1, 0) for n in cur], dtype=torch.int64, device=dev)
        slots = torch.tensor([int(table[i, (n - 1) // BS]) * BS + (n - 1) % BS if n > 0 else -1 for i, n in enumerate(cur)],
                             dtype=torch.int64, device=dev)
        lens_d = torch.tensor(cur, dtype=torch.int32, device=dev)
        qkv = torch.randn(B, (nq + 2 * nkv) * d, device=dev, generator=g).to(dtype)
        row = qkv.clone()
        q, k, v = row.split([nq * d, nkv * d, nkv * d], dim=-1)
        ops.rope_neox_and_cache(q, k, v, pos, csc, d, kb_a, vb_a, slots)
        out_two = torch.empty(B, nq * d, dtype=dtype, device=dev)
        ops.paged_attn_decode(out_two, q, kb_a, vb_a, tb, lens_d, bound, d ** -0.5, partition_size=part)
        q2, k2, v2 = qkv.split([nq * d, nkv * d, nkv * d], dim=-1)
        out_one = torch.empty_like(out_two)
        ops.rope_cache_paged_attn_decode(out_one, q2, k2, v2, pos, csc, kb_b, vb_b, slots, tb, lens_d, bound, d ** -0.5,
                                         partition_size=part)
        torch.cuda.synchronize()
        assert torch.equal(kv_a, kv_b), f"step {step}: cache contents differ"
        assert torch.equal(out_one, out_two), f"step {step}: attention outputs differ"
    assert torch.isfinite(out_one.float()).all()


# ------------------------------------------------------------------ config #4: 8192-token chunked prefill
def _prefill_chunks(dev, q, kb, vb, table, total, chunk):
    from vllm_omni_amd import ops

    T, nq, d = q.shape
    out = torch.full((T, nq * d), float("nan"), dtype=q.dtype, device=dev)
    for c0 in range(0, total, chunk):
        c1 = min(total, c0 + chunk)
        ops.paged_attn_prefill(out[c0:c1], q.view(T, nq * d)[c0:c1], kb, vb, table.to(dev),
                               torch.tensor([c1], dtype=torch.int32, device=dev),
                               torch.tensor([0, c1 - c0], dtype=torch.int32, device=dev), c1 - c0, d ** -0.5)
    torch.cuda.synchronize()
    return out.view(T, nq, d)


@pytest.mark.parametrize("geom", [S_A, S_B], ids=["S-A-shaped-thinker", "S-B"])
def test_prefill_8192_token_prompt_chunked(dev, geom):
    """One 8192-token prompt prefilled in chunks of 8192 / 4096 / 2048 (SURVEY §8d cfg#4): each chunk attends
    causally over the paged cache [0, chunk end).  Chunkings agree with each other; a head subset (first / middle /
    last query head, hence three different KV heads at S-A) is compared with the oracle over the WHOLE 8192 x 8192
    causal problem."""
    nq, nkv, d = geom["nq"], geom["nkv"], geom["d"]
    dtype = torch.bfloat16
    total = 8192
    table, nb = _table(1, [total], seed=9)
    kv, kb, vb = _device_cache(dev, nb, nkv, d, dtype, seed=6)
    q = torch.randn(total, nq, d, generator=torch.Generator().manual_seed(7)).to(dtype).to(dev)
    outs = {c: _prefill_chunks(dev, q, kb, vb, table, total, c).float().cpu() for c in (8192, 4096, 2048)}
    for c, o in outs.items():
        assert torch.isfinite(o).all(), c
    for c in (4096, 2048):
        e = _errs(outs[c], outs[8192])
        _record(f"prefill {nq}/{nkv}/{d} 8192 tokens: chunks of {c} vs one chunk", **e)
        assert e["max_abs"] <= 2e-3, (c, e)                      # same tiles, same order: at most re-association
    ids = table[0].to(torch.int64)
    kc, vc = kb[ids.to(dev)].cpu(), vb[ids.to(dev)].cpu()
    t = torch.arange(ids.numel(), dtype=torch.int32).view(1, -1)
    grp = nq // nkv
    for h in sorted({0, nq // 2, nq - 1}):
        ref = ref_ops.paged_attention_prefill(q[:, h:h + 1].cpu(), kc[:, :, h // grp:h // grp + 1], vc[:, :, h // grp:h // grp + 1],
                                              t, torch.tensor([total], dtype=torch.int32),
                                              torch.tensor([0, total], dtype=torch.int32), d ** -0.5)
        for c, o in outs.items():
            e = _errs(o[:, h:h + 1], ref, out_ulp=2.0 ** -8)
            _record(f"prefill {nq}/{nkv}/{d} 8192 tokens chunk {c} head {h} vs fp32 oracle (bf16 output)", **e)
            # north_star's bar (rtol 1e-3, atol 1e-4) + the one rounding of the bf16 output
            assert e["atol_needed_at_rtol_1e-3"] <= 1e-4, (c, h, e)


@pytest.mark.parametrize("geom", [S_A, S_B], ids=["S-A", "S-B"])
def test_prefill_late_chunks_on_a_16k_context(dev, geom):
    """Chunks far into a long prompt: two sequences in one launch, 500 and 77 new tokens on 16384 and 16000 cached ones
    (tiles that run 512+ steps, ragged last tiles, contexts that are not multiples of the 32-key step or of the
    block), all query heads, every row against the fp32 oracle."""
    from vllm_omni_amd import ops

    nq, nkv, d = geom["nq"], geom["nkv"], geom["d"]
    dtype = torch.bfloat16
    ctxs, chunks = [16384, 16000 + 5], [500, 77]
    lens = [c + n for c, n in zip(ctxs, chunks)]
    table, nb = _table(2, lens, seed=21 + d)
    kv, kb, vb = _device_cache(dev, nb, nkv, d, dtype, seed=8)
    T = sum(chunks)
    q = torch.randn(T, nq, d, generator=torch.Generator().manual_seed(9)).to(dtype).to(dev)
    out = torch.full((T, nq * d), float("nan"), dtype=dtype, device=dev)
    qsl = torch.tensor([0, chunks[0], T], dtype=torch.int32)
    ops.paged_attn_prefill(out, q.view(T, nq * d), kb, vb, table.to(dev), torch.tensor(lens, dtype=torch.int32, device=dev),
                           qsl.to(dev), max(chunks), d ** -0.5)
    torch.cuda.synchronize()
    got = out.view(T, nq, d).float().cpu()
    assert torch.isfinite(got).all()
    for s_ in range(2):
        nblk = (lens[s_] + BS - 1) // BS
        ids = table[s_, :nblk].to(torch.int64)
        kc, vc = kb[ids.to(dev)].cpu(), vb[ids.to(dev)].cpu()
        t = torch.arange(nblk, dtype=torch.int32).view(1, -1)
        a, b = int(qsl[s_]), int(qsl[s_ + 1])
        ref = ref_ops.paged_attention_prefill(q[a:b].cpu(), kc, vc, t, torch.tensor([lens[s_]], dtype=torch.int32),
                                              torch.tensor([0, b - a], dtype=torch.int32), d ** -0.5)
        e = _errs(got[a:b], ref, out_ulp=2.0 ** -8)
        _record(f"prefill {nq}/{nkv}/{d} chunk {b - a} on ctx {ctxs[s_]} vs fp32 oracle (bf16 output)", **e)
        assert e["atol_needed_at_rtol_1e-3"] <= 1e-4, (s_, e)


# ------------------------------------------------------------------ config #5: int8 decode layer, S-B, batch 512
def test_int8_decoder_layer_at_config5_size(dev):
    """One S-B decoder layer (16/2 heads, d 64, H 1024, q/k-norm) with W8A8 linears at batch 512, ctx 4096
    (BASELINE config #5's per-rank shapes at TP=1; the MoE experts stay 16-bit in the reference's scheme, so the
    dense stand-in MLP of qwen3_omni_talker_attn carries the int8 MLP linears) through the runner, against the
    oracle's int8 stack.  Integer parts (quantised weights, scales' placement) are exact by construction of the
    shared scheme; the hidden state is compared at the 16-bit tolerance of the step tests."""
    from oracle import ref_talker
    from vllm_omni_amd.kv_cache import KVCacheManager, normalize_layer_kv
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import NewRequestData, Request, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("qwen3_omni_talker_attn")
    cfg.num_hidden_layers = 1
    B, ctx = 512, 4096
    rows = [0, 100, 311, 511]                       # sequences the oracle replays
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=B, max_model_len=ctx + BS,
                              max_num_batched_tokens=B, block_size=BS, enforce_eager=True, seed=0, quantization="int8")
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=3)
    assert runner.model.load_weights({k: v.to(dev) for k, v in w.items()}) == set(w.keys())
    bps = (ctx + BS) // BS
    nb = B * bps + 1
    runner.initialize_kv_cache(nb)
    g = torch.Generator(device=dev).manual_seed(5)
    kb, vb = normalize_layer_kv(runner.kv_caches[0])
    base = runner.kv_caches[0] if isinstance(runner.kv_caches[0], torch.Tensor) else kb._base
    base.normal_(0.0, 1.0, generator=g)
    mgr = KVCacheManager(nb, BS)
    scramble = mgr.block_pool.get_new_blocks(nb - 1)
    mgr.block_pool.free_blocks(np.random.default_rng(0).permutation(scramble).tolist())
    reqs = []
    gen = torch.Generator().manual_seed(8)
    for i in range(B):
        r = Request(f"r{i}", [int(x) for x in torch.randint(0, cfg.vocab_size, (4,), generator=gen)] + [i % cfg.vocab_size],
                    SamplingParams(temperature=0.0))
        r.prompt_token_ids = r.prompt_token_ids * 1   # only the last id matters for the decode embedding
        assert mgr.allocate_slots(r, ctx) is not None
        r.num_computed_tokens = ctx - 1
        reqs.append(r)
    so = SchedulerOutput(scheduled_new_reqs=[
        NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                       r.num_computed_tokens) for r in reqs],
        num_scheduled_tokens={r.request_id: 1 for r in reqs}, total_num_scheduled_tokens=B)
    assert runner.execute_model(so) is None
    out = runner.sample_tokens()
    got = torch.stack([out.pooler_output[out.req_id_to_index[reqs[i].request_id]]["hidden"][0] for i in rows]).float()
    # ---- oracle on the sampled sequences (their blocks re-numbered into a compact host cache) ----
    n = len(rows)
    table = torch.zeros(n, bps, dtype=torch.int32)
    kc = torch.zeros(n * bps + 1, BS, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    pos = torch.full((n,), ctx - 1, dtype=torch.int64)
    for j, i in enumerate(rows):
        ids = torch.tensor(mgr.get_block_ids(reqs[i].request_id)[0], dtype=torch.int64)
        new = torch.arange(1 + j * bps, 1 + j * bps + ids.numel())
        table[j, :ids.numel()] = new.to(torch.int32)
        kc[new], vc[new] = kb[ids.to(dev)].cpu(), vb[ids.to(dev)].cpu()
    # the step wrote position ctx-1 on the device; the oracle writes its own value there
    last = torch.tensor([reqs[i].prompt_token_ids[-1] for i in rows], dtype=torch.int64)
    emb = ref_talker.decode_embeds(w, last, None, torch.bfloat16)
    slots = ref_ops.slot_mapping(table, torch.arange(n), pos, BS)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, ctx + BS, cfg.rope_theta)
    hidden = ref_talker.talker_step(cfg, w, emb, pos, [(kc, vc)], table, (pos + 1).to(torch.int32),
                                    torch.arange(n + 1, dtype=torch.int32), slots, csc, torch.bfloat16, quant="int8")
    e = _errs(got, hidden)
    frac = float(((got - hidden).abs() <= 5e-2 + 5e-2 * hidden.abs()).float().mean())
    _record("int8 S-B decoder layer, batch 512 ctx 4096: hidden of 4 sampled sequences vs int8 oracle",
            frac_within_5e_2=frac, **e)
    # a one-ulp difference in a 16-bit activation moves a quantised value by one step (1/127 of its row's max); a few
    # of the 4 x 1024 outputs land just outside the step tests' 5e-2 after the final norm (measured: max 0.078 on 3.9)
    assert frac >= 0.995 and e["max_abs"] <= 0.15, (frac, e)
    # the K row written this step is RoPE(k_norm(int8 qkv)) — same placement, same values to 16-bit rounding
    b_idx, off = slots // BS, slots % BS
    dev_rows = torch.stack([kb[int(mgr.get_block_ids(reqs[i].request_id)[0][(ctx - 1) // BS]), (ctx - 1) % BS].cpu() for i in rows])
    torch.testing.assert_close(dev_rows.float(), kc[b_idx, off].float(), rtol=5e-2, atol=5e-2)


def test_int8_linear_integer_parts_exact_at_config5_size(dev):
    """The integer half of W8A8 at config #5's row count (M = 512) and S-B layer shapes (int8_config.py:323-342):
    quantised activations / weights equal the oracle's except isolated 1-LSB flips at exact .5 ties (x / scale is
    one fp32 division on both sides), scales to fp32 rounding; the int32 accumulators of the contraction are EXACT;
    the dequantised output is the oracle's to one 16-bit ulp, > 99 % of elements bit-identical."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(12)
    for n, k in ((1280, 1024), (1024, 1024), (768, 1024), (1024, 384)):      # qkv, o, gate_up, down of the stand-in layer
        x = (torch.randn(512, k, generator=g) * 2).to(torch.bfloat16)
        w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16)
        b = torch.randn(n, generator=g).to(torch.bfloat16)
        for t in (x, w):
            q, sc = ops.dynamic_int8_quant(t.to(dev))
            rq, rs = ref_ops.dynamic_int8_quant(t)
            torch.testing.assert_close(sc.cpu(), rs, rtol=1e-6, atol=0)
            diff = (q.cpu().int() - rq.int()).abs()
            assert diff.max() <= 1 and (diff != 0).float().mean() < 1e-3
        xq, xs = ops.dynamic_int8_quant(x.to(dev))
        wq, ws = ops.dynamic_int8_quant(w.to(dev))
        exact = xq.cpu().to(torch.int32) @ wq.cpu().to(torch.int32).t()
        # the device contraction as ops.int8_linear runs it at this size, and the in-tree MFMA kernel: exact sums
        assert torch.equal(torch._int_mm(xq, wq.t()).cpu(), exact)
        got = ops.int8_linear(x.to(dev), wq, ws, b.to(dev)).cpu().float()
        ref = (exact.float() * xs.cpu()[:, None] * ws.cpu()[None, :] + b.float()).to(torch.bfloat16).float()
        torch.testing.assert_close(got, ref, rtol=2.0 ** -7, atol=1e-5)
        assert (got == ref).float().mean() > 0.99, (n, k)
        mfma = ops.int8_gemm_dequant(xq, xs, wq, ws, b.to(dev), torch.bfloat16).cpu().float()
        assert torch.equal(mfma, got), "library and in-tree int8 contractions disagree"


# ------------------------------------------------------------------ 64-step greedy agreement
def test_64_step_greedy_agreement_mini_talker(dev):
    """64 consecutive decode steps of the 2-layer mini talker through the runner (hipGraph replay) against the
    oracle stack: both sides are fed the DEVICE's token each step (teacher forcing keeps the comparison per-step
    meaningful while the KV caches evolve independently for 64 steps — errors in the written K/V rows compound).
    Greedy ids must agree on every row whose oracle top-2 margin exceeds the logit tolerance; the hidden-state
    error must not grow with the step count."""
    from oracle import ref_talker
    from vllm_omni_amd.kv_cache import KVCacheManager, normalize_layer_kv
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, Request, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    cfg.mrope_section = None
    steps, bs = 64, BS
    ctxs = [5, 16, 40, 1, 100]
    B = len(ctxs)
    nb = 96
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=8, max_model_len=256, max_num_batched_tokens=64,
                              block_size=bs, enforce_eager=False, max_reply_rows=8 * steps, seed=0)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=3)
    assert runner.model.load_weights({k: v.to(dev) for k, v in w.items()}) == set(w.keys())
    runner.initialize_kv_cache(nb)
    runner.model.set_suppress_start_id(cfg.tts_codec_mask_token_id)
    g = torch.Generator().manual_seed(5)
    cpu_caches = []
    for kv in runner.kv_caches:
        init = torch.randn(nb, 2, bs, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16)
        kb, vb = normalize_layer_kv(kv)
        kb.copy_(init[:, 0].to(dev))
        vb.copy_(init[:, 1].to(dev))
        cpu_caches.append((init[:, 0].clone(), init[:, 1].clone()))
    mgr = KVCacheManager(nb, bs)
    reqs = []
    for i, c in enumerate(ctxs):
        r = Request(f"r{i}", list(range(c + 1)), SamplingParams(temperature=0.0))
        r.additional_information = {"thinker_reply_part": (torch.randn(steps, cfg.embedding_size, generator=g) * 0.5).to(torch.bfloat16)}
        assert mgr.allocate_slots(r, c + 1) is not None
        r.num_computed_tokens = c
        reqs.append(r)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
    last_ids = torch.tensor([r.prompt_token_ids[-1] for r in reqs], dtype=torch.int64)
    table = torch.zeros(B, 16, dtype=torch.int32)
    agree = total = 0
    errs = []
    for step in range(steps):
        sched = {r.request_id: 1 for r in reqs}
        if step == 0:
            so = SchedulerOutput(scheduled_new_reqs=[
                NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                               r.num_computed_tokens, additional_information=r.additional_information) for r in reqs],
                num_scheduled_tokens=sched, total_num_scheduled_tokens=B)
        else:
            new_blocks = []
            for r in reqs:
                nb_ = mgr.allocate_slots(r, 1)
                assert nb_ is not None
                new_blocks.append(nb_.get_block_ids() if nb_.blocks[0] else None)
            so = SchedulerOutput(scheduled_cached_reqs=CachedRequestData(
                req_ids=[r.request_id for r in reqs], new_block_ids=new_blocks,
                num_computed_tokens=[r.num_computed_tokens for r in reqs], num_output_tokens=[step] * B),
                num_scheduled_tokens=sched, total_num_scheduled_tokens=B)
        for i, r in enumerate(reqs):
            ids = mgr.get_block_ids(r.request_id)[0]
            table[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        assert runner.execute_model(so) is None
        out = runner.sample_tokens()
        order = [out.req_id_to_index[r.request_id] for r in reqs]
        pos = torch.tensor([r.num_computed_tokens for r in reqs], dtype=torch.int64)
        step_vec = torch.stack([r.additional_information["thinker_reply_part"][step] for r in reqs])
        emb = ref_talker.decode_embeds(w, last_ids, step_vec, torch.bfloat16)
        slots = ref_ops.slot_mapping(table, torch.arange(B), pos, bs)
        hidden = ref_talker.talker_step(cfg, w, emb, pos, cpu_caches, table, (pos + 1).to(torch.int32),
                                        torch.arange(B + 1, dtype=torch.int32), slots, csc, torch.bfloat16)
        logits, ids = ref_talker.logits_and_greedy(cfg, w, hidden, cfg.tts_codec_mask_token_id, torch.bfloat16)
        got_hidden = torch.stack([out.pooler_output[j]["hidden"][0] for j in order]).float()
        errs.append(float((got_hidden - hidden).abs().max()))
        torch.testing.assert_close(got_hidden, hidden, rtol=2e-2, atol=2e-2, msg=lambda m: f"step {step}: {m}")
        got_ids = torch.tensor([out.sampled_token_ids[j][0] for j in order])
        top2 = logits.topk(2, dim=-1).values
        confident = (top2[:, 0] - top2[:, 1]) > 0.05
        total += int(confident.sum())
        agree += int((got_ids[confident] == ids[confident]).sum())
        last_ids = got_ids.clone()
        for r, t in zip(reqs, got_ids.tolist()):
            r.num_computed_tokens += 1
            r.output_token_ids.append(t)
    _record("64-step greedy agreement, mini talker (hipGraph)", confident_rows=total, agreeing=agree,
            max_hidden_err_first8=max(errs[:8]), max_hidden_err_last8=max(errs[-8:]))
    assert total >= steps and agree == total, f"greedy ids disagree on {total - agree} of {total} confident rows"
    assert max(errs[-8:]) <= 2.5 * max(max(errs[:8]), 2e-3), "hidden-state error grows over the 64 steps"
