"""GPU parity: fused sparse-MoE (csrc/fused_moe.hip) vs oracle/ref_moe.py (pinned by HF outputs).

Integer results (expert ids, pair grouping) exactly; routing weights to fp32 rounding; the layer output against the
oracle evaluated with the DEVICE's routing (an expert swap on a near-tie of two probabilities is a legitimate
difference, counted separately) on the 16-bit grid: rtol 2e-2 / atol 2e-2 for bf16 sums of 8 + 1 expert outputs."""
import numpy as np
import pytest
import torch

from oracle import ref_moe

pytestmark = pytest.mark.gpu


def _weights(cfg, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    h, i, e, s = cfg.hidden_size, cfg.moe_intermediate_size, cfg.num_experts, cfg.shared_expert_intermediate_size
    r = lambda *shape, std=0.08: (torch.randn(*shape, generator=g) * std).to(dtype)  # noqa: E731
    w = {"gate.weight": r(e, h, std=0.3), "experts.gate_up_proj": r(e, 2 * i, h), "experts.down_proj": r(e, h, i)}
    if s:
        w.update({"shared_expert.gate_proj.weight": r(s, h), "shared_expert.up_proj.weight": r(s, h),
                  "shared_expert.down_proj.weight": r(h, s), "shared_expert_gate.weight": r(1, h, std=0.3)})
    return w


@pytest.mark.parametrize("E,k,T,norm", [(128, 8, 40, False), (128, 8, 1, False), (64, 4, 33, True), (256, 2, 7, False),
                                        (128, 8, 1500, True),    # two tokens per wave, ragged last workgroup
                                        (128, 8, 8197, False),   # four tokens per wave (prefill chunk)
                                        (64, 6, 4099, True), (256, 8, 1027, False)])
def test_route_matches_oracle(dev, E, k, T, norm):
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(E + T)
    logits = torch.randn(T, E, generator=g) * 2
    if T > 1000:   # thousands of rows: keep every row's values >= 0.02 apart, so that no two probabilities can round to
        # the same fp32 value on one side only (the device's exp is not torch's)
        logits = (torch.argsort(torch.rand(T, E, generator=g), dim=-1).float() * 0.02 + torch.rand(T, 1, generator=g))
    ids, w, counts = ops.moe_route(logits.to(dev), k, norm)
    p = torch.softmax(logits, -1)
    rw, re = torch.topk(p, k, -1)
    if norm:
        rw = rw / rw.sum(-1, keepdim=True)
    assert torch.equal(ids.cpu().long(), re)
    torch.testing.assert_close(w.cpu(), rw, rtol=2e-5, atol=1e-7)
    assert torch.equal(counts.cpu().long(), torch.bincount(re.reshape(-1), minlength=E))


def test_route_ties_take_the_lowest_expert_and_counts_add_up(dev):
    """Logits drawn from four distinct values: every row is full of exact ties.  The kernel's order is descending
    probability, lowest expert id first among equals (a stable sort); the per-expert counts are the histogram of the ids
    whatever the number of tokens a workgroup handles."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(5)
    for T, E, k in [(9, 128, 8), (2050, 128, 8), (8200, 64, 4)]:
        logits = torch.randint(0, 4, (T, E), generator=g).float()
        ids, w, counts = ops.moe_route(logits.to(dev), k, False)
        p = torch.softmax(logits, -1)
        order = torch.sort(p, dim=-1, descending=True, stable=True).indices[:, :k]
        assert torch.equal(ids.cpu().long(), order)
        torch.testing.assert_close(w.cpu(), torch.gather(p, 1, order), rtol=2e-5, atol=1e-7)
        assert torch.equal(counts.cpu().long(), torch.bincount(order.reshape(-1), minlength=E))
        assert int(counts.sum()) == T * k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,E,k,H,I,S", [
    (37, 128, 8, 1024, 384, 768),      # the talker shape, ragged token count
    (1, 128, 8, 1024, 384, 768),       # one token: 8 tiles of one live row
    (300, 64, 4, 256, 128, None),      # several tiles per expert, no shared expert
    (1500, 64, 8, 256, 128, None),     # a prefill chunk: ~190 rows per expert (two 128-row tiles, one 256-row tile)
    (5, 64, 2, 128, 64, 128),
])
def test_fused_moe_matches_oracle(dev, dtype, T, E, k, H, I, S):
    from vllm_omni_amd.moe import MoEConfig, Qwen3OmniMoeSparseMoeBlock

    cfg = MoEConfig(hidden_size=H, moe_intermediate_size=I, num_experts=E, num_experts_per_tok=k, norm_topk_prob=False,
                    shared_expert_intermediate_size=S)
    w = _weights(cfg, seed=T + E, dtype=dtype)
    blk = Qwen3OmniMoeSparseMoeBlock(cfg, dtype, dev)
    assert blk.load_weights({n: v.to(dev) for n, v in w.items()}) == set(w.keys())
    x = (torch.randn(T, H, generator=torch.Generator().manual_seed(9)) * 0.8).to(dtype)
    got = blk(x.to(dev)).cpu().float()
    # both tile heights give the same rows (an output row does not depend on which tile computed it)
    from vllm_omni_amd import ops as _ops
    lg = torch.nn.functional.linear(x.to(dev), blk.gate_weight)
    by_tile = [_ops.fused_moe(x.to(dev), lg, blk.gate_up_proj, blk.down_proj, k, False, rows_per_tile=r)
               for r in (32, 64, 128, 256)]      # 128 / 256: the prefill-chunk tiles of the ring kernel
    assert all(torch.equal(by_tile[0], o) for o in by_tile[1:])
    # the device's routing (from its own 16-bit router logits), checked against the oracle's on the same logits
    from vllm_omni_amd import ops
    logits = torch.nn.functional.linear(x.to(dev), blk.gate_weight)
    ids, rw, _ = ops.moe_route(logits, k, False, round_to=dtype)
    p = torch.softmax(logits.float().cpu(), -1)
    ow, oe = torch.topk(p, k, -1)
    # 16-bit logits tie often; which of two equal experts is taken is unspecified (torch.topk) / lowest id (here):
    # the selected PROBABILITIES must be the top-k values, in descending order, and the ids distinct
    torch.testing.assert_close(p.gather(1, ids.cpu().long()), ow, rtol=1e-6, atol=0)
    assert all(len(set(r)) == k for r in ids.cpu().tolist())
    wf = {n: v.float() for n, v in w.items()}
    ref = ref_moe.moe_block(x.float(), wf, k, False, rt=dtype, routing=(rw.cpu(), ids.cpu().long()))
    # every term is rounded to 16 bits on both sides (expert outputs, shared expert, sum): an element that is a
    # small difference of large terms carries the rounding of the large ones — atol = one ulp of the largest output
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    atol = float(ref.abs().max()) * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10)
    torch.testing.assert_close(got, ref, rtol=tol, atol=atol)
    assert (got - ref).abs().mean() < 0.25 * atol
    # and the free-running oracle (its own fp32 router) agrees wherever it picked the same experts
    ref_free = ref_moe.moe_block(x.float(), wf, k, False, rt=dtype)
    same = (torch.sort(ref_moe.route(x.float(), wf["gate.weight"], k, False)[1], -1).values
            == torch.sort(ids.cpu().long(), -1).values).all(-1)
    assert same.float().mean() > 0.8
    # looser: the oracle's router ran on fp32 logits, the device's on 16-bit ones (routing weights differ by ~1 %)
    torch.testing.assert_close(got[same], ref_free[same], rtol=3 * tol, atol=3 * atol)


@pytest.mark.parametrize("T", [77, 8192, 5000, 513])   # one workgroup; 16, 10 and 2 workgroups (4096 pairs each)
@pytest.mark.parametrize("rpt,given_counts", [(32, True), (64, True), (64, False), (256, False)])
def test_align_groups_pairs_by_expert(dev, rpt, T, given_counts):
    """Bit-exact integer structure: every pair appears exactly once, in a row range owned by its expert, ranges are
    padded to 32 rows with -1, tile_expert names the owner of every tile — for a decode batch (one workgroup) and for
    prefill-sized token counts (several workgroups, each counting the pairs in front of its chunk itself); with the
    caller's per-expert counts, and (``expert_counts`` NULL, what ``fused_moe`` passes) counting the ids inside the launch."""
    from vllm_omni_amd import _C, ops

    lib = _C.lib
    k, E = 8, 128
    g = torch.Generator().manual_seed(1)
    ids = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32).to(dev)
    counts = torch.bincount(ids.reshape(-1).long(), minlength=E).to(torch.int32)
    max_rows = int(lib.mi_moe_max_rows(T, k, E, rpt))
    i32 = dict(dtype=torch.int32, device=dev)
    sp, rop, te, nt = torch.empty(max_rows, **i32), torch.empty(T * k, **i32), torch.empty(max_rows // rpt, **i32), torch.empty(1, **i32)
    ops.check("mi_moe_align", lib.mi_moe_align(sp.data_ptr(), rop.data_ptr(), te.data_ptr(), nt.data_ptr(), ids.data_ptr(),
                                               counts.data_ptr() if given_counts else None, T * k, E, max_rows, rpt,
                                               torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    n_tiles = int(nt.item())
    assert n_tiles == int(((counts + rpt - 1) // rpt).sum()) and n_tiles * rpt <= max_rows
    sp_c, rop_c, te_c, flat = sp.cpu()[: n_tiles * rpt], rop.cpu(), te.cpu()[:n_tiles], ids.cpu().reshape(-1)
    live_rows = (sp_c >= 0).nonzero().flatten()
    live = sp_c[live_rows].long()
    assert torch.equal(live.sort().values, torch.arange(T * k))                      # every pair exactly once
    assert torch.equal(te_c[live_rows // rpt].long(), flat[live].long())              # in a tile its expert owns
    assert torch.equal(rop_c[live].long(), live_rows)                                 # and row_of_pair is the inverse
    assert int((sp_c < 0).sum()) == n_tiles * rpt - T * k                             # everything else is -1 padding
    assert (te_c[1:] >= te_c[:-1]).all()
