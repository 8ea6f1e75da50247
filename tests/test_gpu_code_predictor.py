"""GPU parity: Qwen3-Omni talker code predictor — the incremental-decode implementation against
the oracle's re-prefill restatement of the reference.  Codes are integers: the device's draw at every step must be
the sampler oracle's draw on the device's own logits (exact, up to flagged last-ulp coin flips), and those logits
must match the oracle's teacher-forced logits to 16-bit accumulation noise (five layers, bf16)."""
import numpy as np
import pytest
import torch

from oracle import ref_code_predictor as ref
from oracle import ref_sampler

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[False, True], ids=["launch_chain", "one_launch_stack"])
def stack_mode(request, monkeypatch):
    """Both forms of the single-token passes: the per-layer launches (default) and the whole stack as one persistent
    launch per pass (csrc/cp_stack.hip; VLLM_OMNI_AMD_CP_ONE_LAUNCH=1)."""
    monkeypatch.setenv("VLLM_OMNI_AMD_CP_ONE_LAUNCH", "1" if request.param else "0")
    return request.param


def _mk(dev, cfg, b, seed=0):
    from vllm_omni_amd.code_predictor import MI355XCodePredictor

    w = ref.random_weights(cfg, seed=seed)
    cp = MI355XCodePredictor(cfg, torch.bfloat16, dev, max_batch=8, seed=0)
    loaded = cp.load_weights({k: v.to(dev) for k, v in w.items()})
    assert loaded == set(w.keys())
    g = torch.Generator().manual_seed(seed + 1)
    hidden = torch.randn(b, cfg.hidden_size, generator=g).to(torch.bfloat16)
    code0 = torch.randint(0, cfg.vocab_size, (b, 1), generator=g)
    emb0 = torch.randn(b, 1, cfg.hidden_size, generator=g).to(torch.bfloat16)
    return w, cp, hidden, code0, emb0


@pytest.mark.parametrize("groups", [4, 32])
def test_code_predictor_matches_oracle(dev, groups, stack_mode):
    from vllm_omni_amd.code_predictor import CodePredictorConfig

    cfg = CodePredictorConfig(hidden_size=256, num_hidden_layers=5, num_attention_heads=4, num_key_value_heads=2,
                              head_dim=128, intermediate_size=512, vocab_size=512, num_code_groups=groups)
    b = 5
    w, cp, hidden, code0, emb0 = _mk(dev, cfg, b)
    seed = torch.arange(11, 11 + b, dtype=torch.int64)
    offset = torch.tensor([0, 3, 7, 100, 2**20], dtype=torch.int64)
    codes, buf, logits = cp(code0.to(dev), emb0.to(dev), hidden.to(dev), seed=seed.to(dev), offset=offset.to(dev),
                            return_logits=True)
    torch.cuda.synchronize()
    assert (cp._cp_stack is not None) == stack_mode and (not stack_mode or cp._cp_stack.status() == 0)
    codes_c, buf_c = codes.cpu(), buf.cpu().float()
    assert codes_c.shape == (b, groups, 1) and buf_c.shape == (b, groups + 1, cfg.hidden_size)
    assert torch.equal(codes_c[:, 0], code0)
    # 1. every draw is the sampler oracle's draw on the device's logits ("stored" mode: top_k 50, top_p 0.8, T = 1)
    flips = 0
    for step in range(1, groups):
        lg = logits[step - 1].cpu().numpy()
        want, margins = ref_sampler.sample(lg, [1.0] * b, [50] * b, [0.8] * b, None, None, None, seed.tolist(),
                                           [int(o) * groups + step for o in offset])
        for r in range(b):
            if int(codes_c[r, step, 0]) != int(want[r]):
                m = margins[r]
                assert m["topp_gap"] < 1e-5 or m["race_ratio"] < 1 + 1e-4, (step, r, m)
                flips += 1
    assert flips <= 2
    # 2. teacher-forced oracle (the reference's re-prefill loop): logits and the embedding buffer agree
    _, buf_ref, logits_ref = ref.forward(cfg, w, code0, emb0, hidden, rt=torch.bfloat16, forced_codes=codes_c[:, :, 0])
    torch.testing.assert_close(buf_c, buf_ref, rtol=0, atol=0)           # embeddings are looked up, not computed
    for step in range(1, groups):
        # logits reach |x| ~ 16 here: one bf16 ulp is 0.06-0.12, and five bf16 layers sit in front of them
        torch.testing.assert_close(logits[step - 1].cpu(), logits_ref[step - 1], rtol=3e-2, atol=0.2,
                                   msg=lambda m: f"step {step}: {m}")
        assert (logits[step - 1].cpu() - logits_ref[step - 1]).abs().mean() < 0.03
    # 3. same seeds -> same codes; greedy mode == argmax of the device logits
    codes2, _ = cp(code0.to(dev), emb0.to(dev), hidden.to(dev), seed=seed.to(dev), offset=offset.to(dev))
    assert torch.equal(codes2.cpu(), codes_c)
    gcodes, _, glogits = cp(code0.to(dev), emb0.to(dev), hidden.to(dev), do_sample=False, return_logits=True)
    for step in range(1, groups):
        assert torch.equal(gcodes[:, step, 0].cpu(), glogits[step - 1].argmax(-1).cpu())


def test_code_predictor_incremental_equals_reprefill_on_device(dev, stack_mode):
    """Size-independent property at the real shape (H 1024, 16/8 heads, G 32, batch 8): the hidden state the
    incremental path produces at the last position equals a from-scratch varlen prefill of the whole buffer."""
    from vllm_omni_amd.code_predictor import CodePredictorConfig, MI355XCodePredictor
    from vllm_omni_amd.layers import AttentionMetadata

    cfg = CodePredictorConfig()
    cp = MI355XCodePredictor(cfg, torch.bfloat16, dev, max_batch=8, seed=3)
    b, g, h = 8, cfg.num_code_groups, cfg.hidden_size
    gen = torch.Generator(device=dev).manual_seed(1)
    hidden = torch.randn(b, h, device=dev, generator=gen).to(torch.bfloat16)
    emb0 = torch.randn(b, h, device=dev, generator=gen).to(torch.bfloat16)
    code0 = torch.randint(0, cfg.vocab_size, (b,), device=dev, generator=gen)
    codes, buf, logits = cp(code0, emb0, hidden, return_logits=True)
    # re-prefill positions 0 .. g-1 of every sequence in one varlen call into the same private cache
    t = g
    x = buf[:, :t].reshape(b * t, h).clone()
    pos = torch.arange(t, device=dev).repeat(b)
    slots = (cp.block_table[:b].to(torch.int64) * cp.BLOCK + torch.arange(t, device=dev)).reshape(-1)
    md = AttentionMetadata(num_actual_tokens=b * t, max_query_len=t,
                           query_start_loc=torch.arange(0, b * t + 1, t, dtype=torch.int32, device=dev),
                           seq_lens=torch.full((b,), t, dtype=torch.int32, device=dev), max_seq_len=t,
                           block_table=cp.block_table[:b], slot_mapping=slots, num_reqs=b, num_decodes=0)
    full = cp._stack(x, pos, md).view(b, t, h)
    last = torch.nn.functional.linear(full[:, g - 1], cp.lm_head[g - 2]).float()
    torch.testing.assert_close(last, logits[g - 2], rtol=3e-2, atol=3e-2)
    assert int(codes.min()) >= 0 and int(codes.max()) < cfg.vocab_size


def test_code_predictor_graph_replay_equals_eager(dev, stack_mode):
    from vllm_omni_amd.code_predictor import CodePredictorConfig, MI355XCodePredictor

    cfg = CodePredictorConfig(hidden_size=256, num_hidden_layers=5, num_attention_heads=4, num_key_value_heads=2,
                              head_dim=128, intermediate_size=512, vocab_size=512, num_code_groups=8)
    cp = MI355XCodePredictor(cfg, torch.bfloat16, dev, max_batch=8, seed=1)
    b = 6
    gen = torch.Generator(device=dev).manual_seed(4)
    mk = lambda: (torch.randint(0, 512, (b,), device=dev, generator=gen),  # noqa: E731
                  torch.randn(b, 256, device=dev, generator=gen).to(torch.bfloat16),
                  torch.randn(b, 256, device=dev, generator=gen).to(torch.bfloat16))
    seed = torch.arange(b, device=dev, dtype=torch.int64)
    a1, a2 = mk(), mk()
    off1, off2 = torch.full((b,), 5, device=dev, dtype=torch.int64), torch.full((b,), 6, device=dev, dtype=torch.int64)
    e1 = [t.clone() for t in cp(*a1, seed=seed, offset=off1)]
    e2 = [t.clone() for t in cp(*a2, seed=seed, offset=off2)]
    cp.capture(b)
    g1 = [t.clone() for t in cp(*a1, seed=seed, offset=off1)]
    g2 = [t.clone() for t in cp(*a2, seed=seed, offset=off2)]
    for x, y in zip(e1 + e2, g1 + g2):
        assert torch.equal(x, y)
    assert not torch.equal(e1[0], e2[0])


def test_embed_codes_is_a_gather_and_two_copies(dev):
    """``ops.embed_codes``: the predictor's per-step glue (store the code, embed it, append it to the sequence, feed the
    next step) in one launch — bit-equal to the indexing it replaces, strided destinations, out-of-range ids -> zero row."""
    from vllm_omni_amd import ops

    g = torch.Generator().manual_seed(2)
    V, E, B, G = 2048, 1024, 37, 8
    table = torch.randn(V, E, generator=g).to(torch.bfloat16).to(dev)
    ids = torch.randint(0, V, (B,), generator=g).to(dev)
    ids[3], ids[7] = -1, V
    buf = torch.zeros(B, G + 1, E, dtype=torch.bfloat16, device=dev)
    nxt = torch.full((B, E), 7.0, dtype=torch.bfloat16, device=dev)
    codes = torch.zeros(B, G, dtype=torch.int64, device=dev)
    ops.embed_codes(ids, table, buf[:, 4], nxt, codes[:, 3])
    want = table[ids.clamp(0, V - 1)].clone()
    want[3], want[7] = 0, 0
    assert torch.equal(buf[:, 4], want) and torch.equal(nxt, want) and torch.equal(codes[:, 3], ids)
    assert int(buf[:, :4].abs().sum()) == 0 and int(buf[:, 5:].abs().sum()) == 0 and int(codes[:, :3].abs().sum()) == 0
    ops.embed_codes(ids, table, nxt)          # single destination, no id copy
    assert torch.equal(nxt, want)
