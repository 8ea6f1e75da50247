"""CPU: pins the ORACLE (tests/ may import oracle/; the product never does).

1. Indexing rows (A2/A3/A14): the constants of the reference's own tests, restated —
   tests/core/test_prefix_cache.py:8-13,112-133,173-185;
   tests/distributed/omni_connectors/test_kv_flow.py:57-66,98-141,254-342.
2. Arithmetic rows (A5-A9): the reference pins nothing here (SURVEY §8c "parity unpinned"); the oracle
   is cross-checked against outputs of the installed HF `transformers` talker (fixture
   tests/golden/hf_talker_mini.pt, generator tests/golden/make_golden.py).
"""
import json
import os
from types import SimpleNamespace

import pytest
import torch

from oracle import ref_ops, ref_sampler, ref_talker

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------ reference indexing constants
def test_prefix_cache_slot_identity_reference_constants():
    """NUM_BLOCKS=10, BLOCK_SIZE=4, HIDDEN=2; slots 8..15 <-> rows of blocks 2 and 3."""
    nb, bs, hid = 10, 4, 2
    cache = torch.zeros(nb, bs, hid)
    slot = torch.arange(8, 16)
    rows = torch.rand(8, hid)
    ref_ops.scatter_rows_by_slot(cache, slot, rows)
    flat = cache.view(nb * bs, hid)
    for s, r in zip(slot, rows):
        assert torch.all(flat[s] == r)
    # cached blocks of the hit request: block_table[req, :num_computed // block_size] == [2, 3]
    table = torch.tensor([[2, 3, 7, 0], [5, 0, 0, 0]])
    ids = ref_ops.cached_block_ids(table[0], num_computed=8, block_size=bs)
    assert ids.tolist() == [2, 3]
    assert torch.equal(cache[ids].reshape(-1, hid), rows)
    # and the identity slot = block * b + off
    sm = ref_ops.slot_mapping(table, torch.zeros(8, dtype=torch.long), torch.arange(8), bs)
    assert sm.tolist() == list(range(8, 16))
    assert ref_ops.cached_block_ids(table[1], num_computed=3, block_size=bs).numel() == 0  # only FULL blocks


@pytest.mark.parametrize("layout", ["stacked0", "stacked1", "tuple"])
def test_kv_extraction_reference_constants(layout):
    """num_heads 4, head_dim 16, block_size 8, seq_len 20, block_ids [1,3,5] -> [20, 4, 16]."""
    nb, bs, h, d, seq_len, ids = 10, 8, 4, 16, 20, [1, 3, 5]
    k = torch.randn(nb, bs, h, d)
    v = torch.randn(nb, bs, h, d)
    layer = {"stacked0": torch.stack([k, v], 0), "stacked1": torch.stack([k, v], 1), "tuple": (k, v)}[layout]
    kb, vb = ref_ops.normalize_layer_kv(layer)
    fk, fv = ref_ops.extract_kv(kb, vb, ids, seq_len)
    assert fk.shape == (seq_len, h, d) and fv.shape == (seq_len, h, d)
    assert torch.equal(fk, torch.cat([k[1], k[3], k[5]])[:seq_len])
    assert torch.equal(fv, torch.cat([v[1], v[3], v[5]])[:seq_len])


def test_kv_extraction_mismatched_block_counts():
    """key 3 blocks, value 2, ids [0,1,2], seq_len 32 -> overlapping range only: [2*bs, ...]."""
    bs, h, d = 8, 4, 16
    k, v = torch.randn(3, bs, h, d), torch.randn(2, bs, h, d)
    fk, fv = ref_ops.extract_kv(k, v, [0, 1, 2], 32)
    assert fk.shape == (2 * bs, h, d) and fv.shape == (2 * bs, h, d)
    assert ref_ops.extract_kv(k, v, [7, -1], 4) is None


@pytest.mark.parametrize("bad", ["invalid_stacked_shape", "invalid_tuple_length", "non_tensor_entries", "not_kv"])
def test_normalize_layer_kv_rejects_invalid(bad):
    bs, h, d = 8, 4, 16
    layer = {
        "invalid_stacked_shape": torch.randn(3, bs, h, d),
        "invalid_tuple_length": (torch.randn(2, bs, h, d),) * 3,
        "non_tensor_entries": (torch.randn(2, bs, h, d), "not-a-tensor"),
        "not_kv": [1, 2],
    }[bad]
    assert ref_ops.normalize_layer_kv(layer) is None


def test_normalize_layer_kv_precedence_dim0_first():
    """A [2, 2, ...] tensor is read as K/V at dim 0 (kv_utils.py:419-425 tests shape[0] first)."""
    t = torch.randn(2, 2, 4, 1, 8)
    kb, vb = ref_ops.normalize_layer_kv(t)
    assert torch.equal(kb, t[0]) and torch.equal(vb, t[1])


def test_truncate_block_ids():
    assert ref_ops.truncate_block_ids([4, 9, 2, 7], seq_len=17, block_size=8) == [4, 9, 2]
    assert ref_ops.truncate_block_ids([4, 9], seq_len=17, block_size=8) == [4, 9]
    assert ref_ops.truncate_block_ids([4, 9, 2], seq_len=0, block_size=8) == []


def test_bad_word_processor_branches():
    """qwen2_5_omni_talker.py:150-169, one case per branch."""
    v, end, bos = 20, 12, 11
    base = torch.zeros(2, v)
    eq = ref_ops.bad_word_processor(base.clone(), 12, end, bos)
    assert (eq[0] == -1e9).nonzero().flatten().tolist() == [11] + list(range(13, 20))
    lt = ref_ops.bad_word_processor(base.clone(), 9, end, bos)
    assert (lt[0] == -1e9).nonzero().flatten().tolist() == [9, 10, 11] + list(range(13, 20))
    gt = ref_ops.bad_word_processor(base.clone(), 15, end, bos)
    assert (gt[0] == -1e9).nonzero().flatten().tolist() == [11] + list(range(15, 20))
    none = ref_ops.bad_word_processor(base.clone(), None, end, bos)
    assert (none[0] == -1e9).nonzero().flatten().tolist() == [11]
    beyond = ref_ops.bad_word_processor(base.clone(), 25, end, None)
    assert (beyond == -1e9).sum() == 0
    with pytest.raises(ValueError):
        ref_ops.bad_word_processor(base.clone(), 9, None, bos)


# ------------------------------------------------------------------ HF cross-check (arithmetic)
@pytest.fixture(scope="module")
def hf():
    fx = torch.load(os.path.join(GOLD, "hf_talker_mini.pt"), weights_only=False)
    c = fx["config"]
    cfg = SimpleNamespace(qk_norm=False, tts_codec_end_token_id=None, tts_codec_start_token_id=None, **c)
    w = {k: v.float() for k, v in fx["weights_fp16"].items()}
    return cfg, w, fx["inputs_embeds_fp16"].float(), fx["position_ids"], fx["expected_hidden_fp32"]


def _paged(cfg, T, bs, seed):
    g = torch.Generator().manual_seed(seed)
    nblk = -(-T // bs)
    nb = nblk + 4
    caches = [(torch.full((nb, bs, cfg.num_key_value_heads, cfg.head_dim), float("nan")),
               torch.full((nb, bs, cfg.num_key_value_heads, cfg.head_dim), float("nan"))) for _ in range(cfg.num_hidden_layers)]
    table = (torch.randperm(nb - 1, generator=g)[:nblk] + 1).view(1, nblk).to(torch.int32)
    return caches, table


def test_oracle_prefill_matches_hf_talker(hf):
    cfg, w, x, pos, exp = hf
    T, bs = x.shape[0], 4
    caches, table = _paged(cfg, T, bs, 0)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
    slots = ref_ops.slot_mapping(table, torch.zeros(T, dtype=torch.long), torch.arange(T), bs)
    got = ref_talker.decoder_stack(cfg, w, x, pos, caches, table, torch.tensor([T], dtype=torch.int32),
                                   torch.tensor([0, T], dtype=torch.int32), slots, csc, None)
    torch.testing.assert_close(got, exp, rtol=1e-4, atol=1e-4)


def test_oracle_token_by_token_paged_decode_matches_hf_talker(hf):
    """Full-sequence causal output must equal token-by-token paged decode."""
    cfg, w, x, pos, exp = hf
    T, bs = x.shape[0], 4
    caches, table = _paged(cfg, T, bs, 1)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
    outs = []
    for t in range(T):
        slot = ref_ops.slot_mapping(table, torch.zeros(1, dtype=torch.long), torch.tensor([t]), bs)
        h = ref_talker.decoder_stack(cfg, w, x[t:t + 1], pos[:, t:t + 1], caches, table,
                                     torch.tensor([t + 1], dtype=torch.int32), torch.tensor([0, 1], dtype=torch.int32),
                                     slot, csc, None)
        outs.append(h)
    torch.testing.assert_close(torch.cat(outs), exp, rtol=1e-4, atol=1e-4)


def test_oracle_decode_and_prefill_attention_agree():
    g = torch.Generator().manual_seed(2)
    S, nq, nkv, d, bs = 3, 8, 2, 32, 4
    lens = [1, 9, 14]
    nb = 16
    kb, vb = torch.randn(nb, bs, nkv, d, generator=g), torch.randn(nb, bs, nkv, d, generator=g)
    table = (torch.randperm(nb - 1, generator=g)[:12] + 1).view(3, 4).to(torch.int32)
    q = torch.randn(S, nq, d, generator=g)
    sl = torch.tensor(lens, dtype=torch.int32)
    a = ref_ops.paged_attention_decode(q, kb, vb, table, sl, d ** -0.5)
    b = ref_ops.paged_attention_prefill(q, kb, vb, table, sl, torch.arange(S + 1, dtype=torch.int32), d ** -0.5)
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


def test_rope_mrope_equal_rows_is_plain_rope():
    csc = ref_ops.build_cos_sin_cache(32, 128, 1e6)
    x = torch.randn(5, 4 * 32)
    p = torch.tensor([0, 3, 17, 100, 127])
    a = ref_ops.rope_neox(p, x, 32, csc)
    b = ref_ops.rope_neox(p.repeat(3, 1), x, 32, csc, mrope_section=[4, 6, 6])
    assert torch.equal(a, b)
    assert torch.equal(ref_ops.rope_neox(torch.zeros(5, dtype=torch.long), x, 32, csc), x)  # position 0 = identity


def test_int8_scheme_saturation_and_scales():
    x = torch.tensor([[0.0, 1.0, -2.0, 127.0], [0.0, 0.0, 0.0, 0.0], [1e-3, -1e-3, 5e-4, 2.5e-4]])
    q, s = ref_ops.dynamic_int8_quant(x)
    assert q[0].tolist() == [0, 1, -2, 127] and abs(float(s[0]) - 1.0) < 1e-7
    assert q[1].tolist() == [0, 0, 0, 0] and float(s[1]) == 1.0        # all-zero row: scale 1, no NaN
    assert q[2].tolist() == [127, -127, 64, 32]                         # 63.5 -> 64, 31.75 -> 32
    w = torch.randn(6, 4, generator=torch.Generator().manual_seed(3))
    wq, ws = ref_ops.int8_weight_quant(w)
    y = ref_ops.int8_linear(x, wq, ws)
    # error bound: each of the K products is off by at most (|x| dw + |w| dx), dx = sx/2, dw = sw/2
    bound = (x.abs() @ (ws.view(-1, 1) / 2).expand_as(w).t()) + ((s.view(-1, 1) / 2).expand_as(x) @ w.abs().t())
    assert ((y - x @ w.t()).abs() <= bound * 1.01 + 1e-6).all()


# ------------------------------------------------------------------ sampler oracle (N2)
def test_philox4x32_10_known_answer_vectors():
    """Random123's kat_vectors for philox4x32-10 (Salmon et al. 2011): zero, all-ones, digits of pi."""
    import numpy as np

    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kat:
        got = ref_sampler.philox4x32_10(ctr, key)
        assert tuple(int(x) for x in got) == want
    u = ref_sampler.uniform_noise(4096, seed=11, offset=2**40 + 5)
    assert u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.03 and len(np.unique(u)) > 4000


def test_sampler_oracle_semantics():
    """The rules upstream's `apply_penalties` / `apply_top_k_top_p` / `random_sample` state, one case each."""
    import numpy as np

    lg = np.array([2.0, 1.0, 1.0, -1.0, 0.5], dtype=np.float32)
    assert ref_sampler.sample_row(lg, 0.0)[0] == 0                                   # greedy, first maximum
    assert ref_sampler.sample_row(np.array([1.0, 3.0, 3.0], np.float32), 0.0)[0] == 1
    seen = np.array([True, False, False, True, False])
    # repetition penalty: positive logits divided, negative multiplied, unseen untouched -> 1.0 wins over 2/4
    assert ref_sampler.sample_row(lg, 0.0, rep_penalty=4.0, seen=seen)[0] == 1
    # top-k keeps ties with the k-th value: k = 2 keeps {0, 1, 2}; draws never leave that set
    ids = {ref_sampler.sample_row(lg, 1.0, top_k=2, seed=1, offset=o)[0] for o in range(400)}
    assert ids == {0, 1, 2}
    # top-p keeps an element while the mass ranked before it is < p, at least one
    assert {ref_sampler.sample_row(lg, 1.0, top_p=1e-9, seed=1, offset=o)[0] for o in range(50)} == {0}
    p = np.exp(lg.astype(np.float64)); p /= p.sum()
    ids = {ref_sampler.sample_row(lg, 1.0, top_p=float(p[0]) + 1e-6, seed=2, offset=o)[0] for o in range(600)}
    assert ids == {0, 1, 2}                                                            # the tied pair enters together
    # the exponential race is an exact sampler: frequencies follow p
    draws = np.array([ref_sampler.sample_row(lg, 1.0, seed=5, offset=o)[0] for o in range(4000)])
    freq = np.bincount(draws, minlength=5) / 4000
    assert np.abs(freq - p).max() < 0.03


# ------------------------------------------------------------------ code predictor oracle (N4)
def test_code_predictor_oracle_loop_properties():
    """The restated re-prefill loop (qwen3_code_predictor.py:503-626): the buffer holds [talker hidden, emb(c0),
    emb(c1), ...]; teacher forcing with the codes a free run drew reproduces its logits exactly; positions past
    the frontier never influence a prediction (causality), so zero-filling them — as the reference does — is
    equivalent to truncating the sequence."""
    from types import SimpleNamespace

    from oracle import ref_code_predictor as rcp

    cfg = SimpleNamespace(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=32,
                          intermediate_size=96, vocab_size=40, num_code_groups=5, rms_norm_eps=1e-6, rope_theta=10000.0)
    w = rcp.random_weights(cfg, seed=1, dtype=torch.float32)
    g = torch.Generator().manual_seed(0)
    code0 = torch.randint(0, 40, (3, 1), generator=g)
    emb0, hid = torch.randn(3, 1, 64, generator=g), torch.randn(3, 64, generator=g)
    codes, buf, logits = rcp.forward(cfg, w, code0, emb0, hid, seed=[5, 6, 7], offset=[0, 1, 2])
    assert codes.shape == (3, 5, 1) and buf.shape == (3, 6, 64) and len(logits) == 4
    assert torch.equal(codes[:, 0], code0) and torch.equal(buf[:, 0], hid) and torch.equal(buf[:, 1], emb0[:, 0])
    for s in range(1, 5):
        assert torch.equal(buf[:, s + 1], w[f"model.codec_embedding.{s - 1}.weight"][codes[:, s, 0]])
    codes2, buf2, logits2 = rcp.forward(cfg, w, code0, emb0, hid, forced_codes=codes[:, :, 0])
    assert torch.equal(codes2, codes) and all(torch.equal(a, b) for a, b in zip(logits, logits2))
    # causality: the hidden state at position s of the zero-padded buffer == that of the sequence cut at s + 1
    full = rcp.stack(cfg, w, buf[:, :5])
    cut = rcp.stack(cfg, w, buf[:, :3])
    torch.testing.assert_close(full[:, :3], cut, rtol=1e-5, atol=1e-5)
    greedy, _, gl = rcp.forward(cfg, w, code0, emb0, hid, do_sample=False)
    assert all(torch.equal(greedy[:, s, 0], gl[s - 1].argmax(-1)) for s in range(1, 5))


# ------------------------------------------------------------------ Qwen3-Omni talker modules pinned by HF outputs
@pytest.fixture(scope="module")
def hf_q3():
    return torch.load(os.path.join(GOLD, "hf_qwen3_talker_mini.pt"), weights_only=False)


def test_code_predictor_oracle_stack_matches_hf(hf_q3):
    """oracle/ref_code_predictor.stack == transformers' Qwen3OmniMoeTalkerCodePredictorModel (the module the
    reference's code predictor says it re-implements exactly)."""
    from oracle import ref_code_predictor as rcp

    fx = hf_q3["code_predictor"]
    cfg = SimpleNamespace(**fx["config"])
    w = {k: v.float() for k, v in fx["weights_fp16"].items()}
    got = rcp.stack(cfg, w, fx["inputs_embeds_fp16"].float())
    torch.testing.assert_close(got, fx["expected_hidden_fp32"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["moe", "moe_norm_topk"])
def test_moe_oracle_matches_hf_block(hf_q3, tag):
    """oracle/ref_moe.moe_block == transformers' Qwen3OmniMoeTalkerTextSparseMoeBlock: fp32 softmax router, top-k
    (with and without renormalisation), per-expert SiLU-gated MLPs, sigmoid-gated shared expert."""
    from oracle import ref_moe

    fx = hf_q3[tag]
    w = {k: v.float() for k, v in fx["weights_fp16"].items()}
    c = fx["config"]
    got = ref_moe.moe_block(fx["x_fp16"].float(), w, c["num_experts_per_tok"], c["norm_topk_prob"])
    torch.testing.assert_close(got, fx["expected_fp32"], rtol=1e-4, atol=1e-5)
    rw, re = ref_moe.route(fx["x_fp16"].float(), w["gate.weight"], c["num_experts_per_tok"], c["norm_topk_prob"])
    assert re.shape == (19, 3) and (rw[:, :-1] >= rw[:, 1:]).all()
    assert (abs(rw.sum(-1) - 1) < 1e-6).all() == c["norm_topk_prob"]
