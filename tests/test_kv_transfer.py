"""CPU: the KV hand-off formats — oracle AND host-side mirror against the constants of the
reference's own tests: tests/distributed/omni_connectors/test_kv_flow.py:57-186 (extraction shape, payload
validation) and test_tp_rank_aware.py:100-163,447-503 (rank routing, head merge / slice)."""
import json
import struct

import numpy as np
import pytest
import torch

from oracle import ref_kv_transfer as ref
from vllm_omni_amd import kv_transfer as kt
from vllm_omni_amd.kv_cache import extract_request_kv

C = {"num_layers": 2, "num_heads": 4, "head_dim": 16, "block_size": 8, "seq_len": 20, "req_id": "req_test_1"}


def _caches(seed=0, num_blocks=10):
    g = torch.Generator().manual_seed(seed)
    return [torch.stack([torch.randn(num_blocks, C["block_size"], C["num_heads"], C["head_dim"], generator=g),
                         torch.randn(num_blocks, C["block_size"], C["num_heads"], C["head_dim"], generator=g)], dim=0)
            for _ in range(C["num_layers"])]


def test_extraction_and_payload_roundtrip_reference_constants():
    """block_ids [1, 3, 5], block 8, seq_len 20 -> [20, 4, 16] per layer; bytes round-trip; mirror == oracle."""
    caches = _caches()
    ids = [1, 3, 5]
    keys, vals = extract_request_kv(caches, ids, C["seq_len"])
    meta = {"block_size": C["block_size"], "num_layers": C["num_layers"], "dtype": "float32", "seq_len": C["seq_len"]}
    data = kt.KVCacheTransferData(C["req_id"], {"key_cache": keys, "value_cache": vals}, ids, meta)
    raw = data.to_bytes()
    ok = [ref.extract(c[0].numpy(), c[1].numpy(), ids, C["seq_len"]) for c in caches]
    assert raw == ref.to_bytes(C["req_id"], [k for k, _ in ok], [v for _, v in ok], ids, meta)      # byte-identical
    back = kt.KVCacheTransferData.from_bytes(raw)
    assert back["request_id"] == C["req_id"] and back["block_ids"] == ids and back["metadata"] == meta
    assert len(back["layer_blocks"]["key_cache"]) == C["num_layers"]
    assert back["layer_blocks"]["key_cache"][0].shape == (C["seq_len"], C["num_heads"], C["head_dim"])
    for i, c in enumerate(caches):
        want = torch.cat([c[0][1], c[0][3], c[0][5]])[:C["seq_len"]]
        assert torch.equal(back["layer_blocks"]["key_cache"][i], want)
    hdr, arrs = ref.from_bytes(raw)
    assert hdr["nl"] == C["num_layers"] and np.array_equal(arrs["value_cache"][1], vals[1].numpy())


def _payload():
    key = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    return kt.KVCacheTransferData("req-payload", {"key_cache": [key], "value_cache": [None]}, [1], {"seq_len": 3}).to_bytes(), key


def _rewrite(payload: bytes, mutate) -> bytes:
    hl = struct.unpack(">I", payload[:4])[0]
    header = json.loads(payload[4:4 + hl])
    mutate(header)
    new = json.dumps(header, separators=(",", ":")).encode()
    return struct.pack(">I", len(new)) + new + payload[4 + hl:]


def test_payload_validation_matches_the_reference_errors():
    payload, key = _payload()
    bad_len = struct.pack(">I", len(payload)) + payload[4:]
    bad_span = _rewrite(payload, lambda h: h["td"][0].update({"o": 4096}))
    bad_dtype = _rewrite(payload, lambda h: h["td"][0].update({"d": "cuda"}))
    for bad, msg in ((bad_len, "header_len"), (bad_span, "tensor span"), (bad_dtype, "Unsupported dtype")):
        with pytest.raises(ValueError, match=msg):
            kt.KVCacheTransferData.from_bytes(bad)
        with pytest.raises(ValueError, match=msg):
            kt.KVCacheTransferData.from_bytes_gpu(torch.tensor(list(bad), dtype=torch.uint8))
        with pytest.raises(ValueError):
            ref.from_bytes(bad)
    with pytest.raises(ValueError, match="1-D uint8"):
        kt.KVCacheTransferData.from_bytes_gpu(torch.zeros(4, 4, dtype=torch.uint8))
    # an explicit "i" wins over the name suffix; an absent layer ("x": true) stays None
    explicit = _rewrite(payload, lambda h: h["td"][0].update({"n": "key_cache_extra_suffix", "i": 0}))
    got = kt.KVCacheTransferData.from_bytes(explicit)
    assert torch.equal(got["layer_blocks"]["key_cache"][0], key) and got["layer_blocks"]["value_cache"][0] is None
    # a header padded with spaces (what pack_request_kv writes) is ordinary JSON
    padded = _rewrite(payload, lambda h: None)
    hl = struct.unpack(">I", padded[:4])[0]
    padded = struct.pack(">I", hl + 7) + padded[4:4 + hl] + b" " * 7 + padded[4 + hl:]
    assert torch.equal(kt.KVCacheTransferData.from_bytes(padded)["layer_blocks"]["key_cache"][0], key)
    assert np.array_equal(ref.from_bytes(padded)[1]["key_cache"][0], key.numpy())


def test_rank_routing_reference_constants():
    T = kt.KVTPTopology
    assert kt.get_kv_connector_key("req-1", "stage0", 0, 1, 2) == "req-1_stage0_0_1_2"
    assert kt.get_kv_connector_key("r", "s", 5, 3, 7).split("_") == ["r", "s", "5", "3", "7"]
    table = [((2, 2, 0), [0], [0]), ((2, 2, 1), [1], [1]), ((4, 4, 3), [3], [3]),
             ((4, 2, 0), None, [0, 1]), ((4, 2, 1), None, [2, 3]),
             ((2, 4, 0), [0, 1], [0]), ((2, 4, 1), [2, 3], None)]
    for (s, t, r), targets, sources in table:
        if targets is not None:
            assert kt.get_kv_target_ranks(T(s, t, r)) == targets == ref.target_ranks(s, t, r)
        if sources is not None:
            assert kt.get_kv_source_ranks(T(s, t, r)) == sources == ref.source_ranks(s, t, r)
    with pytest.raises(ValueError, match="divisible"):
        kt.get_kv_source_ranks(T(3, 2, 0))
    assert kt.build_rank_aware_send_keys("r", "a", "b", T(1, 1, 0)) == ["omni_a_to_b_kv_cache_r"]
    assert kt.build_rank_aware_recv_keys("r", "a", "b", T(4, 2, 1)) == [("r_a_0_2_1", 2), ("r_a_0_3_1", 3)]


def _mk(head_values, request_id="req-1"):
    t = torch.tensor(head_values, dtype=torch.float32).view(1, len(head_values), 1).repeat(2, 1, 1)
    return {"request_id": request_id, "layer_blocks": {"key_cache": [t.clone()], "value_cache": [(t + 100).clone()]},
            "block_ids": [0], "metadata": {"seq_len": 2}}


def test_head_merge_and_slice_reference_constants():
    T = kt.KVTPTopology
    one = {"layer_blocks": {"key_cache": [1]}}
    assert kt.merge_received_rank_shards([one]) == one
    m = kt.merge_received_rank_shards([_mk([0.0]), _mk([1.0])])
    k, v = m["layer_blocks"]["key_cache"][0], m["layer_blocks"]["value_cache"][0]
    assert k.shape == (2, 2, 1) and torch.equal(k[:, :, 0], torch.tensor([[0.0, 1.0], [0.0, 1.0]]))
    assert torch.equal(v[:, :, 0], torch.tensor([[100.0, 101.0], [100.0, 101.0]]))
    assert kt.merge_received_rank_shards([{}, {}], merger=lambda p: {"merged": True}) == {"merged": True}
    assert kt.slice_received_rank_shard({"full": True}, T(2, 4, 0), slicer=lambda p: {"sliced": True}) == {"sliced": True}
    s = kt.slice_received_rank_shard(_mk([0.0, 1.0]), T(2, 4, 1))
    assert s["layer_blocks"]["key_cache"][0].shape == (2, 1, 1)
    assert torch.equal(s["layer_blocks"]["key_cache"][0][:, :, 0], torch.tensor([[1.0], [1.0]]))
    assert torch.equal(s["layer_blocks"]["value_cache"][0][:, :, 0], torch.tensor([[101.0], [101.0]]))
    pre = _mk([1.0])
    pre["metadata"]["tp_head_slice"] = {"applied": True, "target_rank": 1}
    assert kt.slice_received_rank_shard(pre, T(2, 4, 1)) is pre
    merged = kt.merge_received_rank_shards([_mk([float(r)]) for r in kt.get_kv_source_ranks(T(4, 2, 1))])
    assert torch.equal(merged["layer_blocks"]["key_cache"][0][:, :, 0], torch.tensor([[2.0, 3.0], [2.0, 3.0]]))
    s3 = kt.slice_received_rank_shard(_mk([2.0, 3.0]), T(2, 4, 3))
    assert torch.equal(s3["layer_blocks"]["key_cache"][0][:, :, 0], torch.tensor([[3.0], [3.0]]))
    # the oracle slices the same heads
    a = np.arange(2 * 4 * 3, dtype=np.float32).reshape(2, 4, 3)
    assert np.array_equal(ref.slice_heads(a, 1, 2), kt.slice_kv_tensor_heads(torch.from_numpy(a), 1, 2).numpy())
    with pytest.raises(ValueError, match="divisible"):
        kt.slice_kv_tensor_heads(torch.zeros(2, 3, 1), 0, 2)
    with pytest.raises(ValueError, match="offset_in_shard"):
        kt.slice_kv_tensor_heads(torch.zeros(2, 4, 1), 2, 2)


def test_pack_request_kv_refuses_host_tensors():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        kt.pack_request_kv(_caches(), "r", [1], 4, C["block_size"], "float32")
