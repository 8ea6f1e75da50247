"""CPU: Omni tensor prefix cache (A14), restating the reference's own test constants
(tests/core/test_prefix_cache.py:8-13: DEFAULT_SEQ_LEN 15, NUM_BLOCKS 10, BLOCK_SIZE 4, HIDDEN 2)."""
from types import SimpleNamespace

import pytest
import torch

SEQ, NB, BS, HID = 15, 10, 4, 2


def _cache():
    from vllm_omni_amd.prefix_cache import OmniTensorPrefixCache

    return OmniTensorPrefixCache(num_blocks=NB, block_size=BS, hidden_size=HID, hs_dtype=torch.float32)


def _batch(num_computed):
    from vllm_omni_amd.block_table import MultiGroupBlockTable

    bt = MultiGroupBlockTable(2, 40, 16, [BS], device=None)
    bt.add_row(([2, 3, 7],), 0)   # slots 8..15 live in blocks 2 and 3
    bt.add_row(([5],), 1)
    return SimpleNamespace(req_ids=["req1", "req2"], req_id_to_index={"req1": 0, "req2": 1},
                           num_computed_tokens_cpu=num_computed, block_table=bt)


def test_initialization():
    c = _cache()
    assert c.hidden_states_cache.shape == (NB, BS, HID) and not c.mm_outputs_cache and not c.mm_cache_keys
    mm = {"foo": torch.rand(SEQ, 100).half(), "bar": torch.rand(SEQ, 50), "baz": torch.rand(SEQ, 10), "n": "x"}
    c.maybe_init_missing_mm_cache_keys(mm, SEQ)
    assert c.mm_cache_keys == {"foo", "bar", "baz"}
    assert c.mm_outputs_cache["foo"].dtype == torch.float16 and c.mm_outputs_cache["bar"].shape == (NB, BS, 50)
    c.mm_outputs_cache["foo"].fill_(1)
    c.maybe_init_missing_mm_cache_keys(mm, SEQ)   # idempotent: does not re-zero
    assert torch.all(c.mm_outputs_cache["foo"] == 1)


@pytest.mark.parametrize("padded", [None, 16])
def test_update_and_merge_hidden_states(padded):
    c = _cache()
    slots = torch.arange(8, 16)
    orig = torch.rand(8, HID)
    c.update_omni_tensor_prefix_cache(orig, None, 8, slots, num_tokens_padded=padded)
    rows = c.hidden_states_cache.view(NB * BS, HID)
    for s, r in zip(slots, orig):
        assert torch.all(rows[s] == r)
    c.add_prefix_cached_new_req_id("req1")
    new = torch.rand(5, HID)
    merged = c.get_merged_hidden_states(query_start_loc=[0, 3], input_batch=_batch(torch.tensor([8, 0])),
                                        hidden_states=new, num_scheduled_tokens={"req1": 3, "req2": 2})
    assert merged["req1"].shape == (11, HID)
    assert torch.all(merged["req1"][:8] == orig) and torch.all(merged["req1"][-3:] == new[:3])
    assert torch.all(merged["req2"] == new[-2:])
    c.reset_prefix_cached_new_req_ids()
    again = c.get_merged_hidden_states(query_start_loc=[0, 3], input_batch=_batch(torch.tensor([8, 0])),
                                       hidden_states=new, num_scheduled_tokens={"req1": 3, "req2": 2})
    assert again["req1"].shape == (3, HID)


def test_update_and_merge_multimodal_outputs():
    c = _cache()
    feat = {"foo": 100, "bar": 50}
    c.maybe_init_missing_mm_cache_keys({k: torch.rand(SEQ, f) for k, f in feat.items()}, SEQ)
    slots = torch.arange(8, 16)
    orig = {k: torch.rand(8, f) for k, f in feat.items()}
    c.update_omni_tensor_prefix_cache(None, orig, 8, slots, num_tokens_padded=16)
    c.add_prefix_cached_new_req_id("req1")
    new = {k: torch.rand(5, f) for k, f in feat.items()}
    new["passthrough_data"] = "Something else"
    new["passthrough_list"] = ["should", "not", "split"]
    out = c.get_merged_multimodal_states([0, 3], _batch(torch.tensor([8, 0])), new, {"req1": 3, "req2": 2})
    assert out["passthrough_data"] == {"req1": "Something else", "req2": "Something else"}
    assert out["passthrough_list"]["req2"] == ["should", "not", "split"]
    for k, f in feat.items():
        assert out[k]["req1"].shape == (11, f)
        assert torch.all(out[k]["req1"][:8] == orig[k]) and torch.all(out[k]["req1"][-3:] == new[k][:3])
        assert torch.all(out[k]["req2"] == new[k][-2:])


def test_only_full_blocks_are_cached():
    c = _cache()
    ids = c._get_cached_block_ids(0, _batch(torch.tensor([11, 0])))   # 11 // 4 = 2 full blocks
    assert ids.tolist() == [2, 3]
    assert c._get_cached_block_ids(1, _batch(torch.tensor([11, 3]))).numel() == 0
