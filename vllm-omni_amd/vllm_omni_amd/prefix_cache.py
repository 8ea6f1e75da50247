"""Omni tensor prefix cache: hidden states / multimodal outputs cached on the host in
tensors shaped like the KV cache's (num_blocks, block_size) grid, so the scheduler's slot mapping and
block ids index them directly.

Mirror of ``OmniTensorPrefixCache`` (vllm_omni/core/prefix_cache.py:14-264) — same constructor,
method names and semantics: rows are written at ``slot_mapping`` (``flat[slot] = state``), a cache-hit
request reads ``cache[block_table[req, :num_computed // block_size]]`` (FULL blocks only) and gets
``cat([cached, new])``.  Host-side indexing only; the device work is the D2H copy.
"""
from __future__ import annotations

from typing import Optional

import torch


class OmniTensorPrefixCache:
    def __init__(self, num_blocks: int, block_size: int, hidden_size: int, hs_dtype: torch.dtype):
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.default_hidden_size = hidden_size
        self.hidden_states_cache = self._get_cache_tensor(dtype=hs_dtype)
        # multimodal output caches appear lazily, once a tensor with a per-token first dim is seen
        self.mm_outputs_cache: dict[str, torch.Tensor] = {}
        self.mm_cache_keys: set[str] = set()
        self._new_req_cache_hit_ids: set[str] = set()

    def _get_cache_tensor(self, dtype: torch.dtype, hidden_size: Optional[int] = None) -> torch.Tensor:
        h = hidden_size if hidden_size is not None else self.default_hidden_size
        return torch.zeros((self.num_blocks, self.block_size, h), dtype=dtype, device="cpu")

    def maybe_init_missing_mm_cache_keys(self, multimodal_outputs: dict, seq_len: int) -> None:
        for key, val in multimodal_outputs.items():
            if isinstance(val, torch.Tensor) and val.shape[0] == seq_len and key not in self.mm_cache_keys:
                self.mm_outputs_cache[key] = self._get_cache_tensor(dtype=val.dtype, hidden_size=val.shape[-1])
                self.mm_cache_keys.add(key)

    def add_prefix_cached_new_req_id(self, req_id: str) -> None:
        self._new_req_cache_hit_ids.add(req_id)

    def reset_prefix_cached_new_req_ids(self) -> None:
        self._new_req_cache_hit_ids.clear()

    @staticmethod
    def _coerce_to_cpu_tensor(t: torch.Tensor) -> torch.Tensor:
        return t.detach().cpu().contiguous()

    def update_omni_tensor_prefix_cache(self, hidden_states: Optional[torch.Tensor],
                                        multimodal_outputs: Optional[dict], num_tokens_unpadded: int,
                                        slot_mapping: torch.Tensor, num_tokens_padded: Optional[int] = None) -> None:
        slots = slot_mapping[:num_tokens_unpadded].to("cpu")
        if num_tokens_padded is None:
            num_tokens_padded = num_tokens_unpadded
        if hidden_states is not None:
            hs = self._coerce_to_cpu_tensor(hidden_states[:num_tokens_unpadded])
            flat = self.hidden_states_cache.view(-1, self.hidden_states_cache.shape[-1])
            flat[slots] = hs
        if multimodal_outputs is not None:
            self.maybe_init_missing_mm_cache_keys(multimodal_outputs, seq_len=num_tokens_padded)
            for key, cache in self.mm_outputs_cache.items():
                if key in multimodal_outputs:
                    st = self._coerce_to_cpu_tensor(multimodal_outputs[key][:num_tokens_unpadded])
                    cache.view(-1, cache.shape[-1])[slots] = st

    def _get_cached_block_ids(self, req_idx: int, input_batch) -> torch.Tensor:
        num_computed = int(input_batch.num_computed_tokens_cpu[req_idx])
        n = num_computed // self.block_size  # only full blocks are cached
        return input_batch.block_table[0].block_table.cpu[req_idx, :n]

    def _get_merged_tensors(self, query_start_loc, input_batch, cache: torch.Tensor, hidden_states: torch.Tensor,
                            num_scheduled_tokens: dict[str, int]) -> dict[str, torch.Tensor]:
        out = {}
        hs = self._coerce_to_cpu_tensor(hidden_states)
        for req_id in input_batch.req_ids:
            i = input_batch.req_id_to_index[req_id]
            start = int(query_start_loc[i])
            new = hs[start:start + num_scheduled_tokens[req_id]]
            if req_id in self._new_req_cache_hit_ids:
                blocks = self._get_cached_block_ids(i, input_batch)
                cached = cache[blocks.to(torch.int64)].reshape(-1, cache.shape[-1])
                out[req_id] = torch.cat([cached, new], dim=0)
            else:
                out[req_id] = new
        return out

    def get_merged_hidden_states(self, *args, **kwargs) -> dict[str, torch.Tensor]:
        return self._get_merged_tensors(*args, **kwargs, cache=self.hidden_states_cache)

    def get_merged_multimodal_states(self, query_start_loc, input_batch, multimodal_outputs: dict,
                                     num_scheduled_tokens: dict[str, int]) -> dict:
        merged: dict = {}
        for key in self.mm_cache_keys:
            if key in multimodal_outputs:
                merged[key] = self._get_merged_tensors(query_start_loc, input_batch, self.mm_outputs_cache[key],
                                                       multimodal_outputs[key], num_scheduled_tokens)
        for key, val in multimodal_outputs.items():
            if key in self.mm_cache_keys:
                continue
            # passthrough data is handed to every request unchanged (lists are NOT split here)
            merged[key] = {req_id: val for req_id in input_batch.req_ids}
        return merged
