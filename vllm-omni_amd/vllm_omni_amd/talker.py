"""Host-side mirror of the talker model wiring (filled in below)."""
from __future__ import annotations

from typing import Optional


def bad_word_mask_spec(vocab: int, suppress_start_id: Optional[int], end_id: Optional[int],
                       bos_id: Optional[int]) -> tuple[list[tuple[int, int]], int]:
    """Column ranges / single column that ``bad_word_processor`` sets to -1e9.

    Branch-for-branch restatement of
    vllm_omni/model_executor/models/qwen2_5_omni/qwen2_5_omni_talker.py:150-169 as data for the
    fused mask+argmax kernel (``ops.mask_logits_argmax``).
    """
    ranges: list[tuple[int, int]] = []
    if suppress_start_id and suppress_start_id < vocab:
        if end_id is None:
            raise ValueError("config must have tts_codec_end_token_id attribute")
        if suppress_start_id == end_id:
            ranges.append((end_id + 1, vocab))
        elif suppress_start_id < end_id:
            ranges.append((suppress_start_id, end_id))
            ranges.append((end_id + 1, vocab))
        else:
            ranges.append((suppress_start_id, vocab))
    return ranges, (-1 if bos_id is None else int(bos_id))
