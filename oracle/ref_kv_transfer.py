"""ORACLE (test infrastructure only — the product never imports this): plain restatement of the KV hand-off
formats of vllm_omni/distributed/omni_connectors, written with numpy / struct / json loops.

* `extract` — `_extract_kv_cache`, kv_transfer_manager.py:786-862: out-of-range block ids dropped, the shared
  block range used when K and V disagree in length, rows truncated to seq_len only when it is shorter.
* `slice_heads` — `slice_kv_tensor_heads`, utils/kv_utils.py:230-259 (heads are dim 1).
* `to_bytes` / `from_bytes` — `KVCacheTransferData.to_bytes` / `from_bytes`, kv_transfer_manager.py:89-142,
  159-260: 4-byte big-endian header length, compact JSON header, key layers then value layers.
* `target_ranks` / `source_ranks` — utils/kv_utils.py:131-153.

Pinned by the constants of the reference's own tests (tests/distributed/omni_connectors/test_kv_flow.py:57-186,
test_tp_rank_aware.py:100-163,447-503), restated in tests/test_kv_transfer.py.
"""
from __future__ import annotations

import json
import struct

import numpy as np


def extract(key_blocks: np.ndarray, value_blocks: np.ndarray, block_ids, seq_len: int):
    """[num_blocks, block_size, heads, d] x2 -> ([rows, heads, d], [rows, heads, d]) or None."""
    top = min(key_blocks.shape[0], value_blocks.shape[0]) - 1
    valid = [b for b in block_ids if 0 <= b <= top]
    if not valid:
        return None
    k = np.concatenate([key_blocks[b] for b in valid], axis=0)
    v = np.concatenate([value_blocks[b] for b in valid], axis=0)
    if seq_len < k.shape[0]:
        k, v = k[:seq_len], v[:seq_len]
    return np.ascontiguousarray(k), np.ascontiguousarray(v)


def slice_heads(t: np.ndarray, offset_in_shard: int, num_slices: int) -> np.ndarray:
    if num_slices <= 0 or not 0 <= offset_in_shard < num_slices or t.shape[1] % num_slices:
        raise ValueError("bad head slice")
    per = t.shape[1] // num_slices
    return np.ascontiguousarray(t[:, offset_in_shard * per:(offset_in_shard + 1) * per])


def to_bytes(request_id, key_layers, value_layers, block_ids, metadata, dtype_names=None) -> bytes:
    """key_layers / value_layers: lists of numpy arrays or None.  dtype_names[i]: torch dtype name to write for
    arrays numpy cannot type (bf16 passed as uint16 views)."""
    desc, chunks, pos = [], [], 0
    for name, layers in (("key_cache", key_layers), ("value_cache", value_layers)):
        for i, a in enumerate(layers):
            if a is None:
                desc.append({"n": f"{name}_{i}", "x": True})
                continue
            raw = np.ascontiguousarray(a).tobytes()
            dt = (dtype_names or {}).get(i) or str(a.dtype)
            desc.append({"n": f"{name}_{i}", "i": i, "d": dt, "s": list(a.shape), "o": pos, "b": len(raw)})
            chunks.append(raw)
            pos += len(raw)
    header = json.dumps({"rid": request_id, "bids": list(block_ids), "meta": metadata, "td": desc, "nl": len(key_layers)},
                        separators=(",", ":")).encode("utf-8")
    return struct.pack(">I", len(header)) + header + b"".join(chunks)


_NP = {"float32": np.float32, "float16": np.float16, "float64": np.float64, "int8": np.int8, "uint8": np.uint8,
       "int16": np.int16, "int32": np.int32, "int64": np.int64, "bfloat16": np.uint16}


def from_bytes(raw: bytes):
    """-> (header dict, {"key_cache": [...], "value_cache": [...]}) with the reference's validation errors."""
    if len(raw) < 4:
        raise ValueError("Corrupted KV payload: missing 4-byte header length")
    (hl,) = struct.unpack(">I", raw[:4])
    if hl > len(raw) - 4:
        raise ValueError(f"Corrupted KV payload: header_len={hl} exceeds buffer size={len(raw)}")
    header = json.loads(raw[4:4 + hl])
    data = raw[4 + hl:]
    nl = header["nl"]
    out = {"key_cache": [None] * nl, "value_cache": [None] * nl}
    for info in header["td"]:
        if info.get("x"):
            continue
        off, nb = info["o"], info["b"]
        if off < 0 or nb < 0 or off + nb > len(data):
            raise ValueError(f"Corrupted KV payload tensor span for {info['n']}")
        if info["d"] not in _NP:
            raise ValueError(f"Unsupported dtype in KV payload: {info['d']}")
        arr = np.frombuffer(data, dtype=_NP[info["d"]], count=nb // np.dtype(_NP[info["d"]]).itemsize, offset=off)
        idx = info.get("i")
        if idx is None:
            idx = int(info["n"].split("_")[-1])
        if not isinstance(idx, int) or not 0 <= idx < nl:
            raise ValueError(f"Invalid layer index in KV payload: {idx}")
        out["key_cache" if info["n"].startswith("key_cache_") else "value_cache"][idx] = arr.reshape(info["s"])
    return header, out


def target_ranks(source_tp: int, target_tp: int, rank: int) -> list[int]:
    if max(source_tp, target_tp) % min(source_tp, target_tp):
        raise ValueError("KV TP mapping must be divisible")
    if source_tp >= target_tp:
        return [rank // (source_tp // target_tp)]
    r = target_tp // source_tp
    return [rank * r + i for i in range(r)]


def source_ranks(source_tp: int, target_tp: int, rank: int) -> list[int]:
    if max(source_tp, target_tp) % min(source_tp, target_tp):
        raise ValueError("KV TP mapping must be divisible")
    if source_tp > target_tp:
        r = source_tp // target_tp
        return [rank * r + i for i in range(r)]
    return [rank // (target_tp // source_tp)]
