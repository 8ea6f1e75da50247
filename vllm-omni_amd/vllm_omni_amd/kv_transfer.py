"""KV hand-off between stages: the data formats either side of the decode path.

Host-side mirror of the reference's interface for this step — same names, argument meaning and error
behaviour, so a maintainer can swap the imports:

* ``KVTPTopology`` / ``validate_kv_tp_topology`` / ``get_kv_target_ranks`` / ``get_kv_source_ranks`` /
  ``get_kv_connector_key`` / ``build_rank_aware_{send,recv}_keys`` — vllm_omni/distributed/omni_connectors/utils/
  kv_utils.py:116-222;
* ``slice_kv_tensor_heads`` / ``slice_layer_blocks`` / ``merge_received_rank_shards`` /
  ``slice_received_rank_shard`` — kv_utils.py:230-375 (KV heads are dim 1 of ``[seq, heads, head_dim]``);
* ``KVCacheTransferData`` with ``to_bytes`` / ``to_gpu_tensor`` / ``from_bytes`` / ``from_bytes_gpu`` —
  kv_transfer_manager.py:76-272.  Wire format: 4-byte big-endian header length, compact JSON
  ``{"rid","bids","meta","td":[{"n","i","d","s","o","b"} | {"n","x":true}],"nl"}``, then every key layer
  and every value layer as raw bytes at offset ``o``.

``pack_request_kv`` is the MI355X path for raw-data connectors: extraction, the sender-side head slice and the
serialisation in ONE kernel launch (csrc/kv_pack.hip) straight into the payload tensor.  Its header is padded
with spaces to a 16-byte boundary so the data region can be written with 16-byte stores; trailing whitespace is
ordinary JSON, ``from_bytes`` / ``from_bytes_gpu`` (here and in the reference) read it unchanged.
"""
from __future__ import annotations

import json
import struct
import warnings
from dataclasses import asdict, dataclass
from typing import Any, Callable, Optional, Sequence

import torch

from .kv_cache import normalize_layer_kv

_SAFE_TORCH_DTYPES = {
    name: getattr(torch, name)
    for name in ("float16", "bfloat16", "float32", "float64", "int8", "uint8", "int16", "int32", "int64", "bool",
                 "float8_e4m3fn", "float8_e5m2")
    if hasattr(torch, name)
}


# ----------------------------------------------------------------- TP topology and routing keys
@dataclass(frozen=True)
class KVTPTopology:
    source_tp_size: int = 1
    target_tp_size: int = 1
    local_rank: int = 0


def validate_kv_tp_topology(topo: KVTPTopology) -> None:
    big, small = max(topo.source_tp_size, topo.target_tp_size), min(topo.source_tp_size, topo.target_tp_size)
    if big % small:
        raise ValueError("KV TP mapping must be divisible: "
                         f"source_tp_size={topo.source_tp_size}, target_tp_size={topo.target_tp_size}")


def get_kv_target_ranks(topo: KVTPTopology) -> list[int]:
    """Remote ranks this local (sending) rank ships shards to."""
    validate_kv_tp_topology(topo)
    s, t, r = topo.source_tp_size, topo.target_tp_size, topo.local_rank
    if s >= t:
        return [r // (s // t)]
    return [r * (t // s) + i for i in range(t // s)]


def get_kv_source_ranks(topo: KVTPTopology) -> list[int]:
    """Remote ranks this local (receiving) rank gets shards from."""
    validate_kv_tp_topology(topo)
    s, t, r = topo.source_tp_size, topo.target_tp_size, topo.local_rank
    if s > t:
        return [r * (s // t) + i for i in range(s // t)]
    return [r // (t // s)]


def get_kv_connector_key(req_id: str, from_stage, chunk_id: int, from_rank: int, to_rank: int) -> str:
    return f"{req_id}_{from_stage}_{chunk_id}_{from_rank}_{to_rank}"


def build_rank_aware_send_keys(request_id: str, from_stage: str, to_stage: str, topo: KVTPTopology,
                               hook: Optional[Callable] = None) -> list[str]:
    if callable(hook):
        keys = list(hook(request_id, from_stage, to_stage))
        if keys:
            return keys
    if topo.source_tp_size <= 1 and topo.target_tp_size <= 1:
        return [f"omni_{from_stage}_to_{to_stage}_kv_cache_{request_id}"]
    return [get_kv_connector_key(request_id, from_stage, 0, topo.local_rank, r) for r in get_kv_target_ranks(topo)]


def build_rank_aware_recv_keys(request_id: str, from_stage: str, to_stage: str, topo: KVTPTopology,
                               hook: Optional[Callable] = None) -> list[tuple[str, Optional[int]]]:
    if callable(hook):
        raw = list(hook(request_id, from_stage, to_stage))
        if raw:
            if isinstance(raw[0], tuple):
                return raw
            src = get_kv_source_ranks(topo)
            return list(zip(raw, src)) if len(raw) == len(src) else [(k, None) for k in raw]
    if topo.source_tp_size <= 1 and topo.target_tp_size <= 1:
        return [(f"omni_{from_stage}_to_{to_stage}_kv_cache_{request_id}", None)]
    return [(get_kv_connector_key(request_id, from_stage, 0, r, topo.local_rank), r) for r in get_kv_source_ranks(topo)]


# ----------------------------------------------------------------- head slicing / merging
def _head_range(heads_in_shard: int, offset_in_shard: int, num_slices: int) -> tuple[int, int]:
    if num_slices <= 0:
        raise ValueError(f"num_slices must be > 0, got {num_slices}")
    if not 0 <= offset_in_shard < num_slices:
        raise ValueError(f"offset_in_shard must be in [0, {num_slices}), got {offset_in_shard}")
    if heads_in_shard % num_slices:
        raise ValueError("KV head count must be divisible for heterogeneous TP slicing: "
                         f"heads_in_shard={heads_in_shard}, num_slices={num_slices}")
    per = heads_in_shard // num_slices
    return offset_in_shard * per, per


def slice_kv_tensor_heads(tensor, offset_in_shard: int, num_slices: int):
    if tensor is None or not isinstance(tensor, torch.Tensor):
        return tensor
    if tensor.dim() < 2:
        raise ValueError(f"Expected KV tensor with a head dimension, got shape={tuple(tensor.shape)}")
    lo, cnt = _head_range(tensor.shape[1], offset_in_shard, num_slices)
    return tensor.narrow(1, lo, cnt).contiguous()


def slice_layer_blocks(layer_blocks: dict[str, Any], offset_in_shard: int, num_slices: int) -> dict[str, list]:
    return {name: [slice_kv_tensor_heads(t, offset_in_shard, num_slices) for t in layer_blocks.get(name, [])]
            for name in ("key_cache", "value_cache")}


def merge_received_rank_shards(payloads: list[dict[str, Any]], merger: Optional[Callable] = None):
    if callable(merger):
        return merger(payloads)
    if not payloads:
        return None
    if len(payloads) == 1:
        return payloads[0]
    first = payloads[0]
    if not isinstance(first, dict) or "layer_blocks" not in first:
        return first
    out = {"request_id": first.get("request_id"), "block_ids": list(first.get("block_ids", [])),
           "metadata": dict(first.get("metadata", {})), "layer_blocks": {}}
    for name in ("key_cache", "value_cache"):
        per_rank = [p.get("layer_blocks", {}).get(name, []) for p in payloads]
        merged = []
        for layer in range(max((len(x) for x in per_rank), default=0)):
            parts = [x[layer] for x in per_rank if layer < len(x) and x[layer] is not None]
            if not parts:
                merged.append(None)
            elif len(parts) == 1 or not isinstance(parts[0], torch.Tensor):
                merged.append(parts[0])
            else:
                merged.append(torch.cat(parts, dim=1).contiguous())
        out["layer_blocks"][name] = merged
    return out


def slice_received_rank_shard(payload, topo: KVTPTopology, slicer: Optional[Callable] = None):
    if callable(slicer):
        return slicer(payload)
    if not payload or topo.target_tp_size <= topo.source_tp_size or "layer_blocks" not in payload:
        return payload
    meta = payload.get("metadata", {})
    tag = meta.get("tp_head_slice") if isinstance(meta, dict) else None
    if isinstance(tag, dict) and tag.get("applied"):
        return payload  # sliced by the sender already
    ratio = topo.target_tp_size // topo.source_tp_size
    off = topo.local_rank % ratio
    new_meta = dict(meta) if isinstance(meta, dict) else {}
    new_meta["tp_head_slice"] = {"applied": True, "side": "receiver", "target_rank": topo.local_rank,
                                 "from_tp": topo.source_tp_size, "to_tp": topo.target_tp_size,
                                 "offset_in_shard": off, "num_slices": ratio}
    return {"request_id": payload.get("request_id"), "layer_blocks": slice_layer_blocks(payload["layer_blocks"], off, ratio),
            "block_ids": list(payload.get("block_ids", [])), "metadata": new_meta}


def sender_slice_metadata(metadata: Optional[dict], topo: KVTPTopology, target_rank: int) -> tuple[dict, int, int]:
    """Metadata tag + (offset_in_shard, num_slices) of ``_slice_transfer_data_for_target``
    (kv_transfer_manager.py:498-519) for one receiving rank when sender TP < receiver TP."""
    ratio = topo.target_tp_size // topo.source_tp_size
    off = target_rank % ratio
    meta = dict(metadata) if isinstance(metadata, dict) else {}
    meta["tp_head_slice"] = {"applied": True, "side": "sender", "target_rank": target_rank, "source_rank": topo.local_rank,
                             "from_tp": topo.source_tp_size, "to_tp": topo.target_tp_size, "offset_in_shard": off,
                             "num_slices": ratio}
    return meta, off, ratio


# ----------------------------------------------------------------- wire format
def _header_bytes(request_id, block_ids, metadata, tensors_desc, num_layers: int, pad_to: int = 1) -> bytes:
    body = json.dumps({"rid": request_id, "bids": block_ids, "meta": metadata, "td": tensors_desc, "nl": num_layers},
                      separators=(",", ":")).encode("utf-8")
    if pad_to > 1:
        body += b" " * (-(4 + len(body)) % pad_to)
    return struct.pack(">I", len(body)) + body


@dataclass
class KVCacheTransferData:
    request_id: str
    layer_blocks: dict[str, Any]
    block_ids: list[int]
    metadata: dict[str, Any]

    def to_dict(self) -> dict[str, Any]:
        return asdict(self)

    def _describe(self, cpu: bool):
        desc, chunks, pos, device = [], [], 0, None
        for name in ("key_cache", "value_cache"):
            for layer, tensor in enumerate(self.layer_blocks.get(name, [])):
                if tensor is None:
                    desc.append({"n": f"{name}_{layer}", "x": True})
                    continue
                t = tensor.detach().contiguous()
                if cpu:
                    t = t.cpu()
                elif device is None and t.is_cuda:
                    device = t.device
                nbytes = t.numel() * t.element_size()
                desc.append({"n": f"{name}_{layer}", "i": layer, "d": str(t.dtype).removeprefix("torch."),
                             "s": list(t.shape), "o": pos, "b": nbytes})
                flat = t.reshape(-1).view(torch.uint8)
                chunks.append(flat.numpy().tobytes() if cpu else flat)
                pos += nbytes
        return desc, chunks, pos, device

    def to_bytes(self) -> bytes:
        desc, chunks, _, _ = self._describe(cpu=True)
        head = _header_bytes(self.request_id, self.block_ids, self.metadata, desc, len(self.layer_blocks.get("key_cache", [])))
        return b"".join([head] + chunks)

    def to_gpu_tensor(self) -> torch.Tensor:
        desc, chunks, total, device = self._describe(cpu=False)
        if device is None:
            raise RuntimeError("No CUDA tensors found, use to_bytes() instead")
        head = _header_bytes(self.request_id, self.block_ids, self.metadata, desc, len(self.layer_blocks.get("key_cache", [])))
        out = torch.empty(len(head) + total, dtype=torch.uint8, device=device)
        out[:len(head)].copy_(torch.frombuffer(bytearray(head), dtype=torch.uint8))
        pos = len(head)
        for c in chunks:
            out[pos:pos + c.numel()].copy_(c)
            pos += c.numel()
        return out

    # ---- loading, with the reference's bounds / dtype validation
    @staticmethod
    def _split_header(buf_len: int, read: Callable[[int, int], bytes]) -> tuple[dict, int]:
        if buf_len < 4:
            raise ValueError("Corrupted KV payload: missing 4-byte header length")
        header_len = struct.unpack(">I", read(0, 4))[0]
        if header_len > buf_len - 4:
            raise ValueError(f"Corrupted KV payload: header_len={header_len} exceeds buffer size={buf_len}")
        return json.loads(read(4, 4 + header_len)), 4 + header_len

    @staticmethod
    def _span(info: dict, data_len: int) -> tuple[int, int]:
        off, nbytes = info["o"], info["b"]
        if off < 0 or nbytes < 0 or off + nbytes > data_len:
            raise ValueError(f"Corrupted KV payload tensor span for {info['n']}: offset={off}, bytes={nbytes}, "
                             f"tensor_data_bytes={data_len}")
        return off, nbytes

    @staticmethod
    def _layer_index(info: dict, num_layers: int) -> int:
        idx = info.get("i")
        if idx is None:
            name = info.get("n")
            for prefix in ("key_cache_", "value_cache_"):
                if isinstance(name, str) and name.startswith(prefix):
                    idx = int(name.removeprefix(prefix))
                    break
            else:
                raise ValueError(f"Invalid KV tensor name in payload: {name}")
        if not isinstance(idx, int):
            raise ValueError(f"Invalid layer index in KV payload: {idx}")
        if not 0 <= idx < num_layers:
            raise ValueError(f"Invalid layer index in KV payload: {idx} (num_layers={num_layers})")
        return idx

    @staticmethod
    def _fill(header: dict, get: Callable[[dict], torch.Tensor]) -> dict[str, Any]:
        nl = header["nl"]
        caches = {"key_cache": [None] * nl, "value_cache": [None] * nl}
        for info in header["td"]:
            if info.get("x"):
                continue
            dt = _SAFE_TORCH_DTYPES.get(str(info["d"]))
            if dt is None:
                raise ValueError(f"Unsupported dtype in KV payload: {info['d']}")
            t = get(info).view(dt).reshape(info["s"])
            idx = KVCacheTransferData._layer_index(info, nl)
            name = info["n"]
            if name.startswith("key_cache_"):
                caches["key_cache"][idx] = t
            elif name.startswith("value_cache_"):
                caches["value_cache"][idx] = t
        return {"request_id": header["rid"], "layer_blocks": caches, "block_ids": header["bids"], "metadata": header["meta"]}

    @staticmethod
    def from_bytes(raw) -> dict[str, Any]:
        mv = raw if isinstance(raw, memoryview) else memoryview(raw)
        header, start = KVCacheTransferData._split_header(len(mv), lambda a, b: bytes(mv[a:b]))
        data = mv[start:]

        def get(info):
            off, nbytes = KVCacheTransferData._span(info, len(data))
            with warnings.catch_warnings():  # `bytes` payloads are read-only buffers; the views are never written
                warnings.simplefilter("ignore", UserWarning)
                return torch.frombuffer(data, dtype=torch.uint8, offset=off, count=nbytes)

        return KVCacheTransferData._fill(header, get)

    @staticmethod
    def from_bytes_gpu(gpu_tensor: torch.Tensor) -> dict[str, Any]:
        if gpu_tensor.dtype != torch.uint8 or gpu_tensor.dim() != 1:
            raise ValueError("Packed GPU KV payload must be a 1-D uint8 tensor")
        total = int(gpu_tensor.numel())
        header, start = KVCacheTransferData._split_header(total, lambda a, b: gpu_tensor[a:b].cpu().numpy().tobytes())

        def get(info):
            off, nbytes = KVCacheTransferData._span(info, total - start)
            return gpu_tensor[start + off:start + off + nbytes].clone()

        return KVCacheTransferData._fill(header, get)


# ----------------------------------------------------------------- fused extract + slice + pack (HIP)
def pack_request_kv(kv_caches: Sequence, request_id: str, block_ids: Sequence[int], seq_len: int, block_size: int,
                    cache_dtype: str, custom_metadata: Optional[dict] = None,
                    head_slice: Optional[tuple[int, int]] = None, metadata_override: Optional[dict] = None
                    ) -> Optional[torch.Tensor]:
    """One request's packed payload (1-D uint8 device tensor) in one kernel launch; ``None`` when no layer yields
    data (as ``_extract_kv_cache`` returns None).  ``head_slice = (offset_in_shard, num_slices)`` applies the
    sender-side slice of ``_slice_transfer_data_for_target``; ``metadata_override`` then carries its tag.
    The GPU library is required: there is no host fallback."""
    from ._C import check, lib

    pairs = [normalize_layer_kv(layer) for layer in kv_caches]
    ref = next((p for p in pairs if p is not None), None)
    if ref is None:
        return None
    kb0 = ref[0]
    if not kb0.is_cuda:
        raise RuntimeError("pack_request_kv needs the paged cache on the GPU (no CPU fallback); "
                           "use extract_request_kv + KVCacheTransferData.to_bytes on the host")
    nl = len(pairs)
    heads, d = kb0.shape[2], kb0.shape[3]
    lo, cnt = (0, heads) if head_slice is None else _head_range(heads, *head_slice)
    elem = kb0.element_size()
    dtype_name = str(kb0.dtype).removeprefix("torch.")
    ids_by_layer, rows_by_layer = [], []
    for p in pairs:
        if p is None:
            ids_by_layer.append(None)
            rows_by_layer.append(0)
            continue
        kb, vb = p
        if kb.stride() != kb0.stride() or vb.stride() != kb0.stride() or kb.dtype != kb0.dtype:
            raise ValueError("pack_request_kv: every layer must share dtype and strides")
        top = min(kb.shape[0], vb.shape[0]) - 1
        valid = [b for b in block_ids if 0 <= b <= top]
        ids_by_layer.append(valid or None)
        rows_by_layer.append(min(seq_len, len(valid) * kb.shape[1]) if valid else 0)
    live = [i for i in range(nl) if ids_by_layer[i]]
    if not live:
        return None
    if any(ids_by_layer[i] != ids_by_layer[live[0]] for i in live):
        raise ValueError("pack_request_kv: layers disagree on the valid block ids")
    valid, rows = ids_by_layer[live[0]], rows_by_layer[live[0]]
    row_bytes = cnt * d * elem
    desc, k_off, v_off, pos = [], [0] * nl, [0] * nl, 0
    for name, offs in (("key_cache", k_off), ("value_cache", v_off)):
        for i in range(nl):
            if not ids_by_layer[i]:
                desc.append({"n": f"{name}_{i}", "x": True})
                continue
            offs[i] = pos
            desc.append({"n": f"{name}_{i}", "i": i, "d": dtype_name, "s": [rows, cnt, d], "o": pos, "b": rows * row_bytes})
            pos += rows * row_bytes
    meta = metadata_override if metadata_override is not None else {
        "block_size": block_size, "num_layers": nl, "dtype": str(cache_dtype), "seq_len": seq_len, **(custom_metadata or {})}
    head = _header_bytes(request_id, list(block_ids), meta, desc, nl, pad_to=16)
    dev = kb0.device
    out = torch.empty(len(head) + pos, dtype=torch.uint8, device=dev)
    out[:len(head)].copy_(torch.frombuffer(bytearray(head), dtype=torch.uint8))
    addr = lambda t: t.data_ptr()  # noqa: E731
    ptrs = torch.tensor([[addr(pairs[i][0]) if ids_by_layer[i] else 0 for i in range(nl)],
                         [addr(pairs[i][1]) if ids_by_layer[i] else 0 for i in range(nl)]], dtype=torch.int64, device=dev)
    offs = torch.tensor([k_off, v_off], dtype=torch.int64, device=dev)
    ids = torch.tensor(valid, dtype=torch.int32, device=dev)
    data = out[len(head):]
    check("mi_kv_pack", lib.mi_kv_pack(
        data.data_ptr(), ptrs[0].data_ptr(), ptrs[1].data_ptr(), offs[0].data_ptr(), offs[1].data_ptr(), nl,
        ids.data_ptr(), len(valid), rows, kb0.shape[1], lo, cnt, d, elem, kb0.stride(0), kb0.stride(1), kb0.stride(2),
        torch.cuda.current_stream(dev).cuda_stream))
    return out
