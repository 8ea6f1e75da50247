"""GPU parity: fused KV extract + head slice + pack (mi_kv_pack) vs the oracle — bit-exact bytes."""
import json
import os
import struct

import numpy as np
import pytest
import torch

from oracle import ref_kv_transfer as ref

pytestmark = pytest.mark.gpu


def _views(layout, nb, bs, h, d, dtype, dev, g):
    if layout == "nb2":
        kv = torch.randn(nb, 2, bs, h, d, generator=g).to(dtype).to(dev)
        return kv, kv[:, 0], kv[:, 1]
    if layout == "2nb":
        kv = torch.randn(2, nb, bs, h, d, generator=g).to(dtype).to(dev)
        return kv, kv[0], kv[1]
    kv = torch.randn(nb, 2, h, bs, d, generator=g).to(dtype).to(dev)       # head-major blocks
    return kv, kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)


def _np(t):
    return t.cpu().view(torch.uint16).numpy() if t.dtype == torch.bfloat16 else t.cpu().numpy()


@pytest.mark.parametrize("layout", ["nb2", "2nb", "head_major"])
@pytest.mark.parametrize("dtype,case", [
    (torch.float32, (2, 4, 16, 8, [1, 3, 5], 20, None)),          # the reference's test constants
    (torch.bfloat16, (3, 4, 128, 16, [7, 2, 9, 4], 50, None)),    # S-A head geometry, ragged tail
    (torch.bfloat16, (3, 4, 128, 16, [7, 2, 9, 4], 64, (1, 2))),  # sender-side slice: heads 2..3
    (torch.float16, (2, 2, 64, 16, [3, 99, -1, 5], 1000, (0, 2))),  # invalid ids dropped, seq_len longer than the rows
])
def test_pack_matches_oracle_bytes(dev, layout, dtype, case):
    _check_pack(dev, layout, dtype, case)


@pytest.mark.parametrize("seed", list(range(6)) + list(range(100, 100 + int(os.environ.get("MI_FUZZ_EXTRA", "0")))))
def test_pack_random_sweep(dev, seed):
    """Seeded sweep (more seeds with MI_FUZZ_EXTRA on a GPU box): layer counts, head geometries, block sizes, ragged and
    over-long seq_len, invalid / repeated block ids, sender-side head slices, every cache layout — bit-exact bytes."""
    rng = np.random.default_rng(6000 + seed)
    layout = str(rng.choice(["nb2", "2nb", "head_major"]))
    dtype = [torch.bfloat16, torch.float16, torch.float32][int(rng.integers(0, 3))]
    nl, h = int(rng.integers(1, 5)), int(rng.choice([1, 2, 4, 8]))
    d, bs = int(rng.choice([32, 64, 128])), int(rng.choice([4, 8, 16, 32]))
    nids = int(rng.integers(1, 9))
    ids = [int(x) for x in rng.integers(0, 12, size=nids)]
    if rng.random() < 0.3:
        ids[int(rng.integers(0, nids))] = int(rng.choice([-1, 12, 99]))          # invalid: dropped
    seq_len = int(rng.choice([1, bs - 1, bs, bs + 1, nids * bs - 1, nids * bs, nids * bs + 7, 1000]))
    hs = None
    if h > 1 and rng.random() < 0.5:                                              # (offset in the shard, number of slices)
        ns = int(rng.choice([x for x in (2, 4, 8) if h % x == 0]))
        hs = (int(rng.integers(0, ns)), ns)
    if not [b for b in ids if 0 <= b < 12]:
        ids.append(int(rng.integers(0, 12)))                                      # at least one valid block
    _check_pack(dev, layout, dtype, (nl, h, d, bs, ids, max(seq_len, 1), hs))


def _check_pack(dev, layout, dtype, case):
    from vllm_omni_amd import kv_transfer as kt

    nl, h, d, bs, ids, seq_len, hs = case
    nb = 12
    g = torch.Generator().manual_seed(nl * 131 + h)
    layers = [_views(layout, nb, bs, h, d, dtype, dev, g) for _ in range(nl)]
    caches = [(kb, vb) for _, kb, vb in layers]
    out = kt.pack_request_kv(caches, "req_test_1", ids, seq_len, bs, str(dtype).removeprefix("torch."), head_slice=hs)
    torch.cuda.synchronize()
    raw = out.cpu().numpy().tobytes()
    hl = struct.unpack(">I", raw[:4])[0]
    assert (4 + hl) % 16 == 0
    header = json.loads(raw[4:4 + hl])
    ks, vs = [], []
    for _, kb, vb in layers:
        k, v = ref.extract(_np(kb), _np(vb), ids, seq_len)
        if hs is not None:
            k, v = ref.slice_heads(k, *hs), ref.slice_heads(v, *hs)
        ks.append(k)
        vs.append(v)
    name = str(dtype).removeprefix("torch.")
    meta = {"block_size": bs, "num_layers": nl, "dtype": name, "seq_len": seq_len}
    want = ref.to_bytes("req_test_1", ks, vs, ids, meta, dtype_names={i: name for i in range(nl)})
    whl = struct.unpack(">I", want[:4])[0]
    assert header == json.loads(want[4:4 + whl])                 # same header fields, shapes, offsets
    assert raw[4 + hl:] == want[4 + whl:]                        # data region: bit-exact
    back = kt.KVCacheTransferData.from_bytes_gpu(out)
    assert torch.equal(back["layer_blocks"]["key_cache"][0].cpu(),
                       kt.KVCacheTransferData.from_bytes(want)["layer_blocks"]["key_cache"][0])


def test_pack_large_request_roundtrip_property(dev):
    """Full-size shape (28 layers, 4096 tokens, 4 x 128 heads, head-major blocks): pack -> from_bytes_gpu gives
    back exactly the rows the block table names (checksum of checksums against a torch gather)."""
    from vllm_omni_amd import kv_transfer as kt

    nl, h, d, bs, seq_len = 28, 4, 128, 16, 4096
    nb = seq_len // bs + 9
    g = torch.Generator(device=dev).manual_seed(5)
    kv = [torch.randn(nb, 2, h, bs, d, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(nl)]
    caches = [(x[:, 0].permute(0, 2, 1, 3), x[:, 1].permute(0, 2, 1, 3)) for x in kv]
    ids = (torch.randperm(nb - 1)[: seq_len // bs] + 1).tolist()
    out = kt.pack_request_kv(caches, "big", ids, seq_len - 5, bs, "bfloat16")
    back = kt.KVCacheTransferData.from_bytes_gpu(out)
    idx = torch.tensor(ids, device=dev)
    for i in (0, 13, 27):
        want_k = caches[i][0].index_select(0, idx).flatten(0, 1)[: seq_len - 5]
        want_v = caches[i][1].index_select(0, idx).flatten(0, 1)[: seq_len - 5]
        assert torch.equal(back["layer_blocks"]["key_cache"][i], want_k)
        assert torch.equal(back["layer_blocks"]["value_cache"][i], want_v)
    assert back["metadata"]["seq_len"] == seq_len - 5 and back["block_ids"] == ids


def test_pack_edge_cases(dev):
    from vllm_omni_amd import kv_transfer as kt

    kv = torch.randn(6, 2, 8, 2, 16, device=dev)
    caches = [(kv[:, 0], kv[:, 1]), "not-a-kv-layer"]
    assert kt.pack_request_kv(caches, "r", [77, -3], 4, 8, "float32") is None       # no valid block anywhere
    out = kt.pack_request_kv(caches, "r", [2], 3, 8, "float32")                      # bad layer -> {"x": true}
    back = kt.KVCacheTransferData.from_bytes_gpu(out)
    assert back["layer_blocks"]["key_cache"][1] is None and back["layer_blocks"]["key_cache"][0].shape == (3, 2, 16)
    assert torch.equal(back["layer_blocks"]["value_cache"][0], kv[2, 1, :3])
    with pytest.raises(ValueError, match="divisible"):
        kt.pack_request_kv(caches[:1], "r", [2], 3, 8, "float32", head_slice=(0, 3))
