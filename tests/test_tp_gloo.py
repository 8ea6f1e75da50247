"""CPU, world_size 2, gloo: the N>1 path — group creation (dp x tp layout), head/row/column sharding
and the all-reduce / all-gather composition must reproduce the unsharded layer.  The per-rank
arithmetic here is the ORACLE's (this is a test; the product has no CPU compute path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_ops


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, tp, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "vllm-omni_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from vllm_omni_amd import distributed as D
    from vllm_omni_amd.layers import shard_column, shard_merged_column, shard_qkv, shard_row

    try:
        st = D.init_distributed_environment(world, rank, tp, backend="gloo",
                                            init_method=f"tcp://127.0.0.1:{port}")
        assert (st.tp_rank, st.dp_rank, st.dp_size) == (rank % tp, rank // tp, world // tp)
        torch.manual_seed(0)  # identical full weights on every rank
        H, nq, nkv, d, I, T, V = 64, 4, 2, 16, 96, 5, 32
        x = torch.randn(T, H)
        wqkv, bqkv = torch.randn((nq + 2 * nkv) * d, H) * 0.1, torch.randn((nq + 2 * nkv) * d) * 0.1
        wo, wgu, wd, wlm = torch.randn(H, nq * d) * 0.1, torch.randn(2 * I, H) * 0.1, torch.randn(H, I) * 0.1, torch.randn(V, H) * 0.1
        # ---- unsharded reference (attention replaced by identity on q: sharding is what is tested)
        qkv = x @ wqkv.t() + bqkv
        q_full = qkv[:, : nq * d]
        o_full = q_full @ wo.t()
        mlp_full = ref_ops.silu_and_mul(o_full @ wgu.t()) @ wd.t()
        logits_full = mlp_full @ wlm.t()
        # ---- this rank's shards
        r, n = st.tp_rank, st.tp_size
        nq_l, nkv_l, _ = D.partition_heads(nq, nkv, n)
        qkv_l = x @ shard_qkv(wqkv, d, nq, nkv, n, r).t() + shard_qkv(bqkv, d, nq, nkv, n, r)
        q_l = qkv_l[:, : nq_l * d]
        assert torch.allclose(q_l, q_full[:, r * nq_l * d:(r + 1) * nq_l * d], atol=1e-5)
        k_l = qkv_l[:, nq_l * d:(nq_l + nkv_l) * d]
        kv0 = list(D.kv_head_for_rank(nkv, n, r))[0]
        assert torch.allclose(k_l[:, :d], qkv[:, (nq + kv0) * d:(nq + kv0 + 1) * d], atol=1e-5)
        o_l = D.tensor_model_parallel_all_reduce(q_l @ shard_row(wo, n, r).t())
        assert torch.allclose(o_l, o_full, atol=1e-4)
        gu_l = o_l @ shard_merged_column(wgu, [I, I], n, r).t()
        mlp_l = D.tensor_model_parallel_all_reduce(ref_ops.silu_and_mul(gu_l) @ shard_row(wd, n, r).t())
        assert torch.allclose(mlp_l, mlp_full, atol=1e-4)
        logits = D.tensor_model_parallel_all_gather(mlp_l @ shard_column(wlm, n, r).t())
        assert logits.shape == (T, V) and torch.allclose(logits, logits_full, atol=1e-4)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"FAIL: {type(e).__name__}: {e}"))
    finally:
        D.destroy_distributed_environment()


@pytest.mark.parametrize("world,tp", [(2, 2), (2, 1), (4, 2)])
@pytest.mark.timeout(120)
def test_tp_sharding_composes_over_gloo(world, tp):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, tp, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
