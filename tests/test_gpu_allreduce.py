"""GPU: the in-tree two-shot all-reduce (csrc/allreduce.hip, C ABI mi_allreduce_*).

One-GPU box: W processes share cuda:0, exchange their IPC handles over gloo and all-reduce through each other's
IPC-mapped buffers — every protocol step (copy-in, flag exchange, rank-ordered reduce-scatter, all-gather, the
device-side call counter under hipGraph replay, the split-K copy-in) runs for real; what a one-GPU box cannot show
is xGMI bandwidth and cross-device cache behaviour (the design note in the kernel's header covers the latter).
Expected values: the fp32 sum over ranks IN RANK ORDER, rounded once — bit-exact, on every rank.
Reference call sites: vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:54-60,118-124 (RowParallelLinear).
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_input(rank, it, shape, dtype):
    g = torch.Generator().manual_seed(1000 * it + rank)
    return (torch.randn(*shape, generator=g) * (1 + rank)).to(dtype)


def _expected(world, it, shape, dtype):
    acc = torch.zeros(*shape, dtype=torch.float32)
    for r in range(world):                       # rank order, fp32, one rounding
        acc += _rank_input(r, it, shape, dtype).float()
    return acc.to(dtype)


def _worker(rank, world, port, q):
    import sys
    for p in (ROOT, os.path.join(ROOT, "vllm-omni_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        from vllm_omni_amd import distributed as D

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        D.init_distributed_environment(world, rank, world, backend="gloo", init_method=f"tcp://127.0.0.1:{port}")
        assert D.enable_custom_all_reduce(dev, max_bytes=4 << 20, timeout_ms=15000)
        ar = D.get_parallel_state().custom_ar
        # first contact: if the two processes' kernels cannot be resident together the bounded spin ends it here
        x = _rank_input(rank, 0, (8, 64), torch.bfloat16).to(dev)
        ar.all_reduce(x)
        torch.cuda.synchronize()
        ar.check()
        assert torch.equal(x.cpu(), _expected(world, 0, (8, 64), torch.bfloat16)), "first all-reduce"
        it = 1
        for shape, dtype in (((256, 3584), torch.bfloat16),      # the decode step's message: 1.8 MB
                             ((512, 1024), torch.float16),        # S-B
                             ((13, 8), torch.bfloat16),           # 13 vectors: ragged slices, some past the end
                             ((1, 8), torch.bfloat16),            # one vector: every rank but 0 has an empty slice
                             ((64, 3584), torch.bfloat16)):
            for rep in range(6):                                   # back to back, fresh data: staging / tmp reuse
                x = _rank_input(rank, it, shape, dtype).to(dev)
                want = _expected(world, it, shape, dtype)
                if rep % 2:
                    out = torch.full_like(x, float("nan"))
                    ar.all_reduce(x, out)
                    assert torch.equal(x.cpu(), _rank_input(rank, it, shape, dtype)), "input modified"
                else:
                    out = D.tensor_model_parallel_all_reduce(x)    # in place, through the layer-facing entry point
                assert torch.equal(out.cpu(), want), (shape, dtype, rep)
                it += 1
        # ---- inside a hipGraph: three all-reduces per replay, inputs refreshed between replays ----
        shape, dtype = (256, 3584), torch.bfloat16
        bufs = [torch.zeros(*shape, dtype=dtype, device=dev) for _ in range(3)]
        outs = [torch.zeros(*shape, dtype=dtype, device=dev) for _ in range(3)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for b, o in zip(bufs, outs):
                ar.all_reduce(b, o)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for b, o in zip(bufs, outs):
                ar.all_reduce(b, o)
        for replay in range(4):
            for j, b in enumerate(bufs):
                b.copy_(_rank_input(rank, it + j, shape, dtype))
            g.replay()
            torch.cuda.synchronize()
            for j, o in enumerate(outs):
                assert torch.equal(o.cpu(), _expected(world, it + j, shape, dtype)), ("graph", replay, j)
            it += 3
        # ---- split-K partials as the input: copy-in = reduce (+ bias on rank 0 only, as RowParallelLinear adds it) ----
        rows, cols, split = 64, 3584, 4
        parts, vals = [], []
        bias = torch.randn(cols, generator=torch.Generator().manual_seed(77)).to(torch.bfloat16)
        for r in range(world):
            p = torch.randn(split, rows, cols, generator=torch.Generator().manual_seed(500 + r))
            acc = torch.zeros(rows, cols)
            for sidx in range(split):
                acc += p[sidx]
            if r == 0:
                acc += bias.float()
            parts.append(p)
            vals.append(acc.to(torch.bfloat16).float())            # what mi_linear_reduce would have stored
        want = torch.zeros(rows, cols)
        for v in vals:
            want += v
        got = ar.all_reduce_partials(parts[rank].to(dev), split, bias.to(dev) if rank == 0 else None, rows, cols,
                                     torch.bfloat16)
        assert torch.equal(got.cpu(), want.to(torch.bfloat16)), "split-K copy-in"
        # message larger than the registered buffers -> MI_ERR_WORKSPACE, nothing launched
        from vllm_omni_amd import _C

        big = torch.zeros(5 << 20, dtype=torch.bfloat16, device=dev)
        assert not ar.supports(big)
        with pytest.raises(_C.MIError) as ei:
            ar.all_reduce(big)
        assert ei.value.code == _C.MI_ERR_WORKSPACE
        torch.cuda.synchronize()
        ar.check()
        D.destroy_distributed_environment()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()[-1800:]))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_two_shot_all_reduce_bit_exact(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=420))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _late_peer_worker(rank, world, port, q):
    import sys
    for p in (ROOT, os.path.join(ROOT, "vllm-omni_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        import torch.distributed as dist

        from vllm_omni_amd import _C, distributed as D

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        D.init_distributed_environment(world, rank, world, backend="gloo", init_method=f"tcp://127.0.0.1:{port}")
        assert D.enable_custom_all_reduce(dev, max_bytes=1 << 20, timeout_ms=300)
        ar = D.get_parallel_state().custom_ar
        x = _rank_input(rank, 0, (8, 64), torch.bfloat16).to(dev)
        ar.all_reduce(x)
        torch.cuda.synchronize()
        ar.poll()                                    # healthy: no raise, no device round trip
        assert torch.equal(x.cpu(), _expected(world, 0, (8, 64), torch.bfloat16))
        for turn in range(world):                    # each rank in turn launches an all-reduce its peers never join
            dist.barrier()
            if rank == turn:
                out = torch.full((8, 64), 7.0, dtype=torch.bfloat16, device=dev)
                ar.all_reduce(_rank_input(rank, 1, (8, 64), torch.bfloat16).to(dev), out)
                torch.cuda.synchronize()             # returns: the spin is bounded (0.3 s), the kernel skipped its phases
                with pytest.raises(RuntimeError, match="did not arrive"):
                    ar.poll()                        # ... and said so where the host can see it without a copy
                assert ar.dead
                with pytest.raises(_C.MIError) as ei:   # the communicator is dead: no further launch
                    ar.all_reduce(out)
                assert ei.value.code == _C.MI_ERR_STATE
        dist.barrier()
        D.destroy_distributed_environment()          # dead communicators skip the pre-close barrier
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()[-1800:]))


@pytest.mark.timeout(300)
def test_late_peer_is_reported_at_the_step_sync_point_not_silently():
    """A peer that never joins an all-reduce: the bounded spin ends the kernel with its output unwritten; the status
    mirrored into host-mapped memory makes ``poll()`` (what the runner calls after each step's output copy) raise, and
    the communicator refuses further launches (no silent garbage with VLLM_OMNI_AMD_ALLREDUCE=custom)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_late_peer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=200))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert sorted(res) == [(r, "ok") for r in range(world)], res
