"""The code predictor's one-launch layer stack (csrc/cp_stack.hip, ``ops.CpStack``) against the chain of stand-alone
launches it replaces, run with the same tile plans: bit-identical hidden states and KV caches (the phases are the
stand-alone kernels' arithmetic behind a grid barrier instead of a kernel boundary), over consecutive passes (the
barrier's generation counter carries from launch to launch), both workgroup shapes (<= 64 rows: 32-row weight tiles;
more: 128-row tiles, several row blocks) and ragged row counts.  Reference of the stack itself:
vllm_omni/model_executor/models/common/qwen3_code_predictor.py:143-191 (layer), 318-420 (the per-step pass);
tests/test_gpu_code_predictor.py holds the comparison of the whole predictor with the fp32 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

H, NQ, NKV, D, INTER, BS = 1024, 16, 8, 128, 3072, 64
EPS = 1e-6


@pytest.fixture
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _weights(dev, layers, dtype, seed):
    g = torch.Generator(device=dev).manual_seed(seed)

    def w(n, k, std=None):
        return (torch.randn(n, k, device=dev, generator=g) * (std or k ** -0.5)).to(dtype)

    def nw(n):
        return (1.0 + 0.1 * torch.randn(n, device=dev, generator=g)).to(dtype)

    return [dict(qkv=w((NQ + 2 * NKV) * D, H), o=w(H, NQ * D), gate_up=w(2 * INTER, H), down=w(H, INTER),
                 input_norm=nw(H), post_norm=nw(H), q_norm=nw(D), k_norm=nw(D)) for _ in range(layers)], nw(H)


def _caches(dev, layers, rows, dtype, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    kv = [(torch.randn(rows + 1, 2, NKV, BS, D, device=dev, generator=g) * 0.5).to(dtype) for _ in range(layers)]
    return kv, [(c[:, 0].permute(0, 2, 1, 3), c[:, 1].permute(0, 2, 1, 3)) for c in kv]


def _cos_sin(dev, max_pos=64):
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float32, device=dev) / D))
    f = torch.arange(max_pos, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
    return torch.cat([f.cos(), f.sin()], dim=-1).contiguous()


def _chain(x, layers, final_norm, views, block_ids, cos_sin, pos, plan):
    """The same pass as separate launches, tile plans pinned to the stack's."""
    from vllm_omni_amd import _C, ops
    from vllm_omni_amd.tuned import linear_plans

    rows = x.shape[0]
    wrows, s1, s2, s4 = plan
    lib = _C.lib
    linear_plans.reset()
    qkv_n = (NQ + 2 * NKV) * D
    for n, k, sp in ((qkv_n, H, s1), (H, NQ * D, s2), (H, INTER, s4)):
        purpose = _C.MI_LINEAR_PLAN_PARTIALS if sp > 1 else _C.MI_LINEAR_PLAN_FULL   # one slice: the 16-bit output itself
        assert lib.mi_linear_plan_set(purpose, 0, 4096, n, k, 64, wrows, sp) == _C.MI_OK
    assert lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_GATED, 0, 4096, INTER, H, 64, wrows, 1) == _C.MI_OK
    try:
        scale = D ** -0.5
        residual = x.clone()
        xn = ops.rms_norm(x, layers[0]["input_norm"], EPS)
        down = None
        for li, layer in enumerate(layers):
            kc, vc = views[li]
            if li > 0:
                if isinstance(down, ops.SplitKPartials):
                    xn = ops.fused_add_rms_norm_splitk(down, residual, layer["input_norm"], EPS)
                else:
                    xn = down
                    ops.fused_add_rms_norm(xn, residual, layer["input_norm"], EPS)
            part = ops.linear_partials(xn, layer["qkv"]) if s1 > 1 else None
            if part is not None:
                attn = ops.rope_cache_attend_block_splitk(part, NQ, NKV, D, pos, cos_sin, kc, vc, block_ids, scale,
                                                          q_norm_weight=layer["q_norm"], k_norm_weight=layer["k_norm"],
                                                          norm_eps=EPS)
            else:
                qkv = ops.linear(xn, layer["qkv"])
                q, k, v = qkv.split([NQ * D, NKV * D, NKV * D], dim=-1)
                attn = ops.rope_cache_attend_block(torch.empty(rows, NQ * D, dtype=x.dtype, device=x.device), q, k, v, pos,
                                                   cos_sin, D, kc, vc, block_ids, scale, q_norm_weight=layer["q_norm"],
                                                   k_norm_weight=layer["k_norm"], norm_eps=EPS)
            o = ops.linear_partials(attn, layer["o"]) if s2 > 1 else None
            if o is not None:
                xn = ops.fused_add_rms_norm_splitk(o, residual, layer["post_norm"], EPS)
            else:
                xn = ops.linear(attn, layer["o"])
                ops.fused_add_rms_norm(xn, residual, layer["post_norm"], EPS)
            act = ops.linear_silu_mul(xn, layer["gate_up"])
            down = ops.linear_partials(act, layer["down"]) if s4 > 1 else None
            if down is None:
                down = ops.linear(act, layer["down"])
        if isinstance(down, ops.SplitKPartials):
            return ops.fused_add_rms_norm_splitk(down, residual, final_norm, EPS)
        ops.fused_add_rms_norm(down, residual, final_norm, EPS)
        return down
    finally:
        linear_plans.reset()


def _plan(rows, cus):
    wrows = 32 if rows <= 64 else 128
    rb = (rows + 63) // 64

    def split(n, k):
        tiles = -(-n // wrows)
        return min(max(cus // (tiles * rb), 1), 8, k // 128)

    return wrows, split((NQ + 2 * NKV) * D, H), split(H, NQ * D), split(H, INTER)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,layers,pos", [(32, 2, 2), (7, 2, 17), (64, 5, 30), (128, 2, 5), (200, 2, 31), (512, 1, 16)])
def test_stack_matches_the_chain_of_launches(dev, rows, layers, pos, dtype):
    from vllm_omni_amd import ops

    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    lw, final_norm = _weights(dev, layers, dtype, 3)
    kv_a, views_a = _caches(dev, layers, rows, dtype, 5)
    kv_b = [c.clone() for c in kv_a]
    views_b = [(c[:, 0].permute(0, 2, 1, 3), c[:, 1].permute(0, 2, 1, 3)) for c in kv_b]
    cos_sin = _cos_sin(dev)
    block_ids = torch.arange(1, rows + 1, dtype=torch.int32, device=dev)
    stack = ops.CpStack([dict(layer, k_cache=views_a[i][0], v_cache=views_a[i][1]) for i, layer in enumerate(lw)], final_norm,
                        cos_sin, NQ, NKV, D, EPS, D ** -0.5, max_rows=rows)
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(rows, H, device=dev, generator=g).to(dtype)
    xa, xb = x.clone(), x.clone()
    for step in range(3 if pos + 2 <= 32 else 1):          # consecutive passes: the output is the next input
        out = stack.decode(torch.empty_like(xa), xa, block_ids, pos + step)
        ref = _chain(xb, lw, final_norm, views_b, block_ids, cos_sin, pos + step, _plan(rows, cus))
        assert stack.status() == 0
        assert torch.isfinite(out.float()).all()
        assert torch.equal(out, ref), (step, (out.float() - ref.float()).abs().max().item())
        for a, b in zip(kv_a, kv_b):
            assert torch.equal(a, b)
        xa, xb = out, ref


def test_stack_refuses_what_it_was_not_built_for(dev):
    from vllm_omni_amd import _C, ops

    lw, final_norm = _weights(dev, 1, torch.bfloat16, 1)
    _, views = _caches(dev, 1, 4, torch.bfloat16, 2)
    stack = ops.CpStack([dict(lw[0], k_cache=views[0][0], v_cache=views[0][1])], final_norm, _cos_sin(dev), NQ, NKV, D, EPS,
                        D ** -0.5, max_rows=4)
    x = torch.zeros(4, H, dtype=torch.bfloat16, device=dev)
    ids = torch.arange(1, 5, dtype=torch.int32, device=dev)
    with pytest.raises(_C.MIError) as ei:
        stack.decode(torch.empty_like(x), x, ids, 40)          # beyond the two 16-token tiles the attention phase holds
    assert ei.value.code == _C.MI_ERR_UNSUPPORTED
    with pytest.raises(ValueError):
        stack.decode(torch.empty(5, H, dtype=torch.bfloat16, device=dev), torch.zeros(5, H, dtype=torch.bfloat16, device=dev),
                     torch.arange(1, 6, dtype=torch.int32, device=dev), 3)
