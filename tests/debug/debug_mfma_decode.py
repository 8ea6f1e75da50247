"""Isolating experiments for the MFMA decode kernel (layout debugging)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd")); sys.path.insert(0, ROOT)
import torch
from vllm_omni_amd import ops
from oracle import ref_ops

dev = torch.device("cuda:0")
torch.manual_seed(0)

def run(q, kb, vb, table, lens, ps, mode):
    S, nq, d = q.shape
    out = torch.full((S, nq * d), float("nan"), dtype=torch.float32, device=dev)
    ops.paged_attn_decode(out, q.view(S, -1).to(dev), kb.to(dev), vb.to(dev), table.to(dev), lens.to(dev),
                          int(lens.max()), d ** -0.5, partition_size=ps, packed_pv=mode != "exact",
                          force_valu=mode == "packed_valu")
    torch.cuda.synchronize()
    return out.cpu().view(S, nq, d)

def case(name, nq, nkv, L, kfn, vfn, qfn, ps=32, bs=16, shuffle=False):
    d = 128
    nblk = (L + bs - 1) // bs
    nb = nblk + 3
    kb = torch.zeros(nb, bs, nkv, d); vb = torch.zeros(nb, bs, nkv, d)
    ids = (torch.randperm(nb - 1)[:nblk] + 1) if shuffle else torch.arange(1, nblk + 1)
    table = ids.view(1, -1).to(torch.int32)
    for t in range(L):
        b, o = int(table[0, t // bs]), t % bs
        for h in range(nkv):
            kb[b, o, h] = kfn(t, h); vb[b, o, h] = vfn(t, h)
    q = torch.stack([qfn(h) for h in range(nq)]).view(1, nq, d)
    kb, vb, q = kb.bfloat16(), vb.bfloat16(), q.bfloat16()
    lens = torch.tensor([L], dtype=torch.int32)
    ref = ref_ops.paged_attention_decode(q, kb, vb, table, lens, d ** -0.5)
    got = run(q, kb, vb, table, lens, ps, "fast")
    chk = run(q, kb, vb, table, lens, ps, "packed_valu")
    e = (got - ref).abs()
    print(f"{name:34s} mfma err max {e.max().item():9.4f} nan={int(torch.isnan(got).sum()):5d}   valu err max {(chk-ref).abs().max().item():.4f}")
    if e.max() > 0.05 or torch.isnan(got).any():
        print("   ref[0,0,:8]", ref[0, 0, :8].tolist()); print("   got[0,0,:8]", got[0, 0, :8].tolist())
        bad = (e > 0.05) | torch.isnan(got)
        print("   bad heads:", bad.any(-1)[0].nonzero().flatten().tolist()[:16], " bad dims(head0):", bad[0, 0].nonzero().flatten().tolist()[:24])
    return got, ref

d = 128
z = lambda *a: torch.zeros(d)
rnd = lambda *a: torch.randn(d)
for L in (32, 16, 7, 64, 100):
    case(f"L={L} K=0 V=1 (normalisation)", 1, 1, L, z, lambda t, h: torch.ones(d), rnd)
    case(f"L={L} K=0 V=dim index", 1, 1, L, z, lambda t, h: torch.arange(d).float() / 16, rnd)
    case(f"L={L} K=0 V=token index", 1, 1, L, z, lambda t, h: torch.full((d,), float(t)), rnd)
    case(f"L={L} K=0 V=random (uniform attn)", 1, 1, L, z, rnd, rnd)
    case(f"L={L} K,q random V=token index", 1, 1, L, rnd, lambda t, h: torch.full((d,), float(t)), rnd)
    case(f"L={L} all random", 1, 1, L, rnd, rnd, rnd)
case("L=64 all random G=7", 7, 1, 64, rnd, rnd, rnd)
case("L=64 all random 28/4", 28, 4, 64, rnd, rnd, rnd)
case("L=300 all random 28/4 ps=128 shuffled", 28, 4, 300, rnd, rnd, rnd, ps=128, shuffle=True)
