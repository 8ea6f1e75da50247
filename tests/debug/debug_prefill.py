import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd")); sys.path.insert(0, ROOT)
import torch
from vllm_omni_amd import ops
from oracle import ref_ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(d, nq, nkv, L, kmode, vmode, bs=16):
    nblk = (L + bs - 1) // bs; nb = nblk + 2
    kv = torch.zeros(nb, 2, bs, nkv, d)
    table = torch.arange(1, nblk + 1).view(1, -1).to(torch.int32)
    for t in range(L):
        b, o = int(table[0, t // bs]), t % bs
        kv[b, 0, o] = torch.randn(nkv, d) if kmode == "rand" else 0
        kv[b, 1, o] = {"one": torch.ones(nkv, d), "dim": torch.arange(d).float().repeat(nkv, 1) / 16,
                       "tok": torch.full((nkv, d), float(t)), "rand": torch.randn(nkv, d)}[vmode]
    kv = kv.bfloat16(); q = torch.randn(L, nq, d).bfloat16()
    sl = torch.tensor([L], dtype=torch.int32); qsl = torch.tensor([0, L], dtype=torch.int32)
    ref = ref_ops.paged_attention_prefill(q, kv[:, 0], kv[:, 1], table, sl, qsl, d ** -0.5)
    kvg = kv.to(dev); out = torch.full((L, nq * d), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.paged_attn_prefill(out, q.view(L, -1).to(dev), kvg[:, 0], kvg[:, 1], table.to(dev), sl.to(dev), qsl.to(dev), L, d ** -0.5)
    torch.cuda.synchronize()
    got = out.cpu().float().view(L, nq, d)
    e = (got - ref).abs()
    nan = torch.isnan(got)
    print(f"d={d} {nq}/{nkv} L={L} K={kmode} V={vmode}: err {e[~nan].max().item() if (~nan).any() else -1:.4f} nan={int(nan.sum())} "
          f"nan rows={nan.any(-1).any(-1).nonzero().flatten().tolist()[:20]} nan dims(row0,h0)={nan[0,0].nonzero().flatten().tolist()[:20]}")
    if (e[~nan] > 0.05).any():
        bad = (e > 0.05)
        print("   bad rows", bad.any(-1).any(-1).nonzero().flatten().tolist()[:20], "bad dims (first bad row)", bad[bad.any(-1).any(-1).nonzero()[0,0], 0].nonzero().flatten().tolist()[:32])
for d in (128, 64):
    for L in (16, 19, 40):
        run(d, 1, 1, L, "zero", "one"); run(d, 1, 1, L, "zero", "dim"); run(d, 1, 1, L, "zero", "tok"); run(d, 1, 1, L, "rand", "rand")
run(64, 16, 2, 19, "rand", "rand")

print("---- detail d=64 L=16")
d, L, bs = 64, 16, 16
kv = torch.zeros(3, 2, bs, 1, d)
for t in range(L):
    kv[1, 1, t, 0] = torch.arange(d).float() + 100 * t      # V[t][dim] = 100 t + dim
kv = kv.bfloat16(); q = torch.randn(L, 1, d).bfloat16()
table = torch.tensor([[1]], dtype=torch.int32); sl = torch.tensor([L], dtype=torch.int32); qsl = torch.tensor([0, L], dtype=torch.int32)
kvg = kv.to(dev); out = torch.full((L, d), float("nan"), dtype=torch.float32, device=dev).bfloat16()
ops.paged_attn_prefill(out, q.view(L, -1).to(dev), kvg[:, 0], kvg[:, 1], table.to(dev), sl.to(dev), qsl.to(dev), L, d ** -0.5)
torch.cuda.synchronize()
g = out.cpu().float()
print("row0 (expect dims 0..63):", g[0].tolist())
print("row1 (expect 50+dim):", g[1, :16].tolist())
