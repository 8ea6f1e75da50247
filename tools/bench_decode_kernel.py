"""Micro-benchmark of the paged-attention decode kernel alone (one layer of the workload).

    python tools/bench_decode_kernel.py [--batch 256 --ctx 4096 --nq 28 --nkv 4 --d 128]

Reports achieved algorithmic GB/s = (2 * nkv * d * 2 B * ctx * batch) / kernel time, for
both P.V modes and a sweep of partition sizes.  Timing with HIP events on the launch stream.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))

import torch  # noqa: E402

from vllm_omni_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--nq", type=int, default=28)
    ap.add_argument("--nkv", type=int, default=4)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layout", default="NB2")
    ap.add_argument("--parts", default="0,256,512,1024,2048,4096")
    ap.add_argument("--modes", default="0,1,2", help="0 = exact fp32-P VALU, 1 = packed dot2 VALU, 2 = MFMA, 3 = MFMA + nt (production), "
                    "4 = MFMA + nt with the single-rounding P of round 2 (A/B)")
    ap.add_argument("--two-kernels", action="store_true", help="merge the partitions with decode_reduce_kernel (A/B)")
    ap.add_argument("--ragged", default=None, metavar="LO",
                    help="sequence lengths uniform in [LO * ctx, ctx] (seeded) instead of every sequence at ctx: what a serving "
                         "batch looks like; GB/s then counts the bytes of the actual lengths")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    blocks_per_seq = (a.ctx + a.bs - 1) // a.bs
    nb = a.batch * blocks_per_seq + 1
    if a.layout == "NB2":
        kv = torch.empty(nb, 2, a.bs, a.nkv, a.d, dtype=torch.bfloat16, device=dev).normal_()
        kb, vb = kv[:, 0], kv[:, 1]
    else:
        kv = torch.empty(nb, 2, a.nkv, a.bs, a.d, dtype=torch.bfloat16, device=dev).normal_()
        kb, vb = kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)
    table = (torch.randperm(nb - 1, device=dev)[: a.batch * blocks_per_seq] + 1).view(a.batch, -1).to(torch.int32)
    seq_lens = torch.full((a.batch,), a.ctx, dtype=torch.int32, device=dev)
    if a.ragged is not None:
        lo = max(1, int(float(a.ragged) * a.ctx))
        seq_lens = torch.randint(lo, a.ctx + 1, (a.batch,), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(dev)
        seq_lens[0] = a.ctx
    q = torch.randn(a.batch, a.nq * a.d, dtype=torch.bfloat16, device=dev)
    out = torch.empty_like(q)
    algo_bytes = 2 * a.nkv * a.d * 2 * int(seq_lens.sum())
    print(f"layout={a.layout} batch={a.batch} ctx={a.ctx} nq={a.nq} nkv={a.nkv} d={a.d} "
          f"algorithmic bytes/launch={algo_bytes / 1e9:.3f} GB")
    for mode in [int(x) for x in a.modes.split(",")]:
        packed, force_valu, nt, rp = mode != 0, mode == 1, mode >= 3, mode == 4
        for ps in [int(x) for x in a.parts.split(",")]:
            eff = ops.decode_auto_partition(a.batch, a.nkv, a.ctx, a.bs, ps, packed_pv=packed, force_valu=force_valu, head_dim=a.d)
            nbytes = max(1, ops.decode_workspace_bytes(a.batch, a.nq, a.d, a.ctx, eff))
            # a prepared workspace = partitions merged inside the attention launch; --two-kernels: the separate merge
            ws = (torch.empty(nbytes, dtype=torch.uint8, device=dev) if a.two_kernels
                  else ops.new_decode_workspace(nbytes, dev))
            for _ in range(10):
                ops.paged_attn_decode(out, q, kb, vb, table, seq_lens, a.ctx, a.d ** -0.5, partition_size=eff,
                                      workspace=ws, packed_pv=packed, force_valu=force_valu, nontemporal=nt, rounded_p=rp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.paged_attn_decode(out, q, kb, vb, table, seq_lens, a.ctx, a.d ** -0.5, partition_size=eff,
                                      workspace=ws, packed_pv=packed, force_valu=force_valu, nontemporal=nt, rounded_p=rp)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"mode={['exact_valu', 'packed_valu', 'mfma', 'mfma_nt', 'mfma_nt_rndP'][mode]:12s} partition={eff:5d}{'(auto)' if ps == 0 else '      '} "
                  f"{ms * 1e3:9.1f} us  {algo_bytes / ms / 1e6:8.1f} GB/s  ({algo_bytes / ms / 1e6 / 8000:.1%} of 8 TB/s)")


if __name__ == "__main__":
    main()
