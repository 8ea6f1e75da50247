"""Per-phase timing of the prefill kernel from the timing build (`make -C vllm-omni_amd/csrc dbg`).

The kernel accumulates s_memrealtime deltas (100 MHz ticks) per wave of workgroup (0,0,0) — the LONGEST
tile (tiles are dispatched longest first): vmcnt wait (my DMA share of the stage landed), barrier, DMA
issue, compute (K reads, QK^T, V reads, softmax, PV).  Reported as mean ns per 32-key step.
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("MI_TIMING_LIB") or os.path.join(ROOT, "vllm-omni_amd", "build", "libvllm_omni_amd_timing.so"))
lib.mi_paged_attn_prefill.restype = ctypes.c_int
lib.mi_paged_attn_prefill.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int32] * 8 + [ctypes.c_float] + [ctypes.c_int64] * 5 + [
    ctypes.c_int, ctypes.c_void_p]


def main():
    prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    nq, nkv, d, bs = 28, 4, 128, 16
    dev = torch.device("cuda:0")
    nblk = prompt // bs
    kv = torch.empty(nblk + 1, 2, nkv, bs, d, dtype=torch.bfloat16, device=dev).normal_()
    kb, vb = kv[:, 0].permute(0, 2, 1, 3), kv[:, 1].permute(0, 2, 1, 3)
    table = (torch.randperm(nblk, device=dev) + 1).view(1, -1).to(torch.int32)
    q = torch.randn(prompt, nq * d, dtype=torch.bfloat16, device=dev)
    out = torch.empty_like(q)
    sl = torch.tensor([prompt], dtype=torch.int32, device=dev)
    qsl = torch.tensor([0, prompt], dtype=torch.int32, device=dev)

    def run():
        rc = lib.mi_paged_attn_prefill(out.data_ptr(), q.data_ptr(), kb.data_ptr(), vb.data_ptr(), table.data_ptr(),
                                       sl.data_ptr(), qsl.data_ptr(), 1, prompt, prompt, nq, nkv, d, bs, table.stride(0),
                                       d ** -0.5, q.stride(0), out.stride(0), kb.stride(0), kb.stride(1), kb.stride(2), 1,
                                       None)
        assert rc == 0, rc
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 256)()
    assert lib.mi_debug_prefill_ticks(buf) == 0
    print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us")
    names = ["vmcnt", "barrier", "dma issue", "Kissue+pre", "Kwait+Vissue", "QK+exp", "Vwait", "PV"]
    for w in range(8):
        row = [buf[w * 16 + i] for i in range(16)]
        steps = row[10]
        if not steps:
            continue
        tot = sum(row[:8])
        print(f"wave {w}: steps={steps} total={tot * 10 / 1e3:.1f} us | " +
              " ".join(f"{n}={row[i] * 10 / steps:.0f}ns" for i, n in enumerate(names)) +
              (f" | shader clock {row[8] / (row[9] * 10e-9) / 1e9:.2f} GHz" if row[9] else ""))


if __name__ == "__main__":
    main()
