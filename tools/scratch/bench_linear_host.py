"""Host-side cost of one F.linear call for prefill-sized M (a shape TunableOp has no entry for).

    python tools/scratch/bench_linear_host.py

Times the ENQUEUE of 50 calls (no sync inside the loop; kernels are ~15 us so the queue never fills)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def host_us(x, w, n=50):
    for _ in range(5):
        F.linear(x, w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        F.linear(x, w)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def main():
    dev = torch.device("cuda:0")
    shapes = [(2048, 1024, 4096), (2048, 4096, 1024), (1999, 1024, 4096), (2048, 3584, 4608), (2048, 18944, 3584)]
    import torch.cuda.tunable as tn
    for mode in ("default", "tunable_on_untuned", "rocblas", "hipblaslt"):
        if mode == "tunable_on_untuned":
            from vllm_omni_amd.tuned import enable_tuned_gemms
            print("tuned file loaded:", enable_tuned_gemms())
        elif mode == "rocblas":
            tn.enable(False)
            torch.backends.cuda.preferred_blas_library("cublas")
        elif mode == "hipblaslt":
            tn.enable(False)
            torch.backends.cuda.preferred_blas_library("cublaslt")
        for (m, k, n) in shapes:
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
            h, tot = host_us(x, w)
            print(f"{mode:20s} M={m} K={k} N={n}: enqueue {h:7.1f} us/call, with drain {tot:7.1f} us/call", flush=True)


if __name__ == "__main__":
    main()
