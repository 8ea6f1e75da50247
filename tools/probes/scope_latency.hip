// Probe (not shipped): round-trip latency of one wave's dependent memory operations at the scopes the one-launch code-predictor
// stack uses — what a phase's "slabs in, rows out" chain is made of.
//   mode 0: dependent plain loads over a small buffer (L2 / L1 hits)      mode 1: the same with sc1 (device scope: past the L2)
//   mode 2: dependent plain loads over a 256 MB buffer (HBM / Infinity Cache)
//   mode 3: store + s_waitcnt vmcnt(0), plain                              mode 4: the same with sc1 (write-through, acknowledged)
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void lat_kernel(uint32_t* buf, int n_steps, int mode, unsigned long long* out) {
  if (threadIdx.x != 0) return;
  uint32_t idx = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (mode == 0 || mode == 2) {
    for (int i = 0; i < n_steps; ++i) {
      uint32_t v;
      asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(buf + idx) : "memory");
      idx = v;
    }
  } else if (mode == 1) {
    for (int i = 0; i < n_steps; ++i) {
      uint32_t v;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(buf + idx) : "memory");
      idx = v;
    }
  } else if (mode == 3) {
    for (int i = 0; i < n_steps; ++i) {
      asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(buf + ((i * 32) & 0xffff)), "v"(idx) : "memory");
    }
  } else {
    for (int i = 0; i < n_steps; ++i) {
      asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(buf + ((i * 32) & 0xffff)), "v"(idx) : "memory");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  out[0] = t1 - t0;
  out[1] = idx;
}

extern "C" int lat_launch(uint32_t* buf, int n_steps, int mode, unsigned long long* out, void* stream) {
  hipLaunchKernelGGL(lat_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, n_steps, mode, out);
  return (int)hipGetLastError();
}
