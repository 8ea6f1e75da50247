// Probe (not shipped): how long a consumer workgroup waits for rows another workgroup has JUST written — the hand-off every
// phase boundary of the one-launch stack makes.  Workgroup 0 writes `bytes` (16 B per lane, write-through sc1),
// waits for the acknowledgements, raises a flag; workgroup `cons` (1: another XCD under round-robin dispatch, 8: the same
// XCD) polls the flag, then reads the bytes with device-scope loads.  Reported: ticks (10 ns) flag-seen -> data in
// registers, flag raise -> flag seen, and the producer's store + acknowledge time.  rounds repeat with fresh values.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(256) void handoff_kernel(u32x4* data, unsigned* flag, unsigned long long* t_raise, int vec_per_thread,
                                                      int cons, int rounds, unsigned long long* out) {
  const int b = blockIdx.x;
  if (b != 0 && b != cons) return;
  unsigned long long acc_load = 0, acc_flag = 0, acc_store = 0;
  unsigned sink = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (b == 0) {
      // wait until the consumer has finished the previous round (flag[1] = r - 1)
      if (threadIdx.x == 0) while (__hip_atomic_load(flag + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r - 1)) {}
      __syncthreads();
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int i = 0; i < vec_per_thread; ++i) {
        const u32x4 v = {(unsigned)r, threadIdx.x, (unsigned)i, 7u};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(data + i * 256 + threadIdx.x), "v"(v) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
      if (threadIdx.x == 0) {
        t_raise[0] = t1;
        __hip_atomic_store(flag, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc_store += t1 - t0;
      }
    } else {
      if (threadIdx.x == 0) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {}
      __syncthreads();
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      // loads and their wait in ONE asm statement: the compiler must not touch a destination register before the data is in it
      // (a separate wait statement lets it schedule the compare — or reuse the register — in between)
      for (int i0 = 0; i0 < vec_per_thread; i0 += 4) {
        const u32x4* src = data + i0 * 256 + threadIdx.x;
        if (i0 + 4 <= vec_per_thread) {
          u32x4 a, b, c, d;
          asm volatile(
              "global_load_dwordx4 %0, %4, off sc1\n\t"
              "global_load_dwordx4 %1, %5, off sc1\n\t"
              "global_load_dwordx4 %2, %6, off sc1\n\t"
              "global_load_dwordx4 %3, %7, off sc1\n\t"
              "s_waitcnt vmcnt(0)"
              : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
              : "v"(src), "v"(src + 256), "v"(src + 512), "v"(src + 768)
              : "memory");
          sink += (a[0] != (unsigned)r) + (b[0] != (unsigned)r) + (c[0] != (unsigned)r) + (d[0] != (unsigned)r);
        } else {
          for (int j = i0; j < vec_per_thread; ++j) {
            u32x4 a;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(data + j * 256 + threadIdx.x) : "memory");
            sink += a[0] != (unsigned)r;
          }
        }
      }
      __syncthreads();
      const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
      if (threadIdx.x == 0) {
        acc_load += t1 - t0;
        acc_flag += t0 - __hip_atomic_load(t_raise, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag + 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (threadIdx.x == 0) {
    if (b == 0) out[2] = acc_store;
    else { out[0] = acc_load; out[1] = acc_flag; }
  }
  if (b != 0) {
    // stale reads over all threads
    atomicAdd((unsigned*)(out + 3), sink);
  }
}

extern "C" int handoff_launch(void* data, unsigned* flag, unsigned long long* t_raise, int vec_per_thread, int cons, int rounds,
                              unsigned long long* out, void* stream) {
  (void)hipMemsetAsync(flag, 0, 512, (hipStream_t)stream);
  (void)hipMemsetAsync(out, 0, 32, (hipStream_t)stream);
  hipLaunchKernelGGL(handoff_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, (u32x4*)data, flag, t_raise, vec_per_thread, cons,
                     rounds, out);
  return (int)hipGetLastError();
}
