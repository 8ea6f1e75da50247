// Probe (not shipped): what a grid-wide barrier inside one persistent launch costs on MI355X, next to streaming a phase's
// weights — the two numbers that decide whether a one-launch-per-call code predictor can beat captured launches.
//   phases x { every workgroup sums `bytes_per_phase / nwg` bytes of `w` (16-B loads, DEPTH in flight) ; barrier }
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>   // 0: release / acquire at agent scope, 1: relaxed atomics (+ nothing), 2: no barrier at all
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
  __syncthreads();
  if constexpr (MODE == 2) return;
  if (threadIdx.x == 0) {
    if constexpr (MODE == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (true) {
      unsigned v;
      if constexpr (MODE == 0) v = __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      else v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= target) break;
      if (++spins > (1u << 16)) { *err = 1; break; }   // never hang the device
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// MODE 3: two levels, agent-scope operations throughout.  The workgroups of one XCD (blockIdx % 8: round-robin dispatch) meet on
// their own counter (32 same-address atomics in a row instead of 256); the last of them adds to the device-wide counter, waits
// for the eight XCDs, and releases its XCD through a flag the others poll.
// mem: [0] device counter, [32 * (1 + x)] XCD x's counter, [32 * (9 + x)] XCD x's flag  (128-byte lines apart)
__device__ __forceinline__ void tree_barrier(unsigned* mem, unsigned gen, unsigned* err) {   // gen = 1, 2, 3, ...
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7u, per = (gridDim.x + 7u - x) / 8u;
    unsigned spins = 0;
    if (__hip_atomic_fetch_add(mem + 32 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen * per - 1u) {
      __hip_atomic_fetch_add(mem, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(mem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u * gen) {
        if (++spins > (1u << 16)) { *err = 2; break; }
      }
      __hip_atomic_store(mem + 32 * (9 + x), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(mem + 32 * (9 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
        if (++spins > (1u << 16)) { atomicOr(err, 4u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(512) void probe_kernel(const u32x4* w, int64_t vec_per_phase, int phases, unsigned* ctr, unsigned* err,
                                                    unsigned* sink) {
  const int nwg = gridDim.x;
  const int64_t per_wg = vec_per_phase / nwg;
  unsigned acc = 0;
  for (int ph = 0; ph < phases; ++ph) {
    const u32x4* base = w + (int64_t)ph * vec_per_phase + (int64_t)blockIdx.x * per_wg;
    for (int64_t i = threadIdx.x; i < per_wg; i += 512 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = i + j * 512 < per_wg ? __builtin_nontemporal_load(base + i + j * 512) : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    }
    if constexpr (MODE == 3) tree_barrier(ctr, (unsigned)(ph + 1), err);
    else grid_barrier<MODE>(ctr, (unsigned)(nwg * (ph + 1)), err);
  }
  if (threadIdx.x == 0) {   // how many workgroups do NOT sit on XCD blockIdx % 8
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    if (xcc != (blockIdx.x & 7u)) atomicAdd(sink + 1, 1u);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int probe_launch(const void* w, int64_t bytes_per_phase, int phases, int nwg, int mode, unsigned* ctr, unsigned* err,
                            unsigned* sink, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(ctr, 0, 4096, s);
  const int64_t vecs = bytes_per_phase / 16;
  if (mode == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(nwg), dim3(512), 0, s, (const u32x4*)w, vecs, phases, ctr, err, sink);
  else if (mode == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(nwg), dim3(512), 0, s, (const u32x4*)w, vecs, phases, ctr, err, sink);
  else if (mode == 3) hipLaunchKernelGGL(probe_kernel<3>, dim3(nwg), dim3(512), 0, s, (const u32x4*)w, vecs, phases, ctr, err, sink);
  else hipLaunchKernelGGL(probe_kernel<2>, dim3(nwg), dim3(512), 0, s, (const u32x4*)w, vecs, phases, ctr, err, sink);
  return (int)hipGetLastError();
}
