// A12 — decode-sized tensor-parallel all-reduce, two-shot, over IPC-mapped peer buffers (SURVEY §8b export list,
// §8e "Collective implementation").
//
// Replaces the all-reduce implicit in upstream `RowParallelLinear` for o_proj / down_proj
// (vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:54-60,118-124) on the group that
// `init_worker_distributed_environment` creates (vllm_omni/worker/gpu_ar_worker.py:69-75), for the messages of a
// decode step: [T, H] in the 16-bit model dtype, 1.8 MB at batch 256 x hidden 3584, 2 per layer.  A ring (RCCL's
// default at this size) is bound by one xGMI link and 2(W-1) latency hops; here every rank talks to every peer
// DIRECTLY over its own link, all links busy at once, two hops:
//
//   phase 0  copy-in      my input -> my staging buffer (shared, uncached).  With `split` > 0 the input is the
//                         streaming GEMM's fp32 split-K partials [split][T][H]: the copy-in IS their reduction
//                         (+ bias, one rounding) — what mi_linear_reduce would have stored — so a row-parallel
//                         projection needs no reduce launch of its own in front of the all-reduce.
//   barrier A             every rank's staging is complete
//   phase 1  reduce-scatter  rank r sums slice r of ALL ranks' staging in fp32, IN RANK ORDER 0..W-1 (the same order
//                         on every rank: results are bit-identical across ranks and run to run), rounds once and
//                         stores the slice in its own `tmp` buffer (shared, uncached)
//   barrier B             every rank's reduced slice is complete
//   phase 2  all-gather   every rank reads slice r from rank r's tmp into its output
//
// No barrier C: staging is rewritten only by the next call's phase 0, which its owner reaches after barrier B of
// this call (by then every peer has finished reading it); tmp only by the next call's phase 1, behind the next
// barrier A (every peer has finished this call's phase 2 before it arrives there).
//
// Memory model.  The shared region (flags + staging + tmp) is `hipExtMallocWithFlags(hipDeviceMallocUncached)`:
// fine-grained, never resident in an L2 that the reading GPU cannot see.  Payload stores and loads are
// `global_* sc0 sc1` (system scope, write-through / cache-bypassing); every storing wave drains its stores
// (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, and only then do W lanes publish the flag to the W
// peers (MI355X_MICROARCH.md "Valid forms": sc1 payload -> drained -> flag).  Thread i of workgroup b reads, in
// every phase, only bytes that thread i of workgroup b of a peer wrote in the phase before (all phases index the
// message as slice * part + idx with the same idx per thread), so a per-workgroup flag exchange is enough: no
// grid-wide barrier.  Flags are monotonic per workgroup and compared wrap-safe; the counter lives in device
// memory, so the launch carries no per-call value and replays from a hipGraph.
//
// Every spin is bounded (`timeout_ms`, s_memrealtime): a peer that never arrives ends the kernel with
// status != 0 instead of hanging the node.
#include <hip/hip_runtime.h>

#include <new>

#include "common.h"

namespace mi {
namespace {

constexpr int kMaxRanks = MI_ALLREDUCE_MAX_RANKS;
constexpr int kMaxBlocks = 64;
constexpr int kThreads = 512;

struct alignas(128) Signal {
  // flag[k][b][r]: barrier k (A = 0, B = 1) of workgroup b, written by rank r
  uint32_t flag[2][kMaxBlocks][kMaxRanks];
  uint32_t counter[kMaxBlocks];   // this rank's call counter per workgroup (read and advanced by the kernel)
  uint32_t status;                // != 0: a spin timed out (sticky)
};

struct Peers {
  Signal* sig[kMaxRanks];
  uint8_t* staging[kMaxRanks];
  uint8_t* tmp[kMaxRanks];
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// System-scope (sc0 sc1) 16-byte loads / stores through buffer descriptors: builtins, so hipcc counts them in
// vmcnt itself (an inline-asm load's destination is unprotected until a hand-placed wait).  aux: bit 0 = sc0,
// bit 4 = sc1.  Descriptors are built from kernel arguments only (wave-uniform).
constexpr int kAuxSys = 1 | 16;
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 load_sys(rsrc_t r, int64_t vec) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(vec * 16), 0, kAuxSys);
}
__device__ __forceinline__ void store_sys(rsrc_t r, int64_t vec, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(vec * 16), 0, kAuxSys);
}
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// All threads call.  `flag` is this call's value for barrier `k`.  Returns false on timeout.
__device__ __forceinline__ bool exchange(const Peers& P, int k, int rank, int world, uint32_t flag, uint64_t timeout_ticks,
                                         uint32_t* host_status) {
  wait_vm();            // this wave's payload stores have left
  __syncthreads();      // ... and every other wave's of the workgroup
  bool ok = true;
  if (threadIdx.x < world) {
    const int peer = threadIdx.x;
    __scoped_atomic_store_n(&P.sig[peer]->flag[k][blockIdx.x][rank], flag, __ATOMIC_RELEASE, __MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = &P.sig[rank]->flag[k][blockIdx.x][peer];
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__scoped_atomic_load_n(mine, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM) - flag) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
        ok = false;
        break;
      }
    }
    __scoped_atomic_thread_fence(__ATOMIC_ACQUIRE, __MEMORY_SCOPE_SYSTEM);
    if (!ok) {
      __scoped_atomic_store_n(&P.sig[rank]->status, 1u + (uint32_t)k, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
      // the same word in host-mapped pinned memory: the host reads it at a synchronisation point it already has
      // (mi_allreduce_poll) instead of paying a device->host copy per step
      __scoped_atomic_store_n(host_status, 1u + (uint32_t)k, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
    }
  }
  return __syncthreads_and(ok);
}

template <typename Tag>
__device__ __forceinline__ void acc8(float (&a)[8], u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[2 * i] += Elem<Tag>::lo(v[i]);
    a[2 * i + 1] += Elem<Tag>::hi(v[i]);
  }
}
template <typename Tag>
__device__ __forceinline__ u32x4 pack8(const float (&a)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = Elem<Tag>::pack(a[2 * i], a[2 * i + 1]);
  return v;
}

// part = 16-byte vectors per slice (the message is padded up to world * part vectors; vectors >= nvec are skipped)
template <typename Tag>
__global__ __launch_bounds__(kThreads) void two_shot_kernel(Peers P, void* __restrict__ out, const void* __restrict__ in,
                                                            const void* __restrict__ bias, int split, int64_t cols,
                                                            int64_t nvec, int64_t part, int64_t buf_bytes, int rank,
                                                            int world, uint64_t timeout_ticks, uint32_t* host_status) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const uint32_t base = P.sig[rank]->counter[blockIdx.x];   // plain: only this workgroup ever writes it
  // ---- phase 0: copy-in (or split-K reduce) into my staging ----
  const rsrc_t stage = make_rsrc(P.staging[rank], buf_bytes);
  for (int r = 0; r < world; ++r) {
    for (int64_t idx = tid; idx < part; idx += stride) {
      const int64_t v = (int64_t)r * part + idx;
      if (v >= nvec) break;
      u32x4 x;
      if (split == 0) {
        x = reinterpret_cast<const u32x4*>(in)[v];
      } else {   // fp32 partials [split][rows][cols]: 8 consecutive elements = two float4 per slice
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float4* p = reinterpret_cast<const float4*>(in) + 2 * v;
        for (int s = 0; s < split; ++s) {
          const float4 lo = p[(int64_t)s * 2 * nvec], hi = p[(int64_t)s * 2 * nvec + 1];
          a[0] += lo.x; a[1] += lo.y; a[2] += lo.z; a[3] += lo.w;
          a[4] += hi.x; a[5] += hi.y; a[6] += hi.z; a[7] += hi.w;
        }
        if (bias) {
          const u32x4 b = reinterpret_cast<const u32x4*>(bias)[(v * 8 % cols) / 8];
          acc8<Tag>(a, b);
        }
        x = pack8<Tag>(a);
      }
      store_sys(stage, v, x);
    }
  }
  bool ok = exchange(P, 0, rank, world, base + 1, timeout_ticks, host_status);
  // ---- phase 1: reduce my slice over all ranks, rank order ----
  if (ok) {
    const rsrc_t tmp = make_rsrc(P.tmp[rank], buf_bytes);
    for (int64_t idx = tid; idx < part; idx += stride) {
      const int64_t v = (int64_t)rank * part + idx;
      if (v >= nvec) break;
      u32x4 x[kMaxRanks];
#pragma unroll
      for (int r = 0; r < kMaxRanks; ++r)
        if (r < world) x[r] = load_sys(make_rsrc(P.staging[r], buf_bytes), v);
      float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < kMaxRanks; ++r)
        if (r < world) acc8<Tag>(a, x[r]);
      store_sys(tmp, idx, pack8<Tag>(a));
    }
    ok = exchange(P, 1, rank, world, base + 1, timeout_ticks, host_status);
  }
  // ---- phase 2: gather every rank's reduced slice ----
  if (ok) {
    u32x4* o = reinterpret_cast<u32x4*>(out);
    for (int64_t idx = tid; idx < part; idx += stride) {
      u32x4 x[kMaxRanks];
#pragma unroll
      for (int r = 0; r < kMaxRanks; ++r)
        if (r < world && (int64_t)r * part + idx < nvec) x[r] = load_sys(make_rsrc(P.tmp[r], buf_bytes), idx);
#pragma unroll
      for (int r = 0; r < kMaxRanks; ++r)
        if (r < world && (int64_t)r * part + idx < nvec) o[(int64_t)r * part + idx] = x[r];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) P.sig[rank]->counter[blockIdx.x] = base + 1;
}

}  // namespace
}  // namespace mi

struct mi_allreduce {
  int rank = 0, world = 0;
  int64_t max_bytes = 0;
  int timeout_ms = 10000;
  uint8_t* region = nullptr;              // Signal | staging | tmp (this rank's, uncached)
  void* peer_base[MI_ALLREDUCE_MAX_RANKS] = {};
  bool opened[MI_ALLREDUCE_MAX_RANKS] = {};
  mi::Peers peers = {};
  bool connected = false;
  uint32_t* host_status = nullptr;        // host-mapped pinned word, written by the kernel when a spin times out
  uint32_t* host_status_dev = nullptr;    // its device address
};

static int64_t region_bytes(int64_t max_bytes) { return (int64_t)sizeof(mi::Signal) + 2 * max_bytes; }

extern "C" int mi_allreduce_init(mi_allreduce** out, int32_t rank, int32_t world, int64_t max_bytes, int32_t timeout_ms,
                                 void* ipc_handle_out) {
  if (!out || !ipc_handle_out || world < 2 || world > MI_ALLREDUCE_MAX_RANKS || rank < 0 || rank >= world ||
      max_bytes <= 0 || max_bytes > (1ll << 30))
    return MI_ERR_INVALID_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == MI_ALLREDUCE_HANDLE_BYTES, "IPC handle size");
  max_bytes = (max_bytes + 4095) / 4096 * 4096;
  mi_allreduce* ar = new (std::nothrow) mi_allreduce();
  if (!ar) return MI_ERR_EXHAUSTED;
  ar->rank = rank;
  ar->world = world;
  ar->max_bytes = max_bytes;
  if (timeout_ms > 0) ar->timeout_ms = timeout_ms;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, region_bytes(max_bytes), hipDeviceMallocUncached) != hipSuccess) {
    delete ar;
    return MI_ERR_EXHAUSTED;
  }
  ar->region = static_cast<uint8_t*>(p);
  void* hs = nullptr;
  void* hs_dev = nullptr;
  if (hipHostMalloc(&hs, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(&hs_dev, hs, 0) != hipSuccess) {
    if (hs) (void)hipHostFree(hs);
    (void)hipFree(p);
    delete ar;
    return MI_ERR_EXHAUSTED;
  }
  ar->host_status = static_cast<uint32_t*>(hs);
  ar->host_status_dev = static_cast<uint32_t*>(hs_dev);
  *ar->host_status = 0;
  if (hipMemset(p, 0, sizeof(mi::Signal)) != hipSuccess ||
      hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(ipc_handle_out), p) != hipSuccess) {
    (void)hipFree(p);
    (void)hipHostFree(hs);
    delete ar;
    return MI_ERR_STATE;
  }
  *out = ar;
  return MI_OK;
}

extern "C" int mi_allreduce_register_buffer(mi_allreduce* ar, const void* ipc_handles) {
  using namespace mi;
  if (!ar || !ipc_handles) return MI_ERR_INVALID_ARG;
  if (ar->connected) return MI_ERR_STATE;
  const hipIpcMemHandle_t* h = reinterpret_cast<const hipIpcMemHandle_t*>(ipc_handles);
  for (int r = 0; r < ar->world; ++r) {
    void* base = ar->region;
    if (r != ar->rank) {
      if (hipIpcOpenMemHandle(&base, h[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return MI_ERR_STATE;
      ar->opened[r] = true;
    }
    ar->peer_base[r] = base;
    uint8_t* b = static_cast<uint8_t*>(base);
    ar->peers.sig[r] = reinterpret_cast<Signal*>(b);
    ar->peers.staging[r] = b + sizeof(Signal);
    ar->peers.tmp[r] = b + sizeof(Signal) + ar->max_bytes;
  }
  ar->connected = true;
  return MI_OK;
}

extern "C" int mi_allreduce_run(mi_allreduce* ar, void* out, const void* in, int64_t count, int32_t split,
                                const void* bias, int64_t cols, int dtype, void* stream) {
  using namespace mi;
  if (!ar || !out || !in || count < 0 || split < 0) return MI_ERR_INVALID_ARG;
  if (!ar->connected) return MI_ERR_STATE;
  // a spin has timed out before: this rank's call counters are out of step with its peers' for good — the
  // communicator is dead, further launches would return garbage
  if (__atomic_load_n(ar->host_status, __ATOMIC_RELAXED) != 0) return MI_ERR_STATE;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (count == 0) return MI_OK;
  if (count % 8 || ((uintptr_t)out & 15) || ((uintptr_t)in & 15)) return MI_ERR_INVALID_ARG;
  if (split > 0 && bias && (cols <= 0 || cols % 8 || count % cols || ((uintptr_t)bias & 15))) return MI_ERR_INVALID_ARG;
  if (split == 0 && bias) return MI_ERR_INVALID_ARG;   // the 16-bit input already carries its bias
  const int64_t nvec = count / 8;
  const int64_t part = (nvec + ar->world - 1) / ar->world;
  if ((int64_t)part * ar->world * 16 > ar->max_bytes) return MI_ERR_WORKSPACE;
  int blocks = (int)((part + kThreads - 1) / kThreads);
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  if (blocks < 1) blocks = 1;
  const uint64_t ticks = (uint64_t)ar->timeout_ms * 100000ull;   // s_memrealtime: 100 MHz
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == MI_DTYPE_BF16)
    two_shot_kernel<bf16_tag><<<blocks, kThreads, 0, s>>>(ar->peers, out, in, bias, split, cols, nvec, part, ar->max_bytes,
                                                          ar->rank, ar->world, ticks, ar->host_status_dev);
  else
    two_shot_kernel<f16_tag><<<blocks, kThreads, 0, s>>>(ar->peers, out, in, bias, split, cols, nvec, part, ar->max_bytes,
                                                         ar->rank, ar->world, ticks, ar->host_status_dev);
  return launch_status();
}

extern "C" int mi_allreduce_status(mi_allreduce* ar) {
  if (!ar || !ar->region) return MI_ERR_INVALID_ARG;
  uint32_t st = 0;
  if (hipMemcpy(&st, &reinterpret_cast<mi::Signal*>(ar->region)->status, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess)
    return MI_ERR_STATE;
  return st == 0 ? MI_OK : MI_ERR_STATE;
}

extern "C" int mi_allreduce_poll(mi_allreduce* ar) {
  if (!ar || !ar->host_status) return MI_ERR_INVALID_ARG;
  return __atomic_load_n(ar->host_status, __ATOMIC_ACQUIRE) == 0 ? MI_OK : MI_ERR_STATE;
}

extern "C" int mi_allreduce_destroy(mi_allreduce* ar) {
  if (!ar) return MI_OK;
  for (int r = 0; r < ar->world; ++r)
    if (ar->opened[r]) (void)hipIpcCloseMemHandle(ar->peer_base[r]);
  if (ar->region) (void)hipFree(ar->region);
  if (ar->host_status) (void)hipHostFree(ar->host_status);
  delete ar;
  return MI_OK;
}
