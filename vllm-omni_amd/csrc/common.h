// Shared device/host helpers for the gfx950 kernels.  gfx950 only: wave = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vllm_omni_amd.h"

namespace mi {

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

// ---- element traits: T is the storage type tag -------------------------
struct bf16_tag {};
struct f16_tag {};

template <typename Tag>
struct Elem;

template <>
struct Elem<bf16_tag> {
  // bf16 -> f32 is a 16-bit shift; lo/hi halves of a packed dword.
  static __device__ __forceinline__ float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
  static __device__ __forceinline__ float from_bits(uint16_t b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
  }
  // round-to-nearest-even via the hardware convert (keeps NaN a NaN).
  static __device__ __forceinline__ uint16_t to_bits(float f) {
    return __builtin_bit_cast(uint16_t, (__bf16)f);
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
  }
  // acc += a.lo*b.lo + a.hi*b.hi, fp32 accumulate (v_dot2c_f32_bf16)
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                           __builtin_bit_cast(bf16x2_t, b), acc, false);
  }
};

template <>
struct Elem<f16_tag> {
  static __device__ __forceinline__ float lo(uint32_t w) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
  }
  static __device__ __forceinline__ float hi(uint32_t w) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
  }
  static __device__ __forceinline__ float from_bits(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
  static __device__ __forceinline__ uint16_t to_bits(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    f16x2_t v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
  }
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b),
                                  acc, false);
  }
};

// ---- DPP helpers (a DPP row is 16 lanes) --------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kDppQuadXor1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;  // lane i <-> 7-i inside each 8
constexpr int kDppRowMirror = 0x140;      // lane i <-> 15-i inside each 16

// all-reduce (sum) over aligned groups of N consecutive lanes, N in {8,16}
template <int N>
__device__ __forceinline__ float group_allreduce_sum(float v) {
  v += dpp_f32<kDppQuadXor1>(v);
  v += dpp_f32<kDppQuadXor2>(v);
  v += dpp_f32<kDppRowHalfMirror>(v);
  if constexpr (N == 16) v += dpp_f32<kDppRowMirror>(v);
  return v;
}

__device__ __forceinline__ float wave_allreduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_allreduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) on the VALU:
// v_permlane{16,32}_swap exchange whole rows between two registers, no LDS round trip
// (ds_bpermute, which __shfl_xor compiles to, costs an LDS access and an lgkmcnt wait each).
// `volatile`: a cross-lane exchange must not be sunk past a divergent branch or early return (a
// non-volatile form was, and read lanes that had already exited — run-to-run differences in the tests).
// Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: hipcc (ROCm 7.2) treats the two results of
// the builtin as one value (it dropped the v_max / emitted `v_add v1, v1, v1` after the swap — wrong
// results on hardware).  `s_nop 1`: the wait states hipcc itself puts
// between a VALU write of an operand and the swap.
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {  // a[32..63] <-> b[0..31]
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {  // odd rows of a <-> even rows of b
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// max without the canonicalising `v_max x, x` hipcc puts in front of every fmaxf whose operand it
// cannot prove quiet (three VALU instructions per max in the softmax).  v_med3_f32 through its builtin:
// med3(a, b, +inf) == max(a, b) for non-NaN operands, one instruction, and — unlike inline asm — an
// instruction the compiler's hazard recogniser knows, so the wait states between an MFMA writing a
// score and the VALU reading it are still inserted (an inline-asm v_max3 fed straight from MFMA
// results read them early: run-to-run differences in tests/test_gpu_attention_decode.py).
__device__ __forceinline__ float max2f(float a, float b) {
  return __builtin_amdgcn_fmed3f(a, b, __builtin_inff());
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return max2f(max2f(a, b), c); }
// (one asm statement: swap, max, swap, max — its input comes from compiler-generated VALU code, never
// straight from an MFMA; `s_nop 0` = the wait state between a swap and the VALU reading its result)
__device__ __forceinline__ float rows_allreduce_max(float v) {
  float a = v, b = v;
  asm volatile(
      "s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\t"
      "v_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0\n\t"
      "v_max_f32 %0, %0, %1"
      : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ float rows_allreduce_sum(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);
  a = b = a + b;
  permlane16_swap(a, b);
  return a + b;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Scalar-cache acquire, first statement of every kernel.  hipcc turns wave-uniform loads of small
// metadata (seq_lens, block tables, slot mapping, token ids) into SCALAR loads; those buffers are
// rewritten by host->device copies between launches, and on this stack (gfx950, ROCm 7.x) the scalar
// data cache was observed to serve stale lines across launches (prefill read a recycled address's
// old block id: tests/debug/debug_prefill.py).  s_dcache_inv is asynchronous, hence the lgkmcnt wait.
__device__ __forceinline__ void scalar_cache_acquire() {
  asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

constexpr float kLog2e = 1.4426950408889634f;

// W8A8 dequantisation of one int32 sum, spelled with explicit roundings so that every kernel that dequantises
// (tiled int8 GEMM, its split-K reduce, the simple MFMA kernel, the stand-alone epilogue) produces the SAME bits:
// fma(rnd(acc * x_scale), w_scale, bias), rounded to fp32 BEFORE the conversion to the 16-bit output.  Left to the
// compiler, `a * xs * ws + b` contracts differently per kernel, and with an f16 output hipcc folds the fma and the
// conversion into one `v_fma_mixlo_f16` (a single rounding) in some kernels and not in others — the empty asm
// pins the fp32 value.
__device__ __forceinline__ float dequant_i32(int acc, float xs, float ws, float b) {
  float r = __fmaf_rn(__fmul_rn((float)acc, xs), ws, b);
  asm volatile("" : "+v"(r));
  return r;
}

// ---- host-side launch check ---------------------------------------------
inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MI_OK : MI_ERR_LAUNCH;
}

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace mi
