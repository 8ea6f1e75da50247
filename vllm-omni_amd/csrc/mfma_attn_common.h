// Shared pieces of the matrix-core attention kernels (decode + prefill): MFMA wrappers, the
// swizzled [rows][256 B] LDS image, inline-asm LDS reads with their waits, counted vmcnt waits.
#pragma once
#include <type_traits>

#include "common.h"

namespace mi {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

template <typename Tag>
__device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mfma16<bf16_tag>(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma16<f16_tag>(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// 32x32x16: A lane l = row l % 32, k = 8 * (l / 32) + j; B lane l = column l % 32, same k; C register i of lane l =
// row 8 * (i / 4) + 4 * (l / 32) + i % 4, column l % 32.  The MFMA holds the SIMD's vector issue port for 8 of its 32
// cycles (16x16x32: 8 of 16), which is what leaves room for a softmax beside it.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <typename Tag>
__device__ __forceinline__ f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c);
template <>
__device__ __forceinline__ f32x16_t mfma32<bf16_tag>(const uint4& a, const uint4& b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16_t mfma32<f16_tag>(const uint4& a, const uint4& b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// K = 16 form: A and B fragments are 4 elements (one 64-bit register pair) per lane, lane l holds
// k = 4*(l>>4) + j — exactly what one `ds_read_b64_tr_b16` delivers, so a transposed read IS an operand.
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
template <typename Tag>
__device__ __forceinline__ f32x4_t mfma16_k16(const u32x2_t& a, const u32x2_t& b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mfma16_k16<bf16_tag>(const u32x2_t& a, const u32x2_t& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b),
                                                   c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma16_k16<f16_tag>(const u32x2_t& a, const u32x2_t& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a), __builtin_bit_cast(f16x4_t, b), c,
                                               0, 0, 0);
}

constexpr int kStep = 32;  // tokens per pipeline step

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// LDS reads are inline asm on purpose: hipcc puts `s_waitcnt vmcnt(0)` in front of any LDS read it
// can see while an LDS-DMA is outstanding, which would drain the prefetched next stage (programming
// guide §5 "Pipelining across barriers", §5.7).  Waits are therefore placed by hand: vmcnt(N) counts
// the DMA instructions issued after the stage being read; the lgkmcnt wait names every destination
// register as "+v" so no consumer can be scheduled above it.

// 8 row reads (two sub-tiles x four k-steps) AND their wait in ONE statement: a compiler-inserted
// copy of a destination register can then only happen after the data has landed (a copy placed
// between a bare asm ds_read and a later wait statement reads stale registers — seen on hardware).
template <int IMM0, int IMM1>
__device__ __forceinline__ void lds_read_rows8(const uint32_t (&addr)[4], u32x4_t (&r0)[4], u32x4_t (&r1)[4]) {
  asm volatile(
      "ds_read_b128 %0, %8 offset:%12\n\t"
      "ds_read_b128 %4, %8 offset:%13\n\t"
      "ds_read_b128 %1, %9 offset:%12\n\t"
      "ds_read_b128 %5, %9 offset:%13\n\t"
      "ds_read_b128 %2, %10 offset:%12\n\t"
      "ds_read_b128 %6, %10 offset:%13\n\t"
      "ds_read_b128 %3, %11 offset:%12\n\t"
      "ds_read_b128 %7, %11 offset:%13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0[0]), "=&v"(r0[1]), "=&v"(r0[2]), "=&v"(r0[3]), "=&v"(r1[0]), "=&v"(r1[1]), "=&v"(r1[2]),
        "=&v"(r1[3])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "i"(IMM0), "i"(IMM1)
      : "memory");
}
// 8 transposed reads (four d-tiles x two sub-tiles) and their wait in one statement.
template <int IMM0, int IMM1>
__device__ __forceinline__ void lds_read_tr8(const uint32_t* addr, u32x2_t (&a0)[4], u32x2_t (&a1)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %1, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %5, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %6, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %3, %11 offset:%12\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a0[0]), "=&v"(a0[1]), "=&v"(a0[2]), "=&v"(a0[3]), "=&v"(a1[0]), "=&v"(a1[1]), "=&v"(a1[2]),
        "=&v"(a1[3])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "i"(IMM0), "i"(IMM1)
      : "memory");
}
// 16 transposed reads (eight d-tiles x two sub-tiles: head_dim 128) and ONE wait: one exposed LDS latency per V tile.
template <int IMM0, int IMM1>
__device__ __forceinline__ void lds_read_tr16(const uint32_t* addr, u32x2_t (&a0)[4], u32x2_t (&a1)[4], u32x2_t (&b0)[4],
                                              u32x2_t (&b1)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16 offset:%24\n\t"
      "ds_read_b64_tr_b16 %4, %16 offset:%25\n\t"
      "ds_read_b64_tr_b16 %1, %17 offset:%24\n\t"
      "ds_read_b64_tr_b16 %5, %17 offset:%25\n\t"
      "ds_read_b64_tr_b16 %2, %18 offset:%24\n\t"
      "ds_read_b64_tr_b16 %6, %18 offset:%25\n\t"
      "ds_read_b64_tr_b16 %3, %19 offset:%24\n\t"
      "ds_read_b64_tr_b16 %7, %19 offset:%25\n\t"
      "ds_read_b64_tr_b16 %8, %20 offset:%24\n\t"
      "ds_read_b64_tr_b16 %12, %20 offset:%25\n\t"
      "ds_read_b64_tr_b16 %9, %21 offset:%24\n\t"
      "ds_read_b64_tr_b16 %13, %21 offset:%25\n\t"
      "ds_read_b64_tr_b16 %10, %22 offset:%24\n\t"
      "ds_read_b64_tr_b16 %14, %22 offset:%25\n\t"
      "ds_read_b64_tr_b16 %11, %23 offset:%24\n\t"
      "ds_read_b64_tr_b16 %15, %23 offset:%25\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a0[0]), "=&v"(a0[1]), "=&v"(a0[2]), "=&v"(a0[3]), "=&v"(a1[0]), "=&v"(a1[1]), "=&v"(a1[2]), "=&v"(a1[3]),
        "=&v"(b0[0]), "=&v"(b0[1]), "=&v"(b0[2]), "=&v"(b0[3]), "=&v"(b1[0]), "=&v"(b1[1]), "=&v"(b1[2]), "=&v"(b1[3])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
        "i"(IMM0), "i"(IMM1)
      : "memory");
}
// 8 row reads at 8 addresses, one immediate, one wait (the 32-row K tile of the 32x32x16 formulation).
template <int IMM>
__device__ __forceinline__ void lds_read_b128x8(const uint32_t (&addr)[8], u32x4_t (&r)[8]) {
  asm volatile(
      "ds_read_b128 %0, %8 offset:%16\n\t"
      "ds_read_b128 %1, %9 offset:%16\n\t"
      "ds_read_b128 %2, %10 offset:%16\n\t"
      "ds_read_b128 %3, %11 offset:%16\n\t"
      "ds_read_b128 %4, %12 offset:%16\n\t"
      "ds_read_b128 %5, %13 offset:%16\n\t"
      "ds_read_b128 %6, %14 offset:%16\n\t"
      "ds_read_b128 %7, %15 offset:%16\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
        "i"(IMM)
      : "memory");
}
// 16 transposed reads at 4 addresses x 4 immediates, one wait: o[4 * n + q] = address n, immediate q.
template <int I0, int I1, int I2, int I3>
__device__ __forceinline__ void lds_read_tr4x4(const uint32_t (&addr)[4], u32x2_t (&o)[16]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16 offset:%20\n\t"
      "ds_read_b64_tr_b16 %1, %16 offset:%21\n\t"
      "ds_read_b64_tr_b16 %2, %16 offset:%22\n\t"
      "ds_read_b64_tr_b16 %3, %16 offset:%23\n\t"
      "ds_read_b64_tr_b16 %4, %17 offset:%20\n\t"
      "ds_read_b64_tr_b16 %5, %17 offset:%21\n\t"
      "ds_read_b64_tr_b16 %6, %17 offset:%22\n\t"
      "ds_read_b64_tr_b16 %7, %17 offset:%23\n\t"
      "ds_read_b64_tr_b16 %8, %18 offset:%20\n\t"
      "ds_read_b64_tr_b16 %9, %18 offset:%21\n\t"
      "ds_read_b64_tr_b16 %10, %18 offset:%22\n\t"
      "ds_read_b64_tr_b16 %11, %18 offset:%23\n\t"
      "ds_read_b64_tr_b16 %12, %19 offset:%20\n\t"
      "ds_read_b64_tr_b16 %13, %19 offset:%21\n\t"
      "ds_read_b64_tr_b16 %14, %19 offset:%22\n\t"
      "ds_read_b64_tr_b16 %15, %19 offset:%23\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
        "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "i"(I0), "i"(I1), "i"(I2), "i"(I3)
      : "memory");
}
// Issue-only forms + a separate wait (guide §5.4 rule 18, form (iii)): the reads' latency runs under whatever is placed
// between the two.  `lds_wait_all()` ends with a full scheduling barrier, so no consumer can be hoisted above the wait;
// what this form does NOT pin is register allocation — a compiler-inserted copy of a destination register between issue
// and wait would read stale data: tools/check_prefill_isa.py checks the built kernel for exactly that.
template <int IMM>
__device__ __forceinline__ void lds_issue_b128x8(const uint32_t (&addr)[8], u32x4_t (&r)[8]) {
  asm volatile(
      "ds_read_b128 %0, %8 offset:%16\n\t"
      "ds_read_b128 %1, %9 offset:%16\n\t"
      "ds_read_b128 %2, %10 offset:%16\n\t"
      "ds_read_b128 %3, %11 offset:%16\n\t"
      "ds_read_b128 %4, %12 offset:%16\n\t"
      "ds_read_b128 %5, %13 offset:%16\n\t"
      "ds_read_b128 %6, %14 offset:%16\n\t"
      "ds_read_b128 %7, %15 offset:%16"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
        "i"(IMM)
      : "memory");
}
template <int I0, int I1, int I2, int I3>
__device__ __forceinline__ void lds_issue_tr4x4(const uint32_t (&addr)[4], u32x2_t (&o)[16]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16 offset:%20\n\t"
      "ds_read_b64_tr_b16 %1, %16 offset:%21\n\t"
      "ds_read_b64_tr_b16 %2, %16 offset:%22\n\t"
      "ds_read_b64_tr_b16 %3, %16 offset:%23\n\t"
      "ds_read_b64_tr_b16 %4, %17 offset:%20\n\t"
      "ds_read_b64_tr_b16 %5, %17 offset:%21\n\t"
      "ds_read_b64_tr_b16 %6, %17 offset:%22\n\t"
      "ds_read_b64_tr_b16 %7, %17 offset:%23\n\t"
      "ds_read_b64_tr_b16 %8, %18 offset:%20\n\t"
      "ds_read_b64_tr_b16 %9, %18 offset:%21\n\t"
      "ds_read_b64_tr_b16 %10, %18 offset:%22\n\t"
      "ds_read_b64_tr_b16 %11, %18 offset:%23\n\t"
      "ds_read_b64_tr_b16 %12, %19 offset:%20\n\t"
      "ds_read_b64_tr_b16 %13, %19 offset:%21\n\t"
      "ds_read_b64_tr_b16 %14, %19 offset:%22\n\t"
      "ds_read_b64_tr_b16 %15, %19 offset:%23"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
        "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "i"(I0), "i"(I1), "i"(I2), "i"(I3)
      : "memory");
}
// head_dim 64 forms of the three above: 4 k-steps of a 32-row K tile; 2 d-blocks x 4 row offsets of the V tile.
template <int IMM>
__device__ __forceinline__ void lds_read_b128x4(const uint32_t (&addr)[4], u32x4_t (&r)[4]) {
  asm volatile(
      "ds_read_b128 %0, %4 offset:%8\n\t"
      "ds_read_b128 %1, %5 offset:%8\n\t"
      "ds_read_b128 %2, %6 offset:%8\n\t"
      "ds_read_b128 %3, %7 offset:%8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "i"(IMM)
      : "memory");
}
template <int IMM>
__device__ __forceinline__ void lds_issue_b128x4(const uint32_t (&addr)[4], u32x4_t (&r)[4]) {
  asm volatile(
      "ds_read_b128 %0, %4 offset:%8\n\t"
      "ds_read_b128 %1, %5 offset:%8\n\t"
      "ds_read_b128 %2, %6 offset:%8\n\t"
      "ds_read_b128 %3, %7 offset:%8"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "i"(IMM)
      : "memory");
}
template <int I0, int I1, int I2, int I3>
__device__ __forceinline__ void lds_issue_tr2x4(const uint32_t (&addr)[2], u32x2_t (&o)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%10\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %3, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %9 offset:%10\n\t"
      "ds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
      "ds_read_b64_tr_b16 %6, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %7, %9 offset:%13"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(addr[0]), "v"(addr[1]), "i"(I0), "i"(I1), "i"(I2), "i"(I3)
      : "memory");
}
__device__ __forceinline__ void lds_wait_all() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// 16-B chunk swizzles of the [rows][HD*2 B] LDS images: position pc of a row holds logical chunk
// pc ^ f(row).  Any f works for correctness as long as the DMA source and every read apply the same
// one; f decides the bank conflicts (microarch guide §LDS: `ds_read_b128` is served in four fixed
// groups of 16 lanes, `ds_read_b64_tr_b16` in two halves of 32 lanes; 64 banks of 4 B).  The K image
// is only ever read by rows and the V image only transposed, so each gets the f that makes ITS read
// conflict-free (the dual-use image of the programming guide, T10 (b), is 2-way on both for the
// 16x16x32 operand — measured here: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE):
//   K, 256-B rows: f = row & 15.  A b128 lane group holds rows {0-3, 12-15} of k-group g' and rows
//     {4-11} of g'+1 (or the complement): chunks c0 ^ {0-3, 12-15} and (c0 ^ 1) ^ {4-11} are 16
//     distinct values.
//   V, 256-B rows: f = 2 * (row & 7).  A 32-lane half reads two 4-row blocks (rows 8n .. 8n+7), 32 B of
//     each row = chunks {2n', 2n'+1}: XOR with 8 distinct EVEN values keeps each pair together and
//     spreads the 8 rows over all 16 chunk slots.
//   128-B rows (head_dim 64, two rows per 256-B bank row): the chunk is XORed with the row-pair
//     index for both images.
template <int HD>
__device__ __forceinline__ int swz_k(int row) {
  if constexpr (HD == 128) return row & 15;
  else return (row >> 1) & 7;
}
template <int HD>
__device__ __forceinline__ int swz_v(int row) {
  if constexpr (HD == 128) return (row & 7) << 1;
  else return (row >> 1) & 7;
}

// 4 row reads (two sub-tiles x two k-steps: head_dim 64) and their wait in one statement.
template <int IMM0, int IMM1>
__device__ __forceinline__ void lds_read_rows4(const uint32_t (&addr)[2], u32x4_t (&r0)[2], u32x4_t (&r1)[2]) {
  asm volatile(
      "ds_read_b128 %0, %4 offset:%6\n\t"
      "ds_read_b128 %2, %4 offset:%7\n\t"
      "ds_read_b128 %1, %5 offset:%6\n\t"
      "ds_read_b128 %3, %5 offset:%7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0[0]), "=&v"(r0[1]), "=&v"(r1[0]), "=&v"(r1[1])
      : "v"(addr[0]), "v"(addr[1]), "i"(IMM0), "i"(IMM1)
      : "memory");
}


}  // namespace
}  // namespace mi
