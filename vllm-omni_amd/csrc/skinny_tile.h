// The first-generation decode-batch linear tile (see skinny_gemm.hip for the design): parameters, fragment helpers and the
// tile itself as a device function — the stand-alone kernel (skinny_gemm.hip) and the code predictor's one-launch stack
// (cp_stack.hip) both run it.
#pragma once
#include "mfma_attn_common.h"

// experiment switches (csrc/Makefile `make variant VARIANT_DEFS=...`); the defaults are the measured best
#ifndef MI_SKINNY_W_AUX
#define MI_SKINNY_W_AUX 2        // cache policy of the weight pieces: 0 default, 2 = nt (streamed once: -0.6 % on the step)
#endif
#ifndef MI_SKINNY_NL5
#define MI_SKINNY_NL5 2          // loader waves of the 160-row tile (10 consumer waves)
#endif
#ifndef MI_SKINNY_MH8
#define MI_SKINNY_MH8 2          // consumer waves per slab at 65-128 rows (8 m-tiles)
#endif
#ifndef MI_SKINNY_RING_KB
// LDS the ring of the first-generation kernel may take: stages = clamp(MI_SKINNY_RING_KB / stage, 3, 8).  A step's
// pieces take ~1.1 us from issue to landing, so a CU needs ~75 KB in flight to take in the ~68 GB/s its L1 path
// delivers (MI355X_MICROARCH.md, ring-gemm); three stages of a 20 KB step keep 40 KB in flight.  (Round 3 measured
// deeper rings level at <= 64 rows on the dp8 shapes — HBM-bound there, profiles/r03k_ring_depth_ab.log.)
#define MI_SKINNY_RING_KB 144
#endif
#ifndef MI_SKINNY_GROUPED_NST
#define MI_SKINNY_GROUPED_NST 3  // ring stages of the grouped (sparse-MoE expert) form: 72 KB, two workgroups per CU
#endif
#ifndef MI_SKINNY_LOADER_PRIO
#define MI_SKINNY_LOADER_PRIO 3  // s_setprio of the loader waves (their issue slots come before the consumers')
#endif

namespace mi {
namespace {

// W8A8 operands (scheme of vllm_omni/quantization/int8_config.py:180-212,323-342): the SAME kernels run on
// int8 rows — a 128-byte image row is then 128 k values instead of 64, a lane's 16-byte fragment 16 k values of
// v_mfma_i32_16x16x64_i8, and every byte offset is unchanged (the host passes K and the row strides in 2-byte
// units).  Sums are int32 (exact); the epilogue dequantises: acc * x_scale[m] * w_scale[n] (+ bias[n]).
struct i8_tag {};
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
template <typename Tag>
struct AccOf { using type = f32x4_t; };
template <>
struct AccOf<i8_tag> { using type = i32x4_t; };

struct SkinnyParams {
  const uint16_t* x;
  const uint16_t* w;
  const uint16_t* bias;   // may be null
  uint16_t* out;          // S == 1
  float* part;            // S > 1: [S][M][N] (fp32 sums; int32 sums for int8 operands)
  const float* xs;        // int8 operands: per-row activation scales [M]
  const float* ws;        // int8 operands: per-output-channel weight scales [N] (GATED: [2 N], gate then up)
  int32_t M, N, K, S;     // GATED: N = output columns I; the weight has 2 I rows (gate rows, then up rows)
  int32_t RB, U;          // first-generation kernel, not GROUPED: row blocks of 16 MT rows; U = tiles x S work items per row block
  // S > 1 with the reduce INSIDE the launch (mi_linear_fused_reduce): arrival counters [tiles x RB], zero before the
  // first launch and left at zero by every launch; the slice of a (tile, row block) that arrives last sums the slabs
  int32_t* cnt = nullptr;
  // GROUPED (sparse-MoE expert GEMMs, fused_moe.hip): grid.y = one tile of `rpt` token rows that share one expert
  const int32_t* row_map;      // [max_rows] pair index per row or -1
  const int32_t* tile_expert;  // [max_tiles]
  const int32_t* num_tiles;    // [1]
  int32_t rpt, gather_div;     // rows per tile (32 / 64); gather_div > 0: source row = row_map[r] / gather_div, else r
  // GROUPED: a workgroup runs `chain` consecutive weight tiles of its row tile back to back through ONE ring — the loaders
  // go on to the next tile's stages while the consumers store the last one (a K of 384 is six steps: a workgroup per tile
  // spent more time filling and draining its ring than streaming)
  int32_t chain = 1;
  int64_t w_expert_stride;     // elements
  int32_t dbg;            // MI_LINEAR_DBG (timing experiments only): 1 = no DMA, 2 = no fragment reads / MFMAs, 4 = no reduce launch, 8 = no output stores
  int64_t ldx, ldo;
};

// operands as plain vectors (a HIP `uint4` is a struct: its copies between address spaces become memcpys that keep
// the staging array in scratch)
template <typename Tag>
__device__ __forceinline__ typename AccOf<Tag>::type mma(u32x4_t a, u32x4_t b, typename AccOf<Tag>::type c) {
  if constexpr (std::is_same_v<Tag, bf16_tag>)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else if constexpr (std::is_same_v<Tag, f16_tag>)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, a), __builtin_bit_cast(i32x4_t, b), c, 0, 0, 0);
}

// One output quad (4 consecutive columns n .. n + 3 of row m) of the epilogue, as fp32 before the output rounding:
// 16-bit operands: the sums themselves (+ bias); int8 operands: dequantised.
template <typename Tag>
__device__ __forceinline__ void finish_quad(const SkinnyParams& p, const typename AccOf<Tag>::type& a, int m, int n_scale,
                                            const float (&bv)[4], float (&v)[4]) {
  if constexpr (std::is_same_v<Tag, i8_tag>) {
    const float xs = p.xs[m];
    const float4 ws = *reinterpret_cast<const float4*>(p.ws + n_scale);
    v[0] = dequant_i32(a[0], xs, ws.x, bv[0]);
    v[1] = dequant_i32(a[1], xs, ws.y, bv[1]);
    v[2] = dequant_i32(a[2], xs, ws.z, bv[2]);
    v[3] = dequant_i32(a[3], xs, ws.w, bv[3]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a[r] + bv[r];
  }
}

constexpr int kAuxAgent = 16;   // buffer load / store aux: sc1 = device (agent) coherence, write-through stores

// a 16-bit output quad: plain, or written through to the device's coherence point
template <bool COH>
__device__ __forceinline__ void store_out_b64(uint16_t* dst, uint2 q) {
  if constexpr (COH) {
    const u32x2_t v = {q.x, q.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
  } else {
    *reinterpret_cast<uint2*>(dst) = q;
  }
}

__device__ __forceinline__ u32x4_t lds_poll128(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_store32(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

__device__ __forceinline__ int kperm(int u, int kg) {
  // logical 16-byte chunk (8 elements) of a 64-wide k block held by k-group kg at k-step u
  const int sig = (0x2130 >> (4 * kg)) & 3;   // sigma = {0, 3, 1, 2}
  return 2 * sig + u;
}

// 8 fragment reads (4 m-tiles x 2 k-steps) and their wait in ONE statement (see mfma_attn_common.h: LDS reads are
// inline asm so that hipcc does not drain the outstanding LDS-DMA in front of them)
template <int OFF>
__device__ __forceinline__ void lds_read_x8(uint32_t a0, uint32_t a1, u32x4_t (&b0)[4], u32x4_t (&b1)[4]) {
  asm volatile(
      "ds_read_b128 %0, %8 offset:%10\n\t"
      "ds_read_b128 %4, %9 offset:%10\n\t"
      "ds_read_b128 %1, %8 offset:%11\n\t"
      "ds_read_b128 %5, %9 offset:%11\n\t"
      "ds_read_b128 %2, %8 offset:%12\n\t"
      "ds_read_b128 %6, %9 offset:%12\n\t"
      "ds_read_b128 %3, %8 offset:%13\n\t"
      "ds_read_b128 %7, %9 offset:%13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(b0[0]), "=&v"(b0[1]), "=&v"(b0[2]), "=&v"(b0[3]), "=&v"(b1[0]), "=&v"(b1[1]), "=&v"(b1[2]), "=&v"(b1[3])
      : "v"(a0), "v"(a1), "i"(OFF), "i"(OFF + 2048), "i"(OFF + 4096), "i"(OFF + 6144)
      : "memory");
}
// the slab's weight fragments: 2 row groups x 2 k-steps
template <int OFF>
__device__ __forceinline__ void lds_read_w(uint32_t a0, uint32_t a1, u32x4_t (&w)[2][2]) {
  asm volatile(
      "ds_read_b128 %0, %4 offset:%6\n\t"
      "ds_read_b128 %1, %5 offset:%6\n\t"
      "ds_read_b128 %2, %4 offset:%7\n\t"
      "ds_read_b128 %3, %5 offset:%7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(w[0][0]), "=&v"(w[0][1]), "=&v"(w[1][0]), "=&v"(w[1][1])
      : "v"(a0), "v"(a1), "i"(OFF), "i"(OFF + 2048)
      : "memory");
}

// vmcnt wait of a loader wave that keeps up to A - 1 later steps (PPL pieces each) in flight behind the one it waits
// for; `rem` = steps issued after that one (fewer than A - 1 at the tail of the k range)
template <int PPL, int A>
__device__ __forceinline__ void wait_steps_behind(int rem) {
  if constexpr (A <= 1) {
    wait_vmcnt<0>();
  } else {
    if (rem >= A - 1) wait_vmcnt<PPL * (A - 1)>();
    else wait_steps_behind<PPL, A - 1>(rem);
  }
}

// NSLAB: 32-row weight slabs per tile (tile = 32 NSLAB rows); MT: 16-row m-tiles (x rows padded to 16 MT);
// MH: consumer waves per slab, each with MT / MH m-tiles; NL: loader waves.
// GATED: gate_up_proj with the SiluAndMul that follows it (qwen2_old.py:65-68: act_fn(gate_up_proj(x))) in the epilogue —
// a slab is 16 gate rows and the 16 matching up rows, a tile covers 16 NSLAB output columns, and the [M, 2 I]
// intermediate never goes to memory.
// NST: stages of the LDS ring; NST - 1 steps are in flight ahead of the consumers (three everywhere: MI_SKINNY_NST4).
// COH: the tile is one stage of a producer / consumer chain INSIDE one launch (csrc/cp_stack.hip): x is read and every output
// written at device scope (sc1: past the XCD's L2), so that a grid barrier — not a kernel boundary — can stand between them.
template <typename Tag, typename OT, int NSLAB, int MT, int MH, int NL, bool GATED, bool GROUPED = false, int NST = 3, bool COH = false>
__device__ __forceinline__ void skinny_tile(const SkinnyParams& p, const int wg_linear, char* const smem) {
  // One pipeline step = one 64-wide k block.  A stage of the LDS ring is one image of 128-byte rows: the 16 MT rows of
  // x, then the tile's 32 NSLAB weight rows; 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7).  The image is
  // filled by LDS-DMA in 1 KiB pieces (8 rows per wave-instruction, the swizzle applied on the source side).
  // Roles: NL loader waves only issue DMA; NSLAB * MH consumer waves only read fragments and run MFMAs — the DMA issue
  // cost (60-180 cycles per piece) then overlaps the other waves' MFMAs instead of sitting in front of them.  One raw
  // s_barrier per step: a loader arrives once its pieces of step `it` have landed, a consumer once it is done reading
  // step it - 1; past it the loaders refill that stage with step it + 2.  Three stages, two steps always in flight.
  constexpr int TR = 32 * NSLAB;
  constexpr int NC = NSLAB * MH;
  constexpr int MTW = MT / MH;
  constexpr int kXBytes = MT * 2048;
  constexpr int kStage = kXBytes + TR * 128;
  constexpr int XI = MT * 2, PIECES = XI + TR / 8;
  constexpr int PPL = PIECES / NL;
  constexpr int AHEAD = NST - 1;   // steps in flight
  static_assert(PIECES % NL == 0 && PPL * (AHEAD - 1) <= 63, "equal shares; the steps in flight must fit vmcnt");
  static_assert(MTW % 4 == 0 && NST * kStage <= 160 * 1024, "m-tiles are read in fours; the ring must fit the LDS");
  using E = Elem<OT>;
  using acc_t = typename AccOf<Tag>::type;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint16_t* wbase = p.w;
  int m0 = 0, wg = wg_linear;
  if constexpr (GROUPED) {
    // the maps were written by the alignment kernel just before this launch: keep their reads off a stale scalar cache
    scalar_cache_acquire();
    if ((int)blockIdx.y >= p.num_tiles[0]) return;   // whole workgroup, before any barrier
    wbase += (int64_t)p.tile_expert[blockIdx.y] * p.w_expert_stride;
  } else {
    // More rows than one image holds (16 MT): RB row blocks per (tile, slice).  Workgroup ids go round-robin over the 8
    // XCDs, so the row blocks of one work item are dealt 8 ids apart: id = 8 RB q + 8 rb + j <-> item u = 8 q + j, row
    // block rb.  They start together and on ONE XCD: the weight tile they share leaves HBM once and serves the others
    // from that XCD's L2 (speed only: nothing depends on the placement).
    const int per = 8 * p.RB, q = wg / per, rem = wg - q * per;
    m0 = (rem >> 3) * (16 * MT);
    wg = 8 * q + (rem & 7);
    if (wg >= p.U) return;   // whole workgroup, before any barrier
  }
  // XCD-aware order: slice = item % S puts every k slice of x on the XCDs congruent to it (S = 8: one slice per XCD) —
  // its rows are then re-read from ONE L2 by all the tiles
  const int s = wg % p.S;
  const int chain = GROUPED ? p.chain : 1;
  const int tile0 = (wg / p.S) * chain;     // first of the workgroup's `chain` tiles
  const int nb = p.K >> 6;
  const int b_lo = (int)((int64_t)s * nb / p.S), b_hi = (int)((int64_t)(s + 1) * nb / p.S);
  const int n_it = b_hi - b_lo;

  if (wid >= NC) {
    // ------------------------------------------------------------------ loader
    const int l = wid - NC;
    if (MI_SKINNY_LOADER_PRIO) __builtin_amdgcn_s_setprio(MI_SKINNY_LOADER_PRIO);
    const int drow = lane >> 3, dpc = lane & 7;
    const char* src[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      const int q = l + NL * i;                    // piece: image rows 8 q .. 8 q + 7
      const int row = 8 * q + drow;
      const int lc = dpc ^ ((row >> 1) & 7);
      if (q < XI) {
        int m = m0 + row < p.M ? m0 + row : p.M - 1;   // rows past M repeat the last one; never stored
        if constexpr (GROUPED) {   // row of the tile -> row of the source matrix (padding rows read some valid row)
          const int R = blockIdx.y * p.rpt + (row < p.rpt ? row : 0);
          if (p.gather_div > 0) {
            const int pr = p.row_map[R];
            m = (pr < 0 ? 0 : pr) / p.gather_div;
          } else {
            m = R;
          }
        }
        src[i] = reinterpret_cast<const char*>(p.x + (int64_t)m * p.ldx + 64 * (int64_t)b_lo + 8 * lc);
      } else {
        const int r = row - 16 * MT;               // row of the weight image
        int n;
        if constexpr (GATED) {
          int col = tile0 * (16 * NSLAB) + 16 * (r >> 5) + (r & 15);
          col = col < p.N ? col : p.N - 1;
          n = ((r >> 4) & 1) * p.N + col;          // rows 16-31 of a slab: the up rows of the same columns
        } else {
          n = tile0 * TR + r;
          n = n < p.N ? n : p.N - 1;               // a tile may hang over N: those row groups are never stored
        }
        src[i] = reinterpret_cast<const char*>(wbase + (int64_t)n * p.K + 64 * (int64_t)b_lo + 8 * lc);
      }
    }
    // step `it` of the workgroup = k block it % n_it of its tile it / n_it: the x pieces start over with every tile, the
    // weight pieces move on by one tile's rows (a chained launch has whole tiles only: the host checks)
    const int64_t w_tile_bytes = (int64_t)(GATED ? 16 * NSLAB : TR) * p.K * 2;
    int kk_next = 0;                 // k block of the next step to issue
    int64_t w_next = 0;              // byte offset of its tile's weight rows
    auto issue = [&](int stage) {
      char* st = smem + stage * kStage;
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        if (l + NL * i < XI)
          __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + 128 * (int64_t)kk_next),
                                           (lds_void*)(st + (l + NL * i) * 1024), 16, 0, COH ? kAuxAgent : 0);
        else
          __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + 128 * (int64_t)kk_next + w_next),
                                           (lds_void*)(st + (l + NL * i) * 1024), 16, 0, MI_SKINNY_W_AUX);
      }
      if (++kk_next == n_it) {
        kk_next = 0;
        w_next += w_tile_bytes;
      }
    };
    const int total = n_it * chain;
#pragma unroll
    for (int j = 0; j < AHEAD; ++j)
      if (j < total && !(p.dbg & 1)) issue(j);
    int stage = AHEAD;   // stage of step it + AHEAD (= the stage step it - 1 was read from)
    for (int it = 0; it < total; ++it) {
      // step `it` landed; the (up to AHEAD - 1) steps issued after it may still be in flight
      wait_steps_behind<PPL, AHEAD>(total - 1 - it);
      __builtin_amdgcn_s_barrier();
      if (it + AHEAD < total && !(p.dbg & 1)) issue(stage);
      stage = stage == NST - 1 ? 0 : stage + 1;
    }
    if constexpr (!GATED && !GROUPED) {
      if (p.S > 1 && p.cnt) {   // the consumers' in-launch reduce meets twice more (see the epilogue)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }

  // -------------------------------------------------------------------- consumer
  const int r16 = lane & 15, kg = lane >> 4;
  const int slab = wid % NSLAB, mh = wid / NSLAB;
  // fragment reads: image row 16 j + r16 -> (row >> 1) & 7 == r16 >> 1
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  uint32_t frag[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) frag[u] = lds0 + r16 * 128 + ((kperm(u, kg) ^ (r16 >> 1)) << 4);
  const uint32_t x_off = mh * (MTW * 2048), w_off = kXBytes + slab * 4096;

  // D[n][m]: lane = column m (row r16 of m-tile j), rows n = 4 kg + r of the row group
  const bool fused_reduce = !GATED && !GROUPED && p.S > 1 && p.cnt != nullptr;
  // COH: the consumer of a chain takes fp32 slabs whatever the split (one slab at S == 1)
  const bool slabs = COH ? true : p.S > 1;
  const bool part_sc1 = fused_reduce || COH;
  const __amdgpu_buffer_rsrc_t part_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.part, 0, part_sc1 ? (int)((int64_t)p.S * p.M * p.N * 4) : 0, 0x00020000);

  acc_t acc[2][MTW];
  int stage = 0;
  for (int ct = 0; ct < chain; ++ct) {   // GROUPED: the workgroup's tiles, one ring through all of them; else one pass
  const int tile = tile0 + ct;
  const int n_base = tile * TR + 32 * slab;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int j = 0; j < MTW; ++j) acc[g][j] = acc_t{0, 0, 0, 0};

  for (int it = 0; it < n_it; ++it) {
    __builtin_amdgcn_s_barrier();
    if (p.dbg & 2) continue;
    const uint32_t a0 = frag[0] + stage * kStage, a1 = frag[1] + stage * kStage;
    u32x4_t a[2][2];
    lds_read_w<0>(a0 + w_off, a1 + w_off, a);
#pragma unroll
    for (int jg = 0; jg < MTW / 4; ++jg) {
      u32x4_t b[2][4];
      if (jg == 0) lds_read_x8<0 * 8192>(a0 + x_off, a1 + x_off, b[0], b[1]);
      else if (jg == 1) lds_read_x8<1 * 8192>(a0 + x_off, a1 + x_off, b[0], b[1]);
      else if (jg == 2) lds_read_x8<2 * 8192>(a0 + x_off, a1 + x_off, b[0], b[1]);
      else lds_read_x8<3 * 8192>(a0 + x_off, a1 + x_off, b[0], b[1]);
      // eight independent accumulators between two uses of the same one
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 2; ++g) acc[g][4 * jg + i] = mma<Tag>(a[g][u], b[u][i], acc[g][4 * jg + i]);
    }
    stage = stage == NST - 1 ? 0 : stage + 1;
  }

  if (p.dbg & 8) {
    if constexpr (GROUPED) continue;
    else return;
  }
  if constexpr (GATED) {
    // acc[0] = gate, acc[1] = up of output columns n .. n + 3.  Rounding as in the unfused pair: both GEMM outputs go
    // to the 16-bit grid first, SiLU and the product in fp32, one more rounding (norm_act.hip silu_mul_kernel)
    const int n = tile * (16 * NSLAB) + 16 * slab + 4 * kg;
    if (n >= p.N) continue;
    const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      int m = m0 + 16 * (mh * MTW + j) + r16;
      if (m >= p.M) continue;
      if constexpr (GROUPED) m += blockIdx.y * p.rpt;
      float v[4], gq[4], uq[4];
      finish_quad<Tag>(p, acc[0][j], m, n, zero4, gq);
      finish_quad<Tag>(p, acc[1][j], m, p.N + n, zero4, uq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gt = E::from_bits(E::to_bits(gq[r])), up = E::from_bits(E::to_bits(uq[r]));
        v[r] = gt / (1.f + __expf(-gt)) * up;
      }
      uint2 q;
      q.x = E::pack(v[0], v[1]);
      q.y = E::pack(v[2], v[3]);
      store_out_b64<COH>(p.out + (int64_t)m * p.ldo + n, q);
    }
    continue;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int n = n_base + 16 * g + 4 * kg;
    if (n >= p.N) continue;   // N % 16 == 0: a row group is all in or all out
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.S == 1 && p.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = E::from_bits(p.bias[n + r]);
    }
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      int m = m0 + 16 * (mh * MTW + j) + r16;
      if (m >= p.M) continue;
      if constexpr (GROUPED) m += blockIdx.y * p.rpt;
      if (!slabs) {
        float v[4];
        finish_quad<Tag>(p, acc[g][j], m, n, bv, v);
        uint2 q;
        q.x = E::pack(v[0], v[1]);
        q.y = E::pack(v[2], v[3]);
        store_out_b64<COH>(p.out + (int64_t)m * p.ldo + n, q);
      } else if (part_sc1) {   // written through to the coherence point (sc1): no L2 write-back needed to publish
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[g][j]), part_rsrc,
                                               (int)((((int64_t)s * p.M + m) * p.N + n) * 4), 0, kAuxAgent);
      } else {   // fp32 sums, or the exact int32 sums of int8 operands (same 16 bytes)
        *reinterpret_cast<acc_t*>(p.part + ((int64_t)s * p.M + m) * p.N + n) = acc[g][j];
      }
    }
  }
  if constexpr (!GATED && !GROUPED && std::is_same_v<acc_t, f32x4_t>) {
    if (!fused_reduce) return;
    // ---- split-K reduce inside the launch: the slice of this (tile, row block) that arrives LAST sums the slabs ------
    // Publish: every wave's slab stores are write-through and complete (vmcnt(0)) before the workgroup's one counter
    // add; acquire: the last arriver reads the slabs with sc1 loads issued after its add has returned
    // (MI355X_MICROARCH.md, valid hand-off forms; the decode kernel's in-launch merge uses the same pair).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const uint32_t flag = (uint32_t)(uintptr_t)(lds_void*)smem;   // the ring is dead: every read retired, no DMA in flight
    if (wid == 0) {
      int32_t* c = p.cnt + (tile * p.RB + m0 / (16 * MT));
      int old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      const bool last = old == p.S - 1;
      if (last && lane == 0) __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      if (lane == 0) lds_store32(flag, last ? 1u : 0u);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (lds_poll128(flag)[0] == 0u) return;
    // slab loads of `JB` m-tiles x both row groups x all slices in flight at once, then summed in slice order
    // (skinny_reduce_kernel's arithmetic: the same bits as the separate reduce launch)
    auto reduce_batch = [&](auto jb_c, auto smax_c, int j0) {
      constexpr int JB = decltype(jb_c)::value, SM = decltype(smax_c)::value;
      f32x4_t v[2][JB][SM];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int n = n_base + 16 * g + 4 * kg;
        const int nn = n < p.N ? n : 0;
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const int m = m0 + 16 * (mh * MTW + j0 + jj) + r16;
          const int mm = m < p.M ? m : p.M - 1;
#pragma unroll
          for (int sl = 0; sl < SM; ++sl)
            if (sl < p.S)
              v[g][jj][sl] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
                  part_rsrc, (int)((((int64_t)sl * p.M + mm) * p.N + nn) * 4), 0, kAuxAgent));
        }
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int n = n_base + 16 * g + 4 * kg;
        if (n >= p.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = E::from_bits(p.bias[n + r]);
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const int m = m0 + 16 * (mh * MTW + j0 + jj) + r16;
          if (m >= p.M) continue;
          f32x4_t a = v[g][jj][0];
#pragma unroll
          for (int sl = 1; sl < SM; ++sl)
            if (sl < p.S) a += v[g][jj][sl];
          uint2 q;
          q.x = E::pack(a[0] + bv[0], a[1] + bv[1]);
          q.y = E::pack(a[2] + bv[2], a[3] + bv[3]);
          *reinterpret_cast<uint2*>(p.out + (int64_t)m * p.ldo + n) = q;
        }
      }
    };
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    if (p.S <= 4) {   // one round trip: 8 m-tile x group quads x <= 4 slices = 128 registers
#pragma unroll
      for (int j0 = 0; j0 < MTW; j0 += 4) reduce_batch(I4{}, I4{}, j0);
    } else {
#pragma unroll
      for (int j0 = 0; j0 < MTW; j0 += 2) reduce_batch(I2{}, I8{}, j0);
    }
  }
  }   // chained tiles (the in-launch reduce above belongs to the one tile of an un-chained workgroup)
}

}  // namespace
}  // namespace mi
