// Decode-batch linear layer: y[M, N] = x[M, K] . W[N, K]^T (+ bias), M <= 256 rows (one decode step's batch), and
// its gated form silu(x Wg^T) * (x Wu^T) (gate_up_proj + SiluAndMul in one launch).
//
// The reference runs these projections (qkv_proj, o_proj, gate_up_proj, down_proj of every decoder layer:
// vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:37-69,72-156 through vLLM's ColumnParallelLinear /
// RowParallelLinear) as library GEMMs.  For a decode batch the job is weight streaming: at most 256 flop per weight
// byte, under the ~310 flop/B ridge of the chip, so the bound is HBM (the weights, read once) plus the per-CU intake
// of re-reading x for every weight tile.  The library's tiles reach 1.8-3.4 TB/s of weight bytes on
// these shapes; this kernel reaches the HBM rate at <= 64 rows (6.4 TB/s) and 2.7-3.8 TB/s at 256:
//   * swapped form D = W . x^T on MFMA 16x16x32: a consumer wave owns 32 weight rows x 128 activation rows;
//   * one 64-wide k block per step; a stage of the LDS ring is ONE image of 128-byte rows — the (<= 256) rows of x, then
//     the tile's 128 / 160 weight rows — with 16-byte chunk c of row r at c ^ ((r >> 1) & 7): `ds_read_b128`
//     operand reads are conflict-free (SQ_LDS_BANK_CONFLICT = 0, profiles/r01_pmc_linear_kernel.md).  The k order
//     inside a block is permuted identically for W and x, chosen so that the two k-groups served by one
//     `ds_read_b128` lane group differ by 6 in chunk index;
//   * the image is filled by LDS-DMA in 1 KiB pieces by dedicated loader waves (weight pieces non-temporal, raised
//     priority); consumer waves only read fragments and run MFMAs; one raw s_barrier per step; three stages, two
//     steps (~100 KB per CU) in flight;
//   * split-K where N alone does not fill the chip (down_proj: 28 tiles x 8 slices = 224 workgroups, slice = workgroup
//     id % 8 so that each XCD's L2 serves one slice of x): fp32 partials to a workspace, reduced in slice order by
//     a small kernel (+ bias, convert) or inside the consumer that follows (mi_fused_add_rms_norm_splitk,
//     mi_rope_neox_and_cache_splitk) — deterministic, no atomics;
//   * gated form: a slab is 16 gate rows + the 16 matching up rows, a tile 80 (64) output columns; SiLU and the product
//     in the epilogue, rounded as the unfused pair rounds.
// Measured in situ at 256 rows: down_proj 47 us + reduce against 77 us for the TunableOp-selected
// library kernel, gate_up_proj + SiluAndMul 71 us against 81 + 7 us; which layers use it: layers.py.
#include <mutex>
#include <vector>

#include "mfma_attn_common.h"
#include "skinny_grouped.h"
#include "skinny_tile.h"

namespace mi {
namespace {

template <typename Tag, typename OT, int NSLAB, int MT, int MH, int NL, bool GATED, bool GROUPED = false, int NST = 3>
__global__ __launch_bounds__(64 * (NSLAB * MH + NL)) void skinny_gemm_kernel(const SkinnyParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  skinny_tile<Tag, OT, NSLAB, MT, MH, NL, GATED, GROUPED, NST>(p, (int)blockIdx.x, smem);
}

// ======================================================================================================= v2 (129-256 rows)
// Same data layout and LDS image as the kernel above; what changes is WHO does what and how the waves meet:
//   * 8 consumer waves + 4 loader waves = 3 waves on every SIMD (a workgroup's waves go to the SIMDs cyclically, so
//     waves c and c + 4 share one): each SIMD runs the same MFMA count per step.  The kernel above ran 10 consumers
//     + 2 loaders: two SIMDs carried 3 consumers, two carried 2 — its matrix-pipe floor was the 3-consumer SIMDs
//     (1536 cycles per step on the 160-row tile against 1280 here).
//   * a consumer owns (a half of the tile's 16-row weight groups) x (a quarter of the 16 m-tiles): 18 fragment reads
//     for 40 MFMAs per step instead of 20 for 32 (LDS read traffic per CU and step: 144 KB against 200 KB).
//     Gated form, 160-row tile (5 gate/up pairs): waves 0-3 take 3 pairs, waves 4-7 two — every SIMD gets one of
//     each, and the SiLU-gate epilogue stays lane-local.
//   * no s_barrier in the loop.  Per ring slot a FULL word per loader wave and a FREE word per consumer wave, in
//     LDS, holding step numbers (monotonic, never reset): a loader publishes FULL once its pieces of the step have
//     landed (counted vmcnt) and refills a slot once every consumer has published FREE for the step that used it; a
//     consumer publishes FREE as soon as the step's fragments are in registers — before its MFMAs.  Waves drift
//     apart instead of meeting once per step: one wave's fragment reads overlap its SIMD partner's MFMAs
//     (MI355X_MICROARCH.md, price list row ring-gemm).  Spins are bounded.
constexpr int kV2Loaders = 4, kV2Consumers = 8;
constexpr uint32_t kSpinBound = 1u << 24;

// NW weight fragments (one per 16-row group) + the 4 m-tile fragments of one k-step, and their wait, in ONE statement
template <int NW>
__device__ __forceinline__ void lds_read_wx(uint32_t aw, uint32_t ax, u32x4_t (&w)[NW], u32x4_t (&x)[4]);
#define MI_RD(i, base, off) "ds_read_b128 %" #i ", %" #base " offset:" #off "\n\t"
template <>
__device__ __forceinline__ void lds_read_wx<4>(uint32_t aw, uint32_t ax, u32x4_t (&w)[4], u32x4_t (&x)[4]) {
  asm volatile(MI_RD(0, 8, 0) MI_RD(4, 9, 0) MI_RD(1, 8, 2048) MI_RD(5, 9, 2048) MI_RD(2, 8, 4096) MI_RD(6, 9, 4096)
               MI_RD(3, 8, 6144) MI_RD(7, 9, 6144) "s_waitcnt lgkmcnt(0)"
               : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
               : "v"(aw), "v"(ax)
               : "memory");
}
template <>
__device__ __forceinline__ void lds_read_wx<5>(uint32_t aw, uint32_t ax, u32x4_t (&w)[5], u32x4_t (&x)[4]) {
  asm volatile(MI_RD(0, 9, 0) MI_RD(5, 10, 0) MI_RD(1, 9, 2048) MI_RD(6, 10, 2048) MI_RD(2, 9, 4096) MI_RD(7, 10, 4096)
               MI_RD(3, 9, 6144) MI_RD(8, 10, 6144) MI_RD(4, 9, 8192) "s_waitcnt lgkmcnt(0)"
               : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]),
                 "=&v"(x[3])
               : "v"(aw), "v"(ax)
               : "memory");
}
template <>
__device__ __forceinline__ void lds_read_wx<6>(uint32_t aw, uint32_t ax, u32x4_t (&w)[6], u32x4_t (&x)[4]) {
  asm volatile(MI_RD(0, 10, 0) MI_RD(6, 11, 0) MI_RD(1, 10, 2048) MI_RD(7, 11, 2048) MI_RD(2, 10, 4096) MI_RD(8, 11, 4096)
               MI_RD(3, 10, 6144) MI_RD(9, 11, 6144) MI_RD(4, 10, 8192) MI_RD(5, 10, 10240) "s_waitcnt lgkmcnt(0)"
               : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(x[0]), "=&v"(x[1]),
                 "=&v"(x[2]), "=&v"(x[3])
               : "v"(aw), "v"(ax)
               : "memory");
}
#undef MI_RD

// G: 16-row weight groups per tile (10 = 160 rows, 8 = 128 rows); G0: groups of consumer waves 0-3 (waves 4-7 take G - G0)
template <typename Tag, typename OT, int G, int G0, bool GATED>
struct V2 {
  static constexpr int MT = 16;
  static constexpr int TR = 16 * G;
  static constexpr int kXBytes = MT * 2048;
  static constexpr int kStage = kXBytes + TR * 128;
  static constexpr int NST = 3;
  static constexpr int XI = MT * 2, PIECES = XI + TR / 8;
  static constexpr int PPL = PIECES / kV2Loaders;
  static constexpr int kFlags = NST * kStage;                     // FULL[NST][4] then FREE[NST][8]
  static constexpr int kLds = kFlags + NST * (4 + 8) * 4;
  static_assert(PIECES % kV2Loaders == 0 && PPL <= 31, "equal shares; two steps of pieces must fit vmcnt");
  static_assert(kLds <= 160 * 1024, "the ring must fit the LDS");
  static_assert(!GATED || (G0 % 2 == 0 && G % 2 == 0), "gate/up pairs are not split between the wave halves");

  // one consumer wave: NG weight groups starting at group g_lo, m-tiles 4 mq .. 4 mq + 3
  template <int NG>
  static __device__ __forceinline__ void consume(const SkinnyParams& p, char* smem, int c, int g_lo, int mq, int tile, int s,
                                                 int n_it) {
    using E = Elem<OT>;
    using acc_t = typename AccOf<Tag>::type;
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.y * 256;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
    uint32_t frag[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) frag[u] = lds0 + r16 * 128 + ((kperm(u, kg) ^ (r16 >> 1)) << 4);
    const uint32_t x_off = mq * (4 * 2048), w_off = kXBytes + g_lo * 2048;
    const uint32_t full0 = lds0 + kFlags, free0 = lds0 + kFlags + NST * 16;
    acc_t acc[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[g][i] = acc_t{0, 0, 0, 0};
    int stage = 0;
    for (int it = 0; it < n_it; ++it) {
      // FULL: every loader's pieces of step `it` have landed
      const uint32_t want = (uint32_t)it + 1;
      for (uint32_t spin = 0; spin < kSpinBound; ++spin) {
        const u32x4_t f = lds_poll128(full0 + stage * 16);
        if (f[0] >= want && f[1] >= want && f[2] >= want && f[3] >= want) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!(p.dbg & 2)) {
        const uint32_t base = stage * kStage;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          u32x4_t a[NG], b[4];
          lds_read_wx<NG>(frag[u] + base + w_off, frag[u] + base + x_off, a, b);
          // FREE as soon as the step's last fragment is in registers: the loaders may refill the slot during the MFMAs
          if (u == 1 && lane == 0) lds_store32(free0 + (stage * 8 + c) * 4, want);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][i] = mma<Tag>(a[g], b[i], acc[g][i]);
        }
      } else if (lane == 0) {
        lds_store32(free0 + (stage * 8 + c) * 4, want);
      }
      stage = stage == NST - 1 ? 0 : stage + 1;
    }
    if (p.dbg & 8) return;
    if constexpr (GATED) {
      // groups come in (gate, up) pairs of the same 16 output columns: acc[2q] = gate, acc[2q + 1] = up
      constexpr int CT = 8 * G;   // output columns per tile
#pragma unroll
      for (int q = 0; q < NG / 2; ++q) {
        const int n = tile * CT + 8 * g_lo + 16 * q + 4 * kg;
        if (n >= p.N) continue;
        const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + 16 * (4 * mq + i) + r16;
          if (m >= p.M) continue;
          float v[4], gq[4], uq[4];
          finish_quad<Tag>(p, acc[2 * q][i], m, n, zero4, gq);
          finish_quad<Tag>(p, acc[2 * q + 1][i], m, p.N + n, zero4, uq);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gt = E::from_bits(E::to_bits(gq[r])), up = E::from_bits(E::to_bits(uq[r]));
            v[r] = gt / (1.f + __expf(-gt)) * up;
          }
          uint2 o;
          o.x = E::pack(v[0], v[1]);
          o.y = E::pack(v[2], v[3]);
          *reinterpret_cast<uint2*>(p.out + (int64_t)m * p.ldo + n) = o;
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int n = tile * TR + 16 * (g_lo + g) + 4 * kg;
        if (n >= p.N) continue;   // N % 16 == 0: a row group is all in or all out
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.S == 1 && p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = E::from_bits(p.bias[n + r]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + 16 * (4 * mq + i) + r16;
          if (m >= p.M) continue;
          if (p.S == 1) {
            float v[4];
            finish_quad<Tag>(p, acc[g][i], m, n, bv, v);
            uint2 o;
            o.x = E::pack(v[0], v[1]);
            o.y = E::pack(v[2], v[3]);
            *reinterpret_cast<uint2*>(p.out + (int64_t)m * p.ldo + n) = o;
          } else {
            *reinterpret_cast<acc_t*>(p.part + ((int64_t)s * p.M + m) * p.N + n) = acc[g][i];
          }
        }
      }
    }
  }
};

template <typename Tag, typename OT, int G, int G0, bool GATED>
__global__ __launch_bounds__(64 * (kV2Consumers + kV2Loaders)) void skinny_gemm_v2_kernel(const SkinnyParams p) {
  using C = V2<Tag, OT, G, G0, GATED>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int s = wg % p.S, tile = wg / p.S;
  const int nb = p.K >> 6;
  const int b_lo = (int)((int64_t)s * nb / p.S), b_hi = (int)((int64_t)(s + 1) * nb / p.S);
  const int n_it = b_hi - b_lo;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  // the flag words start at zero: before any DMA is in flight, so a plain barrier is harmless here
  if (tid < C::NST * 12) lds_store32(lds0 + C::kFlags + tid * 4, 0u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if (wid >= kV2Consumers) {
    // ------------------------------------------------------------------ loader
    const int l = wid - kV2Consumers;
    if (MI_SKINNY_LOADER_PRIO) __builtin_amdgcn_s_setprio(MI_SKINNY_LOADER_PRIO);
    const int drow = lane >> 3, dpc = lane & 7;
    const char* src[C::PPL];
#pragma unroll
    for (int i = 0; i < C::PPL; ++i) {
      const int q = l + kV2Loaders * i;            // piece: image rows 8 q .. 8 q + 7
      const int row = 8 * q + drow;
      const int lc = dpc ^ ((row >> 1) & 7);
      if (q < C::XI) {
        const int m0 = blockIdx.y * 256;
        const int m = m0 + row < p.M ? m0 + row : p.M - 1;   // rows past M repeat the last one; never stored
        src[i] = reinterpret_cast<const char*>(p.x + (int64_t)m * p.ldx + 64 * (int64_t)b_lo + 8 * lc);
      } else {
        const int r = row - 16 * C::MT;            // row of the weight image
        int n;
        if constexpr (GATED) {   // group 2 q = 16 gate rows, group 2 q + 1 = the up rows of the same 16 columns
          int col = tile * (8 * G) + 16 * (r >> 5) + (r & 15);
          col = col < p.N ? col : p.N - 1;
          n = ((r >> 4) & 1) * p.N + col;
        } else {
          n = tile * C::TR + r;
          n = n < p.N ? n : p.N - 1;
        }
        src[i] = reinterpret_cast<const char*>(p.w + (int64_t)n * p.K + 64 * (int64_t)b_lo + 8 * lc);
      }
    }
    auto issue = [&](int it, int stage) {
      char* st = smem + stage * C::kStage;
#pragma unroll
      for (int i = 0; i < C::PPL; ++i) {
        if (l + kV2Loaders * i < C::XI)
          __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + 128 * (int64_t)it),
                                           (lds_void*)(st + (l + kV2Loaders * i) * 1024), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + 128 * (int64_t)it),
                                           (lds_void*)(st + (l + kV2Loaders * i) * 1024), 16, 0, MI_SKINNY_W_AUX);
      }
    };
    const uint32_t full0 = lds0 + C::kFlags, free0 = lds0 + C::kFlags + C::NST * 16;
    if (n_it > 0 && !(p.dbg & 1)) issue(0, 0);
    if (n_it > 1 && !(p.dbg & 1)) issue(1, 1);
    int stage = 0;   // stage of step `it`
    for (int it = 0; it < n_it; ++it) {
      if (it + 1 < n_it) wait_vmcnt<C::PPL>();   // step `it` landed; step it + 1 may still be in flight
      else wait_vmcnt<0>();
      if (lane == 0) lds_store32(full0 + (stage * 4 + l) * 4, (uint32_t)it + 1);
      if (it + 2 < n_it) {
        // step it + 2 goes into the slot of step it - 1: every consumer must have its fragments of that step
        const int st2 = stage == 0 ? C::NST - 1 : stage - 1;
        if (it >= 1) {
          const uint32_t want = (uint32_t)it;
          for (uint32_t spin = 0; spin < kSpinBound; ++spin) {
            const u32x4_t f0 = lds_poll128(free0 + st2 * 32), f1 = lds_poll128(free0 + st2 * 32 + 16);
            if (f0[0] >= want && f0[1] >= want && f0[2] >= want && f0[3] >= want && f1[0] >= want && f1[1] >= want &&
                f1[2] >= want && f1[3] >= want)
              break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        if (!(p.dbg & 1)) issue(it + 2, st2);
      }
      stage = stage == C::NST - 1 ? 0 : stage + 1;
    }
    return;
  }
  // -------------------------------------------------------------------- consumer: (half of the groups) x (quarter of the m-tiles)
  const int c = wid, half = c >> 2, mq = c & 3;
  if (half == 0) C::template consume<G0>(p, smem, c, 0, mq, tile, s, n_it);
  else C::template consume<G - G0>(p, smem, c, G0, mq, tile, s, n_it);
}

template <typename Tag>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(uint16_t* __restrict__ out, const float* __restrict__ part,
                                                            const uint16_t* __restrict__ bias, int M, int N, int S,
                                                            int64_t ldo) {
  using E = Elem<Tag>;
  const int64_t quads = (int64_t)M * N / 4;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const int m = (int)(q / (N / 4)), n = (int)(q - (int64_t)m * (N / 4)) * 4;
  // all the slices' loads in flight at once (S <= 16: skinny_plan), then summed in slice order
  f32x4_t v[16];
#pragma unroll
  for (int s = 0; s < 16; ++s)
    if (s < S) v[s] = *reinterpret_cast<const f32x4_t*>(part + ((int64_t)s * M + m) * N + n);
  f32x4_t a = v[0];
#pragma unroll
  for (int s = 1; s < 16; ++s)
    if (s < S) a += v[s];
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] += E::from_bits(bias[n + r]);
  }
  uint2 o;
  o.x = E::pack(a[0], a[1]);
  o.y = E::pack(a[2], a[3]);
  *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n) = o;
}

// split-K reduce of int8 operands: exact int32 sums over the slices, then the dequantisation (+ bias)
template <typename OT>
__global__ __launch_bounds__(256) void skinny_reduce_i8_kernel(uint16_t* __restrict__ out, const int32_t* __restrict__ part,
                                                               const float* __restrict__ xs, const float* __restrict__ ws,
                                                               const uint16_t* __restrict__ bias, int M, int N, int S,
                                                               int64_t ldo) {
  using E = Elem<OT>;
  const int64_t quads = (int64_t)M * N / 4;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const int m = (int)(q / (N / 4)), n = (int)(q - (int64_t)m * (N / 4)) * 4;
  i32x4_t v[16];
#pragma unroll
  for (int s = 0; s < 16; ++s)
    if (s < S) v[s] = *reinterpret_cast<const i32x4_t*>(part + ((int64_t)s * M + m) * N + n);
  i32x4_t a = v[0];
#pragma unroll
  for (int s = 1; s < 16; ++s)
    if (s < S) a += v[s];
  const float x = xs[m];
  const float4 w = *reinterpret_cast<const float4*>(ws + n);
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = E::from_bits(bias[n + r]);
  }
  const float f[4] = {dequant_i32(a[0], x, w.x, bv[0]), dequant_i32(a[1], x, w.y, bv[1]), dequant_i32(a[2], x, w.z, bv[2]),
                      dequant_i32(a[3], x, w.w, bv[3])};
  uint2 o;
  o.x = E::pack(f[0], f[1]);
  o.y = E::pack(f[2], f[3]);
  *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n) = o;
}

struct SkinnyPlan {
  int nslab, mt, tiles, split;
};

constexpr int ring_stages(int mt, int nslab) {
  const int stage = mt * 2048 + nslab * 4096;
  const int n = MI_SKINNY_RING_KB * 1024 / stage;
  return n < 3 ? 3 : n > 8 ? 8 : n;
}

// Tile shape and split-K factor for one problem.  Goal: one workgroup per CU in ONE round (256 CUs), every CU busy.
//   * 160-row tiles when that lands the tile count just under 256 (gate_up_proj: 37888 / 160 = 237 tiles), else
//     128-row tiles with K cut into S slices so that tiles x S <= 256;
//   * S is capped so the fp32 partials stay a fraction of the weight bytes and every slice keeps >= 4 k-blocks.
// ---- tile plans registered by the host (mi_linear_plan_set): measured per deployment shape, like the TunableOp
// selections of the library GEMMs.  Tile shapes that are built (launch_skinny): 16-bit operands only.
constexpr bool plan_built(int mt, int ns, bool gated) {
  if (mt == 4) return gated ? (ns >= 1 && ns <= 6) : ((ns >= 1 && ns <= 6) || ns == 8);
  if (mt == 8) return ns >= 1 && ns <= 5;
  if (mt == 16) return gated ? (ns == 1 || ns == 2 || ns == 4 || ns == 5) : (ns == 4 || ns == 5);
  return false;
}
struct PlanEntry {
  int purpose, m_lo, m_hi, n, k, mt, nslab, split;
};
std::mutex g_plan_mu;
std::vector<PlanEntry> g_plans;

void plan_fill(SkinnyPlan* pl, int mt, int ns, int split, int n_rows, bool gated) {
  pl->mt = mt; pl->nslab = ns; pl->split = gated ? 1 : split;
  const int cols = gated ? 16 * ns : 32 * ns;
  pl->tiles = (n_rows + cols - 1) / cols;
}

// k == 0 matches any k (the gated entry point's plan does not depend on it)
bool plan_lookup(int purpose, int m, int n, int k, SkinnyPlan* pl) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  for (const PlanEntry& e : g_plans) {
    if (e.purpose != purpose || e.n != n || (k && e.k != k) || m <= e.m_lo || m > e.m_hi) continue;
    plan_fill(pl, e.mt, e.nslab, e.split <= k / 64 || !k ? e.split : 1, n, purpose == MI_LINEAR_PLAN_GATED);
    return true;
  }
  return false;
}

// MI_LINEAR_PLAN=<mt>,<nslab>,<split> / MI_GATED_PLAN=<mt>,<nslab>: tuning overrides, read per call (tools/bench_shard_gemm.py)
bool plan_override(const char* var, int n_rows, bool gated, int k, SkinnyPlan* pl) {
  const char* e = getenv(var);
  if (!e || !*e) return false;
  int mt = 0, ns = 0, sp = 1;
  if (sscanf(e, "%d,%d,%d", &mt, &ns, &sp) < 2) return false;
  if (!plan_built(mt, ns, gated) || sp < 1 || sp > 16) return false;
  plan_fill(pl, mt, ns, sp <= k / 64 ? sp : 1, n_rows, gated);
  return true;
}

SkinnyPlan skinny_plan(int m, int n, int k, int force_split) {
  SkinnyPlan pl;
  pl.mt = m <= 64 ? 4 : m <= 128 ? 8 : 16;
  const int nb = k / 64;
  const int t160 = (n + 159) / 160, t128 = (n + 127) / 128;
  if (t160 > 128 && t160 <= 256 && force_split <= 1) {
    pl.nslab = 5; pl.tiles = t160; pl.split = 1;
    return pl;
  }
  pl.nslab = 4; pl.tiles = t128;
  int s = 256 / t128;
  if (s < 1) s = 1;
  if (s > nb / 4) s = nb / 4 > 0 ? nb / 4 : 1;
  if (s > 16) s = 16;
  if (force_split > 0) s = force_split < 16 ? force_split : 16;
  pl.split = s;
  return pl;
}

}  // namespace
}  // namespace mi

namespace mi {
namespace {

int linear_v2() {   // MI_LINEAR_V2=0: the first-generation kernel at 129-256 rows too (A/B runs; read per call)
  const char* e = getenv("MI_LINEAR_V2");
  return e ? atoi(e) : 1;
}

template <typename Tag, typename OT, int G, int G0, bool GATED>
int launch_v2(const SkinnyParams& p, const dim3 grid, hipStream_t s) {
  using C = V2<Tag, OT, G, G0, GATED>;
  auto kern = skinny_gemm_v2_kernel<Tag, OT, G, G0, GATED>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::kLds);
  if (attr != hipSuccess) return MI_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(64 * (kV2Consumers + kV2Loaders)), C::kLds, s, p);
  return launch_status();
}

// `dtype`: the type of the 16-bit OUTPUT (and of the operands, unless `int8`: then x / w are int8 rows and the sums
// int32).  A launch covers ceil(M / (16 mt)) row blocks (grid.y): M > 256 only arises with int8 operands.
int launch_skinny(const SkinnyParams& p_in, const SkinnyPlan& pl, bool gated, int dtype, hipStream_t s, bool int8 = false) {
  SkinnyParams p = p_in;
  p.RB = (p.M + 16 * pl.mt - 1) / (16 * pl.mt);
  p.U = pl.tiles * pl.split;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (pl.mt == 16 && pl.nslab >= 4 && linear_v2()) {
    const dim3 grid(pl.tiles * pl.split, p.RB);   // second generation: one 256-row block per grid.y
#define MI_V2(TAG, OT)                                                                                                   \
    do {                                                                                                                 \
      if (pl.nslab == 5) return gated ? launch_v2<TAG, OT, 10, 6, true>(p, grid, s) : launch_v2<TAG, OT, 10, 5, false>(p, grid, s); \
      return gated ? launch_v2<TAG, OT, 8, 4, true>(p, grid, s) : launch_v2<TAG, OT, 8, 4, false>(p, grid, s);                     \
    } while (0)
    if (int8) {
      if (dtype == MI_DTYPE_BF16) MI_V2(i8_tag, bf16_tag);
      else MI_V2(i8_tag, f16_tag);
    } else {
      if (dtype == MI_DTYPE_BF16) MI_V2(bf16_tag, bf16_tag);
      else MI_V2(f16_tag, f16_tag);
    }
#undef MI_V2
  }
  const dim3 grid((p.U + 7) / 8 * 8 * p.RB);   // first generation: row blocks folded into the id (see the kernel)
#define MI_SKINNY(TAG, OT, NSLAB, MT, MH, NL, GATED)                                                              \
  do {                                                                                                            \
    constexpr int nst = ring_stages(MT, NSLAB);                                                                   \
    auto kern = skinny_gemm_kernel<TAG, OT, NSLAB, MT, MH, NL, GATED, false, nst>;                                \
    constexpr int lds = nst * (MT * 2048 + NSLAB * 4096);                                                         \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);          \
    if (attr != hipSuccess) return MI_ERR_LAUNCH;                                                                 \
    hipLaunchKernelGGL(kern, grid, dim3(64 * (NSLAB * MH + NL)), lds, s, p);                                      \
  } while (0)
#define MI_SKINNY_T(TAG, OT, GATED)                              \
  do {                                                           \
    if (pl.nslab == 5) {                                         \
      if (pl.mt == 16) MI_SKINNY(TAG, OT, 5, 16, 2, MI_SKINNY_NL5, GATED);   \
      else if (pl.mt == 8) MI_SKINNY(TAG, OT, 5, 8, MI_SKINNY_MH8, MI_SKINNY_NL5, GATED);\
      else MI_SKINNY(TAG, OT, 5, 4, 1, MI_SKINNY_NL5, GATED);                \
    } else {                                                     \
      if (pl.mt == 16) MI_SKINNY(TAG, OT, 4, 16, 2, 4, GATED);   \
      else if (pl.mt == 8) MI_SKINNY(TAG, OT, 4, 8, MI_SKINNY_MH8, 4, GATED);\
      else MI_SKINNY(TAG, OT, 4, 4, 1, 4, GATED);                \
    }                                                            \
  } while (0)
  // gated form, narrow tiles (16 / 32 output columns): a projection whose 64-column tiles would leave most CUs idle
  // (no split-K here: the activation needs the whole sum) — TP shards of the talker's gate_up_proj, the code predictor's
#define MI_SKINNY_NARROW(TAG, OT, NSLAB)                         \
  do {                                                           \
    if (pl.mt == 16) MI_SKINNY(TAG, OT, NSLAB, 16, 2, 4, true);  \
    else if (pl.mt == 8) MI_SKINNY(TAG, OT, NSLAB, 8, 2, 4, true);\
    else MI_SKINNY(TAG, OT, NSLAB, 4, 1, 4, true);               \
  } while (0)
#define MI_SKINNY_G(TAG, OT)                                     \
  do {                                                           \
    if (gated && pl.nslab == 1) MI_SKINNY_NARROW(TAG, OT, 1);    \
    else if (gated && pl.nslab == 2) MI_SKINNY_NARROW(TAG, OT, 2);\
    else if (gated) MI_SKINNY_T(TAG, OT, true);                  \
    else MI_SKINNY_T(TAG, OT, false);                            \
  } while (0)
  // shard shapes (16-bit operands): 64-row blocks with 32 to 256 weight rows per tile, 128-row blocks with 32 to 96 — the
  // tiles that balance a workgroup's x and weight intake when a TP shard has 64-256 rows (shard_plan)
#define MI_SKINNY_SHARD(TAG, OT)                                                         \
  do {                                                                                   \
    if (pl.mt == 4) {                                                                    \
      if (pl.nslab == 1 && !gated) { MI_SKINNY(TAG, OT, 1, 4, 1, 4, false); return launch_status(); }   \
      if (pl.nslab == 2 && !gated) { MI_SKINNY(TAG, OT, 2, 4, 1, 4, false); return launch_status(); }   \
      if (pl.nslab == 3 && !gated) { MI_SKINNY(TAG, OT, 3, 4, 1, 4, false); return launch_status(); }   \
      if (pl.nslab == 3 && gated) { MI_SKINNY(TAG, OT, 3, 4, 1, 4, true); return launch_status(); }     \
      if (pl.nslab == 6 && !gated) { MI_SKINNY(TAG, OT, 6, 4, 1, 4, false); return launch_status(); }   \
      if (pl.nslab == 6 && gated) { MI_SKINNY(TAG, OT, 6, 4, 1, 4, true); return launch_status(); }     \
      if (pl.nslab == 8 && !gated) { MI_SKINNY(TAG, OT, 8, 4, 1, 4, false); return launch_status(); }   \
    } else if (pl.mt == 8) {                                                             \
      if (pl.nslab == 1 && !gated) { MI_SKINNY(TAG, OT, 1, 8, 2, 4, false); return launch_status(); }   \
      if (pl.nslab == 2 && !gated) { MI_SKINNY(TAG, OT, 2, 8, 2, 4, false); return launch_status(); }   \
      if (pl.nslab == 3 && !gated) { MI_SKINNY(TAG, OT, 3, 8, 2, 4, false); return launch_status(); }   \
      if (pl.nslab == 3 && gated) { MI_SKINNY(TAG, OT, 3, 8, 2, 4, true); return launch_status(); }     \
    }                                                                                    \
  } while (0)
  if (!int8) {
    if (dtype == MI_DTYPE_BF16) MI_SKINNY_SHARD(bf16_tag, bf16_tag);
    else MI_SKINNY_SHARD(f16_tag, f16_tag);
  }
#undef MI_SKINNY_SHARD
  if (int8) {
    if (dtype == MI_DTYPE_BF16) MI_SKINNY_G(i8_tag, bf16_tag);
    else MI_SKINNY_G(i8_tag, f16_tag);
  } else {
    if (dtype == MI_DTYPE_BF16) MI_SKINNY_G(bf16_tag, bf16_tag);
    else MI_SKINNY_G(f16_tag, f16_tag);
  }
#undef MI_SKINNY_G
#undef MI_SKINNY_NARROW
#undef MI_SKINNY_T
#undef MI_SKINNY
  return launch_status();
}

// MT: 16-row m-tiles of a tile's row image: 4 (<= 64 token rows per tile: decode batches), 8 / 16 (128 / 256 rows: prefill
// chunks, where an expert sees hundreds of rows and a 64-row tile would stream its weights 8-16 times)
template <typename Tag, bool GATED, int MT>
int launch_grouped_t(const SkinnyParams& p, const dim3 grid, hipStream_t s) {
  constexpr int MH = MT == 4 ? 1 : 2;
  constexpr int nst = MT == 4 ? MI_SKINNY_GROUPED_NST : 3;
  auto kern = skinny_gemm_kernel<Tag, Tag, 4, MT, MH, 4, GATED, true, nst>;
  constexpr int lds = nst * (MT * 2048 + 4 * 4096);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) return MI_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(64 * (4 * MH + 4)), lds, s, p);
  return launch_status();
}
template <typename Tag, bool GATED>
int launch_grouped_rows(const SkinnyParams& p, const dim3 grid, int rpt, hipStream_t s) {
  if (rpt <= 64) return launch_grouped_t<Tag, GATED, 4>(p, grid, s);
  if (rpt <= 128) return launch_grouped_t<Tag, GATED, 8>(p, grid, s);
  return launch_grouped_t<Tag, GATED, 16>(p, grid, s);
}

// Tile of the gated form: 80 output columns when that lands the tile count just under 256, else 64 — unless that leaves
// fewer than half of the CUs with a workgroup, then 32- or 16-column tiles (MI_GATED_NSLAB=<1|2|4|5> overrides: tuning).
SkinnyPlan gated_plan(int m, int inter) {
  SkinnyPlan pl;
  if (plan_override("MI_GATED_PLAN", inter, true, 0, &pl)) return pl;
  if (plan_lookup(MI_LINEAR_PLAN_GATED, m, inter, 0, &pl)) return pl;
  const int mm = m > 256 ? 256 : m;
  pl.mt = mm <= 64 ? 4 : mm <= 128 ? 8 : 16;   // 257-512 rows: two row blocks of 16 m-tiles (grid.y)
  pl.split = 1;
  const int t80 = (inter + 79) / 80, t64 = (inter + 63) / 64, t32 = (inter + 31) / 32, t16 = (inter + 15) / 16;
  static const int force = [] {
    const char* e = getenv("MI_GATED_NSLAB");
    return e ? atoi(e) : 0;
  }();
  int ns;
  if (force == 1 || force == 2 || force == 4 || force == 5) ns = force;
  else if (t80 > 128 && t80 <= 256) ns = 5;
  else if (t64 >= 128) ns = 4;     // measured (profiles/r03r_gated_narrow.log): I 9472 at 64-256 rows: 64-column tiles
  else if (t32 >= 128) ns = 2;     // I 4736 at 128 rows: 38.0 -> 29.6 us (library pair 30.3)
  else ns = 1;                     // I 3072 at 32 rows (the code predictor's MLP): 8.92 -> 8.70 ms per call
  pl.nslab = ns;
  pl.tiles = ns == 5 ? t80 : ns == 4 ? t64 : ns == 2 ? t32 : t16;
  return pl;
}

int linear_dbg() {   // MI_LINEAR_DBG: timing experiments only (see SkinnyParams::dbg); read per call
  const char* e = getenv("MI_LINEAR_DBG");
  return e ? atoi(e) : 0;
}

}  // namespace
}  // namespace mi

namespace {
int forced_split() {   // MI_LINEAR_SPLIT=<S>: tuning override of the split-K factor (tools/bench_skinny_gemm.py)
  static const int v = [] {
    const char* e = getenv("MI_LINEAR_SPLIT");
    return e ? atoi(e) : 0;
  }();
  return v;
}
}  // namespace

namespace {
// 16-bit plan for up to 512 rows: 257-512 rows run as two row blocks per weight tile (grid.y), with the split halved so
// the grid still fits one round — the same arrangement as the W8A8 entry points below
mi::SkinnyPlan linear_plan(int m, int n, int k, int purpose = MI_LINEAR_PLAN_FULL) {
  mi::SkinnyPlan reg;
  if (mi::plan_override("MI_LINEAR_PLAN", n, false, k, &reg)) return reg;
  if (purpose == MI_LINEAR_PLAN_PARTIALS && mi::plan_lookup(MI_LINEAR_PLAN_PARTIALS, m, n, k, &reg)) return reg;
  if (mi::plan_lookup(MI_LINEAR_PLAN_FULL, m, n, k, &reg)) return reg;
  mi::SkinnyPlan pl = mi::skinny_plan(m > 256 ? 256 : m, n, k, forced_split());
  if (m > 256 && forced_split() <= 0) pl.split = pl.split > 1 ? (pl.split + 1) / 2 : 1;
  return pl;
}
}  // namespace

extern "C" int64_t mi_linear_workspace_bytes(int32_t m, int32_t n, int32_t k) {
  using namespace mi;
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const SkinnyPlan pl = linear_plan(m, n, k), pp = linear_plan(m, n, k, MI_LINEAR_PLAN_PARTIALS);
  const int split = pl.split > pp.split ? pl.split : pp.split;   // one workspace serves both entry points
  return split > 1 ? (int64_t)split * m * n * (int64_t)sizeof(float) : 0;
}

extern "C" int mi_linear_plan_set(int32_t purpose, int32_t m_lo, int32_t m_hi, int32_t n, int32_t k, int32_t rows_per_block,
                                  int32_t weight_rows_per_tile, int32_t split) {
  using namespace mi;
  if (purpose < MI_LINEAR_PLAN_FULL || purpose > MI_LINEAR_PLAN_GATED || m_lo < 0 || m_hi <= m_lo || n <= 0 || k < 0) return MI_ERR_INVALID_ARG;
  if (purpose != MI_LINEAR_PLAN_GATED && k <= 0) return MI_ERR_INVALID_ARG;
  if (rows_per_block % 64 || weight_rows_per_tile % 32 || split < 1 || split > 16) return MI_ERR_INVALID_ARG;
  const bool gated = purpose == MI_LINEAR_PLAN_GATED;
  const int mt = rows_per_block / 16, ns = weight_rows_per_tile / 32;
  if (!plan_built(mt, ns, gated)) return MI_ERR_UNSUPPORTED;
  if (gated && split != 1) return MI_ERR_INVALID_ARG;   // the activation needs the whole sum
  if (purpose == MI_LINEAR_PLAN_PARTIALS && split < 2) return MI_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_plan_mu);
  for (PlanEntry& e : g_plans)
    if (e.purpose == purpose && e.n == n && e.k == k && e.m_lo == m_lo && e.m_hi == m_hi) {
      e.mt = mt; e.nslab = ns; e.split = split;
      return MI_OK;
    }
  g_plans.push_back(PlanEntry{purpose, m_lo, m_hi, n, k, mt, ns, split});
  return MI_OK;
}

extern "C" void mi_linear_plan_clear(void) {
  std::lock_guard<std::mutex> lk(mi::g_plan_mu);
  mi::g_plans.clear();
}

extern "C" int mi_linear_plan_get(int32_t purpose, int32_t m, int32_t n, int32_t k, int32_t* out) {
  using namespace mi;
  if (!out || m <= 0 || n <= 0 || purpose < MI_LINEAR_PLAN_FULL || purpose > MI_LINEAR_PLAN_GATED) return MI_ERR_INVALID_ARG;
  const SkinnyPlan pl = purpose == MI_LINEAR_PLAN_GATED ? gated_plan(m, n) : linear_plan(m, n, k, purpose);
  out[0] = 16 * pl.mt; out[1] = 32 * pl.nslab; out[2] = pl.split;
  return MI_OK;
}

extern "C" int mi_linear(void* out, const void* x, const void* weight, const void* bias, int32_t m, int32_t n, int32_t k,
                         int64_t x_row_stride, int64_t out_row_stride, void* workspace, int64_t workspace_bytes,
                         int dtype, void* stream) {
  using namespace mi;
  if (m == 0 || n == 0) return MI_OK;
  if (!out || !x || !weight || m < 0 || n < 0 || k <= 0) return MI_ERR_INVALID_ARG;
  if (m > 512 || k % 64 || n % 16 || x_row_stride % 8 || out_row_stride % 4) return MI_ERR_UNSUPPORTED;
  if ((int64_t)m * x_row_stride * 2 >= ((int64_t)1 << 31)) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)weight) & 15 || ((uintptr_t)out & 7)) return MI_ERR_INVALID_ARG;
  const SkinnyPlan pl = linear_plan(m, n, k);
  if (pl.split > 1 && (!workspace || ((uintptr_t)workspace & 15))) return MI_ERR_INVALID_ARG;
  if (pl.split > 1 && workspace_bytes < (int64_t)pl.split * m * n * (int64_t)sizeof(float)) return MI_ERR_WORKSPACE;
  SkinnyParams p;
  p.x = reinterpret_cast<const uint16_t*>(x);
  p.w = reinterpret_cast<const uint16_t*>(weight);
  p.bias = reinterpret_cast<const uint16_t*>(bias);
  p.out = reinterpret_cast<uint16_t*>(out);
  p.part = reinterpret_cast<float*>(workspace);
  p.xs = p.ws = nullptr;
  p.M = m; p.N = n; p.K = k; p.S = pl.split;
  p.dbg = linear_dbg();
  p.ldx = x_row_stride;
  p.ldo = out_row_stride;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int lrc = launch_skinny(p, pl, false, dtype, s);
  if (lrc != MI_OK) return lrc;
  if (pl.split > 1 && !(p.dbg & 4)) {
    const int64_t quads = (int64_t)m * n / 4;
    const dim3 rgrid((unsigned)((quads + 255) / 256));
    auto o = reinterpret_cast<uint16_t*>(out);
    auto b = reinterpret_cast<const uint16_t*>(bias);
    if (dtype == MI_DTYPE_BF16)
      hipLaunchKernelGGL((skinny_reduce_kernel<bf16_tag>), rgrid, dim3(256), 0, s, o, p.part, b, m, n, pl.split, out_row_stride);
    else
      hipLaunchKernelGGL((skinny_reduce_kernel<f16_tag>), rgrid, dim3(256), 0, s, o, p.part, b, m, n, pl.split, out_row_stride);
  }
  return launch_status();
}

// mi_linear with the split-K reduce inside the launch (first-generation kernel, <= 8 slices): the slice of a (tile, row
// block) that arrives last sums the slabs in slice order — the bits of mi_linear, one launch less.  `counters`: >=
// MI_LINEAR_COUNTER_BYTES of int32, zero before the first launch (every launch leaves them zero).
extern "C" int mi_linear_fused_reduce(void* out, const void* x, const void* weight, const void* bias, int32_t m, int32_t n,
                                      int32_t k, int64_t x_row_stride, int64_t out_row_stride, void* workspace,
                                      int64_t workspace_bytes, void* counters, int64_t counters_bytes, int dtype,
                                      void* stream) {
  using namespace mi;
  if (m == 0 || n == 0) return MI_OK;
  if (!out || !x || !weight || m < 0 || n < 0 || k <= 0) return MI_ERR_INVALID_ARG;
  if (m > 512 || k % 64 || n % 16 || x_row_stride % 8 || out_row_stride % 4) return MI_ERR_UNSUPPORTED;
  if ((int64_t)m * x_row_stride * 2 >= ((int64_t)1 << 31)) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)weight) & 15 || ((uintptr_t)out & 7)) return MI_ERR_INVALID_ARG;
  const SkinnyPlan pl = linear_plan(m, n, k);
  const int rb = (m + 16 * pl.mt - 1) / (16 * pl.mt);
  const bool in_launch = pl.split > 1 && pl.split <= 8 && !(pl.mt == 16 && pl.nslab >= 4 && linear_v2()) && !linear_dbg() &&
                         (int64_t)pl.split * m * n * 4 < ((int64_t)1 << 31);
  if (!in_launch)   // no split, or a plan the in-launch reduce is not built for: the two launches
    return mi_linear(out, x, weight, bias, m, n, k, x_row_stride, out_row_stride, workspace, workspace_bytes, dtype, stream);
  if (!workspace || ((uintptr_t)workspace & 15) || !counters || ((uintptr_t)counters & 3)) return MI_ERR_INVALID_ARG;
  if (workspace_bytes < (int64_t)pl.split * m * n * (int64_t)sizeof(float)) return MI_ERR_WORKSPACE;
  if (counters_bytes < MI_LINEAR_COUNTER_BYTES || (int64_t)pl.tiles * rb * 4 > MI_LINEAR_COUNTER_BYTES) return MI_ERR_WORKSPACE;
  SkinnyParams p;
  p.x = reinterpret_cast<const uint16_t*>(x);
  p.w = reinterpret_cast<const uint16_t*>(weight);
  p.bias = reinterpret_cast<const uint16_t*>(bias);
  p.out = reinterpret_cast<uint16_t*>(out);
  p.part = reinterpret_cast<float*>(workspace);
  p.xs = p.ws = nullptr;
  p.M = m; p.N = n; p.K = k; p.S = pl.split;
  p.dbg = 0;
  p.ldx = x_row_stride;
  p.ldo = out_row_stride;
  p.cnt = reinterpret_cast<int32_t*>(counters);
  return launch_skinny(p, pl, false, dtype, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mi_linear_partials(const void* x, const void* weight, int32_t m, int32_t n, int32_t k,
                                  int64_t x_row_stride, void* workspace, int64_t workspace_bytes, int32_t* split_out,
                                  int dtype, void* stream) {
  using namespace mi;
  if (!x || !weight || !split_out || m <= 0 || n <= 0 || k <= 0) return MI_ERR_INVALID_ARG;
  if (m > 512 || k % 64 || n % 16 || x_row_stride % 8) return MI_ERR_UNSUPPORTED;
  if ((int64_t)m * x_row_stride * 2 >= ((int64_t)1 << 31)) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)weight) & 15) return MI_ERR_INVALID_ARG;
  const SkinnyPlan pl = linear_plan(m, n, k, MI_LINEAR_PLAN_PARTIALS);
  if (pl.split <= 1) return MI_ERR_UNSUPPORTED;   // nothing to defer: mi_linear writes the output directly
  if (!workspace || ((uintptr_t)workspace & 15)) return MI_ERR_INVALID_ARG;
  if (workspace_bytes < (int64_t)pl.split * m * n * (int64_t)sizeof(float)) return MI_ERR_WORKSPACE;
  SkinnyParams p;
  p.x = reinterpret_cast<const uint16_t*>(x);
  p.w = reinterpret_cast<const uint16_t*>(weight);
  p.bias = nullptr;
  p.out = nullptr;
  p.part = reinterpret_cast<float*>(workspace);
  p.xs = p.ws = nullptr;
  p.M = m; p.N = n; p.K = k; p.S = pl.split;
  p.dbg = linear_dbg();
  p.ldx = x_row_stride;
  p.ldo = 0;
  *split_out = pl.split;
  return launch_skinny(p, pl, false, dtype, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mi_linear_reduce(void* out, const void* partials, int32_t split, const void* bias, int32_t m, int32_t n,
                                int64_t out_row_stride, int dtype, void* stream) {
  using namespace mi;
  if (m == 0 || n == 0) return MI_OK;
  if (!out || !partials || split < 1 || split > 16 || m < 0 || n < 0) return MI_ERR_INVALID_ARG;
  if (n % 4 || out_row_stride % 4 || (((uintptr_t)out & 7) | ((uintptr_t)partials & 15))) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t quads = (int64_t)m * n / 4;
  const dim3 rgrid((unsigned)((quads + 255) / 256));
  auto o = reinterpret_cast<uint16_t*>(out);
  auto pp = reinterpret_cast<const float*>(partials);
  auto b = reinterpret_cast<const uint16_t*>(bias);
  if (dtype == MI_DTYPE_BF16)
    hipLaunchKernelGGL((skinny_reduce_kernel<bf16_tag>), rgrid, dim3(256), 0, s, o, pp, b, m, n, split, out_row_stride);
  else if (dtype == MI_DTYPE_F16)
    hipLaunchKernelGGL((skinny_reduce_kernel<f16_tag>), rgrid, dim3(256), 0, s, o, pp, b, m, n, split, out_row_stride);
  else
    return MI_ERR_UNSUPPORTED;
  return launch_status();
}

extern "C" int mi_linear_silu_mul(void* out, const void* x, const void* weight, int32_t m, int32_t inter, int32_t k,
                                  int64_t x_row_stride, int64_t out_row_stride, int dtype, void* stream) {
  using namespace mi;
  if (m == 0 || inter == 0) return MI_OK;
  if (!out || !x || !weight || m < 0 || inter < 0 || k <= 0) return MI_ERR_INVALID_ARG;
  if (m > 512 || k % 64 || inter % 16 || x_row_stride % 8 || out_row_stride % 4) return MI_ERR_UNSUPPORTED;
  if ((int64_t)m * x_row_stride * 2 >= ((int64_t)1 << 31)) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)weight) & 15 || ((uintptr_t)out & 7)) return MI_ERR_INVALID_ARG;
  // a tile = 16 NSLAB output columns (gate + up rows of each); no split-K: the activation needs the whole sum
  SkinnyPlan pl = gated_plan(m, inter);
  SkinnyParams p;
  p.x = reinterpret_cast<const uint16_t*>(x);
  p.w = reinterpret_cast<const uint16_t*>(weight);
  p.bias = nullptr;
  p.out = reinterpret_cast<uint16_t*>(out);
  p.part = nullptr;
  p.xs = p.ws = nullptr;
  p.M = m; p.N = inter; p.K = k; p.S = 1;
  p.dbg = linear_dbg();
  p.ldx = x_row_stride;
  p.ldo = out_row_stride;
  return launch_skinny(p, pl, true, dtype, reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------ W8A8 entry points
namespace {
// int8 rows as the kernels see them: K and row strides in 2-byte units
int int8_shape_ok(int32_t m, int32_t n, int32_t k, int64_t x_row_stride) {
  if (m > 512 || k % 128 || n % 16 || x_row_stride % 16) return 0;
  if ((int64_t)m * x_row_stride >= ((int64_t)1 << 31)) return 0;
  return 1;
}
}  // namespace

extern "C" int64_t mi_int8_linear_workspace_bytes(int32_t m, int32_t n, int32_t k) {
  using namespace mi;
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const SkinnyPlan pl = skinny_plan(m > 256 ? 256 : m, n, k / 2, forced_split());
  if ((int64_t)n * k < ((int64_t)8 << 20)) return 0;
  return pl.split > 1 ? (int64_t)pl.split * m * n * (int64_t)sizeof(int32_t) : 0;
}

extern "C" int mi_int8_linear(void* out, const int8_t* xq, const float* x_scale, const int8_t* wq, const float* w_scale,
                              const void* bias, int32_t m, int32_t n, int32_t k, int64_t x_row_stride,
                              int64_t out_row_stride, void* workspace, int64_t workspace_bytes, int dtype, void* stream) {
  using namespace mi;
  if (m == 0 || n == 0) return MI_OK;
  if (!out || !xq || !x_scale || !wq || !w_scale || m < 0 || n < 0 || k <= 0) return MI_ERR_INVALID_ARG;
  if (!int8_shape_ok(m, n, k, x_row_stride) || out_row_stride % 4) return MI_ERR_UNSUPPORTED;
  if ((((uintptr_t)xq | (uintptr_t)wq | (uintptr_t)w_scale) & 15) || ((uintptr_t)out & 7)) return MI_ERR_INVALID_ARG;
  SkinnyPlan pl = skinny_plan(m > 256 ? 256 : m, n, k / 2, forced_split());
  if (m > 256) {   // two row blocks per weight tile: halve the split so the grid still fits one round
    pl.split = pl.split > 1 ? (pl.split + 1) / 2 : 1;
  }
  // a few MB of weights stream in microseconds on a fraction of the chip: a split-K reduce launch costs more than it saves
  if ((int64_t)n * k < ((int64_t)8 << 20)) pl.split = 1;
  if (pl.split > 1 && (!workspace || ((uintptr_t)workspace & 15))) return MI_ERR_INVALID_ARG;
  if (pl.split > 1 && workspace_bytes < (int64_t)pl.split * m * n * (int64_t)sizeof(int32_t)) return MI_ERR_WORKSPACE;
  SkinnyParams p;
  p.x = reinterpret_cast<const uint16_t*>(xq);
  p.w = reinterpret_cast<const uint16_t*>(wq);
  p.bias = reinterpret_cast<const uint16_t*>(bias);
  p.out = reinterpret_cast<uint16_t*>(out);
  p.part = reinterpret_cast<float*>(workspace);
  p.xs = x_scale;
  p.ws = w_scale;
  p.M = m; p.N = n; p.K = k / 2; p.S = pl.split;
  p.dbg = linear_dbg();
  p.ldx = x_row_stride / 2;
  p.ldo = out_row_stride;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int lrc = launch_skinny(p, pl, false, dtype, s, true);
  if (lrc != MI_OK) return lrc;
  if (pl.split > 1) {
    const int64_t quads = (int64_t)m * n / 4;
    const dim3 rgrid((unsigned)((quads + 255) / 256));
    auto o = reinterpret_cast<uint16_t*>(out);
    auto b = reinterpret_cast<const uint16_t*>(bias);
    auto pp = reinterpret_cast<const int32_t*>(workspace);
    if (dtype == MI_DTYPE_BF16)
      hipLaunchKernelGGL((skinny_reduce_i8_kernel<bf16_tag>), rgrid, dim3(256), 0, s, o, pp, x_scale, w_scale, b, m, n, pl.split, out_row_stride);
    else
      hipLaunchKernelGGL((skinny_reduce_i8_kernel<f16_tag>), rgrid, dim3(256), 0, s, o, pp, x_scale, w_scale, b, m, n, pl.split, out_row_stride);
  }
  return launch_status();
}

// The W8A8 kernel without its reduce launch: exact int32 partials [split][m][n] in `workspace` for a consumer that
// folds the reduction + dequantisation in (mi_fused_add_rms_norm_splitk_int8).  MI_ERR_UNSUPPORTED when the shape's
// plan has no split (use mi_int8_linear).
extern "C" int mi_int8_linear_partials(const int8_t* xq, const int8_t* wq, int32_t m, int32_t n, int32_t k,
                                       int64_t x_row_stride, void* workspace, int64_t workspace_bytes, int32_t* split_out,
                                       void* stream) {
  using namespace mi;
  if (!xq || !wq || !split_out || m <= 0 || n <= 0 || k <= 0) return MI_ERR_INVALID_ARG;
  if (!int8_shape_ok(m, n, k, x_row_stride)) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)xq | (uintptr_t)wq) & 15) return MI_ERR_INVALID_ARG;
  SkinnyPlan pl = skinny_plan(m > 256 ? 256 : m, n, k / 2, forced_split());
  if (m > 256) pl.split = pl.split > 1 ? (pl.split + 1) / 2 : 1;
  if ((int64_t)n * k < ((int64_t)8 << 20)) pl.split = 1;
  if (pl.split <= 1) return MI_ERR_UNSUPPORTED;
  if (!workspace || ((uintptr_t)workspace & 15)) return MI_ERR_INVALID_ARG;
  if (workspace_bytes < (int64_t)pl.split * m * n * (int64_t)sizeof(int32_t)) return MI_ERR_WORKSPACE;
  SkinnyParams p;
  p.x = reinterpret_cast<const uint16_t*>(xq);
  p.w = reinterpret_cast<const uint16_t*>(wq);
  p.bias = nullptr;
  p.out = nullptr;
  p.part = reinterpret_cast<float*>(workspace);
  p.xs = p.ws = nullptr;
  p.M = m; p.N = n; p.K = k / 2; p.S = pl.split;
  p.dbg = linear_dbg();
  p.ldx = x_row_stride / 2;
  p.ldo = 0;
  *split_out = pl.split;
  return launch_skinny(p, pl, false, MI_DTYPE_BF16, reinterpret_cast<hipStream_t>(stream), true);
}

extern "C" int mi_int8_linear_silu_mul(void* out, const int8_t* xq, const float* x_scale, const int8_t* wq,
                                       const float* w_scale, int32_t m, int32_t inter, int32_t k, int64_t x_row_stride,
                                       int64_t out_row_stride, int dtype, void* stream) {
  using namespace mi;
  if (m == 0 || inter == 0) return MI_OK;
  if (!out || !xq || !x_scale || !wq || !w_scale || m < 0 || inter < 0 || k <= 0) return MI_ERR_INVALID_ARG;
  if (!int8_shape_ok(m, inter, k, x_row_stride) || out_row_stride % 4) return MI_ERR_UNSUPPORTED;
  if ((((uintptr_t)xq | (uintptr_t)wq | (uintptr_t)w_scale) & 15) || ((uintptr_t)out & 7)) return MI_ERR_INVALID_ARG;
  SkinnyPlan pl;
  const int mm = m > 256 ? 256 : m;
  pl.mt = mm <= 64 ? 4 : mm <= 128 ? 8 : 16;
  const int t80 = (inter + 79) / 80, t64 = (inter + 63) / 64;
  if (t80 > 128 && t80 <= 256) { pl.nslab = 5; pl.tiles = t80; }
  else { pl.nslab = 4; pl.tiles = t64; }
  pl.split = 1;
  SkinnyParams p;
  p.x = reinterpret_cast<const uint16_t*>(xq);
  p.w = reinterpret_cast<const uint16_t*>(wq);
  p.bias = nullptr;
  p.out = reinterpret_cast<uint16_t*>(out);
  p.part = nullptr;
  p.xs = x_scale;
  p.ws = w_scale;
  p.M = m; p.N = inter; p.K = k / 2; p.S = 1;
  p.dbg = linear_dbg();
  p.ldx = x_row_stride / 2;
  p.ldo = out_row_stride;
  return launch_skinny(p, pl, true, dtype, reinterpret_cast<hipStream_t>(stream), true);
}

// ---------------------------------------------------------------------------------- grouped form (sparse-MoE experts)
namespace mi {
int launch_grouped_skinny(const GroupedGemmArgs& a, hipStream_t s) {
  if (a.k % 64 || a.n_out % (a.gated ? 64 : 128) || a.rows_per_tile > 256 || a.rows_per_tile <= 0) return MI_ERR_UNSUPPORTED;
  if (a.dtype != MI_DTYPE_BF16 && a.dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (a.a_row_stride % 8 || a.out_row_stride % 4 || a.weight_expert_stride % 8) return MI_ERR_UNSUPPORTED;
  SkinnyParams p;
  p.x = reinterpret_cast<const uint16_t*>(a.a_rows);
  p.w = reinterpret_cast<const uint16_t*>(a.weights);
  p.bias = nullptr;
  p.out = reinterpret_cast<uint16_t*>(a.out);
  p.part = nullptr;
  p.xs = p.ws = nullptr;
  p.M = a.rows_per_tile;   // valid rows of a tile (the image holds 64 / 128 / 256: the rest repeat a valid row, never stored)
  p.N = a.n_out;
  p.K = a.k;
  p.S = 1;
  p.row_map = a.sorted_pairs;
  p.tile_expert = a.tile_expert;
  p.num_tiles = a.num_tiles;
  p.rpt = a.rows_per_tile;
  p.gather_div = a.gated ? a.top_k : 0;
  p.w_expert_stride = a.weight_expert_stride;
  p.dbg = 0;
  p.ldx = a.a_row_stride;
  p.ldo = a.out_row_stride;
  // Weight tiles of one row tile chained into one workgroup (one ring through all of them: SkinnyParams::chain): as many as
  // still leave about two workgroups per CU.  MI_MOE_CHAIN_GATED / MI_MOE_CHAIN_DOWN = <tiles per workgroup> override (A/B).
  const int n_tiles = a.gated ? a.n_out / 64 : a.n_out / 128;
  int chain = 1;
  {
    const char* e = getenv(a.gated ? "MI_MOE_CHAIN_GATED" : "MI_MOE_CHAIN_DOWN");
    if (e && atoi(e) > 0) {
      chain = atoi(e);
    } else {
      for (int c = n_tiles; c >= 1; --c)
        if (n_tiles % c == 0 && (int64_t)(n_tiles / c) * a.max_tiles >= 512) {
          chain = c;
          break;
        }
    }
    if (chain > n_tiles || n_tiles % chain) chain = 1;
  }
  p.chain = chain;
  const dim3 grid(n_tiles / chain, a.max_tiles);
  const int rpt = a.rows_per_tile;
  if (a.dtype == MI_DTYPE_BF16)
    return a.gated ? launch_grouped_rows<bf16_tag, true>(p, grid, rpt, s) : launch_grouped_rows<bf16_tag, false>(p, grid, rpt, s);
  return a.gated ? launch_grouped_rows<f16_tag, true>(p, grid, rpt, s) : launch_grouped_rows<f16_tag, false>(p, grid, rpt, s);
}
}  // namespace mi
