// A5 — paged-attention decode, matrix-core variant (gfx950).
//
// Why it exists: PMC counters of the VALU kernel (paged_attn_decode.hip) at B=256, ctx=4096 show
// the waves 10 % waiting on memory, 48 % issuing VALU and 31 % stalled at issue — with G = 7 query
// rows per KV head the score reductions and the per-lane-redundant softmax make the op
// VALU-bound (~660 VALU instructions per 16 tokens), not HBM-bound.  Here the two contractions run
// on MFMA (v_mfma_f32_16x16x32_{bf16,f16}; the G <= 16 query rows are one 16-wide tile, unused
// columns are zero) and the VALU only does the softmax of a 16x32 score tile (~60 instructions per
// 32 tokens).  MFMA time is ~4 % of the HBM time of the same bytes, so the padding is free.
//
// Data flow per wave = (sequence, query heads of one KV head, context partition), 32 tokens/step:
//   * K and V rows go HBM -> LDS by LDS-DMA (`global_load_lds_dwordx4`): one wave-instruction moves
//     4 whole 256-B token rows (full 128-B lines), needs no VGPR, and stays in flight across the
//     math of the previous step.  The LDS image is [32 rows][256 B], XOR-swizzled per 16-B chunk
//     through the per-lane SOURCE address (the DMA destination is lane-linear).
//   * swapped QK^T: S^T[tok, head] = K[tok, :] . Q^T[:, head]; A = K rows (`ds_read_b128`), B = Q^T
//     fragments held in registers.  Each lane then owns ONE head and 8 tokens: the softmax axis is
//     lane-local plus two cross-row shuffles.
//   * P^T (bf16/f16) is the B operand of O^T[d, head] += V^T[d, tok] . P^T[tok, head] as it stands
//     (a k-slot permutation shared by both operands); V^T fragments come from the row-major V
//     image through `ds_read_b64_tr_b16`.
//   * O^T accumulates over all tokens of the wave inside the MFMA, so no end-of-wave merge.
//   * Context partitions merge INSIDE this launch (MI_DECODE_FUSED_MERGE): every wave publishes its (m, l, O^T)
//     partial, bumps an arrival counter of its (sequence, head group), and the wave that arrives LAST merges the
//     group's partitions — the arithmetic of decode_reduce_kernel, lane for lane, so the result has the same bits —
//     while the other workgroups are still streaming.  One launch less per layer, but measured SLOWER than the
//     two-kernel form at every shape tried (391 against 363 us at B 256 / ctx 4096: each wave's tail gains a store
//     acknowledge + an atomic round trip while it holds its LDS slot), so it is opt-in.
// Numerics: fp32 scores/statistics/accumulators.  The MFMA's B operand is 16-bit, so P (fp32, in [0, 1])
// is carried as TWO cache-dtype values, p = hi + lo with hi = rnd16(p), lo = rnd16(p - hi): 16 (bf16) /
// 22 (f16) significant bits instead of 8 / 11, which is what brings this kernel inside north_star's
// rtol 1e-3 (atol 1e-4) against the fp32 oracle.  The PV contraction is a K = 32 MFMA whose k-slots are
// the step's two 16-token sub-tiles; the hi and the lo fragments each take one MFMA against the SAME V
// operand, so the split costs no LDS traffic: 2 x 16x16x32 per d-tile where the single-rounding form
// (SPLIT = false, MI_DECODE_ROUNDED_P: flash-style `dot(P.to(V.dtype), V)`, kept for A/B) issues
// 2 x 16x16x16.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "decode_params.h"
#include "mfma_attn_common.h"

#ifndef MI_DECODE_NS64
#define MI_DECODE_NS64 2   // ring stages at head_dim 64: a stage is half the bytes of head_dim 128's, yet four stages measured
                           // level with two there as well (162.5 / 163.1 us at 512 x 2 heads x ctx 4096, 54.7 / 56.6 at 128)
#endif

namespace mi {
namespace {

__device__ __forceinline__ int64_t bs_mask_of(int bs_log2) { return ((int64_t)1 << bs_log2) - 1; }

// AUX = cache-policy bits of the DMA loads (0 = default, 2 = nt: K/V bytes are read exactly once).
// 16- / 8-byte stores with device (agent) coherence: sc1 = write through to the point the other XCDs' loads see
__device__ __forceinline__ void store_agent_b128(float* p, f32x4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_agent_b64(float* p, f32x2_t v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// NS = stages of the wave's K/V ring (steps in flight).  2 (the product): 32 KiB of LDS per wave, five waves per CU.
// 4 (64 KiB, two waves per CU, ~56 KiB in flight per wave) was built for launches with at most two waves per CU (TP / DP
// shard shapes, small batches on long contexts) on the theory that a wave's own stream rate limits them — measured, it
// is never ahead: 47.6 against 46.7 us at B 32 / ctx 4096 (512 waves), 94 against 88 at 256 x 7/1 heads, 338 against
// 316 at the headline (profiles/r03g_decode_stages.log).  Only NS = 2 is instantiated.
// 8 consecutive 16-bit values <-> fp32 (the rotate-item of csrc/cache_rope.hip)
template <typename Tag>
__device__ __forceinline__ void unpack8(const uint4& r, float (&f)[8]) {
  using E = Elem<Tag>;
  f[0] = E::lo(r.x); f[1] = E::hi(r.x); f[2] = E::lo(r.y); f[3] = E::hi(r.y);
  f[4] = E::lo(r.z); f[5] = E::hi(r.z); f[6] = E::lo(r.w); f[7] = E::hi(r.w);
}
template <typename Tag>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  using E = Elem<Tag>;
  return make_uint4(E::pack(f[0], f[1]), E::pack(f[2], f[3]), E::pack(f[4], f[5]), E::pack(f[6], f[7]));
}
// neox rotation of dims [j0, j0 + 8) (x) and [half + j0, half + j0 + 8) (y) with the fp32 table row of the position:
// rope_cache_kernel's arithmetic, expression for expression (tests/test_gpu_attention_decode.py checks the bits)
template <typename Tag>
__device__ __forceinline__ void rope_rotate8(uint4& xr, uint4& yr, const float* row, int j0, int half) {
  float x[8], y[8], ox[8], oy[8];
  unpack8<Tag>(xr, x);
  unpack8<Tag>(yr, y);
  const float4 c0 = *reinterpret_cast<const float4*>(row + j0), c1 = *reinterpret_cast<const float4*>(row + j0 + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(row + half + j0), s1 = *reinterpret_cast<const float4*>(row + half + j0 + 4);
  const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ox[e] = x[e] * cs[e] - y[e] * sn[e];
    oy[e] = y[e] * cs[e] + x[e] * sn[e];
  }
  xr = pack8<Tag>(ox);
  yr = pack8<Tag>(oy);
}

// Per-head RMSNorm of one rotate-item (x = dims [8 j, 8 j + 8), y = their rotate-half partners) given the head's
// 1 / rms: fp32 normalise -> round to the dtype -> x weight in the dtype (HF order) — rope_cache_kernel's NORM arithmetic
template <typename Tag>
__device__ __forceinline__ void head_norm8(uint4& xr, uint4& yr, float inv, const uint16_t* w, int j0, int half) {
  using E = Elem<Tag>;
  float x[8], y[8], wx[8], wy[8];
  unpack8<Tag>(xr, x);
  unpack8<Tag>(yr, y);
  unpack8<Tag>(*reinterpret_cast<const uint4*>(w + j0), wx);
  unpack8<Tag>(*reinterpret_cast<const uint4*>(w + half + j0), wy);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    x[e] = E::from_bits(E::to_bits(E::from_bits(E::to_bits(x[e] * inv)) * wx[e]));
    y[e] = E::from_bits(E::to_bits(E::from_bits(E::to_bits(y[e] * inv)) * wy[e]));
  }
  xr = pack8<Tag>(x);
  yr = pack8<Tag>(y);
}
// sum of squares of one rotate-item, in rope_cache_kernel's order
template <typename Tag>
__device__ __forceinline__ float item_sumsq(const uint4& xr, const uint4& yr) {
  float x[8], y[8];
  unpack8<Tag>(xr, x);
  unpack8<Tag>(yr, y);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) ss += x[e] * x[e] + y[e] * y[e];
  return ss;
}

template <typename Tag, int HD, int AUX, bool SPLIT, int NS, bool ROPE = false, bool NORM = false>
__global__ __launch_bounds__(64, 2) void decode_mfma_kernel(const DecodeParams p, const int gh) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  constexpr int ROWB = HD * 2;        // bytes per token row of one head
  constexpr int KS = HD / 32;         // k-steps of QK^T
  constexpr int NT = HD / 16;         // 16-dim tiles of O^T
  constexpr int LPR = ROWB / 16;      // lanes (16-B chunks) per row: 16 / 8
  constexpr int RPI = 64 / LPR;       // rows per DMA wave-instruction: 4 / 8
  constexpr int IPS = 16 / RPI;       // DMA instructions per 16-row sub-tile: 4 / 2
  constexpr int IMG = kStep * ROWB;   // one K or V image of a stage (8 KiB)
  constexpr int STAGE = 2 * IMG;      // K image + V image
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];

  const int lane = threadIdx.x;
  const int wave = blockIdx.x;
  const int part = wave % p.num_parts;
  const int hw = (wave / p.num_parts) % (p.num_q_heads / gh);
  const int seq = wave / (p.num_parts * (p.num_q_heads / gh));
  const int qh0 = hw * gh;
  const int kvh = qh0 / p.kv_group;

  const int h = lane & 15;   // query head column / token row of a fragment
  const int g = lane >> 4;   // k-chunk / token group

  // The block-id window (lane i = i-th block of this partition; the host caps a partition at 64 blocks)
  // is fetched BEFORE seq_len is known so the two loads overlap: entries past the sequence's last
  // block are inside the table row (memory-safe) and are never selected (block indices are clamped).
  const int part_steps = p.part_tiles / 2;  // part_tiles counts 16-token tiles
  const int t_begin = part * part_steps;
  const int first_blk = (t_begin * kStep) >> p.bs_log2;
  const int32_t* const bt_row = p.block_table + (int64_t)seq * p.block_table_stride;
  // win_a: ids of the 64-block window the loop is in, win_b: of the window behind it (the DMA runs two steps ahead and
  // crosses into it near the end of a window).  A partition of <= 64 blocks never leaves win_a.
  int win_a = bt_row[min(first_blk + lane, p.block_table_stride - 1)];
  int win_b = win_a;
  int win_first = first_blk;   // logical block of win_a's lane 0 (wave-uniform)

  const int seq_len = p.seq_lens[seq];
  const int steps_total = (seq_len + kStep - 1) / kStep;
  const int t_end = min(t_begin + part_steps, steps_total);
  const bool direct = p.num_parts == 1;
  const bool head_ok = h < gh;

  uint16_t* const out16 = reinterpret_cast<uint16_t*>(p.out);
  float* const out32 = reinterpret_cast<float*>(p.out);

  if (t_begin >= t_end) {
    // Empty partition.  seq_len == 0 has no partition at all: the direct launch and, with the fused merge, the wave of
    // partition 0 write the zeros (without the fused merge decode_reduce_kernel does).
    if ((direct || (p.part_count && part == 0 && seq_len == 0)) && head_ok) {
      const int64_t o = (int64_t)seq * p.out_stride + (int64_t)(qh0 + h) * HD;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (p.out_f32) out32[o + 16 * n + 4 * g + r] = 0.f; else out16[o + 16 * n + 4 * g + r] = 0;
        }
    }
    return;
  }

  const int bs_mask = (1 << p.bs_log2) - 1;
  const int last_blk = (seq_len - 1) >> p.bs_log2;
  const int limit = min(seq_len, t_end * kStep);

  // ---- Q^T fragments (B operand): lane (head h, dims 32s + 8g ..) ------------------------
  uint4 qf[KS];
  {
    const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)seq * p.q_stride +
                         (int64_t)(qh0 + (head_ok ? h : 0)) * HD + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      qf[s] = head_ok ? *reinterpret_cast<const uint4*>(qb + 32 * s) : make_uint4(0, 0, 0, 0);
  }
  // ---- RoPE + KV write of this step's token (A4 + A7 inside the attention launch) ---------------------------------
  // q: rotated in registers AFTER the first DMA steps have been issued (below): the q, position and table-row loads then
  // ride under the DMA flight instead of standing in front of it.  k, v: the wave whose partition ends at the sequence's
  // last token, once per KV head, rotates the token's k and writes it and v to the cache row.  Its own DMA is the only
  // reader of that row in this launch: when the partition is longer than the ring (the row's step is issued from inside
  // the loop) the token's k / v loads also ride under the first DMA steps and only the stores' acknowledgement is waited
  // for; a partition of <= NS steps fetches the row in its prologue, so it writes first and waits before any DMA.
  const float* rope_row = nullptr;
  bool kv_owner = false;
  if constexpr (ROPE) {
    int64_t pos = p.positions[seq];
    pos = pos < 0 ? 0 : (pos >= p.max_position ? (int64_t)p.max_position - 1 : pos);
    rope_row = p.cos_sin + pos * HD;   // rot_dim == head_dim
    kv_owner = t_end == steps_total && qh0 % p.kv_group == 0;   // wave-uniform
  }
  auto kv_write = [&]() {
    const int64_t slot = p.slot_mapping[seq];
    if (slot < 0) return;
    const int64_t crow = (slot >> p.bs_log2) * p.blk_stride + (slot & bs_mask_of(p.bs_log2)) * p.tok_stride +
                         (int64_t)kvh * p.head_stride;
    constexpr int RI = HD / 16;   // rotate-items of one k head: lane j < RI = dims [8 j, 8 j + 8) and their partners
    if (lane < RI) {
      const uint16_t* kn = reinterpret_cast<const uint16_t*>(p.k_new) + (int64_t)seq * p.kn_stride + kvh * HD;
      uint4 x = *reinterpret_cast<const uint4*>(kn + 8 * lane), y = *reinterpret_cast<const uint4*>(kn + HD / 2 + 8 * lane);
      if constexpr (NORM) {   // the RI lanes of the head: rope_cache_kernel's reduction, lane for lane
        float ss = item_sumsq<Tag>(x, y);
        ss += dpp_f32<kDppQuadXor1>(ss);
        ss += dpp_f32<kDppQuadXor2>(ss);
        if constexpr (RI == 8) ss += dpp_f32<kDppRowHalfMirror>(ss);
        const float inv = __builtin_amdgcn_rsqf(ss / (float)HD + p.norm_eps);
        head_norm8<Tag>(x, y, inv, reinterpret_cast<const uint16_t*>(p.k_norm_w), 8 * lane, HD / 2);
      }
      rope_rotate8<Tag>(x, y, rope_row, 8 * lane, HD / 2);
      uint16_t* kc = reinterpret_cast<uint16_t*>(const_cast<void*>(p.k_cache)) + crow;
      *reinterpret_cast<uint4*>(kc + 8 * lane) = x;
      *reinterpret_cast<uint4*>(kc + HD / 2 + 8 * lane) = y;
    } else if (lane < RI + HD / 8) {
      const int j = lane - RI;
      const uint16_t* vn = reinterpret_cast<const uint16_t*>(p.v_new) + (int64_t)seq * p.vn_stride + kvh * HD;
      uint16_t* vc = reinterpret_cast<uint16_t*>(const_cast<void*>(p.v_cache)) + crow;
      *reinterpret_cast<uint4*>(vc + 8 * j) = *reinterpret_cast<const uint4*>(vn + 8 * j);
    }
  };
  const bool kv_early = kv_owner && t_end - t_begin <= NS;
  if constexpr (ROPE) {
    if (kv_early) {
      kv_write();
      wait_vmcnt<0>();   // acknowledged before the prologue DMA below can fetch the row
    }
  }

  // ---- per-lane constants -------------------------------------------------------------
  // DMA source: instruction ii of a sub-tile moves rows RPI*ii + lane/LPR; lane position pc = lane%LPR
  // of the LDS row holds logical chunk pc ^ f(row), f = swz_k / swz_v (mfma_attn_common.h).
  uint32_t src_off_k[IPS], src_off_v[IPS];
#pragma unroll
  for (int ii = 0; ii < IPS; ++ii) {  // 16*j does not change either swizzle
    const uint32_t row_b = (uint32_t)(((int64_t)(lane / LPR) * p.tok_stride + (int64_t)kvh * p.head_stride) * 2);
    src_off_k[ii] = row_b + (((lane % LPR) ^ swz_k<HD>(RPI * ii + lane / LPR)) * 16);
    src_off_v[ii] = row_b + (((lane % LPR) ^ swz_v<HD>(RPI * ii + lane / LPR)) * 16);
  }
  // K row read (A operand of QK^T): row h (+16 j), logical chunk 4s + g
  const int sw_h = swz_k<HD>(h);
  uint32_t k_rd[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) k_rd[s] = (uint32_t)(h * ROWB + (((4 * s + g) ^ sw_h) * 16));
  // V transposed read (A operand of PV): group g, lane i = 4q + pp supplies row 4g + q (+16 j),
  // dims 16n + 4pp .. +3  -> logical chunk 2n + (pp>>1), 8 B half (pp&1)
  const int q4 = (lane & 15) >> 2, pp = lane & 3;
  const int v_row = 4 * g + q4;
  const int sw_v = swz_v<HD>(v_row);
  uint32_t v_rd[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
    v_rd[n] = (uint32_t)(IMG + v_row * ROWB + (((2 * n + (pp >> 1)) ^ sw_v) * 16) + 8 * (pp & 1));

  const char* const kbase_b = reinterpret_cast<const char*>(p.k_cache);
  const char* const vbase_b = reinterpret_cast<const char*>(p.v_cache);

  // DMA groups: 8 wave-instructions fill the K image, 8 the V image of stage `st` with the 32
  // tokens of step t.  They are issued separately so each image is refilled as soon as its reads
  // have retired (rolling ring: ~3 of a wave's 4 images are in flight at any time).
  // Address arithmetic, kept scalar and minimal: ONE 64-bit multiply per 16-row sub-tile (block base),
  // then the four DMA instructions of the sub-tile step a uniform pointer by 4 rows.
  const uint64_t blk_bytes = (uint64_t)p.blk_stride * 2;
  const uint32_t tok_bytes = (uint32_t)p.tok_stride * 2;
  const uint32_t grp_bytes = RPI * tok_bytes;
  // Physical id of logical block `blk` (wave-uniform): a lane of win_a, or of win_b once the prefetch has crossed into
  // the next window.  Two readlanes and a select — no load, so no wait, anywhere near the DMA issue.
  auto phys_of = [&](int blk) -> uint32_t {
    const int rel = blk - win_first;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(win_a, rel & 63);
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane(win_b, rel & 63);
    return rel < 64 ? a : b;
  };
  auto subtile_base = [&](int tok0) -> uint64_t {  // byte offset of token row tok0 (wave-uniform)
    const int blk = min(tok0 >> p.bs_log2, last_blk);
    return (uint64_t)phys_of(blk) * blk_bytes + (uint32_t)(tok0 & bs_mask) * tok_bytes;
  };
  auto issue_k = [&](int t, int st) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* src = kbase_b + subtile_base(t * kStep + 16 * j);
#pragma unroll
      for (int ii = 0; ii < IPS; ++ii) {
        char* kdst = smem + st * STAGE + (16 * j + RPI * ii) * ROWB;
        __builtin_amdgcn_global_load_lds((gbl_void*)(src + src_off_k[ii]), (lds_void*)kdst, 16, 0, AUX);
        src += grp_bytes;
      }
    }
  };
  auto issue_v = [&](int t, int st) {
    const bool vtail = (t + 1) * kStep > limit;  // some V rows are past `limit` (wave-uniform)
    if (!vtail) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* src = vbase_b + subtile_base(t * kStep + 16 * j);
#pragma unroll
        for (int ii = 0; ii < IPS; ++ii) {
          char* vdst = smem + st * STAGE + IMG + (16 * j + RPI * ii) * ROWB;
          __builtin_amdgcn_global_load_lds((gbl_void*)(src + src_off_v[ii]), (lds_void*)vdst, 16, 0, AUX);
          src += grp_bytes;
        }
      }
    } else {
      // rows at or past `limit` take the bytes of row limit-1 instead: their probability is exactly
      // 0, and 0 * (finite) stays 0 where 0 * (stale NaN/Inf) would not.  Once per partition at most.
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int ii = 0; ii < IPS; ++ii) {
          char* vdst = smem + st * STAGE + IMG + (16 * j + RPI * ii) * ROWB;
          const int row0 = t * kStep + 16 * j + RPI * ii;
          const int tok = min(row0 + lane / LPR, limit - 1);              // per lane
          const int blk_u = min(row0, limit - 1) >> p.bs_log2;              // wave-uniform
          const uint32_t phys = phys_of(blk_u);
          const int c = (lane % LPR) ^ swz_v<HD>(RPI * ii + lane / LPR);
          const uint64_t off = (uint64_t)phys * blk_bytes + (uint64_t)((uint32_t)(tok & bs_mask) * tok_bytes) +
                               (uint64_t)((int64_t)kvh * p.head_stride * 2) + (uint32_t)(c * 16);
          __builtin_amdgcn_global_load_lds((gbl_void*)(vbase_b + off), (lds_void*)vdst, 16, 0, AUX);
        }
      }
    }
  };
  constexpr int kGrp = 2 * IPS;  // DMA wave-instructions per image (8)
  // vmcnt bookkeeping.  The only vector-memory operations in flight are DMA groups, retired in
  // issue order:  K0 V0 K1 V1 | K2 (after QK 0)  V2 (after PV 0) | K3 V3 | ...
  // so "image X has landed" == vmcnt(number of instructions issued after X's group).
  static_assert((2 * NS - 1) * kGrp <= 63, "the groups in flight must fit the 6-bit vmcnt");
  static_assert(NS == 2 || NS == 4, "the step loop is unrolled for two or four stages (three skips steps)");
  auto wait_groups = [&](int groups) {  // wave-uniform; `groups` DMA groups were issued after the one waited for
    if constexpr (NS > 2) {
      if (groups >= 7) { wait_vmcnt<7 * kGrp>(); return; }
      if (groups == 6) { wait_vmcnt<6 * kGrp>(); return; }
      if (groups == 5) { wait_vmcnt<5 * kGrp>(); return; }
      if (groups == 4) { wait_vmcnt<4 * kGrp>(); return; }
    }
    if (groups >= 3) wait_vmcnt<3 * kGrp>();
    else if (groups == 2) wait_vmcnt<2 * kGrp>();
    else if (groups == 1) wait_vmcnt<kGrp>();
    else wait_vmcnt<0>();
  };

  float m = -3.0e38f, l = 0.f;
  f32x4_t acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int t, auto stc) {
    constexpr int ST = decltype(stc)::value;
    constexpr int SB = ST * STAGE;
    // steps behind t that are already in the ring (their K and V groups were issued after K(t) / V(t)), and whether
    // step t + NS exists (its K group is issued between this step's two waits)
    const int ahead = min(NS - 1, t_end - 1 - t);
    const bool refill = t + NS < t_end;
    // K(t) landed?  issued after it: V(t), and K, V of the `ahead` steps behind it
    wait_groups(1 + 2 * ahead);
    // ---- S^T = K . Q^T for the two 16-token sub-tiles -------------------------------------
    u32x4_t k0[KS], k1[KS];
    if constexpr (KS == 4) lds_read_rows8<SB, SB + 16 * ROWB>(k_rd, k0, k1);
    else lds_read_rows4<SB, SB + 16 * ROWB>(k_rd, k0, k1);
    if (refill) issue_k(t + NS, ST);  // K image of this stage is free again: refill it now
    f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      s0 = mfma16<Tag>(__builtin_bit_cast(uint4, k0[s]), qf[s], s0);
      s1 = mfma16<Tag>(__builtin_bit_cast(uint4, k1[s]), qf[s], s1);
    }
    // lane: head h, tokens t*32 + 4g + r (s0[r]) and t*32 + 16 + 4g + r (s1[r])
    // VALU budget matters here (the kernel is only just HBM-bound): raw scores are kept, the scale rides
    // in the exp2 argument's fma; max via v_max3 without canonicalising copies; packed fp32 math.
    f32x2_t v[4] = {{s0[0], s0[1]}, {s0[2], s0[3]}, {s1[0], s1[1]}, {s1[2], s1[3]}};
    if ((t + 1) * kStep > limit) {
      const int tb = t * kStep + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (tb + r >= limit) v[r >> 1][r & 1] = -INFINITY;
        if (tb + 16 + r >= limit) v[2 + (r >> 1)][r & 1] = -INFINITY;
      }
    }
    float mx = max3f(max3f(v[0][0], v[0][1], v[1][0]), max3f(v[1][1], v[2][0], v[2][1]), max2f(v[3][0], v[3][1]));
    mx = rows_allreduce_max(mx);
    const float m_new = max2f(m, mx * p.scale_log2e);  // scale > 0: max commutes with it
    if (__any(m_new != m)) {
      const float alpha = fast_exp2(m - m_new);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] *= alpha;
    }
    const f32x2_t sc = {p.scale_log2e, p.scale_log2e}, nm = {-m, -m};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = __builtin_elementwise_fma(v[i], sc, nm);
      v[i][0] = fast_exp2(v[i][0]);
      v[i][1] = fast_exp2(v[i][1]);
    }
    const f32x2_t ls = (v[0] + v[1]) + (v[2] + v[3]);
    l += ls[0] + ls[1];
    // P^T fragments: lane (head h, tokens 4g+0..3) of each sub-tile — exactly the four scores the lane
    // already holds.  k-slot (g, j) of the K = 32 operand = token 4g + j of sub-tile 0 for j < 4, token
    // 4g + j - 4 of sub-tile 1 for j >= 4: the B operand is {pb0, pb1}, the A operand {a0[n], a1[n]}.
    const u32x2_t pb0 = {E::pack(v[0][0], v[0][1]), E::pack(v[1][0], v[1][1])};
    const u32x2_t pb1 = {E::pack(v[2][0], v[2][1]), E::pack(v[3][0], v[3][1])};
    uint4 p_hi = make_uint4(pb0[0], pb0[1], pb1[0], pb1[1]), p_lo = p_hi;
    if constexpr (SPLIT) {
      // lo = rnd16(p - hi): the subtraction is exact in fp32 (hi is p's leading bits)
      const uint32_t w[4] = {pb0[0], pb0[1], pb1[0], pb1[1]};
      uint32_t lo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) lo[i] = E::pack(v[i][0] - E::lo(w[i]), v[i][1] - E::hi(w[i]));
      p_lo = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    // V(t) landed?  issued after it: K, V of the `ahead` steps behind it, and K(t + NS) when that step exists
    wait_groups(2 * ahead + (refill ? 1 : 0));
    // ---- O^T[n] += V^T[n] . P^T, four d-tiles (8 transposed reads) per wait -------------------
#pragma unroll
    for (int nb = 0; nb < NT; nb += 4) {
      u32x2_t a0[4], a1[4];
      lds_read_tr8<SB, SB + 16 * ROWB>(&v_rd[nb], a0, a1);
#pragma unroll
      for (int n = 0; n < 4; ++n) {  // one transposed read == half an A operand: no register shuffling
        if constexpr (SPLIT) {
          const uint4 a = make_uint4(a0[n][0], a0[n][1], a1[n][0], a1[n][1]);
          acc[nb + n] = mfma16<Tag>(a, p_hi, acc[nb + n]);
          acc[nb + n] = mfma16<Tag>(a, p_lo, acc[nb + n]);
        } else {
          acc[nb + n] = mfma16_k16<Tag>(a0[n], pb0, acc[nb + n]);
          acc[nb + n] = mfma16_k16<Tag>(a1[n], pb1, acc[nb + n]);
        }
      }
    }
    if (refill) issue_v(t + NS, ST);  // every read of this stage's V image has retired
  };

  // ---- pipeline: stage t+1 streams in while stage t is consumed ------------------------------
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int steps_per_win = (64 << p.bs_log2) / kStep;
  if (t_end - t_begin > steps_per_win) {   // wave-uniform; before any DMA is in flight
    win_b = bt_row[min(first_blk + 64 + lane, p.block_table_stride - 1)];
    asm volatile("" : "+v"(win_b));
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    if (t_begin + i < t_end) {
      issue_k(t_begin + i, i);
      issue_v(t_begin + i, i);
    }
  }
  if constexpr (ROPE) {
    // q: dims 32 s + 8 g .. of head h pair with dims + HD / 2 = fragment s + KS / 2 of the same lane
    if constexpr (NORM) {
      // the head's 64 / 128 dims sit in the lanes g = 0..3 of column h (lanes 16 apart): item j = 4 s + g of the rope
      // kernel is fragments (s, s + KS / 2) of lane g.  Its reduction — quad xor 1, xor 2 over j (, then the two halves
      // of eight) — is reproduced with the same pairing: g ^ 1 (lane ^ 16), g ^ 2 (lane ^ 32), then s = 0 + s = 1.
      float tot = 0.f;
#pragma unroll
      for (int s = 0; s < KS / 2; ++s) {
        float ss = head_ok ? item_sumsq<Tag>(qf[s], qf[s + KS / 2]) : 0.f;
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        tot = s == 0 ? ss : tot + ss;
      }
      const float inv = __builtin_amdgcn_rsqf(tot / (float)HD + p.norm_eps);
      if (head_ok) {
#pragma unroll
        for (int s = 0; s < KS / 2; ++s)
          head_norm8<Tag>(qf[s], qf[s + KS / 2], inv, reinterpret_cast<const uint16_t*>(p.q_norm_w), 32 * s + 8 * g, HD / 2);
      }
    }
#pragma unroll
    for (int s = 0; s < KS / 2; ++s) rope_rotate8<Tag>(qf[s], qf[s + KS / 2], rope_row, 32 * s + 8 * g, HD / 2);
    if (kv_owner && !kv_early) {
      kv_write();
      // Acknowledged before the loop can issue the DMA of the row's step (it may do so in its very first step: a
      // counted wait only bounds the operations outstanding, it does not order three young stores).  The prologue's
      // DMA steps have landed by now (the q rotation waited for its loads behind them), so this waits for the
      // stores alone and drains nothing.
      wait_vmcnt<0>();
    }
  }
  // A partition longer than 64 blocks (MI_DECODE_LONG_PARTITIONS) is walked window by window: steps_per_win is even for
  // every block size >= 16, so a window starts on stage 0 and a 32-token step never straddles two windows.  At a window
  // change win_b becomes win_a and the ids behind it are loaded with ONE blocking vector load — its wait is forced right
  // here (the empty asm is a use), outside the pipelined loop: placed by the compiler at the first readlane instead it
  // becomes a vmcnt(0) in front of every block-id lookup and drains the LDS-DMA ring on every sub-tile (measured: 2190
  // instead of 345 us per launch).  Cost: the ring of ONE wave runs dry for ~1 us per 1024 tokens (85 us of streaming);
  // the CU's other waves keep the HBM stream going.
  for (int t = t_begin; t < t_end;) {
    const int seg_end = min(t_end, t + steps_per_win);
    for (; t < seg_end; t += NS) {   // steps_per_win is a multiple of NS: a window starts on stage 0
      compute(t, S0{});
      if (t + 1 < t_end) compute(t + 1, S1{});
      if constexpr (NS == 4) {
        if (t + 2 < t_end) compute(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < t_end) compute(t + 3, std::integral_constant<int, 3>{});
      }
    }
    if (t < t_end) {   // next window (only ever taken with multi_win)
      win_a = win_b;
      win_first += 64;
      win_b = bt_row[min(win_first + 64 + lane, p.block_table_stride - 1)];
      asm volatile("" : "+v"(win_b));
    }
  }
  // the last step waited with vmcnt(0): no DMA can land in this workgroup's LDS after it exits

  // ---- epilogue ---------------------------------------------------------------------------------
  l = rows_allreduce_sum(l);
  if (direct) {
    if (!head_ok) return;
    const float inv = 1.f / l;
    const int64_t o = (int64_t)seq * p.out_stride + (int64_t)(qh0 + h) * HD + 4 * g;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (p.out_f32) {
        *reinterpret_cast<float4*>(out32 + o + 16 * n) =
            make_float4(acc[n][0] * inv, acc[n][1] * inv, acc[n][2] * inv, acc[n][3] * inv);
      } else {
        uint2 r;
        r.x = E::pack(acc[n][0] * inv, acc[n][1] * inv);
        r.y = E::pack(acc[n][2] * inv, acc[n][3] * inv);
        *reinterpret_cast<uint2*>(out16 + o + 16 * n) = r;
      }
    }
    return;
  }
  const bool fused = p.part_count != nullptr;
  if (head_ok) {
    const int64_t slot = ((int64_t)seq * p.num_q_heads + (qh0 + h)) * p.num_parts + part;
    float* po = p.part_o + slot * HD + 4 * g;
    if (fused) {
      // device-coherent (sc1) stores: they go past this XCD's L2 to the coherence point themselves, so publishing
      // them needs no L2 write-back — an agent-scope release FENCE here (buffer_wbl2 from every one of thousands of
      // waves) cost 50 % of the kernel's time
#pragma unroll
      for (int n = 0; n < NT; ++n) store_agent_b128(po + 16 * n, acc[n]);
      if (g == 0) store_agent_b64(p.part_ml + slot * 2, f32x2_t{m, l});
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n)
        *reinterpret_cast<float4*>(po + 16 * n) = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
      if (g == 0) *reinterpret_cast<float2*>(p.part_ml + slot * 2) = make_float2(m, l);
    }
  }
  if (!fused) return;   // merged by decode_reduce_kernel

  // ---- fused merge: the wave of this (sequence, head group) that arrives last merges its partitions ------------
  // Release: the partial stores above are device-coherent and COMPLETE (vmcnt(0): one wave = one instruction stream,
  // the wait covers every lane's stores) before the counter moves.
  wait_vmcnt<0>();
  int32_t* const cnt = p.part_count + ((int64_t)seq * (p.num_q_heads / gh) + hw);
  int arrived = 0;
  if (lane == 0) arrived = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  arrived = __builtin_amdgcn_readfirstlane(arrived);
  const int np = min(p.num_parts, (steps_total + part_steps - 1) / part_steps);   // partitions that hold tokens
  if (arrived != np - 1) return;
  // Acquire: every load below is a device-coherent (sc1) load issued after the counter's value came back, so no
  // cache of this CU / XCD can serve a line older than the publishers' stores.
  if (lane == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  constexpr int DPL = HD / 64;   // dims per lane, as in decode_reduce_kernel
  for (int hh = 0; hh < gh; ++hh) {
    const int qh = qh0 + hh;
    const int64_t slot0 = ((int64_t)seq * p.num_q_heads + qh) * p.num_parts;
    const float* ml = p.part_ml + slot0 * 2;
    // (atomic loads: never scalar loads, never hoisted above the acquire)
    float mm = -3.0e38f;
    for (int i = 0; i < np; ++i)
      mm = fmaxf(mm, __hip_atomic_load(ml + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    float L = 0.f, a[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) a[d] = 0.f;
    for (int i = 0; i < np; ++i) {
      const float mi_ = __hip_atomic_load(ml + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float li_ = __hip_atomic_load(ml + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float w = fast_exp2(mi_ - mm);
      L += w * li_;
      const float* po = p.part_o + (slot0 + i) * HD + lane * DPL;
#pragma unroll
      for (int d = 0; d < DPL; ++d) a[d] += w * __hip_atomic_load(po + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float inv = 1.f / L;
    const int64_t o = (int64_t)seq * p.out_stride + (int64_t)qh * HD + lane * DPL;
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
      const float r = a[d] * inv;
      if (p.out_f32) out32[o + d] = r;
      else out16[o + d] = E::to_bits(r);
    }
  }
}

}  // namespace

// Launch the MFMA decode kernel.  Caller guarantees: head_dim in {64, 128}, block_size >= 16,
// partition a multiple of 32 tokens (any number of blocks: the block-id window is reloaded every 64 blocks).
int launch_decode_mfma(const DecodeParams& p, int gh, int head_dim, int dtype, bool nt, bool split_p,
                       hipStream_t stream) {
  const int64_t waves = (int64_t)p.num_seqs * (p.num_q_heads / gh) * p.num_parts;
  if (waves > INT32_MAX) return MI_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)waves), block(64);
#define MI_LAUNCH_NS(TAG, HDIM, NS_)                                                                        \
  do {                                                                                                      \
    if (split_p) {                                                                                          \
      if (nt) hipLaunchKernelGGL((decode_mfma_kernel<TAG, HDIM, 2, true, NS_>), grid, block, 0, stream, p, gh);  \
      else hipLaunchKernelGGL((decode_mfma_kernel<TAG, HDIM, 0, true, NS_>), grid, block, 0, stream, p, gh);     \
    } else {                                                                                                \
      if (nt) hipLaunchKernelGGL((decode_mfma_kernel<TAG, HDIM, 2, false, NS_>), grid, block, 0, stream, p, gh); \
      else hipLaunchKernelGGL((decode_mfma_kernel<TAG, HDIM, 0, false, NS_>), grid, block, 0, stream, p, gh);    \
    }                                                                                                       \
  } while (0)
#define MI_LAUNCH(TAG, HDIM) MI_LAUNCH_NS(TAG, HDIM, 2)
  if (dtype == MI_DTYPE_BF16) {
    if (head_dim == 128) MI_LAUNCH(bf16_tag, 128); else MI_LAUNCH_NS(bf16_tag, 64, MI_DECODE_NS64);
  } else {
    if (head_dim == 128) MI_LAUNCH(f16_tag, 128); else MI_LAUNCH_NS(f16_tag, 64, MI_DECODE_NS64);
  }
#undef MI_LAUNCH
#undef MI_LAUNCH_NS
  return launch_status();
}

int launch_decode_mfma_rope(const DecodeParams& p, int gh, int head_dim, int dtype, hipStream_t stream) {
  const int64_t waves = (int64_t)p.num_seqs * (p.num_q_heads / gh) * p.num_parts;
  if (waves > INT32_MAX) return MI_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)waves), block(64);
  if (p.q_norm_w) {   // per-head q / k RMSNorm in front of the rotation (Qwen3 family)
    if (dtype == MI_DTYPE_BF16) {
      if (head_dim == 128) hipLaunchKernelGGL((decode_mfma_kernel<bf16_tag, 128, 2, true, 2, true, true>), grid, block, 0, stream, p, gh);
      else hipLaunchKernelGGL((decode_mfma_kernel<bf16_tag, 64, 2, true, MI_DECODE_NS64, true, true>), grid, block, 0, stream, p, gh);
    } else {
      if (head_dim == 128) hipLaunchKernelGGL((decode_mfma_kernel<f16_tag, 128, 2, true, 2, true, true>), grid, block, 0, stream, p, gh);
      else hipLaunchKernelGGL((decode_mfma_kernel<f16_tag, 64, 2, true, MI_DECODE_NS64, true, true>), grid, block, 0, stream, p, gh);
    }
    return launch_status();
  }
  if (dtype == MI_DTYPE_BF16) {
    if (head_dim == 128) hipLaunchKernelGGL((decode_mfma_kernel<bf16_tag, 128, 2, true, 2, true>), grid, block, 0, stream, p, gh);
    else hipLaunchKernelGGL((decode_mfma_kernel<bf16_tag, 64, 2, true, MI_DECODE_NS64, true>), grid, block, 0, stream, p, gh);
  } else {
    if (head_dim == 128) hipLaunchKernelGGL((decode_mfma_kernel<f16_tag, 128, 2, true, 2, true>), grid, block, 0, stream, p, gh);
    else hipLaunchKernelGGL((decode_mfma_kernel<f16_tag, 64, 2, true, MI_DECODE_NS64, true>), grid, block, 0, stream, p, gh);
  }
  return launch_status();
}

}  // namespace mi
