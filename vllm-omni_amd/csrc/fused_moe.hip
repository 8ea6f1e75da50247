// N4 — fused sparse-MoE decode for the Qwen3-Omni talker.  Replaces upstream `SharedFusedMoE`
// behind `Qwen3OmniMoeModel` (vllm_omni/model_executor/models/qwen3_omni/qwen3_omni_moe_talker.py:476-513;
// algorithm as in transformers' Qwen3OmniMoeTalkerTextSparseMoeBlock): 128 experts, top-8, H 1024, I 384.
//
//   route   : p = softmax_fp32(router logits); (w, e) = top-k(p); optional renormalisation        one wave / token
//   align   : group the T*k (token, slot) pairs by expert, every group padded to a multiple of 32 rows so that a
//             32-row tile never straddles two experts; tile -> expert map; pair -> row map          one workgroup
//   gemm 1  : act[row, :] = silu(x[tok] . Wg_e^T) * (x[tok] . Wu_e^T)                               MFMA, bf16
//   gemm 2  : part[row, :] = act[row, :] . Wd_e^T                                                   MFMA, bf16
//   combine : out[t] = sum_j w[t, j] * part[row(t, j)] + sigmoid(x[t] . w_sg) * shared[t]           one WG / token
//
// At decode the op is weight-streaming: every expert that is hit (all of them at T*k >= a few hundred pairs)
// streams its 1.2 MB of weights once per 32-row tile; 302 MB per layer at the real shape.  Both GEMMs use the
// swapped form D = W . X^T (A operand = 16 weight rows, B operand = 16 token rows, K-contiguous 16-byte
// fragments straight from global memory): a lane then owns 4 consecutive output features of one token, so the
// SiLU-gate epilogue is lane-local (a wave computes the matching gate and up rows) and stores are 8 bytes.
#include <cstring>

#include "common.h"
#include "skinny_grouped.h"
#include "mfma_attn_common.h"

namespace mi {
namespace {

// Rows (token, slot pairs) per tile: 32 or 64, chosen by the caller from the mean pairs per expert — a tile streams
// its expert's weights once, so an expert whose pairs spill into a second tile reads them twice (measured at
// T = 512, 32 pairs per expert on average: 190 tiles of 32 for 128 experts = 1.5x the weight bytes).

// ---------------------------------------------------------------- route
// wave-wide max on the VALU: DPP row rotations inside the 16-lane rows, then the row exchange (common.h).  The
// `__shfl_xor` form is a ds_bpermute and an lgkmcnt wait per step; the k arg-max rounds of a token are one dependent
// chain, and with twelve of those per round a token took ~5 us.  max is order-independent: same bits.
template <int N>
__device__ __forceinline__ float row_ror(float v) {   // lane l of a 16-lane row <- lane (l + N) % 16
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_max_valu(float v) {
  v = fmaxf(v, row_ror<8>(v));   // fmaxf: a NaN operand loses, as in the first pass over the logits
  v = fmaxf(v, row_ror<4>(v));
  v = fmaxf(v, row_ror<2>(v));
  v = fmaxf(v, row_ror<1>(v));
  return rows_allreduce_max(v);
}

template <int DT, int EPL>  // EPL = experts per lane = E / 64
__global__ __launch_bounds__(1024) void moe_route_kernel(const void* __restrict__ logits, int64_t row_stride,
                                                         int32_t* __restrict__ ids, float* __restrict__ weights,
                                                         int32_t* __restrict__ counts, int T, int top_k, int norm,
                                                         int round_dt) {
  scalar_cache_acquire();
  // One wave per token, 16 waves per workgroup, a strided loop over tokens.  The per-expert counts are gathered in
  // LDS and flushed once per workgroup with consecutive lanes on consecutive counters: T * top_k single-lane global
  // atomics on E addresses (the first version) serialise in L2 — 65536 of them cost 128 us at T = 8192.
  __shared__ int32_t hist[64 * EPL];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 64 * EPL; e += 1024) hist[e] = 0;
  __syncthreads();
  for (int t = blockIdx.x * 16 + wid; t < T; t += gridDim.x * 16) {
    float p[EPL];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = lane + 64 * i;
      if constexpr (DT == MI_DTYPE_F32) p[i] = reinterpret_cast<const float*>(logits)[t * row_stride + e];
      else if constexpr (DT == MI_DTYPE_BF16)
        p[i] = Elem<bf16_tag>::from_bits(reinterpret_cast<const uint16_t*>(logits)[t * row_stride + e]);
      else p[i] = Elem<f16_tag>::from_bits(reinterpret_cast<const uint16_t*>(logits)[t * row_stride + e]);
      mx = fmaxf(mx, p[i]);
    }
    mx = wave_max_valu(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      p[i] = __expf(p[i] - mx);
      s += p[i];
    }
    s = wave_allreduce_sum(s);
    const float inv = 1.f / s;
    float wsum = 0.f, myw = 0.f;
    int mye = 0;
    for (int j = 0; j < top_k; ++j) {  // k rounds of wave arg-max, lowest expert id on ties
      // the largest remaining probability, then the lowest expert id that holds it
      float best = -1.f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) best = p[i] > best ? p[i] : best;   // NaN never wins
      best = wave_max_valu(best);
      int bi = 0x7fffffff;
      if (best >= 0.f) {
#pragma unroll
        for (int i = EPL - 1; i >= 0; --i) {
          const uint64_t m = __ballot(p[i] == best);
          if (m) bi = __builtin_ctzll(m) + 64 * i;
        }
      }
#pragma unroll
      for (int i = 0; i < EPL; ++i)
        if (lane + 64 * i == bi) p[i] = -1.f;  // taken
      if (bi == 0x7fffffff) {  // NaN logits: no comparison succeeds.  Never let an out-of-range id reach the
        bi = j;                // counters or the grouping; experts 0..k-1 with weight NaN keep the poison visible
        best = __builtin_nanf("");
      }
      const float w = best * inv;
      wsum += w;
      if (lane == j) { myw = w; mye = bi; }
    }
    if (lane < top_k) {
      float w = norm ? myw / wsum : myw;
      if (round_dt == MI_DTYPE_BF16) w = Elem<bf16_tag>::from_bits(Elem<bf16_tag>::to_bits(w));
      else if (round_dt == MI_DTYPE_F16) w = Elem<f16_tag>::from_bits(Elem<f16_tag>::to_bits(w));
      ids[t * top_k + lane] = mye;
      weights[t * top_k + lane] = w;
      if (counts) atomicAdd(&hist[mye], 1);
    }
  }
  if (!counts) return;   // the grouping that follows counts the ids itself (mi_moe_align with expert_counts == NULL)
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * EPL; e += 1024) {
    const int c = hist[e];
    if (c) atomicAdd(&counts[e], c);
  }
}

__global__ __launch_bounds__(256) void moe_zero_counts_kernel(int32_t* __restrict__ counts, int E) {
  for (int e = threadIdx.x; e < E; e += 256) counts[e] = 0;
}

// ---------------------------------------------------------------- align
// gridDim.x workgroups, each owning one contiguous chunk of the (token, k) pairs.  The row of a pair is
// start[expert] + (pairs of that expert in EARLIER chunks) + (its arrival rank inside the chunk): a workgroup gets the
// middle term by counting the experts of every pair in front of its chunk itself (a few LDS-atomic passes over ids the
// L2 already holds: the last of 16 workgroups re-reads 240 KB at 8192 tokens x top-8) instead of through a global cursor
// array (every workgroup makes ONE pass over all ids, which also tells it each expert's true pair count for the padding),
// so the launch needs no workspace and no second kernel.  One workgroup (the decode batch) is the round-2 kernel;
// at 8192 tokens that one CU issued all 65 536 scattered stores itself (45-66 us).
__global__ __launch_bounds__(1024) void moe_align_kernel(const int32_t* __restrict__ ids,
                                                         const int32_t* __restrict__ counts, int num_pairs, int E,
                                                         int32_t* __restrict__ sorted_pairs,
                                                         int32_t* __restrict__ row_of_pair,
                                                         int32_t* __restrict__ tile_expert,
                                                         int32_t* __restrict__ num_tiles, int max_rows, int rpt) {
  scalar_cache_acquire();
  __shared__ int32_t start[257];   // padded row offset of every expert
  __shared__ int32_t cursor[256];  // next free row of every expert, relative to start[]
  __shared__ int32_t total[256];   // pairs of every expert as the ids really have them (counts[] is the caller's word)
  const int tid = threadIdx.x;
  const int wg = blockIdx.x, nwg = gridDim.x;
  for (int e = tid; e < E; e += 1024) cursor[e] = total[e] = 0;
  __syncthreads();
  // this workgroup's chunk of pairs (a multiple of 4096, so the four-pairs-per-trip loop below keeps its stride)
  const int chunk = ((num_pairs + nwg - 1) / nwg + 4095) / 4096 * 4096;
  const int lo = min(wg * chunk, num_pairs), hi = min(lo + chunk, num_pairs);
  // one pass over ALL ids: pairs of every expert in front of the chunk (where the chunk's rows start) and in total
  // (where the expert's padding starts — taken from the ids themselves, so a row is either a pair or -1 whatever the
  // caller's counts say)
  for (int i = tid; i < num_pairs; i += 1024) {
    const int e = ids[i];
    if (e >= 0 && e < E) {
      atomicAdd(&total[e], 1);
      if (i < lo) atomicAdd(&cursor[e], 1);
    }
  }
  __syncthreads();
  // padded row offset of every expert — from the caller's counts, or (counts == NULL) from the ids just counted: the
  // roundings in parallel, the short running sum by one thread
  // exclusive prefix sum over the <= 256 experts by the first four waves: a wave scan (six shuffles) + the three wave
  // totals through LDS.  (One thread walking the experts — 128 dependent LDS round trips — was 7 of the launch's 11.5 us at
  // the decode batch.)
  __shared__ int32_t wave_total[4];
  int pad = 0, incl = 0;
  if (tid < 256) {
    pad = tid < E ? ((counts ? counts[tid] : total[tid]) + rpt - 1) / rpt * rpt : 0;
    incl = pad;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if ((tid & 63) >= o) incl += up;
    }
    if ((tid & 63) == 63) wave_total[tid >> 6] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    int before = 0;
    for (int w = 0; w < (tid >> 6); ++w) before += wave_total[w];
    if (tid < E) start[tid] = before + incl - pad;
    if (tid == E - 1) {
      start[E] = before + incl;
      if (wg == 0) num_tiles[0] = (before + incl) / rpt;
    }
  }
  __syncthreads();
  // padding rows (-1) and tile owners: experts are dealt round-robin to the workgroups; a padding row is never a live
  // row of any pair, so nobody else writes it
  for (int e = wg + nwg * (tid >> 6); e < E; e += nwg * 16) {
    const int live_end = start[e] + min(total[e], start[e + 1] - start[e]);
    for (int r = live_end + (tid & 63); r < start[e + 1] && r < max_rows; r += 64) sorted_pairs[r] = -1;
    for (int r = start[e] + rpt * (tid & 63); r < start[e + 1]; r += rpt * 64) tile_expert[r / rpt] = e;
  }
  // four pairs per thread and trip: the id loads, the LDS atomics and the scattered stores of the four are independent,
  // so their latencies overlap (one pair per trip was 64 dependent round trips at 8192 tokens x top-8)
  for (int i0 = lo + tid; i0 < hi; i0 += 4096) {
    int e[4], r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 1024 * u;
      e[u] = i < hi ? ids[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = e[u] >= 0 && e[u] < E;   // caller-supplied routing out of range: row 0, never OOB
      r[u] = ok ? start[e[u]] + atomicAdd(&cursor[e[u]], 1) : -1;
      if (ok && (r[u] >= start[e[u] + 1] || r[u] >= max_rows)) r[u] = -1;   // counts disagree with ids
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 1024 * u;
      if (i < hi) {
        if (r[u] >= 0) sorted_pairs[r[u]] = i;
        row_of_pair[i] = r[u] >= 0 ? r[u] : 0;
      }
    }
  }
}

// ---------------------------------------------------------------- expert GEMMs
struct MoeGemmParams {
  const uint16_t* a_rows;        // gemm 1: x [T, K];  gemm 2: act [rows, K]
  const uint16_t* w;             // [E, N_total, K]
  uint16_t* out;                 // gemm 1: act [rows, I];  gemm 2: part [rows, H]
  const int32_t* sorted_pairs;   // row -> pair (t * top_k + j) or -1
  const int32_t* tile_expert;
  const int32_t* num_tiles;
  int32_t K, N_out, top_k;
  int64_t a_stride, out_stride, w_expert_stride;
};

// GATED: gemm 1 — a wave owns 16 gate rows and the 16 matching up rows; WG = 4 waves = 64 outputs.
// !GATED: gemm 2 — a wave owns 32 weight rows; WG = 4 waves = 128 outputs.
template <typename Tag, bool GATED, int MT>  // MT = 16-row token tiles per workgroup tile (2 or 4)
__global__ __launch_bounds__(256) void moe_gemm_kernel(const MoeGemmParams p) {
  constexpr int kMoeRows = 16 * MT;
  scalar_cache_acquire();
  using E = Elem<Tag>;
  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so hand every XCD
  // a CONTIGUOUS range of tiles — the tiles of one expert then stream its weights through one L2, back to back
  // (gridDim.x is a multiple of 8, so the XCD of a workgroup is blockIdx.x % 8 for every blockIdx.y)
  const int nt = p.num_tiles[0];
  const int per_xcd = (nt + 7) >> 3;   // ranges sized by the LIVE tile count, or the last XCDs would sit idle
  const int slot = blockIdx.x >> 3;
  const int tile = (blockIdx.x & 7) * per_xcd + slot;
  if (slot >= per_xcd || tile >= nt) return;  // whole workgroup: uniform
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int e = p.tile_expert[tile];
  const int row0 = tile * kMoeRows;
  // B operand rows (tokens): MT m-tiles of 16
  const uint16_t* brow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = row0 + 16 * mt + r16;
    if constexpr (GATED) {
      const int pair = p.sorted_pairs[row];
      brow[mt] = p.a_rows + (int64_t)(pair < 0 ? 0 : pair / p.top_k) * p.a_stride + 16 * kg;
    } else {
      brow[mt] = p.a_rows + (int64_t)row * p.a_stride + 16 * kg;
    }
  }
  // A operand rows (weights)
  const int n_base = GATED ? blockIdx.y * 64 + 16 * wid : blockIdx.y * 128 + 32 * wid;
  const uint16_t* wexp = p.w + (int64_t)e * p.w_expert_stride;
  const uint16_t* arow[2];
  arow[0] = wexp + (int64_t)(n_base + r16) * p.K + 16 * kg;                                  // gate / rows 0-15
  arow[1] = wexp + (int64_t)(GATED ? p.N_out + n_base + r16 : n_base + 16 + r16) * p.K + 16 * kg;  // up / rows 16-31
  f32x4_t acc[2][MT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // Software pipeline, by hand (hipcc does not unroll this runtime-trip-count loop: it emitted 4 loads -> wait ->
  // 4 MFMAs per iteration, i.e. one 4 KB k-step in flight per wave).  Groups of UNR k-steps: the 4*UNR fragment
  // loads of the next group are issued before the MFMAs of the current one.
  constexpr int UNR = 2;   // 2 k-steps per group: ~100 VGPRs -> 5 workgroups per CU, every tile of a decode batch resident at once
  uint4 fa[2][UNR][2], fb[2][UNR][MT];
  auto load_group = [&](int buf, int k0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      // k permutation inside a 64-wide block (the MFMA only needs A and B to agree on it): lane group kg holds
      // elements 16 kg .. 16 kg + 7 for the even k-step and 16 kg + 8 .. + 15 for the odd one, so the four lanes of
      // a row cover one whole 128-byte line per instruction pair instead of two half lines
      const int k = k0 + 64 * (u >> 1);
      const bool ok = k < p.K;   // K is a multiple of 64, not necessarily of 32 * UNR
      const int kk = (ok ? k : 0) + 8 * (u & 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[buf][u][i] = *reinterpret_cast<const uint4*>(arow[i] + kk);
#pragma unroll
      for (int i = 0; i < MT; ++i) fb[buf][u][i] = *reinterpret_cast<const uint4*>(brow[i] + kk);
      if (!ok) fa[buf][u][0] = fa[buf][u][1] = make_uint4(0, 0, 0, 0);   // a zero weight fragment adds nothing
    }
  };
  auto mma_group = [&](int buf) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        acc[0][j] = mfma16<Tag>(fa[buf][u][0], fb[buf][u][j], acc[0][j]);
        acc[1][j] = mfma16<Tag>(fa[buf][u][1], fb[buf][u][j], acc[1][j]);
      }
    }
  };
  load_group(0, 0);
  for (int k0 = 0; k0 < p.K; k0 += 2 * 32 * UNR) {
    load_group(1, k0 + 32 * UNR);       // past-the-end groups load row starts and contribute zeros
    mma_group(0);
    if (k0 + 32 * UNR >= p.K) break;
    load_group(0, k0 + 2 * 32 * UNR);
    mma_group(1);
  }
  static_assert(UNR % 2 == 0, "a group covers whole 64-wide k blocks");
  // D[n][m]: lane = column m (token r16 of m-tile), rows n = 4 kg + r
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = row0 + 16 * mt + r16;
    uint16_t* o = p.out + (int64_t)row * p.out_stride;
    if constexpr (GATED) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // the 16-bit grid of the two GEMM outputs first (what separate GEMM + SiLU-mul kernels would see)
        const float g = E::from_bits(E::to_bits(acc[0][mt][r])), u = E::from_bits(E::to_bits(acc[1][mt][r]));
        v[r] = g / (1.f + __expf(-g)) * u;
      }
      uint2 q;
      q.x = E::pack(v[0], v[1]);
      q.y = E::pack(v[2], v[3]);
      *reinterpret_cast<uint2*>(o + n_base + 4 * kg) = q;
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint2 q;
        q.x = E::pack(acc[i][mt][0], acc[i][mt][1]);
        q.y = E::pack(acc[i][mt][2], acc[i][mt][3]);
        *reinterpret_cast<uint2*>(o + n_base + 16 * i + 4 * kg) = q;
      }
    }
  }
}

// ---------------------------------------------------------------- combine
template <typename Tag>
__global__ __launch_bounds__(256) void moe_combine_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ part,
                                                          const int32_t* __restrict__ row_of_pair,
                                                          const float* __restrict__ weights,
                                                          const uint16_t* __restrict__ shared,
                                                          const uint16_t* __restrict__ x,
                                                          const uint16_t* __restrict__ shared_gate_w, int H, int top_k,
                                                          int64_t part_stride, int64_t out_stride, int64_t x_stride,
                                                          int64_t shared_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  __shared__ float red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  float gate = 0.f;
  if (shared) {
    gate = 1.f;
    if (shared_gate_w) {  // sigmoid(x[t] . w_sg)
      float d = 0.f;
      for (int i = tid; i < H; i += 256) d += E::from_bits(x[t * x_stride + i]) * E::from_bits(shared_gate_w[i]);
      d = wave_allreduce_sum(d);
      if ((tid & 63) == 0) red[tid >> 6] = d;
      __syncthreads();
      d = (red[0] + red[1]) + (red[2] + red[3]);
      gate = 1.f / (1.f + __expf(-d));
    }
  }
  for (int c = tid * 4; c < H; c += 1024) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < top_k; ++j) {
      const float w = weights[t * top_k + j];
      const uint2 v = *reinterpret_cast<const uint2*>(part + (int64_t)row_of_pair[t * top_k + j] * part_stride + c);
      a[0] += w * E::lo(v.x); a[1] += w * E::hi(v.x); a[2] += w * E::lo(v.y); a[3] += w * E::hi(v.y);
    }
    if (shared) {
      const uint2 v = *reinterpret_cast<const uint2*>(shared + t * shared_stride + c);
      a[0] += gate * E::lo(v.x); a[1] += gate * E::hi(v.x); a[2] += gate * E::lo(v.y); a[3] += gate * E::hi(v.y);
    }
    uint2 q;
    q.x = E::pack(a[0], a[1]);
    q.y = E::pack(a[2], a[3]);
    *reinterpret_cast<uint2*>(out + t * out_stride + c) = q;
  }
}

}  // namespace
}  // namespace mi

namespace {
// 32 / 64: both expert GEMM kernels; 128 / 256 (prefill chunks): the ring kernel only
bool moe_rows_per_tile_ok(int32_t r) { return r == 32 || r == 64 || r == 128 || r == 256; }
}  // namespace

extern "C" int32_t mi_moe_rows_per_tile(int32_t num_tokens, int32_t top_k, int32_t num_experts) {
  if (num_tokens <= 0 || top_k <= 0 || num_experts <= 0) return 32;
  // The ring kernel (skinny_gemm.hip, grouped form) stages a 64-row image per tile whatever the tile height, and every
  // tile streams its expert's weights once: 64-row tiles win as soon as experts average more than ~24 rows (measured
  // at the talker shape, tools/bench_moe.py: T = 512: 105 us against 116 us with 32-row tiles; T = 256: 99 against 88).
  // The fragments-from-global kernel (MI_MOE_GEMM=direct) is faster with 32-row tiles at every size measured.
  const char* e = getenv("MI_MOE_GEMM");
  if (e && strcmp(e, "direct") == 0) return 32;
  // Prefill chunks (an expert sees hundreds of rows): 256-row tiles stream an expert's weights 4x less often than 64-row
  // ones — 459 against 497 us at 8192 tokens, 277 / 294 at 4096, level at 2048, behind at 1024
  // (profiles/r03w_moe_tall_tiles.log); the gain is small because K is short here (1024 / 384: 16 / 6 ring steps per tile)
  if ((int64_t)num_tokens * top_k >= 256 * (int64_t)num_experts) return 256;
  return (int64_t)num_tokens * top_k > 24 * (int64_t)num_experts ? 64 : 32;
}

extern "C" int64_t mi_moe_max_rows(int32_t num_tokens, int32_t top_k, int32_t num_experts, int32_t rows_per_tile) {
  if (num_tokens <= 0 || top_k <= 0 || num_experts <= 0 || !moe_rows_per_tile_ok(rows_per_tile)) return 0;
  const int64_t pairs = (int64_t)num_tokens * top_k;
  const int64_t hit = pairs < num_experts ? pairs : num_experts;  // each hit expert pads by < rows_per_tile rows
  return (pairs + hit * (rows_per_tile - 1) + rows_per_tile - 1) / rows_per_tile * rows_per_tile;
}

extern "C" int mi_moe_route(int32_t* topk_ids, float* topk_weights, int32_t* expert_counts, const void* router_logits,
                            int32_t num_tokens, int32_t num_experts, int64_t logits_row_stride, int32_t top_k,
                            int32_t norm_topk_prob, int logits_dtype, int weight_round_dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!topk_ids || !topk_weights || !router_logits || num_tokens < 0) return MI_ERR_INVALID_ARG;
  if (num_experts != 64 && num_experts != 128 && num_experts != 256) return MI_ERR_UNSUPPORTED;
  if (top_k <= 0 || top_k > 16 || top_k > num_experts) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // expert_counts == NULL: no counting (and no launch to zero the counters) — mi_moe_align counts the ids itself.
  // (a kernel, not hipMemsetAsync: twenty memset nodes inside the decode hipGraph of the 20-layer talker ended in
  //  "write access to a read-only page" at replay on this stack, while the same graph with 2-3 of them was fine)
  if (expert_counts) hipLaunchKernelGGL(moe_zero_counts_kernel, dim3(1), dim3(256), 0, s, expert_counts, num_experts);
  // tokens per wave: 1 for decode batches (latency), up to 4 for prefill chunks (fewer, fuller count flushes)
  const int tpw = num_tokens >= 8192 ? 4 : num_tokens >= 1024 ? 2 : 1;
  const dim3 grid((num_tokens + 16 * tpw - 1) / (16 * tpw)), block(1024);
#define MI_ROUTE(DT, EPL)                                                                                        \
  hipLaunchKernelGGL((moe_route_kernel<DT, EPL>), grid, block, 0, s, router_logits, logits_row_stride, topk_ids, \
                     topk_weights, expert_counts, num_tokens, top_k, norm_topk_prob, weight_round_dtype)
#define MI_ROUTE_E(DT)                                  \
  do {                                                  \
    if (num_experts == 64) MI_ROUTE(DT, 1);             \
    else if (num_experts == 128) MI_ROUTE(DT, 2);       \
    else MI_ROUTE(DT, 4);                               \
  } while (0)
  switch (logits_dtype) {
    case MI_DTYPE_F32: MI_ROUTE_E(MI_DTYPE_F32); break;
    case MI_DTYPE_BF16: MI_ROUTE_E(MI_DTYPE_BF16); break;
    case MI_DTYPE_F16: MI_ROUTE_E(MI_DTYPE_F16); break;
    default: return MI_ERR_UNSUPPORTED;
  }
#undef MI_ROUTE_E
#undef MI_ROUTE
  return launch_status();
}

extern "C" int mi_moe_align(int32_t* sorted_pairs, int32_t* row_of_pair, int32_t* tile_expert, int32_t* num_tiles,
                            const int32_t* topk_ids, const int32_t* expert_counts, int32_t num_pairs,
                            int32_t num_experts, int64_t max_rows, int32_t rows_per_tile, void* stream) {
  using namespace mi;
  if (num_pairs == 0) return MI_OK;
  if (!sorted_pairs || !row_of_pair || !tile_expert || !num_tiles || !topk_ids || num_pairs < 0)
    return MI_ERR_INVALID_ARG;
  if (num_experts <= 0 || num_experts > 256 || max_rows > 0x7fffffff) return MI_ERR_UNSUPPORTED;
  if (!moe_rows_per_tile_ok(rows_per_tile)) return MI_ERR_INVALID_ARG;
  // one workgroup per 4096 pairs, at most 32 (a decode batch of 512 tokens x top-8 is exactly one)
  int nwg = (num_pairs + 4095) / 4096;
  if (nwg > 32) nwg = 32;
  hipLaunchKernelGGL(moe_align_kernel, dim3(nwg), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), topk_ids,
                     expert_counts, num_pairs, num_experts, sorted_pairs, row_of_pair, tile_expert, num_tiles,
                     (int)max_rows, (int)rows_per_tile);
  return launch_status();
}

extern "C" int mi_moe_expert_gemm(void* out, const void* a_rows, const void* weights, const int32_t* sorted_pairs,
                                  const int32_t* tile_expert, const int32_t* num_tiles, int32_t max_tiles,
                                  int32_t gated, int32_t n_out, int32_t k, int32_t top_k, int64_t a_row_stride,
                                  int64_t out_row_stride, int64_t weight_expert_stride, int32_t rows_per_tile,
                                  int dtype, void* stream) {
  using namespace mi;
  if (max_tiles == 0) return MI_OK;
  if (!out || !a_rows || !weights || !sorted_pairs || !tile_expert || !num_tiles || max_tiles < 0)
    return MI_ERR_INVALID_ARG;
  if (k <= 0 || k % 64 || n_out <= 0 || n_out % (gated ? 64 : 128)) return MI_ERR_UNSUPPORTED;
  if (!moe_rows_per_tile_ok(rows_per_tile)) return MI_ERR_INVALID_ARG;
  if (a_row_stride % 8 || out_row_stride % 4 || weight_expert_stride % 8) return MI_ERR_INVALID_ARG;
  if (((uintptr_t)a_rows | (uintptr_t)weights) & 15 || ((uintptr_t)out & 7)) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    // default: the decode-batch linear kernel's grouped form (skinny_gemm.hip: row-contiguous LDS-DMA weight images,
    // `ds_read_b128` fragments, loader / consumer waves); MI_MOE_GEMM=direct keeps the fragments-from-global kernel
    // below (A/B runs, and shapes outside the ring kernel's envelope)
    const char* e = getenv("MI_MOE_GEMM");
    if (!e || strcmp(e, "direct") != 0) {
      GroupedGemmArgs ga;
      ga.out = out; ga.a_rows = a_rows; ga.weights = weights;
      ga.sorted_pairs = sorted_pairs; ga.tile_expert = tile_expert; ga.num_tiles = num_tiles;
      ga.max_tiles = max_tiles; ga.gated = gated; ga.n_out = n_out; ga.k = k; ga.top_k = top_k;
      ga.rows_per_tile = rows_per_tile;
      ga.a_row_stride = a_row_stride; ga.out_row_stride = out_row_stride; ga.weight_expert_stride = weight_expert_stride;
      ga.dtype = dtype;
      const int rc = launch_grouped_skinny(ga, s);
      if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
  }
  if (rows_per_tile > 64) return MI_ERR_UNSUPPORTED;   // the fragments-from-global kernel has 32- and 64-row tiles only
  MoeGemmParams p;
  p.a_rows = reinterpret_cast<const uint16_t*>(a_rows);
  p.w = reinterpret_cast<const uint16_t*>(weights);
  p.out = reinterpret_cast<uint16_t*>(out);
  p.sorted_pairs = sorted_pairs;
  p.tile_expert = tile_expert;
  p.num_tiles = num_tiles;
  p.K = k;
  p.N_out = n_out;
  p.top_k = top_k;
  p.a_stride = a_row_stride;
  p.out_stride = out_row_stride;
  p.w_expert_stride = weight_expert_stride;
  const dim3 grid((max_tiles + 7) / 8 * 8, gated ? n_out / 64 : n_out / 128), block(256);
#define MI_MOE_GEMM(TAG, G, M) hipLaunchKernelGGL((moe_gemm_kernel<TAG, G, M>), grid, block, 0, s, p)
#define MI_MOE_GEMM_T(TAG)                                \
  do {                                                    \
    if (gated && rows_per_tile == 32) MI_MOE_GEMM(TAG, true, 2);        \
    else if (gated) MI_MOE_GEMM(TAG, true, 4);            \
    else if (rows_per_tile == 32) MI_MOE_GEMM(TAG, false, 2);           \
    else MI_MOE_GEMM(TAG, false, 4);                      \
  } while (0)
  if (dtype == MI_DTYPE_BF16) MI_MOE_GEMM_T(bf16_tag);
  else if (dtype == MI_DTYPE_F16) MI_MOE_GEMM_T(f16_tag);
  else return MI_ERR_UNSUPPORTED;
#undef MI_MOE_GEMM_T
#undef MI_MOE_GEMM
  return launch_status();
}

extern "C" int mi_moe_combine(void* out, const void* partial, const int32_t* row_of_pair, const float* topk_weights,
                              const void* shared_out, const void* x, const void* shared_gate_weight,
                              int32_t num_tokens, int32_t hidden, int32_t top_k, int64_t partial_row_stride,
                              int64_t out_row_stride, int64_t x_row_stride, int64_t shared_row_stride, int dtype,
                              void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!out || !partial || !row_of_pair || !topk_weights || num_tokens < 0 || hidden <= 0 || top_k <= 0)
    return MI_ERR_INVALID_ARG;
  if (hidden % 4 || partial_row_stride % 4 || out_row_stride % 4 || shared_row_stride % 4) return MI_ERR_UNSUPPORTED;
  if (shared_gate_weight && (!shared_out || !x)) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto o = reinterpret_cast<uint16_t*>(out);
  auto pp = reinterpret_cast<const uint16_t*>(partial);
  auto sh = reinterpret_cast<const uint16_t*>(shared_out);
  auto xx = reinterpret_cast<const uint16_t*>(x);
  auto gw = reinterpret_cast<const uint16_t*>(shared_gate_weight);
  if (dtype == MI_DTYPE_BF16)
    hipLaunchKernelGGL((moe_combine_kernel<bf16_tag>), dim3(num_tokens), dim3(256), 0, s, o, pp, row_of_pair, topk_weights, sh, xx, gw, hidden, top_k, partial_row_stride, out_row_stride, x_row_stride, shared_row_stride);
  else if (dtype == MI_DTYPE_F16)
    hipLaunchKernelGGL((moe_combine_kernel<f16_tag>), dim3(num_tokens), dim3(256), 0, s, o, pp, row_of_pair, topk_weights, sh, xx, gw, hidden, top_k, partial_row_stride, out_row_stride, x_row_stride, shared_row_stride);
  else
    return MI_ERR_UNSUPPORTED;
  return launch_status();
}
