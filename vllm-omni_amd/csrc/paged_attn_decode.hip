// A5 — paged-attention decode for gfx950 (MI355X).
//
// Replaces the decode half of upstream `Attention.forward` reached from
// vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:154.
//
// Shape of the problem: per generated token and KV head, G = n_q/n_kv (7 for the
// Qwen2.5-Omni talker) query rows against `ctx` cached keys/values: 3.5 FLOP per KV
// byte, i.e. an HBM-bound stream.  Design:
//   * unit of work = one WAVE: (sequence, G query heads sharing one KV head, one
//     context partition).  No LDS, no barriers: K and V go HBM -> VGPR with 16 B per
//     lane (`global_load_dwordx4`), one wave-instruction = 1 KiB = 64/LPT whole
//     token rows of one head, so every fetched 128-B line is used in full.
//   * lane (g, c): token g of the load, dims [8c, 8c+8).  QK^T = packed bf16/f16
//     `v_dot2c` on the raw cache words (no up-conversion), partial sums all-reduced
//     over the LPT lanes of a token with 3-4 `v_add_f32_dpp` (DPP row ops).
//   * online softmax in fp32 in the exp2 domain; rescale of the accumulators only
//     when some running max moved (wave vote), which after the first tiles is rare.
//   * PV either with P kept fp32 (v_fma, exact) or, default, P rounded to the cache
//     dtype and contracted two tokens at a time with `v_dot2c` (what the reference's
//     Triton kernel does: `tl.dot(P.to(V.dtype), V)`).
//   * tiles of 16 tokens, register double-buffered: the loads of tile t+1 are in
//     flight while tile t is computed (8 KiB per wave in flight).
//   * partitions of the context are merged by `decode_reduce_kernel`.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "decode_params.h"

namespace mi {
namespace {

constexpr int kTile = 16;

// 16-byte K/V load.  Default cache policy: the non-temporal policy that is worth +8 % on the
// matrix-core kernel's LDS-DMA stream measured 5-18 % SLOWER on this kernel's register loads.
__device__ __forceinline__ uint4 load_stream16(const char* p) {
  return *reinterpret_cast<const uint4*>(p);
}
constexpr float kNegBig = -3.0e38f;

template <typename Tag, int HD, int G, bool PACKED_PV>
__global__ __launch_bounds__(256, 2) void decode_kernel(const DecodeParams p) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  constexpr int LPT = HD / 8;      // lanes per token row
  constexpr int TPL = 64 / LPT;    // token rows per wave-load
  constexpr int NL = kTile / TPL;  // loads per tile (per K and per V)
  constexpr int NG = TPL;          // lane groups per wave

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (wave >= p.total_waves) return;
  const int part = wave % p.num_parts;
  const int hw = (wave / p.num_parts) % (p.num_q_heads / G);  // q-head chunk
  const int seq = wave / (p.num_parts * (p.num_q_heads / G));
  const int qh0 = hw * G;
  const int kvh = qh0 / p.kv_group;

  const int g = lane / LPT;
  const int c = lane % LPT;

  const int seq_len = p.seq_lens[seq];
  const int num_tiles = (seq_len + kTile - 1) / kTile;
  const int t_begin = part * p.part_tiles;
  const int t_end = min(t_begin + p.part_tiles, num_tiles);
  const bool direct = p.num_parts == 1;

  uint16_t* const out16 = reinterpret_cast<uint16_t*>(p.out);
  float* const out32 = reinterpret_cast<float*>(p.out);

  if (t_begin >= t_end) {
    // Empty partition.  Only the direct path has to produce output here
    // (seq_len == 0 -> zeros); otherwise the reduce kernel skips this slot.
    if (direct) {
#pragma unroll
      for (int h = 0; h < G; ++h) {
        if (g == h % NG) {
          const int64_t o = (int64_t)seq * p.out_stride + (int64_t)(qh0 + h) * HD + c * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (p.out_f32) out32[o + e] = 0.f; else out16[o + e] = 0;
          }
        }
      }
    }
    return;
  }

  const int32_t* const bt_row = p.block_table + (int64_t)seq * p.block_table_stride;
  const int bs_mask = (1 << p.bs_log2) - 1;
  const int last_blk = (seq_len - 1) >> p.bs_log2;
  const int64_t lane_off = (int64_t)g * p.tok_stride + (int64_t)kvh * p.head_stride + c * 8;
  const int limit = min(seq_len, t_end * kTile);

  // ---- q: G heads, this lane's 8 dims, kept packed for dot2 -------------
  uint4 qv[G];
  {
    const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)seq * p.q_stride +
                         (int64_t)qh0 * HD + c * 8;
#pragma unroll
    for (int h = 0; h < G; ++h) qv[h] = *reinterpret_cast<const uint4*>(qb + h * HD);
  }

  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    m[h] = kNegBig;
    l[h] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;
  }

  // Block ids of this partition live in ONE VGPR (lane i = i-th block of the partition; the host
  // caps a partition at 64 blocks) and are picked with v_readlane: the loop then issues nothing
  // but K/V loads, so the in-order vmcnt queue never makes a tile wait behind a table fetch.
  const int first_blk = (t_begin * kTile) >> p.bs_log2;
  const int win_ids = bt_row[min(first_blk + lane, last_blk)];

  // Addressing: the block base is wave-uniform (SGPR pair), the per-lane part is ONE constant
  // 32-bit byte offset -> `global_load_dwordx4 v, v_off, s[base]` with no per-load VALU math.
  const uint32_t lane_off_b = (uint32_t)(lane_off * 2);
  const char* const kbase_b = reinterpret_cast<const char*>(p.k_cache);
  const char* const vbase_b = reinterpret_cast<const char*>(p.v_cache);
  constexpr int CH = NL >= 4 ? 2 : NL;  // wave-loads per chunk (8 tokens at head_dim 128)
  // byte offset (wave-uniform) of the rows of wave-load i of tile t
  auto row_base = [&](int t, int i) -> int64_t {
    const int tok0 = t * kTile + i * TPL;
    // clamp: rows past seq_len must not walk into unallocated table entries
    const int blk = min(tok0 >> p.bs_log2, last_blk);
    const int phys = __builtin_amdgcn_readlane(win_ids, blk - first_blk);
    return ((int64_t)phys * p.blk_stride + (int64_t)(tok0 & bs_mask) * p.tok_stride) * 2;
  };
  // Loads are issued per chunk, in CONSUMPTION order (K chunk, then V chunk): vmcnt retires in
  // order, so the wait in front of a chunk never covers loads issued after it.
  auto load_k = [&](int t, int c0, uint4(&kr)[NL]) {
#pragma unroll
    for (int i = 0; i < CH; ++i)
      kr[c0 + i] = load_stream16(kbase_b + row_base(t, c0 + i) + lane_off_b);
  };
  auto load_v = [&](int t, int c0, uint4(&vr)[NL]) {
#pragma unroll
    for (int i = 0; i < CH; ++i)
      vr[c0 + i] = load_stream16(vbase_b + row_base(t, c0 + i) + lane_off_b);
  };
  auto load_tile = [&](int t, uint4(&kr)[NL], uint4(&vr)[NL]) {
#pragma unroll
    for (int c0 = 0; c0 < NL; c0 += CH) {
      load_k(t, c0, kr);
      load_v(t, c0, vr);
    }
  };

  // A tile is consumed in chunks of CH wave-loads (8 tokens at head_dim 128): scores, online
  // softmax update and P.V per chunk, which halves the live score/probability registers.
  // Rolling ring: as soon as a chunk's registers have been consumed they are refilled with the
  // same chunk of tile `t_next` (two tiles ahead), so ~2 tiles (16 KiB) per wave stay in flight
  // with no registers beyond the two tile buffers.
  auto compute = [&](int t, uint4(&kr)[NL], uint4(&vr_in)[NL], int t_next) {
    const bool tail = (t + 1) * kTile > limit;  // wave-uniform
#pragma unroll
    for (int c0 = 0; c0 < NL; c0 += CH) {
      // ---- scores: s[h][i] = q_h . k_(token (c0+i)*TPL+g) -------------------
      float s[G][CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
#pragma unroll
        for (int h = 0; h < G; ++h) {
          float a = E::dot2(qv[h].x, kr[c0 + i].x, 0.f);
          a = E::dot2(qv[h].y, kr[c0 + i].y, a);
          a = E::dot2(qv[h].z, kr[c0 + i].z, a);
          a = E::dot2(qv[h].w, kr[c0 + i].w, a);
          s[h][i] = group_allreduce_sum<LPT>(a) * p.scale_log2e;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_k(t_next, c0, kr);  // K chunk consumed -> refill it now
      __builtin_amdgcn_sched_barrier(0);
      uint4 vr[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) vr[i] = vr_in[c0 + i];
      // rows at or past `limit` (end of the sequence, or of this partition when the pipeline
      // runs one tile over) contribute nothing: score -inf, V zeroed (0 * garbage != NaN)
      if (tail) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const bool valid = t * kTile + (c0 + i) * TPL + g < limit;
          if (!valid) {
            vr[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int h = 0; h < G; ++h) s[h][i] = -INFINITY;
          }
        }
      }
      // ---- online softmax (per lane group; groups merged after the loop) ------
      bool moved = false;
      float m_new[G];
#pragma unroll
      for (int h = 0; h < G; ++h) {
        float tm = s[h][0];
#pragma unroll
        for (int i = 1; i < CH; ++i) tm = fmaxf(tm, s[h][i]);
        m_new[h] = fmaxf(m[h], tm);
        moved |= (m_new[h] != m[h]);
      }
      if (__any(moved)) {
#pragma unroll
        for (int h = 0; h < G; ++h) {
          const float alpha = fast_exp2(m[h] - m_new[h]);
          m[h] = m_new[h];
          l[h] *= alpha;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[h][e] *= alpha;
        }
      }
#pragma unroll
      for (int h = 0; h < G; ++h) {
        float ls = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          s[h][i] = fast_exp2(s[h][i] - m[h]);  // s now holds the probabilities
          ls += s[h][i];
        }
        l[h] += ls;
      }
      // ---- PV -------------------------------------------------------------------
      if constexpr (PACKED_PV && CH >= 2) {
#pragma unroll
        for (int i = 0; i < CH; i += 2) {
          uint32_t pp[G];
#pragma unroll
          for (int h = 0; h < G; ++h) pp[h] = E::pack(s[h][i], s[h][i + 1]);
          const uint32_t a[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w};
          const uint32_t b[4] = {vr[i + 1].x, vr[i + 1].y, vr[i + 1].z, vr[i + 1].w};
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            // (tok i, tok i+1) pairs for dims 2w and 2w+1
            const uint32_t lo = __builtin_amdgcn_perm(b[w], a[w], 0x05040100u);
            const uint32_t hi = __builtin_amdgcn_perm(b[w], a[w], 0x07060302u);
#pragma unroll
            for (int h = 0; h < G; ++h) {
              acc[h][2 * w] = E::dot2(pp[h], lo, acc[h][2 * w]);
              acc[h][2 * w + 1] = E::dot2(pp[h], hi, acc[h][2 * w + 1]);
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const uint32_t a[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w};
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float v0 = E::lo(a[w]), v1 = E::hi(a[w]);
#pragma unroll
            for (int h = 0; h < G; ++h) {
              acc[h][2 * w] = __builtin_fmaf(s[h][i], v0, acc[h][2 * w]);
              acc[h][2 * w + 1] = __builtin_fmaf(s[h][i], v1, acc[h][2 * w + 1]);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_v(t_next, c0, vr_in);  // V chunk consumed (vr holds the copy) -> refill it now
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- main loop: register double buffer, static indexing ----------------
  // NO branch around any load (hipcc otherwise waits right after issuing them, §5 "Three .s-level
  // traps" (c) of the programming guide): tile indices are clamped instead, and a tile past the
  // partition is neutralised by the `limit` mask in compute().
  uint4 kA[NL], vA[NL], kB[NL], vB[NL];
  const int t_last = t_end - 1;
  load_tile(t_begin, kA, vA);
  load_tile(min(t_begin + 1, t_last), kB, vB);
  __builtin_amdgcn_sched_barrier(0);
  for (int t = t_begin; t < t_end; t += 2) {
    compute(t, kA, vA, min(t + 2, t_last));
    compute(t + 1, kB, vB, min(t + 3, t_last));
  }

  // ---- merge the NG lane groups of the wave -------------------------------
#pragma unroll
  for (int h = 0; h < G; ++h) {
    float mm = m[h];
#pragma unroll
    for (int o = LPT; o < 64; o <<= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));
    const float f = fast_exp2(m[h] - mm);
    m[h] = mm;
    float ll = l[h] * f;
#pragma unroll
    for (int o = LPT; o < 64; o <<= 1) ll += __shfl_xor(ll, o, 64);
    l[h] = ll;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[h][e] * f;
#pragma unroll
      for (int o = LPT; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
      acc[h][e] = a;
    }
  }

  if (direct) {
#pragma unroll
    for (int h = 0; h < G; ++h) {
      if (g == h % NG) {
        const float inv = 1.f / l[h];
        const int64_t o = (int64_t)seq * p.out_stride + (int64_t)(qh0 + h) * HD + c * 8;
        if (p.out_f32) {
          float4 a = {acc[h][0] * inv, acc[h][1] * inv, acc[h][2] * inv, acc[h][3] * inv};
          float4 b = {acc[h][4] * inv, acc[h][5] * inv, acc[h][6] * inv, acc[h][7] * inv};
          *reinterpret_cast<float4*>(out32 + o) = a;
          *reinterpret_cast<float4*>(out32 + o + 4) = b;
        } else {
          uint4 r;
          r.x = E::pack(acc[h][0] * inv, acc[h][1] * inv);
          r.y = E::pack(acc[h][2] * inv, acc[h][3] * inv);
          r.z = E::pack(acc[h][4] * inv, acc[h][5] * inv);
          r.w = E::pack(acc[h][6] * inv, acc[h][7] * inv);
          *reinterpret_cast<uint4*>(out16 + o) = r;
        }
      }
    }
  } else {
#pragma unroll
    for (int h = 0; h < G; ++h) {
      if (g == h % NG) {
        const int64_t slot = ((int64_t)seq * p.num_q_heads + (qh0 + h)) * p.num_parts + part;
        float* po = p.part_o + slot * HD + c * 8;
        *reinterpret_cast<float4*>(po) = make_float4(acc[h][0], acc[h][1], acc[h][2], acc[h][3]);
        *reinterpret_cast<float4*>(po + 4) = make_float4(acc[h][4], acc[h][5], acc[h][6], acc[h][7]);
        if (c == 0) *reinterpret_cast<float2*>(p.part_ml + slot * 2) = make_float2(m[h], l[h]);
      }
    }
  }
}

// Merge the partitions of one (sequence, q head): one wave per pair.
template <typename Tag, int HD>
__global__ __launch_bounds__(64) void decode_reduce_kernel(const DecodeParams p) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  constexpr int DPL = HD / 64;  // dims per lane
  const int qh = blockIdx.x;
  const int seq = blockIdx.y;
  const int lane = threadIdx.x;
  const int seq_len = p.seq_lens[seq];
  const int num_tiles = (seq_len + kTile - 1) / kTile;
  const int np = min(p.num_parts, (num_tiles + p.part_tiles - 1) / p.part_tiles);
  const int64_t slot0 = ((int64_t)seq * p.num_q_heads + qh) * p.num_parts;
  const float2* ml = reinterpret_cast<const float2*>(p.part_ml) + slot0;
  // Eight partitions per trip, every load of a trip issued before the first use (a loop that loads, uses and loops is
  // `np` dependent round trips: 7.5 us for 16 partitions at batch 8); the sums run in partition order as before.
  constexpr int CH = 8;
  if (p.num_parts <= CH) {
    // The common case (the auto rule gives <= 8 partitions up to 64k tokens): every load of the merge — the (m, l) pairs
    // and the partial rows of ALL the launch's partitions — is issued at once, beside the seq_lens load instead of
    // behind it (three dependent round trips become one); partitions the sequence does not reach hold stale bytes and
    // are dropped by a select AFTER the load.  Same sums in the same order as the loops below.
    float2 x[CH];
    float po[CH][DPL];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const bool have = j < p.num_parts;   // launch parameter: no dependence on seq_lens
      x[j] = have ? ml[j] : make_float2(kNegBig, 0.f);
      const float* src = p.part_o + (slot0 + (have ? j : 0)) * HD + lane * DPL;
#pragma unroll
      for (int d = 0; d < DPL; ++d) po[j][d] = have ? src[d] : 0.f;
    }
    float mm1 = kNegBig;
#pragma unroll
    for (int j = 0; j < CH; ++j) mm1 = fmaxf(mm1, j < np ? x[j].x : kNegBig);
    float L1 = 0.f, a1[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) a1[d] = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (j < np) {
        const float w = fast_exp2(x[j].x - mm1);
        L1 += w * x[j].y;
#pragma unroll
        for (int d = 0; d < DPL; ++d) a1[d] += w * po[j][d];
      }
    }
    const float inv1 = np > 0 ? 1.f / L1 : 0.f;
    const int64_t o1 = (int64_t)seq * p.out_stride + (int64_t)qh * HD + lane * DPL;
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
      const float r = a1[d] * inv1;
      if (p.out_f32) reinterpret_cast<float*>(p.out)[o1 + d] = r;
      else reinterpret_cast<uint16_t*>(p.out)[o1 + d] = E::to_bits(r);
    }
    return;
  }
  float mm = kNegBig;
  for (int i0 = 0; i0 < np; i0 += CH) {
    float2 x[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) x[j] = i0 + j < np ? ml[i0 + j] : make_float2(kNegBig, 0.f);
#pragma unroll
    for (int j = 0; j < CH; ++j) mm = fmaxf(mm, x[j].x);
  }
  float L = 0.f, a[DPL];
#pragma unroll
  for (int d = 0; d < DPL; ++d) a[d] = 0.f;
  for (int i0 = 0; i0 < np; i0 += CH) {
    float2 x[CH];
    float po[CH][DPL];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const bool on = i0 + j < np;
      x[j] = on ? ml[i0 + j] : make_float2(kNegBig, 0.f);
      const float* src = p.part_o + (slot0 + (on ? i0 + j : 0)) * HD + lane * DPL;
#pragma unroll
      for (int d = 0; d < DPL; ++d) po[j][d] = on ? src[d] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (i0 + j < np) {
        const float w = fast_exp2(x[j].x - mm);
        L += w * x[j].y;
#pragma unroll
        for (int d = 0; d < DPL; ++d) a[d] += w * po[j][d];
      }
    }
  }
  const float inv = np > 0 ? 1.f / L : 0.f;
  const int64_t o = (int64_t)seq * p.out_stride + (int64_t)qh * HD + lane * DPL;
#pragma unroll
  for (int d = 0; d < DPL; ++d) {
    const float r = a[d] * inv;
    if (p.out_f32) reinterpret_cast<float*>(p.out)[o + d] = r;
    else reinterpret_cast<uint16_t*>(p.out)[o + d] = E::to_bits(r);
  }
}

template <typename Tag, int HD, int G>
int launch_decode(const DecodeParams& p, bool packed, hipStream_t stream) {
  const int grid = (p.total_waves + 3) / 4;
  if (packed)
    hipLaunchKernelGGL((decode_kernel<Tag, HD, G, true>), dim3(grid), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((decode_kernel<Tag, HD, G, false>), dim3(grid), dim3(256), 0, stream, p);
  return launch_status();
}

template <typename Tag, int HD>
int launch_reduce(const DecodeParams& p, hipStream_t stream) {
  if (p.num_parts > 1)
    hipLaunchKernelGGL((decode_reduce_kernel<Tag, HD>), dim3(p.num_q_heads, p.num_seqs), dim3(64),
                       0, stream, p);
  return launch_status();
}

template <typename Tag, int HD>
int dispatch_group(const DecodeParams& p, int G, bool packed, hipStream_t stream) {
  switch (G) {
    case 1: return launch_decode<Tag, HD, 1>(p, packed, stream);
    case 2: return launch_decode<Tag, HD, 2>(p, packed, stream);
    case 3: return launch_decode<Tag, HD, 3>(p, packed, stream);
    case 4: return launch_decode<Tag, HD, 4>(p, packed, stream);
    case 5: return launch_decode<Tag, HD, 5>(p, packed, stream);
    case 6: return launch_decode<Tag, HD, 6>(p, packed, stream);
    case 7: return launch_decode<Tag, HD, 7>(p, packed, stream);
    case 8: return launch_decode<Tag, HD, 8>(p, packed, stream);
  }
  return MI_ERR_UNSUPPORTED;
}

// q heads handled per wave: the largest divisor of the GQA group that is <= limit
int heads_per_wave(int kv_group, int limit) {
  for (int g = limit; g >= 1; --g)
    if (kv_group % g == 0) return g;
  return 1;
}

int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace
}  // namespace mi

extern "C" int32_t mi_paged_attn_decode_auto_partition(int32_t num_seqs, int32_t num_kv_heads,
                                                       int32_t max_seq_len, int32_t head_dim) {
  // Measured on MI355X (tools/bench_decode_kernel.py; profiles/r03e_decode_long_parts*.log: B x ctx from 32 x 4096 to
  // 256 x 8192 and 64 x 16384, 4 / 2 / 1 KV heads): every partition costs a prologue (Q fragments, block-id window,
  // pipeline fill), an fp32 partial write and its share of the merge launch, so take the FEWEST partitions that still
  // leave >= 512 waves (two per CU) — ONE when (sequences x KV heads) >= 512: direct output, no partials, no merge
  // (B 256 / ctx 4096: 315 us against 340 with four partitions of 1024; fewer than 512 waves loses: 256 waves run at
  // 55 % of that rate).  Never below 256 tokens, never above 8192 (a bound on the tail when ONE sequence of a batch is
  // far longer than the rest: one wave streams its whole partition).
  // head_dim 64: a wave's ring holds half the bytes, so twice the waves (>= 1024: 128 sequences x 2 heads x ctx 4096 run
  // 46.6 us in four partitions against 54.5 in two; 512 x 2 heads stay at ONE: profiles/r03zb_decode_parts_d64.log)
  if (num_seqs <= 0 || num_kv_heads <= 0 || max_seq_len <= 0) return 256;
  const int64_t pairs = (int64_t)num_seqs * num_kv_heads;
  const int64_t want = head_dim > 0 && head_dim <= 64 ? 1024 : 512;
  const int64_t parts = (want + pairs - 1) / pairs;
  int64_t ps = mi::round_up((int32_t)((max_seq_len + parts - 1) / parts), 32);
  if (ps < 256) ps = 256;
  if (ps > 8192) ps = 8192;
  if (ps > mi::round_up(max_seq_len, 32)) ps = mi::round_up(max_seq_len, 32);
  return (int32_t)ps;  // kernels that keep a partition's block ids in ONE register cap it at 64 blocks (below)
}

// The partition size a call will really use: `requested` (0 = auto).  The matrix-core kernel (flags: PACKED_PV without
// FORCE_VALU, block_size >= 16, a multiple of 32 tokens) walks its block ids window by window and takes any length;
// the VALU kernel keeps a partition's block ids in one 64-lane register: capped at 64 KV blocks.
extern "C" int32_t mi_paged_attn_decode_effective_partition(int32_t num_seqs, int32_t num_kv_heads,
                                                            int32_t max_seq_len, int32_t head_dim,
                                                            int32_t block_size, int32_t requested, int32_t flags) {
  if (block_size <= 0) return 0;
  int32_t ps = requested > 0 ? requested
                             : mi_paged_attn_decode_auto_partition(num_seqs, num_kv_heads, max_seq_len, head_dim);
  const bool any_length = (flags & MI_DECODE_PACKED_PV) && !(flags & MI_DECODE_FORCE_VALU) && block_size >= 16 &&
                          ps % 32 == 0;
  const int32_t cap = 64 * block_size;
  if (!any_length && ps > cap) ps = cap;
  return ps;
}

extern "C" int64_t mi_paged_attn_decode_workspace_bytes(int32_t num_seqs, int32_t num_q_heads,
                                                        int32_t head_dim, int32_t max_seq_len,
                                                        int32_t partition_size) {
  if (num_seqs <= 0 || num_q_heads <= 0 || head_dim <= 0 || max_seq_len <= 0 ||
      partition_size <= 0)
    return 0;
  const int64_t parts = (max_seq_len + partition_size - 1) / partition_size;
  if (parts <= 1) return 0;
  // the arrival counters of the fused merge (a region of FIXED size at the front, so that calls of different batch
  // sizes and context bounds on one workspace never put partial sums where another call keeps its counters), then
  // the partial (O, m, l) triples
  return MI_DECODE_COUNTER_BYTES + (int64_t)num_seqs * num_q_heads * parts * (head_dim + 2) * (int64_t)sizeof(float);
}

namespace {
// RoPE + KV write of the step's token inside the launch (mi_rope_cache_paged_attn_decode)
struct RopeFusion {
  const void* k;
  const void* v;
  const int64_t* positions;
  const int64_t* slot_mapping;
  const float* cos_sin;
  int32_t max_position;
  int64_t k_stride, v_stride;
  const void* q_norm_w;
  const void* k_norm_w;
  float norm_eps;
};

int decode_call(void* out, const void* q, const void* k_cache, const void* v_cache, const int32_t* block_table,
                const int32_t* seq_lens, int32_t num_seqs, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                int32_t block_size, int32_t block_table_stride, int32_t max_seq_len, float scale, int64_t q_token_stride,
                int64_t out_token_stride, int64_t cache_block_stride, int64_t cache_token_stride,
                int64_t cache_head_stride, int32_t partition_size, void* workspace, int64_t workspace_bytes, int dtype,
                int out_dtype, int32_t flags, void* stream, const RopeFusion* rope) {
  using namespace mi;
  if (num_seqs == 0) return MI_OK;
  if (!out || !q || !k_cache || !v_cache || !block_table || !seq_lens) return MI_ERR_INVALID_ARG;
  if (num_seqs < 0 || num_q_heads <= 0 || num_kv_heads <= 0 || max_seq_len <= 0 ||
      block_table_stride <= 0)
    return MI_ERR_INVALID_ARG;
  if (num_q_heads % num_kv_heads != 0) return MI_ERR_INVALID_ARG;
  if (head_dim != 64 && head_dim != 128) return MI_ERR_UNSUPPORTED;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (out_dtype != dtype && out_dtype != MI_DTYPE_F32) return MI_ERR_UNSUPPORTED;
  const int tpl = 64 / (head_dim / 8);
  if (!is_pow2(block_size) || block_size < tpl) return MI_ERR_UNSUPPORTED;
  // 16-byte vector access: every stride must keep 8-element alignment
  if (q_token_stride % 8 || out_token_stride % 8 || cache_block_stride % 8 ||
      cache_token_stride % 8 || cache_head_stride % 8)
    return MI_ERR_INVALID_ARG;
  if (((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out) & 15)
    return MI_ERR_INVALID_ARG;

  partition_size = mi_paged_attn_decode_effective_partition(num_seqs, num_kv_heads, max_seq_len, head_dim,
                                                            block_size, partition_size, flags);
  if (partition_size <= 0 || partition_size % kTile) return MI_ERR_INVALID_ARG;
  const int num_parts = (max_seq_len + partition_size - 1) / partition_size;
  const int64_t need = mi_paged_attn_decode_workspace_bytes(num_seqs, num_q_heads, head_dim,
                                                            max_seq_len, partition_size);
  if (need > 0 && (!workspace || workspace_bytes < need)) return MI_ERR_WORKSPACE;

  const int kv_group = num_q_heads / num_kv_heads;
  // Kernel choice: with P allowed to be rounded to the cache dtype (MI_DECODE_PACKED_PV) the
  // matrix-core kernel is used wherever it is built; otherwise (or when forced) the VALU kernel.
  const bool use_mfma = (flags & MI_DECODE_PACKED_PV) && !(flags & MI_DECODE_FORCE_VALU) &&
                        block_size >= 16 && partition_size % 32 == 0;
  const int G = heads_per_wave(kv_group, use_mfma ? 16 : 8);

  DecodeParams p;
  p.out = out;
  p.q = q;
  p.k_cache = k_cache;
  p.v_cache = v_cache;
  p.block_table = block_table;
  p.seq_lens = seq_lens;
  p.part_o = need > 0 ? reinterpret_cast<float*>(static_cast<char*>(workspace) + MI_DECODE_COUNTER_BYTES) : nullptr;
  p.part_ml = p.part_o ? p.part_o + (int64_t)num_seqs * num_q_heads * num_parts * head_dim : nullptr;
  // only the matrix-core kernel merges in its own launch, and only when the caller vouches that the counter region
  // was zero before the first launch on this workspace (MI_DECODE_FUSED_MERGE)
  p.part_count = (p.part_ml && use_mfma && (flags & MI_DECODE_FUSED_MERGE) &&
                  (int64_t)num_seqs * num_q_heads * (int64_t)sizeof(int32_t) <= MI_DECODE_COUNTER_BYTES)
                     ? reinterpret_cast<int32_t*>(workspace)
                     : nullptr;
  p.num_seqs = num_seqs;
  p.num_q_heads = num_q_heads;
  p.kv_group = kv_group;
  p.bs_log2 = __builtin_ctz((unsigned)block_size);
  p.block_table_stride = block_table_stride;
  p.num_parts = num_parts;
  p.part_tiles = partition_size / kTile;
  const int64_t total = (int64_t)num_seqs * (num_q_heads / G) * num_parts;
  if (total > INT32_MAX) return MI_ERR_UNSUPPORTED;
  p.total_waves = (int32_t)total;
  p.out_f32 = out_dtype == MI_DTYPE_F32;
  p.scale_log2e = scale * kLog2e;
  p.q_stride = q_token_stride;
  p.out_stride = out_token_stride;
  p.blk_stride = cache_block_stride;
  p.tok_stride = cache_token_stride;
  p.head_stride = cache_head_stride;

  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // The single-rounding dot2 form of the VALU kernel only runs when asked for by name (PACKED_PV + FORCE_VALU, A/B
  // measurements): where the matrix-core kernel is not built (block 8, odd partitions) PACKED_PV falls back to the
  // fp32-P VALU kernel, so every production launch is inside the rtol 1e-3 / atol 1e-4 bound.
  const bool packed = (flags & MI_DECODE_PACKED_PV) && (flags & MI_DECODE_FORCE_VALU);
  int rc;
  if (rope) {
    // built for the production form only: matrix-core kernel, nt streams, P as a hi + lo pair
    if (!use_mfma || !(flags & MI_DECODE_NONTEMPORAL) || (flags & MI_DECODE_ROUNDED_P)) return MI_ERR_UNSUPPORTED;
    p.k_new = rope->k; p.v_new = rope->v;
    p.positions = rope->positions; p.slot_mapping = rope->slot_mapping;
    p.cos_sin = rope->cos_sin; p.max_position = rope->max_position;
    p.kn_stride = rope->k_stride; p.vn_stride = rope->v_stride;
    p.q_norm_w = rope->q_norm_w; p.k_norm_w = rope->k_norm_w; p.norm_eps = rope->norm_eps;
    rc = launch_decode_mfma_rope(p, G, head_dim, dtype, s);
  } else if (use_mfma) {
    rc = launch_decode_mfma(p, G, head_dim, dtype, (flags & MI_DECODE_NONTEMPORAL) != 0,
                            (flags & MI_DECODE_ROUNDED_P) == 0, s);
  } else if (dtype == MI_DTYPE_BF16) {
    rc = head_dim == 128 ? dispatch_group<bf16_tag, 128>(p, G, packed, s)
                         : dispatch_group<bf16_tag, 64>(p, G, packed, s);
  } else {
    rc = head_dim == 128 ? dispatch_group<f16_tag, 128>(p, G, packed, s)
                         : dispatch_group<f16_tag, 64>(p, G, packed, s);
  }
  if (rc != MI_OK) return rc;
  if (p.part_count) return MI_OK;   // merged inside the launch
  if (flags & MI_DECODE_SKIP_MERGE) return MI_OK;   // measurement of the attention kernel alone
  if (dtype == MI_DTYPE_BF16)
    return head_dim == 128 ? launch_reduce<bf16_tag, 128>(p, s) : launch_reduce<bf16_tag, 64>(p, s);
  return head_dim == 128 ? launch_reduce<f16_tag, 128>(p, s) : launch_reduce<f16_tag, 64>(p, s);
}
}  // namespace

extern "C" int mi_paged_attn_decode(void* out, const void* q, const void* k_cache,
                                    const void* v_cache, const int32_t* block_table,
                                    const int32_t* seq_lens, int32_t num_seqs,
                                    int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                    int32_t block_size, int32_t block_table_stride,
                                    int32_t max_seq_len, float scale, int64_t q_token_stride,
                                    int64_t out_token_stride, int64_t cache_block_stride,
                                    int64_t cache_token_stride, int64_t cache_head_stride,
                                    int32_t partition_size, void* workspace,
                                    int64_t workspace_bytes, int dtype, int out_dtype,
                                    int32_t flags, void* stream) {
  return decode_call(out, q, k_cache, v_cache, block_table, seq_lens, num_seqs, num_q_heads, num_kv_heads, head_dim,
                     block_size, block_table_stride, max_seq_len, scale, q_token_stride, out_token_stride,
                     cache_block_stride, cache_token_stride, cache_head_stride, partition_size, workspace,
                     workspace_bytes, dtype, out_dtype, flags, stream, nullptr);
}

extern "C" int mi_rope_cache_paged_attn_decode(void* out, const void* q, const void* k, const void* v,
                                               const int64_t* positions, const float* cos_sin_cache,
                                               int32_t max_position, int32_t rot_dim, const void* q_norm_weight,
                                               const void* k_norm_weight, float norm_eps, void* k_cache, void* v_cache,
                                               const int64_t* slot_mapping, const int32_t* block_table,
                                               const int32_t* seq_lens, int32_t num_seqs, int32_t num_q_heads,
                                               int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                               int32_t block_table_stride, int32_t max_seq_len, float scale,
                                               int64_t q_token_stride, int64_t k_token_stride, int64_t v_token_stride,
                                               int64_t out_token_stride, int64_t cache_block_stride,
                                               int64_t cache_token_stride, int64_t cache_head_stride,
                                               int32_t partition_size, void* workspace, int64_t workspace_bytes,
                                               int dtype, int out_dtype, int32_t flags, void* stream) {
  if (num_seqs == 0) return MI_OK;
  if (!k || !v || !positions || !cos_sin_cache || !slot_mapping || max_position <= 0) return MI_ERR_INVALID_ARG;
  if (rot_dim != head_dim) return MI_ERR_UNSUPPORTED;   // partial rotary: use the two launches
  if ((q_norm_weight == nullptr) != (k_norm_weight == nullptr)) return MI_ERR_INVALID_ARG;
  if (((uintptr_t)q_norm_weight | (uintptr_t)k_norm_weight) & 15) return MI_ERR_INVALID_ARG;
  if (k_token_stride % 8 || v_token_stride % 8 || (((uintptr_t)k | (uintptr_t)v | (uintptr_t)cos_sin_cache) & 15))
    return MI_ERR_INVALID_ARG;
  const RopeFusion rf{k, v, positions, slot_mapping, cos_sin_cache, max_position, k_token_stride, v_token_stride,
                      q_norm_weight, k_norm_weight, norm_eps};
  return decode_call(out, q, k_cache, v_cache, block_table, seq_lens, num_seqs, num_q_heads, num_kv_heads, head_dim,
                     block_size, block_table_stride, max_seq_len, scale, q_token_stride, out_token_stride,
                     cache_block_stride, cache_token_stride, cache_head_stride, partition_size, workspace,
                     workspace_bytes, dtype, out_dtype, flags, stream, &rf);
}
