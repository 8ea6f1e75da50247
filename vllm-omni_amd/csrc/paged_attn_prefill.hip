// A6 — paged-attention prefill / chunked prefill on the matrix cores (gfx950).
//
// Replaces the prefill half of upstream `Attention.forward` reached from
// vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:154: varlen causal
// attention of query chunks against the paged cache; query row i of a chunk of length L sits at
// absolute position seq_len - L + i and sees cache positions [0, that position].
//
// Same machinery as the decode kernel (paged_attn_decode_mfma.hip): 32-token K/V steps staged into
// swizzled LDS images by LDS-DMA, swapped QK^T (S^T = K . Q^T) so that a lane owns one query
// COLUMN and the softmax axis is lane-local, P^T as the B operand of O^T += V^T . P^T, V^T through
// `ds_read_b64_tr_b16`.  What differs:
//   * a workgroup = the G query heads that share one KV head (one wave per head, G <= 16 waves) x
//     16 consecutive query tokens; the 16 MFMA columns of a wave are its 16 query TOKENS;
//   * the K/V stage is shared by the workgroup: DMA issue is split over up to four waves, two raw
//     `s_barrier`s per step order "stage landed" and "stage consumed" (no __syncthreads: its
//     implied vmcnt(0) would drain the prefetched stage);
//   * the causal mask is per lane (column = query token); only steps that touch the diagonal
//     evaluate it; the context loop stops at the last position any column of the tile may see.
// FLOPs per layer: 2 * S^2 * d * n_q (causal).  MFMA-bound by design; K/V bytes are read once per
// (16 query tokens x G heads) and served from L2/MALL for any but the first tile of a sequence.
#include <cstdlib>

#include "common.h"
#include "mfma_attn_common.h"

namespace mi {
namespace {

struct PrefillParams {
  void* out;
  const void* q;
  const void* k_cache;
  const void* v_cache;
  const int32_t* block_table;
  const int32_t* seq_lens;
  const int32_t* query_start_loc;
  int32_t num_seqs, num_q_heads, kv_group, heads_per_wg, bs_log2, block_table_stride;
  float scale_log2e;
  int64_t q_stride, out_stride, blk_stride, tok_stride, head_stride;
};

constexpr int kQTile = 16;  // query tokens per workgroup

// G = waves per workgroup = query heads per KV head handled together; NI = waves that issue DMA.
template <typename Tag, int HD, int NI>
__global__ __launch_bounds__(512) void prefill_mfma_kernel(const PrefillParams p) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  constexpr int ROWB = HD * 2;
  constexpr int KS = HD / 32;
  constexpr int NT = HD / 16;
  constexpr int LPR = ROWB / 16;     // lanes (16-B chunks) per row: 16 / 8
  constexpr int RPI = 64 / LPR;      // rows per DMA wave-instruction: 4 / 8
  constexpr int IPS = 16 / RPI;      // DMA instructions per 16-row sub-tile: 4 / 2
  constexpr int IMG = kStep * ROWB;  // 8 KiB
  constexpr int STAGE = 2 * IMG;
  constexpr int DMA_PER_STAGE = 4 * IPS;        // 16 wave-instructions fill one stage
  constexpr int SHARE = DMA_PER_STAGE / NI;      // per issuing wave
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int seq = blockIdx.y;
  // blockIdx.z = (kv head, chunk of heads_per_wg query heads of its GQA group)
  const int chunks = p.kv_group / p.heads_per_wg;
  const int kvh = blockIdx.z / chunks;
  const int head = kvh * p.kv_group + (blockIdx.z % chunks) * p.heads_per_wg + wid;  // this wave's query head

  const int q0 = p.query_start_loc[seq];
  const int q_len = p.query_start_loc[seq + 1] - q0;
  const int seq_len = p.seq_lens[seq];
  const int tile0 = blockIdx.x * kQTile;  // first query row of this tile inside the chunk
  if (tile0 >= q_len) return;             // whole workgroup: uniform
  const int ctx_off = seq_len - q_len;    // absolute position of query row 0
  const int rows = min(kQTile, q_len - tile0);

  const int col = lane & 15;  // query token of the tile
  const int g = lane >> 4;
  const bool col_ok = col < rows;
  const int qpos = ctx_off + tile0 + col;                  // absolute position of my column
  const int limit = ctx_off + tile0 + rows;                // keys [0, limit) are visible to the tile
  const int t_end = (limit + kStep - 1) / kStep;
  const int diag_t = (ctx_off + tile0) / kStep;            // first step that can cross the diagonal

  // constant address space: the (wave-uniform) block-id fetch becomes a SCALAR load (lgkmcnt), so
  // it never sits in the vmcnt queue the DMA waits are counted on
  typedef const __attribute__((address_space(4))) int32_t cint32_t;
  cint32_t* const bt_row = (cint32_t*)(p.block_table + (int64_t)seq * p.block_table_stride);
  const int bs_mask = (1 << p.bs_log2) - 1;
  const int last_blk = (limit - 1) >> p.bs_log2;

  // ---- Q^T fragments: column = token tile0+col of head `head`, dims 32s + 8g .. -------------
  uint4 qf[KS];
  {
    const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) +
                         (int64_t)(q0 + tile0 + (col_ok ? col : 0)) * p.q_stride + (int64_t)head * HD + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      qf[s] = col_ok ? *reinterpret_cast<const uint4*>(qb + 32 * s) : make_uint4(0, 0, 0, 0);
  }

  // ---- per-lane constants (same images as the decode kernel) ----------------------------------
  uint32_t src_off[IPS];
#pragma unroll
  for (int ii = 0; ii < IPS; ++ii) {
    const int c = (lane % LPR) ^ swz_hd<HD>(RPI * ii + lane / LPR);  // 16*j does not change the swizzle
    src_off[ii] = (uint32_t)(((int64_t)(lane / LPR) * p.tok_stride + (int64_t)kvh * p.head_stride) * 2 + c * 16);
  }
  const int sw_h = swz_hd<HD>(col);
  uint32_t k_rd[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) k_rd[s] = (uint32_t)(col * ROWB + (((4 * s + g) ^ sw_h) * 16));
  const int q4 = (lane & 15) >> 2, pp = lane & 3;
  const int v_row = 4 * g + q4;
  const int sw_v = swz_hd<HD>(v_row);
  uint32_t v_rd[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
    v_rd[n] = (uint32_t)(IMG + v_row * ROWB + (((2 * n + (pp >> 1)) ^ sw_v) * 16) + 8 * (pp & 1));

  const char* const kbase_b = reinterpret_cast<const char*>(p.k_cache);
  const char* const vbase_b = reinterpret_cast<const char*>(p.v_cache);

  // DMA instruction `idx` of the stage of step t: [0, 2*IPS) K image, [2*IPS, 4*IPS) V image; inside an
  // image the index is IPS*j + ii (sub-tile j, row group ii).  Rows at or past `limit` of the V image take
  // the bytes of row limit-1 (their probability is exactly 0; 0 * stale NaN must not appear).
  auto issue_one = [&](int t, int st, int idx) {
    const bool is_v = idx >= 2 * IPS;
    const int k = idx % (2 * IPS);
    const int j = k / IPS, ii = k % IPS;
    const int row0 = t * kStep + 16 * j + RPI * ii;  // first token row of this instruction
    char* dst = smem + st * STAGE + (is_v ? IMG : 0) + (16 * j + RPI * ii) * ROWB;
    const int tok = min(row0 + lane / LPR, limit - 1);  // per lane (== row0 + lane>>4 off the tail)
    const int blk = __builtin_amdgcn_readfirstlane(min(min(row0, limit - 1) >> p.bs_log2, last_blk));
    const int phys = bt_row[blk];
    const int c = (lane % LPR) ^ swz_hd<HD>(RPI * ii + lane / LPR);
    const int64_t src = ((int64_t)phys * p.blk_stride + (int64_t)(tok & bs_mask) * p.tok_stride +
                         (int64_t)kvh * p.head_stride) * 2 + c * 16;
    __builtin_amdgcn_global_load_lds((gbl_void*)((is_v ? vbase_b : kbase_b) + src), (lds_void*)dst, 16, 0, 0);
  };
  auto issue = [&](int t, int st) {  // my share of the stage (waves >= NI issue nothing)
    if (wid < NI) {
#pragma unroll
      for (int k = 0; k < SHARE; ++k) issue_one(t, st, wid * SHARE + k);
    }
  };

  float m = -3.0e38f, l = 0.f;
  f32x4_t acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int t, auto stc) {
    constexpr int ST = decltype(stc)::value;
    constexpr int SB = ST * STAGE;
    u32x4_t k0[KS], k1[KS];
    if constexpr (KS == 4) lds_read_rows8<SB, SB + 16 * ROWB>(k_rd, k0, k1);
    else lds_read_rows4<SB, SB + 16 * ROWB>(k_rd, k0, k1);
    f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      s0 = mfma16<Tag>(__builtin_bit_cast(uint4, k0[s]), qf[s], s0);
      s1 = mfma16<Tag>(__builtin_bit_cast(uint4, k1[s]), qf[s], s1);
    }
    // lane: query column `col`, keys t*32 + 4g + r (s0[r]) and t*32 + 16 + 4g + r (s1[r])
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = s0[r] * p.scale_log2e;
      v[4 + r] = s1[r] * p.scale_log2e;
    }
    if (t >= diag_t) {  // causal mask, per column: key position <= query position
      const int kb = t * kStep + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kb + r > qpos) v[r] = -INFINITY;
        if (kb + 16 + r > qpos) v[4 + r] = -INFINITY;
      }
    }
    float mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    if (__any(m_new != m)) {
      const float alpha = fast_exp2(m - m_new);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] *= alpha;
    }
    float ls = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = fast_exp2(v[i] - m);
      ls += v[i];
    }
    l += ls;
    const u32x2_t pb0 = {E::pack(v[0], v[1]), E::pack(v[2], v[3])};  // P^T fragments of the K=16 MFMA
    const u32x2_t pb1 = {E::pack(v[4], v[5]), E::pack(v[6], v[7])};
#pragma unroll
    for (int nb = 0; nb < NT; nb += 4) {
      u32x2_t a0[4], a1[4];
      lds_read_tr8<SB, SB + 16 * ROWB>(&v_rd[nb], a0, a1);
#pragma unroll
      for (int n = 0; n < 4; ++n) {  // one transposed read == one A operand
        acc[nb + n] = mfma16_k16<Tag>(a0[n], pb0, acc[nb + n]);
        acc[nb + n] = mfma16_k16<Tag>(a1[n], pb1, acc[nb + n]);
      }
    }
  };

  // ---- pipeline.  Per step: [issue stage t+1] [my DMAs of stage t landed] barrier [compute t]
  //      barrier (stage t may be overwritten by the issue of step t+2).  The issuing waves' only
  //      vector-memory operations are their SHARE DMA instructions per stage, retired in order.
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  issue(0, 0);
  for (int t = 0; t < t_end; t += 2) {
    if (t + 1 < t_end) { issue(t + 1, 1); wait_vmcnt<SHARE>(); } else { wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    compute(t, S0{});
    __builtin_amdgcn_s_barrier();
    if (t + 1 < t_end) {
      if (t + 2 < t_end) { issue(t + 2, 0); wait_vmcnt<SHARE>(); } else { wait_vmcnt<0>(); }
      __builtin_amdgcn_s_barrier();
      compute(t + 1, S1{});
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue: O^T -> out[token][head][dim] -----------------------------------------------------
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (!col_ok) return;
  const float inv = 1.f / l;
  uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (int64_t)(q0 + tile0 + col) * p.out_stride +
                (int64_t)head * HD + 4 * g;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    uint2 r;
    r.x = E::pack(acc[n][0] * inv, acc[n][1] * inv);
    r.y = E::pack(acc[n][2] * inv, acc[n][3] * inv);
    *reinterpret_cast<uint2*>(o + 16 * n) = r;
  }
}

template <typename Tag, int HD>
int launch_prefill(const PrefillParams& p, int num_kv_heads, int max_query_len, hipStream_t s) {
  const int waves = p.heads_per_wg;  // one wave per query head
  const dim3 grid((max_query_len + kQTile - 1) / kQTile, p.num_seqs, num_kv_heads * (p.kv_group / waves)),
      block(64 * waves);
  if (waves >= 4) hipLaunchKernelGGL((prefill_mfma_kernel<Tag, HD, 4>), grid, block, 0, s, p);
  else if (waves >= 2) hipLaunchKernelGGL((prefill_mfma_kernel<Tag, HD, 2>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((prefill_mfma_kernel<Tag, HD, 1>), grid, block, 0, s, p);
  return launch_status();
}

}  // namespace
}  // namespace mi

extern "C" int mi_paged_attn_prefill(void* out, const void* q, const void* k_cache, const void* v_cache,
                                     const int32_t* block_table, const int32_t* seq_lens,
                                     const int32_t* query_start_loc, int32_t num_seqs, int32_t num_tokens,
                                     int32_t max_query_len, int32_t num_q_heads, int32_t num_kv_heads,
                                     int32_t head_dim, int32_t block_size, int32_t block_table_stride,
                                     float scale, int64_t q_token_stride, int64_t out_token_stride,
                                     int64_t cache_block_stride, int64_t cache_token_stride,
                                     int64_t cache_head_stride, int dtype, void* stream) {
  using namespace mi;
  if (num_seqs == 0 || num_tokens == 0) return MI_OK;
  if (!out || !q || !k_cache || !v_cache || !block_table || !seq_lens || !query_start_loc)
    return MI_ERR_INVALID_ARG;
  if (num_seqs < 0 || num_tokens < 0 || max_query_len <= 0 || num_q_heads <= 0 || num_kv_heads <= 0 ||
      block_table_stride <= 0)
    return MI_ERR_INVALID_ARG;
  if (num_q_heads % num_kv_heads) return MI_ERR_INVALID_ARG;
  if (head_dim != 128 && head_dim != 64) return MI_ERR_UNSUPPORTED;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (!is_pow2(block_size) || block_size < 16) return MI_ERR_UNSUPPORTED;
  const int kv_group = num_q_heads / num_kv_heads;
  int heads_per_wg = 1;  // largest divisor of the GQA group that fits 8 waves (256-VGPR budget)
  for (int w = 8; w >= 1; --w)
    if (kv_group % w == 0) { heads_per_wg = w; break; }
  if (q_token_stride % 8 || out_token_stride % 8 || cache_block_stride % 8 || cache_token_stride % 8 ||
      cache_head_stride % 8)
    return MI_ERR_INVALID_ARG;
  if (((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out) & 15) return MI_ERR_INVALID_ARG;
  PrefillParams p;
  p.out = out;
  p.q = q;
  p.k_cache = k_cache;
  p.v_cache = v_cache;
  p.block_table = block_table;
  p.seq_lens = seq_lens;
  p.query_start_loc = query_start_loc;
  p.num_seqs = num_seqs;
  p.num_q_heads = num_q_heads;
  p.kv_group = kv_group;
  p.heads_per_wg = heads_per_wg;
  p.bs_log2 = __builtin_ctz((unsigned)block_size);
  p.block_table_stride = block_table_stride;
  p.scale_log2e = scale * kLog2e;
  p.q_stride = q_token_stride;
  p.out_stride = out_token_stride;
  p.blk_stride = cache_block_stride;
  p.tok_stride = cache_token_stride;
  p.head_stride = cache_head_stride;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MI_DTYPE_BF16)
    return head_dim == 128 ? launch_prefill<bf16_tag, 128>(p, num_kv_heads, max_query_len, s)
                           : launch_prefill<bf16_tag, 64>(p, num_kv_heads, max_query_len, s);
  return head_dim == 128 ? launch_prefill<f16_tag, 128>(p, num_kv_heads, max_query_len, s)
                         : launch_prefill<f16_tag, 64>(p, num_kv_heads, max_query_len, s);
}
