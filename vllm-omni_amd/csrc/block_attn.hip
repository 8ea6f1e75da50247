// One decode step of attention for sequences whose WHOLE context lives in one KV block (<= 64 tokens): (q/k-norm ->)
// neox RoPE -> append K/V to the block -> attention over the block, in ONE launch.
//
// Where it is used: the Qwen3-Omni talker's code predictor (SURVEY §8f N4;
// vllm_omni/model_executor/models/common/qwen3_code_predictor.py:318-626): a 5-layer transformer over the G + 1 <= 33
// positions of one talker token, 31 single-token steps per call (vllm_omni_amd/code_predictor.py keeps a private paged
// cache with one 64-token block per sequence).  Per layer and step that was a RoPE + KV-write launch (~6 us) and a
// launch of the general paged-attention decode kernel (~19 us at batch 512: ring, partitions and block-table walk
// built for contexts of thousands of tokens) for ~17 KB of K/V per (sequence, KV head).  Same math as that pair
// (csrc/cache_rope.hip rope_cache_kernel + csrc/paged_attn_decode_mfma.hip): q and k are rounded to the cache dtype
// after norm and rotation, scores are exact products of those 16-bit values summed in fp32, softmax and P.V in fp32 (P is
// not rounded at all here), one rounding of the output.
//
// One wave per (sequence row, KV head), four waves per workgroup:
//   1. lane l holds the rotation pair (d, d + 64) = (l, l + 64) of every head: norm (wave sum of squares), RoPE, the
//      new k / v go to the cache slot `position`, the rotated q heads to LDS as 16-bit rows;
//   2. scores of the tokens already in the block: S^T = K . Q^T on MFMA 16x16x32, K fragments straight from the cache
//      (lane = token row, 16 B of a 64-B run), Q^T fragments from LDS (columns >= G are zero); the new token's score
//      comes from the registers of step 1 (no read-after-write through memory);
//   3. softmax per q head over (<= 4 tiles x 4 rows per lane) + the new token, cross-lane over the 4 lane groups;
//   4. O = P . V on the VALU: lane l owns dims (2l, 2l + 1), V rows are 4-B coalesced loads, p(token) is a
//      `v_readlane` of the score register that holds it (wave-uniform), fully unrolled.
#include "mfma_attn_common.h"

namespace mi {
namespace {

struct BlockAttnParams {
  uint16_t* out;
  const uint16_t *q, *k, *v;
  int64_t o_tok, q_tok, k_tok, v_tok;
  const int64_t* positions;
  const float* cos_sin;
  int32_t max_position;
  const uint16_t *q_norm_w, *k_norm_w;
  float eps;
  uint16_t *kc, *vc;
  int64_t c_blk, c_tok, c_head;
  const int32_t* block_ids;
  int64_t bt_stride;
  int32_t T, nkv, block_size;
  int32_t uniform_pos;     // >= 0: every row's new token sits at this position (`positions` is not read)
  float scale_log2e;
  // PART: q / k / v are the fp32 split-K partials [S][T][(nq + 2 nkv) * 128] of qkv_proj (mi_linear_partials); a value
  // is rnd16(sum over slices in order (+ bias)) — what mi_linear_reduce would have stored
  const float* part;
  const uint16_t* qkv_bias;
  int32_t S, nq;
};

// sum over the 64 lanes, in every lane: DPP inside each row of 16, row exchanges across (common.h) — no LDS-permute
// butterfly (six dependent ~100-cycle round trips each; this kernel is one latency chain)
__device__ __forceinline__ float wave_sum(float v) { return rows_allreduce_sum(group_allreduce_sum<16>(v)); }

constexpr int kHD = 128;

// TILES: 16-token tiles that cover the tokens already in any block of the launch (host bound on `positions`): the
// K fragments and V rows of a lane are registers, so a launch early in a sequence runs at twice the occupancy.
template <typename Tag, int G, bool NORM, int TILES, bool PART = false>
__global__ __launch_bounds__(256) void block_attn_kernel(const BlockAttnParams p) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  __shared__ __attribute__((aligned(16))) uint16_t lq[4][G * kHD];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int pair = blockIdx.x * 4 + w;
  const bool live = pair < p.T * p.nkv;       // a tail wave repeats the last pair and stores nothing
  if (!live) pair = p.T * p.nkv - 1;
  const int t = pair / p.nkv, kvh = pair - t * p.nkv;
  int64_t pos64 = p.uniform_pos >= 0 ? (int64_t)p.uniform_pos : p.positions[t];
  const int cap = p.block_size - 1 < 16 * TILES ? p.block_size - 1 : 16 * TILES;   // host checks; never index outside the block / the tiles
  pos64 = pos64 < 0 ? 0 : (pos64 > cap ? (int64_t)cap : pos64);
  const int pos = __builtin_amdgcn_readfirstlane((int)pos64);
  const int rpos = pos < p.max_position ? pos : p.max_position - 1;
  const int64_t blk = p.block_ids[(int64_t)t * p.bt_stride];
  uint16_t* kblk = p.kc + blk * p.c_blk + (int64_t)kvh * p.c_head;
  uint16_t* vblk = p.vc + blk * p.c_blk + (int64_t)kvh * p.c_head;

  // ---- K fragments / V rows of the tokens already in the block: independent of everything below, issued first
  const int r16 = lane & 15, kg = lane >> 4;
  const int tiles = (pos + 15) >> 4;          // <= TILES
  uint4 ka[TILES][4];
#pragma unroll
  for (int tile = 0; tile < TILES; ++tile) {
    int tok = 16 * tile + r16;
    tok = tok < p.block_size ? tok : p.block_size - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ka[tile][ks] = make_uint4(0, 0, 0, 0);
      // rows past the last token stay zero and unread (their scores are masked): a partly filled tile costs its tokens
      if (tile < tiles && 16 * tile + r16 < pos)
        ka[tile][ks] = *reinterpret_cast<const uint4*>(kblk + (int64_t)tok * p.c_tok + 32 * ks + 8 * kg);
    }
  }
  uint32_t vw[16 * TILES];
#pragma unroll
  for (int j = 0; j < 16 * TILES; ++j) {
    vw[j] = 0;
    if (j < pos) vw[j] = *reinterpret_cast<const uint32_t*>(vblk + (int64_t)j * p.c_tok + 2 * lane);
  }

  // ---- 1. norm + RoPE of the G q heads and the k head; append k, v
  const float cs = p.cos_sin[(int64_t)rpos * kHD + lane], sn = p.cos_sin[(int64_t)rpos * kHD + 64 + lane];
  auto norm_rope = [&](float x, float y, const uint16_t* wgt, float& ox, float& oy) {
    if constexpr (NORM) {
      // fp32 normalise -> round -> * weight in the 16-bit type -> round (HF order; cache_rope.hip)
      const float ss = wave_sum(x * x + y * y);
      const float inv = __builtin_amdgcn_rsqf(ss / (float)kHD + p.eps);
      x = E::from_bits(E::to_bits(E::from_bits(E::to_bits(x * inv)) * E::from_bits(wgt[lane])));
      y = E::from_bits(E::to_bits(E::from_bits(E::to_bits(y * inv)) * E::from_bits(wgt[64 + lane])));
    }
    ox = E::from_bits(E::to_bits(x * cs - y * sn));   // neox: (x, y) = (d, d + 64)
    oy = E::from_bits(E::to_bits(y * cs + x * sn));
  };
  // column `col` of this token's qkv row (q heads, then k heads, then v heads), as the 16-bit value the GEMM stores
  const int64_t row_len = (int64_t)(p.nq + 2 * p.nkv) * kHD;
  auto from_part = [&](int col) -> float {
    const float* p0 = p.part + (int64_t)t * row_len + col;
    float v[16];
#pragma unroll
    for (int sl = 0; sl < 16; ++sl)
      if (sl < p.S) v[sl] = p0[(int64_t)sl * p.T * row_len];
    float a = v[0];
#pragma unroll
    for (int sl = 1; sl < 16; ++sl)
      if (sl < p.S) a += v[sl];
    if (p.qkv_bias) a += E::from_bits(p.qkv_bias[col]);
    return E::from_bits(E::to_bits(a));
  };
  float kx, ky;
  if constexpr (PART) {
    const int col = (p.nq + kvh) * kHD;
    norm_rope(from_part(col + lane), from_part(col + 64 + lane), p.k_norm_w, kx, ky);
  } else {
    const uint16_t* kr = p.k + (int64_t)t * p.k_tok + kvh * kHD;
    norm_rope(E::from_bits(kr[lane]), E::from_bits(kr[64 + lane]), p.k_norm_w, kx, ky);
  }
  uint32_t vnew;
  if constexpr (PART) {
    const int col = (p.nq + p.nkv + kvh) * kHD + 2 * lane;
    vnew = E::pack(from_part(col), from_part(col + 1));
  } else {
    vnew = *reinterpret_cast<const uint32_t*>(p.v + (int64_t)t * p.v_tok + kvh * kHD + 2 * lane);
  }
  if (live) {
    uint16_t* kd = kblk + (int64_t)pos * p.c_tok;
    kd[lane] = E::to_bits(kx);
    kd[64 + lane] = E::to_bits(ky);
    *reinterpret_cast<uint32_t*>(vblk + (int64_t)pos * p.c_tok + 2 * lane) = vnew;
  }
  float s_new[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float qx, qy;
    if constexpr (PART) {
      const int col = (kvh * G + g) * kHD;
      norm_rope(from_part(col + lane), from_part(col + 64 + lane), p.q_norm_w, qx, qy);
    } else {
      const uint16_t* qr = p.q + (int64_t)t * p.q_tok + (kvh * G + g) * kHD;
      norm_rope(E::from_bits(qr[lane]), E::from_bits(qr[64 + lane]), p.q_norm_w, qx, qy);
    }
    lq[w][g * kHD + lane] = E::to_bits(qx);
    lq[w][g * kHD + 64 + lane] = E::to_bits(qy);
    s_new[g] = wave_sum(qx * kx + qy * ky);
  }
  __syncthreads();

  // ---- 2. scores of the tokens already in the block: D[token][head] = K . Q^T
  f32x4_t acc[TILES];
#pragma unroll
  for (int tile = 0; tile < TILES; ++tile) acc[tile] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4 qb = make_uint4(0, 0, 0, 0);
    if (r16 < G) qb = *reinterpret_cast<const uint4*>(&lq[w][r16 * kHD + 32 * ks + 8 * kg]);
#pragma unroll
    for (int tile = 0; tile < TILES; ++tile)
      if (tile < tiles) acc[tile] = mfma16<Tag>(ka[tile][ks], qb, acc[tile]);
  }

  // ---- 3. softmax of column r16 (a q head when r16 < G): this lane holds tokens 16 tile + 4 kg + r
  float mine = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) mine = r16 == g ? s_new[g] : mine;
  float mx = mine;
#pragma unroll
  for (int tile = 0; tile < TILES; ++tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = 16 * tile + 4 * kg + r < pos;
      mx = ok ? fmaxf(mx, acc[tile][r]) : mx;
    }
  mx = rows_allreduce_max(mx);
  float e[TILES][4];
  float sum = 0.f;
#pragma unroll
  for (int tile = 0; tile < TILES; ++tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = 16 * tile + 4 * kg + r < pos;
      e[tile][r] = ok ? __builtin_amdgcn_exp2f((acc[tile][r] - mx) * p.scale_log2e) : 0.f;
      sum += e[tile][r];
    }
  sum = rows_allreduce_sum(sum);
  const float e_new = __builtin_amdgcn_exp2f((mine - mx) * p.scale_log2e);
  const float inv = 1.f / (sum + e_new);

  // ---- 4. O = P . V; lane = dims (2 lane, 2 lane + 1); p(token j, head g) sits in lane 16 * ((j % 16) / 4) + g
  float o0[G], o1[G];
  const float vn0 = E::lo(vnew), vn1 = E::hi(vnew);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float pn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e_new), g));
    o0[g] = pn * vn0;
    o1[g] = pn * vn1;
  }
#pragma unroll
  for (int tile = 0; tile < TILES; ++tile)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * tile + 4 * q4 + r;
        if (j < pos) {   // wave-uniform
          const float v0 = E::lo(vw[j]), v1 = E::hi(vw[j]);
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e[tile][r]), 16 * q4 + g));
            o0[g] += pj * v0;
            o1[g] += pj * v1;
          }
        }
      }
  if (live) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float ig = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), g));
      *reinterpret_cast<uint32_t*>(p.out + (int64_t)t * p.o_tok + (kvh * G + g) * kHD + 2 * lane) = E::pack(o0[g] * ig, o1[g] * ig);
    }
  }
}

template <typename Tag, int G>
int launch_block_attn(const BlockAttnParams& p, bool norm, int tiles, hipStream_t s) {
  const dim3 grid((unsigned)(((int64_t)p.T * p.nkv + 3) / 4)), block(256);
#define MI_BA_T(N)                                                                                        \
  do {                                                                                                    \
    if (p.part) {                                                                                         \
      if (norm) hipLaunchKernelGGL((block_attn_kernel<Tag, G, true, N, true>), grid, block, 0, s, p);    \
      else hipLaunchKernelGGL((block_attn_kernel<Tag, G, false, N, true>), grid, block, 0, s, p);        \
    } else {                                                                                              \
      if (norm) hipLaunchKernelGGL((block_attn_kernel<Tag, G, true, N>), grid, block, 0, s, p);          \
      else hipLaunchKernelGGL((block_attn_kernel<Tag, G, false, N>), grid, block, 0, s, p);              \
    }                                                                                                     \
  } while (0)
  if (tiles <= 1) MI_BA_T(1);
  else if (tiles == 2) MI_BA_T(2);
  else if (tiles == 3) MI_BA_T(3);
  else MI_BA_T(4);
#undef MI_BA_T
  return launch_status();
}

}  // namespace
}  // namespace mi

namespace {
int block_attend(void* out, const void* q, const void* k, const void* v, const float* partials, int32_t split,
                 const void* qkv_bias, const int64_t* positions, int32_t max_position_bound, const float* cos_sin_cache,
                 int32_t max_position, int32_t rot_dim, const void* q_norm_weight, const void* k_norm_weight, float norm_eps,
                 void* k_cache, void* v_cache, const int32_t* block_ids, int64_t block_ids_stride, int32_t num_tokens,
                 int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size, int64_t out_token_stride,
                 int64_t q_token_stride, int64_t k_token_stride, int64_t v_token_stride, int64_t cache_block_stride,
                 int64_t cache_token_stride, int64_t cache_head_stride, float scale, int dtype, void* stream);
}  // namespace

extern "C" int mi_rope_cache_attend_block_splitk(
    void* out, const void* partials, int32_t split, const void* qkv_bias, const int64_t* positions,
    int32_t max_position_bound, const float* cos_sin_cache, int32_t max_position, int32_t rot_dim,
    const void* q_norm_weight, const void* k_norm_weight, float norm_eps, void* k_cache, void* v_cache,
    const int32_t* block_ids, int64_t block_ids_stride, int32_t num_tokens, int32_t num_q_heads, int32_t num_kv_heads,
    int32_t head_dim, int32_t block_size, int64_t out_token_stride, int64_t cache_block_stride,
    int64_t cache_token_stride, int64_t cache_head_stride, float scale, int dtype, void* stream) {
  if (num_tokens == 0) return MI_OK;
  if (!partials || split < 1 || split > 16 || ((uintptr_t)partials & 3)) return MI_ERR_INVALID_ARG;
  return block_attend(out, nullptr, nullptr, nullptr, reinterpret_cast<const float*>(partials), split, qkv_bias, positions,
                      max_position_bound, cos_sin_cache, max_position, rot_dim, q_norm_weight, k_norm_weight, norm_eps,
                      k_cache, v_cache, block_ids, block_ids_stride, num_tokens, num_q_heads, num_kv_heads, head_dim,
                      block_size, out_token_stride, 0, 0, 0, cache_block_stride, cache_token_stride, cache_head_stride,
                      scale, dtype, stream);
}

extern "C" int mi_rope_cache_attend_block(
    void* out, const void* q, const void* k, const void* v, const int64_t* positions, int32_t max_position_bound,
    const float* cos_sin_cache, int32_t max_position, int32_t rot_dim, const void* q_norm_weight,
    const void* k_norm_weight, float norm_eps, void* k_cache, void* v_cache, const int32_t* block_ids,
    int64_t block_ids_stride, int32_t num_tokens, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
    int32_t block_size, int64_t out_token_stride, int64_t q_token_stride, int64_t k_token_stride,
    int64_t v_token_stride, int64_t cache_block_stride, int64_t cache_token_stride, int64_t cache_head_stride,
    float scale, int dtype, void* stream) {
  if (num_tokens == 0) return MI_OK;
  if (!q || !k || !v || ((uintptr_t)v & 3) || v_token_stride % 2) return MI_ERR_INVALID_ARG;
  return block_attend(out, q, k, v, nullptr, 0, nullptr, positions, max_position_bound, cos_sin_cache, max_position, rot_dim,
                      q_norm_weight, k_norm_weight, norm_eps, k_cache, v_cache, block_ids, block_ids_stride, num_tokens,
                      num_q_heads, num_kv_heads, head_dim, block_size, out_token_stride, q_token_stride, k_token_stride,
                      v_token_stride, cache_block_stride, cache_token_stride, cache_head_stride, scale, dtype, stream);
}

namespace {
int block_attend(void* out, const void* q, const void* k, const void* v, const float* partials, int32_t split,
                 const void* qkv_bias, const int64_t* positions, int32_t max_position_bound, const float* cos_sin_cache,
                 int32_t max_position, int32_t rot_dim, const void* q_norm_weight, const void* k_norm_weight, float norm_eps,
                 void* k_cache, void* v_cache, const int32_t* block_ids, int64_t block_ids_stride, int32_t num_tokens,
                 int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size, int64_t out_token_stride,
                 int64_t q_token_stride, int64_t k_token_stride, int64_t v_token_stride, int64_t cache_block_stride,
                 int64_t cache_token_stride, int64_t cache_head_stride, float scale, int dtype, void* stream) {

  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!out || !cos_sin_cache || !k_cache || !v_cache || !block_ids) return MI_ERR_INVALID_ARG;
  if (max_position_bound < 0 || max_position_bound >= block_size) return MI_ERR_INVALID_ARG;
  if (num_tokens < 0 || num_q_heads <= 0 || num_kv_heads <= 0 || max_position <= 0 || block_size <= 0) return MI_ERR_INVALID_ARG;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (head_dim != kHD || rot_dim != kHD || block_size > 64 || num_q_heads % num_kv_heads) return MI_ERR_UNSUPPORTED;
  if ((q_norm_weight != nullptr) != (k_norm_weight != nullptr)) return MI_ERR_INVALID_ARG;
  if (out_token_stride % 2 || cache_block_stride % 8 || cache_token_stride % 8 || cache_head_stride % 8) return MI_ERR_INVALID_ARG;
  if (((uintptr_t)out & 3) || ((uintptr_t)k_cache | (uintptr_t)v_cache) & 15 || ((uintptr_t)cos_sin_cache & 3))
    return MI_ERR_INVALID_ARG;
  BlockAttnParams p;
  p.out = reinterpret_cast<uint16_t*>(out);
  p.q = reinterpret_cast<const uint16_t*>(q);
  p.k = reinterpret_cast<const uint16_t*>(k);
  p.v = reinterpret_cast<const uint16_t*>(v);
  p.o_tok = out_token_stride; p.q_tok = q_token_stride; p.k_tok = k_token_stride; p.v_tok = v_token_stride;
  p.positions = positions;
  p.cos_sin = cos_sin_cache;
  p.max_position = max_position;
  p.q_norm_w = reinterpret_cast<const uint16_t*>(q_norm_weight);
  p.k_norm_w = reinterpret_cast<const uint16_t*>(k_norm_weight);
  p.eps = norm_eps;
  p.kc = reinterpret_cast<uint16_t*>(k_cache);
  p.vc = reinterpret_cast<uint16_t*>(v_cache);
  p.c_blk = cache_block_stride; p.c_tok = cache_token_stride; p.c_head = cache_head_stride;
  p.block_ids = block_ids;
  p.bt_stride = block_ids_stride;
  p.T = num_tokens; p.nkv = num_kv_heads; p.block_size = block_size;
  p.uniform_pos = positions ? -1 : max_position_bound;
  p.part = partials;
  p.qkv_bias = reinterpret_cast<const uint16_t*>(qkv_bias);
  p.S = split;
  p.nq = num_q_heads;
  const int tiles = (max_position_bound + 15) / 16;
  p.scale_log2e = scale * 1.4426950408889634f;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool norm = q_norm_weight != nullptr;
  const int g = num_q_heads / num_kv_heads;
#define MI_BA(TAG)                                              \
  do {                                                          \
    if (g == 1) return launch_block_attn<TAG, 1>(p, norm, tiles, s);   \
    if (g == 2) return launch_block_attn<TAG, 2>(p, norm, tiles, s);   \
    if (g == 4) return launch_block_attn<TAG, 4>(p, norm, tiles, s);   \
    if (g == 8) return launch_block_attn<TAG, 8>(p, norm, tiles, s);   \
    return MI_ERR_UNSUPPORTED;                                  \
  } while (0)
  if (dtype == MI_DTYPE_BF16) MI_BA(bf16_tag);
  MI_BA(f16_tag);
#undef MI_BA
}
}  // namespace
