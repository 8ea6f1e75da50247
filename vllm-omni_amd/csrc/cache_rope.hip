// A4 / A7 / A7b — paged KV write, neox RoPE (1-D and M-RoPE), per-head q/k RMSNorm,
// fused into one launch per layer (these ops are bandwidth-trivial at decode —
// 2 MB per layer — so the launch count is what matters).
//
// Replaces `self.rotary_pos_emb(positions, q, k)` and the KV-write half of
// `self.attn(q, k, v)` at vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:153-154.
#include "common.h"

namespace mi {
namespace {

// ---- plain KV write ------------------------------------------------------
template <int VEC>  // elements per thread copy (8 -> 16 B, 1 -> 2 B fallback)
__global__ __launch_bounds__(256) void reshape_and_cache_kernel(
    const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, uint16_t* __restrict__ kc,
    uint16_t* __restrict__ vc, const int64_t* __restrict__ slot_mapping, int num_kv_heads,
    int head_dim, int block_size, int64_t k_tok, int64_t v_tok, int64_t c_blk, int64_t c_tok,
    int64_t c_head) {
  scalar_cache_acquire();
  const int t = blockIdx.x;
  const int64_t slot = slot_mapping[t];
  if (slot < 0) return;  // padding token
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  const int per_head = head_dim / VEC;
  const int n = num_kv_heads * per_head;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int h = i / per_head;
    const int d = (i % per_head) * VEC;
    const int64_t src = (int64_t)h * head_dim + d;
    const int64_t dst = blk * c_blk + off * c_tok + (int64_t)h * c_head + d;
    if constexpr (VEC == 8) {
      *reinterpret_cast<uint4*>(kc + dst) = *reinterpret_cast<const uint4*>(k + t * k_tok + src);
      *reinterpret_cast<uint4*>(vc + dst) = *reinterpret_cast<const uint4*>(v + t * v_tok + src);
    } else {
      kc[dst] = k[t * k_tok + src];
      vc[dst] = v[t * v_tok + src];
    }
  }
}

// ---- fused norm -> rope -> cache -----------------------------------------
struct RopeParams {
  uint16_t* q;
  uint16_t* k;
  const uint16_t* v;
  const int64_t* positions;
  int64_t pos_row_stride;
  int32_t sec0, sec01;  // M-RoPE section boundaries in half-dim index space
  const float* cos_sin;
  int32_t max_position, rot_dim;
  const uint16_t* q_norm_w;
  const uint16_t* k_norm_w;
  float eps;
  uint16_t* kc;
  uint16_t* vc;
  const int64_t* slot_mapping;
  int32_t num_q_heads, num_kv_heads, head_dim, block_size;
  int64_t q_tok, k_tok, v_tok, c_blk, c_tok, c_head;
  // PART: q/k/v come as the fp32 split-K partials [S][T][part_row] of qkv_proj (skinny_gemm.hip) instead of 16-bit
  // tensors; a value = rnd(sum over slices in order (+ bias)), the reduce kernel's arithmetic.  q is written to p.q
  // (a fresh [T, q_tok] tensor), k and v only to the cache.
  const float* part;
  const uint16_t* qkv_bias;
  int32_t S, part_T;
  int64_t part_row;
};

template <typename Tag>
__device__ __forceinline__ void unpack8(const uint4& r, float (&f)[8]) {
  using E = Elem<Tag>;
  f[0] = E::lo(r.x); f[1] = E::hi(r.x); f[2] = E::lo(r.y); f[3] = E::hi(r.y);
  f[4] = E::lo(r.z); f[5] = E::hi(r.z); f[6] = E::lo(r.w); f[7] = E::hi(r.w);
}
template <typename Tag>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  using E = Elem<Tag>;
  uint4 r;
  r.x = E::pack(f[0], f[1]); r.y = E::pack(f[2], f[3]);
  r.z = E::pack(f[4], f[5]); r.w = E::pack(f[6], f[7]);
  return r;
}

// One workgroup per token.  A rotate-item = one thread = dims [8j, 8j+8) and their
// rotate-half partners [rot/2 + 8j, rot/2 + 8j + 8) of one head.
template <typename Tag, bool MROPE, bool NORM, bool PART = false>
__global__ __launch_bounds__(256) void rope_cache_kernel(const RopeParams p) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  const int t = blockIdx.x;
  const int HD = p.head_dim;
  const int half = p.rot_dim / 2;
  const int rot_items = p.rot_dim / 16;          // per head
  const int pass_items = (HD - p.rot_dim) / 8;   // per head, un-rotated tail dims
  const int iph = rot_items + pass_items;
  const int n_heads = p.num_q_heads + p.num_kv_heads;

  int64_t slot = -1;
  if (p.kc) slot = p.slot_mapping[t];
  const int64_t cblk = slot >= 0 ? slot / p.block_size : 0;
  const int64_t coff = slot >= 0 ? slot % p.block_size : 0;

  int64_t pos0 = p.positions[t];
  int64_t pos1 = pos0, pos2 = pos0;
  if constexpr (MROPE) {
    pos1 = p.positions[p.pos_row_stride + t];
    pos2 = p.positions[2 * p.pos_row_stride + t];
  }
  auto clampp = [&](int64_t x) {
    return x < 0 ? (int64_t)0 : (x >= p.max_position ? (int64_t)p.max_position - 1 : x);
  };
  pos0 = clampp(pos0); pos1 = clampp(pos1); pos2 = clampp(pos2);

  // 8 consecutive values of this token's qkv row starting at column `col`, as 16-bit words
  auto from_partials = [&](int col) -> uint4 {
    const float* p0 = p.part + (int64_t)t * p.part_row + col;
    float4 vl[16], vh[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
      if (s < p.S) {
        const float* ps = p0 + (int64_t)s * p.part_T * p.part_row;
        vl[s] = *reinterpret_cast<const float4*>(ps);
        vh[s] = *reinterpret_cast<const float4*>(ps + 4);
      }
    float4 lo = vl[0], hi = vh[0];
#pragma unroll
    for (int s = 1; s < 16; ++s)
      if (s < p.S) {
        lo.x += vl[s].x; lo.y += vl[s].y; lo.z += vl[s].z; lo.w += vl[s].w;
        hi.x += vh[s].x; hi.y += vh[s].y; hi.z += vh[s].z; hi.w += vh[s].w;
      }
    if (p.qkv_bias) {
      const uint4 bv = *reinterpret_cast<const uint4*>(p.qkv_bias + col);
      lo.x += E::lo(bv.x); lo.y += E::hi(bv.x); lo.z += E::lo(bv.y); lo.w += E::hi(bv.y);
      hi.x += E::lo(bv.z); hi.y += E::hi(bv.z); hi.z += E::lo(bv.w); hi.w += E::hi(bv.w);
    }
    return make_uint4(E::pack(lo.x, lo.y), E::pack(lo.z, lo.w), E::pack(hi.x, hi.y), E::pack(hi.z, hi.w));
  };

  // gridDim.y > 1 (few tokens: a decode batch of <= 64 rows is <= 64 workgroups otherwise): the items of a token are
  // dealt to several workgroups in chunks of 256 — a multiple of every items-per-head, so a head's lanes stay together
  const int it0 = blockIdx.y * blockDim.x + threadIdx.x, it_step = blockDim.x * gridDim.y;
  for (int it = it0; it < n_heads * iph; it += it_step) {
    const int head = it / iph;
    const int j = it % iph;
    const bool is_q = head < p.num_q_heads;
    const int hh = is_q ? head : head - p.num_q_heads;
    uint16_t* base = is_q ? p.q + t * p.q_tok + (int64_t)hh * HD
                          : (PART ? nullptr : p.k + t * p.k_tok + (int64_t)hh * HD);   // PART: k lives only in the cache
    const int col0 = head * HD;   // column of this head in the qkv row: q heads, then k heads (then v heads)
    uint16_t* cdst = (!is_q && slot >= 0)
                         ? p.kc + cblk * p.c_blk + coff * p.c_tok + (int64_t)hh * p.c_head
                         : nullptr;
    if (j >= rot_items) {  // pass-through dims (partial rotary): only the cache copy
      if (PART && is_q) {    // ... and, from partials, q's own un-rotated tail
        const int d = p.rot_dim + (j - rot_items) * 8;
        *reinterpret_cast<uint4*>(base + d) = from_partials(col0 + d);
      }
      if (cdst) {
        const int d = p.rot_dim + (j - rot_items) * 8;
        uint4 tail;
        if constexpr (PART) tail = from_partials(col0 + d);
        else tail = *reinterpret_cast<const uint4*>(base + d);
        *reinterpret_cast<uint4*>(cdst + d) = tail;
      }
      continue;
    }
    float x[8], y[8];
    if constexpr (PART) {
      unpack8<Tag>(from_partials(col0 + 8 * j), x);
      unpack8<Tag>(from_partials(col0 + half + 8 * j), y);
    } else {
      unpack8<Tag>(*reinterpret_cast<const uint4*>(base + 8 * j), x);
      unpack8<Tag>(*reinterpret_cast<const uint4*>(base + half + 8 * j), y);
    }

    if constexpr (NORM) {
      // RMSNorm over head_dim (host guarantees rot_dim == head_dim, iph in {4,8}):
      // fp32 normalise -> round to dtype -> * weight in dtype (HF order), then rotate.
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += x[e] * x[e] + y[e] * y[e];
      ss += dpp_f32<kDppQuadXor1>(ss);
      ss += dpp_f32<kDppQuadXor2>(ss);
      if (iph == 8) ss += dpp_f32<kDppRowHalfMirror>(ss);
      const float inv = __builtin_amdgcn_rsqf(ss / (float)HD + p.eps);
      const uint16_t* w = is_q ? p.q_norm_w : p.k_norm_w;
      float wx[8], wy[8];
      unpack8<Tag>(*reinterpret_cast<const uint4*>(w + 8 * j), wx);
      unpack8<Tag>(*reinterpret_cast<const uint4*>(w + half + 8 * j), wy);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x[e] = E::from_bits(E::to_bits(E::from_bits(E::to_bits(x[e] * inv)) * wx[e]));
        y[e] = E::from_bits(E::to_bits(E::from_bits(E::to_bits(y[e] * inv)) * wy[e]));
      }
    }

    float cs[8], sn[8];
    if constexpr (MROPE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int jj = 8 * j + e;
        const int64_t pp = jj < p.sec0 ? pos0 : (jj < p.sec01 ? pos1 : pos2);
        const float* row = p.cos_sin + pp * p.rot_dim;
        cs[e] = row[jj];
        sn[e] = row[half + jj];
      }
    } else {
      const float* row = p.cos_sin + pos0 * p.rot_dim;
      const float4 c0 = *reinterpret_cast<const float4*>(row + 8 * j);
      const float4 c1 = *reinterpret_cast<const float4*>(row + 8 * j + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(row + half + 8 * j);
      const float4 s1 = *reinterpret_cast<const float4*>(row + half + 8 * j + 4);
      cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w;
      cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
      sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w;
      sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
    }
    float ox[8], oy[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ox[e] = x[e] * cs[e] - y[e] * sn[e];  // neox: (x, y) = (d, d + rot/2)
      oy[e] = y[e] * cs[e] + x[e] * sn[e];
    }
    const uint4 rx = pack8<Tag>(ox), ry = pack8<Tag>(oy);
    if (base) {
      *reinterpret_cast<uint4*>(base + 8 * j) = rx;
      *reinterpret_cast<uint4*>(base + half + 8 * j) = ry;
    }
    if (cdst) {
      *reinterpret_cast<uint4*>(cdst + 8 * j) = rx;
      *reinterpret_cast<uint4*>(cdst + half + 8 * j) = ry;
    }
  }

  // V -> cache (pure copy)
  if (p.vc && slot >= 0) {
    const int per_head = HD / 8;
    for (int i = it0; i < p.num_kv_heads * per_head; i += it_step) {
      const int h = i / per_head, d = (i % per_head) * 8;
      uint4 vv;
      if constexpr (PART) vv = from_partials((p.num_q_heads + p.num_kv_heads + h) * HD + d);
      else vv = *reinterpret_cast<const uint4*>(p.v + t * p.v_tok + (int64_t)h * HD + d);
      *reinterpret_cast<uint4*>(p.vc + cblk * p.c_blk + coff * p.c_tok + (int64_t)h * p.c_head + d) = vv;
    }
  }
}

template <typename Tag>
int launch_rope(const RopeParams& p, bool mrope, bool norm, int num_tokens, hipStream_t s) {
  const int items = (p.num_q_heads + p.num_kv_heads) * (p.rot_dim / 16 + (p.head_dim - p.rot_dim) / 8);
  const int gy = num_tokens <= 64 ? std::min(4, (items + 255) / 256) : 1;
  const dim3 grid(num_tokens, gy), block(256);
  if (p.part) {
    if (mrope && norm) hipLaunchKernelGGL((rope_cache_kernel<Tag, true, true, true>), grid, block, 0, s, p);
    else if (mrope) hipLaunchKernelGGL((rope_cache_kernel<Tag, true, false, true>), grid, block, 0, s, p);
    else if (norm) hipLaunchKernelGGL((rope_cache_kernel<Tag, false, true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((rope_cache_kernel<Tag, false, false, true>), grid, block, 0, s, p);
    return launch_status();
  }
  if (mrope && norm) hipLaunchKernelGGL((rope_cache_kernel<Tag, true, true>), grid, block, 0, s, p);
  else if (mrope) hipLaunchKernelGGL((rope_cache_kernel<Tag, true, false>), grid, block, 0, s, p);
  else if (norm) hipLaunchKernelGGL((rope_cache_kernel<Tag, false, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((rope_cache_kernel<Tag, false, false>), grid, block, 0, s, p);
  return launch_status();
}

}  // namespace
}  // namespace mi

extern "C" int mi_reshape_and_cache(const void* k, const void* v, void* k_cache, void* v_cache,
                                    const int64_t* slot_mapping, int32_t num_tokens,
                                    int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                    int64_t k_token_stride, int64_t v_token_stride,
                                    int64_t cache_block_stride, int64_t cache_token_stride,
                                    int64_t cache_head_stride, int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!k || !v || !k_cache || !v_cache || !slot_mapping) return MI_ERR_INVALID_ARG;
  if (num_tokens < 0 || num_kv_heads <= 0 || head_dim <= 0 || block_size <= 0)
    return MI_ERR_INVALID_ARG;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool vec = head_dim % 8 == 0 && k_token_stride % 8 == 0 && v_token_stride % 8 == 0 &&
                   cache_block_stride % 8 == 0 && cache_token_stride % 8 == 0 &&
                   cache_head_stride % 8 == 0 &&
                   !(((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15);
  auto kk = reinterpret_cast<const uint16_t*>(k);
  auto vv = reinterpret_cast<const uint16_t*>(v);
  auto kc = reinterpret_cast<uint16_t*>(k_cache);
  auto vc = reinterpret_cast<uint16_t*>(v_cache);
  if (vec)
    hipLaunchKernelGGL((reshape_and_cache_kernel<8>), dim3(num_tokens), dim3(256), 0, s, kk, vv, kc,
                       vc, slot_mapping, num_kv_heads, head_dim, block_size, k_token_stride,
                       v_token_stride, cache_block_stride, cache_token_stride, cache_head_stride);
  else
    hipLaunchKernelGGL((reshape_and_cache_kernel<1>), dim3(num_tokens), dim3(256), 0, s, kk, vv, kc,
                       vc, slot_mapping, num_kv_heads, head_dim, block_size, k_token_stride,
                       v_token_stride, cache_block_stride, cache_token_stride, cache_head_stride);
  return launch_status();
}

extern "C" int mi_rope_neox_and_cache(
    void* q, void* k, const void* v, const int64_t* positions, int64_t positions_row_stride,
    const int32_t* mrope_section, const float* cos_sin_cache, int32_t max_position,
    int32_t rot_dim, const void* q_norm_weight, const void* k_norm_weight, float norm_eps,
    void* k_cache, void* v_cache, const int64_t* slot_mapping, int32_t num_tokens,
    int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
    int64_t q_token_stride, int64_t k_token_stride, int64_t v_token_stride,
    int64_t cache_block_stride, int64_t cache_token_stride, int64_t cache_head_stride, int dtype,
    void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!q || !k || !positions || !cos_sin_cache) return MI_ERR_INVALID_ARG;
  if (num_tokens < 0 || num_q_heads <= 0 || num_kv_heads <= 0 || max_position <= 0)
    return MI_ERR_INVALID_ARG;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (head_dim % 8 || rot_dim % 16 || rot_dim <= 0 || rot_dim > head_dim) return MI_ERR_UNSUPPORTED;
  if (q_token_stride % 8 || k_token_stride % 8) return MI_ERR_INVALID_ARG;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos_sin_cache) & 15) return MI_ERR_INVALID_ARG;
  const bool norm = q_norm_weight != nullptr || k_norm_weight != nullptr;
  if (norm) {
    if (!q_norm_weight || !k_norm_weight) return MI_ERR_INVALID_ARG;
    if (rot_dim != head_dim || (head_dim != 64 && head_dim != 128)) return MI_ERR_UNSUPPORTED;
  }
  if (k_cache) {
    if (!v || !v_cache || !slot_mapping || block_size <= 0) return MI_ERR_INVALID_ARG;
    if (v_token_stride % 8 || cache_block_stride % 8 || cache_token_stride % 8 ||
        cache_head_stride % 8)
      return MI_ERR_INVALID_ARG;
    if (((uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15) return MI_ERR_INVALID_ARG;
  }
  RopeParams p;
  p.q = reinterpret_cast<uint16_t*>(q);
  p.k = reinterpret_cast<uint16_t*>(k);
  p.v = reinterpret_cast<const uint16_t*>(v);
  p.positions = positions;
  p.pos_row_stride = positions_row_stride;
  p.sec0 = 0;
  p.sec01 = 0;
  if (mrope_section) {
    if (mrope_section[0] + mrope_section[1] + mrope_section[2] != rot_dim / 2)
      return MI_ERR_INVALID_ARG;
    p.sec0 = mrope_section[0];
    p.sec01 = mrope_section[0] + mrope_section[1];
  }
  p.cos_sin = cos_sin_cache;
  p.max_position = max_position;
  p.rot_dim = rot_dim;
  p.q_norm_w = reinterpret_cast<const uint16_t*>(q_norm_weight);
  p.k_norm_w = reinterpret_cast<const uint16_t*>(k_norm_weight);
  p.eps = norm_eps;
  p.kc = reinterpret_cast<uint16_t*>(k_cache);
  p.vc = k_cache ? reinterpret_cast<uint16_t*>(v_cache) : nullptr;
  p.slot_mapping = slot_mapping;
  p.num_q_heads = num_q_heads;
  p.num_kv_heads = num_kv_heads;
  p.head_dim = head_dim;
  p.block_size = block_size;
  p.q_tok = q_token_stride;
  p.k_tok = k_token_stride;
  p.v_tok = v_token_stride;
  p.c_blk = cache_block_stride;
  p.c_tok = cache_token_stride;
  p.c_head = cache_head_stride;
  p.part = nullptr;
  p.qkv_bias = nullptr;
  p.S = 0;
  p.part_T = 0;
  p.part_row = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MI_DTYPE_BF16)
    return launch_rope<bf16_tag>(p, mrope_section != nullptr, norm, num_tokens, s);
  return launch_rope<f16_tag>(p, mrope_section != nullptr, norm, num_tokens, s);
}

extern "C" int mi_rope_neox_and_cache_splitk(
    void* q_out, const void* partials, int32_t split, const void* qkv_bias, const int64_t* positions,
    int64_t positions_row_stride, const int32_t* mrope_section, const float* cos_sin_cache, int32_t max_position,
    int32_t rot_dim, const void* q_norm_weight, const void* k_norm_weight, float norm_eps, void* k_cache,
    void* v_cache, const int64_t* slot_mapping, int32_t num_tokens, int32_t num_q_heads, int32_t num_kv_heads,
    int32_t head_dim, int32_t block_size, int64_t q_token_stride, int64_t cache_block_stride,
    int64_t cache_token_stride, int64_t cache_head_stride, int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!q_out || !partials || !positions || !cos_sin_cache || !k_cache || !v_cache || !slot_mapping)
    return MI_ERR_INVALID_ARG;
  if (num_tokens < 0 || num_q_heads <= 0 || num_kv_heads <= 0 || max_position <= 0 || block_size <= 0 || split < 1 ||
      split > 16)
    return MI_ERR_INVALID_ARG;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (head_dim % 8 || rot_dim % 16 || rot_dim <= 0 || rot_dim > head_dim) return MI_ERR_UNSUPPORTED;
  if (q_token_stride % 8 || cache_block_stride % 8 || cache_token_stride % 8 || cache_head_stride % 8)
    return MI_ERR_INVALID_ARG;
  if (((uintptr_t)q_out | (uintptr_t)partials | (uintptr_t)cos_sin_cache | (uintptr_t)k_cache | (uintptr_t)v_cache |
       (uintptr_t)qkv_bias) & 15)
    return MI_ERR_INVALID_ARG;
  const bool norm = q_norm_weight != nullptr || k_norm_weight != nullptr;
  if (norm) {
    if (!q_norm_weight || !k_norm_weight) return MI_ERR_INVALID_ARG;
    if (rot_dim != head_dim || (head_dim != 64 && head_dim != 128)) return MI_ERR_UNSUPPORTED;
  }
  RopeParams p;
  p.q = reinterpret_cast<uint16_t*>(q_out);
  p.k = nullptr;
  p.v = nullptr;
  p.positions = positions;
  p.pos_row_stride = positions_row_stride;
  p.sec0 = 0;
  p.sec01 = 0;
  if (mrope_section) {
    if (mrope_section[0] + mrope_section[1] + mrope_section[2] != rot_dim / 2) return MI_ERR_INVALID_ARG;
    p.sec0 = mrope_section[0];
    p.sec01 = mrope_section[0] + mrope_section[1];
  }
  p.cos_sin = cos_sin_cache;
  p.max_position = max_position;
  p.rot_dim = rot_dim;
  p.q_norm_w = reinterpret_cast<const uint16_t*>(q_norm_weight);
  p.k_norm_w = reinterpret_cast<const uint16_t*>(k_norm_weight);
  p.eps = norm_eps;
  p.kc = reinterpret_cast<uint16_t*>(k_cache);
  p.vc = reinterpret_cast<uint16_t*>(v_cache);
  p.slot_mapping = slot_mapping;
  p.num_q_heads = num_q_heads;
  p.num_kv_heads = num_kv_heads;
  p.head_dim = head_dim;
  p.block_size = block_size;
  p.q_tok = q_token_stride;
  p.k_tok = 0;
  p.v_tok = 0;
  p.c_blk = cache_block_stride;
  p.c_tok = cache_token_stride;
  p.c_head = cache_head_stride;
  p.part = reinterpret_cast<const float*>(partials);
  p.qkv_bias = reinterpret_cast<const uint16_t*>(qkv_bias);
  p.S = split;
  p.part_T = num_tokens;
  p.part_row = (int64_t)(num_q_heads + 2 * num_kv_heads) * head_dim;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MI_DTYPE_BF16)
    return launch_rope<bf16_tag>(p, mrope_section != nullptr, norm, num_tokens, s);
  return launch_rope<f16_tag>(p, mrope_section != nullptr, norm, num_tokens, s);
}
