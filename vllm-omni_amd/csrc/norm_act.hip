// A8 RMSNorm / fused-add RMSNorm, A9 SiLU-gate, A11 talker logit mask + greedy argmax.
// All three are launch-bound at decode (<= 20 MB per call); they are written as plain
// vectorised (16 B/lane) streaming kernels, one workgroup per row.
//
// Replaces upstream RMSNorm / SiluAndMul reached from
// vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:63,67,212-216,223,325 and
// `bad_word_processor` (qwen2_5_omni_talker.py:150-169).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace mi {
namespace {

__device__ __forceinline__ float block_allreduce_sum(float v, float* smem) {
  v = wave_allreduce_sum(v);
  const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) smem[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem[i];
  __syncthreads();
  return r;
}

__device__ __forceinline__ float block_allreduce_max(float v, float* smem) {
  v = wave_allreduce_max(v);
  const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) smem[wid] = v;
  __syncthreads();
  float r = smem[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, smem[i]);
  __syncthreads();
  return r;
}

// one value of the W8A8 activation quantisation (int8_config.py:323-342 scheme; same arithmetic as
// dynamic_int8_quant_kernel: exact division, round-half-to-even, clamp)
__device__ __forceinline__ int quant1(float v, float scale) {
  return (int)fminf(fmaxf(rintf(v / scale), -128.f), 127.f);
}
__device__ __forceinline__ uint32_t quant4(float a, float b, float c, float d, float scale) {
  return (uint32_t)(quant1(a, scale) & 255) | ((uint32_t)(quant1(b, scale) & 255) << 8) |
         ((uint32_t)(quant1(c, scale) & 255) << 16) | ((uint32_t)(quant1(d, scale) & 255) << 24);
}

// rounding order: z = rnd(x + r); y = rnd(rnd(z * inv_rms) * w)
// QUANT: the normalised row leaves as int8 + one fp32 scale (the quantisation that W8A8 linears put in front of
// their GEMM, fused here: the row is in registers anyway) — bit-identical to this kernel followed by
// dynamic_int8_quant_kernel; `out` is then the int8 row [T, hidden] and `q_scale` its scale [T].
template <typename Tag, bool ADD, int MAXC, bool QUANT = false>
__global__ __launch_bounds__(256) void rms_norm_kernel(uint16_t* __restrict__ out,
                                                       const uint16_t* __restrict__ x,
                                                       uint16_t* __restrict__ residual,
                                                       const uint16_t* __restrict__ w, float eps,
                                                       int hidden, int64_t x_stride,
                                                       int64_t out_stride, int64_t r_stride,
                                                       float* __restrict__ q_scale = nullptr) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  __shared__ float smem[4];
  const int row = blockIdx.x;
  const int nchunk = hidden / 8;
  const uint16_t* xr = x + row * x_stride;
  uint16_t* rr = ADD ? residual + row * r_stride : nullptr;
  uint16_t* orow = out + row * out_stride;
  uint4 z[MAXC], wreg[MAXC];   // the norm weights are loaded beside the row, not behind the reduction (one L2 round trip less)
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int ch = threadIdx.x + k * 256;
    if (ch < nchunk) {
      wreg[k] = *reinterpret_cast<const uint4*>(w + ch * 8);
      uint4 a = *reinterpret_cast<const uint4*>(xr + ch * 8);
      if constexpr (ADD) {
        const uint4 b = *reinterpret_cast<const uint4*>(rr + ch * 8);
        a.x = E::pack(E::lo(a.x) + E::lo(b.x), E::hi(a.x) + E::hi(b.x));
        a.y = E::pack(E::lo(a.y) + E::lo(b.y), E::hi(a.y) + E::hi(b.y));
        a.z = E::pack(E::lo(a.z) + E::lo(b.z), E::hi(a.z) + E::hi(b.z));
        a.w = E::pack(E::lo(a.w) + E::lo(b.w), E::hi(a.w) + E::hi(b.w));
        *reinterpret_cast<uint4*>(rr + ch * 8) = a;
      }
      z[k] = a;
      const uint32_t ws[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = E::lo(ws[i]), hi = E::hi(ws[i]);
        ss += lo * lo + hi * hi;
      }
    }
  }
  ss = block_allreduce_sum(ss, smem);
  const float inv = __builtin_amdgcn_rsqf(ss / (float)hidden + eps);
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int ch = threadIdx.x + k * 256;
    if (ch < nchunk) {
      const uint4 wv = wreg[k];
      const uint32_t zs[4] = {z[k].x, z[k].y, z[k].z, z[k].w};
      const uint32_t ws[4] = {wv.x, wv.y, wv.z, wv.w};
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float nlo = E::from_bits(E::to_bits(E::lo(zs[i]) * inv));
        const float nhi = E::from_bits(E::to_bits(E::hi(zs[i]) * inv));
        o[i] = E::pack(nlo * E::lo(ws[i]), nhi * E::hi(ws[i]));
        if constexpr (QUANT) amax = fmaxf(amax, fmaxf(fabsf(E::lo(o[i])), fabsf(E::hi(o[i]))));
      }
      if constexpr (QUANT) z[k] = make_uint4(o[0], o[1], o[2], o[3]);
      else *reinterpret_cast<uint4*>(orow + ch * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
  if constexpr (QUANT) {
    amax = block_allreduce_max(amax, smem);
    const float scale = amax > 0.f ? amax / 127.0f : 1.0f;
    if (threadIdx.x == 0) q_scale[row] = scale;
    int8_t* qrow = reinterpret_cast<int8_t*>(out) + (int64_t)row * out_stride;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int ch = threadIdx.x + k * 256;
      if (ch < nchunk) {
        const uint32_t os[4] = {z[k].x, z[k].y, z[k].z, z[k].w};
        uint2 q;
        q.x = quant4(E::lo(os[0]), E::hi(os[0]), E::lo(os[1]), E::hi(os[1]), scale);
        q.y = quant4(E::lo(os[2]), E::hi(os[2]), E::lo(os[3]), E::hi(os[3]), scale);
        *reinterpret_cast<uint2*>(qrow + ch * 8) = q;
      }
    }
  }
}

// Stand-alone activation quantisation, vectorised (16-byte loads, 8-byte stores), the row kept in registers between
// the absmax pass and the rounding pass: hidden <= 8 * 256 * MAXC.
template <typename Tag, int MAXC>
__global__ __launch_bounds__(256) void int8_quant_rows_kernel(int8_t* __restrict__ xq, float* __restrict__ x_scale,
                                                              const uint16_t* __restrict__ x, int hidden,
                                                              int64_t x_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  __shared__ float smem[4];
  const int row = blockIdx.x;
  const int nchunk = hidden / 8;
  const uint16_t* xr = x + row * x_stride;
  uint4 z[MAXC];
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int ch = threadIdx.x + k * 256;
    if (ch < nchunk) {
      z[k] = *reinterpret_cast<const uint4*>(xr + ch * 8);
      const uint32_t ws[4] = {z[k].x, z[k].y, z[k].z, z[k].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(E::lo(ws[i])), fabsf(E::hi(ws[i]))));
    }
  }
  amax = block_allreduce_max(amax, smem);
  const float scale = amax > 0.f ? amax / 127.0f : 1.0f;
  if (threadIdx.x == 0) x_scale[row] = scale;
  int8_t* qrow = xq + (int64_t)row * hidden;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int ch = threadIdx.x + k * 256;
    if (ch < nchunk) {
      const uint32_t os[4] = {z[k].x, z[k].y, z[k].z, z[k].w};
      uint2 q;
      q.x = quant4(E::lo(os[0]), E::hi(os[0]), E::lo(os[1]), E::hi(os[1]), scale);
      q.y = quant4(E::lo(os[2]), E::hi(os[2]), E::lo(os[3]), E::hi(os[3]), scale);
      *reinterpret_cast<uint2*>(qrow + ch * 8) = q;
    }
  }
}

// The fused-add form fed by the split-K partials of the decode-batch linear kernel (skinny_gemm.hip) instead of a
// 16-bit x: x = rnd(sum_s part[s] (+ bias)) — the reduce kernel's arithmetic, slice order and rounding — then exactly
// the ADD path above.  Saves the reduce launch and the round trip of x through memory (down_proj -> next layer's
// input_layernorm; qwen2_old.py:212-216).
// I8: the partials are the exact int32 sums of a W8A8 projection (mi_int8_linear_partials): x = rnd(dequant(sum over
// slices)) with the dequantisation of skinny_reduce_i8_kernel (dequant_i32).  QUANT: the normalised row leaves as int8 +
// scale for the W8A8 linear behind the norm (as rms_norm_kernel<…, QUANT>).
template <typename Tag, int MAXC, bool I8 = false, bool QUANT = false>
__global__ __launch_bounds__(256) void rms_norm_splitk_kernel(uint16_t* __restrict__ out, const float* __restrict__ part,
                                                              int S, const uint16_t* __restrict__ bias,
                                                              uint16_t* __restrict__ residual,
                                                              const uint16_t* __restrict__ w, float eps, int T, int hidden,
                                                              int64_t out_stride, int64_t r_stride,
                                                              const float* __restrict__ p_xs = nullptr,
                                                              const float* __restrict__ p_ws = nullptr,
                                                              float* __restrict__ q_scale = nullptr) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  __shared__ float smem[4];
  const int row = blockIdx.x;
  const int nchunk = hidden / 8;
  uint16_t* rr = residual + row * r_stride;
  uint16_t* orow = out + row * out_stride;
  uint4 z[MAXC], wreg[MAXC];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int ch = threadIdx.x + k * 256;
    if (ch < nchunk) {
      wreg[k] = *reinterpret_cast<const uint4*>(w + ch * 8);
      const float* p0 = part + (int64_t)row * hidden + ch * 8;
      // every slice's loads in flight at once (S <= 16), then summed in slice order
      float4 vl[16], vh[16];
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (s < S) {
          const float* ps = p0 + (int64_t)s * T * hidden;
          vl[s] = *reinterpret_cast<const float4*>(ps);
          vh[s] = *reinterpret_cast<const float4*>(ps + 4);
        }
      float4 lo, hi;
      if constexpr (I8) {
        int4 il = __builtin_bit_cast(int4, vl[0]), ih = __builtin_bit_cast(int4, vh[0]);
#pragma unroll
        for (int s = 1; s < 16; ++s)
          if (s < S) {
            const int4 a = __builtin_bit_cast(int4, vl[s]), b = __builtin_bit_cast(int4, vh[s]);
            il.x += a.x; il.y += a.y; il.z += a.z; il.w += a.w;
            ih.x += b.x; ih.y += b.y; ih.z += b.z; ih.w += b.w;
          }
        float bl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias) {
          const uint4 bv = *reinterpret_cast<const uint4*>(bias + ch * 8);
          bl[0] = E::lo(bv.x); bl[1] = E::hi(bv.x); bl[2] = E::lo(bv.y); bl[3] = E::hi(bv.y);
          bl[4] = E::lo(bv.z); bl[5] = E::hi(bv.z); bl[6] = E::lo(bv.w); bl[7] = E::hi(bv.w);
        }
        const float xs = p_xs[row];
        const float4 w0 = *reinterpret_cast<const float4*>(p_ws + ch * 8), w1 = *reinterpret_cast<const float4*>(p_ws + ch * 8 + 4);
        lo = make_float4(dequant_i32(il.x, xs, w0.x, bl[0]), dequant_i32(il.y, xs, w0.y, bl[1]),
                         dequant_i32(il.z, xs, w0.z, bl[2]), dequant_i32(il.w, xs, w0.w, bl[3]));
        hi = make_float4(dequant_i32(ih.x, xs, w1.x, bl[4]), dequant_i32(ih.y, xs, w1.y, bl[5]),
                         dequant_i32(ih.z, xs, w1.z, bl[6]), dequant_i32(ih.w, xs, w1.w, bl[7]));
      } else {
        lo = vl[0];
        hi = vh[0];
#pragma unroll
        for (int s = 1; s < 16; ++s)
          if (s < S) {
            lo.x += vl[s].x; lo.y += vl[s].y; lo.z += vl[s].z; lo.w += vl[s].w;
            hi.x += vh[s].x; hi.y += vh[s].y; hi.z += vh[s].z; hi.w += vh[s].w;
          }
        if (bias) {
          const uint4 bv = *reinterpret_cast<const uint4*>(bias + ch * 8);
          lo.x += E::lo(bv.x); lo.y += E::hi(bv.x); lo.z += E::lo(bv.y); lo.w += E::hi(bv.y);
          hi.x += E::lo(bv.z); hi.y += E::hi(bv.z); hi.z += E::lo(bv.w); hi.w += E::hi(bv.w);
        }
      }
      uint4 a = make_uint4(E::pack(lo.x, lo.y), E::pack(lo.z, lo.w), E::pack(hi.x, hi.y), E::pack(hi.z, hi.w));
      const uint4 b = *reinterpret_cast<const uint4*>(rr + ch * 8);
      a.x = E::pack(E::lo(a.x) + E::lo(b.x), E::hi(a.x) + E::hi(b.x));
      a.y = E::pack(E::lo(a.y) + E::lo(b.y), E::hi(a.y) + E::hi(b.y));
      a.z = E::pack(E::lo(a.z) + E::lo(b.z), E::hi(a.z) + E::hi(b.z));
      a.w = E::pack(E::lo(a.w) + E::lo(b.w), E::hi(a.w) + E::hi(b.w));
      *reinterpret_cast<uint4*>(rr + ch * 8) = a;
      z[k] = a;
      const uint32_t ws[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float l = E::lo(ws[i]), h = E::hi(ws[i]);
        ss += l * l + h * h;
      }
    }
  }
  ss = block_allreduce_sum(ss, smem);
  const float inv = __builtin_amdgcn_rsqf(ss / (float)hidden + eps);
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int ch = threadIdx.x + k * 256;
    if (ch < nchunk) {
      const uint4 wv = wreg[k];
      const uint32_t zs[4] = {z[k].x, z[k].y, z[k].z, z[k].w};
      const uint32_t ws[4] = {wv.x, wv.y, wv.z, wv.w};
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float nlo = E::from_bits(E::to_bits(E::lo(zs[i]) * inv));
        const float nhi = E::from_bits(E::to_bits(E::hi(zs[i]) * inv));
        o[i] = E::pack(nlo * E::lo(ws[i]), nhi * E::hi(ws[i]));
        if constexpr (QUANT) amax = fmaxf(amax, fmaxf(fabsf(E::lo(o[i])), fabsf(E::hi(o[i]))));
      }
      if constexpr (QUANT) z[k] = make_uint4(o[0], o[1], o[2], o[3]);
      else *reinterpret_cast<uint4*>(orow + ch * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
  if constexpr (QUANT) {
    amax = block_allreduce_max(amax, smem);
    const float scale = amax > 0.f ? amax / 127.0f : 1.0f;
    if (threadIdx.x == 0) q_scale[row] = scale;
    int8_t* qrow = reinterpret_cast<int8_t*>(out) + (int64_t)row * out_stride;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int ch = threadIdx.x + k * 256;
      if (ch < nchunk) {
        const uint32_t os[4] = {z[k].x, z[k].y, z[k].z, z[k].w};
        uint2 q;
        q.x = quant4(E::lo(os[0]), E::hi(os[0]), E::lo(os[1]), E::hi(os[1]), scale);
        q.y = quant4(E::lo(os[2]), E::hi(os[2]), E::lo(os[3]), E::hi(os[3]), scale);
        *reinterpret_cast<uint2*>(qrow + ch * 8) = q;
      }
    }
  }
}

// Any hidden size / alignment: scalar, two passes over memory (row re-read hits L2).
template <typename Tag, bool ADD>
__global__ __launch_bounds__(256) void rms_norm_generic_kernel(uint16_t* out, const uint16_t* x,
                                                               uint16_t* residual,
                                                               const uint16_t* w, float eps,
                                                               int hidden, int64_t x_stride,
                                                               int64_t out_stride,
                                                               int64_t r_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  __shared__ float smem[4];
  const int row = blockIdx.x;
  const uint16_t* xr = x + row * x_stride;
  uint16_t* rr = ADD ? residual + row * r_stride : nullptr;
  float ss = 0.f;
  for (int i = threadIdx.x; i < hidden; i += 256) {
    float v = E::from_bits(xr[i]);
    if constexpr (ADD) {
      const uint16_t zb = E::to_bits(v + E::from_bits(rr[i]));
      rr[i] = zb;
      v = E::from_bits(zb);
    }
    ss += v * v;
  }
  ss = block_allreduce_sum(ss, smem);  // barriers also order the rr[] writes above
  const float inv = __builtin_amdgcn_rsqf(ss / (float)hidden + eps);
  for (int i = threadIdx.x; i < hidden; i += 256) {
    const float v = ADD ? E::from_bits(rr[i]) : E::from_bits(xr[i]);
    const float n = E::from_bits(E::to_bits(v * inv));
    out[row * out_stride + i] = E::to_bits(n * E::from_bits(w[i]));
  }
}

// out = rnd(rnd(silu(g)) * u): silu in fp32, rounded, then the gate product rounded.
template <typename Tag>
__global__ __launch_bounds__(256) void silu_mul_kernel(uint16_t* __restrict__ out,
                                                       const uint16_t* __restrict__ x, int inter,
                                                       int64_t x_stride, int64_t out_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  const int row = blockIdx.y;
  const uint16_t* g = x + row * x_stride;
  const uint16_t* u = g + inter;
  uint16_t* o = out + row * out_stride;
  const int nchunk = inter / 8;
  for (int ch = blockIdx.x * 256 + threadIdx.x; ch < nchunk; ch += gridDim.x * 256) {
    const uint4 a = *reinterpret_cast<const uint4*>(g + ch * 8);
    const uint4 b = *reinterpret_cast<const uint4*>(u + ch * 8);
    const uint32_t as[4] = {a.x, a.y, a.z, a.w};
    const uint32_t bs[4] = {b.x, b.y, b.z, b.w};
    uint32_t r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g0 = E::lo(as[i]), g1 = E::hi(as[i]);
      const float s0 = E::from_bits(E::to_bits(g0 / (1.f + __expf(-g0))));
      const float s1 = E::from_bits(E::to_bits(g1 / (1.f + __expf(-g1))));
      r[i] = E::pack(s0 * E::lo(bs[i]), s1 * E::hi(bs[i]));
    }
    *reinterpret_cast<uint4*>(o + ch * 8) = make_uint4(r[0], r[1], r[2], r[3]);
  }
}

template <typename Tag>
__global__ __launch_bounds__(256) void silu_mul_scalar_kernel(uint16_t* out, const uint16_t* x,
                                                              int inter, int64_t x_stride,
                                                              int64_t out_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  const int row = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < inter; i += gridDim.x * 256) {
    const float g = E::from_bits(x[row * x_stride + i]);
    const float u = E::from_bits(x[row * x_stride + inter + i]);
    const float s = E::from_bits(E::to_bits(g / (1.f + __expf(-g))));
    out[row * out_stride + i] = E::to_bits(s * u);
  }
}

// logits row: mask ranges to -1e9 in place, then first-index argmax.
template <int DT>  // MI_DTYPE_*
__global__ __launch_bounds__(256) void mask_argmax_kernel(void* logits, int64_t* ids, int vocab,
                                                          int64_t row_stride, int lo0, int hi0,
                                                          int lo1, int hi1, int col) {
  scalar_cache_acquire();
  __shared__ float sv[4];
  __shared__ int si[4];
  const int row = blockIdx.x;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  if constexpr (DT != MI_DTYPE_F32) {
    // 16-bit rows: eight logits per 16-byte load, the mask applied in registers, only chunks that hold a masked column
    // written back (one 2-byte load - store - load round trip per element was 15 us for 256 rows of 8448)
    using E = Elem<typename std::conditional<DT == MI_DTYPE_BF16, bf16_tag, f16_tag>::type>;
    uint16_t* rowp = reinterpret_cast<uint16_t*>(logits) + row * row_stride;
    if (vocab % 8 == 0 && row_stride % 8 == 0 && !((uintptr_t)logits & 15)) {
      const uint16_t neg = E::to_bits(-1e9f);   // f16 saturates to -inf, as torch does
      for (int c = threadIdx.x; c < vocab / 8; c += 256) {
        uint4 q = *reinterpret_cast<const uint4*>(rowp + 8 * c);
        uint16_t h[8];
        __builtin_memcpy(h, &q, 16);
        bool any = false;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = 8 * c + e;
          const bool masked = (i >= lo0 && i < hi0) || (i >= lo1 && i < hi1) || i == col;
          h[e] = masked ? neg : h[e];
          any |= masked;
          const float v = E::from_bits(h[e]);
          if (v > best || (v == best && i < besti)) { best = v; besti = i; }
        }
        if (any) {
          __builtin_memcpy(&q, h, 16);
          *reinterpret_cast<uint4*>(rowp + 8 * c) = q;
        }
      }
      goto reduce;
    }
  }
  for (int i = threadIdx.x; i < vocab; i += 256) {
    const bool masked = (i >= lo0 && i < hi0) || (i >= lo1 && i < hi1) || i == col;
    float v;
    if constexpr (DT == MI_DTYPE_F32) {
      float* p = reinterpret_cast<float*>(logits) + row * row_stride + i;
      if (masked) *p = -1e9f;
      v = *p;
    } else {
      using E = Elem<typename std::conditional<DT == MI_DTYPE_BF16, bf16_tag, f16_tag>::type>;
      uint16_t* p = reinterpret_cast<uint16_t*>(logits) + row * row_stride + i;
      if (masked) *p = E::to_bits(-1e9f);  // f16 saturates to -inf, as torch does
      v = E::from_bits(*p);
    }
    if (v > best || (v == best && i < besti)) { best = v; besti = i; }
  }
reduce:
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i)
      if (sv[i] > best || (sv[i] == best && si[i] < besti)) { best = sv[i]; besti = si[i]; }
    ids[row] = besti == 0x7fffffff ? 0 : besti;
  }
}

// out[t] = embed[ids[t]] (+ reply[idx[t]]): one workgroup per token, 16 B per lane.
template <typename Tag>
__global__ __launch_bounds__(256) void talker_embeds_kernel(uint16_t* __restrict__ out,
                                                            const uint16_t* __restrict__ table,
                                                            const int64_t* __restrict__ ids,
                                                            const uint16_t* __restrict__ reply,
                                                            const int64_t* __restrict__ ridx, int dim,
                                                            int vocab, int64_t nreply,
                                                            int64_t out_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  const int t = blockIdx.x;
  const int64_t id = ids[t];
  const bool ok = id >= 0 && id < vocab;
  int64_t r = (reply && ridx) ? ridx[t] : -1;
  if (r >= nreply) r = -1;
  const uint16_t* src = table + (ok ? id : 0) * (int64_t)dim;
  for (int ch = threadIdx.x; ch < dim / 8; ch += 256) {
    uint4 a = ok ? *reinterpret_cast<const uint4*>(src + ch * 8) : make_uint4(0, 0, 0, 0);
    if (r >= 0) {
      const uint4 b = *reinterpret_cast<const uint4*>(reply + r * (int64_t)dim + ch * 8);
      a.x = E::pack(E::lo(a.x) + E::lo(b.x), E::hi(a.x) + E::hi(b.x));
      a.y = E::pack(E::lo(a.y) + E::lo(b.y), E::hi(a.y) + E::hi(b.y));
      a.z = E::pack(E::lo(a.z) + E::lo(b.z), E::hi(a.z) + E::hi(b.z));
      a.w = E::pack(E::lo(a.w) + E::lo(b.w), E::hi(a.w) + E::hi(b.w));
    }
    *reinterpret_cast<uint4*>(out + t * out_stride + ch * 8) = a;
  }
}

// out_a[t] = out_b[t] = table[ids[t]], ids_out[t] = ids[t]: one workgroup per token, 16 B per lane (any 16-bit type).
__global__ __launch_bounds__(128) void embed_codes_kernel(uint16_t* __restrict__ out_a, uint16_t* __restrict__ out_b,
                                                         int64_t* __restrict__ ids_out, const uint16_t* __restrict__ table,
                                                         const int64_t* __restrict__ ids, int dim, int vocab, int64_t a_stride,
                                                         int64_t b_stride, int64_t ids_stride) {
  scalar_cache_acquire();
  const int t = blockIdx.x;
  const int64_t id = ids[t];
  const bool ok = id >= 0 && id < vocab;
  const uint16_t* src = table + (ok ? id : 0) * (int64_t)dim;
  if (ids_out && threadIdx.x == 0) ids_out[t * ids_stride] = id;
  for (int ch = threadIdx.x; ch < dim / 8; ch += 128) {
    const uint4 a = ok ? *reinterpret_cast<const uint4*>(src + ch * 8) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(out_a + t * a_stride + ch * 8) = a;
    if (out_b) *reinterpret_cast<uint4*>(out_b + t * b_stride + ch * 8) = a;
  }
}

template <typename Tag>
int launch_rms(void* out, const void* x, void* residual, const void* w, float eps, int T, int H,
               int64_t xs, int64_t os, int64_t rs, hipStream_t s) {
  auto o = reinterpret_cast<uint16_t*>(out);
  auto xx = reinterpret_cast<const uint16_t*>(x);
  auto rr = reinterpret_cast<uint16_t*>(residual);
  auto ww = reinterpret_cast<const uint16_t*>(w);
  const bool add = residual != nullptr;
  const bool vec = H % 8 == 0 && xs % 8 == 0 && os % 8 == 0 && rs % 8 == 0 && H <= 8 * 256 * 4 &&
                   !(((uintptr_t)out | (uintptr_t)x | (uintptr_t)residual | (uintptr_t)w) & 15);
  const dim3 grid(T), block(256);
  if (vec) {
    const int c = (H / 8 + 255) / 256;
#define MI_RMS(ADD, C) \
  hipLaunchKernelGGL((rms_norm_kernel<Tag, ADD, C>), grid, block, 0, s, o, xx, rr, ww, eps, H, xs, os, rs)
    if (add) { if (c <= 1) MI_RMS(true, 1); else if (c <= 2) MI_RMS(true, 2); else MI_RMS(true, 4); }
    else { if (c <= 1) MI_RMS(false, 1); else if (c <= 2) MI_RMS(false, 2); else MI_RMS(false, 4); }
#undef MI_RMS
  } else if (add) {
    hipLaunchKernelGGL((rms_norm_generic_kernel<Tag, true>), grid, block, 0, s, o, xx, rr, ww, eps, H, xs, os, rs);
  } else {
    hipLaunchKernelGGL((rms_norm_generic_kernel<Tag, false>), grid, block, 0, s, o, xx, rr, ww, eps, H, xs, os, rs);
  }
  return launch_status();
}

template <typename Tag>
int launch_rms_int8(int8_t* xq, float* q_scale, const void* x, void* residual, const void* w, float eps, int T, int H,
                    int64_t xs, int64_t rs, hipStream_t s) {
  auto o = reinterpret_cast<uint16_t*>(xq);
  auto xx = reinterpret_cast<const uint16_t*>(x);
  auto rr = reinterpret_cast<uint16_t*>(residual);
  auto ww = reinterpret_cast<const uint16_t*>(w);
  if (H % 8 || xs % 8 || rs % 8 || H > 8 * 256 * 4 || (((uintptr_t)xq | (uintptr_t)x | (uintptr_t)residual | (uintptr_t)w) & 15))
    return MI_ERR_UNSUPPORTED;
  const int c = (H / 8 + 255) / 256;
  const dim3 grid(T), block(256);
#define MI_RMSQ(ADD, C) \
  hipLaunchKernelGGL((rms_norm_kernel<Tag, ADD, C, true>), grid, block, 0, s, o, xx, rr, ww, eps, H, xs, (int64_t)H, rs, q_scale)
  if (residual) { if (c <= 1) MI_RMSQ(true, 1); else if (c <= 2) MI_RMSQ(true, 2); else MI_RMSQ(true, 4); }
  else { if (c <= 1) MI_RMSQ(false, 1); else if (c <= 2) MI_RMSQ(false, 2); else MI_RMSQ(false, 4); }
#undef MI_RMSQ
  return launch_status();
}

}  // namespace
}  // namespace mi

// RMSNorm (x) / fused residual-add RMSNorm (x, residual -> residual updated in place) whose normalised output leaves
// as the W8A8 activation: int8 [T, hidden] + fp32 scale [T] — `norm` followed by `mi_dynamic_int8_quant`, one launch,
// same bits.  x is NOT modified.  hidden % 8 == 0, <= 8192; 16-byte aligned rows.
extern "C" int mi_rms_norm_int8(int8_t* xq, float* x_scale, const void* x, void* residual, const void* weight, float eps,
                                int32_t num_tokens, int32_t hidden, int64_t x_token_stride,
                                int64_t residual_token_stride, int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!xq || !x_scale || !x || !weight || num_tokens < 0 || hidden <= 0) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MI_DTYPE_BF16)
    return launch_rms_int8<bf16_tag>(xq, x_scale, x, residual, weight, eps, num_tokens, hidden, x_token_stride,
                                     residual_token_stride, s);
  if (dtype == MI_DTYPE_F16)
    return launch_rms_int8<f16_tag>(xq, x_scale, x, residual, weight, eps, num_tokens, hidden, x_token_stride,
                                    residual_token_stride, s);
  return MI_ERR_UNSUPPORTED;
}

// Vectorised form of mi_dynamic_int8_quant for rows of up to 24576 elements (hidden % 8 == 0, 16-byte aligned):
// same results; MI_ERR_UNSUPPORTED otherwise (callers use mi_dynamic_int8_quant).
extern "C" int mi_int8_quant_rows(int8_t* xq, float* x_scale, const void* x, int32_t num_tokens, int32_t hidden,
                                  int64_t x_token_stride, int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!xq || !x_scale || !x || num_tokens < 0 || hidden <= 0) return MI_ERR_INVALID_ARG;
  if (hidden % 8 || x_token_stride % 8 || hidden > 8 * 256 * 12 || (((uintptr_t)xq & 7) | ((uintptr_t)x & 15)))
    return MI_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto xx = reinterpret_cast<const uint16_t*>(x);
  const int c = (hidden / 8 + 255) / 256;
  const dim3 grid(num_tokens), block(256);
#define MI_Q(TAG, C) hipLaunchKernelGGL((int8_quant_rows_kernel<TAG, C>), grid, block, 0, s, xq, x_scale, xx, hidden, x_token_stride)
#define MI_Q_T(TAG)                    \
  do {                                 \
    if (c <= 1) MI_Q(TAG, 1);          \
    else if (c <= 2) MI_Q(TAG, 2);     \
    else if (c <= 4) MI_Q(TAG, 4);     \
    else MI_Q(TAG, 12);                \
  } while (0)
  if (dtype == MI_DTYPE_BF16) MI_Q_T(bf16_tag);
  else if (dtype == MI_DTYPE_F16) MI_Q_T(f16_tag);
  else return MI_ERR_UNSUPPORTED;
#undef MI_Q_T
#undef MI_Q
  return launch_status();
}

extern "C" int mi_rms_norm(void* out, const void* x, const void* weight, float eps,
                           int32_t num_tokens, int32_t hidden, int64_t x_token_stride,
                           int64_t out_token_stride, int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!out || !x || !weight || num_tokens < 0 || hidden <= 0) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MI_DTYPE_BF16)
    return launch_rms<bf16_tag>(out, x, nullptr, weight, eps, num_tokens, hidden, x_token_stride,
                                out_token_stride, 0, s);
  if (dtype == MI_DTYPE_F16)
    return launch_rms<f16_tag>(out, x, nullptr, weight, eps, num_tokens, hidden, x_token_stride,
                               out_token_stride, 0, s);
  return MI_ERR_UNSUPPORTED;
}

extern "C" int mi_fused_add_rms_norm(void* x, void* residual, const void* weight, float eps,
                                     int32_t num_tokens, int32_t hidden, int64_t x_token_stride,
                                     int64_t residual_token_stride, int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!x || !residual || !weight || num_tokens < 0 || hidden <= 0) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MI_DTYPE_BF16)
    return launch_rms<bf16_tag>(x, x, residual, weight, eps, num_tokens, hidden, x_token_stride,
                                x_token_stride, residual_token_stride, s);
  if (dtype == MI_DTYPE_F16)
    return launch_rms<f16_tag>(x, x, residual, weight, eps, num_tokens, hidden, x_token_stride,
                               x_token_stride, residual_token_stride, s);
  return MI_ERR_UNSUPPORTED;
}

extern "C" int mi_fused_add_rms_norm_splitk(void* out, const void* partials, int32_t split, const void* bias,
                                            void* residual, const void* weight, float eps, int32_t num_tokens,
                                            int32_t hidden, int64_t out_token_stride, int64_t residual_token_stride,
                                            int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!out || !partials || !residual || !weight || num_tokens < 0 || hidden <= 0 || split < 1 || split > 16)
    return MI_ERR_INVALID_ARG;
  if (hidden % 8 || out_token_stride % 8 || residual_token_stride % 8 || hidden > 8 * 256 * 4) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)out | (uintptr_t)partials | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)bias) & 15)
    return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto o = reinterpret_cast<uint16_t*>(out);
  auto pp = reinterpret_cast<const float*>(partials);
  auto bb = reinterpret_cast<const uint16_t*>(bias);
  auto rr = reinterpret_cast<uint16_t*>(residual);
  auto ww = reinterpret_cast<const uint16_t*>(weight);
  const int c = (hidden / 8 + 255) / 256;
  const dim3 grid(num_tokens), block(256);
#define MI_RMS_SK(TAG, C)                                                                                       \
  hipLaunchKernelGGL((rms_norm_splitk_kernel<TAG, C>), grid, block, 0, s, o, pp, split, bb, rr, ww, eps, num_tokens, \
                     hidden, out_token_stride, residual_token_stride)
#define MI_RMS_SK_T(TAG)              \
  do {                                \
    if (c <= 1) MI_RMS_SK(TAG, 1);    \
    else if (c <= 2) MI_RMS_SK(TAG, 2); \
    else MI_RMS_SK(TAG, 4);           \
  } while (0)
  if (dtype == MI_DTYPE_BF16) MI_RMS_SK_T(bf16_tag);
  else if (dtype == MI_DTYPE_F16) MI_RMS_SK_T(f16_tag);
  else return MI_ERR_UNSUPPORTED;
#undef MI_RMS_SK_T
#undef MI_RMS_SK
  return launch_status();
}

// The fused-add RMSNorm fed by the exact int32 split-K partials of a W8A8 projection (mi_int8_linear_partials):
// x = rnd(dequant(sum over slices) (+ bias)), residual <- x + residual, then the norm — the int8 reduce launch and one
// round trip of [T, H] saved.  q_scale == NULL: 16-bit output `out` [T, out_stride]; else the normalised row leaves as
// int8 `out` [T, hidden] + q_scale [T] for the W8A8 linear behind the norm.  Bit-identical to mi_int8_linear (with its
// reduce) followed by mi_fused_add_rms_norm (followed by mi_dynamic_int8_quant).
extern "C" int mi_fused_add_rms_norm_splitk_int8(void* out, float* q_scale, const void* partials, int32_t split,
                                                 const float* x_scale, const float* w_scale, const void* bias,
                                                 void* residual, const void* weight, float eps, int32_t num_tokens,
                                                 int32_t hidden, int64_t out_token_stride, int64_t residual_token_stride,
                                                 int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!out || !partials || !x_scale || !w_scale || !residual || !weight || num_tokens < 0 || hidden <= 0 || split < 1 ||
      split > 16)
    return MI_ERR_INVALID_ARG;
  if (hidden % 8 || residual_token_stride % 8 || hidden > 8 * 256 * 4 || (!q_scale && out_token_stride % 8))
    return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)out | (uintptr_t)partials | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)w_scale) & 15)
    return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto o = reinterpret_cast<uint16_t*>(out);
  auto pp = reinterpret_cast<const float*>(partials);
  auto bb = reinterpret_cast<const uint16_t*>(bias);
  auto rr = reinterpret_cast<uint16_t*>(residual);
  auto ww = reinterpret_cast<const uint16_t*>(weight);
  const int c = (hidden / 8 + 255) / 256;
  const dim3 grid(num_tokens), block(256);
  const int64_t os = q_scale ? (int64_t)hidden : out_token_stride;
#define MI_RMS_SKI(TAG, C, Q)                                                                                            \
  hipLaunchKernelGGL((rms_norm_splitk_kernel<TAG, C, true, Q>), grid, block, 0, s, o, pp, split, bb, rr, ww, eps, num_tokens, \
                     hidden, os, residual_token_stride, x_scale, w_scale, q_scale)
#define MI_RMS_SKI_T(TAG)                                                        \
  do {                                                                           \
    if (q_scale) {                                                               \
      if (c <= 1) MI_RMS_SKI(TAG, 1, true); else if (c <= 2) MI_RMS_SKI(TAG, 2, true); else MI_RMS_SKI(TAG, 4, true);    \
    } else {                                                                     \
      if (c <= 1) MI_RMS_SKI(TAG, 1, false); else if (c <= 2) MI_RMS_SKI(TAG, 2, false); else MI_RMS_SKI(TAG, 4, false); \
    }                                                                            \
  } while (0)
  if (dtype == MI_DTYPE_BF16) MI_RMS_SKI_T(bf16_tag);
  else if (dtype == MI_DTYPE_F16) MI_RMS_SKI_T(f16_tag);
  else return MI_ERR_UNSUPPORTED;
#undef MI_RMS_SKI_T
#undef MI_RMS_SKI
  return launch_status();
}

extern "C" int mi_silu_and_mul(void* out, const void* x, int32_t num_tokens, int32_t inter,
                               int64_t x_token_stride, int64_t out_token_stride, int dtype,
                               void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!out || !x || num_tokens < 0 || inter <= 0) return MI_ERR_INVALID_ARG;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto o = reinterpret_cast<uint16_t*>(out);
  auto xx = reinterpret_cast<const uint16_t*>(x);
  const bool vec = inter % 8 == 0 && x_token_stride % 8 == 0 && out_token_stride % 8 == 0 &&
                   !(((uintptr_t)out | (uintptr_t)x) & 15);
  const int per_row = vec ? inter / 8 : inter;
  int gx = (per_row + 255) / 256;
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  const dim3 grid(gx, num_tokens), block(256);
  if (dtype == MI_DTYPE_BF16) {
    if (vec) hipLaunchKernelGGL((silu_mul_kernel<bf16_tag>), grid, block, 0, s, o, xx, inter, x_token_stride, out_token_stride);
    else hipLaunchKernelGGL((silu_mul_scalar_kernel<bf16_tag>), grid, block, 0, s, o, xx, inter, x_token_stride, out_token_stride);
  } else {
    if (vec) hipLaunchKernelGGL((silu_mul_kernel<f16_tag>), grid, block, 0, s, o, xx, inter, x_token_stride, out_token_stride);
    else hipLaunchKernelGGL((silu_mul_scalar_kernel<f16_tag>), grid, block, 0, s, o, xx, inter, x_token_stride, out_token_stride);
  }
  return launch_status();
}

extern "C" int mi_mask_logits_argmax(void* logits, int64_t* ids, int32_t num_tokens,
                                     int32_t vocab, int64_t row_stride, int32_t mask_lo0,
                                     int32_t mask_hi0, int32_t mask_lo1, int32_t mask_hi1,
                                     int32_t mask_col, int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!logits || !ids || num_tokens < 0 || vocab <= 0) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(num_tokens), block(256);
  switch (dtype) {
    case MI_DTYPE_F32:
      hipLaunchKernelGGL((mask_argmax_kernel<MI_DTYPE_F32>), grid, block, 0, s, logits, ids, vocab, row_stride, mask_lo0, mask_hi0, mask_lo1, mask_hi1, mask_col);
      break;
    case MI_DTYPE_BF16:
      hipLaunchKernelGGL((mask_argmax_kernel<MI_DTYPE_BF16>), grid, block, 0, s, logits, ids, vocab, row_stride, mask_lo0, mask_hi0, mask_lo1, mask_hi1, mask_col);
      break;
    case MI_DTYPE_F16:
      hipLaunchKernelGGL((mask_argmax_kernel<MI_DTYPE_F16>), grid, block, 0, s, logits, ids, vocab, row_stride, mask_lo0, mask_hi0, mask_lo1, mask_hi1, mask_col);
      break;
    default:
      return MI_ERR_UNSUPPORTED;
  }
  return launch_status();
}

extern "C" int mi_talker_decode_embeds(void* out, const void* embed_table, const int64_t* token_ids,
                                       const void* reply_rows, const int64_t* reply_idx,
                                       int32_t num_tokens, int32_t embed_dim, int32_t vocab,
                                       int64_t num_reply_rows, int64_t out_token_stride, int dtype,
                                       void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!out || !embed_table || !token_ids || num_tokens < 0 || embed_dim <= 0 || vocab <= 0)
    return MI_ERR_INVALID_ARG;
  if (embed_dim % 8 || out_token_stride % 8) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)out | (uintptr_t)embed_table | (uintptr_t)reply_rows) & 15) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto o = reinterpret_cast<uint16_t*>(out);
  auto tb = reinterpret_cast<const uint16_t*>(embed_table);
  auto rp = reinterpret_cast<const uint16_t*>(reply_rows);
  if (dtype == MI_DTYPE_BF16)
    hipLaunchKernelGGL((talker_embeds_kernel<bf16_tag>), dim3(num_tokens), dim3(256), 0, s, o, tb, token_ids, rp, reply_idx, embed_dim, vocab, num_reply_rows, out_token_stride);
  else if (dtype == MI_DTYPE_F16)
    hipLaunchKernelGGL((talker_embeds_kernel<f16_tag>), dim3(num_tokens), dim3(256), 0, s, o, tb, token_ids, rp, reply_idx, embed_dim, vocab, num_reply_rows, out_token_stride);
  else
    return MI_ERR_UNSUPPORTED;
  return launch_status();
}

extern "C" int mi_embed_codes(void* out_a, void* out_b, int64_t* ids_out, const void* embed_table, const int64_t* token_ids,
                              int32_t num_tokens, int32_t embed_dim, int32_t vocab, int64_t out_a_token_stride,
                              int64_t out_b_token_stride, int64_t ids_out_stride, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!out_a || !embed_table || !token_ids || num_tokens < 0 || embed_dim <= 0 || vocab <= 0) return MI_ERR_INVALID_ARG;
  if (embed_dim % 8 || out_a_token_stride % 8 || (out_b && out_b_token_stride % 8)) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)out_a | (uintptr_t)out_b | (uintptr_t)embed_table) & 15) return MI_ERR_INVALID_ARG;
  hipLaunchKernelGGL(embed_codes_kernel, dim3(num_tokens), dim3(128), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<uint16_t*>(out_a), reinterpret_cast<uint16_t*>(out_b), ids_out,
                     reinterpret_cast<const uint16_t*>(embed_table), token_ids, embed_dim, vocab, out_a_token_stride,
                     out_b_token_stride, ids_out_stride);
  return launch_status();
}
