// A13 — W8A8 int8 linear for gfx950: dynamic per-token symmetric activation quantisation and an
// int8 x int8 -> int32 MFMA GEMM with the dequantisation (+ bias) fused in the epilogue.
//
// Scheme restated from vllm_omni/quantization/int8_config.py:180-212 (weight int8 [N, K] + fp32
// per-output-channel scale), :323-342 (dynamic, symmetric, per-token activations), :391-418 (online
// weight quantisation).  The reference dispatches this to CUTLASS on CUDA and refuses ROCm
// (:77-83,326-331), so there is no reference output on AMD; rounding mode (half-to-even) and
// clamp range ([-128, 127]) follow the oracle and are documented as unpinned (SURVEY §8c vii).
#include <cstdlib>

#include "common.h"

namespace mi {
namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

// one workgroup per token row: absmax -> scale = absmax / 127 (1 for an all-zero row) -> round
template <typename Tag>
__global__ __launch_bounds__(256) void dynamic_int8_quant_kernel(int8_t* __restrict__ xq,
                                                                 float* __restrict__ x_scale,
                                                                 const uint16_t* __restrict__ x, int hidden,
                                                                 int64_t x_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  __shared__ float smem[4];
  const int row = blockIdx.x;
  const uint16_t* xr = x + row * x_stride;
  float amax = 0.f;
  for (int i = threadIdx.x; i < hidden; i += 256) amax = fmaxf(amax, fabsf(E::from_bits(xr[i])));
  amax = wave_allreduce_max(amax);
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(smem[0], smem[1]), fmaxf(smem[2], smem[3]));
  const float scale = amax > 0.f ? amax / 127.0f : 1.0f;
  if (threadIdx.x == 0) x_scale[row] = scale;
  for (int i = threadIdx.x; i < hidden; i += 256) {
    float q = rintf(E::from_bits(xr[i]) / scale);  // round-half-to-even, exact division
    q = fminf(fmaxf(q, -128.f), 127.f);
    xq[(int64_t)row * hidden + i] = (int8_t)q;
  }
}

// out[m, n] = (sum_k xq[m, k] * wq[n, k]) * x_scale[m] * w_scale[n] (+ bias[n]).
// Workgroup = 4 waves = 64 x 64 outputs; wave = 32 x 32 = 2 x 2 tiles of v_mfma_i32_16x16x64_i8.
// Both operands are K-contiguous, and a lane's fragment is 16 consecutive k of one row/column, so
// fragments are 16-B global loads (re-use across the workgroup's waves comes from L1/L2).
template <typename Tag>
__global__ __launch_bounds__(256) void int8_gemm_dequant_kernel(uint16_t* __restrict__ out,
                                                                const int8_t* __restrict__ xq,
                                                                const float* __restrict__ x_scale,
                                                                const int8_t* __restrict__ wq,
                                                                const float* __restrict__ w_scale,
                                                                const uint16_t* __restrict__ bias, int m, int n,
                                                                int k, int64_t out_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int row0 = blockIdx.y * 64 + (wid >> 1) * 32;
  const int col0 = blockIdx.x * 64 + (wid & 1) * 32;
  const int r = lane & 15, g = lane >> 4;
  // clamp the rows/columns a lane reads so edge tiles stay in bounds; stores are masked
  const int8_t* a0 = xq + (int64_t)min(row0 + r, m - 1) * k + 16 * g;
  const int8_t* a1 = xq + (int64_t)min(row0 + 16 + r, m - 1) * k + 16 * g;
  const int8_t* b0 = wq + (int64_t)min(col0 + r, n - 1) * k + 16 * g;
  const int8_t* b1 = wq + (int64_t)min(col0 + 16 + r, n - 1) * k + 16 * g;
  i32x4_t c00 = {0, 0, 0, 0}, c01 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
  for (int kk = 0; kk < k; kk += 64) {
    const i32x4_t fa0 = *reinterpret_cast<const i32x4_t*>(a0 + kk);
    const i32x4_t fa1 = *reinterpret_cast<const i32x4_t*>(a1 + kk);
    const i32x4_t fb0 = *reinterpret_cast<const i32x4_t*>(b0 + kk);
    const i32x4_t fb1 = *reinterpret_cast<const i32x4_t*>(b1 + kk);
    c00 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa0, fb0, c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa0, fb1, c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa1, fb0, c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa1, fb1, c11, 0, 0, 0);
  }
  // C/D map: column = lane & 15, row = 4 * (lane >> 4) + i
  auto store = [&](const i32x4_t& c, int rbase, int cbase) {
    const int col = cbase + r;
    if (col >= n) return;
    const float ws = w_scale[col];
    const float bv = bias ? E::from_bits(bias[col]) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rbase + 4 * g + i;
      if (row < m) out[row * out_stride + col] = E::to_bits(dequant_i32(c[i], x_scale[row], ws, bv));
    }
  };
  store(c00, row0, col0);
  store(c01, row0, col0 + 16);
  store(c10, row0 + 16, col0);
  store(c11, row0 + 16, col0 + 16);
}

// Dequant epilogue for an int32 accumulator tile produced by a library int8 GEMM:
// out[m, n] = acc[m, n] * x_scale[m] * w_scale[n] (+ bias[n]).  4 columns per lane (16-B reads, 8-B writes).
template <typename Tag>
__global__ __launch_bounds__(256) void int8_dequant_kernel(uint16_t* __restrict__ out, const int32_t* __restrict__ acc,
                                                           const float* __restrict__ x_scale,
                                                           const float* __restrict__ w_scale,
                                                           const uint16_t* __restrict__ bias, int n,
                                                           int64_t acc_stride, int64_t out_stride) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  const int row = blockIdx.y;
  const float xs = x_scale[row];
  for (int c = (blockIdx.x * 256 + threadIdx.x) * 4; c < n; c += gridDim.x * 1024) {
    const int4 a = *reinterpret_cast<const int4*>(acc + row * acc_stride + c);
    const float4 ws = *reinterpret_cast<const float4*>(w_scale + c);
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (bias) {
      const uint2 bb = *reinterpret_cast<const uint2*>(bias + c);
      b0 = E::lo(bb.x); b1 = E::hi(bb.x); b2 = E::lo(bb.y); b3 = E::hi(bb.y);
    }
    uint2 r;
    r.x = E::pack(dequant_i32(a.x, xs, ws.x, b0), dequant_i32(a.y, xs, ws.y, b1));
    r.y = E::pack(dequant_i32(a.z, xs, ws.z, b2), dequant_i32(a.w, xs, ws.w, b3));
    *reinterpret_cast<uint2*>(out + row * out_stride + c) = r;
  }
}

}  // namespace
}  // namespace mi

extern "C" int mi_int8_dequant(void* out, const int32_t* acc, const float* x_scale, const float* w_scale,
                               const void* bias, int32_t m, int32_t n, int64_t acc_row_stride,
                               int64_t out_row_stride, int dtype, void* stream) {
  using namespace mi;
  if (m == 0 || n == 0) return MI_OK;
  if (!out || !acc || !x_scale || !w_scale || m < 0 || n < 0) return MI_ERR_INVALID_ARG;
  if (n % 4 || acc_row_stride % 4 || out_row_stride % 4) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)acc | (uintptr_t)w_scale) & 15 || ((uintptr_t)out | (uintptr_t)bias) & 7) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int gx = (n / 4 + 255) / 256;
  if (gx > 64) gx = 64;
  const dim3 grid(gx, m), block(256);
  auto o = reinterpret_cast<uint16_t*>(out);
  auto b = reinterpret_cast<const uint16_t*>(bias);
  if (dtype == MI_DTYPE_BF16)
    hipLaunchKernelGGL((int8_dequant_kernel<bf16_tag>), grid, block, 0, s, o, acc, x_scale, w_scale, b, n, acc_row_stride, out_row_stride);
  else if (dtype == MI_DTYPE_F16)
    hipLaunchKernelGGL((int8_dequant_kernel<f16_tag>), grid, block, 0, s, o, acc, x_scale, w_scale, b, n, acc_row_stride, out_row_stride);
  else
    return MI_ERR_UNSUPPORTED;
  return launch_status();
}

extern "C" int mi_dynamic_int8_quant(int8_t* xq, float* x_scale, const void* x, int32_t num_tokens,
                                     int32_t hidden, int64_t x_token_stride, int dtype, void* stream) {
  using namespace mi;
  if (num_tokens == 0) return MI_OK;
  if (!xq || !x_scale || !x || num_tokens < 0 || hidden <= 0) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto xx = reinterpret_cast<const uint16_t*>(x);
  if (dtype == MI_DTYPE_BF16)
    hipLaunchKernelGGL((dynamic_int8_quant_kernel<bf16_tag>), dim3(num_tokens), dim3(256), 0, s, xq, x_scale, xx, hidden, x_token_stride);
  else if (dtype == MI_DTYPE_F16)
    hipLaunchKernelGGL((dynamic_int8_quant_kernel<f16_tag>), dim3(num_tokens), dim3(256), 0, s, xq, x_scale, xx, hidden, x_token_stride);
  else
    return MI_ERR_UNSUPPORTED;
  return launch_status();
}

extern "C" int mi_int8_gemm_dequant(void* out, const int8_t* xq, const float* x_scale, const int8_t* wq,
                                    const float* w_scale, const void* bias, int32_t m, int32_t n, int32_t k,
                                    int64_t out_row_stride, int dtype, void* stream) {
  using namespace mi;
  if (m == 0 || n == 0) return MI_OK;
  if (!out || !xq || !x_scale || !wq || !w_scale || m < 0 || n < 0 || k <= 0) return MI_ERR_INVALID_ARG;
  if (k % 64) return MI_ERR_UNSUPPORTED;  // one MFMA consumes 64 k; also keeps fragments 16-B aligned
  if (((uintptr_t)xq | (uintptr_t)wq) & 15) return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((n + 63) / 64, (m + 63) / 64), block(256);
  auto o = reinterpret_cast<uint16_t*>(out);
  auto b = reinterpret_cast<const uint16_t*>(bias);
  if (dtype == MI_DTYPE_BF16)
    hipLaunchKernelGGL((int8_gemm_dequant_kernel<bf16_tag>), grid, block, 0, s, o, xq, x_scale, wq, w_scale, b, m, n, k, out_row_stride);
  else if (dtype == MI_DTYPE_F16)
    hipLaunchKernelGGL((int8_gemm_dequant_kernel<f16_tag>), grid, block, 0, s, o, xq, x_scale, wq, w_scale, b, m, n, k, out_row_stride);
  else
    return MI_ERR_UNSUPPORTED;
  return launch_status();
}
