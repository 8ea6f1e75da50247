// N2 — device-side sampler for the talker loop: repetition penalty, temperature, top-k,
// top-p and the random draw in ONE launch, one 256-thread workgroup per logits row held in LDS as fp32.
//
// Replaces the sampler call of the reference's runner, `self.sampler(logits=..., sampling_metadata=...)`
// at vllm_omni/worker/gpu_ar_model_runner.py:626-657 — upstream vLLM v0.19 `Sampler`: penalties
// (`apply_penalties`: logit > 0 ? logit / r : logit * r for tokens seen in prompt or output) -> temperature
// -> `apply_top_k_top_p` (top-k keeps every value >= the k-th largest; top-p keeps an element while the
// probability mass ranked before it is < p, always at least one) -> `random_sample` (probs / Exp(1) noise,
// argmax).  The shipped talker stage samples with T = 0.9, top_p = 0.8, top_k = 40, repetition_penalty =
// 1.05 (vllm_omni/platforms/rocm/stage_configs/qwen2_5_omni.yaml:51-59).
//
// No sort: both cut-offs are found over the order-preserving integer key of the fp32 values — top-k by a radix select
// (four key bits per step, per-wave LDS histograms); top-p by one wave bisecting the key over the survivors' compacted
// lists when a top-k has left few of them, else block-wide four bits at a time — which is exact for top-k
// (ties with the k-th value are kept, as upstream's
// `logits_sort < kth` mask does) and, for top-p, keeps every element tied with the last kept VALUE (upstream
// breaks such ties by sort position, i.e. arbitrarily).  The draw is the exponential race upstream uses:
// id = argmax_i p_i / q_i, q_i = -ln(u_i), with u_i from Philox4x32-10 keyed by the request's seed and
// counted by (column, draw offset) — reproducible, independent of launch shape, restated bit for bit by
// oracle/ref_sampler.py.  Rows with temperature == 0 take the first index of the maximum (greedy).
#include "common.h"

namespace mi {
namespace {

struct SamplerParams {
  int64_t* ids;
  const void* logits;
  int64_t row_stride;
  int32_t vocab;
  const float* temperature;   // [rows]
  const int32_t* top_k;       // [rows] or null; <= 0 or >= vocab: off
  const float* top_p;         // [rows] or null; >= 1: off
  const float* rep_penalty;   // [rows] or null; 1: off
  uint32_t* seen;             // [state rows][words_per_row] bitmap of prompt + output tokens, or null
  const int32_t* seen_row;    // [rows] state row of each logits row, or null (= row)
  int32_t words_per_row;
  int32_t update_seen;
  const uint64_t* seed;       // [rows]
  const uint64_t* offset;     // [rows] draw counter of the request
  // mi_sample_tokens_embed: the drawn id's row of `emb_table` [vocab, emb_dim] goes to emb_a[row] (and emb_b[row]), the id
  // itself also to ids2[row * ids2_stride] — mi_embed_codes behind the draw, in the draw's launch
  const uint16_t* emb_table = nullptr;
  uint16_t *emb_a = nullptr, *emb_b = nullptr;
  int64_t* ids2 = nullptr;
  int64_t a_stride = 0, b_stride = 0, ids2_stride = 0;
  int32_t emb_dim = 0;
};

// order-preserving map fp32 -> uint32 (larger float <=> larger key; -0 and +0 share a key: they compare equal, and a
// top-k whose k-th value is +0 keeps the -0 entries too, as the comparison `logit < kth` upstream does)
__device__ __forceinline__ uint32_t fkey(float f) {
  uint32_t b = __builtin_bit_cast(uint32_t, f);
  if ((b << 1) == 0u) b = 0u;
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

// -log(u) for u = (n + 0.5) 2^-24, n a 24-bit draw.  n + 0.5 is exact in fp32 only below 2^23; above, work from the
// complement w = 1 - u = (2^24 - n - 0.5) 2^-24, which is: a series where w is small (the race is decided by the
// SMALLEST exponentials, i.e. by u next to 1, where log(u) computed from a rounded u is off by tens of percent, and
// n = 2^24 - 1 gave u = 1, -log = 0 and a candidate that could never win), the library log elsewhere.
__device__ __forceinline__ float neg_log_u24(uint32_t n) {
  // one log per draw and no divergent branch: both forms of u are exact in fp32 on their half of the range, the series
  // (four fused multiply-adds) is selected where w < 2^-6
  const bool hi = n >= (1u << 23);
  const float w = ((float)((1u << 24) - n) - 0.5f) * (1.f / 16777216.f);
  const float u = hi ? 1.f - w : ((float)n + 0.5f) * (1.f / 16777216.f);
  const float series = w * (1.f + w * (0.5f + w * ((1.f / 3.f) + w * 0.25f)));
  return (hi && w < 0.015625f) ? series : -__logf(u);
}

// Philox4x32-10 (Salmon et al., SC'11), first output word
__device__ __forceinline__ uint32_t philox_first(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}

struct BlockReduce {
  float* fbuf;     // [4]
  uint32_t* ubuf;  // [4]
  __device__ __forceinline__ float sum(float v) {
    v = wave_allreduce_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) fbuf[threadIdx.x >> 6] = v;
    __syncthreads();
    return (fbuf[0] + fbuf[1]) + (fbuf[2] + fbuf[3]);
  }
  __device__ __forceinline__ float max(float v) {
    v = wave_allreduce_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) fbuf[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(fbuf[0], fbuf[1]), fmaxf(fbuf[2], fbuf[3]));
  }
  // 15 values at once (one step of the 4-bit searches); `mbuf` = [4][16] words of LDS.  Counts arrive per WAVE
  // (ballot + popcount: no cross-lane traffic at all); masses are summed on the VALU — four row rotations (DPP) and the
  // two row swaps — instead of six ds_bpermute round trips per value (15 x 6 of them were 3 us per step).
  __device__ __forceinline__ void count15(uint32_t (&c)[15], uint32_t* mbuf) {   // c[]: this wave's counts (uniform)
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int d = 0; d < 15; ++d) mbuf[(threadIdx.x >> 6) * 16 + d] = c[d];
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 15; ++d) c[d] = mbuf[d] + mbuf[16 + d] + mbuf[32 + d] + mbuf[48 + d];
  }
  template <int N>
  static __device__ __forceinline__ float ror(float v) {   // lane l of a 16-lane row <- lane (l + N) % 16
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
  }
  static __device__ __forceinline__ float wave_sum(float v) {   // lane 0's association is the one that is used
    v += ror<8>(v);
    v += ror<4>(v);
    v += ror<2>(v);
    v += ror<1>(v);
    float a = v, b = v;
    permlane32_swap(a, b);
    a = b = a + b;
    permlane16_swap(a, b);
    return a + b;
  }
  __device__ __forceinline__ void sum15(float (&m)[15], float* mbuf) {
#pragma unroll
    for (int d = 0; d < 15; ++d) m[d] = wave_sum(m[d]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int d = 0; d < 15; ++d) mbuf[(threadIdx.x >> 6) * 16 + d] = m[d];
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 15; ++d) m[d] = (mbuf[d] + mbuf[16 + d]) + (mbuf[32 + d] + mbuf[48 + d]);
  }
  __device__ __forceinline__ uint32_t count(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ubuf[threadIdx.x >> 6] = v;
    __syncthreads();
    return ubuf[0] + ubuf[1] + ubuf[2] + ubuf[3];
  }
};

template <int DT>
__global__ __launch_bounds__(256) void sample_kernel(const SamplerParams p) {
  scalar_cache_acquire();
  extern __shared__ float vals[];  // [vocab] then 4 floats + 4 uints + 2 ints (+ pad) + [4][16] words + survivor lists
  float* fbuf = vals + p.vocab;
  uint32_t* ubuf = reinterpret_cast<uint32_t*>(fbuf + 4);
  int* ibuf = reinterpret_cast<int*>(ubuf + 4);
  float* mbuf = fbuf + 16;
  BlockReduce red{fbuf, ubuf};
  const int row = blockIdx.x, tid = threadIdx.x, V = p.vocab;
  const float T = p.temperature[row];
  const float pen = p.rep_penalty ? p.rep_penalty[row] : 1.f;
  const int srow = p.seen_row ? p.seen_row[row] : row;
  uint32_t* seen = p.seen ? p.seen + (int64_t)srow * p.words_per_row : nullptr;
  const float invT = T > 0.f ? 1.f / T : 1.f;

  // 1. load -> penalty -> temperature, fp32 in LDS
  for (int i = tid; i < V; i += 256) {
    float v;
    if constexpr (DT == MI_DTYPE_F32) v = reinterpret_cast<const float*>(p.logits)[row * p.row_stride + i];
    else if constexpr (DT == MI_DTYPE_BF16)
      v = Elem<bf16_tag>::from_bits(reinterpret_cast<const uint16_t*>(p.logits)[row * p.row_stride + i]);
    else v = Elem<f16_tag>::from_bits(reinterpret_cast<const uint16_t*>(p.logits)[row * p.row_stride + i]);
    if (seen && pen != 1.f && ((seen[i >> 5] >> (i & 31)) & 1u)) v = v > 0.f ? v / pen : v * pen;
    vals[i] = v * invT;
  }
  __syncthreads();

  int best_i = 0x7fffffff;
  float best_s = -INFINITY;
  if (T > 0.f) {
    // 2. top-k: key of the k-th largest value, built bit by bit (largest tau with count(key >= tau) >= k)
    const int k = p.top_k ? p.top_k[row] : 0;
    if (k > 0 && k < V) {
      // The k-th largest key `tau` (values with a smaller key are dropped; ties with it stay, as `logit < kth` upstream).
      // Fast path — one pass instead of eight: bucket the values by a MONOTONE map of the value itself, 256 equal steps
      // between the row's smallest and largest finite logit (monotone: a value in a higher bucket is a larger value, so the
      // k-th largest lies in the bucket where the suffix count crosses k), then select exactly among that bucket's few
      // members by rank (<= 64 of them: one wave, every member against every other).  Rows the map cannot order — a NaN
      // or +inf logit, all finite logits equal — or a boundary bucket with more than 64 members take the radix select below:
      // the same tau either way.  Measured, 2048 logits, k = 50: 13.4 -> 5.4 us of the launch (profiles/r04y_sampler_bucket_select.log).
      uint32_t tau = 0, kk = (uint32_t)k;
      bool have_tau = false;
      {
        float hi_v = -INFINITY, lo_v = INFINITY;
        for (int i = tid; i < V; i += 256) {
          const float v = vals[i];
          if (!(v < INFINITY)) hi_v = INFINITY;          // +inf or NaN: no bucket map
          else if (v > -INFINITY) { hi_v = fmaxf(hi_v, v); lo_v = fminf(lo_v, v); }
        }
        hi_v = red.max(hi_v);
        lo_v = -red.max(-lo_v);
        const float range = hi_v - lo_v;
        if (range > 0.f && range < INFINITY) {           // block-uniform (reduced values)
          const float sc = 255.f / range;
          auto bucket = [&](float v) -> int {
            if (!(v > -INFINITY)) return 0;
            const int b = (int)((v - lo_v) * sc);
            return b > 255 ? 255 : b;
          };
          uint32_t* hist256 = reinterpret_cast<uint32_t*>(mbuf + 64);   // the survivor lists' space: not in use yet
          uint32_t* cand = hist256 + 256;                                // [64] keys of the boundary bucket
          uint32_t* misc = cand + 64;                                    // [0..3] wave totals, [4] bucket, [5] above, [6] count, [7] tau
          hist256[tid] = 0u;
          if (tid == 0) misc[6] = 0u;
          __syncthreads();
          for (int i = tid; i < V; i += 256) atomicAdd(&hist256[bucket(vals[i])], 1u);
          __syncthreads();
          // suffix counts: s = members of buckets tid .. 255
          const uint32_t c = hist256[tid];
          uint32_t sfx = c;
          const int ln = tid & 63;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_down(sfx, o, 64);
            if (ln + o < 64) sfx += up;
          }
          if (ln == 0) misc[tid >> 6] = sfx;
          __syncthreads();
          for (int w2 = (tid >> 6) + 1; w2 < 4; ++w2) sfx += misc[w2];
          if (sfx >= kk && sfx - c < kk) {   // exactly one bucket: the suffix count is non-increasing and reaches V >= k
            misc[4] = (uint32_t)tid;
            misc[5] = sfx - c;
          }
          __syncthreads();
          const int bsel = (int)misc[4];
          const uint32_t need = kk - misc[5];                   // the need-th largest of the bucket's members
          const uint32_t nb = hist256[bsel];
          if (nb <= 64u) {                                        // block-uniform
            for (int i = tid; i < V; i += 256) {
              const float v = vals[i];
              if (bucket(v) == bsel) cand[atomicAdd(&misc[6], 1u)] = fkey(v);
            }
            __syncthreads();
            if (tid < 64) {
              const uint32_t key = (uint32_t)tid < nb ? cand[tid] : 0u;
              uint32_t gt = 0, ge = 0;
              for (uint32_t j = 0; j < nb; ++j) {
                const uint32_t kj = cand[j];                      // same address in every lane: a broadcast read
                gt += kj > key ? 1u : 0u;
                ge += kj >= key ? 1u : 0u;
              }
              if ((uint32_t)tid < nb && gt < need && need <= ge) misc[7] = key;   // equal keys write the same value
            }
            __syncthreads();
            tau = misc[7];
            have_tau = true;
            __syncthreads();   // misc / cand are the top-p stage's lists next
          }
        }
      }
      if (!have_tau) {
      // radix select, four key bits per step: a 16-bin histogram (one per wave: LDS atomics) of the keys that share the
      // digits found so far; the digit is the largest d whose bins d..15 hold the kk-th largest of them
      uint32_t* hist = reinterpret_cast<uint32_t*>(mbuf);   // [4][16]
      for (int sh = 28; sh >= 0; sh -= 4) {
        if (tid < 64) hist[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < V; i += 256) {
          const uint32_t key = fkey(vals[i]);
          if (sh == 28 || (key >> (sh + 4)) == (tau >> (sh + 4))) atomicAdd(&hist[(tid >> 6) * 16 + ((key >> sh) & 15u)], 1u);
        }
        __syncthreads();
        uint32_t above = 0, dig = 0;
#pragma unroll
        for (int d = 15; d >= 0; --d) {
          const uint32_t c = hist[d] + hist[16 + d] + hist[32 + d] + hist[48 + d];
          if (above + c >= kk) {
            dig = (uint32_t)d;
            break;
          }
          above += c;
        }
        kk -= above;
        tau |= dig << sh;
        __syncthreads();   // every thread has read the bins before they are zeroed again
      }
      }
      for (int i = tid; i < V; i += 256)
        if (fkey(vals[i]) < tau) vals[i] = -INFINITY;
      __syncthreads();
    }
    // 3. softmax (probabilities replace the logits in LDS)
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, vals[i]);
    mx = red.max(mx);
    float s = 0.f;
    for (int i = tid; i < V; i += 256) {
      const float e = __expf(vals[i] - mx);
      vals[i] = e;
      s += e;
    }
    s = red.sum(s);
    const float inv = 1.f / s;
    // 4. top-p: smallest key c with mass(prob key > c) < top_p; keep prob keys >= c (binary search)
    const float tp = p.top_p ? p.top_p[row] : 1.f;
    uint32_t cut = 0;
    if (tp < 1.f) {
      // After a top-k the survivors are few: each wave compacts its own (ballot + prefix count, in index order) into a
      // 64-entry list, and ONE wave bisects the key over those lists — a compare, a DPP / row-swap sum and a scalar
      // broadcast per step, no barrier.  More than 64 survivors in some wave (no top-k, or many ties): the block-wide
      // search below.
      float* cp = mbuf + 64;                                     // [4][64] probabilities
      uint32_t* ck = reinterpret_cast<uint32_t*>(cp + 256);      // [4][64] their keys
      uint32_t* cnt = ck + 256;                                  // [4] survivors per wave
      const int lane = tid & 63, w = tid >> 6;
      uint32_t woff = 0;
      for (int i0 = 0; i0 < V; i0 += 256) {
        const int i = i0 + tid;
        const float pr = i < V ? vals[i] * inv : 0.f;
        const bool on = pr > 0.f;
        const unsigned long long m = __ballot(on);
        if (on) {
          const uint32_t pos = woff + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          if (pos < 64u) {
            cp[w * 64 + pos] = pr;
            ck[w * 64 + pos] = fkey(pr);
          }
        }
        woff += (uint32_t)__popcll(m);
      }
      if (lane == 0) cnt[w] = woff;
      __syncthreads();
      const bool small = cnt[0] <= 64u && cnt[1] <= 64u && cnt[2] <= 64u && cnt[3] <= 64u;
      if (small) {
        if (w == 0) {
          float pj[4];
          uint32_t kj[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool have = (uint32_t)lane < cnt[j];
            pj[j] = have ? cp[j * 64 + lane] : 0.f;
            kj[j] = have ? ck[j * 64 + lane] : 0u;
          }
          uint32_t lo = 0, hi = 0xffffffffu;   // the smallest key c with mass(key > c) < top_p
          while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            float mm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) mm += kj[j] > mid ? pj[j] : 0.f;
            mm = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, BlockReduce::wave_sum(mm))));
            if (mm < tp) hi = mid; else lo = mid + 1;
          }
          if (lane == 0) ubuf[0] = lo;
        }
        __syncthreads();
        cut = ubuf[0];
        __syncthreads();
      } else {
      // the smallest key c with mass(key > c) < top_p, digit by digit: the smallest d whose bucket's LAST key
      // (prefix | d << sh | ones below) satisfies it; d = 15 when none of 0..14 does (the mass above the largest key
      // is 0 < top_p).  The predicate is monotone in c — the masses are sums of non-negative terms in a fixed
      // association — so this lands on the same key as a bit-by-bit bisection.
      uint32_t pre = 0;
      for (int sh = 28; sh >= 0; sh -= 4) {
        const uint32_t below = sh ? ((1u << sh) - 1u) : 0u;
        float m[15];
#pragma unroll
        for (int d = 0; d < 15; ++d) m[d] = 0.f;
        for (int i = tid; i < V; i += 256) {
          const float pr = vals[i] * inv;
          const uint32_t key = fkey(pr);
#pragma unroll
          for (int d = 0; d < 15; ++d)
            if (key > (pre | ((uint32_t)d << sh) | below)) m[d] += pr;
        }
        red.sum15(m, mbuf);
        uint32_t dig = 15;
#pragma unroll
        for (int d = 14; d >= 0; --d)
          if (m[d] < tp) dig = d;
        pre |= dig << sh;
      }
      cut = pre;
      }
    }
    // 5. exponential race over the survivors
    const uint64_t sd = p.seed[row], off = p.offset[row];
    for (int i = tid; i < V; i += 256) {
      const float pr = vals[i] * inv;
      if (pr > 0.f && fkey(pr) >= cut) {
        const uint32_t x = philox_first((uint32_t)i, 0u, (uint32_t)off, (uint32_t)(off >> 32), (uint32_t)sd,
                                        (uint32_t)(sd >> 32));
        const float sc = pr / neg_log_u24(x >> 8);   // u in (0, 1), 24 bits
        if (sc > best_s || (sc == best_s && i < best_i)) { best_s = sc; best_i = i; }
      }
    }
  } else {
    for (int i = tid; i < V; i += 256) {
      const float v = vals[i];
      if (v > best_s || (v == best_s && i < best_i)) { best_s = v; best_i = i; }
    }
  }
  // block argmax, first index on ties
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float os = __shfl_xor(best_s, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (os > best_s || (os == best_s && oi < best_i)) { best_s = os; best_i = oi; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { fbuf[tid >> 6] = best_s; ibuf[tid >> 6] = best_i; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (fbuf[w] > best_s || (fbuf[w] == best_s && ibuf[w] < best_i)) { best_s = fbuf[w]; best_i = ibuf[w]; }
    if (best_i == 0x7fffffff) best_i = 0;  // a row of NaNs
    p.ids[row] = best_i;
    if (seen && p.update_seen) atomicOr(&seen[best_i >> 5], 1u << (best_i & 31));
    if (p.emb_table) {
      if (p.ids2) p.ids2[row * p.ids2_stride] = best_i;
      ibuf[0] = best_i;
    }
  }
  if (p.emb_table) {   // block-uniform
    __syncthreads();
    const uint16_t* src = p.emb_table + (int64_t)ibuf[0] * p.emb_dim;
    for (int ch = tid; ch < p.emb_dim / 8; ch += 256) {
      const uint4 a = *reinterpret_cast<const uint4*>(src + ch * 8);
      *reinterpret_cast<uint4*>(p.emb_a + row * p.a_stride + ch * 8) = a;
      if (p.emb_b) *reinterpret_cast<uint4*>(p.emb_b + row * p.b_stride + ch * 8) = a;
    }
  }
}

}  // namespace
}  // namespace mi

namespace {
int sample_launch(mi::SamplerParams p, int32_t num_rows, int logits_dtype, void* stream) {
  using namespace mi;
  const size_t lds = (size_t)p.vocab * 4 + 64 + 256 + 2048 + 16;   // row, scalars, [4][16] words, survivor lists + counts
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(num_rows), block(256);
  switch (logits_dtype) {
    case MI_DTYPE_F32: hipLaunchKernelGGL((sample_kernel<MI_DTYPE_F32>), grid, block, lds, s, p); break;
    case MI_DTYPE_BF16: hipLaunchKernelGGL((sample_kernel<MI_DTYPE_BF16>), grid, block, lds, s, p); break;
    case MI_DTYPE_F16: hipLaunchKernelGGL((sample_kernel<MI_DTYPE_F16>), grid, block, lds, s, p); break;
    default: return MI_ERR_UNSUPPORTED;
  }
  return launch_status();
}
}  // namespace

extern "C" int mi_sample_tokens(int64_t* ids, const void* logits, int32_t num_rows, int32_t vocab,
                                int64_t row_stride, int logits_dtype, const float* temperature,
                                const int32_t* top_k, const float* top_p, const float* rep_penalty,
                                uint32_t* seen_bitmap, const int32_t* seen_row, int32_t bitmap_words_per_row,
                                int32_t update_seen, const uint64_t* seed, const uint64_t* offset, void* stream) {
  using namespace mi;
  if (num_rows == 0) return MI_OK;
  if (!ids || !logits || !temperature || !seed || !offset || num_rows < 0 || vocab <= 0 || row_stride < vocab)
    return MI_ERR_INVALID_ARG;
  if (vocab > 15744) return MI_ERR_UNSUPPORTED;  // the fp32 row lives in 64 KB of LDS (talker vocabularies: 3072 / 8448)
  if (seen_bitmap && bitmap_words_per_row * 32 < vocab) return MI_ERR_INVALID_ARG;
  SamplerParams p{ids, logits, row_stride, vocab, temperature, top_k, top_p, rep_penalty, seen_bitmap, seen_row,
                  bitmap_words_per_row, update_seen, seed, offset};
  return sample_launch(p, num_rows, logits_dtype, stream);
}

extern "C" int mi_sample_tokens_embed(int64_t* ids, const void* logits, int32_t num_rows, int32_t vocab, int64_t row_stride,
                                      int logits_dtype, const float* temperature, const int32_t* top_k, const float* top_p,
                                      const uint64_t* seed, const uint64_t* offset, const void* embed_table,
                                      int32_t embed_dim, void* out_a, int64_t out_a_token_stride, void* out_b,
                                      int64_t out_b_token_stride, int64_t* ids_out, int64_t ids_out_stride, void* stream) {
  using namespace mi;
  if (num_rows == 0) return MI_OK;
  if (!ids || !logits || !temperature || !seed || !offset || num_rows < 0 || vocab <= 0 || row_stride < vocab || !embed_table ||
      !out_a || embed_dim <= 0)
    return MI_ERR_INVALID_ARG;
  if (vocab > 15744 || embed_dim % 8 || out_a_token_stride % 8 || (out_b && out_b_token_stride % 8)) return MI_ERR_UNSUPPORTED;
  if (((uintptr_t)out_a | (uintptr_t)out_b | (uintptr_t)embed_table) & 15) return MI_ERR_INVALID_ARG;
  SamplerParams p{ids, logits, row_stride, vocab, temperature, top_k, top_p, nullptr, nullptr, nullptr, 0, 0, seed, offset};
  p.emb_table = reinterpret_cast<const uint16_t*>(embed_table);
  p.emb_a = reinterpret_cast<uint16_t*>(out_a);
  p.emb_b = reinterpret_cast<uint16_t*>(out_b);
  p.ids2 = ids_out;
  p.a_stride = out_a_token_stride;
  p.b_stride = out_b_token_stride;
  p.ids2_stride = ids_out_stride;
  p.emb_dim = embed_dim;
  return sample_launch(p, num_rows, logits_dtype, stream);
}
