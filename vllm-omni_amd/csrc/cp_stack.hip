// One decode pass of the Qwen3-Omni code predictor's whole layer stack in ONE launch.
//
// The code predictor (vllm_omni/model_executor/models/common/qwen3_code_predictor.py:143-191 the layer,
// 318-626 the G - 1 = 31 passes of a call) is five small decoder layers (hidden 1024, 16 / 8 heads of 128, MLP 3072) run
// 30 times per talker token on ONE new position each.  As separate launches a layer-step is seven kernels of 4.6-15 us,
// each mostly launch floor (profiles/r04q_cp_b32_timeline.md: 44 us per layer-step at 32 rows for ~10 us of work).
// Here the seven stages of every layer are phases of one persistent grid — one workgroup per CU — separated by a
// two-level grid barrier (2.1 us measured against the ~4.6 us of a kernel boundary: tools/probes/probe_grid_barrier.py,
// profiles/r04z_grid_barrier_probe.log):
//   N1  residual add + input RMSNorm      (a wave per row; consumes the previous layer's down_proj partials)
//   G1  qkv_proj                          (the decode-batch linear tile of skinny_tile.h, fp32 split-K slabs out)
//   A   q/k-norm + RoPE + KV append + attention over the sequence's one KV block (block_attn.hip's wave, slabs in)
//   G2  o_proj                            (slabs out)
//   N2  residual add + post-attention RMSNorm (slabs in)
//   G3  gate_up_proj + SiLU * up          (gated tile, 16-bit out)
//   G4  down_proj                         (slabs out)
// and the final RMSNorm after the last layer.  Nothing a later phase reads may sit in another XCD's L2 or this CU's L1:
// every activation is stored write-through and loaded at device scope (sc1), the x rows of the linear tiles included
// (LDS-DMA with sc1); weights and the KV cache of earlier passes are read-only inside the launch and take the normal path.
// Arithmetic: each phase is the arithmetic of the stand-alone kernel it replaces (same tile code, same slab order, same
// rounding points), so the launch is bit-identical to the chain of launches with the same tile plans
// (tests/test_gpu_cp_stack.py).
#include "skinny_tile.h"

namespace mi {
namespace {

constexpr int kCpMaxLayers = 8;
constexpr int kCpHD = 128;

struct CpStackParams {
  int32_t B, H, NQ, NKV, I, L;     // rows (sequences), hidden, q / kv heads (of 128), MLP intermediate, layers
  int32_t pos;                     // position of the new token (the same in every row): tokens 0 .. pos - 1 are in the block
  int32_t S1, S2, S4;              // split-K slices of qkv_proj, o_proj, down_proj
  float eps, scale_log2e;
  const uint16_t *w_qkv[kCpMaxLayers], *w_o[kCpMaxLayers], *w_gu[kCpMaxLayers], *w_down[kCpMaxLayers];
  const uint16_t *ln1[kCpMaxLayers], *ln2[kCpMaxLayers], *qn[kCpMaxLayers], *kn[kCpMaxLayers];
  const uint16_t* final_norm;
  uint16_t *kc[kCpMaxLayers], *vc[kCpMaxLayers];
  int64_t c_blk, c_tok, c_head;
  const int32_t* block_ids;
  int64_t bt_stride;
  int32_t block_size, max_position;
  const float* cos_sin;
  const uint16_t* x_in;            // [B, H] the step's input rows
  uint16_t *resid, *xn, *attn, *act, *out;   // [B, H] x2, [B, NQ * 128], [B, I], [B, H]
  float *qkv_part, *o_part, *down_part;      // [S1][B][(NQ + 2 NKV) * 128], [S2][B][H], [S4][B][H]
  uint32_t* sync;                  // barrier state (kCpSyncWords words, zero once; never reset)
  uint32_t* err;                   // set when a barrier gave up waiting
};

// ---------------------------------------------------------------------------------------------- grid barrier
// Two levels, agent-scope operations (tools/probes/grid_barrier.hip measured 2.1 us for 256 workgroups against 3.9 us for
// one counter).  Workgroups of XCD x = blockIdx % 8 (round-robin dispatch; the choice only groups arrivals, nothing
// depends on where a workgroup really runs) meet on counter x; the last of them adds to the device counter, waits for all
// eight and releases its group through flag x.  Counters only grow (generation g: group counter = per * g, device
// counter = groups * g, flag = g), so a launch starts from the generation the previous one left in word 0 of the state.
// Waits are bounded: a grid that cannot be co-resident (fewer free CUs than workgroups) ends with *err != 0 instead of hanging.
constexpr int kCpSyncWords = 32 * 20;
constexpr uint32_t kCpSpinLimit = 1u << 22;

__device__ __forceinline__ void cp_grid_barrier(uint32_t* mem, uint32_t gen, uint32_t* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have reached the coherence point
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t groups = gridDim.x < 8u ? gridDim.x : 8u;
    const uint32_t x = blockIdx.x % groups, per = (gridDim.x + groups - 1u - x) / groups;
    uint32_t spins = 0;
    if (__hip_atomic_fetch_add(mem + 32 * (2 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen * per - 1u) {
      __hip_atomic_fetch_add(mem + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((int32_t)(__hip_atomic_load(mem + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - groups * gen) < 0) {
        if (++spins > kCpSpinLimit) { *err = 1u; break; }
      }
      __hip_atomic_store(mem + 32 * (10 + x), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while ((int32_t)(__hip_atomic_load(mem + 32 * (10 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - gen) < 0) {
        if (++spins > kCpSpinLimit) { *err = 2u; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- device-scope I/O
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cp_rsrc(const void* p, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4_t ld128(__amdgpu_buffer_rsrc_t r, int64_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, kAuxAgent);
}
__device__ __forceinline__ uint32_t ld32(__amdgpu_buffer_rsrc_t r, int64_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, kAuxAgent);
}
__device__ __forceinline__ void st128(__amdgpu_buffer_rsrc_t r, int64_t byte_off, u32x4_t v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, kAuxAgent);
}
__device__ __forceinline__ void st32(__amdgpu_buffer_rsrc_t r, int64_t byte_off, uint32_t v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)byte_off, 0, kAuxAgent);
}

// ---------------------------------------------------------------------------------------------- RMSNorm phase
// A wave per row.  x = the step's input (FIRST: residual <- x, no add) or rnd16(sum of the S slabs in slice order)
// (mi_linear_reduce's arithmetic); residual <- rnd16(x + residual); out = rnd16(rnd16(residual * inv) * w) — the arithmetic
// of rms_norm_kernel / rms_norm_splitk_kernel (norm_act.hip), including the order of the sum of squares: lane l holds the
// 8-column chunks l and l + 64, whose wave sums are the two wave sums of the 256-thread kernel, added in its order.
// `out_plain`: the final norm's rows are read by the NEXT launch: a normal store will do.
template <typename Tag>
__device__ __forceinline__ void cp_norm_phase(const CpStackParams& p, const uint16_t* x_in, const float* part, int S,
                                              const uint16_t* w, uint16_t* out, int nwaves, int wave_id) {
  using E = Elem<Tag>;
  const int lane = threadIdx.x & 63;
  const int H = p.H, nchunk = H / 8;     // H <= 1024: two chunks per lane
  const __amdgpu_buffer_rsrc_t r_part = cp_rsrc(part, part ? (int64_t)S * p.B * H * 4 : 0);
  const __amdgpu_buffer_rsrc_t r_res = cp_rsrc(p.resid, (int64_t)p.B * H * 2);
  const __amdgpu_buffer_rsrc_t r_out = cp_rsrc(out, (int64_t)p.B * H * 2);
  for (int row = wave_id; row < p.B; row += nwaves) {
    u32x4_t z[2];
    uint4 wreg[2];
    float ss[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ch = lane + 64 * k;
      z[k] = u32x4_t{0, 0, 0, 0};
      if (ch < nchunk) {
        wreg[k] = *reinterpret_cast<const uint4*>(w + ch * 8);
        u32x4_t a;
        if (x_in) {
          a = __builtin_bit_cast(u32x4_t, *reinterpret_cast<const uint4*>(x_in + (int64_t)row * H + ch * 8));
        } else {
          const int64_t off = ((int64_t)row * H + ch * 8) * 4;
          f32x4_t vl[16], vh[16];
#pragma unroll
          for (int s = 0; s < 16; ++s)
            if (s < S) {
              vl[s] = __builtin_bit_cast(f32x4_t, ld128(r_part, off + (int64_t)s * p.B * H * 4));
              vh[s] = __builtin_bit_cast(f32x4_t, ld128(r_part, off + (int64_t)s * p.B * H * 4 + 16));
            }
          const u32x4_t b = ld128(r_res, ((int64_t)row * H + ch * 8) * 2);
          f32x4_t lo = vl[0], hi = vh[0];
#pragma unroll
          for (int s = 1; s < 16; ++s)
            if (s < S) {
              lo += vl[s];
              hi += vh[s];
            }
          a = u32x4_t{E::pack(lo[0], lo[1]), E::pack(lo[2], lo[3]), E::pack(hi[0], hi[1]), E::pack(hi[2], hi[3])};
#pragma unroll
          for (int i = 0; i < 4; ++i) a[i] = E::pack(E::lo(a[i]) + E::lo(b[i]), E::hi(a[i]) + E::hi(b[i]));
        }
        st128(r_res, ((int64_t)row * H + ch * 8) * 2, a);
        z[k] = a;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float l = E::lo(a[i]), h = E::hi(a[i]);
          ss[k] += l * l + h * h;
        }
      }
    }
    float tot = 0.f;                       // the 256-thread kernel: r = 0; r += wave sums in wave order (waves 2, 3 hold nothing)
    tot += wave_allreduce_sum(ss[0]);
    tot += wave_allreduce_sum(ss[1]);
    tot += 0.f;
    tot += 0.f;
    const float inv = __builtin_amdgcn_rsqf(tot / (float)H + p.eps);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ch = lane + 64 * k;
      if (ch < nchunk) {
        const uint32_t ws[4] = {wreg[k].x, wreg[k].y, wreg[k].z, wreg[k].w};
        u32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float nlo = E::from_bits(E::to_bits(E::lo(z[k][i]) * inv));
          const float nhi = E::from_bits(E::to_bits(E::hi(z[k][i]) * inv));
          o[i] = E::pack(nlo * E::lo(ws[i]), nhi * E::hi(ws[i]));
        }
        st128(r_out, ((int64_t)row * H + ch * 8) * 2, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- attention phase
// block_attn.hip's wave (one per (row, KV head)), fed by qkv_proj's slabs and looping over the pairs of this grid.
__device__ __forceinline__ float cp_wave_sum(float v) { return rows_allreduce_sum(group_allreduce_sum<16>(v)); }

template <typename Tag, int G, int TILES>
__device__ __forceinline__ void cp_attn_phase(const CpStackParams& p, int layer, uint16_t* lq /* this wave's G * 128 */,
                                              int nwaves, int wave_id) {
  using E = Elem<Tag>;
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, kg = lane >> 4;
  const int pos = p.pos;                                  // host: 0 <= pos <= min(16 TILES, block_size - 1)
  const int rpos = pos < p.max_position ? pos : p.max_position - 1;
  const int tiles = (pos + 15) >> 4;
  const int64_t row_len = (int64_t)(p.NQ + 2 * p.NKV) * kCpHD;
  const __amdgpu_buffer_rsrc_t r_part = cp_rsrc(p.qkv_part, (int64_t)p.S1 * p.B * row_len * 4);
  const __amdgpu_buffer_rsrc_t r_out = cp_rsrc(p.attn, (int64_t)p.B * p.NQ * kCpHD * 2);
  const float cs = p.cos_sin[(int64_t)rpos * kCpHD + lane], sn = p.cos_sin[(int64_t)rpos * kCpHD + 64 + lane];
  const uint16_t *qw = p.qn[layer], *kw = p.kn[layer];
  for (int pair = wave_id; pair < p.B * p.NKV; pair += nwaves) {
    const int t = pair / p.NKV, kvh = pair - t * p.NKV;
    const int64_t blk = p.block_ids[(int64_t)t * p.bt_stride];
    uint16_t* kblk = p.kc[layer] + blk * p.c_blk + (int64_t)kvh * p.c_head;
    uint16_t* vblk = p.vc[layer] + blk * p.c_blk + (int64_t)kvh * p.c_head;
    uint4 ka[TILES][4];
#pragma unroll
    for (int tile = 0; tile < TILES; ++tile) {
      int tok = 16 * tile + r16;
      tok = tok < p.block_size ? tok : p.block_size - 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        ka[tile][ks] = make_uint4(0, 0, 0, 0);
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
    auto norm_rope = [&](float x, float y, const uint16_t* wgt, float& ox, float& oy) {
      const float ss = cp_wave_sum(x * x + y * y);
      const float inv = __builtin_amdgcn_rsqf(ss / (float)kCpHD + p.eps);
      x = E::from_bits(E::to_bits(E::from_bits(E::to_bits(x * inv)) * E::from_bits(wgt[lane])));
      y = E::from_bits(E::to_bits(E::from_bits(E::to_bits(y * inv)) * E::from_bits(wgt[64 + lane])));
      ox = E::from_bits(E::to_bits(x * cs - y * sn));
      oy = E::from_bits(E::to_bits(y * cs + x * sn));
    };
    auto from_part = [&](int col) -> float {   // rnd16(sum of the slabs in slice order): the value mi_linear_reduce stores
      const int64_t off = ((int64_t)t * row_len + col) * 4;
      float v[16];
#pragma unroll
      for (int sl = 0; sl < 16; ++sl)
        if (sl < p.S1) v[sl] = __builtin_bit_cast(float, ld32(r_part, off + (int64_t)sl * p.B * row_len * 4));
      float a = v[0];
#pragma unroll
      for (int sl = 1; sl < 16; ++sl)
        if (sl < p.S1) a += v[sl];
      return E::from_bits(E::to_bits(a));
    };
    float kx, ky;
    {
      const int col = (p.NQ + kvh) * kCpHD;
      norm_rope(from_part(col + lane), from_part(col + 64 + lane), kw, kx, ky);
    }
    uint32_t vnew;
    {
      const int col = (p.NQ + p.NKV + kvh) * kCpHD + 2 * lane;
      vnew = E::pack(from_part(col), from_part(col + 1));
    }
    {
      uint16_t* kd = kblk + (int64_t)pos * p.c_tok;
      kd[lane] = E::to_bits(kx);
      kd[64 + lane] = E::to_bits(ky);
      *reinterpret_cast<uint32_t*>(vblk + (int64_t)pos * p.c_tok + 2 * lane) = vnew;
    }
    float s_new[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float qx, qy;
      const int col = (kvh * G + g) * kCpHD;
      norm_rope(from_part(col + lane), from_part(col + 64 + lane), qw, qx, qy);
      lq[g * kCpHD + lane] = E::to_bits(qx);
      lq[g * kCpHD + 64 + lane] = E::to_bits(qy);
      s_new[g] = cp_wave_sum(qx * kx + qy * ky);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS rows: written above, read below
    __builtin_amdgcn_wave_barrier();

    f32x4_t acc[TILES];
#pragma unroll
    for (int tile = 0; tile < TILES; ++tile) acc[tile] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 qb = make_uint4(0, 0, 0, 0);
      if (r16 < G) qb = *reinterpret_cast<const uint4*>(&lq[r16 * kCpHD + 32 * ks + 8 * kg]);
#pragma unroll
      for (int tile = 0; tile < TILES; ++tile)
        if (tile < tiles) acc[tile] = mfma16<Tag>(ka[tile][ks], qb, acc[tile]);
    }
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
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float ig = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), g));
      st32(r_out, ((int64_t)t * p.NQ * kCpHD + (kvh * G + g) * kCpHD + 2 * lane) * 2, E::pack(o0[g] * ig, o1[g] * ig));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // lq is rewritten by the next pair
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------- linear phases
// NSLAB / NL fix the workgroup: NSLAB consumer waves + NL loader waves, 64-row blocks (MT = 4).
template <typename Tag, int NSLAB, int NL, int NST, bool GATED>
__device__ __forceinline__ void cp_linear_phase(const uint16_t* x, const uint16_t* w, float* part, uint16_t* out, int M, int N,
                                                int K, int S, char* smem) {
  SkinnyParams q;
  q.x = x;
  q.w = w;
  q.bias = nullptr;
  q.out = out;
  q.part = part;
  q.xs = nullptr;
  q.ws = nullptr;
  q.M = M;
  q.N = N;
  q.K = K;
  q.S = GATED ? 1 : S;
  q.RB = (M + 63) / 64;
  const int tiles = GATED ? (N + 16 * NSLAB - 1) / (16 * NSLAB) : (N + 32 * NSLAB - 1) / (32 * NSLAB);
  q.U = tiles * q.S;
  q.cnt = nullptr;
  q.row_map = nullptr;
  q.tile_expert = nullptr;
  q.num_tiles = nullptr;
  q.rpt = 0;
  q.gather_div = 0;
  q.w_expert_stride = 0;
  q.dbg = 0;
  q.ldx = K;
  q.ldo = N;
  const int items = (q.U + 7) / 8 * 8 * q.RB;
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    // S == 1 with slabs out: the tile stores 16-bit rows when S == 1, so a one-slice plan is run as the S > 1 form of one slice
    skinny_tile<Tag, Tag, NSLAB, 4, 1, NL, GATED, false, NST, true>(q, item, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the ring is rewritten by the next item's loaders
  }
}

constexpr int cp_ring_stages(int nslab) {
  const int stage = 4 * 2048 + nslab * 4096;
  const int n = 144 * 1024 / stage;
  return n < 3 ? 3 : (n > 8 ? 8 : n);
}

#ifdef MI_CP_TIMING
// debug build (make dbg-style, tools/debug_cp_stack_timing.py): 10 ns timestamps of workgroup MI_CP_TIMING_WG around every
// phase and barrier of the first layer-steps
__device__ unsigned long long mi_cp_ticks[128];
#define CP_TICK()                                                                                    \
  do {                                                                                               \
    if (blockIdx.x == (MI_CP_TIMING_WG) && threadIdx.x == 0 && tick < 128) mi_cp_ticks[tick++] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CP_TICK() do {} while (0)
#endif

template <typename Tag, int NSLAB, int NL, int G, int TILES>
__global__ __launch_bounds__(64 * (NSLAB + NL)) void cp_stack_kernel(const CpStackParams p) {
#ifdef MI_CP_TIMING
  int tick = 0;
#endif
  scalar_cache_acquire();
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int NW = NSLAB + NL;
  constexpr int NST = cp_ring_stages(NSLAB);
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // waves of the grid, workgroup-minor: consecutive rows / pairs land on different CUs
  const int nwaves = NW * gridDim.x, wave_id = wid * gridDim.x + blockIdx.x;
  uint16_t* lq = reinterpret_cast<uint16_t*>(smem) + wid * (G * kCpHD);
  uint32_t gen = __hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left by the previous launch
  const int QKV = (p.NQ + 2 * p.NKV) * kCpHD;
  CP_TICK();
  for (int l = 0; l < p.L; ++l) {
    if (l == 0) cp_norm_phase<Tag>(p, p.x_in, nullptr, 0, p.ln1[0], p.xn, nwaves, wave_id);
    else cp_norm_phase<Tag>(p, nullptr, p.down_part, p.S4, p.ln1[l], p.xn, nwaves, wave_id);
    CP_TICK();
    cp_grid_barrier(p.sync, ++gen, p.err);
    CP_TICK();
    cp_linear_phase<Tag, NSLAB, NL, NST, false>(p.xn, p.w_qkv[l], p.qkv_part, nullptr, p.B, QKV, p.H, p.S1, smem);
    CP_TICK();
    cp_grid_barrier(p.sync, ++gen, p.err);
    CP_TICK();
    cp_attn_phase<Tag, G, TILES>(p, l, lq, nwaves, wave_id);
    CP_TICK();
    cp_grid_barrier(p.sync, ++gen, p.err);
    CP_TICK();
    cp_linear_phase<Tag, NSLAB, NL, NST, false>(p.attn, p.w_o[l], p.o_part, nullptr, p.B, p.H, p.NQ * kCpHD, p.S2, smem);
    CP_TICK();
    cp_grid_barrier(p.sync, ++gen, p.err);
    CP_TICK();
    cp_norm_phase<Tag>(p, nullptr, p.o_part, p.S2, p.ln2[l], p.xn, nwaves, wave_id);
    CP_TICK();
    cp_grid_barrier(p.sync, ++gen, p.err);
    CP_TICK();
    cp_linear_phase<Tag, NSLAB, NL, NST, true>(p.xn, p.w_gu[l], nullptr, p.act, p.B, p.I, p.H, 1, smem);
    CP_TICK();
    cp_grid_barrier(p.sync, ++gen, p.err);
    CP_TICK();
    cp_linear_phase<Tag, NSLAB, NL, NST, false>(p.act, p.w_down[l], p.down_part, nullptr, p.B, p.H, p.I, p.S4, smem);
    CP_TICK();
    cp_grid_barrier(p.sync, ++gen, p.err);
    CP_TICK();
  }
  cp_norm_phase<Tag>(p, nullptr, p.down_part, p.S4, p.final_norm, p.out, nwaves, wave_id);
  CP_TICK();
  // every workgroup read word 0 before its first barrier, and none passes the last one before all have arrived there
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(p.sync, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace
}  // namespace mi

#ifdef MI_CP_TIMING
extern "C" int mi_debug_cp_ticks(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mi::mi_cp_ticks), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t mi_cp_stack_workspace_bytes(int32_t rows, int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads,
                                               int32_t intermediate) {
  using namespace mi;
  if (rows <= 0 || hidden <= 0 || num_q_heads <= 0 || num_kv_heads <= 0 || intermediate <= 0) return 0;
  const int64_t qkv = (int64_t)(num_q_heads + 2 * num_kv_heads) * kCpHD;
  // resid, xn, attn, act, then the three slab sets at their largest split (8), then the barrier state + error word
  int64_t b = (int64_t)rows * hidden * 2 * 2 + (int64_t)rows * num_q_heads * kCpHD * 2 + (int64_t)rows * intermediate * 2;
  b = (b + 255) & ~255ll;
  b += 8 * (int64_t)rows * (qkv + 2 * hidden) * 4;
  b = (b + 255) & ~255ll;
  return b + (kCpSyncWords + 64) * 4;
}

extern "C" void* mi_cp_stack_status(void* workspace, int32_t rows, int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads,
                                    int32_t intermediate) {
  const int64_t total = mi_cp_stack_workspace_bytes(rows, hidden, num_q_heads, num_kv_heads, intermediate);
  if (!workspace || total == 0) return nullptr;
  return reinterpret_cast<char*>(workspace) + total - 64 * 4;
}

extern "C" int mi_cp_stack_decode(void* out, const void* x, const void* const* w_qkv, const void* const* w_o,
                                  const void* const* w_gate_up, const void* const* w_down, const void* const* input_norm,
                                  const void* const* post_attn_norm, const void* const* q_norm, const void* const* k_norm,
                                  const void* final_norm, void* const* k_cache, void* const* v_cache,
                                  const int32_t* block_ids, int64_t block_ids_stride, const float* cos_sin_cache,
                                  int32_t max_position, int32_t position, int32_t rows, int32_t hidden, int32_t num_q_heads,
                                  int32_t num_kv_heads, int32_t head_dim, int32_t intermediate, int32_t num_layers,
                                  int32_t block_size, int64_t cache_block_stride, int64_t cache_token_stride,
                                  int64_t cache_head_stride, float norm_eps, float scale, void* workspace,
                                  int64_t workspace_bytes, int32_t workspace_fresh, int dtype, void* stream) {
  using namespace mi;
  if (rows == 0) return MI_OK;
  if (!out || !x || !w_qkv || !w_o || !w_gate_up || !w_down || !input_norm || !post_attn_norm || !q_norm || !k_norm ||
      !final_norm || !k_cache || !v_cache || !block_ids || !cos_sin_cache || !workspace)
    return MI_ERR_INVALID_ARG;
  if (rows < 0 || num_layers <= 0 || max_position <= 0 || position < 0) return MI_ERR_INVALID_ARG;
  if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_ERR_UNSUPPORTED;
  if (head_dim != kCpHD || num_layers > kCpMaxLayers || num_q_heads != 2 * num_kv_heads) return MI_ERR_UNSUPPORTED;
  if (hidden % 64 || hidden > 1024 || intermediate % 64 || (num_q_heads * kCpHD) % 64) return MI_ERR_UNSUPPORTED;
  if (block_size > 64 || position >= block_size || position > 32) return MI_ERR_UNSUPPORTED;
  if (rows > 4096) return MI_ERR_UNSUPPORTED;   // slab sets are addressed with 32-bit byte offsets
  if (cache_block_stride % 8 || cache_token_stride % 8 || cache_head_stride % 8) return MI_ERR_INVALID_ARG;
  if (((uintptr_t)out | (uintptr_t)x | (uintptr_t)workspace | (uintptr_t)final_norm) & 15) return MI_ERR_INVALID_ARG;
  if (workspace_bytes < mi_cp_stack_workspace_bytes(rows, hidden, num_q_heads, num_kv_heads, intermediate))
    return MI_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  CpStackParams p;
  p.B = rows; p.H = hidden; p.NQ = num_q_heads; p.NKV = num_kv_heads; p.I = intermediate; p.L = num_layers;
  p.pos = position;
  p.eps = norm_eps;
  p.scale_log2e = scale * 1.4426950408889634f;
  for (int l = 0; l < num_layers; ++l) {
    p.w_qkv[l] = reinterpret_cast<const uint16_t*>(w_qkv[l]);
    p.w_o[l] = reinterpret_cast<const uint16_t*>(w_o[l]);
    p.w_gu[l] = reinterpret_cast<const uint16_t*>(w_gate_up[l]);
    p.w_down[l] = reinterpret_cast<const uint16_t*>(w_down[l]);
    p.ln1[l] = reinterpret_cast<const uint16_t*>(input_norm[l]);
    p.ln2[l] = reinterpret_cast<const uint16_t*>(post_attn_norm[l]);
    p.qn[l] = reinterpret_cast<const uint16_t*>(q_norm[l]);
    p.kn[l] = reinterpret_cast<const uint16_t*>(k_norm[l]);
    p.kc[l] = reinterpret_cast<uint16_t*>(k_cache[l]);
    p.vc[l] = reinterpret_cast<uint16_t*>(v_cache[l]);
    if (!p.w_qkv[l] || !p.w_o[l] || !p.w_gu[l] || !p.w_down[l] || !p.ln1[l] || !p.ln2[l] || !p.qn[l] || !p.kn[l] || !p.kc[l] ||
        !p.vc[l])
      return MI_ERR_INVALID_ARG;
    if (((uintptr_t)p.w_qkv[l] | (uintptr_t)p.w_o[l] | (uintptr_t)p.w_gu[l] | (uintptr_t)p.w_down[l] | (uintptr_t)p.ln1[l] |
         (uintptr_t)p.ln2[l] | (uintptr_t)p.kc[l] | (uintptr_t)p.vc[l]) & 15)
      return MI_ERR_INVALID_ARG;
  }
  p.final_norm = reinterpret_cast<const uint16_t*>(final_norm);
  p.c_blk = cache_block_stride; p.c_tok = cache_token_stride; p.c_head = cache_head_stride;
  p.block_ids = block_ids;
  p.bt_stride = block_ids_stride;
  p.block_size = block_size;
  p.max_position = max_position;
  p.cos_sin = cos_sin_cache;
  p.x_in = reinterpret_cast<const uint16_t*>(x);
  p.out = reinterpret_cast<uint16_t*>(out);
  const int64_t qkv = (int64_t)(num_q_heads + 2 * num_kv_heads) * kCpHD;
  char* ws = reinterpret_cast<char*>(workspace);
  p.resid = reinterpret_cast<uint16_t*>(ws);
  p.xn = p.resid + (int64_t)rows * hidden;
  p.attn = p.xn + (int64_t)rows * hidden;
  p.act = p.attn + (int64_t)rows * num_q_heads * kCpHD;
  int64_t off = (int64_t)rows * hidden * 2 * 2 + (int64_t)rows * num_q_heads * kCpHD * 2 + (int64_t)rows * intermediate * 2;
  off = (off + 255) & ~255ll;
  p.qkv_part = reinterpret_cast<float*>(ws + off);
  p.o_part = p.qkv_part + 8 * (int64_t)rows * qkv;
  p.down_part = p.o_part + 8 * (int64_t)rows * hidden;
  off += 8 * (int64_t)rows * (qkv + 2 * hidden) * 4;
  off = (off + 255) & ~255ll;
  p.sync = reinterpret_cast<uint32_t*>(ws + off);
  p.err = p.sync + kCpSyncWords;
  if (workspace_fresh) {   // first use of this workspace: the barrier state starts from zero
    if (hipMemsetAsync(p.sync, 0, (kCpSyncWords + 64) * 4, s) != hipSuccess) return MI_ERR_LAUNCH;
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return MI_ERR_LAUNCH;
  // Split-K so that every linear phase has about one work item per CU: tiles x slices ~ CUs (k slices of >= 128).
  const bool small = rows <= 64;
  auto split_for = [&](int n_cols, int k, int wrows) {
    const int tiles = (n_cols + wrows - 1) / wrows, rb = (rows + 63) / 64;
    int s_ = cus / (tiles * rb);
    const int cap = k / 128;
    s_ = s_ < 1 ? 1 : (s_ > 8 ? 8 : s_);
    return s_ > cap ? cap : s_;
  };
  const int wrows = small ? 32 : 128;
  p.S1 = split_for((int)qkv, hidden, wrows);
  p.S2 = split_for(hidden, num_q_heads * kCpHD, wrows);
  p.S4 = split_for(hidden, intermediate, wrows);
  const dim3 grid(cus);
#define MI_CP_LAUNCH(TAG, NSLAB, NL)                                                                                  \
  do {                                                                                                                \
    auto kern = cp_stack_kernel<TAG, NSLAB, NL, 2, 2>;                                                                \
    constexpr int lds = cp_ring_stages(NSLAB) * (4 * 2048 + NSLAB * 4096);                                            \
    static const hipError_t attr =                                                                                    \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);    \
    if (attr != hipSuccess) return MI_ERR_LAUNCH;                                                                     \
    hipLaunchKernelGGL(kern, grid, dim3(64 * (NSLAB + NL)), lds, s, p);                                               \
  } while (0)
  if (dtype == MI_DTYPE_BF16) {
    if (small) MI_CP_LAUNCH(bf16_tag, 1, 4);
    else MI_CP_LAUNCH(bf16_tag, 4, 4);
  } else {
    if (small) MI_CP_LAUNCH(f16_tag, 1, 4);
    else MI_CP_LAUNCH(f16_tag, 4, 4);
  }
#undef MI_CP_LAUNCH
  return launch_status();
}
