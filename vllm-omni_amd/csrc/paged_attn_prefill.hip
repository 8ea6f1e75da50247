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
//   * a workgroup = the G query heads that share one KV head (one wave per head, G <= 8 waves) x
//     16*NC consecutive query tokens; a wave owns NC column tiles of 16 query TOKENS and uses every
//     K and V fragment it reads from LDS for all of them (the kernel with one column tile per wave was
//     bound by LDS reads: 16 KB per wave and step for 24 MFMAs);
//   * both contractions are K = 32 MFMAs: the two transposed reads of a d-tile (keys 4g.. of either
//     16-row sub-tile) concatenate into one A operand, the two packed P registers pairs into the B
//     operand — the k index only has to name the same key on both sides;
//   * the K/V stages form a ring of FOUR shared by the workgroup: DMA issue is split over up to four
//     waves, ONE raw `s_barrier` per step orders "stage t+1 landed" and "iteration t-1 done" before
//     stage t+3 is issued over stage t-1 (no __syncthreads: its implied vmcnt(0) would drain the ring);
//   * the step is software-pipelined inside the wave: the scores of step t+1 are computed beside the
//     exponentials of step t (see the comment in front of `step`);
//   * the causal mask is per lane (column = query token); only steps that touch a column tile's
//     diagonal evaluate it; the context loop stops at the last position any column may see; tiles
//     are dispatched longest first over all kv heads;
//   * chunks longer than 16 tokens run the 32x32x16 formulation further down (`prefill_mfma32_kernel`, both head
//     sizes): same ring, same DMA, one 32-column tile per wave;
//   * a launch too small to fill the chip (the ragged last chunk of a long prompt, a TP shard with one KV head) cuts
//     every tile's keys into segments — separate workgroups of the same kernels over a key range, normalised fp32
//     partials + log-sum-exp in a workspace — and `prefill_merge_kernel` combines them (`prefill_plan` decides).
// FLOPs per layer: 2 * S^2 * d * n_q (causal).  MFMA-bound by design; K/V bytes are read once per
// (16*NC query tokens x G heads) and served from L2/MALL for any but the first tile of a sequence.
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
  int32_t qt;   // query tokens per workgroup tile (32x32x16 kernel)
  // context segments ("split-KV"): a tile's keys are cut into nseg runs of seg_steps 32-key steps, each run a workgroup
  // of its own that writes a normalised fp32 partial + its log2-sum-exp to `ws`; prefill_merge_kernel combines them.
  // nseg == 1 (seg_steps huge): no split, the kernels store the 16-bit output themselves.
  int32_t nseg, seg_steps;
  float* ws;    // [token][q head][nseg][head_dim + 1]
  float scale_log2e;
  int64_t q_stride, out_stride, blk_stride, tok_stride, head_stride;
};

constexpr int kColTile = 16;  // query tokens per MFMA column tile
#ifndef MI_PREFILL_ABLATE
#define MI_PREFILL_ABLATE 0   // experiments only (wrong results): 1 no softmax arithmetic, 2 no MFMAs, 4 no LDS fragment reads,
                              // 8 barrier on even steps only (an upper bound for a 64-key step), 16 no DMA (32x32x16 kernel)
#endif
#ifndef MI_PREFILL_PRIO
#define MI_PREFILL_PRIO 1
#endif
#ifndef MI_PREFILL_STAGGER
// 32x32x16 kernel, eight waves: waves 4-7 run HALF A STEP behind waves 0-3 — the workgroup's one barrier per step sits
// at the top of the step for waves 0-3 and between the S^T / softmax half and the PV half for waves 4-7, so that on
// every SIMD one wave's PV MFMAs (matrix pipe busy, vector issue idle) run beside its partner's S^T MFMAs +
// exponentials (vector issue busy, matrix pipe half idle) instead of both waves doing the same thing at the same
// time (MI355X_MICROARCH.md, Two waves per SIMD, item 9).  The DMA moves to waves 0-3 (their barrier is the one the
// ring's refill rule is stated for).  0 = every wave at the top of the step (round 3's form).
// Built, parity-green (tests/test_gpu_attention_prefill.py), and measured LEVEL with the unstaggered form on one box,
// interleaved runs (8192-token prompt: 0.726-0.730 against 0.732-0.735 ms per layer in one chunk, 0.789 / 0.802 in chunks of
// 4096, 0.737 / 0.721 in chunks of 2048: profiles/r04j_prefill_stagger_ab.log) — the two waves of a SIMD are not held back
// by running in phase; the default stays the round-3 form.
#define MI_PREFILL_STAGGER 0
#endif
#ifndef MI_PREFILL_LO_LATE
// 32x32x16 kernel, SPLIT form: the lo operand (p - hi, packed) is computed beside the eight hi PV MFMAs — a half of the
// step whose vector issue is nearly idle — instead of beside the S^T MFMAs, whose half is already bound by the vector
// issue port (16 exponentials + their fmas + the hi packs); the lo PV MFMAs follow in a second pass over the same V
// fragments.  0 = round 3's order (hi and lo back to back per (d-block, key half), both packs in the S^T half).
#define MI_PREFILL_LO_LATE 1
#endif
#ifndef MI_PREFILL_ROTATE
// 32x32x16 kernel, with MI_PREFILL_LO_LATE: the loop is rotated by half a step — iteration t runs [barrier] K(t+1) fragment
// reads -> PV(t-1) (16 MFMAs from registers: V(t-1) fragments, P(t-1)) -> wait K -> masks / maximum of step t -> V(t)
// fragment reads -> S^T(t+1) MFMAs beside the exponentials of step t -> wait V.  The K fragments' LDS latency — exposed in
// front of the S^T MFMAs in round 3's order (phase timing: profiles/r04r_prefill_phase_timing.log) — now runs under the
// PV MFMAs; V(t) and P(t) stay in registers across the barrier (+16 registers for the fp32 P of the lo operand).
// Built, parity-green, and SLOWER as compiled (0.805-0.809 against 0.731-0.733 ms per layer at the 8192-token prompt,
// profiles/r04t_prefill_rotate_ab.log): the extra live ranges push the wave past its 256 registers and hipcc spills
// fragments to scratch inside the loop (tools/check_prefill_isa.py flags the scratch_load).  Off.
#define MI_PREFILL_ROTATE 0
#endif
#ifndef MI_PREFILL_SCHED
#define MI_PREFILL_SCHED 1   // sched_group_barrier interleave of the exponentials with the QK^T MFMAs (A/B: -DMI_PREFILL_SCHED=0)
#endif

// Phase timing for tools/debug_prefill_timing.py: compiled only with -DMI_PREFILL_TIMING (`make dbg`
// builds a separate library), never in the product library.  Workgroup (0,0,0) — the longest tile —
// accumulates s_memrealtime deltas (100 MHz) per wave: 0 vmcnt wait, 1 barrier, 2 DMA issue, 3 masks/maximum,
// 4 K fragment reads, 5 QK^T MFMAs + exponentials (issue), 6 V fragment reads, 7 PV MFMAs (issue).
// A sample costs ~50 ns and orders only against instruction ISSUE (an MFMA still in the pipe is not waited for).
#ifdef MI_PREFILL_TIMING
}  // namespace
__device__ unsigned long long mi_prefill_ticks[16][16];
namespace {
#define MI_TICK(i)                                                       \
  do {                                                                   \
    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();    \
    tacc[i] += now_ - tprev;                                             \
    tprev = now_;                                                        \
  } while (0)
#else
#define MI_TICK(i) do { } while (0)
#endif

// waves per workgroup = query heads per KV head handled together; NI = waves that issue DMA;
// NC = column tiles (of 16 query tokens) per wave.
// SPLIT: see prefill_mfma32_kernel — P as a hi + lo pair of cache-dtype values, one extra MFMA per d-tile.
template <typename Tag, int HD, int NI, int NC, bool SPLIT>
__global__ __launch_bounds__(512) void prefill_mfma_kernel(const PrefillParams p) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  constexpr int QT = kColTile * NC;  // query tokens per workgroup
  constexpr int ROWB = HD * 2;
  constexpr int KS = HD / 32;
  constexpr int NT = HD / 16;
  constexpr int LPR = ROWB / 16;     // lanes (16-B chunks) per row: 16 / 8
  constexpr int RPI = 64 / LPR;      // rows per DMA wave-instruction: 4 / 8
  constexpr int IPS = 16 / RPI;      // DMA instructions per 16-row sub-tile: 4 / 2
  constexpr int IMG = kStep * ROWB;  // 8 KiB
  constexpr int STAGE = 2 * IMG;
  constexpr int NSTAGE = 4;
  constexpr int DMA_PER_STAGE = 4 * IPS;        // 16 wave-instructions fill one stage
  constexpr int SHARE = DMA_PER_STAGE / NI;      // per issuing wave
  __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int seq = blockIdx.y;
  // blockIdx.x = tile-major, (kv head, chunk of heads_per_wg query heads of its GQA group) minor: the dispatcher hands
  // out workgroups in ascending blockIdx.x, so ALL kv heads' longest tiles go first (with the kv head in blockIdx.z the
  // order was longest-to-shortest once per kv head, and the kernel lasted 2.4x its longest tile where the work is 2.0x),
  // and consecutive ids — which go round-robin over the 8 XCDs — are different kv heads: an XCD's L2 serves the K/V
  // stream of nz / 8 .. 1 kv heads instead of all of them.
  // (the last chunk of a group may be short: its surplus waves keep their DMA share and the barriers
  // but compute and store nothing)
  const int chunks = (p.kv_group + p.heads_per_wg - 1) / p.heads_per_wg;
  const int nz = (p.num_q_heads / p.kv_group) * chunks;
  const int zz = (int)blockIdx.x % nz, wrank = (int)blockIdx.x / nz;
  const int seg = wrank % p.nseg, tile_rank = wrank / p.nseg;   // segment-minor: a tile's segments are dispatched together
  const int kvh = zz / chunks;
  const int hg = (zz % chunks) * p.heads_per_wg + wid;
  const bool active = hg < p.kv_group;                            // wave-uniform
  const int head = kvh * p.kv_group + min(hg, p.kv_group - 1);    // this wave's query head

  const int q0 = p.query_start_loc[seq];
  const int q_len = p.query_start_loc[seq + 1] - q0;
  const int seq_len = p.seq_lens[seq];
  // longest (latest) tiles first: the dispatcher hands out blockIdx.x in ascending order
  const int tile0 = ((int)gridDim.x / (nz * p.nseg) - 1 - tile_rank) * QT;  // first query row of this tile inside the chunk
  if (tile0 >= q_len) return;              // whole workgroup: uniform
  const int ctx_off = seq_len - q_len;     // absolute position of query row 0
  const int rows = min(QT, q_len - tile0);

  const int col = lane & 15;  // query token inside a column tile
  const int g = lane >> 4;
  const int limit = ctx_off + tile0 + rows;                // keys [0, limit) are visible to the tile
  const int t_full = (limit + kStep - 1) / kStep;          // steps of the whole tile
  const int t_begin = seg * p.seg_steps;                   // this workgroup's run of them (a multiple of 4: stages)
  if (t_begin >= t_full) return;                           // whole workgroup, before any barrier
  // the LAST segment runs to the end of the tile whatever the host's bound on seq_lens was: a low bound costs balance,
  // never keys
  const int t_end = seg == p.nseg - 1 ? t_full : min(t_full, t_begin + p.seg_steps);
  const bool split = t_full > p.seg_steps;                 // the tile has more than one segment: partials + merge
  const int qpos0 = ctx_off + tile0 + col;                 // absolute position of my column in tile 0

  // constant address space: the (wave-uniform) block-id fetch becomes a SCALAR load (lgkmcnt), so
  // it never sits in the vmcnt queue the DMA waits are counted on
  typedef const __attribute__((address_space(4))) int32_t cint32_t;
  cint32_t* const bt_row = (cint32_t*)(p.block_table + (int64_t)seq * p.block_table_stride);
  const int bs_mask = (1 << p.bs_log2) - 1;
  const int last_blk = (limit - 1) >> p.bs_log2;

  // ---- Q^T fragments: column = token tile0 + 16c + col of head `head`, dims 32s + 8g .. -------
  uint4 qf[NC][KS];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool ok = kColTile * c + col < rows;
    const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) +
                         (int64_t)(q0 + tile0 + (ok ? kColTile * c + col : 0)) * p.q_stride + (int64_t)head * HD + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      qf[c][s] = ok ? *reinterpret_cast<const uint4*>(qb + 32 * s) : make_uint4(0, 0, 0, 0);
  }

  // ---- per-lane constants (same images as the decode kernel) ----------------------------------
  const int sw_h = swz_k<HD>(col);
  uint32_t k_rd[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) k_rd[s] = (uint32_t)(col * ROWB + (((4 * s + g) ^ sw_h) * 16));
  const int q4 = (lane & 15) >> 2, pp = lane & 3;
  const int v_row = 4 * g + q4;
  const int sw_v = swz_v<HD>(v_row);
  uint32_t v_rd[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
    v_rd[n] = (uint32_t)(IMG + v_row * ROWB + (((2 * n + (pp >> 1)) ^ sw_v) * 16) + 8 * (pp & 1));

  // ---- DMA issue.  A stage is four 16-row sub-tiles: K rows 0-15, K rows 16-31, V rows 0-15, V rows
  // 16-31; issuing wave `wid` owns SUBS consecutive ones.  Address arithmetic is scalar: ONE block-table
  // load and one 64-bit multiply per sub-tile (a sub-tile never straddles a block: block_size >= 16),
  // then its IPS DMA instructions step a uniform pointer by RPI rows and add a per-lane 32-bit offset.
  constexpr int SUBS = 4 / NI;
  // the LAST NI waves of the workgroup issue: with the younger half prioritised (below) the first waves are the ones
  // the barrier waits for (measured per step: 875 ns of compute + 145 ns of DMA issue on waves 0-2 against 620-760 ns on
  // waves 3-6 of a 7-wave workgroup)
  const int iw = wid - ((int)(blockDim.x >> 6) - NI);
  const uint64_t blk_bytes = (uint64_t)p.blk_stride * 2;
  const uint32_t tok_bytes = (uint32_t)p.tok_stride * 2;
  const uint32_t grp_bytes = RPI * tok_bytes;
  const char* const kbase_b = reinterpret_cast<const char*>(p.k_cache);
  const char* const vbase_b = reinterpret_cast<const char*>(p.v_cache);
  uint32_t my_off[SUBS][IPS];  // per-lane source offset inside a sub-tile: row, head, swizzled chunk
#pragma unroll
  for (int u = 0; u < SUBS; ++u) {
    const bool is_v = iw * SUBS + u >= 2;
#pragma unroll
    for (int ii = 0; ii < IPS; ++ii) {
      const int r = RPI * ii + lane / LPR;
      const int c = (lane % LPR) ^ (is_v ? swz_v<HD>(r) : swz_k<HD>(r));
      my_off[u][ii] = (uint32_t)(((int64_t)(lane / LPR) * p.tok_stride + (int64_t)kvh * p.head_stride) * 2 + c * 16);
    }
  }
  // The block id of a sub-tile is fetched one call AHEAD (issue() is called for t = 0, 1, 2, ... in order): the scalar
  // load's latency sat in front of the DMA instructions of every step otherwise (180 ns per step on the issuing waves).
  int32_t bt_next[SUBS];
#pragma unroll
  for (int u = 0; u < SUBS; ++u) bt_next[u] = iw >= 0 ? bt_row[min((t_begin * kStep + 16 * ((iw * SUBS + u) & 1)) >> p.bs_log2, last_blk)] : 0;
  // `interior`: the caller guarantees that every row of step t is below `limit` (no tail rows, no clamped block index)
  auto issue = [&](int t, int st, auto interior) {  // my share of the stage (the first waves issue nothing)
    if (iw < 0) return;
    const bool tail = !decltype(interior)::value && (t + 1) * kStep > limit;  // wave-uniform: rows of the step past `limit`
#pragma unroll
    for (int u = 0; u < SUBS; ++u) {
      const int sub = iw * SUBS + u;
      const bool is_v = sub >= 2;
      const int j = sub & 1;
      char* dst = smem + st * STAGE + (is_v ? IMG : 0) + 16 * j * ROWB;
      const char* const img = is_v ? vbase_b : kbase_b;
      const int tok0 = t * kStep + 16 * j;
      const int32_t blk_id = bt_next[u];
      bt_next[u] = bt_row[min((tok0 + kStep) >> p.bs_log2, last_blk)];
      if (!(tail && is_v)) {
        // K rows past `limit` may hold anything: their scores are replaced by -inf, never multiplied
        const char* src = img + (uint64_t)(uint32_t)blk_id * blk_bytes + (uint32_t)(tok0 & bs_mask) * tok_bytes;
#pragma unroll
        for (int ii = 0; ii < IPS; ++ii) {
          __builtin_amdgcn_global_load_lds((gbl_void*)(src + my_off[u][ii]), (lds_void*)(dst + RPI * ii * ROWB), 16, 0, 0);
          src += grp_bytes;
        }
      } else {
        // V rows at or past `limit` take the bytes of row limit-1: their probability is exactly 0, and
        // 0 * (finite) stays 0 where 0 * (stale NaN/Inf) would not.  Last step of a tile only.
#pragma unroll
        for (int ii = 0; ii < IPS; ++ii) {
          const int row0 = tok0 + RPI * ii;
          const int tok = min(row0 + lane / LPR, limit - 1);  // per lane
          const int blk = __builtin_amdgcn_readfirstlane(min(row0, limit - 1) >> p.bs_log2);
          const int c = (lane % LPR) ^ swz_v<HD>(RPI * ii + lane / LPR);
          const uint64_t off = (uint64_t)(uint32_t)bt_row[blk] * blk_bytes + (uint64_t)((uint32_t)(tok & bs_mask) * tok_bytes) +
                               (uint64_t)((int64_t)kvh * p.head_stride * 2) + (uint32_t)(c * 16);
          __builtin_amdgcn_global_load_lds((gbl_void*)(img + off), (lds_void*)(dst + RPI * ii * ROWB), 16, 0, 0);
        }
      }
    }
  };

  // Online softmax with the running maximum OFF the critical path (round 2; the step was VALU-issue-bound on the
  // softmax: profiles/r01_pmc_prefill_kernel.md).  mu[c] is a per-column reference maximum in UNSCALED score units.
  // p = exp2(fma(s, scale, -mu * scale)): the subtraction rides in the multiply every exponent needs anyway, and there
  // is no cross-lane maximum in front of the exponentials (issue costs: v_exp 8 cycles, every other vector instruction
  // 4, an MFMA 8 of its 16 — MI355X_MICROARCH.md constants table; the step is bound by that port).  mu is the reference
  // of the PREVIOUS steps (the guide's deferred-max rule, T13): a step whose scores stay within kThr (log2 units)
  // above it keeps it — p may then exceed 1 by at most 2^kThr, harmless for a bf16 / f16 P and fp32 accumulators —
  // and only a step where some score of the wave grows past the threshold (a lane-local test + one ballot) takes the
  // slow path BEFORE its exponentials: row maximum across the four lane rows, mu raised, acc and l (which hold exactly
  // the steps before this one) rescaled by exp2(-growth).  The first step of a tile has no reference yet and always
  // takes that path.
  constexpr float kThr = 6.0f;
  const float thr_raw = kThr / p.scale_log2e;   // the threshold in unscaled score units
  float mu[NC], nms[NC], mu_thr[NC], l[NC];     // reference, -mu * scale (the exponent's addend), mu + threshold
  f32x4_t acc[NC][NT];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    mu[c] = 0.f;
    nms[c] = 0.f;
    mu_thr[c] = 0.f;
    l[c] = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[c][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // Software pipeline inside every wave (MI355X_MICROARCH.md constants table: an MFMA 16x16x32 holds the SIMD's vector
  // issue port for 8 of its 16 cycles; the other 8 take two plain vector instructions or one v_exp): iteration t issues
  //   [masks, lane-local maximum, rare raise of the reference — on the scores of step t, computed in iteration t-1]
  //   K(t+1) fragments;  16 MFMAs  s(t+1) = K(t+1).Q^T   INTERLEAVED with   the exponentials, row sums and packs of step t
  //   V(t) fragments;    16 MFMAs  acc += V(t)^T.P(t)
  // so the vector work of a step sits in the gaps of the matrix work of the next one instead of between two dependent
  // groups of MFMAs.  Two score sets (by stage parity) are live; K(t+1) is read one iteration early, so the ring has
  // four stages and the loaders run three stages ahead: after barrier t every wave is done with iteration t-1 (its last
  // reads: V(t-1), K(t)), and stage t+3 replaces stage t-1.  The last iteration of a tile computes scores of a stage
  // that was never filled; nothing reads them.
#ifdef MI_PREFILL_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memrealtime();
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = tprev;   // shader clock vs the 100 MHz counter
#endif
  f32x4_t sc0[2][NC], sc1[2][NC];
  uint4 pb[NC], pl_[NC];  // P^T operand (hi, lo) per column tile: k index 8g + j <-> key 4g + j (j < 4), 16 + 4g + j - 4
  // MI_PREFILL_LO_LATE (as in the 32x32x16 kernel): the fp32 exponentials stay in registers until the PV half, where the
  // row sum and the lo operand are computed in the gaps of the hi MFMAs; the lo MFMAs follow in a second pass
  constexpr bool LO_LATE = SPLIT && MI_PREFILL_LO_LATE && (MI_PREFILL_ABLATE == 0);
  float vk[NC][8];
  auto kread = [&](auto stc, u32x4_t (&k0)[KS], u32x4_t (&k1)[KS]) {
    constexpr int SB = decltype(stc)::value * STAGE;
    if constexpr (MI_PREFILL_ABLATE & 4) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        k0[s] = u32x4_t{k_rd[s], k_rd[s], k_rd[s], k_rd[s]};
        k1[s] = k0[s];
        asm volatile("" : "+v"(k0[s]), "+v"(k1[s]));
      }
    } else if constexpr (KS == 4) lds_read_rows8<SB, SB + 16 * ROWB>(k_rd, k0, k1);
    else lds_read_rows4<SB, SB + 16 * ROWB>(k_rd, k0, k1);
  };
  auto qk_mfma = [&](const u32x4_t (&k0)[KS], const u32x4_t (&k1)[KS], f32x4_t (&s0)[NC], f32x4_t (&s1)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      s0[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      s1[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if constexpr (MI_PREFILL_ABLATE & 2) {
          s0[c][s & 3] += __builtin_bit_cast(float, k0[s][c]) * 1e-30f;
          s1[c][s & 3] += __builtin_bit_cast(float, k1[s][c]) * 1e-30f;
        } else {
          s0[c] = mfma16<Tag>(__builtin_bit_cast(uint4, k0[s]), qf[c][s], s0[c]);
          s1[c] = mfma16<Tag>(__builtin_bit_cast(uint4, k1[s]), qf[c][s], s1[c]);
        }
      }
    }
  };
  auto pv = [&](auto stc) {
    constexpr int SB = decltype(stc)::value * STAGE;
    u32x2_t a0[NT / 4][4], a1[NT / 4][4];
    if constexpr (MI_PREFILL_ABLATE & 4) {
#pragma unroll
      for (int nb = 0; nb < NT / 4; ++nb)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          a0[nb][n] = u32x2_t{v_rd[4 * nb + n], pb[0].x};
          a1[nb][n] = a0[nb][n];
          asm volatile("" : "+v"(a0[nb][n]), "+v"(a1[nb][n]));
        }
    } else if constexpr (NT == 8) lds_read_tr16<SB, SB + 16 * ROWB>(v_rd, a0[0], a1[0], a0[1], a1[1]);
    else lds_read_tr8<SB, SB + 16 * ROWB>(v_rd, a0[0], a1[0]);
    MI_TICK(6);
#pragma unroll
    for (int nb = 0; nb < NT / 4; ++nb) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {  // two transposed reads == one K = 32 A operand
        const uint4 a = make_uint4(a0[nb][n][0], a0[nb][n][1], a1[nb][n][0], a1[nb][n][1]);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if constexpr (MI_PREFILL_ABLATE & 2) {
            acc[c][4 * nb + n][0] += __builtin_bit_cast(float, a.x ^ pb[c].x) * 1e-30f;
          } else {
            acc[c][4 * nb + n] = mfma16<Tag>(a, pb[c], acc[c][4 * nb + n]);
            if constexpr (SPLIT && !LO_LATE) acc[c][4 * nb + n] = mfma16<Tag>(a, pl_[c], acc[c][4 * nb + n]);
          }
        }
      }
    }
    if constexpr (LO_LATE) {
      // row sums and lo = rnd16(p - hi) (exact subtraction) beside the hi MFMAs above ...
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float(&v)[8] = vk[c];
        l[c] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        const uint32_t w[4] = {pb[c].x, pb[c].y, pb[c].z, pb[c].w};
        uint32_t lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lo[i] = E::pack(v[2 * i] - E::lo(w[i]), v[2 * i + 1] - E::hi(w[i]));
        pl_[c] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
      if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
        for (int i = 0; i < NT * NC; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
          __builtin_amdgcn_sched_group_barrier(0x002, (27 * NC + NT * NC - 1) / (NT * NC), 3);
        }
      }
      // ... then the lo MFMAs over the same V fragments
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int nb = 0; nb < NT / 4; ++nb)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const uint4 a = make_uint4(a0[nb][n][0], a0[nb][n][1], a1[nb][n][0], a1[nb][n][1]);
            acc[c][4 * nb + n] = mfma16<Tag>(a, pl_[c], acc[c][4 * nb + n]);
          }
    }
  };
  // masks, lane-local maximum, the rare raise: everything of step t's softmax that branches
  auto pre = [&](int t, f32x4_t (&s0)[NC], f32x4_t (&s1)[NC], auto interior) {
    if constexpr (MI_PREFILL_ABLATE & 1) return;
    const bool first = t == t_begin;   // wave-uniform
    bool raise = first;
    float mx[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // lane: query column `col` of tile c, keys t*32 + 4g + r (s0[r]) and t*32 + 16 + 4g + r (s1[r]); masked in place
      const int qpos = qpos0 + kColTile * c;
      if (!decltype(interior)::value && (t + 1) * kStep > ctx_off + tile0 + kColTile * c + 1) {  // the step reaches past the tile's first diagonal key
        const int kb = t * kStep + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kb + r > qpos) s0[c][r] = -INFINITY;
          if (kb + 16 + r > qpos) s1[c][r] = -INFINITY;
        }
      }
      mx[c] = max3f(max3f(s0[c][0], s0[c][1], s0[c][2]), max3f(s0[c][3], s1[c][0], s1[c][1]), max2f(s1[c][2], s1[c][3]));
      raise |= mx[c] > mu_thr[c];
    }
    if (__any(raise)) {  // rare after the first steps of a tile
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float m = rows_allreduce_max(mx[c]);   // per column, the same in its four lane rows
        if (first) {
          mu[c] = m > -INFINITY ? m : 0.f;   // acc and l are still zero; a column that sees NO key of a later segment's
                                             // first step keeps a finite reference (its p are all exactly 0)
        } else {
          const float gr = max2f(m - mu[c], 0.f);
          const float alpha = fast_exp2(-gr * p.scale_log2e);
          mu[c] += gr;
          l[c] *= alpha;
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[c][n] *= alpha;
        }
        nms[c] = -mu[c] * p.scale_log2e;
        mu_thr[c] = mu[c] + thr_raw;
      }
    }
  };
  // the straight-line part: exponentials, row sums, P packed
  auto expo = [&](const f32x4_t (&s0)[NC], const f32x4_t (&s1)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if constexpr (MI_PREFILL_ABLATE & 1) {
        pb[c] = make_uint4(__builtin_bit_cast(uint32_t, s0[c][0]), __builtin_bit_cast(uint32_t, s0[c][1]),
                           __builtin_bit_cast(uint32_t, s1[c][2]), __builtin_bit_cast(uint32_t, s1[c][3]));
        l[c] += s0[c][2];
      } else {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fast_exp2(__builtin_fmaf(s0[c][r], p.scale_log2e, nms[c]));
          v[4 + r] = fast_exp2(__builtin_fmaf(s1[c][r], p.scale_log2e, nms[c]));
        }
        if constexpr (LO_LATE) {
#pragma unroll
          for (int r = 0; r < 8; ++r) vk[c][r] = v[r];
        } else {
          l[c] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        pb[c] = make_uint4(E::pack(v[0], v[1]), E::pack(v[2], v[3]), E::pack(v[4], v[5]), E::pack(v[6], v[7]));
        if constexpr (SPLIT && !LO_LATE) {   // lo = rnd16(p - hi): the subtraction is exact in fp32
          const uint32_t w[4] = {pb[c].x, pb[c].y, pb[c].z, pb[c].w};
          uint32_t lo[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) lo[i] = E::pack(v[2 * i] - E::lo(w[i]), v[2 * i + 1] - E::hi(w[i]));
          pl_[c] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
      }
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  // `interior` steps — at least three more steps follow and no key of the step is masked for any column of the tile —
  // carry no end-of-tile or diagonal logic at all: the only branch left in them is the reference raise.  (With that
  // logic inline the step ran a dozen wave-uniform branches, most of them taken.)
  auto step = [&](int t, auto stc, auto interior) {
    constexpr int ST = decltype(stc)::value;
    constexpr bool INT = decltype(interior)::value;
    using NEXT = std::integral_constant<int, (ST + 1) % NSTAGE>;
    // stage t+1 (read below) complete: my DMAs of it landed — only those of stage t+2 may still fly — then the barrier
    if (INT || t + 2 < t_end) wait_vmcnt<SHARE>(); else wait_vmcnt<0>();
    MI_TICK(0);
    __builtin_amdgcn_s_barrier();
    MI_TICK(1);
    if (INT || t + 3 < t_end) issue(t + 3, (ST + 3) % NSTAGE, interior);
    MI_TICK(2);
    if (active) {
      pre(t, sc0[ST & 1], sc1[ST & 1], interior);
      MI_TICK(3);
      u32x4_t k0[KS], k1[KS];
      kread(NEXT{}, k0, k1);
      MI_TICK(4);
      qk_mfma(k0, k1, sc0[(ST + 1) & 1], sc1[(ST + 1) & 1]);
      expo(sc0[ST & 1], sc1[ST & 1]);
      if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
        for (int i = 0; i < 2 * NC * KS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // two plain vector instructions (an exponent's fma first)
          __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);   // one transcendental
        }
      }
      MI_TICK(5);
      pv(stc);
    }
    MI_TICK(7);
  };
  // the second-dispatched half of the workgroup loses the issue arbitration of its SIMD on every step (older wave first:
  // 700 vs 1090 ns of compute per step measured); one static priority raise (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (MI_PREFILL_PRIO && wid >= 4) __builtin_amdgcn_s_setprio(1);
  issue(t_begin, 0, std::false_type{});
  if (t_begin + 1 < t_end) issue(t_begin + 1, 1, std::false_type{});
  if (t_begin + 2 < t_end) issue(t_begin + 2, 2, std::false_type{});
  // scores of step 0 ahead of the loop
  if (t_begin + 2 < t_end) wait_vmcnt<2 * SHARE>();
  else if (t_begin + 1 < t_end) wait_vmcnt<SHARE>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (active) {
    u32x4_t k0[KS], k1[KS];
    kread(S0{}, k0, k1);
    qk_mfma(k0, k1, sc0[0], sc1[0]);
  }
  // interior steps [0, t_int): t + 3 < t_end, (t + 1) * 32 <= first diagonal key of the tile; whole groups of four
  const int t_int = max(t_begin, min(t_end - 3, (ctx_off + tile0 + 1) / kStep)) & ~3;
  int t = t_begin;
  for (; t < t_int; t += 4) {
    step(t, S0{}, std::true_type{});
    step(t + 1, S1{}, std::true_type{});
    step(t + 2, S2{}, std::true_type{});
    step(t + 3, S3{}, std::true_type{});
  }
  for (; t < t_end; t += 4) {
    step(t, S0{}, std::false_type{});
    if (t + 1 < t_end) step(t + 1, S1{}, std::false_type{});
    if (t + 2 < t_end) step(t + 2, S2{}, std::false_type{});
    if (t + 3 < t_end) step(t + 3, S3{}, std::false_type{});
  }

#ifdef MI_PREFILL_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) mi_prefill_ticks[wid][i] = tacc[i];
    mi_prefill_ticks[wid][8] = __builtin_amdgcn_s_memtime() - clk0;
    mi_prefill_ticks[wid][9] = __builtin_amdgcn_s_memrealtime() - rt0;
    mi_prefill_ticks[wid][10] = (unsigned long long)t_end;
  }
#endif
  // ---- epilogue: O^T -> out[token][head][dim] -----------------------------------------------------
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float lc = l[c];
    lc = rows_allreduce_sum(lc);
    if (active && kColTile * c + col < rows) {
      const float inv = lc > 0.f ? 1.f / lc : 0.f;   // lc == 0: a column that saw no key of this segment
      if (split) {
        // normalised fp32 partial + log2-sum-exp of the segment; merged by prefill_merge_kernel
        float* w = p.ws + (((int64_t)(q0 + tile0 + kColTile * c + col) * p.num_q_heads + head) * p.nseg + seg) * (HD + 1);
#pragma unroll
        for (int n = 0; n < NT; ++n)
          *reinterpret_cast<f32x4_t*>(w + 16 * n + 4 * g) =
              f32x4_t{acc[c][n][0] * inv, acc[c][n][1] * inv, acc[c][n][2] * inv, acc[c][n][3] * inv};
        if (g == 0) w[HD] = lc > 0.f ? mu[c] * p.scale_log2e + __builtin_amdgcn_logf(lc) : -INFINITY;
      } else {
        uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (int64_t)(q0 + tile0 + kColTile * c + col) * p.out_stride +
                      (int64_t)head * HD + 4 * g;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          uint2 r;
          r.x = E::pack(acc[c][n][0] * inv, acc[c][n][1] * inv);
          r.y = E::pack(acc[c][n][2] * inv, acc[c][n][3] * inv);
          *reinterpret_cast<uint2*>(o + 16 * n) = r;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The 32x32x16 formulation: chunks longer than one 16-token tile.  Same ring, same DMA, same K image;
// what changes is the shape of the matrix instruction — a wave's 32 query tokens are ONE 32-column tile:
//   S^T (32 keys x 32 queries) = 8 MFMAs 32x32x16 over d;  O^T (128 d x 32 queries) += 4 d-blocks x 2 key halves
// (head_dim 64: 4 MFMAs and 2 d-blocks — measured 0.296 -> 0.278 ms per layer against the 16x16x32 kernel at the 8192-token
// prompt of 16 / 2 heads, 0.564 -> 0.519 at 32 / 4: profiles/r04_prefill_d64_mfma32.log).
// Why: the step is bound by the SIMD's vector issue port, which an MFMA holds for 8 cycles whether it computes 16 or 32
// cycles (MI355X_MICROARCH.md, constants table).  16 MFMAs of 32 cycles leave 16 x 24 = 384 issue cycles per wave and
// step for the ~320 the softmax needs; 32 MFMAs of 16 cycles left 256.
// Lane l: query column l % 32, half h = l / 32.  Score register i holds key 8 * (i / 4) + 4h + i % 4 of the step, so P
// registers [8m, 8m + 8) are the B operand of key half m as they stand: B k-slot j <-> key 16m + 8 * (j / 4) + 4h + j % 4,
// and the A operand (V^T, row d = l % 32 of a d-block) names the same keys: two transposed reads, rows 16m + 4h + (0..3)
// and 16m + 8 + 4h + (0..3).  A 16-lane group of a transposed read supplies 4 key rows x 16 d; the 32 lanes of an LDS
// cycle cover 4 key rows x 64 B, so the V image XORs the 16-B chunk index with 4 * (row & 3): four rows, four different
// 64-B segments, every bank once (the 16x16x32 image's 2 * (row & 7) would be 2-way here).
// NC = 32-column tiles per wave.  NC = 1: up to 8 waves of 256 registers, two per SIMD.  NC = 2: FOUR waves, one per
// SIMD, each with the whole 512-register file (accumulators and Q fragments in the accumulation half) and two column
// tiles: every K and V fragment read from LDS feeds two MFMAs instead of one, a step is 32 MFMAs between two fragment
// waits instead of 16, and the barrier joins four waves instead of eight.
// SPLIT: P enters the PV MFMA as a hi + lo PAIR of cache-dtype values (p = rnd16(p) + rnd16(p - rnd16(p)): 16 / 22
// significant bits instead of 8 / 11), one extra MFMA per (d-block, key half) against the same V operand — what brings the
// kernel inside rtol 1e-3 / atol 1e-4 of the fp32 oracle, as the decode kernel's split does.  SPLIT = false is the
// single-rounding P of flash kernels (`dot(P.to(V.dtype), V)`), kept for A/B (MI_PREFILL_ROUNDED_P=1).
template <typename Tag, int HD, int NI, int NC, bool SPLIT>
__global__ __launch_bounds__(NC == 2 ? 256 : 512) void prefill_mfma32_kernel(const PrefillParams p) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  const int QT = p.qt;               // query tokens per workgroup
  constexpr int ROWB = HD * 2;
  constexpr int KS = HD / 16;        // k-steps of S^T
  constexpr int NB = HD / 32;        // d-blocks of O^T
  constexpr int LPR = ROWB / 16;
  constexpr int RPI = 64 / LPR;
  constexpr int IPS = 16 / RPI;
  constexpr int IMG = kStep * ROWB;
  constexpr int STAGE = 2 * IMG;
  constexpr int NSTAGE = 4;
  constexpr int SHARE = 4 * IPS / NI;
  constexpr int SUBS = 4 / NI;
  __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int seq = blockIdx.y;
  const int chunks = (p.kv_group + p.heads_per_wg - 1) / p.heads_per_wg;
  const int nz = (p.num_q_heads / p.kv_group) * chunks;
  const int zz = (int)blockIdx.x % nz, wrank = (int)blockIdx.x / nz;
  const int seg = wrank % p.nseg, tile_rank = wrank / p.nseg;   // segment-minor: a tile's segments are dispatched together
  const int kvh = zz / chunks;
  const int h_first = (zz % chunks) * p.heads_per_wg;
  const int nheads = min(p.heads_per_wg, p.kv_group - h_first);   // query heads of this workgroup

  const int q0 = p.query_start_loc[seq];
  const int q_len = p.query_start_loc[seq + 1] - q0;
  const int seq_len = p.seq_lens[seq];
  const int tile0 = ((int)gridDim.x / (nz * p.nseg) - 1 - tile_rank) * QT;  // longest tiles first, all kv heads (see the 16x16x32 kernel)
  if (tile0 >= q_len) return;
  const int ctx_off = seq_len - q_len;
  const int rows = min(QT, q_len - tile0);

  const int col = lane & 31;
  const int h = lane >> 5;
  // The workgroup's nheads x QT (head, query) pairs are laid out head-major over the waves' 32-column tiles: column
  // cg = 32 * wave + col is query cg % QT of head cg / QT.  Every column of the swapped formulation is independent (its
  // own q vector, causal position, statistics and output row), so a tile may mix heads.  With QT = 32 and one wave
  // per head this is the plain mapping; with 7 heads the host picks 8 waves and QT = 36 (252 of 256 columns used)
  // instead of 7 waves and an idle eighth: every SIMD then carries two waves, 12.5 % more work per step.
  const bool active = (wid * NC * 32) / QT < nheads;              // wave-uniform: the wave owns a real column
  int qi[NC], head[NC], qpos[NC];
  bool valid[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int cg = (wid * NC + c) * 32 + col;
    const int hl = cg / QT;
    qi[c] = cg - hl * QT;
    valid[c] = hl < nheads && qi[c] < rows;
    head[c] = kvh * p.kv_group + h_first + min(hl, nheads - 1);
    qpos[c] = ctx_off + tile0 + qi[c];
  }
  const int limit = ctx_off + tile0 + rows;
  const int t_full = (limit + kStep - 1) / kStep;          // steps of the whole tile
  const int t_begin = seg * p.seg_steps;                   // this workgroup's run of them (a multiple of 4: stages)
  if (t_begin >= t_full) return;                           // whole workgroup, before any barrier
  // the LAST segment runs to the end of the tile whatever the host's bound on seq_lens was: a low bound costs balance,
  // never keys
  const int t_end = seg == p.nseg - 1 ? t_full : min(t_full, t_begin + p.seg_steps);
  const bool split = t_full > p.seg_steps;                 // the tile has more than one segment: partials + merge

  typedef const __attribute__((address_space(4))) int32_t cint32_t;
  cint32_t* const bt_row = (cint32_t*)(p.block_table + (int64_t)seq * p.block_table_stride);
  const int bs_mask = (1 << p.bs_log2) - 1;
  const int last_blk = (limit - 1) >> p.bs_log2;

  // ---- Q^T fragments: column = token tile0 + col of head `head`, dims 16s + 8h ..
  uint4 qf[NC][KS];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool ok = valid[c];
    const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)(q0 + tile0 + (ok ? qi[c] : 0)) * p.q_stride +
                         (int64_t)head[c] * HD + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[c][s] = ok ? *reinterpret_cast<const uint4*>(qb + 16 * s) : make_uint4(0, 0, 0, 0);
  }
  // ---- per-lane LDS read addresses
  uint32_t k_rd[KS];   // K row `col`, chunk 2s + h (dims 16s + 8h ..), swizzled by row & 15: a b128 lane group = 16 rows x 16 chunks
#pragma unroll
  for (int s = 0; s < KS; ++s) k_rd[s] = (uint32_t)(col * ROWB + (((2 * s + h) ^ swz_k<HD>(col)) * 16));
  const int j16 = lane & 15, db = (lane >> 4) & 1, rho = j16 >> 2;
  uint32_t v_rd[NB];   // V row 4h + rho (+ 16m + 8r as an immediate), dims 32n + 16db + 4 * (j16 & 3) ..
#pragma unroll
  for (int n = 0; n < NB; ++n)
    v_rd[n] = (uint32_t)(IMG + (4 * h + rho) * ROWB + ((((n ^ (HD == 128 ? rho : rho >> 1)) << 2) + 2 * db + ((j16 & 3) >> 1)) * 16) +
                         8 * (j16 & 1));

  // ---- DMA issue (as in the 16x16x32 kernel; only the V swizzle differs).  head_dim 64: a row is two 64-B segments and
  // two rows share a 256-B bank row, so the segment is XORed with bit 1 of the row — the 32 lanes of an LDS cycle (rows
  // r .. r+3, one segment each) then cover (row & 1) x (segment ^ (row >> 1)) = all four 64-B quarters of two bank rows.
  auto swz_v32 = [](int r) { return HD == 128 ? (r & 3) << 2 : ((r >> 1) & 1) << 2; };
  // staggered: eight waves, the second half one half-step behind the first (see MI_PREFILL_STAGGER)
  const bool stag = MI_PREFILL_STAGGER && NC == 1 && blockDim.x == 512;
  const bool late = stag && wid >= 4;                                  // wave-uniform
  const int iw = stag ? (wid < NI ? wid : -1) : wid - ((int)(blockDim.x >> 6) - NI);
  const uint64_t blk_bytes = (uint64_t)p.blk_stride * 2;
  const uint32_t tok_bytes = (uint32_t)p.tok_stride * 2;
  const uint32_t grp_bytes = RPI * tok_bytes;
  const char* const kbase_b = reinterpret_cast<const char*>(p.k_cache);
  const char* const vbase_b = reinterpret_cast<const char*>(p.v_cache);
  uint32_t my_off[SUBS][IPS];
#pragma unroll
  for (int u = 0; u < SUBS; ++u) {
    const bool is_v = iw * SUBS + u >= 2;
#pragma unroll
    for (int ii = 0; ii < IPS; ++ii) {
      const int r = RPI * ii + lane / LPR;
      const int c = (lane % LPR) ^ (is_v ? swz_v32(r) : swz_k<HD>(r));
      my_off[u][ii] = (uint32_t)(((int64_t)(lane / LPR) * p.tok_stride + (int64_t)kvh * p.head_stride) * 2 + c * 16);
    }
  }
  int32_t bt_next[SUBS];
#pragma unroll
  for (int u = 0; u < SUBS; ++u) bt_next[u] = iw >= 0 ? bt_row[min((t_begin * kStep + 16 * ((iw * SUBS + u) & 1)) >> p.bs_log2, last_blk)] : 0;
  auto issue = [&](int t, int st, auto interior) {
    if (iw < 0) return;
    if constexpr ((MI_PREFILL_ABLATE & 16) != 0) return;
    const bool tail = !decltype(interior)::value && (t + 1) * kStep > limit;
#pragma unroll
    for (int u = 0; u < SUBS; ++u) {
      const int sub = iw * SUBS + u;
      const bool is_v = sub >= 2;
      const int j = sub & 1;
      char* dst = smem + st * STAGE + (is_v ? IMG : 0) + 16 * j * ROWB;
      const char* const img = is_v ? vbase_b : kbase_b;
      const int tok0 = t * kStep + 16 * j;
      const int32_t blk_id = bt_next[u];
      bt_next[u] = bt_row[min((tok0 + kStep) >> p.bs_log2, last_blk)];
      if (!(tail && is_v)) {
        const char* src = img + (uint64_t)(uint32_t)blk_id * blk_bytes + (uint32_t)(tok0 & bs_mask) * tok_bytes;
#pragma unroll
        for (int ii = 0; ii < IPS; ++ii) {
          __builtin_amdgcn_global_load_lds((gbl_void*)(src + my_off[u][ii]), (lds_void*)(dst + RPI * ii * ROWB), 16, 0, 0);
          src += grp_bytes;
        }
      } else {  // V rows at or past `limit` take the bytes of row limit-1 (probability exactly 0 times a finite value)
#pragma unroll
        for (int ii = 0; ii < IPS; ++ii) {
          const int row0 = tok0 + RPI * ii;
          const int tok = min(row0 + lane / LPR, limit - 1);
          const int blk = __builtin_amdgcn_readfirstlane(min(row0, limit - 1) >> p.bs_log2);
          const int c = (lane % LPR) ^ swz_v32(RPI * ii + lane / LPR);
          const uint64_t off = (uint64_t)(uint32_t)bt_row[blk] * blk_bytes + (uint64_t)((uint32_t)(tok & bs_mask) * tok_bytes) +
                               (uint64_t)((int64_t)kvh * p.head_stride * 2) + (uint32_t)(c * 16);
          __builtin_amdgcn_global_load_lds((gbl_void*)(img + off), (lds_void*)(dst + RPI * ii * ROWB), 16, 0, 0);
        }
      }
    }
  };

  // ---- online softmax state (deferred reference maximum: see the 16x16x32 kernel)
  constexpr float kThr = 6.0f;
  const float thr_raw = kThr / p.scale_log2e;
  float mu[NC], nms[NC], mu_thr[NC], l[NC];
  f32x16_t acc[NC][NB];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    mu[c] = nms[c] = mu_thr[c] = l[c] = 0.f;
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][n][i] = 0.f;
  }
  f32x16_t sc[2][NC];
  uint4 pb[NC][2], pl_[NC][2];

  auto kread = [&](auto stc, u32x4_t (&k)[KS]) {
    if constexpr (KS == 8) lds_read_b128x8<decltype(stc)::value * STAGE>(k_rd, k);
    else lds_read_b128x4<decltype(stc)::value * STAGE>(k_rd, k);
  };
  auto k_issue = [&](auto off, u32x4_t (&k)[KS]) {
    if constexpr (KS == 8) lds_issue_b128x8<decltype(off)::value>(k_rd, k);
    else lds_issue_b128x4<decltype(off)::value>(k_rd, k);
  };
  auto v_issue = [&](auto off, u32x2_t (&o)[4 * NB]) {
    constexpr int SB = decltype(off)::value;
    if constexpr (NB == 4) lds_issue_tr4x4<SB, SB + 8 * ROWB, SB + 16 * ROWB, SB + 24 * ROWB>(v_rd, o);
    else lds_issue_tr2x4<SB, SB + 8 * ROWB, SB + 16 * ROWB, SB + 24 * ROWB>(v_rd, o);
  };
  // vector instructions the scheduler places behind each MFMA (the same totals per step whatever the head size)
  constexpr int kExpPerMfma = 16 / KS, kPackPerMfma = 8 / KS;
  constexpr int kLoPerMfma = 40 / (2 * NB), kSumPerMfma = (30 + 2 * NB - 1) / (2 * NB);
  auto qk_mfma = [&](const u32x4_t (&k)[KS], f32x16_t (&s)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) s[c][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int c = 0; c < NC; ++c) s[c] = mfma32<Tag>(__builtin_bit_cast(uint4, k[ks]), qf[c][ks], s[c]);
  };
  constexpr bool LO_LATE = SPLIT && MI_PREFILL_LO_LATE && NC == 1;
  // which: 0 = hi and lo back to back (round 3), 1 = the hi operands only, 2 = the lo operands only
  auto pv_mfma = [&](const u32x2_t (&o)[4 * NB], auto which) {
    constexpr int W = decltype(which)::value;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const uint4 a = make_uint4(o[4 * n + 2 * m][0], o[4 * n + 2 * m][1], o[4 * n + 2 * m + 1][0], o[4 * n + 2 * m + 1][1]);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if constexpr (W != 2) acc[c][n] = mfma32<Tag>(a, pb[c][m], acc[c][n]);
          if constexpr (SPLIT && W != 1) acc[c][n] = mfma32<Tag>(a, pl_[c][m], acc[c][n]);
        }
      }
  };
  // lo = rnd16(p - hi): the subtraction is exact in fp32
  auto lo_pack = [&](const float (&v)[NC][16]) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const uint32_t w[4] = {pb[c][m].x, pb[c][m].y, pb[c][m].z, pb[c][m].w};
        uint32_t lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          lo[i] = E::pack(v[c][8 * m + 2 * i] - E::lo(w[i]), v[c][8 * m + 2 * i + 1] - E::hi(w[i]));
        pl_[c][m] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
  };
  auto row_max = [&](const f32x16_t& s) {
    const float ma = max3f(max3f(s[0], s[1], s[2]), max3f(s[3], s[4], s[5]), max3f(s[6], s[7], s[8]));
    const float mb = max3f(max3f(s[9], s[10], s[11]), max3f(s[12], s[13], s[14]), s[15]);
    return max2f(ma, mb);
  };
  float mx_pre[NC];   // lane-local maxima of the NEXT step's unmasked scores, computed beside the PV MFMAs
#pragma unroll
  for (int c = 0; c < NC; ++c) mx_pre[c] = 0.f;
  auto pre = [&](int t, f32x16_t (&s)[NC], auto interior) {
    const bool first = t == t_begin;
    float mx[NC];
    bool raise = first;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if constexpr (decltype(interior)::value) {
        mx[c] = mx_pre[c];
      } else {
        if ((t + 1) * kStep > ctx_off + tile0 + 1) {   // the step reaches the tile's diagonal
          const int kb = t * kStep + 4 * h;
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (kb + 8 * (i >> 2) + (i & 3) > qpos[c]) s[c][i] = -INFINITY;
        }
        mx[c] = row_max(s[c]);
      }
      raise |= mx[c] > mu_thr[c];
    }
    if (__any(raise)) {  // rare after the first steps of a tile
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float a = mx[c], b = mx[c];
        permlane32_swap(a, b);             // the column's other 16 keys
        const float m = max2f(a, b);
        if (first) {
          mu[c] = m > -INFINITY ? m : 0.f;   // acc and l are still zero; a column that sees NO key of a later segment's
                                             // first step keeps a finite reference (its p are all exactly 0)
        } else {
          const float gr = max2f(m - mu[c], 0.f);
          const float alpha = fast_exp2(-gr * p.scale_log2e);
          mu[c] += gr;
          l[c] *= alpha;
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[c][n] *= alpha;
        }
        nms[c] = -mu[c] * p.scale_log2e;
        mu_thr[c] = mu[c] + thr_raw;
      }
    }
  };
  // exponentials and packs (beside the S MFMAs); the row sum of the same values later (beside the PV MFMAs)
  auto expo = [&](const f32x16_t (&s)[NC], float (&v)[NC][16]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[c][i] = fast_exp2(__builtin_fmaf(s[c][i], p.scale_log2e, nms[c]));
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        pb[c][m] = make_uint4(E::pack(v[c][8 * m], v[c][8 * m + 1]), E::pack(v[c][8 * m + 2], v[c][8 * m + 3]),
                              E::pack(v[c][8 * m + 4], v[c][8 * m + 5]), E::pack(v[c][8 * m + 6], v[c][8 * m + 7]));
        if constexpr (SPLIT && !LO_LATE) {   // lo = rnd16(p - hi): the subtraction is exact in fp32
          const uint32_t w[4] = {pb[c][m].x, pb[c][m].y, pb[c][m].z, pb[c][m].w};
          uint32_t lo[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            lo[i] = E::pack(v[c][8 * m + 2 * i] - E::lo(w[i]), v[c][8 * m + 2 * i + 1] - E::hi(w[i]));
          pl_[c][m] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
      }
    }
  };
  auto row_sum = [&](const float (&v)[NC][16]) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
      l[c] += (((v[c][0] + v[c][1]) + (v[c][2] + v[c][3])) + ((v[c][4] + v[c][5]) + (v[c][6] + v[c][7]))) +
              (((v[c][8] + v[c][9]) + (v[c][10] + v[c][11])) + ((v[c][12] + v[c][13]) + (v[c][14] + v[c][15])));
  };

#ifdef MI_PREFILL_TIMING
  // phases of tools/debug_prefill_timing.py: 0 vmcnt wait, 1 barrier, 2 DMA issue, 3 K read issue + masks / maximum,
  // 4 wait for the K fragments + V read issue, 5 S^T MFMAs + exponentials (issue), 6 wait for the V fragments, 7 PV passes
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memrealtime();
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = tprev;
#endif
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  // one iteration (see the 16x16x32 kernel): [branches of step t's softmax] K(t+1); S(t+1) MFMAs beside the exponentials
  // of step t; V(t); O += V(t)^T.P(t).  Ring of four stages, loaders three stages ahead.
  auto step = [&](int t, auto stc, auto interior) {
    constexpr int ST = decltype(stc)::value;
    constexpr bool INT = decltype(interior)::value;
    using NEXT = std::integral_constant<int, (ST + 1) % NSTAGE>;
    if (!late) {   // waves 4-7 of a staggered workgroup meet the others between the two halves of their step (below)
      if (INT || t + 2 < t_end) wait_vmcnt<SHARE>(); else wait_vmcnt<0>();
      MI_TICK(0);
      __builtin_amdgcn_s_barrier();
      MI_TICK(1);
      if (INT || t + 3 < t_end) issue(t + 3, (ST + 3) % NSTAGE, interior);
      MI_TICK(2);
    }
    if (active) {
      constexpr int SN = NEXT::value * STAGE, SB = ST * STAGE;
      // fragment reads run under the work that does not need them: K(t+1) under the branches of step t's softmax,
      // V(t) under the S(t+1) MFMAs and the exponentials
      u32x4_t k[KS];
      k_issue(std::integral_constant<int, SN>{}, k);
      __builtin_amdgcn_sched_barrier(0);   // the reads stay in FRONT of the work that covers their latency
      pre(t, sc[ST & 1], interior);
      MI_TICK(3);
      lds_wait_all();
      u32x2_t o[4 * NB];
      v_issue(std::integral_constant<int, SB>{}, o);
      __builtin_amdgcn_sched_barrier(0);
      MI_TICK(4);
      qk_mfma(k, sc[(ST + 1) & 1]);
      float v[NC][16];
      expo(sc[ST & 1], v);
      if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
        for (int i = 0; i < NC * KS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // one MFMA (32 cycles of pipe, 8 of issue)
          __builtin_amdgcn_sched_group_barrier(0x002, kExpPerMfma, 0);    // two exponents' fma (four at head_dim 64)
          __builtin_amdgcn_sched_group_barrier(0x400, kExpPerMfma, 0);    // as many exponentials
          __builtin_amdgcn_sched_group_barrier(0x002, kPackPerMfma, 0);   // a pack once its pair exists
        }
      }
      MI_TICK(5);
      lds_wait_all();
      MI_TICK(6);
      // a late wave has K(t+1) and V(t) in registers and is done with the LDS stages of this step: it meets the early
      // waves' NEXT top-of-step barrier here and runs its PV half beside their S^T / softmax half
      if (late) __builtin_amdgcn_s_barrier();
      // the PV MFMAs need nothing but P: the row sum of step t and the lane maximum of step t+1's scores (ready since
      // the S MFMAs retired) fill their gaps instead of standing in front of the next step
      if constexpr (LO_LATE) {
        // pass 1: the eight hi MFMAs with the lo operand's arithmetic in their gaps (8 expansions + 16 subtractions + 8
        // packs = 40 vector instructions: five per MFMA); pass 2: the eight lo MFMAs with the row sum and the next
        // step's lane maximum (30: four per MFMA)
        pv_mfma(o, std::integral_constant<int, 1>{});
        lo_pack(v);
        if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
          for (int i = 0; i < 2 * NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x002, kLoPerMfma, 1);
          }
        }
        pv_mfma(o, std::integral_constant<int, 2>{});
        row_sum(v);
#pragma unroll
        for (int c = 0; c < NC; ++c) mx_pre[c] = row_max(sc[(ST + 1) & 1][c]);
        if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
          for (int i = 0; i < 2 * NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
            __builtin_amdgcn_sched_group_barrier(0x002, kSumPerMfma, 2);
          }
        }
      } else {
      pv_mfma(o, std::integral_constant<int, 0>{});
      row_sum(v);
#pragma unroll
      for (int c = 0; c < NC; ++c) mx_pre[c] = row_max(sc[(ST + 1) & 1][c]);
      if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
        for (int i = 0; i < NC * 2 * NB * (SPLIT ? 2 : 1); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x002, (SPLIT ? 2 : 3) * 4 / NB, 1);
        }
      }
      }
      MI_TICK(7);
    } else if (late) {
      __builtin_amdgcn_s_barrier();   // a late wave without a real column still meets the others once per step
    }
  };
  // ---- rotated loop (MI_PREFILL_ROTATE): state that crosses the barrier
  constexpr bool ROT = LO_LATE && MI_PREFILL_ROTATE;
  u32x2_t o_keep[4 * NB];
  float v_keep[NC][16];
  // PV of the step whose V fragments and fp32 P are in o_keep / v_keep (hi operands in pb); `smax`: the scores whose
  // lane maximum the next `pre` wants (nullptr in the drain behind the loop)
  auto pv_block = [&](const f32x16_t* smax) {
    pv_mfma(o_keep, std::integral_constant<int, 1>{});
    lo_pack(v_keep);
    if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
      for (int i = 0; i < 2 * NB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x002, kLoPerMfma, 1);
      }
    }
    pv_mfma(o_keep, std::integral_constant<int, 2>{});
    row_sum(v_keep);
    if (smax) {
#pragma unroll
      for (int c = 0; c < NC; ++c) mx_pre[c] = row_max(smax[c]);
    }
    if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
      for (int i = 0; i < 2 * NB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
        __builtin_amdgcn_sched_group_barrier(0x002, kSumPerMfma, 2);
      }
    }
  };
  auto step_rot = [&](int t, auto stc, auto interior) {
    constexpr int ST = decltype(stc)::value;
    constexpr bool INT = decltype(interior)::value;
    using NEXT = std::integral_constant<int, (ST + 1) % NSTAGE>;
    if (INT || t + 2 < t_end) wait_vmcnt<SHARE>(); else wait_vmcnt<0>();
    MI_TICK(0);
    __builtin_amdgcn_s_barrier();
    MI_TICK(1);
    if (INT || t + 3 < t_end) issue(t + 3, (ST + 3) % NSTAGE, interior);
    MI_TICK(2);
    if (active) {
      constexpr int SN = NEXT::value * STAGE, SB = ST * STAGE;
      u32x4_t k[KS];
      k_issue(std::integral_constant<int, SN>{}, k);   // K(t+1): its latency runs under the PV MFMAs below
      __builtin_amdgcn_sched_barrier(0);
      if (t > t_begin) pv_block(sc[ST & 1]);  // PV(t-1) from registers; lane maxima of step t's scores for `pre`
      MI_TICK(3);
      lds_wait_all();
      MI_TICK(4);
      pre(t, sc[ST & 1], interior);
      v_issue(std::integral_constant<int, SB>{}, o_keep);   // V(t): under the S^T MFMAs
      __builtin_amdgcn_sched_barrier(0);
      qk_mfma(k, sc[(ST + 1) & 1]);
      expo(sc[ST & 1], v_keep);
      if constexpr (MI_PREFILL_SCHED) {
#pragma unroll
        for (int i = 0; i < NC * KS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, kExpPerMfma, 0);
          __builtin_amdgcn_sched_group_barrier(0x400, kExpPerMfma, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, kPackPerMfma, 0);
        }
      }
      MI_TICK(5);
      lds_wait_all();
      MI_TICK(6);
    }
  };
  if (MI_PREFILL_PRIO && wid >= 4) __builtin_amdgcn_s_setprio(1);
  issue(t_begin, 0, std::false_type{});
  if (t_begin + 1 < t_end) issue(t_begin + 1, 1, std::false_type{});
  if (t_begin + 2 < t_end) issue(t_begin + 2, 2, std::false_type{});
  if (t_begin + 2 < t_end) wait_vmcnt<2 * SHARE>();
  else if (t_begin + 1 < t_end) wait_vmcnt<SHARE>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (active) {
    u32x4_t k[KS];
    kread(S0{}, k);
    qk_mfma(k, sc[0]);
#pragma unroll
    for (int c = 0; c < NC; ++c) mx_pre[c] = row_max(sc[0][c]);
  }
  const int t_int = max(t_begin, min(t_end - 3, (ctx_off + tile0 + 1) / kStep)) & ~3;
  // Barrier pairing of a staggered workgroup (n = t_end - t_begin steps): the early waves' top-of-step barrier of step
  // j meets the late waves' mid-step barrier of step j - 1; the late waves therefore wait once more here (with the
  // early waves' barrier of the first step: the stage of step t_begin + 1 has landed behind it), and the early waves
  // once more after their last step (with the late waves' last mid-step barrier).
  if (late) __builtin_amdgcn_s_barrier();
  int t = t_begin;
  if (ROT && !stag) {
    for (; t < t_int; t += 4) {
      step_rot(t, S0{}, std::true_type{});
      step_rot(t + 1, S1{}, std::true_type{});
      step_rot(t + 2, S2{}, std::true_type{});
      step_rot(t + 3, S3{}, std::true_type{});
    }
    for (; t < t_end; t += 4) {
      step_rot(t, S0{}, std::false_type{});
      if (t + 1 < t_end) step_rot(t + 1, S1{}, std::false_type{});
      if (t + 2 < t_end) step_rot(t + 2, S2{}, std::false_type{});
      if (t + 3 < t_end) step_rot(t + 3, S3{}, std::false_type{});
    }
    if (active) pv_block(nullptr);   // the last step's PV
  } else {
  for (; t < t_int; t += 4) {
    step(t, S0{}, std::true_type{});
    step(t + 1, S1{}, std::true_type{});
    step(t + 2, S2{}, std::true_type{});
    step(t + 3, S3{}, std::true_type{});
  }
  for (; t < t_end; t += 4) {
    step(t, S0{}, std::false_type{});
    if (t + 1 < t_end) step(t + 1, S1{}, std::false_type{});
    if (t + 2 < t_end) step(t + 2, S2{}, std::false_type{});
    if (t + 3 < t_end) step(t + 3, S3{}, std::false_type{});
  }
  }
  if (stag && !late) __builtin_amdgcn_s_barrier();
#ifdef MI_PREFILL_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) mi_prefill_ticks[wid][i] = tacc[i];
    mi_prefill_ticks[wid][8] = __builtin_amdgcn_s_memtime() - clk0;
    mi_prefill_ticks[wid][9] = __builtin_amdgcn_s_memrealtime() - rt0;
    mi_prefill_ticks[wid][10] = (unsigned long long)(t_end - t_begin);
  }
#endif

  // ---- epilogue: O^T -> out[token][head][dim]; lane: d = 32n + 8 * (i / 4) + 4h + i % 4
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float la = l[c], lb = l[c];
    permlane32_swap(la, lb);
    const float lsum = la + lb;
    if (valid[c]) {
      const float inv = lsum > 0.f ? 1.f / lsum : 0.f;   // lsum == 0: a column that saw no key of this segment
      if (split) {
        float* w = p.ws + (((int64_t)(q0 + tile0 + qi[c]) * p.num_q_heads + head[c]) * p.nseg + seg) * (HD + 1);
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
            *reinterpret_cast<f32x4_t*>(w + 32 * n + 8 * q4 + 4 * h) =
                f32x4_t{acc[c][n][4 * q4] * inv, acc[c][n][4 * q4 + 1] * inv, acc[c][n][4 * q4 + 2] * inv, acc[c][n][4 * q4 + 3] * inv};
        if (h == 0) w[HD] = lsum > 0.f ? mu[c] * p.scale_log2e + __builtin_amdgcn_logf(lsum) : -INFINITY;
      } else {
        uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (int64_t)(q0 + tile0 + qi[c]) * p.out_stride + (int64_t)head[c] * HD + 4 * h;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            uint2 r;
            r.x = E::pack(acc[c][n][4 * q4] * inv, acc[c][n][4 * q4 + 1] * inv);
            r.y = E::pack(acc[c][n][4 * q4 + 2] * inv, acc[c][n][4 * q4 + 3] * inv);
            *reinterpret_cast<uint2*>(o + 32 * n + 8 * q4) = r;
          }
      }
    }
  }
}

// Combine the context segments of every split tile: out = sum_s 2^(lse_s - max) O_s / sum_s 2^(lse_s - max), fp32, one
// 16-bit rounding.  One workgroup per (query token, sequence) — a launch that is split at all has few tiles, so the
// parallelism has to come from tokens x heads; a token whose tile's context fits one segment was stored by its attention
// workgroup and is skipped.  Segments a tile does not have are never read.
template <typename Tag, int HD>
__global__ __launch_bounds__(256) void prefill_merge_kernel(const PrefillParams p) {
  scalar_cache_acquire();
  using E = Elem<Tag>;
  const int seq = blockIdx.y, r = blockIdx.x;   // r: token inside the sequence's chunk
  const int q0 = p.query_start_loc[seq];
  const int q_len = p.query_start_loc[seq + 1] - q0;
  if (r >= q_len) return;
  const int tile0 = r / p.qt * p.qt;
  const int rows = min(p.qt, q_len - tile0);
  const int limit = p.seq_lens[seq] - q_len + tile0 + rows;
  const int t_full = (limit + kStep - 1) / kStep;
  if (t_full <= p.seg_steps) return;
  const int ns = min((t_full + p.seg_steps - 1) / p.seg_steps, p.nseg);
  constexpr int Q = HD / 4;   // float4 groups per head row
  for (int i = threadIdx.x; i < p.num_q_heads * Q; i += 256) {
    const int d4 = i % Q, hq = i / Q;
    const float* w = p.ws + ((int64_t)(q0 + r) * p.num_q_heads + hq) * p.nseg * (HD + 1);
    float m = -INFINITY;
    for (int sidx = 0; sidx < ns; ++sidx) m = fmaxf(m, w[sidx * (HD + 1) + HD]);
    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (int sidx = 0; sidx < ns; ++sidx) {
      const float wt = fast_exp2(w[sidx * (HD + 1) + HD] - m);   // segment 0 always has a finite lse: m is finite
      den += wt;
      a += *reinterpret_cast<const f32x4_t*>(w + sidx * (HD + 1) + 4 * d4) * wt;
    }
    const float inv = 1.f / den;
    uint2 o;
    o.x = E::pack(a[0] * inv, a[1] * inv);
    o.y = E::pack(a[2] * inv, a[3] * inv);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)(q0 + r) * p.out_stride + (int64_t)hq * HD + 4 * d4) = o;
  }
}

template <typename Tag, int HD>
int launch_prefill_merge(const PrefillParams& p, int max_query_len, hipStream_t s) {
  if (p.nseg <= 1) return MI_OK;
  const dim3 grid(max_query_len, p.num_seqs, 1);
  hipLaunchKernelGGL((prefill_merge_kernel<Tag, HD>), grid, dim3(256), 0, s, p);
  return launch_status();
}

// Everything the host decides about a launch, in one place (the workspace query must size for the same plan the launch
// will take): heads per workgroup, kernel, tile, context segments.
struct PrefillPlan {
  int hpw, nc, kernel32, w4, waves, qt, nseg, seg_steps;
};

PrefillPlan prefill_plan(int max_query_len, int num_seqs, int num_q_heads, int num_kv_heads, int head_dim, int max_seq_len,
                         int force_segments) {
  static const int forced_hpw = [] {
    const char* e = getenv("MI_PREFILL_HPW");
    return e ? atoi(e) : 0;
  }();
  static const int forced_nc = [] {
    const char* e = getenv("MI_PREFILL_NC");
    return e ? atoi(e) : 0;
  }();
  static const bool mm16 = [] {   // MI_PREFILL_MM=16 keeps the 16x16x32 kernel for 32-token tiles (A/B)
    const char* e = getenv("MI_PREFILL_MM");
    return e && atoi(e) == 16;
  }();
  static const bool flat = [] {   // MI_PREFILL_FLAT=0: one wave per head
    const char* e = getenv("MI_PREFILL_FLAT");
    return !e || atoi(e) != 0;
  }();
  static const bool w4 = [] {     // MI_PREFILL_W4=1: the four-wave / 512-register experiment
    const char* e = getenv("MI_PREFILL_W4");
    return e && atoi(e) != 0;
  }();
  PrefillPlan pl;
  const int kv_group = num_q_heads / num_kv_heads;
  // Query heads (= waves) per workgroup: the whole GQA group when it fits 8 waves (256-VGPR budget), else chunks of 8
  // (the last one may be short).  A launch with fewer workgroups than two per CU (a TP shard with one KV head; a short
  // chunk) cannot balance tiles of very different context lengths: split the GQA group over two workgroups there.
  // Measured, 8k prompt: 7 heads / 1 KV head (S-A at TP=4) 0.216 -> 0.194 ms per layer in one chunk, 0.480 -> 0.401 in
  // chunks of 2048; 8 / 1 heads of 64 dims (S-B at TP=2) 0.188 -> 0.152 and 0.432 -> 0.351
  // (profiles/r02_prefill_shard_shapes.log).  MI_PREFILL_HPW overrides.
  int hpw = kv_group <= 8 ? kv_group : 8;
  {
    const int64_t tiles = (max_query_len + 31) / 32;
    if (tiles * num_seqs * num_kv_heads < 2 * 256 && hpw > 4) hpw = (hpw + 1) / 2;
  }
  if (forced_hpw > 0) hpw = forced_hpw < kv_group ? forced_hpw : kv_group;
  if (hpw > 8) hpw = 8;
  pl.hpw = hpw;
  // Column tiles per wave of the 16x16x32 kernel: 2 (32 query tokens per workgroup) once the chunk is long enough to use
  // them; a short chunk (decode-like mixed batches, tails) keeps 16-token tiles so that rows are not wasted.
  pl.nc = forced_nc ? forced_nc : (max_query_len > kColTile ? 2 : 1);
  static const bool mm16_d64 = [] {   // MI_PREFILL_MM64=16: head_dim 64 keeps the 16x16x32 kernel (A/B)
    const char* e = getenv("MI_PREFILL_MM64");
    return e && atoi(e) == 16;
  }();
  pl.kernel32 = pl.nc >= 2 && !mm16 && !(head_dim == 64 && mm16_d64);
  pl.w4 = pl.kernel32 && w4 && hpw >= 4;
  if (pl.w4) {
    // (experiment, off) 4..8 heads as FOUR waves x 2 tiles x 32 columns, one wave per SIMD with the 512-register file.
    // Measured 668-706 TFLOP/s against 842-889 for the 8-wave kernel at the 8k prompt: hipcc keeps the 128 accumulators
    // in the accumulation half but copies 110-190 registers between the halves in every step.
    pl.waves = 4;
    pl.qt = 256 / hpw;
  } else if (pl.kernel32) {
    // one 32-column tile per wave: one wave per head and 32-token tiles when the head count is a power of two, else the
    // next power of two of waves and as many tokens as fill their columns
    pl.waves = hpw;
    pl.qt = 32;
    if (flat && (hpw & (hpw - 1))) {
      pl.waves = hpw > 4 ? 8 : 4;
      pl.qt = 32 * pl.waves / hpw;
    }
  } else {
    pl.waves = hpw;
    pl.qt = kColTile * (pl.nc >= 2 ? 2 : 1);
  }
  // Context segments.  Automatic: only a launch with fewer than two workgroups per CU is split — the ragged last chunk
  // of a long prompt (77 tokens on 16k of context are 12 workgroups of 500 steps each), a TP shard with one KV head —
  // into enough segments for about four workgroups per CU, each at least 16 steps (512 keys) long.
  // force_segments: 1 = never, n > 1 = n segments (tests; at least 4 steps each).
  pl.nseg = 1;
  pl.seg_steps = 1 << 28;
  if (max_seq_len > 0 && force_segments != 1) {
    const int chunks = (kv_group + hpw - 1) / hpw;
    const int64_t base = (int64_t)((max_query_len + pl.qt - 1) / pl.qt) * num_seqs * num_kv_heads * chunks;
    const int t_max = (max_seq_len + kStep - 1) / kStep;
    int want = 1, cap = t_max / 16;
    if (force_segments > 1) {
      want = force_segments;
      cap = t_max / 4;
    } else if (base < 2 * 256) {
      want = (int)((4 * 256 + base - 1) / base);
    }
    int n = want < cap ? want : cap;
    if (n > 64) n = 64;
    if (n > 1) {
      pl.seg_steps = ((t_max + n - 1) / n + 3) & ~3;
      pl.nseg = (t_max + pl.seg_steps - 1) / pl.seg_steps;
      if (pl.nseg <= 1) {
        pl.nseg = 1;
        pl.seg_steps = 1 << 28;
      }
    }
  }
  return pl;
}

template <typename Tag, int HD>
int launch_prefill(PrefillParams p, const PrefillPlan& pl, int num_kv_heads, int max_query_len, hipStream_t s) {
  const int nz = num_kv_heads * ((p.kv_group + pl.hpw - 1) / pl.hpw);
  const dim3 grid(((max_query_len + pl.qt - 1) / pl.qt) * nz * pl.nseg, p.num_seqs, 1), block(64 * pl.waves);
  p.qt = pl.qt;
  // MI_PREFILL_ROUNDED_P=1: the single-rounding P of flash kernels instead of the hi + lo pair (A/B measurements only:
  // 0.546 against 0.781 ms per layer at the 8192-token prompt — the pair is +43 %, and what meets rtol 1e-3 / atol 1e-4)
  const bool rounded_p = [] {   // read per call (tests and A/B tools flip it inside one process)
    const char* e = getenv("MI_PREFILL_ROUNDED_P");
    return e && atoi(e) != 0;
  }();
#define MI_LAUNCH32(NI_, NC_)                                                                              \
  do {                                                                                                     \
    if (rounded_p) hipLaunchKernelGGL((prefill_mfma32_kernel<Tag, HD, NI_, NC_, false>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((prefill_mfma32_kernel<Tag, HD, NI_, NC_, true>), grid, block, 0, s, p);            \
  } while (0)
#define MI_LAUNCH16(NI_, NC_)                                                                                \
  do {                                                                                                       \
    if (rounded_p) hipLaunchKernelGGL((prefill_mfma_kernel<Tag, HD, NI_, NC_, false>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((prefill_mfma_kernel<Tag, HD, NI_, NC_, true>), grid, block, 0, s, p);            \
  } while (0)
  if (pl.kernel32) {
    if (pl.w4) MI_LAUNCH32(4, 2);
    else if (pl.waves >= 4) MI_LAUNCH32(4, 1);
    else if (pl.waves >= 2) MI_LAUNCH32(2, 1);
    else MI_LAUNCH32(1, 1);
  } else {
    if (pl.nc >= 2) {
      if (pl.waves >= 4) MI_LAUNCH16(4, 2);
      else if (pl.waves >= 2) MI_LAUNCH16(2, 2);
      else MI_LAUNCH16(1, 2);
    } else {
      if (pl.waves >= 4) MI_LAUNCH16(4, 1);
      else if (pl.waves >= 2) MI_LAUNCH16(2, 1);
      else MI_LAUNCH16(1, 1);
    }
  }
#undef MI_LAUNCH32
#undef MI_LAUNCH16
  const int rc = launch_status();
  if (rc != MI_OK) return rc;
  return launch_prefill_merge<Tag, HD>(p, max_query_len, s);
}

}  // namespace
}  // namespace mi

#ifdef MI_PREFILL_TIMING
extern "C" int mi_debug_prefill_ticks(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mi::mi_prefill_ticks), sizeof(unsigned long long) * 16 * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t mi_paged_attn_prefill_workspace_bytes(int32_t num_tokens, int32_t num_seqs, int32_t max_query_len,
                                                         int32_t max_seq_len, int32_t num_q_heads, int32_t num_kv_heads,
                                                         int32_t head_dim, int32_t force_segments) {
  using namespace mi;
  if (num_tokens <= 0 || num_seqs <= 0 || max_query_len <= 0 || num_q_heads <= 0 || num_kv_heads <= 0 ||
      num_q_heads % num_kv_heads || (head_dim != 64 && head_dim != 128))
    return 0;
  const PrefillPlan pl = prefill_plan(max_query_len, num_seqs, num_q_heads, num_kv_heads, head_dim, max_seq_len, force_segments);
  return pl.nseg > 1 ? (int64_t)num_tokens * num_q_heads * pl.nseg * (head_dim + 1) * (int64_t)sizeof(float) : 0;
}

extern "C" int mi_paged_attn_prefill_split(void* out, const void* q, const void* k_cache, const void* v_cache,
                                           const int32_t* block_table, const int32_t* seq_lens,
                                           const int32_t* query_start_loc, int32_t num_seqs, int32_t num_tokens,
                                           int32_t max_query_len, int32_t num_q_heads, int32_t num_kv_heads,
                                           int32_t head_dim, int32_t block_size, int32_t block_table_stride,
                                           float scale, int64_t q_token_stride, int64_t out_token_stride,
                                           int64_t cache_block_stride, int64_t cache_token_stride,
                                           int64_t cache_head_stride, int dtype, int32_t max_seq_len,
                                           int32_t force_segments, void* workspace, int64_t workspace_bytes,
                                           void* stream) {
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
  if (q_token_stride % 8 || out_token_stride % 8 || cache_block_stride % 8 || cache_token_stride % 8 ||
      cache_head_stride % 8)
    return MI_ERR_INVALID_ARG;
  if (((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out) & 15) return MI_ERR_INVALID_ARG;
  const PrefillPlan pl = prefill_plan(max_query_len, num_seqs, num_q_heads, num_kv_heads, head_dim,
                                      workspace ? max_seq_len : 0, force_segments);
  if (pl.nseg > 1) {
    if (((uintptr_t)workspace & 15)) return MI_ERR_INVALID_ARG;
    if (workspace_bytes < (int64_t)num_tokens * num_q_heads * pl.nseg * (head_dim + 1) * (int64_t)sizeof(float))
      return MI_ERR_WORKSPACE;
  }
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
  p.kv_group = num_q_heads / num_kv_heads;
  p.heads_per_wg = pl.hpw;
  p.bs_log2 = __builtin_ctz((unsigned)block_size);
  p.block_table_stride = block_table_stride;
  p.scale_log2e = scale * kLog2e;
  p.q_stride = q_token_stride;
  p.out_stride = out_token_stride;
  p.blk_stride = cache_block_stride;
  p.tok_stride = cache_token_stride;
  p.head_stride = cache_head_stride;
  p.qt = pl.qt;
  p.nseg = pl.nseg;
  p.seg_steps = pl.seg_steps;
  p.ws = reinterpret_cast<float*>(workspace);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MI_DTYPE_BF16)
    return head_dim == 128 ? launch_prefill<bf16_tag, 128>(p, pl, num_kv_heads, max_query_len, s)
                           : launch_prefill<bf16_tag, 64>(p, pl, num_kv_heads, max_query_len, s);
  return head_dim == 128 ? launch_prefill<f16_tag, 128>(p, pl, num_kv_heads, max_query_len, s)
                         : launch_prefill<f16_tag, 64>(p, pl, num_kv_heads, max_query_len, s);
}

extern "C" int mi_paged_attn_prefill(void* out, const void* q, const void* k_cache, const void* v_cache,
                                     const int32_t* block_table, const int32_t* seq_lens,
                                     const int32_t* query_start_loc, int32_t num_seqs, int32_t num_tokens,
                                     int32_t max_query_len, int32_t num_q_heads, int32_t num_kv_heads,
                                     int32_t head_dim, int32_t block_size, int32_t block_table_stride,
                                     float scale, int64_t q_token_stride, int64_t out_token_stride,
                                     int64_t cache_block_stride, int64_t cache_token_stride,
                                     int64_t cache_head_stride, int dtype, void* stream) {
  return mi_paged_attn_prefill_split(out, q, k_cache, v_cache, block_table, seq_lens, query_start_loc, num_seqs, num_tokens,
                                     max_query_len, num_q_heads, num_kv_heads, head_dim, block_size, block_table_stride,
                                     scale, q_token_stride, out_token_stride, cache_block_stride, cache_token_stride,
                                     cache_head_stride, dtype, 0, 1, nullptr, 0, stream);
}
