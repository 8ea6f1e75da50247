/*
 * vllm_omni_amd.h — C ABI of the MI355X (gfx950) decode hot path.
 *
 * This is the drop-in boundary underneath the Python plugin surface.  Every
 * entry point is `extern "C"`, takes raw device pointers, plain integer sizes
 * and strides (in ELEMENTS unless the name says bytes), a dtype enum and a
 * `hipStream_t` passed as `void*`.  Nothing here allocates device memory,
 * synchronises the device or throws: the return value is 0 on success or a
 * negative MI_ERR_* code, so every call is hipGraph-capturable on the stream
 * it is given.
 *
 * The reference (lishunyang12/vllm-omni) contains no native code; each entry
 * point replaces an *upstream vLLM call site reached from the reference*.
 * The comment above each declaration cites that call site (paths relative to
 * the reference root) and the SURVEY.md §8a row.
 */
#ifndef VLLM_OMNI_AMD_H_
#define VLLM_OMNI_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 3

/* dtype enum shared by every entry point */
enum {
  MI_DTYPE_F16 = 0,
  MI_DTYPE_BF16 = 1,
  MI_DTYPE_F32 = 2,
};

/* error codes (negative); 0 == success */
enum {
  MI_OK = 0,
  MI_ERR_INVALID_ARG = -1,    /* null pointer, negative size, misaligned stride */
  MI_ERR_UNSUPPORTED = -2,    /* head_dim / group size / block size / dtype not built */
  MI_ERR_WORKSPACE = -3,      /* workspace too small */
  MI_ERR_LAUNCH = -4,         /* hipLaunchKernel reported an error */
  MI_ERR_EXHAUSTED = -5,      /* block pool has no free block (allocator -> None) */
  MI_ERR_STATE = -6,          /* double free / unknown handle */
};

int mi_abi_version(void);
const char* mi_error_string(int code);

/* ------------------------------------------------------------------------
 * A4 — paged KV-cache write.
 * Replaces the KV-write half of upstream `Attention.forward(q,k,v)` called at
 * vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:154 (layout
 * contract: vllm_omni/distributed/omni_connectors/utils/kv_utils.py:378-450).
 *   cache[slot / block_size][slot % block_size][head][:] = src[t][head][:]
 * for every token t with slot_mapping[t] >= 0 (negative slots are padding and
 * are skipped).  Pure copy: bit-exact.
 * Strides are in elements; head_dim is contiguous everywhere.
 * ---------------------------------------------------------------------- */
int mi_reshape_and_cache(const void* k, const void* v, void* k_cache, void* v_cache,
                         const int64_t* slot_mapping, int32_t num_tokens,
                         int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                         int64_t k_token_stride, int64_t v_token_stride,
                         int64_t cache_block_stride, int64_t cache_token_stride,
                         int64_t cache_head_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * A7 / A7b / A4 fused — (optional per-head RMSNorm) -> neox RoPE -> KV write.
 * Replaces `self.rotary_pos_emb(positions, q, k)` + the KV write of
 * `self.attn(q, k, v)` (qwen2_old.py:153-154) and, for the Qwen3 family, the
 * q_norm/k_norm that precede RoPE.
 *  - q [T, num_q_heads*head_dim], k [T, num_kv_heads*head_dim] are rotated
 *    IN PLACE (first rot_dim dims of every head, rotate-half pairing
 *    (i, i + rot_dim/2)); arithmetic in fp32, one rounding to `dtype`.
 *  - cos_sin_cache is fp32 [max_position, rot_dim]: cos in [0, rot_dim/2),
 *    sin in [rot_dim/2, rot_dim) — the table layout of upstream `get_rope`.
 *  - positions is int64 [T] (mrope_section == NULL) or int64 [3, T] with row
 *    stride `positions_row_stride` and mrope_section[3] (sum == rot_dim/2):
 *    half-dim index j takes its angle from row 0/1/2 according to the section
 *    it falls in (M-RoPE rule).
 *  - q_norm_weight / k_norm_weight (dtype, [head_dim]) may be NULL (Qwen2).
 *  - k_cache == NULL skips the cache write (plain RoPE).  v may be NULL then.
 * ---------------------------------------------------------------------- */
int mi_rope_neox_and_cache(void* q, void* k, const void* v, const int64_t* positions,
                           int64_t positions_row_stride, const int32_t* mrope_section,
                           const float* cos_sin_cache, int32_t max_position, int32_t rot_dim,
                           const void* q_norm_weight, const void* k_norm_weight, float norm_eps,
                           void* k_cache, void* v_cache, const int64_t* slot_mapping,
                           int32_t num_tokens, int32_t num_q_heads, int32_t num_kv_heads,
                           int32_t head_dim, int32_t block_size, int64_t q_token_stride,
                           int64_t k_token_stride, int64_t v_token_stride,
                           int64_t cache_block_stride, int64_t cache_token_stride,
                           int64_t cache_head_stride, int dtype, void* stream);
/* The same op fed by the fp32 split-K partials [split][T][(nq + 2 nkv) * head_dim]
 * of qkv_proj (mi_linear_partials) instead of 16-bit q / k / v: a value is
 * rnd(sum over slices in order (+ qkv_bias)) — what mi_linear_reduce would have
 * stored — so results equal mi_linear + mi_rope_neox_and_cache bit for bit, one
 * launch and one round trip of the qkv row less.  q_out [T, q_token_stride]
 * receives the rotated queries; k and v go to the cache only (required). */
int mi_rope_neox_and_cache_splitk(void* q_out, const void* partials, int32_t split,
                                  const void* qkv_bias, const int64_t* positions,
                                  int64_t positions_row_stride, const int32_t* mrope_section,
                                  const float* cos_sin_cache, int32_t max_position,
                                  int32_t rot_dim, const void* q_norm_weight,
                                  const void* k_norm_weight, float norm_eps, void* k_cache,
                                  void* v_cache, const int64_t* slot_mapping, int32_t num_tokens,
                                  int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                  int32_t block_size, int64_t q_token_stride,
                                  int64_t cache_block_stride, int64_t cache_token_stride,
                                  int64_t cache_head_stride, int dtype, void* stream);

/* One decode step of attention for sequences whose WHOLE context lives in ONE
 * KV block of at most 64 tokens — the private cache of the Qwen3-Omni code
 * predictor (models/common/qwen3_code_predictor.py:318-626 runs
 * a 5-layer stack over the <= 33 positions of one talker token):
 * (q/k-norm ->) neox RoPE -> append K/V at slot `positions[t]` of block
 * `block_ids[t * block_ids_stride]` -> attention of row t's q heads over tokens
 * 0..positions[t] of that block, in one launch; the same math as
 * mi_rope_neox_and_cache followed by mi_paged_attn_decode (q, k rounded to the
 * cache dtype after the rotation; fp32 softmax and P.V; one output rounding).
 * q [T, nq*d], k / v [T, nkv*d] (views of one qkv row are fine), out [T, nq*d];
 * head_dim 128 and rot_dim == head_dim only, nq / nkv in {1, 2, 4, 8};
 * both norm weights or neither.  MI_ERR_UNSUPPORTED otherwise.
 * max_position_bound: host-known upper bound of positions[] (< block_size; it
 * sizes the per-lane registers: a launch early in the sequences runs at a higher
 * occupancy); positions == NULL: every row sits AT max_position_bound (the code
 * predictor steps all its sequences in lock-step). */
int mi_rope_cache_attend_block(void* out, const void* q, const void* k, const void* v,
                               const int64_t* positions, int32_t max_position_bound,
                               const float* cos_sin_cache,
                               int32_t max_position, int32_t rot_dim, const void* q_norm_weight,
                               const void* k_norm_weight, float norm_eps, void* k_cache,
                               void* v_cache, const int32_t* block_ids, int64_t block_ids_stride,
                               int32_t num_tokens, int32_t num_q_heads, int32_t num_kv_heads,
                               int32_t head_dim, int32_t block_size, int64_t out_token_stride,
                               int64_t q_token_stride, int64_t k_token_stride,
                               int64_t v_token_stride, int64_t cache_block_stride,
                               int64_t cache_token_stride, int64_t cache_head_stride, float scale,
                               int dtype, void* stream);

/* One decode pass of the code predictor's WHOLE layer stack in one launch
 * (csrc/cp_stack.hip; replaces, per pass, the 7 launches x num_layers of
 * Qwen3OmniMoeTalkerCodePredictor's decoder layers + the final norm:
 * vllm_omni/model_executor/models/common/qwen3_code_predictor.py:143-191,318-420).
 * Every row's new token sits at `position` of its own KV block (block_ids[row *
 * block_ids_stride]); x [rows, hidden] are the step's input rows, out [rows, hidden]
 * the final-norm output.  Weights per layer as arrays of `num_layers` HOST pointers to
 * device tensors: qkv [(nq + 2 nkv) * 128, hidden], o [hidden, nq * 128], gate_up
 * [2 * intermediate, hidden], down [hidden, intermediate], the four norm weights; no
 * biases; q heads = 2 x kv heads of 128 (the predictor's geometry), hidden <= 1024.
 * `workspace` (mi_cp_stack_workspace_bytes) holds the activations between phases and
 * the grid barrier's state: pass workspace_fresh = 1 on the first call with a
 * workspace (zeroes that state), 0 afterwards.  The launch is a persistent grid of one
 * workgroup per CU: it needs the whole device to itself while it runs; a barrier that
 * cannot complete gives up after a bounded wait and the launch reports through
 * mi_cp_stack_status. */
int64_t mi_cp_stack_workspace_bytes(int32_t rows, int32_t hidden, int32_t num_q_heads,
                                    int32_t num_kv_heads, int32_t intermediate);
int mi_cp_stack_decode(void* out, const void* x, const void* const* w_qkv, const void* const* w_o,
                       const void* const* w_gate_up, const void* const* w_down,
                       const void* const* input_norm, const void* const* post_attn_norm,
                       const void* const* q_norm, const void* const* k_norm, const void* final_norm,
                       void* const* k_cache, void* const* v_cache, const int32_t* block_ids,
                       int64_t block_ids_stride, const float* cos_sin_cache, int32_t max_position,
                       int32_t position, int32_t rows, int32_t hidden, int32_t num_q_heads,
                       int32_t num_kv_heads, int32_t head_dim, int32_t intermediate,
                       int32_t num_layers, int32_t block_size, int64_t cache_block_stride,
                       int64_t cache_token_stride, int64_t cache_head_stride, float norm_eps,
                       float scale, void* workspace, int64_t workspace_bytes,
                       int32_t workspace_fresh, int dtype, void* stream);
/* Device address of the workspace's error word (uint32, 0 = every barrier completed). */
void* mi_cp_stack_status(void* workspace, int32_t rows, int32_t hidden, int32_t num_q_heads,
                         int32_t num_kv_heads, int32_t intermediate);

/* The same op fed by the fp32 split-K partials [split][T][(nq + 2 nkv) * 128] of
 * qkv_proj (mi_linear_partials), as mi_rope_neox_and_cache_splitk is: a value is
 * rnd(sum over slices in order (+ qkv_bias)), so results equal mi_linear +
 * mi_rope_cache_attend_block bit for bit — the reduce launch and the RoPE launch
 * of a decode step at <= 32 rows ride in the attention launch. */
int mi_rope_cache_attend_block_splitk(void* out, const void* partials, int32_t split,
                                      const void* qkv_bias, const int64_t* positions,
                                      int32_t max_position_bound, const float* cos_sin_cache,
                                      int32_t max_position, int32_t rot_dim,
                                      const void* q_norm_weight, const void* k_norm_weight,
                                      float norm_eps, void* k_cache, void* v_cache,
                                      const int32_t* block_ids, int64_t block_ids_stride,
                                      int32_t num_tokens, int32_t num_q_heads, int32_t num_kv_heads,
                                      int32_t head_dim, int32_t block_size, int64_t out_token_stride,
                                      int64_t cache_block_stride, int64_t cache_token_stride,
                                      int64_t cache_head_stride, float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * A5 — paged-attention decode (q_len == 1).
 * Replaces the decode half of upstream `Attention.forward` (TRITON_ATTN on
 * ROCm: vllm_omni/engine/stage_init_utils.py:282-293) called at
 * qwen2_old.py:154.
 *   out[s,h] = softmax(q[s,h] . K[s, kv(h), :len_s]^T * scale) . V[s, kv(h), :len_s]
 *   kv(h) = h / (num_q_heads / num_kv_heads); token t of sequence s lives at
 *   cache[block_table[s, t / block_size]][t % block_size][kv(h)].
 * fp32 scores / softmax statistics / accumulators; one rounding to out_dtype.
 * The context is split into partitions of `partition_size` tokens (0 = pick
 * automatically); partial (m, l, O) triples go through `workspace` and are
 * merged by a second small kernel.  `max_seq_len` is a host-side upper bound
 * on seq_lens[] that sizes the grid, so the launch is graph-replayable while
 * sequences grow.  seq_lens[s] == 0 writes zeros.
 * out_dtype: MI_DTYPE_F32 is accepted for parity measurement.
 * flags: MI_DECODE_PACKED_PV lets P.V run on 16-bit operands.  That selects the matrix-core
 * kernel where it is built (head_dim 64/128, block_size >= 16, partition a multiple of 32),
 * which carries each (unnormalised, fp32) probability as a hi + lo PAIR of cache-dtype values
 * (16 / 22 significant bits for bf16 / f16) and so stays within rtol 1e-3 / atol 1e-4 of the
 * fp32 oracle; elsewhere the VALU kernel with P in fp32 (what a call without the flag always
 * gets).
 * MI_DECODE_FORCE_VALU | MI_DECODE_PACKED_PV: the dot2 form of the VALU kernel, which rounds P
 * once to the cache dtype (A/B measurements only).
 * MI_DECODE_ROUNDED_P: matrix-core kernel with P rounded ONCE to the cache dtype (the flash-style
 * `dot(P.to(V.dtype), V)`; 2^-9 / 2^-12 relative per term) — A/B measurements only.
 * ---------------------------------------------------------------------- */
#define MI_DECODE_PACKED_PV 1
#define MI_DECODE_FORCE_VALU 2
#define MI_DECODE_NONTEMPORAL 4 /* nt cache policy on the K/V stream (matrix-core kernel) */
#define MI_DECODE_ROUNDED_P 8
/* MI_DECODE_FUSED_MERGE: the matrix-core kernel merges the context partitions inside its own launch (the wave of a
 * (sequence, head group) that arrives last does it) instead of a second kernel — same bits.  The arrival counters
 * are the FIRST MI_DECODE_COUNTER_BYTES of `workspace` (mi_paged_attn_decode_workspace_bytes includes them; the
 * partial sums follow): with the flag the caller vouches that this region was zero before the first launch on the
 * workspace; every launch leaves it at zero (hipGraph replays need nothing in between). */
#define MI_DECODE_FUSED_MERGE 16
/* MI_DECODE_SKIP_MERGE (measurement only: bench.py's per-kernel roofline timing): the merge launch of a partitioned
 * context is left out — `out` is NOT written when the context is partitioned.  Never set by the product. */
#define MI_DECODE_SKIP_MERGE 32
#define MI_DECODE_COUNTER_BYTES (1 << 20)
int64_t mi_paged_attn_decode_workspace_bytes(int32_t num_seqs, int32_t num_q_heads,
                                             int32_t head_dim, int32_t max_seq_len,
                                             int32_t partition_size);
int mi_paged_attn_decode(void* out, const void* q, const void* k_cache, const void* v_cache,
                         const int32_t* block_table, const int32_t* seq_lens,
                         int32_t num_seqs, int32_t num_q_heads, int32_t num_kv_heads,
                         int32_t head_dim, int32_t block_size, int32_t block_table_stride,
                         int32_t max_seq_len, float scale, int64_t q_token_stride,
                         int64_t out_token_stride, int64_t cache_block_stride,
                         int64_t cache_token_stride, int64_t cache_head_stride,
                         int32_t partition_size, void* workspace, int64_t workspace_bytes,
                         int dtype, int out_dtype, int32_t flags, void* stream);
/* A4 + A7 + A5 in ONE launch for a pure-decode step (round 4): q [S, nq*d], k / v [S, nkv*d] arrive as qkv_proj wrote
 * them (views of one qkv row are fine); every wave rotates its q heads in registers (neox, fp32 table row of
 * positions[s], one rounding — mi_rope_neox_and_cache's arithmetic), and the wave that owns a sequence's last context
 * partition rotates the token's k, writes it and v to cache row slot_mapping[s] (< 0: no write) and waits for the
 * stores before it starts streaming — no other wave of the launch reads that row.  Result and cache contents equal
 * mi_rope_neox_and_cache followed by mi_paged_attn_decode bit for bit; q is left un-rotated in memory.  Matrix-core
 * kernel only (flags: PACKED_PV | NONTEMPORAL, no ROUNDED_P / FORCE_VALU; block_size >= 16; partition % 32 == 0),
 * rot_dim == head_dim, 1-D positions: MI_ERR_UNSUPPORTED otherwise (callers run the two launches).  q_norm_weight /
 * k_norm_weight (both or neither, [head_dim]): the Qwen3 family's per-head RMSNorm in front of the rotation, as in
 * mi_rope_neox_and_cache.
 * Replaces the rotary_pos_emb + attn pair of qwen2_old.py:153-154 on decode steps. */
int mi_rope_cache_paged_attn_decode(void* out, const void* q, const void* k, const void* v,
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
                                    int dtype, int out_dtype, int32_t flags, void* stream);
/* Partition size the auto heuristic picks: the fewest partitions that leave >= 512 waves (>= 1024 at
 * head_dim <= 64) — ONE (direct output, no workspace, no merge launch) once num_seqs * num_kv_heads
 * reaches that; 256..8192 tokens. */
int32_t mi_paged_attn_decode_auto_partition(int32_t num_seqs, int32_t num_kv_heads,
                                            int32_t max_seq_len, int32_t head_dim);
/* Partition size a call with these `flags` really uses: `requested` (0 = auto); any multiple of 32
 * tokens for the matrix-core kernel (it reloads its block-id window every 64 blocks), capped at
 * 64 KV blocks for the VALU kernel.  Size the workspace with THIS value. */
int32_t mi_paged_attn_decode_effective_partition(int32_t num_seqs, int32_t num_kv_heads,
                                                 int32_t max_seq_len, int32_t head_dim,
                                                 int32_t block_size, int32_t requested, int32_t flags);

/* ------------------------------------------------------------------------
 * A6 — paged-attention prefill / chunked prefill (q_len >= 1, varlen, causal
 * with context offset), MFMA bf16/f16.
 * Replaces the prefill half of upstream `Attention.forward` (same call site).
 *   query tokens of sequence s are rows [query_start_loc[s], query_start_loc[s+1]);
 *   query row i (0-based inside the chunk) sits at absolute position
 *   seq_lens[s] - q_len_s + i and attends to cache positions [0, that position].
 * K/V (including this chunk's own tokens) are read from the paged cache, so
 * the KV write (A4) must precede this call on the same stream.
 * ---------------------------------------------------------------------- */
int mi_paged_attn_prefill(void* out, const void* q, const void* k_cache, const void* v_cache,
                          const int32_t* block_table, const int32_t* seq_lens,
                          const int32_t* query_start_loc, int32_t num_seqs,
                          int32_t num_tokens, int32_t max_query_len, int32_t num_q_heads,
                          int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                          int32_t block_table_stride, float scale, int64_t q_token_stride,
                          int64_t out_token_stride, int64_t cache_block_stride,
                          int64_t cache_token_stride, int64_t cache_head_stride, int dtype,
                          void* stream);

/* The same op with its context split into segments where the launch would otherwise leave most CUs idle — the ragged
 * last chunk of a long prompt (77 tokens on 16k of context are 12 workgroups of 500 steps), a TP shard with one KV
 * head: every query tile's keys are cut into runs handled by separate workgroups that write a normalised fp32
 * partial and its log-sum-exp to `workspace`; a merge kernel combines them (same stream, no host involvement).
 *   max_seq_len     host upper bound of seq_lens (0: never split)
 *   force_segments  0 automatic (only launches below two workgroups per CU, >= 512 keys per segment),
 *                   1 never, n > 1 n segments (tests)
 *   workspace       >= mi_paged_attn_prefill_workspace_bytes(...) bytes for the same arguments, 16-byte aligned;
 *                   NULL: never split.  MI_ERR_WORKSPACE when too small.
 * mi_paged_attn_prefill == this with max_seq_len 0 and no workspace. */
int64_t mi_paged_attn_prefill_workspace_bytes(int32_t num_tokens, int32_t num_seqs, int32_t max_query_len,
                                              int32_t max_seq_len, int32_t num_q_heads, int32_t num_kv_heads,
                                              int32_t head_dim, int32_t force_segments);
int mi_paged_attn_prefill_split(void* out, const void* q, const void* k_cache, const void* v_cache,
                                const int32_t* block_table, const int32_t* seq_lens,
                                const int32_t* query_start_loc, int32_t num_seqs, int32_t num_tokens,
                                int32_t max_query_len, int32_t num_q_heads, int32_t num_kv_heads,
                                int32_t head_dim, int32_t block_size, int32_t block_table_stride, float scale,
                                int64_t q_token_stride, int64_t out_token_stride, int64_t cache_block_stride,
                                int64_t cache_token_stride, int64_t cache_head_stride, int dtype,
                                int32_t max_seq_len, int32_t force_segments, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A8 — RMSNorm and fused residual-add RMSNorm.
 * Replaces upstream `RMSNorm.forward(x[, residual])` at qwen2_old.py:212-216,
 * 223,325.  Rounding order (recorded in DESIGN.md): r' = round(x + r);
 * y = round(round(r' * rsqrt(mean(r'^2) + eps)) * w) with an fp32 reduction —
 * the order of the reference's HF-exact restatement
 * (vllm_omni/model_executor/models/common/qwen3_code_predictor.py:46-62).
 *   mi_rms_norm:           out = norm(x) * w            (out may alias x)
 *   mi_fused_add_rms_norm: residual <- x + residual ; x <- norm(residual) * w
 * ---------------------------------------------------------------------- */
int mi_rms_norm(void* out, const void* x, const void* weight, float eps, int32_t num_tokens,
                int32_t hidden, int64_t x_token_stride, int64_t out_token_stride, int dtype,
                void* stream);
int mi_fused_add_rms_norm(void* x, void* residual, const void* weight, float eps,
                          int32_t num_tokens, int32_t hidden, int64_t x_token_stride,
                          int64_t residual_token_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * A9 — SiLU-gate: out[t, j] = silu(x[t, j]) * x[t, inter + j].
 * Replaces upstream `SiluAndMul()` at qwen2_old.py:63,67.
 * ---------------------------------------------------------------------- */
int mi_silu_and_mul(void* out, const void* x, int32_t num_tokens, int32_t inter,
                    int64_t x_token_stride, int64_t out_token_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * A11 helpers — talker logit masking (`bad_word_processor`,
 * vllm_omni/model_executor/models/qwen2_5_omni/qwen2_5_omni_talker.py:150-169)
 * fused with greedy argmax (temperature 0).  Columns in [mask_lo0, mask_hi0),
 * [mask_lo1, mask_hi1) and the single column `mask_col` (if >= 0) are set to
 * -1e9 in place; ids[t] = first index of the row maximum afterwards.
 * logits are fp32 or `dtype`; ids int64.
 * ---------------------------------------------------------------------- */
int mi_mask_logits_argmax(void* logits, int64_t* ids, int32_t num_tokens, int32_t vocab,
                          int64_t row_stride, int32_t mask_lo0, int32_t mask_hi0,
                          int32_t mask_lo1, int32_t mask_hi1, int32_t mask_col, int dtype,
                          void* stream);

/* ------------------------------------------------------------------------
 * N4 — fused sparse-MoE decode.  Replaces upstream
 * `SharedFusedMoE` of the Qwen3-Omni talker (vllm_omni/model_executor/models/
 * qwen3_omni/qwen3_omni_moe_talker.py:476-513; algorithm: transformers
 * Qwen3OmniMoeTalkerTextSparseMoeBlock).  Five launches, no host sync:
 *   mi_moe_route   p = softmax_fp32(router_logits[t, :]); top-k ids (int32) and
 *                  weights (fp32, optionally renormalised, optionally rounded to
 *                  weight_round_dtype as HF does; -1 = keep fp32); per-expert
 *                  pair counts (zeroed inside; expert_counts may be NULL: no counting,
 *                  one launch less).  num_experts in {64, 128, 256}.
 *   mi_moe_align   groups the T*k (token, slot) pairs by expert, each group padded
 *                  to a multiple of rows_per_tile (32, 64, 128 or 256; mi_moe_rows_per_tile()
 *                  picks by mean pairs per expert): sorted_pairs[row] = pair or -1,
 *                  row_of_pair[pair], tile_expert[row / rows_per_tile], num_tiles[0].
 *                  Buffers sized by mi_moe_max_rows().  expert_counts NULL: counted
 *                  from topk_ids inside the launch.
 *   mi_moe_expert_gemm  gated != 0: out[row, :n_out] = silu(a[tok] . Wg_e^T) *
 *                  (a[tok] . Wu_e^T), weights [E, 2*n_out, k] (gate rows then up
 *                  rows), a = x [T, k], tok = sorted_pairs[row] / top_k;
 *                  gated == 0: out[row, :n_out] = a[row] . W_e^T, weights
 *                  [E, n_out, k], a = the gated output.  max_tiles bounds the
 *                  grid; tiles >= num_tiles[0] exit.  k % 64 == 0; n_out % 64
 *                  (gated) / % 128 == 0.
 *   mi_moe_combine out[t] = sum_j w[t, j] * partial[row_of_pair[t*k + j]]
 *                  (+ g * shared_out[t], g = sigmoid(x[t] . shared_gate_weight) or 1).
 * ---------------------------------------------------------------------- */
int32_t mi_moe_rows_per_tile(int32_t num_tokens, int32_t top_k, int32_t num_experts); /* 32, 64 or (prefill chunks) 256 */
int64_t mi_moe_max_rows(int32_t num_tokens, int32_t top_k, int32_t num_experts, int32_t rows_per_tile);
int mi_moe_route(int32_t* topk_ids, float* topk_weights, int32_t* expert_counts, const void* router_logits,
                 int32_t num_tokens, int32_t num_experts, int64_t logits_row_stride, int32_t top_k,
                 int32_t norm_topk_prob, int logits_dtype, int weight_round_dtype, void* stream);
int mi_moe_align(int32_t* sorted_pairs, int32_t* row_of_pair, int32_t* tile_expert, int32_t* num_tiles,
                 const int32_t* topk_ids, const int32_t* expert_counts, int32_t num_pairs,
                 int32_t num_experts, int64_t max_rows, int32_t rows_per_tile, void* stream);
int mi_moe_expert_gemm(void* out, const void* a_rows, const void* weights, const int32_t* sorted_pairs,
                       const int32_t* tile_expert, const int32_t* num_tiles, int32_t max_tiles,
                       int32_t gated, int32_t n_out, int32_t k, int32_t top_k, int64_t a_row_stride,
                       int64_t out_row_stride, int64_t weight_expert_stride, int32_t rows_per_tile, int dtype,
                       void* stream);
int mi_moe_combine(void* out, const void* partial, const int32_t* row_of_pair, const float* topk_weights,
                   const void* shared_out, const void* x, const void* shared_gate_weight,
                   int32_t num_tokens, int32_t hidden, int32_t top_k, int64_t partial_row_stride,
                   int64_t out_row_stride, int64_t x_row_stride, int64_t shared_row_stride, int dtype,
                   void* stream);

/* ------------------------------------------------------------------------
 * N3 — KV extract for the stage hand-off, fused.  Replaces,
 * for raw-data connectors, `_extract_kv_cache` -> `_slice_transfer_data_for_
 * target` -> `KVCacheTransferData.to_gpu_tensor`
 * (vllm_omni/distributed/omni_connectors/kv_transfer_manager.py:786-862,
 * 498-519, 144-157; utils/kv_utils.py:230-273): for every layer, K then V
 * rows [0, rows) of the request — row r lives in block block_ids[r /
 * block_size] at slot r % block_size — restricted to KV heads [head_lo,
 * head_lo + head_cnt), are written contiguously ([rows, head_cnt, head_dim])
 * at dst + k_offsets[layer] / dst + v_offsets[layer].
 * k_layers / v_layers: device arrays of device addresses of each layer's
 * [num_blocks, block_size, kv_heads, head_dim] views (0 = layer absent, its
 * offsets are ignored); all layers share the strides (in elements).
 * block_ids must already be validated (0 <= id < num_blocks); dst 16-byte
 * aligned; head_dim * elem_size a multiple of 16.
 * ---------------------------------------------------------------------- */
int mi_kv_pack(void* dst, const uint64_t* k_layers, const uint64_t* v_layers, const int64_t* k_offsets,
               const int64_t* v_offsets, int32_t num_layers, const int32_t* block_ids,
               int32_t num_block_ids, int32_t rows, int32_t block_size, int32_t head_lo, int32_t head_cnt,
               int32_t head_dim, int32_t elem_size, int64_t cache_block_stride,
               int64_t cache_token_stride, int64_t cache_head_stride, void* stream);

/* ------------------------------------------------------------------------
 * N2 — device-side sampler.  Replaces the sampler call of
 * vllm_omni/worker/gpu_ar_model_runner.py:626-657 (upstream vLLM v0.19
 * `Sampler`: penalties -> temperature -> top-k/top-p -> random_sample) for
 * the talker loop; shipped stage config: T 0.9, top_p 0.8, top_k 40,
 * repetition_penalty 1.05 (platforms/rocm/stage_configs/qwen2_5_omni.yaml:51-59).
 * One launch, one workgroup per row, row in LDS (vocab <= 15744):
 *   v = logits[r, :]; seen tokens: v = v > 0 ? v / rep : v * rep; v /= T
 *   top-k: keep v >= k-th largest (ties kept);  p = softmax(v)
 *   top-p: keep an element while the mass of strictly larger p is < top_p
 *          (elements tied with the last kept value are all kept)
 *   ids[r] = argmax_i p_i / -ln(u_i), u_i = (Philox4x32-10(ctr = {i, 0,
 *            offset_lo, offset_hi}, key = seed)[0] >> 8 + 0.5) / 2^24
 *   T == 0: ids[r] = first index of the maximum of the penalised logits.
 * Per-row arrays are device pointers; top_k / top_p / rep_penalty / seen_* may
 * be NULL (off).  seen_bitmap[seen_row[r]] is the request's prompt+output
 * token set (bit i of word i/32); update_seen != 0 sets the drawn id's bit.
 * logits are not modified.  ids int64.
 * ---------------------------------------------------------------------- */
int mi_sample_tokens(int64_t* ids, const void* logits, int32_t num_rows, int32_t vocab,
                     int64_t row_stride, int logits_dtype, const float* temperature,
                     const int32_t* top_k, const float* top_p, const float* rep_penalty,
                     uint32_t* seen_bitmap, const int32_t* seen_row, int32_t bitmap_words_per_row,
                     int32_t update_seen, const uint64_t* seed, const uint64_t* offset, void* stream);
/* The same draw with mi_embed_codes behind it in ONE launch (the code predictor's step,
 * qwen3_code_predictor.py:556-600: sample, store the code, embed it with that
 * codebook's table, append the row): out_a[r] (and out_b[r], may be NULL) =
 * embed_table[ids[r]] (16-bit rows of embed_dim, embed_dim % 8 == 0), ids_out[r *
 * ids_out_stride] = ids[r] (may be NULL).  The table has `vocab` rows (the drawn id is
 * always inside).  No repetition penalty / seen set in this form. */
int mi_sample_tokens_embed(int64_t* ids, const void* logits, int32_t num_rows, int32_t vocab,
                           int64_t row_stride, int logits_dtype, const float* temperature,
                           const int32_t* top_k, const float* top_p, const uint64_t* seed,
                           const uint64_t* offset, const void* embed_table, int32_t embed_dim,
                           void* out_a, int64_t out_a_token_stride, void* out_b,
                           int64_t out_b_token_stride, int64_t* ids_out, int64_t ids_out_stride,
                           void* stream);

/* ------------------------------------------------------------------------
 * A11 — talker per-step input embedding, batched:
 *   out[t] = embed_table[token_ids[t]] (+ reply_rows[reply_idx[t]] when reply_idx[t] >= 0)
 * i.e. `thinker_reply_part[0:1] + embed_input_ids(last token)` of
 * `_thinker_to_talker_decode_one_step` (vllm_omni/model_executor/models/qwen2_5_omni/
 * qwen2_5_omni.py:796-804) for every decode request at once.  fp32 add, one rounding.
 * reply_rows / reply_idx may be NULL (plain embedding gather).  Token ids outside
 * [0, vocab) yield a zero row (never an out-of-bounds read).
 * ---------------------------------------------------------------------- */
int mi_talker_decode_embeds(void* out, const void* embed_table, const int64_t* token_ids,
                            const void* reply_rows, const int64_t* reply_idx,
                            int32_t num_tokens, int32_t embed_dim, int32_t vocab,
                            int64_t num_reply_rows, int64_t out_token_stride, int dtype,
                            void* stream);

/* Code-predictor step glue (qwen3_code_predictor.py:560-600: the sampled
 * code of codebook s is stored, embedded with that codebook's table and appended to
 * the sequence): out_a[t] = out_b[t] = embed_table[token_ids[t]] (16-bit rows of
 * embed_dim, out_b may be NULL), ids_out[t * ids_out_stride] = token_ids[t]
 * (may be NULL) — one launch instead of a gather and three copies.  Ids outside
 * [0, vocab) yield a zero row. */
int mi_embed_codes(void* out_a, void* out_b, int64_t* ids_out, const void* embed_table,
                   const int64_t* token_ids, int32_t num_tokens, int32_t embed_dim, int32_t vocab,
                   int64_t out_a_token_stride, int64_t out_b_token_stride, int64_t ids_out_stride,
                   void* stream);

/* ------------------------------------------------------------------------
 * A13 — W8A8 int8 linear (scheme of vllm_omni/quantization/int8_config.py:
 * 180-212,323-342,391-418): dynamic symmetric per-token activation quant
 * (scale_t = absmax(x_t)/127, round-half-even, saturate to [-127,127]... see
 * DESIGN.md "unpinned"), int8 x int8 -> int32 MFMA GEMM against weight
 * [N, K] int8 with per-output-channel fp32 scale, dequant + optional bias.
 * ---------------------------------------------------------------------- */
int mi_dynamic_int8_quant(int8_t* xq, float* x_scale, const void* x, int32_t num_tokens,
                          int32_t hidden, int64_t x_token_stride, int dtype, void* stream);
int mi_int8_gemm_dequant(void* out, const int8_t* xq, const float* x_scale, const int8_t* wq,
                         const float* w_scale, const void* bias, int32_t m, int32_t n,
                         int32_t k, int64_t out_row_stride, int dtype, void* stream);
/* The activation quantisation fused into the op that produces the activation, and its vectorised stand-alone form.
 * mi_rms_norm_int8: RMSNorm (residual == NULL) or fused residual-add RMSNorm (residual updated in place, x NOT
 * modified) whose output leaves as int8 [T, hidden] + fp32 scale [T] — bit-identical to mi_rms_norm /
 * mi_fused_add_rms_norm followed by mi_dynamic_int8_quant (qwen2_old.py:212-216 + int8_config.py:323-342), one
 * launch.  mi_int8_quant_rows: mi_dynamic_int8_quant with 16-byte loads, rows <= 24576 elements. */
int mi_rms_norm_int8(int8_t* xq, float* x_scale, const void* x, void* residual, const void* weight, float eps,
                     int32_t num_tokens, int32_t hidden, int64_t x_token_stride, int64_t residual_token_stride,
                     int dtype, void* stream);
int mi_int8_quant_rows(int8_t* xq, float* x_scale, const void* x, int32_t num_tokens, int32_t hidden,
                       int64_t x_token_stride, int dtype, void* stream);
/* The tiled W8A8 kernel (csrc/skinny_gemm.hip instantiated on int8 rows: LDS-DMA ring, v_mfma_i32_16x16x64_i8,
 * dequantisation + bias in the epilogue, ONE launch per linear — two with split-K): out[m, n] =
 * (sum_k xq[m, k] * wq[n, k]) * x_scale[m] * w_scale[n] (+ bias[n]); int32 sums are exact.  m <= 512,
 * k % 128 == 0, n % 16 == 0, x_row_stride (bytes == elements) % 16 == 0.  Replaces `Int8LinearMethod.apply`
 * (vllm_omni/quantization/int8_config.py:323-342).  mi_int8_linear_silu_mul: gate_up_proj + SiluAndMul in one
 * launch — wq [2 * inter, k] (gate rows, then up rows), w_scale [2 * inter]; each dequantised GEMM output is
 * rounded to the 16-bit type before the activation, as the unfused pair rounds.  MI_ERR_UNSUPPORTED outside the
 * envelope (callers use mi_int8_gemm_dequant). */
int64_t mi_int8_linear_workspace_bytes(int32_t m, int32_t n, int32_t k);
int mi_int8_linear(void* out, const int8_t* xq, const float* x_scale, const int8_t* wq, const float* w_scale,
                   const void* bias, int32_t m, int32_t n, int32_t k, int64_t x_row_stride,
                   int64_t out_row_stride, void* workspace, int64_t workspace_bytes, int dtype, void* stream);
/* The split-K W8A8 projection without its reduce launch (exact int32 partials [split][m][n] in `workspace`), and the
 * fused residual-add RMSNorm that consumes them: x = rnd(dequant(sum over slices) (+ bias)), residual <- x + residual,
 * out = norm(residual) * weight as 16-bit rows (q_scale == NULL) or as the next W8A8 linear's int8 rows + q_scale —
 * bit-identical to mi_int8_linear -> mi_fused_add_rms_norm (-> mi_dynamic_int8_quant), two launches fewer. */
int mi_int8_linear_partials(const int8_t* xq, const int8_t* wq, int32_t m, int32_t n, int32_t k, int64_t x_row_stride,
                            void* workspace, int64_t workspace_bytes, int32_t* split_out, void* stream);
int mi_fused_add_rms_norm_splitk_int8(void* out, float* q_scale, const void* partials, int32_t split,
                                      const float* x_scale, const float* w_scale, const void* bias, void* residual,
                                      const void* weight, float eps, int32_t num_tokens, int32_t hidden,
                                      int64_t out_token_stride, int64_t residual_token_stride, int dtype, void* stream);
int mi_int8_linear_silu_mul(void* out, const int8_t* xq, const float* x_scale, const int8_t* wq,
                            const float* w_scale, int32_t m, int32_t inter, int32_t k, int64_t x_row_stride,
                            int64_t out_row_stride, int dtype, void* stream);
/* Dequant epilogue alone, for an int32 accumulator produced by a library int8 GEMM (hipBLASLt through
 * torch._int_mm): out[m, n] = acc[m, n] * x_scale[m] * w_scale[n] (+ bias[n]).  n % 4 == 0. */
int mi_int8_dequant(void* out, const int32_t* acc, const float* x_scale, const float* w_scale,
                    const void* bias, int32_t m, int32_t n, int64_t acc_row_stride,
                    int64_t out_row_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Decode-batch linear layer: out[m, n] = x[m, k] . weight[n, k]^T (+ bias[n]),
 * m <= 512 (two row blocks of 256 above 256) — the projections of one decode
 * step (qkv_proj / o_proj /
 * gate_up_proj / down_proj of vllm_omni/model_executor/models/qwen2_5_omni/
 * qwen2_5_omni_talker.py, which the reference runs as library GEMMs through
 * vLLM's ColumnParallelLinear / RowParallelLinear).  Weight-streaming kernel:
 * weights HBM -> registers once, x through LDS, split-K with fp32 partials in
 * `workspace` where n alone does not fill the chip (deterministic reduce, no
 * atomics).  k % 128 == 0, n % 16 == 0, rows 16-byte aligned.
 * mi_linear_workspace_bytes: bytes mi_linear needs for this shape (0 = none).
 * Returns MI_ERR_UNSUPPORTED for shapes outside the envelope (callers fall to
 * the library GEMM), MI_ERR_WORKSPACE if the workspace is too small. */
int64_t mi_linear_workspace_bytes(int32_t m, int32_t n, int32_t k);
int mi_linear(void* out, const void* x, const void* weight, const void* bias, int32_t m, int32_t n,
              int32_t k, int64_t x_row_stride, int64_t out_row_stride, void* workspace,
              int64_t workspace_bytes, int dtype, void* stream);
/* mi_linear with its split-K reduce INSIDE the launch (round 4): every slice writes its fp32 slab through to the
 * coherence point, and the slice of a (tile, row block) that arrives last — told by the value its one counter add
 * returned — sums the slabs in slice order (+ bias, one rounding): the bits of mi_linear, one launch fewer.  `counters`:
 * MI_LINEAR_COUNTER_BYTES of int32, zero before the FIRST launch; every launch leaves them zero (hipGraph replays need
 * nothing in between).  Plans without a split, with more than 8 slices or on the 256-row kernel run as mi_linear. */
#define MI_LINEAR_COUNTER_BYTES (16 << 10)
int mi_linear_fused_reduce(void* out, const void* x, const void* weight, const void* bias, int32_t m,
                           int32_t n, int32_t k, int64_t x_row_stride, int64_t out_row_stride,
                           void* workspace, int64_t workspace_bytes, void* counters,
                           int64_t counters_bytes, int dtype, void* stream);
/* The two halves of a split-K mi_linear, for callers that fold the reduction
 * into the next kernel: mi_linear_partials runs the streaming kernel only and
 * leaves fp32 partials [split][m][n] in `workspace` (*split_out = split >= 2;
 * MI_ERR_UNSUPPORTED when the shape's plan has no split — use mi_linear);
 * mi_linear_reduce sums them in slice order (+ bias) into the 16-bit output;
 * mi_fused_add_rms_norm_splitk does that sum inside the fused residual-add +
 * RMSNorm that follows o_proj / down_proj (qwen2_old.py:212-216): out =
 * rmsnorm(rnd(sum + bias) + residual) * weight, residual <- rnd(sum + bias) +
 * residual — bit-identical to mi_linear followed by mi_fused_add_rms_norm. */
int mi_linear_partials(const void* x, const void* weight, int32_t m, int32_t n, int32_t k,
                       int64_t x_row_stride, void* workspace, int64_t workspace_bytes,
                       int32_t* split_out, int dtype, void* stream);
int mi_linear_reduce(void* out, const void* partials, int32_t split, const void* bias, int32_t m,
                     int32_t n, int64_t out_row_stride, int dtype, void* stream);
int mi_fused_add_rms_norm_splitk(void* out, const void* partials, int32_t split, const void* bias,
                                 void* residual, const void* weight, float eps, int32_t num_tokens,
                                 int32_t hidden, int64_t out_token_stride,
                                 int64_t residual_token_stride, int dtype, void* stream);
/* gate_up_proj and the SiluAndMul behind it in one launch (Qwen2MLP.forward,
 * qwen2_old.py:65-68: act_fn(gate_up_proj(x))): weight [2 * inter, k] with the
 * gate rows first, then the up rows (MergedColumnParallelLinear layout);
 * out[m, inter] = silu(x . gate^T) * (x . up^T), each GEMM output rounded to
 * the 16-bit type before the activation, as the unfused pair rounds. */
int mi_linear_silu_mul(void* out, const void* x, const void* weight, int32_t m, int32_t inter,
                       int32_t k, int64_t x_row_stride, int64_t out_row_stride, int dtype,
                       void* stream);
/* Tile plans measured for a deployment's shapes (tools/tune_linear_plans.py ->
 * vllm_omni_amd/tuned/linear_plans_gfx950.json), replacing the built-in heuristic of the
 * three entry points above for 16-bit operands — the counterpart of the TunableOp
 * selections the library GEMMs of the same layers run on.  A plan is (rows per row
 * block: 64 / 128 / 256, weight rows per tile: 32 .. 256 in steps of 32 — for the
 * gated form 2 x the output columns of a tile —, split-K factor 1 .. 16) and applies
 * to row counts m_lo < m <= m_hi of the shape (n, k).  purpose: MI_LINEAR_PLAN_FULL
 * (mi_linear), _PARTIALS (mi_linear_partials; falls back to the FULL entry), _GATED
 * (mi_linear_silu_mul: n = inter).  A row block of fewer rows than m runs
 * ceil(m / rows) blocks per weight tile, dealt to one XCD so that the tile leaves
 * HBM once.  Host-side table, thread-safe; affects launches made after the call.
 * MI_ERR_INVALID_ARG for a malformed plan, MI_ERR_UNSUPPORTED for a tile shape that
 * is not built. */
enum { MI_LINEAR_PLAN_FULL = 0, MI_LINEAR_PLAN_PARTIALS = 1, MI_LINEAR_PLAN_GATED = 2 };
int mi_linear_plan_set(int32_t purpose, int32_t m_lo, int32_t m_hi, int32_t n, int32_t k,
                       int32_t rows_per_block, int32_t weight_rows_per_tile, int32_t split);
void mi_linear_plan_clear(void);
/* The plan a call would run: out[0..2] = rows per block, weight rows per tile, split. */
int mi_linear_plan_get(int32_t purpose, int32_t m, int32_t n, int32_t k, int32_t* out);


/* ------------------------------------------------------------------------
 * A12 — tensor-parallel all-reduce for decode-sized messages (SURVEY §8b export
 * list, §8e).  Replaces the all-reduce implicit in upstream `RowParallelLinear`
 * for o_proj / down_proj (vllm_omni/model_executor/models/qwen2_5_omni/
 * qwen2_old.py:54-60,118-124) on the group created at
 * vllm_omni/worker/gpu_ar_worker.py:69-75.  One process per GPU; two-shot
 * (reduce-scatter + all-gather) over IPC-mapped peer buffers, every rank
 * reading every peer directly (all xGMI links at once), fp32 accumulation in
 * RANK ORDER (bit-identical on every rank and run to run), one rounding.  The
 * launch is in-stream, carries no per-call value and replays from a hipGraph;
 * every spin is bounded (timeout -> mi_allreduce_status() != MI_OK, no hang).
 *   mi_allreduce_init             allocates this rank's shared region (flags +
 *       staging + tmp, uncached device memory sized for max_bytes messages) and
 *       writes its 64-byte IPC handle to ipc_handle_out.  The HOST exchanges
 *       the handles (any channel: torch.distributed all_gather, a file, ...).
 *   mi_allreduce_register_buffer  ipc_handles = world x 64 bytes in rank order
 *       (this rank's own entry is ignored): opens the peers' regions.  Collective
 *       in effect: every rank must have called init before any rank runs.
 *   mi_allreduce_run              out[count] = sum over ranks of in[count]
 *       (dtype bf16 / f16, count % 8 == 0, 16-byte aligned; out may alias in).
 *       split > 0: `in` is the fp32 split-K partials [split][count] of
 *       mi_linear_partials and the value entering the all-reduce is
 *       rnd(sum over slices in order (+ bias[col], cols = row length)) — what
 *       mi_linear_reduce would have stored — so a row-parallel projection needs
 *       no reduce launch of its own.  Every rank must issue the same sequence
 *       of calls (same count).  MI_ERR_WORKSPACE if the message exceeds max_bytes.
 *   mi_allreduce_status           host-side, synchronising: MI_ERR_STATE once any
 *       spin has timed out (sticky).
 *   mi_allreduce_poll             the same answer WITHOUT a device round trip: the
 *       kernel mirrors the status into a host-mapped pinned word; it covers the
 *       launches that had completed when the caller last synchronised (the runner
 *       polls right after the event wait of a step's device->host output copy).
 *       After a timeout the communicator is dead: mi_allreduce_run returns
 *       MI_ERR_STATE (this rank's call counters are out of step with its peers').
 *   mi_allreduce_destroy          closes the peer mappings, frees the region.
 * ---------------------------------------------------------------------- */
#define MI_ALLREDUCE_MAX_RANKS 8
#define MI_ALLREDUCE_HANDLE_BYTES 64
typedef struct mi_allreduce mi_allreduce;
int mi_allreduce_init(mi_allreduce** out, int32_t rank, int32_t world, int64_t max_bytes,
                      int32_t timeout_ms, void* ipc_handle_out);
int mi_allreduce_register_buffer(mi_allreduce* ar, const void* ipc_handles);
int mi_allreduce_run(mi_allreduce* ar, void* out, const void* in, int64_t count, int32_t split,
                     const void* bias, int64_t cols, int dtype, void* stream);
int mi_allreduce_status(mi_allreduce* ar);
int mi_allreduce_poll(mi_allreduce* ar);
int mi_allreduce_destroy(mi_allreduce* ar);

/* ------------------------------------------------------------------------
 * A1 — KV block pool (host side, no GPU).  The allocator underneath the
 * `KVCacheManager`-shaped façade the schedulers call
 * (vllm_omni/core/sched/omni_ar_scheduler.py:183-193,480,528,628;
 *  vllm_omni/core/sched/omni_generation_scheduler.py:49,100-110,167-177,210,256).
 * Free list is FIFO over block ids (blocks return to the TAIL in the order
 * they are freed), reference counted; block id 0 is reserved as the null
 * block and never handed out.  Thread-compatible: caller serialises.
 *   alloc: pops `n` ids into out_ids, or returns MI_ERR_EXHAUSTED and pops
 *          nothing (the façade turns that into `None`).
 *   ref:   +1 on an allocated block (prefix sharing), or on a FREE block that
 *          is still indexed by its content hash (a prefix-cache hit: the block
 *          leaves the free list).
 *   free:  -1; at zero the block goes back to the free list, keeping its hash.
 * Prefix cache (what upstream's `KVCacheManager.get_computed_blocks` /
 * `cache_blocks` do under `enable_prefix_caching`, reached from
 * omni_ar_scheduler.py:188 `super().schedule()`; consumed by
 * vllm_omni/core/prefix_cache.py:255-264 `block_table[req, :num_computed // b]`):
 *   cache_insert: index an ALLOCATED full block under a 128-bit content hash
 *          (first writer wins; a block's hash never changes while indexed).
 *   cache_lookup: ids of the longest indexed PREFIX of `n` hashes
 *          (hashes[2*i], hashes[2*i+1] = lo, hi); returns the hit count.
 *   alloc  evicts: popping a free block that still carries a hash drops the
 *          entry; `take_evicted` drains the ids (KV-cache events).
 *   reset_cache: drop every entry; MI_ERR_STATE while any block is in use.
 * ---------------------------------------------------------------------- */
typedef struct mi_blockpool mi_blockpool;
mi_blockpool* mi_blockpool_create(int32_t num_blocks);
void mi_blockpool_destroy(mi_blockpool* pool);
int32_t mi_blockpool_num_free(const mi_blockpool* pool);
int32_t mi_blockpool_num_blocks(const mi_blockpool* pool);
int mi_blockpool_alloc(mi_blockpool* pool, int32_t n, int32_t* out_ids);
int mi_blockpool_ref(mi_blockpool* pool, const int32_t* ids, int32_t n);
int mi_blockpool_free(mi_blockpool* pool, const int32_t* ids, int32_t n);
int32_t mi_blockpool_refcount(const mi_blockpool* pool, int32_t id);
int mi_blockpool_cache_insert(mi_blockpool* pool, int32_t id, uint64_t hash_lo, uint64_t hash_hi);
int32_t mi_blockpool_cache_lookup(const mi_blockpool* pool, const uint64_t* hashes, int32_t n,
                                  int32_t* out_ids);
int32_t mi_blockpool_num_cached(const mi_blockpool* pool);
int mi_blockpool_reset_cache(mi_blockpool* pool);
int32_t mi_blockpool_take_evicted(mi_blockpool* pool, int32_t* out_ids, int32_t capacity);

/* A3 — slot mapping on the host: slot[t] = block_table[req, pos/bs]*bs + pos%bs
 * for the tokens scheduled this step (vectorised restatement of upstream's
 * numpy path reached from vllm_omni/worker/gpu_ar_model_runner.py:437-442).
 * req_indices[T] / positions[T] are int32 / int64; out is int64 [T].
 * Returns MI_ERR_INVALID_ARG if a position indexes past block_table_stride. */
int mi_compute_slot_mapping(int64_t* slot_mapping, const int32_t* block_table,
                            int32_t block_table_stride, const int32_t* req_indices,
                            const int64_t* positions, int32_t num_tokens, int32_t block_size);

#ifdef __cplusplus
}
#endif
#endif /* VLLM_OMNI_AMD_H_ */
