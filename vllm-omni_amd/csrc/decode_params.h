// Launch parameters shared by the two paged-attention decode kernels and the merge kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

struct DecodeParams {
  void* out;
  const void* q;
  const void* k_cache;
  const void* v_cache;
  const int32_t* block_table;
  const int32_t* seq_lens;
  float* part_o;   // [num_seqs, n_q, num_parts, HD]
  float* part_ml;  // [num_seqs, n_q, num_parts, 2]
  // Fused merge (matrix-core kernel): arrival counters [num_seqs, n_q / heads_per_wave], zero before the FIRST launch
  // and left at zero by every launch; nullptr = the partitions are merged by decode_reduce_kernel instead.
  int32_t* part_count;
  int32_t num_seqs, num_q_heads, kv_group;  // kv_group = n_q / n_kv
  int32_t bs_log2, block_table_stride;
  int32_t num_parts, part_tiles;  // tiles (16 tokens) per partition
  int32_t total_waves;
  int32_t out_f32;
  float scale_log2e;
  int64_t q_stride, out_stride, blk_stride, tok_stride, head_stride;
  // Fused RoPE + KV write (mi_rope_cache_paged_attn_decode; matrix-core kernel only): q arrives un-rotated and is
  // rotated in registers; the wave that owns a sequence's LAST context partition rotates the token's k, writes it and
  // v to cache row slot_mapping[seq], and only then starts streaming.  nullptr = q is rotated, the row is in the cache.
  const void* k_new = nullptr;       // [num_seqs, kn_stride] un-rotated k of this step's token
  const void* v_new = nullptr;
  const int64_t* positions = nullptr;      // [num_seqs]
  const int64_t* slot_mapping = nullptr;   // [num_seqs], < 0 = no write
  const float* cos_sin = nullptr;          // [max_position, head_dim] = cos | sin
  int32_t max_position = 0;
  int64_t kn_stride = 0, vn_stride = 0;
  // ... with the per-head q / k RMSNorm of the Qwen3 family in front of the rotation (both or neither; weights [head_dim])
  const void* q_norm_w = nullptr;
  const void* k_norm_w = nullptr;
  float norm_eps = 0.f;
};

// paged_attn_decode_mfma.hip
int launch_decode_mfma(const DecodeParams& p, int heads_per_wave, int head_dim, int dtype, bool nontemporal,
                       bool split_p, hipStream_t stream);
// the same launch with RoPE + KV write inside (p.k_new etc. set); nontemporal + split-P form only
int launch_decode_mfma_rope(const DecodeParams& p, int heads_per_wave, int head_dim, int dtype, hipStream_t stream);

}  // namespace mi
