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
};

// paged_attn_decode_mfma.hip
int launch_decode_mfma(const DecodeParams& p, int heads_per_wave, int head_dim, int dtype, bool nontemporal,
                       bool split_p, hipStream_t stream);

}  // namespace mi
