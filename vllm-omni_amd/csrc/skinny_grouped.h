// Grouped (per-expert) form of the decode-batch linear kernel, shared between skinny_gemm.hip (the kernel) and
// fused_moe.hip (its caller): every workgroup row (grid.y) is one tile of <= 64 token rows that all use ONE expert's
// weight matrix; rows are gathered through the MoE alignment's maps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

struct GroupedGemmArgs {
  void* out;                    // [max_rows, n_out]
  const void* a_rows;           // gated: x [T, k] (row = sorted_pairs[r] / top_k); else the gated output [max_rows, k]
  const void* weights;          // [E, (gated ? 2 : 1) * n_out, k]
  const int32_t* sorted_pairs;  // [max_rows]: pair index or -1 (padding)
  const int32_t* tile_expert;   // [max_tiles]
  const int32_t* num_tiles;     // [1], device
  int32_t max_tiles, gated, n_out, k, top_k, rows_per_tile;
  int64_t a_row_stride, out_row_stride, weight_expert_stride;
  int dtype;
};

// MI_OK, or MI_ERR_UNSUPPORTED when the shape is outside the ring kernel's envelope (the caller then runs its
// direct-from-global kernel).
int launch_grouped_skinny(const GroupedGemmArgs& a, hipStream_t s);

}  // namespace mi
