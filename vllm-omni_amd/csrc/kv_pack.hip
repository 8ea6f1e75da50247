// N3 — KV extract for the stage hand-off: gather one request's K and V rows of EVERY layer
// out of the paged cache, keep only the KV heads the receiving TP rank owns, and lay them down in the wire
// order of the reference's packed payload — all in one launch.
//
// Replaces, for the raw-data connector path, the chain
//   OmniKVTransferManager._extract_kv_cache          (kv_transfer_manager.py:786-862: per layer
//       key_blocks[valid_ids].flatten(0, 1)[:seq_len].contiguous(), same for values)
//   -> _slice_transfer_data_for_target               (:498-519, utils/kv_utils.py:230-273: tensor[:, start:end])
//   -> KVCacheTransferData.to_gpu_tensor             (:144-157: every key layer, then every value layer,
//       each flattened to bytes, concatenated behind the header)
// which is ~6 torch launches and two intermediate copies per layer and cache.  Pure byte movement, HBM-bound:
// one 16-byte chunk per lane, source rows addressed through the block ids, destination contiguous.
#include "common.h"

namespace mi {
namespace {

struct KvPackParams {
  uint8_t* dst;                 // data region of the payload (16-byte aligned)
  const uint64_t* k_layers;     // [num_layers] device addresses of the K block views (0 = layer absent)
  const uint64_t* v_layers;
  const int64_t* k_offsets;     // [num_layers] byte offsets of each tensor inside the data region
  const int64_t* v_offsets;
  const int32_t* block_ids;     // [num_block_ids] validated on the host
  int32_t rows;                 // token rows to copy (<= num_block_ids * block_size)
  int32_t bs_log2;
  int32_t chunks_per_head;      // head_dim * elem_size / 16
  int32_t chunks_per_row;       // head_cnt * chunks_per_head
  int32_t head_lo;
  int64_t blk_bytes, tok_bytes, head_bytes;  // source strides in bytes
};

__global__ __launch_bounds__(256) void kv_pack_kernel(const KvPackParams p) {
  scalar_cache_acquire();
  const int layer = blockIdx.y >> 1;
  const bool is_v = blockIdx.y & 1;
  const uint64_t base = (is_v ? p.v_layers : p.k_layers)[layer];
  if (!base) return;  // whole workgroup: uniform
  const int64_t dst_off = (is_v ? p.v_offsets : p.k_offsets)[layer];
  const int64_t total = (int64_t)p.rows * p.chunks_per_row;
  const int bs_mask = (1 << p.bs_log2) - 1;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int row = (int)(idx / p.chunks_per_row);
    const int ch = (int)(idx - (int64_t)row * p.chunks_per_row);
    const int head = ch / p.chunks_per_head;
    const int within = ch - head * p.chunks_per_head;
    const int64_t src = (int64_t)p.block_ids[row >> p.bs_log2] * p.blk_bytes + (int64_t)(row & bs_mask) * p.tok_bytes +
                        (int64_t)(p.head_lo + head) * p.head_bytes + (int64_t)within * 16;
    const uint4 v = *reinterpret_cast<const uint4*>(base + src);
    *reinterpret_cast<uint4*>(p.dst + dst_off + idx * 16) = v;
  }
}

}  // namespace
}  // namespace mi

extern "C" int mi_kv_pack(void* dst, const uint64_t* k_layers, const uint64_t* v_layers, const int64_t* k_offsets,
                          const int64_t* v_offsets, int32_t num_layers, const int32_t* block_ids,
                          int32_t num_block_ids, int32_t rows, int32_t block_size, int32_t head_lo, int32_t head_cnt,
                          int32_t head_dim, int32_t elem_size, int64_t cache_block_stride,
                          int64_t cache_token_stride, int64_t cache_head_stride, void* stream) {
  using namespace mi;
  if (num_layers == 0 || rows == 0) return MI_OK;
  if (!dst || !k_layers || !v_layers || !k_offsets || !v_offsets || !block_ids) return MI_ERR_INVALID_ARG;
  if (num_layers < 0 || num_block_ids <= 0 || rows < 0 || head_lo < 0 || head_cnt <= 0 || head_dim <= 0)
    return MI_ERR_INVALID_ARG;
  if (!is_pow2(block_size)) return MI_ERR_UNSUPPORTED;
  if ((int64_t)rows > (int64_t)num_block_ids * block_size) return MI_ERR_INVALID_ARG;
  if (elem_size != 1 && elem_size != 2 && elem_size != 4) return MI_ERR_UNSUPPORTED;
  if ((head_dim * elem_size) % 16) return MI_ERR_UNSUPPORTED;  // a head row is moved in 16-byte chunks
  if (((uintptr_t)dst & 15) || ((cache_block_stride * elem_size) % 16) || ((cache_token_stride * elem_size) % 16) ||
      ((cache_head_stride * elem_size) % 16))
    return MI_ERR_INVALID_ARG;
  KvPackParams p;
  p.dst = reinterpret_cast<uint8_t*>(dst);
  p.k_layers = k_layers;
  p.v_layers = v_layers;
  p.k_offsets = k_offsets;
  p.v_offsets = v_offsets;
  p.block_ids = block_ids;
  p.rows = rows;
  p.bs_log2 = __builtin_ctz((unsigned)block_size);
  p.chunks_per_head = head_dim * elem_size / 16;
  p.chunks_per_row = head_cnt * p.chunks_per_head;
  p.head_lo = head_lo;
  p.blk_bytes = cache_block_stride * elem_size;
  p.tok_bytes = cache_token_stride * elem_size;
  p.head_bytes = cache_head_stride * elem_size;
  const int64_t total = (int64_t)rows * p.chunks_per_row;
  int64_t gx = (total + 255) / 256;
  if (gx > 2048) gx = 2048;  // grid-stride beyond that: 2 x layers x 2048 workgroups already fill the chip
  hipLaunchKernelGGL(kv_pack_kernel, dim3((unsigned)gx, (unsigned)(2 * num_layers)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  return launch_status();
}
