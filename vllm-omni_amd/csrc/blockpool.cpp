// A1 / A3 — host-side KV block pool and slot-mapping arithmetic (no GPU code).
//
// The allocator underneath the KVCacheManager-shaped façade that the reference's
// schedulers call (vllm_omni/core/sched/omni_ar_scheduler.py:183-193,480,528,628;
// vllm_omni/core/sched/omni_generation_scheduler.py:49,100-110,167-177,210,256).
// Upstream's allocator (vllm v0.19.0, not in the reference tree) is a doubly linked
// free queue of block objects; here it is an intrusive int32 linked list over a flat
// array sized once for the 288 GB budget (2.6e5 blocks = ~3 MB of host state), so
// alloc/free are O(n) in the ids touched with no per-block Python objects.
#include <stdint.h>

#include <new>
#include <vector>

#include "../../include/vllm_omni_amd.h"

struct mi_blockpool {
  int32_t num_blocks;
  int32_t num_free;
  int32_t head;  // first free id, -1 if none
  int32_t tail;  // last free id
  std::vector<int32_t> next;      // free-list link, -1 = end
  std::vector<int32_t> refcount;  // 0 = free (or the reserved null block)
};

extern "C" {

int mi_abi_version(void) { return MI_ABI_VERSION; }

const char* mi_error_string(int code) {
  switch (code) {
    case MI_OK: return "ok";
    case MI_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or misaligned stride)";
    case MI_ERR_UNSUPPORTED: return "unsupported shape/dtype for the gfx950 kernels";
    case MI_ERR_WORKSPACE: return "workspace missing or too small";
    case MI_ERR_LAUNCH: return "HIP kernel launch failed";
    case MI_ERR_EXHAUSTED: return "KV block pool exhausted";
    case MI_ERR_STATE: return "invalid block state (double free, unknown id)";
  }
  return "unknown error";
}

mi_blockpool* mi_blockpool_create(int32_t num_blocks) {
  if (num_blocks < 2) return nullptr;  // id 0 is the null block
  mi_blockpool* p = new (std::nothrow) mi_blockpool();
  if (!p) return nullptr;
  p->num_blocks = num_blocks;
  p->next.assign(num_blocks, -1);
  p->refcount.assign(num_blocks, 0);
  // free list 1, 2, ..., num_blocks-1 in id order
  for (int32_t i = 1; i + 1 < num_blocks; ++i) p->next[i] = i + 1;
  p->head = 1;
  p->tail = num_blocks - 1;
  p->num_free = num_blocks - 1;
  return p;
}

void mi_blockpool_destroy(mi_blockpool* pool) { delete pool; }

int32_t mi_blockpool_num_free(const mi_blockpool* pool) { return pool ? pool->num_free : 0; }
int32_t mi_blockpool_num_blocks(const mi_blockpool* pool) { return pool ? pool->num_blocks : 0; }

int mi_blockpool_alloc(mi_blockpool* pool, int32_t n, int32_t* out_ids) {
  if (!pool || n < 0 || (n > 0 && !out_ids)) return MI_ERR_INVALID_ARG;
  if (n > pool->num_free) return MI_ERR_EXHAUSTED;  // all-or-nothing
  for (int32_t i = 0; i < n; ++i) {
    const int32_t id = pool->head;
    pool->head = pool->next[id];
    pool->next[id] = -1;
    pool->refcount[id] = 1;
    out_ids[i] = id;
  }
  pool->num_free -= n;
  if (pool->num_free == 0) pool->tail = -1;
  return MI_OK;
}

int mi_blockpool_ref(mi_blockpool* pool, const int32_t* ids, int32_t n) {
  if (!pool || n < 0 || (n > 0 && !ids)) return MI_ERR_INVALID_ARG;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t id = ids[i];
    if (id <= 0 || id >= pool->num_blocks || pool->refcount[id] <= 0) return MI_ERR_STATE;
  }
  for (int32_t i = 0; i < n; ++i) pool->refcount[ids[i]]++;
  return MI_OK;
}

int mi_blockpool_free(mi_blockpool* pool, const int32_t* ids, int32_t n) {
  if (!pool || n < 0 || (n > 0 && !ids)) return MI_ERR_INVALID_ARG;
  // validate first so a bad id leaves the pool untouched
  for (int32_t i = 0; i < n; ++i) {
    const int32_t id = ids[i];
    if (id <= 0 || id >= pool->num_blocks || pool->refcount[id] <= 0) return MI_ERR_STATE;
  }
  for (int32_t i = 0; i < n; ++i) {
    const int32_t id = ids[i];
    if (pool->refcount[id] <= 0) return MI_ERR_STATE;  // same id twice in one call
    if (--pool->refcount[id] == 0) {
      pool->next[id] = -1;
      if (pool->tail >= 0) pool->next[pool->tail] = id; else pool->head = id;
      pool->tail = id;
      pool->num_free++;
    }
  }
  return MI_OK;
}

int32_t mi_blockpool_refcount(const mi_blockpool* pool, int32_t id) {
  if (!pool || id < 0 || id >= pool->num_blocks) return -1;
  return pool->refcount[id];
}

int mi_compute_slot_mapping(int64_t* slot_mapping, const int32_t* block_table,
                            int32_t block_table_stride, const int32_t* req_indices,
                            const int64_t* positions, int32_t num_tokens, int32_t block_size) {
  if (num_tokens == 0) return MI_OK;
  if (!slot_mapping || !block_table || !req_indices || !positions || num_tokens < 0 ||
      block_size <= 0 || block_table_stride <= 0)
    return MI_ERR_INVALID_ARG;
  for (int32_t t = 0; t < num_tokens; ++t) {
    const int64_t pos = positions[t];
    const int64_t bi = pos / block_size;
    if (pos < 0 || bi >= block_table_stride || req_indices[t] < 0) return MI_ERR_INVALID_ARG;
    const int64_t phys = block_table[(int64_t)req_indices[t] * block_table_stride + bi];
    slot_mapping[t] = phys * block_size + pos % block_size;
  }
  return MI_OK;
}

}  // extern "C"
