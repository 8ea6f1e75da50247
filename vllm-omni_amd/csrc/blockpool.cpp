// A1 / A3 — host-side KV block pool and slot-mapping arithmetic (no GPU code).
//
// The allocator underneath the KVCacheManager-shaped façade that the reference's
// schedulers call (vllm_omni/core/sched/omni_ar_scheduler.py:183-193,480,528,628;
// vllm_omni/core/sched/omni_generation_scheduler.py:49,100-110,167-177,210,256).
// Upstream's allocator (vllm v0.19.0, not in the reference tree) is a doubly linked
// free queue of block objects plus a hash -> block dictionary for prefix caching; here it
// is an intrusive doubly linked int32 list over flat arrays sized once for the 288 GB
// budget (2.6e5 blocks = a few MB of host state) and one hash map keyed by a 128-bit
// content hash, so alloc/free/ref are O(n) in the ids touched with no per-block Python
// objects.  Prefix-cache semantics follow upstream's: a freed block keeps its hash and
// waits at the TAIL of the free list; a hit on it (`ref`) takes it back out; `alloc` pops
// the HEAD (least recently freed) and only then drops that block's hash (eviction).
#include <stdint.h>

#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/vllm_omni_amd.h"

// Prefix-cache key: 128 bits (the façade chains blake2b-128 over parent hash + the block's tokens, as upstream
// chains its block hashes), so a collision that would alias two different prefixes is not a practical event.
struct BlockHash {
  uint64_t lo, hi;
  bool operator==(const BlockHash& o) const { return lo == o.lo && hi == o.hi; }
};
struct BlockHashHasher {
  size_t operator()(const BlockHash& h) const { return (size_t)(h.lo ^ (h.hi * 0x9E3779B97F4A7C15ull)); }
};

struct mi_blockpool {
  int32_t num_blocks;
  int32_t num_free;
  int32_t head;  // first free id (next eviction / allocation candidate), -1 if none
  int32_t tail;  // last free id
  std::vector<int32_t> next, prev;  // doubly linked free list (-1 = end): a cache hit unlinks a block from the middle
  std::vector<int32_t> refcount;    // 0 = free (or the reserved null block)
  std::vector<uint8_t> in_free;     // on the free list
  std::vector<uint8_t> hashed;      // has an entry in `cache`
  std::vector<BlockHash> hash;      // the block's key while `hashed`
  std::unordered_map<BlockHash, int32_t, BlockHashHasher> cache;  // full-block content hash -> block id
  std::vector<int32_t> evicted;     // ids whose cache entry was dropped since the last take_evicted

  void unlink(int32_t id) {
    const int32_t p = prev[id], n = next[id];
    if (p >= 0) next[p] = n; else head = n;
    if (n >= 0) prev[n] = p; else tail = p;
    prev[id] = next[id] = -1;
    in_free[id] = 0;
    num_free--;
  }
  void push_tail(int32_t id) {
    prev[id] = tail;
    next[id] = -1;
    if (tail >= 0) next[tail] = id; else head = id;
    tail = id;
    in_free[id] = 1;
    num_free++;
  }
  void drop_hash(int32_t id, bool record) {
    if (!hashed[id]) return;
    auto it = cache.find(hash[id]);
    if (it != cache.end() && it->second == id) cache.erase(it);
    hashed[id] = 0;
    if (record) evicted.push_back(id);
  }
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
  p->prev.assign(num_blocks, -1);
  p->refcount.assign(num_blocks, 0);
  p->in_free.assign(num_blocks, 0);
  p->hashed.assign(num_blocks, 0);
  p->hash.assign(num_blocks, BlockHash{0, 0});
  p->head = p->tail = -1;
  p->num_free = 0;
  for (int32_t i = 1; i < num_blocks; ++i) p->push_tail(i);  // free list 1, 2, ..., num_blocks-1 in id order
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
    pool->unlink(id);
    pool->drop_hash(id, true);  // a free block that still carried cached content is evicted now (LRU: list head)
    pool->refcount[id] = 1;
    out_ids[i] = id;
  }
  return MI_OK;
}

int mi_blockpool_ref(mi_blockpool* pool, const int32_t* ids, int32_t n) {
  if (!pool || n < 0 || (n > 0 && !ids)) return MI_ERR_INVALID_ARG;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t id = ids[i];
    if (id <= 0 || id >= pool->num_blocks) return MI_ERR_STATE;
    // referencing a FREE block is only legal while it is still indexed by its content hash (a prefix-cache hit)
    if (pool->refcount[id] <= 0 && !(pool->in_free[id] && pool->hashed[id])) return MI_ERR_STATE;
  }
  for (int32_t i = 0; i < n; ++i) {
    const int32_t id = ids[i];
    if (pool->in_free[id]) pool->unlink(id);  // revived: no longer an eviction candidate
    pool->refcount[id]++;
  }
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
    if (--pool->refcount[id] == 0) pool->push_tail(id);  // keeps its hash: reusable until evicted by alloc
  }
  return MI_OK;
}

int mi_blockpool_cache_insert(mi_blockpool* pool, int32_t id, uint64_t hash_lo, uint64_t hash_hi) {
  if (!pool) return MI_ERR_INVALID_ARG;
  if (id <= 0 || id >= pool->num_blocks || pool->refcount[id] <= 0) return MI_ERR_STATE;
  const BlockHash h{hash_lo, hash_hi};
  if (pool->hashed[id]) return pool->hash[id] == h ? MI_OK : MI_ERR_STATE;  // a full block never changes content
  // first writer wins: a second block with the same content stays un-indexed (and is simply freed later)
  if (pool->cache.emplace(h, id).second) {
    pool->hashed[id] = 1;
    pool->hash[id] = h;
  }
  return MI_OK;
}

int32_t mi_blockpool_cache_lookup(const mi_blockpool* pool, const uint64_t* hashes, int32_t n, int32_t* out_ids) {
  if (!pool || n < 0 || (n > 0 && (!hashes || !out_ids))) return MI_ERR_INVALID_ARG;
  int32_t hits = 0;
  for (; hits < n; ++hits) {
    auto it = pool->cache.find(BlockHash{hashes[2 * hits], hashes[2 * hits + 1]});
    if (it == pool->cache.end()) break;  // a prefix: stop at the first miss
    out_ids[hits] = it->second;
  }
  return hits;
}

int32_t mi_blockpool_num_cached(const mi_blockpool* pool) { return pool ? (int32_t)pool->cache.size() : 0; }

int mi_blockpool_reset_cache(mi_blockpool* pool) {
  if (!pool) return MI_ERR_INVALID_ARG;
  if (pool->num_free != pool->num_blocks - 1) return MI_ERR_STATE;  // blocks in use: upstream refuses too
  for (int32_t id = 1; id < pool->num_blocks; ++id) pool->drop_hash(id, true);
  return MI_OK;
}

int32_t mi_blockpool_take_evicted(mi_blockpool* pool, int32_t* out_ids, int32_t capacity) {
  if (!pool || capacity < 0 || (capacity > 0 && !out_ids)) return MI_ERR_INVALID_ARG;
  const int32_t n = (int32_t)pool->evicted.size() < capacity ? (int32_t)pool->evicted.size() : capacity;
  for (int32_t i = 0; i < n; ++i) out_ids[i] = pool->evicted[i];
  pool->evicted.erase(pool->evicted.begin(), pool->evicted.begin() + n);
  return n;
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
