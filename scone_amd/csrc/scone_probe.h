// Device-side probes of the f-gram index, shared by the match kernels (scone_index.hip) and the fused
// small-batch lookup kernel (scone_embed_wave.h).
#pragma once
#include "scone_common.h"

struct scone_index_view {
  const scone_slot *slots;
  unsigned long long mask;
  const int32_t *uni;  // direct unigram table or null
  int uni_cap;
  const uint32_t *bloom;  // presence bitmap or null
  unsigned long long bloom_mask;
  int max_n;
};

// ------------------------------------------------------------------ probe
struct scone_bucket_regs {
  ulonglong2 v[SCONE_BUCKET];
};

__device__ __forceinline__ void load_bucket(const scone_slot *__restrict__ slots, unsigned long long b, scone_bucket_regs &r) {
  const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(slots + (b << SCONE_BUCKET_SHIFT));
#pragma unroll
  for (int j = 0; j < SCONE_BUCKET; ++j) r.v[j] = p[j];  // 4 x 16 B, one 64-B sector, issued back to back
}

// id >= 0: found; -1: absent (an empty slot was seen); -2: bucket full without a match -> next bucket
__device__ __forceinline__ int32_t scan_bucket(const scone_bucket_regs &r, unsigned long long lo, uint32_t ext) {
#pragma unroll
  for (int j = 0; j < SCONE_BUCKET; ++j) {
    if (r.v[j].x == lo && (uint32_t)(r.v[j].y >> 32) == ext) return (int32_t)((uint32_t)r.v[j].y - 1u);
    if (r.v[j].x == 0ull) return -1;
  }
  return -2;
}

// Resolve one probe whose home bucket b has already been fetched into r.
__device__ __forceinline__ int32_t probe_finish(const scone_slot *__restrict__ slots, unsigned long long mask,
                                                unsigned long long lo, uint32_t ext, unsigned long long hash,
                                                unsigned long long b, scone_bucket_regs &r) {
  const unsigned long long nbm = mask >> SCONE_BUCKET_SHIFT, step = scone_bucket_step(hash);
  for (unsigned long long probe = 0; probe <= nbm; ++probe) {
    const int32_t id = scan_bucket(r, lo, ext);
    if (id != -2) return id;
    b = (b + step) & nbm;
    load_bucket(slots, b, r);
  }
  return -1;
}

__device__ __forceinline__ int32_t probe_index(const scone_slot *__restrict__ slots, unsigned long long mask,
                                               unsigned long long lo, uint32_t ext) {
  const unsigned long long hash = scone_hash_key(lo, ext);
  const unsigned long long b = scone_bucket_home(hash, mask);
  scone_bucket_regs r;
  load_bucket(slots, b, r);
  return probe_finish(slots, mask, lo, ext, hash, b, r);
}

// id of the f-gram k[0..n) or -1: unigram table, presence bitmap, then the hash table
__device__ __forceinline__ int32_t scone_lookup_key(const scone_index_view &ix, const uint32_t (&k)[SCONE_MAX_N], int n) {
  if (n == 1 && ix.uni && k[0] < (uint32_t)ix.uni_cap) return ix.uni[k[0]];
  const scone_key key = scone_pack_key(k, n, ix.max_n);
  if (!key.ok) return -1;
  const unsigned long long hash = scone_hash_key(key.lo, key.ext);
  if (ix.bloom) {
    const unsigned long long bit = scone_bloom_bit(hash, ix.bloom_mask);
    if (!((ix.bloom[bit >> 5] >> (bit & 31)) & 1u)) return -1;
  }
  return probe_index(ix.slots, ix.mask, key.lo, key.ext);
}
