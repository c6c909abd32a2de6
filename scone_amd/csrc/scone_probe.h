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
__device__ __forceinline__ int32_t probe_index(const scone_slot *__restrict__ slots,
                                               unsigned long long mask, unsigned long long lo,
                                               uint32_t ext) {
  unsigned long long s = scone_hash_key(lo, ext) & mask;
  for (unsigned long long probe = 0; probe <= mask; ++probe) {
    // one 16-byte load per probe
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&slots[s]);
    if (v.x == lo && (uint32_t)(v.y >> 32) == ext) return (int32_t)((uint32_t)v.y - 1u);
    if (v.x == 0ull) return -1;
    s = (s + 1ull) & mask;
  }
  return -1;
}

// Resolve one probe whose first slot has already been fetched (v = slots[s]).
__device__ __forceinline__ int32_t probe_finish(const scone_slot *__restrict__ slots, unsigned long long mask,
                                                unsigned long long lo, uint32_t ext, unsigned long long s, ulonglong2 v) {
  for (unsigned long long probe = 0; probe <= mask; ++probe) {
    if (v.x == lo && (uint32_t)(v.y >> 32) == ext) return (int32_t)((uint32_t)v.y - 1u);
    if (v.x == 0ull) return -1;
    s = (s + 1ull) & mask;
    v = *reinterpret_cast<const ulonglong2 *>(&slots[s]);
  }
  return -1;
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
