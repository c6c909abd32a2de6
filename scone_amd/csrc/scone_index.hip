// f-gram index (exact-key open-addressing hash table in HBM), the per-window match
// kernel and the CSR expansion of the reference's per-position id lists.
//
// Replaces, on the GPU:
//   NGramExtractor.f_grams / f_gram_to_id            scone/tokenization/n_gram_extractor.py:42-44
//   NGramExtractor.get_token_f_grams                 scone/tokenization/n_gram_extractor.py:106-126
//   [f_gram_to_id[g] for g in f_grams]               scone/inference/embedding_cache.py:173
#include "scone_common.h"
#include "scone_probe.h"

namespace {

// ------------------------------------------------------------------ build
// Lock-free insert.  A slot is identified by (lo, ext); lo is claimed with a
// 64-bit CAS, then (ext,id) is published with a second CAS on hi.  Any thread
// whose lo matches may publish hi first -- the slot then simply belongs to that
// key and the others move on -- so nobody ever waits on another thread.
// Duplicate keys: atomicMin keeps the smallest id.
__global__ __launch_bounds__(256) void k_index_insert(scone_slot *__restrict__ slots,
                                                      unsigned long long mask,
                                                      const uint32_t *__restrict__ keys,
                                                      const uint8_t *__restrict__ lens,
                                                      unsigned long long n, unsigned long long id0,
                                                      int max_n,
                                                      unsigned long long *__restrict__ counters,
                                                      uint32_t *__restrict__ status, int32_t *__restrict__ uni,
                                                      int uni_cap, uint32_t *__restrict__ bloom,
                                                      unsigned long long bloom_mask) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int len = lens[i];
  if (len < 1 || len > max_n) {
    atomicOr(status, SCONE_ST_BAD_TOKEN);
    return;
  }
  uint32_t t[SCONE_MAX_N];
  for (int k = 0; k < SCONE_MAX_N; ++k) t[k] = k < len ? keys[i * max_n + k] : 0u;
  scone_key key = scone_pack_key(t, len, max_n);
  if (!key.ok) {
    atomicOr(status, SCONE_ST_BAD_TOKEN);
    return;
  }
  // unigrams are also kept in a direct table (token -> smallest id), read by the fused match
  if (len == 1 && uni && t[0] < (uint32_t)uni_cap) atomicMin(reinterpret_cast<unsigned int *>(&uni[t[0]]), (unsigned int)(id0 + i));
  unsigned long long myhi = ((unsigned long long)key.ext << 32) | (unsigned long long)(uint32_t)(id0 + i + 1ull);
  const unsigned long long hash = scone_hash_key(key.lo, key.ext);
  if (bloom) {
    const unsigned long long bit = scone_bloom_bit(hash, bloom_mask);
    atomicOr(&bloom[bit >> 5], 1u << (bit & 31));
  }
  // bucket sequence of scone_common.h: slots of a bucket front to back, then the next bucket
  const unsigned long long nbm = mask >> SCONE_BUCKET_SHIFT, step = scone_bucket_step(hash);
  unsigned long long b = scone_bucket_home(hash, mask);
  for (unsigned long long probe = 0; probe <= nbm; ++probe) {
    for (int j = 0; j < SCONE_BUCKET; ++j) {
      const unsigned long long s = (b << SCONE_BUCKET_SHIFT) + j;
      unsigned long long old = atomicCAS(&slots[s].lo, 0ull, key.lo);
      if (old == 0ull || old == key.lo) {
        unsigned long long prev = atomicCAS(&slots[s].hi, 0ull, myhi);
        if (prev == 0ull) {
          atomicAdd(&counters[0], 1ull);
          return;
        }
        if ((uint32_t)(prev >> 32) == key.ext) {
          atomicMin(&slots[s].hi, myhi);
          atomicAdd(&counters[1], 1ull);
          return;
        }
      }
    }
    b = (b + step) & nbm;
  }
  atomicOr(status, SCONE_ST_INDEX_FULL);
}

// One thread per (n, position): hits[(n-1)*BT + p] = id of tok[p .. p+n-1] or -1.
// Windows never cross a sequence boundary (callers pass one sequence at a time,
// f_gram_tokenizer.py:77) and must fit in T (n_gram_extractor.py:118).
__global__ __launch_bounds__(256) void k_match(const scone_slot *__restrict__ slots,
                                               unsigned long long mask,
                                               const int32_t *__restrict__ tok, long long BT, int T,
                                               int max_n, int32_t *__restrict__ hits) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= BT * max_n) return;
  int n = (int)(gid / BT) + 1;
  long long p = gid - (long long)(n - 1) * BT;
  int i = (int)(p % T);
  int32_t res = -1;
  if (i + n <= T) {
    uint32_t t[SCONE_MAX_N] = {0u, 0u, 0u, 0u};
    bool ok = true;
    for (int k = 0; k < SCONE_MAX_N; ++k) {
      if (k < n) {
        int32_t v = tok[p + k];
        ok = ok && v >= 0;
        t[k] = (uint32_t)v;
      }
    }
    if (ok) {
      scone_key key = scone_pack_key(t, n, max_n);
      if (key.ok) res = probe_index(slots, mask, key.lo, key.ext);
    }
  }
  hits[gid] = res;
}

// ------------------------------------------------------------------ match -> per-token lists
// The form the fused lookup consumes: for every position a fixed-width record
//   ell[p*W + 0 .. K_own-1] = ids of the f-grams covering p whose rows THIS handle owns,
//                             in the reference's order (n ascending, start ascending, duplicates kept)
//   ell[p*W + W-2]          = K_own | (K_full << 8)      (K_full: all hits, the mean's divisor)
// W = 8 (max_n <= 3, <= 6 ids) or 16 (max_n = 4, <= 10 ids), so the gather kernel fetches a
// token's whole list with ONE aligned scalar load.  A workgroup owns ELL_TILE consecutive
// positions: thread t forms the keys of the windows (n = 1..MAXN) starting at its position, the
// probes that survive the presence bitmap are queued in LDS and resolved a bucket per quad, the
// per-window ids are staged in LDS, then every thread compacts the candidates covering its
// position (the first max_n-1 threads only supply the window starts in front of the tile).
#define ELL_TILE 256

// A pending table probe, staged in LDS between the two phases of the match.
struct __attribute__((aligned(16))) probe_req {
  unsigned long long lo;
  uint32_t ext;
  uint32_t dest;  // (n - 1) * ELL_TILE + thread: where the id goes in win[][]
};

// Phase 1, one thread per window start: tokens -> keys; unigrams resolve through the direct table, every
// other window is first tested against the presence bitmap (a clear bit proves a miss; the bitmap is
// small enough to live in L2) and, if it survives, queued for phase 2.
template <int MAXN>
__device__ __forceinline__ void stage_starts(const int32_t *__restrict__ uni, int uni_cap, const uint32_t *__restrict__ bloom,
                                             unsigned long long bloom_mask, const int32_t *__restrict__ tok, long long BT,
                                             int T, int max_n, long long start, int t, int32_t (*win)[ELL_TILE],
                                             probe_req *queue, uint32_t *q_count) {
  int32_t res[MAXN];
#pragma unroll
  for (int n = 0; n < MAXN; ++n) res[n] = -1;
  if (start >= 0 && start < BT) {
    const int i = BT <= 0x7FFFFFFFll ? (int)((unsigned)start % (unsigned)T) : (int)(start % T);
    uint32_t k[SCONE_MAX_N] = {0u, 0u, 0u, 0u};
    int nvalid = 0;  // longest window starting here that fits in the sequence and has only valid tokens
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      if (j < max_n && i + j < T && nvalid == j) {
        const int32_t v = tok[start + j];
        if (v >= 0) {
          k[j] = (uint32_t)v;
          nvalid = j + 1;
        }
      }
    }
#pragma unroll
    for (int n = 1; n <= MAXN; ++n) {
      if (n > nvalid) continue;
      if (n == 1 && uni && k[0] < (uint32_t)uni_cap) {
        res[0] = uni[k[0]];
        continue;
      }
      const scone_key key = scone_pack_key(k, n, max_n);
      if (!key.ok) continue;
      bool live = true;
      if (bloom) {
        const unsigned long long bit = scone_bloom_bit(scone_hash_key(key.lo, key.ext), bloom_mask);
        live = (bloom[bit >> 5] >> (bit & 31)) & 1u;
      }
      if (live) {
        probe_req rq;
        rq.lo = key.lo, rq.ext = key.ext, rq.dest = (uint32_t)((n - 1) * ELL_TILE + t);
        queue[atomicAdd(q_count, 1u)] = rq;
      }
    }
  }
#pragma unroll
  for (int n = 0; n < MAXN; ++n) win[n][t] = res[n];
}

// Phase 2: the queued probes are resolved by QUADS of lanes -- lane j of a quad reads slot j of the bucket,
// so ONE wave load instruction serves 16 probes and every quad touches one whole 64-B sector (per-lane
// probing spends 4 instructions x 64 scattered addresses on every 64 bucket reads).  A wavefront ballot
// gives every quad its 4 match / empty bits; the bucket sequence is the one of scone_common.h.
// Measured (headline workload, one box): same kernel time as per-lane probing with three buckets in
// flight per lane (41.8 vs 42.3 us) at 27 instead of 70+ VGPRs.  Ablation of the 42 us: token loads +
// key hashing 4 us, record stores 9 us, unigram table + bitmap 1 us, table probes 28 us -- ~0.8M probes
// per 1M tokens, each moving a 128-B line of the 32 MB table from the Infinity Cache into one XCD's L2.
__device__ __forceinline__ void resolve_queue(const scone_slot *__restrict__ slots, unsigned long long mask, int t,
                                              int32_t *win_flat, const probe_req *queue, uint32_t n_req) {
  const unsigned long long nbm = mask >> SCONE_BUCKET_SHIFT;
  const int quad = t >> 2, ql = t & 3, qshift = (t & 63) & ~3;
  for (uint32_t base = 0; base < n_req; base += ELL_TILE / SCONE_BUCKET) {
    const uint32_t q = base + (uint32_t)quad;
    bool active = q < n_req;
    unsigned long long lo = 0, b = 0, step = 1, tries = 0;
    uint32_t ext = 0, dest = 0;
    if (active) {
      const probe_req rq = queue[q];
      lo = rq.lo, ext = rq.ext, dest = rq.dest;
      const unsigned long long hash = scone_hash_key(lo, ext);
      b = scone_bucket_home(hash, mask), step = scone_bucket_step(hash);
    }
    while (__ballot(active)) {
      ulonglong2 v = make_ulonglong2(1ull, 0ull);
      if (active) v = *reinterpret_cast<const ulonglong2 *>(&slots[(b << SCONE_BUCKET_SHIFT) + ql]);
      const bool m = active && v.x == lo && (uint32_t)(v.y >> 32) == ext;
      const bool e = active && v.x == 0ull;
      const uint32_t qm = (uint32_t)(__ballot(m) >> qshift) & 0xFu, qe = (uint32_t)(__ballot(e) >> qshift) & 0xFu;
      if (active) {
        if (qm) {  // slots fill front to back and keys are unique: at most one lane matches
          if (m) win_flat[dest] = (int32_t)((uint32_t)v.y - 1u);
          active = false;
        } else if (qe || ++tries > nbm) {
          active = false;  // an empty slot proves the key absent (win stays -1)
        } else {
          b = (b + step) & nbm;
        }
      }
    }
  }
}

template <int MAXN>
__global__ __launch_bounds__(ELL_TILE) void k_match_ell(const scone_slot *__restrict__ slots, unsigned long long mask,
                                                        const int32_t *__restrict__ uni, int uni_cap,
                                                        const uint32_t *__restrict__ bloom, unsigned long long bloom_mask,
                                                        const int32_t *__restrict__ tok, long long BT, int T, int max_n,
                                                        long long row_begin, long long row_end, int mode,
                                                        int keep_pos, int32_t *__restrict__ ell, int tile) {
  constexpr int HALO = MAXN - 1;
  constexpr int W = MAXN <= 3 ? 8 : 16;
  // `tile` <= ELL_TILE - HALO positions per workgroup (the launcher sizes it so that the grid is a whole number of
  // residency rounds); threads past tile + HALO only keep the barriers company
  __shared__ int32_t win[MAXN][ELL_TILE];
  __shared__ probe_req queue[MAXN * ELL_TILE];  // worst case: every window survives the bitmap
  __shared__ uint32_t q_count;
  const int t = threadIdx.x;
  // Thread t stages the windows that START at position p = tile0 - HALO + t and later compacts the
  // candidates covering p; the first HALO threads only supply the starts in front of the tile (one pass
  // per thread: a second, 2-lane pass for the halo would double wave 0's dependent-load chain and with
  // it, through the barrier, the whole workgroup's).
  const long long p = (long long)blockIdx.x * tile - HALO + t;

  if (t == 0) q_count = 0;
  __syncthreads();
  stage_starts<MAXN>(uni, uni_cap, bloom, bloom_mask, tok, BT, T, max_n, t < tile + HALO ? p : -1ll, t, win, queue, &q_count);
  __syncthreads();
  resolve_queue(slots, mask, t, &win[0][0], queue, q_count);
  __syncthreads();

  if (t < HALO || t >= tile + HALO || p >= BT) return;
  const int i = BT <= 0x7FFFFFFFll ? (int)((unsigned)p % (unsigned)T) : (int)(p % T);
  int32_t rec[W];
#pragma unroll
  for (int j = 0; j < W; ++j) rec[j] = -1;
  int kown = 0, kfull = 0;
  if (mode == SCONE_MODE_LONGEST_SUFFIX) {
    // paper, Algorithm 2: j = smallest j' < i with (sigma_j' .. sigma_i) an f-gram -> the LONGEST
    // f-gram of length >= 2 that ENDS at this position (window start p - (n-1))
#pragma unroll
    for (int nn = MAXN; nn >= 2; --nn) {
      if (kfull == 0 && nn <= max_n && i - (nn - 1) >= 0) {
        const int32_t id = win[nn - 1][t - (nn - 1)];
        if (id >= 0) {
          kfull = 1;
          if (id >= row_begin && id < row_end) rec[0] = id, kown = 1;
        }
      }
    }
  }
#pragma unroll
  for (int nn = 1; nn <= MAXN; ++nn) {
#pragma unroll
    for (int s = nn - 1; s >= 0; --s) {
      if (mode == SCONE_MODE_COVER && nn <= max_n && i - s >= 0) {
        const int32_t id = win[nn - 1][t - s];
        if (id >= 0) {
          // keep_pos (row exchange between shards): an owned id stays at its index in the FULL list,
          // ids of other shards leave a hole (-1); otherwise owned ids are compacted
          const int at = keep_pos ? kfull : kown;
          ++kfull;
          if (id >= row_begin && id < row_end) {
#pragma unroll
            for (int j = 0; j < W - 2; ++j)
              if (j == at) rec[j] = id;
            ++kown;
          }
        }
      }
    }
  }
  rec[W - 2] = kown | (kfull << 8);
  rec[W - 1] = 0;
  int4 *dst = reinterpret_cast<int4 *>(ell + p * W);
#pragma unroll
  for (int j = 0; j < W / 4; ++j) dst[j] = make_int4(rec[4 * j], rec[4 * j + 1], rec[4 * j + 2], rec[4 * j + 3]);
}

// ------------------------------------------------------------------ CSR
// Candidate c of position j enumerates (n, s) with n = 1..max_n, s = n-1..0
// (window start i = j - s ascending): the append order of n_gram_extractor.py:119-124.
__device__ __forceinline__ int32_t candidate(const int32_t *__restrict__ hits, long long BT,
                                             long long p, int i, int n, int s) {
  if (i - s < 0) return -1;
  return hits[(long long)(n - 1) * BT + p - s];
}

__device__ __forceinline__ int count_pos(const int32_t *__restrict__ hits, long long BT, long long p,
                                         int T, int max_n) {
  int i = (int)(p % T);
  int c = 0;
  for (int n = 1; n <= max_n; ++n)
    for (int s = n - 1; s >= 0; --s) c += candidate(hits, BT, p, i, n, s) >= 0;
  return c;
}

#define CSR_BLOCK 256
#define CSR_ITEMS 4
#define CSR_TILE (CSR_BLOCK * CSR_ITEMS)

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
  for (int d = 1; d < 64; d <<= 1) {
    int u = __shfl_up(v, d, 64);
    if (lane >= d) v += u;
  }
  return v;
}

// exclusive scan of one value per thread over the block; returns the exclusive
// prefix and the block total in *total
__device__ __forceinline__ int block_excl_scan(int v, int *total, int *smem /*[CSR_BLOCK/64+1]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_incl_scan(v);
  if (lane == 63) smem[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int k = 0; k < CSR_BLOCK / 64; ++k) {
    int x = smem[k];
    if (k < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(CSR_BLOCK) void k_csr_block_sums(const int32_t *__restrict__ hits,
                                                              long long BT, int T, int max_n,
                                                              int32_t *__restrict__ block_sums) {
  __shared__ int smem[CSR_BLOCK / 64 + 1];
  long long base = (long long)blockIdx.x * CSR_TILE + (long long)threadIdx.x * CSR_ITEMS;
  int c = 0;
  for (int k = 0; k < CSR_ITEMS; ++k)
    if (base + k < BT) c += count_pos(hits, BT, base + k, T, max_n);
  int tot;
  block_excl_scan(c, &tot, smem);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums in place; grand total -> *total
__global__ __launch_bounds__(CSR_BLOCK) void k_csr_scan_sums(int32_t *__restrict__ block_sums,
                                                             long long nb,
                                                             long long *__restrict__ total) {
  __shared__ int smem[CSR_BLOCK / 64 + 1];
  long long carry = 0;
  for (long long start = 0; start < nb; start += CSR_BLOCK) {
    long long idx = start + threadIdx.x;
    int v = idx < nb ? block_sums[idx] : 0;
    int tot;
    int ex = block_excl_scan(v, &tot, smem);
    if (idx < nb) block_sums[idx] = (int32_t)(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(CSR_BLOCK) void k_csr_fill(const int32_t *__restrict__ hits, long long BT,
                                                        int T, int max_n,
                                                        const int32_t *__restrict__ block_sums,
                                                        int32_t *__restrict__ offsets,
                                                        int32_t *__restrict__ ids, long long ids_cap) {
  __shared__ int smem[CSR_BLOCK / 64 + 1];
  long long base = (long long)blockIdx.x * CSR_TILE + (long long)threadIdx.x * CSR_ITEMS;
  int cnt[CSR_ITEMS];
  int c = 0;
  for (int k = 0; k < CSR_ITEMS; ++k) {
    cnt[k] = base + k < BT ? count_pos(hits, BT, base + k, T, max_n) : 0;
    c += cnt[k];
  }
  int tot;
  long long off = (long long)block_sums[blockIdx.x] + block_excl_scan(c, &tot, smem);
  for (int k = 0; k < CSR_ITEMS; ++k) {
    long long p = base + k;
    if (p >= BT) break;
    offsets[p] = (int32_t)off;
    int i = (int)(p % T);
    long long w = off;
    for (int n = 1; n <= max_n; ++n)
      for (int s = n - 1; s >= 0; --s) {
        int32_t id = candidate(hits, BT, p, i, n, s);
        if (id >= 0) {
          if (w < ids_cap) ids[w] = id;
          ++w;
        }
      }
    off += cnt[k];
    if (p == BT - 1) offsets[BT] = (int32_t)off;
  }
}

}  // namespace

// ------------------------------------------------------------------ host side
int scone_launch_match(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_hits,
                       hipStream_t s) {
  long long BT = (long long)B * T;
  long long total = BT * h->cfg.max_n;
  if (total == 0) return SCONE_OK;
  long long blocks = (total + 255) / 256;
  if (!scone_grid_fits((unsigned long long)blocks, 256)) return scone_fail(h, SCONE_EINVAL, "scone_match: too many tokens for one launch");
  hipLaunchKernelGGL(k_match, dim3((unsigned)blocks), dim3(256), 0, s, h->slots, h->cap - 1, d_tok, BT, T,
                     h->cfg.max_n, d_hits);
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

void scone_index_view_of(const scone_handle *h, scone_index_view *v) {
  v->slots = h->slots, v->mask = h->cap - 1, v->uni = h->d_uni, v->uni_cap = SCONE_UNI_CAP;
  v->bloom = h->d_bloom, v->bloom_mask = h->bloom_mask, v->max_n = h->cfg.max_n;
}

int scone_launch_match_ell(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_ell, hipStream_t s) {
  return scone_launch_match_ell_ex(h, d_tok, B, T, d_ell, (long long)h->cfg.row_begin, (long long)h->cfg.row_end, 0, s);
}

int scone_launch_match_ell_ex(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_ell, long long rb,
                              long long re, int keep_pos, hipStream_t s) {
  const long long BT = (long long)B * T;
  if (BT == 0) return SCONE_OK;
  // Positions per workgroup: at most ELL_TILE - halo, and such that the grid is a whole number of residency rounds (a
  // workgroup is 4 waves of 27 VGPRs and 15 KB of LDS: 8 fit a CU).  ceil(BT / 254) workgroups at 2048 x 512 tokens is
  // 4129 = two rounds of 2048 and a third that is nearly empty (profiles/r02g/match_tail.md); smaller tiles, one more
  // FULL round.  SCONE_MATCH_TILE=<n> fixes the tile (A/B experiments).
  const long long tile_max = ELL_TILE - (h->cfg.max_n <= 3 ? 2 : 3);
  const long long resident = (long long)h->n_cus * 8;
  long long blocks = (BT + tile_max - 1) / tile_max;
  long long tile = tile_max;
  if (blocks > resident) {
    const long long rounds = (blocks + resident - 1) / resident;
    tile = (BT + rounds * resident - 1) / (rounds * resident);
    if (tile > tile_max) tile = tile_max;
  }
  if (h->match_tile > 0 && h->match_tile <= tile_max) tile = h->match_tile;
  blocks = (BT + tile - 1) / tile;
  if (!scone_grid_fits((unsigned long long)blocks, ELL_TILE)) return scone_fail(h, SCONE_EINVAL, "scone_embed: too many tokens for one launch");
  if (h->cfg.max_n <= 3)
    hipLaunchKernelGGL((k_match_ell<3>), dim3((unsigned)blocks), dim3(ELL_TILE), 0, s, h->slots, h->cap - 1, h->d_uni,
                       SCONE_UNI_CAP, h->d_bloom, h->bloom_mask, d_tok, BT, T, h->cfg.max_n, rb, re, (int)h->cfg.lookup_mode, keep_pos, d_ell, (int)tile);
  else
    hipLaunchKernelGGL((k_match_ell<4>), dim3((unsigned)blocks), dim3(ELL_TILE), 0, s, h->slots, h->cap - 1, h->d_uni,
                       SCONE_UNI_CAP, h->d_bloom, h->bloom_mask, d_tok, BT, T, h->cfg.max_n, rb, re, (int)h->cfg.lookup_mode, keep_pos, d_ell, (int)tile);
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

// id records matched before the index changed are stale: the staging pipeline of a pinned-host table is dropped with
// whatever it prepared ahead (its records hold cache slots of rows found through the old index).  Like every mutation: not
// concurrent with lookups.
static void index_modified(scone_handle *h) {
  if (h->stage) scone_stage_destroy(h);
}

extern "C" int scone_index_build_device(scone_handle *h, const uint32_t *d_keys, const uint8_t *d_lens,
                                        uint64_t n, uint64_t id0, scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (n == 0) return SCONE_OK;
  if (!d_keys || !d_lens) return scone_fail(h, SCONE_EINVAL, "scone_index_build: null keys/lens");
  if (id0 + n > 0xFFFFFFFEull) return scone_fail(h, SCONE_ERANGE, "scone_index_build: ids must be < 2^32-2");
  SCONE_ON_DEVICE(h);
  index_modified(h);  // (on the handle's device: it frees the pipeline's streams and buffers)
  unsigned long long blocks = (n + 255) / 256;
  if (!scone_grid_fits(blocks, 256)) return scone_fail(h, SCONE_EINVAL, "scone_index_build: chunk too large");
  hipLaunchKernelGGL(k_index_insert, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, h->slots,
                     h->cap - 1, d_keys, d_lens, (unsigned long long)n, (unsigned long long)id0, h->cfg.max_n,
                     h->d_counters, h->d_status, h->d_uni, SCONE_UNI_CAP, h->d_bloom, h->bloom_mask);
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

extern "C" int scone_index_build(scone_handle *h, const uint32_t *h_keys, const uint8_t *h_lens, uint64_t n,
                                 uint64_t id0) {
  if (!h) return SCONE_EINVAL;
  if (n == 0) return SCONE_OK;
  if (!h_keys || !h_lens) return scone_fail(h, SCONE_EINVAL, "scone_index_build: null keys/lens");
  SCONE_ON_DEVICE(h);
  index_modified(h);
  const uint64_t chunk = 1ull << 22;  // keys per staging round
  const int max_n = h->cfg.max_n;
  uint32_t *d_keys = nullptr;
  uint8_t *d_lens = nullptr;
  uint64_t cn = n < chunk ? n : chunk;
  SCONE_HIP(h, hipMalloc(&d_keys, cn * max_n * sizeof(uint32_t)));
  hipError_t e = hipMalloc(&d_lens, cn);
  if (e != hipSuccess) {
    (void)hipFree(d_keys);
    return scone_hip_fail(h, e, "hipMalloc(lens staging)");
  }
  int rc = SCONE_OK;
  for (uint64_t off = 0; off < n && rc == SCONE_OK; off += chunk) {
    uint64_t m = n - off < chunk ? n - off : chunk;
    e = hipMemcpy(d_keys, h_keys + off * max_n, m * max_n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_lens, h_lens + off, m, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      rc = scone_hip_fail(h, e, "hipMemcpy(keys staging)");
      break;
    }
    rc = scone_index_build_device(h, d_keys, d_lens, m, id0 + off, nullptr);
    if (rc == SCONE_OK) {
      e = hipStreamSynchronize(nullptr);
      if (e != hipSuccess) rc = scone_hip_fail(h, e, "hipStreamSynchronize(index build)");
    }
  }
  (void)hipFree(d_keys);
  (void)hipFree(d_lens);
  if (rc != SCONE_OK) return rc;
  uint32_t bits = 0;
  SCONE_HIP(h, hipMemcpy(&bits, h->d_status, sizeof(bits), hipMemcpyDeviceToHost));
  if (bits & SCONE_ST_INDEX_FULL) return scone_fail(h, SCONE_ENOMEM, "scone_index_build: index full (raise index_capacity)");
  if (bits & SCONE_ST_BAD_TOKEN)
    return scone_fail(h, SCONE_ERANGE, "scone_index_build: key length outside 1..max_n or token id not representable");
  return SCONE_OK;
}

extern "C" int scone_index_stats(scone_handle *h, uint64_t *n_keys, uint64_t *capacity, uint64_t *n_dups) {
  if (!h) return SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipDeviceSynchronize());
  unsigned long long c[2] = {0, 0};
  SCONE_HIP(h, hipMemcpy(c, h->d_counters, sizeof(c), hipMemcpyDeviceToHost));
  if (n_keys) *n_keys = c[0];
  if (n_dups) *n_dups = c[1];
  if (capacity) *capacity = h->cap;
  return SCONE_OK;
}

// The built index as three host blobs (hash slots, direct unigram table, presence bitmap): a table file that carries
// them restores a 1e9-key index with three copies instead of a ~16 s rebuild from the keys.
extern "C" int scone_index_blob_sizes(scone_handle *h, uint64_t *slot_bytes, uint64_t *uni_bytes, uint64_t *bloom_bytes) {
  if (!h || !slot_bytes || !uni_bytes || !bloom_bytes) return SCONE_EINVAL;
  *slot_bytes = h->cap * sizeof(scone_slot);
  *uni_bytes = (uint64_t)SCONE_UNI_CAP * sizeof(int32_t);
  *bloom_bytes = (h->bloom_mask + 1) / 8;
  return SCONE_OK;
}

extern "C" int scone_index_export(scone_handle *h, void *h_slots, void *h_uni, void *h_bloom, uint64_t *n_keys) {
  if (!h || !h_slots || !h_uni || !h_bloom) return h ? scone_fail(h, SCONE_EINVAL, "scone_index_export: null pointer") : SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipDeviceSynchronize());
  SCONE_HIP(h, hipMemcpy(h_slots, h->slots, h->cap * sizeof(scone_slot), hipMemcpyDeviceToHost));
  SCONE_HIP(h, hipMemcpy(h_uni, h->d_uni, (size_t)SCONE_UNI_CAP * sizeof(int32_t), hipMemcpyDeviceToHost));
  SCONE_HIP(h, hipMemcpy(h_bloom, h->d_bloom, (h->bloom_mask + 1) / 8, hipMemcpyDeviceToHost));
  unsigned long long c[2] = {0, 0};
  SCONE_HIP(h, hipMemcpy(c, h->d_counters, sizeof(c), hipMemcpyDeviceToHost));
  if (n_keys) *n_keys = c[0];
  return SCONE_OK;
}

// The handle must have been created with the same max_n and index_capacity as the exporting one (the blob sizes are
// checked by the caller through scone_index_blob_sizes); replaces whatever the index held.
extern "C" int scone_index_import(scone_handle *h, const void *h_slots, const void *h_uni, const void *h_bloom, uint64_t n_keys) {
  if (!h || !h_slots || !h_uni || !h_bloom) return h ? scone_fail(h, SCONE_EINVAL, "scone_index_import: null pointer") : SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  index_modified(h);
  SCONE_HIP(h, hipDeviceSynchronize());
  SCONE_HIP(h, hipMemcpy(h->slots, h_slots, h->cap * sizeof(scone_slot), hipMemcpyHostToDevice));
  SCONE_HIP(h, hipMemcpy(h->d_uni, h_uni, (size_t)SCONE_UNI_CAP * sizeof(int32_t), hipMemcpyHostToDevice));
  SCONE_HIP(h, hipMemcpy(h->d_bloom, h_bloom, (h->bloom_mask + 1) / 8, hipMemcpyHostToDevice));
  unsigned long long c[2] = {n_keys, 0};
  SCONE_HIP(h, hipMemcpy(h->d_counters, c, sizeof(c), hipMemcpyHostToDevice));
  return SCONE_OK;
}

extern "C" int scone_match(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_hits,
                           scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (B < 0 || T < 0) return scone_fail(h, SCONE_EINVAL, "scone_match: negative B or T");
  if ((long long)B * T == 0) return SCONE_OK;
  if (!d_tok || !d_hits) return scone_fail(h, SCONE_EINVAL, "scone_match: null pointer");
  SCONE_ON_DEVICE(h);
  return scone_launch_match(h, d_tok, B, T, d_hits, (hipStream_t)stream);
}

extern "C" int scone_match_csr(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_offsets,
                               int32_t *d_ids, int64_t ids_cap, int64_t *h_total, scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (B < 0 || T < 0 || ids_cap < 0) return scone_fail(h, SCONE_EINVAL, "scone_match_csr: negative size");
  if (!d_offsets || !h_total) return scone_fail(h, SCONE_EINVAL, "scone_match_csr: null pointer");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  long long BT = (long long)B * T;
  if (BT == 0) {
    SCONE_HIP(h, hipMemsetAsync(d_offsets, 0, sizeof(int32_t), s));
    SCONE_HIP(h, hipStreamSynchronize(s));
    *h_total = 0;
    return SCONE_OK;
  }
  if (!d_tok) return scone_fail(h, SCONE_EINVAL, "scone_match_csr: null tokens");
  if (BT * SCONE_MAX_CAND > 0x7FFFFFFFll) return scone_fail(h, SCONE_EINVAL, "scone_match_csr: B*T too large for int32 offsets");
  scone_ws_lock ws(scone_ws_acquire(h, s));  // this stream's workspace, held until the total has been read back
  if (!ws.w) return scone_fail(h, SCONE_ENOMEM, "scone_match_csr: out of memory");
  scone_ws *w = ws.w;
  int rc = scone_ensure_hits(h, w, BT);
  if (rc) return rc;
  rc = scone_launch_match(h, d_tok, B, T, w->d_hits, s);
  if (rc) return rc;
  long long nb = (BT + CSR_TILE - 1) / CSR_TILE;
  if (nb > w->block_sums_cap) {
    if (w->d_block_sums) SCONE_HIP(h, hipFree(w->d_block_sums));
    w->d_block_sums = nullptr;
    w->block_sums_cap = 0;
    SCONE_HIP(h, hipMalloc(&w->d_block_sums, (size_t)nb * sizeof(int32_t)));
    w->block_sums_cap = nb;
  }
  if (!w->d_total) SCONE_HIP(h, hipMalloc(&w->d_total, sizeof(int64_t)));
  hipLaunchKernelGGL(k_csr_block_sums, dim3((unsigned)nb), dim3(CSR_BLOCK), 0, s, w->d_hits, BT, T, h->cfg.max_n,
                     w->d_block_sums);
  hipLaunchKernelGGL(k_csr_scan_sums, dim3(1), dim3(CSR_BLOCK), 0, s, w->d_block_sums, nb, (long long *)w->d_total);
  hipLaunchKernelGGL(k_csr_fill, dim3((unsigned)nb), dim3(CSR_BLOCK), 0, s, w->d_hits, BT, T, h->cfg.max_n,
                     w->d_block_sums, d_offsets, d_ids, d_ids ? (long long)ids_cap : 0ll);
  SCONE_HIP(h, hipGetLastError());
  long long total = 0;
  SCONE_HIP(h, hipMemcpyAsync(&total, w->d_total, sizeof(total), hipMemcpyDeviceToHost, s));
  SCONE_HIP(h, hipStreamSynchronize(s));
  *h_total = total;
  if (total > ids_cap) return scone_fail(h, SCONE_ERANGE, "scone_match_csr: ids_cap too small");
  return SCONE_OK;
}
