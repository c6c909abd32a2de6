/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's f-gram lookup path.
 *
 * Second, independent oracle beside oracle/ref_port.py (and a stronger CPU baseline for bench.py's
 * informational `cpu_c_oracle` field).  Nothing under scone_amd/ may link or load it.
 * Built by oracle/Makefile into oracle/_build/liboracle.so; checked against the golden fixtures
 * captured from the real reference in tests/test_oracle_golden.py.
 *
 * Follows (paths relative to the reference checkout):
 *   match  : NGramExtractor.get_token_f_grams      scone/tokenization/n_gram_extractor.py:106-126
 *   id map : f_gram_to_id[g]                       scone/inference/embedding_cache.py:173
 *   gather : EmbeddingCache.get_embeddings         scone/inference/embedding_cache.py:113-147
 *   mean   : embeddings.mean(dim=0), zero fill     scone/inference/engine.py:247-259
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint64_t lo;   /* tokens 0,1 (+1 each, 0 = absent) */
  uint32_t ext;  /* token 2 (+1); for max_n = 4: token 3 in a second word */
  uint32_t ext2;
  int64_t id;    /* -1 = empty */
} slot_t;

typedef struct {
  slot_t *slots;
  uint64_t mask;
  int max_n;
} oracle_index;

static uint64_t mix(uint64_t x) {
  x ^= x >> 33, x *= 0xff51afd7ed558ccdULL, x ^= x >> 33, x *= 0xc4ceb9fe1a85ec53ULL, x ^= x >> 33;
  return x;
}

static void pack(const int64_t *t, int n, uint64_t *lo, uint32_t *e, uint32_t *e2) {
  *lo = ((uint64_t)(uint32_t)(t[0] + 1)) | (n > 1 ? ((uint64_t)(uint32_t)(t[1] + 1)) << 32 : 0);
  *e = n > 2 ? (uint32_t)(t[2] + 1) : 0;
  *e2 = n > 3 ? (uint32_t)(t[3] + 1) : 0;
}

/* keys[n, max_n] (first lens[i] used), id = row; first id wins on duplicates (repo convention) */
oracle_index *oracle_index_new(const uint32_t *keys, const uint8_t *lens, int64_t n, int max_n) {
  oracle_index *ix = (oracle_index *)calloc(1, sizeof *ix);
  uint64_t cap = 16;
  while (cap < (uint64_t)(2 * n + 1)) cap <<= 1;
  ix->slots = (slot_t *)malloc(cap * sizeof(slot_t));
  ix->mask = cap - 1, ix->max_n = max_n;
  for (uint64_t s = 0; s < cap; ++s) ix->slots[s].id = -1;
  for (int64_t i = 0; i < n; ++i) {
    int64_t t[4] = {0, 0, 0, 0};
    for (int k = 0; k < lens[i]; ++k) t[k] = keys[i * max_n + k];
    uint64_t lo; uint32_t e, e2;
    pack(t, lens[i], &lo, &e, &e2);
    uint64_t s = mix(lo ^ ((uint64_t)e << 17) ^ ((uint64_t)e2 << 41)) & ix->mask;
    for (;;) {
      slot_t *sl = &ix->slots[s];
      if (sl->id < 0) { sl->lo = lo, sl->ext = e, sl->ext2 = e2, sl->id = i; break; }
      if (sl->lo == lo && sl->ext == e && sl->ext2 == e2) break;
      s = (s + 1) & ix->mask;
    }
  }
  return ix;
}

void oracle_index_free(oracle_index *ix) {
  if (ix) { free(ix->slots); free(ix); }
}

static int64_t find(const oracle_index *ix, const int64_t *t, int n) {
  for (int k = 0; k < n; ++k)
    if (t[k] < 0 || t[k] >= 0xFFFFFFFFLL) return -1;
  uint64_t lo; uint32_t e, e2;
  pack(t, n, &lo, &e, &e2);
  uint64_t s = mix(lo ^ ((uint64_t)e << 17) ^ ((uint64_t)e2 << 41)) & ix->mask;
  for (;;) {
    const slot_t *sl = &ix->slots[s];
    if (sl->id < 0) return -1;
    if (sl->lo == lo && sl->ext == e && sl->ext2 == e2) return sl->id;
    s = (s + 1) & ix->mask;
  }
}

/* One sequence of T tokens -> CSR of the per-position id lists in the reference's append order
 * (n ascending, window start ascending, duplicates kept).  ids must hold T * max_n(max_n+1)/2 entries. */
int64_t oracle_match_csr(const oracle_index *ix, const int64_t *tok, int64_t T, int64_t *offsets, int64_t *ids) {
  const int max_n = ix->max_n;
  /* hit id per (n, start), then expand per position -- same result as appending while scanning */
  int64_t *hit = (int64_t *)malloc((size_t)(max_n * (T > 0 ? T : 1)) * sizeof(int64_t));
  for (int n = 1; n <= max_n; ++n)
    for (int64_t i = 0; i < T; ++i) hit[(n - 1) * T + i] = (i + n <= T) ? find(ix, tok + i, n) : -1;
  int64_t w = 0;
  for (int64_t j = 0; j < T; ++j) {
    offsets[j] = w;
    for (int n = 1; n <= max_n; ++n)
      for (int64_t i = j - n + 1; i <= j; ++i)
        if (i >= 0 && hit[(n - 1) * T + i] >= 0) ids[w++] = hit[(n - 1) * T + i];
  }
  offsets[T] = w;
  free(hit);
  return w;
}

/* out[T, d] = mean (or sum) of the fp32 rows of each position's list; zeros where the list is empty.
 * Sequential fp32 sum in list order, then one division (torch.mean over dim 0). */
void oracle_aggregate(const float *table, int64_t d, const int64_t *offsets, const int64_t *ids, int64_t T, int mean,
                      float *out) {
  for (int64_t j = 0; j < T; ++j) {
    float *o = out + j * d;
    memset(o, 0, (size_t)d * sizeof(float));
    const int64_t k0 = offsets[j], k1 = offsets[j + 1];
    for (int64_t k = k0; k < k1; ++k) {
      const float *r = table + ids[k] * d;
      for (int64_t e = 0; e < d; ++e) o[e] = o[e] + r[e];
    }
    if (mean && k1 - k0 > 1) {
      const float kf = (float)(k1 - k0);
      for (int64_t e = 0; e < d; ++e) o[e] = o[e] / kf;
    }
  }
}

/* B independent sequences of T tokens (row-major tok[B, T]) -> out[B, T, d]; returns total hits.
 * nthreads > 1 splits the sequences over OpenMP threads (sequences are independent). */
int64_t oracle_embed_batch(const oracle_index *ix, const float *table, int64_t d, const int64_t *tok, int64_t B, int64_t T,
                           int mean, float *out, int nthreads) {
  const int nc = ix->max_n * (ix->max_n + 1) / 2;
  int64_t total = 0;
#pragma omp parallel for num_threads(nthreads) reduction(+ : total) schedule(dynamic, 1)
  for (int64_t b = 0; b < B; ++b) {
    int64_t *off = (int64_t *)malloc((size_t)(T + 1) * sizeof(int64_t));
    int64_t *ids = (int64_t *)malloc((size_t)(T * nc + 1) * sizeof(int64_t));
    total += oracle_match_csr(ix, tok + b * T, T, off, ids);
    oracle_aggregate(table, d, off, ids, T, mean, out + b * T * d);
    free(off);
    free(ids);
  }
  return total;
}
