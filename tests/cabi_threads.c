/* Plain-C client of the C ABI, two host threads on ONE pinned-host handle with a staging pipeline (cfg.stage_tokens > 0):
 * SURVEY 8b "lookups are thread-safe and stream-ordered".  The pipeline exists once per handle; scone_embed and
 * scone_embed_prefetch serialise on the handle's staging lock (round 5).  Each thread owns a HIP stream, looks up its own
 * batches (several chunks each), announces the next one, and compares every result byte for byte with the same lookup on an
 * HBM-resident twin of the table (computed up front on the main thread).  Built and run by
 * tests/test_gpu_parity.py::test_plain_c_client_two_threads_on_a_staged_handle:
 *   gcc tests/cabi_threads.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lscone_amd/csrc -lscone_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lpthread -o /tmp/cabi_threads
 * Table: N rows, d = 64, fp32 (the reference's own format, embedding_cache.py:86); vocabulary: every unigram of a 13-token
 * alphabet + bigrams (a, a + 1 mod 13) + trigrams (a, a, a). */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scone_hip.h"

enum { D = 64, V = 13, N = 3 * V, B = 24, T = 40, NBATCH = 4, ROUNDS = 60 };

static scone_handle *g_pin;
static int32_t *g_tok[NBATCH];       /* device, [B, T] */
static float *g_want[NBATCH];        /* host, [B, T, D] from the HBM twin */
static int g_fail[2];

static int build(scone_handle **out, int pinned) {
  scone_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.device = 0, cfg.max_n = 3, cfg.dim = D, cfg.table_fmt = SCONE_FMT_F32, cfg.n_rows = N;
  cfg.placement = pinned ? SCONE_PLACE_PINNED_HOST : SCONE_PLACE_HBM;
  if (pinned) cfg.hot_rows = 5, cfg.stage_tokens = 128; /* 3 sequences per chunk: 8 chunks per batch */
  int rc = scone_create(&cfg, out);
  if (rc) return rc;
  static uint32_t keys[N][3];
  static uint8_t lens[N];
  static float rows[N][D];
  for (int a = 0; a < V; ++a) {
    keys[a][0] = a, lens[a] = 1;
    keys[V + a][0] = a, keys[V + a][1] = (a + 1) % V, lens[V + a] = 2;
    keys[2 * V + a][0] = keys[2 * V + a][1] = keys[2 * V + a][2] = a, lens[2 * V + a] = 3;
  }
  for (int r = 0; r < N; ++r)
    for (int e = 0; e < D; ++e) rows[r][e] = (float)((r * 37 + e * 11) % 101) / 7.0f - 3.0f;
  rc = scone_index_build(*out, &keys[0][0], lens, N, 0);
  if (rc) return rc;
  return scone_table_upload(*out, rows, NULL, 0, N, 0, NULL);
}

static void *worker(void *arg) {
  const int me = (int)(size_t)arg;
  hipStream_t s;
  float *d_out, *h_out = malloc((size_t)B * T * D * 4);
  if (hipSetDevice(0) || hipStreamCreate(&s) || hipMalloc((void **)&d_out, (size_t)B * T * D * 4) || !h_out) {
    g_fail[me] = 100;
    return NULL;
  }
  unsigned x = 12345u + 77u * (unsigned)me;
  int k = me % NBATCH;
  for (int it = 0; it < ROUNDS && !g_fail[me]; ++it) {
    int rc = scone_embed(g_pin, g_tok[k], B, T, NULL, 0, NULL, 0, NULL, SCONE_REDUCE_MEAN, d_out, SCONE_DT_F32, s);
    x = x * 1664525u + 1013904223u;
    const int next = (int)((x >> 16) % NBATCH);
    if (!rc && it % 3 != 2) rc = scone_embed_prefetch(g_pin, g_tok[next], B, T, 1, s); /* used, dropped or foreign: all legal */
    if (rc) {
      fprintf(stderr, "thread %d: call failed: %d (%s)\n", me, rc, scone_last_error(g_pin));
      g_fail[me] = 1;
      break;
    }
    if (it % 2 == 0) {
      if (hipMemcpyAsync(h_out, d_out, (size_t)B * T * D * 4, hipMemcpyDeviceToHost, s) || hipStreamSynchronize(s)) g_fail[me] = 101;
      else if (memcmp(h_out, g_want[k], (size_t)B * T * D * 4)) {
        fprintf(stderr, "thread %d: batch %d differs from the HBM twin at iteration %d\n", me, k, it);
        g_fail[me] = 2;
      }
    }
    k = next;
  }
  (void)hipStreamSynchronize(s);
  (void)hipFree(d_out);
  (void)hipStreamDestroy(s);
  free(h_out);
  return NULL;
}

int main(void) {
  scone_handle *hbm = NULL;
  if (build(&hbm, 0) || build(&g_pin, 1)) {
    fprintf(stderr, "build failed: %s\n", scone_last_error(g_pin ? g_pin : hbm));
    return 1;
  }
  float *d_out;
  if (hipMalloc((void **)&d_out, (size_t)B * T * D * 4)) return 2;
  unsigned x = 99u;
  for (int b = 0; b < NBATCH; ++b) {
    static int32_t tok[B * T];
    for (int i = 0; i < B * T; ++i) {
      x = x * 1664525u + 1013904223u;
      tok[i] = (i % 7 == 3) ? tok[i - 1] : (int32_t)((x >> 16) % V); /* repeats: bigram / trigram windows do hit */
    }
    g_want[b] = malloc((size_t)B * T * D * 4);
    if (hipMalloc((void **)&g_tok[b], sizeof tok) || hipMemcpy(g_tok[b], tok, sizeof tok, hipMemcpyHostToDevice)) return 2;
    if (scone_embed(hbm, g_tok[b], B, T, NULL, 0, NULL, 0, NULL, SCONE_REDUCE_MEAN, d_out, SCONE_DT_F32, NULL)) return 3;
    if (hipMemcpy(g_want[b], d_out, (size_t)B * T * D * 4, hipMemcpyDeviceToHost)) return 2;
  }
  pthread_t th[2];
  for (size_t i = 0; i < 2; ++i) pthread_create(&th[i], NULL, worker, (void *)i);
  for (int i = 0; i < 2; ++i) pthread_join(th[i], NULL);
  uint32_t bits = 0;
  uint64_t cache_rows = 0, copied = 0, chunks = 0, chunk_tokens = 0;
  if (scone_status(g_pin, &bits, NULL) || scone_stage_counters(g_pin, &cache_rows, &copied, &chunks, &chunk_tokens)) return 4;
  if (g_fail[0] || g_fail[1] || bits || chunks == 0) {
    fprintf(stderr, "FAILED: thread status %d %d, status bits %u, chunks %llu\n", g_fail[0], g_fail[1], bits, (unsigned long long)chunks);
    return 5;
  }
  printf("cabi_threads ok: 2 threads x %d lookups on one staged handle, %llu chunks of %llu tokens, %llu rows over PCIe\n", ROUNDS,
         (unsigned long long)chunks, (unsigned long long)chunk_tokens, (unsigned long long)copied);
  scone_destroy(g_pin);
  scone_destroy(hbm);
  return 0;
}
