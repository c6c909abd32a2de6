/* TEST INFRASTRUCTURE ONLY -- runs oracle/oracle.c under AddressSanitizer / UBSan (oracle/Makefile target `asan`).
 *
 *   oracle_asan <cases.bin> <out.bin>
 *
 * cases.bin (written by tests/test_oracle_golden.py from the golden fixtures, little endian):
 *   int64 n_cases; per case: int64 N, max_n, d, B, T; uint32 keys[N * max_n]; uint8 lens[N]; float table[N * d];
 *   int64 tok[B * T]
 * out.bin: per case: int64 total; int64 offsets[B * (T + 1)] (per sequence); int64 ids[total]; float mean[B * T * d]
 * The sanitizers abort with a non-zero status on any finding; leaks are checked at exit. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct oracle_index oracle_index;
oracle_index *oracle_index_new(const uint32_t *keys, const uint8_t *lens, int64_t n, int max_n);
void oracle_index_free(oracle_index *ix);
int64_t oracle_match_csr(const oracle_index *ix, const int64_t *tok, int64_t T, int64_t *offsets, int64_t *ids);
int64_t oracle_embed_batch(const oracle_index *ix, const float *table, int64_t d, const int64_t *tok, int64_t B, int64_t T,
                           int mean, float *out, int nthreads);

static void rd(void *p, size_t n, FILE *f) {
  if (n && fread(p, 1, n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(3);
  }
}

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
  if (!in || !out) return 2;
  int64_t n_cases = 0;
  rd(&n_cases, 8, in);
  for (int64_t c = 0; c < n_cases; ++c) {
    int64_t h[5];
    rd(h, sizeof h, in);
    const int64_t N = h[0], max_n = h[1], d = h[2], B = h[3], T = h[4];
    uint32_t *keys = (uint32_t *)malloc((size_t)(N * max_n + 1) * 4);
    uint8_t *lens = (uint8_t *)malloc((size_t)N + 1);
    float *table = (float *)malloc((size_t)(N * d + 1) * 4);
    int64_t *tok = (int64_t *)malloc((size_t)(B * T + 1) * 8);
    rd(keys, (size_t)(N * max_n) * 4, in);
    rd(lens, (size_t)N, in);
    rd(table, (size_t)(N * d) * 4, in);
    rd(tok, (size_t)(B * T) * 8, in);
    oracle_index *ix = oracle_index_new(keys, lens, N, (int)max_n);
    const int64_t nc = max_n * (max_n + 1) / 2;
    int64_t *off = (int64_t *)malloc((size_t)(B * (T + 1) + 1) * 8);
    int64_t *ids = (int64_t *)malloc((size_t)(B * T * nc + 1) * 8);
    int64_t total = 0;
    for (int64_t b = 0; b < B; ++b) total += oracle_match_csr(ix, tok + b * T, T, off + b * (T + 1), ids + total);
    float *mean = (float *)malloc((size_t)(B * T * d + 1) * 4);
    const int64_t total2 = oracle_embed_batch(ix, table, d, tok, B, T, 1, mean, 2);
    if (total2 != total) return 4;
    fwrite(&total, 8, 1, out);
    fwrite(off, 8, (size_t)(B * (T + 1)), out);
    fwrite(ids, 8, (size_t)total, out);
    fwrite(mean, 4, (size_t)(B * T * d), out);
    oracle_index_free(ix);
    free(keys), free(lens), free(table), free(tok), free(off), free(ids), free(mean);
  }
  fclose(in);
  fclose(out);
  return 0;
}
