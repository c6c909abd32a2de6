/* Plain-C client of the C ABI (no Python, no torch): proves include/scone_hip.h is usable as the
 * drop-in boundary by itself.  Built and run by tests/test_gpu_parity.py::test_plain_c_client on the
 * GPU box:  gcc tests/cabi_smoke.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *               -Lscone_amd/csrc -lscone_hip -L/opt/rocm/lib -lamdhip64 -o /tmp/cabi_smoke
 * Vocabulary {(7,), (7,7), (7,7,7)} and tokens [7,7,7,7]: position 1 must get ids [0,1,1,2,2]
 * (n_gram_extractor.py:119-124) and the mean of those rows. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scone_hip.h"

#define CHECK(x)                                                        \
  do {                                                                  \
    int rc__ = (x);                                                     \
    if (rc__ != 0) {                                                    \
      fprintf(stderr, "%s failed: %d (%s)\n", #x, rc__, scone_strerror(rc__)); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

int main(void) {
  enum { D = 64, N = 3, T = 4 };
  scone_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.device = 0, cfg.max_n = 3, cfg.dim = D, cfg.table_fmt = SCONE_FMT_F32, cfg.placement = SCONE_PLACE_HBM;
  cfg.n_rows = N;
  scone_handle *h = NULL;
  CHECK(scone_create(&cfg, &h));

  uint32_t keys[N][3] = {{7, 0, 0}, {7, 7, 0}, {7, 7, 7}};
  uint8_t lens[N] = {1, 2, 3};
  CHECK(scone_index_build(h, &keys[0][0], lens, N, 0));

  float rows[N][D];
  for (int r = 0; r < N; ++r)
    for (int e = 0; e < D; ++e) rows[r][e] = (float)(r + 1) + 0.25f * (float)e;
  CHECK(scone_table_upload(h, rows, NULL, 0, N, 0, NULL));

  int32_t tok[T] = {7, 7, 7, 7};
  int32_t *d_tok, *d_off, *d_ids;
  float *d_out;
  if (hipMalloc((void **)&d_tok, sizeof tok) || hipMalloc((void **)&d_off, (T + 1) * 4) ||
      hipMalloc((void **)&d_ids, T * 6 * 4) || hipMalloc((void **)&d_out, T * D * 4))
    return 2;
  if (hipMemcpy(d_tok, tok, sizeof tok, hipMemcpyHostToDevice)) return 2;

  int64_t total = 0;
  CHECK(scone_match_csr(h, d_tok, 1, T, d_off, d_ids, T * 6, &total, NULL));
  int32_t off[T + 1], ids[T * 6];
  if (hipMemcpy(off, d_off, sizeof off, hipMemcpyDeviceToHost) || hipMemcpy(ids, d_ids, (size_t)total * 4, hipMemcpyDeviceToHost))
    return 2;
  const int32_t want1[5] = {0, 1, 1, 2, 2};
  if (total != 16 || off[2] - off[1] != 5 || memcmp(ids + off[1], want1, sizeof want1)) {
    fprintf(stderr, "id list mismatch (total %lld)\n", (long long)total);
    return 3;
  }

  CHECK(scone_embed(h, d_tok, 1, T, NULL, 0, NULL, 0, NULL, SCONE_REDUCE_MEAN, d_out, SCONE_DT_F32, NULL));
  float out[T][D];
  if (hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost)) return 2;
  for (int e = 0; e < D; ++e) {
    float s = rows[0][e];
    s += rows[1][e], s += rows[1][e], s += rows[2][e], s += rows[2][e];
    if (out[1][e] != s / 5.0f) {
      fprintf(stderr, "embedding mismatch at %d: %g vs %g\n", e, out[1][e], s / 5.0f);
      return 4;
    }
  }
  uint32_t bits = 0;
  CHECK(scone_status(h, &bits, NULL));
  if (bits) return 5;
  scone_destroy(h);
  printf("cabi_smoke ok: %lld hits, abi %d\n", (long long)total, scone_abi_version());
  return 0;
}
