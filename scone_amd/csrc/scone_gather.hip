// The bandwidth-bound core: sparse row gather + dequantise + ordered fp32 reduce +
// combine with the base token / position embeddings.  No MFMA: every byte fetched
// is used once; the kernel is priced against the HBM roofline.
//
// Replaces, on the GPU:
//   EmbeddingCache.get_token_embeddings / get_embeddings  scone/inference/embedding_cache.py:113-181
//   embeddings.mean(dim=0), zero-fill, .half()            scone/inference/engine.py:247-266
//   wte(input_ids) + f_gram_embeddings + wpe(position_ids) scone/models/language_model.py:239-254
//
// This file holds the entry points (scone_embed, scone_gather_reduce, scone_embed_partial, scone_finalize, the embed
// halves of the shard exchanges) and the staged-prefetch driver; the kernels are in scone_embed_wave.h (wave per token:
// every d % 8 == 0) and scone_gather_impl.h (k_embed, the lane-group fallback for other dims: a group of LPT lanes owns
// one token, each lane 16-byte vectors of every row).  Accumulation is sequential in the reference's list order in
// fp32 everywhere, so results do not depend on the launch geometry.
#include "scone_gather_impl.h"
#ifdef SCONE_PROBE_PERM
#include <cstdlib>
#endif

using namespace scone_gather;

// One-launch limit (scone_handle::fused_max_tokens, default 32768): measured crossover (round 2's tools/latency.py; now bench.py's `latency` block) -- 8K tokens
// 15.5 -> 9.9 us, 16K 20.5 -> 16.8, 32K 30.2 -> 29.3, 64K 45.7 -> 53.7: above it the two-kernel form wins (one probe per
// window, not per covered token).

namespace {

int launch_fmt(scone_handle *h, const embed_args &a, int src, int mode, int out_dtype, hipStream_t s) {
  if (mode == MODE_FINALIZE) return launch_f32(h, a, src, mode, out_dtype, s);
  switch (h->cfg.table_fmt) {
    case SCONE_FMT_F32: return launch_f32(h, a, src, mode, out_dtype, s);
    case SCONE_FMT_F16: return launch_f16(h, a, src, mode, out_dtype, s);
    case SCONE_FMT_I8: return launch_i8(h, a, src, mode, out_dtype, s);
    case SCONE_FMT_I4: return launch_i4(h, a, src, mode, out_dtype, s);
    default: return scone_fail(h, SCONE_EINVAL, "unknown table_fmt");
  }
}
// formats / dims served by k_embed_wave (scone_embed_wave.h: wave_geom<>::OK)
bool scone_wave_kernel_covers(int fmt, int d) {
  (void)fmt;
  return d % 8 == 0;  // specialised kernels for 768 / 1024 / 1280, the unit-walking kernel otherwise
}

void fill_table_view(const scone_handle *h, table_view &tv) {
  tv.st = scone_store_of(h);
  tv.scales = reinterpret_cast<const __half *>(h->scales);
  tv.row_begin = (long long)h->cfg.row_begin;
  tv.row_end = (long long)h->cfg.row_end;
  tv.n_rows = (long long)h->cfg.n_rows;
  tv.row_bytes = (int)h->row_payload_bytes;
  tv.d = h->cfg.dim;
}

int check_mode(scone_handle *h, const char *who) {
  if (h->cfg.lookup_mode == SCONE_MODE_COVER) return SCONE_OK;
  if (h->cfg.lookup_mode == SCONE_MODE_LONGEST_SUFFIX && scone_wave_kernel_covers(h->cfg.table_fmt, h->cfg.dim)) return SCONE_OK;
  return scone_fail(h, SCONE_EINVAL, who);
}

// decode-size batches are launch-bound: one fused launch (k_embed_fused) instead of match + gather
// (INT4 has no specialised kernel at d = 768 / 1280: a row segment would be narrower than one 16-B access)
bool scone_embed_takes_one_launch(const scone_handle *h, long long BT) {
  return BT <= h->fused_max_tokens && (h->cfg.dim == 768 || h->cfg.dim == 1024 || h->cfg.dim == 1280) &&
         !(h->cfg.table_fmt == SCONE_FMT_I4 && h->cfg.dim != 1024);
}

#ifdef SCONE_PROBE_PERM
// TRAFFIC PROBE of a token-ordered ("run-ordered") lookup (tools/run_order_probe.py; a -DSCONE_PROBE_PERM build only, never the
// shipped library).  The caller's process puts the device address of a permutation perm[B*T] into the environment
// (SCONE_PROBE_PERM_PTR); scone_embed then looks up, at slot j of the launch, the token of position perm[j] -- its id record,
// token id and position row -- and stores the result to out[j].  Every vector is the right one for SOME position, only its
// place in `out` is permuted: the bytes the kernel moves are those of a kernel that visits the positions in the order the
// permutation gives, which is what the probe measures (FETCH_SIZE, kernel time).
__global__ void k_probe_permute(const int32_t *__restrict__ perm, const int32_t *__restrict__ ell, const int32_t *__restrict__ tok,
                                int32_t *__restrict__ ell2, int32_t *__restrict__ tok2, int32_t *__restrict__ pos2, long long BT,
                                int T, int W) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long j = t / W;
  const int w = (int)(t % W);
  if (j >= BT) return;
  const long long p = perm[j];
  ell2[j * W + w] = ell[p * W + w];
  if (w == 0) tok2[j] = tok[p], pos2[j] = (int32_t)(p % T);
}
static int32_t *g_probe_buf = nullptr;
static long long g_probe_cap = 0;
#endif

int need_table(scone_handle *h, const char *who) {
  if (!h) return SCONE_EINVAL;
  if (h->cfg.dim <= 0 || !h->rows) return scone_fail(h, SCONE_ESTATE, who);
  return SCONE_OK;
}

}  // namespace

// Pinned-host table, prefetch through the persistent HBM cache of cold rows (scone_stage.hip): chunks of whole sequences;
// side streams prepare chunk c+2 and copy the rows chunk c+1 misses while the caller's stream reduces chunk c out of
// [hot head | cache].
// chunk geometry of a staged batch: sequences per chunk (0 = a sequence does not fit a chunk)
static int staged_geometry(scone_handle *h, int32_t B, int32_t T, long long *seqs_out) {
  long long seqs = (long long)h->cfg.stage_tokens / T;
  if (seqs < 1) seqs = 1;
  if (seqs > B) seqs = B;
  int rc = scone_stage_prepare(h, seqs * T);
  if (rc) return rc;
  seqs = scone_stage_chunk_tokens(h) / T;
  if (seqs < 1) return scone_fail(h, SCONE_EINVAL, "scone_embed(staged): sequence too long for one staging chunk");
  if (seqs > B) seqs = B;
  *seqs_out = seqs;
  return SCONE_OK;
}

static int embed_staged_chunks(scone_handle *h, const embed_args &full, int32_t B, int32_t T, int32_t out_dtype, hipStream_t s);

// One staging pipeline per handle: the whole call holds stage_mu (calls from several host threads are serialised; on the
// device their chunks follow each other through the pipeline's events like those of consecutive calls from one thread).  A
// call that fails after chunks were prepared drops the whole pipeline (scone_stage_resync): the next call starts a cold cache.
static int embed_staged(scone_handle *h, const embed_args &full, int32_t B, int32_t T, int32_t out_dtype, hipStream_t s) {
  std::lock_guard<std::mutex> g(h->stage_mu);
  const int rc = embed_staged_chunks(h, full, B, T, out_dtype, s);
  if (rc && h->stage) scone_stage_resync(h);
  return rc;
}

static int embed_staged_chunks(scone_handle *h, const embed_args &full, int32_t B, int32_t T, int32_t out_dtype, hipStream_t s) {
  long long seqs = 0;
  int rc = staged_geometry(h, B, T, &seqs);
  if (rc) return rc;
  rc = scone_stage_bind(h, s);
  if (rc) return rc;
  const size_t esz = out_dtype == SCONE_DT_F32 ? 4 : 2;
  const long long nchunks = (B + seqs - 1) / seqs;
  auto chunk_b = [&](long long c) { return (int32_t)((c + 1) * seqs <= B ? seqs : B - c * seqs); };
  // chunks scone_embed_prefetch prepared for exactly this batch are taken over (their side-stream work was ordered behind the
  // stream the caller named THEN); otherwise the side stream starts after everything already queued on the caller's stream
  // (tokens may be produced there)
  long long staged = scone_stage_take_prefetched(h, full.tok, B, T, seqs);
  if (staged == 0) {
    SCONE_HIP(h, hipEventRecord(scone_stage_start_event(h), s));
    SCONE_HIP(h, hipStreamWaitEvent(scone_stage_side(h), scone_stage_start_event(h), 0));
  }
  // chunks 0 .. AHEAD-1 are staged ahead; inside the loop chunk c + AHEAD is queued before chunk c is reduced
  for (; staged < nchunks && staged < SCONE_STAGE_AHEAD; ++staged) {
    rc = scone_stage_chunk(h, full.tok + staged * seqs * T, chunk_b(staged), T);
    if (rc) return rc;
  }
  for (long long c = 0; c < nchunks; ++c) {
    if (staged < nchunks && staged <= c + SCONE_STAGE_AHEAD) {  // (its set was last used by a chunk whose lookup is already queued)
      rc = scone_stage_chunk(h, full.tok + staged * seqs * T, chunk_b(staged), T);
      if (rc) return rc;
      ++staged;
    }
    const int buf = scone_stage_consume_buf(h);
    const long long t0 = c * seqs * T;
    embed_args a = full;
    a.BT = (long long)chunk_b(c) * T, a.ntok = a.BT;
    a.tok = full.tok + t0;
    a.pos = full.pos ? full.pos + t0 : nullptr;
    a.out = reinterpret_cast<uint8_t *>(full.out) + (size_t)t0 * h->cfg.dim * esz;
    a.ell = scone_stage_ell(h, buf);
    a.tv.st.cold = scone_stage_rows(h);  // the records now hold n_hot + cache slot: the cold half of the row store is the cache
    if (h->scale_bytes_per_row) a.tv.scales = reinterpret_cast<const __half *>(scone_stage_scales(h));
    SCONE_HIP(h, hipStreamWaitEvent(s, scone_stage_staged_event(h, buf), 0));
    rc = scone_prof_begin(h, s);
    if (rc) return rc;
    rc = launch_fmt(h, a, SRC_HITS, MODE_FULL, out_dtype, s);
    if (rc) {
      scone_prof_abort(h);
      return rc;
    }
    rc = scone_prof_end(h, s);
    if (rc) return rc;
    rc = scone_stage_mark_consumed(h, buf, s);
    if (rc) return rc;
  }
  return SCONE_OK;
}

// The north-star's "async prefetch" as an entry point: the first chunks of the NEXT batch are matched, placed and copied on the
// side streams now -- behind `stream`, where its tokens are (or were) produced -- so that the scone_embed of that batch finds them
// done.  A no-op for tables without a prefetch pipeline.
extern "C" int scone_embed_prefetch(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t tokens_ready,
                                    scone_stream_t stream) {
  int rc = need_table(h, "scone_embed_prefetch: handle has no table (dim == 0)");
  if (rc) return rc;
  if (B < 0 || T < 0) return scone_fail(h, SCONE_EINVAL, "scone_embed_prefetch: negative B or T");
  if ((long long)B * T == 0) return SCONE_OK;
  if (!d_tok) return scone_fail(h, SCONE_EINVAL, "scone_embed_prefetch: null pointer");
  if (!scone_wave_kernel_covers(h->cfg.table_fmt, h->cfg.dim)) return SCONE_OK;
  SCONE_ON_DEVICE(h);
  // Rows in HBM (or read in place over PCIe): nothing to do.  What could run ahead there is the match of the next batch; built
  // and measured in round 5 (k_match_ell on a side stream of the handle beside the previous gather, 2-4 record buffers, every
  // stream priority): 1-19 % SLOWER than match-then-gather on one stream at every batch size from 33k to 1M tokens -- the
  // gather kernel holds every wave slot of the chip, a match workgroup only gets in where a gather workgroup retires, and
  // the gather loses exactly the time the match takes (profiles/r05b, profiles/r05c; the implementation is kept there as a
  // diff).  Removed.
  if (!(h->cfg.stage_tokens && h->rows_host)) return SCONE_OK;
  std::lock_guard<std::mutex> g(h->stage_mu);
  long long seqs = 0;
  rc = staged_geometry(h, B, T, &seqs);
  if (rc) return rc;
  rc = scone_stage_bind(h, (hipStream_t)stream);
  if (rc) return rc;
  (void)scone_stage_take_prefetched(h, nullptr, -1, -1, -1);  // an earlier prefetch that was never used is dropped
  const long long nchunks = (B + seqs - 1) / seqs;
  if (!tokens_ready) {  // behind whatever is queued on `stream` -- a lookup queued there just before this call included
    SCONE_HIP(h, hipEventRecord(scone_stage_start_event(h), (hipStream_t)stream));
    SCONE_HIP(h, hipStreamWaitEvent(scone_stage_side(h), scone_stage_start_event(h), 0));
  }
  long long n = 0;
  for (; n < nchunks && n < SCONE_STAGE_AHEAD; ++n) {
    const int32_t bc = (int32_t)((n + 1) * seqs <= B ? seqs : B - n * seqs);
    rc = scone_stage_chunk(h, d_tok + n * seqs * T, bc, T);
    if (rc) {
      scone_stage_resync(h);
      return rc;
    }
  }
  return scone_stage_note_prefetched(h, d_tok, B, T, seqs, n);
}

extern "C" int scone_gather_reduce(scone_handle *h, const int32_t *d_offsets, const int32_t *d_ids, int64_t ntok,
                                   const void *d_base, int32_t reduce, void *d_out, int32_t out_dtype,
                                   scone_stream_t stream) {
  int rc = need_table(h, "scone_gather_reduce: handle has no table (dim == 0)");
  if (rc) return rc;
  if (ntok < 0) return scone_fail(h, SCONE_EINVAL, "scone_gather_reduce: negative ntok");
  if (ntok == 0) return SCONE_OK;
  if (!d_offsets || !d_out) return scone_fail(h, SCONE_EINVAL, "scone_gather_reduce: null pointer");
  if (reduce != SCONE_REDUCE_MEAN && reduce != SCONE_REDUCE_SUM)
    return scone_fail(h, SCONE_EINVAL, "scone_gather_reduce: bad reduce");
  SCONE_ON_DEVICE(h);
  embed_args a = {};
  fill_table_view(h, a.tv);
  a.offsets = d_offsets, a.ids = d_ids;
  a.BT = ntok, a.T = (int)(ntok > 0x7FFFFFFF ? 0x7FFFFFFF : ntok), a.max_n = h->cfg.max_n;
  a.tok_begin = 0, a.ntok = ntok;
  a.base = d_base, a.reduce = reduce, a.out = d_out, a.status = h->d_status;
  a.zero_row = h->d_zero_row;
  return launch_fmt(h, a, SRC_CSR, MODE_FULL, out_dtype, (hipStream_t)stream);
}

extern "C" int scone_embed(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, const void *d_wte,
                           int64_t vocab, const void *d_wpe, int64_t n_pos, const int32_t *d_pos, int32_t reduce,
                           void *d_out, int32_t out_dtype, scone_stream_t stream) {
  int rc = need_table(h, "scone_embed: handle has no table (dim == 0)");
  if (rc) return rc;
  if (B < 0 || T < 0) return scone_fail(h, SCONE_EINVAL, "scone_embed: negative B or T");
  const long long BT = (long long)B * T;
  if (BT == 0) return SCONE_OK;
  if (!d_tok || !d_out) return scone_fail(h, SCONE_EINVAL, "scone_embed: null pointer");
  if (reduce != SCONE_REDUCE_MEAN && reduce != SCONE_REDUCE_SUM)
    return scone_fail(h, SCONE_EINVAL, "scone_embed: bad reduce");
  if ((d_wte && vocab <= 0) || (d_wpe && n_pos <= 0))
    return scone_fail(h, SCONE_EINVAL, "scone_embed: wte/wpe given without vocab/n_pos");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  embed_args a = {};
  fill_table_view(h, a.tv);
  a.BT = BT, a.T = T, a.max_n = h->cfg.max_n;
  a.tok_begin = 0, a.ntok = BT;
  a.zero_row = h->d_zero_row, a.mode = (int)h->cfg.lookup_mode;
  rc = check_mode(h, "scone_embed: lookup_mode longest_suffix needs d % 8 == 0");
  if (rc) return rc;
  a.tok = d_tok, a.pos = d_pos, a.wte = d_wte, a.vocab = vocab, a.wpe = d_wpe, a.n_pos = n_pos;
  a.reduce = reduce, a.out = d_out, a.status = h->d_status;
  if (h->cfg.stage_tokens && h->rows_host) {
    if (!scone_wave_kernel_covers(h->cfg.table_fmt, h->cfg.dim))
      return scone_fail(h, SCONE_EINVAL, "scone_embed: staged prefetch needs d % 8 == 0");
    return embed_staged(h, a, B, T, out_dtype, s);
  }
  if (scone_embed_takes_one_launch(h, BT)) {
    a.fused = 1;
    rc = scone_prof_begin(h, s);
    if (rc) return rc;
    rc = launch_fmt(h, a, SRC_HITS, MODE_FULL, out_dtype, s);
    if (rc) {
      scone_prof_abort(h);
      return rc;
    }
    return scone_prof_end(h, s);
  }
  // the workspace of THIS stream, held while the match that writes it and the lookup that reads it are enqueued
  scone_ws_lock ws(scone_ws_acquire(h, s));
  if (!ws.w) return scone_fail(h, SCONE_ENOMEM, "scone_embed: out of memory");
  if (scone_wave_kernel_covers(h->cfg.table_fmt, h->cfg.dim)) {
    // fast path: per-token id records (one scalar load per token in the gather kernel)
    rc = scone_ensure_ell(h, ws.w, BT);
    if (rc) return rc;
    rc = scone_launch_match_ell(h, d_tok, B, T, ws.w->d_ell, s);
    if (rc) return rc;
    a.ell = ws.w->d_ell;
  } else {
    rc = scone_ensure_hits(h, ws.w, BT);
    if (rc) return rc;
    rc = scone_launch_match(h, d_tok, B, T, ws.w->d_hits, s);
    if (rc) return rc;
    a.hits = ws.w->d_hits;
  }
  a.tok = d_tok, a.pos = d_pos, a.wte = d_wte, a.vocab = vocab, a.wpe = d_wpe, a.n_pos = n_pos;
  a.reduce = reduce, a.out = d_out, a.status = h->d_status;
#ifdef SCONE_PROBE_PERM
  if (const char *e = getenv("SCONE_PROBE_PERM_PTR")) {
    const int32_t *perm = reinterpret_cast<const int32_t *>(strtoull(e, nullptr, 0));
    const int W = h->cfg.max_n <= 3 ? 8 : 16;
    if (perm && a.ell && !d_pos) {
      if (g_probe_cap < BT) {
        if (g_probe_buf) (void)hipFree(g_probe_buf);
        SCONE_HIP(h, hipMalloc(&g_probe_buf, (size_t)BT * (W + 2) * sizeof(int32_t)));
        g_probe_cap = BT;
      }
      int32_t *ell2 = g_probe_buf, *tok2 = g_probe_buf + BT * W, *pos2 = tok2 + BT;
      const long long n = BT * W;
      hipLaunchKernelGGL(k_probe_permute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, perm, a.ell, d_tok, ell2, tok2, pos2,
                         BT, T, W);
      a.ell = ell2, a.tok = tok2, a.pos = pos2;
    }
  }
#endif
  rc = scone_prof_begin(h, s);
  if (rc) return rc;
  rc = launch_fmt(h, a, SRC_HITS, MODE_FULL, out_dtype, s);
  if (rc) {
    scone_prof_abort(h);
    return rc;
  }
  return scone_prof_end(h, s);
}

// All-gather form.  The gathered records (one per distinct row, every shard's) join the handle's row map with
// scone_shard_gather_add_records -- all at once, or chunk by chunk as the all-gathers of a pipelined exchange complete --
// and scone_shard_gather_embed_range reduces a run of sequences of the planned batch out of
// [replicated head | records received so far]; scone_shard_gather_embed is the two for a whole batch.
extern "C" int scone_shard_gather_add_records(scone_handle *h, const void *d_records_base, uint64_t record0, uint64_t n_records,
                                              uint64_t n_total, scone_stream_t stream) {
  int rc = need_table(h, "scone_shard_gather_add_records: handle has no table (dim == 0)");
  if (rc) return rc;
  if (!h->shard) return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_add_records: call scone_shard_gather_plan first");
  if (n_records && !d_records_base) return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_add_records: null records");
  SCONE_ON_DEVICE(h);
  const uint8_t *p = reinterpret_cast<const uint8_t *>(d_records_base) + (size_t)record0 * (size_t)scone_shard_rec_bytes(h);
  return scone_shard_gather_add(h, p, n_records, record0, n_total, (hipStream_t)stream);
}

extern "C" int scone_shard_gather_embed_range(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t seq_begin,
                                              int32_t seq_end, const void *d_records_base, uint64_t n_total, const void *d_wte,
                                              int64_t vocab, const void *d_wpe, int64_t n_pos, const int32_t *d_pos,
                                              int32_t reduce, void *d_out, int64_t out_tok0, int32_t out_dtype,
                                              scone_stream_t stream) {
  int rc = need_table(h, "scone_shard_gather_embed: handle has no table (dim == 0)");
  if (rc) return rc;
  if (!h->shard) return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_embed: call scone_shard_gather_plan first");
  int32_t pB = 0, pT = 0;
  scone_shard_plan_shape(h, &pB, &pT);
  if (B < 0 || T <= 0 || B != pB || T != pT || seq_begin < 0 || seq_end < seq_begin || seq_end > B || (n_total && !d_records_base) ||
      out_tok0 < 0 || out_tok0 > (long long)seq_begin * T)
    return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_embed: bad argument (B, T must be the planned batch's)");
  if (!scone_wave_kernel_covers(h->cfg.table_fmt, h->cfg.dim))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_embed: needs d % 8 == 0");
  if (reduce != SCONE_REDUCE_MEAN && reduce != SCONE_REDUCE_SUM)
    return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_embed: bad reduce");
  const long long nt = (long long)(seq_end - seq_begin) * T;
  if (nt == 0) return SCONE_OK;
  if (!d_tok || !d_out) return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_embed: null pointer");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  const int32_t *ell = nullptr;
  const void *scales = nullptr;
  rc = scone_shard_gather_remap(h, T, seq_begin, seq_end, &ell, &scales, s);
  if (rc) return rc;
  embed_args a = {};
  fill_table_view(h, a.tv);
  unsigned long long n_head = 0;
  a.tv.st.hot = scone_shard_head(h, &n_head);  // row store: [replicated head | gathered records], both in record layout
  a.tv.st.n_hot = n_head;
  a.tv.st.cold = n_total ? reinterpret_cast<uint8_t *>(const_cast<void *>(d_records_base)) : reinterpret_cast<uint8_t *>(h->d_zero_row);
  a.tv.st.row_bytes = (unsigned int)scone_shard_rec_bytes(h);
  a.tv.scales = reinterpret_cast<const __half *>(scales);
  a.tv.row_begin = 0, a.tv.row_end = (long long)(n_head + (n_total ? n_total : 1));
  const long long t0 = (long long)seq_begin * T;
  const size_t esz = out_dtype == SCONE_DT_F32 ? 4 : 2;
  a.BT = nt, a.ntok = nt, a.T = T, a.max_n = h->cfg.max_n;
  a.ell = ell, a.zero_row = h->d_zero_row, a.mode = (int)h->cfg.lookup_mode;
  a.tok = d_tok + t0, a.pos = d_pos ? d_pos + t0 : nullptr;
  a.wte = d_wte, a.vocab = vocab, a.wpe = d_wpe, a.n_pos = n_pos;
  a.reduce = reduce, a.out = reinterpret_cast<uint8_t *>(d_out) + (size_t)(t0 - out_tok0) * h->cfg.dim * esz, a.status = h->d_status;
  return launch_fmt(h, a, SRC_HITS, MODE_FULL, out_dtype, s);
}

// Columns exchange (scone_shard.hip): the received payload rows are read in place at the table's own stride, the scales come
// from the caller's [head scales | received scales] buffer, the id lists are resolved through the senders' hash fragments.
extern "C" int scone_shard_cols_embed(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t seq_begin,
                                      int32_t seq_end, const void *d_rows, uint64_t n_total, const void *d_scales_full,
                                      const void *d_frags, uint64_t frag_slots_total, const uint64_t *h_frag_off,
                                      const uint64_t *h_frag_slots, const uint64_t *h_rec_base, const uint64_t *h_row_lo,
                                      int32_t world, const void *d_wte, int64_t vocab,
                                      const void *d_wpe, int64_t n_pos, const int32_t *d_pos, int32_t reduce, void *d_out,
                                      int64_t out_tok0, int32_t out_dtype, scone_stream_t stream) {
  int rc = need_table(h, "scone_shard_cols_embed: handle has no table (dim == 0)");
  if (rc) return rc;
  if (!h->shard) return scone_fail(h, SCONE_ESTATE, "scone_shard_cols_embed: call scone_shard_gather_plan first");
  int32_t pB = 0, pT = 0;
  scone_shard_plan_shape(h, &pB, &pT);
  if (B < 0 || T <= 0 || B != pB || T != pT || seq_begin < 0 || seq_end < seq_begin || seq_end > B || (n_total && !d_rows) ||
      !d_frags || !h_frag_off || !h_frag_slots || !h_rec_base || out_tok0 < 0 || out_tok0 > (long long)seq_begin * T ||
      (h->scale_bytes_per_row && !d_scales_full))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: bad argument (B, T must be the planned batch's)");
  if (!scone_wave_kernel_covers(h->cfg.table_fmt, h->cfg.dim))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: needs d % 8 == 0");
  if (reduce != SCONE_REDUCE_MEAN && reduce != SCONE_REDUCE_SUM)
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: bad reduce");
  const long long nt = (long long)(seq_end - seq_begin) * T;
  if (nt == 0) return SCONE_OK;
  if (!d_tok || !d_out) return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: null pointer");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  const int32_t *ell = nullptr;
  const uint8_t *head_p = nullptr;
  unsigned long long n_head = 0;
  rc = scone_shard_cols_remap(h, T, seq_begin, seq_end, d_frags, frag_slots_total, h_frag_off, h_frag_slots, h_rec_base, h_row_lo,
                              world, n_total, &ell, &head_p, &n_head, s);
  if (rc) return rc;
  embed_args a = {};
  fill_table_view(h, a.tv);
  a.tv.st.hot = const_cast<uint8_t *>(head_p);  // row store: [replicated head | received rows], both at the payload stride
  a.tv.st.n_hot = n_head;
  a.tv.st.cold = n_total ? reinterpret_cast<uint8_t *>(const_cast<void *>(d_rows)) : reinterpret_cast<uint8_t *>(h->d_zero_row);
  a.tv.st.row_bytes = (unsigned int)h->row_payload_bytes;
  a.tv.scales = reinterpret_cast<const __half *>(d_scales_full);
  a.tv.row_begin = 0, a.tv.row_end = (long long)(n_head + (n_total ? n_total : 1));
  const long long t0 = (long long)seq_begin * T;
  const size_t esz = out_dtype == SCONE_DT_F32 ? 4 : 2;
  a.BT = nt, a.ntok = nt, a.T = T, a.max_n = h->cfg.max_n;
  a.ell = ell, a.zero_row = h->d_zero_row, a.mode = (int)h->cfg.lookup_mode;
  a.tok = d_tok + t0, a.pos = d_pos ? d_pos + t0 : nullptr;
  a.wte = d_wte, a.vocab = vocab, a.wpe = d_wpe, a.n_pos = n_pos;
  a.reduce = reduce, a.out = reinterpret_cast<uint8_t *>(d_out) + (size_t)(t0 - out_tok0) * h->cfg.dim * esz, a.status = h->d_status;
  return launch_fmt(h, a, SRC_HITS, MODE_FULL, out_dtype, s);
}

extern "C" int scone_shard_gather_embed(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, const void *d_records,
                                        uint64_t n_records, const void *d_wte, int64_t vocab, const void *d_wpe, int64_t n_pos,
                                        const int32_t *d_pos, int32_t reduce, void *d_out, int32_t out_dtype,
                                        scone_stream_t stream) {
  int rc = scone_shard_gather_add_records(h, d_records, 0, n_records, n_records, stream);
  if (rc) return rc;
  if ((long long)B * T == 0) return SCONE_OK;
  return scone_shard_gather_embed_range(h, d_tok, B, T, 0, B, d_records, n_records, d_wte, vocab, d_wpe, n_pos, d_pos, reduce,
                                        d_out, 0, out_dtype, stream);
}

extern "C" int scone_embed_partial(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, float *d_partial,
                                   int32_t *d_counts, scone_stream_t stream) {
  int rc = need_table(h, "scone_embed_partial: handle has no table (dim == 0)");
  if (rc) return rc;
  if (B < 0 || T < 0) return scone_fail(h, SCONE_EINVAL, "scone_embed_partial: negative B or T");
  const long long BT = (long long)B * T;
  if (BT == 0) return SCONE_OK;
  if (!d_tok || !d_partial || !d_counts) return scone_fail(h, SCONE_EINVAL, "scone_embed_partial: null pointer");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  embed_args a = {};
  fill_table_view(h, a.tv);
  a.BT = BT, a.T = T, a.max_n = h->cfg.max_n;
  a.tok_begin = 0, a.ntok = BT;
  a.zero_row = h->d_zero_row, a.tok = d_tok, a.mode = (int)h->cfg.lookup_mode;
  rc = check_mode(h, "scone_embed_partial: lookup_mode longest_suffix needs d % 8 == 0");
  if (rc) return rc;
  scone_ws_lock ws(scone_ws_acquire(h, s));
  if (!ws.w) return scone_fail(h, SCONE_ENOMEM, "scone_embed_partial: out of memory");
  if (scone_wave_kernel_covers(h->cfg.table_fmt, h->cfg.dim)) {
    rc = scone_ensure_ell(h, ws.w, BT);
    if (rc) return rc;
    rc = scone_launch_match_ell(h, d_tok, B, T, ws.w->d_ell, s);
    if (rc) return rc;
    a.ell = ws.w->d_ell;
  } else {
    rc = scone_ensure_hits(h, ws.w, BT);
    if (rc) return rc;
    rc = scone_launch_match(h, d_tok, B, T, ws.w->d_hits, s);
    if (rc) return rc;
    a.hits = ws.w->d_hits;
  }
  a.reduce = SCONE_REDUCE_SUM, a.partial = d_partial, a.counts = d_counts, a.status = h->d_status;
  return launch_fmt(h, a, SRC_HITS, MODE_PARTIAL, SCONE_DT_F32, s);
}

extern "C" int scone_finalize(scone_handle *h, const float *d_sum, const int32_t *d_counts, const int32_t *d_tok,
                              int32_t B, int32_t T, int64_t tok_begin, int64_t tok_end, const void *d_wte,
                              int64_t vocab, const void *d_wpe, int64_t n_pos, const int32_t *d_pos, int32_t reduce,
                              void *d_out, int32_t out_dtype, scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (h->cfg.dim <= 0) return scone_fail(h, SCONE_ESTATE, "scone_finalize: handle has no table (dim == 0)");
  const long long BT = (long long)B * T;
  if (B < 0 || T < 0 || tok_begin < 0 || tok_end < tok_begin || tok_end > BT)
    return scone_fail(h, SCONE_EINVAL, "scone_finalize: bad token range");
  if (tok_end == tok_begin) return SCONE_OK;
  if (!d_sum || !d_counts || !d_out || ((d_wte || d_wpe) && !d_tok))
    return scone_fail(h, SCONE_EINVAL, "scone_finalize: null pointer");
  if (reduce != SCONE_REDUCE_MEAN && reduce != SCONE_REDUCE_SUM)
    return scone_fail(h, SCONE_EINVAL, "scone_finalize: bad reduce");
  SCONE_ON_DEVICE(h);
  embed_args a = {};
  fill_table_view(h, a.tv);
  a.BT = BT, a.T = T, a.max_n = h->cfg.max_n;
  a.tok_begin = tok_begin, a.ntok = tok_end - tok_begin;
  a.tok = d_tok, a.pos = d_pos, a.wte = d_wte, a.vocab = vocab, a.wpe = d_wpe, a.n_pos = n_pos;
  a.reduce = reduce, a.out = d_out, a.sums = d_sum, a.counts = const_cast<int32_t *>(d_counts);
  a.zero_row = h->d_zero_row, a.mode = (int)h->cfg.lookup_mode;
  a.status = h->d_status;
  return launch_fmt(h, a, SRC_HITS, MODE_FINALIZE, out_dtype, (hipStream_t)stream);
}
