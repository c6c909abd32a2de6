/*
 * scone_hip.h -- C ABI of libscone_hip.so: the MI355X (gfx950) f-gram embedding
 * lookup / aggregation path.
 *
 * The reference (llmsresearch/scone) is pure Python and has no FFI of its own.
 * Each entry point below names the reference code it replaces (file:line under
 * the reference checkout); INTEGRATION.md shows the ctypes stub a maintainer
 * would add on the reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross this boundary
 *   - every function returns 0 (SCONE_OK) or a negative errno-style code and
 *     never throws; scone_last_error(h) holds a human-readable message
 *   - the caller owns every buffer it passes; the library owns the index, the
 *     table and its workspaces
 *   - "d_" pointers are device pointers on the handle's device, "h_" pointers
 *     are host pointers; all device work is enqueued on the given stream
 *     (hipStream_t passed as void*; NULL = the default stream) and the call
 *     returns without synchronising unless documented otherwise
 *   - index and table are immutable on the lookup path, and lookups are thread-safe and stream-ordered: scone_embed
 *     (any batch size), scone_match, scone_match_csr, scone_gather_reduce, scone_embed_partial, scone_finalize and
 *     scone_table_gather_rows may be called concurrently from several host threads on one handle, on the same or on
 *     different streams.  Batches of up to 32768 tokens at d = 768 / 1024 / 1280 take one launch and no workspace; larger
 *     ones use a workspace that belongs to the STREAM of the call (created on first use, grown on demand or by
 *     scone_reserve) and is locked while the call enqueues its kernels.  Lookups and scone_embed_prefetch on a table
 *     created with stage_tokens > 0 share ONE staging pipeline per handle: they may be called from several threads, and are
 *     serialised on a lock of the handle for the length of the call (host side; their device work follows each other through
 *     the pipeline's events).  Not concurrent on one handle: index / table mutation against lookups, the scone_shard_*
 *     exchange calls, and scone_destroy against anything
 *   - every entry point selects the handle's device and restores the caller's current device before it returns
 */
#ifndef SCONE_HIP_H
#define SCONE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCONE_ABI_VERSION 2

typedef struct scone_handle scone_handle;
typedef void *scone_stream_t; /* hipStream_t */

/* table row formats (defined by this library; the reference's cache is always
 * fp32, scone/inference/embedding_cache.py:86,134,139) */
enum {
  SCONE_FMT_F32 = 0, /* [N,d] float                                     row_bytes 4d      */
  SCONE_FMT_F16 = 1, /* [N,d] IEEE half                                 row_bytes 2d      */
  SCONE_FMT_I8 = 2,  /* [N,d] int8 + scales[N] half (per row)           row_bytes d+2     */
  SCONE_FMT_I4 = 3   /* [N,d/2] offset-binary nibbles (elem 2k = low)   row_bytes d/2+2*d/128
                        + scales[N,d/128] half (per 128-group)                            */
};
enum {
  SCONE_PLACE_HBM = 0,        /* rows in device memory                                              */
  SCONE_PLACE_PINNED_HOST = 1 /* rows >= cfg.hot_rows in pinned host DRAM mapped into the GPU, read over PCIe */
};
enum { SCONE_REDUCE_MEAN = 0, SCONE_REDUCE_SUM = 1 };
enum {
  SCONE_MODE_COVER = 0,         /* the reference CODE: every f-gram covering the token, aggregated and ADDED to
                                   the token embedding (n_gram_extractor.py:106-126, engine.py:250,
                                   language_model.py:242-243)                                             */
  SCONE_MODE_LONGEST_SUFFIX = 1 /* the PAPER (Algorithm 2, assets/algorithm.png): the longest f-gram of length
                                   >= 2 ENDING at the token REPLACES the token embedding; causal           */
};
enum { SCONE_DT_F32 = 0, SCONE_DT_F16 = 1, SCONE_DT_BF16 = 2 };

enum {
  SCONE_OK = 0,
  SCONE_ESTATE = -1,  /* call order / missing index or table            */
  SCONE_EHIP = -5,    /* a HIP runtime call failed                      */
  SCONE_ENOMEM = -12, /* allocation failed / index full                 */
  SCONE_ENODEV = -19, /* no usable GPU                                  */
  SCONE_EINVAL = -22, /* bad argument                                   */
  SCONE_ERANGE = -34  /* id / token / row outside the configured range  */
};

typedef struct scone_cfg {
  uint32_t struct_size;    /* = sizeof(scone_cfg)                                          */
  int32_t device;          /* HIP device ordinal                                           */
  int32_t max_n;           /* longest f-gram, 1..4 (NGramExtractor.max_n,
                              scone/tokenization/n_gram_extractor.py:39)                    */
  int32_t dim;             /* embedding_dim d (EmbeddingCache.embedding_dim, :45);
                              multiple of 4 (F32), 8 (F16), 16 (I8), 128 (I4); 0 = index only */
  int32_t table_fmt;       /* SCONE_FMT_*                                                  */
  int32_t placement;       /* SCONE_PLACE_*                                                */
  uint64_t n_rows;         /* N = number of f-grams (global)                               */
  uint64_t row_begin;      /* rows owned by this handle: [row_begin, row_end); the whole   */
  uint64_t row_end;        /* table is 0..N (row_end = 0 means N)                          */
  uint64_t index_capacity; /* hash slots (power of two >= 4); 0 = smallest power of two >= 2*N */
  uint64_t hot_rows;       /* SCONE_PLACE_PINNED_HOST only: global rows [0, hot_rows) stay in HBM, the
                              rest in pinned host DRAM.  f-gram ids are frequency-ordered
                              (Counter.most_common, n_gram_extractor.py:91-99), so the head of
                              the table takes most of the hits.                              */
  uint32_t lookup_mode;    /* SCONE_MODE_* for scone_embed / scone_embed_partial / scone_finalize           */
  uint32_t stage_tokens;   /* SCONE_PLACE_PINNED_HOST only.  0: the lookup kernel reads host rows in place over
                              PCIe.  > 0: prefetch through an HBM cache of cold rows -- the batch is processed in
                              chunks of about this many tokens; on side streams each chunk is matched, the cold rows
                              it references that are not cached take a cache slot (clock eviction) and are copied
                              host -> HBM once, while the previous chunk is reduced on the caller's stream out of
                              [hot head | cache].  The cache lives as long as the table: a row stays resident across
                              chunks, batches and calls until it is evicted (the reference's analogue: the page cache
                              under its memory-mapped table, embedding_cache.py:76-91,132-135)                  */
  uint64_t cache_rows;     /* stage_tokens > 0: row slots of that cache (payload + scale bytes of HBM each).  At least
                              what the pipeline needs -- 7 chunks' worst case (5 record sets in the ring + the chunk being
                              placed + one of slack), 42 x stage_tokens rows for max_n = 3: 11.0M rows = 5.8 GB of INT4
                              d = 1024 rows at stage_tokens = 262144 -- is always provisioned (0 = exactly that); never
                              more than the cold rows                                                              */
} scone_cfg;

/* ---- lifecycle ----------------------------------------------------------- */
int scone_abi_version(void);
const char *scone_strerror(int code);
/* Allocates the (empty) index and, when cfg->dim > 0, the table storage. */
int scone_create(const scone_cfg *cfg, scone_handle **out);
void scone_destroy(scone_handle *h);
const char *scone_last_error(const scone_handle *h);
/* Sticky device-side status bits raised by kernels (bit 0: token outside the
 * base-embedding vocabulary, bit 1: f-gram id outside the table, bit 2: index
 * full, bit 3: the cache of cold rows of a pinned-host lookup had no evictable slot for a row -- sized so that it cannot
 * happen; the affected tokens read a wrong row); synchronises the stream, returns the bits in *bits and clears them. */
int scone_status(scone_handle *h, uint32_t *bits, scone_stream_t stream);
/* The cache of cold rows of a pinned-host table (cfg.stage_tokens > 0): its row slots, the rows copied host -> HBM since the
 * cache was created (every miss crosses PCIe once), the chunks prepared and the tokens per chunk actually used (smaller than
 * cfg.stage_tokens when the cache is small).  All zero before the first lookup.  Synchronises the device. */
int scone_stage_counters(scone_handle *h, uint64_t *cache_rows, uint64_t *rows_copied, uint64_t *chunks, uint64_t *chunk_tokens);

/* ---- index: f-gram -> id (replaces NGramExtractor.f_grams / f_gram_to_id,
 *      n_gram_extractor.py:42-44, and the id map of embedding_cache.py:173) --- */
/* keys[n, max_n] uint32 token ids (first lens[i] used), lens[n] in 1..max_n;
 * key i gets id = id0 + i.  May be called repeatedly.  On a duplicate key the
 * smallest id wins.  Host-pointer variant copies through a staging buffer and
 * synchronises; the device variant is stream-ordered. */
int scone_index_build(scone_handle *h, const uint32_t *h_keys, const uint8_t *h_lens,
                      uint64_t n, uint64_t id0);
int scone_index_build_device(scone_handle *h, const uint32_t *d_keys, const uint8_t *d_lens,
                             uint64_t n, uint64_t id0, scone_stream_t stream);
/* Synchronises.  n_keys = distinct keys stored, n_dups = duplicate insertions seen. */
int scone_index_stats(scone_handle *h, uint64_t *n_keys, uint64_t *capacity, uint64_t *n_dups);

/* The built index as three host blobs -- hash slots, direct unigram table, presence bitmap -- so that a table file can
 * carry it (scone_amd's native file: EmbeddingCache.save_native(with_index=True)) and a shard of a 1e9-key table loads
 * without rebuilding it from the keys.  Import needs a handle created with the same max_n and index_capacity (compare
 * scone_index_blob_sizes) and replaces the index; both synchronise the device. */
int scone_index_blob_sizes(scone_handle *h, uint64_t *slot_bytes, uint64_t *uni_bytes, uint64_t *bloom_bytes);
int scone_index_export(scone_handle *h, void *h_slots, void *h_uni, void *h_bloom, uint64_t *n_keys);
int scone_index_import(scone_handle *h, const void *h_slots, const void *h_uni, const void *h_bloom, uint64_t n_keys);

/* ---- vocabulary construction on the GPU (NGramExtractor.fit, n_gram_extractor.py:72-104) -----
 * d_tokens[n_tokens] int32: the corpus, texts back to back; d_text_offsets[n_texts+1] int64.
 * Counts every n-gram (n = 1..max_n, windows inside one text), keeps those with
 * count >= min_freq, orders them by count descending with ties in first-insertion order
 * (Counter.most_common: texts in order; per text all 1-grams, then all 2-grams, ...;
 * extract_all_n_grams :59-70) and writes the first min(max_f_grams, out_cap) as
 * d_keys_out[*, max_n] / d_lens_out[*] (/ d_counts_out[*], optional): row r is f-gram id r.
 * *h_n_out = rows written, *h_n_distinct = distinct n-grams seen (optional).  Synchronises. */
int scone_fit(int32_t device, const int32_t *d_tokens, int64_t n_tokens, const int64_t *d_text_offsets,
              int64_t n_texts, int32_t max_n, uint32_t min_freq, uint64_t max_f_grams,
              uint32_t *d_keys_out, uint8_t *d_lens_out, uint32_t *d_counts_out, uint64_t out_cap,
              uint64_t *h_n_out, uint64_t *h_n_distinct, scone_stream_t stream);

/* ---- table: rows (replaces EmbeddingCache.cache_embeddings storage,
 *      embedding_cache.py:56-111) ------------------------------------------- */
/* Raw rows already in the handle's format.  rows: nrows * payload bytes
 * (4d / 2d / d / d/2); scales: NULL (F32,F16), half[nrows] (I8), half[nrows, d/128] (I4).
 * src_is_device selects host or device source pointers.  Rows are global ids and must
 * lie inside [row_begin,row_end). */
int scone_table_upload(scone_handle *h, const void *rows, const void *scales, uint64_t row0,
                       uint64_t nrows, int src_is_device, scone_stream_t stream);
/* Inverse of scone_table_upload: raw payload rows (+ scales) of global rows [row0, row0+nrows)
 * in the handle's format, to host or device buffers; synchronises for host destinations.
 * With it a quantised table can be saved and reloaded without re-quantising. */
int scone_table_download(scone_handle *h, void *rows, void *scales, uint64_t row0, uint64_t nrows,
                         int dst_is_device, scone_stream_t stream);
/* fp32 rows on the device -> the handle's format (quantised on the GPU). */
int scone_table_store_f32(scone_handle *h, const float *d_rows_f32, uint64_t row0, uint64_t nrows,
                          scone_stream_t stream);
/* Scattered variant: row i of d_rows_f32 goes to global row d_ids[i]
 * (cache_embeddings(f_gram_ids, embeddings), embedding_cache.py:98-99,110-111). */
int scone_table_store_f32_ids(scone_handle *h, const float *d_rows_f32, const int64_t *d_ids,
                              uint64_t nrows, scone_stream_t stream);
/* Counter-based synthetic rows (bench / full-size tests); any row can be
 * recomputed on the host (oracle/ref_port.py synth_*). */
int scone_table_fill_synthetic(scone_handle *h, uint32_t seed, float base_scale,
                               scone_stream_t stream);
/* Row gather: out[i,:] = dequantised row d_ids[i] as fp32
 * (EmbeddingCache.get_embeddings, embedding_cache.py:113-147). */
int scone_table_gather_rows(scone_handle *h, const int64_t *d_ids, uint64_t n, float *d_out,
                            scone_stream_t stream);

/* ---- hot path ------------------------------------------------------------ */
/* Per-window membership + id (NGramExtractor.get_token_f_grams,
 * n_gram_extractor.py:106-126, one entry per (n, start)):
 * d_hits[(n-1)*B*T + b*T + i] = id of tok[b, i:i+n] or -1. */
int scone_match(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_hits,
                scone_stream_t stream);
/* Per-position id lists in the reference's order (n ascending, start ascending,
 * duplicates kept) as CSR over the B*T positions: d_offsets[B*T+1] (int32),
 * d_ids[ids_cap] (int32).  Writes the total to *h_total after synchronising the
 * stream; returns SCONE_ERANGE (and a valid *h_total) if ids_cap is too small. */
int scone_match_csr(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T,
                    int32_t *d_offsets, int32_t *d_ids, int64_t ids_cap, int64_t *h_total,
                    scone_stream_t stream);
/* Aggregate precomputed id lists: out[t,:] = reduce_k dequant(row ids[offsets[t]+k])
 * (+ base[t,:] when d_base != NULL, same dtype as out); zeros where the list is
 * empty (engine.py:247-259). */
int scone_gather_reduce(scone_handle *h, const int32_t *d_offsets, const int32_t *d_ids,
                        int64_t ntok, const void *d_base, int32_t reduce, void *d_out,
                        int32_t out_dtype, scone_stream_t stream);
/* Fused match + gather + dequantise + reduce + combine
 * (n_gram_extractor.py:106-126 -> embedding_cache.py:113-181 -> engine.py:234-266 ->
 *  language_model.py:239-254 with the projection folded into the table):
 *   out[b,i,:] = cast( (wte[tok[b,i]] + reduce_k row_k) + wpe[pos[b,i]] )
 * d_wte / d_wpe: [vocab,d] / [n_pos,d] in out_dtype, or NULL (term omitted);
 * d_pos: int32 [B,T] or NULL (= arange(T), language_model.py:248-251).
 * Batches of up to 32768 tokens (environment SCONE_FUSED_MAX_TOKENS, read by scone_create) at d = 768 / 1024 / 1280
 * run as ONE launch without any workspace; larger ones use the id-record workspace of the call's stream (grown on first
 * use / by scone_reserve).
 * Stream-ordered, no hidden synchronisation -- with ONE exception: the FIRST lookup (or scone_embed_prefetch) of a pinned-host
 * table with cfg.stage_tokens > 0 on a given caller stream synchronises that stream once, for ~1 ms: the staging pipeline
 * chooses its two side streams among six candidates by MEASURED overlap with the caller's stream (HIP multiplexes streams onto a
 * few hardware queues; side streams that share the caller's queue would serialise the prefetch behind the lookups). */
int scone_embed(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, const void *d_wte,
                int64_t vocab, const void *d_wpe, int64_t n_pos, const int32_t *d_pos,
                int32_t reduce, void *d_out, int32_t out_dtype, scone_stream_t stream);
/* Pinned-host tables with a prefetch pipeline (cfg.stage_tokens > 0): start fetching for the NEXT batch now.  The first chunks
 * of (d_tok, B, T) are matched, their missing cold rows placed in the HBM cache and copied host -> HBM on the handle's side
 * streams, ordered behind `stream` (the stream on which the tokens are produced) -- or, tokens_ready != 0, behind nothing: the
 * tokens are complete (uploaded and synchronised earlier), and the prefetch may run BESIDE a lookup queued on `stream` just
 * before, which is the point of calling it early -- and the call returns.  The scone_embed of exactly that batch (same pointer
 * and shape; the tokens must not change in between; any stream) takes the prepared chunks over instead of starting its pipeline
 * cold: a loop that calls scone_embed_prefetch(next) right after scone_embed(current) never pays the pipeline fill.  One
 * announcement may be pending; one that is never used is dropped by the next call (the rows it cached stay cached).  Results are
 * bit-identical with and without the call.  Thread-safe like scone_embed (serialised on the handle's staging lock).
 * Any other handle (rows in HBM, or read in place): a no-op.  There the match of the next batch could run ahead; built and
 * measured in round 5 -- k_match_ell on a side stream beside the previous gather is 1-19 % SLOWER than match-then-gather on
 * one stream at every batch size (the gather holds every wave slot; profiles/r05b, r05c) -- and removed.
 * (The north-star's "async prefetch"; the reference's memory-mapped table has no counterpart -- embedding_cache.py:132-135
 * faults rows in on first use.) */
int scone_embed_prefetch(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t tokens_ready, scone_stream_t stream);
/* Every per-stream workspace of the handle (those that exist, the default stream's -- created here -- and any created
 * later) holds at least max_tokens tokens: nothing is allocated inside a timed region afterwards. */
int scone_reserve(scone_handle *h, int64_t max_tokens);
/* CU reserve (new here; the reference is a single-stream Python loop).  A large-batch lookup fills every wave slot of the
 * chip for the length of the kernel, so a kernel another stream launches meanwhile -- the send / recv channels of an RCCL
 * collective in the sharded step, a copy kernel -- only gets in where lookup workgroups retire.  With n_reserved > 0 (a
 * multiple of 8: one CU per XCD; 0 switches it off) the lookup kernels of scone_embed (batches above the one-launch limit),
 * scone_embed_partial and the scone_shard_*_embed* calls run on a stream of the handle whose CU mask leaves n_reserved
 * compute units free, ordered into the caller's stream by two events (everything queued on the caller's stream before the
 * call precedes the lookup, everything queued after follows it); their grids are sized for the remaining CUs.  The two
 * cross-stream events cost ~75 us per lookup on MI355X / ROCm 7.2 (measured, profiles/r04c): a caller that owns its loop
 * queues it on the masked stream itself -- scone_lookup_stream returns it (NULL without a reserve; it lives until the next
 * scone_set_cu_reserve / scone_destroy) -- and a lookup called on THAT stream is launched there directly, no events.  Not
 * concurrent with lookups on the same handle (like every table mutation); synchronises the previous masked stream. */
int scone_set_cu_reserve(scone_handle *h, int32_t n_reserved);
int scone_get_cu_reserve(scone_handle *h, int32_t *n_reserved, int32_t *n_cus);
int scone_lookup_stream(scone_handle *h, void **stream);
/* Do kernels queued on stream_b START while a kernel queued before them on stream_a is still running?  HIP multiplexes a
 * process's streams onto a few hardware queues (four by default); two streams that share a queue run one after the other, and
 * which streams share depends on how many streams the process created before (profiles/r06i).  A caller that builds its own
 * overlap around the lookup -- the split-phase exchange of a row-sharded table plans and packs on a side stream -- picks that
 * side stream among a few candidates with this probe (scone_amd.hip_backend.SconeTable.pick_side_stream); the staging
 * pipeline of a pinned-host table does the same internally.  An 80-us spinning kernel on stream_a, a time stamp on stream_b;
 * synchronises both streams; *overlap = 1 or 0.  (No counterpart in the reference: it has no streams.) */
int scone_streams_overlap(scone_handle *h, scone_stream_t stream_a, scone_stream_t stream_b, int32_t *overlap);
/* Optional timing of the gather/reduce kernel launched by scone_embed: while enabled, every
 * call brackets that kernel with HIP events on the launch stream (a ring of 1024 pairs).
 * scone_profile_read synchronises the device, returns the number of timed launches and
 * their summed duration in milliseconds since the last reset, and optionally resets. */
int scone_profile_enable(scone_handle *h, int enable);
int scone_profile_read(scone_handle *h, uint64_t *n_launches, double *total_ms, int reset);
/* The individual launch times (milliseconds, launch order) behind scone_profile_read's sum since the last reset:
 * writes min(*n, cap) of them to h_ms and the number available to *n (at most 65536 are kept).  Synchronises.
 * bench.py reports their min / median / max: the same binary runs this kernel 5-10 % apart from process to
 * process (placement of the big buffers), so a mean alone says little. */
int scone_profile_samples(scone_handle *h, float *h_ms, uint64_t cap, uint64_t *n);

/* ---- row-sharded tables (one handle per GPU; RCCL exchange is done by the
 *      caller between the two calls) ---------------------------------------- */
/* fp32 partial sums over the rows this handle owns + the full per-position hit
 * count (the index is replicated, so every rank knows K). */
int scone_embed_partial(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T,
                        float *d_partial, int32_t *d_counts, scone_stream_t stream);
/* out[t,:] = cast( (wte[tok[t]] + sum[t,:] / K_t) + wpe[pos[t]] ) for t in
 * [tok_begin, tok_end) of the flattened B*T positions; d_sum / d_counts / d_out
 * are indexed from tok_begin (slice-local), d_tok / d_pos are the full [B,T]. */
int scone_finalize(scone_handle *h, const float *d_sum, const int32_t *d_counts,
                   const int32_t *d_tok, int32_t B, int32_t T, int64_t tok_begin, int64_t tok_end,
                   const void *d_wte, int64_t vocab, const void *d_wpe, int64_t n_pos,
                   const int32_t *d_pos, int32_t reduce, void *d_out, int32_t out_dtype,
                   scone_stream_t stream);

/* ---- row-sharded tables, row exchanges (preferred over the partial-sum pair above: the wire carries quantised rows -- 528 B
 *      for an INT4 d=1024 row instead of a 4096-B fp32 partial sum per token and rank -- each DISTINCT row once per
 *      destination, and the receiving rank reduces them in the reference's order, so results are bit-identical to the
 *      unsharded table).  Tokens and index are replicated, so both ends of a transfer derive what is sent; the caller only
 *      moves the buffers.  Two forms, built from the same plan / pack / embed primitives below: the all-gather form (every
 *      rank reduces the whole batch out of [replicated head | the distinct rows of all ranks]) and the slice exchange (rank r
 *      reduces slice r: sequences [r*ceil(B/world), (r+1)*ceil(B/world)); scone_shard_gather_plan_chunks with n_chunks = world
 *      and dedup_across_chunks = 0 lists, per destination, the distinct rows of mine its slice references; ONE
 *      all_to_all_single of records).  (The first form of the slice exchange -- one record per reference, scone_shard_plan /
 *      _pack / _embed -- was superseded in round 2 and removed in round 4 together with ABI version 1.) ------------------ */
int scone_shard_record_bytes(scone_handle *h, uint64_t *bytes);
/* Replicated head (optional; call once after scone_create, before rows are stored).  Global rows [0, n_head) are
 * kept on EVERY shard in addition to the rows it owns and never cross xGMI.  f-gram ids are frequency-ordered
 * (Counter.most_common, n_gram_extractor.py:91-99): the head holds every unigram and the most frequent f-grams,
 * i.e. about half of all row references (every token has its unigram), and without it the rank that owns the head
 * sends ten times what the others send.  scone_table_fill_synthetic fills the head as well;
 * scone_shard_head_store_f32 quantises fp32 rows into it exactly as scone_table_store_f32 does (rows must lie in
 * [0, n_head); every rank stores the same rows). */
int scone_shard_set_head(scone_handle *h, uint64_t n_head);
int scone_shard_head_store_f32(scone_handle *h, const float *d_rows_f32, uint64_t row0, uint64_t nrows,
                               scone_stream_t stream);

/* ---- row-sharded tables, all-gather form: EVERY rank ends up with the whole [B, T, d] output.  Gathering the rows the
 *      batch references costs a quarter of the bytes of gathering the finished 2 KB vectors, and here every DISTINCT row
 *      crosses once however many tokens reference it: scone_shard_gather_plan claims the distinct rows this shard owns
 *      (outside the replicated head) that the batch references and returns their number (synchronises);
 *      scone_shard_gather_pack writes one record [row payload | scales | row id] per claimed row; the caller all-gathers
 *      the record buffers of all ranks (any order); scone_shard_gather_embed indexes the records by row id and reduces
 *      the whole batch out of [replicated head | records] -- bit-identical to the unsharded table. -------------------- */
int scone_shard_gather_plan(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, uint64_t *h_n_records,
                            scone_stream_t stream);
int scone_shard_gather_pack(scone_handle *h, void *d_send_buf, scone_stream_t stream);
int scone_shard_gather_embed(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, const void *d_records,
                             uint64_t n_records, const void *d_wte, int64_t vocab, const void *d_wpe, int64_t n_pos,
                             const int32_t *d_pos, int32_t reduce, void *d_out, int32_t out_dtype, scone_stream_t stream);
/* The same exchange cut into n_chunks (<= 64) runs of sequences (chunk c = sequences [c * ceil(B / n_chunks), ...)) so that
 * the all-gather of chunk c + 1 overlaps the reduction of chunk c:
 *   scone_shard_gather_plan_chunks   one match of the batch, one claim pass per chunk in chunk order.
 *                                    dedup_across_chunks = 1 (all-gather form): a row claimed by an earlier chunk is not
 *                                    claimed again.  dedup_across_chunks = 0 with n_chunks = world (slice exchange: chunk q =
 *                                    the sequences rank q finalises): every chunk lists each DISTINCT row of mine it
 *                                    references -- what I send to rank q, one record per row however many of its tokens
 *                                    reference it; the records go out with ONE all_to_all_single and the receiver uses
 *                                    _add_records + _embed_range for its own slice.  h_chunk_end[c] = records claimed by
 *                                    chunks 0..c (chunk c's are [h_chunk_end[c-1], h_chunk_end[c])); synchronises
 *   scone_shard_gather_pack_range    records [first, first + count) of the plan into d_send_buf, followed by `pad` padding
 *                                    records (row id 0xFFFFFFFF: an all-gather wants equal contributions; receivers skip them)
 *   scone_shard_gather_add_records   receiver: records [record0, record0 + n_records) of the gathered buffer
 *                                    (d_records_base = record 0; the buffer will hold n_total records in all, padding
 *                                    included) join the row map; record0 == 0 starts a new exchange
 *   scone_shard_gather_embed_range   sequences [seq_begin, seq_end) of the planned batch out of [replicated head | records
 *                                    added so far]; every row they reference must have been added.  d_out's first row
 *                                    is token out_tok0 of the flattened batch (0: d_out is the whole [B, T, d];
 *                                    seq_begin * T: d_out holds just this run).  B, T must be the planned batch's.  The
 *                                    id lists of a run are rewritten to record numbers in place, ONCE per plan: the slot
 *                                    remembers which sequences already hold record numbers, so a second call over the
 *                                    same or an overlapping range (a retry, other chunk bounds) reduces them as they are;
 *                                    a new exchange (_add_records with record0 == 0) on lists that were already rewritten
 *                                    is refused (SCONE_ESTATE): plan the batch again first.
 * The receiver's row map (row id -> record number) is an open-addressing hash map sized by the exchange (cache-resident). */
/* Plan slots (0 .. 3; 0 is active at first): the receiver-side state of a planned batch (its id lists, the scales of
 * [head | records], the row map) exists once per slot, so that a serving loop can plan, pack and exchange batch b + 1 (and
 * b + 2: the chain plan -> transfers must then fit TWO reductions, not one) on side streams while batch b is still being
 * reduced on the main stream.  A host-side switch, no device work; the
 * scone_shard_gather_plan* / _add_records / _embed_range / _embed calls that follow work on the selected slot.  The
 * sender-side scratch is shared: plan and pack of one batch must be enqueued before the next plan.  New here (the reference is one process). */
int scone_shard_select_slot(scone_handle *h, int32_t slot);
int scone_shard_gather_plan_chunks(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t n_chunks,
                                   int32_t dedup_across_chunks, uint64_t *h_chunk_end, scone_stream_t stream);
/* The match of a plan, sharded over the ranks.  In the all-gather form every rank needs the id lists of the WHOLE batch
 * (it reduces all of it), which makes the match against a 1e9-key index the largest helper kernel of the step.  Index and
 * tokens are replicated and matching is per sequence, so rank r matches only ITS run of sequences, the runs are
 * all-gathered by the caller (32 B per token for max_n <= 3: scone_ell_width ints per token) and the claim passes run over
 * the gathered lists:
 *   scone_shard_gather_match      sequences [seq_begin, seq_end) of the batch d_tok [B, T] against ALL rows -> their list
 *                                 records, (seq_end - seq_begin) * T * width int32 at d_ell_out; stream-ordered, no state
 *   scone_shard_gather_plan_ell   the claim passes of scone_shard_gather_plan_chunks over lists the caller supplies:
 *                                 d_ell [B * T, width] int32, records of the whole batch in token order.  The buffer is
 *                                 BORROWED by the selected plan slot until the batch has been reduced (_embed_range
 *                                 rewrites it in place); same chunk semantics, synchronises
 * New here (the reference matches one sequence at a time in one process: n_gram_extractor.py:106-126). */
int scone_ell_width(scone_handle *h, uint32_t *ints_per_token);
int scone_shard_gather_match(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t seq_begin,
                             int32_t seq_end, int32_t *d_ell_out, scone_stream_t stream);
int scone_shard_gather_plan_ell(scone_handle *h, int32_t *d_ell, int32_t B, int32_t T, int32_t n_chunks,
                                int32_t dedup_across_chunks, uint64_t *h_chunk_end, scone_stream_t stream);
int scone_shard_gather_pack_range(scone_handle *h, uint64_t first, uint64_t count, uint64_t pad, void *d_send_buf,
                                  scone_stream_t stream);
int scone_shard_gather_add_records(scone_handle *h, const void *d_records_base, uint64_t record0, uint64_t n_records,
                                   uint64_t n_total, scone_stream_t stream);
/* Rewrite the id lists of sequences [seq_begin, seq_end) of the planned batch to record numbers now, on `stream` (every row
 * they reference must have been added): a loop that receives on one stream and reduces on another does this where the
 * records arrive, so that scone_shard_gather_embed_range finds the lists done and only launches the lookup. */
int scone_shard_gather_remap_range(scone_handle *h, int32_t seq_begin, int32_t seq_end, scone_stream_t stream);
int scone_shard_gather_embed_range(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t seq_begin,
                                   int32_t seq_end, const void *d_records_base, uint64_t n_total, const void *d_wte,
                                   int64_t vocab, const void *d_wpe, int64_t n_pos, const int32_t *d_pos, int32_t reduce,
                                   void *d_out, int64_t out_tok0, int32_t out_dtype, scone_stream_t stream);

/* ---- all-gather form with COLUMNS on the wire (one plan = one exchange).  A rank's contribution travels as three arrays
 *      instead of [payload | scales | row id] records: the payload rows at the table's own stride (the lookup reads them in
 *      place, cache-line aligned), the scales (received straight into the caller's [head scales | scales of all ranks]
 *      buffer: nothing to unpack), and the SENDER's hash fragment row id -> position in its contribution (u64 slots; built
 *      while it packs: its own ~65k inserts instead of 0.45M 64-bit CAS on every receiver, ~36 us per step at C5's scale).
 *      The receiver has no indexing pass: a list entry is resolved in its owner's fragment (the owner follows from the id:
 *      contiguous ranges [r N / W, (r + 1) N / W)) and becomes rec_base[owner] + position.
 *   scone_shard_cols_frag_slots   slots of the fragment for `count` rows (a power of two >= 4 count, >= 64): both ends
 *                                 derive it from the exchanged counts
 *   scone_shard_cols_pack         records [first, first + count) of the plan (scone_shard_gather_plan*) as columns;
 *                                 d_frag_out [frag_slots] u64 is cleared and filled; stream-ordered
 *   scone_shard_cols_build_frag   the fragment of an arbitrary id list (position = index in the list): tools and tests
 *                                 stand in for the other ranks with it
 *   scone_shard_head_scales       the replicated head's scales [n_head, scale bytes] into d_out: the front of the scales
 *                                 buffer (once per buffer, not per step)
 *   scone_shard_head_version      a counter bumped by every change of the replicated head (scone_shard_set_head /
 *                                 _head_store_f32 / scone_table_fill_synthetic): a caller's scales buffer whose front was
 *                                 filled at another version takes scone_shard_head_scales again
 *   scone_shard_cols_embed        sequences [seq_begin, seq_end) of the planned batch out of [replicated head | d_rows
 *                                 [n_total, payload bytes]] with scales d_scales_full [n_head + n_total, scale bytes];
 *                                 d_frags holds frag_slots_total u64 slots; h_frag_off[r] (slots into d_frags),
 *                                 h_frag_slots[r], h_rec_base[r] (row number of rank r's first row in d_rows) for r < world
 *                                 (every fragment must lie inside d_frags: SCONE_EINVAL otherwise); h_row_lo[world + 1]:
 *                                 the owners' row ranges -- rank r owns [h_row_lo[r], h_row_lo[r + 1]), ascending from 0 to
 *                                 n_rows (tables of at most 2^32 rows) -- or NULL for the floor partition r * n_rows / world
 *                                 of scone_amd.distributed.shard_range.  Lists are rewritten once per plan, as in
 *                                 scone_shard_gather_embed_range; bit-identical to the unsharded table.
 * New here (the reference keeps its table in one process: embedding_cache.py:49-50). */
/* Sync-free plan (round 4; one chunk): scone_shard_gather_plan_async / _plan_ell_async enqueue match and claim passes and
 * return -- no count comes back to the host; scone_shard_cols_pack_cap packs up to cap_rows of the claimed rows (the capacity
 * both ends sized the transfer for, e.g. the previous batches' counts + 12.5 %), reading the count on the device, and writes
 * d_header_out [2] u64 = {rows claimed, 1 if that exceeded cap_rows}.  The caller ships the header with the columns, reads
 * it when the batch is reduced, and repeats a batch that overflowed with the exact-size calls above (the receivers' lookups
 * of an overflowed exchange raise SCONE_ST_BAD_ID: rows are missing, never read out of bounds). */
int scone_shard_gather_plan_async(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, scone_stream_t stream);
int scone_shard_gather_plan_ell_async(scone_handle *h, int32_t *d_ell, int32_t B, int32_t T, scone_stream_t stream);
int scone_shard_cols_pack_cap(scone_handle *h, uint64_t cap_rows, void *d_rows_out, void *d_scales_out, void *d_frag_out,
                              uint64_t frag_slots, void *d_header_out, scone_stream_t stream);
int scone_shard_cols_frag_slots(uint64_t count, uint64_t *slots);
int scone_shard_cols_pack(scone_handle *h, uint64_t first, uint64_t count, void *d_rows_out, void *d_scales_out,
                          void *d_frag_out, uint64_t frag_slots, scone_stream_t stream);
int scone_shard_cols_build_frag(scone_handle *h, const int32_t *d_ids, uint64_t count, void *d_frag_out, uint64_t frag_slots,
                                scone_stream_t stream);
int scone_shard_head_scales(scone_handle *h, void *d_out, scone_stream_t stream);
int scone_shard_head_version(scone_handle *h, uint64_t *version);
int scone_shard_cols_embed(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t seq_begin, int32_t seq_end,
                           const void *d_rows, uint64_t n_total, const void *d_scales_full, const void *d_frags,
                           uint64_t frag_slots_total, const uint64_t *h_frag_off, const uint64_t *h_frag_slots,
                           const uint64_t *h_rec_base, const uint64_t *h_row_lo, int32_t world, const void *d_wte, int64_t vocab, const void *d_wpe, int64_t n_pos, const int32_t *d_pos,
                           int32_t reduce, void *d_out, int64_t out_tok0, int32_t out_dtype, scone_stream_t stream);

/* ---- transport without kernels for the all-gather form (distributed.py: gather_transport="sdma").  The exchange sends exact
 *      contiguous ranges, so a rank can PUSH its columns into its peers' receive buffers with the copy engines (SDMA over
 *      xGMI) while every wave slot belongs to the lookup kernel -- RCCL's send / recv are kernels that must find room
 *      beside it.  One process per GPU: buffers and events are shared through HIP's interprocess handles (64 opaque
 *      bytes each, moved between the ranks by the caller, e.g. dist.all_gather).
 *   scone_ipc_alloc / _free           device memory on the handle's device + its interprocess handle (a rank's receive buffer)
 *   scone_ipc_open / _close           a peer's buffer mapped into this process (peer access is enabled on demand)
 *   scone_ipc_event_create / _open    an interprocess event (record in the owner, wait anywhere); _destroy frees either kind
 *   scone_ipc_event_record / _wait    stream-ordered.  NB a wait sees the most recent record AT THE TIME OF THE CALL: the
 *                                     caller orders "peer has called record" before "I call wait" itself (a host-side
 *                                     collective between the two, see distributed.py)
 *   scone_ipc_push                    bytes to a (peer-mapped) device pointer on `stream`; copy_engine != 0 forbids the
 *                                     blit-kernel fallback (hipMemcpyDeviceToDeviceNoCU)
 * New here (the reference's only multi-process set-up is training-side DDP, hydra_train.py:32-48). */
int scone_ipc_alloc(scone_handle *h, uint64_t bytes, void **d_ptr, void *handle64);
int scone_ipc_free(scone_handle *h, void *d_ptr);
int scone_ipc_open(scone_handle *h, const void *handle64, void **d_ptr);
int scone_ipc_close(scone_handle *h, void *d_ptr);
int scone_ipc_event_create(scone_handle *h, void **event, void *handle64);
int scone_ipc_event_open(scone_handle *h, const void *handle64, void **event);
int scone_ipc_event_destroy(scone_handle *h, void *event);
int scone_ipc_event_record(scone_handle *h, void *event, scone_stream_t stream);
int scone_ipc_event_wait(scone_handle *h, void *event, scone_stream_t stream);
int scone_ipc_push(scone_handle *h, void *d_dst, const void *d_src, uint64_t bytes, int32_t copy_engine, scone_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SCONE_HIP_H */
