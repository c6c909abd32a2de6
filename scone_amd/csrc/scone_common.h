// Internal declarations shared by the .hip translation units of libscone_hip.so.
// gfx950 only: 64-lane wavefronts are assumed throughout.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/scone_hip.h"

#define SCONE_I4_GROUP 128
#define SCONE_MAX_N 4
#define SCONE_PROF_RING 1024
#define SCONE_PROF_MAX_SAMPLES 65536  // per-launch times kept between two resets
#define SCONE_UNI_CAP (1 << 18)  // direct unigram table: token ids below this skip the hash probe
#define SCONE_MAX_CAND 10  // max_n (max_n + 1) / 2 at max_n = 4

#define SCONE_ST_BAD_TOKEN 1u
#define SCONE_ST_BAD_ID 2u
#define SCONE_ST_INDEX_FULL 4u
#define SCONE_ST_STAGE_OVERFLOW 8u

// ---------------------------------------------------------------- hash index
// 16-byte slot.  lo/ext hold the exact packed key (no fingerprints, so a probe
// never returns a wrong id):  hi = (ext << 32) | (id + 1).  Empty: lo == 0.
struct __attribute__((aligned(16))) scone_slot {
  unsigned long long lo;
  unsigned long long hi;
};

struct scone_key {
  unsigned long long lo;
  uint32_t ext;
  bool ok;  // false: some token cannot be represented -> cannot be in the index
};

// Tokens are stored +1 so that an absent position is 0 and no valid key is all-zero.
// max_n <= 3: 32 bits per token (96-bit key).  max_n == 4: 24 bits per token.
__host__ __device__ inline scone_key scone_pack_key(const uint32_t *t, int n, int max_n) {
  scone_key k;
  k.ok = true;
  if (max_n <= 3) {
    unsigned long long v0 = (unsigned long long)t[0] + 1ull;
    unsigned long long v1 = n > 1 ? (unsigned long long)t[1] + 1ull : 0ull;
    unsigned long long v2 = n > 2 ? (unsigned long long)t[2] + 1ull : 0ull;
    k.ok = (v0 >> 32) == 0 && (v1 >> 32) == 0 && (v2 >> 32) == 0;
    k.lo = v0 | (v1 << 32);
    k.ext = (uint32_t)v2;
  } else {
    uint32_t v[4];
    for (int i = 0; i < 4; ++i) {
      v[i] = i < n ? t[i] + 1u : 0u;
      if (i < n && t[i] >= 0xFFFFFFu) k.ok = false;
    }
    k.lo = (unsigned long long)v[0] | ((unsigned long long)v[1] << 24) |
           ((unsigned long long)(v[2] & 0xFFFFu) << 48);
    k.ext = (v[2] >> 16) | (v[3] << 8);
  }
  return k;
}

__host__ __device__ inline unsigned long long scone_hash_key(unsigned long long lo, uint32_t ext) {
  unsigned long long x = lo ^ ((unsigned long long)ext * 0x9E3779B97F4A7C15ull);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// Probe sequence.  The table is cut into buckets of SCONE_BUCKET consecutive slots (4 x 16 B = one 64-B
// sector, so a probe step costs the memory system the same as a single-slot read).  A key's home bucket is
// hash & (buckets - 1); when a bucket is full the sequence continues at bucket + step (step odd -> visits every
// bucket of the power-of-two table; taken from the high hash bits, so keys that collide on a bucket part again).
// Inside a bucket slots fill front to back, and a key only ever moves past FULL buckets, so a lookup stops at
// the first empty slot it sees.  Compared with slot-by-slot linear probing this bounds the dependent-load
// chain: at load 0.5 a bucket overflows with p ~ 0.05, so the longest chain over 10^6 lookups is ~5 steps
// where linear probing's longest cluster costs 30-40 dependent L2 misses and sets the match kernel's tail.
#define SCONE_BUCKET 4
#define SCONE_BUCKET_SHIFT 2
__host__ __device__ inline unsigned long long scone_bucket_home(unsigned long long hash, unsigned long long slot_mask) {
  return hash & (slot_mask >> SCONE_BUCKET_SHIFT);
}
__host__ __device__ inline unsigned long long scone_bucket_step(unsigned long long hash) { return (hash >> 32) | 1ull; }

// Presence filter in front of the hash probes of the fused match: bit (hash >> 20) & mask is set for
// every inserted key, so a clear bit proves the window is not an f-gram without touching the table
// (most bigram / trigram windows are misses, and every table probe costs a 128-B line).
__host__ __device__ inline unsigned long long scone_bloom_bit(unsigned long long hash, unsigned long long mask) {
  return (hash >> 20) & mask;
}

// lowbias32 finaliser; numpy twin: oracle/ref_port.py hash32
__host__ __device__ inline uint32_t scone_hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// INT4 group scales, physical order inside a row's scale array.  d = 1024: a lane of the wave-per-token kernel owns
// elements of group g = lane / 16 (first 512-element segment) and of group g + 4 (second segment); stored next to
// each other they are ONE dword load per row and lane instead of two 2-byte loads, and one VGPR instead of two
// (slot = 2 (g mod 4) + g / 4).  Every other d: logical order.  The C ABI (scone_table_upload / _download) speaks the
// logical order [N, d/128]; every kernel indexes through this function.
__host__ __device__ inline int scone_i4_scale_slot(int g, int d) { return d == 1024 ? (((g & 3) << 1) | (g >> 2)) : g; }

// Where a local row lives: rows [0, hot) in HBM, the rest (if any) in mapped pinned host memory.
struct scone_row_store {
  uint8_t *hot;
  uint8_t *cold;
  unsigned long long n_hot;
  unsigned int row_bytes;
  __host__ __device__ inline uint8_t *row(unsigned long long lr) const {
    return lr < n_hot ? hot + lr * row_bytes : cold + (lr - n_hot) * row_bytes;
  }
};

struct scone_stage_state;
struct scone_shard_state;

// Per-stream workspaces of the large-batch lookup (id records / dense hits / CSR scan sums).  A call takes the
// workspace of ITS stream and holds its lock while it enqueues the kernels that write and then read it, so lookups of
// any size may be issued concurrently from several host threads -- on different streams (different workspaces) or on
// one (the enqueue sequences do not interleave, and the stream orders their execution).
struct scone_ws {
  hipStream_t stream = nullptr;
  std::mutex mu;
  int32_t *d_hits = nullptr;
  int64_t hits_cap_tokens = 0;
  int32_t *d_ell = nullptr;  // per-token id lists for the fused lookup, [tokens, SCONE_ELL_W(max_n)]
  int64_t ell_cap_tokens = 0;
  int32_t *d_block_sums = nullptr;
  int64_t block_sums_cap = 0;
  int64_t *d_total = nullptr;
  uint64_t last_use = 0;  // handle clock at the last acquire (least recently used idle workspace is recycled)
};

// ---------------------------------------------------------------- handle
struct scone_handle {
  scone_cfg cfg;
  int device;
  int n_cus;  // compute units of the device (256 on MI355X)
  // CU reserve (scone_set_cu_reserve): the large-batch lookup kernel is launched on a stream of THIS handle whose CU mask
  // leaves `cu_reserve` compute units free, between two events that tie it into the caller's stream -- so that kernels of
  // other streams (RCCL's send / recv channels, copy kernels) always find whole CUs to run on while a lookup is resident
  int cu_reserve;
  hipStream_t lookup_stream;
  hipEvent_t lookup_in, lookup_out;
  std::mutex lookup_mu;  // held from the event on the caller's stream to the wait on it: one hop at a time per handle
  long long fused_max_tokens;  // batches up to this many tokens take the one-launch kernel (env SCONE_FUSED_MAX_TOKENS overrides)
  long long match_tile;        // > 0: positions per workgroup of k_match_ell fixed by env SCONE_MATCH_TILE (else whole residency rounds)
  // index
  scone_slot *slots;
  uint64_t cap;  // power of two
  unsigned long long *d_counters;  // [0] inserted, [1] duplicates
  uint32_t *d_status;              // sticky status bits
  int32_t *d_uni;                  // [SCONE_UNI_CAP] token -> unigram id, 0xFFFFFFFF (= -1) if none
  uint32_t *d_bloom;               // presence bitmap over key hashes (one bit per key, ~16 bits of room per key)
  unsigned long long bloom_mask;   // bits - 1 (power of two), 0 = filter disabled
  // table
  void *rows;        // payload rows in HBM: local rows [0, hot_local)
  void *rows_host;   // payload rows in pinned host DRAM: local rows [hot_local, local_rows) (or null)
  uint64_t hot_local;
  void *scales;      // I8: half[rows]; I4: half[rows, d/128]
  size_t row_payload_bytes;
  size_t scale_bytes_per_row;
  uint64_t local_rows;
  // workspaces: one set per stream that has issued a large lookup (scone_ws_acquire)
  std::mutex ws_mu;             // guards the list
  std::vector<scone_ws *> ws;
  int64_t reserve_tokens;       // scone_reserve: every workspace holds at least this many tokens
  uint64_t ws_clock;
  void *d_zero_row;  // dim * 4 zero bytes
  scone_stage_state *stage;  // staged host->HBM prefetch (scone_stage.hip), created on first use
  // one staging pipeline per handle: scone_embed / scone_embed_prefetch / scone_stage_counters on a stage_tokens > 0 table hold
  // this lock for the whole call, so calls from several host threads are serialised (their device work is ordered by the
  // pipeline's events exactly as that of consecutive calls from one thread)
  std::mutex stage_mu;
  scone_shard_state *shard;  // row exchange between shards (scone_shard.hip), created on first use
  // optional kernel timing (scone_profile_*); prof_mu is held from the begin event to the end event of a launch
  std::mutex prof_mu;
  std::mutex err_mu;
  std::atomic<bool> prof_on;  // read without the lock by every launch (scone_prof_begin): off = no mutex traffic at all
  hipEvent_t *prof_ev;  // [2 * SCONE_PROF_RING]
  uint64_t prof_head;   // pairs recorded since the last drain
  uint64_t prof_n;      // launches accumulated
  double prof_ms;
  std::vector<float> prof_samples;  // per-launch milliseconds since the last reset (scone_profile_samples)
  std::string err;
};

int scone_fail(scone_handle *h, int code, const char *what);
int scone_hip_fail(scone_handle *h, hipError_t e, const char *what);
scone_row_store scone_store_of(const scone_handle *h);
struct scone_index_view;
void scone_index_view_of(const scone_handle *h, scone_index_view *v);  // scone_index.hip
// The workspace of stream s, LOCKED (created on first use); release with scone_ws_release or hold it in a scone_ws_lock.
scone_ws *scone_ws_acquire(scone_handle *h, hipStream_t s);
inline void scone_ws_release(scone_ws *w) {
  if (w) w->mu.unlock();
}
struct scone_ws_lock {
  scone_ws *w;
  explicit scone_ws_lock(scone_ws *ws) : w(ws) {}
  ~scone_ws_lock() { scone_ws_release(w); }
  scone_ws_lock(const scone_ws_lock &) = delete;
  scone_ws_lock &operator=(const scone_ws_lock &) = delete;
};
int scone_ensure_hits(scone_handle *h, scone_ws *w, int64_t ntok);
int scone_ensure_ell(scone_handle *h, scone_ws *w, int64_t ntok);
// CU reserve: where a large lookup is launched.  enter: *launch = s (no reserve), or the handle's masked stream, made to wait
// for everything queued on s; leave: s waits for what was launched there.  enter takes lookup_mu when it hops, leave drops it.
int scone_lookup_enter(scone_handle *h, hipStream_t s, hipStream_t *launch);
int scone_lookup_leave(scone_handle *h, hipStream_t s, hipStream_t launch);
static inline int scone_lookup_cus(const scone_handle *h) { return h->n_cus - h->cu_reserve > 0 ? h->n_cus - h->cu_reserve : 1; }
int scone_prof_begin(scone_handle *h, hipStream_t s);  // no-ops unless profiling is enabled; begin takes prof_mu, end drops it
int scone_prof_end(scone_handle *h, hipStream_t s);
void scone_prof_abort(scone_handle *h);                // a launch failed between begin and end

// A dispatch carries its size in WORK-ITEMS in a 32-bit field: blocks x threads must stay below 2^32, beyond that
// the grid silently wraps (found by a 100M-row table whose one-wave-per-row synthetic fill stopped at row 33.5M
// without any error).  Row- and key-parallel kernels therefore walk their items with a grid-stride loop over a
// capped grid (SCONE_MAX_BLOCKS); token-parallel launches check scone_grid_fits().
#define SCONE_MAX_BLOCKS (1u << 20)
static inline bool scone_grid_fits(unsigned long long blocks, unsigned threads) { return blocks * threads < (1ull << 32); }
static inline unsigned scone_capped_blocks(unsigned long long blocks) {
  return (unsigned)(blocks < SCONE_MAX_BLOCKS ? (blocks ? blocks : 1) : SCONE_MAX_BLOCKS);
}

// Every entry point runs on the handle's device and leaves the CALLER's current device as it found it (a process
// may hold tables on several GPUs, or have another device current when a handle is used or garbage-collected).
struct scone_device_guard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit scone_device_guard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~scone_device_guard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  scone_device_guard(const scone_device_guard &) = delete;
  scone_device_guard &operator=(const scone_device_guard &) = delete;
};
#define SCONE_ON_DEVICE(h)                                                                  \
  scone_device_guard dev_guard__((h)->device);                                              \
  if (dev_guard__.err != hipSuccess) return scone_hip_fail((h), dev_guard__.err, "hipSetDevice")

#define SCONE_HIP(h, call)                                       \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) return scone_hip_fail((h), e__, #call); \
  } while (0)

// launchers implemented in the kernel translation units
int scone_launch_match(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_hits,
                       hipStream_t s);
int scone_launch_match_ell(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_ell,
                           hipStream_t s);
// explicit owned range; keep_pos: owned ids stay at their index in the full list (holes = -1)
int scone_launch_match_ell_ex(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t *d_ell,
                              long long row_begin, long long row_end, int keep_pos, hipStream_t s);
#define SCONE_ELL_W(max_n) ((max_n) <= 3 ? 8 : 16)

// table kernels on an arbitrary row store (scone_table.hip); used for the replicated head of a shard
int scone_store_f32_into(scone_handle *h, const scone_row_store &st, void *scales, uint64_t row_begin, uint64_t row_end,
                         const float *d_src, const int64_t *d_ids, uint64_t row0, uint64_t nrows, hipStream_t s);
int scone_fill_synth_into(scone_handle *h, const scone_row_store &st, void *scales, uint64_t row_begin, uint64_t nrows,
                          uint32_t seed, float base_scale, hipStream_t s);
// row exchange between shards (scone_shard.hip)
int scone_shard_fill_head_synth(scone_handle *h, uint32_t seed, float base_scale, hipStream_t s);  // no-op without a head
void scone_shard_destroy(scone_handle *h);
int scone_shard_rec_bytes(const scone_handle *h);
uint8_t *scone_shard_head(const scone_handle *h, unsigned long long *n_head);  // replicated head rows (record layout) or null
int scone_shard_gather_add(scone_handle *h, const void *d_records, uint64_t n, uint64_t record0, uint64_t n_total, hipStream_t s);
int scone_shard_gather_remap(scone_handle *h, int32_t T, int32_t seq0, int32_t seq1, const int32_t **ell, const void **scales,
                             hipStream_t s);
int scone_shard_plan_shape(const scone_handle *h, int32_t *B, int32_t *T);
int scone_shard_cols_remap(scone_handle *h, int32_t T, int32_t seq0, int32_t seq1, const void *d_frags, uint64_t frag_slots_total,
                           const uint64_t *h_frag_off, const uint64_t *h_frag_slots, const uint64_t *h_rec_base,
                           const uint64_t *h_row_lo, int32_t world, uint64_t n_total,
                           const int32_t **ell, const uint8_t **head_p, unsigned long long *n_head_out, hipStream_t s);

// staged prefetch of host-resident rows (scone_stage.hip)
// record sets of the chunk pipeline: a chunk is prepared SCONE_STAGE_AHEAD chunks before it is looked up, and up to
// SCONE_STAGE_AHEAD chunks of the NEXT batch may have been prepared by scone_embed_prefetch -> 2 + 2 + the one being reduced
#define SCONE_STAGE_NBUF 5
#define SCONE_STAGE_AHEAD 2
void scone_stage_destroy(scone_handle *h);
int scone_stage_prepare(scone_handle *h, long long chunk_tokens);
int scone_stage_bind(scone_handle *h, hipStream_t caller);  // choose PREP / COPY among the candidate streams by measured overlap with `caller`
int scone_stage_chunk(scone_handle *h, const int32_t *d_tok, int32_t Bc, int32_t T);  // into the next set of the ring
int scone_stage_consume_buf(scone_handle *h);                                         // the set of the next chunk to look up
// chunks of (d_tok, B, T) that scone_embed_prefetch already prepared (0: none -- a prefetch of another batch is discarded)
long long scone_stage_take_prefetched(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, long long seqs);
void scone_stage_resync(scone_handle *h);  // a call failed mid-batch: the pipeline and its cache are dropped (the next call starts cold)
int scone_stage_note_prefetched(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, long long seqs, long long n);
hipStream_t scone_stage_side(scone_handle *h);
hipEvent_t scone_stage_start_event(scone_handle *h);
hipEvent_t scone_stage_staged_event(scone_handle *h, int buf);
int scone_stage_mark_consumed(scone_handle *h, int buf, hipStream_t main_stream);
const int32_t *scone_stage_ell(scone_handle *h, int buf);
uint8_t *scone_stage_rows(scone_handle *h);      // the cache of cold rows: slot s at s * payload bytes
const void *scone_stage_scales(scone_handle *h);  // [hot rows + cache slots, scale bytes]
long long scone_stage_chunk_tokens(scone_handle *h);
