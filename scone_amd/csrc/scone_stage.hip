// Prefetch for tables whose rows live in pinned host DRAM (SCONE_PLACE_PINNED_HOST with cfg.stage_tokens > 0; BASELINE
// config C4): a PERSISTENT cache of cold rows in HBM in front of a chunk pipeline.
//
// The reference's analogue is its memory-mapped table (embedding_cache.py:76-91, 132-135): rows are faulted in from
// storage on first use and stay in the page cache.  Here the "storage" is host DRAM behind PCIe and the page cache is
// `cap` row slots of HBM (cfg.cache_rows, at least what the pipeline itself needs) that live as long as the table does --
// across chunks, batches and calls: rows are immutable (any table mutation drops the whole state, scone_table.hip
// table_modified), so a cached row is bit-identical to the host row and results do not depend on what happens to be cached.
//
// The batch is cut into chunks of whole sequences.  For chunk c, on a PREP stream:
//   1. k_match_ell     tokens -> per-token id records (global ids)
//   2. k_stage_touch   every reference to a cold row: cached -> its slot is stamped with this chunk's epoch (the clock's
//                      reference bit); not cached -> the first claimer lists the row (LDS stash, one global atomic per
//                      workgroup)
//   3. k_stage_place   every listed row takes a slot from the clock hand (64 candidates per wave and round): a slot not
//                      stamped by this chunk or the chunks whose lookups may still be running; its previous owner leaves
//   4. k_stage_remap   ids in the records -> n_hot + slot
// on a COPY stream:
//   5. k_stage_copy    listed rows: mapped host DRAM -> their slots (+ scales)
// and on the caller's stream, after the chunk's "staged" event: the ordinary lookup kernel with the cache as the cold half
// of its row store.  A ring of SCONE_STAGE_NBUF sets of id records: while chunk c is reduced and chunk c+1 crosses the link,
// chunk c+2 is matched and placed; scone_embed_prefetch prepares the first two chunks of the NEXT batch behind the last ones of
// this one, so that a loop that knows its next tokens early never pays the pipeline fill (the first chunk's preparation and
// copy overlap nothing otherwise: ~0.2 ms of a 1.15-ms step).  A row referenced by several tokens, chunks or batches crosses
// PCIe once for as long as it stays cached.
//
// Round 4, what the kernel trace of the round-3 form showed (profiles/r04a): (i) the claim pass appended every claimed row
// with its own atomicAdd on ONE counter -- ~90 retire per microsecond on one address: 200 us per 262k-token chunk, now a
// stash per workgroup; (ii) the copy kernel ran 4096 waves, each with a 512-B read over PCIe in flight: ten times the
// link's bandwidth-delay product, and those reads sit in the L2's miss queues for microseconds -- every kernel running
// beside it (the next chunk's match: 20 -> 250 us; the lookup: 168 -> 360-630 us) queued behind them.  The copy grid is
// now sized to the link (SCONE_STAGE_COPY_BLOCKS).
#include "scone_common.h"

#include <cstdio>
#include <cstdlib>
#include <new>

namespace {

#define STAGE_PENDING 0xFFFFFFFFu
#define STAGE_FAILED 0xFFFFFFFFu
// a slot stamped by one of the last STAGE_PROTECT chunks is not evicted: the lookups of chunks e - NBUF + 1 .. e - 1 may
// still be queued (their records already hold slot numbers) when chunk e's rows are copied in
#define STAGE_PROTECT (SCONE_STAGE_NBUF + 1)
#define TOUCH_THREADS 1024
#define TOUCH_BLOCKS 512
#define TOUCH_STASH 4096

// slot_of[cold row]: 0 = not cached, STAGE_PENDING = listed by the chunk being prepared, else slot + 1
__global__ __launch_bounds__(TOUCH_THREADS) void k_stage_touch(const int32_t *__restrict__ ell, long long ntok, int W, int NC,
                                                               long long n_hot, uint32_t *__restrict__ slot_of,
                                                               uint32_t *__restrict__ last_use, uint32_t epoch,
                                                               uint32_t *__restrict__ count, int32_t *__restrict__ list,
                                                               uint32_t list_cap, uint32_t *__restrict__ status) {
  __shared__ int32_t stash[TOUCH_STASH];
  __shared__ uint32_t n_stash, base;
  if (threadIdx.x == 0) n_stash = 0;
  __syncthreads();
  const long long per = (ntok + gridDim.x - 1) / gridDim.x;
  const long long t0 = (long long)blockIdx.x * per, t1 = t0 + per < ntok ? t0 + per : ntok;
  for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    const int kown = ell[t * W + W - 2] & 0xFF;
    for (int j = 0; j < NC; ++j) {
      if (j >= kown) break;
      const long long id = ell[t * W + j];
      if (id < n_hot) continue;
      uint32_t *e = &slot_of[id - n_hot];
      uint32_t v = *e;
      if (v == 0u) {
        v = atomicCAS(e, 0u, STAGE_PENDING);
        if (v == 0u) {  // first reference of this chunk to a row that is not cached: list it
          const uint32_t k = atomicAdd(&n_stash, 1u);
          if (k < TOUCH_STASH) {
            stash[k] = (int32_t)id;
          } else {
            const uint32_t s = atomicAdd(count, 1u);
            if (s < list_cap) list[s] = (int32_t)id;
            else atomicOr(status, SCONE_ST_STAGE_OVERFLOW);  // (the list holds every reference of a chunk: unreachable)
          }
          continue;
        }
      }
      if (v != STAGE_PENDING) last_use[v - 1u] = epoch;  // cached: the clock's reference stamp (racing stores write one value)
    }
  }
  __syncthreads();
  const uint32_t n = n_stash < TOUCH_STASH ? n_stash : TOUCH_STASH;
  if (threadIdx.x == 0) base = n ? atomicAdd(count, n) : 0u;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
    if (base + k < list_cap) list[base + k] = stash[k];
    else atomicOr(status, SCONE_ST_STAGE_OVERFLOW);
  }
}

// one thread per listed row; placement is a WAVE's job.  The clock hand is a counter: every round a wave with rows still to
// place takes 64 consecutive positions (ONE atomic), every lane tests one candidate slot -- all 64 lanes, also those that have
// nothing (left) to place, so that the last rows of a wave find their slots in a round or two instead of one candidate per
// round each --, a ballot says which candidates can be taken, and the k-th lane still in need takes the k-th of them.  A slot
// is handed to one lane only (the hand passes it once per `cap` positions, and a launch advances it by a fraction of that), so
// owner / slot_of need no atomics.  A candidate stamped by one of the last STAGE_PROTECT chunks is passed over (a lookup in
// flight may read it); everything else the hand reaches is evicted: first in, first out, except for what is in use right now.
// (Built, measured and removed: a second chance -- a slot referenced since it was filled is spared once, CLOCK proper.  It took
// 12-18 % of the rows off PCIe (16M slots: 44.6k -> 38.1k per step) and made the step 2-4 % SLOWER at 262k-token chunks, 1.5 %
// faster at 131k: the extra rounds of this kernel sit on the preparation chain, the saved rows on a link that was not the
// bound.  profiles/r04n.)
__global__ __launch_bounds__(256) void k_stage_place(const uint32_t *__restrict__ count, const int32_t *__restrict__ list,
                                                     uint32_t *__restrict__ place, uint32_t list_cap, long long n_hot,
                                                     uint32_t *__restrict__ slot_of, uint32_t *__restrict__ owner,
                                                     uint32_t *__restrict__ last_use, uint32_t epoch, uint32_t cap,
                                                     unsigned long long *__restrict__ hand, uint32_t *__restrict__ status) {
  uint32_t n = *count;
  if (n > list_cap) n = list_cap;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool need = i < n;
  const long long cold = need ? (long long)list[i] - n_hot : 0;
  for (int round = 0; round < 64; ++round) {
    const unsigned long long needm = __ballot(need);
    if (!needm) break;
    const uint32_t width = cap < 64u ? cap : 64u;  // (a cache of fewer than 64 slots: no slot twice in one round)
    unsigned long long b = 0;
    if (lane == 0) b = atomicAdd(hand, (unsigned long long)width);
    b = __shfl(b, 0, 64);
    const uint32_t s = (uint32_t)((b + (unsigned long long)lane) % cap);
    const uint32_t lu = last_use[s];
    const bool takeable = (uint32_t)lane < width && epoch - lu >= STAGE_PROTECT;
    const unsigned long long freem = __ballot(takeable);
    const int my_rank = __popcll(needm & ((1ull << lane) - 1ull));
    int src_lane = 0;
    const bool got = need && my_rank < __popcll(freem);
    if (got) {
      unsigned long long m = freem;
      for (int k = 0; k < my_rank; ++k) m &= m - 1ull;
      src_lane = __builtin_ctzll(m);
    }
    const uint32_t sc = (uint32_t)__shfl((int)s, src_lane, 64);  // (every lane takes part in the shuffles)
    const uint32_t luc = (uint32_t)__shfl((int)lu, src_lane, 64);
    // the stamp is the claim: should the hand lap itself inside one launch (a cache barely larger than the chunk's rows), two
    // waves may look at one slot -- only the one whose compare-and-swap moves the stamp from what it saw takes it
    if (got && atomicCAS(&last_use[sc], luc, epoch) == luc) {
      const uint32_t old = owner[sc];
      if (old) slot_of[old - 1u] = 0u;  // the previous owner leaves the cache
      owner[sc] = (uint32_t)cold + 1u;
      slot_of[cold] = sc + 1u;
      place[i] = sc;
      need = false;
    }
  }
  if (need) {  // no evictable slot in 64 rounds: the cache is smaller than what the chunks in flight reference.  Cannot
    slot_of[cold] = 0u;  // happen with the sizes scone_stage_prepare picks (test hook: SCONE_STAGE_CAP_ROWS).
    place[i] = STAGE_FAILED;
    atomicOr(status, SCONE_ST_STAGE_OVERFLOW);
  }
}

__global__ __launch_bounds__(256) void k_stage_remap(int32_t *__restrict__ ell, long long ntok, int W, int NC,
                                                     long long n_hot, const uint32_t *__restrict__ slot_of,
                                                     uint32_t *__restrict__ status) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long t = gid / NC;
  const int j = (int)(gid - t * NC);
  if (t >= ntok) return;
  const int kown = ell[t * W + W - 2] & 0xFF;
  if (j >= kown) return;
  const long long id = ell[t * W + j];
  if (id < n_hot) return;
  uint32_t v = slot_of[id - n_hot];
  if (v == 0u || v == STAGE_PENDING) {  // the row could not be placed (flagged by k_stage_place): stay inside the cache
    atomicOr(status, SCONE_ST_STAGE_OVERFLOW);
    v = 1u;
  }
  ell[t * W + j] = (int32_t)(n_hot + (long long)(v - 1u));
}

// one wave per listed row; 16 bytes per lane per step from mapped host memory.  The grid is what bounds the reads in flight
// over PCIe (one row per wave): 128 workgroups = 512 waves x 512 B = 0.25 MB, twice the link's bandwidth-delay product
// (~50 GB/s x ~2.5 us); the round-3 grid of 1024 workgroups kept 2 MB of PCIe reads in the L2's miss queues and slowed
// every kernel beside it 2-10x.  Measured on the C4 Zipf stream, steady state, 8M-row cache (profiles/r04g): 96 / 128
// workgroups 1.235 / 1.154 ms per 1M-token step at 262k-token chunks, 1.228 / 1.172 at 131k; the cache-filling phase
// (every row a miss): 64 / 128 / 256 / 512 / 1024 workgroups 1.74 / 1.33 / 1.42 / 1.49 / 1.58 ms (profiles/r04b).
// With scone_embed_prefetch, 16M slots: 128 / 256 / 512 workgroups 0.905 / 0.995 / 1.126 ms.  Confining the 128 workgroups to
// 4 / 2 / 1 of the 8 XCDs (only the workgroups the dispatcher deals to those XCDs work) was tried and is slower: 0.945 /
// 0.996 / 1.031 ms against 0.905 (profiles/r04x) -- the lookup's tiles are dealt to XCDs statically, and the XCDs that carry
// the link's reads become its stragglers.
__global__ __launch_bounds__(256) void k_stage_copy(const uint32_t *__restrict__ count, const int32_t *__restrict__ list,
                                                    const uint32_t *__restrict__ place, scone_row_store host,
                                                    uint8_t *__restrict__ cache_rows, const uint8_t *__restrict__ scales,
                                                    uint8_t *__restrict__ cache_scales, int scale_bytes, long long n_hot,
                                                    uint32_t list_cap, unsigned long long *__restrict__ stats) {
  const int lane = threadIdx.x & 63;
  uint32_t n = *count;
  if (n > list_cap) n = list_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[0], (unsigned long long)n);
  const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
  for (unsigned i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += nwaves) {
    const uint32_t s = place[i];
    if (s == STAGE_FAILED) continue;
    const unsigned long long lr = (unsigned long long)list[i];
    const uint4 *src = reinterpret_cast<const uint4 *>(host.row(lr));
    uint4 *dst = reinterpret_cast<uint4 *>(cache_rows + (size_t)s * host.row_bytes);
    for (unsigned v = lane; v < host.row_bytes / 16; v += 64) dst[v] = src[v];
    if (scale_bytes && lane < scale_bytes / 2)
      reinterpret_cast<unsigned short *>(cache_scales + (size_t)(n_hot + s) * scale_bytes)[lane] =
          reinterpret_cast<const unsigned short *>(scales + lr * scale_bytes)[lane];
  }
}

// Overlap probe (scone_stage_bind): does a kernel queued on stream B start while a kernel queued before it on stream A is still
// running?  k_probe_spin holds A for `ticks` of the 100-MHz wall clock (bounded: it ends after 2^22 polls whatever the clock
// says) and leaves its END time; k_probe_stamp leaves its START time.
__global__ void k_probe_spin(long long ticks, long long *out) {
  const long long t0 = wall_clock64();
  long long t = t0;
  for (int guard = 0; t - t0 < ticks && guard < (1 << 22); ++guard) t = wall_clock64();
  out[0] = t;
}
__global__ void k_probe_stamp(long long *out) { out[1] = wall_clock64(); }

}  // namespace

#define SCONE_STAGE_CAND 6  // candidate side streams per pipeline (the runtime spreads streams over 4 hardware queues)

struct scone_stage_state {
  long long chunk_tokens = 0;      // tokens per chunk actually provisioned
  long long requested_tokens = 0;  // what the caller asked for (may exceed chunk_tokens, see prepare)
  uint32_t cap = 0;       // cache slots
  uint32_t list_cap = 0;  // rows one chunk can list
  int copy_blocks = 128;
  hipStream_t prep = nullptr, copy = nullptr;  // two of cand[], chosen by scone_stage_bind
  hipStream_t cand[SCONE_STAGE_CAND] = {};
  hipStream_t bound_to = nullptr;  // the caller's stream the choice was made against
  bool bound = false;
  int other_caller_calls = 0;      // consecutive calls on a different caller stream (re-bind after a few)
  long long *probe_ts = nullptr;   // device [2]
  hipEvent_t prepped[SCONE_STAGE_NBUF] = {}, staged[SCONE_STAGE_NBUF] = {}, consumed[SCONE_STAGE_NBUF] = {}, start = nullptr;
  bool consumed_valid[SCONE_STAGE_NBUF] = {};
  bool staged_valid[SCONE_STAGE_NBUF] = {};  // the set has carried a chunk: its copy kernel read count / list / place
  uint32_t *slot_of = nullptr;   // [cold rows]
  uint32_t *owner = nullptr;     // [cap]: cold row + 1 held by the slot, 0 = free
  uint32_t *last_use = nullptr;  // [cap]: epoch of the last chunk that referenced the slot
  uint32_t epoch = STAGE_PROTECT;
  uint64_t chunks = 0;                  // chunks prepared since the state was created (= the ring position of the next one)
  uint64_t n_consumed = 0;              // chunks handed to a lookup (or discarded) so far: chunks - consumed <= SCONE_STAGE_NBUF
  struct {                              // what scone_embed_prefetch prepared ahead of the scone_embed that will use it
    const int32_t *tok = nullptr;
    int32_t B = 0, T = 0;
    long long seqs = 0, n = 0;
    bool valid = false;
  } pending;
  unsigned long long *hand = nullptr;   // [0] clock hand, [1] rows copied host -> HBM
  uint8_t *rows = nullptr;       // [cap, payload bytes]
  uint8_t *scales = nullptr;     // [hot rows + cap, scale bytes]: the HBM-resident head's scales, then the cached rows'
  int32_t *ell[SCONE_STAGE_NBUF] = {};
  int32_t *list[SCONE_STAGE_NBUF] = {};
  uint32_t *place[SCONE_STAGE_NBUF] = {};
  uint32_t *count[SCONE_STAGE_NBUF] = {};
};

void scone_stage_destroy(scone_handle *h) {
  scone_stage_state *st = h->stage;
  if (!st) return;
  for (hipStream_t c : st->cand)
    if (c) (void)hipStreamDestroy(c);
  if (st->probe_ts) (void)hipFree(st->probe_ts);
  for (int b = 0; b < SCONE_STAGE_NBUF; ++b) {
    if (st->prepped[b]) (void)hipEventDestroy(st->prepped[b]);
    if (st->staged[b]) (void)hipEventDestroy(st->staged[b]);
    if (st->consumed[b]) (void)hipEventDestroy(st->consumed[b]);
    if (st->ell[b]) (void)hipFree(st->ell[b]);
    if (st->list[b]) (void)hipFree(st->list[b]);
    if (st->place[b]) (void)hipFree(st->place[b]);
    if (st->count[b]) (void)hipFree(st->count[b]);
  }
  if (st->start) (void)hipEventDestroy(st->start);
  void *ptrs[] = {st->slot_of, st->owner, st->last_use, st->hand, st->rows, st->scales};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  delete st;
  h->stage = nullptr;
}

int scone_stage_prepare(scone_handle *h, long long chunk_tokens) {
  if (h->stage && h->stage->requested_tokens >= chunk_tokens) return SCONE_OK;
  const long long requested = chunk_tokens;
  scone_stage_destroy(h);
  scone_stage_state *st = new (std::nothrow) scone_stage_state();
  if (!st) return scone_fail(h, SCONE_ENOMEM, "scone_embed(staged): out of memory");
  h->stage = st;
  const int W = SCONE_ELL_W(h->cfg.max_n), NC = h->cfg.max_n * (h->cfg.max_n + 1) / 2;
  const long long n_cold = (long long)(h->local_rows - h->hot_local);
  // The cache must hold every row the chunks in flight reference (a reference that finds no slot would be remapped to a
  // wrong row): a chunk references at most chunk_tokens * NC distinct cold rows and STAGE_PROTECT chunks are protected
  // from eviction while one more is placed -- unless the whole cold table fits, then nothing is ever evicted.
  const long long id_room = 0x7FFFFFF0ll - (long long)h->hot_local;  // record ids are int32: n_hot + slot
  long long cap = (long long)h->cfg.cache_rows;
  const long long need = (STAGE_PROTECT + 1) * chunk_tokens * NC;
  if (cap < need) cap = need;
  if (cap > n_cold) cap = n_cold;
  if (cap > id_room) cap = id_room;
  if (const char *ev = getenv("SCONE_STAGE_CAP_ROWS")) {  // test hook: an undersized cache exercises the overflow path
    const long long forced = atoll(ev);
    if (forced > 0 && forced < cap) cap = forced;
  } else if (cap < n_cold && cap < need) {
    chunk_tokens = cap / ((STAGE_PROTECT + 1) * NC);  // a small cache (or 2^31 ids): smaller chunks
  }
  if (cap < 1) cap = 1;
  long long list_cap = chunk_tokens * NC < n_cold ? chunk_tokens * NC : n_cold;
  if (list_cap < 1) list_cap = 1;
  st->chunk_tokens = chunk_tokens;
  st->requested_tokens = requested;
  st->cap = (uint32_t)cap;
  st->list_cap = (uint32_t)list_cap;
  if (const char *ev = getenv("SCONE_STAGE_COPY_BLOCKS")) {
    const int v = atoi(ev);
    if (v >= 1 && v <= 65535) st->copy_blocks = v;
  }
  // Candidate side streams; which two become PREP and COPY is decided against the caller's stream at the first lookup
  // (scone_stage_bind).  Until then the first two.
  for (int i = 0; i < SCONE_STAGE_CAND; ++i) SCONE_HIP(h, hipStreamCreateWithFlags(&st->cand[i], hipStreamNonBlocking));
  st->prep = st->cand[0], st->copy = st->cand[1];
  SCONE_HIP(h, hipMalloc(&st->probe_ts, 2 * sizeof(long long)));
  SCONE_HIP(h, hipEventCreateWithFlags(&st->start, hipEventDisableTiming));
  const size_t nc = (size_t)(n_cold > 0 ? n_cold : 1);
  SCONE_HIP(h, hipMalloc(&st->slot_of, nc * 4));
  SCONE_HIP(h, hipMemset(st->slot_of, 0, nc * 4));
  SCONE_HIP(h, hipMalloc(&st->owner, (size_t)cap * 4));
  SCONE_HIP(h, hipMemset(st->owner, 0, (size_t)cap * 4));
  SCONE_HIP(h, hipMalloc(&st->last_use, (size_t)cap * 4));
  SCONE_HIP(h, hipMemset(st->last_use, 0, (size_t)cap * 4));
  SCONE_HIP(h, hipMalloc(&st->hand, 4 * sizeof(unsigned long long)));
  SCONE_HIP(h, hipMemset(st->hand, 0, 4 * sizeof(unsigned long long)));
  SCONE_HIP(h, hipMalloc(&st->rows, (size_t)cap * h->row_payload_bytes));
  const size_t sb = h->scale_bytes_per_row;
  if (sb) {  // scales indexed like the rows: [0, n_hot) = the HBM-resident head, then the cache slots
    SCONE_HIP(h, hipMalloc(&st->scales, (size_t)(h->hot_local + cap) * sb + 4));
    SCONE_HIP(h, hipMemcpy(st->scales, h->scales, (size_t)h->hot_local * sb, hipMemcpyDeviceToDevice));
  }
  for (int b = 0; b < SCONE_STAGE_NBUF; ++b) {
    SCONE_HIP(h, hipEventCreateWithFlags(&st->prepped[b], hipEventDisableTiming));
    SCONE_HIP(h, hipEventCreateWithFlags(&st->staged[b], hipEventDisableTiming));
    SCONE_HIP(h, hipEventCreateWithFlags(&st->consumed[b], hipEventDisableTiming));
    SCONE_HIP(h, hipMalloc(&st->ell[b], (size_t)(chunk_tokens > 0 ? chunk_tokens : 1) * W * 4));
    SCONE_HIP(h, hipMalloc(&st->list[b], (size_t)list_cap * 4));
    SCONE_HIP(h, hipMalloc(&st->place[b], (size_t)list_cap * 4));
    SCONE_HIP(h, hipMalloc(&st->count[b], 4));
  }
  return SCONE_OK;
}

// Which side streams?  The pipeline lives on overlap: the COPY kernel of chunk c + 1 (PCIe-bound, ~0.3 ms) must run BESIDE the
// lookup of chunk c on the caller's stream, and the preparation of chunk c + 2 beside both.  HIP multiplexes a process's streams
// onto a few hardware queues (4 by default), and two streams that share a queue run one after the other: with the side streams
// simply created at this point the cached C4 step took 0.92 ms or 1.5 ms depending on HOW MANY OTHER STREAMS the process had used
// before (round 6, profiles/r06i: 0.92 / 0.92 / 1.52 / 0.92 / 0.92 / 1.46 / 1.20 / 0.92 ms for 0 .. 7 torch side streams) --
// any application with a few streams of its own (RCCL, a data loader) plays that lottery.  So the pipeline measures instead of
// hoping: SCONE_STAGE_CAND candidate streams; a candidate is usable as COPY if a kernel queued on it starts while a spinning
// kernel queued before it on the CALLER's stream is still running; PREP likewise, and it must also overlap with the chosen COPY
// stream.  ~0.1 ms per test, at most 2 * SCONE_STAGE_CAND tests, once per pipeline and caller stream (synchronises the caller's
// stream: a hidden sync at the FIRST staged lookup only, never afterwards).  SCONE_STAGE_TRACE=1 prints the outcome.
static int streams_overlap(scone_handle *h, long long *d_ts, hipStream_t a, hipStream_t b, bool *yes) {
  long long ts[2] = {0, 0};
  hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(1), 0, a, (long long)8000, d_ts);  // 80 us on the 100-MHz clock
  hipLaunchKernelGGL(k_probe_stamp, dim3(1), dim3(1), 0, b, d_ts);
  SCONE_HIP(h, hipGetLastError());
  SCONE_HIP(h, hipStreamSynchronize(a));
  SCONE_HIP(h, hipStreamSynchronize(b));
  SCONE_HIP(h, hipMemcpy(ts, d_ts, sizeof(ts), hipMemcpyDeviceToHost));
  *yes = ts[1] < ts[0];
  return SCONE_OK;
}
static int stage_overlaps(scone_handle *h, scone_stage_state *st, hipStream_t a, hipStream_t b, bool *yes) {
  return streams_overlap(h, st->probe_ts, a, b, yes);
}

extern "C" int scone_streams_overlap(scone_handle *h, scone_stream_t stream_a, scone_stream_t stream_b, int32_t *overlap) {
  if (!h) return SCONE_EINVAL;
  if (!overlap) return scone_fail(h, SCONE_EINVAL, "scone_streams_overlap: null pointer");
  *overlap = 0;
  if (stream_a == stream_b) return SCONE_OK;  // one stream: in order by definition
  SCONE_ON_DEVICE(h);
  long long *d_ts = nullptr;
  SCONE_HIP(h, hipMalloc(&d_ts, 2 * sizeof(long long)));
  bool yes = false;
  const int rc = streams_overlap(h, d_ts, (hipStream_t)stream_a, (hipStream_t)stream_b, &yes);
  (void)hipFree(d_ts);
  if (rc) return rc;
  *overlap = yes ? 1 : 0;
  return SCONE_OK;
}

int scone_stage_bind(scone_handle *h, hipStream_t caller) {
  scone_stage_state *st = h->stage;
  if (st->bound && st->bound_to == caller) {
    st->other_caller_calls = 0;
    return SCONE_OK;
  }
  if (st->bound && ++st->other_caller_calls < 4) return SCONE_OK;  // an occasional call from another stream keeps the choice
  // (re-)bind: nothing of the pipeline may be in flight while PREP / COPY change hands (consecutive chunks rely on the order
  // of their preparation kernels on ONE stream)
  SCONE_HIP(h, hipStreamSynchronize(st->prep));
  SCONE_HIP(h, hipStreamSynchronize(st->copy));
  bool ok_main[SCONE_STAGE_CAND] = {};
  int n_ok = 0;
  for (int i = 0; i < SCONE_STAGE_CAND; ++i) {
    int rc = stage_overlaps(h, st, caller, st->cand[i], &ok_main[i]);
    if (rc) return rc;
    n_ok += ok_main[i];
  }
  int copy = -1, prep = -1;
  for (int i = 0; i < SCONE_STAGE_CAND && copy < 0; ++i)
    if (ok_main[i]) copy = i;
  if (copy < 0) copy = 1;  // nothing overlaps with the caller's stream (one hardware queue?): as before
  for (int pass = 0; pass < 2 && prep < 0; ++pass)  // first a stream that also overlaps with the caller's, then any
    for (int j = 0; j < SCONE_STAGE_CAND && prep < 0; ++j) {
      if (j == copy || (pass == 0 && !ok_main[j])) continue;
      bool yes = false;
      int rc = stage_overlaps(h, st, st->cand[copy], st->cand[j], &yes);
      if (rc) return rc;
      if (yes) prep = j;
    }
  if (prep < 0) prep = copy == 0 ? 1 : 0;
  st->copy = st->cand[copy], st->prep = st->cand[prep];
  st->bound = true, st->bound_to = caller, st->other_caller_calls = 0;
  static_assert(SCONE_STAGE_CAND == 6, "the trace line below prints six flags");
  if (const char *ev = getenv("SCONE_STAGE_TRACE"))
    if (*ev && *ev != '0')
      fprintf(stderr, "scone_stage_bind: %d of %d candidate streams overlap with the caller's stream [%d%d%d%d%d%d]; COPY = #%d, PREP = #%d\n",
              n_ok, SCONE_STAGE_CAND, ok_main[0], ok_main[1], ok_main[2], ok_main[3], ok_main[4], ok_main[5], copy, prep);
  return SCONE_OK;
}

// side-stream half of one chunk; leaves the records of buffer `buf` ready (ids -> cache slots), the chunk's missing rows on
// their way into the cache, and records staged[buf]
int scone_stage_chunk(scone_handle *h, const int32_t *d_tok, int32_t Bc, int32_t T) {
  scone_stage_state *st = h->stage;
  const int W = SCONE_ELL_W(h->cfg.max_n), NC = h->cfg.max_n * (h->cfg.max_n + 1) / 2;
  const long long ntok = (long long)Bc * T;
  hipStream_t s = st->prep;
  if (st->chunks - st->n_consumed >= SCONE_STAGE_NBUF)
    return scone_fail(h, SCONE_ESTATE, "scone_embed(staged): every record set of the chunk pipeline is in use (a batch prefetched "
                                       "twice without being embedded?)");
  const int buf = (int)(st->chunks % SCONE_STAGE_NBUF);
  if (st->consumed_valid[buf]) SCONE_HIP(h, hipStreamWaitEvent(s, st->consumed[buf], 0));
  // ... and for the COPY of the chunk that used this set last: a chunk that was prepared and then discarded (a prefetch of
  // another batch, a failed call) has no lookup, hence no newer `consumed`, but its k_stage_copy may still be reading
  // count / list / place[buf] when this chunk's preparation rewrites them (for a chunk that WAS looked up this wait is
  // free: its lookup waited for the same event)
  if (st->staged_valid[buf]) SCONE_HIP(h, hipStreamWaitEvent(s, st->staged[buf], 0));
  st->epoch += 1;
  if (st->epoch >= 0xFFFFFF00u) {  // 2^32 chunks: restart the clock (every slot becomes evictable; nothing is in flight
    SCONE_HIP(h, hipDeviceSynchronize());  // once the device is idle)
    SCONE_HIP(h, hipMemsetAsync(st->last_use, 0, (size_t)st->cap * 4, s));
    st->epoch = STAGE_PROTECT + 1;
  }
  int rc = scone_launch_match_ell(h, d_tok, Bc, T, st->ell[buf], s);
  if (rc) return rc;
  SCONE_HIP(h, hipMemsetAsync(st->count[buf], 0, 4, s));
  const long long work = ntok * NC;
  if (!scone_grid_fits((unsigned long long)(work + 255) / 256, 256))
    return scone_fail(h, SCONE_EINVAL, "scone_embed(staged): chunk too large for one launch (lower stage_tokens)");
  const long long want = (ntok + TOUCH_THREADS - 1) / TOUCH_THREADS;
  const unsigned tb = (unsigned)(want < TOUCH_BLOCKS ? (want > 0 ? want : 1) : TOUCH_BLOCKS);
  hipLaunchKernelGGL(k_stage_touch, dim3(tb), dim3(TOUCH_THREADS), 0, s, st->ell[buf], ntok, W, NC, (long long)h->hot_local,
                     st->slot_of, st->last_use, st->epoch, st->count[buf], st->list[buf], st->list_cap, h->d_status);
  // the list's length stays on the device: the placement grid covers the chunk's worst case, threads past the count retire
  long long worst = ntok * NC < (long long)st->list_cap ? ntok * NC : (long long)st->list_cap;
  if (worst < 1) worst = 1;
  hipLaunchKernelGGL(k_stage_place, dim3((unsigned)((worst + 255) / 256)), dim3(256), 0, s, st->count[buf], st->list[buf],
                     st->place[buf], st->list_cap, (long long)h->hot_local, st->slot_of, st->owner, st->last_use, st->epoch,
                     st->cap, st->hand, h->d_status);
  hipLaunchKernelGGL(k_stage_remap, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, st->ell[buf], ntok, W, NC,
                     (long long)h->hot_local, st->slot_of, h->d_status);
  SCONE_HIP(h, hipGetLastError());
  SCONE_HIP(h, hipEventRecord(st->prepped[buf], s));
  // the copy stream only ever waits for the list of THIS chunk, never for the preparation of the next one
  SCONE_HIP(h, hipStreamWaitEvent(st->copy, st->prepped[buf], 0));
  hipLaunchKernelGGL(k_stage_copy, dim3((unsigned)st->copy_blocks), dim3(256), 0, st->copy, st->count[buf], st->list[buf],
                     st->place[buf], scone_store_of(h), st->rows, (const uint8_t *)h->scales, st->scales,
                     (int)h->scale_bytes_per_row, (long long)h->hot_local, st->list_cap, st->hand + 1);
  SCONE_HIP(h, hipGetLastError());
  SCONE_HIP(h, hipEventRecord(st->staged[buf], st->copy));
  st->staged_valid[buf] = true;
  st->chunks += 1;
  return SCONE_OK;
}

int scone_stage_consume_buf(scone_handle *h) {
  scone_stage_state *st = h->stage;
  const int buf = (int)(st->n_consumed % SCONE_STAGE_NBUF);
  st->n_consumed += 1;
  return buf;
}

long long scone_stage_take_prefetched(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, long long seqs) {
  scone_stage_state *st = h->stage;
  if (!st->pending.valid) return 0;
  st->pending.valid = false;
  if (st->pending.tok == d_tok && st->pending.B == B && st->pending.T == T && st->pending.seqs == seqs) return st->pending.n;
  st->n_consumed += (uint64_t)st->pending.n;  // another batch was prefetched: its chunks are never looked up (the rows they
  return 0;                                 // brought into the cache stay, harmlessly)
}

// A call failed mid-batch (a launch error, an event that could not be recorded): some chunk may have been PLACED -- `owner` /
// `slot_of` already name cache slots for its rows -- without its copy having been queued, and the ring counters no longer say
// which.  Nothing of that state can be trusted, so the whole pipeline goes: the device drains, the cache is freed, and the next
// call builds a cold one (scone_stage_prepare).  Round 6 (the round-5 advisor's finding: resetting the ring counters alone left
// slots that claimed rows never copied, readable by a later lookup without any status bit).
void scone_stage_resync(scone_handle *h) {
  if (!h->stage) return;
  (void)hipDeviceSynchronize();
  scone_stage_destroy(h);
}

int scone_stage_note_prefetched(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, long long seqs, long long n) {
  scone_stage_state *st = h->stage;
  st->pending.tok = d_tok, st->pending.B = B, st->pending.T = T, st->pending.seqs = seqs, st->pending.n = n;
  st->pending.valid = n > 0;
  return SCONE_OK;
}

hipStream_t scone_stage_side(scone_handle *h) { return h->stage->prep; }
hipEvent_t scone_stage_start_event(scone_handle *h) { return h->stage->start; }
hipEvent_t scone_stage_staged_event(scone_handle *h, int buf) { return h->stage->staged[buf]; }
int scone_stage_mark_consumed(scone_handle *h, int buf, hipStream_t main_stream) {
  SCONE_HIP(h, hipEventRecord(h->stage->consumed[buf], main_stream));
  h->stage->consumed_valid[buf] = true;
  return SCONE_OK;
}
const int32_t *scone_stage_ell(scone_handle *h, int buf) { return h->stage->ell[buf]; }
uint8_t *scone_stage_rows(scone_handle *h) { return h->stage->rows; }
const void *scone_stage_scales(scone_handle *h) { return h->stage->scales; }
long long scone_stage_chunk_tokens(scone_handle *h) { return h->stage->chunk_tokens; }

extern "C" int scone_stage_counters(scone_handle *h, uint64_t *cache_rows, uint64_t *rows_copied, uint64_t *chunks,
                                    uint64_t *chunk_tokens) {
  if (!h) return SCONE_EINVAL;
  if (cache_rows) *cache_rows = 0;
  if (rows_copied) *rows_copied = 0;
  if (chunks) *chunks = 0;
  if (chunk_tokens) *chunk_tokens = 0;
  std::lock_guard<std::mutex> g(h->stage_mu);
  scone_stage_state *st = h->stage;
  if (!st) return SCONE_OK;
  SCONE_ON_DEVICE(h);
  unsigned long long v[2] = {0, 0};
  SCONE_HIP(h, hipDeviceSynchronize());
  SCONE_HIP(h, hipMemcpy(v, st->hand, sizeof(v), hipMemcpyDeviceToHost));
  if (cache_rows) *cache_rows = st->cap;
  if (rows_copied) *rows_copied = v[1];
  if (chunks) *chunks = st->chunks;
  if (chunk_tokens) *chunk_tokens = (uint64_t)st->chunk_tokens;
  return SCONE_OK;
}
