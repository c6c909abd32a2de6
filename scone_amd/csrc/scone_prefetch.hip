// scone_embed_prefetch for tables WITHOUT a staging pipeline (rows in HBM, or read in place from pinned host DRAM).
//
// A large-batch lookup is two kernels on the caller's stream: k_match_ell (tokens -> per-token id records; latency-bound,
// ~45 us per 1M tokens against a 1M-key index) and the gather / reduce kernel that reads the records (bandwidth-bound,
// ~0.7 ms).  The match of batch i + 1 does not depend on anything batch i computes: a caller that knows its next tokens
// announces them, the match runs on a side stream of the handle BESIDE the gather of batch i, and the scone_embed of that
// batch only waits for an event that has long completed -- the step becomes the gather kernel alone.
//
// The reference has no counterpart: it matches one sequence at a time inside its Python loop
// (scone/tokenization/n_gram_extractor.py:106-126 called from scone/inference/engine.py:223-250).  Semantics are untouched:
// the records are what k_match_ell writes in the serial path, bit for bit.
//
// SCONE_PF_SLOTS record buffers: prefetch(i + 1) is issued while the lookup of batch i (which reads the other buffer) is
// still queued.  Ordering, all by events:
//   ready[k]     recorded on the side stream behind the match          -> the lookup's stream waits for it
//   consumed[k]  recorded on the lookup's stream behind the gather     -> the next match into buffer k waits for it
//   start        recorded on the caller's stream (tokens_ready == 0)   -> the match waits for the tokens' producer
#include "scone_common.h"

#include <cstdlib>
#include <new>

#define SCONE_PF_SLOTS 4  // buffers that exist; SCONE_PF_SLOTS=<n> (environment, 2..4) limits how many are used (A/B aid)

struct scone_pf_slot {
  int32_t *ell = nullptr;
  int64_t cap_tokens = 0;
  hipEvent_t ready = nullptr, consumed = nullptr;
  bool consumed_valid = false;  // a lookup has read this buffer: the next match into it waits for `consumed`
  bool valid = false;           // holds the records of (tok, B, T), not yet taken by a lookup
  const int32_t *tok = nullptr;
  int32_t B = 0, T = 0;
  uint64_t stamp = 0;           // prefetch order (the oldest pending one is dropped when every buffer is taken)
  uint64_t used = 0;            // clock value when a lookup last took this buffer
};

struct scone_pf_state {
  hipStream_t side = nullptr;
  hipEvent_t start = nullptr;
  scone_pf_slot slot[SCONE_PF_SLOTS];
  int n_slots = 3;
  uint64_t clock = 0;
};

void scone_pf_destroy(scone_handle *h) {
  scone_pf_state *st = h->pf;
  if (!st) return;
  if (st->side) {
    (void)hipStreamSynchronize(st->side);
    (void)hipStreamDestroy(st->side);
  }
  if (st->start) (void)hipEventDestroy(st->start);
  for (scone_pf_slot &sl : st->slot) {
    if (sl.ready) (void)hipEventDestroy(sl.ready);
    if (sl.consumed) (void)hipEventDestroy(sl.consumed);
    if (sl.ell) (void)hipFree(sl.ell);
  }
  delete st;
  h->pf = nullptr;
  h->pf_any.store(false, std::memory_order_release);
}

void scone_pf_invalidate(scone_handle *h) {
  if (!h->pf_any.load(std::memory_order_acquire)) return;
  std::lock_guard<std::mutex> g(h->pf_mu);
  if (!h->pf) return;
  for (scone_pf_slot &sl : h->pf->slot) sl.valid = false;
}

static int pf_state(scone_handle *h) {  // pf_mu held
  if (h->pf) return SCONE_OK;
  scone_pf_state *st = new (std::nothrow) scone_pf_state();
  if (!st) return scone_fail(h, SCONE_ENOMEM, "scone_embed_prefetch: out of memory");
  h->pf = st;
  // The match is short and latency-bound, the gather beside it fills every wave slot for ~0.7 ms in three residency
  // rounds: at a higher priority the dispatcher hands freed slots to the match first (SCONE_PF_PRIORITY=0: default
  // priority, an A/B aid).
  int least = 0, greatest = 0;  // numerically: least priority = largest value
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  int prio = least;
  if (const char *ev = getenv("SCONE_PF_PRIORITY")) {
    if (*ev == 'h') prio = greatest;
    else if (*ev == 'n' || *ev == '0') prio = 0;
  }
  if (const char *ev = getenv("SCONE_PF_SLOTS")) {
    const int v = atoi(ev);
    if (v >= 2 && v <= SCONE_PF_SLOTS) st->n_slots = v;
  }
  SCONE_HIP(h, hipStreamCreateWithPriority(&st->side, hipStreamNonBlocking, prio));
  // device-scope release: these events order streams of ONE device, nothing on the host ever waits for them
  unsigned ev_flags = hipEventDisableTiming | hipEventReleaseToDevice;
  if (const char *ev = getenv("SCONE_PF_EVENT_SYSTEM_SCOPE"))
    if (*ev == '1') ev_flags = hipEventDisableTiming;
  SCONE_HIP(h, hipEventCreateWithFlags(&st->start, ev_flags));
  for (scone_pf_slot &sl : st->slot) {
    SCONE_HIP(h, hipEventCreateWithFlags(&sl.ready, ev_flags));
    SCONE_HIP(h, hipEventCreateWithFlags(&sl.consumed, ev_flags));
  }
  h->pf_any.store(true, std::memory_order_release);
  return SCONE_OK;
}

static int pf_ensure(scone_handle *h, scone_pf_slot &sl, int64_t ntok) {  // pf_mu held
  if (ntok < h->reserve_tokens) ntok = h->reserve_tokens;
  if (ntok <= sl.cap_tokens) return SCONE_OK;
  if (sl.ell) SCONE_HIP(h, hipFree(sl.ell));  // hipFree synchronises the device: nothing still reads the old buffer
  sl.ell = nullptr, sl.cap_tokens = 0, sl.consumed_valid = false;
  SCONE_HIP(h, hipMalloc(&sl.ell, (size_t)ntok * SCONE_ELL_W(h->cfg.max_n) * sizeof(int32_t)));
  sl.cap_tokens = ntok;
  return SCONE_OK;
}

int scone_pf_reserve(scone_handle *h, int64_t max_tokens) {
  (void)max_tokens;  // (h->reserve_tokens already holds the maximum: pf_ensure allocates at least that)
  if (!h->pf_any.load(std::memory_order_acquire)) return SCONE_OK;
  std::lock_guard<std::mutex> g(h->pf_mu);
  if (!h->pf) return SCONE_OK;
  for (int k = 0; k < h->pf->n_slots; ++k) {
    scone_pf_slot &sl = h->pf->slot[k];
    if (!sl.valid) {
      int rc = pf_ensure(h, sl, 1);
      if (rc) return rc;
    }
  }
  return SCONE_OK;
}

int scone_pf_prefetch(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t tokens_ready, hipStream_t stream) {
  std::lock_guard<std::mutex> g(h->pf_mu);
  int rc = pf_state(h);
  if (rc) return rc;
  scone_pf_state *st = h->pf;
  // the buffer: one that already holds this batch (prefetched twice: matched again, the tokens may have changed), else a free
  // one, else the oldest pending prefetch is dropped
  scone_pf_slot *sl = nullptr;
  for (int k = 0; k < st->n_slots; ++k) {
    scone_pf_slot &c = st->slot[k];
    if (c.valid && c.tok == d_tok && c.B == B && c.T == T) sl = &c;
  }
  // a free buffer: the one whose last lookup lies furthest back (its `consumed` event has most likely completed, so the match
  // can start at once, beside whatever runs now -- with 3 buffers in a loop that is the lookup BEFORE the one just queued)
  if (!sl)
    for (int k = 0; k < st->n_slots; ++k) {
      scone_pf_slot &c = st->slot[k];
      if (!c.valid && (!sl || c.used < sl->used)) sl = &c;
    }
  if (!sl)
    for (int k = 0; k < st->n_slots; ++k) {
      scone_pf_slot &c = st->slot[k];
      if (!sl || c.stamp < sl->stamp) sl = &c;
    }
  sl->valid = false;
  rc = pf_ensure(h, *sl, (int64_t)B * T);
  if (rc) return rc;
  if (sl->consumed_valid) SCONE_HIP(h, hipStreamWaitEvent(st->side, sl->consumed, 0));
  if (!tokens_ready) {  // the tokens are produced on `stream`: the match goes behind what is queued there now
    SCONE_HIP(h, hipEventRecord(st->start, stream));
    SCONE_HIP(h, hipStreamWaitEvent(st->side, st->start, 0));
  }
  rc = scone_launch_match_ell(h, d_tok, B, T, sl->ell, st->side);
  if (rc) return rc;
  SCONE_HIP(h, hipEventRecord(sl->ready, st->side));
  sl->tok = d_tok, sl->B = B, sl->T = T;
  sl->stamp = ++st->clock;
  sl->valid = true;
  return SCONE_OK;
}

const int32_t *scone_pf_take(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, hipStream_t s, int *slot) {
  if (!h->pf_any.load(std::memory_order_acquire)) return nullptr;
  h->pf_mu.lock();
  scone_pf_state *st = h->pf;
  if (st) {
    for (int k = 0; k < st->n_slots; ++k) {
      scone_pf_slot &sl = st->slot[k];
      if (!(sl.valid && sl.tok == d_tok && sl.B == B && sl.T == T)) continue;
      sl.valid = false;
      sl.used = ++st->clock;
      if (hipStreamWaitEvent(s, sl.ready, 0) != hipSuccess) {
        (void)hipGetLastError();
        break;  // fall back to the serial match: correctness does not depend on the prefetch
      }
      *slot = k;
      return sl.ell;  // pf_mu stays held until scone_pf_release
    }
  }
  h->pf_mu.unlock();
  return nullptr;
}

int scone_pf_release(scone_handle *h, int slot, hipStream_t s) {
  scone_pf_slot &sl = h->pf->slot[slot];
  // recorded even when the lookup's launch failed: the next match into this buffer then waits for whatever IS queued on s
  const hipError_t e = hipEventRecord(sl.consumed, s);
  sl.consumed_valid = e == hipSuccess;
  h->pf_mu.unlock();
  return e == hipSuccess ? SCONE_OK : scone_hip_fail(h, e, "hipEventRecord");
}
