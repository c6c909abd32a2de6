// Staged prefetch for tables whose rows live in pinned host DRAM (SCONE_PLACE_PINNED_HOST with
// cfg.stage_tokens > 0; BASELINE config C4).
//
// The batch is cut into chunks of whole sequences.  For chunk c, on a SIDE stream:
//   1. k_match_ell            tokens -> per-token id records (global ids)
//   2. k_stage_claim          every distinct cold row referenced by the chunk claims one slot of the
//                             HBM staging buffer (generation-tagged slot map, one CAS per reference,
//                             nobody waits) -> a de-duplicated copy list
//   3. k_stage_remap          ids in the records -> n_hot + slot
//   4. k_stage_copy           one wave per listed row: mapped host DRAM -> HBM staging (+ its scales)
// and on the MAIN stream, after the "staged" event: the ordinary fused lookup kernel with the staging
// buffer as the cold half of the row store.  A row referenced by several tokens of a chunk (an n-gram
// covers n positions; hot f-grams recur) crosses PCIe once per chunk instead of once per reference.
// Steps 1-3 run on a PREP stream and step 4 on a COPY stream, with SCONE_STAGE_NBUF = 3 buffer sets:
// while chunk c is reduced and chunk c+1 crosses the link, chunk c+2 is already being matched and
// claimed, so the link never waits for a match -> claim -> remap chain (with one side stream and two
// buffers it idled for that chain once per chunk: ~40 GB/s instead of ~50 GB/s over Gen5 x16).
#include "scone_common.h"

#include <cstdlib>
#include <new>

namespace {

#define STAGE_PENDING 0xFFFFFFu

__global__ __launch_bounds__(256) void k_stage_claim(const int32_t *__restrict__ ell, long long ntok, int W, int NC,
                                                     long long n_hot, uint32_t *__restrict__ slot_of, uint32_t gen,
                                                     uint32_t *__restrict__ count, int32_t *__restrict__ list,
                                                     uint32_t cap, uint32_t *__restrict__ status) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long t = gid / NC;
  const int j = (int)(gid - t * NC);
  if (t >= ntok) return;
  const int kown = ell[t * W + W - 2] & 0xFF;
  if (j >= kown) return;
  const long long id = ell[t * W + j];
  if (id < n_hot) return;
  uint32_t *e = &slot_of[id - n_hot];
  const uint32_t v = *e;
  if ((v >> 24) == gen) return;  // already claimed for this chunk
  if (atomicCAS(e, v, (gen << 24) | STAGE_PENDING) == v) {
    const uint32_t s = atomicAdd(count, 1u);
    if (s < cap) {
      list[s] = (int32_t)id;
      *e = (gen << 24) | s;  // read only by the next kernel
    } else {
      // more distinct cold rows than staging slots.  scone_stage_prepare sizes the buffer for the worst case of a
      // chunk, so this is unreachable from scone_embed; should it ever happen the reference must not stay PENDING
      // (the remap would send the lookup 16M rows past the buffer): it reads slot 0 and the call is flagged
      *e = (gen << 24) | 0u;
      atomicOr(status, SCONE_ST_STAGE_OVERFLOW);
    }
  }
}

__global__ __launch_bounds__(256) void k_stage_remap(int32_t *__restrict__ ell, long long ntok, int W, int NC,
                                                     long long n_hot, const uint32_t *__restrict__ slot_of) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long t = gid / NC;
  const int j = (int)(gid - t * NC);
  if (t >= ntok) return;
  const int kown = ell[t * W + W - 2] & 0xFF;
  if (j >= kown) return;
  const long long id = ell[t * W + j];
  if (id < n_hot) return;
  ell[t * W + j] = (int32_t)(n_hot + (slot_of[id - n_hot] & 0xFFFFFFu));
}

// one wave per staged row; 16 bytes per lane per step from mapped host memory.  (Tried: two 512-B rows per
// wave and a 2048-block grid to double the row reads in flight -- 10 % SLOWER, 199 -> 181 M tok/s at C4 size:
// the link is already saturated by 4096 waves, more of them only take CUs from the lookup kernel.)
__global__ __launch_bounds__(256) void k_stage_copy(const uint32_t *__restrict__ count, const int32_t *__restrict__ list,
                                                    scone_row_store host, uint8_t *__restrict__ stage_rows,
                                                    const uint8_t *__restrict__ scales, uint8_t *__restrict__ stage_scales,
                                                    int scale_bytes, long long n_hot, uint32_t cap) {
  const int lane = threadIdx.x & 63;
  uint32_t n = *count;
  if (n > cap) n = cap;
  const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
  for (unsigned s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < n; s += nwaves) {
    const unsigned long long lr = (unsigned long long)list[s];
    const uint4 *src = reinterpret_cast<const uint4 *>(host.row(lr));
    uint4 *dst = reinterpret_cast<uint4 *>(stage_rows + (size_t)s * host.row_bytes);
    for (unsigned v = lane; v < host.row_bytes / 16; v += 64) dst[v] = src[v];
    if (scale_bytes && lane < scale_bytes / 2)
      reinterpret_cast<unsigned short *>(stage_scales + (size_t)(n_hot + s) * scale_bytes)[lane] =
          reinterpret_cast<const unsigned short *>(scales + lr * scale_bytes)[lane];
  }
}

}  // namespace

struct scone_stage_state {
  long long chunk_tokens = 0;     // tokens per chunk actually provisioned
  long long requested_tokens = 0; // what the caller asked for (may exceed chunk_tokens, see prepare)
  uint32_t cap = 0;  // staged rows per buffer
  hipStream_t prep = nullptr, copy = nullptr;
  hipEvent_t prepped[SCONE_STAGE_NBUF] = {}, staged[SCONE_STAGE_NBUF] = {}, consumed[SCONE_STAGE_NBUF] = {}, start = nullptr;
  bool consumed_valid[SCONE_STAGE_NBUF] = {};
  uint32_t *slot_of = nullptr;
  uint32_t gen = 0;
  int32_t *ell[SCONE_STAGE_NBUF] = {};
  int32_t *list[SCONE_STAGE_NBUF] = {};
  uint32_t *count[SCONE_STAGE_NBUF] = {};
  uint8_t *rows[SCONE_STAGE_NBUF] = {};
  uint8_t *scales[SCONE_STAGE_NBUF] = {};
};

void scone_stage_destroy(scone_handle *h) {
  scone_stage_state *st = h->stage;
  if (!st) return;
  if (st->prep) (void)hipStreamDestroy(st->prep);
  if (st->copy) (void)hipStreamDestroy(st->copy);
  for (int b = 0; b < SCONE_STAGE_NBUF; ++b) {
    if (st->prepped[b]) (void)hipEventDestroy(st->prepped[b]);
    if (st->staged[b]) (void)hipEventDestroy(st->staged[b]);
    if (st->consumed[b]) (void)hipEventDestroy(st->consumed[b]);
    if (st->ell[b]) (void)hipFree(st->ell[b]);
    if (st->list[b]) (void)hipFree(st->list[b]);
    if (st->count[b]) (void)hipFree(st->count[b]);
    if (st->rows[b]) (void)hipFree(st->rows[b]);
    if (st->scales[b]) (void)hipFree(st->scales[b]);
  }
  if (st->start) (void)hipEventDestroy(st->start);
  if (st->slot_of) (void)hipFree(st->slot_of);
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
  // 24-bit slot numbers: a chunk may reference chunk_tokens * NC distinct cold rows, and every one of them
  // must get a slot (an unclaimed reference would be remapped outside the staging buffer) -> bound the chunk
  if (chunk_tokens * NC > 0xFFFFFEll && n_cold > 0xFFFFFEll) chunk_tokens = 0xFFFFFEll / NC;
  long long cap = chunk_tokens * NC;
  if (cap > n_cold) cap = n_cold;
  if (const char *ev = getenv("SCONE_STAGE_CAP_ROWS")) {  // test hook: an undersized buffer exercises the overflow path
    const long long forced = atoll(ev);
    if (forced > 0 && forced < cap) cap = forced;
  }
  st->chunk_tokens = chunk_tokens;
  st->requested_tokens = requested;
  st->cap = (uint32_t)cap;
  SCONE_HIP(h, hipStreamCreateWithFlags(&st->prep, hipStreamNonBlocking));
  SCONE_HIP(h, hipStreamCreateWithFlags(&st->copy, hipStreamNonBlocking));
  SCONE_HIP(h, hipEventCreateWithFlags(&st->start, hipEventDisableTiming));
  SCONE_HIP(h, hipMalloc(&st->slot_of, (size_t)(n_cold > 0 ? n_cold : 1) * 4));
  SCONE_HIP(h, hipMemset(st->slot_of, 0, (size_t)(n_cold > 0 ? n_cold : 1) * 4));
  const size_t sb = h->scale_bytes_per_row;
  for (int b = 0; b < SCONE_STAGE_NBUF; ++b) {
    SCONE_HIP(h, hipEventCreateWithFlags(&st->prepped[b], hipEventDisableTiming));
    SCONE_HIP(h, hipEventCreateWithFlags(&st->staged[b], hipEventDisableTiming));
    SCONE_HIP(h, hipEventCreateWithFlags(&st->consumed[b], hipEventDisableTiming));
    SCONE_HIP(h, hipMalloc(&st->ell[b], (size_t)chunk_tokens * W * 4));
    SCONE_HIP(h, hipMalloc(&st->list[b], (size_t)(cap > 0 ? cap : 1) * 4));
    SCONE_HIP(h, hipMalloc(&st->count[b], 4));
    SCONE_HIP(h, hipMalloc(&st->rows[b], (size_t)(cap > 0 ? cap : 1) * h->row_payload_bytes));
    if (sb) {
      // scales indexed like the rows: [0, n_hot) = the HBM-resident head, then the staged rows
      SCONE_HIP(h, hipMalloc(&st->scales[b], (size_t)(h->hot_local + cap) * sb + 4));
      SCONE_HIP(h, hipMemcpy(st->scales[b], h->scales, (size_t)h->hot_local * sb, hipMemcpyDeviceToDevice));
    }
  }
  return SCONE_OK;
}

// side-stream half of one chunk; leaves ell / rows / scales of buffer `buf` ready and records staged[buf]
int scone_stage_chunk(scone_handle *h, int buf, const int32_t *d_tok, int32_t Bc, int32_t T) {
  scone_stage_state *st = h->stage;
  const int W = SCONE_ELL_W(h->cfg.max_n), NC = h->cfg.max_n * (h->cfg.max_n + 1) / 2;
  const long long ntok = (long long)Bc * T;
  hipStream_t s = st->prep;
  if (st->consumed_valid[buf]) SCONE_HIP(h, hipStreamWaitEvent(s, st->consumed[buf], 0));
  st->gen += 1;
  if (st->gen > 255) {  // 8-bit generation tags wrapped: forget every old claim
    SCONE_HIP(h, hipMemsetAsync(st->slot_of, 0, (size_t)(h->local_rows - h->hot_local) * 4, s));
    st->gen = 1;
  }
  int rc = scone_launch_match_ell(h, d_tok, Bc, T, st->ell[buf], s);
  if (rc) return rc;
  SCONE_HIP(h, hipMemsetAsync(st->count[buf], 0, 4, s));
  const long long work = ntok * NC;
  if (!scone_grid_fits((unsigned long long)(work + 255) / 256, 256))
    return scone_fail(h, SCONE_EINVAL, "scone_embed(staged): chunk too large for one launch (lower stage_tokens)");
  const unsigned blocks = (unsigned)((work + 255) / 256);
  hipLaunchKernelGGL(k_stage_claim, dim3(blocks), dim3(256), 0, s, st->ell[buf], ntok, W, NC, (long long)h->hot_local,
                     st->slot_of, st->gen, st->count[buf], st->list[buf], st->cap, h->d_status);
  hipLaunchKernelGGL(k_stage_remap, dim3(blocks), dim3(256), 0, s, st->ell[buf], ntok, W, NC, (long long)h->hot_local,
                     st->slot_of);
  SCONE_HIP(h, hipGetLastError());
  SCONE_HIP(h, hipEventRecord(st->prepped[buf], s));
  // the copy stream only ever waits for the list of THIS chunk, never for the preparation of the next one
  SCONE_HIP(h, hipStreamWaitEvent(st->copy, st->prepped[buf], 0));
  hipLaunchKernelGGL(k_stage_copy, dim3(1024), dim3(256), 0, st->copy, st->count[buf], st->list[buf], scone_store_of(h),
                     st->rows[buf], (const uint8_t *)h->scales, st->scales[buf], (int)h->scale_bytes_per_row,
                     (long long)h->hot_local, st->cap);
  SCONE_HIP(h, hipGetLastError());
  SCONE_HIP(h, hipEventRecord(st->staged[buf], st->copy));
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
uint8_t *scone_stage_rows(scone_handle *h, int buf) { return h->stage->rows[buf]; }
const void *scone_stage_scales(scone_handle *h, int buf) { return h->stage->scales[buf]; }
long long scone_stage_chunk_tokens(scone_handle *h) { return h->stage->chunk_tokens; }
