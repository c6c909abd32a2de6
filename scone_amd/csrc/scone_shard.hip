// Row exchanges for row-sharded tables (tables larger than one GPU; BASELINE config C5).
//
// Every rank holds the replicated index and a contiguous range of table rows.  What crosses xGMI are the QUANTISED ROWS
// the batch references, each DISTINCT row once per destination, never partial sums: an INT4 d=1024 row is 528 B where an
// fp32 partial sum is 4096 B per token and rank, and the receiver reduces the rows in the reference's order, so the result
// is bit-identical to the unsharded table.  Because tokens and index are replicated, both sides of every transfer can work
// out what is sent without asking: no request round.
//
//   plan    ONE match of the batch against all rows (or: rank r matches slice r, the 32-B list records are all-gathered),
//           then claim passes over the lists: every reference to a row I own claims the row (generation-tagged map, one CAS
//           per reference at most) -> the list of distinct rows I contribute, per chunk of sequences
//             - all-gather form ("gather_rows": every rank reduces the whole batch): one generation, a row claimed by an
//               earlier chunk is not sent again
//             - slice exchange ("rows": rank q reduces slice q): chunk q = slice q, a claim table per destination, all
//               destinations in one launch
//   pack    one record [payload | scales | row id] per claimed row -- or, the one-piece all-gather form, three COLUMNS:
//           payload rows at the table's own stride | scales | the SENDER's hash fragment row id -> position
//   (the transfer: torch.distributed / RCCL, or copy-engine pushes -- done by the caller, scone_amd/distributed.py)
//   embed   records: indexed by row id in an open-addressing map, the lists remapped; columns: the lists resolved through the
//           senders' fragments, no indexing pass; then the ordinary lookup kernel reads the rows IN PLACE in the receive
//           buffer (row store stride = record size / payload size)
// (The first form of the slice exchange -- one record per (token, list index) reference, scone_shard_plan / _pack / _embed --
// sent a row covering three tokens of a slice three times: 0.97M records instead of 0.43M on the C5-shaped batch.  Superseded
// in round 2, removed in round 4.)
//
// Replicated head (scone_shard_set_head): global rows [0, n_head) -- the unigrams and the most frequent f-grams of
// the frequency-ordered table -- are kept on every shard in record layout and are neither counted, packed nor
// sent; the lookup kernel reads them as the "hot" part of its row store, the received records as the "cold" part.
// On the C5-shaped workload that removes half of all records and the 10x send imbalance of the rank owning the head.
#include "scone_common.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

struct scone_shard_state {
  long long cap_slice = 0, cap_recv = 0;
  int32_t *ell_slice = nullptr;  // [planned tokens, W] the lists of the planned batch (all rows, compacted); remapped in place by embed
  uint32_t *counters = nullptr;  // [64]: the claim counters of a plan (one per chunk)
  uint8_t *scales = nullptr;     // [n_head + received records] scales: the head's, then the unpacked ones
  unsigned long long n_head = 0; // replicated head: global rows [0, n_head) live on every shard
  uint8_t *head_rows = nullptr;  // [n_head, rec_bytes]: payload at the start of every record-sized slot
  uint8_t *head_scales = nullptr;  // [n_head, scale_bytes_per_row]
  uint8_t *head_rows_p = nullptr;  // [n_head, payload bytes]: the head again at the PAYLOAD stride, for the columns exchange
  bool head_p_stale = true;        // (whose received rows are payload only; rebuilt from head_rows after the head changes)
  hipEvent_t head_p_ready = nullptr;  // recorded behind the rebuild: a lookup on ANOTHER stream waits for it
  uint64_t head_version = 0;       // bumped by every change of the head (callers re-copy the head's scales into their buffers)
  // all-gather form: distinct rows
  uint32_t *uniq_claim = nullptr;  // [local rows]: generation of the last batch that claimed the row
  uint32_t uniq_gen = 0;
  int32_t *uniq_list = nullptr;    // [cap_uniq] claimed row ids
  long long cap_uniq = 0;
  unsigned long long n_uniq = 0;   // records of the current plan
  bool n_on_device = false;        // ... planned without a host round trip: the count is in chunk_ends[0] only (n_uniq = its bound)
  uint32_t *chunk_ends = nullptr;  // [64] value of the claim counter after each chunk of the plan
  // slice exchange (a claim generation per chunk): one claim table and one list region PER CHUNK, so that all chunks claim in
  // ONE launch (the single table forced one launch per chunk, in order: 8 x 17 us + 8 copies at W = 8)
  uint32_t *multi_claim = nullptr;  // [multi_tables, local rows]
  int multi_tables = 0;
  uint32_t multi_gen = 0;
  int32_t *regions = nullptr;       // [n_chunks, region capacity] claimed ids per chunk, compacted into uniq_list afterwards
  long long cap_regions = 0;
  int32_t plan_B = 0, plan_T = 0, plan_chunks = 0;
  long long rhash_cap_now = 0;     // capacity the receiver's map was cleared for (current exchange)
  unsigned long long *rhash = nullptr;  // receiver: open-addressing map row id -> record number
  long long cap_rhash = 0;
  // The lists of the planned batch live in ell_slice -- or in a buffer the CALLER owns (scone_shard_gather_plan_ell:
  // the lists were matched slice by slice on several ranks and all-gathered), borrowed until the batch has been reduced.
  int32_t *ell_ext = nullptr;
  std::vector<uint8_t> remapped;   // per sequence of the planned batch: its lists already hold record numbers
  unsigned long long added = 0;    // records added to the row map in the current exchange
  // Plan slots (scone_shard_select_slot): the receiver-side state of a planned batch -- its id lists, the scales of
  // [head | received records], the row map -- exists up to SCONE_SHARD_SLOTS times, so that batches b + 1 (and b + 2) can be
  // planned, packed and exchanged on side streams while batch b is still being reduced.  The fields above are the ACTIVE
  // slot's; the others are parked.
  struct plan_slot {
    int32_t *ell_slice = nullptr;
    long long cap_slice = 0;
    uint8_t *scales = nullptr;
    long long cap_recv = 0;
    unsigned long long *rhash = nullptr;
    long long cap_rhash = 0, rhash_cap_now = 0;
    int32_t plan_B = 0, plan_T = 0, plan_chunks = 0;
    int32_t *ell_ext = nullptr;
    std::vector<uint8_t> remapped;
    unsigned long long added = 0;
  };
#define SCONE_SHARD_SLOTS 4
  plan_slot parked[SCONE_SHARD_SLOTS];  // parked[k]: state of slot k while it is not the active one (parked[slot] is unused)
  int slot = 0;
};

namespace {

#define SHARD_BLOCKS 1024  // persistent grids of the per-token passes

template <typename T>
int grow(scone_handle *h, T **p, long long *cap, long long need, size_t elems_per) {
  if (need <= *cap && *p) return SCONE_OK;
  if (*p) SCONE_HIP(h, hipFree(*p));
  *p = nullptr;
  *cap = 0;
  SCONE_HIP(h, hipMalloc(p, (size_t)(need > 0 ? need : 1) * elems_per * sizeof(T) + 16));
  *cap = need;
  return SCONE_OK;
}

}  // namespace

void scone_shard_destroy(scone_handle *h) {
  scone_shard_state *st = h->shard;
  if (!st) return;
  void *ptrs[] = {st->ell_slice, st->counters, st->scales,
                  st->head_rows, st->head_scales, st->uniq_claim, st->uniq_list, st->rhash, st->chunk_ends,
                  st->head_rows_p, st->multi_claim, st->regions};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (st->head_p_ready) (void)hipEventDestroy(st->head_p_ready);
  for (int k = 0; k < SCONE_SHARD_SLOTS; ++k) {
    if (k == st->slot) continue;
    void *pp[] = {st->parked[k].ell_slice, st->parked[k].scales, st->parked[k].rhash};
    for (void *p : pp)
      if (p) (void)hipFree(p);
  }
  delete st;
  h->shard = nullptr;
}

uint8_t *scone_shard_head(const scone_handle *h, unsigned long long *n_head) {
  *n_head = h->shard ? h->shard->n_head : 0;
  return h->shard ? h->shard->head_rows : nullptr;
}

int scone_shard_rec_bytes(const scone_handle *h) {
  // [payload | scales | 8-byte header], rounded up to 16 B.  (Records padded to 64 / 128 B so that every row the lookup reads
  // in place starts on a sector / cache-line boundary were measured in round 3: -1 % step time for +6 / +18 % wire; removed.
  // The columns exchange gets the alignment for free: its payload rows travel at the table's own stride.)
  const size_t al = 16;
  return (int)((h->row_payload_bytes + h->scale_bytes_per_row + 8 + al - 1) / al * al);
}

extern "C" int scone_shard_record_bytes(scone_handle *h, uint64_t *bytes) {
  if (!h || !bytes) return SCONE_EINVAL;
  if (h->cfg.dim <= 0) return scone_fail(h, SCONE_ESTATE, "scone_shard_record_bytes: handle has no table");
  *bytes = (uint64_t)scone_shard_rec_bytes(h);
  return SCONE_OK;
}

// ---------------------------------------------------------------- replicated head
static scone_row_store head_store(const scone_handle *h) {
  scone_row_store st;
  st.hot = h->shard->head_rows, st.cold = nullptr;
  st.n_hot = h->shard->n_head;
  st.row_bytes = (unsigned int)scone_shard_rec_bytes(h);
  return st;
}

extern "C" int scone_shard_set_head(scone_handle *h, uint64_t n_head) {
  if (!h) return SCONE_EINVAL;
  if (h->cfg.dim <= 0 || !h->rows) return scone_fail(h, SCONE_ESTATE, "scone_shard_set_head: handle has no table");
  if (h->cfg.placement != SCONE_PLACE_HBM) return scone_fail(h, SCONE_EINVAL, "scone_shard_set_head: HBM tables only");
  if (n_head > h->cfg.n_rows) n_head = h->cfg.n_rows;
  if (n_head > 0x7FFFFFFFull) return scone_fail(h, SCONE_EINVAL, "scone_shard_set_head: head too large");
  SCONE_ON_DEVICE(h);
  if (!h->shard) {
    h->shard = new (std::nothrow) scone_shard_state();
    if (!h->shard) return scone_fail(h, SCONE_ENOMEM, "scone_shard_set_head: out of memory");
  }
  scone_shard_state *st = h->shard;
  if (st->head_rows) SCONE_HIP(h, hipFree(st->head_rows));
  if (st->head_scales) SCONE_HIP(h, hipFree(st->head_scales));
  if (st->head_rows_p) SCONE_HIP(h, hipFree(st->head_rows_p));
  st->head_rows = st->head_scales = st->head_rows_p = nullptr;
  st->head_p_stale = true;
  st->head_version += 1;
  st->n_head = 0;
  if (n_head == 0) return SCONE_OK;
  const size_t rec = (size_t)scone_shard_rec_bytes(h);
  SCONE_HIP(h, hipMalloc(&st->head_rows, n_head * rec));
  SCONE_HIP(h, hipMemset(st->head_rows, 0, n_head * rec));
  if (h->scale_bytes_per_row) {
    SCONE_HIP(h, hipMalloc(&st->head_scales, n_head * h->scale_bytes_per_row + 4));
    SCONE_HIP(h, hipMemset(st->head_scales, 0, n_head * h->scale_bytes_per_row + 4));
  }
  st->n_head = n_head;
  return SCONE_OK;
}

extern "C" int scone_shard_head_store_f32(scone_handle *h, const float *d_rows_f32, uint64_t row0, uint64_t nrows,
                                          scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (!h->shard || !h->shard->n_head) return scone_fail(h, SCONE_ESTATE, "scone_shard_head_store_f32: call scone_shard_set_head first");
  if (nrows == 0) return SCONE_OK;
  if (!d_rows_f32) return scone_fail(h, SCONE_EINVAL, "scone_shard_head_store_f32: null rows");
  if (row0 + nrows > h->shard->n_head) return scone_fail(h, SCONE_ERANGE, "scone_shard_head_store_f32: rows outside [0, n_head)");
  SCONE_ON_DEVICE(h);
  h->shard->head_p_stale = true;
  h->shard->head_version += 1;
  return scone_store_f32_into(h, head_store(h), h->shard->head_scales, 0, h->shard->n_head, d_rows_f32, nullptr, row0, nrows,
                              (hipStream_t)stream);
}

int scone_shard_fill_head_synth(scone_handle *h, uint32_t seed, float base_scale, hipStream_t s) {
  if (!h->shard || !h->shard->n_head) return SCONE_OK;
  h->shard->head_p_stale = true;
  h->shard->head_version += 1;
  return scone_fill_synth_into(h, head_store(h), h->shard->head_scales, 0, h->shard->n_head, seed, base_scale, s);
}

// ---------------------------------------------------------------- all-gather form: one record per distinct row
namespace {

// every reference of the batch to a row I own claims the row for this batch; the first claimer of a row appends it to
// the list.  One CAS per reference at most, nobody waits.
#define GATHER_STASH 4096
// 1024-thread workgroups: every workgroup ends with ONE global atomic on the one list counter (~90 retire per microsecond
// on one address: 1024 workgroups of 256 spent 11 us of a 25 us pass there); 256 workgroups of 1024 keep the same number
// of threads in flight with a quarter of those atomics
#define CLAIM_THREADS 1024
#define CLAIM_BLOCKS 512
__device__ __forceinline__ void claim_run(const int32_t *__restrict__ ell, long long ntok, int W, int NC,
                                          long long row_begin, long long send_begin, long long row_end,
                                          uint32_t *__restrict__ claim, uint32_t gen,
                                          uint32_t *__restrict__ count, int32_t *__restrict__ list, long long cap) {
  // the workgroup's claimed ids are stashed in LDS and appended to the list with ONE global atomic at the end (a global
  // atomic per claimed row on the one counter would serialise: ~90 per microsecond); a stash overflow appends directly
  __shared__ int32_t stash[GATHER_STASH];
  __shared__ uint32_t n_stash, base;
  if (threadIdx.x == 0) n_stash = 0;
  __syncthreads();
  const long long per = (ntok + gridDim.x - 1) / gridDim.x;
  const long long t0 = (long long)blockIdx.x * per, t1 = t0 + per < ntok ? t0 + per : ntok;
  for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    int32_t r8[8];
    if (W == 8) {  // the whole record with two 16-B loads instead of up to seven 4-B ones
      const int4 a = reinterpret_cast<const int4 *>(ell + t * 8)[0], b = reinterpret_cast<const int4 *>(ell + t * 8)[1];
      r8[0] = a.x, r8[1] = a.y, r8[2] = a.z, r8[3] = a.w, r8[4] = b.x, r8[5] = b.y, r8[6] = b.z, r8[7] = b.w;
    }
    const int kown = (W == 8 ? r8[6] : ell[t * W + W - 2]) & 0xFF;  // the lists against ALL rows (compacted): pick the ids I own
    for (int j = 0; j < NC; ++j) {
      if (j >= kown) break;
      long long id;
      if (W == 8) {
        id = r8[0];
#pragma unroll
        for (int q = 1; q < 6; ++q) id = j == q ? r8[q] : id;
      } else {
        id = ell[t * W + j];
      }
      if (id < send_begin || id >= row_end) continue;  // another shard's row, or a row of the replicated head
      uint32_t *e = &claim[id - row_begin];
      const uint32_t v = *e;
      if (v == gen) continue;  // already claimed for this batch
      if (atomicCAS(e, v, gen) == v) {
        const uint32_t k = atomicAdd(&n_stash, 1u);
        if (k < GATHER_STASH) {
          stash[k] = (int32_t)id;
        } else {
          const uint32_t s = atomicAdd(count, 1u);
          if ((long long)s < cap) list[s] = (int32_t)id;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t n = n_stash < GATHER_STASH ? n_stash : GATHER_STASH;
  if (threadIdx.x == 0) base = n ? atomicAdd(count, n) : 0u;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x)
    if ((long long)(base + k) < cap) list[base + k] = stash[k];
}

__global__ __launch_bounds__(CLAIM_THREADS) void k_gather_claim(const int32_t *__restrict__ ell, long long ntok, int W, int NC,
                                                                 long long row_begin, long long send_begin, long long row_end,
                                                                 uint32_t *__restrict__ claim, uint32_t gen,
                                                                 uint32_t *__restrict__ count, int32_t *__restrict__ list, long long cap) {
  claim_run(ell, ntok, W, NC, row_begin, send_begin, row_end, claim, gen, count, list, cap);
}

// every chunk of a slice-exchange plan in one launch: blockIdx.y = chunk, with its own claim table, counter and list region
__global__ __launch_bounds__(CLAIM_THREADS) void k_gather_claim_chunks(const int32_t *__restrict__ ell, int B, int T, int n_chunks, int W, int NC,
                                                                       long long row_begin, long long send_begin, long long row_end,
                                                                       uint32_t *__restrict__ claims, long long local_rows, uint32_t gen,
                                                                       uint32_t *__restrict__ counts, int32_t *__restrict__ regions,
                                                                       long long region_cap) {
  const int q = blockIdx.y;
  const long long per = ((long long)B + n_chunks - 1) / n_chunks;
  long long s0 = q * per, s1 = s0 + per;
  s0 = s0 < B ? s0 : B, s1 = s1 < B ? s1 : B;
  claim_run(ell + s0 * T * W, (s1 - s0) * T, W, NC, row_begin, send_begin, row_end, claims + (long long)q * local_rows, gen, counts + q,
            regions + (long long)q * region_cap, region_cap);
}

struct region_offsets {
  unsigned int off[65];
};
// the chunks' regions laid end to end: the list the pack walks (chunk q's ids at [off[q], off[q + 1]))
__global__ __launch_bounds__(256) void k_gather_compact(const int32_t *__restrict__ regions, long long region_cap, const region_offsets ro,
                                                        int32_t *__restrict__ list) {
  const int q = blockIdx.y;
  const unsigned int n = ro.off[q + 1] - ro.off[q];
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    list[ro.off[q] + i] = regions[(long long)q * region_cap + i];
}

// one wave per claimed row: [payload | scales | row id, marker]
__global__ __launch_bounds__(256) void k_gather_pack(const int32_t *__restrict__ list, unsigned long long n, scone_row_store st,
                                                     long long row_begin, const uint8_t *__restrict__ scales, int scale_bytes,
                                                     int rec_bytes, uint8_t *__restrict__ out, int lanes_per_rec) {
  // `lanes_per_rec` lanes (16, 32 or 64: a 512-B INT4 row is 32 x 16 B) copy one record, so a wave carries 64 / lanes_per_rec
  // records at once: the pack is a chain of two dependent misses per record (list entry -> row), its time the number of
  // rounds the records take through the resident waves
  const unsigned sub = (threadIdx.x & 63) / lanes_per_rec, l = (threadIdx.x & 63) % lanes_per_rec;
  const unsigned per_wave = 64 / lanes_per_rec;
  const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned long long nwaves = (unsigned long long)gridDim.x * (blockDim.x >> 6);
  for (unsigned long long p = wave * per_wave + sub; p < n; p += nwaves * per_wave) {
    const long long id = list[p];
    const unsigned long long lr = (unsigned long long)(id - row_begin);
    const uint4 *src = reinterpret_cast<const uint4 *>(st.row(lr));
    uint8_t *rec = out + p * (unsigned long long)rec_bytes;
    uint4 *dst = reinterpret_cast<uint4 *>(rec);
    for (unsigned v = l; v < st.row_bytes / 16; v += lanes_per_rec) dst[v] = src[v];
    for (unsigned b = l; b < (unsigned)scale_bytes / 2; b += lanes_per_rec)
      reinterpret_cast<unsigned short *>(rec + st.row_bytes)[b] = reinterpret_cast<const unsigned short *>(scales + lr * scale_bytes)[b];
    if (l == 0) {
      uint32_t *hdr = reinterpret_cast<uint32_t *>(rec + rec_bytes - 8);
      hdr[0] = (uint32_t)id;
      hdr[1] = 0xFFFFFFFFu;
    }
  }
}

// records [first, first + n) of a send buffer become padding
__global__ __launch_bounds__(256) void k_gather_mark_pad(uint8_t *__restrict__ buf, unsigned long long first, unsigned long long n,
                                                         int rec_bytes) {
  const unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t *hdr = reinterpret_cast<uint32_t *>(buf + (first + p) * (unsigned long long)rec_bytes + rec_bytes - 8);
  hdr[0] = 0xFFFFFFFFu;
  hdr[1] = 0xFFFFFFFFu;
}

// receiver: record p holds row id hdr[0] -> hash map id -> p (key id + 1, empty 0; smallest p wins on a duplicate), scales
// `recv` points at record number `record0` of the gathered buffer; `scales` at its scale slot.  Padding records (a rank's
// contribution is padded to the largest one of its all-gather; the sender marks them with row id 0xFFFFFFFF) are skipped.
__global__ __launch_bounds__(256) void k_gather_index(const uint8_t *__restrict__ recv, unsigned long long n_recv, int rec_bytes,
                                                      int row_bytes, int scale_bytes, unsigned long long *__restrict__ rhash,
                                                      unsigned long long hmask, uint8_t *__restrict__ scales,
                                                      unsigned long long record0) {
  unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_recv) return;
  const uint8_t *rec = recv + p * (unsigned long long)rec_bytes;
  const uint32_t id = reinterpret_cast<const uint32_t *>(rec + rec_bytes - 8)[0];
  if (id == 0xFFFFFFFFu) return;
  if (scale_bytes == 16) {
    *reinterpret_cast<uint4 *>(scales + p * 16) = *reinterpret_cast<const uint4 *>(rec + row_bytes);
  } else {
    for (int b = 0; b < scale_bytes / 2; ++b)
      reinterpret_cast<unsigned short *>(scales + p * scale_bytes)[b] = reinterpret_cast<const unsigned short *>(rec + row_bytes)[b];
  }
  p += record0;
  const unsigned long long key = (unsigned long long)id + 1ull, mine = (key << 32) | p;
  unsigned long long s = scone_hash_key(key, 0u) & hmask;
  for (unsigned long long probe = 0; probe <= hmask; ++probe) {
    const unsigned long long old = atomicCAS(&rhash[s], 0ull, mine);
    if (old == 0ull) return;
    if ((old >> 32) == key) {
      atomicMin(&rhash[s], mine);
      return;
    }
    s = (s + 1ull) & hmask;
  }
}

__global__ __launch_bounds__(256) void k_gather_remap(int32_t *__restrict__ ell, long long ntok, int W, int NC,
                                                      const unsigned long long *__restrict__ rhash, unsigned long long hmask,
                                                      long long n_head, uint32_t *__restrict__ status) {
  const long long per = (ntok + gridDim.x - 1) / gridDim.x;
  const long long t0 = (long long)blockIdx.x * per, t1 = t0 + per < ntok ? t0 + per : ntok;
  for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    int32_t r8[8];
    if (W == 8) {  // W = 8 (max_n <= 3): the record travels as two 16-B accesses each way, written back only if it changed
      const int4 a = reinterpret_cast<const int4 *>(ell + t * 8)[0], b = reinterpret_cast<const int4 *>(ell + t * 8)[1];
      r8[0] = a.x, r8[1] = a.y, r8[2] = a.z, r8[3] = a.w, r8[4] = b.x, r8[5] = b.y, r8[6] = b.z, r8[7] = b.w;
    }
    const int kown = (W == 8 ? r8[6] : ell[t * W + W - 2]) & 0xFF;
    bool dirty = false;
    for (int j = 0; j < NC; ++j) {
      if (j >= kown) break;
      long long id;
      if (W == 8) {
        id = r8[0];
#pragma unroll
        for (int q = 1; q < 6; ++q) id = j == q ? r8[q] : id;
      } else {
        id = ell[t * W + j];
      }
      if (id < n_head) continue;  // a head row: its id is its row number in the lookup's row store
      const unsigned long long key = (unsigned long long)id + 1ull;
      unsigned long long s = scone_hash_key(key, 0u) & hmask;
      long long slot = -1;
      for (unsigned long long probe = 0; probe <= hmask; ++probe) {
        const unsigned long long v = rhash[s];
        if (v == 0ull) break;
        if ((v >> 32) == key) {
          slot = (long long)(v & 0xFFFFFFFFull);
          break;
        }
        s = (s + 1ull) & hmask;
      }
      if (slot < 0) {  // the row did not arrive (the ranks disagree about the batch): report, never read out of bounds
        atomicOr(status, SCONE_ST_BAD_ID);
        slot = 0;
      }
      if (W == 8) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (j == q) r8[q] = (int32_t)(n_head + slot);
        dirty = true;
      } else {
        ell[t * W + j] = (int32_t)(n_head + slot);
      }
    }
    if (W == 8 && dirty) {
      reinterpret_cast<int4 *>(ell + t * 8)[0] = make_int4(r8[0], r8[1], r8[2], r8[3]);
      reinterpret_cast<int4 *>(ell + t * 8)[1] = make_int4(r8[4], r8[5], r8[6], r8[7]);
    }
  }
}

}  // namespace

// chunk c of a batch of B sequences cut into n_chunks: sequences [c * ceil(B / n_chunks), ...)
static void chunk_seqs(int32_t B, int32_t n_chunks, int32_t c, int32_t *s0, int32_t *s1) {
  const long long per = ((long long)B + n_chunks - 1) / n_chunks;
  const long long a = c * per, b = a + per;
  *s0 = (int32_t)(a < B ? a : B);
  *s1 = (int32_t)(b < B ? b : B);
}

static scone_shard_state *shard_state(scone_handle *h) {
  if (!h->shard) h->shard = new (std::nothrow) scone_shard_state();
  return h->shard;
}

// the lists of the planned batch: the library's own buffer, or the one the caller lent (scone_shard_gather_plan_ell)
static inline int32_t *plan_lists(scone_shard_state *st) { return st->ell_ext ? st->ell_ext : st->ell_slice; }

// The claim passes of a plan over the lists `ell` [B*T, W] (matched against ALL rows, compacted).  One claim pass per
// chunk, in chunk order.  dedup_across_chunks (the all-gather form): all in ONE generation -- a row already claimed by an
// earlier chunk is not sent again, the receiver's row map is cumulative and chunk c is reduced after the records of
// chunks 0..c arrived.  Otherwise (the slice exchange: chunk q = what rank q's slice needs from me) every chunk claims
// in a generation of its own: each destination gets every distinct row it references, once.
static int plan_claims(scone_handle *h, scone_shard_state *st, const int32_t *ell, int32_t B, int32_t T, int32_t n_chunks,
                       int32_t dedup_across_chunks, uint64_t *h_chunk_end, hipStream_t s) {
  const int W = SCONE_ELL_W(h->cfg.max_n), NC = h->cfg.max_n * (h->cfg.max_n + 1) / 2;
  const long long ntok = (long long)B * T;
  if (!st->counters) SCONE_HIP(h, hipMalloc(&st->counters, 64 * sizeof(uint32_t)));
  if (!st->chunk_ends) SCONE_HIP(h, hipMalloc(&st->chunk_ends, 64 * sizeof(uint32_t)));
  if (!st->uniq_claim) {
    SCONE_HIP(h, hipMalloc(&st->uniq_claim, (size_t)(h->local_rows ? h->local_rows : 1) * sizeof(uint32_t)));
    SCONE_HIP(h, hipMemset(st->uniq_claim, 0, (size_t)(h->local_rows ? h->local_rows : 1) * sizeof(uint32_t)));
    st->uniq_gen = 0;
  }
  st->uniq_gen += 1;
  if (st->uniq_gen == 0) {  // 2^32 batches: forget every old claim
    SCONE_HIP(h, hipMemsetAsync(st->uniq_claim, 0, (size_t)(h->local_rows ? h->local_rows : 1) * sizeof(uint32_t), s));
    st->uniq_gen = 1;
  }
  // list capacity: every reference could be a distinct row of mine; a row is listed once per plan (all-gather form) or
  // once per CHUNK (slice exchange: every destination gets its own copy)
  const long long rows_cap = (long long)h->local_rows * (dedup_across_chunks ? 1 : n_chunks);
  long long need = ntok * NC < rows_cap ? ntok * NC : rows_cap;
  int rc = grow(h, &st->uniq_list, &st->cap_uniq, need, 1);
  if (rc) return rc;
  SCONE_HIP(h, hipMemsetAsync(st->counters, 0, 64 * sizeof(uint32_t), s));
  const long long n_head = (long long)st->n_head;
  const long long send_begin = (long long)h->cfg.row_begin > n_head ? (long long)h->cfg.row_begin : n_head;
  if (!dedup_across_chunks && n_chunks > 1) {
    // slice exchange: every chunk (= destination) claims independently -> a claim table per chunk, ONE launch for all
    const size_t lr = (size_t)(h->local_rows ? h->local_rows : 1);
    if (st->multi_tables < n_chunks) {
      if (st->multi_claim) SCONE_HIP(h, hipFree(st->multi_claim));
      st->multi_claim = nullptr, st->multi_tables = 0;
      SCONE_HIP(h, hipMalloc(&st->multi_claim, lr * (size_t)n_chunks * sizeof(uint32_t)));
      SCONE_HIP(h, hipMemsetAsync(st->multi_claim, 0, lr * (size_t)n_chunks * sizeof(uint32_t), s));
      st->multi_tables = n_chunks, st->multi_gen = 0;
    }
    st->multi_gen += 1;
    if (st->multi_gen == 0) {
      SCONE_HIP(h, hipMemsetAsync(st->multi_claim, 0, lr * (size_t)st->multi_tables * sizeof(uint32_t), s));
      st->multi_gen = 1;
    }
    const long long per_seqs = ((long long)B + n_chunks - 1) / n_chunks;
    long long region_cap = per_seqs * T * NC < (long long)lr ? per_seqs * T * NC : (long long)lr;
    if (region_cap < 1) region_cap = 1;
    rc = grow(h, &st->regions, &st->cap_regions, region_cap * n_chunks, 1);
    if (rc) return rc;
    const long long nt_max = per_seqs * T;
    const long long want = (nt_max + CLAIM_THREADS - 1) / CLAIM_THREADS;
    long long bx = CLAIM_BLOCKS / n_chunks > 0 ? CLAIM_BLOCKS / n_chunks : 1;  // the same number of threads in flight as one pass
    if (want < bx) bx = want;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_gather_claim_chunks, dim3((unsigned)bx, (unsigned)n_chunks), dim3(CLAIM_THREADS), 0, s, ell, (int)B, (int)T,
                       (int)n_chunks, W, NC, (long long)h->cfg.row_begin, send_begin, (long long)h->cfg.row_end, st->multi_claim,
                       (long long)lr, st->multi_gen, st->counters, st->regions, region_cap);
    SCONE_HIP(h, hipGetLastError());
    uint32_t cnt[64];
    SCONE_HIP(h, hipMemcpyAsync(cnt, st->counters, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SCONE_HIP(h, hipStreamSynchronize(s));
    region_offsets ro = {};
    unsigned long long total = 0;
    for (int c = 0; c < n_chunks; ++c) {
      if ((long long)cnt[c] > region_cap) {  // cannot happen: a chunk lists at most min(its references, my rows) ids
        st->n_uniq = 0;
        return scone_fail(h, SCONE_ERANGE, "scone_shard_gather_plan: claim list overflow");
      }
      ro.off[c] = (unsigned int)total;
      total += cnt[c];
      h_chunk_end[c] = total;
    }
    ro.off[n_chunks] = (unsigned int)total;
    if ((long long)total > st->cap_uniq) {
      st->n_uniq = 0;
      return scone_fail(h, SCONE_ERANGE, "scone_shard_gather_plan: claim list overflow");
    }
    if (total)
      hipLaunchKernelGGL(k_gather_compact, dim3(64, (unsigned)n_chunks), dim3(256), 0, s, st->regions, region_cap, ro, st->uniq_list);
    SCONE_HIP(h, hipGetLastError());
    st->n_uniq = total;
    return SCONE_OK;
  }
  for (int c = 0; c < n_chunks; ++c) {
    int32_t s0, s1;
    chunk_seqs(B, n_chunks, c, &s0, &s1);
    if (c > 0 && !dedup_across_chunks) {
      st->uniq_gen += 1;
      if (st->uniq_gen == 0) {
        SCONE_HIP(h, hipMemsetAsync(st->uniq_claim, 0, (size_t)(h->local_rows ? h->local_rows : 1) * sizeof(uint32_t), s));
        st->uniq_gen = 1;
      }
    }
    const long long nt = (long long)(s1 - s0) * T;
    if (nt > 0) {
      const long long want = (nt + CLAIM_THREADS - 1) / CLAIM_THREADS;
      const unsigned blocks = (unsigned)(want < CLAIM_BLOCKS ? want : CLAIM_BLOCKS);
      hipLaunchKernelGGL(k_gather_claim, dim3(blocks), dim3(CLAIM_THREADS), 0, s, ell + (long long)s0 * T * W, nt, W, NC,
                         (long long)h->cfg.row_begin, send_begin, (long long)h->cfg.row_end, st->uniq_claim, st->uniq_gen,
                         st->counters, st->uniq_list, st->cap_uniq);
    }
    SCONE_HIP(h, hipMemcpyAsync(st->chunk_ends + c, st->counters, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  }
  SCONE_HIP(h, hipGetLastError());
  st->n_on_device = false;
  if (!h_chunk_end) {  // sync-free plan (one chunk): the count stays in chunk_ends[0]; pack with scone_shard_cols_pack_cap
    st->n_uniq = (unsigned long long)st->cap_uniq;
    st->n_on_device = true;
    return SCONE_OK;
  }
  uint32_t ends[64];
  SCONE_HIP(h, hipMemcpyAsync(ends, st->chunk_ends, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  SCONE_HIP(h, hipStreamSynchronize(s));
  for (int c = 0; c < n_chunks; ++c) h_chunk_end[c] = ends[c];
  st->n_uniq = ends[n_chunks - 1];
  if ((long long)st->n_uniq > st->cap_uniq) {  // cannot happen with the capacity above; never hand out records that were not listed
    st->n_uniq = 0;
    return scone_fail(h, SCONE_ERANGE, "scone_shard_gather_plan: claim list overflow");
  }
  return SCONE_OK;
}

static int plan_check(scone_handle *h, int32_t B, int32_t T, int32_t n_chunks, const void *p1, const void *p2) {
  // (p2: the host array of chunk ends -- or, for the sync-free plans, any non-null pointer)
  if (!h) return SCONE_EINVAL;
  if (h->cfg.dim <= 0 || !h->rows) return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_plan: handle has no table");
  if (B < 0 || T <= 0 || !p1 || !p2 || n_chunks < 1 || n_chunks > 64)
    return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_plan: bad argument (1 <= n_chunks <= 64)");
  if (h->cfg.placement != SCONE_PLACE_HBM) return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_plan: HBM tables only");
  if (!scone_grid_fits((unsigned long long)((long long)B * T + 255) / 256, 256))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_plan: too many tokens for one launch");
  return SCONE_OK;
}

static void plan_reset(scone_shard_state *st, int32_t B, int32_t T, int32_t n_chunks, int32_t *ell_ext) {
  st->n_uniq = 0;
  st->n_on_device = false;  // (set again by a sync-free plan, at the end of plan_claims)
  st->plan_B = B, st->plan_T = T, st->plan_chunks = n_chunks;
  st->ell_ext = ell_ext;
  st->remapped.assign((size_t)(B > 0 ? B : 0), 0);
  st->added = 0;
}

extern "C" int scone_shard_gather_plan_chunks(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t n_chunks,
                                              int32_t dedup_across_chunks, uint64_t *h_chunk_end, scone_stream_t stream) {
  int rc = plan_check(h, B, T, n_chunks, d_tok, h_chunk_end);
  if (rc) return rc;
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  scone_shard_state *st = shard_state(h);
  if (!st) return scone_fail(h, SCONE_ENOMEM, "scone_shard_gather_plan: out of memory");
  for (int c = 0; c < n_chunks; ++c) h_chunk_end[c] = 0;
  plan_reset(st, B, T, n_chunks, nullptr);
  const int W = SCONE_ELL_W(h->cfg.max_n);
  const long long ntok = (long long)B * T;
  if (ntok == 0) return SCONE_OK;
  long long cs = st->cap_slice;
  rc = grow(h, &st->ell_slice, &cs, ntok, (size_t)W);
  if (rc) return rc;
  st->cap_slice = cs;
  // ONE match of the whole batch against ALL rows: the lists the lookup kernel will walk, and what the claim passes filter
  rc = scone_launch_match_ell_ex(h, d_tok, B, T, st->ell_slice, 0, (long long)h->cfg.n_rows, 0, s);
  if (rc) return rc;
  return plan_claims(h, st, st->ell_slice, B, T, n_chunks, dedup_across_chunks, h_chunk_end, s);
}

// The match of a plan, SHARDED over the ranks.  Index and tokens are replicated, so every rank can match any part of the
// batch -- and in the all-gather form every rank needs the lists of the WHOLE batch (it reduces all of it), which made
// the match the largest helper kernel of the step: 98 us per rank against the 34 GB index of 1e9 keys, every probe an
// HBM miss.  Matching is per sequence, so rank r matches slice r only (scone_shard_gather_match: sequences
// [seq_begin, seq_end) of the batch, 32 B per token into a buffer of the caller), the slices are all-gathered (32 MB per
// 1M tokens: a transfer, not GPU work) and the claim passes run over the gathered lists (scone_shard_gather_plan_ell).
extern "C" int scone_shard_gather_match(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, int32_t seq_begin,
                                        int32_t seq_end, int32_t *d_ell_out, scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (B < 0 || T <= 0 || seq_begin < 0 || seq_end < seq_begin || seq_end > B)
    return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_match: bad sequence range");
  if (seq_end == seq_begin) return SCONE_OK;
  if (!d_tok || !d_ell_out) return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_match: null pointer");
  SCONE_ON_DEVICE(h);
  return scone_launch_match_ell_ex(h, d_tok + (long long)seq_begin * T, seq_end - seq_begin, T, d_ell_out, 0,
                                   (long long)h->cfg.n_rows, 0, (hipStream_t)stream);
}

extern "C" int scone_shard_gather_plan_ell(scone_handle *h, int32_t *d_ell, int32_t B, int32_t T, int32_t n_chunks,
                                           int32_t dedup_across_chunks, uint64_t *h_chunk_end, scone_stream_t stream) {
  int rc = plan_check(h, B, T, n_chunks, d_ell, h_chunk_end);
  if (rc) return rc;
  SCONE_ON_DEVICE(h);
  scone_shard_state *st = shard_state(h);
  if (!st) return scone_fail(h, SCONE_ENOMEM, "scone_shard_gather_plan: out of memory");
  for (int c = 0; c < n_chunks; ++c) h_chunk_end[c] = 0;
  plan_reset(st, B, T, n_chunks, d_ell);
  if ((long long)B * T == 0) return SCONE_OK;
  return plan_claims(h, st, d_ell, B, T, n_chunks, dedup_across_chunks, h_chunk_end, (hipStream_t)stream);
}

// The same two plans WITHOUT the host round trip (one chunk, all-gather form): match / claim are enqueued and the call returns;
// the number of claimed rows stays on the device, where scone_shard_cols_pack_cap reads it.  A serving loop sizes its transfers
// from the previous batches instead (distributed.py: the counts travel inside the fragment column and are read back when
// the batch is reduced -- off the critical path).
extern "C" int scone_shard_gather_plan_async(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, scone_stream_t stream) {
  int rc = plan_check(h, B, T, 1, d_tok, d_tok);
  if (rc) return rc;
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  scone_shard_state *st = shard_state(h);
  if (!st) return scone_fail(h, SCONE_ENOMEM, "scone_shard_gather_plan: out of memory");
  plan_reset(st, B, T, 1, nullptr);
  const int W = SCONE_ELL_W(h->cfg.max_n);
  const long long ntok = (long long)B * T;
  if (ntok == 0) return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_plan_async: empty batch");
  long long cs = st->cap_slice;
  rc = grow(h, &st->ell_slice, &cs, ntok, (size_t)W);
  if (rc) return rc;
  st->cap_slice = cs;
  rc = scone_launch_match_ell_ex(h, d_tok, B, T, st->ell_slice, 0, (long long)h->cfg.n_rows, 0, s);
  if (rc) return rc;
  return plan_claims(h, st, st->ell_slice, B, T, 1, 1, nullptr, s);
}

extern "C" int scone_shard_gather_plan_ell_async(scone_handle *h, int32_t *d_ell, int32_t B, int32_t T, scone_stream_t stream) {
  int rc = plan_check(h, B, T, 1, d_ell, d_ell);
  if (rc) return rc;
  SCONE_ON_DEVICE(h);
  scone_shard_state *st = shard_state(h);
  if (!st) return scone_fail(h, SCONE_ENOMEM, "scone_shard_gather_plan: out of memory");
  plan_reset(st, B, T, 1, d_ell);
  if ((long long)B * T == 0) return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_plan_ell_async: empty batch");
  return plan_claims(h, st, d_ell, B, T, 1, 1, nullptr, (hipStream_t)stream);
}

extern "C" int scone_ell_width(scone_handle *h, uint32_t *ints_per_token) {
  if (!h || !ints_per_token) return SCONE_EINVAL;
  *ints_per_token = (uint32_t)SCONE_ELL_W(h->cfg.max_n);
  return SCONE_OK;
}

// Two plan slots: everything a planned batch needs on the RECEIVING side until it has been reduced (scone_shard_state::
// plan_slot).  Selecting a slot is a host-side swap; the calls that follow -- plan, add_records, embed_range -- work on
// it.  Sender-side scratch (claim table, record list, counters) is shared: plan + pack of one batch are finished (in
// stream order) before the next plan starts.
static void swap_slot(scone_shard_state *st, scone_shard_state::plan_slot &p) {
  std::swap(st->ell_slice, p.ell_slice);
  std::swap(st->cap_slice, p.cap_slice);
  std::swap(st->scales, p.scales);
  std::swap(st->cap_recv, p.cap_recv);
  std::swap(st->rhash, p.rhash);
  std::swap(st->cap_rhash, p.cap_rhash);
  std::swap(st->rhash_cap_now, p.rhash_cap_now);
  std::swap(st->plan_B, p.plan_B);
  std::swap(st->plan_T, p.plan_T);
  std::swap(st->plan_chunks, p.plan_chunks);
  std::swap(st->ell_ext, p.ell_ext);
  st->remapped.swap(p.remapped);
  std::swap(st->added, p.added);
}

extern "C" int scone_shard_select_slot(scone_handle *h, int32_t slot) {
  if (!h) return SCONE_EINVAL;
  if (slot < 0 || slot >= SCONE_SHARD_SLOTS) return scone_fail(h, SCONE_EINVAL, "scone_shard_select_slot: slot must be 0 .. 3");
  scone_shard_state *st = shard_state(h);
  if (!st) return scone_fail(h, SCONE_ENOMEM, "scone_shard_select_slot: out of memory");
  if (st->slot == slot) return SCONE_OK;
  swap_slot(st, st->parked[st->slot]);  // the active state goes to ITS parking place (now holds the active slot's state) ...
  swap_slot(st, st->parked[slot]);      // ... and the wanted slot's state becomes the active one
  st->slot = slot;
  return SCONE_OK;
}

// Receiver: rewrite the id lists of sequences [seq_begin, seq_end) of the planned batch to record numbers NOW (every row
// they reference must have been added), e.g. on the stream that waited for the records, so that the later
// scone_shard_gather_embed_range -- on the stream that reduces -- finds them done and launches the lookup alone.
extern "C" int scone_shard_gather_remap_range(scone_handle *h, int32_t seq_begin, int32_t seq_end, scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (!h->shard) return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_remap_range: call scone_shard_gather_plan first");
  scone_shard_state *st = h->shard;
  if (seq_begin < 0 || seq_end < seq_begin || seq_end > st->plan_B)
    return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_remap_range: sequences outside the planned batch");
  if (seq_end == seq_begin) return SCONE_OK;
  SCONE_ON_DEVICE(h);
  const int32_t *ell = nullptr;
  const void *scales = nullptr;
  return scone_shard_gather_remap(h, st->plan_T, seq_begin, seq_end, &ell, &scales, (hipStream_t)stream);
}

extern "C" int scone_shard_gather_plan(scone_handle *h, const int32_t *d_tok, int32_t B, int32_t T, uint64_t *h_n_records,
                                       scone_stream_t stream) {
  if (!h_n_records) return h ? scone_fail(h, SCONE_EINVAL, "scone_shard_gather_plan: bad argument") : SCONE_EINVAL;
  return scone_shard_gather_plan_chunks(h, d_tok, B, T, 1, 1, h_n_records, stream);
}

extern "C" int scone_shard_gather_pack_range(scone_handle *h, uint64_t first, uint64_t count, uint64_t pad, void *d_send_buf,
                                             scone_stream_t stream) {
  if (!h || !h->shard) return h ? scone_fail(h, SCONE_ESTATE, "scone_shard_gather_pack: call scone_shard_gather_plan first") : SCONE_EINVAL;
  scone_shard_state *st = h->shard;
  if (st->n_on_device) return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_pack: the plan kept its count on the device (use scone_shard_cols_pack_cap)");
  if (first + count > st->n_uniq) return scone_fail(h, SCONE_ERANGE, "scone_shard_gather_pack: records outside the plan");
  if (count + pad == 0) return SCONE_OK;
  if (!d_send_buf) return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_pack: null send buffer");
  SCONE_ON_DEVICE(h);
  if (count) {
    const size_t vecs = h->row_payload_bytes / 16;
    const int lpr = vecs <= 16 ? 16 : vecs <= 32 ? 32 : 64;
    const unsigned long long rec_per_block = 4ull * (64 / lpr);
    unsigned pb = (unsigned)((count + rec_per_block - 1) / rec_per_block);
    if (pb > 32768) pb = 32768;  // every row read of the pack in flight at once up to 131k waves
    hipLaunchKernelGGL(k_gather_pack, dim3(pb), dim3(256), 0, (hipStream_t)stream, st->uniq_list + first, count,
                       scone_store_of(h), (long long)h->cfg.row_begin, (const uint8_t *)h->scales, (int)h->scale_bytes_per_row,
                       scone_shard_rec_bytes(h), (uint8_t *)d_send_buf, lpr);
  }
  if (pad)
    hipLaunchKernelGGL(k_gather_mark_pad, dim3((unsigned)((pad + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (uint8_t *)d_send_buf, (unsigned long long)count, (unsigned long long)pad, scone_shard_rec_bytes(h));
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

extern "C" int scone_shard_gather_pack(scone_handle *h, void *d_send_buf, scone_stream_t stream) {
  if (!h || !h->shard) return h ? scone_fail(h, SCONE_ESTATE, "scone_shard_gather_pack: call scone_shard_gather_plan first") : SCONE_EINVAL;
  return scone_shard_gather_pack_range(h, 0, h->shard->n_uniq, 0, d_send_buf, stream);
}

// Receiver, step 1: records [record0, record0 + n) of the gathered buffer (d_records points at record record0; the
// whole buffer will hold n_total) join the row map; record0 == 0 starts a new exchange (the hash map is cleared).  The map
// is an open-addressing table sized by the exchange (8 MB for 0.45M records: it lives in L2 / the Infinity Cache); a
// direct-mapped array over all table rows (4 GB at 1e9 rows: every lookup a TLB miss + an HBM access) was built on the round-2
// review's suggestion, measured slower at C5's scale (0.927 against 0.909 ms per step, profiles/r03h) and removed in round 4.
int scone_shard_gather_add(scone_handle *h, const void *d_records, uint64_t n, uint64_t record0, uint64_t n_total,
                           hipStream_t s) {
  scone_shard_state *st = h->shard;
  const size_t sb = h->scale_bytes_per_row;
  const unsigned long long n_head = st->n_head;
  if (n_total > 0xFFFFFFF0ull || record0 + n > n_total) return scone_fail(h, SCONE_EINVAL, "scone_shard_gather_embed: bad record range");
  if (record0 == 0) {
    // (lists rewritten while the map was still EMPTY -- the first chunks of a pipelined exchange brought no records --
    // reference head rows only: nothing in them depends on the map that is (re)started here)
    if (st->added)
      for (uint8_t r : st->remapped)
        if (r) return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_add_records: lists of this plan already hold record numbers of "
                                                  "an earlier exchange (plan the batch again before a new exchange)");
    st->added = 0;
    if (sb) {
      long long cap = st->cap_recv;
      uint8_t *p = st->scales;
      int rc = grow(h, &p, &cap, (long long)(n_head + n_total), sb);
      st->scales = p, st->cap_recv = cap;
      if (rc) return rc;
      if (n_head) SCONE_HIP(h, hipMemcpyAsync(st->scales, st->head_scales, (size_t)n_head * sb, hipMemcpyDeviceToDevice, s));
    }
    long long hcap = 1024;
    while (hcap < 2 * (long long)n_total) hcap <<= 1;
    int rc = grow(h, &st->rhash, &st->cap_rhash, hcap, 1);
    if (rc) return rc;
    SCONE_HIP(h, hipMemsetAsync(st->rhash, 0, (size_t)hcap * sizeof(unsigned long long), s));
    st->rhash_cap_now = hcap;
  } else if (st->rhash_cap_now < 2 * (long long)n_total || (sb && st->cap_recv < (long long)(n_head + n_total))) {
    return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_embed: records added out of order (start with record 0)");
  }
  st->added += n;
  if (n) {
    uint8_t *sc = st->scales ? st->scales + (size_t)(n_head + record0) * sb : nullptr;
    hipLaunchKernelGGL(k_gather_index, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint8_t *)d_records,
                       (unsigned long long)n, scone_shard_rec_bytes(h), (int)h->row_payload_bytes, (int)sb, st->rhash,
                       (unsigned long long)st->rhash_cap_now - 1, sc, (unsigned long long)record0);
  }
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

// Receiver, step 2: the lists of sequences [seq0, seq1) of the planned batch are remapped to record numbers -- in place,
// so every sequence exactly ONCE per plan: the slot remembers which ones already hold record numbers (a second
// scone_shard_gather_embed_range over the same or an overlapping range reduces them as they are).
int scone_shard_gather_remap(scone_handle *h, int32_t T, int32_t seq0, int32_t seq1, const int32_t **ell, const void **scales,
                             hipStream_t s) {
  scone_shard_state *st = h->shard;
  int32_t *lists = plan_lists(st);
  const bool have_map = st->rhash && st->rhash_cap_now > 0;
  if (!have_map || !lists)
    return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_embed: plan the batch and add the records first");
  if ((size_t)seq1 > st->remapped.size()) return scone_fail(h, SCONE_ESTATE, "scone_shard_gather_embed: sequences outside the planned batch");
  const int W = SCONE_ELL_W(h->cfg.max_n), NC = h->cfg.max_n * (h->cfg.max_n + 1) / 2;
  for (int32_t a = seq0; a < seq1;) {
    if (st->remapped[a]) {
      ++a;
      continue;
    }
    int32_t b = a;
    while (b < seq1 && !st->remapped[b]) ++b;  // a run of sequences whose lists still hold row ids
    const long long nt = (long long)(b - a) * T;
    int32_t *e = lists + (long long)a * T * W;
    const unsigned blocks = (unsigned)((nt + 255) / 256 < SHARD_BLOCKS ? (nt + 255) / 256 : SHARD_BLOCKS);
    hipLaunchKernelGGL(k_gather_remap, dim3(blocks), dim3(256), 0, s, e, nt, W, NC, st->rhash,
                       (unsigned long long)st->rhash_cap_now - 1, (long long)st->n_head, h->d_status);
    for (int32_t q = a; q < b; ++q) st->remapped[q] = 1;
    a = b;
  }
  SCONE_HIP(h, hipGetLastError());
  *ell = lists + (long long)seq0 * T * W;
  *scales = st->scales;
  return SCONE_OK;
}

// ---------------------------------------------------------------- all-gather form, COLUMNS on the wire
// The records of the form above are [payload | scales | row id]: the receiver must read every record's header to build its
// row map -- 0.45M 64-bit CAS on an 8 MB table per step at C5's scale, ~36 us whatever the table's form (device-scope
// atomics retire at ~12 per ns) -- and unpack the scales.  Here a rank's contribution travels as three COLUMNS:
//   rows    [count, payload bytes]   the payloads alone, at the table's own stride (512 B for INT4 d = 1024: every row the
//                                    lookup reads in place starts on a cache-line boundary; 544-B records do not)
//   scales  [count, scale bytes]     received straight into [head scales | scales of all ranks]: nothing to unpack
//   frag    [slots] u64              the SENDER's hash fragment: row id + 1 -> position in its own contribution, built
//                                    while it packs (its own ~65k inserts, not 0.45M on every receiver)
// and the receiver has no indexing pass at all: a list entry is looked up in its owner's fragment (the owner follows from
// the id) and becomes rec_base[owner] + position.  One plan = one exchange (n_chunks = 1); 2 % more bytes on the wire.
namespace {

struct cols_owners {  // per owner: where its fragment starts (u64 slots), slots - 1, record number of its first row
  unsigned long long frag_off[64], frag_mask[64], rec_base[64];
  unsigned int row_lo[65];  // first row of every owner's range (row_lo[world] = n_rows): the owner of an id by a guess and a
  float owners_per_row;     // fix-up step instead of the 64-bit division of owner_of() -- 1.1M of them per step made the
};                          // remap 49 us where the single-map form takes 32

// d_n != null (sync-free plan): the number of rows is read from the device and clipped to `n` (the capacity the transfer was
// sized for); hdr[0] = the rows the plan claimed, hdr[1] = 1 if that exceeded the capacity (the receivers' lookups then find
// rows missing -- SCONE_ST_BAD_ID -- and the caller repeats the batch with exact sizes)
__global__ __launch_bounds__(256) void k_cols_pack(const int32_t *__restrict__ list, unsigned long long n, scone_row_store st,
                                                   long long row_begin, const uint8_t *__restrict__ scales, int scale_bytes,
                                                   uint8_t *__restrict__ rows_out, uint8_t *__restrict__ scales_out,
                                                   unsigned long long *__restrict__ frag, unsigned long long fmask, int lanes_per_rec,
                                                   const uint32_t *__restrict__ d_n, unsigned long long *__restrict__ hdr) {
  if (d_n) {
    const unsigned long long have = *d_n;
    if (hdr && blockIdx.x == 0 && threadIdx.x == 0) hdr[0] = have, hdr[1] = have > n ? 1ull : 0ull;
    n = have < n ? have : n;
  }
  const unsigned sub = (threadIdx.x & 63) / lanes_per_rec, l = (threadIdx.x & 63) % lanes_per_rec;
  const unsigned per_wave = 64 / lanes_per_rec;
  const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned long long nwaves = (unsigned long long)gridDim.x * (blockDim.x >> 6);
  for (unsigned long long p = wave * per_wave + sub; p < n; p += nwaves * per_wave) {
    const long long id = list[p];
    const unsigned long long lr = (unsigned long long)(id - row_begin);
    const uint4 *src = reinterpret_cast<const uint4 *>(st.row(lr));
    uint4 *dst = reinterpret_cast<uint4 *>(rows_out + p * (unsigned long long)st.row_bytes);
    for (unsigned v = l; v < st.row_bytes / 16; v += lanes_per_rec) dst[v] = src[v];
    for (unsigned b = l; b < (unsigned)scale_bytes / 2; b += lanes_per_rec)
      reinterpret_cast<unsigned short *>(scales_out + p * scale_bytes)[b] = reinterpret_cast<const unsigned short *>(scales + lr * scale_bytes)[b];
    if (l == 0) {  // the fragment entry of this row (ids of one plan are distinct: the CAS only resolves slot collisions)
      const unsigned long long key = (unsigned long long)id + 1ull, mine = (key << 32) | p;
      unsigned long long s = scone_hash_key(key, 0u) & fmask;
      for (unsigned long long probe = 0; probe <= fmask; ++probe) {
        if (atomicCAS(&frag[s], 0ull, mine) == 0ull) break;
        s = (s + 1ull) & fmask;
      }
    }
  }
}

// the fragment of an arbitrary id list (tools / tests stand in for the other ranks with it)
__global__ __launch_bounds__(256) void k_cols_frag(const int32_t *__restrict__ ids, unsigned long long n,
                                                   unsigned long long *__restrict__ frag, unsigned long long fmask) {
  const unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const unsigned long long key = (unsigned long long)(uint32_t)ids[p] + 1ull, mine = (key << 32) | p;
  unsigned long long s = scone_hash_key(key, 0u) & fmask;
  for (unsigned long long probe = 0; probe <= fmask; ++probe) {
    if (atomicCAS(&frag[s], 0ull, mine) == 0ull) return;
    s = (s + 1ull) & fmask;
  }
}

__global__ __launch_bounds__(256) void k_cols_remap(int32_t *__restrict__ ell, long long ntok, int W, int NC,
                                                    const unsigned long long *__restrict__ frags, const cols_owners owk, int world,
                                                    long long n_rows, long long n_head, unsigned long long n_total,
                                                    uint32_t *__restrict__ status) {
  // the per-owner tables are indexed by a per-lane owner: out of the kernel-argument segment that is a dependent global load
  // in front of every probe (the first version: 49 us against 32 us for the single-map remap); in LDS it is a few cycles
  __shared__ cols_owners ow;
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    ow.frag_off[i] = owk.frag_off[i], ow.frag_mask[i] = owk.frag_mask[i], ow.rec_base[i] = owk.rec_base[i];
    ow.row_lo[i] = owk.row_lo[i];
  }
  if (threadIdx.x == 0) ow.row_lo[64] = owk.row_lo[64], ow.owners_per_row = owk.owners_per_row;
  __syncthreads();
  const long long per = (ntok + gridDim.x - 1) / gridDim.x;
  const long long t0 = (long long)blockIdx.x * per, t1 = t0 + per < ntok ? t0 + per : ntok;
  // one probe of one id: owner, fragment, home slot (the first load of every id of a token is issued before any is looked at)
  auto home = [&](long long id, const unsigned long long *&f, unsigned long long &mask, unsigned long long &s, int &r) {
    r = (int)((float)id * ow.owners_per_row);
    r = r < world - 1 ? r : world - 1;
    while (r > 0 && (unsigned int)id < ow.row_lo[r]) --r;
    while (r < world - 1 && (unsigned int)id >= ow.row_lo[r + 1]) ++r;
    mask = ow.frag_mask[r];
    f = frags + ow.frag_off[r];
    s = scone_hash_key((unsigned long long)id + 1ull, 0u) & mask;
  };
  auto resolve = [&](long long id, const unsigned long long *f, unsigned long long mask, unsigned long long s, int r,
                     unsigned long long v) -> long long {  // v = f[s] already loaded; continues the linear probe if it must
    const unsigned long long key = (unsigned long long)id + 1ull;
    for (unsigned long long probe = 0; probe <= mask; ++probe) {
      if (v == 0ull) return -1;
      if ((v >> 32) == key) return (long long)(ow.rec_base[r] + (v & 0xFFFFFFFFull));
      s = (s + 1ull) & mask;
      v = f[s];
    }
    return -1;
  };
  for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    if (W == 8) {  // max_n <= 3: the record in registers, the home slots of all its cold ids in flight together
      const int4 a = reinterpret_cast<const int4 *>(ell + t * 8)[0], b = reinterpret_cast<const int4 *>(ell + t * 8)[1];
      int32_t r8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const int kown = r8[6] & 0xFF;
      const unsigned long long *f[6];
      unsigned long long mask[6], s[6], v[6];
      int r[6];
      bool cold[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const long long id = r8[j];
        cold[j] = j < kown && j < NC && id >= n_head;
        v[j] = 0ull;
        if (cold[j] && id < n_rows) {
          home(id, f[j], mask[j], s[j], r[j]);
          v[j] = f[j][s[j]];
        }
      }
      bool dirty = false;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (!cold[j]) continue;
        const long long id = r8[j];
        long long slot = id < n_rows ? resolve(id, f[j], mask[j], s[j], r[j], v[j]) : -1;
        if (slot < 0 || (unsigned long long)slot >= n_total) {  // the row did not arrive / a corrupt fragment: report, stay in bounds
          atomicOr(status, SCONE_ST_BAD_ID);
          slot = 0;
        }
        r8[j] = (int32_t)(n_head + slot);
        dirty = true;
      }
      if (dirty) {
        reinterpret_cast<int4 *>(ell + t * 8)[0] = make_int4(r8[0], r8[1], r8[2], r8[3]);
        reinterpret_cast<int4 *>(ell + t * 8)[1] = make_int4(r8[4], r8[5], r8[6], r8[7]);
      }
      continue;
    }
    const int kown = ell[t * W + W - 2] & 0xFF;
    for (int j = 0; j < NC; ++j) {
      if (j >= kown) break;
      const long long id = ell[t * W + j];
      if (id < n_head) continue;  // a head row: its id is its row number in the lookup's row store
      long long slot = -1;
      if (id < n_rows) {
        const unsigned long long *f;
        unsigned long long mask, s;
        int r;
        home(id, f, mask, s, r);
        slot = resolve(id, f, mask, s, r, f[s]);
      }
      if (slot < 0 || (unsigned long long)slot >= n_total) {
        atomicOr(status, SCONE_ST_BAD_ID);
        slot = 0;
      }
      ell[t * W + j] = (int32_t)(n_head + slot);
    }
  }
}

}  // namespace

extern "C" int scone_shard_cols_frag_slots(uint64_t count, uint64_t *slots) {
  if (!slots) return SCONE_EINVAL;
  uint64_t s = 64;
  // load <= 0.25: linear probing at 0.5 cost the receiver's remap 43 us instead of 32 and the sender's pack 30 instead of
  // 23 (profiles/r03u/columns_fragment_load_*: clusters of occupied slots mean dependent probes); twice the slots are 7 MB more on the wire (3 %)
  while (s < 4 * count) s <<= 1;
  *slots = s;
  return SCONE_OK;
}

extern "C" int scone_shard_cols_pack(scone_handle *h, uint64_t first, uint64_t count, void *d_rows_out, void *d_scales_out,
                                     void *d_frag_out, uint64_t frag_slots, scone_stream_t stream) {
  if (!h || !h->shard) return h ? scone_fail(h, SCONE_ESTATE, "scone_shard_cols_pack: call scone_shard_gather_plan first") : SCONE_EINVAL;
  scone_shard_state *st = h->shard;
  if (st->n_on_device) return scone_fail(h, SCONE_ESTATE, "scone_shard_cols_pack: the plan kept its count on the device (use scone_shard_cols_pack_cap)");
  if (first + count > st->n_uniq) return scone_fail(h, SCONE_ERANGE, "scone_shard_cols_pack: records outside the plan");
  uint64_t want = 0;
  scone_shard_cols_frag_slots(count, &want);
  if (!d_frag_out || frag_slots < want || (frag_slots & (frag_slots - 1)))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_pack: fragment needs a power of two >= scone_shard_cols_frag_slots(count) slots");
  if (count && (!d_rows_out || (h->scale_bytes_per_row && !d_scales_out)))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_pack: null output");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  SCONE_HIP(h, hipMemsetAsync(d_frag_out, 0, (size_t)frag_slots * sizeof(unsigned long long), s));
  if (count) {
    const size_t vecs = h->row_payload_bytes / 16;
    const int lpr = vecs <= 16 ? 16 : vecs <= 32 ? 32 : 64;
    const unsigned long long rec_per_block = 4ull * (64 / lpr);
    unsigned pb = (unsigned)((count + rec_per_block - 1) / rec_per_block);
    if (pb > 32768) pb = 32768;
    hipLaunchKernelGGL(k_cols_pack, dim3(pb), dim3(256), 0, s, st->uniq_list + first, (unsigned long long)count, scone_store_of(h),
                       (long long)h->cfg.row_begin, (const uint8_t *)h->scales, (int)h->scale_bytes_per_row, (uint8_t *)d_rows_out,
                       (uint8_t *)d_scales_out, (unsigned long long *)d_frag_out, (unsigned long long)frag_slots - 1, lpr,
                       (const uint32_t *)nullptr, (unsigned long long *)nullptr);
  }
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

// The pack of a sync-free plan: up to cap_rows of the claimed rows (the capacity both ends sized the transfer for), the count
// read on the device; d_header_out [2] u64 = {rows claimed, 1 if more than cap_rows}.
extern "C" int scone_shard_cols_pack_cap(scone_handle *h, uint64_t cap_rows, void *d_rows_out, void *d_scales_out, void *d_frag_out,
                                         uint64_t frag_slots, void *d_header_out, scone_stream_t stream) {
  if (!h || !h->shard) return h ? scone_fail(h, SCONE_ESTATE, "scone_shard_cols_pack_cap: call scone_shard_gather_plan_async first") : SCONE_EINVAL;
  scone_shard_state *st = h->shard;
  if (!st->n_on_device) return scone_fail(h, SCONE_ESTATE, "scone_shard_cols_pack_cap: the plan's count is on the host (use scone_shard_cols_pack)");
  uint64_t want = 0;
  scone_shard_cols_frag_slots(cap_rows, &want);
  if (!d_frag_out || !d_header_out || frag_slots < want || (frag_slots & (frag_slots - 1)))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_pack_cap: fragment needs a power of two >= scone_shard_cols_frag_slots(cap_rows) slots");
  if (cap_rows && (!d_rows_out || (h->scale_bytes_per_row && !d_scales_out)))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_pack_cap: null output");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  SCONE_HIP(h, hipMemsetAsync(d_frag_out, 0, (size_t)frag_slots * sizeof(unsigned long long), s));
  const size_t vecs = h->row_payload_bytes / 16;
  const int lpr = vecs <= 16 ? 16 : vecs <= 32 ? 32 : 64;
  const unsigned long long rec_per_block = 4ull * (64 / lpr);
  unsigned pb = (unsigned)((cap_rows + rec_per_block - 1) / rec_per_block);
  if (pb > 32768) pb = 32768;
  if (pb < 1) pb = 1;
  // (the list holds every claimed row: cap_uniq >= the count, so clipping to cap_rows is the only bound the kernel needs)
  hipLaunchKernelGGL(k_cols_pack, dim3(pb), dim3(256), 0, s, st->uniq_list, (unsigned long long)cap_rows, scone_store_of(h),
                     (long long)h->cfg.row_begin, (const uint8_t *)h->scales, (int)h->scale_bytes_per_row, (uint8_t *)d_rows_out,
                     (uint8_t *)d_scales_out, (unsigned long long *)d_frag_out, (unsigned long long)frag_slots - 1, lpr,
                     (const uint32_t *)st->chunk_ends, (unsigned long long *)d_header_out);
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

extern "C" int scone_shard_cols_build_frag(scone_handle *h, const int32_t *d_ids, uint64_t count, void *d_frag_out, uint64_t frag_slots,
                                           scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  uint64_t want = 0;
  scone_shard_cols_frag_slots(count, &want);
  if (!d_frag_out || frag_slots < want || (frag_slots & (frag_slots - 1)) || (count && !d_ids))
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_build_frag: bad argument");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  SCONE_HIP(h, hipMemsetAsync(d_frag_out, 0, (size_t)frag_slots * sizeof(unsigned long long), s));
  if (count)
    hipLaunchKernelGGL(k_cols_frag, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, d_ids, (unsigned long long)count,
                       (unsigned long long *)d_frag_out, (unsigned long long)frag_slots - 1);
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

// every change of the replicated head bumps this: a caller that copied the head's scales into its own buffer
// (scone_shard_head_scales) copies them again when the number moved
extern "C" int scone_shard_head_version(scone_handle *h, uint64_t *version) {
  if (!h || !version) return SCONE_EINVAL;
  *version = h->shard ? h->shard->head_version : 0;
  return SCONE_OK;
}

// the replicated head's scales, for the front of a [head scales | received scales] buffer the caller owns
extern "C" int scone_shard_head_scales(scone_handle *h, void *d_out, scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (!h->shard || !h->shard->n_head || !h->scale_bytes_per_row) return SCONE_OK;
  if (!d_out) return scone_fail(h, SCONE_EINVAL, "scone_shard_head_scales: null output");
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipMemcpyAsync(d_out, h->shard->head_scales, (size_t)h->shard->n_head * h->scale_bytes_per_row, hipMemcpyDeviceToDevice,
                              (hipStream_t)stream));
  return SCONE_OK;
}

// Receiver of the columns exchange: remap (once per plan) through the owners' fragments; returns the lists, the head at the
// payload stride and the number of head rows for the lookup launch in scone_gather.hip.
int scone_shard_cols_remap(scone_handle *h, int32_t T, int32_t seq0, int32_t seq1, const void *d_frags, uint64_t frag_slots_total,
                           const uint64_t *h_frag_off, const uint64_t *h_frag_slots, const uint64_t *h_rec_base,
                           const uint64_t *h_row_lo, int32_t world, uint64_t n_total,
                           const int32_t **ell, const uint8_t **head_p, unsigned long long *n_head_out, hipStream_t s) {
  scone_shard_state *st = h->shard;
  int32_t *lists = plan_lists(st);
  if (!lists) return scone_fail(h, SCONE_ESTATE, "scone_shard_cols_embed: plan the batch first");
  if ((size_t)seq1 > st->remapped.size()) return scone_fail(h, SCONE_ESTATE, "scone_shard_cols_embed: sequences outside the planned batch");
  if (world < 1 || world > 64) return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: world <= 64");
  // The owner of a row id follows from the owners' row ranges, which the CALLER states (h_row_lo[r] = first row of rank r,
  // h_row_lo[world] = n_rows): handles may be created with any [row_begin, row_end), and nothing here assumes the floor
  // partition of distributed.shard_range (NULL selects it).  Every fragment must lie inside the buffer it was received into.
  if (h->cfg.n_rows > 0xFFFFFFFFull) return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: tables of at most 2^32 rows");
  cols_owners ow = {};
  for (int r = 0; r < world; ++r) {
    if (h_frag_slots[r] == 0 || (h_frag_slots[r] & (h_frag_slots[r] - 1)))
      return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: fragment sizes must be powers of two");
    if (h_frag_off[r] + h_frag_slots[r] > frag_slots_total)
      return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: a fragment lies outside d_frags (frag_slots_total)");
    ow.frag_off[r] = h_frag_off[r], ow.frag_mask[r] = h_frag_slots[r] - 1, ow.rec_base[r] = h_rec_base[r];
    const unsigned long long lo = h_row_lo ? h_row_lo[r] : ((unsigned long long)r * h->cfg.n_rows) / (unsigned long long)world;
    if (lo > h->cfg.n_rows || (r > 0 && lo < ow.row_lo[r - 1]) || (r == 0 && lo != 0))
      return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: h_row_lo must ascend from 0 to n_rows");
    ow.row_lo[r] = (unsigned int)lo;
  }
  if (h_row_lo && h_row_lo[world] != h->cfg.n_rows)
    return scone_fail(h, SCONE_EINVAL, "scone_shard_cols_embed: h_row_lo[world] must be n_rows");
  ow.row_lo[world] = (unsigned int)h->cfg.n_rows;
  ow.owners_per_row = (float)world / (float)(h->cfg.n_rows ? h->cfg.n_rows : 1);  // a first guess; the fix-up steps make it exact
  const size_t pb = h->row_payload_bytes;
  if (st->n_head) {  // the head at the payload stride: rebuilt once per head change, on this stream; other streams wait for it
    if (!st->head_p_ready) SCONE_HIP(h, hipEventCreateWithFlags(&st->head_p_ready, hipEventDisableTiming));
    if (st->head_p_stale || !st->head_rows_p) {
      if (!st->head_rows_p) SCONE_HIP(h, hipMalloc(&st->head_rows_p, (size_t)st->n_head * pb));
      SCONE_HIP(h, hipMemcpy2DAsync(st->head_rows_p, pb, st->head_rows, (size_t)scone_shard_rec_bytes(h), pb, (size_t)st->n_head,
                                    hipMemcpyDeviceToDevice, s));
      SCONE_HIP(h, hipEventRecord(st->head_p_ready, s));
      st->head_p_stale = false;
    } else {
      SCONE_HIP(h, hipStreamWaitEvent(s, st->head_p_ready, 0));
    }
  }
  const int W = SCONE_ELL_W(h->cfg.max_n), NC = h->cfg.max_n * (h->cfg.max_n + 1) / 2;
  for (int32_t a = seq0; a < seq1;) {
    if (st->remapped[a]) {
      ++a;
      continue;
    }
    int32_t b = a;
    while (b < seq1 && !st->remapped[b]) ++b;
    const long long nt = (long long)(b - a) * T;
    const unsigned blocks = (unsigned)((nt + 255) / 256 < SHARD_BLOCKS ? (nt + 255) / 256 : SHARD_BLOCKS);
    hipLaunchKernelGGL(k_cols_remap, dim3(blocks), dim3(256), 0, s, lists + (long long)a * T * W, nt, W, NC,
                       (const unsigned long long *)d_frags, ow, (int)world, (long long)h->cfg.n_rows, (long long)st->n_head,
                       (unsigned long long)n_total, h->d_status);
    for (int32_t q = a; q < b; ++q) st->remapped[q] = 1;
    a = b;
  }
  SCONE_HIP(h, hipGetLastError());
  *ell = lists + (long long)seq0 * T * W;
  *head_p = st->head_rows_p;
  *n_head_out = st->n_head;
  return SCONE_OK;
}

int scone_shard_plan_shape(const scone_handle *h, int32_t *B, int32_t *T) {
  if (!h->shard) return SCONE_ESTATE;
  *B = h->shard->plan_B, *T = h->shard->plan_T;
  return SCONE_OK;
}
