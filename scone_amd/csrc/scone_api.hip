// Handle lifecycle, error reporting and workspaces of libscone_hip.so.
#include "scone_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>

// The message of a handle's last failure; written under a lock so that concurrent lookups cannot corrupt the string
// (scone_last_error returns a pointer into it: read it from the thread whose call failed, before its next call).
int scone_fail(scone_handle *h, int code, const char *what) {
  if (h) {
    std::lock_guard<std::mutex> g(h->err_mu);
    h->err = what ? what : "";
  }
  return code;
}

int scone_hip_fail(scone_handle *h, hipError_t e, const char *what) {
  if (h) {
    std::lock_guard<std::mutex> g(h->err_mu);
    h->err = std::string(what ? what : "HIP call") + ": " + hipGetErrorString(e);
  }
  (void)hipGetLastError();  // clear the sticky error
  return e == hipErrorOutOfMemory ? SCONE_ENOMEM : SCONE_EHIP;
}

scone_row_store scone_store_of(const scone_handle *h) {
  scone_row_store st;
  st.hot = reinterpret_cast<uint8_t *>(h->rows);
  st.cold = reinterpret_cast<uint8_t *>(h->rows_host);
  st.n_hot = h->hot_local;
  st.row_bytes = (unsigned int)h->row_payload_bytes;
  return st;
}

// At most SCONE_MAX_WS workspaces per handle: a process that keeps creating streams re-uses the least recently used
// idle one (its buffers are freed -- hipFree synchronises the device, so nothing still reads them) instead of growing.
#define SCONE_MAX_WS 16
static scone_ws *scone_ws_acquire_once(scone_handle *h, hipStream_t s);
scone_ws *scone_ws_acquire(scone_handle *h, hipStream_t s) {
  for (;;) {
    scone_ws *w = scone_ws_acquire_once(h, s);
    // between finding the stream's workspace and locking it another thread may have recycled it for ITS stream (all
    // other workspaces busy): then it is no longer ours -- look again
    if (!w || w->stream == s) return w;
    w->mu.unlock();
  }
}

static scone_ws *scone_ws_acquire_once(scone_handle *h, hipStream_t s) {
  scone_ws *w = nullptr;
  {
    std::lock_guard<std::mutex> g(h->ws_mu);
    h->ws_clock += 1;
    for (scone_ws *c : h->ws)
      if (c->stream == s) w = c;
    if (!w && h->ws.size() >= SCONE_MAX_WS) {
      scone_ws *victim = nullptr;
      for (scone_ws *c : h->ws)
        if (c->stream != nullptr && (!victim || c->last_use < victim->last_use) && c->mu.try_lock()) {
          if (victim) victim->mu.unlock();
          victim = c;
        }
      if (victim) {  // locked by the try_lock above
        if (victim->d_hits) (void)hipFree(victim->d_hits);
        if (victim->d_ell) (void)hipFree(victim->d_ell);
        if (victim->d_block_sums) (void)hipFree(victim->d_block_sums);
        victim->d_hits = victim->d_ell = victim->d_block_sums = nullptr;
        victim->hits_cap_tokens = victim->ell_cap_tokens = victim->block_sums_cap = 0;
        victim->stream = s;
        victim->last_use = h->ws_clock;
        return victim;  // still locked: the caller's
      }
    }
    if (!w) {
      w = new (std::nothrow) scone_ws();
      if (!w) return nullptr;
      w->stream = s;
      h->ws.push_back(w);
    }
    w->last_use = h->ws_clock;
  }
  w->mu.lock();
  return w;
}

// Growing frees the old buffer: hipFree synchronises the device, so no kernel still reads it.
int scone_ensure_hits(scone_handle *h, scone_ws *w, int64_t ntok) {
  if (ntok < h->reserve_tokens) ntok = h->reserve_tokens;
  if (ntok <= w->hits_cap_tokens) return SCONE_OK;
  if (w->d_hits) SCONE_HIP(h, hipFree(w->d_hits));
  w->d_hits = nullptr;
  w->hits_cap_tokens = 0;
  SCONE_HIP(h, hipMalloc(&w->d_hits, (size_t)ntok * h->cfg.max_n * sizeof(int32_t)));
  w->hits_cap_tokens = ntok;
  return SCONE_OK;
}

int scone_ensure_ell(scone_handle *h, scone_ws *w, int64_t ntok) {
  if (ntok < h->reserve_tokens) ntok = h->reserve_tokens;
  if (ntok <= w->ell_cap_tokens) return SCONE_OK;
  if (w->d_ell) SCONE_HIP(h, hipFree(w->d_ell));
  w->d_ell = nullptr;
  w->ell_cap_tokens = 0;
  SCONE_HIP(h, hipMalloc(&w->d_ell, (size_t)ntok * SCONE_ELL_W(h->cfg.max_n) * sizeof(int32_t)));
  w->ell_cap_tokens = ntok;
  return SCONE_OK;
}

extern "C" int scone_abi_version(void) { return SCONE_ABI_VERSION; }

extern "C" const char *scone_strerror(int code) {
  switch (code) {
    case SCONE_OK: return "ok";
    case SCONE_ESTATE: return "invalid state or call order";
    case SCONE_EHIP: return "HIP runtime error";
    case SCONE_ENOMEM: return "out of memory or index full";
    case SCONE_ENODEV: return "no usable GPU";
    case SCONE_EINVAL: return "invalid argument";
    case SCONE_ERANGE: return "value out of range";
    default: return "unknown error";
  }
}

static thread_local std::string g_create_err;

extern "C" const char *scone_last_error(const scone_handle *h) {
  return h ? h->err.c_str() : g_create_err.c_str();
}

static bool payload_geometry(const scone_cfg &c, size_t *payload, size_t *scale_bytes) {
  const size_t d = (size_t)c.dim;
  switch (c.table_fmt) {
    case SCONE_FMT_F32:
      if (d % 4) return false;
      *payload = 4 * d, *scale_bytes = 0;
      return true;
    case SCONE_FMT_F16:
      if (d % 8) return false;
      *payload = 2 * d, *scale_bytes = 0;
      return true;
    case SCONE_FMT_I8:
      if (d % 16) return false;
      *payload = d, *scale_bytes = 2;
      return true;
    case SCONE_FMT_I4:
      if (d % SCONE_I4_GROUP) return false;
      *payload = d / 2, *scale_bytes = 2 * (d / SCONE_I4_GROUP);
      return true;
    default: return false;
  }
}

// memset of a (possibly tens of GB) pinned region, split over a few host threads
static void zero_host(void *p, size_t bytes) {
  const size_t chunk = (size_t)1 << 30;
  if (bytes <= chunk) {
    memset(p, 0, bytes);
    return;
  }
  unsigned nthr = std::thread::hardware_concurrency();
  if (nthr > 16) nthr = 16;
  if (nthr < 1) nthr = 1;
  std::vector<std::thread> pool;
  const size_t per = (bytes + nthr - 1) / nthr;
  for (unsigned t = 0; t < nthr; ++t) {
    const size_t a = (size_t)t * per, b = a + per < bytes ? a + per : bytes;
    if (a < b) pool.emplace_back([=] { memset(static_cast<char *>(p) + a, 0, b - a); });
  }
  for (auto &th : pool) th.join();
}

extern "C" int scone_create(const scone_cfg *cfg, scone_handle **out) {
  if (!cfg || !out) return SCONE_EINVAL;
  *out = nullptr;
  if (cfg->struct_size != sizeof(scone_cfg)) {
    g_create_err = "scone_create: struct_size mismatch (ABI)";
    return SCONE_EINVAL;
  }
  if (cfg->stage_tokens && (cfg->placement != SCONE_PLACE_PINNED_HOST || cfg->row_begin != 0 ||
                            (cfg->row_end != 0 && cfg->row_end != cfg->n_rows))) {
    g_create_err = "scone_create: stage_tokens needs SCONE_PLACE_PINNED_HOST and an unsharded table";
    return SCONE_EINVAL;
  }
  if (cfg->lookup_mode > SCONE_MODE_LONGEST_SUFFIX) {
    g_create_err = "scone_create: unknown lookup_mode";
    return SCONE_EINVAL;
  }
  if (cfg->max_n < 1 || cfg->max_n > SCONE_MAX_N || cfg->dim < 0) {
    g_create_err = "scone_create: max_n must be 1..4 and dim >= 0";
    return SCONE_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    g_create_err = "scone_create: no HIP device visible";
    return SCONE_ENODEV;
  }
  if (cfg->device < 0 || cfg->device >= ndev) {
    g_create_err = "scone_create: device ordinal out of range";
    return SCONE_EINVAL;
  }
  scone_handle *h = new (std::nothrow) scone_handle();
  if (!h) {
    g_create_err = "scone_create: out of host memory";
    return SCONE_ENOMEM;
  }
  h->cfg = *cfg;
  h->device = cfg->device;
  if (h->cfg.row_end == 0) h->cfg.row_end = h->cfg.n_rows;
  if (h->cfg.row_begin > h->cfg.row_end || h->cfg.row_end > h->cfg.n_rows) {
    g_create_err = "scone_create: need row_begin <= row_end <= n_rows";
    delete h;
    return SCONE_EINVAL;
  }
  h->local_rows = h->cfg.row_end - h->cfg.row_begin;
  h->slots = nullptr, h->d_counters = nullptr, h->d_status = nullptr, h->d_uni = nullptr, h->d_bloom = nullptr, h->bloom_mask = 0;
  h->rows = nullptr, h->rows_host = nullptr, h->scales = nullptr, h->hot_local = 0;
  h->d_zero_row = nullptr, h->reserve_tokens = 0, h->ws_clock = 0;
  h->row_payload_bytes = 0, h->scale_bytes_per_row = 0;
  h->stage = nullptr, h->shard = nullptr;
  h->cu_reserve = 0, h->lookup_stream = nullptr, h->lookup_in = nullptr, h->lookup_out = nullptr;
  h->prof_on = false, h->prof_ev = nullptr, h->prof_head = 0, h->prof_n = 0, h->prof_ms = 0.0;

  uint64_t cap = cfg->index_capacity;
  if (cap == 0) {
    cap = 64;
    while (cap < 2 * cfg->n_rows) cap <<= 1;
  }
  if ((cap & (cap - 1)) || cap < SCONE_BUCKET) {
    g_create_err = "scone_create: index_capacity must be a power of two >= 4";
    delete h;
    return SCONE_EINVAL;
  }
  h->cap = cap;

  int rc = SCONE_OK;
  hipError_t e;
  scone_device_guard dev_guard__(h->device);  // the caller's current device is restored on return
#define CREATE_HIP(call)                         \
  do {                                           \
    e = (call);                                  \
    if (e != hipSuccess) {                       \
      rc = scone_hip_fail(h, e, #call);          \
      goto fail;                                 \
    }                                            \
  } while (0)
  CREATE_HIP(dev_guard__.err);
  {
    int cus = 0;
    CREATE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    h->n_cus = cus > 0 ? cus : 256;
  }
  {  // one-launch limit: default = the measured crossover; the environment can move it (tests run both forms, A/B runs)
    h->fused_max_tokens = 32768;
    const char *ev = getenv("SCONE_FUSED_MAX_TOKENS");
    if (ev && *ev) h->fused_max_tokens = atoll(ev);
    h->match_tile = 0;
    ev = getenv("SCONE_MATCH_TILE");
    if (ev && *ev) h->match_tile = atoll(ev);
  }
  CREATE_HIP(hipMalloc(&h->slots, cap * sizeof(scone_slot)));
  CREATE_HIP(hipMemset(h->slots, 0, cap * sizeof(scone_slot)));
  CREATE_HIP(hipMalloc(&h->d_counters, 2 * sizeof(unsigned long long)));
  CREATE_HIP(hipMemset(h->d_counters, 0, 2 * sizeof(unsigned long long)));
  CREATE_HIP(hipMalloc(&h->d_status, sizeof(uint32_t)));
  CREATE_HIP(hipMemset(h->d_status, 0, sizeof(uint32_t)));
  CREATE_HIP(hipMalloc(&h->d_uni, (size_t)SCONE_UNI_CAP * sizeof(int32_t)));
  CREATE_HIP(hipMemset(h->d_uni, 0xFF, (size_t)SCONE_UNI_CAP * sizeof(int32_t)));
  {  // presence bitmap: 8 bits of room per slot (16 per key at load 0.5), capped at 128 MB
    unsigned long long bits = cap * 8ull;
    if (bits > (1ull << 30)) bits = 1ull << 30;
    if (bits < 1024) bits = 1024;
    CREATE_HIP(hipMalloc(&h->d_bloom, bits / 8));
    CREATE_HIP(hipMemset(h->d_bloom, 0, bits / 8));
    h->bloom_mask = bits - 1;
  }
  if (cfg->dim > 0) {  // a row of zeros: stands in for wte / wpe when the caller passes none
    CREATE_HIP(hipMalloc(&h->d_zero_row, (size_t)cfg->dim * 4 + 64));  // + room for a record's scales / header
    CREATE_HIP(hipMemset(h->d_zero_row, 0, (size_t)cfg->dim * 4 + 64));
  }
  if (cfg->dim > 0) {
    if (!payload_geometry(h->cfg, &h->row_payload_bytes, &h->scale_bytes_per_row)) {
      h->err = "scone_create: dim not compatible with table_fmt (F32 %4, F16 %8, I8 %16, I4 %128)";
      rc = SCONE_EINVAL;
      goto fail;
    }
    size_t scales_bytes = (size_t)h->local_rows * h->scale_bytes_per_row;
    if (cfg->placement == SCONE_PLACE_HBM) {
      h->hot_local = h->local_rows;
    } else if (cfg->placement == SCONE_PLACE_PINNED_HOST) {
      // global rows [0, hot_rows) stay in HBM; the rest of this shard lives in host DRAM mapped
      // into the GPU's address space (scales always stay in HBM)
      uint64_t hot_end = cfg->hot_rows < h->cfg.row_end ? cfg->hot_rows : h->cfg.row_end;
      h->hot_local = hot_end > h->cfg.row_begin ? hot_end - h->cfg.row_begin : 0;
    } else {
      h->err = "scone_create: unknown placement";
      rc = SCONE_EINVAL;
      goto fail;
    }
    {
      // Rows and scales start as zeros: the index may name f-grams whose rows were never stored (the reference raises
      // KeyError there, embedding_cache.py:139; EmbeddingCache.to_device / embed_tokens(check=True) report them) --
      // a lookup must never sum uninitialised memory.  INT4 zero nibbles with a zero scale dequantise to -0.0 * 8 = 0.
      size_t hot_bytes = (size_t)h->hot_local * h->row_payload_bytes;
      size_t cold_bytes = (size_t)(h->local_rows - h->hot_local) * h->row_payload_bytes;
      CREATE_HIP(hipMalloc(&h->rows, hot_bytes ? hot_bytes : 16));
      CREATE_HIP(hipMemsetAsync(h->rows, 0, hot_bytes ? hot_bytes : 16, nullptr));
      if (cold_bytes) {
        CREATE_HIP(hipHostMalloc(&h->rows_host, cold_bytes, hipHostMallocMapped | hipHostMallocPortable));
        zero_host(h->rows_host, cold_bytes);
      }
    }
    if (scales_bytes) {
      CREATE_HIP(hipMalloc(&h->scales, scales_bytes + 4));  // +4: scales are also read as dword pairs
      CREATE_HIP(hipMemsetAsync(h->scales, 0, scales_bytes + 4, nullptr));
    }
    CREATE_HIP(hipStreamSynchronize(nullptr));
  }
#undef CREATE_HIP
  *out = h;
  return SCONE_OK;
fail:
  g_create_err = h->err;
  scone_destroy(h);
  return rc;
}

// CU-masked streams are RETIRED, never destroyed: scone_lookup_stream hands them to the caller, and frameworks keep what they
// are handed -- PyTorch's caching allocator remembers every stream a block was used on (Tensor.record_stream) and records an
// event on each of them when the block is freed, which may be long after this handle dropped its reserve or was destroyed
// (found as a segmentation fault in `del tok` at the end of the 2-rank bench rehearsal).  A retired stream is re-used by the
// next scone_set_cu_reserve with the same device and reserve, so toggling a reserve does not accumulate streams.
namespace {
std::mutex g_retired_mu;
std::map<std::pair<int, int>, std::vector<hipStream_t>> g_retired;  // (device, compute units enabled) -> idle masked streams

void retire_masked_stream(int device, int enabled, hipStream_t s) {
  std::lock_guard<std::mutex> g(g_retired_mu);
  g_retired[{device, enabled}].push_back(s);
}

int masked_enabled_cus(const scone_handle *h, int n_reserved) {
  if (const char *ev = getenv("SCONE_CU_RESERVE_DEBUG_FULL_MASK"))  // measurement aid: the stream hop without any masking
    if (*ev == '1') return h->n_cus;
  return h->n_cus - n_reserved;
}

hipStream_t take_retired_stream(int device, int enabled) {
  std::lock_guard<std::mutex> g(g_retired_mu);
  auto it = g_retired.find({device, enabled});
  if (it == g_retired.end() || it->second.empty()) return nullptr;
  hipStream_t s = it->second.back();
  it->second.pop_back();
  return s;
}
}  // namespace

extern "C" void scone_destroy(scone_handle *h) {
  if (!h) return;
  scone_device_guard dev_guard__(h->device);  // e.g. a handle garbage-collected while another device is current
  if (h->slots) (void)hipFree(h->slots);
  if (h->d_counters) (void)hipFree(h->d_counters);
  if (h->d_status) (void)hipFree(h->d_status);
  if (h->d_uni) (void)hipFree(h->d_uni);
  if (h->d_bloom) (void)hipFree(h->d_bloom);
  if (h->rows) (void)hipFree(h->rows);
  if (h->rows_host) (void)hipHostFree(h->rows_host);
  if (h->scales) (void)hipFree(h->scales);
  for (scone_ws *w : h->ws) {
    if (w->d_hits) (void)hipFree(w->d_hits);
    if (w->d_ell) (void)hipFree(w->d_ell);
    if (w->d_block_sums) (void)hipFree(w->d_block_sums);
    if (w->d_total) (void)hipFree(w->d_total);
    delete w;
  }
  h->ws.clear();
  if (h->d_zero_row) (void)hipFree(h->d_zero_row);
  scone_stage_destroy(h);
  scone_shard_destroy(h);
  if (h->lookup_stream) {  // retired, not destroyed (see scone_set_cu_reserve)
    (void)hipStreamSynchronize(h->lookup_stream);
    retire_masked_stream(h->device, masked_enabled_cus(h, h->cu_reserve), h->lookup_stream);
  }
  if (h->lookup_in) (void)hipEventDestroy(h->lookup_in);
  if (h->lookup_out) (void)hipEventDestroy(h->lookup_out);
  if (h->prof_ev) {
    for (int i = 0; i < 2 * SCONE_PROF_RING; ++i)
      if (h->prof_ev[i]) (void)hipEventDestroy(h->prof_ev[i]);
    delete[] h->prof_ev;
  }
  delete h;
}

extern "C" int scone_status(scone_handle *h, uint32_t *bits, scone_stream_t stream) {
  if (!h || !bits) return SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  SCONE_HIP(h, hipMemcpyAsync(bits, h->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  SCONE_HIP(h, hipMemsetAsync(h->d_status, 0, sizeof(uint32_t), s));
  SCONE_HIP(h, hipStreamSynchronize(s));
  return SCONE_OK;
}

extern "C" int scone_reserve(scone_handle *h, int64_t max_tokens) {
  if (!h || max_tokens < 0) return SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  // every workspace (existing ones now, later ones when they are created) holds at least max_tokens; the default
  // stream's is created here so that a caller on that stream allocates nothing inside its timed region
  {
    std::lock_guard<std::mutex> g(h->ws_mu);
    if (max_tokens > h->reserve_tokens) h->reserve_tokens = max_tokens;
  }
  scone_ws *w0 = scone_ws_acquire(h, nullptr);
  if (!w0) return scone_fail(h, SCONE_ENOMEM, "scone_reserve: out of memory");
  scone_ws_release(w0);
  std::vector<scone_ws *> all;
  {
    std::lock_guard<std::mutex> g(h->ws_mu);
    all = h->ws;
  }
  for (scone_ws *w : all) {
    w->mu.lock();
    scone_ws_lock lk(w);
    int rc = h->cfg.dim > 0 && h->cfg.dim % 8 == 0 ? scone_ensure_ell(h, w, max_tokens) : scone_ensure_hits(h, w, max_tokens);
    if (rc) return rc;
  }
  return SCONE_OK;
}

// ---------------------------------------------------------------- CU reserve
// A resident lookup grid holds every wave slot of the chip for the length of the kernel (three full residency rounds of
// one-wave-per-SIMD workgroups), and whatever another stream launches meanwhile -- the send / recv channels of an RCCL
// collective, a copy kernel -- gets a workgroup in only where a lookup workgroup retires AND the freed registers suffice.
// With a reserve of R compute units the lookup runs on a stream whose CU mask excludes them: the transport kernels of the
// sharded step always find R idle CUs, at the price of R / n_cus of the lookup's issue slots (it is memory-bound: the
// curve is in DESIGN.md section 6).  Mask bits are dealt round-robin over the 8 XCDs by the driver (bit i -> XCD i % 8),
// so clearing the TOP R bits (R a multiple of 8) takes R / 8 CUs from every XCD and leaves the L2s evenly loaded.
extern "C" int scone_set_cu_reserve(scone_handle *h, int32_t n_reserved) {
  if (!h) return SCONE_EINVAL;
  if (n_reserved < 0 || n_reserved >= h->n_cus || n_reserved % 8)
    return scone_fail(h, SCONE_EINVAL, "scone_set_cu_reserve: need 0 <= n_reserved < compute units, a multiple of 8 (one per XCD)");
  SCONE_ON_DEVICE(h);
  std::lock_guard<std::mutex> g(h->lookup_mu);
  if (n_reserved == h->cu_reserve) return SCONE_OK;
  if (h->lookup_stream) {
    SCONE_HIP(h, hipStreamSynchronize(h->lookup_stream));
    retire_masked_stream(h->device, masked_enabled_cus(h, h->cu_reserve), h->lookup_stream);
    h->lookup_stream = nullptr;
  }
  h->cu_reserve = 0;
  if (n_reserved == 0) return SCONE_OK;
  uint32_t mask[32] = {};
  const int words = (h->n_cus + 31) / 32;
  if (words > 32) return scone_fail(h, SCONE_EINVAL, "scone_set_cu_reserve: more than 1024 compute units");
  const int enabled = masked_enabled_cus(h, n_reserved);
  h->lookup_stream = take_retired_stream(h->device, enabled);
  if (!h->lookup_stream) {
    for (int i = 0; i < enabled; ++i) mask[i >> 5] |= 1u << (i & 31);
    SCONE_HIP(h, hipExtStreamCreateWithCUMask(&h->lookup_stream, (uint32_t)words, mask));
  }
  if (!h->lookup_in) SCONE_HIP(h, hipEventCreateWithFlags(&h->lookup_in, hipEventDisableTiming));
  if (!h->lookup_out) SCONE_HIP(h, hipEventCreateWithFlags(&h->lookup_out, hipEventDisableTiming));
  h->cu_reserve = n_reserved;
  return SCONE_OK;
}

extern "C" int scone_get_cu_reserve(scone_handle *h, int32_t *n_reserved, int32_t *n_cus) {
  if (!h) return SCONE_EINVAL;
  if (n_reserved) *n_reserved = h->cu_reserve;
  if (n_cus) *n_cus = h->n_cus;
  return SCONE_OK;
}

extern "C" int scone_lookup_stream(scone_handle *h, void **stream) {
  if (!h || !stream) return SCONE_EINVAL;
  *stream = h->cu_reserve ? (void *)h->lookup_stream : nullptr;
  return SCONE_OK;
}

int scone_lookup_enter(scone_handle *h, hipStream_t s, hipStream_t *launch) {
  *launch = s;
  if (!h->cu_reserve || s == h->lookup_stream) return SCONE_OK;  // no reserve, or the caller already queues on the masked stream
  // (cu_reserve is read without the lock: set_cu_reserve is not concurrent with lookups, like every table mutation)
  h->lookup_mu.lock();
  hipError_t e = hipEventRecord(h->lookup_in, s);
  if (e == hipSuccess) e = hipStreamWaitEvent(h->lookup_stream, h->lookup_in, 0);
  if (e != hipSuccess) {
    h->lookup_mu.unlock();
    return scone_hip_fail(h, e, "scone_lookup_enter");
  }
  *launch = h->lookup_stream;
  return SCONE_OK;
}

int scone_lookup_leave(scone_handle *h, hipStream_t s, hipStream_t launch) {
  if (launch == s) return SCONE_OK;
  hipError_t e = hipEventRecord(h->lookup_out, launch);
  if (e == hipSuccess) e = hipStreamWaitEvent(s, h->lookup_out, 0);
  h->lookup_mu.unlock();
  return e == hipSuccess ? SCONE_OK : scone_hip_fail(h, e, "scone_lookup_leave");
}

// ---------------------------------------------------------------- kernel timing
static int prof_drain(scone_handle *h) {
  if (h->prof_head == 0) return SCONE_OK;
  SCONE_HIP(h, hipDeviceSynchronize());
  for (uint64_t i = 0; i < h->prof_head; ++i) {
    float ms = 0.f;
    SCONE_HIP(h, hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
    h->prof_ms += ms;
    h->prof_n += 1;
    if (h->prof_samples.size() < SCONE_PROF_MAX_SAMPLES) h->prof_samples.push_back(ms);
  }
  h->prof_head = 0;
  return SCONE_OK;
}

// Whether THIS launch is timed is decided once, in scone_prof_begin, under prof_mu -- scone_profile_enable may flip
// prof_on between the begin and the end of a launch on another thread, and an end / abort that re-read the flag would
// then unlock a mutex its begin never took (or skip the unlock of one it did).  begin and end of a launch run on one
// host thread: the decision is latched in a thread-local (the handle it belongs to: a thread may interleave handles).
static thread_local scone_handle *prof_held_by_this_thread = nullptr;

int scone_prof_begin(scone_handle *h, hipStream_t s) {
  if (!h->prof_on.load(std::memory_order_acquire)) return SCONE_OK;  // profiling off: no lock at all
  if (prof_held_by_this_thread) return scone_fail(h, SCONE_ESTATE, "scone_prof_begin: timed launches do not nest on one thread");
  h->prof_mu.lock();  // until scone_prof_end / scone_prof_abort: the ring slot belongs to this launch
  if (!h->prof_on.load(std::memory_order_relaxed)) {  // switched off between the check and the lock
    h->prof_mu.unlock();
    return SCONE_OK;
  }
  if (h->prof_head == SCONE_PROF_RING) {
    int rc = prof_drain(h);
    if (rc) {
      h->prof_mu.unlock();
      return rc;
    }
  }
  hipError_t e = hipEventRecord(h->prof_ev[2 * h->prof_head], s);
  if (e != hipSuccess) {
    h->prof_mu.unlock();
    return scone_hip_fail(h, e, "hipEventRecord");
  }
  prof_held_by_this_thread = h;
  return SCONE_OK;
}

int scone_prof_end(scone_handle *h, hipStream_t s) {
  if (prof_held_by_this_thread != h) return SCONE_OK;  // this launch's begin did not start a measurement
  prof_held_by_this_thread = nullptr;
  hipError_t e = hipEventRecord(h->prof_ev[2 * h->prof_head + 1], s);
  if (e == hipSuccess) h->prof_head += 1;
  h->prof_mu.unlock();
  return e == hipSuccess ? SCONE_OK : scone_hip_fail(h, e, "hipEventRecord");
}

void scone_prof_abort(scone_handle *h) {
  if (prof_held_by_this_thread != h) return;
  prof_held_by_this_thread = nullptr;
  h->prof_mu.unlock();
}

extern "C" int scone_profile_enable(scone_handle *h, int enable) {
  if (!h) return SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  std::lock_guard<std::mutex> g(h->prof_mu);
  if (enable && !h->prof_ev) {
    h->prof_ev = new (std::nothrow) hipEvent_t[2 * SCONE_PROF_RING]();  // null handles: destroy skips them
    if (!h->prof_ev) return scone_fail(h, SCONE_ENOMEM, "scone_profile_enable: out of memory");
    for (int i = 0; i < 2 * SCONE_PROF_RING; ++i) SCONE_HIP(h, hipEventCreate(&h->prof_ev[i]));
  }
  if (!enable) {
    int rc = prof_drain(h);
    if (rc) return rc;
  }
  h->prof_on = enable != 0;
  return SCONE_OK;
}

extern "C" int scone_profile_read(scone_handle *h, uint64_t *n_launches, double *total_ms, int reset) {
  if (!h) return SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  std::lock_guard<std::mutex> g(h->prof_mu);
  int rc = prof_drain(h);
  if (rc) return rc;
  if (n_launches) *n_launches = h->prof_n;
  if (total_ms) *total_ms = h->prof_ms;
  if (reset) h->prof_n = 0, h->prof_ms = 0.0, h->prof_samples.clear();
  return SCONE_OK;
}

extern "C" int scone_profile_samples(scone_handle *h, float *h_ms, uint64_t cap, uint64_t *n) {
  if (!h || !n || (cap && !h_ms)) return SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  std::lock_guard<std::mutex> g(h->prof_mu);
  int rc = prof_drain(h);
  if (rc) return rc;
  *n = h->prof_samples.size();
  for (uint64_t i = 0; i < *n && i < cap; ++i) h_ms[i] = h->prof_samples[i];
  return SCONE_OK;
}
