// GPU f-gram vocabulary construction: count every n-gram (n = 1..max_n) of a tokenised corpus,
// keep those with count >= min_freq, order them by (count descending, first occurrence ascending)
// and return the first max_f_grams -- the id order of the reference's
//   Counter.update(extract_all_n_grams(text)) ... most_common(max_f_grams)
//   (scone/tokenization/n_gram_extractor.py:72-104, :59-70).
//
// "First occurrence" is the position of an n-gram's first insertion into the Counter: texts in
// order; inside a text all 1-grams left to right, then all 2-grams, ... (extract_all_n_grams).
// Each occurrence gets that 64-bit sequence number; ties in count are broken by its minimum.
//
// Counting uses the same exact-key open-addressing table as the lookup index (slot claim by CAS,
// never a wait) with two side arrays: count (atomicAdd) and first sequence number (atomicMin).
// Ordering: two stable LSD radix sorts from rocPRIM (first-seq ascending, then count descending).
#include "scone_common.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

namespace {

// number of n-gram occurrences of a text of length L: sum_{n=1..min(max_n,L)} (L - n + 1)
__host__ __device__ inline unsigned long long occ_of(long long L, int max_n) {
  unsigned long long s = 0;
  for (int n = 1; n <= max_n; ++n)
    if (L - n + 1 > 0) s += (unsigned long long)(L - n + 1);
  return s;
}

__global__ __launch_bounds__(256) void k_fit_text_occ(const long long *__restrict__ offsets, long long n_texts, int max_n,
                                                      unsigned long long *__restrict__ occ) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_texts) occ[t] = occ_of(offsets[t + 1] - offsets[t], max_n);
}

// One thread per (flat token position g, n): count the n-gram starting at g if it fits in its text.
__global__ __launch_bounds__(256) void k_fit_count(scone_slot *__restrict__ slots, unsigned long long mask,
                                                   unsigned int *__restrict__ cnt, unsigned long long *__restrict__ first,
                                                   const int32_t *__restrict__ tok, long long n_tokens,
                                                   const long long *__restrict__ offsets, long long n_texts,
                                                   const unsigned long long *__restrict__ base_seq, int max_n,
                                                   uint32_t *__restrict__ status) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_tokens * max_n) return;
  const int n = (int)(gid / n_tokens) + 1;
  const long long g = gid - (long long)(n - 1) * n_tokens;
  // text of g: largest t with offsets[t] <= g
  long long lo = 0, hi = n_texts;
  while (hi - lo > 1) {
    const long long mid = (lo + hi) >> 1;
    if (offsets[mid] <= g) lo = mid;
    else hi = mid;
  }
  const long long t0 = offsets[lo], L = offsets[lo + 1] - t0, i = g - t0;
  if (i + n > L) return;
  uint32_t k[SCONE_MAX_N] = {0u, 0u, 0u, 0u};
  for (int j = 0; j < SCONE_MAX_N; ++j) {
    if (j < n) {
      const int32_t v = tok[g + j];
      if (v < 0) {
        atomicOr(status, SCONE_ST_BAD_TOKEN);
        return;
      }
      k[j] = (uint32_t)v;
    }
  }
  const scone_key key = scone_pack_key(k, n, max_n);
  if (!key.ok) {
    atomicOr(status, SCONE_ST_BAD_TOKEN);
    return;
  }
  // insertion order inside the text: all 1-grams, then all 2-grams, ...
  unsigned long long seq = base_seq[lo] + (unsigned long long)i;
  for (int m = 1; m < n; ++m) seq += (unsigned long long)(L - m + 1);
  const unsigned long long tag = ((unsigned long long)key.ext << 32) | 1ull;
  unsigned long long s = scone_hash_key(key.lo, key.ext) & mask;
  for (unsigned long long probe = 0; probe <= mask; ++probe) {
    const unsigned long long old = atomicCAS(&slots[s].lo, 0ull, key.lo);
    if (old == 0ull || old == key.lo) {
      const unsigned long long prev = atomicCAS(&slots[s].hi, 0ull, tag);
      if (prev == 0ull || prev == tag) {
        atomicAdd(&cnt[s], 1u);
        atomicMin(&first[s], seq);
        return;
      }
    }
    s = (s + 1ull) & mask;
  }
  atomicOr(status, SCONE_ST_INDEX_FULL);
}

__global__ __launch_bounds__(256) void k_fit_compact(const unsigned int *__restrict__ cnt,
                                                     const unsigned long long *__restrict__ first,
                                                     unsigned long long cap, unsigned int min_freq,
                                                     unsigned long long *__restrict__ n_sel,
                                                     unsigned long long *__restrict__ n_distinct,
                                                     unsigned long long *__restrict__ sel_first,
                                                     unsigned long long *__restrict__ sel_slot) {
  unsigned long long s = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cap) return;
  const unsigned int c = cnt[s];
  if (c == 0) return;
  atomicAdd(n_distinct, 1ull);
  if (c < min_freq) return;
  const unsigned long long j = atomicAdd(n_sel, 1ull);
  sel_first[j] = first[s];
  sel_slot[j] = s;
}

__global__ __launch_bounds__(256) void k_fit_gather_counts(const unsigned int *__restrict__ cnt,
                                                           const unsigned long long *__restrict__ slot,
                                                           unsigned long long m, unsigned int *__restrict__ out) {
  unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) out[j] = cnt[slot[j]];
}

__global__ __launch_bounds__(256) void k_fit_emit(const scone_slot *__restrict__ slots,
                                                  const unsigned long long *__restrict__ slot,
                                                  const unsigned int *__restrict__ cnt_sorted, unsigned long long n_out,
                                                  int max_n, uint32_t *__restrict__ keys, uint8_t *__restrict__ lens,
                                                  uint32_t *__restrict__ counts) {
  unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_out) return;
  const scone_slot sl = slots[slot[r]];
  const uint32_t ext = (uint32_t)(sl.hi >> 32);
  uint32_t v[4];
  if (max_n <= 3) {
    v[0] = (uint32_t)sl.lo, v[1] = (uint32_t)(sl.lo >> 32), v[2] = ext, v[3] = 0;
  } else {
    v[0] = (uint32_t)(sl.lo & 0xFFFFFFu), v[1] = (uint32_t)((sl.lo >> 24) & 0xFFFFFFu);
    v[2] = (uint32_t)((sl.lo >> 48) & 0xFFFFu) | ((ext & 0xFFu) << 16), v[3] = ext >> 8;
  }
  int len = 0;
  for (int j = 0; j < max_n; ++j) {
    keys[r * max_n + j] = v[j] ? v[j] - 1u : 0u;
    if (v[j]) len = j + 1;
  }
  lens[r] = (uint8_t)len;
  if (counts) counts[r] = cnt_sorted[r];
}

struct dev_buf {
  void *p = nullptr;
  ~dev_buf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

}  // namespace

#define FIT_HIP(call)                                            \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) {                                     \
      (void)hipGetLastError();                                   \
      return e__ == hipErrorOutOfMemory ? SCONE_ENOMEM : SCONE_EHIP; \
    }                                                            \
  } while (0)

extern "C" int scone_fit(int32_t device, const int32_t *d_tokens, int64_t n_tokens, const int64_t *d_text_offsets,
                         int64_t n_texts, int32_t max_n, uint32_t min_freq, uint64_t max_f_grams, uint32_t *d_keys_out,
                         uint8_t *d_lens_out, uint32_t *d_counts_out, uint64_t out_cap, uint64_t *h_n_out,
                         uint64_t *h_n_distinct, scone_stream_t stream) {
  if (max_n < 1 || max_n > SCONE_MAX_N || n_tokens < 0 || n_texts < 0 || !h_n_out) return SCONE_EINVAL;
  *h_n_out = 0;
  if (h_n_distinct) *h_n_distinct = 0;
  if (n_tokens == 0 || n_texts == 0 || max_f_grams == 0) return SCONE_OK;
  if (!d_tokens || !d_text_offsets || !d_keys_out || !d_lens_out) return SCONE_EINVAL;
  scone_device_guard dev_guard__(device);  // the caller's current device is restored on return
  FIT_HIP(dev_guard__.err);
  hipStream_t s = (hipStream_t)stream;

  // table sized for the worst case: every occurrence distinct
  const unsigned long long max_occ = (unsigned long long)n_tokens * (unsigned long long)max_n;
  unsigned long long cap = 1024;
  while (cap < 2 * max_occ) cap <<= 1;
  dev_buf slots, cnt, first, occ, base, counters, status, sel_first, sel_slot, k2, v2, c1, c2, tmp;
  FIT_HIP(slots.alloc(cap * sizeof(scone_slot)));
  FIT_HIP(cnt.alloc(cap * sizeof(unsigned int)));
  FIT_HIP(first.alloc(cap * sizeof(unsigned long long)));
  FIT_HIP(occ.alloc((size_t)n_texts * 8));
  FIT_HIP(base.alloc((size_t)n_texts * 8));
  FIT_HIP(counters.alloc(16));
  FIT_HIP(status.alloc(4));
  FIT_HIP(hipMemsetAsync(slots.p, 0, cap * sizeof(scone_slot), s));
  FIT_HIP(hipMemsetAsync(cnt.p, 0, cap * sizeof(unsigned int), s));
  FIT_HIP(hipMemsetAsync(first.p, 0xFF, cap * sizeof(unsigned long long), s));
  FIT_HIP(hipMemsetAsync(counters.p, 0, 16, s));
  FIT_HIP(hipMemsetAsync(status.p, 0, 4, s));

  // base sequence number of every text (exclusive scan of its occurrence count)
  hipLaunchKernelGGL(k_fit_text_occ, dim3((unsigned)((n_texts + 255) / 256)), dim3(256), 0, s,
                     (const long long *)d_text_offsets, (long long)n_texts, max_n, occ.as<unsigned long long>());
  size_t tmp_bytes = 0;
  FIT_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, occ.as<unsigned long long>(), base.as<unsigned long long>(), 0ull,
                                  (size_t)n_texts, rocprim::plus<unsigned long long>(), s));
  FIT_HIP(tmp.alloc(tmp_bytes));
  FIT_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, occ.as<unsigned long long>(), base.as<unsigned long long>(), 0ull,
                                  (size_t)n_texts, rocprim::plus<unsigned long long>(), s));

  const unsigned long long work = max_occ;
  if (!scone_grid_fits((work + 255) / 256, 256)) return SCONE_EINVAL;  // corpus too large for one launch
  hipLaunchKernelGGL(k_fit_count, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, slots.as<scone_slot>(), cap - 1,
                     cnt.as<unsigned int>(), first.as<unsigned long long>(), d_tokens, (long long)n_tokens,
                     (const long long *)d_text_offsets, (long long)n_texts, base.as<unsigned long long>(), max_n,
                     status.as<uint32_t>());
  FIT_HIP(hipGetLastError());

  // eligible entries (count >= min_freq); their number bounds the sort buffers, so count first
  FIT_HIP(sel_first.alloc(max_occ * 8));
  FIT_HIP(sel_slot.alloc(max_occ * 8));
  if (!scone_grid_fits((cap + 255) / 256, 256)) return SCONE_EINVAL;
  hipLaunchKernelGGL(k_fit_compact, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, s, cnt.as<unsigned int>(),
                     first.as<unsigned long long>(), cap, min_freq, counters.as<unsigned long long>(),
                     counters.as<unsigned long long>() + 1, sel_first.as<unsigned long long>(),
                     sel_slot.as<unsigned long long>());
  unsigned long long h_counters[2] = {0, 0};
  uint32_t h_status = 0;
  FIT_HIP(hipMemcpyAsync(h_counters, counters.p, 16, hipMemcpyDeviceToHost, s));
  FIT_HIP(hipMemcpyAsync(&h_status, status.p, 4, hipMemcpyDeviceToHost, s));
  FIT_HIP(hipStreamSynchronize(s));
  if (h_status & SCONE_ST_BAD_TOKEN) return SCONE_ERANGE;
  if (h_status & SCONE_ST_INDEX_FULL) return SCONE_ENOMEM;
  const unsigned long long m = h_counters[0];
  if (h_n_distinct) *h_n_distinct = h_counters[1];
  if (m == 0) return SCONE_OK;

  // (1) stable sort by first sequence number ascending, (2) stable sort by count descending
  FIT_HIP(k2.alloc(m * 8));
  FIT_HIP(v2.alloc(m * 8));
  FIT_HIP(c1.alloc(m * 4));
  FIT_HIP(c2.alloc(m * 4));
  size_t need = 0;
  FIT_HIP(rocprim::radix_sort_pairs(nullptr, need, sel_first.as<unsigned long long>(), k2.as<unsigned long long>(),
                                    sel_slot.as<unsigned long long>(), v2.as<unsigned long long>(), (size_t)m, 0, 64, s));
  dev_buf tmp2;
  FIT_HIP(tmp2.alloc(need));
  FIT_HIP(rocprim::radix_sort_pairs(tmp2.p, need, sel_first.as<unsigned long long>(), k2.as<unsigned long long>(),
                                    sel_slot.as<unsigned long long>(), v2.as<unsigned long long>(), (size_t)m, 0, 64, s));
  hipLaunchKernelGGL(k_fit_gather_counts, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, cnt.as<unsigned int>(),
                     v2.as<unsigned long long>(), m, c1.as<unsigned int>());
  size_t need2 = 0;
  FIT_HIP(rocprim::radix_sort_pairs_desc(nullptr, need2, c1.as<unsigned int>(), c2.as<unsigned int>(),
                                         v2.as<unsigned long long>(), sel_slot.as<unsigned long long>(), (size_t)m, 0, 32, s));
  dev_buf tmp3;
  FIT_HIP(tmp3.alloc(need2));
  FIT_HIP(rocprim::radix_sort_pairs_desc(tmp3.p, need2, c1.as<unsigned int>(), c2.as<unsigned int>(),
                                         v2.as<unsigned long long>(), sel_slot.as<unsigned long long>(), (size_t)m, 0, 32, s));

  unsigned long long n_out = m < max_f_grams ? m : max_f_grams;
  if (n_out > out_cap) n_out = out_cap;
  hipLaunchKernelGGL(k_fit_emit, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, slots.as<scone_slot>(),
                     sel_slot.as<unsigned long long>(), c2.as<unsigned int>(), n_out, max_n, d_keys_out, d_lens_out,
                     d_counts_out);
  FIT_HIP(hipGetLastError());
  FIT_HIP(hipStreamSynchronize(s));
  *h_n_out = n_out;
  return SCONE_OK;
}
