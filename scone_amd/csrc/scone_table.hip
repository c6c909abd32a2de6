// f-gram table storage: upload, on-GPU quantisation of fp32 rows, synthetic fill and
// the plain row gather.
//
// Replaces, on the GPU:
//   EmbeddingCache.cache_embeddings (storage)        scone/inference/embedding_cache.py:56-111
//   EmbeddingCache.get_embeddings                    scone/inference/embedding_cache.py:113-147
//
// Row formats are this library's own (the reference stores fp32 only); the numpy
// statement of the quantisers is oracle/ref_port.py quantize_i8 / quantize_i4.
#include "scone_common.h"

#include <type_traits>

namespace {

__device__ __forceinline__ float wave_max(float v) {
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}

// One wave per row.  src row index = blockIdx-derived r; destination local row =
// (ids ? ids[r] : row0 + r) - row_begin.
template <int FMT>
__global__ __launch_bounds__(256) void k_store_f32(const float *__restrict__ src, const int64_t *__restrict__ ids,
                                                   unsigned long long row0, unsigned long long nrows,
                                                   unsigned long long row_begin, unsigned long long row_end,
                                                   int d, scone_row_store st, __half *__restrict__ scales,
                                                   uint32_t *__restrict__ status) {
  const int lane = threadIdx.x & 63;
  const unsigned long long nwaves = (unsigned long long)gridDim.x * (blockDim.x >> 6);
  for (unsigned long long r = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < nrows; r += nwaves) {
  unsigned long long g = ids ? (unsigned long long)ids[r] : row0 + r;
  if (g < row_begin || g >= row_end) {  // wave-uniform
    if (lane == 0) atomicOr(status, SCONE_ST_BAD_ID);
    continue;
  }
  const unsigned long long lr = g - row_begin;
  const float *x = src + r * (unsigned long long)d;
  if (FMT == SCONE_FMT_F32) {
    float *o = reinterpret_cast<float *>(st.row(lr));
    for (int e = lane; e < d; e += 64) o[e] = x[e];
  } else if (FMT == SCONE_FMT_F16) {
    __half *o = reinterpret_cast<__half *>(st.row(lr));
    for (int e = lane; e < d; e += 64) o[e] = __float2half_rn(x[e]);
  } else if (FMT == SCONE_FMT_I8) {
    float m = 0.f;
    for (int e = lane; e < d; e += 64) m = fmaxf(m, fabsf(x[e]));
    m = wave_max(m);
    const __half sh = __float2half_rn(m / 127.0f);
    const float sf = __half2float(sh);
    int8_t *o = reinterpret_cast<int8_t *>(st.row(lr));
    for (int e = lane; e < d; e += 64) {
      float q = 0.f;
      if (sf > 0.f) q = fminf(fmaxf(rintf(x[e] / sf), -127.f), 127.f);
      o[e] = (int8_t)(int)q;
    }
    if (lane == 0) scales[lr] = sh;
  } else {  // I4: groups of 128, two elements per lane per group
    const int ng = d / SCONE_I4_GROUP;
    uint8_t *o = st.row(lr);
    for (int grp = 0; grp < ng; ++grp) {
      const float a = x[grp * SCONE_I4_GROUP + 2 * lane];
      const float b = x[grp * SCONE_I4_GROUP + 2 * lane + 1];
      const float m = wave_max(fmaxf(fabsf(a), fabsf(b)));
      const __half sh = __float2half_rn(m / 7.0f);
      const float sf = __half2float(sh);
      int qa = 0, qb = 0;
      if (sf > 0.f) {
        qa = (int)fminf(fmaxf(rintf(a / sf), -7.f), 7.f);
        qb = (int)fminf(fmaxf(rintf(b / sf), -7.f), 7.f);
      }
      o[grp * (SCONE_I4_GROUP / 2) + lane] = (uint8_t)((qa + 8) | ((qb + 8) << 4));
      if (lane == 0) scales[lr * ng + scone_i4_scale_slot(grp, d)] = sh;
    }
  }
  }  // grid-stride loop over rows
}

// hipcc selects v_fma_mixlo_f16 for "convert(a * b)" even with -ffp-contract=off: one rounding of
// the exact product instead of the two roundings (fp32, then fp16) the format is specified with.
// Passing the fp32 product through this keeps the two steps apart.
__device__ __forceinline__ float rounded_f32(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ uint32_t synth_row_base(uint32_t seed, unsigned long long g) {
  return scone_hash32((uint32_t)g + 0x9E3779B9u * (uint32_t)(g >> 32)) ^ seed;
}

__device__ __forceinline__ __half synth_scale(uint32_t seed, unsigned long long counter, float base_scale) {
  uint32_t hsh = scone_hash32(synth_row_base(seed, counter) + 0x51ED27u);
  float u = (float)(hsh >> 8) * (1.0f / 16777216.0f);
  return __float2half_rn(rounded_f32(base_scale * (0.5f + u)));
}

// One wave per row, one 4-byte hash word per lane per step.
template <int FMT>
__global__ __launch_bounds__(256) void k_fill_synth(unsigned long long row_begin, unsigned long long local_rows, int d,
                                                    uint32_t seed, float base_scale, scone_row_store st,
                                                    __half *__restrict__ scales) {
  const int lane = threadIdx.x & 63;
  const unsigned long long nwaves = (unsigned long long)gridDim.x * (blockDim.x >> 6);
  for (unsigned long long lr = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); lr < local_rows; lr += nwaves) {
  const unsigned long long g = row_begin + lr;
  const uint32_t base = synth_row_base(seed, g);
  if (FMT == SCONE_FMT_I4) {
    const int nw = d / 8, ng = d / SCONE_I4_GROUP;
    uint32_t *o = reinterpret_cast<uint32_t *>(st.row(lr));
    for (int w = lane; w < nw; w += 64) o[w] = scone_hash32(base + (uint32_t)w);
    for (int grp = lane; grp < ng; grp += 64)
      scales[lr * ng + scone_i4_scale_slot(grp, d)] = synth_scale(seed, g * (unsigned long long)ng + grp, base_scale);
    continue;
  }
  const int nw = d / 4;
  const __half sh = synth_scale(seed, g, base_scale);
  const float sf = __half2float(sh);
  for (int w = lane; w < nw; w += 64) {
    const uint32_t word = scone_hash32(base + (uint32_t)w);
    if (FMT == SCONE_FMT_I8) {
      reinterpret_cast<uint32_t *>(st.row(lr))[w] = word;
    } else {
      float v[4];
      for (int k = 0; k < 4; ++k) v[k] = rounded_f32((float)(int8_t)(word >> (8 * k)) * sf);
      if (FMT == SCONE_FMT_F32) {
        reinterpret_cast<float4 *>(st.row(lr))[w] = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        __half *o = reinterpret_cast<__half *>(st.row(lr)) + 4 * w;
        for (int k = 0; k < 4; ++k) o[k] = __float2half_rn(v[k]);
      }
    }
  }
  if (FMT == SCONE_FMT_I8 && lane == 0) scales[lr] = sh;
  }  // grid-stride loop over rows
}

// out[i, :] = dequantised row ids[i]; one wave per output row.
template <int FMT>
__global__ __launch_bounds__(256) void k_gather_rows(scone_row_store st, const __half *__restrict__ scales,
                                                     const int64_t *__restrict__ ids, unsigned long long n,
                                                     unsigned long long row_begin, unsigned long long row_end, int d,
                                                     float *__restrict__ out, uint32_t *__restrict__ status) {
  const int lane = threadIdx.x & 63;
  const unsigned long long nwaves = (unsigned long long)gridDim.x * (blockDim.x >> 6);
  for (unsigned long long i = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += nwaves) {
  const long long id = ids[i];
  float *o = out + i * (unsigned long long)d;
  if (id < (long long)row_begin || id >= (long long)row_end) {
    if (lane == 0) atomicOr(status, SCONE_ST_BAD_ID);
    for (int e = lane; e < d; e += 64) o[e] = 0.f;
    continue;
  }
  const unsigned long long lr = (unsigned long long)id - row_begin;
  if (FMT == SCONE_FMT_F32) {
    const float *x = reinterpret_cast<const float *>(st.row(lr));
    for (int e = lane; e < d; e += 64) o[e] = x[e];
  } else if (FMT == SCONE_FMT_F16) {
    const __half *x = reinterpret_cast<const __half *>(st.row(lr));
    for (int e = lane; e < d; e += 64) o[e] = __half2float(x[e]);
  } else if (FMT == SCONE_FMT_I8) {
    const int8_t *x = reinterpret_cast<const int8_t *>(st.row(lr));
    const float sf = __half2float(scales[lr]);
    for (int e = lane; e < d; e += 64) o[e] = (float)x[e] * sf;
  } else {
    const uint8_t *x = st.row(lr);
    const int ng = d / SCONE_I4_GROUP;
    for (int b = lane; b < d / 2; b += 64) {
      const float sf = __half2float(scales[lr * ng + scone_i4_scale_slot((2 * b) / SCONE_I4_GROUP, d)]);
      const uint8_t v = x[b];
      o[2 * b] = (float)((int)(v & 0xF) - 8) * sf;
      o[2 * b + 1] = (float)((int)(v >> 4) - 8) * sf;
    }
  }
  }  // grid-stride loop over ids
}

template <typename F>
int dispatch_fmt(int fmt, F &&f) {
  switch (fmt) {
    case SCONE_FMT_F32: f(std::integral_constant<int, SCONE_FMT_F32>()); return 0;
    case SCONE_FMT_F16: f(std::integral_constant<int, SCONE_FMT_F16>()); return 0;
    case SCONE_FMT_I8: f(std::integral_constant<int, SCONE_FMT_I8>()); return 0;
    case SCONE_FMT_I4: f(std::integral_constant<int, SCONE_FMT_I4>()); return 0;
    default: return -1;
  }
}

int check_table(scone_handle *h, const char *who) {
  if (!h) return SCONE_EINVAL;
  if (h->cfg.dim <= 0 || !h->rows) return scone_fail(h, SCONE_ESTATE, who);
  return SCONE_OK;
}

// the staged-prefetch state caches the scales of the HBM-resident head: drop it when the table changes
void table_modified(scone_handle *h) {
  if (h->stage) scone_stage_destroy(h);
}

// logical <-> physical order of INT4 group scales (scone_i4_scale_slot), in place, one thread per row
__global__ __launch_bounds__(256) void k_i4_scales_reorder(__half *__restrict__ s, unsigned long long n_rows, int ng, int d,
                                                           int to_physical) {
  const unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows || ng > 16) return;
  __half v[16];
  for (int g = 0; g < ng; ++g) v[g] = s[r * ng + g];
  for (int g = 0; g < ng; ++g) {
    if (to_physical)
      s[r * ng + scone_i4_scale_slot(g, d)] = v[g];
    else
      s[r * ng + g] = v[scone_i4_scale_slot(g, d)];
  }
}

bool i4_scales_permuted(const scone_handle *h) {
  return h->cfg.table_fmt == SCONE_FMT_I4 && scone_i4_scale_slot(1, h->cfg.dim) != 1;
}

void i4_scales_reorder_host(uint16_t *s, uint64_t n_rows, int ng, int d, bool to_physical) {
  uint16_t v[16];
  for (uint64_t r = 0; r < n_rows; ++r) {
    for (int g = 0; g < ng; ++g) v[g] = s[r * ng + g];
    for (int g = 0; g < ng; ++g) {
      if (to_physical)
        s[r * ng + scone_i4_scale_slot(g, d)] = v[g];
      else
        s[r * ng + g] = v[scone_i4_scale_slot(g, d)];
    }
  }
}

}  // namespace

extern "C" int scone_table_upload(scone_handle *h, const void *rows, const void *scales, uint64_t row0, uint64_t nrows,
                                  int src_is_device, scone_stream_t stream) {
  int rc = check_table(h, "scone_table_upload: handle has no table (dim == 0)");
  if (rc) return rc;
  if (nrows == 0) return SCONE_OK;
  if (!rows) return scone_fail(h, SCONE_EINVAL, "scone_table_upload: null rows");
  if (h->scale_bytes_per_row && !scales) return scone_fail(h, SCONE_EINVAL, "scone_table_upload: format needs scales");
  if (row0 < h->cfg.row_begin || row0 + nrows > h->cfg.row_end)
    return scone_fail(h, SCONE_ERANGE, "scone_table_upload: rows outside [row_begin,row_end)");
  SCONE_ON_DEVICE(h);
  table_modified(h);
  hipStream_t s = (hipStream_t)stream;
  const uint64_t lr = row0 - h->cfg.row_begin;
  const size_t rb = h->row_payload_bytes;
  const uint64_t n_hot = lr < h->hot_local ? (lr + nrows <= h->hot_local ? nrows : h->hot_local - lr) : 0;
  if (n_hot)  // part that lives in HBM
    SCONE_HIP(h, hipMemcpyAsync(reinterpret_cast<uint8_t *>(h->rows) + lr * rb, rows, n_hot * rb,
                                src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  if (n_hot < nrows)  // part that lives in pinned host memory
    SCONE_HIP(h, hipMemcpyAsync(reinterpret_cast<uint8_t *>(h->rows_host) + (lr + n_hot - h->hot_local) * rb,
                                reinterpret_cast<const uint8_t *>(rows) + n_hot * rb, (nrows - n_hot) * rb,
                                src_is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, s));
  if (h->scale_bytes_per_row) {
    uint8_t *dst = reinterpret_cast<uint8_t *>(h->scales) + lr * h->scale_bytes_per_row;
    SCONE_HIP(h, hipMemcpyAsync(dst, scales, nrows * h->scale_bytes_per_row,
                                src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if (i4_scales_permuted(h)) {  // the ABI speaks the logical group order; the table keeps scone_i4_scale_slot's
      hipLaunchKernelGGL(k_i4_scales_reorder, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, (__half *)dst,
                         (unsigned long long)nrows, h->cfg.dim / SCONE_I4_GROUP, h->cfg.dim, 1);
      SCONE_HIP(h, hipGetLastError());
    }
  }
  if (!src_is_device) SCONE_HIP(h, hipStreamSynchronize(s));  // the host buffer may be pageable / reused
  return SCONE_OK;
}

extern "C" int scone_table_download(scone_handle *h, void *rows, void *scales, uint64_t row0, uint64_t nrows,
                                    int dst_is_device, scone_stream_t stream) {
  int rc = check_table(h, "scone_table_download: handle has no table (dim == 0)");
  if (rc) return rc;
  if (nrows == 0) return SCONE_OK;
  if (!rows) return scone_fail(h, SCONE_EINVAL, "scone_table_download: null rows");
  if (h->scale_bytes_per_row && !scales) return scone_fail(h, SCONE_EINVAL, "scone_table_download: format has scales");
  if (row0 < h->cfg.row_begin || row0 + nrows > h->cfg.row_end)
    return scone_fail(h, SCONE_ERANGE, "scone_table_download: rows outside [row_begin,row_end)");
  SCONE_ON_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  const uint64_t lr = row0 - h->cfg.row_begin;
  const size_t rb = h->row_payload_bytes;
  const uint64_t n_hot = lr < h->hot_local ? (lr + nrows <= h->hot_local ? nrows : h->hot_local - lr) : 0;
  if (n_hot)
    SCONE_HIP(h, hipMemcpyAsync(rows, reinterpret_cast<const uint8_t *>(h->rows) + lr * rb, n_hot * rb,
                                dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
  if (n_hot < nrows)
    SCONE_HIP(h, hipMemcpyAsync(reinterpret_cast<uint8_t *>(rows) + n_hot * rb,
                                reinterpret_cast<const uint8_t *>(h->rows_host) + (lr + n_hot - h->hot_local) * rb,
                                (nrows - n_hot) * rb, dst_is_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost, s));
  if (h->scale_bytes_per_row) {
    SCONE_HIP(h, hipMemcpyAsync(scales, reinterpret_cast<const uint8_t *>(h->scales) + lr * h->scale_bytes_per_row,
                                nrows * h->scale_bytes_per_row,
                                dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    if (i4_scales_permuted(h) && dst_is_device) {  // back to the logical group order
      hipLaunchKernelGGL(k_i4_scales_reorder, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, (__half *)scales,
                         (unsigned long long)nrows, h->cfg.dim / SCONE_I4_GROUP, h->cfg.dim, 0);
      SCONE_HIP(h, hipGetLastError());
    }
  }
  if (!dst_is_device) {
    SCONE_HIP(h, hipStreamSynchronize(s));
    if (i4_scales_permuted(h))
      i4_scales_reorder_host(reinterpret_cast<uint16_t *>(scales), nrows, h->cfg.dim / SCONE_I4_GROUP, h->cfg.dim, false);
  }
  return SCONE_OK;
}

int scone_store_f32_into(scone_handle *h, const scone_row_store &st, void *scales, uint64_t row_begin, uint64_t row_end,
                         const float *d_src, const int64_t *d_ids, uint64_t row0, uint64_t nrows, hipStream_t s) {
  const unsigned blocks = scone_capped_blocks((nrows + 3) / 4);
  int bad = dispatch_fmt(h->cfg.table_fmt, [&](auto F) {
    hipLaunchKernelGGL((k_store_f32<decltype(F)::value>), dim3(blocks), dim3(256), 0, s, d_src, d_ids,
                       (unsigned long long)row0, (unsigned long long)nrows, (unsigned long long)row_begin,
                       (unsigned long long)row_end, h->cfg.dim, st, (__half *)scales, h->d_status);
  });
  if (bad) return scone_fail(h, SCONE_EINVAL, "scone_table_store_f32: bad format");
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

int scone_fill_synth_into(scone_handle *h, const scone_row_store &st, void *scales, uint64_t row_begin, uint64_t nrows,
                          uint32_t seed, float base_scale, hipStream_t s) {
  if (nrows == 0) return SCONE_OK;
  const unsigned blocks = scone_capped_blocks((nrows + 3) / 4);
  int bad = dispatch_fmt(h->cfg.table_fmt, [&](auto F) {
    hipLaunchKernelGGL((k_fill_synth<decltype(F)::value>), dim3(blocks), dim3(256), 0, s, (unsigned long long)row_begin,
                       (unsigned long long)nrows, h->cfg.dim, seed, base_scale, st, (__half *)scales);
  });
  if (bad) return scone_fail(h, SCONE_EINVAL, "scone_table_fill_synthetic: bad format");
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

static int store_f32_common(scone_handle *h, const float *d_src, const int64_t *d_ids, uint64_t row0, uint64_t nrows,
                            hipStream_t s) {
  if (nrows == 0) return SCONE_OK;
  if (!d_src) return scone_fail(h, SCONE_EINVAL, "scone_table_store_f32: null rows");
  SCONE_ON_DEVICE(h);
  table_modified(h);
  return scone_store_f32_into(h, scone_store_of(h), h->scales, h->cfg.row_begin, h->cfg.row_end, d_src, d_ids, row0, nrows, s);
}

extern "C" int scone_table_store_f32(scone_handle *h, const float *d_rows_f32, uint64_t row0, uint64_t nrows,
                                     scone_stream_t stream) {
  int rc = check_table(h, "scone_table_store_f32: handle has no table (dim == 0)");
  if (rc) return rc;
  if (row0 < h->cfg.row_begin || row0 + nrows > h->cfg.row_end)
    return scone_fail(h, SCONE_ERANGE, "scone_table_store_f32: rows outside [row_begin,row_end)");
  return store_f32_common(h, d_rows_f32, nullptr, row0, nrows, (hipStream_t)stream);
}

extern "C" int scone_table_store_f32_ids(scone_handle *h, const float *d_rows_f32, const int64_t *d_ids,
                                         uint64_t nrows, scone_stream_t stream) {
  int rc = check_table(h, "scone_table_store_f32_ids: handle has no table (dim == 0)");
  if (rc) return rc;
  if (nrows && !d_ids) return scone_fail(h, SCONE_EINVAL, "scone_table_store_f32_ids: null ids");
  return store_f32_common(h, d_rows_f32, d_ids, 0, nrows, (hipStream_t)stream);
}

extern "C" int scone_table_fill_synthetic(scone_handle *h, uint32_t seed, float base_scale, scone_stream_t stream) {
  int rc = check_table(h, "scone_table_fill_synthetic: handle has no table (dim == 0)");
  if (rc) return rc;
  SCONE_ON_DEVICE(h);
  table_modified(h);
  rc = scone_fill_synth_into(h, scone_store_of(h), h->scales, h->cfg.row_begin, h->local_rows, seed, base_scale,
                             (hipStream_t)stream);
  if (rc) return rc;
  return scone_shard_fill_head_synth(h, seed, base_scale, (hipStream_t)stream);  // the replicated head of a shard, if any
}

extern "C" int scone_table_gather_rows(scone_handle *h, const int64_t *d_ids, uint64_t n, float *d_out,
                                       scone_stream_t stream) {
  int rc = check_table(h, "scone_table_gather_rows: handle has no table (dim == 0)");
  if (rc) return rc;
  if (n == 0) return SCONE_OK;
  if (!d_ids || !d_out) return scone_fail(h, SCONE_EINVAL, "scone_table_gather_rows: null pointer");
  SCONE_ON_DEVICE(h);
  const unsigned blocks = scone_capped_blocks((n + 3) / 4);
  int bad = dispatch_fmt(h->cfg.table_fmt, [&](auto F) {
    hipLaunchKernelGGL((k_gather_rows<decltype(F)::value>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       scone_store_of(h), (const __half *)h->scales, d_ids, (unsigned long long)n,
                       (unsigned long long)h->cfg.row_begin, (unsigned long long)h->cfg.row_end, h->cfg.dim, d_out,
                       h->d_status);
  });
  if (bad) return scone_fail(h, SCONE_EINVAL, "scone_table_gather_rows: bad format");
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}
