// The bandwidth-bound core: sparse row gather + dequantise + ordered fp32 reduce +
// combine with the base token / position embeddings.  No MFMA: every byte fetched
// is used once; the kernel is priced against the HBM roofline.
//
// Replaces, on the GPU:
//   EmbeddingCache.get_token_embeddings / get_embeddings  scone/inference/embedding_cache.py:113-181
//   embeddings.mean(dim=0), zero-fill, .half()            scone/inference/engine.py:247-266
//   wte(input_ids) + f_gram_embeddings + wpe(position_ids) scone/models/language_model.py:239-254
//
// Work decomposition (gfx950, 64-lane waves): a *group* of LPT lanes owns one
// token; a wave therefore carries 64/LPT tokens.  Each lane owns 16-byte vectors
// v = lane, lane+LPT, ... of every row (VEC elements each), so one wave
// instruction reads LPT*16 contiguous bytes of 64/LPT different rows.  For every
// token all K_t row vectors are requested before the first one is consumed
// (K_t <= 10 independent 16-byte loads per lane per vector column); accumulation
// is sequential in the reference's list order in fp32, so results do not depend
// on the launch geometry.
#pragma once
#include "scone_common.h"

#include <type_traits>

namespace scone_gather {


template <int FMT> struct fmt_traits;
template <> struct fmt_traits<SCONE_FMT_F32> { static constexpr int VEC = 4; };
template <> struct fmt_traits<SCONE_FMT_F16> { static constexpr int VEC = 8; };
template <> struct fmt_traits<SCONE_FMT_I8> { static constexpr int VEC = 16; };
template <> struct fmt_traits<SCONE_FMT_I4> { static constexpr int VEC = 32; };

struct table_view {
  scone_row_store st;    // payload rows (local): HBM part + pinned-host part
  const __half *scales;  // I8: [rows]; I4: [rows, d/128]
  long long row_begin;   // owned global id range
  long long row_end;
  long long n_rows;      // global row count
  int row_bytes;         // payload bytes per row
  int d;
};

// ---- output / base dtype helpers -------------------------------------------
template <typename T> struct io;
template <> struct io<float> {
  static __device__ __forceinline__ float ld(const float *p) { return *p; }
  static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct io<__half> {
  static __device__ __forceinline__ float ld(const __half *p) { return __half2float(*p); }
  static __device__ __forceinline__ void st(__half *p, float v) { *p = __float2half_rn(v); }
};
template <> struct io<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
  static __device__ __forceinline__ void st(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};

// load / store N consecutive T elements (N * sizeof(T) is 8 bytes or a multiple of 16)
template <typename T, int N>
__device__ __forceinline__ void load_vec(const T *__restrict__ p, float (&out)[N]) {
  constexpr int BYTES = N * (int)sizeof(T);
  static_assert(BYTES == 8 || BYTES % 16 == 0, "vector width");
  if constexpr (BYTES == 8) {
    const uint2 raw = *reinterpret_cast<const uint2 *>(p);
    const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = io<T>::ld(e + i);
  } else {
    uint4 raw[BYTES / 16];
#pragma unroll
    for (int i = 0; i < BYTES / 16; ++i) raw[i] = reinterpret_cast<const uint4 *>(p)[i];
    const T *e = reinterpret_cast<const T *>(raw);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = io<T>::ld(e + i);
  }
}

template <typename T, int N>
__device__ __forceinline__ void store_vec(T *__restrict__ p, const float (&v)[N]) {
  constexpr int BYTES = N * (int)sizeof(T);
  static_assert(BYTES == 8 || BYTES % 16 == 0, "vector width");
  if constexpr (BYTES == 8) {
    uint2 raw;
    T *e = reinterpret_cast<T *>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) io<T>::st(e + i, v[i]);
    *reinterpret_cast<uint2 *>(p) = raw;
  } else {
    uint4 raw[BYTES / 16];
    T *e = reinterpret_cast<T *>(raw);
#pragma unroll
    for (int i = 0; i < N; ++i) io<T>::st(e + i, v[i]);
#pragma unroll
    for (int i = 0; i < BYTES / 16; ++i) reinterpret_cast<uint4 *>(p)[i] = raw[i];
  }
}

// acc[0..VEC) += dequant(raw); the products scale*q are exact in fp32 (11-bit x 8-bit
// significands), so fmaf(scale, q, acc) == acc + fp32(scale*q): the same value the
// oracle adds when it sums the dequantised fp32 table.
template <int FMT>
__device__ __forceinline__ void accumulate(float (&acc)[fmt_traits<FMT>::VEC], const uint4 &raw, float scale) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  if constexpr (FMT == SCONE_FMT_F32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += __uint_as_float(w[i]);
  } else if constexpr (FMT == SCONE_FMT_F16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] += __half2float(__ushort_as_half((unsigned short)(w[i] & 0xFFFFu)));
      acc[2 * i + 1] += __half2float(__ushort_as_half((unsigned short)(w[i] >> 16)));
    }
  } else if constexpr (FMT == SCONE_FMT_I8) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int q = (int)(w[i] << (24 - 8 * b)) >> 24;  // sign-extended byte b
        acc[4 * i + b] = fmaf(scale, (float)q, acc[4 * i + b]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int q = (int)((w[i] >> (4 * b)) & 0xFu) - 8;
        acc[8 * i + b] = fmaf(scale, (float)q, acc[8 * i + b]);
      }
    }
  }
}

template <int FMT>
__device__ __forceinline__ float load_scale(const table_view &tv, long long lr, int v) {
  if constexpr (FMT == SCONE_FMT_I8) {
    return __half2float(tv.scales[lr]);
  } else if constexpr (FMT == SCONE_FMT_I4) {
    // 32 elements per vector, 128 per group -> 4 vectors share a scale
    return __half2float(tv.scales[lr * (long long)(tv.d / SCONE_I4_GROUP) + scone_i4_scale_slot(v >> 2, tv.d)]);
  } else {
    return 1.0f;
  }
}

// candidate c -> (n, s): n = 1..max_n, s = n-1..0 (window start ascending)
__device__ __forceinline__ void cand_ns(int c, int &n, int &s) {
  n = 1;
  int base = 0;
  while (base + n <= c) {
    base += n;
    ++n;
  }
  s = (n - 1) - (c - base);
}

// how the id list of a token is obtained
enum { SRC_HITS = 0, SRC_CSR = 1 };

struct embed_args {
  table_view tv;
  // id source
  const int32_t *hits;     // [max_n, BT]
  const int32_t *ell;      // per-token owned-id records (wave kernel), or null
  const void *zero_row;    // d * 4 zero bytes
  const int32_t *offsets;  // CSR
  const int32_t *ids;
  long long BT;
  int T;
  int max_n;
  long long tok_begin;  // first flattened position handled (finalize slices)
  long long ntok;       // positions handled
  // combine
  const int32_t *tok;  // [BT] (for wte)
  const int32_t *pos;  // [BT] or null
  const void *wte;
  long long vocab;
  const void *wpe;
  long long n_pos;
  const void *base;  // gather_reduce: [ntok, d] added after the reduce, or null
  int reduce;
  int mode;           // SCONE_MODE_* (honoured by the wave kernels)
  int fused;          // decode-size batch: match inside the lookup kernel (k_embed_fused)
  void *out;          // OutT [ntok, d]       (MODE_FULL / MODE_FINALIZE)
  float *partial;     // fp32 [ntok, d]       (MODE_PARTIAL)
  int32_t *counts;    // [ntok] full K        (MODE_PARTIAL out / MODE_FINALIZE in)
  const float *sums;  // fp32 [ntok, d]       (MODE_FINALIZE in)
  uint32_t *status;
};

enum { MODE_FULL = 0, MODE_PARTIAL = 1, MODE_FINALIZE = 2 };

// D == 0: runtime d.  NCAND = list entries handled per batch (= max_n(max_n+1)/2 of the
// compiled max_n bound for the hit source).
template <int FMT, typename OutT, int D, int LPT, int NCAND, int SRC, int MODE>
__global__ __launch_bounds__(256) void k_embed(const embed_args a) {
  constexpr int VEC = fmt_traits<FMT>::VEC;
  constexpr int GROUPS = 64 / LPT;
  constexpr int UNROLL = D ? (D / VEC + LPT - 1) / LPT : 1;
  static_assert(NCAND <= LPT, "one candidate per lane of the group");
  const int lane = threadIdx.x & 63;
  const int gl = lane & (LPT - 1);      // lane within the group
  const int gbase = lane & ~(LPT - 1);  // first lane of the group
  const long long group = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * GROUPS + lane / LPT;
  const bool active = group < a.ntok;
  const long long p = a.tok_begin + (active ? group : 0);  // flattened position
  const int d = D ? D : a.tv.d;
  const int nv = d / VEC;
  const int i_in_seq = (int)(p % a.T);
  const long long row_begin = a.tv.row_begin, row_end = a.tv.row_end;

  // ---- 1. the token's id list ---------------------------------------------------
  // hit source: lane c of the group fetches candidate c, then the group shares them
  int32_t idc[NCAND];
  int kfull = 0;  // K_t over ALL hits (the mean's divisor), also when some rows are not owned
  int csr_off = 0, csr_k = 0;
  if constexpr (MODE == MODE_FINALIZE) {
    kfull = active ? a.counts[group] : 0;
  } else if constexpr (SRC == SRC_HITS) {
    int32_t my_id = -1;
    if (active && gl < NCAND) {
      int n, s;
      cand_ns(gl, n, s);
      if (n <= a.max_n && i_in_seq - s >= 0) my_id = a.hits[(long long)(n - 1) * a.BT + p - s];
    }
#pragma unroll
    for (int c = 0; c < NCAND; ++c) {
      idc[c] = __shfl(my_id, gbase + c, 64);
      kfull += idc[c] >= 0;
    }
  } else {
    if (active) {
      csr_off = a.offsets[p];
      csr_k = a.offsets[p + 1] - csr_off;
    }
    kfull = csr_k;
  }

  // ---- 2. base-embedding rows (language_model.py:239, :253) ------------------------
  int32_t tokv = 0, posv = 0;
  bool tok_ok = false, pos_ok = false;
  if constexpr (MODE != MODE_PARTIAL) {
    if (a.wte && active) {
      tokv = a.tok[p];
      tok_ok = tokv >= 0 && (long long)tokv < a.vocab;
      if (!tok_ok && gl == 0) atomicOr(a.status, SCONE_ST_BAD_TOKEN);
    }
    if (a.wpe && active) {
      posv = a.pos ? a.pos[p] : i_in_seq;  // default arange(T), language_model.py:248-251
      pos_ok = posv >= 0 && (long long)posv < a.n_pos;
      if (!pos_ok && gl == 0) atomicOr(a.status, SCONE_ST_BAD_TOKEN);
    }
  }
  const OutT *wte = reinterpret_cast<const OutT *>(a.wte);
  const OutT *wpe = reinterpret_cast<const OutT *>(a.wpe);
  const OutT *bas = reinterpret_cast<const OutT *>(a.base);

  // ---- 3. per vector column: gather, ordered reduce, combine, store -----------------
#pragma unroll UNROLL
  for (int v = gl; v < nv; v += LPT) {
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;

    // base rows are requested first so they are in flight with the table rows
    float bw[VEC], bp[VEC], bb[VEC];
    if constexpr (MODE != MODE_PARTIAL) {
      if (tok_ok) load_vec<OutT, VEC>(wte + (long long)tokv * d + (long long)v * VEC, bw);
      if (pos_ok) load_vec<OutT, VEC>(wpe + (long long)posv * d + (long long)v * VEC, bp);
      if (bas && active) load_vec<OutT, VEC>(bas + group * (long long)d + (long long)v * VEC, bb);
    }

    if constexpr (MODE == MODE_FINALIZE) {
      if (active) load_vec<float, VEC>(a.sums + group * (long long)d + (long long)v * VEC, acc);
    } else if constexpr (SRC == SRC_HITS) {
      uint4 raw[NCAND];
      float sc[NCAND];
#pragma unroll
      for (int c = 0; c < NCAND; ++c) {
        const long long id = idc[c];
        if (id >= row_begin && id < row_end) {
          const long long lr = id - row_begin;
          raw[c] = *reinterpret_cast<const uint4 *>(a.tv.st.row((unsigned long long)lr) + (long long)v * 16);
          sc[c] = load_scale<FMT>(a.tv, lr, v);
        }
      }
#pragma unroll
      for (int c = 0; c < NCAND; ++c) {
        const long long id = idc[c];
        if (id >= row_begin && id < row_end) accumulate<FMT>(acc, raw[c], sc[c]);
      }
    } else {
      for (int k0 = 0; k0 < csr_k; k0 += NCAND) {
        uint4 raw[NCAND];
        float sc[NCAND];
        bool own[NCAND];
#pragma unroll
        for (int c = 0; c < NCAND; ++c) {
          own[c] = false;
          if (k0 + c < csr_k) {
            const long long id = a.ids[csr_off + k0 + c];
            if (id < 0 || id >= a.tv.n_rows) {
              if (v == gl) atomicOr(a.status, SCONE_ST_BAD_ID);
            } else if (id >= row_begin && id < row_end) {
              own[c] = true;
              const long long lr = id - row_begin;
              raw[c] = *reinterpret_cast<const uint4 *>(a.tv.st.row((unsigned long long)lr) + (long long)v * 16);
              sc[c] = load_scale<FMT>(a.tv, lr, v);
            }
          }
        }
#pragma unroll
        for (int c = 0; c < NCAND; ++c)
          if (own[c]) accumulate<FMT>(acc, raw[c], sc[c]);
      }
    }

    if (!active) continue;
    if constexpr (MODE == MODE_PARTIAL) {
      store_vec<float, VEC>(a.partial + group * (long long)d + (long long)v * VEC, acc);
    } else {
      if (a.reduce == SCONE_REDUCE_MEAN && kfull > 1) {
        const float kf = (float)kfull;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = acc[e] / kf;  // engine.py:250: sum / K (IEEE division)
      }
      if (tok_ok) {  // language_model.py:242-243: base + f_gram
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = bw[e] + acc[e];
      }
      if (bas) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = bb[e] + acc[e];
      }
      if (pos_ok) {  // language_model.py:253-254: + position embeddings
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = acc[e] + bp[e];
      }
      store_vec<OutT, VEC>(reinterpret_cast<OutT *>(a.out) + group * (long long)d + (long long)v * VEC, acc);
    }
  }
  if constexpr (MODE == MODE_PARTIAL) {
    if (active && gl == 0) a.counts[group] = kfull;
  }
}


// ---------------------------------------------------------------- dispatch
template <int FMT, typename OutT, int D, int LPT, int NCAND, int SRC, int MODE>
int launch_one(scone_handle *h, const embed_args &a, hipStream_t s) {
  constexpr int GROUPS = 64 / LPT;
  const long long groups_per_block = 4 * GROUPS;
  const long long blocks = (a.ntok + groups_per_block - 1) / groups_per_block;
  if (!scone_grid_fits((unsigned long long)blocks, 256)) return scone_fail(h, SCONE_EINVAL, "scone_embed: too many tokens for one launch");
  hipLaunchKernelGGL((k_embed<FMT, OutT, D, LPT, NCAND, SRC, MODE>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

// geometry table: lanes per token for the specialised dims; other dims take the runtime-d kernel
template <int FMT, typename OutT, int NCAND, int SRC, int MODE>
int launch_dim(scone_handle *h, const embed_args &a, hipStream_t s) {
  constexpr int VEC = fmt_traits<FMT>::VEC;
  const int d = a.tv.d;
  if (d == 768 && (768 / VEC) % 16 == 0) return launch_one<FMT, OutT, 768, 16, NCAND, SRC, MODE>(h, a, s);
  if (d == 1024 && (1024 / VEC) % 16 == 0) return launch_one<FMT, OutT, 1024, 16, NCAND, SRC, MODE>(h, a, s);
  return launch_one<FMT, OutT, 0, 16, NCAND, SRC, MODE>(h, a, s);
}

template <int FMT, typename OutT, int SRC, int MODE>
int launch_ncand(scone_handle *h, const embed_args &a, hipStream_t s) {
  if constexpr (MODE == MODE_FINALIZE) {
    return launch_dim<FMT, OutT, 1, SRC, MODE>(h, a, s);
  } else if constexpr (SRC == SRC_CSR) {
    return launch_dim<FMT, OutT, 6, SRC, MODE>(h, a, s);
  } else {
    if (a.max_n <= 3) return launch_dim<FMT, OutT, 6, SRC, MODE>(h, a, s);
    return launch_dim<FMT, OutT, 10, SRC, MODE>(h, a, s);
  }
}

template <int FMT, int SRC, int MODE>
int launch_dtype(scone_handle *h, const embed_args &a, int out_dtype, hipStream_t s) {
  if constexpr (MODE == MODE_PARTIAL) {
    return launch_ncand<FMT, float, SRC, MODE>(h, a, s);
  } else {
    switch (out_dtype) {
      case SCONE_DT_F32: return launch_ncand<FMT, float, SRC, MODE>(h, a, s);
      case SCONE_DT_F16: return launch_ncand<FMT, __half, SRC, MODE>(h, a, s);
      case SCONE_DT_BF16: return launch_ncand<FMT, __hip_bfloat16, SRC, MODE>(h, a, s);
      default: return scone_fail(h, SCONE_EINVAL, "unknown out_dtype");
    }
  }
}

// one definition per table format, each in its own translation unit (parallel builds)
int launch_f32(scone_handle *h, const embed_args &a, int src, int mode, int out_dtype, hipStream_t s);
int launch_f16(scone_handle *h, const embed_args &a, int src, int mode, int out_dtype, hipStream_t s);
int launch_i8(scone_handle *h, const embed_args &a, int src, int mode, int out_dtype, hipStream_t s);
int launch_i4(scone_handle *h, const embed_args &a, int src, int mode, int out_dtype, hipStream_t s);

}  // namespace scone_gather
