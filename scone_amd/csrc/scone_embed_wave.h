// Wave-per-token kernels of the fused lookup: every embedding dim that is a multiple of 8 runs here.
//
//   k_embed_wave      two-kernel form (after k_match_ell): d = 768 / 1024 / 1280 specialised, persistent grid, the wave's
//                     wpe row in registers, the NEXT token's id record prefetched while the current one is reduced
//   k_embed_fused     one launch, match inside (lane c probes candidate window c, wavefront ballot -> K and the id list):
//                     batches up to 32768 tokens, both lookup modes
//   k_embed_wave_any  any other d % 8 == 0: the row is walked in units of 8 elements
//   k_embed_csr_wave  caller-supplied id lists (scone_gather_reduce)
//   k_finalize_wave   second half of the partial-sum exchange of row-sharded tables
//
// One 64-lane wavefront owns one token at a time, so everything that steers the
// gather is wave-uniform: the candidate f-gram ids, the token and position ids and
// the row base addresses live in SGPRs (scalar loads through the constant cache,
// counted by lgkmcnt -- independent of the vector-memory queue), every branch is a
// scalar branch, and each row is fetched by wave instructions that each cover a
// contiguous run of bytes (segment lane map below).  All K_t row loads plus the wte
// row load of a token are issued back to back before the first one is consumed (K is a
// compile-time constant inside a K-way switch, so the waits are exact vmcnt counts);
// memory-level parallelism across tokens comes from occupancy (5-7 waves per SIMD;
// wave_occupancy<> keeps the register budget free of spills).
//
// Same arithmetic as k_embed (scone_gather_impl.h): sequential fp32 accumulation in
// the reference's list order, correctly rounded sum / K, (wte + mean) + wpe, one rounding to OutT.
#pragma once

#include "scone_gather_impl.h"
#include "scone_probe.h"

namespace scone_gather {

struct wave_params {
  long long BT;
  long long row_begin, row_end;
  long long vocab, n_pos;
  int T;
  int max_n;
  int reduce;
  int B;
  int mode;            // SCONE_MODE_*
  int pos_groups;      // ceil(T / 4): a workgroup's 4 waves own 4 consecutive positions
  int seqs_per_block;  // sequences walked by one workgroup
  unsigned n_blocks;   // workgroups with work (the grid is rounded up to a multiple of 8)
};

template <typename T> struct pack_io;
template <> struct pack_io<float> {
  static constexpr int PER_WORD = 1;
  static __device__ __forceinline__ void unpack(uint32_t w, float *o) { o[0] = __uint_as_float(w); }
  static __device__ __forceinline__ uint32_t pack(const float *v) { return __float_as_uint(v[0]); }
};
template <> struct pack_io<__half> {
  static constexpr int PER_WORD = 2;
  static __device__ __forceinline__ void unpack(uint32_t w, float *o) {
    o[0] = __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu)));
    o[1] = __half2float(__ushort_as_half((unsigned short)(w >> 16)));
  }
  static __device__ __forceinline__ uint32_t pack(const float *v) {
    return (uint32_t)__half_as_ushort(__float2half_rn(v[0])) | ((uint32_t)__half_as_ushort(__float2half_rn(v[1])) << 16);
  }
};
template <> struct pack_io<__hip_bfloat16> {
  static constexpr int PER_WORD = 2;
  static __device__ __forceinline__ void unpack(uint32_t w, float *o) {
    o[0] = __uint_as_float(w << 16);
    o[1] = __uint_as_float(w & 0xFFFF0000u);
  }
  static __device__ __forceinline__ uint32_t pack(const float *v) {
    const __hip_bfloat16 a = __float2bfloat16(v[0]), b = __float2bfloat16(v[1]);
    return (uint32_t)(*reinterpret_cast<const unsigned short *>(&a)) |
           ((uint32_t)(*reinterpret_cast<const unsigned short *>(&b)) << 16);
  }
};

// acc[e] = fma(sc, q_e, acc[e]) for the 8 offset-binary nibbles of w, in 2.5 instead of 4 VALU ops per element:
// w ^ 0x88888888 turns every nibble into two's complement; with a nibble in the HIGH half of a byte the
// sign-extending byte convert (v_cvt_f32_i32 sext(BYTE_k), one SDWA instruction) yields 16 q, and
// fma(sc / 16, 16 q, acc) equals fma(sc, q, acc) bit for bit (sc / 16 and the product are exact).
__device__ __forceinline__ void i4_accumulate(uint32_t w, float sc, float *acc) {
  const uint32_t t = w ^ 0x88888888u;
  const uint32_t ev = (t << 4) & 0xF0F0F0F0u;  // elements 0, 2, 4, 6 in bytes 0..3
  const uint32_t od = t & 0xF0F0F0F0u;         // elements 1, 3, 5, 7
  const float sc16 = sc * 0.0625f;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int qe = (int)(ev << (24 - 8 * b)) >> 24;
    const int qo = (int)(od << (24 - 8 * b)) >> 24;
    acc[2 * b] = fmaf(sc16, (float)qe, acc[2 * b]);
    acc[2 * b + 1] = fmaf(sc16, (float)qo, acc[2 * b + 1]);
  }
}

// Lane <-> element map.  A row is cut into segments of 512 elements; inside a segment lane l owns
// 8 consecutive elements (4 in a trailing 256-element segment).  With this map every wave
// instruction -- INT8 row loads (8 or 4 B/lane), fp16 wte / wpe loads and output stores
// (16 or 8 B/lane) -- covers a CONTIGUOUS run of bytes, so every 64-B sector it touches is
// touched whole (a plain "12 consecutive elements per lane" map makes each fp16 access
// 16-of-every-24 bytes: two partial writes per sector).
template <int FMT, int D> struct wave_geom {
  static constexpr int EPL = D / 64;  // elements per lane
  static constexpr int NSEG = (D + 511) / 512;
  static constexpr int BPE4 = FMT == SCONE_FMT_F32 ? 16 : FMT == SCONE_FMT_F16 ? 8 : FMT == SCONE_FMT_I8 ? 4 : 2;  // bytes per 4 elements
  static constexpr int ROW_BYTES = D / 4 * BPE4;
  static constexpr int NBR = ROW_BYTES / 64;  // row bytes per lane
  static constexpr int seg_elems(int s) { return ((D - 512 * s) >= 512 ? 512 : (D - 512 * s)) / 64; }  // per lane
  static constexpr int seg_first(int s) { return 512 * s; }                                            // first element
  static constexpr int seg_acc(int s) { return 8 * s; }                                                // index into acc[]
  static constexpr int seg_row_words(int s) { return seg_elems(s) * BPE4 / 16; }
  static constexpr int seg_row_word0(int s) { return 8 * s * BPE4 / 16; }
  static constexpr bool OK = (D % 256 == 0) && (D <= 1280) && (seg_elems(NSEG - 1) * BPE4 % 16 == 0) &&
                             (FMT != SCONE_FMT_I4 || SCONE_I4_GROUP % 8 == 0);
};

// waves per SIMD to ask of the register allocator: rows in flight (NC x NBR/4) + wte/wpe/out words +
// accumulators + addressing, rounded to the 8-register allocation granule (512 registers per SIMD lane).
// The estimate must not undershoot: a kernel squeezed below its need spills to scratch INSIDE the token loop
// (INT4 d = 1024 at 6 waves/SIMD: 44 B/lane of scratch, +2 % WRITE_SIZE, 5 % slower than at 5 waves without spills).
#ifndef SCONE_WAVE_SLACK
#define SCONE_WAVE_SLACK 14
#endif
// Rows at or beyond this (local) row number are fetched with streaming ("nt") loads: f-gram ids are frequency-ordered
// (Counter.most_common), so the head of the table -- every GPT-2 unigram and the hottest f-grams -- is what far-apart
// tokens re-reference and what is worth keeping in the 4 MB L2 next to the hot wte rows; a row of the tail is shared by
// the 2-3 adjacent tokens it covers (still an L2 hit: they are in flight together) and then dead.  Measured, gather
// kernel, alternating runs on one box: S_uniform 0.690 -> 0.681 ms (thresholds 4K / 64K / 256K: 0.681 / 0.681 / 0.677),
// Zipf stream 0.584 -> 0.569 ms; streaming loads for EVERY row: 0.69 -> 0.75 ms (the unigram rows are evicted too).
#ifndef SCONE_NT_FROM_ROW
#define SCONE_NT_FROM_ROW 65536
#endif
#ifndef SCONE_WAVE_BLOCKS
#define SCONE_WAVE_BLOCKS 4096  // ~256 CUs x 8 resident workgroups x 2 rounds
#endif
// The HIGH-OCCUPANCY variant of k_embed_wave: the wave's position row lives in LDS instead of 6-10 VGPRs, which buys one
// to three more waves per SIMD (INT8 d = 768: 7 -> 8, fp16 d = 768: 6 -> 7, INT4 d = 1024: 5 -> 8).  More waves = more
// row requests in flight per CU, which is what a latency-bound gather converts into bandwidth.  launch_wave takes it
// whenever the position ids are the default arange(T) (with caller-supplied positions the row is per token: nothing to
// keep).  Measured against the round-1 kernel, alternating runs on one box (round 2's tools/ab_r1.sh, gather kernel, 1M tokens):
//   INT4 100M x 1024 0.794 -> 0.742 ms   INT4 1M x 1024 0.814 -> 0.771 ms   fp16 1M x 768 0.792 -> 0.735 ms
//   INT8 1M x 768 Zipf stream 0.572 -> 0.542 ms   INT8 1M x 768 S_uniform 0.726 -> 0.705 ms   INT8 10M x 1024 0.906 -> 0.908 ms
// Switches for A/B builds: SCONE_HIOCC_MASK (bit FMT), SCONE_HIOCC_SLACK_CUT (registers taken off the occupancy
// estimate of the variant).
#ifndef SCONE_HIOCC_MASK
#define SCONE_HIOCC_MASK 15
#endif
#ifndef SCONE_HIOCC_SLACK_CUT
#define SCONE_HIOCC_SLACK_CUT 8
#endif
// TIMING PROBES (tools/timing_probes.sh; results are WRONG on purpose, never part of the shipped library): what is left of the
// kernel's time when one of its memory streams stops missing L2.
//   SCONE_PROBE_NO_WTE      the token's wte row is the handle's zero row (always an L1 / L2 hit)
//   SCONE_PROBE_ROWS_LOCAL  every f-gram row index & 4095: a 3-MB region, L2-resident
//   SCONE_PROBE_NO_STORE    the output store sits behind a run-time condition that never holds
//   SCONE_PROBE_NO_MATH     every load and store stays, the dequantise / sum / mean / combine arithmetic becomes one XOR per word
//   SCONE_LOCKSTEP          (not a probe: results stay exact) a workgroup barrier in front of every token, so that the 4 waves
//                           that own 4 consecutive positions of one sequence issue their row loads together
template <int FMT> struct wave_hiocc {
  static constexpr bool available = ((SCONE_HIOCC_MASK >> FMT) & 1) != 0;
};
template <int FMT, typename OutT, int D, int MAXN, bool FIXED_POS, bool HIOCC = false> struct wave_occupancy {
  static constexpr int NC = MAXN * (MAXN + 1) / 2;
  static constexpr int KL = NC;  // rows in flight
  static constexpr int NWO = wave_geom<FMT, D>::EPL * (int)sizeof(OutT) / 4;
  // INT4 also holds one group-scale word per (row, segment) in flight
  static constexpr int EST = KL * (wave_geom<FMT, D>::NBR / 4) + (FMT == SCONE_FMT_I4 ? KL * (D == 1024 ? 1 : wave_geom<FMT, D>::NSEG) : 0) +
                             (FIXED_POS && !HIOCC ? 4 : 3) * NWO +
                             wave_geom<FMT, D>::EPL + SCONE_WAVE_SLACK - (HIOCC && MAXN <= 3 && sizeof(OutT) == 2 ? SCONE_HIOCC_SLACK_CUT : 0) +
                             (HIOCC && sizeof(OutT) == 4 ? 8 : 0) +  // fp32 output: the LDS read-back of the position row is twice as wide
                             (FMT == SCONE_FMT_I4 && !HIOCC ? 8 : 0) +  // INT4 at 6 waves spills 44 B/lane on the K >= 4 paths
                             (MAXN >= 4 ? (FMT == SCONE_FMT_I4 ? 24 : 8) : 0) +  // the 10-way switch keeps more addresses live
                             (std::is_same<OutT, __hip_bfloat16>::value ? 4 : 0);  // round-to-nearest-even by hand
  static constexpr int ALLOC = (EST + 7) / 8 * 8;
#ifdef SCONE_FORCE_WAVES
  static constexpr int WAVES = SCONE_FORCE_WAVES;  // A/B builds
#else
  static constexpr int WAVES = 512 / ALLOC >= 8 ? 8 : (512 / ALLOC < 1 ? 1 : 512 / ALLOC);
#endif
};

// words of an OutT vector (wte / wpe / out row) owned by this lane, segment by segment
template <int FMT, typename OutT, int D>
__device__ __forceinline__ void ld_out_row(const uint8_t *__restrict__ row, uint32_t lane,
                                           uint32_t (&w)[wave_geom<FMT, D>::EPL * (int)sizeof(OutT) / 4]) {
  using G = wave_geom<FMT, D>;
#pragma unroll
  for (int s = 0; s < G::NSEG; ++s) {
    constexpr int dummy = 0;
    (void)dummy;
    const int nw = G::seg_elems(s) * (int)sizeof(OutT) / 4;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(row + G::seg_first(s) * (int)sizeof(OutT) +
                                                            lane * (uint32_t)(G::seg_elems(s) * (int)sizeof(OutT)));
#pragma unroll
    for (int i = 0; i < nw; ++i) w[G::seg_acc(s) * (int)sizeof(OutT) / 4 + i] = p[i];
  }
}

template <int FMT, typename OutT, int D, bool STREAMING = true>
__device__ __forceinline__ void st_out_row(uint8_t *__restrict__ row, uint32_t lane,
                                           const uint32_t (&w)[wave_geom<FMT, D>::EPL * (int)sizeof(OutT) / 4]) {
  using G = wave_geom<FMT, D>;
#pragma unroll
  for (int s = 0; s < G::NSEG; ++s) {
    const int nw = G::seg_elems(s) * (int)sizeof(OutT) / 4;
    uint32_t *p = reinterpret_cast<uint32_t *>(row + G::seg_first(s) * (int)sizeof(OutT) +
                                               lane * (uint32_t)(G::seg_elems(s) * (int)sizeof(OutT)));
#pragma unroll
    for (int i = 0; i < nw; ++i) {
      if constexpr (STREAMING)
        __builtin_nontemporal_store(w[G::seg_acc(s) * (int)sizeof(OutT) / 4 + i], p + i);  // write-once output
      else
        p[i] = w[G::seg_acc(s) * (int)sizeof(OutT) / 4 + i];
    }
  }
}

// One token with exactly K owned rows: straight-line code, every load unconditional and
// issued before the first use (the K-way switch in the kernel keeps K a compile-time constant,
// so the row registers are plain scalars and the waits are exact vmcnt counts).
template <int FMT, typename OutT, int D, int K, bool FIXED_POS, bool PARTIAL, bool WPE_LDS = false>
__device__ __forceinline__ void embed_token(const scone_row_store &rows, const void *__restrict__ scales_v,
                                            const int32_t *__restrict__ rec, long long row_begin, int kfull, int reduce,
                                            const uint8_t *__restrict__ wte_row, const uint8_t *__restrict__ wpe_row,
                                            const uint32_t (&wpe_words)[wave_geom<FMT, D>::EPL * (int)sizeof(OutT) / 4],
                                            uint8_t *__restrict__ out_row, uint32_t lane,
                                            const uint32_t *__restrict__ wpe_lds = nullptr) {
  using G = wave_geom<FMT, D>;
  constexpr int EPL = G::EPL, NWR = G::NBR / 4, NSEG = G::NSEG;
  constexpr int NWO = EPL * (int)sizeof(OutT) / 4;
  constexpr int OPW = pack_io<OutT>::PER_WORD;
  constexpr int KK = K > 0 ? K : 1;

  uint32_t bw[NWO], bp[NWO];
  if constexpr (PARTIAL) {
#pragma unroll
    for (int w = 0; w < NWO; ++w) bw[w] = bp[w] = 0u;
  } else {
    ld_out_row<FMT, OutT, D>(wte_row, lane, bw);
  }
  if constexpr (PARTIAL) {
  } else if constexpr (FIXED_POS) {
    // this wave's position row, loaded once: in registers, or (high-occupancy variant) read back from LDS at the end
#pragma unroll
    for (int w = 0; w < NWO; ++w) bp[w] = WPE_LDS ? 0u : wpe_words[w];
  } else {
    ld_out_row<FMT, OutT, D>(wpe_row, lane, bp);
  }
  uint32_t raw[KK][NWR];
  uint32_t scw[KK][NSEG];
#pragma unroll
  for (int k = 0; k < K; ++k) {
#ifdef SCONE_PROBE_ROWS_LOCAL
    const long long lr = ((long long)rec[k] - row_begin) & 4095;
#else
    const long long lr = (long long)rec[k] - row_begin;
#endif
    const uint8_t *rp = rows.row((unsigned long long)lr);  // HBM or mapped host DRAM (wave-uniform select)
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
      const uint32_t *p = reinterpret_cast<const uint32_t *>(rp + G::seg_first(s) / 4 * G::BPE4 +
                                                              lane * (uint32_t)(G::seg_elems(s) * G::BPE4 / 4));
#pragma unroll
      for (int i = 0; i < G::seg_row_words(s); ++i)  // wave-uniform choice between a plain and a streaming load
        raw[k][G::seg_row_word0(s) + i] = lr >= SCONE_NT_FROM_ROW ? __builtin_nontemporal_load(p + i) : p[i];
      if constexpr (FMT == SCONE_FMT_I8) {
        scw[k][s] = s == 0 ? reinterpret_cast<const uint32_t *>(scales_v)[lr >> 1] : 0u;  // two half scales per word (scalar load)
      } else if constexpr (FMT == SCONE_FMT_I4) {
        if constexpr (D == 1024) {
          // scone_i4_scale_slot: the lane's two group scales (segment 0: group lane / 16, segment 1: group 4 + lane / 16)
          // are the two halves of ONE dword -- one load and one register per row instead of two
          scw[k][s] = s == 0 ? reinterpret_cast<const uint32_t *>(scales_v)[lr * (D / SCONE_I4_GROUP / 2) + (lane >> 4)] : 0u;
        } else {
          scw[k][s] = reinterpret_cast<const unsigned short *>(scales_v)[lr * (D / SCONE_I4_GROUP) +
                                                                         scone_i4_scale_slot((G::seg_first(s) + lane * G::seg_elems(s)) / SCONE_I4_GROUP, D)];
        }
      } else {
        scw[k][s] = 0;
      }
    }
  }

  float acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float sc0 = 1.0f;
    if constexpr (FMT == SCONE_FMT_I8) {
      const long long lr = (long long)rec[k] - row_begin;
      sc0 = __half2float(__ushort_as_half((unsigned short)((lr & 1) ? (scw[k][0] >> 16) : (scw[k][0] & 0xFFFFu))));
    }
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
      float sc = sc0;
      if constexpr (FMT == SCONE_FMT_I4) {
        if constexpr (D == 1024)
          sc = __half2float(__ushort_as_half((unsigned short)(s == 0 ? (scw[k][0] & 0xFFFFu) : (scw[k][0] >> 16))));
        else
          sc = __half2float(__ushort_as_half((unsigned short)scw[k][s]));
      }
      // acc[seg_acc(s) ..] += dequant(raw[k][seg words]); exact products, list order (see accumulate<>)
#pragma unroll
      for (int i = 0; i < G::seg_row_words(s); ++i) {
        const uint32_t w = raw[k][G::seg_row_word0(s) + i];
#ifdef SCONE_PROBE_NO_MATH
        acc[(G::seg_row_word0(s) + i) % EPL] = __uint_as_float(__float_as_uint(acc[(G::seg_row_word0(s) + i) % EPL]) ^ w ^ __float_as_uint(sc));
        continue;
#endif
        if constexpr (FMT == SCONE_FMT_F32) {
          acc[G::seg_acc(s) + i] += __uint_as_float(w);
        } else if constexpr (FMT == SCONE_FMT_F16) {
          acc[G::seg_acc(s) + 2 * i] += __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu)));
          acc[G::seg_acc(s) + 2 * i + 1] += __half2float(__ushort_as_half((unsigned short)(w >> 16)));
        } else if constexpr (FMT == SCONE_FMT_I8) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int q = (int)(w << (24 - 8 * b)) >> 24;
            acc[G::seg_acc(s) + 4 * i + b] = fmaf(sc, (float)q, acc[G::seg_acc(s) + 4 * i + b]);
          }
        } else {
          i4_accumulate(w, sc, &acc[G::seg_acc(s) + 8 * i]);
        }
      }
    }
  }
  if constexpr (PARTIAL) {
    // row-sharded tables: the fp32 sum over the rows THIS handle owns goes out as is (the other
    // shards' sums are added by the reduce-scatter; scone_finalize divides and combines)
    uint32_t pw[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) pw[e] = __float_as_uint(acc[e]);
    st_out_row<FMT, float, D, false>(out_row, lane, pw);
    return;
  }
#ifndef SCONE_PROBE_NO_MATH
  if (reduce == SCONE_REDUCE_MEAN && kfull > 1) {
    // engine.py:250: sum / K.  Correctly rounded quotient without the full division sequence
    // (Markstein): y = RN(1/K); q0 = RN(x*y); r = x - q0*K (exact in an fma); q = RN(q0 + r*y).
    const float kf = (float)kfull;
    const float y = 1.0f / kf;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const float q0 = acc[e] * y;
      const float r = fmaf(-kf, q0, acc[e]);
      acc[e] = fmaf(r, y, q0);
    }
  }
#endif
  uint32_t ow[NWO];
#ifdef SCONE_PROBE_NO_MATH
#pragma unroll
  for (int w = 0; w < NWO; ++w) ow[w] = bw[w] ^ (WPE_LDS ? wpe_lds[w * 64 + lane] : bp[w]) ^ __float_as_uint(acc[w % EPL]);
  st_out_row<FMT, OutT, D>(out_row, lane, ow);
  return;
#endif
#pragma unroll
  for (int w = 0; w < NWO; ++w) {
    float b[OPW], c[OPW], v[OPW];
    pack_io<OutT>::unpack(bw[w], b);
    if constexpr (WPE_LDS)
      pack_io<OutT>::unpack(wpe_lds[w * 64 + lane], c);
    else
      pack_io<OutT>::unpack(bp[w], c);
#pragma unroll
    for (int k = 0; k < OPW; ++k) v[k] = (b[k] + acc[w * OPW + k]) + c[k];  // language_model.py:242-243, :253-254
    ow[w] = pack_io<OutT>::pack(v);
  }
#ifdef SCONE_PROBE_NO_STORE
  if (reduce != 0x5C0E) return;
#endif
  st_out_row<FMT, OutT, D>(out_row, lane, ow);
}

// Work assignment: a workgroup's 4 waves own 4 CONSECUTIVE positions i0..i0+3 and walk the same
// run of sequences b = b0..b1-1 together.  Adjacent tokens of one sequence are therefore in
// flight on one CU at the same time (the f-gram rows they share hit L1/L2), and with the
// default position ids (arange(T), language_model.py:248-251) a wave's wpe row never changes:
// FIXED_POS keeps it in registers, removing d*sizeof(OutT) bytes of L1/L2 traffic per token.
// HIOCC: the wave's position row lives in LDS instead of registers (see the switches above).  (Tried and dropped for
// the same goal -- more bytes in flight per CU on the small-row formats: two tokens per wave iteration, issue A, issue
// B, finish A, finish B: 160 VGPRs -> 3 waves / SIMD, +5 % time; the NEXT token's wte row requested behind the current
// rows and consumed one iteration later: -1.4 % on the 1M-row table at 5 waves, but it costs the registers that buy a
// sixth and seventh wave, which are worth more.)
template <int FMT, typename OutT, int D, int MAXN, bool FIXED_POS, bool PARTIAL = false, bool HIOCC = false>
__global__ __launch_bounds__(256, (wave_occupancy<FMT, OutT, D, MAXN, FIXED_POS, HIOCC>::WAVES)) void k_embed_wave(
    const scone_row_store rows, const void *__restrict__ scales_v, const int32_t *__restrict__ ell,
    const int32_t *__restrict__ tok, const int32_t *__restrict__ pos, const OutT *__restrict__ wte,
    const OutT *__restrict__ wpe, const uint8_t *__restrict__ zero_row, OutT *__restrict__ out,
    int32_t *__restrict__ counts, uint32_t *__restrict__ status, const wave_params q) {
  constexpr int NC = MAXN * (MAXN + 1) / 2;
  constexpr int W = MAXN <= 3 ? 8 : 16;
  constexpr int NWO = wave_geom<FMT, D>::EPL * (int)sizeof(OutT) / 4;
  const uint32_t lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), so
  // logical id = (b % 8) * (grid / 8) + b / 8 puts NEIGHBOURING position groups of a sequence run on
  // one XCD: the f-gram rows that straddle a group boundary are then served by the same L2.
#ifndef SCONE_NO_XCD_REMAP
  const unsigned per_xcd = gridDim.x >> 3;  // the grid is a multiple of 8
  const unsigned logical = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
#else
  const unsigned logical = blockIdx.x;
#endif
  if (logical >= q.n_blocks) return;
  const int pg = (int)(logical % (unsigned)q.pos_groups);
  const int chunk = (int)(logical / (unsigned)q.pos_groups);
  const int i = pg * 4 + wave;  // position inside the sequence (wave-constant)
  if (i >= q.T) return;
  const int b0 = chunk * q.seqs_per_block;
  int b1 = b0 + q.seqs_per_block;
  if (b1 > q.B) b1 = q.B;
  if (b0 >= b1) return;
  long long p = (long long)b0 * q.T + i;
  const long long p_end = (long long)b1 * q.T;  // p advances by T

  auto load_rec = [&](long long pp, int32_t (&r)[W]) {
#pragma unroll
    for (int j = 0; j < W; ++j) r[j] = ell[pp * W + j];  // one aligned s_load_dwordx8 / x16
  };

  uint32_t wpe_words[NWO];
#pragma unroll
  for (int w = 0; w < NWO; ++w) wpe_words[w] = 0u;
  if constexpr (FIXED_POS) {
    const bool ok = wpe && (long long)i < q.n_pos;
    if (wpe && !ok && lane == 0) atomicOr(status, SCONE_ST_BAD_TOKEN);
    const uint8_t *r = ok ? reinterpret_cast<const uint8_t *>(wpe + (long long)i * D) : zero_row;
    ld_out_row<FMT, OutT, D>(r, lane, wpe_words);
  }
  // HIOCC: the wave's position row lives in LDS (word w of lane l at [w][l]: conflict-free), not in registers
  constexpr bool WPE_LDS = HIOCC && FIXED_POS && !PARTIAL;
  __shared__ uint32_t wpe_sh[WPE_LDS ? 4 * NWO * 64 : 1];
  const uint32_t *wpe_lds = nullptr;
  if constexpr (WPE_LDS) {
    uint32_t *mine = wpe_sh + wave * (NWO * 64);
#pragma unroll
    for (int w = 0; w < NWO; ++w) mine[w * 64 + lane] = wpe_words[w];
    wpe_lds = mine;  // read by this wave only: no barrier
  }

  int32_t rec[W];
  load_rec(p, rec);
  int32_t tokv = wte ? tok[p] : 0;
  int32_t posv = (!FIXED_POS && wpe) ? pos[p] : 0;

  while (true) {
    const bool tok_ok = wte && tokv >= 0 && (long long)tokv < q.vocab;
    bool pos_ok = true;
    if constexpr (!FIXED_POS) pos_ok = wpe && posv >= 0 && (long long)posv < q.n_pos;
    if ((wte && !tok_ok) || (!FIXED_POS && wpe && !pos_ok)) {
      if (lane == 0) atomicOr(status, SCONE_ST_BAD_TOKEN);
    }
    const int kown = rec[W - 2] & 0xFF, kfull = rec[W - 2] >> 8;
    // absent / out-of-range base rows read a row of zeros: the adds stay unconditional
    // paper mode: a matched f-gram REPLACES the token embedding (Algorithm 2), so wte is skipped
#ifdef SCONE_PROBE_NO_WTE
    const bool use_wte = false;
#else
    const bool use_wte = tok_ok && !(q.mode == SCONE_MODE_LONGEST_SUFFIX && kfull > 0);
#endif
    const uint8_t *wte_row = use_wte ? reinterpret_cast<const uint8_t *>(wte + (long long)tokv * D) : zero_row;
    const uint8_t *wpe_row = zero_row;
    if constexpr (!FIXED_POS) {
      if (pos_ok) wpe_row = reinterpret_cast<const uint8_t *>(wpe + (long long)posv * D);
    }
    uint8_t *out_row = reinterpret_cast<uint8_t *>(out + p * D);
    if constexpr (PARTIAL) {
      if (lane == 0) counts[p] = kfull;
    }

    // prefetch the record of this wave's next token (scalar loads, lgkmcnt -- not in the vmcnt queue)
    const long long pn = p + q.T;
    const bool more = pn < p_end;
    int32_t recn[W];
    int32_t tokn = 0, posn = 0;
    if (more) {
      load_rec(pn, recn);
      tokn = wte ? tok[pn] : 0;
      posn = (!FIXED_POS && wpe) ? pos[pn] : 0;
    }
#ifdef SCONE_LOCKSTEP
    __builtin_amdgcn_s_barrier();  // launch_wave only takes this build's kernel when T % 4 == 0 (equal trip counts)
#endif

#define SCONE_CASE(K)                                                                                          \
  case K:                                                                                                      \
    if constexpr (K <= NC)                                                                                     \
      embed_token<FMT, OutT, D, K, FIXED_POS, PARTIAL, WPE_LDS>(rows, scales_v, rec, q.row_begin, kfull, q.reduce,   \
                                                                wte_row, wpe_row, wpe_words, out_row, lane, wpe_lds); \
    break;
    switch (kown) {
      SCONE_CASE(0) SCONE_CASE(1) SCONE_CASE(2) SCONE_CASE(3) SCONE_CASE(4) SCONE_CASE(5) SCONE_CASE(6)
      SCONE_CASE(7) SCONE_CASE(8) SCONE_CASE(9) SCONE_CASE(10)
      default: break;
    }
#undef SCONE_CASE

    if (!more) break;
    p = pn;
#pragma unroll
    for (int j = 0; j < W; ++j) rec[j] = recn[j];
    tokv = tokn;
    posv = posn;
  }
}

// Decode-size batches (a few thousand tokens at most) are launch-bound: match and gather in ONE launch.
// One wave per token.  Lane c < NC probes candidate window c of the token (n = 1..MAXN, start ascending --
// the reference's append order is the lane order); a wavefront ballot of the hits gives K and, bit by
// bit, the compacted id list as scalars; from there on it is the same K-way body as k_embed_wave.
template <int FMT, typename OutT, int D, int MAXN>
__global__ __launch_bounds__(256) void k_embed_fused(const scone_row_store rows, const void *__restrict__ scales_v,
                                                     const scone_index_view ix, const int32_t *__restrict__ tok,
                                                     const int32_t *__restrict__ pos, const OutT *__restrict__ wte,
                                                     const OutT *__restrict__ wpe, const uint8_t *__restrict__ zero_row,
                                                     OutT *__restrict__ out, uint32_t *__restrict__ status,
                                                     const wave_params q) {
  constexpr int NC = MAXN * (MAXN + 1) / 2;
  constexpr int NWO = wave_geom<FMT, D>::EPL * (int)sizeof(OutT) / 4;
  const uint32_t lane = threadIdx.x & 63;
  const long long p = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (p >= q.BT) return;
  const int i = (int)(p % q.T);

  // ---- match: lane c probes candidate c ------------------------------------------------------------
  int32_t my_id = -1;
  if ((int)lane < NC) {
    int n, s;
    cand_ns((int)lane, n, s);
    // the paper's lookup only asks for windows that END at this token (s = n - 1) and have n >= 2
    const bool wanted = q.mode == SCONE_MODE_COVER || (s == n - 1 && n >= 2);
    if (wanted && n <= q.max_n && i - s >= 0 && i - s + n <= q.T) {
      uint32_t k[SCONE_MAX_N] = {0u, 0u, 0u, 0u};
      bool ok = true;
#pragma unroll
      for (int j = 0; j < MAXN; ++j) {
        if (j < n) {
          const int32_t v = tok[p - s + j];
          ok = ok && v >= 0;
          k[j] = (uint32_t)v;
        }
      }
      if (ok) my_id = scone_lookup_key(ix, k, n);
    }
  }
  unsigned long long hit = __ballot(my_id >= 0);
  if (q.mode == SCONE_MODE_LONGEST_SUFFIX && hit) {
    // Algorithm 2: the LONGEST f-gram ending here.  Candidates are ordered n ascending, so it is the highest hit lane.
    hit = 1ull << (63 - __builtin_clzll(hit));
    if (!((hit >> lane) & 1ull)) my_id = -1;
  }
  unsigned long long own = __ballot(my_id >= 0 && (long long)my_id >= q.row_begin && (long long)my_id < q.row_end);
  const int kfull = __popcll(hit), kown = __popcll(own);
  int32_t rec[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int l = own ? __builtin_ctzll(own) : 0;
    rec[k] = __builtin_amdgcn_readlane(my_id, l);
    own &= own - 1;
  }

  // ---- gather + reduce + combine: as k_embed_wave ----------------------------------------------------
  const int32_t tokv = wte ? tok[p] : 0;
  const int32_t posv = wpe ? (pos ? pos[p] : i) : 0;
  const bool tok_ok = wte && tokv >= 0 && (long long)tokv < q.vocab;
  const bool pos_ok = wpe && posv >= 0 && (long long)posv < q.n_pos;
  if ((wte && !tok_ok) || (wpe && !pos_ok)) {
    if (lane == 0) atomicOr(status, SCONE_ST_BAD_TOKEN);
  }
  // paper mode: a matched f-gram REPLACES the token embedding (Algorithm 2), so wte is skipped
  const bool use_wte = tok_ok && !(q.mode == SCONE_MODE_LONGEST_SUFFIX && kfull > 0);
  const uint8_t *wte_row = use_wte ? reinterpret_cast<const uint8_t *>(wte + (long long)tokv * D) : zero_row;
  const uint8_t *wpe_row = pos_ok ? reinterpret_cast<const uint8_t *>(wpe + (long long)posv * D) : zero_row;
  uint8_t *out_row = reinterpret_cast<uint8_t *>(out + p * D);
  uint32_t wpe_words[NWO];
#pragma unroll
  for (int w = 0; w < NWO; ++w) wpe_words[w] = 0u;
#define SCONE_CASE(K)                                                                                            \
  case K:                                                                                                        \
    if constexpr (K <= NC)                                                                                       \
      embed_token<FMT, OutT, D, K, false, false>(rows, scales_v, rec, q.row_begin, kfull, q.reduce, wte_row, wpe_row, \
                                                 wpe_words, out_row, lane);                                      \
    break;
  switch (kown) {
    SCONE_CASE(0) SCONE_CASE(1) SCONE_CASE(2) SCONE_CASE(3) SCONE_CASE(4) SCONE_CASE(5) SCONE_CASE(6)
    SCONE_CASE(7) SCONE_CASE(8) SCONE_CASE(9) SCONE_CASE(10)
    default: break;
  }
#undef SCONE_CASE
}

// returns -1 if the fused kernel does not apply
template <int FMT, typename OutT>
int try_launch_fused(scone_handle *h, const embed_args &a, hipStream_t s) {
  if (a.partial) return -1;
  wave_params q = {};
  q.BT = a.BT, q.T = a.T, q.max_n = a.max_n;
  q.row_begin = a.tv.row_begin, q.row_end = a.tv.row_end;
  q.vocab = a.vocab, q.n_pos = a.n_pos, q.reduce = a.reduce, q.mode = a.mode;
  scone_index_view ix;
  scone_index_view_of(h, &ix);
  const unsigned blocks = (unsigned)((a.BT + 3) / 4);
#define SCONE_FUSED(DD, NN)                                                                                       \
  hipLaunchKernelGGL((k_embed_fused<FMT, OutT, DD, NN>), dim3(blocks), dim3(256), 0, s, a.tv.st,                  \
                     (const void *)a.tv.scales, ix, a.tok, a.pos, (const OutT *)a.wte, (const OutT *)a.wpe,       \
                     (const uint8_t *)a.zero_row, (OutT *)a.out, a.status, q)
  if constexpr (wave_geom<FMT, 768>::OK) {
    if (a.tv.d == 768) {
      if (a.max_n <= 3) SCONE_FUSED(768, 3); else SCONE_FUSED(768, 4);
      SCONE_HIP(h, hipGetLastError());
      return SCONE_OK;
    }
  }
  if constexpr (wave_geom<FMT, 1024>::OK) {
    if (a.tv.d == 1024) {
      if (a.max_n <= 3) SCONE_FUSED(1024, 3); else SCONE_FUSED(1024, 4);
      SCONE_HIP(h, hipGetLastError());
      return SCONE_OK;
    }
  }
  if constexpr (wave_geom<FMT, 1280>::OK) {  // gpt2-large (configs/large_config.yaml:16)
    if (a.tv.d == 1280) {
      if (a.max_n <= 3) SCONE_FUSED(1280, 3); else SCONE_FUSED(1280, 4);
      SCONE_HIP(h, hipGetLastError());
      return SCONE_OK;
    }
  }
#undef SCONE_FUSED
  return -1;
}

// Which instantiation a launch takes: shard partial sums -> <FIXED_POS, PARTIAL>; caller-supplied position ids -> the
// position row is per token, nothing to keep; default positions -> the high-occupancy variant (position row in LDS)
// where the format has one (wave_hiocc<>: all of them by default).
template <int FMT, typename OutT, int D, int MAXN>
int launch_wave(scone_handle *h, const embed_args &a, hipStream_t s) {
  constexpr bool HI = wave_hiocc<FMT>::available;
  const bool hi = HI && !a.partial && !a.pos;
  wave_params q;
  q.BT = a.BT, q.T = a.T, q.max_n = a.max_n;
  q.row_begin = a.tv.row_begin, q.row_end = a.tv.row_end;
  q.vocab = a.vocab, q.n_pos = a.n_pos, q.reduce = a.reduce, q.mode = a.mode;
  q.B = (int)(a.BT / a.T);
  q.pos_groups = (a.T + 3) / 4;
  // Grid = a whole number of residency rounds: a workgroup is one wave per SIMD, so WAVES of them fit a
  // CU; with e.g. 4096 workgroups on 256 x 7 slots the third round is a quarter full and the chip idles
  // (measured: 3 full rounds -3.7 % kernel time vs 4096 workgroups; 1, 2 and 4 rounds within 1 % of 3).
#ifdef SCONE_WAVE_BLOCKS_FIXED
  const long long target = SCONE_WAVE_BLOCKS_FIXED;
#else
  const long long target = 3ll * scone_lookup_cus(h) * (hi ? wave_occupancy<FMT, OutT, D, MAXN, true, true>::WAVES
                                                : wave_occupancy<FMT, OutT, D, MAXN, true, false>::WAVES);
#endif
  long long chunks = (target + q.pos_groups / 2) / q.pos_groups;
  if (chunks < 1) chunks = 1;
  if (chunks > q.B) chunks = q.B;
  q.seqs_per_block = (int)((q.B + chunks - 1) / chunks);
  chunks = (q.B + q.seqs_per_block - 1) / q.seqs_per_block;
  long long blocks = chunks * q.pos_groups;
  if (!scone_grid_fits((unsigned long long)blocks + 8, 256)) return scone_fail(h, SCONE_EINVAL, "scone_embed: too many tokens for one launch");
  q.n_blocks = (unsigned)blocks;
  blocks = (blocks + 7) / 8 * 8;
  if constexpr (std::is_same<OutT, float>::value) {
    if (a.partial) {  // shard mode: fp32 partial sums + full hit counts
      hipLaunchKernelGGL((k_embed_wave<FMT, float, D, MAXN, true, true>), dim3((unsigned)blocks), dim3(256), 0, s,
                         a.tv.st, (const void *)a.tv.scales, a.ell, a.tok, (const int32_t *)nullptr,
                         (const float *)nullptr, (const float *)nullptr, (const uint8_t *)a.zero_row, a.partial,
                         a.counts, a.status, q);
      SCONE_HIP(h, hipGetLastError());
      return SCONE_OK;
    }
  }
  if (a.pos)
    hipLaunchKernelGGL((k_embed_wave<FMT, OutT, D, MAXN, false, false, false>), dim3((unsigned)blocks), dim3(256), 0, s, a.tv.st,
                       (const void *)a.tv.scales, a.ell, a.tok, a.pos, (const OutT *)a.wte, (const OutT *)a.wpe,
                       (const uint8_t *)a.zero_row, (OutT *)a.out, (int32_t *)nullptr, a.status, q);
  else
    hipLaunchKernelGGL((k_embed_wave<FMT, OutT, D, MAXN, true, false, HI>), dim3((unsigned)blocks), dim3(256), 0, s, a.tv.st,
                       (const void *)a.tv.scales, a.ell, a.tok, a.pos, (const OutT *)a.wte, (const OutT *)a.wpe,
                       (const uint8_t *)a.zero_row, (OutT *)a.out, (int32_t *)nullptr, a.status, q);
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

// ---------------------------------------------------------------------------------------------
// Any embedding dim that is a multiple of 8 (2048, 4096, ... -- the specialised kernel above covers
// 768 / 1024 / 1280).  Same wave-per-token scheme and the same arithmetic; the row is walked in UNITS
// of 8 elements (INT8 8 B, fp16 16 B, fp32 32 B, INT4 4 B per lane; 16 B of fp16 output), lane l
// taking units l, l+64, ...: every access is a contiguous run, every 64-B sector is touched whole.
// K stays a compile-time constant (switch outside the unit loop), so the K loads of a unit are issued
// back to back.
template <int FMT, typename OutT, int K, bool PARTIAL>
__device__ __forceinline__ void embed_units(const scone_row_store &rows, const void *__restrict__ scales_v,
                                            const int32_t *__restrict__ rec, long long row_begin, int d, int kfull,
                                            int reduce, const uint8_t *__restrict__ wte_row,
                                            const uint8_t *__restrict__ wpe_row, uint8_t *__restrict__ out_row, uint32_t lane) {
  constexpr int U = 8;                                  // elements per unit
  constexpr int RW = FMT == SCONE_FMT_F32 ? 8 : FMT == SCONE_FMT_F16 ? 4 : FMT == SCONE_FMT_I8 ? 2 : 1;  // row words per unit
  constexpr int OW = U * (int)sizeof(OutT) / 4;         // output words per unit
  constexpr int OPW = pack_io<OutT>::PER_WORD;
  constexpr int KK = K > 0 ? K : 1;
  const int nu = d / U;
  const uint8_t *rp[KK];
  float sc8[KK];
  long long lrs[KK];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    lrs[k] = (long long)rec[k] - row_begin;
    rp[k] = rows.row((unsigned long long)lrs[k]);
    sc8[k] = 1.0f;
    if constexpr (FMT == SCONE_FMT_I8) {
      const uint32_t w = reinterpret_cast<const uint32_t *>(scales_v)[lrs[k] >> 1];
      sc8[k] = __half2float(__ushort_as_half((unsigned short)((lrs[k] & 1) ? (w >> 16) : (w & 0xFFFFu))));
    }
  }
  float kf = (float)kfull, y = 1.0f;
  const bool do_mean = reduce == SCONE_REDUCE_MEAN && kfull > 1;
  if (do_mean) y = 1.0f / kf;
  for (int u = (int)lane; u < nu; u += 64) {
    uint32_t raw[KK][RW];
    uint32_t scw[KK];
    uint32_t bw[OW], bp[OW];
    if constexpr (!PARTIAL) {
      const uint32_t *pw = reinterpret_cast<const uint32_t *>(wte_row) + (size_t)u * OW;
      const uint32_t *pp = reinterpret_cast<const uint32_t *>(wpe_row) + (size_t)u * OW;
#pragma unroll
      for (int i = 0; i < OW; ++i) bw[i] = pw[i], bp[i] = pp[i];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t *p = reinterpret_cast<const uint32_t *>(rp[k]) + (size_t)u * RW;
#pragma unroll
      for (int i = 0; i < RW; ++i) raw[k][i] = p[i];
      scw[k] = 0;
      if constexpr (FMT == SCONE_FMT_I4)
        scw[k] = reinterpret_cast<const unsigned short *>(scales_v)[lrs[k] * (d / SCONE_I4_GROUP) + scone_i4_scale_slot((u * U) / SCONE_I4_GROUP, d)];
    }
    float acc[U];
#pragma unroll
    for (int e = 0; e < U; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float sc = sc8[k];
      if constexpr (FMT == SCONE_FMT_I4) sc = __half2float(__ushort_as_half((unsigned short)scw[k]));
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const uint32_t w = raw[k][i];
        if constexpr (FMT == SCONE_FMT_F32) {
          acc[i] += __uint_as_float(w);
        } else if constexpr (FMT == SCONE_FMT_F16) {
          acc[2 * i] += __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu)));
          acc[2 * i + 1] += __half2float(__ushort_as_half((unsigned short)(w >> 16)));
        } else if constexpr (FMT == SCONE_FMT_I8) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int q = (int)(w << (24 - 8 * b)) >> 24;
            acc[4 * i + b] = fmaf(sc, (float)q, acc[4 * i + b]);
          }
        } else {
          i4_accumulate(w, sc, acc);
        }
      }
    }
    if constexpr (PARTIAL) {
      uint32_t *po = reinterpret_cast<uint32_t *>(out_row) + (size_t)u * U;
#pragma unroll
      for (int e = 0; e < U; ++e) po[e] = __float_as_uint(acc[e]);
    } else {
      if (do_mean) {
#pragma unroll
        for (int e = 0; e < U; ++e) {  // correctly rounded x / K (Markstein, see embed_token)
          const float q0 = acc[e] * y;
          const float r = fmaf(-kf, q0, acc[e]);
          acc[e] = fmaf(r, y, q0);
        }
      }
      uint32_t *po = reinterpret_cast<uint32_t *>(out_row) + (size_t)u * OW;
#pragma unroll
      for (int w = 0; w < OW; ++w) {
        float b[OPW], c[OPW], v[OPW];
        pack_io<OutT>::unpack(bw[w], b);
        pack_io<OutT>::unpack(bp[w], c);
#pragma unroll
        for (int k = 0; k < OPW; ++k) v[k] = (b[k] + acc[w * OPW + k]) + c[k];
        __builtin_nontemporal_store(pack_io<OutT>::pack(v), po + w);
      }
    }
  }
}

template <int FMT, typename OutT, int MAXN, bool PARTIAL>
__global__ __launch_bounds__(256) void k_embed_wave_any(const scone_row_store rows, const void *__restrict__ scales_v,
                                                        const int32_t *__restrict__ ell, const int32_t *__restrict__ tok,
                                                        const int32_t *__restrict__ pos, const OutT *__restrict__ wte,
                                                        const OutT *__restrict__ wpe, const uint8_t *__restrict__ zero_row,
                                                        OutT *__restrict__ out, int32_t *__restrict__ counts,
                                                        uint32_t *__restrict__ status, const wave_params q, int d) {
  constexpr int NC = MAXN * (MAXN + 1) / 2;
  constexpr int W = MAXN <= 3 ? 8 : 16;
  const uint32_t lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pg = (int)(blockIdx.x % (unsigned)q.pos_groups);
  const int chunk = (int)(blockIdx.x / (unsigned)q.pos_groups);
  const int i = pg * 4 + wave;
  if (i >= q.T) return;
  const int b0 = chunk * q.seqs_per_block;
  int b1 = b0 + q.seqs_per_block;
  if (b1 > q.B) b1 = q.B;
  for (int b = b0; b < b1; ++b) {
    const long long p = (long long)b * q.T + i;
    int32_t rec[W];
#pragma unroll
    for (int j = 0; j < W; ++j) rec[j] = ell[p * W + j];
    const int kown = rec[W - 2] & 0xFF, kfull = rec[W - 2] >> 8;
    const uint8_t *wte_row = zero_row, *wpe_row = zero_row;
    if constexpr (PARTIAL) {
      if (lane == 0) counts[p] = kfull;
    } else {
      const int32_t tokv = wte ? tok[p] : 0;
      const int32_t posv = wpe ? (pos ? pos[p] : i) : 0;
      const bool tok_ok = wte && tokv >= 0 && (long long)tokv < q.vocab;
      const bool pos_ok = wpe && posv >= 0 && (long long)posv < q.n_pos;
      if ((wte && !tok_ok) || (wpe && !pos_ok)) {
        if (lane == 0) atomicOr(status, SCONE_ST_BAD_TOKEN);
      }
      if (tok_ok && !(q.mode == SCONE_MODE_LONGEST_SUFFIX && kfull > 0))
        wte_row = reinterpret_cast<const uint8_t *>(wte + (long long)tokv * d);
      if (pos_ok) wpe_row = reinterpret_cast<const uint8_t *>(wpe + (long long)posv * d);
    }
    uint8_t *out_row = reinterpret_cast<uint8_t *>(out + p * d);
#define SCONE_CASE(K)                                                                                              \
  case K:                                                                                                          \
    if constexpr (K <= NC)                                                                                         \
      embed_units<FMT, OutT, K, PARTIAL>(rows, scales_v, rec, q.row_begin, d, kfull, q.reduce, wte_row, wpe_row, out_row, \
                                         lane);                                                                    \
    break;
    switch (kown) {
      SCONE_CASE(0) SCONE_CASE(1) SCONE_CASE(2) SCONE_CASE(3) SCONE_CASE(4) SCONE_CASE(5) SCONE_CASE(6)
      SCONE_CASE(7) SCONE_CASE(8) SCONE_CASE(9) SCONE_CASE(10)
      default: break;
    }
#undef SCONE_CASE
  }
}

template <int FMT, typename OutT, int MAXN>
int launch_wave_any(scone_handle *h, const embed_args &a, hipStream_t s) {
  wave_params q;
  q.BT = a.BT, q.T = a.T, q.max_n = a.max_n;
  q.row_begin = a.tv.row_begin, q.row_end = a.tv.row_end;
  q.vocab = a.vocab, q.n_pos = a.n_pos, q.reduce = a.reduce, q.mode = a.mode;
  q.B = (int)(a.BT / a.T);
  q.pos_groups = (a.T + 3) / 4;
  long long chunks = SCONE_WAVE_BLOCKS / q.pos_groups;
  if (chunks < 1) chunks = 1;
  if (chunks > q.B) chunks = q.B;
  q.seqs_per_block = (int)((q.B + chunks - 1) / chunks);
  chunks = (q.B + q.seqs_per_block - 1) / q.seqs_per_block;
  const long long blocks = chunks * q.pos_groups;
  if (!scone_grid_fits((unsigned long long)blocks, 256)) return scone_fail(h, SCONE_EINVAL, "scone_embed: too many tokens for one launch");
  if constexpr (std::is_same<OutT, float>::value) {
    if (a.partial) {
      hipLaunchKernelGGL((k_embed_wave_any<FMT, float, MAXN, true>), dim3((unsigned)blocks), dim3(256), 0, s, a.tv.st,
                         (const void *)a.tv.scales, a.ell, a.tok, (const int32_t *)nullptr, (const float *)nullptr,
                         (const float *)nullptr, (const uint8_t *)a.zero_row, a.partial, a.counts, a.status, q, a.tv.d);
      SCONE_HIP(h, hipGetLastError());
      return SCONE_OK;
    }
  }
  hipLaunchKernelGGL((k_embed_wave_any<FMT, OutT, MAXN, false>), dim3((unsigned)blocks), dim3(256), 0, s, a.tv.st,
                     (const void *)a.tv.scales, a.ell, a.tok, a.pos, (const OutT *)a.wte, (const OutT *)a.wpe,
                     (const uint8_t *)a.zero_row, (OutT *)a.out, (int32_t *)nullptr, a.status, q, a.tv.d);
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

// ---------------------------------------------------------------------------------------------
// CSR source (scone_gather_reduce: caller-supplied per-token id lists + optional base rows).  One wave per token,
// grid-stride; a list of up to 10 owned ids goes through the same K-way static bodies as k_embed_wave (the base row
// takes the place of the wte row), a longer one -- only a caller can build it, the reference's lists have at most
// max_n (max_n + 1) / 2 entries -- through an ordered loop, one row at a time.  (The lane-group kernel k_embed that
// served this entry point before ran the 1M-token headline lists in 1.65 ms; this one in 0.7 ms.)
template <int FMT, typename OutT, int D>
__device__ __forceinline__ void embed_token_long(const scone_row_store &rows, const void *__restrict__ scales_v,
                                                 const int32_t *__restrict__ ids, int K, long long row_begin, long long row_end,
                                                 int reduce, const uint8_t *__restrict__ base_row, uint8_t *__restrict__ out_row,
                                                 uint32_t lane) {
  using G = wave_geom<FMT, D>;
  constexpr int EPL = G::EPL, NSEG = G::NSEG;
  constexpr int NWO = EPL * (int)sizeof(OutT) / 4;
  constexpr int OPW = pack_io<OutT>::PER_WORD;
  uint32_t bw[NWO];
  ld_out_row<FMT, OutT, D>(base_row, lane, bw);
  float acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
  for (int k = 0; k < K; ++k) {
    const long long id = ids[k];
    if (id < row_begin || id >= row_end) continue;  // another shard's row (or an id reported by the caller of this function)
    const long long lr = id - row_begin;
    const uint8_t *rp = rows.row((unsigned long long)lr);
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
      const uint32_t *p = reinterpret_cast<const uint32_t *>(rp + G::seg_first(s) / 4 * G::BPE4 +
                                                              lane * (uint32_t)(G::seg_elems(s) * G::BPE4 / 4));
      float sc = 1.0f;
      if constexpr (FMT == SCONE_FMT_I8) {
        sc = __half2float(reinterpret_cast<const __half *>(scales_v)[lr]);
      } else if constexpr (FMT == SCONE_FMT_I4) {
        sc = __half2float(reinterpret_cast<const __half *>(scales_v)[lr * (D / SCONE_I4_GROUP) +
                                                                      scone_i4_scale_slot((G::seg_first(s) + lane * G::seg_elems(s)) / SCONE_I4_GROUP, D)]);
      }
#pragma unroll
      for (int i = 0; i < G::seg_row_words(s); ++i) {
        const uint32_t w = p[i];
        if constexpr (FMT == SCONE_FMT_F32) {
          acc[G::seg_acc(s) + i] += __uint_as_float(w);
        } else if constexpr (FMT == SCONE_FMT_F16) {
          acc[G::seg_acc(s) + 2 * i] += __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu)));
          acc[G::seg_acc(s) + 2 * i + 1] += __half2float(__ushort_as_half((unsigned short)(w >> 16)));
        } else if constexpr (FMT == SCONE_FMT_I8) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int q = (int)(w << (24 - 8 * b)) >> 24;
            acc[G::seg_acc(s) + 4 * i + b] = fmaf(sc, (float)q, acc[G::seg_acc(s) + 4 * i + b]);
          }
        } else {
          i4_accumulate(w, sc, &acc[G::seg_acc(s) + 8 * i]);
        }
      }
    }
  }
  if (reduce == SCONE_REDUCE_MEAN && K > 1) {  // correctly rounded x / K (Markstein, see embed_token)
    const float kf = (float)K;
    const float y = 1.0f / kf;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const float q0 = acc[e] * y;
      const float r = fmaf(-kf, q0, acc[e]);
      acc[e] = fmaf(r, y, q0);
    }
  }
  uint32_t ow[NWO];
#pragma unroll
  for (int w = 0; w < NWO; ++w) {
    float b[OPW], v[OPW];
    pack_io<OutT>::unpack(bw[w], b);
#pragma unroll
    for (int k = 0; k < OPW; ++k) v[k] = b[k] + acc[w * OPW + k];
    ow[w] = pack_io<OutT>::pack(v);
  }
#ifdef SCONE_PROBE_NO_STORE
  if (reduce != 0x5C0E) return;
#endif
  st_out_row<FMT, OutT, D>(out_row, lane, ow);
}

template <int FMT, typename OutT, int D>
__global__ __launch_bounds__(256) void k_embed_csr_wave(const scone_row_store rows, const void *__restrict__ scales_v,
                                                        const int32_t *__restrict__ offsets, const int32_t *__restrict__ ids,
                                                        long long ntok, long long row_begin, long long row_end, long long n_rows,
                                                        const OutT *__restrict__ base, const uint8_t *__restrict__ zero_row,
                                                        OutT *__restrict__ out, int reduce, uint32_t *__restrict__ status) {
  constexpr int NCMAX = SCONE_MAX_CAND;
  constexpr int NWO = wave_geom<FMT, D>::EPL * (int)sizeof(OutT) / 4;
  const uint32_t lane = threadIdx.x & 63;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  uint32_t wpe_words[NWO];
#pragma unroll
  for (int w = 0; w < NWO; ++w) wpe_words[w] = 0u;
  for (long long t = (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); t < ntok;
       t += nwaves) {
    const int off0 = offsets[t];
    const int K = offsets[t + 1] - off0;
    const uint8_t *base_row = base ? reinterpret_cast<const uint8_t *>(base + t * D) : zero_row;
    uint8_t *out_row = reinterpret_cast<uint8_t *>(out + t * D);
    int32_t rec[NCMAX];
    int kown = 0;
    bool bad = false;
    // owned ids compacted in list order (static indexing: the registers stay scalars)
#pragma unroll
    for (int j = 0; j < NCMAX; ++j) rec[j] = 0;
    const int kscan = K < NCMAX ? K : NCMAX;
#pragma unroll
    for (int j = 0; j < NCMAX; ++j) {
      if (j < kscan) {
        const long long id = ids[off0 + j];
        bad = bad || id < 0 || id >= n_rows;
        if (id >= row_begin && id < row_end) {
#pragma unroll
          for (int jj = 0; jj < NCMAX; ++jj)
            if (jj == kown) rec[jj] = (int32_t)id;
          ++kown;
        }
      }
    }
    if (K > NCMAX) {
      for (int j = NCMAX; j < K; ++j) {
        const long long id = ids[off0 + j];
        bad = bad || id < 0 || id >= n_rows;
      }
      if (bad && lane == 0) atomicOr(status, SCONE_ST_BAD_ID);
      embed_token_long<FMT, OutT, D>(rows, scales_v, ids + off0, K, row_begin, row_end, reduce, base_row, out_row, lane);
      continue;
    }
    if (bad && lane == 0) atomicOr(status, SCONE_ST_BAD_ID);
#define SCONE_CASE(KK)                                                                                                 \
  case KK:                                                                                                             \
    embed_token<FMT, OutT, D, KK, false, false>(rows, scales_v, rec, row_begin, K, reduce, base_row, zero_row, wpe_words, \
                                                out_row, lane);                                                       \
    break;
    switch (kown) {
      SCONE_CASE(0) SCONE_CASE(1) SCONE_CASE(2) SCONE_CASE(3) SCONE_CASE(4) SCONE_CASE(5) SCONE_CASE(6)
      SCONE_CASE(7) SCONE_CASE(8) SCONE_CASE(9) SCONE_CASE(10)
      default: break;
    }
#undef SCONE_CASE
  }
}

// returns -1 when d is not one of the specialised dims
template <int FMT, typename OutT>
int try_launch_csr_wave(scone_handle *h, const embed_args &a, hipStream_t s) {
  long long blocks = (a.ntok + 3) / 4;
  const long long cap = 8ll * h->n_cus * 4;  // a few residency rounds of 4-wave workgroups
  if (blocks > cap) blocks = cap;
#define SCONE_CSR(DD)                                                                                                     \
  hipLaunchKernelGGL((k_embed_csr_wave<FMT, OutT, DD>), dim3((unsigned)blocks), dim3(256), 0, s, a.tv.st,               \
                     (const void *)a.tv.scales, a.offsets, a.ids, a.ntok, a.tv.row_begin, a.tv.row_end, a.tv.n_rows,      \
                     (const OutT *)a.base, (const uint8_t *)a.zero_row, (OutT *)a.out, a.reduce, a.status)
  if constexpr (wave_geom<FMT, 768>::OK) {
    if (a.tv.d == 768) {
      SCONE_CSR(768);
      SCONE_HIP(h, hipGetLastError());
      return SCONE_OK;
    }
  }
  if constexpr (wave_geom<FMT, 1024>::OK) {
    if (a.tv.d == 1024) {
      SCONE_CSR(1024);
      SCONE_HIP(h, hipGetLastError());
      return SCONE_OK;
    }
  }
  if constexpr (wave_geom<FMT, 1280>::OK) {
    if (a.tv.d == 1280) {
      SCONE_CSR(1280);
      SCONE_HIP(h, hipGetLastError());
      return SCONE_OK;
    }
  }
#undef SCONE_CSR
  return -1;
}

// Shard mode, second half: out[t] = cast((wte[tok] + sum[t] / K_t) + wpe[pos]) for the tokens of one
// slice.  One wave per token, same lane map; sums are read once (fp32), output streamed.
template <typename OutT, int D>
__global__ __launch_bounds__(256) void k_finalize_wave(const float *__restrict__ sums, const int32_t *__restrict__ counts,
                                                       const int32_t *__restrict__ tok, const int32_t *__restrict__ pos,
                                                       const OutT *__restrict__ wte, const OutT *__restrict__ wpe,
                                                       const uint8_t *__restrict__ zero_row, OutT *__restrict__ out,
                                                       uint32_t *__restrict__ status, long long tok_begin, long long ntok,
                                                       int T, long long vocab, long long n_pos, int reduce, int mode) {
  constexpr int FMT = SCONE_FMT_F32;  // lane map of an fp32 row
  using G = wave_geom<FMT, D>;
  constexpr int EPL = G::EPL, NWO = EPL * (int)sizeof(OutT) / 4, OPW = pack_io<OutT>::PER_WORD;
  const uint32_t lane = threadIdx.x & 63;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  long long t = (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (; t < ntok; t += nwaves) {
    const long long p = tok_begin + t;
    const int kfull = counts[t];
    const int32_t tokv = wte ? tok[p] : 0;
    const int32_t posv = wpe ? (pos ? pos[p] : (int32_t)(p % T)) : 0;
    const bool tok_ok = wte && tokv >= 0 && (long long)tokv < vocab;
    const bool pos_ok = wpe && posv >= 0 && (long long)posv < n_pos;
    if ((wte && !tok_ok) || (wpe && !pos_ok)) {
      if (lane == 0) atomicOr(status, SCONE_ST_BAD_TOKEN);
    }
    const bool use_wte = tok_ok && !(mode == SCONE_MODE_LONGEST_SUFFIX && kfull > 0);
    const uint8_t *wte_row = use_wte ? reinterpret_cast<const uint8_t *>(wte + (long long)tokv * D) : zero_row;
    const uint8_t *wpe_row = pos_ok ? reinterpret_cast<const uint8_t *>(wpe + (long long)posv * D) : zero_row;
    uint32_t bw[NWO], bp[NWO], sw[EPL];
    ld_out_row<FMT, OutT, D>(wte_row, lane, bw);
    ld_out_row<FMT, OutT, D>(wpe_row, lane, bp);
    ld_out_row<FMT, float, D>(reinterpret_cast<const uint8_t *>(sums + t * D), lane, sw);
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = __uint_as_float(sw[e]);
    if (reduce == SCONE_REDUCE_MEAN && kfull > 1) {
      const float kf = (float)kfull;
      const float y = 1.0f / kf;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {  // correctly rounded x / K (see embed_token)
        const float q0 = acc[e] * y;
        const float r = fmaf(-kf, q0, acc[e]);
        acc[e] = fmaf(r, y, q0);
      }
    }
    uint32_t ow[NWO];
#pragma unroll
    for (int w = 0; w < NWO; ++w) {
      float b[OPW], c[OPW], v[OPW];
      pack_io<OutT>::unpack(bw[w], b);
      pack_io<OutT>::unpack(bp[w], c);
#pragma unroll
      for (int k = 0; k < OPW; ++k) v[k] = (b[k] + acc[w * OPW + k]) + c[k];
      ow[w] = pack_io<OutT>::pack(v);
    }
    st_out_row<FMT, OutT, D>(reinterpret_cast<uint8_t *>(out + t * D), lane, ow);
  }
}

// returns -1 when d is not covered
template <typename OutT>
int try_launch_finalize_wave(scone_handle *h, const embed_args &a, hipStream_t s) {
  if (a.tv.d != 768 && a.tv.d != 1024 && a.tv.d != 1280) return -1;
  long long blocks = (a.ntok + 3) / 4;
  if (blocks > 8192) blocks = 8192;
#define SCONE_FIN(DD)                                                                                              \
  hipLaunchKernelGGL((k_finalize_wave<OutT, DD>), dim3((unsigned)blocks), dim3(256), 0, s, a.sums, a.counts, a.tok, \
                     a.pos, (const OutT *)a.wte, (const OutT *)a.wpe, (const uint8_t *)a.zero_row, (OutT *)a.out,   \
                     a.status, a.tok_begin, a.ntok, a.T, a.vocab, a.n_pos, a.reduce, a.mode)
  if (a.tv.d == 768) SCONE_FIN(768);
  else if (a.tv.d == 1024) SCONE_FIN(1024);
  else SCONE_FIN(1280);
#undef SCONE_FIN
  SCONE_HIP(h, hipGetLastError());
  return SCONE_OK;
}

// returns -1 when the wave kernel does not cover this (format, d, max_n) -> caller falls back to k_embed
template <int FMT, typename OutT>
int try_launch_wave_on(scone_handle *h, const embed_args &a, hipStream_t s) {
  if constexpr (wave_geom<FMT, 768>::OK) {
    if (a.tv.d == 768) return a.max_n <= 3 ? launch_wave<FMT, OutT, 768, 3>(h, a, s) : launch_wave<FMT, OutT, 768, 4>(h, a, s);
  }
  if constexpr (wave_geom<FMT, 1024>::OK) {
    if (a.tv.d == 1024) return a.max_n <= 3 ? launch_wave<FMT, OutT, 1024, 3>(h, a, s) : launch_wave<FMT, OutT, 1024, 4>(h, a, s);
  }
  if constexpr (wave_geom<FMT, 1280>::OK) {  // gpt2-large (configs/large_config.yaml:16)
    if (a.tv.d == 1280) return a.max_n <= 3 ? launch_wave<FMT, OutT, 1280, 3>(h, a, s) : launch_wave<FMT, OutT, 1280, 4>(h, a, s);
  }
  return a.max_n <= 3 ? launch_wave_any<FMT, OutT, 3>(h, a, s) : launch_wave_any<FMT, OutT, 4>(h, a, s);  // any other dim % 8 == 0
}

// With a CU reserve (scone_set_cu_reserve) the launch goes to the handle's masked stream, tied into `s` by two events.
template <int FMT, typename OutT>
int try_launch_wave(scone_handle *h, const embed_args &a, hipStream_t s) {
  if (!a.ell || a.tv.d % 8) return -1;
  hipStream_t ls = s;
  int rc = scone_lookup_enter(h, s, &ls);
  if (rc) return rc;
  rc = try_launch_wave_on<FMT, OutT>(h, a, ls);
  const int rc2 = scone_lookup_leave(h, s, ls);
  return rc ? rc : rc2;
}

template <int FMT>
int launch_table_fmt(scone_handle *h, const embed_args &a, int src, int mode, int out_dtype, hipStream_t s) {
  if (src == SRC_CSR) {
    int rc = -1;
    if (a.zero_row) {
      switch (out_dtype) {
        case SCONE_DT_F32: rc = try_launch_csr_wave<FMT, float>(h, a, s); break;
        case SCONE_DT_F16: rc = try_launch_csr_wave<FMT, __half>(h, a, s); break;
        case SCONE_DT_BF16: rc = try_launch_csr_wave<FMT, __hip_bfloat16>(h, a, s); break;
        default: return scone_fail(h, SCONE_EINVAL, "unknown out_dtype");
      }
    }
    return rc != -1 ? rc : launch_dtype<FMT, SRC_CSR, MODE_FULL>(h, a, out_dtype, s);
  }
  if (mode == MODE_PARTIAL) {
    const int rc = try_launch_wave<FMT, float>(h, a, s);
    return rc != -1 ? rc : launch_dtype<FMT, SRC_HITS, MODE_PARTIAL>(h, a, out_dtype, s);
  }
  if (mode == MODE_FINALIZE) {
    if constexpr (FMT == SCONE_FMT_F32) {
      int rc = -1;
      switch (out_dtype) {
        case SCONE_DT_F32: rc = try_launch_finalize_wave<float>(h, a, s); break;
        case SCONE_DT_F16: rc = try_launch_finalize_wave<__half>(h, a, s); break;
        case SCONE_DT_BF16: rc = try_launch_finalize_wave<__hip_bfloat16>(h, a, s); break;
        default: return scone_fail(h, SCONE_EINVAL, "unknown out_dtype");
      }
      return rc != -1 ? rc : launch_dtype<FMT, SRC_HITS, MODE_FINALIZE>(h, a, out_dtype, s);
    } else {
      return scone_fail(h, SCONE_EINVAL, "finalize runs on fp32 sums");
    }
  }
  // decode-size batch: match + gather in one launch (scone_embed sets a.fused and skips the match kernel)
  if (a.fused) {
    switch (out_dtype) {
      case SCONE_DT_F32: return try_launch_fused<FMT, float>(h, a, s);
      case SCONE_DT_F16: return try_launch_fused<FMT, __half>(h, a, s);
      case SCONE_DT_BF16: return try_launch_fused<FMT, __hip_bfloat16>(h, a, s);
      default: return scone_fail(h, SCONE_EINVAL, "unknown out_dtype");
    }
  }
  // fused full lookup: wave-per-token kernel where it applies, lane-group kernel otherwise
  int rc = -1;
  switch (out_dtype) {
    case SCONE_DT_F32: rc = try_launch_wave<FMT, float>(h, a, s); break;
    case SCONE_DT_F16: rc = try_launch_wave<FMT, __half>(h, a, s); break;
    case SCONE_DT_BF16: rc = try_launch_wave<FMT, __hip_bfloat16>(h, a, s); break;
    default: return scone_fail(h, SCONE_EINVAL, "unknown out_dtype");
  }
  if (rc != -1) return rc;
  return launch_dtype<FMT, SRC_HITS, MODE_FULL>(h, a, out_dtype, s);
}

}  // namespace scone_gather
