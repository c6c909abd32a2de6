// Stand-in for RCCL's send / recv kernels, for pricing what a transfer costs the sharded step while it is IN FLIGHT on one
// GPU (tools/c5_rank0_step.py --transport-standin).  Not part of the product: one kernel launch = one RCCL "group" -- a few
// channels per peer, one workgroup of 256-512 threads each, every workgroup streaming its share of one segment with 16-B
// loads and stores, several per thread in flight (RCCL's copy loops are unrolled the same way).  Segments are (src, dst,
// bytes) triples: a peer's rows / scales / hash fragment into this rank's receive buffers ("recv": on real links the bytes
// arrive over xGMI and, in RCCL's simple protocol, are then copied FIFO -> user buffer by a local workgroup: local read +
// local write, which is what is emulated) and this rank's packed columns into a peer's buffer ("send": local read, the
// write lands in another device buffer here where a real send writes over xGMI).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define STANDIN_MAX_SEG 64

struct standin_segs {
  const uint4 *src[STANDIN_MAX_SEG];
  uint4 *dst[STANDIN_MAX_SEG];
  unsigned long long n_vec[STANDIN_MAX_SEG];  // 16-byte units
  int first_block[STANDIN_MAX_SEG + 1];       // workgroups [first_block[i], first_block[i + 1]) serve segment i
  int n_seg;
  uint4 *sink;                                // never written in practice (read-only segments)
};

template <int UNROLL>
__global__ __launch_bounds__(512) void k_standin(const standin_segs s) {
  int seg = 0;
  while (seg + 1 < s.n_seg && (int)blockIdx.x >= s.first_block[seg + 1]) ++seg;
  const int nb = s.first_block[seg + 1] - s.first_block[seg];
  const int b = (int)blockIdx.x - s.first_block[seg];
  const unsigned long long n = s.n_vec[seg];
  const unsigned long long per = (n + nb - 1) / nb;
  const unsigned long long a = (unsigned long long)b * per, e = a + per < n ? a + per : n;
  const uint4 *__restrict__ src = s.src[seg];
  uint4 *__restrict__ dst = s.dst[seg];
  // whole tiles: UNROLL unconditional 16-B loads per thread back to back, then the stores (a guarded load is sunk next to its
  // use by the compiler: eight sequential latencies per iteration -- the first version of this kernel moved 3 GB/s per
  // workgroup); the tail goes one vector at a time
  const unsigned long long tile = (unsigned long long)blockDim.x * UNROLL;
  unsigned long long i = a;
  if (!dst) {  // read-only segment: a send whose writes leave the device (xGMI) -- only the HBM reads stay here
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i + tile <= e; i += tile) {
      uint4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = src[i + threadIdx.x + (unsigned long long)u * blockDim.x];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc.x ^= v[u].x, acc.y ^= v[u].y, acc.z ^= v[u].z, acc.w ^= v[u].w;
    }
    for (i += threadIdx.x; i < e; i += blockDim.x) acc.x ^= src[i].x;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u && s.sink) s.sink[blockIdx.x] = acc;  // (keeps the loads alive)
    return;
  }
  for (; i + tile <= e; i += tile) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = src[i + threadIdx.x + (unsigned long long)u * blockDim.x];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) dst[i + threadIdx.x + (unsigned long long)u * blockDim.x] = v[u];
  }
  for (i += threadIdx.x; i < e; i += blockDim.x) dst[i] = src[i];
}

// src / dst / bytes: n_seg entries (bytes multiples of 16); channels[i]: workgroups for segment i.  Returns 0 or a hipError_t.
extern "C" int standin_launch(int n_seg, const void *const *src, void *const *dst, const unsigned long long *bytes,
                              const int *channels, int threads, void *stream) {
  if (n_seg < 1 || n_seg > STANDIN_MAX_SEG || (threads != 256 && threads != 512)) return -1;
  standin_segs s = {};
  s.n_seg = n_seg;
  int blocks = 0;
  for (int i = 0; i < n_seg; ++i) {
    if (bytes[i] % 16 || channels[i] < 1) return -1;
    s.src[i] = (const uint4 *)src[i], s.dst[i] = (uint4 *)dst[i], s.n_vec[i] = bytes[i] / 16;
    s.first_block[i] = blocks;
    blocks += channels[i];
  }
  s.first_block[n_seg] = blocks;
  hipLaunchKernelGGL((k_standin<8>), dim3(blocks), dim3(threads), 0, (hipStream_t)stream, s);
  return (int)hipGetLastError();
}

// ---- which compute units a CU mask enables (tools/cu_mask_probe.py): every workgroup reports the XCD and the hardware id of
// the CU it ran on, after spinning long enough for the dispatcher to have to use every enabled CU
__global__ __launch_bounds__(256) void k_whoami(uint32_t *__restrict__ out, int spin) {
  uint32_t xcc = 0, hwid = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
}

// mask: `words` x 32 bits (NULL = an ordinary stream); h_out: 2 x n_blocks words.  Returns 0 or a hipError_t.
extern "C" int standin_mask_probe(const uint32_t *mask, int words, int n_blocks, int spin_ticks, uint32_t *h_out) {
  hipStream_t s = nullptr;
  hipError_t e = mask ? hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) : hipStreamCreate(&s);
  if (e != hipSuccess) return (int)e;
  uint32_t *d = nullptr;
  e = hipMalloc(&d, (size_t)n_blocks * 8);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_whoami, dim3(n_blocks), dim3(256), 0, s, d, spin_ticks);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(h_out, d, (size_t)n_blocks * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
  }
  (void)hipStreamDestroy(s);
  return (int)e;
}
