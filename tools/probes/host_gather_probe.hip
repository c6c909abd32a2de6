// Two unknowns of a copy-engine fill for the cold-row cache ("next (5)"):
//  (1) how fast can T host threads gather N scattered 528-byte rows of a pinned table into a pinned staging buffer,
//  (2) does hipStreamWaitValue64 / hipStreamWriteValue64 order two streams through a flag in signal memory here.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -pthread host_gather_probe.hip -o host_gather_probe
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));                          \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

__global__ void k_mark(unsigned long long *p, unsigned long long v) { *p = v; }
__global__ void k_read(const unsigned long long *flag, unsigned long long *seen) { *seen = *flag; }

int main(int argc, char **argv) {
  const size_t RB = 528, table_rows = (argc > 1 ? atoll(argv[1]) : 16000000ll);  // 8.4 GB of pinned rows by default
  const int n = argc > 2 ? atoi(argv[2]) : 11000;
  uint8_t *table = nullptr, *staging = nullptr, *dev = nullptr;
  CK(hipHostMalloc((void **)&table, table_rows * RB, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&staging, (size_t)n * RB, hipHostMallocDefault));
  CK(hipMalloc((void **)&dev, (size_t)n * RB));
  {  // touch every page, in parallel
    std::vector<std::thread> th;
    for (int t = 0; t < 16; ++t)
      th.emplace_back([&, t] {
        const size_t per = table_rows * RB / 16;
        memset(table + t * per, t + 1, per);
      });
    for (auto &x : th) x.join();
  }
  std::mt19937_64 rng(7);
  printf("{\"table_GB\": %.1f, \"rows_per_chunk\": %d, \"row_bytes\": %zu, \"gather_us\": {", table_rows * RB / 1e9, n, RB);
  bool first = true;
  for (int T : {1, 2, 4, 8, 12, 16}) {
    double best = 1e30, sum = 0;
    const int reps = 12;
    for (int rep = 0; rep < reps; ++rep) {
      std::vector<uint32_t> ids(n);
      for (auto &v : ids) v = (uint32_t)(rng() % table_rows);
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          const int a = (int)((long long)n * t / T), b = (int)((long long)n * (t + 1) / T);
          for (int i = a; i < b; ++i) {
            if (i + 8 < b) __builtin_prefetch(table + (size_t)ids[i + 8] * RB);
            memcpy(staging + (size_t)i * RB, table + (size_t)ids[i] * RB, RB);
          }
        });
      for (auto &x : th) x.join();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      best = us < best ? us : best;
      sum += us;
    }
    printf("%s\"%d\": {\"best\": %.1f, \"mean\": %.1f}", first ? "" : ", ", T, best, sum / reps);
    first = false;
  }
  printf("}, ");
  // H2D of the staging buffer by the copy engine
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0, best = 1e30f;
  for (int rep = 0; rep < 10; ++rep) {
    CK(hipEventRecord(e0, s1));
    CK(hipMemcpyAsync(dev, staging, (size_t)n * RB, hipMemcpyHostToDevice, s1));
    CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("\"h2d_us\": %.1f, ", best * 1e3);
  // stream wait / write value through signal memory
  int can = 0;
  (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("\"can_use_stream_wait_value\": %d, ", can);
  unsigned long long *flag = nullptr, *seen = nullptr, *mark = nullptr;
  hipError_t e = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
  if (e != hipSuccess) {
    printf("\"signal_memory\": \"%s\"}\n", hipGetErrorString(e));
    return 0;
  }
  CK(hipHostMalloc((void **)&seen, 8, hipHostMallocDefault));
  CK(hipMalloc((void **)&mark, 8));
  *flag = 0;
  *seen = 99;
  // s1: wait(flag >= 5) -> read flag into seen.   s2: (later) write flag = 5.
  e = hipStreamWaitValue64(s1, flag, 5, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
  if (e != hipSuccess) {
    printf("\"wait_value\": \"%s\"}\n", hipGetErrorString(e));
    return 0;
  }
  hipLaunchKernelGGL(k_read, dim3(1), dim3(1), 0, s1, flag, seen);
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  const bool early = hipStreamQuery(s1) == hipSuccess;  // must still be waiting
  e = hipStreamWriteValue64(s2, flag, 5, 0);
  if (e != hipSuccess) {
    printf("\"write_value\": \"%s\"}\n", hipGetErrorString(e));
    return 0;
  }
  CK(hipStreamSynchronize(s2));
  CK(hipStreamSynchronize(s1));
  printf("\"wait_released_early\": %s, \"value_seen_after_wait\": %llu}\n", early ? "true" : "false", *seen);
  return 0;
}
