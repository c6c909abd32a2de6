// Round 6 probe: an output buffer whose PHYSICAL chunks are mapped into one contiguous virtual range in SHUFFLED order
// (HIP virtual memory management: hipMemCreate per chunk, hipMemAddressReserve, hipMemMap).  Question: the lookup kernel's time
// follows the physical placement of the buffer it writes (profiles/r06m); the physically-contiguous allocations were the
// slowest group and the "lucky" fast ones looked like allocations that filled scattered holes -- is a deliberately scattered
// buffer deterministically fast?   tools/out_scatter_probe.py drives it.
// build: hipcc -O2 -std=c++17 -fPIC -shared --offload-arch=gfx950 scatter_alloc.hip -o libscatter_alloc.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

namespace {
struct region {
  void *base;
  size_t size;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::vector<region> g_regions;
}  // namespace

// bytes: size of the buffer; chunk: physical piece size (rounded up to the allocation granularity); mode 0 = chunks mapped in
// creation order, 1 = shuffled (seed), 2 = reversed.  Returns the device pointer or null.
extern "C" void *scatter_alloc(size_t bytes, size_t chunk, int mode, unsigned seed, int device) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) return nullptr;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = (bytes + chunk - 1) / chunk;
  region r;
  r.size = n * chunk;
  if (hipMemAddressReserve(&r.base, r.size, 0, nullptr, 0) != hipSuccess) return nullptr;
  r.handles.resize(n);
  for (size_t i = 0; i < n; ++i)
    if (hipMemCreate(&r.handles[i], chunk, &prop, 0) != hipSuccess) {
      fprintf(stderr, "scatter_alloc: hipMemCreate failed at chunk %zu of %zu\n", i, n);
      return nullptr;
    }
  std::vector<size_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  if (mode == 1) {
    std::mt19937_64 g(seed);
    std::shuffle(order.begin(), order.end(), g);
  } else if (mode == 2) {
    std::reverse(order.begin(), order.end());
  }
  for (size_t i = 0; i < n; ++i)
    if (hipMemMap(static_cast<char *>(r.base) + i * chunk, chunk, 0, r.handles[order[i]], 0) != hipSuccess) {
      fprintf(stderr, "scatter_alloc: hipMemMap failed at chunk %zu\n", i);
      return nullptr;
    }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(r.base, r.size, &acc, 1) != hipSuccess) return nullptr;
  g_regions.push_back(r);
  return r.base;
}

extern "C" void scatter_free(void *p) {
  for (size_t k = 0; k < g_regions.size(); ++k)
    if (g_regions[k].base == p) {
      (void)hipDeviceSynchronize();
      (void)hipMemUnmap(p, g_regions[k].size);
      for (auto h : g_regions[k].handles) (void)hipMemRelease(h);
      (void)hipMemAddressFree(p, g_regions[k].size);
      g_regions.erase(g_regions.begin() + k);
      return;
    }
}

extern "C" size_t scatter_granularity(int device) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  return gran;
}
