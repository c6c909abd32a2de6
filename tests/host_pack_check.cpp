// TEST INFRASTRUCTURE -- the host-side key packing / probe-sequence / scale-slot helpers of scone_amd/csrc/scone_common.h
// (the functions the index build, the match kernels and the table kernels share between host and device) compiled for the
// HOST ONLY under AddressSanitizer + UBSan and checked for the properties the index relies on.  CPU only; built and run by
// tests/test_host_logic.py::test_host_side_packing_under_sanitizers (hipcc --cuda-host-only -fsanitize=address,undefined).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#include "../scone_amd/csrc/scone_common.h"

static uint32_t rng_state = 12345u;
static uint32_t rnd() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return rng_state >> 8;
}

int main() {
  int bad = 0;
  // 1. distinct f-grams -> distinct packed keys (exact-key index: a probe can never return a wrong id), for both layouts
  for (int max_n = 1; max_n <= 4; ++max_n) {
    std::map<std::tuple<unsigned long long, uint32_t>, std::vector<uint32_t>> seen;
    const uint32_t vocab = max_n == 4 ? 0xFFFFFEu : 0xFFFFFFFEu;
    for (int it = 0; it < 200000; ++it) {
      uint32_t t[4] = {0, 0, 0, 0};
      const int n = 1 + (int)(rnd() % (uint32_t)max_n);
      for (int k = 0; k < n; ++k) t[k] = (it & 1) ? rnd() % 7u : (uint32_t)(((unsigned long long)rnd() * 2654435761ull) % vocab);
      const scone_key k = scone_pack_key(t, n, max_n);
      if (!k.ok) { ++bad; continue; }
      if (k.lo == 0 && k.ext == 0) ++bad;                      // an all-zero key would read as an empty slot
      std::vector<uint32_t> g(t, t + n);
      auto key = std::make_tuple(k.lo, k.ext);
      auto f = seen.find(key);
      if (f == seen.end()) seen.emplace(key, g);
      else if (f->second != g) ++bad;                          // two different f-grams, one packed key
    }
  }
  // tokens that cannot be represented are refused, not wrapped
  { uint32_t t[4] = {0xFFFFFFu, 1, 2, 3}; if (scone_pack_key(t, 4, 4).ok) ++bad; }
  { uint32_t t[3] = {0xFFFFFFFFu, 1, 2}; if (scone_pack_key(t, 1, 3).ok) ++bad; }
  // 2. the probe sequence visits every bucket of a power-of-two table (odd step), whatever the hash
  for (int it = 0; it < 2000; ++it) {
    const unsigned long long h = ((unsigned long long)rnd() << 40) ^ ((unsigned long long)rnd() << 16) ^ rnd();
    const unsigned long long slot_mask = (1ull << (2 + it % 9)) * SCONE_BUCKET - 1;  // 4 .. 1024 buckets
    const unsigned long long nb = (slot_mask >> SCONE_BUCKET_SHIFT) + 1;
    std::set<unsigned long long> visited;
    unsigned long long b = scone_bucket_home(h, slot_mask);
    const unsigned long long step = scone_bucket_step(h);
    if (!(step & 1ull)) ++bad;
    for (unsigned long long k = 0; k < nb; ++k) {
      visited.insert(b);
      b = (b + step) & (nb - 1);
    }
    if (visited.size() != nb) ++bad;
  }
  // 3. INT4 group-scale slots: a bijection of the groups for every supported d, the pairing the kernel assumes at d = 1024
  for (int d = 128; d <= 4096; d += 128) {
    std::set<int> s;
    for (int g = 0; g < d / SCONE_I4_GROUP; ++g) {
      const int p = scone_i4_scale_slot(g, d);
      if (p < 0 || p >= d / SCONE_I4_GROUP) ++bad;
      s.insert(p);
    }
    if ((int)s.size() != d / SCONE_I4_GROUP) ++bad;
  }
  for (int lane = 0; lane < 64; ++lane) {   // lane's two groups (segments 0 and 1) are the halves of dword lane / 16
    const int g0 = (lane * 8) / SCONE_I4_GROUP, g1 = (512 + lane * 8) / SCONE_I4_GROUP;
    if (scone_i4_scale_slot(g0, 1024) != 2 * (lane >> 4) || scone_i4_scale_slot(g1, 1024) != 2 * (lane >> 4) + 1) ++bad;
  }
  // 4. grid helpers
  if (scone_grid_fits(1ull << 24, 256) || !scone_grid_fits((1ull << 24) - 1, 256)) ++bad;
  if (scone_capped_blocks(0) != 1 || scone_capped_blocks(1ull << 40) != SCONE_MAX_BLOCKS) ++bad;
  std::printf("host_pack_check: %d problem(s)\n", bad);
  return bad ? 1 : 0;
}
