// Does a 64-bit global atomic OR set exactly the bits it is given?  (scratch: the pair filter's mask-only bug, round 6)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; return x ^ (x >> 16); }
template <int WIDE>
__global__ void k_set(unsigned long long* mask, uint32_t n, uint32_t thr, int rounds) {
  for (uint32_t r0 = blockIdx.x * blockDim.x + threadIdx.x; r0 < n; r0 += gridDim.x * blockDim.x) {
    // a divergent little loop in front, as a drain has: some lanes probe longer than others
    uint32_t h = mix(r0), spin = h & 7u, acc = 0;
    while (spin--) acc += mix(acc + spin);
    for (int t = 0; t < rounds; ++t) {
      const uint32_t r = r0 ^ (uint32_t)t;          // (a few candidates of neighbouring reads per lane)
      const bool hit = r < n && mix(r * 2654435761u + (acc & 0u)) < thr;
      if (hit) {
        if (WIDE) atomicOr(&mask[r >> 6], 1ull << (r & 63u));
        else atomicOr((unsigned int*)mask + (r >> 5), 1u << (r & 31u));
      }
    }
  }
}
int main() {
  const uint32_t n = 1u << 24;
  unsigned long long* d = nullptr;
  if (getenv("VMM")) {   // memory as the library's arena maps it
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    const size_t CH = 1ull << 30; void* base = nullptr; hipMemAddressReserve(&base, 4 * CH, gran, nullptr, 0);
    hipMemGenericAllocationHandle_t h; hipMemCreate(&h, CH, &prop, 0); hipMemMap((char*)base + CH, CH, 0, h, 0);
    hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
    hipMemSetAccess((char*)base + CH, CH, &ad, 1);
    d = (unsigned long long*)((char*)base + CH + 256 * 1000);
    printf("VMM-mapped memory\n");
  } else hipMalloc(&d, n / 8);
  std::vector<unsigned long long> got(n / 64), want(n / 64);
  auto hmix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; return x ^ (x >> 16); };
  for (uint32_t thr : {4000000u, 40000000u, 400000000u, 2000000000u}) {
    std::fill(want.begin(), want.end(), 0ull);
    for (uint32_t r = 0; r < n; ++r) if (hmix(r * 2654435761u) < thr) want[r >> 6] |= 1ull << (r & 63);
    for (int wide = 1; wide >= 0; --wide) {
      long extra = 0, missing = 0;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(d, 0, n / 8);
        if (wide) hipLaunchKernelGGL(k_set<1>, dim3(256), dim3(1024), 0, 0, d, n, thr, 4);
        else hipLaunchKernelGGL(k_set<0>, dim3(256), dim3(1024), 0, 0, d, n, thr, 4);
        hipMemcpy(got.data(), d, n / 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < got.size(); ++i) { extra += __builtin_popcountll(got[i] & ~want[i]); missing += __builtin_popcountll(want[i] & ~got[i]); }
      }
      printf("hit rate %.4f  %s atomics: extra %ld missing %ld (5 runs)\n", thr / 4294967296.0, wide ? "64-bit" : "32-bit", extra, missing);
    }
  }
  return 0;
}
