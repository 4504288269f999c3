// What the first device allocations of a process cost, call by call (scratch: not part of the product).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(what, call) do { double t0 = now(); hipError_t e = (call); printf("  %-44s %8.3f ms  %s\n", what, (now() - t0) * 1e3, e == hipSuccess ? "" : hipGetErrorString(e)); } while (0)
int main(int argc, char** argv) {
  const size_t GB = 1ull << 30, CH = argc > 1 ? (size_t)atoll(argv[1]) << 20 : GB;
  double t0 = now();
  T("hipSetDevice", hipSetDevice(0));
  hipStream_t s; T("hipStreamCreate", hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  void* pin; T("hipHostMalloc 4 MB", hipHostMalloc(&pin, 4 << 20, hipHostMallocDefault));
  printf("  init total %.3f ms\n", (now() - t0) * 1e3);
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; T("hipMemGetAllocationGranularity", hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t fb, tb; T("hipMemGetInfo", hipMemGetInfo(&fb, &tb));
  printf("  free %.1f GB of %.1f\n", fb / 1e9, tb / 1e9);
  void* base = nullptr; const size_t reserve = (tb + 8 * GB + GB - 1) / GB * GB;
  T("hipMemAddressReserve (total + 8 GB)", hipMemAddressReserve(&base, reserve, gran, nullptr, 0));
  hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> hs;
  for (int i = 0; i < 8; ++i) {
    hipMemGenericAllocationHandle_t h; char* at = (char*)base + i * CH;
    printf(" chunk %d (%zu MB)\n", i, CH >> 20);
    T("hipMemCreate", hipMemCreate(&h, CH, &prop, 0));
    T("hipMemMap", hipMemMap(at, CH, 0, h, 0));
    T("hipMemSetAccess", hipMemSetAccess(at, CH, &ad, 1));
    hs.push_back(h);
  }
  T("hipMemsetAsync 8 chunks + sync", (hipMemsetAsync(base, 1, 8 * CH, s), hipStreamSynchronize(s)));
  void* p; T("hipMalloc 6.7 GB", hipMalloc(&p, (size_t)(6.7 * GB)));
  T("hipMemset of it + sync", (hipMemsetAsync(p, 1, (size_t)(6.7 * GB), s), hipStreamSynchronize(s)));
  T("hipFree", hipFree(p));
  printf("  total %.3f ms\n", (now() - t0) * 1e3);
  return 0;
}
