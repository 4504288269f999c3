// What does page-locking cost, and does it get cheaper on huge pages?  (round 6)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  const size_t N = 320u << 20;
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char buf[128] = "";
  if (f) { fgets(buf, sizeof buf, f); fclose(f); }
  printf("THP: %s", buf);
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    void* p = nullptr;
    hipHostMalloc(&p, N, hipHostMallocDefault);
    double t1 = now();
    memset(p, 1, N);
    double t2 = now();
    hipHostFree(p);
    double t3 = now();
    printf("hipHostMalloc 320 MB: %.3f s, first touch %.3f s, free %.3f s\n", t1 - t0, t2 - t1, t3 - t2);
    t0 = now();
    void* m = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(m, N, MADV_HUGEPAGE);
    memset(m, 1, N);
    t1 = now();
    hipError_t e = hipHostRegister(m, N, hipHostRegisterDefault);
    t2 = now();
    printf("mmap + MADV_HUGEPAGE + touch: %.3f s, hipHostRegister: %.3f s (%s)\n", t1 - t0, t2 - t1, hipGetErrorString(e));
    void* d;
    hipMalloc(&d, N);
    t0 = now();
    hipMemcpy(d, m, N, hipMemcpyHostToDevice);
    t1 = now();
    printf("  H2D from it: %.1f GB/s\n", N / (t1 - t0) / 1e9);
    hipFree(d);
    t0 = now();
    if (e == hipSuccess) hipHostUnregister(m);
    munmap(m, N);
    printf("  unregister + unmap: %.3f s\n", now() - t0);
    t0 = now();
    m = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    t1 = now();
    e = hipHostRegister(m, N, hipHostRegisterDefault);
    t2 = now();
    printf("mmap MAP_POPULATE (4K pages): %.3f s, hipHostRegister: %.3f s (%s)\n", t1 - t0, t2 - t1, hipGetErrorString(e));
    if (e == hipSuccess) hipHostUnregister(m);
    munmap(m, N);
  }
  return 0;
}
