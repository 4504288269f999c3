// LDS atomic throughput on gfx950: one 1024-thread block per CU, random slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int TBL = 8192, ITER = 256;
__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE, int ILP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, uint32_t distinct) {
  __shared__ unsigned long long s_k[TBL];
  __shared__ uint32_t s_c[TBL];
  for (int i = threadIdx.x; i < TBL; i += 1024) { s_k[i] = ~0ull; s_c[i] = 0; }
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
  for (int it = 0; it < ITER; it += ILP) {
    uint32_t slot[ILP]; unsigned long long key[ILP], got[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) { const uint32_t r = rnd(s) % distinct; slot[u] = (r * 2654435761u) >> 19; key[u] = r; }
    if (MODE == 0) {        // returning 64-bit CAS then non-returning 32-bit add (the leaf's cache insert)
#pragma unroll
      for (int u = 0; u < ILP; ++u) got[u] = atomicCAS(&s_k[slot[u]], ~0ull, key[u]);
#pragma unroll
      for (int u = 0; u < ILP; ++u) if (got[u] == ~0ull || got[u] == key[u]) atomicAdd(&s_c[slot[u]], 1u); else acc++;
    } else if (MODE == 1) { // non-returning 32-bit add only
#pragma unroll
      for (int u = 0; u < ILP; ++u) atomicAdd(&s_c[slot[u]], 1u);
    } else if (MODE == 2) { // returning 32-bit add
#pragma unroll
      for (int u = 0; u < ILP; ++u) acc += atomicAdd(&s_c[slot[u]], 1u);
    } else if (MODE == 3) { // plain 64-bit read + 32-bit write
#pragma unroll
      for (int u = 0; u < ILP; ++u) { acc += (uint32_t)s_k[slot[u]]; s_c[slot[u]] = acc; }
    } else if (MODE == 4) { // returning 32-bit CAS
#pragma unroll
      for (int u = 0; u < ILP; ++u) acc += atomicCAS(&s_c[slot[u]], 0u, (uint32_t)key[u] + 1u);
    } else if (MODE == 5) { // returning 64-bit CAS only
#pragma unroll
      for (int u = 0; u < ILP; ++u) acc += (uint32_t)atomicCAS(&s_k[slot[u]], ~0ull, key[u]);
    }
  }
  if (acc == 0xDEADBEEF) out[0] = acc;
}

template <int MODE, int ILP>
int run(const char* name, uint32_t distinct) {
  uint32_t* d; CK(hipMalloc(&d, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<MODE, ILP><<<256, 1024>>>(d, distinct);
  CK(hipEventRecord(a));
  for (int r = 0; r < 10; ++r) k<MODE, ILP><<<256, 1024>>>(d, distinct);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
  const double ops = 1024.0 * ITER;  // lane-ops per CU
  printf("%-34s ilp %d distinct %6u: %.3f ms  -> %.2f cycles per lane-op per CU (at 2.4 GHz)\n", name, ILP, distinct, ms, ms * 1e-3 * 2.4e9 / ops);
  CK(hipFree(d)); return 0;
}
int main() {
  for (uint32_t dist : {4096u, 256u}) {
    run<0, 1>("cas64 rtn + add32", dist); run<0, 8>("cas64 rtn + add32", dist);
    run<5, 1>("cas64 rtn", dist); run<5, 8>("cas64 rtn", dist);
    run<4, 1>("cas32 rtn", dist); run<4, 8>("cas32 rtn", dist);
    run<1, 1>("add32", dist); run<1, 8>("add32", dist);
    run<2, 1>("add32 rtn", dist); run<2, 8>("add32 rtn", dist);
    run<3, 1>("read64 + write32", dist); run<3, 8>("read64 + write32", dist);
  }
  return 0;
}
