// Microbenchmark: random global atomics vs working-set size, LDS atomics, scattered run writes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ __forceinline__ uint64_t mix(uint64_t x){ x ^= x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }

// each block works in its own region of `region` slots (block-local working set) or the whole table
__global__ void k_atomic_u32(uint32_t* tab, uint64_t mask, int iters, int per_block_region, uint64_t region_slots) {
  uint64_t base = per_block_region ? (((uint64_t)blockIdx.x * region_slots) & ((1ull << 28) - 1)) : 0;
  uint64_t m = per_block_region ? region_slots - 1 : mask;
  uint64_t s = mix(blockIdx.x * 1024ull + threadIdx.x + 1);
  for (int i = 0; i < iters; ++i) { s = mix(s + i); atomicAdd(&tab[(base + (s & m))], 1u); }
}
__global__ void k_load_then_atomic(uint64_t* keys, uint32_t* cnt, uint64_t mask, int iters) {
  uint64_t s = mix(blockIdx.x * 1024ull + threadIdx.x + 1);
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) { s = mix(s + i); uint64_t slot = s & mask; uint64_t k = keys[slot]; if (k != 12345) atomicAdd(&cnt[slot], 1u); else acc++; }
  if (acc == 0xFFFFFFFF) cnt[0] = acc;
}
__global__ void k_cas64(unsigned long long* keys, uint64_t mask, int iters) {
  uint64_t s = mix(blockIdx.x * 1024ull + threadIdx.x + 1);
  unsigned long long acc = 0;
  for (int i = 0; i < iters; ++i) { s = mix(s + i); acc += atomicCAS(&keys[s & mask], ~0ull, s | 1); }
  if (acc == 1) keys[0] = acc;
}
// LDS atomics: 64-bit CAS + 32-bit add into a 32K-slot LDS table
__global__ void k_lds_atomic(uint32_t* out, int iters) {
  __shared__ unsigned long long lk[4096];
  __shared__ uint32_t lc[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) { lk[i] = ~0ull; lc[i] = 0; }
  __syncthreads();
  uint64_t s = mix(blockIdx.x * 1024ull + threadIdx.x + 1);
  for (int i = 0; i < iters; ++i) {
    s = mix(s + (i & 1023));
    unsigned long long key = (s >> 12) & 2047; uint32_t slot = (uint32_t)(key * 2654435761u) & 4095;  // 2048 distinct keys
    for (;;) { unsigned long long cur = lk[slot]; if (cur == ~0ull) cur = atomicCAS(&lk[slot], ~0ull, key); if (cur == ~0ull || cur == key) { atomicAdd(&lc[slot], 1u); break; } slot = (slot + 1) & 4095; }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lc[0];
}
// scattered run writes: each wave writes runs of `run` bytes to random 'bins' (appending with a cursor atomic)
__global__ void k_scatter_runs(uint8_t* dst, unsigned long long* cursors, int nbins, uint64_t bin_bytes, int run_bytes, int iters) {
  int lane = threadIdx.x & 63;
  uint64_t s = mix(blockIdx.x * 64ull + (threadIdx.x >> 6) + 7);
  int lanes_per_run = run_bytes / 8;  // 8B per lane
  for (int i = 0; i < iters; ++i) {
    // 64 lanes => 64/lanes_per_run runs per iteration, each to a random bin
    int r = lane / lanes_per_run; uint64_t sr = mix(s + i * 131 + r);
    int bin = sr % nbins; unsigned long long off = 0;
    if (lane % lanes_per_run == 0) off = atomicAdd(&cursors[bin], (unsigned long long)run_bytes);
    off = __shfl(off, r * lanes_per_run);
    off %= (bin_bytes - run_bytes);
    *(uint64_t*)(dst + (uint64_t)bin * bin_bytes + off + (lane % lanes_per_run) * 8) = sr;
  }
}
int main() { setvbuf(stdout, NULL, _IONBF, 0);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 2048, threads = 256, iters = 256;
  const double nops = (double)blocks * threads * iters;
  uint64_t maxslots = 1ull << 28;
  uint32_t* tab; CK(hipMalloc(&tab, maxslots * 4)); CK(hipMemset(tab, 0, maxslots * 4));
  uint64_t* keys; CK(hipMalloc(&keys, maxslots * 8)); CK(hipMemset(keys, 0xFF, maxslots * 8));
  float ms;
  for (int lg = 30; lg <= 28; lg += 2) {
    uint64_t mask = (1ull << lg) - 1;
    hipLaunchKernelGGL(k_atomic_u32, dim3(blocks), dim3(threads), 0, 0, tab, mask, 8, 0, 0);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_atomic_u32, dim3(blocks), dim3(threads), 0, 0, tab, mask, iters, 0, 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    double a = nops / ms / 1e6;
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_load_then_atomic, dim3(blocks), dim3(threads), 0, 0, keys, tab, mask, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    double b = nops / ms / 1e6;
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_cas64, dim3(blocks), dim3(threads), 0, 0, (unsigned long long*)keys, mask, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    double c = nops / ms / 1e6;
    CK(hipMemset(keys, 0xFF, maxslots * 8));
    printf("table 2^%d slots (%6.1f MB u32): atomicAdd %.1f G/s | load64+atomicAdd %.1f G/s | CAS64 %.1f G/s\n", lg, (double)(mask + 1) * 4 / 1e6, a, b, c);
  }
  // block-private regions of 64K slots (256KB) -> L2 resident
  for (int lg = 30; lg <= 18; lg += 2) {
    uint64_t rs = 1ull << lg;
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_atomic_u32, dim3(blocks), dim3(threads), 0, 0, tab, 0, iters, 1, rs); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("block-private region 2^%d slots (%.0f KB): atomicAdd %.1f G/s\n", lg, rs * 4 / 1e3, nops / ms / 1e6);
  }
  uint32_t* out; CK(hipMalloc(&out, 8192 * 4));
  for (int thr : {256, 512, 1024}) {
    int bl = 256 * 8 * 256 / thr;
    hipLaunchKernelGGL(k_lds_atomic, dim3(bl), dim3(thr), 0, 0, out, 16);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_atomic, dim3(bl), dim3(thr), 0, 0, out, 2048); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("LDS CAS64+add32 insert (block %d): %.1f G inserts/s\n", thr, (double)bl * thr * 2048 / ms / 1e6);
  }
  // scattered runs
  uint8_t* dst; uint64_t total = 4ull << 30; CK(hipMalloc(&dst, total));
  unsigned long long* cur; CK(hipMalloc(&cur, 65536 * 8));
  for (int nbins : {256, 2048, 8192}) for (int run : {16, 64, 128, 512}) {
    CK(hipMemset(cur, 0, 65536 * 8));
    uint64_t bin_bytes = total / nbins;
    int it = 512;
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_scatter_runs, dim3(2048), dim3(256), 0, 0, dst, cur, nbins, bin_bytes, run, it); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    double bytes = 2048.0 * 256 * it * 8;
    printf("scatter nbins %5d run %3d B: %.0f GB/s\n", nbins, run, bytes / ms / 1e6);
  }
  return 0;
}
