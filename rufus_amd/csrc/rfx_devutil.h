// Device-side constants and helpers shared by the partition / leaf kernels (rfx_p2l.hip, rfx_msp.hip).
#pragma once
#include "rfx_internal.h"

namespace {

constexpr int P2_BLOCK = 512;     // threads = reads per chunk in the partition kernels
// Leaf geometry: 8192-slot LDS table + 2048-entry sort area = 120 KB, one 1024-thread workgroup per CU.
// (Measured alternative: 4096 slots / 512 threads / 16 K bins gives two workgroups per CU and a 12 %
// faster leaf, but the finer bins cost more than that in k_bin_count and k_part2.)  The next bin's
// words are prefetched into registers while the current bin is being sorted and emitted.
constexpr int LEAF_BLOCK = 1024;
constexpr int LEAF_TBL_LOG2 = 13;
constexpr int LEAF_TBL = 1 << LEAF_TBL_LOG2;  // LDS hash slots per bin round
constexpr int LEAF_FILL = 6144;   // distinct keys allowed before the bin is split into more rounds
constexpr int LEAF_SORT = 2048;   // survivors sorted per round
constexpr int LEAF_RMAX = 20;

__device__ __forceinline__ uint64_t gf2_mul(const uint64_t* __restrict__ lut, uint64_t key, int ntab) {
  uint64_t r = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if (t < ntab) r ^= lut[t * 256 + (uint32_t)((key >> (8 * t)) & 255u)];
  return r;
}

constexpr int P1_BINS = 128;
constexpr int P1_S = 8;                       // bases per phase
constexpr int P1_STAGE = P2_BLOCK * P1_S;     // words staged per phase
constexpr int P1_CUR_STRIDE = 64;             // fused path: one 256 B line per coarse-bin cursor (L2 atomics
                                              // on one line serialise; 128 cursors in 4 lines cost 0.2 ms)
constexpr int L2_BLOCK = 1024;
constexpr int L2_PER = 8;                     // words per lane per tile
constexpr int L2_TILE = L2_BLOCK * L2_PER;

// exclusive scan of up to 256 LDS counters by wave 0 (4 per lane); returns the total in s_start[n]
__device__ __forceinline__ void wave_scan256(const uint32_t* s_cnt, uint32_t* s_start, uint32_t n) {
  const uint32_t l = threadIdx.x;  // caller guarantees l < 64
  uint32_t c[4], sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c[i] = 4 * l + i < n ? s_cnt[4 * l + i] : 0;
    sum += c[i];
  }
  uint32_t inc = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off);
    if ((int)l >= off) inc += o;
  }
  uint32_t ex = inc - sum;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (4 * l + i < n) s_start[4 * l + i] = ex;
    ex += c[i];
  }
  if (l == 63) s_start[n] = inc;
}

__device__ __forceinline__ uint32_t leaf_hash(uint64_t w) {
  uint32_t h = (uint32_t)w ^ (uint32_t)(w >> 19) ^ (uint32_t)(w >> 37);
  h *= 0x9E3779B1u;
  return h >> (32 - LEAF_TBL_LOG2);
}

constexpr int LEAF_ILP = 8;       // words loaded per lane before the first insert (64 KB in flight per workgroup)
constexpr int LEAF_BUCKETS = 256;  // survivors are bucketed on the next 8 bits of w, then ranked inside the bucket


}  // namespace
