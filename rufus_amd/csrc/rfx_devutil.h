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

// m-mers per k-mer (window of the sliding minimum); m = k - (MSP_WL - 1) = 13 .. 15 bases for k = 23 .. 25.
// The minimizer must be long enough for the SAMPLE, not for the block: all k-mers that share a minimizer
// m-mer land in one bin however far the partition is refined, and an m-mer that ranks low occurs
// genome_len / 4^m times x ~WL k-mers x coverage.  With WL = 15 (m = 11, the round-1 choice) that is 6.6e5
// instances per top m-mer on a 3.1 Gb genome at 30x -- the leaf re-ran such bins in k-mer hash halves and
// was 12x slower per read at 1 Gb than at 5 Mb; m = 15 gives 2 K.  Shorter windows mean shorter runs
// (3.0 instead of 3.4 k-mers per record), the price for bins that stay bins at any scale.
constexpr int MSP_WL = 11;
// k = 26 .. 31: window k - 15 = 11 .. 16, so that m = 16 (an m-mer still fits 32 bits).  Until the last day of round 6 the
// window was 16 for every k > 25 (m = k - 15 = 11 .. 15): exact, but with few and skewed minimizer bins (see above) -- at
// k = 26 the leaf took 25 times what it takes now, and on the 3.1 Gb genome 329 ms per sample at k = 29 against 178 at
// k = 25 (profiles/r06_selfcheck_sweep.txt).  k = 23 / 24 keep the window of 11 (m = 13 / 14; the leaf 542 / 216 ms per
// 3.1 Gb sample): a window of 10 at k = 23 brought the leaf to 209 ms but makes ~25 runs per read where a run map holds 27
// -- most reads lost their map and were hashed in every pass, 499 M reads/s either way (measured, not kept).
constexpr int MSP_WL_WIDE = 16;
__host__ __device__ __forceinline__ int msp_wl(int k) { return k >= 26 ? k - 15 : MSP_WL; }
__host__ __device__ __forceinline__ int msp_m(int k) { return k - (msp_wl(k) - 1); }
constexpr uint64_t MSP_EMPTY = ~0ull;  // no record looks like this: n - 1 (bits 63:60) stays below 15

// k-mers per record: a whole super-k-mer (all consecutive k-mers of a read that share the minimizer: at most as
// many as a k-mer has m-mers), capped by what the record can carry -- n - 1 < 15 (15 = MSP_EMPTY's) and
// k + n - 1 <= 43 bases (28 in the word, 15 in the plane).  k <= 26: 11; 27: 12; 28: 13; 29, 30: 14; 31: 13.
__host__ __device__ __forceinline__ int msp_nmax(int k) {
  const int w = msp_wl(k), c = 44 - k;
  return w < 15 ? (w < c ? w : c) : (15 < c ? 15 : c);
}

// One multiply each: 32-bit integer multiplies are quarter rate, and k_msp_part1 hashes every base.
// The xor keeps the all-A m-mer (c = 0) from hashing to 0 = always the minimum.
__device__ __forceinline__ uint32_t mmer_hash(uint32_t c) {
#ifdef RFX_P1_NOMUL  // experiment (results void): what the 32-bit multiply costs
  c = (c ^ 0x5BD1E995u) + (c << 7);
#else
  c = (c ^ 0x5BD1E995u) * 0x9E3779B1u;
#endif
  return c ^ (c >> 15);
}

// The low 5 bits of a hash are replaced by the position (mod 32) of the m-mer's last base: the sliding
// minimum then carries the position of the minimizer along for free, and ties between equal m-mers of
// one window are broken consistently.  Only the upper 27 bits decide the bin.
constexpr uint32_t MSP_HMASK = ~31u;
// The minimum of 11 .. 16 hashes crowds towards 0: spread it again before taking the top bits.  The TOP bit -- which half
// of the bin space, i.e. which of two shard passes -- is bit 5 of the hash itself (the lowest bit that is not position; as
// even as any): k_msp_part1 HMODE 4 notes it for every run of a read and would pay eight multiplies per phase for it.
constexpr uint32_t MSP_HALF_BIT = 32u;
__device__ __forceinline__ uint32_t msp_binhash(uint32_t minh) {
  return (((minh & MSP_HMASK) * 0xC2B2AE3Du) >> 1) | ((minh & MSP_HALF_BIT) << 26);
}
__device__ __forceinline__ uint32_t msp_bin(uint32_t minh, int bin_bits) {
  return msp_binhash(minh) >> (32 - bin_bits);  // msp_record_binhash() repeats this from the record
}

__device__ __forceinline__ uint64_t revcomp_bases(uint64_t s, int nbases) {
  uint64_t y = __brevll(~s);  // complement, then reverse: bit pairs end up swapped inside
  y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
  return y >> (64 - 2 * nbases);
}

// ---- super-k-mer record (round 4: one record = one WHOLE super-k-mer, for every k of the MSP path) -------------------
// A run of n consecutive k-mers of a read that share their minimizer (L = k + n - 1 bases) travels as a 64-bit word + a
// 32-bit plane (SoA: the partition kernels move the plane as a payload and never look at it):
//   word  [63:60] n - 1          [59:56] mpos = offset of the minimizer m-mer from the first base HELD IN THE WORD
//         [55:0]  min(L, 28) bases of the run, first base most significant (right-aligned when L < 28)
//   plane the other L - 28 bases; [31] side (k >= 29 only)
// The minimizer lies inside every k-mer of the run, so inside its first k-mer: for k <= 28 the FIRST 28 bases always
// hold it and the plane holds bases 28 .. L-1.  k = 29 .. 31: when the m-mer does not end within the first 28 bases
// (side = 1) the word holds bases s .. s+27 with s = k - 28 -- the last 28 of the first k-mer -- and the plane the s
// bases before them followed by the bases after them.  Either way every partition level gets its bits from the word
// alone (ONE m-mer hash: msp_record_binhash); only the leaf puts the run back together (msp_record_run).
// Until round 3 a record held <= 4 k-mers (8 bytes for k <= 25: 42 records per 150 bp read, 336 B; now 22, 264 B).
__device__ __forceinline__ int msp_record_n(uint64_t x) { return (int)(x >> 60) + 1; }

// k <= 25: a run has at most 35 bases, the plane at most 14 bits -- bits 14 .. 29 carry 16 bits of the record's minimizer
// bin hash (bits 3 .. 18), stamped where the record is made.  The refinement's sizing pass (k_bin_hist) then reads the
// 4-byte planes instead of the 8-byte words and hashes nothing: half its bytes (27 -> 14 ms per W sample).  The leaf
// masks the stamp off (msp_plane_bits); every partition level moves the plane untouched.
constexpr int MSP_STAMP_SHIFT = 14, MSP_STAMP_LO = 3, MSP_STAMP_BITS = 16;
__host__ __device__ __forceinline__ bool msp_stamped(int k) { return k <= 25; }
__device__ __forceinline__ uint32_t msp_stamp(uint32_t binhash, int k) {
  return msp_stamped(k) ? ((binhash >> MSP_STAMP_LO) & ((1u << MSP_STAMP_BITS) - 1u)) << MSP_STAMP_SHIFT : 0u;
}
__device__ __forceinline__ uint32_t msp_plane_bits(uint32_t xe, int k) { return msp_stamped(k) ? xe & ((1u << MSP_STAMP_SHIFT) - 1u) : xe; }
// (binhash >> shift) & (P2 - 1) from a stamped plane; usable when MSP_STAMP_LO <= shift and shift + log2(P2) <= 19
__device__ __forceinline__ uint32_t msp_stamp_sub(uint32_t xe, int shift, uint32_t P2) {
  return (xe >> (MSP_STAMP_SHIFT + shift - MSP_STAMP_LO)) & (P2 - 1);
}

// The COARSE bins k_msp_part1 fills hold word and plane side by side, 12 bytes: one store per record.  The kernel is
// bound by how many lane-stores the memory pipeline takes (every record goes to another line: 2.5 cycles per lane-store
// and CU, measured) -- with two arrays the stores were 38 % of it.  k_part2 reads them back as 12-byte loads (a wave's
// 64 records are 768 contiguous bytes) and writes the two arrays every later level works on.
struct __attribute__((packed, aligned(4))) msp_rec12 {
  uint32_t lo, hi, x;
};
__device__ __forceinline__ void msp_rec12_store(msp_rec12* a, uint64_t i, uint64_t w, uint32_t x) {
  a[i] = msp_rec12{(uint32_t)w, (uint32_t)(w >> 32), x};
}
__device__ __forceinline__ uint64_t msp_rec12_word(const msp_rec12* a, uint64_t i) {  // (the histogram fallback needs no plane)
  return (uint64_t)a[i].lo | ((uint64_t)a[i].hi << 32);
}

template <bool CANON>
__device__ __forceinline__ uint32_t msp_record_binhash(uint64_t x, int k) {
  const int n = msp_record_n(x), m = msp_m(k);
  const int L = min(k + n - 1, 28), mpos = (int)(x >> 56) & 15;
  const uint32_t mmask = m >= 16 ? ~0u : (1u << (2 * m)) - 1;
  const uint32_t f = (uint32_t)((x & ((1ull << 56) - 1)) >> (2 * (L - m - mpos))) & mmask;
  uint32_t c = f;
  if (CANON) {
    uint32_t y = __brev(~f);
    y = ((y & 0xAAAAAAAAu) >> 1) | ((y & 0x55555555u) << 1);
    c = min(f, y >> (32 - 2 * m));
  }
  return msp_binhash(mmer_hash(c));
}

// Word and plane of the run that is the low 2L bits of (hi:lo) (first base most significant); mpos = offset of its
// minimizer m-mer from the run's first base.
__device__ __forceinline__ void msp_record_make(uint64_t lo, uint32_t hi, int k, int n, uint32_t mpos, uint64_t& word,
                                                uint32_t& plane) {
  const int L = k + n - 1, m = msp_m(k);
  const uint64_t m56 = (1ull << 56) - 1;
  uint64_t bases;
  uint32_t mp = mpos;
  if (L <= 28) {
    bases = lo & ((1ull << (2 * L)) - 1);
    plane = 0;
  } else if (k <= 28 || (int)mpos + m <= 28) {  // the first 28 bases; the plane gets the last x
    const int x = L - 28;
    bases = ((lo >> (2 * x)) | ((uint64_t)hi << (64 - 2 * x))) & m56;
    plane = (uint32_t)lo & ((1u << (2 * x)) - 1);
  } else {  // k >= 29, side 1: bases s .. s+27; the plane gets the s before them and the t after them
    const int s = k - 28, t = L - 28 - s;
    const uint64_t v = t ? (lo >> (2 * t)) | ((uint64_t)hi << (64 - 2 * t)) : lo;   // run >> 2t: 56 + 2s <= 62 bits
    bases = v & m56;
    const uint32_t head = (uint32_t)(v >> 56) & ((1u << (2 * s)) - 1);
    plane = (head << (2 * t)) | ((uint32_t)lo & ((1u << (2 * t)) - 1)) | 0x80000000u;
    mp = mpos - (uint32_t)s;
  }
  word = bases | ((uint64_t)mp << 56) | ((uint64_t)(n - 1) << 60);
}

// The run of a record back as the low 2L bits of (hi:lo).
__device__ __forceinline__ void msp_record_run(uint64_t x, uint32_t xe, int k, uint64_t& lo, uint64_t& hi) {
  const int n = msp_record_n(x), L = k + n - 1;
  const uint64_t S = x & ((1ull << 56) - 1);
  if (k <= 28 || !(xe >> 31)) {  // (no branch on the run's length: the leaf runs this with every lane on another record)
    const int x2 = 2 * max(L - 28, 0);
    lo = (S << x2) | (xe & 0x7FFFFFFFu);   // (the plane of a run of <= 28 bases is 0)
    hi = (S >> 1) >> (63 - x2);
  } else {
    const int s = k - 28, t = L - 28 - s;
    const uint64_t head = (xe >> (2 * t)) & ((1u << (2 * s)) - 1), tail = xe & ((1u << (2 * t)) - 1);
    const uint64_t mid = S | (head << 56);  // head:S, 56 + 2s <= 62 bits
    lo = t ? (mid << (2 * t)) | tail : mid;
    hi = t ? mid >> (64 - 2 * t) : 0;
  }
}

// k-mer q (0 = the first) of a run of n k-mers given as (hi:lo)
__device__ __forceinline__ uint64_t msp_run_kmer(uint64_t lo, uint64_t hi, int k, int n, int q) {
  const int sh = 2 * (n - 1 - q);
  return (sh ? (lo >> sh) | (hi << (64 - sh)) : lo) & ((1ull << (2 * k)) - 1);
}

constexpr uint32_t RMAP_ENTRIES = 27;  // entries a read's run map holds (rfx_msp.hip, k_msp_replay)
constexpr int P1_BINS = 128;
constexpr int P1_S = 8;                       // bases per phase
constexpr int P1_STAGE = P2_BLOCK * P1_S;     // words staged per phase
constexpr int P1_CUR_STRIDE = 64;             // fused path: one 256 B line per coarse-bin cursor (L2 atomics
                                              // on one line serialise; 128 cursors in 4 lines cost 0.2 ms)
constexpr int L2_BLOCK = 1024;
#ifndef RFX_L2_PER
#define RFX_L2_PER 8
#endif
constexpr int L2_PER = RFX_L2_PER;            // words per lane per tile
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

__device__ __forceinline__ uint32_t leaf_hash32(uint64_t w) {
  const uint32_t h = (uint32_t)w ^ (uint32_t)(w >> 19) ^ (uint32_t)(w >> 37);
  return h * 0x9E3779B1u;
}
__device__ __forceinline__ uint32_t leaf_hash(uint64_t w) {
  uint32_t h = (uint32_t)w ^ (uint32_t)(w >> 19) ^ (uint32_t)(w >> 37);
  h *= 0x9E3779B1u;
  return h >> (32 - LEAF_TBL_LOG2);
}

constexpr int LEAF_ILP = 8;       // words loaded per lane before the first insert (64 KB in flight per workgroup)
constexpr int LEAF_BUCKETS = 256;  // survivors are bucketed on the next 8 bits of w, then ranked inside the bucket


}  // namespace
