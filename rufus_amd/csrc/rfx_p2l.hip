// "P2L" count path: partition -> LDS count -> sorted emit.  No global atomics per k-mer.
//
// Measured on MI355X (scratch/ubench_atomics.hip): device-scope global atomics top out at ~27 G/s
// whatever the working set (even L2-resident), LDS CAS64+add32 inserts run at ~1 T/s.  So a k-mer
// instance is written ONCE to HBM as an 8-byte key into the bin of its (pos,key) order prefix,
// and each bin is then counted entirely in LDS, sorted there, and emitted in output order:
//
//   k_bin_count    reads -> canonical k-mers -> bin histogram per block           (LDS atomics)
//   k_bin_offsets  per-bin exclusive offsets, block runs grouped by XCD           (tiny)
//   k_bin_scatter  recompute the k-mers, store each key at its reserved slot      (8 B/k-mer write)
//   k_leaf         one workgroup per bin: LDS hash count, survivors sorted in LDS (8 B/k-mer read)
//   k_leaf_compact bins -> dense (pos,key)-sorted records
//
// replaces jf/include/jellyfish/large_hash_array.hpp:298-302,:513-744 (hash insert) and
// jf/include/jellyfish/sorted_dumper.hpp:80-112 (sorted dump) in one go.
#include "rfx_internal.h"
#include "rfx_devutil.h"

namespace {

// Sortable word w = T * key (GF(2), 2k bits): the top lsize bits are pos = M * key, the low 2k-lsize
// bits are the key bits at the free columns of M taken high to low.  T is invertible and numeric
// order of w IS the (pos,key) output order (host side: build_sort_transform in rfx_api.hip), so
// bins and rounds are prefixes of w, survivors sort on one 64-bit compare, and the key comes back
// as Tinv * w for the few records that are emitted.

// Visit every counted window of read r: f(key, pos).
template <bool CANON, typename F>
__device__ __forceinline__ void for_each_kmer(const rfx_reads_view& rv, uint32_t r, int k, const uint64_t* s_lut,
                                              int ntab, int sel_bits, uint64_t pos_lo, uint64_t pos_hi, F&& f) {
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  const int rcshift = 2 * (k - 1);
  const uint32_t wr = rv_off(rv, r);
  const uint32_t len = rv_len(rv, r);
  const uint64_t* cw = rv.codes + wr;
  const uint32_t* cm = rv_acgt(rv, r, wr);  // nullptr: no mask kept, every base counts (compact blocks)
  uint64_t fwd = 0, rc = 0;
  int filled = 0;
  const uint32_t nw = (len + 31) >> 5;
  for (uint32_t wi = 0; wi < nw; ++wi) {
    uint64_t w = cw[wi];
    uint32_t m = cm ? cm[wi] : ~0u;
    const int nb = min(32u, len - (wi << 5));
    for (int b = 0; b < nb; ++b) {
      const uint32_t code = (uint32_t)w & 3u;
      w >>= 2;
      const bool valid = m & 1u;
      m >>= 1;
      fwd = ((fwd << 2) | code) & kmask;
      if (CANON) rc = (rc >> 2) | ((uint64_t)(3u - code) << rcshift);
      filled = valid ? filled + 1 : 0;
      if (filled >= k) {
        const uint64_t key = CANON ? (rc < fwd ? rc : fwd) : fwd;
        const uint64_t w = gf2_mul(s_lut, key, ntab);
        const uint64_t pos = w >> sel_bits;
        if (pos >= pos_lo && pos < pos_hi) f(w);
      }
    }
  }
}

template <bool CANON>
__global__ __launch_bounds__(P2_BLOCK) void k_bin_count(rfx_reads_view rv, const uint64_t* __restrict__ g_lut, int ntab,
                                                         int k, rfx_ord_cfg cfg, uint32_t P, uint64_t pos_lo,
                                                         uint64_t pos_hi, uint32_t* __restrict__ cnt) {
  extern __shared__ uint64_t s_dyn[];
  uint64_t* s_lut = s_dyn;
  uint32_t* s_hist = (uint32_t*)(s_dyn + 8 * 256);
  for (int i = threadIdx.x; i < ntab * 256; i += blockDim.x) s_lut[i] = g_lut[i];
  for (uint32_t b = threadIdx.x; b < P; b += blockDim.x) s_hist[b] = 0;
  __syncthreads();
  const uint32_t n_chunks = (rv.n + P2_BLOCK - 1) / P2_BLOCK;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r = chunk * P2_BLOCK + threadIdx.x;
    if (r < rv.n)
      for_each_kmer<CANON>(rv, r, k, s_lut, ntab, cfg.sel_bits, pos_lo, pos_hi,
                           [&](uint64_t w) { atomicAdd(&s_hist[(uint32_t)(w >> cfg.bin_shift)], 1u); });
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < P; b += blockDim.x) cnt[(uint64_t)blockIdx.x * P + b] = s_hist[b];
}

// cnt[g][b] (instances of bin b seen by block g) -> start of block g's run inside bin b, in place.
// Runs of one bin are ordered by (g % 8, g / 8): blocks that share an XCD (observed dispatch: block
// g runs on XCD g % 8) write neighbouring runs.  Two passes, thread = (XCD group x, bin b) with b
// fastest so every load is coalesced: group sums, then offsets.
__global__ __launch_bounds__(256) void k_bin_group_sums(const uint32_t* __restrict__ cnt, uint32_t G, uint32_t P,
                                                         uint32_t* __restrict__ gsum /* [8][P] */) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 8 * P) return;
  const uint32_t x = t / P, b = t - x * P;
  uint32_t s = 0;
  for (uint32_t g = x; g < G; g += 8) s += cnt[(uint64_t)g * P + b];
  gsum[t] = s;
}

__global__ __launch_bounds__(256) void k_bin_offsets(uint32_t* __restrict__ cnt, uint32_t G, uint32_t P,
                                                      const uint32_t* __restrict__ gsum,
                                                      uint64_t* __restrict__ bin_tot) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 8 * P) return;
  const uint32_t x = t / P, b = t - x * P;
  uint32_t run = 0, tot = 0;
  for (uint32_t y = 0; y < 8; ++y) {
    const uint32_t v = gsum[y * P + b];
    if (y < x) run += v;
    tot += v;
  }
  if (x == 0) bin_tot[b] = tot;
  for (uint32_t g = x; g < G; g += 8) {
    const uint64_t i = (uint64_t)g * P + b;
    const uint32_t c = cnt[i];
    cnt[i] = run;
    run += c;
  }
}

// bin_tot[b] = sum over blocks of cnt[g][b]: all the fused partitions need (they reserve their runs
// with atomics, so the per-block offsets that k_bin_offsets also produces are not used)
__global__ __launch_bounds__(1024) void k_col_sums(const uint32_t* __restrict__ cnt, uint32_t G, uint32_t P,
                                                    uint64_t* __restrict__ bin_tot) {
  __shared__ uint32_t s_part[16][64];
  const uint32_t lane = threadIdx.x & 63, q = threadIdx.x >> 6;  // 16 row groups x 64 bins
  const uint32_t b = blockIdx.x * 64 + lane;
  uint32_t s = 0;
  if (b < P)
    for (uint32_t g = q; g < G; g += 16) s += cnt[(uint64_t)g * P + b];
  s_part[q][lane] = s;
  __syncthreads();
  if (q == 0 && b < P) {
    uint64_t t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s_part[i][lane];
    bin_tot[b] = t;
  }
}

template <bool CANON>
__global__ __launch_bounds__(P2_BLOCK) void k_bin_scatter(rfx_reads_view rv, const uint64_t* __restrict__ g_lut,
                                                           int ntab, int k, rfx_ord_cfg cfg, uint32_t P, uint64_t pos_lo,
                                                           uint64_t pos_hi, const uint32_t* __restrict__ rel,
                                                           const uint64_t* __restrict__ bin_start,
                                                           uint64_t* __restrict__ inst) {
  extern __shared__ uint64_t s_dyn[];
  uint64_t* s_lut = s_dyn;
  uint32_t* s_cur = (uint32_t*)(s_dyn + 8 * 256);
  for (int i = threadIdx.x; i < ntab * 256; i += blockDim.x) s_lut[i] = g_lut[i];
  for (uint32_t b = threadIdx.x; b < P; b += blockDim.x)
    s_cur[b] = (uint32_t)bin_start[b] + rel[(uint64_t)blockIdx.x * P + b];  // a segment holds < 2^32 instances
  __syncthreads();
  const uint32_t n_chunks = (rv.n + P2_BLOCK - 1) / P2_BLOCK;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r = chunk * P2_BLOCK + threadIdx.x;
    if (r < rv.n)
      for_each_kmer<CANON>(rv, r, k, s_lut, ntab, cfg.sel_bits, pos_lo, pos_hi, [&](uint64_t w) {
        const uint32_t i = atomicAdd(&s_cur[(uint32_t)(w >> cfg.bin_shift)], 1u);
        inst[i] = w;
      });
  }
}

// ---------------------------------------------------------------------------------------------
// Two-level partition (used when there are >= 2048 bins).  Scattered 8-byte stores into thousands of
// bins are partial-line writes (measured 0.6-1.1 TB/s); instead the words are reordered in LDS so
// that every global store instruction writes whole runs:
//   k_part1  reads -> words -> P1 = 128 coarse bins, phase by phase (8 bases of 512 reads = up to
//            4096 words per phase, ~32-word runs per coarse bin);
//   k_part2  coarse bin -> its P/P1 fine bins, 8192-word tiles, ~128-word runs.
// Fine bin sizes are exact (k_bin_count), so both levels write into exactly sized regions.
// ---------------------------------------------------------------------------------------------
// cnt[g][f] over fine bins -> cnt1[g][cb] over coarse bins (P2 consecutive fine bins each)
__global__ __launch_bounds__(256) void k_coarse_counts(const uint32_t* __restrict__ cnt, uint32_t G, uint32_t P,
                                                        uint32_t P2, uint32_t* __restrict__ cnt1) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t P1 = P / P2;
  if (t >= G * P1) return;
  const uint32_t g = t / P1, cb = t - g * P1;
  const uint32_t* p = cnt + (uint64_t)g * P + (uint64_t)cb * P2;
  uint32_t s = 0;
  for (uint32_t i = 0; i < P2; ++i) s += p[i];
  cnt1[t] = s;
}

// FUSED = true: the sizing pass (k_bin_count) is folded in.  Coarse bins then have a fixed capacity
// `cap_a` and every phase reserves its runs with one global atomic per coarse bin; the exact fine-bin
// histogram that k_part2 / k_leaf need falls out of a 16-bit LDS histogram (P must be <= 8192).  Both
// shortcuts can fail on pathological input (one k-mer family holding > 20 % of a coarse bin, or
// > 65535 instances of one fine bin inside one block): *flag is raised, nothing is written out of
// bounds, and the host redoes the block on the exact path.
template <bool CANON, bool FUSED>
__global__ __launch_bounds__(P2_BLOCK) void k_part1(rfx_reads_view rv, const uint64_t* __restrict__ g_lut, int ntab,
                                                     int k, rfx_ord_cfg cfg, uint32_t P2, uint64_t pos_lo,
                                                     uint64_t pos_hi, const uint32_t* __restrict__ rel1,
                                                     const uint64_t* __restrict__ fine_start,
                                                     uint64_t* __restrict__ buf_a, uint32_t* __restrict__ coarse_cur,
                                                     uint32_t cap_a, uint32_t* __restrict__ cnt_rows,
                                                     unsigned int* __restrict__ flag) {
  __shared__ uint64_t s_lut[8 * 256];
  __shared__ uint64_t s_stage[P1_STAGE];
  __shared__ uint8_t s_sbin[P1_STAGE];
  __shared__ uint32_t s_cnt[P1_BINS], s_start[P1_BINS + 1], s_gcur[P1_BINS], s_gbase[P1_BINS];
  __shared__ uint32_t s_maxlen;
  __shared__ uint32_t s_fine[FUSED ? 4096 : 1];  // two 16-bit counters per word: fine bins 2i, 2i+1
  const uint32_t P = P2 * P1_BINS;
  uint32_t blk_total = 0;  // words this block produced (uniform)
  for (int i = threadIdx.x; i < ntab * 256; i += blockDim.x) s_lut[i] = g_lut[i];
  if (FUSED)
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) s_fine[i] = 0;
  if (threadIdx.x < P1_BINS) {
    if (!FUSED)
      s_gcur[threadIdx.x] =
          (uint32_t)fine_start[(uint64_t)threadIdx.x * P2] + rel1[(uint64_t)blockIdx.x * P1_BINS + threadIdx.x];
    s_cnt[threadIdx.x] = 0;
  }
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  const int rcshift = 2 * (k - 1);
  const int shift1 = cfg.c_bits - 7;  // log2(P1_BINS) = 7
  const uint32_t n_chunks = (rv.n + P2_BLOCK - 1) / P2_BLOCK;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r = chunk * P2_BLOCK + threadIdx.x;
    const bool live = r < rv.n;
    const uint32_t len = live ? rv_len(rv, r) : 0;
    const uint32_t roff = live ? rv_off(rv, r) : 0;
    const uint64_t* cw = rv.codes + roff;
    const uint32_t* cm = live ? rv_acgt(rv, r, roff) : nullptr;
    if (threadIdx.x == 0) s_maxlen = 0;
    __syncthreads();
    if (len) atomicMax(&s_maxlen, len);
    __syncthreads();
    const uint32_t n_phase = (s_maxlen + P1_S - 1) / P1_S;
    uint64_t fwd = 0, rc = 0, cur_w = 0;
    uint32_t cur_m = 0;
    int filled = 0;
    for (uint32_t ph = 0; ph < n_phase; ++ph) {
      uint64_t wv[P1_S];
      uint32_t br[P1_S];  // (bin << 16) | rank, or ~0 when this base yields no word
      const uint32_t p0 = ph * P1_S;
      if ((ph & 3) == 0 && p0 < len) {  // 32 bases per code word = 4 phases
        cur_w = cw[p0 >> 5];
        cur_m = cm ? cm[p0 >> 5] : ~0u;
      }
#pragma unroll
      for (int b = 0; b < P1_S; ++b) {
        br[b] = ~0u;
        if (p0 + b < len) {
          const uint32_t code = (uint32_t)cur_w & 3u;
          cur_w >>= 2;
          const bool valid = cur_m & 1u;
          cur_m >>= 1;
          fwd = ((fwd << 2) | code) & kmask;
          if (CANON) rc = (rc >> 2) | ((uint64_t)(3u - code) << rcshift);
          filled = valid ? filled + 1 : 0;
          if (filled >= k) {
            const uint64_t key = CANON ? (rc < fwd ? rc : fwd) : fwd;
            const uint64_t w = gf2_mul(s_lut, key, ntab);
            const uint64_t pos = w >> cfg.sel_bits;
            if (pos >= pos_lo && pos < pos_hi) {
              const uint32_t bin = (uint32_t)(w >> shift1);
              wv[b] = w;
              br[b] = (bin << 16) | atomicAdd(&s_cnt[bin], 1u);
              if (FUSED) {  // no-return LDS atomic; a wrapped 16-bit counter shows up in the sum check below
                const uint32_t fb = (uint32_t)(w >> cfg.bin_shift);
                atomicAdd(&s_fine[fb >> 1], 1u << ((fb & 1u) * 16));
              }
            }
          }
        }
      }
      __syncthreads();
      // FUSED: reserve this phase's runs now; the round trip of the global atomic hides behind the
      // scan and the staging, its result is only needed for the write-out.
      uint32_t cn = 0, at = 0;
      if (FUSED && threadIdx.x < P1_BINS) {
        cn = s_cnt[threadIdx.x];
        if (cn) at = atomicAdd(&coarse_cur[threadIdx.x * P1_CUR_STRIDE], cn);
      }
      if (threadIdx.x < 64) wave_scan256(s_cnt, s_start, P1_BINS);
      __syncthreads();
      if (threadIdx.x < P1_BINS) {
        if (!FUSED) {
          s_gbase[threadIdx.x] = s_gcur[threadIdx.x];
          s_gcur[threadIdx.x] += s_cnt[threadIdx.x];
        }
        s_cnt[threadIdx.x] = 0;
      }
#pragma unroll
      for (int b = 0; b < P1_S; ++b)
        if (br[b] != ~0u) {
          const uint32_t bin = br[b] >> 16, e = s_start[bin] + (br[b] & 0xFFFFu);
          s_stage[e] = wv[b];
          s_sbin[e] = (uint8_t)bin;
        }
      if (FUSED && threadIdx.x < P1_BINS) {
        if (at + cn > cap_a) {  // bin over its capacity: drop the run (the block is redone exactly)
          atomicExch(flag, 1u);
          s_gbase[threadIdx.x] = 0xFFFFFFFFu;
        } else {
          s_gbase[threadIdx.x] = threadIdx.x * cap_a + at;
        }
      }
      __syncthreads();
      const uint32_t total = s_start[P1_BINS];
      if (FUSED) blk_total += total;
      for (uint32_t e = threadIdx.x; e < total; e += P2_BLOCK) {
        const uint32_t bin = s_sbin[e];
        if (!FUSED || s_gbase[bin] != 0xFFFFFFFFu) buf_a[(uint64_t)s_gbase[bin] + (e - s_start[bin])] = s_stage[e];
      }
      __syncthreads();
    }
  }
  if (FUSED) {  // this block's row of the exact fine-bin histogram
    __syncthreads();
    if (threadIdx.x == 0) s_maxlen = 0;
    __syncthreads();
    uint32_t sum = 0;
    for (uint32_t b = threadIdx.x; b < P; b += blockDim.x) {
      const uint32_t v = (s_fine[b >> 1] >> ((b & 1u) * 16)) & 0xFFFFu;
      cnt_rows[(uint64_t)blockIdx.x * P + b] = v;
      sum += v;
    }
    // A wrapped low half carries into its neighbour (sum - 65535), a wrapped high half carries out
    // of the word (sum - 65536): any wrap leaves the sum short of the words this block produced.
    atomicAdd(&s_maxlen, sum);
    __syncthreads();
    if (threadIdx.x == 0 && s_maxlen != blk_total) atomicExch(flag, 1u);
  }
}

// The sub-bin of an entry: MODE 0 -- bits of the word itself; MODE 1 / 2 -- an MSP super-k-mer record
// (canonical / not): bits of its minimizer bin hash, re-derived from the record.
template <int MODE>
__device__ __forceinline__ uint32_t sub_bin_of(uint64_t w, int shift2, uint32_t P2, int k) {
  if (MODE == 0) return (uint32_t)(w >> shift2) & (P2 - 1);
  return ((MODE == 1 ? msp_record_binhash<true>(w, k) : msp_record_binhash<false>(w, k)) >> shift2) & (P2 - 1);
}

// Sizes of the P2 sub-bins of every parent bin (extents ps[b] .. ps[b+1]); W workgroups share a parent.
template <int MODE>
__global__ __launch_bounds__(512) void k_bin_hist(const uint64_t* __restrict__ src, const uint64_t* __restrict__ ps,
                                                   uint32_t n_parents, uint32_t W, uint32_t P2, int shift2, int k,
                                                   unsigned long long* __restrict__ fine_tot) {
  __shared__ uint32_t s_cnt[256];
  for (uint32_t blk = blockIdx.x; blk < n_parents * W; blk += gridDim.x) {
    const uint32_t b = blk / W, j = blk - b * W;
    const uint64_t a = ps[b], e = ps[b + 1];
    if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // four loads in flight per lane instead of one (a refinement parent gives a lane ~40 trips): 48 -> 43 ms per W
    // sample, the 208 GB of records at 4.8 TB/s
    const uint64_t stride = (uint64_t)W * blockDim.x;
    uint64_t i = a + (uint64_t)j * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < e; i += 4 * stride) {
      const uint64_t w0 = src[i], w1 = src[i + stride], w2 = src[i + 2 * stride], w3 = src[i + 3 * stride];
      atomicAdd(&s_cnt[sub_bin_of<MODE>(w0, shift2, P2, k)], 1u);
      atomicAdd(&s_cnt[sub_bin_of<MODE>(w1, shift2, P2, k)], 1u);
      atomicAdd(&s_cnt[sub_bin_of<MODE>(w2, shift2, P2, k)], 1u);
      atomicAdd(&s_cnt[sub_bin_of<MODE>(w3, shift2, P2, k)], 1u);
    }
    for (; i < e; i += stride) atomicAdd(&s_cnt[sub_bin_of<MODE>(src[i], shift2, P2, k)], 1u);
    __syncthreads();
    if (threadIdx.x < P2 && s_cnt[threadIdx.x])
      atomicAdd(&fine_tot[(uint64_t)b * P2 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
  }
}

// The same from the records' STAMPED planes (rfx_devutil.h msp_stamp: k <= 25): 4 bytes per record instead of 8, no hash.
__global__ __launch_bounds__(512) void k_bin_hist_stamp(const uint32_t* __restrict__ ext, const uint64_t* __restrict__ ps,
                                                         uint32_t n_parents, uint32_t W, uint32_t P2, int shift2,
                                                         unsigned long long* __restrict__ fine_tot) {
  __shared__ uint32_t s_cnt[256];
  for (uint32_t blk = blockIdx.x; blk < n_parents * W; blk += gridDim.x) {
    const uint32_t b = blk / W, j = blk - b * W;
    const uint64_t a = ps[b], e = ps[b + 1];
    if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // four planes per load (a dwordx4 needs no more than the 4-byte alignment it has), two loads in flight per lane
    const uint64_t stride = (uint64_t)W * blockDim.x * 4;
    uint64_t i = a + ((uint64_t)j * blockDim.x + threadIdx.x) * 4;
    for (; i + stride + 4 <= e; i += 2 * stride) {
      const uint4 v0 = *(const uint4*)(ext + i), v1 = *(const uint4*)(ext + i + stride);
      atomicAdd(&s_cnt[msp_stamp_sub(v0.x, shift2, P2)], 1u);
      atomicAdd(&s_cnt[msp_stamp_sub(v0.y, shift2, P2)], 1u);
      atomicAdd(&s_cnt[msp_stamp_sub(v0.z, shift2, P2)], 1u);
      atomicAdd(&s_cnt[msp_stamp_sub(v0.w, shift2, P2)], 1u);
      atomicAdd(&s_cnt[msp_stamp_sub(v1.x, shift2, P2)], 1u);
      atomicAdd(&s_cnt[msp_stamp_sub(v1.y, shift2, P2)], 1u);
      atomicAdd(&s_cnt[msp_stamp_sub(v1.z, shift2, P2)], 1u);
      atomicAdd(&s_cnt[msp_stamp_sub(v1.w, shift2, P2)], 1u);
    }
    for (; i < e; i += stride)
      for (uint64_t q = i; q < i + 4 && q < e; ++q) atomicAdd(&s_cnt[msp_stamp_sub(ext[q], shift2, P2)], 1u);
    __syncthreads();
    if (threadIdx.x < P2 && s_cnt[threadIdx.x])
      atomicAdd(&fine_tot[(uint64_t)b * P2 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
  }
}

// Both of the above over the slices of nseg arrays in ONE launch (the refinement's first level: a chunk of a W sample is
// 19 read blocks' records -- a launch per block and chunk was 646 launches of ~17 us per sample, each with its ramp and
// the gap behind it).  seg_src / seg_ext / seg_ps: DEVICE arrays of nseg pointers; parent b = bin cs_off + b of every
// array.  STAMP: the counts come from the stamped planes (k <= 25), else from the words (MODE as k_bin_hist).
template <int MODE, bool STAMP>
__global__ __launch_bounds__(512) void k_bin_hist_multi(const uint64_t* const* __restrict__ seg_src, const uint32_t* const* __restrict__ seg_ext,
                                                         const uint64_t* const* __restrict__ seg_ps, int nseg, uint32_t cs_off,
                                                         uint32_t n_parents, uint32_t W, uint32_t P2, int shift2, int k,
                                                         unsigned long long* __restrict__ fine_tot) {
  __shared__ uint32_t s_cnt[256];
  for (uint32_t blk = blockIdx.x; blk < n_parents * W; blk += gridDim.x) {
    const uint32_t b = blk / W, j = blk - b * W;
    if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int sg = 0; sg < nseg; ++sg) {
      const uint64_t* __restrict__ ps = seg_ps[sg] + cs_off;
      const uint64_t a = ps[b], e = ps[b + 1];
      if (STAMP) {
        const uint32_t* __restrict__ ext = seg_ext[sg];
        const uint64_t stride = (uint64_t)W * blockDim.x * 4;
        uint64_t i = a + ((uint64_t)j * blockDim.x + threadIdx.x) * 4;
        for (; i + stride + 4 <= e; i += 2 * stride) {
          const uint4 v0 = *(const uint4*)(ext + i), v1 = *(const uint4*)(ext + i + stride);
          atomicAdd(&s_cnt[msp_stamp_sub(v0.x, shift2, P2)], 1u);
          atomicAdd(&s_cnt[msp_stamp_sub(v0.y, shift2, P2)], 1u);
          atomicAdd(&s_cnt[msp_stamp_sub(v0.z, shift2, P2)], 1u);
          atomicAdd(&s_cnt[msp_stamp_sub(v0.w, shift2, P2)], 1u);
          atomicAdd(&s_cnt[msp_stamp_sub(v1.x, shift2, P2)], 1u);
          atomicAdd(&s_cnt[msp_stamp_sub(v1.y, shift2, P2)], 1u);
          atomicAdd(&s_cnt[msp_stamp_sub(v1.z, shift2, P2)], 1u);
          atomicAdd(&s_cnt[msp_stamp_sub(v1.w, shift2, P2)], 1u);
        }
        for (; i < e; i += stride)
          for (uint64_t q = i; q < i + 4 && q < e; ++q) atomicAdd(&s_cnt[msp_stamp_sub(ext[q], shift2, P2)], 1u);
      } else {
        const uint64_t* __restrict__ src = seg_src[sg];
        const uint64_t stride = (uint64_t)W * blockDim.x;
        uint64_t i = a + (uint64_t)j * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < e; i += 4 * stride) {
          const uint64_t w0 = src[i], w1 = src[i + stride], w2 = src[i + 2 * stride], w3 = src[i + 3 * stride];
          atomicAdd(&s_cnt[sub_bin_of<MODE>(w0, shift2, P2, k)], 1u);
          atomicAdd(&s_cnt[sub_bin_of<MODE>(w1, shift2, P2, k)], 1u);
          atomicAdd(&s_cnt[sub_bin_of<MODE>(w2, shift2, P2, k)], 1u);
          atomicAdd(&s_cnt[sub_bin_of<MODE>(w3, shift2, P2, k)], 1u);
        }
        for (; i < e; i += stride) atomicAdd(&s_cnt[sub_bin_of<MODE>(src[i], shift2, P2, k)], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < P2 && s_cnt[threadIdx.x])
      atomicAdd(&fine_tot[(uint64_t)b * P2 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
  }
}

// Coarse bin cb of A -> its P2 fine bins in B.  W workgroups share a coarse bin (tiles strided).
// PAYLOAD: every word carries a 32-bit count that moves with it (survivors of the MSP leaf).
// MULTI: coarse bin cb is the concatenation of its slices in nseg arrays (the read blocks of a sample, each partitioned
// by itself: seg_a[s][seg_cs[s][cs_off + cb] .. seg_cs[s][cs_off + cb + 1])) -- one launch and full tiles instead of a
// launch per block whose last tile of every bin is mostly empty.
constexpr int L2_MAX_SEGS = 64;
template <bool PAYLOAD, int MODE, bool MULTI = false>
__global__ __launch_bounds__(L2_BLOCK) void k_part2(const uint64_t* __restrict__ buf_a, uint64_t* __restrict__ buf_b,
                                                     const uint64_t* __restrict__ fine_start,
                                                     uint32_t* __restrict__ fine_cur, uint32_t P2, int shift2,
                                                     uint32_t W, const uint32_t* __restrict__ coarse_cur,
                                                     uint32_t cap_a, const uint32_t* __restrict__ pay_a,
                                                     uint32_t* __restrict__ pay_b, uint64_t cap_b,
                                                     const uint64_t* __restrict__ coarse_start, int k,
                                                     uint64_t fine_base,
                                                     const uint64_t* const* __restrict__ seg_a = nullptr,
                                                     const uint64_t* const* __restrict__ seg_cs = nullptr,
                                                     const uint32_t* const* __restrict__ seg_pay = nullptr,
                                                     int nseg = 0, uint32_t cs_off = 0) {
  __shared__ uint64_t s_stage[L2_TILE];
  __shared__ uint32_t s_pay[PAYLOAD ? L2_TILE : 1];
  __shared__ uint8_t s_sbin[L2_TILE];
  __shared__ uint32_t s_cnt[256], s_start[257];
  __shared__ uint64_t s_gbase[256];
  __shared__ uint64_t s_soff[MULTI ? L2_MAX_SEGS + 1 : 1];         // where slice s begins in the concatenation
  __shared__ const uint64_t* s_sptr[MULTI ? L2_MAX_SEGS : 1];      // its first word
  __shared__ const uint32_t* s_spay[MULTI && PAYLOAD ? L2_MAX_SEGS : 1];
  // Workgroup i runs on XCD i % 8, each with an L2 of its own: the W workgroups of a coarse bin append to the SAME 256
  // output runs (neighbouring reservations: the partial lines where two of them meet merge in an L2 only if both writers
  // sit behind it), so they are given ids that land on one XCD.
#ifndef RFX_P2_NO_XCD  // (W, ms per sample: k_part2 94 -> 77, k_part3 94 -> 68, k_surv_part2 22 -> 16)
  uint32_t cb = blockIdx.x / W, j = blockIdx.x - cb * W;
  if (MULTI || coarse_start) {
    // a refinement: every parent bin has records, and the fine bins of neighbouring parents are neighbours in memory too --
    // XCD x takes a contiguous range of ids: [x * per + min(x, rem), ..), per + 1 of them for x < rem (a bijection for any grid)
    const uint32_t xcd = blockIdx.x & 7u, per = gridDim.x >> 3, rem = gridDim.x & 7u;
    const uint32_t bid = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    cb = bid / W;
    j = bid - cb * W;
  } else if (((gridDim.x / W) & 7u) == 0 && gridDim.x % W == 0) {
    // the 128 coarse bins of k_msp_part1 / of the leaf's survivors: a shard pass fills only a contiguous part of them (handing
    // XCD x the bins [16 x, 16 x + 16) left half the chip without work at W), so bin cb goes to XCD cb % 8, its W workgroups
    // are that XCD's slots W * (cb / 8) ..
    const uint32_t slot = blockIdx.x >> 3;
    cb = (slot / W) * 8u + (blockIdx.x & 7u);
    j = slot - (slot / W) * W;
  }
#else
  const uint32_t cb = blockIdx.x / W, j = blockIdx.x - cb * W;
#endif
  if (MULTI) {
    if (threadIdx.x < (uint32_t)nseg) {
      const uint64_t* cs = seg_cs[threadIdx.x] + cs_off;
      const uint64_t b0 = cs[cb], b1 = cs[cb + 1];
      s_soff[threadIdx.x + 1] = b1 - b0;
      s_sptr[threadIdx.x] = seg_a[threadIdx.x] + b0;
      if (PAYLOAD) s_spay[threadIdx.x] = seg_pay[threadIdx.x] + b0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      s_soff[0] = 0;
      for (int s = 0; s < nseg; ++s) s_soff[s + 1] += s_soff[s];
    }
    __syncthreads();
  }
  // coarse bin cb of A: given extents (refinement of an existing partition), fixed-capacity with a fill
  // cursor (fused path), or exactly sized = the extents of its fine bins in B
  const uint64_t a = MULTI ? 0 : coarse_start ? coarse_start[cb] : coarse_cur ? (uint64_t)cb * cap_a : fine_start[(uint64_t)cb * P2];
  const uint64_t e = MULTI ? s_soff[nseg] : coarse_start ? coarse_start[cb + 1]
                     : coarse_cur ? a + min(coarse_cur[cb * P1_CUR_STRIDE], cap_a)
                                  : fine_start[(uint64_t)(cb + 1) * P2];
  // MULTI: the slice a lane's next word lies in (its words come in rising order, so the cursor only moves on)
  int seg = 0;
  uint64_t seg_b = 0, seg_e = MULTI ? s_soff[1] : 0;
  if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t base = a + (uint64_t)j * L2_TILE; base < e; base += (uint64_t)W * L2_TILE) {
    uint64_t wv[L2_PER];
    uint32_t pv[PAYLOAD ? L2_PER : 1];
    uint32_t br[L2_PER];
#pragma unroll
    for (int u = 0; u < L2_PER; ++u) {
      const uint64_t i = base + threadIdx.x + (uint64_t)u * L2_BLOCK;
      if (MULTI) {
        wv[u] = 0;
        if (PAYLOAD) pv[u] = 0;
        if (i < e) {
          while (i >= seg_e) {  // (i < e = s_soff[nseg]: stops at a slice that holds i, empty slices passed over)
            ++seg;
            seg_b = seg_e;
            seg_e = s_soff[seg + 1];
          }
          wv[u] = s_sptr[seg][i - seg_b];
          if (PAYLOAD) pv[u] = s_spay[seg][i - seg_b];
        }
      } else if (MODE != 0 && PAYLOAD && coarse_cur) {  // k_msp_part1's coarse bins: word and plane side by side
        msp_rec12 rr{0u, 0u, 0u};
        if (i < e) rr = ((const msp_rec12*)buf_a)[i];
        wv[u] = (uint64_t)rr.lo | ((uint64_t)rr.hi << 32);
        pv[u] = rr.x;
      } else {
        wv[u] = i < e ? buf_a[i] : 0;
        if (PAYLOAD) pv[u] = i < e ? pay_a[i] : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < L2_PER; ++u) {
      br[u] = ~0u;
      // (MODE 1 / 2: MSP_EMPTY = a slot of a k_msp_part1 slab that nobody took)
      if (base + threadIdx.x + (uint64_t)u * L2_BLOCK < e && (MODE == 0 || wv[u] != MSP_EMPTY)) {
        const uint32_t sub = sub_bin_of<MODE>(wv[u], shift2, P2, k);
        br[u] = (sub << 16) | atomicAdd(&s_cnt[sub], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) wave_scan256(s_cnt, s_start, P2);
    __syncthreads();
    if (threadIdx.x < P2) {
      const uint32_t c = s_cnt[threadIdx.x];
      const uint64_t f = (uint64_t)cb * P2 + threadIdx.x;
      const uint32_t at = c ? atomicAdd(&fine_cur[f], c) : 0u;
      // never write past the fine bin (only possible after the fused part1 raised its flag)
      s_gbase[threadIdx.x] = fine_start[f] + at + c <= fine_start[f + 1] && fine_start[f + 1] - fine_base <= cap_b
                                 ? fine_start[f] - fine_base + at
                                 : ~0ull;
      s_cnt[threadIdx.x] = 0;
    }
#pragma unroll
    for (int u = 0; u < L2_PER; ++u)
      if (br[u] != ~0u) {
        const uint32_t sub = br[u] >> 16, x = s_start[sub] + (br[u] & 0xFFFFu);
        s_stage[x] = wv[u];
        if (PAYLOAD) s_pay[x] = pv[u];
        s_sbin[x] = (uint8_t)sub;
      }
    __syncthreads();
    const uint32_t total = s_start[P2];
    for (uint32_t x = threadIdx.x; x < total; x += L2_BLOCK) {
      const uint32_t sub = s_sbin[x];
      if (s_gbase[sub] != ~0ull) {
        buf_b[s_gbase[sub] + (x - s_start[sub])] = s_stage[x];
        if (PAYLOAD) pay_b[s_gbase[sub] + (x - s_start[sub])] = s_pay[x];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_tmp_start(const uint64_t* const* __restrict__ seg_bs, int nseg, uint32_t P,
                                                    uint64_t* __restrict__ tmp_start) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > P) return;
  uint64_t s = 0;
  for (int i = 0; i < nseg; ++i) s += seg_bs[i][b];
  tmp_start[b] = s;
}

// One workgroup per bin.  If a bin holds more distinct words than the LDS table (or more survivors
// than the sort area) it is re-run split into 2^r sub-ranges of w, in order -- exact for any input.
__global__ __launch_bounds__(LEAF_BLOCK) void k_leaf(const uint64_t* const* __restrict__ seg_inst,
                                                      const uint64_t* const* __restrict__ seg_bs, int nseg,
                                                      const uint64_t* __restrict__ inst0,
                                                      const uint64_t* __restrict__ bs0, uint32_t P,
                                                      rfx_ord_cfg cfg, uint64_t lower, uint64_t upper,
                                                      const uint64_t* __restrict__ tmp_start,
                                                      uint64_t* __restrict__ tmp_w, uint32_t* __restrict__ tmp_counts,
                                                      uint64_t* __restrict__ n_surv, unsigned int* __restrict__ err) {
  __shared__ unsigned long long s_keys[LEAF_TBL];  // hash table keys; reused as the bucketed survivor words
  __shared__ uint32_t s_cnt[LEAF_TBL];             // hash table counts; reused as the bucketed survivor counts
  __shared__ uint64_t s_w[LEAF_SORT];
  __shared__ uint32_t s_c[LEAF_SORT];
  __shared__ uint32_t s_bstart[LEAF_BUCKETS + 1], s_bfill[LEAF_BUCKETS];
  __shared__ uint32_t s_nd, s_ns, s_ovf;
  const int bin_bits = cfg.c_bits - cfg.bin_shift;  // bins are the top bits of the c-bit word

  // First batch of segment 0 of a bin (inst0/bs0 are segment 0's arrays passed by value, no pointer
  // chase).  Issued one bin ahead so the HBM latency hides behind the previous bin's sort and emit.
  uint64_t pre[LEAF_ILP];
  uint64_t pre_a = 0, pre_e = 0, pre_out0 = 0;
  auto prefetch = [&](uint32_t b) {
    if (b >= P) return;
    pre_a = bs0[b];
    pre_e = bs0[b + 1];
    pre_out0 = tmp_start[b];
#pragma unroll
    for (int u = 0; u < LEAF_ILP; ++u) {
      const uint64_t i = pre_a + threadIdx.x + (uint64_t)u * LEAF_BLOCK;
      pre[u] = i < pre_e ? inst0[i] : RFX_EMPTY;
    }
  };
  prefetch(blockIdx.x);

  for (uint32_t bin = blockIdx.x; bin < P; bin += gridDim.x) {
    const uint64_t out0 = pre_out0, a0 = pre_a, e0 = pre_e;
    bool prefetched_next = false;  // until then `pre` holds this bin's first batch
    uint64_t emitted = 0;
    bool done = false;
    for (int r = 0; r <= LEAF_RMAX && !done && bin_bits + r <= cfg.c_bits; ++r) {
      emitted = 0;
      bool ok = true;
      const int sub_shift = cfg.c_bits - bin_bits - r;  // bits of w below the (bin, round) prefix
      for (uint32_t j = 0; j < (1u << r) && ok; ++j) {
        for (int i = threadIdx.x; i < LEAF_TBL; i += LEAF_BLOCK) {
          s_keys[i] = RFX_EMPTY;
          s_cnt[i] = 0;
        }
        if (threadIdx.x < LEAF_BUCKETS) s_bfill[threadIdx.x] = 0;
        if (threadIdx.x == 0) {
          s_nd = 0;
          s_ns = 0;
          s_ovf = 0;
        }
        __syncthreads();
        for (int sg = 0; sg < nseg; ++sg) {
          const uint64_t a = sg == 0 ? a0 : seg_bs[sg][bin], e = sg == 0 ? e0 : seg_bs[sg][bin + 1];
          const uint64_t* __restrict__ src = sg == 0 ? inst0 : seg_inst[sg];
          for (uint64_t base = a; base < e; base += (uint64_t)LEAF_ILP * LEAF_BLOCK) {
            uint64_t w[LEAF_ILP];
            if (sg == 0 && base == a && !prefetched_next) {  // first pass over the bin: already in registers
#pragma unroll
              for (int u = 0; u < LEAF_ILP; ++u) w[u] = pre[u];
            } else {
#pragma unroll
              for (int u = 0; u < LEAF_ILP; ++u) {  // independent loads, all in flight before the first insert
                const uint64_t i = base + threadIdx.x + (uint64_t)u * LEAF_BLOCK;
                w[u] = i < e ? src[i] : RFX_EMPTY;
              }
            }
#pragma unroll
            for (int u = 0; u < LEAF_ILP; ++u) {
              const uint64_t key = w[u];
              // The count of distinct words is looked at before EVERY insert: at most one word per thread
              // can follow LEAF_FILL, which the LEAF_TBL - LEAF_FILL spare slots absorb -- the probe loop
              // below always terminates.  A skipped insert voids the pass (s_ovf).  The probe is one
              // returning CAS ("was empty, now mine" / "already mine" / "someone else's"), as in k_msp_leaf.
              if (key == RFX_EMPTY) continue;
              if (r > 0 && (uint32_t)((key >> sub_shift) & ((1u << r) - 1)) != j) continue;
              if (__hip_atomic_load(&s_nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (uint32_t)LEAF_FILL) {
                s_ovf = 1;
                break;
              }
              uint32_t slot = leaf_hash(key);
              for (;;) {
                unsigned long long old = atomicCAS(&s_keys[slot], (unsigned long long)RFX_EMPTY, (unsigned long long)key);
                if (old == RFX_EMPTY) {
                  atomicAdd(&s_nd, 1u);
                  old = key;
                }
                if (old == key) {
                  atomicAdd(&s_cnt[slot], 1u);
                  break;
                }
                slot = (slot + 1) & (LEAF_TBL - 1);
              }
            }
          }
        }
        if (!prefetched_next) {  // the next bin's loads fly while this bin is ranked and written out
          prefetch(bin + gridDim.x);
          prefetched_next = true;
        }
        __syncthreads();
        if (s_ovf) {
          ok = false;
          break;
        }
        // survivors -> staging area, counting them per bucket (next bits of w below the prefix)
        const int bsh = sub_shift > 8 ? sub_shift - 8 : 0;
        for (int i = threadIdx.x; i < LEAF_TBL; i += LEAF_BLOCK) {
          const uint64_t key = s_keys[i];
          if (key == RFX_EMPTY) continue;
          const uint32_t c = s_cnt[i];
          if (c < lower || c > upper) continue;
          const uint32_t o = atomicAdd(&s_ns, 1u);
          if (o < (uint32_t)LEAF_SORT) {
            s_w[o] = key;
            s_c[o] = c;
            atomicAdd(&s_bfill[(uint32_t)(key >> bsh) & (LEAF_BUCKETS - 1)], 1u);
          }
        }
        __syncthreads();
        const uint32_t ns = s_ns;
        if (ns > (uint32_t)LEAF_SORT) {
          ok = false;
          break;
        }
        // exclusive scan of the 256 bucket sizes by one wave (4 buckets per lane)
        if (threadIdx.x < 64) {
          const uint32_t l = threadIdx.x;
          const uint32_t c0 = s_bfill[4 * l], c1 = s_bfill[4 * l + 1], c2 = s_bfill[4 * l + 2], c3 = s_bfill[4 * l + 3];
          uint32_t sum = c0 + c1 + c2 + c3, inc = sum;
          for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if ((int)l >= off) inc += o;
          }
          const uint32_t ex = inc - sum;
          s_bstart[4 * l] = ex;
          s_bstart[4 * l + 1] = ex + c0;
          s_bstart[4 * l + 2] = ex + c0 + c1;
          s_bstart[4 * l + 3] = ex + c0 + c1 + c2;
          if (l == 63) s_bstart[LEAF_BUCKETS] = inc;
          s_bfill[4 * l] = s_bfill[4 * l + 1] = s_bfill[4 * l + 2] = s_bfill[4 * l + 3] = 0;
        }
        __syncthreads();
        // group by bucket (the table arrays are free now), then rank inside the bucket
        uint64_t* s_w2 = (uint64_t*)s_keys;
        uint32_t* s_c2 = s_cnt;
        for (uint32_t i = threadIdx.x; i < ns; i += LEAF_BLOCK) {
          const uint64_t wi = s_w[i];
          const uint32_t bk = (uint32_t)(wi >> bsh) & (LEAF_BUCKETS - 1);
          const uint32_t p = s_bstart[bk] + atomicAdd(&s_bfill[bk], 1u);
          s_w2[p] = wi;
          s_c2[p] = s_c[i];
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < ns; i += LEAF_BLOCK) {
          const uint64_t wi = s_w2[i];
          const uint32_t bk = (uint32_t)(wi >> bsh) & (LEAF_BUCKETS - 1);
          const uint32_t b0 = s_bstart[bk], b1 = s_bstart[bk + 1];
          uint32_t rank = b0;
          for (uint32_t q = b0; q < b1; ++q) rank += s_w2[q] < wi;  // the words are distinct
          tmp_w[out0 + emitted + rank] = wi;
          tmp_counts[out0 + emitted + rank] = s_c2[i];
        }
        emitted += ns;
        __syncthreads();
      }
      if (ok) done = true;
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      n_surv[bin] = done ? emitted : 0;
      if (!done) atomicExch(err, 1u);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_leaf_compact(const uint64_t* __restrict__ tmp_w,
                                                       const uint32_t* __restrict__ tmp_counts,
                                                       const uint64_t* __restrict__ tmp_start,
                                                       const uint64_t* __restrict__ out_off, uint32_t P,
                                                       const uint64_t* __restrict__ g_lut_inv, int ntab, int sel_bits,
                                                       uint64_t* __restrict__ out_keys, uint32_t* __restrict__ out_counts,
                                                       uint64_t* __restrict__ out_pos) {
  __shared__ uint64_t s_lut[8 * 256];
  for (int i = threadIdx.x; i < ntab * 256; i += blockDim.x) s_lut[i] = g_lut_inv[i];
  __syncthreads();
  for (uint32_t bin = blockIdx.x; bin < P; bin += gridDim.x) {
    const uint64_t src = tmp_start[bin], dst = out_off[bin], n = out_off[bin + 1] - dst;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
      const uint64_t w = tmp_w[src + i];
      out_keys[dst + i] = gf2_mul(s_lut, w, ntab);
      out_counts[dst + i] = tmp_counts[src + i];
      out_pos[dst + i] = w >> sel_bits;
    }
  }
}

// Bin extents of one partition cut in two at record n_own (rfx_count_set_early: the bins below the cut are one table's
// segment, the bins above it another's): bs keeps the first part (later bins empty at its end), bs2 gets the second,
// rebased to its own array (earlier bins empty at its start).
__global__ __launch_bounds__(256) void k_split_bins(uint64_t* __restrict__ bs, uint32_t P, uint64_t n_own,
                                                     uint64_t* __restrict__ bs2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > P) return;
  const uint64_t v = bs[i];
  bs2[i] = v > n_own ? v - n_own : 0;
  bs[i] = v < n_own ? v : n_own;
}

// exclusive scan of v[0..n) in place, v[n] = total (one block): every thread owns a contiguous chunk,
// the 1024 chunk sums are scanned with wave shuffles (two levels) -- a handful of barriers for any n
__global__ __launch_bounds__(1024) void k_scan_tail(uint64_t* __restrict__ v, uint64_t n) {
  __shared__ uint64_t s_wave[16];
  const uint64_t per = (n + 1023) / 1024;
  const uint64_t a = threadIdx.x * per, e = a + per < n ? a + per : n;
  uint64_t sum = 0;
  for (uint64_t i = a; i < e; ++i) sum += v[i];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint64_t inc = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_wave[wv] = inc;
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint64_t w = lane < 16 ? s_wave[lane] : 0;
    uint64_t winc = w;
    for (int off = 1; off < 16; off <<= 1) {
      const uint64_t o = __shfl_up(winc, off);
      if (lane >= off) winc += o;
    }
    if (lane < 16) s_wave[lane] = winc - w;  // exclusive prefix of the wave totals
    if (lane == 15) v[n] = winc;
  }
  __syncthreads();
  uint64_t run = s_wave[wv] + inc - sum;
  for (uint64_t i = a; i < e; ++i) {
    const uint64_t x = v[i];
    v[i] = run;
    run += x;
  }
}

// Large scans (the bin tables of a refinement chunk hold millions of entries): block sums, a one-block scan of
// those, then every block scans its 8192 entries from its offset.  v[n] = total as in k_scan_tail.
constexpr int SCAN_CHUNK = 8192;
__global__ __launch_bounds__(1024) void k_scan_sums(const uint64_t* __restrict__ v, uint64_t n, uint64_t* __restrict__ sums) {
  __shared__ uint64_t s_w[16];
  const uint64_t a = (uint64_t)blockIdx.x * SCAN_CHUNK;
  uint64_t x = 0;
  for (int u = 0; u < SCAN_CHUNK / 1024; ++u) {
    const uint64_t i = a + (uint64_t)u * 1024 + threadIdx.x;
    if (i < n) x += v[i];
  }
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t t = 0;
    for (int i = 0; i < 16; ++i) t += s_w[i];
    sums[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(1024) void k_scan_apply(uint64_t* __restrict__ v, uint64_t n, const uint64_t* __restrict__ sums,
                                                      uint32_t n_blocks) {
  __shared__ uint64_t s_w[16];
  const uint64_t a = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * (SCAN_CHUNK / 1024);
  uint64_t loc[SCAN_CHUNK / 1024], sum = 0;
#pragma unroll
  for (int u = 0; u < SCAN_CHUNK / 1024; ++u) {
    loc[u] = a + u < n ? v[a + u] : 0;
    sum += loc[u];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint64_t inc = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  uint64_t base = sums[blockIdx.x];  // exclusive prefix of the block sums (scanned in place by k_scan_tail)
  for (int i = 0; i < wv; ++i) base += s_w[i];
  uint64_t run = base + inc - sum;
#pragma unroll
  for (int u = 0; u < SCAN_CHUNK / 1024; ++u) {
    if (a + u < n) v[a + u] = run;
    run += loc[u];
  }
  if (blockIdx.x == n_blocks - 1 && threadIdx.x == 1023) v[n] = sums[n_blocks];
}

}  // namespace

namespace rfxk {

int p2l_grid(rfx_ctx* c, uint32_t n_reads) {
  const uint32_t n_chunks = (n_reads + P2_BLOCK - 1) / P2_BLOCK;
  uint32_t g = (uint32_t)c->n_cu * 3;
  if (n_chunks < g) g = n_chunks;
  g = (g + 7) & ~7u;  // whole XCD groups
  return (int)(g < 8 ? 8 : g);
}

void bin_count(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* lut, int ntab, int k, int canonical,
               const rfx_ord_cfg& cfg, uint32_t P, uint64_t pos_lo, uint64_t pos_hi, int grid, uint32_t* cnt) {
  const size_t lds = 8 * 256 * 8 + (size_t)P * 4;
  // 16 K bins: 80 KB of dynamic LDS
  if (!rfxi::lds_opt_in(c, (const void*)k_bin_count<true>, 96 * 1024, 2, "k_bin_count") ||
      !rfxi::lds_opt_in(c, (const void*)k_bin_count<false>, 96 * 1024, 3, "k_bin_count"))
    return;
  rfx_span sp(c, "k_bin_count");
  if (canonical)
    hipLaunchKernelGGL(k_bin_count<true>, dim3(grid), dim3(P2_BLOCK), lds, c->stream, rv, lut, ntab, k, cfg, P, pos_lo,
                       pos_hi, cnt);
  else
    hipLaunchKernelGGL(k_bin_count<false>, dim3(grid), dim3(P2_BLOCK), lds, c->stream, rv, lut, ntab, k, cfg, P, pos_lo,
                       pos_hi, cnt);
}

void bin_totals(rfx_ctx* c, const uint32_t* cnt, uint32_t G, uint32_t P, uint64_t* bin_start) {
  rfx_span sp(c, "k_bin_offsets");
  hipLaunchKernelGGL(k_col_sums, dim3((P + 63) / 64), dim3(1024), 0, c->stream, cnt, G, P, bin_start);
  hipLaunchKernelGGL(k_scan_tail, dim3(1), dim3(1024), 0, c->stream, bin_start, (uint64_t)P);
}

void bin_offsets(rfx_ctx* c, uint32_t* cnt, uint32_t G, uint32_t P, uint32_t* gsum, uint64_t* bin_start) {
  rfx_span sp(c, "k_bin_offsets");
  const uint32_t nb = (8 * P + 255) / 256;
  hipLaunchKernelGGL(k_bin_group_sums, dim3(nb), dim3(256), 0, c->stream, cnt, G, P, gsum);
  hipLaunchKernelGGL(k_bin_offsets, dim3(nb), dim3(256), 0, c->stream, cnt, G, P, gsum, bin_start);
  hipLaunchKernelGGL(k_scan_tail, dim3(1), dim3(1024), 0, c->stream, bin_start, (uint64_t)P);
}

void bin_scatter(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* lut, int ntab, int k, int canonical,
                 const rfx_ord_cfg& cfg, uint32_t P, uint64_t pos_lo, uint64_t pos_hi, int grid, const uint32_t* rel,
                 const uint64_t* bin_start, uint64_t* inst) {
  const size_t lds = 8 * 256 * 8 + (size_t)P * 4;
  rfx_span sp(c, "k_bin_scatter");
  if (canonical)
    hipLaunchKernelGGL(k_bin_scatter<true>, dim3(grid), dim3(P2_BLOCK), lds, c->stream, rv, lut, ntab, k, cfg, P,
                       pos_lo, pos_hi, rel, bin_start, inst);
  else
    hipLaunchKernelGGL(k_bin_scatter<false>, dim3(grid), dim3(P2_BLOCK), lds, c->stream, rv, lut, ntab, k, cfg, P,
                       pos_lo, pos_hi, rel, bin_start, inst);
}

int p1_bins() { return P1_BINS; }
int p1_cur_stride() { return P1_CUR_STRIDE; }

void coarse_counts(rfx_ctx* c, const uint32_t* cnt, uint32_t G, uint32_t P, uint32_t P2, uint32_t* cnt1) {
  const uint32_t n = G * (P / P2);
  hipLaunchKernelGGL(k_coarse_counts, dim3((n + 255) / 256), dim3(256), 0, c->stream, cnt, G, P, P2, cnt1);
}

void part1(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* lut, int ntab, int k, int canonical,
           const rfx_ord_cfg& cfg, uint32_t P2, uint64_t pos_lo, uint64_t pos_hi, int grid, const uint32_t* rel1,
           const uint64_t* fine_start, uint64_t* buf_a) {
  rfx_span sp(c, "k_part1");
  if (canonical)
    hipLaunchKernelGGL((k_part1<true, false>), dim3(grid), dim3(P2_BLOCK), 0, c->stream, rv, lut, ntab, k, cfg, P2,
                       pos_lo, pos_hi, rel1, fine_start, buf_a, (uint32_t*)nullptr, 0u, (uint32_t*)nullptr,
                       (unsigned int*)nullptr);
  else
    hipLaunchKernelGGL((k_part1<false, false>), dim3(grid), dim3(P2_BLOCK), 0, c->stream, rv, lut, ntab, k, cfg, P2,
                       pos_lo, pos_hi, rel1, fine_start, buf_a, (uint32_t*)nullptr, 0u, (uint32_t*)nullptr,
                       (unsigned int*)nullptr);
}

void part1_fused(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* lut, int ntab, int k, int canonical,
                 const rfx_ord_cfg& cfg, uint32_t P2, uint64_t pos_lo, uint64_t pos_hi, int grid, uint64_t* buf_a,
                 uint32_t* coarse_cur, uint32_t cap_a, uint32_t* cnt_rows, unsigned int* flag) {
  rfx_span sp(c, "k_part1");
  if (canonical)
    hipLaunchKernelGGL((k_part1<true, true>), dim3(grid), dim3(P2_BLOCK), 0, c->stream, rv, lut, ntab, k, cfg, P2,
                       pos_lo, pos_hi, (const uint32_t*)nullptr, (const uint64_t*)nullptr, buf_a, coarse_cur, cap_a,
                       cnt_rows, flag);
  else
    hipLaunchKernelGGL((k_part1<false, true>), dim3(grid), dim3(P2_BLOCK), 0, c->stream, rv, lut, ntab, k, cfg, P2,
                       pos_lo, pos_hi, (const uint32_t*)nullptr, (const uint64_t*)nullptr, buf_a, coarse_cur, cap_a,
                       cnt_rows, flag);
}

void part2(rfx_ctx* c, const uint64_t* buf_a, uint64_t* buf_b, const uint64_t* fine_start, uint32_t* fine_cur,
           uint32_t P2, int shift2, const uint32_t* coarse_cur, uint32_t cap_a, const uint32_t* pay_a,
           uint32_t* pay_b, uint64_t cap_b, const char* span, const uint64_t* coarse_start, uint32_t n_coarse,
           uint64_t n_hint, int rec_mode, int k, uint64_t fine_base) {
  rfx_span sp(c, span);
  // 128 coarse bins: up to 16 workgroups share one (two 8192-entry tiles each when the input is small);
  // thousands (refinement): one each
  const uint32_t nc = coarse_start ? n_coarse : (uint32_t)P1_BINS;
  // Workgroups per coarse bin: as many as are resident at once (two per CU), each looping over its share
  // of the 8192-entry tiles (measured on one box, 4 / 8 / 16 per bin: 0.120 / 0.125 / 0.131 ms for the MSP
  // records, 0.470 / 0.471 / 0.478 ms for the P2L words); fewer when the input has only a few tiles.
#ifndef RFX_P2_WMUL
#define RFX_P2_WMUL 8  // (16 workgroups per coarse bin instead of 4 now that they share an L2: k_part2 77 -> 64 ms per W sample)
#endif
  uint32_t W = nc >= 2048 ? 1 : std::max(1u, (uint32_t)c->n_cu * RFX_P2_WMUL / nc);
  if (n_hint && nc < 2048) {
    const uint64_t tiles = (n_hint / nc + L2_TILE - 1) / L2_TILE;
    W = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(W, (tiles + 1) / 2));
  }
#define RFX_PART2(PAY, MODE)                                                                                          \
  hipLaunchKernelGGL((k_part2<PAY, MODE>), dim3(nc * W), dim3(L2_BLOCK), 0, c->stream, buf_a, buf_b, fine_start,       \
                     fine_cur, P2, shift2, W, coarse_cur, cap_a, pay_a, pay_b, cap_b, coarse_start, k, fine_base)
  // payload: counts of survivors / the plane of super-k-mer records (coming out of k_msp_part1's coarse bins it sits
  // beside the word: pay_a is null then)
  const bool pay = pay_a || (rec_mode != 0 && pay_b);
  if (pay && rec_mode == 0) RFX_PART2(true, 0);
  else if (pay && rec_mode == 1) RFX_PART2(true, 1);
  else if (pay) RFX_PART2(true, 2);
  else if (rec_mode == 0) RFX_PART2(false, 0);
  else if (rec_mode == 1) RFX_PART2(false, 1);
  else RFX_PART2(false, 2);
#undef RFX_PART2
}

static bool bin_hist_stamped(int k, uint32_t P2, int shift2, int rec_mode, bool have_ext) {
  return rec_mode != 0 && have_ext && msp_stamped(k) && shift2 >= MSP_STAMP_LO &&
         shift2 + (32 - __builtin_clz(P2 - 1 ? P2 - 1 : 1)) <= MSP_STAMP_LO + MSP_STAMP_BITS && !getenv("RFX_NO_STAMP_HIST");
}

// ONE launch over the slices of nseg arrays (device arrays of pointers, as part2_multi takes them; seg_ext may be null)
void bin_hist_multi(rfx_ctx* c, const uint64_t* const* seg_src, const uint64_t* const* seg_ps, const uint32_t* const* seg_ext,
                    int nseg, uint32_t cs_off, uint32_t n_parents, uint64_t n_hint, uint32_t P2, int shift2, int rec_mode, int k,
                    uint64_t* fine_tot) {
  rfx_span sp(c, "k_bin_hist");
  if (!n_parents || nseg <= 0) return;
  const uint32_t resident = (uint32_t)c->n_cu * 4;
  uint32_t W = n_parents >= resident ? 1 : resident / n_parents;
  const uint64_t per = n_hint / n_parents / 2048 + 1;  // a workgroup should see a few thousand entries
  if (W > per) W = (uint32_t)per;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)n_parents * W, (uint64_t)resident * 8);
#define RFX_BHM(MODE, STAMP)                                                                                              \
  hipLaunchKernelGGL((k_bin_hist_multi<MODE, STAMP>), dim3(grid), dim3(512), 0, c->stream, seg_src, seg_ext, seg_ps, nseg, \
                     cs_off, n_parents, W, P2, shift2, k, (unsigned long long*)fine_tot)
  if (bin_hist_stamped(k, P2, shift2, rec_mode, seg_ext != nullptr)) RFX_BHM(1, true);
  else if (rec_mode == 0) RFX_BHM(0, false);
  else if (rec_mode == 1) RFX_BHM(1, false);
  else RFX_BHM(2, false);
#undef RFX_BHM
}

void bin_hist(rfx_ctx* c, const uint64_t* src, const uint64_t* parent_start, uint32_t n_parents, uint64_t n_hint,
              uint32_t P2, int shift2, int rec_mode, int k, uint64_t* fine_tot, const uint32_t* ext) {
  rfx_span sp(c, "k_bin_hist");
  if (!n_parents) return;
  const uint32_t resident = (uint32_t)c->n_cu * 4;
  uint32_t W = n_parents >= resident ? 1 : resident / n_parents;
  const uint64_t per = n_hint / n_parents / 2048 + 1;  // a workgroup should see a few thousand entries
  if (W > per) W = (uint32_t)per;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)n_parents * W, (uint64_t)resident * 8);
  // super-k-mer records whose planes carry bits 3 .. 18 of the bin hash (k <= 25), when those are the bits asked for
  if (bin_hist_stamped(k, P2, shift2, rec_mode, ext != nullptr)) {
    hipLaunchKernelGGL(k_bin_hist_stamp, dim3(grid), dim3(512), 0, c->stream, ext, parent_start, n_parents, W, P2, shift2,
                       (unsigned long long*)fine_tot);
    return;
  }
#define RFX_BH(MODE)                                                                                               \
  hipLaunchKernelGGL(k_bin_hist<MODE>, dim3(grid), dim3(512), 0, c->stream, src, parent_start, n_parents, W, P2, shift2, \
                     k, (unsigned long long*)fine_tot)
  if (rec_mode == 0) RFX_BH(0);
  else if (rec_mode == 1) RFX_BH(1);
  else RFX_BH(2);
#undef RFX_BH
}

// One launch for the slices of all `nseg` arrays (k_part2<.., MULTI>): seg_a / seg_cs / seg_pay are DEVICE arrays of
// nseg pointers (records, bin extents, planes or null); coarse bin cb = bins cs_off + cb of every array.
void part2_multi(rfx_ctx* c, const uint64_t* const* seg_a, const uint64_t* const* seg_cs, const uint32_t* const* seg_pay,
                 int nseg, uint32_t cs_off, uint32_t n_coarse, uint64_t n_hint, uint64_t* buf_b, const uint64_t* fine_start,
                 uint32_t* fine_cur, uint32_t P2, int shift2, uint32_t* pay_b, int rec_mode, int k, const char* span) {
  rfx_span sp(c, span);
  if (!n_coarse) return;
  // a handful of tiles per workgroup: the launch ends in a tail one workgroup long
  const uint64_t tiles = n_hint / n_coarse / L2_TILE + 1;
#ifndef RFX_P3_TDIV
#define RFX_P3_TDIV 3
#endif
  const uint32_t W = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(16, tiles / RFX_P3_TDIV));
#define RFX_PART2M(PAY, MODE)                                                                                          \
  hipLaunchKernelGGL((k_part2<PAY, MODE, true>), dim3(n_coarse * W), dim3(L2_BLOCK), 0, c->stream,                     \
                     (const uint64_t*)nullptr, buf_b, fine_start, fine_cur, P2, shift2, W, (const uint32_t*)nullptr, 0u, \
                     (const uint32_t*)nullptr, pay_b, ~0ull, (const uint64_t*)nullptr, k, (uint64_t)0, seg_a, seg_cs,  \
                     seg_pay, nseg, cs_off)
  if (seg_pay && rec_mode == 0) RFX_PART2M(true, 0);
  else if (seg_pay && rec_mode == 1) RFX_PART2M(true, 1);
  else if (seg_pay) RFX_PART2M(true, 2);
  else if (rec_mode == 0) RFX_PART2M(false, 0);
  else if (rec_mode == 1) RFX_PART2M(false, 1);
  else RFX_PART2M(false, 2);
#undef RFX_PART2M
}

void tmp_start(rfx_ctx* c, const uint64_t* const* seg_bs, int nseg, uint32_t P, uint64_t* out) {
  hipLaunchKernelGGL(k_tmp_start, dim3((P + 1 + 255) / 256), dim3(256), 0, c->stream, seg_bs, nseg, P, out);
}

void leaf(rfx_ctx* c, const uint64_t* const* seg_inst, const uint64_t* const* seg_bs, int nseg, const uint64_t* inst0,
          const uint64_t* bs0, uint32_t P, const rfx_ord_cfg& cfg, uint64_t lower, uint64_t upper,
          const uint64_t* tmp_start_, uint64_t* tmp_w, uint32_t* tmp_counts, uint64_t* n_surv, unsigned int* err) {
  rfx_span sp(c, "k_leaf");
  // a few bins per workgroup so that the one-bin-ahead prefetch has something to overlap with
  const uint32_t grid = P < (uint32_t)c->n_cu * 4 ? P : (uint32_t)c->n_cu * 4;
  hipLaunchKernelGGL(k_leaf, dim3(grid), dim3(LEAF_BLOCK), 0, c->stream, seg_inst, seg_bs, nseg, inst0, bs0, P, cfg,
                     lower, upper, tmp_start_, tmp_w, tmp_counts, n_surv, err);
}

void split_bins(rfx_ctx* c, uint64_t* bs, uint32_t P, uint64_t n_own, uint64_t* bs2) {
  hipLaunchKernelGGL(k_split_bins, dim3((P + 1 + 255) / 256), dim3(256), 0, c->stream, bs, P, n_own, bs2);
}

void scan_tail(rfx_ctx* c, uint64_t* v, uint64_t n) {
  if (n <= 65536) {
    hipLaunchKernelGGL(k_scan_tail, dim3(1), dim3(1024), 0, c->stream, v, n);
    return;
  }
  // one block would take milliseconds on the multi-million-entry bin tables of WGS-scale refinement chunks
  const uint32_t nb = (uint32_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
  uint64_t* sums = (uint64_t*)rfxi::dmalloc(c, ((size_t)nb + 1) * 8);
  if (!sums) {  // no scratch: the slow way is still correct
    hipLaunchKernelGGL(k_scan_tail, dim3(1), dim3(1024), 0, c->stream, v, n);
    return;
  }
  hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(1024), 0, c->stream, v, n, sums);
  hipLaunchKernelGGL(k_scan_tail, dim3(1), dim3(1024), 0, c->stream, sums, (uint64_t)nb);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(1024), 0, c->stream, v, n, sums, nb);
  rfxi::dfree(c, sums);  // stream-ordered
}

void leaf_compact(rfx_ctx* c, const uint64_t* tmp_w, const uint32_t* tmp_counts, const uint64_t* tmp_start_,
                  const uint64_t* out_off, uint32_t P, const uint64_t* lut_inv, int ntab, int sel_bits,
                  uint64_t* out_keys, uint32_t* out_counts, uint64_t* out_pos) {
  rfx_span sp(c, "k_leaf_compact");
  const uint32_t grid = P < (uint32_t)c->n_cu * 8 ? P : (uint32_t)c->n_cu * 8;
  hipLaunchKernelGGL(k_leaf_compact, dim3(grid), dim3(256), 0, c->stream, tmp_w, tmp_counts, tmp_start_, out_off, P,
                     lut_inv, ntab, sel_bits, out_keys, out_counts, out_pos);
}

}  // namespace rfxk
