// K6: pairwise overlap scoring of the greedy assemblers, K7: per-base mutant k-mer coverage.
//
// K6 replaces the body of Align3 (src/OverlapSam.cpp:33-241, src/Overlap.cpp:169-360,
// src/OverlapRegion.cpp:31-231): for one query A and a list of candidates B_j, every alignment
// offset of the three phases (contained / A-suffix~B-prefix / B-suffix~A-prefix) is scored in
// parallel and reduced exactly as the reference's sequential scan would: the FIRST offset, in the
// reference's loop order, with the strictly greatest score among the accepted ones.
//   * score(offset) = number of positions with equal, non-'N' bases (the reference's quality
//     terms `(int)q > 5` are always true for printable qualities);
//   * the reference aborts an offset (score = -1) as soon as (k - score) > MM; mismatches only
//     accumulate, so that happens iff (len - score) >= MM + 2 -- no early exit needed here;
//   * accepted iff score/len >= minPercent in IEEE single precision (phase 3 of Overlap.cpp: '>').
// The greedy order itself (which read merges into which) stays on the host, where the reference
// keeps it.
#include "rfx_internal.h"

namespace {

struct Best {
  unsigned long long v;  // (score + 1) << 32 | ~ordered index  -> max = greatest score, earliest offset
};

__device__ __forceinline__ unsigned long long pack_best(int score, uint32_t idx) {
  return ((unsigned long long)(uint32_t)(score + 1) << 32) | (uint32_t)(~idx);
}

// One (A, B) pair, both strings already in LDS (sa, sb); the whole workgroup takes part.  out5 as documented in
// rufus_hip.h (rfx_overlap_score).
__device__ __forceinline__ void score_pair(const char* sa, int alen, const char* sb, int blen, float min_pct, int min_ovl,
                                           int strict3, int local_init, unsigned long long* s_p1,
                                           unsigned long long* s_full, int* __restrict__ out5) {
  const bool a_smaller = !(blen > alen);
  const int window = a_smaller ? blen : alen, longest = a_smaller ? alen : blen;
  // int MM = window - (window * minPercent);   float arithmetic, no contraction
  const int mm = (int)__fsub_rn((float)window, __fmul_rn((float)window, min_pct));
  const int n1 = longest - window + 1;
  const int n23 = window - 1 >= min_ovl ? window - min_ovl : 0;
  unsigned long long best1 = 0, bestf = 0;
  for (int t = threadIdx.x; t < n1 + 2 * n23; t += blockDim.x) {
    int phase, i, len, a0, b0;
    if (t < n1) {
      phase = 1; i = t; len = window;
      a0 = a_smaller ? i : 0;
      b0 = a_smaller ? 0 : i;
    } else if (t < n1 + n23) {
      phase = 2; i = window - 1 - (t - n1); len = i + 1;
      a0 = alen - i - 1; b0 = 0;
    } else {
      phase = 3; i = window - 1 - (t - n1 - n23); len = i + 1;
      a0 = 0; b0 = blen - i - 1;
    }
    if (a0 < 0 || b0 < 0) continue;
    int score = 0;
    for (int k = 0; k < len; ++k) {
      const char ca = sa[a0 + k], cb = sb[b0 + k];
      score += (ca == cb) & (ca != 'N');  // equal bases: testing either side for 'N' is the same
    }
    if (len - score >= mm + 2) continue;  // the reference's running abort
    const float pct = __fdiv_rn((float)score, (float)len);
    const bool ok = (phase == 3 && strict3) ? pct > min_pct : pct >= min_pct;
    if (!ok || score <= local_init) continue;
    const unsigned long long p = pack_best(score, (uint32_t)t);
    if (phase == 1 && p > best1) best1 = p;
    if (p > bestf) bestf = p;
  }
  if (best1) atomicMax(s_p1, best1);
  if (bestf) atomicMax(s_full, bestf);
  __syncthreads();
  if (threadIdx.x == 0) {
    auto decode = [&](unsigned long long v, int& score, int& ovl) {
      score = local_init;
      ovl = 0;
      if (!v) return;
      score = (int)(v >> 32) - 1;
      const int t = (int)(~(uint32_t)v);
      if (t < n1) ovl = a_smaller ? -t : t;
      else if (t < n1 + n23) ovl = (window - 1 - (t - n1)) - alen + 1;
      else ovl = blen - (window - 1 - (t - n1 - n23)) - 1;
    };
    int s1, o1, sf, of;
    decode(*s_p1, s1, o1);
    decode(*s_full, sf, of);
    const int perfect = *s_p1 && s1 == window;
    if (perfect) {  // the reference skips phases 2 and 3 once phase 1 found a perfect match
      sf = s1;
      of = o1;
    }
    out5[0] = s1;
    out5[1] = o1;
    out5[2] = perfect;
    out5[3] = sf;
    out5[4] = of;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_overlap_score(const char* __restrict__ a, int alen,
                                                        const char* __restrict__ bcat,
                                                        const uint32_t* __restrict__ boff, int nb, float min_pct,
                                                        int min_ovl, int strict3, int local_init,
                                                        int* __restrict__ out /* nb x 5 */) {
  extern __shared__ char s_str[];  // A then B
  __shared__ unsigned long long s_p1, s_full;
  for (int j = blockIdx.x; j < nb; j += gridDim.x) {
    const int blen = (int)(boff[j + 1] - boff[j]);
    const char* b = bcat + boff[j];
    char* sa = s_str;
    char* sb = s_str + alen;
    for (int i = threadIdx.x; i < alen; i += blockDim.x) sa[i] = a[i];
    for (int i = threadIdx.x; i < blen; i += blockDim.x) sb[i] = b[i];
    if (threadIdx.x == 0) {
      s_p1 = 0;
      s_full = 0;
    }
    __syncthreads();
    score_pair(sa, alen, sb, blen, min_pct, min_ovl, strict3, local_init, &s_p1, &s_full, out + 5 * j);
  }
}

// The same against a device-resident pool of sequences (rfx_ovl_pool): the query is pool entry `query` (or an
// explicit string), forward and / or reverse-complemented in LDS (ACGTN only -- the host checks; Util::RevComp drops
// other characters, src/Util.cpp:187-210), the candidates are pool entries.  Work item w = strand * nb + j.
__global__ __launch_bounds__(256) void k_overlap_pool(const char* __restrict__ arena, const unsigned long long* __restrict__ off,
                                                       const int* __restrict__ len, const char* __restrict__ a_explicit,
                                                       int a_explicit_len, int query, const int* __restrict__ cand, int nb,
                                                       int strand_lo, int strand_hi, float min_pct, int min_ovl,
                                                       int strict3, int local_init, int* __restrict__ out) {
  extern __shared__ char s_str[];
  __shared__ unsigned long long s_p1, s_full;
  const char* a = a_explicit ? a_explicit : arena + off[query];
  const int alen = a_explicit ? a_explicit_len : len[query];
  const int nstr = strand_hi - strand_lo + 1;
  for (int w = blockIdx.x; w < nb * nstr; w += gridDim.x) {
    const int strand = strand_lo + w / nb, j = w % nb;
    const int cj = cand[j];
    const int blen = len[cj];
    const char* b = arena + off[cj];
    char* sa = s_str;
    char* sb = s_str + alen;
    if (strand == 0) {
      for (int i = threadIdx.x; i < alen; i += blockDim.x) sa[i] = a[i];
    } else {
      for (int i = threadIdx.x; i < alen; i += blockDim.x) {
        const char ch = a[alen - 1 - i];
        sa[i] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
      }
    }
    for (int i = threadIdx.x; i < blen; i += blockDim.x) sb[i] = b[i];
    if (threadIdx.x == 0) {
      s_p1 = 0;
      s_full = 0;
    }
    __syncthreads();
    score_pair(sa, alen, sb, blen, min_pct, min_ovl, strict3, local_init, &s_p1, &s_full, out + 5 * (size_t)w);
  }
}

// K7 (src/AnnotateOverlap.cpp:88-134): coverage[j] = number of matching windows that cover base j.
// Same rolling scan as k_filter (good streak >= K, last window excluded), run once per contig.
__device__ __forceinline__ uint32_t set_hash(uint64_t key, int bits) {
  uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  uint32_t h = (lo ^ (hi * 0x9E3779B1u)) * 0x85EBCA6Bu;
  h ^= h >> 15;
  h *= 0xC2B2AE35u;
  return h >> (32 - bits);
}

__global__ __launch_bounds__(64) void k_annotate(rfx_reads_view rv, const uint64_t* __restrict__ slots, int bits,
                                                  int has_all_ones, int k, const uint64_t* __restrict__ base_off,
                                                  uint32_t* __restrict__ cov /* per base, zeroed */) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rv.n) return;
  const uint32_t wr = rv_off(rv, r), len = rv_len(rv, r);
  const uint64_t* cw = rv.codes + wr;
  const uint32_t* cm = rv.good + wr;
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  const uint32_t mask = (1u << bits) - 1;
  const uint32_t stop = len ? len - 1 : 0;  // `i < length - HashSize`: the last window is skipped (:102)
  uint64_t fwd = 0;
  int streak = 0;
  uint32_t* c = cov + base_off[r];
  for (uint32_t p = 0; p < stop; ++p) {
    const uint64_t w = cw[p >> 5] >> (2 * (p & 31));
    const bool good = (cm[p >> 5] >> (p & 31)) & 1u;
    fwd = ((fwd << 2) | (w & 3u)) & kmask;
    streak = good ? streak + 1 : 0;
    if (streak >= k) {
      bool hit;
      if (fwd == RFX_EMPTY) hit = has_all_ones != 0;
      else {
        uint32_t s = set_hash(fwd, bits);
        for (;;) {
          const uint64_t cur = slots[s];
          if (cur == fwd) { hit = true; break; }
          if (cur == RFX_EMPTY) { hit = false; break; }
          s = (s + 1) & mask;
        }
      }
      if (hit)
        for (int j = 0; j < k; ++j) c[p - k + 1 + j] += 1;
    }
  }
}

}  // namespace

namespace rfxk {

void overlap_score(rfx_ctx* c, const char* d_a, int alen, const char* d_bcat, const uint32_t* d_boff, int nb,
                   int max_blen, float min_pct, int min_ovl, int strict3, int local_init, int* d_out) {
  if (nb == 0) return;
  rfx_span sp(c, "k_overlap_score");
  const size_t lds = (size_t)alen + (size_t)max_blen + 16;
  // (per device and cheap: set on every launch rather than behind a process-wide flag)
  if (!rfxi::lds_opt_in(c, (const void*)k_overlap_score, 150 * 1024, 4, "k_overlap_score")) return;
  hipLaunchKernelGGL(k_overlap_score, dim3(nb < 4096 ? nb : 4096), dim3(256), lds, c->stream, d_a, alen, d_bcat, d_boff,
                     nb, min_pct, min_ovl, strict3, local_init, d_out);
}

void overlap_pool(rfx_ctx* c, const char* arena, const uint64_t* off, const int* len, const char* a_explicit,
                  int a_explicit_len, int query, const int* cand, int nb, int strand_lo, int strand_hi, size_t lds,
                  float min_pct, int min_ovl, int strict3, int local_init, int* d_out) {
  const int work = nb * (strand_hi - strand_lo + 1);
  if (work == 0) return;
  rfx_span sp(c, "k_overlap_score");
  if (!rfxi::lds_opt_in(c, (const void*)k_overlap_pool, 150 * 1024, 5, "k_overlap_pool")) return;
  hipLaunchKernelGGL(k_overlap_pool, dim3(work < 8192 ? work : 8192), dim3(256), lds, c->stream, arena,
                     (const unsigned long long*)off, len, a_explicit, a_explicit_len, query, cand, nb, strand_lo, strand_hi,
                     min_pct, min_ovl, strict3, local_init, d_out);
}

void annotate(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* slots, int bits, int has_all_ones, int k,
              const uint64_t* base_off, uint32_t* cov) {
  if (rv.n == 0) return;
  rfx_span sp(c, "k_annotate");
  hipLaunchKernelGGL(k_annotate, dim3((rv.n + 63) / 64), dim3(64), 0, c->stream, rv, slots, bits, has_all_ones, k,
                     base_off, cov);
}

}  // namespace rfxk
