// Device generator of the synthetic trio workload: packed read blocks written straight
// into HBM, bit-identical to rfx_pack_reads() of the text rfx_synth_text() produces (rfx_synth.h holds the
// shared arithmetic).  One thread per read; benchmark / scale-test input, outside every timed region.
#include <algorithm>
#include <cstring>

#include "rfx_internal.h"
#include "rfx_synth.h"

namespace {

__global__ __launch_bounds__(256) void k_synth_reads(rfx_synth p, uint64_t first_pair, uint32_t n_reads, int min_q,
                                                      uint64_t* __restrict__ codes, uint32_t* __restrict__ acgt,
                                                      uint32_t* __restrict__ good, uint32_t* __restrict__ word_off,
                                                      uint32_t* __restrict__ len,
                                                      unsigned long long* __restrict__ nbits) {
  const uint32_t L = p.read_len, wpr = (L + 31) / 32;
  const uint64_t st = rfxs::snv_stride(p);
  const bool j_good = (int)'J' - 33 >= min_q, h_good = (int)'#' - 33 >= min_q;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    const int mate = (int)(r & 1u);
    const rfxs::pair_geom g = rfxs::pair_of(p, first_pair + (r >> 1));
    const uint64_t mk = rfxs::mate_key(g, mate);
    // the (at most two) SNVs whose strata the read touches
    uint64_t sp0 = ~0ull, sp1 = ~0ull;
    if (p.carrier && g.hap && p.n_snv) {
      const uint64_t lo = mate ? g.end - L : g.start;
      const uint64_t i0 = lo >= 1000 ? min((lo - 1000) / st, (uint64_t)p.n_snv - 1) : 0;
      sp0 = rfxs::snv_pos(p, i0);
      if (i0 + 1 < p.n_snv) sp1 = rfxs::snv_pos(p, i0 + 1);
    }
    uint64_t gw = 0, gw_j = ~0ull, rb = 0;
    uint64_t cw = 0;
    uint32_t ma = 0, mg = 0;
    bool any_n = false;
    const size_t w0 = (size_t)r * wpr;
    for (uint32_t j = 0; j < L; ++j) {
      const uint64_t x = rfxs::base_coord(p, g, mate, j);
      if ((x >> 5) != gw_j) {
        gw_j = x >> 5;
        gw = rfxs::genome_word(p, gw_j);
      }
      uint32_t b = (uint32_t)(gw >> (2 * (x & 31))) & 3u;
      if (x == sp0 || x == sp1) b = rfxs::snv_alt(p, (x - 1000) / st < p.n_snv ? (x - 1000) / st : p.n_snv - 1, b);
      if ((j & 1u) == 0) rb = rfxs::mix64(mk + (uint64_t)((j >> 1) + 1) * rfxs::STEP);
      const rfxs::base_out o = rfxs::finish_base(p, b, mate, (uint32_t)(j & 1u ? rb >> 32 : rb));
      const uint32_t bit = j & 31u;
      cw |= (uint64_t)o.code << (2 * bit);
      ma |= (o.is_n ? 0u : 1u) << bit;
      any_n |= o.is_n;
      mg |= ((o.lowq ? h_good : j_good) && !o.is_n ? 1u : 0u) << bit;
      if (bit == 31u || j + 1 == L) {
        codes[w0 + (j >> 5)] = cw;
        acgt[w0 + (j >> 5)] = ma;
        if (good) good[w0 + (j >> 5)] = mg;
        cw = 0;
        ma = mg = 0;
      }
    }
    if (nbits) {  // compact block: flag the reads whose mask has to be kept
      if (any_n) atomicOr((unsigned int*)nbits + (r >> 5), 1u << (r & 31u));  // (32-bit: see k_filter_p's mask bits, rfx_kernels.hip)
      continue;
    }
    word_off[r] = (uint32_t)w0;
    len[r] = L;
    if (r == n_reads - 1) word_off[n_reads] = (uint32_t)(w0 + wpr);
  }
}

// nrank[g] = flagged reads before group g (one workgroup, chunked scan); total -> *n_exc
__global__ __launch_bounds__(1024) void k_flag_rank(const unsigned long long* __restrict__ nbits, uint32_t n_groups,
                                                     uint32_t* __restrict__ nrank, unsigned long long* __restrict__ n_exc) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (uint32_t g0 = 0; g0 < n_groups; g0 += 1024) {
    const uint32_t g = g0 + threadIdx.x;
    const uint32_t v = g < n_groups ? (uint32_t)__popcll(nbits[g]) : 0u;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(inc, off);
      if ((int)(threadIdx.x & 63) >= off) inc += o;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t before = s_base;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_w[w];
    if (g < n_groups) nrank[g] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_base = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_exc = s_base;
}

// the masks of the flagged reads, in read order, out of the dense mask array of the generation pass
__global__ __launch_bounds__(256) void k_flag_gather(const uint32_t* __restrict__ dense, uint32_t n_reads, uint32_t wpr,
                                                      const unsigned long long* __restrict__ nbits,
                                                      const uint32_t* __restrict__ nrank, uint32_t* __restrict__ exc) {
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    const unsigned long long bits = nbits[r >> 6];
    if (!((bits >> (r & 63u)) & 1ull)) continue;
    const size_t x = nrank[r >> 6] + (uint32_t)__popcll(bits & ((1ull << (r & 63u)) - 1ull));
    for (uint32_t w = 0; w < wpr; ++w) exc[x * wpr + w] = dense[(size_t)r * wpr + w];
  }
}

}  // namespace

extern "C" rfx_reads* rfx_synth_reads(rfx_ctx* c, const rfx_synth* p, uint64_t first_pair, uint32_t n_pairs, int min_q,
                                      int want_good) {
  if (!c || rfx_synth_check(p) != RFX_OK) {
    rfxi::set_error("rfx_synth_reads: bad parameters");
    return nullptr;
  }
  const uint32_t wpr = (p->read_len + 31) / 32;
  if ((uint64_t)n_pairs * 2 * wpr >= (1ull << 32) || (uint64_t)n_pairs * 2 >= (1ull << 32)) {
    rfxi::set_error("rfx_synth_reads: a block holds fewer than 2^32 words");
    return nullptr;
  }
  (void)hipSetDevice(c->device);
  rfx_reads* r = new rfx_reads();
  memset(r, 0, sizeof *r);
  r->gen = rfx_next_reads_gen();
  r->ctx = c;
  r->n = n_pairs * 2;
  r->n_words = (uint64_t)r->n * wpr;
  r->n_bases = (uint64_t)r->n * p->read_len;
  r->max_len = p->read_len;
  if (p->read_len < 32) r->short_cnt[p->read_len] = r->n;
  const bool compact = (want_good & 2) != 0;  // RFX_SYNTH_COMPACT
  want_good &= 1;
  r->codes = (uint64_t*)rfxi::dmalloc(c, std::max<uint64_t>(r->n_words, 1) * 8);
  // (compact: the dense mask exists only while the block is generated)
  uint32_t* dense = (uint32_t*)rfxi::dmalloc(c, std::max<uint64_t>(r->n_words, 1) * 4);
  if (want_good) r->good = (uint32_t*)rfxi::dmalloc(c, std::max<uint64_t>(r->n_words, 1) * 4);
  const size_t n_groups = ((size_t)r->n + 63) / 64;
  unsigned long long* d_nexc = nullptr;
  if (compact) {
    r->ulen = p->read_len;
    r->uwpr = wpr;
    r->nbits = (uint64_t*)rfxi::dmalloc(c, std::max<size_t>(n_groups, 1) * 8);
    r->nrank = (uint32_t*)rfxi::dmalloc(c, std::max<size_t>(n_groups, 1) * 4);
    d_nexc = (unsigned long long*)rfxi::dmalloc(c, 8);
  } else {
    r->acgt = dense;
    r->word_off = (uint32_t*)rfxi::dmalloc(c, ((size_t)r->n + 1) * 4);
    r->len = (uint32_t*)rfxi::dmalloc(c, std::max<size_t>(r->n, 1) * 4);
  }
  auto fail = [&](const char* msg) {
    if (msg) rfxi::set_error(msg);
    if (compact) rfxi::dfree(c, dense);
    rfxi::dfree(c, d_nexc);
    rfx_reads_free(r);
    return (rfx_reads*)nullptr;
  };
  if (!r->codes || !dense || (want_good && !r->good) ||
      (compact ? (!r->nbits || !r->nrank || !d_nexc) : (!r->word_off || !r->len)))
    return fail(nullptr);
  if (r->n == 0) {
    if (!compact) (void)hipMemsetAsync(r->word_off, 0, 4, c->stream);
    if (compact) rfxi::dfree(c, dense);
    rfxi::dfree(c, d_nexc);
    return r;
  }
  if (compact && hipMemsetAsync(r->nbits, 0, n_groups * 8, c->stream) != hipSuccess) return fail("rfx_synth_reads: memset failed");
  const uint32_t grid = std::min<uint32_t>((r->n + 255) / 256, (uint32_t)c->n_cu * 32);
  hipLaunchKernelGGL(k_synth_reads, dim3(grid), dim3(256), 0, c->stream, *p, first_pair, r->n, min_q, r->codes, dense,
                     r->good, r->word_off, r->len, (unsigned long long*)r->nbits);
  if (hipGetLastError() != hipSuccess) return fail("rfx_synth_reads: launch failed");
  if (compact) {
    hipLaunchKernelGGL(k_flag_rank, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long*)r->nbits,
                       (uint32_t)n_groups, r->nrank, d_nexc);
    unsigned long long n_exc = 0;
    if (rfxi::queue_read(c, &n_exc, d_nexc, 8) != hipSuccess || rfxi::sync(c) != hipSuccess)
      return fail("rfx_synth_reads: the flagged-read count did not come back");
    r->n_exc = n_exc;
    r->acgt = (uint32_t*)rfxi::dmalloc(c, std::max<uint64_t>(n_exc * wpr, 1) * 4);
    if (!r->acgt) return fail(nullptr);
    hipLaunchKernelGGL(k_flag_gather, dim3(grid), dim3(256), 0, c->stream, (const uint32_t*)dense, r->n, wpr,
                       (const unsigned long long*)r->nbits, (const uint32_t*)r->nrank, r->acgt);
    if (hipGetLastError() != hipSuccess) return fail("rfx_synth_reads: launch failed");
    rfxi::dfree(c, dense);  // (stream-ordered)
    rfxi::dfree(c, d_nexc);
  }
  return r;
}
