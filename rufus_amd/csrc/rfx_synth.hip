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
                                                      uint32_t* __restrict__ len) {
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
      mg |= ((o.lowq ? h_good : j_good) && !o.is_n ? 1u : 0u) << bit;
      if (bit == 31u || j + 1 == L) {
        codes[w0 + (j >> 5)] = cw;
        acgt[w0 + (j >> 5)] = ma;
        if (good) good[w0 + (j >> 5)] = mg;
        cw = 0;
        ma = mg = 0;
      }
    }
    word_off[r] = (uint32_t)w0;
    len[r] = L;
    if (r == n_reads - 1) word_off[n_reads] = (uint32_t)(w0 + wpr);
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
  r->ctx = c;
  r->n = n_pairs * 2;
  r->n_words = (uint64_t)r->n * wpr;
  r->n_bases = (uint64_t)r->n * p->read_len;
  r->max_len = p->read_len;
  if (p->read_len < 32) r->short_cnt[p->read_len] = r->n;
  r->codes = (uint64_t*)rfxi::dmalloc(c, std::max<uint64_t>(r->n_words, 1) * 8);
  r->acgt = (uint32_t*)rfxi::dmalloc(c, std::max<uint64_t>(r->n_words, 1) * 4);
  if (want_good) r->good = (uint32_t*)rfxi::dmalloc(c, std::max<uint64_t>(r->n_words, 1) * 4);
  r->word_off = (uint32_t*)rfxi::dmalloc(c, ((size_t)r->n + 1) * 4);
  r->len = (uint32_t*)rfxi::dmalloc(c, std::max<size_t>(r->n, 1) * 4);
  if (!r->codes || !r->acgt || (want_good && !r->good) || !r->word_off || !r->len) {
    rfx_reads_free(r);
    return nullptr;
  }
  if (r->n == 0) {
    (void)hipMemsetAsync(r->word_off, 0, 4, c->stream);
    return r;
  }
  const uint32_t grid = std::min<uint32_t>((r->n + 255) / 256, (uint32_t)c->n_cu * 32);
  hipLaunchKernelGGL(k_synth_reads, dim3(grid), dim3(256), 0, c->stream, *p, first_pair, r->n, min_q, r->codes, r->acgt,
                     r->good, r->word_off, r->len);
  if (hipGetLastError() != hipSuccess) {
    rfxi::set_error("rfx_synth_reads: launch failed");
    rfx_reads_free(r);
    return nullptr;
  }
  return r;
}
