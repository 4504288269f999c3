// K1 ("read 2-bit pack + validity mask"): strict 4-line FASTQ TEXT -> a packed read block, on the device.
//
// The reference parses text with one producer thread per file (jf/include/jellyfish/mer_overlap_sequence_parser.hpp:
// 179-206: header line, sequence line, '+' line, quality line) and packs bases as the hashing threads consume them
// (jf/include/jellyfish/mer_dna.hpp:46-63).  Until round 6 the drop-in `jellyfish count` did both on the host -- parse
// and AVX2 pack at ~1 us per read and thread -- and uploaded 68 bytes per read; on the GPU box's 16-CPU quota that was
// most of a count's wall time while the device sat idle.  Here the host only moves bytes: record-aligned pieces of the
// text are appended to a device arena (rfx_text_append: one H2D copy each, from pinned memory), and rfx_text_parse turns
// the arena into a read block:
//
//   k_txt_count   newlines per 4 KB tile (exact SWAR zero-byte count of text ^ '\n')          -> scan -> tile ranks
//   k_txt_lines   line_start[i + 1] = the byte after the i-th newline (a 256-thread scan per tile)
//   k_txt_reads   record r = lines 4r .. 4r+3: checks the shape ('@' line, '+' line, quality as long as the sequence),
//                 len[r], words of the read; block totals (bases, longest read, reads shorter than 32) by atomics
//                                                                                            -> scan -> word offsets
//   k_txt_pack    one thread per code word: 32 bases -> 64 bits of jellyfish codes + the ACGT mask (RFX_PACK_COUNT), or
//                 RUFUS codes + the good mask from the quality line (RFX_PACK_FILTER) -- rfx_pack_reads' tables
//
// Anything that is not strict 4-line FASTQ (blank lines between records, multi-line records, a truncated tail) makes
// rfx_text_parse report "not strict" and the caller parses that text on the host, as before (rfx_text_fetch hands it
// back): the device path never guesses.  Two waits per block (line count, then word count): ~50 us against the 20 ms a
// gigabyte of text takes over PCIe.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "rfx_internal.h"

using rfxi::dfree;
using rfxi::dmalloc;
using rfxi::queue_read;

namespace {

int hip_fail(hipError_t e, const char* what) {
  char msg[256];
  snprintf(msg, sizeof msg, "%s: %s", what, hipGetErrorString(e));
  rfxi::set_error(msg);
  return RFX_E_HIP;
}
hipError_t ctx_sync(rfx_ctx* c) { return rfxi::sync(c); }
#define HIPCHK(x)                                   \
  do {                                              \
    hipError_t e_ = (x);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #x);  \
  } while (0)

constexpr int TX_TILE = 4096, TX_BLOCK = 256;  // 16 bytes per thread

struct txt_stats {
  unsigned long long n_bases, n_words;
  unsigned int bad, max_len, lower_cgt;
  unsigned int short_cnt[32];
};

__device__ __forceinline__ uint32_t nl_bits(uint32_t x) {  // bit 7 of every byte that is '\n', exactly
  const uint32_t y = x ^ 0x0A0A0A0Au;
  const uint32_t t = (y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | y | 0x7F7F7F7Fu);
}

__global__ __launch_bounds__(TX_BLOCK) void k_txt_count(const uint4* __restrict__ text, uint64_t n_tiles,
                                                         uint64_t* __restrict__ tile_cnt) {
  __shared__ uint32_t s_sum;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const uint4 v = text[tile * (TX_TILE / 16) + threadIdx.x];
    uint32_t c = __popc(nl_bits(v.x)) + __popc(nl_bits(v.y)) + __popc(nl_bits(v.z)) + __popc(nl_bits(v.w));
#pragma unroll
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_sum, c);
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[tile] = s_sum;
    __syncthreads();
  }
}

__global__ __launch_bounds__(TX_BLOCK) void k_txt_lines(const uint4* __restrict__ text, uint64_t n_tiles,
                                                         const uint64_t* __restrict__ tile_off,
                                                         uint32_t* __restrict__ line_start) {
  __shared__ uint32_t s_wave[TX_BLOCK / 64];
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint4 v = text[tile * (TX_TILE / 16) + threadIdx.x];
    const uint32_t b[4] = {nl_bits(v.x), nl_bits(v.y), nl_bits(v.z), nl_bits(v.w)};
    const uint32_t c = __popc(b[0]) + __popc(b[1]) + __popc(b[2]) + __popc(b[3]);
    uint32_t incl = c;  // inclusive scan over the wave, then over the four waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d, 64);
      if ((int)(threadIdx.x & 63u) >= d) incl += up;
    }
    if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_wave[w];
    uint64_t rank = tile_off[tile] + before;
    const uint32_t pos0 = (uint32_t)(tile * TX_TILE) + threadIdx.x * 16u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t m = b[q];
      while (m) {
        const int bit = __ffs((int)m) - 1;  // 7, 15, 23, 31
        m &= m - 1u;
        line_start[++rank] = pos0 + (uint32_t)q * 4u + (uint32_t)(bit >> 3) + 1u;
      }
    }
    __syncthreads();
  }
}

// rfx_pack_reads' tables (rfx_host.cpp PackLut): jellyfish code of either case / RUFUS code of upper case only
__device__ __forceinline__ bool is_acgt_upper(uint32_t ch) { return ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'; }
__device__ __forceinline__ uint32_t base_code(uint32_t ch) {  // of an A/C/G/T in either case: A0 C1 G2 T3
  const uint32_t x = (ch >> 1) & 3u;
  return x ^ (x >> 1);
}

__global__ __launch_bounds__(256) void k_txt_reads(const uint8_t* __restrict__ text, const uint32_t* __restrict__ line_start,
                                                    uint64_t n_reads, uint32_t* __restrict__ len, uint64_t* __restrict__ words,
                                                    txt_stats* __restrict__ st) {
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t l0 = line_start[4 * r], l1 = line_start[4 * r + 1], l2 = line_start[4 * r + 2], l3 = line_start[4 * r + 3],
                   l4 = line_start[4 * r + 4];
    const uint32_t L = l2 - 1u - l1, Q = l4 - 1u - l3;
    // (an empty header or '+' line cannot be: a line that starts with the marker has a byte)
    const bool ok = l1 - l0 >= 2u && text[l0] == '@' && l3 - l2 >= 2u && text[l2] == '+' && L == Q;
    if (!ok) atomicOr(&st->bad, 1u);
    len[r] = L;
    words[r] = (L + 31u) / 32u;
    atomicAdd(&st->n_bases, (unsigned long long)L);  // (one per read: 4 M atomics per block of text, ~0.2 ms)
    atomicMax(&st->max_len, L);
    if (L < 32u) atomicAdd(&st->short_cnt[L], 1u);
  }
}

// One thread per (read, word): MAXW words per read are walked, a read has the first ceil(len / 32) of them.
template <bool FILTER>
__global__ __launch_bounds__(256) void k_txt_pack(const uint8_t* __restrict__ text, const uint32_t* __restrict__ line_start,
                                                   const uint64_t* __restrict__ word_off64, const uint32_t* __restrict__ len,
                                                   uint64_t n_reads, uint32_t maxw, int min_q, uint64_t* __restrict__ codes,
                                                   uint32_t* __restrict__ mask, uint32_t* __restrict__ word_off,
                                                   txt_stats* __restrict__ st) {
  const uint64_t total = n_reads * maxw;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / maxw;
    const uint32_t wi = (uint32_t)(i - r * maxw);
    const uint32_t L = len[r];
    const uint64_t w0 = word_off64[r];
    if (wi == 0) word_off[r] = (uint32_t)w0;
    if (r == n_reads - 1 && wi == 0) word_off[n_reads] = (uint32_t)word_off64[n_reads];
    if (wi * 32u >= L) continue;
    const uint32_t nb = min(32u, L - wi * 32u);
    const uint8_t* s = text + line_start[4 * r + 1] + wi * 32u;
    const uint8_t* q = FILTER ? text + line_start[4 * r + 3] + wi * 32u : nullptr;
    uint64_t c = 0;
    uint32_t m = 0;
    bool lower = false;
    for (uint32_t b = 0; b < nb; ++b) {
      const uint32_t ch = s[b];
      if (FILTER) {  // src/Util.cpp:51-84: upper-case ACGT, anything else encodes as A; src/RUFUS.Filter.cpp:205: the good mask
        if (is_acgt_upper(ch)) c |= (uint64_t)base_code(ch) << (2 * b);
        const int qv = (int)(signed char)q[b];  // (the reference's plain `char`: a byte above 127 is a negative quality)
        if (!(qv - 33 < min_q || ch == 'N')) m |= 1u << b;
        lower |= ch == 'c' || ch == 'g' || ch == 't';
      } else if (is_acgt_upper(ch & 0xDFu)) {  // jf mer_dna.hpp:46-63: either case
        c |= (uint64_t)base_code(ch) << (2 * b);
        m |= 1u << b;
      }
    }
    codes[w0 + wi] = c;
    mask[w0 + wi] = m;
    if (FILTER && lower) atomicOr(&st->lower_cgt, 1u);
  }
}

}  // namespace

struct rfx_text {
  rfx_ctx* ctx = nullptr;
  uint8_t* arena = nullptr;
  uint64_t cap = 0, used = 0;
  bool vmm = false;
  // The copies run on a stream of their own: text for the NEXT block crosses PCIe while this ctx's stream parses and
  // counts the block before it (a second rfx_text of the same ctx) -- on the ctx stream copies and kernels would take
  // turns.  rfx_text_parse makes the ctx stream wait for the last copy.
  hipStream_t copy = nullptr;
  std::mutex mu;               // append / copied / wait may come from several host threads (the ingest's workers)
  // One event per append: its bytes have left the host buffer.  The events live until close and are recorded again by
  // the appends after a reset (n_ev of them are this generation's): a waiter that still holds a handle from before the
  // reset -- whose copy is long done: the reset follows the parse -- must find an event, not freed memory.
  std::vector<hipEvent_t> ev;
  size_t n_ev = 0;
};

extern "C" {

rfx_text* rfx_text_open(rfx_ctx* c, uint64_t cap_bytes) {
  if (!c || cap_bytes == 0 || cap_bytes > (0xFFFFFFFFull - 2 * TX_TILE)) return nullptr;  // (32-bit line offsets)
  (void)hipSetDevice(c->device);
  rfx_text* t = new rfx_text();
  t->ctx = c;
  t->cap = cap_bytes;
  // (plain hipMalloc, not the ctx's arena of mapped virtual memory: RFX_TEXT_VMM=1 keeps the arena for the A/B)
  t->vmm = getenv("RFX_TEXT_VMM") != nullptr;
  if (t->vmm) {
    t->arena = (uint8_t*)dmalloc(c, (size_t)cap_bytes + 2 * TX_TILE);
    if (t->arena && ctx_sync(c) != hipSuccess) { dfree(c, t->arena); t->arena = nullptr; }
  } else if (hipMalloc((void**)&t->arena, (size_t)cap_bytes + 2 * TX_TILE) != hipSuccess) {
    (void)hipGetLastError();
    t->arena = nullptr;
  }
  if (!t->arena || hipStreamCreateWithFlags(&t->copy, hipStreamNonBlocking) != hipSuccess) {
    if (t->arena) { if (t->vmm) dfree(c, t->arena); else (void)hipFree(t->arena); }
    delete t;
    return nullptr;
  }
  return t;
}

void rfx_text_close(rfx_text* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->copy);
  for (hipEvent_t e : t->ev) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(t->copy);
  if (t->vmm) dfree(t->ctx, t->arena);
  else (void)hipFree(t->arena);
  delete t;
}

uint64_t rfx_text_room(const rfx_text* t) { return t ? t->cap - t->used : 0; }
uint64_t rfx_text_bytes(const rfx_text* t) { return t ? t->used : 0; }

long rfx_text_append(rfx_text* t, const void* host, uint64_t n) {
  if (!t || (!host && n)) return RFX_E_INVAL;
  rfx_ctx* c = t->ctx;
  (void)hipSetDevice(c->device);
  std::lock_guard<std::mutex> g(t->mu);
  if (n > t->cap - t->used) return RFX_E_RANGE;
  if (t->n_ev == t->ev.size()) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return RFX_E_HIP;
    t->ev.push_back(e);
  }
  hipError_t rc = n ? hipMemcpyAsync(t->arena + t->used, host, n, hipMemcpyHostToDevice, t->copy) : hipSuccess;
  if (rc == hipSuccess) rc = hipEventRecord(t->ev[t->n_ev], t->copy);
  if (rc != hipSuccess) return hip_fail(rc, "rfx_text_append");
  t->used += n;
  return (long)t->n_ev++;
}

static int text_event(rfx_text* t, long ticket, hipEvent_t* e) {
  std::lock_guard<std::mutex> g(t->mu);
  if (ticket < 0 || (size_t)ticket >= t->n_ev) return RFX_E_INVAL;
  *e = t->ev[(size_t)ticket];
  return RFX_OK;
}

int rfx_text_copied(rfx_text* t, long ticket) {
  hipEvent_t e;
  if (!t || text_event(t, ticket, &e) != RFX_OK) return RFX_E_INVAL;
  const hipError_t rc = hipEventQuery(e);
  if (rc == hipSuccess) return 1;
  if (rc == hipErrorNotReady) {
    (void)hipGetLastError();
    return 0;
  }
  return hip_fail(rc, "rfx_text_copied");
}

int rfx_text_wait(rfx_text* t, long ticket) {
  hipEvent_t e;
  if (!t || text_event(t, ticket, &e) != RFX_OK) return RFX_E_INVAL;
  (void)hipSetDevice(t->ctx->device);
  HIPCHK(hipEventSynchronize(e));
  return RFX_OK;
}

int rfx_text_fetch(rfx_text* t, void* host) {
  if (!t || !host) return RFX_E_INVAL;
  rfx_ctx* c = t->ctx;
  (void)hipSetDevice(c->device);
  HIPCHK(hipStreamSynchronize(t->copy));
  if (t->used) HIPCHK(hipMemcpyAsync(host, t->arena, t->used, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(ctx_sync(c));
  return RFX_OK;
}

void rfx_text_reset(rfx_text* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->copy);
  std::lock_guard<std::mutex> g(t->mu);
  t->n_ev = 0;
  t->used = 0;
}

rfx_reads* rfx_text_parse(rfx_text* t, int flags, int min_q, int* strict) {
  if (strict) *strict = 1;
  if (!t || !strict || (flags != RFX_PACK_COUNT && flags != RFX_PACK_FILTER)) return nullptr;
  rfx_ctx* c = t->ctx;
  (void)hipSetDevice(c->device);
  const uint64_t n = t->used;
  auto not_strict = [&]() -> rfx_reads* {
    *strict = 0;
    return nullptr;
  };
  if (n == 0) return not_strict();
  {  // the ctx stream waits for the copies (they ran on the arena's own stream)
    std::lock_guard<std::mutex> g(t->mu);
    if (t->n_ev && hipStreamWaitEvent(c->stream, t->ev[t->n_ev - 1], 0) != hipSuccess) return nullptr;
  }
  // whole tiles are read: what lies behind the text holds no newline
  if (hipMemsetAsync(t->arena + n, 0, 2 * TX_TILE, c->stream) != hipSuccess) return nullptr;
  const uint64_t n_tiles = (n + TX_TILE - 1) / TX_TILE;
  uint64_t* tile_off = (uint64_t*)dmalloc(c, (n_tiles + 1) * 8);
  txt_stats* st = (txt_stats*)dmalloc(c, sizeof(txt_stats));
  uint32_t* line_start = nullptr;
  uint64_t* words = nullptr;
  rfx_reads* r = nullptr;
  auto drop = [&] { dfree(c, tile_off); dfree(c, st); dfree(c, line_start); dfree(c, words); };
  auto fail = [&](const char* what, hipError_t e) -> rfx_reads* {
    if (e != hipSuccess) hip_fail(e, what);
    (void)ctx_sync(c);  // (a read-back queued before the failure points at a local of this function: deliver it now)
    drop();
    if (r) rfx_reads_free(r);
    return nullptr;
  };
  if (!tile_off || !st) return fail("rfx_text_parse", hipSuccess);
  hipError_t e = hipMemsetAsync(st, 0, sizeof(txt_stats), c->stream);
  if (e != hipSuccess) return fail("rfx_text_parse", e);
  const int grid = (int)std::min<uint64_t>(n_tiles, (uint64_t)c->n_cu * 8);
  {
    rfx_span sp(c, "k_txt_count");
    hipLaunchKernelGGL(k_txt_count, dim3(grid), dim3(TX_BLOCK), 0, c->stream, (const uint4*)t->arena, n_tiles, tile_off);
  }
  rfxk::scan_tail(c, tile_off, n_tiles);
  uint64_t n_lines = 0;
  unsigned char last = 0;
  e = queue_read(c, &n_lines, tile_off + n_tiles, 8);
  if (e == hipSuccess) e = queue_read(c, &last, t->arena + n - 1, 1);
  if (e == hipSuccess) e = ctx_sync(c);
  if (e != hipSuccess) return fail("rfx_text_parse", e);
  // every line ends with its newline (the caller closes a file's last line), four lines per record
  if (last != '\n' || n_lines == 0 || (n_lines & 3u) || n_lines / 4 > 0xFFFFFFFEull) {
    drop();
    return not_strict();
  }
  const uint64_t n_reads = n_lines / 4;
  line_start = (uint32_t*)dmalloc(c, (n_lines + 1) * 4);
  words = (uint64_t*)dmalloc(c, (n_reads + 1) * 8);
  r = new rfx_reads();
  memset(r, 0, sizeof *r);
  r->gen = rfx_next_reads_gen();
  r->ctx = c;
  r->n = (uint32_t)n_reads;
  r->len = (uint32_t*)dmalloc(c, n_reads * 4);
  r->word_off = (uint32_t*)dmalloc(c, (n_reads + 1) * 4);
  if (!line_start || !words || !r->len || !r->word_off) return fail("rfx_text_parse", hipSuccess);
  e = hipMemsetAsync(line_start, 0, 4, c->stream);
  if (e != hipSuccess) return fail("rfx_text_parse", e);
  {
    rfx_span sp(c, "k_txt_lines");
    hipLaunchKernelGGL(k_txt_lines, dim3(grid), dim3(TX_BLOCK), 0, c->stream, (const uint4*)t->arena, n_tiles, tile_off, line_start);
  }
  {
    rfx_span sp(c, "k_txt_reads");
    hipLaunchKernelGGL(k_txt_reads, dim3((unsigned)std::min<uint64_t>((n_reads + 255) / 256, (uint64_t)c->n_cu * 16)), dim3(256), 0,
                       c->stream, t->arena, line_start, n_reads, r->len, words, st);
  }
  rfxk::scan_tail(c, words, n_reads);
  txt_stats hs;
  uint64_t n_words = 0;
  e = queue_read(c, &hs, st, sizeof hs);
  if (e == hipSuccess) e = queue_read(c, &n_words, words + n_reads, 8);
  if (e == hipSuccess) e = ctx_sync(c);
  if (e != hipSuccess) return fail("rfx_text_parse", e);
  if (hs.bad || n_words > 0xFFFFFFFFull) {
    drop();
    rfx_reads_free(r);
    return not_strict();
  }
  r->n_words = n_words;
  r->n_bases = hs.n_bases;
  r->max_len = hs.max_len;
  memcpy(r->short_cnt, hs.short_cnt, sizeof r->short_cnt);
  r->codes = (uint64_t*)dmalloc(c, std::max<uint64_t>(n_words, 1) * 8);
  uint32_t* mask = (uint32_t*)dmalloc(c, std::max<uint64_t>(n_words, 1) * 4);
  if (flags == RFX_PACK_COUNT) r->acgt = mask;
  else r->good = mask;
  if (!r->codes || !mask) return fail("rfx_text_parse", hipSuccess);
  const uint32_t maxw = std::max<uint32_t>(1, (hs.max_len + 31) / 32);
  const uint64_t total = n_reads * maxw;
  const unsigned pgrid = (unsigned)std::min<uint64_t>((total + 255) / 256, (uint64_t)c->n_cu * 32);
  {
    rfx_span sp(c, "k_txt_pack");
    if (flags == RFX_PACK_FILTER)
      hipLaunchKernelGGL(k_txt_pack<true>, dim3(pgrid), dim3(256), 0, c->stream, t->arena, line_start, words, r->len, n_reads, maxw,
                         min_q, r->codes, mask, r->word_off, st);
    else
      hipLaunchKernelGGL(k_txt_pack<false>, dim3(pgrid), dim3(256), 0, c->stream, t->arena, line_start, words, r->len, n_reads, maxw,
                         min_q, r->codes, mask, r->word_off, st);
  }
  // (the arena may be appended to again once this returns: the pack has read it)
  e = ctx_sync(c);
  drop();
  if (e != hipSuccess) {
    hip_fail(e, "rfx_text_parse");
    rfx_reads_free(r);
    return nullptr;
  }
  return r;
}

}  // extern "C"
