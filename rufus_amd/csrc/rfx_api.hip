// Device half of the C-ABI (include/rufus_hip.h): context, memory, orchestration of the kernels in
// rfx_kernels.hip.  No CPU fallback: rfx_open() fails when no gfx950 device is visible and every
// other entry point needs a ctx.
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <queue>
#include <thread>
#include <vector>

#include <unistd.h>

#include "rfx_internal.h"

namespace {

thread_local char g_err[512] = "";

int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return RFX_E_HIP;
}
#define HIPCHK(x)                                   \
  do {                                              \
    hipError_t e_ = (x);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #x);  \
  } while (0)
#define HIPCHKP(x)                        \
  do {                                    \
    hipError_t e_ = (x);                  \
    if (e_ != hipSuccess) {               \
      hip_fail(e_, #x);                   \
      return nullptr;                     \
    }                                     \
  } while (0)

// ---- device memory ---------------------------------------------------------------------------------
// Default: a growable arena on HIP's virtual memory API (see rfx_ctx).  Fallback (no VMM support, or
// RFX_NO_ARENA=1): hipMalloc per block behind a small size-keyed cache.
void pool_release(rfx_ctx* c) {
  for (auto& kv : c->pool) {
    c->used -= kv.first;
    (void)hipFree(kv.second);
  }
  c->pool.clear();
}

constexpr size_t ARENA_CHUNK = 1ull << 30;  // physical memory is mapped 1 GB at a time

bool arena_init(rfx_ctx* c) {
  if (c->arena || c->arena_off) return c->arena != nullptr;
  if (getenv("RFX_NO_ARENA")) { c->arena_off = true; return false; }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c->device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) {
    c->arena_off = true;
    (void)hipGetLastError();
    return false;
  }
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 288ull << 30;
  size_t reserve = ((total_b + (8ull << 30)) + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK;
  void* base = nullptr;
  if (hipMemAddressReserve(&base, reserve, gran, nullptr, 0) != hipSuccess || !base) {
    c->arena_off = true;
    (void)hipGetLastError();
    return false;
  }
  c->arena = (char*)base;
  c->arena_reserved = reserve;
  c->arena_gran = gran;
  return true;
}

// Map more physical memory behind the high-water mark so that a free range of `need` bytes ends there.
bool arena_grow(rfx_ctx* c, size_t need) {
  size_t tail = 0;  // free bytes already sitting at the end of the mapped part
  if (!c->arena_free.empty()) {
    auto last = std::prev(c->arena_free.end());
    if (last->first + last->second == c->arena_mapped) tail = last->second;
  }
  size_t add = need > tail ? need - tail : 0;
  add = (add + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK;
  if (!add) return true;
  if (c->arena_mapped + add > c->arena_reserved) return false;
  if (c->budget && c->arena_mapped + add > c->budget + ARENA_CHUNK) return false;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c->device;
  // the owner, and the peers that pull survivors out of this arena (rfx_ctx_allow_peers)
  std::vector<hipMemAccessDesc> ads(1 + c->peer_devices.size());
  for (size_t i = 0; i < ads.size(); ++i) {
    ads[i] = hipMemAccessDesc{};
    ads[i].location.type = hipMemLocationTypeDevice;
    ads[i].location.id = i ? c->peer_devices[i - 1] : c->device;
    ads[i].flags = hipMemAccessFlagsProtReadWrite;
  }
  size_t done = 0;
  while (done < add) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, ARENA_CHUNK, &prop, 0) != hipSuccess) break;
    char* at = c->arena + c->arena_mapped + done;
    if (hipMemMap(at, ARENA_CHUNK, 0, h, 0) != hipSuccess) { (void)hipMemRelease(h); break; }
    if (hipMemSetAccess(at, ARENA_CHUNK, ads.data(), ads.size()) != hipSuccess &&
        hipMemSetAccess(at, ARENA_CHUNK, ads.data(), 1) != hipSuccess) {  // (peers refused: copies get staged instead)
      (void)hipMemUnmap(at, ARENA_CHUNK);
      (void)hipMemRelease(h);
      break;
    }
    c->arena_handles.push_back((void*)h);
    done += ARENA_CHUNK;
  }
  (void)hipGetLastError();
  if (done) {  // whatever was mapped joins the free list (merged with a free tail)
    const size_t off = c->arena_mapped;
    c->arena_mapped += done;
    auto last = c->arena_free.empty() ? c->arena_free.end() : std::prev(c->arena_free.end());
    if (last != c->arena_free.end() && last->first + last->second == off) last->second += done;
    else c->arena_free[off] = done;
  }
  return done >= add;
}

void arena_destroy(rfx_ctx* c) {
  if (!c->arena) return;
  for (size_t i = 0; i < c->arena_handles.size(); ++i) {
    (void)hipMemUnmap(c->arena + i * ARENA_CHUNK, ARENA_CHUNK);
    (void)hipMemRelease((hipMemGenericAllocationHandle_t)c->arena_handles[i]);
  }
  (void)hipMemAddressFree(c->arena, c->arena_reserved);
  c->arena = nullptr;
  c->arena_handles.clear();
  c->arena_free.clear();
  c->arena_mapped = 0;
}

void* arena_alloc(rfx_ctx* c, size_t bytes) {
  for (int attempt = 0; attempt < 2; ++attempt) {
    // best fit among the free ranges (ties: lowest address) -- keeps the big holes for the big blocks
    auto best = c->arena_free.end();
    for (auto it = c->arena_free.begin(); it != c->arena_free.end(); ++it)
      if (it->second >= bytes && (best == c->arena_free.end() || it->second < best->second)) best = it;
    if (best != c->arena_free.end()) {
      const size_t off = best->first, len = best->second;
      c->arena_free.erase(best);
      if (len > bytes) c->arena_free[off + bytes] = len - bytes;
      c->used += bytes;
      if (c->used > c->peak_used) c->peak_used = c->used;
      void* p = c->arena + off;
      c->allocs[p] = bytes;
      return p;
    }
    if (attempt || !arena_grow(c, bytes)) break;
  }
  snprintf(g_err, sizeof g_err, "out of device memory: %zu bytes wanted, %zu in use, %zu mapped", bytes, c->used,
           c->arena_mapped);
  return nullptr;
}

void arena_free_range(rfx_ctx* c, void* p, size_t bytes) {
  size_t off = (size_t)((char*)p - c->arena), len = bytes;
  auto next = c->arena_free.lower_bound(off);
  if (next != c->arena_free.begin()) {
    auto prev = std::prev(next);
    if (prev->first + prev->second == off) {
      off = prev->first;
      len += prev->second;
      c->arena_free.erase(prev);
    }
  }
  if (next != c->arena_free.end() && off + len == next->first) {
    len += next->second;
    c->arena_free.erase(next);
  }
  c->arena_free[off] = len;
  c->used -= bytes;
}

void* dmalloc(rfx_ctx* c, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  if (arena_init(c)) {
    if (c->budget && c->used + bytes > c->budget) {
      snprintf(g_err, sizeof g_err, "out of device memory: hbm budget exceeded: %zu + %zu > %zu", c->used, bytes, c->budget);
      return nullptr;
    }
    return arena_alloc(c, bytes);
  }
  auto it = c->pool.lower_bound(bytes);
  if (it != c->pool.end() && it->first <= bytes + bytes / 4) {
    void* p = it->second;
    c->allocs[p] = it->first;
    c->pool.erase(it);
    return p;
  }
  if (c->budget && c->used + bytes > c->budget) pool_release(c);
  if (c->budget && c->used + bytes > c->budget) {
    snprintf(g_err, sizeof g_err, "out of device memory: hbm budget exceeded: %zu + %zu > %zu", c->used, bytes, c->budget);
    return nullptr;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    pool_release(c);
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) {
    hip_fail(e, "hipMalloc");
    return nullptr;
  }
  c->used += bytes;
  c->allocs[p] = bytes;
  return p;
}

void dfree(rfx_ctx* c, void* p) {
  if (!p) return;
  auto it = c->allocs.find(p);
  if (it == c->allocs.end()) {
    (void)hipFree(p);
    return;
  }
  const size_t bytes = it->second;
  c->allocs.erase(it);
  // Everything on this ctx runs on one stream, so a freed block is safe to hand out again: work
  // that still uses it is ordered before any later kernel or copy on that stream.
  if (c->arena && (char*)p >= c->arena && (char*)p < c->arena + c->arena_reserved) arena_free_range(c, p, bytes);
  else c->pool.emplace(bytes, p);
}

// ---- pinned scratch -------------------------------------------------------------------------------
// Stream sync + delivery of the queued read-backs; every synchronisation of the API goes through here.
hipError_t ctx_sync(rfx_ctx* c) {
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess && c->launch_error != hipSuccess) e = c->launch_error;  // a launch that could not be set up
  if (e == hipSuccess)
    for (auto& r : c->pin_reads) memcpy(r.dst, c->pin + r.off, r.n);
  c->pin_reads.clear();
  c->pin_used = 0;
  return e;
}

// Device -> host copy that lands in `dst` at the next ctx_sync().
hipError_t queue_read(rfx_ctx* c, void* dst, const void* d_src, size_t n) {
  if (n == 0) return hipSuccess;
  const size_t need = (n + 63) & ~(size_t)63;
  if (c->pin && c->pin_used + need <= c->pin_cap) {
    const size_t off = c->pin_used;
    c->pin_used += need;
    c->pin_reads.push_back({dst, off, n});
    return hipMemcpyAsync(c->pin + off, d_src, n, hipMemcpyDeviceToHost, c->stream);
  }
  return hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, c->stream);
}

// queue_read() destinations are often locals of the calling function.  Every success path ends with a ctx_sync()
// that delivers and clears them; an error return in between would leave entries pointing into a dead stack frame,
// to be written by whichever ctx_sync() comes next.  A guard at the top of such a function drops, on the way out,
// whatever the function queued and did not deliver.  (Not for rfx_count_finish_begin: its read-backs go to heap
// and caller memory that outlives the call by design.)
struct pin_guard {
  rfx_ctx* c;
  size_t mark;
  explicit pin_guard(rfx_ctx* ctx) : c(ctx), mark(ctx->pin_reads.size()) {}
  ~pin_guard() {
    if (c->pin_reads.size() > mark) {
      (void)hipStreamSynchronize(c->stream);  // the copies into the pinned scratch may still be in flight
      c->pin_reads.resize(mark);
    }
  }
};

// Host -> device copy of a buffer the caller may free right away (staged in the pinned scratch when it fits).
hipError_t upload(rfx_ctx* c, void* d_dst, const void* src, size_t n) {
  if (n == 0) return hipSuccess;
  const size_t need = (n + 63) & ~(size_t)63;
  if (c->pin && c->pin_used + need <= c->pin_cap) {
    char* stage = c->pin + c->pin_used;
    c->pin_used += need;
    memcpy(stage, src, n);
    return hipMemcpyAsync(d_dst, stage, n, hipMemcpyHostToDevice, c->stream);
  }
  const hipError_t e = hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, c->stream);
  return e == hipSuccess ? ctx_sync(c) : e;  // pageable source: make it safe to free
}

int ceil_log2(uint64_t x) {
  int l = 0;
  while (l < 63 && (1ull << l) < x) ++l;
  return l;
}

void build_lut(const uint64_t* cols, int k, std::vector<uint64_t>& lut, int& ntab) {
  const int c = 2 * k;
  ntab = (c + 7) / 8;
  lut.assign((size_t)ntab * 256, 0);
  for (int t = 0; t < ntab; ++t)
    for (int v = 0; v < 256; ++v) {
      uint64_t r = 0;
      for (int j = 0; j < 8; ++j) {
        const int bit = 8 * t + j;
        if (((v >> j) & 1) && bit < c) r ^= cols[c - 1 - bit];
      }
      lut[(size_t)t * 256 + v] = r;
    }
}

rfx_table_view view_of(const rfx_table* t) {
  rfx_table_view v;
  v.keys = t->keys;
  v.counts = t->counts;
  v.cap = t->cap;
  v.slots = t->cap + RFX_TABLE_MARGIN;
  v.rshift = t->lsize >= t->tbits ? t->lsize - t->tbits : 0;
  v.lshift = t->lsize >= t->tbits ? 0 : t->tbits - t->lsize;
  v.pos_lo = t->pos_lo;
  v.pos_hi = t->pos_hi;
  v.ntab = t->ntab;
  v.kshift = 2 * t->k - v.lshift;
  return v;
}

int alloc_table_arrays(rfx_ctx* c, uint64_t cap, uint64_t** keys, uint32_t** counts) {
  const uint64_t slots = cap + RFX_TABLE_MARGIN;
  *keys = (uint64_t*)dmalloc(c, slots * 8);
  if (!*keys) return RFX_E_NOMEM;
  *counts = (uint32_t*)dmalloc(c, slots * 4);
  if (!*counts) {
    dfree(c, *keys);
    *keys = nullptr;
    return RFX_E_NOMEM;
  }
  HIPCHK(hipMemsetAsync(*keys, 0xFF, slots * 8, c->stream));
  HIPCHK(hipMemsetAsync(*counts, 0, slots * 4, c->stream));
  return RFX_OK;
}

int read_stats(rfx_table* t, rfx_table_stats* out) {
  HIPCHK(queue_read(t->ctx, out, t->d_stats, sizeof(*out)));
  HIPCHK(ctx_sync(t->ctx));
  return RFX_OK;
}

// Rehash into a table of new_cap slots.
int table_grow(rfx_table* t, uint64_t new_cap) {
  rfx_ctx* c = t->ctx;
  pin_guard guard(c);
  rfx_table_stats st;
  int rc = read_stats(t, &st);
  if (rc) return rc;
  uint64_t* pk = (uint64_t*)dmalloc(c, (st.distinct + 1) * 8);
  uint32_t* pc = (uint32_t*)dmalloc(c, (st.distinct + 1) * 4);
  unsigned long long* d_n = (unsigned long long*)dmalloc(c, 8);
  if (!pk || !pc || !d_n) {
    dfree(c, pk); dfree(c, pc); dfree(c, d_n);
    return RFX_E_FULL;
  }
  hipError_t e = hipMemsetAsync(d_n, 0, 8, c->stream);
  if (e == hipSuccess) {
    rfxk::table_pairs(c, view_of(t), pk, pc, d_n);
    e = ctx_sync(c);
  }
  if (e != hipSuccess) {
    dfree(c, pk); dfree(c, pc); dfree(c, d_n);
    return hip_fail(e, "table_grow");
  }
  dfree(c, t->keys);
  dfree(c, t->counts);
  t->keys = nullptr;
  t->counts = nullptr;
  rc = alloc_table_arrays(c, new_cap, &t->keys, &t->counts);
  if (rc) {
    dfree(c, pk); dfree(c, pc); dfree(c, d_n);
    return RFX_E_FULL;
  }
  t->cap = new_cap;
  t->tbits = ceil_log2(new_cap);
  e = hipMemsetAsync(t->d_stats, 0, sizeof(rfx_table_stats), c->stream);
  if (e == hipSuccess) {
    // the pairs already passed the pos range; widen it for the re-insert
    rfx_table_view v = view_of(t);
    rfxk::count_pairs(c, pk, pc, st.distinct, v, t->lut, t->d_stats);
    e = ctx_sync(c);
  }
  dfree(c, pk); dfree(c, pc); dfree(c, d_n);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "table_grow");
}

// Sortable-word transform of the P2L path.  M (r x c, r = lsize, c = 2k) has kernel dimension c - r;
// reduce M to row echelon form taking pivots from key bit 0 upward: every kernel vector then has its
// HIGHEST set bit at a free column f and is zero at all other free columns.  Two keys with equal pos
// differ by a kernel vector, so they first differ (from the top) at a free column: comparing the
// free-column bits high to low orders them like the full keys.  T = [M ; e_f for free f, descending]
// is invertible and w = T*key = (pos << (c-r)) | free bits sorts exactly like (pos,key)
// (jf/include/jellyfish/mer_heap.hpp:34-38).  Fills img_t[b] = T*e_b and img_inv[b] = T^-1*e_b;
// returns false when M is rank deficient (P2L then stays off).
bool build_sort_transform(const uint64_t* cols, int r, int c, uint64_t* img_t, uint64_t* img_inv) {
  if (c > 62 || r > c) return false;
  // rows of M as masks over key bits: pos bit i = parity(row[i] & key); key bit b selects cols[c-1-b]
  std::vector<uint64_t> row(r, 0);
  for (int b = 0; b < c; ++b)
    for (int i = 0; i < r; ++i)
      if ((cols[c - 1 - b] >> i) & 1) row[i] |= 1ull << b;
  std::vector<uint64_t> e(row);
  std::vector<bool> is_pivot(c, false);
  int rank = 0;
  for (int b = 0; b < c && rank < r; ++b) {
    int p = -1;
    for (int i = rank; i < r; ++i)
      if ((e[i] >> b) & 1) { p = i; break; }
    if (p < 0) continue;
    std::swap(e[rank], e[p]);
    for (int i = 0; i < r; ++i)
      if (i != rank && ((e[i] >> b) & 1)) e[i] ^= e[rank];
    is_pivot[b] = true;
    ++rank;
  }
  if (rank != r) return false;
  std::vector<int> freec;
  for (int b = c - 1; b >= 0; --b)
    if (!is_pivot[b]) freec.push_back(b);  // descending
  // T as c row masks over key bits, row 0 = most significant bit of w
  std::vector<uint64_t> trow(c);
  for (int i = 0; i < r; ++i) trow[i] = row[r - 1 - i];          // w bit c-1-i = pos bit r-1-i
  for (int j = 0; j < c - r; ++j) trow[r + j] = 1ull << freec[j];  // then the free key bits, high to low
  for (int b = 0; b < c; ++b) {
    uint64_t w = 0;
    for (int i = 0; i < c; ++i)
      if ((trow[i] >> b) & 1) w |= 1ull << (c - 1 - i);
    img_t[b] = w;
  }
  // invert: Gauss-Jordan on [A | I] where A[i] = row mask producing w bit (c-1-i)
  std::vector<uint64_t> a(trow), inv(c);
  for (int i = 0; i < c; ++i) inv[i] = 1ull << i;  // inv[i]: combination of original rows, bit j = row j
  for (int col = 0; col < c; ++col) {
    int p = -1;
    for (int i = col; i < c; ++i)
      if ((a[i] >> col) & 1) { p = i; break; }
    if (p < 0) return false;
    std::swap(a[col], a[p]);
    std::swap(inv[col], inv[p]);
    for (int i = 0; i < c; ++i)
      if (i != col && ((a[i] >> col) & 1)) { a[i] ^= a[col]; inv[i] ^= inv[col]; }
  }
  // now a = I: key bit `col` = XOR over rows j in inv[col] of (w bit c-1-j)
  for (int wb = 0; wb < c; ++wb) {       // image of unit word e_wb
    const int j = c - 1 - wb;             // the T row that produces w bit wb
    uint64_t key = 0;
    for (int col = 0; col < c; ++col)
      if ((inv[col] >> j) & 1) key |= 1ull << col;
    img_inv[wb] = key;
  }
  return true;
}

void lut_from_images(const uint64_t* img, int nbits, std::vector<uint64_t>& lut, int& ntab) {
  ntab = (nbits + 7) / 8;
  lut.assign((size_t)ntab * 256, 0);
  for (int t = 0; t < ntab; ++t)
    for (int v = 0; v < 256; ++v) {
      uint64_t r = 0;
      for (int j = 0; j < 8; ++j)
        if (((v >> j) & 1) && 8 * t + j < nbits) r ^= img[8 * t + j];
      lut[(size_t)t * 256 + v] = r;
    }
}

// Matrix + device lookup tables of (k, lsize), built on first use and kept for the life of the ctx.
const rfx_hash_consts* get_consts(rfx_ctx* c, int k, int lsize, const uint64_t* cols_in) {
  rfx_hash_consts hc;
  memset(hc.cols, 0, sizeof hc.cols);
  if (cols_in) memcpy(hc.cols, cols_in, sizeof(uint64_t) * 2 * k);
  else {
    std::vector<uint64_t>& sc = c->std_cols[k * 64 + lsize];  // generating it costs a Gaussian elimination
    if (sc.empty()) {
      sc.assign(2 * (size_t)k, 0);
      if (rfx_jf_matrix(lsize, k, sc.data()) != RFX_OK) { c->std_cols.erase(k * 64 + lsize); return nullptr; }
    }
    memcpy(hc.cols, sc.data(), sizeof(uint64_t) * 2 * k);
  }
  uint64_t digest = 0xcbf29ce484222325ull;
  for (int i = 0; i < 2 * k; ++i) digest = (digest ^ hc.cols[i]) * 0x100000001b3ull;
  auto key = std::make_pair(k * 64 + lsize, digest);
  auto it = c->consts.find(key);
  if (it != c->consts.end()) return &it->second;
  std::vector<uint64_t> lut, lt, li;
  build_lut(hc.cols, k, lut, hc.ntab);
  hc.lut = (uint64_t*)dmalloc(c, lut.size() * 8);
  if (!hc.lut || hipMemcpyAsync(hc.lut, lut.data(), lut.size() * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess)
    return nullptr;
  uint64_t img_t[64], img_inv[64];
  if (build_sort_transform(hc.cols, lsize, 2 * k, img_t, img_inv)) {
    int nt = 0;
    lut_from_images(img_t, 2 * k, lt, nt);
    lut_from_images(img_inv, 2 * k, li, nt);
    hc.lut_t = (uint64_t*)dmalloc(c, lt.size() * 8);
    hc.lut_tinv = (uint64_t*)dmalloc(c, li.size() * 8);
    if (!hc.lut_t || !hc.lut_tinv ||
        hipMemcpyAsync(hc.lut_t, lt.data(), lt.size() * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(hc.lut_tinv, li.data(), li.size() * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess)
      hc.lut_t = hc.lut_tinv = nullptr;
  }
  if (ctx_sync(c) != hipSuccess) return nullptr;  // host vectors die at return
  return &(c->consts[key] = hc);
}

rfx_records* records_alloc(rfx_ctx* c, int k, int lsize, const uint64_t* cols, uint64_t n) {
  rfx_records* r = new rfx_records();
  r->ctx = c;
  r->k = k;
  r->lsize = lsize;
  r->n = n;
  memcpy(r->cols, cols, sizeof(uint64_t) * 2 * k);
  const rfx_hash_consts* hc = get_consts(c, k, lsize, cols);
  r->keys = (uint64_t*)dmalloc(c, n * 8);
  r->counts = (uint32_t*)dmalloc(c, n * 4);
  r->pos = (uint64_t*)dmalloc(c, n * 8);
  if (!hc || !r->keys || !r->counts || !r->pos) {
    rfx_records_free(r);
    return nullptr;
  }
  r->lut = hc->lut;  // owned by the ctx
  r->ntab = hc->ntab;
  return r;
}

struct HostRun {
  std::vector<uint64_t> keys, pos;
  std::vector<uint32_t> counts;
};

// flags = count in [lo,hi] and key absent from every other file; compacted (order kept) to the host.
int unique_run(rfx_ctx* c, const rfx_records* f, const rfx_records* const* all, int n_all, uint32_t lo, uint32_t hi,
               HostRun& out) {
  pin_guard guard(c);
  out.keys.clear(); out.pos.clear(); out.counts.clear();
  if (f->n == 0) return RFX_OK;
  const uint64_t nblk = (f->n + 2047) / 2048;
  uint8_t* flags = (uint8_t*)dmalloc(c, f->n);
  uint64_t* ok = (uint64_t*)dmalloc(c, f->n * 8);
  uint64_t* op = (uint64_t*)dmalloc(c, f->n * 8);
  uint32_t* oc = (uint32_t*)dmalloc(c, f->n * 4);
  uint64_t* boff = (uint64_t*)dmalloc(c, nblk * 8);
  unsigned long long* d_tot = (unsigned long long*)dmalloc(c, 8);
  auto cleanup = [&] { dfree(c, flags); dfree(c, ok); dfree(c, op); dfree(c, oc); dfree(c, boff); dfree(c, d_tot); };
  if (!flags || !ok || !op || !oc || !boff || !d_tot) { cleanup(); return RFX_E_NOMEM; }
  rfxk::flag_range(c, f->counts, f->n, lo, hi, flags);
  for (int j = 0; j < n_all; ++j) {
    if (all[j] == f) continue;
    rfxk::flag_absent(c, f->keys, f->pos, f->n, all[j]->keys, all[j]->pos, all[j]->n, f->lsize, flags);
  }
  rfxk::compact(c, flags, f->keys, f->counts, f->pos, f->n, ok, oc, op, boff, d_tot);
  // The result is usually tiny (mutant k-mers): fetch the count AND the first entries with one
  // synchronisation; only a result beyond the speculative head needs a second round.
  unsigned long long tot = 0;
  const uint64_t head = std::min<uint64_t>(f->n, 2048);
  out.keys.resize(head); out.pos.resize(head); out.counts.resize(head);
  hipError_t e = queue_read(c, &tot, d_tot, 8);
  if (e == hipSuccess) e = queue_read(c, out.keys.data(), ok, head * 8);
  if (e == hipSuccess) e = queue_read(c, out.pos.data(), op, head * 8);
  if (e == hipSuccess) e = queue_read(c, out.counts.data(), oc, head * 4);
  if (e == hipSuccess) e = ctx_sync(c);
  if (e == hipSuccess) {
    out.keys.resize(tot); out.pos.resize(tot); out.counts.resize(tot);
    if (tot > head) {
      e = queue_read(c, out.keys.data() + head, ok + head, (tot - head) * 8);
      if (e == hipSuccess) e = queue_read(c, out.pos.data() + head, op + head, (tot - head) * 8);
      if (e == hipSuccess) e = queue_read(c, out.counts.data() + head, oc + head, (tot - head) * 4);
      if (e == hipSuccess) e = ctx_sync(c);
    }
  }
  cleanup();
  return e == hipSuccess ? RFX_OK : hip_fail(e, "unique_run");
}

void resolve_spans(rfx_ctx* c) {
  for (auto& s : c->spans) {
    hipEventSynchronize(s.e1);
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.e0, s.e1) == hipSuccess) {
      auto& a = c->acc[s.name];
      a.ms += ms;
      a.launches += 1;
    }
    c->free_events.push_back(s.e0);
    c->free_events.push_back(s.e1);
  }
  c->spans.clear();
}

}  // namespace

namespace rfxi {
void* dmalloc(rfx_ctx* c, size_t bytes) { return ::dmalloc(c, bytes); }
void dfree(rfx_ctx* c, void* p) { ::dfree(c, p); }
void set_error(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
hipError_t sync(rfx_ctx* c) { return ctx_sync(c); }
hipError_t queue_read(rfx_ctx* c, void* dst, const void* d_src, size_t n) { return ::queue_read(c, dst, d_src, n); }
bool lds_opt_in(rfx_ctx* c, const void* fn, size_t bytes, int bit, const char* name) {
  if (c->lds_opt_in & (1u << bit)) return true;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof g_err, "%s: %zu bytes of dynamic LDS refused: %s", name, bytes, hipGetErrorString(e));
    c->launch_error = e;
    return false;
  }
  c->lds_opt_in |= 1u << bit;
  return true;
}
}  // namespace rfxi

extern "C" {

const char* rfx_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
rfx_ctx* rfx_open(int device, size_t hbm_budget_bytes) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0 || device < 0 || device >= n) {
    snprintf(g_err, sizeof g_err, "no HIP device (count=%d, %s)", n, hipGetErrorString(e));
    return nullptr;
  }
  hipDeviceProp_t prop;
  HIPCHKP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(g_err, sizeof g_err, "device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
    return nullptr;
  }
  HIPCHKP(hipSetDevice(device));
  rfx_ctx* c = new rfx_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->budget = hbm_budget_bytes;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return nullptr;
  }
  if (hipHostMalloc((void**)&c->pin, 4u << 20, hipHostMallocDefault) == hipSuccess) c->pin_cap = 4u << 20;
  else c->pin = nullptr;
  return c;
}

int rfx_ctx_allow_peers(rfx_ctx* c, const int* devices, int n) {
  if (!c || n < 0 || (n && !devices)) return RFX_E_INVAL;
  if (c->arena_mapped || !c->allocs.empty()) {
    snprintf(g_err, sizeof g_err, "rfx_ctx_allow_peers: call it before the first allocation of the ctx");
    return RFX_E_INVAL;
  }
  (void)hipSetDevice(c->device);
  c->peer_devices.clear();
  for (int i = 0; i < n; ++i) {
    const int d = devices[i];
    int can = 0;
    if (d == c->device || std::find(c->peer_devices.begin(), c->peer_devices.end(), d) != c->peer_devices.end()) continue;
    if (hipDeviceCanAccessPeer(&can, d, c->device) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
    c->peer_devices.push_back(d);
    if (hipDeviceEnablePeerAccess(d, 0) != hipSuccess) (void)hipGetLastError();  // (hipMalloc'ed blocks; already on: fine)
  }
  return RFX_OK;
}

void rfx_close(rfx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  ctx_sync(c);
  resolve_spans(c);
  for (auto& kv : c->allocs)
    if (!c->arena || (char*)kv.first < c->arena || (char*)kv.first >= c->arena + c->arena_reserved) (void)hipFree(kv.first);
  pool_release(c);
  arena_destroy(c);
  for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
  if (c->pin) (void)hipHostFree(c->pin);
  for (uint8_t* p : c->load_pin)
    if (p) (void)hipHostFree(p);
  if (c->aux) (void)hipStreamDestroy(c->aux);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

int rfx_sync(rfx_ctx* c) {
  if (!c) return RFX_E_NODEVICE;
  HIPCHK(ctx_sync(c));
  return RFX_OK;
}

void* rfx_stream(rfx_ctx* c) { return c ? (void*)c->stream : nullptr; }

}  // extern "C"
namespace {
// Big staging buffers: anonymous memory on transparent huge pages, then registered with the runtime -- page-locking
// goes by the page, and 320 MB of 2 MB pages take 0.02 s where hipHostMalloc's 4 KB pages take 0.06 (and 0.03 to free);
// every executable of the drop-in chain pins ~1 GB before its first byte moves.  (scratch/ubench/pin_cost.cpp; uploads
// from such a buffer run at the same 57 GB/s.)  Where huge pages or the registration are refused: hipHostMalloc.
std::mutex g_host_mu;
struct host_block { size_t len; bool registered; };
std::map<void*, host_block> g_host_blocks;

// anonymous memory on (transparent) huge pages, 2 MB-aligned; no call into the HIP runtime
void* host_map_thp(size_t bytes, size_t* len_out) {
  const size_t H = (size_t)2 << 20;
  const size_t len = ((bytes ? bytes : 1) + H - 1) & ~(H - 1);
  void* m = mmap(nullptr, len + H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return nullptr;
  char* a = (char*)(((uintptr_t)m + H - 1) & ~((uintptr_t)H - 1));  // (the kernel backs only aligned 2 MB ranges with huge pages)
  if (a > (char*)m) (void)munmap(m, (size_t)(a - (char*)m));
  const size_t tail = (size_t)((char*)m + len + H - (a + len));
  if (tail) (void)munmap(a + len, tail);
  (void)madvise(a, len, MADV_HUGEPAGE);
  *len_out = len;
  return a;
}
}  // namespace
extern "C" {

void* rfx_host_alloc(size_t bytes) {
  if (bytes >= ((size_t)8 << 20) && !getenv("RFX_NO_THP_PIN")) {
    size_t len = 0;
    if (void* a = host_map_thp(bytes, &len)) {
      if (hipHostRegister(a, len, hipHostRegisterDefault) == hipSuccess) {
        std::lock_guard<std::mutex> g(g_host_mu);
        g_host_blocks[a] = host_block{len, true};
        return a;
      }
      (void)hipGetLastError();
      (void)munmap(a, len);
    }
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void* rfx_host_alloc_lazy(size_t bytes) {
  size_t len = 0;
  void* a = host_map_thp(bytes, &len);
  if (!a) return nullptr;
  std::lock_guard<std::mutex> g(g_host_mu);
  g_host_blocks[a] = host_block{len, false};
  return a;
}

int rfx_host_pin(void* p) {
  if (!p) return RFX_E_INVAL;
  size_t len = 0;
  {
    std::lock_guard<std::mutex> g(g_host_mu);
    auto it = g_host_blocks.find(p);
    if (it == g_host_blocks.end()) return RFX_OK;  // (rfx_host_alloc's own: page-locked since it was made)
    if (it->second.registered) return RFX_OK;
    len = it->second.len;
  }
  if (hipHostRegister(p, len, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();  // (uploads from it still work, staged by the runtime)
    return RFX_E_HIP;
  }
  std::lock_guard<std::mutex> g(g_host_mu);
  g_host_blocks[p].registered = true;
  return RFX_OK;
}

void rfx_host_free(void* p) {
  if (!p) return;
  host_block b{0, false};
  bool ours = false;
  {
    std::lock_guard<std::mutex> g(g_host_mu);
    auto it = g_host_blocks.find(p);
    if (it != g_host_blocks.end()) {
      b = it->second;
      ours = true;
      g_host_blocks.erase(it);
    }
  }
  if (ours) {
    if (b.registered) (void)hipHostUnregister(p);
    (void)munmap(p, b.len);
  } else {
    (void)hipHostFree(p);
  }
}

int rfx_mem_reserve(rfx_ctx* c, uint64_t bytes) {
  if (!c) return RFX_E_INVAL;
  (void)hipSetDevice(c->device);
  if (!arena_init(c)) return RFX_OK;  // (no arena: nothing to prepare)
  if (c->arena_mapped >= bytes) return RFX_OK;
  size_t tail = 0;
  if (!c->arena_free.empty()) {
    auto last = std::prev(c->arena_free.end());
    if (last->first + last->second == c->arena_mapped) tail = last->second;
  }
  // A request the device cannot meet is refused BEFORE anything is mapped: arena_grow maps chunk after chunk until
  // hipMemCreate fails and keeps what it got -- a "refused" reserve of 200 GiB on a smaller (or shared) device would have
  // pinned nearly all of its free memory into this ctx until rfx_close, starving another ctx, RCCL or a plain hipMalloc.
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return RFX_E_NOMEM;
  }
  const uint64_t grow = bytes - c->arena_mapped;
  if (grow > (uint64_t)((double)free_b * 0.9)) return RFX_E_NOMEM;
  // (arena_grow maps so that a free range of `need` bytes ends at the new high-water mark)
  return arena_grow(c, tail + (size_t)grow) ? RFX_OK : RFX_E_NOMEM;
}

int rfx_mem_stats(rfx_ctx* c, uint64_t* used, uint64_t* peak, uint64_t* mapped) {
  if (!c) return RFX_E_NODEVICE;
  if (used) *used = c->used;
  if (peak) *peak = c->peak_used;
  if (mapped) *mapped = c->arena_mapped;
  return RFX_OK;
}

int rfx_memcpy_dev(rfx_ctx* c, void* d_dst, const void* d_src, size_t bytes) {
  if (!c || (bytes && (!d_dst || !d_src))) return RFX_E_INVAL;
  (void)hipSetDevice(c->device);
  if (bytes) HIPCHK(rfxk::copy_bytes(c, d_dst, d_src, bytes));
  HIPCHK(ctx_sync(c));
  return RFX_OK;
}

int rfx_prof_enable(rfx_ctx* c, int on) {
  if (!c) return RFX_E_NODEVICE;
  c->prof = on != 0;
  return RFX_OK;
}
int rfx_prof_filter(rfx_ctx* c, const char* names) {
  if (!c) return RFX_E_NODEVICE;
  c->prof_filter = (names && *names) ? std::string(",") + names + "," : std::string();
  return RFX_OK;
}
int rfx_prof_reset(rfx_ctx* c) {
  if (!c) return RFX_E_NODEVICE;
  resolve_spans(c);
  c->acc.clear();
  return RFX_OK;
}
int rfx_prof_query(rfx_ctx* c, const char* kernel, double* total_ms, uint64_t* launches) {
  if (!c || !kernel) return RFX_E_INVAL;
  resolve_spans(c);
  auto it = c->acc.find(kernel);
  if (total_ms) *total_ms = it == c->acc.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == c->acc.end() ? 0 : it->second.launches;
  return RFX_OK;
}
int rfx_prof_names(rfx_ctx* c, char* buf, size_t cap) {
  if (!c || !buf || cap == 0) return RFX_E_INVAL;
  resolve_spans(c);
  std::string s;
  for (auto& kv : c->acc) {
    s += kv.first;
    s += '\n';
  }
  if (s.size() + 1 > cap) return RFX_E_RANGE;
  memcpy(buf, s.c_str(), s.size() + 1);
  return RFX_OK;
}

// ---------------------------------------------------------------------------------------------
rfx_reads* rfx_reads_upload(rfx_ctx* c, const uint64_t* codes, const uint32_t* acgt, const uint32_t* good,
                            const uint32_t* word_off, const uint32_t* len, uint32_t n_reads) {
  if (!c || !word_off || !len || (!codes && n_reads)) return nullptr;
  (void)hipSetDevice(c->device);
  rfx_reads* r = new rfx_reads();
  memset(r, 0, sizeof *r);
  r->gen = rfx_next_reads_gen();
  r->ctx = c;
  r->n = n_reads;
  r->n_words = word_off[n_reads];
  for (uint32_t i = 0; i < n_reads; ++i) {
    r->n_bases += len[i];
    r->max_len = std::max(r->max_len, len[i]);
    if (len[i] < 32) ++r->short_cnt[len[i]];
  }
  r->codes = (uint64_t*)dmalloc(c, r->n_words * 8);
  r->word_off = (uint32_t*)dmalloc(c, ((size_t)n_reads + 1) * 4);
  r->len = (uint32_t*)dmalloc(c, (size_t)n_reads * 4);
  if (acgt) r->acgt = (uint32_t*)dmalloc(c, r->n_words * 4);
  if (good) r->good = (uint32_t*)dmalloc(c, r->n_words * 4);
  bool ok = r->codes && r->word_off && r->len && (!acgt || r->acgt) && (!good || r->good);
  if (ok) {
    hipError_t e = hipMemcpyAsync(r->codes, codes, r->n_words * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r->word_off, word_off, ((size_t)n_reads + 1) * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n_reads) e = hipMemcpyAsync(r->len, len, (size_t)n_reads * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && acgt) e = hipMemcpyAsync(r->acgt, acgt, r->n_words * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && good) e = hipMemcpyAsync(r->good, good, r->n_words * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = ctx_sync(c);
    if (e != hipSuccess) {
      hip_fail(e, "rfx_reads_upload");
      ok = false;
    }
  }
  if (!ok) {
    rfx_reads_free(r);
    return nullptr;
  }
  return r;
}

static void msp_forget_pending(rfx_table* t);
static void p2l_drop_segments(rfx_table* t);
static void rfx_reads_release_pending(const rfx_reads* r);

void rfx_reads_free(rfx_reads* r) {
  if (!r) return;
  rfx_reads_release_pending(r);  // a count table may still need these reads to redo its partition
  if (r->ctx->aux) (void)hipStreamSynchronize(r->ctx->aux);  // (a map made ahead may still be hashing them)
  dfree(r->ctx, r->codes); dfree(r->ctx, r->acgt); dfree(r->ctx, r->good);
  dfree(r->ctx, r->word_off); dfree(r->ctx, r->len);
  dfree(r->ctx, r->nbits); dfree(r->ctx, r->nrank);
  delete r;
}
uint32_t rfx_reads_count(const rfx_reads* r) { return r ? r->n : 0; }
uint64_t rfx_reads_bases(const rfx_reads* r) { return r ? r->n_bases : 0; }
uint64_t rfx_reads_words(const rfx_reads* r) { return r ? r->n_words : 0; }
uint64_t rfx_reads_device_bytes(const rfx_reads* r) {
  if (!r) return 0;
  uint64_t b = r->n_words * 8 + (r->good ? r->n_words * 4 : 0);
  if (r->ulen) return b + (((uint64_t)r->n + 63) / 64) * 12 + r->n_exc * r->uwpr * 4;
  return b + (r->acgt ? r->n_words * 4 : 0) + ((uint64_t)r->n + 1) * 4 + (uint64_t)r->n * 4;
}

int rfx_reads_get(const rfx_reads* r, uint64_t* codes, uint32_t* acgt, uint32_t* good, uint32_t* word_off, uint32_t* len) {
  if (!r) return RFX_E_INVAL;
  rfx_ctx* c = r->ctx;
  (void)hipSetDevice(c->device);
  if ((acgt && !r->acgt && !r->ulen) || (good && !r->good)) return RFX_E_INVAL;
  if (r->ulen) {  // compact block: hand out the classic arrays
    if (codes && r->n_words) HIPCHK(hipMemcpyAsync(codes, r->codes, r->n_words * 8, hipMemcpyDeviceToHost, c->stream));
    if (good && r->n_words) HIPCHK(hipMemcpyAsync(good, r->good, r->n_words * 4, hipMemcpyDeviceToHost, c->stream));
    const size_t ng = ((size_t)r->n + 63) / 64;
    std::vector<uint64_t> bits(ng);
    std::vector<uint32_t> exc((size_t)r->n_exc * r->uwpr);
    if (acgt && ng) HIPCHK(hipMemcpyAsync(bits.data(), r->nbits, ng * 8, hipMemcpyDeviceToHost, c->stream));
    if (acgt && !exc.empty()) HIPCHK(hipMemcpyAsync(exc.data(), r->acgt, exc.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(ctx_sync(c));
    size_t x = 0;
    for (uint32_t i = 0; i < r->n; ++i) {
      if (word_off) word_off[i] = i * r->uwpr;
      if (len) len[i] = r->ulen;
      if (!acgt) continue;
      const bool flagged = (bits[i >> 6] >> (i & 63)) & 1;
      for (uint32_t w = 0; w < r->uwpr; ++w) {
        const uint32_t nb = std::min<uint32_t>(32, r->ulen - 32 * w);
        acgt[(size_t)i * r->uwpr + w] = flagged ? exc[x * r->uwpr + w] : (nb == 32 ? ~0u : (1u << nb) - 1);
      }
      x += flagged;
    }
    if (word_off) word_off[r->n] = r->n * r->uwpr;
    return RFX_OK;
  }
  if (codes && r->n_words) HIPCHK(hipMemcpyAsync(codes, r->codes, r->n_words * 8, hipMemcpyDeviceToHost, c->stream));
  if (acgt && r->n_words) HIPCHK(hipMemcpyAsync(acgt, r->acgt, r->n_words * 4, hipMemcpyDeviceToHost, c->stream));
  if (good && r->n_words) HIPCHK(hipMemcpyAsync(good, r->good, r->n_words * 4, hipMemcpyDeviceToHost, c->stream));
  if (word_off) HIPCHK(hipMemcpyAsync(word_off, r->word_off, ((size_t)r->n + 1) * 4, hipMemcpyDeviceToHost, c->stream));
  if (len && r->n) HIPCHK(hipMemcpyAsync(len, r->len, (size_t)r->n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(ctx_sync(c));
  return RFX_OK;
}

// ---------------------------------------------------------------------------------------------
rfx_table* rfx_count_begin(rfx_ctx* c, int k, int canonical, int lsize, uint64_t capacity_slots, uint64_t pos_lo,
                           uint64_t pos_hi) {
  if (!c) return nullptr;
  if (k < 1 || k > 32 || (k == 32 && !canonical) || lsize < 1 || lsize > 63 || lsize > 2 * k) {
    snprintf(g_err, sizeof g_err, "rfx_count_begin: unsupported k=%d canonical=%d lsize=%d", k, canonical, lsize);
    return nullptr;
  }
  (void)hipSetDevice(c->device);
  rfx_table* t = new rfx_table();
  memset(t, 0, sizeof *t);
  t->ctx = c;
  t->k = k;
  t->canonical = canonical != 0;
  t->lsize = lsize;
  uint64_t cap = capacity_slots ? capacity_slots : (1ull << 22);
  if (cap < (1ull << 16)) cap = 1ull << 16;
  t->tbits = ceil_log2(cap);
  t->cap = 1ull << t->tbits;
  t->pos_lo = pos_lo;
  t->pos_hi = pos_hi ? pos_hi : (1ull << lsize);
  const rfx_hash_consts* hc = get_consts(c, k, lsize, nullptr);
  if (!hc) {
    delete t;
    return nullptr;
  }
  memcpy(t->cols, hc->cols, sizeof t->cols);
  t->lut = hc->lut;  // the lookup tables are owned by the ctx
  t->lut_t = hc->lut_t;
  t->lut_tinv = hc->lut_tinv;
  t->ntab = hc->ntab;
  t->segs = new std::vector<rfx_segment>();
  t->early = new std::vector<rfx_segment>();
  t->pend = new std::vector<rfx_pending_add>();
  t->deferred = new std::vector<const rfx_reads*>();
  t->passes = -1;
  t->d_stats = (rfx_table_stats*)dmalloc(c, sizeof(rfx_table_stats));
  t->d_ctl = (rfx_count_ctl*)dmalloc(c, sizeof(rfx_count_ctl));
  const bool ok = t->d_stats && t->d_ctl &&
                  hipMemsetAsync(t->d_stats, 0, sizeof(rfx_table_stats), c->stream) == hipSuccess;
  if (!ok) {
    rfx_count_free(t);
    return nullptr;
  }
  return t;
}

void rfx_count_free(rfx_table* t) {
  if (!t) return;
  dfree(t->ctx, t->keys); dfree(t->ctx, t->counts); dfree(t->ctx, t->d_stats);
  dfree(t->ctx, t->d_ctl); dfree(t->ctx, t->ovf_keys);
  if (t->pend) {
    msp_forget_pending(t);
    delete t->pend;
    delete t->deferred;
  }
  if (t->segs) {
    for (auto& sg : *t->segs) {
      if (!sg.borrowed) dfree(t->ctx, sg.inst);
      if (!sg.borrowed) dfree(t->ctx, sg.ext);  // (the planes: until round 4 this leaked them for k = 26 .. 31)
      dfree(t->ctx, sg.bin_start);
    }
    delete t->segs;
  }
  if (t->runmaps && t->runmaps_owned) rfx_runmaps_free(t->runmaps);
  if (t->early) {  // early segments nobody adopted
    for (auto& sg : *t->early) {
      dfree(t->ctx, sg.inst);
      dfree(t->ctx, sg.ext);
      dfree(t->ctx, sg.bin_start);
    }
    delete t->early;
  }
  delete t;
}

static int ensure_table(rfx_table* t) {
  if (t->keys) return RFX_OK;
  return alloc_table_arrays(t->ctx, t->cap, &t->keys, &t->counts);
}

// ---- P2L path ----------------------------------------------------------------------------------
static rfx_ord_cfg ord_cfg(const rfx_table* t, int bin_bits) {
  rfx_ord_cfg c;
  c.c_bits = 2 * t->k;
  c.sel_bits = 2 * t->k - t->lsize;
  c.bin_shift = c.c_bits - bin_bits;
  return c;
}

static int p2l_add(rfx_table* t, const rfx_reads* r) {
  rfx_ctx* c = t->ctx;
  const uint64_t windows = r->windows_of(t->k);  // exact: the instance arrays below are sized by it
  if (windows >= (1ull << 32)) return RFX_E_RANGE;  // a segment indexes its instances with 32 bits
  if (!t->p2l_bins) {
    if (const char* ev = getenv("RFX_P2L_BINS")) t->p2l_bins = (uint32_t)atoi(ev);  // tuning experiments
  }
  if (!t->p2l_bins) {
    // ~16 K instances per bin: a few thousand distinct keys at sequencing depth, well inside the
    // 6144-key LDS table of k_leaf; denser bins are split into rounds there.
    uint32_t P = 256;
    while (P < 8192 && (uint64_t)P * 16384 < windows) P <<= 1;
    t->p2l_bins = P;
  }
  const uint32_t P = t->p2l_bins;
  const int bin_bits = ceil_log2(P);
  const rfx_ord_cfg cfg = ord_cfg(t, bin_bits);
  // transient + resident need of this path: 8 B per instance now, 12 B more per instance at finish
  uint64_t held = 0;
  for (auto& sg : *t->segs) held += sg.n;
  if (c->budget && c->used + (held + windows) * 20 > c->budget) return RFX_E_NOMEM;
  // the partition kernels fit two blocks per CU: a grid of exactly the resident blocks (see msp_geometry)
  const int G = std::min(rfxk::p2l_grid(c, r->n), c->n_cu * 2);
  uint32_t* cnt = (uint32_t*)dmalloc(c, (size_t)G * P * 4);
  uint64_t* bin_start = (uint64_t*)dmalloc(c, ((size_t)P + 1) * 8);
  uint32_t* gsum = (uint32_t*)dmalloc(c, (size_t)8 * P * 4);
  if (!cnt || !bin_start || !gsum) { dfree(c, cnt); dfree(c, bin_start); dfree(c, gsum); return RFX_E_NOMEM; }
  const rfx_reads_view rv = r->view();
  const uint32_t P1 = (uint32_t)rfxk::p1_bins();
  const bool two_level = P >= 2048 && !getenv("RFX_P2L_ONE_LEVEL");
  const uint32_t P2 = P / P1;
  uint32_t *cnt1 = nullptr, *gsum1 = nullptr, *fine_cur = nullptr;
  uint64_t* tot1 = nullptr;
  auto drop = [&] { dfree(c, cnt); dfree(c, gsum); dfree(c, cnt1); dfree(c, gsum1); dfree(c, fine_cur); dfree(c, tot1); };
  // Optimistic two-level path: the sizing pass is folded into the first partition pass (fixed-capacity
  // coarse bins, 16-bit per-block fine histogram).  If either assumption fails the device raises a flag
  // and the block is redone on the exact path below.  RFX_P2L_EXACT=1 forces the exact path.
  if (two_level && P <= 8192 && !getenv("RFX_P2L_EXACT")) {
    const uint64_t cap64 = windows / P1 + windows / (4ull * P1) + 65536;
    const size_t ncur = (size_t)P1 * rfxk::p1_cur_stride();
    uint32_t* coarse_cur = (uint32_t*)dmalloc(c, (ncur + 1) * 4);  // [ncur] = overflow flag
    fine_cur = (uint32_t*)dmalloc(c, (size_t)P * 4);
    uint64_t* buf_a = cap64 < (1ull << 32) / P1 ? (uint64_t*)dmalloc(c, cap64 * P1 * 8) : nullptr;
    uint64_t* inst = windows ? (uint64_t*)dmalloc(c, windows * 8) : nullptr;
    unsigned int flag = 1;
    if (coarse_cur && fine_cur && buf_a && inst &&
        hipMemsetAsync(coarse_cur, 0, (ncur + 1) * 4, c->stream) == hipSuccess &&
        hipMemsetAsync(fine_cur, 0, (size_t)P * 4, c->stream) == hipSuccess) {
      rfxk::part1_fused(c, rv, t->lut_t, t->ntab, t->k, t->canonical, cfg, P2, t->pos_lo, t->pos_hi, G, buf_a, coarse_cur,
                        (uint32_t)cap64, cnt, coarse_cur + ncur);
      rfxk::bin_totals(c, cnt, (uint32_t)G, P, bin_start);
      rfxk::part2(c, buf_a, inst, bin_start, fine_cur, P2, cfg.bin_shift, coarse_cur, (uint32_t)cap64, nullptr, nullptr,
                  ~0ull, "k_part2", nullptr, 0, windows);
      if (queue_read(c, &flag, coarse_cur + ncur, 4) != hipSuccess || ctx_sync(c) != hipSuccess) flag = 1;
    }
    dfree(c, coarse_cur); dfree(c, buf_a); dfree(c, fine_cur);
    fine_cur = nullptr;
    if (!flag) {
      drop();
      t->segs->push_back(rfx_segment{inst, windows, bin_start, windows, P});
      t->seg_kind = RFX_COUNT_P2L;
      return RFX_OK;
    }
    dfree(c, inst);
    if (hipGetLastError() != hipSuccess) { drop(); dfree(c, bin_start); return RFX_E_HIP; }
  }
  rfxk::bin_count(c, rv, t->lut_t, t->ntab, t->k, t->canonical, cfg, P, t->pos_lo, t->pos_hi, G, cnt);
  if (two_level) {
    cnt1 = (uint32_t*)dmalloc(c, (size_t)G * P1 * 4);
    gsum1 = (uint32_t*)dmalloc(c, (size_t)8 * P1 * 4);
    tot1 = (uint64_t*)dmalloc(c, ((size_t)P1 + 1) * 8);
    fine_cur = (uint32_t*)dmalloc(c, (size_t)P * 4);
    if (!cnt1 || !gsum1 || !tot1 || !fine_cur) { drop(); dfree(c, bin_start); return RFX_E_NOMEM; }
    rfxk::coarse_counts(c, cnt, (uint32_t)G, P, P2, cnt1);   // before bin_offsets rewrites cnt in place
  }
  rfxk::bin_offsets(c, cnt, (uint32_t)G, P, gsum, bin_start);
  if (two_level) rfxk::bin_offsets(c, cnt1, (uint32_t)G, P1, gsum1, tot1);
  // No read-back: the number of instances is bounded by the number of windows, which the host
  // knows; the exact per-bin extents stay on the device (bin_start) where the leaf reads them.
  const uint64_t total = windows;
  if (total == 0) { drop(); dfree(c, bin_start); return RFX_OK; }
  uint64_t* inst = (uint64_t*)dmalloc(c, total * 8);
  if (!inst) { drop(); dfree(c, bin_start); return RFX_E_NOMEM; }
  if (two_level) {
    uint64_t* buf_a = (uint64_t*)dmalloc(c, total * 8);
    if (!buf_a || hipMemsetAsync(fine_cur, 0, (size_t)P * 4, c->stream) != hipSuccess) {
      dfree(c, buf_a); dfree(c, inst); drop(); dfree(c, bin_start);
      return RFX_E_NOMEM;
    }
    rfxk::part1(c, rv, t->lut_t, t->ntab, t->k, t->canonical, cfg, P2, t->pos_lo, t->pos_hi, G, cnt1, bin_start, buf_a);
    rfxk::part2(c, buf_a, inst, bin_start, fine_cur, P2, cfg.bin_shift, nullptr, 0, nullptr, nullptr, ~0ull, "k_part2",
                nullptr, 0, total);
    dfree(c, buf_a);
  } else {
    rfxk::bin_scatter(c, rv, t->lut_t, t->ntab, t->k, t->canonical, cfg, P, t->pos_lo, t->pos_hi, G, cnt, bin_start,
                      inst);
  }
  drop();
  t->segs->push_back(rfx_segment{inst, total, bin_start, total, P});
  t->seg_kind = RFX_COUNT_P2L;
  return RFX_OK;
}

// Count every bin in LDS and emit the records with lower <= count <= upper in (pos,key) order.
static rfx_records* p2l_emit(rfx_table* t, uint64_t lower, uint64_t upper) {
  rfx_ctx* c = t->ctx;
  const uint32_t P = t->p2l_bins;
  const int bin_bits = ceil_log2(P);
  const rfx_ord_cfg cfg = ord_cfg(t, bin_bits);
  const int nseg = (int)t->segs->size();
  std::vector<const uint64_t*> h_inst(nseg), h_bs(nseg);
  uint64_t total_inst = 0;
  for (int i = 0; i < nseg; ++i) {
    h_inst[i] = (*t->segs)[i].inst;
    h_bs[i] = (*t->segs)[i].bin_start;
    total_inst += (*t->segs)[i].n;
  }
  const uint64_t** d_inst = (const uint64_t**)dmalloc(c, nseg * sizeof(void*));
  const uint64_t** d_bs = (const uint64_t**)dmalloc(c, nseg * sizeof(void*));
  uint64_t* tmp_start = (uint64_t*)dmalloc(c, ((size_t)P + 1) * 8);
  uint64_t* n_surv = (uint64_t*)dmalloc(c, ((size_t)P + 1) * 8);
  uint64_t* tmp_keys = (uint64_t*)dmalloc(c, total_inst * 8);
  uint32_t* tmp_counts = (uint32_t*)dmalloc(c, total_inst * 4);
  unsigned int* d_err = (unsigned int*)dmalloc(c, 4);
  auto cleanup = [&] {
    dfree(c, d_inst); dfree(c, d_bs); dfree(c, tmp_start); dfree(c, n_surv); dfree(c, tmp_keys); dfree(c, tmp_counts);
    dfree(c, d_err);
  };
  if (!d_inst || !d_bs || !tmp_start || !n_surv || !tmp_keys || !tmp_counts || !d_err) { cleanup(); return nullptr; }
  hipError_t e = upload(c, d_inst, h_inst.data(), nseg * sizeof(void*));
  if (e == hipSuccess) e = upload(c, d_bs, h_bs.data(), nseg * sizeof(void*));
  if (e == hipSuccess) e = hipMemsetAsync(d_err, 0, 4, c->stream);
  if (e != hipSuccess) { hip_fail(e, "p2l_emit"); cleanup(); return nullptr; }
  rfxk::tmp_start(c, d_bs, nseg, P, tmp_start);
  rfxk::leaf(c, d_inst, d_bs, nseg, h_inst[0], h_bs[0], P, cfg, lower, upper, tmp_start, tmp_keys, tmp_counts, n_surv,
             d_err);
  rfxk::scan_tail(c, n_surv, P);
  uint64_t total_out = 0;
  unsigned int err = 0;
  e = queue_read(c, &total_out, n_surv + P, 8);
  if (e == hipSuccess) e = queue_read(c, &err, d_err, 4);
  if (e == hipSuccess) e = ctx_sync(c);  // h_inst / h_bs stay alive until here
  if (e != hipSuccess || err) {
    if (e != hipSuccess) hip_fail(e, "p2l_emit");
    else snprintf(g_err, sizeof g_err, "P2L: a bin could not be split far enough to fit LDS");
    cleanup();
    return nullptr;
  }
  rfx_records* rec = records_alloc(c, t->k, t->lsize, t->cols, total_out);
  if (!rec) { cleanup(); return nullptr; }
  rfxk::leaf_compact(c, tmp_keys, tmp_counts, tmp_start, n_surv, P, t->lut_tinv, t->ntab, cfg.sel_bits, rec->keys,
                     rec->counts, rec->pos);
  cleanup();  // stream-ordered: the pool hands these blocks out again only to later work of this stream
  return rec;
}

// ---- MSP path (rfx_msp.hip) ---------------------------------------------------------------------
// Partition one read block into super-k-mer records grouped by minimizer bin.  The sizes of the
// intermediate buffers are estimates (records per k-mer depend on the sequence); the device raises a
// flag when one did not hold and the block is redone with exact sizes.  The flag is NOT waited for
// here: it is read with the next synchronisation the table needs anyway (finish), or when the read
// block is about to be freed -- the redo needs the reads.
struct msp_geom {
  uint32_t P, P1, P2, bin_lo, bin_hi;
  uint32_t c_lo, c_n;  // coarse bins the shard's records can fall into: c_lo .. c_lo + c_n - 1
  int bin_bits, G;
  size_t ncur;
  uint64_t windows;
  int rec_mode;  // rfxk::part2 / bin_hist mode of this table's records
};

// memory of a run map: the store's pool when it has one, else the arena
static void* runmaps_alloc(rfx_runmaps* s, size_t bytes) {
  if (!s->pool) return dmalloc(s->ctx, bytes);
  bytes = (bytes + 255) & ~(size_t)255;
  for (auto it = s->pool_free.begin(); it != s->pool_free.end(); ++it)
    if (it->second >= bytes) {
      const size_t off = it->first, len = it->second;
      s->pool_free.erase(it);
      if (len > bytes) s->pool_free[off + bytes] = len - bytes;
      return s->pool + off;
    }
  return nullptr;
}
static void runmaps_release(rfx_runmaps* s, void* p, size_t bytes) {
  if (!p) return;
  if (!s->pool) { dfree(s->ctx, p); return; }
  bytes = (bytes + 255) & ~(size_t)255;
  size_t off = (size_t)((char*)p - s->pool), len = bytes;
  auto next = s->pool_free.lower_bound(off);
  if (next != s->pool_free.begin()) {
    auto prev = std::prev(next);
    if (prev->first + prev->second == off) {
      off = prev->first;
      len += prev->second;
      s->pool_free.erase(prev);
    }
  }
  if (next != s->pool_free.end() && off + len == next->first) {
    len += next->second;
    s->pool_free.erase(next);
  }
  s->pool_free[off] = len;
}

// A run map on its way: the hashing launch is queued and the number of reads that went without a map is being read back.
struct runmap_pending {
  const rfx_reads* r = nullptr;
  void* map_dev = nullptr;
  uint32_t* map_ovf = nullptr;
  uint32_t ovf_cap = 0, n_ovf = 0;
  size_t map_bytes = 0, ovf_bytes = 0;
  bool queued = false;
};

// Maps made AHEAD (rfx_count_prefetch_maps): launches queued on the ctx's second stream, finished -- turned into entries of
// the store -- by whoever next looks at the store (runmaps_collect).
struct runmap_ahead {
  std::vector<runmap_pending> pend;
  hipEvent_t done = nullptr;
  int k = 0, canonical = 0;
};

// Queues the hashing launch (k_msp_part1 HMODE 4) over block r and the read-back of its overflow count into p.n_ovf
// (valid after the next synchronisation of the ctx; p must stay where it is until then).  false: no room, or a failure.
// read_back = false (a launch on the second stream): the caller fetches n_ovf itself.
static bool runmap_launch(rfx_table* t, const rfx_reads* r, runmap_pending& p, bool read_back = true) {
  rfx_ctx* c = t->ctx;
  rfx_runmaps* st = t->runmaps;
  p.r = r;
  p.ovf_cap = r->n / 32 + 4096;
  p.map_bytes = (size_t)r->n * 32;
  p.ovf_bytes = ((size_t)p.ovf_cap + 1) * 4;
  if (st->budget && st->bytes + st->pending_bytes + p.map_bytes + p.ovf_bytes > st->budget) return false;
  p.map_dev = runmaps_alloc(st, p.map_bytes);
  p.map_ovf = (uint32_t*)runmaps_alloc(st, p.ovf_bytes);
  if (!p.map_dev || !p.map_ovf || hipMemsetAsync(p.map_ovf, 0, 4, c->stream) != hipSuccess) return false;
  rfxk::msp_part1(c, r->view(), t->k, t->canonical, 15, 0, 0, 4, rfxk::msp_map_grid(c, r->n), nullptr, nullptr, 0, nullptr, nullptr, 0,
                  p.map_dev, p.map_ovf, p.ovf_cap);
  p.queued = !read_back || queue_read(c, &p.n_ovf, p.map_ovf, 4) == hipSuccess;
  if (p.queued) st->pending_bytes += p.map_bytes + p.ovf_bytes;
  return p.queued;
}

// After the synchronisation (ok: it succeeded): the map becomes an entry of the store -- or its memory goes back and, when
// the block's reads fall into more runs than a map holds, an entry WITHOUT a map remembers that (no later pass tries again).
static rfx_runmap_entry* runmap_finish_in(rfx_runmaps* st, int k, int canonical, runmap_pending& p, bool ok, int* rc);
static rfx_runmap_entry* runmap_finish(rfx_table* t, runmap_pending& p, bool ok, int* rc) {
  return runmap_finish_in(t->runmaps, t->k, t->canonical, p, ok, rc);
}
static rfx_runmap_entry* runmap_finish_in(rfx_runmaps* st, int k_, int canonical_, runmap_pending& p, bool ok, int* rc) {
  struct { int k, canonical; } tt{k_, canonical_};
  auto* t = &tt;
  const rfx_reads* r = p.r;
  if (p.queued) st->pending_bytes -= std::min<uint64_t>(st->pending_bytes, p.map_bytes + p.ovf_bytes);
  ok = ok && p.queued;
  if (ok && p.n_ovf <= p.ovf_cap) {
    rfx_runmap_entry en;
    en.map = p.map_dev;
    en.ovf = p.map_ovf;
    en.n_ovf = p.n_ovf;
    en.n_reads = r->n;
    en.codes = r->codes;
    en.gen = r->gen;
    en.k = t->k;
    en.canonical = t->canonical;
    en.map_bytes = p.map_bytes;
    en.ovf_bytes = p.ovf_bytes;
    en.bytes = p.map_bytes + p.ovf_bytes;
    st->bytes += en.bytes;
    return &(st->m[r] = en);
  }
  // no room (or more reads without a map than the list holds): the passes hash the block as before
  if (p.map_dev) runmaps_release(st, p.map_dev, p.map_bytes);
  if (p.map_ovf) runmaps_release(st, p.map_ovf, p.ovf_bytes);
  if (hipGetLastError() != hipSuccess) *rc = RFX_E_HIP;
  if (ok && p.n_ovf > p.ovf_cap) {
    rfx_runmap_entry none;
    none.n_reads = r->n;
    none.codes = r->codes;
    none.gen = r->gen;
    none.k = t->k;
    none.canonical = t->canonical;
    st->m[r] = none;
  }
  return nullptr;
}

// The maps made ahead become entries (or go back to the pool): waits for the second stream's launches.
static int runmaps_collect(rfx_runmaps* st) {
  if (!st || !st->ahead) return RFX_OK;
  runmap_ahead* a = st->ahead;
  st->ahead = nullptr;
  int rc = RFX_OK;
  bool ok = hipEventSynchronize(a->done) == hipSuccess;
  for (runmap_pending& p : a->pend)  // (a few dozen 4-byte copies: the launches are through)
    if (ok && hipMemcpy(&p.n_ovf, p.map_ovf, 4, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
  for (runmap_pending& p : a->pend) (void)runmap_finish_in(st, a->k, a->canonical, p, ok, &rc);
  (void)hipEventDestroy(a->done);
  delete a;
  if (!ok && rc == RFX_OK) rc = RFX_E_HIP;
  return rc;
}

// The run map of read block r in the table's store: the one that is there, or (make) a new one -- ONE hashing launch
// (k_msp_part1 HMODE 4) + a wait for the number of reads that went without a map.  nullptr: no store, not a block for maps
// (reads of more than 160 bases), no room, or a failure (*rc set then).
static rfx_runmap_entry* runmap_get(rfx_table* t, const rfx_reads* r, bool make, int* rc) {
  rfx_ctx* c = t->ctx;
  rfx_runmaps* st = t->runmaps;
  *rc = RFX_OK;
  if (!st || r->max_len > 160 || r->n == 0 || getenv("RFX_NO_RUNMAP")) return nullptr;
  if (st->ahead && (*rc = runmaps_collect(st)) != RFX_OK) return nullptr;
  auto it = st->m.find(r);
  if (it != st->m.end()) {
    rfx_runmap_entry& en = it->second;
    if (en.gen == r->gen && en.k == t->k && en.canonical == t->canonical && en.n_reads == r->n && en.codes == r->codes)
      return en.map ? &en : nullptr;  // (no map: the block was tried and is no block for one -- nobody tries again)
    // the entry of ANOTHER block that lived at this address (or of another k): gone
    runmaps_release(st, en.map, en.map_bytes);
    runmaps_release(st, en.ovf, en.ovf_bytes);
    st->bytes -= std::min<uint64_t>(st->bytes, en.bytes);
    st->m.erase(it);
  }
  if (!make) return nullptr;
  runmap_pending p;
  if (!runmap_launch(t, r, p)) return runmap_finish(t, p, false, rc);
  const bool ok = ctx_sync(c) == hipSuccess;
  return runmap_finish(t, p, ok, rc);
}

static bool msp_geometry(rfx_table* t, const rfx_reads* r, msp_geom& g) {
  rfx_ctx* c = t->ctx;
  g.windows = r->windows_of(t->k);
  if (!t->p2l_bins) {
    if (const char* ev = getenv("RFX_P2L_BINS")) t->p2l_bins = (uint32_t)atoi(ev);
    // up to 8192 bins come with the partition for free (16-bit LDS histogram fused into k_msp_part1); a big
    // block (>= ~4 M reads) pays a separate histogram pass and gets up to 32768 (256 sub-bins per coarse bin),
    // which saves a whole refinement level at WGS scale
    const uint32_t pcap = g.windows >= (1ull << 29) ? 32768 : 8192;
    uint32_t P = 256;
    while (P < pcap && (uint64_t)P * 16384 < g.windows) P <<= 1;
    if (!t->p2l_bins) t->p2l_bins = P;
    if (t->p2l_bins < 256) t->p2l_bins = 256;    // 128 coarse bins x >= 2 sub-bins
    if (t->p2l_bins > 32768) t->p2l_bins = 32768;
  }
  g.P = t->p2l_bins;
  g.P1 = (uint32_t)rfxk::p1_bins();
  g.P2 = g.P / g.P1;
  g.bin_bits = ceil_log2(g.P);
  // k_msp_part1 fits two 512-thread blocks per CU: a grid of exactly the resident blocks, each looping over
  // its share of the chunks, beats a larger one whose last wave of blocks runs on a half-empty chip
  {
    const uint32_t blk = (uint32_t)rfxk::msp_part1_block();
    const uint32_t chunks = (r->n + blk - 1) / blk;
    g.G = (int)std::max<uint32_t>(8, std::min<uint32_t>((uint32_t)c->n_cu * 2, (chunks + 7) & ~7u));
  }
  g.ncur = (size_t)g.P1 * rfxk::p1_cur_stride();
  // shard passes: the same cut as the multi-GPU owner ranges (on the top 8 bits of the bin index)
  const uint32_t ns = t->n_shards > 1 ? (uint32_t)t->n_shards : 1, sh = t->n_shards > 1 ? (uint32_t)t->shard : 0;
  g.bin_lo = ((sh * 256 + ns - 1) / ns) * (g.P / 256);
  g.bin_hi = (((sh + 1) * 256 + ns - 1) / ns) * (g.P / 256);
  g.c_lo = g.bin_lo / g.P2;
  g.c_n = g.bin_hi > g.bin_lo ? (g.bin_hi - 1) / g.P2 - g.c_lo + 1 : 1;
  g.rec_mode = t->canonical ? 1 : 2;
  return true;
}

static void msp_forget_pending(rfx_table* t) {
  rfx_ctx* c = t->ctx;
  for (auto& p : *t->pend) dfree(c, p.cur);
  t->pend->clear();
  c->pend_tables.erase(std::remove(c->pend_tables.begin(), c->pend_tables.end(), t), c->pend_tables.end());
}

// Exact two-pass sizing (32-bit histogram, then scatter into coarse bins as large as the fullest one).
static int msp_partition_exact(rfx_table* t, const rfx_reads* r, rfx_segment* seg) {
  rfx_ctx* c = t->ctx;
  pin_guard guard(c);
  msp_geom g;
  msp_geometry(t, r, g);
  const uint32_t P = g.P, P1 = g.P1, P2 = g.P2;
  const rfx_reads_view rv = r->view();
  uint32_t* cnt = (uint32_t*)dmalloc(c, (size_t)g.G * P * 4);
  uint32_t* gsum = (uint32_t*)dmalloc(c, (size_t)8 * P * 4);
  uint64_t* bin_start = (uint64_t*)dmalloc(c, ((size_t)P + 1) * 8);
  uint32_t* fine_cur = (uint32_t*)dmalloc(c, (size_t)P * 4);
  uint32_t* cur = (uint32_t*)dmalloc(c, (g.ncur + 1) * 4);
  uint64_t *buf_a = nullptr, *inst = nullptr;
  auto drop = [&] { dfree(c, cnt); dfree(c, gsum); dfree(c, fine_cur); dfree(c, cur); dfree(c, buf_a); };
  auto fail = [&](int rc) { drop(); dfree(c, bin_start); dfree(c, inst); return rc; };
  if (!cnt || !gsum || !bin_start || !fine_cur || !cur) return fail(RFX_E_NOMEM);
  rfxk::msp_part1(c, rv, t->k, t->canonical, g.bin_bits, g.bin_lo, g.bin_hi, 1, g.G, nullptr, nullptr, 0, cnt, nullptr);
  rfxk::bin_totals(c, cnt, (uint32_t)g.G, P, bin_start);
  std::vector<uint64_t> bs((size_t)P + 1);
  if (queue_read(c, bs.data(), bin_start, ((size_t)P + 1) * 8) != hipSuccess || ctx_sync(c) != hipSuccess)
    return fail(RFX_E_HIP);
  const uint64_t total = bs[P];
  uint64_t cap_a = 1;
  for (uint32_t cb = 0; cb < P1; ++cb) cap_a = std::max<uint64_t>(cap_a, bs[(size_t)(cb + 1) * P2] - bs[(size_t)cb * P2]);
  const uint64_t cap_b = total ? total : 1;
  if (cap_a >= (1ull << 32)) return fail(RFX_E_RANGE);
  // only the shard's coarse bins exist: the kernels index coarse bins absolutely, so they get the
  // address coarse bin 0 would have
  // (the coarse bins hold word and plane side by side, 12 bytes a slot: rfx_devutil.h msp_rec12)
  buf_a = (uint64_t*)dmalloc(c, cap_a * g.c_n * 12);
  inst = (uint64_t*)dmalloc(c, cap_b * 8);
  uint32_t* ext = (uint32_t*)dmalloc(c, cap_b * 4);
  if (!buf_a || !inst || !ext) { dfree(c, ext); return fail(RFX_E_NOMEM); }
  char* rec_a0 = (char*)buf_a - (size_t)g.c_lo * cap_a * 12;
  HIPCHK(hipMemsetAsync(cur, 0, (g.ncur + 1) * 4, c->stream));
  HIPCHK(hipMemsetAsync(fine_cur, 0, (size_t)P * 4, c->stream));
  rfxk::msp_part1(c, rv, t->k, t->canonical, g.bin_bits, g.bin_lo, g.bin_hi, 2, g.G, rec_a0, cur, (uint32_t)cap_a, nullptr, cur + g.ncur);
  rfxk::part2(c, (const uint64_t*)rec_a0, inst, bin_start, fine_cur, P2, 32 - g.bin_bits, cur, (uint32_t)cap_a, nullptr, ext, cap_b,
              "k_part2", nullptr, 0, cap_b, g.rec_mode, t->k);
  unsigned int flag = 1;
  if (queue_read(c, &flag, cur + g.ncur, 4) != hipSuccess || ctx_sync(c) != hipSuccess) { dfree(c, ext); return fail(RFX_E_HIP); }
  if (flag) {
    snprintf(g_err, sizeof g_err, "MSP: exact partition overflowed (internal error)");
    dfree(c, ext);
    return fail(RFX_E_HIP);
  }
  drop();
  *seg = rfx_segment{inst, total, bin_start, g.windows, P, ext};
  return RFX_OK;
}

// Super-k-mer records per window (k-mer instance) on ordinary sequence: 2 / (w + 1) for a window of w m-mers, plus
// one per read end -- 0.175 for w = 11, 0.125 for w = 16; with 20 % on top for the estimates (w = 11: 0.21, w = 16: 0.15).
static double msp_records_per_window(int k) { return 2.52 / (double)(rfxk::msp_window(k) + 1); }

static int msp_add_impl(rfx_table* t, const rfx_reads* r);

// Run maps take memory from a finish that was planned to fit without them.  A store the TABLE made for itself
// (rfx_count_set_passes: `jellyfish count`) is therefore given back when a block's partition runs out of memory, and the
// block is partitioned once more by hashing, as if there had never been a store; every later block and pass does the same.
// (A store handed in by the caller -- rfx_count_set_runmaps: rufus_amd/wgs.py -- is the caller's to drop: it sees
// RFX_E_NOMEM and repeats the step without maps.)
static bool msp_give_up_own_runmaps(rfx_table* t) {
  if (!t->runmaps || !t->runmaps_owned) return false;
  (void)ctx_sync(t->ctx);  // (launches that read the maps are still queued)
  rfx_runmaps_free(t->runmaps);
  t->runmaps = nullptr;
  t->runmaps_owned = 0;
  return true;
}

static int msp_add(rfx_table* t, const rfx_reads* r) {
  int rc = msp_add_impl(t, r);
  if (rc == RFX_E_NOMEM && msp_give_up_own_runmaps(t)) rc = msp_add_impl(t, r);
  return rc;
}

static int msp_add_impl(rfx_table* t, const rfx_reads* r) {
  rfx_ctx* c = t->ctx;
  msp_geom g;
  msp_geometry(t, r, g);
  if (g.windows >= (1ull << 32)) return RFX_E_RANGE;
  if (g.windows == 0) return RFX_OK;
  if (getenv("RFX_P2L_EXACT") && g.P <= 8192) {
    rfx_segment seg;
    const int rc = msp_partition_exact(t, r, &seg);
    if (rc) return rc;
    t->segs->push_back(seg);
    t->seg_kind = RFX_COUNT_MSP;
    return RFX_OK;
  }
  const uint32_t P = g.P, P2 = g.P2;
  const rfx_reads_view rv = r->view();
  // ~3 k-mers per record on ordinary sequence: room for 2.5 (of the shard's share of the bins), coarse
  // bins 25 % above even -- 6 % for big blocks, whose bins are even
  const bool big = g.windows >= (1ull << 29) || P > 8192;  // >= ~4 M reads: worth one synchronisation for exact sizes
  const bool wide = rfxk::msp_wide(t->k);                  // k = 26 .. 31: the records' 32-bit plane travels along
  const double share = (double)(g.bin_hi - g.bin_lo) / (double)P;
  uint64_t cap_b = (uint64_t)((double)g.windows * msp_records_per_window(t->k) * share * (share < 1 ? 1.05 : 1.0)) + 65536;
  if (cap_b > g.windows) cap_b = g.windows;
  const uint64_t even = cap_b / g.c_n;
  // k_msp_part1 hands out the coarse bins in slabs per workgroup (rfx_msp.hip): 1/32 of what a workgroup puts into a
  // bin, 16 .. 128 records; up to three of them per workgroup and bin stay (partly) unused
  int slab_log2 = 4;
  while (slab_log2 < 7 && (even / (uint64_t)g.G >> (slab_log2 + 6)) != 0) ++slab_log2;
  if (const char* ev = getenv("RFX_MSP_SLAB")) slab_log2 = std::min(7, std::max(2, atoi(ev)));
  uint64_t cap_a = even + (big ? even / 16 : even / 4) + 16384 + rfxk::msp_part1_slack(g.G, slab_log2);
  if (cap_a >= (1ull << 32)) return RFX_E_RANGE;
  uint64_t* bin_start = (uint64_t*)dmalloc(c, ((size_t)P + 1) * 8);
  // one zeroed block: coarse cursors, [ncur] = flag, then the fine-bin cursors of k_part2
  uint32_t* cur = (uint32_t*)dmalloc(c, (g.ncur + 1 + (size_t)P) * 4);
  uint32_t* fine_cur = cur ? cur + g.ncur + 1 : nullptr;
  if (!bin_start || !cur) { dfree(c, cur); dfree(c, bin_start); return RFX_E_NOMEM; }
  if (big) {
    // Big block: scatter into the coarse bins (no fused histogram: up to 32768 bins), histogram pass over the
    // records, then wait for the record count and the capacity flag and give the segment exactly the memory
    // it needs (at WGS scale the 30 % slack of an estimate is tens of GB) -- and nothing stays pending.
    // rfx_count_set_early: the records of the NEXT shard are cut in the same launch (hashing is the same work whether a
    // run is kept or dropped -- and with every bin kept the kernel does not even ask whose a run is) and partitioned
    // into a segment of their own; the shards' ranges are adjacent, so are their coarse bins.
    msp_geom g2 = g;
    bool early = false;
    if (t->early_on && t->n_shards > 1 && t->shard + 1 < t->n_shards && wide) {
      const uint32_t ns = (uint32_t)t->n_shards, sh1 = (uint32_t)t->shard + 1;
      g2.bin_lo = g.bin_hi;
      g2.bin_hi = (((sh1 + 1) * 256 + ns - 1) / ns) * (g.P / 256);
      // (only when the boundary between the two shards is a boundary of coarse bins: no coarse bin holds records of both)
      early = g2.bin_hi > g2.bin_lo && g.bin_hi > g.bin_lo && g.bin_hi % g.P2 == 0;
    }
    const uint32_t p1_hi = early ? g2.bin_hi : g.bin_hi;
    const uint32_t c_n_all = p1_hi > g.bin_lo ? (p1_hi - 1) / g.P2 - g.c_lo + 1 : 1;  // coarse bins of the launch
    const uint32_t c_mid = g.bin_hi / g.P2;                                             // first coarse bin of the next shard
    // Run maps (rfx_msp.hip): with S > 1 shard passes a big block is hashed ONCE, into its map (by the first pass that
    // adds it); every pass, that one included, cuts its records from reads + map.
    rfx_runmap_entry* have = nullptr;
    int G_rep = 0, G_ovf = 0;
    if (t->runmaps && t->n_shards > 1 && !early && g.bin_hi - g.bin_lo <= 16384u && !getenv("RFX_MSP_REC_HIST")) {
      int mrc = RFX_OK;
      have = runmap_get(t, r, true, &mrc);
      if (mrc) { dfree(c, cur); dfree(c, bin_start); return mrc; }
    }
    if (have) {  // (a chunk of 512 reads puts ~90 records into a coarse bin between two turns of the slabs)
      G_rep = rfxk::msp_replay_grid(c, r->n);
      G_ovf = have->n_ovf ? (int)std::min<uint32_t>(512, ((have->n_ovf + 511) / 512 + 7) & ~7u) : 0;
      slab_log2 = std::max(slab_log2, 7);
      cap_a = even + even / 16 + 16384 + rfxk::msp_part1_slack(G_rep + G_ovf, slab_log2);
      if (cap_a >= (1ull << 32)) { dfree(c, cur); dfree(c, bin_start); return RFX_E_RANGE; }
    }
    // (out of memory with a run-map store of the table's own: msp_add, the caller of this function, gives the store back
    // and runs it once more on the hashing path)
    if (have && t->runmaps_owned && getenv("RFX_TEST_NOMEM_WITH_MAPS")) {  // a test knob: that route, without a full device
      dfree(c, cur);
      dfree(c, bin_start);
      snprintf(g_err, sizeof g_err, "msp_add: out of device memory (injected: RFX_TEST_NOMEM_WITH_MAPS)");
      return RFX_E_NOMEM;
    }
    for (int attempt = 0;; ++attempt) {
      char* buf_a = (char*)dmalloc(c, cap_a * c_n_all * 12);  // 12-byte slots: word and plane side by side
      if (!buf_a) { dfree(c, cur); dfree(c, bin_start); return RFX_E_NOMEM; }
      char* buf_a0 = buf_a - (size_t)g.c_lo * cap_a * 12;  // the address coarse bin 0 would have
      auto fail = [&](int rc) { dfree(c, buf_a); dfree(c, cur); dfree(c, bin_start); return rc; };
      hipError_t e = hipMemsetAsync(cur, 0, (g.ncur + 1 + (size_t)P) * 4, c->stream);
      if (e == hipSuccess) e = hipMemsetAsync(bin_start, 0, ((size_t)P + 1) * 8, c->stream);
      if (e != hipSuccess) return fail(hip_fail(e, "msp_add"));
      // the exact fine histogram comes with the scatter (16-bit LDS counters, 64 KB per workgroup); only if one of
      // them wrapped are the records read once more for it
      const uint32_t rows = have ? (uint32_t)(G_rep + G_ovf) : (uint32_t)g.G;
      uint32_t* cnt = getenv("RFX_MSP_REC_HIST") ? nullptr : (uint32_t*)dmalloc(c, (size_t)rows * P * 4);
      if (have && !cnt) return fail(RFX_E_NOMEM);
      if (have) {
        rfxk::msp_replay(c, rv, have->map, t->k, t->canonical, g.bin_bits, g.bin_lo, g.bin_hi, G_rep, buf_a0, cur, (uint32_t)cap_a, cnt,
                         cur + g.ncur, slab_log2);
        if (have->n_ovf) {  // the reads whose runs did not fit a map: hashed as ever
          rfx_reads_view rvo = rv;
          rvo.idx = have->ovf + 1;
          rvo.n = have->n_ovf;
          rfxk::msp_part1(c, rvo, t->k, t->canonical, g.bin_bits, g.bin_lo, g.bin_hi, 3, G_ovf, buf_a0, cur, (uint32_t)cap_a,
                          cnt + (size_t)G_rep * P, cur + g.ncur, slab_log2);
        }
        rfxk::bin_totals(c, cnt, rows, P, bin_start);
      } else if (cnt) {
        rfxk::msp_part1(c, rv, t->k, t->canonical, g.bin_bits, g.bin_lo, p1_hi, 3, g.G, buf_a0, cur, (uint32_t)cap_a, cnt,
                        cur + g.ncur, slab_log2);
        rfxk::bin_totals(c, cnt, (uint32_t)g.G, P, bin_start);
      } else {
        rfxk::msp_part1(c, rv, t->k, t->canonical, g.bin_bits, g.bin_lo, p1_hi, 2, g.G, buf_a0, cur, (uint32_t)cap_a, nullptr,
                        cur + g.ncur);
        rfxk::surv_hist(c, (const uint64_t*)buf_a0, cur, (uint32_t)cap_a, P2, 32 - g.bin_bits, bin_start, g.rec_mode, t->k, cap_b);
        rfxk::scan_tail(c, bin_start, P);
      }
      uint64_t total = 0, n_own = 0;  // n_own: records of this table's shard (= total without the early cut)
      std::vector<uint32_t> h_cur(g.ncur + 1, 0);
      e = queue_read(c, &total, bin_start + P, 8);
      if (e == hipSuccess && early) e = queue_read(c, &n_own, bin_start + g.bin_hi, 8);
      if (e == hipSuccess) e = queue_read(c, h_cur.data(), cur, (g.ncur + 1) * 4);
      if (e == hipSuccess) e = ctx_sync(c);
      dfree(c, cnt);
      if (e != hipSuccess) return fail(hip_fail(e, "msp_add"));
      if (h_cur[g.ncur]) {  // a coarse bin overflowed: the cursors kept counting, they say what it needs
        uint64_t need = 0;
        for (uint32_t cb = 0; cb < g.P1; ++cb) need = std::max<uint64_t>(need, h_cur[(size_t)cb * rfxk::p1_cur_stride()]);
        if (need <= cap_a && cnt) {  // ... or only a 16-bit counter of the fused histogram wrapped: count the records
          e = hipMemsetAsync(bin_start, 0, ((size_t)P + 1) * 8, c->stream);
          if (e != hipSuccess) return fail(hip_fail(e, "msp_add"));
          rfxk::surv_hist(c, (const uint64_t*)buf_a0, cur, (uint32_t)cap_a, P2, 32 - g.bin_bits, bin_start, g.rec_mode, t->k, cap_b);
          rfxk::scan_tail(c, bin_start, P);
          e = queue_read(c, &total, bin_start + P, 8);
          if (e == hipSuccess && early) e = queue_read(c, &n_own, bin_start + g.bin_hi, 8);
          if (e == hipSuccess) e = ctx_sync(c);
          if (e != hipSuccess) return fail(hip_fail(e, "msp_add"));
        } else {
          dfree(c, buf_a);
          if (attempt >= 3 || need >= (1ull << 32) - 65536) { dfree(c, cur); dfree(c, bin_start); return RFX_E_RANGE; }
          cap_a = need + need / 64 + 1024;
          continue;
        }
      }
      if (early) {
        // two segments out of one set of coarse bins: this shard's bins lie below bin_hi and hold the first n_own
        // records of the bin order, the next shard's the rest -- its bin extents are rebased to an array of its own
        const uint64_t n_next = total - n_own;
        uint64_t* inst = (uint64_t*)dmalloc(c, (n_own ? n_own : 1) * 8);
        uint32_t* ext = (uint32_t*)dmalloc(c, (n_own ? n_own : 1) * 4);
        uint64_t* inst2 = (uint64_t*)dmalloc(c, (n_next ? n_next : 1) * 8);
        uint32_t* ext2 = (uint32_t*)dmalloc(c, (n_next ? n_next : 1) * 4);
        uint64_t* bs2 = (uint64_t*)dmalloc(c, ((size_t)P + 1) * 8);
        uint32_t* cur2 = (uint32_t*)dmalloc(c, (g.ncur + 1) * 4);
        auto drop2 = [&] { dfree(c, inst); dfree(c, ext); dfree(c, inst2); dfree(c, ext2); dfree(c, bs2); dfree(c, cur2); };
        if (!inst || !ext || !inst2 || !ext2 || !bs2 || !cur2) { drop2(); return fail(RFX_E_NOMEM); }
        const size_t split = (size_t)c_mid * rfxk::p1_cur_stride();  // coarse cursors below / from the boundary
        e = hipMemcpyAsync(cur2, cur, (g.ncur + 1) * 4, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess && split) e = hipMemsetAsync(cur2, 0, split * 4, c->stream);
        if (e == hipSuccess && g.ncur > split) e = hipMemsetAsync(cur + split, 0, (g.ncur - split) * 4, c->stream);
        if (e != hipSuccess) { drop2(); return fail(hip_fail(e, "msp_add")); }
        rfxk::split_bins(c, bin_start, P, n_own, bs2);
        rfxk::part2(c, (const uint64_t*)buf_a0, inst, bin_start, fine_cur, P2, 32 - g.bin_bits, cur, (uint32_t)cap_a, nullptr, ext,
                    n_own, "k_part2", nullptr, 0, n_own, g.rec_mode, t->k);
        rfxk::part2(c, (const uint64_t*)buf_a0, inst2, bs2, fine_cur, P2, 32 - g.bin_bits, cur2, (uint32_t)cap_a, nullptr, ext2,
                    n_next, "k_part2", nullptr, 0, n_next, g.rec_mode, t->k);
        dfree(c, buf_a);
        dfree(c, cur);
        dfree(c, cur2);
        const double share2 = (double)(g2.bin_hi - g2.bin_lo) / (double)P;
        const uint64_t nmax = (uint64_t)rfxk::msp_nmax_of(t->k);
        t->segs->push_back(rfx_segment{inst, n_own, bin_start, std::min<uint64_t>((uint64_t)((double)g.windows * share) + 1, n_own * nmax), P, ext});
        t->early->push_back(rfx_segment{inst2, n_next, bs2, std::min<uint64_t>((uint64_t)((double)g.windows * share2) + 1, n_next * nmax), P, ext2});
        t->seg_kind = RFX_COUNT_MSP;
        return RFX_OK;
      }
      if (have) ++t->replayed;
      uint64_t* inst = (uint64_t*)dmalloc(c, (total ? total : 1) * 8);
      uint32_t* ext = wide ? (uint32_t*)dmalloc(c, (total ? total : 1) * 4) : nullptr;
      if (!inst || (wide && !ext)) { dfree(c, inst); dfree(c, ext); return fail(RFX_E_NOMEM); }
      rfxk::part2(c, (const uint64_t*)buf_a0, inst, bin_start, fine_cur, P2, 32 - g.bin_bits, cur, (uint32_t)cap_a, nullptr, ext, total,
                  "k_part2", nullptr, 0, total, g.rec_mode, t->k);
      dfree(c, buf_a);
      dfree(c, cur);
      // (k-mer instances behind the records: the shard's share of the windows -- sizes the survivor arrays)
      t->segs->push_back(rfx_segment{inst, total, bin_start, std::min<uint64_t>((uint64_t)((double)g.windows * share) + 1, total * (uint64_t)rfxk::msp_nmax_of(t->k)), P, ext});
      t->seg_kind = RFX_COUNT_MSP;
      return RFX_OK;
    }
  }
  // (shard passes of a block that has a run map -- the drop-in `jellyfish count` hashes its 4 M-read blocks as they
  // arrive: rfx_count_add of a table with rfx_count_set_passes -- cut their records from reads + map here too)
  rfx_runmap_entry* have = nullptr;
  int G_rep = g.G, G_ovf = 0;
  if (t->runmaps && t->n_shards > 1 && g.bin_hi - g.bin_lo <= 16384u) {
    int mrc = RFX_OK;
    have = runmap_get(t, r, true, &mrc);
    if (mrc) { dfree(c, cur); dfree(c, bin_start); return mrc; }
    if (have) {
      G_rep = rfxk::msp_replay_grid(c, r->n);
      G_ovf = have->n_ovf ? (int)std::min<uint32_t>(64, ((have->n_ovf + 511) / 512 + 7) & ~7u) : 0;
      slab_log2 = std::max(slab_log2, 7);
      cap_a = even + even / 4 + 16384 + rfxk::msp_part1_slack(G_rep + G_ovf, slab_log2);
      if (cap_a >= (1ull << 32)) { dfree(c, cur); dfree(c, bin_start); return RFX_E_RANGE; }
    }
  }
  const uint32_t rows = (uint32_t)(G_rep + G_ovf);
  uint32_t* cnt = (uint32_t*)dmalloc(c, (size_t)rows * P * 4);
  char* buf_a = (char*)dmalloc(c, cap_a * g.c_n * 12);  // 12-byte slots: word and plane side by side
  uint64_t* inst = (uint64_t*)dmalloc(c, cap_b * 8);
  uint32_t* ext = (uint32_t*)dmalloc(c, cap_b * 4);
  auto drop = [&] { dfree(c, cnt); dfree(c, buf_a); };
  if (!cnt || !buf_a || !inst || !ext) {
    drop(); dfree(c, cur); dfree(c, bin_start); dfree(c, inst); dfree(c, ext);
    return RFX_E_NOMEM;
  }
  char* buf_a0 = buf_a - (size_t)g.c_lo * cap_a * 12;  // the address coarse bin 0 would have (see msp_partition_exact)
  HIPCHK(hipMemsetAsync(cur, 0, (g.ncur + 1 + (size_t)P) * 4, c->stream));
  if (have) {
    rfxk::msp_replay(c, rv, have->map, t->k, t->canonical, g.bin_bits, g.bin_lo, g.bin_hi, G_rep, buf_a0, cur, (uint32_t)cap_a, cnt,
                     cur + g.ncur, slab_log2);
    if (have->n_ovf) {  // the reads whose runs did not fit a map: hashed as ever
      rfx_reads_view rvo = rv;
      rvo.idx = have->ovf + 1;
      rvo.n = have->n_ovf;
      rfxk::msp_part1(c, rvo, t->k, t->canonical, g.bin_bits, g.bin_lo, g.bin_hi, 0, G_ovf, buf_a0, cur, (uint32_t)cap_a,
                      cnt + (size_t)G_rep * P, cur + g.ncur, slab_log2);
    }
    ++t->replayed;
  } else {
    rfxk::msp_part1(c, rv, t->k, t->canonical, g.bin_bits, g.bin_lo, g.bin_hi, 0, g.G, buf_a0, cur, (uint32_t)cap_a, cnt, cur + g.ncur,
                    slab_log2);
  }
  rfxk::bin_totals(c, cnt, rows, P, bin_start);
  rfxk::part2(c, (const uint64_t*)buf_a0, inst, bin_start, fine_cur, P2, 32 - g.bin_bits, cur, (uint32_t)cap_a, nullptr, ext, cap_b,
              "k_part2", nullptr, 0, cap_b, g.rec_mode, t->k);
  rfxk::flag_if_gt(c, bin_start + P, cap_b, cur + g.ncur);  // more records than the bin array holds
  drop();
  t->segs->push_back(rfx_segment{inst, cap_b, bin_start, (uint64_t)((double)g.windows * share) + 1, P, ext});
  t->seg_kind = RFX_COUNT_MSP;
  if (t->pend->empty()) c->pend_tables.push_back(t);
  t->pend->push_back(rfx_pending_add{r, cur, t->segs->size() - 1, g.ncur});
  return RFX_OK;
}

// Redo the flagged blocks of `flags` (one per pending add, in order) exactly; forget all pending adds.
static int msp_settle(rfx_table* t, const std::vector<unsigned int>& flags) {
  rfx_ctx* c = t->ctx;
  int rc = RFX_OK;
  for (size_t i = 0; i < t->pend->size() && rc == RFX_OK; ++i) {
    if (!flags[i]) continue;
    const rfx_pending_add& p = (*t->pend)[i];
    rfx_segment& sg = (*t->segs)[p.seg];
    if (!sg.borrowed) dfree(c, sg.inst);
    dfree(c, sg.bin_start);
    if (!sg.borrowed) dfree(c, sg.ext);
    sg.inst = sg.bin_start = nullptr;
    sg.ext = nullptr;
    rfx_segment fresh{};
    rc = msp_partition_exact(t, p.r, &fresh);
    if (rc == RFX_OK) sg = fresh;
  }
  msp_forget_pending(t);
  return rc;
}

// Wait for the flags of the pending adds and settle them (used when a read block goes away first).
static int msp_resolve(rfx_table* t) {
  rfx_ctx* c = t->ctx;
  if (t->pend->empty()) return RFX_OK;
  std::vector<unsigned int> flags(t->pend->size(), 1u);
  for (size_t i = 0; i < t->pend->size(); ++i)
    if (queue_read(c, &flags[i], (*t->pend)[i].cur + (*t->pend)[i].ncur, 4) != hipSuccess) return RFX_E_HIP;
  if (ctx_sync(c) != hipSuccess) return RFX_E_HIP;
  return msp_settle(t, flags);
}

static void rfx_reads_release_pending(const rfx_reads* r) {
  rfx_ctx* c = r->ctx;
  std::vector<rfx_table*> hit;
  for (rfx_table* t : c->pend_tables)
    for (auto& p : *t->pend)
      if (p.r == r) {
        hit.push_back(t);
        break;
      }
  for (rfx_table* t : hit) {
    const int rc = msp_resolve(t);
    if (rc) t->pend_error = rc;
  }
}

// Count every minimizer bin in LDS, then put the survivors in (pos,key) order.  One synchronisation:
// everything is queued (the records array is sized by what the survivor bins can hold), then the
// totals, the capacity flags of this emit AND of the pending adds, and the histogram come back together.
// Queueing and collecting are separate steps so that a caller can queue several tables before it
// waits for the first (rfx_count_finish_begin / _end).
// words behind the ncur coarse cursors of an emit: [ncur] a capacity came short (the emit is run again), [ncur + 1] a bin
// could not be split far enough (error), [ncur + 2] staging chunks the leaf's pool came short by (max over its launches)
constexpr size_t RFX_CUR_TAIL = 3;

struct rfx_finish {
  rfx_table* t = nullptr;
  uint32_t stage_extra = 0;  // staging chunks on top of the estimate: what an earlier attempt came short by
  uint64_t lower = 0, upper = 0;
  uint64_t* histo = nullptr;
  rfx_records* ready = nullptr;  // result computed at begin (paths without a queued form)
  bool failed = false;
  // one queued attempt
  bool queued = false;
  uint64_t cap = 0, room = 0, kmers = 0, total_out = 0;
  const uint64_t** d_inst = nullptr;
  uint64_t *aw = nullptr, *bw = nullptr, *bsq = nullptr, *bs1 = nullptr;
  uint32_t *ac = nullptr, *bc = nullptr;
  bool segs_dropped = false;  // the leaf phase of a big table is final: its records are already freed
  uint32_t cb_lo = 0;         // peers: aw / ac hold coarse bins cb_lo .. only (index them through aw0() / ac0())
  rfx_records* big = nullptr;
  std::vector<const uint64_t*> h_ptrs;
  std::vector<uint32_t> h_cur;
  std::vector<unsigned int> pflags;
};

// Survivors the leaf will stage per k-mer instance: what the last emit on this ctx saw (+30 %), else the first guess of
// the survivor store (msp_emit_queue).
static uint64_t msp_surv_key(uint64_t lower, uint64_t kmers) { return (kmers >> 24) | ((uint64_t)(lower >= 2 ? 1 : 0) << 62); }
// Survivors per instance are unknown before counting.  Samples of one run look alike, so the ratio the last emit of a
// table of this size saw -- else the last emit on this ctx -- is the guess (+30 %); the first emit assumes a quarter
// (singletons dropped; 8 % for big inputs, where a rerun is cheaper than the memory) or 60 %.  A guess that is too
// small costs one rerun with the capacity the cursors report.
static double msp_surv_guess(rfx_ctx* c, uint64_t lower, uint64_t kmers) {
  double seen = c->msp_surv_frac[lower >= 2 ? 1 : 0];
  if (kmers >= (1ull << 28)) {  // (small tables: the last emit's ratio, as ever -- their sizes say little)
    auto it = c->msp_surv_by_size.find(msp_surv_key(lower, kmers));
    if (it != c->msp_surv_by_size.end()) seen = it->second;
  }
  double frac = seen > 0 ? seen * 1.3 : (lower >= 2 ? (kmers > (1ull << 32) ? 0.08 : 0.25) : 0.6);
  if (const char* ev = getenv("RFX_MSP_SURV_FRAC")) frac = atof(ev);
  return frac;
}

// the staging pool of a table's leaf launches (rfxk::msp_stage): keys, counts, and ONE zeroed block of fills + per-launch counters
static int leaf_stage_alloc(rfx_ctx* c, uint32_t chunk, uint32_t n_chunks, uint32_t launches, rfxk::msp_stage* st) {
  st->chunk = chunk;
  st->n_chunks = n_chunks;
  st->keys = (uint64_t*)dmalloc(c, (size_t)n_chunks * chunk * 8);
  st->counts = (uint32_t*)dmalloc(c, (size_t)n_chunks * chunk * 4);
  st->fill = (uint32_t*)dmalloc(c, ((size_t)n_chunks + launches) * 4);
  st->more = st->fill ? st->fill + n_chunks : nullptr;
  if (!st->keys || !st->counts || !st->fill) return RFX_E_NOMEM;
  HIPCHK(hipMemsetAsync(st->fill, 0, ((size_t)n_chunks + launches) * 4, c->stream));
  return RFX_OK;
}
static void leaf_stage_free(rfx_ctx* c, rfxk::msp_stage* st) {  // (stream-ordered pool: reused only by later work of the stream)
  dfree(c, st->keys); dfree(c, st->counts); dfree(c, st->fill);
  *st = rfxk::msp_stage{};
}

static void msp_emit_drop(rfx_finish* f) {
  rfx_ctx* c = f->t->ctx;
  dfree(c, f->d_inst); dfree(c, f->aw); dfree(c, f->ac); dfree(c, f->bw); dfree(c, f->bc); dfree(c, f->bsq);
  dfree(c, f->bs1);
  f->d_inst = nullptr;
  f->aw = f->bw = f->bsq = f->bs1 = nullptr;
  f->ac = f->bc = nullptr;
  f->queued = false;
}

// Leaf phase over segments whose partition is too coarse for the LDS table (many read blocks in one table,
// or a WGS-scale sample): the bins are refined chunk by chunk into a scratch buffer -- histogram of the next
// <= 8 bits of every record's bin hash, scan, scatter (twice when more than 8 bits are missing) -- and the
// leaf counts each chunk as it appears.  Plain streaming passes; the scratch is a fraction of the records.
// Segments come in groups of equal bin count (imports from other ranks may differ from local blocks): every
// group below the target is refined on its own, the leaf then sees all groups' versions of a chunk's bins.
static int msp_leaf_refined(rfx_finish* f, int to_bits, const std::vector<std::vector<uint64_t>>& h_bs, int sel_bits,
                            uint32_t* cur, size_t ncur) {
  rfx_table* t = f->t;
  rfx_ctx* c = t->ctx;
  const int rec_mode = t->canonical ? 1 : 2;
  struct group {
    uint32_t b = 0;
    int from_bits = 0, f1bits = 0, f2bits = 0;
    std::vector<size_t> segs;
    uint64_t *cb1 = nullptr, *cb2 = nullptr, *fine1 = nullptr, *fine2 = nullptr;
    uint32_t *ce1 = nullptr, *ce2 = nullptr;  // wide records: scratch for the 32-bit plane
    uint64_t max_chunk = 0;
    const uint64_t** d_seg = nullptr;  // device: records / bin extents / planes of the group's segments (part2_multi)
  };
  const bool wide = rfxk::msp_wide(t->k);
  std::map<uint32_t, group> groups;
  uint64_t kmers = 0;
  for (size_t i = 0; i < t->segs->size(); ++i) {
    group& g = groups[(*t->segs)[i].bins];
    g.b = (*t->segs)[i].bins;
    g.segs.push_back(i);
    kmers += (*t->segs)[i].kmers;
  }
  const uint32_t bmin = groups.begin()->first;
  const uint32_t Ftot = (1u << to_bits) / bmin;  // fine bins per coarsest bin
  for (auto& kv : groups) {
    group& g = kv.second;
    g.from_bits = ceil_log2(g.b);
    const int fbits = to_bits - g.from_bits;
    if (fbits > 16) {
      snprintf(g_err, sizeof g_err, "MSP: partition would need %d more bits", fbits);
      return RFX_E_FULL;
    }
    g.f1bits = std::min(8, fbits);
    g.f2bits = fbits - g.f1bits;
  }
  // chunks of coarsest-level bins, cut by size (all groups together)
  std::vector<uint64_t> sz(bmin, 0);
  uint64_t R = 0;
  for (auto& kv : groups)
    for (size_t si : kv.second.segs) {
      const uint32_t r = kv.second.b / bmin;
      for (uint32_t p = 0; p < bmin; ++p) sz[p] += h_bs[si][(size_t)(p + 1) * r] - h_bs[si][(size_t)p * r];
      R += h_bs[si][kv.second.b];
    }
  const uint64_t target = std::max<uint64_t>(R / 16, 1ull << 25);
  const uint32_t max_parents = std::max<uint32_t>(1, (1u << 22) / Ftot);
  // A shard pass fills only its share of the bins: the empty stretches before and behind it stay out of the chunks.
  // (They used to ride along with the first and the last chunk: with two passes half of the 8.4 M fine bins of a W
  // launch sequence were empty, and an empty bin still costs the leaf its three barriers and the scan of its table --
  // 2.4 us, 17 % of the kernel's time at W, a third with four passes.)
  uint32_t p_first = 0, p_end = bmin;
  while (p_first < p_end && sz[p_first] == 0) ++p_first;
  while (p_end > p_first && sz[p_end - 1] == 0) --p_end;
  if (p_first == p_end) p_first = 0, p_end = bmin;  // (nothing at all: one chunk of empty bins, as before)
  std::vector<uint32_t> cut{p_first};
  uint64_t acc = 0;
  for (uint32_t p = p_first; p < p_end; ++p) {
    if (p > cut.back() && (acc + sz[p] > target || p - cut.back() >= max_parents)) {
      cut.push_back(p);
      acc = 0;
    }
    acc += sz[p];
  }
  cut.push_back(p_end);
  uint32_t max_np = 0;
  for (size_t i = 0; i + 1 < cut.size(); ++i) max_np = std::max(max_np, cut[i + 1] - cut[i]);
  auto drop = [&] {
    for (auto& kv : groups) {
      dfree(c, kv.second.cb1); dfree(c, kv.second.cb2); dfree(c, kv.second.fine1); dfree(c, kv.second.fine2);
      dfree(c, kv.second.ce1); dfree(c, kv.second.ce2); dfree(c, kv.second.d_seg);
    }
  };
  size_t nleaf = 0;  // segments the leaf sees: one per refined group + the segments already at the target
  for (auto& kv : groups) {
    group& g = kv.second;
    if (g.f1bits == 0) {
      nleaf += g.segs.size();
      continue;
    }
    ++nleaf;
    const uint32_t r = g.b / bmin;
    for (size_t ci = 0; ci + 1 < cut.size(); ++ci) {
      uint64_t ch = 0;
      for (size_t si : g.segs) ch += h_bs[si][(size_t)cut[ci + 1] * r] - h_bs[si][(size_t)cut[ci] * r];
      g.max_chunk = std::max(g.max_chunk, ch);
    }
    const size_t n1max = (size_t)max_np * r << g.f1bits, n2max = (size_t)max_np * Ftot;
    g.cb1 = (uint64_t*)dmalloc(c, std::max<uint64_t>(g.max_chunk, 1) * 8);
    g.fine1 = (uint64_t*)dmalloc(c, (n1max + 1) * 8 + n1max * 4);
    if (wide) g.ce1 = (uint32_t*)dmalloc(c, std::max<uint64_t>(g.max_chunk, 1) * 4);
    if (g.f2bits) {
      g.cb2 = (uint64_t*)dmalloc(c, std::max<uint64_t>(g.max_chunk, 1) * 8);
      g.fine2 = (uint64_t*)dmalloc(c, (n2max + 1) * 8 + n2max * 4);
      if (wide) g.ce2 = (uint32_t*)dmalloc(c, std::max<uint64_t>(g.max_chunk, 1) * 4);
    }
    if (!g.cb1 || !g.fine1 || (g.f2bits && (!g.cb2 || !g.fine2)) || (wide && (!g.ce1 || (g.f2bits && !g.ce2)))) {
      drop();
      return RFX_E_NOMEM;
    }
    // The first refinement level reads the slices of ALL the group's segments in one launch per chunk
    // (rfxk::part2_multi): with a launch per segment (RFX_PART3_PER_SEG=1) a W sample took 646 launches of 2.5 tiles
    // per workgroup.
    if (g.segs.size() > 1 && g.segs.size() <= (size_t)rfxk::part2_max_segs && !getenv("RFX_PART3_PER_SEG")) {
      const size_t ns = g.segs.size();
      std::vector<const uint64_t*> hp(3 * ns, nullptr);
      for (size_t x = 0; x < ns; ++x) {
        const rfx_segment& sg = (*t->segs)[g.segs[x]];
        hp[x] = sg.inst;
        hp[ns + x] = sg.bin_start;
        hp[2 * ns + x] = (const uint64_t*)sg.ext;
      }
      g.d_seg = (const uint64_t**)dmalloc(c, 3 * ns * sizeof(void*));
      if (!g.d_seg) { drop(); return RFX_E_NOMEM; }
      const hipError_t e = upload(c, g.d_seg, hp.data(), 3 * ns * sizeof(void*));
      if (e != hipSuccess) { drop(); return hip_fail(e, "msp_leaf_refined"); }
    }
  }
  const uint64_t** d_ptrs = (const uint64_t**)dmalloc(c, 3 * nleaf * sizeof(void*) * (cut.size() - 1));
  if (!d_ptrs) { drop(); return RFX_E_NOMEM; }
  int geo = (kmers * (uint64_t)(t->n_shards > 1 ? t->n_shards : 1)) >> to_bits <= 12288 ? 1 : 0;
  if (const char* ev = getenv("RFX_MSP_GEO")) geo = atoi(ev) != 0;
  // the leaf's workgroups stage their survivors here before they scatter them into the coarse pos bins
  uint64_t max_chunk_all = 0;  // records of the largest chunk, all groups together
  for (size_t ci = 0; ci + 1 < cut.size(); ++ci) {
    uint64_t ch = 0;
    for (auto& kv : groups)
      for (size_t si : kv.second.segs)
        ch += h_bs[si][(size_t)cut[ci + 1] * (kv.second.b / bmin)] - h_bs[si][(size_t)cut[ci] * (kv.second.b / bmin)];
    max_chunk_all = std::max(max_chunk_all, ch);
  }
  // (the survivors of the largest chunk: its share of the k-mer instances x the expected survivors per instance)
  uint32_t lgrid = 1, lchunk = 1, lpool = 1;
  const uint64_t est_surv = R ? (uint64_t)((double)kmers * msp_surv_guess(c, f->lower, kmers) * (double)max_chunk_all / (double)R) : 0;
  rfxk::msp_leaf_plan(c, (uint32_t)std::min<size_t>((size_t)max_np * Ftot, 1u << 30), geo, max_chunk_all, est_surv, f->stage_extra,
                      &lgrid, &lchunk, &lpool);
  rfxk::msp_stage stage;
  {
    const int src = leaf_stage_alloc(c, lchunk, lpool, (uint32_t)cut.size(), &stage);
    if (src) { drop(); dfree(c, d_ptrs); leaf_stage_free(c, &stage); return src; }
  }
  for (size_t ci = 0; ci + 1 < cut.size(); ++ci) {
    const uint32_t p0 = cut[ci], np = cut[ci + 1] - cut[ci];
    const size_t n2 = (size_t)np * Ftot;
    std::vector<const uint64_t*> ptrs(3 * nleaf, nullptr);  // records, bin extents, planes (wide records)
    size_t li = 0;
    uint64_t chunk_all = 0;
    for (auto& kv : groups) {
      group& g = kv.second;
      const uint32_t r = g.b / bmin, gp0 = p0 * r, gnp = np * r;
      if (g.f1bits == 0) {  // already at the target: the leaf reads the bins in place
        for (size_t si : g.segs) {
          ptrs[li] = (*t->segs)[si].inst;
          ptrs[nleaf + li] = (*t->segs)[si].bin_start + gp0;
          ptrs[2 * nleaf + li] = (const uint64_t*)(*t->segs)[si].ext;
          ++li;
          chunk_all += h_bs[si][(size_t)gp0 + gnp] - h_bs[si][gp0];
        }
        continue;
      }
      uint64_t chunk = 0;
      for (size_t si : g.segs) chunk += h_bs[si][(size_t)gp0 + gnp] - h_bs[si][gp0];
      chunk_all += chunk;
      const uint32_t F1 = 1u << g.f1bits, F2 = 1u << g.f2bits;
      const int shift1 = 32 - (g.from_bits + g.f1bits), shift2 = 32 - to_bits;
      const size_t n1 = (size_t)gnp * F1;
      uint32_t* fcur1 = (uint32_t*)(g.fine1 + n1 + 1);
      HIPCHK(hipMemsetAsync(g.fine1, 0, (n1 + 1) * 8 + n1 * 4, c->stream));
      if (chunk) {
        if (g.d_seg) {  // (one launch for all the group's segments, as the partition below)
          const size_t ns = g.segs.size();
          bool planes = true;
          for (size_t si : g.segs) planes = planes && (*t->segs)[si].ext != nullptr;
          rfxk::bin_hist_multi(c, g.d_seg, g.d_seg + ns, planes ? (const uint32_t* const*)(g.d_seg + 2 * ns) : nullptr, (int)ns, gp0,
                               gnp, chunk, F1, shift1, rec_mode, t->k, g.fine1);
        } else {
          for (size_t si : g.segs)
            rfxk::bin_hist(c, (*t->segs)[si].inst, (*t->segs)[si].bin_start + gp0, gnp, chunk, F1, shift1, rec_mode, t->k,
                           g.fine1, (*t->segs)[si].ext);
        }
        rfxk::scan_tail(c, g.fine1, n1);
        if (g.d_seg) {
          const size_t ns = g.segs.size();
          rfxk::part2_multi(c, g.d_seg, g.d_seg + ns, wide ? (const uint32_t* const*)(g.d_seg + 2 * ns) : nullptr, (int)ns,
                            gp0, gnp, chunk, g.cb1, g.fine1, fcur1, F1, shift1, g.ce1, rec_mode, t->k, "k_part3");
        } else {
          for (size_t si : g.segs)
            rfxk::part2(c, (*t->segs)[si].inst, g.cb1, g.fine1, fcur1, F1, shift1, nullptr, 0, (*t->segs)[si].ext, g.ce1,
                        ~0ull, "k_part3", (*t->segs)[si].bin_start + gp0, gnp, 0, rec_mode, t->k);
        }
      }
      const uint64_t *leaf_src = g.cb1, *leaf_bs = g.fine1;
      const uint32_t* leaf_ext = g.ce1;
      if (g.f2bits) {
        uint32_t* fcur2 = (uint32_t*)(g.fine2 + n2 + 1);
        HIPCHK(hipMemsetAsync(g.fine2, 0, (n2 + 1) * 8 + n2 * 4, c->stream));
        if (chunk) {
          rfxk::bin_hist(c, g.cb1, g.fine1, (uint32_t)n1, chunk, F2, shift2, rec_mode, t->k, g.fine2, g.ce1);
          rfxk::scan_tail(c, g.fine2, n2);
          rfxk::part2(c, g.cb1, g.cb2, g.fine2, fcur2, F2, shift2, nullptr, 0, g.ce1, g.ce2, ~0ull, "k_part4", g.fine1,
                      (uint32_t)n1, 0, rec_mode, t->k);
        }
        leaf_src = g.cb2;
        leaf_bs = g.fine2;
        leaf_ext = g.ce2;
      }
      ptrs[li] = leaf_src;
      ptrs[nleaf + li] = leaf_bs;
      ptrs[2 * nleaf + li] = (const uint64_t*)leaf_ext;
      ++li;
    }
    if (!chunk_all) continue;
    const uint64_t** d = d_ptrs + 3 * nleaf * ci;
    const hipError_t e = upload(c, d, ptrs.data(), 3 * nleaf * sizeof(void*));
    if (e != hipSuccess) { drop(); dfree(c, d_ptrs); leaf_stage_free(c, &stage); return hip_fail(e, "msp_leaf_refined"); }
    rfxk::msp_leaf(c, d, d + nleaf, (int)nleaf, ptrs[0], ptrs[nleaf], (uint32_t)n2, t->k, t->canonical, t->lut_t, t->ntab,
                   sel_bits, 2 * t->k - 7, t->pos_lo, t->pos_hi, f->lower, f->upper, f->aw, f->ac, cur, (uint32_t)f->cap,
                   cur + ncur, cur + ncur + 1, cur + ncur + 2, geo, (const uint32_t* const*)(d + 2 * nleaf),
                   (const uint32_t*)ptrs[2 * nleaf], stage, (uint32_t)ci, std::min<uint32_t>(lgrid, (uint32_t)n2));
  }
  if (getenv("RFX_STAGE_DEBUG")) {  // (chunks each launch took beyond its workgroups' first ones)
    std::vector<uint32_t> more(cut.size(), 0);
    if (ctx_sync(c) == hipSuccess && hipMemcpy(more.data(), stage.more, cut.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
      uint64_t tot = 0;
      for (uint32_t m : more) tot += m;
      fprintf(stderr, "[rfx stage] geo %d grid %u chunk %u pool %u launches %zu: extra chunks taken %llu (first launch %u), est survivors per launch %llu\n",
              geo, lgrid, lchunk, lpool, cut.size() - 1, (unsigned long long)tot, more[0], (unsigned long long)est_surv);
    }
  }
  drop();  // stream-ordered pool
  dfree(c, d_ptrs);
  leaf_stage_free(c, &stage);
  return RFX_OK;
}

// Bin count the leaf needs for the table's current segments, and (when they are coarser than that) the host
// copies of their bin extents, after settling the pending adds: *refine tells which leaf path applies.
static int msp_prepare_leaf(rfx_table* t, int* to_bits_out, bool* refine_out, std::vector<std::vector<uint64_t>>& h_bs,
                            uint64_t* kmers_out, bool force_refine) {
  rfx_ctx* c = t->ctx;
  uint64_t kmers = 0;
  uint32_t pmax = 0, pmin = ~0u;
  for (auto& sg : *t->segs) {
    kmers += sg.kmers;
    pmax = std::max(pmax, sg.bins);
    pmin = std::min(pmin, sg.bins);
  }
  // One bin count for all segments; more bins when the table holds more than ~1.5 x 16 K instances per
  // bin (several read blocks, or blocks far beyond 1 M reads) -- the LDS table of the leaf is fixed.
  int to_bits = ceil_log2(pmax);
  // (a shard pass fills only its share of the bins: density as if every shard were present)
  const uint64_t kfull = kmers * (uint64_t)(t->n_shards > 1 ? t->n_shards : 1);
  const uint64_t per_bin = 24576;
  while (to_bits < 28 && (kfull >> to_bits) > per_bin) ++to_bits;
  // When the partition has to be refined anyway, one more bit costs nothing (as long as it does not take another
  // level of <= 8 bits): bins of half the size go through the half-size leaf, two workgroups per CU, which hides
  // the barriers of one behind the other -- 117 -> 108 ms per sample on the 1 Gb slice.
  if (!getenv("RFX_MSP_NO_HALF") && pmin < (1u << to_bits) && to_bits < 28) {
    const int lo = ceil_log2(pmin), hi = ceil_log2(pmax);
    auto levels = [](int d) { return d <= 0 ? 0 : (d + 7) / 8; };
    if (levels(to_bits + 1 - lo) == levels(to_bits - lo) && levels(to_bits + 1 - hi) == std::max(levels(to_bits - hi), 1))
      ++to_bits;
  }
  if (getenv("RFX_MSP_REFINE_BITS")) to_bits = std::max(to_bits, atoi(getenv("RFX_MSP_REFINE_BITS")));
  const bool refine = force_refine || pmin < (1u << to_bits);
  h_bs.clear();
  if (refine) {
    // the chunks are cut by size: settle the pending adds and fetch every segment's bin extents (one sync)
    int rc = msp_resolve(t);
    if (rc) return rc;
    h_bs.resize(t->segs->size());
    for (size_t i = 0; i < t->segs->size(); ++i) {
      const rfx_segment& sg = (*t->segs)[i];
      h_bs[i].assign((size_t)sg.bins + 1, 0);
      HIPCHK(hipMemcpyAsync(h_bs[i].data(), sg.bin_start, ((size_t)sg.bins + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(ctx_sync(c));
  }
  *to_bits_out = to_bits;
  *refine_out = refine;
  *kmers_out = kmers;
  return RFX_OK;
}

static int msp_add(rfx_table* t, const rfx_reads* r);

// ---- several devices behind one executable (SURVEY 8(e), row E-cli) ---------------------------------------------------
// N tables, one per device, behind ONE process (runRufus.sh starts binaries: no Python, no RCCL).  Round 4: the scheme
// of `north_star` -- table i is given read blocks i, i + N, ... only (1/N of the uploads, 1/N of the hashing); in
// shard pass s every table partitions ITS blocks restricted to the pass, and the bins of a pass are cut into N owner
// ranges (virtual shard s * N + g of S * N: the cut of rfx_count_set_shard, the same on every device): after a barrier
// table g PULLS the record runs of its range from every table (hipMemcpyPeerAsync -- xGMI when the arenas were opened
// with rfx_ctx_allow_peers; its own run is read where it lies), a second barrier lets the partitions go, and the leaf
// counts complete bins -- no partial counts, no reduce.  (RFX_PEERS_REPLICATE=1: round 3's fallback scheme -- every
// table is given every block and keeps minimizer shard i of N; nothing but survivors changes hands.)
// Either way the survivors change hands once more, by OUTPUT position: the leaf leaves them in 128 coarse pos bins, and
// table i takes over coarse bins [i * 128 / N, (i + 1) * 128 / N) of every table before the usual partition + sort --
// its records are then slice i of the (pos,key)-ordered payload, and the .Jhash is the slices one after the other.
// The finishes of the N tables run concurrently on N host threads and meet at barriers.
struct rfx_peers {
  int n = 0;
  bool sharded = true;  // read-block shard + record pull (false: RFX_PEERS_REPLICATE)
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  bool failed = false;
  struct Seg {  // a published partition of one read block (restricted to the pass)
    const uint64_t* inst = nullptr;
    const uint32_t* ext = nullptr;
    uint32_t bins = 0;
    uint64_t kmers = 0;  // the k-mer instances behind its records (rfx_segment::kmers)
    std::vector<uint64_t> bs;  // host copy of its bin extents
  };
  struct Slot {
    int device = 0;
    const uint64_t* aw = nullptr;
    const uint32_t* ac = nullptr;
    uint64_t cap = 0;
    std::vector<uint32_t> fill;  // per coarse bin
    int passes = 0;              // sharded: the shard passes this table planned (all take the maximum)
    std::vector<Seg> segs;       // sharded: this table's partitions of the pass in flight
  };
  std::vector<Slot> slot;
  bool barrier() {  // false: somebody failed
    std::unique_lock<std::mutex> g(mu);
    if (failed) return false;
    const uint64_t my = gen;
    if (++arrived == n) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return true;
    }
    cv.wait(g, [&] { return gen != my || failed; });
    // A barrier that COMPLETED lets every waiter through, whatever happened since: a table that passed it, failed at once
    // and called abort() before a slower waiter woke up must not send that waiter down the "barrier failed" path -- the
    // two would then disagree about who comes to drain(), and the rest would wait there for ever.
    return gen != my;
  }
  void abort() {
    std::lock_guard<std::mutex> g(mu);
    failed = true;
    cv.notify_all();
  }
  // After a failure between two barriers: every table comes here once its own stream is idle, and leaves when all have
  // -- until then somebody may still be copying out of somebody's partitions (barrier() lets go at once after abort()).
  int drained = 0;
  uint64_t drain_gen = 0;
  void drain() {
    std::unique_lock<std::mutex> g(mu);
    const uint64_t my = drain_gen;
    if (++drained == n) {
      drained = 0;
      ++drain_gen;
      cv.notify_all();
      return;
    }
    cv.wait(g, [&] { return drain_gen != my; });
  }
};

// Sharded peers, pass s of S: publish this table's partitions, pull the record runs of this table's owner range from
// every table, leave them as the table's segments.  `own` receives the table's own partitions: the run of its own
// range is counted where it lies, so they stay until the pass's leaf is through.
static int peers_pull_records(rfx_table* t, int s, int S, std::vector<rfx_segment>& own) {
  rfx_ctx* c = t->ctx;
  rfx_peers* p = t->peers;
  const int me = t->peer_index, N = p->n, V = S * N;
  int rc = msp_resolve(t);  // the capacity flags of the adds: a flagged block is redone before anybody reads it
  rfx_peers::Slot& mine = p->slot[(size_t)me];
  mine.device = c->device;
  mine.segs.clear();
  if (rc == RFX_OK) {
    mine.segs.resize(t->segs->size());
    for (size_t i = 0; i < t->segs->size() && rc == RFX_OK; ++i) {
      const rfx_segment& sg = (*t->segs)[i];
      rfx_peers::Seg& ps = mine.segs[i];
      ps.inst = sg.inst;
      ps.ext = sg.ext;
      ps.bins = sg.bins;
      ps.kmers = sg.kmers;
      ps.bs.assign((size_t)sg.bins + 1, 0);
      if (hipMemcpyAsync(ps.bs.data(), sg.bin_start, ((size_t)sg.bins + 1) * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        rc = RFX_E_HIP;
    }
    if (rc == RFX_OK && ctx_sync(c) != hipSuccess) rc = RFX_E_HIP;
  }
  if (rc != RFX_OK) p->abort();
  if (!p->barrier()) return rc != RFX_OK ? rc : RFX_E_HIP;  // (everybody's partitions of the pass are published)
  own.swap(*t->segs);
  t->segs->clear();
  auto cut = [&](uint32_t bins, int q) { return (uint32_t)(((uint64_t)q * 256 + (uint64_t)V - 1) / (uint64_t)V) * (bins / 256); };
  for (int g = 0; g < N && rc == RFX_OK; ++g) {
    const rfx_peers::Slot& sl = p->slot[(size_t)g];
    for (size_t i = 0; i < sl.segs.size() && rc == RFX_OK; ++i) {
      const rfx_peers::Seg& ps = sl.segs[i];
      const uint32_t lo = cut(ps.bins, s * N + me), hi = cut(ps.bins, s * N + me + 1);
      const uint64_t a = ps.bs[lo], n = ps.bs[hi] - a;
      if (!n) continue;
      // bin extents of the pulled run: empty outside [lo, hi)
      std::vector<uint64_t> bs((size_t)ps.bins + 1);
      const uint64_t base = g == me ? 0 : a;  // (the own run stays where it is: absolute extents)
      for (uint32_t b = 0; b <= ps.bins; ++b) bs[b] = (b < lo ? ps.bs[lo] : b > hi ? ps.bs[hi] : ps.bs[b]) - base;
      uint64_t* d_bs = (uint64_t*)dmalloc(c, ((size_t)ps.bins + 1) * 8);
      if (!d_bs) { rc = RFX_E_NOMEM; break; }
      if (upload(c, d_bs, bs.data(), ((size_t)ps.bins + 1) * 8) != hipSuccess) { dfree(c, d_bs); rc = RFX_E_HIP; break; }
      // k-mer instances behind the run (sizes the leaf's bins and the survivor store): the run's share of what its
      // segment holds, never more than its records can carry.  (Until round 4: n x the AVERAGE k-mers per record of
      // ordinary sequence -- on low-complexity data, where records run to their full length, up to 1.8 x too few:
      // bins planned too coarse, extra split passes or RFX_E_FULL, on this path only.)
      const uint64_t nmax = (uint64_t)rfxk::msp_nmax_of(t->k);
      const uint64_t n_all = ps.bs[ps.bins];  // (the records the segment really holds: rfx_segment::n may be a capacity)
      uint64_t kmers = n_all ? (uint64_t)((double)ps.kmers * (double)n / (double)n_all) + 1 : n * nmax;
      kmers = std::min(kmers, n * nmax);
      if (g == me) {
        rfx_segment sg{const_cast<uint64_t*>(ps.inst), n, d_bs, kmers, ps.bins, const_cast<uint32_t*>(ps.ext)};
        sg.borrowed = true;
        t->segs->push_back(sg);
        continue;
      }
      uint64_t* inst = (uint64_t*)dmalloc(c, n * 8);
      uint32_t* ext = (uint32_t*)dmalloc(c, n * 4);
      if (!inst || !ext) { dfree(c, inst); dfree(c, ext); dfree(c, d_bs); rc = RFX_E_NOMEM; break; }
      hipError_t e;
      if (sl.device == c->device) {  // another ctx on the same device (how a one-GPU box tests this)
        e = rfxk::copy_bytes(c, inst, ps.inst + a, n * 8);
        if (e == hipSuccess) e = rfxk::copy_bytes(c, ext, ps.ext + a, n * 4);
      } else {
        e = hipMemcpyPeerAsync(inst, c->device, ps.inst + a, sl.device, n * 8, c->stream);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(ext, c->device, ps.ext + a, sl.device, n * 4, c->stream);
        if (e != hipSuccess) {  // no peer route: let the runtime stage it
          (void)hipGetLastError();
          e = hipMemcpy(inst, ps.inst + a, n * 8, hipMemcpyDefault);
          if (e == hipSuccess) e = hipMemcpy(ext, ps.ext + a, n * 4, hipMemcpyDefault);
        }
      }
      if (e != hipSuccess) { dfree(c, inst); dfree(c, ext); dfree(c, d_bs); rc = hip_fail(e, "rfx peers record pull"); break; }
      t->segs->push_back(rfx_segment{inst, n, d_bs, kmers, ps.bins, ext});
    }
  }
  if (ctx_sync(c) != hipSuccess && rc == RFX_OK) rc = RFX_E_HIP;  // (a failed puller waits for its own copies too)
  if (rc != RFX_OK) p->abort();
  if (!p->barrier()) {  // somebody failed: nobody frees a partition before everybody's pulls have stopped
    (void)ctx_sync(c);
    p->drain();
    return rc != RFX_OK ? rc : RFX_E_HIP;
  }
  // (every pull is complete: the partitions may go)
  // what nobody reads any more: everything of the own partitions but the arrays the own run lies in
  return RFX_OK;
}

// Deferred adds (rfx_count_set_passes): the read blocks stayed with the caller, nothing was partitioned yet.
// S minimizer-shard passes over them: partition the shard's runs of every block, refine + count (the leaf
// appends the shard's survivors to the coarse pos bins), free the records, next shard.  Only 1/S of the
// sample's super-k-mer records ever exist -- the analogue of jellyfish filling its table, dumping a sorted
// partial file and merging at the end (jf/include/jellyfish/hash_counter.hpp:182-202,
// jf/sub_commands/count_main.cc:326-339), except that the shards are disjoint in k-mer space, so the "merge"
// is the one survivor sort that follows anyway.  Leaves f->aw / f->ac / f->cap / f->bsq filled and f->h_cur
// = the final cursors.
static int msp_passes_leaf(rfx_finish* f) {
  rfx_table* t = f->t;
  rfx_ctx* c = t->ctx;
  const uint32_t P1 = (uint32_t)rfxk::p1_bins();
  const size_t ncur = (size_t)P1 * rfxk::p1_cur_stride(), stride = (size_t)rfxk::p1_cur_stride();
  uint64_t windows = 0;
  for (const rfx_reads* r : *t->deferred) windows += r->windows_of(t->k);
  const int outer_shard = t->n_shards > 1 ? t->shard : 0, outer_n = t->n_shards > 1 ? t->n_shards : 1;
  int S = t->passes;
  if (S <= 0) {  // plan: the records of a pass (8 B per ~3 k-mers) + scratch beside what is already resident
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 64ull << 30;
    const double avail = 0.88 * ((double)free_b + (double)(c->arena_mapped - std::min(c->arena_mapped, c->used)));
    // Beside the records of a pass live the survivors gathered so far (12 B each, ~windows / 20 of them at 30x);
    // their sort at the end (12 + 12 + 20 B each) runs when no pass records are left, so it does not add up
    // with them.  (Planning with the sum took 5 passes for a 30x sample where 2 fit: measured, 6.3 s of finish.)
    const double surv = (double)windows / 20.0 * 14.0;
    // (12 B per record)
    const double records = (double)windows * msp_records_per_window(t->k) * 12.0 * 1.05 / (double)outer_n;
    S = 1;
    while (S < 256 && records / S + surv > avail) ++S;
    if (const char* ev = getenv("RFX_COUNT_PASSES")) S = std::max(1, atoi(ev));
  }
  if (S * outer_n > 256) S = std::max(1, 256 / outer_n);
  const bool sharded = t->peers && t->peers->sharded;
  if (sharded) {  // every table takes the same number of passes: the most anybody planned
    rfx_peers* p = t->peers;
    if (t->passes <= 0) S = std::max(1, std::min(256 / p->n, 2 * S));  // (partition + pulled runs: the records twice)
    p->slot[(size_t)t->peer_index].passes = S;
    if (!p->barrier()) return RFX_E_HIP;
    for (int g = 0; g < p->n; ++g) S = std::max(S, p->slot[(size_t)g].passes);
    if (S * p->n > 256) S = std::max(1, 256 / p->n);
    if (!p->barrier()) return RFX_E_HIP;  // (nobody overwrites its slot before everybody has read it)
  }
  // S > 1: the first pass leaves run maps (rfx_msp.hip) as far as the device has room beside a pass, the later passes
  // replay them instead of hashing the reads again.  (Sharded peers run one pass per table from two devices on.)
  struct own_runmaps {  // the table's own store (made by its deferred adds, or here) goes when the passes are over
    rfx_table* t;
    explicit own_runmaps(rfx_table* t_) : t(t_) {}
    void drop() {
      if (t->runmaps && t->runmaps_owned) {
        rfx_runmaps_free(t->runmaps);
        t->runmaps = nullptr;
        t->runmaps_owned = 0;
      }
    }
    ~own_runmaps() { drop(); }
  } own_maps(t);
  if (S == 1 || sharded) own_maps.drop();  // (one pass: the fused kernel hashes and cuts at once; the maps were for nothing)
  if (S > 1 && !t->runmaps && !sharded && !getenv("RFX_NO_RUNMAP")) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    const double avail = 0.88 * ((double)free_b + (double)(c->arena_mapped - std::min(c->arena_mapped, c->used)));
    // a pass's records + the refinement's chunk buffers (two levels of 12 B over 1/16 of them, and the leaf's staging)
    // + the survivors of ALL passes
    const double pass_rec = (double)windows * msp_records_per_window(t->k) * 12.0 * 1.05 / (double)outer_n / S;
    const double pass = pass_rec * 1.15 + (double)windows / 20.0 * 14.0;
    if (avail - pass > (double)(256u << 20)) {
      t->runmaps = rfx_runmaps_create(c, (uint64_t)((avail - pass) * 0.8));
      t->runmaps_owned = t->runmaps != nullptr;
    }
  }
  const size_t zero_bytes = (size_t)RFX_HISTO_BINS * 8 + (ncur + RFX_CUR_TAIL) * 4;
  f->bsq = (uint64_t*)dmalloc(c, zero_bytes);
  if (!f->bsq) return RFX_E_NOMEM;
  uint32_t* cur = (uint32_t*)((unsigned long long*)f->bsq + RFX_HISTO_BINS);
  HIPCHK(hipMemsetAsync(f->bsq, 0, zero_bytes, c->stream));
  std::vector<uint32_t> cur_prev(ncur + RFX_CUR_TAIL, 0);
  f->h_cur.assign(ncur + RFX_CUR_TAIL, 0);
  f->kmers = 0;
  const rfx_ord_cfg cfg0 = ord_cfg(t, 7);
  // (an outer shard -- rfx_count_set_shard before rfx_count_set_passes: one device of several, rfx_count_set_peers --
  // is cut further: pass s of this table is virtual shard outer * S + s of outer_n * S)
  for (int s = 0; s < S; ++s) {
    t->shard = outer_shard * S + s;
    t->n_shards = outer_n * S;
    for (const rfx_reads* r : *t->deferred) {
      if (r->n == 0) continue;
      const int rc = msp_add(t, r);
      if (rc) return rc;
    }
    // sharded peers: this table's partitions of the pass (other tables read them; the own run is counted where it lies)
    struct own_segments : std::vector<rfx_segment> {
      rfx_ctx* c;
      explicit own_segments(rfx_ctx* c_) : c(c_) {}
      void drop() {
        for (auto& sg : *this) { dfree(c, sg.inst); dfree(c, sg.bin_start); dfree(c, sg.ext); }
        clear();
      }
      ~own_segments() { drop(); }
    } own(c);
    auto drop_own = [&] { own.drop(); };
    if (sharded) {
      const int rc = peers_pull_records(t, s, S, own);
      if (rc) { drop_own(); return rc; }
      t->shard = (outer_shard * S + s) * t->peers->n + t->peer_index;  // (the density the leaf's bin count is planned for)
      t->n_shards = outer_n * S * t->peers->n;
    }
    if (t->segs->empty()) { drop_own(); continue; }
    int to_bits = 0;
    bool refine = false;
    uint64_t kmers = 0;
    std::vector<std::vector<uint64_t>> h_bs;
    int rc = msp_prepare_leaf(t, &to_bits, &refine, h_bs, &kmers, true);
    if (rc) return rc;
    f->kmers += kmers;
    if (!f->aw) {  // first pass with records: the store is sized for ONE pass, then for all (below)
      const double frac = msp_surv_guess(c, f->lower, kmers);
      f->cap = (uint64_t)((double)kmers * frac) / P1;
      f->cap += f->cap / 8 + 4096;
      if (f->cap >= (1ull << 32)) f->cap = (1ull << 32) - 1;
      f->aw = (uint64_t*)dmalloc(c, f->cap * P1 * 8);
      f->ac = (uint32_t*)dmalloc(c, f->cap * P1 * 4);
      if (!f->aw || !f->ac) return RFX_E_NOMEM;
    }
    for (int attempt = 0;; ++attempt) {
      rc = msp_leaf_refined(f, to_bits, h_bs, cfg0.sel_bits, cur, ncur);
      if (rc == RFX_E_NOMEM && msp_give_up_own_runmaps(t)) {  // the maps go, the pass's leaf is queued once more
        HIPCHK(upload(c, cur, cur_prev.data(), (ncur + RFX_CUR_TAIL) * 4));
        rc = msp_leaf_refined(f, to_bits, h_bs, cfg0.sel_bits, cur, ncur);
      }
      if (rc) { (void)ctx_sync(c); return rc; }
      hipError_t e = queue_read(c, f->h_cur.data(), cur, (ncur + RFX_CUR_TAIL) * 4);
      if (e == hipSuccess) e = ctx_sync(c);
      if (e != hipSuccess) return hip_fail(e, "msp_passes");
      if (f->h_cur[ncur + 1]) {
        snprintf(g_err, sizeof g_err, "MSP: a bin could not be split far enough to fit LDS");
        return RFX_E_FULL;
      }
      if (!f->h_cur[ncur]) break;
      // (the leaf's staging pool came short: the survivors of the workgroups that found no chunk are missing from the
      // cursors -- the rerun gets the chunks the counters ask for, and a store that is at least no smaller)
      if (f->h_cur[ncur + 2]) f->stage_extra += f->h_cur[ncur + 2] + f->h_cur[ncur + 2] / 4 + 64;
      // A coarse pos bin overflowed in this pass.  The cursors kept counting: enlarge the store, put back what
      // the earlier passes left (their content and cursors), and run the pass's leaf again.
      if (attempt >= 3) {
        snprintf(g_err, sizeof g_err, "MSP: survivor store did not converge (internal error)");
        return RFX_E_FULL;
      }
      uint64_t need = f->cap;
      for (uint32_t cb = 0; cb < P1; ++cb) need = std::max<uint64_t>(need, f->h_cur[cb * stride]);
      const uint64_t ncap = std::min<uint64_t>(need + need / 16 + 4096, (1ull << 32) - 1);
      uint64_t* naw = (uint64_t*)dmalloc(c, ncap * P1 * 8);
      uint32_t* nac = (uint32_t*)dmalloc(c, ncap * P1 * 4);
      if (!naw || !nac) { dfree(c, naw); dfree(c, nac); return RFX_E_NOMEM; }
      for (uint32_t cb = 0; cb < P1; ++cb) {
        const uint64_t n = cur_prev[cb * stride];
        if (!n) continue;
        HIPCHK(rfxk::copy_bytes(c, naw + cb * ncap, f->aw + cb * f->cap, n * 8));
        HIPCHK(rfxk::copy_bytes(c, nac + cb * ncap, f->ac + cb * f->cap, n * 4));
      }
      dfree(c, f->aw);
      dfree(c, f->ac);
      f->aw = naw;
      f->ac = nac;
      f->cap = ncap;
      cur_prev[ncur] = cur_prev[ncur + 1] = cur_prev[ncur + 2] = 0;
      HIPCHK(upload(c, cur, cur_prev.data(), (ncur + RFX_CUR_TAIL) * 4));
    }
    p2l_drop_segments(t);
    drop_own();
    cur_prev = f->h_cur;
    if (s == 0 && S > 1) {  // now sized for all passes: the shards are equal shares of a hash space
      uint64_t mx = 0;
      for (uint32_t cb = 0; cb < P1; ++cb) mx = std::max<uint64_t>(mx, f->h_cur[cb * stride]);
      const uint64_t ncap = std::min<uint64_t>(mx * (uint64_t)S + mx * (uint64_t)S / 12 + 8192, (1ull << 32) - 1);
      if (ncap > f->cap) {
        uint64_t* naw = (uint64_t*)dmalloc(c, ncap * P1 * 8);
        uint32_t* nac = (uint32_t*)dmalloc(c, ncap * P1 * 4);
        if (!naw || !nac) { dfree(c, naw); dfree(c, nac); return RFX_E_NOMEM; }
        for (uint32_t cb = 0; cb < P1; ++cb) {
          const uint64_t n = f->h_cur[cb * stride];
          if (!n) continue;
          HIPCHK(rfxk::copy_bytes(c, naw + cb * ncap, f->aw + cb * f->cap, n * 8));
          HIPCHK(rfxk::copy_bytes(c, nac + cb * ncap, f->ac + cb * f->cap, n * 4));
        }
        dfree(c, f->aw);
        dfree(c, f->ac);
        f->aw = naw;
        f->ac = nac;
        f->cap = ncap;
      }
    }
  }
  t->shard = outer_shard;
  t->n_shards = outer_n;
  if (!f->aw) {  // no k-mer at all
    f->cap = 1;
    f->aw = (uint64_t*)dmalloc(c, P1 * 8);
    f->ac = (uint32_t*)dmalloc(c, P1 * 4);
    if (!f->aw || !f->ac) return RFX_E_NOMEM;
  }
  t->deferred->clear();
  f->segs_dropped = true;
  return RFX_OK;
}

rfx_peers* rfx_peers_create(int n) {
  if (n < 1 || n > 128) return nullptr;
  rfx_peers* p = new rfx_peers();
  p->n = n;
  p->sharded = getenv("RFX_PEERS_REPLICATE") == nullptr;
  p->slot.resize((size_t)n);
  return p;
}
void rfx_peers_free(rfx_peers* p) { delete p; }

int rfx_count_set_peers(rfx_table* t, rfx_peers* p, int index) {
  if (!t || !p || index < 0 || index >= p->n) return RFX_E_INVAL;
  if (t->passes < 0 || !t->deferred->empty() || !t->segs->empty() || t->table_active) {
    snprintf(g_err, sizeof g_err, "rfx_count_set_peers: call rfx_count_set_passes first, and both before the first add");
    return RFX_E_INVAL;
  }
  t->peers = p;
  t->peer_index = index;
  if (p->sharded) {  // the table is given ITS read blocks; who counts what is settled pass by pass at finish
    t->mode = RFX_COUNT_MSP;
    return RFX_OK;
  }
  return rfx_count_set_shard(t, index, p->n);
}

// After the leaf phase of table i: publish the survivor store, pull this table's coarse bins from everybody.
static int peers_exchange(rfx_finish* f, uint32_t* d_cur, size_t ncur) {
  rfx_table* t = f->t;
  rfx_ctx* c = t->ctx;
  rfx_peers* p = t->peers;
  const int me = t->peer_index, N = p->n;
  const uint32_t P1 = (uint32_t)rfxk::p1_bins(), stride = (uint32_t)rfxk::p1_cur_stride();
  {
    rfx_peers::Slot& s = p->slot[(size_t)me];
    s.device = c->device;
    s.aw = f->aw;
    s.ac = f->ac;
    s.cap = f->cap;
    s.fill.assign(P1, 0);
    for (uint32_t cb = 0; cb < P1; ++cb) s.fill[cb] = (uint32_t)std::min<uint64_t>(f->h_cur[(size_t)cb * stride], f->cap);
  }
  if (!p->barrier()) return RFX_E_HIP;
  const uint32_t lo = (uint32_t)((uint64_t)me * P1 / N), hi = (uint32_t)((uint64_t)(me + 1) * P1 / N);
  std::vector<uint64_t> tot(P1, 0);
  uint64_t ncap = 1;
  for (uint32_t cb = lo; cb < hi; ++cb) {
    for (int g = 0; g < N; ++g) tot[cb] += p->slot[(size_t)g].fill[cb];
    ncap = std::max(ncap, tot[cb]);
  }
  int rc = RFX_OK;
  uint64_t *naw = nullptr;
  uint32_t* nac = nullptr;
  if (ncap >= (1ull << 32)) rc = RFX_E_RANGE;
  const uint32_t nb = hi > lo ? hi - lo : 1;
  if (rc == RFX_OK) {
    naw = (uint64_t*)dmalloc(c, (size_t)nb * ncap * 8);
    nac = (uint32_t*)dmalloc(c, (size_t)nb * ncap * 4);
    if (!naw || !nac) rc = RFX_E_NOMEM;
  }
  if (rc == RFX_OK) {
    std::vector<uint64_t> off(P1, 0);
    for (int g = 0; g < N && rc == RFX_OK; ++g) {
      const rfx_peers::Slot& s = p->slot[(size_t)g];
      for (uint32_t cb = lo; cb < hi && rc == RFX_OK; ++cb) {
        const uint64_t n = s.fill[cb];
        if (!n) continue;
        uint64_t* dw = naw + (size_t)(cb - lo) * ncap + off[cb];
        uint32_t* dc = nac + (size_t)(cb - lo) * ncap + off[cb];
        const uint64_t* sw = s.aw + (size_t)cb * s.cap;
        const uint32_t* sc = s.ac + (size_t)cb * s.cap;
        hipError_t e;
        if (s.device == c->device) {  // own share, or another ctx on the same device (how a one-GPU box tests this)
          e = rfxk::copy_bytes(c, dw, sw, n * 8);
          if (e == hipSuccess) e = rfxk::copy_bytes(c, dc, sc, n * 4);
        } else {
          e = hipMemcpyPeerAsync(dw, c->device, sw, s.device, n * 8, c->stream);
          if (e == hipSuccess) e = hipMemcpyPeerAsync(dc, c->device, sc, s.device, n * 4, c->stream);
          if (e != hipSuccess) {  // no peer route: let the runtime stage it
            (void)hipGetLastError();
            e = hipMemcpy(dw, sw, n * 8, hipMemcpyDefault);
            if (e == hipSuccess) e = hipMemcpy(dc, sc, n * 4, hipMemcpyDefault);
          }
        }
        if (e != hipSuccess) rc = hip_fail(e, "rfx peers exchange");
        off[cb] += n;
      }
    }
    if (rc == RFX_OK && ctx_sync(c) != hipSuccess) rc = RFX_E_HIP;
  }
  if (rc != RFX_OK) p->abort();
  const bool all_ok = p->barrier();  // every pull is complete: the old stores may go
  if (rc != RFX_OK || !all_ok) {
    dfree(c, naw);
    dfree(c, nac);
    return rc != RFX_OK ? rc : RFX_E_HIP;
  }
  dfree(c, f->aw);
  dfree(c, f->ac);
  f->aw = naw;
  f->ac = nac;
  f->cap = ncap;
  f->cb_lo = lo;
  for (uint32_t cb = 0; cb < P1; ++cb) f->h_cur[(size_t)cb * stride] = cb >= lo && cb < hi ? (uint32_t)tot[cb] : 0u;
  f->h_cur[ncur] = f->h_cur[ncur + 1] = f->h_cur[ncur + 2] = 0;
  if (upload(c, d_cur, f->h_cur.data(), (ncur + RFX_CUR_TAIL) * 4) != hipSuccess) return RFX_E_HIP;
  return RFX_OK;
}

static int msp_emit_queue(rfx_finish* f) {
  rfx_table* t = f->t;
  rfx_ctx* c = t->ctx;
  const uint32_t P1 = (uint32_t)rfxk::p1_bins();
  const size_t ncur = (size_t)P1 * rfxk::p1_cur_stride();
  // (a table of a group takes the passes route even without a read block of its own: the others wait for it)
  const bool deferred = !t->deferred->empty() || (t->peers && t->passes >= 0 && t->segs->empty());
  int to_bits = 0;
  bool refine = false;
  uint64_t kmers = 0;
  std::vector<std::vector<uint64_t>> h_bs;
  if (deferred) {
    int rc = msp_passes_leaf(f);
    if (rc == RFX_OK && t->peers) {
      uint32_t* d_cur = (uint32_t*)((unsigned long long*)f->bsq + RFX_HISTO_BINS);
      rc = peers_exchange(f, d_cur, ncur);
    } else if (rc && t->peers) {
      t->peers->abort();
    }
    if (rc) {
      (void)ctx_sync(c);
      msp_emit_drop(f);
      p2l_drop_segments(t);
      return rc;
    }
    refine = true;
    kmers = f->kmers;
  } else {
    const int rc = msp_prepare_leaf(t, &to_bits, &refine, h_bs, &kmers, false);
    if (rc) return rc;
  }
  const uint64_t kfull = kmers * (uint64_t)(t->n_shards > 1 ? t->n_shards : 1);
  const int nseg = (int)t->segs->size();
  f->kmers = kmers;
  const rfx_ord_cfg cfg0 = ord_cfg(t, 7);
  auto fail = [&](int rc) {
    msp_emit_drop(f);
    rfx_records_free(f->big);
    f->big = nullptr;
    return rc;
  };
  hipError_t e = hipSuccess;
  if (!deferred) {
    if (!f->cap) {
      const double frac = msp_surv_guess(c, f->lower, f->kmers);  // (survivors per instance: a guess)
      f->cap = (uint64_t)((double)f->kmers * frac) / P1;
      f->cap += f->cap / 8 + 4096;
    }
    if (f->cap >= (1ull << 32)) f->cap = (1ull << 32) - 1;
    f->aw = (uint64_t*)dmalloc(c, f->cap * P1 * 8);
    f->ac = (uint32_t*)dmalloc(c, f->cap * P1 * 4);
    // one zeroed block: histogram, coarse cursors ([ncur] = capacity flag, [ncur+1] = error)
    const size_t zero_bytes = (size_t)RFX_HISTO_BINS * 8 + (ncur + RFX_CUR_TAIL) * 4;
    f->bsq = (uint64_t*)dmalloc(c, zero_bytes);
    if (!f->aw || !f->ac || !f->bsq) return fail(RFX_E_NOMEM);
    e = hipMemsetAsync(f->bsq, 0, zero_bytes, c->stream);
    if (e != hipSuccess) { hip_fail(e, "msp_emit"); return fail(RFX_E_HIP); }
    f->h_cur.assign(ncur + RFX_CUR_TAIL, 0);
  }
  const uint64_t cap = f->cap, room = cap * P1;
  f->room = room;
  unsigned long long* d_histo = (unsigned long long*)f->bsq;
  uint32_t* cur = (uint32_t*)(d_histo + RFX_HISTO_BINS);
  f->pflags.assign(t->pend->size(), 1u);

  // ---- leaf phase: every minimizer bin counted in LDS, survivors appended to 128 coarse pos bins ----
  if (deferred) {
    // done by msp_passes_leaf, shard by shard
  } else if (!refine) {
    const uint32_t P = t->segs->front().bins;
    f->h_ptrs.assign(3 * (size_t)nseg, nullptr);
    for (int i = 0; i < nseg; ++i) {
      f->h_ptrs[i] = (*t->segs)[i].inst;
      f->h_ptrs[nseg + i] = (*t->segs)[i].bin_start;
      f->h_ptrs[2 * nseg + i] = (const uint64_t*)(*t->segs)[i].ext;  // the plane of wide records (else null)
    }
    f->d_inst = (const uint64_t**)dmalloc(c, 3 * nseg * sizeof(void*));
    if (!f->d_inst) return fail(RFX_E_NOMEM);
    e = upload(c, f->d_inst, f->h_ptrs.data(), 3 * nseg * sizeof(void*));
    if (e != hipSuccess) { hip_fail(e, "msp_emit"); return fail(RFX_E_HIP); }
    // sparse bins (small inputs): half-size workgroups, two per CU (measured: -19 % leaf time at 7.7 K
    // instances per bin, +3 % at 15 K)
    int geo = kfull / P < 8192 ? 1 : 0;
    if (const char* ev = getenv("RFX_MSP_GEO")) geo = atoi(ev) != 0;
    uint64_t n_rec_all = 0;
    for (auto& sg : *t->segs) n_rec_all += sg.n;
    uint32_t lgrid = 1, lchunk = 1, lpool = 1;
    // (no more survivors than the store was sized for: its capacity is the estimate, or what a rerun asked for)
    rfxk::msp_leaf_plan(c, P, geo, n_rec_all, std::min<uint64_t>((uint64_t)((double)kmers * msp_surv_guess(c, f->lower, kmers)), room),
                        f->stage_extra, &lgrid, &lchunk, &lpool);
    rfxk::msp_stage stage;
    {
      const int src = leaf_stage_alloc(c, lchunk, lpool, 1, &stage);
      if (src) { leaf_stage_free(c, &stage); return fail(src); }
    }
    rfxk::msp_leaf(c, f->d_inst, f->d_inst + nseg, nseg, f->h_ptrs[0], f->h_ptrs[nseg], P, t->k, t->canonical, t->lut_t,
                   t->ntab, cfg0.sel_bits, cfg0.c_bits - 7, t->pos_lo, t->pos_hi, f->lower, f->upper, f->aw, f->ac, cur,
                   (uint32_t)cap, cur + ncur, cur + ncur + 1, cur + ncur + 2, geo, (const uint32_t* const*)(f->d_inst + 2 * nseg),
                   (const uint32_t*)f->h_ptrs[2 * nseg], stage, 0, lgrid);
    leaf_stage_free(c, &stage);  // stream-ordered pool
  } else {
    {
      const int rc = msp_leaf_refined(f, to_bits, h_bs, cfg0.sel_bits, cur, ncur);
      if (rc) { (void)ctx_sync(c); return fail(rc); }
    }
    // A big table: look at the flags now, while the records are still there for a rerun, and let the records
    // go before the survivors are sorted (at WGS scale both do not fit side by side).
    e = queue_read(c, f->h_cur.data(), cur, (ncur + RFX_CUR_TAIL) * 4);
    if (e == hipSuccess) e = ctx_sync(c);
    if (e != hipSuccess) { hip_fail(e, "msp_emit"); return fail(RFX_E_HIP); }
    if (f->h_cur[ncur + 1]) {
      snprintf(g_err, sizeof g_err, "MSP: a bin could not be split far enough to fit LDS");
      return fail(RFX_E_FULL);
    }
    if (f->h_cur[ncur]) {  // a coarse pos bin overflowed: once more with what the fullest one needs
      uint64_t need = 1;
      for (uint32_t cb = 0; cb < P1; ++cb) need = std::max<uint64_t>(need, f->h_cur[(size_t)cb * rfxk::p1_cur_stride()]);
      const uint32_t short_by = f->h_cur[ncur + 2];  // ... or the leaf's staging pool came short (or both)
      if (short_by) f->stage_extra += short_by + short_by / 4 + 64;
      if (need <= f->cap) need = short_by ? f->cap : f->cap + f->cap / 4;
      msp_emit_drop(f);
      f->cap = short_by && need <= f->cap ? f->cap : need + need / 64 + 1024;
      return msp_emit_queue(f);
    }
    uint64_t rec_bytes = 0;
    for (auto& sg : *t->segs) rec_bytes += sg.n * 8;
    if (rec_bytes > (4ull << 30)) {  // a big table is consumed by its finish (rufus_hip.h: rfx_count_finish)
      p2l_drop_segments(t);
      f->segs_dropped = true;
    }
  }
  // (peers: the store holds coarse bins cb_lo .. only; the kernels index coarse bins absolutely)
  uint64_t* aw0 = f->aw - (size_t)f->cb_lo * cap;
  uint32_t* ac0 = f->ac - (size_t)f->cb_lo * cap;
  if (f->histo) rfxk::histo_bins(c, ac0, cur, (uint32_t)cap, d_histo);  // count-of-counts of exactly the survivors

  // ---- survivors -> fine pos bins (<= 1536 expected per bin) -> sorted records ----
  // (a big table has just been waited for: its survivor count is known, the arrays below are exact)
  uint64_t out_room = room;
  if (refine) {
    out_room = 0;
    for (uint32_t cb = 0; cb < P1; ++cb) out_room += std::min<uint64_t>(f->h_cur[(size_t)cb * rfxk::p1_cur_stride()], cap);
    if (!out_room) out_room = 1;
  }
  uint32_t Pq = 256;
  while (Pq < (1u << 23) && (uint64_t)Pq * 1536 < out_room) Pq <<= 1;
  const uint32_t Pq1 = std::min<uint32_t>(Pq, 32768), P2a = Pq1 / P1, F2 = Pq / Pq1;
  const rfx_ord_cfg cfg1 = ord_cfg(t, ceil_log2(Pq1)), cfg = ord_cfg(t, ceil_log2(Pq));
  f->bw = (uint64_t*)dmalloc(c, out_room * 8);
  f->bc = (uint32_t*)dmalloc(c, out_room * 4);
  const size_t z1 = ((size_t)Pq1 + 1) * 8 + (size_t)Pq1 * 4, z2 = F2 > 1 ? ((size_t)Pq + 1) * 8 + (size_t)Pq * 4 : 0;
  f->bs1 = (uint64_t*)dmalloc(c, z1 + z2);
  f->big = records_alloc(c, t->k, t->lsize, t->cols, out_room);
  f->room = out_room;
  if (!f->bw || !f->bc || !f->bs1 || !f->big) return fail(RFX_E_NOMEM);
  uint64_t* bs1 = f->bs1;
  uint32_t* fcur1 = (uint32_t*)(bs1 + Pq1 + 1);
  uint64_t* bs2 = F2 > 1 ? (uint64_t*)((char*)bs1 + z1) : nullptr;
  uint32_t* fcur2 = bs2 ? (uint32_t*)(bs2 + Pq + 1) : nullptr;
  e = hipMemsetAsync(bs1, 0, z1 + z2, c->stream);
  if (e != hipSuccess) { hip_fail(e, "msp_emit"); return fail(RFX_E_HIP); }
  rfxk::surv_hist(c, aw0, cur, (uint32_t)cap, P2a, cfg1.bin_shift, bs1);
  rfxk::scan_tail(c, bs1, Pq1);
  rfxk::part2(c, aw0, f->bw, bs1, fcur1, P2a, cfg1.bin_shift, cur, (uint32_t)cap, ac0, f->bc, ~0ull, "k_surv_part2",
              nullptr, 0, out_room);
  const uint64_t *sw = f->bw, *sbs = bs1;
  const uint32_t* sc = f->bc;
  if (F2 > 1) {  // more than 32768 bins: a second level, back into the (now free) coarse arrays
    rfxk::bin_hist(c, f->bw, bs1, Pq1, out_room, F2, cfg.bin_shift, 0, 0, bs2);
    rfxk::scan_tail(c, bs2, Pq);
    rfxk::part2(c, f->bw, f->aw, bs2, fcur2, F2, cfg.bin_shift, nullptr, 0, f->bc, f->ac, ~0ull, "k_surv_part3", bs1, Pq1);
    sw = f->aw;
    sc = f->ac;
    sbs = bs2;
  }
  // every survivor is kept and fine bins are exact, so the sort writes the records in place
  rfxk::surv_sort(c, sw, sc, sbs, Pq, cfg.bin_shift, t->lut_tinv, t->ntab, cfg.sel_bits, f->big->keys, f->big->counts,
                  f->big->pos);
  f->total_out = 0;
  e = queue_read(c, &f->total_out, sbs + Pq, 8);
  if (e == hipSuccess && !refine) e = queue_read(c, f->h_cur.data(), cur, (ncur + RFX_CUR_TAIL) * 4);
  for (size_t i = 0; i < f->pflags.size() && e == hipSuccess; ++i)
    e = queue_read(c, &f->pflags[i], (*t->pend)[i].cur + (*t->pend)[i].ncur, 4);
  if (e == hipSuccess && f->histo) e = queue_read(c, f->histo, d_histo, RFX_HISTO_BINS * 8);
  if (e != hipSuccess) { hip_fail(e, "msp_emit"); (void)ctx_sync(c); return fail(RFX_E_HIP); }
  f->queued = true;
  return RFX_OK;
}

// Wait for a queued attempt.  0: *out is the result; 1: queue again (capacity or a block was redone); < 0: error.
static int msp_emit_collect(rfx_finish* f, rfx_records** out) {
  rfx_table* t = f->t;
  rfx_ctx* c = t->ctx;
  const uint32_t P1 = (uint32_t)rfxk::p1_bins();
  const size_t ncur = (size_t)P1 * rfxk::p1_cur_stride();
  const hipError_t e = ctx_sync(c);  // delivers the read-backs of every queued emit of this ctx
  msp_emit_drop(f);
  rfx_records* big = f->big;
  f->big = nullptr;
  if (e != hipSuccess) { hip_fail(e, "msp_emit"); rfx_records_free(big); return RFX_E_HIP; }
  bool redo = false;
  for (unsigned int x : f->pflags) redo |= x != 0;
  if (!t->pend->empty() && f->pflags.size() == t->pend->size()) {
    const int rc = msp_settle(t, f->pflags);
    if (rc) { rfx_records_free(big); return rc; }
  }
  if (redo) {  // a block was re-partitioned: count again
    rfx_records_free(big);
    return 1;
  }
  if (f->h_cur[ncur + 1]) {
    snprintf(g_err, sizeof g_err, "MSP: a bin could not be split far enough to fit LDS");
    rfx_records_free(big);
    return RFX_E_FULL;
  }
  if (f->h_cur[ncur]) {  // a coarse pos bin overflowed: rerun with what the fullest one needs
    const uint64_t had = f->cap;
    f->cap = 1;
    for (uint32_t cb = 0; cb < P1; ++cb)
      f->cap = std::max<uint64_t>(f->cap, f->h_cur[(size_t)cb * rfxk::p1_cur_stride()]);
    if (const uint32_t short_by = f->h_cur[ncur + 2]) {  // the leaf's staging pool came short: the cursors say too little
      f->stage_extra += short_by + short_by / 4 + 64;
      f->cap = std::max(f->cap, had);
    }
    rfx_records_free(big);
    return 1;
  }
  const uint64_t total_out = f->total_out;
  big->n = total_out;
  if (f->kmers) {
    c->msp_surv_frac[f->lower >= 2 ? 1 : 0] = (double)total_out / (double)f->kmers;
    if (c->msp_surv_by_size.size() > 256) c->msp_surv_by_size.clear();
    c->msp_surv_by_size[msp_surv_key(f->lower, f->kmers)] = (double)total_out / (double)f->kmers;
  }
  // Give a large slack back (exact arrays, device copies); a small one is not worth the 40 B/record of
  // copy traffic -- the arrays return to the pool with the records anyway.
  if ((f->room - total_out) * 20 > (2ull << 30)) {
    rfx_records* fit = records_alloc(c, t->k, t->lsize, t->cols, total_out);
    if (fit) {
      hipError_t e2 = rfxk::copy_bytes(c, fit->keys, big->keys, total_out * 8);
      if (e2 == hipSuccess) e2 = rfxk::copy_bytes(c, fit->counts, big->counts, total_out * 4);
      if (e2 == hipSuccess) e2 = rfxk::copy_bytes(c, fit->pos, big->pos, total_out * 8);
      if (e2 == hipSuccess) {
        rfx_records_free(big);  // stream-ordered pool: reused only by later work of this stream
        big = fit;
      } else {
        rfx_records_free(fit);
      }
    }
  }
  *out = big;
  return 0;
}

static rfx_records* msp_emit_finish(rfx_finish* f) {  // collect, re-queueing as often as the flags ask
  rfx_records* rec = nullptr;
  for (int attempt = 0; attempt < 4; ++attempt) {
    if (!f->queued && msp_emit_queue(f) != RFX_OK) return nullptr;
    const int rc = msp_emit_collect(f, &rec);
    if (rc == 0) return rec;
    if (rc < 0) return nullptr;
  }
  snprintf(g_err, sizeof g_err, "MSP: emit did not converge (internal error)");
  return nullptr;
}

static rfx_records* msp_emit(rfx_table* t, uint64_t lower, uint64_t upper, uint64_t* histo) {
  rfx_finish f;
  f.t = t;
  f.lower = lower;
  f.upper = upper;
  f.histo = histo;
  return msp_emit_finish(&f);
}

static void p2l_drop_segments(rfx_table* t) {
  msp_forget_pending(t);
  for (auto& sg : *t->segs) {
    if (!sg.borrowed) dfree(t->ctx, sg.inst);
    dfree(t->ctx, sg.bin_start);
    if (!sg.borrowed) dfree(t->ctx, sg.ext);
  }
  t->segs->clear();
  t->seg_kind = 0;
}

// Table load (distinct / cap) above which the count kernel stops taking chunks and the host grows
// the table.  Linear probing stays short below it and finish tiles stay sparse enough to sort in LDS.
static const double kLoadLimit = 0.55;

static int grow_for(rfx_table* t, uint64_t distinct_after) {
  uint64_t target = t->cap;
  while ((double)target * kLoadLimit < (double)distinct_after) target <<= 1;
  return target > t->cap ? table_grow(t, target) : RFX_OK;
}

int rfx_count_add(rfx_table* t, const rfx_reads* r) {
  if (!t || !r || t->ctx != r->ctx) return RFX_E_INVAL;
  if (!r->acgt && !r->ulen) return RFX_E_INVAL;
  if (r->ulen && (t->lut_t == nullptr || t->mode == RFX_COUNT_TABLE || t->table_active)) {
    // (the global-table kernel stages the dense mask of a chunk in LDS: it has no sparse form)
    snprintf(g_err, sizeof g_err, "compact read blocks are counted by the partition paths only (2k <= 62, full-rank matrix)");
    return RFX_E_INVAL;
  }
  rfx_ctx* c = t->ctx;
  pin_guard guard(c);
  (void)hipSetDevice(c->device);
  if (r->n == 0) return RFX_OK;
  if (t->passes >= 0) {  // rfx_count_set_passes: counted at finish, shard pass by shard pass
    if (r->windows_of(t->k) >= (1ull << 32)) return RFX_E_RANGE;
    t->deferred->push_back(r);
    // The block is HASHED now -- its run map (rfx_msp.hip) --, while the caller is still parsing input and the device has
    // nothing to do: if the sample then takes more than one shard pass, the passes only cut records from reads + map
    // (`jellyfish count` of a 30x sample: 0.3 s less between "input parsed" and "finished on the device").  Maps of up to
    // an eighth of the device's memory (32 B per read against the 68 of the block itself); one pass: they are dropped unused.
    if (!t->peers && rfxk::msp_k_ok(t->k) && t->lut_t && (!t->runmaps || t->runmaps_owned) && !getenv("RFX_NO_RUNMAP")) {
      if (!t->runmaps) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) {
          t->runmaps = rfx_runmaps_create(c, total_b / 8);
          t->runmaps_owned = t->runmaps != nullptr;
        }
      }
      if (t->runmaps) {
        int mrc = RFX_OK;
        (void)runmap_get(t, r, true, &mrc);
        if (mrc) return mrc;
      }
    }
    return RFX_OK;
  }
  if (t->mode != RFX_COUNT_TABLE && !t->table_active && t->lut_t) {  // P2L needs 2k <= 62 and full-rank M
    // segments of one table are of one kind; MSP where the record format allows it
    const bool msp = t->seg_kind ? t->seg_kind == RFX_COUNT_MSP
                                 : (t->mode != RFX_COUNT_P2L && rfxk::msp_k_ok(t->k) && !getenv("RFX_NO_MSP"));
    if (t->mode == RFX_COUNT_MSP && !msp) {
      snprintf(g_err, sizeof g_err, "RFX_COUNT_MSP needs 23 <= k <= 31");
      return RFX_E_INVAL;
    }
    const int rc = msp ? msp_add(t, r) : p2l_add(t, r);
    if (rc == RFX_OK || t->mode == RFX_COUNT_P2L || t->mode == RFX_COUNT_MSP) return rc;
    if (rc != RFX_E_NOMEM && rc != RFX_E_RANGE) return rc;  // auto: no room for the instance lists
    if (r->ulen) return rc;  // (compact blocks cannot fall back to the global table)
  }
  {
    const int rc = ensure_table(t);
    if (rc) return rc;
    t->table_active = 1;
  }
  // Overflow list: every thread in flight could divert all of its windows after the stop flag is up.
  const uint64_t in_flight = (uint64_t)rfxk::count_reads_grid(c, r->n) * rfxk::count_reads_block();
  const uint64_t need = in_flight * (r->max_len ? r->max_len : 1);
  if (need > t->ovf_cap) {
    dfree(c, t->ovf_keys);
    t->ovf_keys = (uint64_t*)dmalloc(c, need * 8);
    t->ovf_cap = t->ovf_keys ? need : 0;
    if (!t->ovf_keys) return RFX_E_NOMEM;
  }
  HIPCHK(hipMemsetAsync(t->d_ctl, 0, sizeof(rfx_count_ctl), c->stream));
  const uint32_t n_chunks = (r->n + rfxk::count_reads_block() - 1) / rfxk::count_reads_block();
  const rfx_reads_view rv = r->view();
  for (;;) {
    const uint64_t limit = (uint64_t)((double)t->cap * kLoadLimit);
    rfxk::count_reads(c, rv, view_of(t), t->lut, t->k, t->canonical, t->d_stats, t->d_ctl, t->ovf_keys, t->ovf_cap,
                      limit);
    rfx_count_ctl ctl;
    rfx_table_stats st;
    HIPCHK(queue_read(c, &ctl, t->d_ctl, sizeof ctl));
    HIPCHK(queue_read(c, &st, t->d_stats, sizeof st));
    HIPCHK(ctx_sync(c));
    if (ctl.lost || st.overflow) {
      snprintf(g_err, sizeof g_err, "count table overflow list exhausted; counts are not exact");
      return RFX_E_FULL;
    }
    if (ctl.ticket >= n_chunks && ctl.ovf_n == 0) return RFX_OK;
    // Stopped early: grow (at least x2) so that the keys seen so far sit below the load limit,
    // put the diverted keys back, clear the stop flag and resume from the ticket.
    int rc = table_grow(t, t->cap * 2);
    if (rc == RFX_OK) rc = grow_for(t, st.distinct + ctl.ovf_n);
    if (rc) return rc;
    if (ctl.ovf_n) {
      rfxk::count_pairs(c, t->ovf_keys, nullptr, ctl.ovf_n, view_of(t), t->lut, t->d_stats);
      HIPCHK(queue_read(c, &st, t->d_stats, sizeof st));
      HIPCHK(ctx_sync(c));
      if (st.overflow) return RFX_E_FULL;
    }
    ctl.stop = 0;
    ctl.ovf_n = 0;
    HIPCHK(hipMemcpyAsync(t->d_ctl, &ctl, sizeof ctl, hipMemcpyHostToDevice, c->stream));
    HIPCHK(ctx_sync(c));
    if (ctl.ticket >= n_chunks) return RFX_OK;
  }
}

int rfx_count_set_shard(rfx_table* t, int shard, int n_shards) {
  if (!t || n_shards < 1 || n_shards > 256 || shard < 0 || shard >= n_shards) return RFX_E_INVAL;
  if (!t->segs->empty() || t->table_active) {
    snprintf(g_err, sizeof g_err, "rfx_count_set_shard: set the shard before the first rfx_count_add");
    return RFX_E_INVAL;
  }
  if (n_shards > 1 && (!rfxk::msp_k_ok(t->k) || !t->lut_t)) {
    snprintf(g_err, sizeof g_err, "rfx_count_set_shard: minimizer shards need the MSP path (23 <= k <= 31)");
    return RFX_E_INVAL;
  }
  t->shard = shard;
  t->n_shards = n_shards;
  if (n_shards > 1) t->mode = RFX_COUNT_MSP;
  return RFX_OK;
}

int rfx_count_set_early(rfx_table* t, int on) {
  if (!t) return RFX_E_INVAL;
  if (on && (t->n_shards < 2 || t->shard + 1 >= t->n_shards)) {
    snprintf(g_err, sizeof g_err, "rfx_count_set_early: a table of shard s < S - 1 of S > 1 (rfx_count_set_shard) can cut the next shard's records");
    return RFX_E_INVAL;
  }
  t->early_on = on != 0;
  return RFX_OK;
}

int rfx_count_early_segments(const rfx_table* t) { return t && t->early ? (int)t->early->size() : 0; }

int rfx_count_adopt_early(rfx_table* dst, rfx_table* src) {
  if (!dst || !src || dst == src || dst->ctx != src->ctx) return RFX_E_INVAL;
  if (dst->n_shards != src->n_shards || dst->shard != src->shard + 1 || dst->k != src->k || dst->canonical != src->canonical ||
      dst->lsize != src->lsize || memcmp(dst->cols, src->cols, sizeof dst->cols) != 0 || dst->table_active ||
      (dst->seg_kind && dst->seg_kind != RFX_COUNT_MSP) || (dst->p2l_bins && dst->p2l_bins != src->p2l_bins) ||
      dst->passes >= 0 || dst->peers) {
    snprintf(g_err, sizeof g_err, "rfx_count_adopt_early: the adopting table must be shard s + 1 of the same count (k, size, shards, bins)");
    return RFX_E_INVAL;
  }
  if (src->early->empty()) return RFX_OK;
  dst->p2l_bins = src->p2l_bins;
  for (auto& sg : *src->early) dst->segs->push_back(sg);
  src->early->clear();
  dst->seg_kind = RFX_COUNT_MSP;
  return RFX_OK;
}

// ---- run maps ------------------------------------------------------------------------------------------------------
rfx_runmaps* rfx_runmaps_create(rfx_ctx* c, uint64_t budget_bytes) {
  if (!c) return nullptr;
  rfx_runmaps* s = new rfx_runmaps();
  s->ctx = c;
  s->budget = budget_bytes;
  return s;
}

static void runmaps_drop_entry(rfx_runmaps* s, std::map<const rfx_reads*, rfx_runmap_entry>::iterator it) {
  runmaps_release(s, it->second.map, it->second.map_bytes);
  runmaps_release(s, it->second.ovf, it->second.ovf_bytes);
  s->bytes -= std::min<uint64_t>(s->bytes, it->second.bytes);
  s->m.erase(it);
}

rfx_runmaps* rfx_runmaps_create_pooled(rfx_ctx* c, uint64_t pool_bytes) {
  if (!c || !pool_bytes) return nullptr;
  (void)hipSetDevice(c->device);
  rfx_runmaps* s = new rfx_runmaps();
  s->ctx = c;
  s->budget = pool_bytes;
  s->pool = (char*)dmalloc(c, pool_bytes);
  if (!s->pool) {
    delete s;
    return nullptr;
  }
  s->pool_free[0] = pool_bytes;
  return s;
}

void rfx_runmaps_free(rfx_runmaps* s) {
  if (!s) return;
  (void)hipSetDevice(s->ctx->device);
  (void)runmaps_collect(s);
  while (!s->m.empty()) runmaps_drop_entry(s, s->m.begin());
  dfree(s->ctx, s->pool);
  delete s;
}

uint64_t rfx_runmaps_bytes(const rfx_runmaps* s) { return s ? s->bytes : 0; }
int rfx_runmaps_blocks(const rfx_runmaps* s) {
  int n = 0;
  if (s)
    for (auto& kv : s->m) n += kv.second.map != nullptr;
  return n;
}

int rfx_runmaps_drop(rfx_runmaps* s, const rfx_reads* r) {
  if (!s || !r) return RFX_E_INVAL;
  (void)hipSetDevice(s->ctx->device);
  if (s->ahead)  // (only if the block is among the maps on their way: the others may keep flying)
    for (const runmap_pending& p : s->ahead->pend)
      if (p.r == r) {
        (void)runmaps_collect(s);
        break;
      }
  auto it = s->m.find(r);
  if (it != s->m.end()) runmaps_drop_entry(s, it);
  return RFX_OK;
}

int rfx_runmaps_clear(rfx_runmaps* s) {
  if (!s) return RFX_E_INVAL;
  (void)hipSetDevice(s->ctx->device);
  (void)runmaps_collect(s);
  while (!s->m.empty()) runmaps_drop_entry(s, s->m.begin());
  return RFX_OK;
}

// Run maps made AHEAD (round 6).  The hashing launch over a block (k_msp_map: 95 ms per W sample, bound by the
// instructions it issues, 54 KB of LDS per CU) and the partition levels of ANOTHER sample's count (bound by the memory,
// one 96 KB workgroup per CU) want different things of a CU and fit it side by side: the maps of the sample that is
// counted NEXT are queued on the ctx's second stream while this sample's records are partitioned, refined and sorted on
// the first.  Only for a pooled store (its memory is not the stream-ordered arena's) with room for them; what does not
// fit is made by rfx_count_prepare_maps when its turn comes, as before.  Returns the number of launches queued (>= 0).
int rfx_count_prefetch_maps(rfx_table* t, rfx_reads* const* blocks, int n) {
  if (!t || n < 0 || (n && !blocks)) return RFX_E_INVAL;
  rfx_ctx* c = t->ctx;
  rfx_runmaps* st = t->runmaps;
  if (!st || !st->pool || t->mode != RFX_COUNT_MSP || getenv("RFX_NO_RUNMAP") || getenv("RFX_NO_MAP_AHEAD")) return 0;
  (void)hipSetDevice(c->device);
  // the second stream is made at the first call that wants it: a ctx that never hashes ahead keeps one queue
  if (!c->aux && hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    c->aux = nullptr;
    return 0;
  }
  if (st->ahead) {
    const int rc = runmaps_collect(st);
    if (rc) return rc;
  }
  runmap_ahead* a = new runmap_ahead();
  a->k = t->k;
  a->canonical = t->canonical;
  a->pend.reserve((size_t)n);
  // the second stream starts behind what the first has queued so far: a region of the pool that a dropped map gave back
  // may still be read by a replay launch of the first stream
  hipEvent_t after = nullptr;
  bool ok = hipEventCreateWithFlags(&after, hipEventDisableTiming) == hipSuccess && hipEventRecord(after, c->stream) == hipSuccess &&
            hipStreamWaitEvent(c->aux, after, 0) == hipSuccess;
  if (after) (void)hipEventDestroy(after);
  std::swap(c->stream, c->aux);  // (the launchers take the ctx's stream)
  for (int i = 0; i < n && ok; ++i) {
    const rfx_reads* r = blocks[i];
    if (!r || r->ctx != c || r->max_len > 160 || r->n == 0 || st->m.count(r)) continue;
    bool twice = false;
    for (const runmap_pending& q : a->pend) twice = twice || q.r == r;
    if (twice) continue;
    a->pend.emplace_back();
    if (!runmap_launch(t, r, a->pend.back(), false)) {  // no room: neither will the rest find any
      int rc = RFX_OK;
      (void)runmap_finish(t, a->pend.back(), false, &rc);
      a->pend.pop_back();
      break;
    }
  }
  ok = ok && hipEventCreateWithFlags(&a->done, hipEventDisableTiming) == hipSuccess && hipEventRecord(a->done, c->stream) == hipSuccess;
  std::swap(c->stream, c->aux);
  if (!ok) {  // (nothing flies, or the launches are waited for here)
    (void)hipStreamSynchronize(c->aux);
    int rc = RFX_OK;
    for (runmap_pending& p : a->pend) (void)runmap_finish(t, p, false, &rc);
    if (a->done) (void)hipEventDestroy(a->done);
    delete a;
    return RFX_E_HIP;
  }
  const int launched = (int)a->pend.size();
  if (launched == 0) {
    (void)hipEventDestroy(a->done);
    delete a;
    return 0;
  }
  st->ahead = a;
  return launched;
}

// The run maps of several blocks with ONE wait: a map made on its own (rfx_count_add) waits for its launch to learn how
// many reads went without a map -- 0.75 ms of idle device per block, 57 blocks per W trio.
int rfx_count_prepare_maps(rfx_table* t, rfx_reads* const* blocks, int n) {
  if (!t || n < 0 || (n && !blocks)) return RFX_E_INVAL;
  if (!t->runmaps || t->mode != RFX_COUNT_MSP || getenv("RFX_NO_RUNMAP")) return RFX_OK;
  rfx_ctx* c = t->ctx;
  (void)hipSetDevice(c->device);
  int rc = runmaps_collect(t->runmaps);  // (maps made ahead: theirs are entries now)
  if (rc) return rc;
  std::vector<runmap_pending> pend;
  pend.reserve((size_t)n);  // (the read-backs point into it)
  for (int i = 0; i < n && rc == RFX_OK; ++i) {
    const rfx_reads* r = blocks[i];
    if (!r || r->ctx != c || r->max_len > 160 || r->n == 0) continue;
    if (runmap_get(t, r, false, &rc) || t->runmaps->m.count(r)) continue;  // there already (or known to be no block for one)
    bool twice = false;  // (a block named twice: ONE launch, one entry)
    for (const runmap_pending& q : pend) twice = twice || q.r == r;
    if (twice) continue;
    pend.emplace_back();
    if (!runmap_launch(t, r, pend.back())) {  // no room: neither will the rest find any
      (void)runmap_finish(t, pend.back(), false, &rc);
      pend.pop_back();
      break;
    }
  }
  const bool ok = pend.empty() || ctx_sync(c) == hipSuccess;
  for (runmap_pending& p : pend) (void)runmap_finish(t, p, ok, &rc);
  if (!ok && rc == RFX_OK) rc = RFX_E_HIP;
  return rc;
}

int rfx_count_set_runmaps(rfx_table* t, rfx_runmaps* s) {
  if (!t || (s && s->ctx != t->ctx)) return RFX_E_INVAL;
  if (t->runmaps && t->runmaps_owned) rfx_runmaps_free(t->runmaps);
  t->runmaps_owned = 0;
  t->runmaps = s;
  return RFX_OK;
}

uint64_t rfx_count_replayed(const rfx_table* t) { return t ? t->replayed : 0; }

int rfx_count_set_passes(rfx_table* t, int passes) {
  if (!t || passes < 0 || passes > 256) return RFX_E_INVAL;
  if (!t->segs->empty() || t->table_active || t->n_shards > 1) {
    snprintf(g_err, sizeof g_err, "rfx_count_set_passes: set the passes before the first rfx_count_add, not on a shard table");
    return RFX_E_INVAL;
  }
  if (!rfxk::msp_k_ok(t->k) || !t->lut_t) {
    snprintf(g_err, sizeof g_err, "rfx_count_set_passes: shard passes need the MSP path (23 <= k <= 31)");
    return RFX_E_INVAL;
  }
  t->passes = passes;
  t->mode = RFX_COUNT_MSP;
  return RFX_OK;
}

int rfx_count_set_mode(rfx_table* t, int mode) {
  if (!t || mode < RFX_COUNT_AUTO || mode > RFX_COUNT_MSP) return RFX_E_INVAL;
  if (t->n_shards > 1 && mode != RFX_COUNT_MSP) return RFX_E_INVAL;  // a shard is a set of minimizer bins
  t->mode = mode;
  return RFX_OK;
}

int rfx_count_add_pairs_dev(rfx_table* t, const uint64_t* d_keys, const uint32_t* d_counts, uint64_t n) {
  if (!t || (n && (!d_keys || !d_counts))) return RFX_E_INVAL;
  rfx_ctx* c = t->ctx;
  (void)hipSetDevice(c->device);
  int rc = ensure_table(t);
  if (rc) return rc;
  t->table_active = 1;
  rfx_table_stats st;
  rc = read_stats(t, &st);
  if (rc) return rc;
  rc = grow_for(t, st.distinct + n);  // worst case every pair is a new key
  if (rc) return rc;
  rfxk::count_pairs(c, d_keys, d_counts, n, view_of(t), t->lut, t->d_stats);
  return RFX_OK;
}

int rfx_count_segments(rfx_table* t) {
  if (!t) return RFX_E_INVAL;
  if (t->pend_error) return t->pend_error;
  return t->seg_kind == RFX_COUNT_MSP ? (int)t->segs->size() : 0;
}

int rfx_count_segment_ext(rfx_table* t, int i, const uint32_t** d_ext) {
  if (!t || t->seg_kind != RFX_COUNT_MSP || i < 0 || i >= (int)t->segs->size() || !d_ext) return RFX_E_INVAL;
  *d_ext = (*t->segs)[(size_t)i].ext;
  return RFX_OK;
}

int rfx_count_segment_get(rfx_table* t, int i, const uint64_t** d_records, const uint64_t** d_bin_start, uint32_t* bins,
                          uint64_t* n_records) {
  if (!t || t->seg_kind != RFX_COUNT_MSP || i < 0 || i >= (int)t->segs->size()) return RFX_E_INVAL;
  rfx_ctx* c = t->ctx;
  pin_guard guard(c);
  (void)hipSetDevice(c->device);
  // One synchronisation for both: the capacity flags of the pending adds and the exact record count of
  // the segment (segments are sized optimistically; the count lives on the device).
  for (int round = 0; round < 2; ++round) {
    rfx_segment& sg = (*t->segs)[(size_t)i];
    uint64_t total = 0;
    std::vector<unsigned int> flags(t->pend->size(), 1u);
    HIPCHK(queue_read(c, &total, sg.bin_start + sg.bins, 8));
    for (size_t p = 0; p < flags.size(); ++p) HIPCHK(queue_read(c, &flags[p], (*t->pend)[p].cur + (*t->pend)[p].ncur, 4));
    HIPCHK(ctx_sync(c));
    bool redo = false;
    for (unsigned int f : flags) redo |= f != 0;
    if (!t->pend->empty()) {
      const int rc = msp_settle(t, flags);
      if (rc) return rc;
    }
    if (redo) continue;  // a block was re-partitioned: its segment (maybe this one) is new
    if (d_records) *d_records = sg.inst;
    if (d_bin_start) *d_bin_start = sg.bin_start;
    if (bins) *bins = sg.bins;
    if (n_records) *n_records = total;
    return RFX_OK;
  }
  return RFX_E_HIP;
}

int rfx_count_add_records_ext_dev(rfx_table* t, const uint64_t* d_records, const uint32_t* d_ext, uint64_t n_records,
                                  const uint64_t* d_bin_start, uint32_t bins) {
  if (!t || !d_bin_start || (n_records && !d_records) || bins < 256 || (bins & (bins - 1))) return RFX_E_INVAL;
  rfx_ctx* c = t->ctx;
  (void)hipSetDevice(c->device);
  if (!rfxk::msp_k_ok(t->k) || !t->lut_t || t->table_active || (t->seg_kind && t->seg_kind != RFX_COUNT_MSP) ||
      (t->mode != RFX_COUNT_AUTO && t->mode != RFX_COUNT_MSP)) {
    snprintf(g_err, sizeof g_err, "rfx_count_add_records_dev: the table is not on the MSP path");
    return RFX_E_INVAL;
  }
  const bool wide = rfxk::msp_wide(t->k);
  if (wide && n_records && !d_ext) {
    snprintf(g_err, sizeof g_err, "rfx_count_add_records_dev: records come with their 32-bit plane "
                                  "(rfx_count_add_records_ext_dev)");
    return RFX_E_INVAL;
  }
  uint64_t* inst = (uint64_t*)dmalloc(c, (n_records ? n_records : 1) * 8);
  uint32_t* ext = wide ? (uint32_t*)dmalloc(c, (n_records ? n_records : 1) * 4) : nullptr;
  uint64_t* bs = (uint64_t*)dmalloc(c, ((size_t)bins + 1) * 8);
  if (!inst || !bs || (wide && !ext)) { dfree(c, inst); dfree(c, bs); dfree(c, ext); return RFX_E_NOMEM; }
  hipError_t e = rfxk::copy_bytes(c, bs, d_bin_start, ((size_t)bins + 1) * 8);
  if (e == hipSuccess && n_records)
    e = rfxk::copy_bytes(c, inst, d_records, n_records * 8);
  if (e == hipSuccess && n_records && wide)
    e = rfxk::copy_bytes(c, ext, d_ext, n_records * 4);
  if (e != hipSuccess) { dfree(c, inst); dfree(c, bs); dfree(c, ext); return hip_fail(e, "rfx_count_add_records_dev"); }
  if (!t->p2l_bins) t->p2l_bins = bins > 8192 ? 8192 : bins;  // geometry of later rfx_count_add calls
  t->segs->push_back(rfx_segment{inst, n_records, bs, n_records * (uint64_t)rfxk::msp_nmax_of(t->k), bins, ext});
  t->seg_kind = RFX_COUNT_MSP;
  return RFX_OK;
}

// The import without the copy: the table READS the caller's arrays until its finish (or rfx_count_free); only the bin
// offsets are copied.  What the multi-GPU driver receives from its peers is counted where it landed: two copies of a
// shard's records alive at the peak (the sender's partition, the receive buffers) instead of three.
int rfx_count_adopt_records_dev(rfx_table* t, const uint64_t* d_records, const uint32_t* d_ext, uint64_t n_records,
                                const uint64_t* d_bin_start, uint32_t bins) {
  if (!t || !d_bin_start || (n_records && !d_records) || bins < 256 || (bins & (bins - 1))) return RFX_E_INVAL;
  rfx_ctx* c = t->ctx;
  (void)hipSetDevice(c->device);
  if (!rfxk::msp_k_ok(t->k) || !t->lut_t || t->table_active || (t->seg_kind && t->seg_kind != RFX_COUNT_MSP) ||
      (t->mode != RFX_COUNT_AUTO && t->mode != RFX_COUNT_MSP)) {
    snprintf(g_err, sizeof g_err, "rfx_count_adopt_records_dev: the table is not on the MSP path");
    return RFX_E_INVAL;
  }
  if (rfxk::msp_wide(t->k) && n_records && !d_ext) {
    snprintf(g_err, sizeof g_err, "rfx_count_adopt_records_dev: records come with their 32-bit plane");
    return RFX_E_INVAL;
  }
  if (n_records == 0) return RFX_OK;
  uint64_t* bs = (uint64_t*)dmalloc(c, ((size_t)bins + 1) * 8);
  if (!bs) return RFX_E_NOMEM;
  const hipError_t e = rfxk::copy_bytes(c, bs, d_bin_start, ((size_t)bins + 1) * 8);
  if (e != hipSuccess) { dfree(c, bs); return hip_fail(e, "rfx_count_adopt_records_dev"); }
  if (!t->p2l_bins) t->p2l_bins = bins > 8192 ? 8192 : bins;
  rfx_segment sg{const_cast<uint64_t*>(d_records), n_records, bs, n_records * (uint64_t)rfxk::msp_nmax_of(t->k), bins,
                 const_cast<uint32_t*>(d_ext)};
  sg.borrowed = true;
  t->segs->push_back(sg);
  t->seg_kind = RFX_COUNT_MSP;
  return RFX_OK;
}

int rfx_count_add_records_dev(rfx_table* t, const uint64_t* d_records, uint64_t n_records, const uint64_t* d_bin_start,
                              uint32_t bins) {
  return rfx_count_add_records_ext_dev(t, d_records, nullptr, n_records, d_bin_start, bins);
}

int rfx_count_stats(rfx_table* t, uint64_t* distinct, uint64_t* capacity, uint64_t* max_displacement) {
  if (!t) return RFX_E_INVAL;
  (void)hipSetDevice(t->ctx->device);
  rfx_table_stats st;
  int rc = read_stats(t, &st);  // describes the table path; P2L instance lists are counted at finish
  if (rc) return rc;
  if (distinct) *distinct = st.distinct;
  if (capacity) *capacity = t->cap;
  if (max_displacement) *max_displacement = st.max_disp;
  return st.overflow ? RFX_E_FULL : RFX_OK;
}

rfx_finish* rfx_count_finish_begin(rfx_table* t, uint64_t lower, uint64_t upper, uint64_t* histo) {
  if (!t) return nullptr;
  (void)hipSetDevice(t->ctx->device);
  rfx_finish* f = new rfx_finish();
  f->t = t;
  f->lower = lower;
  f->upper = upper;
  f->histo = histo;
  if (!t->pend_error && !t->table_active &&
      ((!t->segs->empty() && t->seg_kind == RFX_COUNT_MSP) || !t->deferred->empty() || (t->peers && t->passes >= 0))) {
    if (msp_emit_queue(f) != RFX_OK) f->failed = true;  // nothing waited for: the work is only queued
  } else {
    f->ready = rfx_count_finish(t, lower, upper, histo);  // paths without a queued form finish here
    f->failed = f->ready == nullptr;
  }
  return f;
}

rfx_records* rfx_count_finish_end(rfx_finish* f) {
  if (!f) return nullptr;
  (void)hipSetDevice(f->t->ctx->device);
  rfx_records* r = f->failed ? nullptr : (f->ready ? f->ready : msp_emit_finish(f));
  if (!r && f->t->peers) f->t->peers->abort();
  if (f->queued) {  // an error path left an attempt in flight: let it drain before its buffers go
    (void)ctx_sync(f->t->ctx);
    msp_emit_drop(f);
    rfx_records_free(f->big);
  }
  delete f;
  return r;
}

static rfx_records* count_finish_inner(rfx_table* t, uint64_t lower, uint64_t upper, uint64_t* histo);
rfx_records* rfx_count_finish(rfx_table* t, uint64_t lower, uint64_t upper, uint64_t* histo) {
  if (!t) return nullptr;
  rfx_records* r = count_finish_inner(t, lower, upper, histo);
  // a table of a device group that fails -- wherever: a pending re-partition, out of memory before the first barrier --
  // must not leave the finishes of the other tables waiting at a barrier
  if (!r && t->peers) t->peers->abort();
  return r;
}
static rfx_records* count_finish_inner(rfx_table* t, uint64_t lower, uint64_t upper, uint64_t* histo) {
  rfx_ctx* c = t->ctx;
  (void)hipSetDevice(c->device);
  if (t->pend_error) {
    snprintf(g_err, sizeof g_err, "a deferred MSP re-partition failed (%s); the table is incomplete", rfx_strerror(t->pend_error));
    return nullptr;
  }
  if (!t->deferred->empty() && !t->table_active) return msp_emit(t, lower, upper, histo);
  // (a table of a device group that was given no read block still takes part: the others wait for it at the barriers)
  if (t->peers && t->passes >= 0 && t->segs->empty() && !t->table_active) return msp_emit(t, lower, upper, histo);
  if (!t->segs->empty()) {
    if (!t->table_active) {
      if (t->seg_kind == RFX_COUNT_MSP) return msp_emit(t, lower, upper, histo);
      rfx_records* r = p2l_emit(t, lower, upper);
      if (r && histo && rfx_records_histo(r, histo) != RFX_OK) { rfx_records_free(r); return nullptr; }
      return r;
    }
    // both paths hold data: fold the instance lists into the table as (key,count) pairs
    rfx_records* part = t->seg_kind == RFX_COUNT_MSP ? msp_emit(t, 1, ~0ull, nullptr) : p2l_emit(t, 1, ~0ull);
    if (!part) return nullptr;
    const int rc = rfx_count_add_pairs_dev(t, part->keys, part->counts, part->n);
    rfx_records_free(part);
    if (rc) return nullptr;
    p2l_drop_segments(t);
  }
  if (ensure_table(t)) return nullptr;
  rfx_table_stats st;
  if (read_stats(t, &st)) return nullptr;
  if (st.overflow) {
    snprintf(g_err, sizeof g_err, "count table overflowed its probe margin; counts are not exact");
    return nullptr;
  }
  for (int attempt = 0; attempt < 8; ++attempt) {
    const uint64_t n_tiles = t->cap / RFX_TILE;
    uint32_t* tile_counts = (uint32_t*)dmalloc(c, n_tiles * 4);
    uint64_t* tile_off = (uint64_t*)dmalloc(c, (n_tiles + 1) * 8);
    uint32_t* d_max = (uint32_t*)dmalloc(c, 4);
    auto cleanup = [&] { dfree(c, tile_counts); dfree(c, tile_off); dfree(c, d_max); };
    if (!tile_counts || !tile_off || !d_max) { cleanup(); return nullptr; }
    const rfx_table_view tv = view_of(t);
    const uint32_t halo = st.max_disp;
    rfxk::tile_count(c, tv, t->lut, halo, lower, upper, tile_counts, n_tiles);
    rfxk::tile_scan(c, tile_counts, n_tiles, tile_off, d_max);
    uint64_t total = 0;
    uint32_t mx = 0;
    hipError_t e = queue_read(c, &total, tile_off + n_tiles, 8);
    if (e == hipSuccess) e = queue_read(c, &mx, d_max, 4);
    if (e == hipSuccess) e = ctx_sync(c);
    if (e != hipSuccess) { hip_fail(e, "rfx_count_finish"); cleanup(); return nullptr; }
    uint32_t sort_cap = 64;
    while (sort_cap < mx) sort_cap <<= 1;
    if (sort_cap > 4096) {
      // tiles too dense for the LDS sort: halve the density by doubling the table
      cleanup();
      if (table_grow(t, t->cap * 2) != RFX_OK || read_stats(t, &st)) return nullptr;
      continue;
    }
    rfx_records* r = records_alloc(c, t->k, t->lsize, t->cols, total);
    if (!r) { cleanup(); return nullptr; }
    rfxk::tile_emit(c, tv, t->lut, halo, lower, upper, tile_off, n_tiles, sort_cap, r->keys, r->counts, r->pos);
    e = ctx_sync(c);
    cleanup();
    if (e != hipSuccess) { hip_fail(e, "tile_emit"); rfx_records_free(r); return nullptr; }
    if (histo && rfx_records_histo(r, histo) != RFX_OK) { rfx_records_free(r); return nullptr; }
    return r;
  }
  snprintf(g_err, sizeof g_err, "rfx_count_finish: tiles stay too dense after growing");
  return nullptr;
}

// ---------------------------------------------------------------------------------------------
uint64_t rfx_records_size(const rfx_records* r) { return r ? r->n : 0; }
int rfx_records_k(const rfx_records* r) { return r ? r->k : 0; }
int rfx_records_lsize(const rfx_records* r) { return r ? r->lsize : 0; }
const uint64_t* rfx_records_dev_keys(const rfx_records* r) { return r ? r->keys : nullptr; }
const uint32_t* rfx_records_dev_counts(const rfx_records* r) { return r ? r->counts : nullptr; }
const uint64_t* rfx_records_dev_pos(const rfx_records* r) { return r ? r->pos : nullptr; }

void rfx_records_free(rfx_records* r) {
  if (!r) return;
  dfree(r->ctx, r->keys); dfree(r->ctx, r->counts); dfree(r->ctx, r->pos);
  delete r;
}

int rfx_records_payload(const rfx_records* r, void* out, size_t cap_bytes, int counter_len) {
  if (!r || counter_len < 1 || counter_len > 8) return RFX_E_INVAL;
  rfx_ctx* c = r->ctx;
  (void)hipSetDevice(c->device);
  const int kb = (2 * r->k + 7) / 8;
  const size_t bytes = (size_t)r->n * (kb + counter_len);
  if (bytes > cap_bytes) return RFX_E_RANGE;
  if (bytes == 0) return RFX_OK;
  if (!out) return RFX_E_INVAL;
  uint8_t* d = (uint8_t*)dmalloc(c, bytes + 4);
  if (!d) return RFX_E_NOMEM;
  rfxk::format_records(c, r->keys, r->counts, r->n, kb, counter_len, d);
  hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = ctx_sync(c);
  dfree(c, d);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_records_payload");
}

int rfx_records_payload_range(const rfx_records* r, uint64_t first, uint64_t n, void* out, size_t cap_bytes, int counter_len) {
  if (!r || counter_len < 1 || counter_len > 8 || first > r->n || n > r->n - first) return RFX_E_INVAL;
  rfx_ctx* c = r->ctx;
  (void)hipSetDevice(c->device);
  const int kb = (2 * r->k + 7) / 8;
  const size_t bytes = (size_t)n * (kb + counter_len);
  if (bytes > cap_bytes) return RFX_E_RANGE;
  if (n == 0) return RFX_OK;
  uint8_t* d = (uint8_t*)dmalloc(c, bytes);
  if (!d) return RFX_E_NOMEM;
  rfxk::format_records(c, r->keys + first, r->counts + first, n, kb, counter_len, d);
  hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = ctx_sync(c);
  dfree(c, d);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_records_payload_range");
}

int rfx_records_get(const rfx_records* r, uint64_t* keys, uint32_t* counts, uint64_t* pos) {
  if (!r) return RFX_E_INVAL;
  (void)hipSetDevice(r->ctx->device);
  if (r->n == 0) return RFX_OK;
  rfx_ctx* c = r->ctx;  // copies ride the ctx stream: the records may still be in flight on it
  if (keys) HIPCHK(hipMemcpyAsync(keys, r->keys, r->n * 8, hipMemcpyDeviceToHost, c->stream));
  if (counts) HIPCHK(hipMemcpyAsync(counts, r->counts, r->n * 4, hipMemcpyDeviceToHost, c->stream));
  if (pos) HIPCHK(hipMemcpyAsync(pos, r->pos, r->n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(ctx_sync(c));
  return RFX_OK;
}

rfx_records* rfx_records_load(rfx_ctx* c, int k, int lsize, const uint64_t* cols, const void* payload, uint64_t n,
                              int counter_len) {
  if (!c || !cols || k < 1 || k > 32 || counter_len < 1 || counter_len > 8 || (n && !payload)) return nullptr;
  (void)hipSetDevice(c->device);
  rfx_records* r = records_alloc(c, k, lsize, cols, n);
  if (!r || n == 0) return r;
  const int kb = (2 * k + 7) / 8;
  const size_t bytes = (size_t)n * (kb + counter_len);
  uint8_t* d = (uint8_t*)dmalloc(c, bytes);
  unsigned int* d_bad = (unsigned int*)dmalloc(c, 4);
  unsigned int bad = 0;
  bool ok = d && d_bad;
  if (ok) ok = hipMemcpyAsync(d, payload, bytes, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
               hipMemsetAsync(d_bad, 0, 4, c->stream) == hipSuccess;
  if (ok) {
    rfxk::parse_records(c, d, n, kb, counter_len, r->keys, r->counts);
    rfxk::compute_pos(c, r->keys, n, r->lut, r->ntab, r->pos);
    rfxk::check_sorted(c, r->keys, r->pos, n, d_bad);
    ok = queue_read(c, &bad, d_bad, 4) == hipSuccess &&
         ctx_sync(c) == hipSuccess;
  }
  dfree(c, d);
  dfree(c, d_bad);
  if (ok && bad) {
    snprintf(g_err, sizeof g_err, "records are not in (pos,key) order for this matrix");
    ok = false;
  }
  if (!ok) {
    rfx_records_free(r);
    return nullptr;
  }
  return r;
}

rfx_records* rfx_records_load_fd(rfx_ctx* c, int k, int lsize, const uint64_t* cols, int fd, uint64_t offset, uint64_t n,
                                 int counter_len) {
  if (!c || !cols || k < 1 || k > 32 || counter_len < 1 || counter_len > 8 || fd < 0) return nullptr;
  (void)hipSetDevice(c->device);
  const bool tr = getenv("RFX_TRACE_LOAD") != nullptr;  // phase times on stderr
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  double t_read = 0;
  rfx_records* r = records_alloc(c, k, lsize, cols, n);
  if (!r || n == 0) return r;
  if (tr) fprintf(stderr, "[load_fd %.3f] records allocated (%llu)\n", since(), (unsigned long long)n);
  const int kb = (2 * k + 7) / 8;
  const size_t rl = (size_t)kb + (size_t)counter_len;
  constexpr int NB = 3;
  const uint64_t step = std::min<uint64_t>(n, 8ull << 20);
  uint8_t* pin[NB] = {nullptr, nullptr, nullptr};
  uint8_t* dev[NB] = {nullptr, nullptr, nullptr};
  hipEvent_t done[NB];
  bool ok = true, have_ev[NB] = {false, false, false};
  const int nb = (int)std::min<uint64_t>(NB, (n + step - 1) / step);
  if (c->load_pin_bytes < step * rl) {  // (a ring kept from an earlier load that is too small: start over)
    for (int i = 0; i < NB; ++i) {
      if (c->load_pin[i]) (void)hipHostFree(c->load_pin[i]);
      c->load_pin[i] = nullptr;
    }
    c->load_pin_bytes = step * rl;
  }
  for (int i = 0; i < nb && ok; ++i) {
    if (!c->load_pin[i]) ok = hipHostMalloc((void**)&c->load_pin[i], c->load_pin_bytes, hipHostMallocDefault) == hipSuccess;
    pin[i] = c->load_pin[i];
    ok = ok && (dev[i] = (uint8_t*)dmalloc(c, step * rl)) &&
         hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
    have_ev[i] = ok;
  }
  unsigned int* d_bad = (unsigned int*)dmalloc(c, 4);
  unsigned int bad = 0;
  ok = ok && d_bad && hipMemsetAsync(d_bad, 0, 4, c->stream) == hipSuccess;
  if (tr) fprintf(stderr, "[load_fd %.3f] ring pinned\n", since());
  const unsigned nt = std::max(1u, std::min(8u, rfx_host_cpus()));
  std::atomic<bool> io_ok{true};
  uint64_t at = 0;
  for (uint64_t i = 0; at < n && ok; ++i, at += step) {
    const int bi = (int)(i % NB);
    const uint64_t m = std::min(step, n - at);
    if (i >= (uint64_t)NB) ok = hipEventSynchronize(done[bi]) == hipSuccess;  // its last upload has left the buffer
    if (!ok) break;
    {  // several readers: one pread stream from the page cache (or a disk) does not keep up with the link
      const double ta = since();
      struct Acc { double& t; double a; decltype(since)& f; ~Acc() { t += f() - a; } } acc{t_read, ta, since};
      const size_t bytes = (size_t)m * rl;
      const unsigned parts = bytes >= (8u << 20) ? nt : 1;
      std::vector<std::thread> th;
      for (unsigned t = 0; t < parts; ++t) {
        const size_t lo = bytes * t / parts, hi = bytes * (t + 1) / parts;
        auto job = [&, lo, hi] {
          size_t got = lo;
          while (got < hi) {
            const ssize_t w = ::pread(fd, pin[bi] + got, hi - got, (off_t)(offset + at * rl + got));
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) { io_ok = false; return; }
            got += (size_t)w;
          }
        };
        if (t + 1 < parts) th.emplace_back(job);
        else job();
      }
      for (auto& x : th) x.join();
    }
    if (!io_ok) {
      snprintf(g_err, sizeof g_err, "short read: the file holds fewer records than its size says");
      ok = false;
      break;
    }
    ok = hipMemcpyAsync(dev[bi], pin[bi], (size_t)m * rl, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
         hipEventRecord(done[bi], c->stream) == hipSuccess;
    if (ok) rfxk::parse_records(c, dev[bi], m, kb, counter_len, r->keys + at, r->counts + at);
  }
  if (ok) {
    rfxk::compute_pos(c, r->keys, n, r->lut, r->ntab, r->pos);
    rfxk::check_sorted(c, r->keys, r->pos, n, d_bad);
    ok = queue_read(c, &bad, d_bad, 4) == hipSuccess;
  }
  if (tr) fprintf(stderr, "[load_fd %.3f] all chunks queued (%.3f s in pread)\n", since(), t_read);
  ok = (ctx_sync(c) == hipSuccess) && ok;
  if (tr) fprintf(stderr, "[load_fd %.3f] device done\n", since());
  for (int i = 0; i < NB; ++i) {
    if (have_ev[i]) (void)hipEventDestroy(done[i]);
    if (dev[i]) dfree(c, dev[i]);
  }
  if (d_bad) dfree(c, d_bad);
  if (ok && bad) {
    snprintf(g_err, sizeof g_err, "records are not in (pos,key) order for this matrix");
    ok = false;
  }
  if (!ok) {
    rfx_records_free(r);
    return nullptr;
  }
  return r;
}

rfx_records* rfx_records_from_dev(rfx_ctx* c, int k, int lsize, const uint64_t* cols, const uint64_t* d_keys,
                                  const uint32_t* d_counts, uint64_t n) {
  if (!c || !cols || (n && (!d_keys || !d_counts))) return nullptr;
  (void)hipSetDevice(c->device);
  rfx_records* r = records_alloc(c, k, lsize, cols, n);
  if (!r || n == 0) return r;
  bool ok = rfxk::copy_bytes(c, r->keys, d_keys, n * 8) == hipSuccess &&
            rfxk::copy_bytes(c, r->counts, d_counts, n * 4) == hipSuccess;
  if (ok) {
    rfxk::compute_pos(c, r->keys, n, r->lut, r->ntab, r->pos);
    ok = ctx_sync(c) == hipSuccess;
  }
  if (!ok) {
    rfx_records_free(r);
    return nullptr;
  }
  return r;
}

int rfx_records_histo(const rfx_records* r, uint64_t* histo) {
  if (!r || !histo) return RFX_E_INVAL;
  rfx_ctx* c = r->ctx;
  (void)hipSetDevice(c->device);
  unsigned long long* d = (unsigned long long*)dmalloc(c, RFX_HISTO_BINS * 8);
  if (!d) return RFX_E_NOMEM;
  hipError_t e = hipMemsetAsync(d, 0, RFX_HISTO_BINS * 8, c->stream);
  if (e == hipSuccess) {
    rfxk::histo(c, r->counts, r->n, d);
    e = queue_read(c, histo, d, RFX_HISTO_BINS * 8);
  }
  if (e == hipSuccess) e = ctx_sync(c);
  dfree(c, d);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_records_histo");
}

int rfx_records_verify(const rfx_records* r, uint32_t min_count, uint32_t max_count, uint64_t out[4]) {
  if (!r || !out) return RFX_E_INVAL;
  rfx_ctx* c = r->ctx;
  (void)hipSetDevice(c->device);
  out[0] = out[1] = out[2] = out[3] = 0;
  if (r->n == 0) return RFX_OK;
  unsigned long long* d = (unsigned long long*)dmalloc(c, 4 * 8);
  if (!d) return RFX_E_NOMEM;
  hipError_t e = hipMemsetAsync(d, 0, 4 * 8, c->stream);
  if (e == hipSuccess) {
    const uint64_t pos_mask = r->lsize >= 64 ? ~0ull : ((1ull << r->lsize) - 1);
    rfxk::records_verify(c, r->keys, r->counts, r->pos, r->n, r->lut, r->ntab, pos_mask, min_count, max_count, d);
    e = queue_read(c, out, d, 4 * 8);
  }
  if (e == hipSuccess) e = ctx_sync(c);
  dfree(c, d);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_records_verify");
}

int rfx_records_checksum(const rfx_records* r, uint64_t out[2]) {
  if (!r || !out) return RFX_E_INVAL;
  rfx_ctx* c = r->ctx;
  (void)hipSetDevice(c->device);
  out[0] = out[1] = 0;
  if (r->n == 0) return RFX_OK;
  unsigned long long* d = (unsigned long long*)dmalloc(c, 2 * 8);
  if (!d) return RFX_E_NOMEM;
  hipError_t e = hipMemsetAsync(d, 0, 2 * 8, c->stream);
  if (e == hipSuccess) {
    rfxk::records_checksum(c, r->keys, r->counts, r->n, d);
    e = queue_read(c, out, d, 2 * 8);
  }
  if (e == hipSuccess) e = ctx_sync(c);
  dfree(c, d);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_records_checksum");
}

// ---------------------------------------------------------------------------------------------
static int same_function(const rfx_records* a, const rfx_records* b) {
  return a->k == b->k && a->lsize == b->lsize && memcmp(a->cols, b->cols, sizeof(uint64_t) * 2 * a->k) == 0;
}

int rfx_merge_unique(rfx_ctx* c, const rfx_records* const* files, int n_files, uint32_t min_count,
                     uint64_t* keys_out, uint32_t* counts_out, uint64_t cap, uint64_t* n_out) {
  if (!c || !files || n_files < 1 || !n_out) return RFX_E_INVAL;
  (void)hipSetDevice(c->device);
  for (int i = 0; i < n_files; ++i) {
    if (!files[i] || files[i]->ctx != c) return RFX_E_INVAL;
    // jf/jellyfish/merge_files.cc:193-203: same key length, size and matrix or refuse
    if (!same_function(files[0], files[i])) return RFX_E_FORMAT;
  }
  std::vector<HostRun> runs(n_files);
  for (int i = 0; i < n_files; ++i) {
    int rc = unique_run(c, files[i], files, n_files, min_count, 0xFFFFFFFFu, runs[i]);
    if (rc) return rc;
  }
  // k-way merge of the per-file runs by (pos, key): each run is already sorted, and k is a handful (a trio: 3) --
  // the smallest head is found by looking at all of them, which beats a heap until k is in the dozens.
  uint64_t total = 0;
  for (int f = 0; f < n_files; ++f) total += runs[f].keys.size();
  *n_out = total;
  if (total > cap) return RFX_E_RANGE;
  std::vector<size_t> at((size_t)n_files, 0);
  for (uint64_t o = 0; o < total; ++o) {
    int best = -1;
    uint64_t bp = 0, bk = 0;
    for (int f = 0; f < n_files; ++f) {
      const size_t i = at[(size_t)f];
      if (i >= runs[f].keys.size()) continue;
      const uint64_t p = runs[f].pos[i], k_ = runs[f].keys[i];
      if (best < 0 || p < bp || (p == bp && k_ < bk)) best = f, bp = p, bk = k_;
    }
    if (keys_out) keys_out[o] = bk;
    if (counts_out) counts_out[o] = runs[best].counts[at[(size_t)best]];
    ++at[(size_t)best];
  }
  return RFX_OK;
}

int rfx_query(const rfx_records* db, const uint64_t* keys, uint64_t n, uint32_t* counts_out) {
  if (!db || (n && (!keys || !counts_out))) return RFX_E_INVAL;
  rfx_ctx* c = db->ctx;
  (void)hipSetDevice(c->device);
  if (n == 0) return RFX_OK;
  uint64_t* dq = (uint64_t*)dmalloc(c, n * 8);
  uint32_t* dout = (uint32_t*)dmalloc(c, n * 4);
  if (!dq || !dout) { dfree(c, dq); dfree(c, dout); return RFX_E_NOMEM; }
  hipError_t e = upload(c, dq, keys, n * 8);
  if (e == hipSuccess) {
    rfxk::query(c, dq, n, db->lut, db->ntab, db->keys, db->pos, db->counts, db->n, dout);
    e = queue_read(c, counts_out, dout, n * 4);
  }
  if (e == hipSuccess) e = ctx_sync(c);
  dfree(c, dq); dfree(c, dout);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_query");
}

int rfx_unique_to_subject(rfx_ctx* c, const rfx_records* subject, const rfx_records* const* others, int n_others,
                          uint32_t min_count, uint32_t min_cov, uint32_t max_cov, uint64_t* keys_out,
                          uint32_t* counts_out, uint64_t cap, uint64_t* n_out) {
  if (!c || !subject || n_others < 0 || (n_others && !others) || !n_out) return RFX_E_INVAL;
  (void)hipSetDevice(c->device);
  std::vector<const rfx_records*> all{subject};
  for (int i = 0; i < n_others; ++i) {
    if (!others[i] || !same_function(subject, others[i])) return RFX_E_FORMAT;
    all.push_back(others[i]);
  }
  HostRun run;
  const uint32_t lo = std::max(min_count, min_cov);
  int rc = unique_run(c, subject, all.data(), (int)all.size(), lo, max_cov, run);
  if (rc) return rc;
  *n_out = run.keys.size();
  if (run.keys.size() > cap) return RFX_E_RANGE;
  if (keys_out) memcpy(keys_out, run.keys.data(), run.keys.size() * 8);
  if (counts_out) memcpy(counts_out, run.counts.data(), run.counts.size() * 4);
  return RFX_OK;
}

rfx_records* rfx_records_subtract(rfx_ctx* c, const rfx_records* a, const rfx_records* const* others, int n_others,
                                  uint32_t min_count, uint32_t max_count) {
  if (!c || !a || n_others < 0 || (n_others && !others)) { snprintf(g_err, sizeof g_err, "rfx_records_subtract: bad argument"); return nullptr; }
  (void)hipSetDevice(c->device);
  for (int i = 0; i < n_others; ++i)
    if (!others[i] || !same_function(a, others[i])) { snprintf(g_err, sizeof g_err, "rfx_records_subtract: databases of different hash functions"); return nullptr; }
  // (several contexts / devices per process are a supported configuration: records of another one would be
  // dereferenced on this one's stream)
  if (a->ctx != c) { snprintf(g_err, sizeof g_err, "rfx_records_subtract: the records belong to another context"); return nullptr; }
  for (int i = 0; i < n_others; ++i)
    if (others[i]->ctx != c) { snprintf(g_err, sizeof g_err, "rfx_records_subtract: the records belong to another context"); return nullptr; }
  pin_guard guard(c);
  if (a->n == 0) return records_alloc(c, a->k, a->lsize, a->cols, 0);
  const uint64_t nblk = (a->n + 2047) / 2048;
  uint8_t* flags = (uint8_t*)dmalloc(c, a->n);
  uint64_t* boff = (uint64_t*)dmalloc(c, nblk * 8);
  unsigned long long* d_tot = (unsigned long long*)dmalloc(c, 8);
  auto cleanup = [&] { dfree(c, flags); dfree(c, boff); dfree(c, d_tot); };
  if (!flags || !boff || !d_tot) { cleanup(); snprintf(g_err, sizeof g_err, "rfx_records_subtract: out of device memory"); return nullptr; }
  rfxk::flag_range(c, a->counts, a->n, min_count, max_count, flags);
  for (int j = 0; j < n_others; ++j)
    rfxk::flag_absent(c, a->keys, a->pos, a->n, others[j]->keys, others[j]->pos, others[j]->n, a->lsize, flags);
  rfxk::compact_count(c, flags, a->n, boff, d_tot);
  unsigned long long tot = 0;
  hipError_t e = queue_read(c, &tot, d_tot, 8);
  if (e == hipSuccess) e = ctx_sync(c);
  if (e != hipSuccess) { cleanup(); hip_fail(e, "rfx_records_subtract"); return nullptr; }
  rfx_records* out = records_alloc(c, a->k, a->lsize, a->cols, tot);
  if (!out) { cleanup(); return nullptr; }
  if (tot) rfxk::compact_scatter(c, flags, a->keys, a->counts, a->pos, a->n, boff, out->keys, out->counts, out->pos);
  cleanup();  // (stream-ordered: the scatter is queued before any reuse)
  return out;
}

// ---------------------------------------------------------------------------------------------
rfx_set* rfx_set_build(rfx_ctx* c, const uint64_t* fwd_keys, uint64_t n, int k) {
  if (!c || k < 1 || k > 32 || (n && !fwd_keys)) return nullptr;
  (void)hipSetDevice(c->device);
  rfx_set* s = new rfx_set();
  memset(s, 0, sizeof *s);
  s->ctx = c;
  s->k = k;
  s->n = n;
  // load below 1/8 (it was 3/8): a wave of the filter waits for the longest probe chain among its candidates
  s->bits = std::max(6, ceil_log2(n * 8 + 1));
  if (s->bits > 31) {
    snprintf(g_err, sizeof g_err, "rfx_set_build: %llu keys are more than a mutant set holds (2^30)", (unsigned long long)n);
    delete s;
    return nullptr;
  }
  s->cap = 1ull << s->bits;
  for (uint64_t i = 0; i < n; ++i)
    if (fwd_keys[i] == RFX_EMPTY) s->has_all_ones = 1;
  // window bitmap: 8x more bits than slots (floor 2^16 = the LDS-resident size, cap 2^26 = 8 MB),
  // indexed by the bits of the word just above its two newest bases
  s->bm_bits = std::min(26, std::max(16, s->bits + 3));
  if (s->bm_bits > 2 * k) s->bm_bits = 2 * k;
  s->bm_shift = 2 * k >= s->bm_bits + 4 ? 4 : 0;
  const size_t bm_words = (size_t)1 << (s->bm_bits > 5 ? s->bm_bits - 5 : 0);
  s->slots = (uint64_t*)dmalloc(c, s->cap * 8);
  s->bitmap = (uint32_t*)dmalloc(c, std::max<size_t>(bm_words, 2048) * 4);
  // the fast filter's own pre-filters (small sets only: two 2^16-bit bitmaps that live in LDS)
  const size_t bm2_words = (k >= 16 && n <= 4096) ? 4096 : 0;
  if (bm2_words) s->bitmap2 = (uint32_t*)dmalloc(c, bm2_words * 4);
  const size_t bm3_words = (k >= 20 && n > 4096 && n <= (1u << 17)) ? (size_t)rfxk::filter_big_words() : 0;
  if (bm3_words) {
    s->bitmap3 = (uint32_t*)dmalloc(c, bm3_words * 4);
    if (!s->bitmap3 || hipMemsetAsync(s->bitmap3, 0, bm3_words * 4, c->stream) != hipSuccess) {
      rfx_set_free(s);
      return nullptr;
    }
  }
  s->bm4_bits = rfxk::filter_q_bits(n, k);
  if (s->bm4_bits) {
    const size_t bm4_bytes = (size_t)4 << (s->bm4_bits - 5);
    s->bitmap4 = (uint32_t*)dmalloc(c, bm4_bytes);
    if (!s->bitmap4 || hipMemsetAsync(s->bitmap4, 0, bm4_bytes, c->stream) != hipSuccess) {
      rfx_set_free(s);
      return nullptr;
    }
  }
  if (const int pair = rfxk::filter_p_applies(n, k)) {
    s->bm5_three = pair == 2;
    s->bitmap5 = (uint32_t*)dmalloc(c, rfxk::filter_p_table_bytes());
    if (!s->bitmap5 || hipMemsetAsync(s->bitmap5, 0, rfxk::filter_p_table_bytes(), c->stream) != hipSuccess) {
      rfx_set_free(s);
      return nullptr;
    }
  }
  uint64_t* dk = (uint64_t*)dmalloc(c, n * 8);
  bool ok = s->slots && dk && s->bitmap && (!bm2_words || s->bitmap2);
  if (ok && bm2_words) ok = hipMemsetAsync(s->bitmap2, 0, bm2_words * 4, c->stream) == hipSuccess;
  if (ok) ok = hipMemsetAsync(s->slots, 0xFF, s->cap * 8, c->stream) == hipSuccess &&
               hipMemsetAsync(s->bitmap, 0, std::max<size_t>(bm_words, 2048) * 4, c->stream) == hipSuccess;
  if (ok && n) ok = upload(c, dk, fwd_keys, n * 8) == hipSuccess;
  if (ok) {
    rfxk::set_insert(c, dk, n, s->slots, s->bits);
    rfxk::set_bitmap(c, dk, n, s->bitmap, s->bm_bits, s->bm_shift);
    if (s->bitmap2) rfxk::set_bitmap_packed(c, dk, n, s->bitmap2);
    if (s->bitmap3) rfxk::set_bitmap_big(c, dk, n, s->bitmap3);
    if (s->bitmap4) rfxk::set_bitmap_q(c, dk, n, s->bitmap4, s->bm4_bits, k);
    if (s->bitmap5) rfxk::set_bitmap_p(c, dk, n, s->bitmap5, k, s->bm5_three);
    // no synchronisation: upload() staged the keys, everything else is ordered on the ctx stream, and a
    // device error surfaces at the first rfx_filter / rfx_annotate (which wait for their results)
  }
  dfree(c, dk);
  if (!ok) {
    rfx_set_free(s);
    return nullptr;
  }
  return s;
}
uint64_t rfx_set_size(const rfx_set* s) { return s ? s->n : 0; }
void rfx_set_free(rfx_set* s) {
  if (!s) return;
  dfree(s->ctx, s->slots);
  dfree(s->ctx, s->bitmap);
  dfree(s->ctx, s->bitmap2);
  dfree(s->ctx, s->bitmap3);
  dfree(s->ctx, s->bitmap4);
  dfree(s->ctx, s->bitmap5);
  delete s;
}

}  // extern "C"
namespace {
// One block's filter queued on the ctx stream: launches + read-backs, no wait.  The device buffers go to `held` (freed by
// the caller after the wait); *nh receives the number of reads over threshold at the next synchronisation.
int filter_queue(rfx_set* s, const rfx_reads* r, int thresh, int last_base_skipped, uint32_t* hits_out, uint64_t* hitmask_out,
                 unsigned long long* nh, std::vector<void*>& held) {
  rfx_ctx* c = s->ctx;
  const uint64_t nmask = ((uint64_t)r->n + 63) / 64;
  // (the queue filter counts into the array whether the caller wants the counts or not)
  const bool use_q = s->bitmap4 && !getenv("RFX_FILTER_GENERIC") && !getenv("RFX_FILTER_OLD");
  // the pair filter (round 6; RFX_FILTER_NO_PAIR at rfx_set_build keeps the set without its table): with thresh = 1 and
  // nobody asking for the counts, the hits set the mask's bits themselves -- no count array, no memset, no second pass
  const bool use_p = s->bitmap5 && use_q && !getenv("RFX_FILTER_NO_PAIR");
  const bool mask_only = use_p && thresh == 1 && !hits_out;
  uint32_t* d_hits = (hits_out || use_q) && !mask_only ? (uint32_t*)dmalloc(c, (size_t)r->n * 4) : nullptr;
  uint64_t* d_mask = (uint64_t*)dmalloc(c, nmask * 8);
  unsigned long long* d_n = (unsigned long long*)dmalloc(c, 8);
  held.push_back(d_hits);
  held.push_back(d_mask);
  held.push_back(d_n);
  if (((hits_out || use_q) && !mask_only && !d_hits) || !d_mask || !d_n) return RFX_E_NOMEM;
  hipError_t e = hipMemsetAsync(d_n, 0, 8, c->stream);
  if (e == hipSuccess && use_q && !mask_only) e = hipMemsetAsync(d_hits, 0, (size_t)r->n * 4, c->stream);
  if (e == hipSuccess && mask_only) e = hipMemsetAsync(d_mask, 0, nmask * 8, c->stream);
  if (e != hipSuccess) return hip_fail(e, "rfx_filter");
  const rfx_reads_view rv = r->view();
  // RFX_FILTER_OLD: round 2's k_filter_fast / k_filter_big (kept for A/B runs and as a second opinion in the tests)
  if (use_p)
    rfxk::filter_p(c, rv, s->slots, s->bits, s->has_all_ones, s->bitmap5, s->bm5_three, s->k, thresh, last_base_skipped,
                   d_hits, d_mask, d_n);
  else if (use_q)
    rfxk::filter_q(c, rv, s->slots, s->bits, s->has_all_ones, s->bitmap4, s->bm4_bits, s->k, thresh, last_base_skipped,
                   d_hits, d_mask, d_n);
  else if (s->bitmap2 && !getenv("RFX_FILTER_GENERIC"))
    rfxk::filter_fast(c, rv, s->slots, s->bits, s->has_all_ones, s->bitmap2, s->k, thresh, last_base_skipped,
                      d_hits, d_mask, d_n);
  else if (s->bitmap3 && !getenv("RFX_FILTER_GENERIC"))
    rfxk::filter_big(c, rv, s->slots, s->bits, s->has_all_ones, s->bitmap3, s->k, thresh, last_base_skipped, d_hits,
                     d_mask, d_n);
  else
    rfxk::filter(c, rv, s->slots, s->bits, s->has_all_ones, s->bitmap, s->bm_bits, s->bm_shift, s->k, thresh,
                 last_base_skipped, d_hits, d_mask, d_n);
  e = queue_read(c, nh, d_n, 8);
  if (e == hipSuccess && hits_out) e = queue_read(c, hits_out, d_hits, (size_t)r->n * 4);
  if (e == hipSuccess && hitmask_out) e = queue_read(c, hitmask_out, d_mask, nmask * 8);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_filter");
}
}  // namespace
extern "C" {

int rfx_filter(rfx_set* s, const rfx_reads* r, int thresh, int last_base_skipped, uint32_t* hits_out,
               uint64_t* hitmask_out, uint64_t* n_hit_reads) {
  if (!s || !r || s->ctx != r->ctx || !r->good) return RFX_E_INVAL;
  rfx_ctx* c = s->ctx;
  pin_guard guard(c);
  (void)hipSetDevice(c->device);
  if (n_hit_reads) *n_hit_reads = 0;
  if (r->n == 0) return RFX_OK;
  std::vector<void*> held;
  unsigned long long nh = 0;
  int rc = filter_queue(s, r, thresh, last_base_skipped, hits_out, hitmask_out, &nh, held);
  if (ctx_sync(c) != hipSuccess && rc == RFX_OK) rc = RFX_E_HIP;  // (also after a failure: a read-back into `nh` may be queued)
  if (rc == RFX_OK && n_hit_reads) *n_hit_reads = nh;
  for (void* p : held) dfree(c, p);
  return rc;
}

// The same for the blocks of a sample with ONE wait: a call per block cost 0.7 ms of host time and an idle device between
// two blocks -- 14 of the 35 ms the W subject's 19 blocks took (round 6).
int rfx_filter_many(rfx_set* s, const rfx_reads* const* blocks, int n, int thresh, int last_base_skipped,
                    uint64_t* const* hitmask_out, uint64_t* n_hit_reads) {
  if (!s || n < 0 || (n && !blocks)) return RFX_E_INVAL;
  rfx_ctx* c = s->ctx;
  for (int i = 0; i < n; ++i)
    if (!blocks[i] || blocks[i]->ctx != c || !blocks[i]->good) return RFX_E_INVAL;
  pin_guard guard(c);
  (void)hipSetDevice(c->device);
  std::vector<void*> held;
  std::vector<unsigned long long> nh((size_t)n, 0);
  int rc = RFX_OK;
  for (int i = 0; i < n && rc == RFX_OK; ++i) {
    if (blocks[i]->n == 0) continue;
    rc = filter_queue(s, blocks[i], thresh, last_base_skipped, nullptr, hitmask_out ? hitmask_out[i] : nullptr, &nh[(size_t)i], held);
  }
  if (ctx_sync(c) != hipSuccess && rc == RFX_OK) rc = RFX_E_HIP;  // (also after a failure: read-backs may be queued)
  if (rc == RFX_OK && n_hit_reads)
    for (int i = 0; i < n; ++i) n_hit_reads[i] = nh[(size_t)i];
  for (void* p : held) dfree(c, p);
  return rc;
}

// ---------------------------------------------------------------------------------------------
int rfx_overlap_score(rfx_ctx* c, const char* a, int alen, const char* const* b, const int* blen, int nb,
                      float min_pct, int min_ovl, int variant, int* out) {
  if (!c || !a || alen < 0 || nb < 0 || (nb && (!b || !blen || !out))) return RFX_E_INVAL;
  (void)hipSetDevice(c->device);
  if (nb == 0) return RFX_OK;
  std::vector<uint32_t> off((size_t)nb + 1, 0);
  int max_blen = 0;
  for (int j = 0; j < nb; ++j) {
    if (blen[j] < 0) return RFX_E_INVAL;
    off[(size_t)j + 1] = off[(size_t)j] + (uint32_t)blen[j];
    max_blen = std::max(max_blen, blen[j]);
  }
  if ((size_t)alen + (size_t)max_blen + 16 > 150 * 1024) return RFX_E_RANGE;  // both strings live in LDS
  std::string cat;
  cat.reserve(off[(size_t)nb]);
  for (int j = 0; j < nb; ++j) cat.append(b[j], (size_t)blen[j]);
  char* d_a = (char*)dmalloc(c, (size_t)alen + 1);
  char* d_b = (char*)dmalloc(c, cat.size() + 1);
  uint32_t* d_off = (uint32_t*)dmalloc(c, off.size() * 4);
  int* d_out = (int*)dmalloc(c, (size_t)nb * 5 * sizeof(int));
  auto cleanup = [&] { dfree(c, d_a); dfree(c, d_b); dfree(c, d_off); dfree(c, d_out); };
  if (!d_a || !d_b || !d_off || !d_out) { cleanup(); return RFX_E_NOMEM; }
  hipError_t e = upload(c, d_a, a, (size_t)alen);
  if (e == hipSuccess && !cat.empty()) e = upload(c, d_b, cat.data(), cat.size());
  if (e == hipSuccess) e = upload(c, d_off, off.data(), off.size() * 4);
  if (e == hipSuccess) {
    rfxk::overlap_score(c, d_a, alen, d_b, d_off, nb, max_blen, min_pct, min_ovl, variant == RFX_OVL_CONTIG,
                        variant == RFX_OVL_CONTIG ? -1 : 0, d_out);
    e = queue_read(c, out, d_out, (size_t)nb * 5 * sizeof(int));
  }
  if (e == hipSuccess) e = ctx_sync(c);
  cleanup();
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_overlap_score");
}

// ---- device-resident sequence pool of the greedy assemblers ----------------------------------------------------
struct rfx_ovl_pool {
  rfx_ctx* ctx;
  char* arena = nullptr;
  size_t cap = 0, used = 0;
  uint64_t* d_off = nullptr;
  int* d_len = nullptr;
  std::vector<uint64_t> off;
  std::vector<int> len, room;  // room: bytes reserved for the entry at its current place
  int* d_cand = nullptr;
  int* d_out = nullptr;
  size_t cand_cap = 0;
  char* d_a = nullptr;
  size_t a_cap = 0;
  int* h_out = nullptr;  // pinned
  int* h_cand = nullptr;
};

static int ovl_arena_grow(rfx_ovl_pool* p, size_t need) {
  rfx_ctx* c = p->ctx;
  size_t ncap = std::max<size_t>(p->cap * 2, p->used + need + (1u << 20));
  char* na = (char*)dmalloc(c, ncap);
  if (!na) return RFX_E_NOMEM;
  if (p->used) HIPCHK(rfxk::copy_bytes(c, na, p->arena, p->used));
  dfree(c, p->arena);  // stream-ordered
  p->arena = na;
  p->cap = ncap;
  return RFX_OK;
}

rfx_ovl_pool* rfx_ovl_pool_create(rfx_ctx* c, const char* const* seqs, const int* lens, int n) {
  if (!c || n < 0 || (n && (!seqs || !lens))) return nullptr;
  (void)hipSetDevice(c->device);
  rfx_ovl_pool* p = new rfx_ovl_pool();
  p->ctx = c;
  p->off.resize((size_t)n);
  p->len.assign(lens, lens + n);
  p->room.resize((size_t)n);
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    if (lens[i] < 0) { delete p; return nullptr; }
    p->off[(size_t)i] = total;
    p->room[(size_t)i] = lens[i] + lens[i] / 2 + 64;  // merges lengthen entries: leave room in place
    total += (size_t)p->room[(size_t)i];
  }
  p->cap = total + total / 4 + (1u << 20);
  p->used = total;
  p->arena = (char*)dmalloc(c, p->cap);
  p->d_off = (uint64_t*)dmalloc(c, std::max<size_t>((size_t)n, 1) * 8);
  p->d_len = (int*)dmalloc(c, std::max<size_t>((size_t)n, 1) * 4);
  p->h_out = (int*)rfx_host_alloc((size_t)1 << 22);
  p->h_cand = (int*)rfx_host_alloc((size_t)1 << 20);
  bool ok = p->arena && p->d_off && p->d_len && p->h_out && p->h_cand;
  if (ok && n) {
    std::string cat(total, 'Z');
    for (int i = 0; i < n; ++i) memcpy(&cat[p->off[(size_t)i]], seqs[i], (size_t)lens[i]);
    ok = hipMemcpyAsync(p->arena, cat.data(), total, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
         hipMemcpyAsync(p->d_off, p->off.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
         hipMemcpyAsync(p->d_len, p->len.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
         ctx_sync(c) == hipSuccess;
  }
  if (!ok) {
    rfx_ovl_pool_free(p);
    return nullptr;
  }
  return p;
}

void rfx_ovl_pool_free(rfx_ovl_pool* p) {
  if (!p) return;
  rfx_ctx* c = p->ctx;
  dfree(c, p->arena); dfree(c, p->d_off); dfree(c, p->d_len); dfree(c, p->d_cand); dfree(c, p->d_out); dfree(c, p->d_a);
  rfx_host_free(p->h_out);
  rfx_host_free(p->h_cand);
  delete p;
}

int rfx_ovl_pool_set(rfx_ovl_pool* p, int idx, const char* seq, int len) {
  if (!p || idx < 0 || idx >= (int)p->len.size() || len < 0 || (len && !seq)) return RFX_E_INVAL;
  rfx_ctx* c = p->ctx;
  (void)hipSetDevice(c->device);
  if (len > p->room[(size_t)idx]) {  // does not fit in place: move the entry to the end of the arena
    const size_t room = (size_t)len + (size_t)len / 2 + 64;
    if (p->used + room > p->cap) {
      const int rc = ovl_arena_grow(p, room);
      if (rc) return rc;
    }
    p->off[(size_t)idx] = p->used;
    p->room[(size_t)idx] = (int)room;
    p->used += room;
  }
  p->len[(size_t)idx] = len;
  if (len) HIPCHK(upload(c, p->arena + p->off[(size_t)idx], seq, (size_t)len));
  HIPCHK(upload(c, p->d_off + idx, &p->off[(size_t)idx], 8));
  HIPCHK(upload(c, p->d_len + idx, &p->len[(size_t)idx], 4));
  return RFX_OK;
}

int rfx_ovl_pool_score(rfx_ovl_pool* p, int query, const char* a_explicit, int a_len, const int* cand, int nb,
                       float min_pct, int min_ovl, int variant, int strands, int* out) {
  if (!p || nb < 0 || (nb && (!cand || !out)) || strands < 0 || strands > 2) return RFX_E_INVAL;
  if (!a_explicit && (query < 0 || query >= (int)p->len.size())) return RFX_E_INVAL;
  rfx_ctx* c = p->ctx;
  (void)hipSetDevice(c->device);
  if (nb == 0) return RFX_OK;
  const int alen = a_explicit ? a_len : p->len[(size_t)query];
  int max_blen = 0;
  for (int j = 0; j < nb; ++j) {
    if (cand[j] < 0 || cand[j] >= (int)p->len.size()) return RFX_E_INVAL;
    max_blen = std::max(max_blen, p->len[(size_t)cand[j]]);
  }
  const size_t lds = (size_t)alen + (size_t)max_blen + 16;
  if (lds > 150 * 1024) return RFX_E_RANGE;  // both strings live in LDS
  const int nstr = strands == 2 ? 2 : 1;
  const size_t n_out = (size_t)nb * nstr * 5;
  if ((size_t)nb > p->cand_cap) {  // persistent device buffers, grown on demand: no allocation per call
    dfree(c, p->d_cand);
    dfree(c, p->d_out);
    p->cand_cap = std::max<size_t>((size_t)nb * 2, 4096);
    p->d_cand = (int*)dmalloc(c, p->cand_cap * 4);
    p->d_out = (int*)dmalloc(c, p->cand_cap * 2 * 5 * 4);
    if (!p->d_cand || !p->d_out) { p->cand_cap = 0; return RFX_E_NOMEM; }
  }
  const bool pinned_ok = (size_t)nb * 4 <= ((size_t)1 << 20) && n_out * 4 <= ((size_t)1 << 22);
  hipError_t e;
  if (pinned_ok) {
    memcpy(p->h_cand, cand, (size_t)nb * 4);
    e = hipMemcpyAsync(p->d_cand, p->h_cand, (size_t)nb * 4, hipMemcpyHostToDevice, c->stream);
  } else {
    e = hipMemcpyAsync(p->d_cand, cand, (size_t)nb * 4, hipMemcpyHostToDevice, c->stream);
  }
  const char* d_a = nullptr;
  if (e == hipSuccess && a_explicit) {
    if ((size_t)alen + 1 > p->a_cap) {
      dfree(c, p->d_a);
      p->a_cap = (size_t)alen * 2 + 1024;
      p->d_a = (char*)dmalloc(c, p->a_cap);
      if (!p->d_a) { p->a_cap = 0; return RFX_E_NOMEM; }
    }
    e = upload(c, p->d_a, a_explicit, (size_t)alen);
    d_a = p->d_a;
  }
  if (e != hipSuccess) return hip_fail(e, "rfx_ovl_pool_score");
  rfxk::overlap_pool(c, p->arena, p->d_off, p->d_len, d_a, alen, query, p->d_cand, nb, strands == 1 ? 1 : 0,
                     strands == 0 ? 0 : 1, lds, min_pct, min_ovl, variant == RFX_OVL_CONTIG,
                     variant == RFX_OVL_CONTIG ? -1 : 0, p->d_out);
  e = hipMemcpyAsync(pinned_ok ? p->h_out : out, p->d_out, n_out * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = ctx_sync(c);
  if (e != hipSuccess) return hip_fail(e, "rfx_ovl_pool_score");
  if (pinned_ok) memcpy(out, p->h_out, n_out * 4);
  return RFX_OK;
}

int rfx_annotate(rfx_set* s, const rfx_reads* r, uint32_t* cov_out) {
  if (!s || !r || s->ctx != r->ctx || !r->good || !cov_out) return RFX_E_INVAL;
  rfx_ctx* c = s->ctx;
  (void)hipSetDevice(c->device);
  if (r->n == 0 || r->n_bases == 0) return RFX_OK;
  std::vector<uint32_t> len(r->n, r->ulen);
  hipError_t e = hipSuccess;
  if (!r->ulen) {
    e = hipMemcpyAsync(len.data(), r->len, (size_t)r->n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = ctx_sync(c);
  }
  std::vector<uint64_t> off((size_t)r->n + 1, 0);
  for (uint32_t i = 0; i < r->n; ++i) off[(size_t)i + 1] = off[i] + len[i];
  uint64_t* d_off = (uint64_t*)dmalloc(c, off.size() * 8);
  uint32_t* d_cov = (uint32_t*)dmalloc(c, r->n_bases * 4);
  if (!d_off || !d_cov) { dfree(c, d_off); dfree(c, d_cov); return RFX_E_NOMEM; }
  if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(d_cov, 0, r->n_bases * 4, c->stream);
  if (e == hipSuccess) {
    const rfx_reads_view rv = r->view();
    rfxk::annotate(c, rv, s->slots, s->bits, s->has_all_ones, s->k, d_off, d_cov);
    e = hipMemcpyAsync(cov_out, d_cov, r->n_bases * 4, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = ctx_sync(c);
  dfree(c, d_off);
  dfree(c, d_cov);
  return e == hipSuccess ? RFX_OK : hip_fail(e, "rfx_annotate");
}

}  // extern "C"
