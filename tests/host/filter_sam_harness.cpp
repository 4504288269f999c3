// Host-only harness for `RUFUS.Filter --sam` (rufus_amd/csrc/host/rufus_filter_main.cpp, namespace samf): the tool's
// own main() with the DEVICE entry points of the C-ABI replaced by host stand-ins, so that the threading around the
// scan -- reader, helper threads, pieces handed over and recycled, the pairing of records by QNAME across pieces,
// waiting records copied out of retiring pieces -- runs in the CPU suite.  TEST INFRASTRUCTURE: the stand-in for
// rfx_filter is a plain loop over the packed block (src/RUFUS.Filter.cpp:203-220 on the arrays rfx_pack_spans made);
// nothing here is built into the product, whose RUFUS.Filter has no CPU path.  The host half of the library
// (rfx_hashlist_keys, rfx_pack_spans, rfx_host_cpus) is the real one: build with
//   g++ -O2 -std=c++17 -pthread filter_sam_harness.cpp ../../rufus_amd/csrc/rfx_host.cpp
#include <cstdlib>
#include <unordered_set>
#include <vector>

#include "../../rufus_amd/csrc/host/rufus_filter_main.cpp"

struct rfx_ctx { int device; };
struct rfx_set {
  std::unordered_set<uint64_t> keys;
  int k;
};
struct rfx_reads {
  std::vector<uint64_t> codes;
  std::vector<uint32_t> good, woff, len;
};

extern "C" {

const char* rfx_last_error(void) { return "host stand-in"; }
rfx_ctx* rfx_open(int device, size_t) { return new rfx_ctx{device}; }
void rfx_close(rfx_ctx* c) { delete c; }
int rfx_ctx_allow_peers(rfx_ctx*, const int*, int) { return RFX_OK; }
void* rfx_host_alloc(size_t bytes) { return malloc(bytes); }
void* rfx_host_alloc_lazy(size_t bytes) { return malloc(bytes); }
int rfx_host_pin(void*) { return RFX_OK; }
void rfx_host_free(void* p) { free(p); }

rfx_set* rfx_set_build(rfx_ctx*, const uint64_t* fwd_keys, uint64_t n, int k) {
  rfx_set* s = new rfx_set;
  s->k = k;
  s->keys.insert(fwd_keys, fwd_keys + n);
  return s;
}
void rfx_set_free(rfx_set* s) { delete s; }

rfx_reads* rfx_reads_upload(rfx_ctx*, const uint64_t* codes, const uint32_t*, const uint32_t* good, const uint32_t* word_off,
                            const uint32_t* len, uint32_t n_reads) {
  rfx_reads* r = new rfx_reads;
  const uint32_t words = word_off[n_reads];
  r->codes.assign(codes, codes + words);
  r->good.assign(good, good + words);
  r->woff.assign(word_off, word_off + n_reads + 1);
  r->len.assign(len, len + n_reads);
  return r;
}
void rfx_reads_free(rfx_reads* r) { delete r; }

// hits of a read = positions i < L - last_base_skipped that end a streak of >= k good bases and whose window (first
// base most significant, 2 bits per base) is in the set
int rfx_filter(rfx_set* s, const rfx_reads* r, int thresh, int last_base_skipped, uint32_t* hits_out, uint64_t* hitmask_out,
               uint64_t* n_hit_reads) {
  const int k = s->k;
  const uint64_t kmask = k >= 32 ? ~0ull : (1ull << (2 * k)) - 1;
  const size_t n = r->len.size();
  uint64_t over = 0;
  if (hitmask_out)
    for (size_t w = 0; w < (n + 63) / 64; ++w) hitmask_out[w] = 0;
  for (size_t x = 0; x < n; ++x) {
    const uint32_t L = r->len[x], w0 = r->woff[x];
    const uint32_t stop = last_base_skipped ? (L ? L - 1 : 0) : L;
    uint64_t key = 0;
    int streak = 0;
    uint32_t found = 0;
    for (uint32_t i = 0; i < stop; ++i) {
      const uint64_t code = (r->codes[w0 + i / 32] >> (2 * (i % 32))) & 3u;
      const bool good = (r->good[w0 + i / 32] >> (i % 32)) & 1u;
      key = ((key << 2) | code) & kmask;
      streak = good ? streak + 1 : 0;
      if (streak >= k && s->keys.count(key)) ++found;
    }
    if (hits_out) hits_out[x] = found;
    if (found >= (uint32_t)thresh) {
      ++over;
      if (hitmask_out) hitmask_out[x >> 6] |= 1ull << (x & 63);
    }
  }
  if (n_hit_reads) *n_hit_reads = over;
  return RFX_OK;
}

}  // extern "C"
