// Host-only harness for the drop-in `jellyfish` (rufus_amd/csrc/host/jellyfish_main.cpp: count / histo / merge / query /
// --sam / --spool / RUFUS_GPUS plumbing, the parallel ingest, the .Jhash reader and writer): the tool's own main() with
// the DEVICE entry points of the C-ABI replaced by plain host code (a std::map count, sorted vectors), so that argument
// handling, threading and file formats run in the CPU suite and under the sanitizers.  TEST INFRASTRUCTURE: nothing
// here is built into the product, whose jellyfish has no CPU path; results are compared with the oracle and with
// jellyfish's own md5 known answers by tests/test_jellyfish_host.py.  The host half of the library (matrix, header,
// packers) is the real one:   g++ -O2 -std=c++17 -pthread jellyfish_harness.cpp ../../rufus_amd/csrc/rfx_host.cpp
#include <algorithm>
#include <map>
#include <numeric>
#include <condition_variable>
#include <mutex>
#include <unordered_map>

#include "../../rufus_amd/csrc/host/jellyfish_main.cpp"

struct rfx_ctx { int device; };
struct rfx_reads {
  std::vector<uint64_t> codes;
  std::vector<uint32_t> acgt, woff, len;
};
// (round 4: the tool deals the read blocks to the devices in turn; the tables of a group pool their counts at finish --
// RFX_PEERS_REPLICATE=1: every table is given every block, nothing to pool)
struct rfx_peers {
  int n;
  bool replicate = getenv("RFX_PEERS_REPLICATE") != nullptr;
  std::mutex mu;
  std::condition_variable cv;
  int merged = 0;
  std::unordered_map<uint64_t, uint64_t> total;
};
struct rfx_table {
  int k, canonical, lsize;
  uint64_t pos_lo, pos_hi;
  std::vector<uint64_t> cols;
  std::unordered_map<uint64_t, uint64_t> counts;
  int peer_index = 0, peer_n = 1;
  rfx_peers* peers = nullptr;
};
struct rfx_records {
  int k, lsize;
  std::vector<uint64_t> cols, keys, pos;
  std::vector<uint32_t> counts;
};

static thread_local std::string g_stand_in_err = "host stand-in";

static rfx_records* make_records(int k, int lsize, const uint64_t* cols, std::vector<std::pair<uint64_t, uint32_t>>& kv) {
  rfx_records* r = new rfx_records;
  r->k = k;
  r->lsize = lsize;
  r->cols.assign(cols, cols + 2 * k);
  std::vector<std::pair<uint64_t, size_t>> order(kv.size());
  for (size_t i = 0; i < kv.size(); ++i) order[i] = {rfx_jf_pos(cols, k, lsize, kv[i].first), i};
  std::sort(order.begin(), order.end(), [&](const auto& a, const auto& b) {
    return a.first != b.first ? a.first < b.first : kv[a.second].first < kv[b.second].first;
  });
  for (const auto& o : order) {
    r->pos.push_back(o.first);
    r->keys.push_back(kv[o.second].first);
    r->counts.push_back(kv[o.second].second);
  }
  return r;
}

extern "C" {

const char* rfx_last_error(void) { return g_stand_in_err.c_str(); }
rfx_ctx* rfx_open(int device, size_t) { return new rfx_ctx{device}; }
void rfx_close(rfx_ctx* c) { delete c; }
int rfx_ctx_allow_peers(rfx_ctx*, const int*, int) { return RFX_OK; }
void* rfx_host_alloc(size_t bytes) { return malloc(bytes); }
void* rfx_host_alloc_lazy(size_t bytes) { return malloc(bytes); }
int rfx_host_pin(void*) { return RFX_OK; }
void rfx_host_free(void* p) { free(p); }

rfx_reads* rfx_reads_upload(rfx_ctx*, const uint64_t* codes, const uint32_t* acgt, const uint32_t*, const uint32_t* word_off,
                            const uint32_t* len, uint32_t n_reads) {
  if (!acgt) return nullptr;
  rfx_reads* r = new rfx_reads;
  const uint32_t words = word_off[n_reads];
  r->codes.assign(codes, codes + words);
  r->acgt.assign(acgt, acgt + words);
  r->woff.assign(word_off, word_off + n_reads + 1);
  r->len.assign(len, len + n_reads);
  return r;
}
void rfx_reads_free(rfx_reads* r) { delete r; }
uint32_t rfx_reads_count(const rfx_reads* r) { return r ? (uint32_t)r->len.size() : 0; }

// ---- rfx_text_* (round 6: the text route of the count, host/rfx_ingest.hpp TextIngest): a host stand-in that refuses
// what the device refuses (anything but strict 4-line FASTQ with every line newline-terminated) and packs the rest with
// the library's own host packer.  The tool's threads (copiers, the parsing thread) run against it under the sanitizers.
}  // extern "C"
struct rfx_text {
  std::string buf;
  uint64_t cap = 0;
  long appends = 0;
  std::mutex mu;
};
extern "C" {
rfx_text* rfx_text_open(rfx_ctx*, uint64_t cap_bytes) {
  rfx_text* t = new rfx_text;
  t->cap = cap_bytes;
  return t;
}
void rfx_text_close(rfx_text* t) { delete t; }
uint64_t rfx_text_room(const rfx_text* t) { return t->cap - t->buf.size(); }
uint64_t rfx_text_bytes(const rfx_text* t) { return t->buf.size(); }
long rfx_text_append(rfx_text* t, const void* host, uint64_t n) {
  std::lock_guard<std::mutex> g(t->mu);
  if (n > t->cap - t->buf.size()) return RFX_E_RANGE;
  t->buf.append((const char*)host, (size_t)n);
  return t->appends++;
}
int rfx_text_copied(rfx_text* t, long ticket) {
  std::lock_guard<std::mutex> g(t->mu);
  return ticket >= 0 && ticket < t->appends ? 1 : RFX_E_INVAL;
}
int rfx_text_wait(rfx_text* t, long ticket) { return rfx_text_copied(t, ticket) == 1 ? RFX_OK : RFX_E_INVAL; }
int rfx_text_fetch(rfx_text* t, void* host) {
  memcpy(host, t->buf.data(), t->buf.size());
  return RFX_OK;
}
void rfx_text_reset(rfx_text* t) {
  std::lock_guard<std::mutex> g(t->mu);
  t->buf.clear();
  t->appends = 0;
}
rfx_reads* rfx_text_parse(rfx_text* t, int flags, int, int* strict) {
  *strict = 1;
  if (flags != RFX_PACK_COUNT) return nullptr;
  const std::string& b = t->buf;
  std::vector<uint64_t> start;
  std::vector<uint32_t> slen;
  size_t p = 0;
  auto line = [&](size_t& lo, size_t& hi) {  // [lo, hi) without the newline; false: no newline left
    const size_t nl = b.find('\n', p);
    if (nl == std::string::npos) return false;
    lo = p;
    hi = nl;
    p = nl + 1;
    return true;
  };
  while (p < b.size()) {
    size_t h0, h1, s0, s1, p0, p1, q0, q1;
    if (!line(h0, h1) || !line(s0, s1) || !line(p0, p1) || !line(q0, q1) || h1 == h0 || b[h0] != '@' || p1 == p0 || b[p0] != '+' ||
        s1 - s0 != q1 - q0) {
      *strict = 0;
      return nullptr;
    }
    start.push_back(s0);
    slen.push_back((uint32_t)(s1 - s0));
  }
  if (start.empty()) {
    *strict = 0;
    return nullptr;
  }
  rfx_reads* r = new rfx_reads;
  uint64_t words = 0;
  for (uint32_t l : slen) words += (l + 31) / 32;
  r->codes.assign(words ? words : 1, 0);
  r->acgt.assign(words ? words : 1, 0);
  r->woff.assign(start.size() + 1, 0);
  r->len.assign(start.size(), 0);
  if (rfx_pack_spans(b.data(), start.data(), slen.data(), nullptr, (uint32_t)start.size(), 0, RFX_PACK_COUNT, r->codes.data(),
                     r->acgt.data(), nullptr, r->woff.data(), r->len.data()) != RFX_OK) {
    delete r;
    return nullptr;
  }
  return r;
}

rfx_peers* rfx_peers_create(int n) {
  rfx_peers* p = new rfx_peers;
  p->n = n;
  return p;
}
void rfx_peers_free(rfx_peers* p) { delete p; }

rfx_table* rfx_count_begin(rfx_ctx*, int k, int canonical, int lsize, uint64_t, uint64_t pos_lo, uint64_t pos_hi) {
  if (k < 1 || k > 32 || (k == 32 && !canonical) || lsize > 2 * k) return nullptr;
  rfx_table* t = new rfx_table;
  t->k = k;
  t->canonical = canonical;
  t->lsize = lsize;
  t->pos_lo = pos_lo;
  t->pos_hi = pos_hi;
  t->cols.resize((size_t)2 * k);
  if (rfx_jf_matrix(lsize, k, t->cols.data()) != RFX_OK) {
    delete t;
    return nullptr;
  }
  return t;
}
int rfx_count_set_passes(rfx_table*, int) { return RFX_OK; }
int rfx_mem_stats(rfx_ctx*, uint64_t* used, uint64_t* peak, uint64_t* mapped) {
  if (used) *used = 0;
  if (peak) *peak = 0;
  if (mapped) *mapped = 0;
  return RFX_OK;
}
int rfx_mem_reserve(rfx_ctx*, uint64_t) { return RFX_OK; }
int rfx_count_set_peers(rfx_table* t, rfx_peers* p, int index) {
  t->peer_index = index;
  t->peer_n = p->n;
  t->peers = p;
  return RFX_OK;
}
// every k-mer of every read: jf/include/jellyfish/mer_iterator.hpp:59-88 (a base that is not ACGT restarts the window)
int rfx_count_add(rfx_table* t, const rfx_reads* r) {
  const int k = t->k;
  const uint64_t kmask = k >= 32 ? ~0ull : (1ull << (2 * k)) - 1;
  for (size_t x = 0; x < r->len.size(); ++x) {
    const uint32_t L = r->len[x], w0 = r->woff[x];
    uint64_t fwd = 0, rc = 0;
    int run = 0;
    for (uint32_t i = 0; i < L; ++i) {
      const uint64_t code = (r->codes[w0 + i / 32] >> (2 * (i % 32))) & 3u;
      if (!((r->acgt[w0 + i / 32] >> (i % 32)) & 1u)) {
        run = 0;
        continue;
      }
      fwd = ((fwd << 2) | code) & kmask;
      rc = (rc >> 2) | ((3 - code) << (2 * (k - 1)));
      if (++run >= k) ++t->counts[t->canonical ? std::min(fwd, rc) : fwd];
    }
  }
  return RFX_OK;
}
void rfx_count_free(rfx_table* t) { delete t; }

rfx_records* rfx_count_finish(rfx_table* t, uint64_t lower, uint64_t upper, uint64_t* histo) {
  if (t->peers && !t->peers->replicate) {  // the group's tables hold disjoint read blocks: pool the counts, then all go on
    rfx_peers* p = t->peers;
    std::unique_lock<std::mutex> g(p->mu);
    for (const auto& e : t->counts) p->total[e.first] += e.second;
    if (++p->merged == p->n) p->cv.notify_all();
    p->cv.wait(g, [&] { return p->merged >= p->n; });
    t->counts = p->total;
  }
  std::vector<std::pair<uint64_t, uint32_t>> kv;
  const uint64_t hi = t->pos_hi ? t->pos_hi : (t->lsize >= 64 ? ~0ull : 1ull << t->lsize);
  for (const auto& e : t->counts) {
    if (e.second < lower || e.second > upper) continue;
    const uint64_t pos = rfx_jf_pos(t->cols.data(), t->k, t->lsize, e.first);
    if (pos < t->pos_lo || (t->pos_hi && pos >= hi)) continue;
    kv.push_back({e.first, e.second > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)e.second});
  }
  rfx_records* all = make_records(t->k, t->lsize, t->cols.data(), kv);
  if (t->peer_n > 1) {  // table i of n hands back slice i of the output positions
    const size_t n = all->keys.size(), a = n * (size_t)t->peer_index / (size_t)t->peer_n,
                 b = n * (size_t)(t->peer_index + 1) / (size_t)t->peer_n;
    rfx_records* s = new rfx_records;
    s->k = all->k;
    s->lsize = all->lsize;
    s->cols = all->cols;
    s->keys.assign(all->keys.begin() + a, all->keys.begin() + b);
    s->pos.assign(all->pos.begin() + a, all->pos.begin() + b);
    s->counts.assign(all->counts.begin() + a, all->counts.begin() + b);
    delete all;
    all = s;
  }
  if (histo) {
    for (int i = 0; i < RFX_HISTO_BINS; ++i) histo[i] = 0;
    for (uint32_t c : all->counts) ++histo[c > 10001u ? 10001u : c];
  }
  return all;
}

uint64_t rfx_records_size(const rfx_records* r) { return r->keys.size(); }
int rfx_records_k(const rfx_records* r) { return r->k; }
int rfx_records_lsize(const rfx_records* r) { return r->lsize; }
void rfx_records_free(rfx_records* r) { delete r; }
int rfx_records_get(const rfx_records* r, uint64_t* keys, uint32_t* counts, uint64_t* pos) {
  if (keys) std::copy(r->keys.begin(), r->keys.end(), keys);
  if (counts) std::copy(r->counts.begin(), r->counts.end(), counts);
  if (pos) std::copy(r->pos.begin(), r->pos.end(), pos);
  return RFX_OK;
}
int rfx_records_histo(const rfx_records* r, uint64_t* histo) {
  for (int i = 0; i < RFX_HISTO_BINS; ++i) histo[i] = 0;
  for (uint32_t c : r->counts) ++histo[c > 10001u ? 10001u : c];
  return RFX_OK;
}
// jf/include/jellyfish/binary_dumper.hpp:44-48: ceil(2k/8) key bytes, then counter_len count bytes (saturating), little endian
int rfx_records_payload_range(const rfx_records* r, uint64_t first, uint64_t n, void* out, size_t cap, int counter_len) {
  const int kb = (2 * r->k + 7) / 8;
  const size_t rl = (size_t)kb + (size_t)counter_len;
  if (first + n > r->keys.size() || cap < n * rl) return RFX_E_INVAL;
  unsigned char* o = (unsigned char*)out;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t key = r->keys[first + i];
    uint64_t c = r->counts[first + i];
    if (counter_len < 4) c = std::min<uint64_t>(c, (1ull << (8 * counter_len)) - 1);
    for (int b = 0; b < kb; ++b) o[i * rl + b] = (unsigned char)(key >> (8 * b));
    for (int b = 0; b < counter_len; ++b) o[i * rl + kb + b] = b < 8 ? (unsigned char)(c >> (8 * b)) : 0;
  }
  return RFX_OK;
}
rfx_records* rfx_records_load(rfx_ctx*, int k, int lsize, const uint64_t* cols, const void* payload, uint64_t n, int counter_len) {
  const int kb = (2 * k + 7) / 8;
  const size_t rl = (size_t)kb + (size_t)counter_len;
  const unsigned char* p = (const unsigned char*)payload;
  rfx_records* r = new rfx_records;
  r->k = k;
  r->lsize = lsize;
  r->cols.assign(cols, cols + 2 * k);
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t key = 0, c = 0;
    for (int b = 0; b < kb; ++b) key |= (uint64_t)p[i * rl + b] << (8 * b);
    for (int b = 0; b < counter_len && b < 8; ++b) c |= (uint64_t)p[i * rl + kb + b] << (8 * b);
    const uint64_t pos = rfx_jf_pos(cols, k, lsize, key);
    if (i && (pos < r->pos.back() || (pos == r->pos.back() && key <= r->keys.back()))) {
      g_stand_in_err = "records are not in (pos,key) order";
      delete r;
      return nullptr;
    }
    r->keys.push_back(key);
    r->pos.push_back(pos);
    r->counts.push_back(c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c);
  }
  return r;
}
rfx_records* rfx_records_load_fd(rfx_ctx* c, int k, int lsize, const uint64_t* cols, int fd, uint64_t offset, uint64_t n,
                                 int counter_len) {
  const size_t rl = (size_t)(2 * k + 7) / 8 + (size_t)counter_len;
  std::vector<char> buf(n * rl);
  size_t got = 0;
  while (got < buf.size()) {
    const ssize_t m = ::pread(fd, buf.data() + got, buf.size() - got, (off_t)(offset + got));
    if (m <= 0) {
      g_stand_in_err = "short read";
      return nullptr;
    }
    got += (size_t)m;
  }
  return rfx_records_load(c, k, lsize, cols, buf.data(), n, counter_len);
}

// jf/jellyfish/merge_files.cc:69-155 as RUFUS modified it: keys held by exactly one input, count >= min_count there
int rfx_merge_unique(rfx_ctx*, const rfx_records* const* files, int n_files, uint32_t min_count, uint64_t* keys_out,
                     uint32_t* counts_out, uint64_t cap, uint64_t* n_out) {
  std::map<std::pair<uint64_t, uint64_t>, std::pair<int, uint32_t>> seen;  // (pos, key) -> (inputs holding it, count)
  for (int f = 0; f < n_files; ++f)
    for (size_t i = 0; i < files[f]->keys.size(); ++i) {
      auto& e = seen[{files[f]->pos[i], files[f]->keys[i]}];
      ++e.first;
      e.second = files[f]->counts[i];
    }
  uint64_t n = 0;
  for (const auto& e : seen)
    if (e.second.first == 1 && e.second.second >= min_count) {
      if (n < cap) {
        if (keys_out) keys_out[n] = e.first.second;
        if (counts_out) counts_out[n] = e.second.second;
      }
      ++n;
    }
  if (n_out) *n_out = n;
  return n > cap ? RFX_E_FULL : RFX_OK;
}
int rfx_query(const rfx_records* db, const uint64_t* keys, uint64_t n, uint32_t* counts_out) {
  std::unordered_map<uint64_t, uint32_t> m;
  for (size_t i = 0; i < db->keys.size(); ++i) m[db->keys[i]] = db->counts[i];
  for (uint64_t i = 0; i < n; ++i) {
    const auto it = m.find(keys[i]);
    counts_out[i] = it == m.end() ? 0u : it->second;
  }
  return RFX_OK;
}

}  // extern "C"
