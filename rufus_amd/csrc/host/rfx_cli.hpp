// Shared host-side plumbing of the drop-in executables (text parsing, .Jhash container, k-mer text).
// All arithmetic of the hot path happens behind the C-ABI (include/rufus_hip.h); nothing here counts,
// hashes or compares k-mers.
#pragma once
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <csignal>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/rufus_hip.h"

namespace rfxcli {

[[noreturn]] inline void die(const std::string& msg) {
  // jellyfish's err::die: message on stderr, non-zero exit (jf/include/jellyfish/err.hpp)
  fprintf(stderr, "%s\n", msg.c_str());
  exit(1);
}

inline rfx_ctx* open_ctx() {
  const char* dev = getenv("RUFUS_GPU");
  rfx_ctx* c = rfx_open(dev ? atoi(dev) : 0, 0);
  if (!c) die(std::string("rufus_amd: no MI355X (gfx950) device: ") + rfx_last_error() + " -- there is no CPU fallback");
  return c;
}

// RUFUS_GPUS="0-7" | "0,2,5" | "0-3,6" (a device may be named twice: two contexts on it, how a one-GPU box tests the
// multi-device code): the devices `jellyfish count` and `RUFUS.Filter` spread one sample over.  Unset: the one device
// of RUFUS_GPU (default 0).
inline std::vector<int> gpu_list() {
  std::vector<int> out;
  const char* ev = getenv("RUFUS_GPUS");
  if (!ev || !*ev) {
    const char* dev = getenv("RUFUS_GPU");
    out.push_back(dev ? atoi(dev) : 0);
    return out;
  }
  const char* p = ev;
  while (*p) {
    char* end;
    const long a = strtol(p, &end, 10);
    if (end == p || a < 0) die(std::string("rufus_amd: cannot parse RUFUS_GPUS='") + ev + "'");
    long b = a;
    p = end;
    if (*p == '-') {
      b = strtol(p + 1, &end, 10);
      if (end == p + 1 || b < a) die(std::string("rufus_amd: cannot parse RUFUS_GPUS='") + ev + "'");
      p = end;
    }
    for (long d = a; d <= b; ++d) out.push_back((int)d);
    if (*p == ',') ++p;
    else if (*p) die(std::string("rufus_amd: cannot parse RUFUS_GPUS='") + ev + "'");
  }
  if (out.empty() || out.size() > 64) die(std::string("rufus_amd: RUFUS_GPUS names no device (or more than 64): '") + ev + "'");
  return out;
}
inline std::vector<rfx_ctx*> open_ctxs(const std::vector<int>& gpus) {
  std::vector<rfx_ctx*> out;
  for (int d : gpus) {
    rfx_ctx* c = rfx_open(d, 0);
    if (!c) die(std::string("rufus_amd: no MI355X (gfx950) device ") + std::to_string(d) + ": " + rfx_last_error() +
                " -- there is no CPU fallback");
    if (gpus.size() > 1 && rfx_ctx_allow_peers(c, gpus.data(), (int)gpus.size()) != RFX_OK)
      die(std::string("rufus_amd: ") + rfx_last_error());
    out.push_back(c);
  }
  return out;
}

// Buffered line reader over a file descriptor; works on regular files and on named pipes.
class LineReader {
  int fd_ = -1;
  std::vector<char> buf_;
  size_t beg_ = 0, end_ = 0;
  bool eof_ = false;

  bool fill() {
    if (eof_) return false;
    if (beg_ > 0) {
      memmove(buf_.data(), buf_.data() + beg_, end_ - beg_);
      end_ -= beg_;
      beg_ = 0;
    }
    if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
    ssize_t n;
    do n = ::read(fd_, buf_.data() + end_, buf_.size() - end_);
    while (n < 0 && errno == EINTR);  // a signal is not the end of a pipe
    if (n < 0) die(std::string("read error on input: ") + strerror(errno));
    if (n == 0) {
      eof_ = true;
      return false;
    }
    end_ += (size_t)n;
    return true;
  }

 public:
  explicit LineReader(size_t cap = 1 << 22) : buf_(cap) {}
  bool open(const char* path) {
    fd_ = strcmp(path, "stdin") == 0 || strcmp(path, "/dev/stdin") == 0 ? 0 : ::open(path, O_RDONLY);
    return fd_ >= 0;
  }
  void attach(int fd) { fd_ = fd; }
  // no descriptor: what was preloaded is all there is (text handed over in memory)
  void close_input() { eof_ = true; }
  // Bytes that were already read from the descriptor (format sniffing) go first.
  void preload(const char* p, size_t n) {
    if (end_ + n > buf_.size()) buf_.resize(end_ + n + (1 << 20));
    memcpy(buf_.data() + end_, p, n);
    end_ += n;
  }
  ~LineReader() {
    if (fd_ > 0) ::close(fd_);
  }
  // Next line without its '\n' (std::getline semantics: a final unterminated line is returned,
  // an empty file returns false).  The view is valid until the next call.
  bool getline(const char*& b, const char*& e) {
    for (;;) {
      char* nl = (char*)memchr(buf_.data() + beg_, '\n', end_ - beg_);
      if (nl) {
        b = buf_.data() + beg_;
        e = nl;
        beg_ = (size_t)(nl - buf_.data()) + 1;
        return true;
      }
      if (!fill()) {
        if (beg_ == end_) return false;
        b = buf_.data() + beg_;
        e = buf_.data() + end_;
        beg_ = end_;
        return true;
      }
    }
  }
  int peek() {  // next byte or -1
    if (beg_ == end_ && !fill()) return -1;
    return (unsigned char)buf_[beg_];
  }
};

// A batch of reads as the packer wants them.
struct ReadBatch {
  std::string seq, qual;
  std::vector<uint64_t> off{0};
  void clear() {
    seq.clear();
    qual.clear();
    off.assign(1, 0);
  }
  uint32_t n() const { return (uint32_t)(off.size() - 1); }
  void add(const char* s, size_t ls, const char* q = nullptr, size_t lq = 0, bool want_qual = false) {
    seq.append(s, ls);
    if (want_qual) {  // a quality string shorter than its read reads as '\0' (bad) past its end
      const size_t m = lq < ls ? lq : ls;
      if (q) qual.append(q, m);
      qual.append(ls - m, '\0');
    }
    off.push_back(seq.size());
  }
};

struct PackedBatch {
  std::vector<uint64_t> codes;
  std::vector<uint32_t> acgt, good, word_off, len;
  int pack(const ReadBatch& b, int flags, int min_q) {
    const uint32_t n = b.n();
    const uint64_t nw = rfx_pack_words(b.off.data(), n);
    codes.assign(nw + 1, 0);
    if (flags & RFX_PACK_COUNT) acgt.assign(nw + 1, 0);
    if (flags & RFX_PACK_FILTER) good.assign(nw + 1, 0);
    word_off.assign((size_t)n + 1, 0);
    len.assign((size_t)n + 1, 0);
    return rfx_pack_reads(b.seq.data(), (flags & RFX_PACK_FILTER) ? b.qual.data() : nullptr, b.off.data(), n, min_q, flags,
                          codes.data(), (flags & RFX_PACK_COUNT) ? acgt.data() : nullptr,
                          (flags & RFX_PACK_FILTER) ? good.data() : nullptr, word_off.data(), len.data());
  }
  rfx_reads* upload(rfx_ctx* c, uint32_t n, int flags) {
    return rfx_reads_upload(c, codes.data(), (flags & RFX_PACK_COUNT) ? acgt.data() : nullptr,
                            (flags & RFX_PACK_FILTER) ? good.data() : nullptr, word_off.data(), len.data(), n);
  }
};

// Sequences of a FASTA/FASTQ stream the way jellyfish's parser sees them
// (jf/include/jellyfish/mer_overlap_sequence_parser.hpp:124-251): type sniffed from the first byte,
// multi-line records joined, as many quality characters skipped as there were sequence characters.
// cb(seq, len) is called once per read.  Returns false on an unsupported / malformed file.
template <typename F>
bool parse_sequences(LineReader& in, F&& cb) {
  const int first = in.peek();
  if (first < 0) return true;  // empty file
  const char *b, *e;
  std::string seq;
  if (first == '>') {
    bool have = false;
    while (in.getline(b, e)) {
      if (b < e && *b == '>') {
        if (have) cb(seq.data(), seq.size());
        seq.clear();
        have = true;
      } else {
        seq.append(b, e);
      }
    }
    if (have) cb(seq.data(), seq.size());
    return true;
  }
  if (first != '@') return false;
  while (in.getline(b, e)) {
    if (b == e) continue;
    if (*b != '@') return false;
    seq.clear();
    bool plus = false;
    while (in.getline(b, e)) {
      if (b < e && *b == '+') {
        plus = true;
        break;
      }
      seq.append(b, e);
    }
    size_t quals = 0;
    while (plus && quals < seq.size() && in.getline(b, e)) quals += (size_t)(e - b);
    if (quals != seq.size()) return false;
    cb(seq.data(), seq.size());
  }
  return true;
}

// ---- k-mer text <-> key (jf/include/jellyfish/mer_dna.hpp: first base most significant) -----------
inline std::string key_to_text(uint64_t key, int k) {
  std::string s((size_t)k, 'A');
  for (int i = 0; i < k; ++i) s[(size_t)i] = "ACGT"[(key >> (2 * (k - 1 - i))) & 3];
  return s;
}
inline bool text_to_key(const char* s, size_t n, uint64_t& key) {
  key = 0;
  for (size_t i = 0; i < n; ++i) {
    uint64_t c;
    switch (s[i]) {
      case 'A': case 'a': c = 0; break;
      case 'C': case 'c': c = 1; break;
      case 'G': case 'g': c = 2; break;
      case 'T': case 't': c = 3; break;
      default: return false;
    }
    key = (key << 2) | c;
  }
  return true;
}
inline uint64_t revcomp_key(uint64_t key, int k) {
  uint64_t r = 0;
  for (int i = 0; i < k; ++i) {
    r = (r << 2) | (3 - (key & 3));
    key >>= 2;
  }
  return r;
}

// ---- .Jhash container ----------------------------------------------------------------------------
struct JhashHeader {
  int k = 0, lsize = 0, counter_len = 4;
  bool canonical = false;
  std::string format;
  std::vector<uint64_t> cols;
  size_t payload_offset = 0;
  uint64_t file_size = 0;  // 0: not a regular file
};

// Just enough JSON for the terse header jellyfish writes (Json::FastWriter) and the one we write.
inline bool json_find(const std::string& js, const std::string& key, size_t& pos) {
  const std::string pat = "\"" + key + "\"";
  size_t p = 0;
  while ((p = js.find(pat, p)) != std::string::npos) {
    size_t q = p + pat.size();
    while (q < js.size() && (js[q] == ' ' || js[q] == '\t' || js[q] == '\n')) ++q;
    if (q < js.size() && js[q] == ':') {
      pos = q + 1;
      while (pos < js.size() && (js[pos] == ' ' || js[pos] == '\n')) ++pos;
      return true;
    }
    p = q;
  }
  return false;
}
inline bool json_u64(const std::string& js, const std::string& key, uint64_t& v) {
  size_t p;
  if (!json_find(js, key, p)) return false;
  char* end;
  v = strtoull(js.c_str() + p, &end, 10);
  return end != js.c_str() + p;
}

// payload == nullptr: the header only (h.payload_offset, h.file_size say where the records are)
inline bool read_jhash(const char* path, JhashHeader& h, std::vector<char>* payload) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char digits[10] = {0};
  if (fread(digits, 1, 9, f) != 9) { fclose(f); return false; }
  for (int i = 0; i < 9; ++i)
    if (digits[i] < '0' || digits[i] > '9') { fclose(f); return false; }
  const size_t hlen = (size_t)atol(digits);
  std::string js(hlen, '\0');
  if (hlen < 2 || fread(&js[0], 1, hlen, f) != hlen || js[0] != '{') { fclose(f); return false; }
  while (!js.empty() && js.back() == '\0') js.pop_back();
  uint64_t key_len = 0, size = 0, clen = 4;
  if (!json_u64(js, "key_len", key_len) || !json_u64(js, "size", size)) { fclose(f); return false; }
  json_u64(js, "counter_len", clen);
  h.k = (int)(key_len / 2);
  h.lsize = 0;
  while (h.lsize < 63 && (1ull << h.lsize) < size) ++h.lsize;  // (a size field beyond 2^63 must not shift by 64)
  h.counter_len = (int)clen;
  size_t p;
  h.canonical = json_find(js, "canonical", p) && js.compare(p, 4, "true") == 0;
  if (json_find(js, "format", p) && js[p] == '"') h.format = js.substr(p + 1, js.find('"', p + 1) - p - 1);
  h.cols.clear();
  size_t m;
  if (json_find(js, "matrix1", m)) {
    const std::string sub = js.substr(m);
    size_t cp;
    if (json_find(sub, "columns", cp) && sub[cp] == '[') {
      const char* s = sub.c_str() + cp + 1;
      while (*s && *s != ']') {
        char* end;
        const uint64_t v = strtoull(s, &end, 10);
        if (end == s) break;
        h.cols.push_back(v);
        s = end;
        while (*s == ',' || *s == ' ') ++s;
      }
    }
  }
  if ((int)h.cols.size() != 2 * h.k) { fclose(f); return false; }
  // (a record length of 0 would divide by zero further on; jellyfish writes counter_len 1 .. 8)
  if (key_len < 2 || key_len > 1024 || clen < 1 || clen > 8) { fclose(f); return false; }
  h.payload_offset = 9 + hlen;
  struct stat sb;
  h.file_size = fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode) ? (uint64_t)sb.st_size : 0;
  if (payload) {
    payload->clear();
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) payload->insert(payload->end(), buf, buf + n);
  }
  fclose(f);
  return true;
}

inline bool is_regular_file(const char* path) {
  struct stat sb;
  return ::stat(path, &sb) == 0 && S_ISREG(sb.st_mode);
}

inline rfx_records* load_records(rfx_ctx* c, const char* path, JhashHeader& h) {
  std::vector<char> payload;
  // a pipe / process substitution can be opened once: header and payload in ONE pass
  const bool regular = is_regular_file(path);
  if (!read_jhash(path, h, regular ? nullptr : &payload)) die(std::string("Failed to parse header of file '") + path + "'");
  if (h.format != "binary/sorted") die("Unknown format '" + h.format + "'");
  const size_t rl = (size_t)(2 * h.k + 7) / 8 + (size_t)h.counter_len;
  if (regular && h.file_size >= h.payload_offset) {  // a regular file: streamed from the descriptor
    const uint64_t bytes = h.file_size - h.payload_offset;
    if (bytes % rl != 0)
      die("Size of database (" + std::to_string(bytes) + ") must be a multiple of the length of a record (" +
          std::to_string(rl) + ")");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) die(std::string("Failed to open file '") + path + "'");
    rfx_records* r = rfx_records_load_fd(c, h.k, h.lsize, h.cols.data(), fd, h.payload_offset, bytes / rl, h.counter_len);
    ::close(fd);
    if (!r) die(std::string("rufus_amd: cannot load '") + path + "': " + rfx_last_error());
    return r;
  }
  if (regular && !read_jhash(path, h, &payload)) die(std::string("Failed to parse header of file '") + path + "'");
  if (payload.size() % rl != 0)
    die("Size of database (" + std::to_string(payload.size()) + ") must be a multiple of the length of a record (" +
        std::to_string(rl) + ")");
  rfx_records* r = rfx_records_load(c, h.k, h.lsize, h.cols.data(), payload.data(), payload.size() / rl, h.counter_len);
  if (!r) die(std::string("rufus_amd: cannot load '") + path + "': " + rfx_last_error());
  return r;
}

// A sorted database on disk, cut by ranges of the position (the high-order sort key) without loading it: merge and
// query walk it slice by slice, so their HBM footprint is a slice, not the 64 GB a 30x sample's records take
// (jf/jellyfish/merge_files.cc:69-155 and jf/include/jellyfish/binary_dumper.hpp:156-203 stream / search the file
// the same way on the host).
struct JhashFile {
  std::string path;
  JhashHeader h;
  int fd = -1;
  uint64_t n = 0;
  size_t rl = 0, kb = 0;

  bool open(const char* p) {  // false: not a regular file (the caller loads it whole instead -- it was not touched)
    path = p;
    if (!is_regular_file(p)) return false;
    if (!read_jhash(p, h, nullptr)) die(std::string("Failed to parse header of file '") + p + "'");
    if (h.format != "binary/sorted") die("Unknown format '" + h.format + "'");
    kb = (size_t)(2 * h.k + 7) / 8;
    rl = kb + (size_t)h.counter_len;
    if (h.file_size < h.payload_offset) return false;
    const uint64_t bytes = h.file_size - h.payload_offset;
    if (bytes % rl != 0)
      die("Size of database (" + std::to_string(bytes) + ") must be a multiple of the length of a record (" +
          std::to_string(rl) + ")");
    n = bytes / rl;
    fd = ::open(p, O_RDONLY);
    if (fd < 0) die(std::string("Failed to open file '") + p + "'");
    return true;
  }
  uint64_t pos_at(uint64_t i) const {
    unsigned char b[16] = {0};
    size_t got = 0;
    while (got < kb) {
      const ssize_t w = ::pread(fd, b + got, kb - got, (off_t)(h.payload_offset + i * rl + got));
      if (w < 0 && errno == EINTR) continue;
      if (w <= 0) die("read error on '" + path + "'");
      got += (size_t)w;
    }
    uint64_t key = 0;
    for (size_t j = 0; j < kb; ++j) key |= (uint64_t)b[j] << (8 * j);
    return rfx_jf_pos(h.cols.data(), h.k, h.lsize, key);
  }
  // index of the first record whose position is >= pos (n when there is none): ~32 single-record reads
  uint64_t lower_bound_pos(uint64_t pos) const {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
      const uint64_t mid = lo + (hi - lo) / 2;
      if (pos_at(mid) < pos) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  }
  rfx_records* load(rfx_ctx* c, uint64_t i0, uint64_t i1) const {
    rfx_records* r = rfx_records_load_fd(c, h.k, h.lsize, h.cols.data(), fd, h.payload_offset + i0 * rl, i1 - i0, h.counter_len);
    if (!r) die("rufus_amd: cannot load '" + path + "': " + rfx_last_error());
    return r;
  }
  void close() {
    if (fd >= 0) ::close(fd);
    fd = -1;
  }
};
// first position of slice s of S equal position ranges of a 2^lsize table (ceil, so that slice_of() below agrees)
inline uint64_t slice_start(uint64_t s, uint64_t S, int lsize) {
  const unsigned __int128 num = ((unsigned __int128)s << lsize) + (S - 1);
  return (uint64_t)(num / S);
}
inline uint64_t slice_of(uint64_t pos, uint64_t S, int lsize) { return (uint64_t)(((unsigned __int128)pos * S) >> lsize); }

// ---- newline scanning -----------------------------------------------------------------------------------------
// The text side of every tool is a search for '\n'; a memchr call per (short) FASTQ line costs ~20 ns, which at four
// lines per record is what one reader thread can do and no more.  These walk the buffer 16 / 32 bytes per step.
#if defined(__x86_64__)
#define RFX_X86 1
#endif

// memchr(p, '\n', e - p) for short lines, no call
static inline const char* find_nl(const char* p, const char* e) {
#if RFX_X86
  const __m128i nl = _mm_set1_epi8('\n');
  while (e - p >= 16) {
    const int m = _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)p), nl));
    if (m) return p + __builtin_ctz((unsigned)m);
    p += 16;
  }
#endif
  return p < e ? (const char*)memchr(p, '\n', (size_t)(e - p)) : nullptr;
}

// Walks [p, e) until `want` newlines have been seen: returns the position just after the last one seen that
// counts (e when the buffer ran out first), `got` = how many were seen.
typedef const char* (*skip_lines_fn)(const char* p, const char* e, size_t want, size_t& got);
static inline const char* skip_lines_plain(const char* p, const char* e, size_t want, size_t& got) {
  got = 0;
  while (got < want && p < e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    if (!nl) return e;
    ++got;
    p = nl + 1;
  }
  return got == want ? p : e;
}
#if RFX_X86
__attribute__((target("avx2,popcnt"))) static inline const char* skip_lines_avx2(const char* p, const char* e, size_t want,
                                                                               size_t& got) {
  got = 0;
  if (want == 0) return p;
  const __m256i nl = _mm256_set1_epi8('\n');
  while (e - p >= 32) {
    uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)p), nl));
    const size_t c = (size_t)__builtin_popcount(m);
    if (got + c >= want) {
      for (size_t drop = want - got - 1; drop; --drop) m &= m - 1;
      got = want;
      return p + __builtin_ctz(m) + 1;
    }
    got += c;
    p += 32;
  }
  size_t tail;
  const char* r = skip_lines_plain(p, e, want - got, tail);
  got += tail;
  return r;
}
#endif
static inline skip_lines_fn pick_skip_lines() {
#if RFX_X86
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt")) return skip_lines_avx2;
#endif
  return skip_lines_plain;
}

// starts[i] = offset of line i of [b, e) for i < max (line 0 starts at 0); returns the number of lines found
// (a last line without '\n' counts).
typedef size_t (*index_lines_fn)(const char* b, const char* e, uint64_t* starts, size_t max);
static inline size_t index_lines_plain(const char* b, const char* e, uint64_t* starts, size_t max) {
  size_t li = 0;
  const char* p = b;
  while (p < e && li < max) {
    starts[li++] = (uint64_t)(p - b);
    const char* nl = find_nl(p, e);
    p = nl ? nl + 1 : e;
  }
  return li;
}
#if RFX_X86
__attribute__((target("avx2"))) static inline size_t index_lines_avx2(const char* b, const char* e, uint64_t* starts,
                                                                      size_t max) {
  if (b >= e || max == 0) return 0;
  size_t li = 0;
  starts[li++] = 0;
  const __m256i nl = _mm256_set1_epi8('\n');
  const char* p = b;
  while (e - p >= 32 && li < max) {
    uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)p), nl));
    while (m && li < max) {
      starts[li++] = (uint64_t)(p - b) + (uint64_t)__builtin_ctz(m) + 1;
      m &= m - 1;
    }
    p += 32;
  }
  for (; p < e && li < max; ++p)
    if (*p == '\n') starts[li++] = (uint64_t)(p - b) + 1;
  if (li && starts[li - 1] >= (uint64_t)(e - b)) --li;  // the buffer ended with '\n': no line starts there
  return li;
}
#endif
static inline index_lines_fn pick_index_lines() {
#if RFX_X86
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx2")) return index_lines_avx2;
#endif
  return index_lines_plain;
}

// rfx_host_cpus() for the tools that do not link the device library (the SAM feeders): hardware threads, cut to the
// affinity mask and the cgroup CPU quota.
inline unsigned usable_cpus() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32];
    double per = 0;
    if (fscanf(f, "%31s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0)
      n = std::min<unsigned>(n, (unsigned)std::max(1.0, atof(q) / per + 0.999));
    fclose(f);
  }
  return n;
}

// The end of a tool whose outputs are flushed and closed.  Nothing is left to do but hand memory back: unpinning
// ~1 GB of staging blocks and unmapping the device arena call by call takes 0.2 s (measured: 15 % of a whole
// `jellyfish count` of 64 M reads) -- the kernel does the same at process exit anyway.  RFX_CLEAN_EXIT=1 returns
// instead, so that the caller runs its teardown (leak checks).
inline void leave(int status) {
  if (getenv("RFX_CLEAN_EXIT")) return;
  fflush(nullptr);
  _exit(status);
}

// A piece of a read-only file mapping that has been parsed: its page-table entries are dropped by the worker that parsed it
// (MADV_DONTNEED takes the address-space lock shared, so the workers do it side by side; the pages stay in the page cache).
// Otherwise the 5 million entries of a 20 GB mapping are torn down by ONE thread when the process exits -- 0.2 s of a
// tool that runs for 1.3.  Reading the range again merely faults it in again.  RFX_KEEP_PTES=1: leave them (A/B runs).
inline void drop_mapped(const char* b, const char* e) {
  static const bool keep = getenv("RFX_KEEP_PTES") != nullptr;
  if (keep || !b || e <= b) return;
  const uintptr_t P = 4096;
  const uintptr_t lo = ((uintptr_t)b + P - 1) & ~(P - 1), hi = (uintptr_t)e & ~(P - 1);
  if (hi > lo) (void)madvise((void*)lo, (size_t)(hi - lo), MADV_DONTNEED);
}

// RFX_CLI_TRACE=1: wall-clock marks of a tool's phases on stderr (scratch/cli_scale.sh reads them)
inline void trace(const char* what) {
  static const bool on = getenv("RFX_CLI_TRACE") != nullptr;
  static const auto t0 = std::chrono::steady_clock::now();
  if (on) fprintf(stderr, "[rfx %8.3f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what);
}

// The pages of a big output file cost as much as the copy into them (tmpfs: 5 GB/s on one thread, and neither
// parallel pwrite()s nor parallel faults on a mapping get past that -- inode and page-cache locks).  A tool that knows
// roughly how much it will write starts a background thread that fallocate()s the file beyond its (zero) size while
// the tool is still busy with its input; write_jhash() then sizes the file exactly (which frees any excess) and
// only copies.  Everything here is best effort: an unsupported or failed fallocate just leaves pages for later.
// Round 4: the same thread also enters the pages into the page table of the shared mapping write_jhash() will copy
// through (madvise(MADV_POPULATE_WRITE), Linux >= 5.14; for that the file is given its guessed size and cut to the
// real one at the end): 8.7 M first-touch faults of a 35.7 GB payload were what the copy threads spent their time on
// (profiles/r04_cli_w_sample.txt: 3.6 s for the payload = 9.9 GB/s, "waited 2.9 s for buffers").  RFX_NO_PREMAP=1: off.
// A piped input has no size to guess from: start_growing() + want(bytes) follow the stream (the file grows under a
// mapping of a fixed, large piece of address space; pages past the end of a file are simply not there yet).
// An output that was given a size before its content exists (OutputPrealloc) must not survive its writer's death: the
// reference scripts take a non-empty file for a finished one (`[ ! -s X.Jhash ]`, runRufus.sh:806,816; exit codes are
// not looked at).  The descriptor is registered here until write_jhash() has written everything: exit() (die()) and
// SIGINT / SIGTERM / SIGHUP cut the file back to 0 bytes.  (ftruncate and _exit are async-signal-safe.)
struct UnfinishedOutput {
  static std::atomic<int>& fd() {
    static std::atomic<int> f{-1};
    return f;
  }
  static void cut() {
    const int f = fd().exchange(-1);
    if (f >= 0) (void)!::ftruncate(f, 0);
  }
  static void on_signal(int sig) {
    cut();
    _exit(128 + sig);
  }
  static void watch(int f) {
    static bool installed = false;
    if (!installed) {
      installed = true;
      atexit(cut);
      struct sigaction sa;
      memset(&sa, 0, sizeof sa);
      sa.sa_handler = on_signal;
      for (int sig : {SIGINT, SIGTERM, SIGHUP}) sigaction(sig, &sa, nullptr);
    }
    fd() = f;
  }
  static void done(int f) {
    int want = f;
    fd().compare_exchange_strong(want, -1);
  }
};

class OutputPrealloc {
  int fd_ = -1;
  std::thread th_;
  std::atomic<bool> stop_{false};
  std::atomic<uint64_t> reached_{0}, populated_{0}, target_{0};
  bool growing_ = false;
  char* map_ = nullptr;
  size_t map_len_ = 0;

  void run() {
    // (64 MB: a populate holds the address space's lock for reading while it runs -- ~15 ms per step --, and the
    // device runtime takes it for writing whenever it maps memory)
    uint64_t step = 64ull << 20;
    if (const char* ev = getenv("RFX_PREALLOC_STEP")) step = std::max<uint64_t>(2ull << 20, strtoull(ev, nullptr, 10) & ~((2ull << 20) - 1));
    bool populate = map_ != nullptr;
    uint64_t at = 0;
    while (!stop_.load(std::memory_order_relaxed)) {
      uint64_t tgt = target_.load(std::memory_order_relaxed);
      if (growing_) tgt &= ~((2ull << 20) - 1);  // (a step begins on a page boundary: madvise() wants that)
      if (at >= tgt) {
        if (!growing_) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
        continue;
      }
      const uint64_t len = std::min(step, tgt - at);
      if (growing_ && map_ && ::ftruncate(fd_, (off_t)(at + len)) != 0) break;  // (the mapping's pages exist up to the file's size)
      if (::fallocate(fd_, FALLOC_FL_KEEP_SIZE, (off_t)at, (off_t)len) != 0) break;
      reached_ = at + len;
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
      if (populate && ::madvise(map_ + at, (size_t)len, MADV_POPULATE_WRITE) != 0) populate = false;  // (an older kernel: faults later)
      if (populate) populated_ = at + len;
      at += len;
    }
  }
  void map(size_t bytes) {
    if (getenv("RFX_NO_PREMAP")) return;
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
    if (m == MAP_FAILED) return;
    map_ = (char*)m;
    map_len_ = bytes;
  }

 public:
  // the output will be about `bytes` long
  void start(const char* path, uint64_t bytes) {
    fd_ = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd_ < 0 || bytes == 0) return;
    if (::ftruncate(fd_, (off_t)bytes) == 0) map((size_t)bytes);
    if (!map_) (void)!::ftruncate(fd_, 0);
    else UnfinishedOutput::watch(fd_);
    target_ = bytes;
    th_ = std::thread([this] { run(); });
  }
  // nobody knows yet: want() says how long it will be at least, as often as that changes
  void start_growing(const char* path) {
    fd_ = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    struct stat st;
    if (fd_ < 0) return;
    if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) {  // (-o /dev/stdout, a named pipe: nothing to prepare; the writer opens it)
      ::close(fd_);
      fd_ = -1;
      return;
    }
    growing_ = true;
    map((size_t)1 << 40);
    if (map_) UnfinishedOutput::watch(fd_);
    th_ = std::thread([this] { run(); });
  }
  void want(uint64_t bytes) {
    if (growing_ && bytes > target_.load(std::memory_order_relaxed)) target_ = bytes;
  }
  bool growing() const { return growing_; }
  uint64_t wanted() const { return target_; }
  uint64_t reached() const { return reached_; }      // bytes allocated so far
  uint64_t populated() const { return populated_; }  // .. and entered into the mapping's page table
  // stops the thread; the descriptor (or -1: the writer opens the file itself and reports the error) goes to the caller
  int take() {
    stop_ = true;
    if (th_.joinable()) th_.join();
    const int fd = fd_;
    fd_ = -1;
    return fd;
  }
  // after take(): the shared mapping of the file from its first byte on, .second bytes of address space (the file may be
  // shorter: write_jhash() sizes it), or {nullptr, 0}; the caller unmaps it
  std::pair<char*, size_t> take_mapping() {
    std::pair<char*, size_t> m{map_, map_len_};
    map_ = nullptr;
    map_len_ = 0;
    return m;
  }
  ~OutputPrealloc() {
    const int fd = take();
    if (map_) munmap(map_, map_len_);
    if (fd >= 0) {  // (never handed to a writer: whatever size it was given, it holds nothing)
      UnfinishedOutput::cut();
      ::close(fd);
    }
  }
};

// Records per fetch of the payload drain (x 2 per device slice, x 4 for the single-slice ring); the host-only test of the
// writer builds with a small one (tests/host/write_harness.cpp).
#ifndef RFX_WRITE_STEP
#define RFX_WRITE_STEP (1ull << 20)
#endif

// `lend`: page-locked buffers the caller no longer needs (the ingest's staging blocks), used as the drain ring
// instead of pinning more memory.
// `recs`: the payload in slices (one record set per device, rfx_count_set_peers: slice i holds the i-th range of output
// positions), written one after the other; each slice is fetched by its own thread.
inline void write_jhash(const char* path, const std::vector<rfx_records*>& recs, const uint64_t* cols, bool canonical,
                        int counter_len, int argc, char** argv, const std::vector<std::pair<char*, size_t>>& lend = {},
                        int open_fd = -1, uint64_t preallocated = 0, std::pair<char*, size_t> premap = {nullptr, 0}) {
  rfx_records* rec = recs.at(0);
  const int k = rfx_records_k(rec), lsize = rfx_records_lsize(rec);
  std::vector<char> hdr(1 << 16);
  const long hl = rfx_jhash_header(k, lsize, cols, canonical, counter_len, argc, argv, hdr.data(), hdr.size());
  if (hl < 0) die("rufus_amd: header too large");
  uint64_t n = 0;
  for (rfx_records* r : recs) n += rfx_records_size(r);
  const size_t rl = (size_t)(2 * k + 7) / 8 + (size_t)counter_len;
  int fd = open_fd >= 0 ? open_fd : ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);  // open_fd: OutputPrealloc's
  if (fd < 0) fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);  // a write-only special file
  if (fd < 0) die(std::string("Can't open output file '") + path + "'");
  auto put = [&](const char* p, size_t len, off_t at) {
    while (len) {
      const ssize_t w = ::pwrite(fd, p, len, at);
      if (w < 0 && errno == EINTR) continue;
      if (w <= 0) die(std::string("write error on '") + path + "': " + strerror(errno));
      p += w;
      len -= (size_t)w;
      at += w;
    }
  };
  // The payload of a 30x sample is ~35 GB: formatted on the device and fetched a few M records at a time into a
  // ring of page-locked buffers; writer threads copy finished buffers into the file.  Through a shared mapping of
  // the (pre-sized) file when it can be had: buffered write()s to ONE file serialise on its inode lock -- six
  // pwrite threads gave 3.8 GB/s into tmpfs, one thread's worth -- page faults of a mapping do not.
  const uint64_t total = (uint64_t)hl + n * rl;
  if (::lseek(fd, 0, SEEK_CUR) == (off_t)-1) {  // a pipe (-o /dev/stdout | ...): no offsets, one writer, in order
    auto put_seq = [&](const char* p, size_t len) {
      while (len) {
        const ssize_t w = ::write(fd, p, len);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) die(std::string("write error on '") + path + "': " + strerror(errno));
        p += w;
        len -= (size_t)w;
      }
    };
    put_seq(hdr.data(), (size_t)hl);
    const uint64_t step = RFX_WRITE_STEP;
    std::vector<char> b(step * rl);
    for (rfx_records* r : recs)
      for (uint64_t at = 0, cnt = rfx_records_size(r); at < cnt; at += step) {
        const uint64_t m = std::min<uint64_t>(step, cnt - at);
        if (rfx_records_payload_range(r, at, m, b.data(), (size_t)m * rl, counter_len) != RFX_OK)
          die(std::string("rufus_amd: drain failed: ") + rfx_last_error());
        put_seq(b.data(), (size_t)m * rl);
      }
    if (::close(fd) != 0) die(std::string("write error on '") + path + "'");
    return;
  }
  char* map = nullptr;
  size_t map_len = (size_t)total;  // what munmap() is given
  struct stat st_now;
  if (premap.first && n && premap.second >= total && fstat(fd, &st_now) == 0 &&
      ((uint64_t)st_now.st_size >= total || ::ftruncate(fd, (off_t)total) == 0)) {
    // OutputPrealloc's mapping (its pages are in the page table already, as far as the guess went); the file is cut to
    // its size at the end
    map = premap.first;
    map_len = premap.second;
  } else {
    if (premap.first) munmap(premap.first, premap.second);  // the guess was too small: map again at the real size
    premap = {nullptr, 0};
    // blocks preallocated past the end: ext4 keeps them when the size only grows -- grow over them, then cut back
    if (preallocated > total) (void)!::ftruncate(fd, (off_t)preallocated);
    if (n && ::ftruncate(fd, (off_t)total) == 0) {
      void* m = mmap(nullptr, (size_t)total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      if (m != MAP_FAILED) map = (char*)m;
    }
  }
  struct sigaction old_bus;
  if (map) {  // a full file system shows up as SIGBUS on the mapping, not as a failed write(): report it the same way
    static char bus_msg[512];
    snprintf(bus_msg, sizeof bus_msg, "write error on '%s': no space left on device (or the file was truncated)\n", path);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = [](int) {
      (void)!::write(2, bus_msg, strlen(bus_msg));
      _exit(1);
    };
    sigaction(SIGBUS, &sa, &old_bus);
    memcpy(map, hdr.data(), (size_t)hl);
  } else {
    put(hdr.data(), (size_t)hl, 0);
  }
  trace("write: header out");
  if (recs.size() > 1) {  // one fetch thread per slice (= per device), each with its own small ring
    std::vector<std::thread> th;
    uint64_t first = 0;
    for (rfx_records* r : recs) {
      const uint64_t base = first, cnt = rfx_records_size(r);
      first += cnt;
      th.emplace_back([=] {
        const uint64_t step = 2 * RFX_WRITE_STEP;
        const int NB = 3;
        char* b[NB];
        std::thread w[NB];
        for (int i = 0; i < NB; ++i)
          if (!(b[i] = (char*)rfx_host_alloc(step * rl))) die("rufus_amd: out of pinned host memory");
        for (uint64_t at = 0, i = 0; at < cnt; at += step, ++i) {
          const uint64_t m = std::min<uint64_t>(step, cnt - at);
          const int bi = (int)(i % NB);
          if (w[bi].joinable()) w[bi].join();
          if (rfx_records_payload_range(r, at, m, b[bi], (size_t)m * rl, counter_len) != RFX_OK)
            die(std::string("rufus_amd: drain failed: ") + rfx_last_error());
          const char* src = b[bi];
          const size_t len = (size_t)m * rl;
          const off_t off = (off_t)hl + (off_t)((base + at) * rl);
          if (map) w[bi] = std::thread([=] { memcpy(map + off, src, len); });
          else w[bi] = std::thread([=] { put(src, len, off); });
        }
        for (auto& x : w)
          if (x.joinable()) x.join();
        for (int i = 0; i < NB; ++i) rfx_host_free(b[i]);
      });
    }
    for (auto& t : th) t.join();
    trace("write: payload out (slices)");
    if (map) {
      if (munmap(map, map_len) != 0) die(std::string("write error on '") + path + "'");
      sigaction(SIGBUS, &old_bus, nullptr);
      if (premap.first && ::ftruncate(fd, (off_t)total) != 0) die(std::string("write error on '") + path + "'");
    } else if (open_fd >= 0) {
      (void)!::ftruncate(fd, (off_t)total);
    }
    UnfinishedOutput::done(fd);
    if (::close(fd) != 0) die(std::string("write error on '") + path + "'");
    return;
  }
  const uint64_t step = 4 * RFX_WRITE_STEP;
  const int NBUF = 8;
  char* buf[NBUF];
  int kind[NBUF];  // 0 malloc, 1 pinned here, 2 lent
  size_t lent = 0;
  for (int i = 0; i < NBUF; ++i) {
    buf[i] = nullptr;
    while (lent < lend.size() && !buf[i]) {
      if (lend[lent].second >= step * rl) buf[i] = lend[lent].first, kind[i] = 2;
      ++lent;
    }
    if (!buf[i] && n > (uint64_t)i * step) buf[i] = (char*)rfx_host_alloc(step * rl), kind[i] = 1;  // (only what a small payload needs)
    if (!buf[i]) buf[i] = (char*)malloc(step * rl), kind[i] = 0;
    if (!buf[i]) die("rufus_amd: out of host memory");
  }
  trace("write: buffers ready");
  std::thread writers[NBUF];
  double t_wait = 0, t_fetch = 0;  // main thread: waiting for a free buffer / formatting + device-to-host copy
  for (uint64_t at = 0, i = 0; at < n; at += step, ++i) {
    const uint64_t m = std::min<uint64_t>(step, n - at);
    const int bi = (int)(i % NBUF);
    const auto ta = std::chrono::steady_clock::now();
    if (writers[bi].joinable()) writers[bi].join();
    const auto tb = std::chrono::steady_clock::now();
    if (rfx_records_payload_range(rec, at, m, buf[bi], (size_t)m * rl, counter_len) != RFX_OK)
      die(std::string("rufus_amd: drain failed: ") + rfx_last_error());
    t_wait += std::chrono::duration<double>(tb - ta).count();
    t_fetch += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
    const char* src = buf[bi];
    const size_t len = (size_t)m * rl;
    const off_t off = (off_t)hl + (off_t)(at * rl);
    // (NOT followed by drop_mapped(): dropping the entries of dirty pages of a SHARED mapping from the writer threads was
    // measured slower than the one munmap below -- 37 GB: 2.7 - 4.3 s of writing against 0.7 + 1.55 s)
    if (map) writers[bi] = std::thread([=] { memcpy(map + off, src, len); });
    else writers[bi] = std::thread([=] { put(src, len, off); });
  }
  for (auto& w : writers)
    if (w.joinable()) w.join();
  {
    char msg[128];
    snprintf(msg, sizeof msg, "write: payload out (%s; waited %.3f s for buffers, %.3f s fetching)", map ? "mapped" : "pwrite",
             t_wait, t_fetch);
    trace(msg);
  }
  for (int i = 0; i < NBUF; ++i) {
    if (kind[i] == 1) rfx_host_free(buf[i]);
    else if (kind[i] == 0) free(buf[i]);
  }
  if (map) {
    if (munmap(map, map_len) != 0) die(std::string("write error on '") + path + "'");
    sigaction(SIGBUS, &old_bus, nullptr);
    if (premap.first && ::ftruncate(fd, (off_t)total) != 0) die(std::string("write error on '") + path + "'");  // the guessed size -> the real one
  } else if (open_fd >= 0) {
    (void)!::ftruncate(fd, (off_t)total);  // drop what was preallocated past the end
  }
  UnfinishedOutput::done(fd);
  if (::close(fd) != 0) die(std::string("write error on '") + path + "'");
}

inline void write_jhash(const char* path, rfx_records* rec, const uint64_t* cols, bool canonical, int counter_len,
                        int argc, char** argv, const std::vector<std::pair<char*, size_t>>& lend = {}, int open_fd = -1,
                        uint64_t preallocated = 0, std::pair<char*, size_t> premap = {nullptr, 0}) {
  write_jhash(path, std::vector<rfx_records*>{rec}, cols, canonical, counter_len, argc, argv, lend, open_fd, preallocated, premap);
}

// yaggo's SI suffixes (jf/sub_commands/count_main_cmdline.hpp:104-109): k M G T P E are powers of 1000.
inline bool parse_si(const char* s, uint64_t& v) {
  char* end;
  const unsigned long long x = strtoull(s, &end, 10);
  if (end == s) return false;
  uint64_t mul = 1;
  switch (*end) {
    case 0: break;
    case 'k': mul = 1000ull; ++end; break;
    case 'M': mul = 1000000ull; ++end; break;
    case 'G': mul = 1000000000ull; ++end; break;
    case 'T': mul = 1000000000000ull; ++end; break;
    case 'P': mul = 1000000000000000ull; ++end; break;
    case 'E': mul = 1000000000000000000ull; ++end; break;
    default: return false;
  }
  if (*end) return false;
  v = x * mul;
  return true;
}

}  // namespace rfxcli
