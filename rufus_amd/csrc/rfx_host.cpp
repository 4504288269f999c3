// Host-only half of the C-ABI: jellyfish hash matrix, read packing, hash-list loader, .Jhash header.
// No device code here; everything is callable on a machine without a GPU.
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include <sched.h>
#include <unistd.h>
#include <sys/utsname.h>

#include "../../include/rufus_hip.h"
#include "rfx_synth.h"

namespace {

// glibc's default random(): TYPE_3 additive feedback generator x[i] = x[i-3] + x[i-31], seeded
// with 1 when the program never calls srandom() -- which jellyfish never does
// (jf/lib/misc.cc:74-80 calls random() directly).  Re-implemented so the matrix does not depend on
// the process-wide libc state of whoever loads this library.
struct GlibcRandom {
  uint32_t st[31];
  int f, r;  // front / rear cursors of the 31-word ring (separation 3)
  explicit GlibcRandom(uint32_t seed = 1) : f(3), r(0) {
    int32_t w = (int32_t)(seed ? seed : 1);
    st[0] = (uint32_t)w;
    for (int i = 1; i < 31; ++i) {
      // 16807 * w mod (2^31 - 1) by Schrage's method, as srandom_r does
      const int32_t hi = w / 127773, lo = w % 127773;
      w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      st[i] = (uint32_t)w;
    }
    for (int i = 0; i < 310; ++i) next();  // srandom_r discards 10 * 31 outputs
  }
  uint32_t next() {
    st[f] += st[r];
    const uint32_t out = st[f] >> 1;
    f = (f + 1) % 31;
    r = (r + 1) % 31;
    return out;
  }
};

// jf/lib/misc.cc:74-80 with ConstFloorLog2<RAND_MAX>::val == 30.
uint64_t random_bits64(GlibcRandom& g) {
  uint64_t res = 0;
  for (int i = 0; i < 64; i += 30) res ^= (uint64_t)g.next() << i;
  return res;
}

// Column-wise Gauss-Jordan of the r x c matrix whose top (c-r) rows are an implicit identity;
// returns false if singular.  jf/lib/rectangular_binary_matrix.cc:138-186.
bool pseudo_inverse(std::vector<uint64_t> piv, int r, int c, std::vector<uint64_t>& inv) {
  inv.assign(c, 0);
  const int first = c - r;
  for (int i = first; i < c; ++i) inv[i] = 1ull << (r - 1 - (i - first));
  for (int i = first; i < c; ++i) {
    const uint64_t bit = 1ull << (r - 1 - (i - first));
    if (!(piv[i] & bit)) {
      int j = i + 1;
      while (j < c && !(piv[j] & bit)) ++j;
      if (j == c) return false;
      piv[i] ^= piv[j];
      inv[i] ^= inv[j];
    }
    for (int j = i + 1; j < c; ++j)
      if (piv[j] & bit) {
        piv[j] ^= piv[i];
        inv[j] ^= inv[i];
      }
  }
  for (int i = first; i < c; ++i) {
    const uint64_t bit = 1ull << (r - 1 - (i - first));
    for (int j = 0; j < i; ++j)
      if (piv[j] & bit) {
        piv[j] ^= piv[i];
        inv[j] ^= inv[i];
      }
  }
  return true;
}


// Util::Split (src/Util.cpp:24-33): std::getline tokens, no trailing empty token.
std::vector<std::string> split(const std::string& s, char d) {
  std::vector<std::string> t;
  size_t i = 0;
  while (i < s.size()) {
    const size_t j = s.find(d, i);
    if (j == std::string::npos) {
      t.push_back(s.substr(i));
      break;
    }
    t.push_back(s.substr(i, j - i));
    i = j + 1;
  }
  return t;
}

// Util::HashToLong (src/Util.cpp:51-84) of `s`, returned as a forward jellyfish key of k bases:
// base i contributes its code at bits 2(k-1-i); unknown characters and missing positions are A.
uint64_t rufus_key(const std::string& s, int k) {
  uint64_t key = 0;
  for (int i = 0; i < k; ++i) {
    uint64_t code = 0;
    if (i < (int)s.size()) {
      switch (s[i]) {
        case 'C': code = 1; break;
        case 'G': code = 2; break;
        case 'T': code = 3; break;
        default: break;
      }
    }
    key |= code << (2 * (k - 1 - i));
  }
  return key;
}

// Util::RevComp (src/Util.cpp:187-210): ACGTN complemented, everything else dropped.
std::string rufus_revcomp(const std::string& s) {
  std::string o;
  for (size_t i = s.size(); i-- > 0;) {
    switch (s[i]) {
      case 'A': o += 'T'; break;
      case 'C': o += 'G'; break;
      case 'G': o += 'C'; break;
      case 'T': o += 'A'; break;
      case 'N': o += 'N'; break;
      default: break;
    }
  }
  return o;
}

void json_escape(std::string& out, const char* s) {
  out += '"';
  for (; *s; ++s) {
    const unsigned char ch = (unsigned char)*s;
    if (ch == '"' || ch == '\\') { out += '\\'; out += (char)ch; }
    else if (ch == '\n') out += "\\n";
    else if (ch == '\t') out += "\\t";
    else if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); out += b; }
    else out += (char)ch;
  }
  out += '"';
}

}  // namespace

extern "C" {

const char* rfx_version(void) { return "rufus_amd 0.1 (gfx950)"; }

const char* rfx_strerror(int code) {
  switch (code) {
    case RFX_OK: return "ok";
    case RFX_E_NODEVICE: return "no usable gfx950 device";
    case RFX_E_INVAL: return "invalid argument";
    case RFX_E_NOMEM: return "out of memory / hbm budget exceeded";
    case RFX_E_FULL: return "count table full";
    case RFX_E_HIP: return "HIP runtime error";
    case RFX_E_MIXEDCASE: return "lower-case bases: pack count and filter blocks separately";
    case RFX_E_RANGE: return "output buffer too small";
    case RFX_E_FORMAT: return "malformed input";
    default: return "unknown error";
  }
}

int rfx_jf_matrix(int lsize, int k, uint64_t* cols) {
  const int r = lsize, c = 2 * k;
  if (r < 1 || r > 64 || k < 1 || c < r || !cols) return RFX_E_INVAL;
  GlibcRandom rng(1);
  const uint64_t cmask = ~0ull >> (64 - r);
  std::vector<uint64_t> m(c), inv;
  do {
    for (int i = 0; i < c; ++i) m[i] = random_bits64(rng) & cmask;
  } while (!pseudo_inverse(m, r, c, inv));
  memcpy(cols, inv.data(), sizeof(uint64_t) * c);
  return RFX_OK;
}

uint64_t rfx_jf_pos(const uint64_t* cols, int k, int lsize, uint64_t key) {
  const int c = 2 * k;
  uint64_t res = 0;
  for (int b = 0; b < c && b < 64; ++b)
    if ((key >> b) & 1) res ^= cols[c - 1 - b];
  return lsize >= 64 ? res : res & ((1ull << lsize) - 1);
}

uint64_t rfx_pack_words(const uint64_t* off, uint32_t n_reads) {
  uint64_t w = 0;
  for (uint32_t i = 0; i < n_reads; ++i) w += (off[i + 1] - off[i] + 31) / 32;
  return w;
}

namespace {

// Per character: jellyfish code (either case), "is ACGT/acgt", RUFUS code (upper case only, src/Util.cpp:51-84),
// and "lower-case c/g/t" (where the two encodings disagree).
struct PackLut {
  uint8_t jcode[256], jvalid[256], fcode[256], lower_cgt[256];
  PackLut() {
    memset(this, 0, sizeof *this);
    const char* up = "ACGT";
    const char* lo = "acgt";
    for (int i = 0; i < 4; ++i) {
      jcode[(unsigned char)up[i]] = jcode[(unsigned char)lo[i]] = (uint8_t)i;
      jvalid[(unsigned char)up[i]] = jvalid[(unsigned char)lo[i]] = 1;
      fcode[(unsigned char)up[i]] = (uint8_t)i;
      if (i > 0) lower_cgt[(unsigned char)lo[i]] = 1;
    }
  }
};
const PackLut g_pack_lut;

// One read: L bases at s (qualities at q, may be null = all '\0') into ceil(L / 32) words at codes / acgt / good.
// The packing rules (one place, two implementations that must agree bit for bit):
//   count only      code = A C G T a c g t -> 0 1 2 3 0 1 2 3, anything else 0; acgt bit = one of those eight
//   filter          code = A C G T -> 0 1 2 3, anything else (lower case too) 0     (src/Util.cpp:51-84)
//                   good bit = !(q - 33 < min_q || base == 'N')                     (src/RUFUS.Filter.cpp:205)
//   filter + count  as filter, plus the acgt bit; lower-case c g t -> RFX_E_MIXEDCASE (the two tools disagree there)
typedef int (*pack_one_fn)(const unsigned char* s, const signed char* q, uint32_t L, int min_q, bool want_count,
                           bool want_filter, uint64_t* codes, uint32_t* acgt, uint32_t* good);

int pack_one_scalar(const unsigned char* s, const signed char* q, uint32_t L, int min_q, bool want_count,
                    bool want_filter, uint64_t* codes, uint32_t* acgt, uint32_t* good) {
  const PackLut& T = g_pack_lut;
  for (uint32_t i = 0, w = 0; i < L; i += 32, ++w) {
    uint64_t cw = 0;
    uint32_t ma = 0, mg = 0;
    const uint32_t nb = std::min<uint32_t>(32, L - i);
    if (!want_filter) {
      for (uint32_t b = 0; b < nb; ++b) {
        cw |= (uint64_t)T.jcode[s[i + b]] << (2 * b);
        ma |= (uint32_t)T.jvalid[s[i + b]] << b;
      }
    } else {
      for (uint32_t b = 0; b < nb; ++b) {
        const unsigned char ch = s[i + b];
        cw |= (uint64_t)T.fcode[ch] << (2 * b);
        if (want_count) {
          if (T.lower_cgt[ch]) return RFX_E_MIXEDCASE;
          ma |= (uint32_t)T.jvalid[ch] << b;
        }
        const int qv = q ? (int)q[i + b] : 0;
        mg |= (uint32_t)(!(qv - 33 < min_q || ch == 'N')) << b;
      }
    }
    codes[w] = cw;
    if (acgt) acgt[w] = ma;
    if (good) good[w] = mg;
  }
  return RFX_OK;
}

// 32 bases per step: compares give the masks, and with u = base & 0xDF the code is bit0 = u.1 ^ u.2, bit1 = u.2 ^ u.3
// (A 0x41 -> 00, C 0x43 -> 01, G 0x47 -> 10, T 0x54 -> 11): two byte-mask extractions, interleaved by pdep.
// The host parsers of the drop-in tools spend most of their time here: 3.2 cycles per base scalar.
__attribute__((target("avx2,bmi2"))) int pack_one_avx2(const unsigned char* s, const signed char* q, uint32_t L, int min_q,
                                                       bool want_count, bool want_filter, uint64_t* codes,
                                                       uint32_t* acgt, uint32_t* good) {
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'),
                cT = _mm256_set1_epi8('T'), cN = _mm256_set1_epi8('N'), fold = _mm256_set1_epi8((char)0xDF);
  const int qmin = min_q + 33;  // good needs q >= qmin (signed chars, like the reference's plain `char`)
  const __m256i qlim = _mm256_set1_epi8((char)std::max(-128, std::min(127, qmin)));
  for (uint32_t i = 0, w = 0; i < L; i += 32, ++w) {
    const uint32_t nb = std::min<uint32_t>(32, L - i);
    __m256i v, qv = _mm256_setzero_si256();
    if (nb == 32) {
      v = _mm256_loadu_si256((const __m256i*)(s + i));
      if (want_filter && q) qv = _mm256_loadu_si256((const __m256i*)(q + i));
    } else {  // the last word of a read: through a zero-padded copy (a zero byte packs as 0 / not valid / not good)
      alignas(32) unsigned char tmp[32] = {0};
      memcpy(tmp, s + i, nb);
      v = _mm256_load_si256((const __m256i*)tmp);
      if (want_filter && q) {
        alignas(32) signed char tq[32] = {0};
        memcpy(tq, q + i, nb);
        qv = _mm256_load_si256((const __m256i*)tq);
      }
    }
    const __m256i u = _mm256_and_si256(v, fold);
    const __m256i any_case = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                             _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
    const uint32_t m_any = (uint32_t)_mm256_movemask_epi8(any_case);
    const __m256i t = _mm256_xor_si256(u, _mm256_srli_epi16(u, 1));  // bit 1 = u.1 ^ u.2, bit 2 = u.2 ^ u.3
    uint32_t b0 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(t, 6));
    uint32_t b1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(t, 5));
    uint32_t m_code = m_any, mg = 0;
    if (want_filter) {
      const __m256i upper = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, cA), _mm256_cmpeq_epi8(v, cC)),
                                            _mm256_or_si256(_mm256_cmpeq_epi8(v, cG), _mm256_cmpeq_epi8(v, cT)));
      m_code = (uint32_t)_mm256_movemask_epi8(upper);
      if (want_count) {  // lower-case c g t: in any_case, not upper case, and not 'a'
        const uint32_t m_a = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('a')));
        if (m_any & ~m_code & ~m_a) return RFX_E_MIXEDCASE;
      }
      uint32_t bad = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, cN));
      if (qmin > 127) bad = ~0u;
      else if (qmin > -128) bad |= (uint32_t)_mm256_movemask_epi8(_mm256_cmpgt_epi8(qlim, qv));
      mg = ~bad;
      if (nb < 32) mg &= (1u << nb) - 1;
    }
    b0 &= m_code;
    b1 &= m_code;
    codes[w] = _pdep_u64(b0, 0x5555555555555555ull) | _pdep_u64(b1, 0xAAAAAAAAAAAAAAAAull);
    if (acgt) acgt[w] = m_any;
    if (good) good[w] = mg;
  }
  return RFX_OK;
}

pack_one_fn pick_pack_one() {
  if (getenv("RFX_PACK_SCALAR")) return pack_one_scalar;  // tests compare the two
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") ? pack_one_avx2 : pack_one_scalar;
}
const pack_one_fn g_pack_one = pick_pack_one();

// Reads [r0, r1): their word offsets are already in word_off.  Returns RFX_OK or RFX_E_MIXEDCASE.
int pack_range(const char* seq, const char* qual, const uint64_t* off, uint32_t r0, uint32_t r1, int min_q, bool want_count,
               bool want_filter, uint64_t* codes, uint32_t* acgt, uint32_t* good, const uint32_t* word_off) {
  for (uint32_t r = r0; r < r1; ++r) {
    const uint64_t b0 = off[r], L = off[r + 1] - off[r];
    const uint64_t w = word_off[r];
    const int rc = g_pack_one((const unsigned char*)seq + b0, qual ? (const signed char*)qual + b0 : nullptr, (uint32_t)L,
                              min_q, want_count, want_filter, codes + w, acgt ? acgt + w : nullptr,
                              good ? good + w : nullptr);
    if (rc) return rc;
  }
  return RFX_OK;
}

}  // namespace

unsigned rfx_host_cpus(void) {
  static const unsigned cached = [] {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    auto cut = [&](double quota, double period) {
      if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1.0, std::ceil(quota / period)));
    };
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
      char q[32];
      double per;
      if (fscanf(f, "%31s %lf", q, &per) == 2 && strcmp(q, "max") != 0) cut(atof(q), per);
      fclose(f);
    } else {  // cgroup v1
      double q = -1, per = 0;
      if (FILE* a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(a, "%lf", &q) != 1) q = -1; fclose(a); }
      if (FILE* b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(b, "%lf", &per) != 1) per = 0; fclose(b); }
      cut(q, per);
    }
    return n;
  }();
  return cached;
}

int rfx_pack_reads(const char* seq, const char* qual, const uint64_t* off, uint32_t n_reads, int min_q, int flags,
                   uint64_t* codes, uint32_t* acgt, uint32_t* good, uint32_t* word_off, uint32_t* len) {
  if (!seq || !off || !codes || !word_off || !len) return RFX_E_INVAL;
  const bool want_count = flags & RFX_PACK_COUNT, want_filter = flags & RFX_PACK_FILTER;
  if ((want_count && !acgt) || (want_filter && !good) || (!want_count && !want_filter)) return RFX_E_INVAL;
  uint64_t w = 0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    const uint64_t L = off[r + 1] - off[r];
    if (L > 0xFFFFFFFFull || w > 0xFFFFFFFFull) return RFX_E_RANGE;
    word_off[r] = (uint32_t)w;
    len[r] = (uint32_t)L;
    w += (L + 31) / 32;
  }
  if (w > 0xFFFFFFFFull) return RFX_E_RANGE;
  word_off[n_reads] = (uint32_t)w;
  // Packing is a pure per-read map: split the reads into ranges of equal bases over host threads once the
  // batch is big enough to pay for them (RFX_HOST_THREADS overrides the count; 1 = inline).
  const uint64_t bases = n_reads ? off[n_reads] - off[0] : 0;
  unsigned nt = (unsigned)std::min<uint64_t>(bases / (2u << 20), 32);
  nt = std::min(nt, rfx_host_cpus());
  if (const char* ev = getenv("RFX_HOST_THREADS")) nt = (unsigned)atoi(ev);
  if (nt <= 1 || n_reads < 2 * nt)
    return pack_range(seq, qual, off, 0, n_reads, min_q, want_count, want_filter, codes, acgt, good, word_off);
  std::vector<uint32_t> cut(nt + 1, n_reads);
  cut[0] = 0;
  for (unsigned t = 1; t < nt; ++t)
    cut[t] = (uint32_t)(std::lower_bound(off, off + n_reads, off[0] + bases * t / nt) - off);
  std::atomic<int> err{RFX_OK};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      const int rc = pack_range(seq, qual, off, cut[t], cut[t + 1], min_q, want_count, want_filter, codes, acgt, good,
                                word_off);
      if (rc) err.store(rc);
    });
  for (auto& x : th) x.join();
  return err.load();
}

// Packing of reads that lie scattered in a text buffer (FASTQ records parsed in place): read i is the seq_len[i]
// bytes at base + seq_start[i], its qualities (RFX_PACK_FILTER) the same number of bytes at base + qual_start[i].
// Single-threaded: the callers (the drop-in executables' ingest pipelines) run one call per worker thread on
// disjoint output ranges.  word_off[0] is an INPUT (the first word index of this range in the block), the
// other n entries and len[0..n) are written; codes / acgt / good are indexed by those word offsets.
int rfx_pack_spans(const char* base, const uint64_t* seq_start, const uint32_t* seq_len, const uint64_t* qual_start,
                   uint32_t n_reads, int min_q, int flags, uint64_t* codes, uint32_t* acgt, uint32_t* good,
                   uint32_t* word_off, uint32_t* len) {
  if (!base || !seq_start || !seq_len || !codes || !word_off || !len) return RFX_E_INVAL;
  const bool want_count = flags & RFX_PACK_COUNT, want_filter = flags & RFX_PACK_FILTER;
  if ((want_count && !acgt) || (want_filter && (!good || !qual_start)) || (!want_count && !want_filter)) return RFX_E_INVAL;
  uint64_t w = word_off[0];
  for (uint32_t r = 0; r < n_reads; ++r) {
    const uint32_t L = seq_len[r];
    word_off[r] = (uint32_t)w;
    len[r] = L;
    if (w + (L + 31) / 32 > 0xFFFFFFFFull) return RFX_E_RANGE;
    // (offsets are byte distances from `base` modulo 2^64: a span may lie in another allocation than `base`)
    const int rc = g_pack_one((const unsigned char*)((uintptr_t)base + (uintptr_t)seq_start[r]),
                              want_filter ? (const signed char*)((uintptr_t)base + (uintptr_t)qual_start[r]) : nullptr, L, min_q, want_count,
                              want_filter, codes + w, acgt ? acgt + w : nullptr, good ? good + w : nullptr);
    if (rc) return rc;
    w += (L + 31) / 32;
  }
  word_off[n_reads] = (uint32_t)w;
  return RFX_OK;
}

long rfx_hashlist_keys(const char* text, size_t n, int k, int single_end, uint64_t* keys_out, size_t cap) {
  if (!text || k < 1 || k > 32) return RFX_E_INVAL;
  const char first = single_end ? '\t' : ' ', second = single_end ? ' ' : '\t';
  long count = 0;
  const char *p = text, *end = text + n;
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', end - p);
    const std::string line(p, nl ? nl : end);
    p = nl ? nl + 1 : end;
    std::vector<std::string> t = split(line, first);
    std::string kmer;
    bool have = false;
    if (t.size() == 2) { kmer = t[0]; have = true; }
    else if (t.size() == 4) { kmer = t[3]; have = true; }
    if (t.size() == 1) {
      t = split(line, second);
      if (!t.empty()) { kmer = t[0]; have = true; }
    }
    if (!have) continue;
    // A list entry LONGER than k: Util::HashToLong (src/Util.cpp:51-84) packs up to 32 bases, so its value has bits
    // above 2k and equals no k-base window -- unless every extra base encodes 00 ('A' or an invalid character).
    // Such entries still count (the reference's map holds them) but get a key no window can have.
    auto key_of = [&](const std::string& s_) -> uint64_t {
      uint64_t key = rufus_key(s_, k);
      if (k < 32)
        for (size_t i = (size_t)k; i < s_.size() && i < 32; ++i)
          if (s_[i] == 'C' || s_[i] == 'G' || s_[i] == 'T') return key | (1ull << 63);
      return key;
    };
    const uint64_t fw = key_of(kmer), rv = key_of(rufus_revcomp(kmer));
    if (keys_out) {
      if ((size_t)count + 2 > cap) return RFX_E_RANGE;
      keys_out[count] = fw;
      keys_out[count + 1] = rv;
    }
    count += 2;
  }
  return count;
}

long rfx_jhash_header(int k, int lsize, const uint64_t* cols, int canonical, int counter_len, int argc,
                      const char* const* argv, char* buf, size_t cap) {
  if (!cols || !buf || k < 1 || lsize < 1 || lsize > 63) return RFX_E_INVAL;
  // jf/include/jellyfish/large_hash_array.hpp:37-48: the reprobe limit shrinks until its offset fits the table.
  auto reprobe = [](int i) -> uint64_t { return i == 0 ? 1 : (uint64_t)i * (i + 1) / 2; };
  int max_reprobe = 126;  // jf/sub_commands/count_main_cmdline.hpp:361-369 default
  const uint64_t size = 1ull << lsize;
  while (max_reprobe >= 1 && reprobe(max_reprobe) >= size) --max_reprobe;

  char tmp[4096];
  struct utsname un;
  std::string host = uname(&un) == 0 ? un.nodename : "";
  std::string pwd = getcwd(tmp, sizeof tmp) ? tmp : "";
  ssize_t l = readlink("/proc/self/exe", tmp, sizeof tmp - 1);
  std::string exe = l > 0 ? std::string(tmp, (size_t)l) : "";
  time_t now = time(nullptr);
  std::string when = ctime_r(&now, tmp) ? tmp : "";
  while (!when.empty() && (when.back() == '\n' || when.back() == ' ')) when.pop_back();

  // Keys in std::map (alphabetical) order like Json::FastWriter (generic_file_header.hpp:96-121).
  std::string js = "{\"alignment\":8,\"canonical\":";
  js += canonical ? "true" : "false";
  js += ",\"cmdline\":[";
  for (int i = 0; i < argc; ++i) {
    if (i) js += ',';
    json_escape(js, argv[i]);
  }
  js += "],\"counter_len\":" + std::to_string(counter_len);
  js += ",\"exe_path\":"; json_escape(js, exe.c_str());
  js += ",\"format\":\"binary/sorted\",\"hostname\":"; json_escape(js, host.c_str());
  js += ",\"key_len\":" + std::to_string(2 * k);
  js += ",\"matrix1\":{\"c\":" + std::to_string(2 * k) + ",\"columns\":[";
  for (int i = 0; i < 2 * k; ++i) {
    if (i) js += ',';
    js += std::to_string((unsigned long long)cols[i]);
  }
  js += "],\"r\":" + std::to_string(lsize) + "}";
  js += ",\"max_reprobe\":" + std::to_string(max_reprobe);
  js += ",\"pwd\":"; json_escape(js, pwd.c_str());
  js += ",\"reprobes\":[";
  for (int i = 0; i <= max_reprobe; ++i) {
    if (i) js += ',';
    js += std::to_string((unsigned long long)reprobe(i));
  }
  js += "],\"size\":" + std::to_string((unsigned long long)size);
  js += ",\"time\":"; json_escape(js, when.c_str());
  js += ",\"val_len\":7}";

  size_t hlen = js.size();
  const size_t rem = (9 + js.size()) % 8;
  if (rem) hlen += 8 - rem;
  if (9 + hlen > cap) return RFX_E_RANGE;
  snprintf(buf, 10, "%09zu", hlen);
  memcpy(buf + 9, js.data(), js.size());
  memset(buf + 9 + js.size(), 0, hlen - js.size());
  return (long)(9 + hlen);
}

// ---- synthetic workload, host twin of rfx_synth.hip ---------------------------------
int rfx_synth_check(const rfx_synth* p) {
  if (!p || p->genome_len < 4000 || p->genome_len >= (1ull << 32) || p->read_len == 0 || p->read_len > 256) return RFX_E_INVAL;
  if (p->insert_lo < p->read_len || p->insert_span == 0 || p->insert_lo + p->insert_span + 1 >= p->genome_len) return RFX_E_INVAL;
  if (p->n_snv && (p->genome_len - 2000) / p->n_snv < 2ull * p->read_len + 64) return RFX_E_INVAL;
  return RFX_OK;
}

int rfx_synth_snv(const rfx_synth* p, uint32_t i, uint64_t* pos, char* ref, char* alt) {
  if (rfx_synth_check(p) || i >= p->n_snv) return RFX_E_INVAL;
  const uint64_t q = rfxs::snv_pos(*p, i);
  const uint32_t r = rfxs::genome_base(*p, q);
  if (pos) *pos = q;
  if (ref) *ref = "ACGT"[r];
  if (alt) *alt = "ACGT"[rfxs::snv_alt(*p, i, r)];
  return RFX_OK;
}

int rfx_synth_genome(const rfx_synth* p, uint64_t first, uint64_t n, char* out) {
  if (rfx_synth_check(p) || !out || first + n > p->genome_len) return RFX_E_INVAL;
  for (uint64_t i = 0; i < n; ++i) out[i] = "ACGT"[rfxs::genome_base(*p, first + i)];
  return RFX_OK;
}

int rfx_synth_text(const rfx_synth* p, uint64_t first_pair, uint32_t n_pairs, char* seq, char* qual) {
  if (rfx_synth_check(p) || !seq || !qual) return RFX_E_INVAL;
  const uint32_t L = p->read_len;
  const uint64_t st = rfxs::snv_stride(*p);
  auto one_pair = [&](uint32_t q) {
    const rfxs::pair_geom g = rfxs::pair_of(*p, first_pair + q);
    for (int mate = 0; mate < 2; ++mate) {
      const uint64_t mk = rfxs::mate_key(g, mate);
      char* s = seq + ((size_t)2 * q + mate) * L;
      char* ql = qual + ((size_t)2 * q + mate) * L;
      for (uint32_t j = 0; j < L; ++j) {
        const uint64_t x = rfxs::base_coord(*p, g, mate, j);
        uint32_t b = rfxs::genome_base(*p, x);
        if (p->carrier && g.hap && p->n_snv && x >= 1000) {
          const uint64_t i = std::min<uint64_t>((x - 1000) / st, p->n_snv - 1);
          if (rfxs::snv_pos(*p, i) == x) b = rfxs::snv_alt(*p, i, b);
        }
        const rfxs::base_out o = rfxs::finish_base(*p, b, mate, rfxs::base_bits(mk, j));
        s[j] = o.is_n ? 'N' : "ACGT"[o.code];
        ql[j] = o.lowq ? '#' : 'J';
      }
    }
  };
  unsigned nt = std::min<unsigned>(rfx_host_cpus(), 64);
  if (const char* ev = getenv("RFX_HOST_THREADS")) nt = std::max(1, atoi(ev));
  if (n_pairs < 4096 || nt <= 1) {
    for (uint32_t q = 0; q < n_pairs; ++q) one_pair(q);
    return RFX_OK;
  }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      for (uint32_t q = (uint64_t)n_pairs * t / nt, e = (uint64_t)n_pairs * (t + 1) / nt; q < e; ++q) one_pair(q);
    });
  for (auto& x : th) x.join();
  return RFX_OK;
}

}  // extern "C"
