// TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the RUFUS hot path.
//
// Nothing under rufus_amd/ (the product) may include, link, import or execute this file or its
// build products.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
// only as the checker.  Every function cites the reference file:line it restates
// (paths relative to /root/reference).  "jf/" abbreviates src/modifiedJellyfish/ (byte-identical
// to stock jellyfish 2.2.5 except jellyfish/merge_files.cc).
//
// Parity pins (see oracle/README.md): jellyfish's own md5 known-answer tests
// (tests/parallel_hashing.sh inside src/externals/jellyfish-2.2.5.tar.gz), the matrix/record
// probe values of SURVEY.md 8a-F', and -- for the RUFUS tools, which DO compile here -- direct
// comparison with oracle/_ref/* on the testRun fixtures and on seeded synthetic trios.

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// jellyfish hash matrix  (jf/lib/misc.cc:74-80, jf/include/jellyfish/rectangular_binary_matrix.hpp
// :107-110, jf/lib/rectangular_binary_matrix.cc:138-186 and :209-216)
// ---------------------------------------------------------------------------------------------

// misc.cc:74-80: XOR of glibc random() outputs shifted by multiples of floor(log2(RAND_MAX)) = 30.
uint64_t jf_random_bits64() {
  uint64_t res = 0;
  for (int i = 0; i < 64; i += 30) res ^= (uint64_t)random() << i;
  return res;
}

// rectangular_binary_matrix.cc:138-186.  Columns scol..c-1 are reduced to the "low identity";
// the same column operations are applied to res, which starts as the low identity.
bool jf_pseudo_inverse(const std::vector<uint64_t>& m, unsigned r, unsigned c, std::vector<uint64_t>& res) {
  std::vector<uint64_t> pivot(m);
  res.assign(c, 0);
  const unsigned srow = std::min(r, c), scol = c - srow;
  res[scol] = (uint64_t)1 << (srow - 1);  // init_low_identity(), :36-43
  for (unsigned i = scol + 1; i < c; ++i) res[i] = res[i - 1] >> 1;

  uint64_t mask = (uint64_t)1 << (srow - 1);
  for (unsigned i = scol; i < c; ++i, mask >>= 1) {
    if (!(pivot[i] & mask)) {
      unsigned j;
      for (j = i + 1; j < c; ++j)
        if (pivot[j] & mask) break;
      if (j == c) return false;  // singular
      pivot[i] ^= pivot[j];
      res[i] ^= res[j];
    }
    for (unsigned j = i + 1; j < c; ++j)
      if (pivot[j] & mask) {
        pivot[j] ^= pivot[i];
        res[j] ^= res[i];
      }
  }
  mask = (uint64_t)1 << (srow - 1);
  for (unsigned i = scol; i < c; ++i, mask >>= 1)
    for (unsigned j = 0; j < i; ++j)
      if (pivot[j] & mask) {
        pivot[j] ^= pivot[i];
        res[j] ^= res[i];
      }
  return true;
}

// rectangular_binary_matrix.hpp:206-243 (times_loop): bit b of the key selects column c-1-b.
inline uint64_t jf_times(const uint64_t* cols, unsigned c, uint64_t key) {
  uint64_t res = 0;
  for (unsigned b = 0; b < c && b < 64; ++b)
    if ((key >> b) & 1) res ^= cols[c - 1 - b];
  return res;
}

// jf/include/jellyfish/mer_dna.hpp:46-63: A/a=0 C/c=1 G/g=2 T/t=3, everything else negative.
inline int jf_code(unsigned char ch) {
  switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

struct Counter {
  int k = 0;
  bool canonical = true;
  std::vector<uint64_t> mers;           // every k-mer instance (sorted + run-length encoded at finish)
  std::vector<uint64_t> keys, vals, pos;  // finished records in (pos,key) order
  uint64_t cas_total = 0;                // k-mer instances seen by orc_count_cas
  // rolling state persists across add_seq() calls only inside one read
  void add_read(const char* s, size_t n) {
    // jf/include/jellyfish/mer_iterator.hpp:61-88 -- shift in valid codes, reset on anything else;
    // canonical representative = numeric min(fwd, revcomp) (:59).
    const uint64_t mask = k == 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
    uint64_t fwd = 0, rc = 0;
    int filled = 0;
    for (size_t i = 0; i < n; ++i) {
      int code = jf_code((unsigned char)s[i]);
      if (code < 0) { filled = 0; continue; }
      fwd = ((fwd << 2) | (uint64_t)code) & mask;
      rc = (rc >> 2) | ((uint64_t)(3 - code) << (2 * (k - 1)));
      if (filled < k) ++filled;
      if (filled >= k) mers.push_back(canonical && rc < fwd ? rc : fwd);
    }
  }
};

// One line of text [b,e) without its '\n'.
inline const char* next_line(const char* p, const char* end, const char*& b, const char*& e) {
  b = p;
  const char* nl = (const char*)memchr(p, '\n', end - p);
  if (!nl) { e = end; return end; }
  e = nl;
  return nl + 1;
}

}  // namespace

extern "C" {

// ---- matrix ---------------------------------------------------------------------------------
// jf/include/jellyfish/large_hash_array.hpp:942-950: RectangularBinaryMatrix(ceilLog2(size),
// key_len).randomize_pseudo_inverse() -- the hash matrix is the *pseudo-inverse* of the first
// invertible random draw (rectangular_binary_matrix.cc:209-216); glibc random() is never seeded
// (== srandom(1)).
int orc_jf_matrix(int r, int c, uint64_t* cols_out) {
  if (r <= 0 || r > 64 || c <= 0 || r > c) return -1;
  srandom(1);
  const uint64_t cmask = ~(uint64_t)0 >> (64 - r);
  std::vector<uint64_t> m(c), inv;
  for (;;) {
    for (int i = 0; i < c; ++i) m[i] = jf_random_bits64() & cmask;
    if (jf_pseudo_inverse(m, r, c, inv)) break;
  }
  memcpy(cols_out, inv.data(), sizeof(uint64_t) * c);
  return 0;
}

uint64_t orc_jf_times(const uint64_t* cols, int c, uint64_t key) { return jf_times(cols, c, key); }

// ---- count ----------------------------------------------------------------------------------
void* orc_count_new(int k, int canonical) {
  if (k < 1 || k > 32) return nullptr;
  Counter* c = new Counter;
  c->k = k;
  c->canonical = canonical != 0;
  return c;
}
void orc_count_free(void* h) { delete (Counter*)h; }

// One read's bases.  Reads never share a k-mer: the parser puts an 'N' between them
// (jf/include/jellyfish/mer_overlap_sequence_parser.hpp:195).
void orc_count_add_read(void* h, const char* seq, size_t n) { ((Counter*)h)->add_read(seq, n); }

// Whole FASTA/FASTQ text (mer_overlap_sequence_parser.hpp:124-153 type sniff, :155-177 FASTA,
// :179-206 FASTQ with :231-251 skip_quals: as many quality characters as sequence characters,
// both possibly spread over several lines; blank lines are skipped).
// Returns the number of reads, or -1 on a malformed file.
long orc_count_add_text(void* h, const char* buf, size_t n) {
  Counter* c = (Counter*)h;
  const char *p = buf, *end = buf + n, *b, *e;
  if (p == end) return 0;
  long reads = 0;
  std::string seq;
  if (*p == '>') {
    while (p < end) {
      p = next_line(p, end, b, e);  // header
      seq.clear();
      while (p < end && *p != '>') {
        p = next_line(p, end, b, e);
        seq.append(b, e);
      }
      c->add_read(seq.data(), seq.size());
      ++reads;
    }
    return reads;
  }
  if (*p != '@') return -1;
  while (p < end) {
    if (*p == '\n') { ++p; continue; }
    if (*p != '@') return -1;
    p = next_line(p, end, b, e);  // header
    seq.clear();
    while (p < end && *p != '+') {
      p = next_line(p, end, b, e);
      seq.append(b, e);
    }
    if (p < end) p = next_line(p, end, b, e);  // '+' line
    size_t quals = 0;
    while (p < end && quals < seq.size()) {
      p = next_line(p, end, b, e);
      quals += (size_t)(e - b);
    }
    if (quals != seq.size()) return -1;
    c->add_read(seq.data(), seq.size());
    ++reads;
  }
  return reads;
}

// Sort + run-length encode, keep lower <= count <= upper (applied at output,
// jf/sub_commands/count_main.cc:318-324), order records by (pos, key) with pos = (M*key) &
// (2^lsize - 1) (jf/include/jellyfish/mer_heap.hpp:34-38, sorted_dumper.hpp:80-112).
// Returns the number of records.
size_t orc_count_finish(void* h, int lsize, const uint64_t* cols, uint64_t lower, uint64_t upper) {
  Counter* c = (Counter*)h;
  std::sort(c->mers.begin(), c->mers.end());
  std::vector<uint64_t> k, v;
  for (size_t i = 0; i < c->mers.size();) {
    size_t j = i;
    while (j < c->mers.size() && c->mers[j] == c->mers[i]) ++j;
    uint64_t cnt = j - i;
    if (cnt >= lower && cnt <= upper) { k.push_back(c->mers[i]); v.push_back(cnt); }
    i = j;
  }
  const uint64_t pmask = lsize >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << lsize) - 1);
  std::vector<uint64_t> pos(k.size());
  for (size_t i = 0; i < k.size(); ++i) pos[i] = jf_times(cols, 2 * c->k, k[i]) & pmask;
  std::vector<size_t> idx(k.size());
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
    return pos[a] != pos[b] ? pos[a] < pos[b] : k[a] < k[b];
  });
  c->keys.resize(k.size()); c->vals.resize(k.size()); c->pos.resize(k.size());
  for (size_t i = 0; i < idx.size(); ++i) { c->keys[i] = k[idx[i]]; c->vals[i] = v[idx[i]]; c->pos[i] = pos[idx[i]]; }
  return c->keys.size();
}
uint64_t orc_count_total(void* h) { return ((Counter*)h)->mers.size() + ((Counter*)h)->cas_total; }
void orc_count_get(void* h, uint64_t* keys, uint64_t* vals, uint64_t* pos) {
  Counter* c = (Counter*)h;
  if (keys) memcpy(keys, c->keys.data(), 8 * c->keys.size());
  if (vals) memcpy(vals, c->vals.data(), 8 * c->vals.size());
  if (pos) memcpy(pos, c->pos.data(), 8 * c->pos.size());
}

// ---- count, the way jellyfish does it: one lock-free hash table shared by all threads ------------
// CPU-baseline port of the insert loop (jf/sub_commands/count_main.cc:148-180: every thread iterates its
// share of the reads and calls ary.add(mer, 1)) and of the table (jf/include/jellyfish/
// large_hash_array.hpp:298-302 add, :513-601 claim_key with quadratic reprobes i(i+1)/2, :708-723 CAS
// set_key, :733-744 CAS add_val).  Simplification: slots hold the whole key and a 32-bit count (jellyfish
// packs a key remainder + 7-bit value with overflow entries) -- same one-random-CAS-per-k-mer access
// pattern, same results.  seq = n_reads rows of L bases.  Fills the Counter's sorted records like
// orc_count_finish.  Returns the number of records, or (size_t)-1 when the table is too small.
size_t orc_count_cas(void* h, const char* seq, size_t n_reads, size_t L, int table_bits, int threads, int lsize,
                     const uint64_t* cols, uint64_t lower, uint64_t upper) {
  Counter* c = (Counter*)h;
  const int k = c->k;
  const uint64_t slots = (uint64_t)1 << table_bits, smask = slots - 1, EMPTY = ~(uint64_t)0;
  const uint64_t kmask = k == 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
  std::vector<uint64_t> tk(slots, EMPTY);
  std::vector<uint32_t> tv(slots, 0);
  uint64_t* keys = tk.data();
  uint32_t* vals = tv.data();
  int failed = 0;
  uint64_t total = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1024) reduction(+ : total) reduction(| : failed)
  for (long r = 0; r < (long)n_reads; ++r) {
    const char* s = seq + (size_t)r * L;
    uint64_t fwd = 0, rc = 0;
    int filled = 0;
    for (size_t i = 0; i < L; ++i) {
      const int code = jf_code((unsigned char)s[i]);
      if (code < 0) { filled = 0; continue; }  // mer_iterator.hpp:61-88
      fwd = ((fwd << 2) | (uint64_t)code) & kmask;
      rc = (rc >> 2) | ((uint64_t)(3 - code) << (2 * (k - 1)));
      if (++filled < k) continue;
      const uint64_t key = c->canonical ? std::min(fwd, rc) : fwd;
      uint64_t slot = jf_times(cols, 2 * k, key) & smask;  // hash = M * key (large_hash_array.hpp:298-302)
      bool done = false;
      for (uint64_t i2 = 0; i2 <= 126 && !done; ++i2) {   // max_reprobe 126 (count_main_cmdline.hpp:361-369)
        const uint64_t at = (slot + (i2 ? i2 * (i2 + 1) / 2 : 0)) & smask;
        uint64_t cur = __atomic_load_n(&keys[at], __ATOMIC_RELAXED);
        if (cur == EMPTY) {
          uint64_t expect = EMPTY;
          if (__atomic_compare_exchange_n(&keys[at], &expect, key, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) cur = key;
          else cur = expect;
        }
        if (cur == key) {
          __atomic_fetch_add(&vals[at], 1u, __ATOMIC_RELAXED);
          done = true;
        }
      }
      if (!done) failed = 1;
      ++total;
    }
  }
  if (failed) return (size_t)-1;
  const uint64_t pmask = lsize >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << lsize) - 1);
  struct Rec { uint64_t pos, key, val; };
  std::vector<Rec> recs;
  for (uint64_t i = 0; i < slots; ++i)
    if (keys[i] != EMPTY && vals[i] >= lower && vals[i] <= upper)
      recs.push_back({jf_times(cols, 2 * k, keys[i]) & pmask, keys[i], vals[i]});
  std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.pos != b.pos ? a.pos < b.pos : a.key < b.key; });
  c->keys.resize(recs.size()); c->vals.resize(recs.size()); c->pos.resize(recs.size());
  for (size_t i = 0; i < recs.size(); ++i) { c->keys[i] = recs[i].key; c->vals[i] = recs[i].val; c->pos[i] = recs[i].pos; }
  c->mers.clear();
  c->cas_total = total;
  return recs.size();
}

// ---- set difference -------------------------------------------------------------------------
// RUFUS's modified merge (jf/jellyfish/merge_files.cc:69-155) as a k-way merge of (pos,key)-sorted
// record arrays: a key held by exactly one input with count >= min_count is emitted, in merged order.
// Returns the number emitted (out arrays sized by the caller to the sum of the inputs).
size_t orc_merge_unique(int n_files, const uint64_t* const* keys, const uint64_t* const* vals,
                        const uint64_t* const* pos, const size_t* n, uint64_t min_count, uint64_t* out_keys,
                        uint64_t* out_vals, int* out_file) {
  std::vector<size_t> cur(n_files, 0);
  size_t o = 0;
  for (;;) {
    int best = -1;
    for (int f = 0; f < n_files; ++f) {
      if (cur[f] >= n[f]) continue;
      if (best < 0 || pos[f][cur[f]] < pos[best][cur[best]] ||
          (pos[f][cur[f]] == pos[best][cur[best]] && keys[f][cur[f]] < keys[best][cur[best]]))
        best = f;
    }
    if (best < 0) break;
    const uint64_t key = keys[best][cur[best]], val = vals[best][cur[best]];
    int holders = 0;
    for (int f = 0; f < n_files; ++f)
      if (cur[f] < n[f] && keys[f][cur[f]] == key) { ++holders; ++cur[f]; }
    if (holders == 1 && val >= min_count) {
      out_keys[o] = key;
      out_vals[o] = val;
      if (out_file) out_file[o] = best;
      ++o;
    }
  }
  return o;
}

// ---- RUFUS-side 2-bit codec -----------------------------------------------------------------
// src/Util.cpp:51-84 HashToLong: base i -> bits (2i, 2i+1); A=(0,0) C=(0,1) G=(1,0) T=(1,1),
// i.e. value A0 G1 C2 T3 at shift 2i; any other character leaves 00.
uint64_t orc_hash_to_long(const char* s, size_t n) {
  uint64_t v = 0;
  for (size_t i = 0; i < n && i < 32; ++i) {
    uint64_t lo = 0, hi = 0;
    switch (s[i]) {
      case 'C': hi = 1; break;
      case 'G': lo = 1; break;
      case 'T': lo = hi = 1; break;
      default: break;
    }
    v |= lo << (2 * i) | hi << (2 * i + 1);
  }
  return v;
}

// src/Util.cpp:187-210 RevComp: ACGTN complemented, anything else DROPPED.  out must hold n bytes.
size_t orc_revcomp(const char* s, size_t n, char* out) {
  size_t m = 0;
  for (size_t i = n; i-- > 0;) {
    switch (s[i]) {
      case 'A': out[m++] = 'T'; break;
      case 'C': out[m++] = 'G'; break;
      case 'G': out[m++] = 'C'; break;
      case 'T': out[m++] = 'A'; break;
      case 'N': out[m++] = 'N'; break;
      default: break;
    }
  }
  return m;
}

// ---- RUFUS.Filter ---------------------------------------------------------------------------
// Hash-list loader, src/RUFUS.Filter.cpp:121-143: split on ' '; 2 fields -> field 0, 4 fields ->
// field 3, 1 field -> re-split on TAB, field 0; both HashToLong(kmer) and HashToLong(RevComp(kmer))
// go into the set.  single_end selects src/RUFUS.Filter.ss.cpp:99-118 (split on TAB, then ' ').
void* orc_filter_set_new(const char* text, size_t n, int single_end) {
  auto* set = new std::unordered_set<uint64_t>;
  const char *p = text, *end = text + n, *b, *e;
  std::vector<char> rc;
  while (p < end) {
    p = next_line(p, end, b, e);
    std::string line(b, e);
    auto split = [](const std::string& s, char d) {
      // Util::Split (src/Util.cpp:24-33): getline semantics -- no trailing empty token.
      std::vector<std::string> t;
      size_t i = 0;
      while (i < s.size()) {
        size_t j = s.find(d, i);
        if (j == std::string::npos) { t.push_back(s.substr(i)); i = s.size(); break; }
        t.push_back(s.substr(i, j - i));
        i = j + 1;
      }
      return t;
    };
    std::string kmer;
    bool have = false;
    const char first = single_end ? '\t' : ' ', second = single_end ? ' ' : '\t';
    std::vector<std::string> t = split(line, first);
    if (t.size() == 2) { kmer = t[0]; have = true; }
    else if (t.size() == 4) { kmer = t[3]; have = true; }
    if (t.size() == 1) {
      t = split(line, second);
      if (!t.empty()) { kmer = t[0]; have = true; }
    }
    if (!have) continue;
    set->insert(orc_hash_to_long(kmer.data(), kmer.size()));
    rc.resize(kmer.size() + 1);
    size_t m = orc_revcomp(kmer.data(), kmer.size(), rc.data());
    set->insert(orc_hash_to_long(rc.data(), m));
  }
  return set;
}
void orc_filter_set_free(void* s) { delete (std::unordered_set<uint64_t>*)s; }
size_t orc_filter_set_size(void* s) { return ((std::unordered_set<uint64_t>*)s)->size(); }

// Scan of one read, src/RUFUS.Filter.cpp:203-220 (paired: i < len-1, the last base is never
// examined) / src/RUFUS.Filter.ss.cpp:170-190 (single end: i < len).  A position is "bad" when
// qual-33 < MinQ (signed char) or the base is 'N'; a missing quality character reads as '\0'.
// Returns the number of windows found in the set.
int orc_filter_scan(void* s, const char* seq, size_t len, const char* qual, size_t qlen, int k, int minq,
                    int single_end) {
  auto* set = (std::unordered_set<uint64_t>*)s;
  if (len == 0) return 0;  // the reference wraps length()-1 here (UB); we reject instead
  size_t stop = single_end ? len : len - 1;
  int streak = 0, found = 0;
  for (size_t i = 0; i < stop; ++i) {
    int q = i < qlen ? (int)(signed char)qual[i] : 0;
    if (q - 33 < minq || seq[i] == 'N') streak = 0;
    else ++streak;
    if (streak >= k && set->count(orc_hash_to_long(seq + i - k + 1, k))) ++found;
  }
  return found;
}

// Whole paired run: 4-line FASTQ records read in lock step (src/RUFUS.Filter.cpp:162-194); a pair
// is pulled when mate1 reaches the threshold, else when mate2 does (:222-277).  pulled[] receives
// the 0-based indices of pulled pairs in input order; returns how many.
long orc_filter_pairs(void* s, const char* m1, size_t n1, const char* m2, size_t n2, int k, int minq, int thresh,
                      uint32_t* pulled, size_t cap) {
  const char *p1 = m1, *e1 = m1 + n1, *p2 = m2, *e2 = m2 + n2, *b, *e;
  long n = 0;
  uint32_t idx = 0;
  while (p1 < e1) {
    const char *s1b, *s1e, *q1b, *q1e, *s2b = nullptr, *s2e = nullptr, *q2b = nullptr, *q2e = nullptr;
    p1 = next_line(p1, e1, b, e);
    p1 = next_line(p1, e1, s1b, s1e);
    p1 = next_line(p1, e1, b, e);
    p1 = next_line(p1, e1, q1b, q1e);
    if (p2 < e2) {
      p2 = next_line(p2, e2, b, e);
      p2 = next_line(p2, e2, s2b, s2e);
      p2 = next_line(p2, e2, b, e);
      p2 = next_line(p2, e2, q2b, q2e);
    }
    bool hit = orc_filter_scan(s, s1b, s1e - s1b, q1b, q1e - q1b, k, minq, 0) >= thresh;
    if (!hit && s2b) hit = orc_filter_scan(s, s2b, s2e - s2b, q2b, q2e - q2b, k, minq, 0) >= thresh;
    if (hit) {
      if ((size_t)n < cap) pulled[n] = idx;
      ++n;
    }
    ++idx;
  }
  return n;
}

}  // extern "C"
