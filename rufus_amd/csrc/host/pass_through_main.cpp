// Drop-in SAM -> FASTQ feeders, same argv, same output bytes:
//   PassThroughSamCheck             CHRFILE        stdin SAM -> stdout FASTQ        (src/PassThroughSamCheck.cpp:51-155)
//   PassThroughSamCheck.stranded    CHRFILE STUB   -> STUB.mate1.fastq/.mate2.fastq (src/PassThroughSamCheck.stranded.cpp:78-281)
//   PassThroughSamCheck.stranded.se CHRFILE        stdin SAM -> stdout FASTQ, reverse-strand reads restored
// Built three times from this file with -DPTS_MODE=0/1/2.  Pure host text plumbing in front of the
// count (RunJellyForRUFUS.sh:28) and filter (runRufus.sh:966) stages; buffered output instead of the
// reference's flush per line.
// Reference quirks kept: the chromosome log starts with "notachr" and records the PREVIOUS name at
// every change; QUAL ends at the next TAB (FastqToSam.pl appends one) -- here also at end of line,
// where the reference would run off the string; in the reverse-complement branch any base other
// than ACGTN disappears; the stranded tool emits a pair when its SECOND record arrives (that record
// goes to mate1, the stored one to mate2) and silently drops reads whose mate never shows up.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "rfx_cli.hpp"

#ifndef PTS_MODE
#define PTS_MODE 0
#endif

namespace {

struct Field {
  const char* p;
  size_t n;
};

// Fields 1..11 of a SAM line located by counting TABs (16 bytes per step: eleven memchr calls per line were a
// quarter of the tool's time).
bool split_sam(const char* b, const char* e, Field f[11]) {
  const char* tab[11];
  int nt = 0;
  const char* p = b;
#if RFX_X86
  const __m128i tv = _mm_set1_epi8('\t');
  while (nt < 11 && e - p >= 16) {
    unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)p), tv));
    while (m && nt < 11) {
      tab[nt++] = p + __builtin_ctz(m);
      m &= m - 1;
    }
    p += 16;
  }
#endif
  for (; nt < 11 && p < e; ++p)
    if (*p == '\t') tab[nt++] = p;
  if (nt < 10) return false;  // only the last field may end at end of line
  const char* start = b;
  for (int i = 0; i < 11; ++i) {
    const char* end = i < nt ? tab[i] : e;
    f[i].p = start;
    f[i].n = (size_t)(end - start);
    start = end + 1;
  }
  return true;
}

// Reads waiting for their mate: name -> (sequence, quality), both as the pair will print them.  An open-addressed
// table of name hashes over a pool of reusable strings (name '\0'-free, then sequence, then quality): on a
// coordinate-sorted input a few thousand reads wait at any time, and a node-based map of three std::strings per
// read spent its time in malloc.
[[maybe_unused]] uint64_t name_hash(const char* p, size_t n) {
  uint64_t h = 0xCBF29CE484222325ull ^ n;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
  }
  for (; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 0x100000001B3ull;
  return h ^ (h >> 32);
}
struct [[maybe_unused]] Waiting {
  struct Entry { std::string bytes; uint32_t name_len = 0, seq_len = 0; };
  std::vector<Entry> pool;
  std::vector<uint32_t> free_list;
  std::vector<uint32_t> slot;  // 0 empty, 1 deleted, else pool index + 2
  std::vector<uint64_t> slot_hash;
  size_t used = 0, filled = 0;  // live entries; live + deleted slots
  Waiting() : slot(1 << 12, 0), slot_hash(1 << 12, 0) {}
  void rehash(size_t n) {
    std::vector<uint32_t> os;
    std::vector<uint64_t> oh;
    os.swap(slot);
    oh.swap(slot_hash);
    slot.assign(n, 0);
    slot_hash.assign(n, 0);
    filled = used;
    for (size_t i = 0; i < os.size(); ++i)
      if (os[i] >= 2) {
        size_t j = (size_t)oh[i] & (n - 1);
        while (slot[j]) j = (j + 1) & (n - 1);
        slot[j] = os[i];
        slot_hash[j] = oh[i];
      }
  }
  // index of the slot holding `name`, or -1
  long find(uint64_t h, const char* name, size_t n) const {
    const size_t mask = slot.size() - 1;
    for (size_t j = (size_t)h & mask;; j = (j + 1) & mask) {
      if (slot[j] == 0) return -1;
      if (slot[j] >= 2 && slot_hash[j] == h) {
        const Entry& en = pool[slot[j] - 2];
        if (en.name_len == n && memcmp(en.bytes.data(), name, n) == 0) return (long)j;
      }
    }
  }
  void insert(uint64_t h, const char* name, size_t n, const char* seq, size_t sn, const char* qual, size_t qn) {
    if ((filled + 1) * 2 > slot.size()) rehash(used * 4 > slot.size() ? slot.size() * 2 : slot.size());
    uint32_t idx;
    if (!free_list.empty()) {
      idx = free_list.back();
      free_list.pop_back();
    } else {
      idx = (uint32_t)pool.size();
      pool.emplace_back();
    }
    Entry& en = pool[idx];
    en.bytes.assign(name, n);
    en.bytes.append(seq, sn);
    en.bytes.append(qual, qn);
    en.name_len = (uint32_t)n;
    en.seq_len = (uint32_t)sn;
    const size_t mask = slot.size() - 1;
    size_t j = (size_t)h & mask;
    while (slot[j] >= 2) j = (j + 1) & mask;
    if (slot[j] == 0) ++filled;
    slot[j] = idx + 2;
    slot_hash[j] = h;
    ++used;
  }
  void erase(long j) {
    free_list.push_back(slot[(size_t)j] - 2);
    slot[(size_t)j] = 1;
    --used;
  }
};

// complement of A C G T N, 0 for anything else: such a base disappears, as the reference's switch without default
// makes it (src/PassThroughSamCheck.stranded.cpp:188-196)
struct CompLut {
  unsigned char t[256];
  CompLut() {
    memset(t, 0, sizeof t);
    t['A'] = 'T'; t['C'] = 'G'; t['G'] = 'C'; t['T'] = 'A'; t['N'] = 'N';
  }
};
[[maybe_unused]] const CompLut g_comp;

[[maybe_unused]] void revcomp_into(std::string& out, const Field& s) {
  out.resize(s.n);
  char* w = &out[0];
  const unsigned char* p = (const unsigned char*)s.p;
  for (size_t j = s.n; j-- > 0;) {
    const unsigned char c = g_comp.t[p[j]];
    *w = (char)c;
    w += c != 0;
  }
  out.resize((size_t)(w - out.data()));
}

[[maybe_unused]] void reverse_into(std::string& out, const Field& s) {
  out.resize(s.n);
  for (size_t i = 0; i < s.n; ++i) out[i] = s.p[s.n - 1 - i];
}

[[maybe_unused]] void put_record(FILE* f, const Field& name, const char* seq, size_t ls, const char* qual, size_t lq) {
  fputc('@', f);
  fwrite(name.p, 1, name.n, f);
  fputc('\n', f);
  fwrite(seq, 1, ls, f);
  fputs("\n+\n", f);
  fwrite(qual, 1, lq, f);
  fputc('\n', f);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < (PTS_MODE == 1 ? 3 : 2)) {
    printf("ERROR, Output file could not be opened -%s\n", argc > 1 ? argv[1] : "");
    return 0;
  }
  FILE* chr = fopen(argv[1], "w");
  if (!chr) {
    printf("ERROR, Output file could not be opened -%s\n", argv[1]);
    return 0;
  }
#if PTS_MODE == 1
  FILE* m1 = fopen((std::string(argv[2]) + ".mate1.fastq").c_str(), "w");
  FILE* m2 = fopen((std::string(argv[2]) + ".mate2.fastq").c_str(), "w");
  if (!m1 || !m2) {
    printf("ERROR, Output file could not be opened -%s\n", argv[1]);
    return 0;
  }
  // Two named pipes read in lock step by RUFUS.Filter (the reference alternates: four lines from one, four from the
  // other).  Both sides are written in chunks that hold the SAME pairs and stay well below a pipe's 64 KB: the
  // reader can always consume everything earlier chunks brought, so neither write can wait for the other pipe.
  // (The first version flushed both pipes after every pair: two system calls per pair were most of its time.)
  setvbuf(m1, nullptr, _IONBF, 0);
  setvbuf(m2, nullptr, _IONBF, 0);
  std::string out1, out2;
  const size_t CHUNK = 24u << 10;
  auto flush_pairs = [&]() {
    if (!out1.empty()) fwrite(out1.data(), 1, out1.size(), m1);
    if (!out2.empty()) fwrite(out2.data(), 1, out2.size(), m2);
    out1.clear();
    out2.clear();
  };
  auto put_text = [](std::string& o, const char* name, size_t nn, const char* seq, size_t ls, const char* qual, size_t lq) {
    o.push_back('@');
    o.append(name, nn);
    o.push_back('\n');
    o.append(seq, ls);
    o.append("\n+\n", 3);
    o.append(qual, lq);
    o.push_back('\n');
  };
  Waiting waiting;
#endif
  rfxcli::LineReader in;
  in.attach(0);
  std::string current = "notachr", rs, rq;
  const char *b, *e;
  Field f[11];
  while (in.getline(b, e)) {
    if (!split_sam(b, e, f)) continue;  // fewer than 11 fields: the reference reads past the line here
    if (f[2].n != current.size() || memcmp(f[2].p, current.data(), f[2].n) != 0) {
      fprintf(chr, "%s\n", current.c_str());
      current.assign(f[2].p, f[2].n);
    }
#if PTS_MODE == 0
    put_record(stdout, f[0], f[9].p, f[9].n, f[10].p, f[10].n);
#else
    int flag = 0;  // atoi of the field: optional sign, digits, anything after them ignored
    {
      size_t i = 0;
      bool neg = false;
      while (i < f[1].n && (f[1].p[i] == ' ' || (f[1].p[i] >= 9 && f[1].p[i] <= 13))) ++i;
      if (i < f[1].n && (f[1].p[i] == '+' || f[1].p[i] == '-')) neg = f[1].p[i++] == '-';
      for (; i < f[1].n && f[1].p[i] >= '0' && f[1].p[i] <= '9'; ++i) flag = flag * 10 + (f[1].p[i] - '0');
      if (neg) flag = -flag;
    }
    const bool reverse = (flag & 16) != 0;
    const char *sp = f[9].p, *qp = f[10].p;
    size_t sn = f[9].n, qn = f[10].n;
    if (reverse) {
      revcomp_into(rs, f[9]);
      reverse_into(rq, f[10]);
      sp = rs.data(); sn = rs.size();
      qp = rq.data(); qn = rq.size();
    }
#if PTS_MODE == 2
    put_record(stdout, f[0], sp, sn, qp, qn);
#else
    const uint64_t nh = name_hash(f[0].p, f[0].n);
    const long at = waiting.find(nh, f[0].p, f[0].n);
    if (at < 0) {
      waiting.insert(nh, f[0].p, f[0].n, sp, sn, qp, qn);
    } else {
      const Waiting::Entry& en = waiting.pool[waiting.slot[(size_t)at] - 2];
      put_text(out1, f[0].p, f[0].n, sp, sn, qp, qn);
      put_text(out2, f[0].p, f[0].n, en.bytes.data() + en.name_len, en.seq_len,
               en.bytes.data() + en.name_len + en.seq_len, en.bytes.size() - en.name_len - en.seq_len);
      waiting.erase(at);
      if (out1.size() >= CHUNK || out2.size() >= CHUNK) flush_pairs();
    }
#endif
#endif
  }
  fprintf(chr, "%s\n", current.c_str());
  fclose(chr);
#if PTS_MODE == 1
  flush_pairs();
  fclose(m1);
  fclose(m2);
#endif
  return 0;
}
