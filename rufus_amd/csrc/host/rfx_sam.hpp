// SAM line plumbing shared by the feeders (pass_through_main.cpp) and `RUFUS.Filter --sam` (rufus_filter_main.cpp):
// field split, QNAME hash, the reverse-complement / reverse of a reverse-strand record as
// src/PassThroughSamCheck.stranded.cpp:188-223 prints it, the table of reads that wait for their mate.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "rfx_cli.hpp"

namespace rfxsam {

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

[[maybe_unused]] int sam_flag(const Field& fl) {  // atoi of the field: optional sign, digits, anything after them ignored
  int flag = 0;
  size_t i = 0;
  bool neg = false;
  while (i < fl.n && (fl.p[i] == ' ' || (fl.p[i] >= 9 && fl.p[i] <= 13))) ++i;
  if (i < fl.n && (fl.p[i] == '+' || fl.p[i] == '-')) neg = fl.p[i++] == '-';
  for (; i < fl.n && fl.p[i] >= '0' && fl.p[i] <= '9'; ++i) flag = flag * 10 + (fl.p[i] - '0');
  return neg ? -flag : flag;
}


}  // namespace rfxsam
