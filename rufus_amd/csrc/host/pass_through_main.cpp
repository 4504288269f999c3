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

#include "rfx_cli.hpp"

#ifndef PTS_MODE
#define PTS_MODE 0
#endif

namespace {

struct Field {
  const char* p;
  size_t n;
};

// Fields 1..11 of a SAM line located by counting TABs.
bool split_sam(const char* b, const char* e, Field f[11]) {
  const char* p = b;
  for (int i = 0; i < 11; ++i) {
    const char* t = (const char*)memchr(p, '\t', (size_t)(e - p));
    f[i].p = p;
    if (!t) {
      f[i].n = (size_t)(e - p);
      return i == 10;  // only the last field may end at end of line
    }
    f[i].n = (size_t)(t - p);
    p = t + 1;
  }
  return true;
}

[[maybe_unused]] void revcomp_into(std::string& out, const Field& s) {
  out.clear();
  for (size_t j = s.n; j-- > 0;) {
    switch (s.p[j]) {
      case 'A': out += 'T'; break;
      case 'C': out += 'G'; break;
      case 'G': out += 'C'; break;
      case 'T': out += 'A'; break;
      case 'N': out += 'N'; break;
      default: break;  // dropped, as the reference's switch without default does
    }
  }
}

[[maybe_unused]] void reverse_into(std::string& out, const Field& s) {
  out.assign(s.p, s.n);
  for (size_t i = 0, j = out.size(); i + 1 < j; ++i, --j) std::swap(out[i], out[j - 1]);
}

void put_record(FILE* f, const Field& name, const char* seq, size_t ls, const char* qual, size_t lq) {
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
  // Two named pipes read in lock step by RUFUS.Filter: keep both sides moving record by record.
  setvbuf(m1, nullptr, _IOFBF, 1 << 16);
  setvbuf(m2, nullptr, _IOFBF, 1 << 16);
  std::unordered_map<std::string, std::pair<std::string, std::string>> waiting;
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
    const bool reverse = (atoi(std::string(f[1].p, f[1].n).c_str()) & 16) != 0;
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
    const std::string name(f[0].p, f[0].n);
    auto it = waiting.find(name);
    if (it == waiting.end()) {
      waiting.emplace(name, std::make_pair(std::string(sp, sn), std::string(qp, qn)));
    } else {
      put_record(m1, f[0], sp, sn, qp, qn);
      put_record(m2, f[0], it->second.first.data(), it->second.first.size(), it->second.second.data(),
                 it->second.second.size());
      fflush(m1);  // the consumer alternates between the two pipes
      fflush(m2);
      waiting.erase(it);
    }
#endif
#endif
  }
  fprintf(chr, "%s\n", current.c_str());
  fclose(chr);
#if PTS_MODE == 1
  fclose(m1);
  fclose(m2);
#endif
  return 0;
}
