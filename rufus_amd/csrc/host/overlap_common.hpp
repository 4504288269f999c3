// Shared host logic of the drop-in assemblers (OverlapSam / Overlap / OverlapRegion) and tail tools.
// The greedy merge order is sequential in the reference and stays on the host; candidate scoring
// (Align3) and mutant-k-mer tagging run on the device through the C-ABI.
#pragma once
#include <algorithm>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "rfx_cli.hpp"

namespace ovl {

// Util::Split (src/Util.cpp:24-33): std::getline tokens, no trailing empty token.
inline std::vector<std::string> split(const std::string& s, char d) {
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

// Util::RevComp (src/Util.cpp:187-210): ACGTN complemented, anything else dropped.
inline std::string revcomp(const std::string& s) {
  std::string o;
  o.reserve(s.size());
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

// Util::RevQual (src/Util.cpp:212-222): reversed, NUL characters dropped.
inline std::string revqual(const std::string& s) {
  std::string o;
  o.reserve(s.size());
  for (size_t i = s.size(); i-- > 0;)
    if (s[i] != '\0') o += s[i];
  return o;
}

// FlipStrands (src/OverlapSam.cpp:516-527)
inline std::string flip_strands(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '+') o += '-';
    else if (c == '-') o += '+';
    else if (c == '.') o += '.';
  }
  return o;
}

inline void strand_counts(const std::string& s, int& f, int& r) {
  for (char c : s) {
    if (c == '+') ++f;
    else if (c == '-') ++r;
  }
}

// One Align3 call of the reference: A against the candidates `idx` (in the reference's visiting
// order), with its shared PerfectMatch flag and "first strictly better wins" rule.
struct AlignResult {
  int score = 0, overlap = -1, index = -1;
};

inline AlignResult align3(rfx_ctx* ctx, const std::vector<std::string>& seqs, const std::string& a,
                          const std::vector<int>& idx, float min_pct, int min_ovl, int variant, bool& perfect,
                          int k_init, int index_init) {
  AlignResult res;
  res.overlap = k_init;
  res.index = index_init;
  if (idx.empty()) return res;
  std::vector<const char*> b(idx.size());
  std::vector<int> bl(idx.size());
  for (size_t j = 0; j < idx.size(); ++j) {
    b[j] = seqs[(size_t)idx[j]].data();
    bl[j] = (int)seqs[(size_t)idx[j]].size();
  }
  std::vector<int> out(idx.size() * 5);
  const int rc = rfx_overlap_score(ctx, a.data(), (int)a.size(), b.data(), bl.data(), (int)idx.size(), min_pct, min_ovl,
                                   variant, out.data());
  if (rc) rfxcli::die(std::string("rufus_amd: overlap scoring failed: ") + rfx_strerror(rc) + " " + rfx_last_error());
  for (size_t j = 0; j < idx.size(); ++j) {
    const int* o = &out[5 * j];
    // candidates visited after a perfect match only run phase 1 (the `if (PerfectMatch == false)` guard)
    const int score = perfect ? o[0] : o[3];
    const int ovlp = perfect ? o[1] : o[4];
    if (o[2]) perfect = true;
    if (res.score < score) {
      res.score = score;
      res.overlap = ovlp;
      res.index = idx[j];
    }
  }
  return res;
}

// ColapsContigs, three flavours (src/OverlapSam.cpp:243-357, src/Overlap.cpp:362-466,
// src/OverlapRegion.cpp:233-358).  Returns the merged sequence; bq/bd/bs are updated in place.
enum MergeRule { MERGE_SAM, MERGE_CONTIG, MERGE_REGION };

inline std::string collapse(const std::string& a, const std::string& b, int k, const std::string& aq, std::string& bq,
                            const std::string& ad, std::string& bd, const std::string& as, std::string& bs,
                            MergeRule rule) {
  const int asz = (int)a.size(), bsz = (int)b.size();
  const int aoff = k > 0 ? k : 0, boff = k > 0 ? 0 : -k;
  std::string ns, nq, nd;
  for (int i = 0; i < asz + bsz; ++i) {
    char ab = 'Z', bb = 'Z', aqc = '!', bqc = '!';
    unsigned char adp = 0, bdp = 0;
    if (i - aoff >= 0 && i - aoff < asz) {
      ab = a[(size_t)(i - aoff)];
      aqc = (size_t)(i - aoff) < aq.size() ? aq[(size_t)(i - aoff)] : '\0';
      adp = (size_t)(i - aoff) < ad.size() ? (unsigned char)ad[(size_t)(i - aoff)] : 0;
    }
    if (i - boff >= 0 && i - boff < bsz) {
      bb = b[(size_t)(i - boff)];
      bqc = (size_t)(i - boff) < bq.size() ? bq[(size_t)(i - boff)] : '\0';
      bdp = (size_t)(i - boff) < bd.size() ? (unsigned char)bd[(size_t)(i - boff)] : 0;
    }
    if (ab == bb && ab != 'Z') {
      ns += ab;
      nq += aqc >= bqc ? aqc : bqc;
      nd += (int)adp + (int)bdp < 250 ? (char)(adp + bdp) : (char)250;
    } else if (ab == 'Z' && bb != 'Z') {
      ns += bb; nq += bqc; nd += (char)bdp;
    } else if (ab != 'Z' && bb == 'Z') {
      ns += ab; nq += aqc; nd += (char)adp;
    } else if (ab != 'Z' && bb != 'Z') {
      bool take_a;
      if (rule == MERGE_CONTIG) {
        // Overlap.cpp:437-454: deeper base wins, quality breaks ties
        if (adp > bdp) take_a = true;
        else if (bdp > adp) take_a = false;
        else take_a = aqc >= bqc;
      } else {
        if (ab == 'N' && bb != 'N') take_a = false;
        else if (ab != 'N' && bb == 'N') take_a = true;
        else take_a = aqc >= bqc;
      }
      if (take_a) { ns += ab; nq += aqc; nd += (char)adp; }
      else { ns += bb; nq += bqc; nd += (char)bdp; }
    } else {
      break;  // both exhausted
    }
  }
  bq = nq;
  bd = nd;
  bs += as;
  return ns;
}

inline void write_depths(std::ostream& os, const std::string& d) {
  for (size_t w = 0; w < d.size(); ++w) {
    if (w) os << ' ';
    os << (int)(unsigned char)d[w];
  }
  os << '\n';
}

}  // namespace ovl
