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

// The reference's cross-candidate rules over the device's per-candidate results (5 ints each, rufus_hip.h).
inline AlignResult pick_best(const int* out, const std::vector<int>& idx, bool& perfect, int k_init, int index_init) {
  AlignResult res;
  res.overlap = k_init;
  res.index = index_init;
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

inline bool acgtn_only(const std::string& s) {
  for (char c : s)
    if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') return false;
  return true;
}

// The read pool on the device (uploaded once; a merge patches one entry) and the two Align3 calls of a greedy
// step: the query (pool entry i, == `a`) forward against `fwd_idx`, and -- unless the forward pass found a perfect
// match -- reverse-complemented against `rev_idx`.  When both lists are the same (OverlapSam, OverlapRegion) the
// two strands are scored in ONE launch; the reverse complement is built on the device when the query has only
// ACGTN (Util::RevComp drops other characters: then the host string goes up instead).
struct PoolScorer {
  rfx_ovl_pool* pool = nullptr;
  std::vector<int> out;
  void create(rfx_ctx* ctx, const std::vector<std::string>& seqs) {
    std::vector<const char*> ptr(seqs.size());
    std::vector<int> len(seqs.size());
    for (size_t i = 0; i < seqs.size(); ++i) {
      ptr[i] = seqs[i].data();
      len[i] = (int)seqs[i].size();
    }
    pool = rfx_ovl_pool_create(ctx, ptr.data(), len.data(), (int)seqs.size());
    if (!pool) rfxcli::die(std::string("rufus_amd: cannot build the device read pool: ") + rfx_last_error());
  }
  void set(int idx, const std::string& s) {
    const int rc = rfx_ovl_pool_set(pool, idx, s.data(), (int)s.size());
    if (rc) rfxcli::die(std::string("rufus_amd: pool update failed: ") + rfx_strerror(rc) + " " + rfx_last_error());
  }
  void score(int query, const std::string* explicit_a, const std::vector<int>& idx, float min_pct, int min_ovl, int variant,
             int strands) {
    out.assign(idx.size() * 5 * (strands == 2 ? 2 : 1) + 1, 0);
    if (idx.empty()) return;
    const int rc = rfx_ovl_pool_score(pool, query, explicit_a ? explicit_a->data() : nullptr,
                                      explicit_a ? (int)explicit_a->size() : 0, idx.data(), (int)idx.size(), min_pct, min_ovl,
                                      variant, strands, out.data());
    if (rc) rfxcli::die(std::string("rufus_amd: overlap scoring failed: ") + rfx_strerror(rc) + " " + rfx_last_error());
  }
  // best forward partner; then, if nothing was perfect, the best partner of the reverse complement (in `rev`;
  // rev_done says whether it was looked for).  fwd_init / rev_init: the variant's initial (overlap, index).
  AlignResult both(int query, const std::string& a, const std::vector<int>& fwd_idx, const std::vector<int>& rev_idx,
                   bool same_lists, float min_pct, int min_ovl, int variant, int fwd_k_init, AlignResult& rev, bool& rev_done) {
    bool perfect = false;
    const bool clean = acgtn_only(a);
    rev_done = false;
    if (same_lists && clean) {
      score(query, nullptr, fwd_idx, min_pct, min_ovl, variant, 2);
      AlignResult best = pick_best(out.data(), fwd_idx, perfect, fwd_k_init, -1);
      if (!perfect) {
        rev = pick_best(out.data() + 5 * fwd_idx.size(), fwd_idx, perfect, -1, -1);
        rev_done = true;
      }
      return best;
    }
    score(query, nullptr, fwd_idx, min_pct, min_ovl, variant, 0);
    AlignResult best = pick_best(out.data(), fwd_idx, perfect, fwd_k_init, -1);
    if (!perfect) {
      if (clean) {
        score(query, nullptr, rev_idx, min_pct, min_ovl, variant, 1);
      } else {
        const std::string ra = revcomp(a);
        score(query, &ra, rev_idx, min_pct, min_ovl, variant, 0);
      }
      rev = pick_best(out.data(), rev_idx, perfect, -1, -1);
      rev_done = true;
    }
    return best;
  }
  void release() {
    rfx_ovl_pool_free(pool);
    pool = nullptr;
  }
  ~PoolScorer() { release(); }
};

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
