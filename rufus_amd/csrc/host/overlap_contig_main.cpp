// Drop-in Overlap and OverlapRegion (row G6, -DREGION), same argv and output files:
//   Overlap       FASTQD MinPercent MinOverlap MinCoverage ReportStub SearchHash ACT OutStub LCtrim Threads
//       -> OutStub.fastq, OutStub.fastqd, OutStub.fastqgood.fastq, OutStub.fastqbad.fastq
//       src/Overlap.cpp:614-647 (argv), :717-762 (FASTQD intake), :34-167 (seed index + candidate
//       ranking), :870-1126 (buffered greedy loop), :1132-1174 (output)      scripts/Overlap.shorter.sh:141-165
//   OverlapRegion FASTQD MinPercent MinOverlap MinCoverage OutStub NodeStub LCcut Threads
//       -> OutStub.fastq, OutStub.fastqd      src/OverlapRegion.cpp:502-539, :588-652, :702-841, :846-903
// Greedy order, seed index and merging stay on the host (sequential in the reference); every
// pairwise alignment is scored by rfx_overlap_score on the device.  Only the .fastqd input route the
// pipeline uses is provided.  Reference quirks kept on purpose: depths are compared as SIGNED chars
// when trimming low-coverage ends (128..250 read as negative); the seed index is never updated after
// a merge (`if (found = false)`, Overlap.cpp:1088,:1105); candidate lists of a whole buffer of
// 100*Threads reads are ranked before any of its merges; Overlap names nodes after argv[6] (the seed
// size).  Output equals the reference at the same Threads value, up to its OpenMP races (Threads = 1
// is deterministic) and the 100 000-seed-hit cap (order dependent there).
#include <map>
#include <unordered_map>

#include "overlap_common.hpp"

using namespace ovl;
using rfxcli::die;

namespace {

#ifndef REGION
// Util::HashToLong (src/Util.cpp:51-84) -- only used as the key of the host-side seed index
unsigned long seed_key(const char* s, int n) {
  unsigned long v = 0;
  for (int i = 0; i < n && i < 32; ++i) {
    unsigned long lo = 0, hi = 0;
    if (s[i] == 'C') hi = 1;
    else if (s[i] == 'G') lo = 1;
    else if (s[i] == 'T') lo = hi = 1;
    v |= lo << (2 * i) | hi << (2 * i + 1);
  }
  return v;
}
#endif

// TrimLowCoverageEnds: drop both ends up to the first base whose depth, read as a signed char,
// exceeds the cutoff.  Overlap.cpp:510-552 returns "" when one base or none survives the first pass,
// OverlapRegion.cpp:385-426 returns that remainder.
std::string trim_low_cov(const std::string& s, std::string& q, std::string& d, int cutoff, bool region_flavour) {
  size_t hi = s.size();
  d.resize(s.size(), '\0');  // a missing depth reads as 0
  while (hi > 0 && !((int)(signed char)d[hi - 1] > cutoff)) --hi;
  std::string s1 = s.substr(0, hi), q1, d1 = d.substr(0, std::min(hi, d.size()));
  for (size_t i = 0; i < hi; ++i) q1 += i < q.size() ? q[i] : '\0';
  if (s1.size() <= 1) {
    if (region_flavour) { q = q1; d = d1; return s1; }
    q.clear(); d.clear();
    return "";
  }
  size_t lo = 0;
  while (lo < s1.size() && !((int)(signed char)d1[lo] > cutoff)) ++lo;
  q = q1.substr(lo);
  d = d1.substr(lo);
  return s1.substr(lo);
}

struct Pool {
  std::vector<std::string> seq, qual, depth, strand;
};

void write_nodes(const Pool& p, const std::string& stub, const std::string& node, int min_cov) {
  std::ofstream report((stub + ".fastq").c_str()), dep((stub + ".fastqd").c_str());
  int count = 0;
  for (size_t i = 0; i < p.seq.size(); ++i) {
    if (p.seq[i] == "moved" || p.seq[i].size() < 95) continue;
    int max_dep = -1;
    for (char c : p.depth[i]) max_dep = std::max(max_dep, (int)(unsigned char)c);
    if (max_dep < min_cov) continue;
    ++count;
    int f = 0, r = 0;
    strand_counts(p.strand[i], f, r);
    std::ostringstream h;
    h << "@NODE_" << node << "_" << i << "_L" << p.seq[i].size() << "_D" << max_dep << ":" << f << ":" << r << ":";
    report << h.str() << '\n' << p.seq[i] << "\n+\n" << p.qual[i] << '\n';
    dep << h.str() << '\n' << p.seq[i] << "\n+\n" << p.qual[i] << '\n' << p.strand[i] << '\n';
    write_depths(dep, p.depth[i]);
  }
  std::cout << "\nWrote " << count << " sequences" << std::endl;
}

}  // namespace

int main(int argc, char** argv) {
  std::cout << "you gave " << argc << " Arguments" << std::endl;
#ifdef REGION
  if (argc != 9) {
    std::cout << "ERROR, wrong numbe of arguemnts\nCall is: FASTQ, MinPercent, MinOverlap, MinCoverage, ReportStub, "
                 "NodeStub LCcutoff Threads" << std::endl;
    return 0;
  }
  const std::string stub = argv[5], node = argv[6];
  const int lc_cut = atoi(argv[7]);
#else
  if (argc != 11) {
    std::cout << "ERROR, wrong numbe of arguemnts\nCall is: FASTQ, MinPercent, MinOverlap, MinCoverage, ReportStub, "
                 "SearchHashSize, ACT, OutFile LCendTrimEpth Threads" << std::endl;
    return 0;
  }
  const std::string stub = argv[8], node = argv[6];
  const int search = atoi(argv[6]), act = atoi(argv[7]), lc_cut = atoi(argv[9]);
  const int threads = std::max(1, atoi(argv[10]));
#endif
  const float min_pct = (float)atof(argv[2]);
  const int min_ovl = atoi(argv[3]), min_cov = atoi(argv[4]);
  std::ifstream in(argv[1]);
  if (!in.is_open()) {
    std::cout << "Error, ParentHashFile could not be opened";
    return 0;
  }
  if (std::string(argv[1]).find(".fastqd") == std::string::npos)
    die("rufus_amd: only the .fastqd input route of the pipeline is provided");

  // ---- intake -----------------------------------------------------------------------------------
  Pool p;
#ifndef REGION
  std::ofstream good((stub + ".fastqgood.fastq").c_str()), bad((stub + ".fastqbad.fastq").c_str());
#endif
  std::string l[6];
  while (std::getline(in, l[0])) {
    for (int i = 1; i < 6; ++i)
      if (!std::getline(in, l[i])) l[i].clear();
    const std::vector<std::string> toks = split(l[5], ' ');
#ifdef REGION
    // validateFASTQD (src/OverlapRegion.cpp:470-489)
    if (l[0].empty() || l[0][0] != '@' || l[1].size() != l[3].size() || toks.size() != l[1].size()) {
      std::cout << "ERROR in FASTQD file \n\t " << l[0] << std::endl;
      return 1;
    }
#endif
    std::string depths;
    bool multiple = false;
    for (const std::string& t : toks) {
      const unsigned char c = (unsigned char)atoi(t.c_str());
      depths += (char)c;
      if (c > 1) multiple = true;
    }
    std::string s = l[1], q = l[3];
#ifdef REGION
    if (multiple) s = trim_low_cov(s, q, depths, lc_cut, true);
    if (s.size() > 90) {
#else
    if (multiple) s = trim_low_cov(s, q, depths, lc_cut, false);
    if (s.size() > (size_t)(search + 1)) {
#endif
      p.seq.push_back(s);
      p.qual.push_back(q);
      p.depth.push_back(depths);
      p.strand.push_back(l[4]);
    } else {
#ifndef REGION
      bad << l[0] << '\n' << s << '\n' << l[2] << '\n' << q << '\n';
#endif
    }
  }
#ifndef REGION
  good.close();
  bad.close();
#endif
  const int n = (int)p.seq.size();
  rfx_ctx* ctx = n ? rfxcli::open_ctx() : nullptr;
  PoolScorer scorer;  // the reads live on the device; a merge patches the one entry it changes
  if (n) scorer.create(ctx, p.seq);

#ifdef REGION
  // ---- OverlapRegion: every later read is a candidate (src/OverlapRegion.cpp:37-41) ----------------
  for (int i = 0; i < n; ++i) {
    std::string a = p.seq[(size_t)i], aq = p.qual[(size_t)i], ad = p.depth[(size_t)i], as = p.strand[(size_t)i];
    std::vector<int> idx;
    for (int j = i + 1; j < n; ++j) idx.push_back(j);
    AlignResult rev;
    bool rev_done = false;
    AlignResult best = scorer.both(i, a, idx, idx, true, min_pct, min_ovl, RFX_OVL_REGION, -1, rev, rev_done);
    if (rev_done && rev.score > best.score) {
      a = revcomp(a); aq = revqual(aq); ad = revqual(ad); as = flip_strands(as);
      best = rev;
    }
    if (best.score < min_ovl) continue;
    const size_t bi = (size_t)best.index;
    std::string bq = p.qual[bi], bd = p.depth[bi], bs = p.strand[bi];
    p.seq[bi] = collapse(a, p.seq[bi], best.overlap, aq, bq, ad, bd, as, bs, MERGE_REGION);
    p.qual[bi] = bq; p.depth[bi] = bd; p.strand[bi] = bs;
    scorer.set((int)bi, p.seq[bi]);
    p.seq[(size_t)i] = "moved";
  }
#else
  // ---- Overlap: seed index built once (RebuildHashTable :34-76; rebuilt only every 10^6 reads) -------
  std::unordered_map<unsigned long, std::vector<int>> index;
  auto build_index = [&](int from) {
    index.clear();
    for (int i = from; i < n; ++i) {
      const std::string& s = p.seq[(size_t)i];
      for (int j = 0; j + search < (int)s.size(); ++j) {
        if (memchr(s.data() + j, 'N', (size_t)search)) continue;
        const std::string seed = s.substr((size_t)j, (size_t)search);
        index[seed_key(seed.data(), search)].push_back(i);
        const std::string rc = revcomp(seed);
        index[seed_key(rc.data(), (int)rc.size())].push_back(i);
      }
    }
  };
  // PrepairSearchList (:78-167): later reads ranked by the number of shared seeds
  auto rank = [&](const std::string& a, int ai) {
    std::map<int, int> positions;
    int added = 0;
    for (int i = 0; i + search < (int)a.size(); ++i) {
      if (memchr(a.data() + i, 'N', (size_t)search)) continue;
      auto it = index.find(seed_key(a.data() + i, search));
      if (it == index.end()) continue;
      for (int holder : it->second) {
        if (holder > ai) {
          ++positions[holder];
          ++added;
        }
        if (added > 100000) break;
      }
    }
    std::multimap<int, int> sorted;
    for (auto& kv : positions)
      if (kv.second > act) sorted.insert(std::make_pair(kv.second, kv.first));
    std::vector<int> out;
    int sanity = 0;
    for (auto it = sorted.rbegin(); it != sorted.rend(); ++it) {
      if (it->first >= act) {
        out.push_back(it->second);
        if (++sanity > 1000) break;
      }
    }
    return out;
  };
  build_index(0);
  const int buffer = 100 * threads;
  long since_build = 1;
  for (int b = 0; b < n; b += buffer) {
    since_build += buffer;
    if (since_build > 1000000) {
      build_index(b);
      since_build = 0;
    }
    const int hi = std::min(n, b + buffer);
    std::vector<std::vector<int>> fwd((size_t)(hi - b)), rev((size_t)(hi - b));
    for (int i = b; i < hi; ++i) fwd[(size_t)(i - b)] = rank(p.seq[(size_t)i], i);
    for (int i = b; i < hi; ++i) rev[(size_t)(i - b)] = rank(revcomp(p.seq[(size_t)i]), i);
    for (int i = b; i < hi; ++i) {
      std::string a = p.seq[(size_t)i], aq = p.qual[(size_t)i], ad = p.depth[(size_t)i], as = p.strand[(size_t)i];
      AlignResult r2;
      bool rev_done = false;
      AlignResult best = scorer.both(i, a, fwd[(size_t)(i - b)], rev[(size_t)(i - b)], false, min_pct, min_ovl,
                                     RFX_OVL_CONTIG, 0, r2, rev_done);
      if (rev_done && r2.score > best.score) {
        a = revcomp(a); aq = revqual(aq); ad = revqual(ad); as = flip_strands(as);
        best = r2;
      }
      if (best.score < min_ovl) continue;
      const size_t bi = (size_t)best.index;
      std::string bq = p.qual[bi], bd = p.depth[bi], bs = p.strand[bi];
      p.seq[bi] = collapse(a, p.seq[bi], best.overlap, aq, bq, ad, bd, as, bs, MERGE_CONTIG);
      p.qual[bi] = bq; p.depth[bi] = bd; p.strand[bi] = bs;
      scorer.set((int)bi, p.seq[bi]);
      p.seq[(size_t)i] = "moved";
    }
  }
#endif
  write_nodes(p, stub, node, min_cov);
  scorer.release();  // before the context goes
  if (ctx) rfx_close(ctx);
  return 0;
}
