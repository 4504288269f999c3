// Drop-in OverlapSam, same argv and output files:
//   OverlapSam SAM MinPercent MinOverlap MinCoverage FileStub NodeStub LCcutoff HashList Threads
//       -> FileStub.fastq (4-line) and FileStub.fastqd (6-line: + strand string + depths)
//   scripts/Overlap.shorter.sh:127; src/OverlapSam.cpp:564-613 (argv), :708-852 (intake),
//   :866-1024 (greedy loop), :1029-1132 (output).
// Position-sorted reads are merged greedily: read i goes into its best partner among the next 10
// (forward, then reverse complement unless a perfect match was found).  That order is inherently
// sequential and stays here; the pairwise scoring (Align3) is rfx_overlap_score on the device and
// the "does this read carry a mutant k-mer" tag (CountHashes) is rfx_filter.  Output equals the
// reference run with Threads = 1 (its OpenMP reduction makes ties thread-order dependent).
#include "overlap_common.hpp"

using namespace ovl;
using rfxcli::die;

namespace {

// ReplaceLowQBase (src/OverlapSam.cpp:381-390): quality - 33 < min -> 'N' (a missing quality reads as 0)
std::string mask_low_quality(const std::string& s, const std::string& q, int min) {
  std::string o(s);
  for (size_t i = 0; i < s.size(); ++i) {
    const int c = i < q.size() ? (int)q[i] : 0;
    if (c - 33 < min) o[i] = 'N';
  }
  return o;
}

// TrimNends (src/OverlapSam.cpp:359-380): drop the trailing run of non-ACGT characters
void trim_n_ends(std::string& s, std::string& q) {
  size_t n = s.size();
  while (n > 0 && s[n - 1] != 'A' && s[n - 1] != 'C' && s[n - 1] != 'G' && s[n - 1] != 'T') --n;
  std::string nq;
  for (size_t i = 0; i < n; ++i) nq += i < q.size() ? q[i] : '\0';
  s.resize(n);
  q = nq;
}

struct Pool {
  std::vector<std::string> seq, qual, depth, strand;
};

}  // namespace

int main(int argc, char** argv) {
  std::cout << "you gave " << argc << " Arguments" << std::endl;
  if (argc != 10) {
    std::cout << "ERROR, wrong numbe of arguemnts\nCall is: SAM, MinPercent, MinOverlap, MinCoverage, ReportStub, "
                 "NodeStub LCcutoff HashList Threads"
              << std::endl;
    return 0;
  }
  std::ifstream sam(argv[1]);
  if (!sam.is_open()) {
    std::cout << "Error, ParentHashFile could not be opened";
    return 0;
  }
  const float min_pct = (float)atof(argv[2]);
  const int min_ovl = atoi(argv[3]), min_cov = atoi(argv[4]);
  const std::string stub = argv[5], node = argv[6];
  std::ofstream report((stub + ".fastq").c_str()), dep((stub + ".fastqd").c_str());
  if (!report.is_open()) {
    std::cout << "ERROR, Mut-Output file could not be opened - " << stub << ".fastq" << std::endl;
    return 0;
  }
  std::string list;
  {
    std::ifstream f(argv[8], std::ios::binary);
    if (!f.is_open()) {
      std::cout << "Error, ParentHashFile could not be opened";
      return 0;
    }
    list.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  // hash list: first space-separated field of every line, forward and reverse complement (:677-689)
  std::string first_fields;
  int k = -1;
  {
    std::istringstream ls(list);
    std::string l;
    while (std::getline(ls, l)) {
      const std::vector<std::string> t = split(l, ' ');
      if (t.empty()) continue;
      first_fields += t[0] + "\n";
      k = (int)t[0].size();
    }
  }
  if (k == -1) {
    std::cout << "ERROR Hash Size could not be determined by the HashFile" << std::endl;
    return -1;
  }
  if (k > 32) die("rufus_amd OverlapSam: hash size must be <= 32");

  rfx_ctx* ctx = rfxcli::open_ctx();
  const long nk = rfx_hashlist_keys(first_fields.data(), first_fields.size(), k, 0, nullptr, 0);
  std::vector<uint64_t> keys((size_t)nk + 1);
  rfx_hashlist_keys(first_fields.data(), first_fields.size(), k, 0, keys.data(), keys.size());
  rfx_set* set = rfx_set_build(ctx, keys.data(), (uint64_t)nk, k);
  if (!set) die(std::string("rufus_amd: ") + rfx_last_error());

  // ---- intake (:708-852) -------------------------------------------------------------------
  struct Cand {
    std::string seq, qual;
    int flag;
    bool unmapped;
    size_t read_size;
  };
  std::vector<Cand> cands;
  int rejects = 0;
  std::string line;
  while (std::getline(sam, line)) {
    std::vector<std::string> t = split(line, '\t');
    if (t.size() < 11) continue;  // the reference indexes temp[10] unchecked
    t[9] = mask_low_quality(t[9], t[10], 10);
    const int v = atoi(t[1].c_str());
    int lowq = 0;
    for (char c : t[10]) lowq += ((int)c - 33 < 20);
    const size_t length = t[10].size();
    if ((v & (1 << 8)) || (v & (1 << 11)) || (v & (1 << 10)) || t[9].size() < 50 ||
        (double)lowq / (double)length > 0.33) {
      ++rejects;
      continue;
    }
    Cand c{t[9], t[10], v, (v & 4) != 0, t[10].size()};
    trim_n_ends(c.seq, c.qual);
    cands.push_back(std::move(c));
  }
  // CountHashes (:534-548): windows without 'N', the last one skipped -> k_filter with good = (base != 'N')
  std::vector<uint32_t> hits(cands.size() + 1, 0);
  if (!cands.empty()) {
    rfxcli::ReadBatch b;
    for (auto& c : cands) {
      const std::string q(c.seq.size(), 'J');
      b.add(c.seq.data(), c.seq.size(), q.data(), q.size(), true);
    }
    rfxcli::PackedBatch p;
    if (p.pack(b, RFX_PACK_FILTER, 0) != RFX_OK) die("rufus_amd: pack failed");
    rfx_reads* rd = p.upload(ctx, b.n(), RFX_PACK_FILTER);
    if (!rd) die(std::string("rufus_amd: ") + rfx_last_error());
    uint64_t nh = 0;
    if (rfx_filter(set, rd, 1, 1, hits.data(), nullptr, &nh) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
    rfx_reads_free(rd);
  }
  Pool main_pool, un_pool;
  for (size_t i = 0; i < cands.size(); ++i) {
    const Cand& c = cands[i];
    if (!((double)c.seq.size() / (double)c.read_size > .6)) {
      ++rejects;
      continue;
    }
    Pool& p = c.unmapped ? un_pool : main_pool;
    p.seq.push_back(c.seq);
    p.qual.push_back(c.qual);
    p.depth.push_back(std::string(c.seq.size(), '\x01'));
    const bool tagged = c.seq.size() >= (size_t)k && hits[i] > 0;
    p.strand.push_back(!tagged ? "." : !(c.flag & 1) ? "." : !(c.flag & 16) ? "+" : "-");
  }
  std::cout << "\nDone reading in \n\t\t Read in a total of " << main_pool.seq.size() + (size_t)rejects << " and rejected "
            << rejects << std::endl;

  // ---- greedy merge (:866-1024) ----------------------------------------------------------------
  std::vector<std::string>&seqs = main_pool.seq, &quals = main_pool.qual, &depths = main_pool.depth,
                          &strands = main_pool.strand;
  const int n = (int)seqs.size();
  PoolScorer scorer;  // the reads live on the device; a merge patches the one entry it changes
  if (n) scorer.create(ctx, seqs);
  for (int i = 0; i < n; ++i) {
    std::string a = seqs[(size_t)i], aq = quals[(size_t)i], ad = depths[(size_t)i], as = strands[(size_t)i];
    std::vector<int> idx;
    for (int j = i + 1; j < std::min(n, i + 11); ++j) idx.push_back(j);
    AlignResult rev;
    bool rev_done = false;
    AlignResult best = scorer.both(i, a, idx, idx, true, min_pct, min_ovl, RFX_OVL_SAM, -1, rev, rev_done);
    if (rev_done && rev.score > best.score) {
      a = revcomp(a); aq = revqual(aq); ad = revqual(ad); as = flip_strands(as);
      best = rev;
    }
    if (best.score < min_ovl) continue;
    const size_t bi = (size_t)best.index;
    std::string bq = quals[bi], bd = depths[bi], bs = strands[bi];
    const std::string merged = collapse(a, seqs[bi], best.overlap, aq, bq, ad, bd, as, bs, MERGE_SAM);
    scorer.set((int)bi, merged);
    seqs[bi] = merged;
    quals[bi] = bq;
    depths[bi] = bd;
    strands[bi] = bs;
    seqs[(size_t)i] = "moved";
  }

  // ---- output (:1029-1132) ---------------------------------------------------------------------
  int count = 0;
  for (int i = 0; i < n; ++i) {
    if (seqs[(size_t)i] == "moved" || seqs[(size_t)i].size() < 95) continue;
    int max_dep = -1;
    for (char c : depths[(size_t)i]) max_dep = std::max(max_dep, (int)(unsigned char)c);
    if (max_dep < min_cov) continue;
    ++count;
    int f = 0, r = 0;
    strand_counts(strands[(size_t)i], f, r);
    std::ostringstream h;
    h << "@NODE_" << node << "_" << i << "_L=" << seqs[(size_t)i].size() << "_D=" << max_dep << ":" << f << ":" << r << ":";
    report << h.str() << '\n' << seqs[(size_t)i] << "\n+\n" << quals[(size_t)i] << '\n';
    dep << h.str() << '\n' << seqs[(size_t)i] << "\n+\n" << quals[(size_t)i] << '\n' << strands[(size_t)i] << '\n';
    write_depths(dep, depths[(size_t)i]);
  }
  if (min_cov <= 1) {
    for (size_t i = 0; i < un_pool.seq.size(); ++i) {
      if (un_pool.seq[i] == "moved" || un_pool.seq[i].size() < 95) continue;
      ++count;
      std::ostringstream h;
      h << "@NODE_" << node << "_" << i << "_L=" << un_pool.seq[i].size() << "_D" << -1;
      report << h.str() << '\n' << un_pool.seq[i] << "\n+\n" << un_pool.qual[i] << '\n';
      dep << h.str() << '\n' << un_pool.seq[i] << "\n+\n" << un_pool.qual[i] << '\n' << un_pool.strand[i] << '\n';
      write_depths(dep, un_pool.depth[i]);
    }
  } else {
    std::cout << "min coverage = " << min_cov << " skipping Unaligned sequences" << std::endl;
  }
  std::cout << "\nWrote " << count << " sequences" << std::endl;
  scorer.release();  // before the context goes
  rfx_set_free(set);
  rfx_close(ctx);
  return 0;
}
