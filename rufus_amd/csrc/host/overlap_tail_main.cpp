// Drop-in tail of the assembly chain, built three times with -DTAIL_MODE=0/1/2:
//   0  ReplaceQwithDinFASTQD FILE      quality line := char(min(depth + 33, 126))      src/ReplaceQwithDinFASTQD.cpp:138-202
//   1  ConvertFASTqD.to.FASTQ FILE     keep lines 1-4 of every 6-line record            src/ConvertFASTqD.to.FASTQ.cpp:55-64
//   2  AnnotateOverlap HashList FASTQ|stdin HASHOUT                                      src/AnnotateOverlap.cpp:33-158
//        stdout: header + ":MH0", sequence, '+', per-base mutant-k-mer coverage as char(min(cov,93)+33)
//        HASHOUT: canonical (lexicographically smaller of k-mer / reverse complement) k-mers, " 1" each
//      -- this is the `*.generator.V2.overlap.hashcount.fastq` the downstream steps read.
// Modes 0 and 1 are text plumbing; mode 2's coverage comes from rfx_annotate (K7) on the device.
#include "overlap_common.hpp"

#ifndef TAIL_MODE
#define TAIL_MODE 0
#endif

using namespace ovl;

int main(int argc, char** argv) {
#if TAIL_MODE != 2
  if (argc != 2) {
    std::cout << "ERROR, wrong numbe of arguemnts\nCall is: FASTQD " << std::endl;
    return 0;
  }
  std::ifstream in(argv[1]);
  if (!in.is_open()) {
    std::cout << "Error, ParentHashFile could not be opened";
    return 0;
  }
  std::string l[6];
  while (std::getline(in, l[0])) {
    for (int i = 1; i < 6; ++i)
      if (!std::getline(in, l[i])) l[i].clear();
#if TAIL_MODE == 0
    std::string adj;
    for (const std::string& t : split(l[5], ' ')) {
      const unsigned char d = (unsigned char)atoi(t.c_str());
      adj += (int)d + 33 > 126 ? (char)126 : (char)(d + 33);
    }
    std::cout << l[0] << '\n' << l[1] << '\n' << l[2] << '\n' << adj << '\n' << l[4] << '\n' << l[5] << '\n';
#else
    std::cout << l[0] << '\n' << l[1] << '\n' << l[2] << '\n' << l[3] << '\n';
#endif
  }
  return 0;
#else
  if (argc < 4) {
    std::cout << "ERROR, Call is: HashList FASTQ|stdin HashOut" << std::endl;
    return 0;
  }
  std::string list;
  {
    std::ifstream f(argv[1], std::ios::binary);
    list.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  // loader :55-77: "KMER COUNT" -> field 0, 4 fields -> field 3, one field -> re-split on TAB;
  // the set only holds the forward spelling, the scan tests the window and its reverse complement
  int k = 0;
  {
    std::istringstream ls(list);
    std::string l;
    while (std::getline(ls, l)) {
      std::vector<std::string> t = split(l, ' ');
      if (t.size() == 2) k = (int)t[0].size();
      else if (t.size() == 4) k = (int)t[3].size();
      else if (t.size() == 1) {
        t = split(l, '\t');
        if (!t.empty()) k = (int)t[0].size();
      }
    }
  }
  std::ifstream fq(std::string(argv[2]) == "stdin" ? "/dev/stdin" : argv[2]);
  std::ofstream hash_out(argv[3]);
  struct Rec { std::string h, s, p, q; };
  std::vector<Rec> recs;
  Rec r;
  while (std::getline(fq, r.h)) {
    if (!std::getline(fq, r.s)) r.s.clear();
    if (!std::getline(fq, r.p)) r.p.clear();
    if (!std::getline(fq, r.q)) r.q.clear();
    recs.push_back(r);
  }
  std::vector<uint32_t> cov;
  std::vector<uint64_t> off(1, 0);
  for (auto& x : recs) off.push_back(off.back() + x.s.size());
  if (!recs.empty() && k >= 1 && k <= 32) {
    rfx_ctx* ctx = rfxcli::open_ctx();
    const long nk = rfx_hashlist_keys(list.data(), list.size(), k, 0, nullptr, 0);
    std::vector<uint64_t> keys((size_t)(nk > 0 ? nk : 0) + 1);
    if (nk > 0) rfx_hashlist_keys(list.data(), list.size(), k, 0, keys.data(), keys.size());
    rfx_set* set = rfx_set_build(ctx, keys.data(), (uint64_t)(nk > 0 ? nk : 0), k);
    if (!set) rfxcli::die(std::string("rufus_amd: ") + rfx_last_error());
    rfxcli::ReadBatch b;
    for (auto& x : recs) b.add(x.s.data(), x.s.size(), x.q.data(), x.q.size(), true);
    rfxcli::PackedBatch p;
    // good base: not 'N' and quality - 33 >= 3 (:107-118)
    if (p.pack(b, RFX_PACK_FILTER, 3) != RFX_OK) rfxcli::die("rufus_amd: pack failed");
    rfx_reads* rd = p.upload(ctx, b.n(), RFX_PACK_FILTER);
    if (!rd) rfxcli::die(std::string("rufus_amd: ") + rfx_last_error());
    cov.assign(off.back() + 1, 0);
    if (rfx_annotate(set, rd, cov.data()) != RFX_OK) rfxcli::die(std::string("rufus_amd: ") + rfx_last_error());
    rfx_reads_free(rd);
    rfx_set_free(set);
    rfx_close(ctx);
  } else {
    cov.assign(off.back() + 1, 0);
  }
  for (size_t i = 0; i < recs.size(); ++i) {
    const Rec& x = recs[i];
    std::cout << x.h << ":MH0" << '\n' << x.s << '\n' << x.p << '\n';  // the counter is never incremented (:101,:136)
    std::string line;
    for (size_t j = 0; j < x.s.size(); ++j) {
      const uint32_t c = cov[off[i] + j];
      line += c < 93 ? (char)(c + 33) : (char)126;
    }
    if (x.s.empty()) line += (char)33;  // the reference prints HashPos[0] unconditionally; rejected input upstream
    std::cout << line << '\n';
    for (size_t j = 0; j + (size_t)k < x.s.size(); ++j) {  // every window but the last (:149-156)
      const std::string h = x.s.substr(j, (size_t)k), rc = revcomp(h);
      hash_out << (h < rc ? h : rc) << " 1" << '\n';
    }
  }
  return 0;
#endif
}
