// Drop-in RUFUS.Filter (paired) and RUFUS.Filter.single (-DRFX_SINGLE_END), same argv and outputs:
//   RUFUS.Filter        HashList Mate1.fq Mate2.fq STUB K MinQ HashCountThreshold Threads   (runRufus.sh:967)
//       -> STUB.Mutations.Mate1.fastq, STUB.Mutations.Mate2.fastq        src/RUFUS.Filter.cpp:28,:47-54
//   RUFUS.Filter.single HashList FQ|stdin STUB K MinQ HashCountThreshold Threads
//       -> STUB.Mutations.fastq with ":MH<hits>" appended to each header  src/RUFUS.Filter.ss.cpp:27,:43-49,:198
// The two mate files may be named pipes written in lock step by PassThroughSamCheck.stranded, so
// they are read interleaved, 4 lines each (src/RUFUS.Filter.cpp:165-173) -- never one file ahead.
// Pulled records are written in input order (the reference's order depends on OpenMP scheduling; at
// one thread it is input order too).  The scan itself runs in k_filter; no CPU fallback.
#include <fstream>

#include "rfx_cli.hpp"

using namespace rfxcli;

struct Rec {
  std::string l[4];
};

static bool read_rec(LineReader& in, Rec& r) {
  const char *b, *e;
  if (!in.getline(b, e)) return false;
  r.l[0].assign(b, e);
  for (int i = 1; i < 4; ++i) {
    if (in.getline(b, e)) r.l[i].assign(b, e);
    else r.l[i].clear();
  }
  return true;
}

int main(int argc, char** argv) {
#ifdef RFX_SINGLE_END
  const int need = 8;
  printf("Call is PreBuiltMutHash Mutant.fq|stdin firstpassfile hashsize MinQ HashCountThreshold threads\n");
#else
  const int need = 9;
  printf("Call is PreBuiltMutHash Mutant.Mate1.fq Mutant.Mate2.fq firstpassfile hashsize MinQ HashCountThreshold threads \n");
#endif
  if (argc < need) {
    printf("ERROR: expected %d arguments\n", need - 1);
    return 0;  // the reference tools report on stdout and exit 0; the shell checks for empty outputs
  }
  int a = 1;
  const char* hashlist = argv[a++];
  const char* m1 = argv[a++];
#ifndef RFX_SINGLE_END
  const char* m2 = argv[a++];
#endif
  const std::string stub = argv[a++];
  const int k = atoi(argv[a++]), min_q = atoi(argv[a++]), thresh = atoi(argv[a++]);

  std::string text;
  {
    std::ifstream f(hashlist, std::ios::binary);
    if (!f.is_open()) {
      printf("Error, ParentHashFile could not be opened");
      return 0;
    }
    text.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  LineReader in1;
  if (!in1.open(m1)) {
    printf("Error, MutFile could not be opened");
    return 0;
  }
#ifdef RFX_SINGLE_END
  const int single = 1;
  std::ofstream out1((stub + ".Mutations.fastq").c_str(), std::ios::binary);
#else
  const int single = 0;
  LineReader in2;
  if (!in2.open(m2)) {
    printf("Error, MutFile could not be opened");
    return 0;
  }
  std::ofstream out1((stub + ".Mutations.Mate1.fastq").c_str(), std::ios::binary);
  std::ofstream out2((stub + ".Mutations.Mate2.fastq").c_str(), std::ios::binary);
  if (!out2.is_open()) {
    printf("ERROR, Output file could not be opened -%s\n", stub.c_str());
    return 0;
  }
#endif
  if (!out1.is_open()) {
    printf("ERROR, Output file could not be opened -%s\n", stub.c_str());
    return 0;
  }
  if (k < 1 || k > 32) die("rufus_amd RUFUS.Filter: hash size must be 1..32");

  const long nk = rfx_hashlist_keys(text.data(), text.size(), k, single, nullptr, 0);
  if (nk < 0) die("rufus_amd: cannot parse the hash list");
  std::vector<uint64_t> keys((size_t)nk + 1);
  rfx_hashlist_keys(text.data(), text.size(), k, single, keys.data(), keys.size());
  printf("\nDone Hash Files\n\t Mutations Hash size is %ld\n", nk);

  rfx_ctx* ctx = open_ctx();
  rfx_set* set = rfx_set_build(ctx, keys.data(), (uint64_t)nk, k);
  if (!set) die(std::string("rufus_amd: ") + rfx_last_error());

  const size_t BATCH = 1u << 19;
  std::vector<Rec> r1, r2;
  r1.reserve(BATCH);
  ReadBatch b1, b2;
  PackedBatch p;
  std::vector<uint64_t> mask1, mask2;
  std::vector<uint32_t> hits;
  unsigned long long found = 0, total = 0;
  bool more = true;
  while (more) {
    r1.clear();
    r2.clear();
    b1.clear();
    b2.clear();
    while (r1.size() < BATCH) {
      Rec a1;
      if (!read_rec(in1, a1)) {
        more = false;
        break;
      }
      if (a1.l[1].empty()) die("rufus_amd RUFUS.Filter: empty sequence line (undefined behaviour in the reference) -- rejected");
      b1.add(a1.l[1].data(), a1.l[1].size(), a1.l[3].data(), a1.l[3].size(), true);
      r1.push_back(std::move(a1));
#ifndef RFX_SINGLE_END
      Rec a2;
      read_rec(in2, a2);  // lock step: exactly one record of mate 2 per record of mate 1
      b2.add(a2.l[1].data(), a2.l[1].size(), a2.l[3].data(), a2.l[3].size(), true);
      r2.push_back(std::move(a2));
#endif
    }
    const uint32_t n = (uint32_t)r1.size();
    if (n == 0) break;
    total += n;
    auto scan = [&](ReadBatch& b, std::vector<uint64_t>& mask, std::vector<uint32_t>* h) {
      if (p.pack(b, RFX_PACK_FILTER, min_q) != RFX_OK) die("rufus_amd: pack failed");
      rfx_reads* rd = p.upload(ctx, n, RFX_PACK_FILTER);
      if (!rd) die(std::string("rufus_amd: upload failed: ") + rfx_last_error());
      mask.assign(((size_t)n + 63) / 64, 0);
      if (h) h->assign(n, 0);
      uint64_t nh = 0;
      // paired tool: `i < length()-1`, the last base is never examined (src/RUFUS.Filter.cpp:203)
      const int rc = rfx_filter(set, rd, thresh, single ? 0 : 1, h ? h->data() : nullptr, mask.data(), &nh);
      rfx_reads_free(rd);
      if (rc) die(std::string("rufus_amd: filter failed: ") + rfx_last_error());
    };
#ifdef RFX_SINGLE_END
    scan(b1, mask1, &hits);
    for (uint32_t i = 0; i < n; ++i)
      if ((mask1[i >> 6] >> (i & 63)) & 1) {
        out1 << r1[i].l[0] << ":MH" << hits[i] << '\n' << r1[i].l[1] << '\n' << r1[i].l[2] << '\n' << r1[i].l[3] << '\n';
        ++found;
      }
#else
    scan(b1, mask1, nullptr);
    scan(b2, mask2, nullptr);  // equivalent to the reference's "mate 2 only if mate 1 failed" (:237-277)
    for (uint32_t i = 0; i < n; ++i)
      if (((mask1[i >> 6] | mask2[i >> 6]) >> (i & 63)) & 1) {
        out1 << r1[i].l[0] << '\n' << r1[i].l[1] << '\n' << r1[i].l[2] << '\n' << r1[i].l[3] << '\n';
        out2 << r2[i].l[0] << '\n' << r2[i].l[1] << '\n' << r2[i].l[2] << '\n' << r2[i].l[3] << '\n';
        ++found;
      }
#endif
    printf("Read in %llu lines: Found %llu \r", total * 4, found);
  }
  rfx_set_free(set);
  rfx_close(ctx);
  printf("\nDone running RUFUS.Filter.cpp\n");
  return 0;
}
