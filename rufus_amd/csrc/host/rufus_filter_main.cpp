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

// The 4 lines of every record of a batch, back to back in one arena (no per-line allocations: a batch is
// half a million records).  Line j of record i is text[off[4i+j] .. off[4i+j+1]).
struct RecBatch {
  std::string text;
  std::vector<uint64_t> off{0};
  void clear() {
    text.clear();
    off.assign(1, 0);
  }
  size_t n() const { return (off.size() - 1) / 4; }
  const char* line(size_t i, int j) const { return text.data() + off[4 * i + (size_t)j]; }
  size_t len(size_t i, int j) const { return (size_t)(off[4 * i + (size_t)j + 1] - off[4 * i + (size_t)j]); }
  // Appends one record; false at end of input.  Missing trailing lines read as empty (the reference's
  // getline leaves the previous/empty string there; an incomplete last record is garbage in both).
  bool read(LineReader& in) {
    const char *b, *e;
    if (!in.getline(b, e)) return false;
    text.append(b, e);
    off.push_back(text.size());
    for (int j = 1; j < 4; ++j) {
      if (in.getline(b, e)) text.append(b, e);
      off.push_back(text.size());
    }
    return true;
  }
  void write(std::ostream& os, size_t i, const char* header_suffix = nullptr) const {
    os.write(line(i, 0), (std::streamsize)len(i, 0));
    if (header_suffix) os << header_suffix;
    os.put('\n');
    for (int j = 1; j < 4; ++j) {
      os.write(line(i, j), (std::streamsize)len(i, j));
      os.put('\n');
    }
  }
};

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
  RecBatch r1, r2;
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
    while (r1.n() < BATCH) {
      if (!r1.read(in1)) {
        more = false;
        break;
      }
      const size_t i = r1.n() - 1;
      if (r1.len(i, 1) == 0) die("rufus_amd RUFUS.Filter: empty sequence line (undefined behaviour in the reference) -- rejected");
      b1.add(r1.line(i, 1), r1.len(i, 1), r1.line(i, 3), r1.len(i, 3), true);
#ifndef RFX_SINGLE_END
      if (!r2.read(in2)) {  // lock step: exactly one record of mate 2 per record of mate 1
        r2.off.insert(r2.off.end(), 4, r2.text.size());
      }
      b2.add(r2.line(i, 1), r2.len(i, 1), r2.line(i, 3), r2.len(i, 3), true);
#endif
    }
    const uint32_t n = (uint32_t)r1.n();
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
        const std::string suffix = ":MH" + std::to_string(hits[i]);
        r1.write(out1, i, suffix.c_str());
        ++found;
      }
#else
    scan(b1, mask1, nullptr);
    scan(b2, mask2, nullptr);  // equivalent to the reference's "mate 2 only if mate 1 failed" (:237-277)
    for (uint32_t i = 0; i < n; ++i)
      if (((mask1[i >> 6] | mask2[i >> 6]) >> (i & 63)) & 1) {
        r1.write(out1, i);
        r2.write(out2, i);
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
