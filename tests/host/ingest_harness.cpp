// Host-only harness for the parallel FASTQ ingest of the drop-in `jellyfish count` (rfx_ingest.hpp): runs the
// reader / worker / block pipeline with malloc'ed staging blocks and a sink that only checksums what it is given,
// so the concurrency (block hand-over, sealing, pipe cutting) is tested without a GPU.
//   ingest_harness THREADS CAP_READS CAP_WORDS PIECE_BYTES FILE|-     (- = stdin, taken as a pipe)
// prints: reads bases checksum blocks      checksum = sum over reads of FNV-1a(len, code words, mask words)
#include <cstdio>
#include <cstdlib>

#include "../../rufus_amd/csrc/host/rfx_ingest.hpp"

using namespace rfxcli;

static uint64_t fnv(uint64_t h, uint64_t v) {
  for (int i = 0; i < 8; ++i) {
    h ^= (v >> (8 * i)) & 255u;
    h *= 0x100000001B3ull;
  }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const unsigned threads = (unsigned)atoi(argv[1]);
  const uint32_t cap_reads = (uint32_t)atoi(argv[2]);
  const uint64_t cap_words = strtoull(argv[3], 0, 10);
  uint64_t reads = 0, bases = 0, sum = 0, blocks = 0;
  const bool nosum = getenv("INGEST_NOSUM") != nullptr;
  CountIngest ing(threads, [&](const StageBlock& b) {
    ++blocks;
    if (nosum) {  // throughput runs: INGEST_NOSUM=1
      reads += b.n_reads;
      return;
    }
    for (uint32_t r = 0; r < b.n_reads; ++r) {
      uint64_t h = fnv(0xCBF29CE484222325ull, b.len[r]);
      const uint32_t w0 = b.word_off[r], w1 = b.word_off[r + 1];
      if (w1 - w0 != (b.len[r] + 31) / 32) { fprintf(stderr, "bad offsets at read %u\n", r); exit(3); }
      for (uint32_t w = w0; w < w1; ++w) h = fnv(fnv(h, b.codes[w]), b.acgt[w]);
      sum += h;
      bases += b.len[r];
    }
    reads += b.n_reads;
  }, malloc, free, cap_reads, cap_words);
  ing.set_piece_bytes(strtoull(argv[4], 0, 10));
  bool ok;
  if (strcmp(argv[5], "-") == 0) {
    std::vector<char> head(1 << 12);
    const ssize_t n = ::read(0, head.data(), head.size());
    head.resize(n > 0 ? (size_t)n : 0);
    ok = ing.feed_stream(0, head);
  } else {
    const int fd = ::open(argv[5], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st)) return 1;
    if (getenv("INGEST_MMAP")) {
      void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      ok = ing.feed_mapped((const char*)m, (size_t)st.st_size);
    } else {
      ok = ing.feed_file(fd, (uint64_t)st.st_size);
    }
  }
  printf("%s %llu %llu %llu %llu\n", ok ? "ok" : "not4line", (unsigned long long)reads, (unsigned long long)bases,
         (unsigned long long)sum, (unsigned long long)blocks);
  return 0;
}
