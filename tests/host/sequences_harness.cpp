// Host-only harness for the sequential FASTA/FASTQ parser of the drop-in `jellyfish` (rfx_cli.hpp parse_sequences,
// jf/include/jellyfish/mer_overlap_sequence_parser.hpp:124-251):   sequences_harness FILE|stdin [BUFFER_BYTES]
// prints every sequence it yields on a line of its own, then "ok N" -- or "malformed" when the parser refuses the input.
#include <cstdio>
#include <cstdlib>

#include "../../rufus_amd/csrc/host/rfx_cli.hpp"

using namespace rfxcli;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  LineReader in(argc > 2 ? (size_t)atol(argv[2]) : (size_t)1 << 22);
  if (!in.open(argv[1])) return 3;
  std::string out;
  unsigned long n = 0;
  const bool ok = parse_sequences(in, [&](const char* s, size_t len) {
    out.append(s, len);
    out.push_back('\n');
    ++n;
  });
  if (!ok) {
    printf("malformed\n");
    return 0;
  }
  fwrite(out.data(), 1, out.size(), stdout);
  printf("ok %lu\n", n);
  return 0;
}
