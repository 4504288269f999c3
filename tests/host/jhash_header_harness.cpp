// Host-only harness for the .Jhash header reader of the drop-in tools (rfx_cli.hpp read_jhash; the writer is the
// library's rfx_jhash_header, host code):   jhash_header_harness write FILE K LSIZE CANONICAL COUNTER_LEN NRECORDS
//                                           jhash_header_harness read FILE
// `read` prints "k lsize counter_len canonical format ncols payload_offset file_size col0 colLast" or "bad".
#include <cstdio>
#include <cstdlib>

#include "../../rufus_amd/csrc/host/rfx_cli.hpp"

using namespace rfxcli;

int main(int argc, char** argv) {
  if (argc >= 8 && strcmp(argv[1], "write") == 0) {
    const int k = atoi(argv[3]), lsize = atoi(argv[4]), canonical = atoi(argv[5]), clen = atoi(argv[6]);
    const long n = atol(argv[7]);
    std::vector<uint64_t> cols((size_t)2 * k);
    if (rfx_jf_matrix(lsize, k, cols.data()) != RFX_OK) return 3;
    std::vector<char> buf(1 << 16);
    const long hl = rfx_jhash_header(k, lsize, cols.data(), canonical, clen, 0, nullptr, buf.data(), buf.size());
    if (hl < 0) return 4;
    FILE* f = fopen(argv[2], "wb");
    fwrite(buf.data(), 1, (size_t)hl, f);
    const size_t rl = (size_t)(2 * k + 7) / 8 + (size_t)clen;
    std::vector<char> rec(rl, 'x');
    for (long i = 0; i < n; ++i) fwrite(rec.data(), 1, rl, f);
    fclose(f);
    printf("%ld %llu %llu\n", hl, (unsigned long long)cols[0], (unsigned long long)cols.back());
    return 0;
  }
  if (argc >= 3 && strcmp(argv[1], "read") == 0) {
    JhashHeader h;
    if (!read_jhash(argv[2], h, nullptr)) {
      printf("bad\n");
      return 0;
    }
    printf("%d %d %d %d %s %zu %zu %llu %llu %llu\n", h.k, h.lsize, h.counter_len, (int)h.canonical, h.format.c_str(), h.cols.size(),
           h.payload_offset, (unsigned long long)h.file_size, (unsigned long long)h.cols[0], (unsigned long long)h.cols.back());
    return 0;
  }
  return 2;
}
