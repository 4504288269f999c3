// Host-only check of the line scanners of the drop-in tools (rufus_amd/csrc/host/rfx_cli.hpp): the 32-bytes-per-step
// versions against the memchr ones on random text with every line length from 0 up, buffers cut at every offset
// modulo 32, with and without a final newline.  Prints "ok N" (N = comparisons made) or the first mismatch.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../rufus_amd/csrc/host/rfx_cli.hpp"

using namespace rfxcli;

int main() {
  std::mt19937_64 rng(12345);
  const skip_lines_fn fast_skip = pick_skip_lines();
  const index_lines_fn fast_index = pick_index_lines();
  unsigned long long checks = 0;
  for (int round = 0; round < 400; ++round) {
    std::string text;
    const int n_lines = 1 + (int)(rng() % 300);
    for (int i = 0; i < n_lines; ++i) {
      const size_t len = round % 3 == 0 ? rng() % 4 : rng() % 200;
      for (size_t j = 0; j < len; ++j) text.push_back((char)('A' + rng() % 20));
      text.push_back('\n');
    }
    if (round % 2) text.pop_back();  // no final newline
    const size_t lead = rng() % 33;  // alignment of the buffer start
    std::string padded(lead, '\n');
    padded += text;
    const char *b = padded.data() + lead, *e = padded.data() + padded.size();
    for (size_t want : {(size_t)0, (size_t)1, (size_t)2, (size_t)(n_lines / 2), (size_t)n_lines, (size_t)n_lines + 5}) {
      size_t g1, g2;
      const char* p1 = skip_lines_plain(b, e, want, g1);
      const char* p2 = fast_skip(b, e, want, g2);
      if (want == 0) { g1 = g2 = 0; p1 = p2 = b; }
      if (p1 != p2 || g1 != g2) {
        printf("skip_lines mismatch: round %d want %zu: plain (%td, %zu) fast (%td, %zu)\n", round, want, p1 - b, g1, p2 - b, g2);
        return 1;
      }
      ++checks;
    }
    for (size_t max : {(size_t)1, (size_t)n_lines / 2 + 1, (size_t)n_lines, (size_t)n_lines + 7}) {
      std::vector<uint64_t> s1(max + 1, ~0ull), s2(max + 1, ~0ull);
      const size_t n1 = index_lines_plain(b, e, s1.data(), max), n2 = fast_index(b, e, s2.data(), max);
      if (n1 != n2) { printf("index_lines count mismatch: round %d max %zu: %zu vs %zu\n", round, max, n1, n2); return 1; }
      for (size_t i = 0; i < n1; ++i)
        if (s1[i] != s2[i]) { printf("index_lines mismatch: round %d line %zu\n", round, i); return 1; }
      ++checks;
    }
    for (size_t off = 0; off < 40 && b + off < e; ++off) {  // find_nl from every offset
      const char* f = find_nl(b + off, e);
      const char* m = (const char*)memchr(b + off, '\n', (size_t)(e - (b + off)));
      if (f != m) { printf("find_nl mismatch: round %d off %zu\n", round, off); return 1; }
      ++checks;
    }
  }
  printf("ok %llu\n", checks);
  return 0;
}
