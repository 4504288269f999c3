// Host-only harness for the .Jhash writer of the drop-in `jellyfish count` (rfx_cli.hpp write_jhash): the ring of fetch
// buffers and writer threads, the per-slice fetch threads of a several-device run, the mapped / pwrite / pipe routes --
// over stand-ins for the record sets (record g of the payload is rl bytes derived from g; TEST INFRASTRUCTURE).
//   write_harness OUT COUNTER_LEN N0 [N1 ...]        OUT may be /dev/stdout (a pipe: the sequential route)
// Built with -DRFX_WRITE_STEP=<small> so that a few thousand records go round the rings many times.
#include <cstdio>
#include <cstdlib>

#include "../../rufus_amd/csrc/host/rfx_cli.hpp"

struct rfx_records {
  uint64_t n, base;
};
static const int K = 25, LSIZE = 33;

extern "C" {
const char* rfx_last_error(void) { return "host stand-in"; }
void* rfx_host_alloc(size_t bytes) { return malloc(bytes); }
void rfx_host_free(void* p) { free(p); }
int rfx_records_k(const rfx_records*) { return K; }
int rfx_records_lsize(const rfx_records*) { return LSIZE; }
uint64_t rfx_records_size(const rfx_records* r) { return r->n; }
int rfx_records_payload_range(const rfx_records* r, uint64_t first, uint64_t n, void* out, size_t cap, int counter_len) {
  const size_t rl = (size_t)(2 * K + 7) / 8 + (size_t)counter_len;
  if (first + n > r->n || cap < n * rl) return RFX_E_INVAL;
  unsigned char* o = (unsigned char*)out;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t g = r->base + first + i;
    for (size_t j = 0; j < rl; ++j) o[i * rl + j] = (unsigned char)((g * 2654435761u + j * 40503u) >> 7);
  }
  return RFX_OK;
}
}

using namespace rfxcli;

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int clen = atoi(argv[2]);
  std::vector<rfx_records> store;
  uint64_t base = 0;
  for (int i = 3; i < argc; ++i) {
    store.push_back({strtoull(argv[i], nullptr, 10), base});
    base += store.back().n;
  }
  std::vector<rfx_records*> recs;
  for (auto& r : store) recs.push_back(&r);
  std::vector<uint64_t> cols((size_t)2 * K);
  if (rfx_jf_matrix(LSIZE, K, cols.data()) != RFX_OK) return 3;
  write_jhash(argv[1], recs, cols.data(), true, clen, 0, nullptr);
  return 0;
}
