// Synthetic trio workload of SURVEY.md 8(d): every base of every read is a pure function of
// (rfx_synth parameters, pair index, mate, base index), so the device generator (rfx_synth.hip), the host
// twin (rfx_synth_text in rfx_host.cpp) and the independent numpy restatement in tests/synth.py produce
// the same reads for any slice of a sample -- a 600 M-read sample never has to exist as text.
//
//   genome    base i = 2 bits of mix64(genome_seed ^ (i/32) * PHI): uniform ACGT, codes A0 C1 G2 T3
//   SNVs      one per stratum of (genome_len - 2000) / n_snv bases, alt = one of the other three bases;
//             carried by haplotype 1 of a carrier sample (heterozygous de-novo variants)
//   pairs     start uniform, insert = insert_lo + U[0, insert_span), mate 1 forward, mate 2 reverse
//             complement of the insert's far end; read 2p = mate 1 of pair p, 2p + 1 = mate 2
//   per base  32 random bits: substitution error (err_1024 / 1024), low quality '#' instead of 'J'
//             (lowq_256 / 256), 'N' (n_1024 / 1024)
#pragma once
#include <stdint.h>

#include "../../include/rufus_hip.h"

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define RFX_HD __host__ __device__ __forceinline__
#elif defined(__HIPCC__)
#define RFX_HD __host__ __device__ inline
#else
#define RFX_HD inline
#endif

extern "C" int rfx_synth_check(const rfx_synth*);  // RFX_OK or RFX_E_INVAL

namespace rfxs {

constexpr uint64_t PHI = 0x9E3779B97F4A7C15ull;
constexpr uint64_t STEP = 0xD6E8FEB86659FD93ull;

RFX_HD uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += PHI;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// x in [0, 2^32) scaled to [0, range), range < 2^32
RFX_HD uint64_t scale32(uint64_t r, uint64_t range) { return ((r >> 32) * range) >> 32; }

RFX_HD uint64_t genome_word(const rfx_synth& p, uint64_t j) { return mix64(p.genome_seed ^ (j * PHI)); }

RFX_HD uint32_t genome_base(const rfx_synth& p, uint64_t i) {
  return (uint32_t)(genome_word(p, i >> 5) >> (2 * (i & 31))) & 3u;
}

RFX_HD uint64_t snv_stride(const rfx_synth& p) { return p.n_snv ? (p.genome_len - 2000) / p.n_snv : 0; }

// position and alternative base of SNV i
RFX_HD uint64_t snv_pos(const rfx_synth& p, uint64_t i) {
  const uint64_t st = snv_stride(p);
  return 1000 + i * st + scale32(mix64(p.snv_seed + i), st - 64);
}
RFX_HD uint32_t snv_alt(const rfx_synth& p, uint64_t i, uint32_t ref) {
  return (ref + 1 + (uint32_t)((mix64(p.snv_seed + i) & 0xFFFFu) % 3u)) & 3u;
}

struct pair_geom {
  uint64_t start, end;  // genome interval [start, end) of the insert
  uint32_t hap;         // haplotype of the pair
  uint64_t key;         // per-pair key the mates' base streams derive from
};

RFX_HD pair_geom pair_of(const rfx_synth& p, uint64_t pair) {
  pair_geom g;
  g.key = mix64(p.read_seed ^ (pair * PHI + 1));
  const uint64_t k1 = mix64(g.key + 1);
  g.start = scale32(g.key, p.genome_len - (p.insert_lo + p.insert_span));
  g.end = g.start + p.insert_lo + scale32(k1, p.insert_span);
  g.hap = (uint32_t)(k1 & 1u);
  return g;
}

RFX_HD uint64_t mate_key(const pair_geom& g, int mate) { return mix64(g.key + 2 + (uint64_t)mate); }

// 32 random bits of base j of a mate: [9:0] error, [13:10] substitute, [21:14] quality, [31:22] N
RFX_HD uint32_t base_bits(uint64_t mkey, uint32_t j) {
  const uint64_t r = mix64(mkey + (uint64_t)((j >> 1) + 1) * STEP);
  return (uint32_t)(j & 1u ? r >> 32 : r);
}

// One base of a read.  Returns the code (0..3) and sets is_n / lowq.
struct base_out {
  uint32_t code, is_n, lowq;
};

// genome coordinate of base j of mate `mate`
RFX_HD uint64_t base_coord(const rfx_synth& p, const pair_geom& g, int mate, uint32_t j) {
  return mate ? g.end - 1 - j : g.start + j;
}

RFX_HD base_out finish_base(const rfx_synth& p, uint32_t ref_or_alt, int mate, uint32_t bits) {
  base_out o;
  uint32_t b = mate ? 3u - ref_or_alt : ref_or_alt;
  if ((bits & 1023u) < p.err_1024) b = (b + 1 + ((bits >> 10) & 15u) % 3u) & 3u;
  o.lowq = ((bits >> 14) & 255u) < p.lowq_256;
  o.is_n = (bits >> 22) < p.n_1024;
  o.code = o.is_n ? 0u : b;
  return o;
}

}  // namespace rfxs
