// rfx_synth_fastq: the synthetic workload of SURVEY.md 8(d) as FASTQ text, from the generator's host twin
// (rfx_synth_text) -- the same reads the device generator (rfx_synth_reads) puts into HBM.  Test / benchmark
// tool for the end-to-end path of the drop-in executables (text in -> files out); not part of the RUFUS surface.
//
//   rfx_synth_fastq GENOME_LEN SAMPLE(0=child,1,2=parents) N_SNV SEED FIRST_PAIR N_PAIRS OUT [OUT_MATE2]
//
// One output: reads interleaved (mate 1, mate 2, ...); two outputs: mate files in lock step.  "-" = stdout.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rufus_hip.h"

static rfx_synth sample(uint64_t G, int which, uint32_t n_snv, uint64_t seed) {  // == capi.Synth.sample
  rfx_synth p;
  memset(&p, 0, sizeof p);
  p.genome_len = G;
  p.genome_seed = seed;
  p.snv_seed = seed + 7;
  p.read_seed = seed * 1000 + (uint64_t)which;
  p.n_snv = n_snv;
  p.read_len = 150;
  p.insert_lo = 250;
  p.insert_span = 151;
  p.err_1024 = 5;
  p.lowq_256 = 5;
  p.n_1024 = 1;
  p.carrier = which == 0;
  return p;
}

int main(int argc, char** argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: %s GENOME_LEN SAMPLE N_SNV SEED FIRST_PAIR N_PAIRS OUT [OUT_MATE2]\n", argv[0]);
    return 2;
  }
  const rfx_synth p = sample(strtoull(argv[1], 0, 10), atoi(argv[2]), (uint32_t)atoi(argv[3]), strtoull(argv[4], 0, 10));
  const uint64_t first = strtoull(argv[5], 0, 10), n_pairs = strtoull(argv[6], 0, 10);
  FILE* f1 = strcmp(argv[7], "-") == 0 ? stdout : fopen(argv[7], "wb");
  FILE* f2 = argc > 8 ? fopen(argv[8], "wb") : nullptr;
  if (!f1 || (argc > 8 && !f2)) { perror("open"); return 1; }
  const uint32_t L = p.read_len;
  const bool as_sam = getenv("RFX_SYNTH_SAM") != nullptr;  // SAM lines instead of FASTQ records (for `count --sam`)
  const uint64_t step = 1u << 20;
  std::vector<char> seq(2 * step * L), qual(2 * step * L), out1, out2;
  for (uint64_t at = 0; at < n_pairs; at += step) {
    const uint32_t m = (uint32_t)std::min<uint64_t>(step, n_pairs - at);
    if (rfx_synth_text(&p, first + at, m, seq.data(), qual.data()) != RFX_OK) { fprintf(stderr, "bad parameters\n"); return 1; }
    // text of the records, formatted in parallel into per-thread strings
    const unsigned nt = std::max(1u, std::min(64u, rfx_host_cpus()));
    std::vector<std::string> part1(nt), part2(nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        char hdr[96];
        for (uint32_t q = (uint64_t)m * t / nt, e = (uint64_t)m * (t + 1) / nt; q < e; ++q)
          for (int mate = 0; mate < 2; ++mate) {
            std::string& o = (f2 && mate) ? part2[t] : part1[t];
            if (as_sam) {  // what `samtools view` would print for it (mapped somewhere on chr1 .. chr22, by pair number)
              const unsigned long long pr = (unsigned long long)(first + at + q);
              const int hl = snprintf(hdr, sizeof hdr, "r%llu\t%d\tchr%llu\t%llu\t60\t150M\t=\t1\t0\t", pr,
                                      mate ? 147 : 99, 1 + pr * 22 / (first + n_pairs), 1 + pr % 1000000);
              o.append(hdr, (size_t)hl);
              o.append(seq.data() + ((size_t)2 * q + mate) * L, L);
              o.push_back('\t');
              o.append(qual.data() + ((size_t)2 * q + mate) * L, L);
              o.append("\tNM:i:0\n", 8);
              continue;
            }
            const int hl = snprintf(hdr, sizeof hdr, "@r%llu/%d\n", (unsigned long long)(first + at + q), mate + 1);
            o.append(hdr, (size_t)hl);
            o.append(seq.data() + ((size_t)2 * q + mate) * L, L);
            o.append("\n+\n", 3);
            o.append(qual.data() + ((size_t)2 * q + mate) * L, L);
            o.push_back('\n');
          }
      });
    for (auto& x : th) x.join();
    for (unsigned t = 0; t < nt; ++t) {
      if (fwrite(part1[t].data(), 1, part1[t].size(), f1) != part1[t].size()) { perror("write"); return 1; }
      if (f2 && fwrite(part2[t].data(), 1, part2[t].size(), f2) != part2[t].size()) { perror("write"); return 1; }
    }
  }
  if (f1 != stdout) fclose(f1);
  if (f2) fclose(f2);
  return 0;
}
