// Host-only harness for the greedy assemblers of the drop-in tool set (OverlapSam, Overlap,
// OverlapRegion -- rufus_amd/csrc/host/overlap_sam_main.cpp, overlap_contig_main.cpp, overlap_common.hpp): the tools' own
// main() with the DEVICE entry points replaced by plain host code, so that the SAM intake, the collapse passes and the
// greedy merge loops run in the CPU suite and under the sanitizers against the reference binaries.  The stand-in for the
// scoring kernel is the per-candidate body of Align3 (src/OverlapSam.cpp:47-229, src/Overlap.cpp:176-340) in binary32,
// the same restatement tests/test_overlap_gpu.py holds the kernel to.  TEST INFRASTRUCTURE, never built into the product.
//   g++ -O2 -std=c++17 -pthread -ffp-contract=off -DOVL_WHICH=0|1|2|3 overlap_harness.cpp ../../rufus_amd/csrc/rfx_host.cpp
//   OVL_WHICH: 0 OverlapSam, 1 Overlap, 2 OverlapRegion, 3 AnnotateOverlap (row G7: rfx_annotate, src/AnnotateOverlap.cpp:88-134)
#include <string>
#include <unordered_set>
#include <vector>

#if OVL_WHICH == 0
#include "../../rufus_amd/csrc/host/overlap_sam_main.cpp"
#elif OVL_WHICH == 1
#include "../../rufus_amd/csrc/host/overlap_contig_main.cpp"
#elif OVL_WHICH == 2
#define REGION
#include "../../rufus_amd/csrc/host/overlap_contig_main.cpp"
#else
#define TAIL_MODE 2
#include "../../rufus_amd/csrc/host/overlap_tail_main.cpp"
#endif

struct rfx_ctx { int device; };
struct rfx_ovl_pool { std::vector<std::string> seq; };
struct rfx_set {
  std::unordered_set<uint64_t> keys;
  int k;
};
struct rfx_reads {
  std::vector<uint64_t> codes;
  std::vector<uint32_t> good, woff, len;
};

static void align3_one(const std::string& a, const std::string& b, float min_pct, int min_ovl, bool strict3, int init, int* out) {
  const int al = (int)a.size(), bl = (int)b.size();
  const bool asm_ = !(bl > al);
  const int window = asm_ ? bl : al, longest = asm_ ? al : bl;
  const int mm = (int)((float)window - (float)window * min_pct);
  int best = init, ovl = 0;
  bool perfect = false;
  int ac = 0, bc = 0;
  for (int i = 0; i < longest - window + 1; ++i) {
    float score = 0;
    for (int k = 0; k < window; ++k) {
      if (a[(size_t)(k + ac)] == b[(size_t)(k + bc)] && b[(size_t)(k + bc)] != 'N') score += 1.0f;
      if ((float)k - score > (float)mm) {
        score = -1.0f;
        break;
      }
    }
    if (asm_) ++ac;
    else ++bc;
    if (window && score / (float)window >= min_pct) {
      if ((float)best < score) {
        best = (int)score;
        ovl = asm_ ? -i : i;
      }
      if (score == (float)window) {
        perfect = true;
        break;
      }
    }
  }
  out[0] = best;
  out[1] = ovl;
  out[2] = perfect ? 1 : 0;
  if (!perfect)
    for (int phase = 2; phase <= 3; ++phase)
      for (int i = window - 1; i >= min_ovl; --i) {
        float score = 0;
        int k = 0;
        bool broke = false;
        for (k = 0; k < i + 1; ++k) {
          const char x = phase == 2 ? a[(size_t)(al - i + k - 1)] : b[(size_t)(bl - i + k - 1)];
          const char y = phase == 2 ? b[(size_t)k] : a[(size_t)k];
          if (x == y && y != 'N') score += 1.0f;
          if ((float)k - score > (float)mm) {
            score = -1.0f;
            broke = true;
            break;
          }
        }
        if (!broke) k = i + 1;
        const float pct = k ? score / (float)k : 0.0f;
        const bool ok = (phase == 3 && strict3) ? pct > min_pct : pct >= min_pct;
        if (ok && (float)best < score) {
          best = (int)score;
          ovl = phase == 2 ? i - al + 1 : bl - i - 1;
          if (score == (float)i) break;
        }
      }
  out[3] = best;
  out[4] = ovl;
}

static std::string revcomp_acgtn(const std::string& s) {  // Util::RevComp: other characters vanish
  std::string r;
  for (size_t i = s.size(); i-- > 0;) {
    switch (s[i]) {
      case 'A': r.push_back('T'); break;
      case 'C': r.push_back('G'); break;
      case 'G': r.push_back('C'); break;
      case 'T': r.push_back('A'); break;
      case 'N': r.push_back('N'); break;
      default: break;
    }
  }
  return r;
}

extern "C" {

const char* rfx_last_error(void) { return "host stand-in"; }
rfx_ctx* rfx_open(int device, size_t) { return new rfx_ctx{device}; }
void rfx_close(rfx_ctx* c) { delete c; }

rfx_ovl_pool* rfx_ovl_pool_create(rfx_ctx*, const char* const* seqs, const int* lens, int n) {
  rfx_ovl_pool* p = new rfx_ovl_pool;
  for (int i = 0; i < n; ++i) p->seq.emplace_back(seqs[i], (size_t)lens[i]);
  return p;
}
int rfx_ovl_pool_set(rfx_ovl_pool* p, int idx, const char* seq, int len) {
  if (idx < 0 || idx >= (int)p->seq.size()) return RFX_E_INVAL;
  p->seq[(size_t)idx].assign(seq, (size_t)len);
  return RFX_OK;
}
int rfx_ovl_pool_score(rfx_ovl_pool* p, int query, const char* a_explicit, int a_len, const int* cand, int nb, float min_pct,
                       int min_ovl, int variant, int strands, int* out) {
  if (!a_explicit && (query < 0 || query >= (int)p->seq.size())) return RFX_E_INVAL;
  const std::string a = a_explicit ? std::string(a_explicit, (size_t)a_len) : p->seq[(size_t)query];
  const bool strict3 = variant == RFX_OVL_CONTIG;
  const int init = variant == RFX_OVL_CONTIG ? -1 : 0;
  int* o = out;
  for (int s = 0; s < (strands == 2 ? 2 : 1); ++s) {
    const bool rc = strands == 1 || (strands == 2 && s == 1);
    const std::string q = rc ? revcomp_acgtn(a) : a;
    for (int j = 0; j < nb; ++j, o += 5) {
      if (cand[j] < 0 || cand[j] >= (int)p->seq.size()) return RFX_E_INVAL;
      align3_one(q, p->seq[(size_t)cand[j]], min_pct, min_ovl, strict3, init, o);
    }
  }
  return RFX_OK;
}
void rfx_ovl_pool_free(rfx_ovl_pool* p) { delete p; }

#if OVL_WHICH == 0 || OVL_WHICH == 3
// OverlapSam also scans its reads for mutant k-mers (the same stand-ins as tests/host/filter_sam_harness.cpp)
rfx_set* rfx_set_build(rfx_ctx*, const uint64_t* fwd_keys, uint64_t n, int k) {
  rfx_set* s = new rfx_set;
  s->k = k;
  s->keys.insert(fwd_keys, fwd_keys + n);
  return s;
}
void rfx_set_free(rfx_set* s) { delete s; }
rfx_reads* rfx_reads_upload(rfx_ctx*, const uint64_t* codes, const uint32_t*, const uint32_t* good, const uint32_t* word_off,
                            const uint32_t* len, uint32_t n_reads) {
  rfx_reads* r = new rfx_reads;
  const uint32_t words = word_off[n_reads];
  r->codes.assign(codes, codes + words);
  r->good.assign(good, good + words);
  r->woff.assign(word_off, word_off + n_reads + 1);
  r->len.assign(len, len + n_reads);
  return r;
}
void rfx_reads_free(rfx_reads* r) { delete r; }
int rfx_filter(rfx_set* s, const rfx_reads* r, int thresh, int last_base_skipped, uint32_t* hits_out, uint64_t* hitmask_out,
               uint64_t* n_hit_reads) {
  const int k = s->k;
  const uint64_t kmask = k >= 32 ? ~0ull : (1ull << (2 * k)) - 1;
  const size_t n = r->len.size();
  uint64_t over = 0;
  if (hitmask_out)
    for (size_t w = 0; w < (n + 63) / 64; ++w) hitmask_out[w] = 0;
  for (size_t x = 0; x < n; ++x) {
    const uint32_t L = r->len[x], w0 = r->woff[x];
    const uint32_t stop = last_base_skipped ? (L ? L - 1 : 0) : L;
    uint64_t key = 0;
    int streak = 0;
    uint32_t found = 0;
    for (uint32_t i = 0; i < stop; ++i) {
      const uint64_t code = (r->codes[w0 + i / 32] >> (2 * (i % 32))) & 3u;
      const bool good = (r->good[w0 + i / 32] >> (i % 32)) & 1u;
      key = ((key << 2) | code) & kmask;
      streak = good ? streak + 1 : 0;
      if (streak >= k && s->keys.count(key)) ++found;
    }
    if (hits_out) hits_out[x] = found;
    if (found >= (uint32_t)thresh) {
      ++over;
      if (hitmask_out) hitmask_out[x >> 6] |= 1ull << (x & 63);
    }
  }
  if (n_hit_reads) *n_hit_reads = over;
  return RFX_OK;
}
// coverage of every base by mutant windows: the same scan (good streak >= k, the last base never examined), a hit adds
// one to each of its k bases; contig after contig
int rfx_annotate(rfx_set* s, const rfx_reads* r, uint32_t* cov_out) {
  const int k = s->k;
  const uint64_t kmask = k >= 32 ? ~0ull : (1ull << (2 * k)) - 1;
  size_t base = 0;
  for (size_t x = 0; x < r->len.size(); ++x) {
    const uint32_t L = r->len[x], w0 = r->woff[x];
    for (uint32_t i = 0; i < L; ++i) cov_out[base + i] = 0;
    uint64_t key = 0;
    int streak = 0;
    for (uint32_t i = 0; i + 1 < L; ++i) {
      const uint64_t code = (r->codes[w0 + i / 32] >> (2 * (i % 32))) & 3u;
      const bool good = (r->good[w0 + i / 32] >> (i % 32)) & 1u;
      key = ((key << 2) | code) & kmask;
      streak = good ? streak + 1 : 0;
      if (streak >= k && s->keys.count(key))
        for (int j = 0; j < k; ++j) cov_out[base + i - (uint32_t)k + 1 + (uint32_t)j] += 1;
    }
    base += L;
  }
  return RFX_OK;
}
#endif

}  // extern "C"
