// "MSP" count path (minimizer super-k-mer partition), the default for 23 <= k <= 25.
//
// The P2L path writes every k-mer instance to HBM as an 8-byte sortable word (twice: coarse and fine
// partition) and reads it back once: 32 B of traffic per instance, and one GF(2) LUT multiply per
// instance.  Consecutive k-mers of a read overlap in k-1 bases, so here they travel together:
//
//   k_msp_part1  reads -> for every k-mer the minimum of hash(canonical m-mer) over its 11 m-mers
//                (k <= 25: 11 m-mers, m = k-10; k >= 26: k-15 m-mers, m = 16); the minimizer picks one of P bins.  A super-k-mer --
//                ALL consecutive k-mers of the read with the same minimizer, up to 11 (k <= 25) -- becomes ONE
//                record: 64-bit word + 32-bit plane (rfx_devutil.h).  ~5.7 k-mers per record -> 2.1 B per
//                instance.  128 coarse bins, filled in per-workgroup slabs.
//   k_part2      coarse -> fine bins (same kernel as P2L; the bin is re-derived from the record: ONE m-mer
//                hash, because the record says where its minimizer sits -- so a partition can be refined
//                to any depth by plain streaming passes, whatever the size of the sample)
//   k_msp_leaf   one workgroup per fine bin: expand the records, count canonical k-mers in an LDS
//                hash table (identical records are merged in a small cache first); every instance of
//                a k-mer has the same minimizer, so counts are final.
//                Only the survivors (lower <= count <= upper) get w = T * key and leave, appended to
//                128 coarse pos bins.
//   k_surv_hist, k_part2<payload>, k_surv_sort
//                the (few) survivors are partitioned by pos prefix and sorted in LDS straight into the
//                output records (sizes are exact: no compaction pass).
//
// The result is the same sorted record list (jf/include/jellyfish/sorted_dumper.hpp:80-112 order).
#include <algorithm>
#include <type_traits>

#include "rfx_devutil.h"
#include "rfx_internal.h"

#ifdef RFX_TIMING
// (a row per workgroup: with ONE row every probe of every workgroup met on the same cache line and the kernels took 2 x as long)
__device__ unsigned long long g_tm[2048][32];
#define TM_ROW g_tm[blockIdx.x & 2047u]
#define TM_DECL unsigned long long tm_prev = __builtin_amdgcn_s_memtime()
#define TM(i) do { if (threadIdx.x == 0) { const unsigned long long tm_now = __builtin_amdgcn_s_memtime(); atomicAdd(&TM_ROW[i], tm_now - tm_prev); tm_prev = tm_now; } } while (0)
extern "C" int rfx_debug_timing(unsigned long long* out, int reset) {
  static unsigned long long h[2048][32];
  if (out) {
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tm), sizeof(g_tm)) != hipSuccess) return -1;
    for (int i = 0; i < 32; ++i) {
      out[i] = 0;
      for (int w = 0; w < 2048; ++w) out[i] += h[w][i];
    }
  }
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tm)) != hipSuccess || hipMemset(p, 0, sizeof(g_tm)) != hipSuccess) return -1;
  }
  return 0;
}
#define TMC(i, c) do { if (threadIdx.x == 0 && (c)) atomicAdd(&TM_ROW[i], 1ull); } while (0)
#define TMF_DECL unsigned long long tf_prev = __builtin_amdgcn_s_memtime()
#define TMF(i) do { if (threadIdx.x == 0) { const unsigned long long tf_now = __builtin_amdgcn_s_memtime(); atomicAdd(&TM_ROW[i], tf_now - tf_prev); tf_prev = tf_now; } } while (0)
#else
#define TM_DECL
#define TM(i)
#define TMC(i, c)
#define TMF_DECL
#define TMF(i)
#endif

namespace {

// HMODE 0: scatter into fixed-capacity coarse bins + 16-bit fine histogram (one pass, optimistic; <= 8192 bins)
//       3: the same with 64 KB of LDS for the histogram: up to 32768 bins (big read blocks)
//       1: 32-bit fine histogram only            } the exact redo after HMODE 0 raised its flag;
//       2: scatter only                          } cap_a then comes from the cursors of the failed run
// 512 threads = reads per chunk.  The kernel needs 82 VGPRs = 5 waves per SIMD: two 512-thread workgroups per CU
// (4 waves per SIMD) fit, two of 768 (6) do not -- and with ONE 768-thread workgroup per CU (3 waves per SIMD) the
// VALUs starve.  Measured on the 30x trio of a 1 Gb genome, ms per sample: 384 / 448 / 512 / 576 / 640 / 768 threads
// = 105 / 93 / 90 / 132 / 123 / 110 (768 was the choice of round 1, made on a cache-resident 1 M-read sample when
// the kernel still fit 80 VGPRs; forcing 6 waves per SIMD, `__launch_bounds__(768, 6)`, gives 89).
constexpr int MP1_BLOCK = 512;
// WL: m-mers per k-mer (11: k <= 26; k - 15: k = 27 .. 31).  A record = 64-bit word + 32-bit plane (rfx_devutil.h).
//       4: no record at all -- the launch leaves the block's RUN MAP (see k_msp_replay below): per read 32 bytes that say
//          how it falls into super-k-mers and where their minimizers sit.  With S > 1 shard passes every pass's records
//          are then cut from reads + map without hashing a base again.  (The hashing, the sliding minimum and the run
//          boundaries of this kernel; no close queue, no slabs, no histogram, no "whose bin".)
template <bool CANON, int HMODE, int WL>
__global__ __launch_bounds__(MP1_BLOCK) void k_msp_part1(rfx_reads_view rv, int k, int bin_bits, uint32_t bin_lo,
                                                         uint32_t bin_hi, msp_rec12* __restrict__ rec_a,
                                                         uint32_t* __restrict__ coarse_cur, uint32_t cap_a,
                                                         uint32_t* __restrict__ cnt_rows,
                                                         unsigned int* __restrict__ flag, int slab_log2,
                                                         uint4* __restrict__ map_out, uint32_t* __restrict__ map_ovf,
                                                         uint32_t map_ovf_cap) {
  constexpr bool MAPONLY = HMODE == 4;
  // Two of each, by phase parity (the second barrier of a phase then only has to cover the addresses).  Tried in
  // round 3 and dropped: reserving phase p's runs while phase p + 1 is hashed and storing p's records one phase late
  // (one barrier per phase, nobody waits for the atomic) -- 126 VGPRs, 94 instead of 90 ms per 1 Gb sample: the other
  // workgroup of the CU already covers the round trip, the kernel is bound by what it issues.
  constexpr bool SLABS = HMODE == 0 || HMODE == 3;
  __shared__ uint32_t s_cnt[SLABS ? 1 : 2][SLABS ? 1 : P1_BINS];
  __shared__ uint64_t s_gbase[SLABS ? 1 : 2][SLABS ? 1 : P1_BINS];  // 64-bit: the exact redo may size a coarse bin by a skewed maximum
  // HMODE 0 / 3 (the optimistic one-pass scatter): SLABS.  Round 2 reserved one run per coarse bin and phase with a
  // global atomic BETWEEN two barriers -- the whole workgroup waited out its round trip 19 times per read, and the
  // records of a phase sat in 24 registers until the addresses were known (-DRFX_TIMING: reserve + barriers 35 %,
  // stores 24 % of the kernel; hashing alone, -DRFX_P1_NOCLOSE: 48 of 94 ms).  Now a workgroup owns, per coarse bin,
  // a slab of 2^slab_log2 record slots plus a spare one plus a third whose reservation is in flight: a closing lane
  // takes slot = atomicAdd(s_fill[bin]) in LDS and stores its record at once; every few phases the bins whose slab
  // filled up move on (spare -> current, in-flight -> spare, a new reservation is issued and not waited for).  A bin
  // that gets more than two slabs' worth between two such points falls back to one global atomic per record.  The
  // unused tails are filled with MSP_EMPTY at the end, k_part2 / k_surv_hist skip those.
  static_assert(P1_BINS == 128, "the slab bookkeeping below shifts by log2(P1_BINS) = 7");
  __shared__ uint32_t s_fill[SLABS ? P1_BINS : 1];
  __shared__ uint64_t s_slab[SLABS ? 3 : 1][SLABS ? P1_BINS : 1];  // [0] current, [1] spare ([2]: the in-flight one, at the end)
  // (HMODE 4: the lane's 32 map bytes are staged here, stride 9 dwords: no two lanes of a wave on a bank)
  __shared__ uint32_t s_fine[HMODE == 0 ? 4096 : HMODE == 1 ? 8192 : HMODE == 3 ? 16384 : HMODE == 4 ? MP1_BLOCK * 9 : 1];
  uint8_t* const mapb = MAPONLY ? (uint8_t*)(s_fine + threadIdx.x * 9u) : nullptr;
  __shared__ uint32_t s_maxlen;
  // SLABS (round 4): CLOSE QUEUE.  With whole super-k-mers one lane in six closes a run at any base, so the ~60
  // instructions that turn a closed run into a record (word + plane, bin, slab slot, two stores, fine histogram) ran at
  // nearly every base with a sixth of the lanes live.  A closing lane now only pushes (history, length, where the
  // minimizer sits: 3 dwords) onto its wave's ring in LDS -- ballot + mbcnt, no atomic -- and whenever 64 are queued the
  // WHOLE wave pops one each.  (At four k-mers per record, one lane in three, this was measured in round 3 and gained
  // nothing; the bin is re-derived from the record here, so the ring holds 12 bytes per entry: 12 KB per workgroup.)
  constexpr int QN = 128;  // entries per wave: < 64 left after a drain + <= 64 pushed by one base
  __shared__ uint32_t s_q[SLABS ? MP1_BLOCK / 64 : 1][SLABS ? 3 * QN : 1];
  uint32_t* const sq = SLABS ? s_q[threadIdx.x >> 6] : nullptr;
  uint32_t q_head = 0, q_tail = 0;  // (wave-uniform)
  const uint32_t P = 1u << bin_bits;
  const int sub_bits = bin_bits - 7;  // P1_BINS = 2^7 coarse bins
  const int m = k - (WL - 1);
  const uint32_t mmask = m >= 16 ? ~0u : (1u << (2 * m)) - 1;
  const int rmshift = 2 * (m - 1);
  const int nmax = msp_nmax(k);
  const bool all_mine = bin_lo == 0 && bin_hi == (1u << bin_bits);
  uint32_t n_emit = 0;  // records this thread stored (or dropped over capacity)
  if (HMODE == 0)
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) s_fine[i] = 0;
  if (HMODE == 1)
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) s_fine[i] = 0;
  if (HMODE == 3)
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) s_fine[i] = 0;
  if (!SLABS && threadIdx.x < 2 * P1_BINS) (&s_cnt[0][0])[threadIdx.x] = 0;
  const uint32_t SLAB = 1u << slab_log2;
  const uint32_t tick_mask = SLAB >= 128 ? 3u : SLAB >= 64 ? 1u : 0u;  // bins move on every 4 / 2 / 1 phases
  // threads < P1_BINS: where the slab reserved ahead for their bin starts (the raw value of the cursor: it is looked at
  // only when the slab is taken into use, a turn later -- looking at it at once made the two waves of these threads
  // wait for the atomic, and the workgroup for them at the next barrier, at every turn); P1_NO_SLAB: none
  constexpr uint32_t P1_NO_SLAB = 0xFFFFFFFFu;
  uint32_t pending = P1_NO_SLAB;
  uint32_t tick = 0;
  auto reserve_raw = [&](uint32_t b) -> uint32_t { return atomicAdd(&coarse_cur[b * P1_CUR_STRIDE], SLAB); };
  auto slab_at = [&](uint32_t b, uint32_t at) -> uint64_t {
    if (at == P1_NO_SLAB) return ~0ull;
    if ((uint64_t)at + SLAB > cap_a) {  // over capacity: the host redoes the block; until then records of this bin land
      atomicExch(flag, 1u);             // on its first slab (inside the bin's memory: no test on the store path)
      return (uint64_t)b * cap_a;
    }
    return (uint64_t)b * cap_a + at;
  };
  if (SLABS && threadIdx.x < P1_BINS) {
    // only the coarse bins this shard's records can fall into have memory behind them
    const bool used = bin_hi > bin_lo && threadIdx.x >= (bin_lo >> sub_bits) && threadIdx.x <= ((bin_hi - 1) >> sub_bits);
    s_fill[threadIdx.x] = 0;
    s_slab[0][threadIdx.x] = used ? slab_at(threadIdx.x, reserve_raw(threadIdx.x)) : ~0ull;
    s_slab[1][threadIdx.x] = used ? slab_at(threadIdx.x, reserve_raw(threadIdx.x)) : ~0ull;
    pending = used ? reserve_raw(threadIdx.x) : P1_NO_SLAB;
  }
  // the wave's first `cnt` queued runs -> records -> slabs (lane l takes entry q_head + l)
  auto drain = [&](uint32_t cnt) {
    const uint32_t l = threadIdx.x & 63u;
    if (l < cnt) {
      const uint32_t idx = (q_head + l) & (QN - 1);
      const uint64_t h64 = (uint64_t)sq[idx] | ((uint64_t)sq[QN + idx] << 32);
      const uint32_t meta = sq[2 * QN + idx];
      const int n = (int)((meta >> 22) & 15u);
      const uint32_t back = meta >> 26;
      uint64_t w;
      uint32_t x;
      msp_record_make(h64, meta & 0x3FFFFFu, k, n, (uint32_t)(k + n - 1 - m) - back, w, x);
      // the bin, from the record: the same bits msp_bin(run_h) gave when the lane decided the run was this shard's
      const uint32_t bh = msp_record_binhash<CANON>(w, k);
      x |= msp_stamp(bh, k);
      const uint32_t run_bin = bh >> (32 - bin_bits);
      const uint32_t coarse = run_bin >> sub_bits;
      const uint32_t slot = atomicAdd(&s_fill[coarse], 1u);
      if (slot < 2 * SLAB) {  // (a bin of ours always has a slab behind it: slab_at)
        const uint64_t sb = s_slab[slot >> slab_log2][coarse];
#ifdef RFX_P1_NOSTORE  // experiment: everything but the record stores (results void)
        if (w == 0x123456789ull)
#endif
          msp_rec12_store(rec_a, sb + (slot & (SLAB - 1)), w, x);
      } else {  // more than two slabs' worth since the bins last moved on: one reservation per record
        const uint32_t at = atomicAdd(&coarse_cur[coarse * P1_CUR_STRIDE], 1u);
        if (at < cap_a) {
          msp_rec12_store(rec_a, (uint64_t)coarse * cap_a + at, w, x);
        } else {
          atomicExch(flag, 1u);
        }
      }
      ++n_emit;
      atomicAdd(&s_fine[run_bin >> 1], 1u << ((run_bin & 1u) * 16));
    }
    q_head += cnt;
  };
  const uint32_t n_chunks = (rv.n + MP1_BLOCK - 1) / MP1_BLOCK;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r0 = chunk * MP1_BLOCK + threadIdx.x;
    const bool live = r0 < rv.n;
    const uint32_t r = live && rv.idx ? rv.idx[r0] : r0;  // (rv.idx: a list of reads, e.g. those without a run map)
    const uint32_t len = live ? rv_len(rv, r) : 0;
    const uint32_t lenp = live ? len + 1 : 0;  // one virtual invalid base closes the last run
    const uint32_t roff = live ? rv_off(rv, r) : 0;
    const uint64_t* cw = rv.codes + roff;
    const uint32_t* cm = live ? rv_acgt(rv, r, roff) : nullptr;  // nullptr: all of the read is A/C/G/T (compact blocks)
    // HMODE 4: entries so far, the k-mer end position the next entry begins at, the entries' shard-half flags
    uint32_t mcnt = 0, mpos = (uint32_t)k - 1u, mflags = 0;
    if (threadIdx.x == 0) s_maxlen = 0;
    __syncthreads();
    atomicMax(&s_maxlen, lenp);
    __syncthreads();
    if constexpr (MAPONLY) {
#pragma unroll
      for (int i = 0; i < 8; ++i) ((uint32_t*)mapb)[i] = ~0u;
    }
#ifdef RFX_P1_HALF  // experiment (results void): half of every read -- what is per chunk, what per base?
    const uint32_t n_phase = (s_maxlen + P1_S - 1) / P1_S / 2;
#else
    const uint32_t n_phase = (s_maxlen + P1_S - 1) / P1_S;
#endif
    // Per-lane state between phases: the rolling m-mer and its reverse complement, the hash window, what is left of the
    // 32 bases in flight, the last 64 + 64 bases (`hist`: the newest in the low bits), the base from which k-mers are
    // valid again (`ok_from`: k past the last invalid one), whether the k-mer ending at the phase's last base was
    // valid and its minimizer (`prev_kv`, `prev_mh`: the run in progress), and where that run began (`run_s`: the
    // last base of its first k-mer).
    uint32_t fm = 0, rm = 0, cur_m = 0, prev_mh = 0, prev_kv = 0, run_s = 0, ok_from = (uint32_t)k - 1u;
    uint64_t hist = 0, hist_hi = 0, cur_w = 0;
    uint32_t a[WL - 1 + P1_S];  // hashes of the m-mers ending at bases p0-(WL-1) .. p0+7
#pragma unroll
    for (int i = 0; i < WL - 1 + P1_S; ++i) a[i] = ~0u;
    // Reads of up to 160 bases are loaded whole, here: a load in the middle of the chunk has to wait for every record
    // store issued before it (vmcnt counts loads and stores in one queue), five times per 150 bp read.
    constexpr int PW = 5;
    const bool whole = s_maxlen <= PW * 32 + 1;  // (uniform over the workgroup)
    uint64_t wq[PW];
    uint32_t mq[PW];
    if (whole) {
#pragma unroll
      for (int i = 0; i < PW; ++i) {
        wq[i] = (uint32_t)i * 32u < len ? cw[i] : 0;
        mq[i] = cm ? ((uint32_t)i * 32u < len ? cm[i] : 0u) : ~0u;
      }
      // the wait for these loads belongs HERE (an empty asm that "uses" the registers), not at their first use in the loop
#pragma unroll
      for (int i = 0; i < PW; ++i) asm volatile("" : "+v"(wq[i]), "+v"(mq[i]));
    }
    TM_DECL;
    auto phases = [&](auto whole_tag) {
    constexpr bool WHOLE = decltype(whole_tag)::value;
    for (uint32_t ph = 0; ph < n_phase; ++ph) {
      TM(3);
      const uint32_t X = ph & 1u;
      uint64_t wv[P1_S];
      uint32_t xv[P1_S];
      uint32_t br[P1_S];  // (coarse bin << 16) | rank, or ~0: no record closed at this base
#pragma unroll
      for (int b = 0; b < P1_S; ++b) br[b] = ~0u;
      {
        const uint32_t p0 = ph * P1_S;
        if ((ph & 3) == 0) {
          if (WHOLE) {  // (past the end of a short read: zeros nobody looks at)
            cur_w = wq[0];
            cur_m = mq[0];
#pragma unroll
            for (int i = 0; i + 1 < PW; ++i) {
              wq[i] = wq[i + 1];
              mq[i] = mq[i + 1];
            }
          } else if (p0 < len) {
            cur_w = cw[p0 >> 5];
            cur_m = cm ? cm[p0 >> 5] : ~0u;
          }
        }
        // ---- round 4: run boundaries are found for the eight bases of a phase at once ----
        // Until then every base carried the run state through ~35 instructions (valid-base counter, run length, four
        // compares, selects): 38 of the kernel's 57 ms per 1 Gb sample with the record stores out of the way.  Now a base
        // costs the hash and two minima; what depends on validity and on "same minimizer as the base before" is
        // worked out once per phase on 8-bit masks.
        // (1) which of the 8 k-mers ending here are valid: from the first invalid base on none is (k > 8)
        uint32_t vm8 = cur_m & 0xFFu;
        cur_m >>= 8;
        if (!__all((int)(p0 + P1_S <= len))) vm8 &= p0 < len ? (len - p0 >= 8u ? 0xFFu : (1u << (len - p0)) - 1u) : 0u;
        const uint32_t inv = ~vm8 & 0xFFu;
        const uint32_t fi = inv ? (uint32_t)__ffs((int)inv) - 1u : 8u;
        const uint32_t lo_b = ok_from > p0 ? min(ok_from - p0, 8u) : 0u;
        const uint32_t kvmask = ((1u << fi) - 1u) & ~((1u << lo_b) - 1u);  // bit b: the k-mer ending at p0 + b is valid
        if (inv) ok_from = p0 + (31u - (uint32_t)__clz((int)inv)) + (uint32_t)k;
        // (2) the 8 bases join the history (first base most significant: the packed word has it least significant)
        const uint32_t w16 = (uint32_t)cur_w & 0xFFFFu;
        cur_w >>= 16;
        {
          uint32_t y = __brev(w16) >> 16;
          y = ((y & 0xAAAAu) >> 1) | ((y & 0x5555u) << 1);
          hist_hi = (hist_hi << 16) | (hist >> 48);
          hist = (hist << 16) | y;
        }
        // (3) hashes and window minima; bit b of eqmask: same minimizer as the k-mer before
        // window minimum = min(suffix minimum of the old hashes, prefix minimum of the new ones)
        uint32_t sfx[WL - 1];
        sfx[WL - 2] = a[WL - 2];
  #pragma unroll
        for (int i = WL - 3; i >= 0; --i) sfx[i] = min(a[i], sfx[i + 1]);
        uint32_t pm = ~0u, eqmask = 0, mhv[P1_S];
  #pragma unroll
        for (int b = 0; b < P1_S; ++b) {
          const uint32_t code = (w16 >> (2 * b)) & 3u;
          fm = ((fm << 2) | code) & mmask;
          rm = (rm >> 2) | ((3u ^ code) << rmshift);
          const uint32_t h = (mmer_hash(CANON ? min(fm, rm) : fm) & MSP_HMASK) | ((p0 & 24u) | (uint32_t)b);
          a[WL - 1 + b] = h;
          pm = min(pm, h);
          // A run = consecutive k-mers with the same minimizer HASH (not merely the same bin): every further bit of
          // that hash is then common to the record's k-mers, which is what lets the partition be refined later
          // without separating instances of a k-mer.
          mhv[b] = min(sfx[b], pm);
          eqmask |= (mhv[b] == (b ? mhv[b - 1] : prev_mh) ? 1u : 0u) << b;
        }
        // (4) a run goes on from base b - 1 to b if both k-mers are valid and share the minimizer; where it does not, the
        // run of b - 1 ends (bit b of `ends`) and / or one begins at b (bit b of `starts`)
        const uint32_t KV = (kvmask << 1) | prev_kv;  // bit b: the k-mer ending at p0 + b - 1 is valid
        uint32_t cont = KV & (KV >> 1) & eqmask;
        uint32_t ends = KV & ~cont & 0xFFu, starts = kvmask & ~cont;
        if (nmax < WL) {  // k >= 26: a record holds fewer k-mers than a window has m-mers -- the run is cut at the cap
          // (at most once per phase: nmax > 8; and only if the run in progress did not begin in this phase)
          const uint32_t cb = run_s + (uint32_t)nmax - p0;  // (wraps to a big number when the cap lies behind)
          if (cb < 8u && ((cont >> cb) & 1u) && (starts & ((2u << cb) - 1u)) == 0) {
            ends |= 1u << cb;
            starts |= 1u << cb;
          }
        }
        // (5) the runs that end in this phase.  SLABS: a lane walks the set bits of its `ends` -- the wave makes as many
        // turns as its busiest lane has boundaries (3-4 of 8 bases), where a copy of the push per base ran 8 times; what a
        // turn needs of the minimizer (its position mod 32, for `back`; its bin, for "is this run the table's") is laid
        // out per phase: five bits per base in `qpack`, one in `minemask`.
        if constexpr (MAPONLY) {
          // every run that ends in this phase becomes an entry of the map: a lane walks the set bits of its `ends`.  Per
          // run end six bits of the minimizer's hash are laid out ahead: its position (mod 32) and MSP_HALF_BIT, the
          // half of the bin space the run's bin lies in (bases 0 .. 4 in one word, 5 .. 7 in another: 32-bit shifts).
          uint32_t qa = prev_mh & 63u, qb = 0;
  #pragma unroll
          for (int b = 1; b < 5; ++b) qa |= (mhv[b - 1] & 63u) << (6 * b);
  #pragma unroll
          for (int b = 5; b < P1_S; ++b) qb |= (mhv[b - 1] & 63u) << (6 * (b - 5));
          uint32_t pend = ends;
          while (__ballot(pend != 0)) {
            if (pend) {
              const uint32_t b = (uint32_t)__ffs((int)pend) - 1u;
              pend &= pend - 1u;
              const uint32_t ms = starts & ((1u << b) - 1u);
              const uint32_t rs = ms ? p0 + 31u - (uint32_t)__clz((int)ms) : run_s;
              const uint32_t e = p0 + b - 1u;
              const uint32_t q6 = b < 5u ? qa >> (6u * b) : qb >> (6u * b - 30u);
              const uint32_t back = (e - q6) & 31u;
              // entry: low nibble n - 1 (15: a gap), high nibble `back` (a gap: its length - 1).  K-mer end positions no run
              // covers (invalid bases) are gaps.  A run longer than a record holds (k >= 26) goes in pieces, as the records do.
              // (Entries beyond the map's 27 land on bytes that the flags overwrite: no test on the way.)
              if (rs != mpos) {
                for (uint32_t g = rs - mpos; g != 0;) {
                  const uint32_t l = min(g, 16u);
                  mapb[min(mcnt, 31u)] = (uint8_t)(0xFu | ((l - 1u) << 4));
                  ++mcnt;
                  g -= l;
                }
              }
              const uint32_t at = min(mcnt, 31u);
              mapb[at] = (uint8_t)((e - rs) | (back << 4));
              mflags |= ((q6 >> 5) & 1u) << at;
              ++mcnt;
              mpos = e + 1u;
            }
          }
          if (starts) run_s = p0 + 31u - (uint32_t)__clz((int)starts);
        } else if constexpr (SLABS) {
          uint64_t qpack = prev_mh & 31u;  // 5 bits per run end: position (mod 32) of the minimizer of the run ending at p0 + b - 1
  #pragma unroll
          for (int b = 1; b < P1_S; ++b) qpack |= (uint64_t)(mhv[b - 1] & 31u) << (5 * b);
          uint32_t pend = ends;
  #ifdef RFX_P1_NOCLOSE  // experiment: what the hashing and the sliding minimum cost without the record path
          if (bin_lo != 12345u) pend = 0;
  #endif
          if (!all_mine) {  // (one pass, one device: every bin is this table's)
            uint32_t minemask = 0;
  #pragma unroll
            for (int b = 0; b < P1_S; ++b)
              minemask |= (msp_bin(b ? mhv[b - 1] : prev_mh, bin_bits) - bin_lo < bin_hi - bin_lo ? 1u : 0u) << b;
            pend &= minemask;  // (shard passes: other bins are not ours; unsigned: one compare)
          }
          for (;;) {
            const bool push = pend != 0;
            const unsigned long long bal = __ballot(push);
            if (!bal) break;
            if (push) {
              const uint32_t b = (uint32_t)__ffs((int)pend) - 1u;
              pend &= pend - 1u;
              // the run's k-mers end at bases rs .. e = p0 + b - 1 (rs: the last start before b, in this phase or
              // before); its minimizer m-mer ends at the last base q <= e with q = qpack[b] (mod 32) -- inside the
              // run's LAST k-mer, so back = e - q <= k - m < 16.  The history holds 8 - b bases behind e.
              const uint32_t ms = starts & ((1u << b) - 1u);
              const uint32_t rs = ms ? p0 + 31u - (uint32_t)__clz((int)ms) : run_s;
              const uint32_t e = p0 + b - 1u;
              const uint32_t back = (e - (uint32_t)(qpack >> (5u * b))) & 31u;
              const uint32_t sh = 2u * (P1_S - b);  // 2 .. 16
              const uint64_t hl = (hist >> sh) | (hist_hi << (64u - sh));
              const uint32_t idx =
                  (q_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))) &
                  (QN - 1);
              sq[idx] = (uint32_t)hl;
              sq[QN + idx] = (uint32_t)(hl >> 32);
              sq[2 * QN + idx] = ((uint32_t)(hist_hi >> sh) & 0x3FFFFFu) | ((e - rs + 1u) << 22) | (back << 26);
            }
            q_tail += (uint32_t)__popcll(bal);
            if (q_tail - q_head >= 64u) drain(64u);
          }
          if (starts) run_s = p0 + 31u - (uint32_t)__clz((int)starts);
        } else {
          // the exact redo (rare): base by base, b a constant in each copy
  #pragma unroll
          for (int b = 0; b < P1_S; ++b) {
            const uint32_t run_h = b ? mhv[b - 1] : prev_mh;  // minimizer of the run that ends at p0 + b - 1
            const bool closes = (ends >> b) & 1u;
            const uint32_t run_bin = msp_bin(run_h, bin_bits);
            const bool mine = run_bin - bin_lo < bin_hi - bin_lo;
            const uint32_t e = p0 + (uint32_t)b - 1u;
            const int sh = 2 * (P1_S - b);  // (a constant once the loop is unrolled; 2 .. 16)
            if (closes & mine) {
              if (HMODE != 1) {
                const int n = (int)(e - run_s + 1u), L = k + n - 1;
                const uint32_t coarse = run_bin >> sub_bits;
                const uint32_t mpos = (uint32_t)(L - m) - ((e - run_h) & 31u);
                msp_record_make((hist >> sh) | (hist_hi << (64 - sh)), (uint32_t)(hist_hi >> sh), k, n, mpos, wv[b], xv[b]);
                xv[b] |= msp_stamp(msp_binhash(run_h), k);
                br[b] = (coarse << 16) | atomicAdd(&s_cnt[X][coarse], 1u);
              }
              if (HMODE == 1) atomicAdd(&s_fine[run_bin], 1u);
            }
            run_s = ((starts >> b) & 1u) ? p0 + (uint32_t)b : run_s;
          }
        }
        prev_mh = mhv[P1_S - 1];
        prev_kv = (kvmask >> (P1_S - 1)) & 1u;
  #pragma unroll
        for (int i = 0; i < WL - 1; ++i) a[i] = a[i + P1_S];
      }
      TM(0);
      if (HMODE == 1 || HMODE == 4) continue;
      if constexpr (SLABS) {
        if ((++tick & tick_mask) == 0) {
          __syncthreads();  // every slot of the interval has been handed out
          if (threadIdx.x < P1_BINS) {
            uint32_t f = s_fill[threadIdx.x];
            if (f >= SLAB) {
              if (f >= 2 * SLAB) {  // both slabs are full (what came after them went the slow way)
                s_slab[0][threadIdx.x] = slab_at(threadIdx.x, pending);
                s_slab[1][threadIdx.x] = slab_at(threadIdx.x, reserve_raw(threadIdx.x));
                f = 0;
              } else {
                s_slab[0][threadIdx.x] = s_slab[1][threadIdx.x];
                s_slab[1][threadIdx.x] = slab_at(threadIdx.x, pending);
                f -= SLAB;
              }
              pending = reserve_raw(threadIdx.x);  // looked at when it is taken into use: nobody waits for this one
              s_fill[threadIdx.x] = f;
            }
          }
          __syncthreads();
        }
        TM(1);
        continue;
      }
      uint32_t at_prev = 0, cn_prev = 0;
      auto settle = [&](uint32_t Y) {  // the reserved runs' addresses, for everybody (waits for the atomic)
        if (threadIdx.x < P1_BINS) {
          if ((uint64_t)at_prev + cn_prev > cap_a) {  // over capacity: the run is dropped, the host redoes the block
            atomicExch(flag, 1u);
            s_gbase[Y][threadIdx.x] = ~0ull;
          } else {
            s_gbase[Y][threadIdx.x] = (uint64_t)threadIdx.x * cap_a + at_prev;
          }
        }
      };
      auto reserve = [&](uint32_t Y) {  // one global atomic per coarse bin reserves the phase's run in it
        if (threadIdx.x < P1_BINS) {
          cn_prev = s_cnt[Y][threadIdx.x];
          at_prev = cn_prev ? atomicAdd(&coarse_cur[threadIdx.x * P1_CUR_STRIDE], cn_prev) : 0u;
          s_cnt[Y][threadIdx.x] = 0;  // (next used two phases on: a barrier lies between)
        }
      };
      // Every lane stores its own records straight into the reserved runs (rank inside the run = the value its
      // LDS atomic returned).  The stores of one run come from many lanes, but they fall into the same one or two
      // 128-byte lines within a few hundred cycles and merge in the L2.
      auto store = [&](const uint32_t* brx, const uint64_t* wvx, const uint32_t* xvx, uint32_t Y) {
        uint64_t base[P1_S];
#pragma unroll
        for (int b = 0; b < P1_S; ++b) base[b] = s_gbase[Y][(brx[b] >> 16) & (P1_BINS - 1)];
#pragma unroll
        for (int b = 0; b < P1_S; ++b)
          if (brx[b] != ~0u) {
            if (base[b] != ~0ull) {
              msp_rec12_store(rec_a, base[b] + (brx[b] & 0xFFFFu), wvx[b], xvx[b]);
            }
            ++n_emit;
          }
      };
      __syncthreads();
      reserve(X);
      settle(X);
      __syncthreads();
      TM(1);
      store(br, wv, xv, X);
      TM(2);
      // (s_gbase[X ^ 1] is rewritten two phases on: a barrier lies between)
    }
    };
    // two copies of the loop: the one for whole reads holds no load, so nothing in it waits for the record stores
    // (vmcnt counts loads and stores in one queue: with a load anywhere in the loop the compiler put an
    // `s_waitcnt vmcnt(0)` at the top of EVERY phase, and every phase waited for the scattered stores of the one before)
    if (whole) phases(std::true_type());
    else phases(std::false_type());
    if constexpr (MAPONLY) {
      if (live) {
        uint32_t d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = ((const uint32_t*)mapb)[i];
        // dword 7: the entries' half flags (27 bits), their number in the top 5 (31: more than the map holds -- no map)
        d[7] = mcnt <= RMAP_ENTRIES ? (mflags & ((1u << RMAP_ENTRIES) - 1u)) | (mcnt << 27) : ~0u;
        map_out[2 * (size_t)r] = make_uint4(d[0], d[1], d[2], d[3]);
        map_out[2 * (size_t)r + 1] = make_uint4(d[4], d[5], d[6], d[7]);
        if (mcnt > RMAP_ENTRIES) {  // (~1 % of 150 bp reads: the passes hash these the ordinary way)
          const uint32_t at = atomicAdd(&map_ovf[0], 1u);
          if (at < map_ovf_cap) map_ovf[1 + at] = r;
        }
      }
    }
  }
  if constexpr (SLABS) drain(q_tail - q_head);  // what is left in the wave's queue (< 64)
  __syncthreads();
  if (SLABS) {  // the slots nobody took: MSP_EMPTY
    if (threadIdx.x < P1_BINS) s_slab[2][threadIdx.x] = slab_at(threadIdx.x, pending);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 3u * P1_BINS * SLAB; i += blockDim.x) {
      const uint32_t o = i & (SLAB - 1), bw = i >> slab_log2, b = bw & (P1_BINS - 1), which = bw >> 7;  // P1_BINS = 2^7
      const uint64_t sb = s_slab[which][b];
      if (sb != ~0ull && (which == 2 || which * SLAB + o >= s_fill[b])) msp_rec12_store(rec_a, sb + o, MSP_EMPTY, 0u);
    }
    __syncthreads();
  }
  if (HMODE == 0 || HMODE == 3) {
    __shared__ uint32_t s_emit;
    if (threadIdx.x == 0) {
      s_maxlen = 0;
      s_emit = 0;
    }
    __syncthreads();
    uint32_t sum = 0;
    for (uint32_t b = threadIdx.x; b < P; b += blockDim.x) {
      const uint32_t v = (s_fine[b >> 1] >> ((b & 1u) * 16)) & 0xFFFFu;
      cnt_rows[(uint64_t)blockIdx.x * P + b] = v;
      sum += v;
    }
    // a wrapped 16-bit counter (carry into the neighbour or out of the word) leaves the sum short
    atomicAdd(&s_maxlen, sum);
    atomicAdd(&s_emit, n_emit);
    __syncthreads();
    if (threadIdx.x == 0 && s_maxlen != s_emit) atomicExch(flag, 1u);
  }
  if (HMODE == 1)
    for (uint32_t b = threadIdx.x; b < P; b += blockDim.x) cnt_rows[(uint64_t)blockIdx.x * P + b] = s_fine[b];
}

// ---- shard passes without hashing: the run map --------------------------------------------------------------------
// With S > 1 shard passes every pass used to hash every base (the minimizer of a k-mer says whose it is) -- at W two runs
// of k_msp_part1 were half of what the chain issued.  Now every big block is hashed ONCE (the reference hashes a k-mer
// once too: jf/sub_commands/count_main.cc:148-180), by k_msp_part1<.., HMODE 4>, which leaves per read 32 bytes:
//   bytes 0 .. 26   one entry per stretch of consecutive k-mer end positions, from position k - 1 on:
//                   low nibble n - 1 (the run's k-mers), high nibble `back` (its last k-mer ends `back` bases behind the end
//                   of the minimizer m-mer) -- or low nibble 15: a gap of (high nibble) + 1 positions (invalid bases)
//   dword 7         bits 0 .. 26: entry i's bin lies in the upper half of the bin space (two shard passes: a pass skips
//                   the other's runs without building them); bits 27 .. 31: the number of entries, 31 = the read has more
//                   than 27 (~1 % of 150 bp reads) and is on the block's list of reads every pass hashes the ordinary
//                   way (k_msp_part1 over rv.idx)
// k_msp_replay cuts the records of ITS bins from reads + map: a lane per read walks its entries, cuts the run out
// of the read (staged in LDS with the base order reversed: a run is then a bit field, first base most significant,
// as msp_record_make wants it), re-derives the bin from the record (as the close queue of k_msp_part1 does) and stores
// into the same slabs / counts the same fine histogram as HMODE 3 -- everything behind it is unchanged.
// Reads of up to 160 bases (5 code words).  ~10 wave-instructions per 64 bases where the hashing pass issues 62.
// 8-bit histogram counters (a workgroup puts ~45 records into a fine bin at W; a counter that wraps leaves the sum short
// and the host counts the records instead); two 512-thread workgroups per CU.
// The entries are walked without a loop over the ones to skip: per chunk a lane turns its 27 entries into end positions
// (byte-wise prefix sums of their lengths, SWAR on the 7 dwords) and a bit mask of the runs of this pass's half; a turn
// takes the next set bit.  (The first version stepped over gaps and the other half's runs one by one: 36 wave-
// instructions per 64 bases and pass, 30 of them scalar loop control -- more than half of what the hashing costs.)
constexpr int RP_STRIDE = 25;  // dwords per lane: 10 of read, 7 of entries, 7 of end positions, 1 that keeps the stride odd
// QUEUED (round 6): the runs of a wave's 64 reads go through a per-wave queue and are cut 64 at a time, one per lane.
// With a lane cutting the runs of ITS read, a turn of the loop lasted as long as the read with the most runs of this pass
// had any: 16 - 17 of them where the mean is 11 (half of ~22, binomially), so two lanes in five idled through ~85
// instructions per run.  Now every lane writes its runs' descriptors -- lane | entry index << 6, at most 8 per batch:
// 512 per wave -- to the wave's queue (one LDS add to one address = a wave scan: -amdgpu-atomic-optimizer-strategy=DPP) and
// lane t cuts run t, t + 64, ..: the entry, its end position and the read's dwords come out of the OWNER's staging area.
template <bool CANON, bool QUEUED>
__global__ __launch_bounds__(MP1_BLOCK) void k_msp_replay(rfx_reads_view rv, const uint4* __restrict__ map, int k, int bin_bits,
                                                          uint32_t bin_lo, uint32_t bin_hi, msp_rec12* __restrict__ rec_a,
                                                          uint32_t* __restrict__ coarse_cur, uint32_t cap_a,
                                                          uint32_t* __restrict__ cnt_rows, unsigned int* __restrict__ flag,
                                                          int slab_log2) {
  static_assert(P1_BINS == 128, "the slab bookkeeping below shifts by log2(P1_BINS) = 7");
  __shared__ uint32_t s_fill[P1_BINS];
  __shared__ uint64_t s_slab[3][P1_BINS];
  __shared__ uint32_t s_fine[4096];  // 8-bit counters of bins bin_lo .. bin_lo + 16383 (a shard pass holds at most half of 32768)
  __shared__ uint32_t s_rd[MP1_BLOCK * RP_STRIDE + 4];
  __shared__ uint32_t s_sum, s_emit;
  constexpr int RQ_BATCH = 8, RQ_CAP = 64 * RQ_BATCH;  // runs a lane queues per batch; descriptors per wave
  __shared__ uint16_t s_q[QUEUED ? MP1_BLOCK / 64 : 1][QUEUED ? RQ_CAP : 1];
  __shared__ uint32_t s_qn[QUEUED ? MP1_BLOCK / 64 : 1];
  const uint32_t P = 1u << bin_bits;
  const int sub_bits = bin_bits - 7;
  const int m = msp_m(k);
  const uint32_t mmask = m >= 16 ? ~0u : (1u << (2 * m)) - 1;
  // which half of the bin space this pass's bins lie in (2: both -- the entries' flags do not help)
  const uint32_t half_sel = bin_lo >= P / 2 ? 1u : bin_hi <= P / 2 ? 0u : 2u;
  uint32_t n_emit = 0;
  for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) s_fine[i] = 0;
  // (the slabs: as in k_msp_part1, see there)
  const uint32_t SLAB = 1u << slab_log2;
  constexpr uint32_t P1_NO_SLAB = 0xFFFFFFFFu;
  uint32_t pending = P1_NO_SLAB;
  auto reserve_raw = [&](uint32_t b) -> uint32_t { return atomicAdd(&coarse_cur[b * P1_CUR_STRIDE], SLAB); };
  auto slab_at = [&](uint32_t b, uint32_t at) -> uint64_t {
    if (at == P1_NO_SLAB) return ~0ull;
    if ((uint64_t)at + SLAB > cap_a) {
      atomicExch(flag, 1u);
      return (uint64_t)b * cap_a;
    }
    return (uint64_t)b * cap_a + at;
  };
  if (threadIdx.x < P1_BINS) {
    const bool used = bin_hi > bin_lo && threadIdx.x >= (bin_lo >> sub_bits) && threadIdx.x <= ((bin_hi - 1) >> sub_bits);
    s_fill[threadIdx.x] = 0;
    s_slab[0][threadIdx.x] = used ? slab_at(threadIdx.x, reserve_raw(threadIdx.x)) : ~0ull;
    s_slab[1][threadIdx.x] = used ? slab_at(threadIdx.x, reserve_raw(threadIdx.x)) : ~0ull;
    pending = used ? reserve_raw(threadIdx.x) : P1_NO_SLAB;
  }
  __syncthreads();
  uint32_t* const rd = s_rd + threadIdx.x * RP_STRIDE;
  const uint8_t* const ent8 = (const uint8_t*)(rd + 10);
  const uint8_t* const end8 = (const uint8_t*)(rd + 17);
  const uint32_t n_chunks = (rv.n + MP1_BLOCK - 1) / MP1_BLOCK;
  // the next chunk's map and read are on their way while this one is cut
  uint4 nq0 = make_uint4(0, 0, 0, 0), nq1 = make_uint4(0, 0, 0, 31u << 27);
  uint64_t nw[5] = {0, 0, 0, 0, 0};
  auto fetch = [&](uint32_t chunk) {
    const uint32_t r = chunk * MP1_BLOCK + threadIdx.x;
    nq1.w = 31u << 27;  // (no read: no entry)
    if (chunk < n_chunks && r < rv.n) {
      nq0 = map[2 * (size_t)r];
      nq1 = map[2 * (size_t)r + 1];
      const uint32_t len = rv_len(rv, r);
      const uint64_t* cw = rv.codes + rv_off(rv, r);
#pragma unroll
      for (int i = 0; i < 5; ++i) nw[i] = (uint32_t)i * 32u < len ? cw[i] : 0;
    }
  };
  fetch(blockIdx.x);
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    // base i of the read -> bit pair 159 - i of the staged 320 bits
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      uint64_t y = __brevll(nw[4 - j]);
      y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
      rd[2 * j] = (uint32_t)y;
      rd[2 * j + 1] = (uint32_t)(y >> 32);
    }
    const uint32_t q[7] = {nq0.x, nq0.y, nq0.z, nq0.w, nq1.x, nq1.y, nq1.z};
    const uint32_t flags = nq1.w;
    uint32_t cnt = flags >> 27;
    if (cnt == 31u) cnt = 0;  // no map: the read is on the list of the ordinary launch that follows
    fetch(chunk + gridDim.x);
    // end positions of the entries (as offsets from k - 2: byte i = sum of the lengths of entries 0 .. i) and which are gaps
    uint32_t gaps = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const uint32_t lo = q[j] & 0x0F0F0F0Fu, hi = (q[j] >> 4) & 0x0F0F0F0Fu;
      const uint32_t g = ((lo + 0x01010101u) >> 4) & 0x01010101u;  // 1 in the bytes whose low nibble is 15: a gap
      const uint32_t gm = (g << 8) - g;                              // 0xFF there
      uint32_t p = ((hi & gm) | (lo & ~gm)) + 0x01010101u;           // the entries' lengths: (high nibble | n - 1) + 1
      p += p << 8;
      p += p << 16;
      p += tot * 0x01010101u;  // (a read has at most 160 positions: no byte overflows before the last entry)
      tot = p >> 24;
      rd[10 + j] = q[j];
      rd[17 + j] = p;
      gaps |= ((g * 0x01020408u) >> 24) << (4 * j);
    }
    // (a lane reads only what it wrote itself: no barrier)
    uint32_t mine = ~gaps & ((1u << cnt) - 1u);
    if (half_sel != 2u) mine &= half_sel ? flags : ~flags;
    if constexpr (QUEUED) {
      const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
      uint16_t* const wq = s_q[wv];
      const uint32_t* const wrd = s_rd + (size_t)(wv * 64u) * RP_STRIDE;  // the wave's 64 staging areas
      while (__ballot(mine != 0)) {
        // ---- a batch: every lane queues its next <= RQ_BATCH runs ----
        if (ln == 0) s_qn[wv] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t take = mine;
#pragma unroll
        for (int t = 0; t < RQ_BATCH; ++t) take &= take - 1u;  // `mine` without its RQ_BATCH lowest set bits ...
        take = mine & ~take;                                    // ... = those bits
        mine &= ~take;
        uint32_t at = atomicAdd(&s_qn[wv], (uint32_t)__popc(take));  // (one address: a wave scan)
        while (__ballot(take != 0)) {
          if (take) {
            const uint32_t i = (uint32_t)__ffs((int)take) - 1u;
            take &= take - 1u;
            wq[at++] = (uint16_t)(ln | (i << 6));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t T = s_qn[wv];
        // ---- lane t cuts runs t and t + 64, then t + 128 and t + 192, ..: two per trip, stage by stage, so that their
        // LDS round trips (descriptor, entry, read words, slab slot) overlap instead of queueing up ----
        for (uint32_t t0 = 0; t0 < T; t0 += 128u) {
          bool on[2], put[2];
          uint32_t dsc[2], en[2], pe[2], sh[2], d[2][4], run_bin[2], slot[2], bh[2];
          const uint32_t* ord[2];
          uint64_t lo[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            on[u] = t0 + (uint32_t)u * 64u + ln < T;
            dsc[u] = on[u] ? wq[t0 + (uint32_t)u * 64u + ln] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            ord[u] = wrd + (size_t)(dsc[u] & 63u) * RP_STRIDE;  // the owner's area
            const uint32_t i = dsc[u] >> 6;
            en[u] = ((const uint8_t*)(ord[u] + 10))[i];
            pe[u] = ((const uint8_t*)(ord[u] + 17))[i];
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t e = (uint32_t)k - 2u + pe[u];  // the run's last k-mer ends at base e
            const uint32_t o = 2u * (159u - e), dw = o >> 5;
            sh[u] = o & 31u;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[u][j] = ord[u][dw + j];
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t back = en[u] >> 4;
            lo[u] = (uint64_t)__builtin_amdgcn_alignbit(d[u][1], d[u][0], sh[u]) |
                    ((uint64_t)__builtin_amdgcn_alignbit(d[u][2], d[u][1], sh[u]) << 32);
            const uint32_t f = (uint32_t)(lo[u] >> (2u * back)) & mmask;
            uint32_t c = f;
            if (CANON) {
              uint32_t y = __brev(~f);
              y = ((y & 0xAAAAAAAAu) >> 1) | ((y & 0x55555555u) << 1);
              c = min(f, y >> (32 - 2 * m));
            }
            bh[u] = msp_binhash(mmer_hash(c));
            run_bin[u] = bh[u] >> (32 - bin_bits);
            put[u] = on[u] && run_bin[u] - bin_lo < bin_hi - bin_lo;
            slot[u] = put[u] ? atomicAdd(&s_fill[run_bin[u] >> sub_bits], 1u) : 0u;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (put[u]) {
              const int n = (int)(en[u] & 15u) + 1;
              const uint32_t back = en[u] >> 4;
              const uint32_t hi = __builtin_amdgcn_alignbit(d[u][3], d[u][2], sh[u]) & 0x3FFFFFu;
              uint64_t w;
              uint32_t x;
              msp_record_make(lo[u], hi, k, n, (uint32_t)(k + n - 1 - m) - back, w, x);
              x |= msp_stamp(bh[u], k);
              const uint32_t coarse = run_bin[u] >> sub_bits;
              if (slot[u] < 2 * SLAB) {
                const uint64_t sb = s_slab[slot[u] >> slab_log2][coarse];
                msp_rec12_store(rec_a, sb + (slot[u] & (SLAB - 1)), w, x);
              } else {  // more than two slabs' worth in one chunk: one reservation per record
                const uint32_t at2 = atomicAdd(&coarse_cur[coarse * P1_CUR_STRIDE], 1u);
                if (at2 < cap_a) msp_rec12_store(rec_a, (uint64_t)coarse * cap_a + at2, w, x);
                else atomicExch(flag, 1u);
              }
              ++n_emit;
              const uint32_t fb = run_bin[u] - bin_lo;
              atomicAdd(&s_fine[fb >> 2], 1u << ((fb & 3u) * 8));
            }
          }
        }
      }
    } else {
    // two runs per turn, stage by stage: their LDS round trips (entry, read words, slab slot) overlap instead of queueing up
    while (__ballot(mine != 0)) {
      bool on[2];
      uint32_t en[2], pe[2], d[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        on[u] = mine != 0;
        const uint32_t i = on[u] ? (uint32_t)__ffs((int)mine) - 1u : 0u;
        mine &= mine - 1u;  // (0 stays 0)
        en[u] = ent8[i];
        pe[u] = end8[i];
      }
      uint32_t sh[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t e = (uint32_t)k - 2u + pe[u];  // the run's last k-mer ends at base e
        // bases e - L + 1 .. e = bit pairs 159 - e .. 159 - e + L - 1 of the staged read, base e least significant
        const uint32_t o = 2u * (159u - e), dw = o >> 5;
        sh[u] = o & 31u;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[u][j] = rd[dw + j];  // (past the read: the entries, never used -- 2L bits are)
      }
      uint64_t lo[2];
      uint32_t run_bin[2], slot[2], bh[2];
      bool put[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t back = en[u] >> 4;
        lo[u] = (uint64_t)__builtin_amdgcn_alignbit(d[u][1], d[u][0], sh[u]) |
                ((uint64_t)__builtin_amdgcn_alignbit(d[u][2], d[u][1], sh[u]) << 32);
        // the bin, from the minimizer m-mer itself: it ends `back` bases before e (back + m <= 31 bases: inside `lo`)
        const uint32_t f = (uint32_t)(lo[u] >> (2u * back)) & mmask;
        uint32_t c = f;
        if (CANON) {
          uint32_t y = __brev(~f);
          y = ((y & 0xAAAAAAAAu) >> 1) | ((y & 0x55555555u) << 1);
          c = min(f, y >> (32 - 2 * m));
        }
        bh[u] = msp_binhash(mmer_hash(c));
        run_bin[u] = bh[u] >> (32 - bin_bits);
        put[u] = on[u] && run_bin[u] - bin_lo < bin_hi - bin_lo;  // (more than two passes: the half says little, the bin everything)
        slot[u] = put[u] ? atomicAdd(&s_fill[run_bin[u] >> sub_bits], 1u) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (put[u]) {
          const int n = (int)(en[u] & 15u) + 1;
          const uint32_t back = en[u] >> 4;
          const uint32_t hi = __builtin_amdgcn_alignbit(d[u][3], d[u][2], sh[u]) & 0x3FFFFFu;
          uint64_t w;
          uint32_t x;
          msp_record_make(lo[u], hi, k, n, (uint32_t)(k + n - 1 - m) - back, w, x);
          x |= msp_stamp(bh[u], k);
          const uint32_t coarse = run_bin[u] >> sub_bits;
          if (slot[u] < 2 * SLAB) {
            const uint64_t sb = s_slab[slot[u] >> slab_log2][coarse];
            msp_rec12_store(rec_a, sb + (slot[u] & (SLAB - 1)), w, x);
          } else {  // more than two slabs' worth in one chunk: one reservation per record
            const uint32_t at = atomicAdd(&coarse_cur[coarse * P1_CUR_STRIDE], 1u);
            if (at < cap_a) msp_rec12_store(rec_a, (uint64_t)coarse * cap_a + at, w, x);
            else atomicExch(flag, 1u);
          }
          ++n_emit;
          const uint32_t fb = run_bin[u] - bin_lo;
          atomicAdd(&s_fine[fb >> 2], 1u << ((fb & 3u) * 8));
        }
      }
    }
    }
  // once per chunk (512 reads put ~90 records into each of a pass's coarse bins at S = 2) the bins whose slab filled up move on
    __syncthreads();
    if (threadIdx.x < P1_BINS) {
      uint32_t f = s_fill[threadIdx.x];
      if (f >= SLAB) {
        if (f >= 2 * SLAB) {
          s_slab[0][threadIdx.x] = slab_at(threadIdx.x, pending);
          s_slab[1][threadIdx.x] = slab_at(threadIdx.x, reserve_raw(threadIdx.x));
          f = 0;
        } else {
          s_slab[0][threadIdx.x] = s_slab[1][threadIdx.x];
          s_slab[1][threadIdx.x] = slab_at(threadIdx.x, pending);
          f -= SLAB;
        }
        pending = reserve_raw(threadIdx.x);
        s_fill[threadIdx.x] = f;
      }
    }
    __syncthreads();
  }
  // the slots nobody took: MSP_EMPTY
  if (threadIdx.x < P1_BINS) s_slab[2][threadIdx.x] = slab_at(threadIdx.x, pending);
  if (threadIdx.x == 0) s_sum = s_emit = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 3u * P1_BINS * SLAB; i += blockDim.x) {
    const uint32_t o = i & (SLAB - 1), bw = i >> slab_log2, b = bw & (P1_BINS - 1), which = bw >> 7;
    const uint64_t sb = s_slab[which][b];
    if (sb != ~0ull && (which == 2 || which * SLAB + o >= s_fill[b])) msp_rec12_store(rec_a, sb + o, MSP_EMPTY, 0u);
  }
  uint32_t sum = 0;
  for (uint32_t b = threadIdx.x; b < P; b += blockDim.x) {
    const uint32_t fb = b - bin_lo;
    const uint32_t v = fb < bin_hi - bin_lo ? (s_fine[fb >> 2] >> ((fb & 3u) * 8)) & 0xFFu : 0u;
    cnt_rows[(uint64_t)blockIdx.x * P + b] = v;
    sum += v;
  }
  atomicAdd(&s_sum, sum);  // a wrapped 8-bit counter (carry into the neighbour or out of the word) leaves the sum short
  atomicAdd(&s_emit, n_emit);
  __syncthreads();
  if (threadIdx.x == 0 && s_sum != s_emit) atomicExch(flag, 1u);
}

__device__ __forceinline__ uint32_t split_hash(uint64_t key) {
  return (uint32_t)((key * 0xD6E8FEB86659FD93ull) >> 32);
}

// Distinct keys a counting pass can end with WITHOUT being void: a thread looks at the count before every insert, so up to
// one key per thread can follow the limit FILL - 1 (see insert_fwd) -- and with lower = 1 every one of them is a survivor.
// Threads of the half-size leaf workgroup (GEO 1).  Round 5: 768, not 512 -- the 79 KB of LDS tables allow two workgroups
// per CU whatever their size, so 2 x 12 waves (six per SIMD: 76 VGPRs fit) hide more of each other's LDS round trips than
// 2 x 8, and a bin's ~1630 records are three per lane instead of four (phase A is a chain of one probe loop per record).
// 1 Gb slice, ms per sample: 512 / 640 / 768 / 896 / 1024 threads = 87 / 128 / 78 / 115 / 80 (640 and 896 split their
// waves unevenly over the four SIMDs; 1024 needs 64 VGPRs and spills).
#ifndef RFX_LEAF_BLK
#define RFX_LEAF_BLK 768
#endif
constexpr uint32_t MSP_LEAF_BLK(int geo) { return geo ? (uint32_t)RFX_LEAF_BLK : 1024u; }
#ifndef RFX_LEAF_WPE
#define RFX_LEAF_WPE (2 * RFX_LEAF_BLK / 256)
#endif
constexpr int MSP_LEAF_WPE(int geo) { return geo ? RFX_LEAF_WPE : 4; }  // waves per SIMD the registers must leave room for
constexpr uint32_t MSP_LEAF_PASS_MAX(int geo) { return (geo ? 4096u : 8192u) * 3 / 4 + MSP_LEAF_BLK(geo); }
// Survivors leave the leaf through STAGING CHUNKS in global memory (round 6): a workgroup owns one chunk at a time, phase C
// appends (key, count) pairs to it, and when the chunk could not take another pass the workgroup writes down how full it
// is and takes the next free chunk of the launch's pool (one global atomic per ~30 bins, by thread 0, nobody waits for it
// but its own wave).  k_surv_place, launched behind the leaf, turns every chunk into sorted runs of the 128 coarse pos
// bins.  Until round 5 the workgroup did that itself ("flush", every ~10 bins): T * key from a borrowed LDS table, a
// count, a two-wave scan, a reservation between two barriers, a sort through the k-mer table's arrays -- 8 % of the
// kernel's time, and the reason the compiler wanted 84 VGPRs where two 768-thread workgroups per CU leave 80: a tenth of
// the issued instructions were v_readlane / v_writelane of spilled SGPRs (the 30 kernel arguments, half of them the flush's).
constexpr uint32_t MSP_LEAF_CHUNK(int geo, bool big) { return big ? (geo ? 16384u : 32768u) : MSP_LEAF_PASS_MAX(geo) + 512u; }
#ifndef MSP_ILP_OVERRIDE
#define MSP_ILP_OVERRIDE 3
#endif
constexpr int MSP_ILP = MSP_ILP_OVERRIDE;  // records a lane loads before its first probe (a bin holds ~2 per lane of 768, 1.6 of 1024)
#ifndef RFX_RC_PROBES
#define RFX_RC_PROBES 16
#endif
#ifndef RFX_RC_SHRINK
#define RFX_RC_SHRINK 2
#endif
constexpr int MSP_RC_PROBES = RFX_RC_PROBES;  // a record that finds no cache slot within this many probes bypasses the cache
constexpr unsigned long long MSP_RK_MUL = 0x9E3779B97F4A7C15ull;  // record cache key = word ^ plane * this (odd: injective in the plane)
constexpr uint32_t MSP_RC_UNLISTED = 0x80000000u;  // record cache count: the record found no room in the k-mer map

// One workgroup per fine minimizer bin (round 4: rebuilt around three barriers per bin).
//
//   A  every record of the bin goes into a small cache keyed by the record (word + plane): reads that cover the same
//      stretch of genome cut it into the same super-k-mers, so at sequencing depth most records are copies and cost one
//      CAS + three adds nobody waits for.  The lane that takes a fresh slot also books the record's n k-mers in a dense
//      map, two to an entry (s_kmap[..] = slot << 3 | pair; one LDS add reserves the entries).
//   B  two k-mers per lane over the map: cut k-mer q out of the cached run, canonical form, one returning CAS into the
//      k-mer table, count += multiplicity of the record.  Every instance of a k-mer has the same minimizer, so the
//      counts are final.
//   C  the table is scanned once: survivors (lower <= count <= upper) go as (key, count) to a staging chunk of the
//      workgroup in global memory, every slot and the cache are cleared on the way (no separate clearing pass).
//   F  only when the staging chunk (CH entries, > the table's fill limit) could not take another bin -- every ~20 bins
//      on 30x data with the 8192-entry chunk of a big input: w = T * key, and the chunk
//      is scattered into the 128 coarse pos bins -- until round 3 every bin paid for this (four barriers, seven table
//      reads per survivor and the round trip of 128 global atomics: 25 % of the kernel).
//
// A bin (or part of it) whose distinct k-mers overflow the table is split in two by a hash bit and each half retried.
// GEO 0: one 1024-thread workgroup per CU (8192-slot table); GEO 1: half of everything, two workgroups per CU.
// (waves_per_eu: two 768-thread workgroups per CU are six waves per SIMD = 80 registers; left to itself the compiler takes 84
// since the flush sorts its chunk, and only ONE workgroup fits)
template <bool CANON, int GEO>
__global__ __launch_bounds__(MSP_LEAF_BLK(GEO)) __attribute__((amdgpu_waves_per_eu(MSP_LEAF_WPE(GEO), MSP_LEAF_WPE(GEO)))) void k_msp_leaf(
    const uint64_t* const* __restrict__ seg_inst, const uint64_t* const* __restrict__ seg_bs, int nseg,
    const uint64_t* __restrict__ inst0, const uint64_t* __restrict__ bs0, const uint32_t* const* __restrict__ seg_ext,
    const uint32_t* __restrict__ ext0, uint32_t P, int k, uint64_t lower, uint64_t upper,
    uint64_t* __restrict__ stage_k, uint32_t* __restrict__ stage_c, uint32_t CH, uint32_t* __restrict__ stage_fill,
    uint32_t* __restrict__ stage_more, uint32_t n_chunks, unsigned int* __restrict__ flag, unsigned int* __restrict__ err,
    unsigned int* __restrict__ stage_short, int force_mixed) {
  constexpr int TBL_LOG2 = GEO ? 12 : 13, TBL = 1 << TBL_LOG2, BLK = (int)MSP_LEAF_BLK(GEO), FILL = TBL * 3 / 4;
  static_assert(MSP_LEAF_PASS_MAX(GEO) >= (uint32_t)(FILL + BLK) && FILL + BLK <= TBL, "what a pass can leave behind fits the chunk and the table");
  constexpr int RC_LOG2 = TBL_LOG2 - (GEO ? RFX_RC_SHRINK : 2), RC = 1 << RC_LOG2, KMAP = TBL;
  static_assert(RC <= 8192, "a k-mer map entry is slot << 3 | pair in 16 bits");
  __shared__ __attribute__((aligned(16))) unsigned long long s_keys[TBL];
  __shared__ uint32_t s_cnt[TBL];
  // record cache.  A record is 96 bits, an LDS compare-and-swap takes 64: the slot's key is word ^ plane * odd -- two
  // records with the same word and different planes (an error in the last bases of a long run: a few per bin) get
  // different keys, and a probe stays ONE returning CAS.  Every record that lands on a slot adds 1 and folds its plane
  // into the slot's minimum and maximum, three adds nobody waits for: min == max proves that all of them were the
  // same record (same key + same plane = same word).  Two different records under one key -- a 64-bit coincidence,
  // not seen yet -- void the cached pass of the bin, which is then counted without the cache.
  __shared__ unsigned long long s_rk[RC];
  __shared__ uint32_t s_rc[RC], s_rmin[RC], s_rmax[RC];
  __shared__ uint32_t s_mixed[2];
  __shared__ __attribute__((aligned(16))) uint16_t s_kmap[KMAP];
  __shared__ uint32_t s_chunk;  // the staging chunk the workgroup appends to (written by thread 0 between two passes)
  // by parity of the pass: distinct keys, overflow, k-mer map fill, survivors of the scan (zeroed for the NEXT pass by
  // thread 0 after the first barrier of a pass: nobody reads the other parity's between that barrier and the next pass)
  __shared__ uint32_t s_nd[2], s_ovf[2], s_nk[2], s_ns[2];

  uint64_t pre[MSP_ILP];
  uint32_t prex[MSP_ILP];
  uint64_t pre_a = 0, pre_e = 0;
  // The extents of a bin are fetched one bin earlier than its records (round 5): prefetch() used to load bs0[b] and wait
  // for it before it could issue the record loads -- an exposed round trip to memory per bin (~6 % of the kernel; that
  // the leaf feels such waits showed when a mask applied to the prefetched planes at load time cost 7 %).
  uint64_t nx_a = blockIdx.x < P ? bs0[blockIdx.x] : 0, nx_e = blockIdx.x < P ? bs0[blockIdx.x + 1] : 0;
  auto prefetch = [&](uint32_t b) {
    if (b >= P) return;
    pre_a = nx_a;
    pre_e = nx_e;
    if (b + gridDim.x < P) {  // (used by the next prefetch: nobody waits for these here)
      nx_a = bs0[b + gridDim.x];
      nx_e = bs0[b + gridDim.x + 1];
    }
#pragma unroll
    for (int u = 0; u < MSP_ILP; ++u) {
      const uint64_t i = pre_a + threadIdx.x + (uint64_t)u * BLK;
      pre[u] = i < pre_e ? inst0[i] : MSP_EMPTY;
      prex[u] = i < pre_e ? ext0[i] : 0u;  // (the stamp is masked off where the plane is used: here it would wait for the load)
    }
  };
  prefetch(blockIdx.x);
  for (int i = threadIdx.x; i < TBL; i += BLK) {
    s_keys[i] = RFX_EMPTY;
    s_cnt[i] = 0;
  }
  for (int i = threadIdx.x; i < RC; i += BLK) {
    s_rk[i] = MSP_EMPTY;
    s_rc[i] = 0;
    s_rmin[i] = ~0u;
    s_rmax[i] = 0;
  }
  if (threadIdx.x == 0) s_chunk = blockIdx.x;  // (the launch's pool: gridDim.x chunks handed out here, the others by stage_more)
  if (threadIdx.x < 2) s_nd[threadIdx.x] = s_ovf[threadIdx.x] = s_nk[threadIdx.x] = s_ns[threadIdx.x] = s_mixed[threadIdx.x] = 0;
  uint32_t used = 0;  // entries of the staging chunk in use (the same in every thread)
  uint32_t X = 0;     // parity of the pass
  __syncthreads();

  for (uint32_t bin = blockIdx.x; bin < P; bin += gridDim.x) {
    const uint64_t a0 = pre_a, e0 = pre_e;
    bool prefetched_next = false;
    // sub-range (r, j): k-mers whose top r bits of split_hash equal j; depth-first over the halves
    int r = 0;
    uint32_t j = 0;
    bool failed = false, nocache = false;
    TM_DECL;
    for (;;) {
      TM(15);
      // forward k-mer `fwd` into the table, counted `mult` times
      auto insert_fwd = [&](uint64_t fwd, uint32_t mult) {
        uint64_t key = fwd;
        if (CANON) {
          const uint64_t rc = revcomp_bases(fwd, k);
          key = rc < fwd ? rc : fwd;
        }
        if (r > 0 && (split_hash(key) >> (32 - r)) != j) return;
        // The probe is ONE returning CAS: it yields "was empty, now mine", "already mine" or "someone
        // else's" without a separate read-and-branch for the new-key case.  New keys are only counted
        // (no returned ticket); the count is looked at before every insert, so at most one key per thread can
        // follow FILL, which the TBL - FILL spare slots absorb -- probing always terminates.  A skipped insert voids
        // the pass.
        if (__hip_atomic_load(&s_nd[X], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (uint32_t)FILL) {
          s_ovf[X] = 1;
          return;
        }
        // double hashing (an odd step from other bits of the hash): a wave probes as long as its unluckiest lane, and
        // linear probing's clusters give that lane long walks at the 40-50 % load of a full bin
        const uint32_t hk = leaf_hash32(key);
        uint32_t slot = hk >> (32 - TBL_LOG2);
        const uint32_t step = ((hk >> 3) | 1u) & (TBL - 1);
        // (the loop only finds the slot -- it runs as long as the wave's unluckiest lane probes; what follows a hit
        // comes after it, once)
        bool fresh = false;
        for (;;) {
          const unsigned long long old = atomicCAS(&s_keys[slot], (unsigned long long)RFX_EMPTY, (unsigned long long)key);
          fresh = old == RFX_EMPTY;
          if (fresh || old == key) break;
          slot = (slot + step) & (TBL - 1);
        }
        atomicAdd(&s_cnt[slot], mult);
        if (fresh) atomicAdd(&s_nd[X], 1u);
      };
      auto insert_record = [&](uint64_t x, uint32_t xe, uint32_t mult) {  // all k-mers of a record, one after the other
        uint64_t lo, hi;
        msp_record_run(x, xe, k, lo, hi);
        const int n = msp_record_n(x);
        for (int q = 0; q < n; ++q) insert_fwd(msp_run_kmer(lo, hi, k, n, q), mult);
      };
      // ---- A: records -> cache (+ k-mer map) ----
      for (int sg = 0; sg < nseg; ++sg) {
        const uint64_t a = sg == 0 ? a0 : seg_bs[sg][bin], e = sg == 0 ? e0 : seg_bs[sg][bin + 1];
        const uint64_t* __restrict__ src = sg == 0 ? inst0 : seg_inst[sg];
        const uint32_t* __restrict__ srcx = sg == 0 ? ext0 : seg_ext[sg];
        for (uint64_t base = a; base < e; base += (uint64_t)MSP_ILP * BLK) {
          uint64_t rec[MSP_ILP];
          uint32_t recx[MSP_ILP];
          if (sg == 0 && base == a && !prefetched_next) {
#pragma unroll
            for (int u = 0; u < MSP_ILP; ++u) {
              rec[u] = pre[u];
              recx[u] = prex[u];
            }
          } else {
#pragma unroll
            for (int u = 0; u < MSP_ILP; ++u) {
              const uint64_t i = base + threadIdx.x + (uint64_t)u * BLK;
              rec[u] = i < e ? src[i] : MSP_EMPTY;
              recx[u] = i < e ? srcx[i] : 0u;
            }
          }
#pragma unroll
          for (int u = 0; u < MSP_ILP; ++u) {
            const uint64_t x = rec[u];
            const uint32_t xe = msp_plane_bits(recx[u], k);
            if (x == MSP_EMPTY) continue;
            uint32_t h = (uint32_t)x ^ (uint32_t)(x >> 23) ^ (uint32_t)(x >> 41) ^ (xe * 0x85EBCA6Bu);
            h *= 0x9E3779B1u;
            const uint32_t hstep = ((h >> 5) | 1u) & (RC - 1);
            h >>= 32 - RC_LOG2;
            // The probe loop only FINDS the slot -- it runs as long as the slowest lane of the wave probes, so nothing
            // else lives in it (the booking of a new record's k-mers, inside it, made phase A 2.4 x slower).
            int state = 0;  // 1: slot taken (h), 2: found, 0: count the record directly
            const unsigned long long ck = x ^ ((unsigned long long)xe * MSP_RK_MUL);
            if (!nocache && ck != MSP_EMPTY) {
              for (int p = 0; p < MSP_RC_PROBES; ++p) {
                const unsigned long long old = atomicCAS(&s_rk[h], (unsigned long long)MSP_EMPTY, ck);
                if (old == MSP_EMPTY || old == ck) {
                  state = old == ck ? 2 : 1;
                  break;
                }
                h = (h + hstep) & (RC - 1);
              }
            }
            if (state) {
              atomicAdd(&s_rc[h], 1u);
              atomicMin(&s_rmin[h], xe);
              atomicMax(&s_rmax[h], xe);
            }
            TMC(22, 1);
            TMC(23, state == 0);
            if (state == 1) {  // book the record's k-mers in the map, two to an entry
              const uint32_t np = ((uint32_t)msp_record_n(x) + 1u) >> 1;
              const uint32_t kb = atomicAdd(&s_nk[X], np);
              if (kb + np <= (uint32_t)KMAP) {
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q)  // (no loop: the wave would run it as often as its longest record asks)
                  if (q < np) s_kmap[kb + q] = (uint16_t)((h << 3) | q);
              } else {  // no room in the map: B looks for these in the cache itself (what is left of the map: no entry)
                for (uint32_t q = kb; q < (uint32_t)KMAP; ++q) s_kmap[q] = 0xFFFFu;
                atomicOr(&s_rc[h], MSP_RC_UNLISTED);
              }
            }
            if (state == 0) insert_record(x, xe, 1u);
          }
        }
      }
      if (!prefetched_next) {  // the next bin's loads fly while this one is counted
        prefetch(bin + gridDim.x);
        prefetched_next = true;
      }
      TM(16);  // (-DRFX_TIMING: what follows up to the next probe is the wait at the barrier)
      __syncthreads();
      TM(9);
      // ---- B: two k-mers per lane ----
      if (threadIdx.x == 0) s_nd[X ^ 1] = s_ovf[X ^ 1] = s_nk[X ^ 1] = s_ns[X ^ 1] = s_mixed[X ^ 1] = 0;
      const uint32_t nk_all = s_nk[X], nk = min(nk_all, (uint32_t)KMAP);
      for (uint32_t i = threadIdx.x; i < nk; i += BLK) {
        const uint32_t e = s_kmap[i];
        if (e == 0xFFFFu) continue;
        const uint32_t slot = e >> 3, q = (e & 7u) << 1;
        const uint32_t xe = s_rmax[slot];
        const uint64_t x = s_rk[slot] ^ ((uint64_t)xe * MSP_RK_MUL);
        if (q == 0 && s_rmin[slot] != xe) s_mixed[X] = 1;  // two different records met in this slot
        uint64_t lo, hi;
        msp_record_run(x, xe, k, lo, hi);
        const int n = msp_record_n(x);
        const uint32_t mult = s_rc[slot];
        insert_fwd(msp_run_kmer(lo, hi, k, n, (int)q), mult);
        if ((int)q + 1 < n) insert_fwd(msp_run_kmer(lo, hi, k, n, (int)q + 1), mult);
      }
      if (nk_all > (uint32_t)KMAP) {  // (rare) records that found no room in the map
        for (int i = threadIdx.x; i < RC; i += BLK) {
          const uint32_t c = s_rc[i];
          if (c & MSP_RC_UNLISTED) {
            const uint32_t xe = s_rmax[i];
            if (s_rmin[i] != xe) s_mixed[X] = 1;
            insert_record(s_rk[i] ^ ((uint64_t)xe * MSP_RK_MUL), xe, c & ~MSP_RC_UNLISTED);
          }
        }
      }
      TM(17);  // (-DRFX_TIMING: what follows up to the next probe is the wait at the barrier)
      __syncthreads();
      TM(11);
      // (the pass is void: once more, without the cache.  force_mixed -- RFX_LEAF_FORCE_MIXED, a test knob: the recount
      // route is taken by every third bin, a 64-bit coincidence of two records being too rare to wait for)
      const bool mixed = s_mixed[X] != 0 || (force_mixed && !nocache && bin % 3u == 1u);
      const bool ovf = s_ovf[X] != 0 || mixed;  // from here on: "nothing of this pass leaves"
      const bool split = s_ovf[X] != 0 && !mixed;
      TMC(20, split);
      TMC(24, mixed);
      TMC(21, 1);
      // ---- C: survivors out, everything cleared ----
      uint64_t* const stk = stage_k + (size_t)s_chunk * CH;
      uint32_t* const stc = stage_c + (size_t)s_chunk * CH;
      for (int i = threadIdx.x; i < TBL; i += BLK) {
        const uint64_t key = s_keys[i];
        if (key == RFX_EMPTY) continue;
        const uint32_t c = s_cnt[i];
        s_keys[i] = RFX_EMPTY;
        s_cnt[i] = 0;
        if (!ovf && c >= lower && c <= upper) {
          const uint32_t o = used + atomicAdd(&s_ns[X], 1u);
          stk[o] = key;
          stc[o] = c;
        }
      }
      for (int i = threadIdx.x; i < RC; i += BLK) {
        s_rk[i] = MSP_EMPTY;
        s_rc[i] = 0;
        s_rmin[i] = ~0u;
        s_rmax[i] = 0;
      }
      TM(18);  // (-DRFX_TIMING: what follows up to the next probe is the wait at the barrier)
      __syncthreads();
      TM(12);
      used += s_ns[X];
      X ^= 1u;
      // The chunk must be able to take whatever the next pass can leave: FILL - 1 + BLK keys, not FILL.  (Until late in
      // round 4 the test was `used > CH - FILL`: a pass that ended between FILL and FILL + BLK distinct keys, all of them
      // survivors, could write past the workgroup's chunk into its neighbour's -- a handful of garbage keys per 10^8
      // records with -L 1 on a sparse sample, found by tests/test_scale_gpu.py::test_wgs_slice_properties.)
      if (used > CH && threadIdx.x == 0) atomicExch(err, 1u);  // (cannot happen: fail loudly rather than corrupt)
      if (used > CH - MSP_LEAF_PASS_MAX(GEO)) {  // the chunk could not take another pass: on to the next free one
        // (nobody reads s_chunk between this barrier and phase C of the next pass, two barriers on)
        if (threadIdx.x == 0) {
          const uint32_t mine = s_chunk;
          stage_fill[mine] = used;
          uint32_t nx = gridDim.x + atomicAdd(stage_more, 1u);
          if (nx >= n_chunks) {  // the pool is spent: the host runs the launch again with the pool the counters ask for;
            atomicExch(flag, 1u);  // until then this workgroup writes over its own chunk (results void)
            atomicMax(stage_short, nx - n_chunks + 1u);
            nx = mine;
          }
          s_chunk = nx;
        }
        used = 0;
        TM(13);
      }
      if (mixed) {  // the same sub-range again
        nocache = true;
        continue;
      }
      if (split) {  // split this sub-range
        if (r >= LEAF_RMAX) {
          failed = true;
          break;
        }
        ++r;
        j <<= 1;
      } else {  // next sub-range in depth-first order
        while (r > 0 && (j & 1u)) {
          j >>= 1;
          --r;
        }
        if (r == 0) break;
        ++j;
      }
    }
    if (failed && threadIdx.x == 0) atomicExch(err, 1u);
  }
  if (threadIdx.x == 0) stage_fill[s_chunk] = used;
}

// The staging chunks of a leaf launch -> the 128 coarse pos bins (round 6; until then the leaf's own "flush").  A chunk is
// stage_fill[ch] <= CH (key, count) pairs as the leaf's phase C left them.  Rounds of SP_N entries: w = T * key (the 8 x 256
// table of T in LDS), count per coarse bin, ONE reservation per bin and round (128 global atomics whose round trip the
// other workgroup of the CU covers), the round placed in bin order in LDS and written out by consecutive lanes to
// consecutive addresses: a bin's ~32 entries of a round are 256 contiguous bytes.  Entries outside [pos_lo, pos_hi)
// (a table restricted to a range of positions) are dropped; a bin without room raises `flag`, its cursor keeps counting
// and the host reruns the emit with the capacity the cursors ask for.  Every chunk's fill is put back to 0 on the way:
// the next launch finds the pool as this one found it.
#ifndef RFX_SP_BLK
#define RFX_SP_BLK 512
#endif
#ifndef RFX_SP_NE
#define RFX_SP_NE 8
#endif
constexpr int SP_BLK = RFX_SP_BLK, SP_NE = RFX_SP_NE, SP_N = SP_BLK * SP_NE;  // 512 x 8: 64 KB of LDS, two workgroups per CU
__global__ __launch_bounds__(SP_BLK) void k_surv_place(const uint64_t* __restrict__ stage_k, const uint32_t* __restrict__ stage_c,
                                                       uint32_t CH, uint32_t* __restrict__ stage_fill,
                                                       uint32_t* __restrict__ stage_more, uint32_t n_first, uint32_t n_chunks,
                                                       const uint64_t* __restrict__ g_lut, int ntab, int sel_bits, int shift1,
                                                       uint64_t pos_lo, uint64_t pos_hi, uint64_t* __restrict__ out_w,
                                                       uint32_t* __restrict__ out_c, uint32_t* __restrict__ cur, uint32_t cap,
                                                       unsigned int* __restrict__ flag) {
  __shared__ uint64_t s_lut[8 * 256];
  __shared__ uint64_t s_w[SP_N];
  __shared__ uint32_t s_c[SP_N];
  __shared__ uint32_t s_pc[P1_BINS], s_pst[P1_BINS], s_nfl;
  __shared__ uint64_t s_pbase[P1_BINS];
  static_assert(P1_BINS == 128, "the scan below is two waves wide");
  for (int i = threadIdx.x; i < ntab * 256; i += SP_BLK) s_lut[i] = g_lut[i];
  if (threadIdx.x < P1_BINS) s_pc[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t n_used = min(n_first + *stage_more, n_chunks);
  // The rounds of this workgroup's chunks, one after the other; a round's entries are asked for a round ahead (and a
  // chunk's fill a chunk ahead): with the loads at the top of the round a workgroup spent most of its time waiting for
  // them, then for the 128 reservations, with one other workgroup on the CU to cover for it (22 ms per W sample).
  uint32_t ch = blockIdx.x, base = 0;
  uint32_t used = ch < n_used ? min(stage_fill[ch], CH) : 0u;
  uint32_t ch_n = ch + gridDim.x;
  uint32_t used_n = ch_n < n_used ? min(stage_fill[ch_n], CH) : 0u;
  auto settle = [&]() {  // (ch, base) -> the next round that has entries, or ch >= n_used
    while (ch < n_used && base >= used) {
      if (threadIdx.x == 0 && used) stage_fill[ch] = 0;
      ch = ch_n;
      used = used_n;
      base = 0;
      ch_n += gridDim.x;
      used_n = ch_n < n_used ? min(stage_fill[ch_n], CH) : 0u;
    }
  };
  uint64_t nk[SP_NE];
  uint32_t nc[SP_NE];
  auto fetch = [&]() {
    const bool any = ch < n_used;
    const uint32_t n_in = any ? min((uint32_t)SP_N, used - base) : 0u;
    const uint64_t* const stk = stage_k + (size_t)(any ? ch : 0u) * CH + base;
    const uint32_t* const stc = stage_c + (size_t)(any ? ch : 0u) * CH + base;
#pragma unroll
    for (int u = 0; u < SP_NE; ++u) {
      const uint32_t i = threadIdx.x + (uint32_t)u * SP_BLK;
      nk[u] = i < n_in ? stk[i] : RFX_EMPTY;
      nc[u] = i < n_in ? stc[i] : 0u;
    }
  };
  settle();
  fetch();
  while (ch < n_used) {
    uint64_t kv[SP_NE];
    uint32_t cv[SP_NE];
#pragma unroll
    for (int u = 0; u < SP_NE; ++u) {
      kv[u] = nk[u];
      cv[u] = nc[u];
    }
    base += (uint32_t)SP_N;
    settle();
    fetch();  // (the next round's: nobody waits for them before the next trip)
#pragma unroll
    for (int u = 0; u < SP_NE; ++u) {
      if (kv[u] == RFX_EMPTY) continue;
      uint64_t w = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < ntab) w ^= s_lut[t * 256 + (uint32_t)((kv[u] >> (8 * t)) & 255u)];
      const uint64_t pos = w >> sel_bits;
      if (pos >= pos_lo && pos < pos_hi) atomicAdd(&s_pc[(uint32_t)(w >> shift1)], 1u);
      else w = RFX_EMPTY;
      kv[u] = w;
    }
    __syncthreads();
    if (threadIdx.x < P1_BINS) {  // (two waves) room in the coarse bins; where a bin starts in the sorted round
      const uint32_t cn = s_pc[threadIdx.x];
      const uint32_t at = cn ? atomicAdd(&cur[threadIdx.x * P1_CUR_STRIDE], cn) : 0u;
      uint32_t incl = cn;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63u) >= d) incl += up;
      }
      uint32_t low = s_pc[threadIdx.x & 63u];  // the first wave's total, for the second
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) low += __shfl_xor(low, d, 64);
      const uint32_t start = incl - cn + (threadIdx.x >= 64 ? low : 0u);
      s_pst[threadIdx.x] = start;
      if ((uint64_t)at + cn > cap) {  // dropped; the host reruns with the capacity the cursors ask for
        // (a plain store: once a store is full every round of every workgroup comes here for most of the 128 bins, and
        // returning exchanges on the ONE flag word made the last launches of such a finish take 30 ms instead of 0.5)
        __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_pbase[threadIdx.x] = ~0ull;
      } else {
        s_pbase[threadIdx.x] = (uint64_t)threadIdx.x * cap + at - start;  // (+ the entry's place in the round)
      }
      if (threadIdx.x == P1_BINS - 1) s_nfl = start + cn;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SP_NE; ++u) {
      const uint64_t w = kv[u];
      if (w == RFX_EMPTY) continue;
      const uint32_t at = atomicAdd(&s_pst[(uint32_t)(w >> shift1)], 1u);
      s_w[at] = w;
      s_c[at] = cv[u];
    }
    __syncthreads();
    const uint32_t n_out = s_nfl;
    for (uint32_t i = threadIdx.x; i < n_out; i += SP_BLK) {
      const uint64_t w = s_w[i];
      const uint64_t pb = s_pbase[(uint32_t)(w >> shift1)];
      if (pb != ~0ull) {
        out_w[pb + i] = w;
        out_c[pb + i] = s_c[i];
      }
    }
    if (threadIdx.x < P1_BINS) s_pc[threadIdx.x] = 0;
    __syncthreads();
  }
}

// Fine bin sizes of the entries in fixed-capacity coarse bins: fine_tot[cb * P2 + sub] += ...  (64-bit
// counters, scanned later).  MODE 0: survivors, sub-bin = bits of the word; MODE 1 / 2: super-k-mer records
// (canonical / not), sub-bin = bits of the record's minimizer bin hash.
template <int MODE>
__global__ __launch_bounds__(L2_BLOCK) void k_surv_hist(const uint64_t* __restrict__ buf_a,
                                                         const uint32_t* __restrict__ coarse_cur, uint32_t cap_a,
                                                         uint32_t P2, int shift2, uint32_t W, int k,
                                                         unsigned long long* __restrict__ fine_tot) {
  __shared__ uint32_t s_cnt[256];
  const uint32_t cb = blockIdx.x / W, jj = blockIdx.x - cb * W;
  const uint64_t a = (uint64_t)cb * cap_a, e = a + min(coarse_cur[cb * P1_CUR_STRIDE], cap_a);
  if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t i = a + (uint64_t)jj * L2_BLOCK + threadIdx.x; i < e; i += (uint64_t)W * L2_BLOCK) {
    const uint64_t w = MODE == 0 ? buf_a[i] : msp_rec12_word((const msp_rec12*)buf_a, i);  // (records: k_msp_part1's 12-byte slots)
    if (MODE != 0 && w == MSP_EMPTY) continue;  // a slot of a k_msp_part1 slab that nobody took
    const uint32_t sub = MODE == 0 ? (uint32_t)(w >> shift2) & (P2 - 1)
                                   : ((MODE == 1 ? msp_record_binhash<true>(w, k) : msp_record_binhash<false>(w, k)) >> shift2) & (P2 - 1);
    atomicAdd(&s_cnt[sub], 1u);
  }
  __syncthreads();
  if (threadIdx.x < P2 && s_cnt[threadIdx.x])
    atomicAdd(&fine_tot[(uint64_t)cb * P2 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

__global__ void k_flag_if_gt(const uint64_t* __restrict__ v, uint64_t limit, unsigned int* __restrict__ flag) {
  if (*v > limit) atomicExch(flag, 1u);
}

// Count-of-counts (jf/sub_commands/histo_main.cc:40-49: bin = min(count, 10001)) of the survivors as the
// leaf left them: the filled part of each fixed-capacity coarse bin.
__global__ __launch_bounds__(256) void k_histo_bins(const uint32_t* __restrict__ counts,
                                                     const uint32_t* __restrict__ coarse_cur, uint32_t cap, uint32_t W,
                                                     unsigned long long* __restrict__ g_histo) {
  __shared__ uint32_t s_h[RFX_HISTO_BINS];
  for (int i = threadIdx.x; i < RFX_HISTO_BINS; i += blockDim.x) s_h[i] = 0;
  __syncthreads();
  const uint32_t cb = blockIdx.x / W, jj = blockIdx.x - cb * W;
  const uint64_t a = (uint64_t)cb * cap, e = a + min(coarse_cur[cb * P1_CUR_STRIDE], cap);
  // four loads in flight per lane (one load per trip left the kernel at 0.4 TB/s: 2048 waves x 256 B against ~2 us)
  const uint64_t step = (uint64_t)W * blockDim.x;
  uint64_t i = a + (uint64_t)jj * blockDim.x + threadIdx.x;
  for (; i + 3 * step < e; i += 4 * step) {
    uint32_t c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = counts[i + u * step];
#pragma unroll
    for (int u = 0; u < 4; ++u) atomicAdd(&s_h[c[u] > 10001u ? 10001u : c[u]], 1u);
  }
  for (; i < e; i += step) {
    const uint32_t c = counts[i];
    atomicAdd(&s_h[c > 10001u ? 10001u : c], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < RFX_HISTO_BINS; b += blockDim.x)
    if (s_h[b]) atomicAdd(&g_histo[b], (unsigned long long)s_h[b]);
}

constexpr int SS_BLOCK = 512;
constexpr int SS_CAP = 2048;

// Final step: one fine pos bin of survivors (distinct words, all kept) -> its slice of the output
// records in w order.  Bucket on the 8 bits below the bin prefix, rank inside the bucket.
__global__ __launch_bounds__(SS_BLOCK) void k_surv_sort(const uint64_t* __restrict__ bw,
                                                         const uint32_t* __restrict__ bc,
                                                         const uint64_t* __restrict__ bs, uint32_t P, int bin_shift,
                                                         const uint64_t* __restrict__ g_lut_inv, int ntab, int sel_bits,
                                                         uint64_t* __restrict__ out_keys,
                                                         uint32_t* __restrict__ out_counts,
                                                         uint64_t* __restrict__ out_pos) {
  __shared__ uint64_t s_lut[8 * 256];
  // a bin's entries wait in registers (SS_CAP / SS_BLOCK = 4 per lane) between the bucket count and the bucket scatter:
  // 42 KB of LDS instead of 66 -- three workgroups per CU instead of two -- and no staging write + read
  __shared__ uint64_t s_w2[SS_CAP];
  __shared__ uint32_t s_c2[SS_CAP];
  __shared__ uint32_t s_bstart[LEAF_BUCKETS + 1], s_bfill[LEAF_BUCKETS];
  for (int i = threadIdx.x; i < ntab * 256; i += blockDim.x) s_lut[i] = g_lut_inv[i];
  const int bsh = bin_shift > 8 ? bin_shift - 8 : 0;
  for (uint32_t bin = blockIdx.x; bin < P; bin += gridDim.x) {
    const uint64_t a = bs[bin];
    const uint64_t n = bs[bin + 1] - a;
    __syncthreads();
    if (n == 0) continue;
    if (n > (uint64_t)SS_CAP) {  // oversize bin (only adversarial input gets here): rank against the whole bin
      for (uint64_t i = threadIdx.x; i < n; i += SS_BLOCK) {
        const uint64_t wi = bw[a + i];
        uint64_t rank = 0;
        for (uint64_t q = 0; q < n; ++q) rank += bw[a + q] < wi;
        out_keys[a + rank] = gf2_mul(s_lut, wi, ntab);
        out_counts[a + rank] = bc[a + i];
        out_pos[a + rank] = wi >> sel_bits;
      }
      continue;
    }
    if (threadIdx.x < LEAF_BUCKETS) s_bfill[threadIdx.x] = 0;
    __syncthreads();
    constexpr int PER = SS_CAP / SS_BLOCK;
    uint64_t rw[PER];
    uint32_t rc[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const uint32_t i = threadIdx.x + (uint32_t)u * SS_BLOCK;
      rw[u] = i < (uint32_t)n ? bw[a + i] : 0;
      rc[u] = i < (uint32_t)n ? bc[a + i] : 0;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (threadIdx.x + (uint32_t)u * SS_BLOCK < (uint32_t)n)
        atomicAdd(&s_bfill[(uint32_t)(rw[u] >> bsh) & (LEAF_BUCKETS - 1)], 1u);
    __syncthreads();
    if (threadIdx.x < 64) wave_scan256(s_bfill, s_bstart, LEAF_BUCKETS);
    __syncthreads();
    if (threadIdx.x < LEAF_BUCKETS) s_bfill[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (threadIdx.x + (uint32_t)u * SS_BLOCK < (uint32_t)n) {
        const uint32_t bk = (uint32_t)(rw[u] >> bsh) & (LEAF_BUCKETS - 1);
        const uint32_t p = s_bstart[bk] + atomicAdd(&s_bfill[bk], 1u);
        s_w2[p] = rw[u];
        s_c2[p] = rc[u];
      }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)n; i += SS_BLOCK) {
      const uint64_t wi = s_w2[i];
      const uint32_t bk = (uint32_t)(wi >> bsh) & (LEAF_BUCKETS - 1);
      const uint32_t b0 = s_bstart[bk], b1 = s_bstart[bk + 1];
      uint32_t rank = b0;
      for (uint32_t q = b0; q < b1; ++q) rank += s_w2[q] < wi;
      out_keys[a + rank] = gf2_mul(s_lut, wi, ntab);
      out_counts[a + rank] = s_c2[i];
      out_pos[a + rank] = wi >> sel_bits;
    }
  }
}

}  // namespace

namespace rfxk {

int msp_k_ok(int k) { return k >= 23 && k <= 31; }
int msp_part1_block() { return MP1_BLOCK; }  // m = k-10 in 13..15 (an m-mer fits 32 bits); k+3 bases fit 56 bits

int msp_nmax_of(int k) { return msp_nmax(k); }
int msp_wide(int) { return 1; }  // (round 4) every record is a 64-bit word + a 32-bit plane
int msp_window(int k) { return msp_wl(k); }  // m-mers per k-mer (rfx_devutil.h), for the host's estimates

void msp_part1(rfx_ctx* c, const rfx_reads_view& rv, int k, int canonical, int bin_bits, uint32_t bin_lo, uint32_t bin_hi,
               int hmode, int grid, void* rec_a, uint32_t* coarse_cur, uint32_t cap_a, uint32_t* cnt_rows,
               unsigned int* flag, int slab_log2, void* map_out, uint32_t* map_ovf, uint32_t map_ovf_cap) {
  rfx_span sp(c, hmode == 1 ? "k_msp_count" : hmode == 4 ? "k_msp_map" : "k_msp_part1");
#define RFX_MSP_P1(CANON, HM, WL)                                                                             \
  hipLaunchKernelGGL((k_msp_part1<CANON, HM, WL>), dim3(grid), dim3(MP1_BLOCK), 0, c->stream, rv, k, bin_bits, \
                     bin_lo, bin_hi, (msp_rec12*)rec_a, coarse_cur, cap_a, cnt_rows, flag, slab_log2, (uint4*)map_out, \
                     map_ovf, map_ovf_cap)
#define RFX_MSP_P1_HM(CANON, WL)          \
  do {                                          \
    if (hmode == 0) RFX_MSP_P1(CANON, 0, WL);      \
    else if (hmode == 1) RFX_MSP_P1(CANON, 1, WL); \
    else if (hmode == 3) RFX_MSP_P1(CANON, 3, WL); \
    else if (hmode == 4) RFX_MSP_P1(CANON, 4, WL); \
    else RFX_MSP_P1(CANON, 2, WL);                 \
  } while (0)
#define RFX_MSP_P1_WL(WL)                      \
  case WL:                                     \
    if (canonical) RFX_MSP_P1_HM(true, WL);    \
    else RFX_MSP_P1_HM(false, WL);             \
    break
  switch (msp_wl(k)) {  // (rfx_devutil.h: 11 for k <= 26, k - 15 from there on)
    RFX_MSP_P1_WL(11);
    RFX_MSP_P1_WL(12);
    RFX_MSP_P1_WL(13);
    RFX_MSP_P1_WL(14);
    RFX_MSP_P1_WL(15);
    default:
      if (canonical) RFX_MSP_P1_HM(true, MSP_WL_WIDE);
      else RFX_MSP_P1_HM(false, MSP_WL_WIDE);
  }
#undef RFX_MSP_P1_WL
#undef RFX_MSP_P1_HM
#undef RFX_MSP_P1
}

int msp_map_grid(rfx_ctx* c, uint32_t n_reads) {  // (k_msp_part1 HMODE 4: 69 VGPRs, 18 KB of LDS: three workgroups per CU)
  const uint32_t chunks = (n_reads + MP1_BLOCK - 1) / MP1_BLOCK;
  return (int)std::max<uint32_t>(8, std::min<uint32_t>((uint32_t)c->n_cu * 3, (chunks + 7) & ~7u));
}

int msp_replay_grid(rfx_ctx* c, uint32_t n_reads) {
  const uint32_t chunks = (n_reads + MP1_BLOCK - 1) / MP1_BLOCK;
  return (int)std::max<uint32_t>(8, std::min<uint32_t>((uint32_t)c->n_cu * 2, (chunks + 7) & ~7u));
}

void msp_replay(rfx_ctx* c, const rfx_reads_view& rv, const void* map, int k, int canonical, int bin_bits, uint32_t bin_lo,
                uint32_t bin_hi, int grid, void* rec_a, uint32_t* coarse_cur, uint32_t cap_a, uint32_t* cnt_rows,
                unsigned int* flag, int slab_log2) {
  rfx_span sp(c, "k_msp_replay");
  static const bool queued = getenv("RFX_REPLAY_OLD") == nullptr;  // (A/B: a lane cuts the runs of its own read, as in round 5)
#define RFX_REPLAY(CANON, Q)                                                                                                  \
  hipLaunchKernelGGL((k_msp_replay<CANON, Q>), dim3(grid), dim3(MP1_BLOCK), 0, c->stream, rv, (const uint4*)map, k, bin_bits, \
                     bin_lo, bin_hi, (msp_rec12*)rec_a, coarse_cur, cap_a, cnt_rows, flag, slab_log2)
  if (canonical) {
    if (queued) RFX_REPLAY(true, true);
    else RFX_REPLAY(true, false);
  } else {
    if (queued) RFX_REPLAY(false, true);
    else RFX_REPLAY(false, false);
  }
#undef RFX_REPLAY
}

// Grid, staging chunk and pool of a leaf launch over P bins holding ~n_records records of which ~est_survivors k-mers
// will leave: a workgroup per ~4096 records at least (a small input does not pay for 2048 staging chunks), the big chunk
// only where it is amortised over many bins.  The pool: a chunk per workgroup + what the expected survivors fill when
// every chunk is left at its emptiest (CH - PASS_MAX entries), half as much again, + `extra` (what an earlier attempt
// came short by).
void msp_leaf_plan(rfx_ctx* c, uint32_t P, int geo, uint64_t n_records, uint64_t est_survivors, uint32_t extra, uint32_t* grid,
                   uint32_t* chunk, uint32_t* n_chunks) {
  const uint32_t per_cu = geo ? 8 : 4;  // (1..8 per CU measured: no difference)
  uint64_t g = std::min<uint64_t>(P, (uint64_t)c->n_cu * per_cu);
  g = std::min<uint64_t>(g, std::max<uint64_t>(1, n_records >> 12));
  *grid = (uint32_t)std::max<uint64_t>(g, 1);
  *chunk = MSP_LEAF_CHUNK(geo, n_records >= (1ull << 24));
  const uint64_t per = *chunk - MSP_LEAF_PASS_MAX(geo);
  uint64_t more = (est_survivors + est_survivors / 2) / per + 16 + extra;
  // (a test knob: no chunk beyond the workgroups' first ones until a rerun asks for them -- the route a pool that came
  // short takes is otherwise taken only when the survivor estimate is far off)
  if (getenv("RFX_LEAF_STAGE_TEST")) more = extra;
  *n_chunks = (uint32_t)std::min<uint64_t>((uint64_t)*grid + more, 1u << 30);
}

void msp_leaf(rfx_ctx* c, const uint64_t* const* seg_inst, const uint64_t* const* seg_bs, int nseg,
              const uint64_t* inst0, const uint64_t* bs0, uint32_t P, int k, int canonical, const uint64_t* lut,
              int ntab, int sel_bits, int shift1, uint64_t pos_lo, uint64_t pos_hi, uint64_t lower, uint64_t upper,
              uint64_t* out_w, uint32_t* out_c, uint32_t* cur, uint32_t cap, unsigned int* flag, unsigned int* err,
              unsigned int* stage_short, int geo, const uint32_t* const* seg_ext, const uint32_t* ext0, const msp_stage& st,
              uint32_t launch, uint32_t grid) {
  const int force_mixed = getenv("RFX_LEAF_FORCE_MIXED") != nullptr;
  uint32_t* const more = st.more + launch;
  {
    rfx_span sp(c, "k_msp_leaf");
#define RFX_MSP_LEAF(CANON, GEO)                                                                                        \
  hipLaunchKernelGGL((k_msp_leaf<CANON, GEO>), dim3(grid), dim3(MSP_LEAF_BLK(GEO)), 0, c->stream, seg_inst, seg_bs, nseg, \
                     inst0, bs0, seg_ext, ext0, P, k, lower, upper, st.keys, st.counts, st.chunk, st.fill, more,           \
                     st.n_chunks, flag, err, stage_short, force_mixed)
    if (canonical) {
      if (geo) RFX_MSP_LEAF(true, 1);
      else RFX_MSP_LEAF(true, 0);
    } else {
      if (geo) RFX_MSP_LEAF(false, 1);
      else RFX_MSP_LEAF(false, 0);
    }
#undef RFX_MSP_LEAF
  }
  {
    rfx_span sp(c, "k_surv_place");
    const uint32_t pg = std::min<uint32_t>(st.n_chunks, (uint32_t)c->n_cu * (uint32_t)std::max(1, 160 * 1024 / (16 * 1024 + SP_N * 12 + 2048)));
    hipLaunchKernelGGL(k_surv_place, dim3(pg), dim3(SP_BLK), 0, c->stream, st.keys, st.counts, st.chunk, st.fill, more, grid,
                       st.n_chunks, lut, ntab, sel_bits, shift1, pos_lo, pos_hi, out_w, out_c, cur, cap, flag);
  }
}

void surv_hist(rfx_ctx* c, const uint64_t* buf_a, const uint32_t* coarse_cur, uint32_t cap_a, uint32_t P2, int shift2,
               uint64_t* fine_tot, int rec_mode, int k, uint64_t n_hint) {
  rfx_span sp(c, rec_mode ? "k_rec_hist" : "k_surv_hist");
  uint32_t W = 8;
  if (n_hint > (1ull << 27)) W = (uint32_t)c->n_cu * 2 / P1_BINS * 4;  // big inputs: every CU busy, several rounds
#define RFX_SH(MODE)                                                                                                \
  hipLaunchKernelGGL(k_surv_hist<MODE>, dim3(P1_BINS * W), dim3(L2_BLOCK), 0, c->stream, buf_a, coarse_cur, cap_a, P2, \
                     shift2, W, k, (unsigned long long*)fine_tot)
  if (rec_mode == 0) RFX_SH(0);
  else if (rec_mode == 1) RFX_SH(1);
  else RFX_SH(2);
#undef RFX_SH
}

void flag_if_gt(rfx_ctx* c, const uint64_t* d_value, uint64_t limit, unsigned int* d_flag) {
  hipLaunchKernelGGL(k_flag_if_gt, dim3(1), dim3(1), 0, c->stream, d_value, limit, d_flag);
}

void histo_bins(rfx_ctx* c, const uint32_t* counts, const uint32_t* coarse_cur, uint32_t cap,
                unsigned long long* d_histo) {
  rfx_span sp(c, "k_histo");
  const uint32_t W = 16;
  hipLaunchKernelGGL(k_histo_bins, dim3(P1_BINS * W), dim3(256), 0, c->stream, counts, coarse_cur, cap, W, d_histo);
}

void surv_sort(rfx_ctx* c, const uint64_t* bw, const uint32_t* bc, const uint64_t* bs, uint32_t P, int bin_shift,
               const uint64_t* lut_inv, int ntab, int sel_bits, uint64_t* out_keys, uint32_t* out_counts,
               uint64_t* out_pos) {
  rfx_span sp(c, "k_surv_sort");
  // the resident blocks: three per CU since a bin waits in registers (42 KB of LDS) -- 29 -> 21 ms per W sample; a fourth
  // (seven tables, 40 KB) added nothing
  const uint32_t grid = P < (uint32_t)c->n_cu * 3 ? P : (uint32_t)c->n_cu * 3;
  hipLaunchKernelGGL(k_surv_sort, dim3(grid), dim3(SS_BLOCK), 0, c->stream, bw, bc, bs, P, bin_shift, lut_inv, ntab,
                     sel_bits, out_keys, out_counts, out_pos);
}

}  // namespace rfxk
