// CDNA4 (gfx950) kernels of the RUFUS hot path.  Integer / hash work: no MFMA anywhere; the
// levers are coalesced 16-byte streaming of the packed reads through LDS, an LDS-resident GF(2)
// lookup for jellyfish's matrix hash, 64-wide wave ballots for hit compaction, and a table layout
// whose slot order already is (almost) the output order.
//
// Kernel  | replaces (reference file:line)
// K2 count_reads  jf/include/jellyfish/mer_iterator.hpp:59-88 + large_hash_array.hpp:298-302,:513-744
// K3 tile_*       jf/include/jellyfish/sorted_dumper.hpp:80-112, mer_heap.hpp:34-38, histo_main.cc:33-89
// K4 flag_*/query jf/jellyfish/merge_files.cc:69-155, jf/include/jellyfish/binary_dumper.hpp:156-203
// K5 filter       src/RUFUS.Filter.cpp:196-277, src/RUFUS.Filter.ss.cpp:164-203
#include "rfx_internal.h"

namespace {

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
// pos = M * key over GF(2): XOR of one 256-entry table per key byte
// (jf/include/jellyfish/rectangular_binary_matrix.hpp:206-243 computes the same product bit by bit).
__device__ __forceinline__ uint64_t gf2_pos(const uint64_t* __restrict__ lut, uint64_t key, int ntab) {
  uint64_t r = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if (t < ntab) r ^= lut[t * 256 + (uint32_t)((key >> (8 * t)) & 255u)];
  return r;
}

// Home slot: monotone in (pos, key) so that slot order is (almost) the output order.
__device__ __forceinline__ uint64_t home_of(const rfx_table_view& tv, uint64_t pos, uint64_t key) {
  return tv.lshift ? ((pos << tv.lshift) | (key >> tv.kshift)) : (pos >> tv.rshift);
}

__device__ __forceinline__ void load_lut(uint64_t* s_lut, const uint64_t* g_lut, int ntab) {
  for (int i = threadIdx.x; i < ntab * 256; i += blockDim.x) s_lut[i] = g_lut[i];
}

// Linear-probe insert.  The plain load is only a hint: keys never change once written, so a
// non-empty value is final; a stale "empty" is corrected by the CAS.  Returns false when the probe
// ran longer than RFX_PROBE_LIMIT (or off the end): the caller diverts the key.
__device__ __forceinline__ bool table_add(const rfx_table_view& tv, uint64_t key, uint32_t inc, uint64_t home,
                                          uint32_t& n_new, uint32_t& max_disp) {
  uint64_t slot = home;
  const uint64_t last = min(tv.slots, home + RFX_PROBE_LIMIT);
  for (; slot < last; ++slot) {
    uint64_t cur = tv.keys[slot];
    if (cur == RFX_EMPTY) {
      cur = atomicCAS((unsigned long long*)&tv.keys[slot], (unsigned long long)RFX_EMPTY, (unsigned long long)key);
      if (cur == RFX_EMPTY) {
        ++n_new;
        uint32_t d = (uint32_t)(slot - home);
        max_disp = d > max_disp ? d : max_disp;
        cur = key;
      }
    }
    if (cur == key) {
      atomicAdd(&tv.counts[slot], inc);
      return true;
    }
  }
  return false;
}

__device__ __forceinline__ void flush_stats(rfx_table_stats* stats, uint32_t n_new, uint32_t max_disp,
                                            uint32_t overflow) {
  for (int off = WAVE / 2; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off);
    uint32_t o = __shfl_down(max_disp, off);
    max_disp = o > max_disp ? o : max_disp;
    overflow |= __shfl_down(overflow, off);
  }
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    if (n_new) atomicAdd(&stats->distinct, (unsigned long long)n_new);
    if (max_disp) atomicMax(&stats->max_disp, max_disp);
    if (overflow) atomicOr(&stats->overflow, 1u);
  }
}

// Stage the code and mask words of reads [r0, r0+nr) into LDS with coalesced loads.  Returns false
// (nothing staged) when the chunk is larger than the staging area.
template <int STAGE_WORDS>
__device__ __forceinline__ bool stage_chunk(const uint64_t* __restrict__ codes, const uint32_t* __restrict__ mask,
                                            uint32_t w0, uint32_t w1, uint64_t* s_codes, uint32_t* s_mask) {
  uint32_t nw = w1 - w0;
  if (nw > (uint32_t)STAGE_WORDS) return false;
  for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) {
    s_codes[i] = codes[w0 + i];
    s_mask[i] = mask[w0 + i];
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// K2: count canonical k-mers of a read block
// ---------------------------------------------------------------------------------------------
constexpr int K2_BLOCK = 512;
constexpr int K2_STAGE = K2_BLOCK * 6;  // words: reads up to 192 bases on average are staged

// Chunks of K2_BLOCK reads are handed out by ticket.  A block stops asking for chunks as soon as
// the table passes its load limit or any key had to be diverted; the host then grows the table,
// re-inserts the diverted keys and relaunches from the ticket -- nothing is lost or double counted.
template <bool CANON>
__global__ __launch_bounds__(K2_BLOCK) void k_count_reads(rfx_reads_view rv, rfx_table_view tv,
                                                           const uint64_t* __restrict__ g_lut, int k,
                                                           rfx_table_stats* stats, rfx_count_ctl* ctl,
                                                           uint64_t* __restrict__ ovf_keys, uint64_t ovf_cap,
                                                           uint64_t load_limit) {
  __shared__ uint64_t s_lut[8 * 256];
  __shared__ uint64_t s_codes[K2_STAGE];
  __shared__ uint32_t s_mask[K2_STAGE];
  __shared__ uint32_t s_chunk;
  load_lut(s_lut, g_lut, tv.ntab);

  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  const int rcshift = 2 * (k - 1);
  const uint32_t n_chunks = (rv.n + K2_BLOCK - 1) / K2_BLOCK;

  for (;;) {
    __syncthreads();  // previous chunk fully consumed (and LUT visible on the first trip)
    if (threadIdx.x == 0) {
      uint32_t c = 0xFFFFFFFFu;
      const unsigned int stop = __hip_atomic_load(&ctl->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long d = __hip_atomic_load(&stats->distinct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!stop && d > load_limit) atomicExch(&ctl->stop, 1u);
      else if (!stop) {
        // never hand out a ticket past the end: the ticket then equals the number of chunks done
        c = atomicAdd(&ctl->ticket, 1u);
        if (c >= n_chunks) {
          atomicSub(&ctl->ticket, 1u);
          c = 0xFFFFFFFFu;
        }
      }
      s_chunk = c;
    }
    __syncthreads();
    const uint32_t chunk = s_chunk;
    if (chunk == 0xFFFFFFFFu) break;

    const uint32_t r0 = chunk * K2_BLOCK;
    const uint32_t r1 = min(rv.n, r0 + K2_BLOCK);
    const uint32_t w0 = rv.word_off[r0], w1 = rv.word_off[r1];
    const bool staged = stage_chunk<K2_STAGE>(rv.codes, rv.acgt, w0, w1, s_codes, s_mask);
    __syncthreads();

    uint32_t n_new = 0, max_disp = 0;
    const uint32_t r = r0 + threadIdx.x;
    if (r < r1) {
      const uint32_t wr = rv_off(rv, r);
      const uint32_t len = rv_len(rv, r);
      const uint64_t* cw = staged ? s_codes + (wr - w0) : rv.codes + wr;
      const uint32_t* cm = staged ? s_mask + (wr - w0) : rv.acgt + wr;
      uint64_t fwd = 0, rc = 0;
      int filled = 0;
      const uint32_t nw = (len + 31) >> 5;
      for (uint32_t wi = 0; wi < nw; ++wi) {
        uint64_t w = cw[wi];
        uint32_t m = cm[wi];
        const int nb = min(32u, len - (wi << 5));
        for (int b = 0; b < nb; ++b) {
          const uint32_t code = (uint32_t)w & 3u;
          w >>= 2;
          const bool valid = m & 1u;
          m >>= 1;
          fwd = ((fwd << 2) | code) & kmask;
          if (CANON) rc = (rc >> 2) | ((uint64_t)(3u - code) << rcshift);
          filled = valid ? filled + 1 : 0;
          if (filled >= k) {
            const uint64_t key = CANON ? (rc < fwd ? rc : fwd) : fwd;
            const uint64_t pos = gf2_pos(s_lut, key, tv.ntab);
            if (pos >= tv.pos_lo && pos < tv.pos_hi &&
                !table_add(tv, key, 1u, home_of(tv, pos, key), n_new, max_disp)) {
              const unsigned long long o = atomicAdd(&ctl->ovf_n, 1ull);
              if (o < ovf_cap) ovf_keys[o] = key;
              else atomicExch(&ctl->lost, 1u);
              atomicExch(&ctl->stop, 1u);
            }
          }
        }
      }
    }
    flush_stats(stats, n_new, max_disp, 0u);
  }
}

// Merge pre-aggregated pairs (multi-GPU owner reduce, rehash, overflow re-insert).  The host sizes
// the table first, so a diverted key here is an error (stats->overflow).
__global__ __launch_bounds__(256) void k_count_pairs(const uint64_t* __restrict__ keys,
                                                      const uint32_t* __restrict__ counts, uint64_t n,
                                                      rfx_table_view tv, const uint64_t* __restrict__ g_lut,
                                                      rfx_table_stats* stats) {
  __shared__ uint64_t s_lut[8 * 256];
  load_lut(s_lut, g_lut, tv.ntab);
  __syncthreads();
  uint32_t n_new = 0, max_disp = 0, overflow = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];
    const uint32_t c = counts ? counts[i] : 1u;
    if (c == 0) continue;
    const uint64_t pos = gf2_pos(s_lut, key, tv.ntab);
    if (pos >= tv.pos_lo && pos < tv.pos_hi && !table_add(tv, key, c, home_of(tv, pos, key), n_new, max_disp))
      overflow = 1;
  }
  flush_stats(stats, n_new, max_disp, overflow);
}

// Every occupied slot as an unordered (key,count) pair (rehash source).
__global__ __launch_bounds__(256) void k_table_pairs(rfx_table_view tv, uint64_t* out_keys, uint32_t* out_counts,
                                                      unsigned long long* d_n) {
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < tv.slots;
       s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = tv.keys[s];
    const bool live = key != RFX_EMPTY;
    const unsigned long long m = __ballot(live);
    unsigned long long base = 0;
    const int lane = threadIdx.x & (WAVE - 1);
    if (lane == 0 && m) base = atomicAdd(d_n, (unsigned long long)__popcll(m));
    base = __shfl(base, 0);
    if (live) {
      const uint64_t o = base + __popcll(m & ((1ull << lane) - 1));
      out_keys[o] = key;
      out_counts[o] = tv.counts[s];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K3: table -> (pos,key)-sorted records
// A tile owns the RFX_TILE home slots [a, a+T).  Linear probing only moves an entry forward by at
// most max_disp slots, so every entry whose home is in the tile lives in [a, a+T+halo).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool tile_take(const rfx_table_view& tv, const uint64_t* s_lut, uint64_t s, uint64_t a,
                                          uint64_t lower, uint64_t upper, uint64_t& key, uint32_t& cnt,
                                          uint64_t& pos) {
  key = tv.keys[s];
  if (key == RFX_EMPTY) return false;
  cnt = tv.counts[s];
  if (cnt < lower || cnt > upper) return false;
  pos = gf2_pos(s_lut, key, tv.ntab);
  const uint64_t h = home_of(tv, pos, key);
  return h >= a && h < a + RFX_TILE;
}

__global__ __launch_bounds__(256) void k_tile_count(rfx_table_view tv, const uint64_t* __restrict__ g_lut,
                                                     uint32_t halo, uint64_t lower, uint64_t upper,
                                                     uint32_t* __restrict__ tile_counts, uint64_t n_tiles) {
  __shared__ uint64_t s_lut[8 * 256];
  __shared__ uint32_t s_n;
  load_lut(s_lut, g_lut, tv.ntab);
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t a = t * RFX_TILE;
    const uint64_t end = min(tv.slots, a + RFX_TILE + halo);
    uint32_t mine = 0;
    for (uint64_t s = a + threadIdx.x; s < end; s += blockDim.x) {
      uint64_t key, pos;
      uint32_t cnt;
      mine += tile_take(tv, s_lut, s, a, lower, upper, key, cnt, pos);
    }
    for (int off = WAVE / 2; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & (WAVE - 1)) == 0 && mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[t] = s_n;
    __syncthreads();
  }
}

// Exclusive scan of the tile counts (one block; n_tiles is small next to the table).
__global__ __launch_bounds__(1024) void k_tile_scan(const uint32_t* __restrict__ tile_counts, uint64_t n_tiles,
                                                     uint64_t* __restrict__ tile_off, uint32_t* d_max) {
  __shared__ uint64_t s_part[1024];
  __shared__ uint64_t s_carry;
  __shared__ uint32_t s_max;
  if (threadIdx.x == 0) {
    s_carry = 0;
    s_max = 0;
  }
  __syncthreads();
  uint32_t mx = 0;
  for (uint64_t base = 0; base < n_tiles; base += 1024) {
    const uint64_t i = base + threadIdx.x;
    const uint32_t v = i < n_tiles ? tile_counts[i] : 0;
    mx = v > mx ? v : mx;
    s_part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      uint64_t add = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
      __syncthreads();
      s_part[threadIdx.x] += add;
      __syncthreads();
    }
    const uint64_t incl = s_part[threadIdx.x];
    if (i < n_tiles) tile_off[i] = s_carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry += incl;
    __syncthreads();
  }
  atomicMax(&s_max, mx);
  __syncthreads();
  if (threadIdx.x == 0) {
    tile_off[n_tiles] = s_carry;
    *d_max = s_max;
  }
}

__global__ __launch_bounds__(256) void k_tile_emit(rfx_table_view tv, const uint64_t* __restrict__ g_lut,
                                                    uint32_t halo, uint64_t lower, uint64_t upper,
                                                    const uint64_t* __restrict__ tile_off, uint64_t n_tiles,
                                                    uint32_t sort_cap, uint64_t* __restrict__ out_keys,
                                                    uint32_t* __restrict__ out_counts, uint64_t* __restrict__ out_pos) {
  extern __shared__ uint64_t s_dyn[];  // [lut 2048][pos sort_cap][key sort_cap][cnt sort_cap (u32)]
  uint64_t* s_lut = s_dyn;
  uint64_t* s_pos = s_dyn + 8 * 256;
  uint64_t* s_key = s_pos + sort_cap;
  uint32_t* s_cnt = (uint32_t*)(s_key + sort_cap);
  __shared__ uint32_t s_n;
  load_lut(s_lut, g_lut, tv.ntab);
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t off = tile_off[t];
    const uint32_t n_t = (uint32_t)(tile_off[t + 1] - off);
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (n_t == 0) continue;
    const uint64_t a = t * RFX_TILE;
    const uint64_t end = min(tv.slots, a + RFX_TILE + halo);
    for (uint64_t s = a + threadIdx.x; s < end; s += blockDim.x) {
      uint64_t key, pos;
      uint32_t cnt;
      if (tile_take(tv, s_lut, s, a, lower, upper, key, cnt, pos)) {
        const uint32_t i = atomicAdd(&s_n, 1u);
        s_pos[i] = pos;
        s_key[i] = key;
        s_cnt[i] = cnt;
      }
    }
    uint32_t P = 1;
    while (P < n_t) P <<= 1;
    for (uint32_t i = n_t + threadIdx.x; i < P; i += blockDim.x) {
      s_pos[i] = ~0ull;
      s_key[i] = ~0ull;
      s_cnt[i] = 0;
    }
    __syncthreads();
    // bitonic sort by (pos, key)
    for (uint32_t kk = 2; kk <= P; kk <<= 1) {
      for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
        for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
          const uint32_t x = i ^ j;
          if (x > i) {
            const uint64_t pa = s_pos[i], pb = s_pos[x], ka = s_key[i], kb = s_key[x];
            const bool gt = pa > pb || (pa == pb && ka > kb);
            const bool up = (i & kk) == 0;
            if (gt == up) {
              s_pos[i] = pb;
              s_pos[x] = pa;
              s_key[i] = kb;
              s_key[x] = ka;
              const uint32_t ca = s_cnt[i];
              s_cnt[i] = s_cnt[x];
              s_cnt[x] = ca;
            }
          }
        }
        __syncthreads();
      }
    }
    for (uint32_t i = threadIdx.x; i < n_t; i += blockDim.x) {
      out_keys[off + i] = s_key[i];
      out_counts[off + i] = s_cnt[i];
      out_pos[off + i] = s_pos[i];
    }
    __syncthreads();
  }
}

// Count-of-counts: bin = min(count, 10001) (jf/sub_commands/histo_main.cc:40-49 with base 0, ceil 10001).
__global__ __launch_bounds__(256) void k_histo(const uint32_t* __restrict__ counts, uint64_t n,
                                                unsigned long long* __restrict__ g_histo) {
  __shared__ uint32_t s_h[RFX_HISTO_BINS];
  for (int i = threadIdx.x; i < RFX_HISTO_BINS; i += blockDim.x) s_h[i] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t c = counts[i];
    atomicAdd(&s_h[c > 10001u ? 10001u : c], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RFX_HISTO_BINS; i += blockDim.x)
    if (s_h[i]) atomicAdd(&g_histo[i], (unsigned long long)s_h[i]);
}

// SoA -> file records through an LDS byte buffer so the global stores are whole dwords.
__global__ __launch_bounds__(256) void k_format_records(const uint64_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ counts, uint64_t n,
                                                         int key_bytes, int counter_len, uint8_t* __restrict__ out) {
  __shared__ uint32_t s_buf[256 * 16 / 4];
  uint8_t* sb = (uint8_t*)s_buf;
  const int rl = key_bytes + counter_len;
  const uint64_t n_blocks = (n + 255) / 256;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t i = blk * 256 + threadIdx.x;
    if (i < n) {
      const uint64_t key = keys[i];
      uint64_t c = counts[i];
      if (counter_len < 4) {
        const uint64_t cap = (1ull << (8 * counter_len)) - 1;
        c = c > cap ? cap : c;
      }
      uint8_t* p = sb + threadIdx.x * rl;
      for (int b = 0; b < key_bytes; ++b) p[b] = (uint8_t)(key >> (8 * b));
      for (int b = 0; b < counter_len; ++b) p[key_bytes + b] = (uint8_t)(c >> (8 * b));
    }
    __syncthreads();
    const uint64_t nrec = min((uint64_t)256, n - blk * 256);
    const uint64_t nbytes = nrec * rl;
    uint8_t* o = out + blk * 256 * rl;  // 256*rl is a multiple of 4
    const uint64_t ndw = nbytes / 4;
    for (uint64_t d = threadIdx.x; d < ndw; d += blockDim.x) ((uint32_t*)o)[d] = s_buf[d];
    for (uint64_t b = ndw * 4 + threadIdx.x; b < nbytes; b += blockDim.x) o[b] = sb[b];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_parse_records(const uint8_t* __restrict__ in, uint64_t n, int key_bytes,
                                                        int counter_len, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ counts) {
  const int rl = key_bytes + counter_len;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t* p = in + i * rl;
    uint64_t key = 0, c = 0;
    for (int b = 0; b < key_bytes; ++b) key |= (uint64_t)p[b] << (8 * b);
    for (int b = 0; b < counter_len; ++b) c |= (uint64_t)p[key_bytes + b] << (8 * b);
    keys[i] = key;
    counts[i] = c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c;
  }
}

__global__ __launch_bounds__(256) void k_compute_pos(const uint64_t* __restrict__ keys, uint64_t n,
                                                      const uint64_t* __restrict__ g_lut, int ntab,
                                                      uint64_t* __restrict__ pos) {
  __shared__ uint64_t s_lut[8 * 256];
  load_lut(s_lut, g_lut, ntab);
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    pos[i] = gf2_pos(s_lut, keys[i], ntab);
}

__global__ __launch_bounds__(256) void k_check_sorted(const uint64_t* __restrict__ keys,
                                                       const uint64_t* __restrict__ pos, uint64_t n,
                                                       unsigned int* d_bad) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const bool ok = pos[i] < pos[i + 1] || (pos[i] == pos[i + 1] && keys[i] < keys[i + 1]);
    if (!ok) atomicOr(d_bad, 1u);
  }
}

// Invariants of a finished record set, checked where it lies (full-size runs are far beyond the oracle):
// out[0] += records out of strict (pos,key) order, out[1] += records whose stored pos is not M * key (& mask),
// out[2] += records with count < min_count or > max_count, out[3] += sum of the counts.
__global__ __launch_bounds__(256) void k_records_verify(const uint64_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ counts,
                                                         const uint64_t* __restrict__ pos, uint64_t n,
                                                         const uint64_t* __restrict__ g_lut, int ntab, uint64_t pos_mask,
                                                         uint32_t min_count, uint32_t max_count,
                                                         unsigned long long* __restrict__ out) {
  __shared__ uint64_t s_lut[8 * 256];
  load_lut(s_lut, g_lut, ntab);
  __syncthreads();
  unsigned long long bad_order = 0, bad_pos = 0, bad_cnt = 0, sum = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = pos[i], ky = keys[i];
    if (i + 1 < n) bad_order += !(p < pos[i + 1] || (p == pos[i + 1] && ky < keys[i + 1]));
    bad_pos += (gf2_pos(s_lut, ky, ntab) & pos_mask) != p;
    const uint32_t cc = counts[i];
    bad_cnt += cc < min_count || cc > max_count;
    sum += cc;
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    bad_order += __shfl_xor(bad_order, o);
    bad_pos += __shfl_xor(bad_pos, o);
    bad_cnt += __shfl_xor(bad_cnt, o);
    sum += __shfl_xor(sum, o);
  }
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    if (bad_order) atomicAdd(&out[0], bad_order);
    if (bad_pos) atomicAdd(&out[1], bad_pos);
    if (bad_cnt) atomicAdd(&out[2], bad_cnt);
    atomicAdd(&out[3], sum);
  }
}

// A checksum of the record MULTISET, independent of how the count was cut into shard passes, devices or slices:
// out[0] += sum of mix(key) * count, out[1] += sum of mix(key) (mod 2^64; mix = the splitmix64 finaliser).  Two runs hold
// the same (key, count) pairs iff (with overwhelming probability) both sums agree.
__device__ __forceinline__ uint64_t checksum_mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_records_checksum(const uint64_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ counts, uint64_t n,
                                                           unsigned long long* __restrict__ out) {
  unsigned long long a = 0, b = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t m = checksum_mix(keys[i]);
    a += m * counts[i];
    b += m;
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    atomicAdd(&out[0], a);
    atomicAdd(&out[1], b);
  }
}

// ---------------------------------------------------------------------------------------------
// K4: set difference on sorted records
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flag_range(const uint32_t* __restrict__ counts, uint64_t n, uint32_t lo,
                                                     uint32_t hi, uint8_t* __restrict__ flags) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = counts[i];
    flags[i] = c >= lo && c <= hi;
  }
}

__device__ __forceinline__ uint64_t lower_bound(const uint64_t* __restrict__ bkeys, const uint64_t* __restrict__ bpos,
                                                uint64_t nb, uint64_t p, uint64_t k) {
  uint64_t lo = 0, hi = nb;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    const uint64_t mp = bpos[mid];
    const bool less = mp < p || (mp == p && bkeys[mid] < k);
    if (less) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// lower_bound with a starting guess: pos is a hash, so record number pos * nb / 2^lsize is within a few
// sqrt(nb) of the answer.  Gallop out from there until the target is bracketed, then bisect the bracket:
// ~14 probes inside a few KB instead of log2(nb) ~ 23 probes across the whole array.
__device__ __forceinline__ uint64_t lower_bound_near(const uint64_t* __restrict__ bkeys,
                                                     const uint64_t* __restrict__ bpos, uint64_t nb, uint64_t p,
                                                     uint64_t k, int lsize) {
  auto less_at = [&](uint64_t i) {
    const uint64_t mp = bpos[i];
    return mp < p || (mp == p && bkeys[i] < k);
  };
  uint64_t g = (uint64_t)(((unsigned __int128)p * nb) >> lsize);
  if (g >= nb) g = nb - 1;
  uint64_t lo, hi;  // invariant: every index < lo is less, every index >= hi is not
  if (less_at(g)) {
    lo = g + 1;
    hi = nb;
    for (uint64_t step = 256; lo < nb; step <<= 1) {
      const uint64_t j = lo + step - 1 < nb ? lo + step - 1 : nb - 1;
      if (less_at(j)) lo = j + 1;
      else {
        hi = j;
        break;
      }
    }
  } else {
    hi = g;
    lo = 0;
    for (uint64_t step = 256; hi > 0; step <<= 1) {
      const uint64_t j = hi >= step ? hi - step : 0;
      if (less_at(j)) {
        lo = j + 1;
        break;
      }
      hi = j;
    }
  }
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (less_at(mid)) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// Clear the flag of every record that also occurs in B (search by (pos,key)).
__global__ __launch_bounds__(256) void k_flag_absent(const uint64_t* __restrict__ keys,
                                                      const uint64_t* __restrict__ pos, uint64_t n,
                                                      const uint64_t* __restrict__ bkeys,
                                                      const uint64_t* __restrict__ bpos, uint64_t nb, int lsize,
                                                      uint8_t* __restrict__ flags) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (!flags[i]) continue;
    const uint64_t k = keys[i], p = pos[i];
    const uint64_t lo = lower_bound_near(bkeys, bpos, nb, p, k, lsize);
    if (lo < nb && bkeys[lo] == k) flags[i] = 0;
  }
}

// The same for big inputs (a 30x sample's candidates against a control's records: 1.5e9 lookups of ~14 dependent
// probes each through L2): both lists are in (pos,key) order, so the records of B that can match a tile of 2048
// candidates are one short range -- between the lower bounds of the tile's first candidate and of the next tile's
// (k_fa_bounds: one search per tile) -- which is staged in LDS and bisected there.
// The tile is sized by the launcher so that the control's range of a tile fits the LDS with room to spare (2048
// candidates at most, fewer when the control holds more records than there are candidates).
// Small tiles (16 KB of LDS): eight workgroups per CU take turns at fetching and searching; with 48 KB tiles three did,
// and the kernel ran at a quarter of the HBM rate.
constexpr int FA_TILE_MAX = 768, FA_BCAP = 1024, FA_STEPS = 11;  // 2^FA_STEPS > FA_BCAP

__global__ __launch_bounds__(256) void k_fa_bounds(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ pos,
                                                    uint64_t n, const uint64_t* __restrict__ bkeys,
                                                    const uint64_t* __restrict__ bpos, uint64_t nb, int lsize,
                                                    uint64_t n_tiles, uint32_t tile, uint64_t* __restrict__ bounds /* n_tiles + 1 */) {
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n_tiles; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = t * tile;
    bounds[t] = i < n ? lower_bound_near(bkeys, bpos, nb, pos[i], keys[i], lsize) : nb;
  }
}

__global__ __launch_bounds__(256) void k_flag_absent_tiled(const uint64_t* __restrict__ keys,
                                                            const uint64_t* __restrict__ pos, uint64_t n,
                                                            const uint64_t* __restrict__ bkeys,
                                                            const uint64_t* __restrict__ bpos, uint64_t nb, int lsize,
                                                            uint64_t n_tiles, uint32_t tile,
                                                            const uint64_t* __restrict__ bounds,
                                                            uint8_t* __restrict__ flags) {
  __shared__ uint64_t s_bk[FA_BCAP], s_bp[FA_BCAP];
  // (the bounds of the next tile are fetched a tile ahead: what the loads of a tile need is there when they are issued)
  uint64_t nx_lo = 0, nx_hi = 0;
  if (blockIdx.x < n_tiles) {
    nx_lo = bounds[blockIdx.x];
    nx_hi = bounds[blockIdx.x + 1];
  }
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t a0 = t * tile, a1 = a0 + tile < n ? a0 + tile : n;
    const uint64_t lo = nx_lo, hi = nx_hi < nb ? nx_hi + 1 : nb;  // B records that can match the tile
    if (t + gridDim.x < n_tiles) {
      nx_lo = bounds[t + gridDim.x];
      nx_hi = bounds[t + gridDim.x + 1];
    }
    const uint32_t nbr = hi - lo <= (uint64_t)FA_BCAP ? (uint32_t)(hi - lo) : 0u;
    if (hi - lo <= (uint64_t)FA_BCAP) {
      // the tile's candidates are fetched together with the control's range: one HBM round trip per tile (a load per
      // candidate inside the search loop, flag first, was seventeen of them one after the other: 81 us per tile)
      constexpr int PER = FA_TILE_MAX / 256;  // (tile <= FA_TILE_MAX: what lies beyond a1 is not fetched)
      uint64_t ck[PER], cp[PER];
      uint8_t cf[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const uint64_t i = a0 + threadIdx.x + (uint64_t)u * 256;
        const bool in = i < a1;
        cf[u] = in ? flags[i] : (uint8_t)0;
        ck[u] = in ? keys[i] : 0;
        cp[u] = in ? pos[i] : 0;
      }
      for (uint32_t j = threadIdx.x; j < nbr; j += blockDim.x) {
        s_bk[j] = bkeys[lo + j];
        s_bp[j] = bpos[lo + j];
      }
      __syncthreads();
      // the lane's searches advance together, one comparison each per step: eight independent LDS round trips in flight
      // instead of a chain of 11 x 2 per candidate
      uint32_t at[PER], len[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        at[u] = 0;
        len[u] = cf[u] ? nbr : 0u;
      }
      for (int step = 0; step < FA_STEPS; ++step) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const bool act = len[u] > 0;
          const uint32_t half = len[u] >> 1, mid = at[u] + half;
          const uint64_t mp = s_bp[act ? mid : 0u], mk = s_bk[act ? mid : 0u];
          const bool less = mp < cp[u] || (mp == cp[u] && mk < ck[u]);
          at[u] = act && less ? mid + 1 : at[u];
          len[u] = act ? (less ? len[u] - half - 1 : half) : 0u;
        }
      }
#pragma unroll
      for (int u = 0; u < PER; ++u)
        if (cf[u] && at[u] < nbr && s_bk[at[u]] == ck[u] && s_bp[at[u]] == cp[u])
          flags[a0 + threadIdx.x + (uint64_t)u * 256] = 0;
      __syncthreads();
    } else {  // few candidates against many records: every candidate searches for itself
      for (uint64_t i = a0 + threadIdx.x; i < a1; i += blockDim.x) {
        if (!flags[i]) continue;
        const uint64_t k = keys[i], p = pos[i];
        const uint64_t at = lower_bound_near(bkeys, bpos, nb, p, k, lsize);
        if (at < nb && bkeys[at] == k) flags[i] = 0;
      }
    }
  }
}

constexpr int CP_ITEMS = 2048;  // elements per compaction block

__global__ __launch_bounds__(256) void k_compact_count(const uint8_t* __restrict__ flags, uint64_t n,
                                                        uint64_t* __restrict__ block_cnt) {
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * CP_ITEMS;
  uint32_t mine = 0;
  for (uint32_t j = threadIdx.x; j < CP_ITEMS; j += blockDim.x)
    if (base + j < n) mine += flags[base + j] != 0;
  for (int off = WAVE / 2; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & (WAVE - 1)) == 0 && mine) atomicAdd(&s_n, mine);
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_n;
}

__global__ __launch_bounds__(1024) void k_scan_u64(uint64_t* __restrict__ v, uint64_t n,
                                                    unsigned long long* d_total) {
  // in-place exclusive scan, one block
  __shared__ uint64_t s_part[1024];
  __shared__ uint64_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n; base += 1024) {
    const uint64_t i = base + threadIdx.x;
    const uint64_t x = i < n ? v[i] : 0;
    s_part[threadIdx.x] = x;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      uint64_t add = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
      __syncthreads();
      s_part[threadIdx.x] += add;
      __syncthreads();
    }
    const uint64_t incl = s_part[threadIdx.x];
    if (i < n) v[i] = s_carry + incl - x;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *d_total = s_carry;
}

// Order-preserving scatter: each wave owns a contiguous run of 64*32 flags of its block.
__global__ __launch_bounds__(64) void k_compact_scatter(const uint8_t* __restrict__ flags,
                                                         const uint64_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ counts,
                                                         const uint64_t* __restrict__ pos, uint64_t n,
                                                         const uint64_t* __restrict__ block_off,
                                                         uint64_t* __restrict__ out_keys,
                                                         uint32_t* __restrict__ out_counts,
                                                         uint64_t* __restrict__ out_pos) {
  const uint64_t base = (uint64_t)blockIdx.x * CP_ITEMS;
  uint64_t o = block_off[blockIdx.x];
  const int lane = threadIdx.x;
  // four groups of flags are fetched before any is looked at (one flag load, then the loads it allows, per trip made
  // a block of mostly-zero flags -- the strike-out against a control -- 64 dependent round trips long)
  constexpr int G = 4;
  static_assert(CP_ITEMS % (G * WAVE) == 0, "");
  for (uint32_t j = 0; j < CP_ITEMS; j += G * WAVE) {
    bool f[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const uint64_t i = base + j + u * WAVE + lane;
      f[u] = i < n && flags[i] != 0;
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const uint64_t i = base + j + u * WAVE + lane;
      const unsigned long long m = __ballot(f[u]);
      if (f[u]) {
        const uint64_t d = o + __popcll(m & ((1ull << lane) - 1));
        out_keys[d] = keys[i];
        out_counts[d] = counts[i];
        if (out_pos) out_pos[d] = pos[i];
      }
      o += __popcll(m);
    }
  }
}

__global__ __launch_bounds__(256) void k_query(const uint64_t* __restrict__ qkeys, uint64_t nq,
                                                const uint64_t* __restrict__ g_lut, int ntab,
                                                const uint64_t* __restrict__ keys, const uint64_t* __restrict__ pos,
                                                const uint32_t* __restrict__ counts, uint64_t n,
                                                uint32_t* __restrict__ out) {
  __shared__ uint64_t s_lut[8 * 256];
  load_lut(s_lut, g_lut, ntab);
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t k = qkeys[i];
    const uint64_t p = gf2_pos(s_lut, k, ntab);
    const uint64_t lo = lower_bound(keys, pos, n, p, k);
    out[i] = (lo < n && keys[lo] == k) ? counts[lo] : 0u;
  }
}

// ---------------------------------------------------------------------------------------------
// K5: read filter
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t set_hash(uint64_t key, int bits) {
  uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  uint32_t h = (lo ^ (hi * 0x9E3779B1u)) * 0x85EBCA6Bu;
  h ^= h >> 15;
  h *= 0xC2B2AE35u;
  return h >> (32 - bits);
}

__global__ __launch_bounds__(256) void k_set_insert(const uint64_t* __restrict__ keys, uint64_t n,
                                                     uint64_t* __restrict__ slots, int bits) {
  const uint32_t mask = (1u << bits) - 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];
    if (key == RFX_EMPTY) continue;  // carried by has_all_ones
    uint32_t s = set_hash(key, bits);
    for (;;) {
      const uint64_t cur =
          atomicCAS((unsigned long long*)&slots[s], (unsigned long long)RFX_EMPTY, (unsigned long long)key);
      if (cur == RFX_EMPTY || cur == key) break;
      s = (s + 1) & mask;
    }
  }
}

constexpr int K5_BLOCK = 256;
constexpr int K5_LDS_BM_BITS = 16;  // a 2^16-bit (8 KB) window bitmap is copied to LDS

__device__ __forceinline__ bool set_has(const uint64_t* __restrict__ slots, int bits, uint64_t key) {
  const uint32_t mask = (1u << bits) - 1;
  uint32_t s = set_hash(key, bits);
  for (;;) {
    const uint64_t cur = slots[s];
    if (cur == key) return true;
    if (cur == RFX_EMPTY) return false;
    s = (s + 1) & mask;
  }
}

// One bit per value of bm_bits window bits of the forward word (no hashing: the rolling word IS the
// index).  A clear bit proves the window is not in the set; only set bits pay for the real probe.
__global__ __launch_bounds__(256) void k_set_bitmap(const uint64_t* __restrict__ keys, uint64_t n,
                                                     uint32_t* __restrict__ bm, int bm_bits, int bm_shift) {
  const uint32_t mask = (1u << bm_bits) - 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t idx = (uint32_t)(keys[i] >> bm_shift) & mask;
    atomicOr(&bm[idx >> 5], 1u << (idx & 31));
  }
}

// Thread per read, words straight from HBM/L2 (a wave's 64 reads are one contiguous 2.5 KB span),
// 32 waves per CU.  Per base: roll the forward word, roll the good streak, test one bitmap bit.
__global__ __launch_bounds__(K5_BLOCK) void k_filter(rfx_reads_view rv, const uint64_t* __restrict__ g_slots, int bits,
                                                      int has_all_ones, const uint32_t* __restrict__ g_bm, int bm_bits,
                                                      int bm_shift, int k, int thresh, int last_base_skipped,
                                                      uint32_t* __restrict__ hits_out,
                                                      uint64_t* __restrict__ hitmask,
                                                      unsigned long long* __restrict__ d_nhit) {
  __shared__ uint32_t s_bm[1 << (K5_LDS_BM_BITS - 5)];
  const bool lds_bm = bm_bits == K5_LDS_BM_BITS;
  if (lds_bm) {
    for (uint32_t i = threadIdx.x; i < (1u << (K5_LDS_BM_BITS - 5)); i += blockDim.x) s_bm[i] = g_bm[i];
    __syncthreads();
  }
  const uint32_t* __restrict__ bm = lds_bm ? s_bm : g_bm;
  const uint32_t bm_mask = (1u << bm_bits) - 1;
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);

  const uint32_t n_chunks = (rv.n + K5_BLOCK - 1) / K5_BLOCK;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r = chunk * K5_BLOCK + threadIdx.x;
    uint32_t found = 0;
    if (r < rv.n) {
      const uint32_t wr = rv_off(rv, r);
      const uint32_t len = rv_len(rv, r);
      const uint64_t* __restrict__ cw = rv.codes + wr;
      const uint32_t* __restrict__ cm = rv.good + wr;
      // src/RUFUS.Filter.cpp:203: `i < length()-1` -- the last base is never examined.
      const uint32_t stop = last_base_skipped ? (len ? len - 1 : 0) : len;
      uint64_t fwd = 0;
      int streak = 0;
      const uint32_t nw = (stop + 31) >> 5;
      for (uint32_t wi = 0; wi < nw; ++wi) {
        uint64_t w = cw[wi];
        uint32_t m = cm[wi];
        const int nb = min(32u, stop - (wi << 5));
        for (int b = 0; b < nb; ++b) {
          fwd = ((fwd << 2) | (w & 3u)) & kmask;
          w >>= 2;
          streak = (m & 1u) ? streak + 1 : 0;
          m >>= 1;
          if (streak >= k) {
            const uint32_t idx = (uint32_t)(fwd >> bm_shift) & bm_mask;
            if ((bm[idx >> 5] >> (idx & 31)) & 1u)
              found += fwd == RFX_EMPTY ? (has_all_ones != 0) : set_has(g_slots, bits, fwd);
          }
        }
      }
      if (hits_out) hits_out[r] = found;
    }
    // wave ballot -> one 64-bit word of the hit mask per 64 reads
    const bool pass = r < rv.n && (int)found >= thresh;
    const unsigned long long mm = __ballot(pass);
    if ((threadIdx.x & (WAVE - 1)) == 0 && r < rv.n) {
      if (hitmask) hitmask[r >> 6] = mm;
      if (mm) atomicAdd(d_nhit, (unsigned long long)__popcll(mm));
    }
  }
}

// ---- fast filter (k >= 16, sets of <= 4096 keys) -------------------------------------------------
// k_filter rolls a 64-bit word and a streak counter base by base: ~60 issue slots per base, a dependent
// chain.  Here nothing rolls: the packed words ARE the windows.
//   * V = "a fully good window ends here" for the 32 bases of a word at once, by AND-ing shifted copies
//     of the good mask (run-length doubling, ~20 instructions per word);
//   * two pre-filter bitmaps in LDS (2^16 bits each): A over the last 8 bases of the k-mer, B over the 8
//     bases before those, both indexed by the packed codes as they lie in the word -- a constant-shift
//     extract once the j loop is unrolled, no 64-bit arithmetic, and the 64 LDS reads of a word are
//     independent (all in flight).  With ~1000 keys a random window passes both with probability 2e-4;
//   * a window that passes both is cut out of the two packed words with one 128-bit shift, turned into
//     the forward key (2-bit groups reversed) and probed exactly.
constexpr int FF_NB = 8;

__global__ __launch_bounds__(256) void k_set_bitmap_packed(const uint64_t* __restrict__ keys, uint64_t n,
                                                            uint32_t* __restrict__ bm /* A then B */) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];  // first base most significant: the last base is bits [1:0]
    uint32_t ia = 0, ib = 0;
#pragma unroll
    for (int b = 0; b < FF_NB; ++b) {  // packed order: the earlier base in the lower bits
      ia |= (uint32_t)((key >> (2 * b)) & 3u) << (2 * (FF_NB - 1 - b));
      ib |= (uint32_t)((key >> (2 * (b + FF_NB))) & 3u) << (2 * (FF_NB - 1 - b));
    }
    atomicOr(&bm[ia >> 5], 1u << (ia & 31));
    atomicOr(&bm[2048 + (ib >> 5)], 1u << (ib & 31));
  }
}

__global__ __launch_bounds__(K5_BLOCK) void k_filter_fast(rfx_reads_view rv, const uint64_t* __restrict__ g_slots,
                                                           int bits, int has_all_ones,
                                                           const uint32_t* __restrict__ g_bm, int k, int thresh,
                                                           int last_base_skipped, uint32_t* __restrict__ hits_out,
                                                           uint64_t* __restrict__ hitmask,
                                                           unsigned long long* __restrict__ d_nhit) {
  __shared__ uint32_t s_bm[4096];  // A: [0, 2048), B: [2048, 4096)
  for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) s_bm[i] = g_bm[i];
  __syncthreads();
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  const uint32_t n_chunks = (rv.n + K5_BLOCK - 1) / K5_BLOCK;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r = chunk * K5_BLOCK + threadIdx.x;
    uint32_t found = 0;
    if (r < rv.n) {
      const uint32_t wr = rv_off(rv, r);
      const uint32_t len = rv_len(rv, r);
      const uint64_t* __restrict__ cw = rv.codes + wr;
      const uint32_t* __restrict__ cm = rv.good + wr;
      // src/RUFUS.Filter.cpp:203: `i < length()-1` -- the last base is never examined.
      const uint32_t stop = last_base_skipped ? (len ? len - 1 : 0) : len;
      const uint32_t nw = (stop + 31) >> 5;
      uint64_t prev_c = 0;
      uint32_t prev_g = 0;
      for (uint32_t wi = 0; wi < nw; ++wi) {
        const uint64_t cur_c = cw[wi];
        uint32_t cur_g = cm[wi];
        const uint32_t nb = stop - (wi << 5);
        if (nb < 32) cur_g &= (1u << nb) - 1;  // positions at or beyond `stop` end no window
        // V bit (32 + j): the k good bits ending at base j of this word are all set (the positions before
        // the read are zero bits of prev_g, so a window cannot start before the read either)
        const uint64_t X = ((uint64_t)cur_g << 32) | prev_g;
        uint64_t acc = ~0ull, run = X;
        int off = 0;
        for (int bit = 0; (k >> bit) != 0; ++bit) {
          if ((k >> bit) & 1) {
            acc &= run << off;
            off += 1 << bit;
          }
          run &= run << (1 << bit);
        }
        const uint32_t V = (uint32_t)(acc >> 32);
        if (V) {
          uint32_t cand = 0;
#pragma unroll
          for (int j = 0; j < 32; ++j) {
            // packed codes of bases j-7 .. j and j-15 .. j-8 (may reach into the previous word)
            const int sa = 2 * (j - FF_NB + 1), sb = 2 * (j - 2 * FF_NB + 1);
            const uint32_t ia = (uint32_t)(sa >= 0 ? cur_c >> sa : (cur_c << (-sa)) | (prev_c >> (64 + sa))) & 0xFFFFu;
            const uint32_t ib = (uint32_t)(sb >= 0 ? cur_c >> sb : (cur_c << (-sb)) | (prev_c >> (64 + sb))) & 0xFFFFu;
            const uint32_t hit = (s_bm[ia >> 5] >> (ia & 31)) & (s_bm[2048 + (ib >> 5)] >> (ib & 31)) & 1u;
            cand |= hit << j;
          }
          cand &= V;
          while (cand) {  // rare
            const int j = __ffs(cand) - 1;
            cand &= cand - 1;
            // window = bases j-k+1 .. j of (prev word, this word): one 128-bit shift, then the 2-bit groups
            // reversed to get the forward key (first base most significant)
            const int sh = 2 * (j - k + 1) + 64;  // 2 .. 126
            const uint64_t packed = sh >= 64 ? cur_c >> (sh - 64) : (prev_c >> sh) | (cur_c << (64 - sh));
            uint64_t y = __brevll(packed);  // reverses bits: swap them back inside every pair
            y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
            const uint64_t fwd = (k == 32 ? y : y >> (64 - 2 * k)) & kmask;
            found += fwd == RFX_EMPTY ? (has_all_ones != 0) : set_has(g_slots, bits, fwd);
          }
        }
        prev_c = cur_c;
        prev_g = cur_g;
      }
      if (hits_out) hits_out[r] = found;
    }
    // wave ballot -> one 64-bit word of the hit mask per 64 reads
    const bool pass = r < rv.n && (int)found >= thresh;
    const unsigned long long mm = __ballot(pass);
    if ((threadIdx.x & (WAVE - 1)) == 0 && r < rv.n) {
      if (hitmask) hitmask[r >> 6] = mm;
      if (mm) atomicAdd(d_nhit, (unsigned long long)__popcll(mm));
    }
  }
}

inline int grid_for(rfx_ctx* c, uint64_t work_items, int block, int per_cu) {
  uint64_t blocks = (work_items + block - 1) / block;
  uint64_t cap = (uint64_t)c->n_cu * per_cu;
  if (blocks < 1) blocks = 1;
  return (int)(blocks < cap ? blocks : cap);
}


// k_filter_fast's idea for sets beyond 4096 keys (a 30x WGS trio with 1000 de-novo SNVs has 5e4): the two 2^16-bit
// bitmaps would be nearly full, so ONE bitmap over the last 10 bases of a window -- 2^20 bits = 128 KB of LDS, one
// 1024-thread workgroup per CU -- takes their place: 5 % of the positions pass at 5e4 keys and are probed exactly.
constexpr int FB_NB = 10;
constexpr int FB_WORDS = 1 << (2 * FB_NB - 5);
constexpr int FB_BLOCK = 1024;

__global__ __launch_bounds__(256) void k_set_bitmap_big(const uint64_t* __restrict__ keys, uint64_t n,
                                                         uint32_t* __restrict__ bm) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];  // first base most significant: the last base is bits [1:0]
    uint32_t ia = 0;
#pragma unroll
    for (int b = 0; b < FB_NB; ++b) ia |= (uint32_t)((key >> (2 * b)) & 3u) << (2 * (FB_NB - 1 - b));  // packed order
    atomicOr(&bm[ia >> 5], 1u << (ia & 31));
  }
}

__global__ __launch_bounds__(FB_BLOCK) void k_filter_big(rfx_reads_view rv, const uint64_t* __restrict__ g_slots, int bits,
                                                          int has_all_ones, const uint32_t* __restrict__ g_bm, int k,
                                                          int thresh, int last_base_skipped,
                                                          uint32_t* __restrict__ hits_out, uint64_t* __restrict__ hitmask,
                                                          unsigned long long* __restrict__ d_nhit) {
  extern __shared__ uint32_t s_bmb[];  // FB_WORDS
  for (uint32_t i = threadIdx.x; i < (uint32_t)FB_WORDS; i += blockDim.x) s_bmb[i] = g_bm[i];
  __syncthreads();
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  const uint32_t n_chunks = (rv.n + FB_BLOCK - 1) / FB_BLOCK;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r = chunk * FB_BLOCK + threadIdx.x;
    uint32_t found = 0;
    if (r < rv.n) {
      const uint32_t wr = rv_off(rv, r);
      const uint32_t len = rv_len(rv, r);
      const uint64_t* __restrict__ cw = rv.codes + wr;
      const uint32_t* __restrict__ cm = rv.good + wr;
      const uint32_t stop = last_base_skipped ? (len ? len - 1 : 0) : len;  // src/RUFUS.Filter.cpp:203
      const uint32_t nw = (stop + 31) >> 5;
      uint64_t prev_c = 0;
      uint32_t prev_g = 0;
      for (uint32_t wi = 0; wi < nw; ++wi) {
        const uint64_t cur_c = cw[wi];
        uint32_t cur_g = cm[wi];
        const uint32_t nb = stop - (wi << 5);
        if (nb < 32) cur_g &= (1u << nb) - 1;
        const uint64_t X = ((uint64_t)cur_g << 32) | prev_g;
        uint64_t acc = ~0ull, run = X;
        int off = 0;
        for (int bit = 0; (k >> bit) != 0; ++bit) {  // V: the k good bits ending here are all set (see k_filter_fast)
          if ((k >> bit) & 1) {
            acc &= run << off;
            off += 1 << bit;
          }
          run &= run << (1 << bit);
        }
        const uint32_t V = (uint32_t)(acc >> 32);
        if (V) {
          uint32_t cand = 0;
#pragma unroll
          for (int j = 0; j < 32; ++j) {
            const int sa = 2 * (j - FB_NB + 1);
            const uint32_t ia = (uint32_t)(sa >= 0 ? cur_c >> sa : (cur_c << (-sa)) | (prev_c >> (64 + sa))) & ((1u << (2 * FB_NB)) - 1);
            cand |= ((s_bmb[ia >> 5] >> (ia & 31)) & 1u) << j;
          }
          cand &= V;
          while (cand) {
            const int j = __ffs(cand) - 1;
            cand &= cand - 1;
            const int sh = 2 * (j - k + 1) + 64;  // 2 .. 126
            const uint64_t packed = sh >= 64 ? cur_c >> (sh - 64) : (prev_c >> sh) | (cur_c << (64 - sh));
            uint64_t y = __brevll(packed);
            y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
            const uint64_t fwd = (k == 32 ? y : y >> (64 - 2 * k)) & kmask;
            found += fwd == RFX_EMPTY ? (has_all_ones != 0) : set_has(g_slots, bits, fwd);
          }
        }
        prev_c = cur_c;
        prev_g = cur_g;
      }
      if (hits_out) hits_out[r] = found;
    }
    const bool pass = r < rv.n && (int)found >= thresh;
    const unsigned long long mm = __ballot(pass);
    if ((threadIdx.x & (WAVE - 1)) == 0 && r < rv.n) {
      if (hitmask) hitmask[r >> 6] = mm;
      if (mm) atomicAdd(d_nhit, (unsigned long long)__popcll(mm));
    }
  }
}

// ---- queue filter (k >= 10, sets of <= 2^18 keys): the K5 of round 3 -----------------------------------------------
// What k_filter_big did not do (97 ms for the 6.2e8 reads of config W, 0.049 of the HBM roofline):
//   * a lane loaded its read one 8-byte word per loop trip, 40 bytes apart from its neighbour's: the 2.5 KB span of a
//     wave was fetched five times through an L1 that the other 15 waves had flushed in between.  A uniform 150 bp read is
//     now loaded whole (5 + 5 words in one burst of loads) before anything is computed;
//   * every window that passed the bitmap was probed on the spot: a dependent L2 round trip per candidate, taken by
//     the whole wave whenever ONE lane had a candidate (3-4 trips per word, 20 per read).  Candidates now go into a
//     per-wave LDS queue (ballot + mbcnt compaction, 4 bytes each: lane of the read | position) and are probed 64 at a
//     time by full waves, which cut the window out of the packed read themselves;
//   * the bitmap is sized by the set (2^16 .. 2^20 bits, <= 5 % full): word = LOW bits of the packed 10-mer, bit = the 5
//     bits above them, so address and bit index are one alignbit + and / shift each: 5 VALU + 1 LDS read per window;
//     groups of 8 positions where no lane has a fully good window (the first k-1 bases, the tail) are skipped.
constexpr int FQ_NB = 10;
constexpr int FQ_QCAP = 128;  // queued candidates per wave (12 bytes each: key + read)

__global__ __launch_bounds__(256) void k_set_bitmap_q(const uint64_t* __restrict__ keys, uint64_t n,
                                                       uint32_t* __restrict__ bm, int bm_bits, int two) {
  const uint32_t wmask = (1u << (bm_bits - 5)) - 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];  // first base most significant: the last base is bits [1:0]
    uint32_t ia = 0;
#pragma unroll
    for (int b = 0; b < FQ_NB; ++b) ia |= (uint32_t)((key >> (2 * b)) & 3u) << (2 * (FQ_NB - 1 - b));  // packed order
    uint32_t bitsel = 1u << ((ia >> (bm_bits - 5)) & 31u);
    // second bit of the same word (k >= 13): chosen by 5 bits of the three bases BEFORE the 10-mer (the kernel takes them
    // from the stream with one more alignbit): a random window passes both with probability ~1 %, not 4.7 %
    if (two) {
      uint32_t ib = 0;
#pragma unroll
      for (int b = 0; b < 3; ++b) ib |= (uint32_t)((key >> (2 * (FQ_NB + b))) & 3u) << (2 * (2 - b));  // packed order, 6 bits
      bitsel |= 1u << (ib >> 1);
    }
    atomicOr(&bm[ia & wmask], bitsel);
  }
}

// 32 bits of the packed base stream (prev word's high half | cur) that start 2 bits below the 10-mer ending at base J
// of `cur`: bits [2, 2 + bm_bits - 5) are the bitmap word's byte address / 4, the 5 bits above them the bit.
template <int J, int BELOW = 2>
__device__ __forceinline__ uint32_t fq_window(uint32_t p1, uint32_t c0, uint32_t c1) {
  constexpr int t = 2 * J + 14 - BELOW;  // 32 + 2 (J - 9) - BELOW, counted from bit 0 of p1
  if (t < 32) return __builtin_amdgcn_alignbit(c0, p1, t);
  if (t == 32) return c0;
  if (t < 64) return __builtin_amdgcn_alignbit(c1, c0, t - 32);
  return c1 >> (t - 64);
}

template <int G, bool TWO>
__device__ __forceinline__ uint32_t fq_group(const uint32_t* s_bm, uint32_t amask, int ishift, uint32_t p1, uint32_t c0,
                                             uint32_t c1, uint32_t cand) {
  uint32_t x[8], w[8];
#define FQ_X(u) x[u] = fq_window<8 * G + u>(p1, c0, c1)
  FQ_X(0); FQ_X(1); FQ_X(2); FQ_X(3); FQ_X(4); FQ_X(5); FQ_X(6); FQ_X(7);
#undef FQ_X
  uint32_t x2[8];
#define FQ_X2(u) x2[u] = TWO ? fq_window<8 * G + u, 5>(p1, c0, c1) : 0u
  FQ_X2(0); FQ_X2(1); FQ_X2(2); FQ_X2(3); FQ_X2(4); FQ_X2(5); FQ_X2(6); FQ_X2(7);
#undef FQ_X2
  // (the bitmap starts at LDS address 0 -- the kernel checks it --, so the masked window IS the address: going through
  // the array's symbol costs a v_add_u32 v, 0, v per window that the compiler does not fold)
  typedef __attribute__((address_space(3))) const uint32_t lds_word;
#pragma unroll
  for (int u = 0; u < 8; ++u) w[u] = *(lds_word*)(uintptr_t)(x[u] & amask);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    uint32_t t = w[u] >> ((x[u] >> ishift) & 31u);
    if (TWO) t &= w[u] >> (x2[u] & 31u);  // (the 5 bits below the 10-mer)
    cand = __builtin_amdgcn_alignbit(t, cand, 1);
  }
  return cand;
}

#ifdef FQ_TIMING
__device__ unsigned long long g_fq[8];
}  // namespace
extern "C" int rfx_debug_fq(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fq), sizeof(g_fq)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fq), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
namespace {
#endif
// UW > 0: every read of the (compact) block has UW code words, loaded in one burst.
// Hit counts go to g_hits (zeroed by the caller) by atomic add -- hits are rare --, so the queue need not be empty when
// a chunk of reads ends: it is drained when it is full, whichever chunk its candidates came from.  An entry is the
// forward KEY of the window (cut out of the lane's registers when it is queued) + the read: a first version queued
// (read, position) and let the draining lanes fetch the words again -- by then evicted from L2, a random HBM access per
// candidate that cost as much as everything else together (43 ms with, 27 ms without the drain at W).
// k_hits_mask turns the counts into the per-read mask afterwards.
template <int BLOCK, int UW, bool TWO>
__global__ __launch_bounds__(BLOCK) void k_filter_q(rfx_reads_view rv, const uint64_t* __restrict__ g_slots, int bits,
                                                     int has_all_ones, const uint32_t* __restrict__ g_bm, int bm_bits,
                                                     int k, int last_base_skipped, uint32_t* __restrict__ g_hits) {
#ifdef FQ_TIMING
  const unsigned long long t_kernel0 = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ uint32_t s_fq[];  // bitmap | per-wave queues: FQ_QCAP keys (64-bit), then FQ_QCAP reads
  const uint32_t bm_words = 1u << (bm_bits - 5);
  const uint32_t wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  uint32_t* s_bm = s_fq;
  uint64_t* s_qk = (uint64_t*)(s_fq + bm_words) + wave * FQ_QCAP;
  uint32_t* s_qr = s_fq + bm_words + (BLOCK / WAVE) * FQ_QCAP * 2 + wave * FQ_QCAP;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)s_fq != 0u) __builtin_trap();  // see fq_group
  for (uint32_t i = threadIdx.x; i < bm_words; i += BLOCK) s_bm[i] = g_bm[i];
  __syncthreads();
  const uint32_t amask = (bm_words - 1) << 2;
  const int ishift = bm_bits - 5 + 2;
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  const uint32_t smask = (1u << bits) - 1;
  const uint32_t n_chunks = (rv.n + BLOCK - 1) / BLOCK;
  uint32_t qn = 0;  // wave-uniform
  // the queued candidates, two per lane at a time: probe the set exactly
  auto drain = [&]() {
#ifdef FQ_TIMING
    const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
    const uint32_t qn0 = qn;
#endif
#ifndef FQ_NOFENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
    for (uint32_t b0 = 0; b0 < qn; b0 += 2 * WAVE) {
      // Two entries per lane, two adjacent slots of each in flight.  What a drain costs is its SLOWEST lane: a wave waits
      // out one L2 round trip per probe step of the longest chain among its 128 entries (at the load of 0.375 the set
      // used to have that was 6 - 8 steps = 5 us per drain, half the kernel); the set now keeps its load below 1/8.
      uint64_t fwd[2], g0[2], g1[2];
      uint32_t rr[2], sl[2];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t idx = b0 + (uint32_t)u * WAVE + lane;
        ok[u] = idx < qn;
        fwd[u] = ok[u] ? s_qk[idx] : 0ull;
        rr[u] = ok[u] ? s_qr[idx] : 0u;
        sl[u] = set_hash(fwd[u], bits);
        g0[u] = ok[u] ? g_slots[sl[u]] : RFX_EMPTY;
        g1[u] = ok[u] ? g_slots[(sl[u] + 1) & smask] : RFX_EMPTY;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!ok[u]) continue;
        bool hit;
        if (fwd[u] == RFX_EMPTY) {
          hit = has_all_ones != 0;
        } else {
          uint64_t v0 = g0[u], v1 = g1[u];
          uint32_t q = sl[u];
          while (v0 != fwd[u] && v0 != RFX_EMPTY && v1 != fwd[u] && v1 != RFX_EMPTY) {  // (rare)
            q = (q + 2) & smask;
            v0 = g_slots[q];
            v1 = g_slots[(q + 1) & smask];
          }
          hit = v0 == fwd[u] || (v0 != RFX_EMPTY && v1 == fwd[u]);
        }
        if (hit) atomicAdd(&g_hits[rr[u]], 1u);
      }
    }
#ifndef FQ_NOFENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
    qn = 0;
#ifdef FQ_TIMING
    if (lane == 0) {
      atomicAdd(&g_fq[0], __builtin_amdgcn_s_memtime() - tq0);
      atomicAdd(&g_fq[1], 1ull);
      atomicAdd(&g_fq[2], (unsigned long long)qn0);
    }
#endif
  };
  uint64_t ncv[UW > 0 ? UW : 1];
  uint32_t ngv[UW > 0 ? UW : 1];
  auto load_ahead = [&](uint32_t chunk_) {
    const uint32_t r_ = chunk_ * BLOCK + wave * WAVE + lane;
    const bool live_ = chunk_ < n_chunks && r_ < rv.n;
    const uint32_t wr_ = live_ ? r_ * (uint32_t)UW : 0u;
#pragma unroll
    for (int i = 0; i < (UW > 0 ? UW : 1); ++i) {
      ncv[i] = live_ && UW > 0 ? rv.codes[wr_ + i] : 0ull;
      ngv[i] = live_ && UW > 0 ? rv.good[wr_ + i] : 0u;
    }
  };
  if (UW > 0) load_ahead(blockIdx.x);
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r = chunk * BLOCK + wave * WAVE + lane;
    const bool live = r < rv.n;
    // (before the next chunk's loads are issued: a drain waits for everything in flight)
    if (qn >= (uint32_t)FQ_QCAP / 2) drain();
    // one code word + its good mask: V = "a fully good window ends here", bitmap bit(s) per position, queue the rest
    auto word = [&](uint64_t prev_c, uint64_t cur_c, uint32_t prev_g, uint32_t cur_g) {
      const uint64_t X = ((uint64_t)cur_g << 32) | prev_g;
      uint64_t acc = ~0ull, run = X;
      int off = 0;
      for (int bit = 0; (k >> bit) != 0; ++bit) {  // run-length doubling, see k_filter_fast
        if ((k >> bit) & 1) {
          acc &= run << off;
          off += 1 << bit;
        }
        run &= run << (1 << bit);
      }
      const uint32_t V = (uint32_t)(acc >> 32);
      const uint32_t p1 = (uint32_t)(prev_c >> 32), c0 = (uint32_t)cur_c, c1 = (uint32_t)(cur_c >> 32);
      uint32_t cand = 0;
      if (__ballot((V & 0x000000FFu) != 0)) cand = fq_group<0, TWO>(s_bm, amask, ishift, p1, c0, c1, cand); else cand >>= 8;
      if (__ballot((V & 0x0000FF00u) != 0)) cand = fq_group<1, TWO>(s_bm, amask, ishift, p1, c0, c1, cand); else cand >>= 8;
      if (__ballot((V & 0x00FF0000u) != 0)) cand = fq_group<2, TWO>(s_bm, amask, ishift, p1, c0, c1, cand); else cand >>= 8;
      if (__ballot((V & 0xFF000000u) != 0)) cand = fq_group<3, TWO>(s_bm, amask, ishift, p1, c0, c1, cand); else cand >>= 8;
      cand &= V;
#ifdef FQ_NOPUSH  // experiment: what the lookups alone cost
      if (cand == 0x12345678u && V == 0x9ABCDEF0u) s_qr[0] = cand;
      cand = 0;
#endif
      for (;;) {
        const unsigned long long act = __ballot(cand != 0);
        if (!act) break;
        const uint32_t cnt = (uint32_t)__popcll(act);
        if (qn + cnt > (uint32_t)FQ_QCAP) drain();
        if (cand) {
          const int j = __ffs(cand) - 1;
          cand &= cand - 1u;
          // window = bases j-k+1 .. j of (prev word, this word): one 128-bit shift, then the 2-bit groups reversed to
          // get the forward key (first base most significant)
          const int sh = 2 * (j - k + 1) + 64;  // 2 .. 126
          const uint64_t packed = sh >= 64 ? cur_c >> (sh - 64) : (prev_c >> sh) | (cur_c << (64 - sh));
          uint64_t y = __brevll(packed);
          y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
          const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
          s_qk[slot] = (k == 32 ? y : y >> (64 - 2 * k)) & kmask;
          s_qr[slot] = r;
        }
        qn += cnt;
      }
    };
    if (UW > 0) {
      uint64_t cv[UW > 0 ? UW : 1];
      uint32_t gv[UW > 0 ? UW : 1];
#pragma unroll
      for (int i = 0; i < UW; ++i) {  // loaded one turn ahead: the HBM latency is spent under the previous chunk
        cv[i] = ncv[i];
        gv[i] = ngv[i];
      }
      load_ahead(chunk + gridDim.x);
      // src/RUFUS.Filter.cpp:203: `i < length()-1` -- the last base is never examined.
      const uint32_t stop = live ? (last_base_skipped ? rv.ulen - 1 : rv.ulen) : 0u;
#pragma unroll
      for (int i = 0; i < UW; ++i) {
        const uint32_t lo = (uint32_t)i * 32u;
        uint32_t g = gv[i];
        if (stop < lo + 32u) g = stop > lo ? g & ((1u << (stop - lo)) - 1u) : 0u;
        gv[i] = g;
        word(i ? cv[i - 1] : 0ull, cv[i], i ? gv[i - 1] : 0u, g);
      }
    } else {
      const uint32_t wr = live ? rv_off(rv, r) : 0u;
      const uint32_t len = live ? rv_len(rv, r) : 0u;
      const uint32_t stop = last_base_skipped ? (len ? len - 1 : 0) : len;
      uint32_t nw = (stop + 31) >> 5, nw_max = nw;
#pragma unroll
      for (int o = 32; o; o >>= 1) nw_max = max(nw_max, (uint32_t)__shfl_xor((int)nw_max, o));
      uint64_t prev_c = 0;
      uint32_t prev_g = 0;
      for (uint32_t wi = 0; wi < nw_max; ++wi) {  // every lane takes part in the ballots of every trip
        const uint64_t cur_c = wi < nw ? rv.codes[wr + wi] : 0ull;
        uint32_t cur_g = wi < nw ? rv.good[wr + wi] : 0u;
        const uint32_t lo = wi << 5;
        if (stop < lo + 32u) cur_g = stop > lo ? cur_g & ((1u << (stop - lo)) - 1u) : 0u;
        word(prev_c, cur_c, prev_g, cur_g);
        prev_c = cur_c;
        prev_g = cur_g;
      }
    }
  }
  if (qn) drain();
#ifdef FQ_TIMING
  if (lane == 0) {
    atomicAdd(&g_fq[3], __builtin_amdgcn_s_memtime() - t_kernel0);
    atomicAdd(&g_fq[4], 1ull);
  }
#endif
}

// ---- pair filter (round 6; k >= 16, sets of more than 4096 keys): ONE table lookup per TWO windows ----------------------
// k_filter_q is bound twice over: by the instructions it issues (8 VALU per window: SQ_INSTS_VALU x 4 cycles = its run
// time to 1 %) and, at 60 % already, by the LDS cycles of one random bitmap read per window (two thirds of them bank
// conflicts, which random addresses have whatever the layout).  Fewer lookups, cheaper lookups:
//   * a lookup at every EVEN read position e only: the halfword (16 bits) of a 2^16-halfword table that the 8 bases ending
//     at e address, and in it the three bits that the three base pairs before those 8 bases pick (14 bases = 28 bits of
//     the window decide).  A key leaves TWO entries: its last 14 bases (the window ends at e) and the 14 bases before its
//     last one (the window ends at e + 1); a lookup that passes makes candidates of both windows -- the valid-window mask
//     and the exact probe sort them out, as before.  Twice the entries with three bits each fill a quarter of the bits:
//     1.6 % of the lookups pass on random sequence, 2.4 candidate windows per 150 bp read against 1.4 before;
//   * two lookups that lie 8 bases apart are worked on as the two halves of one register (v_pk_lshrrev_b16 shifts both
//     halfwords by their own 4-bit amounts, which are the low nibbles of the halves of a stream register read 2, 4 and 6
//     bases earlier: no instruction extracts an index): 7 funnel shifts + 4 x 10 instructions per 16 bases, 2.9 per
//     window where k_filter_q has 8, and half its LDS reads.
// Everything else -- a 150 bp read loaded whole a chunk ahead, the mask of fully good windows by run-length doubling,
// candidates ballot-compacted into per-wave LDS queues and probed 128 at a time -- is k_filter_q's.
constexpr int FP_TABLE_BYTES = 2 << 16;  // 2^16 halfwords = 2^20 bits

// what a key leaves in the table (kind 0: its window ends at the lookup position, kind 1: one base behind it): the
// halfword's index and its bits, from the same bases in the same order as fp_group below reads them off the stream
__device__ __forceinline__ void fp_entry(uint64_t key, int k, int kind, bool three, uint32_t& half, uint32_t& bitsel) {
  auto base = [&](int off) { return (uint32_t)(key >> (2 * (k - 1 - off))) & 3u; };  // (first base most significant)
  const int e = k - 1 - kind;
  half = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) half |= base(e - 7 + i) << (2 * i);
  const uint32_t i1 = base(e - 9) | (base(e - 8) << 2), i2 = base(e - 11) | (base(e - 10) << 2),
                 i3 = base(e - 13) | (base(e - 12) << 2);
  bitsel = (1u << i1) | (1u << i2) | (three ? 1u << i3 : 0u);
}

__global__ __launch_bounds__(256) void k_set_bitmap_p(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ bm,
                                                       int k, int three) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      uint32_t half, bitsel;
      fp_entry(key, k, kind, three != 0, half, bitsel);
      atomicOr(&bm[half >> 1], bitsel << ((half & 1u) * 16u));
    }
  }
}

__device__ __forceinline__ uint32_t fp_pk_lshr(uint32_t amount, uint32_t value) {  // both halves: value.h >> (amount.h & 15)
  uint32_t d;
  asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(d) : "v"(amount), "v"(value));
  return d;
}

// The 8 lookups of 16 bases: (lo, hi) = the stream dwords the 16 bases END in (hi) and the dword before (lo); lookups at the
// even positions 0, 2 .. 14 of the 16; bit m of the result: the lookup at position 2 m passed.
template <bool THREE>
__device__ __forceinline__ uint32_t fp_group(uint32_t lo, uint32_t hi) {
  typedef __attribute__((address_space(3))) const uint16_t lds_half;
  // x[j]: the 32 stream bits that begin 6 + 4 j bits into (lo, hi).  x[q + 3]: low half = the 8 bases ending at position
  // 2 q of the 16, high half = the 8 bases ending 8 bases on; x[q + 2], x[q + 1], x[q] begin 2, 4, 6 bases earlier: their
  // halves' low nibbles are the base pairs before the two 8-mers
  uint32_t x[7];
#define FP_X(j) x[j] = __builtin_amdgcn_alignbit(hi, lo, 6 + 4 * j)
  FP_X(0); FP_X(1); FP_X(2); FP_X(3); FP_X(4); FP_X(5); FP_X(6);
#undef FP_X
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // (the table starts at LDS address 0 -- the kernel checks it --: the doubled index IS the address)
    // (byte address = 2 x halfword: the shift with the half selected in the operand -- the compiler finds that form for
    // the upper half only and spends a shift and a mask on the lower)
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(a) : "v"(1u), "v"(x[q + 3]));
    const uint32_t b = (x[q + 3] >> 16) << 1;
    w[q] = (uint32_t)*(lds_half*)(uintptr_t)a | ((uint32_t)*(lds_half*)(uintptr_t)b << 16);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int q = 3; q >= 0; --q) {
    uint32_t t = fp_pk_lshr(x[q + 2], w[q]) & fp_pk_lshr(x[q + 1], w[q]);
    if (THREE) t &= fp_pk_lshr(x[q], w[q]);
    acc = (acc << 1) | (t & 0x00010001u);
  }
  return (acc & 0xFu) | ((acc >> 12) & 0xF0u);
}

__device__ __forceinline__ uint32_t fp_spread(uint32_t x) {  // bit m of 16 -> bits 2 m and 2 m + 1
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x | (x << 1);
}

// UW > 0: every read of the (compact) block has UW code words, loaded in one burst.  MASK: thresh = 1 and nobody wants the
// counts -- a hit sets the read's bit of g_mask (zeroed by the caller) and the per-read count array, its memset and the
// pass that turned it into the mask do not exist (8 of k_filter_q's 69 bytes per read).
template <int BLOCK, int UW, bool THREE, bool MASK>
__global__ __launch_bounds__(BLOCK) void k_filter_p(rfx_reads_view rv, const uint64_t* __restrict__ g_slots, int bits,
                                                     int has_all_ones, const uint32_t* __restrict__ g_bm, int k,
                                                     int last_base_skipped, uint32_t* __restrict__ g_hits,
                                                     unsigned long long* __restrict__ g_mask) {
  extern __shared__ uint32_t s_fq[];  // table | per-wave queues: FQ_QCAP keys (64-bit), then FQ_QCAP reads
  constexpr uint32_t bm_words = FP_TABLE_BYTES / 4;
  const uint32_t wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  uint64_t* s_qk = (uint64_t*)(s_fq + bm_words) + wave * FQ_QCAP;
  uint32_t* s_qr = s_fq + bm_words + (BLOCK / WAVE) * FQ_QCAP * 2 + wave * FQ_QCAP;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)s_fq != 0u) __builtin_trap();  // see fp_group
  for (uint32_t i = threadIdx.x; i < bm_words / 4; i += BLOCK) ((uint4*)s_fq)[i] = ((const uint4*)g_bm)[i];
  __syncthreads();
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  const uint32_t smask = (1u << bits) - 1;
  const uint32_t n_chunks = (rv.n + BLOCK - 1) / BLOCK;
  uint32_t qn = 0;  // wave-uniform
  auto drain = [&]() {  // the queued candidates, two per lane at a time: probe the set exactly (k_filter_q's)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t b0 = 0; b0 < qn; b0 += 2 * WAVE) {
      uint64_t fwd[2], g0[2], g1[2];
      uint32_t rr[2], sl[2];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t idx = b0 + (uint32_t)u * WAVE + lane;
        ok[u] = idx < qn;
        // (queued as cut from the stream, last base most significant: the 2-bit groups are put in key order HERE, by all
        // 64 lanes at once -- in the queueing loop, where a lane or two are live per trip, it was 14 of its 35 instructions)
        uint64_t y = __brevll(ok[u] ? s_qk[idx] : 0ull);
        y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
        fwd[u] = (k == 32 ? y : y >> (64 - 2 * k)) & kmask;
        rr[u] = ok[u] ? s_qr[idx] : 0u;
        sl[u] = set_hash(fwd[u], bits);
        g0[u] = ok[u] ? g_slots[sl[u]] : RFX_EMPTY;
        g1[u] = ok[u] ? g_slots[(sl[u] + 1) & smask] : RFX_EMPTY;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!ok[u]) continue;
        bool hit;
        if (fwd[u] == RFX_EMPTY) {
          hit = has_all_ones != 0;
        } else {
          uint64_t v0 = g0[u], v1 = g1[u];
          uint32_t q = sl[u];
          while (v0 != fwd[u] && v0 != RFX_EMPTY && v1 != fwd[u] && v1 != RFX_EMPTY) {  // (rare)
            q = (q + 2) & smask;
            v0 = g_slots[q];
            v1 = g_slots[(q + 1) & smask];
          }
          hit = v0 == fwd[u] || (v0 != RFX_EMPTY && v1 == fwd[u]);
        }
        if (hit) {
          // (the bit by a 32-bit atomic on the half of the mask word that holds it.  The 64-bit form -- atomicOr(&g_mask[r >> 6],
          // 1ull << (r & 63)) -- set bits of reads WITHOUT a hit and lost a true one now and then, differently from run to run,
          // once a set of > 10^5 keys made the queue drain all the time: found by a self-check at 300x (the filter's
          // counting mode, k_filter_q and the generic kernel agreed with each other and with this form); with 32-bit atomics
          // six runs of three variants were exact.  tests/test_gpu_parity.py::test_filter_mask_only_equals_counts_on_large_sets)
          if (MASK) atomicOr((unsigned int*)g_mask + (rr[u] >> 5), 1u << (rr[u] & 31u));
          else atomicAdd(&g_hits[rr[u]], 1u);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    qn = 0;
  };
  uint64_t ncv[UW > 0 ? UW : 1];
  uint32_t ngv[UW > 0 ? UW : 1];
  auto load_ahead = [&](uint32_t chunk_) {
    const uint32_t r_ = chunk_ * BLOCK + wave * WAVE + lane;
    const bool live_ = chunk_ < n_chunks && r_ < rv.n;
    const uint32_t wr_ = live_ ? r_ * (uint32_t)UW : 0u;
#pragma unroll
    for (int i = 0; i < (UW > 0 ? UW : 1); ++i) {
      ncv[i] = live_ && UW > 0 ? rv.codes[wr_ + i] : 0ull;
      ngv[i] = live_ && UW > 0 ? rv.good[wr_ + i] : 0u;
    }
  };
  if (UW > 0) load_ahead(blockIdx.x);
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t r = chunk * BLOCK + wave * WAVE + lane;
    const bool live = r < rv.n;
    if (qn >= (uint32_t)FQ_QCAP / 2) drain();  // (before the next chunk's loads are issued: a drain waits for everything in flight)
    auto word = [&](uint64_t prev_c, uint64_t cur_c, uint32_t prev_g, uint32_t cur_g) {
      // V = "a fully good window ends here": no bad base among the k - 1 before a position or at it.  The bad bits of
      // (previous word, this word) are smeared k - 1 places upwards in doubling steps (1, 2, 4 .. then what is left: five
      // steps for k = 25); only this word's half of the result is wanted, so a step is a funnel shift + or on it and a
      // shift-or on the lower half -- 3 instructions where the 64-bit run-length doubling of k_filter_q has 8.
      uint32_t blo = ~prev_g, bhi = ~cur_g;
      for (uint32_t reach = 0, n = (uint32_t)k - 1u; reach < n;) {
        const uint32_t st = min(reach + 1u, n - reach);  // (wave-uniform: k is; st <= 16)
        bhi |= __builtin_amdgcn_alignbit(bhi, blo, 32u - st);
        blo |= blo << st;
        reach += st;
      }
      const uint32_t V = ~bhi;
      const uint32_t p1 = (uint32_t)(prev_c >> 32), c0 = (uint32_t)cur_c, c1 = (uint32_t)(cur_c >> 32);
      // (a half without a fully good window in any lane -- the first k - 1 bases, the tail -- is not looked up)
      uint32_t L = 0;
      if (__ballot((V & 0x0000FFFFu) != 0)) L = fp_group<THREE>(p1, c0);
      if (__ballot((V & 0xFFFF0000u) != 0)) L |= fp_group<THREE>(c0, c1) << 8;
      uint32_t cand = fp_spread(L) & V;
#ifdef FP_NOPUSH  // experiment (results void): what the lookups alone cost
      if (cand == 0x12345678u && V == 0x9ABCDEF0u) s_qr[0] = cand;
      cand = 0;
#endif
#ifdef FP_NOLOOKUP  // experiment (results void): everything but the lookups
      cand = V & (c0 == 0x12345678u ? ~0u : 0u);
#endif
      for (;;) {
        const unsigned long long act = __ballot(cand != 0);
        if (!act) break;
        const uint32_t cnt = (uint32_t)__popcll(act);
        if (qn + cnt > (uint32_t)FQ_QCAP) drain();
        if (cand) {
          const int j = __ffs(cand) - 1;
          cand &= cand - 1u;
          // window = bases j-k+1 .. j of (prev word, this word): one 128-bit shift, then the 2-bit groups reversed to
          // get the forward key (first base most significant)
          const int sh = 2 * (j - k + 1) + 64;  // 2 .. 126
          const uint64_t packed = sh >= 64 ? cur_c >> (sh - 64) : (prev_c >> sh) | (cur_c << (64 - sh));
          const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
          s_qk[slot] = packed;  // (the bits above the window's 2 k go when the drain turns it into the key)
          s_qr[slot] = r;
        }
        qn += cnt;
      }
    };
    if (UW > 0) {
      uint64_t cv[UW > 0 ? UW : 1];
      uint32_t gv[UW > 0 ? UW : 1];
#pragma unroll
      for (int i = 0; i < UW; ++i) {  // loaded one turn ahead: the HBM latency is spent under the previous chunk
        cv[i] = ncv[i];
        gv[i] = ngv[i];
      }
      load_ahead(chunk + gridDim.x);
      // src/RUFUS.Filter.cpp:203: `i < length()-1` -- the last base is never examined.
      const uint32_t stop = live ? (last_base_skipped ? rv.ulen - 1 : rv.ulen) : 0u;
#pragma unroll
      for (int i = 0; i < UW; ++i) {
        const uint32_t lo = (uint32_t)i * 32u;
        uint32_t g = gv[i];
        if (stop < lo + 32u) g = stop > lo ? g & ((1u << (stop - lo)) - 1u) : 0u;
        gv[i] = g;
        word(i ? cv[i - 1] : 0ull, cv[i], i ? gv[i - 1] : 0u, g);
      }
    } else {
      const uint32_t wr = live ? rv_off(rv, r) : 0u;
      const uint32_t len = live ? rv_len(rv, r) : 0u;
      const uint32_t stop = last_base_skipped ? (len ? len - 1 : 0) : len;
      uint32_t nw = (stop + 31) >> 5, nw_max = nw;
#pragma unroll
      for (int o = 32; o; o >>= 1) nw_max = max(nw_max, (uint32_t)__shfl_xor((int)nw_max, o));
      uint64_t prev_c = 0;
      uint32_t prev_g = 0;
      for (uint32_t wi = 0; wi < nw_max; ++wi) {  // every lane takes part in the ballots of every trip
        const uint64_t cur_c = wi < nw ? rv.codes[wr + wi] : 0ull;
        uint32_t cur_g = wi < nw ? rv.good[wr + wi] : 0u;
        const uint32_t lo = wi << 5;
        if (stop < lo + 32u) cur_g = stop > lo ? cur_g & ((1u << (stop - lo)) - 1u) : 0u;
        word(prev_c, cur_c, prev_g, cur_g);
        prev_c = cur_c;
        prev_g = cur_g;
      }
    }
  }
  if (qn) drain();
}

// the number of set bits of a hit mask (k_filter_p MASK: the mask is made by the hits themselves)
__global__ __launch_bounds__(256) void k_mask_count(const unsigned long long* __restrict__ mask, uint64_t n_words,
                                                     unsigned long long* __restrict__ d_nhit) {
  unsigned long long c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
    c += (unsigned long long)__popcll(mask[i]);
#pragma unroll
  for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & (WAVE - 1)) == 0 && c) atomicAdd(d_nhit, c);
}

// per-read hit counts -> one bit per read (count >= thresh) and the number of such reads
__global__ __launch_bounds__(256) void k_hits_mask(const uint32_t* __restrict__ hits, uint32_t n, int thresh,
                                                    uint64_t* __restrict__ hitmask,
                                                    unsigned long long* __restrict__ d_nhit) {
  const uint32_t n_pad = (n + WAVE - 1) & ~(uint32_t)(WAVE - 1);
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_pad; r += (uint64_t)gridDim.x * blockDim.x) {
    const bool pass = r < n && (int)hits[r] >= thresh;
    const unsigned long long mm = __ballot(pass);
    if ((threadIdx.x & (WAVE - 1)) == 0) {
      if (hitmask) hitmask[r >> 6] = mm;
      if (mm) atomicAdd(d_nhit, (unsigned long long)__popcll(mm));
    }
  }
}

// Device-to-device copy of big blocks.  hipMemcpyAsync between a hipMalloc'ed buffer (another runtime's: torch's
// exchange buffers) and the ctx's VMM-mapped arena took 140 ms per 11 GB segment (~80 GB/s: not a blit kernel);
// this streams 16 bytes per lane.
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
namespace rfxk {

hipError_t copy_bytes(rfx_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  if (bytes < (1u << 20) || (((uintptr_t)dst | (uintptr_t)src) & 15u))
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream);
  const uint64_t n16 = bytes / 16;
  {
    rfx_span sp(c, "k_copy16");
    hipLaunchKernelGGL(k_copy16, dim3(grid_for(c, n16, 256, 16)), dim3(256), 0, c->stream, (const uint4*)src, (uint4*)dst, n16);
  }
  const size_t done = (size_t)n16 * 16;
  if (done < bytes) return hipMemcpyAsync((char*)dst + done, (const char*)src + done, bytes - done, hipMemcpyDeviceToDevice, c->stream);
  return hipSuccess;  // (a failed launch surfaces at the next synchronisation, like every other kernel here)
}

int count_reads_block() { return K2_BLOCK; }
int count_reads_grid(rfx_ctx* c, uint32_t n_reads) { return grid_for(c, (n_reads + K2_BLOCK - 1) / K2_BLOCK, 1, 3); }

void count_reads(rfx_ctx* c, const rfx_reads_view& rv, const rfx_table_view& tv, const uint64_t* lut, int k,
                 int canonical, rfx_table_stats* stats, rfx_count_ctl* ctl, uint64_t* ovf_keys, uint64_t ovf_cap,
                 uint64_t load_limit) {
  if (rv.n == 0) return;
  const int grid = count_reads_grid(c, rv.n);
  rfx_span sp(c, "k_count_reads");
  if (canonical)
    hipLaunchKernelGGL(k_count_reads<true>, dim3(grid), dim3(K2_BLOCK), 0, c->stream, rv, tv, lut, k, stats, ctl,
                       ovf_keys, ovf_cap, load_limit);
  else
    hipLaunchKernelGGL(k_count_reads<false>, dim3(grid), dim3(K2_BLOCK), 0, c->stream, rv, tv, lut, k, stats, ctl,
                       ovf_keys, ovf_cap, load_limit);
}

void count_pairs(rfx_ctx* c, const uint64_t* keys, const uint32_t* counts, uint64_t n, const rfx_table_view& tv,
                 const uint64_t* lut, rfx_table_stats* stats) {
  if (n == 0) return;
  rfx_span sp(c, "k_count_pairs");
  hipLaunchKernelGGL(k_count_pairs, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, counts, n, tv, lut,
                     stats);
}

void table_pairs(rfx_ctx* c, const rfx_table_view& tv, uint64_t* out_keys, uint32_t* out_counts,
                 unsigned long long* d_n) {
  rfx_span sp(c, "k_table_pairs");
  hipLaunchKernelGGL(k_table_pairs, dim3(grid_for(c, tv.slots, 256, 8)), dim3(256), 0, c->stream, tv, out_keys,
                     out_counts, d_n);
}

void tile_count(rfx_ctx* c, const rfx_table_view& tv, const uint64_t* lut, uint32_t halo, uint64_t lower,
                uint64_t upper, uint32_t* tile_counts, uint64_t n_tiles) {
  rfx_span sp(c, "k_tile_count");
  hipLaunchKernelGGL(k_tile_count, dim3(grid_for(c, n_tiles, 1, 8)), dim3(256), 0, c->stream, tv, lut, halo, lower,
                     upper, tile_counts, n_tiles);
}

void tile_scan(rfx_ctx* c, const uint32_t* tile_counts, uint64_t n_tiles, uint64_t* tile_off, uint32_t* d_max) {
  rfx_span sp(c, "k_tile_scan");
  hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, c->stream, tile_counts, n_tiles, tile_off, d_max);
}

void tile_emit(rfx_ctx* c, const rfx_table_view& tv, const uint64_t* lut, uint32_t halo, uint64_t lower,
               uint64_t upper, const uint64_t* tile_off, uint64_t n_tiles, uint32_t sort_cap, uint64_t* out_keys,
               uint32_t* out_counts, uint64_t* out_pos) {
  const size_t lds = (size_t)8 * 256 * 8 + (size_t)sort_cap * 20;
  if (!rfxi::lds_opt_in(c, (const void*)k_tile_emit, 160 * 1024 - 64, 0, "k_tile_emit")) return;
  rfx_span sp(c, "k_tile_emit");
  hipLaunchKernelGGL(k_tile_emit, dim3(grid_for(c, n_tiles, 1, 8)), dim3(256), lds, c->stream, tv, lut, halo, lower,
                     upper, tile_off, n_tiles, sort_cap, out_keys, out_counts, out_pos);
}

void histo(rfx_ctx* c, const uint32_t* counts, uint64_t n, unsigned long long* d_histo) {
  if (n == 0) return;
  rfx_span sp(c, "k_histo");
  hipLaunchKernelGGL(k_histo, dim3(grid_for(c, n, 256 * 16, 2)), dim3(256), 0, c->stream, counts, n, d_histo);
}

void format_records(rfx_ctx* c, const uint64_t* keys, const uint32_t* counts, uint64_t n, int key_bytes,
                    int counter_len, uint8_t* out) {
  if (n == 0) return;
  rfx_span sp(c, "k_format_records");
  hipLaunchKernelGGL(k_format_records, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, counts, n,
                     key_bytes, counter_len, out);
}

void parse_records(rfx_ctx* c, const uint8_t* in, uint64_t n, int key_bytes, int counter_len, uint64_t* keys,
                   uint32_t* counts) {
  if (n == 0) return;
  rfx_span sp(c, "k_parse_records");
  hipLaunchKernelGGL(k_parse_records, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, in, n, key_bytes,
                     counter_len, keys, counts);
}

void compute_pos(rfx_ctx* c, const uint64_t* keys, uint64_t n, const uint64_t* lut, int ntab, uint64_t* pos) {
  if (n == 0) return;
  rfx_span sp(c, "k_compute_pos");
  hipLaunchKernelGGL(k_compute_pos, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, n, lut, ntab, pos);
}

void records_verify(rfx_ctx* c, const uint64_t* keys, const uint32_t* counts, const uint64_t* pos, uint64_t n,
                    const uint64_t* lut, int ntab, uint64_t pos_mask, uint32_t min_count, uint32_t max_count,
                    unsigned long long* d_out) {
  if (n == 0) return;
  rfx_span sp(c, "k_records_verify");
  hipLaunchKernelGGL(k_records_verify, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, counts, pos, n, lut,
                     ntab, pos_mask, min_count, max_count, d_out);
}

void records_checksum(rfx_ctx* c, const uint64_t* keys, const uint32_t* counts, uint64_t n, unsigned long long* out) {
  rfx_span sp(c, "k_records_checksum");
  hipLaunchKernelGGL(k_records_checksum, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, counts, n, out);
}

void check_sorted(rfx_ctx* c, const uint64_t* keys, const uint64_t* pos, uint64_t n, unsigned int* d_bad) {
  if (n < 2) return;
  rfx_span sp(c, "k_check_sorted");
  hipLaunchKernelGGL(k_check_sorted, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, pos, n, d_bad);
}

void flag_range(rfx_ctx* c, const uint32_t* counts, uint64_t n, uint32_t lo, uint32_t hi, uint8_t* flags) {
  if (n == 0) return;
  rfx_span sp(c, "k_flag_range");
  hipLaunchKernelGGL(k_flag_range, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, counts, n, lo, hi, flags);
}

void flag_absent(rfx_ctx* c, const uint64_t* keys, const uint64_t* pos, uint64_t n, const uint64_t* bkeys,
                 const uint64_t* bpos, uint64_t nb, int lsize, uint8_t* flags) {
  if (n == 0 || nb == 0) return;
  rfx_span sp(c, "k_flag_absent");
  // candidates per tile: the control's share of a tile (nb / n records per candidate) should fill ~2/3 of the LDS range
  uint32_t tile = FA_TILE_MAX;
  if (nb > n) tile = (uint32_t)std::max<uint64_t>(64, std::min<uint64_t>(FA_TILE_MAX, (uint64_t)(FA_BCAP * 2 / 3) * n / nb / 64 * 64));
  const uint64_t n_tiles = (n + tile - 1) / tile;
  const char* tmin = getenv("RFX_K4_TILE_MIN");  // (tests lower it to put small inputs through the tiles)
  uint64_t* bounds = n >= (tmin ? strtoull(tmin, nullptr, 10) : 1ull << 20) ? (uint64_t*)rfxi::dmalloc(c, (n_tiles + 1) * 8) : nullptr;
  if (bounds) {  // big inputs: a short range of B per tile, bisected in LDS
    hipLaunchKernelGGL(k_fa_bounds, dim3(grid_for(c, n_tiles + 1, 256, 8)), dim3(256), 0, c->stream, keys, pos, n, bkeys, bpos,
                       nb, lsize, n_tiles, tile, bounds);
    hipLaunchKernelGGL(k_flag_absent_tiled, dim3((unsigned)std::min<uint64_t>(n_tiles, (uint64_t)c->n_cu * 16)), dim3(256), 0,
                       c->stream, keys, pos, n, bkeys, bpos, nb, lsize, n_tiles, tile, bounds, flags);
    rfxi::dfree(c, bounds);  // (stream-ordered)
    return;
  }
  hipLaunchKernelGGL(k_flag_absent, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, pos, n, bkeys, bpos,
                     nb, lsize, flags);
}

void compact(rfx_ctx* c, const uint8_t* flags, const uint64_t* keys, const uint32_t* counts, const uint64_t* pos,
             uint64_t n, uint64_t* out_keys, uint32_t* out_counts, uint64_t* out_pos, uint64_t* block_off,
             unsigned long long* d_total) {
  const uint64_t nblk = (n + CP_ITEMS - 1) / CP_ITEMS;
  if (nblk == 0) {
    hipMemsetAsync(d_total, 0, sizeof(unsigned long long), c->stream);
    return;
  }
  rfx_span sp(c, "k_compact");
  hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nblk), dim3(256), 0, c->stream, flags, n, block_off);
  hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, c->stream, block_off, nblk, d_total);
  hipLaunchKernelGGL(k_compact_scatter, dim3((unsigned)nblk), dim3(64), 0, c->stream, flags, keys, counts, pos, n,
                     block_off, out_keys, out_counts, out_pos);
}

// the two halves of compact(), for a caller that sizes the output by the total (one synchronisation in between)
void compact_count(rfx_ctx* c, const uint8_t* flags, uint64_t n, uint64_t* block_off, unsigned long long* d_total) {
  const uint64_t nblk = (n + CP_ITEMS - 1) / CP_ITEMS;
  if (nblk == 0) {
    hipMemsetAsync(d_total, 0, sizeof(unsigned long long), c->stream);
    return;
  }
  rfx_span sp(c, "k_compact");
  hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nblk), dim3(256), 0, c->stream, flags, n, block_off);
  hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, c->stream, block_off, nblk, d_total);
}
void compact_scatter(rfx_ctx* c, const uint8_t* flags, const uint64_t* keys, const uint32_t* counts, const uint64_t* pos,
                     uint64_t n, const uint64_t* block_off, uint64_t* out_keys, uint32_t* out_counts, uint64_t* out_pos) {
  const uint64_t nblk = (n + CP_ITEMS - 1) / CP_ITEMS;
  if (nblk == 0) return;
  rfx_span sp(c, "k_compact");
  hipLaunchKernelGGL(k_compact_scatter, dim3((unsigned)nblk), dim3(64), 0, c->stream, flags, keys, counts, pos, n,
                     block_off, out_keys, out_counts, out_pos);
}

void query(rfx_ctx* c, const uint64_t* qkeys, uint64_t nq, const uint64_t* lut, int ntab, const uint64_t* keys,
           const uint64_t* pos, const uint32_t* counts, uint64_t n, uint32_t* out) {
  if (nq == 0) return;
  rfx_span sp(c, "k_query");
  hipLaunchKernelGGL(k_query, dim3(grid_for(c, nq, 256, 8)), dim3(256), 0, c->stream, qkeys, nq, lut, ntab, keys, pos,
                     counts, n, out);
}

void set_insert(rfx_ctx* c, const uint64_t* keys, uint64_t n, uint64_t* slots, int bits) {
  if (n == 0) return;
  rfx_span sp(c, "k_set_insert");
  hipLaunchKernelGGL(k_set_insert, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, n, slots, bits);
}

void set_bitmap(rfx_ctx* c, const uint64_t* keys, uint64_t n, uint32_t* bm, int bm_bits, int bm_shift) {
  if (n == 0) return;
  rfx_span sp(c, "k_set_bitmap");
  hipLaunchKernelGGL(k_set_bitmap, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, n, bm, bm_bits,
                     bm_shift);
}

void set_bitmap_packed(rfx_ctx* c, const uint64_t* keys, uint64_t n, uint32_t* bm) {
  if (n == 0) return;
  rfx_span sp(c, "k_set_bitmap");
  hipLaunchKernelGGL(k_set_bitmap_packed, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, n, bm);
}

void filter_fast(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* slots, int bits, int has_all_ones,
                 const uint32_t* bm, int k, int thresh, int last_base_skipped, uint32_t* hits, uint64_t* hitmask,
                 unsigned long long* d_nhit) {
  if (rv.n == 0) return;
  rfx_span sp(c, "k_filter");
  const int grid = grid_for(c, (rv.n + K5_BLOCK - 1) / K5_BLOCK, 1, 8);
  hipLaunchKernelGGL(k_filter_fast, dim3(grid), dim3(K5_BLOCK), 0, c->stream, rv, slots, bits, has_all_ones, bm, k,
                     thresh, last_base_skipped, hits, hitmask, d_nhit);
}

int filter_big_words() { return FB_WORDS; }

void set_bitmap_big(rfx_ctx* c, const uint64_t* keys, uint64_t n, uint32_t* bm) {
  if (n == 0) return;
  rfx_span sp(c, "k_set_bitmap");
  hipLaunchKernelGGL(k_set_bitmap_big, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, n, bm);
}

void filter_big(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* slots, int bits, int has_all_ones,
                const uint32_t* bm, int k, int thresh, int last_base_skipped, uint32_t* hits, uint64_t* hitmask,
                unsigned long long* d_nhit) {
  if (rv.n == 0) return;
  if (!rfxi::lds_opt_in(c, (const void*)k_filter_big, FB_WORDS * 4, 1, "k_filter_big")) return;  // 128 KB of dynamic LDS
  rfx_span sp(c, "k_filter");
  const uint32_t chunks = (rv.n + FB_BLOCK - 1) / FB_BLOCK;
  const int grid = (int)std::min<uint32_t>(chunks, (uint32_t)c->n_cu);  // one resident workgroup per CU
  hipLaunchKernelGGL(k_filter_big, dim3(grid), dim3(FB_BLOCK), FB_WORDS * 4, c->stream, rv, slots, bits, has_all_ones, bm, k,
                     thresh, last_base_skipped, hits, hitmask, d_nhit);
}

static int filter_q_two(int k) {  // two bits per key in the bitmap word (RFX_FQ_ONE=1: one, for A/B runs)
  static const bool one = getenv("RFX_FQ_ONE") != nullptr;
  return k >= FQ_NB + 3 && !one;
}

int filter_q_bits(uint64_t n_keys, int k) {  // 0: the queue filter does not apply
  if (k < FQ_NB || n_keys > (1u << 18)) return 0;
  // Small sets: <= 1.5 % of the bits set, 8 .. 32 KB of LDS, many small workgroups.  Beyond 4096 keys always 2^20 bits
  // (one 1024-thread workgroup per CU = the 16 waves two 512-thread workgroups with 2^19 bits would be, with half
  // the candidates: a drain costs two dependent memory round trips however few candidates it carries).
  if (n_keys > 4096) return 20;
  int b = 16;
  while (b < 18 && (n_keys * 64 >> b) != 0) ++b;
  return b;
}

void set_bitmap_q(rfx_ctx* c, const uint64_t* keys, uint64_t n, uint32_t* bm, int bm_bits, int k) {
  if (n == 0) return;
  rfx_span sp(c, "k_set_bitmap");
  hipLaunchKernelGGL(k_set_bitmap_q, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, n, bm, bm_bits,
                     filter_q_two(k));
}

void filter_q(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* slots, int bits, int has_all_ones, const uint32_t* bm,
              int bm_bits, int k, int thresh, int last_base_skipped, uint32_t* hits /* zeroed, never null */,
              uint64_t* hitmask, unsigned long long* d_nhit) {
  if (rv.n == 0) return;
  const int block = bm_bits >= 20 ? 1024 : bm_bits == 19 ? 512 : 256;
  const size_t lds = ((size_t)(1u << (bm_bits - 5)) + (size_t)(block / WAVE) * FQ_QCAP * 3) * 4;
  const int per_cu = std::max<int>(1, std::min<int>(2048 / block, (int)((160 * 1024) / lds)));
  const uint32_t chunks = (rv.n + block - 1) / block;
  const int grid = (int)std::min<uint32_t>(chunks, (uint32_t)c->n_cu * per_cu);  // the resident workgroups
  const bool u5 = rv.ulen && rv.uwpr == 5;
  rfx_span sp(c, "k_filter");
#define RFX_FQ2(BLOCK, UW, TWO, BIT)                                                                                       \
  do {                                                                                                                     \
    if (lds > 64 * 1024 && !rfxi::lds_opt_in(c, (const void*)k_filter_q<BLOCK, UW, TWO>, lds, BIT, "k_filter_q")) return;  \
    hipLaunchKernelGGL((k_filter_q<BLOCK, UW, TWO>), dim3(grid), dim3(BLOCK), lds, c->stream, rv, slots, bits,            \
                       has_all_ones, bm, bm_bits, k, last_base_skipped, hits);                                            \
  } while (0)
#define RFX_FQ(BLOCK, UW, BIT)                 \
  do {                                         \
    if (filter_q_two(k)) RFX_FQ2(BLOCK, UW, true, BIT); \
    else RFX_FQ2(BLOCK, UW, false, BIT + 6);    \
  } while (0)
  if (block == 1024) {
    if (u5) RFX_FQ(1024, 5, 6);
    else RFX_FQ(1024, 0, 7);
  } else if (block == 512) {
    if (u5) RFX_FQ(512, 5, 8);
    else RFX_FQ(512, 0, 9);
  } else {
    if (u5) RFX_FQ(256, 5, 10);
    else RFX_FQ(256, 0, 11);
  }
#undef RFX_FQ
#undef RFX_FQ2
  hipLaunchKernelGGL(k_hits_mask, dim3(grid_for(c, rv.n, 256, 8)), dim3(256), 0, c->stream, hits, rv.n, thresh, hitmask,
                     d_nhit);
}

int filter_p_applies(uint64_t n_keys, int k) {
  if (getenv("RFX_FILTER_NO_PAIR")) return 0;  // (A/B runs and the tests' second opinion: k_filter_q)
  // Up to 81 920 set entries (a hash list of 40 960 k-mers, both orientations): beyond, the table's 2^20 bits fill up -- two entries
  // per key -- and k_filter_q, one entry per key, is the faster one (k = 25, ms per 1.3 * 10^8 reads: 120 000 entries 10.8 against
  // 17.7, 244 000: 24.7 against 60.2; below: 48 000, the W trio's, 7.4 against 6.6, and the k = 31 config's 67 526 68.7 against
  // 45.9 per step -- profiles/r06_filter_vs_hashlist.txt).
  // RFX_FILTER_PAIR_MAX_LOG2 = 13 .. 18: the tests keep the pair filter on sets where its queue drains all the time.
  uint64_t max_n = (1u << 16) + (1u << 14);
  if (const char* ev = getenv("RFX_FILTER_PAIR_MAX_LOG2")) max_n = 1ull << std::min(18, std::max(13, atoi(ev)));
  if (k < 16 || n_keys <= 4096 || n_keys > max_n) return 0;
  // two entries per key; with two bits each up to ~16 K keys fill 6 % of the table and 0.4 % of the lookups pass; beyond,
  // a third bit (a quarter of the bits set by 50 K keys: 1.6 % pass, with two bits it would be 3.3 %)
  if (const char* ev = getenv("RFX_FILTER_PAIR_BITS")) return atoi(ev) == 2 ? 1 : 2;
  return n_keys <= 16384 ? 1 : 2;
}
size_t filter_p_table_bytes() { return FP_TABLE_BYTES; }

void set_bitmap_p(rfx_ctx* c, const uint64_t* keys, uint64_t n, uint32_t* bm, int k, int three) {
  if (n == 0) return;
  rfx_span sp(c, "k_set_bitmap");
  hipLaunchKernelGGL(k_set_bitmap_p, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, keys, n, bm, k, three);
}

void filter_p(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* slots, int bits, int has_all_ones, const uint32_t* bm, int three,
              int k, int thresh, int last_base_skipped, uint32_t* hits, uint64_t* hitmask, unsigned long long* d_nhit) {
  if (rv.n == 0) return;
  constexpr int BLOCK = 1024;
  const size_t lds = (size_t)FP_TABLE_BYTES + (size_t)(BLOCK / WAVE) * FQ_QCAP * 12;
  const uint32_t chunks = (rv.n + BLOCK - 1) / BLOCK;
  const int grid = (int)std::min<uint32_t>(chunks, (uint32_t)c->n_cu);  // one resident workgroup per CU
  const bool u5 = rv.ulen && rv.uwpr == 5;
  const bool mask = hits == nullptr;
  {
    rfx_span sp(c, "k_filter");
#define RFX_FP3(UW, THREE, MASK, BIT)                                                                                       \
  do {                                                                                                                     \
    if (!rfxi::lds_opt_in(c, (const void*)k_filter_p<BLOCK, UW, THREE, MASK>, lds, BIT, "k_filter_p")) return;             \
    hipLaunchKernelGGL((k_filter_p<BLOCK, UW, THREE, MASK>), dim3(grid), dim3(BLOCK), lds, c->stream, rv, slots, bits,     \
                       has_all_ones, bm, k, last_base_skipped, hits, (unsigned long long*)hitmask);                        \
  } while (0)
#define RFX_FP2(UW, THREE, BIT)              \
  do {                                       \
    if (mask) RFX_FP3(UW, THREE, true, BIT); \
    else RFX_FP3(UW, THREE, false, BIT + 1); \
  } while (0)
    if (u5) {
      if (three) RFX_FP2(5, true, 18);
      else RFX_FP2(5, false, 20);
    } else {
      if (three) RFX_FP2(0, true, 22);
      else RFX_FP2(0, false, 24);
    }
#undef RFX_FP2
#undef RFX_FP3
  }
  if (mask)
    hipLaunchKernelGGL(k_mask_count, dim3(grid_for(c, ((uint64_t)rv.n + 63) / 64, 256, 8)), dim3(256), 0, c->stream,
                       (const unsigned long long*)hitmask, ((uint64_t)rv.n + 63) / 64, d_nhit);
  else
    hipLaunchKernelGGL(k_hits_mask, dim3(grid_for(c, rv.n, 256, 8)), dim3(256), 0, c->stream, hits, rv.n, thresh, hitmask,
                       d_nhit);
}

void filter(rfx_ctx* c, const rfx_reads_view& rv, const uint64_t* slots, int bits, int has_all_ones, const uint32_t* bm,
            int bm_bits, int bm_shift, int k, int thresh, int last_base_skipped, uint32_t* hits, uint64_t* hitmask,
            unsigned long long* d_nhit) {
  if (rv.n == 0) return;
  rfx_span sp(c, "k_filter");
  const int grid = grid_for(c, (rv.n + K5_BLOCK - 1) / K5_BLOCK, 1, 8);
  hipLaunchKernelGGL(k_filter, dim3(grid), dim3(K5_BLOCK), 0, c->stream, rv, slots, bits, has_all_ones, bm, bm_bits,
                     bm_shift, k, thresh, last_base_skipped, hits, hitmask, d_nhit);
}

}  // namespace rfxk
