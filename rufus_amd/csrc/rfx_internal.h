// Internal structures shared by the C-ABI layer (rfx_api.hip) and the kernels (rfx_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "../../include/rufus_hip.h"

#define RFX_EMPTY 0xFFFFFFFFFFFFFFFFull
#define RFX_TILE 1024u          // table slots per finish tile
#define RFX_TABLE_MARGIN 65536u // probe run-off slots behind the last home slot (no wrap-around)
#define RFX_PROBE_LIMIT 2048u   // longer probes divert the key to the overflow list and stop the launch

struct rfx_prof_span {
  std::string name;
  hipEvent_t e0, e1;
};
struct rfx_prof_acc {
  double ms = 0;
  uint64_t launches = 0;
};

// Per (k, lsize) constants: jellyfish matrix and the byte-indexed GF(2) lookup tables on the device.
struct rfx_hash_consts {
  uint64_t cols[64];
  int ntab = 0;
  uint64_t* lut = nullptr;       // M:      key -> pos
  uint64_t* lut_t = nullptr;     // T:      key -> sortable word (null if M is rank deficient / 2k > 62)
  uint64_t* lut_tinv = nullptr;  // T^-1:   sortable word -> key
};

struct rfx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t aux = nullptr;  // second stream: run maps made AHEAD (rfx_count_prefetch_maps) beside the main stream's work
  size_t budget = 0, used = 0;
  int n_cu = 256;
  bool prof = false;
  std::string prof_filter;  // ",name1,name2," -- when non-empty only these kernels are bracketed
  std::vector<rfx_prof_span> spans;
  std::vector<hipEvent_t> free_events;  // recycled: creating two events per launch costs more than recording them
  std::map<std::string, rfx_prof_acc> acc;
  std::map<void*, size_t> allocs;
  std::multimap<size_t, void*> pool;  // freed blocks kept for reuse, keyed by size (fallback allocator)
  // Arena allocator (default): one contiguous virtual range, physical memory mapped behind a high-water mark
  // in big granules and never given back before rfx_close, a first-fit free list over it.  At WGS scale the
  // library runs at ~90 % of the HBM with 2-50 GB blocks coming and going; hipMalloc/hipFree per block (and
  // the frees of a size-keyed cache under pressure) cost seconds per trio and fragment.  Everything on a ctx
  // runs on one stream, so a freed range can be handed out again at once.
  char* arena = nullptr;
  size_t arena_reserved = 0, arena_mapped = 0, arena_gran = 0;
  std::vector<void*> arena_handles;       // hipMemGenericAllocationHandle_t of the mapped granules
  std::map<size_t, size_t> arena_free;    // offset -> length of free ranges inside [0, arena_mapped)
  bool arena_off = false;                 // VMM unavailable (or RFX_NO_ARENA): size-keyed cache of hipMalloc blocks
  size_t peak_used = 0;
  std::vector<int> peer_devices;  // rfx_ctx_allow_peers: devices that may read this ctx's memory directly (xGMI)
  // kernels that need more than 64 KB of dynamic LDS opt in once per ctx (= per device: the attribute belongs to the
  // device's copy of the function); a refused opt-in is remembered and fails the next synchronisation
  uint32_t lds_opt_in = 0;
  hipError_t launch_error = hipSuccess;
  std::vector<struct rfx_table*> pend_tables;  // tables with unread MSP capacity flags
  double msp_surv_frac[2] = {0, 0};            // survivors / instances seen by the last MSP emit ([lower >= 2])
  // ... and by the last emit of a table of (about) the same number of k-mer instances: a driver that counts samples of
  // different depth in turn (tumor 60x / normal 30x: BASELINE configs[4]) finds each sample's own ratio, not its
  // predecessor's -- the normal's store was sized by the tumor's ratio, half its own, overflowed, and the whole leaf
  // phase of every pass ran twice (102 instead of 68 leaf launches per sample chain, rounds 2 - 5).
  // key: instances >> 24 | lower class << 62
  std::map<uint64_t, double> msp_surv_by_size;
  // pinned host scratch: small read-backs and uploads go through it (pageable copies cost a
  // staging round trip each); a bump allocator that is reset at every stream synchronisation
  char* pin = nullptr;
  size_t pin_cap = 0, pin_used = 0;
  // page-locked ring of rfx_records_load_fd, kept for the next load (pinning 280 MB and letting it go again costs
  // 0.2 s, as much as reading a 3.7 GB database through it)
  uint8_t* load_pin[3] = {nullptr, nullptr, nullptr};
  size_t load_pin_bytes = 0;
  struct pin_read { void* dst; size_t off, n; };
  std::vector<pin_read> pin_reads;
  std::map<std::pair<int, uint64_t>, rfx_hash_consts> consts;  // (k*64+lsize, matrix digest) -> device tables
  std::map<int, std::vector<uint64_t>> std_cols;                // k*64+lsize -> jellyfish's own matrix
};

// Device-side statistics of a count table.
struct rfx_table_stats {
  unsigned long long distinct;  // occupied slots
  unsigned int max_disp;        // largest (slot - home) of any occupied slot
  unsigned int overflow;        // a probe ran past the margin: the table content is NOT exact
};

// Launch control of the read-count kernel: chunks of reads are handed out by ticket so a launch can
// stop early (table getting full) and be resumed exactly after the host has grown the table.
struct rfx_count_ctl {
  unsigned int ticket;          // next chunk to hand out; chunks below it are fully processed
  unsigned int stop;            // set by the first block that sees the load limit or diverts a key
  unsigned int lost;            // overflow list itself overflowed: counts are NOT exact (fatal)
  unsigned int pad;
  unsigned long long ovf_n;     // keys diverted to the overflow list (each stands for one instance)
};

// What kernels see of a table.  home(pos) = pos >> rshift (or << lshift): monotone in pos, so a
// linear-probed slot array is "almost" in (pos,key) order and finish only sorts inside tiles.
struct rfx_table_view {
  uint64_t* keys;
  uint32_t* counts;
  uint64_t slots;  // cap + margin
  uint64_t cap;
  int rshift, lshift;
  uint64_t pos_lo, pos_hi;
  int ntab;    // byte-indexed GF(2) lookup tables in use (ceil(2k/8))
  int kshift;  // 2k - lshift: top key bits order equal-pos entries when the table is finer than pos
};

// Geometry of the sortable word w = T * key of the P2L count path (rfx_p2l.hip).
struct rfx_ord_cfg {
  int c_bits;     // 2k: width of w
  int sel_bits;   // 2k - lsize: pos = w >> sel_bits
  int bin_shift;  // bin = w >> bin_shift  (= c_bits - log2(number of bins))
};

struct rfx_segment {  // the k-mer instances of one rfx_count_add call, grouped by bin
  uint64_t* inst;       // P2L: sortable words; MSP: super-k-mer records
  uint64_t n;           // entries of inst
  uint64_t* bin_start;  // device, P+1 entries
  uint64_t kmers;       // upper bound of the k-mer instances represented (P2L: n)
  uint32_t bins;        // MSP: bins of THIS segment (segments are brought to a common count before the leaf)
  uint32_t* ext = nullptr;  // MSP, k = 26 .. 31: the 32-bit plane of the records (rfx_devutil.h)
  bool borrowed = false;    // inst / ext belong to the caller (rfx_count_adopt_records_dev): never freed here
};

struct rfx_reads;
// Run maps (rfx_msp.hip): what the ONE hashing launch over a big read block leaves; every shard pass cuts its records
// from reads + map.  A store is shared by the tables of a sample's shard passes (rfx_count_set_runmaps) or made by a table
// for the passes it runs itself (rfx_count_set_passes).
struct rfx_runmap_entry {
  void* map = nullptr;       // device: 32 B per read
  uint32_t* ovf = nullptr;   // device: [0] count, [1 ..] the reads without a map
  uint32_t n_ovf = 0;
  uint32_t n_reads = 0;
  const uint64_t* codes = nullptr;  // (of the block the map was made from: a block freed and another in its place is not it)
  uint64_t gen = 0;                 // rfx_reads::gen of that block: the arena hands a freed block's addresses out again
  int k = 0, canonical = 0;
  size_t bytes = 0, map_bytes = 0, ovf_bytes = 0;
};
struct runmap_ahead;  // (rfx_api.hip) the maps whose hashing launches run on the ctx's second stream
struct rfx_runmaps {
  rfx_ctx* ctx = nullptr;
  runmap_ahead* ahead = nullptr;
  uint64_t budget = 0;  // bytes of maps the store may hold (0: no limit)
  uint64_t bytes = 0;
  uint64_t pending_bytes = 0;  // maps whose hashing launch is queued (rfx_count_prepare_maps): they count against the budget
  std::map<const rfx_reads*, rfx_runmap_entry> m;
  // pooled store (rfx_runmaps_create_pooled): ONE device allocation made with the store, the maps are cut out of it
  // (first fit) -- at 90 % of the HBM gigabyte-sized maps that come and go between the transients of a pass would
  // otherwise leave the arena in pieces
  char* pool = nullptr;
  std::map<size_t, size_t> pool_free;  // offset -> length
};
struct rfx_pending_add {  // an MSP partition whose capacity flag has not been read back yet
  const rfx_reads* r;    // needed for the exact redo, so rfx_reads_free settles the add first
  uint32_t* cur;         // device: coarse cursors, flag at cur[ncur]
  size_t seg;            // index into rfx_table::segs
  size_t ncur;
};

struct rfx_reads_view {
  const uint64_t* codes;
  const uint32_t* acgt;
  const uint32_t* good;
  const uint32_t* word_off;
  const uint32_t* len;
  uint32_t n;
  // Compact blocks (every read `ulen` bases long, rfx_reads::ulen): no word_off / len arrays -- read r starts at word
  // r * uwpr -- and the ACGT mask is kept only for the reads that have a non-ACGT base: bit r % 64 of nbits[r / 64]
  // says so, their uwpr mask words follow each other in `acgt` in read order, nrank[r / 64] = index of the first
  // flagged read of the group of 64 among the flagged.  (A 150 bp read: 40 B instead of 68; `good` stays dense.)
  uint32_t ulen, uwpr;
  const uint64_t* nbits;
  const uint32_t* nrank;
  // k_msp_part1 only: when set, the launch covers reads idx[0 .. n-1] of the block instead of reads 0 .. n-1
  const uint32_t* idx = nullptr;
};
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t rv_len(const rfx_reads_view& rv, uint32_t r) { return rv.ulen ? rv.ulen : rv.len[r]; }
__device__ __forceinline__ uint32_t rv_off(const rfx_reads_view& rv, uint32_t r) { return rv.ulen ? r * rv.uwpr : rv.word_off[r]; }
// the ACGT mask words of read r (starting at word `off`), or nullptr: every base of the read is A, C, G or T
__device__ __forceinline__ const uint32_t* rv_acgt(const rfx_reads_view& rv, uint32_t r, uint32_t off) {
  if (!rv.nbits) return rv.acgt + off;
  const uint64_t bits = rv.nbits[r >> 6];
  if (!((bits >> (r & 63u)) & 1ull)) return nullptr;
  return rv.acgt + (size_t)(rv.nrank[r >> 6] + (uint32_t)__popcll(bits & ((1ull << (r & 63u)) - 1ull))) * rv.uwpr;
}
#endif

inline uint64_t rfx_next_reads_gen() {  // process-wide, never 0
  static std::atomic<uint64_t> next{1};
  return next.fetch_add(1, std::memory_order_relaxed);
}
struct rfx_reads {
  rfx_ctx* ctx;
  // Unique per block for the life of the process: what is keyed by the block's ADDRESS (run maps) checks it, because a
  // freed block is often followed by one of the same read count at the same host and device addresses.
  uint64_t gen = rfx_next_reads_gen();
  uint32_t n;
  uint64_t n_words, n_bases;
  uint32_t max_len;
  uint64_t* codes;
  uint32_t *acgt, *good, *word_off, *len;
  // compact form (see rfx_reads_view): ulen > 0, word_off == len == nullptr, acgt = the masks of the n_exc flagged reads
  uint32_t ulen, uwpr;
  uint64_t* nbits;
  uint32_t* nrank;
  uint64_t n_exc;
  rfx_reads_view view() const { return rfx_reads_view{codes, acgt, good, word_off, len, n, ulen, uwpr, nbits, nrank}; }
  uint32_t short_cnt[32];  // reads of length 0..31: they have no window for k > length, see windows_of()
  // exact number of length-k windows of the block: sum over reads of max(0, len - k + 1)
  uint64_t windows_of(int k) const {
    uint64_t bases = n_bases, reads = n;
    for (int l = 0; l < k - 1 && l < 32; ++l) {  // a read shorter than k-1 must not subtract k-1
      bases -= (uint64_t)l * short_cnt[l];
      reads -= short_cnt[l];
    }
    return bases - (uint64_t)(k - 1) * reads;
  }
};

struct rfx_table {
  rfx_ctx* ctx;
  int k, canonical, lsize;
  uint64_t cap;  // power of two
  int tbits;
  uint64_t pos_lo, pos_hi;
  uint64_t* keys;
  uint32_t* counts;
  uint64_t* lut;  // device: ntab x 256 uint64
  int ntab;
  rfx_table_stats* d_stats;
  rfx_count_ctl* d_ctl;
  uint64_t* ovf_keys;
  uint64_t ovf_cap;
  uint64_t cols[64];
  // P2L path (rfx_p2l.hip): instances partitioned by bin, counted in LDS at finish
  int mode;           // 0 auto, 1 global table only, 2 P2L only
  int table_active;   // the global table holds data
  uint32_t p2l_bins;  // 0 until the first P2L / MSP add
  int seg_kind;       // what the segments hold: 0 nothing yet, RFX_COUNT_P2L words, RFX_COUNT_MSP records
  std::vector<rfx_pending_add>* pend;
  int shard, n_shards;  // n_shards > 1: keep only the minimizer bins of this shard (rfx_count_set_shard)
  struct rfx_peers* peers;  // rfx_count_set_peers: table `peer_index` of a group of tables on several devices
  int peer_index;
  int passes;           // -1: adds count at once; >= 0: adds are deferred, finish runs that many shard passes (0 = plan)
  std::vector<const rfx_reads*>* deferred;  // read blocks of the deferred adds (not owned)
  int pend_error;     // a deferred redo failed: the table cannot be finished
  std::vector<rfx_segment>* segs;
  // rfx_count_set_early: while this table (shard s of S) partitions a big block it also partitions the block's records of
  // shard s + 1 -- ONE k_msp_part1 launch for both -- into segments kept here until a table of shard s + 1 adopts them
  int early_on;
  std::vector<rfx_segment>* early;
  rfx_runmaps* runmaps;      // rfx_count_set_runmaps (not owned), or the table's own (deferred adds / passes)
  int runmaps_owned;
  uint64_t replayed;         // big blocks added from their run map (statistics: rfx_count_replayed)
  uint64_t* lut_t;     // device LUT of T (key -> sortable word), null when M is rank deficient
  uint64_t* lut_tinv;  // device LUT of T^-1
};

struct rfx_records {
  rfx_ctx* ctx;
  int k, lsize, ntab;
  uint64_t n;
  uint64_t* keys;
  uint32_t* counts;
  uint64_t* pos;
  uint64_t* lut;
  uint64_t cols[64];
};

struct rfx_set {
  rfx_ctx* ctx;
  int k, bits;
  uint64_t n, cap;
  uint64_t* slots;
  int has_all_ones;  // K = 32 poly-T collides with the empty sentinel
  uint32_t* bitmap;  // 2^bm_bits bits indexed by (fwd >> bm_shift): pre-filter of the probe
  int bm_bits, bm_shift;
  uint32_t* bitmap2;  // k >= 16, <= 4096 keys: the two packed-order bitmaps of k_filter_fast (else null)
  uint32_t* bitmap3;  // k >= 20, 4096 < keys <= 2^17: the 2^20-bit packed-order bitmap of k_filter_big (else null)
  uint32_t* bitmap4;  // k >= 10, keys <= 2^18: the queue filter's bitmap of 2^bm4_bits bits (else null)
  int bm4_bits;
  uint32_t* bitmap5;  // k >= 16, 4096 < keys <= 2^18: the pair filter's table of 2^16 halfwords (else null)
  int bm5_three;      // three bits per entry (two: sets small enough that two decide as well)
};

// ---- kernel launchers (rfx_kernels.hip) -------------------------------------------------------
namespace rfxk {
int count_reads_grid(rfx_ctx*, uint32_t n_reads);
int count_reads_block();
void count_reads(rfx_ctx*, const rfx_reads_view&, const rfx_table_view&, const uint64_t* lut, int k, int canonical,
                 rfx_table_stats* stats, rfx_count_ctl* ctl, uint64_t* ovf_keys, uint64_t ovf_cap,
                 uint64_t load_limit);
// counts == nullptr: every key stands for one instance (overflow list re-insert)
void count_pairs(rfx_ctx*, const uint64_t* keys, const uint32_t* counts, uint64_t n, const rfx_table_view&,
                 const uint64_t* lut, rfx_table_stats* stats);
void table_pairs(rfx_ctx*, const rfx_table_view&, uint64_t* out_keys, uint32_t* out_counts,
                 unsigned long long* d_n);  // every occupied slot, unordered
void tile_count(rfx_ctx*, const rfx_table_view&, const uint64_t* lut, uint32_t halo, uint64_t lower, uint64_t upper,
                uint32_t* tile_counts, uint64_t n_tiles);
void tile_scan(rfx_ctx*, const uint32_t* tile_counts, uint64_t n_tiles, uint64_t* tile_off /* n_tiles+1 */,
               uint32_t* d_max);
void tile_emit(rfx_ctx*, const rfx_table_view&, const uint64_t* lut, uint32_t halo, uint64_t lower, uint64_t upper,
               const uint64_t* tile_off, uint64_t n_tiles, uint32_t sort_cap, uint64_t* out_keys, uint32_t* out_counts,
               uint64_t* out_pos);
void histo(rfx_ctx*, const uint32_t* counts, uint64_t n, unsigned long long* d_histo);
void format_records(rfx_ctx*, const uint64_t* keys, const uint32_t* counts, uint64_t n, int key_bytes, int counter_len,
                    uint8_t* out);
hipError_t copy_bytes(rfx_ctx* c, void* dst, const void* src, size_t bytes);  // big device-to-device copies
void parse_records(rfx_ctx*, const uint8_t* in, uint64_t n, int key_bytes, int counter_len, uint64_t* keys,
                   uint32_t* counts);
void compute_pos(rfx_ctx*, const uint64_t* keys, uint64_t n, const uint64_t* lut, int ntab, uint64_t* pos);
void check_sorted(rfx_ctx*, const uint64_t* keys, const uint64_t* pos, uint64_t n, unsigned int* d_bad);
void records_verify(rfx_ctx*, const uint64_t* keys, const uint32_t* counts, const uint64_t* pos, uint64_t n,
                    const uint64_t* lut, int ntab, uint64_t pos_mask, uint32_t min_count, uint32_t max_count,
                    unsigned long long* d_out /* 4 counters, see k_records_verify */);
void records_checksum(rfx_ctx*, const uint64_t* keys, const uint32_t* counts, uint64_t n, unsigned long long* d_out);
void flag_range(rfx_ctx*, const uint32_t* counts, uint64_t n, uint32_t lo, uint32_t hi, uint8_t* flags);
void flag_absent(rfx_ctx*, const uint64_t* keys, const uint64_t* pos, uint64_t n, const uint64_t* bkeys,
                 const uint64_t* bpos, uint64_t nb, int lsize, uint8_t* flags);
void compact(rfx_ctx*, const uint8_t* flags, const uint64_t* keys, const uint32_t* counts, const uint64_t* pos,
             uint64_t n, uint64_t* out_keys, uint32_t* out_counts, uint64_t* out_pos, uint64_t* block_off,
             unsigned long long* d_total);
void compact_count(rfx_ctx*, const uint8_t* flags, uint64_t n, uint64_t* block_off, unsigned long long* d_total);
void compact_scatter(rfx_ctx*, const uint8_t* flags, const uint64_t* keys, const uint32_t* counts, const uint64_t* pos,
                     uint64_t n, const uint64_t* block_off, uint64_t* out_keys, uint32_t* out_counts, uint64_t* out_pos);
void query(rfx_ctx*, const uint64_t* qkeys, uint64_t nq, const uint64_t* lut, int ntab, const uint64_t* keys,
           const uint64_t* pos, const uint32_t* counts, uint64_t n, uint32_t* out);
void set_insert(rfx_ctx*, const uint64_t* keys, uint64_t n, uint64_t* slots, int bits);
void set_bitmap(rfx_ctx*, const uint64_t* keys, uint64_t n, uint32_t* bm, int bm_bits, int bm_shift);
// k >= 16 and a small set: two 2^16-bit pre-filter bitmaps (last 8 bases, the 8 before) in packed order
void set_bitmap_packed(rfx_ctx*, const uint64_t* keys, uint64_t n, uint32_t* bm /* 4096 words */);
void filter_fast(rfx_ctx*, const rfx_reads_view&, const uint64_t* slots, int bits, int has_all_ones, const uint32_t* bm,
                 int k, int thresh, int last_base_skipped, uint32_t* hits, uint64_t* hitmask, unsigned long long* d_nhit);
// k >= 20, 4096 < keys <= 2^17: one 2^20-bit bitmap over a window's last 10 bases, LDS resident
int filter_big_words();
void set_bitmap_big(rfx_ctx*, const uint64_t* keys, uint64_t n, uint32_t* bm /* filter_big_words() */);
void filter_big(rfx_ctx*, const rfx_reads_view&, const uint64_t* slots, int bits, int has_all_ones, const uint32_t* bm,
                int k, int thresh, int last_base_skipped, uint32_t* hits, uint64_t* hitmask, unsigned long long* d_nhit);
// k >= 10, <= 2^18 keys: bitmap of 2^16 .. 2^20 bits over a window's last 10 bases (LDS resident), candidates queued per
// wave and probed by full waves (filter_q_bits: the bitmap's size for a set, 0 = not applicable)
int filter_q_bits(uint64_t n_keys, int k);
void set_bitmap_q(rfx_ctx*, const uint64_t* keys, uint64_t n, uint32_t* bm /* 2^(bm_bits-5) words */, int bm_bits, int k);
void filter_q(rfx_ctx*, const rfx_reads_view&, const uint64_t* slots, int bits, int has_all_ones, const uint32_t* bm,
              int bm_bits, int k, int thresh, int last_base_skipped, uint32_t* hits, uint64_t* hitmask,
              unsigned long long* d_nhit);
// k >= 16, 4096 < keys <= 2^18: one lookup per two windows in a table of 2^16 halfwords, two lookups per packed
// instruction (k_filter_p).  filter_p_applies: 0 = not for this set, 1 = two bits per entry, 2 = three.
// hits == nullptr: thresh must be 1 -- the hits set the bits of `hitmask` (zeroed by the caller) themselves.
int filter_p_applies(uint64_t n_keys, int k);
size_t filter_p_table_bytes();
void set_bitmap_p(rfx_ctx*, const uint64_t* keys, uint64_t n, uint32_t* bm /* filter_p_table_bytes(), zeroed */, int k, int three);
void filter_p(rfx_ctx*, const rfx_reads_view&, const uint64_t* slots, int bits, int has_all_ones, const uint32_t* bm, int three,
              int k, int thresh, int last_base_skipped, uint32_t* hits /* zeroed, or null */, uint64_t* hitmask,
              unsigned long long* d_nhit);
void filter(rfx_ctx*, const rfx_reads_view&, const uint64_t* slots, int bits, int has_all_ones, const uint32_t* bm,
            int bm_bits, int bm_shift, int k, int thresh, int last_base_skipped, uint32_t* hits, uint64_t* hitmask,
            unsigned long long* d_nhit);
void overlap_score(rfx_ctx*, const char* d_a, int alen, const char* d_bcat, const uint32_t* d_boff, int nb, int max_blen,
                   float min_pct, int min_ovl, int strict3, int local_init, int* d_out /* nb x 5 */);
void overlap_pool(rfx_ctx*, const char* arena, const uint64_t* off, const int* len, const char* a_explicit,
                  int a_explicit_len, int query, const int* cand, int nb, int strand_lo, int strand_hi, size_t lds,
                  float min_pct, int min_ovl, int strict3, int local_init, int* d_out);
void annotate(rfx_ctx*, const rfx_reads_view&, const uint64_t* slots, int bits, int has_all_ones, int k,
              const uint64_t* base_off, uint32_t* cov);
int p2l_grid(rfx_ctx*, uint32_t n_reads);
void bin_count(rfx_ctx*, const rfx_reads_view&, const uint64_t* lut, int ntab, int k, int canonical,
               const rfx_ord_cfg&, uint32_t P, uint64_t pos_lo, uint64_t pos_hi, int grid, uint32_t* cnt);
void bin_offsets(rfx_ctx*, uint32_t* cnt, uint32_t G, uint32_t P, uint32_t* gsum /* 8*P */,
                 uint64_t* bin_start /* P+1 */);
// exclusive bin starts from the per-block histogram rows (no per-block offsets)
void bin_totals(rfx_ctx*, const uint32_t* cnt, uint32_t G, uint32_t P, uint64_t* bin_start /* P+1 */);
void bin_scatter(rfx_ctx*, const rfx_reads_view&, const uint64_t* lut, int ntab, int k, int canonical,
                 const rfx_ord_cfg&, uint32_t P, uint64_t pos_lo, uint64_t pos_hi, int grid, const uint32_t* rel,
                 const uint64_t* bin_start, uint64_t* inst);
int p1_bins();
int p1_cur_stride();
void coarse_counts(rfx_ctx*, const uint32_t* cnt, uint32_t G, uint32_t P, uint32_t P2, uint32_t* cnt1);
void part1(rfx_ctx*, const rfx_reads_view&, const uint64_t* lut, int ntab, int k, int canonical, const rfx_ord_cfg&,
           uint32_t P2, uint64_t pos_lo, uint64_t pos_hi, int grid, const uint32_t* rel1, const uint64_t* fine_start,
           uint64_t* buf_a);
void part1_fused(rfx_ctx*, const rfx_reads_view&, const uint64_t* lut, int ntab, int k, int canonical,
                 const rfx_ord_cfg&, uint32_t P2, uint64_t pos_lo, uint64_t pos_hi, int grid, uint64_t* buf_a,
                 uint32_t* coarse_cur, uint32_t cap_a, uint32_t* cnt_rows, unsigned int* flag);
// coarse_cur != null: coarse bins are fixed-capacity (cap_a) with their fill in coarse_cur (padded cursors);
// pay_a != null: 32-bit payload per word moves along.  sub-bin of a word = (w >> shift2) & (P2 - 1).
void part2(rfx_ctx*, const uint64_t* buf_a, uint64_t* buf_b, const uint64_t* fine_start, uint32_t* fine_cur,
           uint32_t P2, int shift2, const uint32_t* coarse_cur, uint32_t cap_a, const uint32_t* pay_a,
           uint32_t* pay_b, uint64_t cap_b /* entries buf_b can hold */, const char* span,
           const uint64_t* coarse_start = nullptr /* n_coarse+1 explicit coarse extents */, uint32_t n_coarse = 0,
           uint64_t n_hint = 0 /* expected entries: sizes the grid of a small run */,
           int rec_mode = 0 /* 0: sub-bin = bits of the word; 1 / 2: MSP record (canonical / not), sub-bin = bits
                               of its minimizer bin hash (shift2 then counts from bit 0 of that 32-bit hash) */,
           int k = 0, uint64_t fine_base = 0 /* buf_b holds the entries from fine_start value fine_base on */);
// the same for coarse bins that lie in slices of nseg arrays (device arrays of nseg pointers; nseg <= part2_max_segs)
constexpr int part2_max_segs = 64;
void part2_multi(rfx_ctx*, const uint64_t* const* seg_a, const uint64_t* const* seg_cs, const uint32_t* const* seg_pay,
                 int nseg, uint32_t cs_off, uint32_t n_coarse, uint64_t n_hint, uint64_t* buf_b, const uint64_t* fine_start,
                 uint32_t* fine_cur, uint32_t P2, int shift2, uint32_t* pay_b, int rec_mode, int k, const char* span);
// sizes of the P2 sub-bins of every parent bin: fine_tot[parent * P2 + sub] += ...
void bin_hist(rfx_ctx*, const uint64_t* src, const uint64_t* parent_start, uint32_t n_parents, uint64_t n_hint,
              uint32_t P2, int shift2, int rec_mode, int k, uint64_t* fine_tot,
              const uint32_t* ext = nullptr /* the records' planes: k <= 25, they carry the bin-hash bits (rfx_devutil.h msp_stamp) */);
// the same in ONE launch over the slices of nseg arrays (device arrays of nseg pointers; parent b = bin cs_off + b of each)
void bin_hist_multi(rfx_ctx*, const uint64_t* const* seg_src, const uint64_t* const* seg_ps, const uint32_t* const* seg_ext,
                    int nseg, uint32_t cs_off, uint32_t n_parents, uint64_t n_hint, uint32_t P2, int shift2, int rec_mode, int k,
                    uint64_t* fine_tot);
// MSP path (rfx_msp.hip)
int msp_k_ok(int k);
int msp_part1_block();  // threads = reads per chunk of k_msp_part1
int msp_nmax_of(int k);   // k-mers a record holds at most (rfx_devutil.h msp_nmax)
int msp_window(int k);  // m-mers per k-mer (rfx_devutil.h msp_wl)
int msp_wide(int k);  // 1 (round 4: every record is a 64-bit word + a 32-bit plane, rfx_devutil.h)
// rec_a: the coarse bins, 12 bytes per slot (word + plane side by side, rfx_devutil.h msp_rec12): what part2 / surv_hist
// take as `buf_a` when rec_mode != 0 and the coarse bins are fixed-capacity (coarse_cur != null)
void msp_part1(rfx_ctx*, const rfx_reads_view&, int k, int canonical, int bin_bits, uint32_t bin_lo, uint32_t bin_hi,
               int hmode, int grid, void* rec_a, uint32_t* coarse_cur, uint32_t cap_a, uint32_t* cnt_rows,
               unsigned int* flag,
               int slab_log2 = 4 /* hmode 0 / 3: a workgroup fills slabs of 2^slab_log2 records per coarse bin; cap_a
                                    must leave room for msp_part1_slack(grid, slab_log2) unused slots per bin */,
               void* map_out = nullptr /* hmode 4 (no records, bins ignored): the block's run map, 32 B per read (rfx_msp.hip) */,
               uint32_t* map_ovf = nullptr /* [0]: reads without a map (zeroed by the caller), [1 ..]: their indices */,
               uint32_t map_ovf_cap = 0);
// a shard pass of a block from its reads + run map: what msp_part1(hmode 3) leaves, without hashing (reads <= 160 bases,
// bin_hi - bin_lo <= 16384); msp_replay_grid: its workgroups (two per CU)
int msp_replay_grid(rfx_ctx*, uint32_t n_reads);
int msp_map_grid(rfx_ctx*, uint32_t n_reads);
void msp_replay(rfx_ctx*, const rfx_reads_view&, const void* map, int k, int canonical, int bin_bits, uint32_t bin_lo,
                uint32_t bin_hi, int grid, void* rec_a, uint32_t* coarse_cur, uint32_t cap_a, uint32_t* cnt_rows,
                unsigned int* flag, int slab_log2);
inline uint64_t msp_part1_slack(int grid, int slab_log2) { return (uint64_t)grid * 3u << slab_log2; }
// The staging pool of the leaf's launches (rfx_msp.hip k_msp_leaf / k_surv_place): n_chunks chunks of `chunk` (key, count)
// pairs, fill[n_chunks] (zeroed once: k_surv_place puts every fill back to 0), more[launches] (zeroed once: chunks handed
// out beyond the first `grid` of launch i).
struct msp_stage {
  uint64_t* keys = nullptr;
  uint32_t* counts = nullptr;
  uint32_t* fill = nullptr;
  uint32_t* more = nullptr;
  uint32_t chunk = 0, n_chunks = 0;
};
void msp_leaf(rfx_ctx*, const uint64_t* const* seg_inst, const uint64_t* const* seg_bs, int nseg,
              const uint64_t* inst0, const uint64_t* bs0, uint32_t P, int k, int canonical, const uint64_t* lut,
              int ntab, int sel_bits, int shift1, uint64_t pos_lo, uint64_t pos_hi, uint64_t lower, uint64_t upper,
              uint64_t* out_w, uint32_t* out_c, uint32_t* cur, uint32_t cap, unsigned int* flag, unsigned int* err,
              unsigned int* stage_short /* chunks the pool came short by (max over the launches) */,
              int geo /* 0: 1024 threads + 8192 slots, 1: 768 + 4096 (two per CU) */,
              const uint32_t* const* seg_ext, const uint32_t* ext0 /* the records' planes */, const msp_stage& st,
              uint32_t launch /* index into st.more */, uint32_t grid /* from msp_leaf_plan */);
void msp_leaf_plan(rfx_ctx*, uint32_t P, int geo, uint64_t n_records, uint64_t est_survivors, uint32_t extra, uint32_t* grid,
                   uint32_t* chunk, uint32_t* n_chunks);
void surv_hist(rfx_ctx*, const uint64_t* buf_a, const uint32_t* coarse_cur, uint32_t cap_a, uint32_t P2, int shift2,
               uint64_t* fine_tot, int rec_mode = 0 /* 1 / 2: super-k-mer records, see part2 */, int k = 0,
               uint64_t n_hint = 0);
void flag_if_gt(rfx_ctx*, const uint64_t* d_value, uint64_t limit, unsigned int* d_flag);
// count-of-counts over the filled part of fixed-capacity coarse bins
void histo_bins(rfx_ctx*, const uint32_t* counts, const uint32_t* coarse_cur, uint32_t cap, unsigned long long* d_histo);
void surv_sort(rfx_ctx*, const uint64_t* bw, const uint32_t* bc, const uint64_t* bs, uint32_t P, int bin_shift,
               const uint64_t* lut_inv, int ntab, int sel_bits, uint64_t* out_keys, uint32_t* out_counts,
               uint64_t* out_pos);
void tmp_start(rfx_ctx*, const uint64_t* const* seg_bs, int nseg, uint32_t P, uint64_t* out /* P+1 */);
void leaf(rfx_ctx*, const uint64_t* const* seg_inst, const uint64_t* const* seg_bs, int nseg, const uint64_t* inst0,
          const uint64_t* bs0, uint32_t P, const rfx_ord_cfg&, uint64_t lower, uint64_t upper,
          const uint64_t* tmp_start, uint64_t* tmp_w, uint32_t* tmp_counts, uint64_t* n_surv, unsigned int* err);
void split_bins(rfx_ctx*, uint64_t* bs, uint32_t P, uint64_t n_own, uint64_t* bs2);
void scan_tail(rfx_ctx*, uint64_t* v, uint64_t n);  // exclusive scan in place, v[n] = total
void leaf_compact(rfx_ctx*, const uint64_t* tmp_w, const uint32_t* tmp_counts, const uint64_t* tmp_start,
                  const uint64_t* out_off, uint32_t P, const uint64_t* lut_inv, int ntab, int sel_bits,
                  uint64_t* out_keys, uint32_t* out_counts, uint64_t* out_pos);
}  // namespace rfxk

// Memory / error plumbing of rfx_api.hip for the other translation units.
namespace rfxi {
void* dmalloc(rfx_ctx*, size_t bytes);
void dfree(rfx_ctx*, void*);
void set_error(const char* msg);
hipError_t sync(rfx_ctx*);                                               // stream sync + queued read-backs
hipError_t queue_read(rfx_ctx*, void* dst, const void* d_src, size_t n);  // lands at the next sync
// once per ctx and kernel (`bit` names the kernel): allow `bytes` of dynamic LDS; false (and the ctx poisoned) if refused
bool lds_opt_in(rfx_ctx*, const void* fn, size_t bytes, int bit, const char* name);
}  // namespace rfxi

// Launch bracket: records a HIP-event span on the ctx stream when profiling is on.
struct rfx_span {
  rfx_ctx* c;
  rfx_prof_span s;
  bool on;
  rfx_span(rfx_ctx* ctx, const char* name)
      : c(ctx),
        on(ctx->prof && (ctx->prof_filter.empty() ||
                         ctx->prof_filter.find(std::string(",") + name + ",") != std::string::npos)) {
    if (on) {
      s.name = name;
      auto get = [&](hipEvent_t& e) {
        if (c->free_events.empty()) (void)hipEventCreate(&e);
        else {
          e = c->free_events.back();
          c->free_events.pop_back();
        }
      };
      get(s.e0);
      get(s.e1);
      (void)hipEventRecord(s.e0, c->stream);
    }
  }
  ~rfx_span() {
    if (on) {
      (void)hipEventRecord(s.e1, c->stream);
      c->spans.push_back(s);
    }
  }
};
