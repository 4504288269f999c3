/* rufus_hip.h -- C-ABI of librufus_hip.so, the MI355X (gfx950) implementation of the RUFUS
 * k-mer count -> unique-k-mer set difference -> read filter hot path.
 *
 * The reference (jandrewrfarrell/RUFUS) has no FFI: its boundary is a set of executables glued by
 * bash (runRufus.sh:748-759).  The drop-in executables under rufus_amd/csrc/host/ keep that argv /
 * file contract and call the entry points below; each entry point names the reference code whose
 * arithmetic it replaces (paths relative to the reference root, "jf/" = src/modifiedJellyfish/).
 *
 * Conventions: plain C, opaque handles, return 0 on success or a negative RFX_E_* code; the
 * caller owns every host buffer, the library owns device memory unless a function name ends in
 * _dev (caller-provided device pointers).  One rfx_ctx per (process, device); a ctx and the
 * objects made from it must be used from one thread at a time.  There is NO CPU fallback: every
 * device entry point fails with RFX_E_NODEVICE when no gfx950 GPU is visible.
 */
#ifndef RUFUS_HIP_H
#define RUFUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFX_OK 0
#define RFX_E_NODEVICE (-1) /* no usable gfx950 device / HIP runtime error at open */
#define RFX_E_INVAL (-2)    /* bad argument */
#define RFX_E_NOMEM (-3)    /* device or host allocation failed / hbm budget exceeded */
#define RFX_E_FULL (-4)     /* count table cannot grow inside the budget (use a key range pass) */
#define RFX_E_HIP (-5)      /* a HIP call failed; see rfx_last_error() */
#define RFX_E_MIXEDCASE (-6) /* lower-case acgt present: count and filter encodings differ, pack separately */
#define RFX_E_RANGE (-7)    /* output buffer too small */
#define RFX_E_FORMAT (-8)   /* malformed input text / file */

#define RFX_HISTO_BINS 10002 /* jf/sub_commands/histo_main.cc:33-89 with the default low=1 high=10000 inc=1 */

typedef struct rfx_ctx rfx_ctx;
typedef struct rfx_reads rfx_reads;     /* a block of 2-bit packed reads resident in HBM */
typedef struct rfx_table rfx_table;     /* exact k-mer count table in HBM */
typedef struct rfx_records rfx_records; /* (pos,key)-sorted records in HBM == payload of a .Jhash file */
typedef struct rfx_set rfx_set;         /* mutant k-mer set (filter / annotate) */

const char* rfx_version(void);
const char* rfx_strerror(int code);
const char* rfx_last_error(void); /* text of the last HIP failure on this thread */

/* ---------------------------------------------------------------------------------------------
 * Host-only helpers (no device needed)
 * ------------------------------------------------------------------------------------------- */

/* Hash matrix of a 2^lsize-slot jellyfish table for k-mers: 2k columns of lsize bits.
 * Replaces jf/include/jellyfish/large_hash_array.hpp:942-950 (RectangularBinaryMatrix(ceilLog2(size),
 * 2k).randomize_pseudo_inverse()), jf/lib/misc.cc:74-80 (random_bits over the unseeded glibc
 * random()) and jf/lib/rectangular_binary_matrix.cc:138-186,:209-216. */
int rfx_jf_matrix(int lsize, int k, uint64_t* cols);
/* pos = (M * key) & (2^lsize - 1); jf/include/jellyfish/rectangular_binary_matrix.hpp:206-243. */
uint64_t rfx_jf_pos(const uint64_t* cols, int k, int lsize, uint64_t key);

/* Packing.  A read of L bases occupies ceil(L/32) 64-bit code words (base i of the read at bits
 * 2*(i%32) of word i/32, jellyfish codes A0 C1 G2 T3) and as many 32-bit mask words.
 *   RFX_PACK_COUNT : acgt mask bit = base is one of ACGTacgt (jf/include/jellyfish/mer_dna.hpp:46-63);
 *                    codes follow that table (lower case accepted).
 *   RFX_PACK_FILTER: good mask bit = (qual-33 >= min_q as signed char) && base != 'N'
 *                    (src/RUFUS.Filter.cpp:205); codes follow Util::HashToLong (src/Util.cpp:51-84):
 *                    upper-case ACGT only, anything else encodes as A.
 * Both flags together return RFX_E_MIXEDCASE if a lower-case c/g/t is present (the encodings differ).
 * seq/qual are concatenated bytes, off[n_reads+1] their byte offsets (qual shares off; a missing
 * quality byte -- qual == NULL -- reads as '\0', i.e. bad).  Outputs sized by rfx_pack_words();
 * a block holds fewer than 2^32 words. */
#define RFX_PACK_COUNT 1
#define RFX_PACK_FILTER 2
uint64_t rfx_pack_words(const uint64_t* off, uint32_t n_reads);
int rfx_pack_reads(const char* seq, const char* qual, const uint64_t* off, uint32_t n_reads, int min_q, int flags,
                   uint64_t* codes, uint32_t* acgt, uint32_t* good, uint32_t* word_off /* n_reads+1 */,
                   uint32_t* len /* n_reads */);

/* The same packing for reads that lie scattered in a text buffer (FASTQ parsed in place): read i = seq_len[i]
 * bytes at base + seq_start[i], qualities (RFX_PACK_FILTER only) as many bytes at base + qual_start[i] (byte
 * distances modulo 2^64: a span may lie in another allocation than `base`, e.g. a rewritten copy of the read).
 * Single-threaded, so that a caller can pack disjoint ranges of one block from several threads: word_off[0] is
 * an INPUT (first word of this range in the block); word_off[1..n] and len[0..n) are written, codes / acgt / good
 * are indexed by those offsets. */
int rfx_pack_spans(const char* base, const uint64_t* seq_start, const uint32_t* seq_len, const uint64_t* qual_start,
                   uint32_t n_reads, int min_q, int flags, uint64_t* codes, uint32_t* acgt, uint32_t* good,
                   uint32_t* word_off /* n_reads+1 */, uint32_t* len /* n_reads */);

/* RUFUS.Filter hash-list loader (src/RUFUS.Filter.cpp:121-143; single_end: src/RUFUS.Filter.ss.cpp:
 * 100-118): every line contributes HashToLong(kmer) and HashToLong(RevComp(kmer)) (src/Util.cpp:51-84,
 * :187-210), returned here as forward keys in jellyfish encoding (first base most significant).
 * keys_out may be NULL to query the count.  Returns the number of keys (2 per accepted line, not
 * de-duplicated) or a negative code. */
long rfx_hashlist_keys(const char* text, size_t n, int k, int single_end, uint64_t* keys_out, size_t cap);

/* .Jhash header (jf/include/jellyfish/generic_file_header.hpp:96-121, file_header.hpp:33-110):
 * 9-digit length + terse JSON + NUL pad to 8 bytes.  Writes into buf (cap bytes), returns its
 * length or a negative code.  cmdline may be NULL. */
long rfx_jhash_header(int k, int lsize, const uint64_t* cols, int canonical, int counter_len, int argc,
                      const char* const* argv, char* buf, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * Device context
 * ------------------------------------------------------------------------------------------- */
rfx_ctx* rfx_open(int device, size_t hbm_budget_bytes); /* NULL on failure (no CPU fallback) */
void rfx_close(rfx_ctx*);
int rfx_sync(rfx_ctx*);
/* Page-locked host memory for staging buffers of the ingest pipelines (uploads from it run at PCIe speed). */
void* rfx_host_alloc(size_t bytes);
/* The same memory without a call into the HIP runtime: usable as host memory at once -- a tool fills its staging
 * buffers while the device is still being opened on another thread -- and page-locked later, by rfx_host_pin (idempotent;
 * ~1 ms per 320 MB of huge pages), before the first upload from it.  Freed by rfx_host_free. */
void* rfx_host_alloc_lazy(size_t bytes);
int rfx_host_pin(void* p);
void rfx_host_free(void*);
/* Host threads this process can keep busy: hardware threads, cut to its CPU affinity and to the CPU-bandwidth quota
 * of its cgroup (a container given 16 CPUs' worth of time on a 256-thread host runs 64 parser threads three times
 * SLOWER than 16: measured).  The -t / Threads arguments of the drop-in tools are capped by this. */
unsigned rfx_host_cpus(void);
void* rfx_stream(rfx_ctx*); /* the hipStream_t every kernel of this ctx is launched on */
/* Device memory of the ctx: bytes in use now, the high-water mark of that, and bytes of HBM mapped into the ctx's
 * arena (the library sub-allocates one growable virtual range; mapped memory is kept until rfx_close). */
/* The device memory a ctx hands out is mapped into its address range as it is first needed (and kept); mapping costs
 * ~4 ms per GiB.  A caller that knows it will need `bytes` in total -- `jellyfish count` while it still parses its
 * input: the shard passes at finish take ~130 GB for a 30x sample -- has them mapped ahead, off its critical path.
 * Never more than the device can give: a request beyond 90 % of the device's free memory is refused with RFX_E_NOMEM
 * before anything is mapped (nothing is lost and nothing is taken from other users of the device; the memory is mapped
 * when it is needed). */
int rfx_mem_reserve(rfx_ctx*, uint64_t bytes);
int rfx_mem_stats(rfx_ctx*, uint64_t* used, uint64_t* peak, uint64_t* mapped);
/* Device-to-device copy on the ctx stream, then stream sync (hand-off to / from buffers another
 * runtime owns, e.g. the RCCL exchange buffers of the multi-GPU path). */
int rfx_memcpy_dev(rfx_ctx*, void* d_dst, const void* d_src, size_t bytes);

/* Per-kernel HIP-event timing on the ctx stream (used by bench.py for roofline.achieved). */
int rfx_prof_enable(rfx_ctx*, int on);
/* Restrict the brackets to a comma-separated list of kernel names (NULL or "" = every launch): an
 * event pair per launch costs ~10 us of host time, which matters when a step is ~100 launches. */
int rfx_prof_filter(rfx_ctx*, const char* kernel_names);
int rfx_prof_reset(rfx_ctx*);
int rfx_prof_query(rfx_ctx*, const char* kernel, double* total_ms, uint64_t* launches);
int rfx_prof_names(rfx_ctx*, char* buf, size_t cap); /* '\n'-separated kernel names seen so far */

/* ---------------------------------------------------------------------------------------------
 * Read blocks
 * ------------------------------------------------------------------------------------------- */
/* H2D copy of a packed block (arrays as produced by rfx_pack_reads; good/acgt may be NULL when
 * the block will only be counted / only be filtered). */
rfx_reads* rfx_reads_upload(rfx_ctx*, const uint64_t* codes, const uint32_t* acgt, const uint32_t* good,
                            const uint32_t* word_off, const uint32_t* len, uint32_t n_reads);
void rfx_reads_free(rfx_reads*);
uint32_t rfx_reads_count(const rfx_reads*);
uint64_t rfx_reads_bases(const rfx_reads*);
uint64_t rfx_reads_words(const rfx_reads*);
/* D2H copy of a block's arrays (sized by rfx_reads_words / rfx_reads_count; any pointer may be NULL). */
int rfx_reads_get(const rfx_reads*, uint64_t* codes, uint32_t* acgt, uint32_t* good, uint32_t* word_off, uint32_t* len);

/* -------------------------------------------------------------------------------------------
 * Text in, packed reads out -- on the device (kernel K1)
 * Replaces, for strict 4-line FASTQ, the reference's text parser and base packer
 * (jf/include/jellyfish/mer_overlap_sequence_parser.hpp:179-206 -- header, sequence, '+', quality line --;
 * jf/include/jellyfish/mer_dna.hpp:46-63; for the filter's blocks src/Util.cpp:51-84 and the quality test of
 * src/RUFUS.Filter.cpp:205), which the drop-in executables ran on the host until round 6.  The host only moves
 * bytes: it appends RECORD-ALIGNED pieces of the text (every piece begins at a record's '@' line and ends with the
 * newline of a quality line) to a device arena, each piece one host-to-device copy queued on the ctx stream -- from
 * page-locked memory (rfx_host_alloc) it runs at PCIe speed and the call returns at once -- and rfx_text_parse
 * turns what was appended into a read block exactly as rfx_pack_reads(flags, min_q) would have packed the same
 * records (flags = RFX_PACK_COUNT or RFX_PACK_FILTER, one of them).
 *   rfx_text_open    an arena for up to cap_bytes (< 4 GiB) of text; NULL on failure
 *   rfx_text_append  >= 0: a ticket for this piece; < 0: an RFX_E_* code (RFX_E_RANGE: no room).  The host buffer must
 *                    stay as it is until rfx_text_copied(ticket) returns 1 (0: the copy has not run yet; < 0: error),
 *                    rfx_text_wait(ticket) or rfx_text_parse has returned.  The copies run on a stream of the arena's
 *                    own: with two arenas the next block's text crosses PCIe while this block is parsed and counted.
 *                    append / copied / wait may be called from several host threads; parse, fetch, reset and close
 *                    from one, while nobody appends
 *   rfx_text_parse   the block, or NULL: *strict == 0 then says the text is NOT strict 4-line FASTQ (a blank line
 *                    between records, a multi-line record, a quality line of another length, a missing final
 *                    newline ...) -- no error: the caller parses that text on the host (rfx_text_fetch copies it
 *                    back); *strict == 1 with NULL is a failure (rfx_last_error).  Waits for its kernels: when it
 *                    returns the arena is free for the next block's text (rfx_text_reset).
 *   rfx_text_fetch   the appended bytes, copied to `host` (rfx_text_bytes of them)
 * ------------------------------------------------------------------------------------------- */
typedef struct rfx_text rfx_text;
rfx_text* rfx_text_open(rfx_ctx*, uint64_t cap_bytes);
void rfx_text_close(rfx_text*);
uint64_t rfx_text_room(const rfx_text*);
uint64_t rfx_text_bytes(const rfx_text*);
long rfx_text_append(rfx_text*, const void* host, uint64_t n);
int rfx_text_copied(rfx_text*, long ticket);
int rfx_text_wait(rfx_text*, long ticket); /* blocks until that append's bytes have left the host buffer */
rfx_reads* rfx_text_parse(rfx_text*, int flags, int min_q, int* strict);
int rfx_text_fetch(rfx_text*, void* host);
void rfx_text_reset(rfx_text*);

/* Synthetic trio workload (BASELINE.json configs[1]-[4]) -- benchmark and scale-test input,
 * not a replacement of any reference code.  Every base is a pure function of (parameters, pair, mate, base
 * index) -- see rufus_amd/csrc/rfx_synth.h -- so a 30x WGS sample (6.2e8 reads) is generated straight into
 * packed read blocks in HBM, and any slice of it can be regenerated as text on the host for the oracle. */
typedef struct rfx_synth {
  uint64_t genome_len;  /* bases, 4000 <= genome_len < 2^32 */
  uint64_t genome_seed; /* uniform random genome */
  uint64_t snv_seed;    /* planted SNVs: one per stratum of (genome_len - 2000) / n_snv bases */
  uint64_t read_seed;   /* one per sample */
  uint32_t n_snv;       /* (genome_len - 2000) / n_snv must be >= 2 * read_len + 64 */
  uint32_t read_len;    /* <= 256 */
  uint32_t insert_lo, insert_span; /* insert = insert_lo + U[0, insert_span), insert_lo >= read_len */
  uint32_t err_1024;    /* substitution errors per 1024 bases */
  uint32_t lowq_256;    /* bases with quality '#' (else 'J') per 256 */
  uint32_t n_1024;      /* 'N' per 1024 bases */
  uint32_t carrier;     /* 1: pairs of haplotype 1 carry the SNVs (the child); 0: reference only (a parent) */
} rfx_synth;
/* Reads 2*first_pair .. 2*(first_pair+n_pairs)-1 of the sample (read 2p = mate 1 of pair p, 2p+1 = mate 2),
 * packed as rfx_pack_reads(RFX_PACK_COUNT [| RFX_PACK_FILTER with min_q]) would pack their text. */
/* want_good: bit 0 = also the filter's `good` mask; bit 1 (RFX_SYNTH_COMPACT) = the compact block form -- reads of
 * one length need no offset / length arrays, and the ACGT mask is kept only for the reads that hold an N (a bit per
 * read says which): 43 instead of 68 bytes per 150 bp read resident in HBM.  The kernels read both forms; the
 * global-table count path (k = 32) takes only the dense one.  rfx_reads_get hands out the dense arrays either way. */
#define RFX_SYNTH_GOOD 1
#define RFX_SYNTH_COMPACT 2
rfx_reads* rfx_synth_reads(rfx_ctx*, const rfx_synth*, uint64_t first_pair, uint32_t n_pairs, int min_q,
                           int want_good);
/* Bytes of device memory a read block holds. */
uint64_t rfx_reads_device_bytes(const rfx_reads*);
/* Host twin: the same reads as text, read after read: seq and qual hold 2*n_pairs*read_len bytes each. */
int rfx_synth_text(const rfx_synth*, uint64_t first_pair, uint32_t n_pairs, char* seq, char* qual);
/* SNV i: 0-based genome position, reference and alternative base ('A','C','G','T'). */
int rfx_synth_snv(const rfx_synth*, uint32_t i, uint64_t* pos, char* ref, char* alt);
/* Genome bases [first, first+n) as text (host). */
int rfx_synth_genome(const rfx_synth*, uint64_t first, uint64_t n, char* out);

/* ---------------------------------------------------------------------------------------------
 * K2: canonical k-mer count  (jellyfish count: jf/sub_commands/count_main.cc:148-180;
 * jf/include/jellyfish/mer_iterator.hpp:59-88; hash_counter.hpp:98-119; large_hash_array.hpp:298-302)
 * ------------------------------------------------------------------------------------------- */
/* k <= 31 (32 when canonical).  lsize = ceilLog2 of jellyfish's -s.  Only k-mers whose pos lies in
 * [pos_lo, pos_hi) are counted (pos_hi == 0 means 2^lsize): key-range passes / multi-GPU ownership.
 * capacity_slots == 0 picks an initial size; the table grows by rehash inside the ctx budget. */
rfx_table* rfx_count_begin(rfx_ctx*, int k, int canonical, int lsize, uint64_t capacity_slots, uint64_t pos_lo,
                           uint64_t pos_hi);
/* Three exact implementations sit behind rfx_count_add(); results are identical.
 *  RFX_COUNT_MSP   (23 <= k <= 31) cuts reads into super-k-mers (ALL consecutive k-mers of a read that share their
 *                  minimizer: one record of 12 bytes -- a 64-bit word + a 32-bit plane -- per ~5.7 k-mers at k = 25),
 *                  partitions those, counts every bin in LDS and sorts only the surviving (key,count) pairs into
 *                  (pos,key) order.  Fastest, least HBM.
 *  RFX_COUNT_P2L   (2k <= 62) partitions one 8-byte sortable word per k-mer instance by (pos,key) prefix
 *                  and counts + sorts every bin in LDS.
 *  RFX_COUNT_TABLE inserts into an open-addressed table in HBM (any k, grows by rehash; also what
 *                  rfx_count_add_pairs_dev uses).
 * RFX_COUNT_AUTO (default) takes MSP where it applies, else P2L, and falls back to the table when the
 * transient buffers do not fit the budget.  MSP sizes its buffers optimistically; if the device reports
 * that one did not hold, the affected read block is partitioned again with exact sizes -- at
 * rfx_count_finish, or inside rfx_reads_free if the block is freed first (the redo needs the reads). */
#define RFX_COUNT_AUTO 0
#define RFX_COUNT_TABLE 1
#define RFX_COUNT_P2L 2
#define RFX_COUNT_MSP 3
int rfx_count_set_mode(rfx_table*, int mode);
/* Shard passes (MSP path only, call before the first add): count only the k-mers whose minimizer bin belongs
 * to shard `shard` of `n_shards` (<= 256) -- the same cut as the multi-GPU owner ranges.  The shards of a
 * sample are disjoint and cover it (their records interleave to the full result in (pos,key) order), and
 * the shard of a k-mer is the same for every sample, so count -> set difference can run shard by shard
 * when a sample's records do not fit the HBM at once, or on every GPU over all reads without any exchange. */
int rfx_count_set_shard(rfx_table*, int shard, int n_shards);
/* Shard passes hash every read once per pass: what cuts a read into super-k-mers is the same work whichever shard's
 * runs are kept.  rfx_count_set_early(t, 1) on a table of shard s < S - 1: while a BIG block (>= 2^29 windows) is
 * added, the runs of shard s + 1 are kept too -- the one k_msp_part1 launch covers both shards' bins (and, when those
 * are all the bins, no longer asks whose a run is) -- and partitioned into segments of their own, held aside (at the
 * cost of their memory: 12 bytes per record of shard s + 1) until rfx_count_adopt_early(t_next, t) hands them to the
 * table of shard s + 1 of the same sample; that table is then NOT given those blocks.  Applies only where the boundary
 * between the two shards is a boundary of coarse bins (S = 2, 4, ...; otherwise the add is an ordinary one:
 * rfx_count_early_segments() says how many blocks went early, in the order they were added).  The WGS driver uses it
 * for as many blocks of the subject as the headroom of the device allows (rufus_amd/wgs.py). */
int rfx_count_set_early(rfx_table*, int on);
int rfx_count_early_segments(const rfx_table*);
int rfx_count_adopt_early(rfx_table* next_shard, rfx_table* from);
/* Hash every base ONCE over all shard passes (the reference hashes a k-mer once: jf/sub_commands/count_main.cc:148-180;
 * its spill-and-merge analogue, jf/include/jellyfish/hash_counter.hpp:182-202, does not re-read the input either).  A
 * store of RUN MAPS is shared by the tables of one sample's shard passes (rfx_count_set_runmaps on each, before its
 * adds): the first table that adds a big block (>= 2^29 windows, reads of <= 160 bases, its shard at most half of the
 * bins) also writes the block's map -- 32 bytes per read that say how the read falls into super-k-mers and where
 * their minimizers sit -- and every later table of another shard rebuilds ITS records from reads + map (k_msp_replay)
 * instead of hashing the block again: the same records, bit for bit.  budget_bytes bounds the maps held (0: no bound);
 * a block beyond it is hashed by every pass as before.  rfx_runmaps_drop frees one block's map (after the last
 * pass has added it); the store must be freed before the read blocks it was made from.  rfx_count_set_passes tables
 * make and use a store of their own when the device has room.  rfx_count_replayed: blocks a table added by replay. */
typedef struct rfx_runmaps rfx_runmaps;
rfx_runmaps* rfx_runmaps_create(rfx_ctx*, uint64_t budget_bytes);
/* the same with ONE device allocation of pool_bytes made now, out of which the maps are cut: a store that serves many
 * samples one after the other (the WGS driver) does not leave the device memory in pieces; NULL when it does not fit */
rfx_runmaps* rfx_runmaps_create_pooled(rfx_ctx*, uint64_t pool_bytes);
void rfx_runmaps_free(rfx_runmaps*);
uint64_t rfx_runmaps_bytes(const rfx_runmaps*);
int rfx_runmaps_blocks(const rfx_runmaps*);
int rfx_runmaps_drop(rfx_runmaps*, const rfx_reads*);
int rfx_runmaps_clear(rfx_runmaps*); /* every map of the store */
int rfx_count_set_runmaps(rfx_table*, rfx_runmaps*);
/* the maps of several blocks the table is about to add, made with ONE wait for the device (a map made by rfx_count_add
 * waits for its own launch); blocks that have a map, or are no blocks for one, are skipped; no store: nothing happens */
int rfx_count_prepare_maps(rfx_table*, rfx_reads* const* blocks, int n);
/* The same launches queued on the ctx's SECOND stream, no wait: the maps of the sample that is counted next are made
 * beside this sample's partition, refinement and sort (the hashing launch is bound by the instructions it issues, those
 * by the memory; they share a CU).  Pooled store only, as far as it has room; whoever next asks the store about one of
 * these blocks (rfx_count_prepare_maps, rfx_count_add, rfx_runmaps_drop / _clear / _free) waits for the launches first.
 * The blocks must stay alive until then.  Returns the number of launches queued (0: nothing to do or no room), < 0: error. */
int rfx_count_prefetch_maps(rfx_table*, rfx_reads* const* blocks, int n);
uint64_t rfx_count_replayed(const rfx_table*);
/* Bounded-HBM counting of a whole sample (MSP path, call before the first add): rfx_count_add() only
 * REMEMBERS the read blocks -- they must stay alive until finish -- and rfx_count_finish() runs `passes`
 * minimizer-shard passes over them (0: planned from the free HBM, 1 if everything fits): per pass the shard's
 * super-k-mer records of every block are built, counted and freed again, the survivors of all passes are sorted
 * once.  A 30x human sample is 187 GB of records but 42 GB of packed reads: this is how one GPU counts it, the
 * analogue of jellyfish's --disk spill-and-merge (jf/include/jellyfish/hash_counter.hpp:182-202,
 * jf/sub_commands/count_main.cc:326-339).  The table is consumed by its finish. */
int rfx_count_set_passes(rfx_table*, int passes);
int rfx_count_add(rfx_table*, const rfx_reads*);
/* Several devices behind ONE executable (runRufus.sh:776-797 calls binaries, so the N GPUs of a node have
 * to be reachable from `jellyfish count` itself, RUFUS_GPUS=0-7).  N tables, one per device.  Each is given ITS read
 * blocks only -- block b goes to table b mod N: the sample is sharded by read block, a device uploads and hashes 1/N of
 * it.  At finish, pass by pass, every table partitions its blocks; the minimizer bins of a pass are cut into N owner
 * ranges (the cut of rfx_count_set_shard, the same on every device) and table g PULLS the record runs of its range
 * from every table (device-to-device, xGMI) before it counts them: every instance of a canonical k-mer has the same
 * minimizer, so an owner sees complete bins -- exact counts, no partial sums, no reduce (runRufus.sh shards by
 * chromosome on the CPU; here by read block and minimizer owner).  Before the survivors are sorted they change hands
 * once more so that table i ends up with slice i of the OUTPUT POSITIONS: rfx_count_finish of table i returns slice i
 * of the (pos,key)-ordered payload, and the .Jhash is the slices one after the other -- the owner partition of jf's
 * sorted dumper (jf/include/jellyfish/sorted_dumper.hpp:80-112) without a merge.  A table that was given no block
 * still takes part.  (RFX_PEERS_REPLICATE=1 in the environment when the group is created: round 3's fallback -- every
 * table is given EVERY block and keeps minimizer shard i of N; only survivors change hands.)
 *   rfx_ctx_allow_peers   before the first allocation of a ctx: the devices that may read its memory directly
 *   rfx_peers_create(n)   the meeting point of n tables
 *   rfx_count_set_peers   after rfx_count_set_passes, before the first add: this table is number `index` of the group
 * The n rfx_count_finish calls must run concurrently (one host thread each): they meet at barriers (two to agree on the
 * number of passes, two per pass, two for the survivors); if one fails they all fail. */
typedef struct rfx_peers rfx_peers;
int rfx_ctx_allow_peers(rfx_ctx*, const int* devices, int n);
rfx_peers* rfx_peers_create(int n);
void rfx_peers_free(rfx_peers*);
int rfx_count_set_peers(rfx_table*, rfx_peers*, int index);
/* Merge pre-aggregated (key,count) pairs (device pointers): owner-side reduce of the multi-GPU
 * exchange, and the rehash path. */
int rfx_count_add_pairs_dev(rfx_table*, const uint64_t* d_keys, const uint32_t* d_counts, uint64_t n);
int rfx_count_stats(rfx_table*, uint64_t* distinct, uint64_t* capacity, uint64_t* max_displacement);

/* Multi-GPU sharding of the MSP path (rufus_amd/dist.py): the super-k-mer records of a read block are
 * grouped by minimizer bin, and every instance of a canonical k-mer lives in the same bin on every
 * rank.  So ranks exchange RECORDS by bin owner (contiguous runs of the record array) and each owner
 * counts complete bins: no partial counts, no reduce.
 *   rfx_count_segments       number of record segments held (one per rfx_count_add), after settling
 *                            any pending capacity check; < 0 on error, 0 if the table is not on the MSP path.
 *   rfx_count_segment_get    device pointers of segment i: records grouped by bin, bin_start[bins+1].
 *                            Valid until the next add/finish/free on the table.
 *   rfx_count_add_records_ext_dev  append a copy of records grouped the same way (bin b = bin_start[b]..
 *                            bin_start[b+1]); `bins` is a power of two >= 256 and the table must be MSP
 *                            capable (23 <= k <= 31).  A record is a 64-bit word AND a 32-bit plane entry
 *                            (rfx_count_segment_ext): both arrays travel.  All segments of a table must come from
 *                            the same k / canonical setting; bins may differ (finish refines to a common count). */
int rfx_count_segments(rfx_table*);
int rfx_count_segment_get(rfx_table*, int i, const uint64_t** d_records, const uint64_t** d_bin_start, uint32_t* bins,
                          uint64_t* n_records);
int rfx_count_add_records_dev(rfx_table*, const uint64_t* d_records, uint64_t n_records, const uint64_t* d_bin_start,
                              uint32_t bins);
/* The same import without the copy: the table reads d_records (and d_ext) where they are until rfx_count_finish /
 * rfx_count_free -- the caller keeps them alive that long; only the bin offsets are copied. */
int rfx_count_adopt_records_dev(rfx_table*, const uint64_t* d_records, const uint32_t* d_ext, uint64_t n_records,
                                const uint64_t* d_bin_start, uint32_t bins);
/* A record is a 64-bit word plus a 32-bit plane entry (the bases of a super-k-mer beyond the 28 the word holds: up to
 * 35 bases at k = 25, 43 at k = 31); the plane is grouped like the words and travels with them.  (Until round 3 only
 * k = 26 .. 31 had a plane; rfx_count_add_records_dev -- words alone -- now always fails.) */
int rfx_count_segment_ext(rfx_table*, int i, const uint32_t** d_ext);
int rfx_count_add_records_ext_dev(rfx_table*, const uint64_t* d_records, const uint32_t* d_ext, uint64_t n_records,
                                  const uint64_t* d_bin_start, uint32_t bins);
void rfx_count_free(rfx_table*);

/* K3: table -> records with lower <= count <= upper in (pos,key) order (jf/include/jellyfish/
 * sorted_dumper.hpp:80-112, mer_heap.hpp:34-38; -L/-U at output count_main.cc:318-324) plus the
 * count-of-counts histogram of exactly those records (histo_main.cc:33-89), histo may be NULL.
 * A table may be finished more than once (different bounds, more reads in between) -- except an MSP table
 * holding more than 4 GB of super-k-mer records: its finish frees them before the survivors are sorted
 * (at WGS scale both do not fit side by side), so it can only be freed afterwards. */
rfx_records* rfx_count_finish(rfx_table*, uint64_t lower, uint64_t upper, uint64_t* histo /* RFX_HISTO_BINS */);
/* The same in two steps, so that several tables can be queued on the device before the host waits for
 * the first: _begin launches the work (nothing is waited for on the MSP path; other paths finish inside
 * _begin), _end waits, returns the records (NULL on error) and frees the handle.  `histo` must stay
 * valid until _end; the table must not be touched in between. */
typedef struct rfx_finish rfx_finish;
rfx_finish* rfx_count_finish_begin(rfx_table*, uint64_t lower, uint64_t upper, uint64_t* histo /* RFX_HISTO_BINS */);
rfx_records* rfx_count_finish_end(rfx_finish*);

/* ---------------------------------------------------------------------------------------------
 * Records (the .Jhash payload; jf/include/jellyfish/binary_dumper.hpp:44-48)
 * ------------------------------------------------------------------------------------------- */
uint64_t rfx_records_size(const rfx_records*);
int rfx_records_k(const rfx_records*);
int rfx_records_lsize(const rfx_records*);
/* Formatted file records (ceil(2k/8) key bytes + counter_len count bytes, saturating) -> host. */
int rfx_records_payload(const rfx_records*, void* out, size_t cap_bytes, int counter_len);
/* The same for records [first, first + n): lets a writer stream a 30 GB payload through a small buffer. */
int rfx_records_payload_range(const rfx_records*, uint64_t first, uint64_t n, void* out, size_t cap_bytes,
                              int counter_len);
int rfx_records_get(const rfx_records*, uint64_t* keys, uint32_t* counts, uint64_t* pos); /* any may be NULL */
/* Load a payload read from a .Jhash file back into HBM (verifies (pos,key) order). */
rfx_records* rfx_records_load(rfx_ctx*, int k, int lsize, const uint64_t* cols, const void* payload, uint64_t n,
                              int counter_len);
/* The same straight from an open .Jhash file: n records at byte `offset` of fd, streamed through a ring of
 * page-locked buffers (several threads pread, the copies and the parsing overlap) -- neither a host copy of the
 * payload nor a device copy of it is ever whole (a 30x sample's file is 35 GB).  The file is only read. */
rfx_records* rfx_records_load_fd(rfx_ctx*, int k, int lsize, const uint64_t* cols, int fd, uint64_t offset, uint64_t n,
                                 int counter_len);
rfx_records* rfx_records_from_dev(rfx_ctx*, int k, int lsize, const uint64_t* cols, const uint64_t* d_keys,
                                  const uint32_t* d_counts, uint64_t n); /* copies; computes pos */
const uint64_t* rfx_records_dev_keys(const rfx_records*);
const uint32_t* rfx_records_dev_counts(const rfx_records*);
const uint64_t* rfx_records_dev_pos(const rfx_records*);
int rfx_records_histo(const rfx_records*, uint64_t* histo /* RFX_HISTO_BINS */);
/* Self-check of a record set where it lies (full-size runs are beyond any CPU oracle): out[0] = records out of the
 * strict (pos,key) order of jf/include/jellyfish/sorted_dumper.hpp:80-112, out[1] = records whose pos is not M * key
 * (jf/include/jellyfish/rectangular_binary_matrix.hpp:206-243), out[2] = counts outside [min_count, max_count],
 * out[3] = sum of the counts.  A correct set gives 0, 0, 0. */
int rfx_records_verify(const rfx_records*, uint32_t min_count, uint32_t max_count, uint64_t out[4]);
/* A checksum of the record MULTISET that does not depend on how the count was cut into shard passes, devices or slices
 * (sums over the slices add up, mod 2^64): out[0] = sum of mix(key) * count, out[1] = sum of mix(key), mix = the
 * splitmix64 finaliser (x += 0x9E3779B97F4A7C15; x = (x ^ x >> 30) * 0xBF58476D1CE4E5B9; x = (x ^ x >> 27) *
 * 0x94D049BB133111EB; x ^ x >> 31).  The full-size runs compare it between S and S + 1 passes (there is no .Jhash of a
 * real jellyfish to hold a 3e9-record table against: same (key, count) pairs from two different cuts of the work is the
 * strongest statement available); the parity tests hold it against the oracle's records. */
int rfx_records_checksum(const rfx_records*, uint64_t out[2]);
void rfx_records_free(rfx_records*);

/* ---------------------------------------------------------------------------------------------
 * K4: set difference
 * ------------------------------------------------------------------------------------------- */
/* RUFUS's modified `jellyfish merge` (jf/jellyfish/merge_files.cc:69-155): keys present in exactly
 * one of the inputs with count >= min_count (5 in the reference), in global (pos,key) order. */
int rfx_merge_unique(rfx_ctx*, const rfx_records* const* files, int n_files, uint32_t min_count, uint64_t* keys_out,
                     uint32_t* counts_out, uint64_t cap, uint64_t* n_out);
/* `jellyfish query` lookups (jf/include/jellyfish/binary_dumper.hpp:156-203): count of each key, 0 if
 * absent.  Keys must already be canonical when the database is. */
int rfx_query(const rfx_records* db, const uint64_t* keys, uint64_t n, uint32_t* counts_out);
/* Fused net semantics of runRufus.sh:925-926 + scripts/CheckJellyHashList.sh:12: subject keys with
 * max(min_count,min_cov) <= count <= max_cov that occur in no other input, in (pos,key) order. */
int rfx_unique_to_subject(rfx_ctx*, const rfx_records* subject, const rfx_records* const* others, int n_others,
                          uint32_t min_count, uint32_t min_cov, uint32_t max_cov, uint64_t* keys_out,
                          uint32_t* counts_out, uint64_t cap, uint64_t* n_out);
/* The same difference one input at a time, device to device: the records of `a` with min_count <= count <=
 * max_count that occur in none of `others`, as a new (pos,key)-ordered record set (NULL on error).  A caller that
 * counts the controls one after the other keeps only the shrinking candidate set of the subject instead of every
 * sample's records: subtract(subject, {}, MinCov, MaxDepth), then subtract(candidates, {control_i}, 0, ~0) per control
 * gives the keys of rfx_unique_to_subject (jf/jellyfish/merge_files.cc:69-155 + scripts/CheckJellyHashList.sh:12). */
rfx_records* rfx_records_subtract(rfx_ctx*, const rfx_records* a, const rfx_records* const* others, int n_others,
                                  uint32_t min_count, uint32_t max_count);

/* ---------------------------------------------------------------------------------------------
 * K5: read filter (src/RUFUS.Filter.cpp:196-277, src/RUFUS.Filter.ss.cpp:164-203)
 * ------------------------------------------------------------------------------------------- */
rfx_set* rfx_set_build(rfx_ctx*, const uint64_t* fwd_keys, uint64_t n, int k); /* keys from rfx_hashlist_keys */
uint64_t rfx_set_size(const rfx_set*);
void rfx_set_free(rfx_set*);
/* Per read: number of good-streak windows found in the set.  last_base_skipped = 1 reproduces the
 * paired tool's `i < length-1` loop bound (src/RUFUS.Filter.cpp:203), 0 the single-end tool.
 * hits_out[n_reads] and hitmask_out[ceil(n_reads/64)] (bit r%64 of word r/64 set when
 * hits >= thresh) are host buffers, either may be NULL; *n_hit_reads counts reads over threshold. */
int rfx_filter(rfx_set*, const rfx_reads*, int thresh, int last_base_skipped, uint32_t* hits_out,
               uint64_t* hitmask_out, uint64_t* n_hit_reads);
/* The blocks of a sample behind ONE wait for the device: hitmask_out[i] (may be NULL, as may the array) and n_hit_reads[i]
 * per block as rfx_filter gives them; no per-read counts. */
int rfx_filter_many(rfx_set*, const rfx_reads* const* blocks, int n, int thresh, int last_base_skipped,
                    uint64_t* const* hitmask_out, uint64_t* n_hit_reads);

/* ---------------------------------------------------------------------------------------------
 * N4: coverage model fit (src/ModelDist.cpp; runRufus.sh:849 runs it on every sample's histogram,
 * :862-868 read MutantMinCov and MutantSC from lines 2 and 4 of HISTO.7.7.model)
 * ------------------------------------------------------------------------------------------- */
/* One candidate model: the arguments of testModel / testModelLog (src/ModelDist.cpp:72-74, :200-202). */
typedef struct rfx_model_params {
  double sc, stdev, factor, skew, power;
} rfx_model_params;
/* Residuals of n_cand (<= 64) candidates against one histogram in one pass: what the reference computes by
 * n_cand calls of testModelLog (log_resid = 1: sum of (ln histo[i] - ln model[i])^2, src/ModelDist.cpp:72-198) or
 * testModel (0: sum of (histo[i] - model[i])^2, :200-318) inside its `omp parallel for` (:548-553 and the four
 * loops after it), i running over [inflection, sc * max_copy).  histo[0..n) as the reference indexes it (entry 0
 * unused, entry 1 = the first non-empty row of the file).  RFX_E_INVAL / RFX_E_RANGE where the reference would
 * index outside its tables (sc/2 < 1, fewer than 2 copies, sc * max_copy > n). */
int rfx_model_residuals(rfx_ctx*, const int64_t* histo, uint32_t n, const rfx_model_params* cand, int n_cand,
                        int log_resid, int inflection, int max_copy, double* resid_out);
/* Tables of one model as main() builds them for the output files (src/ModelDist.cpp:716-772; columns are summed
 * from row 0 there): dist[n][*n_cols + 1] row-major (column 0 is zero, column 1 the half-copy curve, column 1 + j
 * the j-copy curve; the last column is not normalised, as in the reference) and rowtot[n] = sum of columns
 * 1..*n_cols - 1 of each row.  *n_cols is set even when dist_cap (in doubles) is too small (RFX_E_RANGE). */
int rfx_model_tables(rfx_ctx*, uint32_t n, const rfx_model_params* model, uint32_t* n_cols, double* dist,
                     size_t dist_cap, double* rowtot);

/* ---------------------------------------------------------------------------------------------
 * K6: overlap scoring of the greedy assemblers; K7: per-base mutant k-mer coverage of contigs
 * ------------------------------------------------------------------------------------------- */
/* Align3 of OverlapSam / Overlap / OverlapRegion (src/OverlapSam.cpp:33-241, src/Overlap.cpp:169-360,
 * src/OverlapRegion.cpp:31-231) for ONE query `a` against `nb` candidates: every offset of the three
 * alignment phases is scored on the device and reduced in the reference's loop order (first offset
 * with the strictly greatest accepted score).  Per candidate j, out[5j..5j+4] =
 *   { phase-1 best score, its overlap, perfect flag (score == window),
 *     best score over all three phases (== phase 1 when perfect), its overlap };
 * a score equal to RFX_OVL_NONE_* (the variant's initial LocalBestScore) means "nothing accepted".
 * The caller applies the reference's cross-candidate rules (shared PerfectMatch, first-best wins).
 * variant: RFX_OVL_SAM / RFX_OVL_REGION (LocalBestScore starts at 0, '>=' in every phase) or
 * RFX_OVL_CONTIG (Overlap.cpp: starts at -1, phase 3 accepts with a strict '>'). */
#define RFX_OVL_SAM 0
#define RFX_OVL_CONTIG 1
#define RFX_OVL_REGION 2
int rfx_overlap_score(rfx_ctx*, const char* a, int alen, const char* const* b, const int* blen, int nb, float min_pct,
                      int min_ovl, int variant, int* out /* nb x 5 */);
/* The same scoring against a DEVICE-RESIDENT pool of sequences: the assemblers' outer loops are sequential (read i
 * merges into its best partner, which is seen later: src/OverlapSam.cpp:866-1024, src/Overlap.cpp:935-1126,
 * src/OverlapRegion.cpp:702-841), so one scoring call per read is unavoidable -- but not uploading the candidates
 * again for every call (OverlapRegion scores read i against ALL later reads).  The pool is uploaded once,
 * rfx_ovl_pool_set() patches the one entry a merge changes, a score call moves only the candidate indices and the
 * results, with no allocation.  query: a pool entry, or an explicit string (a_explicit != NULL).  strands: 0 = the
 * query as it is, 1 = its reverse complement (built on the device: the query must consist of ACGTN only, as
 * Util::RevComp drops other characters), 2 = both in one launch: out holds nb x 5 for the forward strand, then
 * nb x 5 for the reverse complement. */
typedef struct rfx_ovl_pool rfx_ovl_pool;
rfx_ovl_pool* rfx_ovl_pool_create(rfx_ctx*, const char* const* seqs, const int* lens, int n);
int rfx_ovl_pool_set(rfx_ovl_pool*, int idx, const char* seq, int len);
int rfx_ovl_pool_score(rfx_ovl_pool*, int query, const char* a_explicit, int a_len, const int* cand, int nb,
                       float min_pct, int min_ovl, int variant, int strands, int* out);
void rfx_ovl_pool_free(rfx_ovl_pool*);
/* AnnotateOverlap (src/AnnotateOverlap.cpp:88-134): for each packed contig of the block (good mask =
 * base != 'N' && qual-33 >= 3, i.e. RFX_PACK_FILTER with min_q 3) the number of mutant windows that
 * cover every base; cov_out holds rfx_reads_bases() counters, contig after contig. */
int rfx_annotate(rfx_set*, const rfx_reads*, uint32_t* cov_out);

#ifdef __cplusplus
}
#endif
#endif /* RUFUS_HIP_H */
