// Parallel FASTQ ingest of the drop-in `jellyfish count`: text in, packed read blocks in HBM out.
//
// The reference parses with one producer per file (jf/include/jellyfish/mer_overlap_sequence_parser.hpp:
// 124-251) and hashes in T threads.  Here the device counts ~100x faster than one core can parse, so the host
// side is the pipeline to get right:
//
//   regular file   mmap; the byte range is cut into ~32 MB pieces at record boundaries, every worker parses
//                  and packs pieces on its own (no reader thread in the way)
//   pipe / FIFO    one reader thread (the stream cannot be split) cuts the byte stream every 4k lines and hands
//                  the pieces to the workers
//   workers        find the sequence lines of their piece, reserve a range of the current staging block (pinned
//                  host memory: 2-bit codes, ACGT mask, offsets, lengths) and pack straight into it
//                  (rfx_pack_spans) -- no intermediate copy of the text
//   main thread    uploads every full block (rfx_reads_upload) and hands it to the caller (count it, or keep it
//                  for the shard passes of rfx_count_set_passes)
//
// Only strict 4-line FASTQ takes this path (what PassThroughSamCheck and every sequencer write); FASTA and
// multi-line FASTQ go through the sequential parser of rfx_cli.hpp, which follows the reference's grammar.
#pragma once
#include <sys/mman.h>
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

#include "rfx_cli.hpp"

#include "rfx_sam.hpp"
#include "rfx_packed_cache.hpp"

namespace rfxcli {

// The stream cutter's line count.  Fast pass: newlines of [p, e) and how many of them are blank lines (a '\n' at a
// line start); at_start (in / out): the byte at p begins a line.
static inline void count_newlines_plain(const char* p, const char* e, bool& at_start, size_t& nl, size_t& blanks) {
  for (; p < e; ++p) {
    if (*p == '\n') {
      ++nl;
      blanks += at_start;
      at_start = true;
    } else {
      at_start = false;
    }
  }
}
#if RFX_X86
__attribute__((target("avx2,popcnt"))) static inline void count_newlines_avx2(const char* p, const char* e, bool& at_start,
                                                                             size_t& nl, size_t& blanks) {
  const __m256i v = _mm256_set1_epi8('\n');
  while (e - p >= 32) {
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)p), v));
    nl += (size_t)__builtin_popcount(m);
    blanks += (size_t)__builtin_popcount(m & ((m << 1) | (at_start ? 1u : 0u)));
    at_start = (m >> 31) & 1u;
    p += 32;
  }
  count_newlines_plain(p, e, at_start, nl, blanks);
}
#endif
static inline void count_newlines(const char* p, const char* e, bool& at_start, size_t& nl, size_t& blanks) {
  nl = blanks = 0;
#if RFX_X86
  static const bool fast = (__builtin_cpu_init(), __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt"));
  if (fast) return count_newlines_avx2(p, e, at_start, nl, blanks);
#endif
  count_newlines_plain(p, e, at_start, nl, blanks);
}
// Slow pass (only for stretches that hold blank lines): the grammar parse_piece applies -- blank lines BETWEEN records
// are skipped, an empty sequence or quality line inside a record is a line.  line_in_rec: lines of the current
// record seen so far; rec_end: set to just after the last line that completed a record (untouched if none did).
static inline void walk_fastq_lines(const char* p, const char* e, bool& at_start, uint64_t& line_in_rec, const char*& rec_end) {
  bool partial = !at_start;  // p continues a line that began earlier: it is not blank
  while (p < e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    if (!nl) {
      at_start = false;
      return;
    }
    const bool blank = nl == p && !partial;
    partial = false;
    if (!(line_in_rec == 0 && blank) && ++line_in_rec == 4) {
      line_in_rec = 0;
      rec_end = nl + 1;
    }
    p = nl + 1;
    at_start = true;
  }
}

struct StageBlock {
  uint64_t* codes = nullptr;
  uint32_t *acgt = nullptr, *word_off = nullptr, *len = nullptr;
  bool pinned = true;  // false: made by a lazy allocator (no call into the device runtime), page-locked before its first upload
  uint32_t cap_reads = 0, n_reads = 0;
  uint64_t cap_words = 0, n_words = 0;
  int writers = 0;
  bool sealed = false;
};

class CountIngest {
  std::function<void(const StageBlock&)> sink_;
  unsigned nthreads_;
  void (*dealloc_)(void*);
  int (*pin_)(void*) = nullptr;  // late page-locking of the staging blocks (rfx_host_alloc_lazy + rfx_host_pin)
  void pin_block(StageBlock& b) {
    if (b.pinned || !pin_) return;
    (void)pin_(b.codes); (void)pin_(b.acgt); (void)pin_(b.word_off); (void)pin_(b.len);  // (a refusal: the upload is staged by the runtime)
    b.pinned = true;
  }
  std::vector<StageBlock> blocks_;
  std::mutex mu_;
  std::condition_variable cv_;  // one for every state change: pieces, blocks, failure
  struct Piece {
    const char* b;
    const char* e;
    std::vector<char>* owner;  // heap buffer of a pipe piece (returned to the pool), null otherwise
    int fd = -1;               // >= 0: a byte range [lo, hi) of a regular file -- the worker reads it itself and
    uint64_t lo = 0, hi = 0;   // parses the records that START inside the range
    uint64_t fsize = 0;
    uint64_t seq = 0;          // SAM mode: position of the piece in the stream (the chromosome log is stitched in order)
    uint64_t spool_off = 0;    // set_spool: where the piece's bytes go in the spool file
  };
  uint64_t stream_bytes_ = 0;  // feed_stream: bytes read so far
  int spool_fd_ = -1;          // set_spool: the stream is also written to this file, piece by piece, by the workers
  uint64_t spooled_ = 0;
  // set_keep_packed (`jellyfish count --sam .. --keep-packed FILE`): every SAM piece also leaves a chunk
  // of the packed-read cache `RUFUS.Filter --packed` scans (rfx_packed_cache.hpp) -- the records as the filter would
  // see them, packed once, here, by the threads that parse the text anyway
  int cache_fd_ = -1, cache_minq_ = 0;
  std::atomic<uint64_t> cache_at_{0};
  std::deque<Piece> work_;
  std::deque<int> ready_, free_;
  std::deque<std::vector<char>*> pool_;
  int current_ = -1;
  bool closing_ = false;
  size_t pieces_open_ = 0;
  std::atomic<bool> drop_mapped_{false};  // pieces without an owner are ranges of a file mapping (feed_mapped)
  std::atomic<bool> failed_{false};
  std::string fail_msg_;
  std::vector<std::thread> workers_;
  std::thread pinner_;
  uint64_t reads_total_ = 0;

  size_t PIECE = 32u << 20;  // bytes of text per piece (tests shrink it)
  // SAM mode (`jellyfish count --sam`): one record per line, the sequence is field 10; the runs of equal RNAME
  // (field 3) of every piece are kept, in stream order they are PassThroughSamCheck's chromosome log
  // (src/PassThroughSamCheck.cpp:140-155)
  bool sam_ = false;
  uint64_t next_seq_ = 0;
  std::map<uint64_t, std::vector<std::string>> chr_runs_;
  const skip_lines_fn skip_lines_ = pick_skip_lines();

  void fail(const std::string& m) {
    std::lock_guard<std::mutex> g(mu_);
    if (!failed_.exchange(true)) fail_msg_ = m;
    cv_.notify_all();
  }

  // Reserve n reads / w words in the current block (sealing it and taking a fresh one when they do not fit).
  bool reserve(uint32_t n, uint64_t w, int& blk, uint32_t& r0, uint64_t& w0) {
    std::unique_lock<std::mutex> g(mu_);
    for (;;) {
      if (failed_) return false;
      if (current_ >= 0) {
        StageBlock& b = blocks_[(size_t)current_];
        if (b.n_reads + (uint64_t)n <= b.cap_reads && b.n_words + w <= b.cap_words) {
          blk = current_;
          r0 = b.n_reads;
          w0 = b.n_words;
          b.n_reads += n;
          b.n_words += w;
          ++b.writers;
          return true;
        }
        b.sealed = true;
        if (b.writers == 0) {
          ready_.push_back(current_);
          cv_.notify_all();
        }
        current_ = -1;
      }
      if (n > blocks_[0].cap_reads || w > blocks_[0].cap_words) {
        g.unlock();
        fail("a read is longer than a staging block");
        return false;
      }
      // (another worker may install the next block while this one waits: look again before taking one)
      cv_.wait(g, [&] { return current_ >= 0 || !free_.empty() || failed_; });
      if (failed_) return false;
      if (current_ >= 0) continue;
      current_ = free_.front();
      free_.pop_front();
      StageBlock& b = blocks_[(size_t)current_];
      b.n_reads = 0;
      b.n_words = 0;
      b.writers = 0;
      b.sealed = false;
      cv_.notify_all();
    }
  }

  void release(int blk) {
    std::lock_guard<std::mutex> g(mu_);
    StageBlock& b = blocks_[(size_t)blk];
    if (--b.writers == 0 && b.sealed) {
      ready_.push_back(blk);
      cv_.notify_all();
    }
  }

  void parse_piece(const Piece& pc) {
    std::vector<uint64_t> start;
    std::vector<uint32_t> slen;
    start.reserve(1 << 17);
    slen.reserve(1 << 17);
    const char *p = pc.b, *e = pc.e;
    uint64_t words = 0;
    while (p < e) {
      if (*p == '\n') { ++p; continue; }  // blank lines between records are skipped (as the reference's parser does)
      if (*p != '@') return fail("parallel FASTQ reader: a record does not start with '@' (multi-line FASTQ? use RFX_HOST_THREADS=1)");
      const char* nl = find_nl(p, e);
      if (!nl) return fail("truncated FASTQ record");
      const char* s = nl + 1;
      nl = find_nl(s, e);
      if (!nl) return fail("truncated FASTQ record");
      const size_t L = (size_t)(nl - s);
      const char* plus = nl + 1;
      if (plus >= e || *plus != '+') return fail("parallel FASTQ reader: multi-line FASTQ records (use RFX_HOST_THREADS=1)");
      nl = plus + 1 < e && plus[1] == '\n' ? plus + 1 : find_nl(plus, e);  // almost always a bare "+"
      if (!nl) return fail("truncated FASTQ record");
      const char* q = nl + 1;
      // the quality line has to be as long as the sequence: look there first, search only if that is not its end
      const char* qe = (size_t)(e - q) > L && q[L] == '\n' && !memchr(q, '\n', L) ? q + L : find_nl(q, e);
      if (!qe) qe = e;  // last record of a file without a final newline
      if ((size_t)(qe - q) != L) return fail("parallel FASTQ reader: quality and sequence lengths differ (multi-line FASTQ? use RFX_HOST_THREADS=1)");
      start.push_back((uint64_t)(s - pc.b));
      slen.push_back((uint32_t)L);
      words += (L + 31) / 32;
      p = qe < e ? qe + 1 : e;
    }
    pack_spans_of(pc, start, slen);
    (void)words;
  }

  // SAM text (no header lines: `samtools view` without -h, as scripts/RunJellyForRUFUS.sh feeds it): what
  // PassThroughSamCheck would print as the sequence line of each record, packed in place.
  void parse_piece_sam(const Piece& pc) {
    std::vector<uint64_t> start;
    std::vector<uint32_t> slen;
    start.reserve(1 << 17);
    slen.reserve(1 << 17);
    std::vector<std::string> runs, cache_runs;
    rfxcache::ChunkBuilder cache;
    const char *p = pc.b, *e = pc.e;
    const char *cur_chr = nullptr, *cache_chr = nullptr;
    size_t cur_len = 0, cache_chr_len = 0;
    while (p < e) {
      const char* nl = find_nl(p, e);
      const char* le = nl ? nl : e;
      // tabs 2 and 3 bound RNAME, tabs 9 and 10 (or the end of the line) bound SEQ
      const char* tab[10];
      int nt = 0;
      for (const char* q = p; nt < 10 && q < le;) {
        const char* t = (const char*)memchr(q, '\t', (size_t)(le - q));
        if (!t) break;
        tab[nt++] = t;
        q = t + 1;
      }
      if (nt < 9) return fail("--sam: a line has fewer than 10 tab-separated fields (a header? feed `samtools view` without -h)");
      const char* chr = tab[1] + 1;
      const size_t chr_len = (size_t)(tab[2] - chr);
      if (!cur_chr || chr_len != cur_len || memcmp(chr, cur_chr, chr_len) != 0) {
        runs.emplace_back(chr, chr_len);
        cur_chr = chr;
        cur_len = chr_len;
      }
      const char* sq = tab[8] + 1;
      const char* sq_e = nt >= 10 ? tab[9] : le;
      start.push_back((uint64_t)(sq - pc.b));
      slen.push_back((uint32_t)(sq_e - sq));
      if (cache_fd_ >= 0 && nt >= 10) {  // (fewer than 11 fields: no record for the filter -- the feeders skip the line)
        // the filter's chromosome log lists the lines it is given (rufus_filter_main.cpp, split_sam): the cache keeps
        // runs of its own, of these lines only
        if (!cache_chr || chr_len != cache_chr_len || memcmp(chr, cache_chr, chr_len) != 0) {
          cache_runs.emplace_back(chr, chr_len);
          cache_chr = chr;
          cache_chr_len = chr_len;
        }
        const char* ql = tab[9] + 1;
        const char* ql_e = (const char*)memchr(ql, '\t', (size_t)(le - ql));
        if (!ql_e) ql_e = le;
        cache.add(pc.b, p, le, p, tab[0], tab[0] + 1, tab[1], sq, sq_e, ql, ql_e);
      }
      p = nl ? nl + 1 : e;
    }
    if (cache_fd_ >= 0) {
      std::vector<char> chunk;
      cache.finish(pc.seq, pc.spool_off, cache_runs, cache_minq_, chunk);
      const uint64_t at = cache_at_.fetch_add(chunk.size());
      const char* w = chunk.data();
      size_t len = chunk.size();
      off_t o = (off_t)at;
      while (len) {
        const ssize_t n = ::pwrite(cache_fd_, w, len, o);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) die(std::string("write error on the packed-read cache: ") + strerror(errno));
        w += n;
        len -= (size_t)n;
        o += n;
      }
    }
    {
      std::lock_guard<std::mutex> g(mu_);
      chr_runs_[pc.seq] = std::move(runs);
    }
    pack_spans_of(pc, start, slen);
  }

  void pack_spans_of(const Piece& pc, const std::vector<uint64_t>& start, const std::vector<uint32_t>& slen) {
    // normally one reservation per piece; a piece with more reads than a staging block holds goes in slices
    const size_t total = start.size();
    size_t at = 0;
    std::vector<uint32_t> woff_tmp;
    while (at < total) {
      uint32_t n = 0;
      uint64_t w = 0;
      while (at + n < total && n < blocks_[0].cap_reads / 2 + 1) {
        const uint64_t wr = (slen[at + n] + 31) / 32;
        if (n && w + wr > blocks_[0].cap_words / 2) break;
        w += wr;
        ++n;
      }
      int blk;
      uint32_t r0;
      uint64_t w0;
      if (!reserve(n, w, blk, r0, w0)) return;
      StageBlock& b = blocks_[(size_t)blk];
      // rfx_pack_spans writes n + 1 offsets, the last one being the neighbour range's first: it gets an array of this
      // thread's own, and only the range's n entries go into the block (two threads storing the same value into one
      // entry is still a data race -- ThreadSanitizer, tests/test_tsan_host.py); the entry behind the block's last
      // read is the consumer's to write
      woff_tmp.resize((size_t)n + 1);
      woff_tmp[0] = (uint32_t)w0;
      const int rc = rfx_pack_spans(pc.b, start.data() + at, slen.data() + at, nullptr, n, 0, RFX_PACK_COUNT, b.codes, b.acgt,
                                    nullptr, woff_tmp.data(), b.len + r0);
      if (rc) fail(std::string("rfx_pack_spans: ") + rfx_strerror(rc));
      memcpy(b.word_off + r0, woff_tmp.data(), (size_t)n * sizeof(uint32_t));
      release(blk);
      at += n;
    }
  }

  // A range of a regular file: read it (plus one byte before and the tail of the record that straddles its
  // end) into the worker's own buffer -- pread scales with the threads, faulting a 20 GB mapping page by page
  // does not -- and parse the records that start inside [lo, hi).
  void parse_range(const Piece& pc, std::vector<char>& buf) {
    const uint64_t SLACK = 1u << 20;  // longest record the tail may hold
    const uint64_t from = pc.lo ? pc.lo - 1 : 0, to = std::min<uint64_t>(pc.fsize, pc.hi + SLACK);
    buf.resize((size_t)(to - from));
    size_t got = 0;
    while (got < buf.size()) {
      const ssize_t n = ::pread(pc.fd, buf.data() + got, buf.size() - got, (off_t)(from + got));
      if (n < 0 && errno == EINTR) continue;
      if (n <= 0) return fail(std::string("read error on input: ") + strerror(errno));
      got += (size_t)n;
    }
    const char *b = buf.data(), *e = buf.data() + buf.size();
    const char* s0 = pc.lo ? record_start(b + 1, b, e) : b;                       // first record starting at >= lo
    const char* hi_p = b + (pc.hi - from);
    const char* s1 = pc.hi >= pc.fsize ? e : record_start(hi_p, b, e);             // first record starting at >= hi
    if (pc.hi < pc.fsize && s1 == e && to < pc.fsize) return fail("a FASTQ record is longer than 1 MB");
    if (s0 >= s1) return;
    parse_piece(Piece{s0, s1, nullptr});
  }

  void worker() {
    std::vector<char> buf;
    for (;;) {
      Piece pc;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return !work_.empty() || closing_ || failed_; });
        if (failed_ || (work_.empty() && closing_)) return;
        pc = work_.front();
        work_.pop_front();
      }
      if (pc.fd >= 0) parse_range(pc, buf);
      else if (sam_) parse_piece_sam(pc);
      else parse_piece(pc);
      if (pc.fd < 0 && !pc.owner && drop_mapped_) drop_mapped(pc.b, pc.e);  // a piece of feed_mapped()'s mapping
      if (spool_fd_ >= 0 && pc.fd < 0 && pc.owner) {  // a piece of a pipe: its bytes, at their place in the stream
        const char* p = pc.b;
        size_t len = (size_t)(pc.e - pc.b);
        off_t at = (off_t)pc.spool_off;
        while (len) {
          const ssize_t w = ::pwrite(spool_fd_, p, len, at);
          if (w < 0 && errno == EINTR) continue;
          if (w <= 0) die(std::string("write error on the spool file: ") + strerror(errno));
          p += w;
          len -= (size_t)w;
          at += w;
        }
      }
      {
        std::lock_guard<std::mutex> g(mu_);
        if (pc.owner) pool_.push_back(pc.owner);
        --pieces_open_;
        cv_.notify_all();
      }
    }
  }

  void push_piece(const Piece& pc) {
    std::lock_guard<std::mutex> g(mu_);
    work_.push_back(pc);
    ++pieces_open_;
    cv_.notify_all();
  }

  // Upload every block that is ready; with `all`, wait until every queued piece is packed and seal the last block.
  void drain(bool all) {
    for (;;) {
      int blk = -1;
      {
        std::unique_lock<std::mutex> g(mu_);
        if (all) {
          cv_.wait(g, [&] { return !ready_.empty() || pieces_open_ == 0 || failed_; });
          if (failed_) break;
          if (ready_.empty() && pieces_open_ == 0) {
            if (current_ >= 0) {  // the partly filled last block
              blocks_[(size_t)current_].sealed = true;
              ready_.push_back(current_);
              current_ = -1;
            } else {
              break;
            }
          }
        }
        if (ready_.empty()) break;
        blk = ready_.front();
        ready_.pop_front();
      }
      StageBlock& b = blocks_[(size_t)blk];
      if (b.n_reads) {
        b.word_off[b.n_reads] = (uint32_t)b.n_words;
        reads_total_ += b.n_reads;
        pin_block(b);
        sink_(b);  // uploads it (rfx_reads_upload): the block is reused as soon as this returns
      }
      {
        std::lock_guard<std::mutex> g(mu_);
        free_.push_back(blk);
        cv_.notify_all();
      }
    }
    if (failed_) die("rufus_amd jellyfish: " + fail_msg_);
  }

  friend class TextIngest;
  // First position >= p where a 4-line FASTQ record starts ('@' line whose third line starts with '+' and whose
  // second and fourth lines are equally long); e when there is none.
  static const char* record_start(const char* p, const char* b, const char* e) {
    if (p <= b) return b;
    const char* nl = (const char*)memchr(p - 1, '\n', (size_t)(e - (p - 1)));
    if (!nl) return e;
    const char* l0 = nl + 1;
    for (int tries = 0; tries < 8 && l0 < e; ++tries) {
      const char* n0 = (const char*)memchr(l0, '\n', (size_t)(e - l0));
      if (!n0) return e;
      const char* l1 = n0 + 1;
      const char* n1 = l1 < e ? (const char*)memchr(l1, '\n', (size_t)(e - l1)) : nullptr;
      const char* l2 = n1 ? n1 + 1 : e;
      const char* n2 = l2 < e ? (const char*)memchr(l2, '\n', (size_t)(e - l2)) : nullptr;
      const char* l3 = n2 ? n2 + 1 : e;
      const char* n3 = l3 < e ? (const char*)memchr(l3, '\n', (size_t)(e - l3)) : nullptr;
      const char* l3e = n3 ? n3 : e;
      if (*l0 == '@' && n1 && n2 && *l2 == '+' && (n1 - l1) == (l3e - l3)) return l0;
      l0 = l1;
    }
    return e;
  }

 public:
  // alloc / dealloc: page-locked memory on the GPU box (rfx_host_alloc), plain malloc in host-only tests
  // pin != nullptr: `alloc` makes plain memory without touching the device runtime (rfx_host_alloc_lazy) -- the workers
  // parse into it while another thread is still opening the device -- and `pin` page-locks a block before its first upload
  CountIngest(unsigned threads, std::function<void(const StageBlock&)> sink, void* (*alloc)(size_t) = rfx_host_alloc,
              void (*dealloc)(void*) = rfx_host_free, uint32_t cap_reads = 4u << 20, uint64_t cap_words = 24ull << 20,
              int (*pin_fn)(void*) = nullptr)
      : sink_(std::move(sink)), nthreads_(threads ? threads : 1), dealloc_(dealloc), pin_(pin_fn) {
    blocks_.resize(3);
    for (StageBlock& b : blocks_) {
      b.cap_reads = cap_reads;
      b.cap_words = cap_words;
      b.pinned = pin_fn == nullptr;
    }
    // Page-locking ~1 GB takes 0.2 s: the workers start on the first block while the others are still being pinned.
    auto pin = [this, alloc](size_t i) {
      StageBlock& b = blocks_[i];
      b.codes = (uint64_t*)alloc(b.cap_words * 8);
      b.acgt = (uint32_t*)alloc(b.cap_words * 4);
      b.word_off = (uint32_t*)alloc(((size_t)b.cap_reads + 1) * 4);
      b.len = (uint32_t*)alloc((size_t)b.cap_reads * 4);
      if (!b.codes || !b.acgt || !b.word_off || !b.len) die("rufus_amd: cannot allocate pinned staging memory");
      std::lock_guard<std::mutex> g(mu_);
      free_.push_back((int)i);
      cv_.notify_all();
    };
    pin(0);
    pinner_ = std::thread([this, pin] {
      for (size_t i = 1; i < blocks_.size(); ++i) pin(i);
    });
    for (unsigned t = 0; t < nthreads_; ++t) workers_.emplace_back([this] { worker(); });
  }

  // The staging memory cut into page-locked pieces of `piece` bytes, for a caller that is done feeding (the count
  // tool drains its result through them).  Valid until this object is destroyed.
  std::vector<std::pair<char*, size_t>> lend_buffers(size_t piece) {
    if (pinner_.joinable()) pinner_.join();
    std::vector<std::pair<char*, size_t>> out;
    for (StageBlock& b : blocks_) {
      pin_block(b);
      for (size_t at = 0; at + piece <= b.cap_words * 8; at += piece) out.emplace_back((char*)b.codes + at, piece);
      for (size_t at = 0; at + piece <= b.cap_words * 4; at += piece) out.emplace_back((char*)b.acgt + at, piece);
    }
    return out;
  }

  ~CountIngest() {
    if (pinner_.joinable()) pinner_.join();
    {
      std::lock_guard<std::mutex> g(mu_);
      closing_ = true;
      cv_.notify_all();
    }
    for (auto& t : workers_) t.join();
    for (auto& b : blocks_) {
      dealloc_(b.codes);
      dealloc_(b.acgt);
      dealloc_(b.word_off);
      dealloc_(b.len);
    }
    for (auto* v : pool_) delete v;
  }

  uint64_t reads() const { return reads_total_; }
  void set_piece_bytes(size_t n) { PIECE = n; }

  // Does the stream look like strict 4-line FASTQ?  (head: its first bytes)
  static bool looks_4line(const char* head, size_t n) {
    if (n == 0 || head[0] != '@') return false;
    const char* e = head + n;
    const char* r = record_start(head, head, e);
    if (r != head) return false;
    // the second record (when the head is long enough to hold it) must follow immediately
    const char* p = head;
    for (int i = 0; i < 4; ++i) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
      if (!nl) return true;
      p = nl + 1;
    }
    return p >= e || *p == '@' || *p == '\n';
  }

  // A whole regular file, mapped.  Returns false if it is not strict 4-line FASTQ (nothing consumed).
  void set_sam(bool on) { sam_ = on; }
  // The bytes of a PIPE input are also written to `fd` (a regular file), so that the stage that reads the same stream
  // next -- RUFUS.Filter after the subject's count, runRufus.sh:966 after scripts/RunJellyForRUFUS.sh:28 -- need not run
  // the generator (samtools view of a BAM) a second time.  Written piece by piece by the parser threads (pwrite).
  void set_spool(int fd) { spool_fd_ = fd; }
  // feed_stream(): called by the reading thread with the number of bytes the stream has delivered so far
  std::function<void(uint64_t)> on_stream_bytes;
  void set_keep_packed(int fd, int min_q) {
    cache_fd_ = fd;
    cache_minq_ = min_q;
    rfxcache::FileHeader h;
    h.min_q = min_q;
    if (::pwrite(fd, &h, sizeof h, 0) != (ssize_t)sizeof h) die("write error on the packed-read cache");
    cache_at_ = sizeof h;
  }
  // after the last feed_*(): the header once more, now with what makes the file a cache (rfx_packed_cache.hpp)
  void finish_keep_packed(uint64_t stream_bytes) {
    if (cache_fd_ < 0) return;
    rfxcache::FileHeader h;
    h.min_q = cache_minq_;
    h.n_chunks = next_seq_;
    h.stream_bytes = stream_bytes;
    h.done = rfxcache::DONE_MAGIC;
    if (::pwrite(cache_fd_, &h, sizeof h, 0) != (ssize_t)sizeof h || ::close(cache_fd_) != 0)
      die(std::string("write error on the packed-read cache: ") + strerror(errno));
    cache_fd_ = -1;
  }
  uint64_t spooled_bytes() const { return spooled_; }
  // PassThroughSamCheck's side file: "notachr", then the name of every run of equal RNAME, in stream order
  std::vector<std::string> chr_log() {
    std::vector<std::string> out{"notachr"};
    for (auto& kv : chr_runs_)
      for (auto& name : kv.second)
        if (out.size() == 1 || out.back() != name) out.push_back(name);
    return out;
  }

  // `data` must be a read-only FILE mapping: the workers drop the page-table entries of what they have parsed
  // (rfx_cli.hpp drop_mapped -- on anonymous memory that would zero it).
  bool feed_mapped(const char* data, size_t size) {
    if (!sam_ && !looks_4line(data, std::min<size_t>(size, 1u << 16))) return false;
    drop_mapped_ = true;
    const char *b = data, *e = data + size;
    const char* at = b;
    while (at < e) {
      const char* want = at + PIECE;
      const char* cut;
      if (want >= e) cut = e;
      else if (sam_) {
        const char* nl = (const char*)memchr(want, '\n', (size_t)(e - want));
        cut = nl ? nl + 1 : e;
      } else cut = record_start(want, b, e);
      Piece pc{at, cut, nullptr};
      pc.seq = next_seq_++;
      pc.spool_off = (uint64_t)(at - b);  // (a regular file is its own spool)
      push_piece(pc);
      at = cut;
      drain(false);
    }
    drain(true);
    return true;
  }

  // A whole regular file by descriptor: byte ranges, read by the workers themselves.  Returns false if the file
  // does not start like strict 4-line FASTQ (nothing consumed).
  bool feed_file(int fd, uint64_t size) {
    std::vector<char> head((size_t)std::min<uint64_t>(size, 1u << 16));
    size_t got = 0;
    while (got < head.size()) {
      const ssize_t n = ::pread(fd, head.data() + got, head.size() - got, (off_t)got);
      if (n < 0 && errno == EINTR) continue;
      if (n <= 0) break;
      got += (size_t)n;
    }
    if (!looks_4line(head.data(), got)) return false;
    for (uint64_t lo = 0; lo < size; lo += PIECE) {
      Piece pc{nullptr, nullptr, nullptr};
      pc.fd = fd;
      pc.lo = lo;
      pc.hi = std::min<uint64_t>(size, lo + PIECE);
      pc.fsize = size;
      push_piece(pc);
      drain(false);
    }
    drain(true);
    return true;
  }

  // A pipe: `head` = bytes already read from it (at least the sniffed prefix).  One reader (this thread) cuts the
  // stream every 4k lines.  Returns false if the head is not strict 4-line FASTQ (the caller then parses
  // head + rest sequentially).
  bool feed_stream(int fd, std::vector<char>& head) {
    if (!sam_ && !looks_4line(head.data(), head.size())) return false;
    const uint64_t lines_per_rec = sam_ ? 1 : 4;
#ifdef F_SETPIPE_SZ
    (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);  // a pipe: fewer, larger reads (ignored on anything else)
#endif
    std::vector<char>* buf = nullptr;
    size_t fill = 0;
    uint64_t line_in_rec = 0;  // lines of the current record already inside buf[0, scanned)
    size_t scanned = 0, last_cut = 0;
    // FASTQ: blank lines do not count (parse_piece skips them between records, as the mapped and pread routes do);
    // line_start = the byte at `scanned` begins a line
    bool line_start = true;
    auto fresh = [&](size_t at_least) {
      // bound the text in flight (2 x workers pieces); this thread is also the uploader, so full blocks are
      // uploaded while it waits -- the workers may be waiting for exactly that
      std::vector<char>* v = nullptr;
      for (;;) {
        {
          std::unique_lock<std::mutex> g(mu_);
          cv_.wait(g, [&] { return !ready_.empty() || !pool_.empty() || pieces_open_ < 2 * (size_t)nthreads_ + 2 || failed_; });
          if (failed_) break;
          if (ready_.empty()) {
            if (!pool_.empty()) {
              v = pool_.front();
              pool_.pop_front();
            }
            break;
          }
        }
        drain(false);
      }
      if (!v) v = new std::vector<char>(PIECE + std::max<size_t>(PIECE / 8, 1u << 16));
      if (v->size() < at_least + PIECE) v->resize(at_least + PIECE + std::max<size_t>(PIECE / 8, 1u << 16));
      return v;
    };
    buf = fresh(head.size());
    memcpy(buf->data(), head.data(), head.size());
    fill = head.size();
    bool eof = false, need_more = false;
    while (!eof || fill > 0) {
      if (failed_) break;
      // read until the buffer holds a piece (and, after a pass that found no record end, something new)
      while (!eof && (fill < PIECE || need_more)) {
        if (fill == buf->size()) {  // one record longer than the buffer: grow it (a real limit ends the run)
          if (buf->size() >= ((size_t)1 << 31)) die("a FASTQ record is longer than 2 GiB");
          buf->resize(buf->size() * 2);
        }
        const ssize_t n = ::read(fd, buf->data() + fill, buf->size() - fill);
        if (n < 0) {
          if (errno == EINTR) continue;
          die(std::string("read error on input: ") + strerror(errno));
        }
        if (n == 0) eof = true;
        else fill += (size_t)n;
        if (n > 0 && on_stream_bytes) on_stream_bytes(stream_bytes_ += (uint64_t)n);
        need_more = false;
      }
      // last record boundary inside [0, fill): count the new lines (32 bytes per step), then walk to the newline
      // that completes the last whole record
      const char* d = buf->data();
      size_t got = 0, blanks = 0;
      bool st = line_start;
      count_newlines(d + scanned, d + fill, st, got, blanks);
      if (sam_ || blanks == 0) {
        const uint64_t total_lines = line_in_rec + got, recs = total_lines / lines_per_rec;
        if (recs) {
          size_t g2 = 0;
          const char* cut_at = skip_lines_(d + scanned, d + fill, (size_t)(recs * lines_per_rec - line_in_rec), g2);
          last_cut = (size_t)(cut_at - d);
        }
        line_in_rec = total_lines - recs * lines_per_rec;
      } else {  // blank lines: between records they do not count (parse_piece skips them there)
        st = line_start;
        const char* rec_end = nullptr;
        walk_fastq_lines(d + scanned, d + fill, st, line_in_rec, rec_end);
        if (rec_end) last_cut = (size_t)(rec_end - d);
      }
      scanned = fill;
      line_start = st;
      size_t cut = eof ? fill : last_cut;
      if (cut == 0) {
        if (eof) break;
        need_more = true;  // not one whole record yet: keep reading into the same buffer
        continue;
      }
      const size_t rest = fill - cut;
      std::vector<char>* next = fresh(rest);
      memcpy(next->data(), d + cut, rest);
      {
        Piece pc{d, d + cut, buf};
        pc.seq = next_seq_++;
        pc.spool_off = spooled_;
        spooled_ += cut;
        push_piece(pc);
      }
      buf = next;
      fill = rest;
      scanned = rest;  // the carried bytes hold line_in_rec complete lines of an unfinished record (already counted)
      last_cut = 0;
      drain(false);
      if (eof && fill == 0) break;
    }
    {
      std::lock_guard<std::mutex> g(mu_);
      pool_.push_back(buf);
    }
    drain(true);
    return true;
  }
};

// ---- text to the device, parsed there (round 6: SURVEY section 2, kernel K1; rufus_amd/csrc/rfx_text.hip) --------------
// For a regular file of strict 4-line FASTQ counted on one device the host does not parse at all: the mapped text is cut
// into record-aligned pieces (at the cut points only: CountIngest::record_start), the workers copy each piece into
// page-locked memory and queue its host-to-device copy behind the others of the arena that is being filled
// (rfx_text_append: the arena's own stream), and this thread turns every full arena into a read block
// (rfx_text_parse) and hands it to the sink -- while the workers fill the other arena.  What the 16-CPU quota of the
// GPU box spends per read drops from ~1 us of parsing and packing to a memcpy of 330 bytes; the text crosses PCIe at
// ~55 GB/s where the packed blocks of the host route left it idle most of the time.  Text the device refuses (not
// strict 4-line FASTQ after all: a blank line between records, a multi-line record) comes back and goes through the
// caller's host parser, arena by arena: the records of an arena are whole, so nothing is lost or seen twice.
class TextIngest {
  rfx_ctx* ctx_;
  unsigned nthreads_;
  std::function<void(rfx_reads*)> sink_;
  std::function<void(const char*, size_t)> host_text_;
  size_t piece_;
  struct Arena {
    rfx_text* t = nullptr;
    uint64_t gen = 0;  // bumped at every reset: a worker's ticket of an earlier generation is a copy long done
    int state = 0;     // 0 free, 1 filling, 2 full (waiting for this thread)
  };
  Arena arena_[2];
  int filling_ = -1;
  std::deque<int> full_;
  struct Piece {
    const char* b;
    const char* e;
    int fd = -1;  // >= 0: bytes [lo, hi) of a regular file -- the worker reads them itself (pread into its page-locked
    uint64_t lo = 0, hi = 0, fsize = 0;  // buffer) and appends the records that START inside the range
  };
  std::deque<Piece> work_;
  size_t pieces_open_ = 0;
  bool closing_ = false;
  std::atomic<bool> failed_{false};
  std::string fail_msg_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<std::thread> workers_;
  uint64_t reads_ = 0, host_bytes_ = 0;
  static constexpr uint64_t TAIL = 1u << 20;  // longest record a range's tail may hold (feed_file)
  double t_parse_ = 0, t_sink_ = 0;  // this thread: inside rfx_text_parse (it waits for the arena's copies) / in the sink
  unsigned n_arenas_ = 0;
  // the workers, summed (ns): copying a piece into page-locked memory, waiting for the lock + an arena with room, inside
  // rfx_text_append, waiting for the copy that last read the buffer
  std::atomic<uint64_t> w_memcpy_{0}, w_room_{0}, w_append_{0}, w_settle_{0}, w_idle_{0};
  static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  void fail(const std::string& m) {
    std::lock_guard<std::mutex> g(mu_);
    if (!failed_.exchange(true)) fail_msg_ = m;
    cv_.notify_all();
  }

  void worker() {
    struct Buf { char* p = nullptr; int arena = -1; uint64_t gen = 0; long ticket = -1; } buf[2];
    const size_t cap = piece_ + TAIL + (1u << 20);
    for (Buf& b : buf) {
      b.p = (char*)rfx_host_alloc(cap);
      if (!b.p) return fail("cannot allocate page-locked text buffers");
    }
    // true: the copy that last read the buffer is done (or belonged to an arena that has been parsed since)
    auto settle = [&](Buf& b) -> bool {
      if (b.ticket < 0) return true;
      auto same_gen = [&] {
        std::lock_guard<std::mutex> g(mu_);
        return arena_[b.arena].gen == b.gen;
      };
      if (same_gen() && rfx_text_wait(arena_[b.arena].t, b.ticket) != RFX_OK && same_gen()) {
        fail(std::string("rfx_text_wait: ") + rfx_last_error());
        return false;
      }
      b.ticket = -1;
      return true;
    };
    int j = 0;
    for (;;) {
      Piece pc;
      {
        const uint64_t t0 = now_ns();
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return !work_.empty() || closing_ || failed_; });
        if (failed_ || (work_.empty() && closing_)) break;
        pc = work_.front();
        work_.pop_front();
        w_idle_ += now_ns() - t0;
      }
      Buf& b = buf[j];
      j ^= 1;
      const uint64_t ta = now_ns();
      if (!settle(b)) break;  // (the copy that last read this buffer)
      const uint64_t tb = now_ns();
      size_t n;
      char* src = b.p;  // what is appended: [src, src + n)
      if (pc.fd >= 0) {
        // one byte before the range (is `lo` a line start?) and, behind it, the tail of the record that straddles `hi`
        const uint64_t from = pc.lo ? pc.lo - 1 : 0, to = std::min<uint64_t>(pc.fsize, pc.hi + TAIL);
        size_t got = 0;
        const size_t want = (size_t)(to - from);
        while (got < want) {
          const ssize_t r = ::pread(pc.fd, b.p + got, want - got, (off_t)(from + got));
          if (r < 0 && errno == EINTR) continue;
          if (r <= 0) break;
          got += (size_t)r;
        }
        if (got != want) { fail(std::string("read error on input: ") + strerror(errno)); break; }
        const char *tb = b.p, *te = b.p + want;
        const char* s0 = pc.lo ? CountIngest::record_start(tb + 1, tb, te) : tb;
        const char* s1 = pc.hi >= pc.fsize ? te : CountIngest::record_start(tb + (pc.hi - from), tb, te);
        if (pc.hi < pc.fsize && s1 == te && to < pc.fsize) { fail("a FASTQ record is longer than 1 MB"); break; }
        if (s0 > s1) s0 = s1;
        src = const_cast<char*>(s0);
        n = (size_t)(s1 - s0);
        if (n && src[n - 1] != '\n' && pc.hi >= pc.fsize) src[n++] = '\n';  // (a file without a final newline)
      } else {
        n = (size_t)(pc.e - pc.b);
        if (n + 1 > cap) { fail("a FASTQ record is longer than a text buffer"); break; }
        memcpy(b.p, pc.b, n);
        if (n && b.p[n - 1] != '\n') b.p[n++] = '\n';  // (the last line of a file without a final newline)
      }
      const uint64_t tc = now_ns();
      w_settle_ += tb - ta;
      w_memcpy_ += tc - tb;
      {
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
          if (failed_) break;
          if (filling_ >= 0 && rfx_text_room(arena_[filling_].t) >= n) break;
          if (filling_ >= 0) {  // no room: this arena is this thread's ... the main thread's now
            arena_[filling_].state = 2;
            full_.push_back(filling_);
            filling_ = -1;
            cv_.notify_all();
          }
          int f = -1;
          for (int a = 0; a < 2; ++a)
            if (arena_[a].state == 0) { f = a; break; }
          if (f >= 0) {
            arena_[f].state = 1;
            filling_ = f;
            continue;
          }
          cv_.wait(g);
        }
        if (failed_) break;
        const uint64_t td = now_ns();
        w_room_ += td - tc;
        const long tk = n ? rfx_text_append(arena_[filling_].t, src, n) : rfx_text_append(arena_[filling_].t, b.p, 0);
        w_append_ += now_ns() - td;
        if (tk < 0) {
          g.unlock();
          fail(std::string("rfx_text_append: ") + rfx_strerror((int)tk) + " " + rfx_last_error());
          break;
        }
        b.arena = filling_;
        b.gen = arena_[filling_].gen;
        b.ticket = tk;
        --pieces_open_;
        cv_.notify_all();
      }
    }
    // (the buffers outlive their last copies: wait for them)
    for (Buf& b : buf) {
      (void)settle(b);
      rfx_host_free(b.p);
    }
  }

  void finish_arena(int a) {  // this thread only: parse, hand on, give the arena back
    rfx_text* t = arena_[a].t;
    if (rfx_text_bytes(t)) {
      int strict = 1;
      const auto t0 = std::chrono::steady_clock::now();
      rfx_reads* r = rfx_text_parse(t, RFX_PACK_COUNT, 0, &strict);
      const auto t1 = std::chrono::steady_clock::now();
      t_parse_ += std::chrono::duration<double>(t1 - t0).count();
      ++n_arenas_;
      if (r) {
        reads_ += rfx_reads_count(r);
        sink_(r);
        t_sink_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
      } else if (!strict) {
        std::vector<char> text((size_t)rfx_text_bytes(t));
        if (rfx_text_fetch(t, text.data()) != RFX_OK) die(std::string("rufus_amd: rfx_text_fetch: ") + rfx_last_error());
        host_bytes_ += text.size();
        host_text_(text.data(), text.size());
      } else {
        die(std::string("rufus_amd: rfx_text_parse: ") + rfx_last_error());
      }
    }
    {  // (the generation first: a worker that finds its ticket gone looks at it again before it calls that a failure)
      std::lock_guard<std::mutex> g(mu_);
      ++arena_[a].gen;
    }
    rfx_text_reset(t);
    std::lock_guard<std::mutex> g(mu_);
    arena_[a].state = 0;
    cv_.notify_all();
  }

  // the full arenas that wait; all: until every piece is copied, then the partly filled arena too
  void drain(bool all) {
    for (;;) {
      int a = -1;
      {
        std::unique_lock<std::mutex> g(mu_);
        if (all) cv_.wait(g, [&] { return !full_.empty() || pieces_open_ == 0 || failed_; });
        if (failed_) break;
        if (!full_.empty()) {
          a = full_.front();
          full_.pop_front();
        } else if (all && pieces_open_ == 0) {
          if (filling_ < 0) break;
          a = filling_;
          arena_[a].state = 2;
          filling_ = -1;
        } else {
          break;
        }
      }
      finish_arena(a);
    }
    if (failed_) die("rufus_amd jellyfish: " + fail_msg_);
  }

 public:
  TextIngest(rfx_ctx* ctx, unsigned threads, std::function<void(rfx_reads*)> sink,
             std::function<void(const char*, size_t)> host_text, size_t arena_bytes = (size_t)1 << 30, size_t piece = 8u << 20)
      : ctx_(ctx), nthreads_(threads ? threads : 1), sink_(std::move(sink)), host_text_(std::move(host_text)), piece_(piece) {
    for (Arena& a : arena_) {
      a.t = rfx_text_open(ctx_, arena_bytes);
      if (!a.t) die(std::string("rufus_amd: rfx_text_open: ") + rfx_last_error());
    }
    for (unsigned t = 0; t < nthreads_; ++t) workers_.emplace_back([this] { worker(); });
  }
  ~TextIngest() {
    {
      std::lock_guard<std::mutex> g(mu_);
      closing_ = true;
      cv_.notify_all();
    }
    for (auto& t : workers_) t.join();
    for (Arena& a : arena_) rfx_text_close(a.t);
  }
  uint64_t reads() const { return reads_; }
  uint64_t host_parsed_bytes() const { return host_bytes_; }
  std::string timing() const {
    char b[320];
    snprintf(b, sizeof b, "%u arenas: %.3f s waiting for copies + parsing on the device, %.3f s in the sink (count); workers, "
             "summed: memcpy %.3f s, lock + room %.3f s, rfx_text_append %.3f s, waiting for a buffer's copy %.3f s, for a piece %.3f s", n_arenas_,
             t_parse_, t_sink_, w_memcpy_.load() * 1e-9, w_room_.load() * 1e-9, w_append_.load() * 1e-9, w_settle_.load() * 1e-9,
             w_idle_.load() * 1e-9);
    return b;
  }

  // A whole regular file by descriptor: byte ranges that the workers read themselves, straight into their page-locked
  // buffers -- no mapping whose 4 KB pages have to be faulted in one by one and, worse, torn down again by ONE thread
  // (munmap of a 20 GB mapping: 0.3 s of a count that takes 1.5).  false: the file does not start like strict 4-line
  // FASTQ (nothing consumed).
  bool feed_file(int fd, uint64_t size) {
    std::vector<char> head((size_t)std::min<uint64_t>(size, 1u << 16));
    size_t got = 0;
    while (got < head.size()) {
      const ssize_t n = ::pread(fd, head.data() + got, head.size() - got, (off_t)got);
      if (n < 0 && errno == EINTR) continue;
      if (n <= 0) break;
      got += (size_t)n;
    }
    if (!CountIngest::looks_4line(head.data(), got)) return false;
    for (uint64_t lo = 0; lo < size; lo += piece_) {
      Piece pc{nullptr, nullptr};
      pc.fd = fd;
      pc.lo = lo;
      pc.hi = std::min<uint64_t>(size, lo + piece_);
      pc.fsize = size;
      {
        std::lock_guard<std::mutex> g(mu_);
        work_.push_back(pc);
        ++pieces_open_;
        cv_.notify_all();
      }
      drain(false);
    }
    drain(true);
    return true;
  }

  // A whole regular file, mapped.  false: it does not start like strict 4-line FASTQ (nothing consumed).
  bool feed_mapped(const char* data, size_t size) {
    if (!CountIngest::looks_4line(data, std::min<size_t>(size, 1u << 16))) return false;
    const char *b = data, *e = data + size, *at = b;
    while (at < e) {
      const char* want = at + piece_;
      const char* cut = want >= e ? e : CountIngest::record_start(want, b, e);
      if (cut == e && want < e && (size_t)(e - at) > piece_ + (1u << 20)) {
        // no record start within reach: not the format this route is for -- what is left goes to the host parser whole
        drain(true);
        host_bytes_ += (size_t)(e - at);
        host_text_(at, (size_t)(e - at));
        return true;
      }
      {
        std::lock_guard<std::mutex> g(mu_);
        work_.push_back(Piece{at, cut});
        ++pieces_open_;
        cv_.notify_all();
      }
      at = cut;
      drain(false);
    }
    drain(true);
    return true;
  }
};

}  // namespace rfxcli
