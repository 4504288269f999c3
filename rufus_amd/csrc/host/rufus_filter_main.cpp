// Drop-in RUFUS.Filter (paired) and RUFUS.Filter.single (-DRFX_SINGLE_END), same argv and outputs:
//   RUFUS.Filter        HashList Mate1.fq Mate2.fq STUB K MinQ HashCountThreshold Threads   (runRufus.sh:967)
//       -> STUB.Mutations.Mate1.fastq, STUB.Mutations.Mate2.fastq        src/RUFUS.Filter.cpp:28,:47-54
//   RUFUS.Filter.single HashList FQ|stdin STUB K MinQ HashCountThreshold Threads
//       -> STUB.Mutations.fastq with ":MH<hits>" appended to each header  src/RUFUS.Filter.ss.cpp:27,:43-49,:198
// The two mate files may be named pipes written in lock step by PassThroughSamCheck.stranded, so
// they are read interleaved, 4 lines each (src/RUFUS.Filter.cpp:165-173) -- never one file ahead.
// Pulled records are written in input order (the reference's order depends on OpenMP scheduling; at
// one thread it is input order too).  The scan itself runs in k_filter; no CPU fallback.
#include <condition_variable>
#include <deque>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>

#include <sys/mman.h>
#include <sys/stat.h>

#include "rfx_cli.hpp"

using namespace rfxcli;

#ifndef RFX_SINGLE_END
#include <unordered_set>

#include "rfx_sam.hpp"
#include "rfx_packed_cache.hpp"
// ---- RUFUS.Filter --sam CHRFILE HashList SAM|stdin STUB K MinQ HashCountThreshold Threads -----------------------------
// SURVEY row N1, filter half.  runRufus.sh:964-967 runs `generator | PassThroughSamCheck.stranded CHR STUB.temp` into two
// named pipes that RUFUS.Filter reads in lock step: SAM text -> pairing by QNAME -> FASTQ text -> two pipes -> parsed
// again -> packed.  Here the SAM stream is the filter's own input: helper threads find the fields of every line, put
// reverse-strand records back the way the feeder prints them (src/PassThroughSamCheck.stranded.cpp:188-223: reverse
// complement, bases other than ACGTN vanish, quality reversed), pack bases + quality mask and run the scan on EVERY
// record by itself (k_filter; `i < length()-1` as the paired tool, src/RUFUS.Filter.cpp:203); one thread then walks
// the pieces in stream order, writes the chromosome log (CHRFILE: "notachr", then the previous RNAME at every change)
// and pairs the records by name as the feeder does -- a record meets its waiting mate or waits -- and a pair one of
// whose records was hit is written: the LATER record to STUB.Mutations.Mate1.fastq, the stored one to Mate2, in the
// order the pairs complete.  The same bytes as the two-process route; no FASTQ text exists for the other pairs.
// A waiting record is a pointer into its piece (no copy); a piece is recycled LAG pieces later, and only the few
// records that still wait then (mates far apart, unpaired reads) are copied out.
namespace samf {
using namespace rfxsam;

struct Rec {
  const char *name, *seq, *qual;
  uint32_t name_len, seq_len, qual_len;
  uint64_t hash;
};
// Where the printed forms of a piece's records go (reverse strand: bases and qualities again; a quality string shorter
// than its read: a padded copy): blocks taken as they are needed and kept for the next piece -- most records are
// forward-strand and need none.  (Until round 4 every piece buffer had room for the worst case behind its lines,
// zero-filled by vector::resize: 96 MB touched per 32 MB piece, 8 GB at Threads = 40.)
struct SideArena {
  static constexpr size_t BLOCK = (size_t)4 << 20;
  std::vector<std::pair<std::unique_ptr<char[]>, size_t>> blocks;
  size_t cur = 0, used = 0;
  void reset() { cur = used = 0; }
  char* take(size_t n) {  // n bytes that stay where they are until the next reset()
    while (cur < blocks.size() && used + n > blocks[cur].second) {
      ++cur;
      used = 0;
    }
    if (cur == blocks.size()) {
      const size_t cap = std::max(BLOCK, n);
      blocks.emplace_back(std::unique_ptr<char[]>(new char[cap]), cap);
      used = 0;
    }
    char* p = blocks[cur].first.get() + used;
    used += n;
    return p;
  }
};
struct SPiece {
  std::vector<char> text;  // [0, size): SAM lines (pipe route)
  size_t size = 0;
  const char* mapped = nullptr;  // a regular input file: the lines lie in its mapping
  SideArena side;               // the printed form of reverse-strand records
  std::vector<Rec> recs;
  std::vector<std::string> chr_runs;
  std::vector<uint64_t> mask;  // hit bit per record
  std::vector<uint32_t> waited;  // records of this piece that went into the waiting table
  uint64_t seq = 0;
};
struct Owned {
  Rec r;
  std::string bytes;
};
struct Slot {
  uint64_t hash = 0;
  const Rec* rec = nullptr;
  Owned* owned = nullptr;
  uint32_t state = 0;  // 0 empty, 1 deleted, 2 live
  uint32_t hit = 0;
};
struct WaitTable {
  std::vector<Slot> slot;
  size_t used = 0, filled = 0;
  WaitTable() : slot(1 << 14) {}
  void rehash(size_t n) {
    std::vector<Slot> old;
    old.swap(slot);
    slot.assign(n, Slot());
    filled = used;
    for (const Slot& s : old)
      if (s.state == 2) {
        size_t j = (size_t)s.hash & (n - 1);
        while (slot[j].state) j = (j + 1) & (n - 1);
        slot[j] = s;
      }
  }
  long find(uint64_t h, const char* name, size_t n) const {
    const size_t mask = slot.size() - 1;
    for (size_t j = (size_t)h & mask;; j = (j + 1) & mask) {
      const Slot& s = slot[j];
      if (s.state == 0) return -1;
      if (s.state == 2 && s.hash == h && s.rec->name_len == n && memcmp(s.rec->name, name, n) == 0) return (long)j;
    }
  }
  void insert(const Rec* r, uint32_t hit) {
    if ((filled + 1) * 2 > slot.size()) rehash(used * 4 > slot.size() ? slot.size() * 2 : slot.size());
    const size_t mask = slot.size() - 1;
    size_t j = (size_t)r->hash & mask;
    while (slot[j].state == 2) j = (j + 1) & mask;
    if (slot[j].state == 0) ++filled;
    slot[j].hash = r->hash;
    slot[j].rec = r;
    slot[j].owned = nullptr;
    slot[j].state = 2;
    slot[j].hit = hit;
    ++used;
  }
  void erase(long j) {
    delete slot[(size_t)j].owned;
    slot[(size_t)j].owned = nullptr;
    slot[(size_t)j].state = 1;
    --used;
  }
};

static bool in_process = false;  // run() called by run_packed: it has a temporary file to remove afterwards

static int run(int argc, char** argv) {
  printf("Call is --sam CHRFILE PreBuiltMutHash SAM|stdin firstpassfile hashsize MinQ HashCountThreshold threads\n");
  if (argc < 10) {
    printf("ERROR: expected 9 arguments\n");
    return 0;
  }
  const char* chr_path = argv[2];
  const char* hashlist = argv[3];
  const char* sam = argv[4];
  const std::string stub = argv[5];
  const int k = atoi(argv[6]), min_q = atoi(argv[7]), thresh = atoi(argv[8]);
  std::string text;
  {
    std::ifstream f(hashlist, std::ios::binary);
    if (!f.is_open()) {
      printf("Error, ParentHashFile could not be opened");
      return 0;
    }
    text.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  const int fd = strcmp(sam, "stdin") == 0 || strcmp(sam, "/dev/stdin") == 0 || strcmp(sam, "-") == 0 ? 0 : ::open(sam, O_RDONLY);
  if (fd < 0) {
    printf("Error, MutFile could not be opened");
    return 0;
  }
  FILE* chr = fopen(chr_path, "w");
  std::ofstream out1((stub + ".Mutations.Mate1.fastq").c_str(), std::ios::binary);
  std::ofstream out2((stub + ".Mutations.Mate2.fastq").c_str(), std::ios::binary);
  if (!chr || !out1.is_open() || !out2.is_open()) {
    printf("ERROR, Output file could not be opened -%s\n", stub.c_str());
    return 0;
  }
  if (k < 1 || k > 32) die("rufus_amd RUFUS.Filter: hash size must be 1..32");
  const long nk = rfx_hashlist_keys(text.data(), text.size(), k, 0, nullptr, 0);
  if (nk < 0) die("rufus_amd: cannot parse the hash list");
  std::vector<uint64_t> keys((size_t)nk + 1);
  rfx_hashlist_keys(text.data(), text.size(), k, 0, keys.data(), keys.size());
  printf("\nDone Hash Files\n\t Mutations Hash size is %ld\n", nk);
  // RUFUS_GPUS: the pieces are dealt to the devices by helper thread (the filter shards by read block)
  const std::vector<int> gpus = gpu_list();
  const int n_gpu = (int)gpus.size();
  std::vector<rfx_ctx*> ctxs = open_ctxs(gpus);
  std::vector<rfx_set*> sets;
  for (rfx_ctx* c : ctxs) {
    rfx_set* st = rfx_set_build(c, keys.data(), (uint64_t)nk, k);
    if (!st) die(std::string("rufus_amd: ") + rfx_last_error());
    sets.push_back(st);
  }
  trace("filter --sam: device open, set built");
#ifdef F_SETPIPE_SZ
  (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);
#endif

  unsigned helpers = (unsigned)std::max(1, atoi(argv[9]));
  helpers = std::min(helpers, rfx_host_cpus() > 2 ? rfx_host_cpus() - 2 : 1u);
  if (const char* ev = getenv("RFX_HOST_THREADS")) helpers = (unsigned)std::max(1, atoi(ev));
  helpers = std::max(helpers, (unsigned)n_gpu);
  size_t PIECE = 32u << 20;
  if (const char* ev = getenv("RFX_INGEST_PIECE")) PIECE = (size_t)std::max(1024, atoi(ev));
  const size_t LAG = 3;
  const size_t MAX_IN_FLIGHT = 2 * (size_t)helpers + 2 + LAG;

  std::mutex mu;
  std::vector<std::mutex> dev_mu((size_t)n_gpu);
  std::condition_variable cv;
  std::deque<SPiece*> todo, spare;
  std::map<uint64_t, SPiece*> done;
  bool input_end = false;
  uint64_t n_pieces = 0;
  size_t in_flight = 0;
  // A regular file (the spool `jellyfish count --spool` left, a SAM file): mapped and cut at line ends, no copy -- one
  // reader thread copying out of the page cache gave 5.8 GB/s, less than the helpers parse.
  const char* map = nullptr;
  size_t map_size = 0;
  {
    struct stat sb;
    if (fd != 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && !getenv("RFX_FILTER_NO_MMAP")) {
      void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m != MAP_FAILED) {
        (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
        map = (const char*)m;
        map_size = (size_t)sb.st_size;
      }
    }
  }
  std::thread reader([&] {
    if (map) {
      size_t at = 0;
      while (at < map_size) {
        size_t end = std::min(map_size, at + PIECE);
        if (end < map_size) {
          const char* nl = (const char*)memchr(map + end, '\n', map_size - end);
          end = nl ? (size_t)(nl - map) + 1 : map_size;
        }
        SPiece* pc = nullptr;
        {
          std::unique_lock<std::mutex> g(mu);
          cv.wait(g, [&] { return in_flight < MAX_IN_FLIGHT; });
          if (!spare.empty()) { pc = spare.front(); spare.pop_front(); }
          ++in_flight;
        }
        if (!pc) pc = new SPiece();
        pc->mapped = map + at;
        pc->size = end - at;
        at = end;
        std::lock_guard<std::mutex> g(mu);
        pc->seq = n_pieces++;
        todo.push_back(pc);
        cv.notify_all();
      }
      std::lock_guard<std::mutex> g(mu);
      input_end = true;
      cv.notify_all();
      return;
    }
    std::vector<char> carry;
    bool eof = false;
    while (!eof) {
      SPiece* pc = nullptr;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return in_flight < MAX_IN_FLIGHT; });
        if (!spare.empty()) { pc = spare.front(); spare.pop_front(); }
        ++in_flight;
      }
      if (!pc) pc = new SPiece();
      pc->mapped = nullptr;
      size_t fill = carry.size();
      if (pc->text.size() < fill + PIECE + (1u << 20)) pc->text.resize(fill + PIECE + (1u << 20));
      if (fill) memcpy(pc->text.data(), carry.data(), fill);  // (an empty vector may hand out a null pointer)
      carry.clear();
      bool have_nl = fill && memchr(pc->text.data(), '\n', fill);
      while (fill < PIECE || !have_nl) {  // at least PIECE bytes AND one line end
        if (fill == pc->text.size()) pc->text.resize(fill * 2);
        const size_t room = pc->text.size() - fill, want_more = fill < PIECE ? PIECE + (1u << 20) - fill : (size_t)1 << 20;
        const ssize_t n = ::read(fd, pc->text.data() + fill, std::min(room, want_more));
        if (n < 0 && errno == EINTR) continue;
        if (n < 0) die(std::string("read error on input: ") + strerror(errno));
        if (n == 0) { eof = true; break; }
        if (!have_nl && memchr(pc->text.data() + fill, '\n', (size_t)n)) have_nl = true;
        fill += (size_t)n;
      }
      size_t cut = fill;
      if (!eof) {
        while (cut > 0 && pc->text[cut - 1] != '\n') --cut;
        carry.assign(pc->text.data() + cut, pc->text.data() + fill);
      }
      pc->size = cut;
      std::lock_guard<std::mutex> g(mu);
      pc->seq = n_pieces++;
      todo.push_back(pc);
      cv.notify_all();
    }
    std::lock_guard<std::mutex> g(mu);
    input_end = true;
    cv.notify_all();
  });

  auto helper = [&](unsigned me) {
    const size_t dev = (size_t)me % (size_t)n_gpu;
    rfx_ctx* ctx = ctxs[dev];
    rfx_set* set = sets[dev];
    struct Pinned {
      void* p = nullptr;
      size_t cap = 0;
      void* need(size_t bytes) {
        if (bytes > cap) {
          if (p) rfx_host_free(p);
          cap = bytes + bytes / 4;
          p = rfx_host_alloc(cap);
          if (!p) die("rufus_amd: cannot allocate pinned staging memory");
        }
        return p;
      }
      ~Pinned() { if (p) rfx_host_free(p); }
    } pin_codes, pin_good, pin_woff, pin_lens;
    std::vector<uint64_t> so, qo;
    std::vector<uint32_t> sl;
    std::string rs, rq;
    Field f[11];
    for (;;) {
      SPiece* pcp;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return !todo.empty() || input_end; });
        if (todo.empty()) return;
        pcp = todo.front();
        todo.pop_front();
      }
      SPiece& pc = *pcp;
      pc.recs.clear();
      pc.chr_runs.clear();
      pc.waited.clear();
      // (offsets handed to rfx_pack_spans are byte distances from `base`, modulo 2^64: the printed forms lie in other
      // allocations than the lines)
      const char* base = pc.mapped ? pc.mapped : pc.text.data();
      const char *p = base, *e = base + pc.size;
      pc.side.reset();
      const char* cur = nullptr;
      size_t cur_len = 0;
      uint64_t words = 0;
      while (p < e) {
        const char* nl = find_nl(p, e);
        const char* le = nl ? nl : e;
        if (split_sam(p, le, f)) {  // fewer than 11 fields: skipped, as the feeders do
          if (!cur || f[2].n != cur_len || memcmp(f[2].p, cur, cur_len) != 0) {
            pc.chr_runs.emplace_back(f[2].p, f[2].n);
            cur = f[2].p;
            cur_len = f[2].n;
          }
          Rec r;
          r.name = f[0].p; r.name_len = (uint32_t)f[0].n;
          r.seq = f[9].p; r.seq_len = (uint32_t)f[9].n;
          r.qual = f[10].p; r.qual_len = (uint32_t)f[10].n;
          if (sam_flag(f[1]) & 16) {
            revcomp_into(rs, f[9]);
            reverse_into(rq, f[10]);
            char* side = pc.side.take(rs.size() + rq.size());
            memcpy(side, rs.data(), rs.size());
            memcpy(side + rs.size(), rq.data(), rq.size());
            r.seq = side; r.seq_len = (uint32_t)rs.size();
            r.qual = side + rs.size(); r.qual_len = (uint32_t)rq.size();
          }
          r.hash = name_hash(r.name, r.name_len);
          words += (r.seq_len + 31) / 32;
          pc.recs.push_back(r);
        }
        p = nl ? nl + 1 : e;
      }
      const size_t n = pc.recs.size();
      pc.mask.assign((n + 63) / 64, 0);
      if (n) {
        so.resize(n); qo.resize(n); sl.resize(n);
        for (size_t i = 0; i < n; ++i) {
          const Rec& r = pc.recs[i];
          so[i] = (uint64_t)((uintptr_t)r.seq - (uintptr_t)base);
          sl[i] = r.seq_len;
          if (r.qual_len >= r.seq_len) {
            qo[i] = (uint64_t)((uintptr_t)r.qual - (uintptr_t)base);
          } else {  // a quality string shorter than its read: the missing characters are bad ('\0'), as in the file route
            char* side = pc.side.take(r.seq_len);
            memcpy(side, r.qual, r.qual_len);
            memset(side + r.qual_len, 0, r.seq_len - r.qual_len);
            qo[i] = (uint64_t)((uintptr_t)side - (uintptr_t)base);
          }
        }
        uint64_t* codes = (uint64_t*)pin_codes.need((words + 1) * 8);
        uint32_t* good = (uint32_t*)pin_good.need((words + 1) * 4);
        uint32_t* woff = (uint32_t*)pin_woff.need((n + 1) * 4);
        uint32_t* lens = (uint32_t*)pin_lens.need((n + 1) * 4);
        woff[0] = 0;
        if (rfx_pack_spans(base, so.data(), sl.data(), qo.data(), (uint32_t)n, min_q, RFX_PACK_FILTER, codes, nullptr, good,
                           woff, lens) != RFX_OK)
          die("rufus_amd: pack failed");
        std::lock_guard<std::mutex> g(dev_mu[dev]);
        rfx_reads* rd = rfx_reads_upload(ctx, codes, nullptr, good, woff, lens, (uint32_t)n);
        if (!rd) die(std::string("rufus_amd: upload failed: ") + rfx_last_error());
        uint64_t nh = 0;
        const int rc = rfx_filter(set, rd, thresh, 1, nullptr, pc.mask.data(), &nh);
        rfx_reads_free(rd);
        if (rc) die(std::string("rufus_amd: filter failed: ") + rfx_last_error());
      }
      std::lock_guard<std::mutex> g(mu);
      done[pc.seq] = pcp;
      cv.notify_all();
    }
  };
  std::vector<std::thread> workers;
  helpers = std::max(helpers, (unsigned)n_gpu);
  for (unsigned t = 0; t < helpers; ++t) workers.emplace_back(helper, t);

  WaitTable waiting;
  std::deque<SPiece*> live;
  std::string current = "notachr", o1, o2;
  unsigned long long n_rec = 0, n_pairs = 0, n_found = 0;
  auto put_text = [](std::string& o, const char* name, size_t nn, const char* seq, size_t ls, const char* qual, size_t lq) {
    o.push_back('@');
    o.append(name, nn);
    o.push_back('\n');
    o.append(seq, ls);
    o.append("\n+\n", 3);
    o.append(qual, lq);
    o.push_back('\n');
  };
  auto retire = [&](SPiece* pc) {  // what still waits from this piece leaves it
    for (uint32_t idx : pc->waited) {
      const Rec* r = &pc->recs[idx];
      const long at = waiting.find(r->hash, r->name, r->name_len);
      if (at < 0 || waiting.slot[(size_t)at].rec != r) continue;
      Owned* o = new Owned();
      o->bytes.assign(r->name, r->name_len);
      o->bytes.append(r->seq, r->seq_len);
      o->bytes.append(r->qual, r->qual_len);
      o->r = *r;
      o->r.name = o->bytes.data();
      o->r.seq = o->bytes.data() + r->name_len;
      o->r.qual = o->bytes.data() + r->name_len + r->seq_len;
      waiting.slot[(size_t)at].rec = &o->r;
      waiting.slot[(size_t)at].owned = o;
    }
    std::lock_guard<std::mutex> g(mu);
    spare.push_back(pc);
    --in_flight;
    cv.notify_all();
  };
  for (uint64_t want = 0;; ++want) {
    SPiece* pc;
    {
      std::unique_lock<std::mutex> g(mu);
      cv.wait(g, [&] { return done.count(want) || (input_end && want >= n_pieces); });
      if (!done.count(want)) break;
      pc = done[want];
      done.erase(want);
    }
    for (const std::string& name : pc->chr_runs)
      if (name != current) {
        fprintf(chr, "%s\n", current.c_str());
        current = name;
      }
    const size_t n = pc->recs.size();
    for (size_t i = 0; i < n; ++i) {
      const Rec& r = pc->recs[i];
      const uint32_t hit = (uint32_t)((pc->mask[i >> 6] >> (i & 63)) & 1);
      const long at = waiting.find(r.hash, r.name, r.name_len);
      if (at < 0) {
        waiting.insert(&r, hit);
        pc->waited.push_back((uint32_t)i);
      } else {
        const Slot& s = waiting.slot[(size_t)at];
        // (the later record is mate 1 of the pair: the paired tool rejects an empty mate-1 sequence line)
        if (r.seq_len == 0)
          die("rufus_amd RUFUS.Filter: empty sequence line (undefined behaviour in the reference) -- rejected");
        if (hit | s.hit) {
          put_text(o1, r.name, r.name_len, r.seq, r.seq_len, r.qual, r.qual_len);
          put_text(o2, r.name, r.name_len, s.rec->seq, s.rec->seq_len, s.rec->qual, s.rec->qual_len);
          ++n_found;
        }
        ++n_pairs;
        waiting.erase(at);
      }
    }
    n_rec += n;
    if (!o1.empty()) {
      out1.write(o1.data(), (std::streamsize)o1.size());
      out2.write(o2.data(), (std::streamsize)o2.size());
      o1.clear();
      o2.clear();
    }
    live.push_back(pc);
    if (live.size() > LAG) {
      retire(live.front());
      live.pop_front();
    }
    printf("Read in %llu lines: Found %llu \r", n_pairs * 4, n_found);
  }
  while (!live.empty()) {
    retire(live.front());
    live.pop_front();
  }
  fprintf(chr, "%s\n", current.c_str());
  fclose(chr);
  reader.join();
  for (auto& w : workers) w.join();
  trace("filter --sam: all pieces done");
  printf("\nSAM records %llu, pairs %llu, pulled %llu\n", n_rec, n_pairs, n_found);
  out1.close();
  out2.close();
  printf("\nDone running RUFUS.Filter.cpp\n");
  if (!in_process) leave(0);  // (outputs closed: see the FASTQ route's end)
  for (rfx_set* st : sets) rfx_set_free(st);
  for (rfx_ctx* c : ctxs) rfx_close(c);
  return 0;
}

// `RUFUS.Filter --packed CACHE CHRFILE PreBuiltMutHash SPOOL firstpassfile hashsize MinQ HashCountThreshold threads`
// (rfx_packed_cache.hpp): the subject's records were packed once, by `jellyfish count --sam ..
// --keep-packed CACHE`.  Here they are uploaded as they lie in CACHE and scanned; only the lines of the names with a hit
// are gathered from SPOOL (the stream's bytes: `--spool`, or the SAM file itself) and go through the text route above.
static int run_packed(int argc, char** argv) {
  printf("Call is --packed CACHE CHRFILE PreBuiltMutHash SPOOL firstpassfile hashsize MinQ HashCountThreshold threads\n");
  if (argc < 11) {
    printf("ERROR: expected 10 arguments\n");
    return 0;
  }
  const char *cache_path = argv[2], *chr_path = argv[3], *hashlist = argv[4], *spool = argv[5];
  const int k = atoi(argv[7]), min_q = atoi(argv[8]), thresh = atoi(argv[9]);
  auto text_route = [&](const char* sam_path, const char* chr_to) {  // the ordinary --sam route on `sam_path`
    std::vector<std::string> a{argv[0], "--sam", chr_to, hashlist, sam_path, argv[6], argv[7], argv[8], argv[9], argv[10]};
    std::vector<char*> av;
    for (std::string& x : a) av.push_back(&x[0]);
    return run((int)av.size(), av.data());
  };
  // the cache and the spool, mapped
  struct Map {
    const char* p = nullptr;
    size_t n = 0;
    bool open(const char* path) {
      const int fd = ::open(path, O_RDONLY);
      struct stat sb;
      if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { if (fd >= 0) ::close(fd); return false; }
      n = (size_t)sb.st_size;
      if (n) {
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); return false; }
        p = (const char*)m;
      }
      ::close(fd);
      return true;
    }
  } cm, sm;
  std::vector<rfxcache::ChunkView> chunks;
  int cached_q = 0;
  if (!sm.open(spool)) {
    printf("Error, MutFile could not be opened");
    return 0;
  }
  if (!cm.open(cache_path) || !rfxcache::read_chunks(cm.p, cm.n, sm.n, cached_q, chunks) || cached_q != min_q) {
    // no cache, one its producer did not finish, a cache of another stream (length), or packed for another MinQ: the text decides
    fprintf(stderr, "rufus_amd RUFUS.Filter: %s is not a usable packed-read cache for MinQ %d: scanning the text\n", cache_path, min_q);
    return text_route(spool, chr_path);
  }
  std::string text;
  {
    std::ifstream f(hashlist, std::ios::binary);
    if (!f.is_open()) {
      printf("Error, ParentHashFile could not be opened");
      return 0;
    }
    text.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  if (k < 1 || k > 32) die("rufus_amd RUFUS.Filter: hash size must be 1..32");
  const long nk = rfx_hashlist_keys(text.data(), text.size(), k, 0, nullptr, 0);
  if (nk < 0) die("rufus_amd: cannot parse the hash list");
  std::vector<uint64_t> keys((size_t)nk + 1);
  rfx_hashlist_keys(text.data(), text.size(), k, 0, keys.data(), keys.size());
  std::unordered_set<uint64_t> hit_names;
  uint64_t n_rec = 0, n_hit = 0, n_always = 0;
  {
    const std::vector<int> gpus = gpu_list();
    std::vector<rfx_ctx*> ctxs = open_ctxs(gpus);
    std::vector<rfx_set*> sets;
    for (rfx_ctx* c : ctxs) {
      rfx_set* st = rfx_set_build(c, keys.data(), (uint64_t)nk, k);
      if (!st) die(std::string("rufus_amd: ") + rfx_last_error());
      sets.push_back(st);
    }
    trace("filter --packed: device open, set built");
    // chunk i goes to device i mod N; one thread per device uploads and scans (the arrays lie in the cache as the
    // upload wants them: no parsing, no packing)
    std::mutex mu;
    std::vector<std::thread> th;
    for (size_t d = 0; d < ctxs.size(); ++d)
      th.emplace_back([&, d] {
        std::vector<uint64_t> mask;
        std::vector<uint64_t> local;
        uint64_t rec = 0, hit = 0, always = 0;
        for (size_t i = d; i < chunks.size(); i += ctxs.size()) {
          const rfxcache::ChunkView& v = chunks[i];
          const uint32_t n = v.h->n;
          if (!n) continue;
          mask.assign(((size_t)n + 63) / 64, 0);
          rfx_reads* rd = rfx_reads_upload(ctxs[d], v.codes, nullptr, v.good, v.word_off, v.len, n);
          if (!rd) die(std::string("rufus_amd: upload failed: ") + rfx_last_error());
          uint64_t nh = 0;
          const int rc = rfx_filter(sets[d], rd, thresh, 1, nullptr, mask.data(), &nh);
          rfx_reads_free(rd);
          if (rc) die(std::string("rufus_amd: filter failed: ") + rfx_last_error());
          for (uint32_t r = 0; r < n; ++r) {
            const bool al = v.flags[r] & rfxcache::REC_ALWAYS;
            if (al || ((mask[r >> 6] >> (r & 63)) & 1)) {
              local.push_back(v.hash[r]);
              if (al) ++always; else ++hit;
            }
          }
          rec += n;
        }
        std::lock_guard<std::mutex> g(mu);
        hit_names.insert(local.begin(), local.end());
        n_rec += rec;
        n_hit += hit;
        n_always += always;
      });
    for (auto& t : th) t.join();
    for (rfx_set* st : sets) rfx_set_free(st);
    for (rfx_ctx* c : ctxs) rfx_close(c);
  }
  trace("filter --packed: cache scanned");
  // every line whose name (hash) has a hit, in stream order
  std::string mini;
  uint64_t n_lines = 0;
  for (const rfxcache::ChunkView& v : chunks)
    for (uint32_t r = 0; r < v.h->n; ++r)
      if (hit_names.count(v.hash[r])) {
        const uint64_t at = v.h->stream_off + v.line_off[r];  // (inside the spool: read_chunks checked every line)
        // the line must be the record the chunk was made from: same QNAME (a cache left over from another stream of
        // the same length would otherwise hand the text route the wrong lines without anybody noticing)
        const char* ln = sm.p + at;
        const char* tab = (const char*)memchr(ln, '\t', v.line_len[r]);
        if (!tab || rfxsam::name_hash(ln, (size_t)(tab - ln)) != v.hash[r]) {
          fprintf(stderr, "rufus_amd RUFUS.Filter: %s does not describe %s (a line is not the record it was packed from): scanning the text\n",
                  cache_path, spool);
          return text_route(spool, chr_path);
        }
        mini.append(ln, v.line_len[r]);
        mini.push_back('\n');
        ++n_lines;
      }
  printf("\npacked cache: %llu records scanned, %llu hit, %llu left to the text route outright; %llu lines gathered\n",
         (unsigned long long)n_rec, (unsigned long long)n_hit, (unsigned long long)n_always, (unsigned long long)n_lines);
  // the chromosome log of the WHOLE stream (src/PassThroughSamCheck.stranded.cpp: "notachr", then every run of RNAME)
  {
    FILE* chr = fopen(chr_path, "w");
    if (!chr) {
      printf("ERROR, Output file could not be opened -%s\n", chr_path);
      return 0;
    }
    std::string current = "notachr";
    for (const rfxcache::ChunkView& v : chunks) {
      const char *p = v.runs, *e = v.runs + v.h->runs_bytes;
      while (p < e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
        if (!nl) nl = e;
        if (current.size() != (size_t)(nl - p) || memcmp(current.data(), p, current.size()) != 0) {
          fprintf(chr, "%s\n", current.c_str());
          current.assign(p, (size_t)(nl - p));
        }
        p = nl + 1;
      }
    }
    fprintf(chr, "%s\n", current.c_str());
    fclose(chr);
  }
  // the gathered lines through the text route (a file the route can map; its own chromosome log is not the stream's)
  const char* tmpdir = access("/dev/shm", W_OK) == 0 ? "/dev/shm" : "/tmp";
  std::string tmp = std::string(tmpdir) + "/rfx_packed_XXXXXX";
  const int tfd = mkstemp(&tmp[0]);
  if (tfd < 0) die("rufus_amd RUFUS.Filter --packed: cannot create a temporary file");
  for (size_t at = 0; at < mini.size();) {
    const ssize_t w = ::write(tfd, mini.data() + at, mini.size() - at);
    if (w < 0 && errno == EINTR) continue;
    if (w <= 0) die("rufus_amd RUFUS.Filter --packed: write error on the temporary file");
    at += (size_t)w;
  }
  ::close(tfd);
  in_process = true;
  const int rc = text_route(tmp.c_str(), "/dev/null");
  ::unlink(tmp.c_str());
  return rc;
}
}  // namespace samf
#endif

int main(int argc, char** argv) {
#ifndef RFX_SINGLE_END
  if (argc > 1 && strcmp(argv[1], "--sam") == 0) return samf::run(argc, argv);
  if (argc > 1 && strcmp(argv[1], "--packed") == 0) return samf::run_packed(argc, argv);
#endif
#ifdef RFX_SINGLE_END
  const int need = 8;
  printf("Call is PreBuiltMutHash Mutant.fq|stdin firstpassfile hashsize MinQ HashCountThreshold threads\n");
#else
  const int need = 9;
  printf("Call is PreBuiltMutHash Mutant.Mate1.fq Mutant.Mate2.fq firstpassfile hashsize MinQ HashCountThreshold threads \n");
#endif
  if (argc < need) {
    printf("ERROR: expected %d arguments\n", need - 1);
    return 0;  // the reference tools report on stdout and exit 0; the shell checks for empty outputs
  }
  int a = 1;
  const char* hashlist = argv[a++];
  const char* m1 = argv[a++];
#ifndef RFX_SINGLE_END
  const char* m2 = argv[a++];
#endif
  const std::string stub = argv[a++];
  const int k = atoi(argv[a++]), min_q = atoi(argv[a++]), thresh = atoi(argv[a++]);

  std::string text;
  {
    std::ifstream f(hashlist, std::ios::binary);
    if (!f.is_open()) {
      printf("Error, ParentHashFile could not be opened");
      return 0;
    }
    text.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  const int fd1 = strcmp(m1, "stdin") == 0 || strcmp(m1, "/dev/stdin") == 0 ? 0 : ::open(m1, O_RDONLY);
  if (fd1 < 0) {
    printf("Error, MutFile could not be opened");
    return 0;
  }
#ifdef RFX_SINGLE_END
  const int single = 1;
  std::ofstream out1((stub + ".Mutations.fastq").c_str(), std::ios::binary);
#else
  const int single = 0;
  const int fd2 = ::open(m2, O_RDONLY);
  if (fd2 < 0) {
    printf("Error, MutFile could not be opened");
    return 0;
  }
  std::ofstream out1((stub + ".Mutations.Mate1.fastq").c_str(), std::ios::binary);
  std::ofstream out2((stub + ".Mutations.Mate2.fastq").c_str(), std::ios::binary);
  if (!out2.is_open()) {
    printf("ERROR, Output file could not be opened -%s\n", stub.c_str());
    return 0;
  }
#endif
  if (!out1.is_open()) {
    printf("ERROR, Output file could not be opened -%s\n", stub.c_str());
    return 0;
  }
  if (k < 1 || k > 32) die("rufus_amd RUFUS.Filter: hash size must be 1..32");

  const long nk = rfx_hashlist_keys(text.data(), text.size(), k, single, nullptr, 0);
  if (nk < 0) die("rufus_amd: cannot parse the hash list");
  std::vector<uint64_t> keys((size_t)nk + 1);
  rfx_hashlist_keys(text.data(), text.size(), k, single, keys.data(), keys.size());
  printf("\nDone Hash Files\n\t Mutations Hash size is %ld\n", nk);

  trace("filter: hash list parsed");
  // RUFUS_GPUS: the pieces are dealt to the devices by worker thread (the filter shards by read block)
  const std::vector<int> gpus = gpu_list();
  const int n_gpu = (int)gpus.size();
  // The device is opened (0.1 - 0.25 s of runtime start-up) and the set built beside the readers and the workers' first
  // pieces: a worker needs them when its first piece is packed (RFX_SYNC_OPEN=1: before anything else, as it used to be).
  std::vector<rfx_ctx*> ctxs;
  std::vector<rfx_set*> sets;
  std::mutex open_mu;
  std::condition_variable open_cv;
  bool dev_ready = false;
  std::thread opener([&] {
    std::vector<rfx_ctx*> cs = open_ctxs(gpus);
    std::vector<rfx_set*> ss;
    for (rfx_ctx* c : cs) {
      rfx_set* st = rfx_set_build(c, keys.data(), (uint64_t)nk, k);
      if (!st) die(std::string("rufus_amd: ") + rfx_last_error());
      ss.push_back(st);
    }
    trace("filter: device open, set built");
    std::lock_guard<std::mutex> g(open_mu);
    ctxs = std::move(cs);
    sets = std::move(ss);
    dev_ready = true;
    open_cv.notify_all();
  });
  auto wait_device = [&] {
    std::unique_lock<std::mutex> g(open_mu);
    open_cv.wait(g, [&] { return dev_ready; });
  };
  if (getenv("RFX_SYNC_OPEN")) wait_device();

  // Pipeline (the device scans ~1000x faster than one core parses, so the host side is what counts):
  //   one reader per mate stream cuts it into pieces of PIECE_RECS records (4 lines each, counted blindly like the
  //   reference's getline x 4); the two pipes are always being drained, so a writer in lock step
  //   (PassThroughSamCheck.stranded, runRufus.sh:966) never blocks on the one while we wait on the other;
  //   workers take piece i of both streams, find the lines, pack bases + quality mask of both mates
  //   (rfx_pack_spans), run the scan (k_filter; device calls are serialised, they take microseconds) and format
  //   the pulled records;
  //   the main thread writes the formatted pieces in input order.
  const size_t PIECE_RECS = 1u << 16;
  // A piece: PIECE_RECS records (the last one of a stream: fewer) = 4 * recs lines of text, each ending in '\n'.
  // Its bytes are the piece's own buffer (filled by read()) or lie in the mapping of a regular input file.
  struct Piece {
    const char* data = nullptr;
    size_t size = 0, recs = 0;
    std::unique_ptr<char[]> own;
    size_t cap = 0;
  };
  struct Stream {
    int fd = -1;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Piece*> q;
    bool done = false;
    bool abort = false;         // mate 1 ended: nobody takes this stream's pieces any more
    const char* map = nullptr;  // regular file: mapped
    size_t map_size = 0;
  };
  const int n_streams = single ? 1 : 2;
  Stream st[2];
  st[0].fd = fd1;
#ifndef RFX_SINGLE_END
  st[1].fd = fd2;
#endif
  for (int i = 0; i < n_streams; ++i) {
    struct stat sb;
    if (fstat(st[i].fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && !getenv("RFX_FILTER_NO_MMAP")) {
      void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, st[i].fd, 0);
      if (m != MAP_FAILED) {
        (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
        st[i].map = (const char*)m;
        st[i].map_size = (size_t)sb.st_size;
      }
    } else {
#ifdef F_SETPIPE_SZ
      (void)fcntl(st[i].fd, F_SETPIPE_SZ, 1 << 20);  // a pipe: fewer, larger reads (ignored on anything else)
#endif
    }
  }
  const skip_lines_fn skip_lines = pick_skip_lines();
  const index_lines_fn index_lines = pick_index_lines();
  const size_t MAX_AHEAD = 48;  // pieces a reader may be ahead of the workers (~1 GB of text per stream)
  auto new_piece = [](size_t cap) {
    Piece* p = new Piece();
    p->own.reset(new char[cap]);  // (not value-initialised: nothing is written that read() will not overwrite)
    p->cap = cap;
    p->data = p->own.get();
    return p;
  };
  auto grow = [](Piece* p, size_t cap) {
    std::unique_ptr<char[]> nb(new char[cap]);
    memcpy(nb.get(), p->data, p->size);
    p->own = std::move(nb);
    p->cap = cap;
    p->data = p->own.get();
  };
  auto reader = [&](Stream& S) {
    auto push = [&](Piece* pc) {
      std::unique_lock<std::mutex> g(S.mu);
      S.cv.wait(g, [&] { return S.q.size() < MAX_AHEAD || S.abort; });
      if (S.abort) {  // (the reference stops at the end of file 1: what mate 2 still holds is read and dropped)
        delete pc;
        return;
      }
      S.q.push_back(pc);
      S.cv.notify_all();
    };
    // the end of a stream, std::getline semantics: a final unterminated line still counts, and the missing lines
    // of an incomplete last record read as empty
    auto finish_tail = [&](Piece* cur, size_t lines) {
      if (cur->size) {
        if (cur->cap < cur->size + 8) grow(cur, cur->size + 8);
        char* d = cur->own.get();
        if (d[cur->size - 1] != '\n') {
          d[cur->size++] = '\n';
          ++lines;
        }
        cur->recs = (lines + 3) / 4;
        while (lines < 4 * cur->recs) {
          d[cur->size++] = '\n';
          ++lines;
        }
      }
      std::lock_guard<std::mutex> g(S.mu);
      if (cur->recs) S.q.push_back(cur);
      else delete cur;
      S.done = true;
      S.cv.notify_all();
    };
    if (S.map) {  // cut the mapping in place: only the last piece is copied (it may need lines added)
      const char *p = S.map, *e = S.map + S.map_size;
      for (;;) {
        size_t got;
        const char* cut = skip_lines(p, e, 4 * PIECE_RECS, got);
        if (got == 4 * PIECE_RECS) {
          Piece* pc = new Piece();
          pc->data = p;
          pc->size = (size_t)(cut - p);
          pc->recs = PIECE_RECS;
          push(pc);
          p = cut;
          continue;
        }
        Piece* last = new_piece((size_t)(e - p) + 8);
        memcpy(last->own.get(), p, (size_t)(e - p));
        last->size = (size_t)(e - p);
        finish_tail(last, got);
        return;
      }
    }
    const size_t CAP0 = PIECE_RECS * 360, RD = 1u << 20;
    Piece* cur = new_piece(CAP0);
    size_t lines = 0, scanned = 0;
    for (;;) {
      if (cur->cap - cur->size < RD) grow(cur, cur->cap + cur->cap / 2 + RD);
      const ssize_t n = ::read(S.fd, cur->own.get() + cur->size, RD);
      if (n < 0 && errno == EINTR) continue;
      if (n < 0) die(std::string("read error on input: ") + strerror(errno));
      if (n == 0) break;
      cur->size += (size_t)n;
      while (scanned < cur->size) {
        size_t got;
        const char* stop = skip_lines(cur->data + scanned, cur->data + cur->size, 4 * PIECE_RECS - lines, got);
        lines += got;
        scanned = (size_t)(stop - cur->data);
        if (lines < 4 * PIECE_RECS) break;  // (stop == end of what has been read)
        Piece* nx = new_piece(CAP0);
        nx->size = cur->size - scanned;
        memcpy(nx->own.get(), cur->data + scanned, nx->size);
        cur->size = scanned;
        cur->recs = PIECE_RECS;
        push(cur);
        cur = nx;
        lines = 0;
        scanned = 0;
      }
    }
    finish_tail(cur, lines);
  };
  std::thread readers[2];
  for (int i = 0; i < n_streams; ++i) readers[i] = std::thread(reader, std::ref(st[i]));

  struct Result { std::string out1, out2; unsigned long long recs = 0, found = 0; bool ready = false; };
  std::mutex res_mu, take_mu;
  std::vector<std::mutex> dev_mu((size_t)n_gpu);
  std::condition_variable res_cv;
  std::map<uint64_t, Result> results;
  uint64_t next_seq = 0;
  bool input_done = false;
  auto take = [&](Piece*& a, Piece*& b, uint64_t& seq) -> bool {  // piece `seq` of both streams, in order
    std::lock_guard<std::mutex> tg(take_mu);
    a = b = nullptr;
    {
      std::unique_lock<std::mutex> g(st[0].mu);
      st[0].cv.wait(g, [&] { return !st[0].q.empty() || st[0].done; });
      if (st[0].q.empty()) {
        {
          std::lock_guard<std::mutex> rg(res_mu);
          input_done = true;
          res_cv.notify_all();
        }
        if (n_streams == 2) {  // a mate-2 reader waiting for room must not wait for ever
          std::lock_guard<std::mutex> g2(st[1].mu);
          st[1].abort = true;
          st[1].cv.notify_all();
        }
        return false;
      }
      a = st[0].q.front();
      st[0].q.pop_front();
      st[0].cv.notify_all();
    }
    if (n_streams == 2) {
      std::unique_lock<std::mutex> g(st[1].mu);
      st[1].cv.wait(g, [&] { return !st[1].q.empty() || st[1].done; });
      if (!st[1].q.empty()) {
        b = st[1].q.front();
        st[1].q.pop_front();
        st[1].cv.notify_all();
      } else {
        b = new Piece();  // mate 2 ran out: its records read as empty (lock step: one per record of mate 1)
      }
    }
    seq = next_seq++;
    return true;
  };
  auto worker = [&](unsigned me) {
    const size_t dev = (size_t)me % (size_t)n_gpu;
    rfx_ctx* ctx = nullptr;  // (known once the opener is done: wait_device() before the first upload)
    rfx_set* set = nullptr;
    std::vector<uint64_t> ls[2], ss[2], qs[2], mask;  // line starts, sequence / quality starts
    std::vector<uint32_t> sl[2], hits;
    std::vector<char> fix;  // private copies of quality strings that are shorter than their read
    // packed reads go up from page-locked memory (a pageable source costs a staging copy inside the runtime, under
    // the device lock)
    // (while the device is still being opened the buffers are plain memory, page-locked before their first upload)
    struct Pinned {
      void* p = nullptr;
      size_t cap = 0;
      bool locked = false;
      void* need(size_t bytes, bool device_up = true) {
        if (bytes > cap) {
          if (p) rfx_host_free(p);
          cap = bytes + bytes / 4;
          p = device_up ? rfx_host_alloc(cap) : rfx_host_alloc_lazy(cap);
          locked = device_up;
          if (!p) die("rufus_amd: cannot allocate pinned staging memory");
        }
        return p;
      }
      void lock() {
        if (p && !locked) (void)rfx_host_pin(p);  // (refused: the upload stages the pageable buffer itself)
        locked = true;
      }
      ~Pinned() { if (p) rfx_host_free(p); }
    } pin_codes, pin_good, pin_woff, pin_lens;
    for (;;) {
      Piece *pc[2];
      uint64_t seq;
      if (!take(pc[0], pc[1], seq)) return;
      const size_t n = pc[0]->recs;
      for (int m = 0; m < n_streams; ++m) {
        const char* t = pc[m]->data;
        const size_t tsize = pc[m]->size;
        // (one line start more than the records need: a mate-2 piece may hold more records than mate 1's -- the extra
        // ones are never looked at, as the reference stops at the end of file 1 -- and the last record's quality line
        // must end where line 4n starts, not at the end of the piece)
        ls[m].resize(4 * n + 2);
        const size_t found_lines = index_lines(t, t + tsize, ls[m].data(), 4 * n + 1);
        for (size_t li = found_lines; li <= 4 * n; ++li) ls[m][li] = tsize;  // (mate 2 ran out: empty lines)
        ss[m].resize(n);
        qs[m].resize(n);
        sl[m].resize(n);
        for (size_t i = 0; i < n; ++i) {
          auto len_of = [&](size_t line) {
            const uint64_t a0 = ls[m][line], a1 = ls[m][line + 1];
            return a1 > a0 ? (uint32_t)(a1 - a0 - 1) : 0u;  // without the '\n'
          };
          ss[m][i] = ls[m][4 * i + 1];
          sl[m][i] = len_of(4 * i + 1);
          qs[m][i] = ls[m][4 * i + 3];
          if (m == 0 && sl[m][i] == 0)
            die("rufus_amd RUFUS.Filter: empty sequence line (undefined behaviour in the reference) -- rejected");
        }
      }
      // pack: reads of mate 1, then of mate 2, into one block
      const size_t nr = n * (size_t)n_streams;
      uint64_t words = 0;
      for (int m = 0; m < n_streams; ++m)
        for (size_t i = 0; i < n; ++i) words += (sl[m][i] + 31) / 32;
      const bool up = ctx != nullptr;
      uint64_t* codes = (uint64_t*)pin_codes.need((words + 1) * 8, up);
      uint32_t* good = (uint32_t*)pin_good.need((words + 1) * 4, up);
      uint32_t* woff = (uint32_t*)pin_woff.need((nr + 1) * 4, up);
      uint32_t* lens = (uint32_t*)pin_lens.need((nr + 1) * 4, up);
      uint32_t w0 = 0;
      for (int m = 0; m < n_streams; ++m) {
        const char* t = pc[m]->data;
        woff[m * n] = w0;
        // Runs of ordinary reads are packed in place.  A quality line shorter than its read would make the packer
        // read the next line's bytes: such a read is packed on its own from a private copy of its sequence and
        // quality string, the latter padded with '\0' (= bad, what the reference's missing chars are).
        size_t at = 0;
        while (at < n) {
          size_t run = at;
          auto qlen = [&](size_t i) {
            const uint64_t q0 = ls[m][4 * i + 3], q1 = ls[m][4 * i + 4];
            return q1 > q0 ? (uint32_t)(q1 - q0 - 1) : 0u;
          };
          while (run < n && qlen(run) >= sl[m][run]) ++run;
          if (run > at &&
              rfx_pack_spans(t, ss[m].data() + at, sl[m].data() + at, qs[m].data() + at, (uint32_t)(run - at), min_q,
                             RFX_PACK_FILTER, codes, nullptr, good, woff + m * n + at, lens + m * n + at) != RFX_OK)
            die("rufus_amd: pack failed");
          if (run < n) {
            const uint32_t L = sl[m][run], ql = qlen(run);
            fix.assign((size_t)2 * L, '\0');
            memcpy(fix.data(), t + ss[m][run], L);
            memcpy(fix.data() + L, t + qs[m][run], ql);
            const uint64_t s0 = 0, q0 = L;
            if (rfx_pack_spans(fix.data(), &s0, &L, &q0, 1, min_q, RFX_PACK_FILTER, codes, nullptr, good,
                               woff + m * n + run, lens + m * n + run) != RFX_OK)
              die("rufus_amd: pack failed");
            ++run;
          }
          at = run;
        }
        w0 = woff[(m + 1) * n];
      }
      mask.assign((nr + 63) / 64, 0);
      if (single) hits.assign(nr, 0);
      if (!ctx) {
        wait_device();
        ctx = ctxs[dev];
        set = sets[dev];
        pin_codes.lock();
        pin_good.lock();
        pin_woff.lock();
        pin_lens.lock();
      }
      {
        std::lock_guard<std::mutex> g(dev_mu[dev]);
        rfx_reads* rd = rfx_reads_upload(ctx, codes, nullptr, good, woff, lens, (uint32_t)nr);
        if (!rd) die(std::string("rufus_amd: upload failed: ") + rfx_last_error());
        uint64_t nh = 0;
        // paired tool: `i < length()-1`, the last base is never examined (src/RUFUS.Filter.cpp:203)
        const int rc = rfx_filter(set, rd, thresh, single ? 0 : 1, single ? hits.data() : nullptr, mask.data(), &nh);
        rfx_reads_free(rd);
        if (rc) die(std::string("rufus_amd: filter failed: ") + rfx_last_error());
      }
      Result res;
      res.recs = n;
      auto bit = [&](size_t r) { return (mask[r >> 6] >> (r & 63)) & 1; };
      auto put = [&](std::string& o, int m, size_t i, const char* suffix) {
        const char* t = pc[m]->data;
        for (int j = 0; j < 4; ++j) {
          const uint64_t a0 = ls[m][4 * i + j], a1 = ls[m][4 * i + j + 1];
          const size_t len = a1 > a0 ? (size_t)(a1 - a0 - 1) : 0;
          o.append(t + a0, len);
          if (j == 0 && suffix) o.append(suffix);
          o.push_back('\n');
        }
      };
      for (size_t i = 0; i < n; ++i) {
        if (single) {
          if (bit(i)) {
            const std::string suffix = ":MH" + std::to_string(hits[i]);
            put(res.out1, 0, i, suffix.c_str());
            ++res.found;
          }
        } else if (bit(i) | bit(n + i)) {  // == the reference's "mate 2 only if mate 1 failed" (:237-277)
          put(res.out1, 0, i, nullptr);
          put(res.out2, 1, i, nullptr);
          ++res.found;
        }
      }
      for (int m = 0; m < n_streams; ++m) {
        if (!pc[m]->own && st[m].map) drop_mapped(pc[m]->data, pc[m]->data + pc[m]->size);  // a range of the file's mapping
        delete pc[m];
      }
      res.ready = true;
      std::lock_guard<std::mutex> g(res_mu);
      results[seq] = std::move(res);
      res_cv.notify_all();
    }
  };
  unsigned nthreads = (unsigned)std::max(1, atoi(argv[a]));
  nthreads = std::min(nthreads, rfx_host_cpus());
  if (const char* ev = getenv("RFX_HOST_THREADS")) nthreads = (unsigned)std::max(1, atoi(ev));
  nthreads = std::max(nthreads, (unsigned)n_gpu);
  std::vector<std::thread> workers;
  for (unsigned t = 0; t < nthreads; ++t) workers.emplace_back(worker, t);
  unsigned long long found = 0, total = 0;
  for (uint64_t want = 0;; ++want) {
    Result res;
    {
      std::unique_lock<std::mutex> g(res_mu);
      res_cv.wait(g, [&] { return results.count(want) || (input_done && want >= next_seq); });
      if (!results.count(want)) break;
      res = std::move(results[want]);
      results.erase(want);
    }
    out1.write(res.out1.data(), (std::streamsize)res.out1.size());
#ifndef RFX_SINGLE_END
    out2.write(res.out2.data(), (std::streamsize)res.out2.size());
#endif
    total += res.recs;
    found += res.found;
    printf("Read in %llu lines: Found %llu \r", total * 4, found);
  }
  trace("filter: all pieces written");
  out1.close();
#ifndef RFX_SINGLE_END
  out2.close();
#endif
  printf("\nDone running RUFUS.Filter.cpp\n");
  opener.join();
  // Everything the caller will read is on disk: the process leaves here (rfx_cli.hpp leave(): unmapping 20 GB of input,
  // unpinning the staging blocks and the runtime's own teardown took 0.5 s of a 1.5 s run).  RFX_CLEAN_EXIT=1: the orderly way.
  leave(0);
  for (auto& w : workers) w.join();
  for (int i = 0; i < n_streams; ++i) readers[i].join();
  for (rfx_set* st : sets) rfx_set_free(st);
  for (rfx_ctx* c : ctxs) rfx_close(c);
  trace("filter: closed");
  return 0;
}
