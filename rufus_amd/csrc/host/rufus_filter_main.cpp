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

int main(int argc, char** argv) {
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
  rfx_ctx* ctx = open_ctx();
  rfx_set* set = rfx_set_build(ctx, keys.data(), (uint64_t)nk, k);
  trace("filter: device open, set built");
  if (!set) die(std::string("rufus_amd: ") + rfx_last_error());

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
      S.cv.wait(g, [&] { return S.q.size() < MAX_AHEAD; });
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
  std::mutex res_mu, dev_mu, take_mu;
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
        std::lock_guard<std::mutex> rg(res_mu);
        input_done = true;
        res_cv.notify_all();
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
  auto worker = [&]() {
    std::vector<uint64_t> ls[2], ss[2], qs[2], mask;  // line starts, sequence / quality starts
    std::vector<uint32_t> sl[2], hits;
    std::vector<char> fix;  // private copies of quality strings that are shorter than their read
    // packed reads go up from page-locked memory (a pageable source costs a staging copy inside the runtime, under
    // the device lock)
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
    for (;;) {
      Piece *pc[2];
      uint64_t seq;
      if (!take(pc[0], pc[1], seq)) return;
      const size_t n = pc[0]->recs;
      for (int m = 0; m < n_streams; ++m) {
        const char* t = pc[m]->data;
        const size_t tsize = pc[m]->size;
        ls[m].resize(4 * n + 1);
        const size_t found_lines = index_lines(t, t + tsize, ls[m].data(), 4 * n);
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
      uint64_t* codes = (uint64_t*)pin_codes.need((words + 1) * 8);
      uint32_t* good = (uint32_t*)pin_good.need((words + 1) * 4);
      uint32_t* woff = (uint32_t*)pin_woff.need((nr + 1) * 4);
      uint32_t* lens = (uint32_t*)pin_lens.need((nr + 1) * 4);
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
      {
        std::lock_guard<std::mutex> g(dev_mu);
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
      for (int m = 0; m < n_streams; ++m) delete pc[m];
      res.ready = true;
      std::lock_guard<std::mutex> g(res_mu);
      results[seq] = std::move(res);
      res_cv.notify_all();
    }
  };
  unsigned nthreads = (unsigned)std::max(1, atoi(argv[a]));
  nthreads = std::min(nthreads, rfx_host_cpus());
  if (const char* ev = getenv("RFX_HOST_THREADS")) nthreads = (unsigned)std::max(1, atoi(ev));
  std::vector<std::thread> workers;
  for (unsigned t = 0; t < nthreads; ++t) workers.emplace_back(worker);
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
  for (auto& w : workers) w.join();
  for (int i = 0; i < n_streams; ++i) readers[i].join();
  rfx_set_free(set);
  rfx_close(ctx);
  trace("filter: closed");
  printf("\nDone running RUFUS.Filter.cpp\n");
  return 0;
}
