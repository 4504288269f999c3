// Drop-in SAM -> FASTQ feeders, same argv, same output bytes:
//   PassThroughSamCheck             CHRFILE        stdin SAM -> stdout FASTQ        (src/PassThroughSamCheck.cpp:51-155)
//   PassThroughSamCheck.stranded    CHRFILE STUB   -> STUB.mate1.fastq/.mate2.fastq (src/PassThroughSamCheck.stranded.cpp:78-281)
//   PassThroughSamCheck.stranded.se CHRFILE        stdin SAM -> stdout FASTQ, reverse-strand reads restored
// Built three times from this file with -DPTS_MODE=0/1/2.  Pure host text plumbing in front of the
// count (RunJellyForRUFUS.sh:28) and filter (runRufus.sh:966) stages; buffered output instead of the
// reference's flush per line.
// Reference quirks kept: the chromosome log starts with "notachr" and records the PREVIOUS name at
// every change; QUAL ends at the next TAB (FastqToSam.pl appends one) -- here also at end of line,
// where the reference would run off the string; in the reverse-complement branch any base other
// than ACGTN disappears; the stranded tool emits a pair when its SECOND record arrives (that record
// goes to mate1, the stored one to mate2) and silently drops reads whose mate never shows up.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <mutex>
#include <map>
#include <deque>
#include <condition_variable>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "rfx_cli.hpp"
#include "rfx_sam.hpp"

#ifndef PTS_MODE
#define PTS_MODE 0
#endif

using namespace rfxsam;

namespace {

// The tools with helpers: the stream is cut into pieces at line ends; worker threads find the fields of every line
// and prepare what the record will print (reverse-strand reads put back the way they were sequenced; for the
// one-output tools the FASTQ text itself); ONE thread then walks the pieces in stream order and does what cannot
// be split -- the chromosome log, and for the stranded tool the pairing (a record meets its mate, or waits).
struct Parsed {
  const char *name, *chr, *seq, *qual;
  uint32_t name_len, chr_len, seq_len, qual_len;
  uint64_t hash;
};
struct Piece {
  std::vector<char> text;
  size_t size = 0;
  std::string arena;  // stranded: reverse-complemented sequences / reversed qualities (never reallocated);
                      // one-output tools: the FASTQ text of the piece
  std::vector<Parsed> recs;
  std::vector<std::string> chr_runs;  // one-output tools: the runs of equal RNAME in this piece
  uint64_t seq = 0;
};

// reader thread -> `helpers` x parse -> consume in stream order (on the calling thread)
template <class Parse, class Consume>
void process_stream(unsigned helpers, Parse parse, Consume consume) {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Piece*> todo, spare;
  std::map<uint64_t, Piece*> done;
  bool input_end = false;
  uint64_t n_pieces = 0;
  const size_t PIECE = 4u << 20;
  const size_t MAX_IN_FLIGHT = 2 * (size_t)helpers + 4;
  size_t in_flight = 0;
  std::thread reader([&] {
    std::vector<char> carry;
    bool eof = false;
    while (!eof) {
      Piece* pc = nullptr;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return in_flight < MAX_IN_FLIGHT; });
        if (!spare.empty()) { pc = spare.front(); spare.pop_front(); }
        ++in_flight;
      }
      if (!pc) pc = new Piece();
      size_t fill = carry.size();
      if (pc->text.size() < fill + PIECE + (1u << 20)) pc->text.resize(fill + PIECE + (1u << 20));
      if (fill) memcpy(pc->text.data(), carry.data(), fill);  // (an empty vector may hand out a null pointer)
      carry.clear();
      // a piece = at least PIECE bytes AND at least one line end (a line may be longer than any buffer so far)
      bool have_nl = fill && memchr(pc->text.data(), '\n', fill);
      while (fill < PIECE || !have_nl) {
        if (fill == pc->text.size()) pc->text.resize(fill * 2);
        const ssize_t n = ::read(0, pc->text.data() + fill, pc->text.size() - fill);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) { eof = true; break; }
        if (!have_nl && memchr(pc->text.data() + fill, '\n', (size_t)n)) have_nl = true;
        fill += (size_t)n;
      }
      size_t cut = fill;
      if (!eof) {  // back to the last line end (there is one)
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
  std::vector<std::thread> workers;
  for (unsigned t = 0; t < helpers; ++t)
    workers.emplace_back([&] {
      for (;;) {
        Piece* pc;
        {
          std::unique_lock<std::mutex> g(mu);
          cv.wait(g, [&] { return !todo.empty() || input_end; });
          if (todo.empty()) return;
          pc = todo.front();
          todo.pop_front();
        }
        parse(*pc);
        std::lock_guard<std::mutex> g(mu);
        done[pc->seq] = pc;
        cv.notify_all();
      }
    });
  for (uint64_t want = 0;; ++want) {
    Piece* pc;
    {
      std::unique_lock<std::mutex> g(mu);
      cv.wait(g, [&] { return done.count(want) || (input_end && want >= n_pieces); });
      if (!done.count(want)) break;
      pc = done[want];
      done.erase(want);
    }
    consume(*pc);
    std::lock_guard<std::mutex> g(mu);
    spare.push_back(pc);
    --in_flight;
    cv.notify_all();
  }
  reader.join();
  for (auto& w : workers) w.join();
}

#if PTS_MODE != 1
// one-output tools: the FASTQ text of a piece and its runs of equal RNAME
void format_piece(Piece& pc) {
  pc.arena.clear();
  pc.arena.reserve(pc.size + pc.size / 8);
  pc.chr_runs.clear();
  const char *p = pc.text.data(), *e = p + pc.size;
  Field f[11];
  std::string rs, rq;
  const char* cur = nullptr;
  size_t cur_len = 0;
  while (p < e) {
    const char* nl = rfxcli::find_nl(p, e);
    const char* le = nl ? nl : e;
    if (split_sam(p, le, f)) {
      if (!cur || f[2].n != cur_len || memcmp(f[2].p, cur, cur_len) != 0) {
        pc.chr_runs.emplace_back(f[2].p, f[2].n);
        cur = f[2].p;
        cur_len = f[2].n;
      }
      const char *sp = f[9].p, *qp = f[10].p;
      size_t sn = f[9].n, qn = f[10].n;
#if PTS_MODE == 2
      if (sam_flag(f[1]) & 16) {
        revcomp_into(rs, f[9]);
        reverse_into(rq, f[10]);
        sp = rs.data(); sn = rs.size();
        qp = rq.data(); qn = rq.size();
      }
#endif
      pc.arena.push_back('@');
      pc.arena.append(f[0].p, f[0].n);
      pc.arena.push_back('\n');
      pc.arena.append(sp, sn);
      pc.arena.append("\n+\n", 3);
      pc.arena.append(qp, qn);
      pc.arena.push_back('\n');
    }
    p = nl ? nl + 1 : e;
  }
}
#endif

#if PTS_MODE == 1

void parse_piece(Piece& pc) {
  pc.arena.clear();
  pc.arena.reserve(pc.size + 16);
  pc.recs.clear();
  const char *p = pc.text.data(), *e = p + pc.size;
  Field f[11];
  std::string rs, rq;
  while (p < e) {
    const char* nl = rfxcli::find_nl(p, e);
    const char* le = nl ? nl : e;
    if (split_sam(p, le, f)) {  // fewer than 11 fields: skipped (the reference reads past the line there)
      Parsed r;
      r.name = f[0].p; r.name_len = (uint32_t)f[0].n;
      r.chr = f[2].p; r.chr_len = (uint32_t)f[2].n;
      r.seq = f[9].p; r.seq_len = (uint32_t)f[9].n;
      r.qual = f[10].p; r.qual_len = (uint32_t)f[10].n;
      if (sam_flag(f[1]) & 16) {
        revcomp_into(rs, f[9]);
        reverse_into(rq, f[10]);
        const size_t at = pc.arena.size();
        pc.arena.append(rs);  // (within the reserved capacity: the pointers below stay valid)
        pc.arena.append(rq);
        r.seq = pc.arena.data() + at; r.seq_len = (uint32_t)rs.size();
        r.qual = pc.arena.data() + at + rs.size(); r.qual_len = (uint32_t)rq.size();
      }
      r.hash = name_hash(r.name, r.name_len);
      pc.recs.push_back(r);
    }
    p = nl ? nl + 1 : e;
  }
}
#endif

}  // namespace

int main(int argc, char** argv) {
  if (argc < (PTS_MODE == 1 ? 3 : 2)) {
    printf("ERROR, Output file could not be opened -%s\n", argc > 1 ? argv[1] : "");
    return 0;
  }
  FILE* chr = fopen(argv[1], "w");
  if (!chr) {
    printf("ERROR, Output file could not be opened -%s\n", argv[1]);
    return 0;
  }
#if PTS_MODE == 1
  FILE* m1 = fopen((std::string(argv[2]) + ".mate1.fastq").c_str(), "w");
  FILE* m2 = fopen((std::string(argv[2]) + ".mate2.fastq").c_str(), "w");
  if (!m1 || !m2) {
    printf("ERROR, Output file could not be opened -%s\n", argv[1]);
    return 0;
  }
  // Two named pipes read in lock step by RUFUS.Filter (the reference alternates: four lines from one, four from the
  // other).  Both sides are written in chunks that hold the SAME pairs and stay well below a pipe's 64 KB: the
  // reader can always consume everything earlier chunks brought, so neither write can wait for the other pipe.
  // (The first version flushed both pipes after every pair: two system calls per pair were most of its time.)
  setvbuf(m1, nullptr, _IONBF, 0);
  setvbuf(m2, nullptr, _IONBF, 0);
  std::string out1, out2;
  size_t CHUNK = 24u << 10;
#ifdef F_SETPIPE_SZ
  // bigger pipes, bigger chunks (a quarter of the pipe: the argument above holds for any chunk below its capacity)
  {
    const int c1 = fcntl(fileno(m1), F_SETPIPE_SZ, 1 << 20), c2 = fcntl(fileno(m2), F_SETPIPE_SZ, 1 << 20);
    if (c1 >= (1 << 20) && c2 >= (1 << 20) && !getenv("RFX_PTS_SMALL_CHUNKS")) CHUNK = 256u << 10;
  }
#endif
  // The two writes of a chunk happen on a writer thread (when there are helpers at all): a full pipe then stops the
  // writer, not the pairing.
  std::mutex wmu;
  std::condition_variable wcv;
  std::deque<std::pair<std::string, std::string>> wq;
  bool w_end = false, w_on = false;
  std::thread writer;
  auto write_chunk = [&](const std::string& a, const std::string& b) {
    if (!a.empty()) fwrite(a.data(), 1, a.size(), m1);
    if (!b.empty()) fwrite(b.data(), 1, b.size(), m2);
  };
  auto flush_pairs = [&]() {
    if (!w_on) {
      write_chunk(out1, out2);
    } else {
      std::unique_lock<std::mutex> g(wmu);
      wcv.wait(g, [&] { return wq.size() < 4; });
      wq.emplace_back(std::move(out1), std::move(out2));
      wcv.notify_all();
    }
    out1.clear();
    out2.clear();
  };
  auto start_writer = [&]() {
    w_on = true;
    writer = std::thread([&] {
      for (;;) {
        std::pair<std::string, std::string> c;
        {
          std::unique_lock<std::mutex> g(wmu);
          wcv.wait(g, [&] { return !wq.empty() || w_end; });
          if (wq.empty()) return;
          c = std::move(wq.front());
          wq.pop_front();
          wcv.notify_all();
        }
        write_chunk(c.first, c.second);
      }
    });
  };
  auto stop_writer = [&]() {
    if (!w_on) return;
    {
      std::lock_guard<std::mutex> g(wmu);
      w_end = true;
      wcv.notify_all();
    }
    writer.join();
    w_on = false;
  };
  auto put_text = [](std::string& o, const char* name, size_t nn, const char* seq, size_t ls, const char* qual, size_t lq) {
    o.push_back('@');
    o.append(name, nn);
    o.push_back('\n');
    o.append(seq, ls);
    o.append("\n+\n", 3);
    o.append(qual, lq);
    o.push_back('\n');
  };
  // A chunk never grows past CHUNK: it is flushed BEFORE a pair that would not fit (a record may be tens of KB: a
  // chunk of CHUNK - 1 bytes plus one such record would not fit a 64 KB pipe, and the alternating reader would wait
  // on the other pipe for ever); a pair that is larger than CHUNK by itself travels alone -- mate 1 then mate 2,
  // the order the reader consumes them in, which is what the reference does for every pair.
  auto add_pair = [&](const char* name, size_t nn, const char* s1, size_t ls1, const char* q1, size_t lq1, const char* s2,
                      size_t ls2, const char* q2, size_t lq2) {
    const size_t a = nn + ls1 + lq1 + 6, b = nn + ls2 + lq2 + 6;
    if ((!out1.empty() || !out2.empty()) && (out1.size() + a > CHUNK || out2.size() + b > CHUNK)) flush_pairs();
    put_text(out1, name, nn, s1, ls1, q1, lq1);
    put_text(out2, name, nn, s2, ls2, q2, lq2);
    if (out1.size() >= CHUNK || out2.size() >= CHUNK) flush_pairs();
  };
  Waiting waiting;
  // (measured on the 16-CPU GPU box, 32 M reads into drained pipes: 0 / 4 / 8 helpers = 6.7 / 12.4 / 8.9 M reads/s --
  // the pairing thread is the limit from four on, and the filter behind the pipes wants CPUs too)
  unsigned helpers = std::min(4u, rfxcli::usable_cpus() > 2 ? rfxcli::usable_cpus() - 2 : 0u);
  if (const char* ev = getenv("RFX_PTS_THREADS")) helpers = (unsigned)std::max(0, atoi(ev));
  if (helpers > 0) {
    start_writer();
    std::string current = "notachr";
    process_stream(helpers, parse_piece, [&](Piece& pc) {
      for (const Parsed& r : pc.recs) {
        if (r.chr_len != current.size() || memcmp(r.chr, current.data(), r.chr_len) != 0) {
          fprintf(chr, "%s\n", current.c_str());
          current.assign(r.chr, r.chr_len);
        }
        const long at = waiting.find(r.hash, r.name, r.name_len);
        if (at < 0) {
          waiting.insert(r.hash, r.name, r.name_len, r.seq, r.seq_len, r.qual, r.qual_len);
        } else {
          const Waiting::Entry& en = waiting.pool[waiting.slot[(size_t)at] - 2];
          add_pair(r.name, r.name_len, r.seq, r.seq_len, r.qual, r.qual_len, en.bytes.data() + en.name_len, en.seq_len,
                   en.bytes.data() + en.name_len + en.seq_len, en.bytes.size() - en.name_len - en.seq_len);
          waiting.erase(at);
        }
      }
    });
    fprintf(chr, "%s\n", current.c_str());
    fclose(chr);
    flush_pairs();
    stop_writer();
    fclose(m1);
    fclose(m2);
    return 0;
  }
#endif
#if PTS_MODE != 1
  {
    unsigned helpers = std::min(4u, rfxcli::usable_cpus() > 2 ? rfxcli::usable_cpus() - 2 : 0u);
    if (const char* ev = getenv("RFX_PTS_THREADS")) helpers = (unsigned)std::max(0, atoi(ev));
    if (helpers > 0) {
      std::string current = "notachr";
      process_stream(helpers, format_piece, [&](Piece& pc) {
        for (const std::string& name : pc.chr_runs)
          if (name != current) {
            fprintf(chr, "%s\n", current.c_str());
            current = name;
          }
        fwrite(pc.arena.data(), 1, pc.arena.size(), stdout);
      });
      fprintf(chr, "%s\n", current.c_str());
      fclose(chr);
      fflush(stdout);
      return 0;
    }
  }
#endif
  rfxcli::LineReader in;
  in.attach(0);
  std::string current = "notachr", rs, rq;
  const char *b, *e;
  Field f[11];
  while (in.getline(b, e)) {
    if (!split_sam(b, e, f)) continue;  // fewer than 11 fields: the reference reads past the line here
    if (f[2].n != current.size() || memcmp(f[2].p, current.data(), f[2].n) != 0) {
      fprintf(chr, "%s\n", current.c_str());
      current.assign(f[2].p, f[2].n);
    }
#if PTS_MODE == 0
    put_record(stdout, f[0], f[9].p, f[9].n, f[10].p, f[10].n);
#else
    const bool reverse = (sam_flag(f[1]) & 16) != 0;
    const char *sp = f[9].p, *qp = f[10].p;
    size_t sn = f[9].n, qn = f[10].n;
    if (reverse) {
      revcomp_into(rs, f[9]);
      reverse_into(rq, f[10]);
      sp = rs.data(); sn = rs.size();
      qp = rq.data(); qn = rq.size();
    }
#if PTS_MODE == 2
    put_record(stdout, f[0], sp, sn, qp, qn);
#else
    const uint64_t nh = name_hash(f[0].p, f[0].n);
    const long at = waiting.find(nh, f[0].p, f[0].n);
    if (at < 0) {
      waiting.insert(nh, f[0].p, f[0].n, sp, sn, qp, qn);
    } else {
      const Waiting::Entry& en = waiting.pool[waiting.slot[(size_t)at] - 2];
      add_pair(f[0].p, f[0].n, sp, sn, qp, qn, en.bytes.data() + en.name_len, en.seq_len,
               en.bytes.data() + en.name_len + en.seq_len, en.bytes.size() - en.name_len - en.seq_len);
      waiting.erase(at);
    }
#endif
#endif
  }
  fprintf(chr, "%s\n", current.c_str());
  fclose(chr);
#if PTS_MODE == 1
  flush_pairs();
  fclose(m1);
  fclose(m2);
#endif
  return 0;
}
