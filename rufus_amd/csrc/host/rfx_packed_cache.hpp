// Packed-read cache between `jellyfish count --sam .. --keep-packed FILE` and `RUFUS.Filter --packed FILE ..`
// (scripts/RunJellyForRUFUS.sh:28 parses the subject's SAM stream for the count, runRufus.sh:966
// parses the SAME stream again -- feeder, two FASTQ pipes, filter -- to pull the mutant pairs).
//
// The count's parser threads already walk every SAM line; with --keep-packed they also leave, per piece of the stream,
// a CHUNK of this file (put it in /dev/shm): every record as RUFUS.Filter would see it -- the sequence and quality the
// stranded feeder prints (src/PassThroughSamCheck.stranded.cpp:188-223: reverse-strand records reverse-complemented,
// qualities reversed), packed with RFX_PACK_FILTER for a given MinQ: 2-bit codes + "good" mask, 60 bytes per 150 bp --,
// a 64-bit hash of its QNAME and where its line lies in the stream (= in the spool file `--spool` writes, or in the SAM
// file itself).  The filter then does NOT parse the stream: it uploads the chunks as they are, scans them on the device,
// and touches text only for the records whose NAME (hash) belongs to a hit: their lines are gathered from the spool in
// stream order and go through the ordinary `--sam` route (exact pairing by name, exact formatting) -- ~10^4 pairs of a
// 30x sample instead of 6 * 10^8 records.
//
// Exactness does not rest on the cache being right in every corner: a record the producer cannot pack exactly like the
// text route would (a base outside ACGTN, a quality string of another length than its sequence, an empty sequence) is
// flagged ALWAYS and handed to the text route whatever the scan says; a name-hash collision only adds lines to the
// gathered text (the text route pairs by the names themselves).  What is gathered is a superset of the records of every
// name with a hit, in stream order, so pairs, mates and output order are those of the full text route.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rufus_hip.h"
#include "rfx_sam.hpp"

namespace rfxcache {

constexpr uint64_t FILE_MAGIC = 0x31484341434B5052ull;   // "RPKCACH1"
constexpr uint64_t CHUNK_MAGIC = 0x314B4E5548434B50ull;  // "PKCHUNK1"
constexpr uint8_t REC_ALWAYS = 1;  // not packed exactly: always handed to the text route

constexpr uint64_t DONE_MAGIC = 0x454E4F4448434143ull;   // "CACHDONE": the producer finished the file
// The header is written twice: at the start without `done`, and again when the count has parsed the whole stream --
// with the number of chunks and the length of the stream they describe.  A cache whose producer died on the way, or one
// left over from another stream, is then NOT a cache (read_chunks), however many well-formed chunks it holds.
struct FileHeader {
  uint64_t magic = FILE_MAGIC;
  int32_t min_q = 0;
  uint32_t reserved = 0;
  uint64_t done = 0;          // DONE_MAGIC once the producer has written everything
  uint64_t n_chunks = 0;      // pieces of the stream = chunks in the file
  uint64_t stream_bytes = 0;  // length of the stream (spool file / SAM file) the line offsets refer to
  uint64_t pad[3] = {0, 0, 0};
};

// A chunk = one piece of the stream.  Arrays follow the header in this order, each padded to 8 bytes:
//   hash u64[n] | line_off u32[n] (from the piece's first byte) | line_len u32[n] | flags u8[n] |
//   len u32[n] | word_off u32[n + 1] | codes u64[n_words] | good u32[n_words] | chr runs ('\n' after each name)
struct ChunkHeader {
  uint64_t magic = CHUNK_MAGIC;
  uint64_t seq = 0;        // position of the piece in the stream
  uint64_t stream_off = 0; // byte offset of the piece in the stream (spool file / SAM file)
  uint64_t bytes = 0;      // of the whole chunk, header included
  uint32_t n = 0, n_words = 0;
  uint32_t runs_bytes = 0, reserved = 0;
};

inline size_t pad8(size_t x) { return (x + 7) & ~(size_t)7; }

class ChunkBuilder {
  std::vector<uint64_t> hash_;
  std::vector<uint32_t> off_, llen_;
  std::vector<uint8_t> flags_;
  // where the printed sequence / quality of a record lie: forward-strand records are packed straight out of the
  // piece's text (offsets from the piece's first byte), reverse-strand ones from `side_` (offsets into it, marked)
  std::vector<uint64_t> so_, qo_;
  std::vector<uint32_t> sl_;
  std::vector<uint8_t> in_side_;
  std::vector<char> side_;
  const char* piece_ = nullptr;

 public:
  // one SAM line [line, line_end) of the piece starting at `piece`: QNAME, FLAG, SEQ, QUAL as located by the caller
  void add(const char* piece, const char* line, const char* line_end, const char* name, const char* name_end,
           const char* flag, const char* flag_end, const char* seq, const char* seq_end, const char* qual,
           const char* qual_end) {
    using namespace rfxsam;
    piece_ = piece;
    hash_.push_back(name_hash(name, (size_t)(name_end - name)));
    off_.push_back((uint32_t)(line - piece));
    llen_.push_back((uint32_t)(line_end - line));
    const size_t ls = (size_t)(seq_end - seq), lq = (size_t)(qual_end - qual);
    // plain: only A C G T N (the feeder's reverse complement drops anything else), as many qualities as bases
    static const struct PlainLut {
      unsigned char t[256];
      PlainLut() {
        memset(t, 1, sizeof t);
        t['A'] = t['C'] = t['G'] = t['T'] = t['N'] = 0;
      }
    } lut;
    unsigned bad = ls == 0 || ls != lq;
    for (size_t i = 0; i < ls; ++i) bad |= lut.t[(unsigned char)seq[i]];
    if (bad) {  // the text route decides
      flags_.push_back(REC_ALWAYS);
      sl_.push_back(0);
      so_.push_back(0);
      qo_.push_back(0);
      in_side_.push_back(0);
      return;
    }
    flags_.push_back(0);
    sl_.push_back((uint32_t)ls);
    if (sam_flag(Field{flag, (size_t)(flag_end - flag)}) & 16) {  // as the stranded feeder prints it
      const size_t at = side_.size();
      side_.resize(at + 2 * ls);
      char* w = side_.data() + at;
      for (size_t j = 0; j < ls; ++j) w[j] = (char)g_comp.t[(unsigned char)seq[ls - 1 - j]];
      for (size_t j = 0; j < ls; ++j) w[ls + j] = qual[ls - 1 - j];
      so_.push_back(at);
      qo_.push_back(at + ls);
      in_side_.push_back(1);
    } else {
      so_.push_back((uint64_t)(seq - piece));
      qo_.push_back((uint64_t)(qual - piece));
      in_side_.push_back(0);
    }
  }

  void finish(uint64_t seq, uint64_t stream_off, const std::vector<std::string>& runs, int min_q, std::vector<char>& out) {
    const uint32_t n = (uint32_t)hash_.size();
    uint64_t words = 0;
    for (uint32_t l : sl_) words += (l + 31) / 32;
    std::string rb;
    for (const std::string& r : runs) {
      rb += r;
      rb.push_back('\n');
    }
    ChunkHeader h;
    h.seq = seq;
    h.stream_off = stream_off;
    h.n = n;
    h.n_words = (uint32_t)words;
    h.runs_bytes = (uint32_t)rb.size();
    const size_t o_hash = sizeof h, o_off = o_hash + (size_t)n * 8, o_llen = o_off + pad8((size_t)n * 4),
                 o_flags = o_llen + pad8((size_t)n * 4), o_len = o_flags + pad8(n), o_woff = o_len + pad8((size_t)n * 4),
                 o_codes = o_woff + pad8(((size_t)n + 1) * 4), o_good = o_codes + (size_t)words * 8,
                 o_runs = o_good + pad8((size_t)words * 4), total = o_runs + pad8(rb.size());
    h.bytes = total;
    out.assign(total, 0);
    memcpy(out.data(), &h, sizeof h);
    if (n) {
      memcpy(out.data() + o_hash, hash_.data(), (size_t)n * 8);
      memcpy(out.data() + o_off, off_.data(), (size_t)n * 4);
      memcpy(out.data() + o_llen, llen_.data(), (size_t)n * 4);
      memcpy(out.data() + o_flags, flags_.data(), n);
      // (rfx_pack_spans takes byte distances from `base` modulo 2^64: the side buffer is another allocation than the piece)
      const uint64_t side_delta = (uint64_t)((uintptr_t)side_.data() - (uintptr_t)piece_);
      for (uint32_t i = 0; i < n; ++i)
        if (in_side_[i]) {
          so_[i] += side_delta;
          qo_[i] += side_delta;
        }
      uint32_t* woff = (uint32_t*)(out.data() + o_woff);
      woff[0] = 0;
      const int rc = rfx_pack_spans(piece_, so_.data(), sl_.data(), qo_.data(), n, min_q, RFX_PACK_FILTER,
                                    (uint64_t*)(out.data() + o_codes), nullptr, (uint32_t*)(out.data() + o_good), woff,
                                    (uint32_t*)(out.data() + o_len));
      if (rc != RFX_OK) rfxcli::die(std::string("packed-read cache: rfx_pack_spans: ") + rfx_strerror(rc));
    }
    memcpy(out.data() + o_runs, rb.data(), rb.size());
  }
};

// What the reader sees of a chunk (pointers into the mapped file).
struct ChunkView {
  const ChunkHeader* h = nullptr;
  const uint64_t* hash = nullptr;
  const uint32_t *line_off = nullptr, *line_len = nullptr;
  const uint8_t* flags = nullptr;
  const uint32_t *len = nullptr, *word_off = nullptr;
  const uint64_t* codes = nullptr;
  const uint32_t* good = nullptr;
  const char* runs = nullptr;
};

// Chunks of a mapped cache file in stream order; false: not a cache, not finished by its producer, truncated, not the
// cache of a stream of `stream_bytes` bytes, or inconsistent in itself (every offset the reader follows is checked here).
inline bool read_chunks(const char* base, size_t size, uint64_t stream_bytes, int& min_q, std::vector<ChunkView>& out) {
  if (size < sizeof(FileHeader)) return false;
  FileHeader fh;
  memcpy(&fh, base, sizeof fh);
  if (fh.magic != FILE_MAGIC || fh.done != DONE_MAGIC || fh.stream_bytes != stream_bytes) return false;
  min_q = fh.min_q;
  size_t at = sizeof fh;
  while (at + sizeof(ChunkHeader) <= size) {
    const ChunkHeader* h = (const ChunkHeader*)(base + at);
    if (h->magic != CHUNK_MAGIC) break;  // (the file may have been pre-sized: zeros behind the last chunk)
    if (h->bytes < sizeof *h || at + h->bytes > size) return false;
    const size_t n = h->n, words = h->n_words;
    ChunkView v;
    v.h = h;
    size_t o = at + sizeof *h;
    v.hash = (const uint64_t*)(base + o); o += n * 8;
    v.line_off = (const uint32_t*)(base + o); o += pad8(n * 4);
    v.line_len = (const uint32_t*)(base + o); o += pad8(n * 4);
    v.flags = (const uint8_t*)(base + o); o += pad8(n);
    v.len = (const uint32_t*)(base + o); o += pad8(n * 4);
    v.word_off = (const uint32_t*)(base + o); o += pad8((n + 1) * 4);
    v.codes = (const uint64_t*)(base + o); o += words * 8;
    v.good = (const uint32_t*)(base + o); o += pad8(words * 4);
    v.runs = base + o; o += pad8(h->runs_bytes);
    if (o != at + h->bytes) return false;
    // what the device and the gather will follow: word offsets inside the chunk's words, lines inside the stream
    if (n && (v.word_off[0] != 0 || v.word_off[n] != words)) return false;
    for (size_t r = 0; r < n; ++r) {
      if (v.word_off[r] > v.word_off[r + 1] || (uint64_t)(v.word_off[r + 1] - v.word_off[r]) * 32 < v.len[r]) return false;
      if (h->stream_off + (uint64_t)v.line_off[r] + v.line_len[r] > stream_bytes) return false;
    }
    out.push_back(v);
    at += h->bytes;
  }
  if (out.size() != fh.n_chunks) return false;  // the producer wrote more (or fewer) than the file holds
  std::sort(out.begin(), out.end(), [](const ChunkView& a, const ChunkView& b) { return a.h->seq < b.h->seq; });
  for (size_t i = 0; i < out.size(); ++i)
    if (out[i].h->seq != i) return false;  // a piece is missing
  return true;
}

}  // namespace rfxcache
