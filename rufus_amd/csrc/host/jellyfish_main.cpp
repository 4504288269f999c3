// Drop-in `jellyfish` for the sub-commands RUFUS runs on its hot path, same argv and file formats:
//   count  --disk -m K -L L -s SIZE -t T -o OUT -C IN...      scripts/RunJellyForRUFUS.sh:29
//   histo  -f -o OUT DB                                        scripts/RunJellyForRUFUS.sh:37
//   query  -s FASTA DB                                         scripts/CheckJellyHashList.sh:12
//   dump   -c DB                                               scripts/Overlap.shorter.sh:247
//   merge  F1 F2 ...   (RUFUS's MODIFIED merge: prints the k-mers unique to one input, count >= 5)
//                                                              runRufus.sh:925, jf/jellyfish/merge_files.cc:69-155
// Install it at both $RDIR/bin/externals/jellyfish/src/jellyfish_project/bin/jellyfish and
// $RDIR/bin/externals/modified_jellyfish/src/modified_jellyfish_project/bin/jellyfish.
// Host code only parses text and moves bytes; counting, sorting, set difference and lookups run in
// the HIP kernels behind include/rufus_hip.h.  No CPU fallback.
#include <getopt.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <limits>
#include <memory>

#include <functional>

#include "rfx_ingest.hpp"

using namespace rfxcli;

static double secs_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---------------------------------------------------------------------------------------------
static int count_main(int argc, char** argv, int full_argc, char** full_argv) {
  const auto t_start = std::chrono::steady_clock::now();
  int k = 0, threads = 1, out_counter_len = 4;
  uint64_t size = 0, lower = 0, upper = std::numeric_limits<uint64_t>::max();
  bool canonical = false, size_given = false;
  const char* out = "mer_counts.jf";
  const char* timing = nullptr;
  // --sam CHR_FILE (not in jellyfish): the input is SAM text and this process does what
  // `PassThroughSamCheck CHR_FILE | jellyfish count` does (scripts/RunJellyForRUFUS.sh:28-29) -- the sequence of
  // every record counted, the chromosome log written -- without the FASTQ text in between: SURVEY 8 row N1.
  const char* sam_chr = nullptr;
  // --spool FILE (not in jellyfish): the bytes of a PIPE input are also written to FILE, so that the
  // stage that reads the same stream next (RUFUS.Filter on the subject, runRufus.sh:966) reads FILE instead of
  // running the subject's generator -- `samtools view` of a BAM -- a second time (`RUFUS.Filter --sam CHR HashList FILE ...`).
  const char* spool_path = nullptr;
  // --keep-packed FILE [--keep-minq Q] (not in jellyfish; SURVEY 8 row N2, with --sam): the parser threads also leave the
  // records as RUFUS.Filter would see them -- printed orientation, packed for MinQ Q (default 15), name hash, place in
  // the stream -- in FILE (rfx_packed_cache.hpp; put it in /dev/shm): `RUFUS.Filter --packed FILE CHR HashList SPOOL ...`
  // scans that instead of parsing the stream a second time.
  const char* keep_path = nullptr;
  int keep_minq = 15;
  enum { OPT_DISK = 1000, OPT_OCL, OPT_TIMING, OPT_TEXT, OPT_SAM, OPT_SPOOL, OPT_KEEP, OPT_KEEPQ };
  static option lo[] = {{"mer-len", 1, 0, 'm'},      {"size", 1, 0, 's'},        {"threads", 1, 0, 't'},
                        {"output", 1, 0, 'o'},       {"counter-len", 1, 0, 'c'}, {"out-counter-len", 1, 0, OPT_OCL},
                        {"canonical", 0, 0, 'C'},    {"disk", 0, 0, OPT_DISK},   {"lower-count", 1, 0, 'L'},
                        {"upper-count", 1, 0, 'U'},  {"timing", 1, 0, OPT_TIMING}, {"text", 0, 0, OPT_TEXT}, {"sam", 1, 0, OPT_SAM},
                        {"spool", 1, 0, OPT_SPOOL},  {"keep-packed", 1, 0, OPT_KEEP}, {"keep-minq", 1, 0, OPT_KEEPQ},
                        {"reprobes", 1, 0, 'p'},     {0, 0, 0, 0}};
  optind = 1;
  int ch;
  while ((ch = getopt_long(argc, argv, "m:s:t:o:c:CL:U:p:", lo, nullptr)) != -1) {
    switch (ch) {
      case 'm': k = atoi(optarg); break;
      case 's': if (!parse_si(optarg, size)) die(std::string("Invalid size '") + optarg + "'"); size_given = true; break;
      case 't': threads = atoi(optarg); break;
      case 'o': out = optarg; break;
      case 'c': break;  // in-memory counter width of the reference table: no meaning here
      case 'p': break;
      case 'C': canonical = true; break;
      case 'L': if (!parse_si(optarg, lower)) die("Invalid lower count"); break;
      case 'U': if (!parse_si(optarg, upper)) die("Invalid upper count"); break;
      case OPT_DISK: break;  // table growth / merging is internal
      case OPT_OCL: out_counter_len = atoi(optarg); break;
      case OPT_TIMING: timing = optarg; break;
      case OPT_SAM: sam_chr = optarg; break;
      case OPT_SPOOL: spool_path = optarg; break;
      case OPT_KEEP: keep_path = optarg; break;
      case OPT_KEEPQ: keep_minq = atoi(optarg); break;
      case OPT_TEXT: die("rufus_amd jellyfish: --text output is not on the RUFUS path");
      default: die("Usage: jellyfish count -m K -s SIZE [-C] [-L n] [-U n] [-t T] [-o OUT] [--disk] file...");
    }
  }
  if (k < 1 || !size_given) die("Missing required switch: -m, --mer-len and -s, --size");
  if (optind >= argc) die("Missing sequence file");
  int lsize = 0;
  while ((1ull << lsize) < size) ++lsize;
  if (lsize < 1) lsize = 1;

  // RUFUS_GPUS: one sample over several devices (include/rufus_hip.h rfx_count_set_peers) -- a device
  // gets every N-th read block and partitions it; the owners of the minimizer bins pull their records, the survivors
  // change hands by output position; the file is the devices' slices one after the other.  Needs the super-k-mer path
  // (23 <= k <= 31): else the first device alone.
  std::vector<int> gpus = gpu_list();
  if (gpus.size() > 1 && !(k >= 23 && k <= 31)) {
    fprintf(stderr, "rufus_amd jellyfish: k = %d is counted on one device (RUFUS_GPUS needs 23 <= k <= 31)\n", k);
    gpus.resize(1);
  }
  const int n_gpu = (int)gpus.size();
  // The devices are opened (HIP runtime start, context, arena: 0.1 - 0.25 s) on a thread of their own while this one
  // looks at the inputs, starts the output's page allocation and the parser threads: the staging blocks are plain memory
  // until their first upload (rfx_host_alloc_lazy / rfx_host_pin), so parsing needs no device.  Everything that touches
  // a ctx or a table goes through device_ready() first.  (Round 6: ~0.3 s of a 1.3 s count of 64 M reads.)
  std::vector<rfx_ctx*> ctxs;
  std::vector<rfx_table*> tabs;
  rfx_peers* peers = nullptr;
  rfx_ctx* ctx = nullptr;
  rfx_table* tab = nullptr;
  bool defer = false;  // (decided below, before the opener looks at it)
  std::atomic<bool> inputs_known{false};
  std::mutex open_mu;
  std::condition_variable open_cv;
  std::thread opener([&] {
    ctxs = open_ctxs(gpus);
    peers = n_gpu > 1 ? rfx_peers_create(n_gpu) : nullptr;
    for (int g = 0; g < n_gpu; ++g) {
      rfx_table* tb = rfx_count_begin(ctxs[g], k, canonical, lsize, 0, 0, 0);
      if (!tb) die(std::string("rufus_amd: ") + rfx_last_error());
      tabs.push_back(tb);
    }
    ctx = ctxs[0];
    tab = tabs[0];
    trace("count: device open, table made");
    {  // (the passes / peers settings depend on what the inputs are: wait for the main thread's look at them)
      std::unique_lock<std::mutex> g(open_mu);
      open_cv.wait(g, [&] { return inputs_known.load(); });
    }
    for (int g = 0; g < n_gpu; ++g) {
      if (defer && rfx_count_set_passes(tabs[g], 0) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
      if (peers && rfx_count_set_peers(tabs[g], peers, g) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
    }
  });
  auto device_ready = [&] {
    if (opener.joinable()) opener.join();
  };
  const bool sync_open = getenv("RFX_SYNC_OPEN") != nullptr;  // (A/B: the device opened before anything else, as until round 6)
  const auto t_init = std::chrono::steady_clock::now();

  // Inputs: regular files are mapped (their size is known), pipes are streamed.  When the input is a pipe
  // (its size is unknown: RUFUS feeds 30x genomes through FIFOs, scripts/RunJellyForRUFUS.sh:23-31) or a big
  // file, the packed read blocks stay in HBM and the table counts them in minimizer-shard passes at finish
  // (rfx_count_set_passes: bounded HBM, the analogue of --disk); small inputs are counted block by block.
  struct Input { std::string path; int fd; bool regular; size_t size; };
  std::vector<Input> inputs;
  size_t known_bytes = 0;
  bool any_stream = false;
  for (int i = optind; i < argc; ++i) {
    Input in{argv[i], -1, false, 0};
    in.fd = strcmp(argv[i], "stdin") == 0 || strcmp(argv[i], "/dev/stdin") == 0 ? 0 : ::open(argv[i], O_RDONLY);
    if (in.fd < 0) die(std::string("Failed to open input file '") + argv[i] + "'");
    struct stat st;
    if (fstat(in.fd, &st) == 0 && S_ISREG(st.st_mode)) {
      in.regular = true;
      in.size = (size_t)st.st_size;
      known_bytes += in.size;
    } else {
      any_stream = true;
    }
    inputs.push_back(in);
  }
  // ~11 bytes of output per distinct solid k-mer: 0.18 of the FASTQ bytes at 30x, less on shallow or filtered input
  OutputPrealloc prealloc;
  double prealloc_frac = 0;
  size_t prealloc_min = 0;
  {
    size_t min_bytes = 256u << 20;  // RFX_PREALLOC_MIN / RFX_PREALLOC_FRAC: the tests reach both sides of the guess
    double frac = 0.19;
    if (const char* ev = getenv("RFX_PREALLOC_MIN")) min_bytes = (size_t)strtoull(ev, nullptr, 10);
    if (const char* ev = getenv("RFX_PREALLOC_FRAC")) frac = atof(ev);
    if (!any_stream && known_bytes > min_bytes && !getenv("RFX_NO_PREALLOC"))
      prealloc.start(out, (uint64_t)((double)known_bytes * frac));
    // a piped input (scripts/RunJellyForRUFUS.sh:28: samtools view | ... | jellyfish count /dev/fd/0): the guess follows
    // the bytes that have come in (profiles/r03_cli_w_trio.txt: 8.5 s of page faults after the device had finished)
    else if (any_stream && inputs.size() == 1 && !getenv("RFX_NO_PREALLOC"))
      prealloc.start_growing(out);
    prealloc_frac = frac;
    prealloc_min = min_bytes;
  }
  if (keep_path && (!sam_chr || inputs.size() != 1))
    die("rufus_amd jellyfish count: --keep-packed goes with --sam and ONE input (a pipe with --spool, or a SAM file)");
  if (keep_path && any_stream && !spool_path)  // (the cache points into the stream: without a copy of it there is nothing to point into)
    die("rufus_amd jellyfish count: --keep-packed of a piped input needs --spool FILE (the cache refers to lines of the stream)");
  if (spool_path && (!any_stream || inputs.size() != 1))
    die("rufus_amd jellyfish count: --spool copies ONE piped input; a regular file can be given to the next stage as it is");
  unsigned nthreads = (unsigned)std::max(1, threads);
  nthreads = std::min(nthreads, rfx_host_cpus());  // -t 40 on a 16-CPU cgroup: 16 parsers
  if (const char* ev = getenv("RFX_HOST_THREADS")) nthreads = (unsigned)std::max(1, atoi(ev));
  const bool msp_ok = k >= 23 && k <= 31;  // the super-k-mer path (rfx_count_set_passes needs it)
  defer = msp_ok && (any_stream || known_bytes > (16ull << 30) || getenv("RFX_COUNT_PASSES"));
  if (const char* ev = getenv("RFX_COUNT_DEFER")) defer = msp_ok && atoi(ev) != 0;
  if (n_gpu > 1) defer = true;
  {
    std::lock_guard<std::mutex> g(open_mu);
    inputs_known = true;
    open_cv.notify_all();
  }
  if (sync_open) device_ready();
  std::vector<std::vector<rfx_reads*>> resident((size_t)n_gpu);
  // A sample counted in shard passes at finish needs ~3 bytes of device memory per byte of packed reads beside them
  // (records of a pass, survivors): it is mapped while the input is still parsed, a few GiB per uploaded block -- in a fresh
  // process mapping 130 GB took 0.6 s of the time between "input parsed" and "finished on the device".
  std::vector<uint64_t> reserved((size_t)n_gpu, 0);
  auto sink_to = [&](int g, rfx_reads* r) {
    device_ready();
    if (!r) die(std::string("rufus_amd: upload failed: ") + rfx_last_error());
    const int rc = rfx_count_add(tabs[g], r);
    if (rc) die(std::string("rufus_amd: count failed: ") + rfx_strerror(rc) + " " + rfx_last_error());
    if (defer) {
      resident[(size_t)g].push_back(r);
      if (!getenv("RFX_NO_RESERVE")) {
        uint64_t used = 0, peak = 0, mapped = 0;
        rfx_ctx* c = n_gpu == 1 ? ctx : ctxs[g];
        if (rfx_mem_stats(c, &used, &peak, &mapped) == RFX_OK) {
          const uint64_t want = std::min<uint64_t>(used * 4, 200ull << 30);
          if (want > reserved[(size_t)g] + (4ull << 30)) {  // (in steps of >= 4 GiB; a refusal is no error: mapped when needed)
            if (rfx_mem_reserve(c, want) == RFX_OK) reserved[(size_t)g] = want;
            else reserved[(size_t)g] = ~0ull >> 1;
          }
        }
      }
    } else {
      rfx_reads_free(r);
    }
  };
  // Several devices: block b of packed reads goes to device b mod N -- each device partitions its blocks, the owners of
  // the minimizer bins pull their record runs at finish (rfx_count_set_peers).  RFX_PEERS_REPLICATE=1: round 3's scheme,
  // every block to EVERY device (each over its own PCIe link: one thread per device).
  const bool replicate = getenv("RFX_PEERS_REPLICATE") != nullptr;
  std::atomic<uint64_t> next_block{0};
  auto to_all = [&](const std::function<rfx_reads*(rfx_ctx*)>& up) {
    device_ready();
    if (n_gpu == 1) {
      sink_to(0, up(ctx));
      return;
    }
    if (!replicate) {
      const int g = (int)(next_block.fetch_add(1) % (uint64_t)n_gpu);
      static std::mutex dev_mu[128];  // (one upload at a time per device: the table of a device is not re-entrant)
      std::lock_guard<std::mutex> lk(dev_mu[g]);
      sink_to(g, up(ctxs[g]));
      return;
    }
    std::vector<std::thread> th;
    for (int g = 0; g < n_gpu; ++g) th.emplace_back([&, g] { sink_to(g, up(ctxs[g])); });
    for (auto& t : th) t.join();
  };

  ReadBatch batch;
  PackedBatch packed;
  auto flush = [&]() {
    if (batch.n() == 0) return;
    int rc = packed.pack(batch, RFX_PACK_COUNT, 0);
    if (rc) die(std::string("rufus_amd: pack failed: ") + rfx_strerror(rc));
    to_all([&](rfx_ctx* c) { return packed.upload(c, batch.n(), RFX_PACK_COUNT); });
    batch.clear();
  };
  auto sequential = [&](LineReader& in) {  // the reference's grammar: FASTA, multi-line FASTQ
    const bool ok = parse_sequences(in, [&](const char* s, size_t n) {
      batch.add(s, n);
      if (batch.seq.size() >= (256u << 20) || batch.n() >= (1u << 22)) flush();
    });
    if (!ok) die("Unsupported format");
    flush();  // k-mers never span files
  };
  std::unique_ptr<CountIngest> ingest;  // (kept to the end: its page-locked blocks serve the output drain)
  // Round 6 (kernel K1): a regular FASTQ file counted on one device need not be parsed on the host at
  // all -- its text can go to the device as it lies and be parsed and packed there (rfx_ingest.hpp TextIngest,
  // csrc/rfx_text.hip): RFX_DEVICE_PARSE=1.  It is NOT the default, because on the GPU box it is no faster
  // (profiles/r06_text_route.txt, 20 GB of FASTQ, 16-CPU quota): 16 parser threads parse and pack at 30 - 35 GB/s and
  // upload 68 bytes per read; the text route has to move 330 bytes per read over PCIe and reaches ~30 GB/s with 8 MB
  // pieces from 16 threads on one copy stream -- a draw for one count (0.6 - 0.8 s of ingest either way), and a draw for the
  // three counts of `runRufus.sh -pj` at once (2.5 s both; with pread instead of the mapping the text route takes 4.1 s:
  // three processes' readers on one tmpfs file system get in each other's way).  The route stays: it is what a host
  // with fewer cores per GPU than this box wants (it needs a memcpy per read where the parser needs a microsecond).
  const bool text_ok = n_gpu == 1 && nthreads > 1 && !sam_chr && !spool_path && !keep_path;
  const bool device_text = text_ok && getenv("RFX_DEVICE_PARSE") && !getenv("RFX_HOST_PARSE");
  std::unique_ptr<TextIngest> text_ingest;
  {
    for (Input& in : inputs) {
      bool done = false;
      if (device_text && in.regular && in.size > 0) {
        if (!text_ingest) {
          device_ready();
          text_ingest.reset(new TextIngest(
              ctx, nthreads, [&](rfx_reads* r) { sink_to(0, r); },
              [&](const char* text, size_t n) {  // refused by the device (not strict 4-line FASTQ): the reference's grammar
                LineReader lr(n + (1u << 20));
                lr.preload(text, n);
                lr.close_input();
                sequential(lr);
              },
              (size_t)1 << 30, getenv("RFX_INGEST_PIECE") ? (size_t)std::max(1024ll, atoll(getenv("RFX_INGEST_PIECE"))) : (size_t)8 << 20));
          trace("count: text arenas open, copiers up");
        }
        if (!getenv("RFX_TEXT_PREAD")) {  // (the mapped route; RFX_TEXT_PREAD=1: the workers pread their ranges -- faster for one
                                          // count, 0.69 against 0.92 s per 20 GB, much slower for three at once)
          void* m = mmap(nullptr, in.size, PROT_READ, MAP_PRIVATE, in.fd, 0);
          if (m != MAP_FAILED) {
            (void)madvise(m, in.size, MADV_SEQUENTIAL);
            done = text_ingest->feed_mapped((const char*)m, in.size);
            trace("count: text fed");
            munmap(m, in.size);
            trace("count: input unmapped");
          }
        } else {
          done = text_ingest->feed_file(in.fd, in.size);
          trace("count: text fed");
        }
        if (done) {
          if (in.fd > 0) ::close(in.fd);
          continue;
        }
      }
      if (nthreads > 1 || sam_chr || spool_path || keep_path) {
        if (!ingest) {
          ingest.reset(new CountIngest(nthreads, [&](const StageBlock& b) {
            to_all([&](rfx_ctx* c) { return rfx_reads_upload(c, b.codes, b.acgt, nullptr, b.word_off, b.len, b.n_reads); });
          }, sync_open ? rfx_host_alloc : rfx_host_alloc_lazy, rfx_host_free, 4u << 20, 24ull << 20, sync_open ? nullptr : rfx_host_pin));
          ingest->set_sam(sam_chr != nullptr);
          if (spool_path) {
            const int sfd = ::open(spool_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (sfd < 0) die(std::string("Can't open spool file '") + spool_path + "'");
            ingest->set_spool(sfd);
          }
          if (keep_path) {
            const int kfd = ::open(keep_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (kfd < 0) die(std::string("Can't open the packed-read cache '") + keep_path + "'");
            ingest->set_keep_packed(kfd, keep_minq);
          }
          if (const char* ev = getenv("RFX_INGEST_PIECE")) ingest->set_piece_bytes((size_t)std::max(1024ll, atoll(ev)));
          if (prealloc.growing())
            ingest->on_stream_bytes = [&prealloc, prealloc_frac, prealloc_min](uint64_t so_far) {
              if (so_far > prealloc_min) prealloc.want((uint64_t)((double)so_far * prealloc_frac));
            };
          trace("count: staging blocks pinned, workers up");
        }
        if (in.regular) {
          if (in.size == 0) { if (in.fd > 0) ::close(in.fd); continue; }
          // mapped, not pread: measured on the 256-core box (20 GB FASTQ in tmpfs, 64 workers) 1.1 s against 3.2 s
          // with every worker pread-ing its range into a private buffer (RFX_INGEST_PREAD=1 keeps that path testable)
          if (getenv("RFX_INGEST_PREAD") && !sam_chr) {
            done = ingest->feed_file(in.fd, in.size);
          } else {
            void* m = mmap(nullptr, in.size, PROT_READ, MAP_PRIVATE, in.fd, 0);
            if (m != MAP_FAILED) {
              (void)madvise(m, in.size, MADV_SEQUENTIAL);
              done = ingest->feed_mapped((const char*)m, in.size);
              munmap(m, in.size);
            }
          }
          if (!done) {
            LineReader lr;
            lr.attach(in.fd);
            in.fd = -1;  // closed by the reader
            sequential(lr);
            done = true;
          }
        } else {
          std::vector<char> head(1u << 16);
          size_t got = 0;
          for (;;) {
            const ssize_t n = ::read(in.fd, head.data() + got, head.size() - got);
            if (n < 0 && errno == EINTR) continue;
            if (n < 0) die(std::string("read error on input: ") + strerror(errno));
            if (n == 0) break;
            got += (size_t)n;
            if (got == head.size()) break;
          }
          head.resize(got);
          done = got == 0 || ingest->feed_stream(in.fd, head);
          if (!done && spool_path) die("rufus_amd jellyfish count: --spool needs SAM (--sam) or strict 4-line FASTQ on the pipe");
          if (!done) {
            LineReader lr;
            lr.preload(head.data(), head.size());
            lr.attach(in.fd);
            in.fd = -1;
            sequential(lr);
            done = true;
          }
        }
      } else {
        LineReader lr;
        lr.attach(in.fd);
        in.fd = -1;
        sequential(lr);
      }
      if (in.fd > 0) ::close(in.fd);
    }
  }
  // the packed-read cache is a cache only from here on: its header now says how many chunks describe how long a stream
  if (keep_path && ingest) ingest->finish_keep_packed(inputs[0].regular ? inputs[0].size : ingest->spooled_bytes());
  if (sam_chr) {
    FILE* cf = fopen(sam_chr, "w");
    if (!cf) {  // src/PassThroughSamCheck.cpp:37-44
      printf("ERROR, Output file could not be opened -%s\n", sam_chr);
    } else {
      if (ingest)
        for (const std::string& name : ingest->chr_log()) fprintf(cf, "%s\n", name.c_str());
      else fprintf(cf, "notachr\n");
      fclose(cf);
    }
  }
  // (device-parsed input: no staging blocks to lend to the output drain -- its ring of page-locked buffers is made now,
  // beside the device's finish, instead of in front of the first write)
  std::vector<std::pair<char*, size_t>> out_ring;
  std::thread out_ring_pinner;
  if (!ingest && text_ingest) {
    trace(("count: text route, " + text_ingest->timing()).c_str());
    text_ingest.reset();  // (its page-locked text buffers go first)
    out_ring_pinner = std::thread([&out_ring] {
      for (int i = 0; i < 8; ++i)
        if (char* p = (char*)rfx_host_alloc(48u << 20)) out_ring.emplace_back(p, (size_t)48u << 20);
    });
  }
  device_ready();  // (an empty input: nobody asked for the device yet)
  const auto t_count = std::chrono::steady_clock::now();
  trace("count: input parsed and queued");
  if (getenv("RFX_CLI_TRACE")) {
    char msg[160];
    snprintf(msg, sizeof msg, "write: output pages wanted %llu, allocated %llu, in the page table %llu",
             (unsigned long long)prealloc.wanted(), (unsigned long long)prealloc.reached(), (unsigned long long)prealloc.populated());
    trace(msg);
  }

  // RFX_COUNT_HISTO=1 (opt-in, not jellyfish behaviour): also write OUT.histo, byte for byte what
  // `jellyfish histo -f -o OUT.histo OUT` would -- the count has the histogram anyway, and
  // scripts/RunJellyForRUFUS.sh:36-38 runs histo only when that file is missing (a 35 GB re-read saved per sample).
  const bool side_histo = getenv("RFX_COUNT_HISTO") != nullptr;
  std::vector<uint64_t> hist(side_histo ? RFX_HISTO_BINS : 0);
  std::vector<rfx_records*> recs((size_t)n_gpu, nullptr);
  if (n_gpu == 1) {
    recs[0] = rfx_count_finish(tab, lower, upper, side_histo ? hist.data() : nullptr);
  } else {  // the finishes meet at the survivor exchange: one thread each
    std::vector<std::vector<uint64_t>> hs((size_t)n_gpu, std::vector<uint64_t>(side_histo ? RFX_HISTO_BINS : 0));
    std::vector<std::thread> th;
    for (int g = 0; g < n_gpu; ++g)
      th.emplace_back([&, g] { recs[(size_t)g] = rfx_count_finish(tabs[g], lower, upper, side_histo ? hs[(size_t)g].data() : nullptr); });
    for (auto& t : th) t.join();
    if (side_histo)
      for (int g = 0; g < n_gpu; ++g)
        for (int i = 0; i < RFX_HISTO_BINS; ++i) hist[(size_t)i] += hs[(size_t)g][(size_t)i];
  }
  for (rfx_records* r : recs)
    if (!r) die(std::string("rufus_amd: finish failed: ") + rfx_last_error());
  rfx_records* rec = recs[0];
  trace("count: finished on the device");
  for (auto& v : resident)
    for (rfx_reads* r : v) rfx_reads_free(r);
  std::vector<uint64_t> cols(2 * (size_t)k);
  rfx_jf_matrix(lsize, k, cols.data());
  if (side_histo && rec) {
    const std::string hp = std::string(out) + ".histo";
    if (FILE* hf = fopen(hp.c_str(), "w")) {
      for (int i = 0; i < RFX_HISTO_BINS; ++i) fprintf(hf, "%d %llu\n", i, (unsigned long long)hist[(size_t)i]);
      fclose(hf);
    }
  }
  const int out_fd = prealloc.take();
  if (getenv("RFX_CLI_TRACE")) {
    char msg[128];
    snprintf(msg, sizeof msg, "write: %llu bytes preallocated, %llu in the mapping's page table",
             (unsigned long long)prealloc.reached(), (unsigned long long)prealloc.populated());
    trace(msg);
  }
  if (out_ring_pinner.joinable()) out_ring_pinner.join();
  write_jhash(out, recs, cols.data(), canonical, out_counter_len, full_argc, full_argv,
              ingest ? ingest->lend_buffers(48u << 20) : out_ring, out_fd, prealloc.reached(), prealloc.take_mapping());
  trace("count: output closed");
  if (!timing && !getenv("RFX_CLEAN_EXIT")) {
    // The device memory goes back BEFORE the process leaves (0.15 s for the 200 GB a 30x sample maps), not with the
    // kernel's teardown of an exited process: the tool that runs next then takes 1.86 instead of 2.0 s (`histo` of a 30x
    // database, three A/B pairs in one box) -- and on some boxes its first device allocation had waited 1 s (after a
    // 64 M-read count) or 6 s (after a 30x sample) for memory that was still on its way back; that wait could not be
    // reproduced at will, so this is a precaution with a measured small gain.  RFX_LEAVE_NO_CLOSE=1: as it was.
    if (!getenv("RFX_LEAVE_NO_CLOSE")) {
      for (rfx_ctx* c : ctxs) rfx_close(c);
      trace("count: device closed");
    }
    leave(0);
  }
  ingest.reset();
  for (auto& b : out_ring) rfx_host_free(b.first);
  for (rfx_records* r : recs) rfx_records_free(r);
  for (rfx_table* tb : tabs) rfx_count_free(tb);
  if (peers) rfx_peers_free(peers);
  for (rfx_ctx* c : ctxs) rfx_close(c);
  trace("count: closed");
  if (timing) {
    if (FILE* f = fopen(timing, "w")) {
      fprintf(f, "Init     %g\nCounting %g\nWriting  %g\n", std::chrono::duration<double>(t_init - t_start).count(),
              std::chrono::duration<double>(t_count - t_init).count(), secs_since(t_count));
      fclose(f);
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
static int histo_main(int argc, char** argv) {
  bool full = false;
  const char* out = nullptr;
  uint64_t low = 1, high = 10000, inc = 1;
  static option lo[] = {{"full", 0, 0, 'f'}, {"output", 1, 0, 'o'}, {"low", 1, 0, 'l'},      {"high", 1, 0, 'h'},
                        {"increment", 1, 0, 'i'}, {"threads", 1, 0, 't'}, {0, 0, 0, 0}};
  optind = 1;
  int ch;
  while ((ch = getopt_long(argc, argv, "fo:l:h:i:t:", lo, nullptr)) != -1) {
    switch (ch) {
      case 'f': full = true; break;
      case 'o': out = optarg; break;
      case 'l': parse_si(optarg, low); break;
      case 'h': parse_si(optarg, high); break;
      case 'i': parse_si(optarg, inc); break;
      case 't': break;
      default: die("Usage: jellyfish histo [-f] [-o OUT] db");
    }
  }
  if (optind >= argc) die("Missing database");
  if (low != 1 || high != 10000 || inc != 1)
    die("rufus_amd jellyfish histo: only the default --low 1 --high 10000 --increment 1 is supported");
  rfx_ctx* ctx = open_ctx();
  trace("histo: device open");
  std::vector<uint64_t> hist(RFX_HISTO_BINS, 0);
  rfx_records* rec = nullptr;
  JhashFile db;
  if (!db.open(argv[optind])) {  // not a regular file: loaded whole
    JhashHeader h;
    rec = load_records(ctx, argv[optind], h);
    if (rfx_records_histo(rec, hist.data()) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
  } else {  // 256 M records (5 GB of HBM) at a time, whatever the size of the database
    uint64_t per = 256ull << 20;
    if (const char* ev = getenv("RFX_HISTO_SLICE_RECORDS")) per = std::max<uint64_t>(1, strtoull(ev, nullptr, 10));
    std::vector<uint64_t> part_hist(RFX_HISTO_BINS);
    for (uint64_t at = 0; at < db.n; at += per) {
      rfx_records* part = db.load(ctx, at, std::min(db.n, at + per));
      if (rfx_records_histo(part, part_hist.data()) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
      for (int i = 0; i < RFX_HISTO_BINS; ++i) hist[(size_t)i] += part_hist[(size_t)i];
      rfx_records_free(part);
    }
  }
  trace("histo: counted");
  FILE* f = out ? fopen(out, "w") : stdout;
  if (!f) die(std::string("Error opening output file '") + out + "'");
  for (int i = 0; i < RFX_HISTO_BINS; ++i)  // jf/sub_commands/histo_main.cc:82-84
    if (hist[(size_t)i] > 0 || full) fprintf(f, "%d %llu\n", i, (unsigned long long)hist[(size_t)i]);
  if (out) fclose(f);
  leave(0);
  db.close();
  if (rec) rfx_records_free(rec);
  rfx_close(ctx);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// `jellyfish query [-s FASTA]... DB [mers...]` (jf/sub_commands/query_main.cc:44-51,:110-115), and, SURVEY row N3:
// SEVERAL databases in one call -- `jellyfish query -s FA -o OUT1 -o OUT2 -o OUT3 DB1 DB2 DB3` writes to OUTi exactly
// the lines `jellyfish query -s FA DBi` prints.  scripts/Overlap.shorter.sh:265-299 looks the same two k-mer lists up in
// the subject's and every control's 30x database with one process each (scripts/CheckJellyHashList.sh:12); here the
// k-mers are parsed, canonicalised and dealt to position ranges once, the device is opened once, and of every database
// only the ranges that got a k-mer are read.  (One -o, or none, with several databases: one line per k-mer with a
// count column per database.)  A positional argument after the first that names an existing file is a database.
static int query_main(int argc, char** argv) {
  std::vector<const char*> seq_files, outs;
  static option lo[] = {{"sequence", 1, 0, 's'}, {"output", 1, 0, 'o'}, {"load", 0, 0, 'l'}, {"no-load", 0, 0, 'L'},
                        {0, 0, 0, 0}};
  optind = 1;
  int ch;
  while ((ch = getopt_long(argc, argv, "s:o:lL", lo, nullptr)) != -1) {
    switch (ch) {
      case 's': seq_files.push_back(optarg); break;
      case 'o': outs.push_back(optarg); break;
      case 'l': case 'L': break;
      default: die("Usage: jellyfish query [-s FASTA] [-o OUT]... db [db...] [mers...]");
    }
  }
  if (optind >= argc) die("Missing database");
  std::vector<const char*> db_paths{argv[optind]}, mers;
  // (the reference takes `db mers...`; further databases are an extension -- an argument spelt like a mer stays a mer
  // even if the working directory holds a file of that name)
  auto looks_like_mer = [](const char* a) {
    const size_t n = strlen(a);
    if (n == 0 || n > 32) return false;
    for (size_t i = 0; i < n; ++i)
      if (!strchr("ACGTacgt", a[i])) return false;
    return true;
  };
  for (int i = optind + 1; i < argc; ++i) {
    if (!looks_like_mer(argv[i]) && ::access(argv[i], R_OK) == 0) db_paths.push_back(argv[i]);
    else mers.push_back(argv[i]);
  }
  const size_t n_db = db_paths.size();
  if (outs.size() > 1 && outs.size() != n_db) die("jellyfish query: one -o per database (or a single one)");
  rfx_ctx* ctx = open_ctx();
  std::vector<JhashFile> dbs(n_db);
  std::vector<rfx_records*> whole(n_db, nullptr);
  std::vector<bool> sliced(n_db, false);
  for (size_t d = 0; d < n_db; ++d) {
    sliced[d] = dbs[d].open(db_paths[d]);
    if (!sliced[d]) whole[d] = load_records(ctx, db_paths[d], dbs[d].h);  // (a pipe: header and payload in one pass)
    if (dbs[d].h.k != dbs[0].h.k || dbs[d].h.canonical != dbs[0].h.canonical)
      die("jellyfish query: the databases of one call must have the same mer length and canonical flag");
  }
  const JhashHeader& h = dbs[0].h;
  const int k = h.k;
  const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  std::vector<uint64_t> keys;
  // every k-mer of the query sequences (jf/sub_commands/query_main.cc:44-51), canonicalised when the
  // database is (:115); the lookups themselves run on the device
  for (const char* path : seq_files) {
    LineReader in;
    if (!in.open(path)) die(std::string("Failed to open input file '") + path + "'");
    const bool ok = parse_sequences(in, [&](const char* s, size_t n) {
      uint64_t fwd = 0, rc = 0;
      int filled = 0;
      for (size_t i = 0; i < n; ++i) {
        uint64_t one;
        if (!text_to_key(s + i, 1, one)) { filled = 0; continue; }
        fwd = ((fwd << 2) | one) & kmask;
        rc = (rc >> 2) | ((3 - one) << (2 * (k - 1)));
        if (filled < k) ++filled;
        if (filled >= k) keys.push_back(h.canonical && rc < fwd ? rc : fwd);
      }
    });
    if (!ok) die("Unsupported format");
  }
  for (const char* m : mers) {  // query_from_cmdline
    uint64_t key;
    if ((int)strlen(m) != k || !text_to_key(m, (size_t)k, key)) {
      fprintf(stderr, "Invalid mer '%s'\n", m);
      continue;
    }
    keys.push_back(h.canonical ? std::min(key, revcomp_key(key, k)) : key);
  }
  if (keys.size() > 0xFFFFFFFFull) die("rufus_amd jellyfish query: too many k-mers in one call");
  std::vector<std::vector<uint32_t>> counts(n_db, std::vector<uint32_t>(keys.size() + 1, 0));
  std::vector<uint64_t> key_pos;  // position of every query, computed once per hash function:
  int key_pos_lsize = -1;         // the function key_pos holds now -- NOT that of the database before this one, which
  std::vector<uint64_t> key_pos_cols;  // may have been piped or empty and never have touched key_pos
  for (size_t d = 0; d < n_db; ++d) {
    const JhashHeader& hd = dbs[d].h;
    JhashFile& db = dbs[d];
    if (!sliced[d]) {
      if (rfx_query(whole[d], keys.data(), keys.size(), counts[d].data()) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
      continue;
    }
    if (keys.empty() || !db.n) continue;
    // The database stays on disk: it is cut into position ranges of ~64 M records (RFX_QUERY_SLICE_RECORDS), the
    // queries are dealt to the ranges by their own position, and only ranges that got queries are read -- a few
    // hundred MB of HBM whatever the database size.
    uint64_t per = 64ull << 20;
    if (const char* ev = getenv("RFX_QUERY_SLICE_RECORDS")) per = std::max<uint64_t>(1, strtoull(ev, nullptr, 10));
    const uint64_t S = std::max<uint64_t>(1, (db.n + per - 1) / per);
    if (key_pos.size() != keys.size() || hd.lsize != key_pos_lsize || hd.cols != key_pos_cols) {
      key_pos.resize(keys.size());
      for (size_t i = 0; i < keys.size(); ++i) key_pos[i] = rfx_jf_pos(hd.cols.data(), hd.k, hd.lsize, keys[i]);
      key_pos_lsize = hd.lsize;
      key_pos_cols.assign(hd.cols.begin(), hd.cols.end());
    }
    // (a located position costs ~30 single-record reads, a streamed record 0.3 ns: worth it below one query per few
    // thousand records -- RFX_QUERY_SPARSE_RATIO, default 4096)
    uint64_t sparse_ratio = 4096;
    if (const char* ev = getenv("RFX_QUERY_SPARSE_RATIO")) sparse_ratio = std::max<uint64_t>(1, strtoull(ev, nullptr, 10));
    if (keys.size() * sparse_ratio < db.n && !getenv("RFX_QUERY_NO_SPARSE")) {
      // Few k-mers against a big database (the hash-list lookup of runRufus.sh:925-926: thousands against 10^8..10^10
      // records): every position range would get one, i.e. the whole file would travel.  Instead the records AT the
      // queried positions are located on the host (a binary search per distinct position over the sorted file -- what
      // binary_dumper.hpp:156-203 does for every lookup), read, and uploaded as one small sorted database; the lookups
      // still run on the device.
      std::vector<uint64_t> ps(key_pos);
      std::sort(ps.begin(), ps.end());
      ps.erase(std::unique(ps.begin(), ps.end()), ps.end());
      std::vector<std::pair<uint64_t, uint64_t>> rg(ps.size());
      const unsigned nt = std::max(1u, std::min(16u, rfx_host_cpus()));
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
          for (size_t j = t; j < ps.size(); j += nt) {
            const uint64_t i0 = db.lower_bound_pos(ps[j]);
            uint64_t i1 = i0;
            while (i1 < db.n && db.pos_at(i1) == ps[j]) ++i1;
            rg[j] = {i0, i1};
          }
        });
      for (auto& x : th) x.join();
      uint64_t tot = 0;
      for (auto& r : rg) tot += r.second - r.first;
      if (tot == 0) continue;  // nothing stored at any queried position: every count is 0
      std::vector<char> buf((size_t)tot * db.rl);
      size_t at = 0;
      for (auto& r : rg) {
        size_t want = (size_t)(r.second - r.first) * db.rl, got = 0;
        while (got < want) {
          const ssize_t w = ::pread(db.fd, buf.data() + at + got, want - got, (off_t)(hd.payload_offset + r.first * db.rl + got));
          if (w < 0 && errno == EINTR) continue;
          if (w <= 0) die("read error on '" + db.path + "'");
          got += (size_t)w;
        }
        at += want;
      }
      rfx_records* part = rfx_records_load(ctx, hd.k, hd.lsize, hd.cols.data(), buf.data(), tot, hd.counter_len);
      if (!part) die("rufus_amd: cannot load '" + db.path + "': " + rfx_last_error());
      if (rfx_query(part, keys.data(), keys.size(), counts[d].data()) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
      rfx_records_free(part);
      continue;
    }
    std::vector<std::vector<uint32_t>> in_slice(S);
    for (size_t i = 0; i < keys.size(); ++i) in_slice[slice_of(key_pos[i], S, hd.lsize)].push_back((uint32_t)i);
    std::vector<uint64_t> sub;
    std::vector<uint32_t> got;
    for (uint64_t sidx = 0; sidx < S; ++sidx) {
      if (in_slice[sidx].empty()) continue;
      const uint64_t i0 = sidx == 0 ? 0 : db.lower_bound_pos(slice_start(sidx, S, hd.lsize));
      const uint64_t i1 = sidx + 1 == S ? db.n : db.lower_bound_pos(slice_start(sidx + 1, S, hd.lsize));
      if (i1 == i0) continue;  // an empty range: its queries count 0
      rfx_records* part = db.load(ctx, i0, i1);
      sub.clear();
      for (uint32_t qi : in_slice[sidx]) sub.push_back(keys[qi]);
      got.assign(sub.size() + 1, 0);
      if (rfx_query(part, sub.data(), sub.size(), got.data()) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
      for (size_t j = 0; j < sub.size(); ++j) counts[d][in_slice[sidx][j]] = got[j];
      rfx_records_free(part);
    }
  }
  if (outs.size() > 1) {  // one file per database: the lines a one-database call prints
    for (size_t d = 0; d < n_db; ++d) {
      FILE* f = fopen(outs[d], "w");
      if (!f) die(std::string("Error opening output file '") + outs[d] + "'");
      for (size_t i = 0; i < keys.size(); ++i) fprintf(f, "%s %u\n", key_to_text(keys[i], k).c_str(), counts[d][i]);
      fclose(f);
    }
  } else {
    FILE* f = outs.empty() ? stdout : fopen(outs[0], "w");
    if (!f) die(std::string("Error opening output file '") + outs[0] + "'");
    for (size_t i = 0; i < keys.size(); ++i) {
      fputs(key_to_text(keys[i], k).c_str(), f);
      for (size_t d = 0; d < n_db; ++d) fprintf(f, " %u", counts[d][i]);
      fputc('\n', f);
    }
    if (!outs.empty()) fclose(f);
  }
  leave(0);
  for (size_t d = 0; d < n_db; ++d) {
    dbs[d].close();
    if (whole[d]) rfx_records_free(whole[d]);
  }
  rfx_close(ctx);
  return 0;
}

// ---------------------------------------------------------------------------------------------
static int dump_main(int argc, char** argv) {
  bool column = false, tab = false;
  const char* out = nullptr;
  uint64_t lower = 0, upper = std::numeric_limits<uint64_t>::max();
  static option lo[] = {{"column", 0, 0, 'c'},      {"tab", 0, 0, 't'},         {"lower-count", 1, 0, 'L'},
                        {"upper-count", 1, 0, 'U'}, {"output", 1, 0, 'o'},      {0, 0, 0, 0}};
  optind = 1;
  int ch;
  while ((ch = getopt_long(argc, argv, "ctL:U:o:", lo, nullptr)) != -1) {
    switch (ch) {
      case 'c': column = true; break;
      case 't': tab = true; break;
      case 'L': parse_si(optarg, lower); break;
      case 'U': parse_si(optarg, upper); break;
      case 'o': out = optarg; break;
      default: die("Usage: jellyfish dump [-c] [-t] [-L n] [-U n] [-o OUT] db");
    }
  }
  if (optind >= argc) die("Missing database");
  rfx_ctx* ctx = open_ctx();
  JhashHeader h;
  rfx_records* rec = load_records(ctx, argv[optind], h);
  const uint64_t n = rfx_records_size(rec);
  std::vector<uint64_t> keys(n + 1);
  std::vector<uint32_t> counts(n + 1);
  if (rfx_records_get(rec, keys.data(), counts.data(), nullptr) != RFX_OK) die(std::string("rufus_amd: ") + rfx_last_error());
  FILE* f = out ? fopen(out, "w") : stdout;
  if (!f) die(std::string("Error opening output file '") + out + "'");
  for (uint64_t i = 0; i < n; ++i) {  // jf/sub_commands/dump_main.cc:36-53
    if (counts[i] < lower || counts[i] > upper) continue;
    if (column) fprintf(f, "%s%c%u\n", key_to_text(keys[i], h.k).c_str(), tab ? '\t' : ' ', counts[i]);
    else fprintf(f, ">%u\n%s\n", counts[i], key_to_text(keys[i], h.k).c_str());
  }
  if (out) fclose(f);
  leave(0);
  rfx_records_free(rec);
  rfx_close(ctx);
  return 0;
}

// ---------------------------------------------------------------------------------------------
static int merge_main(int argc, char** argv, int full_argc, char** full_argv) {
  const char* out = "mer_counts_merged.jf";
  static option lo[] = {{"output", 1, 0, 'o'}, {"lower-count", 1, 0, 'L'}, {"upper-count", 1, 0, 'U'}, {0, 0, 0, 0}};
  optind = 1;
  int ch;
  while ((ch = getopt_long(argc, argv, "o:L:U:", lo, nullptr)) != -1) {
    if (ch == 'o') out = optarg;
    else if (ch != 'L' && ch != 'U') die("Usage: jellyfish merge [-o OUT] db1 db2 ...");
  }
  if (optind >= argc) die("Missing database");
  rfx_ctx* ctx = open_ctx();
  const int nf = argc - optind;
  std::vector<JhashFile> jf((size_t)nf);
  std::vector<JhashHeader> hs((size_t)nf);
  std::vector<rfx_records*> whole((size_t)nf, nullptr);  // inputs that are not regular files: loaded as before
  bool all_sliced = true;
  for (int i = 0; i < nf; ++i) {
    if (!jf[(size_t)i].open(argv[optind + i])) {
      all_sliced = false;
      whole[(size_t)i] = load_records(ctx, argv[optind + i], hs[(size_t)i]);
    } else {
      hs[(size_t)i] = jf[(size_t)i].h;
    }
  }
  for (size_t i = 1; i < hs.size(); ++i) {  // jf/jellyfish/merge_files.cc:193-203
    if (hs[i].k != hs[0].k) die("Can't merge hashes of different key lengths");
    if (hs[i].lsize != hs[0].lsize) die("Can't merge hash with different size");
    if (hs[i].cols != hs[0].cols) die("Can't merge hash with different hash function");
  }
  // formatted by hand: printf + std::string per line is 10x slower, and the list can have 1e8 lines
  std::vector<char> line((size_t)1 << 20);
  auto emit = [&](const uint64_t* keys, const uint32_t* counts, uint64_t n) {
    size_t fill = 0;
    const int kk = hs[0].k;
    for (uint64_t i = 0; i < n; ++i) {
      if (fill + (size_t)kk + 16 > line.size()) {
        fwrite(line.data(), 1, fill, stdout);
        fill = 0;
      }
      for (int b = 0; b < kk; ++b) line[fill++] = "ACGT"[(keys[i] >> (2 * (kk - 1 - b))) & 3u];
      line[fill++] = '\t';
      char dig[12];
      int nd = 0;
      uint32_t v = counts[i];
      do { dig[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
      while (nd) line[fill++] = dig[--nd];
      line[fill++] = '\n';
    }
    fwrite(line.data(), 1, fill, stdout);
  };
  // The result is a small part of the inputs (k-mers private to one sample): room for 64 M of them first, the
  // exact number -- which a short call reports -- if that was not enough.  (Sizing by the inputs would be 115 GB
  // of host memory for a 30x trio.)
  std::vector<uint64_t> keys;
  std::vector<uint32_t> counts;
  auto merge_and_emit = [&](const std::vector<rfx_records*>& files) {
    uint64_t total = 0, n = 0;
    for (auto* r : files) total += rfx_records_size(r);
    uint64_t cap = std::min<uint64_t>(total, 64ull << 20);
    if (keys.size() < cap + 1) keys.resize(cap + 1), counts.resize(cap + 1);
    int rc = rfx_merge_unique(ctx, files.data(), (int)files.size(), 5, keys.data(), counts.data(), cap, &n);
    if (rc == RFX_E_RANGE && n > cap) {
      cap = n;
      keys.resize(cap + 1);
      counts.resize(cap + 1);
      rc = rfx_merge_unique(ctx, files.data(), (int)files.size(), 5, keys.data(), counts.data(), cap, &n);
    }
    if (rc) die(std::string("rufus_amd: merge failed: ") + rfx_strerror(rc) + " " + rfx_last_error());
    emit(keys.data(), counts.data(), n);
  };
  if (!all_sliced) {
    for (int i = 0; i < nf; ++i)
      if (!whole[(size_t)i]) whole[(size_t)i] = jf[(size_t)i].load(ctx, 0, jf[(size_t)i].n);
    merge_and_emit(whole);
  } else {
    // A merge-path join over position ranges (the files are sorted by position first): range by range, the part
    // of every input that falls into it is read, joined on the device and printed -- ranges come in increasing
    // order, so the output is the same sorted list.  ~2.4e9 records (48 GB of HBM) per range: one range for
    // anything small, 5 for a 30x trio (RFX_MERGE_SLICES overrides; a range costs ~0.6 s of fixed work: three 30x
    // samples of a 1 Gb genome, 3.1e9 records, merge in 1.8 s as one range and in 3.6 s as three).
    uint64_t total = 0;
    for (auto& f : jf) total += f.n;
    uint64_t S = std::max<uint64_t>(1, (total + 2399999999ull) / 2400000000ull);
    if (const char* ev = getenv("RFX_MERGE_SLICES")) S = std::max<uint64_t>(1, strtoull(ev, nullptr, 10));
    std::vector<uint64_t> lo((size_t)nf, 0);
    for (uint64_t sidx = 0; sidx < S; ++sidx) {
      std::vector<rfx_records*> part((size_t)nf, nullptr);
      for (int i = 0; i < nf; ++i) {
        const JhashFile& f = jf[(size_t)i];
        const uint64_t hi = sidx + 1 == S ? f.n : f.lower_bound_pos(slice_start(sidx + 1, S, hs[0].lsize));
        part[(size_t)i] = f.load(ctx, lo[(size_t)i], hi);
        lo[(size_t)i] = hi;
      }
      trace("merge: a position range of every input loaded");
      merge_and_emit(part);
      for (auto* r : part) rfx_records_free(r);
      trace("merge: joined and printed");
    }
  }
  fflush(stdout);
  // the reference leaves a header-only database behind (merge_files.cc:207-224; testRun/clean.sh:1 removes it)
  std::vector<char> hdr(1 << 16);
  const long hl = rfx_jhash_header(hs[0].k, hs[0].lsize, hs[0].cols.data(), hs[0].canonical, hs[0].counter_len,
                                   full_argc, full_argv, hdr.data(), hdr.size());
  if (hl > 0)
    if (FILE* f = fopen(out, "wb")) {
      fwrite(hdr.data(), 1, (size_t)hl, f);
      fclose(f);
    }
  leave(0);
  for (auto* r : whole)
    if (r) rfx_records_free(r);
  for (auto& f : jf) f.close();
  rfx_close(ctx);
  return 0;
}

int main(int argc, char** argv) {
  trace("main");
  if (argc < 2) die("Usage: jellyfish <count|histo|query|dump|merge> [options]");
  const std::string cmd = argv[1];
  if (cmd == "count") return count_main(argc - 1, argv + 1, argc, argv);
  if (cmd == "histo") return histo_main(argc - 1, argv + 1);
  if (cmd == "query") return query_main(argc - 1, argv + 1);
  if (cmd == "dump") return dump_main(argc - 1, argv + 1);
  if (cmd == "merge") return merge_main(argc - 1, argv + 1, argc, argv);
  if (cmd == "--version" || cmd == "-V") {
    printf("jellyfish 2.2.5 (rufus_amd drop-in: %s)\n", rfx_version());
    return 0;
  }
  die("Unknown sub-command '" + cmd + "' (rufus_amd provides count, histo, query, dump, merge)");
}
