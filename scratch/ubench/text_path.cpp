// How fast can FASTQ text get from the page cache to the device?  (round 6, before wiring rfx_text_* into the ingest)
//   ./text_path FILE [threads]
// 1. memcpy from a mapping into pinned buffers (threads), 2. pread into pinned buffers, 3. H2D from pinned,
// 4. hipHostRegister of the mapping + H2D straight from it.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const char* path = argv[1];
  const int nt = argc > 2 ? atoi(argv[2]) : 16;
  int fd = open(path, O_RDONLY);
  struct stat st;
  fstat(fd, &st);
  const size_t size = (size_t)st.st_size, PIECE = 32u << 20;
  const size_t np = (size + PIECE - 1) / PIECE;
  printf("file %.2f GB, %d threads\n", size / 1e9, nt);
  std::vector<char*> pin((size_t)nt);
  double t0 = now();
  for (auto& p : pin) hipHostMalloc((void**)&p, PIECE, hipHostMallocDefault);
  printf("pin %d x 32 MB: %.3f s\n", nt, now() - t0);
  char* dev;
  hipMalloc((void**)&dev, PIECE * (size_t)nt);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int pass = 0; pass < 2; ++pass) {
    char* m = (char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    madvise(m, size, MADV_SEQUENTIAL);
    std::atomic<size_t> next{0};
    t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= np) break;
          const size_t n = std::min(PIECE, size - i * PIECE);
          memcpy(pin[(size_t)t], m + i * PIECE, n);
        }
      });
    for (auto& x : th) x.join();
    double dt = now() - t0;
    printf("memcpy mapping -> pinned (pass %d): %.3f s = %.1f GB/s\n", pass, dt, size / dt / 1e9);
    munmap(m, size);
  }
  {
    std::atomic<size_t> next{0};
    t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= np) break;
          const size_t n = std::min(PIECE, size - i * PIECE);
          size_t got = 0;
          while (got < n) {
            ssize_t r = pread(fd, pin[(size_t)t] + got, n - got, (off_t)(i * PIECE + got));
            if (r <= 0) break;
            got += (size_t)r;
          }
        }
      });
    for (auto& x : th) x.join();
    double dt = now() - t0;
    printf("pread -> pinned: %.3f s = %.1f GB/s\n", dt, size / dt / 1e9);
  }
  {
    // pipeline: threads copy a piece into their pinned buffer, then H2D it and wait (one stream per thread)
    char* m = (char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    std::atomic<size_t> next{0};
    t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        hipStream_t st_;
        hipStreamCreate(&st_);
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= np) break;
          const size_t n = std::min(PIECE, size - i * PIECE);
          memcpy(pin[(size_t)t], m + i * PIECE, n);
          hipMemcpyAsync(dev + (size_t)t * PIECE, pin[(size_t)t], n, hipMemcpyHostToDevice, st_);
          hipStreamSynchronize(st_);
        }
        hipStreamDestroy(st_);
      });
    for (auto& x : th) x.join();
    double dt = now() - t0;
    printf("memcpy + H2D per thread: %.3f s = %.1f GB/s\n", dt, size / dt / 1e9);
    munmap(m, size);
  }
  {
    t0 = now();
    const int reps = (int)std::min<size_t>(np, 256);
    for (int i = 0; i < reps; ++i) hipMemcpyAsync(dev + (size_t)(i % nt) * PIECE, pin[(size_t)(i % nt)], PIECE, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    double dt = now() - t0;
    printf("H2D from pinned, one stream: %.1f GB/s\n", reps * (double)PIECE / dt / 1e9);
  }
  {
    char* m = (char*)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
    const size_t chunk = std::min<size_t>(size, 2ull << 30);
    t0 = now();
    hipError_t e = hipHostRegister(m, chunk, hipHostRegisterReadOnly);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(m, chunk, hipHostRegisterDefault); }
    double dt = now() - t0;
    printf("hipHostRegister of %.1f GB of the mapping: %s, %.3f s = %.1f GB/s\n", chunk / 1e9, hipGetErrorString(e), dt, chunk / dt / 1e9);
    if (e == hipSuccess) {
      t0 = now();
      for (size_t at = 0; at + PIECE <= chunk; at += PIECE) hipMemcpyAsync(dev, m + at, PIECE, hipMemcpyHostToDevice, s);
      hipStreamSynchronize(s);
      dt = now() - t0;
      printf("H2D straight from the registered mapping: %.1f GB/s\n", chunk / dt / 1e9);
      hipHostUnregister(m);
    }
    munmap(m, size);
  }
  return 0;
}
