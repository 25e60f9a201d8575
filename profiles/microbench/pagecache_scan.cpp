// How fast can T threads read a file out of the page cache on this host, with nothing shared but an atomic chunk counter?
// mode m: mmap (one mapping, MADV_POPULATE_READ per chunk) then memchr over the chunk;  mode p: pread into a private buffer, memchr.
//   g++ -O3 -pthread pagecache_scan.cpp -o pagecache_scan && ./pagecache_scan FILE [chunkMB]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char** argv) {
  const char* path = argv[1]; const size_t CH = (argc > 2 ? atoi(argv[2]) : 8) * (size_t)1 << 20;
  int fd = open(path, O_RDONLY); struct stat st; fstat(fd, &st); const size_t len = (size_t)st.st_size;
  const char* map = (const char*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
  const size_t nch = (len + CH - 1) / CH;
  for (char mode : {'m', 'p', 'c'}) for (int T : {4, 8, 16, 32, 64, 128}) {
    if (mode == 'm') { munmap((void*)map, len); map = (const char*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0); }   // fresh page tables
    std::atomic<size_t> next{0}; std::atomic<size_t> lines{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&] {
      std::vector<char> buf; if (mode != 'm') buf.resize(CH);
      std::vector<char> dst; if (mode == 'c') dst.resize(CH);
      size_t ln = 0;
      for (size_t c; (c = next.fetch_add(1)) < nch;) {
        const size_t a = c * CH, n = std::min(CH, len - a);
        const char* p;
        if (mode == 'm') { madvise((void*)(map + a), n, MADV_POPULATE_READ); p = map + a; }
        else { size_t have = 0; while (have < n) { ssize_t r = pread(fd, buf.data() + have, n - have, (off_t)(a + have)); if (r <= 0) break; have += (size_t)r; } p = buf.data(); }
        for (const char* q = p; (q = (const char*)memchr(q, '\n', (size_t)(p + n - q))) != nullptr; ++q) ++ln;
        if (mode == 'c') memcpy(dst.data(), p, n);        // + a copy of the chunk (what the copy tasks add)
      }
      lines += ln;
    });
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %c threads %3d: %.3f s  %.1f GB/s  (%zu lines)\n", mode, T, dt, len / dt / 1e9, lines.load()); fflush(stdout);
  }
}
