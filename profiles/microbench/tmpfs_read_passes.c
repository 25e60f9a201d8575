// tmpfs_read_passes.c -- why is the FIRST pass over a freshly written tmpfs file so much slower than the second?
// Writes a file of N MB with one thread (like numpy.tofile), then reads it T-threaded several times, either with pread()
// into per-thread buffers or through a fresh mmap (MADV_POPULATE_READ per 2 MB slice), summing the bytes so that every
// cache line is touched.  Prints GB/s per pass and the CPU seconds spent.
//   gcc -O2 -pthread tmpfs_read_passes.c -o /tmp/trp && /tmp/trp /dev/shm/x.bin 2048 32 pread|mmap [writers]
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static const char* path; static size_t len; static int T, use_mmap; static const char* map; static volatile uint64_t sink;
static size_t next_chunk; static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
#define CH ((size_t)2 << 20)
static void* reader(void* a) {
  int fd = open(path, O_RDONLY); char* buf = malloc(CH); uint64_t s = 0;
  while (1) {
    pthread_mutex_lock(&mu); size_t c = next_chunk++; pthread_mutex_unlock(&mu);
    size_t off = c * CH; if (off >= len) break;
    size_t n = len - off < CH ? len - off : CH;
    const char* p;
    if (use_mmap) { p = map + off; madvise((void*)p, n, MADV_POPULATE_READ); }
    else { size_t h = 0; while (h < n) { ssize_t r = pread(fd, buf + h, n - h, off + h); if (r <= 0) break; h += r; } p = buf; }
    for (size_t i = 0; i < n; i += 64) s += (unsigned char)p[i];
  }
  sink += s; close(fd); free(buf); return 0;
}
struct wr { size_t a, b; };
static void* writer(void* a) {
  struct wr* w = a; int fd = open(path, O_WRONLY); char* buf = malloc(CH); memset(buf, 'A', CH);
  for (size_t off = w->a; off < w->b; off += CH) { size_t n = w->b - off < CH ? w->b - off : CH; pwrite(fd, buf, n, off); }
  close(fd); free(buf); return 0;
}
int main(int argc, char** argv) {
  path = argv[1]; len = (size_t)atol(argv[2]) << 20; T = atoi(argv[3]); use_mmap = !strcmp(argv[4], "mmap");
  int W = argc > 5 ? atoi(argv[5]) : 1;
  unlink(path);
  int fd = open(path, O_CREAT | O_RDWR, 0644); close(fd);
  double t = now();
  { pthread_t th[256]; struct wr w[256];
    for (int i = 0; i < W; ++i) { w[i].a = len / W * i / CH * CH; w[i].b = i == W - 1 ? len : len / W * (i + 1) / CH * CH; pthread_create(&th[i], 0, writer, &w[i]); }
    for (int i = 0; i < W; ++i) pthread_join(th[i], 0); }
  printf("write (%d writer%s): %.2f GB/s\n", W, W > 1 ? "s" : "", len / (now() - t) / 1e9);
  for (int pass = 0; pass < 4; ++pass) {
    if (use_mmap) { fd = open(path, O_RDONLY); map = mmap(0, len, PROT_READ, MAP_PRIVATE, fd, 0); close(fd); }
    next_chunk = 0;
    struct timespec c0, c1; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &c0);
    t = now();
    pthread_t th[256];
    for (int i = 0; i < T; ++i) pthread_create(&th[i], 0, reader, 0);
    for (int i = 0; i < T; ++i) pthread_join(th[i], 0);
    double dt = now() - t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &c1);
    printf("%s pass %d, %d threads: %.2f GB/s, cpu %.2f s\n", use_mmap ? "mmap" : "pread", pass, T, len / dt / 1e9, (c1.tv_sec - c0.tv_sec) + 1e-9 * (c1.tv_nsec - c0.tv_nsec));
    if (use_mmap) munmap((void*)map, len);
  }
  unlink(path);
  return 0;
}
