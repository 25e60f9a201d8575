/* how does zlib inflate of independent BGZF blocks scale with threads on this host?  (profiles/r03: the ingest engine's BGZF helpers)
 *   gcc -O2 -pthread -o /tmp/bgzf_scal profiles/microbench/bgzf_inflate_scaling.c -lz && /tmp/bgzf_scal FILE.bgzf.gz [sequential_advice]
 * every thread takes groups of 16 blocks off a shared counter and inflates them into a private 2 MiB buffer */
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
#include <zlib.h>
static const unsigned char* map; static size_t len; static size_t* offs; static long nblk; static long next_grp; static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void* work(void* a) {
  z_stream zs; memset(&zs, 0, sizeof(zs)); inflateInit2(&zs, 15 + 16);
  /* COLD=1: the output walks through a pre-faulted 256 MiB area per thread (allocated and touched by main before the clock starts)
     instead of staying in one cache-resident 2 MiB buffer */
  const size_t AREA = getenv("COLD") ? ((size_t)256 << 20) : ((size_t)2 << 20);
  unsigned char* area = (unsigned char*)a; size_t ao = 0; size_t tot = 0;
  for (;;) {
    pthread_mutex_lock(&mu); long g = next_grp++; pthread_mutex_unlock(&mu);
    long b0 = g * 16, b1 = b0 + 16 < nblk ? b0 + 16 : nblk;
    if (b0 >= nblk) break;
    unsigned char* out = area + ao; ao += (size_t)2 << 20; if (ao + ((size_t)2 << 20) > AREA) ao = 0;
    size_t o = 0;
    for (long b = b0; b < b1; ++b) {
      inflateReset(&zs);
      zs.next_in = (Bytef*)(map + offs[b]); zs.avail_in = (uInt)(offs[b + 1] - offs[b]); zs.next_out = out + o; zs.avail_out = (uInt)((2 << 20) - o);
      if (inflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "bad block %ld\n", b); exit(1); }
      o = (size_t)(zs.next_out - out);
    }
    tot += o;
  }
  if (getenv("DBG")) fprintf(stderr, "thread got %zu bytes\n", tot);
  inflateEnd(&zs);
  return (void*)tot;
}
int main(int argc, char** argv) {
  int fd = open(argv[1], O_RDONLY); struct stat st; fstat(fd, &st); len = (size_t)st.st_size;
  map = mmap(0, len, PROT_READ, MAP_PRIVATE, fd, 0);
  if (argc > 2) madvise((void*)map, len, MADV_SEQUENTIAL);
  offs = malloc(sizeof(size_t) * (len / 28 + 2)); size_t p = 0; nblk = 0;
  while (p + 18 <= len) { size_t bs = ((size_t)map[p + 16] | ((size_t)map[p + 17] << 8)) + 1; offs[nblk++] = p; p += bs; }
  offs[nblk] = p;
  int ths[] = {1, 4, 8, 16, 32, 64, 128};
  for (int k = 0; k < 7; ++k) {
    int T = ths[k]; pthread_t th[128]; next_grp = 0;
    const size_t AREA = getenv("COLD") ? ((size_t)256 << 20) : ((size_t)2 << 20);
    static unsigned char* areas[128];
    for (int i = 0; i < T; ++i) if (!areas[i]) { areas[i] = malloc(AREA); memset(areas[i], 1, AREA); }
    double t0 = now(); size_t tot = 0;
    for (int i = 0; i < T; ++i) pthread_create(&th[i], 0, work, areas[i]);
    for (int i = 0; i < T; ++i) { void* r; pthread_join(th[i], &r); tot += (size_t)r; }
    double dt = now() - t0;
    printf("%3d threads: %.3f s  %.2f GB/s out (%.0f MB/s per thread)\n", T, dt, tot / dt / 1e9, tot / dt / 1e6 / T);
  }
  return 0;
}
