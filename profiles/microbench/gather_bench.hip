// Random-gather micro-benchmark: the "HBM random-access roofline" of SURVEY.md section 8d(ii).
// Every lane performs dependent-free 16-byte loads from uniformly random 64-byte-aligned slots of a
// table as large as the C2 k-mer hash (4 GiB); reports G loads/s and the implied 64 B-sector GB/s.
// build: hipcc -O3 --offload-arch=gfx950 gather_bench.hip -o gather_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void gather(const uint4* __restrict__ tab, unsigned long long mask, int iters, unsigned long long* sink) {
  unsigned long long x = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL + 12345;
  unsigned long long acc = 0;
  for (int i = 0; i < iters; i += 4) {
    unsigned long long a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 29; a[j] = (x & mask) * 4; }
    uint4 v0 = tab[a[0]], v1 = tab[a[1]], v2 = tab[a[2]], v3 = tab[a[3]];
    acc += v0.x + v1.y + v2.z + v3.w;
  }
  if (acc == 0x1234567) *sink = acc;
}

int main(int argc, char** argv) {
  size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (4ULL << 30);
  uint4* tab; unsigned long long* sink;
  hipMalloc(&tab, bytes); hipMemset(tab, 1, bytes); hipMalloc(&sink, 8);
  unsigned long long lines = bytes / 64, mask = 1; while (mask * 2 <= lines) mask *= 2; mask -= 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpb : {4, 8, 16}) for (int blocksPerCU : {4, 8}) {
    int blocks = 256 * blocksPerCU, threads = 64 * wpb, iters = 256;
    gather<<<blocks, threads>>>(tab, mask, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather<<<blocks, threads>>>(tab, mask, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double loads = (double)blocks * threads * iters;
    printf("table %.1f GiB waves/block %d blocks/CU %d: %.2f G loads/s = %.1f GB/s of 64B sectors (%.3f ms)\n",
           bytes / 1073741824.0, wpb, blocksPerCU, loads / ms / 1e6, loads * 64 / ms / 1e6, ms);
  }
  return 0;
}
