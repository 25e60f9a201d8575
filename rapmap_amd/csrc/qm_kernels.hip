// qm_kernels.hip -- gfx950 kernels of libqmap_mi355.so and their launch wrappers.
//
//   qm_map_kernel<NS>     one 64-lane wavefront per read pair (qm_mapper.inl); persistent
//                         grid, 4 waves per workgroup, per-wave LDS slab, integer only.
//   build_sainfo_kernel   index flattening: (transcript id, offset) for every SA entry
//                         (replaces rank9b::rank + txpOffsets lookups on the hot path,
//                         src/rank9b.cpp:56-61, src/RapMapSAIndex.cpp:92-94)
//   build_slots_kernel    index flattening: open-addressing k-mer table from hash.bin records
//   gather_hits_kernel    bump-allocated hits -> CSR order
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

#include "qm_mapper.inl"
#include "qm_device.h"

namespace qm {

// WPS = minimum waves per SIMD the register allocator must leave room for
template <int NS, int WPS>
__global__ __launch_bounds__(256, WPS) void qm_map_kernel(DevIndex ix, Batch B) {
  __shared__ WaveMem<NS> mem[4];
  const int wave = threadIdx.x >> 6;
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const long long nw = (long long)gridDim.x * 4;
  u64* gscr = B.gscratch + gw * (4 * QM_GCAP);
  WaveCounters wc = {0, 0, 0, 0, 0, 0};
  for (long long unit = gw; unit < B.n; unit += nw) map_unit<NS>(ix, B, unit, mem[wave], gscr, wc);
  if ((threadIdx.x & 63) == 0) {
    if (wc.pe) atomicAdd(&B.counters[0], wc.pe);
    if (wc.se) atomicAdd(&B.counters[1], wc.se);
    if (wc.tot) atomicAdd(&B.counters[2], wc.tot);
    if (wc.reads) atomicAdd(&B.counters[3], wc.reads);
    if (wc.tooMany) atomicAdd(&B.counters[4], wc.tooMany);
    if (wc.mapped) atomicAdd(&B.counters[5], wc.mapped);
  }
}

__global__ void build_sainfo_kernel(const int* SA, long long nSA, const int* offsets, long long T, SaInfo* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < nSA; i += stride) {
    int p = SA[i];
    long long lo = 0, hi = T;   // upper_bound(offsets, p) - 1  == #'$' in text[0,p)
    while (lo < hi) { long long mid = (lo + hi) >> 1; if (offsets[mid] <= p) lo = mid + 1; else hi = mid; }
    long long tid = lo - 1;
    SaInfo e; e.tid = (u32)tid; e.pos = p - offsets[tid];
    out[i] = e;
  }
}

// records: K x {u64 key, i32 lb, i32 ub} exactly as streamed from hash.bin
__global__ void build_slots_kernel(const Slot* recs, long long K, Slot* slots, u64 hmask) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < K; i += stride) {
    Slot r = recs[i];
    u64 j = hash_mix(r.key) & hmask;
    while (true) {
      u64 prev = atomicCAS((unsigned long long*)&slots[j].key, ~0ULL, r.key);
      if (prev == ~0ULL) { slots[j].lb = r.lb; slots[j].ub = r.ub; break; }
      j = (j + 1) & hmask;
    }
  }
}

__global__ void gather_hits_kernel(long long n, const u32* cnt, const long long* tmp_off, const long long* offs,
                                   const qm_hit* tmp, qm_hit* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 c = cnt[i];
  const uint4* s = reinterpret_cast<const uint4*>(tmp + tmp_off[i]);
  uint4* d = reinterpret_cast<uint4*>(out + offs[i]);
  for (u32 j = 0; j < 2 * c; ++j) d[j] = s[j];
}

struct U32ToI64 { __device__ long long operator()(u32 x) const { return (long long)x; } };

}  // namespace qm

using namespace qm;

extern "C" {

hipError_t qmk_build_sainfo(const int* SA, long long nSA, const int* offsets, long long T, void* out, hipStream_t st) {
  hipLaunchKernelGGL(build_sainfo_kernel, dim3(4096), dim3(256), 0, st, SA, nSA, offsets, T, (SaInfo*)out);
  return hipGetLastError();
}

hipError_t qmk_build_slots(const void* recs, long long K, void* slots, unsigned long long cap, hipStream_t st) {
  hipError_t e = hipMemsetAsync(slots, 0xff, cap * sizeof(Slot), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(build_slots_kernel, dim3(4096), dim3(256), 0, st, (const Slot*)recs, K, (Slot*)slots, cap - 1);
  return hipGetLastError();
}

int qmk_map_grid(long long n, int num_cu) {
  long long want = (n + 3) / 4;
  long long cap = (long long)num_cu * QMK_BLOCKS_PER_CU;
  return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}

hipError_t qmk_map(const void* ixp, const void* bp, int ns, int grid, hipStream_t st) {
  const DevIndex& ix = *(const DevIndex*)ixp;
  const Batch& B = *(const Batch*)bp;
  static int wps = -1;
  if (wps < 0) { const char* e = getenv("QM_WPS"); wps = e ? atoi(e) : 3; }
  if (ns == 2) {
    if (wps == 4) hipLaunchKernelGGL((qm_map_kernel<2, 4>), dim3(grid), dim3(256), 0, st, ix, B);
    else if (wps == 5) hipLaunchKernelGGL((qm_map_kernel<2, 5>), dim3(grid), dim3(256), 0, st, ix, B);
    else if (wps == 6) hipLaunchKernelGGL((qm_map_kernel<2, 6>), dim3(grid), dim3(256), 0, st, ix, B);
    else if (wps == 8) hipLaunchKernelGGL((qm_map_kernel<2, 8>), dim3(grid), dim3(256), 0, st, ix, B);
    else hipLaunchKernelGGL((qm_map_kernel<2, 3>), dim3(grid), dim3(256), 0, st, ix, B);
  } else {
    hipLaunchKernelGGL((qm_map_kernel<4, 2>), dim3(grid), dim3(256), 0, st, ix, B);
  }
  return hipGetLastError();
}

size_t qmk_scan_temp_bytes(long long n) {
  size_t bytes = 0;
  auto it = rocprim::make_transform_iterator((const u32*)nullptr, U32ToI64());
  (void)rocprim::exclusive_scan(nullptr, bytes, it, (long long*)nullptr, 0LL, (size_t)n, rocprim::plus<long long>());
  return bytes;
}

hipError_t qmk_scan_counts(void* temp, size_t temp_bytes, const u32* cnt, long long* offs, long long n, hipStream_t st) {
  auto it = rocprim::make_transform_iterator(cnt, U32ToI64());
  return rocprim::exclusive_scan(temp, temp_bytes, it, offs, 0LL, (size_t)n, rocprim::plus<long long>(), st);
}

hipError_t qmk_gather(long long n, const u32* cnt, const long long* tmp_off, const long long* offs, const void* tmp,
                      void* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  int blocks = (int)((n + 255) / 256);
  hipLaunchKernelGGL(gather_hits_kernel, dim3(blocks), dim3(256), 0, st, n, cnt, tmp_off, offs, (const qm_hit*)tmp,
                     (qm_hit*)out);
  return hipGetLastError();
}

}  // extern "C"
