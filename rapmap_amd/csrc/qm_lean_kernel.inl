// qm_lean_kernel.inl -- qm_lean_kernel and its launcher, shared by the compile units that instantiate it (qm_kernels_lean.hip: the first
// pass; qm_kernels_leanq.hip: the N-aware pass over the queue of what the first pass left)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "qm_lean.inl"
#include "qm_device.h"

namespace qm {

template <bool PAIRED, bool SEL, bool PH, bool WIDE = false, bool NQ = false>
__global__ __launch_bounds__(256, 8) void qm_lean_kernel(DevIndex ix_, ReadBatch B_) {
  // the argument structs are read through the kernarg segment where they are used (see qm_read_kernel)
  struct Args { DevIndex ix; ReadBatch B; };
  typedef const Args __attribute__((address_space(4)))* AP4;
  const Args* args = (const Args*)(AP4)__builtin_amdgcn_kernarg_segment_ptr();
  const DevIndex& ix = args->ix; const ReadBatch& B = args->B;
  __shared__ __attribute__((aligned(16))) LeanMem mem[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int gw = (int)blockIdx.x * 4 + wave;
  const int nw = (int)gridDim.x * 4;
  const int nit = WIDE ? (int)B.nreads : (int)((B.nreads + 1) >> 1);   // iterations: two reads each, one in the wide edition (reads per launch < 2^31)
  LeanMem& M = mem[wave];
  {                                                          // the words behind the images stay zero
    const int l = (int)(threadIdx.x & 63);
    if (WIDE) { if (l < 16) (&M.pk[0][0][0])[16 * (l >> 3) + 8 + (l & 7)] = 0; }
    else if (l < 16) M.pk[l >> 3][(l >> 2) & 1][4 + (l & 3)] = 0;
    if (l < 5 * QM_LEAN_NMW) (&M.nm[0][0][0])[l] = 0;        // the N flags (and nmz behind them): lean_iter writes words 0-3 of a read with N's, the rest stays zero
  }
  WaveAlloc wa; wa.base = -1; wa.used = 0; wa.ivBase = -1; wa.ivUsed = 0;
  lean_stage_offsets<PAIRED, WIDE, NQ>(B, gw, nit, M, 0);
  lds_dma_wait();
  lean_stage_chars<PAIRED, WIDE, NQ>(B, gw, nit, M, 0);
  lean_stage_offsets<PAIRED, WIDE, NQ>(B, gw + nw, nit, M, 1);
  lds_dma_wait();
  int par = 0;
  for (int it = gw; it < nit; it += nw) {
    lean_iter<PAIRED, SEL, PH, WIDE, NQ>(ix, B, it, nit, nw, par, M, wa);
    par ^= 1;
  }
}

}  // namespace qm

using namespace qm;

// grid: QM_GRID_OVERSUB times the resident blocks (qmk_map_grid), but no more blocks than iterations / 4
template <bool PAIRED, bool SEL, bool PH, bool WIDE = false, bool NQ = false>
static hipError_t launch_lean(const DevIndex& ix, const ReadBatch& B, int num_cu, hipStream_t st) {
  static const int nb = [] {
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, qm_lean_kernel<PAIRED, SEL, PH, WIDE, NQ>, 256, 0) != hipSuccess || v < 1) v = 8;
    const char* ov = getenv("QM_BLOCKS_PER_CU");
    if (ov && atoi(ov) > 0 && atoi(ov) < v) v = atoi(ov);
    return v;
  }();
  const long long nit = WIDE ? B.nreads : (B.nreads + 1) >> 1;
  // (the plain kernel: twice the general kernels' oversubscription -- 455-466 -> 468-471 M pairs/s with the batch in two parts, whose launches are
  // short enough for their tails to show; its waves reserve list room in quarters of theirs, QM_LEAN_CHUNK)
  // (the N-aware pass over a queue -- a few hundred thousand reads -- : the resident grid, so that a wave's prologue is paid for by more than two or three iterations)
  long long g = (long long)num_cu * nb * (NQ ? 1 : (PH ? qmk_grid_oversub_ph() : (SEL ? qmk_grid_oversub() : 2 * qmk_grid_oversub())));
  const long long want = (nit + 3) / 4;
  if (g > want) g = want;
  if (g < 1) g = 1;
  hipLaunchKernelGGL((qm_lean_kernel<PAIRED, SEL, PH, WIDE, NQ>), dim3((unsigned)g), dim3(256), 0, st, ix, B);
  return hipGetLastError();
}
