// qm_kernels_lean.hip -- the lean stage-A kernel (qm_lean.inl): two reads per wavefront and iteration, for the reads that make up
// nearly all of a batch; what it leaves is mapped by qm_read_kernel (qm_host.hip, run_stage_a)
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "qm_lean.inl"
#include "qm_device.h"

namespace qm {

template <bool PAIRED, bool SEL, bool PH, bool WIDE = false>
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
  }
  WaveAlloc wa; wa.base = -1; wa.used = 0; wa.ivBase = -1; wa.ivUsed = 0;
  lean_stage_offsets<PAIRED, WIDE>(B, gw, nit, M, 0);
  lds_dma_wait();
  lean_stage_chars<PAIRED, WIDE>(B, gw, nit, M, 0);
  lean_stage_offsets<PAIRED, WIDE>(B, gw + nw, nit, M, 1);
  lds_dma_wait();
  int par = 0;
  for (int it = gw; it < nit; it += nw) {
    lean_iter<PAIRED, SEL, PH, WIDE>(ix, B, it, nit, nw, par, M, wa);
    par ^= 1;
  }
}

}  // namespace qm

using namespace qm;

// grid: QM_GRID_OVERSUB times the resident blocks (qmk_map_grid), but no more blocks than iterations / 4
template <bool PAIRED, bool SEL, bool PH, bool WIDE = false>
static hipError_t launch_lean(const DevIndex& ix, const ReadBatch& B, int num_cu, hipStream_t st) {
  static const int nb = [] {
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, qm_lean_kernel<PAIRED, SEL, PH, WIDE>, 256, 0) != hipSuccess || v < 1) v = 8;
    const char* ov = getenv("QM_BLOCKS_PER_CU");
    if (ov && atoi(ov) > 0 && atoi(ov) < v) v = atoi(ov);
    return v;
  }();
  const long long nit = WIDE ? B.nreads : (B.nreads + 1) >> 1;
  // (the plain kernel: twice the general kernels' oversubscription -- 455-466 -> 468-471 M pairs/s with the batch in two parts, whose launches are
  // short enough for their tails to show; its waves reserve list room in quarters of theirs, QM_LEAN_CHUNK)
  long long g = (long long)num_cu * nb * (PH ? qmk_grid_oversub_ph() : (SEL ? qmk_grid_oversub() : 2 * qmk_grid_oversub()));
  const long long want = (nit + 3) / 4;
  if (g > want) g = want;
  if (g < 1) g = 1;
  hipLaunchKernelGGL((qm_lean_kernel<PAIRED, SEL, PH, WIDE>), dim3((unsigned)g), dim3(256), 0, st, ix, B);
  return hipGetLastError();
}
// grid: QM_GRID_OVERSUB times the resident blocks (qmk_map_grid), but no more blocks than iterations / 4.
// B.selscr set: the chain-scoring collector of a -s call (intervals and foundHit out, no lists); ix.ph set: the compact -p image.
extern "C" hipError_t qmk_launch_lean(const void* ixp, const void* bp, int num_cu, hipStream_t st) {
  const DevIndex& ix = *(const DevIndex*)ixp; const ReadBatch& B = *(const ReadBatch*)bp;
  const int v = (B.seq2 ? 4 : 0) | (B.selscr ? 2 : 0) | (ix.ph ? 1 : 0);
  if (B.lean_wide) {                                       // reads of 129 .. 256 characters: one per wavefront
    if (!ix.saext2) return hipErrorInvalidValue;
    switch (v) {
      case 0: return launch_lean<false, false, false, true>(ix, B, num_cu, st);
      case 1: return launch_lean<false, false, true, true>(ix, B, num_cu, st);
      case 2: return launch_lean<false, true, false, true>(ix, B, num_cu, st);
      case 3: return launch_lean<false, true, true, true>(ix, B, num_cu, st);
      case 4: return launch_lean<true, false, false, true>(ix, B, num_cu, st);
      case 5: return launch_lean<true, false, true, true>(ix, B, num_cu, st);
      case 6: return launch_lean<true, true, false, true>(ix, B, num_cu, st);
      default: return launch_lean<true, true, true, true>(ix, B, num_cu, st);
    }
  }
  switch (v) {
    case 0: return launch_lean<false, false, false>(ix, B, num_cu, st);
    case 1: return launch_lean<false, false, true>(ix, B, num_cu, st);
    case 2: return launch_lean<false, true, false>(ix, B, num_cu, st);
    case 3: return launch_lean<false, true, true>(ix, B, num_cu, st);
    case 4: return launch_lean<true, false, false>(ix, B, num_cu, st);
    case 5: return launch_lean<true, false, true>(ix, B, num_cu, st);
    case 6: return launch_lean<true, true, false>(ix, B, num_cu, st);
    default: return launch_lean<true, true, true>(ix, B, num_cu, st);
  }
}
