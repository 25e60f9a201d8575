// qm_kernels_duo.hip -- the pair kernel (qm_duo.inl): the two mates of a read pair walked in lockstep by the two halves of one wavefront
// and merged there; what it leaves is mapped by qm_read_kernel and merged by stage B (qm_host.hip, run_stage_a / run_stage_b)
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "qm_duo.inl"
#include "qm_device.h"

namespace qm {

template <bool PH, bool COV>
__global__ __launch_bounds__(256, 8) void qm_duo_kernel(DevIndex ix_, ReadBatch B_) {
  // the argument structs are read through the kernarg segment where they are used (see qm_read_kernel)
  struct Args { DevIndex ix; ReadBatch B; };
  typedef const Args __attribute__((address_space(4)))* AP4;
  const Args* args = (const Args*)(AP4)__builtin_amdgcn_kernarg_segment_ptr();
  const DevIndex& ix = args->ix; const ReadBatch& B = args->B;
  __shared__ __attribute__((aligned(16))) DuoMem mem[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int gw = (int)blockIdx.x * 4 + wave;
  const int nw = (int)gridDim.x * 4;
  const int nit = (int)(B.nreads >> 1);                     // pairs (reads per launch < 2^31)
  DuoMem& M = mem[wave];
  {                                                          // the words behind the images stay zero
    const int l = (int)(threadIdx.x & 63);
    if (l < 16) M.pk[l >> 3][(l >> 2) & 1][4 + (l & 3)] = 0;
  }
  WaveAlloc wa; wa.base = -1; wa.used = 0; wa.ivBase = -1; wa.ivUsed = 0;
  DuoCtr ctr = {0, 0, 0, 0, 0, 0};
  duo_stage_offsets(B, gw, nit, M, 0);
  lds_dma_wait();
  duo_stage_chars(B, gw, nit, M, 0);
  duo_stage_offsets(B, gw + nw, nit, M, 1);
  lds_dma_wait();
  int par = 0;
  DuoNext N;
  {
    const int l = (int)(threadIdx.x & 63);
    N.Lv[l] = 0; N.defv[l] = 0; N.ck[l] = 0; N.flg[l] = 0;
  }
  duo_prepare<PH>(ix, B, gw, nit, nw, 0, M, N);             // (every later pair is prepared by the iteration before it)
  for (int it = gw; it < nit; it += nw) {
    duo_iter<PH, COV>(ix, B, it, nit, nw, par, M, wa, ctr, N);
    par ^= 1;
  }
  // the HitCounters of the pairs this wave merged (stage B's count pass adds the others')
  {
    const int l = (int)(threadIdx.x & 63);
    const u32 v = l == 0 ? ctr.pe : (l == 1 ? ctr.se : (l == 2 ? ctr.tot : (l == 3 ? ctr.reads : (l == 4 ? ctr.tooMany : ctr.mapped))));
    if (l < 6 && v) atomicAdd((unsigned long long*)(B.cursor + 1 + l), (unsigned long long)v);
  }
}

}  // namespace qm

using namespace qm;

template <bool PH, bool COV>
static hipError_t launch_duo(const DevIndex& ix, const ReadBatch& B, int num_cu, hipStream_t st) {
  static const int nb = [] {
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, qm_duo_kernel<PH, COV>, 256, 0) != hipSuccess || v < 1) v = 8;
    const char* ov = getenv("QM_BLOCKS_PER_CU");
    if (ov && atoi(ov) > 0 && atoi(ov) < v) v = atoi(ov);
    return v;
  }();
  static const int over = [] { const char* e = getenv("QM_DUO_OVERSUB"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
  const long long nit = B.nreads >> 1;
  long long g = (long long)num_cu * nb * (over ? over : (PH ? qmk_grid_oversub_ph() : 2 * qmk_grid_oversub()));
  const long long want = (nit + 3) / 4;
  if (g > want) g = want;
  if (g < 1) g = 1;
  hipLaunchKernelGGL((qm_duo_kernel<PH, COV>), dim3((unsigned)g), dim3(256), 0, st, ix, B);
  return hipGetLastError();
}
// paired reads of up to 128 characters, dense table or (ix.ph set) the compact -p image
extern "C" hipError_t qmk_launch_duo(const void* ixp, const void* bp, int num_cu, hipStream_t st) {
  const DevIndex& ix = *(const DevIndex*)ixp; const ReadBatch& B = *(const ReadBatch*)bp;
  if (!B.seq2 || (B.nreads & 1)) return hipErrorInvalidValue;
  if (B.quasi_cov > 0.0) return ix.ph ? launch_duo<true, true>(ix, B, num_cu, st) : launch_duo<false, true>(ix, B, num_cu, st);   // (--quasiCoverage: the walks also add up their coverage)
  return ix.ph ? launch_duo<true, false>(ix, B, num_cu, st) : launch_duo<false, false>(ix, B, num_cu, st);
}
