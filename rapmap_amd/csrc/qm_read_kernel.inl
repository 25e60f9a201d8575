// qm_read_kernel.inl -- the stage-A kernel template and its per-slot-count launch wrappers.  The 36 instantiations
// (read-length class x index flavour x --noSensitive x -s, plus the collector-only stage entry) are spread over four
// translation units -- qm_kernels_ns{2,3,4,8}.hip, each defining qmk_launch_reads_ns<N> -- so that they compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "qm_mapper.inl"
#include "qm_device.h"

namespace qm {

template <bool ON> struct SelLds { SelScratchLds s; QM_DEV SelScratchLds* ptr() { return &s; } };
template <> struct SelLds<false> { QM_DEV SelScratchLds* ptr() { return nullptr; } };

// stage A: one wavefront per read.  WPS = minimum waves per SIMD the register allocator must leave room for.
// F: compile-time feature flags (QM_F_PH, QM_F_NIP) -- the default kernel carries no optional code.
// Waves per workgroup: four, except in the long-read kernels, whose per-wave LDS slab (41 KB at NS = 32) allows two.
template <int NS> struct WavesPerBlock { static constexpr int value = NS > 16 ? 2 : 4; };
template <int NS, int WPS, int F>
__global__ __launch_bounds__(64 * WavesPerBlock<NS>::value, WPS) void qm_read_kernel(DevIndex ix_, ReadBatch B_) {
  constexpr int WB = WavesPerBlock<NS>::value;
  // The two argument structs (~70 dwords) are read through the kernarg segment where they are used -- scalar loads from
  // constant memory -- instead of being loaded into SGPRs at entry and kept there: at 8 waves/SIMD a wave has 78 SGPRs, and
  // half of the v_readlane / v_writelane spill traffic of this kernel was for these words.
  struct Args { DevIndex ix; ReadBatch B; };
  typedef const Args __attribute__((address_space(4)))* AP4;
  const Args* args = (const Args*)(AP4)__builtin_amdgcn_kernarg_segment_ptr();
  const DevIndex& ix = args->ix; const ReadBatch& B = args->B;
  // The collector-only kernels never reach finish_read, the one user of WaveMem::buf (its first member, 1.5 KB): their waves' slabs are laid
  // over each other by that much -- wave w's buf is the tail of wave w-1's slab, wave 0's a pad in front -- and four waves take 4.6 KB less:
  // 8 blocks per CU instead of 6 at three slots (-s on 2 x 150 bp), 6 instead of 5 at four (2 x 250 bp).
  constexpr unsigned SKIP = (F & QM_F_COLLECT) ? (unsigned)sizeof(((WaveMem<NS>*)nullptr)->buf) : 0u;
  constexpr unsigned STRIDE = (unsigned)sizeof(WaveMem<NS>) - SKIP;
  static_assert(STRIDE % alignof(WaveMem<NS>) == 0 && SKIP % alignof(WaveMem<NS>) == 0, "slab overlap keeps the alignment");
  __shared__ __attribute__((aligned(16))) unsigned char memraw[SKIP + WB * STRIDE];
  __shared__ SelLds<(F & QM_F_SEL) != 0 && (F & QM_F_COLLECT) == 0> sels[WB];   // -s kernels that chain: the LDS edition of the scratch
  // the wave index is wave-uniform: keep it (and every address derived from it) on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int gw = (int)blockIdx.x * WB + wave;            // reads per launch < 2^31: 32-bit slot arithmetic
  const int nw = (int)gridDim.x * WB;
  const int nreads = (int)B.nreads;
  // the wave's 112 KB of device-memory scratch: its own by launch index, or -- a grid several times the resident one -- a slot taken
  // from the flags of the XCD the wave runs on (linear probing from a hashed start inside that eighth of the flags: at least twice as
  // many slots as an XCD's resident waves, so one is always free).  A slot only ever passes between waves of one XCD -- one L2 --, so
  // handing it on needs no cache maintenance: the holder waits for its own stores to be acknowledged, then clears the flag.  (An
  // agent-scope release per wave writes the XCD's L2 back: measured, it made a launch over 20 000 reads take a millisecond.)
  int gslot = gw;
  if (B.gslots) {
    int s = 0;
    if ((threadIdx.x & 63) == 0) {
      const unsigned nx = B.gxcd > 0 ? (unsigned)B.gxcd : 1u;                            // (the host derives it from the device it launches on: gscr_for)
      const unsigned per = (unsigned)B.ngslots / nx;
      const unsigned base = ((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) % nx) * per;   // HW_REG_XCC_ID, bits 3:0
      unsigned h = ((unsigned)gw * 2654435761u) % per;
      unsigned tries = 0;                                    // (every spin is bounded: flags that were never cleared must not hang the device)
      while (atomicCAS(&B.gslots[base + h], 0u, 1u) != 0u && ++tries < 4u * per) h = h + 1 == per ? 0u : h + 1;
      if (tries >= 4u * per) { atomicOr(B.status, 64); s = -1; }   // the host fails the call
      else s = (int)(base + h);
    }
    gslot = __builtin_amdgcn_readfirstlane(s);
    if (gslot < 0) return;                                   // no slot: this wave maps nothing rather than write into scratch another wave holds
  }
  u64* gscr = B.gscratch + (long long)gslot * QM_GSCR_U64;
  WaveAlloc wa; wa.base = -1; wa.used = 0; wa.ivBase = -1; wa.ivUsed = 0;
#ifdef QM_TIMING
  if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 9; ++i) qm_tim[wave][i] = 0; qm_tim[wave][9] = __builtin_readcyclecounter(); }
#endif
  // reads gw, gw + nw, ...: characters of the next read and offsets of the one after are staged in LDS while a read is mapped
  WaveMem<NS>& M = *reinterpret_cast<WaveMem<NS>*>(memraw + (unsigned)wave * STRIDE);
  stage_offsets<NS, F>(B, gw, M, 0);
  lds_dma_wait();
  stage_chars<NS, F>(B, gw, M, 0);
  stage_offsets<NS, F>(B, gw + nw, M, 1);
  lds_dma_wait();
  int par = 0;
  for (int r = gw; r < nreads; r += nw) {
    map_read<NS, F>(ix, B, read_id<F, NS>(B, r), r, nw, par, M, gscr, wa, (F & QM_F_SEL) ? B.selscr + gw : nullptr, sels[wave].ptr(),
                    ((F & QM_F_SEL) && B.dyn) ? B.dyn + gw : nullptr);
    par ^= 1;
  }
  if (B.gslots) {
    __builtin_amdgcn_s_waitcnt(0);                           // (this wave's stores to the slot have reached the L2 before the next holder's can)
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(&B.gslots[gslot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef QM_TIMING
  if ((threadIdx.x & 63) == 0) for (int i = 0; i < 7; ++i) atomicAdd((unsigned long long*)&B.cursor[20 + i], (unsigned long long)qm_tim[wave][i]);
#endif
}


// launch of one slot-count class; collect: the collector-only stage entry (QM_F_COLLECT)
template <int NS, int W0, int WPH, int WNIP, int WPHNIP, int WSEL, bool WITH_COLLECT, int WCOLL = W0>     // WCOLL: the -s collector of the default index
static hipError_t launch_reads_ns(const DevIndex& ix, const ReadBatch& B, bool collect, int grid, int num_cu, hipStream_t st) {
  const int F = (ix.ph ? QM_F_PH : 0) | (B.sensitive ? 0 : QM_F_NIP) | (B.selscr ? QM_F_SEL : 0);
  // A persistent grid (every wave strides over the reads) of QM_GRID_OVERSUB times the blocks that are resident at once: see
  // qmk_map_grid in qm_kernels.hip.  The occupancy of the chosen instantiation (VGPR/LDS dependent) decides the resident count.
#define QM_LAUNCH(WPS_, F_) do {                                                                               \
    static const int nb = [] { int v = 0;                                                                       \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, qm_read_kernel<NS, WPS_, F_>, 64 * WavesPerBlock<NS>::value, 0) != hipSuccess || v < 1) v = WPS_; \
      return v; }();                                                                                            \
    static const char* ov = getenv("QM_BLOCKS_PER_CU");   /* tuning knob: fewer resident blocks than the occupancy allows */ \
    long long g = (long long)num_cu * ((ov && atoi(ov) > 0 && atoi(ov) < nb) ? atoi(ov) : nb) * (((F_) & QM_F_PH) ? qmk_grid_oversub_ph() : qmk_grid_oversub());  \
    if (g > grid) g = grid;                                                                                     \
    hipLaunchKernelGGL((qm_read_kernel<NS, WPS_, F_>), dim3((unsigned)g), dim3(64 * WavesPerBlock<NS>::value), 0, st, ix, B); \
  } while (0)
  if constexpr (WITH_COLLECT) if (collect) {
    // collector-only kernels: the chain-scoring flavours for every slot count (first pass of a fused -s call: without the
    // chaining code they fit the register budget of the default kernel, so they are built for its occupancy), the others
    // only at eight slots (the stage entry qm_collect_reads, any read length)
    if constexpr (NS <= 8 || NS == 32) {                   // (32: the long-read pass of a -s call)
      switch (F) {
        case QM_F_SEL: QM_LAUNCH(WCOLL, QM_F_SEL | QM_F_COLLECT); return hipGetLastError();
        case QM_F_SEL | QM_F_PH: QM_LAUNCH(WPH, QM_F_SEL | QM_F_PH | QM_F_COLLECT); return hipGetLastError();
        case QM_F_SEL | QM_F_NIP: QM_LAUNCH(WNIP, QM_F_SEL | QM_F_NIP | QM_F_COLLECT); return hipGetLastError();
        case QM_F_SEL | QM_F_PH | QM_F_NIP: QM_LAUNCH(WPHNIP, QM_F_SEL | QM_F_PH | QM_F_NIP | QM_F_COLLECT); return hipGetLastError();
        default: break;
      }
    }
    if (F & QM_F_SEL) return hipErrorInvalidValue;
    switch (F) {
      default:
        if constexpr (NS == 8 || NS == 32) {
          switch (F) {
            case 0: QM_LAUNCH(W0, QM_F_COLLECT); break;
            case QM_F_PH: QM_LAUNCH(WPH, QM_F_PH | QM_F_COLLECT); break;
            case QM_F_NIP: QM_LAUNCH(WNIP, QM_F_NIP | QM_F_COLLECT); break;
            default: QM_LAUNCH(WPHNIP, QM_F_PH | QM_F_NIP | QM_F_COLLECT); break;
          }
        } else return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (F) {
    case 0: QM_LAUNCH(W0, 0); break;
    case QM_F_PH: QM_LAUNCH(WPH, QM_F_PH); break;
    case QM_F_NIP: QM_LAUNCH(WNIP, QM_F_NIP); break;
    case QM_F_PH | QM_F_NIP: QM_LAUNCH(WPHNIP, QM_F_PH | QM_F_NIP); break;
    // -s never runs as one fused kernel: the host launches the chain-scoring collector (above) and qm_h2m_kernel<QM_F_SEL>
    default: return hipErrorInvalidValue;
  }
#undef QM_LAUNCH
  return hipGetLastError();
}

}  // namespace qm
