// qm_kernels_lean.hip -- the lean stage-A kernel (qm_lean.inl): two reads per wavefront and iteration, for the reads that make up
// nearly all of a batch; what it leaves is mapped by qm_read_kernel (qm_host.hip, run_stage_a)
#include "qm_lean_kernel.inl"

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
