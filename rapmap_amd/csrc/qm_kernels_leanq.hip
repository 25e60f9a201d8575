// qm_kernels_leanq.hip -- qm_lean_kernel's N-aware edition (lean_iter<..., NQ>): the second pass of stage A over the queue of the reads the
// first pass left, when enough of them are there because of an N (qm_host.hip, run_stage_a)
#include "qm_lean_kernel.inl"

using namespace qm;

// The N-aware edition over a queue (B.slowq[0 .. B.nreads): reads of the batch the first pass left): reads whose characters outside A C G T
// are all N are mapped here, the others marked again (qm_host.hip, run_stage_a).  Not for the wide edition.
extern "C" hipError_t qmk_launch_lean_nq(const void* ixp, const void* bp, int num_cu, hipStream_t st) {
  const DevIndex& ix = *(const DevIndex*)ixp; const ReadBatch& B = *(const ReadBatch*)bp;
  if (B.lean_wide || !B.slowq) return hipErrorInvalidValue;
  const int v = (B.seq2 ? 4 : 0) | (B.selscr ? 2 : 0) | (ix.ph ? 1 : 0);
  switch (v) {
    case 0: return launch_lean<false, false, false, false, true>(ix, B, num_cu, st);
    case 1: return launch_lean<false, false, true, false, true>(ix, B, num_cu, st);
    case 2: return launch_lean<false, true, false, false, true>(ix, B, num_cu, st);
    case 3: return launch_lean<false, true, true, false, true>(ix, B, num_cu, st);
    case 4: return launch_lean<true, false, false, false, true>(ix, B, num_cu, st);
    case 5: return launch_lean<true, false, true, false, true>(ix, B, num_cu, st);
    case 6: return launch_lean<true, true, false, false, true>(ix, B, num_cu, st);
    default: return launch_lean<true, true, true, false, true>(ix, B, num_cu, st);
  }
}
