// qm_kernels_ns4.hip -- stage-A kernels for reads of up to 256 characters (4 64-character slots per read); see qm_read_kernel.inl
#include "qm_read_kernel.inl"
extern "C" hipError_t qmk_launch_reads_ns4(const void* ixp, const void* bp, int collect, int grid, int num_cu, hipStream_t st) {
  return qm::launch_reads_ns<4, 3, 3, 3, 3, 2, true>(*(const qm::DevIndex*)ixp, *(const qm::ReadBatch*)bp, collect != 0, grid, num_cu, st);
}
