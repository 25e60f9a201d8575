// qm_kernels_ns3.hip -- stage-A kernels for reads of up to 192 characters (3 64-character slots per read); see qm_read_kernel.inl
#include "qm_read_kernel.inl"
extern "C" hipError_t qmk_launch_reads_ns3(const void* ixp, const void* bp, int collect, int grid, int num_cu, hipStream_t st) {
  return qm::launch_reads_ns<3, 6, 5, 6, 5, 3, true, 8>(*(const qm::DevIndex*)ixp, *(const qm::ReadBatch*)bp, collect != 0, grid, num_cu, st);
}
