// qm_kernels_ns32.hip -- stage-A kernels of the long-read pass: reads of 513 .. 2048 characters (32 64-character slots per read,
// two wavefronts per workgroup, the Wide<> flag word); launched over the queue of reads the main launch set aside.  See
// qm_read_kernel.inl and map_read (qm_mapper.inl).  Of -s its collector flavours (qm_read_kernel<32,.,SEL|COLLECT>): the long reads' intervals
// for the list kernel.
#include "qm_read_kernel.inl"
extern "C" hipError_t qmk_launch_reads_ns32(const void* ixp, const void* bp, int collect, int grid, int num_cu, hipStream_t st) {
  return qm::launch_reads_ns<32, 1, 1, 1, 1, 1, true>(*(const qm::DevIndex*)ixp, *(const qm::ReadBatch*)bp, collect != 0, grid, num_cu, st);
}
