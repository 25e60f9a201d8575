// qm_kernels_ns2.hip -- stage-A kernels for reads of up to 128 characters (2 64-character slots per read); see qm_read_kernel.inl
#include "qm_read_kernel.inl"
extern "C" hipError_t qmk_launch_reads_ns2(const void* ixp, const void* bp, int collect, int grid, int num_cu, hipStream_t st) {
  return qm::launch_reads_ns<2, QMK_DEFAULT_WPS, QMK_WPS_PH, QMK_WPS_NIP, QMK_WPS_PHNIP, 4, true>(*(const qm::DevIndex*)ixp, *(const qm::ReadBatch*)bp, collect != 0, grid, num_cu, st);
}
