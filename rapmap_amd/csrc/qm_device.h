// qm_device.h -- launch wrappers shared by qm_kernels.hip and qm_host.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define QMK_BLOCKS_PER_CU 8
// waves per SIMD the register allocator must leave room for, per flavour of the stage-A kernel (reads <= 128 bp)
#define QM_SEL_CHUNKS_B 4      // -s: chunks of units the plan -> ksw2 -> finish kernels go through (host side)
#define QMK_DEFAULT_WPS 8
#define QMK_WPS_PH 6
#define QMK_WPS_NIP 8
#define QMK_WPS_PHNIP 6

extern "C" {
hipError_t qmk_build_sainfo(const unsigned int* SA, long long nSA, const unsigned int* offsets, long long T, void* out, hipStream_t st);
hipError_t qmk_build_saext(const unsigned char* text, long long n, const unsigned int* SA, long long nSA, int k, const void* sainfo, void* out, hipStream_t st);
hipError_t qmk_build_saext2(const unsigned char* text, long long n, const unsigned int* SA, long long nSA, int k, const void* sainfo, void* out, hipStream_t st);
size_t qmk_saext2_bytes(void);
hipError_t qmk_build_sanext(const unsigned char* text, long long n, const unsigned int* SA, long long nSA, int k, unsigned int* out, hipStream_t st);
hipError_t qmk_build_slots(const void* recs, long long K, void* slots, unsigned long long cap, int k, hipStream_t st);
hipError_t qmk_build_slots_from_ph(const void* dev_index, long long n, void* slots, unsigned long long cap, unsigned long long* d_bad, hipStream_t st);
hipError_t qmk_build_phrecs(const unsigned int* data, const unsigned char* lens, long long n, const void* dev_index, void* out, hipStream_t st);
hipError_t qmk_build_phfilter(const void* recs, long long n, void* filter, unsigned long long mask, int k, hipStream_t st);
int qmk_map_grid(long long n, int num_cu);
int qmk_map_grid_ex(long long n, int num_cu, int ph_compact);   // ph_compact: the compact -p kernels' oversubscription
int qmk_grid_oversub_ph(void);
// blocks launched per resident block of a persistent stage-A grid (QM_GRID_OVERSUB, default 4); qmk_map_grid counts them in,
// qmk_resident_grid does not (kernels that own per-wave scratch in device memory: the list kernels)
int qmk_grid_oversub(void);
int qmk_resident_grid(long long n, int num_cu);
hipError_t qmk_launch_reads_ns2(const void* dev_index, const void* read_batch, int collect, int grid, int num_cu, hipStream_t st);
hipError_t qmk_launch_reads_ns3(const void* dev_index, const void* read_batch, int collect, int grid, int num_cu, hipStream_t st);
hipError_t qmk_launch_reads_ns4(const void* dev_index, const void* read_batch, int collect, int grid, int num_cu, hipStream_t st);
hipError_t qmk_launch_reads_ns8(const void* dev_index, const void* read_batch, int collect, int grid, int num_cu, hipStream_t st);
hipError_t qmk_launch_reads_ns32(const void* dev_index, const void* read_batch, int collect, int grid, int num_cu, hipStream_t st);
// the lean stage-A kernel (qm_kernels_lean.hip): reads of up to 128 clean characters, two per wavefront and iteration
hipError_t qmk_launch_lean(const void* dev_index, const void* read_batch, int num_cu, hipStream_t st);
// ... its N-aware edition over the queue read_batch.slowq[0 .. nreads) (qm_kernels_leanq.hip)
hipError_t qmk_launch_lean_nq(const void* dev_index, const void* read_batch, int num_cu, hipStream_t st);
// the pair kernel (qm_kernels_duo.hip): the two mates of a pair of up to 128 clean characters each in lockstep in one wavefront, merged there
hipError_t qmk_launch_duo(const void* dev_index, const void* read_batch, int num_cu, hipStream_t st);
hipError_t qmk_map_reads(const void* dev_index, const void* read_batch, int ns, int grid, int num_cu, hipStream_t st);
hipError_t qmk_map_reads_ex(const void* dev_index, const void* read_batch, int ns, int collect, int grid, int num_cu, hipStream_t st);
// ns < 0: the "collector only" stage entry (NS=4 kernels with QM_F_COLLECT)
hipError_t qmk_h2m(const void* dev_index, const void* read_batch, int grid, int num_cu, hipStream_t st);
// the -s list kernel, several reads per wavefront; reads it leaves for qmk_h2m are queued in todoq (count: scalar slot QM_SC_TODO)
// ... its wide edition over the queue `ids` (*nids entries) the narrow kernel left; what it hands on goes to todoq (count: QM_SC_TODO2)
hipError_t qmk_h2m_packw(const void* dev_index, const void* read_batch, const long long* ids, const unsigned long long* nids, long long* todoq, int grid, int num_cu, hipStream_t st);
// dst[i] = src[i] + add, i < n (a part's hit offsets into the whole batch's: map_device_split)
hipError_t qmk_rebase_offsets(const long long* src, long long* dst, long long n, long long add, hipStream_t st);
hipError_t qmk_h2m_pack(const void* dev_index, const void* read_batch, long long* todoq, int grid, int num_cu, hipStream_t st);
hipError_t qmk_sel_merge(const void* pair_batch, const void* sel_batch, hipStream_t st);
size_t qmk_sel_scratch_bytes(void);
size_t qmk_sel_dyn_struct_bytes(void);
unsigned long long qmk_sel_dyn_bytes(long long n);
void qmk_sel_dyn_bind(void* host_struct, void* dev_base, long long n);
hipError_t qmk_collect_slow(const unsigned int* lcnt, long long nreads, long long* q, unsigned long long* count, hipStream_t st);
hipError_t qmk_collect_lean(const unsigned int* lcnt, long long nreads, long long* q, unsigned long long* count, hipStream_t st);
hipError_t qmk_sel_slots(const void* pair_batch, hipStream_t st);
hipError_t qmk_sel_plan(const void* pair_batch, const void* sel_batch, int num_cu, hipStream_t st);
size_t qmk_sel_side_bytes(void);
hipError_t qmk_sel_align_finish(const void* pair_batch, const void* sel_batch, int num_cu, hipStream_t st);
size_t qmk_sel_task_bytes(void);
size_t qmk_sel_gmem_rows_bytes(int num_cu);   // device memory of qm_sel_align_gmem_kernel's alignment blocks (long reads under a band beyond 97)
hipError_t qmk_sel_compact(const void* pair_batch, const void* tmp, const void* toff, hipStream_t st);
hipError_t qmk_pair_count(const void* pair_batch, hipStream_t st);
hipError_t qmk_pair_write(const void* pair_batch, hipStream_t st);
// 2-bit packed reads -> the ASCII image the kernels read (include/qmap_mi355.h, "2-bit packed reads"): groups of four, then exceptions
hipError_t qmk_unpack_reads(const unsigned char* packed, const long long* off, long long n, long long total_chars, unsigned char* seq,
                            const void* exc, long long n_exc, hipStream_t st);
size_t qmk_scan_temp_bytes(long long n);
hipError_t qmk_scan_counts(void* temp, size_t temp_bytes, const unsigned int* cnt, long long* offs, long long n,
                           hipStream_t st);
// the same over (cnt & 0x7fffffff): list lengths carry the foundHit flag in bit 31
hipError_t qmk_scan_counts_masked(void* temp, size_t temp_bytes, const unsigned int* cnt, long long* offs, long long n, hipStream_t st);
// per read: its interval records and list words from their bump-allocated homes to CSR order (qm_fetch_stages)
hipError_t qmk_stage_gather(long long nreads, const unsigned int* ivcnt, const long long* ivoff, const void* iv, const long long* ivcsr, void* iv_out,
                            const unsigned int* lcnt, const long long* loff, const unsigned long long* lists, const long long* lcsr,
                            unsigned long long* words_out, hipStream_t st);
}
