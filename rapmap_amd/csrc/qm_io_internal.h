// qm_io_internal.h -- between qm_io.cpp (reader) and qm_stream.hip (pipelined FASTQ -> hits): one batch's buffers and the
// reader call that fills them.  Not part of the public C ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/qmap_mi355.h"
extern "C" {
typedef struct qm_batch_bufs {
  char* seq[2]; int64_t* off[2]; char* names[2]; int64_t* noff[2];
  size_t cap_seq[2], cap_off[2], cap_names[2], cap_noff[2];
  // QM_INGEST_PACK: the same reads 2-bit packed (include/qmap_mi355.h, "2-bit packed reads"): read i from byte (off[i] >> 2) + i on,
  // exceptions appended by the copy tasks (n_exc counts every one of them: beyond cap_exc the batch has to travel as characters)
  uint8_t* pk[2]; qm_pack_exc* exc[2]; size_t cap_pk[2], cap_exc[2]; int64_t n_exc[2];
  void* (*alloc)(size_t); void (*release)(void*);          // where the buffers live: malloc / pinned host memory
} qm_batch_bufs;
/* The ingest engine (qm_ingest.cpp): files -> packed batches in `n_slots` slots whose buffers come from alloc/release,
 * filled by n_threads workers, several batches in flight.  qm_ingest_next blocks for the next batch in input order (any
 * number of consumer threads; n_units == 0: end of input) and lends its slot until qm_ingest_release. */
typedef struct qm_ingest qm_ingest;
enum { QM_INGEST_NO_NAMES = 1,       /* read names are not kept (names / noff stay NULL) */
       QM_INGEST_PACK = 2 };         /* the copy tasks also leave every batch 2-bit packed (pk / exc / n_exc) */
int qm_ingest_open(const char* path1, const char* path2, int32_t n_threads, int64_t batch_units, int32_t n_slots, uint32_t flags,
                   void* (*alloc)(size_t), void (*release)(void*), qm_ingest** out);
int qm_ingest_next(qm_ingest* g, int* slot, int64_t* n_units, int64_t* seq_no, const qm_batch_bufs** bufs);
void qm_ingest_release(qm_ingest* g, int slot);
void qm_ingest_stats(qm_ingest* g, double* out8);
void qm_ingest_cancel(qm_ingest* g);
void qm_ingest_close(qm_ingest* g);
int qm_io_fail(int code, const char* fmt, ...);
}
