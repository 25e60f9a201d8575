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
  void* (*alloc)(size_t); void (*release)(void*);          // where the buffers live: malloc / pinned host memory
} qm_batch_bufs;
int qm_reader_next_into(qm_reader* r, int64_t max_units, int64_t* n_units, qm_batch_bufs* B);
void qm_reader_estimate(qm_reader* r, int s, double* seq_bytes, double* name_bytes);
}
