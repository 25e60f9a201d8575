// qm_stream.hip -- FASTA/FASTQ files -> mapped batches, pipelined (the ingest side of the path).
//
// The reference feeds its mapping threads from one kseq producer thread and hands every record over as std::strings
// (src/FastxParser.cpp:229-328); results leave through per-thread string buffers.  Here the three things a batch needs
// overlap instead of following each other:
//     reader thread    parses the next batch with the library's multi-threaded reader straight into PINNED host buffers
//     two map threads  each with a device context of its own (they share the index replica): upload (true async DMA out of
//                      pinned memory, chunk by chunk under the kernels), stage A/B kernels, download of the hits into the
//                      batch's pinned result buffers -- while one context downloads, the other one's kernels run
//     the caller       drains finished batches in input order; everything it is handed (reads, names, hit offsets, hits)
//                      sits in pinned memory and stays valid until it asks for the next batch: no copy into cold pages
// A batch travels through a fixed ring of slots, so memory is bounded and allocated once.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/qmap_mi355.h"
#include "qm_io_internal.h"

namespace {

void* pin_alloc(size_t bytes) { void* p = nullptr; return hipHostMalloc(&p, bytes ? bytes : 64, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
void pin_free(void* p) { if (p) hipHostFree(p); }

struct Slot {
  qm_batch_bufs in;
  int64_t n = 0, seqNo = -1;
  int64_t* hitOff = nullptr; size_t capHitOff = 0;
  qm_hit* hits = nullptr; size_t capHits = 0;
  int64_t nHits = 0; qm_counters ctr{}; double gpuMs = 0;
  int state = 0;                 // 0 free, 1 being read, 2 read, 3 being mapped, 4 mapped, 5 with the caller
  int rc = 0; char err[256] = "";
};

}  // namespace

struct qm_stream {
  const qm_index* ix = nullptr;
  int device = 0;
  qm_opts opts{};
  bool paired = false;
  int64_t batchUnits = 0;
  qm_reader* reader = nullptr;
  std::vector<qm_ctx*> ctx;
  std::vector<Slot> slots;
  std::mutex mu; std::condition_variable cv;
  int64_t nextRead = 0, nextOut = 0;      // sequence numbers
  bool eof = false, stop = false;
  int held = -1;                           // slot the caller holds
  std::thread readerThread; std::vector<std::thread> mapThreads;
  int failed = 0; char err[256] = "";
  // seconds spent by the stages (qm_stream_stats): reading, mapping (upload + kernels), downloading, the caller waiting, opening
  double tRead = 0, tMap = 0, tFetch = 0, tWait = 0, tOpen = 0, tAlloc = 0;
};
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static thread_local char g_serr[512] = "";
static int sfail(int code, const char* msg) { snprintf(g_serr, sizeof(g_serr), "%s", msg); return code; }

static void reader_loop(qm_stream* s) {
  hipSetDevice(s->device);
  while (true) {
    int si = -1;
    {
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv.wait(lk, [&] { if (s->stop || s->failed) return true; for (size_t i = 0; i < s->slots.size(); ++i) if (s->slots[i].state == 0) return true; return false; });
      if (s->stop || s->failed) return;
      for (size_t i = 0; i < s->slots.size(); ++i) if (s->slots[i].state == 0) { si = (int)i; break; }
      s->slots[(size_t)si].state = 1;
    }
    Slot& S = s->slots[(size_t)si];
    int64_t n = 0;
    const double t0 = now_s();
    const int rc = qm_reader_next_into(s->reader, s->batchUnits, &n, &S.in);
    const double t1 = now_s();
    std::unique_lock<std::mutex> lk(s->mu);
    s->tRead += t1 - t0;
    if (rc) { s->failed = rc; snprintf(s->err, sizeof(s->err), "%s", qm_io_last_error()); S.state = 0; s->cv.notify_all(); return; }
    if (n == 0) { s->eof = true; S.state = 0; s->cv.notify_all(); return; }
    S.n = n; S.seqNo = s->nextRead++; S.state = 2;
    s->cv.notify_all();
  }
}

static void map_loop(qm_stream* s, int which) {
  hipSetDevice(s->device);
  qm_ctx* c = s->ctx[(size_t)which];
  while (true) {
    int si = -1;
    {
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv.wait(lk, [&] {
        if (s->stop || s->failed) return true;
        for (size_t i = 0; i < s->slots.size(); ++i) if (s->slots[i].state == 2) return true;
        return s->eof;
      });
      if (s->stop || s->failed) return;
      int64_t best = -1;
      for (size_t i = 0; i < s->slots.size(); ++i)
        if (s->slots[i].state == 2 && (best < 0 || s->slots[i].seqNo < best)) { best = s->slots[i].seqNo; si = (int)i; }
      if (si < 0) return;                                   // end of input and nothing left to map
      s->slots[(size_t)si].state = 3;
    }
    Slot& S = s->slots[(size_t)si];
    int rc;
    const double t0 = now_s(); double t1 = t0, t2 = t0, ta = 0;
    if (s->paired) rc = qm_map_pairs(c, &s->opts, S.n, S.in.seq[0], S.in.off[0], S.in.seq[1], S.in.off[1], &S.nHits, &S.ctr);
    else rc = qm_map_reads(c, &s->opts, S.n, S.in.seq[0], S.in.off[0], &S.nHits, &S.ctr);
    t1 = now_s();
    if (!rc) {
      if (S.capHitOff < (size_t)S.n + 1) { pin_free(S.hitOff); S.capHitOff = (size_t)S.n + 1 + (size_t)S.n / 4; S.hitOff = (int64_t*)pin_alloc(S.capHitOff * 8); }
      if (S.capHits < (size_t)S.nHits + 1) { pin_free(S.hits); S.capHits = (size_t)S.nHits + 1 + (size_t)S.nHits / 4; S.hits = (qm_hit*)pin_alloc(S.capHits * sizeof(qm_hit)); }
      ta = now_s() - t1;
      if (!S.hitOff || !S.hits) rc = QM_E_NOMEM;
      else rc = qm_fetch_hits_pinned(c, S.hitOff, S.hits);
      t2 = now_s();
      double a = 0, b = 0;
      if (!rc && qm_last_kernel_ms(c, &a, &b) == QM_OK) S.gpuMs = b;
    }
    std::unique_lock<std::mutex> lk(s->mu);
    if (rc) { s->failed = rc; snprintf(s->err, sizeof(s->err), "%s", rc == QM_E_NOMEM ? "out of pinned memory" : qm_last_error()); s->cv.notify_all(); return; }
    S.state = 4;
    s->tMap += t1 - t0; s->tFetch += t2 - t1 - ta; s->tAlloc += ta;
    s->cv.notify_all();
  }
}

extern "C" {

const char* qm_stream_last_error(void) { return g_serr; }

int qm_stream_open(const qm_index* ix, int device_id, uint32_t ctx_flags, const qm_opts* opts, const char* path1, const char* path2,
                   int64_t batch_units, int32_t reader_threads, qm_stream** out) {
  if (!ix || !opts || !path1 || !out || batch_units <= 0) return sfail(QM_E_ARG, "qm_stream_open: bad argument");
  qm_stream* s = new qm_stream();
  const double tOpen0 = now_s();
  s->ix = ix; s->device = device_id; s->opts = *opts; s->paired = path2 != nullptr; s->batchUnits = batch_units;
  int rc = qm_reader_open(path1, path2, reader_threads > 0 ? reader_threads : 8, &s->reader);
  if (rc) { sfail(rc, qm_io_last_error()); delete s; return rc; }
  s->slots.resize(4);
  for (Slot& S : s->slots) { memset(&S.in, 0, sizeof(S.in)); S.in.alloc = pin_alloc; S.in.release = pin_free; }
  // The batches' pinned buffers, sized from the head of the files and pinned while the contexts are being created: pinning
  // ~300 MB inside the first batches cost as much as reading 5 M pairs.  (A wrong guess only means the reader grows them.)
  std::vector<std::thread> pinners;
  {
    const int nsrc = path2 ? 2 : 1;
    for (int m = 0; m < nsrc; ++m) {
      double sb = 0, nb = 0;
      qm_reader_estimate(s->reader, m, &sb, &nb);
      if (sb <= 0) continue;
      const size_t capSeq = (size_t)((double)batch_units * (sb * 1.05 + 1.0)) + 4096, capNames = (size_t)((double)batch_units * (nb * 1.05 + 1.0)) + 4096;
      const size_t capOff = (size_t)batch_units + 1 + 4096;
      for (Slot& S : s->slots)
        pinners.emplace_back([&S, m, capSeq, capNames, capOff, device_id]() {
          hipSetDevice(device_id);
          if ((S.in.seq[m] = (char*)pin_alloc(capSeq))) S.in.cap_seq[m] = capSeq;
          if ((S.in.names[m] = (char*)pin_alloc(capNames))) S.in.cap_names[m] = capNames;
          if ((S.in.off[m] = (int64_t*)pin_alloc(capOff * 8))) S.in.cap_off[m] = capOff;
          if ((S.in.noff[m] = (int64_t*)pin_alloc(capOff * 8))) S.in.cap_noff[m] = capOff;
        });
    }
  }
  auto joinPinners = [&]() { for (auto& t : pinners) t.join(); pinners.clear(); };
  for (int i = 0; i < 2; ++i) {
    qm_ctx* c = nullptr;
    rc = qm_ctx_create_ex(ix, device_id, ctx_flags, &c);
    if (rc) {
      sfail(rc, qm_last_error()); joinPinners();
      for (Slot& S : s->slots) for (int m = 0; m < 2; ++m) { pin_free(S.in.seq[m]); pin_free(S.in.off[m]); pin_free(S.in.names[m]); pin_free(S.in.noff[m]); }
      for (qm_ctx* x : s->ctx) qm_ctx_destroy(x);
      qm_reader_close(s->reader); delete s; return rc;
    }
    s->ctx.push_back(c);
  }
  joinPinners();
  s->readerThread = std::thread(reader_loop, s);
  for (int i = 0; i < 2; ++i) s->mapThreads.emplace_back(map_loop, s, i);
  s->tOpen = now_s() - tOpen0;
  *out = s;
  return QM_OK;
}

int qm_stream_next(qm_stream* s, qm_stream_batch* b) {
  if (!s || !b) return sfail(QM_E_ARG, "qm_stream_next: bad argument");
  std::unique_lock<std::mutex> lk(s->mu);
  if (s->held >= 0) { s->slots[(size_t)s->held].state = 0; s->held = -1; s->cv.notify_all(); }   // the caller is done with the previous batch
  int si = -1;
  const double tw0 = now_s();
  s->cv.wait(lk, [&] {
    if (s->failed) return true;
    for (size_t i = 0; i < s->slots.size(); ++i) if (s->slots[i].state == 4 && s->slots[i].seqNo == s->nextOut) { si = (int)i; return true; }
    if (!s->eof) return false;
    for (size_t i = 0; i < s->slots.size(); ++i) if (s->slots[i].state != 0 && s->slots[i].state != 5) return false;   // still in flight
    return true;                                            // drained
  });
  s->tWait += now_s() - tw0;
  if (s->failed) return sfail(s->failed, s->err);
  memset(b, 0, sizeof(*b));
  if (si < 0) return QM_OK;                                 // n_units == 0: end of input
  Slot& S = s->slots[(size_t)si];
  S.state = 5; s->held = si; s->nextOut++;
  b->n_units = S.n;
  b->seq1 = S.in.seq[0]; b->off1 = S.in.off[0]; b->names1 = S.in.names[0]; b->name_off1 = S.in.noff[0];
  if (s->paired) { b->seq2 = S.in.seq[1]; b->off2 = S.in.off[1]; b->names2 = S.in.names[1]; b->name_off2 = S.in.noff[1]; }
  b->hit_offsets = S.hitOff; b->hits = S.hits; b->n_hits = S.nHits; b->counters = S.ctr; b->gpu_ms = S.gpuMs;
  return QM_OK;
}

/* seconds: [0] reader (parse + pack), [1] upload + kernels (both contexts), [2] download, [3] the caller waiting in
 * qm_stream_next, [4] qm_stream_open, [5] growing the pinned result buffers */
int qm_stream_stats(qm_stream* s, double* out6) {
  if (!s || !out6) return sfail(QM_E_ARG, "qm_stream_stats: bad argument");
  std::unique_lock<std::mutex> lk(s->mu);
  out6[0] = s->tRead; out6[1] = s->tMap; out6[2] = s->tFetch; out6[3] = s->tWait; out6[4] = s->tOpen; out6[5] = s->tAlloc;
  return QM_OK;
}

void qm_stream_close(qm_stream* s) {
  if (!s) return;
  { std::unique_lock<std::mutex> lk(s->mu); s->stop = true; s->cv.notify_all(); }
  if (s->readerThread.joinable()) s->readerThread.join();
  for (auto& t : s->mapThreads) if (t.joinable()) t.join();
  hipSetDevice(s->device);
  for (Slot& S : s->slots) {
    for (int m = 0; m < 2; ++m) { pin_free(S.in.seq[m]); pin_free(S.in.off[m]); pin_free(S.in.names[m]); pin_free(S.in.noff[m]); }
    pin_free(S.hitOff); pin_free(S.hits);
  }
  for (qm_ctx* c : s->ctx) qm_ctx_destroy(c);
  qm_reader_close(s->reader);
  delete s;
}

}  // extern "C"
