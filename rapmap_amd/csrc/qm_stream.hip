// qm_stream.hip -- FASTA/FASTQ files -> mapped batches, pipelined, on one or several GPUs (SURVEY.md section 8f-3 and the
// product half of section 8e: "read batches are statically sharded across the GPUs of one node").
//
// The reference feeds its mapping threads from one kseq producer thread and hands every record over as std::strings
// (src/FastxParser.cpp:229-328, src/RapMapSAMapper.cpp:853,869-871); results leave through per-thread string buffers, in
// whatever order the threads finish.  Here:
//     ingest engine    (qm_ingest.cpp) worker threads parse the files chunk-parallel and pack batches STRAIGHT into PINNED
//                      host slots, several batches in flight
//     map threads      two per device, each with a device context of its own (the contexts of a device share the index
//                      replica): take the next batch in input order, upload (true async DMA out of pinned memory), stage
//                      A/B kernels, download of the hits into the slot's pinned result buffers -- while one context
//                      downloads, the other one's kernels run; with N devices the batches go round the 2N contexts
//     the caller       drains finished batches IN INPUT ORDER whichever device mapped them (deterministic output, unlike the
//                      reference's); everything it is handed (reads, names, hit offsets, hits) sits in pinned memory and
//                      stays valid until it asks for the next batch: no copy into cold pages
// A batch travels through a fixed ring of slots, so memory is bounded and allocated once.  Counters come per batch; their sum
// over the run is the caller's (one add per batch) -- across PROCESSES it is the one collective of the path (rapmap_amd/dist.py).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/qmap_mi355.h"
#include "qm_io_internal.h"

namespace {

// Pinned host memory for the slots.  Pinning is slow on this platform -- 5.5 GB/s whatever the number of threads (the driver
// serialises it; profiles/microbench/pin_cost.py): the 430 MB of a stream's six slots take 77 ms, half the time the whole
// stream needs for 10 M pairs, and the first batches wait for their slots.  So the library keeps a process-wide POOL of
// pinned arenas: qm_stream_reserve() pins it in a background thread (the CLI calls it before it creates its contexts, i.e.
// under the upload of the index), streams carve their buffers out of it (first fit over 1 MB granules, neighbours merged
// on release) and hand them back when they close -- the next stream of the process (the next file of a multi-file run)
// pins nothing.  Requests the pool cannot serve fall through to hipHostMalloc.  Portable: the slots are read by the DMA
// engines of every device of the stream.
struct PinPool {
  struct Arena { char* base; size_t bytes; std::vector<unsigned char> used; };   // one flag per 1 MB granule
  static constexpr size_t GR = (size_t)1 << 20;
  static constexpr size_t ARENA = (size_t)64 << 20;   // every arena is one hipHostMalloc of this size
  std::mutex mu; std::condition_variable cv;
  std::vector<Arena> arenas;
  size_t pending = 0;                               // bytes a background reservation still has to pin
  void* take(size_t bytes) {
    const size_t g = (bytes + GR - 1) / GR;
    if (g > ARENA / GR) return nullptr;             // no arena can ever hold it: do not wait for the reservation, pin it directly
    std::unique_lock<std::mutex> lk(mu);
    while (true) {
      for (Arena& A : arenas) {
        const size_t n = A.used.size();
        for (size_t i = 0; i + g <= n;) {
          size_t j = i; while (j < i + g && !A.used[j]) ++j;
          if (j == i + g) { for (size_t t = i; t < i + g; ++t) A.used[t] = (unsigned char)(t == i ? 2 : 1); return A.base + i * GR; }   // 2: first granule of a block
          i = j + 1;
        }
      }
      if (pending == 0) return nullptr;
      cv.wait(lk);                                  // a reservation is under way: what it pins next may serve this request
    }
  }
  bool give(void* p) {                              // true: p was a pool block
    std::lock_guard<std::mutex> lk(mu);
    for (Arena& A : arenas) {
      if ((char*)p < A.base || (char*)p >= A.base + A.bytes) continue;
      size_t i = (size_t)((char*)p - A.base) / GR;
      if (A.used[i] != 2) return true;              // (not the start of a block: ignore)
      A.used[i] = 0;
      for (size_t t = i + 1; t < A.used.size() && A.used[t] == 1; ++t) A.used[t] = 0;
      cv.notify_all();
      return true;
    }
    return false;
  }
  size_t total() { std::lock_guard<std::mutex> lk(mu); size_t t = pending; for (Arena& A : arenas) t += A.bytes; return t; }
};
PinPool& pin_pool() { static PinPool* P = new PinPool(); return *P; }      // (never destroyed: its memory lives as long as the process)

void* pin_alloc(size_t bytes) {
  if (void* q = pin_pool().take(bytes ? bytes : 64)) return q;
  void* p = nullptr;
  return hipHostMalloc(&p, bytes ? bytes : 64, hipHostMallocPortable) == hipSuccess ? p : nullptr;
}
void pin_free(void* p) { if (p && !pin_pool().give(p)) hipHostFree(p); }

struct OutSlot {                 // the result side of ingest slot i
  const qm_batch_bufs* in = nullptr;
  int64_t n = 0, seqNo = -1;
  int64_t* hitOff = nullptr; size_t capHitOff = 0;
  qm_hit* hits = nullptr; size_t capHits = 0;
  int64_t nHits = 0; qm_counters ctr{}; double gpuMs = 0; int device = 0;
  int state = 0;                 // 0 idle, 1 being mapped, 2 mapped, 3 with the caller
};

}  // namespace

struct qm_stream {
  const qm_index* ix = nullptr;
  qm_opts opts{};
  bool paired = false;
  bool packed = true;                      // batches go to the device 2-bit packed (QM_STREAM_NO_PACK=1: as characters)
  int64_t nPacked = 0;                     // batches that did
  qm_ingest* g = nullptr;
  std::vector<qm_ctx*> ctx; std::vector<int> ctxDev;
  std::vector<OutSlot> slots;
  std::mutex mu; std::condition_variable cv;
  int64_t nextOut = 0;
  int mapDone = 0;                         // map threads that have seen the end of the input
  bool stop = false;
  int held = -1;                           // slot the caller holds
  std::vector<std::thread> mapThreads;
  int failed = 0; char err[384] = "";
  // seconds spent by the stages (qm_stream_stats): mapping (upload + kernels), downloading, the caller waiting, opening
  double tMap = 0, tFetch = 0, tWait = 0, tOpen = 0, tAlloc = 0, t0 = 0, tLastMapped = 0;
};
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static thread_local char g_serr[512] = "";
static int sfail(int code, const char* msg) { snprintf(g_serr, sizeof(g_serr), "%s", msg); return code; }

static void map_loop(qm_stream* s, int which) {
  hipSetDevice(s->ctxDev[(size_t)which]);
  qm_ctx* c = s->ctx[(size_t)which];
  while (true) {
    { std::unique_lock<std::mutex> lk(s->mu); if (s->stop || s->failed) break; }
    int si = -1; int64_t n = 0, seq = 0; const qm_batch_bufs* in = nullptr;
    int rc = qm_ingest_next(s->g, &si, &n, &seq, &in);
    if (rc) {
      std::unique_lock<std::mutex> lk(s->mu);
      if (!s->failed) { s->failed = rc; snprintf(s->err, sizeof(s->err), "%s", qm_io_last_error()); }
      break;
    }
    if (si < 0 || n == 0) break;                                     // end of input (or the stream is closing)
    OutSlot& S = s->slots[(size_t)si];
    { std::unique_lock<std::mutex> lk(s->mu); S.in = in; S.n = n; S.seqNo = seq; S.state = 1; S.device = s->ctxDev[(size_t)which]; }
    const double t0 = now_s(); double t1 = t0, t2 = t0, ta = 0;
    // the batch travels 2-bit packed (26 bytes per 100-bp read instead of 100) unless its exceptions outgrew their list
    const bool packed = s->packed && in->pk[0] && in->n_exc[0] <= (int64_t)in->cap_exc[0] && (!s->paired || (in->pk[1] && in->n_exc[1] <= (int64_t)in->cap_exc[1]));
    if (packed) {
      if (s->paired) rc = qm_map_pairs_packed(c, &s->opts, n, in->pk[0], in->off[0], in->exc[0], in->n_exc[0], in->pk[1], in->off[1], in->exc[1], in->n_exc[1], &S.nHits, &S.ctr);
      else rc = qm_map_reads_packed(c, &s->opts, n, in->pk[0], in->off[0], in->exc[0], in->n_exc[0], &S.nHits, &S.ctr);
      __atomic_fetch_add(&s->nPacked, 1, __ATOMIC_RELAXED);
    }
    else if (s->paired) rc = qm_map_pairs(c, &s->opts, n, in->seq[0], in->off[0], in->seq[1], in->off[1], &S.nHits, &S.ctr);
    else rc = qm_map_reads(c, &s->opts, n, in->seq[0], in->off[0], &S.nHits, &S.ctr);
    t1 = now_s();
    if (!rc) {
      if (S.capHitOff < (size_t)n + 1) { pin_free(S.hitOff); S.capHitOff = (size_t)n + 1 + (size_t)n / 4; S.hitOff = (int64_t*)pin_alloc(S.capHitOff * 8); }
      if (S.capHits < (size_t)S.nHits + 1) { pin_free(S.hits); S.capHits = (size_t)S.nHits + 1 + (size_t)S.nHits / 4; S.hits = (qm_hit*)pin_alloc(S.capHits * sizeof(qm_hit)); }
      ta = now_s() - t1;
      if (!S.hitOff || !S.hits) rc = QM_E_NOMEM;
      else rc = qm_fetch_hits_pinned(c, S.hitOff, S.hits);
      t2 = now_s();
      double a = 0, b = 0;
      if (!rc && qm_last_kernel_ms(c, &a, &b) == QM_OK) S.gpuMs = b;
    }
    std::unique_lock<std::mutex> lk(s->mu);
    if (rc) { if (!s->failed) { s->failed = rc; snprintf(s->err, sizeof(s->err), "%s", rc == QM_E_NOMEM ? "out of pinned memory" : qm_last_error()); } break; }
    S.state = 2;
    s->tMap += t1 - t0; s->tFetch += t2 - t1 - ta; s->tAlloc += ta; s->tLastMapped = t2 - s->t0;
    s->cv.notify_all();
  }
  std::unique_lock<std::mutex> lk(s->mu);
  s->mapDone++;
  s->cv.notify_all();
}

extern "C" {

const char* qm_stream_last_error(void) { return g_serr; }

int qm_stream_open_ex(const qm_index* ix, const int32_t* devices, int32_t n_devices, uint32_t ctx_flags, const qm_opts* opts, const char* path1,
                      const char* path2, int64_t batch_units, int32_t reader_threads, uint32_t stream_flags, qm_stream** out) {
  if (!ix || !opts || !path1 || !out || batch_units <= 0 || !devices || n_devices <= 0) return sfail(QM_E_ARG, "qm_stream_open: bad argument");
  qm_stream* s = new qm_stream();
  s->t0 = now_s();
  s->ix = ix; s->opts = *opts; s->paired = path2 != nullptr;
  { const char* np = getenv("QM_STREAM_NO_PACK"); s->packed = !(np && atoi(np) != 0); }
  const char* cpd = getenv("QM_STREAM_CTX_PER_DEVICE");
  // contexts (map threads) per device: each uploads, maps and downloads its batch in turn, so several of them keep the link and
  // the kernels busy at once -- 2 / 3 / 4 / 6 contexts: 118 / 124-132 / 133-144 / 116-122 M pairs/s on 40 M pairs (round 4,
  // profiles/r04/stream_contexts_sweep.log)
  const int perDev = cpd && atoi(cpd) > 0 ? atoi(cpd) : 4;
  const int nctx = perDev * n_devices;
  // the files first (they may not exist), then the engine fills its slots while the contexts are being created
  int rc = qm_ingest_open(path1, path2, reader_threads > 0 ? reader_threads : 8, batch_units, 2 * nctx + 2,
                          ((stream_flags & QM_STREAM_NO_NAMES) ? QM_INGEST_NO_NAMES : 0u) | (s->packed ? QM_INGEST_PACK : 0u), pin_alloc, pin_free, &s->g);
  if (rc) { sfail(rc, qm_io_last_error()); delete s; return rc; }
  s->slots.resize((size_t)(2 * nctx + 2));
  // one thread per device creates that device's contexts one after the other: the first builds the index replica, the
  // others share it
  s->ctx.assign((size_t)nctx, nullptr); s->ctxDev.assign((size_t)nctx, 0);
  std::vector<int> rcs((size_t)n_devices, 0); std::vector<std::string> errs((size_t)n_devices);
  {
    std::vector<std::thread> th;
    for (int d = 0; d < n_devices; ++d)
      th.emplace_back([&, d]() {
        for (int i = 0; i < perDev; ++i) {
          qm_ctx* c = nullptr;
          const int r = qm_ctx_create_ex(ix, devices[d], ctx_flags, &c);
          if (r) { rcs[(size_t)d] = r; errs[(size_t)d] = qm_last_error(); return; }
          // context i of device d maps every (i * n_devices + d)-th batch: consecutive batches go to different devices
          s->ctx[(size_t)(i * n_devices + d)] = c; s->ctxDev[(size_t)(i * n_devices + d)] = devices[d];
        }
      });
    for (auto& t : th) t.join();
  }
  for (int d = 0; d < n_devices; ++d)
    if (rcs[(size_t)d]) {
      rc = rcs[(size_t)d]; sfail(rc, errs[(size_t)d].c_str());
      qm_ingest_cancel(s->g);
      for (qm_ctx* x : s->ctx) if (x) qm_ctx_destroy(x);
      qm_ingest_close(s->g); delete s; return rc;
    }
  for (int i = 0; i < nctx; ++i) s->mapThreads.emplace_back(map_loop, s, i);
  s->tOpen = now_s() - s->t0;
  *out = s;
  return QM_OK;
}

/* Pin `bytes` of host memory for the slots of the streams this process will open, in the background (returns at once; a stream
 * that opens meanwhile takes what is already there and waits for the rest).  The pool is never shrunk. */
int qm_stream_reserve(int64_t bytes) {
  if (bytes <= 0) return QM_OK;
  PinPool& P = pin_pool();
  const size_t ARENA = PinPool::ARENA;
  size_t want = ((size_t)bytes + ARENA - 1) / ARENA * ARENA;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    size_t have = P.pending; for (PinPool::Arena& A : P.arenas) have += A.bytes;
    if (have >= want) return QM_OK;
    want -= have; P.pending += want;
  }
  std::thread([want]() {
    const size_t ARENA = PinPool::ARENA;
    PinPool& P = pin_pool();
    for (size_t done = 0; done < want; done += ARENA) {
      void* p = nullptr;
      const bool ok = hipHostMalloc(&p, ARENA, hipHostMallocPortable) == hipSuccess;
      std::lock_guard<std::mutex> lk(P.mu);
      if (ok) { PinPool::Arena A; A.base = (char*)p; A.bytes = ARENA; A.used.assign(ARENA / PinPool::GR, 0); P.arenas.push_back(std::move(A)); }
      P.pending -= ARENA;
      P.cv.notify_all();
    }
  }).detach();
  return QM_OK;
}

int qm_stream_open(const qm_index* ix, int device_id, uint32_t ctx_flags, const qm_opts* opts, const char* path1, const char* path2,
                   int64_t batch_units, int32_t reader_threads, qm_stream** out) {
  const int32_t dev = device_id;
  return qm_stream_open_ex(ix, &dev, 1, ctx_flags, opts, path1, path2, batch_units, reader_threads, 0, out);
}

int qm_stream_next(qm_stream* s, qm_stream_batch* b) {
  if (!s || !b) return sfail(QM_E_ARG, "qm_stream_next: bad argument");
  std::unique_lock<std::mutex> lk(s->mu);
  if (s->held >= 0) {                                       // the caller is done with the previous batch: its slot may be refilled
    const int h = s->held;
    s->slots[(size_t)h].state = 0; s->held = -1;
    lk.unlock(); qm_ingest_release(s->g, h); lk.lock();
  }
  int si = -1;
  const double tw0 = now_s();
  s->cv.wait(lk, [&] {
    if (s->failed) return true;
    for (size_t i = 0; i < s->slots.size(); ++i) if (s->slots[i].state == 2 && s->slots[i].seqNo == s->nextOut) { si = (int)i; return true; }
    if (s->mapDone < (int)s->mapThreads.size()) return false;
    for (size_t i = 0; i < s->slots.size(); ++i) if (s->slots[i].state == 1 || s->slots[i].state == 2) return false;   // (cannot happen: kept for clarity)
    return true;                                            // drained
  });
  s->tWait += now_s() - tw0;
  if (s->failed) return sfail(s->failed, s->err);
  memset(b, 0, sizeof(*b));
  if (si < 0) return QM_OK;                                 // n_units == 0: end of input
  OutSlot& S = s->slots[(size_t)si];
  S.state = 3; s->held = si; s->nextOut++;
  b->n_units = S.n;
  b->seq1 = S.in->seq[0]; b->off1 = S.in->off[0]; b->names1 = S.in->names[0]; b->name_off1 = S.in->noff[0];
  if (s->paired) { b->seq2 = S.in->seq[1]; b->off2 = S.in->off[1]; b->names2 = S.in->names[1]; b->name_off2 = S.in->noff[1]; }
  b->hit_offsets = S.hitOff; b->hits = S.hits; b->n_hits = S.nHits; b->counters = S.ctr; b->gpu_ms = S.gpuMs; b->device = S.device;
  return QM_OK;
}

/* seconds spent so far: [0] the ingest engine, open to its last batch packed (wall), [1] upload + kernels (summed over the
 * contexts), [2] download (summed), [3] the caller waiting in qm_stream_next, [4] qm_stream_open, [5] growing the pinned result
 * buffers; qm_stream_stats_ex adds [6] open to the first batch packed, [7] parse tasks (summed over the workers), [8] copy tasks
 * (summed), [9] inflate threads, [10] bytes parsed, [11] open to the last batch mapped and downloaded (wall), [12] batches that
 * went to the device 2-bit packed */
int qm_stream_stats_ex(qm_stream* s, double* out, int32_t n) {
  if (!s || !out || n < 0) return sfail(QM_E_ARG, "qm_stream_stats: bad argument");
  double v[13] = {0}; double ing[8] = {0};
  qm_ingest_stats(s->g, ing);
  std::unique_lock<std::mutex> lk(s->mu);
  v[0] = ing[6]; v[1] = s->tMap; v[2] = s->tFetch; v[3] = s->tWait; v[4] = s->tOpen; v[5] = s->tAlloc;
  v[6] = ing[0]; v[7] = ing[1]; v[8] = ing[2]; v[9] = ing[3]; v[10] = ing[4]; v[11] = s->tLastMapped;
  v[12] = (double)__atomic_load_n(&s->nPacked, __ATOMIC_RELAXED);
  for (int i = 0; i < n && i < 13; ++i) out[i] = v[i];
  return QM_OK;
}
int qm_stream_stats(qm_stream* s, double* out6) { return qm_stream_stats_ex(s, out6, 6); }

void qm_stream_close(qm_stream* s) {
  if (!s) return;
  { std::unique_lock<std::mutex> lk(s->mu); s->stop = true; s->cv.notify_all(); }
  qm_ingest_cancel(s->g);                                   // wakes map threads that wait for a batch
  for (auto& t : s->mapThreads) if (t.joinable()) t.join();
  for (OutSlot& S : s->slots) { pin_free(S.hitOff); pin_free(S.hits); }
  for (qm_ctx* c : s->ctx) if (c) qm_ctx_destroy(c);
  qm_ingest_close(s->g);
  delete s;
}

}  // extern "C"
