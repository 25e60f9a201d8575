// qm_host.hip -- host side of libqmap_mi355.so: q5 index reader (mmap), HBM replica,
// and the C ABI declared in include/qmap_mi355.h.
//
// Reader follows the writers in src/RapMapSAIndexer.cpp:109-110,242-243 (sa.bin),
// :694-731 (rsd.bin, txpInfo.bin), :433-441 + include/sparsepp/spp.h:2355-2366 (hash.bin),
// include/IndexHeader.hpp:45-57 (header.json); semantics of
// RapMapSAIndex::load (src/RapMapSAIndex.cpp:97-176).  Files are mmap'd, never
// deserialised into containers; the GPU flattens them (qm_kernels.hip).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <functional>

#include "qm_mapper.inl"
#include "qm_lean.inl"
#include "qm_device.h"
#include "qm_phflat.h"

using namespace qm;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return code;
}
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(QM_E_NOGPU, "%s: %s", #x, hipGetErrorString(_e)); } while (0)

struct MMap {
  void* p = nullptr; size_t len = 0;
  int open(const std::string& path) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return -1;
    struct stat st; if (fstat(fd, &st) != 0) { ::close(fd); return -1; }
    len = (size_t)st.st_size;
    if (len == 0) { ::close(fd); return -1; }
    p = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) { p = nullptr; return -1; }
    return 0;
  }
  void close() { if (p) munmap(p, len); p = nullptr; }
};

struct qm_index {
  std::string dir;
  int k = 31; bool big = false, perfect = false;
  MMap sa, txp, rsd, hash;
  // SA entries, transcript offsets and interval bounds are UNSIGNED 32-bit from here on (qm_mapper.inl, struct Iv).  A BigSA index
  // (int64 on disk: text > 2^31 - 1 characters, src/RapMapSAIndexer.cpp:682-683) is narrowed into the *Narrow vectors at open.
  const uint32_t* SA = nullptr; int64_t nSA = 0;
  std::vector<std::string> names;
  const uint32_t* offsets = nullptr; int64_t nTxp = 0;
  std::vector<uint32_t> saNarrow, offNarrow, phDataNarrow; std::vector<uint8_t> hashNarrow;
  const uint8_t* text = nullptr; int64_t n = 0;
  const uint32_t* completeLens = nullptr;
  std::vector<int64_t> lens;
  const uint8_t* hashRecs = nullptr; int64_t nKeys = 0;   // K x 16 B records
  uint64_t rsdBits = 0;
  // perfect-hash flavour (hash_info.bph / hash_info.val)
  MMap bph, val;
  struct PhLevel { uint64_t domain, nchar, nranks; const uint8_t* words; const uint8_t* ranks; };
  std::vector<PhLevel> phLevels;
  uint64_t phLastRank = 0, phNelem = 0;
  std::vector<std::pair<uint64_t, uint64_t>> phFinal;
  const uint8_t* phData = nullptr; const uint8_t* phLens = nullptr;
  std::vector<std::pair<uint32_t, uint32_t>> phOverflow;
};

// The index replica of one device: built by the first context on (index, device), shared read-only by every later one
// (one context per host thread is the intended use: a Salmon-style caller has many), freed with the last of them.
struct Replica {
  int device = 0;
  uint8_t* d_text = nullptr; uint32_t* d_SA = nullptr; void* d_sainfo = nullptr; void* d_slots = nullptr; uint64_t cap = 0;
  void* d_ph = nullptr; PhIndex hPh; std::vector<void*> phAllocs; uint32_t* d_txpOff = nullptr; int32_t* d_txpLen = nullptr; int64_t devBytes = 0;
  std::mutex sanextMu; unsigned int* d_sanext = nullptr;   // -s: built by the first -s call of any context of this replica
  void* d_saext = nullptr;                                  // the packed characters behind every suffix's k-mer (SaExt), or null
  void* d_saext2 = nullptr; bool saext2Tried = false;       // ... the wide edition (SaExt2), built when reads of 129 .. 256 characters first ask (sanextMu)
  ~Replica() {
    hipSetDevice(device);
    void* ptrs[] = {d_text, d_SA, d_sainfo, d_slots, d_txpOff, d_txpLen, d_sanext, d_saext, d_saext2};
    for (void* p : ptrs) if (p) hipFree(p);
    for (void* p : phAllocs) if (p) hipFree(p);
  }
};
static std::mutex g_repMu;
static std::map<std::pair<const qm_index*, int>, std::weak_ptr<Replica>> g_reps;   // key: index, device * 2 + (compact perfect hash)
static std::map<std::pair<const qm_index*, int>, std::shared_ptr<std::mutex>> g_repBuild;   // one builder per key at a time

struct qm_ctx {
  const qm_index* ix = nullptr;
  std::shared_ptr<Replica> rep;
  int device = 0, numCU = 256;
  hipStream_t stream = nullptr, copyStream = nullptr;      // kernels / host-buffer uploads (overlapped chunk by chunk)
  unsigned int* d_sanext = nullptr;                         // -s: text characters behind every suffix's k-mer (the replica's, built at its first -s call)
  void* d_saext = nullptr;                                  // the replica's SaExt table
  void* d_saext2 = nullptr;                                 // ... and its wide edition, once built
  hipStream_t planStream = nullptr;                         // -s: the plan kernels of the later chunks, under the ksw2 kernel of the earlier ones
  hipEvent_t evPlan[QM_SEL_CHUNKS_B + 1] = {};              // ... [i]: chunk i planned; [last]: the plan stream may start
  u64* d_ntk = nullptr;                                     // ... per chunk: a task counter, then (at QM_SEL_CHUNKS_B + i) a counter of alignment questions
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evA = nullptr, evB = nullptr, evCopy = nullptr, evStage[2] = {nullptr, nullptr};
  hipEvent_t evP0 = nullptr, evP1 = nullptr;                // around the second launches of stage A (the N-aware pass, the general kernel over what is left): evA / evB time the whole call
  unsigned char* h_stage = nullptr;                        // pinned, 2 x 32 MB: result download (qm_fetch_hits)
  // index replica
  uint8_t* d_text = nullptr; uint32_t* d_SA = nullptr; void* d_sainfo = nullptr; void* d_slots = nullptr;
  uint64_t cap = 0;
  void* d_ph = nullptr; std::vector<void*> phAllocs;       // perfect-hash flavour: PhIndex struct + its arrays
  PhIndex hPh;                                              // host copy (passed to the kernels by value inside DevIndex)
  int64_t devBytes = 0;
  // work buffers
  int64_t capCnt = 0, capOffs = 0, capLcnt = 0, capLoff = 0, capLists = 0, capHits = 0, capSeq1 = 0, capSeq2 = 0, capOff1 = 0, capOff2 = 0, capGrid = 0;
  uint32_t* d_cnt = nullptr; long long* d_offs = nullptr;      // per unit: hits, exclusive scan
  uint32_t* d_lcnt = nullptr; long long* d_loff = nullptr;     // per read: list length / offset
  u64* d_lists = nullptr;                                      // bump-allocated per-read hit lists
  qm_hit* d_hits = nullptr;
  u64* d_scal = nullptr; /* cursor, counters[6], status */ u64* d_gscr = nullptr; unsigned* d_gslots = nullptr;   // (scratch slots' flags: gscr_for)
  u64* d_skip = nullptr; std::vector<uint64_t> skipList; int64_t lastSkipped = 0;   // reads the last call skipped (ReadBatch::skiplist)
  void* d_scanTmp = nullptr; size_t scanTmpBytes = 0;
  uint8_t* d_seq1 = nullptr; uint8_t* d_seq2 = nullptr; long long* d_off1 = nullptr; long long* d_off2 = nullptr;
  // SA-interval hits as an output (qm_fetch_intervals), foundHit flags, tooMany flags of a merge-only call
  qm_sa_interval_hit* d_iv = nullptr; uint32_t* d_ivcnt = nullptr; long long* d_ivoff = nullptr; int64_t capIv = 0, capIvCnt = 0, capIvOff = 0; int debug = 0;
  unsigned char* d_found = nullptr; int64_t capFound = 0; unsigned char* d_tooMany = nullptr; int64_t capTooMany = 0;
  // inputs of the stage entries
  qm_sa_interval_hit* d_ivIn = nullptr; int64_t capIvIn = 0; long long* d_ivInOff = nullptr; int64_t capIvInOff = 0;
  int* d_lenIn = nullptr; int64_t capLenIn = 0; unsigned char* d_foundIn = nullptr; int64_t capFoundIn = 0;
  int64_t lastIvTotal = 0, lastIvReads = -1, lastFoundReads = -1, lastListReads = -1, lastListWords = 0, lastTooManyUnits = -1;
  // -s (selective alignment) work areas
  uint32_t* d_txpOff = nullptr; int32_t* d_txpLen = nullptr;
  unsigned char* d_selscr = nullptr; int64_t capSelScr = 0;
  uint8_t* d_pk1 = nullptr; uint8_t* d_pk2 = nullptr; int64_t capPk1 = 0, capPk2 = 0;          // 2-bit packed reads as uploaded (qm_map_*_packed)
  qm_pack_exc* d_exc1 = nullptr; qm_pack_exc* d_exc2 = nullptr; int64_t capExc1 = 0, capExc2 = 0;
  unsigned char* d_kswRows = nullptr; int64_t capKswRows = 0;   // -s: alignment blocks of the device-memory edition (long reads, band beyond 97)
  long long* d_todoq = nullptr; int64_t capTodoq = 0; long long* d_todoq2 = nullptr; int64_t capTodoq2 = 0;          // -s: reads the packed list kernel left for the one-read-per-wavefront kernel
  long long* d_slowq = nullptr; int64_t capSlowq = 0;          // -s slow pass: queue, per-wave scratch descriptors and their memory
  unsigned char* d_dyn = nullptr; int64_t capDyn = 0; unsigned char* d_dynmem = nullptr; int64_t capDynMem = 0;
  long long* d_toff = nullptr; int64_t capToff = 0;
  qm_hit* d_tmp = nullptr; int64_t capTmp = 0; unsigned char* d_sides = nullptr; int64_t capSides = 0; int* d_tsc = nullptr; int64_t capTsc = 0;
  int* d_tref = nullptr; int64_t capTref = 0; u64* d_torder = nullptr; int64_t capTorder = 0; unsigned char* d_tasks = nullptr; int64_t capTasks = 0;
  // qm_fetch_stages: CSR offsets of the per-read interval records / list words (scans queued behind stage A), their totals
  // (pinned: they arrive with stage B's synchronisation), the compacted copies
  long long* d_ivcsr = nullptr; int64_t capIvcsr = 0; long long* d_lcsr = nullptr; int64_t capLcsr = 0;
  qm_sa_interval_hit* d_ivC = nullptr; int64_t capIvC = 0; u64* d_wordsC = nullptr; int64_t capWordsC = 0;
  long long* h_tot = nullptr; int64_t stReads = -1, stUnits = -1;
  // last result
  int64_t lastUnits = -1, lastHits = 0; bool lastPaired = false;
  double lastMapMs = 0, lastTotalMs = 0;
  int64_t lastSelQuestions = 0, lastKswTasks = 0, lastStripTasks = 0;             // -s: alignment questions beyond PERFECT chains of the last call, ksw2 alignments run for them
  int64_t lastDefer[4] = {0, 0, 0, 0};            // QM_STAT_DEFER_*
  int64_t lastNPass = 0;                          // QM_STAT_N_PASS_READS: reads the N-aware pass of stage A mapped
  int64_t lastDuoPairs = -1, lastDuoMerged = 0;   // pairs the pair kernel was launched over (-1: not used), pairs it merged itself
  int64_t lastRelaunches = 0, lastSlowReads = 0, lastLeanReads = -1, lastLeanDeferred = 0;   // lastLeanReads: reads the lean kernel was launched over (-1: not used)
  // qm_map_device on a large batch: its parts on helper contexts of the same replica, in flight together (map_device_split)
  uint32_t flags = 0; bool isHelper = false;
  std::vector<qm_ctx*> helpers; struct SplitPool* pool = nullptr;
};

// a few parked threads that run one job per part and call (thread exit in a process with device mappings is slow: they live with the context)
struct SplitPool {
  std::vector<std::thread> th; std::mutex mu; std::condition_variable cv, done;
  std::function<void(int)> job; unsigned long gen = 0; int pending = 0, parts = 0; bool stop = false;
  explicit SplitPool(int n) {
    for (int i = 0; i < n; ++i) th.emplace_back([this, i]() {
      unsigned long seen = 0;
      for (;;) {
        std::function<void(int)> j;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || (gen != seen && i < parts); }); if (stop) return; seen = gen; j = job; }
        j(i);
        { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) done.notify_all(); }
      }
    });
  }
  void run(int n, std::function<void(int)> f) {
    std::unique_lock<std::mutex> lk(mu);
    job = std::move(f); parts = n; pending = n; ++gen;
    cv.notify_all();
    done.wait(lk, [&] { return pending == 0; });
  }
  ~SplitPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
};
// where the parts of a split call meet: every part brings its hit count once its units are counted; the last one sizes the
// caller's result array, then each part writes its hits behind those of the parts before it
struct SplitJoin {
  qm_ctx* owner; int K; std::mutex mu; std::condition_variable cv;
  long long total[8] = {0}, base[9] = {0}; int arrived = 0; bool ready = false; int failed = 0;
  int arrive(int part, long long tot, long long& b, qm_hit*& dst);
  void abort(int rc) { std::lock_guard<std::mutex> lk(mu); if (!failed) failed = rc ? rc : QM_E_STATE; cv.notify_all(); }
  // stage A one part after the other (stagger): part i starts its stage A when part i - 1 has finished its own, so that a part's
  // stage A (scalar unit, waiting) runs beside the alignments of the part before it (VALU) instead of beside another stage A
  bool stagger = false; int turn = 0;
  int wait_turn(int part) {
    if (!stagger) return QM_OK;
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return turn >= part || failed; });
    return failed ? fail(failed, "another part of the batch failed") : QM_OK;
  }
  void pass_turn(int part) { if (!stagger) return; std::lock_guard<std::mutex> lk(mu); if (turn < part + 1) turn = part + 1; cv.notify_all(); }
};

template <typename T>
static int ensure(T*& p, int64_t& cap, int64_t want, int64_t minGrow = 0) {
  if (want <= cap && p) return QM_OK;
  if (p) hipFree(p);
  p = nullptr;
  int64_t nc = want + minGrow;
  if (cap > 0 && nc < cap + cap / 2) nc = cap + cap / 2;     // a buffer that grows again grows by half: callers whose batches vary in size (the
                                                             // compat header's batching service) stop paying a round of hipFree / hipMalloc per new maximum
  if (nc > want + minGrow && hipMalloc((void**)&p, (size_t)nc * sizeof(T)) != hipSuccess) { p = nullptr; (void)hipGetLastError(); nc = want + minGrow; }   // no room for the slack: the exact size
  if (!p && hipMalloc((void**)&p, (size_t)nc * sizeof(T)) != hipSuccess) { cap = 0; return fail(QM_E_NOMEM, "hipMalloc of %lld bytes failed", (long long)(nc * sizeof(T))); }
  cap = nc;
  return QM_OK;
}

extern "C" {

const char* qm_last_error(void) { return g_err; }
const char* qm_version(void) { return "qmap_mi355 0.1 (RapMap 0.6.0 quasimap hot path, gfx950)"; }

int qm_opts_default(qm_opts* o) {
  if (!o) return fail(QM_E_ARG, "null opts");
  o->sensitive = 1; o->strict_check = 1; o->max_num_hits = 200; o->no_orphans = 0; o->no_dovetail = 0;
  o->fuzzy = 0; o->max_interval = 1000; o->sel_aln = 0; o->quasi_cov = 0.0;
  o->hard_filter = 0; o->match_score = 2; o->mismatch_penalty = -4; o->gap_open = 4; o->gap_extend = 2; o->dp_bandwidth = 15;
  o->max_mmp_extension = 7; o->aln_policy = 0; o->min_score_fraction = 0.65; o->consensus_slack = 0.2;
  return QM_OK;
}

// --------------------------------------------------------------------------- index
static bool json_field(const std::string& js, const char* name, std::string& out) {
  std::string key = std::string("\"") + name + "\"";
  size_t p = js.find(key);
  if (p == std::string::npos) return false;
  p = js.find(':', p);
  if (p == std::string::npos) return false;
  ++p;
  while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n')) ++p;
  size_t e = p;
  if (js[p] == '"') { ++p; e = js.find('"', p); }
  else { while (e < js.size() && js[e] != ',' && js[e] != '\n' && js[e] != '}') ++e; }
  out = js.substr(p, e - p);
  while (!out.empty() && (out.back() == ' ' || out.back() == '\r')) out.pop_back();
  return true;
}

// int64 -> uint32 over a large array (a BigSA index at open), on a few threads; false if a value does not fit
static bool narrow_i64(const uint8_t* src, size_t n, uint32_t* dst) {
  const int nt = n > (1u << 22) ? 16 : 1;
  std::atomic<bool> ok{true};
  auto work = [&](int t) {
    const size_t b = n * (size_t)t / nt, e = n * (size_t)(t + 1) / nt;
    bool good = true;
    for (size_t i = b; i < e; ++i) { int64_t v; memcpy(&v, src + 8 * i, 8); if (v < 0 || v >= 0xffffffffLL) good = false; dst[i] = (uint32_t)v; }
    if (!good) ok = false;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return ok;
}

int qm_index_open(const char* dirIn, qm_index** out) {
  if (!dirIn || !out) return fail(QM_E_ARG, "null argument");
  qm_index* ix = new qm_index();
  ix->dir = dirIn;
  if (ix->dir.empty() || ix->dir.back() != '/') ix->dir += '/';
  auto bail = [&](int code) { ix->sa.close(); ix->txp.close(); ix->rsd.close(); ix->hash.close(); ix->bph.close(); ix->val.close(); delete ix; return code; };
  {  // header.json (IndexHeader.hpp:45-57)
    FILE* f = fopen((ix->dir + "header.json").c_str(), "rb");
    if (!f) return bail(fail(QM_E_IO, "cannot open %sheader.json", ix->dir.c_str()));
    std::string js; char buf[4096]; size_t r;
    while ((r = fread(buf, 1, sizeof(buf), f)) > 0) js.append(buf, r);
    fclose(f);
    std::string v;
    if (!json_field(js, "KmerLen", v)) return bail(fail(QM_E_IO, "header.json: no KmerLen"));
    ix->k = atoi(v.c_str());
    if (json_field(js, "BigSA", v)) ix->big = (v == "true");
    if (json_field(js, "PerfectHash", v)) ix->perfect = (v == "true");
    if (json_field(js, "IndexVersion", v) && v != "q5") return bail(fail(QM_E_IO, "index version %s != q5", v.c_str()));
    if (ix->k < 1 || ix->k > 31) return bail(fail(QM_E_IO, "bad k %d", ix->k));
  }

  if (ix->sa.open(ix->dir + "sa.bin")) return bail(fail(QM_E_IO, "cannot map sa.bin"));
  if (ix->txp.open(ix->dir + "txpInfo.bin")) return bail(fail(QM_E_IO, "cannot map txpInfo.bin"));
  if (ix->rsd.open(ix->dir + "rsd.bin")) return bail(fail(QM_E_IO, "cannot map rsd.bin"));
  if (!ix->perfect) { if (ix->hash.open(ix->dir + "hash.bin")) return bail(fail(QM_E_IO, "cannot map hash.bin")); }
  else {
    if (ix->bph.open(ix->dir + "hash_info.bph")) return bail(fail(QM_E_IO, "cannot map hash_info.bph"));
    if (ix->val.open(ix->dir + "hash_info.val")) return bail(fail(QM_E_IO, "cannot map hash_info.val"));
  }
  {  // sa.bin: u64 n, n x i32
    const uint8_t* p = (const uint8_t*)ix->sa.p;
    if (ix->sa.len < 8) return bail(fail(QM_E_IO, "sa.bin truncated"));
    uint64_t n; memcpy(&n, p, 8);
    const size_t isz = ix->big ? 8 : 4;
    if (ix->sa.len != 8 + n * isz) return bail(fail(QM_E_IO, "sa.bin size mismatch"));
    // the device's offsets are 32-bit unsigned: a BigSA index fits as long as its text does (0xffffffff is the "none" value)
    if (n >= 0xfffffffeULL) return bail(fail(QM_E_UNSUPPORTED, "text of %llu characters: more than the 2^32 - 2 the device's 32-bit offsets hold", (unsigned long long)n));
    if (ix->big) {
      ix->saNarrow.resize(n);
      if (!narrow_i64(p + 8, n, ix->saNarrow.data())) return bail(fail(QM_E_IO, "sa.bin: entry out of range"));
      ix->SA = ix->saNarrow.data();
    } else ix->SA = (const uint32_t*)(p + 8);
    ix->nSA = (int64_t)n;
  }
  {  // txpInfo.bin: vector<string>, vector<i32>, string(text), vector<u32>
    const uint8_t* p = (const uint8_t*)ix->txp.p; size_t len = ix->txp.len, off = 0;
    auto rd64 = [&](uint64_t& v) { if (off + 8 > len) return false; memcpy(&v, p + off, 8); off += 8; return true; };
    uint64_t cnt;
    if (!rd64(cnt)) return bail(fail(QM_E_IO, "txpInfo.bin truncated"));
    ix->names.reserve(cnt);
    for (uint64_t i = 0; i < cnt; ++i) {
      uint64_t l; if (!rd64(l) || off + l > len) return bail(fail(QM_E_IO, "txpInfo.bin truncated (names)"));
      ix->names.emplace_back((const char*)p + off, (size_t)l); off += l;
    }
    const size_t isz = ix->big ? 8 : 4;                  // vector<int64_t> in a BigSA index (src/RapMapSAIndexer.cpp:711-722)
    uint64_t c2; if (!rd64(c2) || off + c2 * isz > len) return bail(fail(QM_E_IO, "txpInfo.bin truncated (offsets)"));
    if (ix->big) {
      ix->offNarrow.resize(c2);
      if (!narrow_i64(p + off, c2, ix->offNarrow.data())) return bail(fail(QM_E_IO, "txpInfo.bin: offset out of range"));
      ix->offsets = ix->offNarrow.data();
    } else ix->offsets = (const uint32_t*)(p + off);
    ix->nTxp = (int64_t)c2; off += c2 * isz;
    uint64_t tl; if (!rd64(tl) || off + tl > len) return bail(fail(QM_E_IO, "txpInfo.bin truncated (text)"));
    ix->text = p + off; ix->n = (int64_t)tl; off += tl;
    uint64_t c3; if (!rd64(c3) || off + c3 * 4 > len) return bail(fail(QM_E_IO, "txpInfo.bin truncated (lens)"));
    ix->completeLens = (const uint32_t*)(p + off); off += c3 * 4;
    if ((int64_t)cnt != ix->nTxp || ix->nTxp == 0) return bail(fail(QM_E_IO, "txpInfo.bin: name/offset count mismatch"));
    if (ix->n != ix->nSA) return bail(fail(QM_E_IO, "text length %lld != SA length %lld", (long long)ix->n, (long long)ix->nSA));
    // src/RapMapSAIndex.cpp:151-163
    ix->lens.resize(ix->nTxp);
    for (int64_t i = 0; i + 1 < ix->nTxp; ++i) ix->lens[i] = (int64_t)ix->offsets[i + 1] - 1 - ix->offsets[i];
    ix->lens[ix->nTxp - 1] = (ix->nSA - 1) - ix->offsets[ix->nTxp - 1];
  }
  {  // rsd.bin: u64 nbits + bytes (only validated; the device derives transcript ids from the offsets)
    const uint8_t* p = (const uint8_t*)ix->rsd.p;
    if (ix->rsd.len < 8) return bail(fail(QM_E_IO, "rsd.bin truncated"));
    memcpy(&ix->rsdBits, p, 8);
    if (ix->rsd.len != 8 + (ix->rsdBits + 7) / 8) return bail(fail(QM_E_IO, "rsd.bin size mismatch"));
    if ((int64_t)ix->rsdBits != ix->n) return bail(fail(QM_E_IO, "rsd.bin bits != text length"));
  }
  if (ix->perfect) {
    {  // hash_info.bph: boomphf::mphf::save (include/BooPHF.hpp:1172-1197), bitVector::save (:773-781)
      const uint8_t* p = (const uint8_t*)ix->bph.p; size_t len = ix->bph.len, off = 0;
      auto rd = [&](void* d, size_t n) { if (off + n > len) return false; memcpy(d, p + off, n); off += n; return true; };
      double gamma; int32_t nl;
      if (!rd(&gamma, 8) || !rd(&nl, 4) || !rd(&ix->phLastRank, 8) || !rd(&ix->phNelem, 8) || nl < 2 || nl > 64)
        return bail(fail(QM_E_IO, "hash_info.bph truncated"));
      ix->phLevels.resize(nl);
      for (int i = 0; i < nl; ++i) {
        uint64_t size, nchar, nr;
        if (!rd(&size, 8) || !rd(&nchar, 8) || off + nchar * 8 > len) return bail(fail(QM_E_IO, "hash_info.bph truncated (level)"));
        ix->phLevels[i].nchar = nchar; ix->phLevels[i].words = p + off; off += nchar * 8;
        if (!rd(&nr, 8) || off + nr * 8 > len) return bail(fail(QM_E_IO, "hash_info.bph truncated (ranks)"));
        ix->phLevels[i].nranks = nr; ix->phLevels[i].ranks = p + off; off += nr * 8;
      }
      uint64_t fn; if (!rd(&fn, 8) || off + fn * 16 != len) return bail(fail(QM_E_IO, "hash_info.bph size mismatch"));
      for (uint64_t i = 0; i < fn; ++i) { uint64_t kk, vv; rd(&kk, 8); rd(&vv, 8); ix->phFinal.emplace_back(kk, vv); }
      // level domains are not stored: recomputed exactly like mphf::load (:1219-1230)
      double n = (double)ix->phNelem;
      double proba = 1.0 - pow(((gamma * n - 1) / (gamma * n)), (double)ix->phNelem - 1);
      uint64_t hd = (size_t)(ceil(n * gamma));
      for (int i = 0; i < nl; ++i) {
        uint64_t d = (((uint64_t)(hd * pow(proba, i)) + 63) / 64) * 64;
        if (d == 0) d = 64;
        ix->phLevels[i].domain = d;
        if (ix->phLevels[i].nchar < d / 64) return bail(fail(QM_E_IO, "hash_info.bph: level %d smaller than its domain", i));
      }
    }
    {  // hash_info.val: vector<i32> data_, vector<u8> lens_, sparsepp map overflow_ (FrugalBooMap.hpp:199-213)
      const uint8_t* p = (const uint8_t*)ix->val.p; size_t len = ix->val.len, off = 0;
      uint64_t n1, n2;
      if (len < 8) return bail(fail(QM_E_IO, "hash_info.val truncated"));
      memcpy(&n1, p, 8); off = 8;
      const size_t isz = ix->big ? 8 : 4;                // data_ is vector<IndexT>, overflow_ maps IndexT -> IndexT
      if (off + n1 * isz + 8 > len) return bail(fail(QM_E_IO, "hash_info.val truncated (data)"));
      if (ix->big) {
        ix->phDataNarrow.resize(n1);
        if (!narrow_i64(p + off, n1, ix->phDataNarrow.data())) return bail(fail(QM_E_IO, "hash_info.val: entry out of range"));
        ix->phData = (const uint8_t*)ix->phDataNarrow.data();
      } else ix->phData = p + off;
      off += n1 * isz;
      memcpy(&n2, p + off, 8); off += 8;
      if (off + n2 > len || n1 != n2 || n1 != ix->phNelem) return bail(fail(QM_E_IO, "hash_info.val: size mismatch"));
      ix->phLens = p + off; off += n2;
      auto be = [&](uint64_t& v) {
        if (off + 4 > len) return false;
        v = ((uint64_t)p[off] << 24) | ((uint64_t)p[off + 1] << 16) | ((uint64_t)p[off + 2] << 8) | p[off + 3]; off += 4;
        if (v == 0xFFFFFFFFULL) { if (off + 8 > len) return false; v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | p[off + i]; off += 8; }
        return true;
      };
      uint64_t magic, tsize, nb;
      if (!be(magic) || !be(tsize) || !be(nb) || magic != 0x24687531ULL) return bail(fail(QM_E_IO, "hash_info.val: bad overflow map"));
      off += ((tsize + 31) / 32) * 4;
      if (off + nb * 2 * isz != len) return bail(fail(QM_E_IO, "hash_info.val size mismatch"));
      for (uint64_t i = 0; i < nb; ++i) {
        uint64_t a = 0, b = 0; memcpy(&a, p + off, isz); memcpy(&b, p + off + isz, isz); off += 2 * isz;
        ix->phOverflow.emplace_back((uint32_t)a, (uint32_t)b);
      }
      ix->nKeys = (int64_t)ix->phNelem;
    }
  } else {  // hash.bin: 3 x big-endian u32 (0xFFFFFFFF escapes to u64), group bitmaps, records
    const uint8_t* p = (const uint8_t*)ix->hash.p; size_t len = ix->hash.len, off = 0;
    auto be = [&](uint64_t& v) {
      if (off + 4 > len) return false;
      v = ((uint64_t)p[off] << 24) | ((uint64_t)p[off + 1] << 16) | ((uint64_t)p[off + 2] << 8) | p[off + 3]; off += 4;
      if (v == 0xFFFFFFFFULL) {
        if (off + 8 > len) return false;
        v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | p[off + i];
        off += 8;
      }
      return true;
    };
    uint64_t magic, tsize, nb;
    if (!be(magic) || !be(tsize) || !be(nb)) return bail(fail(QM_E_IO, "hash.bin truncated"));
    if (magic != 0x24687531ULL) return bail(fail(QM_E_IO, "hash.bin bad magic"));
    off += ((tsize + 31) / 32) * 4;
    const size_t rsz = ix->big ? 24 : 16;                // {u64 key, IndexT begin, IndexT end}
    if (off + nb * rsz != len) return bail(fail(QM_E_IO, "hash.bin size mismatch (%zu + %llu*%zu != %zu)", off, (unsigned long long)nb, rsz, len));
    if (ix->big) {
      ix->hashNarrow.resize(nb * 16);
      for (uint64_t i = 0; i < nb; ++i) {
        const uint8_t* r = p + off + 24 * i; uint8_t* w = ix->hashNarrow.data() + 16 * i;
        int64_t b, e; memcpy(&b, r + 8, 8); memcpy(&e, r + 16, 8);
        if (b < 0 || e < b || e > (int64_t)ix->nSA) return bail(fail(QM_E_IO, "hash.bin: interval out of range"));
        const uint32_t b32 = (uint32_t)b, e32 = (uint32_t)e;
        memcpy(w, r, 8); memcpy(w + 8, &b32, 4); memcpy(w + 12, &e32, 4);
      }
      ix->hashRecs = ix->hashNarrow.data();
    } else ix->hashRecs = p + off;
    ix->nKeys = (int64_t)nb;
  }
  *out = ix;
  return QM_OK;
}

int qm_index_close(qm_index* ix) {
  if (!ix) return QM_OK;
  {   // a later index may be allocated at this address: nothing registered under it may survive it
    std::lock_guard<std::mutex> lk(g_repMu);
    for (auto it = g_reps.begin(); it != g_reps.end();) { if (it->first.first == ix) it = g_reps.erase(it); else ++it; }
    for (auto it = g_repBuild.begin(); it != g_repBuild.end();) { if (it->first.first == ix) it = g_repBuild.erase(it); else ++it; }
  }
  ix->sa.close(); ix->txp.close(); ix->rsd.close(); ix->hash.close(); ix->bph.close(); ix->val.close();
  delete ix;
  return QM_OK;
}

int qm_index_info_get(const qm_index* ix, qm_index_info* info) {
  if (!ix || !info) return fail(QM_E_ARG, "null argument");
  info->k = ix->k; info->big_sa = ix->big; info->perfect_hash = ix->perfect; info->pad = 0;
  info->text_len = ix->n; info->n_txps = ix->nTxp; info->n_keys = ix->nKeys;
  return QM_OK;
}
const char* qm_index_txp_name(const qm_index* ix, int64_t tid) {
  if (!ix || tid < 0 || tid >= ix->nTxp) return nullptr;
  return ix->names[tid].c_str();
}
int64_t qm_index_txp_len(const qm_index* ix, int64_t tid) {
  if (!ix || tid < 0 || tid >= ix->nTxp) return -1;
  return ix->lens[tid];
}

int qm_index_arrays(const qm_index* ix, const uint8_t** text, int64_t* text_len, const int32_t** txp_offsets,
                    int64_t* n_txps) {
  if (!ix) return fail(QM_E_ARG, "null index");
  if (text) *text = ix->text;
  if (text_len) *text_len = ix->n;
  if (txp_offsets) *txp_offsets = (const int32_t*)ix->offsets;      // unsigned values for a BigSA index
  if (n_txps) *n_txps = ix->nTxp;
  return QM_OK;
}

int qm_index_raw(const qm_index* ix, int which, const void** data, int64_t* count) {
  if (!ix || !data || !count) return fail(QM_E_ARG, "null argument");
  switch (which) {
    case QM_RAW_SA: *data = ix->SA; *count = ix->nSA; break;
    case QM_RAW_HASH: *data = ix->perfect ? nullptr : ix->hashRecs; *count = ix->perfect ? 0 : ix->nKeys; break;
    case QM_RAW_COMPLETE_LENS: *data = ix->completeLens; *count = ix->nTxp; break;
    default: return fail(QM_E_ARG, "qm_index_raw: unknown array %d", which);
  }
  return QM_OK;
}

// --------------------------------------------------------------------------- context
int qm_ctx_destroy(qm_ctx* c) {
  if (!c) return QM_OK;
  hipSetDevice(c->device);
  delete c->pool; c->pool = nullptr;
  for (qm_ctx* h : c->helpers) qm_ctx_destroy(h);
  c->helpers.clear();
  if (!c->rep) {               // creation failed half-way: the index arrays are still this context's own
    void* own[] = {c->d_text, c->d_SA, c->d_sainfo, c->d_slots, c->d_txpOff, c->d_txpLen, c->d_saext};
    for (void* p : own) if (p) hipFree(p);
    for (void* p : c->phAllocs) if (p) hipFree(p);
  }
  void* ptrs[] = {c->d_cnt, c->d_lcnt, c->d_loff, c->d_lists, c->d_hits, c->d_offs,
                  c->d_scal, c->d_skip, c->d_gscr, c->d_scanTmp, c->d_seq1, c->d_seq2, c->d_off1, c->d_off2, c->d_iv, c->d_ivcnt, c->d_ivoff, c->d_found, c->d_tooMany, c->d_ivIn, c->d_ivInOff, c->d_lenIn, c->d_foundIn,
                  c->d_selscr, c->d_kswRows, c->d_pk1, c->d_pk2, c->d_exc1, c->d_exc2, c->d_slowq, c->d_todoq, c->d_todoq2, c->d_dyn, c->d_dynmem, c->d_toff, c->d_tmp, c->d_sides, c->d_tsc, c->d_tref, c->d_torder, c->d_tasks};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->evA) hipEventDestroy(c->evA);
  if (c->evB) hipEventDestroy(c->evB);
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->copyStream) hipStreamDestroy(c->copyStream);
  if (c->planStream) hipStreamDestroy(c->planStream);
  for (hipEvent_t e : c->evPlan) if (e) hipEventDestroy(e);
  if (c->d_ntk) hipFree(c->d_ntk);
  if (c->d_gslots) hipFree(c->d_gslots);
  if (c->d_ivcsr) hipFree(c->d_ivcsr);
  if (c->d_lcsr) hipFree(c->d_lcsr);
  if (c->d_ivC) hipFree(c->d_ivC);
  if (c->d_wordsC) hipFree(c->d_wordsC);
  if (c->h_tot) hipHostFree(c->h_tot);
  if (c->evCopy) hipEventDestroy(c->evCopy);
  if (c->evP0) hipEventDestroy(c->evP0);
  if (c->evP1) hipEventDestroy(c->evP1);
  if (c->evStage[0]) hipEventDestroy(c->evStage[0]);
  if (c->evStage[1]) hipEventDestroy(c->evStage[1]);
  if (c->h_stage) hipHostFree(c->h_stage);
  {   // the last context on (index, device) frees the replica
    std::lock_guard<std::mutex> lk(g_repMu);
    c->rep.reset();
    for (auto it = g_reps.begin(); it != g_reps.end();) { if (it->second.expired()) it = g_reps.erase(it); else ++it; }
  }
  delete c;
  return QM_OK;
}

int qm_ctx_create(const qm_index* ix, int device_id, qm_ctx** out) { return qm_ctx_create_ex(ix, device_id, 0, out); }

static bool ensure_saext2(qm_ctx* c);
int qm_ctx_create_ex(const qm_index* ix, int device_id, uint32_t flags, qm_ctx** out) {
  if (!ix || !out) return fail(QM_E_ARG, "null argument");
  const bool phCompact = ix->perfect && (flags & QM_CTX_PH_COMPACT);
  const int repKey = device_id * 2 + (phCompact ? 1 : 0);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(QM_E_NOGPU, "no HIP device visible");
  if (device_id < 0 || device_id >= ndev) return fail(QM_E_ARG, "device %d out of range (%d devices)", device_id, ndev);
  HIPCHK(hipSetDevice(device_id));
  qm_ctx* c = new qm_ctx();
  c->ix = ix; c->device = device_id; c->flags = flags;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->numCU = prop.multiProcessorCount;
#define CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { int rc = fail(QM_E_NOGPU, "%s: %s", #x, hipGetErrorString(_e)); qm_ctx_destroy(c); return rc; } } while (0)
  CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c->planStream, hipStreamNonBlocking));
  for (hipEvent_t& e : c->evPlan) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CK(hipMalloc((void**)&c->d_ntk, 3 * QM_SEL_CHUNKS_B * sizeof(u64)));
  CK(hipEventCreateWithFlags(&c->evCopy, hipEventDisableTiming));
  CK(hipEventCreate(&c->ev0)); CK(hipEventCreate(&c->ev1)); CK(hipEventCreate(&c->evA)); CK(hipEventCreate(&c->evB));
  CK(hipEventCreate(&c->evP0)); CK(hipEventCreate(&c->evP1));
  CK(hipMalloc((void**)&c->d_scal, QM_SC_WORDS * sizeof(u64)));
  CK(hipMalloc((void**)&c->d_skip, QM_SKIP_CAP * sizeof(u64)));
  // one builder per (index, device image) at a time: a second thread that asks for the same replica while the first one is
  // still uploading waits here and then shares it
  std::shared_ptr<std::mutex> buildMu;
  {
    std::lock_guard<std::mutex> lk(g_repMu);
    auto& m = g_repBuild[std::make_pair(ix, repKey)];
    if (!m) m = std::make_shared<std::mutex>();
    buildMu = m;
  }
  std::lock_guard<std::mutex> buildLock(*buildMu);
  {
    std::lock_guard<std::mutex> lk(g_repMu);
    auto it = g_reps.find(std::make_pair(ix, repKey));
    if (it != g_reps.end()) c->rep = it->second.lock();
    if (c->rep) {
      Replica& R = *c->rep;
      c->d_text = R.d_text; c->d_SA = R.d_SA; c->d_sainfo = R.d_sainfo; c->d_slots = R.d_slots; c->cap = R.cap; c->d_ph = R.d_ph; c->hPh = R.hPh;
      c->d_txpOff = R.d_txpOff; c->d_txpLen = R.d_txpLen; c->devBytes = R.devBytes;
      { std::lock_guard<std::mutex> l2(R.sanextMu); c->d_sanext = R.d_sanext; }
      c->d_saext = R.d_saext;
    }
  }
  if (c->rep) {
    if (flags & QM_CTX_WIDE_READS) (void)ensure_saext2(c);   // (a replica without room for it goes on with the general kernels)
    *out = c;
    return QM_OK;
  }
  const size_t pad = 256;
  CK(hipMalloc((void**)&c->d_text, (size_t)ix->n + pad));
  CK(hipMemsetAsync(c->d_text + ix->n, 0, pad, c->stream));
  CK(hipMemcpyAsync(c->d_text, ix->text, (size_t)ix->n, hipMemcpyHostToDevice, c->stream));
  CK(hipMalloc((void**)&c->d_SA, (size_t)ix->nSA * 4));
  CK(hipMemcpyAsync(c->d_SA, ix->SA, (size_t)ix->nSA * 4, hipMemcpyHostToDevice, c->stream));
  CK(hipMalloc(&c->d_sainfo, (size_t)ix->nSA * sizeof(SaInfo)));
  uint32_t* d_offsets = nullptr; void* d_recs = nullptr;
  CK(hipMalloc((void**)&d_offsets, (size_t)ix->nTxp * 4));
  CK(hipMemcpyAsync(d_offsets, ix->offsets, (size_t)ix->nTxp * 4, hipMemcpyHostToDevice, c->stream));
  CK(qmk_build_sainfo(c->d_SA, ix->nSA, d_offsets, ix->nTxp, c->d_sainfo, c->stream));
  if (!getenv("QM_NO_SAEXT") && ix->nSA > 0 && ix->nTxp < (1LL << QM_EXT_TID_BITS)) {
    // the packed characters behind every suffix's k-mer: an MMP extension becomes one trip instead of two (saext_entry);
    // 32 bytes per suffix-array entry.  QM_NO_SAEXT (profiling): without the table, extensions read suffix array and text.
    if (hipMalloc(&c->d_saext, (size_t)ix->nSA * sizeof(SaExt)) != hipSuccess) { c->d_saext = nullptr; (void)hipGetLastError(); }   // no room: the text path
    else { CK(qmk_build_saext(c->d_text, ix->n, c->d_SA, ix->nSA, ix->k, c->d_sainfo, c->d_saext, c->stream)); }
  }
  if (!ix->perfect) {
    c->cap = bucket_count(ix->nKeys);                  // 64-byte buckets of two canonical entries, at least two buckets per key
    CK(hipMalloc(&c->d_slots, c->cap * sizeof(Bucket)));
    CK(hipMalloc(&d_recs, (size_t)(ix->nKeys > 0 ? ix->nKeys : 1) * 16));
    if (ix->nKeys > 0) CK(hipMemcpyAsync(d_recs, ix->hashRecs, (size_t)ix->nKeys * 16, hipMemcpyHostToDevice, c->stream));
    CK(qmk_build_slots(d_recs, ix->nKeys, c->d_slots, c->cap, ix->k, c->stream));
  } else {
    // flatten BooPHF + FrugalBooMap: all levels' words / rank samples concatenated, small maps re-hashed
    c->cap = 1;
    PhIndex P; memset(&P, 0, sizeof(P));
    const int nl = (int)ix->phLevels.size();
    std::vector<PhLevelIn> lin((size_t)nl);
    for (int i = 0; i < nl; ++i) {
      lin[i].words = (const uint64_t*)ix->phLevels[i].words; lin[i].nchar = ix->phLevels[i].nchar; lin[i].domain = ix->phLevels[i].domain;
      lin[i].ranks = (const uint64_t*)ix->phLevels[i].ranks; lin[i].nranks = ix->phLevels[i].nranks;
    }
    std::vector<uint64_t> blocks, tab;
    if (!ph_flatten_blocks(lin, blocks, tab)) { int rc = fail(QM_E_IO, "hash_info.bph: rank samples do not match the bit arrays"); qm_ctx_destroy(c); return rc; }
    auto dalloc = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr; c->phAllocs.push_back(q); c->devBytes += (int64_t)bytes; return q; };
    u64* dW = (u64*)dalloc(blocks.size() * 8); u64* dT = (u64*)dalloc(tab.size() * 8);
    PhRec* dRec = (PhRec*)dalloc((size_t)ix->phNelem * sizeof(PhRec));
    unsigned int* dD = nullptr; unsigned char* dL = nullptr; // staging for the record builder
    if (hipMalloc((void**)&dD, (size_t)(ix->phNelem ? ix->phNelem : 1) * 4) != hipSuccess || hipMalloc((void**)&dL, (size_t)(ix->phNelem ? ix->phNelem : 1)) != hipSuccess) dD = nullptr;
    if (!dW || !dT || !dRec || !dD || !dL) { int rc = fail(QM_E_NOMEM, "hipMalloc (perfect hash) failed"); qm_ctx_destroy(c); return rc; }
    CK(hipMemcpyAsync(dW, blocks.data(), blocks.size() * 8, hipMemcpyHostToDevice, c->stream));
    CK(hipMemcpyAsync(dT, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->stream));
    CK(hipMemcpyAsync(dD, ix->phData, (size_t)ix->phNelem * 4, hipMemcpyHostToDevice, c->stream));
    CK(hipMemcpyAsync(dL, ix->phLens, (size_t)ix->phNelem, hipMemcpyHostToDevice, c->stream));
    {
      DevIndex dix; memset(&dix, 0, sizeof(dix));
      dix.text = c->d_text; dix.n = ix->n; dix.SA = c->d_SA; dix.nSA = ix->nSA; dix.k = ix->k;
      CK(qmk_build_phrecs(dD, dL, ix->phNelem, &dix, dRec, c->stream));
      CK(hipStreamSynchronize(c->stream));
      hipFree(dD); hipFree(dL);
    }
    auto mix = [](u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; };
    std::vector<OvfSlot> ov; std::vector<Slot> fin;
    {
      u64 cap = 16; while (cap < ix->phOverflow.size() * 2) cap <<= 1;
      ov.assign(cap, OvfSlot{~0u, 0u});
      for (auto& kv : ix->phOverflow) { u64 j = mix((u64)kv.first) & (cap - 1); while (ov[j].key != ~0u) j = (j + 1) & (cap - 1); ov[j].key = kv.first; ov[j].val = kv.second; }
      P.ovfMask = cap - 1;
      u64 fc = 16; while (fc < ix->phFinal.size() * 2) fc <<= 1;
      Slot e; e.key = ~0ULL; e.lb = 0; e.ub = 0;
      fin.assign(fc, e);
      for (auto& kv : ix->phFinal) { u64 j = mix(kv.first) & (fc - 1); while (fin[j].key != ~0ULL) j = (j + 1) & (fc - 1); fin[j].key = kv.first; fin[j].lb = (int)kv.second; }
      P.finMask = fc - 1;
    }
    OvfSlot* dO = (OvfSlot*)dalloc(ov.size() * sizeof(OvfSlot)); Slot* dF = (Slot*)dalloc(fin.size() * sizeof(Slot));
    void* dP = dalloc(sizeof(PhIndex));
    if (!dO || !dF || !dP) { int rc = fail(QM_E_NOMEM, "hipMalloc (perfect hash) failed"); qm_ctx_destroy(c); return rc; }
    CK(hipMemcpyAsync(dO, ov.data(), ov.size() * sizeof(OvfSlot), hipMemcpyHostToDevice, c->stream));
    CK(hipMemcpyAsync(dF, fin.data(), fin.size() * sizeof(Slot), hipMemcpyHostToDevice, c->stream));
    P.blocks = dW; P.levelTab = dT; P.recs = dRec; P.ovf = dO; P.fin = dF;
    P.lastbitsetrank = ix->phLastRank; P.nelem = ix->phNelem; P.nb_levels = nl;
    if (phCompact && !getenv("QM_PH_NO_FILTER")) {
      // the membership pre-filter of the compact image (PhIndex::filter): absent k-mers -- most lookups -- are rejected after
      // one sector instead of a walk through the levels.  QM_PH_NO_FILTER (profiling): the bare reference structure.
      const u64 fw = ph_filter_words(ix->phNelem);
      u64* dFil = (u64*)dalloc(fw * 8);
      if (!dFil) { int rc = fail(QM_E_NOMEM, "hipMalloc (perfect hash filter) failed"); qm_ctx_destroy(c); return rc; }
      CK(hipMemsetAsync(dFil, 0, fw * 8, c->stream));
      CK(qmk_build_phfilter(dRec, (long long)ix->phNelem, dFil, fw - 1, ix->k, c->stream));
      P.filter = dFil; P.filterMask = fw - 1;
    }
    CK(hipMemcpyAsync(dP, &P, sizeof(P), hipMemcpyHostToDevice, c->stream));
    CK(hipStreamSynchronize(c->stream));    // P, tab, ov, fin are locals
    c->d_ph = dP; c->hPh = P;
    if (!phCompact) {
      // Default device image of a -p index: the same one-sector bucket table a dense index gets, filled from the MPHF's own
      // records after each was looked up through the BooPHF walk (so the table is known to answer like FrugalBooMap::find on
      // this file).  288 GB of HBM make the frugal structure unnecessary on this device; QM_CTX_PH_COMPACT keeps it.
      c->cap = bucket_count((long long)ix->phNelem);
      unsigned long long* dBad = nullptr; unsigned long long hBad = 0;
      CK(hipMalloc(&c->d_slots, c->cap * sizeof(Bucket)));
      CK(hipMalloc((void**)&dBad, sizeof(unsigned long long)));
      DevIndex dix; memset(&dix, 0, sizeof(dix));
      dix.text = c->d_text; dix.n = ix->n; dix.SA = c->d_SA; dix.nSA = ix->nSA; dix.k = ix->k; dix.ph = (const PhIndex*)dP; dix.phv = P;
      CK(qmk_build_slots_from_ph(&dix, (long long)ix->phNelem, c->d_slots, c->cap, dBad, c->stream));
      CK(hipMemcpyAsync(&hBad, dBad, sizeof(hBad), hipMemcpyDeviceToHost, c->stream));
      CK(hipStreamSynchronize(c->stream));
      hipFree(dBad);
      if (hBad) { int rc = fail(QM_E_IO, "hash_info.bph / hash_info.val: %llu k-mers are not found by the perfect hash they were stored with", hBad); qm_ctx_destroy(c); return rc; }
      for (void* q : c->phAllocs) if (q) hipFree(q);
      c->phAllocs.clear(); c->d_ph = nullptr; c->devBytes = (int64_t)(c->cap * sizeof(Bucket));
    }
  }
  CK(hipStreamSynchronize(c->stream));
  c->d_txpOff = d_offsets;                               // kept: -s reads transcript sequences by (offset, length)
  {
    std::vector<int32_t> l32((size_t)ix->nTxp);
    for (int64_t t = 0; t < ix->nTxp; ++t) l32[(size_t)t] = (int32_t)ix->lens[(size_t)t];
    CK(hipMalloc((void**)&c->d_txpLen, (size_t)(ix->nTxp ? ix->nTxp : 1) * 4));
    CK(hipMemcpy(c->d_txpLen, l32.data(), (size_t)ix->nTxp * 4, hipMemcpyHostToDevice));
  }
  if (d_recs) hipFree(d_recs);
  c->devBytes += ix->n + pad + ix->nSA * 4 + ix->nSA * (int64_t)sizeof(SaInfo) + (int64_t)(ix->perfect ? 0 : c->cap * sizeof(Bucket))
                 + (c->d_saext ? ix->nSA * (int64_t)sizeof(SaExt) : 0);
#undef CK
  {
    auto R = std::make_shared<Replica>();
    R->device = device_id; R->d_text = c->d_text; R->d_SA = c->d_SA; R->d_sainfo = c->d_sainfo; R->d_slots = c->d_slots; R->cap = c->cap;
    R->d_ph = c->d_ph; R->hPh = c->hPh; R->phAllocs.swap(c->phAllocs); R->d_txpOff = c->d_txpOff; R->d_txpLen = c->d_txpLen; R->devBytes = c->devBytes;
    R->d_saext = c->d_saext;
    std::lock_guard<std::mutex> lk(g_repMu);
    c->rep = R;
    g_reps[std::make_pair(ix, repKey)] = R;
  }
  // the wide extension table with the replica when the caller knows reads of 129 .. 256 characters are coming: otherwise the first call
  // that has such reads builds it (0.2 s for config 2) inside that call
  if (flags & QM_CTX_WIDE_READS) (void)ensure_saext2(c);
  *out = c;
  return QM_OK;
}

int64_t qm_ctx_device_bytes(const qm_ctx* c) { return c ? c->devBytes : 0; }
int qm_ctx_set_debug(qm_ctx* c, int keep) { if (!c) return fail(QM_E_ARG, "null ctx"); c->debug = keep; return QM_OK; }

static int check_opts(const qm_opts* o) {
  if (!o) return fail(QM_E_ARG, "null opts");
  if (o->max_num_hits < 0 || o->max_interval < 1) return fail(QM_E_ARG, "bad max_num_hits / max_interval");
  return QM_OK;
}

// Host-buffer callers (qm_map_pairs / qm_map_reads) hand stage A its input chunk by chunk: `upload(u0, u1)` queues the
// characters of units [u0, u1) on the copy stream, the kernel for those units waits for that copy only, so the upload of
// chunk i + 1 runs under the kernel of chunk i.  Device-buffer callers pass no feeder: one launch over everything.
struct ChunkFeeder {
  int64_t chunk;                                             // units per chunk
  int (*upload)(void* self, int64_t u0, int64_t u1);         // QM_OK or an error code (already recorded with fail())
  void* self;
};

// What a call asks of the path.  The fused calls (qm_map_pairs / _reads / _device) run everything; the stage entries run one
// of the reference's three entry points on its own (SACollector::operator(), hitsToMappingsSimple, mergeLeftRightHits[Fuzzy]).
enum { QM_RUN_FUSED = 0, QM_RUN_COLLECT = 1, QM_RUN_FROM_INTERVALS = 2 };
struct RunReq {
  int mode = QM_RUN_FUSED;
  bool keepIntervals = false;     // SA-interval hits kept for qm_fetch_intervals
  bool keepFound = false;         // foundHit per read kept for qm_fetch_found
  bool mergeOnly = false;         // stage B without the caller-level bookkeeping, tooMany flags kept
  bool stageView = false;         // qm_map_pairs_stages: queue the CSR scans of the per-read outputs behind stage A (qm_fetch_stages)
  bool longReads = false;         // the batch holds reads beyond QM_MAX_READ_LEN (set by map_device_impl from max_read_len)
  int shortLen = 0;               // the longest read that is not beyond QM_MAX_READ_LEN (0: unknown)
  // QM_RUN_FROM_INTERVALS: device arrays
  const qm_sa_interval_hit* ivIn = nullptr; const long long* ivInOff = nullptr; const int* lenIn = nullptr; const unsigned char* foundIn = nullptr;
  SplitJoin* join = nullptr; int part = 0;   // one part of a split call (map_device_split): the hits go to the caller's array
  bool noDuo = false;             // this part takes qm_lean_kernel even where the pair kernel could (map_device_split: parts of both kinds in flight)
};

static DevIndex dev_index(const qm_ctx* c) {
  DevIndex ix; ix.text = c->d_text; ix.n = c->ix->n; ix.SA = c->d_SA; ix.nSA = c->ix->nSA;
  ix.sainfo = (const SaInfo*)c->d_sainfo; ix.slots = (const Bucket*)c->d_slots; ix.hmask = c->cap - 1; ix.ph = (const PhIndex*)c->d_ph; ix.k = c->ix->k;
  memset(&ix.phv, 0, sizeof(ix.phv)); if (c->d_ph) ix.phv = c->hPh;
  ix.sanext = c->d_sanext; ix.saext = (const SaExt*)c->d_saext; ix.saext2 = (const SaExt2*)c->d_saext2;
  return ix;
}

// ---- stage A: one wavefront per read (collector + hits->mappings, or one of the two alone), with its retries: the per-read
// lists or the interval output outgrew their buffers (grow, redo), -s reads left on the slow queue (second, small launch).
// the longest read a call takes: QM_MAX_LONG_READ_LEN; with -s only while the band's ring edition of the alignment kernel has a
// long-image form (--dpBandwidth 0 .. 97; the full-band ring holds every column of a 512-base alignment and no more)
static int len_limit(const qm_opts* o) {
  (void)o;                                                   // (round 4: with -s too, whatever the band -- a band beyond 97 takes the device-memory edition of the alignment kernel)
  return QM_MAX_LONG_READ_LEN;
}

// The wide extension table (SaExt2: 224 characters behind every suffix's k-mer, 64 bytes per suffix-array entry) for the one-read-per-
// wavefront lean kernel: built by the first call that has reads of 129 .. 256 characters, shared by the replica's contexts; a replica
// that has no room for it (or no SaExt) goes on with the general kernels.
static bool ensure_saext2(qm_ctx* c) {
  if (c->d_saext2) return true;
  if (!c->rep || !c->d_saext || c->ix->nSA <= 0) return false;
  Replica& R = *c->rep;
  std::lock_guard<std::mutex> lk(R.sanextMu);
  if (!R.d_saext2 && !R.saext2Tried) {
    R.saext2Tried = true;
    void* p = nullptr;
    // room for the table AND a margin for the work buffers of the replica's contexts (lists, hits, scratch: they grow with the batches);
    // a replica that cannot spare it stays with the general kernels for such reads
    size_t freeB = 0, totalB = 0;
    const size_t need = (size_t)c->ix->nSA * qmk_saext2_bytes(), margin = (size_t)8 << 30;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess || freeB < need + margin) { (void)hipGetLastError(); return false; }
    if (hipMalloc(&p, need) != hipSuccess) { (void)hipGetLastError(); return false; }
    hipError_t e = qmk_build_saext2(c->d_text, c->ix->n, c->d_SA, c->ix->nSA, c->ix->k, c->d_sainfo, p, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { hipFree(p); (void)hipGetLastError(); return false; }
    R.d_saext2 = p; R.devBytes += c->ix->nSA * (int64_t)qmk_saext2_bytes();
  }
  c->d_saext2 = R.d_saext2; c->devBytes = R.devBytes;
  return c->d_saext2 != nullptr;
}

// The general kernels' per-wave scratch in device memory (QM_GSCR_U64 words, 112 KB) for a launch of `grid` blocks: one per launched
// wave while that is no more than the waves that can be resident at once (64 per CU, generously); beyond that -- the oversubscribed
// grids -- one per SLOT, and the waves take and return slots as they start and end (ReadBatch::gslots).  Returns the flags, or null.
// XCDs of the device a context launches on: 32 CUs each on gfx950 (a whole MI355X: 8; its DPX / QPX / CPX partitions: 4 / 2 / 1)
static int xcds_of(const qm_ctx* c) { const int x = c->numCU / 32; return x < 1 ? 1 : (x > 8 ? 8 : x); }
static int gscr_for(qm_ctx* c, int grid, unsigned*& slots, int& nslots) {
  // (64 slots per CU = 2 048 per XCD of 32 CUs, twice the 1 024 wavefronts that can be resident there, whatever the partition)
  const int64_t waves = (int64_t)grid * 4, cap = (int64_t)c->numCU * 64;
  slots = nullptr; nslots = 0;
  const char* fe = getenv("QM_GSCR_SLOTS");                  // (tests: slots for launches that would not need them)
  const bool force = fe && atoi(fe) != 0;
  int rc = ensure(c->d_gscr, c->capGrid, (waves > cap || force ? cap : waves) * QM_GSCR_U64);
  if (rc || (waves <= cap && !force)) return rc;
  if (!c->d_gslots && hipMalloc((void**)&c->d_gslots, (size_t)cap * sizeof(unsigned)) != hipSuccess) { c->d_gslots = nullptr; return fail(QM_E_NOMEM, "hipMalloc of the scratch slots' flags failed"); }
  HIPCHK(hipMemsetAsync(c->d_gslots, 0, (size_t)cap * sizeof(unsigned), c->stream));
  slots = c->d_gslots; nslots = (int)cap;
  return QM_OK;
}

static int run_stage_a(qm_ctx* c, const qm_opts* o, const RunReq& rq, int64_t n, const void* d_seq1, const void* d_off1, const void* d_seq2,
                       const void* d_off2, int ns, ChunkFeeder* feeder, u64* hscal) {
  int rc;
  const bool paired = d_seq2 != nullptr;
  const int64_t nreads = paired ? 2 * n : n;
  const int phc = c->d_ph ? 1 : 0;                        // (the compact -p image: its kernels take a larger grid)
  int grid = qmk_map_grid_ex(nreads, c->numCU, phc);
  if ((rc = ensure(c->d_lcnt, c->capLcnt, nreads + 1))) return rc;
  if ((rc = ensure(c->d_loff, c->capLoff, nreads + 1))) return rc;
  // The lean kernel (qm_lean.inl: two reads per wavefront and iteration, reads of up to 128 clean characters) takes the fused default
  // call on a dense table; the reads it marks instead of mapping go through the general kernel in a second, small launch below.
  // It owns no per-wave scratch in device memory: that is only reserved -- for the small grid -- when the second launch happens.
  static const bool leanOff = [] { const char* e = getenv("QM_NO_LEAN"); return e && atoi(e) != 0; }();
  // Reads of 129 .. 256 characters (slot classes 3 and 4) take its wide edition -- one read per wavefront -- once the replica holds the
  // wide extension table.
  static const bool wideOff = [] { const char* e = getenv("QM_NO_LEAN_WIDE"); return e && atoi(e) != 0; }();
  const bool leanBase = !leanOff && rq.mode == QM_RUN_FUSED && o->sensitive && (c->d_slots || c->d_ph) && c->d_saext && c->ix->k <= 31;
  // (a call that keeps the SA-interval records takes the general kernel: the lean kernels do not write them; foundHit they do -- stage views without intervals)
  const bool leanWide = leanBase && !wideOff && (ns == 3 || ns == 4) && (o->sel_aln || !rq.keepIntervals) && ensure_saext2(c);
  const bool useLean = leanBase && !o->sel_aln && (ns == 2 || leanWide) && !rq.keepIntervals;
  // Pairs of such reads take the pair kernel (qm_duo.inl): the two mates walked in lockstep by the two halves of a wavefront, and -- in a
  // plain fused call -- merged there (pair_cnt: stage B's count pass finds the pair done, its write pass expands the records)
  static const bool duoOff = [] { const char* e = getenv("QM_NO_DUO"); return e && atoi(e) != 0; }();
  // (dense table only: on the compact -p image a position's two orientations are two walks of the BooPHF levels, and a lane per position
  // serialises them -- 301 M pairs/s against qm_lean_kernel's 323, profiles/r06/exp_mix.txt)
  static const bool duoPh = [] { const char* e = getenv("QM_DUO_PH"); return e && atoi(e) != 0; }();
  const bool useDuo = useLean && !duoOff && !rq.noDuo && !(c->flags & QM_CTX_NO_PAIR_KERNEL) && paired && ns == 2 && (!c->d_ph || duoPh);
  // (qm_lean_kernel holds both mates of a pair in one wavefront too, but merging there was measured and dropped: that kernel is bound by the CU's scalar unit and
  // the merge's bookkeeping cost it 1.3 ms per 10 M pairs -- 62 instead of 28 spilled scalar registers -- where stage B saved 0.5: profiles/r06/exp_mix.txt)
  const bool duoMerge = useDuo && !rq.mergeOnly && !rq.stageView;
  if (duoMerge) { if ((rc = ensure(c->d_cnt, c->capCnt, n + 1))) return rc; }
  unsigned* gslots = nullptr; int ngslots = 0;
  if (!useLean) {
    // the general kernels' scratch (gscr_for); when even that does not fit next to the index, the launch falls back to the resident grid
    rc = gscr_for(c, grid, gslots, ngslots);
    if (rc == QM_E_NOMEM) { grid = qmk_resident_grid(nreads, c->numCU); rc = gscr_for(c, grid, gslots, ngslots); }
    if (rc) return rc;
  }
  // ... and its -s edition stands in for the chain-scoring collector of a fused -s call (intervals and foundHit out; the list kernels
  // that follow are the same)
  const bool useLeanSel = leanBase && o->sel_aln && (ns == 2 || leanWide);
  if (rq.mode != QM_RUN_COLLECT) {
    int64_t wantLists = nreads * 4 + (int64_t)grid * 4 * QM_CHUNK * 2;   // chunked bump allocator: up to one open chunk per wave
    if (c->capLists < wantLists) { if ((rc = ensure(c->d_lists, c->capLists, wantLists))) return rc; }
  }
  // A fused -s call runs stage A as two launches: the chain-scoring collector alone (its memory-bound walk, at the default
  // kernel's occupancy: without the chaining code it needs half the registers and no LDS scratch) leaves every read's
  // SA-interval hits in HBM, then one wavefront per read turns them into the read's list (sort, slack intersection, chaining).
  const bool twoPass = rq.mode == QM_RUN_FUSED && o->sel_aln != 0;
  const bool wantIv = rq.keepIntervals || rq.mode == QM_RUN_COLLECT || twoPass;
  if (wantIv) {
    if ((rc = ensure(c->d_ivcnt, c->capIvCnt, nreads + 1))) return rc;
    if ((rc = ensure(c->d_ivoff, c->capIvOff, nreads + 1))) return rc;
    const int64_t want = nreads * (o->sel_aln ? 16 : 4) + (int64_t)grid * 4 * QM_IVCHUNK * 2;   // chunked allocator: up to one open chunk per wave
    if (c->capIv < want) { if ((rc = ensure(c->d_iv, c->capIv, want))) return rc; }
  }
  const bool wantFound = rq.keepFound || rq.mode == QM_RUN_COLLECT || twoPass;
  if (wantFound) { if ((rc = ensure(c->d_found, c->capFound, nreads + 1))) return rc; }
  if (o->sel_aln && rq.mode != QM_RUN_FROM_INTERVALS && !c->d_sanext && c->ix->nSA > 0 && c->rep) {
    // first -s call on this replica: the table that turns a capped MMP extension into one trip (sanext_entry); 4 bytes per
    // suffix-array entry, built from text and SA as they sit in HBM, shared by every context of the replica
    Replica& R = *c->rep;
    std::lock_guard<std::mutex> lk(R.sanextMu);
    if (!R.d_sanext) {
      unsigned int* p = nullptr;
      HIPCHK(hipMalloc((void**)&p, (size_t)c->ix->nSA * sizeof(unsigned int)));
      hipError_t e = qmk_build_sanext(c->d_text, c->ix->n, c->d_SA, c->ix->nSA, c->ix->k, p, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { hipFree(p); return fail(QM_E_NOGPU, "building the -s extension table: %s", hipGetErrorString(e)); }
      R.d_sanext = p; R.devBytes += c->ix->nSA * 4;
    }
    c->d_sanext = R.d_sanext; c->devBytes = R.devBytes;
  }
  const DevIndex ix = dev_index(c);
  c->lastRelaunches = 0; c->lastSlowReads = 0; c->lastIvTotal = 0; c->lastLeanReads = useLean ? nreads : -1; c->lastLeanDeferred = 0;
  c->lastDuoPairs = useDuo ? n : -1; c->lastDuoMerged = 0;
  for (int i = 0; i < 4; ++i) c->lastDefer[i] = 0;
  c->lastNPass = 0;
  float leanExtraMs = 0;
  while (true) {
    ReadBatch B; memset(&B, 0, sizeof(B));
    B.seq1 = (const unsigned char*)d_seq1; B.off1 = (const long long*)d_off1;
    B.seq2 = (const unsigned char*)d_seq2; B.off2 = (const long long*)d_off2; B.nreads = nreads;
    B.lcnt = c->d_lcnt; B.loff = c->d_loff; B.lists = c->d_lists; B.cursor = c->d_scal; B.lists_cap = c->capLists;
    B.status = (int*)(c->d_scal + QM_SC_STATUS); B.gscratch = c->d_gscr; B.gslots = gslots; B.ngslots = ngslots; B.gxcd = xcds_of(c); B.skiplist = c->d_skip;
    B.lean_wide = leanWide ? 1 : 0;
    if (duoMerge) { B.pair_cnt = c->d_cnt; B.max_num_hits = o->max_num_hits; B.no_orphans = o->no_orphans; B.no_dovetail = o->no_dovetail; }
    if (wantIv) { B.iv_out = c->d_iv; B.iv_cnt = c->d_ivcnt; B.iv_off = c->d_ivoff; B.iv_cap = c->capIv; }
    if (wantFound) B.found_out = c->d_found;
    B.iv_in = rq.ivIn; B.iv_in_off = rq.ivInOff; B.len_in = rq.lenIn; B.found_in = rq.foundIn;
    B.strict_check = o->strict_check; B.max_interval = o->max_interval; B.quasi_cov = o->quasi_cov; B.sensitive = o->sensitive; B.fuzzy = paired ? o->fuzzy : 0;
    if (rq.mode == QM_RUN_FROM_INTERVALS) B.fuzzy = o->fuzzy;   // the caller says what kind of list it wants (both orientations kept or not)
    if (o->sel_aln) {                                   // -s: chain scoring + per-wave scratch for chaining (qm_sel.inl)
      if ((rc = ensure(c->d_selscr, c->capSelScr, (int64_t)qmk_resident_grid(nreads, c->numCU) * 4 * (int64_t)qmk_sel_scratch_bytes()))) return rc;   // (the list kernels' grids stay within residency)
      B.selscr = (SelScratch*)c->d_selscr;
      B.max_mmp_ext = o->max_mmp_extension > 0 ? o->max_mmp_extension : 7;
      const float cs = (float)o->consensus_slack;        // MappingOpts::consensusSlack is a float (RapMapSAMapper.cpp:138,184-185)
      B.consensus_fraction = cs < 0 ? -cs : ((cs == 0.0) ? 1.0 : (1.0 - cs));   // negative: MappingConfig::consensusFraction itself (qmap_mi355.h)
    }
    HIPCHK(hipMemsetAsync(c->d_scal, 0, QM_SC_WORDS * sizeof(u64), c->stream));
    if (rq.mode == QM_RUN_COLLECT || twoPass) HIPCHK(hipMemsetAsync(c->d_lcnt, 0, (size_t)(nreads + 1) * sizeof(uint32_t), c->stream));   // collector-only kernels write no list lengths: the array only carries the long-read marks
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    auto launch = [&](const ReadBatch& X, int g) -> hipError_t {
      if (useDuo) return qmk_launch_duo(&ix, &X, c->numCU, c->stream);
      if (useLean || (useLeanSel && ix.sanext)) return qmk_launch_lean(&ix, &X, c->numCU, c->stream);
      if (rq.mode == QM_RUN_FROM_INTERVALS) return qmk_h2m(&ix, &X, g, c->numCU, c->stream);
      if (twoPass) return qmk_map_reads_ex(&ix, &X, ns, 1, g, c->numCU, c->stream);
      return qmk_map_reads(&ix, &X, rq.mode == QM_RUN_COLLECT ? -1 : ns, g, c->numCU, c->stream);
    };
    // The N-aware pass (round 6).  The first-pass kernels leave every read with a character outside A C G T; when many of a batch's reads are
    // there for that reason (sequencers write N where a base call failed), qm_lean_kernel's N-aware edition goes over the queue of what was
    // left before the general kernel does: reads whose odd characters are all N are mapped at its rate (lean_iter<..., NQ>: k-mers with an N
    // are stepped over, MMPs end at one), the others are marked again and counted anew.  hscal: the scalars after the first pass -> after this one.
    auto n_pass = [&]() -> int {
      const char* npe = getenv("QM_NPASS_MIN");                      // (read per call: tests switch it; negative: never)
      const long long minDirty = npe ? atoll(npe) : 2048LL;
      if (leanWide || minDirty < 0 || nreads <= 0) return QM_OK;
      const int64_t nq = (int64_t)hscal[QM_SC_LEANQ];
#ifndef QM_TIMING
      const int64_t dirtyReads = (int64_t)hscal[QM_SC_DEFER0];
#else
      const int64_t dirtyReads = nq;
#endif
      if (nq <= 0 || dirtyReads < minDirty || ((int)(hscal[QM_SC_STATUS] & 0xffffffffu) & 23)) return QM_OK;
      int r;
      if ((r = ensure(c->d_slowq, c->capSlowq, nq))) return r;
      HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_SLOWQ, 0, sizeof(u64), c->stream));
      HIPCHK(qmk_collect_lean(c->d_lcnt, nreads, c->d_slowq, (unsigned long long*)(c->d_scal + QM_SC_SLOWQ), c->stream));
      HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_LEANQ, 0, sizeof(u64), c->stream));
#ifndef QM_TIMING
      HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_DEFER0, 0, 4 * sizeof(u64), c->stream));
#endif
      ReadBatch Q = B;
      Q.slowq = c->d_slowq; Q.nreads = nq;
      HIPCHK(hipEventRecord(c->evP0, c->stream));
      HIPCHK(qmk_launch_lean_nq(&ix, &Q, c->numCU, c->stream));
      HIPCHK(hipEventRecord(c->evP1, c->stream));
      HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_SLOWQ, 0, sizeof(u64), c->stream));
      HIPCHK(hipMemcpyAsync(hscal, c->d_scal, QM_SC_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      float t = 0; if (hipEventElapsedTime(&t, c->evP0, c->evP1) == hipSuccess) leanExtraMs += t;
      c->lastNPass = nq - (int64_t)hscal[QM_SC_LEANQ];
      return QM_OK;
    };
    // second pass of a two-pass -s call (also the slow pass's kernel): intervals -> lists
    ReadBatch H = B;
    if (twoPass) {
      H.iv_in = c->d_iv; H.iv_in_off = c->d_ivoff; H.iv_in_cnt = c->d_ivcnt; H.found_in = c->d_found;
      if (!rq.keepIntervals) { }                       // the interval output stays valid either way: it is this pass's input
      H.iv_out = nullptr; H.found_out = nullptr;
    }
    if (feeder && n > 0) {
      // first pass over host buffers: one launch per chunk, each behind its own upload.  A launch sees its chunk through
      // shifted pointers (offsets are absolute, the per-read / per-unit arrays start at the chunk), the bump allocators,
      // the counters and the status word are shared.
      for (int64_t u0 = 0; u0 < n; u0 += feeder->chunk) {
        const int64_t u1 = u0 + feeder->chunk < n ? u0 + feeder->chunk : n;
        if ((rc = feeder->upload(feeder->self, u0, u1))) return rc;
        HIPCHK(hipEventRecord(c->evCopy, c->copyStream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->evCopy, 0));
        ReadBatch C = B;
        const int64_t r0 = paired ? 2 * u0 : u0, r1 = paired ? 2 * u1 : u1;
        C.off1 = B.off1 + u0; if (paired) C.off2 = B.off2 + u0;
        C.nreads = r1 - r0; C.lcnt = B.lcnt + r0; C.loff = B.loff + r0;
        if (C.iv_cnt) { C.iv_cnt = B.iv_cnt + r0; C.iv_off = B.iv_off + r0; }
        if (C.found_out) C.found_out = B.found_out + r0;
        if (C.pair_cnt) C.pair_cnt = B.pair_cnt + u0;
        C.read_base = r0;                                 // (what the launch calls read 0: for the skip list)
        HIPCHK(launch(C, qmk_map_grid_ex(r1 - r0, c->numCU, phc)));
      }
      feeder = nullptr;                                   // a retry finds everything resident
    } else if (nreads > 0) HIPCHK(launch(B, grid));
    if (useLeanSel && ix.sanext && nreads > 0) {
      // what the lean collector marked instead of walking: gathered and walked by the general chain-scoring collector, before the
      // list kernels go over all reads
      HIPCHK(hipMemcpyAsync(hscal, c->d_scal, QM_SC_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if ((rc = n_pass())) return rc;
      const int st0 = (int)(hscal[QM_SC_STATUS] & 0xffffffffu);
#ifndef QM_TIMING
      for (int i = 0; i < 4; ++i) c->lastDefer[i] = (int64_t)hscal[QM_SC_DEFER0 + i];
#endif
      if (hscal[QM_SC_LEANQ] > 0 && !(st0 & 23)) {
        const int64_t nq = (int64_t)hscal[QM_SC_LEANQ];
        if ((rc = ensure(c->d_slowq, c->capSlowq, nq))) return rc;
        HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_SLOWQ, 0, sizeof(u64), c->stream));
        HIPCHK(qmk_collect_lean(c->d_lcnt, nreads, c->d_slowq, (unsigned long long*)(c->d_scal + QM_SC_SLOWQ), c->stream));
        ReadBatch S2 = B;
        S2.slowq = c->d_slowq; S2.nreads = nq;
        const int g2 = qmk_map_grid_ex(nq, c->numCU, phc);
        HIPCHK(qmk_map_reads_ex(&ix, &S2, ns, 1, g2 < grid ? g2 : grid, c->numCU, c->stream));
        HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_SLOWQ, 0, sizeof(u64), c->stream));   // (the passes below gather with the same counter)
        c->lastLeanDeferred = nq;
      }
      c->lastLeanReads = nreads;
    }
    if (twoPass && rq.longReads && nreads > 0) {
      // -s with reads beyond QM_MAX_READ_LEN in the batch: the collector set them aside (map_read); their intervals come from a
      // second, small launch of the 32-slot chain-scoring collector, before the list kernel goes over all reads
      HIPCHK(hipMemcpyAsync(hscal, c->d_scal, QM_SC_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      const int st1 = (int)(hscal[QM_SC_STATUS] & 0xffffffffu);
      if (hscal[QM_SC_SLOWCNT] > 0 && !(st1 & 23)) {
        const int64_t nl_ = (int64_t)hscal[QM_SC_SLOWCNT];
        if ((rc = ensure(c->d_slowq, c->capSlowq, nl_))) return rc;
        HIPCHK(qmk_collect_slow(c->d_lcnt, nreads, c->d_slowq, (unsigned long long*)(c->d_scal + QM_SC_SLOWQ), c->stream));
        ReadBatch S2 = B;
        S2.slowq = c->d_slowq; S2.nreads = nl_;
        const int g2 = qmk_map_grid_ex(nl_, c->numCU, phc);
        HIPCHK(qmk_map_reads(&ix, &S2, -32, g2 < grid ? g2 : grid, c->numCU, c->stream));
        // the list kernel's own slow queue (reads whose intervals overflow its scratch) starts from zero
        HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_SLOWCNT, 0, 3 * sizeof(u64), c->stream));
        c->lastSlowReads = nl_;
      }
    }
    if (twoPass && nreads > 0) {
      const char* pe = getenv("QM_SEL_PACK");              // 0: the one-read-per-wavefront list kernel for every read (A/B timing, tests)
      const bool packed = !(pe && atoi(pe) == 0);
      if (packed) {
        // several reads per wavefront first (qm_selpack.inl); what that kernel cannot take -- hits on both strands, more than 64
        // intervals or suffixes -- it queues, and the one-read-per-wavefront kernel runs over the queue (its length stays on the device)
        if ((rc = ensure(c->d_todoq, c->capTodoq, nreads))) return rc;
        // ... then the wide edition (256 lanes' worth per batch: reads of 150 bp and more) over that queue, which leaves one of its own.
        // A batch of reads beyond 192 characters goes to the wide edition directly: hardly any of them fits the narrow one's 64 lanes
        if ((rc = ensure(c->d_todoq2, c->capTodoq2, nreads))) return rc;
        static const int wideFrom = [] { const char* e = getenv("QM_SEL_WIDE_FROM"); return e ? atoi(e) : 192; }();   // (tuning knob)
        if (rq.shortLen > wideFrom) HIPCHK(qmk_h2m_packw(&ix, &H, nullptr, nullptr, c->d_todoq2, grid, c->numCU, c->stream));
        else {
          HIPCHK(qmk_h2m_pack(&ix, &H, c->d_todoq, grid, c->numCU, c->stream));
          HIPCHK(qmk_h2m_packw(&ix, &H, c->d_todoq, (const unsigned long long*)(c->d_scal + QM_SC_TODO), c->d_todoq2, grid, c->numCU, c->stream));
        }
        ReadBatch T = H; T.slowq = c->d_todoq2; T.nreads_dev = c->d_scal + QM_SC_TODO2;
        HIPCHK(qmk_h2m(&ix, &T, grid, c->numCU, c->stream));
      } else HIPCHK(qmk_h2m(&ix, &H, grid, c->numCU, c->stream));
    }
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipMemcpyAsync(hscal, c->d_scal, QM_SC_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (useLean && (rc = n_pass())) return rc;
    int status = (int)(hscal[QM_SC_STATUS] & 0xffffffffu);
#ifndef QM_TIMING
    if (useLean) for (int i = 0; i < 4; ++i) c->lastDefer[i] = (int64_t)hscal[QM_SC_DEFER0 + i];
#endif
    if (duoMerge) c->lastDuoMerged = (int64_t)hscal[4];        // (numReads so far: the pairs the pair kernel merged; stage B's count pass adds the others)
    if (useLean && hscal[QM_SC_LEANQ] > 0 && !(status & 23)) {
      // what the lean kernel marked instead of mapping (a character that is not A C G T, a long run of one base, a read beyond 128
      // characters, a wide interval, hits on both strands ...): gathered into a queue and mapped by the general kernel; everything it
      // writes goes where the first launch would have put it, and the reads it sets aside in turn (beyond its slot class) take the
      // long-read pass below
      const int64_t nq = (int64_t)hscal[QM_SC_LEANQ];
      if ((rc = ensure(c->d_slowq, c->capSlowq, nq))) return rc;
      const int g2 = qmk_map_grid_ex(nq, c->numCU, 0);
      unsigned* gs2 = nullptr; int ngs2 = 0;
      if ((rc = gscr_for(c, g2, gs2, ngs2))) return rc;
      HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_SLOWQ, 0, sizeof(u64), c->stream));
      HIPCHK(qmk_collect_lean(c->d_lcnt, nreads, c->d_slowq, (unsigned long long*)(c->d_scal + QM_SC_SLOWQ), c->stream));
      ReadBatch S2 = B;
      S2.slowq = c->d_slowq; S2.nreads = nq; S2.gscratch = c->d_gscr; S2.gslots = gs2; S2.ngslots = ngs2;      // (B was filled in before the scratch existed)
      HIPCHK(hipEventRecord(c->evP0, c->stream));
      HIPCHK(qmk_map_reads(&ix, &S2, ns, g2, c->numCU, c->stream));
      HIPCHK(hipEventRecord(c->evP1, c->stream));
      HIPCHK(hipMemsetAsync(c->d_scal + QM_SC_SLOWQ, 0, sizeof(u64), c->stream));   // (the long-read pass gathers with the same counter)
      HIPCHK(hipMemcpyAsync(hscal, c->d_scal, QM_SC_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      status = (int)(hscal[QM_SC_STATUS] & 0xffffffffu);
      float t = 0; if (hipEventElapsedTime(&t, c->evP0, c->evP1) == hipSuccess) leanExtraMs += t;
      c->lastLeanDeferred = nq;
    }
    if ((!o->sel_aln || rq.mode == QM_RUN_COLLECT) && rq.mode != QM_RUN_FROM_INTERVALS && hscal[QM_SC_SLOWCNT] > 0 && !(status & 23)) {
      // reads longer than the slot class of this launch (always: longer than QM_MAX_READ_LEN) were set aside: gather them
      // and map them with the 32-slot kernels -- a second, small launch; everything it writes (lists, intervals, foundHit)
      // goes where the first pass would have put it
      const int64_t nl_ = (int64_t)hscal[QM_SC_SLOWCNT];
      if ((rc = ensure(c->d_slowq, c->capSlowq, nl_))) return rc;
      HIPCHK(qmk_collect_slow(c->d_lcnt, nreads, c->d_slowq, (unsigned long long*)(c->d_scal + QM_SC_SLOWQ), c->stream));
      ReadBatch S2 = B;
      S2.slowq = c->d_slowq; S2.nreads = nl_;
      const int g2 = qmk_map_grid_ex(nl_, c->numCU, phc);
      if (useLean) { unsigned* gs3 = nullptr; int ngs3 = 0; if ((rc = gscr_for(c, g2 < grid ? g2 : grid, gs3, ngs3))) return rc; S2.gscratch = c->d_gscr; S2.gslots = gs3; S2.ngslots = ngs3; }
      HIPCHK(qmk_map_reads(&ix, &S2, rq.mode == QM_RUN_COLLECT ? -32 : 32, g2 < grid ? g2 : grid, c->numCU, c->stream));
      HIPCHK(hipMemcpyAsync(hscal, c->d_scal, QM_SC_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      status = (int)(hscal[QM_SC_STATUS] & 0xffffffffu);
      c->lastSlowReads = nl_;
    }
    if (o->sel_aln && rq.mode != QM_RUN_COLLECT && hscal[QM_SC_SLOWCNT] > 0 && !(status & 23)) {
      // -s: reads whose SA intervals hold more suffixes than a wave's scratch (repeats, low-complexity reads) were left on
      // the slow queue: gather them, give a few waves scratch sized for the largest, and map them with the same kernel
      const int64_t ns_ = (int64_t)hscal[QM_SC_SLOWCNT];
      const int64_t need = (((int64_t)hscal[QM_SC_SLOWMAX] + 63) / 64) * 64 + 64;
      if ((rc = ensure(c->d_slowq, c->capSlowq, ns_))) return rc;
      HIPCHK(qmk_collect_slow(c->d_lcnt, nreads, c->d_slowq, (unsigned long long*)(c->d_scal + QM_SC_SLOWQ), c->stream));
      const unsigned long long per = (qmk_sel_dyn_bytes(need) + 255) & ~255ULL;
      int64_t waves = ns_ < 256 ? ns_ : 256;
      while (waves > 4 && (unsigned long long)waves * per > (8ULL << 30)) waves /= 2;      // at most 8 GB of scratch
      const int sgrid = (int)((waves + 3) / 4);
      if ((rc = ensure(c->d_dynmem, c->capDynMem, (int64_t)((unsigned long long)sgrid * 4 * per)))) return rc;
      const size_t sb = qmk_sel_dyn_struct_bytes();
      std::vector<unsigned char> hd((size_t)sgrid * 4 * sb);
      for (int w = 0; w < sgrid * 4; ++w) qmk_sel_dyn_bind(hd.data() + (size_t)w * sb, c->d_dynmem + (unsigned long long)w * per, need);
      if ((rc = ensure(c->d_dyn, c->capDyn, (int64_t)hd.size()))) return rc;
      HIPCHK(hipMemcpyAsync(c->d_dyn, hd.data(), hd.size(), hipMemcpyHostToDevice, c->stream));
      ReadBatch S2 = twoPass ? H : B;
      S2.slowq = c->d_slowq; S2.dyn = (SelScratchDyn*)c->d_dyn; S2.nreads = ns_;
      S2.iv_out = nullptr; S2.found_out = nullptr;            // already written by the first pass
      if (twoPass) HIPCHK(qmk_h2m(&ix, &S2, sgrid, c->numCU, c->stream));
      else HIPCHK(launch(S2, sgrid));
      HIPCHK(hipMemcpyAsync(hscal, c->d_scal, QM_SC_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));                 // hd is a local
      status = (int)(hscal[QM_SC_STATUS] & 0xffffffffu);
      c->lastSlowReads = ns_;
    }
#ifdef QM_TIMING
    {
      static const char* nm[7] = {"read->LDS", "strand setup", "probe windows", "extension", "collector rest", "hits->mappings", "write-out+loop"};
      double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)hscal[20 + i];
      for (int i = 0; i < 7; ++i) fprintf(stderr, "[qm timing] %-16s %6.2f %%  %10.0f clk/read\n", nm[i], 100.0 * hscal[20 + i] / tot, (double)hscal[20 + i] / (double)nreads);
      static const char* hm[8] = {"intervals in", "suffix gather", "sort", "groups+chain", "list assembly", "(after)", "write-out", "-"};
      double ht = 0; for (int i = 0; i < 8; ++i) ht += (double)hscal[32 + i];
      if (ht > 0) for (int i = 0; i < 8; ++i) fprintf(stderr, "[qm timing h2m] %-16s %6.2f %%  %10.0f clk/read\n", hm[i], 100.0 * hscal[32 + i] / ht, (double)hscal[32 + i] / (double)nreads);
      static const char* pm[8] = {"cand+intervals", "gather+keys", "rank sort", "heads+count", "chaining", "words", "write-out", "chaining, parallel"};
      double pt = 0; for (int i = 0; i < 8; ++i) pt += (double)hscal[40 + i];
      if (pt > 0) for (int i = 0; i < 8; ++i) fprintf(stderr, "[qm timing pack] %-16s %6.2f %%  %10.0f clk/read\n", pm[i], 100.0 * hscal[40 + i] / pt, (double)hscal[40 + i] / (double)nreads);
    }
#endif
    if (status & 4) return fail(QM_E_TOOLONG, "a read is longer than %d characters (-s: the alignment kernels are sized for that; otherwise the long-read pass takes up to %d)", len_limit(o), QM_MAX_LONG_READ_LEN);
    if (status & 2) return fail(QM_E_UNSUPPORTED, "an SA-interval list exceeded %d entries (max_interval too large)", QM_GCAP);
    if (status & 8) return fail(QM_E_STATE, "selective alignment: a read overflowed the scratch sized for it (internal error)");
    if (status & 64) return fail(QM_E_STATE, "a wavefront found no free scratch slot on its XCD (internal error)");
    if (status & 17) {           // a bump allocator ran out: grow and redo the batch
      if (status & 1) {
        int64_t want = (int64_t)hscal[0] + nreads + (int64_t)grid * 4 * QM_CHUNK;
        if (want < c->capLists * 2) want = c->capLists * 2;
        if ((rc = ensure(c->d_lists, c->capLists, want))) return rc;
      }
      if (status & 16) {
        int64_t want = (int64_t)hscal[QM_SC_IVCUR] + nreads + (int64_t)grid * 4 * QM_IVCHUNK;
        if (want < c->capIv * 2) want = c->capIv * 2;
        if ((rc = ensure(c->d_iv, c->capIv, want))) return rc;
      }
      c->lastRelaunches += 1;
      continue;
    }
    break;
  }
  // reads that were skipped, not mapped (beyond QM_MAX_LONG_READ_LEN characters; interval lists beyond the scratch): their list
  c->lastSkipped = (int64_t)hscal[QM_SC_SKIPCNT]; c->skipList.clear();
  if (c->lastSkipped > 0) {
    c->skipList.resize((size_t)(c->lastSkipped < QM_SKIP_CAP ? c->lastSkipped : QM_SKIP_CAP));
    HIPCHK(hipMemcpy(c->skipList.data(), c->d_skip, c->skipList.size() * sizeof(u64), hipMemcpyDeviceToHost));
  }
  c->lastIvTotal = wantIv ? (int64_t)hscal[QM_SC_IVCUR] : 0;
  c->lastIvReads = wantIv ? nreads : -1;
  c->lastFoundReads = wantFound ? nreads : -1;
  c->lastListReads = (rq.mode != QM_RUN_COLLECT && !duoMerge) ? nreads : -1;   // (pairs the pair kernel merged have no per-read lists)
  c->lastListWords = (int64_t)hscal[0];
  float ms = 0; hipEventElapsedTime(&ms, c->ev0, c->ev1); c->lastMapMs = ms + leanExtraMs;
  return QM_OK;
}

// ---- stage B (+ C with -s): the per-read lists in d_lists -> the units' hits in CSR order.  One thread per unit: count -> scan -> write.
static int run_stage_b(qm_ctx* c, const qm_opts* o, const RunReq& rq, int64_t n, bool paired, const void* d_seq1, const void* d_off1,
                       const void* d_seq2, const void* d_off2, u64* hscal, long long& total) {
  int rc;
  if ((rc = ensure(c->d_cnt, c->capCnt, n + 1))) return rc;
  if ((rc = ensure(c->d_offs, c->capOffs, n + 1))) return rc;
  size_t stb = qmk_scan_temp_bytes(n + 1);
  if (stb > c->scanTmpBytes || !c->d_scanTmp) {
    if (c->d_scanTmp) hipFree(c->d_scanTmp);
    c->d_scanTmp = nullptr; c->scanTmpBytes = 0;
    HIPCHK(hipMalloc(&c->d_scanTmp, stb ? stb : 16));
    c->scanTmpBytes = stb;
  }
  PairBatch P; memset(&P, 0, sizeof(P));
  P.n = n; P.paired = paired ? 1 : 0; P.off1 = (const long long*)d_off1; P.off2 = (const long long*)d_off2;
  P.lcnt = c->d_lcnt; P.loff = c->d_loff; P.lists = c->d_lists; P.cnt = c->d_cnt; P.offs = c->d_offs;
  P.counters = c->d_scal + 1; P.max_num_hits = o->max_num_hits; P.no_orphans = rq.mergeOnly ? 0 : o->no_orphans;
  P.no_dovetail = rq.mergeOnly ? 0 : o->no_dovetail; P.fuzzy = o->fuzzy; P.merge_only = rq.mergeOnly ? 1 : 0;
  if (rq.mergeOnly) {
    if ((rc = ensure(c->d_tooMany, c->capTooMany, n + 1))) return rc;
    HIPCHK(hipMemsetAsync(c->d_tooMany, 0, (size_t)(n + 1), c->stream));
    P.too_many = c->d_tooMany;
  }
  c->lastTooManyUnits = rq.mergeOnly ? n : -1;
  HIPCHK(hipMemsetAsync(c->d_cnt + n, 0, sizeof(uint32_t), c->stream));
  if (o->sel_aln) {
    // -s: merge + selective alignment + filter per unit into temp slots, then compaction (qm_sel.inl)
    HIPCHK(qmk_sel_slots(&P, c->stream));
    HIPCHK(qmk_scan_counts(c->d_scanTmp, c->scanTmpBytes, c->d_cnt, c->d_offs, n + 1, c->stream));
    if ((rc = ensure(c->d_toff, c->capToff, n + 1))) return rc;
    HIPCHK(hipMemcpyAsync(c->d_toff, c->d_offs, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice, c->stream));
    long long slots = 0;
    HIPCHK(hipMemcpyAsync(&slots, c->d_offs + n, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    // chunks of units for plan -> ksw2 -> finish, and where each chunk's slots start (its share of the task buffer)
    const char* cuEnv = getenv("QM_SEL_CHUNK_UNITS");        // (tests: chunk small batches too)
    const int64_t minChunk = cuEnv && atoll(cuEnv) > 0 ? atoll(cuEnv) : 65536;
    const int K = n >= (int64_t)QM_SEL_CHUNKS_B * minChunk ? QM_SEL_CHUNKS_B : 1;
    long long cu[QM_SEL_CHUNKS_B + 1], cslot[QM_SEL_CHUNKS_B + 1];
    for (int i = 0; i <= K; ++i) { cu[i] = n * i / K; cslot[i] = 0; }
    for (int i = 1; i < K; ++i) HIPCHK(hipMemcpyAsync(&cslot[i], c->d_offs + cu[i], sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    cslot[K] = slots;
    if ((rc = ensure(c->d_tmp, c->capTmp, slots + 1))) return rc;
    if ((rc = ensure(c->d_tsc, c->capTsc, 2 * slots + 2))) return rc;
    SelBatch A; memset(&A, 0, sizeof(A));
    A.seq1 = (const unsigned char*)d_seq1; A.seq2 = (const unsigned char*)d_seq2; A.text = c->d_text;
    A.txp_off = c->d_txpOff; A.txp_len = c->d_txpLen; A.tmp = c->d_tmp; A.toff = c->d_toff; A.tsc = c->d_tsc;
    A.match = o->match_score; A.mismatch = o->mismatch_penalty; A.gap_open = o->gap_open; A.gap_extend = o->gap_extend;
    A.long_reads = rq.longReads ? 1 : 0;
    if (rq.longReads && !rq.mergeOnly && sel_ksw_ring_slots(o->dp_bandwidth) > 128) {
      // reads beyond QM_MAX_READ_LEN under a band beyond 97: the alignment blocks of that (slow, rare) edition live in device memory
      if ((rc = ensure(c->d_kswRows, c->capKswRows, (int64_t)qmk_sel_gmem_rows_bytes(c->numCU)))) return rc;
      A.ksw_rows = c->d_kswRows;
    }
    A.short_len = rq.shortLen;
    A.bandwidth = o->dp_bandwidth; A.hard_filter = o->hard_filter; A.policy = o->aln_policy; A.min_score_fraction = o->min_score_fraction;
    { static const int noDiag = [] { const char* e = getenv("QM_SEL_NO_DIAG"); return e ? atoi(e) : 0; }(); A.no_diag = noDiag; }
    HIPCHK(hipMemsetAsync(c->d_cnt + n, 0, sizeof(uint32_t), c->stream));
    if (rq.mergeOnly) {
      HIPCHK(qmk_sel_merge(&P, &A, c->stream));            // the merge alone: no alignment, chain statuses stay in aln_score
    } else {
      // plan (per unit) -> ksw2 extension alignments, four per wavefront, any band -> finish (per unit)
      if ((rc = ensure(c->d_tref, c->capTref, 2 * slots + 2))) return rc;
      if ((rc = ensure(c->d_tasks, c->capTasks, (2 * slots + 2 * K + 2) * (int64_t)qmk_sel_task_bytes()))) return rc;
      if ((rc = ensure(c->d_sides, c->capSides, (2 * slots + 2 * K + 2) * (int64_t)qmk_sel_side_bytes()))) return rc;
      if ((rc = ensure(c->d_torder, c->capTorder, 2 * (2 * slots + 2 * K + 2)))) return rc;      // (two order lists per chunk: ksw2, strip)
      static const bool noStrip = [] { const char* e = getenv("QM_SEL_NO_STRIP"); return e && atoi(e) != 0; }();
      A.tref = c->d_tref;
      HIPCHK(hipMemsetAsync(c->d_ntk, 0, 3 * QM_SEL_CHUNKS_B * sizeof(u64), c->stream));
      // The plan (three kernels: per unit, per alignment question, per question) waits on scattered loads and the ksw2 kernel is
      // bound by VALU issue: the plans of chunks 1.. run on a stream of their own while the alignments of the chunks before
      // them are computed.
      HIPCHK(hipEventRecord(c->evPlan[QM_SEL_CHUNKS_B], c->stream));
      HIPCHK(hipStreamWaitEvent(c->planStream, c->evPlan[QM_SEL_CHUNKS_B], 0));
      SelBatch Ak[QM_SEL_CHUNKS_B];
      static const bool serial = [] { const char* e = getenv("QM_SEL_SERIAL"); return e && atoi(e) != 0; }();   // (profiling: no plan stream, every kernel's duration is its own)
      hipStream_t planStream = K > 1 && !serial ? c->planStream : c->stream;
      for (int i = 0; i < K; ++i) {
        Ak[i] = A; Ak[i].u0 = cu[i]; Ak[i].u1 = cu[i + 1];
        Ak[i].tasks = (SelTask*)(c->d_tasks + (size_t)(2 * cslot[i] + 2 * i) * qmk_sel_task_bytes());   // at most two tasks per slot
        Ak[i].ntasks = c->d_ntk + i;
        Ak[i].sides = (SelSide*)(c->d_sides + (size_t)(2 * cslot[i] + 2 * i) * qmk_sel_side_bytes());       // ... and at most two questions
        Ak[i].nsides = c->d_ntk + QM_SEL_CHUNKS_B + i;
        Ak[i].torder = c->d_torder + (size_t)(2 * cslot[i] + 2 * i);
        if (!noStrip) { Ak[i].torder2 = c->d_torder + (size_t)(2 * slots + 2 * K + 2) + (size_t)(2 * cslot[i] + 2 * i); Ak[i].ntasks2 = c->d_ntk + 2 * QM_SEL_CHUNKS_B + i; }
        HIPCHK(qmk_sel_plan(&P, &Ak[i], c->numCU, planStream));
        if (K > 1) HIPCHK(hipEventRecord(c->evPlan[i], planStream));
      }
      for (int i = 0; i < K; ++i) {
        if (K > 1) HIPCHK(hipStreamWaitEvent(c->stream, c->evPlan[i], 0));
        HIPCHK(qmk_sel_align_finish(&P, &Ak[i], c->numCU, c->stream));
      }
    }
  } else {
    HIPCHK(qmk_pair_count(&P, c->stream));
  }
  HIPCHK(qmk_scan_counts(c->d_scanTmp, c->scanTmpBytes, c->d_cnt, c->d_offs, n + 1, c->stream));
  total = 0;
  HIPCHK(hipMemcpyAsync(&total, c->d_offs + n, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(hscal, c->d_scal, QM_SC_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  u64 h[3 * QM_SEL_CHUNKS_B];
  const bool selStats = o->sel_aln && !rq.mergeOnly;
  if (selStats) HIPCHK(hipMemcpyAsync(h, c->d_ntk, sizeof(h), hipMemcpyDeviceToHost, c->stream));   // (with the synchronisation that follows anyway)
  HIPCHK(hipStreamSynchronize(c->stream));
  c->lastSelQuestions = 0; c->lastKswTasks = 0;
  if (selStats) {
    c->lastStripTasks = 0;
    for (int i = 0; i < QM_SEL_CHUNKS_B; ++i) { c->lastKswTasks += (int64_t)h[i]; c->lastSelQuestions += (int64_t)h[QM_SEL_CHUNKS_B + i]; c->lastStripTasks += (int64_t)h[2 * QM_SEL_CHUNKS_B + i]; }
    static const bool dbg = [] { const char* e = getenv("QM_SEL_DEBUG"); return e && atoi(e) != 0; }();
    if (dbg) fprintf(stderr, "[qm -s] %lld units: %lld alignment questions beyond PERFECT chains, %lld ksw2 alignments, %lld strip alignments\n", (long long)n, (long long)c->lastSelQuestions, (long long)c->lastKswTasks, (long long)c->lastStripTasks);
  }
  if (rq.join) {
    long long b = 0; qm_hit* dst = nullptr;
    if ((rc = rq.join->arrive(rq.part, total, b, dst))) return rc;
    P.hits = dst + b;
  } else {
    if ((rc = ensure(c->d_hits, c->capHits, (int64_t)total + 1, total / 8))) return rc;
    P.hits = c->d_hits;
  }
  if (o->sel_aln) HIPCHK(qmk_sel_compact(&P, c->d_tmp, c->d_toff, c->stream));
  else HIPCHK(qmk_pair_write(&P, c->stream));
  return QM_OK;
}

static int map_device_impl(qm_ctx* c, const qm_opts* o, int64_t n, const void* d_seq1, const void* d_off1, const void* d_seq2,
                           const void* d_off2, int32_t max_read_len, int64_t* n_hits, qm_counters* counters, ChunkFeeder* feeder,
                           const RunReq& rq = RunReq(), int32_t short_read_len = 0) {
  if (!c || n < 0 || (n > 0 && (!d_seq1 || !d_off1))) return fail(QM_E_ARG, "bad argument");
  int rc = check_opts(o);
  if (rc) return rc;
  if ((d_seq2 == nullptr) != (d_off2 == nullptr)) return fail(QM_E_ARG, "seq2/off2 must both be given or both be null");
  // (a read beyond len_limit() is skipped, not mapped: qm_fetch_skipped)
  HIPCHK(hipSetDevice(c->device));
  // 64-character slots per read: picks the kernel instantiation.  `short_read_len` (host callers: the longest read that is not
  // beyond QM_MAX_READ_LEN) picks it when the batch also holds long reads -- those are set aside by the launch whatever its
  // slot class and mapped by the long-read pass
  const int pick = (max_read_len > QM_MAX_READ_LEN && short_read_len > 0) ? short_read_len : (max_read_len > QM_MAX_READ_LEN ? QM_MAX_READ_LEN : max_read_len);
  const int ns = pick <= 128 ? 2 : (pick <= 192 ? 3 : (pick <= 256 ? 4 : 8));
  const bool paired = d_seq2 != nullptr;
  u64 hscal[QM_SC_WORDS];
  c->lastUnits = -1;
  HIPCHK(hipEventRecord(c->evA, c->stream));
  RunReq r2 = rq;
  r2.keepIntervals = rq.keepIntervals || c->debug != 0;
  r2.longReads = max_read_len > QM_MAX_READ_LEN;
  r2.shortLen = pick;
  if (rq.join && (rc = rq.join->wait_turn(rq.part))) return rc;
  rc = run_stage_a(c, o, r2, n, d_seq1, d_off1, d_seq2, d_off2, ns, feeder, hscal);
  if (rq.join) rq.join->pass_turn(rq.part);
  if (rc) return rc;
  c->stReads = -1; c->stUnits = -1;
  if (r2.stageView) {
    // where every read's interval records and list words will sit in CSR order: two scans behind stage A; their totals come
    // down with the synchronisation stage B needs anyway
    const int64_t nreads = paired ? 2 * n : n;
    if ((rc = ensure(c->d_ivcsr, c->capIvcsr, nreads + 1))) return rc;
    if ((rc = ensure(c->d_lcsr, c->capLcsr, nreads + 1))) return rc;
    if (!c->h_tot) HIPCHK(hipHostMalloc((void**)&c->h_tot, 2 * sizeof(long long), hipHostMallocDefault));
    const size_t stb = qmk_scan_temp_bytes(nreads + 1);
    if (stb > c->scanTmpBytes || !c->d_scanTmp) {
      if (c->d_scanTmp) hipFree(c->d_scanTmp);
      c->d_scanTmp = nullptr; c->scanTmpBytes = 0;
      HIPCHK(hipMalloc(&c->d_scanTmp, stb ? stb : 16));
      c->scanTmpBytes = stb;
    }
    if (!r2.keepIntervals) {
      // a stage view without the SA-interval records (QM_STAGES_NO_INTERVALS): every read's count is zero
      if ((rc = ensure(c->d_ivcnt, c->capIvCnt, nreads + 1))) return rc;
      if ((rc = ensure(c->d_ivoff, c->capIvOff, nreads + 1))) return rc;
      if ((rc = ensure(c->d_iv, c->capIv, 1))) return rc;
      HIPCHK(hipMemsetAsync(c->d_ivcnt, 0, (size_t)(nreads + 1) * sizeof(uint32_t), c->stream));
      HIPCHK(hipMemsetAsync(c->d_ivoff, 0, (size_t)(nreads + 1) * sizeof(long long), c->stream));
    }
    HIPCHK(hipMemsetAsync(c->d_ivcnt + nreads, 0, sizeof(uint32_t), c->stream));
    HIPCHK(hipMemsetAsync(c->d_lcnt + nreads, 0, sizeof(uint32_t), c->stream));
    HIPCHK(qmk_scan_counts(c->d_scanTmp, c->scanTmpBytes, c->d_ivcnt, c->d_ivcsr, nreads + 1, c->stream));
    HIPCHK(qmk_scan_counts_masked(c->d_scanTmp, c->scanTmpBytes, c->d_lcnt, c->d_lcsr, nreads + 1, c->stream));
    HIPCHK(hipMemcpyAsync(&c->h_tot[0], c->d_ivcsr + nreads, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&c->h_tot[1], c->d_lcsr + nreads, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    c->stReads = nreads;
  }
  long long total = 0;
  if ((rc = run_stage_b(c, o, r2, n, paired, d_seq1, d_off1, d_seq2, d_off2, hscal, total))) return rc;
  HIPCHK(hipEventRecord(c->evB, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  float ms = 0; hipEventElapsedTime(&ms, c->evA, c->evB); c->lastTotalMs = ms;
  c->lastUnits = n; c->lastHits = total; c->lastPaired = paired;
  if (r2.stageView) c->stUnits = n;
  if (n_hits) *n_hits = total;
  if (counters) {
    counters->pe_hits = hscal[1]; counters->se_hits = hscal[2]; counters->tot_hits = hscal[3];
    counters->num_reads = hscal[4]; counters->too_many_hits = hscal[5]; counters->mapped = hscal[6];
  }
  return QM_OK;
}

int SplitJoin::arrive(int part, long long tot, long long& b, qm_hit*& dst) {
  std::unique_lock<std::mutex> lk(mu);
  total[part] = tot;
  if (++arrived == K && !failed) {
    for (int i = 0; i < K; ++i) base[i + 1] = base[i] + total[i];
    const int rc = ensure(owner->d_hits, owner->capHits, (int64_t)base[K] + 1, base[K] / 8);
    if (rc) failed = rc; else ready = true;
    cv.notify_all();
  }
  cv.wait(lk, [&] { return ready || failed; });
  if (failed) return failed == QM_E_NOMEM ? fail(QM_E_NOMEM, "hipMalloc of the result array failed") : fail(failed, "another part of the batch failed");
  b = base[part]; dst = owner->d_hits;
  return QM_OK;
}

// A large device-resident batch in K parts, each on a helper context of the same replica (own stream, own work buffers),
// in flight together: the tail of one part's persistent stage-A grid, its scans, its host round trips and (with -s) its
// VALU-bound alignment kernels run under the other parts' stage A, which is bound by the scalar unit and by waiting
// (profiles/r04/sel_overlap.txt).  Units, hits and counters are those of one call: part i maps units [n i / K, n (i + 1) / K),
// its hits land in the caller's array behind those of the parts before it, its offsets are rebased into the caller's.
static int map_device_split(qm_ctx* c, const qm_opts* o, int K, int64_t n, const void* d_seq1, const void* d_off1, const void* d_seq2,
                            const void* d_off2, int32_t max_read_len, int64_t* n_hits, qm_counters* counters) {
  int rc;
  HIPCHK(hipSetDevice(c->device));
  while ((int)c->helpers.size() < K) {
    qm_ctx* h = nullptr;
    if ((rc = qm_ctx_create_ex(c->ix, c->device, c->flags, &h))) return rc;
    h->isHelper = true; c->helpers.push_back(h);
  }
  if (!c->pool || (int)c->pool->th.size() < K) { delete c->pool; c->pool = new SplitPool(K); }
  if ((rc = ensure(c->d_offs, c->capOffs, n + 1))) return rc;
  SplitJoin J; J.owner = c; J.K = K;
  { const char* sg = getenv("QM_SPLIT_STAGGER"); J.stagger = sg ? atoi(sg) != 0 : false; }   // (measured: 73.6-75.4 against 76.4 M pairs/s with all parts started at once -- the kernels are all bound by instruction issue, whichever unit)
  int64_t nh[8] = {0}; qm_counters ctr[8]; int rcs[8] = {0}; char errs[8][256];
  memset(ctr, 0, sizeof(ctr));
  const bool paired = d_seq2 != nullptr;
  static const int firstPct = [] { const char* e = getenv("QM_SPLIT_FIRST"); const int v = e ? atoi(e) : 0; return v > 0 && v < 100 ? v : 0; }();   // (tuning knob: two parts of unequal size)
  // Which parts take the pair kernel (a bit per part; QM_DUO_PARTS): the others take qm_lean_kernel.  The pair kernel is bound by the
  // vector units (919 VALU + 470 scalar-side instructions per pair), qm_lean_kernel by the CU's scalar unit (756 + 1 125): with a part
  // of each kind in flight the two kinds of wavefront share every CU and each loads the unit the other leaves idle -- 492 M pairs/s
  // against 447 (all parts the pair kernel) and 473 (all parts qm_lean_kernel), profiles/r06/exp_mix.txt
  static const int duoParts = [] { const char* e = getenv("QM_DUO_PARTS"); return e ? atoi(e) : 0x55; }();   // default: every other part
  c->lastUnits = -1;
  HIPCHK(hipEventRecord(c->evA, c->stream));
  c->pool->run(K, [&](int i) {
    qm_ctx* h = c->helpers[(size_t)i];
    int64_t u0 = n * i / K, u1 = n * (i + 1) / K;
    if (K == 2 && firstPct > 0) { const int64_t cut = n * firstPct / 100; u0 = i == 0 ? 0 : cut; u1 = i == 0 ? cut : n; }
    RunReq rq; rq.join = &J; rq.part = i;
    rq.noDuo = ((duoParts >> i) & 1) == 0;
    errs[i][0] = 0;
    int r = map_device_impl(h, o, u1 - u0, d_seq1, (const long long*)d_off1 + u0, d_seq2, d_seq2 ? (const long long*)d_off2 + u0 : nullptr,
                            max_read_len, &nh[i], &ctr[i], nullptr, rq);
    if (r == QM_OK) {
      // this part's offsets into the caller's array (the last part also writes the closing one)
      hipError_t e = qmk_rebase_offsets(h->d_offs, c->d_offs + u0, (u1 - u0) + (i == K - 1 ? 1 : 0), J.base[i], h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) r = fail(QM_E_NOGPU, "rebasing a part's offsets: %s", hipGetErrorString(e));
    }
    if (r != QM_OK) { snprintf(errs[i], sizeof(errs[i]), "%s", g_err); J.abort(r); }
    rcs[i] = r;
  });
  for (int i = 0; i < K; ++i) if (rcs[i]) return fail(rcs[i], "%s", errs[i]);
  HIPCHK(hipEventRecord(c->evB, c->stream));
  HIPCHK(hipEventSynchronize(c->evB));
  // stage A's span: first start to last end over the parts (events of different streams of one device compare)
  float first = 1e30f, last = 0, t = 0;
  for (int i = 0; i < K; ++i) {
    qm_ctx* h = c->helpers[(size_t)i];
    if (hipEventElapsedTime(&t, c->evA, h->ev0) == hipSuccess && t < first) first = t;
    if (hipEventElapsedTime(&t, c->evA, h->ev1) == hipSuccess && t > last) last = t;
  }
  c->lastMapMs = last > first ? last - first : 0;
  float ms = 0; hipEventElapsedTime(&ms, c->evA, c->evB); c->lastTotalMs = ms;
  c->lastRelaunches = 0; c->lastSlowReads = 0; c->lastSkipped = 0; c->skipList.clear(); c->lastSelQuestions = 0; c->lastKswTasks = 0; c->lastStripTasks = 0;
  c->lastDuoPairs = -1; c->lastDuoMerged = 0; c->lastLeanReads = -1; c->lastLeanDeferred = 0;
  for (int i = 0; i < 4; ++i) c->lastDefer[i] = 0;
  c->lastNPass = 0;
  qm_counters sum; memset(&sum, 0, sizeof(sum));
  for (int i = 0; i < K; ++i) {
    sum.pe_hits += ctr[i].pe_hits; sum.se_hits += ctr[i].se_hits; sum.tot_hits += ctr[i].tot_hits; sum.num_reads += ctr[i].num_reads;
    sum.too_many_hits += ctr[i].too_many_hits; sum.mapped += ctr[i].mapped;
    qm_ctx* h = c->helpers[(size_t)i];
    if (h->lastLeanReads >= 0) { c->lastLeanReads = (c->lastLeanReads < 0 ? 0 : c->lastLeanReads) + h->lastLeanReads; c->lastLeanDeferred += h->lastLeanDeferred; }
    if (h->lastDuoPairs >= 0) { c->lastDuoMerged += h->lastDuoMerged; }
    for (int t = 0; t < 4; ++t) c->lastDefer[t] += h->lastDefer[t];
    c->lastNPass += h->lastNPass;
    c->lastRelaunches += h->lastRelaunches; c->lastSlowReads += h->lastSlowReads; c->lastSelQuestions += h->lastSelQuestions; c->lastKswTasks += h->lastKswTasks; c->lastStripTasks += h->lastStripTasks;
    // the part's skipped reads, as reads of the whole batch
    int64_t u0 = n * i / K;
    if (K == 2 && firstPct > 0) u0 = i == 0 ? 0 : n * firstPct / 100;
    c->lastSkipped += h->lastSkipped;
    for (uint64_t e : h->skipList)
      if (c->skipList.size() < QM_SKIP_CAP) c->skipList.push_back(((e & ((1ULL << 56) - 1)) + (uint64_t)(paired ? 2 * u0 : u0)) | (e & ~((1ULL << 56) - 1)));
  }
  c->lastUnits = n; c->lastHits = J.base[K]; c->lastPaired = paired;
  c->lastIvReads = -1; c->lastFoundReads = -1; c->lastListReads = -1; c->lastTooManyUnits = -1; c->stReads = -1; c->stUnits = -1;
  if (n_hits) *n_hits = J.base[K];
  if (counters) *counters = sum;
  return QM_OK;
}

int qm_map_device(qm_ctx* c, const qm_opts* o, int64_t n, const void* d_seq1, const void* d_off1, const void* d_seq2,
                  const void* d_off2, int32_t max_read_len, int64_t* n_hits, qm_counters* counters) {
  // Two parts in flight together for batches of 2 M units and more (QM_SPLIT: how many; 1: none).  With -s the parts' alignment kernels
  // run under each other's stage A.  Without -s a call was stage A and little else until round 5; behind the lean kernel the pair
  // kernels, scans and synchronisations of stage B are a tenth of the step: with two parts started together (their stage-A launches
  // share the chip, each about as long as one launch over the whole batch) whichever part is ahead runs its stage B under the other's
  // stage A (434.8 -> 456.7 M pairs/s on config 2).
  const char* me = getenv("QM_SPLIT_MIN");                 // (tests: split small batches too)
  const int64_t minUnits = me && atoll(me) > 0 ? atoll(me) : ((int64_t)1 << 21);
  if (c && o && !c->isHelper && !c->debug && n >= minUnits && d_seq1 && d_off1 && (d_seq2 == nullptr) == (d_off2 == nullptr) && check_opts(o) == QM_OK) {
    const char* se = getenv("QM_SPLIT");
    int K = se ? atoi(se) : 2;
    if (K > 8) K = 8;
    if (K > 1) return map_device_split(c, o, K, n, d_seq1, d_off1, d_seq2, d_off2, max_read_len, n_hits, counters);
  }
  return map_device_impl(c, o, n, d_seq1, d_off1, d_seq2, d_off2, max_read_len, n_hits, counters, nullptr);
}

// offsets of one mate: monotone, longest read; device copies of the offsets (copy stream) and room for the characters
static int stage_offsets(qm_ctx* c, int64_t n, const int64_t* off, uint8_t*& d_seq, int64_t& capSeq, long long*& d_off, int64_t& capOff,
                         int32_t& maxLen, int32_t& maxShort) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t l = off[i + 1] - off[i];
    if (l < 0) return fail(QM_E_ARG, "offsets not monotone");
    if (l > maxLen) maxLen = (int32_t)(l > 0x7fffffff ? 0x7fffffff : l);
    if (l <= QM_MAX_READ_LEN && l > maxShort) maxShort = (int32_t)l;
  }
  int rc;
  if ((rc = ensure(d_seq, capSeq, off[n] + 64))) return rc;
  if ((rc = ensure(d_off, capOff, n + 1))) return rc;
  HIPCHK(hipMemcpyAsync(d_off, off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->copyStream));
  return QM_OK;
}

struct HostFeed { qm_ctx* c; const char* seq1; const int64_t* off1; const char* seq2; const int64_t* off2; };
static int host_feed_upload(void* self, int64_t u0, int64_t u1) {
  HostFeed* f = (HostFeed*)self;
  const int64_t a1 = f->off1[u0], b1 = f->off1[u1];
  // (pageable memory through the runtime's own staging: 35 GB/s on the round's boxes -- eight threads copying into pinned buffers of our own
  // were no faster, profiles/r06/exp_mix.txt)
  if (b1 > a1) HIPCHK(hipMemcpyAsync(f->c->d_seq1 + a1, f->seq1 + a1, (size_t)(b1 - a1), hipMemcpyHostToDevice, f->c->copyStream));
  if (f->seq2) {
    const int64_t a2 = f->off2[u0], b2 = f->off2[u1];
    if (b2 > a2) HIPCHK(hipMemcpyAsync(f->c->d_seq2 + a2, f->seq2 + a2, (size_t)(b2 - a2), hipMemcpyHostToDevice, f->c->copyStream));
  }
  return QM_OK;
}

static int map_host(qm_ctx* c, const qm_opts* o, int64_t n, const char* seq1, const int64_t* off1, const char* seq2,
                    const int64_t* off2, int64_t* n_hits, qm_counters* counters, const RunReq& rq) {
  if (!c || n < 0 || (n > 0 && (!seq1 || !off1))) return fail(QM_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  int32_t maxLen = 0, maxShort = 0; int rc;
  static const int64_t zero = 0;
  if (n == 0) { off1 = &zero; if (seq2) off2 = &zero; }
  if ((rc = check_opts(o))) return rc;
  if ((rc = stage_offsets(c, n, off1, c->d_seq1, c->capSeq1, c->d_off1, c->capOff1, maxLen, maxShort))) return rc;
  if (seq2 && (rc = stage_offsets(c, n, off2, c->d_seq2, c->capSeq2, c->d_off2, c->capOff2, maxLen, maxShort))) return rc;
  const int32_t lim = len_limit(o);
  // the characters follow chunk by chunk, each chunk's kernel behind its own copy (ChunkFeeder); QM_HOST_CHUNK = units per chunk
  HostFeed hf = {c, seq1, off1, seq2, off2};
  const char* ce = getenv("QM_HOST_CHUNK");
  ChunkFeeder fd; fd.chunk = ce && atoll(ce) > 0 ? atoll(ce) : (1 << 20); fd.upload = host_feed_upload; fd.self = &hf;
  rc = map_device_impl(c, o, n, c->d_seq1, c->d_off1, seq2 ? c->d_seq2 : nullptr, seq2 ? c->d_off2 : nullptr, maxLen, n_hits, counters, &fd, rq, maxShort > 0 ? maxShort : 1);
  hipStreamSynchronize(c->copyStream);                     // nothing of the caller's buffers is in flight after return (error paths too)
  return rc;
}

int qm_map_pairs(qm_ctx* c, const qm_opts* o, int64_t n, const char* seq1, const int64_t* off1, const char* seq2,
                 const int64_t* off2, int64_t* n_hits, qm_counters* counters) {
  if (!seq2 || !off2) return fail(QM_E_ARG, "qm_map_pairs needs both mates");
  return map_host(c, o, n, seq1, off1, seq2, off2, n_hits, counters, RunReq());
}

int qm_map_reads(qm_ctx* c, const qm_opts* o, int64_t n, const char* seq, const int64_t* off, int64_t* n_hits,
                 qm_counters* counters) {
  return map_host(c, o, n, seq, off, nullptr, nullptr, n_hits, counters, RunReq());
}

// 2-bit packed reads in, the ASCII image the kernels read built on the device (include/qmap_mi355.h)
static int unpack_mate(qm_ctx* c, int64_t n, const uint8_t* pk, const int64_t* off, const qm_pack_exc* exc, int64_t nexc, uint8_t*& d_seq, int64_t& capSeq,
                       long long*& d_off, int64_t& capOff, uint8_t*& d_pk, int64_t& capPk, qm_pack_exc*& d_exc, int64_t& capExc, int32_t& maxLen, int32_t& maxShort) {
  int32_t mateMax = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t l = off[i + 1] - off[i];
    if (l < 0) return fail(QM_E_ARG, "offsets not monotone");
    if (l > mateMax) mateMax = (int32_t)(l > 0x7fffffff ? 0x7fffffff : l);
    if (l <= QM_MAX_READ_LEN && l > maxShort) maxShort = (int32_t)l;
  }
  if (mateMax > maxLen) maxLen = mateMax;
  if (off[n] > 0xffffffffLL) return fail(QM_E_ARG, "more than 2^32 characters in one packed batch");
  int rc;
  const int64_t pkBytes = qm_packed_bytes(off, n);
  if ((rc = ensure(d_seq, capSeq, off[n] + 64))) return rc;
  if ((rc = ensure(d_off, capOff, n + 1))) return rc;
  if ((rc = ensure(d_pk, capPk, pkBytes))) return rc;
  if ((rc = ensure(d_exc, capExc, nexc + 1))) return rc;
  HIPCHK(hipMemcpyAsync(d_off, off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_pk, pk, (size_t)pkBytes - 8, hipMemcpyHostToDevice, c->stream));
  if (nexc > 0) HIPCHK(hipMemcpyAsync(d_exc, exc, (size_t)nexc * sizeof(qm_pack_exc), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(d_seq + off[n], 0, 64, c->stream));      // the mapper fetches reads a word at a time: defined bytes behind the last one
  HIPCHK(qmk_unpack_reads(d_pk, d_off, n, off[n], d_seq, d_exc, nexc, c->stream));
  return QM_OK;
}

static int map_packed(qm_ctx* c, const qm_opts* o, int64_t n, const uint8_t* pk1, const int64_t* off1, const qm_pack_exc* exc1, int64_t nexc1,
                      const uint8_t* pk2, const int64_t* off2, const qm_pack_exc* exc2, int64_t nexc2, int64_t* n_hits, qm_counters* counters, const RunReq& rq = RunReq()) {
  if (!c || n < 0 || (n > 0 && (!pk1 || !off1)) || nexc1 < 0 || nexc2 < 0 || (nexc1 > 0 && !exc1) || (nexc2 > 0 && !exc2)) return fail(QM_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  int rc;
  if ((rc = check_opts(o))) return rc;
  static const int64_t zero[2] = {0, 0};
  static const uint8_t none[8] = {0};
  if (n == 0) { off1 = zero; pk1 = none; if (pk2) { off2 = zero; pk2 = none; } }
  int32_t maxLen = 0, maxShort = 0;
  if ((rc = unpack_mate(c, n, pk1, off1, exc1, nexc1, c->d_seq1, c->capSeq1, c->d_off1, c->capOff1, c->d_pk1, c->capPk1, c->d_exc1, c->capExc1, maxLen, maxShort))) return rc;
  if (pk2 && (rc = unpack_mate(c, n, pk2, off2, exc2, nexc2, c->d_seq2, c->capSeq2, c->d_off2, c->capOff2, c->d_pk2, c->capPk2, c->d_exc2, c->capExc2, maxLen, maxShort))) return rc;
  rc = map_device_impl(c, o, n, c->d_seq1, c->d_off1, pk2 ? c->d_seq2 : nullptr, pk2 ? c->d_off2 : nullptr, maxLen, n_hits, counters, nullptr, rq, maxShort > 0 ? maxShort : 1);
  hipStreamSynchronize(c->stream);                         // nothing of the caller's buffers is in flight after return (error paths too)
  return rc;
}

int qm_map_pairs_packed(qm_ctx* c, const qm_opts* o, int64_t n, const uint8_t* pk1, const int64_t* off1, const qm_pack_exc* exc1, int64_t nexc1,
                        const uint8_t* pk2, const int64_t* off2, const qm_pack_exc* exc2, int64_t nexc2, int64_t* n_hits, qm_counters* counters) {
  if (!pk2 || !off2) return fail(QM_E_ARG, "qm_map_pairs_packed needs both mates");
  return map_packed(c, o, n, pk1, off1, exc1, nexc1, pk2, off2, exc2, nexc2, n_hits, counters);
}
int qm_map_reads_packed(qm_ctx* c, const qm_opts* o, int64_t n, const uint8_t* pk, const int64_t* off, const qm_pack_exc* exc, int64_t nexc,
                        int64_t* n_hits, qm_counters* counters) {
  return map_packed(c, o, n, pk, off, exc, nexc, nullptr, nullptr, nullptr, 0, n_hits, counters);
}

// device -> the caller's (pageable) memory, one array
static int staged_download(qm_ctx* c, void* dstv, const void* d_src, size_t bytes) {
    if (bytes < ((size_t)64 << 20)) HIPCHK(hipMemcpy(dstv, d_src, bytes, hipMemcpyDeviceToHost));
    else {
      unsigned char* const hits = (unsigned char*)dstv;
      // (a result array the caller has just allocated is a million page faults: ask for huge pages where the range holds whole ones --
      // a hint, honoured where transparent huge pages are on `madvise` or `always`)
      {
        const uintptr_t a = ((uintptr_t)hits + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1), e = ((uintptr_t)hits + bytes) & ~(((uintptr_t)2 << 20) - 1);
        static const bool thp = [] { const char* v = getenv("QM_FETCH_THP"); return !(v && atoi(v) == 0); }();
        if (thp && e > a) madvise((void*)a, (size_t)(e - a), MADV_HUGEPAGE);
      }
      // A large result lands in memory the caller has usually just allocated (every page still to be faulted in), where a
      // plain pageable hipMemcpy runs at ~9 GB/s.  Instead: DMA into two pinned staging buffers in turn at PCIe rate while
      // host threads copy the previous chunk into the caller's array, so the page faults are spread over several cores.
      const size_t CH = (size_t)32 << 20;
      if (!c->h_stage) HIPCHK(hipHostMalloc((void**)&c->h_stage, 2 * CH, hipHostMallocDefault));
      if (!c->evStage[0]) { HIPCHK(hipEventCreateWithFlags(&c->evStage[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->evStage[1], hipEventDisableTiming)); }
      const size_t nch = (bytes + CH - 1) / CH;
      auto drain = [&](size_t i) -> hipError_t {             // chunk i: staging -> caller's array, 8 threads
        hipError_t e = hipEventSynchronize(c->evStage[i & 1]);
        if (e != hipSuccess) return e;
        const size_t off = i * CH, len = off + CH <= bytes ? CH : bytes - off;
        const unsigned char* src = c->h_stage + (i & 1) * CH; unsigned char* dst = hits + off;
        const int nt = 8; std::vector<std::thread> th;
        const size_t per = ((len + nt - 1) / nt + 4095) & ~(size_t)4095;
        for (int t = 0; t < nt; ++t) {
          const size_t a0 = per * (size_t)t; if (a0 >= len) break;
          const size_t l0 = a0 + per <= len ? per : len - a0;
          th.emplace_back([=]() { memcpy(dst + a0, src + a0, l0); });
        }
        for (auto& x : th) x.join();
        return hipSuccess;
      };
      for (size_t i = 0; i < nch; ++i) {
        const size_t off = i * CH, len = off + CH <= bytes ? CH : bytes - off;
        HIPCHK(hipMemcpyAsync(c->h_stage + (i & 1) * CH, (const unsigned char*)d_src + off, len, hipMemcpyDeviceToHost, c->copyStream));
        HIPCHK(hipEventRecord(c->evStage[i & 1], c->copyStream));
        if (i > 0) HIPCHK(drain(i - 1));
      }
      HIPCHK(drain(nch - 1));
    }
  return QM_OK;
}

int qm_fetch_hits(qm_ctx* c, int64_t* hit_offsets, qm_hit* hits) {
  if (!c || c->lastUnits < 0) return fail(QM_E_STATE, "no mapping result to fetch");
  HIPCHK(hipSetDevice(c->device));
  int rc;
  if (hit_offsets && (rc = staged_download(c, hit_offsets, c->d_offs, (size_t)(c->lastUnits + 1) * 8))) return rc;
  if (hits && c->lastHits > 0 && (rc = staged_download(c, hits, c->d_hits, (size_t)c->lastHits * sizeof(qm_hit)))) return rc;
  return QM_OK;
}

int qm_fetch_hits_pinned(qm_ctx* c, int64_t* hit_offsets, qm_hit* hits) {
  if (!c || c->lastUnits < 0) return fail(QM_E_STATE, "no mapping result to fetch");
  HIPCHK(hipSetDevice(c->device));
  if (hit_offsets) HIPCHK(hipMemcpyAsync(hit_offsets, c->d_offs, (size_t)(c->lastUnits + 1) * 8, hipMemcpyDeviceToHost, c->copyStream));
  if (hits && c->lastHits > 0) HIPCHK(hipMemcpyAsync(hits, c->d_hits, (size_t)c->lastHits * sizeof(qm_hit), hipMemcpyDeviceToHost, c->copyStream));
  HIPCHK(hipStreamSynchronize(c->copyStream));
  return QM_OK;
}

int qm_result_device(qm_ctx* c, const void** d_hit_offsets, const void** d_hits) {
  if (!c || c->lastUnits < 0) return fail(QM_E_STATE, "no mapping result");
  if (d_hit_offsets) *d_hit_offsets = c->d_offs;
  if (d_hits) *d_hits = c->d_hits;
  return QM_OK;
}

int qm_fetch_intervals(qm_ctx* c, int64_t* int_offsets, qm_sa_interval_hit* ints, int64_t cap) {
  if (!c || c->lastIvReads < 0) return fail(QM_E_STATE, "no SA-interval hits kept (qm_ctx_set_debug / qm_collect_reads / qm_map_pairs_stages before mapping)");
  if (!int_offsets) return fail(QM_E_ARG, "null int_offsets");
  HIPCHK(hipSetDevice(c->device));
  // per unit: a pair's four lists (left fwd, left rc, right fwd, right rc) follow each other; a single read's two
  const int64_t nreads = c->lastIvReads;
  const int mates = (c->lastUnits >= 0 && c->lastPaired && c->lastIvReads == 2 * c->lastUnits) ? 2 : 1;
  const int64_t n = nreads / mates;
  std::vector<uint32_t> cnt((size_t)nreads + 1); std::vector<long long> off((size_t)nreads + 1);
  if (nreads) {
    HIPCHK(hipMemcpy(cnt.data(), c->d_ivcnt, (size_t)nreads * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(off.data(), c->d_ivoff, (size_t)nreads * 8, hipMemcpyDeviceToHost));
  }
  int_offsets[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t t = 0;
    for (int m = 0; m < mates; ++m) t += cnt[(size_t)(i * mates + m)];
    int_offsets[i + 1] = int_offsets[i] + t;
  }
  if (!ints) return QM_OK;
  if (cap < int_offsets[n]) return fail(QM_E_ARG, "interval buffer too small");
  // the cursor of the chunked allocator may stand up to one chunk behind the buffer's end (a chunk is reserved whole, the
  // overflow check is per read): only what lies inside the allocation is copied -- every recorded interval does
  const int64_t used = c->lastIvTotal < c->capIv ? c->lastIvTotal : c->capIv;
  std::vector<qm_sa_interval_hit> all((size_t)used + 1);
  if (used) HIPCHK(hipMemcpy(all.data(), c->d_iv, (size_t)used * sizeof(qm_sa_interval_hit), hipMemcpyDeviceToHost));
  int64_t w = 0;
  for (int64_t r = 0; r < nreads; ++r)
    for (uint32_t j = 0; j < cnt[(size_t)r]; ++j) {
      if (off[(size_t)r] + (long long)j >= used) return fail(QM_E_STATE, "interval list of read %lld lies outside the buffer (internal error)", (long long)r);
      ints[w++] = all[(size_t)(off[(size_t)r] + j)];
    }
  return QM_OK;
}

// ---- the reference's three entry points as calls of their own (include/qmap_rapmap_compat.hpp sits on these) ----------------

static int upload(qm_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  return QM_OK;
}

static int map_host(qm_ctx* c, const qm_opts* o, int64_t n, const char* seq1, const int64_t* off1, const char* seq2,
                    const int64_t* off2, int64_t* n_hits, qm_counters* counters, const RunReq& rq);

int qm_collect_reads(qm_ctx* c, const qm_opts* o, int64_t n, const char* seq, const int64_t* off, int64_t* n_intervals) {
  if (!c || n < 0 || (n > 0 && (!seq || !off))) return fail(QM_E_ARG, "bad argument");
  int rc = check_opts(o);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  static const int64_t zero = 0;
  if (n == 0) off = &zero;
  int32_t maxLen = 0;
  for (int64_t i = 0; i < n; ++i) { const int64_t l = off[i + 1] - off[i]; if (l < 0) return fail(QM_E_ARG, "offsets not monotone"); if (l > maxLen) maxLen = (int32_t)(l > 0x7fffffff ? 0x7fffffff : l); }
  if (maxLen > (len_limit(o))) return fail(QM_E_TOOLONG, "read length %d > %d", maxLen, len_limit(o));
  if ((rc = ensure(c->d_seq1, c->capSeq1, off[n] + 64))) return rc;
  if ((rc = ensure(c->d_off1, c->capOff1, n + 1))) return rc;
  if ((rc = upload(c, c->d_off1, off, (size_t)(n + 1) * 8))) return rc;
  if ((rc = upload(c, c->d_seq1, seq, (size_t)off[n]))) return rc;
  RunReq rq; rq.mode = QM_RUN_COLLECT;
  u64 hscal[QM_SC_WORDS];
  c->lastUnits = -1;
  if ((rc = run_stage_a(c, o, rq, n, c->d_seq1, c->d_off1, nullptr, nullptr, 8, nullptr, hscal))) return rc;
  if (n_intervals) {
    std::vector<uint32_t> cnt((size_t)n + 1);
    if (n) HIPCHK(hipMemcpy(cnt.data(), c->d_ivcnt, (size_t)n * 4, hipMemcpyDeviceToHost));
    int64_t t = 0; for (int64_t i = 0; i < n; ++i) t += cnt[(size_t)i];
    *n_intervals = t;
  }
  return QM_OK;
}

int qm_fetch_found(qm_ctx* c, uint8_t* found) {
  if (!c || c->lastFoundReads < 0) return fail(QM_E_STATE, "no foundHit flags kept by the last call");
  if (!found) return fail(QM_E_ARG, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  if (c->lastFoundReads) HIPCHK(hipMemcpy(found, c->d_found, (size_t)c->lastFoundReads, hipMemcpyDeviceToHost));
  return QM_OK;
}

int qm_hits_to_mappings(qm_ctx* c, const qm_opts* o, int64_t n, const int32_t* read_len, const int64_t* int_offsets,
                        const qm_sa_interval_hit* ints, int64_t* n_words) {
  if (!c || n < 0 || (n > 0 && (!read_len || !int_offsets))) return fail(QM_E_ARG, "bad argument");
  int rc = check_opts(o);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  static const int64_t zero = 0;
  if (n == 0) int_offsets = &zero;
  const int64_t ni = int_offsets[n];
  if (ni > 0 && !ints) return fail(QM_E_ARG, "null intervals");
  for (int64_t i = 0; i < n; ++i) {
    if (int_offsets[i + 1] < int_offsets[i]) return fail(QM_E_ARG, "interval offsets not monotone");
    if (read_len[i] < 0 || read_len[i] > (len_limit(o))) return fail(QM_E_TOOLONG, "read length %d > %d", read_len[i], len_limit(o));
    int nf = 0, nr = 0; bool seenRc = false;
    for (int64_t j = int_offsets[i]; j < int_offsets[i + 1]; ++j) {
      if (ints[j].query_rc) { ++nr; seenRc = true; } else { ++nf; if (seenRc) return fail(QM_E_ARG, "read %lld: forward-strand intervals must precede the reverse-complement ones", (long long)i); }
      if ((int64_t)(uint32_t)ints[j].end > c->ix->nSA || (uint32_t)ints[j].end < (uint32_t)ints[j].begin) return fail(QM_E_ARG, "read %lld: SA interval out of range", (long long)i);
    }
    if (nf > QM_ICAP + QM_IOVF || nr > QM_ICAP + QM_IOVF) return fail(QM_E_ARG, "read %lld: more than %d intervals on one strand", (long long)i, QM_ICAP + QM_IOVF);
  }
  if ((rc = ensure(c->d_ivIn, c->capIvIn, ni + 1))) return rc;
  if ((rc = ensure(c->d_ivInOff, c->capIvInOff, n + 1))) return rc;
  if ((rc = ensure(c->d_lenIn, c->capLenIn, n + 1))) return rc;
  if ((rc = upload(c, c->d_ivIn, ints, (size_t)ni * sizeof(qm_sa_interval_hit)))) return rc;
  if ((rc = upload(c, c->d_ivInOff, int_offsets, (size_t)(n + 1) * 8))) return rc;
  if ((rc = upload(c, c->d_lenIn, read_len, (size_t)n * 4))) return rc;
  RunReq rq; rq.mode = QM_RUN_FROM_INTERVALS; rq.ivIn = c->d_ivIn; rq.ivInOff = c->d_ivInOff; rq.lenIn = c->d_lenIn;
  u64 hscal[QM_SC_WORDS];
  c->lastUnits = -1;
  // the kernel walks the reads of one "mate"; seq/off are not touched in this mode (a non-null seq1 keeps the argument checks simple)
  if ((rc = run_stage_a(c, o, rq, n, c->d_lenIn, c->d_ivInOff, nullptr, nullptr, 4, nullptr, hscal))) return rc;
  if (n_words) {
    std::vector<uint32_t> cnt((size_t)n + 1);
    if (n) HIPCHK(hipMemcpy(cnt.data(), c->d_lcnt, (size_t)n * 4, hipMemcpyDeviceToHost));
    int64_t t = 0; for (int64_t i = 0; i < n; ++i) t += cnt[(size_t)i] & 0x7fffffffu;
    *n_words = t;
  }
  return QM_OK;
}

int qm_fetch_read_lists(qm_ctx* c, int64_t* list_offsets, uint64_t* words, int64_t cap) {
  if (!c || c->lastListReads < 0) return fail(QM_E_STATE, "no per-read hit lists kept by the last call");
  if (!list_offsets) return fail(QM_E_ARG, "null list_offsets");
  HIPCHK(hipSetDevice(c->device));
  const int64_t nreads = c->lastListReads;
  std::vector<uint32_t> cnt((size_t)nreads + 1); std::vector<long long> off((size_t)nreads + 1);
  if (nreads) {
    HIPCHK(hipMemcpy(cnt.data(), c->d_lcnt, (size_t)nreads * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(off.data(), c->d_loff, (size_t)nreads * 8, hipMemcpyDeviceToHost));
  }
  list_offsets[0] = 0;
  for (int64_t r = 0; r < nreads; ++r) list_offsets[r + 1] = list_offsets[r] + (int64_t)(cnt[(size_t)r] & 0x7fffffffu);
  if (!words) return QM_OK;
  if (cap < list_offsets[nreads]) return fail(QM_E_ARG, "list buffer too small");
  // the lists sit in bump-allocated chunks: bring the used part of the buffer down once, gather on the host
  const int64_t used = c->lastListWords < c->capLists ? c->lastListWords : c->capLists;
  std::vector<uint64_t> all((size_t)used + 1);
  if (used) HIPCHK(hipMemcpy(all.data(), c->d_lists, (size_t)used * 8, hipMemcpyDeviceToHost));
  for (int64_t r = 0; r < nreads; ++r) {
    const int64_t k = (int64_t)(cnt[(size_t)r] & 0x7fffffffu);
    if (k) memcpy(words + list_offsets[r], all.data() + off[(size_t)r], (size_t)k * 8);
  }
  return QM_OK;
}

int qm_merge_lists(qm_ctx* c, const qm_opts* o, int64_t n, const int64_t* loff_left, const uint64_t* words_left, const int64_t* loff_right,
                   const uint64_t* words_right, const uint8_t* found_left, const uint8_t* found_right, const int32_t* len_left,
                   const int32_t* len_right, int64_t* n_hits, qm_counters* counters) {
  if (!c || n < 0 || (n > 0 && (!loff_left || !loff_right || !len_left || !len_right))) return fail(QM_E_ARG, "bad argument");
  int rc = check_opts(o);
  if (rc) return rc;
  if (o->sel_aln && !(found_left && found_right)) return fail(QM_E_ARG, "mergeLeftRightHitsFuzzy needs leftMatches / rightMatches");
  HIPCHK(hipSetDevice(c->device));
  static const int64_t zero = 0;
  if (n == 0) { loff_left = &zero; loff_right = &zero; }
  const int64_t wl = loff_left[n], wr = loff_right[n];
  if ((wl > 0 && !words_left) || (wr > 0 && !words_right)) return fail(QM_E_ARG, "null list words");
  // device image: read 2u = left list of unit u, read 2u + 1 = right list; left words first, right words behind them
  std::vector<uint32_t> cnt((size_t)(2 * n) + 1); std::vector<long long> off((size_t)(2 * n) + 1), o1((size_t)n + 1), o2((size_t)n + 1);
  const bool flags = o->fuzzy || o->sel_aln;
  o1[0] = 0; o2[0] = 0;
  for (int64_t u = 0; u < n; ++u) {
    const int64_t a = loff_left[u + 1] - loff_left[u], b = loff_right[u + 1] - loff_right[u];
    if (a < 0 || b < 0 || a > 0x7ffffffe || b > 0x7ffffffe) return fail(QM_E_ARG, "list offsets not monotone");
    cnt[(size_t)(2 * u)] = (uint32_t)a | ((flags && found_left && found_left[u]) ? 0x80000000u : 0u);
    cnt[(size_t)(2 * u + 1)] = (uint32_t)b | ((flags && found_right && found_right[u]) ? 0x80000000u : 0u);
    off[(size_t)(2 * u)] = loff_left[u]; off[(size_t)(2 * u + 1)] = wl + loff_right[u];
    o1[(size_t)u + 1] = o1[(size_t)u] + len_left[u]; o2[(size_t)u + 1] = o2[(size_t)u] + len_right[u];
  }
  if ((rc = ensure(c->d_lcnt, c->capLcnt, 2 * n + 1))) return rc;
  if ((rc = ensure(c->d_loff, c->capLoff, 2 * n + 1))) return rc;
  if (c->capLists < wl + wr + 1) { if ((rc = ensure(c->d_lists, c->capLists, wl + wr + 1))) return rc; }
  if ((rc = ensure(c->d_off1, c->capOff1, n + 1))) return rc;
  if ((rc = ensure(c->d_off2, c->capOff2, n + 1))) return rc;
  if ((rc = upload(c, c->d_lcnt, cnt.data(), (size_t)(2 * n) * 4))) return rc;
  if ((rc = upload(c, c->d_loff, off.data(), (size_t)(2 * n) * 8))) return rc;
  if ((rc = upload(c, c->d_lists, words_left, (size_t)wl * 8))) return rc;
  if ((rc = upload(c, c->d_lists + wl, words_right, (size_t)wr * 8))) return rc;
  if ((rc = upload(c, c->d_off1, o1.data(), (size_t)(n + 1) * 8))) return rc;
  if ((rc = upload(c, c->d_off2, o2.data(), (size_t)(n + 1) * 8))) return rc;
  HIPCHK(hipMemsetAsync(c->d_scal, 0, QM_SC_WORDS * sizeof(u64), c->stream));
  RunReq rq; rq.mergeOnly = true;
  u64 hscal[QM_SC_WORDS]; long long total = 0;
  c->lastUnits = -1; c->lastListReads = -1; c->lastIvReads = -1; c->lastFoundReads = -1;
  if ((rc = run_stage_b(c, o, rq, n, true, nullptr, c->d_off1, nullptr, c->d_off2, hscal, total))) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->lastUnits = n; c->lastHits = total; c->lastPaired = true; c->lastMapMs = 0; c->lastTotalMs = 0;
  if (n_hits) *n_hits = total;
  if (counters) {
    counters->pe_hits = hscal[1]; counters->se_hits = hscal[2]; counters->tot_hits = hscal[3];
    counters->num_reads = hscal[4]; counters->too_many_hits = hscal[5]; counters->mapped = hscal[6];
  }
  return QM_OK;
}

int qm_fetch_too_many(qm_ctx* c, uint8_t* too_many) {
  if (!c || c->lastTooManyUnits < 0) return fail(QM_E_STATE, "no tooManyHits flags kept by the last call");
  if (!too_many) return fail(QM_E_ARG, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  if (c->lastTooManyUnits) HIPCHK(hipMemcpy(too_many, c->d_tooMany, (size_t)c->lastTooManyUnits, hipMemcpyDeviceToHost));
  return QM_OK;
}

int qm_map_pairs_stages(qm_ctx* c, const qm_opts* o, int64_t n, const char* seq1, const int64_t* off1, const char* seq2,
                        const int64_t* off2, int64_t* n_hits, qm_counters* counters) {
  if (!seq2 || !off2) return fail(QM_E_ARG, "qm_map_pairs_stages needs both mates");
  RunReq rq; rq.keepIntervals = true; rq.keepFound = true; rq.mergeOnly = true; rq.stageView = true;
  return map_host(c, o, n, seq1, off1, seq2, off2, n_hits, counters, rq);
}

static RunReq stage_req(uint32_t stage_flags) {
  RunReq rq; rq.keepIntervals = !(stage_flags & QM_STAGES_NO_INTERVALS); rq.keepFound = true; rq.mergeOnly = true; rq.stageView = true;
  return rq;
}
int qm_map_pairs_stages_ex(qm_ctx* c, const qm_opts* o, int64_t n, const char* seq1, const int64_t* off1, const char* seq2,
                           const int64_t* off2, uint32_t stage_flags, int64_t* n_hits, qm_counters* counters) {
  if (!seq2 || !off2) return fail(QM_E_ARG, "qm_map_pairs_stages_ex needs both mates");
  return map_host(c, o, n, seq1, off1, seq2, off2, n_hits, counters, stage_req(stage_flags));
}
int qm_map_pairs_stages_packed(qm_ctx* c, const qm_opts* o, int64_t n, const uint8_t* pk1, const int64_t* off1, const qm_pack_exc* exc1, int64_t nexc1,
                               const uint8_t* pk2, const int64_t* off2, const qm_pack_exc* exc2, int64_t nexc2, uint32_t stage_flags, int64_t* n_hits, qm_counters* counters) {
  if (!pk2 || !off2) return fail(QM_E_ARG, "qm_map_pairs_stages_packed needs both mates");
  return map_packed(c, o, n, pk1, off1, exc1, nexc1, pk2, off2, exc2, nexc2, n_hits, counters, stage_req(stage_flags));
}

// layout of a qm_fetch_stages arena: eight arrays, each aligned to 64 bytes
namespace {
struct StageLayout { size_t ivOff, iv, found, listOff, words, hitOff, hits, tooMany, total; };
StageLayout stage_layout(int64_t nreads, int64_t n, int64_t nIv, int64_t nWords, int64_t nHits) {
  StageLayout L; size_t p = 0;
  auto take = [&](size_t bytes) { const size_t at = p; p += (bytes + 63) & ~(size_t)63; return at; };
  L.ivOff = take((size_t)(nreads + 1) * 8); L.iv = take((size_t)(nIv + 1) * sizeof(qm_sa_interval_hit)); L.found = take((size_t)nreads + 1);
  L.listOff = take((size_t)(nreads + 1) * 8); L.words = take((size_t)(nWords + 1) * 8);
  L.hitOff = take((size_t)(n + 1) * 8); L.hits = take((size_t)(nHits + 1) * sizeof(qm_hit)); L.tooMany = take((size_t)n + 1);
  L.total = p;
  return L;
}
}  // namespace

int qm_stage_bytes(qm_ctx* c, int64_t* bytes) {
  if (!c || !bytes) return fail(QM_E_ARG, "null argument");
  if (c->stUnits < 0 || c->stReads < 0 || c->lastUnits != c->stUnits) return fail(QM_E_STATE, "qm_stage_bytes: the last call on this context was not qm_map_pairs_stages");
  *bytes = (int64_t)stage_layout(c->stReads, c->stUnits, c->h_tot[0], c->h_tot[1], c->lastHits).total;
  return QM_OK;
}

int qm_fetch_stages(qm_ctx* c, void* arena, int64_t arena_bytes, qm_stage_view* v) {
  if (!c || !arena || !v) return fail(QM_E_ARG, "null argument");
  if (c->stUnits < 0 || c->stReads < 0 || c->lastUnits != c->stUnits) return fail(QM_E_STATE, "qm_fetch_stages: the last call on this context was not qm_map_pairs_stages");
  HIPCHK(hipSetDevice(c->device));
  const int64_t nreads = c->stReads, n = c->stUnits, nIv = c->h_tot[0], nWords = c->h_tot[1], nHits = c->lastHits;
  const StageLayout L = stage_layout(nreads, n, nIv, nWords, nHits);
  if ((int64_t)L.total > arena_bytes) return fail(QM_E_ARG, "qm_fetch_stages: arena of %lld bytes, %lld needed (qm_stage_bytes)", (long long)arena_bytes, (long long)L.total);
  int rc;
  if ((rc = ensure(c->d_ivC, c->capIvC, nIv + 1, nIv / 4))) return rc;
  if ((rc = ensure(c->d_wordsC, c->capWordsC, nWords + 1, nWords / 4))) return rc;
  HIPCHK(qmk_stage_gather(nreads, c->d_ivcnt, c->d_ivoff, c->d_iv, c->d_ivcsr, c->d_ivC, c->d_lcnt, c->d_loff, (const unsigned long long*)c->d_lists, c->d_lcsr,
                          (unsigned long long*)c->d_wordsC, c->stream));
  unsigned char* A = (unsigned char*)arena;
  auto down = [&](size_t at, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(A + at, src, bytes, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
  };
  HIPCHK(down(L.ivOff, c->d_ivcsr, (size_t)(nreads + 1) * 8));
  HIPCHK(down(L.iv, c->d_ivC, (size_t)nIv * sizeof(qm_sa_interval_hit)));
  HIPCHK(down(L.found, c->d_found, (size_t)nreads));
  HIPCHK(down(L.listOff, c->d_lcsr, (size_t)(nreads + 1) * 8));
  HIPCHK(down(L.words, c->d_wordsC, (size_t)nWords * 8));
  HIPCHK(down(L.hitOff, c->d_offs, (size_t)(n + 1) * 8));
  HIPCHK(down(L.hits, c->d_hits, (size_t)nHits * sizeof(qm_hit)));
  HIPCHK(down(L.tooMany, c->d_tooMany, (size_t)n));
  HIPCHK(hipStreamSynchronize(c->stream));
  v->n_units = n; v->n_reads = nreads;
  v->iv_off = (const int64_t*)(A + L.ivOff); v->iv = (const qm_sa_interval_hit*)(A + L.iv); v->found = A + L.found;
  v->list_off = (const int64_t*)(A + L.listOff); v->words = (const uint64_t*)(A + L.words);
  v->hit_off = (const int64_t*)(A + L.hitOff); v->hits = (const qm_hit*)(A + L.hits); v->too_many = A + L.tooMany;
  if (nreads == 0) { ((int64_t*)(A + L.ivOff))[0] = 0; ((int64_t*)(A + L.listOff))[0] = 0; }
  if (n == 0) ((int64_t*)(A + L.hitOff))[0] = 0;
  return QM_OK;
}

void* qm_pinned_alloc(int64_t bytes) {
  void* p = nullptr;
  return hipHostMalloc(&p, bytes > 0 ? (size_t)bytes : 64, hipHostMallocPortable) == hipSuccess ? p : nullptr;
}
void qm_pinned_free(void* p) { if (p) hipHostFree(p); }

int qm_ctx_stat(const qm_ctx* c, int which, int64_t* value) {
  if (!c || !value) return fail(QM_E_ARG, "null argument");
  switch (which) {
    case QM_STAT_RELAUNCHES: *value = c->lastRelaunches; break;
    case QM_STAT_LIST_WORDS: *value = c->capLists; break;
    case QM_STAT_SLOW_READS: *value = c->lastSlowReads; break;
    case QM_STAT_LEAN_READS: *value = c->lastLeanReads; break;
    case QM_STAT_LEAN_DEFERRED: *value = c->lastLeanDeferred; break;
    case QM_STAT_SKIPPED_READS: *value = c->lastSkipped; break;
    case QM_STAT_SEL_QUESTIONS: *value = c->lastSelQuestions; break;
    case QM_STAT_KSW2_ALIGNMENTS: *value = c->lastKswTasks; break;
    case QM_STAT_STRIP_ALIGNMENTS: *value = c->lastStripTasks; break;
    case QM_STAT_PAIR_KERNEL_PAIRS: *value = c->lastDuoPairs; break;
    case QM_STAT_PAIRS_MERGED: *value = c->lastDuoMerged; break;
    case QM_STAT_N_PASS_READS: *value = c->lastNPass; break;
    case QM_STAT_DEFER_DIRTY: case QM_STAT_DEFER_HOMOPOLYMER: case QM_STAT_DEFER_WIDE: case QM_STAT_DEFER_BOTH_STRANDS: *value = c->lastDefer[which - QM_STAT_DEFER_DIRTY]; break;
    default: return fail(QM_E_ARG, "unknown statistic %d", which);
  }
  return QM_OK;
}

int qm_fetch_skipped(const qm_ctx* c, int64_t* reads, int32_t* codes, int64_t cap, int64_t* total) {
  if (!c || !total) return fail(QM_E_ARG, "null argument");
  *total = c->lastSkipped;
  const int64_t n = (int64_t)c->skipList.size() < cap ? (int64_t)c->skipList.size() : cap;
  for (int64_t i = 0; i < n; ++i) {
    if (reads) reads[i] = (int64_t)(c->skipList[(size_t)i] & ((1ULL << 56) - 1));
    if (codes) codes[i] = (int32_t)(c->skipList[(size_t)i] >> 56);
  }
  return QM_OK;
}

int qm_last_kernel_ms(const qm_ctx* c, double* map_ms, double* total_ms) {
  if (!c || c->lastUnits < 0) return fail(QM_E_STATE, "no mapping result");
  if (map_ms) *map_ms = c->lastMapMs;
  if (total_ms) *total_ms = c->lastTotalMs;
  return QM_OK;
}

}  // extern "C"
