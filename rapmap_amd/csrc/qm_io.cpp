// qm_io.cpp -- host-side callers of the hot path:
//   * qm_reader_*    FASTA/FASTQ(.gz) ingest into the packed (bytes, offsets[n+1]) batches qm_map_pairs takes:
//                    the ingest engine of qm_ingest.cpp (which replaces the reference's single kseq producer +
//                    per-record std::string queue, include/FastxParser.hpp:62-66, src/FastxParser.cpp:229-328)
//                    with malloc'd slots, handed out one batch at a time.
//   * qm_sam_*       SAM text for a mapped batch, same bytes as `rapmap quasimap -o`
//                    (writeSAMHeader include/RapMapUtils.hpp:97-115, writeAlignmentsToStream
//                    src/RapMapUtils.cpp:198-588, writeUnalignedPairToStream :137-196, getSamFlags /
//                    adjustOverhang include/RapMapUtils.hpp:687-810), formatted by n_threads workers.
// Plain C++ (no HIP); part of libqmap_mi355.so.
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/qmap_mi355.h"
#include "qm_io_internal.h"
#include "qm_pack.h"

static thread_local char g_ioerr[512] = "";
int qm_io_fail(int code, const char* fmt, ...) {          // (shared with qm_ingest.cpp)
  va_list ap; va_start(ap, fmt); vsnprintf(g_ioerr, sizeof(g_ioerr), fmt, ap); va_end(ap);
  return code;
}
#define io_fail qm_io_fail
extern "C" const char* qm_io_last_error(void) { return g_ioerr; }

namespace {

// Worker threads that live as long as the reader / writer that owns them: a batch of 2^18 pairs is a few milliseconds of
// work per phase, about what starting and joining 64 threads costs.  run(T, fn) calls fn(0) .. fn(T-1), the caller included.
class Pool {
  std::vector<std::thread> th_;
  std::mutex mu_; std::condition_variable work_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int T_ = 0, next_ = 0, finished_ = 0; bool stop_ = false;
  void drain(std::unique_lock<std::mutex>& lk) {
    while (next_ < T_) {
      const int t = next_++;
      const std::function<void(int)>* f = fn_;
      lk.unlock(); (*f)(t); lk.lock();
      if (++finished_ == T_) done_.notify_all();
    }
  }
 public:
  explicit Pool(int n) {
    for (int i = 1; i < n; ++i) th_.emplace_back([this]() {
      std::unique_lock<std::mutex> lk(mu_);
      while (true) { work_.wait(lk, [&] { return stop_ || next_ < T_; }); if (stop_) return; drain(lk); }
    });
  }
  ~Pool() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } work_.notify_all(); for (auto& t : th_) t.join(); }
  int size() const { return (int)th_.size() + 1; }
  void run(int T, const std::function<void(int)>& fn) {
    if (T <= 1) { if (T == 1) fn(0); return; }
    std::unique_lock<std::mutex> lk(mu_);
    fn_ = &fn; T_ = T; next_ = 0; finished_ = 0;
    work_.notify_all();
    drain(lk);
    done_.wait(lk, [&] { return finished_ == T_; });
    T_ = 0; next_ = 0; fn_ = nullptr;
  }
};

}  // namespace

// ------------------------------------------------------------------ qm_reader_*: the ingest engine (qm_ingest.cpp) behind the reader calls
// A reader is the engine with plain malloc'd slots; the engine parses and packs ahead of the caller (three slots).  The
// batch size is fixed by the first qm_reader_next call; a later call that asks for fewer units gets the current batch in pieces.
struct qm_reader {
  std::string path1, path2; bool paired = false; int nthreads = 1;
  qm_ingest* g = nullptr; int64_t batchUnits = 0;
  int slot = -1; int64_t n = 0, served = 0; const qm_batch_bufs* bufs = nullptr;
  std::vector<int64_t> offTmp[2], noffTmp[2];       // rebased offsets of a piece that does not start the batch
};

extern "C" {

int64_t qm_packed_offset(const int64_t* off, int64_t i) { return (off[i] >> 2) + i; }
int64_t qm_packed_bytes(const int64_t* off, int64_t n) { return (off[n] >> 2) + n + 8; }

int qm_pack_reads(const char* seq, const int64_t* off, int64_t n, uint8_t* packed, qm_pack_exc* exc, int64_t exc_cap, int64_t* n_exc) {
  if (n < 0 || (n > 0 && (!seq || !off || !packed)) || !n_exc) return io_fail(QM_E_ARG, "qm_pack_reads: bad argument");
  if (n > 0 && off[n] > 0xffffffffLL) return io_fail(QM_E_ARG, "qm_pack_reads: more than 2^32 characters in one batch");
  // reads never share a packed byte: the batch is cut into runs of reads that threads pack side by side
  const int nt = n >= (1 << 16) ? 8 : 1;
  std::vector<std::vector<qm_pack_exc>> ex((size_t)nt);
  std::vector<std::thread> th;
  auto work = [&](int t) {
    const int64_t a = n * t / nt, b = n * (t + 1) / nt;
    for (int64_t i = a; i < b; ++i) {
      const int64_t o = off[i], len = off[i + 1] - o;
      if (len < 0) continue;
      qm_pack::pack_read((const unsigned char*)seq + o, (size_t)len, packed + qm_packed_offset(off, i),
                         [&](size_t j, unsigned char c) { ex[(size_t)t].push_back(qm_pack_exc{(uint32_t)(o + (int64_t)j), (uint32_t)c}); });
    }
  };
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  int64_t tot = 0; for (auto& v : ex) tot += (int64_t)v.size();
  *n_exc = tot;
  if (tot > exc_cap || (tot > 0 && !exc)) return io_fail(QM_E_ARG, "qm_pack_reads: %lld exceptions, room for %lld", (long long)tot, (long long)exc_cap);
  int64_t w = 0; for (auto& v : ex) { if (!v.empty()) memcpy(exc + w, v.data(), v.size() * sizeof(qm_pack_exc)); w += (int64_t)v.size(); }
  return QM_OK;
}

int qm_reader_open(const char* path1, const char* path2, int32_t n_threads, qm_reader** out) {
  if (!path1 || !out) return io_fail(QM_E_ARG, "qm_reader_open: null argument");
  for (const char* p : {path1, path2}) {
    if (!p) continue;
    int fd = ::open(p, O_RDONLY);
    if (fd < 0) return io_fail(QM_E_IO, "cannot open %s", p);
    ::close(fd);
  }
  qm_reader* r = new qm_reader();
  r->path1 = path1; r->paired = path2 != nullptr; if (path2) r->path2 = path2;
  r->nthreads = n_threads > 0 ? n_threads : 1;
  *out = r;
  return QM_OK;
}

void qm_reader_close(qm_reader* r) {
  if (!r) return;
  if (r->g) qm_ingest_close(r->g);
  delete r;
}

int qm_reader_next(qm_reader* r, int64_t max_units, int64_t* n_units, const char** seq1, const int64_t** off1,
                   const char** names1, const int64_t** name_off1, const char** seq2, const int64_t** off2,
                   const char** names2, const int64_t** name_off2) {
  if (!r || !n_units || max_units <= 0) return io_fail(QM_E_ARG, "qm_reader_next: bad argument");
  int rc;
  if (!r->g) {
    r->batchUnits = max_units;
    if ((rc = qm_ingest_open(r->path1.c_str(), r->paired ? r->path2.c_str() : nullptr, r->nthreads, max_units, 3, 0, malloc, free, &r->g))) return rc;
  }
  if (r->slot < 0 || r->served >= r->n) {
    if (r->slot >= 0) { qm_ingest_release(r->g, r->slot); r->slot = -1; }
    if ((rc = qm_ingest_next(r->g, &r->slot, &r->n, nullptr, &r->bufs))) return rc;
    r->served = 0;
    if (r->n == 0) { *n_units = 0; return QM_OK; }
  }
  const int64_t i0 = r->served, cnt = std::min(max_units, r->n - i0);
  r->served += cnt;
  *n_units = cnt;
  const qm_batch_bufs& B = *r->bufs;
  const char** seqOut[2] = {seq1, seq2}; const int64_t** offOut[2] = {off1, off2};
  const char** nmOut[2] = {names1, names2}; const int64_t** noffOut[2] = {name_off1, name_off2};
  for (int s = 0; s < (r->paired ? 2 : 1); ++s) {
    if (i0 == 0) {
      if (seqOut[s]) *seqOut[s] = B.seq[s];
      if (offOut[s]) *offOut[s] = B.off[s];
      if (nmOut[s]) *nmOut[s] = B.names[s];
      if (noffOut[s]) *noffOut[s] = B.noff[s];
    } else {                                   // a later piece of the batch: same bytes, offsets rebased to the piece
      r->offTmp[s].resize((size_t)cnt + 1); r->noffTmp[s].resize((size_t)cnt + 1);
      for (int64_t i = 0; i <= cnt; ++i) { r->offTmp[s][(size_t)i] = B.off[s][i0 + i] - B.off[s][i0]; r->noffTmp[s][(size_t)i] = B.noff[s][i0 + i] - B.noff[s][i0]; }
      if (seqOut[s]) *seqOut[s] = B.seq[s] + B.off[s][i0];
      if (offOut[s]) *offOut[s] = r->offTmp[s].data();
      if (nmOut[s]) *nmOut[s] = B.names[s] + B.noff[s][i0];
      if (noffOut[s]) *noffOut[s] = r->noffTmp[s].data();
    }
  }
  return QM_OK;
}

}  // extern "C"


// ------------------------------------------------------------------ SAM
namespace {

struct RcTab { char t[256]; RcTab() { memset(t, 'N', 256); const char* a = "ACGTUacgtu"; const char* b = "TGCAATGCAA"; for (int i = 0; a[i]; ++i) t[(unsigned char)a[i]] = b[i]; } };
static const RcTab RCT;
static void reverse_read(const char* s, int64_t n, std::string& out) {      // src/RapMapUtils.cpp:107-128
  out.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) out[(size_t)(n - 1 - i)] = RCT.t[(unsigned char)s[i]];
}
// Output buffer of one formatter thread: raw bytes with amortised growth, no per-character bookkeeping.
struct Out {
  char* b = nullptr; size_t n = 0, cap = 0;
  Out() = default;
  Out(const Out&) = delete; Out& operator=(const Out&) = delete;
  Out(Out&& o) noexcept : b(o.b), n(o.n), cap(o.cap) { o.b = nullptr; o.n = o.cap = 0; }
  Out& operator=(Out&& o) noexcept { if (this != &o) { free(b); b = o.b; n = o.n; cap = o.cap; o.b = nullptr; o.n = o.cap = 0; } return *this; }
  ~Out() { free(b); }
  inline void need(size_t m) {
    if (n + m > cap) { cap = (n + m) * 2 + 4096; b = (char*)realloc(b, cap); }
  }
  inline void raw(const char* p, size_t l) { memcpy(b + n, p, l); n += l; }            // caller reserved
  inline void ch(char c) { b[n++] = c; }
  inline void num(long long v) {                                                         // decimal, no printf
    char t[24]; int k = 0;
    unsigned long long u = v < 0 ? (unsigned long long)(-(v + 1)) + 1ULL : (unsigned long long)v;
    do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) b[n++] = '-';
    while (k) b[n++] = t[--k];
  }
  template <size_t N> inline void lit(const char (&x)[N]) { memcpy(b + n, x, N - 1); n += N - 1; }
};
struct Cigar { char c[48]; int n; };
static inline void cg_num(Cigar& g, long long v) { char t[24]; int k = 0; do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k) g.c[g.n++] = t[--k]; }
// include/RapMapUtils.hpp:687-711
static int32_t adjust_overhang(int32_t pos, uint32_t readLen, int64_t txpLen, Cigar& g) {
  g.n = 0;
  const long long rl = readLen;
  if (pos + rl < 0) { cg_num(g, rl); g.c[g.n++] = 'S'; return 0; }
  if (pos < 0) { long long match = rl + pos, clip = rl - match; cg_num(g, clip); g.c[g.n++] = 'S'; cg_num(g, match); g.c[g.n++] = 'M'; return 0; }
  if (pos > txpLen) { cg_num(g, rl); g.c[g.n++] = 'S'; return pos; }
  if (pos + rl > txpLen) { long long match = txpLen - pos, clip = rl - match; cg_num(g, match); g.c[g.n++] = 'M'; cg_num(g, clip); g.c[g.n++] = 'S'; return pos; }
  cg_num(g, rl); g.c[g.n++] = 'M'; return pos;
}
static inline void tags(Out& o, long long nh, long long hi, long long as) {
  o.lit("\tNH:i:"); o.num(nh); o.lit("\tHI:i:"); o.num(hi); o.lit("\tAS:i:"); o.num(as); o.ch('\n');
}

struct SamCtx {
  int maxHits;
  std::vector<const char*> tname; std::vector<uint32_t> tnl; std::vector<int64_t> tlen;   // per transcript
};
static inline void name_len(const char* nm, int64_t n, bool trimMate, int64_t& l) {       // src/RapMapUtils.cpp:334-351
  const char* sp = (const char*)memchr(nm, ' ', (size_t)n);
  l = sp ? sp - nm : n;
  if (trimMate && l > 2 && nm[l - 2] == '/') l -= 2;
}

static void format_pair(const SamCtx& C, const char* nm1, int64_t nl1, const char* s1, int64_t l1, const char* nm2,
                        int64_t nl2, const char* s2, int64_t l2, const qm_hit* h, int64_t nh, Out& o,
                        std::string& rev1, std::string& rev2) {
  int64_t a1, a2;
  name_len(nm1, nl1, true, a1); name_len(nm2, nl2, true, a2);
  if (nh == 0 || nh > C.maxHits) {                       // writeUnalignedPairToStream
    o.need((size_t)(a1 + a2 + l1 + l2) + 256);
    for (int m = 0; m < 2; ++m) {
      o.raw(m == 0 ? nm1 : nm2, (size_t)(m == 0 ? a1 : a2)); o.ch('\t'); o.num(0x1 | 0x4 | 0x8 | (m == 0 ? 0x40 : 0x80));
      o.lit("\t*\t0\t255\t*\t*\t*\t0\t"); o.raw(m == 0 ? s1 : s2, (size_t)(m == 0 ? l1 : l2)); o.lit("\t*\tNH:i:0\tHI:i:0\tAS:i:0\n");
    }
    return;
  }
  bool have1 = false, have2 = false;
  Cigar c1, c2;
  for (int64_t i = 0; i < nh; ++i) {
    const qm_hit& q = h[i];
    const char* tname = C.tname[q.tid]; const size_t tnl = C.tnl[q.tid];
    const int64_t tlen = C.tlen[q.tid];
    o.need((size_t)(a1 + a2 + l1 + l2) + 2 * tnl + 512);
    const bool fwd = q.fwd != 0, mfwd = q.mate_is_fwd != 0, paired = q.is_paired != 0;
    // getSamFlags (include/RapMapUtils.hpp:771-810)
    int f1 = 0x1 | (paired ? 0x2 : 0), f2 = f1;
    if (q.mate_status == 2) { f1 |= 0x4; f2 |= 0x8; }
    if (q.mate_status == 1) { f2 |= 0x4; f1 |= 0x8; }
    if (!fwd) { f1 |= 0x10; f2 |= 0x20; }
    if (!mfwd) { f1 |= 0x20; f2 |= 0x10; }
    f1 |= 0x40; f2 |= 0x80;
    if (i != 0) { f1 |= 0x100; f2 |= 0x100; }
    if (paired) {
      int32_t pos = adjust_overhang(q.pos, q.read_len, tlen, c1);
      int32_t mpos = adjust_overhang(q.mate_pos, q.mate_len, tlen, c2);
      if (!fwd && !have1) { reverse_read(s1, l1, rev1); have1 = true; }
      if (!mfwd && !have2) { reverse_read(s2, l2, rev2); have2 = true; }
      const bool r1First = pos < mpos;
      long long frag = (int32_t)q.frag_len;              // src/RapMapUtils.cpp:407-411 (int32 casts)
      const long long minPos = r1First ? pos : mpos;
      if (minPos + frag > tlen) frag = tlen - minPos;
      o.raw(nm1, (size_t)a1); o.ch('\t'); o.num(f1); o.ch('\t'); o.raw(tname, tnl); o.ch('\t'); o.num(pos + 1LL); o.lit("\t1\t"); o.raw(c1.c, (size_t)c1.n); o.lit("\t=\t");
      o.num(mpos + 1LL); o.ch('\t'); o.num(r1First ? frag : -frag); o.ch('\t');
      if (fwd) o.raw(s1, (size_t)l1); else o.raw(rev1.data(), rev1.size());
      o.lit("\t*"); tags(o, nh, i + 1, q.aln_score);
      o.raw(nm2, (size_t)a2); o.ch('\t'); o.num(f2); o.ch('\t'); o.raw(tname, tnl); o.ch('\t'); o.num(mpos + 1LL); o.lit("\t1\t"); o.raw(c2.c, (size_t)c2.n); o.lit("\t=\t");
      o.num(pos + 1LL); o.ch('\t'); o.num(r1First ? -frag : frag); o.ch('\t');
      if (mfwd) o.raw(s2, (size_t)l2); else o.raw(rev2.data(), rev2.size());
      o.lit("\t*"); tags(o, nh, i + 1, q.aln_score);
    } else {
      const bool left = q.mate_status == 1;
      const char* an = left ? nm1 : nm2; const size_t anl = (size_t)(left ? a1 : a2);
      const char* un = left ? nm2 : nm1; const size_t unl = (size_t)(left ? a2 : a1);
      const int afl = left ? f1 : f2, ufl = left ? f2 : f1;
      int32_t pos = adjust_overhang(q.pos, q.read_len, tlen, c1);
      o.raw(an, anl); o.ch('\t'); o.num(afl); o.ch('\t'); o.raw(tname, tnl); o.ch('\t'); o.num(pos + 1LL); o.lit("\t1\t"); o.raw(c1.c, (size_t)c1.n); o.lit("\t=\t");
      o.num(pos + 1LL); o.lit("\t0\t");
      if (fwd) { if (left) o.raw(s1, (size_t)l1); else o.raw(s2, (size_t)l2); }
      else if (left) { if (!have1) { reverse_read(s1, l1, rev1); have1 = true; } o.raw(rev1.data(), rev1.size()); }
      else { if (!have2) { reverse_read(s2, l2, rev2); have2 = true; } o.raw(rev2.data(), rev2.size()); }
      o.lit("\t*"); tags(o, nh, i + 1, q.aln_score);
      o.raw(un, unl); o.ch('\t'); o.num(ufl); o.ch('\t'); o.raw(tname, tnl); o.ch('\t'); o.num(pos + 1LL); o.lit("\t0\t*\t=\t"); o.num(pos + 1LL); o.lit("\t0\t");
      if (left) o.raw(s2, (size_t)l2); else o.raw(s1, (size_t)l1);
      o.lit("\t*"); tags(o, nh, i + 1, q.aln_score);
    }
  }
}

// single-end records (src/RapMapUtils.cpp:198-311): MAPQ 255, 0x10 for rc, 0x900 on secondary hits; the
// single-end writer strips the name at the first blank only.
static void format_single(const SamCtx& C, const char* nm, int64_t nl, const char* s, int64_t l, const qm_hit* h,
                          int64_t nh, Out& o, std::string& rev) {
  int64_t a1; name_len(nm, nl, false, a1);
  if (nh == 0) {
    o.need((size_t)(a1 + l) + 128);
    o.raw(nm, (size_t)a1); o.lit("\t4\t*\t0\t255\t*\t*\t0\t0\t"); o.raw(s, (size_t)l); o.lit("\t*\tNH:i:0\tHI:i:0\tAS:i:0\n");
    return;
  }
  bool have = false; Cigar c1;
  for (int64_t i = 0; i < nh; ++i) {
    const qm_hit& q = h[i];
    int fl = q.fwd ? 0 : 0x10;
    if (i != 0) fl |= 0x900;
    int32_t pos = adjust_overhang(q.pos, q.read_len, C.tlen[q.tid], c1);
    o.need((size_t)(a1 + l) + C.tnl[q.tid] + 256);
    o.raw(nm, (size_t)a1); o.ch('\t'); o.num(fl); o.ch('\t'); o.raw(C.tname[q.tid], C.tnl[q.tid]); o.ch('\t'); o.num(pos + 1LL);
    o.lit("\t255\t"); o.raw(c1.c, (size_t)c1.n); o.lit("\t*\t0\t"); o.num((long long)q.frag_len); o.ch('\t');
    if (q.fwd) o.raw(s, (size_t)l); else { if (!have) { reverse_read(s, l, rev); have = true; } o.raw(rev.data(), rev.size()); }
    o.lit("\t*"); tags(o, nh, i + 1, q.aln_score);
  }
}

}  // namespace

extern "C" {

void qm_buf_free(char* p) { free(p); }

int qm_sam_header(const qm_index* ix, char** out, int64_t* out_len) {
  if (!ix || !out || !out_len) return io_fail(QM_E_ARG, "qm_sam_header: null argument");
  qm_index_info info;
  if (qm_index_info_get(ix, &info)) return io_fail(QM_E_ARG, "qm_sam_header: bad index");
  std::string o = "@HD\tVN:1.0\tSO:unknown\n";
  for (int64_t t = 0; t < info.n_txps; ++t) { o += "@SQ\tSN:"; o += qm_index_txp_name(ix, t); o += "\tLN:"; o += std::to_string((long long)qm_index_txp_len(ix, t)); o += '\n'; }
  o += "@PG\tID:rapmap\tPN:rapmap\tVN:0.6.0\n";
  char* b = (char*)malloc(o.size() + 1);
  if (!b) return io_fail(QM_E_NOMEM, "out of memory");
  memcpy(b, o.data(), o.size()); b[o.size()] = 0;
  *out = b; *out_len = (int64_t)o.size();
  return QM_OK;
}

// formatting buffers kept between calls (a batch's worth of SAM text is hundreds of MB: handing it back to the allocator
// means unmapping it, and faulting fresh zeroed pages in again for the next batch)
static std::mutex g_poolMu;
static std::vector<std::vector<Out>> g_pool;
struct PartPool {
  std::vector<Out> parts;
  PartPool() { std::lock_guard<std::mutex> lk(g_poolMu); if (!g_pool.empty()) { parts.swap(g_pool.back()); g_pool.pop_back(); } }
  ~PartPool() { std::lock_guard<std::mutex> lk(g_poolMu); if (g_pool.size() < 2) { g_pool.emplace_back(); g_pool.back().swap(parts); } }
};

// The parts of a batch go out in order from ONE thread.  Writes to one inode serialise on its lock, and contending for it
// costs more than it gains: on this box's tmpfs one thread writes 7.7 GB/s, four or sixteen threads calling pwrite at
// disjoint offsets 3.3 GB/s together, copies into a shared mapping 1-3.5 GB/s (profiles/microbench/tmpfs_write_ceiling.py).
static int write_parts(int fd, const std::vector<Out>& parts) {
  for (auto& p : parts) {
    const char* b = p.b; size_t left = p.n;
    while (left > 0) {
      ssize_t w = ::write(fd, b, left);
      if (w < 0) { if (errno == EINTR) continue; return io_fail(QM_E_IO, "write failed: %s", strerror(errno)); }
      b += w; left -= (size_t)w;
    }
  }
  return QM_OK;
}

static int sam_parts(const qm_index* ix, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                     const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                     const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits, int32_t max_num_hits,
                     int32_t n_threads, std::vector<Out>& parts, Pool* pool = nullptr) {
  if (!ix || !names1 || !name_off1 || !seq1 || !off1 || !hit_offsets || n < 0)
    return io_fail(QM_E_ARG, "qm_sam_records: bad argument");
  const bool paired = seq2 != nullptr;
  if (paired && (!names2 || !name_off2 || !off2)) return io_fail(QM_E_ARG, "qm_sam_records: incomplete mate arrays");
  qm_index_info info;
  if (qm_index_info_get(ix, &info)) return io_fail(QM_E_ARG, "qm_sam_records: bad index");
  SamCtx C; C.maxHits = max_num_hits;
  C.tname.resize((size_t)info.n_txps); C.tnl.resize((size_t)info.n_txps); C.tlen.resize((size_t)info.n_txps);
  for (int64_t t = 0; t < info.n_txps; ++t) { C.tname[(size_t)t] = qm_index_txp_name(ix, t); C.tnl[(size_t)t] = (uint32_t)strlen(C.tname[(size_t)t]); C.tlen[(size_t)t] = qm_index_txp_len(ix, t); }
  for (int64_t i = 0, e = hit_offsets[n]; i < e; ++i) if ((int64_t)hits[i].tid >= info.n_txps) return io_fail(QM_E_ARG, "qm_sam_records: hit with an out-of-range transcript id");
  int T = std::max(1, std::min<int>(n_threads, (int)((n + 4095) / 4096)));
  if (parts.size() < (size_t)T) parts.resize((size_t)T);
  for (auto& p : parts) p.n = 0;
  auto work = [&](int t) {
    Out& o = parts[(size_t)t];
    std::string r1, r2;
    const int64_t b = n * t / T, e = n * (t + 1) / T;
    o.need((size_t)(e - b) * 512 + 4096);
    for (int64_t u = b; u < e; ++u) {
      const qm_hit* h = hits + hit_offsets[u];
      const int64_t nh = hit_offsets[u + 1] - hit_offsets[u];
      if (paired)
        format_pair(C, names1 + name_off1[u], name_off1[u + 1] - name_off1[u], seq1 + off1[u], off1[u + 1] - off1[u],
                    names2 + name_off2[u], name_off2[u + 1] - name_off2[u], seq2 + off2[u], off2[u + 1] - off2[u], h, nh, o, r1, r2);
      else
        format_single(C, names1 + name_off1[u], name_off1[u + 1] - name_off1[u], seq1 + off1[u], off1[u + 1] - off1[u], h, nh, o, r1);
    }
  };
  if (T == 1) work(0);
  else if (pool) pool->run(T, work);
  else { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
  return QM_OK;
}

int qm_sam_records(const qm_index* ix, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                   const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                   const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits, int32_t max_num_hits,
                   int32_t n_threads, char** out, int64_t* out_len) {
  if (!out || !out_len) return io_fail(QM_E_ARG, "qm_sam_records: bad argument");
  std::vector<Out> parts;
  int rc = sam_parts(ix, n, names1, name_off1, seq1, off1, names2, name_off2, seq2, off2, hit_offsets, hits, max_num_hits, n_threads, parts);
  if (rc) return rc;
  size_t tot = 0; for (auto& p : parts) tot += p.n;
  char* b = (char*)malloc(tot + 1);
  if (!b) return io_fail(QM_E_NOMEM, "out of memory");
  size_t at = 0; for (auto& p : parts) { if (p.n) memcpy(b + at, p.b, p.n); at += p.n; }
  b[tot] = 0;
  *out = b; *out_len = (int64_t)tot;
  return QM_OK;
}

int qm_sam_write(const qm_index* ix, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                 const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                 const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits, int32_t max_num_hits,
                 int32_t n_threads, int fd, int64_t* bytes_written) {
  PartPool pool;                                   // formatting buffers live across calls: no fresh pages per batch
  std::vector<Out>& parts = pool.parts;
  int rc = sam_parts(ix, n, names1, name_off1, seq1, off1, names2, name_off2, seq2, off2, hit_offsets, hits, max_num_hits, n_threads, parts);
  if (rc) return rc;
  int64_t tot = 0;
  for (auto& p : parts) tot += (int64_t)p.n;
  if ((rc = write_parts(fd, parts))) return rc;
  if (bytes_written) *bytes_written = tot;
  return QM_OK;
}

/* ---- SAM writer: formatting of batch i+1 overlaps the write of batch i ---- */
// one gzip member (RFC 1952) per formatter part: the members are compressed side by side by the formatter's workers and
// written one after the other -- a concatenation of members is a valid .gz file (zlib's gzread, gzip -d and zcat read it
// as one stream), which is what makes `-x` parallel; the reference deflates its whole output in the writer thread
static int gz_member(const Out& in, Out& z, int level) {
  z.n = 0;
  if (in.n == 0) return QM_OK;
  z_stream zs; memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return io_fail(QM_E_IO, "deflateInit2 failed");
  size_t done = 0;
  while (done < in.n) {                                   // zlib counts in 32 bits: feed at most 1 GB at a time
    const size_t chunk = std::min(in.n - done, (size_t)1 << 30);
    zs.next_in = (Bytef*)(in.b + done); zs.avail_in = (uInt)chunk; done += chunk;
    const int flush = done == in.n ? Z_FINISH : Z_NO_FLUSH;
    int rc;
    do {
      z.need(std::max<size_t>((size_t)deflateBound(&zs, (uLong)zs.avail_in) + 64, 1 << 16));
      zs.next_out = (Bytef*)(z.b + z.n); zs.avail_out = (uInt)std::min<size_t>(z.cap - z.n, (size_t)1 << 30);
      const uInt before = zs.avail_out;
      rc = deflate(&zs, flush);
      z.n += before - zs.avail_out;
      if (rc == Z_STREAM_ERROR) { deflateEnd(&zs); return io_fail(QM_E_IO, "deflate failed"); }
    } while (zs.avail_in > 0 || (flush == Z_FINISH && rc != Z_STREAM_END));
  }
  deflateEnd(&zs);
  return QM_OK;
}

struct qm_sam_writer {
  const qm_index* ix = nullptr; int fd = -1; int threads = 1; int32_t maxHits = 0;
  bool gzip = false; int level = 1;
  std::vector<Out> bufs[2];
  std::vector<Out> zbufs[2];        // gzip members of the parts (QM_SAM_GZIP)
  Pool* pool = nullptr;             // the formatter's workers
  int state[2] = {0, 0};            // 0 free, 1 formatted (waits for the writer)
  int fill = 0, drain = 0;
  bool stop = false; int err = 0; char errmsg[256] = ""; int64_t bytes = 0;
  std::mutex mu; std::condition_variable cv; std::thread wt;
};
static void sam_writer_loop(qm_sam_writer* w) {
  while (true) {
    int k;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->state[w->drain] == 1 || w->stop; });
      if (w->state[w->drain] != 1) return;               // stop, nothing left
      k = w->drain;
    }
    std::vector<Out>& outp = w->gzip ? w->zbufs[k] : w->bufs[k];
    int rc = w->err ? 0 : write_parts(w->fd, outp);         // after a failure the remaining batches are dropped
    std::unique_lock<std::mutex> lk(w->mu);
    if (rc) { w->err = rc; snprintf(w->errmsg, sizeof(w->errmsg), "%s", qm_io_last_error()); }
    else if (!w->err) for (auto& p : outp) w->bytes += (int64_t)p.n;
    w->state[k] = 0; w->drain ^= 1;
    w->cv.notify_all();
  }
}
int qm_sam_writer_open(const qm_index* ix, int fd, int32_t max_num_hits, int32_t n_threads, qm_sam_writer** out) {
  return qm_sam_writer_open_ex(ix, fd, max_num_hits, n_threads, 0, out);
}
// queue what is in bufs[k] (compressing it first with QM_SAM_GZIP)
static int sam_writer_submit(qm_sam_writer* w, int k) {
  if (w->gzip) {
    std::vector<Out>& P = w->bufs[k]; std::vector<Out>& Z = w->zbufs[k];
    if (Z.size() < P.size()) Z.resize(P.size());
    for (auto& z : Z) z.n = 0;
    std::vector<int> rcs(P.size(), 0);
    const std::function<void(int)> fn = [&](int t) { rcs[(size_t)t] = gz_member(P[(size_t)t], Z[(size_t)t], w->level); };
    w->pool->run((int)P.size(), fn);
    for (int r : rcs) if (r) return r;
  }
  std::unique_lock<std::mutex> lk(w->mu);
  w->state[k] = 1; w->fill ^= 1;
  w->cv.notify_all();
  return QM_OK;
}
int qm_sam_writer_open_ex(const qm_index* ix, int fd, int32_t max_num_hits, int32_t n_threads, uint32_t flags, qm_sam_writer** out) {
  if (!ix || fd < 0 || !out) return io_fail(QM_E_ARG, "qm_sam_writer_open: bad argument");
  qm_sam_writer* w = new qm_sam_writer();
  w->ix = ix; w->fd = fd; w->threads = n_threads > 0 ? n_threads : 1; w->maxHits = max_num_hits;
  w->gzip = (flags & QM_SAM_GZIP) != 0;
  w->level = (int)((flags >> 8) & 15u); if (w->level < 1 || w->level > 9) w->level = 1;
  w->pool = new Pool(w->threads);
  w->wt = std::thread(sam_writer_loop, w);
  *out = w;
  return QM_OK;
}
int qm_sam_writer_put(qm_sam_writer* w, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1, const int64_t* off1,
                      const char* names2, const int64_t* name_off2, const char* seq2, const int64_t* off2,
                      const int64_t* hit_offsets, const qm_hit* hits) {
  if (!w) return io_fail(QM_E_ARG, "qm_sam_writer_put: bad argument");
  int k;
  {
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->state[w->fill] == 0; });
    if (w->err) return io_fail(w->err, "%s", w->errmsg);
    k = w->fill;
  }
  int rc = sam_parts(w->ix, n, names1, name_off1, seq1, off1, names2, name_off2, seq2, off2, hit_offsets, hits, w->maxHits, w->threads, w->bufs[k], w->pool);
  if (rc) return rc;
  return sam_writer_submit(w, k);
}
int qm_sam_writer_header(qm_sam_writer* w) {
  if (!w) return io_fail(QM_E_ARG, "qm_sam_writer_header: bad argument");
  char* h = nullptr; int64_t hn = 0;
  int rc = qm_sam_header(w->ix, &h, &hn);
  if (rc) return rc;
  int k;
  {
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->state[w->fill] == 0; });
    if (w->err) { qm_buf_free(h); return io_fail(w->err, "%s", w->errmsg); }
    k = w->fill;
  }
  std::vector<Out>& P = w->bufs[k];
  if (P.empty()) P.resize(1);
  for (auto& p : P) p.n = 0;
  P[0].need((size_t)hn + 1); P[0].raw(h, (size_t)hn);
  qm_buf_free(h);
  return sam_writer_submit(w, k);
}
int qm_sam_writer_close(qm_sam_writer* w, int64_t* bytes_written) {
  if (!w) return QM_OK;
  {
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->state[0] == 0 && w->state[1] == 0; });
    w->stop = true; w->cv.notify_all();
  }
  w->wt.join();
  const int rc = w->err;
  if (bytes_written) *bytes_written = w->bytes;
  if (rc) io_fail(rc, "%s", w->errmsg);
  delete w->pool;
  delete w;
  return rc;
}

}  // extern "C"
