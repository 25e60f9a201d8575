// qm_io.cpp -- host-side callers of the hot path:
//   * qm_reader_*    FASTA/FASTQ(.gz) ingest into the packed (bytes, offsets[n+1]) batches qm_map_pairs takes.
//                    Replaces the reference's single kseq producer + per-record std::string queue
//                    (include/FastxParser.hpp:62-66, src/FastxParser.cpp:229-328): plain files are mmap'd and
//                    parsed by n_threads workers on disjoint byte ranges, .gz files are inflated by one
//                    thread per file; qualities are dropped like the reference's parser does.
//   * qm_sam_*       SAM text for a mapped batch, same bytes as `rapmap quasimap -o`
//                    (writeSAMHeader include/RapMapUtils.hpp:97-115, writeAlignmentsToStream
//                    src/RapMapUtils.cpp:198-588, writeUnalignedPairToStream :137-196, getSamFlags /
//                    adjustOverhang include/RapMapUtils.hpp:687-810), formatted by n_threads workers.
// Plain C++ (no HIP); part of libqmap_mi355.so.
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
#include <string>
#include <thread>
#include <vector>

#include "../../include/qmap_mi355.h"

static thread_local char g_ioerr[512] = "";
static int io_fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_ioerr, sizeof(g_ioerr), fmt, ap); va_end(ap);
  return code;
}
extern "C" const char* qm_io_last_error(void) { return g_ioerr; }

namespace {

// ------------------------------------------------------------------ parsed records of one file
struct Records {
  std::vector<char> seq, names;
  std::vector<int64_t> off{0}, noff{0};
  int64_t n() const { return (int64_t)off.size() - 1; }
  void clear() { seq.clear(); names.clear(); off.assign(1, 0); noff.assign(1, 0); }
  void push(const char* nm, size_t nl, const char* s, size_t sl) {
    names.insert(names.end(), nm, nm + nl); noff.push_back((int64_t)names.size());
    seq.insert(seq.end(), s, s + sl); off.push_back((int64_t)seq.size());
  }
};

static inline const char* eol(const char* p, const char* e) {
  const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
  return q ? q : e;
}
static inline size_t rstrip(const char* b, const char* e) { while (e > b && (e[-1] == '\r' || e[-1] == '\n')) --e; return (size_t)(e - b); }

// Parse complete records in [p, e); returns the first byte not consumed (start of an incomplete record).
// FASTQ: 4-line records; FASTA: header + one or more sequence lines (joined).  final: the buffer ends the file.
static const char* parse_block(const char* p, const char* e, bool final, Records& R, bool& bad) {
  while (p < e) {
    while (p < e && (*p == '\n' || *p == '\r')) ++p;
    if (p >= e) break;
    const char* rec = p;
    const char* l1 = eol(p, e);
    if (l1 == e && !final) return rec;
    if (*p == '@') {
      if (l1 == e) { bad = true; return rec; }
      const char* s = l1 + 1; const char* l2 = eol(s, e);
      if (l2 == e) { if (!final) return rec; bad = true; return rec; }
      const char* pl = l2 + 1; const char* l3 = eol(pl, e);
      if (l3 == e) { if (!final) return rec; bad = true; return rec; }
      const char* q = l3 + 1; const char* l4 = eol(q, e);
      if (l4 == e && !final) return rec;
      if (pl >= e || *pl != '+') { bad = true; return rec; }
      R.push(p + 1, rstrip(p + 1, l1), s, rstrip(s, l2));
      p = l4 < e ? l4 + 1 : e;
    } else if (*p == '>') {
      const char* s = l1 < e ? l1 + 1 : e;
      size_t nl = rstrip(p + 1, l1);
      size_t seq0 = R.seq.size();
      const char* c = s;
      while (c < e && *c != '>') {
        const char* le = eol(c, e);
        if (le == e && !final) { R.seq.resize(seq0); return rec; }
        R.seq.insert(R.seq.end(), c, c + rstrip(c, le));
        c = le < e ? le + 1 : e;
      }
      if (c >= e && !final) { R.seq.resize(seq0); return rec; }
      R.names.insert(R.names.end(), p + 1, p + 1 + nl); R.noff.push_back((int64_t)R.names.size());
      R.off.push_back((int64_t)R.seq.size());
      p = c;
    } else { bad = true; return rec; }
  }
  return e;
}

// start of the first FASTQ/FASTA record at or after p (p may be mid-record): a line starting with '@' whose
// line-after-next starts with '+' (a quality line may itself start with '@'), or any line starting with '>'.
static const char* sync_record(const char* base, const char* p, const char* e, bool fastq) {
  if (p == base) return p;
  const char* q = eol(p - 1, e);             // go to the next line start
  p = q < e ? q + 1 : e;
  while (p < e) {
    if (!fastq) { if (*p == '>') return p; }
    else if (*p == '@') {
      const char* l1 = eol(p, e); if (l1 == e) return e;
      const char* l2 = eol(l1 + 1, e); if (l2 == e) return e;
      if (l2 + 1 < e && l2[1] == '+') {
        // the line after '+' must be a quality line as long as the sequence line
        const char* l3 = eol(l2 + 1, e);
        const char* l4 = l3 < e ? eol(l3 + 1, e) : e;
        if (l3 < e && rstrip(l3 + 1, l4) == rstrip(l1 + 1, l2)) return p;
      }
    }
    const char* l = eol(p, e);
    p = l < e ? l + 1 : e;
  }
  return e;
}

struct Source {
  std::string path;
  bool gz = false, fastq = true, eof = false, bad = false;
  // plain
  const char* map = nullptr; size_t len = 0, pos = 0;
  // gz
  gzFile gzf = nullptr; std::vector<char> carry;
  Records ready;            // parsed, not yet handed out
  int64_t taken = 0;        // records of `ready` already handed out

  int open(const char* p) {
    path = p;
    int fd = ::open(p, O_RDONLY);
    if (fd < 0) return io_fail(QM_E_IO, "cannot open %s", p);
    unsigned char magic[2] = {0, 0};
    ssize_t got = ::read(fd, magic, 2);
    struct stat st; fstat(fd, &st);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
      ::close(fd); gz = true;
      gzf = gzopen(p, "rb");
      if (!gzf) return io_fail(QM_E_IO, "cannot gzopen %s", p);
      gzbuffer(gzf, 1 << 20);
      fastq = true;          // decided at the first block
      return 0;
    }
    len = (size_t)st.st_size;
    if (len > 0) {
      map = (const char*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
      if (map == MAP_FAILED) { map = nullptr; ::close(fd); return io_fail(QM_E_IO, "cannot mmap %s", p); }
      madvise((void*)map, len, MADV_SEQUENTIAL);
      fastq = map[0] != '>';
    } else eof = true;
    ::close(fd);
    return 0;
  }
  void close() { if (map) munmap((void*)map, len); map = nullptr; if (gzf) gzclose(gzf); gzf = nullptr; }

  // parse roughly `bytes` more input into `ready`
  void fill(size_t bytes, int nthreads) {
    if (taken > 0) {          // drop what was handed out
      Records r; int64_t n = ready.n();
      if (taken < n) {
        r.seq.assign(ready.seq.begin() + ready.off[taken], ready.seq.end());
        r.names.assign(ready.names.begin() + ready.noff[taken], ready.names.end());
        for (int64_t i = taken + 1; i <= n; ++i) { r.off.push_back(ready.off[i] - ready.off[taken]); r.noff.push_back(ready.noff[i] - ready.noff[taken]); }
      }
      ready = std::move(r); taken = 0;
    }
    if (eof || bad) return;
    if (gz) {
      size_t old = carry.size();
      carry.resize(old + bytes);
      int got = gzread(gzf, carry.data() + old, (unsigned)bytes);
      if (got < 0) { bad = true; return; }
      carry.resize(old + (size_t)got);
      bool final = (size_t)got < bytes;
      if (final) eof = true;
      if (carry.empty()) return;
      fastq = carry[0] != '>';
      const char* b = carry.data(); const char* e = b + carry.size();
      const char* rest = parse_block(b, e, final, ready, bad);
      carry.erase(carry.begin(), carry.begin() + (rest - b));
      return;
    }
    size_t end = std::min(len, pos + bytes);
    const bool final = end == len;
    const char* b = map + pos; const char* e = map + end;
    int T = std::max(1, std::min(nthreads, (int)((end - pos) >> 20)));
    if (T == 1) {
      const char* rest = parse_block(b, e, final, ready, bad);
      pos = (size_t)(rest - map);
    } else {
      // disjoint byte ranges, each starting at a record boundary; the last range's tail is left for the next call
      std::vector<const char*> cut(T + 1);
      cut[0] = b; cut[T] = e;
      for (int t = 1; t < T; ++t) cut[t] = sync_record(b, b + (size_t)(e - b) * t / T, e, fastq);
      for (int t = 1; t <= T; ++t) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
      std::vector<Records> parts(T); std::vector<const char*> rest(T); std::vector<char> badv(T, 0);
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() { bool bd = false; rest[t] = parse_block(cut[t], cut[t + 1], t == T - 1 ? final : true, parts[t], bd); badv[t] = bd; });
      for (auto& x : th) x.join();
      for (int t = 0; t < T; ++t) {
        if (badv[t] || (t < T - 1 && rest[t] != cut[t + 1])) { bad = true; break; }
        const Records& r = parts[t];
        int64_t so = (int64_t)ready.seq.size(), no = (int64_t)ready.names.size();
        ready.seq.insert(ready.seq.end(), r.seq.begin(), r.seq.end());
        ready.names.insert(ready.names.end(), r.names.begin(), r.names.end());
        for (int64_t i = 1; i <= r.n(); ++i) { ready.off.push_back(so + r.off[i]); ready.noff.push_back(no + r.noff[i]); }
      }
      pos = (size_t)(rest[T - 1] - map);
    }
    if (pos >= len) eof = true;
  }
  int64_t avail() const { return ready.n() - taken; }
};

}  // namespace

struct qm_reader {
  Source src[2]; int nsrc = 0; int nthreads = 1;
  Records out[2];
};

extern "C" {

int qm_reader_open(const char* path1, const char* path2, int32_t n_threads, qm_reader** out) {
  if (!path1 || !out) return io_fail(QM_E_ARG, "qm_reader_open: null argument");
  qm_reader* r = new qm_reader();
  r->nthreads = n_threads > 0 ? n_threads : 1;
  r->nsrc = path2 ? 2 : 1;
  int rc = r->src[0].open(path1);
  if (!rc && path2) rc = r->src[1].open(path2);
  if (rc) { r->src[0].close(); r->src[1].close(); delete r; return rc; }
  *out = r;
  return QM_OK;
}

void qm_reader_close(qm_reader* r) {
  if (!r) return;
  r->src[0].close(); r->src[1].close();
  delete r;
}

int qm_reader_next(qm_reader* r, int64_t max_units, int64_t* n_units, const char** seq1, const int64_t** off1,
                   const char** names1, const int64_t** name_off1, const char** seq2, const int64_t** off2,
                   const char** names2, const int64_t** name_off2) {
  if (!r || !n_units || max_units <= 0) return io_fail(QM_E_ARG, "qm_reader_next: bad argument");
  const size_t block = (size_t)64 << 20;
  for (int s = 0; s < r->nsrc; ++s) {
    Source& S = r->src[s];
    while (S.avail() < max_units && !S.eof && !S.bad) {
      if (r->nsrc == 2 && s == 0) {
        // both files advance together: parse them concurrently
        Source& S2 = r->src[1];
        std::thread t2([&]() { if (S2.avail() < max_units && !S2.eof && !S2.bad) S2.fill(block, std::max(1, r->nthreads / 2)); });
        S.fill(block, std::max(1, r->nthreads / 2));
        t2.join();
      } else S.fill(block, r->nthreads);
    }
    if (S.bad) return io_fail(QM_E_FORMAT, "%s: malformed FASTA/FASTQ record", S.path.c_str());
  }
  int64_t n = std::min(max_units, r->src[0].avail());
  if (r->nsrc == 2) {
    n = std::min(n, r->src[1].avail());
    if (n == 0 && (r->src[0].avail() > 0) != (r->src[1].avail() > 0) && r->src[0].eof && r->src[1].eof)
      return io_fail(QM_E_FORMAT, "paired files have different numbers of records");
  }
  for (int s = 0; s < r->nsrc; ++s) {
    Source& S = r->src[s]; Records& O = r->out[s];
    O.clear();
    const int64_t a = S.taken, b = S.taken + n;
    O.seq.assign(S.ready.seq.begin() + S.ready.off[a], S.ready.seq.begin() + S.ready.off[b]);
    O.names.assign(S.ready.names.begin() + S.ready.noff[a], S.ready.names.begin() + S.ready.noff[b]);
    O.off.resize((size_t)n + 1); O.noff.resize((size_t)n + 1);
    for (int64_t i = 0; i <= n; ++i) { O.off[i] = S.ready.off[a + i] - S.ready.off[a]; O.noff[i] = S.ready.noff[a + i] - S.ready.noff[a]; }
    S.taken = b;
  }
  *n_units = n;
  if (seq1) *seq1 = r->out[0].seq.data();
  if (off1) *off1 = r->out[0].off.data();
  if (names1) *names1 = r->out[0].names.data();
  if (name_off1) *name_off1 = r->out[0].noff.data();
  if (r->nsrc == 2) {
    if (seq2) *seq2 = r->out[1].seq.data();
    if (off2) *off2 = r->out[1].off.data();
    if (names2) *names2 = r->out[1].names.data();
    if (name_off2) *name_off2 = r->out[1].noff.data();
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
static void read_name(const char* nm, int64_t n, bool trimMate, std::string& out) {   // src/RapMapUtils.cpp:334-351
  const char* sp = (const char*)memchr(nm, ' ', (size_t)n);
  int64_t l = sp ? sp - nm : n;
  if (trimMate && l > 2 && nm[l - 2] == '/') l -= 2;
  out.assign(nm, (size_t)l);
}
static void app(std::string& o, long long v) { char b[24]; int n = snprintf(b, sizeof(b), "%lld", v); o.append(b, (size_t)n); }
// include/RapMapUtils.hpp:687-711
static int32_t adjust_overhang(int32_t pos, uint32_t readLen, int64_t txpLen, std::string& cigar) {
  cigar.clear();
  const long long rl = readLen;
  if (pos + rl < 0) { app(cigar, rl); cigar += 'S'; return 0; }
  if (pos < 0) { long long match = rl + pos, clip = rl - match; app(cigar, clip); cigar += 'S'; app(cigar, match); cigar += 'M'; return 0; }
  if (pos > txpLen) { app(cigar, rl); cigar += 'S'; return pos; }
  if (pos + rl > txpLen) { long long match = txpLen - pos, clip = rl - match; app(cigar, match); cigar += 'M'; app(cigar, clip); cigar += 'S'; return pos; }
  app(cigar, rl); cigar += 'M'; return pos;
}
static void tags(std::string& o, long long nh, long long hi, long long as) {
  o += "\tNH:i:"; app(o, nh); o += "\tHI:i:"; app(o, hi); o += "\tAS:i:"; app(o, as); o += '\n';
}

struct SamCtx { const qm_index* ix; int maxHits; };

static void format_pair(const SamCtx& C, const char* nm1, int64_t nl1, const char* s1, int64_t l1, const char* nm2,
                        int64_t nl2, const char* s2, int64_t l2, const qm_hit* h, int64_t nh, std::string& o,
                        std::string& n1, std::string& n2, std::string& rev1, std::string& rev2, std::string& c1, std::string& c2) {
  read_name(nm1, nl1, true, n1); read_name(nm2, nl2, true, n2);
  if (nh == 0 || nh > C.maxHits) {                       // writeUnalignedPairToStream
    for (int m = 0; m < 2; ++m) {
      o += m == 0 ? n1 : n2; o += '\t'; app(o, (0x1 | 0x4 | 0x8 | (m == 0 ? 0x40 : 0x80)));
      o += "\t*\t0\t255\t*\t*\t*\t0\t"; o.append(m == 0 ? s1 : s2, (size_t)(m == 0 ? l1 : l2)); o += "\t*\tNH:i:0\tHI:i:0\tAS:i:0\n";
    }
    return;
  }
  bool have1 = false, have2 = false;
  for (int64_t i = 0; i < nh; ++i) {
    const qm_hit& q = h[i];
    const char* tname = qm_index_txp_name(C.ix, q.tid);
    const int64_t tlen = qm_index_txp_len(C.ix, q.tid);
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
      o += n1; o += '\t'; app(o, f1); o += '\t'; o += tname; o += '\t'; app(o, pos + 1LL); o += "\t1\t"; o += c1; o += "\t=\t";
      app(o, mpos + 1LL); o += '\t'; app(o, r1First ? frag : -frag); o += '\t';
      if (fwd) o.append(s1, (size_t)l1); else o += rev1;
      o += "\t*"; tags(o, nh, i + 1, q.aln_score);
      o += n2; o += '\t'; app(o, f2); o += '\t'; o += tname; o += '\t'; app(o, mpos + 1LL); o += "\t1\t"; o += c2; o += "\t=\t";
      app(o, pos + 1LL); o += '\t'; app(o, r1First ? -frag : frag); o += '\t';
      if (mfwd) o.append(s2, (size_t)l2); else o += rev2;
      o += "\t*"; tags(o, nh, i + 1, q.aln_score);
    } else {
      const bool left = q.mate_status == 1;
      const std::string& an = left ? n1 : n2; const std::string& un = left ? n2 : n1;
      const int afl = left ? f1 : f2, ufl = left ? f2 : f1;
      int32_t pos = adjust_overhang(q.pos, q.read_len, tlen, c1);
      o += an; o += '\t'; app(o, afl); o += '\t'; o += tname; o += '\t'; app(o, pos + 1LL); o += "\t1\t"; o += c1; o += "\t=\t";
      app(o, pos + 1LL); o += "\t0\t";
      if (fwd) { if (left) o.append(s1, (size_t)l1); else o.append(s2, (size_t)l2); }
      else if (left) { if (!have1) { reverse_read(s1, l1, rev1); have1 = true; } o += rev1; }
      else { if (!have2) { reverse_read(s2, l2, rev2); have2 = true; } o += rev2; }
      o += "\t*"; tags(o, nh, i + 1, q.aln_score);
      o += un; o += '\t'; app(o, ufl); o += '\t'; o += tname; o += '\t'; app(o, pos + 1LL); o += "\t0\t*\t=\t"; app(o, pos + 1LL); o += "\t0\t";
      if (left) o.append(s2, (size_t)l2); else o.append(s1, (size_t)l1);
      o += "\t*"; tags(o, nh, i + 1, q.aln_score);
    }
  }
}

// single-end records (src/RapMapUtils.cpp:198-311): MAPQ 255, 0x10 for rc, 0x900 on secondary hits; the
// single-end writer strips the name at the first blank only.
static void format_single(const SamCtx& C, const char* nm, int64_t nl, const char* s, int64_t l, const qm_hit* h,
                          int64_t nh, std::string& o, std::string& n1, std::string& rev, std::string& c1) {
  read_name(nm, nl, false, n1);
  if (nh == 0) {
    o += n1; o += "\t4\t*\t0\t255\t*\t*\t0\t0\t"; o.append(s, (size_t)l); o += "\t*\tNH:i:0\tHI:i:0\tAS:i:0\n";
    return;
  }
  bool have = false;
  for (int64_t i = 0; i < nh; ++i) {
    const qm_hit& q = h[i];
    int fl = q.fwd ? 0 : 0x10;
    if (i != 0) fl |= 0x900;
    int32_t pos = adjust_overhang(q.pos, q.read_len, qm_index_txp_len(C.ix, q.tid), c1);
    o += n1; o += '\t'; app(o, fl); o += '\t'; o += qm_index_txp_name(C.ix, q.tid); o += '\t'; app(o, pos + 1LL);
    o += "\t255\t"; o += c1; o += "\t*\t0\t"; app(o, (long long)q.frag_len); o += '\t';
    if (q.fwd) o.append(s, (size_t)l); else { if (!have) { reverse_read(s, l, rev); have = true; } o += rev; }
    o += "\t*"; tags(o, nh, i + 1, q.aln_score);
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
  for (int64_t t = 0; t < info.n_txps; ++t) { o += "@SQ\tSN:"; o += qm_index_txp_name(ix, t); o += "\tLN:"; app(o, qm_index_txp_len(ix, t)); o += '\n'; }
  o += "@PG\tID:rapmap\tPN:rapmap\tVN:0.6.0\n";
  char* b = (char*)malloc(o.size() + 1);
  if (!b) return io_fail(QM_E_NOMEM, "out of memory");
  memcpy(b, o.data(), o.size()); b[o.size()] = 0;
  *out = b; *out_len = (int64_t)o.size();
  return QM_OK;
}

static int sam_parts(const qm_index* ix, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                     const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                     const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits, int32_t max_num_hits,
                     int32_t n_threads, std::vector<std::string>& parts) {
  if (!ix || !names1 || !name_off1 || !seq1 || !off1 || !hit_offsets || n < 0)
    return io_fail(QM_E_ARG, "qm_sam_records: bad argument");
  const bool paired = seq2 != nullptr;
  if (paired && (!names2 || !name_off2 || !off2)) return io_fail(QM_E_ARG, "qm_sam_records: incomplete mate arrays");
  int T = std::max(1, std::min<int>(n_threads, (int)((n + 4095) / 4096)));
  parts.assign((size_t)T, std::string());
  SamCtx C{ix, max_num_hits};
  auto work = [&](int t) {
    std::string& o = parts[(size_t)t];
    std::string n1, n2, r1, r2, c1, c2;
    const int64_t b = n * t / T, e = n * (t + 1) / T;
    o.reserve((size_t)(e - b) * 420);
    for (int64_t u = b; u < e; ++u) {
      const qm_hit* h = hits + hit_offsets[u];
      const int64_t nh = hit_offsets[u + 1] - hit_offsets[u];
      if (paired)
        format_pair(C, names1 + name_off1[u], name_off1[u + 1] - name_off1[u], seq1 + off1[u], off1[u + 1] - off1[u],
                    names2 + name_off2[u], name_off2[u + 1] - name_off2[u], seq2 + off2[u], off2[u + 1] - off2[u], h, nh, o,
                    n1, n2, r1, r2, c1, c2);
      else
        format_single(C, names1 + name_off1[u], name_off1[u + 1] - name_off1[u], seq1 + off1[u], off1[u + 1] - off1[u], h, nh, o, n1, r1, c1);
    }
  };
  if (T == 1) work(0);
  else { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
  return QM_OK;
}

int qm_sam_records(const qm_index* ix, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                   const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                   const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits, int32_t max_num_hits,
                   int32_t n_threads, char** out, int64_t* out_len) {
  if (!out || !out_len) return io_fail(QM_E_ARG, "qm_sam_records: bad argument");
  std::vector<std::string> parts;
  int rc = sam_parts(ix, n, names1, name_off1, seq1, off1, names2, name_off2, seq2, off2, hit_offsets, hits, max_num_hits, n_threads, parts);
  if (rc) return rc;
  size_t tot = 0; for (auto& p : parts) tot += p.size();
  char* b = (char*)malloc(tot + 1);
  if (!b) return io_fail(QM_E_NOMEM, "out of memory");
  size_t at = 0; for (auto& p : parts) { memcpy(b + at, p.data(), p.size()); at += p.size(); }
  b[tot] = 0;
  *out = b; *out_len = (int64_t)tot;
  return QM_OK;
}

int qm_sam_write(const qm_index* ix, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                 const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                 const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits, int32_t max_num_hits,
                 int32_t n_threads, int fd, int64_t* bytes_written) {
  std::vector<std::string> parts;
  int rc = sam_parts(ix, n, names1, name_off1, seq1, off1, names2, name_off2, seq2, off2, hit_offsets, hits, max_num_hits, n_threads, parts);
  if (rc) return rc;
  int64_t tot = 0;
  for (auto& p : parts) {
    const char* b = p.data(); size_t left = p.size();
    while (left > 0) {
      ssize_t w = ::write(fd, b, left);
      if (w < 0) return io_fail(QM_E_IO, "write failed");
      b += w; left -= (size_t)w; tot += w;
    }
  }
  if (bytes_written) *bytes_written = tot;
  return QM_OK;
}

}  // extern "C"
