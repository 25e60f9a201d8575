// qm_ingest.cpp -- FASTA/FASTQ(.gz) files -> packed batches (bytes, offsets[n+1]) of qm_map_pairs / qm_map_reads, at the
// rate the mapping kernels consume them.
//
// The reference feeds its mapping threads from ONE kseq producer thread that hands every record over as std::strings
// (src/FastxParser.cpp:229-328, used at src/RapMapSAMapper.cpp:853,869-871).  Here nothing about ingest is serial but the
// bookkeeping:
//   * a plain file is cut at fixed byte offsets into chunks; a chunk starts at the first record boundary at or behind its
//     offset (record resync: sync_record), so any worker can parse any chunk without having seen the bytes before it;
//   * a PARSE task turns one chunk into a table of (name, sequence) positions plus running byte counts -- the only pass
//     that looks for line ends;
//   * chunks complete out of order; as soon as the in-order frontier of both files holds a batch's worth of records the
//     batch gets a slot (buffers from the caller's allocator: pinned host memory for qm_stream) and is cut into COPY tasks,
//     each of which moves the characters of a run of records from the file mapping STRAIGHT into their final place in the
//     slot and writes their offsets (the destination of every record is known from the running counts: no staging copy,
//     no second pass over a finished batch);
//   * workers prefer copy tasks (they finish batches) over parse tasks (they start new ones); several batches are in
//     flight at once, bounded by the number of slots and by a read-ahead limit per file;
//   * .gz input: one inflate thread per file produces blocks that end on a record boundary; they enter the same parse ->
//     copy pipeline, so decompression of block i+1 overlaps parsing and packing of block i.
// Batches leave in input order.  Qualities are dropped, as the reference's parser does.
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
#include <sched.h>
#include <sys/syscall.h>
#include <pthread.h>
#include <unistd.h>
#include <zlib.h>
#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/qmap_mi355.h"
#include "qm_io_internal.h"
#include "qm_pgz.h"
#include "qm_pack.h"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline const char* eol(const char* p, const char* e) {
  const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
  return q ? q : e;
}
inline size_t rstrip(const char* b, const char* e) { while (e > b && (e[-1] == '\r' || e[-1] == '\n')) --e; return (size_t)(e - b); }

// true when [a, a + len) holds no '\n'.  16 bytes at a time (SSE2: every x86-64 has it), the tail by an overlapping load when the
// range is at least 16 long.  The FASTQ fast path of parse_chunk asks this about every line of a record whose layout it
// predicts from the record before: four libc memchr calls per 212-byte record were what bounded a parse task at ~2.6 GB/s.
inline bool no_newline(const char* a, size_t len) {
  if (len < 16) { for (size_t i = 0; i < len; ++i) if (a[i] == '\n') return false; return true; }
  const __m128i nl = _mm_set1_epi8('\n');
  __m128i acc = _mm_setzero_si128();
  size_t i = 0;
  for (; i + 16 <= len; i += 16) acc = _mm_or_si128(acc, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(a + i)), nl));
  if (i < len) acc = _mm_or_si128(acc, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(a + len - 16)), nl));
  return _mm_movemask_epi8(acc) == 0;
}

// Start of the first FASTQ/FASTA record at or after p (p may be mid-record): a line starting with '@' whose
// line-after-next starts with '+' and whose quality line is as long as its sequence line (a quality line may
// itself start with '@'), or any line starting with '>'.  *hitEnd: the search looked at the end of the buffer -- with more
// bytes behind e the answer could be different (callers that see only a window of the file widen it and ask again).
const char* sync_record(const char* base, const char* p, const char* e, bool fastq, bool* hitEnd = nullptr) {
  bool dummy; bool& hit = hitEnd ? *hitEnd : dummy; hit = false;
  if (p <= base) return base;
  if (p >= e) { hit = true; return e; }
  const char* q = eol(p - 1, e);             // go to the next line start
  p = q < e ? q + 1 : e;
  while (p < e) {
    if (!fastq) { if (*p == '>') return p; }
    else if (*p == '@') {
      const char* l1 = eol(p, e); if (l1 == e) break;
      const char* l2 = eol(l1 + 1, e); if (l2 == e) break;
      if (l2 + 1 >= e) break;
      if (l2[1] == '+') {
        const char* l3 = eol(l2 + 1, e);
        if (l3 == e) break;
        const char* l4 = eol(l3 + 1, e);
        if (l4 == e) hit = true;
        if (rstrip(l3 + 1, l4) == rstrip(l1 + 1, l2)) return p;
        if (l4 == e) break;
      }
    }
    const char* l = eol(p, e);
    p = l < e ? l + 1 : e;
  }
  hit = true;
  return e;
}

// One record of a parsed chunk: where its name and its sequence start (relative to the chunk's first byte; ARENA: the
// sequence was joined from several FASTA lines and sits in the chunk's arena) and how many sequence / name bytes precede it
// in the chunk.  recs[n] closes the table with the totals.
struct Rec { uint32_t nameOff, seqOff, cumSeq, cumName; };
constexpr uint32_t ARENA = 0x80000000u;

struct Chunk {
  int src = 0; int64_t id = 0;
  const char* base = nullptr; const char* end = nullptr;
  std::vector<Rec> recs; uint32_t n = 0;
  std::vector<char> arena;           // joined multi-line FASTA sequences
  std::vector<char> data;            // gz: the decompressed bytes this chunk owns
  int pending = 0;                   // copy tasks still reading from it
  bool assigned = false;             // every record belongs to a batch
  bool bad = false;
};

// Parse the whole records of [b, e) -- the range starts on a record boundary and ends on one (or at the end of the file).
// FASTQ: 4-line records; FASTA: header + sequence lines, a multi-line sequence is joined in the arena.
bool parse_chunk(Chunk& C, bool fastq) {
  const char* b = C.base; const char* e = C.end; const char* p = b;
  std::vector<Rec>& R = C.recs;
  R.clear(); C.arena.clear();
  if ((size_t)(e - b) >= (size_t)ARENA) return false;
  R.reserve((size_t)(e - b) / (fastq ? 160 : 400) + 16);
  uint32_t cs = 0, cn = 0;
  // layout of the previous FASTQ record when it was "plain" (no blank lines before it, no '\r'): lengths of its four lines
  // including their '\n'.  Nearly every record of a file repeats it (fixed-length reads; the name grows a digit now and then).
  size_t n1 = 0, n2 = 0, n3 = 0, n4 = 0;
  while (p < e) {
    if (fastq && n1 && (size_t)(e - p) >= n1 + n2 + n3 + n4) {
      // The record at p has the previous one's layout exactly when: '@' starts it, each predicted line end is a '\n' that is
      // not preceded by a '\r', the third line starts with '+', and NO other '\n' lies inside the four lines (then the first
      // newline behind each line start is the predicted one, which is all the general path below would establish).
      const char* s = p + n1; const char* pl = s + n2; const char* q = pl + n3; const char* nx = q + n4;
      if (*p == '@' && s[-1] == '\n' && pl[-1] == '\n' && *pl == '+' && q[-1] == '\n' && nx[-1] == '\n' &&
          (n1 < 2 || s[-2] != '\r') && (n2 < 2 || pl[-2] != '\r') && (n4 < 2 || nx[-2] != '\r') &&
          no_newline(p, n1 - 1) && no_newline(s, n2 - 1) && no_newline(pl, n3 - 1) && no_newline(q, n4 - 1)) {
        R.push_back(Rec{(uint32_t)(p + 1 - b), (uint32_t)(s - b), cs, cn});
        cs += (uint32_t)(n2 - 1); cn += (uint32_t)(n1 - 2);
        p = nx;
        continue;
      }
    }
    n1 = 0;
    while (p < e && (*p == '\n' || *p == '\r')) ++p;
    if (p >= e) break;
    const char* l1 = eol(p, e);
    if (*p == '@' && fastq) {
      if (l1 == e) return false;
      const char* s = l1 + 1; const char* l2 = eol(s, e);
      if (l2 == e) return false;
      const char* pl = l2 + 1; const char* l3 = eol(pl, e);
      if (l3 == e || *pl != '+') return false;
      const char* q = l3 + 1; const char* l4 = eol(q, e);
      const uint32_t nl = (uint32_t)rstrip(p + 1, l1), sl = (uint32_t)rstrip(s, l2);
      R.push_back(Rec{(uint32_t)(p + 1 - b), (uint32_t)(s - b), cs, cn});
      cs += sl; cn += nl;
      // a plain record (every line ended by a bare '\n', nothing stripped): its layout is the prediction for the next one
      if (l4 < e && nl == (uint32_t)(l1 - (p + 1)) && sl == (uint32_t)(l2 - s) && (l3 == pl + 1 || l3[-1] != '\r') && (l4 == q || l4[-1] != '\r')) {
        n1 = (size_t)(l1 + 1 - p); n2 = (size_t)(l2 + 1 - s); n3 = (size_t)(l3 + 1 - pl); n4 = (size_t)(l4 + 1 - q);
      }
      p = l4 < e ? l4 + 1 : e;
    } else if (*p == '>' && !fastq) {
      const char* s = l1 < e ? l1 + 1 : e;
      const uint32_t nl = (uint32_t)rstrip(p + 1, l1);
      // single-line sequence: referenced in place; several lines: joined in the arena
      const char* c = s; int lines = 0; const char* first = s; size_t firstLen = 0;
      const size_t a0 = C.arena.size();
      while (c < e && *c != '>') {
        const char* le = eol(c, e);
        const size_t ll = rstrip(c, le);
        if (ll > 0) {
          if (lines == 0) { first = c; firstLen = ll; }
          else {
            if (lines == 1) C.arena.insert(C.arena.end(), first, first + firstLen);
            C.arena.insert(C.arena.end(), c, c + ll);
          }
          ++lines;
        }
        c = le < e ? le + 1 : e;
      }
      uint32_t sl;
      if (lines <= 1) { R.push_back(Rec{(uint32_t)(p + 1 - b), (uint32_t)(first - b), cs, cn}); sl = (uint32_t)firstLen; }
      else {
        if (C.arena.size() >= (size_t)ARENA) return false;
        R.push_back(Rec{(uint32_t)(p + 1 - b), (uint32_t)a0 | ARENA, cs, cn}); sl = (uint32_t)(C.arena.size() - a0);
      }
      cs += sl; cn += nl;
      p = c;
    } else return false;
  }
  C.n = (uint32_t)R.size();
  R.push_back(Rec{0, 0, cs, cn});
  return true;
}

struct CopyTask { int slot; Chunk* ch; uint32_t r0, r1; int64_t dstRec, dstSeq, dstName; };

struct InSlot {
  qm_batch_bufs bufs;
  int64_t n = 0, seqNo = -1;
  int pending = 0;
  int state = 0;                 // 0 free, 1 being filled, 2 complete, 3 with a consumer
  int allocLeft = 0;             // buffers the allocator threads still owe this slot (it is not used before they are done)
};

struct Src {
  std::string path;
  bool gz = false, fastq = true;
  const char* map = nullptr; size_t len = 0;      // plain: the file; gz: the compressed file
  int fd = -1; bool pread = false;                // plain, QM_INGEST_PREAD: chunks are read() into buffers instead of mapped
  size_t chunkBytes = 0; int64_t nChunks = 0;      // plain
  int64_t nextParse = 0;                           // next chunk id for a parser
  int64_t frontier = 0;                            // chunks [0, frontier) are linked in order
  std::map<int64_t, Chunk*> done;                  // parsed, waiting for their predecessors
  std::deque<Chunk*> line;                         // linked in order, not yet fully assigned
  uint32_t headRec = 0;                            // first unassigned record of line.front()
  int64_t avail = 0;                               // records linked and not yet assigned
  int64_t inParse = 0;                             // chunks with a parser right now
  int64_t doneRecs = 0;                            // records of the chunks in `done`
  int64_t linkedChunks = 0, linkedRecs = 0;        // totals so far (records per chunk: what a chunk in flight is expected to bring)
  bool allHanded = false;                          // plain: nextParse == nChunks; gz: the inflater is done and its queue is empty
  bool eof() const { return allHanded && inParse == 0 && done.empty(); }
  // gz: blocks from the inflate thread, in order
  std::thread inflater; std::deque<Chunk*> blocks; bool inflDone = false;
  int bgzfHelpers = 0;                             // gz: > 0 for a BGZF file (mapped): that many helper threads inflate its blocks
  int pgzThreads = 0;                              // gz: > 0 for any other gzip file (mapped): one stream inflated by that many threads (qm_pgz.h)
};

}  // namespace

struct qm_ingest {
  int nsrc = 0; Src src[2];
  int64_t batchUnits = 0; uint32_t flags = 0;
  std::vector<InSlot> slots;
  std::vector<std::thread> workers, allocators;
  std::mutex mu; std::condition_variable cvWork, cvOut, cvInfl;
  std::deque<CopyTask> copyQ;
  int sleepers = 0;                    // workers waiting on cvWork (a worker that takes a task and sees more wakes ONE of them)
  std::vector<Chunk*> freeChunks;
  int64_t nextSeq = 0, nextHand = 0;
  bool ended = false, stop = false;
  int failed = 0; char err[384] = "";
  // statistics
  double tOpen = 0, tFirst = 0, tEnd = 0, tParse = 0, tCopy = 0, tInfl = 0; int64_t bytesParsed = 0, nParse = 0, nCopy = 0;
};

namespace {

void set_fail(qm_ingest* g, int code, const char* fmt, ...) {
  if (g->failed) return;
  g->failed = code;
  va_list ap; va_start(ap, fmt); vsnprintf(g->err, sizeof(g->err), fmt, ap); va_end(ap);
}

template <typename T>
bool ensure_buf(const qm_batch_bufs& B, T** p, size_t* cap, size_t want) {
  if (*p && *cap >= want) return true;
  if (*p) B.release(*p);
  const size_t nc = want + want / 4 + 4096;
  *p = (T*)B.alloc(nc * sizeof(T)); *cap = *p ? nc : 0;
  return *p != nullptr;
}

Chunk* get_chunk(qm_ingest* g) {                    // (lock held) parse tables are recycled: fresh vectors are fresh pages
  if (!g->freeChunks.empty()) { Chunk* c = g->freeChunks.back(); g->freeChunks.pop_back(); return c; }
  return new Chunk();
}
void put_chunk(qm_ingest* g, Chunk* c) {            // (lock held)
  c->pending = 0; c->assigned = false; c->bad = false; c->n = 0; c->base = c->end = nullptr;
  if (c->data.capacity() > ((size_t)64 << 20)) { std::vector<char>().swap(c->data); }
  if (g->freeChunks.size() < 256) g->freeChunks.push_back(c); else delete c;
}

// (lock held) Turn linked records into batches while both files hold enough of them and a slot is free.
void form_batches(qm_ingest* g) {
  while (!g->failed && !g->ended) {
    int64_t n = g->batchUnits;
    bool ready = true, allEof = true;
    for (int s = 0; s < g->nsrc; ++s) {
      Src& S = g->src[s];
      if (S.avail < g->batchUnits && !S.eof()) ready = false;
      if (!S.eof()) allEof = false;
      n = std::min(n, S.avail);
    }
    if (!ready) return;
    if (n == 0) {
      if (allEof) {
        for (int s = 0; s < g->nsrc; ++s) if (g->src[s].avail != 0) { set_fail(g, QM_E_FORMAT, "paired files have different numbers of records"); break; }
        g->ended = true; g->cvOut.notify_all();
      } else {
        // one file has ended with nothing left, the other still has records: more of it than of its mate
        bool someEofEmpty = false; for (int s = 0; s < g->nsrc; ++s) if (g->src[s].eof() && g->src[s].avail == 0) someEofEmpty = true;
        if (someEofEmpty) { set_fail(g, QM_E_FORMAT, "paired files have different numbers of records"); g->cvOut.notify_all(); }
      }
      return;
    }
    int si = -1;
    for (size_t i = 0; i < g->slots.size(); ++i) if (g->slots[i].state == 0 && g->slots[i].allocLeft == 0) { si = (int)i; break; }
    if (si < 0) return;
    InSlot& L = g->slots[(size_t)si];
    qm_batch_bufs& B = L.bufs;
    const bool names = !(g->flags & QM_INGEST_NO_NAMES);
    // sizes first (the buffers may have to grow), then the copy tasks
    struct Piece { Chunk* ch; uint32_t r0, r1; };
    std::vector<Piece> pieces[2];
    bool ok = true;
    for (int s = 0; s < g->nsrc && ok; ++s) {
      Src& S = g->src[s];
      int64_t left = n, seqBytes = 0, nameBytes = 0; uint32_t h = S.headRec;
      for (size_t ci = 0; left > 0; ++ci) {
        Chunk* c = S.line[ci];
        const uint32_t take = (uint32_t)std::min<int64_t>(left, (int64_t)c->n - h);
        if (take) {
          pieces[s].push_back(Piece{c, h, h + take});
          seqBytes += c->recs[h + take].cumSeq - c->recs[h].cumSeq; nameBytes += c->recs[h + take].cumName - c->recs[h].cumName;
        }
        left -= take; h = 0;
      }
      ok = ensure_buf(B, &B.off[s], &B.cap_off[s], (size_t)n + 1) && ensure_buf(B, &B.seq[s], &B.cap_seq[s], (size_t)seqBytes + 64);
      if (ok && names) ok = ensure_buf(B, &B.noff[s], &B.cap_noff[s], (size_t)n + 1) && ensure_buf(B, &B.names[s], &B.cap_names[s], (size_t)nameBytes + 1);
      if (ok && (g->flags & QM_INGEST_PACK)) {
        ok = ensure_buf(B, &B.pk[s], &B.cap_pk[s], (size_t)(seqBytes >> 2) + (size_t)n + 8) && ensure_buf(B, &B.exc[s], &B.cap_exc[s], (size_t)(seqBytes >> 5) + 4096);
        B.n_exc[s] = 0;
      }
      if (ok) {
        B.off[s][n] = seqBytes; memset(B.seq[s] + seqBytes, 0, 64);   // the mapper fetches reads a word at a time: defined bytes behind the last one
        if (names) B.noff[s][n] = nameBytes;
      }
    }
    if (!ok) { set_fail(g, QM_E_NOMEM, "out of memory for a batch of %lld reads", (long long)n); g->cvOut.notify_all(); return; }
    L.n = n; L.seqNo = g->nextSeq++; L.state = 1; L.pending = 0;
    const char* mre = getenv("QM_INGEST_COPY_RUN");
    const uint32_t maxRun = mre && atoi(mre) > 0 ? (uint32_t)atoi(mre) : 16384u;   // records per copy task
    size_t made = 0;
    for (int s = 0; s < g->nsrc; ++s) {
      Src& S = g->src[s];
      int64_t dRec = 0, dSeq = 0, dName = 0;
      for (const Piece& P : pieces[s]) {
        for (uint32_t r = P.r0; r < P.r1; r += maxRun) {
          const uint32_t r1 = std::min(P.r1, r + maxRun);
          g->copyQ.push_back(CopyTask{si, P.ch, r, r1, dRec, dSeq, dName});
          dRec += r1 - r; dSeq += P.ch->recs[r1].cumSeq - P.ch->recs[r].cumSeq; dName += P.ch->recs[r1].cumName - P.ch->recs[r].cumName;
          P.ch->pending++; L.pending++; ++made;
        }
        if (P.r1 == P.ch->n) P.ch->assigned = true;
      }
      // advance the head of the line
      int64_t left = n;
      while (left > 0) {
        Chunk* c = S.line.front();
        const int64_t take = std::min<int64_t>(left, (int64_t)c->n - S.headRec);
        left -= take; S.headRec += (uint32_t)take;
        if (S.headRec == c->n) { S.line.pop_front(); S.headRec = 0; }
      }
      S.avail -= n;
    }
    // ONE sleeper is woken here; it wakes the next one when it finds more tasks than it takes (worker_loop).  Waking them all
    // for every batch made the engine SLOWER with every worker added beyond ~16: dozens of threads woke, queued for the mutex,
    // found nothing and went back to sleep, in the way of the ones with work (profiles/r04/ingest_hyp.log).
    if (made > 0) g->cvWork.notify_one();
  }
}

// (lock held) a parsed chunk joins the line as soon as every chunk before it has
void link_done(qm_ingest* g, Src& S) {
  while (true) {
    auto it = S.done.find(S.frontier);
    if (it == S.done.end()) break;
    Chunk* c = it->second; S.done.erase(it); S.frontier++;
    S.doneRecs -= c->n; S.linkedChunks++; S.linkedRecs += c->n;
    if (c->n == 0) { put_chunk(g, c); continue; }
    S.line.push_back(c); S.avail += c->n;
  }
}

void run_copy(qm_ingest* g, const CopyTask& T) {
  InSlot& L = g->slots[(size_t)T.slot];
  const Chunk& C = *T.ch; const int s = C.src;
  qm_batch_bufs& B = L.bufs;
  const bool names = !(g->flags & QM_INGEST_NO_NAMES);
  const bool pack = (g->flags & QM_INGEST_PACK) != 0 && B.pk[s] != nullptr;
  uint8_t* pk = B.pk[s]; qm_pack_exc* exc = B.exc[s]; int64_t* nexc = &B.n_exc[s]; const int64_t excCap = (int64_t)B.cap_exc[s];
  char* seq = B.seq[s]; int64_t* off = B.off[s]; char* nm = names ? B.names[s] : nullptr; int64_t* noff = names ? B.noff[s] : nullptr;
  const Rec* R = C.recs.data();
  const int64_t s0 = T.dstSeq - R[T.r0].cumSeq, n0 = T.dstName - R[T.r0].cumName;
  int64_t d = T.dstRec;
  // Short reads and names are moved 16 bytes at a time without a call: the store may run up to 15 bytes past the record's end --
  // over the places of the NEXT records of this task, which are written right after -- and the load up to 15 bytes past its
  // source, which must still lie inside the chunk.  A record whose over-run would leave the task's own stretch of the
  // destination (its neighbour there belongs to another task, or is the batch's padding) and anything long or arena-backed
  // go through memcpy.
  const char* const srcEnd = C.end - 16;
  const int64_t endSeq = s0 + R[T.r1].cumSeq, endName = n0 + R[T.r1].cumName;
  auto put = [srcEnd](char* base, int64_t at, int64_t end, const char* src, uint32_t n, bool exact) {
    char* dst = base + at;
    if (exact || n > 512 || src + n > srcEnd || at + (int64_t)((n + 15u) & ~15u) > end) { memcpy(dst, src, n); return; }
    for (uint32_t i = 0; i < n; i += 16) _mm_storeu_si128((__m128i*)(dst + i), _mm_loadu_si128((const __m128i*)(src + i)));
  };
  for (uint32_t i = T.r0; i < T.r1; ++i, ++d) {
    const uint32_t sl = R[i + 1].cumSeq - R[i].cumSeq;
    const bool arena = (R[i].seqOff & ARENA) != 0;
    const char* sp = arena ? C.arena.data() + (R[i].seqOff & ~ARENA) : C.base + R[i].seqOff;
    off[d] = s0 + R[i].cumSeq;
    put(seq, s0 + R[i].cumSeq, endSeq, sp, sl, arena);
    if (pack) {
      // the same characters four to a byte, in the bytes that are this read's alone; what is not A C G T goes on the slot's
      // exception list (an atomic counter: such characters are rare)
      const int64_t o = s0 + R[i].cumSeq;
      qm_pack::pack_read((const unsigned char*)sp, sl, pk + (o >> 2) + d, [&](size_t j, unsigned char c) {
        const int64_t at = __atomic_fetch_add(nexc, (int64_t)1, __ATOMIC_RELAXED);
        if (at < excCap) exc[at] = qm_pack_exc{(uint32_t)(o + (int64_t)j), (uint32_t)c};
      });
    }
    if (names) {
      noff[d] = n0 + R[i].cumName;
      put(nm, n0 + R[i].cumName, endName, C.base + R[i].nameOff, R[i + 1].cumName - R[i].cumName, false);
    }
  }
}

// A parser's next job, if any (lock held): the next chunk (plain) or the oldest inflated block (gz) of the file that has
// the fewest records in sight, unless that file is already two batches ahead of the batch being formed.
Chunk* take_parse(qm_ingest* g) {
  int best = -1; int64_t bestSight = 0;
  for (int s = 0; s < g->nsrc; ++s) {
    Src& S = g->src[s];
    if (S.allHanded) continue;
    if (S.gz && S.blocks.empty()) continue;
    const int64_t live = (int64_t)S.line.size() + (int64_t)S.done.size() + S.inParse;
    const int64_t per = S.linkedChunks ? std::max<int64_t>(1, S.linkedRecs / S.linkedChunks) : 1;
    const int64_t sight = S.avail + S.doneRecs + S.inParse * per;
    // read-ahead limit.  It is counted in RECORDS in sight, never in chunks alone: a batch needs batchUnits records linked
    // before it can be formed, however many chunks that takes (a cap on live chunks hung the stream when batch_units x bytes per
    // record outgrew cap x chunk size)
    if (sight >= 2 * g->batchUnits && live >= 4) continue;
    if (best < 0 || sight < bestSight) { best = s; bestSight = sight; }
  }
  if (best < 0) return nullptr;
  Src& S = g->src[best];
  Chunk* c;
  if (S.gz) {
    c = S.blocks.front(); S.blocks.pop_front();
    if (S.blocks.empty() && S.inflDone) S.allHanded = true;
    g->cvInfl.notify_all();
  } else {
    c = get_chunk(g);
    c->src = best;
    if (S.nextParse + 1 == S.nChunks) S.allHanded = true;
  }
  c->id = S.nextParse++;
  S.inParse++;
  return c;
}

// Plain file without a mapping: the bytes of chunk c -- from the first record boundary at or behind its offset to the first one
// at or behind the next chunk's offset -- are pread() into the chunk's own buffer (one copy out of the page cache, no page
// faults, no address-space lock), with enough slack behind the chunk's end to find that boundary; widened until both cuts
// were decided without looking at the end of the window, so neighbouring chunks agree on them.
bool load_chunk_pread(Src& S, Chunk& c) {
  const size_t off0 = (size_t)c.id * S.chunkBytes, off1 = std::min(S.len, off0 + S.chunkBytes);
  const size_t A = off0 ? off0 - 1 : 0;
  size_t have = 0;
  for (size_t slack = (size_t)256 << 10;; slack *= 4) {
    const size_t B = std::min(S.len, off1 + slack);
    if (c.data.size() < B - A) c.data.resize(B - A);
    while (have < B - A) {
      const ssize_t r = ::pread(S.fd, c.data.data() + have, B - A - have, (off_t)(A + have));
      if (r < 0) { if (errno == EINTR) continue; return false; }
      if (r == 0) return false;                            // the file shrank under us
      have += (size_t)r;
    }
    const char* buf = c.data.data(); const char* e = buf + (B - A);
    bool h0 = false, h1 = false;
    const char* beg = off0 ? sync_record(buf, buf + 1, e, S.fastq, &h0) : buf;
    const char* end = off1 >= S.len ? e : sync_record(buf, buf + (off1 - A), e, S.fastq, &h1);
    if ((h0 || h1) && B < S.len) continue;
    c.base = beg; c.end = end < beg ? beg : end;
    return true;
  }
}

void worker_loop(qm_ingest* g) {
  std::unique_lock<std::mutex> lk(g->mu);
  while (true) {
    if (g->stop) return;
    if (!g->copyQ.empty()) {
      CopyTask T = g->copyQ.front(); g->copyQ.pop_front();
      if (!g->copyQ.empty() && g->sleepers > 0) g->cvWork.notify_one();      // more where this came from: pass the word on
      lk.unlock();
      const double t0 = now_s();
      run_copy(g, T);
      const double dt = now_s() - t0;
      lk.lock();
      g->tCopy += dt; g->nCopy++;
      InSlot& L = g->slots[(size_t)T.slot];
      if (--L.pending == 0) { L.state = 2; g->tEnd = now_s() - g->tOpen; if (g->tFirst == 0) g->tFirst = g->tEnd; g->cvOut.notify_all(); }
      Chunk* c = T.ch;
      if (--c->pending == 0 && c->assigned) { put_chunk(g, c); g->cvWork.notify_one(); }   // room for another parse
      continue;
    }
    if (!g->failed && !g->ended) {
      Chunk* c = take_parse(g);
      if (c) {
        Src& S = g->src[c->src];
        if (g->sleepers > 0) g->cvWork.notify_one();                           // there may be another chunk to parse: one more worker looks
        lk.unlock();
        const double t0 = now_s();
        bool ok = true;
        if (!S.gz && S.pread) ok = load_chunk_pread(S, *c);
        else if (!S.gz) {
          const char* b = S.map; const char* e = S.map + S.len;
          c->base = sync_record(b, b + (size_t)c->id * S.chunkBytes, e, S.fastq);
          c->end = c->id + 1 >= S.nChunks ? e : sync_record(b, b + (size_t)(c->id + 1) * S.chunkBytes, e, S.fastq);
          if (c->end < c->base) c->end = c->base;
#ifdef MADV_POPULATE_READ
          if (c->end > c->base) {        // map this chunk's pages in one call instead of a fault per 16 pages
            const uintptr_t a = (uintptr_t)c->base & ~(uintptr_t)4095;
            madvise((void*)a, (size_t)((uintptr_t)c->end - a), MADV_POPULATE_READ);
          }
#endif
        }
        if (ok) ok = parse_chunk(*c, S.fastq);
        const double dt = now_s() - t0;
        lk.lock();
        g->tParse += dt; g->nParse++; g->bytesParsed += (int64_t)(c->end - c->base);
        S.inParse--;
        if (!ok) { set_fail(g, QM_E_FORMAT, "%s: malformed FASTA/FASTQ record", S.path.c_str()); put_chunk(g, c); g->cvOut.notify_all(); g->cvWork.notify_all(); continue; }
        S.done[c->id] = c; S.doneRecs += c->n;
        link_done(g, S);
        form_batches(g);
        if (g->ended || g->failed) g->cvWork.notify_all();
        continue;
      }
    }
    ++g->sleepers;
    g->cvWork.wait(lk);
    --g->sleepers;
  }
}

// The last position in [b, e) up to which the buffer holds whole records (b is a record start): FASTQ -- resync a little
// before the end and walk whole records from there; FASTA -- the last header line (its record may continue behind e).
const char* last_boundary(const char* b, const char* e, bool fastq) {
  if (!fastq) {
    const char* p = e;
    while (p > b) {
      const char* q = (const char*)memrchr(b, '>', (size_t)(p - b));
      if (!q) return b;
      if (q == b || q[-1] == '\n') return q;
      p = q;
    }
    return b;
  }
  for (size_t back = (size_t)64 << 10;; back *= 4) {
    const bool whole = back >= (size_t)(e - b);
    const char* p = whole ? b : sync_record(b, e - back, e, true);
    if (p < e) {
      const char* cut = p;
      while (true) {                               // walk whole 4-line records
        const char* q = cut; int lines = 0;
        while (lines < 4 && q < e) { const char* l = eol(q, e); if (l == e) break; q = l + 1; ++lines; }
        if (lines < 4) break;
        cut = q;
      }
      if (cut > p || whole) return cut;
    }
    if (whole) return b;
  }
}

// A BGZF file (bgzip, htslib; also what bcl2fastq writes): a series of complete gzip members of at most 64 KiB, each announcing
// its own compressed size in an extra field ('B','C', RFC 1952 FEXTRA; SAM specification 4.1).  One gzip stream can only be
// inflated by one thread; these members are independent, and their boundaries are known without inflating anything, so a few
// helper threads inflate groups of them side by side and the inflate thread of the source hands the groups on in file order.
// (The reference reads every gz input through one zlib thread, src/FastxParser.cpp:229-328; a plain gzip file still takes
// that road here: gzread below.)
struct BgzfReader {
  const unsigned char* map = nullptr; size_t len = 0, pos = 0;    // the compressed file, and the next block to hand to a helper
  struct Job { const unsigned char* in; size_t inLen; char* out; size_t outLen; bool done = false, bad = false; };   // out: inside the chunk the group belongs to
  std::deque<Job*> q;                          // in file order; helpers take the first one not yet taken
  size_t nextTake = 0;                         // q[nextTake] is the next job for a helper
  std::vector<std::thread> helpers;
  std::mutex mu; std::condition_variable cvJob, cvDone;
  bool stop = false, failed = false;
  double busy = 0; long njobs = 0;             // helpers' inflate time, summed (profiling)
  size_t jobBytes = getenv("QM_INGEST_BGZF_JOB") && atoll(getenv("QM_INGEST_BGZF_JOB")) > 0 ? (size_t)atoll(getenv("QM_INGEST_BGZF_JOB")) : ((size_t)256 << 10);

  // is there a BGZF block at `p`?  -> its total size and uncompressed size
  static bool block_at(const unsigned char* p, size_t avail, size_t& bsize, size_t& isize) {
    if (avail < 28 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (12 + xlen > avail) return false;
    size_t o = 12; bool found = false;
    while (o + 4 <= 12 + xlen) {
      const size_t slen = (size_t)p[o + 2] | ((size_t)p[o + 3] << 8);
      if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) { bsize = ((size_t)p[o + 4] | ((size_t)p[o + 5] << 8)) + 1; found = true; }
      o += 4 + slen;
    }
    if (!found || bsize < 12 + xlen + 8 || bsize > avail) return false;
    isize = (size_t)p[bsize - 4] | ((size_t)p[bsize - 3] << 8) | ((size_t)p[bsize - 2] << 16) | ((size_t)p[bsize - 1] << 24);
    return isize <= 65536;
  }
  // BGZF from the first byte to the last?  (walks the block headers: a file that merely starts like one -- a BGZF member followed
  // by ordinary gzip members, a file cut inside a block -- is left to the single zlib stream, which reads any valid gzip file and
  // reports a truncated one)
  static bool is_bgzf(const unsigned char* p, size_t avail) {
    size_t o = 0;
    while (o < avail) { size_t b, i; if (!block_at(p + o, avail - o, b, i)) return false; o += b; }
    return avail > 0;
  }

  void helper_loop() {
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) { std::lock_guard<std::mutex> lk(mu); failed = true; cvDone.notify_all(); return; }
    // inflate writes AND re-reads its output (matches copy from the last 32 KiB): into a small buffer of the thread's own, then
    // one copy to the group's place in the chunk (measured no slower than inflating straight into the chunk, profiles/r03/bgzf_notes.txt)
    std::vector<char> hot;
    while (true) {
      Job* j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cvJob.wait(lk, [&] { return stop || nextTake < q.size(); });
        if (stop) break;
        j = q[nextTake++];
      }
      // the group: whole members one after the other (zlib checks each member's CRC-32 and length)
      const double th0 = now_s();
      size_t in = 0, out = 0; bool bad = false;
      if (hot.size() < j->outLen + 1) hot.resize(j->outLen + 1);
      while (in < j->inLen && !bad) {
        inflateReset(&zs);
        zs.next_in = (Bytef*)(j->in + in); zs.avail_in = (uInt)std::min<size_t>(j->inLen - in, 1u << 30);
        zs.next_out = (Bytef*)(hot.data() + out); zs.avail_out = (uInt)(j->outLen - out);
        const int rc = inflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END) { bad = true; break; }
        in = (size_t)((const unsigned char*)zs.next_in - j->in); out = (size_t)((char*)zs.next_out - hot.data());
      }
      if (out != j->outLen) bad = true;
      else memcpy(j->out, hot.data(), out);
      std::lock_guard<std::mutex> lk(mu);
      busy += now_s() - th0; ++njobs;
      j->bad = bad; j->done = true;
      cvDone.notify_all();
    }
    inflateEnd(&zs);
  }
  void start(const unsigned char* m, size_t l, int nHelpers) {
    map = m; len = l; pos = 0;
    for (int i = 0; i < nHelpers; ++i) helpers.emplace_back([this] { helper_loop(); });
  }
  // The next chunk's worth of blocks (about `want` bytes of output, at least one block; 0 at the end of the file, -1 when the
  // file stops being BGZF): how many bytes they inflate to.  plan() only walks the headers.
  struct Plan { size_t from, to, outLen; };
  long plan(size_t want, Plan& P) {
    size_t p = pos, outSum = 0;
    while (p < len && outSum < want) {
      size_t bs, is;
      if (!block_at(map + p, len - p, bs, is)) { std::lock_guard<std::mutex> lk(mu); failed = true; return -1; }
      p += bs; outSum += is;
    }
    P.from = pos; P.to = p; P.outLen = outSum; pos = p;
    return (long)outSum;
  }
  // hand the plan's blocks to the helpers in groups of about jobBytes of output, each group with its place in dst[0, P.outLen)
  void submit(const Plan& P, char* dst, std::vector<Job*>& jobs) {
    size_t p = P.from, o = 0;
    std::lock_guard<std::mutex> lk(mu);
    while (p < P.to) {
      size_t q0 = p, outSum = 0;
      while (p < P.to && outSum < jobBytes) { size_t bs, is; block_at(map + p, len - p, bs, is); p += bs; outSum += is; }
      Job* j = new Job(); j->in = map + q0; j->inLen = p - q0; j->out = dst + o; j->outLen = outSum;
      o += outSum;
      q.push_back(j); jobs.push_back(j);
      cvJob.notify_one();
    }
  }
  // all of a chunk's jobs done?  false: a block was corrupt
  bool wait(std::vector<Job*>& jobs) {
    std::unique_lock<std::mutex> lk(mu);
    bool ok = true;
    for (Job* j : jobs) {
      cvDone.wait(lk, [&] { return j->done || failed; });
      if (failed || j->bad) ok = false;
    }
    // the finished jobs are the front of the queue (chunks are waited for in the order they were submitted)
    for (Job* j : jobs) { if (!q.empty() && q.front() == j) { q.pop_front(); --nextTake; } }
    for (Job* j : jobs) delete j;
    jobs.clear();
    return ok && !failed;
  }
  ~BgzfReader() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; cvJob.notify_all(); }
    for (auto& t : helpers) t.join();
    for (Job* j : q) delete j;
  }
};

// The inflate thread of a BGZF source: it only plans and stitches.  A chunk is ~4 MiB worth of blocks, whose inflated sizes
// are known from the block trailers, so the helpers inflate STRAIGHT into the chunk's buffer at their offsets (no copy of the
// decompressed bytes anywhere); a few chunks are in flight at once.  When a chunk's groups are all in, the partial record the
// previous chunk ended with is put in front (a gap is left for it), the chunk is cut at its last record boundary and queued for
// the parse workers like a block of the single-stream gz path.
void bgzf_loop(qm_ingest* g, int s, BgzfReader* bz) {
  Src& S = g->src[s];
  const size_t GAP = (size_t)256 << 10;
  // tuning knobs (profiling): bytes per chunk, chunks in flight
  const size_t BLK = getenv("QM_INGEST_BGZF_CHUNK") && atoll(getenv("QM_INGEST_BGZF_CHUNK")) > 0 ? (size_t)atoll(getenv("QM_INGEST_BGZF_CHUNK")) : ((size_t)4 << 20);
  const size_t DEPTH = getenv("QM_INGEST_BGZF_DEPTH") && atoi(getenv("QM_INGEST_BGZF_DEPTH")) > 0 ? (size_t)atoi(getenv("QM_INGEST_BGZF_DEPTH")) : 4;
  struct Fly { Chunk* c; BgzfReader::Plan P; std::vector<BgzfReader::Job*> jobs; };
  std::deque<Fly*> fly;
  std::vector<char> carry; bool first = true, planEnd = false, bad = false;
  double dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const double tStart = now_s();
  auto finish = [&](bool failed) {
    std::lock_guard<std::mutex> lk(g->mu);
    if (failed) set_fail(g, QM_E_IO, "%s: BGZF block is corrupt or truncated", S.path.c_str());
    for (Fly* f : fly) { put_chunk(g, f->c); }
    S.inflDone = true; S.allHanded = S.blocks.empty();
    g->cvWork.notify_all(); if (failed) g->cvOut.notify_all();
  };
  while (true) {
    {
      const double tg0 = now_s();
      std::unique_lock<std::mutex> lk(g->mu);
      g->cvInfl.wait(lk, [&] { return g->stop || g->failed || S.blocks.size() + fly.size() < 8 + DEPTH || !fly.empty(); });
      dbg[6] += now_s() - tg0;
      if (g->stop || g->failed) break;
    }
    while (!planEnd && !bad && fly.size() < DEPTH) {
      { std::lock_guard<std::mutex> lk(g->mu); if (S.blocks.size() + fly.size() >= 8 + DEPTH && !fly.empty()) break; }
      BgzfReader::Plan P;
      const double tp0 = now_s();
      const long n = bz->plan(BLK, P);
      dbg[0] += now_s() - tp0;
      if (n < 0) { bad = true; break; }
      if (n == 0 && P.from == P.to) { planEnd = true; break; }
      Fly* f = new Fly(); f->P = P;
      const double ts0 = now_s();
      { std::lock_guard<std::mutex> lk(g->mu); f->c = get_chunk(g); }
      const double ts1 = now_s();
      f->c->src = s; f->c->data.resize(GAP + P.outLen);
      const double ts2 = now_s();
      bz->submit(P, f->c->data.data() + GAP, f->jobs);
      dbg[1] += ts1 - ts0; dbg[2] += ts2 - ts1; dbg[3] += now_s() - ts2;
      fly.push_back(f);
    }
    if (fly.empty()) { if (bad) { finish(true); for (Fly* f : fly) delete f; fly.clear(); return; } if (planEnd) break; continue; }
    const double t0 = now_s();
    Fly* f = fly.front(); fly.pop_front();
    const bool ok = bz->wait(f->jobs);
    dbg[4] += now_s() - t0;
    Chunk* c = f->c; const size_t outLen = f->P.outLen;
    delete f;
    if (!ok || bad) {
      // (the helpers may still be writing into the chunks in flight: let them finish before the buffers are recycled)
      for (Fly* x : fly) bz->wait(x->jobs);
      { std::lock_guard<std::mutex> lk(g->mu); put_chunk(g, c); }
      finish(true);
      for (Fly* x : fly) delete x;
      fly.clear();
      return;
    }
    char* base;
    if (carry.size() <= GAP) { base = c->data.data() + GAP - carry.size(); if (!carry.empty()) memcpy(base, carry.data(), carry.size()); }
    else {                                                 // a record longer than the gap: the slow way
      std::vector<char> t(carry.size() + outLen);
      memcpy(t.data(), carry.data(), carry.size()); memcpy(t.data() + carry.size(), c->data.data() + GAP, outLen);
      c->data.swap(t); base = c->data.data();
    }
    const char* end = base + carry.size() + outLen;
    if (first && end > base) { S.fastq = base[0] != '>'; first = false; }
    // the last chunk is decided from the READER's position, not from whether plan() has already run into the end: when the
    // parse workers are behind, the gate above stops planning, so the final chunk can be popped before plan() was asked again
    if (fly.empty() && !planEnd && bz->pos == bz->len) planEnd = true;
    const bool last = planEnd && fly.empty();
    const char* cut = last ? end : last_boundary(base, end, S.fastq);
    carry.assign(cut, end);
    c->base = base; c->end = cut;
    const double dt = now_s() - t0;
    std::unique_lock<std::mutex> lk(g->mu);
    g->tInfl += dt;
    if (c->end > c->base) S.blocks.push_back(c); else put_chunk(g, c);
    dbg[5] += dt;
    if (last) {
      S.inflDone = true; if (S.blocks.empty()) S.allHanded = true; g->cvWork.notify_all();
      if (getenv("QM_INGEST_DEBUG")) fprintf(stderr, "[bgzf %d] plan %.3f get_chunk %.3f resize %.3f submit %.3f wait %.3f (wait+stitch %.3f) gate %.3f total %.3f s\n", s, dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], now_s() - tStart);
      return;
    }
    g->cvWork.notify_one();
  }
  // stopped, or the file was empty of blocks
  for (Fly* x : fly) bz->wait(x->jobs);
  {
    std::lock_guard<std::mutex> lk(g->mu);
    for (Fly* x : fly) put_chunk(g, x->c);
    if (!carry.empty() && !g->stop && !g->failed) {        // never drop a tail: what is still carried is the file's last record(s)
      Chunk* c = get_chunk(g);
      c->src = s; c->data.assign(carry.begin(), carry.end());
      c->base = c->data.data(); c->end = c->base + c->data.size();
      S.blocks.push_back(c); carry.clear();
    }
    if (!S.inflDone) { S.inflDone = true; S.allHanded = S.blocks.empty(); }
    g->cvWork.notify_all();
  }
  for (Fly* x : fly) delete x;
}

void inflate_loop(qm_ingest* g, int s) {
  Src& S = g->src[s];
  // BGZF (independent gzip members that say how long they are): S.bgzfHelpers threads inflate side by side; any other gzip
  // file: one zlib stream, one thread
  BgzfReader* bz = nullptr;
  gzFile f = nullptr;
  if (S.map && S.bgzfHelpers > 0) {
    bz = new BgzfReader(); bz->start((const unsigned char*)S.map, S.len, S.bgzfHelpers);
    bgzf_loop(g, s, bz);
    if (getenv("QM_INGEST_DEBUG")) fprintf(stderr, "[bgzf %d] helpers: %ld jobs, %.3f s busy in total\n", s, bz->njobs, bz->busy);
    delete bz;
    return;
  }
  pgz::PGz* pz = nullptr;
  if (S.map && S.pgzThreads > 0) {
    // one gzip stream, several threads (qm_pgz.h): guessed block starts, symbols for the unknown window, every guess checked
    pz = new pgz::PGz();
    if (!pz->open((const unsigned char*)S.map, S.len, S.pgzThreads)) { delete pz; pz = nullptr; }     // (not a header it knows: zlib's turn)
  }
  if (!pz) {
    f = gzopen(S.path.c_str(), "rb");
    if (!f) { std::lock_guard<std::mutex> lk(g->mu); set_fail(g, QM_E_IO, "cannot gzopen %s", S.path.c_str()); S.inflDone = true; S.allHanded = S.blocks.empty(); g->cvWork.notify_all(); g->cvOut.notify_all(); return; }
    gzbuffer(f, 1 << 20);
  }
  const size_t BLK = (size_t)4 << 20;
  std::vector<char> carry; bool first = true;
  while (true) {
    Chunk* c;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cvInfl.wait(lk, [&] { return g->stop || g->failed || S.blocks.size() < 8; });
      if (g->stop || g->failed) break;
      c = get_chunk(g);
    }
    const double t0 = now_s();
    c->src = s; c->data.resize(carry.size() + BLK);
    if (!carry.empty()) memcpy(c->data.data(), carry.data(), carry.size());
    size_t have = carry.size(); carry.clear();
    bool eof = false, bad = false; const char* cut = nullptr;
    while (true) {
      const long got = pz ? pz->read(c->data.data() + have, c->data.size() - have) : (long)gzread(f, c->data.data() + have, (unsigned)(c->data.size() - have));
      if (got < 0) { bad = true; break; }
      have += (size_t)got;
      if ((size_t)got < c->data.size() - (have - (size_t)got)) eof = true;
      if (first && have > 0) { S.fastq = c->data[0] != '>'; first = false; }
      if (eof) { cut = c->data.data() + have; break; }
      cut = last_boundary(c->data.data(), c->data.data() + have, S.fastq);
      if (cut > c->data.data()) break;
      c->data.resize(c->data.size() * 2);                 // one record larger than the block: keep reading
    }
    if (!bad) {
      carry.assign(cut, (const char*)c->data.data() + have);
      c->base = c->data.data(); c->end = cut;
    }
    const double dt = now_s() - t0;
    std::unique_lock<std::mutex> lk(g->mu);
    g->tInfl += dt;
    if (bad) { set_fail(g, QM_E_IO, "%s: gzip stream is corrupt", S.path.c_str()); put_chunk(g, c); S.inflDone = true; S.allHanded = S.blocks.empty(); g->cvWork.notify_all(); g->cvOut.notify_all(); break; }
    if (c->end > c->base) S.blocks.push_back(c); else put_chunk(g, c);
    if (eof) { S.inflDone = true; if (S.blocks.empty()) S.allHanded = true; g->cvWork.notify_all(); break; }
    g->cvWork.notify_one();
  }
  if (f) gzclose(f);
  if (pz) {
    if (getenv("QM_INGEST_DEBUG")) fprintf(stderr, "[pgz %d] %d threads, %ld rounds: %ld stretches taken as guessed, %ld decoded again; decode %.3f stitch %.3f resolve %.3f s\n", s, S.pgzThreads, pz->rounds, pz->accepted, pz->redone, pz->tDecode, pz->tStitch, pz->tResolve);
    delete pz;
  }
  std::lock_guard<std::mutex> lk(g->mu);
  if (!S.inflDone) { S.inflDone = true; S.allHanded = S.blocks.empty(); }
  g->cvWork.notify_all();
}

int open_src(Src& S, const char* p) {
  S.path = p;
  int fd = ::open(p, O_RDONLY);
  if (fd < 0) return qm_io_fail(QM_E_IO, "cannot open %s", p);
  struct stat st;
  if (fstat(fd, &st) != 0) { ::close(fd); return qm_io_fail(QM_E_IO, "cannot stat %s", p); }
  S.len = (size_t)st.st_size;
  unsigned char magic[2] = {0, 0};
  const ssize_t got = ::pread(fd, magic, 2, 0);
  S.gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
  if (S.gz && S.len >= 28 && !(getenv("QM_INGEST_NO_BGZF") && atoi(getenv("QM_INGEST_NO_BGZF")) != 0)) {
    const char* m = (const char*)mmap(nullptr, S.len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) {
      if (BgzfReader::is_bgzf((const unsigned char*)m, S.len)) { S.map = m; S.bgzfHelpers = 1; madvise((void*)m, S.len, MADV_SEQUENTIAL); }   // the count is set at open
      else if (!(getenv("QM_INGEST_NO_PGZ") && atoi(getenv("QM_INGEST_NO_PGZ")) != 0)) { S.map = m; S.pgzThreads = 1; madvise((void*)m, S.len, MADV_SEQUENTIAL); }   // an ordinary gzip file: qm_pgz.h
      else munmap((void*)m, S.len);
    }
  }
  if (S.len > 0 && !S.gz) {
    S.map = (const char*)mmap(nullptr, S.len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (S.map == MAP_FAILED) { S.map = nullptr; ::close(fd); return qm_io_fail(QM_E_IO, "cannot mmap %s", p); }
    madvise((void*)S.map, S.len, MADV_SEQUENTIAL);
    S.fastq = S.map[0] != '>';
    const char* pe = getenv("QM_INGEST_PREAD");
    S.pread = pe && atoi(pe) != 0;
    if (S.pread) { S.fd = fd; return 0; }
  }
  ::close(fd);
  return 0;
}

}  // namespace

extern "C" {

int qm_ingest_open(const char* path1, const char* path2, int32_t n_threads, int64_t batch_units, int32_t n_slots, uint32_t flags,
                   void* (*alloc)(size_t), void (*release)(void*), qm_ingest** out) {
  if (!path1 || !out || batch_units <= 0 || !alloc || !release) return qm_io_fail(QM_E_ARG, "qm_ingest_open: bad argument");
  qm_ingest* g = new qm_ingest();
  g->tOpen = now_s();
  g->nsrc = path2 ? 2 : 1; g->batchUnits = batch_units; g->flags = flags;
  int rc = open_src(g->src[0], path1);
  if (!rc && path2) rc = open_src(g->src[1], path2);
  if (rc) { for (int s = 0; s < 2; ++s) { if (g->src[s].map) munmap((void*)g->src[s].map, g->src[s].len); if (g->src[s].fd >= 0) ::close(g->src[s].fd); } delete g; return rc; }
  const char* ce = getenv("QM_INGEST_CHUNK");
  // chunk = one parse task: large enough that the bookkeeping between tasks (one mutex) stays small next to the task --
  // 8 MB chunks scaled to 64 workers where 2 MB ones stopped at 16 (profiles/r04/ingest_hyp.log) -- but never so large that a
  // small file leaves workers without a chunk
  size_t chunkBytes = (size_t)8 << 20;
  {
    size_t longest = 0; for (int s = 0; s < g->nsrc; ++s) if (!g->src[s].gz) longest = std::max(longest, g->src[s].len);
    const size_t per = longest / (size_t)(4 * std::max(1, (int)n_threads)) + 1;
    if (per < chunkBytes) chunkBytes = std::max(per, (size_t)1 << 20);
  }
  if (ce && atoll(ce) > 0) chunkBytes = (size_t)atoll(ce);
  for (int s = 0; s < g->nsrc; ++s) {
    Src& S = g->src[s];
    if (S.gz) {
      if (S.bgzfHelpers) {                               // helper threads per BGZF file: the workers' share, 2 .. 16 (QM_INGEST_BGZF_THREADS)
        const char* be = getenv("QM_INGEST_BGZF_THREADS");
        int h = be && atoi(be) > 0 ? atoi(be) : (int)n_threads / g->nsrc;
        S.bgzfHelpers = h < 2 ? 2 : (h > 16 && !be ? 16 : h);
      }
      if (S.pgzThreads) {                                // threads that inflate an ordinary gzip file's ONE stream side by side: the workers' share, 1 .. 32 (QM_INGEST_PGZ_THREADS)
        const char* be = getenv("QM_INGEST_PGZ_THREADS");
        int h = be && atoi(be) > 0 ? atoi(be) : (int)n_threads / g->nsrc;
        S.pgzThreads = h < 1 ? 1 : (h > 32 && !be ? 32 : h);
      }
      continue;
    }
    S.chunkBytes = chunkBytes;
    S.nChunks = (int64_t)((S.len + chunkBytes - 1) / chunkBytes);
    if (S.nChunks == 0) S.allHanded = true;
  }
  g->slots.resize((size_t)std::max(2, (int)n_slots));
  for (InSlot& L : g->slots) { memset(&L.bufs, 0, sizeof(L.bufs)); L.bufs.alloc = alloc; L.bufs.release = release; }
  // The slots' buffers are sized from the head of the files and allocated by threads of their own WHILE the workers already
  // parse (pinning a few hundred MB takes tens of milliseconds: as long as reading millions of records): one thread per
  // array kind walks the slots in order, so slot 0 is complete first and the first batch can be packed a few milliseconds
  // after the open.  A slot whose size could not be guessed (gz input) starts empty; form_batches grows what is too small.
  {
    const bool names = !(flags & QM_INGEST_NO_NAMES);
    for (InSlot& L : g->slots) L.allocLeft = 0;
    for (int s = 0; s < g->nsrc; ++s) {
      Src& S = g->src[s];
      if (S.gz || !S.map) continue;
      Chunk c; c.base = S.map; c.end = S.map + std::min(S.len, (size_t)256 << 10);
      if (c.end < S.map + S.len) { const char* lb = last_boundary(c.base, c.end, S.fastq); if (lb > c.base) c.end = lb; else continue; }
      if (!parse_chunk(c, S.fastq) || c.n == 0) continue;
      const double sb = (double)c.recs[c.n].cumSeq / c.n, nb = (double)c.recs[c.n].cumName / c.n;
      const size_t capSeq = (size_t)((double)batch_units * (sb * 1.05 + 1.0)) + 4096, capNames = (size_t)((double)batch_units * (nb * 1.05 + 1.0)) + 4096;
      const size_t capOff = (size_t)batch_units + 1 + 4096;
      const bool packed = (flags & QM_INGEST_PACK) != 0;
      for (int kind = 0; kind < 6; ++kind) {
        if ((kind == 2 || kind == 3) && !names) continue;
        if (kind >= 4 && !packed) continue;
        { std::lock_guard<std::mutex> lk(g->mu); for (InSlot& L : g->slots) L.allocLeft++; }   // (allocators of earlier kinds are already counting down)
        g->allocators.emplace_back([g, s, kind, capSeq, capNames, capOff]() {
          for (size_t i = 0; i < g->slots.size(); ++i) {
            { std::lock_guard<std::mutex> lk(g->mu); if (g->stop) return; }
            qm_batch_bufs& B = g->slots[i].bufs;
            const size_t capPk = capSeq / 4 + capOff + 8, capExc = capSeq / 32 + 4096;
            const size_t bytes = kind == 0 ? capSeq : (kind == 2 ? capNames : (kind == 4 ? capPk : (kind == 5 ? capExc * sizeof(qm_pack_exc) : capOff * 8)));
            void* p = B.alloc(bytes);
            std::lock_guard<std::mutex> lk(g->mu);
            if (p) {
              if (kind == 0) { B.seq[s] = (char*)p; B.cap_seq[s] = capSeq; }
              else if (kind == 1) { B.off[s] = (int64_t*)p; B.cap_off[s] = capOff; }
              else if (kind == 2) { B.names[s] = (char*)p; B.cap_names[s] = capNames; }
              else if (kind == 3) { B.noff[s] = (int64_t*)p; B.cap_noff[s] = capOff; }
              else if (kind == 4) { B.pk[s] = (uint8_t*)p; B.cap_pk[s] = capPk; }
              else { B.exc[s] = (qm_pack_exc*)p; B.cap_exc[s] = capExc; }
            }
            if (--g->slots[i].allocLeft == 0) { form_batches(g); g->cvWork.notify_all(); }
          }
        });
      }
    }
  }
  for (int s = 0; s < g->nsrc; ++s) if (g->src[s].gz) g->src[s].inflater = std::thread(inflate_loop, g, s);
  const int W = std::max(1, (int)n_threads);
  for (int i = 0; i < W; ++i) g->workers.emplace_back(worker_loop, g);
  // The workers of one engine stay on ONE NUMA node -- the node of the thread that opened it -- when they fit there: they pass
  // chunks, parse tables and slot buffers to each other, and with the parse at memory speed (round 4) a worker set spread over
  // both sockets of this host ran at 107 M pairs/s where the same 32 workers on one socket ran at 185
  // (profiles/r04/ingest_affinity_after_fastparse.log).  Opt-in (round 5): a library does not change thread affinity behind its
  // caller's back, and engines opened side by side on one file would all pick the same node -- QM_INGEST_PIN=1 turns it on; the
  // `quasimap` CLI and bench.py's end_to_end leg (one engine per run, whole-machine jobs) set it.
  {
    const char* pe = getenv("QM_INGEST_PIN");
    if (pe && atoi(pe) != 0) {
      cpu_set_t set; CPU_ZERO(&set); int ncpu = 0;
      unsigned cpu = 0, node = 0;
      bool haveNode = getcpu(&cpu, &node) == 0;
      // ... or rather the node that holds the FILE's pages, when the file is mapped and in the page cache: the opener's own
      // node is wherever the scheduler left that thread, and workers pinned across the socket link from their input ran at
      // half speed (bimodal stream timings, profiles/r04/stream_after_pin.log).  move_pages with no target nodes only asks.
      for (int si = 0; si < g->nsrc; ++si) {
        const Src& S = g->src[si];
        if (S.gz || !S.map || S.len < 4096) continue;
        int votes[64] = {0}; int best = -1;
        for (int k = 0; k < 5; ++k) {
          const size_t at = ((S.len - 1) / 4 * (size_t)k) & ~(size_t)4095;
          volatile char touch = S.map[at]; (void)touch;
          void* page = (void*)((uintptr_t)(S.map + at) & ~(uintptr_t)4095); int status = -1;
          if (syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0) == 0 && status >= 0 && status < 64) {
            if (++votes[status] > (best < 0 ? 0 : votes[best])) best = status;
          }
        }
        if (best >= 0) { node = (unsigned)best; haveNode = true; }
        break;                                                       // (the first mapped source decides)
      }
      if (haveNode) {
        char path[96]; snprintf(path, sizeof(path), "/sys/devices/system/node/node%u/cpulist", node);
        if (FILE* f = fopen(path, "r")) {
          char buf[1024]; buf[0] = 0;
          if (fgets(buf, sizeof(buf), f)) {
            for (char* q = buf; *q && *q != '\n';) {                 // "0-63,128-191"
              char* e1; const long a = strtol(q, &e1, 10); long b = a;
              if (e1 == q) break;
              if (*e1 == '-') { char* e2; b = strtol(e1 + 1, &e2, 10); e1 = e2; }
              for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); ++ncpu; }
              q = *e1 == ',' ? e1 + 1 : e1;
              if (*e1 != ',' ) break;
            }
          }
          fclose(f);
        }
      }
      // (only a subset of what this process may run on, and only when every worker gets a hardware thread of its own there)
      cpu_set_t mine; CPU_ZERO(&mine);
      if (ncpu > 0 && sched_getaffinity(0, sizeof(mine), &mine) == 0) {
        cpu_set_t both; CPU_AND(&both, &set, &mine);
        if (CPU_COUNT(&both) >= W) for (auto& t : g->workers) pthread_setaffinity_np(t.native_handle(), sizeof(both), &both);
      }
    }
  }
  {   // empty inputs end the stream without a single task
    std::lock_guard<std::mutex> lk(g->mu);
    form_batches(g);
  }
  *out = g;
  return QM_OK;
}

int qm_ingest_next(qm_ingest* g, int* slot, int64_t* n_units, int64_t* seq_no, const qm_batch_bufs** bufs) {
  if (!g || !slot || !n_units) return qm_io_fail(QM_E_ARG, "qm_ingest_next: bad argument");
  std::unique_lock<std::mutex> lk(g->mu);
  int si = -1;
  g->cvOut.wait(lk, [&] {
    if (g->failed || g->stop) return true;
    for (size_t i = 0; i < g->slots.size(); ++i) if (g->slots[i].state == 2 && g->slots[i].seqNo == g->nextHand) { si = (int)i; return true; }
    return g->ended && g->nextHand == g->nextSeq;
  });
  if (g->failed) return qm_io_fail(g->failed, "%s", g->err);
  *slot = -1; *n_units = 0;
  if (si < 0) return QM_OK;
  InSlot& L = g->slots[(size_t)si];
  L.state = 3; g->nextHand++;
  *slot = si; *n_units = L.n;
  if (seq_no) *seq_no = L.seqNo;
  if (bufs) *bufs = &L.bufs;
  return QM_OK;
}

void qm_ingest_release(qm_ingest* g, int slot) {
  if (!g || slot < 0 || (size_t)slot >= g->slots.size()) return;
  std::lock_guard<std::mutex> lk(g->mu);
  g->slots[(size_t)slot].state = 0;
  form_batches(g);
  g->cvWork.notify_one();
}

/* [0] seconds from open to the first complete batch, [1] parse tasks (summed over the workers), [2] copy tasks (summed),
 * [3] inflate threads, [4] bytes parsed, [5] tasks run, [6] seconds from open to the last complete batch so far, [7] batches */
void qm_ingest_stats(qm_ingest* g, double* out8) {
  if (!g || !out8) return;
  std::lock_guard<std::mutex> lk(g->mu);
  out8[0] = g->tFirst; out8[1] = g->tParse; out8[2] = g->tCopy; out8[3] = g->tInfl; out8[4] = (double)g->bytesParsed; out8[5] = (double)(g->nParse + g->nCopy);
  out8[6] = g->tEnd; out8[7] = (double)g->nextSeq;
}

/* no further batches: consumers waiting in qm_ingest_next return (end of input), workers wind down */
void qm_ingest_cancel(qm_ingest* g) {
  if (!g) return;
  std::lock_guard<std::mutex> lk(g->mu);
  g->stop = true; g->cvWork.notify_all(); g->cvOut.notify_all(); g->cvInfl.notify_all();
}

void qm_ingest_close(qm_ingest* g) {
  if (!g) return;
  { std::lock_guard<std::mutex> lk(g->mu); g->stop = true; g->cvWork.notify_all(); g->cvOut.notify_all(); g->cvInfl.notify_all(); }
  for (auto& t : g->workers) t.join();
  for (auto& t : g->allocators) t.join();
  for (int s = 0; s < g->nsrc; ++s) if (g->src[s].inflater.joinable()) g->src[s].inflater.join();
  for (int s = 0; s < g->nsrc; ++s) {
    Src& S = g->src[s];
    for (auto& kv : S.done) delete kv.second;
    for (Chunk* c : S.line) delete c;
    for (Chunk* c : S.blocks) delete c;
    if (S.map) munmap((void*)S.map, S.len);
    if (S.fd >= 0) ::close(S.fd);
  }
  for (Chunk* c : g->freeChunks) delete c;
  for (InSlot& L : g->slots) {
    qm_batch_bufs& B = L.bufs;
    for (int m = 0; m < 2; ++m) { if (B.seq[m]) B.release(B.seq[m]); if (B.off[m]) B.release(B.off[m]); if (B.names[m]) B.release(B.names[m]); if (B.noff[m]) B.release(B.noff[m]); if (B.pk[m]) B.release(B.pk[m]); if (B.exc[m]) B.release(B.exc[m]); }
  }
  delete g;
}

}  // extern "C"
