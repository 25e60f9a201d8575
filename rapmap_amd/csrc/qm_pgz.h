// qm_pgz.h -- one ordinary gzip stream inflated by several threads (round 5; the reference reads .gz input through one zlib stream,
// src/FastxParser.cpp:229-328 over include/kseq.h, and so did the ingest engine: 2.4 M pairs/s).
//
// A deflate stream has no index: a block can only be decoded by whoever decoded everything before it, because (a) block starts
// are not byte aligned and not marked, (b) back-references reach up to 32 KiB into what came before.  Both can be worked around
// for TEXT (the approach of pugz, Kerbiriou & Chikhi 2019, restated here from its idea, not its code):
//   * a worker that is dropped at an arbitrary byte offset tries every BIT position from there as the start of a dynamic-Huffman
//     block: the header must describe complete, not over-subscribed codes with an end-of-block symbol (a random position passes that
//     about once in 10^6), the block must then decode without an invalid symbol or distance, every literal must be a text character,
//     and the blocks after it must do the same up to the end of the worker's stretch;
//   * it decodes with the window unknown: output symbols are 16 bits wide, a back-reference that reaches before the worker's start
//     produces "the byte w positions before my start" (256 + w), and copies of such symbols stay symbols;
//   * when the worker in front has finished, the last 32 KiB it produced ARE that window: one pass turns symbols into bytes.
// Nothing depends on a guess being right: a worker's stretch is only used if the worker in front of it arrives, bit for bit, at the
// position it started from; otherwise that stretch is decoded again behind the one in front, with the window known.  Every member's
// CRC-32 and length are checked against its trailer (zlib's crc32 / crc32_combine over the stretches).
//
// Interface: PGz z; z.open(mapped file, bytes, threads); while ((n = z.read(buf, cap)) > 0) ...; z.error() says why a read returned -1.
// Rounds of `threads` stretches of QM_PGZ_STRETCH compressed bytes; the output of a round is handed out before the next one starts.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace pgz {

// Work buffers of tens of MB that a dozen threads fill for the first time at once: anonymous mappings with transparent huge pages
// asked for (a first touch per 2 MB instead of per 4 KB: the page faults of 24 threads in one address space were most of a round)
struct Buf {
  uint8_t* p = nullptr; size_t cap = 0;
  Buf() {}
  Buf(const Buf&) = delete; Buf& operator=(const Buf&) = delete;
  ~Buf() { release(); }
  void release() { if (p) munmap(p, cap); p = nullptr; cap = 0; }
  void need(size_t n) {                                   // contents are NOT kept
    if (n <= cap) return;
    release();
    const size_t c = (n + (n >> 2) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* q = mmap(nullptr, c, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (q == MAP_FAILED) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
    madvise(q, c, MADV_HUGEPAGE);
#endif
    p = (uint8_t*)q; cap = c;
  }
};

// CRC-32 (the gzip polynomial, reflected) eight bytes at a time: this zlib's crc32() does about 1 GB/s per thread, which was most of
// the symbols -> bytes phase; the combination of the stretches' values stays zlib's crc32_combine.
struct Crc8 {
  uint32_t t[8][256];
  Crc8() {
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; t[0][i] = c; }
    for (uint32_t i = 0; i < 256; ++i) for (int j = 1; j < 8; ++j) t[j][i] = (t[j - 1][i] >> 8) ^ t[0][t[j - 1][i] & 0xffu];
  }
  uint32_t run(uint32_t crc, const uint8_t* p, size_t n) const {
    crc = ~crc;
    while (n && ((uintptr_t)p & 7u)) { crc = t[0][(crc ^ *p++) & 0xffu] ^ (crc >> 8); --n; }
    while (n >= 8) {
      uint64_t w; memcpy(&w, p, 8); w ^= crc;
      crc = t[7][w & 0xff] ^ t[6][(w >> 8) & 0xff] ^ t[5][(w >> 16) & 0xff] ^ t[4][(w >> 24) & 0xff] ^
            t[3][(w >> 32) & 0xff] ^ t[2][(w >> 40) & 0xff] ^ t[1][(w >> 48) & 0xff] ^ t[0][w >> 56];
      p += 8; n -= 8;
    }
    while (n--) crc = t[0][(crc ^ *p++) & 0xffu] ^ (crc >> 8);
    return ~crc;
  }
};
static inline const Crc8& crc8() { static const Crc8 c; return c; }

// symbols of a guessed stretch -> bytes, with the window in front of the stretch known (wn: its last wl bytes).  Runs of plain
// literals -- nearly everything -- are packed sixteen at a time.  false: a reference beyond the start of the stream.
static inline bool resolve_symbols(const uint16_t* s, size_t n, const uint8_t* wn, size_t wl, uint8_t* d) {
  size_t j = 0; bool ok = true;
  auto one = [&](size_t i) { const uint16_t v = s[i]; if (v < 256) d[i] = (uint8_t)v; else { const size_t w = (size_t)(v - 256); if (w < wl) d[i] = wn[wl - 1 - w]; else { d[i] = 0; ok = false; } } };
#if defined(__SSE2__)
  for (; j + 16 <= n; j += 16) {
    const __m128i a = _mm_loadu_si128((const __m128i*)(s + j)), b = _mm_loadu_si128((const __m128i*)(s + j + 8));
    const __m128i hi = _mm_or_si128(_mm_srli_epi16(a, 8), _mm_srli_epi16(b, 8));
    if (_mm_movemask_epi8(_mm_cmpeq_epi8(hi, _mm_setzero_si128())) == 0xffff) _mm_storeu_si128((__m128i*)(d + j), _mm_packus_epi16(a, b));
    else for (size_t i = j; i < j + 16; ++i) one(i);
  }
#endif
  for (; j < n; ++j) one(j);
  return ok;
}

#ifndef QM_PGZ_STRETCH
#define QM_PGZ_STRETCH (2u << 20)     // compressed bytes per worker and round
#endif

struct Bits {                          // LSB-first bit reader over memory
  const uint8_t* base; const uint8_t* p; const uint8_t* end; uint64_t buf; int cnt;
  void init(const uint8_t* b, const uint8_t* e, uint64_t bitpos) { base = b; end = e; p = b + (bitpos >> 3); buf = 0; cnt = 0; refill(); drop((int)(bitpos & 7)); }
  inline void refill() {
    if (p + 8 <= end) { uint64_t w; memcpy(&w, p, 8); buf |= w << cnt; const int adv = (63 - cnt) >> 3; p += adv; cnt += adv * 8; }
    else while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
  }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  inline void drop(int n) { buf >>= n; cnt -= n; }
  inline uint64_t pos() const { return (uint64_t)(p - base) * 8 - (uint64_t)cnt; }
  inline bool ok() const { return cnt >= 0; }      // (cnt < 0: read past the end)
};

// canonical Huffman decoding table: primary look-up of PB bits, second-level tables for longer codes.
// entry: bits 0-3 code length (bits to drop; second level: bits beyond PB), bit 4 = pointer to a second-level table (then bits 16..:
// its offset, bits 0-3: its index width), bits 16..: the symbol
struct Huff {
  std::vector<uint32_t> t; int pb = 0;
  // lens[0..n): code lengths (0 = unused).  zlib's rules (inftrees.c): over-subscribed never; incomplete only as a single one-bit
  // code of a length / distance alphabet (allowOne).  false: not a valid code.
  bool build(const uint8_t* lens, int n, int PB, bool allowOne) {
    int count[16] = {0}; int maxl = 0;
    for (int i = 0; i < n; ++i) { count[lens[i]]++; if (lens[i] > maxl) maxl = lens[i]; }
    pb = PB;
    if (maxl == 0) {                                 // no codes at all: allowed for distances (a block of literals); every look-up fails
      if (!allowOne) return false;
      t.assign((size_t)1 << PB, 0xffff0001u); return true;
    }
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; }
    if (left > 0 && !(allowOne && maxl == 1)) return false;
    int next[16]; { int code = 0; count[0] = 0; for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; } }
    const int sb = maxl > PB ? maxl - PB : 0;        // index width of the second-level tables
    t.assign((size_t)1 << PB, 0xffff0001u);          // (an unassigned slot of an incomplete code: an invalid symbol of length 1)
    for (int i = 0; i < n; ++i) {
      const int l = lens[i];
      if (!l) continue;
      uint32_t code = (uint32_t)next[l]++, rev = 0;
      for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
      if (l <= PB) {
        const uint32_t e = ((uint32_t)i << 16) | (uint32_t)l;
        for (uint32_t k = rev; k < (1u << PB); k += 1u << l) t[k] = e;
      } else {
        const uint32_t pre = rev & ((1u << PB) - 1);
        if (!(t[pre] & 16u)) { const uint32_t off = (uint32_t)t.size(); t.resize(t.size() + ((size_t)1 << sb), 0xffff0001u); t[pre] = (off << 16) | 16u | (uint32_t)sb; }
        const uint32_t off = t[pre] >> 16, e = ((uint32_t)i << 16) | (uint32_t)(l - PB);
        for (uint32_t k = rev >> PB; k < (1u << sb); k += 1u << (l - PB)) t[off + k] = e;
      }
    }
    return true;
  }
  inline uint32_t decode(Bits& B) const {            // -> symbol (0xffff: invalid)
    uint32_t e = t[B.peek(pb)];
    if (e & 16u) { B.drop(pb); e = t[(e >> 16) + B.peek((int)(e & 15u))]; }
    B.drop((int)(e & 15u));
    return e >> 16;
  }
};

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Codes { Huff lit, dist; };

// a dynamic block's header behind BFINAL / BTYPE (RFC 1951 3.2.7) -> its two codes
static inline bool read_dynamic(Bits& B, Codes& C) {
  B.refill();
  const int hlit = (int)B.peek(5) + 257; B.drop(5);
  const int hdist = (int)B.peek(5) + 1; B.drop(5);
  const int hclen = (int)B.peek(4) + 4; B.drop(4);
  if (hlit > 286 || hdist > 30) return false;
  static const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  uint8_t cl[19] = {0};
  for (int i = 0; i < hclen; ++i) { B.refill(); cl[ord[i]] = (uint8_t)B.peek(3); B.drop(3); }
  Huff H;
  if (!H.build(cl, 19, 7, false)) return false;
  uint8_t lens[286 + 30 + 138];
  int n = 0;
  while (n < hlit + hdist) {
    B.refill();
    if (!B.ok()) return false;
    const uint32_t s = H.decode(B);
    if (s < 16) lens[n++] = (uint8_t)s;
    else if (s == 16) { if (n == 0) return false; const int r = 3 + (int)B.peek(2); B.drop(2); const uint8_t v = lens[n - 1]; for (int j = 0; j < r; ++j) lens[n++] = v; }
    else if (s == 17) { const int r = 3 + (int)B.peek(3); B.drop(3); for (int j = 0; j < r; ++j) lens[n++] = 0; }
    else if (s == 18) { const int r = 11 + (int)B.peek(7); B.drop(7); for (int j = 0; j < r; ++j) lens[n++] = 0; }
    else return false;
    if (n > hlit + hdist) return false;
  }
  if (lens[256] == 0) return false;                  // no end-of-block code
  if (!C.lit.build(lens, hlit, 10, false)) return false;
  return C.dist.build(lens + hlit, hdist, 8, true);
}
static inline void fixed_codes(Codes& C) {
  uint8_t l[288]; for (int i = 0; i < 144; ++i) l[i] = 8; for (int i = 144; i < 256; ++i) l[i] = 9; for (int i = 256; i < 280; ++i) l[i] = 7; for (int i = 280; i < 288; ++i) l[i] = 8;
  C.lit.build(l, 288, 10, false);
  uint8_t d[30]; for (int i = 0; i < 30; ++i) d[i] = 5;
  C.dist.build(d, 30, 8, true);
}

static inline bool texty(uint32_t c) { return (c >= 32 && c < 127) || c == 10 || c == 13 || c == 9; }
struct TextLut { bool ok[65536 >> 8]; TextLut() { for (uint32_t c = 0; c < 256; ++c) ok[c] = texty(c); } };
static const TextLut TEXT;

// What a worker produces.  SYM = uint16_t: window unknown (symbols 256 + w for "the byte w positions before my start"), literals
// checked to be text; SYM = uint8_t: bytes, the window is the `hist` bytes in front of out[0].
template <typename SYM>
struct Sink {
  SYM* out; size_t n, cap; size_t hist;              // hist (bytes mode): characters available in front of out
  inline bool room(size_t k) const { return n + k <= cap; }
};

// Decode blocks from B on until (a) the end of the member (BFINAL block done) -> 1, (b) a block boundary at or behind bit
// position `limit` -> 0, (c) the sink is full at a block boundary?  No: a full sink is an error of the caller's sizing -> -2,
// (d) invalid data -> -1.  `end`: the bit position reached (a block boundary, or just behind the final block).
template <typename SYM>
static int inflate_blocks(Bits& B, uint64_t limit, Sink<SYM>& S, uint64_t& end) {
  const bool spec = sizeof(SYM) == 2;
  Codes C;
  while (true) {
    if (B.pos() >= limit) { end = B.pos(); return 0; }
    B.refill();
    if (B.cnt < 3) return -1;
    const int fin = (int)B.peek(1); B.drop(1);
    const int typ = (int)B.peek(2); B.drop(2);
    if (typ == 3) return -1;
    if (typ == 0) {
      B.drop(B.cnt & 7);                             // to the byte boundary (cnt counts the bits buffered in front of p)
      B.refill();
      if (B.cnt < 32) return -1;
      const uint32_t len = B.peek(16); B.drop(16);
      const uint32_t nlen = B.peek(16); B.drop(16);
      if ((len ^ 0xffffu) != nlen) return -1;
      if (!S.room(len)) return -2;
      for (uint32_t i = 0; i < len; ++i) {
        B.refill();
        if (B.cnt < 8) return -1;
        const uint32_t c = B.peek(8); B.drop(8);
        if (spec && !texty(c)) return -1;
        S.out[S.n++] = (SYM)c;
      }
    } else {
      if (typ == 1) fixed_codes(C);
      else if (!read_dynamic(B, C)) return -1;
      while (true) {
        if (S.n + 320 > S.cap) return -2;               // (room for three literals and the longest copy, checked once per trip)
        B.refill();
        if (!B.ok()) return -1;
        // up to three literals per refill (three codes of at most 15 bits, then 5 extra bits of a length: 50 of the 56 buffered)
        uint32_t s = C.lit.decode(B);
        if (s < 256) {
          if (spec && !TEXT.ok[s]) return -1;
          S.out[S.n++] = (SYM)s;
          s = C.lit.decode(B);
          if (s < 256) {
            if (spec && !TEXT.ok[s]) return -1;
            S.out[S.n++] = (SYM)s;
            s = C.lit.decode(B);
            if (s < 256) {
              if (spec && !TEXT.ok[s]) return -1;
              S.out[S.n++] = (SYM)s;
              continue;
            }
          }
        }
        if (s == 256) break;
        s -= 257;
        if (s >= 29) return -1;
        const uint32_t len = LBASE[s] + B.peek(LEXT[s]); B.drop(LEXT[s]);
        B.refill();
        const uint32_t ds = C.dist.decode(B);
        if (ds >= 30) return -1;
        const uint32_t dist = DBASE[ds] + B.peek(DEXT[ds]); B.drop(DEXT[ds]);
        if (!B.ok()) return -1;
        if (spec) {
          const int64_t src0 = (int64_t)S.n - (int64_t)dist;
          if (src0 >= 0 && dist >= len) { memcpy(S.out + S.n, S.out + src0, (size_t)len * sizeof(SYM)); S.n += len; }
          else for (uint32_t j = 0; j < len; ++j) {
            const int64_t src = (int64_t)S.n - (int64_t)dist;
            S.out[S.n] = src >= 0 ? S.out[src] : (SYM)(256 + (uint32_t)(-src - 1));
            ++S.n;
          }
        } else {
          if ((size_t)dist > S.n + S.hist) return -1;
          SYM* d = S.out + S.n; const SYM* q = d - dist;
          if (dist >= len) memcpy(d, q, len); else for (uint32_t j = 0; j < len; ++j) d[j] = q[j];
          S.n += len;
        }
      }
    }
    if (fin) { end = B.pos(); return 1; }
  }
}

// gzip member header at byte p (RFC 1952) -> bytes it takes, 0: not a header / truncated
static inline size_t gz_header(const uint8_t* p, const uint8_t* end) {
  if (end - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return 0;
  const int flg = p[3];
  const uint8_t* q = p + 10;
  if (flg & 4) { if (end - q < 2) return 0; const size_t x = q[0] | (q[1] << 8); q += 2; if ((size_t)(end - q) < x) return 0; q += x; }
  if (flg & 8) { while (q < end && *q) ++q; if (q >= end) return 0; ++q; }
  if (flg & 16) { while (q < end && *q) ++q; if (q >= end) return 0; ++q; }
  if (flg & 2) { if (end - q < 2) return 0; q += 2; }
  return (size_t)(q - p);
}

class PGz {
 public:
  ~PGz() { stop(); }
  bool open(const uint8_t* data, size_t len, int threads) {
    stop();
    base_ = data; end_ = data + len; T_ = threads < 1 ? 1 : threads;
    { const char* e = getenv("QM_PGZ_STRETCH"); stretchBytes_ = e && atoll(e) >= 1024 ? (uint64_t)atoll(e) : (uint64_t)QM_PGZ_STRETCH; }   // (tests: small stretches on small files)
    const size_t h = gz_header(base_, end_);
    if (!h) { err_ = "not a gzip file"; return false; }
    bit_ = (uint64_t)h * 8; memberOpen_ = true; crc_ = 0; isize_ = 0;
    W_.reset(new Work[(size_t)T_]);
    win_.clear(); eof_ = false; cur_ = nullptr; curPos_ = 0; failed_ = false; done_ = false; quit_ = false;
    // the rounds run ahead of the reader on a thread of their own (two finished rounds may wait): the workers decode the next
    // stretches while the caller is busy with the bytes of the last ones
    producer_ = std::thread([this] {
      while (true) {
        Out* o;
        { std::unique_lock<std::mutex> lk(mu_);
          cvFree_.wait(lk, [&] { return quit_ || ready_.size() < 2; });
          if (quit_) return;
          if (!free_.empty()) { o = free_.back(); free_.pop_back(); } else { all_.emplace_back(new Out()); o = all_.back().get(); } }
        o->len = 0;
        bool ok = true;
        try { ok = memberOpen_ ? round(*o) : true; } catch (const std::bad_alloc&) { err_ = "out of memory while inflating"; ok = false; }
        const bool last = !ok || !memberOpen_;
        { std::lock_guard<std::mutex> lk(mu_);
          if (!ok) failed_ = true;
          if (ok && o->len) ready_.push_back(o); else free_.push_back(o);
          if (last) done_ = true; }
        cvReady_.notify_all();
        if (last) return;
      }
    });
    return true;
  }
  const std::string& error() const { return err_; }
  // the next bytes of the decompressed stream; 0: end, -1: error
  long read(void* dst, size_t cap) {
    size_t got = 0;
    while (got < cap) {
      if (!cur_ || curPos_ == cur_->len) {
        std::unique_lock<std::mutex> lk(mu_);
        if (cur_) { free_.push_back(cur_); cur_ = nullptr; cvFree_.notify_all(); }
        cvReady_.wait(lk, [&] { return !ready_.empty() || done_; });
        if (ready_.empty()) { if (failed_) return -1; break; }
        cur_ = ready_.front(); ready_.erase(ready_.begin()); curPos_ = 0;
        cvFree_.notify_all();
        continue;
      }
      const size_t k = std::min(cap - got, cur_->len - curPos_);
      memcpy((char*)dst + got, cur_->d.p + curPos_, k); got += k; curPos_ += k;
    }
    return (long)got;
  }
  // statistics: stretches taken as guessed / decoded again behind the one in front
  long accepted = 0, redone = 0, rounds = 0;
  double tDecode = 0, tStitch = 0, tResolve = 0;      // seconds in the three phases of the rounds

 private:
  struct Work {
    uint64_t from = 0, start = 0, end = 0; bool found = false; int status = -1;    // status of inflate_blocks (0 boundary, 1 member end)
    Buf symBuf; size_t symCap = 0; size_t n = 0;     // (only the pages a stretch fills are touched)
    uint16_t* sym() { return (uint16_t*)symBuf.p; }
    Buf bytes; size_t histLen = 0;                   // stretch 0, and stretches that were decoded again: window, then output
    bool asBytes = false, badRef = false; uint32_t crc = 0;
  };
  struct Out { Buf d; size_t len = 0; };             // the bytes of one round
  void stop() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cvFree_.notify_all();
    if (producer_.joinable()) producer_.join();
    ready_.clear(); free_.clear(); all_.clear(); cur_ = nullptr;
  }
  // a worker dropped at bit `from`: the first position that passes as a block start, decoded to the first boundary at or behind `limit`
  void speculate(Work& w, uint64_t from, uint64_t limit, uint64_t giveUp) {
    w.found = false; w.asBytes = false;
    const size_t cap = (size_t)((limit - from) / 8 + 65536) * 8 + (1u << 20);       // (text deflates 3-5x; a stretch that outgrows this is decoded again)
    if (w.symCap < cap) { w.symBuf.need(cap * 2); w.symCap = w.symBuf.cap / 2; }
    Bits B;
    for (uint64_t b = from; b < giveUp; ++b) {
      B.init(base_, end_, b);
      if (B.cnt < 17) return;
      if ((B.peek(3) & 7u) != 4u) continue;          // BFINAL 0, BTYPE 2 (dynamic): bits 0, 0, 1 from the low end
      Bits H = B; H.drop(3);
      Codes C;
      if (!read_dynamic(H, C)) continue;
      Sink<uint16_t> S{w.sym(), 0, w.symCap, 0};
      uint64_t e = 0;
      const int st = inflate_blocks<uint16_t>(B, limit, S, e);
      // the symbol buffer is full: this WAS a block start (a false one fails within a few symbols), and the data deflates better than the
      // buffer was sized for (scanning on from here decoded every true block start of the stretch to the cap and threw it away).
      // Leave the stretch to the serial redo, whose buffer grows with its output
      if (st == -2) return;
      if (st < 0 || S.n < 1024) continue;            // (a real block of text is kilobytes)
      w.found = true; w.start = b; w.end = e; w.status = st; w.n = S.n;
      return;
    }
  }
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  bool round(Out& O) {
    const double tA = now();
    const uint64_t total = (uint64_t)(end_ - base_) * 8;
    const uint64_t stretch = stretchBytes_ * 8;
    int T = serial_ ? 1 : T_;
    while (T > 1 && bit_ + (uint64_t)(T - 1) * stretch + 8 * 65536 >= total) --T;      // no stretch that starts in the file's last bytes
    Work* W = W_.get();
    for (int i = 0; i < T; ++i) { W[(size_t)i].found = false; W[(size_t)i].asBytes = false; W[(size_t)i].badRef = false; W[(size_t)i].n = 0; W[(size_t)i].status = -1; }
    std::vector<std::thread> th;
    // stretch 0: from the known position with the known window, as bytes; stretches 1..: guessed
    std::vector<uint8_t> first;
    const uint64_t lim0 = std::min(total, bit_ + stretch);
    auto run0 = [&]() {
      Work& w = W[0];
      size_t cap = (size_t)((lim0 - bit_) / 8 + 65536) * 8 + (1u << 20);
      while (true) {
        w.bytes.need(win_.size() + cap);
        memcpy(w.bytes.p, win_.data(), win_.size());
        Bits B; B.init(base_, end_, bit_);
        Sink<uint8_t> S{w.bytes.p + win_.size(), 0, cap, win_.size()};
        w.status = inflate_blocks<uint8_t>(B, lim0, S, w.end);
        if (w.status == -2) { cap *= 2; continue; }
        w.n = S.n; w.asBytes = true; w.start = bit_; w.histLen = win_.size();
        break;
      }
    };
    std::atomic<bool> oom{false};                    // (Buf::need throws: an exception that leaves a thread's function ends the process)
    th.emplace_back([&]() { try { run0(); } catch (const std::bad_alloc&) { oom = true; } });
    for (int i = 1; i < T; ++i) th.emplace_back([&, i]() {
      const uint64_t from = bit_ + (uint64_t)i * stretch, lim = std::min(total, bit_ + (uint64_t)(i + 1) * stretch);
      W[(size_t)i].from = from;
      try { speculate(W[(size_t)i], from, lim, lim); } catch (const std::bad_alloc&) { W[(size_t)i].found = false; }   // (no buffer for the guess: the serial redo takes the stretch)
    });
    for (auto& t : th) t.join();
    if (oom) { err_ = "out of memory while inflating"; return false; }
    const double tB = now();
    if (W[0].status < 0) { err_ = "corrupt deflate data"; return false; }
    // stitch: a guessed stretch counts if the one in front arrived exactly where it started; else it is decoded again, as bytes
    std::vector<uint8_t> window = tail(W[0].bytes.p + win_.size(), W[0].n, win_);
    std::vector<std::vector<uint8_t>> wins((size_t)T);
    uint64_t cur = W[0].end; int st = W[0].status; int used = 1;
    for (int i = 1; i < T && st == 0; ++i) {
      Work& w = W[(size_t)i];
      wins[(size_t)i] = window;
      if (w.found && w.start == cur) {
        ++accepted;
        window = tail_sym(w.sym(), w.n, window);
      } else {
        if (cur >= bit_ + (uint64_t)(i + 1) * stretch && i + 1 < T) { w.n = 0; w.asBytes = true; w.end = cur; w.status = 0; used = i + 1; continue; }   // the one in front already covers this stretch
        ++redone;
        const uint64_t lim = std::min(total, bit_ + (uint64_t)(i + 1) * stretch);
        size_t cap = (size_t)((lim > cur ? lim - cur : 0) / 8 + 65536) * 8 + (1u << 20);
        while (true) {
          w.bytes.need(window.size() + cap);
          memcpy(w.bytes.p, window.data(), window.size());
          Bits B; B.init(base_, end_, cur);
          Sink<uint8_t> S{w.bytes.p + window.size(), 0, cap, window.size()};
          w.status = inflate_blocks<uint8_t>(B, lim, S, w.end);
          if (w.status == -2) { cap *= 2; continue; }
          w.n = S.n;
          break;
        }
        if (w.status < 0) { err_ = "corrupt deflate data"; return false; }
        w.asBytes = true; w.histLen = window.size();
        window = tail(w.bytes.p + w.histLen, w.n, window);
      }
      cur = w.end; st = w.status; used = i + 1;
    }
    const double tC = now();
    // symbols -> bytes, every stretch on its own thread; CRC per stretch
    size_t totalOut = 0; std::vector<size_t> at((size_t)used);
    for (int i = 0; i < used; ++i) { at[(size_t)i] = totalOut; totalOut += W[(size_t)i].n; }
    O.d.need(totalOut);
    O.len = totalOut;
    // guesses that keep failing (data that is not text: the block-start test rejects it; data that deflates 10:1 and more: the symbol buffers
    // overflow) cost a scan or a decode per stretch for nothing: from then on one stretch per round, decoded from the known position
    if (!serial_ && rounds >= 2 && accepted * 2 < redone) serial_ = true;
    th.clear();
    for (int i = 0; i < used; ++i) th.emplace_back([&, i]() {
      Work& w = W[(size_t)i];
      uint8_t* d = O.d.p + at[(size_t)i];
      if (w.asBytes) { if (w.n) memcpy(d, w.bytes.p + w.histLen, w.n); }
      else if (!resolve_symbols(w.sym(), w.n, wins[(size_t)i].data(), wins[(size_t)i].size(), d)) w.badRef = true;
      const uint32_t c = crc8().run(0u, d, w.n);
      w.crc = c;
    });
    for (auto& t : th) t.join();
    for (int i = 0; i < used; ++i) {
      Work& w = W[(size_t)i];
      if (w.badRef) { err_ = "corrupt deflate data (a back-reference beyond the start of the stream)"; return false; }
      crc_ = (uint32_t)crc32_combine(crc_, w.crc, (z_off_t)w.n); isize_ += (uint32_t)w.n;
    }
    win_ = window; bit_ = cur;
    tDecode += tB - tA; tStitch += tC - tB; tResolve += now() - tC; ++rounds;
    if (st == 1) {
      // the member ended: trailer (CRC-32, ISIZE) behind the next byte boundary, then another member or the end
      const uint8_t* q = base_ + ((bit_ + 7) >> 3);
      if (end_ - q < 8) { err_ = "truncated gzip trailer"; return false; }
      const uint32_t c = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24), n = q[4] | (q[5] << 8) | (q[6] << 16) | ((uint32_t)q[7] << 24);
      if (c != crc_ || n != isize_) { err_ = "gzip CRC / length mismatch"; return false; }
      q += 8;
      const size_t h = gz_header(q, end_);
      if (h) { bit_ = (uint64_t)(q + h - base_) * 8; crc_ = (uint32_t)crc32(0L, Z_NULL, 0); isize_ = 0; win_.clear(); }
      else memberOpen_ = false;                      // (anything else behind the last member is ignored, as zlib's gzread does)
    } else if (bit_ >= total) { err_ = "truncated gzip stream"; return false; }
    return true;
  }
  static std::vector<uint8_t> tail(const uint8_t* d, size_t n, const std::vector<uint8_t>& before) {
    std::vector<uint8_t> w;
    if (n >= 32768) w.assign(d + n - 32768, d + n);
    else { const size_t keep = std::min(before.size(), (size_t)32768 - n); w.assign(before.end() - (long)keep, before.end()); w.insert(w.end(), d, d + n); }
    return w;
  }
  static std::vector<uint8_t> tail_sym(const uint16_t* s, size_t n, const std::vector<uint8_t>& before) {
    const size_t wl = before.size();
    auto val = [&](uint16_t v) -> uint8_t { return v < 256 ? (uint8_t)v : (wl > (size_t)(v - 256) ? before[wl - 1 - (size_t)(v - 256)] : 0); };
    std::vector<uint8_t> w;
    if (n >= 32768) { w.resize(32768); for (size_t j = 0; j < 32768; ++j) w[j] = val(s[n - 32768 + j]); }
    else { const size_t keep = std::min(wl, (size_t)32768 - n); w.assign(before.end() - (long)keep, before.end()); for (size_t j = 0; j < n; ++j) w.push_back(val(s[j])); }
    return w;
  }
  const uint8_t* base_ = nullptr; const uint8_t* end_ = nullptr; int T_ = 1;
  uint64_t bit_ = 0, stretchBytes_ = QM_PGZ_STRETCH; bool memberOpen_ = false, eof_ = false;
  uint32_t crc_ = 0, isize_ = 0;
  std::vector<uint8_t> win_;                         // the last 32 KiB that were produced
  std::thread producer_; std::mutex mu_; std::condition_variable cvReady_, cvFree_;
  std::vector<std::unique_ptr<Out>> all_; std::vector<Out*> ready_, free_; Out* cur_ = nullptr; size_t curPos_ = 0;
  bool failed_ = false, done_ = false, quit_ = false;
  bool serial_ = false;                              // the guesses keep failing: one stretch per round from here on
  std::unique_ptr<Work[]> W_;                        // the workers' buffers, kept from round to round
  std::string err_;
};

}  // namespace pgz
