// qm_mapper.inl -- the quasi-mapping hot path for ONE wavefront owning ONE read pair
// (or one single-end read).  Written against qm_wave.h: wave-uniform state lives in
// plain scalars, per-lane state in LV<T>.  Compiled for gfx950 by qm_kernels.hip and,
// for tests only, lane-emulated on the CPU by tests/emu/qm_emu.cpp.
//
// Stages (reference file:line each stage restates; nothing here is copied from it):
//   1. strand setup     Kmer.hpp:525-542 (2-bit encode), :92-100 (RC), :484-487 (homopolymer)
//                       -- all L-k+1 k-mers of a read at once: ballots build bit planes of the
//                       2-bit codes, every lane slices its own 31-mer out of the planes.
//   2. seed probes      RapMapUtils.hpp:65-67,226-239 (khash.find) -- all k-mers of both
//                       strands are probed up front (independent loads, memory-level
//                       parallelism) and reduced to presence bitmaps + lane-held intervals.
//   3. collector        SACollector.hpp:108-362 (operator()), :441-677 (getSAHits_),
//                       :366-431 (spotCheck_) replayed as a scalar state machine over the
//                       bitmaps; a run of misses is one masked popcount.
//   4. MMP extension    SASearcher.hpp:88-309 (extendSearchNaive): three binary searches, the
//                       text comparison is 64 characters per step across the lanes.
//   5. hits->mappings   HitManager.cpp:691-882 (+ :587-689, :449-493, :308-322) on small
//                       u64 lists in LDS (global scratch for the rare > QM_CAP lists).
//   6. pair merge       RapMapUtils.hpp:1185-1264 + RapMapSAMapper.cpp:461-551,684-701.
#pragma once
#include "qm_wave.h"
#include "../../include/qmap_mi355.h"

namespace qm {

#define QM_CAP 64      // entries per LDS list
#define QM_GCAP 2048   // entries per global-scratch list (2 strands x <1000 SA entries)
#define QM_DBG_CAP 64  // debug interval records per unit

struct Slot { u64 key; int lb; int ub; };          // 16 B open-addressing slot, key == ~0 empty
struct SaInfo { u32 tid; int pos; };               // transcript id + offset in transcript of SA[i]

struct DevIndex {
  const unsigned char* text;  // n bytes + >= 64 bytes of zero padding
  long long n;
  const int* SA;
  long long nSA;
  const SaInfo* sainfo;
  const Slot* slots;
  u64 hmask;
  int k;
};

struct Batch {
  const unsigned char* seq1; const long long* off1;
  const unsigned char* seq2; const long long* off2;   // null => single-end
  long long n;
  // outputs
  u32* hit_count;          // [n]
  long long* tmp_off;      // [n] offset of the unit's hits in tmp_hits
  qm_hit* tmp_hits;        // [tmp_cap] bump-allocated, compacted into CSR order afterwards
  u64* cursor;             // bump pointer
  long long tmp_cap;
  u64* counters;           // [6] qm_counters
  u64* gscratch;           // per wave: 4 * QM_GCAP u64
  int* status;             // sticky error flags (bit0: tmp overflow, bit1: list overflow)
  qm_sa_interval_hit* dbg_ints;  // optional [n * QM_DBG_CAP]
  u32* dbg_count;                // optional [n]
  // options
  int strict_check, max_num_hits, no_orphans, no_dovetail, max_interval;
  double quasi_cov;
};

template <int NS>
struct WaveMem {
  u64 buf[4][QM_CAP];              // A, B (sort ping-pong), RL, RR
  alignas(8) unsigned char str[4][64 * NS + 16];   // left fwd, left rc, right fwd, right rc (+16: 8-byte over-reads)
};

// ------------------------------------------------------------------ bit helpers
template <int NS>
struct Bits {
  u64 w[NS];
  QM_DEV bool test(int p) const {
    u64 x = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) if (s == (p >> 6)) x = w[s];
    return (x >> (p & 63)) & 1;
  }
};

template <int NS> QM_DEV Bits<NS> b_and(const Bits<NS>& a, const Bits<NS>& b) { Bits<NS> r;
#pragma unroll
  for (int s = 0; s < NS; ++s) r.w[s] = a.w[s] & b.w[s]; return r; }
template <int NS> QM_DEV Bits<NS> b_or(const Bits<NS>& a, const Bits<NS>& b) { Bits<NS> r;
#pragma unroll
  for (int s = 0; s < NS; ++s) r.w[s] = a.w[s] | b.w[s]; return r; }
template <int NS> QM_DEV Bits<NS> b_andn(const Bits<NS>& a, const Bits<NS>& b) { Bits<NS> r;
#pragma unroll
  for (int s = 0; s < NS; ++s) r.w[s] = a.w[s] & ~b.w[s]; return r; }

// first set bit at position >= p, or 64*NS
template <int NS> QM_DEV int first_set_from(const Bits<NS>& b, int p) {
  int res = 64 * NS;
#pragma unroll
  for (int s = NS - 1; s >= 0; --s) {
    u64 x = b.w[s];
    int ws = p >> 6;
    if (s < ws) x = 0;
    else if (s == ws) x &= (~0ULL << (p & 63));
    if (x) res = 64 * s + ctz64(x);
  }
  return res;
}
// number of set bits in [a, b)
template <int NS> QM_DEV int popc_range(const Bits<NS>& b, int a, int e) {
  int c = 0;
  if (e <= a) return 0;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    int lo = a - 64 * s, hi = e - 64 * s;
    if (hi <= 0 || lo >= 64) continue;
    u64 m = ~0ULL;
    if (lo > 0) m &= (~0ULL << lo);
    if (hi < 64) m &= lanemask_lt(hi);
    c += popc64(b.w[s] & m);
  }
  return c;
}
// out bit q = in bit (P-1-q), q < P
template <int NS> QM_DEV Bits<NS> mirror(const Bits<NS>& in, int P) {
  u64 rev[2 * NS + 1];
#pragma unroll
  for (int s = 0; s < NS; ++s) rev[s] = brev64(in.w[NS - 1 - s]);
#pragma unroll
  for (int s = NS; s < 2 * NS + 1; ++s) rev[s] = 0;
  int sh = 64 * NS - P, ws = sh >> 6, bs = sh & 63;
  Bits<NS> out;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    u64 lo = 0, hi = 0;
#pragma unroll
    for (int t = 0; t < 2 * NS; ++t) { if (t == s + ws) { lo = rev[t]; hi = rev[t + 1]; } }
    out.w[s] = (lo >> bs) | (bs ? (hi << (64 - bs)) : 0ULL);
  }
  return out;
}

QM_DEV u64 spread32(u64 x) {
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
  x = (x | (x << 2)) & 0x3333333333333333ULL;
  x = (x | (x << 1)) & 0x5555555555555555ULL;
  return x;
}
// Kmer.hpp:92-100
QM_DEV u64 word_rc(u64 w, int k) {
  w = ((w >> 2) & 0x3333333333333333ULL) | ((w & 0x3333333333333333ULL) << 2);
  w = ((w >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((w & 0x0F0F0F0F0F0F0F0FULL) << 4);
  w = ((w >> 8) & 0x00FF00FF00FF00FFULL) | ((w & 0x00FF00FF00FF00FFULL) << 8);
  w = ((w >> 16) & 0x0000FFFF0000FFFFULL) | ((w & 0x0000FFFF0000FFFFULL) << 16);
  w = (w >> 32) | (w << 32);
  return (~w) >> (2 * (32 - k));
}
// Kmer.hpp:484-487
QM_DEV bool homopolymer(u64 w, int k) {
  u64 mask = (1ULL << (2 * k)) - 1;   // k <= 31
  return w == (mask & ((w << 2) | (w & 3)));
}
QM_DEV u64 hash_mix(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
QM_DEV int upc(unsigned char c) {   // ::toupper on a (signed) char, C locale
  int v = (signed char)c;
  return (v >= 'a' && v <= 'z') ? v - 32 : v;
}
// src/RapMapUtils.cpp:63-72 reverseRead table
QM_DEV unsigned char rc_char(unsigned char c) {
  unsigned char l = c | 0x20;
  return l == 'a' ? 'T' : l == 'c' ? 'G' : l == 'g' ? 'C' : (l == 't' || l == 'u') ? 'A' : 'N';
}

// dense seed hash: exact lookup (RapMapUtils.hpp:65-67).  Two keys at once so both
// first loads are in flight together.
QM_DEV void probe2(const DevIndex& ix, bool doit, u64 ka, u64 kb, bool& fa, int& alb, int& aub,
                   bool& fb, int& blb, int& bub) {
  fa = fb = false; alb = aub = blb = bub = 0;
  if (!doit) return;
  u64 ia = hash_mix(ka) & ix.hmask, ib = hash_mix(kb) & ix.hmask;
  Slot sa = ix.slots[ia], sb = ix.slots[ib];
  while (true) {
    if (sa.key == ka) { fa = true; alb = sa.lb; aub = sa.ub; break; }
    if (sa.key == ~0ULL) break;
    ia = (ia + 1) & ix.hmask; sa = ix.slots[ia];
  }
  while (true) {
    if (sb.key == kb) { fb = true; blb = sb.lb; bub = sb.ub; break; }
    if (sb.key == ~0ULL) break;
    ib = (ib + 1) & ix.hmask; sb = ix.slots[ib];
  }
}

// ------------------------------------------------------------------ stage 1+2
template <int NS>
struct Strand {
  Bits<NS> E;    // eligible in getSAHits_: no N in [p,p+k), not a homopolymer (SACollector.hpp:498-536)
  Bits<NS> E2;   // eligible in the first-hit scan: no N in [p,p+k] (:176-192, note the <=)
  Bits<NS> AV;   // all k characters are ACGT (fromChars succeeds, :602)
  Bits<NS> F;    // khash.find(mer) hit
  Bits<NS> C;    // khash.find(mer.getRC()) hit
  LV<int> flb[NS], fub[NS];   // interval of mer at position 64*s+lane
  LV<int> clb[NS], cub[NS];   // interval of its reverse complement
  bool clean;                 // only ACGTN (either case): the rc strand is the exact mirror
};

template <int NS>
QM_DEV void setup_strand(const DevIndex& ix, const unsigned char* str, int L, Strand<NS>& S) {
  const int k = ix.k;
  const int P = L - k + 1;
  u64 B0[NS + 1], B1[NS + 1], NM[NS + 1], INV[NS + 1];
  bool clean = true;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    LV<bool> b0, b1, nn, iv, odd;
    QM_LANES(l) {
      int idx = 64 * s + l;
      unsigned char c = idx < L ? str[idx] : 0;
      unsigned char cl = c | 0x20;
      bool valid = idx < L && (cl == 'a' || cl == 'c' || cl == 'g' || cl == 't');
      int x = (c >> 1) & 3;
      int code = x ^ (x >> 1);   // A0 C1 G2 T3 (Kmer.hpp:40-51)
      b0[l] = valid && (code & 1);
      b1[l] = valid && (code & 2);
      nn[l] = idx < L && cl == 'n';
      iv[l] = !valid;
      odd[l] = idx < L && !valid && cl != 'n';
    }
    B0[s] = ballot(b0); B1[s] = ballot(b1); NM[s] = ballot(nn); INV[s] = ballot(iv);
    if (ballot(odd)) clean = false;
  }
  B0[NS] = 0; B1[NS] = 0; NM[NS] = 0; INV[NS] = ~0ULL;
  S.clean = clean;
  const u64 maskk = (1ULL << k) - 1, maskk1 = (1ULL << (k + 1)) - 1;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    LV<bool> e, e2, av, ff, fc;
    QM_LANES(l) {
      int p = 64 * s + l;
      u64 nmw = (NM[s] >> l) | (l ? (NM[s + 1] << (64 - l)) : 0ULL);
      u64 ivw = (INV[s] >> l) | (l ? (INV[s + 1] << (64 - l)) : 0ULL);
      u64 x0 = (B0[s] >> l) | (l ? (B0[s + 1] << (64 - l)) : 0ULL);
      u64 x1 = (B1[s] >> l) | (l ? (B1[s + 1] << (64 - l)) : 0ULL);
      bool nwin = (nmw & maskk) != 0, nwin2 = (nmw & maskk1) != 0;
      int d = ctz64(ivw | (1ULL << k));          // chars before the first non-ACGT one, capped at k
      u64 keep = (1ULL << d) - 1;                // partial word when fromChars stops early (Kmer.hpp:535-538)
      x0 &= keep; x1 &= keep;
      u64 r0 = brev64(x0) >> (64 - k), r1 = brev64(x1) >> (64 - k);
      u64 w = spread32(r0) | (spread32(r1) << 1);
      bool hom = homopolymer(w, k);
      bool inP = p < P;
      e[l] = inP && !nwin && !hom;
      e2[l] = inP && !nwin2 && !hom;
      av[l] = inP && d >= k;
      bool fa, fb; int alb, aub, blb, bub;
      probe2(ix, inP && !nwin, w, word_rc(w, k), fa, alb, aub, fb, blb, bub);
      ff[l] = fa; fc[l] = fb;
      S.flb[s][l] = alb; S.fub[s][l] = aub; S.clb[s][l] = blb; S.cub[s][l] = bub;
    }
    S.E.w[s] = ballot(e); S.E2.w[s] = ballot(e2); S.AV.w[s] = ballot(av);
    S.F.w[s] = ballot(ff); S.C.w[s] = ballot(fc);
  }
}

template <int NS> QM_DEV int lane_arr_get(const LV<int> (&a)[NS], int p) {
  int r = 0;
#pragma unroll
  for (int s = 0; s < NS; ++s) if (s == (p >> 6)) r = read_lane(a[s], p & 63);
  return r;
}

// What getSAHits_ sees of one strand of one read.
template <int NS>
struct StrandView {
  Bits<NS> E, AV, F, C;
  const Strand<NS>* src;
  int mode;   // 0: src's mer intervals at p; 1: src's complement intervals at P-1-p (mirrored rc strand)
  int P;
  QM_DEV void interval(int p, int& lb, int& ub) const {
    if (mode == 0) { lb = lane_arr_get<NS>(src->flb, p); ub = lane_arr_get<NS>(src->fub, p); }
    else { int q = P - 1 - p; lb = lane_arr_get<NS>(src->clb, q); ub = lane_arr_get<NS>(src->cub, q); }
  }
};

// SA-interval hits of one strand, lane-distributed: hit i lives in lane i&63 of slot i>>6
template <int NS>
struct IntervalList {
  LV<int> b[NS], e[NS]; LV<u32> len[NS], q[NS];
  int n;
  QM_DEV void push(int lb, int ub, u32 ln, u32 qp) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s == (n >> 6)) {
        QM_LANES(l) { if (l == (n & 63)) { b[s][l] = lb; e[s][l] = ub; len[s][l] = ln; q[s][l] = qp; } }
      }
    }
    ++n;
  }
  QM_DEV void get(int i, int& lb, int& ub, u32& ln, u32& qp) const {
    lb = ub = 0; ln = qp = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s == (i >> 6)) {
        lb = read_lane(b[s], i & 63); ub = read_lane(e[s], i & 63);
        ln = read_lane(len[s], i & 63); qp = read_lane(q[s], i & 63);
      }
    }
  }
};

// ------------------------------------------------------------------ stage 4
// first index i >= i0 at which the comparison loop of SASearcher.hpp:154-180 stops;
// rel: 0 ran off the query/text, 1 query char < text char, 2 query char > text char
QM_DEV int cmp_from(const DevIndex& ix, long long s, const unsigned char* q, int m, int i0, int sentinel,
                    int& rel) {
  int b = i0;
  while (true) {
    LV<bool> stopv; LV<int> relv;
    QM_LANES(l) {
      int idx = b + l;
      bool valid = idx < m && s + idx < ix.n;
      int qc = 0, tc = 0;
      if (valid) {
        qc = (sentinel && idx == m - 1) ? sentinel : upc(q[idx]);
        tc = (signed char)ix.text[s + idx];
      }
      stopv[l] = !valid || qc != tc;
      relv[l] = !valid ? 0 : (qc < tc ? 1 : (qc > tc ? 2 : 0));
    }
    u64 mk = ballot(stopv);
    if (mk) { int f = ctz64(mk); rel = read_lane(relv, f); return b + f; }
    b += 64;
  }
}

// Closed form of extendSearchNaive for intervals of <= 64 suffixes.
// All suffixes strictly between the fences share the query's first `startAt` characters and are
// sorted, so the three binary searches of SASearcher.hpp:150-304 return exactly
//   maxLen = max_j LCP(query, suffix_j),  [lower, upper) = the (contiguous) block attaining it.
// (Search 1 ends with both neighbours of the insertion point probed, and the maximum LCP of a sorted
// list sits next to the insertion point; searches 2/3 bracket the suffixes that have query[0,maxLen)
// as a prefix.  The one case where the reference's loop deviates -- a suffix running off the END of
// the text, SASearcher.hpp:154,180 -- needs the query to match the text's final '$'; queries that
// contain '$' therefore take the literal path below.)
// Every lane owns one suffix: one coalesced SA load, then 8 text bytes per step against a
// wave-uniform 8-byte query word -- 2 dependent loads instead of ~2 per binary-search step.
QM_DEV bool extend_search_wide(const DevIndex& ix, int lbIn, int ubIn, int startAt, const unsigned char* q, int m0,
                               int& lbOut, int& ubOut, int& lenOut) {
  const int width = ubIn - lbIn - 1;
  if (width < 1 || width > 64) return false;
  LV<long long> sv; LV<int> lcp; LV<bool> act;
  QM_LANES(l) {
    bool a = l < width;
    sv[l] = a ? (long long)ix.SA[lbIn + 1 + l] : 0;
    lcp[l] = a ? startAt : -1;
    act[l] = a;
  }
  bool dollar = false;
  for (int i = startAt; i < m0; i += 8) {
    // wave-uniform 8 query bytes starting at q[i] (aligned reads + funnel shift; LDS rows are padded)
    const unsigned char* qa = q + i;
    unsigned long long addr = (unsigned long long)qa;
    const u64* al = (const u64*)(addr & ~7ULL);
    int sh = (int)(addr & 7ULL) * 8;
    u64 lo = al[0], hi = al[1];
    u64 qw = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    qw = uniform(qw);
    int nb = m0 - i < 8 ? m0 - i : 8;
    u64 valid = nb == 8 ? ~0ULL : ((1ULL << (8 * nb)) - 1);
    // any '$' among the valid query bytes?  (haszero trick on qw ^ 0x24..24)
    u64 z = (qw ^ 0x2424242424242424ULL) | ~valid;
    if (((z - 0x0101010101010101ULL) & ~z & 0x8080808080808080ULL) != 0) dollar = true;
    LV<bool> cont;
    QM_LANES(l) {
      bool c = false;
      if (act[l] && lcp[l] == i) {
        u64 tw = load_u64_unaligned(ix.text + sv[l] + i);
        u64 x = (tw ^ qw) & valid;
        if (x) lcp[l] = i + (ctz64(x) >> 3);
        else { lcp[l] = i + nb; c = nb == 8; }
      }
      cont[l] = c;
    }
    if (dollar) return false;
    if (!ballot(cont)) break;
  }
  int mx = wave_max(lcp);
  LV<bool> best;
  QM_LANES(l) { best[l] = act[l] && lcp[l] == mx; }
  u64 bm = ballot(best);
  lbOut = lbIn + 1 + ctz64(bm);
  ubOut = lbIn + 1 + (63 - clz64(bm)) + 1;
  lenOut = mx;
  return true;
}

// SASearcher::extendSearchNaive (SASearcher.hpp:88-309)
QM_DEV void extend_search(const DevIndex& ix, int lbIn, int ubIn, int startAt, const unsigned char* q, int m0,
                          int& lbOut, int& ubOut, int& lenOut) {
  int rel;
  if (extend_search_wide(ix, lbIn, ubIn, startAt, q, m0, lbOut, ubOut, lenOut)) return;
  if (ubIn - lbIn == 2) {                         // :109-126
    lbIn += 1;
    long long s = uniform((int)ix.SA[lbIn]);
    int i = cmp_from(ix, s, q, m0, startAt, 0, rel);
    lbOut = lbIn; ubOut = ubIn; lenOut = i;
    return;
  }
  long long l = lbIn, r = ubIn, c;
  int lcpLP = startAt, lcpRP = startAt, prevILow = startAt, prevIHigh = startAt, maxLen = 0, i;
  while (true) {                                  // :150-209
    c = (l + r) / 2;
    i = lcpLP < lcpRP ? lcpLP : lcpRP;
    long long s = uniform((int)ix.SA[c]);
    i = cmp_from(ix, s, q, m0, i, 0, rel);
    bool plt = rel != 2;
    if (rel == 2) { if (i > prevILow) prevILow = i; }
    else { if (i > prevIHigh) prevIHigh = i; }    // q<t mismatch, or ran off either end
    if (plt) {
      if (c == l + 1) { maxLen = i > prevILow ? i : prevILow; if (prevIHigh > maxLen) maxLen = prevIHigh; break; }
      r = c; lcpRP = i;
    } else {
      if (c == r - 1) { maxLen = i > prevILow ? i : prevILow; if (prevIHigh > maxLen) maxLen = prevIHigh; break; }
      l = c; lcpLP = i;
    }
  }
  int m = maxLen + 1;
  long long bound1 = 0, bound2 = 0;
  for (int pass = 0; pass < 2; ++pass) {          // :215-258, :261-304
    int sentinel = pass == 0 ? '#' : '{';
    l = pass == 0 ? (long long)lbIn : bound1 - 1;
    r = ubIn; lcpLP = startAt; lcpRP = startAt;
    long long res;
    while (true) {
      c = (l + r) / 2;
      i = lcpLP < lcpRP ? lcpLP : lcpRP;
      long long s = uniform((int)ix.SA[c]);
      i = cmp_from(ix, s, q, m, i, sentinel, rel);
      if (rel != 2) { if (c == l + 1) { res = c; break; } r = c; lcpRP = i; }
      else { if (c == r - 1) { res = r; break; } l = c; lcpLP = i; }
    }
    if (pass == 0) bound1 = res; else bound2 = res;
  }
  if (bound1 == bound2) bound2 += 1;              // :307
  lbOut = (int)bound1; ubOut = (int)bound2; lenOut = maxLen;
}

// ------------------------------------------------------------------ stage 3
// SACollector::getSAHits_ (SACollector.hpp:441-677), NIP disabled
template <int NS>
QM_DEV void get_sa_hits(const DevIndex& ix, const Batch& B, const StrandView<NS>& V, const unsigned char* str, int L,
                        int startPos, bool haveInterval, int lb, int ub, long long& cov, u32& strandHits,
                        u32& otherHits, IntervalList<NS>& out) {
  const int k = ix.k, P = L - k + 1;
  int p = startPos;
  bool skip = haveInterval, lastSearch = false;
  int prevMMPEnd = 0;
  Bits<NS> hitm = b_and(V.E, V.F);
  Bits<NS> missC = b_and(b_andn(V.E, V.F), V.C);
  while (true) {
    if (!skip) {
      if (p >= P) break;
      int ph = first_set_from(hitm, p);
      int stop = ph < P ? ph : P;
      otherHits += (u32)popc_range(missC, p, stop);   // misses: spotCheck_ of the complement (:667-675)
      if (ph >= P) break;
      strandHits += 1;                                 // spotCheck_ on the hit (:545)
      otherHits += V.C.test(ph) ? 1u : 0u;
      p = ph;
      V.interval(p, lb, ub);
    }
    skip = false;
    lb = lb - 1 > 0 ? lb - 1 : 0;                      // :553
    int mlen;
    extend_search(ix, lb, ub, k, str + p, L - p, lb, ub, mlen);
    if (ub > lb && ub - lb < B.max_interval) {          // :577-618
      out.push(lb, ub, (u32)mlen, (u32)p);
      int corr = prevMMPEnd > p ? prevMMPEnd - p : 0;
      cov += mlen - corr;
      prevMMPEnd = p + mlen;
      if (p + mlen < L) {
        int kp = p + mlen - (k - 1);
        if (V.AV.test(kp)) { strandHits += V.F.test(kp) ? 1u : 0u; otherHits += V.C.test(kp) ? 1u : 0u; }
      }
    }
    if (lastSearch) return;
    if (p + mlen >= L) return;
    p = p + mlen - (k - 1);                             // NIP off: lce == matchedLen (:635-647)
    if (p + k == L) lastSearch = true;
  }
}

// SACollector::operator() (SACollector.hpp:108-362), disableNIP_ == true.
// str[0] = read, str[1] = reverseRead(read).  Returns foundHit.
template <int NS>
QM_DEV bool collect_read(const DevIndex& ix, const Batch& B, const unsigned char* fwdStr, unsigned char* rcStr, int L,
                         IntervalList<NS>& fwdInts, IntervalList<NS>& rcInts) {
  const int k = ix.k, P = L - k + 1;
  fwdInts.n = 0; rcInts.n = 0;
  if (P <= 0) return false;
  Strand<NS> S;
  setup_strand<NS>(ix, fwdStr, L, S);
  // first-hit scan (:167-237)
  Bits<NS> cand = b_and(S.E2, b_or(S.F, S.C));
  int p0 = first_set_from(cand, 0);
  if (p0 >= P) return false;
  u32 fwdHit = S.F.test(p0) ? 1u : 0u;
  u32 rcHit = S.C.test(p0) ? 1u : 0u;
  long long fwdCov = 0, rcCov = 0;
  const bool useCoverageCheck = B.strict_check != 0;   // disableNIP_ && strictCheck_ (:138)

  StrandView<NS> VF; VF.E = S.E; VF.AV = S.AV; VF.F = S.F; VF.C = S.C; VF.src = &S; VF.mode = 0; VF.P = P;
  bool didCheckFwd = false;
  if (fwdHit) {                                         // :247-254
    didCheckFwd = true;
    int lb, ub; VF.interval(p0, lb, ub);
    get_sa_hits<NS>(ix, B, VF, fwdStr, L, p0, true, lb, ub, fwdCov, fwdHit, rcHit, fwdInts);
  }
  bool checkRC = useCoverageCheck ? (rcHit > 0) : (rcHit >= fwdHit);
  if (checkRC) {                                        // :258-265
    if (S.clean) {
      StrandView<NS> VR; VR.E = mirror(S.E, P); VR.AV = mirror(S.AV, P); VR.F = mirror(S.C, P); VR.C = mirror(S.F, P);
      VR.src = &S; VR.mode = 1; VR.P = P;
      get_sa_hits<NS>(ix, B, VR, rcStr, L, 0, false, 0, 0, rcCov, rcHit, fwdHit, rcInts);
    } else {
      // IUPAC / 'U' characters: reverseRead() is not the mirror image of the 2-bit
      // encoding any more, so the rc strand gets its own setup and probes.
      Strand<NS> R;
      setup_strand<NS>(ix, rcStr, L, R);
      StrandView<NS> VR; VR.E = R.E; VR.AV = R.AV; VR.F = R.F; VR.C = R.C; VR.src = &R; VR.mode = 0; VR.P = P;
      get_sa_hits<NS>(ix, B, VR, rcStr, L, 0, false, 0, 0, rcCov, rcHit, fwdHit, rcInts);
    }
  }
  bool checkFwd = useCoverageCheck ? (fwdHit > 0) : (fwdHit >= rcHit);
  if (!didCheckFwd && checkFwd) {                       // :271-278
    get_sa_hits<NS>(ix, B, VF, fwdStr, L, 0, false, 0, 0, fwdCov, fwdHit, rcHit, fwdInts);
  }
  if (B.strict_check) {                                 // :280-288 (coverage mode; slack 0)
    if (fwdCov > rcCov) rcInts.n = 0;
    else if (rcCov > fwdCov) fwdInts.n = 0;
  }
  if (B.quasi_cov > 0.0) {                              // :343-358
    if (fwdInts.n > 0) { double f = (double)fwdCov / (double)L; if (f < B.quasi_cov) fwdInts.n = 0; }
    if (rcInts.n > 0) { double f = (double)rcCov / (double)L; if (f < B.quasi_cov) rcInts.n = 0; }
  }
  return true;
}

// ------------------------------------------------------------------ stage 5
// list element: tid(31) | isRC(1) | hitPos(32)  -> ascending order == (tid, fwd before rc)
QM_DEV u64 mk_elem(u32 tid, bool isRC, int pos) { return ((u64)tid << 33) | ((u64)(isRC ? 1 : 0) << 32) | (u32)pos; }
QM_DEV u32 el_tid(u64 e) { return (u32)(e >> 33); }
QM_DEV bool el_rc(u64 e) { return (e >> 32) & 1; }
QM_DEV int el_pos(u64 e) { return (int)(u32)e; }

// rank sort of n distinct-after-tiebreak u64 keys: out[rank(in[i])] = in[i]
QM_DEV void rank_sort(const u64* in, u64* out, int n) {
  wave_fence();
  for (int base = 0; base < n; base += 64) {
    QM_LANES(l) {
      int e = base + l;
      if (e < n) {
        u64 ke = in[e];
        int rank = 0;
        for (int j = 0; j < n; ++j) { u64 kj = in[j]; rank += (kj < ke || (kj == ke && j < e)) ? 1 : 0; }
        out[rank] = ke;
      }
    }
  }
  wave_fence();
}

// keep the first element of every run with equal (key >> shift); returns the new length.
// `conv` turns a kept sorted key into the output element.  dst may alias nothing else.
template <typename Conv>
QM_DEV int unique_emit(const u64* sorted, int n, int shift, u64* dst, int dstOff, Conv conv) {
  int outn = 0;
  for (int base = 0; base < n; base += 64) {
    LV<bool> head; LV<u64> val;
    QM_LANES(l) {
      int i = base + l;
      bool h = false; u64 v = 0;
      if (i < n) { v = sorted[i]; h = (i == 0) || ((sorted[i - 1] >> shift) != (v >> shift)); }
      head[l] = h; val[l] = v;
    }
    u64 hm = ballot(head);
    QM_LANES(l) { if (head[l]) dst[dstOff + outn + popc64(hm & lanemask_lt(l))] = conv(val[l]); }
    outn += popc64(hm);
  }
  wave_fence();
  return outn;
}

struct Bufs { u64* A; u64* B; u64* R; };   // sort ping-pong + the read's output list

// collectFromSingleInterval (HitManager.cpp:716-807, considerMultiPos == false)
QM_DEV int single_interval(const DevIndex& ix, const Bufs& bf, int rOff, int lb, int ub, u32 qpos, bool isRC) {
  int n = ub - lb;
  for (int base = 0; base < n; base += 64) {
    QM_LANES(l) {
      int i = base + l;
      if (i < n) {
        SaInfo e = ix.sainfo[lb + i];
        int hitPos = (int)((u32)e.pos - qpos);
        bf.A[i] = ((u64)e.tid << 32) | ((u32)hitPos ^ 0x80000000u);   // signed order on pos (:761-767)
      }
    }
  }
  rank_sort(bf.A, bf.B, n);
  return unique_emit(bf.B, n, 32, bf.R, rOff, [isRC](u64 v) {
    return mk_elem((u32)(v >> 32), isRC, (int)((u32)v ^ 0x80000000u)); });
}

// intersectSAHits + collectHitsSimpleSA (HitManager.cpp:587-689, :449-493, :308-322),
// consensusFraction == 1 (maxSlack 0), strictFilter off.
template <int NS>
QM_DEV int multi_interval(const DevIndex& ix, const Bufs& bf, int rOff, const IntervalList<NS>& ints, bool isRC) {
  const int m = ints.n;
  int minIdx = 0, minSpan = 0x7fffffff;
  for (int i = 0; i < m; ++i) {                       // first smallest span (:636-641)
    int lb, ub; u32 ln, qp; ints.get(i, lb, ub, ln, qp);
    if (ub - lb < minSpan) { minSpan = ub - lb; minIdx = i; }
  }
  int lb0, ub0; u32 ln0, q0; ints.get(minIdx, lb0, ub0, ln0, q0);
  int n0 = ub0 - lb0;
  for (int base = 0; base < n0; base += 64) {
    QM_LANES(l) {
      int i = base + l;
      if (i < n0) { SaInfo e = ix.sainfo[lb0 + i]; bf.A[i] = ((u64)e.tid << 32) | (u32)e.pos; }
    }
  }
  rank_sort(bf.A, bf.B, n0);
  // S: A[j] = tid<<32 | lastActiveInterval(0-based order), B[j] = best (pos<<8 | order)
  // (the per-transcript minimum position, ties -> earliest inserted, :309-313)
  // first build best into A temporarily, then split.
  int s = unique_emit(bf.B, n0, 32, bf.A, 0, [](u64 v) { return v; });
  for (int base = 0; base < s; base += 64) {
    QM_LANES(l) {
      int j = base + l;
      if (j < s) { u64 v = bf.A[j]; bf.B[j] = ((u64)(u32)v << 8); bf.A[j] = (v >> 32) << 32; }
    }
  }
  wave_fence();
  int order = 0;
  for (int ii = 0; ii < m; ++ii) {
    if (ii == minIdx) continue;
    ++order;                                          // intervalCounter - 1
    int lb, ub; u32 ln, qp; ints.get(ii, lb, ub, ln, qp);
    int n = ub - lb;
    for (int base = 0; base < n; base += 64) {
      QM_LANES(l) {
        int i = base + l;
        if (i < n) {
          SaInfo e = ix.sainfo[lb + i];
          int lo = 0, hi = s;                         // lower bound of tid in S
          while (lo < hi) { int mid = (lo + hi) >> 1; if ((u32)(bf.A[mid] >> 32) < e.tid) lo = mid + 1; else hi = mid; }
          if (lo < s) {
            u64 a = bf.A[lo];
            u32 last = (u32)a;
            if ((u32)(a >> 32) == e.tid && (last == (u32)(order - 1) || last == (u32)order)) {   // slack <= 0 (:474-478)
              bf.A[lo] = ((u64)e.tid << 32) | (u32)order;
              atomic_min_u64(&bf.B[lo], ((u64)(u32)e.pos << 8) | (u64)order);
            }
          }
        }
      }
    }
    wave_fence();
  }
  // active <=> seen in every interval (:669-678); ascending tid (std::map order)
  int outn = 0;
  for (int base = 0; base < s; base += 64) {
    LV<bool> act; LV<u64> tp; LV<int> iidx;
    QM_LANES(l) {
      int j = base + l;
      bool a = false; u64 v = 0; int idx = 0;
      if (j < s) {
        u64 aa = bf.A[j], best = bf.B[j];
        a = (u32)aa == (u32)(m - 1);
        int ord = (int)(best & 0xff);
        idx = ord == 0 ? minIdx : (ord <= minIdx ? ord - 1 : ord);   // insertion order -> interval index
        v = ((aa >> 32) << 32) | (u64)(u32)(best >> 8);              // tid<<32 | min pos
      }
      act[l] = a; tp[l] = v; iidx[l] = idx;
    }
    // queryPos of the interval that supplied the minimum (m is tiny: uniform loop)
    LV<u32> qps;
    QM_LANES(l) { qps[l] = 0; }
    for (int ii = 0; ii < m; ++ii) {
      int lb, ub; u32 ln, qp; ints.get(ii, lb, ub, ln, qp);
      QM_LANES(l) { if (iidx[l] == ii) qps[l] = qp; }
    }
    u64 am = ballot(act);
    QM_LANES(l) {
      if (act[l]) {
        u64 v = tp[l];
        int hitPos = (int)((u32)v - qps[l]);                          // pos - queryPos (:315)
        bf.R[rOff + outn + popc64(am & lanemask_lt(l))] = mk_elem((u32)(v >> 32), isRC, hitPos);
      }
    }
    outn += popc64(am);
  }
  wave_fence();
  return outn;
}

// hitsToMappingsSimple (HitManager.cpp:691-882): leaves the read's hits (sorted by tid,
// unique, fwd preferred) in bf.R[0..return)
template <int NS>
QM_DEV int hits_to_mappings(const DevIndex& ix, const Bufs& bf, const IntervalList<NS>& fwdInts,
                            const IntervalList<NS>& rcInts) {
  int nf = 0, nr = 0;
  if (fwdInts.n > 1) nf = multi_interval<NS>(ix, bf, 0, fwdInts, false);
  else if (fwdInts.n == 1) { int lb, ub; u32 ln, qp; fwdInts.get(0, lb, ub, ln, qp); nf = single_interval(ix, bf, 0, lb, ub, qp, false); }
  if (rcInts.n > 1) nr = multi_interval<NS>(ix, bf, nf, rcInts, true);
  else if (rcInts.n == 1) { int lb, ub; u32 ln, qp; rcInts.get(0, lb, ub, ln, qp); nr = single_interval(ix, bf, nf, lb, ub, qp, true); }
  if (nf > 0 && nr > 0) {
    // stable merge by tid, fwd first on ties, duplicates collapse to the first (:834-881)
    int n = nf + nr;
    for (int base = 0; base < n; base += 64) { QM_LANES(l) { int i = base + l; if (i < n) bf.A[i] = bf.R[i]; } }
    rank_sort(bf.A, bf.B, n);
    return unique_emit(bf.B, n, 33, bf.R, 0, [](u64 v) { return v; });
  }
  return nf + nr;
}

// largest list any stage will hold for this read (decides LDS vs global scratch)
template <int NS>
QM_DEV int list_bound(const IntervalList<NS>& a, const IntervalList<NS>& b) {
  int tot = 0;
  const IntervalList<NS>* ls[2] = {&a, &b};
  for (int t = 0; t < 2; ++t) {
    int m = ls[t]->n, mn = 0x7fffffff;
    for (int i = 0; i < m; ++i) { int lb, ub; u32 ln, qp; ls[t]->get(i, lb, ub, ln, qp); if (ub - lb < mn) mn = ub - lb; }
    if (m > 0) tot += mn;
  }
  return tot;
}

template <int NS>
QM_DEV void dump_intervals(const Batch& B, long long unit, int list, const IntervalList<NS>& L, int& cnt) {
  for (int i = 0; i < L.n; ++i) {
    int lb, ub; u32 ln, qp; L.get(i, lb, ub, ln, qp);
    if (cnt < QM_DBG_CAP) {
      QM_LANES(l) {
        if (l == 0) {
          qm_sa_interval_hit h; h.begin = lb; h.end = ub; h.len = ln; h.query_pos = qp;
          h.query_rc = (uint8_t)(list & 1); h.list = (uint8_t)list; h.pad = 0;
          B.dbg_ints[unit * QM_DBG_CAP + cnt] = h;
        }
      }
    }
    ++cnt;
  }
}

// ------------------------------------------------------------------ stage 6 + driver
struct WaveCounters { u64 pe, se, tot, reads, tooMany, mapped; };

QM_DEV qm_hit orphan_hit(u64 e, u32 readLen, int mateStatus) {
  qm_hit h; h.tid = el_tid(e); h.pos = el_pos(e); h.mate_pos = 0; h.frag_len = 0; h.read_len = readLen; h.mate_len = 0;
  h.fwd = el_rc(e) ? 0 : 1; h.mate_is_fwd = 1; h.is_paired = 0; h.mate_status = (uint8_t)mateStatus; h.aln_score = 0;
  return h;
}
QM_DEV bool dovetail(const qm_hit& h) {                 // RapMapSAMapper.cpp:684-698
  if (h.fwd != h.mate_is_fwd) {
    if (h.fwd && h.pos > h.mate_pos) return true;
    else if (h.mate_is_fwd && h.mate_pos > h.pos) return true;
  }
  return false;
}

// allocate `cnt` hit records for this unit (lane-uniform); returns the base or -1 on overflow
QM_DEV long long alloc_hits(const Batch& B, long long unit, int cnt) {
  long long base = 0;
  if (cnt > 0) {
    LV<u64> bv;
    QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor, (u64)cnt); }
    base = (long long)read_lane(bv, 0);
    if (base + cnt > B.tmp_cap) {
      QM_LANES(l) { if (l == 0) { *B.status |= 1; B.hit_count[unit] = 0; B.tmp_off[unit] = 0; } }
      return -1;
    }
  }
  QM_LANES(l) { if (l == 0) { B.hit_count[unit] = (u32)cnt; B.tmp_off[unit] = base; } }
  return base;
}

// One unit = one read pair (or one read when B.seq2 == nullptr).
template <int NS>
QM_DEV void map_unit(const DevIndex& ix, const Batch& B, long long unit, WaveMem<NS>& M, u64* gscr, WaveCounters& wc) {
  const bool paired = B.seq2 != nullptr;
  const int nreads = paired ? 2 : 1;
  int L[2] = {0, 0};
  int nlist[2] = {0, 0};
  u64* lists[2] = {M.buf[2], M.buf[3]};
  int dbg = 0;
  wc.reads += 1;
  for (int r = 0; r < nreads; ++r) {
    const unsigned char* src = r == 0 ? B.seq1 : B.seq2;
    const long long* off = r == 0 ? B.off1 : B.off2;
    long long o0 = off[unit], o1 = off[unit + 1];
    int len = (int)(o1 - o0);
    if (len > 64 * NS) { QM_LANES(l) { if (l == 0) *B.status |= 4; } len = 64 * NS; }
    L[r] = len;
    unsigned char* fs = M.str[2 * r];
    unsigned char* rs = M.str[2 * r + 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      QM_LANES(l) {
        int idx = 64 * s + l;
        // the forward read is kept upper-cased (every consumer applies ::toupper anyway, SASearcher.hpp:111,155)
        if (idx < len) { unsigned char c = src[o0 + idx]; fs[idx] = (unsigned char)upc(c); rs[len - 1 - idx] = rc_char(c); }
      }
    }
    wave_fence();
    IntervalList<NS> fi, ri;
    collect_read<NS>(ix, B, fs, rs, len, fi, ri);
    if (B.dbg_ints) { dump_intervals<NS>(B, unit, 2 * r, fi, dbg); dump_intervals<NS>(B, unit, 2 * r + 1, ri, dbg); }
    Bufs bf;
    int bound = list_bound<NS>(fi, ri);
    if (bound <= QM_CAP) { bf.A = M.buf[0]; bf.B = M.buf[1]; bf.R = M.buf[2 + r]; }
    else { bf.A = gscr; bf.B = gscr + QM_GCAP; bf.R = gscr + (2 + r) * QM_GCAP; }
    lists[r] = bf.R;
    if (bound > QM_GCAP) {               // only reachable with max_interval > 1000
      QM_LANES(l) { if (l == 0) *B.status |= 2; }
      nlist[r] = 0;
    } else {
      nlist[r] = hits_to_mappings<NS>(ix, bf, fi, ri);
    }
  }
  if (B.dbg_count) { QM_LANES(l) { if (l == 0) B.dbg_count[unit] = (u32)dbg; } }

  const int maxHits = B.max_num_hits;
  if (!paired) {                                        // RapMapSAMapper.cpp:232-250
    int n = nlist[0];
    wc.tot += (u64)n;
    if (n > maxHits) n = 0;
    long long base = alloc_hits(B, unit, n);
    if (base >= 0) {
      for (int b0 = 0; b0 < n; b0 += 64) {
        QM_LANES(l) { int i = b0 + l; if (i < n) B.tmp_hits[base + i] = orphan_hit(lists[0][i], (u32)L[0], 0); }
      }
    }
    if (n > 0) wc.mapped += 1;
    return;
  }

  // mergeLeftRightHits (RapMapUtils.hpp:1185-1264)
  const u64* LL = lists[0]; const u64* RR = lists[1];
  const int nl = nlist[0], nr = nlist[1];
  int nm = 0;       // matches on transcript id
  int nkeep = 0;    // ... that survive --noDovetail
  if (nl > 0 && nr > 0) {
    for (int b0 = 0; b0 < nl; b0 += 64) {
      LV<bool> mt, kp;
      QM_LANES(l) {
        int i = b0 + l; bool m = false, kk = false;
        if (i < nl) {
          u32 tid = el_tid(LL[i]);
          int lo = 0, hi = nr;
          while (lo < hi) { int mid = (lo + hi) >> 1; if (el_tid(RR[mid]) < tid) lo = mid + 1; else hi = mid; }
          if (lo < nr && el_tid(RR[lo]) == tid) {
            m = true; kk = true;
            if (B.no_dovetail) {
              qm_hit h; h.fwd = el_rc(LL[i]) ? 0 : 1; h.mate_is_fwd = el_rc(RR[lo]) ? 0 : 1;
              h.pos = el_pos(LL[i]) > 0 ? el_pos(LL[i]) : 0; h.mate_pos = el_pos(RR[lo]) > 0 ? el_pos(RR[lo]) : 0;
              kk = !dovetail(h);
            }
          }
        }
        mt[l] = m; kp[l] = kk;
      }
      nm += popc64(ballot(mt)); nkeep += popc64(ballot(kp));
    }
  }
  bool tooMany = nm > maxHits;                          // :1233-1234
  if (tooMany) wc.tooMany += 1;
  int cnt = 0; int kind = 0;                            // kind 1 paired, 2 orphans
  if (!tooMany && nm > 0) { wc.pe += (u64)nm; cnt = nkeep; kind = 1; }
  else if (!tooMany && nl + nr > 0) {
    wc.se += (u64)(nl + nr);
    cnt = nl + nr; kind = 2;
    if (cnt > maxHits) { cnt = 0; kind = 0; }           // RapMapSAMapper.cpp:534-536
    if (B.no_orphans) { cnt = 0; kind = 0; }            // :539-551
    if (kind == 2 && B.no_dovetail) {
      // the reference evaluates the dovetail predicate on orphans with an uninitialised
      // matePos; we define matePos = 0, mateIsFwd = true (same as the oracle).
      int c2 = 0;
      for (int t = 0; t < 2; ++t) {
        const u64* X = t == 0 ? LL : RR; int nx = t == 0 ? nl : nr;
        for (int b0 = 0; b0 < nx; b0 += 64) {
          LV<bool> kp;
          QM_LANES(l) { int i = b0 + l; kp[l] = i < nx && !dovetail(orphan_hit(X[i], 0, 1)); }
          c2 += popc64(ballot(kp));
        }
      }
      cnt = c2;
    }
  }
  long long base = alloc_hits(B, unit, cnt);
  wc.tot += (u64)cnt;
  if (cnt > 0) wc.mapped += 1;
  if (base < 0 || cnt == 0) return;
  if (kind == 1) {
    int outn = 0;
    for (int b0 = 0; b0 < nl; b0 += 64) {
      LV<bool> kp; LV<int> rj;
      QM_LANES(l) {
        int i = b0 + l; bool kk = false; int jj = 0;
        if (i < nl) {
          u32 tid = el_tid(LL[i]);
          int lo = 0, hi = nr;
          while (lo < hi) { int mid = (lo + hi) >> 1; if (el_tid(RR[mid]) < tid) lo = mid + 1; else hi = mid; }
          if (lo < nr && el_tid(RR[lo]) == tid) { kk = true; jj = lo; }
        }
        kp[l] = kk; rj[l] = jj;
      }
      LV<bool> kp2;
      LV<qm_hit> hv;
      QM_LANES(l) {
        bool kk = kp[l];
        if (kk) {
          int i = b0 + l;
          u64 le = LL[i], re = RR[rj[l]];
          int s1 = el_pos(le) > 0 ? el_pos(le) : 0, s2 = el_pos(re) > 0 ? el_pos(re) : 0;   // :1213-1214
          bool r1First = s1 < s2;
          int fragStart = r1First ? s1 : s2;
          int fragEnd = r1First ? (int)((u32)s2 + (u32)L[1]) : (int)((u32)s1 + (u32)L[0]);
          qm_hit h; h.tid = el_tid(le); h.pos = s1; h.mate_pos = s2; h.frag_len = (u32)(fragEnd - fragStart);
          h.read_len = (u32)L[0]; h.mate_len = (u32)L[1]; h.fwd = el_rc(le) ? 0 : 1; h.mate_is_fwd = el_rc(re) ? 0 : 1;
          h.is_paired = 1; h.mate_status = 3; h.aln_score = 0;
          if (B.no_dovetail && dovetail(h)) kk = false;
          hv[l] = h;
        }
        kp2[l] = kk;
      }
      u64 km = ballot(kp2);
      QM_LANES(l) { if (kp2[l]) B.tmp_hits[base + outn + popc64(km & lanemask_lt(l))] = hv[l]; }
      outn += popc64(km);
    }
  } else {
    int outn = 0;
    for (int t = 0; t < 2; ++t) {
      const u64* X = t == 0 ? LL : RR; int nx = t == 0 ? nl : nr;
      for (int b0 = 0; b0 < nx; b0 += 64) {
        LV<bool> kp; LV<qm_hit> hv;
        QM_LANES(l) {
          int i = b0 + l; bool kk = false;
          if (i < nx) { qm_hit h = orphan_hit(X[i], (u32)L[t], t == 0 ? 1 : 2); kk = !(B.no_dovetail && dovetail(h)); hv[l] = h; }
          kp[l] = kk;
        }
        u64 km = ballot(kp);
        QM_LANES(l) { if (kp[l]) B.tmp_hits[base + outn + popc64(km & lanemask_lt(l))] = hv[l]; }
        outn += popc64(km);
      }
    }
  }
}

}  // namespace qm
