// qm_lean.inl -- stage A for the reads that make up nearly all of a batch: at most 128 characters, nothing but A C G T (either
// case), no run of k equal bases, seed intervals of at most 64 suffixes, hits on one strand.  Same algorithm and same answers
// as map_read (SACollector::operator() + hitsToMappingsSimple, the lines qm_mapper.inl cites), another shape:
//
//   * TWO reads per wavefront and iteration (the two mates of a pair, or two consecutive single-end reads): lanes 0-31 take
//     the first, lanes 32-63 the second -- 32 lanes x 4 characters is exactly a 128-character read -- so the regular phases
//     (offsets, characters -> 2-bit image, the first probe: eight look-ups in one round) are issued once for both.  The walks
//     of the two reads follow one another.
//   * no per-position flag registers and no interval table: a probe of 32 positions leaves two 32-bit masks in SGPRs (k-mer
//     found / reverse complement found) and the intervals in the lanes that looked them up; the walk is find-first / popcount
//     on those masks and a v_readlane for the interval.  The walk never turns back, so nothing older than the current window
//     is needed (a position that falls out of the window and is asked for again is looked up again: look-ups are pure).
//   * the reverse-complemented read is not a string of its own: every lane that packs four characters also writes the
//     reverse complement of its byte at the mirrored place of a second image; position q of reverseRead(read) is position
//     q + (128 - L) of that image.  The complement of a window's k-mers comes out of it by the same funnel shift as the
//     k-mers themselves (no per-lane word_rc), and a read that maps to the reverse strand needs no second strand set-up.
//   * hits -> mappings in registers for every list shape: the (transcript, position) of each suffix arrives with the MMP
//     extension's own load (SaExt), is parked in LDS per strand, and one loop of lane reads does the per-transcript minimum,
//     the intersection over the intervals (a bit per interval) and the rank by transcript; the list goes from the registers
//     to B.lists.
//
// A read this kernel does not take -- dirty or long reads, wide intervals, more than 64 suffixes on a strand, hits on both
// strands, a 128-character read that matches beyond the SaExt window -- is marked (QM_LCNT_LEAN in its list-length word, count in
// scalar slot QM_SC_LEANQ) before anything else was written for it; the host gathers the marks into a queue and qm_read_kernel
// maps the queue in a second, small launch.
// Options the kernel is built for: dense table (or the -p image expanded into one; PH: the compact -p image), sensitive mode, lists out
// or (SEL) the -s collector's interval records; the host launches the general kernel for everything else.
//
// WIDE (end of round 5): the same code with ONE read of up to 256 characters per wavefront and iteration -- 64 lanes x 4 characters,
// the two strands' images 16 words apart, MMP extensions through the 224-character table (SaExt2).  Reads of 129 .. 256 characters
// take it (qm_host.hip, run_stage_a: leanWide).
#pragma once
#include "qm_mapper.inl"

namespace qm {

#define QM_LEAN_MAXLEN 128       // two reads per wavefront (32 lanes x 4 characters each); the wide edition -- one read over all 64 lanes -- takes 256
#define QM_LEAN_SUF 64           // suffixes a strand's intervals may hold together
#define QM_LEAN_MAXIV 32         // intervals per strand (a bit each)
#define QM_SC_LEANQ 30           // scalar slot: reads on the lean kernel's queue
#define QM_SC_DEFER0 20          // ... and why they are there, four slots (not in -DQM_TIMING builds, whose phase sums sit here): 0 a character that is not
                                 // A C G T or more than 128 (256) characters, 1 a window of k equal bases, 2 an interval wider than the kernel's lanes / more
                                 // suffixes or intervals than its stash / a match beyond the extension table, 3 hits on the other strand as well
#define QM_LEAN_NMW 12           // words of a strand's N flags: four of flags, zeros behind them (lean_ndist looks at five from any position up to 128)
#define QM_LEAN_CHUNK 1024       // list elements a wave reserves per bump-allocator round trip (a list here holds at most 64): a quarter of the general
                                 // kernels' QM_CHUNK, so that twice their grid leaves half their slack in the list buffer (the host sizes it for theirs)

struct LeanSuf { u32 tid, pos, qp, iv; };      // one suffix of a recorded interval: transcript, offset in it, queryPos and index of the interval

struct LeanMem {                               // one wave's LDS slab (1 632 bytes)
  u64 pk[2][2][8];                             // [read of the iteration][0: the read, 1: mirrored reverse complement][word]: 2 bits per base, first
                                               // base in the top bits of word 0; words 4-7 stay zero (extension queries read past the image)
  union {
    LeanSuf suf[QM_LEAN_SUF];                  // suffixes of the intervals recorded for the read being mapped (one strand is walked)
    IntRec ints[QM_LEAN_MAXIV];                // -s collector: the interval records themselves (SAIntervalHit)
  };
  u32 stage[2][36];                            // raw characters of the next iteration's two reads (global_load_lds target)
  u32 ostage[2][8];                            // offsets (dwords) of the next / the next but one iteration
  u32 nm[2][2][QM_LEAN_NMW];                   // [read][strand]: a bit per position that holds an N (round 6; words 4 .. stay zero), written for
  u32 nmz[QM_LEAN_NMW];                        // reads with N's only; every other read's walks look at nmz: all zero
};

// the current probe window of a walk: positions [wb, wb + ww) of the strand, bit j of Fm / Cm = k-mer / reverse complement of
// position wb + j found, lane j (< 32) holds that position's interval
struct LeanWin { int wb, ww, sh; u32 Fm, Cm, Km, Xm; LV<u32> lb, ub; };   // sh: bit / lane j = position wb + (j << sh) (a strided probe of the -s walk; 0: every position)   // Km: positions of the window that were looked up (-s probes every stride-th)
struct LeanStrand { int n, sufN, minIdx, minSpan, cov; };   // intervals, their suffixes, the first smallest interval (HitManager.cpp:636-641), coverage

QM_DEV int ctz32(u32 x) { return x ? __builtin_ctz(x) : 32; }
QM_DEV int popc32(u32 x) { return __builtin_popcount(x); }

// k-mer word at position q of an image (Kmer.hpp:525-542 over clean characters)
QM_DEV u64 lean_kmer(const QM_LDS(u64)* img, int q, int k) {
  const int j = q >> 5, sh = 2 * (q & 31);
  const u64 w0 = img[j], w1 = img[j + 1];
  return ((w0 << sh) | ((w1 >> 1) >> (63 - sh))) >> (64 - 2 * k);
}

// Reads with N's (round 6).  SACollector never looks up a k-mer that holds an N (operator(): SACollector.hpp:172-181, getSAHits_: :497-512 --
// it goes on behind the N, which is what stepping over every such position comes to), spot-checks none (:603-604) and an MMP ends at one
// (extendSearchNaive compares characters).  nm: the strand's N flags, a bit per position (LeanMem::nm; all zero for a read without).
// The flags of positions q .. q + 31:
QM_DEV u32 lean_nbits(const QM_LDS(u32)* nm, int q) {
  const int i = q >> 5, sh = q & 31;
  const u32 a = nm[i], b = nm[i + 1];
  return (a >> sh) | ((b << 1) << (31 - sh));
}
// characters from position q to the strand's next N (128 and more: none)
QM_DEV int lean_ndist(const QM_LDS(u32)* nm, int q) {
  const int i = q >> 5, sh = q & 31;
  const u32 a = nm[i], b = nm[i + 1], c = nm[i + 2], d = nm[i + 3], e = nm[i + 4];
  const u32 x0 = (a >> sh) | ((b << 1) << (31 - sh)), x1 = (b >> sh) | ((c << 1) << (31 - sh));
  const u32 x2 = (c >> sh) | ((d << 1) << (31 - sh)), x3 = (d >> sh) | ((e << 1) << (31 - sh));
  return x0 ? __builtin_ctz(x0) : (x1 ? 32 + __builtin_ctz(x1) : (x2 ? 64 + __builtin_ctz(x2) : (x3 ? 96 + __builtin_ctz(x3) : 128)));
}

// khash.find (RapMapUtils.hpp:65-67) for one key per lane in one round of loads; lanes that are not `on` read bucket 0.
// ck: the canonical word of the lane's k-mer (the smaller of it and its reverse complement), isr: the k-mer is the larger one --
// the two lanes that ask about a position's k-mer and about its reverse complement read the same 64-byte bucket.
QM_DEV void lean_find(const DevIndex& ix, const LV<u64>& ck, const LV<bool>& isr, const LV<bool>& on, LV<bool>& hit, LV<u32>& lb, LV<u32>& ub) {
  LV<bool> more; LV<u64> bkt;
  QM_LANES(l) {
    const u64 b = on[l] ? ((u64)bucket_hash(ck[l]) & ix.hmask) : 0ULL;
    U4 a, c;
    const unsigned char* bp = (const unsigned char*)&ix.slots[b];
    load_16x2(bp, bp + (isr[l] ? 32 : 16), a, c);
    QM_CNT(1, on[l] ? 1 : 0);
    const u64 k0r = ((u64)a.y << 32) | a.x, k1 = ((u64)a.w << 32) | a.z;
    // keys are 2k <= 62 bits: an empty entry (~0) and the overflow mark (bit 63) never equal one
    const bool h0 = (k0r & ~QM_BK_OVF) == ck[l], h1 = k1 == ck[l];
    const u32 vlb = h0 ? c.x : c.z;
    hit[l] = on[l] && (h0 || h1) && vlb != QM_IV_NONE;
    lb[l] = vlb; ub[l] = h0 ? c.y : c.w;
    more[l] = on[l] && !(h0 || h1) && k0r != ~0ULL && (k0r & QM_BK_OVF) != 0;
    bkt[l] = b;
  }
  // 0.4 % of the buckets: a key that hashes here lives in a later bucket.  A wave-level loop, no per-lane control flow: the lanes
  // that still look step to their next bucket together, the others read bucket 0 again
  while (ballot(more)) {
    QM_LANES(l) {
      const u64 b = more[l] ? ((bkt[l] + 1) & ix.hmask) : 0ULL;
      U4 a, c;
      const unsigned char* bp = (const unsigned char*)&ix.slots[b];
      load_16x2(bp, bp + (isr[l] ? 32 : 16), a, c);
      QM_CNT(1, more[l] ? 1 : 0);
      const u64 k0r = ((u64)a.y << 32) | a.x, k1 = ((u64)a.w << 32) | a.z;
      const bool h0 = (k0r & ~QM_BK_OVF) == ck[l], h1 = k1 == ck[l];
      const u32 vlb = h0 ? c.x : c.z;
      if (more[l] && (h0 || h1)) { hit[l] = vlb != QM_IV_NONE; lb[l] = vlb; ub[l] = h0 ? c.y : c.w; }
      more[l] = more[l] && !(h0 || h1) && k0r != ~0ULL && (k0r & QM_BK_OVF) != 0;
      bkt[l] = b;
    }
  }
}

// the same through the compact -p image: the pre-filter for the whole round (one sector per key, no per-lane control flow), then the
// BooPHF walk for the keys it lets through (present keys and 2e-4 of the absent ones).
// Round 6 -- a walk costs three sectors and most of them were wasted: a window that starts behind a sequencing error holds up to thirty
// k-mers that ARE in the index, the walk only ever uses the FIRST of them (the hit it goes on from) and the reverse complements in front of
// it (spotCheck_); whatever lies behind that hit is looked at again only if the next MMP ends inside the window, which it rarely does.  So
// with `upto` the k-mers are walked in position order up to the first one that is confirmed (lanes 0-31: the k-mers, lanes 32-63: the
// reverse complements of the same positions) and *upto says how many positions of the window were resolved: the caller cuts its
// window there (a position that falls out of a window and is asked for again is looked up again: look-ups are pure).  8.5 -> 3.5 walks per
// read on the benchmark's input (tests/emu, QM_PROFILE).
QM_DEV void lean_find_ph(const DevIndex& ix, const LV<u64>& key, const LV<u64>& krc, const LV<bool>& on, LV<bool>& hit, LV<u32>& lb, LV<u32>& ub, int* upto = nullptr) {
  LV<bool> want;
  QM_LANES(l) { want[l] = on[l]; hit[l] = false; lb[l] = 0; ub[l] = 0; }
  ph_filter_round(ix, key, krc, want);
  if (!upto) {
    QM_LANES(l) {
      bool h = false; u32 a = 0, b = 0;
      if (want[l]) h = find_kmer<QM_F_PH>(ix, key[l], a, b);
      QM_CNT(21, h ? 1 : 0); QM_CNT(22, on[l] ? 1 : 0);
      hit[l] = h; lb[l] = a; ub[l] = b;
    }
    return;
  }
  const u32 fcand = (u32)ballot(want);                       // positions whose k-mer passed the filter
  int done = 0;
  while (true) {
    const u32 rest = done < 32 ? (fcand >> done) : 0u;
    const int cutoff = rest ? done + ctz32(rest) : 32;       // the first candidate at or behind `done` (32: none)
    QM_LANES(l) {
      const int j = l & 31;
      bool h = false; u32 a = 0, b = 0;
      const bool mine = want[l] && j >= done && j <= cutoff;
      if (mine) h = find_kmer<QM_F_PH>(ix, key[l], a, b);
      QM_CNT(21, h ? 1 : 0);
      if (mine) { hit[l] = h; lb[l] = a; ub[l] = b; }
    }
    if (cutoff >= 32) { *upto = 32; return; }                // every position resolved
    if ((ballot(hit) >> cutoff) & 1ULL) { *upto = cutoff + 1; return; }
    done = cutoff + 1;                                       // (a false positive of the filter: on to the next candidate)
  }
}

// Probe positions [wb, wb + ww) (ww <= 32) of strand V of a read: lanes 0-31 ask for the k-mers, lanes 32-63 for their reverse
// complements.  Both words of a position come out of the read's two images by the same funnel shift -- the reverse complement
// of the k-mer at q is the k-mer at P - 1 - q of the other image -- and the two lanes of a position read the same bucket.
// pk2: the read's two images (8 words each), D = 128 - L: where reverseRead(read) starts in the second one.
// stride (a power of two): lane j asks about position wb + j * stride -- the -s walk, whose capped MMPs advance by exactly
// maxMMPExtension + 1 positions while the read keeps matching, never asks about the positions in between (probe_window), so ONE probe
// answers for every hit the walk can still make on a 100-bp read (round 6; until then a window spanned 32 positions, four of them asked).
// PH: the compact image of a -p index (FrugalBooMap::find over the BooPHF walk, find_kmer<QM_F_PH>, behind the membership pre-filter):
// the structure is keyed by the k-mer itself, so every lane looks up its own word.
template <bool PH, int IW = 8, bool NQ = false>
QM_DEV void lean_probe(const DevIndex& ix, const QM_LDS(u64)* pk2, const QM_LDS(u32)* nm, int D, int V, int P, int k, int wb, int ww, LeanWin& W, int stride = 1, int cut = 1) {
  const int sh = 31 - __builtin_clz((unsigned)stride);         // (stride: a power of two)
  { const int left = (P - wb + stride - 1) >> sh; if (ww > left) ww = left; }      // ww: positions asked for, wb + (j << sh) each
  QM_CNT(3, 1); QM_CNT(4, ww);
  LV<u64> ck, cr; LV<bool> isr, on, hit, inw, xk;
  const u32 kmask = (1u << k) - 1u;
  QM_LANES(l) {
    const int j = l & 31;
    const bool in = j < ww;
    const int q = in ? wb + (j << sh) : 0;
    bool clean = true;
    inw[l] = in; xk[l] = false;
    if (NQ) {
      const u32 nb = lean_nbits(nm, q);                    // an N inside the k-mer: not looked up (a position like any other that holds nothing);
      clean = (nb & kmask) == 0;                            // one right behind it: the first-hit scan passes over that k-mer as well (:175 `<=`)
      xk[l] = in && l < 32 && ((nb >> k) & 1u) != 0;
    }
    const u64 w = lean_kmer(pk2 + IW * V, q + (V ? D : 0), k), wr = lean_kmer(pk2 + IW * (1 - V), P - 1 - q + (V ? 0 : D), k);
    if (PH) {
      const bool comp = l >= 32;                           // lanes 32-63: the reverse complement = the other image's k-mer at P - 1 - q
      ck[l] = comp ? wr : w; cr[l] = comp ? w : wr;        // (the structure is keyed by the k-mer itself; the pre-filter's word by the canonical one)
      isr[l] = false;
    } else {
      const bool big = wr < w;                             // the strand's k-mer is the larger of the two
      ck[l] = big ? wr : w;
      isr[l] = big != (l >= 32);                           // lanes 32-63 ask for the other orientation
    }
    on[l] = in && clean;
  }
  int upto = 32;
  // (cut = 0: the -s collector -- its strided windows hold the NEXT hits of the walk on purpose, lean_iter checks up to four of them together)
  if (PH) lean_find_ph(ix, ck, cr, on, hit, W.lb, W.ub, cut ? &upto : nullptr);
  else lean_find(ix, ck, isr, on, hit, W.lb, W.ub);
  const u64 fm = ballot(hit);
  u32 km = (u32)ballot(inw);
  if (PH && upto < ww) { ww = upto; km &= (1u << upto) - 1u; }      // the window ends behind the first confirmed k-mer (lean_find_ph)
  W.Fm = (u32)fm; W.Cm = (u32)(fm >> 32); W.Km = km; W.Xm = NQ ? (u32)ballot(xk) : 0u; W.wb = wb; W.ww = ww; W.sh = sh;
}

// hitsToMappingsSimple (HitManager.cpp:691-882) for one strand whose intervals hold n <= 64 suffixes, in registers.  Lane l
// holds suffix l.  Its transcript survives when it was seen in every interval (:669-678, slack 0: a bit per interval), and it
// is represented by the entry with the smallest position, the earlier interval in processing order on ties (the first smallest
// interval first: :636-641, :309-313; one interval: the stable sort + std::unique of :757-803).  Survivors go out in ascending
// transcript order: slot = number of surviving entries with a smaller transcript id.
QM_DEV int lean_h2m(const QM_LDS(LeanSuf)* suf, const LeanStrand& S, bool isRC, LV<u64>& elem, LV<bool>& keep, LV<int>& slot) {
  const int n = S.sufN, m = S.n;
  QM_CNT(12, 1); QM_CNT(13, n);
  LV<u32> tid, pos, ordl, ivb, seen, less;
  QM_LANES(l) {
    const QM_LDS(LeanSuf)* e = suf + (l < n ? l : 0);
    const u32 t = e->tid, ps = e->pos, qp = e->qp, iv = e->iv;
    tid[l] = l < n ? t : 0xffffffffu; pos[l] = ps;
    const u32 ord = iv == (u32)S.minIdx ? 0u : (iv < (u32)S.minIdx ? iv + 1u : iv);
    ordl[l] = ord * 64u + (u32)l;
    ivb[l] = 1u << iv; seen[l] = ivb[l]; less[l] = 0;
    elem[l] = mk_elem(t, isRC, (int)(ps - qp));          // pos - queryPos (:315, :761-767)
    keep[l] = l < n;
  }
  for (int j = 0; j < n; ++j) {
    const u32 tj = read_lane(tid, j), pj = read_lane(pos, j), oj = read_lane(ordl, j), bj = read_lane(ivb, j);
    QM_LANES(l) {
      const bool same = tj == tid[l];
      if (same && (pj < pos[l] || (pj == pos[l] && oj < ordl[l]))) keep[l] = false;
      less[l] += tj < tid[l] ? 1u : 0u;
      if (same) seen[l] |= bj;
    }
  }
  const u32 all = m >= 32 ? 0xffffffffu : ((1u << m) - 1u);
  QM_LANES(l) { keep[l] = keep[l] && seen[l] == all; }
  const u64 km = ballot(keep);
  const int cnt = popc64(km);
  if (cnt == n) { QM_LANES(l) { slot[l] = (int)less[l]; } }          // nothing dropped: every transcript once
  else {
    QM_LANES(l) { slot[l] = 0; }
    for (u64 r = km; r; r &= r - 1) {
      const u32 tj = read_lane(tid, ctz64(r));
      QM_LANES(l) { slot[l] += tj < tid[l] ? 1 : 0; }
    }
  }
  return cnt;
}

struct PairCtr { u32 pe, se, tot, reads, tooMany, mapped; };   // HitCounters of a pair merged in its wavefront / of all such pairs of a wave (wave-uniform)

// A pair finished in the wavefront that mapped both its mates: mergeLeftRightHits (RapMapUtils.hpp:1185-1264) + the per-pair driver
// (RapMapSAMapper.cpp:527-551,684-701) on the two sorted lists -- unit_merge.  lst: the two lists in LDS in list order, [0, cntA) the left
// mate's, [32, 32 + cntB) the right one's (at most 32 elements each); lane i takes left element i and looks for its transcript among the
// right ones.  Out: pair_cnt[pair] = the pair's hits, lcnt[2 pair] = QM_LCNT_PAIR, loff[2 pair] = where its records sit in B.lists -- two
// words per hit, {left element, right element} or {element, QM_DUO_ORPHAN | MateStatus} (duo_hit, qm_mapper.inl) --, the HitCounters into ctr.
QM_DEV void pair_merge(const ReadBatch& B, int pair, const QM_LDS(u64)* lst, int cntA, int cntB, WaveAlloc& wa, PairCtr& ctr) {
  ctr.pe = 0; ctr.se = 0;
  const int maxHits = B.max_num_hits;
  LV<u64> mine, partner; LV<bool> fnd;
  QM_LANES(l) {
    const int h = l >> 5, i = l & 31;
    mine[l] = lst[32 * h + (i < (h ? cntB : cntA) ? i : 0)];
    partner[l] = 0; fnd[l] = false;
  }
  if (cntA > 0) {
    for (int jj = 0; jj < cntB; ++jj) {
      QM_LANES(l) {
        const u64 f = lst[32 + jj];
        if (l < cntA && el_tid(f) == el_tid(mine[l])) { partner[l] = f; fnd[l] = true; }
      }
    }
  }
  const int nm = popc64(ballot(fnd));
  const int tooMany = nm > maxHits ? 1 : 0;               // :1233-1234
  LV<bool> kp; LV<u64> w1;
  int cnt = 0;
  if (!tooMany && nm > 0) {
    QM_LANES(l) {
      bool kq = fnd[l];
      if (kq && B.no_dovetail) {                          // RapMapSAMapper.cpp:684-698 on the hit paired_hit() would make
        const int s1 = el_pos(mine[l]) > 0 ? el_pos(mine[l]) : 0, s2 = el_pos(partner[l]) > 0 ? el_pos(partner[l]) : 0;
        const bool fwd = !el_rc(mine[l]), mfwd = !el_rc(partner[l]);
        if (fwd != mfwd && ((fwd && s1 > s2) || (mfwd && s2 > s1))) kq = false;
      }
      kp[l] = kq; w1[l] = partner[l];
    }
    ctr.pe = (u32)nm;
  } else {
    const int no = cntA + cntB;
    const int keepAll = (!tooMany && no > 0 && no <= maxHits && !B.no_orphans) ? 1 : 0;   // RapMapSAMapper.cpp:534-551
    if (!tooMany && no > 0) ctr.se = (u32)no;
    QM_LANES(l) {
      const int h = l >> 5, i = l & 31;
      bool kq = keepAll && i < (h ? cntB : cntA);
      // --noDovetail on orphans: matePos = 0, mateIsFwd = true (unit_merge, the oracle): a reverse-strand hit left of the transcript's start
      if (kq && B.no_dovetail && el_rc(mine[l]) && el_pos(mine[l]) < 0) kq = false;
      kp[l] = kq; w1[l] = QM_DUO_ORPHAN | (u64)(h ? 2 : 1);
    }
  }
  const u64 kmask = ballot(kp);
  cnt = popc64(kmask);
  long long base = 0;
  if (cnt > 0) {
    const int nwd = 2 * cnt;
    if (wa.base < 0 || wa.used + nwd > QM_LEAN_CHUNK) {
      LV<u64> bv;
      QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor, (u64)QM_LEAN_CHUNK); }
      wa.base = (long long)read_lane(bv, 0); wa.used = 0;
    }
    base = wa.base + wa.used;
    if (base + nwd > B.lists_cap) { QM_LANES(l) { if (l == 0) *B.status |= 1; } cnt = 0; base = 0; }
    else wa.used += nwd;
  }
  if (cnt > 0) {
    QM_LANES(l) {
      if (kp[l]) { const int rk = popc64(kmask & lanemask_lt(l)); B.lists[base + 2 * rk] = mine[l]; B.lists[base + 2 * rk + 1] = w1[l]; }
    }
  }
  ctr.reads = 1; ctr.tooMany = (u32)tooMany; ctr.tot = (u32)cnt; ctr.mapped = cnt > 0 ? 1u : 0u;       // (this pair's share of the HitCounters: the caller adds it up)
  QM_LANES(l) { if (l == 0) { B.pair_cnt[pair] = (u32)cnt; B.lcnt[2 * pair] = QM_LCNT_PAIR; B.loff[2 * pair] = base; } }
}

// a read for the general kernel: marked in its list-length word and counted (the host gathers the marks into a queue, like the
// reads the general kernels set aside for the long-read pass); nothing else was written for it
QM_DEV void lean_defer(const ReadBatch& B, int read, int why) {
  QM_CNT(19, 1);
  QM_LANES(l) {
    if (l == 0) {
      B.lcnt[read] = QM_LCNT_LEAN; B.loff[read] = 0; atomic_add_u64(B.cursor + QM_SC_LEANQ, 1ULL);
#ifndef QM_TIMING
      atomic_add_u64(B.cursor + QM_SC_DEFER0 + why, 1ULL);
#endif
    }
  }
}

// offsets of iteration `it` into ostage[par]: pairs: off1[it], off1[it + 1], off2[it], off2[it + 1]; single-end reads 2 it and
// 2 it + 1: off1[2 it .. 2 it + 2] (the last one only if the second read exists).  Iterations, like reads, are below 2^31 per launch.
template <bool PAIRED, bool WIDE = false, bool NQ = false>
QM_DEV void lean_stage_offsets(const ReadBatch& B, int it, int nit, LeanMem& M, int par) {
  if (it >= nit) return;
  if (NQ) {
    // the N-aware pass over a queue: slots 2 it and 2 it + 1 name the reads (pairs: read r = mate r & 1 of pair r >> 1); their offsets go where
    // those of a pair's mates go -- dwords 0-3 and 4-7
    QM_LANES(l) {
      const long long s = 2LL * it + (l >> 2);
      if (l < 8 && s < (long long)B.nreads) {
        const long long r = B.slowq[s];
        const long long* o = PAIRED ? (((r & 1) ? B.off2 : B.off1) + (r >> 1)) : (B.off1 + r);
        lds_dma_u32((const u32*)o + (l & 3), M.ostage[par], l);
      }
    }
    return;
  }
  if (WIDE) {                                              // one read per iteration: its two offsets (pairs: read it = mate it & 1 of pair it >> 1)
    QM_LANES(l) {
      if (l < 4) { const long long* o = PAIRED ? (((it & 1) ? B.off2 : B.off1) + (it >> 1)) : (B.off1 + it); lds_dma_u32((const u32*)o + l, M.ostage[par], l); }
    }
    return;
  }
  const int nd = PAIRED ? 8 : (2 * it + 1 < (int)B.nreads ? 6 : 4);
  QM_LANES(l) {
    if (l < nd) {
      const long long* o = PAIRED ? ((l < 4 ? B.off1 : B.off2) + it) : (B.off1 + 2 * it);
      lds_dma_u32((const u32*)o + (PAIRED ? (l & 3) : l), M.ostage[par], l);
    }
  }
}
QM_DEV long long lean_off64(const LV<u32>& ov, int d) { return (long long)(((u64)read_lane(ov, d + 1) << 32) | (u64)read_lane(ov, d)); }
// the offsets in ostage[par] (landed) into the request for the two reads' characters
template <bool PAIRED, bool WIDE = false, bool NQ = false>
QM_DEV void lean_stage_chars(const ReadBatch& B, int it, int nit, LeanMem& M, int par) {
  if (it >= nit) return;
  if (WIDE) {
    LV<u32> ovw;
    QM_LANES(l) { ovw[l] = M.ostage[par][l & 7]; }
    const long long o0 = lean_off64(ovw, 0);
    int len = (int)(read_lane(ovw, 2) - (u32)o0);
    if (len > 2 * QM_LEAN_MAXLEN) len = 2 * QM_LEAN_MAXLEN;
    const unsigned char* p = ((PAIRED && (it & 1)) ? B.seq2 : B.seq1) + o0;
    const int mis = (int)((unsigned long long)p & 3ULL);
    const u32* g = (const u32*)(p - mis);
    const int nd = (mis + len + 3) >> 2;                  // <= 65: the rows of both reads of the narrow edition, one behind the other, hold them
    u32* row = &M.stage[0][0];
    QM_LANES(l) { if (l < nd) lds_dma_u32(g + l, row, l); }
    if (nd > 64) { QM_LANES(l) { if (l == 0) lds_dma_u32(g + 64, row + 64, 0); } }
    return;
  }
  const bool have1 = 2 * it + 1 < (int)B.nreads;
  LV<u32> ov;
  QM_LANES(l) { ov[l] = M.ostage[par][l & 7]; }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h && !have1) break;
    const int d0 = h ? ((PAIRED || NQ) ? 4 : 2) : 0;
    const long long o0 = lean_off64(ov, d0);
    int len = (int)(read_lane(ov, d0 + 2) - (u32)o0);
    if (len > QM_LEAN_MAXLEN) len = QM_LEAN_MAXLEN;
    const bool second = NQ ? (PAIRED && (B.slowq[2LL * it + h] & 1) != 0) : (h && PAIRED);
    const unsigned char* p = (second ? B.seq2 : B.seq1) + o0;
    const int mis = (int)((unsigned long long)p & 3ULL);
    const u32* g = (const u32*)(p - mis);
    const int nd = (mis + len + 3) >> 2;                  // <= 33
    QM_LANES(l) { if (l < nd) lds_dma_u32(g + l, M.stage[h], l); }
  }
}

// One iteration: reads 2 it and 2 it + 1.  Wave-uniform flags are ints on purpose: a bool that lives across a branch is kept as a
// 64-bit lane mask by this compiler (three scalar instructions per test instead of a compare), and the scalar unit is what these
// kernels run out of first.
// WIDE: ONE read per iteration over all 64 lanes (up to 256 characters; iteration it = read it); the images of the two strands are 16
// words apart instead of 8, the MMP extension reads the 224-character table (SaExt2).  The window probes, the walk and hits->mappings
// are the same code.
template <bool PAIRED, bool SEL, bool PH, bool WIDE = false, bool NQ = false>
QM_DEV void lean_iter(const DevIndex& ix, const ReadBatch& B, int it, int nit, int nw, int par, LeanMem& M, WaveAlloc& wa) {
  constexpr int IW = WIDE ? 16 : 8;                        // words between the two images of a read
  constexpr int HW = WIDE ? 64 : 32;                       // lanes per read
  constexpr int MAXLEN = WIDE ? 2 * QM_LEAN_MAXLEN : QM_LEAN_MAXLEN;
  constexpr int NH = WIDE ? 1 : 2;                         // reads per iteration
  const int k = ix.k;
  const int r0 = WIDE ? it : 2 * it;
  const int have1 = (!WIDE && r0 + 1 < (int)B.nreads) ? 1 : 0;
  // ---- the two reads' characters -> 2-bit images of both strands, four characters per lane
  LV<u32> ov;
  QM_LANES(l) { ov[l] = M.ostage[par][l & 7]; }
  const u32 a0 = read_lane(ov, 0), a1 = read_lane(ov, 2);
  const u32 b0 = read_lane(ov, (PAIRED || NQ) ? 4 : 2), b1 = read_lane(ov, (PAIRED || NQ) ? 6 : 4);
  // the reads of the iteration: 2 it and 2 it + 1, or (NQ: the N-aware pass over the queue of the reads the first pass left) what those slots name
  const int ra = NQ ? (int)B.slowq[r0] : r0, rb = NQ ? (have1 ? (int)B.slowq[r0 + 1] : 0) : r0 + 1;
  const bool sec0 = NQ ? (PAIRED && (ra & 1)) : (WIDE && PAIRED && (it & 1)), sec1 = NQ ? (PAIRED && (rb & 1)) : PAIRED;
  const int raw0 = (int)(a1 - a0), raw1 = have1 ? (int)(b1 - b0) : 0;
  const int len0 = raw0 > MAXLEN ? MAXLEN : raw0, len1 = raw1 > MAXLEN ? MAXLEN : raw1;
  const int mis0 = (int)(((u32)(unsigned long long)(sec0 ? B.seq2 : B.seq1) + a0) & 3u);
  const int mis1 = (int)(((u32)(unsigned long long)(sec1 ? B.seq2 : B.seq1) + b0) & 3u);
  QM_LDS(unsigned char)* PKb = (QM_LDS(unsigned char)*)&M.pk[0][0][0];
  LV<bool> bad, rep;
  QM_LANES(l) {
    const int h = WIDE ? 0 : (l >> 5), jj = l & (HW - 1), base = 4 * jj;
    const int len = h ? len1 : len0, mis = h ? mis1 : mis0;
    const u32* srow = WIDE ? &M.stage[0][0] : &M.stage[h][0];
    const u32 w0 = srow[jj], w1 = srow[jj + 1];
    const u32 d = align_bytes(w1, w0, mis);                                   // characters base .. base + 3
    const int nb = len - base;
    const u32 lenmask = nb >= 4 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
    const u32 t = (d & 0xdfdfdfdfu) ^ canon4(d, false);
    const u32 valid = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu) & lenmask;       // 0x80: A C G T in either case
    bad[l] = (~valid & lenmask & 0x80808080u) != 0;
    const u32 x = (d >> 1) & 0x03030303u;
    const u32 code = (x ^ ((x >> 1) & 0x01010101u)) & ((valid >> 7) * 3u);     // A0 C1 G2 T3 (Kmer.hpp:40-51)
    const u32 pk = (code * 0x40100401u) >> 24;                                 // first character in the top bits
    rep[l] = nb >= 4 && ((pk ^ (pk >> 2)) & 0x3fu) == 0;
    // the same four bases reverse-complemented (Kmer.hpp:92-100 on a byte): order reversed, every code inverted
    u32 r = brev32(pk) >> 24;
    r = (~(((r >> 1) & 0x55u) | ((r & 0x55u) << 1))) & 0xffu;
    const int img = 16 * IW * h;                                               // bytes: image (h, strand) starts at 16 IW h + 8 IW strand
    PKb[img + 8 * (jj >> 3) + 7 - (jj & 7)] = (unsigned char)pk;
    const int mj = HW - 1 - jj;
    PKb[img + 8 * IW + 8 * (mj >> 3) + 7 - (mj & 7)] = (unsigned char)r;
  }
  const u64 dirty = ballot(bad), reps = ballot(rep);
  // a read whose characters outside A C G T are all N / n stays here: the N flags of its two strands go to M.nm (lean_nbits).  Anything
  // else in it and it is left to the general kernel as before (what the reference makes of such a character is Kmer::fromChars' business)
  int nn0 = 0, nn1 = 0;
  if (NQ && !WIDE && dirty != 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!(u32)(dirty >> (32 * h))) continue;
      const int len = h ? len1 : len0, mis = h ? mis1 : mis0;
      const QM_LDS(unsigned char)* sb = (const QM_LDS(unsigned char)*)&M.stage[h][0] + mis;
      LV<bool> n0, n1, x0, x1;
      QM_LANES(l) {
        const u32 c0 = l < len ? (u32)sb[l] : (u32)'A', c1 = l + 64 < len ? (u32)sb[l + 64] : (u32)'A';
        const u32 u0 = c0 & 0xdfu, u1 = c1 & 0xdfu;
        n0[l] = u0 == 'N'; n1[l] = u1 == 'N';
        x0[l] = !(u0 == 'A' || u0 == 'C' || u0 == 'G' || u0 == 'T' || u0 == 'N');
        x1[l] = !(u1 == 'A' || u1 == 'C' || u1 == 'G' || u1 == 'T' || u1 == 'N');
      }
      const u64 nlo = ballot(n0), nhi = ballot(n1);
      if ((ballot(x0) | ballot(x1)) != 0) continue;
      // reverseRead(read): position q holds what position len - 1 - q held -- the 128 flags bit-reversed, shifted down by 128 - len
      const u64 bl = ((u64)brev32((u32)nhi) << 32) | (u64)brev32((u32)(nhi >> 32)), bh = ((u64)brev32((u32)nlo) << 32) | (u64)brev32((u32)(nlo >> 32));
      const int sft = 128 - len;                             // 0 .. 127
      const u64 rlo = sft >= 64 ? (bh >> (sft - 64)) : (sft ? ((bl >> sft) | (bh << (64 - sft))) : bl);
      const u64 rhi = sft >= 64 ? 0ULL : (bh >> sft);
      QM_LANES(l) {
        if (l < 8) {
          const u64 w = (l & 4) ? ((l & 2) ? rhi : rlo) : ((l & 2) ? nhi : nlo);
          M.nm[h][l >> 2][l & 3] = (u32)(w >> (32 * (l & 1)));
        }
      }
      if (h) nn1 = 1; else nn0 = 1;
    }
  }
  wave_fence();
  // the staging rows are free again: the next iteration's characters, and the offsets of the one after it
  lean_stage_chars<PAIRED, WIDE, NQ>(B, it + nw, nit, M, par ^ 1);
  lean_stage_offsets<PAIRED, WIDE, NQ>(B, it + 2 * nw, nit, M, par);
#if defined(QM_LEAN_ABLATE) && QM_LEAN_ABLATE == 1    // profiling builds (profiles/r05/ablate_build.sh): the phases up to here, empty lists out
  lds_dma_wait();
  QM_LANES(l) { if (l < 2 && r0 + l < (int)B.nreads) { B.lcnt[r0 + l] = (u32)(dirty & reps & 1); B.loff[r0 + l] = 0; } }
  return;
#endif
  // what this kernel takes: no character but A C G T, no window of k equal bases (k equal characters cover at least (k - 6) / 4
  // whole lanes above: setup_strand's rule for its lazy strands), at most 128 characters
  // (1: a character that is not A C G T or too many of them, 2: a window of k equal bases -- lean_defer's `why` + 1)
  // (bit 4 of the codes: the read has N's and M.nm holds their flags)
  const int defer0 = WIDE ? ((raw0 > MAXLEN || dirty != 0) ? 1 : (4 * popc64(reps) + 6 >= k ? 2 : 0))
                          : (((raw0 > MAXLEN || ((u32)dirty != 0 && !nn0)) ? 1 : (4 * popc32((u32)reps) + 6 >= k ? 2 : 0)) | (nn0 << 4));
  const int defer1 = ((raw1 > MAXLEN || ((u32)(dirty >> 32) != 0 && !nn1)) ? 1 : (4 * popc32((u32)(reps >> 32)) + 6 >= k ? 2 : 0)) | (nn1 << 4);
  const int P0 = len0 - k + 1, P1 = len1 - k + 1;
  const int ok0 = (!(defer0 & 15) && P0 >= 1) ? 1 : 0, ok1 = (have1 && !(defer1 & 15) && P1 >= 1) ? 1 : 0;
  // ---- the first probe of both reads in one round (SACollector.hpp:167-237 starts at position 0; the read's last k-mer is the
  // first thing the reverse-complement pass asks for): lanes 0-3 of each half = the read's k-mer 0, its k-mer P - 1, and --
  // from the second image -- the reverse complements of those two
  LV<u64> ck, cr; LV<bool> isr, on, hit, x0k; LV<u32> plb, pub;
  QM_LANES(l) {
    const int h = WIDE ? 0 : (l >> 5), jj = l & 31;
    const int P = h ? P1 : P0, D = MAXLEN - (h ? len1 : len0);
    const bool lastq = jj == 1 || jj == 2;                 // jj 0: read[0]  1: read[P-1]  2: rc[P-1] (= complement of read[0])  3: rc[0]
    bool o = (h ? ok1 : ok0) != 0 && jj < 4 && (P > 1 || !(jj & 1)) && (!WIDE || l < 32);
    const int q = (o && lastq) ? P - 1 : 0;                // position in the lane's own strand (jj >> 1) ...
    const int qo = o ? P - 1 - q : 0;                      // ... and of the reverse complement in the other one
    x0k[l] = false;
    if (NQ) {                                              // (a k-mer with an N in it is not looked up: lean_nbits)
      const QM_LDS(u32)* nmh = (((h ? defer1 : defer0) >> 4) && !WIDE) ? (const QM_LDS(u32)*)&M.nm[h][(jj >> 1) & 1][0] : (const QM_LDS(u32)*)&M.nmz[0];
      const u32 nb = lean_nbits(nmh, q);
      x0k[l] = o && jj == 0 && ((nb >> k) & 1u) != 0;
      o = o && (nb & ((1u << k) - 1u)) == 0;
    }
    const QM_LDS(u64)* pkh = (const QM_LDS(u64)*)&M.pk[0][0][0] + 2 * IW * h;
    const bool s = (jj >> 1) & 1;
    const u64 w = lean_kmer(pkh + (s ? IW : 0), q + (s ? D : 0), k);
    const u64 wr = lean_kmer(pkh + (s ? 0 : IW), qo + (s ? 0 : D), k);
    if (PH) { ck[l] = w; cr[l] = wr; isr[l] = false; }       // (the lane's own word: jj 2 / 3 read the second image)
    else {
      const bool big = wr < w;
      ck[l] = big ? wr : w; isr[l] = big;
    }
    on[l] = o;
  }
  QM_CNT(3, 1);
  if (PH) lean_find_ph(ix, ck, cr, on, hit, plb, pub);
  else lean_find(ix, ck, isr, on, hit, plb, pub);
  const u64 fm0 = ballot(hit), xm0 = NQ ? ballot(x0k) : 0ULL;
  lds_dma_wait();                                          // what was requested above has landed by now: no store follows an open request
#if defined(QM_LEAN_ABLATE) && QM_LEAN_ABLATE == 2
  QM_LANES(l) { if (l < 2 && r0 + l < (int)B.nreads) { B.lcnt[r0 + l] = (u32)(fm0 & 1 & plb[l] & pub[l]); B.loff[r0 + l] = 0; } }
  return;
#endif
  const int useCov = B.strict_check != 0 ? 1 : 0;          // disableNIP_ && strictCheck_ (SACollector.hpp:138)
  const u32 maxIv = (u32)B.max_interval;
#pragma nounroll
  for (int h = 0; h < NH; ++h) {
    if (h > have1) break;
    const int read = h ? rb : ra;
    if ((h ? defer1 : defer0) & 15) { lean_defer(B, read, ((h ? defer1 : defer0) & 15) - 1); continue; }
    const int L = h ? len1 : len0, P = L - k + 1, D = MAXLEN - L;
    int n = 0, foundHit = 0, bail = 0, selV = 0;
    LV<u64> elem; LV<bool> keep; LV<int> slot;
    QM_LANES(l) { keep[l] = false; slot[l] = 0; elem[l] = 0; }
    if (P >= 1) {
      const QM_LDS(u64)* pk2 = (const QM_LDS(u64)*)&M.pk[0][0][0] + 2 * IW * h;
      const u32 fmh = (u32)(fm0 >> (32 * h));
      const u32 F0 = fmh & 1u, C0 = (fmh >> 2) & 1u;
      const QM_LDS(u32)* nm0 = (NQ && !WIDE && ((h ? defer1 : defer0) >> 4)) ? (const QM_LDS(u32)*)&M.nm[h][0][0] : (const QM_LDS(u32)*)&M.nmz[0];   // the N flags of the read (of reverseRead(read): QM_LEAN_NMW words on)
      const int nmStep = (NQ && !WIDE && ((h ? defer1 : defer0) >> 4)) ? QM_LEAN_NMW : 0;
      LeanWin W; W.wb = 0; W.ww = 1; W.sh = 0; W.Fm = F0; W.Cm = C0; W.Km = 1u; W.Xm = (u32)(xm0 >> (32 * h)) & 1u;
      { const u32 s0lb = read_lane(plb, 32 * h), s0ub = read_lane(pub, 32 * h); QM_LANES(l) { W.lb[l] = s0lb; W.ub[l] = s0ub; } }
      // first-hit scan (:167-237): the first position whose k-mer or reverse complement is in the hash
      int p0 = 0;
      while (p0 < P) {
        if ((unsigned)(p0 - W.wb) >= (unsigned)W.ww) lean_probe<PH, IW, NQ>(ix, pk2, nm0, D, 0, P, k, p0, 32, W, 1, SEL ? 0 : 1);
        const u32 mm = ((W.Fm | W.Cm) & ~W.Xm) >> (p0 - W.wb);
        if (mm) { p0 += ctz32(mm); foundHit = 1; break; }
        p0 = W.wb + W.ww;
      }
      if (foundHit) {
        // ONE walk per read: the read itself from its first hit when that hit is a forward one (:247-254), else reverseRead(read)
        // from 0 (:258-265).  ha / hb: hits of the walked strand / of the other one.  A read that then asks for the other strand as
        // well (:258, :271: k-mers of the other orientation seen on the way) is rare and left to the general kernel.
        const int rel0 = p0 - W.wb;
        const int V = ((W.Fm >> rel0) & 1u) ? 0 : 1;
        u32 ha = 1u, hb = V ? 0u : ((W.Cm >> rel0) & 1u);
        int p = 0, skip = 0;
        u32 lb = 0, ub = 0;
        if (V == 0) { p = p0; skip = 1; lb = read_lane(W.lb, rel0); ub = read_lane(W.ub, rel0); }
        else {
          // what the first probe learned about the read's last k-mer is the first k-mer of reverseRead(read)
          const u32 Fl = P > 1 ? (fmh >> 1) & 1u : F0, Cl = P > 1 ? (fmh >> 3) & 1u : C0;
          const int rl = 32 * h + (P > 1 ? 3 : 2);
          const u32 rlb = read_lane(plb, rl), rub = read_lane(pub, rl);
          W.wb = 0; W.ww = 1; W.sh = 0; W.Fm = Cl; W.Cm = Fl; W.Km = 1u;
          QM_LANES(l) { W.lb[l] = rlb; W.ub[l] = rub; }
        }
        // ---- SACollector::getSAHits_ (SACollector.hpp:441-677, NIP disabled) over a clean strand: every position below P is
        // eligible, every window is A C G T.  The MMP extension (SASearcher.hpp:88-309) is the closed form of extend_search_wide
        // against the packed characters behind every suffix's k-mer (one lane per suffix, one 32-byte load each); the
        // (transcript, position) words of the block it settles on go to M.suf.
        QM_LDS(LeanSuf)* suf = (QM_LDS(LeanSuf)*)M.suf;
        QM_LDS(IntRec)* ints = (QM_LDS(IntRec)*)M.ints;
        const int imgOff = V ? D : 0;
        const QM_LDS(u32)* nmV = nm0 + (V ? nmStep : 0);
        const int ext = SEL ? B.max_mmp_ext : 0;                     // -s: every MMP but a read's first is cut at k + maxMMPExtension (:557-575)
        int lastSearch = 0, prevEnd = 0, width = 1, spot = 0, stopAfter = 0, pstride = 1;
        int sn = 0, sufN = 0, minIdx = 0, minSpan = 0x7fffffff, cov = 0;
        QM_CNT(17, 1);
        while (true) {
          if (!skip) {
            if (p >= P) break;
            {
              const unsigned relp = (unsigned)(p - W.wb);
              const bool known = (relp & ((1u << W.sh) - 1u)) == 0 && (relp >> W.sh) < (unsigned)W.ww;
              if (!known) lean_probe<PH, IW, NQ>(ix, pk2, nmV, D, V, P, k, p, width, W, SEL ? pstride : 1, SEL ? 0 : 1);
            }
            width = 32; pstride = 1;
            const int rel = (p - W.wb) >> W.sh;
            u32 fm = W.Fm >> rel, cm = W.Cm >> rel;                   // (no bits beyond the window)
            int avail = W.ww - rel;
            if (SEL && W.sh) { avail = 1; fm &= 1u; cm &= 1u; }       // a strided window: the position behind p was not asked about
            if (spot) {                                               // the k-mer the walk goes on with, spot-checked (:602-611)
              ha += fm & 1u; hb += cm & 1u; spot = 0;
              if (stopAfter) break;
            }
            const u32 below = (fm & (0u - fm)) - 1u;                  // the positions before the first hit (all of them without one)
            hb += (u32)popc32(cm & ~fm & below);                      // misses: spotCheck_ of the complement (:667-675)
            if (!fm) { p += avail; continue; }
            const int ph = ctz32(fm);
            ha += 1;                                                  // spotCheck_ on the hit (:545)
            hb += (cm >> ph) & 1u;
            p += ph;
            lb = read_lane(W.lb, (p - W.wb) >> W.sh); ub = read_lane(W.ub, (p - W.wb) >> W.sh);
          }
          skip = 0;
          // -s, a run of capped MMPs in one round: while the read keeps matching, the hits of the walk are maxMMPExtension + 1 positions
          // apart and each is cut at k + maxMMPExtension, so the next ones are known before this one is extended (the strided probe asked
          // about all of them).  Up to eight hits at a time, eight lanes each -- or four with sixteen lanes each when an interval among
          // the first four is wider than eight suffixes --: the lanes check the extension's characters against the narrow table's entries
          // of the hit's suffixes, all hits at once, and the leading hits that do match that far are recorded together -- what the loop
          // below would have done one hit, one round of loads and ~150 scalar instructions at a time -- and the walk goes on behind them.
          // Per-hit values live in lanes 0-7 (no scalar arrays).  Anything else (a hit whose extension stops short, a wide interval, the
          // read's end) takes the loop.
          // (a hit the walk found by itself -- behind an error -- in a window that does not hold the walk's next position: the strided probe
          // that would follow its extension is asked for now, and the hit starts a run instead of taking a round of the loop on its own)
          if (SEL && p != 0 && !lastSearch && ext >= 1 && ext <= QM_NEXT_BASES && ix.sanext && ((ext + 1) & ext) == 0 && ext + 1 <= 16 &&
              W.sh == 0 && (p - W.wb) + ext + 1 >= W.ww && p + k + ext < L && p + ext + 1 < P)
            lean_probe<PH, IW, NQ>(ix, pk2, nmV, D, V, P, k, p, 32, W, ext + 1, 0);
          if (SEL && p != 0 && !lastSearch && ext >= 1 && ext <= QM_NEXT_BASES && ix.sanext && ((ext + 1) & ext) == 0 && ext + 1 <= 16 &&
              ((ext + 1) & ((1 << W.sh) - 1)) == 0) {
            const int st = ext + 1, mlenC = k + ext;
            const int relb0 = (p - W.wb) >> W.sh, step = st >> W.sh;  // hit j of the run: the window's bit / lane relb0 + j * step
            LV<int> idxv;
            QM_LANES(l) { idxv[l] = (relb0 + (l & 7) * step) & 31; }
            LV<u32> av, bv;
            wave_read(W.lb, idxv, av); wave_read(W.ub, idxv, bv);
            LV<u32> aiV, wV; LV<bool> okb, nwb, tlb;
            QM_LANES(l) {
              const int jh = l & 7, idx = relb0 + jh * step, pj = p + jh * st;
              const u32 ai = av[l] ? av[l] - 1 : 0;
              const int w = (int)(bv[l] - ai - 1);
              // (a hit whose cut MMP would reach the read's end -- the walk's last -- may close the run: `tail`)
              const bool ok = l < 8 && idx < W.ww && idx < 32 && ((W.Fm >> (idx & 31)) & 1u) != 0 && pj + k <= L && sn + jh < QM_LEAN_MAXIV && w >= 1 && w <= 16;
              aiV[l] = ok ? ai : 0u; wV[l] = ok ? (u32)w : 0u; okb[l] = ok; nwb[l] = ok && w <= 8;      // (lanes without a hit: entry 1 of the table, ignored)
              tlb[l] = ok && pj + mlenC >= L;
            }
            const u32 okm = (u32)ballot(okb) & 0xffu, nwm = (u32)ballot(nwb) & 0xffu, tlm = (u32)ballot(tlb) & 0xffu;
            const int T = ctz32(tlm);                                   // the first hit that would reach the read's end (32: none in sight)
            int J16 = ctz32(~okm); J16 = J16 > 4 ? 4 : J16; J16 = J16 > T + 1 ? T + 1 : J16;
            int J8 = ctz32(~nwm); J8 = J8 > T + 1 ? T + 1 : J8;
            const int lsh = J8 > J16 ? 3 : 4;                          // lanes per hit: 8 or 16
            const int J = J8 > J16 ? J8 : J16;
            if (J >= 2) {
              LV<int> gv;
              QM_LANES(l) { gv[l] = l >> lsh; }
              LV<u32> aiG, wG;
              wave_read(aiV, gv, aiG); wave_read(wV, gv, wG);
              LV<bool> fullb; LV<int> mv;
              QM_LANES(l) {
                const int g = l >> lsh, sI = l & ((1 << lsh) - 1);
                const bool act = g < J && sI < (int)wG[l];
                const u32 qn = (u32)lean_kmer(pk2 + IW * V, p + g * st + k + imgOff, ext);
                const u32 e = ix.sanext[aiG[l] + 1 + (u32)(act ? sI : 0)];
                const u32 x = ((e & 0x0fffffffu) >> (28 - 2 * ext)) ^ qn;
                int matched = x ? ((__builtin_clz(x) - (32 - 2 * ext)) >> 1) : ext;
                const int nv = (int)(e >> 28);
                matched = matched < nv ? matched : nv;
                if (NQ) { const u32 nb = lean_nbits(nmV, p + g * st + k); const int nc = nb ? __builtin_ctz(nb) : 32; matched = matched < nc ? matched : nc; }   // (an MMP ends at an N)
                const int remg = L - (p + g * st + k) < ext ? L - (p + g * st + k) : ext;      // the characters this hit's extension may use (the tail: what is left of the read)
                fullb[l] = act && matched >= remg;
                mv[l] = act ? matched : -1;
              }
              const u64 bqb = ballot(fullb);
              // lane j < J: the run of hit j's suffixes whose extension matched all it may use
              LV<u32> fV, cV; LV<bool> goodb;
              QM_LANES(l) {
                const int jh = l & 7;
                const u32 m = (u32)(bqb >> ((jh << lsh) & 63)) & ((1u << (1 << lsh)) - 1u);
                const u32 f = m ? (u32)__builtin_ctz(m) : 0u, c = m ? 32u - (u32)__builtin_clz(m) - f : 0u;
                fV[l] = f; cV[l] = c; goodb[l] = l < J && m != 0 && c < maxIv;
              }
              const u32 goodm = (u32)ballot(goodb) & 0xffu;
              const int Jd = ctz32(~goodm);                             // the leading hits with an interval to record
              // ... and the hit behind them, when its extension stops short on every suffix (the read's next error): its longest match and the
              // suffixes that reach it are in the lanes already -- recorded here as well instead of by a round of the loop below
              int Fml = -1, Ff = 0, Fc = 0;
              if (Jd >= 1 && Jd < J && Jd != T && !((bqb >> ((Jd << lsh) & 63)) & ((1ULL << (1 << lsh)) - 1ULL))) {
                LV<int> gm;
                QM_LANES(l) { gm[l] = (l >> lsh) == Jd ? mv[l] : -1; }
                const int mm = wave_max(gm);
                LV<bool> bb;
                QM_LANES(l) { bb[l] = (l >> lsh) == Jd && mv[l] == mm && mm >= 0; }
                const u32 m = (u32)(ballot(bb) >> ((Jd << lsh) & 63)) & ((1u << (1 << lsh)) - 1u);
                if (m) {
                  const u32 f = (u32)__builtin_ctz(m), c = 32u - (u32)__builtin_clz(m) - f;
                  if (c < maxIv) { Fml = mm; Ff = (int)f; Fc = (int)c; }
                }
              }
              const int Jr = Jd + (Fml >= 0 ? 1 : 0);                    // hits recorded by this round
              if (Jr >= 2) {
                LV<bool> cb;
                QM_LANES(l) {
                  {
                    // (every lane stores: a lane without a record to the second half of the stash's room, which the collector does not use -- a branch
                    // on a per-lane condition costs the scalar unit its exec-mask juggling, and the scalar unit is what this kernel runs out of)
                    const bool shortOne = l == Jd && Fml >= 0;
                    QM_LDS(IntRec)* d = (l < Jr) ? ints + sn + l : ints + QM_LEAN_MAXIV + (l & 31);
                    const u32 f = shortOne ? (u32)Ff : fV[l], c = shortOne ? (u32)Fc : cV[l];
                    d->b = aiV[l] + 1 + f; d->e = aiV[l] + 1 + f + c; d->q = (u32)(p + l * st);
                    d->len = shortOne ? (u32)(k + Fml) : (u32)(l == T ? L - (p + l * st) : mlenC);     // (the tail's MMP ends with the read)
                  }
                  cb[l] = l >= 1 && l < Jr && ((W.Cm >> ((relb0 + (l & 7) * step) & 31)) & 1u) != 0;
                }
                // the spot checks and hit counts of hits 1 .. Jr - 1 (each: the spot check behind the hit before it, then the hit itself)
                ha += 2u * (u32)(Jr - 1);
                hb += 2u * (u32)popc64(ballot(cb));
                QM_CNT(18, Jr);
                const int corr0 = prevEnd > p ? prevEnd - p : 0;
                sn += Jr;
                if (Jd - 1 == T) {
                  // the run's last hit matched to the read's end: hits 0 .. T - 1 as ever, the tail overlaps the one before it like they do, and
                  // the walk is through (`if (p + mlen >= L) break`)
                  cov += mlenC - corr0 + (T - 1) * st + (L - (p + T * st) - (mlenC - st));
                  break;
                }
                cov += mlenC - corr0 + (Jd - 1) * st;                    // (hit j >= 1 overlaps the one before it by mlen - (maxMMPExtension + 1))
                const int pl = p + (Jd - 1) * st;
                prevEnd = pl + mlenC;
                spot = 1; stopAfter = 0; width = 32;
                if (Fml >= 0) {
                  // the short one: len k + Fml at pl + st; it overlaps the hit before it by mlenC - st, and the walk goes on one past its end (kp)
                  cov += Fml + 1;
                  prevEnd = pl + st + k + Fml;
                  lb = read_lane(aiV, Jd) + 1 + (u32)Ff; ub = lb + (u32)Fc;
                  pstride = 1;
                  p = pl + st + Fml + 1;
                } else {
                  lb = read_lane(aiV, Jd - 1) + 1 + read_lane(fV, Jd - 1); ub = lb + read_lane(cV, Jd - 1);
                  pstride = st <= 32 ? st : 1;
                  p = pl + st;                                           // kp of the last one
                }
                if (p + k == L) lastSearch = 1;
                continue;
              }
            }
          }
          const u32 lbIn = lb ? lb - 1 : 0;                           // :553
          const int wiv = (int)(ub - lbIn - 1);
          if (wiv < 1 || wiv > 64) { bail = 1; break; }
          QM_CNT(18, 1);
          const int pos = p + k;
          // -s: only the MMP that starts the read may run to its end; one that then stops beyond k + maxMMPExtension (and short of the
          // read's end) is redone cut, from the same interval
          int capped = (SEL && p != 0) ? 1 : 0;
          int mlen = 0, first = 0, cnt = 0;
          LV<u32> tdv, tpv; LV<int> ncap;
          if (NQ && !WIDE) { QM_LANES(l) { ncap[l] = lean_ndist(nmV, pos); } }   // the extension ends at the strand's next N (extendSearchNaive compares characters)
          while (true) {
            const int cutrem = (L - pos) < ext ? (L - pos) : ext;     // characters a capped extension may use
            const int rem = capped ? cutrem : L - pos;
            LV<int> lc; LV<bool> fullv;
            if (SEL && capped && rem >= 1 && rem <= QM_NEXT_BASES && ix.sanext) {
              // a capped extension: the rem (<= 14) characters behind the k-mer against the narrow table's entry of every suffix
              QM_LANES(l) {
                const u32 qn = (u32)lean_kmer(pk2 + IW * V, pos + imgOff, rem);
                const u32 e = ix.sanext[lbIn + 1 + (u32)(l < wiv ? l : wiv - 1)];
                const u32 x = ((e & 0x0fffffffu) >> (28 - 2 * rem)) ^ qn;
                int matched = x ? ((__builtin_clz(x) - (32 - 2 * rem)) >> 1) : rem;
                const int nv = (int)(e >> 28);
                matched = matched < nv ? matched : nv;
                if (NQ && !WIDE) matched = matched < ncap[l] ? matched : ncap[l];
                lc[l] = l < wiv ? k + matched : -1;
                fullv[l] = false; tdv[l] = 0; tpv[l] = 0;
              }
            } else if (WIDE) {
              // the wide table: 224 characters behind every suffix's k-mer, one 64-byte entry (SaExt2)
              const int cap = rem < QM_EXT2_BASES ? rem : QM_EXT2_BASES;
              QM_LANES(l) {
                const int gq = pos + imgOff, j = gq >> 5, sh = 2 * (gq & 31);
                const QM_LDS(u64)* img = pk2 + IW * V + j;
                const unsigned char* ep = (const unsigned char*)&ix.saext2[lbIn + 1 + (u32)(l < wiv ? l : wiv - 1)];
                U4 e0, e1, e2, e3;
                load_32(ep, e0, e1); load_32(ep + 32, e2, e3);
                const u64 tw[7] = {((u64)e0.y << 32) | e0.x, ((u64)e0.w << 32) | e0.z, ((u64)e1.y << 32) | e1.x, ((u64)e1.w << 32) | e1.z,
                                   ((u64)e2.y << 32) | e2.x, ((u64)e2.w << 32) | e2.z, ((u64)e3.y << 32) | e3.x};
                u64 wprev = img[0];
                int matched = QM_EXT2_BASES; bool open = true;
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                  const u64 wnext = img[t + 1];
                  const u64 q = (wprev << sh) | ((wnext >> 1) >> (63 - sh));
                  const u64 x = tw[t] ^ q;
                  if (open && x) { matched = 32 * t + (clz64(x) >> 1); open = false; }
                  wprev = wnext;
                }
                const int nv = (int)(e3.z >> QM_EXT2_TID_BITS);
                matched = matched < nv ? matched : nv;
                matched = matched < cap ? matched : cap;
                fullv[l] = matched == QM_EXT2_BASES;
                lc[l] = l < wiv ? k + matched : -1;
                tdv[l] = e3.z & ((1u << QM_EXT2_TID_BITS) - 1); tpv[l] = e3.w;
              }
              if (rem > QM_EXT2_BASES) { if (ballot(fullv)) { bail = 1; break; } }   // (a 256-character read matching beyond what the table holds)
            } else {
              const int cap = rem < QM_EXT_BASES ? rem : QM_EXT_BASES;
              QM_LANES(l) {
                // the strand's characters from pos on, packed like the table's entries (the same words in every lane: broadcast reads)
                const int gq = pos + imgOff, j = gq >> 5, sh = 2 * (gq & 31);
                const QM_LDS(u64)* img = pk2 + IW * V + j;
                const u64 w0 = img[0], w1 = img[1], w2 = img[2], w3 = img[3];
                const u64 q0 = (w0 << sh) | ((w1 >> 1) >> (63 - sh)), q1 = (w1 << sh) | ((w2 >> 1) >> (63 - sh)), q2 = (w2 << sh) | ((w3 >> 1) >> (63 - sh));
                U4 a, b;
                load_32(&ix.saext[lbIn + 1 + (u32)(l < wiv ? l : wiv - 1)], a, b);
                const u64 x0 = (((u64)a.y << 32) | a.x) ^ q0, x1 = (((u64)a.w << 32) | a.z) ^ q1, x2 = (((u64)b.y << 32) | b.x) ^ q2;
                const int nv = (int)(b.z >> QM_EXT_TID_BITS);
                // the first word that differs and where (selects, no branches)
                const u64 xs = x0 ? x0 : (x1 ? x1 : x2);
                const int xb = x0 ? 0 : (x1 ? 32 : 64);
                int matched = xs ? xb + (clz64(xs | 1ULL) >> 1) : QM_EXT_BASES;
                matched = matched < nv ? matched : nv;
                matched = matched < cap ? matched : cap;
                if (NQ) matched = matched < ncap[l] ? matched : ncap[l];
                fullv[l] = matched == QM_EXT_BASES;
                lc[l] = l < wiv ? k + matched : -1;
                tdv[l] = b.z & ((1u << QM_EXT_TID_BITS) - 1); tpv[l] = b.w;
              }
              if (rem > QM_EXT_BASES) { if (ballot(fullv)) { bail = 1; break; } }   // (a 128-character read matching beyond what the table holds)
            }
            // the suffixes with the longest match are one run of lanes.  A capped extension usually matches all it may use on some
            // of them (the read keeps matching): that case needs no maximum over the lanes
            u64 bq = 0;
            if (SEL && capped) {
              LV<bool> fullc;
              QM_LANES(l) { fullc[l] = lc[l] == k + rem; }
              bq = ballot(fullc);
              mlen = k + rem;
            }
            if (!bq) {
              mlen = wave_max(lc);
              LV<bool> best;
              QM_LANES(l) { best[l] = lc[l] == mlen; }
              bq = ballot(best);
            }
            first = ctz64(bq); cnt = 64 - clz64(bq) - first;
            if (SEL && !capped && mlen < L && mlen >= k + ext) {          // (p == 0 here: mlen >= L <=> the whole read)
              // redone cut at k + maxMMPExtension from the same interval (:568-575) -- without a second round of loads: the suffixes that match
              // at least that far are known from this round's lengths
              LV<bool> fc;
              QM_LANES(l) { fc[l] = lc[l] >= k + ext; }
              bq = ballot(fc);
              mlen = k + ext; first = ctz64(bq); cnt = 64 - clz64(bq) - first;
              capped = 1;
            }
            break;
          }
          if (bail) break;
          lb = lbIn + 1 + (u32)first; ub = lb + (u32)cnt;
          const int kp = p + mlen - (k - 1);
          int recorded = 0;
          if ((u32)cnt < maxIv) {                                      // ub > lb && ub - lb < maxInterval (:577-618)
            if ((!SEL && sufN + cnt > QM_LEAN_SUF) || sn >= QM_LEAN_MAXIV) { bail = 1; break; }
            if (SEL) {
              QM_LANES(l) { QM_LDS(IntRec)* d = l == 0 ? ints + sn : ints + QM_LEAN_MAXIV + (l & 31); d->b = lb; d->e = ub; d->len = (u32)mlen; d->q = (u32)p; }   // (lanes 1 .. 63: into unused room, no branch)
            } else {
              QM_LANES(l) {
                if (l >= first && l < first + cnt) {
                  QM_LDS(LeanSuf)* d = suf + (sufN + l - first);
                  d->tid = tdv[l]; d->pos = tpv[l]; d->qp = (u32)p; d->iv = (u32)sn;
                }
              }
              if (cnt < minSpan) { minSpan = cnt; minIdx = sn; }
              sufN += cnt;
            }
            sn += 1;
            const int corr = prevEnd > p ? prevEnd - p : 0;
            cov += mlen - corr;
            prevEnd = p + mlen;
            recorded = 1;
          }
          if (p + mlen >= L) break;
          spot = recorded;                                             // (:602-611: only behind a recorded interval; kp < P here)
          stopAfter = lastSearch;
          if (lastSearch && !spot) break;
          // -s: an MMP that was cut at k + maxMMPExtension is followed by the next one maxMMPExtension + 1 positions on, and so on while
          // the read matches: the spot check's probe looks up those positions only
          if (SEL && recorded && mlen == k + ext) { const int st = ext + 1; if ((st & (st - 1)) == 0 && st <= 32) pstride = st; }
          p = kp;                                                      // NIP off: lce == matchedLen (:635-647)
          width = stopAfter ? 1 : 32;
          if (p + k == L) lastSearch = 1;
        }
        // the other strand's turn (:258 checkRC after the read's own pass, :271 checkFwd after the reverse complement's)?
        if (!bail && (useCov ? (hb > 0) : (hb >= ha))) bail = 2;
        if (!bail) {
          // (:283-288: the other strand has no coverage and no intervals -- nothing to clear, with or without the slack of -s)
          if (B.quasi_cov > 0.0 && sn > 0) { const double f = (double)cov / (double)L; if (f < B.quasi_cov) sn = 0; }   // :343-358
#if defined(QM_LEAN_ABLATE) && QM_LEAN_ABLATE == 3
          foundHit = ((sn + cov + minIdx + sufN) & 0x40000000) != 0; sn = 0;
#endif
          if (SEL) { n = sn; selV = V; }
          else if (sn > 0) {
            LeanStrand S; S.n = sn; S.sufN = sufN; S.minIdx = minIdx; S.minSpan = 0; S.cov = 0;
            wave_fence();
            n = lean_h2m(suf, S, V != 0, elem, keep, slot);
          }
        }
      }
    }
    if (bail) { lean_defer(B, read, bail == 2 ? 3 : 2); continue; }
    if (SEL) {
      // ---- the collector's output: the read's SA-interval records (one strand's) through the wave's chunk of B.iv_out (dump_intervals),
      // and what SACollector::operator() returned
      long long base = 0;
      if (n > 0) {
        if (wa.ivBase < 0 || wa.ivUsed + n > QM_IVCHUNK) {
          LV<u64> bv;
          QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor + QM_SC_IVCUR, (u64)QM_IVCHUNK); }
          wa.ivBase = (long long)read_lane(bv, 0); wa.ivUsed = 0;
        }
        base = wa.ivBase + wa.ivUsed; wa.ivUsed += n;
      }
      const bool fits = base + n <= B.iv_cap;
      if (!fits) { QM_LANES(l) { if (l == 0) *B.status |= 16; } }
      const int mate = PAIRED ? ((WIDE || NQ) ? (read & 1) : h) : 0;
      wave_fence();
      QM_LANES(l) {
        if (l == 0) { B.iv_cnt[read] = fits ? (u32)n : 0u; B.iv_off[read] = base; B.found_out[read] = foundHit ? 1 : 0; if (NQ) B.lcnt[read] = 0; }   // (NQ: the first pass's mark goes)
        if (fits && l < n) {
          const QM_LDS(IntRec)* r = (const QM_LDS(IntRec)*)M.ints + l;
          qm_sa_interval_hit hh; hh.begin = (int)r->b; hh.end = (int)r->e; hh.len = r->len; hh.query_pos = r->q;
          hh.query_rc = (uint8_t)selV; hh.list = (uint8_t)(2 * mate + selV); hh.pad = 0;
          B.iv_out[base + l] = hh;
        }
      }
      continue;
    }
    // ---- the list to B.lists through the wave's chunk of the bump allocator (finish_read)
    long long base = 0;
    if (n > 0) {
      if (wa.base < 0 || wa.used + n > QM_LEAN_CHUNK) {
        LV<u64> bv;
        QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor, (u64)QM_LEAN_CHUNK); }
        wa.base = (long long)read_lane(bv, 0); wa.used = 0;
      }
      base = wa.base + wa.used;
      if (base + n > B.lists_cap) { QM_LANES(l) { if (l == 0) *B.status |= 1; } n = 0; base = 0; }
      else wa.used += n;
    }
    if (n > 0) { QM_LANES(l) { if (keep[l]) B.lists[base + slot[l]] = elem[l]; } }
    const u32 flag = (B.fuzzy && foundHit) ? 0x80000000u : 0u;
    QM_LANES(l) { if (l == 0) { B.lcnt[read] = (u32)n | flag; B.loff[read] = base; if (B.found_out) B.found_out[read] = foundHit ? 1 : 0; } }   // (found_out: stage views without interval records)
  }
}

}  // namespace qm
