// qm_duo.inl -- stage A AND stage B of a read pair in the wavefront that holds it (round 6).
//
// qm_lean_kernel (qm_lean.inl) put the two mates of a pair into one wavefront but walked them one after the other, every step of
// either walk decided on the scalar unit: 1 125 scalar-side instructions per pair against 756 vector ones, the CU's one scalar
// unit 90 % busy and its four vector units 40 % idle (profiles/r05/kernel_stats_and_pmc_dense_r05zf.txt).  Here the two walks run
// in LOCKSTEP: lanes 0-31 own the first mate, lanes 32-63 the second, and everything a walk knows -- position, window, masks,
// interval, counts, flags -- is a per-lane value that is the same in the 32 lanes of a half.  One pass of the loop is "probe a
// window where a half needs one; step both walks to their next hit; extend both hits", so an instruction serves two reads whether
// it is a vector or a scalar one, the scalar unit only decides whether ANY half still needs a phase, and the number of passes is
// the longer of the two walks instead of their sum.
//
//   * a window probe is 32 positions of ONE read in 32 lanes: the canonical table (qm_mapper.inl, Bucket) answers a position's
//     k-mer and its reverse complement from the same 64-byte bucket, so one lane loads the bucket's keys and both interval
//     pairs (48 bytes) and knows both strands' answers (qm_lean_kernel gave a position two lanes);
//   * cross-lane traffic stays inside a half: ballots are split by half, "the interval of the lane that owns position p" is a
//     ds_bpermute (the index differs between the halves), the longest match of an extension a DPP maximum over two rows;
//   * hits -> mappings runs for both mates at once over the two halves' suffix stashes (<= 32 suffixes each);
//   * and with both mates' lists in registers the PAIR is finished on the spot (mergeLeftRightHits, RapMapUtils.hpp:1185-1264, and
//     the per-pair driver, RapMapSAMapper.cpp:527-551,684-701): the hit count goes to pair_cnt[u], the merged records -- two list
//     elements per hit -- through the bump allocator, the HitCounters into wave-level sums.  Stage B's count pass has nothing left
//     to do for such a pair and its write pass only expands records (unit_merge, QM_LCNT_PAIR).
//
// Same answers as lean_iter / map_read read by read (tests/emu maps every paired batch with all three).  A read this kernel does
// not take is marked exactly like qm_lean_kernel's (QM_LCNT_LEAN, scalar slot QM_SC_LEANQ) and mapped by the general kernel; its
// mate's list is then written per read, and stage B merges that pair the old way.  Besides qm_lean_kernel's reasons a read is
// left when an interval is wider than 32 suffixes or its intervals hold more than 32 together (a half has 32 lanes).
#pragma once
#include "qm_lean.inl"

namespace qm {

#ifndef QM_DUO_SUF
#define QM_DUO_SUF 32            // suffixes a read's intervals may hold together, and the widest interval
#endif
#define QM_DUO_MAXIV 32          // intervals per read (a bit each)

struct DuoMem {                                // one wave's LDS slab (4 656 bytes)
  u64 pk[2][2][8];                             // as LeanMem::pk: [mate][0: the read, 1: mirrored reverse complement][word]; words 4-7 stay zero
  union {
    LeanSuf suf[2][QM_DUO_SUF];                // [mate]: suffixes of the intervals recorded for it
    u64 lst[2][32];                            // ... later its hit list, sorted, for the merge
  };
  LeanSuf trash[64];                           // where a lane that has nothing to record stores (no branch around the store)
  U4 pf[3][34];                                // the next pair's first probe: the 48 bytes of the buckets of its mates' first and last k-mers (lanes 0, 1, 32, 33),
                                               // fetched straight into here (global_load_lds_dwordx4) while the pair before it is finished
  u32 stage[100];                              // raw characters of the next pair: dwords [0, 32) mate 0, [32, 64) mate 1, [64] / [96] the 33rd dword of mate 0 / 1
  u32 ostage[2][8];                            // offsets of the next / the next but one pair: off1[u], off1[u + 1], off2[u], off2[u + 1]
};
typedef PairCtr DuoCtr;

// khash.find for one POSITION per lane: w the k-mer as the read has it, wr its reverse complement.  One bucket of the canonical
// table holds both; fh / (flb, fub): the k-mer is in the index / its interval, ch / (clb, cub): the same for the reverse complement.
// Lanes that are not `on` read bucket 0 and come back with nothing.
// duo_find_rest: what follows the first bucket's 48 bytes (a: the two keys, f / r: the interval pairs of the canonical k-mers and of
// their reverse complements) -- whoever loaded them: duo_find itself, or the prefetch of a pair's first probe (duo_prepare)
QM_DEV void duo_find_rest(const DevIndex& ix, const LV<u64>& ckv, const LV<u32>& bigv, const LV<u32>& on, LV<U4>& av, LV<U4>& fv, LV<U4>& rv, LV<u64>& bkt,
                          LV<u32>& fh, LV<u32>& ch, LV<u32>& flb, LV<u32>& fub, LV<u32>& clb, LV<u32>& cub) {
  LV<u32> more;
  QM_LANES(l) {
    const U4 a = av[l], f = fv[l], r = rv[l];
    const u64 ck = ckv[l];
    const bool big = bigv[l] != 0;
    const u64 k0r = ((u64)a.y << 32) | a.x, k1 = ((u64)a.w << 32) | a.z;
    const bool h0 = (k0r & ~QM_BK_OVF) == ck, h1 = k1 == ck;
    const u32 cfl = h0 ? f.x : f.z, cfu = h0 ? f.y : f.w, crl = h0 ? r.x : r.z, cru = h0 ? r.y : r.w;   // the canonical k-mer's interval, its reverse complement's
    const bool m = on[l] != 0 && (h0 || h1);
    const u32 al = big ? crl : cfl, bl = big ? cfl : crl;
    flb[l] = al; fub[l] = big ? cru : cfu; clb[l] = bl; cub[l] = big ? cfu : cru;
    fh[l] = (m && al != QM_IV_NONE) ? 1u : 0u; ch[l] = (m && bl != QM_IV_NONE) ? 1u : 0u;
    more[l] = (on[l] != 0 && !(h0 || h1) && k0r != ~0ULL && (k0r & QM_BK_OVF) != 0) ? 1u : 0u;
  }
  // 0.4 % of the buckets: a key that hashes here lives in a later bucket (lean_find)
  while (true) {
    LV<bool> mb; QM_LANES(l) { mb[l] = more[l] != 0; }
    if (!ballot(mb)) break;
    QM_LANES(l) {
      const u64 b = more[l] ? ((bkt[l] + 1) & ix.hmask) : 0ULL;
      U4 a, f, r;
      load_48(&ix.slots[b], a, f, r);
      QM_CNT(1, more[l] ? 1 : 0);
      const u64 k0r = ((u64)a.y << 32) | a.x, k1 = ((u64)a.w << 32) | a.z;
      const bool h0 = (k0r & ~QM_BK_OVF) == ckv[l], h1 = k1 == ckv[l];
      if (more[l] && (h0 || h1)) {
        const u32 cfl = h0 ? f.x : f.z, cfu = h0 ? f.y : f.w, crl = h0 ? r.x : r.z, cru = h0 ? r.y : r.w;
        const bool big = bigv[l] != 0;
        const u32 al = big ? crl : cfl, bl = big ? cfl : crl;
        flb[l] = al; fub[l] = big ? cru : cfu; clb[l] = bl; cub[l] = big ? cfu : cru;
        fh[l] = al != QM_IV_NONE ? 1u : 0u; ch[l] = bl != QM_IV_NONE ? 1u : 0u;
      }
      more[l] = (more[l] && !(h0 || h1) && k0r != ~0ULL && (k0r & QM_BK_OVF) != 0) ? 1u : 0u;
      bkt[l] = b;
    }
  }
}
QM_DEV void duo_find(const DevIndex& ix, const LV<u64>& w, const LV<u64>& wr, const LV<u32>& on, LV<u32>& fh, LV<u32>& ch,
                     LV<u32>& flb, LV<u32>& fub, LV<u32>& clb, LV<u32>& cub) {
  LV<u64> bkt, ckv; LV<u32> bigv; LV<U4> av, fv, rv;
  QM_LANES(l) {
    const bool big = wr[l] < w[l];
    const u64 ck = big ? wr[l] : w[l];
    const u64 b = on[l] ? ((u64)bucket_hash(ck) & ix.hmask) : 0ULL;
    U4 a, f, r;
    load_48(&ix.slots[b], a, f, r);
    QM_CNT(1, on[l] ? 1 : 0);
    av[l] = a; fv[l] = f; rv[l] = r; bkt[l] = b; ckv[l] = ck; bigv[l] = big ? 1u : 0u;
  }
  duo_find_rest(ix, ckv, bigv, on, av, fv, rv, bkt, fh, ch, flb, fub, clb, cub);
}

// the same through the compact -p image (lean_find_ph): the structure is keyed by the k-mer itself, so a position asks twice -- but the
// pre-filter's word is chosen by the canonical k-mer, one sector for both questions, and only orientations that pass it walk the levels
QM_DEV void duo_find_ph(const DevIndex& ix, const LV<u64>& w, const LV<u64>& wr, const LV<u32>& on, LV<u32>& fh, LV<u32>& ch,
                        LV<u32>& flb, LV<u32>& fub, LV<u32>& clb, LV<u32>& cub) {
  const PhIndex& P = ix.phv;
  LV<bool> wf, wc;
  QM_LANES(l) {
    bool a = on[l] != 0, b = on[l] != 0;
    if (P.filter) {
      u64 wd, bf, wd2, bc;
      ph_filter_slot(w[l], wr[l], P.filterMask, wd, bf);
      ph_filter_slot(wr[l], w[l], P.filterMask, wd2, bc);          // (the same word: the canonical k-mer chooses it)
      const u64 x = P.filter[on[l] ? wd : 0ULL];
      QM_CNT(1, on[l] ? 1 : 0);
      a = a && (x & bf) == bf; b = b && (x & bc) == bc;
    }
    wf[l] = a; wc[l] = b;
  }
  // nearly every position that passes the filter does so in ONE orientation: one walk of the levels for the wave, the k-mer where it
  // passed and else the reverse complement; a second walk only when some position passed in both
  LV<bool> both;
  QM_LANES(l) {
    const bool first = wf[l] || wc[l];
    bool h = false; u32 a = 0, b = 0;
    if (first) h = find_kmer<QM_F_PH>(ix, wf[l] ? w[l] : wr[l], a, b);
    const bool isF = wf[l];
    fh[l] = (h && isF) ? 1u : 0u; flb[l] = isF ? a : 0u; fub[l] = isF ? b : 0u;
    ch[l] = (h && !isF) ? 1u : 0u; clb[l] = isF ? 0u : a; cub[l] = isF ? 0u : b;
    both[l] = wf[l] && wc[l];
  }
  if (ballot(both)) {
    QM_LANES(l) {
      bool h = false; u32 a = 0, b = 0;
      if (both[l]) h = find_kmer<QM_F_PH>(ix, wr[l], a, b);
      if (both[l]) { ch[l] = h ? 1u : 0u; clb[l] = a; cub[l] = b; }
    }
  }
}

// What a half's walk carries: every member is one value per lane, the same in the 32 lanes of a half, packed so that the whole state
// is a dozen registers (the kernel runs at 8 waves per SIMD: 64 registers per lane for everything).
//   fl: bits 0-1 mode (0: nothing (more) to do, 1: first-hit scan, SACollector.hpp:167-237, 2: getSAHits_ over strand V), then the flags below
#define QM_DW_MODE 3u
#define QM_DW_V 4u            // the strand walked: reverseRead(read)
#define QM_DW_SKIP 8u         // the walk stands on a hit (position p of its window)
#define QM_DW_SPOT 16u        // the next k-mer is a spot check behind a recorded interval (:602-611)
#define QM_DW_STOP 32u        // ... after which the walk ends (stopAfter)
#define QM_DW_LAST 64u        // lastSearch
#define QM_DW_BAIL 128u       // left to the general kernel
#define QM_DW_FOUND 256u      // foundHit
#define QM_DW_DEF 512u        // not taken at all (a character that is not A C G T, a homopolymer window, more than 128 characters)
#define QM_DW_FL 1024u        // what the first probe learned about the read's last k-mer: it is in the index ...
#define QM_DW_CL 2048u        // ... its reverse complement is (= the first k-mer of reverseRead(read))
#define QM_DW_DEFH 4096u      // (with QM_DW_DEF: because of a window of k equal bases)
struct DuoWalk {
  LV<u32> fl;
  LV<int> p;                                   // the position the walk stands on
  LV<u32> wbw; LV<u32> Fm, Cm;                 // the window: wb | ww << 8 -- positions [wb, wb + ww) of the strand, bit j = k-mer / reverse complement of position wb + j found
  LV<u32> lbw, ubw;                            // ... lane j of the half: the interval of the k-mer at position wb + j
  LV<u32> rlb, rub;                            // the interval of reverseRead(read)'s first k-mer (the first probe looked it up: the reverse complement of the read's last k-mer)
  LV<u32> hab;                                 // ha | hb << 16: hits of the walked strand / of the other one (SACollector.hpp:258,271)
  LV<u32> cntr;                                // sn | sufN << 8: intervals recorded, their suffixes
  LV<u32> mins;                                // minIdx | minSpan << 8: the first smallest interval (HitManager.cpp:636-641)
  LV<int> cov, prevEnd;                        // COV only (quasiCoverage > 0): SACollector.hpp:343-358
};

// Probe positions [p, p + width) of its strand for every half that asked (`need`): lane j of the half looks up position p + j -- the
// k-mer out of the strand's image, its reverse complement out of the other one (lean_probe) -- and the half's window is replaced.
// width: 1 for the walk's last look (stopAfter), else 32.
template <bool PH>
QM_DEV void duo_probe(const DevIndex& ix, const QM_LDS(u64)* pkw, int k, const LV<int>& Lv, const LV<u32>& need, DuoWalk& W) {
  LV<u64> w, wr; LV<u32> on, fh, ch, flb, fub, clb, cub;
  LV<int> wwn;
  QM_LANES(l) {
    const int h = l >> 5, j = l & 31;
    const int P = Lv[l] - k + 1, D = QM_LEAN_MAXLEN - Lv[l];
    const int V = ((W.fl[l] & QM_DW_MODE) == 2u && (W.fl[l] & QM_DW_V)) ? 1 : 0;
    int nw = (W.fl[l] & QM_DW_STOP) ? 1 : 32;
    nw = W.p[l] + nw > P ? P - W.p[l] : nw;
    const bool in = need[l] != 0 && j < nw;
    const int q = in ? W.p[l] + j : 0;
    const QM_LDS(u64)* pkh = pkw + 16 * h;
    w[l] = lean_kmer(pkh + 8 * V, q + (V ? D : 0), k);
    wr[l] = lean_kmer(pkh + 8 * (1 - V), (in ? P - 1 - q : 0) + (V ? 0 : D), k);
    on[l] = in ? 1u : 0u; wwn[l] = nw;
  }
  QM_CNT(3, 1);
  if (PH) duo_find_ph(ix, w, wr, on, fh, ch, flb, fub, clb, cub);
  else duo_find(ix, w, wr, on, fh, ch, flb, fub, clb, cub);
  LV<bool> fb, cb; LV<u32> fm, cm;
  QM_LANES(l) { fb[l] = fh[l] != 0; cb[l] = ch[l] != 0; }
  half_ballot(fb, fm); half_ballot(cb, cm);
  QM_LANES(l) {
    const bool nd = need[l] != 0;
    W.wbw[l] = nd ? ((u32)W.p[l] | ((u32)wwn[l] << 8)) : W.wbw[l];
    W.Fm[l] = nd ? fm[l] : W.Fm[l]; W.Cm[l] = nd ? cm[l] : W.Cm[l];
    W.lbw[l] = nd ? flb[l] : W.lbw[l]; W.ubw[l] = nd ? fub[l] : W.ubw[l];
  }
}

// offsets of pair `it` into ostage[par] (lean_stage_offsets<true>)
QM_DEV void duo_stage_offsets(const ReadBatch& B, int it, int nit, DuoMem& M, int par) {
  if (it >= nit) return;
  QM_LANES(l) {
    if (l < 8) { const long long* o = (l < 4 ? B.off1 : B.off2) + it; lds_dma_u32((const u32*)o + (l & 3), M.ostage[par], l); }
  }
}
// the offsets in ostage[par] (landed) into the request for the two mates' characters: every lane works out its own mate's address
QM_DEV void duo_stage_chars(const ReadBatch& B, int it, int nit, DuoMem& M, int par) {
  if (it >= nit) return;
  QM_LANES(l) {
    const int h = l >> 5, j = l & 31;
    const QM_LDS(u32)* os = (const QM_LDS(u32)*)&M.ostage[par][4 * h];
    const u32 olo = os[0], ohi = os[1], o1 = os[2];
    int len = (int)(o1 - olo);
    if (len > QM_LEAN_MAXLEN) len = QM_LEAN_MAXLEN;
    const unsigned char* p = (h ? B.seq2 : B.seq1) + (long long)(((u64)ohi << 32) | (u64)olo);
    const int mis = (int)((unsigned long long)p & 3ULL);
    const u32* g = (const u32*)(p - mis);
    const int nd = (mis + len + 3) >> 2;                   // <= 33
    if (j < nd) lds_dma_u32(g + j, &M.stage[0], l);        // dword 32 h + j
    if (j == 0 && nd > 32) lds_dma_u32(g + 32, &M.stage[64], l);   // dword 64 + 32 h
  }
}

// What duo_prepare leaves in registers for the iteration that maps the pair
struct DuoNext { LV<int> Lv, defv; LV<u64> ck; LV<u32> flg; };    // flg: bit 0 the lane asked (first / last k-mer of its mate), bit 1 its k-mer is the larger of the two orientations

// Pair `it`, first half: the two mates' characters (staged by the iteration before) -> 2-bit images of both strands, four characters
// per lane (lean_iter); the staging of the pairs behind it; and the FIRST PROBE's requests -- lane 0 of a half = its mate's position 0,
// lane 1 = position P - 1 (SACollector.hpp:167-237 starts at position 0; the read's last k-mer is the first thing the
// reverse-complement pass asks for), both orientations each: one canonical bucket per lane, its 48 bytes sent straight to LDS.
// Runs BEFORE the pair ahead of it is finished (hits -> mappings, merge, write-out), so that trip is over when duo_iter starts.
template <bool PH>
QM_DEV void duo_prepare(const DevIndex& ix, const ReadBatch& B, int it, int nit, int nw, int par, DuoMem& M, DuoNext& N) {
  if (it >= nit) return;
  const int k = ix.k;
  const QM_LDS(u64)* pkw = (const QM_LDS(u64)*)&M.pk[0][0][0];
  LV<int> rawv;
  LV<bool> bad, rep;
  {
    QM_LDS(unsigned char)* PKb = (QM_LDS(unsigned char)*)&M.pk[0][0][0];
    QM_LANES(l) {
      const int h = l >> 5, jj = l & 31, base = 4 * jj;
      const QM_LDS(u32)* os = (const QM_LDS(u32)*)&M.ostage[par][4 * h];
      const u32 a0 = os[0], a1 = os[2];
      const int raw = (int)(a1 - a0), len = raw > QM_LEAN_MAXLEN ? QM_LEAN_MAXLEN : raw;
      const int mis = (int)(((u32)(unsigned long long)(h ? B.seq2 : B.seq1) + a0) & 3u);
      const QM_LDS(u32)* srow = (const QM_LDS(u32)*)&M.stage[32 * h];
      const u32 w0 = srow[jj], w1 = jj == 31 ? M.stage[64 + 32 * h] : srow[jj + 1];
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
      u32 r = brev32(pk) >> 24;                                                  // the same four bases reverse-complemented (Kmer.hpp:92-100 on a byte)
      r = (~(((r >> 1) & 0x55u) | ((r & 0x55u) << 1))) & 0xffu;
      const int img = 128 * h;                                                   // bytes: image (h, strand) starts at 128 h + 64 strand
      PKb[img + 8 * (jj >> 3) + 7 - (jj & 7)] = (unsigned char)pk;
      const int mj = 31 - jj;
      PKb[img + 64 + 8 * (mj >> 3) + 7 - (mj & 7)] = (unsigned char)r;
      rawv[l] = raw; N.Lv[l] = len;
    }
  }
  LV<u32> dirty, reps;
  half_ballot(bad, dirty); half_ballot(rep, reps);
  wave_fence();
  // the staging rows are free again: the next pair's characters, and the offsets of the one after it
  duo_stage_chars(B, it + nw, nit, M, par ^ 1);
  duo_stage_offsets(B, it + 2 * nw, nit, M, par);
  // what this kernel takes: no character but A C G T, no window of k equal bases, at most 128 characters (lean_iter)
  // 1: a character that is not A C G T or too many characters, 2: a window of k equal bases
  QM_LANES(l) { N.defv[l] = (rawv[l] > QM_LEAN_MAXLEN || dirty[l] != 0) ? 1 : (4 * popc32(reps[l]) + 6 >= k ? 2 : 0); }
  QM_LANES(l) {
    const int h = l >> 5, jj = l & 31;
    const int P = N.Lv[l] - k + 1, D = QM_LEAN_MAXLEN - N.Lv[l];
    const bool o = !N.defv[l] && P >= 1 && jj < 2 && (jj == 0 || P > 1);
    const int q = (o && jj == 1) ? P - 1 : 0;
    const QM_LDS(u64)* pkh = pkw + 16 * h;
    const u64 w = lean_kmer(pkh, q, k), wr = lean_kmer(pkh + 8, (o ? P - 1 - q : 0) + D, k);
    const bool big = wr < w;
    N.ck[l] = PH ? w : (big ? wr : w);
    N.flg[l] = (o ? 1u : 0u) | (big ? 2u : 0u);
    if (!PH && o) {
      const unsigned char* bp = (const unsigned char*)&ix.slots[(u64)bucket_hash(N.ck[l]) & ix.hmask];
      QM_CNT(1, 1);
      lds_dma_u128(bp, &M.pf[0][0], l); lds_dma_u128(bp + 16, &M.pf[1][0], l); lds_dma_u128(bp + 32, &M.pf[2][0], l);
    }
  }
}

// Pair `it` (reads 2 it and 2 it + 1), second half: the first probe's answers, the two walks in lockstep, hits -> mappings, the merge.
// Between the walks and the rest it runs duo_prepare for the pair the wave maps next.
template <bool PH, bool COV>
QM_DEV void duo_iter(const DevIndex& ix, const ReadBatch& B, int it, int nit, int nw, int par, DuoMem& M, WaveAlloc& wa, DuoCtr& ctr, DuoNext& N) {
  const int k = ix.k;
  const QM_LDS(u64)* pkw = (const QM_LDS(u64)*)&M.pk[0][0][0];
  LV<int> Lv, Pv, defv;
  QM_LANES(l) { Lv[l] = N.Lv[l]; Pv[l] = N.Lv[l] - k + 1; defv[l] = N.defv[l]; }
  DuoWalk W;
  const u32 maxIv = (u32)B.max_interval;
  {
    LV<u32> on, fh, ch, flb, fub, clb, cub;
    QM_LANES(l) { on[l] = N.flg[l] & 1u; }
    QM_CNT(3, 1);
    lds_dma_wait();                                        // the first probe's buckets (and everything else that was requested) have landed
    if (PH) {
      LV<u64> w, wr;
      QM_LANES(l) {
        const int h = l >> 5, jj = l & 31;
        const int P = Pv[l], D = QM_LEAN_MAXLEN - Lv[l];
        const int q = (on[l] && jj == 1) ? P - 1 : 0;
        const QM_LDS(u64)* pkh = pkw + 16 * h;
        w[l] = lean_kmer(pkh, q, k);
        wr[l] = lean_kmer(pkh + 8, (on[l] ? P - 1 - q : 0) + D, k);
      }
      duo_find_ph(ix, w, wr, on, fh, ch, flb, fub, clb, cub);
    } else {
      LV<U4> av, fv, rv; LV<u64> bkt; LV<u32> bigv;
      QM_LANES(l) {
        const int sl = (l & 31) < 2 ? l : 0;                // (only lanes 0, 1, 32, 33 asked: the others read lane 0's slot and ignore it)
        av[l] = M.pf[0][sl]; fv[l] = M.pf[1][sl]; rv[l] = M.pf[2][sl];
        bkt[l] = (u64)bucket_hash(N.ck[l]) & ix.hmask; bigv[l] = (N.flg[l] >> 1) & 1u;
      }
      duo_find_rest(ix, N.ck, bigv, on, av, fv, rv, bkt, fh, ch, flb, fub, clb, cub);
    }
    LV<bool> fb, cb; LV<u32> fm, cm, s0lb, s0ub, rlb, rub; LV<int> i0, i1;
    QM_LANES(l) { fb[l] = fh[l] != 0; cb[l] = ch[l] != 0; i0[l] = 0; i1[l] = Pv[l] > 1 ? 1 : 0; }
    half_ballot(fb, fm); half_ballot(cb, cm);
    half_read(flb, i0, s0lb); half_read(fub, i0, s0ub);    // the interval of the read's first k-mer ...
    half_read(clb, i1, rlb); half_read(cub, i1, rub);      // ... and of the reverse complement of its last one = the first k-mer of reverseRead(read)
    // ---- where the walks start: the read itself from its first hit when that hit is a forward one (SACollector.hpp:247-254), else
    // reverseRead(read) from 0 (:258-265) -- lean_iter; a read whose first k-mer is in the index in neither orientation scans on (mode 1)
    QM_LANES(l) {
      const bool ok = !defv[l] && Pv[l] >= 1;
      const u32 F0 = fm[l] & 1u, C0 = cm[l] & 1u;
      const u32 Fl = Pv[l] > 1 ? (fm[l] >> 1) & 1u : F0, Cl = Pv[l] > 1 ? (cm[l] >> 1) & 1u : C0;
      const bool fwd = ok && F0 != 0, rev = ok && F0 == 0 && C0 != 0, scan = ok && F0 == 0 && C0 == 0;
      W.fl[l] = (defv[l] ? QM_DW_DEF : 0u) | (defv[l] == 2 ? QM_DW_DEFH : 0u) | (fwd ? (2u | QM_DW_FOUND | QM_DW_SKIP) : 0u) | (rev ? (2u | QM_DW_FOUND | QM_DW_V) : 0u) | (scan ? 1u : 0u) | (Fl ? QM_DW_FL : 0u) | (Cl ? QM_DW_CL : 0u);
      W.p[l] = scan ? 1 : 0;
      W.wbw[l] = 0u | (1u << 8);
      W.Fm[l] = rev ? Cl : F0; W.Cm[l] = rev ? Fl : C0;    // (reverse: what the first probe learned about the read's last k-mer)
      W.lbw[l] = rev ? rlb[l] : s0lb[l]; W.ubw[l] = rev ? rub[l] : s0ub[l];
      W.rlb[l] = rlb[l]; W.rub[l] = rub[l];
      W.hab[l] = (fwd || rev) ? (1u | (fwd ? C0 << 16 : 0u)) : 0u;
      W.cntr[l] = 0; W.mins[l] = 0x7fffff00u; W.cov[l] = 0; W.prevEnd[l] = 0;
    }
  }
  // ---- the two walks in lockstep.  Straight-line selects, not branches, inside a phase: a branch on a per-lane condition costs the
  // scalar unit three instructions (exec mask saved, tested, restored), and the scalar unit is what stage A runs out of first
  while (true) {
    // A: a walk that stands behind its read's last k-mer is through; a window is probed where a walk stands outside its own
    LV<u32> need; LV<bool> live, nb;
    QM_LANES(l) {
      const u32 f = W.fl[l];
      const bool act = (f & QM_DW_MODE) != 0 && !(f & QM_DW_SKIP);
      const bool end = act && W.p[l] >= Lv[l] - k + 1;                         // (scan: no hit anywhere; walk: `if (p >= P) break`)
      const u32 f2 = end ? (f & ~QM_DW_MODE) : f;
      const bool inw = (u32)(W.p[l] - (int)(W.wbw[l] & 0xffu)) < (W.wbw[l] >> 8);
      const bool nd = act && !end && !inw;
      W.fl[l] = f2; need[l] = nd ? 1u : 0u; nb[l] = nd; live[l] = (f2 & QM_DW_MODE) != 0;
    }
    if (!ballot(live)) break;
    if (ballot(nb)) duo_probe<PH>(ix, pkw, k, Lv, need, W);
    // B1: the first-hit scan (SACollector.hpp:167-237): the first position whose k-mer or reverse complement is in the hash
    {
      LV<bool> sc; QM_LANES(l) { sc[l] = (W.fl[l] & QM_DW_MODE) == 1u; }
      if (ballot(sc)) {
        LV<int> rel0;
        QM_LANES(l) {
          const int wb = (int)(W.wbw[l] & 0xffu), ww = (int)(W.wbw[l] >> 8);
          const int rel = (W.p[l] - wb) & 31;
          const u32 mm = (W.Fm[l] | W.Cm[l]) >> rel;
          const bool found = sc[l] && mm != 0;
          const int adv = mm ? ctz32(mm) : ww - rel;                          // to the hit, or behind the window
          W.p[l] = sc[l] ? W.p[l] + adv : W.p[l];
          rel0[l] = found ? rel + adv : 0;
          sc[l] = found;
        }
        QM_LANES(l) {
          const bool fwd = sc[l] && ((W.Fm[l] >> rel0[l]) & 1u) != 0, rev = sc[l] && !fwd;
          // forward: the walk stands on the hit.  Reverse: reverseRead(read) from 0, whose first k-mer the first probe looked up: what it
          // learned about the read's last k-mer (SACollector.hpp:258-265)
          W.hab[l] = sc[l] ? (1u | (fwd ? ((W.Cm[l] >> rel0[l]) & 1u) << 16 : 0u)) : W.hab[l];
          W.fl[l] = sc[l] ? ((W.fl[l] & ~QM_DW_MODE) | 2u | QM_DW_FOUND | (fwd ? QM_DW_SKIP : QM_DW_V)) : W.fl[l];
          W.p[l] = rev ? 0 : W.p[l];
          W.wbw[l] = rev ? (0u | (1u << 8)) : W.wbw[l];
          W.Fm[l] = rev ? ((W.fl[l] & QM_DW_CL) ? 1u : 0u) : W.Fm[l]; W.Cm[l] = rev ? ((W.fl[l] & QM_DW_FL) ? 1u : 0u) : W.Cm[l];
          W.lbw[l] = rev ? W.rlb[l] : W.lbw[l]; W.ubw[l] = rev ? W.rub[l] : W.ubw[l];
        }
      }
    }
    // B2: SACollector::getSAHits_ (SACollector.hpp:441-677, NIP disabled) steps to the walk's next hit inside the window
    LV<u32> ext; LV<int> relh; LV<bool> eb;
    QM_LANES(l) {
      const u32 f = W.fl[l];
      const int wb = (int)(W.wbw[l] & 0xffu), ww = (int)(W.wbw[l] >> 8);
      const int relr = W.p[l] - wb;
      const bool st = (f & (QM_DW_MODE | QM_DW_SKIP)) == 2u && (u32)relr < (u32)ww;   // walking, its window covers p
      const int rel = relr & 31;
      const u32 fm = W.Fm[l] >> rel, cm = W.Cm[l] >> rel;                      // (no bits beyond the window)
      const int avail = ww - rel;
      const bool sp = st && (f & QM_DW_SPOT) != 0;                             // the k-mer the walk goes on with, spot-checked (:602-611)
      const bool stop = sp && (f & QM_DW_STOP) != 0;
      const bool go = st && !stop;
      const u32 below = (fm & (0u - fm)) - 1u;                                 // the positions before the first hit (all of them without one)
      const bool hit = go && fm != 0;
      const int ph = ctz32(fm | 0x80000000u);
      u32 add = sp ? ((fm & 1u) | ((cm & 1u) << 16)) : 0u;
      add += go ? ((u32)popc32(cm & ~fm & below) << 16) : 0u;                  // misses: spotCheck_ of the complement (:667-675)
      add += hit ? (1u | (((cm >> ph) & 1u) << 16)) : 0u;                      // spotCheck_ on the hit (:545)
      W.hab[l] += add;
      W.p[l] += go ? (hit ? ph : avail) : 0;
      u32 f2 = st ? (f & ~QM_DW_SPOT) : f;
      f2 = stop ? (f2 & ~QM_DW_MODE) : f2;
      W.fl[l] = f2;
      const bool e = hit || (f & (QM_DW_MODE | QM_DW_SKIP)) == (2u | QM_DW_SKIP);
      ext[l] = e ? 1u : 0u; eb[l] = e; relh[l] = (W.p[l] - wb) & 31;
    }
    if (!ballot(eb)) continue;
    LV<u32> lbv, ubv;                                                           // the interval of the hit: the lane of the window that owns position p has it
    half_read(W.lbw, relh, lbv); half_read(W.ubw, relh, ubv);
    // C: the MMP extension (SASearcher.hpp:88-309) in the closed form of extend_search_wide against the packed characters behind
    // every suffix's k-mer (one lane per suffix, one 32-byte load each, lean_iter); the (transcript, position) words of the block it
    // settles on go to the half's stash
    LV<int> lc; LV<u32> tdv, tpv, okw; LV<bool> fullv;
    QM_LANES(l) {
      const int h = l >> 5, j = l & 31;
      const u32 lbIn = lbv[l] ? lbv[l] - 1 : 0;                                 // :553
      const int wiv = (int)(ubv[l] - lbIn - 1);
      const bool e = ext[l] != 0;
      const bool okv = e && wiv >= 1 && wiv <= QM_DUO_SUF;
      const int L = Lv[l], V = (W.fl[l] & QM_DW_V) ? 1 : 0;
      const int pos = W.p[l] + k, rem = L - pos;
      const int cap = rem < QM_EXT_BASES ? rem : QM_EXT_BASES;
      const int gq = (okv ? pos : 0) + (V ? QM_LEAN_MAXLEN - L : 0), jw = gq >> 5, sh = 2 * (gq & 31);
      const QM_LDS(u64)* img = pkw + 16 * h + 8 * V + jw;
      const u64 w0 = img[0], w1 = img[1], w2 = img[2], w3 = img[3];
      const u64 q0 = (w0 << sh) | ((w1 >> 1) >> (63 - sh)), q1 = (w1 << sh) | ((w2 >> 1) >> (63 - sh)), q2 = (w2 << sh) | ((w3 >> 1) >> (63 - sh));
      U4 a, b;
      load_32(&ix.saext[okv ? lbIn + 1 + (u32)(j < wiv ? j : wiv - 1) : 0u], a, b);
      const u64 x0 = (((u64)a.y << 32) | a.x) ^ q0, x1 = (((u64)a.w << 32) | a.z) ^ q1, x2 = (((u64)b.y << 32) | b.x) ^ q2;
      const int nv = (int)(b.z >> QM_EXT_TID_BITS);
      const u64 xs = x0 ? x0 : (x1 ? x1 : x2);
      const int xb = x0 ? 0 : (x1 ? 32 : 64);
      int matched = xs ? xb + (clz64(xs | 1ULL) >> 1) : QM_EXT_BASES;
      matched = matched < nv ? matched : nv;
      matched = matched < cap ? matched : cap;
      fullv[l] = okv && rem > QM_EXT_BASES && matched == QM_EXT_BASES;         // (a 128-character read matching beyond what the table holds)
      lc[l] = (okv && j < wiv) ? k + matched : -1;
      tdv[l] = b.z & ((1u << QM_EXT_TID_BITS) - 1); tpv[l] = b.w;
      okw[l] = okv ? 1u : 0u;
      W.fl[l] = (e && !okv) ? ((W.fl[l] & ~(QM_DW_MODE | QM_DW_SKIP)) | QM_DW_BAIL) : W.fl[l];   // an interval wider than a half
    }
    LV<u32> fullm; half_ballot(fullv, fullm);
    LV<int> mx; QM_LANES(l) { mx[l] = lc[l]; }
    half_max(mx);
    LV<bool> best; QM_LANES(l) { best[l] = okw[l] != 0 && lc[l] == mx[l]; }
    LV<u32> bq; half_ballot(best, bq);
    QM_LANES(l) {
      const int h = l >> 5, j = l & 31;
      const u32 f = W.fl[l];
      const bool ok = okw[l] != 0, full = ok && fullm[l] != 0, go = ok && !full;
      const int L = Lv[l], pcur = W.p[l], mlen = mx[l];
      const int first = ctz32(bq[l] | 0x80000000u), cnt = 32 - __builtin_clz(bq[l] | 1u) - first;
      const int sn = (int)(W.cntr[l] & 0xffu), sufN = (int)(W.cntr[l] >> 8);
      const bool narrow = (u32)cnt < maxIv;                                    // ub > lb && ub - lb < maxInterval (:577-618)
      const bool over = go && narrow && (sufN + cnt > QM_DUO_SUF || sn >= QM_DUO_MAXIV);
      const bool rec = go && narrow && !over;
      const bool mine = rec && j >= first && j < first + cnt;
      // (a lane that records nothing writes to a slot of its own behind the stash: no branch around the store)
      QM_LDS(LeanSuf)* d = mine ? (QM_LDS(LeanSuf)*)&M.suf[h][sufN + j - first] : (QM_LDS(LeanSuf)*)&M.trash[l];
      d->tid = tdv[l]; d->pos = tpv[l]; d->qp = (u32)pcur; d->iv = (u32)sn;
      W.mins[l] = (rec && (u32)cnt < (W.mins[l] >> 8)) ? ((u32)sn | ((u32)cnt << 8)) : W.mins[l];
      W.cntr[l] += rec ? (1u + ((u32)cnt << 8)) : 0u;
      if (COV) {
        const int corr = W.prevEnd[l] > pcur ? W.prevEnd[l] - pcur : 0;
        W.cov[l] += rec ? mlen - corr : 0;
        W.prevEnd[l] = rec ? pcur + mlen : W.prevEnd[l];
      }
      // what follows the hit: the read is through, or the walk goes on at kp with a spot check (:602-611: only behind a recorded interval)
      const bool through = pcur + mlen >= L;
      const bool last = (f & QM_DW_LAST) != 0;
      const bool on = go && !over && !through && !(last && !rec);
      const int kp = pcur + mlen - (k - 1);                                    // NIP off: lce == matchedLen (:635-647)
      u32 f2 = f & ~(QM_DW_SPOT | QM_DW_STOP | QM_DW_SKIP);
      f2 |= (rec ? QM_DW_SPOT : 0u) | (last ? QM_DW_STOP : 0u) | ((kp + k == L) ? QM_DW_LAST : 0u);
      f2 = on ? f2 : (f & ~(QM_DW_MODE | QM_DW_SKIP));
      f2 |= (full || over) ? QM_DW_BAIL : 0u;
      W.fl[l] = ok ? f2 : f;
      W.p[l] = (ok && on) ? kp : pcur;
    }
  }
  // ---- the other strand's turn (:258 checkRC after the read's own pass, :271 checkFwd after the reverse complement's)?  Such a read is
  // left to the general kernel, like every read that bailed out above.  (:343-358: quasiCoverage)
  const int useCov = B.strict_check != 0 ? 1 : 0;          // disableNIP_ && strictCheck_ (SACollector.hpp:138)
  LV<int> nsuf, taken, snv, why;
  QM_LANES(l) {
    const u32 ha = W.hab[l] & 0xffffu, hb = W.hab[l] >> 16;
    const bool other = (W.fl[l] & QM_DW_FOUND) != 0 && (useCov ? (hb > 0) : (hb >= ha));
    int sn = (int)(W.cntr[l] & 0xffu);
    if (COV) { if (sn > 0 && Lv[l] > 0) { const double fr = (double)W.cov[l] / (double)Lv[l]; if (fr < B.quasi_cov) sn = 0; } }
    taken[l] = ((W.fl[l] & (QM_DW_DEF | QM_DW_BAIL)) != 0 || other) ? 0 : 1;
    why[l] = (W.fl[l] & QM_DW_DEF) ? ((W.fl[l] & QM_DW_DEFH) ? 1 : 0) : ((W.fl[l] & QM_DW_BAIL) ? 2 : 3);
    nsuf[l] = (taken[l] && sn > 0) ? (int)(W.cntr[l] >> 8) : 0;
    snv[l] = sn;
  }
  // ---- the pair this wave maps next: its images and its first probe's requests, under way while this pair is finished (the images
  // of this pair are not needed any more: what follows works on the suffix stashes)
  duo_prepare<PH>(ix, B, it + nw, nit, nw, par ^ 1, M, N);
  // ---- hitsToMappingsSimple (HitManager.cpp:691-882) for both mates at once, lane j of a half = suffix j of its stash (lean_h2m): a
  // transcript survives when it was seen in every interval, represented by the entry with the smallest position (the earlier interval
  // in processing order on ties); survivors go out in ascending transcript order
  LV<u64> elem; LV<u32> keepv; LV<int> slot; LV<u32> km;
  const int n0 = read_lane(nsuf, 0), n1 = read_lane(nsuf, 32);
  const int nmax = n0 > n1 ? n0 : n1;
  {
    LV<u32> tid, seen, lt; LV<u64> key;
    QM_LANES(l) {
      const int h = l >> 5, j = l & 31, n = nsuf[l];
      const QM_LDS(LeanSuf)* e = (const QM_LDS(LeanSuf)*)&M.suf[h][j < n ? j : 0];
      const u32 t = e->tid, ps = e->pos, qp = e->qp, iv = e->iv;
      const u32 mi = W.mins[l] & 0xffu;
      const u32 ord = iv == mi ? 0u : (iv < mi ? iv + 1u : iv);
      tid[l] = j < n ? t : 0xffffffffu;
      key[l] = j < n ? (((u64)ps << 32) | (u64)(ord * 64u + (u32)j)) : ~0ULL;
      seen[l] = j < n ? (1u << iv) : 0u; lt[l] = 0;
      elem[l] = mk_elem(t, (W.fl[l] & QM_DW_V) != 0, (int)(ps - qp));                       // pos - queryPos (:315, :761-767)
      keepv[l] = j < n ? 1u : 0u;
    }
    if (nmax > 0) {
      wave_fence();
      QM_LANES(l) {                                                            // the stash, rewritten for the all-pairs pass: {transcript, interval bit, (position, order)}; unused slots never match
        const int h = l >> 5, j = l & 31;
        QM_LDS(LeanSuf)* d = (QM_LDS(LeanSuf)*)&M.suf[h][j];
        d->tid = tid[l]; d->pos = seen[l]; d->qp = (u32)key[l]; d->iv = (u32)(key[l] >> 32);
      }
      wave_fence();
      for (int jj = 0; jj < nmax; ++jj) {
        QM_LANES(l) {
          const int h = l >> 5;
          const QM_LDS(LeanSuf)* e = (const QM_LDS(LeanSuf)*)&M.suf[h][jj];
          const u32 tj = e->tid, bj = e->pos; const u64 kj = ((u64)e->iv << 32) | (u64)e->qp;
          const bool same = tj == tid[l];
          if (same && kj < key[l]) keepv[l] = 0;
          if (tj < tid[l]) lt[l] |= 1u << jj;
          if (same) seen[l] |= bj;
        }
      }
    }
    LV<bool> kb;
    QM_LANES(l) {
      const int m = snv[l];
      const u32 all = m >= 32 ? 0xffffffffu : ((1u << m) - 1u);
      kb[l] = keepv[l] != 0 && seen[l] == all;
    }
    half_ballot(kb, km);
    QM_LANES(l) { keepv[l] = kb[l] ? 1u : 0u; slot[l] = popc32(km[l] & lt[l]); }   // slot = surviving entries with a smaller transcript id
  }
  const int tk0 = read_lane(taken, 0), tk1 = read_lane(taken, 32);
  const int cntA = tk0 ? popc32(read_lane(km, 0)) : 0, cntB = tk1 ? popc32(read_lane(km, 32)) : 0;
  const int r0 = 2 * it;
  if (tk0 && tk1 && B.pair_cnt && !B.fuzzy) {
    // ---- the pair, finished here: mergeLeftRightHits (RapMapUtils.hpp:1185-1264) + the per-pair driver (RapMapSAMapper.cpp:527-551,
    // 684-701) on the two sorted lists -- unit_merge.  The lists go through LDS in list order; lane i of half 0 takes left element i and
    // looks for its transcript among the right ones.
    wave_fence();
    QM_LANES(l) { if (keepv[l]) M.lst[l >> 5][slot[l]] = elem[l]; }
    wave_fence();
    PairCtr d;
    pair_merge(B, it, (const QM_LDS(u64)*)&M.lst[0][0], cntA, cntB, wa, d);
    ctr.pe += d.pe; ctr.se += d.se; ctr.tot += d.tot; ctr.reads += d.reads; ctr.tooMany += d.tooMany; ctr.mapped += d.mapped;
    return;
  }
  // ---- a mate was left to the general kernel (or the caller wants lists): the mapped mates' lists per read (finish_read), the others marked
  {
    const int na = cntA + cntB;
    long long base = 0;
    int fits = 1;
    if (na > 0) {
      if (wa.base < 0 || wa.used + na > QM_LEAN_CHUNK) {
        LV<u64> bv;
        QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor, (u64)QM_LEAN_CHUNK); }
        wa.base = (long long)read_lane(bv, 0); wa.used = 0;
      }
      base = wa.base + wa.used;
      if (base + na > B.lists_cap) { QM_LANES(l) { if (l == 0) *B.status |= 1; } fits = 0; base = 0; }
      else wa.used += na;
    }
    QM_LANES(l) {
      const int h = l >> 5, j = l & 31;
      const long long bh = base + (h ? cntA : 0);
      if (taken[l]) {
        if (fits && keepv[l]) B.lists[bh + slot[l]] = elem[l];
        if (j == 0) {
          const u32 flag = (B.fuzzy && (W.fl[l] & QM_DW_FOUND)) ? 0x80000000u : 0u;
          B.lcnt[r0 + h] = (fits ? (u32)(h ? cntB : cntA) : 0u) | flag; B.loff[r0 + h] = fits ? bh : 0;
          if (B.found_out) B.found_out[r0 + h] = (W.fl[l] & QM_DW_FOUND) ? 1 : 0;       // (stage views without interval records)
        }
      } else if (j == 0) {
        B.lcnt[r0 + h] = QM_LCNT_LEAN; B.loff[r0 + h] = 0; atomic_add_u64(B.cursor + QM_SC_LEANQ, 1ULL);
#ifndef QM_TIMING
        atomic_add_u64(B.cursor + QM_SC_DEFER0 + why[l], 1ULL);
#endif
      }
    }
  }
}

}  // namespace qm
