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

#define QM_DUO_SUF 32            // suffixes a read's intervals may hold together, and the widest interval
#define QM_DUO_MAXIV 32          // intervals per read (a bit each)

struct DuoMem {                                // one wave's LDS slab (2 000 bytes)
  u64 pk[2][2][8];                             // as LeanMem::pk: [mate][0: the read, 1: mirrored reverse complement][word]; words 4-7 stay zero
  union {
    LeanSuf suf[2][QM_DUO_SUF];                // [mate]: suffixes of the intervals recorded for it
    u64 lst[2][32];                            // ... later its hit list, sorted, for the merge
  };
  u32 stage[100];                              // raw characters of the next pair: dwords [0, 32) mate 0, [32, 64) mate 1, [64] / [96] the 33rd dword of mate 0 / 1
  u32 ostage[2][8];                            // offsets of the next / the next but one pair: off1[u], off1[u + 1], off2[u], off2[u + 1]
};
struct DuoCtr { u32 pe, se, tot, reads, tooMany, mapped; };   // HitCounters of the pairs this wave merged (wave-uniform)

// khash.find for one POSITION per lane: w the k-mer as the read has it, wr its reverse complement.  One bucket of the canonical
// table holds both; fh / (flb, fub): the k-mer is in the index / its interval, ch / (clb, cub): the same for the reverse complement.
// Lanes that are not `on` read bucket 0 and come back with nothing.
QM_DEV void duo_find(const DevIndex& ix, const LV<u64>& w, const LV<u64>& wr, const LV<u32>& on, LV<u32>& fh, LV<u32>& ch,
                     LV<u32>& flb, LV<u32>& fub, LV<u32>& clb, LV<u32>& cub) {
  LV<u32> more; LV<u64> bkt, ckv; LV<u32> bigv;
  QM_LANES(l) {
    const bool big = wr[l] < w[l];
    const u64 ck = big ? wr[l] : w[l];
    const u64 b = on[l] ? ((u64)bucket_hash(ck) & ix.hmask) : 0ULL;
    U4 a, f, r;
    load_48(&ix.slots[b], a, f, r);
    QM_CNT(1, on[l] ? 1 : 0);
    const u64 k0r = ((u64)a.y << 32) | a.x, k1 = ((u64)a.w << 32) | a.z;
    const bool h0 = (k0r & ~QM_BK_OVF) == ck, h1 = k1 == ck;
    const u32 cfl = h0 ? f.x : f.z, cfu = h0 ? f.y : f.w, crl = h0 ? r.x : r.z, cru = h0 ? r.y : r.w;   // the canonical k-mer's interval, its reverse complement's
    const bool m = on[l] != 0 && (h0 || h1);
    const u32 al = big ? crl : cfl, bl = big ? cfl : crl;
    flb[l] = al; fub[l] = big ? cru : cfu; clb[l] = bl; cub[l] = big ? cfu : cru;
    fh[l] = (m && al != QM_IV_NONE) ? 1u : 0u; ch[l] = (m && bl != QM_IV_NONE) ? 1u : 0u;
    more[l] = (on[l] != 0 && !(h0 || h1) && k0r != ~0ULL && (k0r & QM_BK_OVF) != 0) ? 1u : 0u;
    bkt[l] = b; ckv[l] = ck; bigv[l] = big ? 1u : 0u;
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
  QM_LANES(l) {
    bool h = false; u32 a = 0, b = 0;
    if (wf[l]) h = find_kmer<QM_F_PH>(ix, w[l], a, b);
    fh[l] = h ? 1u : 0u; flb[l] = a; fub[l] = b;
  }
  QM_LANES(l) {
    bool h = false; u32 a = 0, b = 0;
    if (wc[l]) h = find_kmer<QM_F_PH>(ix, wr[l], a, b);
    ch[l] = h ? 1u : 0u; clb[l] = a; cub[l] = b;
  }
}

// what a half's walk carries (every member: one value per lane, the same in the 32 lanes of a half)
struct DuoWalk {
  LV<int> mode;                                // 0: nothing (more) to do, 1: first-hit scan (SACollector.hpp:167-237), 2: getSAHits_ over strand V
  LV<int> p, V, skip, spot, stopAfter, lastSearch, width, prevEnd;
  LV<int> wb, ww; LV<u32> Fm, Cm;              // the window: positions [wb, wb + ww) of the strand, bit j = k-mer / reverse complement of position wb + j found
  LV<u32> lbw, ubw;                            // ... lane j of the half: the interval of the k-mer at position wb + j
  LV<u32> lb, ub;                              // the interval the walk stands on
  LV<u32> ha, hb;                              // hits of the walked strand / of the other one (SACollector.hpp:258,271)
  LV<int> sn, sufN, minIdx, minSpan, cov, bail, foundHit;
};

// Probe positions [p, p + width) of its strand for every half that asked (`need`): lane j of the half looks up position p + j -- the
// k-mer out of the strand's image, its reverse complement out of the other one (lean_probe) -- and the half's window is replaced.
template <bool PH>
QM_DEV void duo_probe(const DevIndex& ix, const QM_LDS(u64)* pkw, int k, const LV<int>& Pv, const LV<int>& Lv, const LV<u32>& need, DuoWalk& W) {
  LV<u64> w, wr; LV<u32> on, fh, ch, flb, fub, clb, cub;
  LV<int> wwn;
  QM_LANES(l) {
    const int h = l >> 5, j = l & 31;
    const int P = Pv[l], D = QM_LEAN_MAXLEN - Lv[l];
    const int V = W.mode[l] == 2 ? W.V[l] : 0;
    int nw = W.width[l];
    if (W.p[l] + nw > P) nw = P - W.p[l];
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
    if (need[l]) { W.wb[l] = W.p[l]; W.ww[l] = wwn[l]; W.Fm[l] = fm[l]; W.Cm[l] = cm[l]; W.lbw[l] = flb[l]; W.ubw[l] = fub[l]; }
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

// One pair: reads 2 it and 2 it + 1.
template <bool PH>
QM_DEV void duo_iter(const DevIndex& ix, const ReadBatch& B, int it, int nit, int nw, int par, DuoMem& M, WaveAlloc& wa, DuoCtr& ctr) {
  const int k = ix.k;
  const QM_LDS(u64)* pkw = (const QM_LDS(u64)*)&M.pk[0][0][0];
  // ---- the two mates' characters -> 2-bit images of both strands, four characters per lane (lean_iter)
  LV<int> rawv, Lv, Pv;
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
      rawv[l] = raw; Lv[l] = len; Pv[l] = len - k + 1;
    }
  }
  LV<u32> dirty, reps;
  half_ballot(bad, dirty); half_ballot(rep, reps);
  wave_fence();
  // the staging rows are free again: the next pair's characters, and the offsets of the one after it
  duo_stage_chars(B, it + nw, nit, M, par ^ 1);
  duo_stage_offsets(B, it + 2 * nw, nit, M, par);
  // what this kernel takes: no character but A C G T, no window of k equal bases, at most 128 characters (lean_iter)
  LV<int> defv;
  QM_LANES(l) { defv[l] = (rawv[l] > QM_LEAN_MAXLEN || dirty[l] != 0 || 4 * popc32(reps[l]) + 6 >= k) ? 1 : 0; }
  // ---- the first probe of both mates in one round (SACollector.hpp:167-237 starts at position 0; the read's last k-mer is the first
  // thing the reverse-complement pass asks for): lane 0 of a half = position 0, lane 1 = position P - 1, both orientations each
  DuoWalk W;
  LV<u32> F0, C0, Fl, Cl, s0lb, s0ub, rlb, rub;
  {
    LV<u64> w, wr; LV<u32> on, fh, ch, flb, fub, clb, cub;
    QM_LANES(l) {
      const int h = l >> 5, jj = l & 31;
      const int P = Pv[l], D = QM_LEAN_MAXLEN - Lv[l];
      const bool o = !defv[l] && P >= 1 && jj < 2 && (jj == 0 || P > 1);
      const int q = (o && jj == 1) ? P - 1 : 0;
      const QM_LDS(u64)* pkh = pkw + 16 * h;
      w[l] = lean_kmer(pkh, q, k);
      wr[l] = lean_kmer(pkh + 8, (o ? P - 1 - q : 0) + D, k);
      on[l] = o ? 1u : 0u;
    }
    QM_CNT(3, 1);
    if (PH) duo_find_ph(ix, w, wr, on, fh, ch, flb, fub, clb, cub);
    else duo_find(ix, w, wr, on, fh, ch, flb, fub, clb, cub);
    lds_dma_wait();                                        // what was requested above has landed by now: no store follows an open request
    LV<bool> fb, cb; LV<u32> fm, cm; LV<int> i0, i1;
    QM_LANES(l) { fb[l] = fh[l] != 0; cb[l] = ch[l] != 0; i0[l] = 0; i1[l] = Pv[l] > 1 ? 1 : 0; }
    half_ballot(fb, fm); half_ballot(cb, cm);
    half_read(flb, i0, s0lb); half_read(fub, i0, s0ub);    // the interval of the read's first k-mer ...
    half_read(clb, i1, rlb); half_read(cub, i1, rub);      // ... and of the reverse complement of its last one = the first k-mer of reverseRead(read)
    QM_LANES(l) {
      F0[l] = fm[l] & 1u; C0[l] = cm[l] & 1u;
      Fl[l] = Pv[l] > 1 ? (fm[l] >> 1) & 1u : F0[l]; Cl[l] = Pv[l] > 1 ? (cm[l] >> 1) & 1u : C0[l];
    }
  }
  const u32 maxIv = (u32)B.max_interval;
  // ---- where the walks start: the read itself from its first hit when that hit is a forward one (SACollector.hpp:247-254), else
  // reverseRead(read) from 0 (:258-265) -- lean_iter; a read whose first k-mer is in the index in neither orientation scans on (mode 1)
  QM_LANES(l) {
    const bool ok = !defv[l] && Pv[l] >= 1;
    W.mode[l] = 0; W.p[l] = 0; W.V[l] = 0; W.skip[l] = 0; W.spot[l] = 0; W.stopAfter[l] = 0; W.lastSearch[l] = 0; W.width[l] = 32; W.prevEnd[l] = 0;
    W.wb[l] = 0; W.ww[l] = 1; W.Fm[l] = F0[l]; W.Cm[l] = C0[l]; W.lbw[l] = s0lb[l]; W.ubw[l] = s0ub[l];
    W.lb[l] = 0; W.ub[l] = 0; W.ha[l] = 0; W.hb[l] = 0;
    W.sn[l] = 0; W.sufN[l] = 0; W.minIdx[l] = 0; W.minSpan[l] = 0x7fffffff; W.cov[l] = 0; W.bail[l] = 0; W.foundHit[l] = 0;
    if (ok) {
      if (F0[l]) { W.mode[l] = 2; W.foundHit[l] = 1; W.skip[l] = 1; W.lb[l] = s0lb[l]; W.ub[l] = s0ub[l]; W.ha[l] = 1; W.hb[l] = C0[l]; }
      else if (C0[l]) {
        W.mode[l] = 2; W.foundHit[l] = 1; W.V[l] = 1; W.ha[l] = 1;
        W.Fm[l] = Cl[l]; W.Cm[l] = Fl[l]; W.lbw[l] = rlb[l]; W.ubw[l] = rub[l];
      } else { W.mode[l] = 1; W.p[l] = 1; }
    }
  }
  // ---- the two walks in lockstep
  while (true) {
    // A: a walk that stands behind its read's last k-mer is through; a window is probed where a walk stands outside its own
    LV<u32> need; LV<bool> live, nb;
    QM_LANES(l) {
      if (W.mode[l] != 0 && !W.skip[l] && W.p[l] >= Pv[l]) W.mode[l] = 0;      // (scan: no hit anywhere; walk: `if (p >= P) break`)
      const bool inw = (unsigned)(W.p[l] - W.wb[l]) < (unsigned)W.ww[l];
      need[l] = (W.mode[l] != 0 && !W.skip[l] && !inw) ? 1u : 0u;
      live[l] = W.mode[l] != 0; nb[l] = need[l] != 0;
    }
    if (!ballot(live)) break;
    if (ballot(nb)) duo_probe<PH>(ix, pkw, k, Pv, Lv, need, W);
    // B1: the first-hit scan (SACollector.hpp:167-237): the first position whose k-mer or reverse complement is in the hash
    {
      LV<bool> sc; QM_LANES(l) { sc[l] = W.mode[l] == 1; }
      if (ballot(sc)) {
        LV<int> rel0; LV<u32> found, t0, t1;
        QM_LANES(l) {
          found[l] = 0; rel0[l] = 0;
          if (W.mode[l] == 1) {
            const int rel = W.p[l] - W.wb[l];
            const u32 mm = (W.Fm[l] | W.Cm[l]) >> rel;
            if (mm) { W.p[l] += ctz32(mm); found[l] = 1; rel0[l] = W.p[l] - W.wb[l]; }
            else W.p[l] = W.wb[l] + W.ww[l];
          }
        }
        half_read(W.lbw, rel0, t0); half_read(W.ubw, rel0, t1);
        QM_LANES(l) {
          if (found[l]) {
            W.mode[l] = 2; W.foundHit[l] = 1; W.ha[l] = 1;
            if ((W.Fm[l] >> rel0[l]) & 1u) { W.V[l] = 0; W.skip[l] = 1; W.lb[l] = t0[l]; W.ub[l] = t1[l]; W.hb[l] = (W.Cm[l] >> rel0[l]) & 1u; }
            else {
              // what the first probe learned about the read's last k-mer is the first k-mer of reverseRead(read)
              W.V[l] = 1; W.hb[l] = 0; W.p[l] = 0; W.wb[l] = 0; W.ww[l] = 1; W.Fm[l] = Cl[l]; W.Cm[l] = Fl[l]; W.lbw[l] = rlb[l]; W.ubw[l] = rub[l];
            }
          }
        }
      }
    }
    // B2: SACollector::getSAHits_ (SACollector.hpp:441-677, NIP disabled) steps to the walk's next hit inside the window
    LV<u32> ext; LV<int> relh; LV<bool> eb;
    QM_LANES(l) {
      u32 e = (W.mode[l] == 2 && W.skip[l]) ? 1u : 0u;
      if (W.mode[l] == 2 && !W.skip[l]) {
        const int rel = W.p[l] - W.wb[l];
        const u32 fm = W.Fm[l] >> rel, cm = W.Cm[l] >> rel;                    // (no bits beyond the window)
        const int avail = W.ww[l] - rel;
        bool go = true;
        if (W.spot[l]) {                                                       // the k-mer the walk goes on with, spot-checked (:602-611)
          W.ha[l] += fm & 1u; W.hb[l] += cm & 1u; W.spot[l] = 0;
          if (W.stopAfter[l]) { W.mode[l] = 0; go = false; }
        }
        if (go) {
          const u32 below = (fm & (0u - fm)) - 1u;                             // the positions before the first hit (all of them without one)
          W.hb[l] += (u32)popc32(cm & ~fm & below);                            // misses: spotCheck_ of the complement (:667-675)
          if (!fm) W.p[l] += avail;
          else {
            const int ph = ctz32(fm);
            W.ha[l] += 1;                                                      // spotCheck_ on the hit (:545)
            W.hb[l] += (cm >> ph) & 1u;
            W.p[l] += ph; e = 1u;
          }
        }
      }
      ext[l] = e; eb[l] = e != 0; relh[l] = W.p[l] - W.wb[l];
    }
    if (!ballot(eb)) continue;
    {
      LV<u32> t0, t1;
      half_read(W.lbw, relh, t0); half_read(W.ubw, relh, t1);
      QM_LANES(l) { if (ext[l] && !W.skip[l]) { W.lb[l] = t0[l]; W.ub[l] = t1[l]; } W.skip[l] = 0; }
    }
    // C: the MMP extension (SASearcher.hpp:88-309) in the closed form of extend_search_wide against the packed characters behind
    // every suffix's k-mer (one lane per suffix, one 32-byte load each, lean_iter); the (transcript, position) words of the block it
    // settles on go to the half's stash
    LV<int> lc; LV<u32> tdv, tpv, okw; LV<bool> fullv;
    QM_LANES(l) {
      const int h = l >> 5, j = l & 31;
      const u32 lbIn = W.lb[l] ? W.lb[l] - 1 : 0;                               // :553
      const int wiv = (int)(W.ub[l] - lbIn - 1);
      const bool e = ext[l] != 0;
      const bool okv = e && wiv >= 1 && wiv <= QM_DUO_SUF;
      if (e && !okv) { W.bail[l] = 1; W.mode[l] = 0; }
      const int L = Lv[l], V = W.V[l];
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
    }
    LV<u32> fullm; half_ballot(fullv, fullm);
    LV<int> mx; QM_LANES(l) { mx[l] = lc[l]; }
    half_max(mx);
    LV<bool> best; QM_LANES(l) { best[l] = okw[l] != 0 && lc[l] == mx[l]; }
    LV<u32> bq; half_ballot(best, bq);
    QM_LANES(l) {
      const int h = l >> 5, j = l & 31;
      if (okw[l] && fullm[l]) { W.bail[l] = 1; W.mode[l] = 0; }
      else if (okw[l]) {
        const int L = Lv[l], pcur = W.p[l], mlen = mx[l];
        const u32 lbIn = W.lb[l] ? W.lb[l] - 1 : 0;
        const int first = ctz32(bq[l]), cnt = 32 - __builtin_clz(bq[l] | 1u) - first;
        W.lb[l] = lbIn + 1 + (u32)first; W.ub[l] = W.lb[l] + (u32)cnt;
        const int kp = pcur + mlen - (k - 1);
        int recorded = 0;
        if ((u32)cnt < maxIv) {                                                // ub > lb && ub - lb < maxInterval (:577-618)
          if (W.sufN[l] + cnt > QM_DUO_SUF || W.sn[l] >= QM_DUO_MAXIV) { W.bail[l] = 1; W.mode[l] = 0; }
          else {
            if (j >= first && j < first + cnt) {
              QM_LDS(LeanSuf)* d = (QM_LDS(LeanSuf)*)&M.suf[h][W.sufN[l] + j - first];
              d->tid = tdv[l]; d->pos = tpv[l]; d->qp = (u32)pcur; d->iv = (u32)W.sn[l];
            }
            if (cnt < W.minSpan[l]) { W.minSpan[l] = cnt; W.minIdx[l] = W.sn[l]; }
            W.sufN[l] += cnt; W.sn[l] += 1;
            const int corr = W.prevEnd[l] > pcur ? W.prevEnd[l] - pcur : 0;
            W.cov[l] += mlen - corr;
            W.prevEnd[l] = pcur + mlen;
            recorded = 1;
          }
        }
        if (W.mode[l] != 0) {
          if (pcur + mlen >= L) W.mode[l] = 0;                                 // the read is through
          else {
            W.spot[l] = recorded;                                              // (:602-611: only behind a recorded interval; kp < P here)
            W.stopAfter[l] = W.lastSearch[l];
            if (W.lastSearch[l] && !recorded) W.mode[l] = 0;
            else {
              W.p[l] = kp;                                                     // NIP off: lce == matchedLen (:635-647)
              W.width[l] = W.stopAfter[l] ? 1 : 32;
              if (kp + k == L) W.lastSearch[l] = 1;
            }
          }
        }
      }
    }
  }
  // ---- the other strand's turn (:258 checkRC after the read's own pass, :271 checkFwd after the reverse complement's)?  Such a read is
  // left to the general kernel, like every read that bailed out above.  (:343-358: quasiCoverage)
  const int useCov = B.strict_check != 0 ? 1 : 0;          // disableNIP_ && strictCheck_ (SACollector.hpp:138)
  LV<int> nsuf, taken;
  if (B.quasi_cov > 0.0) {
    QM_LANES(l) { if (W.sn[l] > 0 && Lv[l] > 0) { const double f = (double)W.cov[l] / (double)Lv[l]; if (f < B.quasi_cov) W.sn[l] = 0; } }
  }
  QM_LANES(l) {
    if (W.foundHit[l] && !W.bail[l] && (useCov ? (W.hb[l] > 0) : (W.hb[l] >= W.ha[l]))) W.bail[l] = 1;
    taken[l] = (defv[l] || W.bail[l]) ? 0 : 1;
    nsuf[l] = (taken[l] && W.sn[l] > 0) ? W.sufN[l] : 0;
  }
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
      const u32 mi = (u32)W.minIdx[l];
      const u32 ord = iv == mi ? 0u : (iv < mi ? iv + 1u : iv);
      tid[l] = j < n ? t : 0xffffffffu;
      key[l] = j < n ? (((u64)ps << 32) | (u64)(ord * 64u + (u32)j)) : ~0ULL;
      seen[l] = j < n ? (1u << iv) : 0u; lt[l] = 0;
      elem[l] = mk_elem(t, W.V[l] != 0, (int)(ps - qp));                       // pos - queryPos (:315, :761-767)
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
      const int m = W.sn[l];
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
    const int maxHits = B.max_num_hits;
    LV<u64> mine, partner; LV<bool> fnd;
    QM_LANES(l) {
      const int h = l >> 5, i = l & 31;
      mine[l] = M.lst[h][i < (h ? cntB : cntA) ? i : 0];
      partner[l] = 0; fnd[l] = false;
    }
    if (cntA > 0) {
      for (int jj = 0; jj < cntB; ++jj) {
        QM_LANES(l) {
          const u64 f = M.lst[1][jj];
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
      ctr.pe += (u32)nm;
    } else {
      const int no = cntA + cntB;
      const int keepAll = (!tooMany && no > 0 && no <= maxHits && !B.no_orphans) ? 1 : 0;   // RapMapSAMapper.cpp:534-551
      if (!tooMany && no > 0) ctr.se += (u32)no;
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
    ctr.reads += 1; ctr.tooMany += (u32)tooMany; ctr.tot += (u32)cnt; ctr.mapped += cnt > 0 ? 1u : 0u;
    QM_LANES(l) { if (l == 0) { B.pair_cnt[it] = (u32)cnt; B.lcnt[r0] = QM_LCNT_PAIR; B.loff[r0] = base; } }
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
          const u32 flag = (B.fuzzy && W.foundHit[l]) ? 0x80000000u : 0u;
          B.lcnt[r0 + h] = (fits ? (u32)(h ? cntB : cntA) : 0u) | flag; B.loff[r0 + h] = fits ? bh : 0;
        }
      } else if (j == 0) { B.lcnt[r0 + h] = QM_LCNT_LEAN; B.loff[r0 + h] = 0; atomic_add_u64(B.cursor + QM_SC_LEANQ, 1ULL); }
    }
  }
}

}  // namespace qm
