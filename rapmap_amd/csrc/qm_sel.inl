// qm_sel.inl -- the selective-alignment (-s) variant of the path (SURVEY.md section 8 row a17 and the chaining /
// multi-position branches of rows a9-a11, a13).  Included at the end of qm_mapper.inl, namespace qm.
//
//   stage A   sel_hits_to_mappings: after the (chain-scoring) collector, the wave's lanes fetch (transcript, offset)
//             of every suffix of every SA interval in parallel -- the memory-bound part -- and lane 0 then replays
//             intersectSAHits with slack (HitManager.cpp:587-689), the minimap2-style chaining DP with equally good
//             chains kept as multiple positions (:84-326) and mergeOrientationUnique (:834-881) on that small set.
//             A read's list becomes groups of words:  header = tid | primaryRC<<32 | chainStatus<<33 | nP<<36, then
//             own position | nO<<32, then nP positions of the surviving orientation and nO positions of the other one.
//   stage B+C plan / align / finish: mergeLeftRightHitsFuzzy on the position lists (RapMapUtils.hpp:864-1183) and
//             getAlnScore for every hit (SelectiveAlignmentUtils.hpp:260-373) per unit, the ksw_extz2_sse41 alignments
//             that have to be run four to a wavefront with the SSE lanes reproduced exactly (src/ksw2pp/ksw2_extz2_sse.c),
//             then the score gate and the soft / hard filter (RapMapSAMapper.cpp:554-667, 246-318).
// The double arithmetic of the chain score must not be contracted into FMAs: the reference is plain x86-64 code.

#define QM_SEL_CAP 4096            // SA entries of one strand's intervals a read may bring (else status bit 3)
enum : int { QM_CS_PERFECT = 0, QM_CS_UNGAPPED = 1, QM_CS_REGULAR = 4 };   // rapmap::utils::ChainStatus

struct SelRec { u32 tid, pos, qpos, len, iv; };
struct SelGroup {                  // ppos: the hit's own position (QuasiAlignment::pos); 24 bytes (the LDS edition holds 96 of them)
  u32 tid; int ppos; double score; int npos; u32 offcs;      // offcs: first position in pos[] (28 bits) | chain status << 28
  QM_DEV int off() const { return (int)(offcs & 0x0fffffffu); }
  QM_DEV int cs() const { return (int)(offcs >> 28); }
  QM_DEV void set_off(int o) { offcs = (offcs & 0xf0000000u) | (u32)o; }
  QM_DEV void set_cs(int c) { offcs = (offcs & 0x0fffffffu) | ((u32)c << 28); }
};
template <int CAP, int OUTCAP>
struct SelScratchT {                // working set of one read
  static constexpr int NCH = 4;     // 64-record chunks the lane-parallel sort / chaining handle (sel_wave_sort)
  QM_DEV int cap() const { return CAP; } QM_DEV int outcap() const { return OUTCAP; } QM_DEV int gcap() const { return CAP; }
  QM_DEV int tmp_bytes() const { return (int)sizeof(tmp); }
  SelRec rec[CAP], tmp[CAP];
  double f[CAP]; int p[CAP]; int seen[CAP]; int ends[CAP]; int starts[CAP];
  SelGroup grp[2][CAP]; int pos[2][CAP]; int ngrp[2], npos[2];
  u64 out[OUTCAP];
  QM_DEV SelGroup* grpp(int s) { return grp[s]; }
};
struct SelScratch : SelScratchT<QM_SEL_CAP, QM_CHUNK> {};   // per wave, global memory: the general case
// The third edition: arrays sized by the host from the largest read of the slow queue (reads whose intervals hold more
// suffixes than SelScratch -- repeats, low-complexity reads).  Same member names, pointers instead of arrays.
struct SelScratchDyn {
  static constexpr int NCH = 4;
  int capN, outN;
  SelRec* rec; SelRec* tmp;
  double* f; int* p; int* seen; int* ends; int* starts;
  SelGroup* grp0; SelGroup* grp1; int* pos[2]; int* ngrp; int* npos;
  u64* out;
  QM_DEV int cap() const { return capN; } QM_DEV int outcap() const { return outN; } QM_DEV int gcap() const { return capN; }
  QM_DEV int tmp_bytes() const { return capN * (int)sizeof(SelRec); }
  QM_DEV SelGroup* grpp(int s) { return s == 0 ? grp0 : grp1; }
  // bytes of device memory one wave's arrays take for `n` suffixes per strand (host: qmk_sel_dyn_bytes / qmk_sel_dyn_bind)
  static inline unsigned long long bytes_for(long long n) {
    return (unsigned long long)n * (2 * sizeof(SelRec) + sizeof(double) + 4 * sizeof(int) + 2 * sizeof(SelGroup) + 2 * sizeof(int) + 6 * sizeof(u64)) + 64;
  }
  inline void bind(unsigned char* base, long long n) {
    capN = (int)n; outN = (int)(6 * n);
    unsigned char* q = base;
    f = (double*)q; q += n * sizeof(double);
    out = (u64*)q; q += 6 * n * sizeof(u64);
    grp0 = (SelGroup*)q; q += n * sizeof(SelGroup); grp1 = (SelGroup*)q; q += n * sizeof(SelGroup);
    rec = (SelRec*)q; q += n * sizeof(SelRec); tmp = (SelRec*)q; q += n * sizeof(SelRec);
    p = (int*)q; q += n * 4; seen = (int*)q; q += n * 4; ends = (int*)q; q += n * 4; starts = (int*)q; q += n * 4;
    pos[0] = (int*)q; q += n * 4; pos[1] = (int*)q; q += n * 4;
    ngrp = (int*)q; npos = ngrp + 2;
  }
};
#ifndef QM_SEL_SMALL
#define QM_SEL_SMALL 64
#endif
// The LDS edition (almost every read fits): one 64-record chunk, 6.4 KB per wave -- four waves per SIMD stay resident in
// qm_h2m_kernel<QM_F_SEL>, which the kernel's registers limit it to anyway.  The sort keys
// share their bytes with the second strand's groups (written only after that strand's sort), and the read's list is
// assembled in the sort buffers of WaveMem, which the -s path does not use otherwise (`out`, 3 * QM_CAP words).
struct SelScratchLds {
  static constexpr int NCH = 1;     // at most QM_SEL_SMALL < 64 records: one chunk, a quarter of the code and registers
  QM_DEV int cap() const { return QM_SEL_SMALL; } QM_DEV int outcap() const { return 3 * QM_CAP; } QM_DEV int gcap() const { return QM_SEL_SMALL; }
  QM_DEV int tmp_bytes() const { return (int)sizeof(tmp); }
  SelRec rec[QM_SEL_SMALL];
  double f[QM_SEL_SMALL]; int p[QM_SEL_SMALL]; int seen[QM_SEL_SMALL]; int ends[QM_SEL_SMALL]; int starts[QM_SEL_SMALL];
  SelGroup grp0[QM_SEL_SMALL];
  union { SelGroup grp1[QM_SEL_SMALL]; SelRec tmp[QM_SEL_SMALL]; };
  int pos[2][QM_SEL_SMALL]; int ngrp[2], npos[2];
  QM_LDS(u64)* out;                 // (typed: a plain pointer read back from this struct would be generic, its stores FLAT)
  QM_DEV SelGroup* grpp(int s) { return s == 0 ? grp0 : grp1; }
};
QM_DEV const u64* sel_out(const SelScratch& S) { return S.out; }

QM_DEV u64 sel_header(u32 tid, bool primaryRC, int cs, int nP) {
  return (u64)tid | ((u64)(primaryRC ? 1 : 0) << 32) | ((u64)(cs & 7) << 33) | ((u64)(u32)nP << 36);
}
QM_DEV u32 selh_tid(u64 h) { return (u32)h; }
QM_DEV bool selh_rc(u64 h) { return (h >> 32) & 1; }
QM_DEV int selh_cs(u64 h) { return (int)((h >> 33) & 7); }
QM_DEV int selh_np(u64 h) { return (int)(h >> 36); }

// fastapprox's fastlog2 as used by the chain score (HitManager.cpp:33-42)
QM_DEV float sel_fastlog2(float x) {
#pragma clang fp contract(off)
  union { float f; u32 i; } vx = { x };
  union { u32 i; float f; } mx = { (vx.i & 0x007FFFFF) | 0x3f000000 };
  float y = (float)vx.i;
  y *= 1.1920928955078125e-7f;
  return y - 124.22551499f - 1.498030302f * mx.f - 1.72587999f / (0.3520887068f + mx.f);
}

// stable merge sort of r[0..n) (scratch t), by `less`
template <typename Less>
QM_DEV void sel_sort(SelRec* r, SelRec* t, int n, Less less) {
  for (int w = 1; w < n; w <<= 1) {
    for (int lo = 0; lo < n; lo += 2 * w) {
      int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int i = lo, j = mid, o = lo;
      while (i < mid && j < hi) t[o++] = less(r[j], r[i]) ? r[j++] : r[i++];
      while (i < mid) t[o++] = r[i++];
      while (j < hi) t[o++] = r[j++];
    }
    for (int i = 0; i < n; ++i) r[i] = t[i];
  }
}

// collectHitsSimpleSA, chaining branch (HitManager.cpp:107-307), for the hn hits H of one transcript (already in chain
// order).  f, p, seen, ends, starts: hn entries of scratch each.  Returns the number of chain starts (0: no hit), fills g
// (all but its offset) and writes the sorted start positions to posOut (may alias ends).
QM_DEV int sel_chain_group(const SelRec* H, int hn, double* f, int* p, int* seen, int* ends, int* starts, int maxDist,
                           SelGroup& g, int* posOut) {
#pragma clang fp contract(off)
  double bestScore = -1.7976931348623157e308; int lastBest = -1; int nEnds = 0;
  const double seedLen = 31.0;
  for (int i = 0; i < hn; ++i) {
    const u32 qEnd = H[i].qpos + H[i].len, tEnd = H[i].pos + H[i].len;
    const double leni = (double)(int)H[i].len;
    int pi = i; double fi = (double)H[i].len;
    int looksLeft = 2;
    for (int j = i - 1; j >= 0; --j) {
      const u32 qEndJ = H[j].qpos + H[j].len, tEndJ = H[j].pos + H[j].len;
      const int dq = (int)(qEnd - qEndJ), dt = (int)(tEnd - tEndJ);
      // alpha
      double gapMin = (dq < dt) ? (double)dq : (double)dt;
      double alpha = (leni < gapMin) ? leni : gapMin;
      // beta
      double beta;
      if (dq < 0 || ((dq > dt ? dq : dt) > maxDist)) beta = __builtin_inf();
      else {
        double l = (double)dq - (double)dt;
        int al = (int)(l < 0 ? -l : l);
        beta = (l == 0) ? 0.0 : (0.01 * seedLen * al + 0.5 * sel_fastlog2((float)al));
      }
      double cand = f[j] + alpha - beta;
      bool take = cand > fi;
      pi = take ? j : pi;
      fi = take ? cand : fi;
      if (pi < i) { looksLeft--; if (looksLeft <= 0) break; }
    }
    p[i] = pi; f[i] = fi;
    if (fi > bestScore) { bestScore = fi; lastBest = i; nEnds = 0; ends[nEnds++] = i; }
    else if (fi == bestScore) ends[nEnds++] = i;
  }
  // multi-chain backtracking (:206-246)
  for (int i = 0; i < hn; ++i) seen[i] = 0;
  int nOptimal = 0, nStarts = 0;
  for (int e = 0; e < nEnds; ++e) {
    int cur = ends[e];
    bool fresh = true;
    int prev = p[cur];
    while (prev < cur) {
      if (seen[cur]) { fresh = false; break; }
      seen[cur] = 1;
      cur = prev;
      prev = p[cur];
    }
    if (seen[cur]) fresh = false;
    if (fresh) { ++nOptimal; starts[nStarts++] = prev; }
  }
  if (nStarts == 0) return 0;
  g.tid = H[0].tid; g.offcs = 0; g.set_cs(QM_CS_REGULAR); g.score = bestScore; g.npos = nStarts;
  g.ppos = (int)(H[starts[0]].pos - H[starts[0]].qpos);                      // the first chain's start (:259-262)
  for (int t = 0; t < nStarts; ++t) {                                        // allPositions is sorted (:272-276)
    const int v = (int)(H[starts[t]].pos - H[starts[t]].qpos);
    int b = t - 1;
    while (b >= 0 && posOut[b] > v) { posOut[b + 1] = posOut[b]; --b; }
    posOut[b + 1] = v;
  }
  if (hn > 1 && nOptimal == 1 && lastBest == hn - 1) {             // gapless chain (:283-305)
    long long qSpan = (long long)(H[hn - 1].qpos + H[hn - 1].len) - (long long)H[0].qpos;
    long long tSpan = (long long)(H[hn - 1].pos + H[hn - 1].len) - (long long)H[0].pos;
    if (qSpan == tSpan && qSpan == (long long)maxDist) g.set_cs(QM_CS_UNGAPPED);
  }
  return nStarts;
}

// set of SA intervals a transcript occurs in, for the lane-parallel path (at most 64 * QM_SEL_CHUNKS = 256 records, hence at
// most 256 intervals): four words updated by selects -- indexing an array with iv >> 6 would put it in scratch memory
struct SelIvSet {
  u64 a = 0, b = 0, c = 0, d = 0;
  QM_DEV void add(u32 iv) { const u64 bit = 1ULL << (iv & 63); const u32 w = iv >> 6; a |= w == 0 ? bit : 0ULL; b |= w == 1 ? bit : 0ULL; c |= w == 2 ? bit : 0ULL; d |= w == 3 ? bit : 0ULL; }
  QM_DEV int count() const { return popc64(a) + popc64(b) + popc64(c) + popc64(d); }
};

// One strand: S.rec[0..n) holds every (tid, pos, qpos, len, interval) of the strand's m intervals.  Lane-0 code.
template <typename SS>
QM_DEV void sel_strand(SS& S, int s, int n, int m, u32 readLen, int mate, float consensusFraction, bool presorted) {
#pragma clang fp contract(off)
  int ng = 0, np = 0;
  SelGroup* G = S.grpp(s); int* P = S.pos[s];
  if (m == 1) {
    // collectFromSingleInterval + mergeUnique (HitManager.cpp:716-807): hitPos = pos - queryPos, sorted by (tid, hitPos)
    if (!presorted) sel_sort(S.rec, S.tmp, n, [](const SelRec& a, const SelRec& b) {
      return a.tid != b.tid ? a.tid < b.tid : (int)(a.pos - a.qpos) < (int)(b.pos - b.qpos); });
    for (int i = 0; i < n; ++i) {
      const SelRec& r = S.rec[i];
      if (ng == 0 || G[ng - 1].tid != r.tid) {
        SelGroup g; g.tid = r.tid; g.offcs = 0; g.set_cs(r.len == readLen ? QM_CS_PERFECT : QM_CS_REGULAR); g.score = -1.7976931348623157e308; g.npos = 0; g.set_off(np);
        g.ppos = (int)(r.pos - r.qpos);
        G[ng++] = g;
      }
      P[np++] = (int)(r.pos - r.qpos); G[ng - 1].npos++;
    }
    S.ngrp[s] = ng; S.npos[s] = np;
    return;
  }
  // intersectSAHits (HitManager.cpp:587-689): a transcript is kept when it occurs in at least requiredNumHits of the m
  // intervals; with slack every occurrence is recorded, without it only transcripts present in ALL intervals survive,
  // whose occurrences are then all recorded as well -- so both cases reduce to counting distinct intervals per transcript.
  const float requiredFrac = (float)m * consensusFraction;
  int requiredNumHits = m, maxSlack = 0;
  if (consensusFraction < 1.0) {
    int fl = (int)requiredFrac;                          // floor of a non-negative float
    requiredNumHits = fl > 1 ? fl : 1;
    maxSlack = m - requiredNumHits;
  }
  // chain order within a transcript: by reference end, then query end (HitManager.cpp:129-139); then group by transcript
  if (!presorted) {
    sel_sort(S.rec, S.tmp, n, [](const SelRec& a, const SelRec& b) {
      u32 r1 = a.pos + a.len, r2 = b.pos + b.len, q1 = a.qpos + a.len, q2 = b.qpos + b.len;
      return (r1 < r2) ? true : ((r2 < r1) ? false : (q1 < q2)); });
    sel_sort(S.rec, S.tmp, n, [](const SelRec& a, const SelRec& b) { return a.tid < b.tid; });
  }
  // first pass: is any transcript active?  Distinct intervals of a transcript are counted with stamps (S.p[iv] = group): any
  // number of intervals (long reads bring hundreds); S.p is free until the chaining below, and n >= m entries long
  bool anyActive = false;
  for (int i = 0; i < m; ++i) S.p[i] = -1;
  for (int g0 = 0; g0 < n;) {
    int g1 = g0; int na = 0;
    while (g1 < n && S.rec[g1].tid == S.rec[g0].tid) { const int iv = (int)S.rec[g1].iv; if (S.p[iv] != g0) { S.p[iv] = g0; ++na; } ++g1; }
    S.seen[g0] = na;                                   // parked: #intervals of the group starting at g0
    if (na >= requiredNumHits) anyActive = true;
    g0 = g1;
  }
  const bool allActive = maxSlack > 0 && !anyActive;   // :682-686
  const int maxDist = (int)readLen;
  for (int g0 = 0; g0 < n;) {
    int g1 = g0;
    while (g1 < n && S.rec[g1].tid == S.rec[g0].tid) ++g1;
    const int na = S.seen[g0];
    if (na >= requiredNumHits || allActive) {
      SelGroup g;
      const int ns = sel_chain_group(S.rec + g0, g1 - g0, S.f + g0, S.p + g0, S.seen + g0, S.ends + g0, S.starts + g0, (int)readLen, g, P + np);
      if (ns > 0) { g.set_off(np); np += ns; G[ng++] = g; }
    }
    (void)mate;
    g0 = g1;
  }
  S.ngrp[s] = ng; S.npos[s] = np;
}

// mergeOrientationUnique (HitManager.cpp:834-881) of the two strands' groups, written as the read's list:
//   header, own position, nP positions of the surviving orientation, nO of the other.  Returns the word count
//   (-1: does not fit).  Lane-0 code.
template <typename SS>
QM_DEV int sel_emit(SS& S) {
  int i = 0, j = 0, o = 0;
  const int nf = S.ngrp[0], nr = S.ngrp[1];
  const SelGroup* GF = S.grpp(0); const SelGroup* GR = S.grpp(1);
  while (i < nf || j < nr) {
    const bool haveF = i < nf, haveR = j < nr;
    const u32 tf = haveF ? GF[i].tid : 0xffffffffu, tr = haveR ? GR[j].tid : 0xffffffffu;
    const SelGroup* pg; const SelGroup* og = nullptr; bool prc; int ps, os = 0;
    if (haveF && (!haveR || tf < tr)) { pg = &GF[i]; prc = false; ps = 0; ++i; }
    else if (haveR && (!haveF || tr < tf)) { pg = &GR[j]; prc = true; ps = 1; ++j; }
    else {
      // same transcript on both strands: the better chain score survives, forward on ties (stable inplace_merge)
      const bool rcFirst = GR[j].score > GF[i].score;
      if (rcFirst) { pg = &GR[j]; og = &GF[i]; prc = true; ps = 1; os = 0; }
      else { pg = &GF[i]; og = &GR[j]; prc = false; ps = 0; os = 1; }
      ++i; ++j;
    }
    const int nP = pg->npos, nO = og ? og->npos : 0;
    if (o + 2 + nP + nO > S.outcap()) return -1;
    S.out[o++] = sel_header(pg->tid, prc, pg->cs(), nP);
    S.out[o++] = (u64)(u32)pg->ppos | ((u64)(u32)nO << 32);
    for (int t = 0; t < nP; ++t) S.out[o++] = (u64)(u32)S.pos[ps][pg->off() + t];
    for (int t = 0; t < nO; ++t) S.out[o++] = (u64)(u32)S.pos[os][og->off() + t];
  }
  return o;
}

// The two stable sorts of sel_strand as one rank sort by all lanes (n <= 64 * QM_SEL_CHUNKS, lane l owns the records
// l, l + 64, ...): order (tid, reference end, query end, input order) for several intervals, (tid, hit position, input
// order) for one -- what lane 0's merge sorts produce.
#define QM_SEL_CHUNKS 4      // the most chunks any scratch edition handles lane-parallel (SS::NCH)
template <typename SS>
QM_DEV void sel_wave_sort(SS& S, int n, int m) {
  u64* K = (u64*)S.tmp;                              // keys where every lane can read them (2 words per record)
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) {
    if (64 * c >= n) continue;
    QM_LANES(l) {
      const int i = 64 * c + l;
      if (i < n) {
        const SelRec r = S.rec[i];
        if (m == 1) { K[2 * i] = ((u64)r.tid << 32) | (u64)(((u32)(r.pos - r.qpos)) ^ 0x80000000u); K[2 * i + 1] = (u64)i; }
        else { K[2 * i] = ((u64)r.tid << 32) | (u64)(u32)(r.pos + r.len); K[2 * i + 1] = ((u64)(u32)(r.qpos + r.len) << 16) | (u64)i; }
      }
    }
  }
  wave_fence();
  LV<int> rank[SS::NCH]; LV<SelRec> mine[SS::NCH];
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) {
    if (64 * c >= n) continue;
    QM_LANES(l) {
      const int i = 64 * c + l;
      int rk = 0;
      if (i < n) {
        const u64 k1 = K[2 * i], k2 = K[2 * i + 1];
        for (int j = 0; j < n; ++j) { const u64 a = K[2 * j], b = K[2 * j + 1]; rk += (a < k1 || (a == k1 && b < k2)) ? 1 : 0; }
        mine[c][l] = S.rec[i];
      }
      rank[c][l] = rk;
    }
  }
  wave_fence();
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) { if (64 * c < n) { QM_LANES(l) { if (64 * c + l < n) S.rec[rank[c][l]] = mine[c][l]; } } }
  wave_fence();
}

// sel_strand for several intervals when the n <= 64 * SS::NCH records are already in chain order (sel_wave_sort): the
// lane that holds a transcript's first record counts its intervals and chains its hits, all transcripts of a 64-record
// chunk at once; the groups are then packed in transcript order by prefix sums over the lanes and chunks.
template <typename SS>
QM_DEV void sel_strand_wave(SS& S, int s, int n, int m, u32 readLen, float consensusFraction) {
#pragma clang fp contract(off)
  SelGroup* G = S.grpp(s); int* P = S.pos[s];
  const float requiredFrac = (float)m * consensusFraction;
  int requiredNumHits = m, maxSlack = 0;
  if (consensusFraction < 1.0) {
    int fl = (int)requiredFrac;
    requiredNumHits = fl > 1 ? fl : 1;
    maxSlack = m - requiredNumHits;
  }
  u64 hm[SS::NCH];
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) hm[c] = 0;
  LV<bool> head[SS::NCH];
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) {
    QM_LANES(l) { const int i = 64 * c + l; head[c][l] = i < n && (i == 0 || S.rec[i].tid != S.rec[i - 1].tid); }
    hm[c] = ballot(head[c]);
  }
  LV<int> g1v[SS::NCH]; LV<bool> req[SS::NCH];
  bool anyReq = false;
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) {
    if (64 * c >= n) continue;
    QM_LANES(l) {
      g1v[c][l] = 0; req[c][l] = false;
      if (head[c][l]) {
        const int i = 64 * c + l;
        // the group ends at the next head: above this lane in the same chunk, else the first head of a later chunk
        int g1 = n;
        u64 rest = l < 63 ? (hm[c] & ~lanemask_lt(l + 1)) : 0ULL;
        if (rest) g1 = 64 * c + ctz64(rest);
        else {
#pragma unroll
          for (int d = SS::NCH - 1; d > 0; --d) if (d > c && hm[d]) g1 = 64 * d + ctz64(hm[d]);   // the lowest such d wins
        }
        int nIv;
        if (m <= 64) { u64 mk = 0; for (int j = i; j < g1; ++j) mk |= 1ULL << S.rec[j].iv; nIv = popc64(mk); }   // (m is the same for every lane)
        else { SelIvSet mk; for (int j = i; j < g1; ++j) mk.add(S.rec[j].iv); nIv = mk.count(); }
        g1v[c][l] = g1;
        req[c][l] = nIv >= requiredNumHits;
      }
    }
    anyReq = anyReq || ballot(req[c]) != 0;
  }
  const bool allActive = maxSlack > 0 && !anyReq;           // HitManager.cpp:682-686
  LV<int> nsv[SS::NCH]; LV<SelGroup> gv[SS::NCH]; LV<bool> em[SS::NCH];
  u64 emm[SS::NCH];
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) emm[c] = 0;
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) {
    QM_LANES(l) {
      nsv[c][l] = 0; em[c][l] = false;
      if (64 * c < n && head[c][l] && (req[c][l] || allActive)) {
        const int i = 64 * c + l;
        nsv[c][l] = sel_chain_group(S.rec + i, g1v[c][l] - i, S.f + i, S.p + i, S.seen + i, S.ends + i, S.starts + i, (int)readLen, gv[c][l], S.ends + i);
        em[c][l] = nsv[c][l] > 0;
      }
    }
    emm[c] = ballot(em[c]);
    QM_LANES(l) { if (em[c][l]) S.starts[64 * c + l] = nsv[c][l]; }
  }
  wave_fence();
  int ngAll = 0;
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) ngAll += popc64(emm[c]);
  if (ngAll > S.gcap()) { QM_LANES(l) { if (l == 0) S.ngrp[s] = -1; } return; }   // more transcripts than this scratch holds
#pragma unroll
  for (int c = 0; c < SS::NCH; ++c) {
    QM_LANES(l) {
      if (em[c][l]) {
        int mine = 0, ord = 0;                             // positions / groups emitted by the heads before this one
#pragma unroll
        for (int d = 0; d < SS::NCH; ++d) {
          if (d > c) continue;
          u64 r = d < c ? emm[d] : (emm[d] & lanemask_lt(l));
          ord += popc64(r);
          for (; r; r &= r - 1) mine += S.starts[64 * d + ctz64(r)];
        }
        SelGroup g = gv[c][l]; g.set_off(mine);
        G[ord] = g;
        for (int t = 0; t < nsv[c][l]; ++t) P[mine + t] = S.ends[64 * c + l + t];
      }
    }
  }
  QM_LANES(l) {
    if (l == 0) {
      int tot = 0;
#pragma unroll
      for (int d = 0; d < SS::NCH; ++d) for (u64 r = emm[d]; r; r &= r - 1) tot += S.starts[64 * d + ctz64(r)];
      S.ngrp[s] = ngAll; S.npos[s] = tot;
    }
  }
}

// Stage A, -s variant of hits_to_mappings on scratch S: returns the number of list words in S.out, -1 when S is too small.
template <typename SS>
QM_DEV int sel_h2m_on(const DevIndex& ix, const ReadBatch& B, const IntervalList& fwdInts, const IntervalList& rcInts,
                      u32 readLen, int mate, SS& S) {
  for (int s = 0; s < 2; ++s) {
    const IntervalList& L = s == 0 ? fwdInts : rcInts;
    int n = 0;
    // at most 64 intervals (every read of the short-read kernels): lane j takes interval j's width, the widths go to S.p (free until the
    // chaining) and every lane adds them up -- the walk over the list is not lane 0's with a broadcast per field
    const bool few = L.n <= 64;
    if (few) {
      QM_LANES(l) {
        if (l < L.n) {
          u32 b, e;
          if (l < QM_ICAP) { b = L.lds[l].b; e = L.lds[l].e; } else { b = L.ovf[l - QM_ICAP].b; e = L.ovf[l - QM_ICAP].e; }
          S.p[l] = (int)(e - b);
        }
      }
      wave_fence();
      LV<int> tot;
      QM_LANES(l) { int a = 0; for (int j = 0; j < L.n; ++j) a += S.p[j]; tot[l] = a; }
      n = read_lane(tot, 0);
    } else {
      for (int ii = 0; ii < L.n; ++ii) { u32 lb, ub, ln, qp; L.get(ii, lb, ub, ln, qp); n += (int)(ub - lb); }
    }
    if (n > S.cap()) return -1;
    if (n <= 64 * SS::NCH && few) {
      // the usual case: one lane per suffix over all the intervals at once -- one trip to sainfo per 64 suffixes of the strand;
      // a lane finds the interval of its suffix by running down the widths
      for (int base = 0; base < n; base += 64) {
        QM_LANES(l) {
          const int i = base + l;
          if (i < n) {
            int ii = 0, start = 0, acc = 0;
            for (int j = 0; j < L.n; ++j) { acc += S.p[j]; if (i >= acc) { ii = j + 1; start = acc; } }
            IntRec q;
            if (ii < QM_ICAP) { q.b = L.lds[ii].b; q.e = L.lds[ii].e; q.len = L.lds[ii].len; q.q = L.lds[ii].q; } else q = L.ovf[ii - QM_ICAP];
            SaInfo e = ix.sainfo[q.b + (u32)(i - start)];
            SelRec r; r.tid = e.tid; r.pos = (u32)e.pos; r.qpos = q.q; r.len = q.len; r.iv = (u32)ii;
            S.rec[i] = r;
          }
        }
      }
    } else if (n <= 64 * SS::NCH) {
      for (int base = 0; base < n; base += 64) {
        LV<u32> sa, qv, lv, iv; LV<bool> hs;
        QM_LANES(l) { sa[l] = 0; hs[l] = false; qv[l] = 0; lv[l] = 0; iv[l] = 0; }
        int acc = 0;
        for (int ii = 0; ii < L.n && acc < base + 64; ++ii) {
          u32 lb, ub, ln, qp; L.get(ii, lb, ub, ln, qp);
          const int span = (int)(ub - lb);
          QM_LANES(l) { const int i = base + l; if (i >= acc && i < acc + span) { sa[l] = lb + (u32)(i - acc); hs[l] = true; qv[l] = qp; lv[l] = ln; iv[l] = (u32)ii; } }
          acc += span;
        }
        QM_LANES(l) {
          if (hs[l]) {
            SaInfo e = ix.sainfo[sa[l]];
            SelRec r; r.tid = e.tid; r.pos = (u32)e.pos; r.qpos = qv[l]; r.len = lv[l]; r.iv = iv[l];
            S.rec[base + l] = r;
          }
        }
      }
    } else {
      n = 0;
      for (int ii = 0; ii < L.n; ++ii) {
        u32 lb, ub, ln, qp; L.get(ii, lb, ub, ln, qp);
        const int cnt = (int)(ub - lb);
        for (int base = 0; base < cnt; base += 64) {
          QM_LANES(l) {
            int i = base + l;
            if (i < cnt) {
              SaInfo e = ix.sainfo[lb + i];
              SelRec r; r.tid = e.tid; r.pos = (u32)e.pos; r.qpos = qp; r.len = ln; r.iv = (u32)ii;
              S.rec[n + i] = r;
            }
          }
        }
        n += cnt;
      }
    }
    wave_fence();
    QM_T(1);
    const bool presorted = n <= 64 * SS::NCH && 2 * n * (int)sizeof(u64) <= S.tmp_bytes();
    if (presorted && L.n > 0) sel_wave_sort(S, n, L.n);
    QM_T(2);
    if (presorted && L.n > 1) sel_strand_wave(S, s, n, L.n, readLen, B.consensus_fraction);
    else QM_LANES(l) { if (l == 0) { if (L.n > 0) sel_strand(S, s, n, L.n, readLen, mate, B.consensus_fraction, presorted); else { S.ngrp[s] = 0; S.npos[s] = 0; } } }
    wave_fence();
    QM_T(3);
    LV<int> ngv; QM_LANES(l) { ngv[l] = S.ngrp[s]; }
    if (read_lane(ngv, 0) < 0) return -1;
  }
  LV<int> nw;
  QM_LANES(l) { nw[l] = 0; if (l == 0) nw[l] = sel_emit(S); }
  wave_fence();
  QM_T(4);
  return read_lane(nw, 0);
}

// LDS scratch first (small reads: nearly all), the wave's global scratch otherwise.  A read that overflows that as well
// (repeats, low-complexity reads: a strand may bring several SA intervals of up to maxInterval suffixes each) is put on the
// slow queue -- returns -2, the caller marks the read -- and is redone by a second, small launch of the same kernel whose
// waves own scratch sized from the largest queued read (ReadBatch::dyn); the reference maps such reads like any other.
QM_DEV int sel_hits_to_mappings(const DevIndex& ix, const ReadBatch& B, const IntervalList& fwdInts, const IntervalList& rcInts,
                                u32 readLen, int mate, SelScratch& G, SelScratchLds* L, u64* ldsOut, const u64*& src, SelScratchDyn* dyn) {
  int n = -1;
  if (!dyn) {
    if (L) { L->out = (QM_LDS(u64)*)ldsOut; n = sel_h2m_on(ix, B, fwdInts, rcInts, readLen, mate, *L); src = ldsOut; }
    if (n < 0) { n = sel_h2m_on(ix, B, fwdInts, rcInts, readLen, mate, static_cast<SelScratchT<QM_SEL_CAP, QM_CHUNK>&>(G)); src = G.out; }
    if (n >= 0) return n;
    int need = 0;                                           // suffixes of the larger strand: sizes the slow pass's scratch
    for (int s = 0; s < 2; ++s) {
      const IntervalList& I = s == 0 ? fwdInts : rcInts;
      int t = 0;
      for (int ii = 0; ii < I.n; ++ii) { u32 lb, ub, ln, qp; I.get(ii, lb, ub, ln, qp); t += (int)(ub - lb); }
      need = t > need ? t : need;
    }
    QM_LANES(l) { if (l == 0) { atomic_add_u64(B.cursor + QM_SC_SLOWCNT, 1ULL); atomic_max_u64(B.cursor + QM_SC_SLOWMAX, (u64)need); } }
    return -2;
  }
  SelScratchDyn D = *dyn;
  n = sel_h2m_on(ix, B, fwdInts, rcInts, readLen, mate, D); src = D.out;
  if (n < 0) { QM_LANES(l) { if (l == 0) *B.status |= 8; } n = 0; }   // cannot happen: the scratch was sized for this read
  return n;
}

// ------------------------------------------------------------------ stages B + C: plan (per unit) -> ksw2 (four per wavefront) -> finish (per unit)
struct SelBatch {                    // launch arguments of the -s kernels (on top of PairBatch)
  const unsigned char* seq1; const unsigned char* seq2;
  const unsigned char* text; const u32* txp_off; const int* txp_len;
  qm_hit* tmp; const long long* toff;          // per-unit slots for jointHits before the filter
  int* tsc;                                    // score per slot-side (left / right)
  int* tref;                                   // per slot-side: -1 score is final / pending in tsc, <= -2 copy of unit-local entry -(ref)-2
  struct SelSide* sides; u64* nsides;          // the chunk's alignment questions (everything but PERFECT chains)
  struct SelTask* tasks; u64* torder; u64* ntasks;   // ksw2 work: tasks[x] of question x, walked in the order of torder[0 .. *ntasks) (null: tasks[0 ..) as they are)
  u64* torder2; u64* ntasks2;                   // ... and the alignments the strip kernel takes (sel_tasks_strip), torder2[0 .. *ntasks2); null: none
  long long u0, u1;                            // the units [u0, u1) this launch of plan / finish covers (the batch goes through
                                               // plan -> align -> finish in chunks: the plan of chunk i+1 runs under the align of chunk i)
  int match, mismatch, gap_open, gap_extend, bandwidth, hard_filter, policy;
  int no_diag;                                 // profiling (QM_SEL_NO_DIAG): 1 queue every REGULAR alignment, 2 only rule 1 of sel_side_score (no two-mismatch answers), 3 rules 1 and 2 but no strip alignments
  int long_reads;                              // the batch holds reads beyond QM_MAX_READ_LEN: the long editions of the alignment kernel
  void* ksw_rows;                              // long reads under a band beyond 97: the alignment blocks in device memory (KswRowT<QM_KSW_RING_GMEM, QM_KSW_MAXLEN_LONG>, four per wavefront)
  int short_len;                               // longest read of the batch that is not beyond QM_MAX_READ_LEN (0: unknown): reads of up to 128
                                               // characters run the alignment kernel with 192-byte images (twice the resident wavefronts)
  double min_score_fraction;
};

// ksw_extz2_sse41, score only, exact max, no z-drop (src/ksw2pp/ksw2_extz2_sse.c:18-304), as the reference calls it for
// every extension alignment (KSW2Aligner::operator(), EXTENSION).  The SSE kernel works on 16-byte vectors of int8
// differences and also computes the lanes of a vector that lie outside the band; those values are read back as neighbours
// when the band moves, so the out-of-band lanes, the byte wrap-around and the zero-initialised state of columns nobody
// has computed yet are all reproduced.
//
// Four alignments per wavefront: every row of 16 lanes is one __m128i of the SSE kernel and walks the 16-column vectors
// of its own alignment's band one after the other, like the original's inner loop.  Column state lives in a ring of RING
// slots in LDS (column t in slot t & (RING - 1), u|v<<8|x<<16|y<<24 in one word), reset when the 16-aligned window start
// moves.  A round touches the columns [st, max(en, smax)] with st = 16-aligned band start, en = 16-aligned band end,
// smax <= band end + 15: at most w + 31 columns, so RING >= w + 31 (64 slots for --dpBandwidth <= 33, 128 up to 97, 1024
// for anything else -- 1024 slots hold every column of the longest alignment, the window then never moves).
// The score phase of the original reads the reversed query and the target out of one zeroed block (sf = target + zeros up
// to tlen16, directly followed by qr = reversed query + zeros) and its 16-wide vectors run past both ends of the band; the
// two images hold exactly what those reads return:
//   QX[16 + i] = query[i] (0 <= i < qlen), zero before and after    -- the query character of cell (r, t) is QX[16 + r - t]
//   TX[t] = target[t] (t < tlen), 0 (t < tlen16), query[qlen - 1 - (t - tlen16)] beyond -- a column's target character
// Everything is per-lane (row-uniform) VALU work: no scalar control per alignment.
#define QM_KSW_MAXLEN (QM_MAX_READ_LEN + 32)                // read + 20 extra target characters, rounded up
#define QM_KSW_MAXLEN_LONG (QM_MAX_LONG_READ_LEN + 32)      // ... of a batch that holds reads beyond QM_MAX_READ_LEN (the long editions of the kernel)
template <int RING, int MAXLEN = QM_KSW_MAXLEN>
struct KswRowT {                                  // one alignment's LDS block (1232 bytes at RING = 64)
  unsigned char QX[MAXLEN + 40], TX[MAXLEN + 40];
  u32 ST[RING]; int HH[RING]; unsigned char SS[RING];
};
// --dpBandwidth <= 15 (the default): a round touches at most 32 columns, two per lane -- the register edition
// (sel_ksw_extz2_rows_reg, "32 slots"): no ring at all, the block holds the two images only
template <int MAXLEN>
struct KswRowT<32, MAXLEN> { unsigned char QX[MAXLEN + 40], TX[MAXLEN + 40]; };
inline constexpr int sel_ksw_ring_slots(int w) {   // host + device (constexpr)
  return (w >= 0 && w <= 15) ? 32 : ((w >= 0 && w <= 33) ? 64 : ((w >= 0 && w <= 97) ? 128 : 1024)); }
static_assert(QM_KSW_MAXLEN + 48 <= 1024, "the full-band ring must hold every column of the longest alignment");
// A batch that holds reads beyond QM_MAX_READ_LEN under a band beyond 97: the ring that holds every column of a 2048-base alignment
// has 4096 slots -- 41 KB per alignment, four alignments per wavefront: more than a CU's LDS -- so those blocks live in device
// memory (SelBatch::ksw_rows; qm_sel_align_gmem_kernel).  Slow, and rare: the reference takes any read under any band
// (src/RapMapSAMapper.cpp:1017, src/ksw2pp/ksw2_extz2_sse.c has no length bound), a batch must not fail over one such read.
#define QM_KSW_RING_GMEM 4096
static_assert(QM_KSW_MAXLEN_LONG + 48 <= QM_KSW_RING_GMEM, "the device-memory ring must hold every column of the longest alignment");
// qlenv / tlenv: the row's alignment (0: the row idles); blk[row]; the images must be in place.  wIn < 0: the band is the
// whole matrix (ksw2_extz2_sse.c:45).  Returns max(mqe, mte) per row.
// The register edition of the row kernel below, for bands of at most 15 (w + 1 <= 16 band cells, a round's columns
// [st, max(en, smax)] within 32 of the 16-aligned window start): lane c of a row OWNS the columns t0 = stv + c and t1 = t0 + 16
// for as long as the window stands (it advances by exactly 16 columns every ~32 rounds: t0's state <- t1's, t1 starts fresh), so
// the four difference bytes, the score byte and H of both columns live in registers -- no ring in LDS, no slot arithmetic, no
// stores; neighbours come over DPP.  Cell for cell the same arithmetic as sel_ksw_extz2_rows<RING> (see there for the SSE kernel's
// conventions it reproduces); the two are held against each other and against the reference's kernel by tests/test_ksw_variants.py.
// UNI: the rows that hold an alignment all have the lengths (uq, ut) -- four reads of one length well inside their transcripts,
// nearly every wavefront of a fixed-length run -- so the band's bounds, the window, the "moved" / diagonal / last-rounds tests are
// the same numbers in every lane: the compiler keeps them on the scalar unit (a third of the round's VALU instructions); rows
// without an alignment run along on whatever their images hold, nobody reads their score.
// SETS = 2 (UNI only): the wavefront carries two groups of four alignments of that one shape -- rows blk[0..3] and blk[4..7], two sets of
// state registers -- through the same rounds: the kernel is bound by the CU's scalar unit (one instruction per cycle for all four
// SIMDs), and the scalar work of a round is then shared by eight alignments instead of four.
struct KswSetState { LV<int> ST0, ST1, HB, SSP, TQ, QS, freshNext, mqe, mte; };
template <int MAXLEN, bool UNI, int SETS = 1>
QM_DEV void sel_ksw_extz2_rows_reg(const LV<int>* qlenvIn, const LV<int>* tlenvIn, int uq, int ut, KswRowT<32, MAXLEN>* blk, const signed char* mat, int q, int e, int wIn,
                                   LV<int>* score) {
  static_assert(UNI || SETS == 1, "several sets per wavefront share the band's geometry");
  LV<int> qlenv, tlenv;
  QM_LANES(l) { qlenv[l] = UNI ? uq : qlenvIn[0][l]; tlenv[l] = UNI ? ut : tlenvIn[0][l]; }
  typedef KswRowT<32, MAXLEN> Row;
  const int NEG = -0x40000000;
  const int m = 5;
  const int qe = q + e;
  int min_sc = mat[1];
  for (int t = 1; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
  if (-min_sc > 2 * (q + e)) { for (int s = 0; s < SETS; ++s) { QM_LANES(l) { score[s][l] = NEG; } } return; }
  const int sc_mch = (unsigned char)mat[0], sc_mis = (unsigned char)mat[1], sc_N = (unsigned char)mat[m * m - 1], m1 = m - 1;
  const int qe2 = (unsigned char)((q + e) * 2), max_sc_v = (unsigned char)(mat[0] + (q + e) * 2), qv = (unsigned char)q;
  const u32 QE2 = ((u32)qe2 << 8) | ((u32)qe2 << 24), MAXSC = ((u32)max_sc_v << 8) | ((u32)max_sc_v << 24), QV = ((u32)qv << 8) | ((u32)qv << 24);
  // the score of a pair of characters (nt4 codes 0 .. 4) as one byte lookup: y = (t ^ q) | ((t | q) & 4) is 0 for a match, 1 .. 3 for
  // a mismatch, 4 .. 7 when either is an N (ksw_gen_simple_mat's matrix: mat[0], mat[1], mat[24])
  const u32 LUTLO = (u32)sc_mch | ((u32)sc_mis << 8) | ((u32)sc_mis << 16) | ((u32)sc_mis << 24), LUTHI = (u32)sc_N * 0x01010101u;
  LV<int> lastSt; LV<bool> done;
  KswSetState W[SETS];                              // per set: ST0, ST1, HB, SSP -- the two owned columns: (u, v, x, y) bytes; their score bytes (byte 1 / 3: the packed
                                                    // recurrence's layout); H of the one that is a band cell (a band has at most 16 cells, so
                                                    // never both: a column enters the band as its top cell, whose H comes from the column to
                                                    // the left, and the lane's other column is 16 away)
                                                    // TQ, QS: target characters of the two owned columns (byte 0 / 1; they stand with the
                                                    // window) and their query characters of this round (a shift register along the row)
  QM_LANES(l) { lastSt[l] = -1; done[l] = qlenv[l] <= 0 || tlenv[l] <= 0; }
#pragma unroll
  for (int s = 0; s < SETS; ++s) {
    QM_LANES(l) {
      W[s].mqe[l] = NEG; W[s].mte[l] = NEG; W[s].freshNext[l] = 0;
      W[s].ST0[l] = 0; W[s].ST1[l] = 0; W[s].HB[l] = NEG; W[s].SSP[l] = 0; W[s].TQ[l] = 0; W[s].QS[l] = 0;
    }
  }
  const int FAR = 0x40000000;                       // "column offset" of an idle row: every range test below fails
  bool uDone = false; int uLastSt = -1;             // UNI: the wave-level tests of a round as plain numbers (no ballots)
  for (int r = 0; ; ++r) {
    bool uAct = false, uMoved = false, uDiag = false, uLate = false, uFirst = false;
    int uSt = 0, uEn = 0;
    if (UNI) {
      int st = 0, en = ut - 1;
      if (st < r - uq + 1) st = r - uq + 1;
      if (en > r) en = r;
      QM_SCALAR(en);
      if (st < (r - wIn + 1) >> 1) st = (r - wIn + 1) >> 1;
      if (en > (r + wIn) >> 1) en = (r + wIn) >> 1;
      uAct = !uDone && r < uq + ut - 1 && st <= en;
      if (!uAct) uDone = true;
      const int stw = st & ~15, rd = r - st, dc = rd - (stw - st);     // the diagonal's column is owned by lane dc (t0) or dc - 16 (t1) of the row
      uMoved = uAct && stw != uLastSt; uFirst = uLastSt < 0;
      uDiag = uAct && (((en + 16) >> 4) << 4) - 1 - st >= rd && dc >= 0 && dc < 32;
      uLate = uAct && (en == ut - 1 || rd == uq - 1);
      if (uAct) uLastSt = stw;
      uSt = st; uEn = en;
    }
    // everything below is in offsets from st0, the band's first column: d0 = owned column t0 - st0 (t1: d0 + 16), eb = en0 - st0,
    // ce = 16-aligned window end - st0, rd = r - st0 (the diagonal's column); stv is the window start
    LV<int> stv, d0v, ebv, cev, rdv, en0v; LV<bool> act, moved;
    QM_LANES(l) {
      int st, en; bool a, mv;
      if (UNI) { st = uSt; en = uEn; a = uAct; mv = uMoved; }
      else {
        const int qlen = qlenv[l], tlen = tlenv[l], w = wIn;
        st = 0; en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        a = !done[l] && r < qlen + tlen - 1 && st <= en;
        if (!a) done[l] = true;
        mv = a && (st & ~15) != lastSt[l];
      }
      act[l] = a;
      // (st >= 0; an active row has st <= en <= st + w, so the quotient of the original's (en - st) / 16 is 0 here; en >= 0)
      stv[l] = st & ~15;
      d0v[l] = a ? stv[l] + (l & 15) - st : FAR;
      ebv[l] = en - st > 0 ? en - st : 0; cev[l] = (((en + 16) >> 4) << 4) - 1 - st; rdv[l] = r - st; en0v[l] = en;
      moved[l] = mv;
    }
    if (UNI ? !uAct : !ballot(act)) break;
    // v, x of the column before the window: what its last owner (the row's last lane, as t0) left, in the round the window moves;
    // nothing (or the first column's boundary) otherwise
    LV<int> bpack[SETS];
#pragma unroll
    for (int s = 0; s < SETS; ++s) { QM_LANES(l) { bpack[s][l] = stv[l] > 0 ? 0 : ((r ? qv : 0) << 8); } }
    // the query character of cell (r, t) is QX[16 + r - t]: next round it belongs to the column to the right, so the characters
    // travel along the row (column t1 = t0 + 16 continues where the row's last lane leaves off) and only the row's first lane
    // reads a new one; the target characters stand with the window
    // (read one round ahead -- freshNext, asked for last round with that round's window: when the window has moved since, the
    // branch below reads the row's characters afresh anyway -- so that no round waits for its LDS read)
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
      LV<int> rq;
      row_rotate_up(W[s].QS, rq);
      QM_LANES(l) { W[s].QS[l] = (l & 15) == 0 ? (int)((u32)W[s].freshNext[l] | (((u32)rq[l] & 0xffu) << 8)) : rq[l]; }
      QM_LANES(l) {
        int qi = 17 + r - stv[l]; qi = act[l] ? (qi > MAXLEN + 39 ? MAXLEN + 39 : qi) : 0;
        W[s].freshNext[l] = blk[4 * s + (l >> 4)].QX[qi];
      }
    }
    if (UNI ? uMoved : ballot(moved) != 0) {              // rare: every ~32 rounds per row (and each row's first round)
#pragma unroll
      for (int s = 0; s < SETS; ++s) {
      LV<int> lastS;
      row_last(W[s].ST0, lastS);
      QM_LANES(l) {
        if (moved[l]) {
          const bool first = UNI ? uFirst : lastSt[l] < 0;
          if (!first && stv[l] > 0) bpack[s][l] = (int)((u32)lastS[l] & 0x00ffff00u);
          W[s].ST0[l] = first ? 0 : W[s].ST1[l]; W[s].SSP[l] = first ? 0 : (int)((u32)W[s].SSP[l] >> 16);
          W[s].ST1[l] = 0;
          Row& B = blk[4 * s + (l >> 4)];
          const int t0 = stv[l] + (l & 15), t1 = t0 + 16;
          int qi0 = 16 + r - t0; qi0 = qi0 < 0 ? 0 : (qi0 > MAXLEN + 39 ? MAXLEN + 39 : qi0);
          int qi1 = 16 + r - t1; qi1 = qi1 < 0 ? 0 : (qi1 > MAXLEN + 39 ? MAXLEN + 39 : qi1);
          const int ti0 = t0 > MAXLEN + 39 ? MAXLEN + 39 : t0, ti1 = t1 > MAXLEN + 39 ? MAXLEN + 39 : t1;
          W[s].QS[l] = (int)((u32)B.QX[qi0] | ((u32)B.QX[qi1] << 8));
          W[s].TQ[l] = (int)((u32)B.TX[ti0] | ((u32)B.TX[ti1] << 8));
        }
      }
      }
    }
    LV<bool> diag;
    QM_LANES(l) { diag[l] = cev[l] >= rdv[l] && (d0v[l] == rdv[l] || d0v[l] + 16 == rdv[l]); }
    if (UNI ? uDiag : ballot(diag) != 0) {                // only while the band still touches the diagonal (the first ~w rounds)
#pragma unroll
    for (int s = 0; s < SETS; ++s)
    QM_LANES(l) {                                       // the diagonal cell: y8[r] = 0, u8[r] = r ? q : 0
      if (diag[l]) {
        const u32 uval = (u32)(r ? qv : 0);
        if (d0v[l] == rdv[l]) W[s].ST0[l] = (int)(((u32)W[s].ST0[l] & 0x00ffff00u) | uval);
        else W[s].ST1[l] = (int)(((u32)W[s].ST1[l] & 0x00ffff00u) | uval);
      }
    }
    }
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
    // the scores of the 16 columns st0 .. st0 + 15 (the original's 16-wide score vectors start at st0)
    QM_LANES(l) {
      const u32 qs = (u32)W[s].QS[l], tq = (u32)W[s].TQ[l];
      const u32 y = (tq ^ qs) | ((tq | qs) & 0x0404u);                                    // bytes 0, 1: the two columns' lookup indices
      const u32 sn = perm8(LUTHI, LUTLO, perm8(y, 0x0c0c0c0cu, 0x05000400u));              // their scores in bytes 1, 3 (0 in bytes 0, 2)
      const bool inScore0 = (u32)d0v[l] <= 15u, inScore1 = (u32)(d0v[l] + 16) <= 15u;      // (never both)
      const u32 keep = inScore0 ? 0xffff0000u : (inScore1 ? 0x0000ffffu : 0xffffffffu);
      W[s].SSP[l] = (int)(((u32)W[s].SSP[l] & keep) | (sn & ~keep));
    }
    // the difference recurrence on the two owned columns (13 packed 16-bit instructions, see sel_ksw_extz2_rows)
    LV<int> r0, r1, carry;
    row_rotate_up(W[s].ST0, r0); row_rotate_up(W[s].ST1, r1); row_rotate_up(bpack[s], carry);
    QM_LANES(l) {
      const bool inCore0 = d0v[l] <= cev[l], inCore1 = d0v[l] + 16 <= cev[l];
      const u32 nb0 = (l & 15) == 0 ? (u32)carry[l] : (u32)r0[l], nb1 = (l & 15) == 0 ? (u32)r0[l] : (u32)r1[l];
      const u32 o0 = (u32)W[s].ST0[l], o1 = (u32)W[s].ST1[l];
      const u32 U = perm8(o1, o0, 0x040c000cu), Y = perm8(o1, o0, 0x070c030cu);
      const u32 V1 = perm8(nb1, nb0, 0x050c010cu), X1 = perm8(nb1, nb0, 0x060c020cu);
      const u32 S = (u32)W[s].SSP[l];
      u32 Z = pk_add(S, QE2), A = pk_add(X1, V1), Bq = pk_add(Y, U);
      Z = pk_max_i(Z, A);
      Z = pk_max_u(Z, Bq);
      Z = pk_min_u(Z, MAXSC);
      const u32 UN = pk_sub(Z, V1), VN = pk_sub(Z, U);
      Z = pk_sub(Z, QV);
      A = pk_sub(A, Z); Bq = pk_sub(Bq, Z);
      const u32 XN = pk_max_i(A, 0u), YN = pk_max_i(Bq, 0u);
      const u32 uv = perm8(VN, UN, 0x07030501u), xy = perm8(YN, XN, 0x07030501u);
      if (inCore0) W[s].ST0[l] = (int)perm8(xy, uv, 0x05040100u);
      if (inCore1) W[s].ST1[l] = (int)perm8(xy, uv, 0x07060302u);
    }
    // H (exact max) on the band cells st0..en0: at most 16 of them, so at most one of a lane's two columns
    LV<int> rh, hnv, cellv;
    row_rotate_up(W[s].HB, rh);                                // H of the column to the left (the top cell's neighbour), before this round's updates
    QM_LANES(l) {
      const bool in0 = (u32)d0v[l] <= (u32)ebv[l], in1 = (u32)(d0v[l] + 16) <= (u32)ebv[l];
      const int dsel = in1 ? d0v[l] + 16 : d0v[l], en0 = en0v[l];
      const u32 pk = (u32)(in1 ? W[s].ST1[l] : W[s].ST0[l]);
      const int hOld = W[s].HB[l], hl = rh[l];
      const int un = (int)(pk & 0xff), vn = (int)((pk >> 8) & 0xff);
      const int hOwn = hOld + vn - qe;
      const int hTop = en0 > 0 ? (hl + un - qe) : hOwn;
      const bool top = dsel == ebv[l];
      const int hn = r > 0 ? (top ? hTop : hOwn) : (vn - qe - qe);
      if (in0 || in1) W[s].HB[l] = hn;
      hnv[l] = hn; cellv[l] = (in0 || in1) ? dsel : -1;
    }
    // the two maxima live on the last target column and the last query row: only a row's last rounds have such a cell
    LV<bool> late;
    QM_LANES(l) { late[l] = d0v[l] != FAR && (en0v[l] == tlenv[l] - 1 || rdv[l] == qlenv[l] - 1); }
    if (UNI ? uLate : ballot(late) != 0)
    QM_LANES(l) {
      if (late[l] && cellv[l] >= 0) {
        const int hn = hnv[l];
        if (cellv[l] == ebv[l] && en0v[l] == tlenv[l] - 1 && hn > W[s].mte[l]) W[s].mte[l] = hn;
        if (cellv[l] == 0 && rdv[l] == qlenv[l] - 1 && hn > W[s].mqe[l]) W[s].mqe[l] = hn;
      }
    }
    }
    QM_LANES(l) { if (act[l]) lastSt[l] = stv[l]; }
  }
#pragma unroll
  for (int s = 0; s < SETS; ++s) {
    LV<int> neg;
    QM_LANES(l) { const int sc = W[s].mqe[l] > W[s].mte[l] ? W[s].mqe[l] : W[s].mte[l]; neg[l] = -sc; }
    group_min(neg, 16);
    QM_LANES(l) { score[s][l] = (qlenvIn[s][l] <= 0 || tlenvIn[s][l] <= 0) ? NEG : -neg[l]; }
  }
}

template <int RING, int MAXLEN = QM_KSW_MAXLEN>
QM_DEV void sel_ksw_extz2_rows(const LV<int>& qlenv, const LV<int>& tlenv, KswRowT<RING, MAXLEN>* blk, const signed char* mat, int q, int e, int wIn,
                               LV<int>& score) {
  if constexpr (RING == 32) {
    // do the rows that hold an alignment share their lengths?
    LV<int> key; LV<bool> odd;
    QM_LANES(l) { key[l] = (qlenv[l] > 0 && tlenv[l] > 0) ? ((qlenv[l] << 16) | tlenv[l]) : -1; }
    const int kmax = wave_max(key);
    QM_LANES(l) { odd[l] = key[l] >= 0 && key[l] != kmax; }
    if (kmax >= 0 && !ballot(odd)) sel_ksw_extz2_rows_reg<MAXLEN, true>(&qlenv, &tlenv, kmax >> 16, kmax & 0xffff, blk, mat, q, e, wIn, &score);
    else sel_ksw_extz2_rows_reg<MAXLEN, false>(&qlenv, &tlenv, 0, 0, blk, mat, q, e, wIn, &score);
    return;
  }
  else {
  typedef KswRowT<RING, MAXLEN> Row;
  constexpr int RM = RING - 1;
  constexpr int NV = RING / 16 + 1;                 // 16-column vectors a round may touch
  const int NEG = -0x40000000;
  const int m = 5;
  const int qe = q + e;
  int min_sc = mat[1];
  for (int t = 1; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
  if (-min_sc > 2 * (q + e)) { QM_LANES(l) { score[l] = NEG; } return; }
  const int sc_mch = (unsigned char)mat[0], sc_mis = (unsigned char)mat[1], sc_N = (unsigned char)mat[m * m - 1], m1 = m - 1;
  const int qe2 = (unsigned char)((q + e) * 2), max_sc_v = (unsigned char)(mat[0] + (q + e) * 2), qv = (unsigned char)q;
  LV<int> lastSt, lastEn, hb, mqe, mte, wv; LV<bool> done;
  QM_LANES(l) {
    lastSt[l] = -1; lastEn[l] = -1; hb[l] = NEG; mqe[l] = NEG; mte[l] = NEG; done[l] = qlenv[l] <= 0 || tlenv[l] <= 0;
    wv[l] = wIn < 0 ? (tlenv[l] > qlenv[l] ? tlenv[l] : qlenv[l]) : wIn;
  }
  for (int r = 0; ; ++r) {
    LV<int> st0v, en0v, stv, env, smaxv; LV<bool> act;
    QM_LANES(l) {
      const int qlen = qlenv[l], tlen = tlenv[l], w = wv[l];
      int st = 0, en = tlen - 1;
      if (st < r - qlen + 1) st = r - qlen + 1;
      if (en > r) en = r;
      if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
      if (en > (r + w) >> 1) en = (r + w) >> 1;
      const bool a = !done[l] && r < qlen + tlen - 1 && st <= en;
      if (!a) done[l] = true;
      act[l] = a;
      st0v[l] = st; en0v[l] = en; stv[l] = st / 16 * 16; env[l] = (en + 16) / 16 * 16 - 1;
      smaxv[l] = st + ((en - st) / 16) * 16 + 15;
    }
    if (!ballot(act)) break;
    // boundary of the first vector (x / v of column st - 1 as the last round left them), H of column st - 1 when the window moves
    LV<int> bpack, hbNew; LV<bool> moved;
    QM_LANES(l) {
      Row& B = blk[l >> 4];
      const int st = stv[l];
      const int left = (int)(B.ST[(st - 1) & RM] & 0x00ffff00u), hleft = B.HH[(st - 1) & RM];   // v, x and H of column st - 1
      bpack[l] = st > 0 ? ((st - 1 >= lastSt[l] && st - 1 <= lastEn[l]) ? left : 0) : ((r ? qv : 0) << 8);
      moved[l] = act[l] && st != lastSt[l];
      hbNew[l] = (st > 0 && lastSt[l] >= 0) ? hleft : NEG;
    }
    wave_fence();
    if (ballot(moved))                                    // rare: a row's window advances every ~32 rounds
    QM_LANES(l) {
      if (moved[l]) {
        Row& B = blk[l >> 4];
        hb[l] = hbNew[l];
        for (int k = 0; k < RING / 16; ++k) {
          const int slot = (l & 15) + 16 * k;
          const int t = stv[l] + ((slot - stv[l]) & RM);
          const int told = lastSt[l] < 0 ? -1 : lastSt[l] + ((slot - lastSt[l]) & RM);
          if (told != t) { B.ST[slot] = 0; B.HH[slot] = NEG; B.SS[slot] = 0; }
        }
      }
    }
    wave_fence();
    LV<bool> diag;
    QM_LANES(l) { diag[l] = act[l] && env[l] >= r && (l & 15) == (r & 15); }
    if (ballot(diag))                                     // only while the band still touches the diagonal (the first ~w rounds)
    QM_LANES(l) {                                       // the diagonal cell: y8[r] = 0, u8[r] = r ? q : 0
      if (diag[l]) {
        Row& B = blk[l >> 4];
        B.ST[r & RM] = (B.ST[r & RM] & 0x00ffff00u) | (u32)(r ? qv : 0);
      }
    }
    wave_fence();
    // The difference recurrence, TWO 16-column vectors per pass: a lane owns column t0 = st + 32 it + c and column t0 + 16.
    // The four int8 quantities of its two cells travel as four registers of two 16-bit halves with the int8 in the HIGH byte
    // of each half: there a 16-bit add / subtract wraps exactly like the SSE kernel's 8-bit one, a signed / unsigned 16-bit
    // max / min orders like the 8-bit one, and the whole recurrence is 13 packed instructions for the two cells
    // (v_pk_add_u16, v_pk_sub_i16, v_pk_max_i16, v_pk_max_u16, v_pk_min_u16) between four byte shuffles in and four out.
    const u32 QE2 = ((u32)qe2 << 8) | ((u32)qe2 << 24), MAXSC = ((u32)max_sc_v << 8) | ((u32)max_sc_v << 24), QV = ((u32)qv << 8) | ((u32)qv << 24);
    LV<int> prevOld;
    QM_LANES(l) { prevOld[l] = bpack[l]; }
    for (int it = 0; ; ++it) {
      // every lane computes its cells whether they are in range or not (all loads stay inside the row's block: ring slots, and
      // image indices clamped into the images); only the stores are predicated -- far fewer exec-mask round trips
      LV<int> old0, old1, s0, s1; LV<bool> inCore0, inCore1, inScore0, inScore1;
      QM_LANES(l) {
        const int t0 = stv[l] + 32 * it + (l & 15), t1 = t0 + 16;
        Row& B = blk[l >> 4];
        inCore0[l] = act[l] && t0 <= env[l]; inCore1[l] = act[l] && t1 <= env[l];
        inScore0[l] = act[l] && t0 >= st0v[l] && t0 <= smaxv[l]; inScore1[l] = act[l] && t1 >= st0v[l] && t1 <= smaxv[l];
        old0[l] = (int)B.ST[t0 & RM]; old1[l] = (int)B.ST[t1 & RM]; s0[l] = B.SS[t0 & RM]; s1[l] = B.SS[t1 & RM];
      }
      QM_LANES(l) {
        Row& B = blk[l >> 4];
        const int t0 = stv[l] + 32 * it + (l & 15), t1 = t0 + 16;
        int qi0 = 16 + r - t0; qi0 = qi0 < 0 ? 0 : (qi0 > MAXLEN + 39 ? MAXLEN + 39 : qi0);
        int qi1 = 16 + r - t1; qi1 = qi1 < 0 ? 0 : (qi1 > MAXLEN + 39 ? MAXLEN + 39 : qi1);
        const int ti0 = t0 > MAXLEN + 39 ? MAXLEN + 39 : t0, ti1 = t1 > MAXLEN + 39 ? MAXLEN + 39 : t1;
        const int sv0 = B.QX[qi0], sq0 = B.TX[ti0], sv1 = B.QX[qi1], sq1 = B.TX[ti1];
        int tmp0 = (sq0 == sv0) ? sc_mch : sc_mis; tmp0 = (sq0 == m1 || sv0 == m1) ? sc_N : tmp0;
        int tmp1 = (sq1 == sv1) ? sc_mch : sc_mis; tmp1 = (sq1 == m1 || sv1 == m1) ? sc_N : tmp1;
        s0[l] = inScore0[l] ? tmp0 : s0[l]; s1[l] = inScore1[l] ? tmp1 : s1[l];
        if (inScore0[l]) B.SS[t0 & RM] = (unsigned char)tmp0;
        if (inScore1[l]) B.SS[t1 & RM] = (unsigned char)tmp1;
      }
      LV<int> r0, r1, carry;
      row_rotate_up(old0, r0); row_rotate_up(old1, r1); row_rotate_up(prevOld, carry);
      QM_LANES(l) {
        Row& B = blk[l >> 4];
        const int t0 = stv[l] + 32 * it + (l & 15), t1 = t0 + 16;
        // column t - 1 as the last round left it: the lower neighbour's word; for the first lane of a vector the last column
        // of the vector before it (the boundary column for the very first)
        const u32 nb0 = (l & 15) == 0 ? (u32)carry[l] : (u32)r0[l], nb1 = (l & 15) == 0 ? (u32)r0[l] : (u32)r1[l];
        const u32 o0 = (u32)old0[l], o1 = (u32)old1[l];
        const u32 U = perm8(o1, o0, 0x040c000cu), Y = perm8(o1, o0, 0x070c030cu);      // u, y of the own columns
        const u32 V1 = perm8(nb1, nb0, 0x050c010cu), X1 = perm8(nb1, nb0, 0x060c020cu); // v, x of the columns before them
        const u32 S = ((u32)s0[l] << 8) | ((u32)s1[l] << 24);
        u32 Z = pk_add(S, QE2), A = pk_add(X1, V1), Bq = pk_add(Y, U);
        Z = pk_max_i(Z, A);                                 // _mm_max_epi8
        Z = pk_max_u(Z, Bq);                                // _mm_max_epu8
        Z = pk_min_u(Z, MAXSC);                             // _mm_min_epu8
        const u32 UN = pk_sub(Z, V1), VN = pk_sub(Z, U);
        Z = pk_sub(Z, QV);
        A = pk_sub(A, Z); Bq = pk_sub(Bq, Z);
        const u32 XN = pk_max_i(A, 0u), YN = pk_max_i(Bq, 0u);
        const u32 uv = perm8(VN, UN, 0x07030501u), xy = perm8(YN, XN, 0x07030501u);    // (u0 v0 u1 v1), (x0 y0 x1 y1)
        if (inCore0[l]) B.ST[t0 & RM] = perm8(xy, uv, 0x05040100u);
        if (inCore1[l]) B.ST[t1 & RM] = perm8(xy, uv, 0x07060302u);
        prevOld[l] = inCore1[l] ? old1[l] : (inCore0[l] ? old0[l] : prevOld[l]);
      }
      wave_fence();
      // another pass only when some row's columns reach beyond these 32 (never for --dpBandwidth <= 15: st .. max(en, smax)
      // spans at most 31 columns there); decided from the bounds alone, before anything is loaded
      if (it + 1 >= (NV + 1) / 2) break;
      LV<bool> more;
      QM_LANES(l) { more[l] = act[l] && stv[l] + 32 * (it + 1) <= (env[l] > smaxv[l] ? env[l] : smaxv[l]); }
      if (!ballot(more)) break;
    }
    // H (exact max) over the band cells st0..en0 (up to w + 1 of them): all reads, then the writes
    LV<int> hLeft;                                      // H[en0 - 1] before this round's updates
    QM_LANES(l) {
      hLeft[l] = NEG;
      if (act[l] && en0v[l] > 0) hLeft[l] = en0v[l] > stv[l] ? blk[l >> 4].HH[(en0v[l] - 1) & RM] : hb[l];
    }
    wave_fence();
    for (int k = 0; ; ++k) {
      LV<int> hn; LV<bool> has;
      QM_LANES(l) {
        const int t = st0v[l] + (l & 15) + 16 * k;
        has[l] = act[l] && t <= en0v[l];
        Row& B = blk[l >> 4];
        const int en0 = en0v[l];
        const u32 pk = B.ST[t & RM];
        const int un = (int)(pk & 0xff), vn = (int)((pk >> 8) & 0xff);
        const int hOwn = B.HH[t & RM] + vn - qe;
        const int hTop = en0 > 0 ? (hLeft[l] + un - qe) : hOwn;
        hn[l] = r > 0 ? (t == en0 ? hTop : hOwn) : (vn - qe - qe);   // r == 0: the only cell is t == 0
      }
      wave_fence();
      QM_LANES(l) {
        if (has[l]) {
          Row& B = blk[l >> 4];
          const int t = st0v[l] + (l & 15) + 16 * k;
          B.HH[t & RM] = hn[l];
          if (t == en0v[l] && en0v[l] == tlenv[l] - 1 && hn[l] > mte[l]) mte[l] = hn[l];
          if (t == st0v[l] && r - st0v[l] == qlenv[l] - 1 && hn[l] > mqe[l]) mqe[l] = hn[l];
        }
      }
      wave_fence();
      if (k + 1 >= RING / 16) break;                      // band cells beyond these 16 (only with --dpBandwidth > 15)?
      LV<bool> moreH;
      QM_LANES(l) { moreH[l] = act[l] && st0v[l] + 16 * (k + 1) <= en0v[l]; }
      if (!ballot(moreH)) break;
    }
    QM_LANES(l) { if (act[l]) { lastSt[l] = stv[l]; lastEn[l] = env[l]; } }
  }
  LV<int> neg;
  QM_LANES(l) { const int s = mqe[l] > mte[l] ? mqe[l] : mte[l]; neg[l] = -s; }
  group_min(neg, 16);
  QM_LANES(l) { score[l] = (qlenv[l] <= 0 || tlenv[l] <= 0) ? NEG : -neg[l]; }
  }
}

QM_DEV unsigned char sel_nt4(unsigned char c) {               // seq_nt4_table_loc (KSW2Aligner.cpp:61-72), branch-free
  const unsigned char l = c | 0x20;
  const bool acgt = (l == 'a') | (l == 'c') | (l == 'g') | (l == 't');
  const unsigned x = (c >> 1) & 3;
  return acgt ? (unsigned char)(x ^ (x >> 1)) : (c < 4 ? c : (unsigned char)4);   // bytes 0..3 map to themselves in that table
}

struct SelTask { const unsigned char* rd; const unsigned char* tx; int gslot, rl, roff, rlen, tlen1, fwd; };   // one ksw2 extension alignment: the read, its
                                     // length, where the target starts -- resolved by the plan, so that the alignment kernel's first load is its only descriptor load

// the read as the alignment sees it: forward, or reverseRead() of it (src/RapMapUtils.cpp:107-128)
QM_DEV unsigned char sel_read_char(const unsigned char* r, int len, bool fwd, int i) { return fwd ? r[i] : rc_char(r[len - 1 - i]); }

// one list group as stage B sees it
struct SelG { u32 tid; bool rc; int cs, np, no, ppos; const u64* P; const u64* O; int words; };
QM_DEV SelG sel_group(const u64* X) {
  SelG g; const u64 h = X[0];
  g.tid = selh_tid(h); g.rc = selh_rc(h); g.cs = selh_cs(h); g.np = selh_np(h); g.no = (int)(X[1] >> 32);
  g.ppos = (int)(u32)X[1]; g.P = X + 2; g.O = X + 2 + g.np; g.words = 2 + g.np + g.no;
  return g;
}
// findBestHitFWRC (RapMapUtils.hpp:923-988) on two sorted position lists
QM_DEV bool sel_best_fwrc(const u64* F, int nf, const u64* R, int nr, int fwdReadLen, int& oF, int& oR, int& oGap) {
  if (nf == 0 || nr == 0) return false;
  int bestGap = 0x7fffffff, bf = 0, br = 0;
  for (int i = 0; i < nf; ++i) {
    const int p1 = (int)(u32)F[i];
    int lo = 0, hi = nr;                                   // lower_bound(R, p1)
    while (lo < hi) { int mid = (lo + hi) >> 1; if ((int)(u32)R[mid] < p1) lo = mid + 1; else hi = mid; }
    for (int t = 0; t < 2; ++t) {
      int c;
      if (lo == nr) { if (t) break; c = lo - 1; }
      else if (lo == 0) { if (t) break; c = lo; }
      else c = t == 0 ? lo : lo - 1;
      const int rp = (int)(u32)R[c];
      int gap = 0x7fffffff;
      if (rp >= p1) { int d = rp - (p1 + fwdReadLen); gap = d < 0 ? -d : d; }
      if (gap < bestGap) { bestGap = gap; bf = i; br = c; }
    }
  }
  if (bestGap == 0x7fffffff) return false;
  oF = (int)(u32)F[bf]; oR = (int)(u32)R[br]; oGap = bestGap;
  return true;
}

// mergeLeftRightHitsFuzzy + the -s driver of one pair (RapMapSAMapper.cpp:461-701) / one single read (:225-320).
// Writes the surviving hits to A.tmp + A.toff[u] (alignment scores in aln_score) and returns their number.
QM_DEV int sel_unit_merge(const PairBatch& P, const SelBatch& A, long long u, UnitCounters* uc) {
  const int maxHits = P.max_num_hits;
  qm_hit* T = A.tmp + A.toff[u];
  const int cap = (int)(A.toff[u + 1] - A.toff[u]);
  int n = 0;
  if (uc) uc->reads += 1;
  const u32 l1 = (u32)(P.off1[u + 1] - P.off1[u]);
  if (!P.paired) {
    const int nw = (int)(P.lcnt[u] & 0x7fffffffu);
    const u64* X = P.lists + P.loff[u];
    int g = 0;
    for (int i = 0; i < nw;) { SelG q = sel_group(X + i); i += q.words; ++g; }
    if (uc) uc->tot += (u64)g;                           // counted before the maxNumHits clear (:240-245)
    if (g <= maxHits) {
      for (int i = 0; i < nw && n < cap;) {
        SelG q = sel_group(X + i); i += q.words;
        qm_hit h; h.tid = q.tid; h.pos = q.ppos; h.mate_pos = 0; h.frag_len = 0; h.read_len = l1; h.mate_len = 0;
        h.fwd = q.rc ? 0 : 1; h.mate_is_fwd = 1; h.is_paired = 0; h.mate_status = 0; h.aln_score = q.cs;
        T[n++] = h;
      }
    }
    return n;
  }
  const u32 c0 = P.lcnt[2 * u], c1 = P.lcnt[2 * u + 1];
  const int wl = (int)(c0 & 0x7fffffffu), wr = (int)(c1 & 0x7fffffffu);
  const bool lh = (c0 >> 31) != 0, rh = (c1 >> 31) != 0;
  const u64* LL = P.lists + P.loff[2 * u];
  const u64* RR = P.lists + P.loff[2 * u + 1];
  const u32 l2 = (u32)(P.off2[u + 1] - P.off2[u]);
  auto orphan = [&](const SelG& q, u32 ln, int mateStatus) {
    qm_hit h; h.tid = q.tid; h.pos = q.ppos; h.mate_pos = 0; h.frag_len = 0; h.read_len = ln; h.mate_len = 0;
    h.fwd = q.rc ? 0 : 1; h.mate_is_fwd = 1; h.is_paired = 0; h.mate_status = (uint8_t)mateStatus;
    h.aln_score = mateStatus == 1 ? q.cs : (q.cs << 4);   // chain status parked: left in bits 0-3, right in bits 4-7
    return h;
  };
  bool tooMany = false, sameTxp = false;
  if (wl == 0 || wr == 0) {
    const int t = wl == 0 ? 1 : 0;
    const bool otherMatched = wl == 0 ? lh : rh;
    const u64* X = t == 0 ? LL : RR; const int nx = t == 0 ? wl : wr; const u32 ln = t == 0 ? l1 : l2;
    if (!otherMatched && nx > 0) {
      int g = 0;
      for (int i = 0; i < nx;) { SelG q = sel_group(X + i); i += q.words; if (n < cap) T[n++] = orphan(q, ln, t == 0 ? 1 : 2); ++g; }
      if (uc) { uc->se += (u64)g; uc->pe += (u64)g; }
    }
  } else {
    int i = 0, j = 0, nm = 0;
    while (i < wl && j < wr) {
      SelG a = sel_group(LL + i), b = sel_group(RR + j);
      if (a.tid < b.tid) { i += a.words; continue; }
      if (b.tid < a.tid) { j += b.words; continue; }
      sameTxp = true;
      // positions by strand (:991-996)
      const u64* lF = a.rc ? a.O : a.P; const int nlF = a.rc ? a.no : a.np;
      const u64* lR = a.rc ? a.P : a.O; const int nlR = a.rc ? a.np : a.no;
      const u64* rF = b.rc ? b.O : b.P; const int nrF = b.rc ? b.no : b.np;
      const u64* rR = b.rc ? b.P : b.O; const int nrR = b.rc ? b.np : b.no;
      int f1 = 0, q1 = 0, g1 = 0x7fffffff, f2 = 0, q2 = 0, g2 = 0x7fffffff;
      const bool fwrc = sel_best_fwrc(lF, nlF, rR, nrR, (int)l1, f1, q1, g1);
      const bool rcfw = sel_best_fwrc(rF, nrF, lR, nlR, (int)l2, f2, q2, g2);
      if (fwrc || rcfw) {
        int leftPos = -1, rightPos = -1, bestGap = 0x7fffffff; bool leftFwd = false;
        if (fwrc) { leftPos = f1; rightPos = q1; bestGap = g1; leftFwd = true; }
        if (rcfw && g2 < bestGap) { leftPos = q2; rightPos = f2; leftFwd = false; }
        const int s1 = leftPos > 0 ? leftPos : 0, s2 = rightPos > 0 ? rightPos : 0;
        const bool r1First = s1 < s2;
        const int fragStart = r1First ? s1 : s2;
        const int fragEnd = r1First ? (int)((u32)s2 + l2) : (int)((u32)s1 + l1);
        qm_hit h; h.tid = a.tid; h.pos = leftPos; h.mate_pos = rightPos; h.frag_len = (u32)(fragEnd - fragStart);
        h.read_len = l1; h.mate_len = l2; h.fwd = leftFwd ? 1 : 0; h.mate_is_fwd = leftFwd ? 0 : 1;
        h.is_paired = 1; h.mate_status = 3; h.aln_score = a.cs | (b.cs << 4);
        ++nm;
        if (nm > maxHits) { tooMany = true; break; }
        if (n < cap) T[n++] = h;
      }
      i += a.words; j += b.words;
    }
    if (tooMany) { n = 0; if (uc) uc->tooMany += 1; }
    if (uc && n > 0) uc->pe += (u64)n;
  }
  if (uc && P.too_many) P.too_many[u] = (tooMany ? 1 : 0) | (sameTxp ? 2 : 0);
  if (P.merge_only) return n;
  if (n > maxHits) n = 0;                                  // :534-536
  if (n > 0 && P.no_orphans && T[0].mate_status != 3) n = 0;   // :539-551
  return n;
}

// ------------------------------------------------------------------ plan / align / finish
// The per-pair work of the reference (merge, getAlnScore for every hit, gate, filter) is cut in three so that the ksw2
// alignments run four to a wavefront (sel_ksw_extz2_rows):  plan (per unit: merge, everything of getAlnScore but the ksw2
// scores; alignments that have to be run become tasks, alignment-cache hits become references to the entry that owns the
// score), align (a row of 16 lanes per task), finish (per unit: gate, filter, counters).

// Round 5: the plan is three flat steps instead of one divergent thread per unit (which spent 47 ms per 10 M pairs waiting
// on thirteen dependent rounds of scattered 8-byte loads per alignment question, with its lanes idling behind the unit that has
// the most hits):
//   sides   (per unit)  merge; a question per hit and mate that is not a PERFECT chain goes on the chunk's list of SelSide
//   score   (16 lanes per question on the device, 1 in the emulation)  everything of getAlnScore but ksw2: the geometry, the
//           alignment-cache key of the target window, the ungapped score, the alignment whose answer is known without running
//           it; a question that needs ksw2 leaves its SelTask in tasks[x]
//   dedupe  (per question)  the alignment cache: the first earlier question of the unit and mate with the same key owns the score
//           (a later one points at it); owners that need ksw2 put their index on the order list the alignment kernel walks
struct SelSide { long long u; long long g; u64 key; int kind; int nth; };   // kind: bit 0 keyed (multi-mapping unit), bit 1 needs ksw2; nth: how many questions of the unit precede this one

// which read, orientation, position and chain status the question of slot-side g asks about
struct SelQ { const unsigned char* read; int readLen; bool fwd; int pos; int cs; };
QM_DEV SelQ sel_side_question(const PairBatch& P, const SelBatch& A, long long u, const qm_hit& h, int side) {
  SelQ q;
  if (side == 0) {
    q.read = A.seq1 + P.off1[u]; q.readLen = (int)(P.off1[u + 1] - P.off1[u]); q.fwd = h.fwd != 0; q.pos = h.pos; q.cs = h.aln_score & 15;
  } else {
    q.read = A.seq2 + P.off2[u]; q.readLen = (int)(P.off2[u + 1] - P.off2[u]); q.cs = (h.aln_score >> 4) & 15;
    if (h.mate_status == 3) { q.pos = h.mate_pos; q.fwd = h.mate_is_fwd != 0; } else { q.pos = h.pos; q.fwd = h.fwd != 0; }
  }
  return q;
}
// the sides of hit h that exist: bit 0 left / single read, bit 1 right
QM_DEV int sel_hit_sides(const PairBatch& P, const qm_hit& h) { return !P.paired ? 1 : (h.mate_status == 3 ? 3 : (h.mate_status == 1 ? 1 : 2)); }

// merge + post filters of one unit into its temp slots (chain statuses parked in aln_score); returns the hit count
QM_DEV int sel_unit_merge(const PairBatch& P, const SelBatch& A, long long u, UnitCounters* uc);

// step 1a: merge, PERFECT chains answered (maxScore: SelectiveAlignmentUtils, getAlnScore's first branch); returns how many
// questions the unit puts on the list
QM_DEV int sel_unit_sides_count(const PairBatch& P, const SelBatch& A, long long u, UnitCounters* uc) {
  const int n = sel_unit_merge(P, A, u, uc);
  P.cnt[u] = (u32)n;                                       // hits before the score filter; sel_unit_finish overwrites it
  const qm_hit* T = A.tmp + A.toff[u];
  const long long gbase = 2 * A.toff[u];
  const int maxL = A.match * (int)(P.off1[u + 1] - P.off1[u]);
  const int maxR = P.paired ? A.match * (int)(P.off2[u + 1] - P.off2[u]) : 0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const qm_hit& h = T[i];
    const int sides = sel_hit_sides(P, h);
    for (int sd = 0; sd < 2; ++sd) {
      if (!((sides >> sd) & 1)) continue;
      const int cs = sd == 0 ? (h.aln_score & 15) : ((h.aln_score >> 4) & 15);
      if (cs == QM_CS_PERFECT) { A.tsc[gbase + 2 * i + sd] = sd == 0 ? maxL : maxR; A.tref[gbase + 2 * i + sd] = -1; }
      else ++m;
    }
  }
  return m;
}
// step 1b: the unit's questions at sides[base ..)
QM_DEV void sel_unit_sides_write(const PairBatch& P, const SelBatch& A, long long u, long long base) {
  const int n = (int)P.cnt[u];
  const qm_hit* T = A.tmp + A.toff[u];
  const long long gbase = 2 * A.toff[u];
  int nth = 0;
  for (int i = 0; i < n; ++i) {
    const qm_hit& h = T[i];
    const int sides = sel_hit_sides(P, h);
    for (int sd = 0; sd < 2; ++sd) {
      if (!((sides >> sd) & 1)) continue;
      const int cs = sd == 0 ? (h.aln_score & 15) : ((h.aln_score >> 4) & 15);
      if (cs == QM_CS_PERFECT) continue;
      SelSide S; S.u = u; S.g = gbase + 2 * i + sd; S.key = 0; S.kind = 0; S.nth = nth++;
      A.sides[base++] = S;
    }
  }
}

// bytes i0 .. i0+7 of p[0 .. len), zero beyond the end -- the last, partial word is the last eight bytes shifted down: no byte loop
QM_DEV u64 sel_word_at(const unsigned char* p, int i0, int len) {
  if (i0 + 8 <= len) return load_u64_unaligned(p + i0);
  if (i0 >= len) return 0;
  if (len >= 8) return load_u64_unaligned(p + len - 8) >> (8 * (i0 + 8 - len));
  u64 w = 0;
  for (int t = 0; t < 8 && i0 + t < len; ++t) w |= (u64)p[i0 + t] << (8 * t);
  return w;
}
// characters j0 .. j0+7 of the read as the alignment sees it (forward, or reverseRead() of it), zero beyond its end
QM_DEV u64 sel_read_word(const unsigned char* r, int len, bool fwd, int j0) {
  if (fwd) return sel_word_at(r, j0, len);
  if (j0 >= len) return 0;
  u64 v;                                                   // source bytes len-1-j0-7 .. len-1-j0, last one first
  const int lo = len - 1 - j0 - 7;
  if (lo >= 0) v = load_u64_unaligned(r + lo);
  else if (len >= 8) v = load_u64_unaligned(r) << (8 * -lo);
  else { v = 0; for (int t = 0; t < 8 && j0 + t < len; ++t) v |= (u64)r[len - 1 - j0 - t] << (8 * (7 - t)); }
  v = __builtin_bswap64(v);
  const int nv = len - j0 < 8 ? len - j0 : 8;              // the complement table makes 'N' of the zero bytes: cleared again
  const u64 w = (u64)rc_char4((u32)v) | ((u64)rc_char4((u32)(v >> 32)) << 32);
  return nv >= 8 ? w : (w & ((1ULL << (8 * nv)) - 1ULL));
}
struct SelRedOne { QM_DEV int add(int v) const { return v; } QM_DEV int min(int v) const { return v; } QM_DEV int max(int v) const { return v; } QM_DEV u64 bxor(u64 v) const { return v; } };   // a group of one lane (emulation)

// step 2: question x, by lane l of a group of G lanes (lane l takes the 8-character words l, l + G, ...); red sums / xors over the group
template <int G, typename Red>
QM_DEV void sel_side_score(const PairBatch& P, const SelBatch& A, long long x, int l, const Red& red) {
  const int LOWEST = (int)0x80000000;
  const long long u = A.sides[x].u, g = A.sides[x].g;
  const qm_hit h = A.tmp[g >> 1];
  const SelQ q = sel_side_question(P, A, u, h, (int)(g & 1));
  const bool multiMapping = P.cnt[u] > 1;
  const unsigned char* read = q.read; const int readLen = q.readLen; const bool fwd = q.fwd;
  const unsigned char* tseq = A.text + A.txp_off[h.tid];
  const int tlen = A.txp_len[h.tid];
  int pos = q.pos;
  int s = LOWEST, kind = 0; u64 key = 0;
  int roff = 0, rlen = readLen;
  const bool invalidStart = pos < 0;
  const bool invalidEnd = pos + rlen >= tlen;
  if (invalidStart) { roff = -pos; rlen += pos; pos = 0; }
  const bool dropped = (invalidStart || invalidEnd) && (A.policy == 1 || A.policy == 2);
  if (!dropped && pos < tlen) {
    const bool doUngapped = !invalidStart && q.cs == QM_CS_UNGAPPED;
    const u32 buf = doUngapped ? 0u : 20u;
    const u32 lnobuf = (u32)(tlen - pos), lbuf = (u32)(rlen + (int)buf);
    const bool useBuf = lbuf < lnobuf;
    const int tlen1 = (int)(lbuf < lnobuf ? lbuf : lnobuf);
    const unsigned char* tseq1 = tseq + pos;
    const int keyLen = useBuf ? tlen1 - (int)buf : tlen1;
    // Rule 1 -- the alignment whose answer is known without running it.  The extension alignment starts at (0, 0) and its score is
    // max(mqe, mte) (SelectiveAlignmentUtils.hpp:355-356, score only, no drop-off: RapMapSAMapper.cpp:198-204).  With the target
    // at least as long as the query, every path that opens a gap scores at most Smax - (q + e) -- Smax: every query character
    // at its best score, a match or 0 for an N -- whether it ends in the query's last row or in the target's last column (those
    // skip target characters); the gapless path from (0, 0) lies in every band and ends in the last row.  So when that path
    // loses no more than q + e against Smax -- one mismatch under the default scores -- its score IS the alignment's, and no task
    // is queued.  (Not with --dpBandwidth 0: that band's odd anti-diagonals are empty and the kernel stops at the second one.
    // Scores small enough for the 8-bit kernel to be exact; tests/test_ksw_variants.py, the parity tests and the bench's -s leg hold
    // every such score against the oracle's ksw2.)
    // Rule 2 -- two mismatches (loss 2M, M = a - b <= q + e, no N anywhere): besides the gapless path only a path with ONE gap run
    // and no mismatch at all can lose less -- a run and a mismatch lose q + e + M >= 2M, two runs 2(q + e) >= 2M.  Such a path
    // follows diagonal 0 up to the run at query position p <= m1 (the first mismatch) and diagonal +L (L target characters
    // skipped) or -L (L query characters skipped) behind it, so it exists exactly when that diagonal has no mismatch from p on:
    // the last mismatch of diagonal +L lies before m1, of diagonal -L before m1 + L.  L ranges over the run lengths that lose
    // less than 2M (q + L e, or q + L (e + a) with the query characters of the run: one and two skipped target characters and one
    // skipped query character under the default scores); the target has to reach past the longest of them, so that ending in
    // the target's last column is no option, and the band has to hold them.  The diagonals next to the main one cost a funnel
    // shift of the words already loaded.
    int a = (signed char)A.match, b = (signed char)A.mismatch;
    a = a < 0 ? -a : a; b = b > 0 ? -b : b;
    const int gq = (int)(signed char)A.gap_open, ge = (int)(signed char)A.gap_extend, qe = gq + ge, M = a - b;
    const bool diag = !doUngapped && A.no_diag != 1 && A.bandwidth != 0 && tlen1 >= rlen && rlen > 0 && a >= 1 && gq >= 0 && ge >= 1 && a - b + qe <= 96;
    int Ld = 0, Li = 0;
    while (Ld < 4 && gq + (Ld + 1) * ge < 2 * M) ++Ld;
    while (Li < 3 && gq + (Li + 1) * (ge + a) < 2 * M) ++Li;
    const bool rule2 = diag && A.no_diag != 2 && M <= qe && Ld <= 3 && Li <= 2 && (A.bandwidth < 0 || A.bandwidth >= 8) && tlen1 >= rlen + Ld + 1;
    const int alnLen = rlen < tlen1 ? rlen : tlen1;
    const int cmpLen = doUngapped ? alnLen : (diag ? rlen : 0);
    const int hashLen = multiMapping ? keyLen : 0;
    const int span = cmpLen > hashLen ? cmpLen : hashLen;
    int sc = 0, loss = 0, bad = 0, m1 = 0x7fffffff; u64 hx = 0;
    int hmD[3] = {-1, -1, -1}, hmI[2] = {-1, -1};            // last mismatch of diagonals +1 +2 +3, -1 -2
    for (int i0 = 8 * l; i0 < span; i0 += 8 * G) {
      const u64 tw = sel_word_at(tseq1, i0, tlen1);
      if (i0 < hashLen) {
        const u64 w = i0 + 8 <= hashLen ? tw : (tw & ((1ULL << (8 * (hashLen - i0))) - 1ULL));
        const u32 j = (u32)(i0 >> 3) + 1u;                               // position-salted, xor-combined: any grouping of the words gives the same key
        hx ^= hash_mix(w + ((u64)(j * 0x7F4A7C15u) | ((u64)(j * 0x9E3779B9u) << 32)));
      }
      if (i0 < cmpLen) {
        const int nt = cmpLen - i0 < 8 ? cmpLen - i0 : 8;
        const u64 rw = sel_read_word(read, readLen, fwd, roff + i0);
        // eight characters at a time: 0x80 per byte where ...
        const u32 vm0 = nt >= 4 ? 0x80808080u : (0x80808080u >> (8 * (4 - nt))), vm1 = nt >= 8 ? 0x80808080u : (nt > 4 ? (0x80808080u >> (8 * (8 - nt))) : 0u);
        const u32 t0 = (u32)tw, t1 = (u32)(tw >> 32), r0 = (u32)rw, r1 = (u32)(rw >> 32);
        if (doUngapped) {
          // ... the characters are the same or one of them is 'N' (SelectiveAlignmentUtils.hpp ungapped branch)
          const u32 m0 = (eq_bytes(t0 ^ r0, 0) | eq_bytes(t0, 'N') | eq_bytes(r0, 'N')) & vm0;
          const u32 m1b = (eq_bytes(t1 ^ r1, 0) | eq_bytes(t1, 'N') | eq_bytes(r1, 'N')) & vm1;
          const int m = __builtin_popcount(m0) + __builtin_popcount(m1b);
          sc += m * A.match + (nt - m) * A.mismatch;
        } else if (((eq_bytes(t0 & 0xfcfcfcfcu, 0) | eq_bytes(r0 & 0xfcfcfcfcu, 0)) & vm0) | ((eq_bytes(t1 & 0xfcfcfcfcu, 0) | eq_bytes(r1 & 0xfcfcfcfcu, 0)) & vm1)) {
          // a byte below 4 is its own code in seq_nt4_table: character by character
          for (int t = 0; t < nt; ++t) {
            const unsigned char ct = sel_nt4((unsigned char)(tw >> (8 * t))), cq = sel_nt4((unsigned char)(rw >> (8 * t)));
            const int mx = cq < 4 ? a : 0, v = (ct < 4 && cq < 4) ? (ct == cq ? a : b) : 0;
            sc += v; loss += mx - v;
          }
          bad += 1;
        } else {
          // ... the byte is one of a c g t in either case (code < 4), and where both are and they are the same letter
          const u32 at0 = eq_bytes((t0 & 0xdfdfdfdfu) ^ canon4(t0, false), 0), at1 = eq_bytes((t1 & 0xdfdfdfdfu) ^ canon4(t1, false), 0);
          const u32 aq0 = eq_bytes((r0 & 0xdfdfdfdfu) ^ canon4(r0, false), 0) & vm0, aq1 = eq_bytes((r1 & 0xdfdfdfdfu) ^ canon4(r1, false), 0) & vm1;
          const u32 e0 = eq_bytes((t0 ^ r0) & 0xdfdfdfdfu, 0) & at0 & aq0, e1 = eq_bytes((t1 ^ r1) & 0xdfdfdfdfu, 0) & at1 & aq1;
          const int nQ = __builtin_popcount(aq0) + __builtin_popcount(aq1), nBoth = __builtin_popcount(at0 & aq0) + __builtin_popcount(at1 & aq1);
          const int nEq = __builtin_popcount(e0) + __builtin_popcount(e1);
          const int v = a * nEq + b * (nBoth - nEq);
          sc += v; loss += a * nQ - v;
          if (rule2) {
            // the first position that is not a match, whether anything here is not a c g t, the last mismatch of the neighbouring diagonals
            const u64 mm = (u64)(vm0 & ~e0) | ((u64)(vm1 & ~e1) << 32);
            if (mm) { const int p = i0 + (__builtin_ctzll(mm) >> 3); m1 = p < m1 ? p : m1; }
            bad += __builtin_popcount(vm0 & ~(at0 & aq0)) + __builtin_popcount(vm1 & ~(at1 & aq1));
            const u64 tn = sel_word_at(tseq1, i0 + 8, tlen1), tp = i0 >= 8 ? load_u64_unaligned(tseq1 + i0 - 8) : 0ULL;
#pragma unroll
            for (int d = 1; d <= 3; ++d) {
              if (d > Ld) continue;
              const u64 td = (tw >> (8 * d)) | (tn << (64 - 8 * d));
              const u32 d0 = (u32)td, d1 = (u32)(td >> 32);
              const u64 dm = (u64)(vm0 & ~eq_bytes((d0 ^ r0) & 0xdfdfdfdfu, 0)) | ((u64)(vm1 & ~eq_bytes((d1 ^ r1) & 0xdfdfdfdfu, 0)) << 32);
              if (dm) { const int p = i0 + ((63 - __builtin_clzll(dm)) >> 3); hmD[d - 1] = p > hmD[d - 1] ? p : hmD[d - 1]; }
              if (d == Ld) bad += __builtin_popcount(vm0 & ~eq_bytes((d0 & 0xdfdfdfdfu) ^ canon4(d0, false), 0)) + __builtin_popcount(vm1 & ~eq_bytes((d1 & 0xdfdfdfdfu) ^ canon4(d1, false), 0));
            }
#pragma unroll
            for (int d = 1; d <= 2; ++d) {
              if (d > Li) continue;
              const u64 td = (tw << (8 * d)) | (tp >> (64 - 8 * d));
              const u32 d0 = (u32)td, d1 = (u32)(td >> 32);
              const u64 dm = (u64)(vm0 & ~eq_bytes((d0 ^ r0) & 0xdfdfdfdfu, 0)) | ((u64)(vm1 & ~eq_bytes((d1 ^ r1) & 0xdfdfdfdfu, 0)) << 32);
              if (dm) { const int p = i0 + ((63 - __builtin_clzll(dm)) >> 3); hmI[d - 1] = p > hmI[d - 1] ? p : hmI[d - 1]; }
            }
          }
        }
      }
    }
    sc = red.add(sc); loss = red.add(loss);
    if (multiMapping) { key = red.bxor(hx) ^ ((u64)(u32)keyLen * 0x9E3779B97F4A7C15ULL); kind |= 1; }
    bool known = diag && loss <= qe;
    if (rule2 && !known && loss == 2 * M && red.add(bad) == 0) {
      m1 = red.min(m1);
      int best = 2 * M;
#pragma unroll
      for (int d = 1; d <= 3; ++d) if (d <= Ld && red.max(hmD[d - 1]) <= m1 - 1) { const int c = gq + d * ge; best = c < best ? c : best; }
#pragma unroll
      for (int d = 1; d <= 2; ++d) if (d <= Li && red.max(hmI[d - 1]) <= m1 + d - 1) { const int c = gq + d * (ge + a); best = c < best ? c : best; }
      known = true; sc = a * rlen - best;
    }
    if (doUngapped) s = sc;
    else if (known) s = sc;
    else {
      kind |= 2;
      // An alignment whose gapless path loses no more than q + 7 e: no path that scores as much leaves the diagonals -7 .. +7 (gap runs
      // that move it further lose more than that), so an exact affine-gap DP over that strip -- sixteen lanes, one per diagonal
      // (sel_tasks_strip) -- returns what ksw2 returns, at a quarter of its instructions.
      if (diag && A.torder2 && A.no_diag != 3 && loss <= gq + 7 * ge && (A.bandwidth < 0 || A.bandwidth >= 9) && readLen <= QM_MAX_READ_LEN) kind |= 4;
      if (l == 0) {
        SelTask t; t.rd = read; t.tx = tseq1; t.gslot = (int)g; t.rl = readLen; t.roff = roff; t.rlen = rlen; t.tlen1 = tlen1; t.fwd = fwd ? 1 : 0;
        A.tasks[x] = t;
      }
    }
  }
  if (l == 0) { A.tsc[g] = s; A.tref[g] = -1; A.sides[x].key = key; A.sides[x].kind = kind; }
}

// step 3: the alignment cache (SelectiveAlignmentUtils.hpp: alnCache keyed by the hash of the target window, per read and mate) and
// the work list of the alignment kernel
// (returns the list the question goes on: 0 none, 1 the ksw2 kernel's, 2 the strip kernel's; the caller appends -- qm_sel_dedupe_kernel one
// atomic per wavefront and list, not one per question: 0.85 M additions to ONE address per chunk were 1.4 of the kernel's 1.8 ms)
QM_DEV int sel_side_dedupe_list(const SelBatch& A, long long x) {
  const SelSide S = A.sides[x];
  if (S.kind == 0) return 0;
  if (S.kind & 1) {
    long long own = -1;                                    // (the unit's earlier questions are the S.nth before this one: loads that do not wait for one another)
    for (long long y = x - 1; y >= x - S.nth; --y) {
      const SelSide& Y = A.sides[y];
      if ((Y.kind & 1) && ((Y.g ^ S.g) & 1) == 0 && Y.key == S.key) own = y;
    }
    if (own >= 0) {
      A.tref[S.g] = -(int)(A.sides[own].g - 2 * A.toff[S.u]) - 2;
      A.tsc[S.g] = (int)0x80000000;
      return 0;
    }
  }
  return (S.kind & 4) ? 2 : ((S.kind & 2) ? 1 : 0);
}
QM_DEV void sel_side_dedupe(const SelBatch& A, long long x) {
  const int w = sel_side_dedupe_list(A, x);
  if (w == 2) A.torder2[atomic_add_u64(A.ntasks2, 1ULL)] = (u64)x;
  else if (w == 1) A.torder[atomic_add_u64(A.ntasks, 1ULL)] = (u64)x;
}

// The strip alignments: tasks t0 .. t0+3 of torder2, one per row of 16 lanes, lane c = diagonal c - 7 (target index = query index + c - 7).
// The extension alignment of ksw_extz2 as a plain recurrence in 32-bit integers -- H(-1,-1) = 0, a gap of length L costs q + L e,
// score = max over the query's last row and the target's last column -- restricted to the strip, which holds every path that can
// score as much as the gapless one (sel_side_score sends an alignment here only then).  One query row per trip: the diagonal
// move stays in its lane, the move down comes from the lane above (row_shl), the moves along the row are an exclusive max scan
// over the lanes (H + e c is what a gap from lane c' < c brings to lane c, minus q + e c).  Held against the oracle's ksw2 like
// every other score: tests/test_ksw_variants.py restates it in Python, the emulation and GPU parity tests run it.
struct StripMem { unsigned char q[4][QM_MAX_READ_LEN + 16]; unsigned char t[4][QM_MAX_READ_LEN + 48]; };
QM_DEV void sel_tasks_strip(const SelBatch& A, unsigned long long t0, unsigned long long nt, StripMem& M) {
  const int NEGI = -(1 << 28);
  LV<const unsigned char*> rd, tx; LV<int> ql, tl, gs, rl, ro, fw;
  QM_LANES(l) {
    const unsigned long long ti = t0 + (unsigned long long)(l >> 4);
    ql[l] = 0; tl[l] = 0; gs[l] = -1; rd[l] = nullptr; tx[l] = nullptr; rl[l] = 0; ro[l] = 0; fw[l] = 0;
    if (ti < nt) {
      const SelTask t = A.tasks[A.torder2[ti]];
      rd[l] = t.rd; rl[l] = t.rl; tx[l] = t.tx; ql[l] = t.rlen; tl[l] = t.tlen1; gs[l] = t.gslot; ro[l] = t.roff; fw[l] = t.fwd;
    }
  }
  const int maxq = wave_max(ql), maxt = wave_max(tl);
  for (int i0 = 0; i0 < maxt; i0 += 16) {
    QM_LANES(l) {
      const int i = i0 + (l & 15), g = l >> 4;
      if (gs[l] >= 0) {
        if (i < ql[l]) M.q[g][i] = sel_nt4(sel_read_char(rd[l], rl[l], fw[l] != 0, ro[l] + i));
        if (i < tl[l]) M.t[g][i] = sel_nt4(tx[l][i]);
      }
    }
  }
  wave_fence();
  int a = (signed char)A.match, b = (signed char)A.mismatch;
  a = a < 0 ? -a : a; b = b > 0 ? -b : b;
  const int go = (int)(signed char)A.gap_open, ge = (int)(signed char)A.gap_extend;
  LV<int> Hp, Fp, mq, mt;
  QM_LANES(l) {
    const int j = (l & 15) - 8;                             // the row above the first: H(-1, -1) = 0, H(-1, j) = -(q + e (j + 1))
    Hp[l] = j == -1 ? 0 : ((j >= 0 && j < tl[l]) ? -(go + ge * (j + 1)) : NEGI);
    Fp[l] = NEGI; mq[l] = NEGI; mt[l] = NEGI;
  }
  for (int i = 0; i < maxq; ++i) {
    LV<int> Hn, Fn, Ht, F, X;
    row16_shl1(Hp, Hn, NEGI); row16_shl1(Fp, Fn, NEGI);
    // (no tests for "not reachable": such a value is NEGI give or take a few gap costs, far below every real score, and loses every max)
    QM_LANES(l) {
      const int c = l & 15, g = l >> 4, j = i + c - 7;
      const bool act = gs[l] >= 0 && i < ql[l], inr = act && j >= 0 && j < tl[l], bnd = act && j == -1;
      const int qc = M.q[g][act ? i : 0], tc = M.t[g][inr ? j : 0];
      const int s = (qc < 4 && tc < 4) ? (qc == tc ? a : b) : 0;
      const int m = Hp[l] + s, f1 = Hn[l] - go - ge, f2 = Fn[l] - ge;
      const int f = f1 > f2 ? f1 : f2;
      const int ht = bnd ? -(go + ge * (i + 1)) : (inr ? (m > f ? m : f) : NEGI);
      Ht[l] = ht; F[l] = inr ? f : NEGI; X[l] = ht + ge * c;
    }
    row16_scan_max_excl(X, NEGI);
    QM_LANES(l) {
      const int c = l & 15, j = i + c - 7;
      const bool act = gs[l] >= 0 && i < ql[l], inr = act && j >= 0 && j < tl[l], bnd = act && j == -1;
      const int e = X[l] - go - ge * c;
      const int h = bnd ? Ht[l] : (inr ? (Ht[l] > e ? Ht[l] : e) : NEGI);
      if (inr && i == ql[l] - 1 && h > mq[l]) mq[l] = h;
      if (inr && j == tl[l] - 1 && h > mt[l]) mt[l] = h;
      if (act) { Hp[l] = h; Fp[l] = F[l]; }
    }
  }
  LV<int> r;
  QM_LANES(l) { r[l] = mq[l] > mt[l] ? mq[l] : mt[l]; }
  row16_max_all(r);
  QM_LANES(l) { if (gs[l] >= 0 && (l & 15) == 0) A.tsc[gs[l]] = r[l]; }
  wave_fence();
}

// Tasks t0 .. t0+3 (those below nt), one per row of 16 lanes: stage the two score-phase images straight from the read and
// the transcript text, run the row kernel, lane 0 of every row stores its score.
#define QM_KSW_STAGE_UNROLL 4
// character -> nt4 code of the score phase, for a read taken as it is (codes[c]) and reverse-complemented (codes[256 + c]): one
// table look-up in LDS per staged character instead of a dozen compares (the block fills it once, sel_ksw_fill_codes)
QM_DEV void sel_ksw_fill_codes(QM_LDS(unsigned char)* codes, int tid, int nthreads) {
  for (int c = tid; c < 256; c += nthreads) { codes[c] = sel_nt4((unsigned char)c); codes[256 + c] = sel_nt4(rc_char((unsigned char)c)); }
}
// stage the images of tasks t0 .. t0+3 (those below nt) into blk[0..3]; ql / tl / gs: each row's lengths and score slot (-1: no task)
template <int RING, int MAXLEN>
QM_DEV void sel_tasks_stage(const SelBatch& A, unsigned long long t0, unsigned long long nt, KswRowT<RING, MAXLEN>* blk,
                            const QM_LDS(unsigned char)* codes, LV<int>& ql, LV<int>& tl, LV<int>& gs) {
  LV<const unsigned char*> rd, tx; LV<int> rl, ro, fw;
  QM_LANES(l) {
    const unsigned long long ti = t0 + (unsigned long long)(l >> 4);
    ql[l] = 0; tl[l] = 0; gs[l] = -1; rd[l] = nullptr; tx[l] = nullptr; rl[l] = 0; ro[l] = 0; fw[l] = 0;
    if (ti < nt) {
      const SelTask t = A.tasks[A.torder ? A.torder[ti] : ti];
      rd[l] = t.rd; rl[l] = t.rl; tx[l] = t.tx;
      ql[l] = t.rlen; tl[l] = t.tlen1; gs[l] = t.gslot; ro[l] = t.roff; fw[l] = t.fwd;
    }
  }
  // the images are read up to column max(en) + 31 < tlen + 31 and query index qlen + 30; their defined part ends at
  // tlen16 + qlen (TX) and 16 + qlen (QX): stage that much (of the longest of the four alignments), not the whole block
  LV<int> need;
  QM_LANES(l) { need[l] = gs[l] >= 0 ? ((tl[l] + 15) / 16 * 16) + ql[l] + 48 : 0; }
  const int needMax = wave_max(need);
  const int stageEnd = needMax < MAXLEN + 40 ? needMax : MAXLEN + 40;
  // four positions per lane and pass, no branches: an address (clamped to something readable) and a select per image -- the eight
  // loads of a pass are in flight together, then their eight code look-ups
  for (int i0 = 0; i0 < stageEnd; i0 += 16 * QM_KSW_STAGE_UNROLL) {
    QM_LANES(l) {
      if (gs[l] >= 0) {
        KswRowT<RING, MAXLEN>& B = blk[l >> 4];
        const int qlen = ql[l], tlen = tl[l], tlen16 = (tlen + 15) / 16 * 16;
        const int rcOff = fw[l] != 0 ? 0 : 256;
        const bool fwd = fw[l] != 0;
        const unsigned char* rp = rd[l]; const unsigned char* tp = tx[l];
        const int rlast = rl[l] - 1, rof = ro[l];
        unsigned char cq[QM_KSW_STAGE_UNROLL], ct[QM_KSW_STAGE_UNROLL];
#pragma unroll
        for (int h = 0; h < QM_KSW_STAGE_UNROLL; ++h) {
          const int i = i0 + 16 * h + (l & 15);
          // QX[i]: query character i - 16
          const bool qok = i >= 16 && i < 16 + qlen;
          const int qj = qok ? rof + i - 16 : 0;
          cq[h] = rp[fwd ? qj : rlast - qj];
          // TX[i]: target character i, or query character qlen - 1 - (i - tlen16) behind the padded target
          const int j = i - tlen16;
          const bool isT = i < tlen, isQ = !isT && j >= 0 && j < qlen;
          const int tj = isQ ? rof + qlen - 1 - j : 0;
          const unsigned char* ta = isT ? tp + i : rp + (fwd ? tj : rlast - tj);
          ct[h] = *ta;
        }
#pragma unroll
        for (int h = 0; h < QM_KSW_STAGE_UNROLL; ++h) {
          const int i = i0 + 16 * h + (l & 15);
          const bool isT = i < tlen;
          cq[h] = codes[rcOff + cq[h]];
          ct[h] = codes[(isT ? 0 : rcOff) + ct[h]];
        }
#pragma unroll
        for (int h = 0; h < QM_KSW_STAGE_UNROLL; ++h) {
          const int i = i0 + 16 * h + (l & 15);
          const bool qok = i >= 16 && i < 16 + qlen;
          const int j = i - tlen16;
          const bool isT = i < tlen, isQ = !isT && j >= 0 && j < qlen;
          if (i < MAXLEN + 40) {
            B.QX[i] = qok ? cq[h] : (unsigned char)0;
            B.TX[i] = (isT || isQ) ? ct[h] : (unsigned char)0;
          }
        }
      }
    }
  }
  wave_fence();
}
QM_DEV void sel_ksw_matrix(const SelBatch& A, signed char* mat) {      // ksw_gen_simple_mat as the reference's aligner sets it up
  int a = (signed char)A.match, b = (signed char)A.mismatch;
  a = a < 0 ? -a : a; b = b > 0 ? -b : b;
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[i * 5 + j] = (signed char)(i == j ? a : b); mat[i * 5 + 4] = 0; }
  for (int j = 0; j < 5; ++j) mat[20 + j] = 0;
}
template <int RING, int MAXLEN = QM_KSW_MAXLEN>
QM_DEV void sel_tasks_align_rows(const PairBatch& P, const SelBatch& A, unsigned long long t0, unsigned long long nt, KswRowT<RING, MAXLEN>* blk,
                                 const QM_LDS(unsigned char)* codes) {
  LV<int> ql, tl, gs;
  sel_tasks_stage<RING, MAXLEN>(A, t0, nt, blk, codes, ql, tl, gs);
  signed char mat[25];
  sel_ksw_matrix(A, mat);
  LV<int> sc;
  sel_ksw_extz2_rows<RING, MAXLEN>(ql, tl, blk, mat, (signed char)A.gap_open, (signed char)A.gap_extend, A.bandwidth, sc);
  QM_LANES(l) { if (gs[l] >= 0 && (l & 15) == 0) A.tsc[gs[l]] = sc[l]; }
  wave_fence();
}
// The register edition, eight tasks per wavefront (t0 .. t0+7 in blk[0..7]): when all of them share their lengths they go through
// the rounds together as two sets (sel_ksw_extz2_rows_reg<.., UNI, 2>), else four and four
template <int MAXLEN>
QM_DEV void sel_tasks_align_rows2(const PairBatch& P, const SelBatch& A, unsigned long long t0, unsigned long long nt, KswRowT<32, MAXLEN>* blk,
                                  const QM_LDS(unsigned char)* codes) {
  LV<int> ql[2], tl[2], gs[2];
  sel_tasks_stage<32, MAXLEN>(A, t0, nt, blk, codes, ql[0], tl[0], gs[0]);
  sel_tasks_stage<32, MAXLEN>(A, t0 + 4, nt, blk + 4, codes, ql[1], tl[1], gs[1]);
  signed char mat[25];
  sel_ksw_matrix(A, mat);
  LV<int> sc[2];
  LV<int> key; LV<bool> odd;
  QM_LANES(l) { const int k0 = gs[0][l] >= 0 ? ((ql[0][l] << 16) | tl[0][l]) : -1, k1 = gs[1][l] >= 0 ? ((ql[1][l] << 16) | tl[1][l]) : -1; key[l] = k0 > k1 ? k0 : k1; }
  const int kmax = wave_max(key);
  QM_LANES(l) {
    const int k0 = gs[0][l] >= 0 ? ((ql[0][l] << 16) | tl[0][l]) : -1, k1 = gs[1][l] >= 0 ? ((ql[1][l] << 16) | tl[1][l]) : -1;
    odd[l] = (k0 >= 0 && k0 != kmax) || (k1 >= 0 && k1 != kmax) || (ql[0][l] <= 0 && gs[0][l] >= 0) || (tl[0][l] <= 0 && gs[0][l] >= 0);
  }
  LV<bool> has1; QM_LANES(l) { has1[l] = gs[1][l] >= 0; }
  const bool both = ballot(has1) != 0;
  if (kmax > 0 && both && !ballot(odd) && (kmax >> 16) > 0 && (kmax & 0xffff) > 0 && A.bandwidth >= 0 && A.bandwidth <= 15) {
    sel_ksw_extz2_rows_reg<MAXLEN, true, 2>(ql, tl, kmax >> 16, kmax & 0xffff, blk, mat, (signed char)A.gap_open, (signed char)A.gap_extend, A.bandwidth, sc);
  } else {
    sel_ksw_extz2_rows<32, MAXLEN>(ql[0], tl[0], blk, mat, (signed char)A.gap_open, (signed char)A.gap_extend, A.bandwidth, sc[0]);
    sel_ksw_extz2_rows<32, MAXLEN>(ql[1], tl[1], blk + 4, mat, (signed char)A.gap_open, (signed char)A.gap_extend, A.bandwidth, sc[1]);
  }
  for (int s = 0; s < 2; ++s) { QM_LANES(l) { if (gs[s][l] >= 0 && (l & 15) == 0) A.tsc[gs[s][l]] = sc[s][l]; } }
  wave_fence();
}

QM_DEV int sel_unit_finish(const PairBatch& P, const SelBatch& A, long long u, UnitCounters* uc) {
  const int LOWEST = (int)0x80000000;
  int n = (int)P.cnt[u];
  qm_hit* T = A.tmp + A.toff[u];
  const long long gbase = 2 * A.toff[u];
  auto score_of = [&](int local) { const int r = A.tref[gbase + local]; return r <= -2 ? A.tsc[gbase + (-(r) - 2)] : A.tsc[gbase + local]; };
  const u32 l1 = (u32)(P.off1[u + 1] - P.off1[u]);
  int bestScore = LOWEST;
  if (!P.paired) {
    const int maxReadScore = A.match * (int)l1;
    for (int i = 0; i < n; ++i) {
      const int s = score_of(2 * i);
      const int score = ((double)s < A.min_score_fraction * (double)maxReadScore) ? LOWEST : s;
      bestScore = score > bestScore ? score : bestScore;
      T[i].aln_score = score;
    }
  } else {
    const u32 l2 = (u32)(P.off2[u + 1] - P.off2[u]);
    const int maxLeftScore = A.match * (int)l1, maxRightScore = A.match * (int)l2;
    for (int i = 0; i < n; ++i) {
      qm_hit& h = T[i];
      int score = LOWEST;
      if (h.mate_status == 3) {
        int s1 = score_of(2 * i), s2 = score_of(2 * i + 1);
        if (h.fwd != h.mate_is_fwd && P.no_dovetail) {
          if (h.fwd && h.pos > h.mate_pos) { s1 = LOWEST; s2 = LOWEST; }
          else if (h.mate_is_fwd && h.mate_pos > h.pos) { s1 = LOWEST; s2 = LOWEST; }
        }
        if (((double)s1 < A.min_score_fraction * (double)maxLeftScore) || ((double)s2 < A.min_score_fraction * (double)maxRightScore)) score = LOWEST;
        else score = s1 + s2;
      } else if (h.mate_status == 1) {
        const int s = score_of(2 * i);
        score = ((double)s < A.min_score_fraction * (double)maxLeftScore) ? LOWEST : s;
      } else {
        const int s = score_of(2 * i + 1);
        score = ((double)s < A.min_score_fraction * (double)maxRightScore) ? LOWEST : s;
      }
      bestScore = score > bestScore ? score : bestScore;
      h.aln_score = score;
    }
  }
  int o = 0;
  if (bestScore > LOWEST)
    for (int i = 0; i < n; ++i) { const bool rem = A.hard_filter ? (T[i].aln_score < bestScore) : (T[i].aln_score == LOWEST); if (!rem) { if (o != i) T[o] = T[i]; ++o; } }
  n = o;
  if (uc) { if (P.paired) uc->tot += (u64)n; if (n > 0) uc->mapped += 1; }
  return n;
}
