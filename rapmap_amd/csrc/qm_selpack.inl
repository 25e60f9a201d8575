// qm_selpack.inl -- the list kernel of -s, SEVERAL reads per wavefront (round 4).  Included at the end of qm_mapper.inl.
//
// qm_h2m_kernel<QM_F_SEL> gives a wavefront one read: its suffixes (a dozen on a transcriptome: the SA intervals of a read
// hold one suffix per isoform) occupy a fifth of the lanes, and the kernel is bound by the instructions it issues, not by
// memory.  Here a wavefront takes a run of consecutive reads whose intervals AND suffixes fit its 64 lanes -- four or five
// reads at a time on 2 x 100 bp -- and walks them through the same steps together: one lane per interval, one lane per
// suffix, a rank sort inside each read's segment of lanes (key compare = the borrow of a 96-bit subtraction, a sentinel key past
// the segment), every record ORs its interval's bit into its transcript's first word (one LDS atomic) and checks its own link of
// the chain hit 0 <- hit 1 <- ... -- a transcript whose hits all pass is chained by a prefix sum; the others (several diagonals,
// ties) keep a serial job on the lane of their first record: sel_chain_diag8 / sel_chain_diag_mem (integers, one diagonal) or
// sel_chain_group (the doubles, unchanged) --, prefix sums place every group's words in the batch's output.
// What it computes per read is sel_hits_to_mappings for a read whose hits lie on ONE strand (HitManager.cpp:587-689 slack
// intersection, :84-326 chaining, :716-807 single interval, :834-881 with an empty other side); a read with hits on both
// strands, with more than 64 intervals or more than 64 suffixes goes on a queue for qm_h2m_kernel, which runs after this
// one over that queue alone.

#define QM_PK_READS 16             // reads a wavefront looks at per batch (it keeps the leading ones that fit)
#define QM_SC_TODO 28              // slot of the context's scalar block: reads left for qm_h2m_kernel

// (k1, k2) < (a1, a2) for the rank sorts' keys (k2 and a2 below 2^32: query end << 16 | lane), counted into cnt: the borrow of the
// 96-bit subtraction (k1 : k2) - (a1 : a2), added with carry -- four full-rate instructions.  (Written as compares, or as __builtin_subc,
// the compiler makes three 64-bit compares of it, which run at a fraction of the rate.)  A key of all ones is below nothing.
QM_DEV void key_count(u64 c1, u64 c2, u64 a1, u64 a2, int& cnt) {
#ifdef QM_EMU
  cnt += (int)(c1 < a1) | ((int)(c1 == a1) & (int)((u32)c2 < (u32)a2));
#else
  u32 tmp;
  asm("v_sub_co_u32 %1, vcc, %2, %3\n\t"
      "v_subb_co_u32 %1, vcc, %4, %5, vcc\n\t"
      "v_subb_co_u32 %1, vcc, %6, %7, vcc\n\t"
      "v_addc_co_u32 %0, vcc, 0, %0, vcc"
      : "+v"(cnt), "=&v"(tmp)
      : "v"((u32)c2), "v"((u32)a2), "v"((u32)c1), "v"((u32)a1), "v"((u32)(c1 >> 32)), "v"((u32)(a1 >> 32))
      : "vcc");
#endif
}

struct PackMem {                   // one wavefront's LDS: 5.3 KB
  long long ivoff[QM_PK_READS];    // per read of the batch: where its interval records start in iv_in,
  int ivcnt[QM_PK_READS];          //   how many (m: the intervals of its one strand),
  int rlen[QM_PK_READS];           //   its length,
  int ivs[QM_PK_READS];            //   the lane of its first interval,
  int rb[QM_PK_READS], rn[QM_PK_READS];   // the lane of its first suffix, its suffixes (-1: the read is left for qm_h2m_kernel)
  int rrc[QM_PK_READS];            //   1: its hits are on the reverse-complement strand
  union {
    struct { IntRec iv[64]; int slot[64]; int rs[64]; } a;   // per interval: the record, its read, the lane of its first suffix
    u64 out[192];                                           // the batch's list words (a group of h records: at most 2 + h words)
  };
  int mark[64];
  SelRec rec[64];                  // (iv: interval in its read | read of the batch << 8)
  union {
    struct { u64 k1[64 + 1], k2[64 + 1]; } k;                // sort keys (+1: the rank sort's sentinel, a key below nothing)
    struct { double f[64]; int p[64], seen[64]; } c;         // chaining DP
  };
  int ends[64], starts[64];
  int sw[64];                      // a scan, where every lane can read it
};

// sel_chain_group for hn <= 8 hits H (in chain order) that share one diagonal (pos - qpos): the same DP, end collection and
// backtracking with everything in registers.  With dq == dt for every pair of hits the gap cost beta is 0 (dq >= 0 by the
// order, dq <= the read's length), alpha = min(len_i, dq), and every score is an integer below 2^31: the comparisons
// `cand > fi`, `fi > bestScore`, `fi == bestScore` come out as they do on the doubles.  Returns 0 when the group is not of that
// kind (the caller takes sel_chain_group); otherwise the number of chain starts, g filled (all but its offset; the score is
// not needed by a read with hits on one strand) and posOut[0 .. starts) = the diagonal.
QM_DEV int sel_chain_diag8(const SelRec* H, int hn, int maxDist, SelGroup& g, int* posOut) {
  if (hn > 8) return 0;
  int e[8], ln[8];
  const SelRec h0 = H[0];
  const int diag = (int)(h0.pos - h0.qpos);
  bool same = true;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    e[i] = 0; ln[i] = 0;
    if (i < hn) { const SelRec h = H[i]; e[i] = (int)(h.qpos + h.len); ln[i] = (int)h.len; same = same && (int)(h.pos - h.qpos) == diag; }
  }
  if (!same) return 0;
  int f[8]; u32 P = 0;                                     // p[i]: four bits each
  int best = -1, lastBest = -1; u32 endsMask = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f[i] = 0;
    if (i < hn) {
      int fi = ln[i], pi = i, looksLeft = 2; bool done = false;
#pragma unroll
      for (int j = i - 1; j >= 0; --j) {
        if (!done) {
          const int dq = e[i] - e[j];
          const int cand = f[j] + (ln[i] < dq ? ln[i] : dq);
          const bool take = cand > fi;
          pi = take ? j : pi; fi = take ? cand : fi;
          if (pi < i) { --looksLeft; if (looksLeft <= 0) done = true; }
        }
      }
      f[i] = fi; P |= (u32)pi << (4 * i);
      if (fi > best) { best = fi; lastBest = i; endsMask = 1u << i; }
      else if (fi == best) endsMask |= 1u << i;
    }
  }
  // multi-chain backtracking (:206-246)
  u32 seen = 0; int nOptimal = 0, nStarts = 0;
  for (u32 em = endsMask; em; em &= em - 1) {
    int cur = __builtin_ctz(em);
    bool fresh = true;
    int prev = (int)((P >> (4 * cur)) & 15u);
    while (prev < cur) {
      if ((seen >> cur) & 1u) { fresh = false; break; }
      seen |= 1u << cur;
      cur = prev;
      prev = (int)((P >> (4 * cur)) & 15u);
    }
    if ((seen >> cur) & 1u) fresh = false;
    if (fresh) { ++nOptimal; posOut[nStarts++] = diag; }   // every start lies on the diagonal: allPositions is that value, nStarts times
  }
  g.tid = h0.tid; g.offcs = 0; g.set_cs(QM_CS_REGULAR); g.score = (double)best; g.npos = nStarts; g.ppos = diag;
  if (hn > 1 && nOptimal == 1 && lastBest == hn - 1) {       // gapless chain (:283-305): on one diagonal the two spans are equal
    const SelRec hl = H[hn - 1];
    const long long qSpan = (long long)(hl.qpos + hl.len) - (long long)h0.qpos;
    if (qSpan == (long long)maxDist) g.set_cs(QM_CS_UNGAPPED);
  }
  return nStarts;
}

QM_DEV int sel_chain_diag_mem(const SelRec* H, int hn, int maxDist, int* f, int* p, int* seen, int* ends, SelGroup& g, int* posOut);
// One batch: the reads r0 .. of the wave's range [r0, rEnd).  Returns how many it consumed (>= 1).
QM_DEV int sel_pack_batch(const DevIndex& ix, const ReadBatch& B, long long r0, long long rEnd, PackMem& M, WaveAlloc& wa, long long* todoq) {
#pragma clang fp contract(off)
  const bool paired = B.seq2 != nullptr;
  // ---- the candidates: interval count, offsets, length and foundHit of up to 16 reads, one per lane
  LV<int> cnt, scan; LV<u32> fflag;
  QM_LANES(l) {
    const long long read = r0 + l;
    int c = 0; u32 ff = 0;
    if (l < QM_PK_READS && read < rEnd) {
      c = (int)B.iv_in_cnt[read];
      M.ivoff[l] = B.iv_in_off[read];
      const int mate = paired ? (int)(read & 1) : 0; const long long unit = paired ? (read >> 1) : read;
      const long long* off = mate ? B.off2 : B.off1;
      M.rlen[l] = (int)(off[unit + 1] - off[unit]);
      M.ivcnt[l] = c;
      ff = (B.found_in && B.found_in[read]) ? 0x80000000u : 0u;
    }
    cnt[l] = c; scan[l] = c; fflag[l] = ff;
  }
  lane_scan_add(scan);
  LV<bool> ok;
  QM_LANES(l) { ok[l] = l < QM_PK_READS && r0 + l < rEnd && scan[l] <= 64; }
  int R = ctz64(~ballot(ok));                             // the leading reads whose intervals fit the 64 lanes
  if (R == 0) {                                           // more than 64 intervals in one read (long reads)
    QM_LANES(l) { if (l == 0) { const u64 q = atomic_add_u64(B.cursor + QM_SC_TODO, 1ULL); todoq[q] = r0; } }
    return 1;
  }
  const int NI = read_lane(scan, R - 1);
  QM_LANES(l) { if (l < QM_PK_READS) M.ivs[l] = scan[l] - cnt[l]; M.mark[l] = 0; }
  wave_fence();
  // ---- one lane per interval; a lane finds its read by a running maximum over marks left at every read's first lane
  QM_LANES(l) { if (l < R && cnt[l] > 0) M.mark[scan[l] - cnt[l]] = l; }
  wave_fence();
  LV<int> slot;
  QM_LANES(l) { slot[l] = M.mark[l]; }
  lane_scan_max(slot);
  LV<int> w, wsc; LV<bool> isF, isR;
  QM_LANES(l) {
    int ww = 0; bool f = false, r = false;
    if (l < NI) {
      const int s = slot[l];
      const qm_sa_interval_hit h = B.iv_in[M.ivoff[s] + (long long)(l - M.ivs[s])];
      IntRec q; q.b = (u32)h.begin; q.e = (u32)h.end; q.len = h.len; q.q = h.query_pos;
      M.a.iv[l] = q; M.a.slot[l] = s;
      ww = (int)(q.e - q.b); r = h.query_rc != 0; f = !r;
    }
    w[l] = ww; wsc[l] = ww; isF[l] = f; isR[l] = r;
  }
  const u64 fm = ballot(isF), rm = ballot(isR);
  QM_T(0);
  lane_scan_add(wsc);
  QM_LANES(l) { M.sw[l] = wsc[l]; M.mark[l] = 0; }
  wave_fence();
  // ---- per read: its suffixes, whether this kernel takes it; then the leading reads whose suffixes fit the 64 lanes
  LV<int> rcnt, rscan; LV<bool> pk;
  QM_LANES(l) {
    int n = 0; bool p = false; int rc = 0;
    if (l < R) {
      const int c = cnt[l], s = scan[l] - c;
      p = true;
      if (c > 0) {
        n = M.sw[s + c - 1] - (s > 0 ? M.sw[s - 1] : 0);
        const u64 range = lanemask_lt(s + c) & ~lanemask_lt(s);
        const bool hf = (fm & range) != 0, hr = (rm & range) != 0;
        p = !(hf && hr) && n <= 64;
        rc = hr ? 1 : 0;
      }
      if (!p) n = 0;
      M.rrc[l] = rc;
    }
    rcnt[l] = n; rscan[l] = n; pk[l] = p;
  }
  lane_scan_add(rscan);
  LV<bool> fit;
  QM_LANES(l) { fit[l] = l < R && rscan[l] <= 64; }
  R = ctz64(~ballot(fit));                                // >= 1: no read brings more than 64
  const int NR = read_lane(rscan, R - 1);
  QM_LANES(l) { if (l < R) { M.rb[l] = rscan[l] - rcnt[l]; M.rn[l] = pk[l] ? rcnt[l] : -1; } }
  LV<bool> td;
  QM_LANES(l) { td[l] = l < R && !pk[l]; }
  const u64 tdm = ballot(td);
  if (tdm) {
    LV<u64> qb;
    QM_LANES(l) { qb[l] = 0; if (l == 0) qb[l] = atomic_add_u64(B.cursor + QM_SC_TODO, (u64)popc64(tdm)); }
    const u64 q0 = read_lane(qb, 0);
    QM_LANES(l) { if (td[l]) todoq[q0 + (u64)popc64(tdm & lanemask_lt(l))] = r0 + l; }
  }
  wave_fence();
  // ---- one lane per suffix: marks at every interval's first suffix, a running maximum again, one trip to sainfo
  QM_LANES(l) {
    if (l < NI) {
      const int s = slot[l];
      if (s < R && M.rn[s] > 0 && w[l] > 0) {
        const int first = M.ivs[s];
        const int rs = M.rb[s] + (wsc[l] - w[l]) - (first > 0 ? M.sw[first - 1] : 0);
        M.a.rs[l] = rs; M.mark[rs] = l;
      }
    }
  }
  wave_fence();
  LV<int> ivl;
  QM_LANES(l) { ivl[l] = M.mark[l]; }
  lane_scan_max(ivl);
  LV<SelRec> mine; LV<int> myslot;
  QM_LANES(l) {
    myslot[l] = 0;
    if (l < NR) {
      const int j = ivl[l];
      const IntRec q = M.a.iv[j]; const int s = M.a.slot[j];
      const SaInfo e = ix.sainfo[q.b + (u32)(l - M.a.rs[j])];
      SelRec r; r.tid = e.tid; r.pos = (u32)e.pos; r.qpos = q.q; r.len = q.len; r.iv = (u32)(j - M.ivs[s]) | ((u32)s << 8);
      // the order of sel_wave_sort: (tid, hit position, input order) for one interval, (tid, reference end, query end, input order) for several
      if (M.ivcnt[s] == 1) { M.k.k1[l] = ((u64)r.tid << 32) | (u64)(((u32)(r.pos - r.qpos)) ^ 0x80000000u); M.k.k2[l] = (u64)l; }
      else { M.k.k1[l] = ((u64)r.tid << 32) | (u64)(u32)(r.pos + r.len); M.k.k2[l] = ((u64)(u32)(r.qpos + r.len) << 16) | (u64)l; }
      mine[l] = r; myslot[l] = s;
    }
  }
  QM_LANES(l) { if (l == 0) { M.k.k1[64] = ~0ULL; M.k.k2[64] = ~0ULL; } }     // (the chaining arrays of the previous batch lie over it)
  wave_fence();
  QM_T(1);
  // ---- rank sort inside every read's segment
  LV<int> rin;
  QM_LANES(l) { rin[l] = l < R ? rcnt[l] : 0; }
  const int maxn = wave_max(rin);                         // the largest segment of the batch
  LV<int> rank;
  QM_LANES(l) {
    int rk = 0;
    if (l < NR) {
      const int s = myslot[l]; const int b = M.rb[s], n = M.rn[s];
      const u64 a1 = M.k.k1[l], a2 = M.k.k2[l];
      // four trips at a time, their LDS reads requested before the first compare; past the lane's segment the sentinel (no branch, no t < n test)
      for (int t = 0; t < maxn; t += 4) {
        u64 c1[4], c2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = t + u < n ? b + t + u : 64; c1[u] = M.k.k1[j]; c2[u] = M.k.k2[j]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) key_count(c1[u], c2[u], a1, a2, rk);
      }
      rk += b;
    }
    rank[l] = rk;
  }
  wave_fence();
  QM_LANES(l) { if (l < NR) M.rec[rank[l]] = mine[l]; }
  wave_fence();
  QM_T(2);
  // ---- groups: the lane of a transcript's first record of a read
  LV<SelRec> rr; LV<bool> head; LV<int> sl;
  QM_LANES(l) {
    head[l] = false; sl[l] = 0;
    if (l < NR) {
      const SelRec x = M.rec[l]; rr[l] = x;
      const int s = (int)(x.iv >> 8); sl[l] = s;
      head[l] = l == M.rb[s] || M.rec[l - 1].tid != x.tid;
      if (M.ivcnt[s] == 1) M.ends[l] = (int)(x.pos - x.qpos);      // one interval: every occurrence is a position (HitManager.cpp:716-807)
    }
  }
  const u64 hm = ballot(head);
  // the set of a transcript's intervals (a 64-bit mask): every record ORs its bit into the word of its group's first lane (the sort keys' array
  // is free by now) -- a loop of the first lane over its records was a chain of LDS round trips as long as the largest group
  LV<int> hidx;
  QM_LANES(l) { hidx[l] = head[l] ? l : 0; if (head[l]) M.k.k1[l] = 0; }
  lane_scan_max(hidx);
  wave_fence();
  QM_LANES(l) { if (l < NR && M.ivcnt[sl[l]] > 1) atomic_or_u64(&M.k.k1[hidx[l]], 1ULL << (rr[l].iv & 63u)); }
  wave_fence();
  LV<int> g1v, reqN; LV<bool> req;
  QM_LANES(l) {
    g1v[l] = 0; req[l] = false; reqN[l] = 0;
    if (head[l]) {
      const u64 rest = l < 63 ? (hm & ~lanemask_lt(l + 1)) : 0ULL;
      const int g1 = rest ? ctz64(rest) : NR;               // (the next read's first record is a head too)
      g1v[l] = g1;
      const int m = M.ivcnt[sl[l]];
      if (m > 1) {
        // intersectSAHits (HitManager.cpp:587-689) reduces to counting the distinct intervals of a transcript (see sel_strand)
        const float requiredFrac = (float)m * B.consensus_fraction;
        int requiredNumHits = m;
        if (B.consensus_fraction < 1.0) { const int fl = (int)requiredFrac; requiredNumHits = fl > 1 ? fl : 1; }
        const u64 mk = M.k.k1[l];
        req[l] = popc64(mk) >= requiredNumHits;
        reqN[l] = requiredNumHits;
      }
    }
  }
  const u64 reqm = ballot(req);
  QM_T(3);
  LV<int> nsv, chainN, chainLen; LV<bool> em; LV<SelGroup> gv;
  QM_LANES(l) {
    nsv[l] = 0; em[l] = false; chainN[l] = 0; chainLen[l] = 0;
    if (head[l]) {
      const int s = sl[l]; const int m = M.ivcnt[s];
      const int hn = g1v[l] - l;
      const u32 readLen = (u32)M.rlen[s];
      if (m == 1) {
        const SelRec x = rr[l];
        SelGroup g; g.tid = x.tid; g.offcs = 0; g.set_cs(x.len == readLen ? QM_CS_PERFECT : QM_CS_REGULAR); g.score = -1.7976931348623157e308;
        g.npos = hn; g.ppos = (int)(x.pos - x.qpos);
        gv[l] = g; nsv[l] = hn; em[l] = true;
      } else {
        const int b = M.rb[s], n = M.rn[s];
        const u64 range = lanemask_lt(b + n) & ~lanemask_lt(b);
        const bool allActive = (m - reqN[l]) > 0 && (reqm & range) == 0;          // HitManager.cpp:682-686
        if (req[l] || allActive) { chainN[l] = hn; chainLen[l] = (int)readLen; }
      }
    }
  }
  // chaining, lane-parallel for the groups whose chain is hit 0 <- hit 1 <- ... on one diagonal (see sel_pack_batch_wide for the argument):
  // f is a prefix sum over the record lanes, every record checks its own conditions, a failed check leaves the group to the editions below
  {
    LV<int> Sv;
    QM_LANES(l) {
      int g = 0;
      if (l < NR && M.ivcnt[sl[l]] > 1) {
        const SelRec x = rr[l];
        g = (int)x.len;
        if (!head[l]) { const SelRec y = M.rec[l - 1]; const int dq = (int)(x.qpos + x.len) - (int)(y.qpos + y.len); g = g < dq ? g : dq; }
      }
      Sv[l] = g;
      if (head[l]) M.k.k1[l] = 0;                           // (the interval set has been read: the word now collects failed checks)
    }
    lane_scan_add(Sv);
    QM_LANES(l) { M.sw[l] = Sv[l]; }
    wave_fence();
    QM_LANES(l) {
      if (l < NR && M.ivcnt[sl[l]] > 1 && !head[l]) {
        const int h = hidx[l];
        const int base = h > 0 ? M.sw[h - 1] : 0;
        const int F = Sv[l] - base;
        const SelRec x = rr[l]; const SelRec hd = M.rec[h];
        int ok = (int)((int)(x.pos - x.qpos) == (int)(hd.pos - hd.qpos)) & (int)(F > (int)x.len);
        if (l >= h + 2) {
          const SelRec y2 = M.rec[l - 2];
          const int dq2 = (int)(x.qpos + x.len) - (int)(y2.qpos + y2.len);
          const int cand2 = (M.sw[l - 2] - base) + ((int)x.len < dq2 ? (int)x.len : dq2);
          ok &= (int)(cand2 <= F);
        }
        if (!ok) atomic_or_u64(&M.k.k1[h], 1ULL);
      }
    }
    wave_fence();
    QM_LANES(l) {
      const int hn = chainN[l];
      if (hn > 0 && M.k.k1[l] == 0) {
        const int base = l > 0 ? M.sw[l - 1] : 0;
        const int Flast = M.sw[l + hn - 1] - base;
        const SelRec x = rr[l];
        if (hn == 1 || Flast > (int)x.len) {
          SelGroup g; g.tid = x.tid; g.offcs = 0; g.set_cs(QM_CS_REGULAR); g.score = 0; g.npos = 1; g.ppos = (int)(x.pos - x.qpos);
          if (hn > 1) {
            const int Fprev = M.sw[l + hn - 2] - base; const SelRec hl = M.rec[l + hn - 1];
            if (Flast > Fprev && (int)(hl.qpos + hl.len) - (int)x.qpos == chainLen[l]) g.set_cs(QM_CS_UNGAPPED);
          }
          M.ends[l] = g.ppos;
          gv[l] = g; nsv[l] = 1; em[l] = true; chainN[l] = 0;
        }
      }
    }
    wave_fence();
  }
  // chaining (HitManager.cpp:107-307).  Nearly every transcript's hits lie on ONE diagonal; up to eight of them go through an edition of the
  // DP that lives in registers -- on one diagonal the gap cost is zero and every score a small integer, so integer arithmetic decides every
  // comparison as the doubles do --; anything else takes sel_chain_group, and the wavefront only enters that when some lane needs it
  LV<bool> slow;
  QM_LANES(l) {
    slow[l] = false;
    if (chainN[l] > 0) {
      SelGroup g;
      const int ns = chainN[l] <= 8 ? sel_chain_diag8(M.rec + l, chainN[l], chainLen[l], g, M.ends + l)
                                    : sel_chain_diag_mem(M.rec + l, chainN[l], chainLen[l], (int*)(M.c.f + l), M.c.p + l, M.c.seen + l, M.ends + l, g, M.ends + l);
      if (ns > 0) { gv[l] = g; nsv[l] = ns; em[l] = true; } else slow[l] = true;
    }
  }
  if (ballot(slow)) {
    QM_LANES(l) {
      if (slow[l]) {
        SelGroup g;
        const int ns = sel_chain_group(M.rec + l, chainN[l], M.c.f + l, M.c.p + l, M.c.seen + l, M.ends + l, M.starts + l, chainLen[l], g, M.ends + l);
        if (ns > 0) { gv[l] = g; nsv[l] = ns; em[l] = true; }
      }
    }
  }
  wave_fence();
  QM_T(4);
  // ---- the batch's words: header, own position, positions of every group (mergeOrientationUnique with an empty other side)
  LV<int> wv, ws;
  QM_LANES(l) { wv[l] = em[l] ? 2 + nsv[l] : 0; ws[l] = wv[l]; }
  lane_scan_add(ws);
  QM_LANES(l) { M.sw[l] = ws[l]; }
  const int W = read_lane(ws, 63);
  long long base = 0; bool fits = true;
  if (W > 0) {
    if (wa.base < 0 || wa.used + W > QM_CHUNK) {
      LV<u64> bv;
      QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor, (u64)QM_CHUNK); }
      wa.base = (long long)read_lane(bv, 0); wa.used = 0;
    }
    base = wa.base + wa.used;
    if (base + W > B.lists_cap) { QM_LANES(l) { if (l == 0) *B.status |= 1; } fits = false; }
    else wa.used += W;
  }
  QM_LANES(l) {
    if (em[l]) {
      const int o = ws[l] - wv[l];
      const SelGroup g = gv[l];
      M.out[o] = sel_header(g.tid, M.rrc[sl[l]] != 0, g.cs(), nsv[l]);
      M.out[o + 1] = (u64)(u32)g.ppos;
      for (int t = 0; t < nsv[l]; ++t) M.out[o + 2 + t] = (u64)(u32)M.ends[l + t];
    }
  }
  wave_fence();
  QM_T(5);
  QM_LANES(l) {
    if (l < R && pk[l]) {
      const int n = rcnt[l], b = rscan[l] - rcnt[l];
      int nw = 0, wb = 0;
      if (n > 0) { wb = b > 0 ? M.sw[b - 1] : 0; nw = M.sw[b + n - 1] - wb; }
      if (!fits) nw = 0;
      B.lcnt[r0 + l] = (u32)nw | fflag[l];
      B.loff[r0 + l] = nw > 0 ? base + wb : 0;
    }
  }
  if (fits) { for (int b0 = 0; b0 < W; b0 += 64) { QM_LANES(l) { if (b0 + l < W) B.lists[base + b0 + l] = M.out[b0 + l]; } } }
  wave_fence();
  QM_T(6);
  return R;
}

// sel_chain_group for hn hits of ONE diagonal, any hn: the DP of sel_chain_diag8 with its arrays in memory (f as integers).  Returns 0
// when the hits are not on one diagonal.  posOut may alias ends.
QM_DEV int sel_chain_diag_mem(const SelRec* H, int hn, int maxDist, int* f, int* p, int* seen, int* ends, SelGroup& g, int* posOut) {
  // The chain of a read's hits on one transcript is nearly always hit i-1 <- hit i: f and the query end of the two previous hits stay in
  // registers (the look-back stops two looks after the first predecessor), the next record is requested before this one's stores, the
  // one-diagonal test rides along (a return of 0 midway leaves scratch that sel_chain_group overwrites), and a chain that is linear from an
  // end beyond hit 0 has one start without the walk back (every later end runs into the first one's marks).
  const SelRec h0 = H[0];
  const int diag = (int)(h0.pos - h0.qpos);
  int best = -1, lastBest = -1, nEnds = 0, firstEnd = 0;
  int f1 = 0, e1 = 0, f2 = 0, e2 = 0;
  bool linear = true;
  SelRec hi = h0, hnx = H[1 < hn ? 1 : 0];                 // (records are requested two hits ahead: an LDS round trip is longer than a hit's arithmetic)
  for (int i = 0; i < hn; ++i) {
    const SelRec hnx2 = H[i + 2 < hn ? i + 2 : hn - 1];
    if ((int)(hi.pos - hi.qpos) != diag) return 0;
    const int ei = (int)(hi.qpos + hi.len), leni = (int)hi.len;
    int fi = leni, pi = i, looksLeft = 2;
    bool done = false;
    if (i >= 1) {
      const int dq = ei - e1;
      const int cand = f1 + (leni < dq ? leni : dq);
      const bool take = cand > fi;
      pi = take ? i - 1 : pi; fi = take ? cand : fi;
      if (pi < i) --looksLeft;
    }
    if (i >= 2) {
      const int dq = ei - e2;
      const int cand = f2 + (leni < dq ? leni : dq);
      const bool take = cand > fi;
      pi = take ? i - 2 : pi; fi = take ? cand : fi;
      if (pi < i) { --looksLeft; done = looksLeft <= 0; }
    }
    if (!done) {
      for (int j = i - 3; j >= 0; --j) {
        const int dq = ei - (int)(H[j].qpos + H[j].len);
        const int cand = f[j] + (leni < dq ? leni : dq);
        const bool take = cand > fi;
        pi = take ? j : pi; fi = take ? cand : fi;
        if (pi < i) { --looksLeft; if (looksLeft <= 0) break; }
      }
    }
    p[i] = pi; f[i] = fi;
    linear = linear && (i == 0 || pi == i - 1);
    if (fi > best) { best = fi; lastBest = i; nEnds = 0; ends[nEnds++] = i; firstEnd = i; }
    else if (fi == best) ends[nEnds++] = i;
    f2 = f1; e2 = e1; f1 = fi; e1 = ei; hi = hnx; hnx = hnx2;
  }
  int nOptimal = 0, nStarts = 0;
  if (linear && firstEnd > 0) { nOptimal = 1; nStarts = 1; }
  else {
    for (int i = 0; i < hn; ++i) seen[i] = 0;
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
      if (fresh) { ++nOptimal; ++nStarts; }
    }
  }
  for (int t = 0; t < nStarts; ++t) posOut[t] = diag;
  g.tid = h0.tid; g.offcs = 0; g.set_cs(QM_CS_REGULAR); g.score = (double)best; g.npos = nStarts; g.ppos = diag;
  if (hn > 1 && nOptimal == 1 && lastBest == hn - 1) {
    const SelRec hl = H[hn - 1];
    if ((long long)(hl.qpos + hl.len) - (long long)h0.qpos == (long long)maxDist) g.set_cs(QM_CS_UNGAPPED);
  }
  return nStarts;
}

// ------------------------------------------------------------------ the wide edition: 64 * C intervals / suffixes per batch
// Reads of 150 and 250 bp bring 45 .. 90 suffixes per strand (an interval every maxMMPExtension + 1 positions, a suffix per isoform in
// each): they do not fit the 64 lanes above and went to the one-read kernel's device-memory scratch -- 725 ms per 4 M reads of 250 bp.
// Here a lane owns C records (i = 64 c + l), the scans carry from chunk to chunk, a read still has at most 64 intervals (their set is a
// 64-bit mask) and hits on one strand.  It runs over the queue the narrow kernel leaves (ids), and leaves a queue of its own.
#define QM_SC_TODO2 29
template <int V> struct IntC { static constexpr int value = V; };     // a compile-time int as a lambda's argument
template <int C>
struct PackMemW {
  static constexpr int N = 64 * C;
  long long ivoff[QM_PK_READS]; long long rid[QM_PK_READS];
  int ivcnt[QM_PK_READS], rlen[QM_PK_READS], ivs[QM_PK_READS], rb[QM_PK_READS], rn[QM_PK_READS], rrc[QM_PK_READS];
  int hasF[QM_PK_READS], hasR[QM_PK_READS], anyreq[QM_PK_READS];
  union {
    struct { IntRec iv[N]; int slot[N]; int rs[N]; } a;
    u64 out[3 * N];
  };
  int mark[N];
  SelRec rec[N];
  union {
    struct { u64 k1[N + 1], k2[N + 1]; } k;                 // (+1: the rank sort's sentinel, a key below nothing)
    struct { double f[N]; int p[N], seen[N]; } c;
  };
  int ends[N], starts[N];
  int sw[N];
};                                 // (19.8 KB for C = 4: two blocks of four waves per CU -- one more array and it is one)
template <int C> QM_DEV void scan_add_c(LV<int> (&x)[C]) {
  int carry = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) { lane_scan_add(x[c]); QM_LANES(l) { x[c][l] += carry; } carry = read_lane(x[c], 63); }
}
template <int C> QM_DEV void scan_max_c(LV<int> (&x)[C]) {
  int carry = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) { lane_scan_max(x[c]); QM_LANES(l) { x[c][l] = x[c][l] > carry ? x[c][l] : carry; } carry = read_lane(x[c], 63); }
}

// One batch of the queue slots q0 .. of the wave's range [q0, qEnd): read ids[q].  Returns how many slots it consumed (>= 1).
template <int C>
QM_DEV int sel_pack_batch_wide(const DevIndex& ix, const ReadBatch& B, const long long* ids, long long q0, long long qEnd, PackMemW<C>& M,
                               WaveAlloc& wa, long long* todoq) {
#pragma clang fp contract(off)
  constexpr int N = 64 * C;
  const bool paired = B.seq2 != nullptr;
  LV<int> cnt, scan; LV<u32> fflag; LV<bool> big; LV<long long> rdv;
  QM_LANES(l) {
    int c = 0; u32 ff = 0; bool bg = false; long long read = 0;
    if (l < QM_PK_READS && q0 + l < qEnd) {
      read = ids ? ids[q0 + l] : q0 + l;                     // (no queue: every read of the batch, in order)
      c = (int)B.iv_in_cnt[read];
      M.ivoff[l] = B.iv_in_off[read];
      const int mate = paired ? (int)(read & 1) : 0; const long long unit = paired ? (read >> 1) : read;
      const long long* off = mate ? B.off2 : B.off1;
      M.rlen[l] = (int)(off[unit + 1] - off[unit]);
      bg = c > 64;                                          // the interval set of a transcript is a 64-bit mask: such a read is handed on
      if (bg) c = 0;
      M.ivcnt[l] = c; M.hasF[l] = 0; M.hasR[l] = 0; M.anyreq[l] = 0;
      ff = (B.found_in && B.found_in[read]) ? 0x80000000u : 0u;
    }
    cnt[l] = c; scan[l] = c; fflag[l] = ff; big[l] = bg; rdv[l] = read;
  }
  lane_scan_add(scan);
  LV<bool> ok;
  QM_LANES(l) { ok[l] = l < QM_PK_READS && q0 + l < qEnd && scan[l] <= N; }
  int R = ctz64(~ballot(ok));                             // >= 1: a read's (counted) intervals are at most 64
  const int NI = read_lane(scan, R - 1);
  QM_LANES(l) { if (l < QM_PK_READS) M.ivs[l] = scan[l] - cnt[l]; }
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { M.mark[64 * c + l] = 0; } }
  wave_fence();
  QM_LANES(l) { if (l < R && cnt[l] > 0) M.mark[scan[l] - cnt[l]] = l; }
  wave_fence();
  LV<int> slot[C], w[C], wsc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { slot[c][l] = M.mark[64 * c + l]; } }
  scan_max_c<C>(slot);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      const int i = 64 * c + l;
      int ww = 0;
      if (i < NI) {
        const int s = slot[c][l];
        const qm_sa_interval_hit h = B.iv_in[M.ivoff[s] + (long long)(i - M.ivs[s])];
        IntRec q; q.b = (u32)h.begin; q.e = (u32)h.end; q.len = h.len; q.q = h.query_pos;
        M.a.iv[i] = q; M.a.slot[i] = s;
        ww = (int)(q.e - q.b);
        if (h.query_rc != 0) M.hasR[s] = 1; else M.hasF[s] = 1;
      }
      w[c][l] = ww; wsc[c][l] = ww;
    }
  }
  scan_add_c<C>(wsc);
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { M.sw[64 * c + l] = wsc[c][l]; M.mark[64 * c + l] = 0; } }
  wave_fence();
  LV<int> rcnt, rscan; LV<bool> pk;
  QM_LANES(l) {
    int n = 0; bool p = false;
    if (l < R) {
      const int c = cnt[l], s = scan[l] - c;
      p = !big[l];
      int rc = 0;
      if (c > 0) {
        n = M.sw[s + c - 1] - (s > 0 ? M.sw[s - 1] : 0);
        const bool hf = M.hasF[l] != 0, hr = M.hasR[l] != 0;
        p = !(hf && hr) && n <= N;
        rc = hr ? 1 : 0;
      }
      if (!p) n = 0;
      M.rrc[l] = rc;
    }
    rcnt[l] = n; rscan[l] = n; pk[l] = p;
  }
  lane_scan_add(rscan);
  LV<bool> fit;
  QM_LANES(l) { fit[l] = l < R && rscan[l] <= N; }
  R = ctz64(~ballot(fit));
  const int NR = read_lane(rscan, R - 1);
  QM_LANES(l) { if (l < R) { M.rb[l] = rscan[l] - rcnt[l]; M.rn[l] = pk[l] ? rcnt[l] : -1; } }
  LV<bool> td;
  QM_LANES(l) { td[l] = l < R && !pk[l]; }
  const u64 tdm = ballot(td);
  if (tdm) {
    LV<u64> qb;
    QM_LANES(l) { qb[l] = 0; if (l == 0) qb[l] = atomic_add_u64(B.cursor + QM_SC_TODO2, (u64)popc64(tdm)); }
    const u64 qq = read_lane(qb, 0);
    QM_LANES(l) { if (td[l]) todoq[qq + (u64)popc64(tdm & lanemask_lt(l))] = rdv[l]; }
  }
  wave_fence();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      const int i = 64 * c + l;
      if (i < NI) {
        const int s = slot[c][l];
        if (s < R && M.rn[s] > 0 && w[c][l] > 0) {
          const int first = M.ivs[s];
          const int rs = M.rb[s] + (wsc[c][l] - w[c][l]) - (first > 0 ? M.sw[first - 1] : 0);
          M.a.rs[i] = rs; M.mark[rs] = i;
        }
      }
    }
  }
  wave_fence();
  QM_T(0);
  LV<int> ivl[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { ivl[c][l] = M.mark[64 * c + l]; } }
  scan_max_c<C>(ivl);
  LV<SelRec> mine[C]; LV<int> myslot[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      const int i = 64 * c + l;
      myslot[c][l] = 0;
      if (i < NR) {
        const int j = ivl[c][l];
        const IntRec q = M.a.iv[j]; const int s = M.a.slot[j];
        const SaInfo e = ix.sainfo[q.b + (u32)(i - M.a.rs[j])];
        SelRec r; r.tid = e.tid; r.pos = (u32)e.pos; r.qpos = q.q; r.len = q.len; r.iv = (u32)(j - M.ivs[s]) | ((u32)s << 8);
        if (M.ivcnt[s] == 1) { M.k.k1[i] = ((u64)r.tid << 32) | (u64)(((u32)(r.pos - r.qpos)) ^ 0x80000000u); M.k.k2[i] = (u64)i; }
        else { M.k.k1[i] = ((u64)r.tid << 32) | (u64)(u32)(r.pos + r.len); M.k.k2[i] = ((u64)(u32)(r.qpos + r.len) << 16) | (u64)i; }
        mine[c][l] = r; myslot[c][l] = s;
      }
    }
  }
  QM_LANES(l) { if (l == 0) { M.k.k1[N] = ~0ULL; M.k.k2[N] = ~0ULL; } }    // (the chaining arrays of the previous batch lie over it)
  wave_fence();
  QM_T(1);
  LV<int> rin;
  QM_LANES(l) { rin[l] = l < R ? rcnt[l] : 0; }
  const int maxn = wave_max(rin);
  // the trip over a segment's records is the OUTER loop and the chunks are inside it, no branch around the loads (clamped index, select):
  // a trip's 2 C LDS reads are in flight together -- with a loop per chunk every trip waited for its own two (57 % of the kernel on 2 x 250 bp)
  LV<int> rank[C], sb[C], sn[C]; LV<u64> ka[C], kb[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      const int i = 64 * c + l;
      rank[c][l] = 0; sb[c][l] = 0; sn[c][l] = 0; ka[c][l] = 0; kb[c][l] = 0;
      if (i < NR) { const int s = myslot[c][l]; sb[c][l] = M.rb[s]; sn[c][l] = M.rn[s]; ka[c][l] = M.k.k1[i]; kb[c][l] = M.k.k2[i]; }
    }
  }
  // (one straight-line edition per number of chunks in use: a test per chunk inside the trip would put a branch -- and a wait -- between the chunks' reads)
  auto trips = [&](auto ca) {
    constexpr int CA = decltype(ca)::value;
    for (int t = 0; t < maxn; t += 2) {                     // two trips at a time (t + 1 == maxn: no lane has that many), every read of the
      LV<u64> x1[CA], x2[CA], y1[CA], y2[CA];               // pair requested before the first compare
#pragma unroll
      for (int c = 0; c < CA; ++c) {
        QM_LANES(l) {
          const int n = sn[c][l]; const int j0 = sb[c][l] + t;
          const int ja = t < n ? j0 : N; const int jb = t + 1 < n ? j0 + 1 : N;          // (past the lane's segment: the sentinel)
          x1[c][l] = M.k.k1[ja]; x2[c][l] = M.k.k2[ja]; y1[c][l] = M.k.k1[jb]; y2[c][l] = M.k.k2[jb];
        }
      }
#pragma unroll
      for (int c = 0; c < CA; ++c) {
        QM_LANES(l) {
          const u64 a1 = ka[c][l], a2 = kb[c][l];
          key_count(x1[c][l], x2[c][l], a1, a2, rank[c][l]);
          key_count(y1[c][l], y2[c][l], a1, a2, rank[c][l]);
        }
      }
    }
  };
  {
    const int CAn = (NR + 63) >> 6;
    if (CAn <= 1) trips(IntC<1>{});
    else if (CAn == 2) trips(IntC<(C >= 2 ? 2 : C)>{});
    else if (CAn == 3) trips(IntC<(C >= 3 ? 3 : C)>{});
    else trips(IntC<C>{});
  }
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { rank[c][l] += sb[c][l]; } }
  wave_fence();
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { if (64 * c + l < NR) M.rec[rank[c][l]] = mine[c][l]; } }
  wave_fence();
  QM_T(2);
  LV<SelRec> rr[C]; LV<bool> head[C]; LV<int> sl[C];
  u64 hm[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      const int i = 64 * c + l;
      head[c][l] = false; sl[c][l] = 0;
      if (i < NR) {
        const SelRec x = M.rec[i]; rr[c][l] = x;
        const int s = (int)(x.iv >> 8); sl[c][l] = s;
        head[c][l] = i == M.rb[s] || M.rec[i - 1].tid != x.tid;
        if (M.ivcnt[s] == 1) M.ends[i] = (int)(x.pos - x.qpos);
      }
    }
    hm[c] = ballot(head[c]);
  }
  // (the interval sets: as in the narrow kernel, every record ORs its bit into the word of its group's first record)
  LV<int> hidx[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { hidx[c][l] = head[c][l] ? 64 * c + l : 0; if (head[c][l]) M.k.k1[64 * c + l] = 0; } }
  scan_max_c<C>(hidx);
  wave_fence();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) { if (64 * c + l < NR && M.ivcnt[sl[c][l]] > 1) atomic_or_u64(&M.k.k1[hidx[c][l]], 1ULL << (rr[c][l].iv & 63u)); }
  }
  wave_fence();
  LV<int> g1v[C], reqN[C]; LV<bool> req[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      g1v[c][l] = 0; req[c][l] = false; reqN[c][l] = 0;
      if (head[c][l]) {
        const int i = 64 * c + l;
        int g1 = NR;
        const u64 rest = l < 63 ? (hm[c] & ~lanemask_lt(l + 1)) : 0ULL;
        if (rest) g1 = 64 * c + ctz64(rest);
        else {
#pragma unroll
          for (int d = C - 1; d > 0; --d) if (d > c && hm[d]) g1 = 64 * d + ctz64(hm[d]);   // the lowest such d wins
        }
        g1v[c][l] = g1;
        const int s = sl[c][l];
        const int m = M.ivcnt[s];
        if (m > 1) {
          const float requiredFrac = (float)m * B.consensus_fraction;
          int requiredNumHits = m;
          if (B.consensus_fraction < 1.0) { const int fl = (int)requiredFrac; requiredNumHits = fl > 1 ? fl : 1; }
          const u64 mk = M.k.k1[i];
          const bool rq = popc64(mk) >= requiredNumHits;
          req[c][l] = rq; reqN[c][l] = requiredNumHits;
          if (rq) M.anyreq[s] = 1;
        }
      }
    }
  }
  wave_fence();
  LV<int> nsv[C], chainN[C], chainLen[C]; LV<bool> em[C]; LV<SelGroup> gv[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      nsv[c][l] = 0; em[c][l] = false; chainN[c][l] = 0; chainLen[c][l] = 0;
      if (head[c][l]) {
        const int i = 64 * c + l;
        const int s = sl[c][l]; const int m = M.ivcnt[s];
        const int hn = g1v[c][l] - i;
        const u32 readLen = (u32)M.rlen[s];
        if (m == 1) {
          const SelRec x = rr[c][l];
          SelGroup g; g.tid = x.tid; g.offcs = 0; g.set_cs(x.len == readLen ? QM_CS_PERFECT : QM_CS_REGULAR); g.score = -1.7976931348623157e308;
          g.npos = hn; g.ppos = (int)(x.pos - x.qpos);
          gv[c][l] = g; nsv[c][l] = hn; em[c][l] = true;
        } else {
          const bool allActive = (m - reqN[c][l]) > 0 && M.anyreq[s] == 0;
          if (req[c][l] || allActive) { chainN[c][l] = hn; chainLen[c][l] = (int)readLen; }
        }
      }
    }
  }
  QM_T(3);
  // Chaining, lane-parallel for the usual group: hits on one diagonal whose chain is hit 0 <- hit 1 <- ... (HitManager.cpp:107-307 with
  // p[i] = i - 1).  Under that hypothesis f[i] = len_0 + sum of min(len_t, e_t - e_(t-1)) -- a prefix sum over the record lanes --, and the
  // hypothesis holds exactly when every hit takes its predecessor (f[i-1] + gain > len_i) and does not prefer the hit before it
  // (f[i-2] + min(len_i, e_i - e_(i-2)) <= f[i]): the look-back stops there (two looks after the first predecessor).  Every record checks
  // its own two conditions and the diagonal; a group with a failed check (or with no gain at all: its first end would be hit 0) keeps
  // its job for the serial editions below.  A linear chain from an end beyond hit 0 has ONE start (sel_chain_diag_mem).
  {
    LV<int> gn[C], Sv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      QM_LANES(l) {
        const int i = 64 * c + l;
        int g = 0;
        if (i < NR && M.ivcnt[sl[c][l]] > 1) {
          const SelRec x = rr[c][l];
          g = (int)x.len;
          if (!head[c][l]) { const SelRec y = M.rec[i - 1]; const int dq = (int)(x.qpos + x.len) - (int)(y.qpos + y.len); g = g < dq ? g : dq; }
        }
        gn[c][l] = g; Sv[c][l] = g;
        if (head[c][l]) M.k.k1[i] = 0;                      // (the group's interval set has been read: the word now collects failed checks)
      }
    }
    scan_add_c<C>(Sv);
#pragma unroll
    for (int c = 0; c < C; ++c) { QM_LANES(l) { M.sw[64 * c + l] = Sv[c][l]; } }
    wave_fence();
#pragma unroll
    for (int c = 0; c < C; ++c) {
      QM_LANES(l) {
        const int i = 64 * c + l;
        if (i < NR && M.ivcnt[sl[c][l]] > 1 && !head[c][l]) {
          const int h = hidx[c][l];
          const int base = h > 0 ? M.sw[h - 1] : 0;
          const int F = Sv[c][l] - base;
          const SelRec x = rr[c][l]; const SelRec hd = M.rec[h];
          int ok = (int)((int)(x.pos - x.qpos) == (int)(hd.pos - hd.qpos)) & (int)(F > (int)x.len);
          if (i >= h + 2) {
            const SelRec y2 = M.rec[i - 2];
            const int dq2 = (int)(x.qpos + x.len) - (int)(y2.qpos + y2.len);
            const int cand2 = (M.sw[i - 2] - base) + ((int)x.len < dq2 ? (int)x.len : dq2);
            ok &= (int)(cand2 <= F);
          }
          if (!ok) atomic_or_u64(&M.k.k1[h], 1ULL);
        }
      }
    }
    wave_fence();
#pragma unroll
    for (int c = 0; c < C; ++c) {
      QM_LANES(l) {
        const int hn = chainN[c][l];
        if (hn > 0) {
          const int i = 64 * c + l;
          if (M.k.k1[i] == 0) {
            const int base = i > 0 ? M.sw[i - 1] : 0;
            const int Flast = M.sw[i + hn - 1] - base;
            const SelRec x = rr[c][l];
            if (hn == 1 || Flast > (int)x.len) {
              SelGroup g; g.offcs = 0; g.set_cs(QM_CS_REGULAR);
              if (hn > 1) {
                const int Fprev = M.sw[i + hn - 2] - base; const SelRec hl = M.rec[i + hn - 1];
                if (Flast > Fprev && (int)(hl.qpos + hl.len) - (int)x.qpos == chainLen[c][l]) g.set_cs(QM_CS_UNGAPPED);
              }
              const int diag = (int)(x.pos - x.qpos);
              M.starts[i] = 1 | (g.cs() << 16); M.c.seen[i] = diag; M.ends[i] = diag;
              chainN[c][l] = -hn;                           // done: no job
            }
          }
        }
      }
    }
    wave_fence();
  }
  QM_T(7);                                                // (the lane-parallel chaining; 4: the serial jobs that are left)
  // chaining: the groups that need it are spread over the chunks' lanes; gathered into one list, a lane per job, they run side by side in as
  // few divergent rounds as there are 64 jobs -- results through LDS: starts[first] = chain starts | status << 16, seen[first] = own position
  int njobs = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    LV<bool> hj;
    QM_LANES(l) { hj[l] = chainN[c][l] > 0; }
    const u64 jm = ballot(hj);
    QM_LANES(l) { if (hj[l]) M.mark[njobs + popc64(jm & lanemask_lt(l))] = (64 * c + l) | (chainN[c][l] << 16); }   // (mark: free since the suffix gather)
    njobs += popc64(jm);
  }
  wave_fence();
  for (int j0 = 0; j0 < njobs; j0 += 64) {
    LV<bool> slow; LV<int> ji, jn, jl;
    QM_LANES(l) {
      slow[l] = false; ji[l] = 0; jn[l] = 0; jl[l] = 0;
      if (j0 + l < njobs) {
        const int jb = M.mark[j0 + l];
        const int i = jb & 0xffff, hn = jb >> 16;
        const int readLen = M.rlen[M.rec[i].iv >> 8];
        ji[l] = i; jn[l] = hn; jl[l] = readLen;
        SelGroup g;
        int ns = hn <= 8 ? sel_chain_diag8(M.rec + i, hn, readLen, g, M.ends + i)
                         : sel_chain_diag_mem(M.rec + i, hn, readLen, (int*)(M.c.f + i), M.c.p + i, M.c.seen + i, M.ends + i, g, M.ends + i);
        if (ns > 0) { M.starts[i] = ns | (g.cs() << 16); M.c.seen[i] = g.ppos; } else slow[l] = true;
      }
    }
    if (ballot(slow)) {
      QM_LANES(l) {
        if (slow[l]) {
          const int i = ji[l];
          SelGroup g; g.offcs = 0; g.ppos = 0;
          const int ns = sel_chain_group(M.rec + i, jn[l], M.c.f + i, M.c.p + i, M.c.seen + i, M.ends + i, M.starts + i, jl[l], g, M.ends + i);
          M.starts[i] = ns > 0 ? (ns | (g.cs() << 16)) : 0;
          M.c.seen[i] = g.ppos;
        }
      }
    }
  }
  wave_fence();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      if (chainN[c][l] != 0) {
        const int i = 64 * c + l;
        const int rs = M.starts[i];
        const int ns = rs & 0xffff;
        if (ns > 0) {
          SelGroup g; g.tid = rr[c][l].tid; g.offcs = 0; g.set_cs(rs >> 16); g.score = 0; g.npos = ns; g.ppos = M.c.seen[i];
          gv[c][l] = g; nsv[c][l] = ns; em[c][l] = true;
        }
      }
    }
  }
  wave_fence();
  QM_T(4);
  LV<int> wv[C], ws[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { wv[c][l] = em[c][l] ? 2 + nsv[c][l] : 0; ws[c][l] = wv[c][l]; } }
  scan_add_c<C>(ws);
#pragma unroll
  for (int c = 0; c < C; ++c) { QM_LANES(l) { M.sw[64 * c + l] = ws[c][l]; } }
  const int W = read_lane(ws[C - 1], 63);
  long long base = 0; bool fits = true;
  if (W > 0) {
    if (wa.base < 0 || wa.used + W > QM_CHUNK) {
      LV<u64> bv;
      QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor, (u64)QM_CHUNK); }
      wa.base = (long long)read_lane(bv, 0); wa.used = 0;
    }
    base = wa.base + wa.used;
    if (base + W > B.lists_cap) { QM_LANES(l) { if (l == 0) *B.status |= 1; } fits = false; }
    else wa.used += W;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    QM_LANES(l) {
      if (em[c][l]) {
        const int i = 64 * c + l;
        const int o = ws[c][l] - wv[c][l];
        const SelGroup g = gv[c][l];
        M.out[o] = sel_header(g.tid, M.rrc[sl[c][l]] != 0, g.cs(), nsv[c][l]);
        M.out[o + 1] = (u64)(u32)g.ppos;
        for (int t = 0; t < nsv[c][l]; ++t) M.out[o + 2 + t] = (u64)(u32)M.ends[i + t];
      }
    }
  }
  wave_fence();
  QM_T(5);
  QM_LANES(l) {
    if (l < R && pk[l]) {
      const int n = rcnt[l], b = rscan[l] - rcnt[l];
      int nw = 0, wb = 0;
      if (n > 0) { wb = b > 0 ? M.sw[b - 1] : 0; nw = M.sw[b + n - 1] - wb; }
      if (!fits) nw = 0;
      B.lcnt[rdv[l]] = (u32)nw | fflag[l];
      B.loff[rdv[l]] = nw > 0 ? base + wb : 0;
    }
  }
  if (fits) { for (int b0 = 0; b0 < W; b0 += 64) { QM_LANES(l) { if (b0 + l < W) B.lists[base + b0 + l] = M.out[b0 + l]; } } }
  wave_fence();
  QM_T(6);
  return R;
}
