// tests/emu/qm_emu.cpp -- TEST-ONLY lane emulation of the device mapper.
// Compiles rapmap_amd/csrc/qm_mapper.inl with -DQM_EMU so that every LV<T> is a
// 64-entry array and every QM_LANES block a loop: the kernel's source runs on the
// CPU one wavefront at a time.  Used by tests/ (not gpu-marked) to check the wave
// algorithm against the oracle without a GPU.  Never part of libqmap_mi355.so.
#define QM_EMU 1
#ifdef QM_PROFILE
namespace qm { unsigned long long qm_prof[32]; }
#endif
#include "../../rapmap_amd/csrc/qm_mapper.inl"
#include "../../rapmap_amd/csrc/qm_lean.inl"
#include "../../rapmap_amd/csrc/qm_duo.inl"
#include "../../rapmap_amd/csrc/qm_phflat.h"
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>

using namespace qm;

// the N-aware pass of stage A (lean_iter<..., NQ>, qm_host.hip run_stage_a): the lean kernel once more over the queue of the reads its first pass
// marked; what it maps is compared like everything else the lean kernel maps, what it marks again stays marked
template <bool PAIRED, bool SEL>
static void emu_n_pass(const qm::DevIndex& ix, qm::ReadBatch Lb, std::vector<long long>& q, u64* scal) {
  using namespace qm;
  if (q.empty()) return;
  scal[QM_SC_LEANQ] = 0;
  for (int i = 0; i < 4; ++i) scal[QM_SC_DEFER0 + i] = 0;
  Lb.slowq = q.data(); Lb.nreads = (long long)q.size();
  const long long nit = (Lb.nreads + 1) >> 1, NW = 3;
  static LeanMem Ms[3];
  for (long long w = 0; w < NW; ++w) {
    LeanMem& M = Ms[w]; memset(&M, 0, sizeof(M));
    WaveAlloc wl; wl.base = -1; wl.used = 0; wl.ivBase = -1; wl.ivUsed = 0;
    lean_stage_offsets<PAIRED, false, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<PAIRED, false, true>(Lb, (int)w, (int)nit, M, 0);
    lean_stage_offsets<PAIRED, false, true>(Lb, (int)(w + NW), (int)nit, M, 1);
    int par = 0;
    for (long long it = w; it < nit; it += NW) {
      if (ix.ph) lean_iter<PAIRED, SEL, true, false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl);
      else lean_iter<PAIRED, SEL, false, false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl);
      par ^= 1;
    }
  }
}

extern "C" {
#ifdef QM_PROFILE
unsigned long long* qe_prof() { return qm::qm_prof; }   // event counters, see QM_CNT in qm_mapper.inl
#endif

// slots: (hmask+1) 32-byte buckets; sainfo: nSA x {u32 tid, i32 pos}; text padded by >= 64 bytes
int qe_map(int k, const unsigned char* text, long long n, const u32* SA, long long nSA, const void* sainfo,
           const void* slots, unsigned long long hmask, const qm_opts* o, long long nunits,
           const unsigned char* seq1, const long long* off1, const unsigned char* seq2, const long long* off2,
           int ns, const void* ph, long long* hit_offsets, qm_hit** hits_out, unsigned long long* counters, long long* int_offsets,
           qm_sa_interval_hit** ints_out, int* status_out, const u32* txp_off, const int* txp_len) {
  DevIndex ix; ix.text = text; ix.n = n; ix.SA = SA; ix.nSA = nSA; ix.sainfo = (const SaInfo*)sainfo;
  ix.slots = (const Bucket*)slots; ix.hmask = hmask; ix.k = k; ix.ph = (const PhIndex*)ph;
  memset(&ix.phv, 0, sizeof(ix.phv)); if (ph) ix.phv = *(const PhIndex*)ph;
  std::vector<u32> sanext;                 // -s: the table of qm_host.hip's first -s call (build_sanext_kernel)
  ix.sanext = nullptr;
  static std::vector<SaExt> saext; static const u32* saextFor = nullptr; static const unsigned char* saextText = nullptr;   // the replica's SaExt table (build_saext_kernel), kept between calls on one index
  ix.saext = nullptr;
  if (!getenv("QM_NO_SAEXT")) {
    if (saextFor != SA || saextText != text || (long long)saext.size() != nSA) {
      saext.resize((size_t)nSA);
      for (long long i = 0; i < nSA; ++i) { const SaInfo si = ((const SaInfo*)sainfo)[i]; saext[(size_t)i] = saext_entry(text, n, (long long)SA[i] + k, si.tid, si.pos); }
      saextFor = SA; saextText = text;
    }
    ix.saext = saext.data();
  }
  // the wide table of the one-read-per-wavefront lean kernel (build_saext2_kernel): only for batches with reads of 129 .. 256 characters
  static std::vector<SaExt2> saext2; static const u32* saext2For = nullptr; static const unsigned char* saext2Text = nullptr;
  ix.saext2 = nullptr;
  {
    long long mx = 0;
    for (long long u = 0; u < nunits; ++u) { mx = std::max(mx, (long long)(off1[u + 1] - off1[u])); if (off2) mx = std::max(mx, (long long)(off2[u + 1] - off2[u])); }
    if (ix.saext && (ix.slots || ix.ph) && mx > QM_LEAN_MAXLEN && mx <= 2 * QM_LEAN_MAXLEN) {
      if (saext2For != SA || saext2Text != text || (long long)saext2.size() != nSA) {
        saext2.resize((size_t)nSA);
        for (long long i = 0; i < nSA; ++i) { const SaInfo si = ((const SaInfo*)sainfo)[i]; saext2[(size_t)i] = saext2_entry(text, n, (long long)SA[i] + k, si.tid, si.pos); }
        saext2For = SA; saext2Text = text;
      }
      ix.saext2 = saext2.data();
    }
  }
  if (o->sel_aln) {
    sanext.resize((size_t)nSA);
    for (long long i = 0; i < nSA; ++i) sanext[(size_t)i] = sanext_entry(text, n, (long long)SA[i] + k);
    ix.sanext = sanext.data();
  }
  const bool paired = seq2 != nullptr;
  const long long nreads = paired ? 2 * nunits : nunits;
  ReadBatch B; memset(&B, 0, sizeof(B));
  B.seq1 = seq1; B.off1 = off1; B.seq2 = seq2; B.off2 = off2; B.nreads = nreads;
  std::vector<u32> lcnt(nreads + 1, 0); std::vector<long long> loff(nreads + 1, 0);
  long long cap = nreads * 4 + 16 * QM_CHUNK;
  std::vector<u64> lists;
  std::vector<u64> gs(QM_GSCR_U64);
  std::vector<qm_sa_interval_hit> dints((size_t)nreads * 600 + 1024); std::vector<u32> dcnt(nreads + 1, 0); std::vector<long long> doff(nreads + 1, 0);
  int status = 0; u64 scal[QM_SC_WORDS]; u64& cursor = scal[0];
  B.lcnt = lcnt.data(); B.loff = loff.data(); B.cursor = scal;
  B.gscratch = gs.data(); B.status = &status; B.iv_out = dints.data(); B.iv_cnt = dcnt.data(); B.iv_off = doff.data(); B.iv_cap = (long long)dints.size();
  B.strict_check = o->strict_check; B.max_interval = o->max_interval; B.quasi_cov = o->quasi_cov; B.sensitive = o->sensitive; B.fuzzy = (seq2 != nullptr) ? o->fuzzy : 0; B.max_mmp_ext = o->max_mmp_extension > 0 ? o->max_mmp_extension : 7;
  static SelScratchLds sellds;                            // used for half of the reads so that both scratch sizes are exercised
  static SelScratch* selscr = nullptr;
  if (o->sel_aln && !selscr) selscr = new SelScratch();
  {
    float cs = (float)o->consensus_slack;                 // MappingOpts::consensusSlack is a float (RapMapSAMapper.cpp:138,184-185)
    B.consensus_fraction = (cs == 0.0) ? 1.0 : (1.0 - cs);
  }
  while (true) {
    lists.assign((size_t)cap, 0);
    B.lists = lists.data(); B.lists_cap = cap; memset(scal, 0, sizeof(scal)); status = 0;
    // emulate 7 interleaved "waves", each with its own chunk allocator
    WaveAlloc wa[7];
    for (auto& w : wa) { w.base = -1; w.used = 0; w.ivBase = -1; w.ivUsed = 0; }
    {
      // the persistent loop of qm_read_kernel with three "waves": wave w maps slots w, w + 3, ... with the kernel's own software
      // pipeline (characters of the next read and offsets of the one after staged while a read is mapped), so that what rides on
      // it -- the first-probe prefetch for the wave's next read -- runs here as it does on the device
      const int F = (ix.ph ? QM_F_PH : 0) | (B.sensitive ? 0 : QM_F_NIP) | (o->sel_aln ? QM_F_SEL : 0);
#define QE_CALL(NS_, F_) { const long long NW = 3; static WaveMem<NS_> Ms[3];                                                  \
        for (long long w = 0; w < NW; ++w) { WaveMem<NS_>& M = Ms[w];                                              \
          stage_offsets<NS_, F_>(B, w, M, 0); stage_chars<NS_, F_>(B, w, M, 0); stage_offsets<NS_, F_>(B, w + NW, M, 1);          \
          int par = 0;                                                                                                             \
          for (long long r = w; r < nreads; r += NW) {                                                                             \
            map_read<NS_, F_>(ix, B, r, r, NW, par, M, gs.data(), wa[r % 7], selscr, (r & 2) ? &sellds : nullptr); par ^= 1; } } }
      if (ns == 2) { switch (F) { case 0: QE_CALL(2, 0) break; case 1: QE_CALL(2, 1) break; case 2: QE_CALL(2, 2) break; case 3: QE_CALL(2, 3) break;
                                  case 4: QE_CALL(2, 4) break; case 5: QE_CALL(2, 5) break; case 6: QE_CALL(2, 6) break; default: QE_CALL(2, 7) break; } }
      else if (ns == 3) { switch (F) { case 0: QE_CALL(3, 0) break; case 1: QE_CALL(3, 1) break; case 2: QE_CALL(3, 2) break; case 3: QE_CALL(3, 3) break;
                                       case 4: QE_CALL(3, 4) break; case 5: QE_CALL(3, 5) break; case 6: QE_CALL(3, 6) break; default: QE_CALL(3, 7) break; } }
      else if (ns == 8) { switch (F) { case 0: QE_CALL(8, 0) break; case 1: QE_CALL(8, 1) break; case 2: QE_CALL(8, 2) break; case 3: QE_CALL(8, 3) break;
                                       case 4: QE_CALL(8, 4) break; case 5: QE_CALL(8, 5) break; case 6: QE_CALL(8, 6) break; default: QE_CALL(8, 7) break; } }
      else { switch (F) { case 0: QE_CALL(4, 0) break; case 1: QE_CALL(4, 1) break; case 2: QE_CALL(4, 2) break; case 3: QE_CALL(4, 3) break;
                          case 4: QE_CALL(4, 4) break; case 5: QE_CALL(4, 5) break; case 6: QE_CALL(4, 6) break; default: QE_CALL(4, 7) break; } }
#undef QE_CALL
    }
    if (!o->sel_aln && scal[QM_SC_SLOWCNT] > 0 && !(status & 1)) {
      // the long-read pass (see qm_host.hip): reads beyond the slot count of the first pass again, on the 32-slot kernels
      std::vector<long long> q;
      for (long long r = 0; r < nreads; ++r) if (lcnt[r] == QM_LCNT_SLOW) q.push_back(r);
      ReadBatch S2 = B; S2.slowq = q.data(); S2.nreads = (long long)q.size();
      {                                                   // (a read beyond QM_MAX_LONG_READ_LEN is skipped by that pass: empty result, scalar slot QM_SC_SKIPCNT)
        const int F = (ix.ph ? QM_F_PH : 0) | (B.sensitive ? 0 : QM_F_NIP);
        for (long long r = 0; r < (long long)q.size(); ++r) {
#define QE_LONG(F_) { static WaveMem<32> M; stage_offsets<32, F_>(S2, r, M, 0); stage_chars<32, F_>(S2, r, M, 0); \
                      map_read<32, F_>(ix, S2, read_id<F_, 32>(S2, r), r, S2.nreads, 0, M, gs.data(), wa[r % 7]); }
          switch (F) { case 0: QE_LONG(0) break; case 1: QE_LONG(1) break; case 2: QE_LONG(2) break; default: QE_LONG(3) break; }
#undef QE_LONG
        }
      }
    }
    auto rawLen = [&](long long r) -> long long {
      const unsigned char* src; const long long* off; long long unit; read_src(B, r, src, off, unit);
      return off[unit + 1] - off[unit];
    };
    if (o->sel_aln && scal[QM_SC_SLOWCNT] > 0 && !(status & 1)) {
      // -s, reads beyond the slot class (the device: the 32-slot chain-scoring collector, then the list kernel; here the fused
      // 32-slot kernel).  A read whose intervals overflow the scratch is queued again, for the slow pass below.
      std::vector<long long> q;
      for (long long r = 0; r < nreads; ++r) if (lcnt[r] == QM_LCNT_SLOW && rawLen(r) > 64 * ns) q.push_back(r);
      if (!q.empty()) {
        ReadBatch S2 = B; S2.slowq = q.data(); S2.nreads = (long long)q.size();
        const int F = (ix.ph ? QM_F_PH : 0) | (B.sensitive ? 0 : QM_F_NIP) | QM_F_SEL;
        for (long long r = 0; r < (long long)q.size(); ++r) {
#define QE_LONGS(F_) { static WaveMem<32> M; stage_offsets<32, F_>(S2, r, M, 0); stage_chars<32, F_>(S2, r, M, 0); \
                       map_read<32, F_>(ix, S2, read_id<F_, 32>(S2, r), r, S2.nreads, 0, M, gs.data(), wa[r % 7], selscr, nullptr); }
          switch (F) { case 4: QE_LONGS(4) break; case 5: QE_LONGS(5) break; case 6: QE_LONGS(6) break; default: QE_LONGS(7) break; }
#undef QE_LONGS
        }
      }
    }
    if (o->sel_aln && scal[QM_SC_SLOWCNT] > 0 && !(status & (1 | 4))) {
      // the slow pass of -s (see qm_host.hip): the queued reads again, on scratch sized for the largest of them
      const long long need = (((long long)scal[QM_SC_SLOWMAX] + 63) / 64) * 64 + 64;
      std::vector<unsigned char> dmem((size_t)SelScratchDyn::bytes_for(need));
      SelScratchDyn dyn; dyn.bind(dmem.data(), need);
      std::vector<long long> q;
      for (long long r = 0; r < nreads; ++r) if (lcnt[r] == QM_LCNT_SLOW) q.push_back(r);
      ReadBatch S2 = B; S2.slowq = q.data(); S2.dyn = &dyn; S2.nreads = (long long)q.size(); S2.iv_out = nullptr;
      const int F = (ix.ph ? QM_F_PH : 0) | (B.sensitive ? 0 : QM_F_NIP) | QM_F_SEL;
      for (long long r = 0; r < (long long)q.size(); ++r) {
#define QE_SLOW(NS_, F_) { static WaveMem<NS_> M; stage_offsets<NS_, F_>(S2, r, M, 0); stage_chars<NS_, F_>(S2, r, M, 0); \
                           map_read<NS_, F_>(ix, S2, read_id<F_>(S2, r), r, S2.nreads, 0, M, gs.data(), wa[r % 7], selscr, &sellds, &dyn); }
#define QE_SLOWL(F_) { static WaveMem<32> M; stage_offsets<32, F_>(S2, r, M, 0); stage_chars<32, F_>(S2, r, M, 0); \
                       map_read<32, F_>(ix, S2, read_id<F_, 32>(S2, r), r, S2.nreads, 0, M, gs.data(), wa[r % 7], selscr, &sellds, &dyn); }
        if (rawLen(q[(size_t)r]) > 64 * ns) { switch (F) { case 4: QE_SLOWL(4) break; case 5: QE_SLOWL(5) break; case 6: QE_SLOWL(6) break; default: QE_SLOWL(7) break; } }
        else
        if (ns == 2) { switch (F) { case 4: QE_SLOW(2, 4) break; case 5: QE_SLOW(2, 5) break; case 6: QE_SLOW(2, 6) break; default: QE_SLOW(2, 7) break; } }
        else if (ns == 3) { switch (F) { case 4: QE_SLOW(3, 4) break; case 5: QE_SLOW(3, 5) break; case 6: QE_SLOW(3, 6) break; default: QE_SLOW(3, 7) break; } }
        else if (ns == 8) { switch (F) { case 4: QE_SLOW(8, 4) break; case 5: QE_SLOW(8, 5) break; case 6: QE_SLOW(8, 6) break; default: QE_SLOW(8, 7) break; } }
        else { switch (F) { case 4: QE_SLOW(4, 4) break; case 5: QE_SLOW(4, 5) break; case 6: QE_SLOW(4, 6) break; default: QE_SLOW(4, 7) break; } }
#undef QE_SLOW
#undef QE_SLOWL
      }
    }
    if (o->sel_aln && !(status & (1 | 4)) && !getenv("QM_EMU_NO_PACK")) {
      // the packed list kernel (qm_selpack.inl: several reads per wavefront) over the intervals the passes above left behind, three
      // "waves" with a contiguous range of the reads each; every list it writes must be the one the fused path wrote for that read
      // word for word, and the reads it leaves for the one-read kernel must be exactly those on its queue
      std::vector<u32> lcnt2(nreads + 1, 0); std::vector<long long> loff2(nreads + 1, 0); std::vector<unsigned char> fnd(nreads + 1, 0);
      std::vector<long long> todo((size_t)nreads + 1, -1);
      for (long long r = 0; r < nreads; ++r) fnd[r] = (lcnt[r] >> 31) & 1;
      ReadBatch H = B; H.iv_in = dints.data(); H.iv_in_off = doff.data(); H.iv_in_cnt = dcnt.data(); H.found_in = fnd.data();
      H.iv_out = nullptr; H.lcnt = lcnt2.data(); H.loff = loff2.data();
      scal[QM_SC_TODO] = 0;
      static PackMem pm[3];
      const long long NW = 3, per = (nreads + NW - 1) / NW;
      for (long long w = 0; w < NW; ++w) {
        WaveAlloc pw; pw.base = -1; pw.used = 0; pw.ivBase = -1; pw.ivUsed = 0;
        long long r = w * per; const long long rEnd = r + per < nreads ? r + per : nreads;
        while (r < rEnd) r += sel_pack_batch(ix, H, r, rEnd, pm[w], pw, todo.data());
      }
      // the wide edition (256 intervals / suffixes per batch) over the queue the narrow one left, two "waves"
      std::vector<long long> todo2((size_t)nreads + 1, -1);
      scal[QM_SC_TODO2] = 0;
      if (!getenv("QM_EMU_NO_PACKW")) {
        static PackMemW<4> pw4[2];
        const long long nq = (long long)scal[QM_SC_TODO], NQW = 2, perq = (nq + NQW - 1) / NQW;
        for (long long w = 0; w < NQW; ++w) {
          WaveAlloc pw; pw.base = -1; pw.used = 0; pw.ivBase = -1; pw.ivUsed = 0;
          long long q = w * perq; const long long qEnd = q + perq < nq ? q + perq : nq;
          while (q < qEnd) q += sel_pack_batch_wide<4>(ix, H, todo.data(), q, qEnd, pw4[w], pw, todo2.data());
        }
        if (getenv("QM_EMU_PACK_STATS")) fprintf(stderr, "[qm emu] wide packed list kernel took %lld of %lld queued reads\n", nq - (long long)scal[QM_SC_TODO2], nq);
        todo.swap(todo2); scal[QM_SC_TODO] = scal[QM_SC_TODO2];
      }
      if (!(status & 1)) {
        std::vector<char> isTodo(nreads + 1, 0);
        for (long long q = 0; q < (long long)scal[QM_SC_TODO]; ++q) isTodo[todo[q]] = 1;
        long long bad = 0, npk = 0;
        for (long long r = 0; r < nreads; ++r) {
          if (isTodo[r]) continue;
          ++npk;
          bool same = lcnt2[r] == lcnt[r];
          const long long nwd = lcnt[r] & 0x7fffffffu;
          for (long long t = 0; same && t < nwd; ++t) same = lists[loff2[r] + t] == lists[loff[r] + t];
          if (!same) { if (bad < 5) fprintf(stderr, "[qm emu] packed list kernel: read %lld differs (words %u vs %u)\n", r, lcnt2[r] & 0x7fffffffu, lcnt[r] & 0x7fffffffu); ++bad; }
          else { loff[r] = loff2[r]; }        // downstream reads the packed kernel's copy
        }
        if (getenv("QM_EMU_PACK_STATS")) fprintf(stderr, "[qm emu] packed list kernel took %lld of %lld reads\n", npk, nreads);
        if (bad) status |= 64;
      }
    }
    if (!(status & 1)) break;
    cap *= 4;
  }
  const bool leanWide = (ns == 3 || ns == 4) && ix.saext2 != nullptr;       // qm_host.hip, run_stage_a: the wide edition's turn
  if ((ns == 2 || leanWide) && (ix.slots || ix.ph) && ix.saext && B.sensitive && !o->sel_aln && !(status & 1) && !getenv("QM_EMU_NO_LEAN")) {
    // the lean kernel (qm_lean.inl: two reads per wavefront and iteration) over the same batch, three "waves" with the kernel's own
    // software pipeline: every list it writes must be the one the general kernel wrote for that read word for word (flag bit
    // included); the reads it marks instead are the general kernel's
    std::vector<u32> lcnt2(nreads + 1, 0); std::vector<long long> loff2(nreads + 1, 0);
    std::vector<u64> lists2((size_t)cap, 0);
    u64 scal2[QM_SC_WORDS]; memset(scal2, 0, sizeof(scal2)); int status2 = 0;
    ReadBatch Lb = B; Lb.lcnt = lcnt2.data(); Lb.loff = loff2.data(); Lb.lists = lists2.data(); Lb.lists_cap = cap; Lb.cursor = scal2; Lb.status = &status2;
    Lb.iv_out = nullptr; Lb.iv_cnt = nullptr; Lb.iv_off = nullptr;
#ifdef QM_PROFILE
    const unsigned long long prof1 = qm::qm_prof[1], prof3 = qm::qm_prof[3];
    struct ProfOut { unsigned long long a, b; long long n; ~ProfOut() { fprintf(stderr, "[qm emu prof] lean kernel: %.2f bucket loads, %.2f probe rounds per read\n", (double)(qm::qm_prof[1] - a) / n, (double)(qm::qm_prof[3] - b) / n); } } profOut{prof1, prof3, nreads};
#endif
    const long long nit = leanWide ? nreads : (nreads + 1) >> 1, NW = 3;
    static LeanMem Ms[3];
    for (long long w = 0; w < NW; ++w) {
      LeanMem& M = Ms[w]; memset(&M, 0, sizeof(M));
      WaveAlloc wl; wl.base = -1; wl.used = 0; wl.ivBase = -1; wl.ivUsed = 0;
      if (leanWide && paired) {
        lean_stage_offsets<true, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<true, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_offsets<true, true>(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        for (long long it = w; it < nit; it += NW) { if (ix.ph) lean_iter<true, false, true, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); else lean_iter<true, false, false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); par ^= 1; }
      } else if (leanWide) {
        lean_stage_offsets<false, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<false, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_offsets<false, true>(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        for (long long it = w; it < nit; it += NW) { if (ix.ph) lean_iter<false, false, true, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); else lean_iter<false, false, false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); par ^= 1; }
      } else if (paired) {
        lean_stage_offsets<true>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<true>(Lb, (int)w, (int)nit, M, 0); lean_stage_offsets<true>(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        for (long long it = w; it < nit; it += NW) { if (ix.ph) lean_iter<true, false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); else lean_iter<true, false, false>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); par ^= 1; }
      } else {
        lean_stage_offsets<false>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<false>(Lb, (int)w, (int)nit, M, 0); lean_stage_offsets<false>(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        for (long long it = w; it < nit; it += NW) { if (ix.ph) lean_iter<false, false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); else lean_iter<false, false, false>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); par ^= 1; }
      }
    }
    if (!leanWide && !getenv("QM_EMU_NO_NPASS")) {
      std::vector<long long> q;
      for (long long r = 0; r < nreads; ++r) if (lcnt2[r] == QM_LCNT_LEAN) q.push_back(r);
      const size_t before = q.size();
      if (paired) emu_n_pass<true, false>(ix, Lb, q, scal2); else emu_n_pass<false, false>(ix, Lb, q, scal2);
      if (getenv("QM_EMU_LEAN_STATS")) fprintf(stderr, "[qm emu] N-aware pass over %zu reads, %llu marked again\n", before, (unsigned long long)scal2[QM_SC_LEANQ]);
    }
    long long bad = 0, deferred = 0;
    for (long long r = 0; r < nreads; ++r) {
      if (lcnt2[r] == QM_LCNT_LEAN) { ++deferred; continue; }
      bool same = lcnt2[r] == lcnt[r];
      const long long nwd = lcnt[r] & 0x7fffffffu;
      for (long long t = 0; same && t < nwd; ++t) same = lists2[loff2[r] + t] == lists[loff[r] + t];
      if (!same) { if (bad < 5) fprintf(stderr, "[qm emu] lean kernel: read %lld differs (words %u vs %u)\n", r, lcnt2[r], lcnt[r]); ++bad; }
    }
    if ((long long)scal2[QM_SC_LEANQ] != deferred) { fprintf(stderr, "[qm emu] lean kernel: %lld marks, counter says %llu\n", deferred, (unsigned long long)scal2[QM_SC_LEANQ]); ++bad; }
    if (getenv("QM_EMU_LEAN_STATS")) fprintf(stderr, "[qm emu] lean kernel took %lld of %lld reads\n", nreads - deferred, nreads);
    if (bad || (status2 & ~1)) status |= 128;
  }
  if ((ns == 2 || leanWide) && (ix.slots || ix.ph) && ix.saext && ix.sanext && B.sensitive && o->sel_aln && !(status & 1) && !getenv("QM_EMU_NO_LEAN")) {
    // the lean kernel's -s edition (chain-scoring collector: intervals + foundHit out) over the same batch: every read it takes must
    // carry exactly the interval records the general kernel's walk left for it
    std::vector<u32> lcnt2(nreads + 1, 0); std::vector<long long> loff2(nreads + 1, 0);
    std::vector<qm_sa_interval_hit> di2(dints.size()); std::vector<u32> dc2(nreads + 1, 0); std::vector<long long> do2(nreads + 1, 0);
    std::vector<unsigned char> fo2(nreads + 1, 0);
    u64 scal2[QM_SC_WORDS]; memset(scal2, 0, sizeof(scal2)); int status2 = 0;
    ReadBatch Lb = B; Lb.lcnt = lcnt2.data(); Lb.loff = loff2.data(); Lb.cursor = scal2; Lb.status = &status2;
    Lb.iv_out = di2.data(); Lb.iv_cnt = dc2.data(); Lb.iv_off = do2.data(); Lb.iv_cap = (long long)di2.size(); Lb.found_out = fo2.data();
    const long long nit = leanWide ? nreads : (nreads + 1) >> 1, NW = 3;
    static LeanMem Ms[3];
    for (long long w = 0; w < NW; ++w) {
      LeanMem& M = Ms[w]; memset(&M, 0, sizeof(M));
      WaveAlloc wl; wl.base = -1; wl.used = 0; wl.ivBase = -1; wl.ivUsed = 0;
      if (leanWide && paired) {
        lean_stage_offsets<true, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<true, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_offsets<true, true>(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        for (long long it = w; it < nit; it += NW) { if (ix.ph) lean_iter<true, true, true, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); else lean_iter<true, true, false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); par ^= 1; }
      } else if (leanWide) {
        lean_stage_offsets<false, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<false, true>(Lb, (int)w, (int)nit, M, 0); lean_stage_offsets<false, true>(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        for (long long it = w; it < nit; it += NW) { if (ix.ph) lean_iter<false, true, true, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); else lean_iter<false, true, false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); par ^= 1; }
      } else if (paired) {
        lean_stage_offsets<true>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<true>(Lb, (int)w, (int)nit, M, 0); lean_stage_offsets<true>(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        for (long long it = w; it < nit; it += NW) { if (ix.ph) lean_iter<true, true, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); else lean_iter<true, true, false>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); par ^= 1; }
      } else {
        lean_stage_offsets<false>(Lb, (int)w, (int)nit, M, 0); lean_stage_chars<false>(Lb, (int)w, (int)nit, M, 0); lean_stage_offsets<false>(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        for (long long it = w; it < nit; it += NW) { if (ix.ph) lean_iter<false, true, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); else lean_iter<false, true, false>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl); par ^= 1; }
      }
    }
    if (!leanWide && !getenv("QM_EMU_NO_NPASS")) {
      std::vector<long long> q;
      for (long long r = 0; r < nreads; ++r) if (lcnt2[r] == QM_LCNT_LEAN) q.push_back(r);
      const size_t before = q.size();
      if (paired) emu_n_pass<true, true>(ix, Lb, q, scal2); else emu_n_pass<false, true>(ix, Lb, q, scal2);
      if (getenv("QM_EMU_LEAN_STATS")) fprintf(stderr, "[qm emu] N-aware pass (-s collector) over %zu reads, %llu marked again\n", before, (unsigned long long)scal2[QM_SC_LEANQ]);
    }
    long long bad = 0, deferred = 0;
    for (long long r = 0; r < nreads; ++r) {
      if (lcnt2[r] == QM_LCNT_LEAN) { ++deferred; continue; }
      bool same = dc2[r] == dcnt[r] && (fo2[r] != 0) == (((lcnt[r] >> 31) & 1) != 0);
      for (u32 t = 0; same && t < dcnt[r]; ++t) {
        const qm_sa_interval_hit& a = di2[(size_t)(do2[r] + t)]; const qm_sa_interval_hit& b = dints[(size_t)(doff[r] + t)];
        same = a.begin == b.begin && a.end == b.end && a.len == b.len && a.query_pos == b.query_pos && a.query_rc == b.query_rc && a.list == b.list;
      }
      if (!same) { if (bad < 5) fprintf(stderr, "[qm emu] lean -s collector: read %lld differs (%u vs %u records, found %d vs %u)\n", r, dc2[r], dcnt[r], (int)fo2[r], (lcnt[r] >> 31) & 1); ++bad; }
    }
    if ((long long)scal2[QM_SC_LEANQ] != deferred) { fprintf(stderr, "[qm emu] lean -s collector: %lld marks, counter says %llu\n", deferred, (unsigned long long)scal2[QM_SC_LEANQ]); ++bad; }
    if (getenv("QM_EMU_LEAN_STATS")) fprintf(stderr, "[qm emu] lean -s collector took %lld of %lld reads\n", nreads - deferred, nreads);
    if (bad || (status2 & ~1)) status |= 128;
  }
  if (ns == 2 && paired && (ix.slots || ix.ph) && ix.saext && B.sensitive && !o->sel_aln && !(status & 1) && !getenv("QM_EMU_NO_DUO")) {
    // the pair kernel (qm_duo.inl: the two mates in lockstep in the two halves of a wavefront, the pair merged there) over the same
    // batch, three "waves" with the kernel's own software pipeline.  Every pair it merges must carry exactly the hits -- and add
    // exactly the counters -- that unit_merge makes of the general kernel's lists; every list it writes per read must be the general
    // kernel's word for word; the reads it marks are the general kernel's.  Twice: with the merge (fused calls) and without
    // (pair_cnt == null: lists only).
    for (int pass = 0; pass < 2; ++pass) {
      const char* kname = "pair kernel";
      std::vector<u32> lcnt2(nreads + 1, 0); std::vector<long long> loff2(nreads + 1, 0); std::vector<u32> pcnt(nunits + 1, 0xdeadbeefu);
      std::vector<u64> lists2((size_t)cap, 0);
      u64 scal2[QM_SC_WORDS]; memset(scal2, 0, sizeof(scal2)); int status2 = 0;
      ReadBatch Lb = B; Lb.lcnt = lcnt2.data(); Lb.loff = loff2.data(); Lb.lists = lists2.data(); Lb.lists_cap = cap; Lb.cursor = scal2; Lb.status = &status2;
      Lb.iv_out = nullptr; Lb.iv_cnt = nullptr; Lb.iv_off = nullptr;
      Lb.pair_cnt = pass == 0 ? pcnt.data() : nullptr; Lb.max_num_hits = o->max_num_hits; Lb.no_orphans = o->no_orphans; Lb.no_dovetail = o->no_dovetail;
#ifdef QM_PROFILE
      const unsigned long long prof1 = qm::qm_prof[1], prof3 = qm::qm_prof[3];
      struct ProfOut { unsigned long long a, b; long long n; ~ProfOut() { fprintf(stderr, "[qm emu prof] pair kernel: %.2f bucket loads, %.2f probe rounds per read\n", (double)(qm::qm_prof[1] - a) / n, (double)(qm::qm_prof[3] - b) / n); } } profOut{prof1, prof3, nreads};
#endif
      const long long nit = nunits, NW = 3;
      static DuoMem Ms[3];
      DuoCtr dc[3];
      for (long long w = 0; w < NW; ++w) {
        DuoMem& M = Ms[w]; memset(&M, 0xA5, sizeof(M));
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int c = 4; c < 8; ++c) M.pk[a][b][c] = 0;      // (the kernel's own initialisation)
        WaveAlloc wl; wl.base = -1; wl.used = 0; wl.ivBase = -1; wl.ivUsed = 0;
        dc[w] = DuoCtr{0, 0, 0, 0, 0, 0};
        duo_stage_offsets(Lb, (int)w, (int)nit, M, 0); duo_stage_chars(Lb, (int)w, (int)nit, M, 0); duo_stage_offsets(Lb, (int)(w + NW), (int)nit, M, 1);
        int par = 0;
        static DuoNext Nx; memset(&Nx, 0, sizeof(Nx));
        if (ix.ph) duo_prepare<true>(ix, Lb, (int)w, (int)nit, (int)NW, 0, M, Nx); else duo_prepare<false>(ix, Lb, (int)w, (int)nit, (int)NW, 0, M, Nx);
        for (long long it = w; it < nit; it += NW) {
          if (B.quasi_cov > 0.0) { if (ix.ph) duo_iter<true, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl, dc[w], Nx); else duo_iter<false, true>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl, dc[w], Nx); }
          else { if (ix.ph) duo_iter<true, false>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl, dc[w], Nx); else duo_iter<false, false>(ix, Lb, (int)it, (int)nit, (int)NW, par, M, wl, dc[w], Nx); }
          par ^= 1;
        }
      }
      PairBatch Pg; memset(&Pg, 0, sizeof(Pg));           // the general kernel's lists through stage B: what a merged pair must equal
      std::vector<u32> hcg(nunits + 1, 0);
      Pg.n = nunits; Pg.paired = 1; Pg.off1 = off1; Pg.off2 = off2; Pg.lcnt = lcnt.data(); Pg.loff = loff.data(); Pg.lists = lists.data(); Pg.cnt = hcg.data();
      Pg.max_num_hits = o->max_num_hits; Pg.no_orphans = o->no_orphans; Pg.no_dovetail = o->no_dovetail; Pg.fuzzy = o->fuzzy;
      PairBatch Pd = Pg; Pd.lcnt = lcnt2.data(); Pd.loff = loff2.data(); Pd.lists = lists2.data(); Pd.cnt = pcnt.data();
      UnitCounters want = {0, 0, 0, 0, 0, 0};
      long long bad = 0, deferred = 0, merged = 0;
      for (long long u = 0; u < nunits; ++u) {
        if (lcnt2[2 * u] == QM_LCNT_PAIR) {
          ++merged;
          if (pass == 1 || o->fuzzy) { if (bad < 5) fprintf(stderr, "[qm emu] %s: pair %lld merged although it may not be\n", kname, u); ++bad; continue; }
          UnitCounters uc = {0, 0, 0, 0, 0, 0};
          const int cg = unit_merge(Pg, u, nullptr, 0, &uc);
          want.pe += uc.pe; want.se += uc.se; want.tot += uc.tot; want.reads += uc.reads; want.tooMany += uc.tooMany; want.mapped += uc.mapped;
          bool same = (u32)cg == pcnt[u];
          if (same && cg > 0) {
            std::vector<qm_hit> hg((size_t)cg), hd((size_t)cg);
            unit_merge(Pg, u, hg.data(), cg, nullptr); unit_merge(Pd, u, hd.data(), cg, nullptr);
            same = memcmp(hg.data(), hd.data(), sizeof(qm_hit) * (size_t)cg) == 0;
          }
          if (!same) { if (bad < 5) fprintf(stderr, "[qm emu] %s: pair %lld differs (%u hits vs %d)\n", kname, u, pcnt[u], cg); ++bad; }
          continue;
        }
        for (int m = 0; m < 2; ++m) {
          const long long r = 2 * u + m;
          if (lcnt2[r] == QM_LCNT_LEAN) { ++deferred; continue; }
          bool same = lcnt2[r] == lcnt[r];
          const long long nwd = lcnt[r] & 0x7fffffffu;
          for (long long t = 0; same && t < nwd; ++t) same = lists2[loff2[r] + t] == lists[loff[r] + t];
          if (!same) { if (bad < 5) fprintf(stderr, "[qm emu] %s: read %lld differs (words %u vs %u)\n", kname, r, lcnt2[r], lcnt[r]); ++bad; }
        }
      }
      DuoCtr got = {0, 0, 0, 0, 0, 0};
      for (long long w = 0; w < NW; ++w) { got.pe += dc[w].pe; got.se += dc[w].se; got.tot += dc[w].tot; got.reads += dc[w].reads; got.tooMany += dc[w].tooMany; got.mapped += dc[w].mapped; }
      if (got.pe != want.pe || got.se != want.se || got.tot != want.tot || got.reads != want.reads || got.tooMany != want.tooMany || got.mapped != want.mapped) {
        fprintf(stderr, "[qm emu] %s: counters of the merged pairs %u %u %u %u %u %u, stage B says %llu %llu %llu %llu %llu %llu\n", kname, got.pe, got.se, got.tot, got.reads, got.tooMany, got.mapped,
                (unsigned long long)want.pe, (unsigned long long)want.se, (unsigned long long)want.tot, (unsigned long long)want.reads, (unsigned long long)want.tooMany, (unsigned long long)want.mapped);
        ++bad;
      }
      if ((long long)scal2[QM_SC_LEANQ] != deferred) { fprintf(stderr, "[qm emu] %s: %lld marks, counter says %llu\n", kname, deferred, (unsigned long long)scal2[QM_SC_LEANQ]); ++bad; }
      if (getenv("QM_EMU_LEAN_STATS")) fprintf(stderr, "[qm emu] %s (pass %d) took %lld of %lld reads, merged %lld of %lld pairs\n", kname, pass, nreads - deferred, nreads, merged, nunits);
      if (bad || (status2 & ~1)) status |= 32;
    }
  }
  PairBatch P; memset(&P, 0, sizeof(P));
  std::vector<u32> hc(nunits + 1, 0); std::vector<long long> offs(nunits + 1, 0);
  u64 ctr[6] = {0, 0, 0, 0, 0, 0};
  P.n = nunits; P.paired = paired ? 1 : 0; P.off1 = off1; P.off2 = off2; P.lcnt = lcnt.data(); P.loff = loff.data();
  P.lists = lists.data(); P.cnt = hc.data(); P.offs = offs.data(); P.counters = ctr;
  P.max_num_hits = o->max_num_hits; P.no_orphans = o->no_orphans; P.no_dovetail = o->no_dovetail; P.fuzzy = o->fuzzy;
  UnitCounters uc = {0, 0, 0, 0, 0, 0};
  qm_hit* out = nullptr;
  if (o->sel_aln) {
    // -s: stage B + C per unit into per-unit temp slots, then compaction (qm_sel.inl)
    std::vector<long long> toff(nunits + 1, 0);
    for (long long u = 0; u < nunits; ++u) {
      long long w = paired ? (long long)(lcnt[2 * u] & 0x7fffffffu) + (lcnt[2 * u + 1] & 0x7fffffffu) : (long long)(lcnt[u] & 0x7fffffffu);
      toff[u + 1] = toff[u] + w / 3 + 1;
    }
    std::vector<qm_hit> tmp((size_t)toff[nunits] + 1); std::vector<int> tsc(2 * (size_t)toff[nunits] + 2);
    SelBatch A; memset(&A, 0, sizeof(A));
    A.seq1 = seq1; A.seq2 = seq2; A.text = text; A.txp_off = txp_off; A.txp_len = txp_len; A.tmp = tmp.data(); A.toff = toff.data();
    A.tsc = tsc.data();
    A.match = o->match_score; A.mismatch = o->mismatch_penalty; A.gap_open = o->gap_open; A.gap_extend = o->gap_extend;
    A.bandwidth = o->dp_bandwidth; A.hard_filter = o->hard_filter; A.policy = o->aln_policy; A.min_score_fraction = o->min_score_fraction;
    {
      std::vector<int> tref(2 * (size_t)toff[nunits] + 2);
      std::vector<SelTask> tasks(2 * (size_t)toff[nunits] + 2); u64 ntasks = 0, nsides = 0;
      std::vector<SelSide> sides(2 * (size_t)toff[nunits] + 2); std::vector<u64> torder(2 * (size_t)toff[nunits] + 2), torder2(2 * (size_t)toff[nunits] + 2); u64 ntasks2 = 0;
      A.tref = tref.data(); A.tasks = tasks.data(); A.ntasks = &ntasks; A.sides = sides.data(); A.nsides = &nsides; A.torder = torder.data();
      if (!getenv("QM_SEL_NO_STRIP")) { A.torder2 = torder2.data(); A.ntasks2 = &ntasks2; }
      A.u0 = 0; A.u1 = nunits;
      // the plan's three steps as the kernels run them (qm_sel_sides_kernel, qm_sel_score_kernel with one lane per question, qm_sel_dedupe_kernel)
      for (long long u = 0; u < nunits; ++u) { const int m = sel_unit_sides_count(P, A, u, &uc); if (m > 0) { sel_unit_sides_write(P, A, u, (long long)nsides); nsides += (u64)m; } }
      for (u64 x = 0; x < nsides; ++x) sel_side_score<1>(P, A, (long long)x, 0, SelRedOne());
      for (u64 x = 0; x < nsides; ++x) sel_side_dedupe(A, (long long)x);
      { static StripMem SM; for (u64 t = 0; t < ntasks2; t += 4) sel_tasks_strip(A, t, ntasks2, SM); }      // as qm_sel_strip_kernel
      if (getenv("QM_EMU_SEL_STATS")) fprintf(stderr, "[qm emu] -s: %llu questions, %llu ksw2 alignments, %llu strip alignments\n", (unsigned long long)nsides, (unsigned long long)ntasks, (unsigned long long)ntasks2);
      unsigned char codes[512]; sel_ksw_fill_codes(codes, 0, 1);
      bool longReads = false;
      for (long long u = 0; u < nunits; ++u) { if (off1[u + 1] - off1[u] > QM_MAX_READ_LEN || (paired && off2[u + 1] - off2[u] > QM_MAX_READ_LEN)) longReads = true; }
      A.long_reads = longReads ? 1 : 0;
      if (longReads) switch (sel_ksw_ring_slots(A.bandwidth)) {
        case 32: { std::vector<KswRowT<32, QM_KSW_MAXLEN_LONG>> rows(4); for (u64 t = 0; t < ntasks; t += 4) sel_tasks_align_rows<32, QM_KSW_MAXLEN_LONG>(P, A, t, ntasks, rows.data(), codes); } break;
        case 64: { std::vector<KswRowT<64, QM_KSW_MAXLEN_LONG>> rows(4); for (u64 t = 0; t < ntasks; t += 4) sel_tasks_align_rows<64, QM_KSW_MAXLEN_LONG>(P, A, t, ntasks, rows.data(), codes); } break;
        case 128: { std::vector<KswRowT<128, QM_KSW_MAXLEN_LONG>> rows(4); for (u64 t = 0; t < ntasks; t += 4) sel_tasks_align_rows<128, QM_KSW_MAXLEN_LONG>(P, A, t, ntasks, rows.data(), codes); } break;
        default: { std::vector<KswRowT<QM_KSW_RING_GMEM, QM_KSW_MAXLEN_LONG>> rows(4); for (u64 t = 0; t < ntasks; t += 4) sel_tasks_align_rows<QM_KSW_RING_GMEM, QM_KSW_MAXLEN_LONG>(P, A, t, ntasks, rows.data(), codes); } break;   // as qm_sel_align_gmem_kernel
      }
      else
      switch (sel_ksw_ring_slots(A.bandwidth)) {         // same rule as the launch wrapper
        case 32: { std::vector<KswRowT<32>> rows(8); for (u64 t = 0; t < ntasks; t += 8) sel_tasks_align_rows2<QM_KSW_MAXLEN>(P, A, t, ntasks, rows.data(), codes); } break;   // as qm_sel_align2_kernel
        case 64: { std::vector<KswRowT<64>> rows(4); for (u64 t = 0; t < ntasks; t += 4) sel_tasks_align_rows<64>(P, A, t, ntasks, rows.data(), codes); } break;
        case 128: { std::vector<KswRowT<128>> rows(4); for (u64 t = 0; t < ntasks; t += 4) sel_tasks_align_rows<128>(P, A, t, ntasks, rows.data(), codes); } break;
        default: { std::vector<KswRowT<1024>> rows(4); for (u64 t = 0; t < ntasks; t += 4) sel_tasks_align_rows<1024>(P, A, t, ntasks, rows.data(), codes); } break;
      }
      for (long long u = 0; u < nunits; ++u) hc[u] = (u32)sel_unit_finish(P, A, u, &uc);
    }
    hit_offsets[0] = 0;
    for (long long u = 0; u < nunits; ++u) hit_offsets[u + 1] = hit_offsets[u] + hc[u];
    out = (qm_hit*)malloc(sizeof(qm_hit) * (size_t)(hit_offsets[nunits] + 1));
    for (long long u = 0; u < nunits; ++u) for (u32 i = 0; i < hc[u]; ++i) out[hit_offsets[u] + i] = tmp[(size_t)toff[u] + i];
  } else {
  for (long long u = 0; u < nunits; ++u) hc[u] = (u32)unit_merge(P, u, nullptr, 0, &uc);
  hit_offsets[0] = 0;
  for (long long u = 0; u < nunits; ++u) hit_offsets[u + 1] = hit_offsets[u] + hc[u];
  for (long long u = 0; u <= nunits; ++u) offs[u] = hit_offsets[u];
  out = (qm_hit*)malloc(sizeof(qm_hit) * (size_t)(hit_offsets[nunits] + 1));
  P.hits = out;
  for (long long u = 0; u < nunits; ++u) if (hc[u]) unit_merge(P, u, out + offs[u], (int)hc[u], nullptr);
  }
  *hits_out = out;
  counters[0] = uc.pe; counters[1] = uc.se; counters[2] = uc.tot; counters[3] = uc.reads;
  counters[4] = uc.tooMany; counters[5] = uc.mapped;
  const int mates = paired ? 2 : 1;
  int_offsets[0] = 0;
  for (long long u = 0; u < nunits; ++u) {
    long long t = 0;
    for (int m = 0; m < mates; ++m) t += dcnt[u * mates + m];
    int_offsets[u + 1] = int_offsets[u] + t;
  }
  qm_sa_interval_hit* io = (qm_sa_interval_hit*)malloc(sizeof(qm_sa_interval_hit) * (size_t)(int_offsets[nunits] + 1));
  {
    long long w = 0;
    for (long long r = 0; r < nreads; ++r) for (u32 j = 0; j < dcnt[r]; ++j) io[w++] = dints[(size_t)(doff[r] + j)];
  }
  *ints_out = io;
  *status_out = status | (int)(scal[QM_SC_SLOWCNT] << 8) | (int)((scal[QM_SC_SKIPCNT] & 0x7f) << 24);   // bits 8..: reads that took the slow pass of -s; bits 24..: skipped reads
  return 0;
}

void qe_free(void* p) { free(p); }

// perfect-hash flavour: assemble a PhIndex over caller-owned arrays (same flattening as qm_ctx_create)
void* qe_ph_create(const unsigned long long* words, const unsigned long long* ranks, const unsigned long long* levelTab,
                   int nb_levels, const u32* data, const unsigned char* lens, unsigned long long nelem,
                   unsigned long long lastbitsetrank, const u32* ovf_kv, long long n_ovf,
                   const unsigned long long* fin_kv, long long n_fin,
                   unsigned long long total_words, unsigned long long total_ranks,
                   int k, const unsigned char* text, long long n, const u32* SA, long long nSA) {
  PhIndex* P = new PhIndex();
  memset(P, 0, sizeof(*P));
  // levelTab here: {domain, first word, first rank sample} per level, as read from hash_info.bph
  std::vector<PhLevelIn> lin((size_t)nb_levels);
  for (int i = 0; i < nb_levels; ++i) {
    const u64 w0 = levelTab[3 * i + 1], r0 = levelTab[3 * i + 2];
    const u64 w1 = i + 1 < nb_levels ? levelTab[3 * (i + 1) + 1] : total_words, r1 = i + 1 < nb_levels ? levelTab[3 * (i + 1) + 2] : total_ranks;
    lin[i].words = (const uint64_t*)words + w0; lin[i].nchar = w1 - w0; lin[i].domain = levelTab[3 * i];
    lin[i].ranks = (const uint64_t*)ranks + r0; lin[i].nranks = r1 - r0;
  }
  std::vector<uint64_t>* blocks = new std::vector<uint64_t>(); std::vector<uint64_t>* tab = new std::vector<uint64_t>();
  if (!ph_flatten_blocks(lin, *blocks, *tab)) return nullptr;
  P->blocks = (const u64*)blocks->data(); P->levelTab = (const u64*)tab->data();
  PhRec* recs = new PhRec[nelem ? nelem : 1];
  DevIndex dix; memset(&dix, 0, sizeof(dix)); dix.text = text; dix.n = n; dix.SA = SA; dix.nSA = nSA; dix.k = k;
  for (u64 i = 0; i < nelem; ++i) {
    PhRec r; r.data = data[i]; r.len = lens[i]; r.pad[0] = r.pad[1] = r.pad[2] = 0;
    u64 m = 0;
    if ((long long)r.data < nSA) text_kmer(dix, (long long)SA[r.data], k, m);
    r.key = m;
    recs[i] = r;
  }
  P->recs = recs;
  P->nelem = nelem; P->lastbitsetrank = lastbitsetrank; P->nb_levels = nb_levels;
  if (!getenv("QM_PH_NO_FILTER")) {      // the membership pre-filter of the compact image, as qm_ctx_create_ex builds it on the GPU
    const u64 fw = ph_filter_words(nelem);
    u64* fil = new u64[fw];
    for (u64 i = 0; i < fw; ++i) fil[i] = 0;
    for (u64 i = 0; i < nelem; ++i) { u64 w, bits; ph_filter_slot(recs[i].key, word_rc(recs[i].key, k), fw - 1, w, bits); fil[w] |= bits; }
    P->filter = fil; P->filterMask = fw - 1;
  }
  u64 cap = 16; while (cap < (u64)n_ovf * 2) cap <<= 1;
  OvfSlot* ov = new OvfSlot[cap];
  for (u64 i = 0; i < cap; ++i) { ov[i].key = ~0u; ov[i].val = 0; }
  for (long long i = 0; i < n_ovf; ++i) { u64 j = hash_mix((u64)ovf_kv[2 * i]) & (cap - 1); while (ov[j].key != ~0u) j = (j + 1) & (cap - 1); ov[j].key = ovf_kv[2 * i]; ov[j].val = ovf_kv[2 * i + 1]; }
  P->ovf = ov; P->ovfMask = cap - 1;
  u64 fc = 16; while (fc < (u64)n_fin * 2) fc <<= 1;
  Slot* fin = new Slot[fc];
  for (u64 i = 0; i < fc; ++i) { fin[i].key = ~0ULL; fin[i].lb = 0; fin[i].ub = 0; }
  for (long long i = 0; i < n_fin; ++i) { u64 j = hash_mix(fin_kv[2 * i]) & (fc - 1); while (fin[j].key != ~0ULL) j = (j + 1) & (fc - 1); fin[j].key = fin_kv[2 * i]; fin[j].lb = (int)fin_kv[2 * i + 1]; }
  P->fin = fin; P->finMask = fc - 1;
  return P;
}

// host-side flattening for the emulation only (the product does this on the GPU,
// rapmap_amd/csrc/qm_kernels.hip: build_sainfo_kernel / build_slots_kernel)
// four alignments at once through the 16-lane-row kernel: qlen[4], tlen[4], pointers to the code strings, out[4]
}  // extern "C"
template <int RING>
static void ksw_rows_run(const int* qlen, const unsigned char* const* query, const int* tlen, const unsigned char* const* target,
                         const signed char* mat, int q, int e, int w, int* out) {
  std::vector<KswRowT<RING>> blk(4);
  LV<int> ql, tl, sc;
  for (int g = 0; g < 4; ++g) {
    const int tlen16 = (tlen[g] + 15) / 16 * 16;
    memset(&blk[g], 0xAB, sizeof(KswRowT<RING>));        // the kernel must not depend on what the block held before
    // only what sel_tasks_align_rows stages for an alignment of this size: the rest keeps the garbage
    for (int i = 0; i < QM_KSW_MAXLEN + 40 && i < tlen16 + qlen[g] + 48; ++i) {
      blk[g].QX[i] = (i >= 16 && i < 16 + qlen[g]) ? query[g][i - 16] : 0;
      const int j = i - tlen16;
      blk[g].TX[i] = i < tlen[g] ? target[g][i] : (i < tlen16 ? 0 : (j < qlen[g] ? query[g][qlen[g] - 1 - j] : 0));
    }
    for (int c = 0; c < 16; ++c) { ql[g * 16 + c] = qlen[g]; tl[g * 16 + c] = tlen[g]; }
  }
  sel_ksw_extz2_rows<RING>(ql, tl, blk.data(), mat, q, e, w, sc);
  for (int g = 0; g < 4; ++g) out[g] = sc[g * 16];
}
// eight alignments of ONE shape (n of them real, the rest idle rows) as two sets through the register edition: out[8]
static void ksw_rows_run8(int n, int qlen, const unsigned char* const* query, int tlen, const unsigned char* const* target,
                          const signed char* mat, int q, int e, int w, int* out) {
  std::vector<KswRowT<32>> blk(8);
  LV<int> ql[2], tl[2], sc[2];
  const int tlen16 = (tlen + 15) / 16 * 16;
  for (int g = 0; g < 8; ++g) {
    memset(&blk[g], 0xAB, sizeof(KswRowT<32>));
    if (g < n) for (int i = 0; i < QM_KSW_MAXLEN + 40 && i < tlen16 + qlen + 48; ++i) {
      blk[g].QX[i] = (i >= 16 && i < 16 + qlen) ? query[g][i - 16] : 0;
      const int j = i - tlen16;
      blk[g].TX[i] = i < tlen ? target[g][i] : (i < tlen16 ? 0 : (j < qlen ? query[g][qlen - 1 - j] : 0));
    }
    for (int c = 0; c < 16; ++c) { ql[g / 4][(g & 3) * 16 + c] = g < n ? qlen : 0; tl[g / 4][(g & 3) * 16 + c] = g < n ? tlen : 0; }
  }
  sel_ksw_extz2_rows_reg<QM_KSW_MAXLEN, true, 2>(ql, tl, qlen, tlen, blk.data(), mat, q, e, w, sc);
  for (int g = 0; g < 8; ++g) out[g] = sc[g / 4][(g & 3) * 16];
}
extern "C" {
void qe_ksw_rows8(int n, int qlen, const unsigned char* const* query, int tlen, const unsigned char* const* target,
                  int a, int b, int q, int e, int w, int* out) {
  signed char mat[25];
  a = a < 0 ? -a : a; b = b > 0 ? -b : b;
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[i * 5 + j] = (signed char)(i == j ? a : b); mat[i * 5 + 4] = 0; }
  for (int j = 0; j < 5; ++j) mat[20 + j] = 0;
  ksw_rows_run8(n, qlen, query, tlen, target, mat, q, e, w, out);
}
// ring < 0: the ring the launch wrapper would pick for this band; otherwise force 64 / 128 / 1024 slots
void qe_ksw_rows(const int* qlen, const unsigned char* const* query, const int* tlen, const unsigned char* const* target,
                 int a, int b, int q, int e, int w, int* out, int ring) {
  signed char mat[25];
  a = a < 0 ? -a : a; b = b > 0 ? -b : b;
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[i * 5 + j] = (signed char)(i == j ? a : b); mat[i * 5 + 4] = 0; }
  for (int j = 0; j < 5; ++j) mat[20 + j] = 0;
  if (ring < 0) ring = sel_ksw_ring_slots(w);
  if (ring == 32) ksw_rows_run<32>(qlen, query, tlen, target, mat, q, e, w, out);
  else if (ring == 64) ksw_rows_run<64>(qlen, query, tlen, target, mat, q, e, w, out);
  else if (ring == 128) ksw_rows_run<128>(qlen, query, tlen, target, mat, q, e, w, out);
  else ksw_rows_run<1024>(qlen, query, tlen, target, mat, q, e, w, out);
}
unsigned long long qe_slots_cap(long long nkeys) { return bucket_count(nkeys); }   // buckets of 64 bytes
void qe_flatten(const u32* SA, long long nSA, const u32* offsets, long long T, void* sainfo_out,
                const unsigned long long* keys, const u32* lb, const u32* ub, long long K, void* slots_out,
                unsigned long long cap, int k) {
  SaInfo* si = (SaInfo*)sainfo_out;
  for (long long i = 0; i < nSA; ++i) {
    const u32 p = SA[i];
    long long lo = 0, hi = T;   // upper_bound(offsets, p) - 1
    while (lo < hi) { long long mid = (lo + hi) >> 1; if (offsets[mid] <= p) lo = mid + 1; else hi = mid; }
    long long tid = lo - 1;
    si[i].tid = (u32)tid; si[i].pos = (int)(p - offsets[tid]);
  }
  Bucket* bk = (Bucket*)slots_out;
  memset(bk, 0xff, (size_t)cap * sizeof(Bucket));
  for (long long i = 0; i < K; ++i)
    bucket_insert(bk, cap - 1, keys[i], k, lb[i], ub[i],
                  [](u64* p, u64 cmp, u64 val) { const u64 o = *p; if (o == cmp) *p = val; return o; },
                  [](u64* p, u64 v) { *p |= v; });
}
}

// differential check of the integer one-diagonal chaining editions (qm_selpack.inl) against sel_chain_group (the doubles): recs = hn x
// {tid, pos, qpos, len, iv} in chain order.  Returns 0 when the edition declines the group (not one diagonal / more than 8 hits for which == 8),
// 1 when it agrees with sel_chain_group in starts, status, own position and positions, -1 when it differs.
extern "C" int qe_chain_diag_check(const unsigned* recs, int hn, int maxDist, int which) {
  using namespace qm;
  std::vector<SelRec> H(hn);
  for (int i = 0; i < hn; ++i) { H[i].tid = recs[5 * i]; H[i].pos = recs[5 * i + 1]; H[i].qpos = recs[5 * i + 2]; H[i].len = recs[5 * i + 3]; H[i].iv = recs[5 * i + 4]; }
  std::vector<double> f(hn + 1); std::vector<int> p(hn + 1), seen(hn + 1), ends(hn + 1), starts(hn + 1), posA(hn + 1), posB(hn + 1);
  SelGroup ga, gb; ga.offcs = 0; gb.offcs = 0;
  const int na = sel_chain_group(H.data(), hn, f.data(), p.data(), seen.data(), ends.data(), starts.data(), maxDist, ga, posA.data());
  std::vector<int> fi(2 * hn + 2), pi(hn + 1), si(hn + 1), ei(hn + 1);
  const int nb = which == 8 ? sel_chain_diag8(H.data(), hn, maxDist, gb, posB.data())
                            : sel_chain_diag_mem(H.data(), hn, maxDist, fi.data(), pi.data(), si.data(), ei.data(), gb, posB.data());
  if (nb == 0) return 0;
  if (na != nb || ga.cs() != gb.cs() || ga.ppos != gb.ppos || ga.tid != gb.tid) return -1;
  for (int t = 0; t < na; ++t) if (posA[t] != posB[t]) return -1;
  return 1;
}
