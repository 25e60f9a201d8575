// qm_kernels.hip -- gfx950 kernels of libqmap_mi355.so and their launch wrappers.
//
//   qm_read_kernel<NS>    stage A: one 64-lane wavefront per READ (qm_mapper.inl: collector + hits->mappings);
//                         persistent grid, 4 waves per workgroup, per-wave LDS slab, integer only.
//   qm_pair_count/write   stage B: one thread per read pair (mergeLeftRightHits + driver), count -> scan -> write
//   build_sainfo_kernel   index flattening: (transcript id, offset) for every SA entry
//                         (replaces rank9b::rank + txpOffsets lookups on the hot path,
//                         src/rank9b.cpp:56-61, src/RapMapSAIndex.cpp:92-94)
//   build_slots_kernel    index flattening: bucketized k-mer table (32-byte buckets of two slots) from hash.bin records
#include <hip/hip_runtime.h>
#include <cstring>
#include <type_traits>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

#include "qm_mapper.inl"
#include "qm_device.h"

namespace qm {

template <bool ON> struct SelLds { SelScratchLds s; QM_DEV SelScratchLds* ptr() { return &s; } };
template <> struct SelLds<false> { QM_DEV SelScratchLds* ptr() { return nullptr; } };

// Stage entry "from intervals" (hit_manager::hitsToMappingsSimple as a call of its own, include/HitManager.hpp:130-135): one
// wavefront per read, the read's SA-interval hits come from the caller instead of the collector.  F: 0 or QM_F_SEL.
struct H2mMem { u64 buf[3][QM_CAP]; IntRec ints[2][QM_ICAP]; };        // 2 KB per wave: sort buffers + the first intervals of each strand
#ifndef QM_H2M_WPS
#define QM_H2M_WPS 4       // waves per SIMD the list kernel is built for (its LDS -- 39 KB per block -- allows no more)
#endif
template <int F>
__global__ __launch_bounds__(256, QM_H2M_WPS) void qm_h2m_kernel(DevIndex ix, ReadBatch B) {
  __shared__ H2mMem mem[4];
  __shared__ SelLds<(F & QM_F_SEL) != 0> sels[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const long long nw = (long long)gridDim.x * 4;
  u64* gscr = B.gscratch + gw * QM_GSCR_U64;
  WaveAlloc wa; wa.base = -1; wa.used = 0; wa.ivBase = -1; wa.ivUsed = 0;
#ifdef QM_TIMING       // phases of this kernel (-s): 0 interval records in, 1 suffixes gathered, 2 sort, 3 groups + chaining, 4 list assembled, 6 write-out
  if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 9; ++i) qm_tim[wave][i] = 0; qm_tim[wave][9] = __builtin_readcyclecounter(); }
#endif
  long long nslots = B.nreads;
  if (B.nreads_dev) { const long long q = (long long)uniform(*B.nreads_dev); nslots = q < nslots ? q : nslots; }
  for (long long r = gw; r < nslots; r += nw) {
    const long long read = read_id<F>(B, r);
    H2mMem& M = mem[wave];
    IntervalList fi, ri;
    fi.lds = (QM_LDS(IntRec)*)M.ints[0]; ri.lds = (QM_LDS(IntRec)*)M.ints[1];
    fi.ovf = (IntRec*)(gscr + 3 * QM_GCAP); ri.ovf = fi.ovf + QM_IOVF;
    fi.n = 0; ri.n = 0; fi.pf = nullptr; ri.pf = nullptr; fi.pfcap = 0; ri.pfcap = 0;
    long long i0, i1; int len, mate = 0;
    if (B.iv_in_cnt) {                                  // second pass of a fused -s call: what the collector pass left for this read
      i0 = uniform(B.iv_in_off[read]); i1 = i0 + (long long)uniform(B.iv_in_cnt[read]);
      const unsigned char* src; const long long* off; long long unit;
      read_src(B, read, src, off, unit);
      len = (int)(uniform(off[unit + 1]) - uniform(off[unit]));
      mate = B.seq2 ? (int)(read & 1) : 0;
    } else {
      i0 = uniform(B.iv_in_off[read]); i1 = uniform(B.iv_in_off[read + 1]);
      len = uniform(B.len_in[read]);
    }
    // all of the read's records in one round of loads: lane l takes record base + l and files it behind the records of its
    // strand that precede it (forward-strand records come first)
    for (long long base = i0; base < i1; base += 64) {
      LV<bool> isF, isR;
      qm_sa_interval_hit h; h.begin = 0; h.end = 0; h.len = 0; h.query_pos = 0; h.query_rc = 0;
      const bool have = base + (long long)(threadIdx.x & 63) < i1;
      if (have) h = B.iv_in[base + (long long)(threadIdx.x & 63)];
      isF.v[0] = have && h.query_rc == 0; isR.v[0] = have && h.query_rc != 0;
      const u64 fm = ballot(isF), rm = ballot(isR);
      if (have) {
        const int l = (int)(threadIdx.x & 63);
        const bool rc = h.query_rc != 0;
        const int idx = rc ? ri.n + popc64(rm & lanemask_lt(l)) : fi.n + popc64(fm & lanemask_lt(l));
        IntRec r; r.b = (u32)h.begin; r.e = (u32)h.end; r.len = h.len; r.q = h.query_pos;
        IntervalList& L = rc ? ri : fi;
        if (idx < QM_ICAP) { L.lds[idx].b = r.b; L.lds[idx].e = r.e; L.lds[idx].len = r.len; L.lds[idx].q = r.q; }
        else L.ovf[idx - QM_ICAP] = r;
      }
      fi.n += popc64(fm); ri.n += popc64(rm);
    }
    wave_fence();
    QM_T(0);
    const bool found = B.found_in ? uniform((int)B.found_in[read]) != 0 : false;
    finish_read<4, F>(ix, B, read, len, mate, found, M.buf, gscr, wa, fi, ri, (F & QM_F_SEL) ? B.selscr + gw : nullptr, sels[wave].ptr(),
                      ((F & QM_F_SEL) && B.dyn) ? B.dyn + gw : nullptr);
  }
#ifdef QM_TIMING
  if ((threadIdx.x & 63) == 0) for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&B.cursor[32 + i], (unsigned long long)qm_tim[wave][i]);
#endif
}

// The list kernel of a fused -s call, several reads per wavefront (qm_selpack.inl): every wavefront owns a contiguous range of
// the reads and walks it in batches of as many reads as fit its 64 lanes; what it cannot take goes to `todoq` for qm_h2m_kernel.
#ifndef QM_PK_WPS
#define QM_PK_WPS 6
#endif
__global__ __launch_bounds__(256, QM_PK_WPS) void qm_h2m_pack_kernel(DevIndex ix, ReadBatch B, long long* todoq) {
  __shared__ PackMem mem[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const long long nw = (long long)gridDim.x * 4;
  const long long per = (B.nreads + nw - 1) / nw;
  long long r = gw * per;
  const long long rEnd = r + per < B.nreads ? r + per : B.nreads;
  WaveAlloc wa; wa.base = -1; wa.used = 0; wa.ivBase = -1; wa.ivUsed = 0;
#ifdef QM_TIMING       // phases: 0 candidates + intervals, 1 suffix gather + keys, 2 rank sort, 3 heads + interval counting, 4 chaining, 5 words, 6 write-out
  if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 9; ++i) qm_tim[wave][i] = 0; qm_tim[wave][9] = __builtin_readcyclecounter(); }
#endif
  while (r < rEnd) r += (long long)sel_pack_batch(ix, B, r, rEnd, mem[wave], wa, todoq);
#ifdef QM_TIMING
  if ((threadIdx.x & 63) == 0) for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&B.cursor[40 + i], (unsigned long long)qm_tim[wave][i]);
#endif
}

// ... and its wide edition (256 intervals / suffixes per batch) over the queue the narrow one leaves: reads of 150 bp and more
__global__ __launch_bounds__(256, 2) void qm_h2m_packw_kernel(DevIndex ix, ReadBatch B, const long long* ids, const u64* nids, long long* todoq) {
  __shared__ PackMemW<4> mem[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const long long nw = (long long)gridDim.x * 4;
  long long nq = B.nreads;
  if (nids) { const long long d = (long long)uniform(*nids); nq = d < nq ? d : nq; }
  const long long per = (nq + nw - 1) / nw;
  long long q = gw * per;
  const long long qEnd = q + per < nq ? q + per : nq;
  WaveAlloc wa; wa.base = -1; wa.used = 0; wa.ivBase = -1; wa.ivUsed = 0;
#ifdef QM_TIMING
  if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 9; ++i) qm_tim[wave][i] = 0; qm_tim[wave][9] = __builtin_readcyclecounter(); }
#endif
  while (q < qEnd) q += (long long)sel_pack_batch_wide<4>(ix, B, ids, q, qEnd, mem[wave], wa, todoq);
#ifdef QM_TIMING       // (the narrow kernel's slots: [qm timing pack] of a batch of long reads is this kernel)
  if ((threadIdx.x & 63) == 0) for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&B.cursor[40 + i], (unsigned long long)qm_tim[wave][i]);
#endif
}

// stage B pass 1: hits per unit + the HitCounters
__global__ __launch_bounds__(256) void qm_pair_count_kernel(PairBatch P) {
  __shared__ unsigned long long sc[6];
  if (threadIdx.x < 6) sc[threadIdx.x] = 0;
  __syncthreads();
  long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  UnitCounters uc = {0, 0, 0, 0, 0, 0};
  if (u < P.n) P.cnt[u] = (u32)unit_merge(P, u, nullptr, 0, &uc);
  if (uc.pe) atomicAdd(&sc[0], uc.pe);
  if (uc.se) atomicAdd(&sc[1], uc.se);
  if (uc.tot) atomicAdd(&sc[2], uc.tot);
  if (uc.reads) atomicAdd(&sc[3], uc.reads);
  if (uc.tooMany) atomicAdd(&sc[4], uc.tooMany);
  if (uc.mapped) atomicAdd(&sc[5], uc.mapped);
  __syncthreads();
  if (threadIdx.x < 6 && sc[threadIdx.x]) atomicAdd((unsigned long long*)&P.counters[threadIdx.x], sc[threadIdx.x]);
}

// stage B pass 2: write the hits in CSR order.  A thread per unit merges (or, for a pair the pair kernel merged, expands) into the wavefront's
// LDS stage -- the 64 units of a wavefront own ONE contiguous stretch of P.hits, 6.5 KB on the benchmark's input -- and the wavefront copies the
// stretch out 16 bytes per lane, consecutive lanes to consecutive addresses: a thread per unit writing its own 100 bytes issued a store per
// hit and lane to 64 different lines (1.05 ms per 5 M pairs at 1.6 TB/s; profiles/r06/timeline.sh).  A wavefront whose units hold more
// than the stage takes the direct path.
#define QM_PW_CAP 320
__global__ __launch_bounds__(256) void qm_pair_write_kernel(PairBatch P) {
  __shared__ __attribute__((aligned(16))) qm_hit stage[4][QM_PW_CAP];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = u < P.n ? (int)P.cnt[u] : 0;
  int incl = c;
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
  const int total = __shfl(incl, 63);
  if (total == 0) return;
  if (total > QM_PW_CAP) {
    if (c > 0) unit_merge(P, u, P.hits + P.offs[u], c, nullptr);
    return;
  }
  if (c > 0) unit_merge(P, u, &stage[wave][incl - c], c, nullptr);
  const long long base = __shfl(c > 0 ? P.offs[u] - (long long)(incl - c) : 0LL, __builtin_ctzll(__ballot(c > 0)));   // where the wavefront's stretch starts
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // (the merges' stores went through generic pointers: they are in LDS before any lane reads them back)
  __builtin_amdgcn_wave_barrier();
  const uint4* src = (const uint4*)&stage[wave][0];
  uint4* dst = (uint4*)(P.hits + base);
  for (int i = lane; i < 2 * total; i += 64) dst[i] = src[i];
}

// -s: per-unit temp slots needed = (list words of the unit) / 3 + 1 (every group has at least three words)
__global__ __launch_bounds__(256) void qm_sel_slots_kernel(PairBatch P) {
  long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u > P.n) return;
  u32 w = 0;
  if (u < P.n) w = P.paired ? (P.lcnt[2 * u] & 0x7fffffffu) + (P.lcnt[2 * u + 1] & 0x7fffffffu) : (P.lcnt[u] & 0x7fffffffu);
  P.cnt[u] = u < P.n ? w / 3 + 1 : 0;
}

// -s stages B + C: plan (per unit) -> ksw2 alignments, four per wavefront -> finish (per unit)
QM_DEV void sel_flush_counters(unsigned long long* sc, const UnitCounters& uc, u64* counters) {
  if (uc.pe) atomicAdd(&sc[0], uc.pe);
  if (uc.se) atomicAdd(&sc[1], uc.se);
  if (uc.tot) atomicAdd(&sc[2], uc.tot);
  if (uc.reads) atomicAdd(&sc[3], uc.reads);
  if (uc.tooMany) atomicAdd(&sc[4], uc.tooMany);
  if (uc.mapped) atomicAdd(&sc[5], uc.mapped);
  __syncthreads();
  if (threadIdx.x < 6 && sc[threadIdx.x]) atomicAdd((unsigned long long*)&counters[threadIdx.x], sc[threadIdx.x]);
}
// plan, step 1: a thread per unit; the wavefront reserves the room of its units' questions with one atomic
__global__ __launch_bounds__(256) void qm_sel_sides_kernel(PairBatch P, SelBatch A) {
  __shared__ unsigned long long sc[6];
  if (threadIdx.x < 6) sc[threadIdx.x] = 0;
  __syncthreads();
  const long long u = A.u0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  UnitCounters uc = {0, 0, 0, 0, 0, 0};
  const int m = u < A.u1 ? sel_unit_sides_count(P, A, u, &uc) : 0;
  const int lane = (int)(threadIdx.x & 63);
  int incl = m;
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
  const int total = __shfl(incl, 63);
  unsigned long long base = 0;
  if (lane == 63 && total > 0) base = atomicAdd((unsigned long long*)A.nsides, (unsigned long long)total);
  base = (unsigned long long)__shfl((long long)base, 63);
  if (m > 0) sel_unit_sides_write(P, A, u, (long long)base + incl - m);
  sel_flush_counters(sc, uc, P.counters);
}
// plan, step 2: a group of lanes per question (qm_sel_score_kernel<G>); the groups' sums / xors by DPP
struct SelRed16 {                       // all-reduce over a row of 16 lanes by rotations (DPP row_ror:8, 4, 2, 1): every lane ends up with the result
  template <int CTRL> static QM_DEV int rot(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
  QM_DEV int add(int v) const { v += rot<0x128>(v); v += rot<0x124>(v); v += rot<0x122>(v); v += rot<0x121>(v); return v; }
  QM_DEV int min(int v) const { int o; o = rot<0x128>(v); v = o < v ? o : v; o = rot<0x124>(v); v = o < v ? o : v; o = rot<0x122>(v); v = o < v ? o : v; o = rot<0x121>(v); v = o < v ? o : v; return v; }
  QM_DEV int max(int v) const { int o; o = rot<0x128>(v); v = o > v ? o : v; o = rot<0x124>(v); v = o > v ? o : v; o = rot<0x122>(v); v = o > v ? o : v; o = rot<0x121>(v); v = o > v ? o : v; return v; }
  QM_DEV u64 bxor(u64 v) const {
    int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    lo ^= rot<0x128>(lo); hi ^= rot<0x128>(hi); lo ^= rot<0x124>(lo); hi ^= rot<0x124>(hi);
    lo ^= rot<0x122>(lo); hi ^= rot<0x122>(hi); lo ^= rot<0x121>(lo); hi ^= rot<0x121>(hi);
    return (u64)(unsigned)lo | ((u64)(unsigned)hi << 32);
  }
};
// ... or over the eight lanes of a half row: row_half_mirror (lane i <-> 7 - i), then the two quad permutations
struct SelRed8 {
  template <int CTRL> static QM_DEV int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
  QM_DEV int add(int v) const { v += dpp<0x141>(v); v += dpp<0xB1>(v); v += dpp<0x4E>(v); return v; }
  QM_DEV int min(int v) const { int o; o = dpp<0x141>(v); v = o < v ? o : v; o = dpp<0xB1>(v); v = o < v ? o : v; o = dpp<0x4E>(v); v = o < v ? o : v; return v; }
  QM_DEV int max(int v) const { int o; o = dpp<0x141>(v); v = o > v ? o : v; o = dpp<0xB1>(v); v = o > v ? o : v; o = dpp<0x4E>(v); v = o > v ? o : v; return v; }
  QM_DEV u64 bxor(u64 v) const {
    int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    lo ^= dpp<0x141>(lo); hi ^= dpp<0x141>(hi); lo ^= dpp<0xB1>(lo); hi ^= dpp<0xB1>(hi); lo ^= dpp<0x4E>(lo); hi ^= dpp<0x4E>(hi);
    return (u64)(unsigned)lo | ((u64)(unsigned)hi << 32);
  }
};
struct SelRed4 {                       // ... or the four lanes of a quad
  template <int CTRL> static QM_DEV int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
  QM_DEV int add(int v) const { v += dpp<0xB1>(v); v += dpp<0x4E>(v); return v; }
  QM_DEV int min(int v) const { int o; o = dpp<0xB1>(v); v = o < v ? o : v; o = dpp<0x4E>(v); v = o < v ? o : v; return v; }
  QM_DEV int max(int v) const { int o; o = dpp<0xB1>(v); v = o > v ? o : v; o = dpp<0x4E>(v); v = o > v ? o : v; return v; }
  QM_DEV u64 bxor(u64 v) const {
    int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    lo ^= dpp<0xB1>(lo); hi ^= dpp<0xB1>(hi); lo ^= dpp<0x4E>(lo); hi ^= dpp<0x4E>(hi);
    return (u64)(unsigned)lo | ((u64)(unsigned)hi << 32);
  }
};
struct SelRed2 {                       // ... or a pair of lanes
  template <int CTRL> static QM_DEV int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
  QM_DEV int add(int v) const { return v + dpp<0xB1>(v); }
  QM_DEV int min(int v) const { const int o = dpp<0xB1>(v); return o < v ? o : v; }
  QM_DEV int max(int v) const { const int o = dpp<0xB1>(v); return o > v ? o : v; }
  QM_DEV u64 bxor(u64 v) const { int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32); lo ^= dpp<0xB1>(lo); hi ^= dpp<0xB1>(hi); return (u64)(unsigned)lo | ((u64)(unsigned)hi << 32); }
};
// Lanes per alignment question (measured on 2 x 100 bp, ms per chunk of 10.7 M questions: 16 lanes 4.08, 8 lanes 3.07, 4 lanes 2.42, 2 lanes 2.09,
// 1 lane 4.12): the geometry and bookkeeping of a question is per lane whatever the group, so few lanes with several 8-character words each
// beat one word per lane -- until a lane's loads no longer share lines with its neighbours'.  Longer reads take wider groups.
template <int G>
__global__ __launch_bounds__(256) void qm_sel_score_kernel(PairBatch P, SelBatch A) {
  const unsigned long long n = *A.nsides;
  constexpr int PER = 256 / G;
  const int l = (int)(threadIdx.x & (G - 1));
  typename std::conditional<G == 16, SelRed16, typename std::conditional<G == 8, SelRed8, typename std::conditional<G == 4, SelRed4, SelRed2>::type>::type>::type red;
  for (unsigned long long x = (unsigned long long)blockIdx.x * PER + (threadIdx.x / G); x < n; x += (unsigned long long)gridDim.x * PER)
    sel_side_score<G>(P, A, (long long)x, l, red);
}
// the strip alignments (sel_tasks_strip): four per wavefront, sixteen lanes each
__global__ __launch_bounds__(256) void qm_sel_strip_kernel(SelBatch A) {
  __shared__ StripMem mem[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned long long nt = *A.ntasks2;
  for (unsigned long long t = ((unsigned long long)blockIdx.x * 4 + wave) * 4; t < nt; t += (unsigned long long)gridDim.x * 16) sel_tasks_strip(A, t, nt, mem[wave]);
}
// plan, step 3: a thread per question
__global__ __launch_bounds__(256) void qm_sel_dedupe_kernel(SelBatch A) {
  const unsigned long long n = *A.nsides;
  const int lane = (int)(threadIdx.x & 63u);
  const unsigned long long below = (1ULL << lane) - 1ULL;
  for (unsigned long long x0 = (unsigned long long)blockIdx.x * 256 + (threadIdx.x & ~63u); x0 < n; x0 += (unsigned long long)gridDim.x * 256) {
    const unsigned long long x = x0 + (unsigned long long)lane;
    const int w = x < n ? sel_side_dedupe_list(A, (long long)x) : 0;
    // the work lists: a wavefront's questions take consecutive entries, one addition per wavefront and list
    const unsigned long long m1 = __ballot(w == 1), m2 = __ballot(w == 2);
    if (m1) {
      const int lead = __builtin_ctzll(m1);
      unsigned long long b = 0;
      if (lane == lead) b = atomic_add_u64(A.ntasks, (u64)__builtin_popcountll(m1));
      b = __shfl(b, lead);
      if (w == 1) A.torder[b + (unsigned long long)__builtin_popcountll(m1 & below)] = x;
    }
    if (m2) {
      const int lead = __builtin_ctzll(m2);
      unsigned long long b = 0;
      if (lane == lead) b = atomic_add_u64(A.ntasks2, (u64)__builtin_popcountll(m2));
      b = __shfl(b, lead);
      if (w == 2) A.torder2[b + (unsigned long long)__builtin_popcountll(m2 & below)] = x;
    }
  }
}
// one row of 16 lanes per ksw2 alignment, four alignments per wavefront (sel_ksw_extz2_rows); RING = column slots per
// alignment, chosen from --dpBandwidth at launch (sel_ksw_ring_slots)
// MAXLEN: QM_KSW_MAXLEN, or QM_KSW_MAXLEN_LONG for a batch with reads beyond QM_MAX_READ_LEN (images of 2 120 bytes)
template <int RING, int WAVES, int MAXLEN = QM_KSW_MAXLEN>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void qm_sel_align_kernel(PairBatch P, SelBatch A) {
  __shared__ KswRowT<RING, MAXLEN> rows[WAVES][4];
  __shared__ unsigned char codes[512];
  sel_ksw_fill_codes((QM_LDS(unsigned char)*)codes, (int)threadIdx.x, 64 * WAVES);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned long long nt = *A.ntasks;
  for (unsigned long long t = ((unsigned long long)blockIdx.x * WAVES + wave) * 4; t < nt; t += (unsigned long long)gridDim.x * (4 * WAVES))
    sel_tasks_align_rows<RING, MAXLEN>(P, A, t, nt, rows[wave], (const QM_LDS(unsigned char)*)codes);
}
// reads beyond QM_MAX_READ_LEN under a band beyond 97: the same row kernel on blocks in device memory (QM_KSW_RING_GMEM)
#define QMK_GMEM_WAVES 2
__global__ __launch_bounds__(64 * QMK_GMEM_WAVES) void qm_sel_align_gmem_kernel(PairBatch P, SelBatch A) {
  typedef KswRowT<QM_KSW_RING_GMEM, QM_KSW_MAXLEN_LONG> Row;
  __shared__ unsigned char codes[512];
  sel_ksw_fill_codes((QM_LDS(unsigned char)*)codes, (int)threadIdx.x, 64 * QMK_GMEM_WAVES);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  Row* rows = (Row*)A.ksw_rows + ((size_t)blockIdx.x * QMK_GMEM_WAVES + (size_t)wave) * 4;
  const unsigned long long nt = *A.ntasks;
  for (unsigned long long t = ((unsigned long long)blockIdx.x * QMK_GMEM_WAVES + wave) * 4; t < nt; t += (unsigned long long)gridDim.x * (4 * QMK_GMEM_WAVES))
    sel_tasks_align_rows<QM_KSW_RING_GMEM, QM_KSW_MAXLEN_LONG>(P, A, t, nt, rows, (const QM_LDS(unsigned char)*)codes);
}
// the register edition for reads of up to QM_MAX_READ_LEN, eight alignments per wavefront (sel_tasks_align_rows2): 37.9 KB of LDS per
// block, four blocks per CU -- as many alignments in flight as eight blocks of the four-per-wavefront kernel, half its scalar work
template <int MAXLEN>
__global__ __launch_bounds__(256) void qm_sel_align2_kernel(PairBatch P, SelBatch A) {
  __shared__ KswRowT<32, MAXLEN> rows[4][8];
  __shared__ unsigned char codes[512];
  sel_ksw_fill_codes((QM_LDS(unsigned char)*)codes, (int)threadIdx.x, 256);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned long long nt = *A.ntasks;
  for (unsigned long long t = ((unsigned long long)blockIdx.x * 4 + wave) * 8; t < nt; t += (unsigned long long)gridDim.x * 32)
    sel_tasks_align_rows2<MAXLEN>(P, A, t, nt, rows[wave], (const QM_LDS(unsigned char)*)codes);
}
__global__ __launch_bounds__(256) void qm_sel_finish_kernel(PairBatch P, SelBatch A) {
  __shared__ unsigned long long sc[6];
  if (threadIdx.x < 6) sc[threadIdx.x] = 0;
  __syncthreads();
  const long long u = A.u0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  UnitCounters uc = {0, 0, 0, 0, 0, 0};
  if (u < A.u1) P.cnt[u] = (u32)sel_unit_finish(P, A, u, &uc);
  sel_flush_counters(sc, uc, P.counters);
}

// -s, stage entry "merge only": mergeLeftRightHitsFuzzy on the position lists into the per-unit temp slots (chain statuses
// stay parked in aln_score), no alignment
__global__ __launch_bounds__(256) void qm_sel_merge_kernel(PairBatch P, SelBatch A) {
  __shared__ unsigned long long sc[6];
  if (threadIdx.x < 6) sc[threadIdx.x] = 0;
  __syncthreads();
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  UnitCounters uc = {0, 0, 0, 0, 0, 0};
  if (u < P.n) P.cnt[u] = (u32)sel_unit_merge(P, A, u, &uc);
  sel_flush_counters(sc, uc, P.counters);
}

// -s: the reads stage A left on the slow queue (lcnt == QM_LCNT_SLOW), gathered into q[0 .. *count)
__global__ __launch_bounds__(256) void qm_collect_slow_kernel(const u32* lcnt, long long nreads, long long* q, u64* count, u32 mark) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool m = r < nreads && lcnt[r] == mark;
  // one addition per wavefront (a batch with N's queues a tenth of its reads: 380 k additions to ONE address were 0.75 ms of a 9 ms step),
  // and a wavefront's reads stay neighbours in the queue -- the N-aware pass maps two slots per wavefront
  const unsigned long long b = __ballot(m);
  if (b) {
    const int lane = (int)(threadIdx.x & 63u), lead = __builtin_ctzll(b);
    unsigned long long base = 0;
    if (lane == lead) base = atomicAdd((unsigned long long*)count, (unsigned long long)__builtin_popcountll(b));
    base = __shfl(base, lead);
    if (m) q[base + (unsigned long long)__builtin_popcountll(b & ((1ULL << lane) - 1ULL))] = r;
  }
}

// -s: surviving hits from the per-unit temp slots to CSR order
__global__ __launch_bounds__(256) void qm_sel_compact_kernel(PairBatch P, const qm_hit* tmp, const long long* toff) {
  // (through the wavefront's LDS stage and out coalesced, like qm_pair_write_kernel)
  __shared__ __attribute__((aligned(16))) qm_hit stage[4][QM_PW_CAP];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = u < P.n ? (int)P.cnt[u] : 0;
  int incl = c;
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
  const int total = __shfl(incl, 63);
  if (total == 0) return;
  if (total > QM_PW_CAP) {
    for (int i = 0; i < c; ++i) P.hits[P.offs[u] + i] = tmp[toff[u] + i];
    return;
  }
  for (int i = 0; i < c; ++i) stage[wave][incl - c + i] = tmp[toff[u] + i];
  const long long base = __shfl(c > 0 ? P.offs[u] - (long long)(incl - c) : 0LL, __builtin_ctzll(__ballot(c > 0)));
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const uint4* src = (const uint4*)&stage[wave][0];
  uint4* dst = (uint4*)(P.hits + base);
  for (int i = lane; i < 2 * total; i += 64) dst[i] = src[i];
}

__global__ void build_sainfo_kernel(const u32* SA, long long nSA, const u32* offsets, long long T, SaInfo* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < nSA; i += stride) {
    const u32 p = SA[i];
    long long lo = 0, hi = T;   // upper_bound(offsets, p) - 1  == #'$' in text[0,p)
    while (lo < hi) { long long mid = (lo + hi) >> 1; if (offsets[mid] <= p) lo = mid + 1; else hi = mid; }
    long long tid = lo - 1;
    SaInfo e; e.tid = (u32)tid; e.pos = (int)(p - offsets[tid]);
    out[i] = e;
  }
}

// ---- the extension tables (saext_entry / saext2_entry / sanext_entry of every suffix-array entry).  Round 5 built them a thread per
// suffix and a byte load per character: 399 ms and 1.36 TB fetched to write the 8.3 GB of SaExt (82 sectors per entry).  Round 6 packs
// the text ONCE -- 2 bits per character in the entries' own layout (first character in the top bits of a word) plus one validity bit per
// character (A C G T) -- and an entry is then a funnel shift out of four to eight consecutive words of that image (65 + 33 MB for config
// 2: it stays in the last-level cache) and a count of leading validity bits: two or three sectors per entry instead of 82.
// T[w]: characters [32 w, 32 w + 32); V[w]: characters [64 w, 64 w + 64), bit 63 - t = character 64 w + t is A C G T and below n
__global__ void pack_text_kernel(const unsigned char* text, long long n, u64* T, u64* V, long long nv64) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; w < nv64; w += stride) {
    u64 t0 = 0, t1 = 0, v = 0;
    const long long base = 64 * w;
#pragma unroll 4
    for (int t = 0; t < 64; t += 8) {
      u64 ch = 0;
      if (base + t + 8 <= n) ch = load_u64_unaligned(text + base + t);
      else { for (int b = 0; b < 8; ++b) if (base + t + b < n) ch |= (u64)text[base + t + b] << (8 * b); }
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned c = (unsigned)(ch >> (8 * b)) & 0xffu;
        const bool ok = c == 'A' || c == 'C' || c == 'G' || c == 'T';
        const u64 x = (c >> 1) & 3u;
        const u64 code = ok ? (x ^ (x >> 1)) : 0ULL;
        const int tt = t + b;
        if (tt < 32) t0 |= code << (62 - 2 * tt); else t1 |= code << (62 - 2 * (tt - 32));
        v |= (u64)(ok ? 1 : 0) << (63 - tt);
      }
    }
    T[2 * w] = t0; T[2 * w + 1] = t1; V[w] = v;
  }
}
// NW words of packed characters from character p on (the image is padded with zero words), and how many of the first `bases` are valid
template <int NW>
QM_DEV int packed_entry_words(const u64* T, const u64* V, long long p, int bases, u64* w) {
  const long long j = p >> 5; const int sh = 2 * (int)(p & 31);
  u64 prev = T[j];
#pragma unroll
  for (int t = 0; t < NW; ++t) { const u64 nx = T[j + t + 1]; w[t] = (prev << sh) | ((nx >> 1) >> (63 - sh)); prev = nx; }
  // leading valid characters: validity bits from p on, 64 at a time
  const long long jv = p >> 6; const int sv = (int)(p & 63);
  int nv = 0;
  u64 pv = V[jv];
#pragma unroll
  for (int t = 0; t < (64 * NW / 2 + 63) / 64 + 1; ++t) {           // (NW words = 32 NW characters)
    const u64 nx = V[jv + t + 1];
    const u64 x = (pv << sv) | ((nx >> 1) >> (63 - sv));
    const int run = x == ~0ULL ? 64 : __builtin_clzll(~x);
    if (nv == 64 * t) nv += run;
    pv = nx;
  }
  nv = nv < bases ? nv : bases;
#pragma unroll
  for (int t = 0; t < NW; ++t) {
    const int keep = nv - 32 * t;
    w[t] = keep >= 32 ? w[t] : (keep <= 0 ? 0ULL : (w[t] & (~0ULL << (64 - 2 * keep))));
  }
  return nv;
}
__global__ void build_saext_kernel(const u64* T, const u64* V, long long n, const u32* SA, long long nSA, int k, const SaInfo* sainfo, SaExt* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < nSA; i += stride) {
    const SaInfo si = sainfo[i];
    long long p = (long long)SA[i] + k; if (p > n) p = n;            // (behind the text: nothing valid)
    SaExt e; u64 w[3];
    const int nv = packed_entry_words<3>(T, V, p, QM_EXT_BASES, w);
    e.w[0] = w[0]; e.w[1] = w[1]; e.w[2] = w[2]; e.pos = si.pos;
    e.tidnv = (si.tid & ((1u << QM_EXT_TID_BITS) - 1)) | ((u32)nv << QM_EXT_TID_BITS);
    out[i] = e;
  }
}
__global__ void build_saext2_kernel(const u64* T, const u64* V, long long n, const u32* SA, long long nSA, int k, const SaInfo* sainfo, SaExt2* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < nSA; i += stride) {
    const SaInfo si = sainfo[i];
    long long p = (long long)SA[i] + k; if (p > n) p = n;
    SaExt2 e; u64 w[7];
    const int nv = packed_entry_words<7>(T, V, p, QM_EXT2_BASES, w);
#pragma unroll
    for (int t = 0; t < 7; ++t) e.w[t] = w[t];
    e.pos = si.pos;
    e.tidnv = (si.tid & ((1u << QM_EXT2_TID_BITS) - 1)) | ((u32)nv << QM_EXT2_TID_BITS);
    out[i] = e;
  }
}
// -s: the text characters behind the k-mer of every suffix (sanext_entry)
__global__ void build_sanext_kernel(const u64* T, const u64* V, long long n, const u32* SA, long long nSA, int k, u32* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < nSA; i += stride) {
    long long p = (long long)SA[i] + k; if (p > n) p = n;
    u64 w[1];
    const int nv = packed_entry_words<1>(T, V, p, QM_NEXT_BASES, w);
    out[i] = (u32)(w[0] >> 36) | ((u32)nv << 28);                      // (14 characters at 2 bits: the top 28 bits of the word, in bits 27 .. 0)
  }
}
// the byte-per-character builders of rounds 3-5: kept as the reference the packed builders are checked against (QM_TABLE_CHECK=1: tests)
__global__ void build_saext_ref_kernel(const unsigned char* text, long long n, const u32* SA, long long nSA, int k, const SaInfo* sainfo, const SaExt* got, u64* bad) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < nSA; i += stride) {
    const SaInfo si = sainfo[i]; const SaExt e = saext_entry(text, n, (long long)SA[i] + k, si.tid, si.pos); const SaExt g = got[i];
    if (e.w[0] != g.w[0] || e.w[1] != g.w[1] || e.w[2] != g.w[2] || e.tidnv != g.tidnv || e.pos != g.pos) atomicAdd((unsigned long long*)bad, 1ULL);
  }
}
__global__ void build_saext2_ref_kernel(const unsigned char* text, long long n, const u32* SA, long long nSA, int k, const SaInfo* sainfo, const SaExt2* got, u64* bad) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < nSA; i += stride) {
    const SaInfo si = sainfo[i]; const SaExt2 e = saext2_entry(text, n, (long long)SA[i] + k, si.tid, si.pos); const SaExt2 g = got[i];
    bool same = e.tidnv == g.tidnv && e.pos == g.pos;
    for (int t = 0; t < 7; ++t) same = same && e.w[t] == g.w[t];
    if (!same) atomicAdd((unsigned long long*)bad, 1ULL);
  }
}
__global__ void build_sanext_ref_kernel(const unsigned char* text, long long n, const u32* SA, long long nSA, int k, const u32* got, u64* bad) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < nSA; i += stride) if (sanext_entry(text, n, (long long)SA[i] + k) != got[i]) atomicAdd((unsigned long long*)bad, 1ULL);
}

// records: K x {u64 key, u32 lb, u32 ub}: the records of hash.bin (a BigSA index's int64 pairs narrowed by the loader)
__global__ void build_slots_kernel(const Slot* recs, long long K, Bucket* buckets, u64 hmask, int k) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < K; i += stride) {
    Slot r = recs[i];
    bucket_insert(buckets, hmask, r.key, k, (u32)r.lb, (u32)r.ub,
                  [](u64* p, u64 cmp, u64 val) { return (u64)atomicCAS((unsigned long long*)p, (unsigned long long)cmp, (unsigned long long)val); },
                  [](u64* p, u64 v) { atomicOr((unsigned long long*)p, (unsigned long long)v); });
  }
}

// perfect-hash value records: {k-mer word at text[SA[data_[i]]] (partial word if a '$' is hit), data_[i], lens_[i]}
__global__ void build_phrec_kernel(const u32* data, const unsigned char* lens, long long n, DevIndex ix, PhRec* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    PhRec r; r.data = data[i]; r.len = lens[i]; r.pad[0] = r.pad[1] = r.pad[2] = 0;
    u64 m = 0;
    if ((long long)r.data < ix.nSA) text_kmer(ix, (long long)ix.SA[r.data], ix.k, m);
    r.key = m;
    out[i] = r;
  }
}

// membership pre-filter of the compact -p image (PhIndex::filter): four bits per key of the index
__global__ void build_phfilter_kernel(const PhRec* recs, long long n, unsigned long long* filter, u64 mask, int k) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    u64 w, bits;
    ph_filter_slot(recs[i].key, word_rc(recs[i].key, k), mask, w, bits);
    atomicOr(&filter[w], (unsigned long long)bits);
  }
}

// A perfect-hash index expanded into the dense bucket table: slot i of the MPHF holds the interval [data_[i], data_[i] +
// lens_[i]) of the k-mer that starts the suffix SA[data_[i]] -- every (k-mer, interval) pair of the index, i.e. exactly
// the contents of a dense hash.bin.  Each record is first looked up through the BooPHF walk itself (find_kmer<QM_F_PH>, what
// FrugalBooMap::find does with this file): it must come back with its own interval, so the table is known to answer like
// the on-disk structure; `bad` counts the records for which it does not.
__global__ void build_slots_from_ph_kernel(DevIndex ix, long long n, Bucket* buckets, u64 hmask, unsigned long long* bad) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  const PhIndex& P = ix.phv;
  for (; i < n; i += stride) {
    const PhRec r = P.recs[i];
    u32 lb = 0, ub = 0;
    const bool found = find_kmer<QM_F_PH>(ix, r.key, lb, ub);
    if (!found || lb != r.data) { atomicAdd(bad, 1ULL); continue; }
    bucket_insert(buckets, hmask, r.key, ix.k, lb, ub,
                  [](u64* p, u64 cmp, u64 val) { return (u64)atomicCAS((unsigned long long*)p, (unsigned long long)cmp, (unsigned long long)val); },
                  [](u64* p, u64 v) { atomicOr((unsigned long long*)p, (unsigned long long)v); });
  }
}

// 2-bit packed reads (include/qmap_mi355.h): thread (read r, group g) turns packed byte (off[r] >> 2) + r + g into the characters
// off[r] + 4g .. + 3 of the ASCII image; exceptions (anything but upper-case A C G T) are written over it afterwards
// (eight threads per read, each striding over the read's groups: a launch sized by the LONGEST read of the batch -- one 2048-character
// read among 2^19 reads of 100 -- would start twenty times the threads the batch needs)
#define QM_UNPACK_TPR 8
__global__ __launch_bounds__(256) void qm_unpack_kernel(const unsigned char* packed, const long long* off, long long n, unsigned char* seq) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long r = t / QM_UNPACK_TPR; const int sub = (int)(t - r * QM_UNPACK_TPR);
  if (r >= n) return;
  const long long o = off[r]; const int len = (int)(off[r + 1] - o);
  const unsigned lut = 0x54474341u;                       // 'A' 'C' 'G' 'T' from the low byte up
  for (int g = sub; 4 * g < len; g += QM_UNPACK_TPR) {
    const unsigned b = packed[(o >> 2) + r + g];
    unsigned char* d = seq + o + 4 * g;
    const int m = len - 4 * g < 4 ? len - 4 * g : 4;
    for (int j = 0; j < m; ++j) d[j] = (unsigned char)(lut >> (8 * ((b >> (2 * j)) & 3u)));
  }
}
// (an exception whose position lies outside the batch's characters is ignored: the list is caller-supplied)
__global__ __launch_bounds__(256) void qm_unpack_exc_kernel(const qm_pack_exc* exc, long long n, unsigned char* seq, unsigned long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n && (unsigned long long)exc[t].pos < total) seq[exc[t].pos] = (unsigned char)exc[t].ch;
}

struct U32ToI64 { __device__ long long operator()(u32 x) const { return (long long)x; } };
struct U32MaskToI64 { __device__ long long operator()(u32 x) const { return (long long)(x & 0x7fffffffu); } };

// qm_fetch_stages: read r's SA-interval records and list words leave their bump-allocated chunks (one open chunk per wave: the
// used part of those buffers is hundreds of MB for a batch of 20 000 reads) for CSR order -- four threads per read
__global__ __launch_bounds__(256) void qm_stage_gather_kernel(long long nreads, const u32* ivcnt, const long long* ivoff, const qm_sa_interval_hit* iv,
                                                              const long long* ivcsr, qm_sa_interval_hit* ivOut, const u32* lcnt, const long long* loff,
                                                              const u64* lists, const long long* lcsr, u64* wordsOut) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long r = t >> 2; const int sub = (int)(t & 3);
  if (r >= nreads) return;
  const int ni = (int)ivcnt[r];
  if (ni > 0) {
    const u32* src = (const u32*)(iv + ivoff[r]); u32* dst = (u32*)(ivOut + ivcsr[r]);
    for (int i = sub; i < 5 * ni; i += 4) dst[i] = src[i];          // 20-byte records
  }
  const int nl = (int)(lcnt[r] & 0x7fffffffu);
  if (nl > 0) {
    const u64* src = lists + loff[r]; u64* dst = wordsOut + lcsr[r];
    for (int i = sub; i < nl; i += 4) dst[i] = src[i];
  }
}

}  // namespace qm

using namespace qm;

extern "C" {

hipError_t qmk_build_sainfo(const unsigned int* SA, long long nSA, const unsigned int* offsets, long long T, void* out, hipStream_t st) {
  hipLaunchKernelGGL(build_sainfo_kernel, dim3(4096), dim3(256), 0, st, SA, nSA, offsets, T, (SaInfo*)out);
  return hipGetLastError();
}

// the packed image of the text for the three builders below: allocated, filled and (after the builder) freed per call -- a replica builds
// each table once.  -> T (n / 32 + 16 words), V (n / 64 + 8 words) in one allocation
static hipError_t packed_text(const unsigned char* text, long long n, u64** T, u64** V, hipStream_t st) {
  const long long nv64 = (n + 63) / 64, nT = 2 * nv64 + 16, nV = nv64 + 8;
  u64* p = nullptr;
  hipError_t e = hipMalloc((void**)&p, (size_t)(nT + nV) * sizeof(u64));
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(p, 0, (size_t)(nT + nV) * sizeof(u64), st);
  if (e != hipSuccess) { hipFree(p); return e; }
  *T = p; *V = p + nT;
  if (nv64 > 0) hipLaunchKernelGGL(pack_text_kernel, dim3((unsigned)((nv64 + 255) / 256 < 16384 ? (nv64 + 255) / 256 : 16384)), dim3(256), 0, st, text, n, *T, *V, nv64);
  return hipGetLastError();
}
static bool table_check() { const char* e = getenv("QM_TABLE_CHECK"); return e && atoi(e) != 0; }   // (read at every build: tests switch it on for one context)
static hipError_t table_verdict(u64* d_bad, hipStream_t st) {      // QM_TABLE_CHECK: entries that differ from the byte-per-character builder's
  unsigned long long bad = 0;
  hipError_t e = hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(d_bad);
  if (e == hipSuccess && bad) { fprintf(stderr, "[qm] QM_TABLE_CHECK: %llu table entries differ from the reference builder's\n", bad); return hipErrorAssert; }
  return e;
}
hipError_t qmk_build_saext(const unsigned char* text, long long n, const unsigned int* SA, long long nSA, int k, const void* sainfo, void* out, hipStream_t st) {
  if (nSA <= 0) return hipSuccess;
  u64 *T = nullptr, *V = nullptr;
  hipError_t e = packed_text(text, n, &T, &V, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(build_saext_kernel, dim3(8192), dim3(256), 0, st, (const u64*)T, (const u64*)V, n, SA, nSA, k, (const SaInfo*)sainfo, (SaExt*)out);
  e = hipGetLastError();
  if (e == hipSuccess && table_check()) {
    u64* d_bad = nullptr; hipMalloc((void**)&d_bad, 8); hipMemsetAsync(d_bad, 0, 8, st);
    hipLaunchKernelGGL(build_saext_ref_kernel, dim3(8192), dim3(256), 0, st, text, n, SA, nSA, k, (const SaInfo*)sainfo, (const SaExt*)out, d_bad);
    e = table_verdict(d_bad, st);
  }
  hipError_t e2 = hipStreamSynchronize(st);                  // (the image is freed: the builder has to be through)
  hipFree(T);
  return e != hipSuccess ? e : e2;
}
hipError_t qmk_build_saext2(const unsigned char* text, long long n, const unsigned int* SA, long long nSA, int k, const void* sainfo, void* out, hipStream_t st) {
  if (nSA <= 0) return hipSuccess;
  u64 *T = nullptr, *V = nullptr;
  hipError_t e = packed_text(text, n, &T, &V, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(build_saext2_kernel, dim3(8192), dim3(256), 0, st, (const u64*)T, (const u64*)V, n, SA, nSA, k, (const SaInfo*)sainfo, (SaExt2*)out);
  e = hipGetLastError();
  if (e == hipSuccess && table_check()) {
    u64* d_bad = nullptr; hipMalloc((void**)&d_bad, 8); hipMemsetAsync(d_bad, 0, 8, st);
    hipLaunchKernelGGL(build_saext2_ref_kernel, dim3(8192), dim3(256), 0, st, text, n, SA, nSA, k, (const SaInfo*)sainfo, (const SaExt2*)out, d_bad);
    e = table_verdict(d_bad, st);
  }
  hipError_t e2 = hipStreamSynchronize(st);
  hipFree(T);
  return e != hipSuccess ? e : e2;
}
size_t qmk_saext2_bytes(void) { return sizeof(SaExt2); }
hipError_t qmk_build_sanext(const unsigned char* text, long long n, const unsigned int* SA, long long nSA, int k, unsigned int* out, hipStream_t st) {
  if (nSA <= 0) return hipSuccess;
  u64 *T = nullptr, *V = nullptr;
  hipError_t e = packed_text(text, n, &T, &V, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(build_sanext_kernel, dim3(4096), dim3(256), 0, st, (const u64*)T, (const u64*)V, n, SA, nSA, k, out);
  e = hipGetLastError();
  if (e == hipSuccess && table_check()) {
    u64* d_bad = nullptr; hipMalloc((void**)&d_bad, 8); hipMemsetAsync(d_bad, 0, 8, st);
    hipLaunchKernelGGL(build_sanext_ref_kernel, dim3(4096), dim3(256), 0, st, text, n, SA, nSA, k, (const u32*)out, d_bad);
    e = table_verdict(d_bad, st);
  }
  hipError_t e2 = hipStreamSynchronize(st);
  hipFree(T);
  return e != hipSuccess ? e : e2;
}
hipError_t qmk_build_slots(const void* recs, long long K, void* slots, unsigned long long cap, int k, hipStream_t st) {
  hipError_t e = hipMemsetAsync(slots, 0xff, cap * sizeof(Bucket), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(build_slots_kernel, dim3(4096), dim3(256), 0, st, (const Slot*)recs, K, (Bucket*)slots, cap - 1, k);
  return hipGetLastError();
}

hipError_t qmk_build_slots_from_ph(const void* dev_index, long long n, void* slots, unsigned long long cap, unsigned long long* d_bad, hipStream_t st) {
  hipError_t e = hipMemsetAsync(slots, 0xff, cap * sizeof(Bucket), st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  if (n > 0) hipLaunchKernelGGL(build_slots_from_ph_kernel, dim3(4096), dim3(256), 0, st, *(const DevIndex*)dev_index, n, (Bucket*)slots, cap - 1, d_bad);
  return hipGetLastError();
}

hipError_t qmk_build_phrecs(const unsigned int* data, const unsigned char* lens, long long n, const void* dev_index, void* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(build_phrec_kernel, dim3(4096), dim3(256), 0, st, data, lens, n, *(const DevIndex*)dev_index, (PhRec*)out);
  return hipGetLastError();
}

hipError_t qmk_build_phfilter(const void* recs, long long n, void* filter, unsigned long long mask, int k, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(build_phfilter_kernel, dim3(4096), dim3(256), 0, st, (const PhRec*)recs, n, (unsigned long long*)filter, (u64)mask, k);
  return hipGetLastError();
}

int qmk_grid_oversub(void) {
  static const int m = [] { const char* e = getenv("QM_GRID_OVERSUB"); return e && atoi(e) > 0 && atoi(e) <= 16 ? atoi(e) : 4; }();
  return m;
}
// the compact -p kernels (BooPHF walked per lookup: reads differ more in work) keep gaining up to twelve times the
// resident blocks: 181.9 / 185.5 / 187.4 / 189.1 / 189.0 M pairs/s at 4 / 6 / 8 / 12 / 16 (profiles/r04/ph_oversub.txt)
int qmk_grid_oversub_ph(void) {
  static const int m = [] { const char* e = getenv("QM_GRID_OVERSUB_PH"); return e && atoi(e) > 0 && atoi(e) <= 16 ? atoi(e) : (getenv("QM_GRID_OVERSUB") ? qmk_grid_oversub() : 12); }();
  return m;
}
int qmk_resident_grid(long long nreads, int num_cu) {
  long long want = (nreads + 3) / 4;
  long long cap = (long long)num_cu * QMK_BLOCKS_PER_CU;
  return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}
int qmk_map_grid(long long nreads, int num_cu) { return qmk_map_grid_ex(nreads, num_cu, 0); }
int qmk_map_grid_ex(long long nreads, int num_cu, int ph_compact) {
  long long want = (nreads + 3) / 4;
  long long cap = (long long)num_cu * QMK_BLOCKS_PER_CU * (ph_compact ? qmk_grid_oversub_ph() : qmk_grid_oversub());
  return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}

// The stage-A kernels are persistent grids: every wave strides over the reads.  Launched with exactly the blocks that are
// resident at once, the reads are divided statically and the launch lasts as long as its slowest wave -- 13 % longer than the
// average one on config 2.  Launched with QM_GRID_OVERSUB (4) times that many, a block owns a quarter of the reads and the
// hardware hands out the blocks as slots free up: 41.6 -> 36.1 ms (profiles/r04/grid_oversub.txt; 8x: the same).  The occupancy
// of the chosen instantiation (VGPR/LDS dependent) decides the resident count.
// stage A: the slot-count class picks the translation unit that holds its instantiations (qm_read_kernel.inl).
// ns < 0: the "collector only" stage entry, on the eight-slot kernels (every read length up to QM_MAX_READ_LEN).
// collect: collector-only kernels (ns < 0: those of the stage entry, eight slots; otherwise the chain-scoring ones of slot count ns)
hipError_t qmk_map_reads(const void* ixp, const void* bp, int ns, int grid, int num_cu, hipStream_t st) { return qmk_map_reads_ex(ixp, bp, ns, 0, grid, num_cu, st); }
hipError_t qmk_map_reads_ex(const void* ixp, const void* bp, int ns, int collect, int grid, int num_cu, hipStream_t st) {
  if (ns == -32) return qmk_launch_reads_ns32(ixp, bp, 1, grid, num_cu, st);   // long-read pass of the collector-only stage entry
  if (ns == 32) return qmk_launch_reads_ns32(ixp, bp, 0, grid, num_cu, st);    // long-read pass (reads of 513 .. 2048 characters)
  if (ns < 0) return qmk_launch_reads_ns8(ixp, bp, 1, grid, num_cu, st);
  if (collect) {
    if (ns == 2) return qmk_launch_reads_ns2(ixp, bp, 1, grid, num_cu, st);
    if (ns == 3) return qmk_launch_reads_ns3(ixp, bp, 1, grid, num_cu, st);
    if (ns > 4) return qmk_launch_reads_ns8(ixp, bp, 1, grid, num_cu, st);
    return qmk_launch_reads_ns4(ixp, bp, 1, grid, num_cu, st);
  }
  if (ns == 2) return qmk_launch_reads_ns2(ixp, bp, 0, grid, num_cu, st);
  if (ns == 3) return qmk_launch_reads_ns3(ixp, bp, 0, grid, num_cu, st);
  if (ns > 4) return qmk_launch_reads_ns8(ixp, bp, 0, grid, num_cu, st);
  return qmk_launch_reads_ns4(ixp, bp, 0, grid, num_cu, st);
}

hipError_t qmk_h2m(const void* ixp, const void* bp, int grid, int num_cu, hipStream_t st) {
  const DevIndex& ix = *(const DevIndex*)ixp;
  const ReadBatch& B = *(const ReadBatch*)bp;
  static int nbSel = 0, nbPlain = 0;                         // persistent grids: no more blocks than are resident at once
  if (nbSel == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbSel, qm_h2m_kernel<QM_F_SEL>, 256, 0) != hipSuccess || nbSel < 1)) nbSel = 2;
  if (nbPlain == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbPlain, qm_h2m_kernel<0>, 256, 0) != hipSuccess || nbPlain < 1)) nbPlain = 2;
  const int nb = B.selscr ? nbSel : nbPlain;
  const unsigned g = (unsigned)(grid < num_cu * nb ? grid : num_cu * nb);
  if (B.selscr) hipLaunchKernelGGL(qm_h2m_kernel<QM_F_SEL>, dim3(g), dim3(256), 0, st, ix, B);
  else hipLaunchKernelGGL(qm_h2m_kernel<0>, dim3(g), dim3(256), 0, st, ix, B);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void qm_rebase_offsets_kernel(const long long* src, long long* dst, long long n, long long add) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i] + add;
}
hipError_t qmk_rebase_offsets(const long long* src, long long* dst, long long n, long long add, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qm_rebase_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n, add);
  return hipGetLastError();
}
hipError_t qmk_h2m_pack(const void* ixp, const void* bp, long long* todoq, int grid, int num_cu, hipStream_t st) {
  const DevIndex& ix = *(const DevIndex*)ixp;
  const ReadBatch& B = *(const ReadBatch*)bp;
  static int nb = 0;
  if (nb == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, qm_h2m_pack_kernel, 256, 0) != hipSuccess || nb < 1)) nb = 2;
  const long long cap = (long long)num_cu * nb * qmk_grid_oversub();                 // contiguous ranges of reads per wave: more, smaller ranges balance better
  const unsigned g = (unsigned)(grid < cap ? grid : cap);
  hipLaunchKernelGGL(qm_h2m_pack_kernel, dim3(g), dim3(256), 0, st, ix, B, todoq);
  return hipGetLastError();
}
hipError_t qmk_h2m_packw(const void* ixp, const void* bp, const long long* ids, const unsigned long long* nids, long long* todoq, int grid, int num_cu, hipStream_t st) {
  const DevIndex& ix = *(const DevIndex*)ixp;
  const ReadBatch& B = *(const ReadBatch*)bp;
  static int nb = 0;
  if (nb == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, qm_h2m_packw_kernel, 256, 0) != hipSuccess || nb < 1)) nb = 1;
  const long long cap = (long long)num_cu * nb * qmk_grid_oversub();
  const unsigned g = (unsigned)(grid < cap ? grid : cap);
  hipLaunchKernelGGL(qm_h2m_packw_kernel, dim3(g), dim3(256), 0, st, ix, B, ids, (const u64*)nids, todoq);
  return hipGetLastError();
}
hipError_t qmk_sel_merge(const void* pp, const void* ap, hipStream_t st) {
  const PairBatch& P = *(const PairBatch*)pp; const SelBatch& A = *(const SelBatch*)ap;
  if (P.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qm_sel_merge_kernel, dim3((unsigned)((P.n + 255) / 256)), dim3(256), 0, st, P, A);
  return hipGetLastError();
}
size_t qmk_sel_scratch_bytes(void) { return sizeof(SelScratch); }
size_t qmk_sel_dyn_struct_bytes(void) { return sizeof(SelScratchDyn); }
unsigned long long qmk_sel_dyn_bytes(long long n) { return SelScratchDyn::bytes_for(n); }
// host image of wave w's SelScratchDyn over device memory at `base`
void qmk_sel_dyn_bind(void* host_struct, void* dev_base, long long n) { ((SelScratchDyn*)host_struct)->bind((unsigned char*)dev_base, n); }
hipError_t qmk_collect_slow(const unsigned int* lcnt, long long nreads, long long* q, unsigned long long* count, hipStream_t st) {
  if (nreads <= 0) return hipSuccess;
  hipLaunchKernelGGL(qm_collect_slow_kernel, dim3((unsigned)((nreads + 255) / 256)), dim3(256), 0, st, lcnt, nreads, q, (u64*)count, (u32)QM_LCNT_SLOW);
  return hipGetLastError();
}
// the same for the reads qm_lean_kernel marked (QM_LCNT_LEAN)
hipError_t qmk_collect_lean(const unsigned int* lcnt, long long nreads, long long* q, unsigned long long* count, hipStream_t st) {
  if (nreads <= 0) return hipSuccess;
  hipLaunchKernelGGL(qm_collect_slow_kernel, dim3((unsigned)((nreads + 255) / 256)), dim3(256), 0, st, lcnt, nreads, q, (u64*)count, (u32)QM_LCNT_LEAN);
  return hipGetLastError();
}
hipError_t qmk_sel_slots(const void* pp, hipStream_t st) {
  const PairBatch& P = *(const PairBatch*)pp;
  hipLaunchKernelGGL(qm_sel_slots_kernel, dim3((unsigned)((P.n + 1 + 255) / 256)), dim3(256), 0, st, P);
  return hipGetLastError();
}
// the three steps of a chunk of units [A.u0, A.u1): plan (per unit) -> ksw2 (four alignments per wavefront) -> finish (per unit)
hipError_t qmk_sel_plan(const void* pp, const void* ap, int num_cu, hipStream_t st) {
  const PairBatch& P = *(const PairBatch*)pp; const SelBatch& A = *(const SelBatch*)ap;
  if (A.u1 <= A.u0) return hipSuccess;
  const long long nu = A.u1 - A.u0;
  hipLaunchKernelGGL(qm_sel_sides_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, st, P, A);
  // the two flat steps stride over a list whose length only the device knows: grids that fill the chip, no larger than the units could need
  const long long want = (long long)num_cu * 16;
  const unsigned gs = (unsigned)(nu / 8 + 1 < want ? nu / 8 + 1 : want),   /* (16 or 32 questions per block and trip) */ gd = (unsigned)(nu / 128 + 1 < want ? nu / 128 + 1 : want);
  static const int lanesEnv = [] { const char* e = getenv("QM_SCORE_LANES"); return e ? atoi(e) : 0; }();     // (tuning knob: 2, 4, 8 or 16)
  const int lanes = lanesEnv ? lanesEnv : (A.long_reads ? 8 : (A.short_len > 0 && A.short_len <= 256 ? 2 : 4));   // (2 x 150 and 2 x 250 bp: 2 lanes 55.1 / 19.8, 4 lanes 54.3 / 19.6, 8 lanes 53.5 / 19.6 M pairs/s)
  switch (lanes) {
    case 2: hipLaunchKernelGGL(qm_sel_score_kernel<2>, dim3(gs), dim3(256), 0, st, P, A); break;
    case 4: hipLaunchKernelGGL(qm_sel_score_kernel<4>, dim3(gs), dim3(256), 0, st, P, A); break;
    case 16: hipLaunchKernelGGL(qm_sel_score_kernel<16>, dim3(gs), dim3(256), 0, st, P, A); break;
    default: hipLaunchKernelGGL(qm_sel_score_kernel<8>, dim3(gs), dim3(256), 0, st, P, A); break;
  }
  hipLaunchKernelGGL(qm_sel_dedupe_kernel, dim3(gd), dim3(256), 0, st, A);
  if (A.torder2) hipLaunchKernelGGL(qm_sel_strip_kernel, dim3((unsigned)(num_cu * 8)), dim3(256), 0, st, A);   // (17 KB of LDS per block: up to nine blocks per CU)
  return hipGetLastError();
}
size_t qmk_sel_side_bytes(void) { return sizeof(SelSide); }
hipError_t qmk_sel_align_finish(const void* pp, const void* ap, int num_cu, hipStream_t st) {
  const PairBatch& P = *(const PairBatch*)pp; const SelBatch& A = *(const SelBatch*)ap;
  if (A.u1 <= A.u0) return hipSuccess;
  const unsigned nb = (unsigned)((A.u1 - A.u0 + 255) / 256);
  // the alignment kernels stride over the tasks like stage A over the reads: oversubscribed grids for the same reason (qmk_map_grid)
  static const int ao = getenv("QM_ALIGN_OVERSUB") && atoi(getenv("QM_ALIGN_OVERSUB")) > 0 ? atoi(getenv("QM_ALIGN_OVERSUB")) : qmk_grid_oversub();
  if (A.long_reads) {                                    // reads of 513 .. 2048 characters in the batch: two waves per block, long images
    switch (sel_ksw_ring_slots(A.bandwidth)) {
      case 32: hipLaunchKernelGGL((qm_sel_align_kernel<32, 2, QM_KSW_MAXLEN_LONG>), dim3((unsigned)(num_cu * 4 * ao)), dim3(128), 0, st, P, A); break;
      case 64: hipLaunchKernelGGL((qm_sel_align_kernel<64, 2, QM_KSW_MAXLEN_LONG>), dim3((unsigned)(num_cu * 4 * ao)), dim3(128), 0, st, P, A); break;
      case 128: hipLaunchKernelGGL((qm_sel_align_kernel<128, 2, QM_KSW_MAXLEN_LONG>), dim3((unsigned)(num_cu * 3 * ao)), dim3(128), 0, st, P, A); break;
      default:                                             // bands beyond 97: blocks in device memory (the host sized A.ksw_rows for num_cu blocks)
        if (!A.ksw_rows) return hipErrorInvalidValue;
        hipLaunchKernelGGL(qm_sel_align_gmem_kernel, dim3((unsigned)num_cu), dim3(64 * QMK_GMEM_WAVES), 0, st, P, A); break;
    }
  } else
  switch (sel_ksw_ring_slots(A.bandwidth)) {            // one kernel for every --dpBandwidth: the band decides the ring
    case 32:                                                                             // register edition (--dpBandwidth <= 15), eight per wavefront
      // images sized by the read-length class of the batch (the stage-A kernels' slot classes): 13 / 17 / 21 / 37 KB of LDS per block
      if (A.short_len > 0 && A.short_len <= 128) hipLaunchKernelGGL(qm_sel_align2_kernel<128 + 32>, dim3((unsigned)(num_cu * 8 * ao)), dim3(256), 0, st, P, A);
      else if (A.short_len > 0 && A.short_len <= 192) hipLaunchKernelGGL(qm_sel_align2_kernel<192 + 32>, dim3((unsigned)(num_cu * 8 * ao)), dim3(256), 0, st, P, A);
      else if (A.short_len > 0 && A.short_len <= 256) hipLaunchKernelGGL(qm_sel_align2_kernel<256 + 32>, dim3((unsigned)(num_cu * 7 * ao)), dim3(256), 0, st, P, A);
      else hipLaunchKernelGGL(qm_sel_align2_kernel<QM_KSW_MAXLEN>, dim3((unsigned)(num_cu * 4 * ao)), dim3(256), 0, st, P, A);
      break;
    case 64: hipLaunchKernelGGL((qm_sel_align_kernel<64, 4>), dim3((unsigned)(num_cu * 8 * ao)), dim3(256), 0, st, P, A); break;
    case 128: hipLaunchKernelGGL((qm_sel_align_kernel<128, 4>), dim3((unsigned)(num_cu * 4 * ao)), dim3(256), 0, st, P, A); break;
    default: hipLaunchKernelGGL((qm_sel_align_kernel<1024, 2>), dim3((unsigned)(num_cu * 2 * ao)), dim3(128), 0, st, P, A); break;   // 83 KB of LDS per block
  }
  hipLaunchKernelGGL(qm_sel_finish_kernel, dim3(nb), dim3(256), 0, st, P, A);
  return hipGetLastError();
}
size_t qmk_sel_task_bytes(void) { return sizeof(SelTask); }
size_t qmk_sel_gmem_rows_bytes(int num_cu) { return (size_t)num_cu * QMK_GMEM_WAVES * 4 * sizeof(KswRowT<QM_KSW_RING_GMEM, QM_KSW_MAXLEN_LONG>); }
hipError_t qmk_sel_compact(const void* pp, const void* tmp, const void* toff, hipStream_t st) {
  const PairBatch& P = *(const PairBatch*)pp;
  if (P.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qm_sel_compact_kernel, dim3((unsigned)((P.n + 255) / 256)), dim3(256), 0, st, P, (const qm_hit*)tmp, (const long long*)toff);
  return hipGetLastError();
}

hipError_t qmk_pair_count(const void* pp, hipStream_t st) {
  const PairBatch& P = *(const PairBatch*)pp;
  if (P.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qm_pair_count_kernel, dim3((unsigned)((P.n + 255) / 256)), dim3(256), 0, st, P);
  return hipGetLastError();
}

hipError_t qmk_pair_write(const void* pp, hipStream_t st) {
  const PairBatch& P = *(const PairBatch*)pp;
  if (P.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qm_pair_write_kernel, dim3((unsigned)((P.n + 255) / 256)), dim3(256), 0, st, P);
  return hipGetLastError();
}

hipError_t qmk_unpack_reads(const unsigned char* packed, const long long* off, long long n, long long total_chars, unsigned char* seq, const void* exc,
                            long long n_exc, hipStream_t st) {
  if (n > 0) {
    const long long threads = n * (long long)QM_UNPACK_TPR;
    hipLaunchKernelGGL(qm_unpack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, packed, off, n, seq);
  }
  if (n_exc > 0) hipLaunchKernelGGL(qm_unpack_exc_kernel, dim3((unsigned)((n_exc + 255) / 256)), dim3(256), 0, st, (const qm_pack_exc*)exc, n_exc, seq, (unsigned long long)total_chars);
  return hipGetLastError();
}

size_t qmk_scan_temp_bytes(long long n) {
  size_t bytes = 0;
  auto it = rocprim::make_transform_iterator((const u32*)nullptr, U32ToI64());
  (void)rocprim::exclusive_scan(nullptr, bytes, it, (long long*)nullptr, 0LL, (size_t)n, rocprim::plus<long long>());
  return bytes;
}

hipError_t qmk_scan_counts(void* temp, size_t temp_bytes, const u32* cnt, long long* offs, long long n, hipStream_t st) {
  auto it = rocprim::make_transform_iterator(cnt, U32ToI64());
  return rocprim::exclusive_scan(temp, temp_bytes, it, offs, 0LL, (size_t)n, rocprim::plus<long long>(), st);
}

hipError_t qmk_scan_counts_masked(void* temp, size_t temp_bytes, const u32* cnt, long long* offs, long long n, hipStream_t st) {
  auto it = rocprim::make_transform_iterator(cnt, U32MaskToI64());
  return rocprim::exclusive_scan(temp, temp_bytes, it, offs, 0LL, (size_t)n, rocprim::plus<long long>(), st);
}

hipError_t qmk_stage_gather(long long nreads, const u32* ivcnt, const long long* ivoff, const void* iv, const long long* ivcsr, void* iv_out,
                            const u32* lcnt, const long long* loff, const unsigned long long* lists, const long long* lcsr,
                            unsigned long long* words_out, hipStream_t st) {
  if (nreads <= 0) return hipSuccess;
  hipLaunchKernelGGL(qm_stage_gather_kernel, dim3((unsigned)((nreads * 4 + 255) / 256)), dim3(256), 0, st, nreads, ivcnt, ivoff,
                     (const qm_sa_interval_hit*)iv, ivcsr, (qm_sa_interval_hit*)iv_out, lcnt, loff, (const u64*)lists, lcsr, (u64*)words_out);
  return hipGetLastError();
}

}  // extern "C"
