// tests/emu/qm_emu.cpp -- TEST-ONLY lane emulation of the device mapper.
// Compiles rapmap_amd/csrc/qm_mapper.inl with -DQM_EMU so that every LV<T> is a
// 64-entry array and every QM_LANES block a loop: the kernel's source runs on the
// CPU one wavefront at a time.  Used by tests/ (not gpu-marked) to check the wave
// algorithm against the oracle without a GPU.  Never part of libqmap_mi355.so.
#define QM_EMU 1
#include "../../rapmap_amd/csrc/qm_mapper.inl"
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace qm;

extern "C" {

// slots: cap x {u64 key, i32 lb, i32 ub}; sainfo: nSA x {u32 tid, i32 pos}; text padded by >= 64 bytes
int qe_map(int k, const unsigned char* text, long long n, const int* SA, long long nSA, const void* sainfo,
           const void* slots, unsigned long long hmask, const qm_opts* o, long long nunits,
           const unsigned char* seq1, const long long* off1, const unsigned char* seq2, const long long* off2,
           int ns, long long* hit_offsets, qm_hit** hits_out, unsigned long long* counters, long long* int_offsets,
           qm_sa_interval_hit** ints_out, int* status_out) {
  DevIndex ix; ix.text = text; ix.n = n; ix.SA = SA; ix.nSA = nSA; ix.sainfo = (const SaInfo*)sainfo;
  ix.slots = (const Slot*)slots; ix.hmask = hmask; ix.k = k;
  Batch B; memset(&B, 0, sizeof(B));
  B.seq1 = seq1; B.off1 = off1; B.seq2 = seq2; B.off2 = off2; B.n = nunits;
  std::vector<u32> hc(nunits, 0); std::vector<long long> toff(nunits, 0);
  long long cap = nunits * 16 + 1024;
  std::vector<qm_hit> tmp;
  std::vector<u64> gs(4 * QM_GCAP);
  std::vector<qm_sa_interval_hit> dints((size_t)nunits * QM_DBG_CAP); std::vector<u32> dcnt(nunits, 0);
  int status = 0; u64 cursor = 0; u64 ctr[6] = {0, 0, 0, 0, 0, 0};
  B.hit_count = hc.data(); B.tmp_off = toff.data(); B.cursor = &cursor; B.counters = ctr;
  B.gscratch = gs.data(); B.status = &status; B.dbg_ints = dints.data(); B.dbg_count = dcnt.data();
  B.strict_check = o->strict_check; B.max_num_hits = o->max_num_hits; B.no_orphans = o->no_orphans;
  B.no_dovetail = o->no_dovetail; B.max_interval = o->max_interval; B.quasi_cov = o->quasi_cov;
  WaveCounters wc = {0, 0, 0, 0, 0, 0};
  while (true) {
    tmp.assign((size_t)cap, qm_hit());
    B.tmp_hits = tmp.data(); B.tmp_cap = cap; cursor = 0; status = 0;
    wc = WaveCounters{0, 0, 0, 0, 0, 0};
    for (long long u = 0; u < nunits; ++u) {
      if (ns == 2) { static WaveMem<2> M; map_unit<2>(ix, B, u, M, gs.data(), wc); }
      else { static WaveMem<4> M; map_unit<4>(ix, B, u, M, gs.data(), wc); }
    }
    if (!(status & 1)) break;
    cap *= 4;
  }
  hit_offsets[0] = 0;
  for (long long u = 0; u < nunits; ++u) hit_offsets[u + 1] = hit_offsets[u] + hc[u];
  qm_hit* out = (qm_hit*)malloc(sizeof(qm_hit) * (size_t)(hit_offsets[nunits] + 1));
  for (long long u = 0; u < nunits; ++u)
    for (u32 j = 0; j < hc[u]; ++j) out[hit_offsets[u] + j] = tmp[toff[u] + j];
  *hits_out = out;
  counters[0] = wc.pe; counters[1] = wc.se; counters[2] = wc.tot; counters[3] = wc.reads;
  counters[4] = wc.tooMany; counters[5] = wc.mapped;
  int_offsets[0] = 0;
  for (long long u = 0; u < nunits; ++u) int_offsets[u + 1] = int_offsets[u] + (dcnt[u] < QM_DBG_CAP ? dcnt[u] : QM_DBG_CAP);
  qm_sa_interval_hit* io = (qm_sa_interval_hit*)malloc(sizeof(qm_sa_interval_hit) * (size_t)(int_offsets[nunits] + 1));
  for (long long u = 0; u < nunits; ++u)
    for (long long j = 0; j < int_offsets[u + 1] - int_offsets[u]; ++j) io[int_offsets[u] + j] = dints[u * QM_DBG_CAP + j];
  *ints_out = io;
  *status_out = status;
  return 0;
}

void qe_free(void* p) { free(p); }

// host-side flattening for the emulation only (the product does this on the GPU,
// rapmap_amd/csrc/qm_kernels.hip: build_sainfo_kernel / build_slots_kernel)
unsigned long long qe_slots_cap(long long nkeys) { unsigned long long c = 16; while (c < (unsigned long long)nkeys * 2) c <<= 1; return c; }
void qe_flatten(const int* SA, long long nSA, const int* offsets, long long T, void* sainfo_out,
                const unsigned long long* keys, const int* lb, const int* ub, long long K, void* slots_out,
                unsigned long long cap) {
  SaInfo* si = (SaInfo*)sainfo_out;
  for (long long i = 0; i < nSA; ++i) {
    int p = SA[i];
    long long lo = 0, hi = T;   // upper_bound(offsets, p) - 1
    while (lo < hi) { long long mid = (lo + hi) >> 1; if (offsets[mid] <= p) lo = mid + 1; else hi = mid; }
    long long tid = lo - 1;
    si[i].tid = (u32)tid; si[i].pos = p - offsets[tid];
  }
  Slot* sl = (Slot*)slots_out;
  for (unsigned long long i = 0; i < cap; ++i) { sl[i].key = ~0ULL; sl[i].lb = 0; sl[i].ub = 0; }
  for (long long i = 0; i < K; ++i) {
    u64 j = hash_mix(keys[i]) & (cap - 1);
    while (sl[j].key != ~0ULL) j = (j + 1) & (cap - 1);
    sl[j].key = keys[i]; sl[j].lb = lb[i]; sl[j].ub = ub[i];
  }
}
}
