// qm_wave.h -- the 64-lane wavefront abstraction the mapper is written against.
//
// The mapper (qm_mapper.inl) is explicit about what is wave-uniform (plain
// scalars -> SGPRs / scalar branches on gfx950) and what varies per lane
// (LV<T>).  On the device an LV<T> is ONE register per lane and QM_LANES runs
// its body once with the hardware lane id; ballots, lane reads and LDS atomics
// map to the CDNA4 instructions.  When compiled with -DQM_EMU (tests/emu only,
// never shipped in the product library) an LV<T> is a 64-entry array and
// QM_LANES is a loop, so the *same source* runs lane-by-lane on a CPU and can
// be checked against the oracle without a GPU.
#pragma once
#include <stdint.h>

// QM_LDS(T): T in this wave's LDS slab, for pointers that must never be mistaken for generic ones (a generic pointer is
// dereferenced with FLAT instructions -- the vector-memory path, even when it lands in LDS)
#ifdef QM_EMU
#define QM_LDS(T) T
#define QM_DEV inline
#define QM_NL 64
#define QM_LANES(l) for (int l = 0; l < 64; ++l)
#define QM_SCALAR(x) ((void)0)
#else
#include <hip/hip_runtime.h>
#define QM_LDS(T) T __attribute__((address_space(3)))
#define QM_DEV __device__ __forceinline__
#define QM_NL 1
#define QM_LANES(l) for (int _qm_once = 0, l = (int)(threadIdx.x & 63); _qm_once < 1; ++_qm_once)
// pins a wave-uniform int in an SGPR at this point (an opaque copy): keeps the compiler from fusing scalar arithmetic into a
// three-operand VALU instruction (v_min3 / v_max3 have no scalar form; it then pays moves and a v_readfirstlane around one)
#define QM_SCALAR(x) asm volatile("" : "+s"(x))
#endif

namespace qm {

typedef unsigned long long u64;
typedef unsigned int u32;

template <typename T>
struct LV {
  T v[QM_NL];
  QM_DEV T& operator[](int l) { return v[QM_NL == 1 ? 0 : l]; }
  QM_DEV const T& operator[](int l) const { return v[QM_NL == 1 ? 0 : l]; }
};

#ifdef QM_EMU
QM_DEV u64 ballot(const LV<bool>& b) {
  u64 m = 0;
  for (int l = 0; l < 64; ++l) m |= (u64)(b.v[l] ? 1 : 0) << l;
  return m;
}
template <typename T> QM_DEV T read_lane(const LV<T>& x, int lane) { return x.v[lane]; }
QM_DEV int ctz64(u64 x) { return x ? __builtin_ctzll(x) : 64; }
QM_DEV int popc64(u64 x) { return __builtin_popcountll(x); }
QM_DEV u64 brev64(u64 x) {
  x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  return __builtin_bswap64(x);
}
QM_DEV u32 brev32(u32 x) { return (u32)(brev64((u64)x) >> 32); }
QM_DEV void wave_fence() {}
QM_DEV void atomic_min_u64(u64* p, u64 v) { if (v < *p) *p = v; }
QM_DEV void atomic_max_u64(u64* p, u64 v) { if (v > *p) *p = v; }
QM_DEV void atomic_or_u64(u64* p, u64 v) { *p |= v; }
QM_DEV u64 atomic_add_u64(u64* p, u64 v) { u64 o = *p; *p = o + v; return o; }
template <typename T> QM_DEV T uniform(T x) { return x; }
#else
QM_DEV u64 ballot(const LV<bool>& b) { return __builtin_amdgcn_ballot_w64(b.v[0]); }
// the lane index is wave-uniform by construction (one `lane` for the whole wavefront): v_readlane returns the value in
// an SGPR, so everything computed from it stays on the scalar unit (a __shfl result would drag it onto the VALU)
QM_DEV int read_lane(const LV<int>& x, int lane) { return __builtin_amdgcn_readlane(x.v[0], __builtin_amdgcn_readfirstlane(lane)); }
QM_DEV u32 read_lane(const LV<u32>& x, int lane) { return (u32)__builtin_amdgcn_readlane((int)x.v[0], __builtin_amdgcn_readfirstlane(lane)); }
QM_DEV u64 read_lane(const LV<u64>& x, int lane) {
  const int ln = __builtin_amdgcn_readfirstlane(lane);
  int lo = __builtin_amdgcn_readlane((int)(u32)x.v[0], ln), hi = __builtin_amdgcn_readlane((int)(u32)(x.v[0] >> 32), ln);
  return ((u64)(u32)hi << 32) | (u32)lo;
}
QM_DEV int ctz64(u64 x) { return x ? __builtin_ctzll(x) : 64; }
QM_DEV int popc64(u64 x) { return __builtin_popcountll(x); }
QM_DEV u64 brev64(u64 x) { return __brevll(x); }
QM_DEV u32 brev32(u32 x) { return __builtin_bitreverse32(x); }
// orders this wave's LDS / global accesses across lanes (same-wave RAW through memory)
QM_DEV void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
QM_DEV void atomic_min_u64(u64* p, u64 v) { atomicMin(p, v); }
QM_DEV void atomic_max_u64(u64* p, u64 v) { atomicMax(p, v); }
QM_DEV void atomic_or_u64(u64* p, u64 v) { atomicOr(p, v); }
QM_DEV u64 atomic_add_u64(u64* p, u64 v) { return atomicAdd(p, v); }
// tell the compiler a value is wave-uniform (keeps control flow on the scalar unit)
QM_DEV int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
QM_DEV u32 uniform(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
QM_DEV u64 uniform(u64 x) {
  u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)x);
  u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(x >> 32));
  return ((u64)hi << 32) | lo;
}
QM_DEV long long uniform(long long x) { return (long long)uniform((u64)x); }
QM_DEV bool uniform(bool x) { return __builtin_amdgcn_readfirstlane((int)x) != 0; }
#endif

#ifdef QM_EMU
QM_DEV u64 mulhi64(u64 a, u64 b) { return (u64)(((unsigned __int128)a * (unsigned __int128)b) >> 64); }
#else
QM_DEV u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }
#endif

#ifdef QM_EMU
QM_DEV int wave_max(const LV<int>& x) { int m = x.v[0]; for (int l = 1; l < 64; ++l) m = x.v[l] > m ? x.v[l] : m; return m; }
// inclusive scans over the lanes: x[l] = x[0] + ... + x[l]  /  max(x[0], ..., x[l]) (values >= 0)
QM_DEV void lane_scan_add(LV<int>& x) { for (int l = 1; l < 64; ++l) x.v[l] += x.v[l - 1]; }
QM_DEV void lane_scan_max(LV<int>& x) { for (int l = 1; l < 64; ++l) x.v[l] = x.v[l - 1] > x.v[l] ? x.v[l - 1] : x.v[l]; }
// out[l] = in[(l - 1) & 63]: every lane reads its lower neighbour (wrapping)
QM_DEV void lane_rotate_up(const LV<int>& in, LV<int>& out) { for (int l = 0; l < 64; ++l) out.v[l] = in.v[(l + 63) & 63]; }
// the same within every row of 16 lanes: lane c of a row reads lane (c - 1) & 15 of that row
QM_DEV void row_rotate_up(const LV<int>& in, LV<int>& out) { for (int l = 0; l < 64; ++l) out.v[l] = in.v[(l & ~15) | ((l + 15) & 15)]; }
// every lane of a row of 16 gets the value of the row's last lane
QM_DEV void row_last(const LV<int>& in, LV<int>& out) { for (int l = 0; l < 64; ++l) out.v[l] = in.v[l | 15]; }
// rows of 16 lanes (the -s strip alignments, one diagonal per lane): lane c reads lane c + 1 of its row (the last one: fill);
// x[c] = max over the lanes below c of its row (lane 0: fill); every lane gets its row's maximum
QM_DEV void row16_shl1(const LV<int>& in, LV<int>& out, int fill) { for (int l = 0; l < 64; ++l) out.v[l] = (l & 15) == 15 ? fill : in.v[l + 1]; }
QM_DEV void row16_scan_max_excl(LV<int>& x, int fill) {
  for (int b = 0; b < 64; b += 16) { int run = fill; for (int l = b; l < b + 16; ++l) { const int t = x.v[l]; x.v[l] = run; run = t > run ? t : run; } }
}
QM_DEV void row16_max_all(LV<int>& x) { for (int b = 0; b < 64; b += 16) { int m = x.v[b]; for (int l = b + 1; l < b + 16; ++l) m = x.v[l] > m ? x.v[l] : m; for (int l = b; l < b + 16; ++l) x.v[l] = m; } }
// every lane gets the minimum over its aligned group of G lanes (G a power of two, wave-uniform)
QM_DEV void group_min(LV<int>& x, int G) {
  for (int b = 0; b < 64; b += G) {
    int m = x.v[b];
    for (int l = b + 1; l < b + G; ++l) m = x.v[l] < m ? x.v[l] : m;
    for (int l = b; l < b + G; ++l) x.v[l] = m;
  }
}
QM_DEV int clz64(u64 x) { return x ? __builtin_clzll(x) : 64; }
QM_DEV u64 load_u64_unaligned(const unsigned char* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
QM_DEV u32 load_u32_unaligned(const unsigned char* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
// v_perm_b32 with selectors 0..7: byte j of the result = byte sel_j of the eight bytes {hi, lo} (0..3 from lo, 4..7 from hi)
QM_DEV u32 perm8(u32 hi, u32 lo, u32 sel) {                 // (selector 0x0c: the constant byte 0x00)
  const u64 tab = ((u64)hi << 32) | lo; u32 r = 0;
  for (int j = 0; j < 4; ++j) { const u32 sj = (sel >> (8 * j)) & 0xff; if (sj != 0x0c) r |= (u32)((tab >> (8 * (sj & 7))) & 0xff) << (8 * j); }
  return r;
}
// packed 16-bit arithmetic on the two halves of a dword (v_pk_add_u16, v_pk_sub_i16, v_pk_max_i16, v_pk_max_u16, v_pk_min_u16)
QM_DEV u32 pk_add(u32 a, u32 b) { return ((a + b) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16); }
QM_DEV u32 pk_sub(u32 a, u32 b) { return ((a - b) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16); }
QM_DEV u32 pk_max_i(u32 a, u32 b) {
  const short al = (short)a, bl = (short)b, ah = (short)(a >> 16), bh = (short)(b >> 16);
  return (u32)(unsigned short)(al > bl ? al : bl) | ((u32)(unsigned short)(ah > bh ? ah : bh) << 16);
}
QM_DEV u32 pk_max_u(u32 a, u32 b) { const u32 al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16; return (al > bl ? al : bl) | ((ah > bh ? ah : bh) << 16); }
QM_DEV u32 pk_min_u(u32 a, u32 b) { const u32 al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16; return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16); }
// out[l] = in[l ^ 1]: every lane reads its neighbour within a pair
QM_DEV void lane_xor1(const LV<u32>& in, LV<u32>& out) { for (int l = 0; l < 64; ++l) out.v[l] = in.v[l ^ 1]; }
// out[l] = in[l ^ 32]: the two halves of the wavefront trade places
template <typename T> QM_DEV void swap32(const LV<T>& in, LV<T>& out) { for (int l = 0; l < 64; ++l) out.v[l] = in.v[l ^ 32]; }
// v_alignbyte_b32: the eight bytes {hi, lo} shifted right by `shift` (0..3) bytes, low dword
QM_DEV u32 align_bytes(u32 hi, u32 lo, int shift) { return (u32)((((u64)hi << 32) | lo) >> (8 * shift)); }
struct U4 { u32 x, y, z, w; };
QM_DEV U4 load_16(const void* p) { U4 v; __builtin_memcpy(&v, p, 16); return v; }
QM_DEV long long load_uniform_i64(const long long* p) { return *p; }
// lane `lane` copies one dword from global memory straight into ldsBase[lane] (no register in between); lds_dma_wait()
// before the wave reads what it asked for
QM_DEV void lds_dma_u32(const u32* g, u32* ldsBase, int lane) { ldsBase[lane] = *g; }
QM_DEV void lds_dma_wait() {}
// the same for 16 bytes per lane: ldsBase + 16 * lane
QM_DEV void lds_dma_u128(const void* g, void* ldsBase, int lane) { __builtin_memcpy((unsigned char*)ldsBase + 16 * lane, g, 16); }
QM_DEV void load_32(const void* p, U4& a, U4& b) { a = load_16(p); b = load_16((const unsigned char*)p + 16); }
QM_DEV void load_16x2(const void* p, const void* q, U4& a, U4& b) { a = load_16(p); b = load_16(q); }
QM_DEV void load_8_16(const u64* p8, const u64* p16, u64& a, U4& b) { a = *p8; b = load_16(p16); }
template <int N> QM_DEV void load_8_16xN(const u64* const* B, const int* bit, u64* word, U4* meta) {
  for (int t = 0; t < N; ++t) load_8_16(B[t] + (bit[t] >> 6), B[t] + 6, word[t], meta[t]);
}
// ---- half-wavefront helpers (qm_duo.inl: lanes 0-31 and lanes 32-63 each work on a read of their own)
// every lane gets the 32 ballot bits of its own half
QM_DEV void half_ballot(const LV<bool>& b, LV<u32>& out) { const u64 m = ballot(b); for (int l = 0; l < 64; ++l) out.v[l] = (u32)(m >> (l & 32)); }
// out[l] = x[lane idx[l] of l's half]
QM_DEV void half_read(const LV<u32>& x, const LV<int>& idx, LV<u32>& out) { LV<u32> t; for (int l = 0; l < 64; ++l) t.v[l] = x.v[(l & 32) + (idx.v[l] & 31)]; out = t; }
QM_DEV void wave_read(const LV<u32>& x, const LV<int>& idx, LV<u32>& out) { LV<u32> t; for (int l = 0; l < 64; ++l) t.v[l] = x.v[idx.v[l] & 63]; out = t; }
// every lane gets the maximum over its half
QM_DEV void half_max(LV<int>& x) {
  for (int b = 0; b < 64; b += 32) { int m = x.v[b]; for (int l = b + 1; l < b + 32; ++l) m = x.v[l] > m ? x.v[l] : m; for (int l = b; l < b + 32; ++l) x.v[l] = m; }
}
QM_DEV void load_48(const void* p, U4& a, U4& b, U4& c) { a = load_16(p); b = load_16((const unsigned char*)p + 16); c = load_16((const unsigned char*)p + 32); }
#else
// DPP reductions (profiles/microbench/dpp_check.hip: row_shr:n gives lane i the value of lane i - n of its row of 16;
// a lane without a source keeps `old`).  A step is one VALU instruction and no trip through the LDS crossbar --
// __shfl_xor is a ds_bpermute with an address computation and a wait each.
template <int CTRL, int ROWMASK> QM_DEV int dpp_get(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false); }
QM_DEV int wave_max(const LV<int>& x) {
  int v = x.v[0], t;
  t = dpp_get<0x111, 0xf>(v); v = t > v ? t : v;           // row_shr:1, 2, 4, 8: lane 15 of every row holds the row's maximum
  t = dpp_get<0x112, 0xf>(v); v = t > v ? t : v;
  t = dpp_get<0x114, 0xf>(v); v = t > v ? t : v;
  t = dpp_get<0x118, 0xf>(v); v = t > v ? t : v;
  t = dpp_get<0x142, 0xa>(v); v = t > v ? t : v;           // row_bcast:15 into rows 1 and 3
  t = dpp_get<0x143, 0xc>(v); v = t > v ? t : v;           // row_bcast:31 into rows 2 and 3: lane 63 has it all
  return __builtin_amdgcn_readlane(v, 63);
}
// inclusive scans over the 64 lanes (all lanes active): four row_shr steps inside every row of 16 -- a lane without a source
// takes the identity 0 --, then the last lane of row 0 / 2 into row 1 / 3 and lane 31 into rows 2 and 3
QM_DEV void lane_scan_add(LV<int>& x) {
  int v = x.v[0];
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  x.v[0] = v;
}
QM_DEV void lane_scan_max(LV<int>& x) {     // values >= 0
  int v = x.v[0], t;
  t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
  x.v[0] = v;
}
// DPP wave_ror:1 -- one VALU instruction, no trip through the LDS crossbar
QM_DEV void lane_rotate_up(const LV<int>& in, LV<int>& out) { out.v[0] = __builtin_amdgcn_update_dpp(0, in.v[0], 0x13C, 0xf, 0xf, false); }
// DPP row_ror:1
QM_DEV void row_rotate_up(const LV<int>& in, LV<int>& out) { out.v[0] = __builtin_amdgcn_update_dpp(0, in.v[0], 0x121, 0xf, 0xf, true); }   // (every lane has a source: no old value to keep)
// (rare callers only: a trip through the LDS crossbar)
QM_DEV void row_last(const LV<int>& in, LV<int>& out) { out.v[0] = __shfl(in.v[0], (int)((threadIdx.x & 63) | 15), 64); }
// rows of 16 lanes (the -s strip alignments): row_shl:1 -- lane c reads lane c + 1, the last lane keeps `fill`; an exclusive max scan by
// row_shr steps (a lane without a source keeps the fill); the row's maximum in every lane by rotations
QM_DEV void row16_shl1(const LV<int>& in, LV<int>& out, int fill) { out.v[0] = __builtin_amdgcn_update_dpp(fill, in.v[0], 0x101, 0xf, 0xf, false); }
QM_DEV void row16_scan_max_excl(LV<int>& x, int fill) {
  int v = __builtin_amdgcn_update_dpp(fill, x.v[0], 0x111, 0xf, 0xf, false), t;
  t = __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(fill, v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(fill, v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(fill, v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;
  x.v[0] = v;
}
QM_DEV void row16_max_all(LV<int>& x) {
  int v = x.v[0], t;
  t = __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false); v = t > v ? t : v;
  x.v[0] = v;
}
// butterfly inside a row of 16 with DPP pairings: lanes ^1, lanes ^2 (quad permutes), then the mirror image within 8 and
// within 16 -- once the quads are uniform any pairing of the two halves does; across rows the crossbar
QM_DEV void group_min(LV<int>& x, int G) {
  int v = x.v[0], t;
  if (G > 1) { t = dpp_get<0xB1, 0xf>(v); v = t < v ? t : v; }         // quad_perm:[1,0,3,2]
  if (G > 2) { t = dpp_get<0x4E, 0xf>(v); v = t < v ? t : v; }         // quad_perm:[2,3,0,1]
  if (G > 4) { t = dpp_get<0x141, 0xf>(v); v = t < v ? t : v; }        // row_half_mirror
  if (G > 8) { t = dpp_get<0x140, 0xf>(v); v = t < v ? t : v; }        // row_mirror
  if (G > 16) { t = __shfl_xor(v, 16, 64); v = t < v ? t : v; }
  if (G > 32) { t = __shfl_xor(v, 32, 64); v = t < v ? t : v; }
  x.v[0] = v;
}
QM_DEV int clz64(u64 x) { return x ? __builtin_clzll(x) : 64; }
QM_DEV u64 load_u64_unaligned(const unsigned char* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
QM_DEV u32 load_u32_unaligned(const unsigned char* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
QM_DEV u32 perm8(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }     // v_perm_b32
// packed 16-bit arithmetic on the two halves of a dword: one VALU instruction for two values
typedef short qm_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short qm_u16x2 __attribute__((ext_vector_type(2)));
QM_DEV u32 pk_add(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(qm_u16x2, a) + __builtin_bit_cast(qm_u16x2, b)); }
QM_DEV u32 pk_sub(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(qm_u16x2, a) - __builtin_bit_cast(qm_u16x2, b)); }
QM_DEV u32 pk_max_i(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(qm_s16x2, a), __builtin_bit_cast(qm_s16x2, b))); }
QM_DEV u32 pk_max_u(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(qm_u16x2, a), __builtin_bit_cast(qm_u16x2, b))); }
QM_DEV u32 pk_min_u(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(qm_u16x2, a), __builtin_bit_cast(qm_u16x2, b))); }
QM_DEV void lane_xor1(const LV<u32>& in, LV<u32>& out) {                                     // DPP quad_perm:[1,0,3,2]
  out.v[0] = (u32)__builtin_amdgcn_update_dpp(0, (int)in.v[0], 0xB1, 0xf, 0xf, false);
}
QM_DEV void swap32(const LV<u32>& in, LV<u32>& out) { out.v[0] = (u32)__shfl_xor((int)in.v[0], 32, 64); }
QM_DEV void swap32(const LV<u64>& in, LV<u64>& out) {
  const u32 lo = (u32)__shfl_xor((int)(u32)in.v[0], 32, 64), hi = (u32)__shfl_xor((int)(u32)(in.v[0] >> 32), 32, 64);
  out.v[0] = ((u64)hi << 32) | lo;
}
QM_DEV u32 align_bytes(u32 hi, u32 lo, int shift) { return __builtin_amdgcn_alignbyte(hi, lo, (u32)shift); }
// wave-uniform 8-byte load on the scalar unit (s_load): read-only data, uniform address
QM_DEV long long load_uniform_i64(const long long* p) {
  typedef const long long __attribute__((address_space(4)))* cptr;
  return *(cptr)(unsigned long long)p;
}
// global_load_lds_dword: every active lane fetches one dword from its own global address, the hardware writes it to
// LDS at ldsBase + 4 * lane id -- an asynchronous gather that costs no VGPR for the data.  `lane` must be the lane id.
QM_DEV void lds_dma_u32(const u32* g, u32* ldsBase, int) {
  typedef const u32 __attribute__((address_space(1)))* gptr; typedef u32 __attribute__((address_space(3)))* lptr;
  __builtin_amdgcn_global_load_lds((gptr)(unsigned long long)g, (lptr)(unsigned)(unsigned long long)ldsBase, 4, 0, 0);
}
// global_load_lds_dwordx4 (gfx950): 16 bytes per lane to ldsBase + 16 * lane id
QM_DEV void lds_dma_u128(const void* g, void* ldsBase, int) {
  typedef const u32 __attribute__((address_space(1)))* gptr; typedef u32 __attribute__((address_space(3)))* lptr;
  __builtin_amdgcn_global_load_lds((gptr)(unsigned long long)g, (lptr)(unsigned)(unsigned long long)ldsBase, 16, 0, 0);
}
QM_DEV void lds_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }   // vmcnt(0)
// one 16-byte load that the compiler cannot split into a key load plus a dependent value load
struct U4 { u32 x, y, z, w; };
QM_DEV U4 load_16(const void* p) {
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  v4u q = *(const v4u*)p;
  U4 v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w; return v;
}
// two 16-byte loads issued back to back and waited for together (the empty asm keeps the compiler from sinking
// the second load behind a branch on the first one's result)
QM_DEV void load_32(const void* p, U4& a, U4& b) {
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  v4u x = ((const v4u*)p)[0], y = ((const v4u*)p)[1];
  asm volatile("" : "+v"(x), "+v"(y));
  a.x = x.x; a.y = x.y; a.z = x.z; a.w = x.w; b.x = y.x; b.y = y.y; b.z = y.z; b.w = y.w;
}
// two 16-byte loads from two addresses issued back to back and waited for together
QM_DEV void load_16x2(const void* p, const void* q, U4& a, U4& b) {
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  v4u x = *(const v4u*)p, y = *(const v4u*)q;
  asm volatile("" : "+v"(x), "+v"(y));
  a.x = x.x; a.y = x.y; a.z = x.z; a.w = x.w; b.x = y.x; b.y = y.y; b.z = y.z; b.w = y.w;
}
// an 8-byte and a 16-byte load issued together (see load_32)
QM_DEV void load_8_16(const u64* p8, const u64* p16, u64& a, U4& b) {
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  u64 x = *p8; v4u y = *(const v4u*)p16;
  asm volatile("" : "+v"(x), "+v"(y));
  a = x; b.x = y.x; b.y = y.y; b.z = y.z; b.w = y.w;
}
// N (word, rank words) pairs of re-blocked BooPHF levels, all loads in flight before any is consumed
template <int N> QM_DEV void load_8_16xN(const u64* const* B, const int* bit, u64* word, U4* meta) {
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  u64 x[N]; v4u y[N];
#pragma unroll
  for (int t = 0; t < N; ++t) { x[t] = B[t][bit[t] >> 6]; y[t] = *(const v4u*)(B[t] + 6); }
#pragma unroll
  for (int t = 0; t < N; ++t) asm volatile("" : "+v"(x[t]), "+v"(y[t]));
#pragma unroll
  for (int t = 0; t < N; ++t) { word[t] = x[t]; meta[t].x = y[t].x; meta[t].y = y[t].y; meta[t].z = y[t].z; meta[t].w = y[t].w; }
}
// ---- half-wavefront helpers (qm_duo.inl: lanes 0-31 and lanes 32-63 each work on a read of their own)
QM_DEV void half_ballot(const LV<bool>& b, LV<u32>& out) {
  const u64 m = __builtin_amdgcn_ballot_w64(b.v[0]);
  out.v[0] = (threadIdx.x & 32) ? (u32)(m >> 32) : (u32)m;
}
// a trip through the LDS crossbar (ds_bpermute_b32): the index differs between the halves, so it cannot be a v_readlane
// ... the same over the whole wavefront: lane l reads lane idx[l]'s value
QM_DEV void wave_read(const LV<u32>& x, const LV<int>& idx, LV<u32>& out) {
  out.v[0] = (u32)__builtin_amdgcn_ds_bpermute((int)((u32)(idx.v[0] & 63) << 2), (int)x.v[0]);
}
QM_DEV void half_read(const LV<u32>& x, const LV<int>& idx, LV<u32>& out) {
  out.v[0] = (u32)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 32) + (u32)(idx.v[0] & 31)) << 2), (int)x.v[0]);
}
QM_DEV void half_max(LV<int>& x) {
  int v = x.v[0], t;
  t = dpp_get<0x111, 0xf>(v); v = t > v ? t : v;           // row_shr:1, 2, 4, 8: lane 15 of every row holds the row's maximum
  t = dpp_get<0x112, 0xf>(v); v = t > v ? t : v;
  t = dpp_get<0x114, 0xf>(v); v = t > v ? t : v;
  t = dpp_get<0x118, 0xf>(v); v = t > v ? t : v;
  t = dpp_get<0x142, 0xa>(v); v = t > v ? t : v;           // row_bcast:15 into rows 1 and 3: lanes 31 and 63 hold their half's maximum
  x.v[0] = __builtin_amdgcn_ds_bpermute((int)((threadIdx.x | 31u) << 2), v);
}
// three 16-byte loads from one 64-byte bucket, issued back to back and waited for together
QM_DEV void load_48(const void* p, U4& a, U4& b, U4& c) {
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  v4u x = ((const v4u*)p)[0], y = ((const v4u*)p)[1], z = ((const v4u*)p)[2];
  asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
  a.x = x.x; a.y = x.y; a.z = x.z; a.w = x.w; b.x = y.x; b.y = y.y; b.z = y.z; b.w = y.w; c.x = z.x; c.y = z.y; c.z = z.z; c.w = z.w;
}
#endif

QM_DEV u64 lanemask_lt(int l) { return l ? (~0ULL >> (64 - l)) : 0ULL; }

// An N x 64-bit unsigned integer with just the operators the collector's per-position flag word needs (shifts, and / or / not,
// comparison with small constants): the flag word of the long-read kernels, whose six flags x NS slots no longer fit a register
// pair (qm_mapper.inl, Strand).  Slow and simple on purpose: those kernels run on the few reads of a batch that need them.
template <int N>
struct Wide {
  u64 w[N];
  QM_DEV Wide() { for (int i = 0; i < N; ++i) w[i] = 0; }
  QM_DEV Wide(unsigned long long x) { w[0] = x; for (int i = 1; i < N; ++i) w[i] = 0; }
  QM_DEV Wide operator<<(int n) const {
    Wide r; const int ws = n >> 6, bs = n & 63;
    for (int i = 0; i < N; ++i) {
      u64 v = 0;
      for (int j = 0; j < N; ++j) {
        if (j == i - ws) v |= w[j] << bs;
        if (j == i - ws - 1 && bs) v |= w[j] >> (64 - bs);
      }
      r.w[i] = v;
    }
    return r;
  }
  QM_DEV Wide operator>>(int n) const {
    Wide r; const int ws = n >> 6, bs = n & 63;
    for (int i = 0; i < N; ++i) {
      u64 v = 0;
      for (int j = 0; j < N; ++j) {
        if (j == i + ws) v |= w[j] >> bs;
        if (j == i + ws + 1 && bs) v |= w[j] << (64 - bs);
      }
      r.w[i] = v;
    }
    return r;
  }
  QM_DEV Wide operator&(const Wide& o) const { Wide r; for (int i = 0; i < N; ++i) r.w[i] = w[i] & o.w[i]; return r; }
  QM_DEV Wide operator|(const Wide& o) const { Wide r; for (int i = 0; i < N; ++i) r.w[i] = w[i] | o.w[i]; return r; }
  QM_DEV Wide operator~() const { Wide r; for (int i = 0; i < N; ++i) r.w[i] = ~w[i]; return r; }
  QM_DEV Wide& operator|=(const Wide& o) { for (int i = 0; i < N; ++i) w[i] |= o.w[i]; return *this; }
  QM_DEV bool operator==(const Wide& o) const { bool e = true; for (int i = 0; i < N; ++i) e = e && w[i] == o.w[i]; return e; }
  QM_DEV bool operator!=(const Wide& o) const { return !(*this == o); }
};
template <int N> QM_DEV Wide<N> read_lane(const LV<Wide<N>>& x, int lane) {
  Wide<N> r;
#ifdef QM_EMU
  r = x.v[lane];
#else
  const int ln = __builtin_amdgcn_readfirstlane(lane);
  for (int i = 0; i < N; ++i) {
    const int lo = __builtin_amdgcn_readlane((int)(u32)x.v[0].w[i], ln), hi = __builtin_amdgcn_readlane((int)(u32)(x.v[0].w[i] >> 32), ln);
    r.w[i] = ((u64)(u32)hi << 32) | (u32)lo;
  }
#endif
  return r;
}
#ifndef QM_EMU
template <int N> QM_DEV void swap32(const LV<Wide<N>>& in, LV<Wide<N>>& out) {
  for (int i = 0; i < N; ++i) {
    const u32 lo = (u32)__shfl_xor((int)(u32)in.v[0].w[i], 32, 64), hi = (u32)__shfl_xor((int)(u32)(in.v[0].w[i] >> 32), 32, 64);
    out.v[0].w[i] = ((u64)hi << 32) | lo;
  }
}
#endif

}  // namespace qm
