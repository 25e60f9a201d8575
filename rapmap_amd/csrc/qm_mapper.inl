// qm_mapper.inl -- the quasi-mapping hot path.
//
//   stage A  (map_read)   ONE wavefront owns ONE read: SACollector::operator() + hitsToMappingsSimple.
//   stage B  (pair_merge) one thread per read pair: mergeLeftRightHits + the per-pair driver.
// This file is the GENERAL stage A: every read length class, index flavour and option.  Since round 5 the reads that make up
// nearly all of a batch (<= 128 clean characters, dense or -p table, sensitive mode) are mapped by qm_lean.inl -- two reads per
// wavefront, window masks on the scalar unit -- and this kernel takes what that one leaves, the longer read classes and the other
// option sets.  It is bound by instruction issue on both pipes at 8 waves per SIMD, not by latency.
//
// Written against qm_wave.h: wave-uniform state lives in plain scalars (SGPRs), per-lane state in
// LV<T>, long-lived tables in the wave's LDS slab.  Compiled for gfx950 by qm_kernels.hip and, for
// tests only, lane-emulated on the CPU by tests/emu/qm_emu.cpp.
//
// Stages of map_read (reference file:line each one restates; nothing here is copied from it):
//   1. strand setup     Kmer.hpp:525-542 (2-bit encode), :92-100 (RC), :484-487 (homopolymer) -- setup_strand: four characters
//                       per lane (SWAR classification), a packed 2-bit image + N / non-ACGT masks of the strand in LDS.  A read
//                       that is pure A C G T without a run of k equal bases (nearly all) tabulates nothing: every position
//                       with a whole k-mer is eligible and a probe shifts its word out of the image when it gets there.
//   2. seed probes      RapMapUtils.hpp:65-67,226-239 (khash.find) -- LAZILY, where the collector's walk arrives: probe_first
//                       (the first eligible k-mer and the read's last, both strands' words) and probe_window (up to 32
//                       positions, k-mer and reverse complement, one round of independent bucket loads; with -s only the
//                       positions the capped MMPs will visit).  Results live where they are used: six per-position flags
//                       (eligible x2, probed, k-mer found, reverse complement found, vote entry) are bits of ONE vector
//                       register per strand (lane l owns positions l, 64 + l, ...: struct Strand), intervals in LDS.
//   3. collector        SACollector.hpp:108-362 (operator()), :441-677 (getSAHits_), :366-431 (spotCheck_): collect_read /
//                       get_sa_hits walk the flags -- "next hit at or after p" is a compare + ballot + find-first, a run of
//                       misses a ballot + popcount.
//   4. MMP extension    SASearcher.hpp:88-309 (extendSearchNaive): closed form over <= 64 suffixes, one lane per suffix, in
//                       ONE trip against the packed characters behind every suffix's k-mer (saext / sanext tables);
//                       text path and literal three-binary-search fallback otherwise.
//   5. hits->mappings   HitManager.cpp:691-882 (+ :587-689, :449-493, :308-322): one interval of <= 64 suffixes in
//                       registers, otherwise small u64 lists in LDS (global scratch for the rare > QM_CAP lists).
// pair_merge:           RapMapUtils.hpp:1185-1264 + RapMapSAMapper.cpp:461-551,684-701 (pairs),
//                       RapMapSAMapper.cpp:232-250 (single-end).
#pragma once
#include "qm_wave.h"
#include "../../include/qmap_mi355.h"

namespace qm {

#define QM_CAP 64      // entries per LDS list
#define QM_GCAP 2048   // entries per global-scratch list (2 strands x <1000 SA entries)
#define QM_ICAP 16     // SA-interval hits per strand kept in LDS (more spill to global scratch)
#define QM_IOVF 2048   // ... overflow capacity per strand (>= 64*NS - k + 1 for NS = 32, the long-read kernels)
#define QM_CHUNK 4096  // list elements a wave reserves per bump-allocator round trip (>= QM_GCAP)
#define QM_GSCR_U64 (3 * QM_GCAP + 2 * QM_IOVF * 2)   // u64 words of global scratch per wave
// slots of the context's scalar block (ReadBatch::cursor points at slot 0): bump pointer, qm_counters[6], status, ksw2 task
// count, then the slow queue of -s (reads that overflowed the per-wave scratch: how many, suffixes of the largest)
#ifdef QM_TIMING
#define QM_SC_WORDS 48               // + the phase sums of qm_h2m_kernel at [32, 40)
#else
#define QM_SC_WORDS 32
#endif
#define QM_SC_STATUS 8
#define QM_SC_NTASKS 9
#define QM_SC_SLOWCNT 16
#define QM_SC_SLOWMAX 17
#define QM_SC_SLOWQ 18
#define QM_SC_IVCUR 19             // bump pointer of the SA-interval output
#define QM_SC_SKIPCNT 31           // reads that were skipped, not mapped (ReadBatch::skiplist)
#define QM_SKIP_CAP 4096           // ... of which this many are listed
#define QM_LCNT_SLOW 0x7fffffffu   // lcnt value of a read waiting on the slow queue
#define QM_LCNT_LEAN 0x7ffffffeu   // ... of a read qm_lean_kernel left to the general kernel (qm_lean.inl)
#define QM_LCNT_PAIR 0x7ffffffdu   // lcnt[2u] of a pair qm_duo_kernel merged itself (qm_duo.inl): the pair's records are at loff[2u], their count in pair_cnt[u]
#define QM_DUO_ORPHAN 0xffffff0000000000ULL   // ... second word of an orphan's record (| its MateStatus); a paired hit's is the mate's list element

struct Slot { u64 key; int lb; int ub; };          // hash.bin record / small linear-probing tables, key == ~0 empty
// SA indexes and text positions are UNSIGNED 32-bit on the device: an index whose text needs the reference's int64 instantiation
// (BigSA, text > 2^31 - 1 characters) fits as long as the text is below 2^32 - 2 characters, at half the bytes per SA entry
// and per interval of the reference's int64 form
struct Iv { u32 lb, ub; };                         // seed interval
// Dense k-mer table (round 5: canonical buckets).  A read asks for every k-mer AND its reverse complement (the collector's spot
// checks), so the table is keyed by the CANONICAL word -- the smaller of a k-mer and its reverse complement -- and an entry holds the
// SA intervals of both orientations: one 64-byte HBM sector answers both questions about a read position (the table of rounds 1-4
// spent one sector per orientation, and those sectors were 90 % of the stage-A kernel's traffic).  A bucket is one sector: two
// entries {key, interval of the canonical k-mer (f), interval of its reverse complement (r)}; an interval whose lb is ~0 says
// "that orientation is not in the index".  At least two buckets per key (load <= 25 % of the entries).  A lookup is ONE round of
// two 16-byte loads -- the keys and, in the same round, the interval pair of the orientation asked for -- with no probe chain and
// no dependent load for the value.  Keys are 2k <= 62 bits; ~0 marks an empty entry, bit 63 of key[0] says "a key that hashes here
// was placed in a later bucket" (the only case a lookup walks on: 0.4 % of the buckets at this load).
struct Bucket { u64 key[2]; Iv f[2]; Iv r[2]; u64 pad[2]; };   // 64 B; hipMalloc aligns the array
#define QM_BK_OVF (1ULL << 63)
#define QM_IV_NONE 0xffffffffu
QM_DEV u64 hash_mix(u64 x);
QM_DEV u64 word_rc(u64 w, int k);
inline constexpr u64 bucket_count(long long nkeys) { u64 c = 32; while (c < 2 * (u64)nkeys) c <<= 1; return c; }   // host + device
// Bucket of a key: one 32 x 32 -> 64-bit multiply of the key's two halves, folded (the core of wyhash / "mum").  On the
// k-mers of a transcriptome it fills the buckets like the two-multiply 64-bit finaliser it replaced (Poisson occupancy,
// measured), at a fifth of the issue slots -- 32-bit integer multiplies are quarter rate on this chip.
QM_DEV u32 bucket_hash(u64 key) {
  const u64 p = (u64)((u32)key ^ 0x9E3779B1u) * (u64)((u32)(key >> 32) ^ 0x85EBCA6Bu);
  return (u32)(p >> 32) ^ (u32)p;
}
struct SaInfo { u32 tid; int pos; };               // transcript id + offset in transcript of SA[i]
struct IntRec { u32 b, e; u32 len, q; };           // SAIntervalHit (RapMapUtils.hpp:516-525)

// Perfect-hash (`quasiindex -p`) seed map, flattened: BooPHF levels + FrugalBooMap values
// (include/BooPHF.hpp, include/FrugalBooMap.hpp).  The level bit arrays are re-blocked so that one 64-byte
// sector answers "bit set?" and "rank?" (qm_phflat.h); levelTab[2*i + {0,1}] = {hash domain, first block} of
// level i.  data_/lens_ are merged into one 16-byte record per slot together with the slot's k-mer word itself
// (what the reference re-encodes from the text on every lookup: `Kmer(txt + SA[data_[idx]]) == key`,
// FrugalBooMap.hpp:149-167), so the verification costs no SA and no text read -- 288 GB of HBM pays for it.
#ifndef QM_PH_BLOCK_BITS
#define QM_PH_BLOCK_BITS 384
#endif
#ifndef QM_PH_SPEC
#define QM_PH_SPEC 2      // BooPHF levels looked up per round of loads (measured behind the pre-filter, round 4: 1 -> 164, 2 -> 170, 3 -> 164 M pairs/s)
#endif
struct OvfSlot { u32 key; u32 val; };              // overflow_: interval start -> length (>= 255); key ~0 empty
struct PhRec { u64 key; u32 data; unsigned char len; unsigned char pad[3]; };
struct PhIndex {
  const u64* blocks;            // 8 u64 per block: 6 words of bits, rank of the first bit, 6 x 9-bit popcount prefix
  const u64* levelTab;
  const PhRec* recs;            // {k-mer word of the slot, data_[idx]: SA index where the interval starts, lens_[idx] (255 => overflow)}
  const OvfSlot* ovf; u64 ovfMask;
  const Slot* fin; u64 finMask; // _final_hash: key -> value (in lb), key ~0 empty
  u64 lastbitsetrank, nelem;
  int nb_levels;
  // Membership pre-filter (not part of the reference's structure; it changes no answer): one 64-bit word per ~1-2 keys, four
  // bits set per key (ph_filter_slot).  Round 5: the word is chosen by the CANONICAL k-mer (the smaller of the k-mer and its reverse
  // complement) and the four bits by the orientation as well, so the two strands' questions about a read position -- the k-mer and its
  // reverse complement -- read the same word: one sector per position instead of two.  The large majority of the k-mers a read asks for are NOT in the index (every k-mer
  // over a sequencing error, every opposite-strand k-mer); through the levels such a key costs 3.3 sectors on average before
  // it is rejected, here it costs one.  No false negatives, ~2e-4 false positives (which then take the walk and fail there).
  const u64* filter; u64 filterMask;   // null: no filter
};
// word and bit mask of a key (krc: its reverse complement) in the pre-filter: one 32 x 32 -> 64-bit multiply of the canonical key's
// halves (the bucket hash's product); the word comes from the fold of the product's halves, the four bit positions from 24 bits of it --
// the low ones for a key that is its own canonical form, the next ones for one that is the reverse complement of it
QM_DEV void ph_filter_slot(u64 key, u64 krc, u64 mask, u64& word, u64& bits) {
  const u64 c = krc < key ? krc : key;
  const u64 p = (u64)((u32)c ^ 0x9E3779B1u) * (u64)((u32)(c >> 32) ^ 0x85EBCA6Bu);
  word = (u64)((u32)(p >> 32) ^ (u32)p ^ (u32)(p >> 13)) & mask;
  const u32 lo = krc < key ? (u32)(p >> 24) : (u32)p;
  bits = (1ULL << (lo & 63)) | (1ULL << ((lo >> 6) & 63)) | (1ULL << ((lo >> 12) & 63)) | (1ULL << ((lo >> 18) & 63));
}
inline constexpr u64 ph_filter_words(u64 nelem) { u64 c = 64; while (c < nelem / 2 + 1) c <<= 1; return c; }

struct DevIndex {
  const unsigned char* text;  // n bytes + >= 64 bytes of zero padding
  long long n;
  const u32* SA;
  long long nSA;
  const SaInfo* sainfo;
  const Bucket* slots;        // dense index: hmask + 1 buckets (null for a perfect-hash index)
  u64 hmask;
  const u32* sanext;          // -s: per SA entry, the QM_NEXT_BASES text characters behind its k-mer (sanext_entry), or null
  const struct SaExt* saext;  // per SA entry, the QM_EXT_BASES text characters behind its k-mer (saext_entry), or null
  const struct SaExt2* saext2; // ... the QM_EXT2_BASES characters of the wide edition (saext2_entry), or null: built when reads of 129 .. 256 characters first ask
  const PhIndex* ph;          // perfect-hash index (null for a dense index): the flag the kernels are chosen by
  PhIndex phv;                // ... and its contents, by value: as kernel arguments the fields are scalar loads and the
                              // pointers are known to be global (loaded from a struct in memory they would be generic
                              // pointers, dereferenced with FLAT instructions)
  int k;
};

// stage A launch arguments: reads in, one sorted unique hit list per read out
struct ReadBatch {
  const unsigned char* seq1; const long long* off1;
  const unsigned char* seq2; const long long* off2;   // null => single-end
  long long nreads;        // 2 * pairs, or the number of single-end reads
  u32* lcnt;               // [nreads] list length
  long long* loff;         // [nreads] offset of the list in `lists`
  u64* lists;              // bump-allocated list elements
  u64* cursor;             // bump pointer
  long long lists_cap;
  u64* gscratch;           // per wave: QM_GSCR_U64 words -- of wave gw of the launch, or (gslots != null) of the slot the wave holds while it runs
  int lean_wide;           // qmk_launch_lean: the one-read-per-wavefront edition (reads of up to 256 characters)
  long long read_base;     // index of this launch's read 0 in the call (chunked host-buffer calls launch per chunk with shifted arrays): for the skip list
  int gxcd;                // ... the XCDs the slots are divided among (the device's: 8 on a whole MI355X, 4 / 2 / 1 on a DPX / QPX / CPX partition)
  u32* gslots; int ngslots; // oversubscribed grids: one flag per scratch slot (a multiple of gxcd: a share per XCD, each at least the waves that can be
                           // resident there); a wave takes a free one of its XCD when it starts and gives it back when it ends, so the scratch is
                           // sized by residency, not by the launch
  int* status;             // sticky error flags (bit0: lists overflow, bit1: interval too wide, bit2: read too long, bit4: interval output overflow)
  // SA-interval hits as an output of their own (HitCollectorInfo::fwdSAInts / rcSAInts): read r's records sit at
  // iv_out[iv_off[r] .. + iv_cnt[r]), forward strand first.  Null unless the caller asked for them.
  qm_sa_interval_hit* iv_out; u32* iv_cnt; long long* iv_off; long long iv_cap;
  unsigned char* found_out;      // optional [nreads]: what SACollector::operator() returned
  // stage entry "from intervals" (qm_h2m_kernel): read r's intervals are iv_in[iv_in_off[r] .. iv_in_off[r + 1]), its length len_in[r]
  // iv_in_cnt != null: the (offset, count) form the collector pass of the same call left behind (iv_off / iv_cnt), lengths from off1 / off2
  const qm_sa_interval_hit* iv_in; const long long* iv_in_off; const u32* iv_in_cnt; const int* len_in; const unsigned char* found_in;
  int strict_check, max_interval;
  int sensitive;           // 0: --noSensitive (NIP skipping via SASearcher::lce, k-mer vote instead of coverage)
  double quasi_cov;
  int fuzzy;               // --fuzzyIntersection: lists keep both orientations of a transcript, lcnt bit31 = foundHit
  int max_mmp_ext;         // --maxMMPExtension (QM_F_SEL kernels only)
  float consensus_fraction; // 1 - consensusSlack (MappingConfig, RapMapSAMapper.cpp:184-185), QM_F_SEL only
  struct SelScratch* selscr; // QM_F_SEL: one SelScratch per wave
  // second launch of a -s batch over the slow queue: slot r of the launch maps read slowq[r] on scratch dyn[wave]
  const long long* slowq;
  struct SelScratchDyn* dyn;
  const u64* nreads_dev;   // qm_h2m_kernel: when set, the launch covers min(nreads, *nreads_dev) slots (a queue filled by the kernel before it)
  // reads the device does not map: a read beyond QM_MAX_LONG_READ_LEN characters (code 1), a read whose interval lists outgrow the
  // scratch (code 2: only with max_interval above its default).  Such a read gets an empty result and an entry here (read | code << 56);
  // the rest of the batch is mapped as if it were not there.  Count: scalar slot QM_SC_SKIPCNT.
  u64* skiplist;
  // qm_duo_kernel (qm_duo.inl: the two mates of a pair in one wavefront): when set, a pair whose mates were both mapped there is also
  // MERGED there (mergeLeftRightHits + the per-pair driver): pair_cnt[u] = its hits, lcnt[2u] = QM_LCNT_PAIR, loff[2u] = where its
  // records sit in `lists` (two words per hit), the HitCounters added to cursor[1..6]; stage B only expands the records
  u32* pair_cnt; int max_num_hits, no_orphans, no_dovetail;
};

// stage B launch arguments
struct PairBatch {
  long long n;             // pairs (or single-end reads)
  int paired;
  const long long* off1; const long long* off2;
  const u32* lcnt; const long long* loff; const u64* lists;
  u32* cnt;                // [n+1] hits per unit (pass 1 writes, scan turns into offs)
  const long long* offs;   // [n+1] exclusive scan of cnt (pass 2 reads)
  qm_hit* hits;            // pass 2 output, CSR order
  u64* counters;           // [6] qm_counters
  int max_num_hits, no_orphans, no_dovetail, fuzzy;
  // stage entry "merge only": mergeLeftRightHits[Fuzzy] as a call of its own -- none of the caller's bookkeeping that follows
  // it in processReadsPairSA (the jointHits.size() > maxNumHits clear, --noOrphans, --noDovetail; RapMapSAMapper.cpp:534-551,684-701)
  int merge_only;
  unsigned char* too_many;   // optional [n]: the merge's tooManyHits out-parameter
};

template <int NS>
struct WaveMem {
  u64 buf[3][QM_CAP];              // A, B (sort ping-pong), R (the read's hit list)
  Iv tab[2][64 * NS];              // seed interval of the k-mer at position p of the read / of reverseRead(read)
  u64 planes[2][4][NS + 2];        // per strand: packed 2-bit read (2 rows), N mask, non-ACGT mask
  IntRec ints[2][QM_ICAP];         // recorded SA-interval hits, fwd / rc strand
  alignas(8) unsigned char str[2][64 * NS + 16];   // read, reverseRead(read) (+16: 8-byte over-reads)
  // software pipeline of the persistent loop (see ReadStage below): raw characters of the next read, offsets of the next two
  u32 stage[64 * ((16 * NS + 1 + 63) / 64)];
  u32 ostage[2][4];
};

// ------------------------------------------------------------------ bit helpers
// one bit per read position, NS 64-bit words, wave-uniform (the --noSensitive vote's bitmaps; the walk itself keeps its
// per-position flags in a vector register, see Strand)
template <int NS>
struct Bits {
  u64 w[NS];
};

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
    for (int t = 0; t < 2 * NS; ++t) { const bool m = t == s + ws; lo |= m ? rev[t] : 0ULL; hi |= m ? rev[t + 1] : 0ULL; }
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
  // reversing all 64 bits reverses the base order and swaps the two bits of every base: swap them back
  u64 r = brev64(w);
  r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
  return (~r) >> (2 * (32 - k));
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

// Compile-time feature flags of a stage-A instantiation (the default dense + sensitive kernel carries none of
// the optional code, which would otherwise cost it ~40 VGPRs and a whole wave per SIMD).
// event counters of the lane-emulation build (tests/emu, -DQM_PROFILE): dynamic call counts per read, used with
// the static instruction counts of each routine to see where the VALU issue slots go.  No-ops on the device.
#if defined(QM_EMU) && defined(QM_PROFILE)
extern unsigned long long qm_prof[32];
#define QM_CNT(id, n) (qm_prof[id] += (unsigned long long)(n))
#else
#define QM_CNT(id, n) ((void)0)
#endif
// phase timers of a -DQM_TIMING device build (profiles/ab): QM_T(id) charges the shader-clock time since the wave's
// previous mark to phase id (0 read->LDS, 1 strand setup, 2 hash probe windows, 3 MMP extension, 4 rest of the
// collector, 5 hits->mappings, 6 list write-out + loop); the kernel adds the per-wave sums to B.cursor[9..15].
#if defined(QM_TIMING) && !defined(QM_EMU)
__shared__ u64 qm_tim[4][10];
#define QM_T(id) do { u64 t_ = __builtin_readcyclecounter(); int w_ = (int)(threadIdx.x >> 6);                  \
    if ((threadIdx.x & 63) == 0) { qm_tim[w_][id] += t_ - qm_tim[w_][9]; qm_tim[w_][9] = t_; } } while (0)
#else
#define QM_T(id) ((void)0)
#endif
#define QM_F_PH 1      // perfect-hash (-p) index
#define QM_F_NIP 2     // --noSensitive: NIP skipping + k-mer vote
#define QM_F_SEL 4     // --selAln: chain scoring in the collector (MMPs capped at k + maxMMPExtension), coverage slack 1
#define QM_F_COLLECT 8 // stage entry: the collector alone (intervals + foundHit out, no hit list)

// -s caps every MMP but a read's first at k + maxMMPExtension characters (SACollector.hpp:557-575): what such an extension
// compares is the handful of text characters behind the k-mer of each suffix of the interval.  sanext[i] holds them for
// suffix SA[i], indexed like the interval itself, so the capped extension is ONE trip (to this table) instead of two
// dependent ones (suffix array, then text): QM_NEXT_BASES characters at 2 bits, the first in the highest bits of a
// 28-bit field, and in bits 28-31 how many of them are A C G T before a '$' or the end of the text.
#define QM_NEXT_BASES 14
// The same idea for EVERY extension of a clean strand (round 3): saext[i] holds the QM_EXT_BASES text characters behind the
// k-mer of suffix SA[i] at 2 bits -- three words, the first character in the highest bits of w[0] -- and how many of them
// are A C G T before a '$' or the end of the text.  An MMP extension over an interval of <= 64 suffixes is then ONE trip (a
// 32-byte load per suffix, indexed like the interval itself) instead of two dependent ones (suffix array, then text), and
// its comparison is three XORs and a count-leading-zeros instead of byte compares over 16-byte text chunks.  96 characters
// cover every extension of a read of up to 127 characters; a longer match continues on the text (the old path).
// 32 bytes per suffix-array entry: 8.3 GB for config 2 -- HBM spent to shorten the chain of dependent round trips.
#define QM_EXT_BASES 96
// The entry also carries what hits->mappings will ask about the suffix -- its transcript and the offset in it (the sainfo
// record; the count of valid characters rides in the top 7 bits of the transcript word, so the table serves indices of up
// to 2^25 transcripts) -- so the extension's one trip also brings the (tid, pos) of every suffix of the interval it
// settles on: for a read with one interval (three in four) there is no trip to sainfo at all.
#define QM_EXT_TID_BITS 25
struct SaExt { u64 w[3]; u32 tidnv; int pos; };
QM_DEV SaExt saext_entry(const unsigned char* text, long long n, long long pos, u32 tid, int tpos) {
  SaExt e; e.w[0] = 0; e.w[1] = 0; e.w[2] = 0; e.pos = tpos;
  int nv = 0;
  for (int t = 0; t < QM_EXT_BASES; ++t) {
    if (pos + t >= n) break;
    const unsigned char c = text[pos + t];
    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') break;
    const u64 x = (c >> 1) & 3u;
    const u64 code = x ^ (x >> 1);
    const u64 bit = code << (62 - 2 * (t & 31));
    if (t < 32) e.w[0] |= bit; else if (t < 64) e.w[1] |= bit; else e.w[2] |= bit;
    nv = t + 1;
  }
  e.tidnv = (tid & ((1u << QM_EXT_TID_BITS) - 1)) | ((u32)nv << QM_EXT_TID_BITS);
  return e;
}
// The wide edition for reads of 129 .. 256 characters (the one-read-per-wavefront lean kernel, qm_lean.inl): 224 characters behind the
// k-mer -- seven words -- with the transcript (24 bits), the count of valid characters (8 bits) and the offset: 64 bytes, one sector
// per suffix.  Built at the first call that needs it (like sanext): 16.5 GB for config 2.
#define QM_EXT2_BASES 224
#define QM_EXT2_TID_BITS 24
struct SaExt2 { u64 w[7]; u32 tidnv; int pos; };
QM_DEV SaExt2 saext2_entry(const unsigned char* text, long long n, long long pos, u32 tid, int tpos) {
  SaExt2 e; for (int i = 0; i < 7; ++i) e.w[i] = 0;
  e.pos = tpos;
  int nv = 0;
  for (int t = 0; t < QM_EXT2_BASES; ++t) {
    if (pos + t >= n) break;
    const unsigned char c = text[pos + t];
    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') break;
    const u64 x = (c >> 1) & 3u;
    e.w[t >> 5] |= (x ^ (x >> 1)) << (62 - 2 * (t & 31));
    nv = t + 1;
  }
  e.tidnv = (tid & ((1u << QM_EXT2_TID_BITS) - 1)) | ((u32)nv << QM_EXT2_TID_BITS);
  return e;
}
// where extend_search may leave the (tid, pos) of the suffixes of the interval it returns (LDS; IntervalList::pf)
struct ExtStage { u32* pf; int pfcap; bool done; };
// the query side of such an extension: the strand's characters behind the k-mer, packed the same way (clean strands only)
struct ExtQuery { u64 q[3]; int nq; };   // nq: characters of the query behind the k-mer (0: no packed query, take the text path)

QM_DEV u32 sanext_entry(const unsigned char* text, long long n, long long pos) {
  u32 e = 0; int nv = 0;
  for (int t = 0; t < QM_NEXT_BASES; ++t) {
    if (pos + t >= n) break;
    const unsigned char c = text[pos + t];
    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') break;
    const u32 x = (c >> 1) & 3u;
    e |= (x ^ (x >> 1)) << (26 - 2 * t);
    nv = t + 1;
  }
  return e | ((u32)nv << 28);
}

// khash.find.
// dense: exact lookup in the bucket table (RapMapUtils.hpp:65-67), a whole probe round at a time: find_dense_round.
// perfect hash (find_kmer): FrugalBooMap::find (FrugalBooMap.hpp:149-167) over mphf::lookup (BooPHF.hpp:971-1009,
// getLevel :1318-1351, hash64 :394-407, xorshift next :493-499, fastrange64 :815-820, bitVector::rank :756-769):
// the key is not stored -- the candidate interval's first suffix is re-encoded from the text and compared.
QM_DEV u64 boo_hash64(u64 key, u64 seed) {
  u64 hash = seed;
  hash ^= (hash << 7) ^ key * (hash >> 3) ^ (~((hash << 11) + (key ^ (hash >> 5))));
  hash = (~hash) + (hash << 21);
  hash = hash ^ (hash >> 24);
  hash = (hash + (hash << 3)) + (hash << 8);
  hash = hash ^ (hash >> 14);
  hash = (hash + (hash << 2)) + (hash << 4);
  hash = hash ^ (hash >> 28);
  hash = hash + (hash << 31);
  return hash;
}
QM_DEV u64 fastrange64(u64 word, u64 p) { return mulhi64(word, p); }

// 2-bit word of text[pos, pos+k); false if a non-ACGT character (the '$' separator) is inside
QM_DEV bool text_kmer(const DevIndex& ix, long long pos, int k, u64& w) {
  w = 0;
  bool ok = pos + k <= ix.n;
  for (int base = 0; base < k; base += 8) {
    u64 x = load_u64_unaligned(ix.text + pos + base);       // text is padded
    int nb = k - base < 8 ? k - base : 8;
    for (int t = 0; t < nb; ++t) {
      unsigned c = (unsigned)(x >> (8 * t)) & 0xff;
      bool v = c == 'A' || c == 'C' || c == 'G' || c == 'T';
      unsigned y = (c >> 1) & 3;
      ok = ok && v;
      w = (w << 2) | (u64)(y ^ (y >> 1));
    }
  }
  return ok;
}

// khash.find on the dense table for a whole probe round, without per-lane control flow: every lane issues its bucket load
// (lanes with nothing to look up read bucket 0, a line that stays in cache), key match and value are selects, and the walk to
// a following bucket -- 0.4 % of the buckets carry the overflow mark -- is a wave-level branch taken when any lane needs it.
// A per-lane `while` here costs ~35 scalar instructions of exec-mask bookkeeping per round; the scalar unit is what this
// kernel runs out of first.  krc: the reverse complement of every key (the bucket is that of the smaller).
QM_DEV void find_dense_round(const DevIndex& ix, const LV<u64>& key, const LV<u64>& krc, const LV<bool>& want, LV<bool>& hit, LV<Iv>& val) {
  LV<bool> more; LV<u64> bkt;
  QM_LANES(l) {
    const bool isr = krc[l] < key[l];                      // the key is the reverse complement of its bucket's key
    const u64 ck = isr ? krc[l] : key[l];
    const u64 b = want[l] ? ((u64)bucket_hash(ck) & ix.hmask) : 0ULL;
    U4 a, c;
    const unsigned char* bp = (const unsigned char*)&ix.slots[b];
    load_16x2(bp, bp + (isr ? 32 : 16), a, c);
    QM_CNT(0, want[l] ? 1 : 0); QM_CNT(1, want[l] ? 1 : 0);
    const u64 k0r = ((u64)a.y << 32) | a.x, k1 = ((u64)a.w << 32) | a.z;
    const bool h0 = (k0r & ~QM_BK_OVF) == ck, h1 = k1 == ck;   // (keys are <= 62 bits: an empty entry, ~0, never matches)
    Iv v; v.lb = h0 ? c.x : c.z; v.ub = h0 ? c.y : c.w;
    const bool h = want[l] && (h0 || h1) && v.lb != QM_IV_NONE;
    if (!h) { v.lb = 0; v.ub = 0; }
    hit[l] = h; val[l] = v;
    more[l] = want[l] && !(h0 || h1) && k0r != ~0ULL && (k0r & QM_BK_OVF) != 0;
    bkt[l] = b;
  }
  if (ballot(more)) {
    QM_LANES(l) {
      if (more[l]) {
        const bool isr = krc[l] < key[l];
        const u64 ck = isr ? krc[l] : key[l];
        u64 b = (bkt[l] + 1) & ix.hmask;
        while (true) {
          U4 a, c;
          const unsigned char* bp = (const unsigned char*)&ix.slots[b];
          load_16x2(bp, bp + (isr ? 32 : 16), a, c);
          QM_CNT(1, 1);
          const u64 k0r = ((u64)a.y << 32) | a.x, k1 = ((u64)a.w << 32) | a.z;
          if ((k0r & ~QM_BK_OVF) == ck) { if (c.x != QM_IV_NONE) { hit[l] = true; val[l].lb = c.x; val[l].ub = c.y; } break; }
          if (k1 == ck) { if (c.z != QM_IV_NONE) { hit[l] = true; val[l].lb = c.z; val[l].ub = c.w; } break; }
          if (k0r == ~0ULL || !(k0r & QM_BK_OVF)) break;
          b = (b + 1) & ix.hmask;
        }
      }
    }
  }
}

// The table's one insertion routine (device builders and the emulation's): the entry of the key's canonical word is found or
// created along the bucket chain, the interval goes to the orientation the key has in it (both for a k-mer that is its own reverse
// complement: even k only).  cas / orf: atomic compare-and-swap / or on a u64 (plain operations in the emulation).
template <typename Cas, typename Orf>
QM_DEV void bucket_insert(Bucket* buckets, u64 hmask, u64 key, int k, u32 lb, u32 ub, Cas cas, Orf orf) {
  const u64 rc = word_rc(key, k);
  const u64 ck = rc < key ? rc : key;
  u64 b = (u64)bucket_hash(ck) & hmask;
  while (true) {
    Bucket* bk = &buckets[b];
    int got = -1;
    for (int t = 0; t < 2 && got < 0; ++t) {
      const u64 old = cas(&bk->key[t], ~0ULL, ck);
      if (old == ~0ULL || (old & ~QM_BK_OVF) == ck) got = t;
    }
    if (got >= 0) {
      if (key == ck) { bk->f[got].lb = lb; bk->f[got].ub = ub; }
      if (rc == ck) { bk->r[got].lb = lb; bk->r[got].ub = ub; }
      return;
    }
    orf(&bk->key[0], QM_BK_OVF);                           // full: remember that lookups must walk on
    b = (b + 1) & hmask;
  }
}

// The pre-filter for a whole probe round, without per-lane control flow: every lane loads its word (lanes with nothing to look
// up read word 0), `want` loses the keys that cannot be in the index.  Only the survivors walk the levels.
QM_DEV void ph_filter_round(const DevIndex& ix, const LV<u64>& key, const LV<u64>& krc, LV<bool>& want) {
  const PhIndex& P = ix.phv;
  if (!P.filter) return;
  QM_LANES(l) {
    u64 w, bits;
    ph_filter_slot(key[l], krc[l], P.filterMask, w, bits);
    const u64 x = P.filter[want[l] ? w : 0ULL];
    QM_CNT(1, want[l] ? 1 : 0);
    want[l] = want[l] && (x & bits) == bits;
  }
}

template <int F>
QM_DEV bool find_kmer(const DevIndex& ix, u64 key, u32& lb, u32& ub) {     // perfect-hash flavour only (dense: find_dense_round)
  const PhIndex& P = ix.phv;
  QM_CNT(20, 1);
  u64 s0 = 0, s1 = 0, h = 0;
  u64 idx = 0;
  bool inLevel = false;
  // QM_PH_SPEC levels per round: their loads are issued together, so the walk (whose depth is the
  // maximum over the 64 keys of a probe window) needs half as many dependent trips to HBM.
  const int nlv = P.nb_levels - 1;
  for (int ii = 0; ii < nlv; ii += QM_PH_SPEC) {
    const u64* Bp[QM_PH_SPEC]; int bitp[QM_PH_SPEC]; u64 word[QM_PH_SPEC]; U4 meta[QM_PH_SPEC];
#pragma unroll
    for (int t = 0; t < QM_PH_SPEC; ++t) {
      const int lv = ii + t < nlv ? ii + t : nlv - 1;       // past the last level: a harmless repeat
      if (ii + t < nlv) {
        if (lv == 0) { s0 = boo_hash64(key, 0xAAAAAAAA55555555ULL); h = s0; }
        else if (lv == 1) { s1 = boo_hash64(key, 0x33333333CCCCCCCCULL); h = s1; }
        else { u64 a = s0; const u64 b = s1; s0 = b; a ^= a << 23; s1 = a ^ b ^ (a >> 17) ^ (b >> 26); h = s1 + b; }
      }
      const u64 dom = P.levelTab[2 * lv], fb = P.levelTab[2 * lv + 1];
      const u64 pos = fastrange64(h, dom);
      const u64 blk = pos / QM_PH_BLOCK_BITS;
      bitp[t] = (int)(pos - blk * QM_PH_BLOCK_BITS);
      Bp[t] = P.blocks + (fb + blk) * 8;
    }
    load_8_16xN<QM_PH_SPEC>(Bp, bitp, word, meta);
    int t = -1;
#pragma unroll
    for (int u = QM_PH_SPEC - 1; u >= 0; --u) if (ii + u < nlv && ((word[u] >> (bitp[u] & 63)) & 1)) t = u;
    if (t >= 0) {
      u64 wd = word[0]; U4 mt = meta[0]; int bit = bitp[0];
#pragma unroll
      for (int u = 1; u < QM_PH_SPEC; ++u) if (t == u) { wd = word[u]; mt = meta[u]; bit = bitp[u]; }
      const u64 base = ((u64)mt.y << 32) | mt.x, pref = ((u64)mt.w << 32) | mt.z;
      idx = base + ((pref >> (9 * (bit >> 6))) & 511) + (u64)popc64(wd & ((1ULL << (bit & 63)) - 1));
      inLevel = true;
      break;
    }
  }
  if (!inLevel) {
    if (!P.fin) return false;
    u64 i = hash_mix(key) & P.finMask;
    while (true) {
      Slot x = P.fin[i];
      if (x.key == key) { idx = (u64)(u32)x.lb + P.lastbitsetrank; break; }
      if (x.key == ~0ULL) return false;
      i = (i + 1) & P.finMask;
    }
  }
  if (idx >= P.nelem) return false;                         // FrugalBooMap.hpp:151
  const U4 rr = load_16(&P.recs[idx]);                      // {key lo, key hi, data, len}
  if ((((u64)rr.y << 32) | rr.x) != key) return false;      // Kmer(txt + SA[data_[idx]]) == key, precomputed
  const u32 ind = rr.z;
  u32 l = rr.w & 0xff;
  if (l == 255) {
    u64 i = hash_mix((u64)ind) & P.ovfMask;
    while (true) { OvfSlot x = P.ovf[i]; if (x.key == ind) { l = x.val; break; } if (x.key == ~0u) { l = 0; break; } i = (i + 1) & P.ovfMask; }
  }
  lb = ind; ub = ind + l;
  return true;
}

// ------------------------------------------------------------------ stage 1+2
// One strand of one read.  E/E2 depend on the characters only and are built once; F/C/K are filled
// window by window: the reference consults the hash only at the positions its MMP walk visits
// (~47 finds per read instead of the 2(L-k+1) of an exhaustive pre-probe), and every probe is a random
// 64-byte sector from HBM -- the resource this kernel is bound by (profiles/).
//
// The six per-position flags live in ONE vector register: lane l holds, for each word s < NS, the flags of position
// 64 s + l (bit f * NS + s = flag f).  Round 1 kept them as six NS x 64-bit bitmaps in SGPRs: every test, shift-in and
// find-first was scalar code, 24+ SGPRs were live through the whole walk (151 of them spilled), and the scalar unit --
// one per CU, shared by the four SIMDs -- was what the kernel ran out of first (profiles/r02_stage_a_ablation.txt).
// Here a probe leaves its result in the lane that owns the position, a find-first is a compare + ballot, and the walk's
// state is a register per strand.
enum { FL_E = 0,    // eligible in getSAHits_: no N in [p,p+k), not a homopolymer (SACollector.hpp:498-536)
       FL_E2 = 1,   // eligible in the first-hit scan: no N in [p,p+k] (:176-192, note the <=)
       FL_K = 2,    // the F / C flags of the position are known
       FL_F = 3,    // khash.find(mer) hit
       FL_C = 4,    // khash.find(mer.getRC()) hit
       FL_V = 5 };  // the position produced a KmerDirScore entry (only used by the --noSensitive vote)
// the flag word of a slot count: a dword up to 5 slots, a register pair up to 10, Wide<> beyond (the long-read kernels)
template <int BITS, bool FITS32 = (BITS <= 32), bool FITS64 = (BITS <= 64)> struct FlagWord { typedef Wide<(BITS + 63) / 64> type; };
template <int BITS> struct FlagWord<BITS, true, true> { typedef u32 type; };
template <int BITS> struct FlagWord<BITS, false, true> { typedef u64 type; };
template <int NS>
struct Strand {
  typedef typename FlagWord<6 * NS>::type FT;
  LV<FT> fl;
  static QM_DEV FT bit(int f, int s) { return (FT)1 << (f * NS + s); }
  // flag word of the lane that owns position p (wave-uniform), and one flag of it
  QM_DEV FT word_at(int p) const { return read_lane(fl, p & 63); }
  static QM_DEV bool flag(FT w, int f, int p) { return (unsigned)p < 64u * NS && ((w >> (f * NS + (p >> 6))) & 1) != 0; }
  QM_DEV bool test(int f, int p) const { return flag(word_at(p), f, p); }
  // fl |= m << (p >> 6) in the lane that owns p; m holds flags of word 0
  QM_DEV void or_at(int p, FT m) {
    if ((unsigned)p >= 64u * NS) return;
    const FT mm = m << (p >> 6);
    QM_LANES(l) { fl[l] |= (l == (p & 63)) ? mm : (FT)0; }
  }
  const u64* planes;   // LDS: [4][NS+2]
  Iv* tab;             // LDS: interval of mer at position p (valid where F)
  bool dollar;         // the strand's string contains '$' (extensions then take the literal binary searches)
  bool lazy;           // pure-ACGT read without a long run: tab holds no k-mer words, a probe shifts its word out of the image
  bool clean;          // nothing but A C G T in the strand: the packed image alone describes it
  int P;
};

// first position >= p whose flags satisfy pred(word, s) (64 NS: none)
template <int NS, typename Pred>
QM_DEV int fl_first(const Strand<NS>& S, int p, Pred pred) {
  int res = 64 * NS;
#pragma unroll
  for (int s = NS - 1; s >= 0; --s) {
    LV<bool> pr;
    QM_LANES(l) { pr[l] = pred(S.fl[l], s) && 64 * s + l >= p; }
    const u64 m = ballot(pr);
    if (m) res = 64 * s + ctz64(m);
  }
  return res;
}
// number of positions in [a, e) whose flags satisfy pred
template <int NS, typename Pred>
QM_DEV int fl_count(const Strand<NS>& S, int a, int e, Pred pred) {
  int c = 0;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    LV<bool> pr;
    QM_LANES(l) { const int q = 64 * s + l; pr[l] = pred(S.fl[l], s) && q >= a && q < e; }
    c += popc64(ballot(pr));
  }
  return c;
}
// one flag as a bitmap over the positions (the --noSensitive vote works on those)
template <int NS>
QM_DEV Bits<NS> fl_bits(const Strand<NS>& S, int f) {
  Bits<NS> r;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    LV<bool> pr;
    QM_LANES(l) { pr[l] = ((S.fl[l] >> (f * NS + s)) & 1) != 0; }
    r.w[s] = ballot(pr);
  }
  return r;
}

// ---- four characters at a time (one dword per lane) ----
// 0x80 in every byte of x that equals c (exact zero-byte test of x ^ cccc)
QM_DEV u32 eq_bytes(u32 x, u32 c) {
  const u32 t = x ^ (c * 0x01010101u);
  return ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);
}
// bit j = bit 7 of byte j
QM_DEV u32 movemask4(u32 m) { return (((m >> 7) & 0x01010101u) * 0x01020408u) >> 24; }
// ::toupper of every byte (C locale: only 'a'..'z' change)
QM_DEV u32 upcase4(u32 d) {
  const u32 lo = d & 0x7f7f7f7fu;
  const u32 ge_a = lo + 0x1f1f1f1fu, gt_z = lo + 0x05050505u;          // bit 7: byte >= 'a' / byte > 'z'
  const u32 lower = ge_a & ~gt_z & ~d & 0x80808080u;
  return d ^ (lower >> 2);
}
// canonical upper-case letter of every byte whose low three bits name one of A C G T [U] (0xFF elsewhere): a byte b is that
// nucleotide, in either case, exactly when (b & 0xDF) equals it
QM_DEV u32 canon4(u32 d, bool withU) {
  // index: 1 'A', 3 'C', 4 'T', 5 'U', 7 'G'
  return perm8(withU ? 0x47ff5554u : 0x47ffff54u, 0x43ff41ffu, d & 0x07070707u);
}
// src/RapMapUtils.cpp:63-72 reverseRead table, four characters: A<->T, C<->G, U->A, everything else 'N' (byte order kept)
QM_DEV u32 rc_char4(u32 d) {
  const u32 canon = canon4(d, true);
  const u32 t = (d & 0xdfdfdfdfu) ^ canon;
  const u32 valid = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);    // 0x80 where the byte is A C G T U (either case)
  const u32 comp = perm8(0x43ff4141u, 0x47ff54ffu, d & 0x07070707u);           // 1->'T' 3->'G' 4->'A' 5->'A' 7->'C'
  const u32 vm = (valid >> 7) * 0xffu;
  return (comp & vm) | (0x4e4e4e4eu & ~vm);
}

// LDS image of one strand ("planes", [4][NS+2] u64): rows 0-1 hold the read packed 2 bits per base,
// 32 bases per word, first base in the highest bits (PK[0 .. 2NS], one zero word of padding); row 2 the
// 'N' mask and row 3 the non-ACGT mask, one bit per base (NS words + two words of padding: 0 / ~0).
//
// k-mer word at position p of the strand (partial-word semantics of Kmer.hpp:535-538), plus window flags:
// the word is a funnel shift of two adjacent packed words.
template <int NS>
QM_DEV u64 kmer_at(const u64* planes, int p, int k, bool& nwin, bool& nwin2, int& d) {
  const u64* PK = planes; const u64* NM = planes + 2 * (NS + 2); const u64* INV = planes + 3 * (NS + 2);
  const int j = p >> 5, sh = 2 * (p & 31);
  u64 w = ((PK[j] << sh) | ((PK[j + 1] >> 1) >> (63 - sh))) >> (64 - 2 * k);
  const int s = p >> 6, l = p & 63;
  const u64 nmw = (NM[s] >> l) | ((NM[s + 1] << 1) << (63 - l));
  const u64 ivw = (INV[s] >> l) | ((INV[s + 1] << 1) << (63 - l));
  const u64 maskk = (1ULL << k) - 1, maskk1 = (1ULL << (k + 1)) - 1;
  nwin = (nmw & maskk) != 0; nwin2 = (nmw & maskk1) != 0;
  d = k;
  if (ivw & maskk) {                           // rare: only the characters before the first non-ACGT one count
    d = ctz64(ivw);
    w &= ~0ULL << (2 * (k - d));
  }
  return w;
}

// k-mer word at position p of a strand whose characters are all A C G T (Strand::lazy)
QM_DEV u64 clean_kmer(const u64* planes, int p, int k) {
  const int j = p >> 5, sh = 2 * (p & 31);
  return ((planes[j] << sh) | ((planes[j + 1] >> 1) >> (63 - sh))) >> (64 - 2 * k);
}

template <int NS>
QM_DEV void setup_strand(const DevIndex& ix, const unsigned char* str, int L, Strand<NS>& S, u64* planes, Iv* tab) {
  const int k = ix.k;
  const int P = L - k + 1;
  QM_CNT(2, 1); QM_T(4);
  S.planes = planes; S.tab = tab; S.P = P;
  // Four characters per lane, straight from the string's words in LDS: SWAR classification, the four 2-bit codes of a lane
  // folded into one byte of the packed image by a multiply, the mask nibbles of two neighbouring lanes joined over DPP --
  // every byte of the image is written exactly once (no zeroing pass, no LDS atomics, no ballots).
  unsigned char* PKb = (unsigned char*)planes;
  unsigned char* NMb = (unsigned char*)(planes + 2 * (NS + 2));
  unsigned char* IVb = (unsigned char*)(planes + 3 * (NS + 2));
  constexpr int NC = (NS + 3) / 4;
  u64 dirty = 0;                                                           // lanes holding a character that is not A C G T
  int runs = 0;                                                            // lanes whose four characters are one repeated base
  S.lazy = false; S.clean = false;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    LV<u32> nn, iv, nn2, iv2; LV<bool> bad, rep;
    QM_LANES(l) {
      const int base = 256 * c + 4 * l;
      u32 pk = 0, nnib = 0, vnib = 0;
      bad[l] = false; rep[l] = false;
      if (base < 64 * NS) {
        const u32 d = ((const u32*)str)[base >> 2];
        const int nb = L - base;                                          // characters of this word inside the read
        const u32 lenmask = nb >= 4 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
        const u32 t = (d & 0xdfdfdfdfu) ^ canon4(d, false);
        const u32 valid = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu) & lenmask;    // 0x80: A C G T in either case
        const u32 nmask = eq_bytes(d | 0x20202020u, (u32)'n') & lenmask;
        const u32 x = (d >> 1) & 0x03030303u;
        const u32 code = (x ^ ((x >> 1) & 0x01010101u)) & ((valid >> 7) * 3u);                  // A0 C1 G2 T3 (Kmer.hpp:40-51)
        pk = (code * 0x40100401u) >> 24;                                                        // first character in the top bits
        nnib = movemask4(nmask); vnib = movemask4(valid);
        bad[l] = (~valid & lenmask & 0x80808080u) != 0;
        rep[l] = nb >= 4 && ((pk ^ (pk >> 2)) & 0x3fu) == 0;
      }
      const int bi = 64 * c + l;                                          // byte of the packed image: characters 4 bi .. 4 bi + 3
      if ((bi >> 3) < 2 * NS + 2) PKb[8 * (bi >> 3) + 7 - (bi & 7)] = (unsigned char)pk;
      nn[l] = nnib; iv[l] = (~vnib) & 0xfu;
    }
    dirty |= ballot(bad);
    runs += popc64(ballot(rep));
    lane_xor1(nn, nn2); lane_xor1(iv, iv2);
    QM_LANES(l) {
      const int mb = 32 * c + (l >> 1);                                   // byte of the masks: characters 8 mb .. 8 mb + 7
      if (!(l & 1) && mb < 8 * (NS + 2)) { NMb[mb] = (unsigned char)(nn[l] | (nn2[l] << 4)); IVb[mb] = (unsigned char)(iv[l] | (iv2[l] << 4)); }
    }
  }
  QM_LANES(l) {                                                           // padding words the lanes above do not reach
    if (2 * NS + 2 > 8 * NC && l < 2 * NS + 2 - 8 * NC) planes[8 * NC + l] = 0;
    if (NS + 2 > 4 * NC && l < NS + 2 - 4 * NC) { planes[2 * (NS + 2) + 4 * NC + l] = 0; planes[3 * (NS + 2) + 4 * NC + l] = ~0ULL; }
  }
  wave_fence();
  S.clean = dirty == 0;
  if (!dirty && 4 * runs + 6 < k) {
    // nearly every read: nothing but A C G T and no homopolymer window (k equal characters cover at least (k - 6) / 4 whole
    // lanes of the loop above).  Every position with a whole k-mer is eligible and nothing is tabulated: a probe shifts
    // its word out of the packed image when it gets there (clean_kmer), tab only ever receives intervals.
    QM_LANES(l) {
      typename Strand<NS>::FT f = 0;
#pragma unroll
      for (int s = 0; s < NS; ++s) f |= (64 * s + l < P) ? (Strand<NS>::bit(FL_E, s) | Strand<NS>::bit(FL_E2, s)) : 0;
      S.fl[l] = f;
    }
    S.lazy = true;
    QM_T(1);
    return;
  }
  QM_LANES(l) { S.fl[l] = 0; }
  if (!dirty) {
    // pure A C G T with a long run: no window holds an N or a partial word -- the k-mer at p is one funnel shift of two
    // packed words and the only thing to test is the homopolymer rule
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      QM_LANES(l) {
        const int p = 64 * s + l;
        const int j = p >> 5, sh = 2 * (p & 31);
        const u64 w = ((planes[j] << sh) | ((planes[j + 1] >> 1) >> (63 - sh))) >> (64 - 2 * k);
        const bool inP = p < P;
        if (inP && !homopolymer(w, k)) S.fl[l] |= Strand<NS>::bit(FL_E, s) | Strand<NS>::bit(FL_E2, s);
        if (inP) ((u64*)tab)[p] = w;
      }
    }
    QM_T(1);
    return;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    QM_LANES(l) {
      int p = 64 * s + l;
      bool nwin, nwin2; int d;
      u64 w = kmer_at<NS>(planes, p, k, nwin, nwin2, d);
      bool hom = homopolymer(w, k);
      bool inP = p < P;
      if (inP && !nwin && !hom) S.fl[l] |= Strand<NS>::bit(FL_E, s);
      if (inP && !nwin2 && !hom) S.fl[l] |= Strand<NS>::bit(FL_E2, s);
      // the k-mer word waits in the position's interval slot until the position is probed (~0: an N in the window)
      if (inP) ((u64*)tab)[p] = nwin ? ~0ULL : w;
    }
  }
  QM_T(1);
}

// all k characters at [p, p+k) are ACGT, i.e. Kmer::fromChars succeeds (SACollector.hpp:602); p is wave-uniform
template <int NS>
QM_DEV bool all_acgt(const Strand<NS>& S, int p, int k) {
  if (p >= S.P) return false;
  const u64* INV = S.planes + 3 * (NS + 2);
  const int s = p >> 6, l = p & 63;
  const u64 ivw = uniform((INV[s] >> l) | ((INV[s + 1] << 1) << (63 - l)));
  return (ivw & ((1ULL << k) - 1)) == 0;
}

// Probe positions [p, p+width) (width <= 32), k-mer and reverse complement -- khash.find (RapMapUtils.hpp:65-67) -- in one
// round of independent loads.  Lane t takes the position whose flags it owns, or whose flags its partner t ^ 32 owns:
//   d = (t - p) & 63:   d < 32: the k-mer of position p + d (that position lives in lane t);
//                       d >= 32: the reverse complement of the k-mer of position p + d - 32 (which lives in lane t ^ 32)
// so a result is where it is kept, or one lane swap away from it.
// `stride` (a power of two): only every stride-th position of the window is looked up -- the -s walk, whose capped MMPs advance by
// exactly maxMMPExtension + 1 positions while the read keeps matching, never asks about the positions in between (each of
// which would cost two random sectors); a position that was not looked up stays unknown, and the walk probes it when it gets there.
template <int NS, int F>
QM_DEV void probe_window(const DevIndex& ix, Strand<NS>& S, int p, int width, int stride = 1) {
  typedef typename Strand<NS>::FT FT;
  const int k = ix.k;
  if (p + width > S.P) width = S.P - p;
  if (width <= 0) return;
  QM_CNT(3, 1); QM_CNT(4, (width + stride - 1) / stride); QM_T(4);
  LV<FT> flx;
  swap32(S.fl, flx);
  LV<bool> found, fresh, want; LV<u64> kq, kr; LV<int> posv; LV<Iv> val;
  QM_LANES(l) {
    const int d = (l - p) & 63, j = d & 31;
    const bool in = j < width && (j & (stride - 1)) == 0;
    const int q = in ? p + j : p;
    const FT w = d >= 32 ? flx[l] : S.fl[l];
    // positions probed before keep their flags and their interval (the slot no longer holds the word)
    fresh[l] = in && ((w >> (FL_K * NS + (q >> 6))) & 1) == 0;
    posv[l] = q;
    // every lane fetches its word before any lane replaces one by an interval
    const u64 w0 = S.lazy ? clean_kmer(S.planes, q, k) : ((const u64*)S.tab)[q];
    const u64 rc = word_rc(w0, k);
    want[l] = fresh[l] && w0 != ~0ULL;                     // ~0: an N in the window
    kq[l] = d >= 32 ? rc : w0; kr[l] = d >= 32 ? w0 : rc;
  }
  wave_fence();
  if (!(F & QM_F_PH)) find_dense_round(ix, kq, kr, want, found, val);
  else {
    ph_filter_round(ix, kq, kr, want);
    QM_LANES(l) {
      bool hit = false; Iv v = {0, 0};
      if (want[l]) hit = find_kmer<F>(ix, kq[l], v.lb, v.ub);
      found[l] = hit; val[l] = v;
    }
  }
  LV<u32> fo, fx;
  QM_LANES(l) { fo[l] = found[l] ? 1u : 0u; }
  swap32(fo, fx);                                          // the partner's result: the same position's reverse complement
  QM_LANES(l) {
    if (fresh[l] && ((l - p) & 63) < 32) {
      const int s = posv[l] >> 6;
      S.fl[l] |= Strand<NS>::bit(FL_K, s) | (found[l] ? Strand<NS>::bit(FL_F, s) : (FT)0) | (fx[l] ? Strand<NS>::bit(FL_C, s) : (FT)0);
      S.tab[posv[l]] = val[l];
    }
  }
  wave_fence();
  QM_T(2);
}

// The first probe of a read: position p (lanes 0 / 32: k-mer / reverse complement) and, in the same round of
// loads, the read's last k-mer (position P-1, lanes 1 / 33).  The reverse complement of the last k-mer is the
// FIRST k-mer of reverseRead(read), i.e. exactly what the reverse-complement pass would have to look up first;
// its interval goes to rtab0 (= that strand's tab[0]).  Returns true when position P-1 was looked up here.
template <int NS, int F>
QM_DEV bool probe_first(const DevIndex& ix, Strand<NS>& S, int p, Iv* rtab0) {
  typedef typename Strand<NS>::FT FT;
  const int k = ix.k;
  const int last = S.P - 1;
  QM_CNT(3, 1); QM_CNT(4, last != p ? 2 : 1); QM_T(4);
  LV<bool> found, want, on; LV<u64> kq, kr; LV<Iv> val;
  QM_LANES(l) {
    const int j = l & 31;
    on[l] = j == 0 || (j == 1 && last != p);
    const int q = j == 0 ? p : last;
    const u64 key = S.lazy ? clean_kmer(S.planes, q, k) : ((const u64*)S.tab)[q];
    const u64 rc = word_rc(key, k);
    want[l] = on[l] && key != ~0ULL;
    kq[l] = l >= 32 ? rc : key; kr[l] = l >= 32 ? key : rc;
  }
  wave_fence();
  if (!(F & QM_F_PH)) find_dense_round(ix, kq, kr, want, found, val);
  else {
    ph_filter_round(ix, kq, kr, want);
    QM_LANES(l) {
      bool hit = false; Iv v = {0, 0};
      if (want[l]) hit = find_kmer<F>(ix, kq[l], v.lb, v.ub);
      found[l] = hit; val[l] = v;
    }
  }
  QM_LANES(l) {
    if (on[l]) {
      const int pos = (l & 31) == 0 ? p : last;
      if (l < 32) S.tab[pos] = val[l];
      else if (pos == last) *rtab0 = val[l];
    }
  }
  const u64 fm = ballot(found);
  S.or_at(p, Strand<NS>::bit(FL_K, 0) | ((fm & 1) ? Strand<NS>::bit(FL_F, 0) : (FT)0) | (((fm >> 32) & 1) ? Strand<NS>::bit(FL_C, 0) : (FT)0));
  if (last != p)
    S.or_at(last, Strand<NS>::bit(FL_K, 0) | (((fm >> 1) & 1) ? Strand<NS>::bit(FL_F, 0) : (FT)0) | (((fm >> 33) & 1) ? Strand<NS>::bit(FL_C, 0) : (FT)0));
  wave_fence();
  QM_T(2);
  return true;
}

// first position >= p that is NOT yet probed (or 64*NS)
template <int NS> QM_DEV int known_end(const Strand<NS>& S, int p) {
  return fl_first(S, p, [](typename Strand<NS>::FT w, int s) { return ((w >> (FL_K * NS + s)) & 1) == 0; });
}

QM_DEV void skip_read(const ReadBatch& B, long long read, int code) {
  QM_LANES(l) {
    if (l == 0) {
      const u64 i = atomic_add_u64(B.cursor + QM_SC_SKIPCNT, 1ULL);
      if (B.skiplist && i < QM_SKIP_CAP) B.skiplist[i] = (u64)(read + B.read_base) | ((u64)code << 56);
    }
  }
}

// SA-interval hits of one strand: the first QM_ICAP in LDS, the rest in the wave's global scratch
struct IntervalList {
  QM_LDS(IntRec)* lds; IntRec* ovf;
  int n;
  u32* pf; int pfcap;      // LDS staging for the (tid, pos) of the first interval's suffixes: tids at pf[0..), positions at pf[pfcap..)
  QM_DEV void push(u32 lb, u32 ub, u32 ln, u32 qp) {
    IntRec r; r.b = lb; r.e = ub; r.len = ln; r.q = qp;
    // a branch per home, not one pointer that is either: that would be a FLAT store (vector-memory path even into LDS)
    if (n < QM_ICAP) { QM_LANES(l) { if (l == 0) { lds[n].b = r.b; lds[n].e = r.e; lds[n].len = r.len; lds[n].q = r.q; } } }
    else { QM_LANES(l) { if (l == 0) ovf[n - QM_ICAP] = r; } }
    ++n;
    wave_fence();
  }
  QM_DEV void get(int i, u32& lb, u32& ub, u32& ln, u32& qp) const {
    IntRec r;
    if (i < QM_ICAP) { r.b = lds[i].b; r.e = lds[i].e; r.len = lds[i].len; r.q = lds[i].q; }
    else r = ovf[i - QM_ICAP];
    lb = uniform(r.b); ub = uniform(r.e); ln = uniform(r.len); qp = uniform(r.q);
  }
};

// ------------------------------------------------------------------ stage 4
// first index i >= i0 at which the comparison loop of SASearcher.hpp:154-180 stops;
// rel: 0 ran off the query/text, 1 query char < text char, 2 query char > text char
QM_DEV int cmp_from(const DevIndex& ix, long long s, const unsigned char* q, int m, int i0, int sentinel,
                    int& rel) {
  int b = i0;
  while (true) {
    QM_CNT(16, 1);
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
// the text, SASearcher.hpp:154,180 -- needs the query to match the text's final '$'; reads that
// contain '$' therefore take the literal path below: Strand::dollar, found while the read is loaded.)
// The wave is split into groups of G = 64 / 2^ceil(log2(width)) lanes, one group per suffix; lane c of a group
// compares the 16 bytes at offset 16c of the current round, so a 2x100 bp read needs one round for up to 8
// suffixes: one coalesced SA load, then one round of text loads, instead of a dependent load per 8 bytes.
QM_DEV bool extend_search_wide(const DevIndex& ix, u32 lbIn, u32 ubIn, int startAt, const unsigned char* q, int m0,
                               u32& lbOut, u32& ubOut, int& lenOut, u32 qn, int nq, const ExtQuery* xq, ExtStage* xs) {
  const int width = (int)(ubIn - lbIn - 1);
  if (width < 1 || width > 64) return false;
  if (xq && xq->nq >= 0 && ix.saext && !(nq > 0 && ix.sanext)) {
    // a clean strand: the query's characters behind the k-mer against the packed characters behind every suffix's k-mer --
    // one lane per suffix, one 32-byte load each, no trip to the suffix array or the text
    const int rem = xq->nq;                               // characters of the query behind the first startAt
    const int cap = rem < QM_EXT_BASES ? rem : QM_EXT_BASES;
    LV<int> lc, tpv; LV<u32> tdv; LV<bool> on, full;
    QM_LANES(l) {
      int v = -1; bool fl = false;
      tdv[l] = 0; tpv[l] = 0;
      if (l < width) {
        U4 a, b;
        load_32(&ix.saext[lbIn + 1 + l], a, b);
        const u64 w0 = ((u64)a.y << 32) | a.x, w1 = ((u64)a.w << 32) | a.z, w2 = ((u64)b.y << 32) | b.x;
        const int nv = (int)(b.z >> QM_EXT_TID_BITS);
        tdv[l] = b.z & ((1u << QM_EXT_TID_BITS) - 1); tpv[l] = (int)b.w;
        const u64 x0 = w0 ^ xq->q[0], x1 = w1 ^ xq->q[1], x2 = w2 ^ xq->q[2];
        int matched = x0 ? (clz64(x0) >> 1) : (x1 ? 32 + (clz64(x1) >> 1) : (x2 ? 64 + (clz64(x2) >> 1) : 96));
        matched = matched < nv ? matched : nv;
        matched = matched < cap ? matched : cap;
        fl = matched == QM_EXT_BASES && rem > QM_EXT_BASES;   // the match may go on behind what the table holds
        v = startAt + matched;
      }
      lc[l] = v; on[l] = l < width; full[l] = fl;
    }
    if (!ballot(full)) {
      const int mxq = wave_max(lc);
      LV<bool> bestq;
      QM_LANES(l) { bestq[l] = on[l] && lc[l] == mxq; }
      const u64 bq = ballot(bestq);
      const int first = ctz64(bq), cnt = (63 - clz64(bq)) + 1 - first;
      lbOut = lbIn + 1 + first;
      ubOut = lbOut + cnt;
      lenOut = mxq;
      if (xs && xs->pf && cnt <= xs->pfcap) {               // the (tid, pos) of the block's suffixes, where hits->mappings looks for them
        QM_LANES(l) { if (l >= first && l < first + cnt) { xs->pf[l - first] = tdv[l]; xs->pf[xs->pfcap + l - first] = (u32)tpv[l]; } }
        xs->done = true;
        wave_fence();
      }
      return true;
    }
    // (rare: reads longer than k + 96 that match that far -- the text path below)
  }
  if (nq > 0 && ix.sanext) {
    // a capped extension of a clean strand: the nq (<= QM_NEXT_BASES) query characters behind the k-mer, packed like the
    // table's entries (qn), against the entry of every suffix of the interval -- one lane per suffix, one load
    LV<int> lc; LV<bool> on;
    QM_LANES(l) {
      int v = -1;
      if (l < width) {
        const u32 e = ix.sanext[lbIn + 1 + l];
        const u32 x = ((e & 0x0fffffffu) >> (28 - 2 * nq)) ^ qn;
        int matched = x ? ((__builtin_clz(x) - (32 - 2 * nq)) >> 1) : nq;
        const int nv = (int)(e >> 28);
        matched = matched < nv ? matched : nv;
        v = startAt + matched;
      }
      lc[l] = v; on[l] = l < width;
    }
    const int mxq = wave_max(lc);
    LV<bool> bestq;
    QM_LANES(l) { bestq[l] = on[l] && lc[l] == mxq; }
    const u64 bq = ballot(bestq);
    lbOut = lbIn + 1 + ctz64(bq);
    ubOut = lbIn + 1 + (63 - clz64(bq)) + 1;
    lenOut = mxq;
    return true;
  }
  QM_CNT(5, 1); QM_CNT(6, width); QM_CNT(9, width == 1);
  int lg = 0;
  while ((1 << lg) < width) ++lg;
  const int gs = 6 - lg, G = 1 << gs;                 // lanes per suffix
  LV<long long> sv; LV<int> lcp; LV<bool> act;
  QM_LANES(l) {
    const int j = l >> gs;
    bool a = j < width;
    sv[l] = a ? (long long)ix.SA[lbIn + 1 + j] : 0;
    lcp[l] = a ? 0x7fffffff : -1;                     // 0x7fffffff: still matching
    act[l] = a;
  }
  for (int i0 = startAt; ; i0 += 16 * G) {
    QM_CNT(7, 1);
    LV<int> cand;
    QM_LANES(l) {
      int cd = 0x7fffffff;
      if (act[l] && lcp[l] == 0x7fffffff) {
        const int off = i0 + 16 * (l & (G - 1));
        int nb = m0 - off; nb = nb > 16 ? 16 : nb;
        int cnt = 0;
        if (nb > 0) {
          // 16 query bytes at q[off]: aligned LDS words + funnel shift (rows are padded for the over-read)
          // (the aligned pointer by pointer arithmetic: through an integer it would lose its address space and the three
          // loads would be FLAT instructions instead of LDS reads)
          const int mis = (int)((unsigned long long)(q + off) & 7ULL);
          const u64* al = (const u64*)(q + off - mis);
          const int sh = mis * 8;
          const u64 w0 = al[0], w1 = al[1], w2 = al[2];
          const u64 q0 = (w0 >> sh) | ((w1 << 1) << (63 - sh)), q1 = (w1 >> sh) | ((w2 << 1) << (63 - sh));
          long long tv = ix.n - (sv[l] + off);            // text bytes left (the array is padded for the over-read)
          if (tv > 0) {
            const u64 t0 = load_u64_unaligned(ix.text + sv[l] + off), t1 = load_u64_unaligned(ix.text + sv[l] + off + 8);
            const u64 x0 = t0 ^ q0, x1 = t1 ^ q1;
            cnt = x0 ? (ctz64(x0) >> 3) : (x1 ? 8 + (ctz64(x1) >> 3) : 16);
            if (cnt > nb) cnt = nb;
            if ((long long)cnt > tv) cnt = (int)tv;
          }
        }
        if (cnt < 16) cd = off + cnt;                    // this chunk ends the match (mismatch, end of query or text)
      }
      cand[l] = cd;
    }
    group_min(cand, G);
    LV<bool> cont;
    QM_LANES(l) {
      bool c = false;
      if (act[l] && lcp[l] == 0x7fffffff) { if (cand[l] != 0x7fffffff) lcp[l] = cand[l]; else c = true; }
      cont[l] = c;
    }
    if (!ballot(cont)) break;
  }
  int mx = wave_max(lcp);
  LV<bool> best;
  QM_LANES(l) { best[l] = act[l] && (l & (G - 1)) == 0 && lcp[l] == mx; }
  u64 bm = ballot(best);
  lbOut = lbIn + 1 + (ctz64(bm) >> gs);
  ubOut = lbIn + 1 + ((63 - clz64(bm)) >> gs) + 1;
  lenOut = mx;
  return true;
}

// SASearcher::extendSearchNaive (SASearcher.hpp:88-309)
QM_DEV void extend_search(const DevIndex& ix, u32 lbIn, u32 ubIn, int startAt, const unsigned char* q, int m0,
                          u32& lbOut, u32& ubOut, int& lenOut, bool qDollar, u32 qn = 0, int nq = 0, const ExtQuery* xq = nullptr,
                          ExtStage* xs = nullptr) {
  int rel;
  if (!qDollar && extend_search_wide(ix, lbIn, ubIn, startAt, q, m0, lbOut, ubOut, lenOut, qn, nq, xq, xs)) return;
  QM_CNT(8, 1);
  if (ubIn - lbIn == 2) {                         // :109-126
    lbIn += 1;
    long long s = (long long)uniform(ix.SA[lbIn]);
    int i = cmp_from(ix, s, q, m0, startAt, 0, rel);
    lbOut = lbIn; ubOut = ubIn; lenOut = i;
    return;
  }
  long long l = lbIn, r = ubIn, c;
  int lcpLo = startAt, lcpHi = startAt, seenBelow = startAt, seenAbove = startAt, maxLen = 0, i;
  while (true) {                                  // :150-209
    c = (l + r) / 2;
    i = lcpLo < lcpHi ? lcpLo : lcpHi;
    long long s = (long long)uniform(ix.SA[c]);
    i = cmp_from(ix, s, q, m0, i, 0, rel);
    bool plt = rel != 2;
    if (rel == 2) { if (i > seenBelow) seenBelow = i; }
    else { if (i > seenAbove) seenAbove = i; }    // q<t mismatch, or ran off either end
    if (plt) {
      if (c == l + 1) { maxLen = i > seenBelow ? i : seenBelow; if (seenAbove > maxLen) maxLen = seenAbove; break; }
      r = c; lcpHi = i;
    } else {
      if (c == r - 1) { maxLen = i > seenBelow ? i : seenBelow; if (seenAbove > maxLen) maxLen = seenAbove; break; }
      l = c; lcpLo = i;
    }
  }
  int m = maxLen + 1;
  long long firstAt = 0, pastAt = 0;
  for (int pass = 0; pass < 2; ++pass) {          // :215-258, :261-304
    int sentinel = pass == 0 ? '#' : '{';
    l = pass == 0 ? (long long)lbIn : firstAt - 1;
    r = ubIn; lcpLo = startAt; lcpHi = startAt;
    long long res;
    while (true) {
      c = (l + r) / 2;
      i = lcpLo < lcpHi ? lcpLo : lcpHi;
      long long s = (long long)uniform(ix.SA[c]);
      i = cmp_from(ix, s, q, m, i, sentinel, rel);
      if (rel != 2) { if (c == l + 1) { res = c; break; } r = c; lcpHi = i; }
      else { if (c == r - 1) { res = r; break; } l = c; lcpLo = i; }
    }
    if (pass == 0) firstAt = res; else pastAt = res;
  }
  if (firstAt == pastAt) pastAt += 1;              // :307
  lbOut = (u32)firstAt; ubOut = (u32)pastAt; lenOut = maxLen;
}

// SASearcher::lce (SASearcher.hpp:318-334), NIP only.  Restated literally, including its double use of
// startAt (o1/o2 already contain it and `len` starts from it again).
QM_DEV int lce_wave(const DevIndex& ix, u32 p1, u32 p2, int startAt, int stopAt) {
  const long long o1 = (long long)uniform(ix.SA[p1]) + startAt, o2 = (long long)uniform(ix.SA[p2]) + startAt;
  const long long maxIndex = o1 > o2 ? o1 : o2;
  const long long textLen = ix.n;
  int base = startAt;
  while (true) {
    LV<bool> stopv;
    QM_LANES(l) {
      long long j = (long long)base + l;
      bool inb = maxIndex + j < textLen;
      bool st = true;
      if (inb) {
        unsigned char a = ix.text[o1 + j], b = ix.text[o2 + j];
        st = (a != b) || a == '$' || j >= stopAt;
      }
      stopv[l] = st;
    }
    u64 mk = ballot(stopv);
    if (mk) return base + ctz64(mk);
    base += 64;
  }
}
template <int NS> QM_DEV void set_bit(Bits<NS>& b, int p) {
#pragma unroll
  for (int s = 0; s < NS; ++s) b.w[s] |= (s == (p >> 6)) ? (1ULL << (p & 63)) : 0ULL;   // every word written: see Bits::test
}

// characters [pos, pos + 96) of a clean strand's packed image as three words (first character in the top bits of q[0]);
// rem = characters of the strand from pos on.  Words that would lie behind the image read as zero.
template <int NS>
QM_DEV void ext_query(const u64* planes, int pos, int rem, ExtQuery& xq) {
  xq.nq = rem < 0 ? 0 : rem;
  const int j = pos >> 5, sh = 2 * (pos & 31);
  u64 w[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) w[t] = (j + t) < 2 * NS + 2 ? uniform(planes[(j + t) < 2 * NS + 2 ? j + t : 0]) : 0ULL;
#pragma unroll
  for (int t = 0; t < 3; ++t) xq.q[t] = (w[t] << sh) | ((w[t + 1] >> 1) >> (63 - sh));
}

// ------------------------------------------------------------------ stage 3
// SACollector::getSAHits_ (SACollector.hpp:441-677), NIP disabled
template <int NS, int F>
QM_DEV void get_sa_hits(const DevIndex& ix, const ReadBatch& B, Strand<NS>& V, const unsigned char* str, int L,
                        int startPos, bool haveInterval, u32 lb, u32 ub, long long& cov, u32& strandHits,
                        u32& otherHits, IntervalList& out) {
  const int k = ix.k, P = L - k + 1;
  QM_CNT(17, 1);
  int p = startPos;
  bool skip = haveInterval, lastSearch = false;
  int prevMMPEnd = 0;
  int width = 1;      // first window of a pass: a single position; after an MMP jump the walk crosses ~k positions
  while (true) {
    if (!skip) {
      if (p >= P) break;
      typedef typename Strand<NS>::FT FT;
      if (!V.test(FL_K, p)) { probe_window<NS, F>(ix, V, p, width); width = 32; }
      int kend = known_end(V, p);
      if (kend > P) kend = P;
      const int ph = fl_first(V, p, [](FT w, int s) { return ((w >> (FL_E * NS + s)) & (w >> (FL_F * NS + s)) & 1) != 0; });
      int stop = ph < kend ? ph : kend;
      // misses: spotCheck_ of the complement (:667-675)
      otherHits += (u32)fl_count(V, p, stop, [](FT w, int s) { return ((w >> (FL_E * NS + s)) & ~(w >> (FL_F * NS + s)) & (w >> (FL_C * NS + s)) & 1) != 0; });
      if (((F & QM_F_NIP) != 0)) {                     // spotCheck_ entries (vote): V |= E over [p, e)
        const int e = ph < kend ? ph + 1 : stop;
        QM_LANES(l) {
#pragma unroll
          for (int s2 = 0; s2 < NS; ++s2) {
            const int q = 64 * s2 + l;
            if (q >= p && q < e && ((V.fl[l] >> (FL_E * NS + s2)) & 1) != 0) V.fl[l] |= Strand<NS>::bit(FL_V, s2);
          }
        }
      }
      if (ph >= kend) { p = kend; width = 32; continue; }   // nothing in the probed stretch: next window (or the end)
      strandHits += 1;                                 // spotCheck_ on the hit (:545)
      otherHits += V.test(FL_C, ph) ? 1u : 0u;
      p = ph;
      Iv v = V.tab[p];
      lb = uniform(v.lb); ub = uniform(v.ub);
    }
    skip = false;
    lb = lb ? lb - 1 : 0;                              // :553
    int mlen;
    QM_CNT(18, 1); QM_T(4);
    // the strand's characters behind this k-mer, packed like the entries of ix.saext (clean strands only)
    ExtQuery xq; xq.nq = -1;
    // (-s: every MMP but a read's first is capped and answered by the narrower table above: no packed query needed there)
    if (ix.saext && V.clean && (!(F & QM_F_SEL) || p == 0 || !ix.sanext)) ext_query<NS>(V.planes, p + k, L - (p + k), xq);
    ExtStage xs; xs.pf = nullptr; xs.pfcap = 0; xs.done = false;
    if (!(F & QM_F_SEL)) {
      if (out.n == 0) { xs.pf = out.pf; xs.pfcap = out.pfcap; }     // the first interval of the strand: its sainfo records are staged for hits->mappings
      extend_search(ix, lb, ub, k, str + p, L - p, lb, ub, mlen, V.dollar, 0u, 0, &xq, &xs);
    } else {
      // chain scoring (SACollector.hpp:557-575): only the MMP that starts the read may run to its end, every other
      // one is cut at k + maxMMPExtension characters; a first MMP longer than that (and shorter than the read) is redone cut
      const int cut = p + k + B.max_mmp_ext < L ? p + k + B.max_mmp_ext : L;
      const bool firstAttempt = p == 0;
      const u32 lbP = lb, ubP = ub;
      // the characters a capped extension may use, packed like the entries of ix.sanext (clean strands only: the image
      // holds nothing but A C G T there)
      const int nqc = cut - p - k;
      u32 qn = 0; int nq = 0;
      if (V.lazy && nqc >= 1 && nqc <= QM_NEXT_BASES) { qn = (u32)clean_kmer(V.planes, p + k, nqc); nq = nqc; }
      ExtQuery xc = xq; if (xc.nq >= 0 && !firstAttempt) xc.nq = cut - p - k;      // a capped extension compares fewer characters
      extend_search(ix, lb, ub, k, str + p, (firstAttempt ? L : cut) - p, lb, ub, mlen, V.dollar, firstAttempt ? 0u : qn, firstAttempt ? 0 : nq, &xc);
      if (firstAttempt && !(mlen >= L) && mlen >= k + B.max_mmp_ext)
        { ExtQuery x2 = xq; if (x2.nq >= 0) x2.nq = cut - p - k; extend_search(ix, lbP, ubP, k, str + p, cut - p, lb, ub, mlen, V.dollar, qn, nq, &x2); }
    }
    QM_T(3);
    const bool more = !lastSearch && p + mlen < L;     // the walk continues at kp after this MMP
    const int kp = p + mlen - (k - 1);
    if (ub > lb && ub - lb < (u32)B.max_interval) {     // :577-618
      if (!(F & QM_F_SEL) && out.n == 0 && ub - lb <= (u32)out.pfcap && !xs.done) {
        // three reads in four end with exactly one interval per strand, whose (tid, pos) entries hits->mappings needs next:
        // ask for them now, straight into LDS, so that the trip to sainfo runs under the rest of the walk
        QM_LANES(l) {
          if (l < (int)(ub - lb)) { const u32* g = (const u32*)&ix.sainfo[lb + l]; lds_dma_u32(g, out.pf, l); lds_dma_u32(g + 1, out.pf + out.pfcap, l); }
        }
      }
      out.push(lb, ub, (u32)mlen, (u32)p);
      int corr = prevMMPEnd > p ? prevMMPEnd - p : 0;
      cov += mlen - corr;
      prevMMPEnd = p + mlen;
      if (p + mlen < L) {
        if (all_acgt(V, kp, k)) {
          if (!V.test(FL_K, kp)) {
            // -s: an MMP that was cut at k + maxMMPExtension is followed by the next one maxMMPExtension + 1 positions on, and
            // so on while the read matches: look up those positions only (one in eight by default: 8 sectors instead of 64)
            int stride = 1;
            if ((F & QM_F_SEL) && mlen == k + B.max_mmp_ext) { const int st = B.max_mmp_ext + 1; if ((st & (st - 1)) == 0 && st <= 32) stride = st; }
            probe_window<NS, F>(ix, V, kp, (more && !(F & QM_F_NIP)) ? 32 : 1, stride);
          }
          const typename Strand<NS>::FT wk = V.word_at(kp);
          strandHits += Strand<NS>::flag(wk, FL_F, kp) ? 1u : 0u; otherHits += Strand<NS>::flag(wk, FL_C, kp) ? 1u : 0u;
          if (((F & QM_F_NIP) != 0)) V.or_at(kp, Strand<NS>::bit(FL_V, 0));
        }
      }
    }
    if (lastSearch) return;
    if (p + mlen >= L) return;
    if (!(F & QM_F_NIP)) {
      p = kp;                                           // NIP off: lce == matchedLen (:635-647)
      width = 32;
    } else {                                            // NIP: jump by the LCE of the interval's ends (:634-657)
      int lceLen = lce_wave(ix, lb, ub - 1, mlen, L - (p + mlen));
      int skipLCE = p + lceLen - (k - 1);
      int np = kp > skipLCE ? kp : skipLCE;
      if (lceLen > mlen && L > k) np = np < L - k ? np : L - k;
      p = np;
      width = 1;
    }
    if (p + k == L) lastSearch = true;
  }
}

// rs = reverseRead(fs) (src/RapMapUtils.cpp:63-72,107-128; the table is case-blind, so the upper-cased string serves),
// four characters per lane
template <int NS>
QM_DEV void reverse_string(const unsigned char* fs, unsigned char* rs, int len) {
#pragma unroll
  for (int c = 0; c < (NS + 3) / 4; ++c) {
    QM_LANES(l) {
      const int base = 256 * c + 4 * l;
      if (base < len) {
        const int nb = len - base < 4 ? len - base : 4;
        const u32 rc = rc_char4(((const u32*)fs)[base >> 2]);          // byte j = complement of character base + j
        rs[len - 1 - base] = (unsigned char)rc;
        if (nb > 1) rs[len - 2 - base] = (unsigned char)(rc >> 8);
        if (nb > 2) rs[len - 3 - base] = (unsigned char)(rc >> 16);
        if (nb > 3) rs[len - 4 - base] = (unsigned char)(rc >> 24);
      }
    }
  }
  wave_fence();
}

// SACollector::operator() (SACollector.hpp:108-362), disableNIP_ == true.
// M.str[0] = read (upper-cased); M.str[1] receives reverseRead(read) when that strand is walked.  Returns foundHit.
template <int NS, int F>
QM_DEV bool collect_read(const DevIndex& ix, const ReadBatch& B, WaveMem<NS>& M, int L, bool hasDollar,
                         IntervalList& fwdInts, IntervalList& rcInts) {
  const int k = ix.k, P = L - k + 1;
  const unsigned char* fwdStr = M.str[0];
  unsigned char* rcStr = M.str[1];
  fwdInts.n = 0; rcInts.n = 0;
  fwdInts.pf = nullptr; fwdInts.pfcap = 0; rcInts.pf = nullptr; rcInts.pfcap = 0;
  if (P <= 0) return false;
  // the interval tables have one slot per read position: the slots past the last k-mer are free for the sainfo staging
  fwdInts.pf = (u32*)(M.tab[0] + P); rcInts.pf = (u32*)(M.tab[1] + P); fwdInts.pfcap = rcInts.pfcap = 64 * NS - P;
  Strand<NS> S;
  setup_strand<NS>(ix, fwdStr, L, S, &M.planes[0][0][0], M.tab[0]);
  S.dollar = hasDollar;
  // first-hit scan (:167-237): first E2 position whose k-mer or reverse complement is in the hash
  typedef typename Strand<NS>::FT FT;
  auto isE2 = [](FT w, int s) { return ((w >> (FL_E2 * NS + s)) & 1) != 0; };
  int p0 = fl_first(S, 0, isE2);
  int width = 1;
  bool found = false;
  bool seedR = false;                                  // the rc strand's first k-mer was looked up with the first probe
  while (p0 < P) {
    if (!S.test(FL_K, p0)) {
      if (width == 1) seedR = probe_first<NS, F>(ix, S, p0, &M.tab[1][0]);
      else probe_window<NS, F>(ix, S, p0, width);
      width = 32;
    }
    int kend = known_end(S, p0);
    if (kend > P) kend = P;
    int ph = fl_first(S, p0, [](FT w, int s) { return ((w >> (FL_E2 * NS + s)) & ((w >> (FL_F * NS + s)) | (w >> (FL_C * NS + s))) & 1) != 0; });
    if (ph < kend) { p0 = ph; found = true; break; }
    p0 = fl_first(S, kend, isE2);
  }
  if (!found) return false;
  const FT w0 = S.word_at(p0);
  u32 fwdHit = Strand<NS>::flag(w0, FL_F, p0) ? 1u : 0u;
  u32 rcHit = Strand<NS>::flag(w0, FL_C, p0) ? 1u : 0u;
  // what the first probe learned about the last k-mer, as seen from the reverse-complemented read (only when the
  // window is pure ACGT: reverseRead() and the 2-bit reverse complement agree there)
  seedR = seedR && all_acgt(S, P - 1, k);
  const FT wl = S.word_at(P - 1);
  const bool seedRF = seedR && Strand<NS>::flag(wl, FL_C, P - 1), seedRC = seedR && Strand<NS>::flag(wl, FL_F, P - 1);
  long long fwdCov = 0, rcCov = 0;
  const bool useCoverageCheck = ((F & QM_F_NIP) == 0) && B.strict_check != 0;   // disableNIP_ && strictCheck_ (:138)
  const bool vote = !useCoverageCheck && B.strict_check != 0;
  if (vote) S.or_at(p0, Strand<NS>::bit(FL_V, 0));     // the scan's own KmerDirScore entry (:206-225)

  bool didCheckFwd = false;
  if (fwdHit) {                                         // :247-254
    didCheckFwd = true;
    Iv v = S.tab[p0];
    get_sa_hits<NS, F>(ix, B, S, fwdStr, L, p0, true, uniform(v.lb), uniform(v.ub), fwdCov, fwdHit, rcHit, fwdInts);
  }
  bool checkRC = useCoverageCheck ? (rcHit > 0) : (rcHit >= fwdHit);
  const bool fwdFirst = didCheckFwd;
  Strand<NS> R;
  bool haveR = false;
  if (checkRC) {                                        // :258-265
    // the reverse-complemented read is treated as a string of its own (also correct for IUPAC / 'U'
    // characters, where reverseRead() is not the mirror image of the 2-bit encoding)
    LV<Iv> seed0;                                       // the first probe left this strand's first interval in tab[1][0]
    QM_LANES(l) { if (l == 0) seed0[l] = M.tab[1][0]; }
    wave_fence();
    reverse_string<NS>(fwdStr, rcStr, L);
    setup_strand<NS>(ix, rcStr, L, R, &M.planes[1][0][0], M.tab[1]);
    R.dollar = false;                                   // reverseRead() maps '$' to 'N'
    if (seedR) {
      R.or_at(0, Strand<NS>::bit(FL_K, 0) | (seedRF ? Strand<NS>::bit(FL_F, 0) : (FT)0) | (seedRC ? Strand<NS>::bit(FL_C, 0) : (FT)0));
      QM_LANES(l) { if (l == 0) M.tab[1][0] = seed0[l]; }
      wave_fence();
    }
    haveR = true;
    get_sa_hits<NS, F>(ix, B, R, rcStr, L, 0, false, 0, 0, rcCov, rcHit, fwdHit, rcInts);
  }
  bool checkFwd = useCoverageCheck ? (fwdHit > 0) : (fwdHit >= rcHit);
  if (!didCheckFwd && checkFwd) {                       // :271-278
    if (!(F & QM_F_NIP)) {
      // rare (forward k-mers first seen while walking the reverse complement): rebuild the forward strand's
      // masks instead of keeping them in registers across the whole reverse-complement pass
      Strand<NS> S2;
      setup_strand<NS>(ix, fwdStr, L, S2, &M.planes[0][0][0], M.tab[0]);
      S2.dollar = hasDollar;
      get_sa_hits<NS, F>(ix, B, S2, fwdStr, L, 0, false, 0, 0, fwdCov, fwdHit, rcHit, fwdInts);
    } else {
      get_sa_hits<NS, F>(ix, B, S, fwdStr, L, 0, false, 0, 0, fwdCov, fwdHit, rcHit, fwdInts);
    }
  }
  if (useCoverageCheck) {                               // :283-288 (strictCheckSlack_: 1 with chain scoring, else 0)
    const long long slack = (F & QM_F_SEL) ? 1 : 0;
    if (fwdCov > rcCov + slack) rcInts.n = 0;
    else if (rcCov > fwdCov + slack) fwdInts.n = 0;
  } else if (vote) {                                    // :289-337: the k-mer "spot check" vote
    if (fwdHit > 0 && rcHit == 0) rcInts.n = 0;
    else if (rcHit > 0 && fwdHit == 0) fwdInts.n = 0;
    else {
      // one entry per forward-strand position (std::sort + std::unique on kpos); an rc-strand entry at q
      // describes forward position P-1-q with (fwd,rc) status = (C_r[q], F_r[q]) (:417-429).  When both
      // strands visited a position the earlier entry survives: the seeded forward pass precedes the rc pass,
      // the forward pass from 0 follows it (the scan's own entry always comes first).
      int fwdScore = 0, rcScore = 0;
      Bits<NS> VRm, FRm, CRm;                           // rc-strand masks mirrored to forward positions
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) { VRm.w[s2] = 0; FRm.w[s2] = 0; CRm.w[s2] = 0; }
      if (haveR) { VRm = mirror(fl_bits(R, FL_V), P); FRm = mirror(fl_bits(R, FL_C), P); CRm = mirror(fl_bits(R, FL_F), P); }
      const Bits<NS> SV = fl_bits(S, FL_V), SF = fl_bits(S, FL_F), SC = fl_bits(S, FL_C);
      Bits<NS> first;                                   // forward entries that precede the rc pass
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) first.w[s2] = fwdFirst ? SV.w[s2] : 0;
      set_bit(first, p0);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        u64 a1 = first.w[s2];                           // forward entries, first in time
        u64 a2 = VRm.w[s2] & ~a1;                       // rc entries
        u64 a3 = SV.w[s2] & ~a1 & ~a2;                  // forward entries of the late pass
        u64 ff = (SF.w[s2] & (a1 | a3)) | (FRm.w[s2] & a2);
        u64 cc = (SC.w[s2] & (a1 | a3)) | (CRm.w[s2] & a2);
        u64 all = a1 | a2 | a3;
        fwdScore += 2 * popc64(ff & all) - popc64(all);
        rcScore += 2 * popc64(cc & all) - popc64(all);
      }
      if (fwdScore > rcScore) rcInts.n = 0;
      else if (rcScore > fwdScore) fwdInts.n = 0;
    }
  }
  if (B.quasi_cov > 0.0 && !(F & QM_F_NIP)) {               // :343-358 (only with NIP disabled)
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
  QM_CNT(10, 1); QM_CNT(11, n);
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
QM_DEV int single_interval(const DevIndex& ix, const Bufs& bf, int rOff, u32 lb, u32 ub, u32 qpos, bool isRC, const u32* pf, int pfcap) {
  int n = (int)(ub - lb);
  QM_CNT(12, 1); QM_CNT(13, n);
  if (pf && n <= pfcap) {                                // the only interval of the list is the first one recorded: staged by get_sa_hits
    lds_dma_wait();
    QM_LANES(l) {
      if (l < n) {
        const int hitPos = (int)(pf[pfcap + l] - qpos);
        bf.A[l] = ((u64)pf[l] << 32) | ((u32)hitPos ^ 0x80000000u);
      }
    }
  } else
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

// collectFromSingleInterval for an interval of at most 64 suffixes, in registers: lane l holds suffix l's (tid, position);
// a lane's entry survives when no other entry of its transcript sorts before it (position, then input order -- the first
// element of the transcript's run in the stable sort of :761-767), and goes to the slot given by the number of surviving
// entries with a smaller transcript id.  No sort buffers, no trips through LDS but the one that delivers the list.
QM_DEV int single_interval_small(const DevIndex& ix, u64* R, int rOff, u32 lb, u32 ub, u32 qpos, bool isRC, const u32* pf, int pfcap) {
  const int n = (int)(ub - lb);
  QM_CNT(12, 1); QM_CNT(13, n);
  LV<u64> e;
  if (pf && n <= pfcap) {                                // staged by get_sa_hits while the walk went on
    lds_dma_wait();
    QM_LANES(l) { e[l] = l < n ? (((u64)pf[l] << 32) | ((u32)(pf[pfcap + l] - qpos) ^ 0x80000000u)) : ~0ULL; }
  } else {
    QM_LANES(l) {
      u64 v = ~0ULL;
      if (l < n) { const SaInfo si = ix.sainfo[lb + l]; v = ((u64)si.tid << 32) | (((u32)si.pos - qpos) ^ 0x80000000u); }
      e[l] = v;
    }
  }
  LV<bool> keep;
  QM_LANES(l) { keep[l] = l < n; }
  for (int j = 0; j < n; ++j) {
    const u64 kj = read_lane(e, j);
    QM_LANES(l) { if ((u32)(kj >> 32) == (u32)(e[l] >> 32) && (kj < e[l] || (kj == e[l] && j < l))) keep[l] = false; }
  }
  const u64 km = ballot(keep);
  LV<int> slot;
  QM_LANES(l) { slot[l] = 0; }
  for (u64 r = km; r; r &= r - 1) {
    const u32 tj = (u32)(read_lane(e, ctz64(r)) >> 32);
    QM_LANES(l) { slot[l] += tj < (u32)(e[l] >> 32) ? 1 : 0; }
  }
  QM_LANES(l) { if (keep[l]) R[rOff + slot[l]] = mk_elem((u32)(e[l] >> 32), isRC, (int)((u32)e[l] ^ 0x80000000u)); }
  wave_fence();
  return popc64(km);
}

// intersectSAHits + collectHitsSimpleSA (HitManager.cpp:587-689, :449-493, :308-322),
// consensusFraction == 1 (maxSlack 0), strictFilter off.
QM_DEV int multi_interval(const DevIndex& ix, const Bufs& bf, int rOff, const IntervalList& ints, bool isRC) {
  const int m = ints.n;
  QM_CNT(14, 1); QM_CNT(15, m);
  int minIdx = 0, minSpan = 0x7fffffff;
  for (int i = 0; i < m; ++i) {                       // first smallest span (:636-641)
    u32 lb, ub, ln, qp; ints.get(i, lb, ub, ln, qp);
    if ((int)(ub - lb) < minSpan) { minSpan = (int)(ub - lb); minIdx = i; }
  }
  u32 lb0, ub0, ln0, q0; ints.get(minIdx, lb0, ub0, ln0, q0);
  int n0 = (int)(ub0 - lb0);
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
    u32 lb, ub, ln, qp; ints.get(ii, lb, ub, ln, qp);
    int n = (int)(ub - lb);
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
      u32 lb, ub, ln, qp; ints.get(ii, lb, ub, ln, qp);
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
// keepBoth (--fuzzyIntersection): a transcript hit in both orientations keeps both entries, fwd first --
// the second one is the survivor's oppositeStrandPositions (mergeOrientationUnique, :846-866).
QM_DEV int hits_to_mappings(const DevIndex& ix, const Bufs& bf, const IntervalList& fwdInts,
                            const IntervalList& rcInts, bool keepBoth) {
  int nf = 0, nr = 0;
  if (fwdInts.n > 1) nf = multi_interval(ix, bf, 0, fwdInts, false);
  else if (fwdInts.n == 1) {
    u32 lb, ub, ln, qp; fwdInts.get(0, lb, ub, ln, qp);
    nf = ub - lb <= 64u ? single_interval_small(ix, bf.R, 0, lb, ub, qp, false, fwdInts.pf, fwdInts.pfcap)
                       : single_interval(ix, bf, 0, lb, ub, qp, false, fwdInts.pf, fwdInts.pfcap);
  }
  if (rcInts.n > 1) nr = multi_interval(ix, bf, nf, rcInts, true);
  else if (rcInts.n == 1) {
    u32 lb, ub, ln, qp; rcInts.get(0, lb, ub, ln, qp);
    nr = ub - lb <= 64u ? single_interval_small(ix, bf.R, nf, lb, ub, qp, true, rcInts.pf, rcInts.pfcap)
                       : single_interval(ix, bf, nf, lb, ub, qp, true, rcInts.pf, rcInts.pfcap);
  }
  if (nf > 0 && nr > 0) {
    // stable merge by tid, fwd first on ties, duplicates collapse to the first (:834-881)
    int n = nf + nr;
    for (int base = 0; base < n; base += 64) { QM_LANES(l) { int i = base + l; if (i < n) bf.A[i] = bf.R[i]; } }
    rank_sort(bf.A, bf.B, n);
    if (keepBoth) {
      for (int base = 0; base < n; base += 64) { QM_LANES(l) { int i = base + l; if (i < n) bf.R[i] = bf.B[i]; } }
      wave_fence();
      return n;
    }
    return unique_emit(bf.B, n, 33, bf.R, 0, [](u64 v) { return v; });
  }
  return nf + nr;
}

// largest list any stage will hold for this read (decides LDS vs global scratch)
QM_DEV int list_bound(const IntervalList& a, const IntervalList& b) {
  int tot = 0;
  const IntervalList* ls[2] = {&a, &b};
  for (int t = 0; t < 2; ++t) {
    int m = ls[t]->n, mn = 0x7fffffff;
    for (int i = 0; i < m; ++i) { u32 lb, ub, ln, qp; ls[t]->get(i, lb, ub, ln, qp); if ((int)(ub - lb) < mn) mn = (int)(ub - lb); }
    if (m > 0) tot += mn;
  }
  return tot;
}

// the read's SA-interval hits to B.iv_out.  Like the hit lists they go through a chunked bump allocator: a returning atomic on
// one word per read would cap the kernel at ~88 M reads/s (and did: the first pass of -s took 246 ms per 20 M reads with it).
#define QM_IVCHUNK 2048
struct WaveAlloc;
QM_DEV void dump_intervals(const ReadBatch& B, long long read, int mate, const IntervalList& F, const IntervalList& R, long long& ivBase, int& ivUsed) {
  const int n = F.n + R.n;
  long long base = 0;
  if (n > QM_IVCHUNK) {                                   // cannot happen below 1024 positions per strand; kept for safety
    LV<u64> bv;
    QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor + QM_SC_IVCUR, (u64)n); }
    base = (long long)read_lane(bv, 0);
  } else if (n > 0) {
    if (ivBase < 0 || ivUsed + n > QM_IVCHUNK) {
      LV<u64> bv;
      QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor + QM_SC_IVCUR, (u64)QM_IVCHUNK); }
      ivBase = (long long)read_lane(bv, 0); ivUsed = 0;
    }
    base = ivBase + ivUsed; ivUsed += n;
  }
  const bool fits = base + n <= B.iv_cap;
  if (!fits) { QM_LANES(l) { if (l == 0) *B.status |= 16; } }
  QM_LANES(l) { if (l == 0) { B.iv_cnt[read] = fits ? (u32)n : 0u; B.iv_off[read] = base; } }
  if (!fits) return;
  // one lane per record (forward strand's first): the lists were lane 0's business while they grew, writing them out is not
  for (int i0 = 0; i0 < n; i0 += 64) {
    QM_LANES(l) {
      const int i = i0 + l;
      if (i < n) {
        const int t = i >= F.n ? 1 : 0, j = t ? i - F.n : i;
        IntRec r;
        // a branch per home of a record (LDS / the wave's global scratch), as in get()
        if (j < QM_ICAP) { if (t) { r.b = R.lds[j].b; r.e = R.lds[j].e; r.len = R.lds[j].len; r.q = R.lds[j].q; } else { r.b = F.lds[j].b; r.e = F.lds[j].e; r.len = F.lds[j].len; r.q = F.lds[j].q; } }
        else r = t ? R.ovf[j - QM_ICAP] : F.ovf[j - QM_ICAP];
        qm_sa_interval_hit h; h.begin = (int)r.b; h.end = (int)r.e; h.len = r.len; h.query_pos = r.q;
        h.query_rc = (uint8_t)t; h.list = (uint8_t)(2 * mate + t); h.pad = 0;
        B.iv_out[base + i] = h;
      }
    }
  }
}

struct SelScratch;
struct SelScratchDyn;
template <int CAP, int OUTCAP> struct SelScratchT;
QM_DEV int sel_hits_to_mappings(const DevIndex& ix, const ReadBatch& B, const IntervalList& fwdInts, const IntervalList& rcInts,
                                u32 readLen, int mate, SelScratch& G, struct SelScratchLds* L, u64* ldsOut, const u64*& src, SelScratchDyn* dyn);

// ------------------------------------------------------------------ stage A driver
// One read: load -> collect -> hits->mappings -> list to global memory.
struct WaveAlloc { long long base; int used; long long ivBase; int ivUsed; };   // the wave's current chunks of B.lists / B.iv_out (wave-uniform)

// Software pipeline of the persistent loop: the offsets of the read after next and the characters of the next read are
// requested while the current read is processed, so the two dependent round trips that start a read (offsets ->
// characters) overlap with it.  Both requests are LDS-direct loads (global_load_lds_dword: the data goes from memory
// straight into the wave's staging rows, no destination register), because a prefetch held in registers does not survive
// this compiler: scalar loads have to land in SGPRs that stay live across the whole read (it waited for them on the spot
// and spilled them), vector loads are waited for wherever their registers are next written or packed, and any wait that
// follows the previous read's list stores also waits for those stores (gfx9 counts loads and stores in one vmcnt).  With
// LDS staging the only wait is the one finish_read places before its stores, by which time the loads have long landed.
//   stage[]    raw bytes of the next read, fetched as aligned dwords from (src + o0) & ~3 on
//   ostage[j]  two offsets (4 dwords) of a read: slot parity j holds the current read's, j ^ 1 the next one's
// slot of a launch -> read: the identity, except in the launches that walk a queue of reads an earlier pass set aside (the slow
// pass of -s, the long-read pass, the reads qm_lean_kernel leaves to qm_read_kernel)
template <int F, int NS = 0>
QM_DEV long long read_id(const ReadBatch& B, long long slot) {
  return B.slowq ? uniform(B.slowq[slot]) : slot;
}
QM_DEV void read_src(const ReadBatch& B, long long read, const unsigned char*& src, const long long*& off, long long& unit) {
  // seq1, off1, seq2, off2 sit next to each other in ReadBatch: mate m's pair of pointers is one indexed scalar load
  const int paired = B.seq2 != nullptr ? 1 : 0;
  const int mate = (int)(read & paired);
  unit = read >> paired;
  src = (&B.seq1)[2 * mate];
  off = (const long long*)(&B.seq1)[2 * mate + 1];
}
// request the two offsets of the read in `slot` into ostage[par]
template <int NS, int F>
QM_DEV void stage_offsets(const ReadBatch& B, long long slot, WaveMem<NS>& M, int par) {
  if (slot >= B.nreads) return;
  const unsigned char* src; const long long* off; long long unit;
  read_src(B, read_id<F, NS>(B, slot), src, off, unit);
  QM_LANES(l) { if (l < 4) lds_dma_u32((const u32*)(off + unit) + l, M.ostage[par], l); }
}
QM_DEV long long staged_offset(const u32* o, int j) { return (long long)(((u64)uniform(o[2 * j + 1]) << 32) | (u64)uniform(o[2 * j])); }
// turn the offsets in ostage[par] (which belong to the read in `slot`, and have landed) into the request for its characters
template <int NS, int F>
QM_DEV void stage_chars(const ReadBatch& B, long long slot, WaveMem<NS>& M, int par) {
  if (slot >= B.nreads) return;
  const unsigned char* src; const long long* off; long long unit;
  read_src(B, read_id<F, NS>(B, slot), src, off, unit);
  const long long o0 = staged_offset(M.ostage[par], 0), o1 = staged_offset(M.ostage[par], 1);
  int len = (int)(o1 - o0);
  if (len > 64 * NS) len = 64 * NS;
  const unsigned char* p = src + o0;
  const int mis = (int)((unsigned long long)p & 3ULL);
  const u32* g = (const u32*)(p - mis);                                  // aligned: never reads past the word holding the last character
  const int nd = (mis + len + 3) >> 2;
#pragma unroll
  for (int c = 0; c < (16 * NS + 1 + 63) / 64; ++c) {
    if (64 * c < nd) { QM_LANES(l) { if (64 * c + l < nd) lds_dma_u32(g + 64 * c + l, M.stage + 64 * c, l); } }
  }
}

// Second half of a read: SA-interval hits -> the read's hit list in B.lists (hitsToMappingsSimple), shared by the fused
// kernel and by the stage entry that starts from caller-supplied intervals (qm_h2m_kernel).
template <int NS, int F>
QM_DEV void finish_read(const DevIndex& ix, const ReadBatch& B, long long read, int len, int mate, bool foundHit, u64 (*buf)[QM_CAP], u64* gscr,
                        WaveAlloc& wa, IntervalList& fi, IntervalList& ri, SelScratch* ss, struct SelScratchLds* sl, SelScratchDyn* dyn) {
  int bound = list_bound(fi, ri);
  int n = 0;
  const u64* listSrc = nullptr;
  if (F & QM_F_SEL) {                  // -s: chaining + multi-position groups (qm_sel.inl)
    n = sel_hits_to_mappings(ix, B, fi, ri, (u32)len, mate, *ss, sl, &buf[0][0], listSrc, dyn);
    if (n == -2) {                     // waits on the slow queue: no list yet
      QM_LANES(l) { if (l == 0) { B.lcnt[read] = QM_LCNT_SLOW; B.loff[read] = 0; } }
      return;
    }
  } else if (bound > QM_GCAP) {        // only reachable with max_interval > 1000: this read goes without hits, and says so
    skip_read(B, read, 2);
    // (include/qmap_mi355.h: such a read has no intervals and foundHit false, like a read the long-read pass skips)
    QM_LANES(l) { if (l == 0) { if (B.iv_out) { B.iv_cnt[read] = 0; B.iv_off[read] = 0; } if (B.found_out) B.found_out[read] = 0; } }
  } else if (bound <= QM_CAP) {
    // the two homes of the sort buffers are two expansions of the routine: behind one set of pointers that may be LDS or
    // global every access is a FLAT instruction -- through the vector-memory path even when it lands in LDS (it was a third
    // of this phase's time)
    Bufs bf; bf.A = buf[0]; bf.B = buf[1]; bf.R = buf[2];
    n = hits_to_mappings(ix, bf, fi, ri, B.fuzzy != 0);
    listSrc = bf.R;
  } else {
    Bufs bf; bf.A = gscr; bf.B = gscr + QM_GCAP; bf.R = gscr + 2 * QM_GCAP;
    n = hits_to_mappings(ix, bf, fi, ri, B.fuzzy != 0);
    listSrc = bf.R;
  }
  QM_T(5);
  // Everything this wave has in flight -- the staged prefetch of the next read -- lands before the first store: a later
  // wait could not tell the loads from the stores (one counter), and the next iteration reads the staging rows without one.
  lds_dma_wait();
  // hand the list to stage B.  One returning atomic on a single word saturates at ~88 M/s on this chip
  // (MI355X_MICROARCH.md "dequeue"), far below the read rate, so a wave reserves QM_CHUNK elements at
  // a time and sub-allocates from its chunk.
  long long base = 0;
  if (n > QM_CHUNK) {                  // only the slow pass of -s makes lists this long: an allocation of its own
    LV<u64> bv;
    QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor, (u64)n); }
    base = (long long)read_lane(bv, 0);
    if (base + n > B.lists_cap) { QM_LANES(l) { if (l == 0) *B.status |= 1; } n = 0; base = 0; }
  } else if (n > 0) {
    if (wa.base < 0 || wa.used + n > QM_CHUNK) {
      LV<u64> bv;
      QM_LANES(l) { bv[l] = 0; if (l == 0) bv[l] = atomic_add_u64(B.cursor, (u64)QM_CHUNK); }
      wa.base = (long long)read_lane(bv, 0); wa.used = 0;
    }
    base = wa.base + wa.used;
    if (base + n > B.lists_cap) { QM_LANES(l) { if (l == 0) *B.status |= 1; } n = 0; base = 0; }
    else wa.used += n;
  }
  if (!(F & QM_F_SEL) && bound <= QM_CAP) {          // the usual case, spelled out so that the source is addressed as LDS (not flat)
    const u64* src = buf[2];
    for (int b0 = 0; b0 < n; b0 += 64) { QM_LANES(l) { int i = b0 + l; if (i < n) B.lists[base + i] = src[i]; } }
  } else if ((F & QM_F_SEL) && listSrc == (const u64*)&buf[0][0]) {   // -s, the LDS edition of the scratch: likewise
    const u64* src = &buf[0][0];
    for (int b0 = 0; b0 < n; b0 += 64) { QM_LANES(l) { int i = b0 + l; if (i < n) B.lists[base + i] = src[i]; } }
  } else {
    for (int b0 = 0; b0 < n; b0 += 64) { QM_LANES(l) { int i = b0 + l; if (i < n) B.lists[base + i] = listSrc[i]; } }
  }
  const u32 flag = ((B.fuzzy || (F & QM_F_SEL)) && foundHit) ? 0x80000000u : 0u;      // lh / rh of RapMapSAMapper.cpp:472-478
  QM_LANES(l) { if (l == 0) { B.lcnt[read] = (u32)n | flag; B.loff[read] = base; } }
  QM_T(6);
}


// slot: position in the launch (the pipeline prefetches slot + nw and slot + 2 nw); par: parity of the iteration
template <int NS, int F>
QM_DEV void map_read(const DevIndex& ix, const ReadBatch& B, long long read, long long slot, long long nw, int par, WaveMem<NS>& M, u64* gscr, WaveAlloc& wa,
                     SelScratch* ss = nullptr, struct SelScratchLds* sl = nullptr, SelScratchDyn* dyn = nullptr) {
  const bool paired = B.seq2 != nullptr;
  const int mate = paired ? (int)(read & 1) : 0;
  // this read's offsets and raw characters were staged while the previous read was mapped
  const long long o0 = staged_offset(M.ostage[par], 0), o1 = staged_offset(M.ostage[par], 1);
  const int rawLen = (int)(o1 - o0);
  const bool tooLong = rawLen > 64 * NS;
  // A read that does not fit this kernel's slots is set aside for the long-read pass (a second, small launch of the NS = 32
  // kernels over the queue of such reads; the host sizes nothing from it but checks the longest against QM_MAX_LONG_READ_LEN).
  // (-s: the set-aside reads get their intervals from the 32-slot chain-scoring collector before the list kernel runs.)
  // A read beyond the 32-slot kernels too (QM_MAX_LONG_READ_LEN characters) is not mapped: empty result, an entry in the batch's
  // list of skipped reads (round 5; before, it failed the whole batch).
  const bool setAside = tooLong && NS < 32, skipLong = tooLong && NS >= 32;
  if (tooLong) {
    QM_LANES(l) {
      if (l == 0) {
        if (setAside) { B.lcnt[read] = QM_LCNT_SLOW; B.loff[read] = 0; atomic_add_u64(B.cursor + QM_SC_SLOWCNT, 1ULL); atomic_max_u64(B.cursor + QM_SC_SLOWMAX, (u64)rawLen); }
        else { B.lcnt[read] = 0; B.loff[read] = 0; }
      }
    }
    if (skipLong) skip_read(B, read, 1);
  }
  // uniform(): the length must stay in an SGPR -- merged into the lane-0 branch above it became a per-lane value and
  // with it every position, mask and branch of the collector moved from the scalar unit to the VALU
  const int len = uniform(tooLong ? 64 * NS : rawLen);
  int mis;
  { const unsigned char* src; const long long* off; long long unit; read_src(B, read, src, off, unit); mis = (int)((unsigned long long)(src + o0) & 3ULL); }
  unsigned char* fs = M.str[0];
  QM_T(6);
  // four characters per lane: the upper-cased read goes to fs as whole words (every consumer applies ::toupper anyway,
  // SASearcher.hpp:111,155); reverseRead() of it is only written when the collector turns to that strand (reverse_string)
  LV<bool> dl;
  QM_LANES(l) { dl[l] = false; }
#pragma unroll
  for (int c = 0; c < (NS + 3) / 4; ++c) {
    QM_LANES(l) {
      const int base = 256 * c + 4 * l;
      if (base < len) {
        const u32 w0 = M.stage[64 * c + l], w1 = M.stage[64 * c + l + 1];   // (the row has a spare word behind the last one read)
        const u32 d = align_bytes(w1, w0, mis);                            // characters base .. base + 3
        const int nb = len - base < 4 ? len - base : 4;
        const u32 lenmask = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
        ((u32*)fs)[base >> 2] = upcase4(d);
        dl[l] = dl[l] || (eq_bytes(d, (u32)'$') & lenmask) != 0;
      }
    }
  }
  const bool hasDollar = ballot(dl) != 0;
  wave_fence();
  // the staging rows are free again: next read's characters, and the offsets of the one after it
  stage_chars<NS, F>(B, slot + nw, M, par ^ 1);
  stage_offsets<NS, F>(B, slot + 2 * nw, M, par);
  QM_T(0);
  if (setAside || skipLong) {                             // mapped by the long-read pass / not at all (the next iteration reads the staging rows: what was just requested must have landed)
    lds_dma_wait();
    if (skipLong) { QM_LANES(l) { if (l == 0) { if (B.iv_out) { B.iv_cnt[read] = 0; B.iv_off[read] = 0; } if (B.found_out) B.found_out[read] = 0; } } }
    return;
  }
  IntervalList fi, ri;
  fi.lds = (QM_LDS(IntRec)*)M.ints[0]; ri.lds = (QM_LDS(IntRec)*)M.ints[1];
  fi.ovf = (IntRec*)(gscr + 3 * QM_GCAP); ri.ovf = fi.ovf + QM_IOVF;
  const bool foundHit = collect_read<NS, F>(ix, B, M, len, hasDollar, fi, ri);
  QM_T(4);
  if (B.iv_out || B.found_out || (F & QM_F_COLLECT)) lds_dma_wait();   // the staged prefetch must have landed before any store follows it
  if (B.iv_out) dump_intervals(B, read, mate, fi, ri, wa.ivBase, wa.ivUsed);
  if (B.found_out) { QM_LANES(l) { if (l == 0) B.found_out[read] = foundHit ? 1 : 0; } }
  if (F & QM_F_COLLECT) return;          // stage entry "collector only" (SACollector::operator() as a call of its own)
  finish_read<NS, F>(ix, B, read, len, mate, foundHit, M.buf, gscr, wa, fi, ri, ss, sl, dyn);
}

// ------------------------------------------------------------------ stage B: one thread per unit
struct UnitCounters { u64 pe, se, tot, reads, tooMany, mapped; };

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
QM_DEV qm_hit paired_hit(u64 le, u64 re, u32 l1, u32 l2) {   // RapMapUtils.hpp:1212-1231
  int s1 = el_pos(le) > 0 ? el_pos(le) : 0, s2 = el_pos(re) > 0 ? el_pos(re) : 0;
  bool r1First = s1 < s2;
  int fragStart = r1First ? s1 : s2;
  int fragEnd = r1First ? (int)((u32)s2 + l2) : (int)((u32)s1 + l1);
  qm_hit h; h.tid = el_tid(le); h.pos = s1; h.mate_pos = s2; h.frag_len = (u32)(fragEnd - fragStart);
  h.read_len = l1; h.mate_len = l2; h.fwd = el_rc(le) ? 0 : 1; h.mate_is_fwd = el_rc(re) ? 0 : 1;
  h.is_paired = 1; h.mate_status = 3; h.aln_score = 0;
  return h;
}

// One transcript group of a fuzzy-mode list: the surviving hit plus, when the transcript was hit in
// both orientations, the position on the other strand.  With considerMultiPos off every position
// vector of the reference holds one element (HitManager.cpp:321,736,860).
struct FzGroup { u32 tid; bool hasF, hasR; int f, r; int width; u64 first; };
QM_DEV FzGroup fz_group(const u64* X, int i, int n) {
  FzGroup g; u64 e = X[i];
  g.first = e; g.tid = el_tid(e); g.width = 1; g.hasF = false; g.hasR = false; g.f = 0; g.r = 0;
  if (el_rc(e)) { g.hasR = true; g.r = el_pos(e); }
  else {
    g.hasF = true; g.f = el_pos(e);
    if (i + 1 < n && el_tid(X[i + 1]) == g.tid) { g.hasR = true; g.r = el_pos(X[i + 1]); g.width = 2; }
  }
  return g;
}
// findBestHitFWRC (RapMapUtils.hpp:923-988) for single-element position lists
QM_DEV bool fz_best(bool hasF, int f, bool hasR, int r, int fwdLen, int& gap) {
  if (!hasF || !hasR) return false;
  if (r < f) return false;                                // updateBestGap leaves maxGap
  int d = r - (f + fwdLen);
  gap = d < 0 ? -d : d;
  return true;
}
QM_DEV int fz_groups(const u64* X, int n) { int g = 0; for (int i = 0; i < n; i += fz_group(X, i, n).width) ++g; return g; }

// mergeLeftRightHitsFuzzy (RapMapUtils.hpp:864-1183) + the per-pair driver; same contract as unit_merge.
QM_DEV int unit_merge_fuzzy(const PairBatch& P, long long u, qm_hit* out, int cap, UnitCounters* uc) {
  const int maxHits = P.max_num_hits;
  const u32 c0 = P.lcnt[2 * u], c1 = P.lcnt[2 * u + 1];
  const int nl = (int)(c0 & 0x7fffffffu), nr = (int)(c1 & 0x7fffffffu);
  const bool lh = (c0 >> 31) != 0, rh = (c1 >> 31) != 0;
  const u64* LL = P.lists + P.loff[2 * u];
  const u64* RR = P.lists + P.loff[2 * u + 1];
  const u32 l1 = (u32)(P.off1[u + 1] - P.off1[u]), l2 = (u32)(P.off2[u + 1] - P.off2[u]);
  if (uc) uc->reads += 1;
  int cnt = 0;
  if (nl == 0 || nr == 0) {
    // orphans only when the other end found no seed k-mer at all (:880-901)
    const int t = nl == 0 ? 1 : 0;
    const bool otherMatched = nl == 0 ? lh : rh;
    const u64* X = t == 0 ? LL : RR; const int nx = t == 0 ? nl : nr; const u32 ln = t == 0 ? l1 : l2;
    if (!otherMatched && nx > 0) {
      const int g = fz_groups(X, nx);
      if (uc) { uc->se += (u64)g; uc->pe += (u64)g; }     // jointHits.size() is added to peHits as well (:1176-1179)
      bool keep = P.merge_only || (g <= maxHits && !P.no_orphans);   // RapMapSAMapper.cpp:534-551
      if (keep) {
        for (int i = 0; i < nx;) {
          FzGroup gr = fz_group(X, i, nx); i += gr.width;
          qm_hit h = orphan_hit(gr.first, ln, t == 0 ? 1 : 2);
          if (P.no_dovetail && dovetail(h)) continue;
          if (out && cnt < cap) out[cnt] = h;
          ++cnt;
        }
      }
    }
  } else {
    int i = 0, j = 0, nm = 0, nkeep = 0;
    bool tooMany = false, sameTxp = false;
    while (i < nl && j < nr) {
      FzGroup a = fz_group(LL, i, nl), b = fz_group(RR, j, nr);
      if (a.tid < b.tid) { i += a.width; continue; }
      if (b.tid < a.tid) { j += b.width; continue; }
      sameTxp = true;
      int gFR = 0x7fffffff, gRF = 0x7fffffff;
      const bool fwrc = fz_best(a.hasF, a.f, b.hasR, b.r, (int)l1, gFR);   // left fwd, right rc
      const bool rcfw = fz_best(b.hasF, b.f, a.hasR, a.r, (int)l2, gRF);   // right fwd, left rc
      if (fwrc || rcfw) {
        int leftPos, rightPos; bool leftFwd;
        if (fwrc && !(rcfw && gRF < gFR)) { leftPos = a.f; rightPos = b.r; leftFwd = true; }
        else { leftPos = a.r; rightPos = b.f; leftFwd = false; }
        int s1 = leftPos > 0 ? leftPos : 0, s2 = rightPos > 0 ? rightPos : 0;
        bool r1First = s1 < s2;
        int fragStart = r1First ? s1 : s2;
        int fragEnd = r1First ? (int)((u32)s2 + l2) : (int)((u32)s1 + l1);
        qm_hit h; h.tid = a.tid; h.pos = leftPos; h.mate_pos = rightPos; h.frag_len = (u32)(fragEnd - fragStart);
        h.read_len = l1; h.mate_len = l2; h.fwd = leftFwd ? 1 : 0; h.mate_is_fwd = leftFwd ? 0 : 1;
        h.is_paired = 1; h.mate_status = 3; h.aln_score = 0;
        ++nm;
        if (nm > maxHits) { tooMany = true; break; }      // :1153
        if (!(P.no_dovetail && dovetail(h))) {
          if (out && nkeep < cap) out[nkeep] = h;
          ++nkeep;
        }
      }
      i += a.width; j += b.width;
    }
    if (uc && tooMany) uc->tooMany += 1;
    if (uc && P.too_many) P.too_many[u] = (tooMany ? 1 : 0) | (sameTxp ? 2 : 0);
    if (!tooMany && nm > 0) { if (uc) uc->pe += (u64)nm; cnt = nkeep; }
  }
  if (uc) { uc->tot += (u64)cnt; if (cnt > 0) uc->mapped += 1; }
  return cnt;
}

// mergeLeftRightHits (RapMapUtils.hpp:1185-1264) + per-pair driver (RapMapSAMapper.cpp:461-551,684-701),
// or the single-end driver (:232-250).  out == nullptr: count only (and add to the counters);
// otherwise write the unit's hits to out[0..min(return, cap)), cap = the count pass's result.
// a record qm_duo_kernel left for a pair it merged itself (qm_duo.inl): {left element, right element} of a paired hit, or
// {element, QM_DUO_ORPHAN | MateStatus} of an orphan
QM_DEV qm_hit duo_hit(u64 w0, u64 w1, u32 l1, u32 l2) {
  if ((w1 & QM_DUO_ORPHAN) == QM_DUO_ORPHAN) { const int ms = (int)(w1 & 0xffu); return orphan_hit(w0, ms == 1 ? l1 : l2, ms); }
  return paired_hit(w0, w1, l1, l2);
}
QM_DEV int unit_merge(const PairBatch& P, long long u, qm_hit* out, int cap, UnitCounters* uc) {
  const int maxHits = P.max_num_hits;
  if (P.paired && P.lcnt[2 * u] == QM_LCNT_PAIR) {
    // merged where it was mapped: the count (and the counters) are there already, the write pass expands the records
    const int c = (int)P.cnt[u];
    if (out) {
      const u64* R = P.lists + P.loff[2 * u];
      const u32 l1 = (u32)(P.off1[u + 1] - P.off1[u]), l2 = (u32)(P.off2[u + 1] - P.off2[u]);
      for (int i = 0; i < c && i < cap; ++i) out[i] = duo_hit(R[2 * i], R[2 * i + 1], l1, l2);
    }
    return c;
  }
  if (P.paired && P.fuzzy) return unit_merge_fuzzy(P, u, out, cap, uc);
  if (!P.paired) {
    int n = (int)(P.lcnt[u] & 0x7fffffffu);
    const u64* X = P.lists + P.loff[u];
    u32 len = (u32)(P.off1[u + 1] - P.off1[u]);
    if (uc) { uc->reads += 1; uc->tot += (u64)n; }       // counted before the maxNumHits clear (:240-245)
    if (n > maxHits) n = 0;
    if (uc && n > 0) uc->mapped += 1;
    if (out) for (int i = 0; i < n && i < cap; ++i) out[i] = orphan_hit(X[i], len, 0);
    return n;
  }
  const int nl = (int)P.lcnt[2 * u], nr = (int)P.lcnt[2 * u + 1];
  const u64* LL = P.lists + P.loff[2 * u];
  const u64* RR = P.lists + P.loff[2 * u + 1];
  const u32 l1 = (u32)(P.off1[u + 1] - P.off1[u]), l2 = (u32)(P.off2[u + 1] - P.off2[u]);
  if (uc) uc->reads += 1;
  // two-pointer intersection on transcript id (both lists are sorted and unique)
  int nm = 0, nkeep = 0;
  if (nl > 0 && nr > 0) {
    int i = 0, j = 0;
    while (i < nl && j < nr) {
      u32 a = el_tid(LL[i]), b = el_tid(RR[j]);
      if (a < b) ++i;
      else if (b < a) ++j;
      else {
        ++nm;
        if (!(P.no_dovetail && dovetail(paired_hit(LL[i], RR[j], l1, l2)))) {
          if (out && nkeep < cap) out[nkeep] = paired_hit(LL[i], RR[j], l1, l2);
          ++nkeep;
        }
        ++i; ++j;
      }
    }
  }
  const bool tooMany = nm > maxHits;                    // :1233-1234
  if (uc && tooMany) uc->tooMany += 1;
  if (uc && P.too_many) P.too_many[u] = (tooMany ? 1 : 0) | (nm > 0 ? 2 : 0);
  int cnt = 0;
  if (!tooMany && nm > 0) {
    if (uc) uc->pe += (u64)nm;
    cnt = nkeep;
  } else if (!tooMany && nl + nr > 0) {
    if (uc) uc->se += (u64)(nl + nr);
    bool keep = true;
    if (nl + nr > maxHits && !P.merge_only) keep = false;   // RapMapSAMapper.cpp:534-536
    if (P.no_orphans && !P.merge_only) keep = false;        // :539-551
    if (keep) {
      // --noDovetail on orphans: the reference evaluates the predicate with an uninitialised matePos;
      // we define matePos = 0, mateIsFwd = true (same as the oracle).
      for (int t = 0; t < 2; ++t) {
        const u64* X = t == 0 ? LL : RR; int nx = t == 0 ? nl : nr; u32 ln = t == 0 ? l1 : l2;
        for (int i = 0; i < nx; ++i) {
          qm_hit h = orphan_hit(X[i], ln, t == 0 ? 1 : 2);
          if (P.no_dovetail && dovetail(h)) continue;
          if (out && cnt < cap) out[cnt] = h;
          ++cnt;
        }
      }
    }
  }
  if (uc) { uc->tot += (u64)cnt; if (cnt > 0) uc->mapped += 1; }
  return cnt;
}

#include "qm_sel.inl"
#include "qm_selpack.inl"

}  // namespace qm
