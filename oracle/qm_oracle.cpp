// =============================================================================
// oracle/qm_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A plain CPU restatement of the RapMap v0.6.0 `quasimap` hot path, written
// from the algorithm as described by the reference sources (cited per function
// as <file>:<lines> relative to /root/reference).  It exists only so that the
// HIP path in rapmap_amd/ can be checked bit-for-bit.  Only tests/,
// __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it; the
// product (rapmap_amd/csrc, libqmap_mi355.so) never links, imports or calls it.
//
// PARITY PINNING STATUS.  The reference's own tests hold no golden vectors, and its quasimap
// cannot be built in this image under the round's rules: every template of the path (SACollector, SASearcher, HitManager)
// includes the un-vendored cereal library, and writing a stand-in is not allowed.  What IS pinned against the reference
// itself, compiled from its own source files in place (oracle/Makefile.ref -> oracle/_ref, tests/test_oracle_ref.py):
//   kswExtz2 (the byte-exact ksw_extz2_sse41 emulation)  == ksw2pp::KSW2Aligner(EXTENSION) as getAlnScore calls it
//   kmerFromChars / wordRC / isHomopolymer               == Kmer<32,1>::fromChars / getRC / isHomoPolymer
//   rank                                                 == rank9b::rank over the same rsd.bin bits
//   (and oracle/q5ph.py's BooPHF lookup                  == boomphf::mphf::load + lookup on a .bph the reference wrote)
// The collector, the MMP search, hits->mappings and the merges are "parity unpinned": corroborated by the SAM the survey
// stage's probe build of the unmodified reference wrote (tests/golden/, tests/test_oracle_golden.py -- 30 option sets incl.
// the sample_data digest of SURVEY.md section 8c) and by an index the reference wrote (tests/test_reference_index.py), both
// produced with a cereal stand-in and therefore not a pin under the rules.  See DESIGN.md section 2.
//
// Layout of the inputs (all flat arrays owned by the caller, see oracle/q5.py):
//   text  : the concatenated transcript text, one '$' after each transcript
//   SA    : suffix array of `text` (int32)
//   txpOffsets : start of each transcript in `text`
//   rsd   : bit i set <=> text[i]=='$'  (64-bit little-endian words)
//   hkeys/hlb/hub : the k-mer -> [lb,ub) SA-interval map (any order)
// =============================================================================
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

namespace {

// ----------------------------------------------------------------------------
// work counters (per pair averages feed the roofline's "algorithmic bytes",
// SURVEY.md section 8d)
// ----------------------------------------------------------------------------
struct Work {
  uint64_t n_probe = 0;  // hash finds
  uint64_t n_sa = 0;     // SA element reads (binary search + interval walks)
  uint64_t n_text = 0;   // text bytes compared
  uint64_t n_rank = 0;   // rank / transcriptAtPosition calls
  uint64_t n_hits = 0;   // final hits written
  uint64_t n_aln = 0;    // -s: ksw2 extension alignments run (cache misses that are not PERFECT / UNGAPPED)
  uint64_t n_cells = 0;  // -s: DP cells inside the band of those alignments
  uint64_t n_ungapped = 0;  // -s: characters compared by the ungapped shortcut
  void add(const Work& o) {
    n_probe += o.n_probe; n_sa += o.n_sa; n_text += o.n_text;
    n_rank += o.n_rank; n_hits += o.n_hits; n_aln += o.n_aln; n_cells += o.n_cells; n_ungapped += o.n_ungapped;
  }
};

// ----------------------------------------------------------------------------
// k-mer codec -- include/Kmer.hpp:40-51 (codes), :92-100 (reverse complement),
// :484-487 (homopolymer), :525-542 (fromCharsIter_)
// ----------------------------------------------------------------------------
struct Codec {
  int8_t code[256];
  Codec() {
    for (int i = 0; i < 256; ++i) code[i] = -1;
    code['A'] = code['a'] = 0; code['C'] = code['c'] = 1;
    code['G'] = code['g'] = 2; code['T'] = code['t'] = 3;
  }
};
static const Codec CODEC;

// Kmer.hpp:525-542: consume up to k chars, first char in the highest bits;
// stop at the first non-ACGT char leaving a partially filled word.
static inline bool kmerFromChars(const char* s, int64_t avail, int k, uint64_t& w) {
  w = 0;
  int shift = 2 * k - 2;
  for (int i = 0; i < k; ++i, shift -= 2) {
    int c = (i < avail) ? CODEC.code[(uint8_t)s[i]] : -1;  // past the end == '\0'
    if (c < 0) return false;
    w |= (uint64_t)c << shift;
  }
  return true;
}

// Kmer.hpp:92-100
static inline uint64_t wordRC(uint64_t w, int k) {
  w = ((w >> 2) & 0x3333333333333333ULL) | ((w & 0x3333333333333333ULL) << 2);
  w = ((w >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((w & 0x0F0F0F0F0F0F0F0FULL) << 4);
  w = ((w >> 8) & 0x00FF00FF00FF00FFULL) | ((w & 0x00FF00FF00FF00FFULL) << 8);
  w = ((w >> 16) & 0x0000FFFF0000FFFFULL) | ((w & 0x0000FFFF0000FFFFULL) << 16);
  w = (w >> 32) | (w << 32);
  return (~w) >> (2 * (32 - k));
}

// Kmer.hpp:484-487
static inline bool isHomopolymer(uint64_t w, int k) {
  uint64_t mask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t nuc = w & 3;
  return w == (mask & ((w << 2) | nuc));
}

// src/RapMapUtils.cpp:63-72,107-128 (reverseRead): A<->T, C<->G either case
// -> upper; 'U'/'u' and 'T'... table: 84('T')->65, 85('U')->65; all else 'N'.
struct RcTable {
  char t[256];
  RcTable() {
    for (int i = 0; i < 256; ++i) t[i] = 'N';
    t['A'] = t['a'] = 'T'; t['C'] = t['c'] = 'G'; t['G'] = t['g'] = 'C';
    t['T'] = t['t'] = 'A'; t['U'] = t['u'] = 'A';
  }
};
static const RcTable RCT;
static void reverseRead(const char* s, int64_t len, std::string& out) {
  out.resize(len);
  // the reference indexes a 128-entry table with (int8_t)c; bytes >= 128 are
  // out of its domain -- we map them to 'N' like every other non-nucleotide.
  for (int64_t i = 0; i < len; ++i) out[len - 1 - i] = RCT.t[(uint8_t)s[i]];
}

// ----------------------------------------------------------------------------
// index view
// ----------------------------------------------------------------------------
// RapMapSAIndex<IndexT, ...>::IndexType: int32_t, or int64_t for a BigSA index (src/RapMapSAMapper.cpp:1209-1240).  The file is
// compiled once per type: libqm_oracle.so (int32_t) and libqm_oracle64.so (-DQO_INDEX_T=int64_t), see the Makefile.
#ifndef QO_INDEX_T
#define QO_INDEX_T int32_t
#endif
typedef QO_INDEX_T IndexT;
struct SAInterval { IndexT lb, ub; };

struct OIndex {
  int k = 31;
  const uint8_t* text = nullptr; int64_t n = 0;
  const IndexT* SA = nullptr; int64_t nSA = 0;
  const IndexT* txpOffsets = nullptr; int64_t nTxp = 0;
  const uint64_t* rsd = nullptr; uint64_t nbits = 0;
  std::vector<uint64_t> cum;  // #set bits before word w
  std::vector<int64_t> txpLens;   // src/RapMapSAIndex.cpp:151-163
  // own open-addressing table (the oracle's data structure is free; semantics
  // = exact map lookup, RapMapUtils.hpp:65-67,226-239)
  std::vector<uint64_t> hk; std::vector<SAInterval> hv; uint64_t hmask = 0;

  static inline uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
  }
  const SAInterval* find(uint64_t key, Work& w) const {
    ++w.n_probe;
    uint64_t i = mix(key) & hmask;
    while (true) {
      uint64_t kk = hk[i];
      if (kk == key) return &hv[i];
      if (kk == ~0ULL) return nullptr;
      i = (i + 1) & hmask;
    }
  }
  // src/rank9b.cpp:56-61 semantics: number of set bits in [0,p)
  inline uint64_t rank(uint64_t p, Work& w) const {
    ++w.n_rank;
    uint64_t word = p >> 6;
    uint64_t r = cum[word];
    unsigned off = p & 63;
    if (off) r += __builtin_popcountll(rsd[word] & ((1ULL << off) - 1));
    return r;
  }
};

struct Opts {
  int32_t sensitive;      // !--noSensitive  (RapMapSAMapper.cpp:1113) -> disableNIP
  int32_t strictCheck;    // !--noStrict     (:1114)
  int32_t maxNumHits;     // -m, default 200
  int32_t noOrphans;
  int32_t noDovetail;
  int32_t fuzzy;          // --fuzzyIntersection
  int32_t maxInterval;    // SACollector.hpp:77, 1000
  int32_t pad;
  double quasiCov;        // -z
  // --selAln and its sub-options (RapMapSAMapper.cpp:1011-1023,1120-1175); all ignored unless selAln != 0
  int32_t selAln, hardFilter, matchScore, mismatchPenalty, gapOpen, gapExtend, dpBandwidth, maxMMPExtension;
  int32_t alnPolicy;      // 0 DEFAULT, 1 BT2, 2 BT2_STRICT (SelectiveAlignmentUtils.hpp:13)
  int32_t recoverOrphans; // not restated: qo_map refuses it
  double minScoreFraction, consensusSlack;
};
struct MapCfg {           // rapmap::utils::MappingConfig (RapMapUtils.hpp:83-89) as set up at RapMapSAMapper.cpp:181-189,409-417
  bool doChaining = false, considerMultiPos = false; float consensusFraction = 1.0f;
};
static MapCfg mapCfg(const Opts& o) {
  MapCfg mc;
  mc.doChaining = o.selAln != 0;
  if (mc.doChaining) {
    float consensusSlack = (float)o.consensusSlack;     // MappingOpts::consensusSlack is a float
    mc.consensusFraction = (consensusSlack == 0.0) ? 1.0 : (1.0 - consensusSlack);
    mc.considerMultiPos = true;
  }
  return mc;
}

// POD of SURVEY.md section 8 row a16
struct Hit {
  uint32_t tid; int32_t pos; int32_t matePos; uint32_t fragLen;
  uint32_t readLen; uint32_t mateLen;
  uint8_t fwd, mateIsFwd, isPaired, mateStatus;
  int32_t alnScore;
};
static_assert(sizeof(Hit) == 32, "hit POD is 32 bytes");

struct SAIntervalHit { IndexT begin, end; uint32_t len, queryPos; uint8_t queryRC; };

enum : uint8_t { SINGLE_END = 0, PE_LEFT = 1, PE_RIGHT = 2, PE_PAIRED = 3 };

// ----------------------------------------------------------------------------
// SASearcher::extendSearchNaive -- include/SASearcher.hpp:88-309
// ----------------------------------------------------------------------------
static inline signed char up(char c) {
  return (c >= 'a' && c <= 'z') ? (signed char)(c - 32) : (signed char)c;
}

static std::tuple<IndexT, IndexT, IndexT>
extendSearchNaive(const OIndex& ix, IndexT lbIn, IndexT ubIn, IndexT startAt,
                  const char* qb, int64_t m, Work& w) {
  const IndexT* SA = ix.SA;
  const signed char* sb = (const signed char*)ix.text;
  const int64_t n = ix.n;

  if (ubIn - lbIn == 2) {                      // :109-126
    lbIn += 1;
    int64_t i = startAt;
    ++w.n_sa;
    int64_t s = SA[lbIn];
    while (i < m && s + i < n) {
      ++w.n_text;
      signed char q = up(qb[i]);
      if (q < sb[s + i]) break; else if (q > sb[s + i]) break;
      ++i;
    }
    return std::make_tuple(lbIn, ubIn, (IndexT)i);
  }

  int64_t l = lbIn, r = ubIn;
  int64_t lcpLP = startAt, lcpRP = startAt;
  int64_t c = 0, i = 0;
  int64_t prevILow = startAt, prevIHigh = startAt;
  int64_t maxLen = 0;
  bool plt = true;
  while (true) {                               // :150-209
    c = (l + r) / 2;
    plt = true;
    i = std::min(lcpLP, lcpRP);
    ++w.n_sa;
    int64_t s = SA[c];
    while (i < m && s + i < n) {
      ++w.n_text;
      signed char q = up(qb[i]);
      if (q < sb[s + i]) {
        if (i > prevIHigh) prevIHigh = i;
        break;
      } else if (q > sb[s + i]) {
        if (i > prevILow) prevILow = i;
        plt = false;
        break;
      }
      ++i;
    }
    if (i == m || s + i == n) {
      if (i > prevIHigh) prevIHigh = i;
    }
    if (plt) {
      if (c == l + 1) { maxLen = std::max(std::max(i, prevILow), prevIHigh); break; }
      r = c; lcpRP = i;
    } else {
      if (c == r - 1) { maxLen = std::max(std::max(i, prevILow), prevIHigh); break; }
      l = c; lcpLP = i;
    }
  }

  m = maxLen + 1;                              // :212
  int64_t bound1 = 0, bound2 = 0;
  for (int pass = 0; pass < 2; ++pass) {       // :215-258 and :261-304
    const signed char sentinel = pass == 0 ? '#' : '{';
    l = pass == 0 ? (int64_t)lbIn : bound1 - 1;
    r = ubIn;
    lcpLP = startAt; lcpRP = startAt;
    while (true) {
      c = (l + r) / 2;
      plt = true;
      i = std::min(lcpLP, lcpRP);
      ++w.n_sa;
      int64_t s = SA[c];
      while (i < m && s + i < n) {
        ++w.n_text;
        signed char q = (i < m - 1) ? up(qb[i]) : sentinel;
        if (q < sb[s + i]) break;
        else if (q > sb[s + i]) { plt = false; break; }
        ++i;
      }
      int64_t res;
      if (plt) {
        if (c == l + 1) { res = c; goto done; }
        r = c; lcpRP = i;
      } else {
        if (c == r - 1) { res = r; goto done; }
        l = c; lcpLP = i;
      }
      continue;
    done:
      if (pass == 0) bound1 = res; else bound2 = res;
      break;
    }
  }
  if (bound1 == bound2) bound2 += 1;           // :307
  return std::make_tuple((IndexT)bound1, (IndexT)bound2, (IndexT)maxLen);
}

// SASearcher::lce -- include/SASearcher.hpp:318-334 (NIP only)
static IndexT lce(const OIndex& ix, IndexT p1, IndexT p2, IndexT startAt,
                  IndexT stopAt, Work& w) {
  IndexT len = startAt;
  w.n_sa += 2;
  int64_t o1 = (int64_t)ix.SA[p1] + startAt;
  int64_t o2 = (int64_t)ix.SA[p2] + startAt;
  int64_t maxIndex = std::max(o1, o2);
  IndexT textLen = (IndexT)ix.n;
  while (maxIndex + len < textLen && ix.text[o1 + len] == ix.text[o2 + len]) {
    ++w.n_text;
    if (ix.text[o1 + len] == '$') break;
    if (len >= stopAt) break;
    ++len;
  }
  return len;
}

// ----------------------------------------------------------------------------
// SACollector -- include/SACollector.hpp
// ----------------------------------------------------------------------------
static std::atomic<long long> g_kposTies{0}, g_kposConflicts{0}, g_kposConflictsNoU{0}, g_kposConflictReads{0}, g_kposDecisionDiffers{0};   // the --noSensitive vote: see SACollector::operator()'s restatement
struct KmerDirScore { uint64_t kmer; int32_t kpos; int8_t fwdScore, rcScore; };

struct Collector {
  const OIndex& ix;
  bool disableNIP, strictCheck;
  double covReq;
  int32_t maxInterval;
  Work& w;
  std::string rcBuffer;
  bool doChaining = false;            // enableChainScoring (SACollector.hpp:64)
  int32_t maxMMPExtension = 7;        // :71,686
  size_t strictCheckSlack = 0;        // :64-65

  Collector(const OIndex& i, const Opts& o, Work& wk)
      : ix(i), disableNIP(o.sensitive != 0), strictCheck(o.strictCheck != 0),
        covReq(o.quasiCov), maxInterval(o.maxInterval), w(wk) {
    if (o.selAln) {                                      // RapMapSAMapper.cpp:187-188,415-416
      doChaining = true; strictCheckSlack = 1;
      if (o.maxMMPExtension > 0) maxMMPExtension = o.maxMMPExtension;
    }
  }

  static size_t findN(const char* s, size_t len, size_t from) {
    for (size_t i = from; i < len; ++i) if (s[i] == 'N' || s[i] == 'n') return i;
    return std::string::npos;
  }

  // SACollector.hpp:366-431
  void spotCheck(uint64_t mer, size_t pos, size_t readLen, const SAInterval** merItPtr,
                 bool isRC, uint32_t& strandHits, uint32_t& otherStrandHits,
                 std::vector<KmerDirScore>& kmerScores) {
    const int k = ix.k;
    uint64_t complementMer = wordRC(mer, k);
    const SAInterval* merIt = merItPtr ? *merItPtr : ix.find(mer, w);
    const SAInterval* compIt = ix.find(complementMer, w);
    int8_t status, cstatus;
    if (merIt) { ++strandHits; status = 1; } else status = -1;
    if (compIt) { ++otherStrandHits; cstatus = 1; } else cstatus = -1;
    int8_t fwdStatus = isRC ? cstatus : status;
    int8_t rcStatus = isRC ? status : cstatus;
    if (strictCheck) {
      if (isRC) { pos = readLen - pos - k; mer = complementMer; }
      kmerScores.push_back({mer, (int32_t)pos, fwdStatus, rcStatus});
    }
  }

  // SACollector.hpp:441-677
  void getSAHits(const char* read, size_t readLen, size_t startPos,
                 const SAInterval* startInterval, size_t& cov, uint32_t& strandHits,
                 uint32_t& otherStrandHits, std::vector<SAIntervalHit>& saInts,
                 std::vector<KmerDirScore>& kmerScores, bool isRC) {
    const int k = ix.k;
    const int64_t skipOverlap = k - 1;
    int64_t rb = 0;                 // offsets instead of iterators
    const int64_t readEnd = (int64_t)readLen;
    IndexT lb = 0, ub = 0, matchedLen = 0;
    size_t invalidPos = 0, pos = 0;
    uint64_t mer = 0;
    const SAInterval* merIt = nullptr;
    bool lastSearch = false;
    size_t prevMMPEnd = 0;
    bool validMer = true;
    bool skipSetup = (startInterval != nullptr);
    if (skipSetup) {
      rb = (int64_t)startPos;
      pos = startPos;
      lb = startInterval->lb; ub = startInterval->ub;
    }
    while (skipSetup || rb + k <= readEnd) {
      bool hit;
      if (skipSetup) {
        hit = true;
      } else {
        pos = (size_t)rb;
        validMer = kmerFromChars(read + pos, readEnd - (int64_t)pos, k, mer);
        if (!validMer) {
          invalidPos = findN(read, readLen, pos);
          if (invalidPos < pos + k) { rb = (int64_t)invalidPos + 1; continue; }
        }
        if (isHomopolymer(mer, k)) { rb += 1; continue; }
        merIt = ix.find(mer, w);
        hit = (merIt != nullptr);
        if (hit) {
          spotCheck(mer, pos, readLen, &merIt, isRC, strandHits, otherStrandHits, kmerScores);
          lb = merIt->lb; ub = merIt->ub;
        }
      }
      if (hit) {
        skipSetup = false;
        lb = std::max((IndexT)0, lb - 1);
        // :557-575 -- with chain scoring only the MMP that starts the read may run to the read's end; every
        // other one is cut at k + maxMMPExtension characters, and a first MMP longer than that is redone cut
        bool firstAttempt = doChaining ? (rb == 0) : true;
        int64_t endOff = firstAttempt ? readEnd : std::min(rb + k + maxMMPExtension, readEnd);
        const IndexT lbP = lb, ubP = ub;
        std::tie(lb, ub, matchedLen) =
            extendSearchNaive(ix, lb, ub, k, read + rb, endOff - rb, w);
        if (doChaining && firstAttempt && !(matchedLen >= (IndexT)readLen) && matchedLen >= (IndexT)(k + maxMMPExtension)) {
          endOff = std::min(rb + k + maxMMPExtension, readEnd);
          std::tie(lb, ub, matchedLen) = extendSearchNaive(ix, lbP, ubP, k, read + rb, endOff - rb, w);
        }
        IndexT diff = ub - lb;
        if (ub > lb && diff < maxInterval) {
          uint32_t queryStart = (uint32_t)rb;
          saInts.push_back({lb, ub, (uint32_t)matchedLen, queryStart, (uint8_t)isRC});
          size_t matchOffset = (size_t)rb;
          size_t correction = 0;
          if (prevMMPEnd > matchOffset) correction = prevMMPEnd - matchOffset;
          cov += (matchedLen - correction);
          prevMMPEnd = matchOffset + matchedLen;
          if (rb + matchedLen < readEnd) {
            uint32_t kmerPos = (uint32_t)(rb + matchedLen - skipOverlap);
            bool validNucs = kmerFromChars(read + kmerPos, readEnd - (int64_t)kmerPos, k, mer);
            if (validNucs)
              spotCheck(mer, kmerPos, readLen, nullptr, isRC, strandHits, otherStrandHits, kmerScores);
          }
        }
        if (lastSearch) return;
        int64_t mismatch = rb + matchedLen;
        if (mismatch >= readEnd) return;
        int64_t remaining = readEnd - mismatch;
        IndexT lceLen = disableNIP ? matchedLen
                                   : lce(ix, lb, ub - 1, matchedLen, (IndexT)remaining, w);
        int64_t skipMatch = mismatch - skipOverlap;
        int64_t skipLCE = rb + lceLen - skipOverlap;
        rb = std::max(skipMatch, skipLCE);
        if (!disableNIP && lceLen > matchedLen) {
          if ((int64_t)readLen > k) rb = std::min(readEnd - k, rb);
        }
        if (rb + k == readEnd) lastSearch = true;
      } else {
        const SAInterval* endIt = nullptr;
        spotCheck(mer, pos, readLen, &endIt, isRC, strandHits, otherStrandHits, kmerScores);
        rb += 1;
      }
    }
  }

  // SACollector.hpp:108-362
  bool collect(const char* read, size_t readLen, std::vector<SAIntervalHit>& fwdSAInts,
               std::vector<SAIntervalHit>& rcSAInts) {
    const int k = ix.k;
    int64_t rb = 0;
    const int64_t readEnd = (int64_t)readLen;
    uint32_t fwdHit = 0, rcHit = 0;
    size_t fwdCov = 0, rcCov = 0;
    bool foundHit = false;
    uint64_t mer = 0, rcMer = 0;
    bool useCoverageCheck = disableNIP && strictCheck;
    std::vector<KmerDirScore> kmerScores;
    const SAInterval* merIt = nullptr;
    const SAInterval* rcMerIt = nullptr;
    size_t pos = 0, invalidPos = 0;

    while (rb + k <= readEnd) {                         // :167-237
      pos = (size_t)rb;
      if (invalidPos != std::string::npos) {
        invalidPos = findN(read, readLen, pos);
        if (invalidPos <= pos + k) { rb = (int64_t)invalidPos + 1; continue; }
      }
      kmerFromChars(read + pos, readEnd - (int64_t)pos, k, mer);   // result ignored (:187)
      if (isHomopolymer(mer, k)) { rb += 1; continue; }
      rcMer = wordRC(mer, k);
      merIt = ix.find(mer, w);
      rcMerIt = ix.find(rcMer, w);
      if (merIt) {
        ++fwdHit;
        if (rcMerIt) {
          ++rcHit;
          if (strictCheck) kmerScores.push_back({mer, (int32_t)pos, 1, 1});
        } else {
          if (strictCheck) kmerScores.push_back({mer, (int32_t)pos, 1, -1});
        }
      }
      if (rcMerIt) {
        if (!fwdHit) {
          ++rcHit;
          if (strictCheck) kmerScores.push_back({mer, (int32_t)pos, -1, 1});
        }
      }
      if (fwdHit + rcHit > 0) { foundHit = true; break; }
      rb += 1;
    }
    if (!foundHit) return false;

    bool didCheckFwd = false;
    if (fwdHit) {                                       // :247-254
      didCheckFwd = true;
      getSAHits(read, readLen, (size_t)rb, merIt, fwdCov, fwdHit, rcHit, fwdSAInts, kmerScores, false);
    }
    bool checkRC = useCoverageCheck ? (rcHit > 0) : (rcHit >= fwdHit);
    if (checkRC) {                                      // :258-265
      reverseRead(read, (int64_t)readLen, rcBuffer);
      getSAHits(rcBuffer.data(), readLen, 0, nullptr, rcCov, rcHit, fwdHit, rcSAInts, kmerScores, true);
    }
    bool checkFwd = useCoverageCheck ? (fwdHit > 0) : (fwdHit >= rcHit);
    if (!didCheckFwd && checkFwd) {                     // :271-278
      didCheckFwd = true;
      getSAHits(read, readLen, 0, nullptr, fwdCov, fwdHit, rcHit, fwdSAInts, kmerScores, false);
    }

    if (strictCheck) {                                  // :280-339
      if (useCoverageCheck) {
        if (fwdCov > rcCov + strictCheckSlack) rcSAInts.clear();
        else if (rcCov > fwdCov + strictCheckSlack) fwdSAInts.clear();
      } else {
        if (fwdHit > 0 && rcHit == 0) rcSAInts.clear();
        else if (rcHit > 0 && fwdHit == 0) fwdSAInts.clear();
        else {
          // std::sort + std::unique on kpos (:297-298).  Entries with equal
          // kpos describe the same read k-mer, hence carry equal statuses for
          // ACGTN reads; a stable sort makes the survivor well defined.
          std::vector<KmerDirScore> asRef(kmerScores.begin(), kmerScores.end());   // (for the comparison below)
          std::stable_sort(kmerScores.begin(), kmerScores.end(),
                           [](const KmerDirScore& a, const KmerDirScore& b) { return a.kpos < b.kpos; });
          // Checked, not assumed.  Entries with equal kpos CAN carry different scores: a window holding 'U' / 'u' is a partial word
          // in the forward pass (Kmer::fromChars stops at the U: absent / absent) and a whole k-mer in the reverse-complement
          // pass (reverseRead maps U to A).  There the reference's answer depends on what its unstable std::sort does with
          // ties -- insertion sort (stable) up to 16 entries in libstdc++, introsort beyond.  This restatement keeps the FIRST
          // entry (stable sort); qo_kpos_ties() reports how many tied positions were seen, how many disagreed, and for how many
          // reads libstdc++'s std::sort on the same sequence would have led to a different strand decision.
          bool conflict = false;
          for (size_t i = 1; i < kmerScores.size(); ++i)
            if (kmerScores[i].kpos == kmerScores[i - 1].kpos) {
              g_kposTies.fetch_add(1, std::memory_order_relaxed);
              if (kmerScores[i].fwdScore != kmerScores[i - 1].fwdScore || kmerScores[i].rcScore != kmerScores[i - 1].rcScore) {
                g_kposConflicts.fetch_add(1, std::memory_order_relaxed); conflict = true;
                bool hasU = false;
                for (int32_t t = kmerScores[i].kpos; t < kmerScores[i].kpos + (int32_t)k && t < (int32_t)readLen; ++t) hasU = hasU || read[t] == 'U' || read[t] == 'u';
                if (!hasU) g_kposConflictsNoU.fetch_add(1, std::memory_order_relaxed);
              }
            }
          if (conflict) {
            std::sort(asRef.begin(), asRef.end(), [](const KmerDirScore& a, const KmerDirScore& b) { return a.kpos < b.kpos; });
            auto e2 = std::unique(asRef.begin(), asRef.end(), [](const KmerDirScore& a, const KmerDirScore& b) { return a.kpos == b.kpos; });
            int32_t f2 = 0, r2 = 0, f1 = 0, r1 = 0;
            for (auto it = asRef.begin(); it != e2; ++it) { f2 += it->fwdScore; r2 += it->rcScore; }
            for (size_t i = 0; i < kmerScores.size(); ++i) if (i == 0 || kmerScores[i].kpos != kmerScores[i - 1].kpos) { f1 += kmerScores[i].fwdScore; r1 += kmerScores[i].rcScore; }
            const int d1 = f1 > r1 ? 1 : (r1 > f1 ? -1 : 0), d2 = f2 > r2 ? 1 : (r2 > f2 ? -1 : 0);
            g_kposConflictReads.fetch_add(1, std::memory_order_relaxed);
            if (d1 != d2) g_kposDecisionDiffers.fetch_add(1, std::memory_order_relaxed);
          }
          auto e = std::unique(kmerScores.begin(), kmerScores.end(),
                               [](const KmerDirScore& a, const KmerDirScore& b) { return a.kpos == b.kpos; });
          int32_t fwdScore = 0, rcScore = 0;
          for (auto it = kmerScores.begin(); it != e; ++it) { fwdScore += it->fwdScore; rcScore += it->rcScore; }
          if (fwdScore > rcScore) rcSAInts.clear();
          else if (rcScore > fwdScore) fwdSAInts.clear();
        }
      }
    }
    if (covReq > 0.0 && disableNIP) {                   // :343-358
      if (!fwdSAInts.empty()) {
        double f = fwdCov / static_cast<double>(readLen);
        if (f < covReq) fwdSAInts.clear();
      }
      if (!rcSAInts.empty()) {
        double f = rcCov / static_cast<double>(readLen);
        if (f < covReq) rcSAInts.clear();
      }
    }
    return foundHit;
  }
};

// ----------------------------------------------------------------------------
// hit_manager -- src/HitManager.cpp
// ----------------------------------------------------------------------------
enum : uint8_t { CS_PERFECT = 0, CS_UNGAPPED = 1, CS_REGULAR = 4 };   // rapmap::utils::ChainStatus (RapMapUtils.hpp:289-295)

struct QA {   // the fields of QuasiAlignment that are defined on this path (RapMapUtils.hpp:399-502)
  uint32_t tid; int32_t pos; bool fwd; uint32_t readLen; uint8_t mateStatus;
  // allPositions always starts with pos (HitManager.cpp:269,321,736); more entries only with considerMultiPos.
  // oppositeStrandPositions: the positions of the same-transcript hit of the other orientation (:846-866).
  std::vector<int32_t> allPositions, oppositeStrandPositions;
  bool hasMultiPos = false;
  uint8_t csLeft = CS_REGULAR, csRight = CS_REGULAR;                  // FragmentChainStatus
  double chainScore = std::numeric_limits<double>::lowest();
  QA(uint32_t t, int32_t p, bool f, uint32_t rl, uint8_t ms) : tid(t), pos(p), fwd(f), readLen(rl), mateStatus(ms) {}
};

struct TQ { uint32_t pos, queryPos; bool queryRC; uint32_t len; };     // SATxpQueryPos
struct PSAHit { std::vector<TQ> tqvec; bool active = false; uint32_t numActive = 1; uint32_t lastActiveInterval = 1; };

// HitManager.cpp:587-689 + :449-493 (strictFilter off)
static std::map<int, PSAHit> intersectSAHits(const OIndex& ix, std::vector<SAIntervalHit>& inHits,
                                             float consensusFraction, Work& w) {
  std::map<int, PSAHit> outHits;
  const int32_t sInHitsSize = (int32_t)inHits.size();
  float requiredFrac = sInHitsSize * consensusFraction;
  int32_t requiredNumHits = sInHitsSize;
  int32_t maxSlack = 0;
  if (consensusFraction < 1.0) {                                       // :622-628
    requiredNumHits = std::max((int32_t)1, (int32_t)std::floor(requiredFrac));
    maxSlack = sInHitsSize - requiredNumHits;
  }
  SAIntervalHit* minHit = &inHits[0];
  for (auto& h : inHits)
    if ((h.end - h.begin) < (minHit->end - minHit->begin)) minHit = &h;
  for (IndexT i = minHit->begin; i < minHit->end; ++i) {
    ++w.n_sa;
    IndexT globalPos = ix.SA[i];
    int tid = (int)ix.rank((uint64_t)globalPos, w);
    int32_t txpPos = (int32_t)(globalPos - ix.txpOffsets[tid]);
    auto& oh = outHits[tid];
    oh.tqvec.push_back({(uint32_t)txpPos, minHit->queryPos, (bool)minHit->queryRC, minHit->len});
    oh.lastActiveInterval = 1;
  }
  const bool nonStrictIntersection = maxSlack > 0;
  uint32_t intervalCounter = 2;
  for (auto& h : inHits) {
    if (&h == minHit) continue;
    for (IndexT i = h.begin; i != h.end; ++i) {       // :463-492
      ++w.n_sa;
      IndexT globalPos = ix.SA[i];
      int txpID = (int)ix.rank((uint64_t)globalPos, w);
      auto it = outHits.find(txpID);
      bool inOutputSet = (it != outHits.end());
      int32_t occ = inOutputSet ? (int32_t)it->second.numActive : 0;
      int32_t slack = ((int32_t)intervalCounter - 1) - occ;
      if (nonStrictIntersection || slack <= maxSlack) {
        int32_t localPos = (int32_t)(globalPos - ix.txpOffsets[txpID]);
        if (inOutputSet) {
          it->second.numActive += (it->second.lastActiveInterval == intervalCounter) ? 0 : 1;
          it->second.lastActiveInterval = intervalCounter;
          it->second.tqvec.push_back({(uint32_t)localPos, h.queryPos, (bool)h.queryRC, h.len});
        } else {
          auto& oh = outHits[txpID];
          oh.tqvec.push_back({(uint32_t)localPos, h.queryPos, (bool)h.queryRC, h.len});
          oh.lastActiveInterval = intervalCounter;
        }
      }
    }
    ++intervalCounter;
  }
  size_t numActive = 0;
  for (auto& kv : outHits) {
    kv.second.active = ((int32_t)kv.second.numActive >= requiredNumHits);
    numActive += kv.second.active ? 1 : 0;
  }
  if (maxSlack > 0 && numActive == 0)                                  // :682-686
    for (auto& kv : outHits) kv.second.active = true;
  return outHits;
}

// fastapprox's fastlog2 as used by the chain score (HitManager.cpp:33-42)
static inline float fastlog2(float x) {
  union { float f; uint32_t i; } vx = { x };
  union { uint32_t i; float f; } mx = { (vx.i & 0x007FFFFF) | 0x3f000000 };
  float y = vx.i;
  y *= 1.1920928955078125e-7f;
  return y - 124.22551499f - 1.498030302f * mx.f - 1.72587999f / (0.3520887068f + mx.f);
}

// HitManager.cpp:84-326
static void collectHitsSimpleSA(std::map<int, PSAHit>& processed, uint32_t readLen, int32_t maxDist,
                                std::vector<QA>& hits, uint8_t mateStatus, const MapCfg& mc) {
  const bool findBestChain = mc.doChaining, considerMultiPos = mc.considerMultiPos;
  std::vector<double> f; std::vector<int32_t> p; std::vector<int32_t> bestChainEndInds;
  for (auto& ph : processed) {
    if (!ph.second.active) continue;
    const uint32_t tid = (uint32_t)ph.first;
    if (findBestChain) {                                               // :107-307 (minimap2-style chaining)
      auto& hitVector = ph.second.tqvec;
      std::sort(hitVector.begin(), hitVector.end(), [](const TQ& p1, const TQ& p2) -> bool {
        auto r1 = p1.pos + p1.len; auto r2 = p2.pos + p2.len;
        auto q1 = p1.queryPos + p1.len; auto q2 = p2.queryPos + p2.len;
        return (r1 < r2) ? true : ((r2 < r1) ? false : (q1 < q2));
      });
      auto alpha = [](int32_t qdiff, int32_t rdiff, int32_t ilen) -> double {
        double score = ilen;
        double mindiff = (qdiff < rdiff) ? qdiff : rdiff;
        return (score < mindiff) ? score : mindiff;
      };
      auto beta = [maxDist](int32_t qdiff, int32_t rdiff, double avgseed) -> double {
        if (qdiff < 0 || (std::max(qdiff, rdiff) > maxDist)) return std::numeric_limits<double>::infinity();
        double l = qdiff - rdiff;
        int32_t al = std::abs(l);
        return (l == 0) ? 0.0 : (0.01 * avgseed * al + 0.5 * fastlog2(static_cast<float>(al)));
      };
      double bestScore = std::numeric_limits<double>::lowest();
      int32_t bestChainEnd = -1;
      const double avgseed = 31.0;
      bestChainEndInds.clear(); f.clear(); p.clear();
      const int32_t lastHitId = (int32_t)hitVector.size() - 1;
      for (int32_t i = 0; i < (int32_t)hitVector.size(); ++i) {
        auto& hi = hitVector[i];
        auto qposi = hi.queryPos + hi.len; auto rposi = hi.pos + hi.len;
        p.push_back(i); f.push_back((double)hi.len);
        int32_t numRounds = 2;
        for (int32_t j = i - 1; j >= 0; --j) {
          auto& hj = hitVector[j];
          auto qposj = hj.queryPos + hj.len; auto rposj = hj.pos + hj.len;
          auto qdiff = qposi - qposj; auto rdiff = rposi - rposj;      // uint32 differences, narrowed by the callees
          auto extensionScore = f[j] + alpha(qdiff, rdiff, hi.len) - beta(qdiff, rdiff, avgseed);
          bool extendWithJ = (extensionScore > f[i]);
          p[i] = extendWithJ ? j : p[i];
          f[i] = extendWithJ ? extensionScore : f[i];
          if (p[i] < i) { numRounds--; if (numRounds <= 0) break; }
        }
        if (f[i] > bestScore) {
          bestScore = f[i]; bestChainEnd = i;
          if (considerMultiPos) { bestChainEndInds.clear(); bestChainEndInds.push_back(bestChainEnd); }
        } else if (considerMultiPos && f[i] == bestScore) {
          bestChainEndInds.push_back(i);
        }
      }
      if (!considerMultiPos) bestChainEndInds.push_back(bestChainEnd);
      // multi-chain backtracking (:206-246)
      size_t numDistinctOpt = 0;
      std::vector<int8_t> seen(f.size(), 0);
      std::vector<int32_t> startPositions;                             // indices into hitVector
      const int32_t lastChainHit = bestChainEnd;
      for (int32_t bestChainEndInd : bestChainEndInds) {
        bool validChain = true;
        if (bestChainEndInd >= 0) {
          int32_t lastPtr = p[bestChainEndInd];
          while (lastPtr < bestChainEndInd) {
            if (seen[bestChainEndInd]) { validChain = false; break; }
            seen[bestChainEndInd] = 1;
            bestChainEndInd = lastPtr;
            lastPtr = p[bestChainEndInd];
          }
          if (seen[bestChainEndInd]) validChain = false;
          if (validChain) { ++numDistinctOpt; startPositions.push_back(lastPtr); }
        }
      }
      if (startPositions.empty()) continue;                            // the reference would dereference end()
      {
        const TQ& st = hitVector[startPositions[0]];
        int32_t hitPos = (int32_t)(st.pos - st.queryPos);
        hits.emplace_back(tid, hitPos, !st.queryRC, readLen, mateStatus);
        QA& currHit = hits.back();
        currHit.chainScore = bestScore;
        currHit.allPositions.push_back(hitPos);
        if (startPositions.size() > 1) {
          currHit.hasMultiPos = true;
          for (size_t t = 1; t < startPositions.size(); ++t) {
            const TQ& o = hitVector[startPositions[t]];
            currHit.allPositions.push_back((int32_t)(o.pos - o.queryPos));
          }
          std::sort(currHit.allPositions.begin(), currHit.allPositions.end());
        }
      }
      if (hitVector.size() > 1 && numDistinctOpt == 1 && lastChainHit == lastHitId) {   // gapless chain (:283-305)
        const TQ& lastHit = hitVector[lastHitId];
        const TQ& minPos = hitVector[0];
        int64_t queryRange = (int64_t)(lastHit.queryPos + lastHit.len) - minPos.queryPos;
        int64_t refRange = (int64_t)(lastHit.pos + lastHit.len) - minPos.pos;
        if (queryRange == refRange && queryRange == (int64_t)readLen) {
          if (mateStatus == PE_RIGHT) hits.back().csRight = CS_UNGAPPED;
          else if (mateStatus == SINGLE_END || mateStatus == PE_LEFT) hits.back().csLeft = CS_UNGAPPED;
        }
      }
    } else {                                                           // :308-322
      auto& tq = ph.second.tqvec;
      auto minIt = std::min_element(tq.begin(), tq.end(),
                                    [](const TQ& a, const TQ& b) { return a.pos < b.pos; });
      int32_t hitPos = (int32_t)(minIt->pos - minIt->queryPos);
      hits.emplace_back(tid, hitPos, !minIt->queryRC, readLen, mateStatus);
      hits.back().allPositions.push_back(hitPos);
    }
  }
}

// HitManager.cpp:691-882
static void hitsToMappingsSimple(const OIndex& ix, const MapCfg& mc, uint8_t mateStatus, uint32_t readLen,
                                 std::vector<SAIntervalHit>& fwdSAInts,
                                 std::vector<SAIntervalHit>& rcSAInts, std::vector<QA>& hits, Work& w) {
  const int32_t maxDist = (int32_t)readLen;                            // HitCollectorInfo::maxDist (SACollector.hpp:145-146)
  size_t fwdHitsStart = hits.size();
  auto collectFromSingleInterval = [&](std::vector<SAIntervalHit>& saInts, bool isFw) {   // :716-807
    auto& h = saInts.front();
    size_t initialSize = hits.size();
    for (IndexT i = h.begin; i != h.end; ++i) {
      ++w.n_sa;
      IndexT globalPos = ix.SA[i];
      uint32_t txpID = (uint32_t)ix.rank((uint64_t)globalPos, w);
      int32_t pos = (int32_t)(globalPos - ix.txpOffsets[txpID]);
      int32_t hitPos = (int32_t)((uint32_t)pos - h.queryPos);
      hits.emplace_back(txpID, hitPos, isFw, readLen, mateStatus);
      QA& lastHit = hits.back();
      lastHit.allPositions.push_back(hitPos);
      const uint8_t cs = (h.len == readLen) ? CS_PERFECT : CS_REGULAR;  // :741-752
      if (mateStatus == PE_RIGHT) lastHit.csRight = cs;
      else if (mateStatus == PE_LEFT || mateStatus == SINGLE_END) lastHit.csLeft = cs;
    }
    std::sort(hits.begin() + initialSize, hits.end(), [](const QA& a, const QA& b) {
      return (a.tid == b.tid) ? (a.pos < b.pos) : (a.tid < b.tid);
    });
    auto first = hits.begin() + initialSize, last = hits.end();
    if (first == last) return;
    auto result = first;
    if (mc.considerMultiPos) {                                          // mergeUnique :770-793
      while (++first != last) {
        bool distinct = !(result->tid == first->tid);
        if (distinct && ++result != first) { *result = std::move(*first); }
        else if (!distinct) { result->hasMultiPos = true; result->allPositions.push_back(first->pos); }
      }
    } else {                                                            // std::unique on tid :797-803
      while (++first != last)
        if (!(result->tid == first->tid) && ++result != first) *result = std::move(*first);
    }
    hits.erase(++result, hits.end());
  };
  if (fwdSAInts.size() > 1) {
    auto ph = intersectSAHits(ix, fwdSAInts, mc.consensusFraction, w);
    collectHitsSimpleSA(ph, readLen, maxDist, hits, mateStatus, mc);
  } else if (fwdSAInts.size() == 1) {
    collectFromSingleInterval(fwdSAInts, true);
  }
  size_t fwdHitsEnd = hits.size();
  size_t rcHitsStart = fwdHitsEnd;
  if (rcSAInts.size() > 1) {
    auto ph = intersectSAHits(ix, rcSAInts, mc.consensusFraction, w);
    collectHitsSimpleSA(ph, readLen, maxDist, hits, mateStatus, mc);
  } else if (rcSAInts.size() == 1) {
    collectFromSingleInterval(rcSAInts, false);
  }
  size_t rcHitsEnd = hits.size();
  if (fwdHitsEnd > fwdHitsStart && rcHitsEnd > rcHitsStart) {   // :834-881
    // ties on tid: the better chain score first (:838-842); without chaining every score is equal and the
    // stable merge keeps the forward entry first
    std::inplace_merge(hits.begin() + fwdHitsStart, hits.begin() + fwdHitsEnd, hits.begin() + rcHitsEnd,
                       [](const QA& a, const QA& b) { return (a.tid == b.tid) ? a.chainScore > b.chainScore : a.tid < b.tid; });
    // mergeOrientationUnique :846-866 -- the surviving entry of a same-transcript run keeps the
    // positions of the dropped one as its opposite-strand positions
    auto first = hits.begin() + fwdHitsStart, last = hits.begin() + rcHitsEnd;
    auto result = first;
    while (++first != last) {
      bool distinct = !(result->tid == first->tid);
      if (distinct && ++result != first) { *result = std::move(*first); }
      else if (!distinct) { result->oppositeStrandPositions = first->allPositions; }
    }
    hits.erase(++result, hits.begin() + rcHitsEnd);
  }
}

struct Counters { uint64_t peHits, seHits, totHits, numReads, tooManyHits, mappedUnits; };

// include/RapMapUtils.hpp:1185-1264
static void mergeLeftRightHits(std::vector<QA>& leftHits, std::vector<QA>& rightHits,
                               std::vector<Hit>& joint, uint32_t maxNumHits, bool& tooManyHits,
                               Counters& hctr) {
  auto mk = [](const QA& q) {
    Hit h{}; h.tid = q.tid; h.pos = q.pos; h.matePos = 0; h.fragLen = 0; h.readLen = q.readLen;
    h.mateLen = 0; h.fwd = q.fwd; h.mateIsFwd = 1; h.isPaired = 0; h.mateStatus = q.mateStatus; h.alnScore = 0;
    return h;
  };
  if (!leftHits.empty()) {
    auto leftIt = leftHits.begin(), leftEnd = leftHits.end();
    if (!rightHits.empty()) {
      auto rightIt = rightHits.begin(), rightEnd = rightHits.end();
      size_t numHits = 0;
      while (leftIt != leftEnd && rightIt != rightEnd) {
        uint32_t leftTxp = leftIt->tid, rightTxp = rightIt->tid;
        if (leftTxp < rightTxp) { ++leftIt; }
        else {
          if (!(rightTxp < leftTxp)) {
            int32_t startRead1 = std::max(leftIt->pos, 0);
            int32_t startRead2 = std::max(rightIt->pos, 0);
            bool read1First = startRead1 < startRead2;
            int32_t fragStartPos = read1First ? startRead1 : startRead2;
            int32_t fragEndPos = read1First ? (int32_t)(startRead2 + rightIt->readLen)
                                            : (int32_t)(startRead1 + leftIt->readLen);
            uint32_t fragLen = (uint32_t)(fragEndPos - fragStartPos);
            Hit h{}; h.tid = leftTxp; h.pos = startRead1; h.fwd = leftIt->fwd; h.readLen = leftIt->readLen;
            h.fragLen = fragLen; h.isPaired = 1; h.mateLen = rightIt->readLen; h.matePos = startRead2;
            h.mateIsFwd = rightIt->fwd; h.mateStatus = PE_PAIRED; h.alnScore = 0;
            joint.push_back(h);
            ++numHits;
            if (numHits > maxNumHits) { tooManyHits = true; break; }
            ++leftIt;
          }
          ++rightIt;
        }
      }
    }
    if (tooManyHits) { joint.clear(); ++hctr.tooManyHits; }
  }
  if (!joint.empty()) {
    hctr.peHits += joint.size();
  } else if (leftHits.size() + rightHits.size() > 0 && !tooManyHits) {
    hctr.seHits += leftHits.size() + rightHits.size();
    for (auto& q : leftHits) joint.push_back(mk(q));
    for (auto& q : rightHits) joint.push_back(mk(q));
  }
}

// A jointHits entry before it is flattened into the POD: the hit plus what selective alignment needs.
struct JH { Hit h; uint8_t csLeft = CS_REGULAR, csRight = CS_REGULAR; };

// include/RapMapUtils.hpp:864-1183 (--fuzzyIntersection / --selAln).  Returns the MergeResult's only
// consumer-visible effect through `joint`; the result code itself only feeds orphan recovery.
static void mergeLeftRightHitsFuzzy(bool leftMatches, bool rightMatches, std::vector<QA>& leftHits,
                                    std::vector<QA>& rightHits, std::vector<JH>& joint,
                                    uint32_t maxNumHits, bool& tooManyHits, Counters& hctr) {
  auto mk = [](const QA& q) {
    JH j; Hit& h = j.h; h = Hit{};
    h.tid = q.tid; h.pos = q.pos; h.matePos = 0; h.fragLen = 0; h.readLen = q.readLen;
    h.mateLen = 0; h.fwd = q.fwd; h.mateIsFwd = 1; h.isPaired = 0; h.mateStatus = q.mateStatus; h.alnScore = 0;
    j.csLeft = q.csLeft; j.csRight = q.csRight;
    return j;
  };
  constexpr int32_t maxGap = std::numeric_limits<int32_t>::max();
  // findBestHitFWRC :923-988: (fwPos, rcPos, gap) of the closest rc position at or after a fwd position
  auto findBestHitFWRC = [&](std::vector<int32_t>& fwdHits, std::vector<int32_t>& rcHits, int32_t fwdReadLen,
                             int32_t& oF, int32_t& oR, int32_t& oGap) -> bool {
    if (fwdHits.empty() || rcHits.empty()) return false;
    int32_t bestGap = maxGap;
    auto bestFW = fwdHits.begin(); auto bestRC = rcHits.begin();
    auto updateBestGap = [&](std::vector<int32_t>::iterator fwdPosIt, std::vector<int32_t>::iterator rcPosIt) {
      int32_t gap = ((*rcPosIt) >= (*fwdPosIt)) ? std::abs((*rcPosIt) - ((*fwdPosIt) + fwdReadLen)) : maxGap;
      if (gap < bestGap) { bestGap = gap; bestFW = fwdPosIt; bestRC = rcPosIt; }
    };
    auto rcBeg = rcHits.begin(), rcEnd = rcHits.end();
    for (auto pIt = fwdHits.begin(); pIt != fwdHits.end(); ++pIt) {
      auto lbIt = std::lower_bound(rcBeg, rcEnd, *pIt);
      if (lbIt == rcEnd) updateBestGap(pIt, lbIt - 1);
      else if (lbIt == rcBeg) updateBestGap(pIt, lbIt);
      else { updateBestGap(pIt, lbIt); updateBestGap(pIt, lbIt - 1); }
    }
    if (bestGap == maxGap) return false;
    oF = *bestFW; oR = *bestRC; oGap = bestGap;
    return true;
  };
  if (leftHits.empty()) {
    if (!leftMatches && !rightHits.empty()) {
      for (auto& q : rightHits) joint.push_back(mk(q));
      hctr.seHits += rightHits.size();
    }
  } else if (rightHits.empty()) {
    if (!rightMatches) {
      for (auto& q : leftHits) joint.push_back(mk(q));
      hctr.seHits += leftHits.size();
    }
  } else {
    auto leftIt = leftHits.begin(), leftEnd = leftHits.end();
    auto rightIt = rightHits.begin(), rightEnd = rightHits.end();
    size_t numHits = 0;
    while (leftIt != leftEnd && rightIt != rightEnd) {
      uint32_t leftTxp = leftIt->tid, rightTxp = rightIt->tid;
      if (leftTxp < rightTxp) { ++leftIt; }
      else {
        if (!(rightTxp < leftTxp)) {
          auto& leftFwdHits = leftIt->fwd ? leftIt->allPositions : leftIt->oppositeStrandPositions;    // :991-996
          auto& leftRCHits = leftIt->fwd ? leftIt->oppositeStrandPositions : leftIt->allPositions;
          auto& rightFwdHits = rightIt->fwd ? rightIt->allPositions : rightIt->oppositeStrandPositions;
          auto& rightRCHits = rightIt->fwd ? rightIt->oppositeStrandPositions : rightIt->allPositions;
          int32_t f1 = 0, r1 = 0, g1 = maxGap, f2 = 0, r2 = 0, g2 = maxGap;
          bool bestFWRC = findBestHitFWRC(leftFwdHits, rightRCHits, (int32_t)leftIt->readLen, f1, r1, g1);
          bool bestRCFW = findBestHitFWRC(rightFwdHits, leftRCHits, (int32_t)rightIt->readLen, f2, r2, g2);
          bool foundValidHit = false, leftFwd = false, rightFwd = false;
          int32_t bestGap = maxGap, leftPos = -1, rightPos = -1;
          if (bestFWRC) { leftPos = f1; rightPos = r1; bestGap = g1; leftFwd = true; rightFwd = false; foundValidHit = true; }
          if (bestRCFW) {
            if (g2 < bestGap) { leftPos = r2; rightPos = f2; leftFwd = false; rightFwd = true; }
            foundValidHit = true;
          }
          if (foundValidHit) {                                     // :1124-1151
            int32_t startRead1 = std::max(leftPos, 0), startRead2 = std::max(rightPos, 0);
            bool read1First = startRead1 < startRead2;
            int32_t fragStartPos = read1First ? startRead1 : startRead2;
            int32_t fragEndPos = read1First ? (int32_t)(startRead2 + rightIt->readLen)
                                            : (int32_t)(startRead1 + leftIt->readLen);
            JH j; Hit& h = j.h; h = Hit{};
            h.tid = leftTxp; h.pos = leftPos; h.fwd = leftFwd; h.readLen = leftIt->readLen;
            h.fragLen = (uint32_t)(fragEndPos - fragStartPos); h.isPaired = 1; h.mateLen = rightIt->readLen;
            h.matePos = rightPos; h.mateIsFwd = rightFwd; h.mateStatus = PE_PAIRED; h.alnScore = 0;
            j.csLeft = leftIt->csLeft; j.csRight = rightIt->csRight;
            joint.push_back(j);
            ++numHits;
            if (numHits > maxNumHits) { tooManyHits = true; break; }
          }
          ++leftIt;
        }
        ++rightIt;
      }
    }
    if (tooManyHits) { joint.clear(); ++hctr.tooManyHits; }
  }
  if (!joint.empty()) hctr.peHits += joint.size();                 // :1176-1179 (orphans are counted too)
}

// ----------------------------------------------------------------------------
// selective alignment (--selAln): ksw2 extension alignment + score gate
// ----------------------------------------------------------------------------
// ksw_extz2_sse41 (src/ksw2pp/ksw2_extz2_sse.c:18-304) in the configuration the mapper uses: score only, exact
// max (no KSW_EZ_APPROX_MAX), no z-drop.  The SSE kernel works on 16-byte vectors of int8 differences and also
// computes the lanes of a vector that lie outside the band; those values are later picked up as neighbours when
// the band moves, so the byte-level layout and every out-of-band lane are reproduced here one byte at a time.
struct KswOut { int32_t mqe, mte; };
static KswOut kswExtz2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int8_t m, const int8_t* mat,
                       int8_t q, int8_t e, int w, uint64_t* cells) {
  KswOut ez{-0x40000000, -0x40000000};
  if (m <= 0 || qlen <= 0 || tlen <= 0) return ez;
  const int qe = q + e;
  if (w < 0) w = tlen > qlen ? tlen : qlen;
  const int wl = w, wr = w;
  const int tlen_ = (tlen + 15) / 16, qlen_ = (qlen + 15) / 16;
  int min_sc = mat[1];
  for (int t = 1; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
  if (-min_sc > 2 * (q + e)) return ez;
  // one zeroed block laid out like the kernel's: u v x y s (tlen_ vectors each), sf (tlen_), qr (qlen_ + 1)
  std::vector<uint8_t> mem((size_t)(tlen_ * 6 + qlen_ + 1) * 16 + 16, 0);
  uint8_t* u8 = mem.data(); uint8_t* v8 = u8 + tlen_ * 16; uint8_t* x8 = v8 + tlen_ * 16; uint8_t* y8 = x8 + tlen_ * 16;
  uint8_t* s8 = y8 + tlen_ * 16; uint8_t* sf = s8 + tlen_ * 16; uint8_t* qr = sf + tlen_ * 16;
  std::vector<int32_t> H((size_t)tlen_ * 16, -0x40000000);
  for (int t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
  memcpy(sf, target, (size_t)tlen);
  const uint8_t sc_mch = (uint8_t)mat[0], sc_mis = (uint8_t)mat[1], sc_N = (uint8_t)mat[m * m - 1], m1 = (uint8_t)(m - 1);
  const uint8_t qe2 = (uint8_t)((q + e) * 2), max_sc_v = (uint8_t)(mat[0] + (q + e) * 2), qv = (uint8_t)q;
  int last_st = -1, last_en = -1;
  for (int r = 0; r < qlen + tlen - 1; ++r) {
    int st = 0, en = tlen - 1;
    const uint8_t* qrr = qr + (qlen - 1 - r);
    if (st < r - qlen + 1) st = r - qlen + 1;
    if (en > r) en = r;
    if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
    if (en > (r + wl) >> 1) en = (r + wl) >> 1;
    if (st > en) break;                                                // zdropped
    const int st0 = st, en0 = en;
    if (cells) *cells += (uint64_t)(en0 - st0 + 1);
    st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
    uint8_t x1, v1;
    if (st > 0) {
      if (st - 1 >= last_st && st - 1 <= last_en) { x1 = x8[st - 1]; v1 = v8[st - 1]; }
      else { x1 = 0; v1 = 0; }
    } else { x1 = 0; v1 = r ? qv : 0; }
    if (en >= r) { y8[r] = 0; u8[r] = r ? qv : 0; }
    for (int t = st0; t <= en0; t += 16)                                // scores: whole vectors, unaligned loads
      for (int l = 0; l < 16; ++l) {
        const uint8_t sq = sf[t + l], sv = qrr[t + l];
        uint8_t tmp = (sq == sv) ? sc_mch : sc_mis;
        if (sq == m1 || sv == m1) tmp = sc_N;
        s8[t + l] = tmp;
      }
    for (int t = st; t <= en; ++t) {                                    // the int8 recurrences, lane by lane
      uint8_t z = (uint8_t)(s8[t] + qe2);
      const uint8_t xt1 = x1; x1 = x8[t];                               // x[r-1][t-1]
      const uint8_t vt1 = v1; v1 = v8[t];                               // v[r-1][t-1]
      uint8_t a = (uint8_t)(xt1 + vt1);
      const uint8_t ut = u8[t];
      uint8_t b = (uint8_t)(y8[t] + ut);
      z = (uint8_t)((int8_t)z > (int8_t)a ? z : a);                     // _mm_max_epi8 (signed)
      z = z > b ? z : b;                                                // _mm_max_epu8
      z = z < max_sc_v ? z : max_sc_v;                                  // _mm_min_epu8
      u8[t] = (uint8_t)(z - vt1);
      v8[t] = (uint8_t)(z - ut);
      z = (uint8_t)(z - qv);
      a = (uint8_t)(a - z); b = (uint8_t)(b - z);
      x8[t] = (int8_t)a > 0 ? a : 0;                                    // _mm_max_epi8(a, 0)
      y8[t] = (int8_t)b > 0 ? b : 0;
    }
    // exact max with the 32-bit H array (:229-268); max_H / max_t only feed the (disabled) z-drop
    if (r > 0) {
      H[en0] = en0 > 0 ? H[en0 - 1] + u8[en0] - qe : H[en0] + v8[en0] - qe;
      for (int t = st0; t < en0; ++t) H[t] += (int32_t)v8[t] - qe;
    } else H[0] = v8[0] - qe - qe;
    if (en0 == tlen - 1 && H[en0] > ez.mte) ez.mte = H[en0];
    if (r - st0 == qlen - 1 && H[st0] > ez.mqe) ez.mqe = H[st0];
    last_st = st; last_en = en;
  }
  return ez;
}

struct Aligner {                                                       // KSW2Aligner as configured at RapMapSAMapper.cpp:420-436
  int8_t mat[25]; int8_t q, e; int w;
  std::vector<uint8_t> qbuf, tbuf;
  Work* wk = nullptr;
  Aligner(int a, int b, int gapo, int gape, int bw) : q((int8_t)gapo), e((int8_t)gape), w(bw) {
    a = (int8_t)a; b = (int8_t)b;                                      // KSW2Aligner(int8_t match, int8_t mismatch)
    a = a < 0 ? -a : a; b = b > 0 ? -b : b;
    const int m = 5;
    for (int i = 0; i < m - 1; ++i) { for (int j = 0; j < m - 1; ++j) mat[i * m + j] = (int8_t)(i == j ? a : b); mat[i * m + m - 1] = 0; }
    for (int j = 0; j < m; ++j) mat[(m - 1) * m + j] = 0;
  }
  static uint8_t nt4(unsigned char c) {                                // seq_nt4_table_loc (KSW2Aligner.cpp:61-72)
    switch (c) {
      case 0: case 'A': case 'a': return 0; case 1: case 'C': case 'c': return 1;
      case 2: case 'G': case 'g': return 2; case 3: case 'T': case 't': return 3; default: return 4;
    }
  }
  int32_t extension(const char* qs, int ql, const char* ts, int tl) {  // operator()(…, EXTENSION) -> max(mqe, mte)
    qbuf.resize((size_t)ql); tbuf.resize((size_t)tl);
    for (int i = 0; i < ql; ++i) qbuf[(size_t)i] = nt4((unsigned char)qs[i]);
    for (int i = 0; i < tl; ++i) tbuf[(size_t)i] = nt4((unsigned char)ts[i]);
    if (wk) ++wk->n_aln;
    KswOut ez = kswExtz2(ql, qbuf.data(), tl, tbuf.data(), 5, mat, q, e, w, wk ? &wk->n_cells : nullptr);
    return std::max(ez.mqe, ez.mte);
  }
};

// The alignment cache (tsl::hopscotch_map keyed by MetroHash64 of the target window,
// SelectiveAlignmentUtils.hpp:23-30,317-330,361-367) is restated with the window itself as the key: equal
// up to 64-bit hash collisions.  Note that the key leaves out the `buf` extra target characters although the
// alignment sees them -- a cached score is reused for windows that differ only there.
using AlnCache = std::map<std::string, int32_t>;

// selective_alignment::utils::getAlnScore (SelectiveAlignmentUtils.hpp:260-373)
static int32_t getAlnScore(Aligner& aligner, int32_t pos, const char* rptr, int32_t rlen, const char* tseq, int32_t tlen,
                           int8_t mscore, int8_t mmcost, int32_t maxScore, uint8_t chainStat, bool multiMapping,
                           int32_t ap, uint32_t buf, AlnCache& alnCache) {
  if (chainStat == CS_PERFECT) return maxScore;
  auto ungappedAln = [mscore, mmcost](const char* ref, const char* query, int32_t len) -> int32_t {
    int32_t ungappedScore = 0;
    for (int32_t i = 0; i < len; ++i) {
      char c1 = ref[i], c2 = query[i];
      c1 = (c1 == 'N' || c2 == 'N') ? c2 : c1;
      ungappedScore += (c1 == c2) ? mscore : mmcost;
    }
    return ungappedScore;
  };
  int32_t s = std::numeric_limits<int32_t>::lowest();
  bool invalidStart = (pos < 0);
  bool invalidEnd = (pos + rlen >= tlen);
  if (invalidStart) { rptr += -pos; rlen += pos; pos = 0; }
  if ((invalidStart || invalidEnd) && (ap == 1 || ap == 2)) return s;
  if (pos < tlen) {
    bool doUngapped = (!invalidStart) && (chainStat == CS_UNGAPPED);
    buf = doUngapped ? 0 : buf;
    uint32_t lnobuf = (uint32_t)(tlen - pos);
    uint32_t lbuf = (uint32_t)(rlen + buf);
    bool useBuf = (lbuf < lnobuf);
    uint32_t tlen1 = std::min(lbuf, lnobuf);
    const char* tseq1 = tseq + pos;
    std::string key;
    bool didHash = false;
    if (!alnCache.empty()) {
      uint32_t keyLen = useBuf ? tlen1 - buf : tlen1;
      key.assign(tseq1, keyLen); didHash = true;
      auto hit = alnCache.find(key);
      if (hit != alnCache.end()) s = hit->second;
    }
    if (s == std::numeric_limits<int32_t>::lowest()) {
      if (doUngapped) {
        int32_t tlen1s = (int32_t)tlen1;
        int32_t alnLen = rlen < tlen1s ? rlen : tlen1s;
        s = ungappedAln(tseq1, rptr, alnLen);
        if (aligner.wk) aligner.wk->n_ungapped += (uint64_t)alnLen;
      } else {
        s = aligner.extension(rptr, rlen, tseq1, (int)tlen1);
      }
      if (multiMapping) {
        if (!didHash) { uint32_t keyLen = useBuf ? tlen1 - buf : tlen1; key.assign(tseq1, keyLen); }
        alnCache[key] = s;
      }
    }
  }
  return s;
}

// the score gate + soft/hard filter shared by the paired (RapMapSAMapper.cpp:646-683) and single-end (:289-318) drivers
template <typename V, typename GetHit>
static void filterByScore(V& joint, std::vector<int32_t>& scores, int32_t bestScore, bool hardFilter, GetHit hit) {
  if (bestScore > std::numeric_limits<int32_t>::min()) {
    size_t ctr = 0, o = 0;
    for (size_t i = 0; i < joint.size(); ++i) {
      bool rem = hardFilter ? (scores[ctr] < bestScore) : (scores[ctr] == std::numeric_limits<int32_t>::min());
      ++ctr;
      if (!rem) { if (o != i) joint[o] = joint[i]; hit(joint[o]).alnScore = scores[i]; ++o; }
    }
    joint.resize(o);
  } else {
    joint.clear();
  }
}

// per-pair driver -- src/RapMapSAMapper.cpp:461-701
static void mapPair(const OIndex& ix, const Opts& o, const MapCfg& mc, Collector& col, Aligner* aligner, const char* r1, size_t l1,
                    const char* r2, size_t l2, std::vector<Hit>& joint, Counters& hctr, Work& w,
                    std::vector<SAIntervalHit>* dumpInts /* 4 lists or null */) {
  std::vector<SAIntervalHit> lf, lr, rf, rr;
  std::vector<QA> leftHits, rightHits;
  std::vector<JH> jj;
  bool tooManyHits = false;
  ++hctr.numReads;
  joint.clear();
  bool lh = col.collect(r1, l1, lf, lr);
  bool rh = col.collect(r2, l2, rf, rr);
  if (dumpInts) { dumpInts[0] = lf; dumpInts[1] = lr; dumpInts[2] = rf; dumpInts[3] = rr; }
  hitsToMappingsSimple(ix, mc, PE_LEFT, (uint32_t)l1, lf, lr, leftHits, w);
  hitsToMappingsSimple(ix, mc, PE_RIGHT, (uint32_t)l2, rf, rr, rightHits, w);
  if (o.fuzzy || o.selAln) {                                              // useSmartIntersect (:450,488-493)
    mergeLeftRightHitsFuzzy(lh, rh, leftHits, rightHits, jj, (uint32_t)o.maxNumHits, tooManyHits, hctr);
  } else {
    mergeLeftRightHits(leftHits, rightHits, joint, (uint32_t)o.maxNumHits, tooManyHits, hctr);
    for (auto& h : joint) { JH j; j.h = h; jj.push_back(j); }
    joint.clear();
  }
  if (jj.size() > (size_t)o.maxNumHits) jj.clear();                       // :534-536
  if (!jj.empty() && o.noOrphans && jj.front().h.mateStatus != PE_PAIRED) jj.clear();   // :539-551
  auto dovetail = [](const Hit& h) {
    if (h.fwd != h.mateIsFwd) {
      if (h.fwd && (h.pos > h.matePos)) return true;
      else if (h.mateIsFwd && (h.matePos > h.pos)) return true;
    }
    return false;
  };
  if (o.selAln && !jj.empty()) {                                          // :554-667
    AlnCache alnCacheLeft, alnCacheRight;
    std::string rc1, rc2; bool have1 = false, have2 = false;
    const int8_t a = (int8_t)o.matchScore, b = (int8_t)o.mismatchPenalty;
    int32_t bestScore = std::numeric_limits<int32_t>::lowest();
    std::vector<int32_t> scores(jj.size(), bestScore);
    const double optFrac = o.minScoreFraction;
    const int32_t maxLeftScore = a * (int32_t)l1, maxRightScore = a * (int32_t)l2;
    const bool multiMapping = jj.size() > 1;
    size_t idx = 0;
    for (auto& j : jj) {
      Hit& h = j.h;
      int32_t score = std::numeric_limits<int32_t>::min();
      const char* tseq = (const char*)ix.text + ix.txpOffsets[h.tid];
      const int32_t tlen = (int32_t)ix.txpLens[h.tid];
      const uint32_t buf = 20;
      if (h.mateStatus == PE_PAIRED) {
        if (!h.fwd && !have1) { reverseRead(r1, (int64_t)l1, rc1); have1 = true; }
        if (!h.mateIsFwd && !have2) { reverseRead(r2, (int64_t)l2, rc2); have2 = true; }
        const char* r1ptr = h.fwd ? r1 : rc1.data();
        const char* r2ptr = h.mateIsFwd ? r2 : rc2.data();
        int32_t s1 = getAlnScore(*aligner, h.pos, r1ptr, (int32_t)l1, tseq, tlen, a, b, maxLeftScore, j.csLeft, multiMapping, o.alnPolicy, buf, alnCacheLeft);
        int32_t s2 = getAlnScore(*aligner, h.matePos, r2ptr, (int32_t)l2, tseq, tlen, a, b, maxRightScore, j.csRight, multiMapping, o.alnPolicy, buf, alnCacheRight);
        if (h.fwd != h.mateIsFwd && o.noDovetail) {
          if (h.fwd && (h.pos > h.matePos)) { s1 = s2 = std::numeric_limits<int32_t>::min(); }
          else if (h.mateIsFwd && (h.matePos > h.pos)) { s1 = s2 = std::numeric_limits<int32_t>::min(); }
        }
        if ((s1 < (optFrac * maxLeftScore)) || (s2 < (optFrac * maxRightScore))) score = std::numeric_limits<int32_t>::min();
        else score = s1 + s2;
      } else if (h.mateStatus == PE_LEFT) {
        if (!h.fwd && !have1) { reverseRead(r1, (int64_t)l1, rc1); have1 = true; }
        const char* rptr = h.fwd ? r1 : rc1.data();
        int32_t s = getAlnScore(*aligner, h.pos, rptr, (int32_t)l1, tseq, tlen, a, b, maxLeftScore, j.csLeft, multiMapping, o.alnPolicy, buf, alnCacheLeft);
        score = (s < (optFrac * maxLeftScore)) ? std::numeric_limits<int32_t>::min() : s;
      } else if (h.mateStatus == PE_RIGHT) {
        if (!h.fwd && !have2) { reverseRead(r2, (int64_t)l2, rc2); have2 = true; }
        const char* rptr = h.fwd ? r2 : rc2.data();
        int32_t s = getAlnScore(*aligner, h.pos, rptr, (int32_t)l2, tseq, tlen, a, b, maxRightScore, j.csRight, multiMapping, o.alnPolicy, buf, alnCacheRight);
        score = (s < (optFrac * maxRightScore)) ? std::numeric_limits<int32_t>::min() : s;
      }
      bestScore = (score > bestScore) ? score : bestScore;
      scores[idx++] = score;
    }
    filterByScore(jj, scores, bestScore, o.hardFilter != 0, [](JH& j) -> Hit& { return j.h; });
  } else if (o.noDovetail) {                                              // :668-683
    jj.erase(std::remove_if(jj.begin(), jj.end(), [&](const JH& j) { return dovetail(j.h); }), jj.end());
  }
  for (auto& j : jj) joint.push_back(j.h);
  hctr.totHits += joint.size();                                           // :701
  if (!joint.empty()) ++hctr.mappedUnits;
  w.n_hits += joint.size();
}

// single-end driver -- src/RapMapSAMapper.cpp:225-320
static void mapSingle(const OIndex& ix, const Opts& o, const MapCfg& mc, Collector& col, Aligner* aligner, const char* r, size_t l,
                      std::vector<Hit>& out, Counters& hctr, Work& w) {
  std::vector<SAIntervalHit> f, rc;
  std::vector<QA> hits;
  ++hctr.numReads;
  out.clear();
  col.collect(r, l, f, rc);
  hitsToMappingsSimple(ix, mc, SINGLE_END, (uint32_t)l, f, rc, hits, w);
  hctr.totHits += hits.size();            // counted before the maxNumHits clear (:240-245)
  if (hits.size() > (size_t)o.maxNumHits) hits.clear();
  std::vector<JH> jj;
  for (auto& q : hits) {
    JH j; Hit& h = j.h; h = Hit{};
    h.tid = q.tid; h.pos = q.pos; h.readLen = q.readLen; h.fwd = q.fwd; h.mateIsFwd = 1;
    h.mateStatus = SINGLE_END; j.csLeft = q.csLeft; j.csRight = q.csRight; jj.push_back(j);
  }
  if (o.selAln) {                                                          // :246-318
    AlnCache alnCache;
    std::string rc1; bool have1 = false;
    const int8_t a = (int8_t)o.matchScore, b = (int8_t)o.mismatchPenalty;
    int32_t bestScore = std::numeric_limits<int32_t>::lowest();
    std::vector<int32_t> scores(jj.size(), bestScore);
    const double optFrac = o.minScoreFraction;
    const int32_t maxReadScore = a * (int32_t)l;
    const bool multiMapping = jj.size() > 1;
    size_t idx = 0;
    for (auto& j : jj) {
      Hit& h = j.h;
      const char* tseq = (const char*)ix.text + ix.txpOffsets[h.tid];
      const int32_t tlen = (int32_t)ix.txpLens[h.tid];
      if (!h.fwd && !have1) { reverseRead(r, (int64_t)l, rc1); have1 = true; }
      const char* rptr = h.fwd ? r : rc1.data();
      int32_t s = getAlnScore(*aligner, h.pos, rptr, (int32_t)l, tseq, tlen, a, b, maxReadScore, j.csLeft, multiMapping, o.alnPolicy, 20, alnCache);
      int32_t score = (s < (optFrac * maxReadScore)) ? std::numeric_limits<int32_t>::min() : s;
      bestScore = (score > bestScore) ? score : bestScore;
      scores[idx++] = score;
    }
    filterByScore(jj, scores, bestScore, o.hardFilter != 0, [](JH& j) -> Hit& { return j.h; });
  }
  for (auto& j : jj) out.push_back(j.h);
  if (!out.empty()) ++hctr.mappedUnits;
  w.n_hits += out.size();
}

}  // namespace

// =============================================================================
// C entry points (ctypes; see oracle/oracle.py)
// =============================================================================
extern "C" {

int qo_index_bytes(void) { return (int)sizeof(IndexT); }     // 4, or 8 in libqm_oracle64.so

void* qo_index_create(int k, const uint8_t* text, int64_t n, const IndexT* SA, int64_t nSA,
                      const IndexT* txpOffsets, int64_t nTxp, const uint64_t* rsd, uint64_t nbits,
                      const uint64_t* hkeys, const IndexT* hlb, const IndexT* hub, int64_t nKeys) {
  OIndex* ix = new OIndex();
  ix->k = k; ix->text = text; ix->n = n; ix->SA = SA; ix->nSA = nSA;
  ix->txpOffsets = txpOffsets; ix->nTxp = nTxp; ix->rsd = rsd; ix->nbits = nbits;
  uint64_t nwords = (nbits + 63) / 64;
  ix->cum.resize(nwords + 1);
  uint64_t c = 0;
  for (uint64_t i = 0; i < nwords; ++i) { ix->cum[i] = c; c += __builtin_popcountll(rsd[i]); }
  ix->cum[nwords] = c;
  ix->txpLens.resize((size_t)nTxp);
  for (int64_t t = 0; t < nTxp; ++t)
    ix->txpLens[(size_t)t] = t + 1 < nTxp ? (int64_t)txpOffsets[t + 1] - 1 - txpOffsets[t] : nSA - 1 - txpOffsets[t];
  uint64_t cap = 16;
  while (cap < (uint64_t)nKeys * 2) cap <<= 1;
  ix->hk.assign(cap, ~0ULL); ix->hv.resize(cap); ix->hmask = cap - 1;
  for (int64_t i = 0; i < nKeys; ++i) {
    uint64_t s = OIndex::mix(hkeys[i]) & ix->hmask;
    while (ix->hk[s] != ~0ULL) s = (s + 1) & ix->hmask;
    ix->hk[s] = hkeys[i]; ix->hv[s] = {hlb[i], hub[i]};
  }
  return ix;
}

void qo_index_destroy(void* h) { delete (OIndex*)h; }
// the k-mer vote of --noSensitive (include/SACollector.hpp:289-337): positions that entered kmerScores more than once, and how many
// of those carried different (fwdScore, rcScore) in their entries; reset = 1 zeroes both after reading
void qo_kpos_ties(long long* out5, int reset) {
  // [0] tied positions, [1] ties whose entries disagree, [2] ... in a window without U / u, [3] reads with a disagreeing tie,
  // [4] reads where libstdc++'s std::sort (what the reference calls) would have decided the strand differently
  if (out5) { out5[0] = g_kposTies.load(); out5[1] = g_kposConflicts.load(); out5[2] = g_kposConflictsNoU.load(); out5[3] = g_kposConflictReads.load(); out5[4] = g_kposDecisionDiffers.load(); }
  if (reset) { g_kposTies = 0; g_kposConflicts = 0; g_kposConflictsNoU = 0; g_kposConflictReads = 0; g_kposDecisionDiffers = 0; }
}

// Map n read pairs (or n single reads when seq2 == nullptr).
// seqX: concatenated read bytes; offX[n+1]: offsets.
// Outputs: hit_offsets[n+1]; *hits_out = malloc'ed Hit array (free with qo_free);
// counters[6] = {peHits, seHits, totHits, numReads, tooManyHits, mappedUnits};
// work[10] = {n_probe, n_sa, n_text, n_rank, n_hits, n_aln, n_cells, n_ungapped, ns of the mapping section (threads started
// to threads joined: what the reference's ScopedTimer around its worker threads covers, src/RapMapSAMapper.cpp:856-889),
// ns of the whole call (+ assembling one contiguous result array, which the reference never does)}.
// If ints_out != nullptr (pairs only): *ints_out = malloc'ed SA-interval records
// {begin,end,len,queryPos,rc,list} as int32[6], list = 0..3 for
// left-fwd,left-rc,right-fwd,right-rc, and ints_offsets[n+1].
int qo_map(void* hidx, const Opts* opts, int64_t n, const char* seq1, const int64_t* off1,
           const char* seq2, const int64_t* off2, int nthreads, int64_t* hit_offsets, Hit** hits_out,
           uint64_t* counters, uint64_t* work, int64_t* ints_offsets, int32_t** ints_out) {
  const OIndex& ix = *(const OIndex*)hidx;
  if (opts->selAln && opts->recoverOrphans) return -2;      // --recoverOrphans is not restated
  if (nthreads < 1) nthreads = 1;
  std::vector<std::vector<Hit>> perHits(nthreads);
  std::vector<std::vector<int32_t>> perInts(nthreads);
  // per-thread counters on cache lines of their own: they are bumped on every probe, and packed side by side (as they
  // were) the threads spent their time stealing each other's lines
  struct alignas(128) PerThread { Work wk; Counters ctr{0, 0, 0, 0, 0, 0}; };
  std::vector<PerThread> pt(nthreads);
  std::vector<int64_t> cnt(n + 1, 0), icnt(n + 1, 0);
  // static contiguous split: thread t owns [t*n/T,(t+1)*n/T) -- deterministic order
  auto worker = [&](int t) {
    int64_t b = n * t / nthreads, e = n * (t + 1) / nthreads;
    Work& wkt = pt[t].wk; Counters& ctrt = pt[t].ctr;
    Collector col(ix, *opts, wkt);
    const MapCfg mc = mapCfg(*opts);
    Aligner aligner(opts->matchScore, opts->mismatchPenalty, opts->gapOpen, opts->gapExtend, opts->dpBandwidth);
    aligner.wk = &wkt;
    std::vector<Hit> joint;
    perHits[t].reserve((size_t)(e - b) * 4);
    std::vector<SAIntervalHit> dump[4];
    for (int64_t i = b; i < e; ++i) {
      if (seq2) {
        mapPair(ix, *opts, mc, col, &aligner, seq1 + off1[i], (size_t)(off1[i + 1] - off1[i]), seq2 + off2[i],
                (size_t)(off2[i + 1] - off2[i]), joint, ctrt, wkt, ints_out ? dump : nullptr);
        if (ints_out) {
          for (int l = 0; l < 4; ++l)
            for (auto& s : dump[l]) {
              int32_t rec[6] = {(int32_t)(uint32_t)s.begin, (int32_t)(uint32_t)s.end, (int32_t)s.len, (int32_t)s.queryPos, (int32_t)s.queryRC, l};
              perInts[t].insert(perInts[t].end(), rec, rec + 6);
              ++icnt[i + 1];
            }
        }
      } else {
        mapSingle(ix, *opts, mc, col, &aligner, seq1 + off1[i], (size_t)(off1[i + 1] - off1[i]), joint, ctrt, wkt);
      }
      cnt[i + 1] = (int64_t)joint.size();
      perHits[t].insert(perHits[t].end(), joint.begin(), joint.end());
    }
  };
  const auto tc0 = std::chrono::steady_clock::now();
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& x : th) x.join();
  }
  const auto tc1 = std::chrono::steady_clock::now();
  // one contiguous result: every thread places its own share (offsets first: a thread's units are contiguous)
  std::vector<int64_t> base((size_t)nthreads + 1, 0);
  for (int t = 0; t < nthreads; ++t) base[(size_t)t + 1] = base[(size_t)t] + (int64_t)perHits[t].size();
  const int64_t total = base[(size_t)nthreads];
  Hit* out = (Hit*)malloc(sizeof(Hit) * (size_t)std::max<int64_t>(total, 1));
  hit_offsets[0] = 0;
  {
    auto place = [&](int t) {
      const int64_t b = n * t / nthreads, e = n * (t + 1) / nthreads;
      int64_t h = base[(size_t)t];
      for (int64_t i = b; i < e; ++i) { h += cnt[i + 1]; hit_offsets[i + 1] = h; }
      if (!perHits[t].empty()) memcpy(out + base[(size_t)t], perHits[t].data(), perHits[t].size() * sizeof(Hit));
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(place, t);
    place(0);
    for (auto& x : th) x.join();
  }
  *hits_out = out;
  Counters c{0, 0, 0, 0, 0, 0}; Work w;
  for (int t = 0; t < nthreads; ++t) {
    const Counters& ct = pt[t].ctr;
    c.peHits += ct.peHits; c.seHits += ct.seHits; c.totHits += ct.totHits;
    c.numReads += ct.numReads; c.tooManyHits += ct.tooManyHits; c.mappedUnits += ct.mappedUnits;
    w.add(pt[t].wk);
  }
  counters[0] = c.peHits; counters[1] = c.seHits; counters[2] = c.totHits; counters[3] = c.numReads;
  counters[4] = c.tooManyHits; counters[5] = c.mappedUnits;
  work[0] = w.n_probe; work[1] = w.n_sa; work[2] = w.n_text; work[3] = w.n_rank; work[4] = w.n_hits;
  work[5] = w.n_aln; work[6] = w.n_cells; work[7] = w.n_ungapped;
  work[8] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(tc1 - tc0).count();
  if (ints_out) {
    ints_offsets[0] = 0;
    for (int64_t i = 0; i < n; ++i) ints_offsets[i + 1] = ints_offsets[i] + icnt[i + 1];
    int64_t tot = ints_offsets[n];
    int32_t* io = (int32_t*)malloc(sizeof(int32_t) * 6 * (size_t)std::max<int64_t>(tot, 1));
    int64_t q = 0;
    for (int t = 0; t < nthreads; ++t) {
      if (!perInts[t].empty()) memcpy(io + q, perInts[t].data(), perInts[t].size() * sizeof(int32_t));
      q += (int64_t)perInts[t].size();
    }
    *ints_out = io;
  }
  work[9] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tc0).count();
  return 0;
}

void qo_free(void* p) { free(p); }

// small helpers exposed for unit tests of the codec / searcher
uint64_t qo_kmer_encode(const char* s, int64_t avail, int k, int* valid) {
  uint64_t w; *valid = kmerFromChars(s, avail, k, w) ? 1 : 0; return w;
}
uint64_t qo_kmer_rc(uint64_t w, int k) { return wordRC(w, k); }
int qo_kmer_homopolymer(uint64_t w, int k) { return isHomopolymer(w, k) ? 1 : 0; }
void qo_reverse_read(const char* s, int64_t len, char* out) {
  std::string o; reverseRead(s, len, o); memcpy(out, o.data(), (size_t)len);
}
void qo_extend_search(void* hidx, int32_t lbIn, int32_t ubIn, int32_t startAt, const char* q, int64_t m,
                      int32_t* out3) {
  Work w; IndexT a, b, c;
  std::tie(a, b, c) = extendSearchNaive(*(const OIndex*)hidx, lbIn, ubIn, startAt, q, m, w);
  out3[0] = (int32_t)a; out3[1] = (int32_t)b; out3[2] = (int32_t)c;
}
int qo_hash_find(void* hidx, uint64_t key, int32_t* lbub) {
  Work w; const SAInterval* it = ((const OIndex*)hidx)->find(key, w);
  if (!it) return 0;
  lbub[0] = (int32_t)it->lb; lbub[1] = (int32_t)it->ub; return 1;
}
uint64_t qo_rank(void* hidx, uint64_t p) { Work w; return ((const OIndex*)hidx)->rank(p, w); }
// the ksw2 extension kernel alone (nt4 codes in, max(mqe, mte) out) -- for differential tests of the device kernels
int qo_ksw_extz2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int a, int b, int q, int e, int w) {
  int8_t mat[25];
  a = a < 0 ? -a : a; b = b > 0 ? -b : b;
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[i * 5 + j] = (int8_t)(i == j ? a : b); mat[i * 5 + 4] = 0; }
  for (int j = 0; j < 5; ++j) mat[20 + j] = 0;
  KswOut ez = kswExtz2(qlen, query, tlen, target, 5, mat, (int8_t)q, (int8_t)e, w, nullptr);
  return ez.mqe > ez.mte ? ez.mqe : ez.mte;
}

}  // extern "C"
