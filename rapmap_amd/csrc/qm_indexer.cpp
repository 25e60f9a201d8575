// qm_indexer.cpp -- `rapmap quasiindex [-p]` for the int32 index variants (dense hash and perfect hash), host only.
//
// Produces the on-disk "q5" quasi-index that qm_index_open() mmaps:
//   header.json  (include/IndexHeader.hpp:45-57)       sa.bin      (src/RapMapSAIndexer.cpp:109-110,242-243)
//   txpInfo.bin  (src/RapMapSAIndexer.cpp:705-731)     rsd.bin     (:694-703, src/bit_array.c:2951-2989)
//   hash.bin     (:433-441; container layout of include/sparsepp/spp.h:2355-2366,2420-2429 so that the
//                 reference can unserialize it in place: slot = XXH64(key) & (size-1), triangular probing)
// Text rules restate indexTranscriptsSA (:449-735): isprint filter, upper-casing, pseudo-random
// replacement of non-ACGT (std::default_random_engine(271828) + uniform_int_distribution<>(0,3)),
// poly-A clipping, duplicate removal keyed on XXH64 of the raw sequence, '$' after every transcript.
// The suffix array is built with our own SA-IS (the reference links libdivsufsort; the suffix
// array of a text is unique, so the file is identical), the k-mer -> SA-interval map is the run
// structure of the k-prefixes of the sorted suffixes (buildHash, :262-443).
// The four SHA digests of header.json are informational (nothing reads them back); they are
// written as empty strings.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <zlib.h>

#include "qmap_mi355.h"

namespace {

// ---------------------------------------------------------------- XXH64 (public algorithm, own code)
const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
               P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t xround(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
inline uint64_t xmerge(uint64_t acc, uint64_t v) { return (acc ^ xround(0, v)) * P1 + P4; }
uint64_t xxh64(const void* data, size_t len, uint64_t seed) {
  const uint8_t* p = (const uint8_t*)data; const uint8_t* end = p + len;
  uint64_t h;
  if (len >= 32) {
    uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    const uint8_t* lim = end - 32;
    do { v1 = xround(v1, rd64(p)); v2 = xround(v2, rd64(p + 8)); v3 = xround(v3, rd64(p + 16)); v4 = xround(v4, rd64(p + 24)); p += 32; } while (p <= lim);
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
  } else {
    h = seed + P5;
  }
  h += (uint64_t)len;
  while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
  if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
  while (p < end) { h ^= (*p) * P5; h = rotl(h, 11) * P1; ++p; }
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

// ---------------------------------------------------------------- SA-IS (Nong, Zhang & Chan 2009)
template <typename S, typename I>
void getBuckets(const S* s, I* bkt, I n, I K, bool end) {
  std::fill(bkt, bkt + K, 0);
  for (I i = 0; i < n; ++i) ++bkt[s[i]];
  I sum = 0;
  for (I i = 0; i < K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
}
template <typename S, typename I>
void induceL(const uint8_t* t, I* SA, const S* s, I* bkt, I n, I K) {
  getBuckets(s, bkt, n, K, false);
  for (I i = 0; i < n; ++i) { I j = SA[i] - 1; if (j >= 0 && !t[j]) SA[bkt[s[j]]++] = j; }
}
template <typename S, typename I>
void induceS(const uint8_t* t, I* SA, const S* s, I* bkt, I n, I K) {
  getBuckets(s, bkt, n, K, true);
  for (I i = n - 1; i >= 0; --i) { I j = SA[i] - 1; if (j >= 0 && t[j]) SA[--bkt[s[j]]] = j; }
}
// s[n-1] must be the unique smallest character; I = int32_t, or int64_t for a text that needs the reference's 64-bit
// suffix array (src/RapMapSAIndexer.cpp:743-750)
template <typename S, typename I>
void sais(const S* s, I* SA, I n, I K) {
  std::vector<uint8_t> tv((size_t)n);
  uint8_t* t = tv.data();       // 1 = S-type
  t[n - 1] = 1;
  if (n >= 2) t[n - 2] = 0;
  for (I i = n - 3; i >= 0; --i) t[i] = (s[i] < s[i + 1] || (s[i] == s[i + 1] && t[i + 1])) ? 1 : 0;
  auto isLMS = [&](I i) { return i > 0 && t[i] && !t[i - 1]; };
  std::vector<I> bktv((size_t)K);
  I* bkt = bktv.data();
  getBuckets(s, bkt, n, K, true);
  std::fill(SA, SA + n, -1);
  for (I i = 1; i < n; ++i) if (isLMS(i)) SA[--bkt[s[i]]] = i;
  induceL(t, SA, s, bkt, n, K);
  induceS(t, SA, s, bkt, n, K);
  I n1 = 0;
  for (I i = 0; i < n; ++i) if (isLMS(SA[i])) SA[n1++] = SA[i];
  std::fill(SA + n1, SA + n, -1);
  I name = 0, prev = -1;
  for (I i = 0; i < n1; ++i) {
    I pos = SA[i];
    bool diff = false;
    for (I d = 0; d < n; ++d) {
      if (prev == -1 || s[pos + d] != s[prev + d] || t[pos + d] != t[prev + d]) { diff = true; break; }
      else if (d > 0 && (isLMS(pos + d) || isLMS(prev + d))) break;
    }
    if (diff) { ++name; prev = pos; }
    SA[n1 + pos / 2] = name - 1;
  }
  for (I i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];
  I* SA1 = SA; I* s1 = SA + n - n1;
  if (name < n1) sais<I, I>(s1, SA1, n1, name);
  else for (I i = 0; i < n1; ++i) SA1[s1[i]] = i;
  getBuckets(s, bkt, n, K, true);
  for (I i = 1, j = 0; i < n; ++i) if (isLMS(i)) s1[j++] = i;
  for (I i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
  std::fill(SA + n1, SA + n, -1);
  for (I i = n1 - 1; i >= 0; --i) { I j = SA[i]; SA[i] = -1; SA[--bkt[s[j]]] = j; }
  induceL(t, SA, s, bkt, n, K);
  induceS(t, SA, s, bkt, n, K);
}

struct KmerRun { uint64_t key; int64_t lb, ub; };

int8_t g_code[256];
struct CodeInit { CodeInit() { memset(g_code, -1, 256); g_code['A'] = g_code['a'] = 0; g_code['C'] = g_code['c'] = 1; g_code['G'] = g_code['g'] = 2; g_code['T'] = g_code['t'] = 3; } } g_codeInit;

// 2-bit word of text[p, p+k) or false when it runs off the text / crosses a '$'
inline bool kmerAt(const std::string& text, int64_t p, int k, uint64_t& w) {
  if (p + k > (int64_t)text.size()) return false;
  w = 0;
  for (int i = 0; i < k; ++i) { int c = g_code[(uint8_t)text[p + i]]; if (c < 0) return false; w = (w << 2) | (uint64_t)c; }
  return true;
}

bool writeAll(const std::string& path, const void* a, size_t na, const void* b = nullptr, size_t nb = 0,
              const void* c = nullptr, size_t nc = 0) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = true;
  if (na) ok &= fwrite(a, 1, na, f) == na;
  if (nb) ok &= fwrite(b, 1, nb, f) == nb;
  if (nc) ok &= fwrite(c, 1, nc, f) == nc;
  ok &= fclose(f) == 0;
  return ok;
}

// ---------------------------------------------------------------- BBHash / BooPHF (`quasiindex -p`)
// Restates the construction of boomphf::mphf (include/BooPHF.hpp:891-963,1285-1316,1353-1366,1036-1165) with
// gamma = 2, 25 levels, SingleHashFunctor hash64 (:394-407), xorshift128* continuation (:493-499), fastrange64
// (:815-820): level i keeps the keys that hash alone to a bit of its array, the rest cascade; bitVector ranks
// every 512 bits (:741-754).  The assignment is order-independent, so the file equals the reference's.
inline uint64_t boo_hash64(uint64_t key, uint64_t seed) {
  uint64_t hash = seed;
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
inline uint64_t fastrange64(uint64_t word, uint64_t p) { return (uint64_t)(((__uint128_t)word * (__uint128_t)p) >> 64); }
// hash of `key` for level `lvl` (0-based), replaying the generator
inline uint64_t boo_level_hash(uint64_t key, int lvl) {
  uint64_t s0 = boo_hash64(key, 0xAAAAAAAA55555555ULL);
  if (lvl == 0) return s0;
  uint64_t s1 = boo_hash64(key, 0x33333333CCCCCCCCULL);
  if (lvl == 1) return s1;
  uint64_t a = s0, b = s1, h = 0;      // s[0] = a, s[1] = b
  for (int i = 2; i <= lvl; ++i) {
    uint64_t x = a; const uint64_t y = b;
    a = y;
    x ^= x << 23;
    b = x ^ y ^ (x >> 17) ^ (y >> 26);
    h = b + y;
  }
  return h;
}

struct BooLevel { uint64_t domain; std::vector<uint64_t> words; std::vector<uint64_t> ranks; };

// spp_hash<int64_t> (include/sparsepp/spp_utils.h:189-198,272-275: Thomas Wang's 64-bit mix), the hasher of a BigSA index's
// overflow_ map; spp_hash<int32_t> is the identity
inline uint64_t spp_mix_64(uint64_t a) {
  a = (~a) + (a << 21); a = a ^ (a >> 24); a = (a + (a << 3)) + (a << 8); a = a ^ (a >> 14);
  a = (a + (a << 2)) + (a << 4); a = a ^ (a >> 28); a = a + (a << 31);
  return a;
}

// big: IndexT = int64_t (data_ and overflow_ hold 8-byte values)
int writePerfectHash(const std::string& outDir, const std::vector<KmerRun>& runs, int n_threads, bool big) {
  const uint64_t n = runs.size();
  const double gamma = 2.0;
  const int nb_levels = 25;
  const double proba = 1.0 - pow(((gamma * (double)n - 1) / (gamma * (double)n)), (double)n - 1);   // BooPHF.hpp:1285
  const uint64_t hash_domain = (size_t)(ceil(double(n) * gamma));
  std::vector<BooLevel> lv(nb_levels);
  for (int i = 0; i < nb_levels; ++i) {
    uint64_t d = (((uint64_t)(hash_domain * pow(proba, i)) + 63) / 64) * 64;
    if (d == 0) d = 64;
    lv[i].domain = d;
    lv[i].words.assign(1 + d / 64, 0);            // bitVector(n): _nchar = 1 + n/64 (:568-572)
  }
  std::vector<uint64_t> remaining(n);
  for (uint64_t i = 0; i < n; ++i) remaining[i] = runs[i].key;
  std::vector<std::pair<uint64_t, uint64_t>> finalHash;
  uint64_t offset = 0;
  for (int i = 0; i < nb_levels; ++i) {
    BooLevel& L = lv[i];
    if (i == nb_levels - 1) {                     // cascade stops: exact map (:1101-1110)
      for (uint64_t j = 0; j < remaining.size(); ++j) finalHash.emplace_back(remaining[j], j);
    } else if (!remaining.empty()) {
      std::vector<uint64_t> coll(L.words.size(), 0);
      std::vector<uint64_t> pos(remaining.size());
      for (size_t j = 0; j < remaining.size(); ++j) {
        uint64_t p = fastrange64(boo_level_hash(remaining[j], i), L.domain);
        pos[j] = p;
        uint64_t bit = 1ULL << (p & 63);
        if (L.words[p >> 6] & bit) coll[p >> 6] |= bit; else L.words[p >> 6] |= bit;
      }
      for (size_t w = 0; w < L.words.size(); ++w) L.words[w] &= ~coll[w];   // clearCollisions (:652-663)
      std::vector<uint64_t> next;
      for (size_t j = 0; j < remaining.size(); ++j)
        if (!((L.words[pos[j] >> 6] >> (pos[j] & 63)) & 1)) next.push_back(remaining[j]);
      remaining.swap(next);
    }
    // build_ranks(offset) (:741-754)
    uint64_t cur = offset;
    for (size_t w = 0; w < L.words.size(); ++w) {
      if (((w * 64) % 512) == 0) L.ranks.push_back(cur);
      cur += (uint64_t)__builtin_popcountll(L.words[w]);
    }
    offset = cur;
  }
  const uint64_t lastbitsetrank = offset;
  auto lookup = [&](uint64_t key) -> uint64_t {   // mphf::lookup (:971-1009)
    for (int i = 0; i < nb_levels - 1; ++i) {
      uint64_t p = fastrange64(boo_level_hash(key, i), lv[i].domain);
      if ((lv[i].words[p >> 6] >> (p & 63)) & 1) {
        uint64_t block = p / 512, r = lv[i].ranks[block];
        for (uint64_t w = block * 8; w < (p >> 6); ++w) r += (uint64_t)__builtin_popcountll(lv[i].words[w]);
        r += (uint64_t)__builtin_popcountll(lv[i].words[p >> 6] & ((1ULL << (p & 63)) - 1));
        return r;
      }
    }
    for (auto& kv : finalHash) if (kv.first == key) return kv.second + lastbitsetrank;
    return ~0ULL;
  };
  // values in MPHF order (FrugalBooMap::add + reorder_fn_, FrugalBooMap.hpp:101-112,283-305)
  std::vector<int64_t> data(n); std::vector<uint8_t> lens(n);
  std::vector<std::pair<int64_t, int64_t>> overflow;
  {
    auto work = [&](int t) {
      for (uint64_t j = n * t / n_threads; j < n * (t + 1) / n_threads; ++j) {
        uint64_t idx = lookup(runs[j].key);
        int64_t l = runs[j].ub - runs[j].lb;
        data[idx] = runs[j].lb;
        lens[idx] = l >= 255 ? 255 : (uint8_t)l;
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (uint64_t j = 0; j < n; ++j) if (runs[j].ub - runs[j].lb >= 255) overflow.emplace_back(runs[j].lb, runs[j].ub - runs[j].lb);
  }
  {  // hash_info.bph (mphf::save, :1172-1197)
    FILE* o = fopen((outDir + "hash_info.bph").c_str(), "wb");
    if (!o) return -1;
    int32_t nl = nb_levels;
    fwrite(&gamma, 8, 1, o); fwrite(&nl, 4, 1, o); fwrite(&lastbitsetrank, 8, 1, o); fwrite(&n, 8, 1, o);
    for (auto& L : lv) {
      uint64_t size = L.domain, nchar = L.words.size(), nr = L.ranks.size();
      fwrite(&size, 8, 1, o); fwrite(&nchar, 8, 1, o); fwrite(L.words.data(), 8, nchar, o);
      fwrite(&nr, 8, 1, o); fwrite(L.ranks.data(), 8, nr, o);
    }
    uint64_t fn = finalHash.size(); fwrite(&fn, 8, 1, o);
    for (auto& kv : finalHash) { fwrite(&kv.first, 8, 1, o); fwrite(&kv.second, 8, 1, o); }
    if (fclose(o) != 0) return -1;
  }
  {  // hash_info.val (FrugalBooMap::save, FrugalBooMap.hpp:199-213); overflow_ in sparsepp's layout with
     // spp_hash<int> == identity (include/sparsepp/spp_utils.h) and triangular probing
    FILE* o = fopen((outDir + "hash_info.val").c_str(), "wb");
    if (!o) return -1;
    const size_t isz = big ? 8 : 4;
    uint64_t c = n; fwrite(&c, 8, 1, o);
    if (big) fwrite(data.data(), 8, n, o);
    else { std::vector<int32_t> d32(data.begin(), data.end()); fwrite(d32.data(), 4, n, o); }
    fwrite(&c, 8, 1, o); fwrite(lens.data(), 1, n, o);
    uint64_t tsize = 32;
    while (overflow.size() * 2 > tsize) tsize <<= 1;
    std::vector<int64_t> slot(tsize, -1);
    for (size_t r = 0; r < overflow.size(); ++r) {
      uint64_t pos = (big ? spp_mix_64((uint64_t)overflow[r].first) : (uint64_t)(size_t)(int32_t)overflow[r].first) & (tsize - 1), probes = 0;
      while (slot[pos] >= 0) { ++probes; pos = (pos + probes) & (tsize - 1); }
      slot[pos] = (int64_t)r;
    }
    auto be = [&](uint64_t x, int nb) { for (int i = nb - 1; i >= 0; --i) fputc((int)((x >> (8 * i)) & 0xff), o); };
    be(0x24687531ULL, 4); be(tsize, 4); be((uint64_t)overflow.size(), 4);
    std::vector<uint32_t> bm(tsize / 32, 0);
    for (uint64_t p = 0; p < tsize; ++p) if (slot[p] >= 0) bm[p >> 5] |= (1u << (p & 31));
    fwrite(bm.data(), 4, bm.size(), o);
    for (uint64_t p = 0; p < tsize; ++p) if (slot[p] >= 0) { fwrite(&overflow[slot[p]].first, isz, 1, o); fwrite(&overflow[slot[p]].second, isz, 1, o); }
    if (fclose(o) != 0) return -1;
  }
  return 0;
}

thread_local char g_ierr[256];

}  // namespace

extern "C" const char* qm_indexer_last_error(void) { return g_ierr; }
// the indexer's XXH64 (hash.bin slot placement = KmerKeyHasher, include/RapMapUtils.hpp:236-238; duplicate removal): exported so
// that tests can hold it against the reference's src/xxhash.c
extern "C" uint64_t qm_xxh64(const void* data, uint64_t len, uint64_t seed) { return xxh64(data, (size_t)len, seed); }

extern "C" int qm_build_index_ex(const char* fasta_path, const char* out_dir_c, int32_t k, int32_t no_clip_poly_a,
                                 int32_t keep_duplicates, int32_t n_threads, int32_t perfect_hash, const char* header_sep);
extern "C" int qm_build_index(const char* fasta_path, const char* out_dir_c, int32_t k, int32_t no_clip_poly_a,
                              int32_t keep_duplicates, int32_t n_threads, int32_t perfect_hash) {
  return qm_build_index_ex(fasta_path, out_dir_c, k, no_clip_poly_a, keep_duplicates, n_threads, perfect_hash, nullptr);
}
extern "C" int qm_build_index_ex(const char* fasta_path, const char* out_dir_c, int32_t k, int32_t no_clip_poly_a,
                                 int32_t keep_duplicates, int32_t n_threads, int32_t perfect_hash, const char* header_sep) {
  auto fail = [&](int code, const char* msg) { snprintf(g_ierr, sizeof(g_ierr), "%s", msg); return code; };
  if (!fasta_path || !out_dir_c) return fail(QM_E_ARG, "null path");
  if (k < 1 || k > 31 || (k % 2) == 0) return fail(QM_E_ARG, "k must be odd and <= 31 (RapMapSAIndexer.cpp:870-877)");
  if (n_threads < 1) n_threads = 1;
  std::string outDir(out_dir_c);
  if (outDir.empty() || outDir.back() != '/') outDir += '/';
  mkdir(outDir.c_str(), 0755);

  // ---- step 1: read + transform the transcripts (:500-640)
  // through zlib like the reference's kseq reader (src/FastxParser.cpp:229-328: gzopen / kseq): plain and gzip'd FASTA alike
  gzFile f = gzopen(fasta_path, "rb");
  if (!f) return fail(QM_E_IO, "cannot open FASTA");
  gzbuffer(f, 1 << 20);
  std::default_random_engine eng(271828);
  std::uniform_int_distribution<> dis(0, 3);
  const char bases[] = {'A', 'C', 'G', 'T'};
  const uint32_t polyAClipLength = 10;
  const std::string polyA(polyAClipLength, 'A');
  const std::string sepStr = header_sep ? header_sep : " \t";   // -s / --headerSep (RapMapSAIndexer.cpp:833-835,868): a set of characters, as find_first_of takes it
  struct DupInfo { uint64_t txId, txOffset; uint32_t txLen; };
  std::map<uint64_t, std::vector<DupInfo>> potentialDuplicates;
  std::vector<std::pair<std::string, std::string>> dupNames;   // (retained, dropped)
  std::vector<std::string> names;
  std::vector<int64_t> starts;
  std::vector<uint32_t> completeLens;
  std::vector<uint64_t> onePos;
  std::string text;
  uint32_t n = 0; size_t currIndex = 0;

  std::string recName, readStr;
  bool haveRec = false;
  auto process = [&]() {
    if (!haveRec) return;
    readStr.erase(std::remove_if(readStr.begin(), readStr.end(), [](const char a) { return !isprint((unsigned char)a); }), readStr.end());
    uint32_t readLen = (uint32_t)readStr.size();
    uint32_t completeLen = readLen;
    uint64_t h = xxh64(readStr.data(), readLen, 0);
    for (size_t b = 0; b < readLen; ++b) {
      readStr[b] = (char)::toupper((unsigned char)readStr[b]);
      if (g_code[(uint8_t)readStr[b]] < 0) readStr[b] = bases[dis(eng)];
    }
    if (!no_clip_poly_a) {
      if (readStr.size() > polyAClipLength && readStr.compare(readStr.size() - polyAClipLength, polyAClipLength, polyA) == 0) {
        size_t newEnd = readStr.find_last_not_of("Aa");
        if (newEnd == std::string::npos) readStr.resize(0); else readStr.resize(newEnd + 1);
      }
    }
    readLen = (uint32_t)readStr.size();
    if (readLen == 0) return;
    uint32_t txpIndex = n++;
    std::string processedName = recName.substr(0, recName.find_first_of(sepStr));
    bool didCollide = false;
    auto it = potentialDuplicates.find(h);
    if (it != potentialDuplicates.end()) {
      for (auto& d : it->second) {
        if (readLen == d.txLen && text.compare(d.txOffset, readLen, readStr) == 0) {
          didCollide = true;
          dupNames.emplace_back(names[d.txId], processedName);
        }
      }
    }
    if (!keep_duplicates && didCollide) { --n; return; }
    names.push_back(processedName);
    starts.push_back((int64_t)currIndex);
    completeLens.push_back(completeLen);
    if (!keep_duplicates || !didCollide) potentialDuplicates[h].push_back({txpIndex, currIndex, readLen});
    text += readStr;
    text += '$';
    currIndex += readLen + 1;
    onePos.push_back(currIndex - 1);
  };
  {
    std::vector<char> buf(1 << 20);
    std::string line;
    auto handleLine = [&](std::string& ln) {
      while (!ln.empty() && (ln.back() == '\n' || ln.back() == '\r')) ln.pop_back();
      if (!ln.empty() && ln[0] == '>') {
        process();
        // kseq: name = header up to the first whitespace
        size_t e = 1; while (e < ln.size() && !isspace((unsigned char)ln[e])) ++e;
        recName = ln.substr(1, e - 1); readStr.clear(); haveRec = true;
      } else if (haveRec) {
        readStr += ln;
      }
    };
    while (gzgets(f, buf.data(), (int)buf.size())) {
      line += buf.data();
      if (!line.empty() && line.back() != '\n' && !gzeof(f)) continue;   // long line: keep reading
      handleLine(line);
      line.clear();
    }
    if (!line.empty()) handleLine(line);
    process();
    gzclose(f);
  }
  if (names.empty()) return fail(QM_E_IO, "no transcripts in FASTA");
  const size_t tlen = text.size();
  // a text beyond int32 gets the reference's int64 instantiation (BigSA: 8-byte transcript starts, suffix array entries and
  // interval bounds, src/RapMapSAIndexer.cpp:682-683,711-722,743-765).  QM_FORCE_BIGSA=1 (tests) writes that form for any text.
  const bool big = tlen + 1 > (size_t)0x7fffffff || (getenv("QM_FORCE_BIGSA") && atoi(getenv("QM_FORCE_BIGSA")) != 0);

  {  // duplicate_clusters.tsv (:660-670)
    std::string s = "RetainedTxp\tDuplicateTxp\n";
    for (auto& p : dupNames) { s += p.first; s += '\t'; s += p.second; s += '\n'; }
    writeAll(outDir + "duplicate_clusters.tsv", s.data(), s.size());
  }
  {  // rsd.bin: u64 nbits + ceil(nbits/8) bytes, bit i <=> text[i]=='$'
    uint64_t nbits = tlen; std::vector<uint8_t> bits((nbits + 7) / 8, 0);
    for (uint64_t p : onePos) bits[p >> 3] |= (uint8_t)(1u << (p & 7));
    if (!writeAll(outDir + "rsd.bin", &nbits, 8, bits.data(), bits.size())) return fail(QM_E_IO, "cannot write rsd.bin");
  }
  {  // txpInfo.bin
    FILE* o = fopen((outDir + "txpInfo.bin").c_str(), "wb");
    if (!o) return fail(QM_E_IO, "cannot write txpInfo.bin");
    uint64_t c = names.size(); fwrite(&c, 8, 1, o);
    for (auto& nm : names) { uint64_t l = nm.size(); fwrite(&l, 8, 1, o); fwrite(nm.data(), 1, nm.size(), o); }
    c = starts.size(); fwrite(&c, 8, 1, o);
    if (big) { std::vector<int64_t> st64(starts.begin(), starts.end()); fwrite(st64.data(), 8, st64.size(), o); }
    else { std::vector<int32_t> st32(starts.size()); for (size_t i = 0; i < starts.size(); ++i) st32[i] = (int32_t)starts[i]; fwrite(st32.data(), 4, st32.size(), o); }
    c = tlen; fwrite(&c, 8, 1, o); fwrite(text.data(), 1, tlen, o);
    c = completeLens.size(); fwrite(&c, 8, 1, o); fwrite(completeLens.data(), 4, completeLens.size(), o);
    if (fclose(o) != 0) return fail(QM_E_IO, "cannot write txpInfo.bin");
  }

  // ---- step 2: suffix array (divsufsort in the reference, :99-101,232-234)
  std::vector<int32_t> SA; std::vector<int64_t> SA64;
  {
    // map to {1..5} with a unique 0 sentinel appended ('$' < A < C < G < T as bytes)
    std::vector<uint8_t> s(tlen + 1);
    for (size_t i = 0; i < tlen; ++i) {
      char ch = text[i];
      s[i] = ch == '$' ? 1 : (uint8_t)(2 + g_code[(uint8_t)ch]);
    }
    s[tlen] = 0;
    if (big) {
      SA64.resize(tlen + 1);
      sais<uint8_t, int64_t>(s.data(), SA64.data(), (int64_t)(tlen + 1), (int64_t)6);
      SA64.erase(SA64.begin());              // drop the sentinel suffix
    } else {
      std::vector<int32_t> sa1(tlen + 1);
      sais<uint8_t, int32_t>(s.data(), sa1.data(), (int32_t)(tlen + 1), 6);
      SA.assign(sa1.begin() + 1, sa1.end());
    }
  }
  {
    uint64_t c = tlen;
    if (!(big ? writeAll(outDir + "sa.bin", &c, 8, SA64.data(), SA64.size() * 8) : writeAll(outDir + "sa.bin", &c, 8, SA.data(), SA.size() * 4)))
      return fail(QM_E_IO, "cannot write sa.bin");
  }
  auto saAt = [&](int64_t i) -> int64_t { return big ? SA64[(size_t)i] : (int64_t)SA[(size_t)i]; };

  // ---- step 3: k-mer -> SA interval (:262-443): maximal runs of suffixes sharing a valid k-prefix
  std::vector<std::vector<KmerRun>> parts((size_t)n_threads);
  {
    const int64_t N = (int64_t)tlen;
    auto worker = [&](int t) {
      int64_t b = N * t / n_threads, e = N * (t + 1) / n_threads;
      // snap the chunk start forward to the first run boundary
      uint64_t w, pw;
      if (b > 0) {
        while (b < e) {
          bool v = kmerAt(text, saAt(b), k, w), pv = kmerAt(text, saAt(b - 1), k, pw);
          if (v && pv && w == pw) ++b; else break;
        }
      }
      int64_t i = b;
      auto& out = parts[t];
      while (i < e) {
        if (!kmerAt(text, saAt(i), k, w)) { ++i; continue; }
        int64_t j = i + 1; uint64_t w2;
        while (j < N && kmerAt(text, saAt(j), k, w2) && w2 == w) ++j;   // may run past e: the next chunk snapped
        out.push_back({w, i, j});
        i = j;
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& x : th) x.join();
  }
  size_t K = 0;
  for (auto& p : parts) K += p.size();

  // ---- step 4: the k-mer map on disk
  if (perfect_hash) {
    std::vector<KmerRun> runs; runs.reserve(K);
    for (auto& p : parts) { runs.insert(runs.end(), p.begin(), p.end()); std::vector<KmerRun>().swap(p); }
    int rc = writePerfectHash(outDir, runs, n_threads, big);
    if (rc) return fail(QM_E_IO, "cannot write hash_info.bph / hash_info.val");
  } else {
    uint64_t tsize = 32;
    while (K * 2 > tsize) tsize <<= 1;          // occupancy <= 50 % like spp's resize policy
    const uint64_t mask = tsize - 1;
    std::vector<uint32_t> slotOf(tsize, 0xFFFFFFFFu);    // index into the flattened run list
    std::vector<KmerRun> runs; runs.reserve(K);
    for (auto& p : parts) { runs.insert(runs.end(), p.begin(), p.end()); std::vector<KmerRun>().swap(p); }
    for (size_t r = 0; r < runs.size(); ++r) {
      uint64_t key = runs[r].key;
      uint64_t pos = xxh64(&key, 8, 0) & mask, probes = 0;
      while (slotOf[pos] != 0xFFFFFFFFu) { ++probes; pos = (pos + probes) & mask; }   // JUMP_ = num_probes (spp.h:2498)
      slotOf[pos] = (uint32_t)r;
    }
    FILE* o = fopen((outDir + "hash.bin").c_str(), "wb");
    if (!o) return fail(QM_E_IO, "cannot write hash.bin");
    auto be32or64 = [&](uint64_t v) {
      auto be = [&](uint64_t x, int nb) { for (int i = nb - 1; i >= 0; --i) fputc((int)((x >> (8 * i)) & 0xff), o); };
      if (v < 0xFFFFFFFFULL) be(v, 4); else { be(0xFFFFFFFFULL, 4); be(v, 8); }
    };
    be32or64(0x24687531ULL); be32or64(tsize); be32or64((uint64_t)runs.size());
    std::vector<uint32_t> bitmaps(tsize / 32, 0);
    for (uint64_t p = 0; p < tsize; ++p) if (slotOf[p] != 0xFFFFFFFFu) bitmaps[p >> 5] |= (1u << (p & 31));
    fwrite(bitmaps.data(), 4, bitmaps.size(), o);
    std::vector<uint8_t> rec; rec.reserve(1 << 20);
    for (uint64_t p = 0; p < tsize; ++p) {
      uint32_t r = slotOf[p];
      if (r == 0xFFFFFFFFu) continue;
      uint8_t b[24]; memcpy(b, &runs[r].key, 8);          // {key, SAInterval<IndexT>}: 16 bytes, 24 in a BigSA index
      if (big) { memcpy(b + 8, &runs[r].lb, 8); memcpy(b + 16, &runs[r].ub, 8); }
      else { const int32_t l32 = (int32_t)runs[r].lb, u32 = (int32_t)runs[r].ub; memcpy(b + 8, &l32, 4); memcpy(b + 12, &u32, 4); }
      rec.insert(rec.end(), b, b + (big ? 24 : 16));
      if (rec.size() >= (1 << 20)) { fwrite(rec.data(), 1, rec.size(), o); rec.clear(); }
    }
    if (!rec.empty()) fwrite(rec.data(), 1, rec.size(), o);
    if (fclose(o) != 0) return fail(QM_E_IO, "cannot write hash.bin");
  }
  {  // header.json / refInfo.json
    char js[1024];
    snprintf(js, sizeof(js),
             "{\n    \"value0\": {\n    \"IndexType\": 1,\n    \"IndexVersion\": \"q5\",\n    \"UsesKmers\": true,\n"
             "    \"KmerLen\": %d,\n    \"BigSA\": %s,\n    \"PerfectHash\": %s,\n    \"SeqHash\": \"\",\n"
             "    \"NameHash\": \"\",\n    \"SeqHash512\": \"\",\n    \"NameHash512\": \"\"\n    }\n}", k, big ? "true" : "false", perfect_hash ? "true" : "false");
    if (!writeAll(outDir + "header.json", js, strlen(js))) return fail(QM_E_IO, "cannot write header.json");
    std::string ri = std::string("{\n    \"ReferenceFiles\": [\n        \"") + fasta_path + "\"\n    ]\n}";
    writeAll(outDir + "refInfo.json", ri.data(), ri.size());
  }
  return QM_OK;
}
