// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE.  C entry points over the pieces of the REFERENCE that compile from their own
// source files without anything this image lacks (oracle/Makefile.ref builds them where they lie under /root/reference, into
// oracle/_ref/libqm_ref.so; nothing of the reference is copied into this repository).  The hot path's templates
// (SACollector / SASearcher / HitManager) all include the un-vendored cereal headers and are NOT among them; what is:
//   ksw2pp::KSW2Aligner + ksw_extz2_sse{2,41}   src/ksw2pp/{KSW2Aligner.cpp,ksw2_extz2_sse.c,ksw2_extz.c,kalloc.c}   (row a17)
//   rapmap's Kmer<32,1> codec                    include/Kmer.hpp                                                   (row a1)
//   boomphf::mphf (load + lookup)                include/BooPHF.hpp                                                 (row a3)
//   rank9b + BIT_ARRAY                           src/rank9b.cpp, src/bit_array.c                                    (row a8)
//   spp::sparse_hash_map unserialize + find      include/sparsepp/spp.h, include/SparseHashSerializer.hpp           (row a2: hash.bin)
//   XXH64                                        src/xxhash.c                                                       (row a2: KmerKeyHasher)
//   fastx_parser::FastxParser<ReadPair|ReadSeq>  src/FastxParser.cpp, include/FastxParser.hpp, include/kseq.h       (row f3: read ingest)
// The oracle's restatements of exactly these functions are checked against them in tests/test_oracle_ref.py.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "ksw2pp/KSW2Aligner.hpp"
#include "Kmer.hpp"
#include "BooPHF.hpp"
#include "rank9b.h"
#include "sparsepp/spp.h"
#include "SparseHashSerializer.hpp"
#include "xxhash.h"
#include "FastxParser.hpp"
extern "C" {
#include "bit_array.h"
}

extern "C" {

// getAlnScore's aligner call (include/SelectiveAlignmentUtils.hpp:316-359) with the configuration of
// src/RapMapSAMapper.cpp:198-208: score-only extension alignment, max(mqe, mte)
int ref_ksw_extension(const char* query, int qlen, const char* target, int tlen, int match, int mismatch, int gapo, int gape, int bandwidth) {
  ksw2pp::KSW2Config config;
  config.dropoff = -1;
  config.gapo = (int8_t)gapo;
  config.gape = (int8_t)gape;
  config.bandwidth = bandwidth;
  config.flag = 0;
  config.flag |= KSW_EZ_SCORE_ONLY;
  ksw2pp::KSW2Aligner aligner((int8_t)match, (int8_t)mismatch);
  aligner.config() = config;
  ksw_extz_t ez;
  memset(&ez, 0, sizeof(ksw_extz_t));
  ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
  ez.max = 0, ez.mqe = ez.mte = KSW_NEG_INF;
  ez.n_cigar = 0;
  aligner(query, qlen, target, tlen, &ez, ksw2pp::EnumToType<ksw2pp::KSW2AlignmentType::EXTENSION>());
  return ez.mqe > ez.mte ? ez.mqe : ez.mte;
}

// Kmer<32,1>: fromChars (returns whether all k characters were valid), the word, its reverse complement, isHomoPolymer
using ref_mer = combinelib::kmers::Kmer<32, 1>;
int ref_kmer(const char* s, int k, uint64_t* word, uint64_t* rc, int* homopolymer) {
  ref_mer::k(k);
  ref_mer m;
  const bool ok = m.fromChars(s);
  *word = m.word(0);
  *rc = m.getRC().word(0);
  *homopolymer = m.isHomoPolymer() ? 1 : 0;
  return ok ? 1 : 0;
}

// boomphf::mphf<uint64_t, SingleHashFunctor<uint64_t>> as FrugalBooMap instantiates it (include/FrugalBooMap.hpp:90-91)
using ref_mphf = boomphf::mphf<uint64_t, boomphf::SingleHashFunctor<uint64_t>>;
void* ref_mphf_load(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return nullptr;
  ref_mphf* p = new ref_mphf();
  p->load(f);
  return p;
}
void ref_mphf_lookup(void* h, const uint64_t* keys, int64_t n, uint64_t* out) {
  ref_mphf* p = (ref_mphf*)h;
  for (int64_t i = 0; i < n; ++i) out[i] = p->lookup(keys[i]);
}
void ref_mphf_free(void* h) { delete (ref_mphf*)h; }

// rank9b over the bits of an rsd.bin image (u64 nbits + bytes): as RapMapSAIndex::load builds it (src/RapMapSAIndex.cpp:118-131)
struct RefRank { BIT_ARRAY* ba; rank9b* r; };
void* ref_rank_create(const uint8_t* bytes, uint64_t nbits) {
  RefRank* x = new RefRank();
  x->ba = bit_array_create(nbits);
  for (uint64_t i = 0; i < nbits; ++i) if ((bytes[i >> 3] >> (i & 7)) & 1) bit_array_set_bit(x->ba, i);
  x->r = new rank9b(x->ba->words, nbits);
  return x;
}
void ref_rank_query(void* h, const uint64_t* pos, int64_t n, uint64_t* out) {
  RefRank* x = (RefRank*)h;
  for (int64_t i = 0; i < n; ++i) out[i] = x->r->rank(pos[i]);
}
void ref_rank_free(void* h) { RefRank* x = (RefRank*)h; delete x->r; bit_array_free(x->ba); delete x; }


// The dense k-mer hash as RapMapSAIndex::load reads it (src/RapMapSAIndex.cpp:67-76): RegHashT<uint64_t, SAInterval<int32_t>,
// KmerKeyHasher> = spp::sparse_hash_map (include/RapMapUtils.hpp:67) unserialized with pod_hash_serializer.  The value and
// hasher types live in include/RapMapUtils.hpp, which includes the un-vendored cereal; they are two trivial definitions --
// a pair of int32 and "XXH64 of the 8 key bytes, seed 0" (:180-205, :236-238) -- repeated here member for member.  The
// container, its (un)serialisation, its probing and XXH64 are the reference's code.
struct RefSAInterval { int32_t begin_; int32_t end_; };
struct RefKmerKeyHasher { size_t operator()(const uint64_t& m) const { return XXH64(static_cast<void*>(const_cast<uint64_t*>(&m)), sizeof(m), 0); } };
using RefDenseHash = spp::sparse_hash_map<uint64_t, RefSAInterval, RefKmerKeyHasher>;
void* ref_spp_load(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return nullptr;
  RefDenseHash* h = new RefDenseHash();
  if (!h->unserialize(spp_utils::pod_hash_serializer<uint64_t, RefSAInterval>(), &f)) { delete h; return nullptr; }
  return h;
}
int64_t ref_spp_size(void* h) { return (int64_t)((RefDenseHash*)h)->size(); }
// iteration order of the container (= file order of the records)
void ref_spp_dump(void* h, uint64_t* keys, int32_t* lb, int32_t* ub) {
  int64_t i = 0;
  for (auto& kv : *(RefDenseHash*)h) { keys[i] = kv.first; lb[i] = kv.second.begin_; ub[i] = kv.second.end_; ++i; }
}
// khash.find(key) for n keys: found[i], and the interval when found
void ref_spp_find(void* h, const uint64_t* keys, int64_t n, uint8_t* found, int32_t* lb, int32_t* ub) {
  RefDenseHash* m = (RefDenseHash*)h;
  for (int64_t i = 0; i < n; ++i) {
    auto it = m->find(keys[i]);
    found[i] = it != m->end();
    if (found[i]) { lb[i] = it->second.begin_; ub[i] = it->second.end_; }
  }
}
void ref_spp_free(void* h) { delete (RefDenseHash*)h; }
uint64_t ref_xxh64(const void* p, uint64_t len, uint64_t seed) { return XXH64(p, (size_t)len, seed); }

// FrugalBooMap::overflow_ of a BigSA index (include/FrugalBooMap.hpp:318 with IndexT = int64_t: default hasher spp_hash<int64_t>),
// unserialised from a file that holds just the map (the tail of hash_info.val, :244) and asked for n keys
void* ref_spp64_load(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return nullptr;
  auto* h = new spp::sparse_hash_map<int64_t, int64_t>();
  if (!h->unserialize(spp_utils::pod_hash_serializer<int64_t, int64_t>(), &f)) { delete h; return nullptr; }
  return h;
}
int64_t ref_spp64_size(void* h) { return (int64_t)((spp::sparse_hash_map<int64_t, int64_t>*)h)->size(); }
void ref_spp64_find(void* h, const int64_t* keys, int64_t n, uint8_t* found, int64_t* val) {
  auto* m = (spp::sparse_hash_map<int64_t, int64_t>*)h;
  for (int64_t i = 0; i < n; ++i) { auto it = m->find(keys[i]); found[i] = it != m->end(); if (found[i]) val[i] = it->second; }
}
void ref_spp64_free(void* h) { delete (spp::sparse_hash_map<int64_t, int64_t>*)h; }


// The reference's read parser as processReadsPairSA / processReadsSingleSA drive it (src/RapMapSAMapper.cpp:853,869-871 and
// :461-463: getReadGroup, refill, iterate), one parsing thread, one consumer: every record as "name<TAB>seq" ("<TAB>name2<TAB>seq2"
// for pairs) and a newline, in file order, into one malloc'd buffer.  (Names are kseq's: the header up to the first blank.)
int64_t ref_fastx_dump(const char* path1, const char* path2, char** out) {
  std::string buf;
  if (path2) {
    std::vector<std::string> f1{path1}, f2{path2};
    fastx_parser::FastxParser<fastx_parser::ReadPair> parser(f1, f2, 1, 1);
    parser.start();
    auto rg = parser.getReadGroup();
    while (parser.refill(rg))
      for (auto& rp : rg) { buf += rp.first.name; buf += '\t'; buf += rp.first.seq; buf += '\t'; buf += rp.second.name; buf += '\t'; buf += rp.second.seq; buf += '\n'; }
    parser.stop();
  } else {
    std::vector<std::string> f1{path1};
    fastx_parser::FastxParser<fastx_parser::ReadSeq> parser(f1, 1, 1);
    parser.start();
    auto rg = parser.getReadGroup();
    while (parser.refill(rg))
      for (auto& r : rg) { buf += r.name; buf += '\t'; buf += r.seq; buf += '\n'; }
    parser.stop();
  }
  *out = (char*)malloc(buf.size() + 1);
  memcpy(*out, buf.data(), buf.size()); (*out)[buf.size()] = 0;
  return (int64_t)buf.size();
}
void ref_free(void* p) { free(p); }

}  // extern "C"
