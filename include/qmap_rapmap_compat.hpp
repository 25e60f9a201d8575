// qmap_rapmap_compat.hpp -- RapMap's C++ call surface on top of libqmap_mi355.so.
//
// A RapMap / Salmon-style caller drives the quasi-mapping path per read through three entry points
//     SACollector<IndexT>::operator()(read, saSearcher, hcInfo)              include/SACollector.hpp:35-108
//     rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, mateStatus, hcinfo, hits)   include/HitManager.hpp:59-72,130-135
//     rapmap::utils::mergeLeftRightHits / mergeLeftRightHitsFuzzy(...)       include/RapMapUtils.hpp:1185, :864
// in the sequence of src/RapMapSAMapper.cpp:466-531.  This header declares those names -- same namespaces, same argument
// lists, same result types -- so that such a caller compiles against it unchanged, and fills every call from the GPU
// library (include/qmap_mi355.h: qm_collect_reads, qm_hits_to_mappings, qm_merge_lists).  No mapping work is done on the
// host: a call either takes its answer out of a chunk that was mapped in one batched pass (see prefetch below) or runs a
// batch of one on the device.
//
// A GPU wants batches, the reference's loop is per read.  The bridge is ONE added line per chunk of reads:
//
//     while (parser->refill(rg)) {
//       hitCollector.prefetch(rg);                         // <- added: the whole chunk in one fused GPU pass
//       for (auto& rpair : rg) {                           //    everything below is the reference's loop, unchanged
//         bool lh = hitCollector(rpair.first.seq, saSearcher, leftHCInfo);
//         bool rh = hitCollector(rpair.second.seq, saSearcher, rightHCInfo);
//         rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, MateStatus::PAIRED_END_LEFT, leftHCInfo, leftHits);
//         rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, MateStatus::PAIRED_END_RIGHT, rightHCInfo, rightHits);
//         rapmap::utils::mergeLeftRightHits(leftHits, rightHits, jointHits, readLen, maxNumHits, tooManyHits, hctr);
//         ...
//
// prefetch() maps the chunk with qm_map_pairs_stages (collector, hits->mappings and merge of every pair, each stage's
// output kept); the per-read calls recognise their read (by the address of its characters), hand out the stage's output and
// tag it, and the next stage recognises the tag.  Without prefetch, or when a call's input is not what the chunk produced
// (a caller that edits hcInfo or the hit vectors between the calls), the call runs its stage on the device for that one
// read: correct, and as slow as one GPU launch per call must be.
//
// Thread model as in the reference: the index object is shared, SACollector / SASearcher / HitCollectorInfo are per
// thread.  Every host thread gets its own device context; contexts of one index on one device share the index replica.
#ifndef QMAP_RAPMAP_COMPAT_HPP
#define QMAP_RAPMAP_COMPAT_HPP

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <initializer_list>
#include <map>
#include <mutex>
#include <thread>
#include <limits>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "qmap_mi355.h"

namespace qmap {
class Error : public std::runtime_error {
 public:
  Error(int code, const char* what) : std::runtime_error(what), code_(code) {}
  int code() const { return code_; }
 private:
  int code_;
};
namespace detail {
inline void check(int rc) { if (rc) throw Error(rc, qm_last_error()); }
struct Chunk;
inline void drop_services(const qm_index* ix);       // the batching service of an index (below) goes before the index does
}  // namespace detail

// The reference's position lists are chobo::small_vector<int32_t> (include/RapMapUtils.hpp:469-470; static capacity 16): a hit
// that is created per read and per transcript must not cost a heap allocation.  Same idea, for trivially copyable elements:
// the first N live inside the object, more go to the heap.
template <typename T, size_t N = 16>
class small_vector {
  static_assert(std::is_trivially_copyable<T>::value, "qmap::small_vector holds trivially copyable elements");
 public:
  typedef T value_type; typedef T* iterator; typedef const T* const_iterator; typedef size_t size_type;
  typedef T& reference; typedef const T& const_reference;
  small_vector() : p_(inl()), n_(0), cap_(N) {}
  small_vector(const small_vector& o) : p_(inl()), n_(0), cap_(N) { assign(o.begin(), o.end()); }
  small_vector(small_vector&& o) noexcept : p_(inl()), n_(0), cap_(N) { take(o); }
  small_vector(std::initializer_list<T> il) : p_(inl()), n_(0), cap_(N) { assign(il.begin(), il.end()); }
  ~small_vector() { if (p_ != inl()) ::operator delete(p_); }
  small_vector& operator=(const small_vector& o) { if (this != &o) assign(o.begin(), o.end()); return *this; }
  small_vector& operator=(small_vector&& o) noexcept { if (this != &o) { if (p_ != inl()) ::operator delete(p_); p_ = inl(); n_ = 0; cap_ = N; take(o); } return *this; }
  iterator begin() { return p_; } iterator end() { return p_ + n_; }
  const_iterator begin() const { return p_; } const_iterator end() const { return p_ + n_; }
  const_iterator cbegin() const { return p_; } const_iterator cend() const { return p_ + n_; }
  size_t size() const { return n_; } bool empty() const { return n_ == 0; } size_t capacity() const { return cap_; }
  T* data() { return p_; } const T* data() const { return p_; }
  T& operator[](size_t i) { return p_[i]; } const T& operator[](size_t i) const { return p_[i]; }
  T& front() { return p_[0]; } const T& front() const { return p_[0]; }
  T& back() { return p_[n_ - 1]; } const T& back() const { return p_[n_ - 1]; }
  void clear() { n_ = 0; }
  void reserve(size_t c) { if (c > cap_) grow(c); }
  void resize(size_t n, const T& v = T()) { reserve(n); for (size_t i = n_; i < n; ++i) p_[i] = v; n_ = n; }
  void push_back(const T& v) { if (n_ == cap_) grow(2 * cap_); p_[n_++] = v; }
  template <typename... A> T& emplace_back(A&&... a) { if (n_ == cap_) grow(2 * cap_); p_[n_] = T(std::forward<A>(a)...); return p_[n_++]; }
  void pop_back() { --n_; }
  template <typename It> void assign(It first, It last) { n_ = 0; for (; first != last; ++first) push_back(*first); }
  iterator erase(const_iterator pos) { T* q = p_ + (pos - p_); std::memmove(q, q + 1, (size_t)(p_ + n_ - (q + 1)) * sizeof(T)); --n_; return q; }
  iterator insert(const_iterator pos, const T& v) {
    const size_t at = (size_t)(pos - p_);
    if (n_ == cap_) grow(2 * cap_);
    std::memmove(p_ + at + 1, p_ + at, (n_ - at) * sizeof(T)); p_[at] = v; ++n_;
    return p_ + at;
  }
  bool operator==(const small_vector& o) const { return n_ == o.n_ && (n_ == 0 || std::memcmp(p_, o.p_, n_ * sizeof(T)) == 0); }
  bool operator!=(const small_vector& o) const { return !(*this == o); }
 private:
  T* inl() { return reinterpret_cast<T*>(st_); }
  void grow(size_t c) {
    T* q = static_cast<T*>(::operator new(c * sizeof(T)));
    if (n_) std::memcpy(q, p_, n_ * sizeof(T));
    if (p_ != inl()) ::operator delete(p_);
    p_ = q; cap_ = c;
  }
  void take(small_vector& o) {                      // *this is empty and inline
    if (o.p_ != o.inl()) { p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = o.inl(); o.cap_ = N; }
    else { if (o.n_) std::memcpy(p_, o.p_, o.n_ * sizeof(T)); n_ = o.n_; }
    o.n_ = 0;
  }
  T* p_; size_t n_, cap_;
  alignas(T) unsigned char st_[N * sizeof(T)];
};
// A counter many threads bump and few read: every thread adds into a cache line of its own (a slot picked once per thread), a
// read sums the slots.  The interface is the part of std::atomic<uint64_t> that counter code uses.
class sharded_counter {
 public:
  static constexpr int kSlots = 64;
  sharded_counter() { for (auto& s : s_) s.v.store(0, std::memory_order_relaxed); }
  sharded_counter(uint64_t v) : sharded_counter() { s_[0].v.store(v, std::memory_order_relaxed); }
  sharded_counter(const sharded_counter&) = delete;
  sharded_counter& operator=(const sharded_counter&) = delete;
  uint64_t load(std::memory_order = std::memory_order_seq_cst) const {
    uint64_t t = 0; for (const auto& s : s_) t += s.v.load(std::memory_order_relaxed); return t;
  }
  operator uint64_t() const { return load(); }
  void store(uint64_t v, std::memory_order = std::memory_order_seq_cst) {
    for (auto& s : s_) s.v.store(0, std::memory_order_relaxed);
    s_[0].v.store(v, std::memory_order_relaxed);
  }
  uint64_t operator=(uint64_t v) { store(v); return v; }
  uint64_t fetch_add(uint64_t d, std::memory_order = std::memory_order_seq_cst) { const uint64_t before = load(); mine().fetch_add(d, std::memory_order_relaxed); return before; }
  // (no value comes back from ++ / +=: a sum over the slots per increment would undo the point; code that wants one uses fetch_add)
  void operator+=(uint64_t d) { mine().fetch_add(d, std::memory_order_relaxed); }
  void operator++() { mine().fetch_add(1, std::memory_order_relaxed); }
  void operator++(int) { mine().fetch_add(1, std::memory_order_relaxed); }
 private:
  struct alignas(64) Slot { std::atomic<uint64_t> v; };
  static int slot_of_thread() {
    static std::atomic<int> next{0};
    thread_local int mineSlot = next.fetch_add(1, std::memory_order_relaxed) % kSlots;
    return mineSlot;
  }
  std::atomic<uint64_t>& mine() { return s_[slot_of_thread()].v; }
  Slot s_[kSlots];
};
}  // namespace qmap

// ------------------------------------------------------------------------------------------------ rapmap::utils types
namespace rapmap {
namespace utils {

// include/RapMapUtils.hpp:289-295
enum class ChainStatus : uint8_t { PERFECT = 0, UNGAPPED = 1, ALIGNED_ON_LEFT = 2, ALIGNED_ON_RIGHT = 3, REGULAR = 4 };

// include/RapMapUtils.hpp:321-354
class FragmentChainStatus {
 public:
  FragmentChainStatus() : left(static_cast<uint8_t>(ChainStatus::REGULAR)), right(static_cast<uint8_t>(ChainStatus::REGULAR)) {}
  FragmentChainStatus(ChainStatus ls, ChainStatus rs) : left(static_cast<uint8_t>(ls)), right(static_cast<uint8_t>(rs)) {}
  void setLeft(ChainStatus s) { left = static_cast<uint8_t>(s); }
  void setRight(ChainStatus s) { right = static_cast<uint8_t>(s); }
  ChainStatus getLeft() const { return static_cast<ChainStatus>(left); }
  ChainStatus getRight() const { return static_cast<ChainStatus>(right); }
 private:
  uint8_t left : 4, right : 4;
};

// include/RapMapUtils.hpp:356-362
enum class MateStatus : uint8_t {
  SINGLE_END = 0, PAIRED_END_LEFT = 1, PAIRED_END_RIGHT = 2, PAIRED_END_PAIRED = 3, NOTHING = std::numeric_limits<uint8_t>::max()
};

// include/RapMapUtils.hpp:83-89
class MappingConfig {
 public:
  bool consistentHits{false};
  bool doChaining{false};
  float consensusFraction{1.0};
  bool considerMultiPos{false};
};

// include/RapMapUtils.hpp:208-216.  The reference's members are std::atomic<uint64_t> shared by every worker thread and bumped
// several times per read pair -- harmless when a pair costs microseconds of mapping, but here the per-pair host work is ~0.1 us
// and 32 workers incrementing one cache line ran 30x slower than one (profiles/r04/compat_probe.txt).  Same member names and
// the operations callers use on them (++, +=, load(), store(), conversion to uint64_t), counted in per-thread cache lines and
// summed when read.
struct HitCounters {
  qmap::sharded_counter peHits;
  qmap::sharded_counter seHits;
  qmap::sharded_counter trueHits;
  qmap::sharded_counter totHits;
  qmap::sharded_counter numReads;
  qmap::sharded_counter tooManyHits;
  qmap::sharded_counter lastPrint;
};

// include/RapMapUtils.hpp:399-502 (chobo::small_vector<int32_t> -> qmap::small_vector<int32_t>: same inline capacity, no heap
// allocation for the usual one-position hit)
// With RAPMAP_SALMON_SUPPORT defined the type carries what Salmon reads from it (include/RapMapUtils.hpp:41-43,407-421,446-467):
// logProb, logBias, format / libFormat() and fragLengthPedantic().  LibraryFormat is Salmon's own type (its LibraryFormat.hpp),
// which the caller includes before this header, as the reference's header does.
struct QuasiAlignment {
  QuasiAlignment()
      : tid(std::numeric_limits<uint32_t>::max()), pos(std::numeric_limits<int32_t>::max()), fwd(true),
        fragLen(std::numeric_limits<uint32_t>::max()), readLen(std::numeric_limits<uint32_t>::max()), isPaired(false) {}
  QuasiAlignment(uint32_t tidIn, int32_t posIn, bool fwdIn, uint32_t readLenIn, uint32_t fragLenIn = 0, bool isPairedIn = false)
      : tid(tidIn), pos(posIn), fwd(fwdIn), fragLen(fragLenIn), readLen(readLenIn), isPaired(isPairedIn) {}
  inline void setChainScore(double chainScoreIn) { chainScore_ = chainScoreIn; }
  inline double chainScore() const { return chainScore_; }
  inline uint32_t transcriptID() const { return tid; }
  inline double score() const { return score_; }
  inline void score(double scoreIn) { score_ = scoreIn; }
  inline int32_t alnScore() const { return alnScore_; }
  inline void alnScore(int32_t alnScoreIn) { alnScore_ = alnScoreIn; }
  inline uint32_t fragLength() const { return fragLen; }
  inline int32_t hitPos() { return pos < matePos ? pos : matePos; }
#ifdef RAPMAP_SALMON_SUPPORT
  // include/RapMapUtils.hpp:446-467
  inline uint32_t fragLengthPedantic(uint32_t txpLen) const {
    if (mateStatus != MateStatus::PAIRED_END_PAIRED || fwd == mateIsFwd) return 0;
    int32_t p1 = fwd ? pos : matePos;
    const int32_t sTxpLen = static_cast<int32_t>(txpLen);
    p1 = (p1 < 0) ? 0 : p1;
    p1 = (p1 > sTxpLen) ? sTxpLen : p1;
    int32_t p2 = fwd ? static_cast<int32_t>(matePos + mateLen) : static_cast<int32_t>(pos + readLen);
    p2 = (p2 < 0) ? 0 : p2;
    p2 = (p2 > sTxpLen) ? sTxpLen : p2;
    return (p1 > p2) ? p1 - p2 : p2 - p1;
  }
  double logProb{HUGE_VAL};
  double logBias{HUGE_VAL};
  inline LibraryFormat libFormat() { return format; }
  LibraryFormat format{LibraryFormat::formatFromID(0)};       // (both constructors: RapMapUtils.hpp:407-421)
#endif

  bool hasMultiPos{false};
  qmap::small_vector<int32_t> allPositions;
  qmap::small_vector<int32_t> oppositeStrandPositions;
  uint32_t tid;
  int32_t pos;
  int32_t matePos{0};
  bool fwd;
  bool mateIsFwd{true};
  uint32_t fragLen;
  uint32_t readLen;
  uint32_t mateLen{0};
  bool isPaired;
  MateStatus mateStatus{MateStatus::NOTHING};
  double score_{1.0};
  int32_t alnScore_{0};
  FragmentChainStatus chainStatus;
  // The device does not hand chain scores out: a hit that came from it holds lowest() here, as non-chained hits do in the
  // reference.  (mergeOrientationUnique, the one consumer of chainScore, runs inside hitsToMappingsSimple on the device.)
  double chainScore_{std::numeric_limits<double>::lowest()};
  // where this hit came from when it was handed out of a prefetched chunk (not part of the reference's type)
  const qmap::detail::Chunk* qm_chunk_{nullptr};
  uint64_t qm_gen_{0};
  int64_t qm_read_{-1};
  int32_t qm_count_{0};                // on the first hit of a read's list as hitsToMappingsSimple handed it out: how many there were
};

// include/RapMapUtils.hpp:516-525
template <typename OffsetT>
struct SAIntervalHit {
  SAIntervalHit(OffsetT beginIn, OffsetT endIn, uint32_t lenIn, uint32_t queryPosIn, bool queryRCIn)
      : begin(beginIn), end(endIn), len(lenIn), queryPos(queryPosIn), queryRC(queryRCIn) {}
  OffsetT span() { return end - begin; }
  OffsetT begin, end;
  uint32_t len, queryPos;
  bool queryRC;
};

// include/RapMapUtils.hpp:855-862
enum class MergeResult : uint8_t { HAD_NONE, HAD_EMPTY_INTERSECTION, HAD_CONCORDANT, HAD_DISCORDANT, HAD_ONLY_LEFT, HAD_ONLY_RIGHT };

}  // namespace utils

namespace hit_manager {
// include/HitManager.hpp:52-72
template <typename T> using SAIntervalVector = std::vector<T>;
template <typename SAIntervalHitT>
class HitCollectorInfo {
 public:
  void clear() {
    readLen = 0;
    maxDist = 0;
    fwdSAInts.clear();
    rcSAInts.clear();
    qm_chunk_ = nullptr; qm_read_ = -1;
  }
  size_t readLen{0};
  int32_t maxDist{0};
  SAIntervalVector<SAIntervalHitT> fwdSAInts;
  SAIntervalVector<SAIntervalHitT> rcSAInts;
  // set by SACollector::operator() when the intervals came out of a prefetched chunk (not part of the reference's type)
  const qmap::detail::Chunk* qm_chunk_{nullptr};
  uint64_t qm_gen_{0};
  int64_t qm_read_{-1};
};
}  // namespace hit_manager
}  // namespace rapmap

// ------------------------------------------------------------------------------------------------ index
// Hash flavours of the reference's four index instantiations (src/RapMapSAMapper.cpp:1209-1240); tags only: the on-disk
// header says which table the directory holds and the library loads either.
struct RegHashT {};
struct PerfectHashT {};

// RapMapSAIndex<IndexT, HashT> (include/RapMapSAIndex.hpp:47-83): load() + the members a mapping caller reads.
template <typename IndexT, typename HashT>
class RapMapSAIndex {
 public:
  using IndexType = IndexT;
  using HashType = HashT;
  RapMapSAIndex() = default;
  ~RapMapSAIndex() { if (ix_) { qmap::detail::drop_services(ix_); qm_index_close(ix_); } }
  RapMapSAIndex(const RapMapSAIndex&) = delete;
  RapMapSAIndex& operator=(const RapMapSAIndex&) = delete;

  bool load(const std::string& indDir) {
    qmap::detail::check(qm_index_open(indDir.c_str(), &ix_));
    qm_index_info info;
    qmap::detail::check(qm_index_info_get(ix_, &info));
    // the caller picks the instantiation from header.json's BigSA like the reference does (src/RapMapSAMapper.cpp:1209-1240); a
    // mismatch would make the reference misread the files, here it is an error
    if ((info.big_sa != 0) != (sizeof(IndexT) == 8)) throw std::runtime_error("RapMapSAIndex: IndexT does not match the index's BigSA flag");
    k_ = info.k; perfect_ = info.perfect_hash != 0;
    const uint8_t* text = nullptr; int64_t tl = 0; const int32_t* offs = nullptr; int64_t nt = 0;
    qmap::detail::check(qm_index_arrays(ix_, &text, &tl, &offs, &nt));
    seq.assign(reinterpret_cast<const char*>(text), static_cast<size_t>(tl));
    { const uint32_t* uo = reinterpret_cast<const uint32_t*>(offs); txpOffsets.clear(); for (int64_t i = 0; i < nt; ++i) txpOffsets.push_back(static_cast<IndexT>(uo[i])); }   // unsigned on the device path
    txpNames.clear(); txpLens.clear();
    for (int64_t i = 0; i < nt; ++i) { txpNames.emplace_back(qm_index_txp_name(ix_, i)); txpLens.push_back(static_cast<IndexT>(qm_index_txp_len(ix_, i))); }
    return true;
  }
  // GPU the contexts of this index live on (before the first mapping call; default 0)
  void setDevice(int d) { device_ = d; }
  int device() const { return device_; }
  uint32_t k() const { return static_cast<uint32_t>(k_); }
  bool perfectHash() const { return perfect_; }
  const qm_index* handle() const { return ix_; }

  std::string seq;
  std::vector<std::string> txpNames;
  std::vector<IndexT> txpOffsets;
  std::vector<IndexT> txpLens;

 private:
  qm_index* ix_{nullptr};
  int device_{0};
  int k_{31};
  bool perfect_{false};
};
using SAIndex32BitDense = RapMapSAIndex<int32_t, RegHashT>;
using SAIndex32BitPerfect = RapMapSAIndex<int32_t, PerfectHashT>;
using SAIndex64BitDense = RapMapSAIndex<int64_t, RegHashT>;          // BigSA indices (src/HitManager.cpp:889-892)
using SAIndex64BitPerfect = RapMapSAIndex<int64_t, PerfectHashT>;

// ------------------------------------------------------------------------------------------------ plumbing
namespace qmap {
namespace detail {

// One thread's device context per index, for the calls that run a batch of one (contexts are not shared between threads; the
// index replica in HBM is).  Chunks of read groups do not use these: they go through the batching service below.
inline qm_ctx*& last_ctx() { thread_local qm_ctx* c = nullptr; return c; }   // the calling thread's most recent context
struct LastIndex { const qm_index* ix{nullptr}; int device{0}; };
inline LastIndex& last_index() { thread_local LastIndex li; return li; }     // ... and the index its calls were about
struct ThreadCtx {
  const qm_index* ix{nullptr};
  qm_ctx* ctx{nullptr};
  ~ThreadCtx() { if (ctx) qm_ctx_destroy(ctx); }
};
inline qm_ctx* thread_ctx_of(const qm_index* ix, int device) {
  thread_local std::vector<std::unique_ptr<ThreadCtx>> tl;
  for (auto& t : tl) if (t->ix == ix) return last_ctx() = t->ctx;
  std::unique_ptr<ThreadCtx> t(new ThreadCtx());
  t->ix = ix;
  check(qm_ctx_create(ix, device, &t->ctx));
  tl.push_back(std::move(t));
  return last_ctx() = tl.back()->ctx;
}
template <typename RapMapIndexT>
inline qm_ctx* thread_ctx(RapMapIndexT& rmi) { last_index().ix = rmi.handle(); last_index().device = rmi.device(); return thread_ctx_of(rmi.handle(), rmi.device()); }
// the context a merge that has to go to the device runs on: the thread's most recent one, or one for the index its last
// collector call was about
inline qm_ctx* merge_ctx() {
  if (last_ctx()) return last_ctx();
  if (last_index().ix) return thread_ctx_of(last_index().ix, last_index().device);
  return nullptr;
}

// Everything one fused pass over a chunk produced, per read (paired: read 2u = left mate of pair u, 2u + 1 = right).
// page-locked host memory, grown on demand (qm_pinned_alloc: the device's DMA engines read / write it directly)
struct PinnedBuf {
  void* p{nullptr}; size_t cap{0};
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { if (p) qm_pinned_free(p); }
  void* need(size_t bytes, bool keep = false) {
    if (bytes <= cap) return p;
    const size_t nc = (cap ? 2 * bytes : bytes + bytes / 2) + 4096;      // (page-locking costs ~25 ms per 100 MB: a buffer that grows again doubles)
    void* q = qm_pinned_alloc(static_cast<int64_t>(nc));
    if (!q) throw Error(QM_E_NOMEM, "out of page-locked host memory");
    if (keep && p && cap) std::memcpy(q, p, cap);
    if (p) qm_pinned_free(p);
    p = q; cap = nc;
    return p;
  }
};
// ---- the batching service -------------------------------------------------------------------------------------------
// The reference's workers each take a read group of ~10 000 pairs from the parser and map it on their own
// (src/RapMapSAMapper.cpp:752-799,853).  One GPU pass per group and thread is the wrong shape for the device: a pass has a fixed
// cost (a dozen launches, four synchronisations) that dwarfs the 40 us of kernel time 10 000 pairs need, and 32 host threads
// each driving a context of their own spend their time inside the runtime's locks (profiles/r04/compat_probe.txt: 4.9 M
// pairs/s on one thread, 4.4 on 32).  So the groups of all threads are mapped TOGETHER: a worker's prefetch() joins the batch
// that is currently open -- it reserves its place in the batch's page-locked input buffers under a lock, packs its reads there
// side by side with the other workers, and waits -- and a dispatcher thread that owns a device context closes the batch when
// it is free to, maps it in one fused pass (qm_map_pairs_stages), brings every stage's output down in one go
// (qm_fetch_stages) and wakes the workers, which then run the reference's per-read loop on their slice of the shared result.
// While one batch is on the device the next one fills: the batch size adapts to the load by itself (one group when one thread
// asks, dozens when many do).  Two dispatchers (QMAP_COMPAT_CONTEXTS) keep two batches in flight.
struct Batch {
  PinnedBuf in[4];                         // characters and offsets of both mates, filled by the joining workers
  char* s1{nullptr}; char* s2{nullptr}; int64_t* o1{nullptr}; int64_t* o2{nullptr};
  size_t cap1{0}, cap2{0}, used1{0}, used2{0};
  int64_t capUnits{0}, units{0};
  int joined{0}, packed{0}, readers{0};
  int state{0};                            // 0 free, 1 open, 2 closed (packing / on the device), 3 done, 4 failed
  std::chrono::steady_clock::time_point t0;   // when the first group joined
  qm_opts opts{};
  uint32_t stageFlags{0};                  // QM_STAGES_NO_INTERVALS when the groups' collectors said they do not look at the interval records
  // round 6: the same reads 2-bit packed (qm_pack_reads' layout: reads never share a byte, so every worker packs its own slice), the
  // form they cross PCIe in: 26 bytes per 100-bp read instead of 100.  Characters that are not upper-case A C G T are exceptions; a
  // batch whose exceptions outgrow the list travels as characters.
  PinnedBuf pk[2]; uint8_t* p1{nullptr}; uint8_t* p2{nullptr};
  std::vector<qm_pack_exc> exc1, exc2; bool packedOk{true};
  qm_stage_view v{}; PinnedBuf arena;
  int rc{0}; std::string err;
  void size_for(size_t b1, size_t b2, int64_t n) {
    const size_t w1 = b1 + 64 > (size_t(40) << 20) ? b1 + 64 : (size_t(40) << 20), w2 = b2 + 64 > (size_t(40) << 20) ? b2 + 64 : (size_t(40) << 20);
    const int64_t wu = n + 1 > (int64_t(1) << 19) ? n + 1 : (int64_t(1) << 19);
    s1 = static_cast<char*>(in[0].need(w1)); s2 = static_cast<char*>(in[1].need(w2));
    o1 = static_cast<int64_t*>(in[2].need(static_cast<size_t>(wu + 1) * 8)); o2 = static_cast<int64_t*>(in[3].need(static_cast<size_t>(wu + 1) * 8));
    cap1 = in[0].cap - 64; cap2 = in[1].cap - 64; capUnits = static_cast<int64_t>(in[2].cap / 8) - 2;
    if (static_cast<int64_t>(in[3].cap / 8) - 2 < capUnits) capUnits = static_cast<int64_t>(in[3].cap / 8) - 2;
    p1 = static_cast<uint8_t*>(pk[0].need((in[0].cap >> 2) + static_cast<size_t>(capUnits) + 64));
    p2 = static_cast<uint8_t*>(pk[1].need((in[1].cap >> 2) + static_cast<size_t>(capUnits) + 64));
  }
};
class Service {
 public:
  Service(const qm_index* ix, int device) : ix_(ix), device_(device) {
    const char* e = std::getenv("QMAP_COMPAT_CONTEXTS");
    int n = e && std::atoi(e) > 0 ? std::atoi(e) : 2;
    if (n > 8) n = 8;
    const char* l = std::getenv("QMAP_COMPAT_LINGER_US");
    lingerUs_ = l && std::atoi(l) >= 0 ? std::atoi(l) : 300;
    debug_ = std::getenv("QMAP_COMPAT_DEBUG") != nullptr;
    const char* np = std::getenv("QMAP_COMPAT_NO_PACK");
    packUploads_ = !(np && std::atoi(np) != 0);
    for (int i = 0; i < n; ++i) th_.emplace_back([this] { dispatch(); });
  }
  ~Service() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cvWork_.notify_all(); cvDone_.notify_all();
    for (auto& t : th_) t.join();
  }
  // a worker's group joins a batch: its place among the units / characters of the batch
  struct Place { Batch* b; int64_t u0; size_t c1, c2; };
  Place join(const qm_opts& opts, int64_t n, size_t b1, size_t b2, uint32_t stageFlags = 0) {
    std::unique_lock<std::mutex> lk(mu_);
    Batch* b = open_;
    if (b && (std::memcmp(&b->opts, &opts, sizeof(qm_opts)) != 0 || b->stageFlags != stageFlags || b->units + n > b->capUnits || b->used1 + b1 > b->cap1 || b->used2 + b2 > b->cap2)) {
      b->state = 2; ready_.push_back(b); open_ = nullptr; b = nullptr;        // other settings, or full: off it goes
    }
    if (!b) {
      for (auto& x : all_) if (x->state == 0) { b = x.get(); break; }
      if (!b) { all_.emplace_back(new Batch()); b = all_.back().get(); }
      b->size_for(b1, b2, n);
      b->opts = opts; b->stageFlags = stageFlags; b->units = 0; b->used1 = 0; b->used2 = 0; b->joined = 0; b->packed = 0; b->readers = 0; b->rc = 0; b->err.clear();
      b->exc1.clear(); b->exc2.clear(); b->packedOk = packUploads_;
      b->o1[0] = 0; b->o2[0] = 0;
      b->state = 1; open_ = b;
      b->t0 = std::chrono::steady_clock::now();
    }
    Place p{b, b->units, b->used1, b->used2};
    b->units += n; b->used1 += b1; b->used2 += b2; b->joined++; b->readers++;
    lk.unlock();
    cvWork_.notify_all();
    return p;
  }
  // a worker's slice of the batch, 2-bit packed next to the other workers' (place: the group's first unit and first characters); its
  // exceptions join the batch's lists, a slice with too many of them (lower-case reads ...) sends the whole batch as characters
  void pack_slice(Batch* b, int64_t u0, int64_t n, size_t c1, size_t c2, const int64_t* o1, const int64_t* o2) {
    if (!b->packedOk || n <= 0) return;
    qm_pack_exc ex[2][256]; int64_t ne[2] = {0, 0};
    std::vector<int64_t> off(static_cast<size_t>(n) + 1);
    bool ok = true;
    for (int m = 0; m < 2 && ok; ++m) {
      const int64_t* o = m ? o2 : o1;
      off[0] = static_cast<int64_t>(m ? c2 : c1);
      for (int64_t i = 1; i <= n; ++i) off[static_cast<size_t>(i)] = o[i];         // (o[0] belongs to the group in front: this group's start is its place)
      ok = qm_pack_reads(m ? b->s2 : b->s1, off.data(), n, (m ? b->p2 : b->p1) + u0, ex[m], 256, &ne[m]) == QM_OK;
    }
    std::lock_guard<std::mutex> lk(mu_);
    if (!ok) { b->packedOk = false; return; }
    b->exc1.insert(b->exc1.end(), ex[0], ex[0] + ne[0]); b->exc2.insert(b->exc2.end(), ex[1], ex[1] + ne[1]);
  }
  // the worker has packed its reads (it may go on with other work: prefetch_async) ...
  void packed(Batch* b) {
    { std::lock_guard<std::mutex> lk(mu_); b->packed++; }
    cvWork_.notify_all();
  }
  // ... and waits until the batch has been mapped
  void packed_and_wait(Batch* b) { packed(b); wait_done(b); }
  void wait_done(Batch* b) {
    std::unique_lock<std::mutex> lk(mu_);
    cvDone_.wait(lk, [&] { return b->state >= 3 || stop_; });
    if (b->state != 3) {
      const int rc = b->rc ? b->rc : QM_E_STATE; const std::string msg = b->err.empty() ? std::string("the batching service stopped") : b->err;
      release_locked(b);
      throw Error(rc, msg.c_str());
    }
  }
  void release(Batch* b) { std::lock_guard<std::mutex> lk(mu_); release_locked(b); }
  // a collector starts / stops sending its groups here (how many there are tells the dispatchers how long to wait for company)
  void attach() { std::lock_guard<std::mutex> lk(mu_); ++workers_; }
  void detach() { std::lock_guard<std::mutex> lk(mu_); if (workers_ > 0) --workers_; }
 private:
  void release_locked(Batch* b) { if (--b->readers == 0 && b->state >= 3) b->state = 0; }
  void dispatch() {
    qm_ctx* ctx = nullptr;
    std::unique_lock<std::mutex> lk(mu_);
    while (!stop_) {
      Batch* b = nullptr;
      if (!ready_.empty()) { b = ready_.front(); ready_.pop_front(); }
      else if (open_ && open_->joined > 0) {
        // A pass has a fixed cost of about a millisecond whatever it carries, so an idle dispatcher does not run off with the
        // first group that shows up when many workers are at it: it gives the others a moment -- until half of the workers
        // this service has seen are in, or lingerUs_ after the first one came.  One worker alone never waits.
        Batch* o = open_;
        const int want = workers_ > 1 ? (workers_ + 1) / 2 : 1;
        if (o->joined < want) {
          const auto dl = o->t0 + std::chrono::microseconds(lingerUs_);
          if (std::chrono::steady_clock::now() < dl) { cvWork_.wait_until(lk, dl); continue; }
        }
        b = o; open_ = nullptr; b->state = 2;
      }
      if (!b) { cvWork_.wait(lk); continue; }
      cvWork_.wait(lk, [&] { return b->packed == b->joined || stop_; });      // the last joiners are still copying their characters
      if (stop_) { b->state = 4; break; }
      const auto tw = std::chrono::steady_clock::now();
      lk.unlock();
      int rc = 0; std::string err;
      if (!ctx) { rc = qm_ctx_create(ix_, device_, &ctx); if (rc) { err = qm_last_error(); ctx = nullptr; } }
      auto t1 = tw, t2 = tw, t3 = tw;
      if (!rc) {
        int64_t nHits = 0; qm_counters c{};
        // one pass at a time uploads + computes, one at a time brings its results down: passes of the dispatchers that would start
        // together take turns instead, and the upload of one runs under the download of the other (the link is full duplex)
        { std::lock_guard<std::mutex> ph(muMap_);
          if (b->packedOk) rc = qm_map_pairs_stages_packed(ctx, &b->opts, b->units, b->p1, b->o1, b->exc1.empty() ? nullptr : b->exc1.data(), static_cast<int64_t>(b->exc1.size()),
                                                           b->p2, b->o2, b->exc2.empty() ? nullptr : b->exc2.data(), static_cast<int64_t>(b->exc2.size()), b->stageFlags, &nHits, &c);
          else rc = qm_map_pairs_stages_ex(ctx, &b->opts, b->units, b->s1, b->o1, b->s2, b->o2, b->stageFlags, &nHits, &c); }
        t1 = std::chrono::steady_clock::now();
        int64_t need = 0;
        if (!rc) rc = qm_stage_bytes(ctx, &need);
        if (!rc) {
          try { b->arena.need(static_cast<size_t>(need)); } catch (const Error& e) { rc = e.code(); err = e.what(); }
        }
        t2 = std::chrono::steady_clock::now();
        if (!rc) { std::lock_guard<std::mutex> ph(muFetch_); rc = qm_fetch_stages(ctx, b->arena.p, static_cast<int64_t>(b->arena.cap), &b->v); }
        t3 = std::chrono::steady_clock::now();
        if (rc && err.empty()) err = qm_last_error();
      }
      if (debug_) {
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(c - a).count(); };
        std::fprintf(stderr, "[qmap service] batch of %lld pairs in %d groups (%s, %s): waited %lld us for company / packers, map %lld us, arena %lld us, fetch %lld us\n",
                     (long long)b->units, b->joined, b->packedOk ? "2-bit packed" : "characters", (b->stageFlags & QM_STAGES_NO_INTERVALS) ? "no interval records" : "interval records",
                     us(b->t0, tw), us(tw, t1), us(t1, t2), us(t2, t3));
      }
      lk.lock();
      b->rc = rc; b->err = err; b->state = rc ? 4 : 3;
      if (b->readers == 0) b->state = 0;
      cvDone_.notify_all();
    }
    lk.unlock();
    cvDone_.notify_all();
    if (ctx) qm_ctx_destroy(ctx);
  }
  const qm_index* ix_; int device_;
  std::mutex mu_; std::condition_variable cvWork_, cvDone_;
  std::mutex muMap_, muFetch_;
  std::vector<std::unique_ptr<Batch>> all_;
  std::deque<Batch*> ready_;
  Batch* open_{nullptr};
  std::vector<std::thread> th_;
  bool stop_{false};
  int workers_{0}, lingerUs_{300};
  bool debug_{false}, packUploads_{true};
};
// the services of this process, one per (index, device); created on first use, dropped when their index is closed (an index
// that is never closed keeps its service until the process ends: the registry itself is never destroyed)
struct Registry { std::mutex mu; std::map<std::pair<const qm_index*, int>, std::unique_ptr<Service>> m; };
inline Registry& registry() { static Registry* r = new Registry(); return *r; }
inline Service* service_of(const qm_index* ix, int device) {
  Registry& R = registry();
  std::lock_guard<std::mutex> lk(R.mu);
  auto& s = R.m[std::make_pair(ix, device)];
  if (!s) s.reset(new Service(ix, device));
  return s.get();
}
inline void drop_services(const qm_index* ix) {
  Registry& R = registry();
  std::vector<std::unique_ptr<Service>> gone;
  { std::lock_guard<std::mutex> lk(R.mu);
    for (auto it = R.m.begin(); it != R.m.end();) { if (it->first.first == ix) { gone.push_back(std::move(it->second)); it = R.m.erase(it); } else ++it; } }
  gone.clear();                                                // (joins the dispatchers, which destroy their contexts)
}

// A worker's view of its read group inside a mapped batch (paired: read 2u = left mate of pair u, 2u + 1 = right).
struct Chunk {
  uint64_t gen{0};
  bool paired{false};
  qm_opts opts{};                       // what the chunk was mapped with: a stage call with other settings does not use it
  int64_t nreads{0};
  std::vector<const char*> key; std::vector<size_t> keyLen;
  qm_stage_view v{};                    // every stage's output of the BATCH, per read / per pair (absolute CSR offsets) ...
  int64_t rbase{0}, ubase{0};           // ... and where this group's reads / pairs start in it
  Service* svc{nullptr}; Batch* batch{nullptr};
  int64_t cursor{0};
  // the group's reads as they were packed for the device (the batch's input buffers; o1 / o2 start at this group's first pair)
  const char* s1{nullptr}; const char* s2{nullptr}; const int64_t* o1{nullptr}; const int64_t* o2{nullptr};
  int64_t nPacked{0};
  const void* firstRead{nullptr};      // the group a prefetch_async was issued for (prefetch / wait recognise it)
  uint64_t settings{0}; const void* owner{nullptr};   // the collector that sent the group and the state of its settings then
  void release() { if (batch && svc) svc->release(batch); batch = nullptr; nreads = 0; }
  void bind(Service* s) { svc = s; }       // (the service counts COLLECTORS, not chunks: SACollector::count_in -- a collector holds two or three chunk objects)
  ~Chunk() { release(); }
};
inline uint64_t next_gen() { static std::atomic<uint64_t> g{1}; return g++; }

inline void decode_hit(const qm_hit& h, rapmap::utils::QuasiAlignment& q) {
  using namespace rapmap::utils;
  q.tid = h.tid; q.pos = h.pos; q.matePos = h.mate_pos; q.fwd = h.fwd != 0; q.mateIsFwd = h.mate_is_fwd != 0;
  q.fragLen = h.frag_len; q.readLen = h.read_len; q.mateLen = h.mate_len; q.isPaired = h.is_paired != 0;
  q.mateStatus = static_cast<MateStatus>(h.mate_status);
}

// per-read list words (qmap_mi355.h, "List words") -> the vector hitsToMappingsSimple fills
inline void decode_list(const uint64_t* w, int64_t n, bool chained, uint32_t readLen, rapmap::utils::MateStatus ms,
                        std::vector<rapmap::utils::QuasiAlignment>& hits, const Chunk* src = nullptr, uint64_t gen = 0, int64_t read = -1) {
  using namespace rapmap::utils;
  if (!chained) {
    for (int64_t i = 0; i < n; ++i) {
      hits.emplace_back(static_cast<uint32_t>(w[i] >> 33), static_cast<int32_t>(static_cast<uint32_t>(w[i])), ((w[i] >> 32) & 1) == 0, readLen);
      { QuasiAlignment& q = hits.back(); q.mateStatus = ms; q.qm_chunk_ = src; q.qm_gen_ = gen; q.qm_read_ = read; }
      // --fuzzyIntersection lists keep both orientations of a transcript: the second entry is the first one's
      // oppositeStrandPositions (mergeOrientationUnique, src/HitManager.cpp:846-866)
      const size_t m = hits.size();
      if (m >= 2 && hits[m - 2].tid == hits[m - 1].tid) { hits[m - 2].oppositeStrandPositions.push_back(hits[m - 1].pos); hits.pop_back(); continue; }
      hits.back().allPositions.push_back(hits.back().pos);
    }
    return;
  }
  for (int64_t i = 0; i < n;) {
    const uint64_t h = w[i];
    const int32_t np = static_cast<int32_t>(h >> 36), no = static_cast<int32_t>(w[i + 1] >> 32);
    hits.emplace_back(static_cast<uint32_t>(h), static_cast<int32_t>(static_cast<uint32_t>(w[i + 1])), ((h >> 32) & 1) == 0, readLen);
    QuasiAlignment& q = hits.back();
    q.mateStatus = ms; q.qm_chunk_ = src; q.qm_gen_ = gen; q.qm_read_ = read;
    const ChainStatus cs = static_cast<ChainStatus>((h >> 33) & 7);
    if (ms == MateStatus::PAIRED_END_RIGHT) q.chainStatus.setRight(cs); else q.chainStatus.setLeft(cs);
    for (int32_t t = 0; t < np; ++t) q.allPositions.push_back(static_cast<int32_t>(static_cast<uint32_t>(w[i + 2 + t])));
    for (int32_t t = 0; t < no; ++t) q.oppositeStrandPositions.push_back(static_cast<int32_t>(static_cast<uint32_t>(w[i + 2 + np + t])));
    q.hasMultiPos = np > 1;
    i += 2 + np + no;
  }
}

// the inverse, for a merge call whose hit vectors did not come out of a chunk
inline void encode_list(const std::vector<rapmap::utils::QuasiAlignment>& hits, bool chained, bool right, std::vector<uint64_t>& w) {
  using namespace rapmap::utils;
  for (const QuasiAlignment& q : hits) {
    if (!chained) {
      w.push_back((static_cast<uint64_t>(q.tid) << 33) | (static_cast<uint64_t>(q.fwd ? 0 : 1) << 32) | static_cast<uint32_t>(q.pos));
      for (int32_t p : q.oppositeStrandPositions)
        w.push_back((static_cast<uint64_t>(q.tid) << 33) | (static_cast<uint64_t>(q.fwd ? 1 : 0) << 32) | static_cast<uint32_t>(p));
    } else {
      const ChainStatus cs = right ? q.chainStatus.getRight() : q.chainStatus.getLeft();
      w.push_back(static_cast<uint64_t>(q.tid) | (static_cast<uint64_t>(q.fwd ? 0 : 1) << 32) | (static_cast<uint64_t>(cs) << 33) |
                  (static_cast<uint64_t>(q.allPositions.size()) << 36));
      w.push_back(static_cast<uint64_t>(static_cast<uint32_t>(q.pos)) | (static_cast<uint64_t>(q.oppositeStrandPositions.size()) << 32));
      for (int32_t p : q.allPositions) w.push_back(static_cast<uint32_t>(p));
      for (int32_t p : q.oppositeStrandPositions) w.push_back(static_cast<uint32_t>(p));
    }
  }
}

inline bool same_stage_opts(const qm_opts& a, const qm_opts& b) {
  return a.sensitive == b.sensitive && a.strict_check == b.strict_check && a.max_interval == b.max_interval && a.quasi_cov == b.quasi_cov &&
         a.sel_aln == b.sel_aln && a.max_mmp_extension == b.max_mmp_extension && a.consensus_slack == b.consensus_slack && a.fuzzy == b.fuzzy;
}

// MappingConfig -> the library's options for hitsToMappingsSimple.  The device implements the two configurations the
// reference's own driver uses (src/RapMapSAMapper.cpp:180-190): plain, and chaining with multiple positions.
inline void apply_mc(const rapmap::utils::MappingConfig& mc, qm_opts& o) {
  if (mc.consistentHits) throw Error(QM_E_UNSUPPORTED, "MappingConfig::consistentHits is not on the device path");
  if (mc.doChaining != mc.considerMultiPos) throw Error(QM_E_UNSUPPORTED, "MappingConfig: doChaining and considerMultiPos go together on the device path");
  if (!mc.doChaining && mc.consensusFraction != 1.0f) throw Error(QM_E_UNSUPPORTED, "MappingConfig: consensusFraction < 1 needs doChaining on the device path");
  o.sel_aln = mc.doChaining ? 1 : 0;
  o.consensus_slack = mc.doChaining ? -static_cast<double>(mc.consensusFraction) : 0.2;   // negative: the fraction itself (qmap_mi355.h)
}

}  // namespace detail
}  // namespace qmap

// ------------------------------------------------------------------------------------------------ SASearcher / SACollector
// SASearcher (include/SASearcher.hpp): the collector's suffix-array search object.  Its work happens on the device inside
// the collector call; the class exists so that the caller's `SASearcher<RapMapIndexT> saSearcher(&rmi);` compiles.
template <typename RapMapIndexT>
class SASearcher {
 public:
  using OffsetT = typename RapMapIndexT::IndexType;
  explicit SASearcher(RapMapIndexT* rmi) : rmi_(rmi) {}
  RapMapIndexT* index() const { return rmi_; }
 private:
  RapMapIndexT* rmi_;
};

// SACollector (include/SACollector.hpp:35-108)
template <typename RapMapIndexT>
class SACollector {
 public:
  using OffsetT = typename RapMapIndexT::IndexType;
  using HCInfo = rapmap::hit_manager::HitCollectorInfo<rapmap::utils::SAIntervalHit<OffsetT>>;

  // (every setter moves settings_: a group that was sent under other settings is not what a later per-read call asks for)
  void disableNIP() { disableNIP_ = true; ++settings_; }
  void enableNIP() { disableNIP_ = false; ++settings_; }
  void setCoverageRequirement(double req) { covReq_ = req; ++settings_; }
  double getCoverageRequirement() const { return covReq_; }
  void setMaxInterval(OffsetT maxInterval) { maxInterval_ = maxInterval; ++settings_; }
  OffsetT getMaxInterval(OffsetT) const { return maxInterval_; }
  bool getStrictCheck() const { return strictCheck_; }
  void setStrictCheck(bool sc) { strictCheck_ = sc; ++settings_; }
  void enableChainScoring() { doChaining_ = true; ++settings_; }
  void disableChainScoring() { doChaining_ = false; ++settings_; }
  bool getChainScoring() const { return doChaining_; }
  void setMaxMMPExtension(int32_t ext) { if (ext > 0) { maxMMPExtension_ = ext; ++settings_; } }
  int32_t getMaxMMPExtension() const { return maxMMPExtension_; }

  // Not in the reference: a caller that hands HitCollectorInfo straight on to hitsToMappingsSimple (as src/RapMapSAMapper.cpp:466-486 does)
  // and never looks inside it can say so -- the groups it sends then come back WITHOUT the SA-interval records (fwdSAInts / rcSAInts stay
  // empty; foundHit, the hit lists and the merges are what they always are), the device pass runs on the pair / lean kernels and half the
  // bytes come down.  Default: the records are there.
  void setKeepIntervals(bool keep) { keepIntervals_ = keep; ++settings_; }
  bool getKeepIntervals() const { return keepIntervals_; }

  explicit SACollector(RapMapIndexT* rmi) : rmi_(rmi) {}
  ~SACollector() { pending_.clear(); chunk_.reset(); spare_.clear(); count_in(nullptr); }   // (the groups first, then this collector's place among the service's workers)
  SACollector(const SACollector&) = delete;
  SACollector& operator=(const SACollector&) = delete;

  // ---- the added call (see the head of this file): map a whole chunk in one fused GPU pass --------------------------
  // PairRange: anything iterable whose elements have .first.seq and .second.seq (fastx_parser's ReadPair chunk, a
  // std::vector<std::pair<Read, Read>>, ...).  The strings must stay where they are until their per-read calls were made.
  // `mc` and `fuzzyMerge` say how the later stages will be called (defaults: the plain configuration).
  template <typename PairRange>
  void prefetch(PairRange& rg, const rapmap::utils::MappingConfig& mc = rapmap::utils::MappingConfig(), bool fuzzyMerge = false,
                uint32_t maxNumHits = 200) {
    std::vector<const std::string*>& l = ptrs_[0]; std::vector<const std::string*>& r = ptrs_[1];
    l.clear(); r.clear();
    for (auto& rp : rg) { l.push_back(&rp.first.seq); r.push_back(&rp.second.seq); }
    prefetchPairs(l, r, mc, fuzzyMerge, maxNumHits);
  }
  // The same in two halves (round 5), so that a worker can have its NEXT group on the device while it runs the per-read loop over
  // the current one: prefetch_async(rg) packs the group into the open batch and returns; wait() -- or prefetch(rg) for the same
  // group -- makes the oldest group that was sent this way the current one.  Groups become current in the order they were sent.
  //     auto next = parser.getReadGroup(); parser.refill(next); hitCollector.prefetch_async(next);
  //     for (...) { hitCollector.wait(); /* per-read calls over the current group */ ... refill + prefetch_async of the one after next ... }
  template <typename PairRange>
  void prefetch_async(PairRange& rg, const rapmap::utils::MappingConfig& mc = rapmap::utils::MappingConfig(), bool fuzzyMerge = false,
                      uint32_t maxNumHits = 200) {
    std::vector<const std::string*>& l = ptrs_[0]; std::vector<const std::string*>& r = ptrs_[1];
    l.clear(); r.clear();
    for (auto& rp : rg) { l.push_back(&rp.first.seq); r.push_back(&rp.second.seq); }
    submitPairs(l, r, mc, fuzzyMerge, maxNumHits);
  }
  size_t pending() const { return pending_.size(); }
  // the oldest group sent with prefetch_async becomes the current one (blocks until its batch has been mapped)
  void wait() {
    using namespace qmap::detail;
    if (pending_.empty()) throw qmap::Error(QM_E_STATE, "SACollector::wait(): no group was sent with prefetch_async");
    std::unique_ptr<Chunk> ch = std::move(pending_.front());
    pending_.pop_front();
    if (ch->batch) {
      Batch* bt = ch->batch;
      // (a batch that failed: the group that was current stays current, this one is dropped)
      try { ch->svc->wait_done(bt); } catch (...) { ch->batch = nullptr; throw; }
      ch->v = bt->v;
      ch->nreads = 2 * ch->nPacked;
    }
    // the group that was current is given back: its batch is released, a new generation number makes everything that was handed
    // out of it stale, and the object itself is kept for a later group (hits and interval lists hold a pointer to it)
    if (chunk_) { chunk_->release(); chunk_->gen = next_gen(); spare_.push_back(std::move(chunk_)); }
    last_index().ix = rmi_->handle(); last_index().device = rmi_->device();
    chunk_ = std::move(ch);
  }
  void prefetchPairs(const std::vector<const std::string*>& left, const std::vector<const std::string*>& right,
                     const rapmap::utils::MappingConfig& mc, bool fuzzyMerge, uint32_t maxNumHits) {
    // the group was sent ahead: wait for it; otherwise send it now (behind nothing) and wait
    const void* first = left.empty() ? nullptr : static_cast<const void*>(left[0]);
    if (!pending_.empty()) {
      if (pending_.front()->firstRead != first || pending_.front()->nPacked != static_cast<int64_t>(left.size()))
        throw qmap::Error(QM_E_STATE, "SACollector::prefetch(): another group was sent with prefetch_async before this one; groups become current in the order they were sent");
    } else submitPairs(left, right, mc, fuzzyMerge, maxNumHits);
    wait();
  }
 private:
  void submitPairs(const std::vector<const std::string*>& left, const std::vector<const std::string*>& right,
                   const rapmap::utils::MappingConfig& mc, bool fuzzyMerge, uint32_t maxNumHits) {
    using namespace qmap::detail;
    const int64_t n = static_cast<int64_t>(left.size());
    // a chunk object per group in flight; a new generation number per group makes everything that was handed out of an earlier
    // group stale once that group's chunk is given back
    std::unique_ptr<Chunk> chp;
    if (!spare_.empty()) { chp = std::move(spare_.back()); spare_.pop_back(); } else chp.reset(new Chunk());
    Chunk* ch = chp.get();
    ch->gen = next_gen(); ch->paired = true; ch->nreads = 0; ch->cursor = 0;
    stageOpts(ch->opts);
    ch->settings = settings_; ch->owner = this;
    apply_mc(mc, ch->opts);
    ch->opts.fuzzy = (fuzzyMerge || mc.doChaining) ? 1 : 0;
    ch->opts.max_num_hits = static_cast<int32_t>(maxNumHits);
    ch->firstRead = n ? static_cast<const void*>(left[0]) : nullptr;
    ch->nPacked = n;
    if (n > 0) {
      size_t b1 = 0, b2 = 0;
      for (int64_t i = 0; i < n; ++i) { b1 += left[i]->size(); b2 += right[i]->size(); }
      // everything that can throw happens before the group joins a batch (a joined group that never reports "packed" would
      // hold the batch's dispatcher, and every other worker in the batch, forever)
      ch->key.resize(static_cast<size_t>(2 * n)); ch->keyLen.resize(static_cast<size_t>(2 * n));
      pending_.emplace_back(nullptr);                          // (the deque's node, allocated now)
      pending_.pop_back();
      // join the batch that is open (qmap::detail::Service): this group's place in the batch's page-locked input buffers ...
      Service* svc = service_of(rmi_->handle(), rmi_->device());
      count_in(svc);
      ch->bind(svc);
      const Service::Place pl = svc->join(ch->opts, n, b1, b2, keepIntervals_ ? 0u : static_cast<uint32_t>(QM_STAGES_NO_INTERVALS));
      Batch* bt = pl.b;
      ch->batch = bt;
      // ... the reads packed there, next to the other workers' (the upload is a DMA straight out of these buffers) ...
      size_t p1 = pl.c1, p2 = pl.c2;
      int64_t* o1 = bt->o1 + pl.u0; int64_t* o2 = bt->o2 + pl.u0;
      for (int64_t i = 0; i < n; ++i) {
        const std::string& l = *left[i]; const std::string& r = *right[i];
        std::memcpy(bt->s1 + p1, l.data(), l.size()); p1 += l.size(); o1[i + 1] = static_cast<int64_t>(p1);
        std::memcpy(bt->s2 + p2, r.data(), r.size()); p2 += r.size(); o2[i + 1] = static_cast<int64_t>(p2);
        ch->key[2 * i] = l.data(); ch->keyLen[2 * i] = l.size();
        ch->key[2 * i + 1] = r.data(); ch->keyLen[2 * i + 1] = r.size();
      }
      ch->ubase = pl.u0; ch->rbase = 2 * pl.u0;
      ch->s1 = bt->s1; ch->s2 = bt->s2; ch->o1 = o1; ch->o2 = o2;
      svc->pack_slice(bt, pl.u0, n, pl.c1, pl.c2, o1, o2);
      // ... and handed to the dispatcher: the batch's one fused pass brings every stage's output of every group in it down in one
      // go (intervals and foundHit per read, per-read lists, merge results and tooMany flags per pair); wait() picks it up
      svc->packed(bt);
    }
    pending_.push_back(std::move(chp));
  }
 public:

  // SACollector::operator() (include/SACollector.hpp:108-362)
  bool operator()(std::string& read, SASearcher<RapMapIndexT>& /*saSearcher*/, HCInfo& hcInfo) {
    using namespace qmap::detail;
    hcInfo.readLen = read.length();
    hcInfo.maxDist = static_cast<int32_t>(read.length());
    Chunk* ch = chunk_.get();
    if (ch) {
      // the group was sent under the collector's settings of that moment (stageOpts): the same ones now, or the stage settings are compared
      bool sameSettings = ch->settings == settings_ && ch->owner == this;
      if (!sameSettings) {
        qm_opts now; stageOpts(now); now.sel_aln = ch->opts.sel_aln; now.consensus_slack = ch->opts.consensus_slack; now.fuzzy = ch->opts.fuzzy;
        sameSettings = same_stage_opts(now, ch->opts);
      }
      // The chunk's reads are recognised in order: by the address and length of their characters AND by the characters
      // themselves (a parser that refills its string buffers in place hands out the same addresses with new contents).  A
      // read the caller skipped does not end the fast path: the next few entries are looked at as well.
      int64_t idx = -1;
      for (int64_t c = ch->cursor, lim = c + 64 < ch->nreads ? c + 64 : ch->nreads; c < lim; ++c) {
        if (ch->key[c] != read.data() || ch->keyLen[c] != read.size()) continue;
        const char* sq = (ch->paired && (c & 1)) ? ch->s2 : ch->s1;
        const int64_t* so = (ch->paired && (c & 1)) ? ch->o2 : ch->o1;
        const int64_t u = ch->paired ? (c >> 1) : c;
        if (u >= ch->nPacked || static_cast<size_t>(so[u + 1] - so[u]) != read.size()) continue;
        if (read.size() && std::memcmp(sq + so[u], read.data(), read.size()) != 0) continue;
        idx = c; ch->cursor = c + 1; break;
      }
      if (idx >= 0 && sameSettings && (doChaining_ ? 1 : 0) == ch->opts.sel_aln) {
        for (int64_t j = ch->v.iv_off[ch->rbase + idx]; j < ch->v.iv_off[ch->rbase + idx + 1]; ++j) {
          const qm_sa_interval_hit& h = ch->v.iv[j];
          (h.query_rc ? hcInfo.rcSAInts : hcInfo.fwdSAInts).emplace_back(static_cast<OffsetT>(static_cast<uint32_t>(h.begin)), static_cast<OffsetT>(static_cast<uint32_t>(h.end)), h.len, h.query_pos, h.query_rc != 0);
        }
        hcInfo.qm_chunk_ = ch; hcInfo.qm_gen_ = ch->gen; hcInfo.qm_read_ = idx;
        return ch->v.found[ch->rbase + idx] != 0;
      }
    }
    // a batch of one
    qm_opts o; stageOpts(o);
    qm_ctx* ctx = thread_ctx(*rmi_);
    const int64_t off[2] = {0, static_cast<int64_t>(read.size())};
    int64_t ni = 0;
    check(qm_collect_reads(ctx, &o, 1, read.data(), off, &ni));
    int64_t ioff[2] = {0, 0};
    std::vector<qm_sa_interval_hit> iv(static_cast<size_t>(ni) + 1);
    check(qm_fetch_intervals(ctx, ioff, iv.data(), ni));
    for (int64_t j = 0; j < ioff[1]; ++j) {
      const qm_sa_interval_hit& h = iv[static_cast<size_t>(j)];
      (h.query_rc ? hcInfo.rcSAInts : hcInfo.fwdSAInts).emplace_back(static_cast<OffsetT>(static_cast<uint32_t>(h.begin)), static_cast<OffsetT>(static_cast<uint32_t>(h.end)), h.len, h.query_pos, h.query_rc != 0);
    }
    hcInfo.qm_chunk_ = nullptr; hcInfo.qm_read_ = -1;
    uint8_t f = 0;
    check(qm_fetch_found(ctx, &f));
    return f != 0;
  }

 private:
  void stageOpts(qm_opts& o) const {
    qm_opts_default(&o);
    o.sensitive = disableNIP_ ? 1 : 0;       // --noSensitive leaves NIP skipping on; the CLI default disables it (RapMapSAMapper.cpp:1113-1114)
    o.strict_check = strictCheck_ ? 1 : 0;
    o.quasi_cov = covReq_;
    o.max_interval = static_cast<int32_t>(maxInterval_);
    o.sel_aln = doChaining_ ? 1 : 0;
    o.max_mmp_extension = maxMMPExtension_;
  }
  RapMapIndexT* rmi_;
  // the reference's constructor defaults (SACollector.hpp:77-81)
  bool disableNIP_{false};
  double covReq_{0.0};
  OffsetT maxInterval_{1000};
  bool strictCheck_{false};
  bool doChaining_{false};
  int32_t maxMMPExtension_{7};
  uint64_t settings_{0};                                       // moved by every setter
  std::vector<const std::string*> ptrs_[2];                    // (the group's strings, listed: kept between calls)
  std::unique_ptr<qmap::detail::Chunk> chunk_;                 // the current group
  std::deque<std::unique_ptr<qmap::detail::Chunk>> pending_;   // groups sent ahead (prefetch_async), oldest first
  std::vector<std::unique_ptr<qmap::detail::Chunk>> spare_;    // chunk objects of groups that were given back
  // this collector counted once among its service's workers (the dispatcher closes a batch when half of them have joined), whatever the
  // number of chunk objects it holds (current, sent ahead, spare)
  bool keepIntervals_{true};
  qmap::detail::Service* counted_{nullptr};
  void count_in(qmap::detail::Service* s) { if (counted_ != s) { if (counted_) counted_->detach(); counted_ = s; if (s) s->attach(); } }
};

// ------------------------------------------------------------------------------------------------ hitsToMappingsSimple
namespace rapmap {
namespace hit_manager {

// include/HitManager.hpp:130-135, src/HitManager.cpp:691-882
template <typename RapMapIndexT>
void hitsToMappingsSimple(RapMapIndexT& rmi, rapmap::utils::MappingConfig& mc, rapmap::utils::MateStatus mateStatus,
                          HitCollectorInfo<rapmap::utils::SAIntervalHit<typename RapMapIndexT::IndexType>>& hcinfo,
                          std::vector<rapmap::utils::QuasiAlignment>& hits) {
  using namespace qmap::detail;
  const size_t before = hits.size();
  const uint32_t readLen = static_cast<uint32_t>(hcinfo.readLen);
  const Chunk* ch = hcinfo.qm_chunk_;
  if (mc.consistentHits || mc.doChaining != mc.considerMultiPos || (!mc.doChaining && mc.consensusFraction != 1.0f)) { qm_opts t; qm_opts_default(&t); apply_mc(mc, t); }   // (throws)
  if (ch && ch->gen == hcinfo.qm_gen_ && hcinfo.qm_read_ >= 0 && (mc.doChaining ? 1 : 0) == ch->opts.sel_aln &&
      (!mc.doChaining || -static_cast<double>(mc.consensusFraction) == ch->opts.consensus_slack) &&
      static_cast<int64_t>(hcinfo.fwdSAInts.size() + hcinfo.rcSAInts.size()) == ch->v.iv_off[ch->rbase + hcinfo.qm_read_ + 1] - ch->v.iv_off[ch->rbase + hcinfo.qm_read_]) {
    // the chunk's pass already turned exactly these intervals into the read's list
    const int64_t r = hcinfo.qm_read_, ar = ch->rbase + r;
    decode_list(ch->v.words + ch->v.list_off[ar], ch->v.list_off[ar + 1] - ch->v.list_off[ar], mc.doChaining, readLen, mateStatus, hits, ch, ch->gen, r);
    if (hits.size() > before) hits[before].qm_count_ = static_cast<int32_t>(hits.size() - before);
    return;
  }
  // a batch of one, from the intervals the caller holds
  qm_opts o; qm_opts_default(&o);
  apply_mc(mc, o);
  std::vector<qm_sa_interval_hit> iv;
  for (int t = 0; t < 2; ++t) {
    auto& v = t == 0 ? hcinfo.fwdSAInts : hcinfo.rcSAInts;
    for (auto& h : v) { qm_sa_interval_hit x; x.begin = static_cast<int32_t>(h.begin); x.end = static_cast<int32_t>(h.end); x.len = h.len; x.query_pos = h.queryPos; x.query_rc = static_cast<uint8_t>(t); x.list = static_cast<uint8_t>(t); x.pad = 0; iv.push_back(x); }
  }
  const int32_t len = static_cast<int32_t>(readLen);
  const int64_t ioff[2] = {0, static_cast<int64_t>(iv.size())};
  o.fuzzy = 1;                                               // both orientations kept: decode_list folds them into oppositeStrandPositions
  qm_ctx* ctx = thread_ctx(rmi);
  int64_t nw = 0;
  check(qm_hits_to_mappings(ctx, &o, 1, &len, ioff, iv.data(), &nw));
  int64_t loff[2] = {0, 0};
  std::vector<uint64_t> w(static_cast<size_t>(nw) + 1);
  check(qm_fetch_read_lists(ctx, loff, w.data(), nw));
  decode_list(w.data(), loff[1], mc.doChaining, readLen, mateStatus, hits);
}

}  // namespace hit_manager

// ------------------------------------------------------------------------------------------------ the merges
namespace utils {
namespace qm_detail {

// the pair both vectors were produced for, if they came out of one chunk untouched
inline const qmap::detail::Chunk* merge_source(const std::vector<QuasiAlignment>& l, const std::vector<QuasiAlignment>& r, bool chained, bool fuzzy,
                                               uint32_t maxNumHits, int64_t& unit) {
  const QuasiAlignment* a = l.empty() ? nullptr : &l.front();
  const QuasiAlignment* b = r.empty() ? nullptr : &r.front();
  const QuasiAlignment* any = a ? a : b;
  if (!any || !any->qm_chunk_ || any->qm_chunk_->gen != any->qm_gen_) return nullptr;
  const qmap::detail::Chunk* ch = any->qm_chunk_;
  if (!ch->paired || (ch->opts.sel_aln != 0) != chained || ((ch->opts.fuzzy != 0) != (fuzzy || chained)) || static_cast<uint32_t>(ch->opts.max_num_hits) != maxNumHits) return nullptr;
  if (a && b && (b->qm_chunk_ != ch || b->qm_gen_ != ch->gen || b->qm_read_ != a->qm_read_ + 1)) return nullptr;
  const int64_t rd = a ? a->qm_read_ : b->qm_read_ - 1;
  if (rd < 0 || (rd & 1)) return nullptr;
  // both vectors must be what the chunk has for the two mates: as many hits as hitsToMappingsSimple handed out (noted on the first
  // one), and an empty vector only where the chunk's list is empty (an emptied vector is an edit)
  auto intact = [&](const std::vector<QuasiAlignment>& v, int64_t read) {
    if (!v.empty()) return v.front().qm_count_ == static_cast<int32_t>(v.size());
    const int64_t ar = ch->rbase + read;
    return ch->v.list_off[ar + 1] == ch->v.list_off[ar];
  };
  if (!intact(l, rd) || !intact(r, rd + 1)) return nullptr;
  unit = rd >> 1;
  return ch;
}

template <typename HitCountersT>
inline MergeResult merge_impl(bool fuzzy, bool leftMatches, bool rightMatches, std::vector<QuasiAlignment>& leftHits,
                              std::vector<QuasiAlignment>& rightHits, std::vector<QuasiAlignment>& jointHits, bool chained,
                              uint32_t maxNumHits, bool& tooManyHits, HitCountersT& hctr, qm_ctx* ctx_or_null) {
  using namespace qmap::detail;
  MergeResult res = MergeResult::HAD_NONE;
  // orphans are the caller's own objects moved over, exactly as the reference moves them (RapMapUtils.hpp:880-901,1251-1262)
  auto moveAll = [&](std::vector<QuasiAlignment>& v) { jointHits.insert(jointHits.end(), std::make_move_iterator(v.begin()), std::make_move_iterator(v.end())); };
  if (fuzzy) {
    if (leftHits.empty()) {
      if (!leftMatches && !rightHits.empty()) { moveAll(rightHits); hctr.seHits += rightHits.size(); res = MergeResult::HAD_ONLY_RIGHT; }
      if (jointHits.size() > 0) hctr.peHits += jointHits.size();
      return res;
    }
    if (rightHits.empty()) {
      if (!rightMatches) { moveAll(leftHits); hctr.seHits += leftHits.size(); res = MergeResult::HAD_ONLY_LEFT; }
      if (jointHits.size() > 0) hctr.peHits += jointHits.size();
      return res;
    }
  } else if (leftHits.empty() || rightHits.empty()) {
    const size_t numHits = leftHits.size() + rightHits.size();
    if (numHits > 0) { hctr.seHits += numHits; moveAll(leftHits); moveAll(rightHits); }
    return res;
  }
  // both mates have hits: the intersection is device work
  const uint32_t l1 = leftHits.front().readLen, l2 = rightHits.front().readLen;
  int64_t unit = -1;
  const Chunk* ch = merge_source(leftHits, rightHits, chained, fuzzy, maxNumHits, unit);
  std::vector<qm_hit> own; const qm_hit* hb = nullptr; int64_t nh = 0; uint8_t flags = 0;
  if (ch) {
    const int64_t au = ch->ubase + unit;
    hb = ch->v.hits + ch->v.hit_off[au]; nh = ch->v.hit_off[au + 1] - ch->v.hit_off[au]; flags = ch->v.too_many[au];
  } else {
    if (!ctx_or_null) throw qmap::Error(QM_E_STATE, "merge of hit vectors that did not come from this thread's device context");
    qm_opts o; qm_opts_default(&o);
    o.fuzzy = fuzzy ? 1 : 0; o.sel_aln = chained ? 1 : 0; o.max_num_hits = static_cast<int32_t>(maxNumHits);
    std::vector<uint64_t> wl, wr;
    encode_list(leftHits, chained, false, wl); encode_list(rightHits, chained, true, wr);
    const int64_t ol[2] = {0, static_cast<int64_t>(wl.size())}, orr[2] = {0, static_cast<int64_t>(wr.size())};
    const uint8_t fl = leftMatches ? 1 : 0, fr = rightMatches ? 1 : 0;
    const int32_t ll = static_cast<int32_t>(l1), lr = static_cast<int32_t>(l2);
    qm_counters c{};
    check(qm_merge_lists(ctx_or_null, &o, 1, ol, wl.data(), orr, wr.data(), &fl, &fr, &ll, &lr, &nh, &c));
    int64_t ho[2] = {0, 0};
    own.resize(static_cast<size_t>(nh) + 1);
    check(qm_fetch_hits(ctx_or_null, ho, own.data()));
    check(qm_fetch_too_many(ctx_or_null, &flags));
    hb = own.data();
  }
  const bool tooMany = (flags & 1) != 0, sameTxp = (flags & 2) != 0;
  if (tooMany) { tooManyHits = true; ++hctr.tooManyHits; }
  int64_t paired = 0;
  for (int64_t i = 0; i < nh; ++i) if (hb[i].is_paired) ++paired;
  if (paired > 0) {
    for (int64_t i = 0; i < nh; ++i) {
      jointHits.emplace_back();
      decode_hit(hb[i], jointHits.back());
      if (chained) jointHits.back().chainStatus = FragmentChainStatus(static_cast<ChainStatus>(hb[i].aln_score & 15), static_cast<ChainStatus>((hb[i].aln_score >> 4) & 15));
    }
    hctr.peHits += jointHits.size();
    res = MergeResult::HAD_CONCORDANT;
  } else if (fuzzy) {
    res = tooMany ? MergeResult::HAD_CONCORDANT : (sameTxp ? MergeResult::HAD_DISCORDANT : MergeResult::HAD_EMPTY_INTERSECTION);
  } else if (!tooMany) {
    // no common transcript: the single-end hits of both mates (RapMapUtils.hpp:1246-1262)
    hctr.seHits += leftHits.size() + rightHits.size();
    moveAll(leftHits); moveAll(rightHits);
  }
  return res;
}
}  // namespace qm_detail

// The merges have no index argument; one that has to go to the device uses the calling thread's most recent context (the
// collector / hitsToMappingsSimple calls that produced its inputs set it).

// include/RapMapUtils.hpp:1185-1264
inline void mergeLeftRightHits(std::vector<QuasiAlignment>& leftHits, std::vector<QuasiAlignment>& rightHits,
                               std::vector<QuasiAlignment>& jointHits, uint32_t /*readLen*/, uint32_t maxNumHits, bool& tooManyHits,
                               HitCounters& hctr) {
  qm_detail::merge_impl(false, true, true, leftHits, rightHits, jointHits, false, maxNumHits, tooManyHits, hctr, qmap::detail::merge_ctx());
}

// include/RapMapUtils.hpp:864-1183
inline MergeResult mergeLeftRightHitsFuzzy(bool leftMatches, bool rightMatches, std::vector<QuasiAlignment>& leftHits,
                                           std::vector<QuasiAlignment>& rightHits, std::vector<QuasiAlignment>& jointHits,
                                           rapmap::utils::MappingConfig& mc, uint32_t /*readLen*/, uint32_t maxNumHits, bool& tooManyHits,
                                           HitCounters& hctr) {
  return qm_detail::merge_impl(true, leftMatches, rightMatches, leftHits, rightHits, jointHits, mc.considerMultiPos, maxNumHits, tooManyHits,
                               hctr, qmap::detail::merge_ctx());
}

}  // namespace utils
}  // namespace rapmap

#endif  // QMAP_RAPMAP_COMPAT_HPP
