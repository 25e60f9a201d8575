// qmap_compat.hpp -- header-only C++ face of libqmap_mi355.so with the reference's vocabulary.
//
// RapMap's callers (and Salmon) work with rapmap::utils::QuasiAlignment / HitCounters / MappingConfig and call,
// per read pair, SACollector::operator() x2 -> hitsToMappingsSimple x2 -> mergeLeftRightHits
// (src/RapMapSAMapper.cpp:461-551).  This header keeps those types and collapses the five calls into
// QuasiMapper::mapReadPairs() over a whole chunk of pairs -- the granularity a GPU needs.
// Field names and meanings follow include/RapMapUtils.hpp:208-216 (HitCounters), :356-362 (MateStatus),
// :399-502 (QuasiAlignment); only fields that are defined on this path exist.
#ifndef QMAP_COMPAT_HPP
#define QMAP_COMPAT_HPP

#include <atomic>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "qmap_mi355.h"

namespace qmap {

enum class MateStatus : uint8_t { SINGLE_END = 0, PAIRED_END_LEFT = 1, PAIRED_END_RIGHT = 2, PAIRED_END_PAIRED = 3 };

struct QuasiAlignment {
  uint32_t tid;
  int32_t pos;
  int32_t matePos;
  bool fwd;
  bool mateIsFwd;
  uint32_t fragLen;
  uint32_t readLen;
  uint32_t mateLen;
  bool isPaired;
  MateStatus mateStatus;
  double score_{1.0};
  int32_t alnScore_{0};
  uint32_t transcriptID() const { return tid; }
  double score() const { return score_; }
  int32_t alnScore() const { return alnScore_; }
  uint32_t fragLength() const { return fragLen; }
};

struct HitCounters {
  std::atomic<uint64_t> peHits{0}, seHits{0}, trueHits{0}, totHits{0}, numReads{0}, tooManyHits{0}, lastPrint{0};
};

// the hot-path subset of MappingOpts (src/RapMapSAMapper.cpp:114-152)
struct MappingOpts {
  uint32_t maxNumHits{200};
  double quasiCov{0.0};
  bool sensitive{true};      // !--noSensitive
  bool strictCheck{true};    // !--noStrictCheck
  bool fuzzy{false};
  bool selAln{false};
  bool noOrphans{false};
  bool noDovetail{false};
  // sub-options of --selAln (src/RapMapSAMapper.cpp:1011-1023,1135-1175); read only when selAln is set
  bool hardFilter{false};
  int32_t matchScore{2}, mismatchPenalty{-4}, gapOpenPenalty{4}, gapExtendPenalty{2}, dpBandwidth{15}, maxMMPExtension{7};
  int32_t alnPolicy{0};          // 0 default, 1 --mimicBT2, 2 --mimicStrictBT2
  double minScoreFrac{0.65};
  float consensusSlack{0.2f};
};

class Error : public std::runtime_error {
 public:
  Error(int code, const char* what) : std::runtime_error(what), code_(code) {}
  int code() const { return code_; }
 private:
  int code_;
};
inline void check(int rc) { if (rc) throw Error(rc, qm_last_error()); }

// RapMapSAIndex<int32_t, ...>: load() + txpNames / txpLens (include/RapMapSAIndex.hpp:48-82)
class QuasiIndex {
 public:
  QuasiIndex() = default;
  explicit QuasiIndex(const std::string& dir) { load(dir); }
  ~QuasiIndex() { if (ix_) qm_index_close(ix_); }
  QuasiIndex(const QuasiIndex&) = delete;
  QuasiIndex& operator=(const QuasiIndex&) = delete;
  bool load(const std::string& dir) {
    check(qm_index_open(dir.c_str(), &ix_));
    check(qm_index_info_get(ix_, &info_));
    txpNames.clear(); txpLens.clear();
    for (int64_t i = 0; i < info_.n_txps; ++i) { txpNames.emplace_back(qm_index_txp_name(ix_, i)); txpLens.push_back((uint32_t)qm_index_txp_len(ix_, i)); }
    return true;
  }
  uint32_t k() const { return (uint32_t)info_.k; }
  bool perfectHash() const { return info_.perfect_hash != 0; }
  const qm_index* handle() const { return ix_; }
  std::vector<std::string> txpNames;
  std::vector<uint32_t> txpLens;
 private:
  qm_index* ix_{nullptr};
  qm_index_info info_{};
};

// One GPU context; one instance per host thread / per device.
class QuasiMapper {
 public:
  QuasiMapper(const QuasiIndex& index, int device, const MappingOpts& m = MappingOpts()) {
    check(qm_ctx_create(index.handle(), device, &ctx_));
    qm_opts_default(&o_);
    o_.sensitive = m.sensitive; o_.strict_check = m.strictCheck; o_.max_num_hits = (int32_t)m.maxNumHits;
    o_.no_orphans = m.noOrphans; o_.no_dovetail = m.noDovetail; o_.quasi_cov = m.quasiCov; o_.fuzzy = m.fuzzy; o_.sel_aln = m.selAln;
    o_.hard_filter = m.hardFilter; o_.match_score = m.matchScore; o_.mismatch_penalty = m.mismatchPenalty; o_.gap_open = m.gapOpenPenalty;
    o_.gap_extend = m.gapExtendPenalty; o_.dp_bandwidth = m.dpBandwidth; o_.max_mmp_extension = m.maxMMPExtension; o_.aln_policy = m.alnPolicy;
    o_.min_score_fraction = m.minScoreFrac; o_.consensus_slack = m.consensusSlack;
  }
  ~QuasiMapper() { if (ctx_) qm_ctx_destroy(ctx_); }
  QuasiMapper(const QuasiMapper&) = delete;
  QuasiMapper& operator=(const QuasiMapper&) = delete;

  // jointHits[i] receives what processReadsPairSA holds in `jointHits` for pair i right before it writes SAM
  // (src/RapMapSAMapper.cpp:701); hctr is updated like the reference's shared counters.
  void mapReadPairs(const std::vector<std::pair<std::string, std::string>>& pairs,
                    std::vector<std::vector<QuasiAlignment>>& jointHits, HitCounters& hctr) {
    s1_.clear(); s2_.clear(); o1_.assign(1, 0); o2_.assign(1, 0);
    for (auto& p : pairs) {
      s1_.insert(s1_.end(), p.first.begin(), p.first.end()); o1_.push_back((int64_t)s1_.size());
      s2_.insert(s2_.end(), p.second.begin(), p.second.end()); o2_.push_back((int64_t)s2_.size());
    }
    int64_t n = (int64_t)pairs.size(), nHits = 0;
    qm_counters c{};
    check(qm_map_pairs(ctx_, &o_, n, s1_.data(), o1_.data(), s2_.data(), o2_.data(), &nHits, &c));
    finish(n, nHits, c, jointHits, hctr);
  }
  // single-end: processReadsSingleSA (src/RapMapSAMapper.cpp:232-250)
  void mapReads(const std::vector<std::string>& reads, std::vector<std::vector<QuasiAlignment>>& hits, HitCounters& hctr) {
    s1_.clear(); o1_.assign(1, 0);
    for (auto& r : reads) { s1_.insert(s1_.end(), r.begin(), r.end()); o1_.push_back((int64_t)s1_.size()); }
    int64_t n = (int64_t)reads.size(), nHits = 0;
    qm_counters c{};
    check(qm_map_reads(ctx_, &o_, n, s1_.data(), o1_.data(), &nHits, &c));
    finish(n, nHits, c, hits, hctr);
  }

 private:
  void finish(int64_t n, int64_t nHits, const qm_counters& c, std::vector<std::vector<QuasiAlignment>>& out, HitCounters& hctr) {
    off_.resize((size_t)n + 1); raw_.resize((size_t)nHits);
    check(qm_fetch_hits(ctx_, off_.data(), raw_.data()));
    out.assign((size_t)n, {});
    for (int64_t i = 0; i < n; ++i) {
      for (int64_t j = off_[i]; j < off_[i + 1]; ++j) {
        const qm_hit& h = raw_[(size_t)j];
        QuasiAlignment q;
        q.tid = h.tid; q.pos = h.pos; q.matePos = h.mate_pos; q.fwd = h.fwd != 0; q.mateIsFwd = h.mate_is_fwd != 0;
        q.fragLen = h.frag_len; q.readLen = h.read_len; q.mateLen = h.mate_len; q.isPaired = h.is_paired != 0;
        q.mateStatus = static_cast<MateStatus>(h.mate_status); q.alnScore_ = h.aln_score;
        out[(size_t)i].push_back(q);
      }
    }
    hctr.numReads += c.num_reads; hctr.peHits += c.pe_hits; hctr.seHits += c.se_hits;
    hctr.totHits += c.tot_hits; hctr.tooManyHits += c.too_many_hits;
  }
  qm_ctx* ctx_{nullptr};
  qm_opts o_{};
  std::vector<char> s1_, s2_;
  std::vector<int64_t> o1_, o2_, off_;
  std::vector<qm_hit> raw_;
};

}  // namespace qmap
#endif  // QMAP_COMPAT_HPP
