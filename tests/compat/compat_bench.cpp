// tests/compat/compat_bench.cpp -- throughput of the REFERENCE'S OWN CALL SURFACE (include/qmap_rapmap_compat.hpp) driven the
// way `rapmap quasimap` drives it: T worker threads (src/RapMapSAMapper.cpp:752-799), each taking read groups of CHUNK pairs
// from a shared hand-out (the parser's chunks of 10 000, :853,:869-871) and running the body of processReadsPairSA
// (:461-551) on every pair -- collector x2 -> hitsToMappingsSimple x2 -> mergeLeftRightHits -> maxNumHits / counters.  The
// only added lines are `hitCollector.prefetch(rg)` and -- for the next group, sent ahead -- `hitCollector.prefetch_async(rg)`.  Compiled against the header ALONE (bench.py's `compat_face` leg and
// tests/test_rapmap_compat.py build it with g++).
//
//   compat_bench INDEX READS.bin NPAIRS READLEN THREADS CHUNK [--no-prefetch] [--repeat R] [--use N] [--mixed] [--depth D] [--no-intervals]
//
// READS.bin: NPAIRS*READLEN characters of the left mates, then as many of the right mates (what bench.py holds in HBM for the
// headline, copied to the host).  The read groups (std::string pairs, as the parser hands them out) are built before the
// timed region; a group is only ever touched by the thread that took it.  Prints one JSON line: pairs/s, the HitCounters and
// an order-sensitive digest of every jointHits vector (bench.py / the test compute the same digest from the oracle's hits).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "qmap_rapmap_compat.hpp"

struct Read { std::string seq; };
using ReadPair = std::pair<Read, Read>;
using Group = std::vector<ReadPair>;

static inline uint64_t hit_digest(uint64_t unit, uint64_t j, const rapmap::utils::QuasiAlignment& q) {
  using rapmap::utils::MateStatus;
  const bool paired = q.mateStatus == MateStatus::PAIRED_END_PAIRED;
  const uint64_t matePos = paired ? (uint64_t)(uint32_t)q.matePos : 0, fragLen = paired ? (uint64_t)q.fragLen : 0;
  const uint64_t flags = (q.fwd ? 1u : 0u) | ((paired ? q.mateIsFwd : true) ? 2u : 0u) | ((uint64_t)(uint8_t)q.mateStatus << 2);
  uint64_t v = unit * 0x9E3779B97F4A7C15ull + j * 0xD6E8FEB86659FD93ull + (uint64_t)q.tid * 0xC2B2AE3D27D4EB4Full +
               (uint64_t)(uint32_t)q.pos * 0x165667B19E3779F9ull + matePos * 0x27D4EB2F165667C5ull + flags * 0x85EBCA77C2B2AE63ull +
               fragLen * 0xFF51AFD7ED558CCDull;
  v ^= v >> 31; v *= 0xC4CEB9FE1A85EC53ull; v ^= v >> 29;
  return v;
}

static bool g_keepIntervals = true; // --no-intervals: the collectors say they do not look inside HitCollectorInfo (SACollector::setKeepIntervals(false))
static bool g_digest = true;      // --digest-once: the digest of every jointHits vector is taken in the first repeat only (the check), the later repeats time the reference's loop alone
struct Totals { uint64_t ph[5] = {0, 0, 0, 0, 0}; uint64_t digest = 0, pe = 0, se = 0, tot = 0, reads = 0, tooMany = 0, mapped = 0; double prefetchS = 0, loopS = 0, firstAt = 0, lastAt = 0, goSeenAt = 0; };
static inline uint64_t tick() { return __builtin_ia32_rdtsc(); }
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename RapMapIndexT>
static void worker(RapMapIndexT& rmi, std::vector<Group>& groups, const std::vector<size_t>& firstUnit, std::atomic<size_t>& next, bool prefetch,
                   uint32_t maxNumHits, Totals& T, rapmap::utils::HitCounters& hctr, Group* warm, std::atomic<int>& ready, std::atomic<int>& go, bool mixed = false,
                   int depth = 2) {
  using OffsetT = typename RapMapIndexT::IndexType;
  using rapmap::utils::MateStatus;
  using rapmap::utils::QuasiAlignment;
  std::vector<QuasiAlignment> leftHits, rightHits, jointHits;
  SACollector<RapMapIndexT> hitCollector(&rmi);
  hitCollector.disableNIP();
  hitCollector.setStrictCheck(true);
  hitCollector.setKeepIntervals(g_keepIntervals);
  rapmap::hit_manager::HitCollectorInfo<rapmap::utils::SAIntervalHit<OffsetT>> leftHCInfo, rightHCInfo;
  rapmap::utils::MappingConfig mc;
  mc.consistentHits = false; mc.doChaining = false;
  SASearcher<RapMapIndexT> saSearcher(&rmi);
  bool tooManyHits = false;
  uint32_t readLen = 0;
  rapmap::utils::HitCounters scratchCtr;
  Totals mine;                                         // (T is one element of a vector shared with the other threads: written once, at the end)
  auto run_group = [&](Group& rg, size_t unit0, bool count, bool fuzzy = false) {
    Totals& T = mine;
    rapmap::utils::HitCounters& hc = count ? hctr : scratchCtr;
    const double tp0 = now_s();
    if (prefetch) hitCollector.prefetch(rg, mc, fuzzy, maxNumHits);           // <- the one added line
    const double tp1 = now_s();
    size_t u = unit0;
    static const bool phases = std::getenv("COMPAT_BENCH_PHASES") != nullptr;     // where the loop's time goes (cycle counter around each stage)
    if (phases) {
      for (auto& rpair : rg) {
        const uint64_t c0 = tick();
        tooManyHits = false; readLen = rpair.first.seq.length(); ++hc.numReads;
        leftHCInfo.clear(); rightHCInfo.clear(); jointHits.clear(); leftHits.clear(); rightHits.clear();
        const uint64_t c1 = tick();
        bool lh = hitCollector(rpair.first.seq, saSearcher, leftHCInfo);
        bool rh = hitCollector(rpair.second.seq, saSearcher, rightHCInfo);
        (void)lh; (void)rh;
        const uint64_t c2 = tick();
        rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, MateStatus::PAIRED_END_LEFT, leftHCInfo, leftHits);
        rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, MateStatus::PAIRED_END_RIGHT, rightHCInfo, rightHits);
        const uint64_t c3 = tick();
        rapmap::utils::mergeLeftRightHits(leftHits, rightHits, jointHits, readLen, maxNumHits, tooManyHits, hc);
        if (jointHits.size() > maxNumHits) { jointHits.clear(); }
        hc.totHits += jointHits.size();
        const uint64_t c4 = tick();
        if (count) { if (!jointHits.empty()) ++T.mapped; if (g_digest) for (size_t j = 0; j < jointHits.size(); ++j) T.digest += hit_digest(u, j, jointHits[j]); }
        const uint64_t c5 = tick();
        if (count) { T.ph[0] += c1 - c0; T.ph[1] += c2 - c1; T.ph[2] += c3 - c2; T.ph[3] += c4 - c3; T.ph[4] += c5 - c4; }
        ++u;
      }
    } else
    for (auto& rpair : rg) {
      // ---- src/RapMapSAMapper.cpp:461-551
      tooManyHits = false;
      readLen = rpair.first.seq.length();
      ++hc.numReads;
      leftHCInfo.clear(); rightHCInfo.clear();
      jointHits.clear(); leftHits.clear(); rightHits.clear();
      bool lh = hitCollector(rpair.first.seq, saSearcher, leftHCInfo);
      bool rh = hitCollector(rpair.second.seq, saSearcher, rightHCInfo);
      (void)lh; (void)rh;
      rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, MateStatus::PAIRED_END_LEFT, leftHCInfo, leftHits);
      rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, MateStatus::PAIRED_END_RIGHT, rightHCInfo, rightHits);
      if (fuzzy) rapmap::utils::mergeLeftRightHitsFuzzy(lh, rh, leftHits, rightHits, jointHits, mc, readLen, maxNumHits, tooManyHits, hc);
      else rapmap::utils::mergeLeftRightHits(leftHits, rightHits, jointHits, readLen, maxNumHits, tooManyHits, hc);
      if (jointHits.size() > maxNumHits) { jointHits.clear(); }
      hc.totHits += jointHits.size();
      if (count) {
        if (!jointHits.empty()) ++T.mapped;
        if (g_digest) for (size_t j = 0; j < jointHits.size(); ++j) T.digest += hit_digest(u, j, jointHits[j]);
      }
      ++u;
    }
    if (count) { const double te = now_s(); T.prefetchS += tp1 - tp0; T.loopS += te - tp1; if (T.firstAt == 0) T.firstAt = tp0; T.lastAt = te; }
  };
  if (warm) run_group(*warm, 0, false);               // context creation, first launches, buffer growth: outside the timed region
  ++ready;
  while (!go.load()) std::this_thread::yield();
  mine.goSeenAt = now_s();
  // up to `depth` groups of this worker are on their way at a time: the next group is packed and sent (prefetch_async) before the
  // per-read loop over the current one starts, so that the device pass of one hides under the host loop of the other
  std::deque<size_t> taken;
  auto take = [&]() {
    const size_t g = next.fetch_add(1);
    if (g >= groups.size()) return false;
    if (prefetch && depth > 1) hitCollector.prefetch_async(groups[g], mc, mixed && (g & 1), maxNumHits);    // <- the second added line
    taken.push_back(g);
    return true;
  };
  while ((int)taken.size() < depth && take()) {}
  while (!taken.empty()) {
    const size_t g = taken.front(); taken.pop_front();
    run_group(groups[g], firstUnit[g], true, mixed && (g & 1));        // --mixed: odd groups under --fuzzyIntersection
    take();
  }
  T = mine;
}

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: compat_bench INDEX READS.bin NPAIRS READLEN THREADS CHUNK [--no-prefetch] [--repeat R] [--use N]\n"); return 2; }
  try {
    const char* idx = argv[1]; const char* path = argv[2];
    const size_t nFile = (size_t)std::atoll(argv[3]), L = (size_t)std::atoll(argv[4]);
    const int threads = std::atoi(argv[5]); const size_t chunk = (size_t)std::atoll(argv[6]);
    bool prefetch = true, mixed = false, digestOnce = false; int repeat = 1, depth = 2; size_t n = nFile;
    for (int i = 7; i < argc; ++i) {
      if (!std::strcmp(argv[i], "--no-prefetch")) prefetch = false;
      else if (!std::strcmp(argv[i], "--mixed")) mixed = true;
      else if (!std::strcmp(argv[i], "--digest-once")) digestOnce = true;
      else if (!std::strcmp(argv[i], "--no-intervals")) g_keepIntervals = false;
      else if (!std::strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--depth") && i + 1 < argc) depth = std::max(1, std::atoi(argv[++i]));   // groups a worker has in flight (1: prefetch() alone)
      else if (!std::strcmp(argv[i], "--use") && i + 1 < argc) n = std::min(nFile, (size_t)std::atoll(argv[++i]));   // only the first N pairs of the file
    }
    std::vector<char> raw(2 * nFile * L);
    { FILE* f = std::fopen(path, "rb"); if (!f || std::fread(raw.data(), 1, raw.size(), f) != raw.size()) { std::fprintf(stderr, "cannot read %s\n", path); return 2; } std::fclose(f); }
    SAIndex32BitDense rmi;
    rmi.load(idx);
    const size_t ng = (n + chunk - 1) / chunk;
    std::vector<size_t> firstUnit(ng);
    for (size_t g = 0; g < ng; ++g) firstUnit[g] = g * chunk;
    auto build = [&](std::vector<Group>& groups) {
      groups.assign(ng, Group());
      std::atomic<size_t> nb{0};
      std::vector<std::thread> th;
      for (int t = 0; t < std::max(1, threads); ++t)
        th.emplace_back([&] {
          for (size_t g; (g = nb.fetch_add(1)) < ng;) {
            const size_t a = g * chunk, b = std::min(n, a + chunk);
            groups[g].resize(b - a);
            for (size_t u = a; u < b; ++u) { groups[g][u - a].first.seq.assign(raw.data() + u * L, L); groups[g][u - a].second.seq.assign(raw.data() + (nFile + u) * L, L); }
          }
        });
      for (auto& t : th) t.join();
    };
    double best = 0, secs = 0, joinS = 0; Totals tot; uint64_t ctr[5] = {0, 0, 0, 0, 0};
    uint64_t digest0 = 0;
    for (int rep = 0; rep < repeat; ++rep) {
      g_digest = !digestOnce || rep == 0;
      std::vector<Group> groups; build(groups);
      std::vector<Group> warm((size_t)threads);
      for (int t = 0; t < threads; ++t) { const Group& g0 = groups[(size_t)t % ng]; warm[(size_t)t] = g0; }
      std::vector<Totals> T((size_t)threads);
      rapmap::utils::HitCounters hctr;
      std::atomic<size_t> next{0}; std::atomic<int> ready{0}, go{0};
      std::vector<std::thread> th;
      for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] { worker(rmi, groups, firstUnit, next, prefetch, 200u, T[(size_t)t], hctr, &warm[(size_t)t], ready, go, mixed, depth); });
      while (ready.load() < threads) std::this_thread::yield();
      const auto t0 = std::chrono::steady_clock::now();
      const double t0s = now_s();
      go = 1;
      for (auto& t : th) t.join();
      const double joined = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      // The timed span ends when the LAST read group has been processed, not when the last worker thread has been joined: a
      // thread that exits in a process holding GPU mappings pays milliseconds of teardown (its stack's munmap runs the
      // driver's MMU notifiers; measured 6 ms per thread, serialised: profiles/r04/compat_probe_timeline.txt), which a caller
      // whose workers live as long as the run (the reference's do: src/RapMapSAMapper.cpp:752-799) pays once, after mapping.
      double dt = 0;
      for (auto& x : T) if (x.lastAt) dt = std::max(dt, x.lastAt - t0s);
      if (dt <= 0) dt = joined;
      if (std::getenv("COMPAT_BENCH_VERBOSE")) {
        double gmax = 0, fmin = 1e30, fmax = 0, lmin = 1e30, lmax = 0;
        for (auto& x : T) { gmax = std::max(gmax, x.goSeenAt - t0s); if (x.firstAt) { fmin = std::min(fmin, x.firstAt - t0s); fmax = std::max(fmax, x.firstAt - t0s); lmin = std::min(lmin, x.lastAt - t0s); lmax = std::max(lmax, x.lastAt - t0s); } }
        std::fprintf(stderr, "[compat_bench] repeat %d timeline (s after go): last thread saw go at %.4f; first group started %.4f .. %.4f; last group ended %.4f .. %.4f; all joined at %.4f\n",
                     rep, gmax, fmin, fmax, lmin, lmax, joined);
      }
      Totals S;
      if (std::getenv("COMPAT_BENCH_PHASES")) {
        uint64_t ph[5] = {0, 0, 0, 0, 0}; for (auto& x : T) for (int i = 0; i < 5; ++i) ph[i] += x.ph[i];
        std::fprintf(stderr, "[compat_bench] cycles per pair: clear %.0f, collector x2 %.0f, hitsToMappingsSimple x2 %.0f, merge %.0f, digest %.0f\n",
                     (double)ph[0] / n, (double)ph[1] / n, (double)ph[2] / n, (double)ph[3] / n, (double)ph[4] / n);
      }
      for (auto& x : T) { S.digest += x.digest; S.mapped += x.mapped; S.prefetchS += x.prefetchS; S.loopS += x.loopS; }
      if (rep == 0 || (double)n / dt > best) { best = (double)n / dt; secs = dt; joinS = joined - dt; }
      tot = S;
      if (rep == 0) digest0 = S.digest;
      if (digestOnce) { tot.digest = digest0; if (rep == 0 && repeat > 1) best = 0; }   // (the checked repeat is not the timed one)
      if (std::getenv("COMPAT_BENCH_VERBOSE")) {
        double mx = 0, mn = 1e30; for (auto& x : T) { const double b = x.prefetchS + x.loopS; mx = b > mx ? b : mx; mn = b < mn ? b : mn; }
        std::fprintf(stderr, "[compat_bench] repeat %d: wall %.4f s, per thread busy min %.4f max %.4f s (prefetch %.3f + loop %.3f thread-s over %d threads)\n",
                     rep, dt, mn, mx, S.prefetchS, S.loopS, threads);
      }
      ctr[0] = hctr.peHits.load(); ctr[1] = hctr.seHits.load(); ctr[2] = hctr.totHits.load(); ctr[3] = hctr.numReads.load(); ctr[4] = hctr.tooManyHits.load();
    }
    std::printf("{\"pairs\": %zu, \"read_len\": %zu, \"threads\": %d, \"chunk\": %zu, \"prefetch\": %s, \"seconds\": %.6f, \"thread_join_seconds\": %.6f, \"mpairs_per_s\": %.4f, "
                "\"prefetch_thread_s\": %.4f, \"loop_thread_s\": %.4f, \"digest\": \"%016llx\", \"mapped\": %llu, \"peHits\": %llu, \"seHits\": %llu, \"totHits\": %llu, \"numReads\": %llu, \"tooManyHits\": %llu}\n",
                n, L, threads, chunk, prefetch ? "true" : "false", secs, joinS, best / 1e6, tot.prefetchS, tot.loopS, (unsigned long long)tot.digest, (unsigned long long)tot.mapped,
                (unsigned long long)ctr[0], (unsigned long long)ctr[1], (unsigned long long)ctr[2], (unsigned long long)ctr[3], (unsigned long long)ctr[4]);
    return 0;
  } catch (const qmap::Error& e) { std::printf("qmap error %d: %s\n", e.code(), e.what()); return 3; }
}
