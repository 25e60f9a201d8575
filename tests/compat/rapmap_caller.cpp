// tests/compat/rapmap_caller.cpp -- a caller written against RapMap's own call sequence (the body of processReadsPairSA,
// /root/reference/src/RapMapSAMapper.cpp:376-551: collector x2 -> hitsToMappingsSimple x2 -> mergeLeftRightHits[Fuzzy] ->
// the maxNumHits / noOrphans bookkeeping), compiled against include/qmap_rapmap_compat.hpp ALONE.  Only the names the
// reference declares are used; the single addition is the `hitCollector.prefetch(rg)` line (skipped with --no-prefetch:
// every call is then a batch of one on the device).
//
//   rapmap_caller INDEX PAIRS.txt OUT.txt [--fuzzy] [--chain] [--no-prefetch] [--noOrphans] [--maxNumHits N] [--edit] [--refill] [--evens-first]
//
// PAIRS.txt: "left right" per line.  OUT.txt: per pair "<n> tid:pos:matePos:fwd mateIsFwd:fragLen:mateStatus ..." and a
// last line with the HitCounters.  --edit: drops the last forward interval of every 5th left read between the collector
// and hitsToMappingsSimple (a caller that touches hcInfo): that read must be re-done from the edited intervals.
// --refill: after the prefetch, every 7th pair's left read is overwritten IN PLACE with the left read of the pair behind it when
// the two are equally long (a parser that refills its string buffers without a new prefetch: same address, same length, other
// characters): the stale chunk entry must not be handed out.  --evens-first: the chunk's pairs are processed 0, 2, 4, ... and
// then 1, 3, 5, ... (a caller that skips reads and comes back): answers still belong to the read that was asked about.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "qmap_rapmap_compat.hpp"

struct Read { std::string seq; };
using ReadPair = std::pair<Read, Read>;

template <typename RapMapIndexT>
int run(RapMapIndexT& rmi, std::vector<ReadPair>& all, const char* outPath, bool fuzzy, bool chain, bool prefetch, bool noOrphans,
        uint32_t maxNumHits, bool edit, bool refill, bool evensFirst, bool noIntervals = false) {
  using OffsetT = typename RapMapIndexT::IndexType;
  using rapmap::utils::MateStatus;
  using rapmap::utils::QuasiAlignment;
  // ---- src/RapMapSAMapper.cpp:385-440
  rapmap::utils::HitCounters hctr;
  std::vector<QuasiAlignment> leftHits, rightHits, jointHits;
  SACollector<RapMapIndexT> hitCollector(&rmi);
  hitCollector.disableNIP();                    // sensitive (the CLI default)
  hitCollector.setStrictCheck(true);
  if (noIntervals) hitCollector.setKeepIntervals(false);     // (not in the reference: "I never look inside HitCollectorInfo")
  rapmap::hit_manager::HitCollectorInfo<rapmap::utils::SAIntervalHit<OffsetT>> leftHCInfo, rightHCInfo;
  rapmap::utils::MappingConfig mc;
  mc.consistentHits = false;
  mc.doChaining = chain;
  if (mc.doChaining) {
    float consensusSlack = 0.2f;
    mc.consensusFraction = (consensusSlack == 0.0) ? 1.0 : (1.0 - consensusSlack);
    mc.considerMultiPos = true;
    hitCollector.enableChainScoring();
    hitCollector.setMaxMMPExtension(7);
  }
  bool useSmartIntersect = fuzzy || chain;
  SASearcher<RapMapIndexT> saSearcher(&rmi);
  std::ofstream outFile(outPath);
  bool tooManyHits = false;
  uint32_t readLen = 0;
  const size_t chunkSize = 5000;                // the parser's read groups
  size_t pairNo = 0;
  for (size_t c0 = 0; c0 < all.size(); c0 += chunkSize) {
    std::vector<ReadPair> rg(all.begin() + c0, all.begin() + std::min(all.size(), c0 + chunkSize));
    if (prefetch) hitCollector.prefetch(rg, mc, fuzzy, maxNumHits);           // <- the one added line
    if (refill)
      for (size_t i = 0; i + 1 < rg.size(); i += 7)
        if (rg[i].first.seq.size() == rg[i + 1].first.seq.size())
          std::memcpy(&rg[i].first.seq[0], rg[i + 1].first.seq.data(), rg[i].first.seq.size());
    std::vector<size_t> order;
    for (size_t i = 0; i < rg.size(); i += evensFirst ? 2 : 1) order.push_back(i);
    if (evensFirst) for (size_t i = 1; i < rg.size(); i += 2) order.push_back(i);
    std::vector<std::string> lines(rg.size());
    for (size_t oi : order) {
      auto& rpair = rg[oi];
      std::ostringstream out;
      // ---- src/RapMapSAMapper.cpp:461-551
      tooManyHits = false;
      readLen = rpair.first.seq.length();
      ++hctr.numReads;
      leftHCInfo.clear();
      rightHCInfo.clear();
      jointHits.clear();
      leftHits.clear();
      rightHits.clear();

      bool lh = hitCollector(rpair.first.seq, saSearcher, leftHCInfo);
      bool rh = hitCollector(rpair.second.seq, saSearcher, rightHCInfo);

      if (edit && pairNo % 5 == 0 && !leftHCInfo.fwdSAInts.empty()) leftHCInfo.fwdSAInts.pop_back();

      rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, MateStatus::PAIRED_END_LEFT, leftHCInfo, leftHits);
      rapmap::hit_manager::hitsToMappingsSimple(rmi, mc, MateStatus::PAIRED_END_RIGHT, rightHCInfo, rightHits);

      if (useSmartIntersect) {
        rapmap::utils::mergeLeftRightHitsFuzzy(lh, rh, leftHits, rightHits, jointHits, mc, readLen, maxNumHits, tooManyHits, hctr);
      } else {
        rapmap::utils::mergeLeftRightHits(leftHits, rightHits, jointHits, readLen, maxNumHits, tooManyHits, hctr);
      }
      // If the read mapped to > maxReadOccs places, discard it
      if (jointHits.size() > maxNumHits) { jointHits.clear(); }
      if (!jointHits.empty()) {
        bool isPaired = jointHits.front().mateStatus == rapmap::utils::MateStatus::PAIRED_END_PAIRED;
        if (noOrphans) { if (!isPaired) { jointHits.clear(); } }
      }
      hctr.totHits += jointHits.size();
      out << jointHits.size();
      for (auto& q : jointHits) {
        const bool paired = q.mateStatus == MateStatus::PAIRED_END_PAIRED;
        out << ' ' << q.tid << ':' << q.pos << ':' << (paired ? q.matePos : 0) << ':' << q.fwd << (paired ? q.mateIsFwd : true) << ':'
            << (paired ? q.fragLen : 0u) << ':' << (int)q.mateStatus;
      }
      out << '\n';
      lines[oi] = out.str();
      ++pairNo;
    }
    for (auto& l : lines) outFile << l;
  }
  outFile << "counters " << hctr.peHits.load() << ' ' << hctr.seHits.load() << ' ' << hctr.totHits.load() << ' ' << hctr.numReads.load() << ' '
      << hctr.tooManyHits.load() << '\n';
  return 0;
}

template <typename RapMapIndexT>
int mainT(int argc, char** argv) {
  RapMapIndexT rmi;
  rmi.load(argv[1]);
  std::printf("k %u txps %zu ph %d big %d\n", rmi.k(), rmi.txpNames.size(), (int)rmi.perfectHash(), (int)(sizeof(typename RapMapIndexT::IndexType) == 8));
  if (argc < 4) return 0;
  bool fuzzy = false, chain = false, prefetch = true, noOrphans = false, edit = false, refill = false, evensFirst = false, noIntervals = false; uint32_t maxNumHits = 200;
  for (int i = 4; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--fuzzy")) fuzzy = true;
    else if (!std::strcmp(argv[i], "--chain")) chain = true;
    else if (!std::strcmp(argv[i], "--no-prefetch")) prefetch = false;
    else if (!std::strcmp(argv[i], "--noOrphans")) noOrphans = true;
    else if (!std::strcmp(argv[i], "--edit")) edit = true;
    else if (!std::strcmp(argv[i], "--refill")) refill = true;
    else if (!std::strcmp(argv[i], "--evens-first")) evensFirst = true;
    else if (!std::strcmp(argv[i], "--no-intervals")) noIntervals = true;
    else if (!std::strcmp(argv[i], "--maxNumHits") && i + 1 < argc) maxNumHits = (uint32_t)std::atoi(argv[++i]);
  }
  std::vector<ReadPair> all;
  std::ifstream f(argv[2]); std::string a, b;
  while (f >> a >> b) { ReadPair p; p.first.seq = a == "-" ? "" : a; p.second.seq = b == "-" ? "" : b; all.push_back(p); }
  return run(rmi, all, argv[3], fuzzy, chain, prefetch, noOrphans, maxNumHits, edit, refill, evensFirst, noIntervals);
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: rapmap_caller INDEX [PAIRS OUT [flags]]\n"); return 2; }
  try {
    // the instantiation is picked from header.json like `rapmap quasimap` does (src/RapMapSAMapper.cpp:1209-1240)
    std::ifstream hf(std::string(argv[1]) + "/header.json");
    std::string hs((std::istreambuf_iterator<char>(hf)), std::istreambuf_iterator<char>());
    const size_t bp = hs.find("\"BigSA\"");
    const bool big = bp != std::string::npos && hs.find("true", bp) < hs.find(',', bp);
    return big ? mainT<SAIndex64BitDense>(argc, argv) : mainT<SAIndex32BitDense>(argc, argv);   // RapMapSAIndex<int64_t | int32_t, RegHashT>
  } catch (const qmap::Error& e) { std::printf("qmap error %d: %s\n", e.code(), e.what()); return 3; }
}
