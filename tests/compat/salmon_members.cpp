// tests/compat/salmon_members.cpp -- a translation unit that uses what Salmon reads from rapmap::utils::QuasiAlignment under
// RAPMAP_SALMON_SUPPORT (the reference's include/RapMapUtils.hpp:41-43,407-421,446-467): compiled by tests/test_compat_header.py
// against include/qmap_rapmap_compat.hpp alone.  LibraryFormat is Salmon's own type; a minimal one stands in for its header here.
#include <cstdint>
struct LibraryFormat {
  uint8_t id;
  static LibraryFormat formatFromID(uint8_t i) { return LibraryFormat{i}; }
};
#define RAPMAP_SALMON_SUPPORT 1
#include "qmap_rapmap_compat.hpp"

int main() {
  using rapmap::utils::QuasiAlignment;
  using rapmap::utils::MateStatus;
  QuasiAlignment a;                                  // default: format 0, log-probabilities unset
  if (a.libFormat().id != 0 || a.logProb != HUGE_VAL || a.logBias != HUGE_VAL) return 1;
  QuasiAlignment q(7, 100, true, 100, 250, true);
  q.matePos = 250; q.mateLen = 100; q.mateIsFwd = false; q.mateStatus = MateStatus::PAIRED_END_PAIRED;
  if (q.fragLengthPedantic(1000) != 250) return 2;   // fwd read at 100, mate's end at 350
  if (q.fragLengthPedantic(300) != 200) return 3;    // clipped at the transcript's end
  q.mateIsFwd = true;
  if (q.fragLengthPedantic(1000) != 0) return 4;     // same orientation: no fragment
  q.logProb = -1.5; q.logBias = 0.25; q.format = LibraryFormat::formatFromID(5);
  QuasiAlignment c = q;
  if (c.logProb != -1.5 || c.logBias != 0.25 || c.libFormat().id != 5) return 5;
  return 0;
}
