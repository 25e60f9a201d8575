// qm_phflat.h -- host-side re-blocking of the BooPHF level bit arrays (include/BooPHF.hpp) for the device.
// The reference keeps, per level, a bit array plus one rank sample per 512 bits in a separate vector, so a
// lookup that finds its bit set touches two unrelated places.  Here every 64-byte block (= one HBM sector)
// carries 384 bits of a level, the rank of its first bit and the popcount prefix of its six words:
//   word 0..5  the bits        word 6  rank of the block's first bit (continues across levels, like the
//   reference's bitVector::build_ranks(offset))     word 7  six 9-bit prefix popcounts (word w at bits 9w..)
// so "is my bit set, and what is its rank" is answered by one sector.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace qm {

struct PhLevelIn { const uint64_t* words; uint64_t nchar; uint64_t domain; const uint64_t* ranks; uint64_t nranks; };

#ifndef QM_PH_BLOCK_BITS
#define QM_PH_BLOCK_BITS 384
#endif

// blocks: 8 u64 per block, levels concatenated; tab[2*i] = hash domain of level i, tab[2*i+1] = its first block.
// Returns false if the recomputed ranks disagree with the rank samples stored in the index.
inline bool ph_flatten_blocks(const std::vector<PhLevelIn>& levels, std::vector<uint64_t>& blocks, std::vector<uint64_t>& tab) {
  blocks.clear(); tab.assign(2 * levels.size(), 0);
  uint64_t rank = 0;
  bool ok = true;
  for (size_t li = 0; li < levels.size(); ++li) {
    const PhLevelIn& L = levels[li];
    tab[2 * li] = L.domain; tab[2 * li + 1] = blocks.size() / 8;
    const uint64_t nbits = L.nchar * 64;
    const uint64_t nblk = (nbits + QM_PH_BLOCK_BITS - 1) / QM_PH_BLOCK_BITS;
    for (uint64_t b = 0; b < nblk; ++b) {
      uint64_t w[8] = {0, 0, 0, 0, 0, 0, rank, 0};
      uint64_t pref = 0, cnt = 0;
      for (int t = 0; t < 6; ++t) {
        const uint64_t wi = b * 6 + t;
        // the reference samples the rank every 512 bits (8 words): cross-check where the grids coincide
        if (wi < L.nchar && (wi & 7) == 0 && (wi >> 3) < L.nranks && L.ranks[wi >> 3] != rank + cnt) ok = false;
        w[t] = wi < L.nchar ? L.words[wi] : 0;
        pref |= cnt << (9 * t);
        cnt += (uint64_t)__builtin_popcountll(w[t]);
      }
      w[7] = pref;
      rank += cnt;
      blocks.insert(blocks.end(), w, w + 8);
    }
  }
  return ok;
}

}  // namespace qm
