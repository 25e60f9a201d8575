// qm_pack.h -- the host-side packer of the 2-bit read format (include/qmap_mi355.h, "2-bit packed reads"): shared by the public
// qm_pack_reads (qm_io.cpp) and the ingest engine's copy tasks (qm_ingest.cpp).
#pragma once
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/qmap_mi355.h"

namespace qm_pack {

// A 0, C 1, G 2, T 3 (include/Kmer.hpp:40-51: (c >> 1) & 3 with the two high codes swapped); anything else: not valid
static inline bool code_of(unsigned char c, unsigned& code) {
  const unsigned x = (c >> 1) & 3u; code = x ^ (x >> 1);
  return c == 'A' || c == 'C' || c == 'G' || c == 'T';
}

// Characters [s, s + len) -> ceil(len / 4) bytes at dst; exceptions reported through `on_exc(index in the read, character)`.
template <typename OnExc>
static inline void pack_scalar(const unsigned char* s, size_t len, uint8_t* dst, OnExc on_exc, size_t from = 0) {
  for (size_t i = from; i < len; i += 4) {
    unsigned b = 0;
    for (size_t j = 0; j < 4 && i + j < len; ++j) {
      unsigned c; if (!code_of(s[i + j], c)) { on_exc(i + j, s[i + j]); c = 0; }
      b |= c << (2 * j);
    }
    dst[i >> 2] = (uint8_t)b;
  }
}

// 32 characters per step: compare against the four letters, codes by shifts, four codes to a byte by two multiply-adds
template <typename OnExc>
__attribute__((target("avx2"))) static inline void pack_avx2(const unsigned char* s, size_t len, uint8_t* dst, OnExc on_exc) {
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
  const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1);
  const __m256i m14 = _mm256_set1_epi16(0x0401), m116 = _mm256_set1_epi32(0x00100001);
  const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  size_t i = 0;
  for (; i + 32 <= len; i += 32) {
    const __m256i c = _mm256_loadu_si256((const __m256i*)(s + i));
    const __m256i valid = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(c, cA), _mm256_cmpeq_epi8(c, cC)), _mm256_or_si256(_mm256_cmpeq_epi8(c, cG), _mm256_cmpeq_epi8(c, cT)));
    const __m256i x = _mm256_and_si256(_mm256_srli_epi16(c, 1), three);
    const __m256i code = _mm256_and_si256(_mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), one)), valid);
    const __m256i t = _mm256_maddubs_epi16(code, m14);             // c0 + 4 c1 per 16-bit lane
    const __m256i u = _mm256_madd_epi16(t, m116);                  // + 16 (c2 + 4 c3) per 32-bit lane: one packed byte
    const __m256i g = _mm256_shuffle_epi8(u, gather);
    const uint32_t lo = (uint32_t)_mm256_extract_epi32(g, 0), hi = (uint32_t)_mm256_extract_epi32(g, 4);
    __builtin_memcpy(dst + (i >> 2), &lo, 4); __builtin_memcpy(dst + (i >> 2) + 4, &hi, 4);
    unsigned bad = ~(unsigned)_mm256_movemask_epi8(valid);
    while (bad) { const unsigned j = (unsigned)__builtin_ctz(bad); on_exc(i + j, s[i + j]); bad &= bad - 1; }
  }
  if (i < len) pack_scalar(s, len, dst, on_exc, i);
}

static inline bool have_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }

template <typename OnExc>
static inline void pack_read(const unsigned char* s, size_t len, uint8_t* dst, OnExc on_exc) {
  if (len >= 32 && have_avx2()) pack_avx2(s, len, dst, on_exc); else pack_scalar(s, len, dst, on_exc);
}

}  // namespace qm_pack
