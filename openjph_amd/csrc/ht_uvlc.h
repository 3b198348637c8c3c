// openjph_amd/csrc/ht_uvlc.h -- U-VLC of a quad pair in quad rows other than the first, by arithmetic.
//
// The reference looks the two U-VLC prefixes of a pair up in uvlc_tbl1 (ojph_block_common.cpp:294-336, used at
// ojph_block_decoder32.cpp:1065-1085).  In the GPU decoder that look-up sits on the serial chain of step 1 -- where
// the next codeword starts depends on its result -- and an LDS access costs that chain more than the dozen ALU
// operations below.  T.814 table 3: prefix "1" -> u_pfx 1; "01" -> 2; "001" -> 3 + 1 suffix bit; "000" -> 5 +
// 5 suffix bits; the prefixes of both quads come first, then the suffixes.  tests/test_uvlc_alu.py compares this
// function with the table for every index.
#ifndef OJPH_HT_UVLC_H
#define OJPH_HT_UVLC_H
#include <stdint.h>

#if defined(__HIPCC__)
#define OJPH_HD __host__ __device__ __forceinline__
#else
#define OJPH_HD inline
#endif

namespace ojphgpu {

// one byte per number of trailing zeros of the next three bits (3 = none of them set): len | u_pfx << 2 | suffix len << 5
constexpr uint32_t UVLC_PFX_BYTES = (1u | 1u << 2 | 0u << 5) | (2u | 2u << 2 | 0u << 5) << 8 | (3u | 3u << 2 | 1u << 5) << 16 |
                                    (3u | 5u << 2 | 5u << 5) << 24;

OJPH_HD uint32_t uvlc_prefix(uint32_t bytes, uint32_t v)     // bytes: UVLC_PFX_BYTES, or 0 for a quad without U-VLC
{
  const uint32_t z8 = (uint32_t)__builtin_ctz(v | 8u) << 3;
  return (bytes >> z8) & 0xFFu;
}

// v: the stream bits behind the pair's two VLC codewords (LSB first, >= 16 of them); uoff0 / uoff1: non-zero when the
// quad has a U-VLC.  Returns the number of bits consumed; u0 / u1 = u_q - kappa of the two quads.
OJPH_HD uint32_t uvlc_pair_other_rows(uint32_t v, uint32_t uoff0, uint32_t uoff1, uint32_t& u0, uint32_t& u1)
{
  const uint32_t f0 = uvlc_prefix(uoff0 ? UVLC_PFX_BYTES : 0u, v);
  v >>= f0 & 3u;
  const uint32_t f1 = uvlc_prefix(uoff1 ? UVLC_PFX_BYTES : 0u, v);
  v >>= f1 & 3u;
  const uint32_t s0 = f0 >> 5, s1 = f1 >> 5;
  const uint32_t tmp = v & ~(0xFFFFFFFFu << (s0 + s1));
  u0 = ((f0 >> 2) & 7u) + (tmp & ~(0xFFFFFFFFu << s0));
  u1 = ((f1 >> 2) & 7u) + (tmp >> s0);
  return (f0 & 3u) + (f1 & 3u) + s0 + s1;
}

}  // namespace ojphgpu
#endif
