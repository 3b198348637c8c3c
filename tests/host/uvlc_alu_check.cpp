// Exhaustive check of the arithmetic U-VLC decoder (openjph_amd/csrc/ht_uvlc.h) against the look-up table it replaces
// (dec_uvlc1 of ht_tables.cpp == the reference's uvlc_tbl1): every mode, every 16-bit continuation of the stream.
#include <cstdio>
#include "../../openjph_amd/csrc/ht_tables.h"
#include "../../openjph_amd/csrc/ht_uvlc.h"

int main()
{
  static ojphgpu::HtTables t;
  ojphgpu::build_ht_tables(t);
  unsigned long checked = 0;
  for (unsigned mode = 0; mode < 4; ++mode)
    for (unsigned bits = 0; bits < (1u << 16); ++bits) {
      // the table path, as ojph_block_decoder32.cpp:1065-1085 walks it
      unsigned v = bits, used = 0;
      unsigned entry = t.dec_uvlc1[(mode << 6) + (v & 0x3F)];
      v >>= (entry & 7u); used += entry & 7u; entry >>= 3;
      unsigned len = entry & 0xFu;
      const unsigned tmp = v & ((1u << len) - 1u);
      used += len; entry >>= 4;
      len = entry & 7u; entry >>= 3;
      const unsigned u0 = (entry & 7u) + (tmp & ~(0xFFu << len));
      const unsigned u1 = (entry >> 3) + (tmp >> len);
      uint32_t a0, a1;
      const uint32_t aused = ojphgpu::uvlc_pair_other_rows(bits, mode & 1u, mode & 2u, a0, a1);
      if (aused != used || a0 != u0 || a1 != u1) {
        std::printf("MISMATCH mode %u bits %04x: table used %u u0 %u u1 %u, alu used %u u0 %u u1 %u\n", mode, bits, used, u0, u1, aused, a0, a1);
        return 1;
      }
      ++checked;
    }
  std::printf("OK %lu\n", checked);
  return 0;
}
