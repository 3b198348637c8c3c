// openjph_amd/csrc/ht_tables.cpp -- see ht_tables.h
#include "ht_tables.h"
#include <string.h>

#include "ht_vlc_tables.inc"

namespace ojphgpu {

namespace {
inline int r_cq(unsigned r) { return (int)(r & 7); }
inline int r_rho(unsigned r) { return (int)((r >> 3) & 15); }
inline int r_uoff(unsigned r) { return (int)((r >> 7) & 1); }
inline int r_ek(unsigned r) { return (int)((r >> 8) & 15); }
inline int r_e1(unsigned r) { return (int)((r >> 12) & 15); }
inline int r_cwd(unsigned r) { return (int)((r >> 16) & 127); }
inline int r_len(unsigned r) { return (int)((r >> 23) & 7); }
}  // namespace

void build_ht_tables(HtTables& t)
{
  memset(&t, 0, sizeof(t));
  const unsigned* src[2] = { HT_VLC_SRC0, HT_VLC_SRC1 };
  const int n[2] = { (int)(sizeof(HT_VLC_SRC0) / sizeof(unsigned)), (int)(sizeof(HT_VLC_SRC1) / sizeof(unsigned)) };
  for (int k = 0; k < 2; ++k) {
    // encoder: for every (context, significance pattern, exponent-max pattern) choose the row
    // the reference chooses: u_off = 0 rows when no sample attains the bound, otherwise the
    // compatible u_off = 1 row whose e_k has most bits (last one wins ties).
    for (int i = 0; i < 2048; ++i) {
      const int c_q = i >> 8, rho = (i >> 4) & 15, emb = i & 15;
      if ((emb & rho) != emb || (rho == 0 && c_q == 0)) continue;
      int best = -1, best_cnt = -1;
      for (int j = 0; j < n[k]; ++j) {
        const unsigned r = src[k][j];
        if (r_cq(r) != c_q || r_rho(r) != rho) continue;
        if (emb) {
          if (r_uoff(r) == 1 && (emb & r_ek(r)) == r_e1(r)) {
            const int cnt = __builtin_popcount((unsigned)r_ek(r));
            if (cnt >= best_cnt) { best = j; best_cnt = cnt; }
          }
        } else if (r_uoff(r) == 0) { best = j; break; }
      }
      if (best >= 0) {
        const unsigned r = src[k][best];
        t.enc_vlc[k][i] = (uint16_t)((r_cwd(r) << 8) | (r_len(r) << 4) | r_ek(r));
      }
    }
    // decoder: 3 context bits + the next 7 stream bits
    for (int i = 0; i < 1024; ++i) {
      const int cwd = i & 0x7F, c_q = i >> 7;
      for (int j = 0; j < n[k]; ++j) {
        const unsigned r = src[k][j];
        if (r_cq(r) == c_q && r_cwd(r) == (cwd & ((1 << r_len(r)) - 1)))
          t.dec_vlc[k][i] = (uint16_t)((r_rho(r) << 4) | (r_uoff(r) << 3) | (r_ek(r) << 12) | (r_e1(r) << 8) | r_len(r));
      }
      const unsigned e = t.dec_vlc[k][i], rho = (e >> 4) & 15u, e1 = (e >> 8) & 15u, ek = (e >> 12) & 15u;
      unsigned packed = (rho & (rho - 1u)) ? 0x100u : 0u;
      for (int s = 0; s < 4; ++s) packed |= (((rho >> s) & 1u) + ((ek >> s) & 1u) + ((e1 >> s) & 1u)) << (2 * s);
      // bits 8..10 (e_1 / e_k live in the upper half here): what the CHAIN derives from rho for its next look-ups, ready made --
      // bit 8 = a significant sample in the quad's right column (rho bits 2 | 3: the next quad's context bit 8 in rows below
      // the first), bits 9, 10 = its bottom-row samples (rho bits 1, 3: the row below's neighbourhood)
      const unsigned chain = (((rho >> 2) | (rho >> 3)) & 1u) << 8 | ((rho >> 1) & 1u) << 9 | ((rho >> 3) & 1u) << 10;
      t.dec_vlc32[k][i] = (e & 0xFFu) | chain | (packed << 16);
    }
  }
  // U-VLC prefix (T.814 table 3), indexed by the next 3 stream bits:
  //   prefix length | suffix length << 2 | u_pfx << 5
  static const uint8_t pfx[8] = {
    3 | (5 << 2) | (5 << 5), 1 | (0 << 2) | (1 << 5), 2 | (0 << 2) | (2 << 5), 1 | (0 << 2) | (1 << 5),
    3 | (1 << 2) | (3 << 5), 1 | (0 << 2) | (1 << 5), 2 | (0 << 2) | (2 << 5), 1 | (0 << 2) | (1 << 5) };
  auto pack = [](unsigned tp, unsigned ts, unsigned s0, unsigned u0, unsigned u1) {
    return (uint16_t)(tp | (ts << 3) | (s0 << 7) | (u0 << 10) | (u1 << 13));
  };
  for (int i = 0; i < 320; ++i) {
    const int mode = i >> 6; const unsigned vlc = (unsigned)i & 0x3F;
    const unsigned d0 = pfx[vlc & 7], d1 = pfx[(vlc >> (d0 & 3)) & 7];
    const unsigned both_tp = (d0 & 3) + (d1 & 3), s0 = (d0 >> 2) & 7, both_ts = s0 + ((d1 >> 2) & 7);
    uint16_t first = 0, other = 0;
    if (mode == 1) first = other = pack(d0 & 3, s0, s0, d0 >> 5, 0);
    else if (mode == 2) first = other = pack(d0 & 3, s0, 0, 0, d0 >> 5);
    else if (mode == 3) {
      other = pack(both_tp, both_ts, s0, d0 >> 5, d1 >> 5);
      // initial row, MEL event 0: if u_q0 > 2 the second quad is a single bit (u_q1 in {1,2})
      if ((d0 & 3) == 3) first = pack((d0 & 3) + 1, s0, s0, d0 >> 5, ((vlc >> (d0 & 3)) & 1) + 1);
      else first = other;
    } else if (mode == 4) first = pack(both_tp, both_ts, s0, (d0 >> 5) + 2, (d1 >> 5) + 2);
    t.dec_uvlc0[i] = first;
    if (i < 256) t.dec_uvlc1[i] = other;
  }
}

}  // namespace ojphgpu
