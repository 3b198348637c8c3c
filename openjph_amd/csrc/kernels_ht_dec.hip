// openjph_amd/csrc/kernels_ht_dec.hip -- HT block decoder (cleanup pass) for gfx950 with the
// de-quantise transfer fused into its sample stores.  ONE WAVEFRONT PER CODE-BLOCK.
//
// Reference: ojph_decode_codeblock32 (src/core/coding/ojph_block_decoder32.cpp:742-1316):
//   MEL reader/decoder :93-269, backward VLC reader :308-439, forward MagSgn reader :581-723,
//   step 1 (MEL + VLC + U-VLC -> per-quad {rho, e_1, e_k, u}) :854-1089,
//   step 2 (MagSgn -> samples) :1091-1316;
// de-quantise transfer gen_rev/irv_tx_from_cb32 (src/core/codestream/ojph_codestream_gen.cpp:
// 124-168); zero blocks / failures: codeblock::decode + pull_line (ojph_codeblock.cpp:190-266).
//
// Wavefront formulation:
//   A. all 64 lanes de-stuff the three byte segments into flat LDS bit buffers.  Un-stuffing is
//      a function of adjacent raw bytes only (0xFF -> next byte has 7 bits; >0x8F then 0x7F
//      when reading backwards), so the bit offset of every byte is a wavefront prefix sum --
//      the same idea the reference's AVX2 decoder uses (ojph_block_decoder_avx2.cpp:277-386).
//   B. step 1 is inherently serial (each codeword's position depends on all earlier ones and
//      on the context): it runs as wave-uniform code on register bit windows that are refilled
//      from the flat buffers, with the tables in LDS.  Throughput comes from thousands of
//      code-blocks in flight, not from this loop.
//   C. step 2 is parallel per quad row: lane = quad, kappa from the exponents of the row above
//      (kept in LDS), bit offsets from a prefix sum of the m_n, samples extracted from the flat
//      MagSgn buffer and written straight to the sub-band plane, de-quantised.
// SigProp / MagRef passes (:1318-1609) are not implemented yet: blocks carrying them decode
// their cleanup pass only -- the same result as the reference for its own encoder's streams,
// which never contain these passes (ojph_block_encoder.cpp:548).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ojphgpu.h"
#include "ht_tables.h"

namespace ojphgpu {
__device__ uint16_t g_dec_vlc[2][1024];
__device__ uint16_t g_dec_uvlc0[320];
__device__ uint16_t g_dec_uvlc1[256];
}

namespace {

constexpr int WAVES = 4;
constexpr uint32_t MEL_FLAT_BYTES = 1280;   // > (1024 + 512 events) * 6 bits
constexpr uint32_t VLC_FLAT_BYTES = 2048;   // > 512 pairs * 30 bits
constexpr uint32_t EXP_BYTES = 1024 + 8;    // msb index of v_n per column of one sample row (two rows kept)

__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t rdfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

__device__ __forceinline__ void or_bits(uint32_t* buf, uint32_t pos, uint32_t v, uint32_t n)
{
  if (n == 0) return;
  const uint32_t w = pos >> 5, sh = pos & 31;
  atomicOr(&buf[w], v << sh);
  if (sh + n > 32) atomicOr(&buf[w + 1], v >> (32 - sh));
}

__device__ __forceinline__ uint32_t get32(const uint32_t* buf, uint32_t pos)
{
  const uint32_t w = pos >> 5, sh = pos & 31;
  return __funnelshift_r(buf[w], buf[w + 1], sh);
}

// De-stuffs n bytes src[0..n) (forward order, "after 0xFF only 7 bits") into the flat LSB-first
// bit buffer; at most cap_bits are kept.  msb_first: the stream is read MSB first (MEL), so every
// byte is bit-reversed on the way in.  or_last: OR-ed into byte n-1 (MEL/VLC overlap rule).
// Returns the number of bits written (wave-uniform).
__device__ uint32_t destuff_forward(const uint8_t* __restrict__ src, uint32_t n, uint32_t* flat,
                                    uint32_t cap_bits, bool msb_first, uint32_t or_last, int lane)
{
  uint32_t total = 0;
  for (uint32_t base = 0; base < n && total < cap_bits; base += 256) {
    const uint32_t i0 = base + 4u * (uint32_t)lane;
    uint32_t prev = (i0 > 0 && i0 - 1 < n) ? src[i0 - 1] : 0u;
    if (i0 > 0 && i0 - 1 == n - 1) prev |= or_last;
    uint32_t val = 0, nb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = i0 + k;
      if (i < n) {
        uint32_t b = src[i];
        if (i == n - 1) b |= or_last;
        const uint32_t bits = prev == 0xFF ? 7u : 8u;
        uint32_t v = b & ((1u << bits) - 1u);
        if (msb_first) v = __brev(v) >> (32 - bits);
        val |= v << nb; nb += bits;
        prev = b;
      }
    }
    const uint32_t incl = wave_incl_scan(nb, lane);
    const uint32_t at = total + incl - nb;
    if (at + nb <= cap_bits) or_bits(flat, at, val, nb);
    total += rdlane(incl, 63);
  }
  return total < cap_bits ? total : cap_bits;
}

// De-stuffs the VLC segment: bytes last-1 .. first going DOWN in memory, after the 4 (or 3)
// bits of the half byte at `last` (block_decoder32.cpp:374-405).
__device__ uint32_t destuff_backward(const uint8_t* __restrict__ top, uint32_t n, uint32_t first_prev,
                                     uint32_t* flat, uint32_t start_bits, uint32_t cap_bits, int lane)
{
  // top[-k], k = 0..n-1 are the bytes in reading order; first_prev is the byte "read before" top[0]
  uint32_t total = start_bits;
  for (uint32_t base = 0; base < n && total < cap_bits; base += 256) {
    const uint32_t i0 = base + 4u * (uint32_t)lane;
    uint32_t prev = i0 == 0 ? first_prev : (i0 - 1 < n ? (uint32_t) * (top - (i0 - 1)) : 0u);
    uint32_t val = 0, nb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = i0 + k;
      if (i < n) {
        const uint32_t b = *(top - i);
        const uint32_t bits = (prev > 0x8F && (b & 0x7F) == 0x7F) ? 7u : 8u;
        val |= (b & ((1u << bits) - 1u)) << nb; nb += bits;
        prev = b;
      }
    }
    const uint32_t incl = wave_incl_scan(nb, lane);
    const uint32_t at = total + incl - nb;
    if (at + nb <= cap_bits) or_bits(flat, at, val, nb);
    total += rdlane(incl, 63);
  }
  return total < cap_bits ? total : cap_bits;
}

// wave-uniform LSB-first bit window over a flat buffer; bits past `limit` read as `fill`
struct Window {
  uint64_t win; uint32_t bits, next_word, nwords;
  __device__ __forceinline__ void init(const uint32_t* buf, uint32_t words) {
    nwords = words; next_word = 0; win = 0; bits = 0; refill(buf); refill(buf);
  }
  __device__ __forceinline__ void refill(const uint32_t* buf) {
    if (bits <= 32) {
      const uint32_t w = next_word < nwords ? rdfirst(buf[next_word]) : 0u;
      next_word++;
      win |= (uint64_t)w << bits; bits += 32;
    }
  }
  __device__ __forceinline__ uint32_t peek() const { return (uint32_t)win; }
  __device__ __forceinline__ void skip(const uint32_t* buf, uint32_t n) { win >>= n; bits -= n; refill(buf); }
};

__device__ __forceinline__ uint32_t mel_exp(uint32_t k)
{
  const uint64_t tbl = 0ull | (1ull << 9) | (1ull << 12) | (1ull << 15) | (2ull << 18) | (2ull << 21) | (2ull << 24) |
                       (3ull << 27) | (3ull << 30) | (4ull << 33) | (5ull << 36);
  return (uint32_t)(tbl >> (3 * k)) & 7u;
}

struct MelDec {              // T.814 decodeMELSym; same runs as block_decoder32.cpp:170-269
  Window w; uint32_t k, run, one;
  __device__ __forceinline__ uint32_t sym(const uint32_t* buf) {
    if (run == 0 && one == 0) {
      const uint32_t e = mel_exp(k);
      const uint32_t v = w.peek();
      if (v & 1u) { run = 1u << e; k = k < 12 ? k + 1 : 12; w.skip(buf, 1); }
      else {
        const uint32_t raw = (v >> 1) & ((1u << e) - 1u);      // e bits, first stream bit at the LSB
        run = e ? (__brev(raw) >> (32 - e)) : 0u;              // the run count is sent MSB first
        k = k > 0 ? k - 1 : 0; one = 1; w.skip(buf, e + 1);
      }
    }
    if (run > 0) { run--; return 0; }
    one = 0; return 1;
  }
};

__global__ __launch_bounds__(64 * WAVES) void ht_decode_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    uint32_t* __restrict__ coef, uint8_t* __restrict__ block_status, uint32_t ms_words, uint32_t quad_words)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  // layout: tables | per wave { ms flat | vlc flat | mel flat | quad info | exponents }
  uint16_t* s_vlc = reinterpret_cast<uint16_t*>(smem);                    // 2 * 1024
  uint16_t* s_uvlc0 = s_vlc + 2048;                                       // 320
  uint16_t* s_uvlc1 = s_uvlc0 + 320;                                      // 256
  const uint32_t table_words = (2048 + 320 + 256) / 2;
  const uint32_t vlc_words = VLC_FLAT_BYTES / 4 + 2, mel_words = MEL_FLAT_BYTES / 4 + 2;
  const uint32_t wave_words = ms_words + vlc_words + mel_words + quad_words + 2 * (EXP_BYTES / 4);
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_vlc[i] = (&ojphgpu::g_dec_vlc[0][0])[i];
  for (int i = threadIdx.x; i < 320; i += blockDim.x) s_uvlc0[i] = ojphgpu::g_dec_uvlc0[i];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s_uvlc1[i] = ojphgpu::g_dec_uvlc1[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t bi = blockIdx.x * WAVES + wave;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  uint32_t* f_ms = smem + table_words + (uint32_t)wave * wave_words;
  uint32_t* f_vlc = f_ms + ms_words;
  uint32_t* f_mel = f_vlc + vlc_words;
  uint32_t* quad = f_mel + mel_words;
  uint8_t* vexp_a = reinterpret_cast<uint8_t*>(quad + quad_words);
  uint8_t* vexp_b = vexp_a + EXP_BYTES;

  const uint32_t W = d.w, H = d.h, pitch = d.pitch;
  if (W == 0 || H == 0) return;
  uint32_t* dst = coef + d.coef_off;
  const bool rev = d.reversible != 0;
  const uint32_t K = d.K_max;

  auto zero_block = [&]() {                                  // mem_clear path, ojph_codeblock.cpp:247
    for (uint32_t y = 0; y < H; ++y)
      for (uint32_t x = lane; x < W; x += 64) dst[(size_t)y * pitch + x] = 0u;
  };
  auto fail = [&](uint8_t code) { zero_block(); if (lane == 0) block_status[bi] = code; };

  if (lane == 0) block_status[bi] = 0;
  if (d.len1 == 0 || d.num_passes == 0) { zero_block(); return; }
  const uint32_t missing_msbs = d.missing_msbs;
  if (d.num_passes > 3 || missing_msbs >= 30 || d.len1 < 2) { fail(1); return; }   // :760-811
  const uint32_t p = 30 - missing_msbs;
  const uint8_t* cb = data + d.data_off;
  const uint32_t lcup = d.len1;
  const uint32_t scup = ((uint32_t)cb[lcup - 1] << 4) + (cb[lcup - 2] & 0xFu);      // :817
  if (scup < 2 || scup > lcup || scup > 4079) { fail(1); return; }
  const uint32_t ms_len = lcup - scup;
  if (((ms_len + 3) >> 2) + 2 > ms_words) { fail(2); return; }                       // LDS sized too small

  // ---- A. de-stuff the three segments ----
  for (uint32_t i = lane; i < ms_words + vlc_words + mel_words; i += 64) f_ms[i] = 0;
  wave_sync();
  const uint32_t ms_bits = destuff_forward(cb, ms_len, f_ms, (ms_words - 2) * 32, false, 0, lane);
  const uint32_t mel_bits = destuff_forward(cb + ms_len, scup - 1, f_mel, MEL_FLAT_BYTES * 8, true, 0xF, lane);
  {
    const uint32_t dd = cb[lcup - 2], t = dd >> 4;
    const uint32_t nb0 = 4u - ((t & 7u) == 7u ? 1u : 0u);                           // :382-385
    if (lane == 0) f_vlc[0] = t & ((1u << nb0) - 1u);
    wave_sync();
    destuff_backward(cb + lcup - 3, scup - 2, dd | 0xFu, f_vlc, nb0, VLC_FLAT_BYTES * 8, lane);
  }
  wave_sync();
  // MEL: past its end the stream continues with 0xFF bytes (:98); make the tail all ones
  {
    const uint32_t cap = MEL_FLAT_BYTES * 8;
    if (mel_bits < cap) {
      for (uint32_t w = (mel_bits >> 5) + (uint32_t)lane; w < mel_words; w += 64) {
        uint32_t m = 0xFFFFFFFFu;
        if (w == (mel_bits >> 5)) m <<= (mel_bits & 31);
        atomicOr(&f_mel[w], m);
      }
    }
  }
  wave_sync();

  // ---- B. step 1: wave-uniform serial decode of MEL / VLC / U-VLC ----
  const uint32_t QW = (W + 1) >> 1, QH = (H + 1) >> 1;
  const uint32_t qstr = QW + 1;                   // one spare quad right of the row (always zero)
  // quad[] holds QH rows of qstr entries; entry = t-word | (u_q << 16)
  if (QH * qstr > quad_words) { fail(2); return; }
  for (uint32_t i = lane; i < QH * qstr; i += 64) quad[i] = 0;
  wave_sync();
  {
    Window vw; vw.init(f_vlc, vlc_words);
    MelDec mel; mel.w.init(f_mel, mel_words); mel.k = 0; mel.run = 0; mel.one = 0;
    for (uint32_t qy = 0; qy < QH; ++qy) {
      uint32_t* row = quad + qy * qstr;
      const uint32_t* above = qy ? row - qstr : row;
      const uint16_t* tbl = s_vlc + (qy ? 1024 : 0);
      uint32_t tleft = 0;
      for (uint32_t qx = 0; qx < QW; qx += 2) {
        uint32_t t[2] = { 0, 0 };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t x = qx + j;
          if (x < QW) {
            uint32_t c_q;
            if (qy == 0) c_q = ((tleft & 0x10u) << 3) | ((tleft & 0xE0u) << 2);       // :903,:934
            else {
              c_q = ((tleft & 0x40u) << 2) | ((tleft & 0x80u) << 1);                  // :1022,:1059
              if (x > 0) c_q |= rdfirst(above[x - 1]) & 0x80u;                        // :1024,:1061
              c_q |= (rdfirst(above[x]) & 0xA0u) << 2;                                // :990,:1026
              c_q |= (rdfirst(above[x + 1]) & 0x20u) << 4;                            // :991,:1027
            }
            uint32_t tv = rdfirst((uint32_t)tbl[c_q + (vw.peek() & 0x7Fu)]);
            if (c_q == 0) { if (mel.sym(f_mel) == 0) tv = 0; }                        // :882-894
            vw.skip(f_vlc, tv & 7u);
            t[j] = tv; tleft = tv;
          }
        }
        uint32_t mode = ((t[0] & 0x8u) << 3) | ((t[1] & 0x8u) << 4);
        uint32_t entry;
        if (qy == 0) {
          if (mode == 0xC0u) { if (mel.sym(f_mel)) mode += 0x40u; }                   // :943-952
          entry = rdfirst((uint32_t)s_uvlc0[mode + (vw.peek() & 0x3Fu)]);
        } else entry = rdfirst((uint32_t)s_uvlc1[mode + (vw.peek() & 0x3Fu)]);
        vw.skip(f_vlc, entry & 7u); entry >>= 3;
        uint32_t len = entry & 0xFu;
        const uint32_t tmp = vw.peek() & ((1u << len) - 1u);
        vw.skip(f_vlc, len); entry >>= 4;
        len = entry & 7u; entry >>= 3;
        const uint32_t kap = qy == 0 ? 1u : 0u;                                       // :971-974 / :1082-1085
        const uint32_t u0 = kap + (entry & 7u) + (tmp & ~(0xFFu << len));
        const uint32_t u1 = kap + (entry >> 3) + (tmp >> len);
        if (lane == 0) {
          row[qx] = t[0] | (u0 << 16);
          if (qx + 1 < QW) row[qx + 1] = t[1] | (u1 << 16);
        }
      }
      wave_sync();                                 // the next row reads this one as "above"
    }
  }
  wave_sync();

  // ---- C. step 2: MagSgn, one quad per lane, one quad row at a time ----
  const uint32_t mmsbp2 = missing_msbs + 2;
  const uint32_t shift = 31 - K;
  const float delta = d.delta;
  for (uint32_t i = lane; i < 2 * EXP_BYTES; i += 64) vexp_a[i] = 0;
  wave_sync();
  uint32_t mpos = 0;                              // wave-uniform bit position in the flat MagSgn stream
  bool bad = false;
  for (uint32_t qy = 0; qy < QH && !bad; ++qy) {
    const uint8_t* vexp = (qy & 1) ? vexp_b : vexp_a;     // exponents of the sample row above
    uint8_t* vnew = (qy & 1) ? vexp_a : vexp_b;           // ... and of this quad row's bottom samples
    for (uint32_t qb = 0; qb < QW; qb += 64) {
      const uint32_t qx = qb + (uint32_t)lane;
      const bool act = qx < QW;
      const uint32_t ent = act ? quad[qy * qstr + qx] : 0u;
      const uint32_t inf = ent & 0xFFFFu;
      uint32_t U_q = ent >> 16;
      if (qy > 0 && act) {
        uint32_t gamma = inf & 0xF0u; gamma &= gamma - 0x10u;                           // :1218
        // exponents of columns 2qx-1 .. 2qx+2 of the sample row above (vexp is offset by 1)
        const uint32_t em = max(max((uint32_t)vexp[2 * qx], (uint32_t)vexp[2 * qx + 1]),
                                max((uint32_t)vexp[2 * qx + 2], (uint32_t)vexp[2 * qx + 3]));
        U_q += gamma ? max(em, 1u) : 1u;                                                // :1219-1223
      }
      if (__ballot(act && U_q > mmsbp2) != 0ull) { bad = true; break; }                 // :1114,:1224
      uint32_t m[4], tot = 0;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        m[nn] = (inf & (1u << (4 + nn))) ? U_q - ((inf >> (12 + nn)) & 1u) : 0u;
        tot += m[nn];
      }
      const uint32_t incl = wave_incl_scan(tot, lane);
      uint32_t at = mpos + incl - tot;
      mpos += rdlane(incl, 63);
      uint32_t out[4];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        uint32_t val = 0, v_n = 0;
        if (inf & (1u << (4 + nn))) {
          uint32_t ms_val;
          if (at >= ms_bits) ms_val = 0xFFFFFFFFu;                                      // exhausted: feeds 0xFF (:609-632)
          else {
            ms_val = get32(f_ms, at);
            if (at + 32 > ms_bits) ms_val |= 0xFFFFFFFFu << (ms_bits - at);
          }
          const uint32_t mn = m[nn];
          at += mn;
          val = ms_val << 31;                                                           // :1127-1133
          v_n = ms_val & ((mn >= 32 ? 0u : (1u << mn)) - 1u);
          v_n |= ((inf >> (8 + nn)) & 1u) << mn;
          v_n |= 1u;
          val |= (v_n + 2u) << (p - 1);
        }
        // de-quantise transfer (ojph_codestream_gen.cpp:124-168)
        const uint32_t mag = val & 0x7FFFFFFFu;
        if (rev) { const int iv = (int)(mag >> shift); out[nn] = (uint32_t)((val >> 31) ? -iv : iv); }
        else { const float fv = __fmul_rn((float)mag, delta); out[nn] = __float_as_uint((val >> 31) ? -fv : fv); }
        if ((nn & 1) && act) vnew[2 * qx + (nn >> 1) + 1] = (uint8_t)(v_n ? 31 - __clz((int)v_n) : 0);
      }
      if (act) {
        const uint32_t x = 2 * qx, y = 2 * qy;
        const bool c1 = x + 1 < W, r1 = y + 1 < H;
        uint32_t* o = dst + (size_t)y * pitch + x;
        o[0] = out[0]; if (c1) o[1] = out[2];
        if (r1) { o[pitch] = out[1]; if (c1) o[pitch + 1] = out[3]; }
      }
    }
    wave_sync();
  }
  if (bad) { fail(1); return; }
}

}  // namespace

extern "C" int ojphgpu_ht_decode(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                                  const uint8_t* d_data, void* d_coef, uint8_t* d_block_status,
                                  uint32_t max_len1, uint32_t nominal_w, uint32_t nominal_h)
{
  if (n == 0) return OJPHGPU_OK;
  if (ojphgpu::ensure_tables() != 0) return OJPHGPU_E_HIP;
  if (!d_blocks || !d_data || !d_coef || !d_block_status) return OJPHGPU_E_INVALID;
  const uint32_t ms_words = ((max_len1 + 3) >> 2) + 4;
  const uint32_t table_words = (2048 + 320 + 256) / 2;
  if (nominal_w == 0 || nominal_h == 0 || nominal_w > 1024 || nominal_h > 1024) return OJPHGPU_E_INVALID;
  // per-quad records: QH rows of (QW + 1) entries for any block of at most nominal_w x nominal_h
  // samples and at most 4096 samples in total
  const uint32_t qwn = (nominal_w + 1) / 2, qhn = (nominal_h + 1) / 2;
  uint32_t quad_words = (qwn * qhn < 1024u ? qwn * qhn : 1024u) + qhn;
  quad_words = (quad_words + 3) & ~3u;
  const uint32_t wave_words = ms_words + (VLC_FLAT_BYTES / 4 + 2) + (MEL_FLAT_BYTES / 4 + 2) + quad_words + 2 * (EXP_BYTES / 4);
  const size_t lds = (size_t)(table_words + WAVES * wave_words) * 4;
  if (lds > 160 * 1024) return OJPHGPU_E_INVALID;
  static bool attr_set = false;
  if (!attr_set || lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ht_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) return OJPHGPU_E_HIP;
    attr_set = true;
  }
  dim3 grid((n + WAVES - 1) / WAVES);
  hipLaunchKernelGGL(ht_decode_kernel, grid, dim3(64 * WAVES), lds, (hipStream_t)stream, d_blocks, n, d_data,
                     (uint32_t*)d_coef, d_block_status, ms_words, quad_words);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

namespace ojphgpu {
int upload_dec_tables(const HtTables& t)
{
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_dec_vlc), t.dec_vlc, sizeof(t.dec_vlc)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_dec_uvlc0), t.dec_uvlc0, sizeof(t.dec_uvlc0)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_dec_uvlc1), t.dec_uvlc1, sizeof(t.dec_uvlc1)) != hipSuccess) return -1;
  return 0;
}
}
