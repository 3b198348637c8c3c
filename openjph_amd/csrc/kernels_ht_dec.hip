// openjph_amd/csrc/kernels_ht_dec.hip -- HT block decoder (cleanup pass) for gfx950 with the
// de-quantise transfer fused into its sample stores.
//
// Reference: ojph_decode_codeblock32 (src/core/coding/ojph_block_decoder32.cpp:742-1316):
//   MEL reader/decoder :93-269, backward VLC reader :308-439, forward MagSgn reader :581-723,
//   step 1 (MEL + VLC + U-VLC -> per-quad {rho, e_1, e_k, u}) :854-1089,
//   step 2 (MagSgn -> samples) :1091-1316;
// de-quantise transfer gen_rev/irv_tx_from_cb32 (src/core/codestream/ojph_codestream_gen.cpp:
// 124-168); zero blocks / failures: codeblock::decode + pull_line (ojph_codeblock.cpp:190-266).
//
// Two launches per frame:
//   step 1  (ht_dec_step1_kernel)  The MEL / VLC / U-VLC stage is a serial state machine: where a
//           codeword starts depends on every earlier codeword and on the neighbour context.  A
//           wavefront cannot split one block's chain, so the first version of this file ran the
//           chain as wave-uniform code, one wavefront per block -- 63 of 64 lanes repeated the same
//           work and the kernel took 4.8 ms per 8K frame (profiles/r01_a_*).  Now ONE LANE owns one
//           code-block's chain and a wavefront advances 64 independent chains in lock step: the
//           bit windows (un-stuffed on the fly, 4 bytes at a time) live in registers, the decode
//           tables in LDS, the significance of the quad row above in two 64-bit masks.  Output:
//           one 32-bit record per quad {t-word, u_q} in HBM scratch.
//   step 2  (ht_dec_step2_kernel)  ONE WAVEFRONT PER CODE-BLOCK.  All 64 lanes de-stuff the
//           MagSgn segment into a flat LDS bit buffer (un-stuffing only looks at the previous raw
//           byte, so bit offsets are a wavefront prefix sum -- the idea of the reference's AVX2
//           decoder, ojph_block_decoder_avx2.cpp:277-386).  Then, per quad row: lane = quad, kappa
//           from the exponents of the row above (LDS), bit offsets from a prefix sum of the m_n,
//           samples extracted from the flat buffer, de-quantised and stored to the sub-band plane.
// SigProp / MagRef passes (:1318-1609) are not implemented yet: blocks carrying them decode
// their cleanup pass only -- identical to the reference for streams of its own encoder, which
// never emits these passes (ojph_block_encoder.cpp:548).
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

__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// wave64 inclusive prefix sum with DPP adds (row shifts inside 16-lane rows, then row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int)
{
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
  return (uint32_t)x;
}

__device__ __forceinline__ uint32_t mel_exp(uint32_t k)      // {0,0,0,1,1,1,2,2,2,3,3,4,5}
{
  const uint64_t tbl = 0ull | (1ull << 9) | (1ull << 12) | (1ull << 15) | (2ull << 18) | (2ull << 21) | (2ull << 24) |
                       (3ull << 27) | (3ull << 30) | (4ull << 33) | (5ull << 36);
  return (uint32_t)(tbl >> (3 * k)) & 7u;
}

// validates what the reference validates before touching the block (block_decoder32.cpp:752-819);
// returns scup or 0
__device__ __forceinline__ uint32_t check_block(const ojphgpu_cb_desc& d, const uint8_t* cb)
{
  if (d.num_passes > 3 || d.missing_msbs >= 30 || d.len1 < 2) return 0;
  const uint32_t lcup = d.len1;
  const uint32_t scup = ((uint32_t)cb[lcup - 1] << 4) + (cb[lcup - 2] & 0xFu);
  if (scup < 2 || scup > lcup || scup > 4079) return 0;
  return scup;
}

// -------------------------------------------------------------------------------------------------
// step 1: one lane = one code-block
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p)
{
  uint32_t v; __builtin_memcpy(&v, p, 4); return v;
}

// Both readers keep the NEXT four raw bytes in a register, loaded one refill ahead, so the HBM/L2
// latency of the byte stream never sits on the serial decode chain.
struct VlcReader {          // backward reader with un-stuffing (block_decoder32.cpp:308-439)
  const uint8_t* p; int left; uint64_t tmp; uint32_t bits, unstuff, nxt;
  __device__ __forceinline__ void fetch() {           // raw bytes p, p-1, p-2, p-3 -> nxt (first byte on top)
    if (left >= 4) nxt = load_u32_unaligned(p - 3);
    else {
      nxt = 0;
      if (left > 0) nxt |= (uint32_t)p[0] << 24;
      if (left > 1) nxt |= (uint32_t)p[-1] << 16;
      if (left > 2) nxt |= (uint32_t)p[-2] << 8;
    }
  }
  __device__ __forceinline__ void init(const uint8_t* cb, uint32_t lcup, uint32_t scup) {
    const uint32_t d = cb[lcup - 2];
    tmp = d >> 4;
    bits = 4u - (((uint32_t)tmp & 7u) == 7u ? 1u : 0u);
    unstuff = (d | 0xFu) > 0x8Fu;
    p = cb + lcup - 3; left = (int)scup - 2;
    fetch(); refill(); refill();
  }
  __device__ __forceinline__ void refill() {
    if (bits > 32) return;
    const uint32_t v = nxt;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                     // bytes past the segment read as 0 (:313-331)
      const uint32_t b = (v >> (24 - 8 * k)) & 0xFFu;
      const uint32_t nb = 8u - ((unstuff && (b & 0x7Fu) == 0x7Fu) ? 1u : 0u);
      tmp |= (uint64_t)(b & ((1u << nb) - 1u)) << bits;
      bits += nb;
      unstuff = b > 0x8Fu;
    }
    p -= 4; left = left > 4 ? left - 4 : 0;
    fetch();
  }
  __device__ __forceinline__ uint32_t peek() const { return (uint32_t)tmp; }
  __device__ __forceinline__ void skip(uint32_t n) { tmp >>= n; bits -= n; }
};

struct MelReader {          // forward, MSB first, 0xFF -> 7 bits (block_decoder32.cpp:93-269)
  const uint8_t* p; int left; uint64_t tmp; int bits; uint32_t unstuff, nxt; uint32_t k, run, one;
  __device__ __forceinline__ void fetch() {           // raw bytes p .. p+3 -> nxt (first byte in the LSB)
    if (left >= 4) { nxt = load_u32_unaligned(p); if (left == 4) nxt |= 0x0F000000u; }   // last byte |= 0xF (:116)
    else {
      nxt = 0xFFFFFFFFu;                              // past the end the segment continues with 0xFF (:98)
      if (left > 0) nxt = (nxt & ~0xFFu) | (uint32_t)p[0] | (left == 1 ? 0xFu : 0u);
      if (left > 1) nxt = (nxt & ~0xFF00u) | (((uint32_t)p[1] | (left == 2 ? 0xFu : 0u)) << 8);
      if (left > 2) nxt = (nxt & ~0xFF0000u) | (((uint32_t)p[2] | (left == 3 ? 0xFu : 0u)) << 16);
    }
  }
  __device__ __forceinline__ void init(const uint8_t* cb, uint32_t lcup, uint32_t scup) {
    p = cb + lcup - scup; left = (int)scup - 1; tmp = 0; bits = 0; unstuff = 0; k = 0; run = 0; one = 0;
    fetch(); refill();
  }
  __device__ __forceinline__ void refill() {
    if (bits > 32) return;
    const uint32_t v = nxt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t b = (v >> (8 * i)) & 0xFFu;
      const int nb = 8 - (int)unstuff;
      tmp |= (uint64_t)(b & ((1u << nb) - 1u)) << (64 - bits - nb);
      bits += nb;
      unstuff = (b == 0xFF);
    }
    p += 4; left = left > 4 ? left - 4 : 0;
    fetch();
  }
  __device__ __forceinline__ uint32_t sym() {     // T.814 decodeMELSym; same runs as :170-269
    if (run == 0 && one == 0) {
      const uint32_t e = mel_exp(k);
      if (tmp >> 63) { run = 1u << e; k = k < 12 ? k + 1 : 12; tmp <<= 1; bits -= 1; }
      else {
        run = e ? (uint32_t)((tmp << 1) >> (64 - e)) : 0u;
        k = k > 0 ? k - 1 : 0; one = 1; tmp <<= (e + 1); bits -= (int)(e + 1);
      }
    }
    if (run > 0) { run--; return 0; }
    one = 0; return 1;
  }
};

__global__ __launch_bounds__(64) void ht_dec_step1_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    uint32_t* __restrict__ quads, uint8_t* __restrict__ block_status)
{
  __shared__ uint16_t s_vlc[2048];
  __shared__ uint16_t s_uvlc0[320];
  __shared__ uint16_t s_uvlc1[256];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_vlc[i] = (&ojphgpu::g_dec_vlc[0][0])[i];
  for (int i = threadIdx.x; i < 320; i += blockDim.x) s_uvlc0[i] = ojphgpu::g_dec_uvlc0[i];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s_uvlc1[i] = ojphgpu::g_dec_uvlc1[i];
  __syncthreads();

  const uint32_t bi = blockIdx.x * blockDim.x + threadIdx.x;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (d.w == 0 || d.h == 0) { block_status[bi] = 0; return; }
  if (d.len1 == 0 || d.num_passes == 0) { block_status[bi] = 0; return; }     // not coded: zero block
  const uint8_t* cb = data + d.data_off;
  const uint32_t scup = check_block(d, cb);
  if (scup == 0) { block_status[bi] = 1; return; }
  block_status[bi] = 0;
  const uint32_t lcup = d.len1;
  const uint32_t QW = ((uint32_t)d.w + 1) >> 1, QH = ((uint32_t)d.h + 1) >> 1;
  uint32_t* rec = quads + d.scratch_cap;          // QH rows of QW records

  VlcReader vlc; vlc.init(cb, lcup, scup);
  MelReader mel; mel.init(cb, lcup, scup);

  const bool small = QW <= 64;                    // significance of the row above fits two 64-bit masks
  uint64_t a_prev = 0, b_prev = 0, a_cur = 0, b_cur = 0;   // bit x: rho bit 1 (bottom-left) / bit 3 (bottom-right)
  for (uint32_t qy = 0; qy < QH; ++qy) {
    const uint16_t* tbl = s_vlc + (qy ? 1024 : 0);
    uint32_t* row = rec + qy * QW;
    const uint32_t* above = row - QW;
    uint32_t tleft = 0;
    a_cur = 0; b_cur = 0;
    for (uint32_t qx = 0; qx < QW; qx += 2) {
      uint32_t t[2] = { 0, 0 };
      vlc.refill();                       // > 32 bits: a pair consumes at most 2*7 + 6 + 10 of them
      mel.refill();                       // > 32 bits: a pair consumes at most 3 symbols of <= 6 bits
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t x = qx + j;
        if (x < QW) {
          uint32_t c_q;
          if (qy == 0) c_q = ((tleft & 0x10u) << 3) | ((tleft & 0xE0u) << 2);              // :903,:934
          else {
            c_q = ((tleft & 0x40u) << 2) | ((tleft & 0x80u) << 1);                         // :1022,:1059
            uint32_t nw, nn_l, nn_r, ne;     // sigma of columns 2x-1, 2x, 2x+1, 2x+2 of the sample row above
            if (small) {
              nw = x ? (uint32_t)(b_prev >> (x - 1)) & 1u : 0u;
              nn_l = (uint32_t)(a_prev >> x) & 1u; nn_r = (uint32_t)(b_prev >> x) & 1u;
              ne = x + 1 < 64 ? (uint32_t)(a_prev >> (x + 1)) & 1u : 0u;
            } else {
              nw = x ? (above[x - 1] >> 7) & 1u : 0u;
              const uint32_t up = above[x];
              nn_l = (up >> 5) & 1u; nn_r = (up >> 7) & 1u;
              ne = x + 1 < QW ? (above[x + 1] >> 5) & 1u : 0u;
            }
            c_q |= (nw | nn_l) << 7;                                                        // :990,:1024,:1026
            c_q |= (nn_r | ne) << 9;                                                        // :991,:1027
          }
          uint32_t tv = tbl[c_q + (vlc.peek() & 0x7Fu)];
          if (c_q == 0) { if (mel.sym() == 0) tv = 0; }                                     // :882-894
          vlc.skip(tv & 7u);
          t[j] = tv; tleft = tv;
          if (small) { a_cur |= (uint64_t)((tv >> 5) & 1u) << x; b_cur |= (uint64_t)((tv >> 7) & 1u) << x; }
        }
      }
      uint32_t mode = ((t[0] & 0x8u) << 3) | ((t[1] & 0x8u) << 4);
      uint32_t entry;
      if (qy == 0) {
        if (mode == 0xC0u) { if (mel.sym()) mode += 0x40u; }                                // :943-952
        entry = s_uvlc0[mode + (vlc.peek() & 0x3Fu)];
      } else entry = s_uvlc1[mode + (vlc.peek() & 0x3Fu)];
      vlc.skip(entry & 7u); entry >>= 3;
      uint32_t len = entry & 0xFu;
      const uint32_t tmp = vlc.peek() & ((1u << len) - 1u);
      vlc.skip(len); entry >>= 4;
      len = entry & 7u; entry >>= 3;
      const uint32_t kap = qy == 0 ? 1u : 0u;                                               // :971-974 / :1082-1085
      const uint32_t u0 = kap + (entry & 7u) + (tmp & ~(0xFFu << len));
      const uint32_t u1 = kap + (entry >> 3) + (tmp >> len);
      row[qx] = t[0] | (u0 << 16);
      if (qx + 1 < QW) row[qx + 1] = t[1] | (u1 << 16);
    }
    a_prev = a_cur; b_prev = b_cur;
  }
}

// -------------------------------------------------------------------------------------------------
// step 2: one wavefront = one code-block
// -------------------------------------------------------------------------------------------------
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

// De-stuffs n bytes (forward order, "after 0xFF only 7 bits") into the flat LSB-first bit buffer.
// Returns the number of bits written (wave-uniform).  (block_decoder32.cpp:609-653)
__device__ uint32_t destuff_forward(const uint8_t* __restrict__ src, uint32_t n, uint32_t* flat, int lane)
{
  uint32_t total = 0;
  for (uint32_t base = 0; base < n; base += 256) {
    const uint32_t i0 = base + 4u * (uint32_t)lane;
    uint32_t prev = (i0 > 0 && i0 - 1 < n) ? src[i0 - 1] : 0u;
    uint32_t val = 0, nb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = i0 + k;
      if (i < n) {
        const uint32_t b = src[i];
        const uint32_t bits = prev == 0xFF ? 7u : 8u;
        val |= (b & ((1u << bits) - 1u)) << nb; nb += bits;
        prev = b;
      }
    }
    const uint32_t incl = wave_incl_scan(nb, lane);
    or_bits(flat, total + incl - nb, val, nb);
    total += rdlane(incl, 63);
  }
  return total;
}

__global__ __launch_bounds__(64 * WAVES) void ht_dec_step2_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    const uint32_t* __restrict__ quads, uint32_t* __restrict__ coef, uint8_t* __restrict__ block_status,
    uint32_t ms_words, uint32_t exp_words)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t bi = blockIdx.x * WAVES + wave;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  const uint32_t wave_words = ms_words + 2 * exp_words;
  uint32_t* f_ms = smem + (uint32_t)wave * wave_words;
  uint8_t* vexp_a = reinterpret_cast<uint8_t*>(f_ms + ms_words);
  uint8_t* vexp_b = vexp_a + exp_words * 4;

  const uint32_t W = d.w, H = d.h, pitch = d.pitch;
  if (W == 0 || H == 0) return;
  uint32_t* dst = coef + d.coef_off;
  const bool rev = d.reversible != 0;
  const uint32_t K = d.K_max;

  auto zero_block = [&]() {                                  // mem_clear path, ojph_codeblock.cpp:247
    for (uint32_t y = 0; y < H; ++y)
      for (uint32_t x = lane; x < W; x += 64) dst[(size_t)y * pitch + x] = 0u;
  };
  if (d.len1 == 0 || d.num_passes == 0 || block_status[bi] != 0) { zero_block(); return; }
  const uint32_t missing_msbs = d.missing_msbs;
  const uint32_t p = 30 - missing_msbs;
  const uint8_t* cb = data + d.data_off;
  const uint32_t lcup = d.len1;
  const uint32_t scup = ((uint32_t)cb[lcup - 1] << 4) + (cb[lcup - 2] & 0xFu);
  const uint32_t ms_len = lcup - scup;
  const uint32_t QW = (W + 1) >> 1, QH = (H + 1) >> 1;
  if (((ms_len + 3) >> 2) + 2 > ms_words || 2 * QW + 8 > exp_words * 4) {        // LDS sized too small
    zero_block(); if (lane == 0) block_status[bi] = 2; return;
  }
  const uint32_t used_words = ((ms_len + 3) >> 2) + 2;
  for (uint32_t i = lane; i < used_words; i += 64) f_ms[i] = 0;
  for (uint32_t i = lane; i < 2 * exp_words; i += 64) (f_ms + ms_words)[i] = 0;
  wave_sync();
  const uint32_t ms_bits = destuff_forward(cb, ms_len, f_ms, lane);
  wave_sync();

  const uint32_t* rec = quads + d.scratch_cap;
  const uint32_t mmsbp2 = missing_msbs + 2;
  const uint32_t shift = 31 - K;
  const float delta = d.delta;
  uint32_t mpos = 0;                              // wave-uniform bit position in the flat MagSgn stream
  bool bad = false;
  uint32_t ent_next = (uint32_t)lane < QW ? rec[lane] : 0u;      // records are fetched one step ahead
  for (uint32_t qy = 0; qy < QH && !bad; ++qy) {
    const uint8_t* vexp = (qy & 1) ? vexp_b : vexp_a;     // exponents of the sample row above
    uint8_t* vnew = (qy & 1) ? vexp_a : vexp_b;           // ... and of this quad row's bottom samples
    for (uint32_t qb = 0; qb < QW; qb += 64) {
      const uint32_t qx = qb + (uint32_t)lane;
      const bool act = qx < QW;
      const uint32_t ent = ent_next;
      {
        uint32_t nqb = qb + 64, nqy = qy;
        if (nqb >= QW) { nqb = 0; nqy = qy + 1; }
        const uint32_t nqx = nqb + (uint32_t)lane;
        ent_next = (nqy < QH && nqx < QW) ? rec[nqy * QW + nqx] : 0u;
      }
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
  if (bad) { zero_block(); if (lane == 0) block_status[bi] = 1; }
}

}  // namespace

extern "C" int ojphgpu_ht_decode(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                                  const uint8_t* d_data, void* d_coef, uint32_t* d_quad_scratch,
                                  uint8_t* d_block_status, uint32_t max_len1, uint32_t nominal_w,
                                  uint32_t nominal_h)
{
  if (n == 0) return OJPHGPU_OK;
  if (ojphgpu::ensure_tables() != 0) return OJPHGPU_E_HIP;
  if (!d_blocks || !d_data || !d_coef || !d_block_status || !d_quad_scratch) return OJPHGPU_E_INVALID;
  if (nominal_w == 0 || nominal_h == 0 || nominal_w > 1024 || nominal_h > 1024) return OJPHGPU_E_INVALID;
  const uint32_t ms_words = ((max_len1 + 3) >> 2) + 4;
  const uint32_t exp_words = (nominal_w + 8 + 3) / 4 + 1;
  const size_t lds = (size_t)WAVES * (ms_words + 2 * exp_words) * 4;
  if (lds > 160 * 1024) return OJPHGPU_E_INVALID;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(ht_dec_step2_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return OJPHGPU_E_HIP;
  hipLaunchKernelGGL(ht_dec_step1_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_blocks, n, d_data,
                     d_quad_scratch, d_block_status);
  hipLaunchKernelGGL(ht_dec_step2_kernel, dim3((n + WAVES - 1) / WAVES), dim3(64 * WAVES), lds, (hipStream_t)stream,
                     d_blocks, n, d_data, (const uint32_t*)d_quad_scratch, (uint32_t*)d_coef, d_block_status,
                     ms_words, exp_words);
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
