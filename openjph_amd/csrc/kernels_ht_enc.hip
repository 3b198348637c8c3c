// openjph_amd/csrc/kernels_ht_enc.hip -- HT cleanup-pass block encoder for gfx950, with the
// quantise transfer fused into its sample loads.  ONE WAVEFRONT PER CODE-BLOCK.
//
// Two kernels share the work: ht_encode_kernel (below, second half of the file) handles blocks up
// to 64 columns wide -- every block of the nominal 64x64 / 32x32 partitions -- with ONE LANE PER
// QUAD PAIR, four quad rows per step, and neighbour state exchanged between lanes;
// ht_encode_wide_kernel (first half, the original formulation: pairs in raster order, neighbours
// re-read from the block) handles the rare wider shapes (128x32 ... 1024x4).
//
// Reference: ojph_encode_codeblock32 (src/core/coding/ojph_block_encoder.cpp:542-1017) and its
// writers (mel :273-347, vlc :352-407, ms :446-534, terminate_mel_vlc :412-441);
// quantise transfer gen_rev/irv_tx_to_cb32 (src/core/codestream/ojph_codestream_gen.cpp:59-121);
// "is there anything to code" test codeblock::encode (ojph_codeblock.cpp:142-175).
//
// The reference walks the block quad pair by quad pair with three serial bit writers.  Here (narrow kernel; the wide
// one keeps the first formulation, neighbours re-read from the block)
//   * a lane owns one quad PAIR (8 samples) per step, the wave covers 64 consecutive pairs in raster order; everything a
//     quad contributes (rho, exponents, context, kappa, u, VLC codeword, MagSgn bits) depends only on sample values, so it
//     is computed in parallel, with instruction COST in mind (tools/micro/valu_issue.hip: add / and / or / xor / shift-right
//     and fp32 add-mul issue at 2.4 cycles per wave64 instruction per SIMD, everything else -- shifts left, bit-field,
//     compare, select, cross-lane, 3-operand forms -- at 4.2): the quantise transfer is one multiply / and by a per-column
//     constant, a quad's exponents / rho / eps / MagSgn lengths are the bytes of one word, and what the row below needs
//     from this row is worked out here and handed down as ONE packed word per lane;
//   * MagSgn and VLC bits are OR-ed by all lanes into flat, un-stuffed LDS bit buffers at offsets given by ONE wavefront
//     prefix sum (sample pairs of at most 32 bits when no sample of the step has more than 16);
//   * byte stuffing (0xFF -> 7 bits forward; >0x8F,0x7F backward) is resolved by a speculative pass: every lane proposes
//     its bytes assuming no stuffing event inside the window, a ballot finds the first event, lanes up to it commit, and
//     the window restarts; lazily, only for whole windows;
//   * MEL is an adaptive run-length coder and stays serial, but it only sees the "1" events: event masks are ballots, the
//     zero run in front of a "1" is a population count; the serial part appends raw code bits, the bytes are made once
//     per block by the whole wavefront (mel_stuff), and steps without a significant sample skip everything else;
//   * the block's bytes are staged in LDS (MagSgn growing up, VLC growing down); a stage that fills up flushes its whole
//     dwords to the block's scratch slot in HBM and goes on; when the three lengths are known the block takes its place in
//     the compacted output with one atomicAdd -- on the cursor of one of 16 regions, not on one cursor for all (claim_output);
//   * launches whose blocks are all at most 32 columns wide (the IMF profile's 32x32) use a layout of 8 quad pairs x 8
//     quad rows per step instead of 16 x 4 (template parameter LOGP).
// The produced bytes are identical to the reference's (oracle/ht_oracle.c is the CPU model and is pinned against the
// reference; DESIGN.md section 4.2 has the instruction counts and the ablations).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>
#include "../../include/ojphgpu.h"
#include "ht_tables.h"

namespace ojphgpu {
__device__ __attribute__((aligned(16))) uint16_t g_enc_vlc[2][2048];   // filled by ojphgpu_upload_tables()
}

namespace {

constexpr int MS_WORDS = 512;     // 64 lanes * 8 samples * 31 bits + carry  < 2048 bytes
constexpr int VLC_WORDS = 64;     // 64 lanes * 30 bits + carry < 256 bytes
constexpr int MEL_CAP = 192;      // ojph_block_encoder.cpp:554
constexpr int VLC_CAP = 3072 - MEL_CAP;   // :556
constexpr int WAVES = 4;
constexpr uint32_t NARROW_MAX_W = 64;   // blocks up to this width take the lane-per-column kernel

constexpr int MS_WORDS64 = 1040;  // 64-bit samples: 64 lanes * 8 samples * 63 bits + carry
constexpr int VLC_WORDS64 = 96;   //                 64 lanes * 38 bits + carry
template <int MSW, int VLW>
struct WaveLdsT {
  uint32_t ms[MSW];
  uint32_t vlc[VLW];
  uint32_t ev[8];                 // compacted MEL event bits of one step (<= 192)
  uint8_t  mel[MEL_CAP];
};

__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

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

__device__ __forceinline__ uint32_t get_bits(const uint32_t* buf, uint32_t pos, uint32_t n)
{
  const uint32_t w = pos >> 5, sh = pos & 31;
  const uint32_t lo = buf[w], hi = buf[w + 1];
  const uint32_t v = __funnelshift_r(lo, hi, sh);
  return v & ((1u << n) - 1u);
}

__device__ __forceinline__ void or_bits(uint32_t* buf, uint32_t pos, uint32_t v, uint32_t n)
{
  if (n == 0) return;
  const uint32_t w = pos >> 5, sh = pos & 31;
  atomicOr(&buf[w], v << sh);
  if (sh + n > 32) atomicOr(&buf[w + 1], v >> (32 - sh));
}

__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t rdfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// quantise transfer of one raw coefficient: sign | magnitude, MSB aligned.  A coefficient with more than K_max magnitude
// bits (Part-2 kernels whose gain outruns the guard bits) leaves the reference's transfer the way its 32-bit arithmetic
// has it: reversible, |v| << shift drops what does not fit and bit K_max of |v| lands on the sign position
// (ojph_codestream_gen.cpp:70-76); irreversible, the float -> int conversion of a product beyond 2^31 gives INT_MIN (what
// cvttss2si and its vector forms return for every out-of-range input, NaN included), i.e. the word 0x80000000: a zero.
// Either way that bit counts in max_val, and codeblock::encode (ojph_codeblock.cpp:142-175) codes a block whose max_val
// is not zero even when no sample of it is significant: `over` collects it.
__device__ __forceinline__ uint32_t to_sign_mag(uint32_t raw, bool reversible, uint32_t shift, float delta_inv, uint32_t& over)
{
  if (reversible) {
    const int v = (int)raw;
    const uint32_t m = (v >= 0 ? (uint32_t)v : 0u - (uint32_t)v) << shift;
    over |= m >> 31;
    return (v >= 0 ? 0u : 0x80000000u) | m;
  }
  const float f = __fmul_rn(__uint_as_float(raw), delta_inv);            // :113-118, C truncation
  const bool out = !(fabsf(f) < 2147483648.0f);
  const int t = out ? (int)0x80000000u : (int)f;
  const uint32_t m = t >= 0 ? (uint32_t)t : 0u - (uint32_t)t;
  over |= m >> 31;
  return (t >= 0 ? 0u : 0x80000000u) | m;
}

__device__ __forceinline__ uint32_t expo(uint32_t val) { return val ? 32u - (uint32_t)__clz((int)(val - 1)) : 0u; }
__device__ __forceinline__ uint32_t expo(uint64_t val) { return val ? 64u - (uint32_t)__clzll((long long)(val - 1)) : 0u; }

// MEL exponents {0,0,0,1,1,1,2,2,2,3,3,4,5} packed 3 bits each (ojph_block_encoder.cpp:324)
__device__ __forceinline__ uint32_t mel_exp(uint32_t k)
{
  const uint64_t tbl = 0ull | (1ull << 9) | (1ull << 12) | (1ull << 15) | (2ull << 18) | (2ull << 21) | (2ull << 24) |
                       (3ull << 27) | (3ull << 30) | (4ull << 33) | (5ull << 36);
  return (uint32_t)(tbl >> (3 * k)) & 7u;
}

struct MelState {          // all wave-uniform
  uint32_t k, run, acc, nb, pos, lastff, err;
};

__device__ __forceinline__ void mel_put(MelState& m, uint8_t* buf, uint32_t code, uint32_t n, int lane)
{
  m.acc = (m.acc << n) | code; m.nb += n;
  for (;;) {
    const uint32_t need = m.lastff ? 7u : 8u;
    if (m.nb < need) break;
    const uint32_t byte = (m.acc >> (m.nb - need)) & ((1u << need) - 1u);
    m.nb -= need; m.acc &= (1u << m.nb) - 1u;
    if (m.pos >= (uint32_t)MEL_CAP) { m.err = 1; break; }
    if (lane == 0) buf[m.pos] = (uint8_t)byte;
    m.pos++; m.lastff = (byte == 0xFF);
  }
}

__device__ __forceinline__ void mel_zero_run(MelState& m, uint8_t* buf, uint32_t n, int lane)
{
  while (n > 0) {
    const uint32_t thr = 1u << mel_exp(m.k);
    const uint32_t need = thr - m.run;
    if (n >= need) { mel_put(m, buf, 1, 1, lane); n -= need; m.run = 0; m.k = m.k < 12 ? m.k + 1 : 12; }
    else { m.run += n; n = 0; }
  }
}

__device__ __forceinline__ void mel_one(MelState& m, uint8_t* buf, int lane)
{
  const uint32_t e = mel_exp(m.k);
  mel_put(m, buf, m.run, e + 1, lane);          // a 0 followed by e bits of the run count
  m.run = 0; m.k = m.k > 0 ? m.k - 1 : 0;
}

// A slot of `bytes` (multiple of 4) in the compacted output for block bi; 0xFFFFFFFF when it does not fit.
// One cursor for the whole output is ONE cache line that every block's atomic goes to: same-address device-scope
// atomics complete at ~55 M/s on this part (they are resolved at the memory side, the XCDs' L2s are not coherent
// with each other), 18 000 of them are 0.33 ms -- the top resolution's launch could not get below 0.39 ms, a quarter of
// it waiting in that queue (measured with the allocation taken out).  With `nreg` regions (a power of two) block bi
// allocates in region bi % nreg: regions[2r] = first byte, regions[2r + 1] = bytes, its cursor in a cache line of its
// own at cursor[32 r].  nreg = 0: the single cursor of the C ABI entry point.
__device__ __forceinline__ uint32_t claim_output(uint32_t* cursor, const uint32_t* regions, uint32_t nreg, uint32_t bi,
                                                 uint32_t bytes, uint32_t out_cap, int lane)
{
  uint32_t base = 0, cap = out_cap;
  uint32_t* cur = cursor;
  if (nreg) {
    const uint32_t r = bi & (nreg - 1u);
    base = regions[2u * r]; cap = regions[2u * r + 1u];
    cur = cursor + 32u * r;
  }
  uint32_t off = 0;
  if (lane == 0) off = atomicAdd(cur, bytes);
  off = rdfirst(off);
  return (off > cap || bytes > cap - off) ? 0xFFFFFFFFu : base + off;
}

// S64: the blocks of components on the 64-bit sample path (ojph_encode_codeblock64, ojph_block_encoder.cpp:1026-1520; the
// transfer gen_rev_tx_to_cb64, ojph_codestream_gen.cpp:81-100): int64 samples, exponents up to 63, MagSgn values of up to
// 63 bits and the U-VLC extension for u > 32 (:245-253, :1269-1293, :1487-1492) -- the same symbols otherwise.
template <bool S64>
__global__ __launch_bounds__(64 * WAVES) void ht_encode_wide_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint32_t* __restrict__ coef,
    uint8_t* __restrict__ scratch, uint8_t* __restrict__ out, uint32_t out_cap,
    ojphgpu_cb_result* __restrict__ results, uint32_t* __restrict__ cursor, uint32_t* __restrict__ status,
    const uint32_t* __restrict__ regions, uint32_t nreg)
{
  using V = typename std::conditional<S64, uint64_t, uint32_t>::type;     // a sign-magnitude sample
  constexpr uint32_t VBITS = S64 ? 64u : 32u;
  constexpr int MSW = S64 ? MS_WORDS64 : MS_WORDS, VLW = S64 ? VLC_WORDS64 : VLC_WORDS;
  __shared__ __attribute__((aligned(16))) uint16_t s_vlc[2][2048];
  __shared__ WaveLdsT<MSW, VLW> s_wave[WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and the compiler knows it
  const uint32_t bi = blockIdx.x * WAVES + wave;
  // narrow blocks of 32-bit samples belong to ht_encode_kernel; blocks of 64-bit samples, of any width, to the S64 instantiation
  const bool mine = bi < n && (S64 ? (blocks[bi].reversible & 4u) != 0 : (blocks[bi].w > NARROW_MAX_W && (blocks[bi].reversible & 4u) == 0));
  if (!__syncthreads_or(mine ? 1 : 0)) return;
  // (16 bytes per lane and turn.  A workgroup lives for 12-40 us and meets at the barrier below before anything else: with
  // 2-byte turns, sixteen of them, the 8K frame's encode was 0.458 ms instead of 0.440, 32 x 32 blocks 0.603 instead of 0.562)
  for (int i = threadIdx.x; i < 2 * 2048 / 8; i += blockDim.x)
    reinterpret_cast<uint4*>(&s_vlc[0][0])[i] = reinterpret_cast<const uint4*>(&ojphgpu::g_enc_vlc[0][0])[i];
  __syncthreads();
  if (!mine) return;
  const ojphgpu_cb_desc d = blocks[bi];
  WaveLdsT<MSW, VLW>& L = s_wave[wave];
  const uint32_t W = d.w, H = d.h;
  if (W == 0 || H == 0) { if (lane == 0) { results[bi].offset = 0; results[bi].length = 0; } return; }
  const uint32_t K = d.K_max, p = VBITS - 1u - K;   // missing_msbs = K_max - 1, p = 30 (62) - missing_msbs
  const bool rev = (d.reversible & 1u) != 0;
  const float delta_inv = rev ? 0.0f : __fdiv_rn(1.0f, d.delta);         // ojph_codeblock.cpp:98
  const uint32_t* src = coef + d.coef_off;
  const uint32_t pitch = d.pitch;
  uint8_t* ms_out = scratch + d.data_off;
  const uint32_t ms_cap = d.scratch_cap > (uint32_t)VLC_CAP ? d.scratch_cap - VLC_CAP : 0;
  uint8_t* vlc_last = scratch + d.data_off + d.scratch_cap - 1;          // VLC grows downwards from here

  const uint32_t QW = (W + 1) >> 1, QH = (H + 1) >> 1, PW = (QW + 1) >> 1, NP = PW * QH;

  for (int i = lane; i < MSW; i += 64) L.ms[i] = 0;
  for (int i = lane; i < VLW; i += 64) L.vlc[i] = 0;
  if (lane < 8) L.ev[lane] = 0;
  wave_sync();
  if (lane == 0) L.vlc[0] = 0xF;                                          // vlc_init: 4 bits already used (:365-375)
  wave_sync();

  // wave-uniform stream state
  uint32_t ms_carry = 0, ms_k = 0, ms_ff = 0;           // carried bits, bytes written, last byte was 0xFF
  uint32_t v_carry = 4, v_pos = 1, v_prev = 0xFF;       // carried bits, bytes "written" (incl. the 0xFF head), last byte
  MelState mel = { 0, 0, 0, 0, 0, 0, 0 };
  uint32_t err = 0, any_sig = 0;
  uint32_t carry_rho = 0;                               // rho of the last quad of the previous step

  uint32_t over = 0;                                    // some magnitude reached the sign position (see to_sign_mag)
  auto sample = [&](int x, int y) -> V {                // quantised sign-magnitude, 0 outside the block
    if (x < 0 || y < 0 || x >= (int)W || y >= (int)H) return (V)0;
    if constexpr (S64) {                                // gen_rev_tx_to_cb64
      const long long v = reinterpret_cast<const long long*>(src)[(size_t)y * pitch + x];
      const uint64_t m = (v >= 0 ? (uint64_t)v : 0ull - (uint64_t)v) << p;
      over |= (uint32_t)(m >> 63);
      return (v >= 0 ? 0ull : 0x8000000000000000ull) | m;
    } else
      return to_sign_mag(src[(size_t)y * pitch + x], rev, p, delta_inv, over);
  };

  for (uint32_t base = 0; base < NP; base += 64) {
    const uint32_t P = base + lane;
    const bool active = P < NP;
    const uint32_t qy = active ? P / PW : 0, px = active ? P - qy * PW : 0;
    const int x0 = (int)(4 * px), y0 = (int)(2 * qy);
    const bool has_q1 = active && (x0 + 2 < (int)W);
    const bool first_row = qy == 0;

    // ---- samples of the pair: t[q*4+n], n = 0:(x,y) 1:(x,y+1) 2:(x+1,y) 3:(x+1,y+1) ----
    V val[8]; uint32_t sgn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = i >> 2, nn = i & 3;
      V t = active ? sample(x0 + 2 * q + (nn >> 1), y0 + (nn & 1)) : (V)0;
      val[i] = ((V)(t + t) >> p) & ~(V)1;               // 2*mu_p        (:592-595)
      sgn[i] = (uint32_t)(t >> (VBITS - 1u));
    }
    // ---- bottom sample row of the quad row above: columns x0-1 .. x0+4 ----
    uint32_t Eab[6], Sab[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      V t = (active && !first_row) ? sample(x0 - 1 + i, y0 - 1) : (V)0;
      V v = ((V)(t + t) >> p) & ~(V)1;
      Eab[i] = expo(v); Sab[i] = v != 0;
    }

    // ---- per quad symbols ----
    uint32_t rho[2], cq[2], uq[2], Uq[2], tup[2];
    V msv[8]; uint32_t msl[8];
    uint32_t rho_q[2] = { 0, 0 }, emax[2] = { 0, 0 }, e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      e[i] = expo(val[i]);
      if (val[i]) rho_q[i >> 2] |= 1u << (i & 3);
      emax[i >> 2] = max(emax[i >> 2], e[i]);
    }
    if (!has_q1) { rho_q[1] = 0; emax[1] = 0; }
    // rho of the quad to the left of q0: q1 of the previous pair (previous lane / previous step)
    uint32_t rho_prev = __shfl_up(rho_q[1], 1);
    if (lane == 0) rho_prev = carry_rho;
    if (px == 0) rho_prev = 0;
    carry_rho = rdlane(rho_q[1], 63);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t rl = q == 0 ? rho_prev : rho_q[0];
      uint32_t kappa = 1, c;
      if (first_row) c = (rl >> 1) | (rl & 1);                                      // :731,:788
      else {
        const uint32_t* E = Eab + 2 * q; const uint32_t* S = Sab + 2 * q;
        uint32_t me = max(max(E[0], E[1]), max(E[2], E[3]));
        int max_e = (int)me - 1;
        if (rho_q[q] & (rho_q[q] - 1)) kappa = (uint32_t)max(1, max_e);             // :862,:950
        c = (S[0] | S[1]) | ((S[2] | S[3]) << 2) | ((rl & 4) >> 1) | ((rl & 8) >> 2); // :802,:878,:951,:967,:991
      }
      const uint32_t U = max(emax[q], kappa), u = U - kappa;
      uint32_t eps = 0;
      if (u > 0) {
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) eps |= (uint32_t)(e[q * 4 + nn] == emax[q]) << nn;
      }
      const uint32_t tuple = s_vlc[first_row ? 0 : 1][(c << 8) + (rho_q[q] << 4) + eps];
      rho[q] = rho_q[q]; cq[q] = c; uq[q] = u; Uq[q] = U; tup[q] = tuple;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        const int i = q * 4 + nn;
        const uint32_t m = ((rho_q[q] >> nn) & 1u) ? U - ((tuple >> nn) & 1u) : 0u;   // :667-674
        const V s = val[i] ? val[i] - 2u + sgn[i] : (V)0;                           // v_n = 2(mu-1)+sign (:601)
        msl[i] = m;
        msv[i] = m ? (s & ((m >= VBITS ? (V)0 : ((V)1 << m)) - (V)1)) : (V)0;
      }
    }
    const bool q0on = active, q1on = has_q1;
    if (!q1on) { uq[1] = 0; for (int nn = 4; nn < 8; ++nn) { msl[nn] = 0; msv[nn] = 0; } }
    if (!q0on) { uq[0] = 0; for (int nn = 0; nn < 4; ++nn) { msl[nn] = 0; msv[nn] = 0; } }
    any_sig |= (__ballot(((rho[0] | rho[1]) != 0 && active) || over != 0u) != 0ull) ? 1u : 0u;

    // ---- VLC bits of the pair: cwd(q0) cwd(q1) then the interleaved U-VLC ----
    uint64_t vb = 0; uint32_t vl = 0;                   // (S64: up to 2 x 7 + 2 x (3 + 5 + 4) = 38 bits)
    auto vadd = [&](uint32_t c, uint32_t len) { vb |= (uint64_t)c << vl; vl += len; };
    uint32_t x0e = 0, xl0 = 0, x1e = 0, xl1 = 0;        // the extensions of the pair's two codewords (S64 only)
    auto uvlc = [](uint32_t u, uint32_t& pre, uint32_t& pl, uint32_t& suf, uint32_t& sl, uint32_t& ext, uint32_t& el) {   // :196-255
      ext = 0; el = 0;
      if (u == 0) { pre = 0; pl = 0; suf = 0; sl = 0; }
      else if (u == 1) { pre = 1; pl = 1; suf = 0; sl = 0; }
      else if (u == 2) { pre = 2; pl = 2; suf = 0; sl = 0; }
      else if (u <= 4) { pre = 4; pl = 3; suf = u - 3; sl = 1; }
      else if (u <= 32 || !S64) { pre = 0; pl = 3; suf = u - 5; sl = 5; }
      else { pre = 0; pl = 3; suf = 28u + ((u - 33u) & 3u); sl = 5; ext = (u - 33u) >> 2; el = 4; }   // :245-253
    };
    bool ev2_valid = false; uint32_t ev2_bit = 0;
    if (q0on) {
      vadd(tup[0] >> 8, (tup[0] >> 4) & 7);
      if (q1on) vadd(tup[1] >> 8, (tup[1] >> 4) & 7);
      const uint32_t u0 = uq[0], u1 = uq[1];
      uint32_t p0, l0, s0, sl0, p1, l1, s1, sl1;
      if (first_row && u0 > 0 && u1 > 0) { ev2_valid = true; ev2_bit = min(u0, u1) > 2; }    // :763-764
      if (first_row && u0 > 2 && u1 > 2) {                                                    // :766-772
        uvlc(u0 - 2, p0, l0, s0, sl0, x0e, xl0); uvlc(u1 - 2, p1, l1, s1, sl1, x1e, xl1);
        vadd(p0, l0); vadd(p1, l1); vadd(s0, sl0); vadd(s1, sl1); vadd(x0e, xl0); vadd(x1e, xl1);
      } else if (first_row && u0 > 2 && u1 > 0) {                                             // :773-778
        uvlc(u0, p0, l0, s0, sl0, x0e, xl0);
        vadd(p0, l0); vadd(u1 - 1, 1); vadd(s0, sl0); vadd(x0e, xl0);
      } else {                                                                                // :779-785, :985-988
        uvlc(u0, p0, l0, s0, sl0, x0e, xl0); uvlc(u1, p1, l1, s1, sl1, x1e, xl1);
        vadd(p0, l0); vadd(p1, l1); vadd(s0, sl0); vadd(s1, sl1); vadd(x0e, xl0); vadd(x1e, xl1);
      }
    }

    // ---- MEL events of the pair, compacted in pair order ----
    const bool ev0_valid = q0on && cq[0] == 0, ev1_valid = q1on && cq[1] == 0;
    const uint32_t ev0_bit = rho[0] != 0, ev1_bit = rho[1] != 0;
    {
      const uint32_t cnt = (uint32_t)ev0_valid + (uint32_t)ev1_valid + (uint32_t)ev2_valid;
      const uint32_t incl = wave_incl_scan(cnt, lane);
      const uint32_t nev = rdlane(incl, 63);
      uint32_t at = incl - cnt;
      if (ev0_valid) { if (ev0_bit) atomicOr(&L.ev[at >> 5], 1u << (at & 31)); at++; }
      if (ev1_valid) { if (ev1_bit) atomicOr(&L.ev[at >> 5], 1u << (at & 31)); at++; }
      if (ev2_valid) { if (ev2_bit) atomicOr(&L.ev[at >> 5], 1u << (at & 31)); at++; }
      wave_sync();
      uint32_t done = 0;
      while (done < nev) {                       // wave-uniform: whole zero runs at a time
        const uint32_t w = done >> 5, sh = done & 31;
        uint32_t word = rdfirst(L.ev[w]) >> sh;
        const uint32_t avail = min(32u - sh, nev - done);
        if (word == 0) { mel_zero_run(mel, L.mel, avail, lane); done += avail; continue; }
        const uint32_t z = (uint32_t)__builtin_ctz(word);
        if (z >= avail) { mel_zero_run(mel, L.mel, avail, lane); done += avail; continue; }
        mel_zero_run(mel, L.mel, z, lane);
        mel_one(mel, L.mel, lane);
        done += z + 1;
      }
      wave_sync();
      if (lane < 8) L.ev[lane] = 0;
    }

    // ---- MagSgn: OR the pair's bits into the flat buffer, then stuff a window at a time ----
    {
      uint32_t tot = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) tot += msl[i];
      const uint32_t incl = wave_incl_scan(tot, lane);
      const uint32_t T = ms_carry + rdlane(incl, 63);
      uint32_t at = ms_carry + incl - tot;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (S64) {
          const uint32_t lo = msl[i] < 32u ? msl[i] : 32u;
          or_bits(L.ms, at, (uint32_t)msv[i], lo);
          if (msl[i] > 32u) or_bits(L.ms, at + 32u, (uint32_t)(msv[i] >> 32), msl[i] - 32u);
        } else or_bits(L.ms, at, (uint32_t)msv[i], msl[i]);
        at += msl[i];
      }
      wave_sync();
      uint32_t pos = 0;
      for (;;) {
        const uint32_t first_n = ms_ff ? 7u : 8u;
        const uint32_t start = pos + (lane == 0 ? 0u : first_n + 8u * (uint32_t)(lane - 1));
        const uint32_t nb = lane == 0 ? first_n : 8u;
        const bool ok = start + nb <= T;
        const uint32_t v = ok ? get_bits(L.ms, start, nb) : 0u;
        const uint64_t m_ok = __ballot(ok), m_ff = __ballot(ok && v == 0xFF);
        const uint32_t n_ok = m_ok == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m_ok);
        const uint32_t f_ff = m_ff ? (uint32_t)__builtin_ctzll(m_ff) : 64u;
        const uint32_t nc = min(n_ok, f_ff + 1u);
        if (nc == 0) break;
        if (ms_k + nc > ms_cap) { err = 1; break; }
        if ((uint32_t)lane < nc) ms_out[ms_k + lane] = (uint8_t)v;
        ms_k += nc;
        pos += first_n + 8u * (nc - 1u);
        ms_ff = f_ff < n_ok ? 1u : 0u;
      }
      const uint32_t rem = T - pos;                       // < 8 bits stay for the next step
      const uint32_t cv = rem ? get_bits(L.ms, pos, rem) : 0u;
      wave_sync();
      const uint32_t used = (T >> 5) + 2;
      for (uint32_t i = lane; i < used && i < (uint32_t)MSW; i += 64) L.ms[i] = 0;
      wave_sync();
      if (lane == 0) L.ms[0] = cv;
      ms_carry = rem;
      wave_sync();
    }

    // ---- VLC: same idea, bytes grow downwards and the stuffing rule looks at the byte above ----
    {
      const uint32_t incl = wave_incl_scan(vl, lane);
      const uint32_t T = v_carry + rdlane(incl, 63);
      {
        const uint32_t at = v_carry + incl - vl, lo = vl < 32u ? vl : 32u;
        or_bits(L.vlc, at, (uint32_t)vb, lo);
        if (vl > 32u) or_bits(L.vlc, at + 32u, (uint32_t)(vb >> 32), vl - 32u);
      }
      wave_sync();
      uint32_t pos = 0;
      for (;;) {
        const uint32_t start = pos + 8u * (uint32_t)lane;
        const bool have7 = start + 7 <= T, have8 = start + 8 <= T;
        const uint32_t v8 = have7 ? get_bits(L.vlc, start, 8) : 0u;     // bits past T are zero
        uint32_t pv = __shfl_up(v8, 1);
        if (lane == 0) pv = v_prev;
        const bool special = have7 && pv > 0x8F && (v8 & 0x7F) == 0x7F;    // :386-405
        const bool stop = !have8 && !special;
        const uint64_t m_sp = __ballot(special), m_st = __ballot(stop);
        const uint32_t fs = m_sp ? (uint32_t)__builtin_ctzll(m_sp) : 64u;
        const uint32_t ft = m_st ? (uint32_t)__builtin_ctzll(m_st) : 64u;
        const uint32_t n8 = min(fs, ft);
        const bool sp = fs < ft;
        if (n8 == 0 && !sp) break;
        if (v_pos + n8 + 1 >= (uint32_t)VLC_CAP) { err = 1; break; }
        if ((uint32_t)lane < n8) *(vlc_last - (v_pos + lane)) = (uint8_t)v8;
        if (sp && (uint32_t)lane == fs) *(vlc_last - (v_pos + lane)) = 0x7F;
        const uint32_t last8 = n8 ? rdlane(v8, (int)(n8 - 1)) : v_prev;
        v_pos += n8 + (sp ? 1u : 0u);
        pos += 8u * n8 + (sp ? 7u : 0u);
        v_prev = sp ? 0x7Fu : last8;
      }
      const uint32_t rem = T - pos;
      const uint32_t cv = rem ? get_bits(L.vlc, pos, rem) : 0u;
      wave_sync();
      for (int i = lane; i < VLW; i += 64) L.vlc[i] = 0;
      wave_sync();
      if (lane == 0) L.vlc[0] = cv;
      v_carry = rem;
      wave_sync();
    }
    if (err) break;
  }

  err |= mel.err;
  uint32_t total = 0, ms_len = ms_k;
  if (!err && any_sig) {
    // ---- ms_terminate (:517-534) ----
    const uint32_t ms_tmp0 = rdfirst(L.ms[0]);
    if (ms_carry) {
      const uint32_t maxb = ms_ff ? 7u : 8u, t = maxb - ms_carry;
      const uint32_t tmp = ms_tmp0 | ((0xFFu & ((1u << t) - 1u)) << ms_carry);
      if (tmp != 0xFF) {
        if (ms_len >= ms_cap) err = 1;
        else { if (lane == 0) ms_out[ms_len] = (uint8_t)tmp; ms_len++; }
      }
    } else if (ms_ff) ms_len--;
    // ---- terminate_mel_vlc (:412-441) ----
    if (mel.run > 0) mel_put(mel, L.mel, 1, 1, lane);
    const uint32_t need = mel.lastff ? 7u : 8u, remaining = need - mel.nb;
    const uint32_t mel_tmp = (mel.acc << remaining) & 0xFFu;
    const uint32_t mel_mask = (0xFFu << remaining) & 0xFFu;
    const uint32_t vlc_tmp = rdfirst(L.vlc[0]) & 0xFFu;
    const uint32_t vlc_mask = v_carry ? (0xFFu >> (8 - v_carry)) : 0u;
    if ((mel_mask | vlc_mask) != 0) {
      if (mel.pos >= (uint32_t)MEL_CAP) err = 1;
      else {
        const uint32_t fuse = mel_tmp | vlc_tmp;
        if (((((fuse ^ mel_tmp) & mel_mask) | ((fuse ^ vlc_tmp) & vlc_mask)) == 0) && fuse != 0xFF && v_pos > 1) {
          if (lane == 0) L.mel[mel.pos] = (uint8_t)fuse;
          mel.pos++;
        } else {
          if (v_pos >= (uint32_t)VLC_CAP) err = 1;
          else {
            if (lane == 0) { L.mel[mel.pos] = (uint8_t)mel_tmp; *(vlc_last - v_pos) = (uint8_t)vlc_tmp; }
            mel.pos++; v_pos++;
          }
        }
      }
    }
    err |= mel.err;
    total = ms_len + mel.pos + v_pos;
  }
  wave_sync();
  __threadfence_block();

  // ---- claim a slot in the compacted output and copy MagSgn | MEL | VLC (:1003-1014) ----
  uint32_t off = 0;
  if (err) total = 0;
  if (total) {                                            // slots are 4-byte aligned (ht_encode_kernel stores dwords)
    off = claim_output(cursor, regions, nreg, bi, (total + 3u) & ~3u, out_cap, lane);
    if (off == 0xFFFFFFFFu) { err = 1; total = 0; off = 0; }
  }
  if (total) {
    const uint32_t scup = mel.pos + v_pos;
    uint8_t* dst = out + off;
    const uint8_t* vsrc = vlc_last - v_pos + 1;
    for (uint32_t i = lane; i < total; i += 64) {
      uint32_t b;
      if (i < ms_len) b = ms_out[i];
      else if (i < ms_len + mel.pos) b = L.mel[i - ms_len];
      else b = vsrc[i - ms_len - mel.pos];
      if (i == total - 1) b = scup >> 4;
      else if (i == total - 2) b = (b & 0xF0u) | (scup & 0xFu);
      dst[i] = (uint8_t)b;
    }
  }
  if (lane == 0) {
    results[bi].offset = off; results[bi].length = total;
    if (err) atomicOr(status, 1u);
  }
}

// -------------------------------------------------------------------------------------------------
// narrow blocks (W <= 64): one lane = one quad pair, four quad rows per step
// -------------------------------------------------------------------------------------------------
// lane = r * 16 + px: quad row 4 * step + r, quad pair px (sample columns 4 px .. 4 px + 3).  Pairs
// in lane order are pairs in raster order, so every stream position is a wavefront prefix sum.
//   * the lane loads its 2 x 4 samples as two 16-byte segments, requested one step ahead;
//   * what a quad needs from the sample row above (exponents for kappa, significance for the
//     context) is not recomputed from memory: every lane packs the exponents / significance of
//     its bottom sample row into one register, the lane 16 positions up (the row above in the same
//     step, or the last row of the previous step) hands it over with one ds_bpermute, and the left /
//     right neighbours of that come from DPP wave shifts;
//   * MagSgn and VLC bits are OR-ed into flat, un-stuffed LDS bit buffers; byte stuffing is
//     speculative (every lane proposes one byte, a ballot finds the first stuffing event, the
//     window restarts behind it) and LAZY: a window only runs when all 64 lanes have a full byte,
//     the < 64 pending bytes stay in the bit buffer until the next step or the final flush;
//   * coded bytes are staged in LDS (MagSgn grows up from 0, VLC grows down from the end, like
//     the reference's single buffer) and leave the CU once, as aligned dwords, when the block's
//     length is known; only blocks that outgrow the stage spill their MagSgn bytes to the HBM
//     scratch slot.
#ifndef ABL
#define ABL 0                           // ablation bits for attribution experiments (tools/enc_only.py, tools/enc_counters.sh); 0 in the product:
#endif                                  // 1 no MEL walk, 2 no MagSgn / VLC bits into LDS, 4 no byte stuffing windows, 8 symbols only,
                                        // 16 sample loads + quantise transfer only, 32 (with any) samples read as if stored block by block
#ifndef NWAVES
#define NWAVES 4                        // wavefronts (code-blocks) per workgroup of the narrow kernel
#endif
#ifndef NOUT_CAP
#define NOUT_CAP 5120
#endif
#ifndef NWAVES_PER_EU
#define NWAVES_PER_EU 4                 // register budget: wavefronts the kernel must fit per SIMD
#endif
constexpr uint32_t OUT_CAP = NOUT_CAP;  // bytes of coded output staged in LDS per wavefront
constexpr int PMS_WORDS = 576;          // 64 lanes * 8 samples * 31 bits + < 257 pending bytes (a window is 256 bytes)
constexpr int PVLC_WORDS = 112;         // 64 pairs * 30 bits of a step behind what earlier steps left pending (compacted on demand)

struct NarrowLds {
  uint32_t out[OUT_CAP / 4];
  uint32_t ms[PMS_WORDS];
  uint32_t vlc[PVLC_WORDS];
  uint8_t  mel[MEL_CAP + 8];             // (the raw MEL bit string, MEL_RAW_WORDS words, until the block ends; then its bytes)
};
static_assert(MEL_CAP + 8 >= 4 * (MEL_CAP / 4 + 2), "raw MEL words");
#ifndef NWG_PER_CU
#define NWG_PER_CU 4                    // workgroups of the narrow kernel that must fit one CU's 160 KB of LDS
#endif
static_assert(NWG_PER_CU * (2 * 2048 * 2 + 64 * 4 + NWAVES * sizeof(NarrowLds)) <= 160 * 1024, "the narrow kernel's workgroups do not fit the LDS as planned");

struct __attribute__((aligned(4))) U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t dpp_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false); }  // wave_shr:1 (0 into lane 0)
__device__ __forceinline__ uint32_t dpp_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }  // wave_shl:1 (0 into lane 63)

// U-VLC codeword of u (ojph_block_encoder.cpp:196-255), branch-free: prefix | plen<<8 | suffix<<16 | slen<<24
__device__ __forceinline__ uint32_t uvlc_word(uint32_t u)
{
  const uint32_t lo = u < 3u ? 1u : 0u, mid = (u >= 3u && u < 5u) ? 1u : 0u;
  const uint32_t pre = lo ? u : (mid ? 4u : 0u);
  const uint32_t pl = u < 3u ? u : 3u;
  const uint32_t suf = lo ? 0u : (mid ? u - 3u : u - 5u);
  const uint32_t sl = lo ? 0u : (mid ? 1u : 5u);
  return pre | (pl << 8) | (suf << 16) | (sl << 24);
}

// MEL coder of the narrow kernel (ojph_block_encoder.cpp:317-362).  The adaptive run-length state machine is serial -- what an
// event costs depends on the state k the events before it left -- and on content whose significance is scattered (a "1"
// event every few quads: the usual lossy rates) a scalar walk that produced every event's bits was the larger half of the
// whole encode (8K frame at 0.2 bytes per sample: 0.43 ms of block coding, 0.21 of it without the walk).  So the serial part
// does the minimum -- it carries the STATE from "1" event to "1" event -- and the bits are made by the lanes:
//   * the state is k and the zeros seen since the coder was last at the start of run k, kept as ONE number c = P[k] + run,
//     P[k] = the zeros it takes from k = 0 to reach k (0 1 2 3 5 7 9 13 17 21 29 37 53; beyond k = 12 every run is 32).  Zeros
//     just add to c; k and run at any moment follow from c with a shift, a population count and a leading-zero count on the
//     64-bit constant that has bit P[k] set for every k (mel_at).  A "1" event sends the coder to (k - 1, run 0), c = P[k - 1];
//   * the events of a step are brought into coding order across the lanes (a permute of the lanes' flag bits), one ballot gives
//     the "1" events of 64 consecutive events, and the scalar loop visits those only: it notes (k, c) at the event in the
//     event's lane and steps the state -- about 25 scalar instructions per "1" event, none per "0" event;
//   * every event lane then works out its own bits from (k, c) -- the '1' bits of the runs completed since the last "1" event,
//     the '0' and the e bits of the open run's length -- a prefix sum over their lengths places them, and they are OR-ed,
//     MSB first, into the raw bit string in LDS.  Zeros behind the last "1" event stay in c until the next one (or the
//     end of the block) asks for their bits; c is kept below 85 + what a step can add, so no lane has more than 24 bits.
// The byte stuffing ("7 bits after an 0xFF") is done once per block by the whole wavefront (mel_stuff).
constexpr uint64_t MEL_PM = (1ull << 0) | (1ull << 1) | (1ull << 2) | (1ull << 3) | (1ull << 5) | (1ull << 7) | (1ull << 9) | (1ull << 13) |
                            (1ull << 17) | (1ull << 21) | (1ull << 29) | (1ull << 37) | (1ull << 53);
constexpr uint32_t MEL_RAW_BITS = 8u * MEL_CAP;   // more raw bits than that cannot fit MEL_CAP bytes either
constexpr uint32_t MEL_RAW_WORDS = MEL_CAP / 4 + 2;   // (+ the word a code may straddle into, + the zero word mel_stuff reads behind the last one)

// the coder that started run k with c - P[k] zeros behind it, `c` zeros later: K = where it is now, run = the zeros of its
// open run, ones = the runs it has completed on the way (a '1' bit each)
__device__ __forceinline__ void mel_at(uint32_t k, uint32_t c, uint32_t& K, uint32_t& run, uint32_t& ones, uint32_t& pKm1)
{
  const uint32_t full = c >= 85u ? (c - 53u) >> 5 : 0u;        // whole runs of 32 at k = 12
  const uint32_t cc = c - 32u * full;
  if (cc >= 53u) { K = 12u; run = cc - 53u; pKm1 = 37u; }
  else {
    const uint64_t m = MEL_PM << (63u - cc);                   // the P[j] <= cc, the largest of them in the top set bit
    const uint32_t lz = (uint32_t)__builtin_clzll(m);          // (bit 0 of MEL_PM: m != 0)
    K = (uint32_t)__popcll(m) - 1u; run = lz;
    const uint64_t m2 = m & ~(0x8000000000000000ull >> lz);
    pKm1 = m2 ? cc - (uint32_t)__builtin_clzll(m2) : 0u;
  }
  ones = K - k + full;
}

// raw bit string -> MEL bytes (mel_emit_bit / byte emission of :326-347): `raw` holds total_bits bits, MSB first, a
// zero word behind them; bytes go to `out`.  Returns the state the termination code expects: bytes written, whether
// the last one was 0xFF, and the bits (fewer than a byte) left over.
__device__ __forceinline__ void mel_stuff(const uint32_t* raw, uint32_t total_bits, uint8_t* out, MelState& st, int lane)
{
  uint32_t pos = 0, bytes = 0, lastff = 0, err = 0;
  for (;;) {
    const uint32_t first = lastff ? 7u : 8u;
    const uint32_t start = pos + (lane == 0 ? 0u : first + 8u * ((uint32_t)lane - 1u));
    const uint32_t len = lane == 0 ? first : 8u;
    const bool have = start + len <= total_bits;
    const uint32_t w = start >> 5, sh = start & 31u;
    const uint32_t x = have ? __funnelshift_l(raw[w + 1], raw[w], sh) >> (32u - len) : 0u;
    const uint64_t m_have = __ballot(have), m_ff = __ballot(have && x == 0xFFu);
    const uint32_t n_ok = m_have == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m_have);
    const uint32_t f_ff = m_ff ? (uint32_t)__builtin_ctzll(m_ff) : 64u;
    const uint32_t nc = min(n_ok, f_ff + 1u);
    if (nc == 0) break;
    if (bytes + nc > (uint32_t)MEL_CAP) { err = 1; break; }
    if ((uint32_t)lane < nc) out[bytes + lane] = (uint8_t)x;
    bytes += nc;
    pos += first + 8u * (nc - 1u);
    lastff = f_ff < n_ok ? 1u : 0u;
  }
  const uint32_t left = total_bits - pos;                  // < 8
  const uint32_t w = pos >> 5, sh = pos & 31u;
  const uint32_t x = left ? rdfirst(__funnelshift_l(raw[w + 1], raw[w], sh)) >> (32u - left) : 0u;
  st.k = 0; st.run = 0; st.acc = x; st.nb = left; st.pos = bytes; st.lastff = lastff; st.err = err;
}

// REV: the quantise transfer of the blocks this instantiation codes (5/3 integer or 9/7 float coefficients) is a
// compile-time property -- a launch over blocks of both kinds runs both instantiations, each skipping the other's.
// LOGP: a step covers 2^LOGP quad pairs across (lane & (2^LOGP - 1)) by 64 / 2^LOGP quad rows down.  4 = the layout
// described above (blocks up to 64 columns, 4 quad rows per step); 3 = blocks up to 32 columns (the 32 x 32 blocks of
// the IMF profile), 8 quad rows per step -- with the 16-pair layout half of the lanes of such a block would idle.
template <bool REV, int LOGP>
__global__ __launch_bounds__(64 * NWAVES) __attribute__((amdgpu_waves_per_eu(NWAVES_PER_EU, 8))) void ht_encode_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint32_t* __restrict__ coef,
    uint8_t* __restrict__ scratch, uint8_t* __restrict__ out, uint32_t out_cap,
    ojphgpu_cb_result* __restrict__ results, uint32_t* __restrict__ cursor, uint32_t* __restrict__ status,
    const uint32_t* __restrict__ regions, uint32_t nreg)
{
  __shared__ __attribute__((aligned(16))) uint16_t s_vlc[2][2048];
  __shared__ uint32_t s_uvlc[64];                   // U-VLC codewords of u = 0..63 (u <= 31 here), see uvlc_word
  __shared__ NarrowLds s_wave[NWAVES];
  // (16 bytes per lane and turn.  A workgroup lives for 12-40 us and meets at the barrier below before anything else: with
  // 2-byte turns, sixteen of them, the 8K frame's encode was 0.458 ms instead of 0.440, 32 x 32 blocks 0.603 instead of 0.562)
  for (int i = threadIdx.x; i < 2 * 2048 / 8; i += blockDim.x)
    reinterpret_cast<uint4*>(&s_vlc[0][0])[i] = reinterpret_cast<const uint4*>(&ojphgpu::g_enc_vlc[0][0])[i];
  if (threadIdx.x < 64) s_uvlc[threadIdx.x] = uvlc_word(threadIdx.x);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and the compiler knows it
  const uint32_t bi = blockIdx.x * NWAVES + wave;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  const uint32_t W = d.w, H = d.h;
  constexpr uint32_t PPR = 1u << LOGP, RPS = 64u >> LOGP;                 // quad pairs per row of a step, quad rows per step
  // LOGP 5 (32 pairs by 2 quad rows): the blocks of 65..128 columns of a launch none of whose blocks is wider (128 x 32)
  const bool my_width = LOGP == 5 ? (W > NARROW_MAX_W && W <= 4u * PPR) : (W <= NARROW_MAX_W && W <= 4u * PPR);
  if (!my_width || ((d.reversible & 1u) != 0) != REV || (d.reversible & 4u) != 0) return;   // ht_encode_wide_kernel's, or another instantiation's
  if (W == 0 || H == 0) { if (lane == 0) { results[bi].offset = 0; results[bi].length = 0; } return; }
  NarrowLds& L = s_wave[wave];
  uint8_t* outb = reinterpret_cast<uint8_t*>(L.out);
  const uint32_t K = d.K_max, p = 31u - K;      // missing_msbs = K_max - 1, p = 30 - missing_msbs
  constexpr bool rev = REV;
  const float delta_inv = rev ? 0.0f : __fdiv_rn(1.0f, d.delta);         // ojph_codeblock.cpp:98
  const uint32_t* src = coef + d.coef_off;
  const uint32_t pitch = d.pitch;
  uint8_t* ms_spill = scratch + d.data_off;                              // only used when the LDS stage overflows
  const uint32_t ms_cap = d.scratch_cap > (uint32_t)VLC_CAP ? d.scratch_cap - VLC_CAP : 0;
  const uint32_t QW = (W + 1) >> 1, QH = (H + 1) >> 1, PW = (QW + 1) >> 1;
  const uint32_t nsteps = (QH + RPS - 1u) / RPS;

  for (int i = lane; i < PMS_WORDS; i += 64) L.ms[i] = 0;
  for (int i = lane; i < PVLC_WORDS; i += 64) L.vlc[i] = 0;
  if (lane < (int)MEL_RAW_WORDS) reinterpret_cast<uint32_t*>(L.mel)[lane] = 0;
  wave_sync();
  if (lane == 0) { L.vlc[0] = 0xF; outb[OUT_CAP - 1] = 0xFF; }            // vlc_init: head byte, 4 bits already used (:365-375)
  wave_sync();

  // wave-uniform stream state
  uint32_t ms_pend = 0, ms_base = 0, ms_k = 0, ms_ff = 0;   // pending bits in L.ms and where they start, bytes written, last byte was 0xFF
  uint32_t v_pend = 4, v_base = 0, v_pos = 1, v_prev = 0xFF;   // pending bits in L.vlc and where they start, bytes "written" (incl. the head), last byte
  uint32_t mel_k = 0, mel_c = 0, mel_bits = 0, mel_err = 0;           // MEL coder: state (see mel_at), raw bits written
  uint32_t* const mel_raw = reinterpret_cast<uint32_t*>(L.mel);       // raw MEL bit string until the block ends, then its bytes
  // `len` bits of `code` at bit `pos` of the raw string, MSB first (len <= 24)
  auto mel_or = [&](uint32_t code, uint32_t len, uint32_t pos) {
    if (len) {
      const uint32_t sh = pos & 31u, w = pos >> 5;
      const uint64_t v = (uint64_t)code << (64u - sh - len);
      atomicOr(&mel_raw[w], (uint32_t)(v >> 32));
      if (sh + len > 32u) atomicOr(&mel_raw[w + 1u], (uint32_t)v);
    }
  };
  // '1' bits of completed runs, from the scalar side (lane 0 writes them): n <= 24
  auto mel_ones = [&](uint32_t n) {
    if (n == 0u) return;
    if (mel_bits + n > MEL_RAW_BITS) mel_err = 1;
    else if (lane == 0) mel_or((1u << n) - 1u, n, mel_bits);
    mel_bits += n;
  };
  // zeros that no "1" event follows in this step: c keeps them; once the coder is through a run at k = 12 the finished runs
  // leave their bits (c stays below 85 between steps)
  auto mel_zeros = [&](uint32_t n) {
    mel_c += n;
    if (mel_c >= 85u) {
      const uint32_t full = (mel_c - 53u) >> 5;
      mel_ones(12u - mel_k + full);
      mel_k = 12u; mel_c -= 32u * full;
    }
  };
  uint32_t err = 0, any_sig = 0;
  bool prev_sig = false;                                 // the previous step had a significant sample

  // The output stage in LDS holds the bytes [ms_out, ms_k) of the MagSgn stream (growing up from outb[0]) and the VLC
  // bytes (growing down from the top).  Most blocks fit; when the two would meet (more than ~5 KB: deep samples at
  // high rates, e.g. 16-bit lossless at 1.7 bytes per sample) the whole dwords of the MagSgn part are flushed to the
  // block's 64-byte aligned scratch slot in HBM, the 0..3 bytes left over move to the front, and coding goes on in
  // LDS -- so every byte that leaves for HBM leaves in aligned dwords, also at the end.
  uint32_t ms_out = 0;
  auto flush_stage = [&]() -> bool {
    const uint32_t have = ms_k - ms_out, nwd = have >> 2;
    if (ms_out + 4u * nwd > ms_cap) return false;
    uint32_t* g = reinterpret_cast<uint32_t*>(ms_spill + ms_out);
    for (uint32_t i = lane; i < nwd; i += 64) g[i] = L.out[i];
    const uint32_t keep = L.out[nwd];                     // the bytes beyond the left-over ones are rewritten by what comes next
    wave_sync();
    if (lane == 0) L.out[0] = keep;
    wave_sync();
    ms_out += 4u * nwd;
    return true;
  };

  const uint32_t r = (uint32_t)lane >> LOGP, px = (uint32_t)lane & (PPR - 1u);
  const bool pxok = px < PW;
  const uint32_t x0 = 4u * px;
  const bool has_q1 = pxok && x0 + 2 < W;
  uint32_t last_S = 0;                                   // what the last quad row of the previous step hands to the first one of this step

  // Samples of one quad row of this lane's pair: top[0..3], bot[0..3].  The loads are UNCONDITIONAL: rows and columns
  // are clamped into the block, a lane / row / column that does not exist fetches something valid and is masked when
  // the values are consumed, one step later.  (With a branch per case the two rows of the full-width case were
  // separated by a wait for everything in flight -- the zero-initialisation of the registers the partial-width case
  // loads into -- and the wavefront sat out a full memory latency in the middle of every step: SQ_WAIT_ANY 52 %.)
  // Whether the block is a multiple of four columns wide is wave-uniform: the usual blocks take two 16-byte loads per
  // lane, the others eight dword loads with clamped columns.
  const bool w4 = (W & 3u) == 0;
  auto load_rows = [&](uint32_t qy, uint32_t* top, uint32_t* bot) {
    const uint32_t qyc = min(qy, QH - 1u);
    const uint32_t y0 = 2u * qyc, y1 = min(y0 + 1u, H - 1u);
    const uint32_t* r0 = src + (size_t)y0 * pitch;
    const uint32_t* r1 = src + (size_t)y1 * pitch;
    if (ABL & 32) { r0 = coef + (size_t)bi * 4096u + y0 * 64u; r1 = coef + (size_t)bi * 4096u + y1 * 64u; }   // timing experiment: block-contiguous samples
    if (w4) {
      const uint32_t xc = min(x0, W - 4u);
      const U4 a = *reinterpret_cast<const U4*>(r0 + xc);
      const U4 c = *reinterpret_cast<const U4*>(r1 + xc);
      top[0] = a.x; top[1] = a.y; top[2] = a.z; top[3] = a.w;
      bot[0] = c.x; bot[1] = c.y; bot[2] = c.z; bot[3] = c.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const uint32_t xk = min(x0 + (uint32_t)k, W - 1u); top[k] = r0[xk]; bot[k] = r1[xk]; }
    }
  };
  // Stuffs whole 256-byte windows of the MagSgn bit buffer ("after 0xFF only 7 bits", :471-491): a lane
  // takes four consecutive bytes, speculating that none of the window's bytes is 0xFF; the window is cut
  // behind the first 0xFF (the byte after it carries 7 bits and shifts everything that follows) and the
  // next one starts there.  `flush` also emits the final partial window.  Returns the bit position reached.
  auto ms_windows = [&](uint32_t base, uint32_t T, bool flush) -> uint32_t {
    uint32_t pos = base;
    for (;;) {
      const uint32_t first_n = ms_ff ? 7u : 8u;
      if (!flush && pos + first_n + 8u * 255u > T) break;
      // byte j of the window starts at bit pos + (j ? first_n + 8 (j - 1) : 0)
      const uint32_t start = pos + (lane == 0 ? 0u : first_n + 8u * (4u * (uint32_t)lane - 1u));
      const uint32_t lead = lane == 0 ? first_n : 8u;                 // bits of the lane's first byte
      uint32_t v = 0, cnt = 0;                                        // the lane's bits; its complete bytes (0..4)
      if (start + lead <= T) {
        const uint32_t w = start >> 5, sh = start & 31u;
        v = __funnelshift_r(L.ms[w], L.ms[w + 1], sh);               // bits past T are zero
        cnt = min(4u, 1u + ((T - start - lead) >> 3));
      }
      // the four bytes in byte positions (lane 0's first byte may have 7 bits)
      const uint32_t bytes = (lane == 0 && first_n == 7u) ? ((v & 0x7Fu) | ((v >> 7) << 8)) : v;
      const uint32_t have = cnt == 4u ? 0xFFFFFFFFu : (1u << (8u * cnt)) - 1u;
      const uint32_t ffm = (((bytes & 0x7F7F7F7Fu) + 0x01010101u) & bytes & 0x80808080u) & have;   // 0x80 in every complete byte that is 0xFF
      const uint64_t m_full = __ballot(cnt == 4u), m_ff = __ballot(ffm != 0u);
      const uint32_t nfull = m_full == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m_full);
      const uint32_t n_ok = nfull == 64u ? 256u : 4u * nfull + rdlane(cnt, (int)nfull);
      uint32_t f_ff = 256u;
      if (m_ff) { const uint32_t lf = (uint32_t)__builtin_ctzll(m_ff); f_ff = 4u * lf + ((uint32_t)__builtin_ctz(rdlane(ffm, (int)lf)) >> 3); }
      const uint32_t nc = min(n_ok, f_ff + 1u);
      if (nc == 0) break;
      if (ms_k - ms_out + nc + v_pos + 72u > OUT_CAP && !flush_stage()) { err = 1; break; }   // the stage is full
      uint8_t* dstb = outb + (ms_k - ms_out) + 4u * (uint32_t)lane;
      const uint32_t mine = nc > 4u * (uint32_t)lane ? min(4u, nc - 4u * (uint32_t)lane) : 0u;
      if (mine > 0) dstb[0] = (uint8_t)bytes;
      if (mine > 1) dstb[1] = (uint8_t)(bytes >> 8);
      if (mine > 2) dstb[2] = (uint8_t)(bytes >> 16);
      if (mine > 3) dstb[3] = (uint8_t)(bytes >> 24);
      ms_k += nc;
      pos += first_n + 8u * (nc - 1u);
      ms_ff = f_ff < n_ok ? 1u : 0u;
    }
    return pos;
  };
  // The same for the windows of a step, where every lane of a window has its four bytes by construction (the loop only
  // runs while 256 whole bytes are pending): no per-lane byte counts, lane 0's 7-bit byte only when the previous window
  // ended on an 0xFF, and the lane's four bytes leave as ONE (unaligned) dword store -- the lane at the cut writes up to
  // three bytes too many, into space the next window overwrites (the capacity test keeps 72 bytes free above the cursor).
  auto ms_windows_full = [&](uint32_t base, uint32_t T) -> uint32_t {
    uint32_t pos = base;
    for (;;) {
      const uint32_t first_n = ms_ff ? 7u : 8u;
      if (pos + first_n + 8u * 255u > T) break;
      const uint32_t start = lane == 0 ? pos : pos + first_n - 8u + 32u * (uint32_t)lane;
      const uint32_t w = start >> 5, sh = start & 31u;
      uint32_t bytes = __funnelshift_r(L.ms[w], L.ms[w + 1], sh);
      if (ms_ff) bytes = lane == 0 ? ((bytes & 0x7Fu) | ((bytes >> 7) << 8)) : bytes;
      const uint32_t ffm = ((bytes & 0x7F7F7F7Fu) + 0x01010101u) & bytes & 0x80808080u;   // 0x80 in every byte that is 0xFF
      const uint64_t m_ff = __ballot(ffm != 0u);
      uint32_t nc = 256u, ff = 0u;
      if (m_ff) { const uint32_t lf = (uint32_t)__builtin_ctzll(m_ff); nc = 4u * lf + ((uint32_t)__builtin_ctz(rdlane(ffm, (int)lf)) >> 3) + 1u; ff = 1u; }
      if (ms_k - ms_out + nc + v_pos + 72u > OUT_CAP && !flush_stage()) { err = 1; break; }   // the stage is full
      if (4u * (uint32_t)lane < nc) __builtin_memcpy(outb + (ms_k - ms_out) + 4u * (uint32_t)lane, &bytes, 4);
      ms_k += nc;
      pos += first_n + 8u * (nc - 1u);
      ms_ff = ff;
    }
    return pos;
  };
  // same for the VLC buffer: bytes grow downwards, the rule looks at the byte above (:386-405)
  auto vlc_windows_full = [&](uint32_t base, uint32_t T) -> uint32_t {     // whole 64-byte windows only
    uint32_t pos = base;
    for (;;) {
      if (pos + 8u * 64u > T) break;
      const uint32_t v8 = get_bits(L.vlc, pos + 8u * (uint32_t)lane, 8);
      uint32_t pv = dpp_prev(v8);
      if (lane == 0) pv = v_prev;
      const uint64_t m_sp = __ballot(pv > 0x8Fu && (v8 & 0x7Fu) == 0x7Fu);
      uint32_t n8 = 64u, sp = 0u;
      if (m_sp) { n8 = (uint32_t)__builtin_ctzll(m_sp); sp = 1u; }
      if (v_pos + n8 + 1 >= (uint32_t)VLC_CAP) { err = 1; break; }
      if (ms_k - ms_out + v_pos + n8 + 72u > OUT_CAP && !flush_stage()) { err = 1; break; }
      if ((uint32_t)lane < n8 + sp) outb[OUT_CAP - 1 - (v_pos + lane)] = (uint8_t)((uint32_t)lane == n8 ? 0x7Fu : v8);
      const uint32_t last8 = n8 ? rdlane(v8, (int)(n8 - 1)) : v_prev;
      v_pos += n8 + sp;
      pos += 8u * n8 + 7u * sp;
      v_prev = sp ? 0x7Fu : last8;
    }
    return pos;
  };
  auto vlc_windows = [&](uint32_t base, uint32_t T, bool flush) -> uint32_t {
    uint32_t pos = base;
    for (;;) {
      if (!flush && pos + 8u * 64u > T) break;
      const uint32_t start = pos + 8u * (uint32_t)lane;
      const bool have7 = start + 7 <= T, have8 = start + 8 <= T;
      const uint32_t v8 = have7 ? get_bits(L.vlc, start, 8) : 0u;     // bits past T are zero
      uint32_t pv = dpp_prev(v8);
      if (lane == 0) pv = v_prev;
      const bool special = have7 && pv > 0x8F && (v8 & 0x7F) == 0x7F;
      const bool stop = !have8 && !special;
      const uint64_t m_sp = __ballot(special), m_st = __ballot(stop);
      const uint32_t fs = m_sp ? (uint32_t)__builtin_ctzll(m_sp) : 64u;
      const uint32_t ft = m_st ? (uint32_t)__builtin_ctzll(m_st) : 64u;
      const uint32_t n8 = min(fs, ft);
      const bool sp = fs < ft;
      if (n8 == 0 && !sp) break;
      if (v_pos + n8 + 1 >= (uint32_t)VLC_CAP) { err = 1; break; }
      if (ms_k - ms_out + v_pos + n8 + 72u > OUT_CAP && !flush_stage()) { err = 1; break; }
      if ((uint32_t)lane < n8) outb[OUT_CAP - 1 - (v_pos + lane)] = (uint8_t)v8;
      if (sp && (uint32_t)lane == fs) outb[OUT_CAP - 1 - (v_pos + lane)] = 0x7F;
      const uint32_t last8 = n8 ? rdlane(v8, (int)(n8 - 1)) : v_prev;
      v_pos += n8 + (sp ? 1u : 0u);
      pos += 8u * n8 + (sp ? 7u : 0u);
      v_prev = sp ? 0x7Fu : last8;
    }
    return pos;
  };
  // moves the un-emitted bits [pos, T) of a bit buffer to its front and clears the rest
  auto compact = [&](uint32_t* buf, uint32_t pos, uint32_t T, uint32_t nwords) -> uint32_t {
    const uint32_t rem = T - pos, nw = (rem + 31u) >> 5;           // nw <= 64 (MagSgn: < 2048 pending bits), <= 17 (VLC)
    const uint32_t w0 = pos >> 5, sh = pos & 31u;
    uint32_t keep = 0;
    if ((uint32_t)lane < nw) {
      keep = __funnelshift_r(buf[w0 + lane], buf[w0 + lane + 1], sh);
      if (lane == (int)nw - 1 && (rem & 31u)) keep &= (1u << (rem & 31u)) - 1u;
    }
    wave_sync();
    const uint32_t used = min((T >> 5) + 2u, nwords);
    for (uint32_t i = lane; i < used; i += 64) buf[i] = 0;
    wave_sync();
    if ((uint32_t)lane < nw) buf[lane] = keep;
    wave_sync();
    return rem;
  };

  // ---- per-lane constants of the block ----
  // The quantise transfer of a sample is ONE multiply (9/7) or ONE and (5/3) by a per-lane, per-column constant that is
  // zero for the columns of this lane outside the block: what the clamped loads fetched there quantises to zero, so
  // rho, exponents and MagSgn lengths of samples that do not exist come out as zero without a select per sample.
  // 9/7: floor(|x| * (1/delta)) >> p == floor(|x| * ((1/delta) * 2^-p)) -- the scaling by a power of two is exact.
  float colf[4]; uint32_t colm[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = pxok && x0 + (uint32_t)k < W;
    colf[k] = in ? ldexpf(delta_inv, -(int)p) : 0.0f;
    colm[k] = in ? (1u << (31u - p)) - 1u : 0u;                          // the magnitude bits the coder keeps: mu < 2^K_max
  }
  const bool ragged = (QH & (RPS - 1u)) != 0 || (H & 1u) != 0;           // the last step has quad rows / sample rows that do not exist

  // one 32-bit value of at most 32 bits OR-ed into a bit buffer at bit position pos (the second word gets zeros when the
  // value does not straddle).  Lanes with nothing to add stay out: a lane without bits has the position of the next lane
  // that has some, and on sparse content -- most quads of a step without a significant sample -- sixty lanes sent their
  // zeros to ONE word, same-address LDS atomics that the bank serves one after the other (8K frame at 0.01 bytes per
  // sample: 0.46 ms of block coding, 0.18 with these ten atomics per lane and step left out altogether).
  // (the callers test once per buffer: `tot` / `vl`, the lane's bits of the step)
  auto or32 = [&](uint32_t* buf, uint32_t pos, uint32_t v) {
    const uint32_t sh = pos & 31u;
    uint32_t* wp = buf + (pos >> 5);
    atomicOr(wp, v << sh);
    atomicOr(wp + 1, (v >> 1) >> (sh ^ 31u));
  };
  // bytes of x that are not zero -> 0x80 in that byte (bytes < 0x80); 0x80-flags of four bytes -> bits 0..3
  auto nz_flags = [](uint32_t x) -> uint32_t { return (x + 0x7F7F7F7Fu) & 0x80808080u; };
  auto gather4 = [](uint32_t f) -> uint32_t { return ((f >> 7) * 0x10204080u) >> 28; };
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  auto pk_max = [](uint32_t a, uint32_t b2) -> uint32_t {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b2)));
  };
  // neighbour lanes of the same quad row (zero at the row's ends)
  auto left_of = [&](uint32_t v) -> uint32_t {
    if (LOGP == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
    const uint32_t t = dpp_prev(v); return px == 0 ? 0u : t;
  };
  auto right_of = [&](uint32_t v) -> uint32_t {
    if (LOGP == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, false);      // row_shl:1
    const uint32_t t = dpp_next(v); return px == PPR - 1u ? 0u : t;
  };

  uint32_t ntop[4], nbot[4];
  load_rows(r, ntop, nbot);
  for (uint32_t step = 0; step < nsteps && !err; ++step) {
    const uint32_t qy = RPS * step + r;
    const bool active = pxok && qy < QH;
    // ---- quantise transfer + the coder's view of a sample (ojph_codestream_gen.cpp:59-121, ojph_block_encoder.cpp:592-601):
    // mu = the magnitude above bit-plane p (the reference's val = 2 mu), the exponent e = bit length of 2 mu - 1, and
    // sv = 2 (mu - 1) + sign, what MagSgn takes its bits from.  The sign-magnitude word of the reference is never formed.
    if (ragged && step + 1 == nsteps) {                 // rows below the block: what was fetched for them counts as zero
      const bool bot = active && 2u * qy + 1u < H;
#pragma unroll
      for (int k = 0; k < 4; ++k) { ntop[k] = active ? ntop[k] : 0u; nbot[k] = bot ? nbot[k] : 0u; }
    }
    uint32_t mu[8], sx[8];                              // sx: a word whose bit 31 is the sample's sign
    uint32_t nz = 0;                                    // OR of the step's magnitudes (rev: before the mask, bit K_max included)
#pragma unroll
    for (int k = 0; k < 4; ++k) {                       // quad order: n = 0:(x,y) 1:(x,y+1) 2:(x+1,y) 3:(x+1,y+1)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t raw = j ? nbot[k] : ntop[k];
        if (rev) {
          const int v = (int)raw;
          const uint32_t a = (uint32_t)(v >= 0 ? v : -v);
          mu[2 * k + j] = a & colm[k];                                        // :70-76, with the same wrap-around when |v| >= 2^K_max:
          sx[2 * k + j] = raw | (a << p);                                     // its top bit lands on the sign
          nz |= a;                                                            // (clamped loads: every lane holds samples of the block's own rows)
        } else {
          mu[2 * k + j] = (uint32_t)__fmul_rn(fabsf(__uint_as_float(raw)), colf[k]);   // :113-118, C truncation
          sx[2 * k + j] = raw;
          nz |= mu[2 * k + j];
        }
      }
    }
    // A magnitude of more than K_max bits (Part-2 kernels whose gain outruns the guard bits; see to_sign_mag): reversible,
    // the bits above K_max are dropped and bit K_max has landed on the sign; irreversible, the reference's conversion of a
    // product beyond 2^31 returns INT_MIN -- the sample codes as a zero.  Both leave a bit in the reference's max_val,
    // and a block with a non-zero max_val is coded even if none of its samples is significant (ojph_codeblock.cpp:142-175).
    if (__ballot((nz >> K) != 0u) != 0ull) {            // (wave-uniform, and never taken by a Part-1 codestream)
      any_sig |= (rev ? __ballot(((nz >> K) & 1u) != 0u) != 0ull : true) ? 1u : 0u;
      if (!rev) {
        nz = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { mu[i] = (mu[i] >> K) ? 0u : mu[i]; nz |= mu[i]; }
      }
    }
    if (rev) nz &= (1u << K) - 1u;                      // = the OR of the mu of the block's samples
    if (step + 1 < nsteps) load_rows(qy + RPS, ntop, nbot); // request the next step's samples now
    if (ABL & 16) { any_sig |= (uint32_t)(__ballot((mu[0] ^ mu[1] ^ mu[2] ^ mu[3] ^ mu[4] ^ mu[5] ^ mu[6] ^ mu[7]) == 0x12345u) != 0ull); continue; }
    // A step without a significant sample, below a step without one: every quad has context 0 and rho 0, i.e. one
    // MEL "0" event and nothing else (no VLC codeword, no U-VLC, no MagSgn bits).  The events are all alike, so only
    // their number matters.  (Smooth content at moderate rates is mostly such steps in the top resolution's sub-bands.)
    {
      const bool step_sig = __ballot(nz != 0u) != 0ull;
      const bool skip = !step_sig && !prev_sig;
      prev_sig = step_sig;
      if (skip) {
        const uint32_t nq = (uint32_t)__popcll(__ballot(active)) + (uint32_t)__popcll(__ballot(has_q1 && active));
        mel_zeros(nq);
        last_S = 0;
        continue;
      }
      any_sig |= step_sig ? 1u : 0u;
    }
    uint32_t e[8], sv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t m1 = mu[i] - 1u, w = m1 + mu[i];                         // 2 mu - 1 (all ones when mu = 0)
      e[i] = (0u - (uint32_t)__builtin_clz(w)) & 31u;                         // 32 - clz(2 mu - 1), 0 for mu = 0 (w is never 0)
      sv[i] = __builtin_amdgcn_alignbit(m1, sx[i], 31);                       // 2 (mu - 1) + sign (:601)
    }
    // exponents of a quad as four bytes; its significance pattern rho and, below, which samples reach the maximum
    const uint32_t ep0 = e[0] | (e[1] << 8) | (e[2] << 16) | (e[3] << 24);
    const uint32_t ep1 = e[4] | (e[5] << 8) | (e[6] << 16) | (e[7] << 24);
    const uint32_t sf0 = nz_flags(ep0), sf1 = nz_flags(ep1);
    const uint32_t rho0 = gather4(sf0), rho1 = gather4(sf1);
    const uint32_t emax0 = max(max(e[0], e[1]), max(e[2], e[3])), emax1 = max(max(e[4], e[5]), max(e[6], e[7]));

    // ---- what the quad row below needs from this one, worked out HERE and handed down as one word per lane: for each
    // of its two quads the largest exponent among the four bottom-row samples above it (columns x-1 .. x+2: kappa) in
    // bits 0..4 of a 16-bit half, and whether the two left / the two right ones of them hold a significant sample
    // (context bits "nw | n" and "ne | nf") in bits 5 / 7 -- :802, :862, :878, :950-:991.
    uint32_t S;
    {
      const uint32_t b0 = e[1], b1 = e[3], b2 = e[5], b3 = e[7];             // bottom-row exponents of columns 0..3
      const uint32_t m12 = max(b1, b2);
      const uint32_t own = max(b0, m12) | (max(m12, b3) << 16);              // columns 0..2 (quad 0 below) | columns 1..3 (quad 1)
      const uint32_t X = b3 | (b0 << 16), Y = b0 | (b3 << 16);
      const uint32_t nb = (left_of(X) & 0xFFFFu) | (right_of(X) & 0xFFFF0000u);   // left lane's column 3 | right lane's column 0
      const uint32_t mx = pk_max(own, nb);
      const uint32_t F = (pk_max(nb, Y) + 0x007F001Fu) & 0x00800020u;        // quad 0: bit 5 = nw | n; quad 1: bit 7 = ne | nf
      const uint32_t f12 = ((m12 + 31u) & 32u) * 0x10004u;                   // columns 1, 2: quad 0's ne | nf (bit 7), quad 1's nw | n (bit 5)
      S = mx | F | f12;
    }
    const uint32_t give = (uint32_t)lane >= 64u - PPR ? last_S : S;
    const uint32_t above = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane - (int)PPR) & 63) << 2), (int)give);
    last_S = S;                                          // (the first quad row of the block gets the zero last_S starts with)

    // ---- per quad symbols ----
    const uint32_t rho_left = left_of(rho1);            // rho of the quad to the left of quad 0
    uint32_t uq[2], tup[2], Uq[2], chi[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t rl = q == 0 ? rho_left : rho0, rho = q ? rho1 : rho0, emax = q ? emax1 : emax0, ep = q ? ep1 : ep0;
      const uint32_t aq = q ? above >> 16 : above;
      // context, kept as c << 5: rows below the first take nw / ne from above and "a quad to the left has a significant
      // sample in its right column" (rho_left >= 4); the first row takes all three bits from the left quad (:731, :788)
      uint32_t c5 = (aq & 0xA0u) | ((4u * rl + 48u) & 64u);
      uint32_t tsel = 2048u;
      if (step == 0) {
        const bool fr = qy == 0;
        c5 = fr ? ((rl >> 1) | (rl & 1u)) << 5 : c5;
        tsel = fr ? 0u : 2048u;
      }
      // kappa = max(1, max exponent above - 1) for a quad with more than one significant sample, else 1 (:862, :950)
      const uint32_t multi = 0u - ((0xFEE8u >> rho) & 1u);
      const uint32_t me = aq & 31u;
      const uint32_t kappa = 1u + ((me > 2u ? me - 2u : 0u) & multi);
      const uint32_t U = max(emax, kappa), u = U - kappa;
      const uint32_t xe = ep ^ (emax * 0x01010101u);
      uint32_t eps = gather4((nz_flags(xe) ^ 0x80808080u));               // samples whose exponent is the quad's maximum
      eps = u ? eps : 0u;
      tup[q] = (&s_vlc[0][0])[tsel + (c5 << 3) + (rho << 4) + eps];
      uq[q] = u; Uq[q] = U; chi[q] = c5;
    }

    // ---- VLC bits of the pair: cwd(q0) cwd(q1) then the interleaved U-VLC ----
    uint32_t vb = 0, vl = 0;
    bool ev2_valid = false; uint32_t ev2_bit = 0;
    {
      const uint32_t u0 = uq[0], u1 = has_q1 ? uq[1] : 0u;
      vb = tup[0] >> 8; vl = (tup[0] >> 4) & 7u;
      if (has_q1) { vb |= (tup[1] >> 8) << vl; vl += (tup[1] >> 4) & 7u; }
      const bool first_row = qy == 0;
      const bool both_big = first_row && u0 > 2 && u1 > 2;                                   // :766-772
      const bool one_big = first_row && !both_big && u0 > 2 && u1 > 0;                       // :773-778
      ev2_valid = first_row && u0 > 0 && u1 > 0; ev2_bit = min(u0, u1) > 2;                   // :763-764
      const uint32_t w0 = s_uvlc[both_big ? u0 - 2u : u0];
      uint32_t w1 = s_uvlc[both_big ? u1 - 2u : u1];
      if (one_big) w1 = (u1 - 1u) | (1u << 8);          // u1 in {1,2} is a single bit, no suffix
      vb |= (w0 & 0xFFu) << vl; vl += (w0 >> 8) & 0xFFu;                                     // prefix q0, prefix q1,
      vb |= (w1 & 0xFFu) << vl; vl += (w1 >> 8) & 0xFFu;                                     // suffix q0, suffix q1 (:779-785, :985-988)
      vb |= ((w0 >> 16) & 0xFFu) << vl; vl += w0 >> 24;
      vb |= ((w1 >> 16) & 0xFFu) << vl; vl += w1 >> 24;
      if (!active) { vb = 0; vl = 0; ev2_valid = false; }
    }

    // ---- MagSgn lengths: m_n = U - e_k bit for a significant sample, 0 otherwise (:667-674), four bytes per quad ----
    uint32_t mp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t sf = q ? sf1 : sf0;
      const uint32_t sig = (sf - (sf >> 7)) | sf;                             // 0xFF in the bytes of significant samples
      const uint32_t ek = ((tup[q] & 15u) * 0x204081u) & 0x01010101u;         // the four e_k bits, one per byte
      mp[q] = (Uq[q] * 0x01010101u - ek) & sig;
    }
    const uint32_t tot0 = __builtin_amdgcn_sad_u8(mp[0], 0u, 0u), tot = __builtin_amdgcn_sad_u8(mp[1], 0u, tot0);

    // ---- stream positions of the lane's MagSgn and VLC bits: ONE wavefront prefix sum over both counts ----
    const uint32_t incl = wave_incl_scan(tot | (vl << 16), lane);
    const uint32_t sums = rdlane(incl, 63);
    const uint32_t step_bits = sums & 0xFFFFu, step_vbits = sums >> 16;
    // the pending bits of a buffer live at [base, base + pend); they only move to its front (and the buffer is
    // cleared) when this step's bits would not fit behind them -- every third or fourth step on typical content
    if (ms_base + ms_pend + step_bits + 64u > 32u * (uint32_t)PMS_WORDS) {
      ms_pend = compact(L.ms, ms_base, ms_base + ms_pend, PMS_WORDS);
      ms_base = 0;
    }
    if (v_base + v_pend + step_vbits + 64u > 32u * (uint32_t)PVLC_WORDS) {
      v_pend = compact(L.vlc, v_base, v_base + v_pend, PVLC_WORDS);
      v_base = 0;
    }

    // ---- MEL events of the step (quads with context 0, and in the first row the "both u > 0" event, :664, :763, :883):
    // lane-major, within a lane quad 0, quad 1, then the u event (see the note at mel_at)
    if (ABL & 8) { any_sig |= (uint32_t)(__ballot((vb ^ mp[0] ^ mp[1] ^ sv[0] ^ sv[7] ^ chi[1] ^ incl) == 0x12345u) != 0ull); continue; }
    if (!(ABL & 1)) {
      // the lane's events: bits 0..2 valid (quad 0, quad 1, u), bits 4..6 their values
      const uint32_t fl = ((active && chi[0] == 0u) ? (rho0 != 0u ? 0x11u : 0x01u) : 0u) |
                          ((active && has_q1 && chi[1] == 0u) ? (rho1 != 0u ? 0x22u : 0x02u) : 0u) |
                          ((step == 0 && ev2_valid) ? (ev2_bit != 0u ? 0x44u : 0x04u) : 0u);
      if (__ballot((fl & 7u) != 0u) != 0ull) {                                // (dense content: most steps have no quad with context 0)
        // 64 consecutive events at a time, event e in lane e & 63: the first quad row of the block has three events per
        // lane (its PPR lanes come first), every other row two
        const uint32_t nwords = step == 0 ? 3u : 2u;
        for (uint32_t wd = 0; wd < nwords; ++wd) {
          const uint32_t e = 64u * wd + (uint32_t)lane;
          uint32_t from, slot;
          if (step == 0) {
            const uint32_t e2 = e - 3u * PPR;
            const bool fr = e < 3u * PPR;
            from = fr ? e / 3u : PPR + (e2 >> 1);
            slot = fr ? e - 3u * (e / 3u) : e2 & 1u;
          } else { from = e >> 1; slot = e & 1u; }
          const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((from & 63u) << 2), (int)fl) >> slot;
          const bool valid = from < 64u && (g & 1u) != 0u, one = valid && (g & 16u) != 0u;
          const uint64_t Vw = __ballot(valid), Bw = __ballot(one);
          if (Vw == 0ull) continue;
          const uint32_t nvalid = (uint32_t)__popcll(Vw);
          if (Bw == 0ull) { mel_c += nvalid; continue; }
          // valid events in front of the lane's, in this word
          const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(Vw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Vw, 0u));
          uint32_t info = 0;                                                  // (k, c) at the lane's "1" event
          {
            uint64_t ones = Bw;
            uint32_t consumed = 0;
            while (ones) {
              const uint32_t l = (uint32_t)__builtin_ctzll(ones);
              asm("s_bitset0_b64 %0, %1" : "+s"(ones) : "s"(l));
              const uint32_t bf = rdlane(before, (int)l);
              mel_c += bf - consumed; consumed = bf + 1u;
              info = (uint32_t)lane == l ? (mel_k | (mel_c << 4)) : info;     // (a v_writelane would need the lane number in M0: no shorter)
              // :352-358: the coder steps down, its run starts anew -- k = max(K - 1, 0), c = P[k] (mel_at's pKm1: 0 for K = 0)
              uint32_t K, pKm1;
              if (mel_c >= 85u) mel_c -= (mel_c - 53u) & ~31u;                // (whole runs at k = 12 behind it: rare)
              if (mel_c >= 53u) { K = 12u; pKm1 = 37u; }
              else {
                const uint64_t m = MEL_PM << (63u - mel_c);
                K = (uint32_t)__popcll(m) - 1u;
                const uint64_t m2 = m & ~(0x8000000000000000ull >> (uint32_t)__builtin_clzll(m));
                pKm1 = m2 ? mel_c - (uint32_t)__builtin_clzll(m2) : 0u;
              }
              int km1 = (int)K - 1;
              asm("s_max_i32 %0, %1, 0" : "=s"(km1) : "s"(km1) : "scc");      // (kept scalar: the compiler's saturating subtract is a vector instruction and a read-back)
              mel_k = (uint32_t)km1; mel_c = pKm1;
            }
            mel_c += nvalid - consumed;
          }
          // the lanes' bits: '1' for every run completed since the last "1" event, then '0' and the open run's length in e bits
          uint32_t code = 0, len = 0;
          if (one) {
            uint32_t K, run, nones, pKm1;
            mel_at(info & 15u, info >> 4, K, run, nones, pKm1);
            const uint32_t eb = mel_exp(K);
            code = (((1u << nones) - 1u) << (eb + 1u)) | run;
            len = nones + eb + 1u;
          }
          const uint32_t at_incl = wave_incl_scan(len, lane);
          const uint32_t wbits = rdlane(at_incl, 63);
          if (mel_bits + wbits > MEL_RAW_BITS) mel_err = 1;
          else mel_or(code, len, mel_bits + at_incl - len);
          mel_bits += wbits;
        }
        mel_zeros(0u);
      }
    }

    // ---- MagSgn and VLC bits into the flat, un-stuffed bit buffers ----
    if (ABL & 2) { any_sig |= (uint32_t)(__ballot((vb ^ mp[0] ^ mp[1] ^ sv[0] ^ sv[3] ^ sv[5] ^ sv[7] ^ incl) == 0x12345u) != 0ull); }
    else {
      const uint32_t at = ms_base + ms_pend + (incl & 0xFFFFu) - tot;
      const bool wide_bits = __ballot(max(Uq[0], Uq[1]) > 16u) != 0ull;      // wave-uniform: some sample of the step has more than 16 bits
      if (!wide_bits) {
        // two samples make at most 32 bits: the lane's eight values leave as four words
        uint32_t pr[4], ln[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t m = mp[q];
          const uint32_t m0 = m & 0xFFu, m2 = (m >> 16) & 0xFFu;
          const uint32_t v0 = __builtin_amdgcn_ubfe(sv[4 * q + 0], 0u, m0), v1 = __builtin_amdgcn_ubfe(sv[4 * q + 1], 0u, (m >> 8) & 0xFFu);
          const uint32_t v2 = __builtin_amdgcn_ubfe(sv[4 * q + 2], 0u, m2), v3 = __builtin_amdgcn_ubfe(sv[4 * q + 3], 0u, m >> 24);
          pr[2 * q] = v0 | (v1 << m0); pr[2 * q + 1] = v2 | (v3 << m2);
          ln[q] = (m + (m >> 8)) & 0xFFu;                                     // bits of the quad's first two samples
        }
        if (tot != 0u) {
          or32(L.ms, at, pr[0]);
          or32(L.ms, at + ln[0], pr[1]);
          or32(L.ms, at + tot0, pr[2]);
          or32(L.ms, at + tot0 + ln[1], pr[3]);
        }
      } else {
        uint32_t pos = at;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t m = (mp[i >> 2] >> (8 * (i & 3))) & 0xFFu;
          or_bits(L.ms, pos, __builtin_amdgcn_ubfe(sv[i], 0u, m), m);
          pos += m;
        }
      }
      if (vl != 0u) or32(L.vlc, v_base + v_pend + (incl >> 16) - vl, vb);
    }
    wave_sync();
    // ---- byte stuffing of the windows that are complete ----
    if (ABL & 4) { ms_base = 0; ms_pend = step_bits & 7u; v_base = 0; v_pend = step_vbits & 7u; continue; }
    {
      const uint32_t T = ms_base + ms_pend + step_bits;
      const uint32_t pos = ms_windows_full(ms_base, T);
      ms_base = pos; ms_pend = T - pos;
    }
    {
      const uint32_t T = v_base + v_pend + step_vbits;
      const uint32_t pos = vlc_windows_full(v_base, T);
      v_base = pos; v_pend = T - pos;
    }
  }

  // ---- final flush of both bit buffers: what remains is < 8 bits each ----
  uint32_t ms_carry = 0, v_carry = 0;
  if (!err) {
    uint32_t pos = ms_windows(ms_base, ms_base + ms_pend, true);
    ms_carry = compact(L.ms, pos, ms_base + ms_pend, PMS_WORDS);
    pos = vlc_windows(v_base, v_base + v_pend, true);
    v_carry = compact(L.vlc, pos, v_base + v_pend, PVLC_WORDS);
  }

  err |= mel_err;
  uint32_t total = 0, ms_len = ms_k;
  MelState mel = { 0, 0, 0, 0, 0, 0, 0 };                 // the byte-level state the termination works on (mel_stuff)
  if (!err && any_sig) {
    // ---- ms_terminate (:517-534) ----
    const uint32_t ms_tmp0 = rdfirst(L.ms[0]);
    if (ms_carry) {
      const uint32_t maxb = ms_ff ? 7u : 8u, tt = maxb - ms_carry;
      const uint32_t tmp = ms_tmp0 | ((0xFFu & ((1u << tt) - 1u)) << ms_carry);
      if (tmp != 0xFF) {
        if (lane == 0) outb[ms_len - ms_out] = (uint8_t)tmp;
        ms_len++;
      }
    } else if (ms_ff) ms_len--;
    // ---- terminate_mel_vlc (:412-441) ----
    {
      // the zeros behind the last "1" event: a '1' for every run they completed, and one for the run under way (:412-415)
      uint32_t K, run, nones, pKm1;
      mel_at(mel_k, mel_c, K, run, nones, pKm1);
      mel_ones(nones + (run > 0u ? 1u : 0u));
    }
    err |= mel_err;
    {
      // the raw words move to the (now idle) MagSgn bit buffer, the bytes are written where the words were
      uint32_t* rawc = L.ms + 128;
      const uint32_t total_bits = mel_bits;
      wave_sync();
      if ((uint32_t)lane < MEL_RAW_WORDS) rawc[lane] = (uint32_t)lane < MEL_RAW_WORDS - 1u ? mel_raw[lane] : 0u;
      wave_sync();
      mel_stuff(rawc, err ? 0u : total_bits, L.mel, mel, lane);
      wave_sync();
    }
    const uint32_t need = mel.lastff ? 7u : 8u, remaining = need - mel.nb;
    const uint32_t mel_tmp = (mel.acc << remaining) & 0xFFu;
    const uint32_t mel_mask = (0xFFu << remaining) & 0xFFu;
    const uint32_t vlc_tmp = rdfirst(L.vlc[0]) & 0xFFu;
    const uint32_t vlc_mask = v_carry ? (0xFFu >> (8 - v_carry)) : 0u;
    if ((mel_mask | vlc_mask) != 0) {
      if (mel.pos >= (uint32_t)MEL_CAP) err = 1;
      else {
        const uint32_t fuse = mel_tmp | vlc_tmp;
        if (((((fuse ^ mel_tmp) & mel_mask) | ((fuse ^ vlc_tmp) & vlc_mask)) == 0) && fuse != 0xFF && v_pos > 1) {
          if (lane == 0) L.mel[mel.pos] = (uint8_t)fuse;
          mel.pos++;
        } else {
          if (v_pos >= (uint32_t)VLC_CAP) err = 1;
          else {
            if (lane == 0) { L.mel[mel.pos] = (uint8_t)mel_tmp; outb[OUT_CAP - 1 - v_pos] = (uint8_t)vlc_tmp; }
            mel.pos++; v_pos++;
          }
        }
      }
    }
    err |= mel.err;
    total = ms_len + mel.pos + v_pos;
  }
  wave_sync();
  __threadfence_block();

  // ---- claim a 4-byte aligned slot in the compacted output: MagSgn | MEL | VLC (:1003-1014) ----
  uint32_t off = 0;
  if (err) total = 0;
  if (total) {
    off = claim_output(cursor, regions, nreg, bi, (total + 3u) & ~3u, out_cap, lane);
    if (off == 0xFFFFFFFFu) { err = 1; total = 0; off = 0; }
  }
  if (total) {
    const uint32_t scup = mel.pos + v_pos;
    const uint32_t tail = mel.pos + v_pos;
    const uint32_t gpart = min(ms_out, ms_len), local = ms_len - gpart;   // MagSgn bytes in the scratch slot / still in the stage
    uint8_t* dst = out + off;
    if (gpart) {                                          // (ms_out is a multiple of 4; ms_len < ms_out only when a final 0xFF was dropped)
      const uint32_t nwd = gpart >> 2;
      const uint32_t* s32 = reinterpret_cast<const uint32_t*>(ms_spill);
      uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
      for (uint32_t i = lane; i < nwd; i += 64) d32[i] = s32[i];
      if ((uint32_t)lane < (gpart & 3u)) dst[4u * nwd + lane] = ms_spill[4u * nwd + lane];
    }
    if ((gpart & 3u) == 0 && local + tail <= OUT_CAP) {
      // close the gap: MEL and VLC bytes move down behind the MagSgn bytes of the stage (destination <= source)
      for (uint32_t base = 0; base < tail; base += 64) {
        const uint32_t i = base + lane;
        uint32_t b = 0;
        if (i < mel.pos) b = L.mel[i];
        else if (i < tail) b = outb[OUT_CAP - v_pos + (i - mel.pos)];
        wave_sync();
        if (i < tail) {
          if (i == tail - 1) b = scup >> 4;
          else if (i == tail - 2) b = (b & 0xF0u) | (scup & 0xFu);
          outb[local + i] = (uint8_t)b;
        }
        wave_sync();
      }
      uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + gpart);
      const uint32_t nw = (local + tail + 3u) >> 2;
      for (uint32_t i = lane; i < nw; i += 64) d32[i] = L.out[i];
    } else {
      for (uint32_t i = gpart + lane; i < total; i += 64) {
        uint32_t b;
        if (i < ms_len) b = outb[i - ms_out];
        else if (i < ms_len + mel.pos) b = L.mel[i - ms_len];
        else b = outb[OUT_CAP - v_pos + (i - ms_len - mel.pos)];
        if (i == total - 1) b = scup >> 4;
        else if (i == total - 2) b = (b & 0xF0u) | (scup & 0xFu);
        dst[i] = (uint8_t)b;
      }
    }
  }
  if (lane == 0) {
    results[bi].offset = off; results[bi].length = total;
    if (err) atomicOr(status, 1u);
  }
}

}  // namespace

namespace ojphgpu {
// `widths`: bit 0 = the range holds blocks up to 64 samples wide, bit 1 = it holds wider ones; bit 2 = it holds
// blocks of reversibly transformed components, bit 3 = of irreversibly transformed ones; bit 4 = none of its blocks of
// up to 64 samples is wider than 32 (they take the 8-pairs-by-8-rows layout); bit 5 = it holds blocks of 64-bit samples
// (cb_desc.reversible bit 2); bit 6 = none of its wider blocks is wider than 128 samples (they take the narrow kernel's
// 32-pairs-by-2-rows layout instead of the raster-order wide kernel).  Every kernel skips the blocks of the other kind, so a caller that does not know passes
// 3 | 32 (wavelet bits clear = both).
int ht_encode_launch(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n, const void* d_coef, uint8_t* d_scratch,
                     uint8_t* d_out, uint32_t out_cap, ojphgpu_cb_result* d_results, uint32_t* d_cursor, uint32_t* d_status,
                     int widths, const uint32_t* d_regions, uint32_t nreg)
{
  if (n == 0) return OJPHGPU_OK;
  if (ensure_tables() != 0) return OJPHGPU_E_HIP;
  if (!d_blocks || !d_coef || !d_scratch || !d_out || !d_results || !d_cursor || !d_status) return OJPHGPU_E_INVALID;
  if (nreg && (!d_regions || (nreg & (nreg - 1u)))) return OJPHGPU_E_INVALID;
  dim3 grid((n + WAVES - 1) / WAVES);
  if ((widths & 12) == 0) widths |= 12;                   // the caller does not know the wavelets: both instantiations
  // timing experiment: dynamic LDS nobody uses lowers the workgroups per CU (OJPHGPU_ENC_LDS_BALLAST bytes)
  static const unsigned ballast = [] { const char* e = getenv("OJPHGPU_ENC_LDS_BALLAST"); const long v = e ? atol(e) : 0; return v > 0 && v < 100000 ? (unsigned)v : 0u; }();
  const dim3 ngrid((n + NWAVES - 1) / NWAVES);
  if ((widths & 1) && (widths & 4) && (widths & 16))
    hipLaunchKernelGGL((ht_encode_kernel<true, 3>), ngrid, dim3(64 * NWAVES), ballast, (hipStream_t)stream, d_blocks, n,
                       (const uint32_t*)d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, d_regions, nreg);
  else if ((widths & 1) && (widths & 4))
    hipLaunchKernelGGL((ht_encode_kernel<true, 4>), ngrid, dim3(64 * NWAVES), ballast, (hipStream_t)stream, d_blocks, n,
                       (const uint32_t*)d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, d_regions, nreg);
  if ((widths & 1) && (widths & 8) && (widths & 16))
    hipLaunchKernelGGL((ht_encode_kernel<false, 3>), ngrid, dim3(64 * NWAVES), ballast, (hipStream_t)stream, d_blocks, n,
                       (const uint32_t*)d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, d_regions, nreg);
  else if ((widths & 1) && (widths & 8))
    hipLaunchKernelGGL((ht_encode_kernel<false, 4>), ngrid, dim3(64 * NWAVES), ballast, (hipStream_t)stream, d_blocks, n,
                       (const uint32_t*)d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, d_regions, nreg);
  if ((widths & 2) && (widths & 64)) {                      // wider blocks, none of them wider than 128 samples: 32 pairs by 2 quad rows
    if (widths & 4)
      hipLaunchKernelGGL((ht_encode_kernel<true, 5>), ngrid, dim3(64 * NWAVES), ballast, (hipStream_t)stream, d_blocks, n,
                         (const uint32_t*)d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, d_regions, nreg);
    if (widths & 8)
      hipLaunchKernelGGL((ht_encode_kernel<false, 5>), ngrid, dim3(64 * NWAVES), ballast, (hipStream_t)stream, d_blocks, n,
                         (const uint32_t*)d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, d_regions, nreg);
  } else if (widths & 2)
    hipLaunchKernelGGL(ht_encode_wide_kernel<false>, grid, dim3(64 * WAVES), 0, (hipStream_t)stream, d_blocks, n,
                       (const uint32_t*)d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, d_regions, nreg);
  if (widths & 32)                                          // blocks of components on the 64-bit sample path
    hipLaunchKernelGGL(ht_encode_wide_kernel<true>, grid, dim3(64 * WAVES), 0, (hipStream_t)stream, d_blocks, n,
                       (const uint32_t*)d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, d_regions, nreg);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}
}  // namespace ojphgpu

extern "C" int ojphgpu_ht_encode(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                                  const void* d_coef, uint8_t* d_scratch, uint8_t* d_out, uint32_t out_cap,
                                  ojphgpu_cb_result* d_results, uint32_t* d_cursor, uint32_t* d_status)
{
  return ojphgpu::ht_encode_launch(stream, d_blocks, n, d_coef, d_scratch, d_out, out_cap, d_results, d_cursor, d_status, 3 | 32, nullptr, 0);
}

namespace ojphgpu {
int upload_enc_tables(const HtTables& t)
{
  return hipMemcpyToSymbol(HIP_SYMBOL(g_enc_vlc), t.enc_vlc, sizeof(t.enc_vlc)) == hipSuccess ? 0 : -1;
}
}
