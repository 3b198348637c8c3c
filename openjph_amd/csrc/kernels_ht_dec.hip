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
// Round 3: ONE launch per frame in the common case -- ht_dec_fused_kernel, chain workgroups (step 1, reading the raw
// cleanup segments through partner wavefronts: no prep launch, no flat strings in memory) followed by persistent step-2
// worker wavefronts that take their blocks 8 quad rows at a time as the chains publish progress; see the comment above
// that kernel.  The stages below are its parts, and remain launches of their own for blocks wider than 64 samples, mixed
// wavelets, refinement passes (OJPHGPU_DEC_FUSED=0 / OJPHGPU_DEC_PREP=1 force the older forms):
//   prep    (ht_dec_prep_kernel)   ONE WAVEFRONT PER CODE-BLOCK.  Byte un-stuffing only looks at the
//           previous raw byte, so the bit offset of every byte is a wavefront prefix sum (the idea
//           of the reference's AVX2 decoder, ojph_block_decoder_avx2.cpp:277-386).  The backward
//           VLC segment and the forward MEL segment are rewritten as flat, un-stuffed bit strings
//           in an HBM scratch area, padded the way the reference pads (VLC: zeros, MEL: ones).
//   step 1  (ht_dec_step1_kernel)  The MEL / VLC / U-VLC stage is a serial state machine: where a
//           codeword starts depends on every earlier codeword and on the neighbour context.  ONE
//           LANE owns one code-block's chain and a wavefront advances 64 independent chains in
//           lock step.  Such a wavefront is alone on its SIMD and issues one instruction every ~6 cycles
//           whatever it waits for: the launch time is instructions per quad pair x pairs x 6 cycles
//           (measured: no change without the VLC word loads; slower when the LDS round trips were
//           hidden by software pipelining at the price of 20 more instructions; slower with 32 or 16
//           blocks per wavefront).  So everything that can leave the chain wavefront has left it:
//           bits come from the flat strings (VLC: two words + one prefetched word in registers, one
//           v_alignbit per look, no branch; no un-stuffing); the adaptive MEL run-length code is
//           decoded by a PARTNER WAVEFRONT of the same workgroup (another SIMD; most are idle during
//           this launch) into an event string in LDS that the chain reads one bit per event; the
//           U-VLC of rows after the first is decoded by arithmetic (ht_uvlc.h) instead of a table in
//           LDS; the significance of the quad row above lives in two 64-bit masks, the VLC tables in
//           LDS.  Output: one 32-bit record per quad {t-word, u_q}.  Earlier attempts to take the MEL
//           decoder out of the chain, dropped: (a) run by the prep kernel, one bit per MEL event for
//           step 1: step 1 0.21 -> 0.18 ms, but the decoder is scalar code there (one wavefront per
//           block), 300 M scalar instructions per 8K frame through one scalar unit per CU: prep 0.07 ->
//           1.0 ms; (b) a launch of its own, one lane per block beside prep: 0.24 ms for that launch
//           (byte-wise un-stuffing and the symbol loop diverge between the 64 blocks of a wavefront).
//   step 2  (ht_dec_step2_kernel)  ONE WAVEFRONT PER CODE-BLOCK, ONE LANE PER SAMPLE COLUMN.  A lane
//           decodes the two samples of its column in the current quad row -- they are adjacent in
//           the MagSgn bit string, so a wavefront prefix sum of the lanes' bit counts gives every
//           lane its offset -- and keeps the exponent of its bottom sample for the next row's
//           kappa, which neighbours fetch with DPP moves (no LDS, no barrier).  Each row of 64
//           de-quantised samples leaves as one coalesced 256-byte store.  MagSgn bytes are
//           un-stuffed 256 at a time into a small LDS ring, so LDS use does not depend on the
//           size of the code-block and the CU stays fully occupied.
// SigProp / MagRef passes (:1318-1609), which only foreign codestreams carry (the reference's own
// encoder never emits them, ojph_block_encoder.cpp:548), run in a fourth launch, ht_dec_refine_kernel,
// that the codec objects skip when no block of the frame has more than one pass.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <stdint.h>
#include <type_traits>
#include "../../include/ojphgpu.h"
#include "ht_tables.h"
#include "ht_uvlc.h"

namespace ojphgpu {
__device__ uint16_t g_dec_vlc[2][1024];
__device__ uint32_t g_dec_vlc32[2][1024];    // the same + the 9 bits step 2 wants, for the fused launch's 16-bit records (ht_tables.h)
__device__ uint16_t g_dec_uvlc0[320];
}

namespace {

constexpr int WAVES = 4;
constexpr uint32_t RING_WORDS = 256;          // step 2: 8 Kbit of un-stuffed MagSgn per wavefront (>= 4160 + 2048 + 32 in flight)
constexpr uint32_t RING_MASK = RING_WORDS - 1;
// The ring is followed by COPIES of its first two words (every writer of word 0 / 1 writes word 256 / 257 as well): a row
// reads three consecutive words from wi & RING_MASK on without wrapping each index (three address computations less per row)
constexpr uint32_t RING_ALLOC = RING_WORDS + 2;
constexpr uint32_t ROW_BITS_MAX = 64 * 2 * 32; // a quad row of 64 columns consumes at most this many bits
constexpr uint32_t EXP_BYTES = 1024 + 8;      // wide blocks (> 64 columns): exponent row in LDS

// words of the flat VLC / MEL strings of a cleanup segment with MEL+VLC length scup (incl. 2 pad words)
__host__ __device__ __forceinline__ uint32_t vlc_words(uint32_t scup) { return ((scup - 2u) * 8u + 4u + 8u + 31u) / 32u + 2u; }
__host__ __device__ __forceinline__ uint32_t mel_words(uint32_t scup) { return ((scup - 1u) * 8u + 31u) / 32u + 2u; }
__host__ __device__ __forceinline__ uint32_t aux_words(uint32_t len1)
{
  const uint32_t s = len1 < 2u ? 2u : (len1 > 4079u ? 4079u : len1);
  return vlc_words(s) + mel_words(s);
}

__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t rdfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }   // a wave-uniform value into a scalar register

// wave64 inclusive prefix sum with DPP adds (row shifts inside 16-lane rows, then row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
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
// value of lane-1 / lane+1 (0 at the ends); quad_perm swap of lanes 2k <-> 2k+1
// (bound_ctrl: a lane without a source reads 0 -- no register has to be cleared for the "old" value first, and a max / add
// of the result folds into the DPP instruction)
__device__ __forceinline__ uint32_t from_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t from_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t from_pair(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }  // quad_perm:[1,0,3,2]

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
  // (32-bit path: block_decoder32.cpp:768-789; the 64-bit function has no such test (:792-827): its p = 62 - missing_msbs
  // must stay >= 1 for the cleanup pass, >= 2 when refinement passes follow)
  const uint32_t mm_lim = (d.reversible & 4u) ? ((d.num_passes > 1 && d.len2 > 0) ? 61u : 62u) : 30u;
  if (d.num_passes > 3 || d.missing_msbs >= mm_lim || d.len1 < 2) return 0;
  const uint32_t lcup = d.len1;
  const uint32_t scup = ((uint32_t)cb[lcup - 1] << 4) + (cb[lcup - 2] & 0xFu);
  if (scup < 2 || scup > lcup || scup > 4079) return 0;
  return scup;
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p)
{
  uint32_t v; __builtin_memcpy(&v, p, 4); return v;
}

__device__ __forceinline__ void or_bits(uint32_t* buf, uint32_t pos, uint32_t v, uint32_t n)
{
  if (n == 0) return;
  const uint32_t w = pos >> 5, sh = pos & 31;
  atomicOr(&buf[w], v << sh);
  if (sh + n > 32) atomicOr(&buf[w + 1], v >> (32 - sh));
}

// -------------------------------------------------------------------------------------------------
// prep: one wavefront = one code-block; VLC (backward) and MEL (forward) segments -> flat bit strings
// -------------------------------------------------------------------------------------------------
// One stream: `count` bytes, byte k given by get(k) (with its predecessor for the stuffing rule);
// bits(b, prev) = number of payload bits of byte b; REV = string is consumed MSB first (MEL).
// The reference's byte readers OR every raw byte into their window with all 8 bits and only then
// advance by 7 or 8 (rev_read :308-359, frwd_read :609-655): after a stuffing event the 7-bit
// byte's MSB lands on the LSB of the byte that follows.  In a conforming stream that MSB is 0; to
// stay bit-identical on arbitrary bytes the un-stuffers here work on "effective" bytes
//   eff[k] = raw[k] | (byte k-1 carried only 7 bits ? raw[k-1] >> 7 : 0)
// which is a local rule and keeps the whole thing a prefix sum.  (For MEL, MSB first, the stray
// bit lands on a bit that is already 1, mel_read :126-148, so nothing changes.)
template <bool MEL>
__device__ void flatten(const uint8_t* __restrict__ cb, uint32_t lcup, uint32_t scup, uint32_t* __restrict__ out,
                        uint32_t* lds, int lane)
{
  // VLC: one zero fill byte is appended so that a stray bit of the last byte is kept (:313-331)
  const uint32_t real = MEL ? scup - 1u : scup - 2u;
  const uint32_t count = MEL ? real : real + 1u;
  for (int i = lane; i < 68; i += 64) lds[i] = 0;
  wave_sync();
  uint32_t cursor = 0, wpos = 0;
  const uint32_t d0 = cb[lcup - 2];
  if (!MEL) {                                              // rev_init (block_decoder32.cpp:380-400)
    const uint32_t t = d0 >> 4;
    if (lane == 0) lds[0] = t;
    cursor = 4u - ((t & 7u) == 7u ? 1u : 0u);
    wave_sync();
  }
  // VLC raw byte k (k = -1: the byte holding the 4 initial bits, with its low nibble forced to ones)
  auto vraw = [&](int k) -> uint32_t { return k < 0 ? (d0 | 0xFu) : ((uint32_t)k < real ? (uint32_t)cb[lcup - 3 - k] : 0u); };
  for (uint32_t base = 0; base < count; base += 256) {
    const uint32_t k0 = base + 4u * (uint32_t)lane;
    uint32_t val = 0, nb = 0;
    if (k0 < count) {
      if (MEL) {
        uint32_t prev = k0 ? cb[lcup - scup + k0 - 1] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          const uint32_t k = k0 + j;
          if (k < count) {                                 // mel_read (:93-157): after 0xFF only 7 bits, MSB first
            uint32_t b = cb[lcup - scup + k];
            if (k == count - 1) b |= 0xFu;                 // the last MEL byte shares its low nibble with VLC (:116)
            const uint32_t n = 8u - (prev == 0xFFu ? 1u : 0u);
            val |= (__brev(b & ((1u << n) - 1u)) >> (32u - n)) << nb;      // stream order = MSB first
            nb += n; prev = b;
          }
        }
      } else {
        // state of the byte before this lane's first one: was it a 7-bit byte, and its raw value
        uint32_t p1 = vraw((int)k0 - 1), p2 = k0 >= 1 ? vraw((int)k0 - 2) : 0u;
        bool p1_short = k0 >= 1 && p2 > 0x8Fu && (p1 & 0x7Fu) == 0x7Fu;    // rev_read: > 0x8F then x1111111 -> 7 bits
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          const uint32_t k = k0 + j;
          if (k < count) {
            const uint32_t b = vraw((int)k);
            const bool is_short = p1 > 0x8Fu && (b & 0x7Fu) == 0x7Fu;
            const uint32_t n = is_short ? 7u : 8u;
            const uint32_t eff = b | (p1_short ? p1 >> 7 : 0u);
            val |= (eff & ((1u << n) - 1u)) << nb;
            nb += n; p1 = b; p1_short = is_short;
          }
        }
      }
    }
    const uint32_t incl = wave_incl_scan(nb);
    or_bits(lds, cursor + incl - nb, val, nb);
    const uint32_t T = cursor + rdlane(incl, 63);
    wave_sync();
    const uint32_t nfull = T >> 5;
    for (uint32_t i = lane; i < nfull; i += 64) out[wpos + i] = MEL ? __brev(lds[i]) : lds[i];
    const uint32_t carry = lds[nfull];
    wave_sync();
    for (int i = lane; i < 68; i += 64) lds[i] = 0;
    wave_sync();
    if (lane == 0) lds[0] = carry;
    wave_sync();
    wpos += nfull; cursor = T & 31u;
  }
  if (lane == 0 && cursor) {
    uint32_t w = lds[0];
    if (MEL) w = __brev(w | (0xFFFFFFFFu << cursor));      // past the end the MEL segment continues with 1s (:98)
    out[wpos] = w;
  }
  if (cursor) wpos++;
  // ... and the VLC segment with 0s (:313-331): up to the nominal length of the string (un-stuffing made it shorter; a
  // reader that runs past the segment -- corrupt streams do -- must find the fill value, not what the scratch held before)
  const uint32_t nominal = MEL ? mel_words(scup) : vlc_words(scup);
  for (uint32_t i = wpos + (uint32_t)lane; i < nominal; i += 64) out[i] = MEL ? 0xFFFFFFFFu : 0u;
  wave_sync();
}

__global__ __launch_bounds__(64 * WAVES) void ht_dec_prep_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data, uint32_t* __restrict__ aux)
{
  __shared__ uint32_t s_buf[WAVES][68];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and the compiler knows it
  const uint32_t bi = blockIdx.x * WAVES + wave;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (d.w == 0 || d.h == 0 || d.len1 == 0 || d.num_passes == 0 || (d.reversible & 4u)) return;   // (64-bit sample path: ht_dec64_prep_kernel)
  const uint8_t* cb = data + d.data_off;
  const uint32_t scup = check_block(d, cb);
  if (scup == 0) return;
  uint32_t* out = aux + d.reserved;
  flatten<false>(cb, d.len1, scup, out, s_buf[wave], lane);
  flatten<true>(cb, d.len1, scup, out + vlc_words(scup), s_buf[wave], lane);
}

// -------------------------------------------------------------------------------------------------
// step 1: one lane = one code-block
// -------------------------------------------------------------------------------------------------
// Both readers take their bits from the flat strings of the prep kernel.  The word that will be
// appended next is requested unconditionally once per quad pair -- outside any divergent branch, so
// that the wait for it lands at its use one pair later and never on the chain.
// Reader of the flat LSB-first VLC string: two consecutive words in registers, a bit position inside the lower one,
// one more word in flight.  peek() is ONE v_alignbit_b32; advance(k) crosses at most one word boundary (k <= 32),
// where the prefetched word moves in -- selects, no branch.  prefetch() is issued once per quad pair, after
// advance(): the word it requests is only looked at one pair later.
// agent-scope accesses: what chains, workers of other slices and other XCDs (each with an L2 of its own) exchange inside
// ONE launch of the fused kernel goes through memory, not through a cache that only its writer's XCD sees
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct FlatRd {
  const uint32_t* w; uint32_t idx, last, lo, hi, pre, bp, lane = 0;
  __device__ __forceinline__ void init(const uint32_t* p, uint32_t nwords) {
    w = p; last = nwords - 1u;
    lo = p[0]; hi = p[last < 1u ? last : 1u]; pre = p[last < 2u ? last : 2u];
    idx = 2; bp = 0;
  }
  __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(hi, lo, bp); }   // 32 bits from the current position
  __device__ __forceinline__ void advance(uint32_t k) {
    bp += k;
    const bool cross = bp >= 32u;
    lo = cross ? hi : lo; hi = cross ? pre : hi;
    bp = cross ? bp - 32u : bp; idx += cross ? 1u : 0u;
  }
  __device__ __forceinline__ void prefetch() { pre = w[idx < last ? idx : last]; }
  __device__ __forceinline__ void settle() {}
};

// MEL: the adaptive run-length code is decoded by a PARTNER WAVEFRONT (the second wavefront of the workgroup, on
// another SIMD of the CU; most SIMDs are idle during this launch) into an event string in LDS -- bit i of the string
// = the i-th MEL event of the lane's code-block -- so that the chain wavefront only reads event bits.  The events do
// not depend on the VLC stream (only how many of them get consumed does), so the producer runs ahead freely; it
// publishes how many 32-event words are complete and the chain polls that count in the rare case it catches up.
typedef __attribute__((address_space(3))) uint32_t lds_u32;   // pointers kept in structs must not decay to flat ones:
                                                               // a flat load counts as a global one and drags vmcnt waits in
// A chain that waits for its partner waits for a wavefront of its own workgroup: resident by construction, so the wait ends;
// the bound (about a second of s_sleep(2) polls) only keeps a broken build from hanging the queue.
constexpr uint32_t LDS_WAIT_SPINS = 1u << 23;
constexpr uint32_t EV_WORDS = 44;          // <= 1024 quads + 256 initial-row pairs events = 40 words, + 2 of read-ahead
// The event string lives in a RING of EV_RING words per lane: the producer never writes word cons + EV_AHEAD or beyond, the
// chain never reads below word cons (EvRd: the word it may still fetch again is word idx = cons), so a word is overwritten
// only once it is EV_RING - EV_AHEAD words behind what the chain reads.  (The whole string, 44 words per lane, was 45 KB of
// the fused launch's workgroup: with the rows of records staged in LDS as well, only one workgroup fitted a CU.)
constexpr uint32_t EV_RING = 8;
__device__ __forceinline__ uint32_t ev_words_of(uint32_t QW, uint32_t QH)
{
  const uint32_t w = (QW * QH + ((QW + 1u) >> 1) + 31u) / 32u + 2u;
  return w < EV_WORDS ? w : EV_WORDS;
}

// T.814 decodeMELSym, run by run (same runs as block_decoder32.cpp:170-269), MSB-first flat string of the prep kernel.
// Demand driven: a lane stays at most EV_AHEAD words in front of what its chain has taken (s_cons), so that blocks
// which hardly use the MEL stream (dense ones) do not pay for decoding all of it, and stops when the chain is done.
constexpr uint32_t EV_AHEAD = 4;
// one run of the adaptive MEL code (T.814 decodeMELSym, block_decoder32.cpp:170-269) from the MSB-first window, without a
// branch: either 2^e "0" events (k up), or `run` < 2^e "0" events and a "1" (k down)
__device__ __forceinline__ void mel_run(uint64_t& win, uint32_t& n, uint32_t& k, uint64_t& ev, uint32_t& nev)
{
  const uint32_t e = mel_exp(k);
  const uint32_t top = (uint32_t)(win >> 32);
  const bool zr = (top >> 31) != 0u;
  const uint32_t run = (top >> (31u - e)) & ((1u << e) - 1u);
  const uint32_t adv = zr ? 1u : e + 1u;
  ev |= zr ? 0ull : 1ull << (nev + run);
  nev += zr ? 1u << e : run + 1u;
  k = zr ? (k < 12u ? k + 1u : 12u) : (k > 0u ? k - 1u : 0u);
  win <<= adv; n -= adv;
}

__device__ __forceinline__ void mel_producer(const uint32_t* __restrict__ w, uint32_t nwords, uint32_t out_words,
                                             lds_u32* s_ev, volatile lds_u32* s_prog, const volatile lds_u32* s_cons,
                                             const volatile lds_u32* s_done, uint32_t lane)
{
  const uint32_t last = nwords - 1u;
  uint64_t win = ((uint64_t)w[0] << 32) | (uint64_t)w[last < 1u ? last : 1u];
  uint32_t n = 64, idx = 2, k = 0, nev = 0, wr = 0;
  uint32_t pre = w[last < 2u ? last : 2u];
  uint64_t ev = 0;
  while (wr < out_words && s_done[lane] == 0u) {
    const bool go = wr < s_cons[lane] + EV_AHEAD;
    if (go) {
      if (n <= 32u) { win |= (uint64_t)pre << (32u - n); n += 32u; ++idx; pre = w[idx < last ? idx : last]; }
      while (nev <= 31u && n >= 6u) mel_run(win, n, k, ev, nev);
      if (nev >= 32u) {
        s_ev[(wr & (EV_RING - 1u)) * 64u + lane] = (uint32_t)ev; ev >>= 32; nev -= 32u; ++wr;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     // LDS only: no wait on global memory
        s_prog[lane] = wr;
      }
    }
    if (__ballot(go) == 0ull) __builtin_amdgcn_s_sleep(16);     // every lane far enough ahead: leave the issue slots alone
  }
}

// chain side: 64 events in two registers + one word in flight, as FlatRd (LSB = next event)
struct EvRd {
  const lds_u32* ev; volatile lds_u32* prog; volatile lds_u32* cons; uint32_t idx, last, lo, hi, pre, bp, avail, lane; bool stuck;
  __device__ __forceinline__ uint32_t fetch(uint32_t want) {
    if (want >= avail && !stuck) {                           // rare: the producer is not that far yet
      uint32_t spins = 0;
      do { __builtin_amdgcn_s_sleep(2); avail = prog[lane]; } while (want >= avail && ++spins < LDS_WAIT_SPINS);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      stuck = stuck || want >= avail;                        // cannot happen (the partner always progresses); never hang
    }
    return ev[(want & (EV_RING - 1u)) * 64u + lane];
  }
  __device__ __forceinline__ void init(const lds_u32* e, volatile lds_u32* p, volatile lds_u32* c, uint32_t nwords, uint32_t l) {
    ev = e; prog = p; cons = c; last = nwords - 1u; lane = l; avail = 0; stuck = false;
    lo = fetch(0); hi = fetch(last < 1u ? last : 1u); pre = fetch(last < 2u ? last : 2u);
    idx = 2; bp = 0;
  }
  __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(hi, lo, bp); }
  __device__ __forceinline__ void advance(uint32_t k) {
    bp += k;
    const bool cross = bp >= 32u;
    lo = cross ? hi : lo; hi = cross ? pre : hi;
    bp = cross ? bp - 32u : bp; idx += cross ? 1u : 0u;
  }
  __device__ __forceinline__ void prefetch() { cons[lane] = idx; pre = fetch(idx < last ? idx : last); }   // idx: words taken so far
};

// Both records of a quad pair leave as ONE 8-byte store, also when the second quad does not exist
// (odd QW): the extra word lands on the first record of the next row, which is written later, or on
// the pad element every block's record area ends with.  A fixed number of stores per pair keeps the
// compiler's vmcnt bookkeeping from ever waiting on a store when it needs a prefetched word.
struct __attribute__((aligned(4))) U2 { uint32_t x, y; };
__device__ __forceinline__ void store_pair(uint32_t* p, uint32_t a, uint32_t b)
{
  U2 v; v.x = a; v.y = b;
  *reinterpret_cast<U2*>(p) = v;
}

constexpr uint32_t REC_STRIDE = 128;      // elements between consecutive quad pairs of one block (64 blocks x 2 records)

// The window moves of advance() must be done BEFORE the next word is requested (otherwise the compiler sinks them
// below the poll branch of the event reader, keeps the old and the new prefetched word in two registers and copies
// one into the other at the end of the step -- with a wait for the load that was only just issued)
#define PIN_WINDOW(r) asm volatile("" : "+v"((r).lo), "+v"((r).hi))

// The quad rows of one code-block (one lane).  NARROW: QW <= 32 for every lane of the wavefront.
// FUSED: the records are stored with agent scope and after every S2_ROWS quad rows the wavefront says so in `flag`
// ((epoch << 16) | rows done), once its stores have completed: step-2 workers of the same launch are waiting for them.
#ifndef S2_ROWS_N
#define S2_ROWS_N 8
#endif
constexpr uint32_t S2_ROWS = S2_ROWS_N, S2_MAX_PER_WAVE = 8;
// The fused launch's slices of quad rows, the same for every block of a launch whose tallest block has `max_qh` quad rows:
// S2_ROWS rows each -- but the LAST S2_ROWS-row slice is cut at max_qh - 4 and max_qh - 2: when the chains end, the workers
// have the last slice still to decode, and how long the launch goes on after its chains is that slice's length (measured,
// 8K frame: chains end at 0.245 ms, the launch with a last slice of 8 rows at 0.32 ms).  Slice `sl` = quad rows [lo, hi).
struct SliceSched {
  uint32_t tail, b1, b2, max_qh, n;        // the last full-size boundary; the two cuts (== tail / b1 when they fall away); slices in all
  uint32_t h1, h2;                         // cuts of the FIRST full-size slice (0: none): [0, h1) [h1, h2) [h2, S2_ROWS)
  __host__ __device__ explicit SliceSched(uint32_t mq) {
    max_qh = mq ? mq : 1u;
    tail = ((max_qh - 1u) / S2_ROWS) * S2_ROWS;
#ifndef S2_TAIL
#define S2_TAIL 2                 // the last slice's cuts: 2 = 4 + 2 + 2 rows, 1 = 4 + 4, 0 = none (A/B switch)
#endif
    b1 = (S2_TAIL >= 1 && max_qh > tail + 4u) ? max_qh - 4u : tail;
    b2 = (S2_TAIL >= 2 && max_qh > b1 + 2u) ? max_qh - 2u : b1;
#ifndef S2_HEAD
#define S2_HEAD 0                 // the first slice's cuts: 0 = none, 1 = 4 + 4 rows, 2 = 2 + 2 + 4, 3 = 2 + 6 (A/B switch)
#endif
    // (only where a second full-size slice follows: the workers start on the first rows while the chains are still far from done)
    const bool head = S2_HEAD != 0 && tail >= S2_ROWS && S2_ROWS == 8u;
    h1 = !head ? 0u : S2_HEAD == 1 ? 4u : 2u;
    h2 = !head ? 0u : S2_HEAD == 2 ? 4u : 0u;
    n = tail / S2_ROWS + 1u + (b1 > tail ? 1u : 0u) + (b2 > b1 ? 1u : 0u) + (h1 ? 1u : 0u) + (h2 ? 1u : 0u);
  }
  __host__ __device__ void bounds(uint32_t sl, uint32_t& lo, uint32_t& hi) const {
    const uint32_t extra = (h1 ? 1u : 0u) + (h2 ? 1u : 0u);       // pieces the first slice is cut into, beyond one
    if (sl <= extra) {                                            // [0, h1) [h1, h2) [h2 or h1 or 0, first boundary)
      const uint32_t first_end = tail >= S2_ROWS ? S2_ROWS : 0u;  // (0: no full-size slice at all -- then extra == 0 and the code below serves sl 0)
      if (extra && first_end) {
        const uint32_t c0 = 0u, c1 = h1, c2 = h2 ? h2 : first_end, c3 = first_end;
        if (sl == 0u) { lo = c0; hi = c1; return; }
        if (sl == 1u) { lo = c1; hi = c2; return; }
        lo = c2; hi = c3; return;
      }
    }
    sl -= extra;
    const uint32_t full = tail / S2_ROWS;
    if (sl < full) { lo = sl * S2_ROWS; hi = lo + S2_ROWS; return; }
    uint32_t j = sl - full;                // pieces of [tail, max_qh): [tail, b1) [b1, b2) [b2, max_qh), the empty ones left out
    lo = tail;
    if (b1 > tail) { if (j == 0u) { hi = b1; return; } --j; lo = b1; }
    if (b2 > b1) { if (j == 0u) { hi = b2; return; } --j; lo = b2; }
    hi = max_qh;
  }
  __host__ __device__ bool publishes_after(uint32_t rows) const {   // rows = quad rows complete; (the last rows are published by the wavefront's end)
    return rows < max_qh && (rows % S2_ROWS == 0u || (rows == b1 && b1 > tail) || (rows == b2 && b2 > b1) || (h1 && rows == h1) || (h2 && rows == h2));
  }
};
constexpr int S2_RINGS = 5;                   // fused launch: worker wavefronts with at most this many blocks keep a ring per block
// The fused launch's records.  The separate launches keep a 32-bit record per quad, pair-major over the 64 blocks of a
// chain wavefront (a chain store = 512 contiguous bytes) -- but a step-2 wavefront then reads 16 pieces of 8 bytes, 512
// bytes apart, per quad row, and an agent-scope load fetches a whole sector for each: the launch read SIX times the
// records it used (PMC, round 3: 0.60 GB of its 0.67 GB of reads).  The fused launch therefore keeps
//   * 16 bits per quad: step 2 needs 9 bits of the table entry (dec_vlc32: 2 bits per sample + gamma) and u < 64,
//     record = packed9 | u << 9;
//   * in 16-byte pieces -- a QUARTER of a quad row (8 quads) of one block -- laid out [row][quarter][block of the 64]:
//     the chain (a lane per block) collects a row's pair words in LDS ([pair][lane]: conflict free both ways) and at the
//     end of the row stores them with four 16-byte agent-scope stores, each 1 KB contiguous over the wavefront (a quarter
//     of the store instructions, every line written whole); a worker's load of a row touches 4 pieces instead of 16.
// (Block-major -- a block's row as 64 contiguous bytes, a worker's row ONE piece -- was built as well: the workers alone
// are as fast, 0.200 ms against 0.232 with the 32-bit records, but every lane of the chain then writes 16 bytes into a line
// of its own, and with the workers' stores beside them those partial writes cost the launch 0.04 ms, 0.38 against 0.34.)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t REC16_ROW_WORDS = 16;                                  // 32 quads x 16 bits
__device__ __forceinline__ void flush_row16(const lds_u32* s_rec, uint32_t lane, uint32_t* dst)
{
  u32x4 q[4];
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    q[j].x = s_rec[(4 * j + 0) * 64 + lane]; q[j].y = s_rec[(4 * j + 1) * 64 + lane];
    q[j].z = s_rec[(4 * j + 2) * 64 + lane]; q[j].w = s_rec[(4 * j + 3) * 64 + lane];
  }
  // (one asm: the sixteen words are read with one wait; agent scope = write through, like the atomic stores)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc1\n\t"
               "global_store_dwordx4 %0, %3, off offset:2048 sc1\n\tglobal_store_dwordx4 %0, %4, off offset:3072 sc1"
               :: "v"(dst), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]) : "memory");
}
#ifdef FUSED_TIMELINE
__device__ uint32_t g_tl_pub[1024 * 8];       // chain wavefront cw, publication k (8 quad rows each): when (s_memrealtime) -- before and after the wait for the stores
__device__ uint32_t* g_tl_flags;
#endif
__device__ __forceinline__ void publish_rows(uint32_t* flag, uint32_t epoch, uint32_t rows, uint32_t lane)
{
#ifdef FUSED_TIMELINE
  const uint32_t tl_a = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the record stores of these rows have completed
  if (lane == (uint32_t)__builtin_ctzll(__ballot(1))) {
    st_agent(flag, (epoch << 16) | rows);
#ifdef FUSED_TIMELINE
    const uint32_t k = rows == 0xFFFFu ? 7u : (rows & 7u) ? 8u : (rows >> 3) - 1u;     // (only the publications of whole slices of 8 rows are kept)
    if (k < 8u) { g_tl_pub[((flag - g_tl_flags) & 1023u) * 8u + k] = (uint32_t)__builtin_amdgcn_s_memrealtime(); if (k < 4u) g_tl_pub[((flag - g_tl_flags) & 1023u) * 8u + 4u + k] = tl_a; }
#endif
  }
}

// W64: the block is on the 64-bit sample path (ojph_decode_codeblock64): a decoded u above 32 -- before the initial
// row's bias of 2 when both quads have u > 2 -- is followed by a 4-bit extension, u += 4 ext (block_decoder64.cpp:
// 997-1011, :1119-1133).  A pair can then take 38 bits: the window is advanced and refilled in between.
template <class VlcRd>
__device__ __forceinline__ void uvlc_extension(VlcRd& vlc, uint32_t& used, uint32_t& u0, uint32_t& u1, uint32_t b0, uint32_t b1)
{
  const bool c0 = u0 - b0 > 32u, c1 = u1 - b1 > 32u;
  if (__ballot(c0 | c1) == 0ull) return;                  // (wave-uniform: no lane of the wavefront needs one)
  vlc.settle(); vlc.advance(used); vlc.prefetch(); vlc.settle(); used = 0;
  uint32_t v = vlc.peek();
  if (c0) { u0 += (v & 0xFu) << 2; v >>= 4; used += 4u; }
  if (c1) { u1 += (v & 0xFu) << 2; used += 4u; }
}

// FUSED: `rec` = the block's first 16 bytes of quad row 0 in the 16-bit layout, s_vlc = the dec_vlc32 entries,
// s_rec = the wavefront's staging area for one row of pair words (16 x 64 words of LDS).
// NARROW: 0 = blocks of any width (the significance of the row above is re-read from the records: a memory round trip on the
// chain), 1 = at most 32 quads per row (one 64-bit mask per lane), 2 = at most 64 quads per row (a 128-bit mask: 128 x 32 blocks)
template <int NARROW, bool FUSED = false, bool W64 = false, class VlcRd, class TblT>
__device__ __forceinline__ void step1_rows(VlcRd& vlc, EvRd& mel, uint32_t* __restrict__ rec, uint32_t QW, uint32_t QH,
                                           const TblT* s_vlc, const uint16_t* s_uvlc0, uint32_t* flag = nullptr, uint32_t epoch = 0,
                                           lds_u32* s_rec = nullptr, SliceSched sched = SliceSched(S2_ROWS))
{
  static_assert(!FUSED || NARROW == 1, "the fused launch's records are for blocks of at most 64 columns");
  typedef typename std::conditional<NARROW == 2, unsigned __int128, uint64_t>::type Mask;
  // the first quarter of row qy of the block in the 16-bit layout (see flush_row16)
  auto row16 = [&](uint32_t qy_) { return rec + (size_t)qy_ * (64u * REC16_ROW_WORDS); };
  // bit c of sig_prev: the bottom sample of column c of the quad row above is significant
  // (rho bit 1 of quad c/2 for even c, rho bit 3 for odd c)
  Mask sig_prev = 0;
  const uint32_t PW = (QW + 1) >> 1;              // quad pairs per row

  // ---- initial quad row (block_decoder32.cpp:854-975) ----
  {
    uint32_t tleft = 0; Mask sig_cur = 0;
    for (uint32_t qx = 0; qx < QW; qx += 2) {
      uint32_t v = vlc.peek(), used = 0;  // 32 bits: a pair consumes at most 2*7 + 6 + 10 of them
      const uint32_t evq = mel.peek(); uint32_t ecnt = 0;     // the next 32 MEL events: a pair consumes at most 3
      uint32_t c_q = ((tleft & 0x10u) << 3) | ((tleft & 0xE0u) << 2);                       // :903
      uint32_t t0 = s_vlc[c_q + (v & 0x7Fu)];
      if (c_q == 0) { if (((evq >> ecnt) & 1u) == 0) t0 = 0; ecnt++; }                      // :882-894
      v >>= (t0 & 7u); used += t0 & 7u;
      uint32_t t1 = 0;
      if (qx + 1 < QW) {
        c_q = ((t0 & 0x10u) << 3) | ((t0 & 0xE0u) << 2);                                    // :934
        t1 = s_vlc[c_q + (v & 0x7Fu)];
        if (c_q == 0) { if (((evq >> ecnt) & 1u) == 0) t1 = 0; ecnt++; }
        v >>= (t1 & 7u); used += t1 & 7u;
      }
      tleft = t1;
      if (NARROW) {
        const uint32_t nib = FUSED ? (((t0 >> 9) & 3u) | ((t1 >> 7) & 0xCu))
                                   : (((t0 >> 5) & 1u) | ((t0 >> 6) & 2u) | ((t1 >> 3) & 4u) | ((t1 >> 4) & 8u));
        sig_cur |= (Mask)nib << (2u * qx);
      }
      uint32_t mode = ((t0 & 0x8u) << 3) | ((t1 & 0x8u) << 4);
      if (mode == 0xC0u) { if ((evq >> ecnt) & 1u) mode += 0x40u; ecnt++; }                 // :943-952
      uint32_t entry = s_uvlc0[mode + (v & 0x3Fu)];
      v >>= (entry & 7u); used += entry & 7u; entry >>= 3;
      uint32_t len = entry & 0xFu;
      const uint32_t tmp = v & ((1u << len) - 1u);
      used += len; entry >>= 4;
      len = entry & 7u; entry >>= 3;
      uint32_t u0 = 1u + (entry & 7u) + (tmp & ~(0xFFu << len));                            // kappa = 1 (:971-974)
      uint32_t u1 = 1u + (entry >> 3) + (tmp >> len);
      if (W64) { const uint32_t bias = mode == 0x100u ? 3u : 1u; uvlc_extension(vlc, used, u0, u1, bias, bias); }   // (kappa + what the encoder took off)
      vlc.settle(); vlc.advance(used); PIN_WINDOW(vlc); vlc.prefetch(); mel.advance(ecnt); mel.prefetch();
      if (FUSED) s_rec[(qx >> 1) * 64u + vlc.lane] = (t0 >> 16) | (u0 << 9) | (t1 & 0xFFFF0000u) | (u1 << 25);
      else store_pair(rec + (size_t)(qx >> 1) * REC_STRIDE, (uint32_t)t0 | (u0 << 16), (uint32_t)t1 | (u1 << 16));
    }
    if (FUSED) flush_row16(s_rec, vlc.lane, row16(0));
    sig_prev = sig_cur;
  }
  // ---- other quad rows (:977-1089) ----
  const TblT* tbl = s_vlc + 1024;
  for (uint32_t qy = 1; qy < QH; ++qy) {
    uint32_t* row = rec + (size_t)qy * PW * REC_STRIDE;                 // quad pair px of this row: row + px * REC_STRIDE (not FUSED)
    const uint32_t* above = row - (size_t)PW * REC_STRIDE;
    auto above_rec = [&](uint32_t q) { return above[(size_t)(q >> 1) * REC_STRIDE + (q & 1u)]; };
    uint32_t tleft = 0; Mask sig_cur = 0;
    const Mask sig_or = sig_prev | (sig_prev >> 1);
    uint32_t carry = ((uint32_t)sig_prev & 1u) << 7;              // "column -1 | column 0", as bit 7 of the first context
    for (uint32_t qx = 0; qx < QW; qx += 2) {
      uint32_t v = vlc.peek(), used = 0;
      const uint32_t evq = mel.peek(); uint32_t ecnt = 0;
      // k0 / k1: what the sample row above contributes to the contexts of quad qx / qx+1 (:990-991,
      // :1024-1027): bit 7 = nw | n, bit 9 = ne | nf, over the columns 2qx-1 .. 2qx+4
      uint32_t k0, k1;
      if (NARROW) {
        // sig_or bit j = column j | column j+1 of the row above: the three pairs a context needs sit at bits 2qx-1
        // (kept from the previous pair as `carry`, already in place), 2qx+1 and 2qx+3
        const uint32_t w = (uint32_t)(sig_or >> (2u * qx));
        k0 = carry | ((w & 2u) << 8);
        k1 = (w & 0xAu) << 6;
        carry = (w & 8u) << 4;
      } else {
        const uint32_t up0 = above_rec(qx);
        const uint32_t upl = qx ? above_rec(qx - 1) : 0u;
        const uint32_t up1 = qx + 1 < QW ? above_rec(qx + 1) : 0u;
        const uint32_t up2 = qx + 2 < QW ? above_rec(qx + 2) : 0u;
        const uint32_t sg = ((upl >> 7) & 1u) | (((up0 >> 5) & 1u) << 1) | (((up0 >> 7) & 1u) << 2) | (((up1 >> 5) & 1u) << 3) |
                            (((up1 >> 7) & 1u) << 4) | (((up2 >> 5) & 1u) << 5);
        k0 = (((sg | (sg >> 1)) & 1u) << 7) | ((((sg >> 2) | (sg >> 3)) & 1u) << 9);
        k1 = ((((sg >> 2) | (sg >> 3)) & 1u) << 7) | ((((sg >> 4) | (sg >> 5)) & 1u) << 9);
      }
      // both look-ups and their MEL corrections as selects, no branches (the second quad of an odd-width block's last
      // pair does not exist: its look-up is made all the same and dropped)
      // (FUSED: the table's entries carry "a significant sample in the right column" ready made in bit 8, and the bottom
      // row's two significance bits in bits 9, 10 -- ht_tables.cpp)
      const uint32_t c_q0 = FUSED ? ((tleft & 0x100u) | k0) : (((tleft & 0x40u) << 2) | ((tleft & 0x80u) << 1) | k0);           // :1022
      uint32_t t0 = tbl[c_q0 + (v & 0x7Fu)];
      const bool z0 = c_q0 == 0;
      t0 = (z0 & ((evq & 1u) == 0)) ? 0u : t0;
      ecnt = z0 ? 1u : 0u;
      v >>= (t0 & 7u); used += t0 & 7u;
      const bool ex1 = qx + 1 < QW;
      const uint32_t c_q1 = FUSED ? ((t0 & 0x100u) | k1) : (((t0 & 0x40u) << 2) | ((t0 & 0x80u) << 1) | k1);                 // :1059
      uint32_t t1 = tbl[c_q1 + (v & 0x7Fu)];
      const bool z1 = ex1 & (c_q1 == 0);
      t1 = ((!ex1) | (z1 & (((evq >> ecnt) & 1u) == 0))) ? 0u : t1;
      ecnt += z1 ? 1u : 0u;
      v >>= (t1 & 7u); used += t1 & 7u;
      tleft = t1;
      if (NARROW) {
        const uint32_t nib = FUSED ? (((t0 >> 9) & 3u) | ((t1 >> 7) & 0xCu))
                                   : (((t0 >> 5) & 1u) | ((t0 >> 6) & 2u) | ((t1 >> 3) & 4u) | ((t1 >> 4) & 8u));
        sig_cur |= (Mask)nib << (2u * qx);
      }
      // the pair's U-VLC by arithmetic instead of the uvlc_tbl1 look-up (:1065-1085): one LDS round trip less on the chain
      uint32_t u0, u1;
      used += ojphgpu::uvlc_pair_other_rows(v, t0 & 0x8u, t1 & 0x8u, u0, u1);
      if (W64) uvlc_extension(vlc, used, u0, u1, 0u, 0u);
      vlc.settle(); vlc.advance(used); PIN_WINDOW(vlc); vlc.prefetch(); mel.advance(ecnt); mel.prefetch();
      if (FUSED) s_rec[(qx >> 1) * 64u + vlc.lane] = (t0 >> 16) | (u0 << 9) | (t1 & 0xFFFF0000u) | (u1 << 25);
      else store_pair(row + (size_t)(qx >> 1) * REC_STRIDE, (uint32_t)t0 | (u0 << 16), (uint32_t)t1 | (u1 << 16));
    }
    if (FUSED) flush_row16(s_rec, vlc.lane, row16(qy));
    sig_prev = sig_cur;
    if (FUSED && sched.publishes_after(qy + 1u)) publish_rows(flag, epoch, qy + 1u, vlc.lane);
  }
}

// -------------------------------------------------------------------------------------------------
// step 1 without a prep launch: the partner wavefront un-stuffs the block's VLC and MEL bytes itself
// -------------------------------------------------------------------------------------------------
// The flat bit strings the chain reads do not have to pass through HBM: the partner wavefront of a chain wavefront (one
// lane per code-block, like the chain) reads the raw bytes of its block's cleanup segment -- the VLC part backwards from
// the end, the MEL part forwards -- un-stuffs them with the rules flatten() applies, and hands the chain
//   * the VLC bits as 32-bit words of the same flat LSB-first string, through a ring of VR_WORDS words per lane in LDS,
//   * the MEL events decoded from the MEL bits, as before,
// both demand driven: it stays a few words ahead of what the chain has taken.  Neither the chain nor anything else waits
// for a prep launch (0.065 ms of an 8K frame's decode), the flat strings (two thirds of a byte per coded byte, written and
// read once) never exist in memory, and the chain wavefront has no global load left in its loop: its record stores are the
// only thing its vmcnt counts, so nothing on the chain ever waits for memory.
constexpr uint32_t VR_WORDS = 16, VR_LOW = 6, EV_LOW = 2;    // high / low marks of the partner's bursts
#ifndef MEL_RUNS_PER_PASS
#define MEL_RUNS_PER_PASS 16
#endif          // ring of flat VLC words per lane (the chain holds three more in registers)

// chain side: the flat VLC string through the ring -- same window as FlatRd, words fetched from LDS
struct RingRd {
  const lds_u32* ring; volatile lds_u32* prog; volatile lds_u32* cons; uint32_t idx, lo, hi, pre, bp, av, pidx, lane; bool stuck;
  __device__ __forceinline__ uint32_t fetch(uint32_t want) {      // blocking: waits until the partner has produced word `want`
    if (stuck) return 0u;                                    // (gave up on this block before: no second wait)
    uint32_t avail = prog[lane], spins = 0;
    while (want >= avail && ++spins < LDS_WAIT_SPINS) { __builtin_amdgcn_s_sleep(2); avail = prog[lane]; }

    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    stuck = stuck || want >= avail;                          // cannot happen (the partner always progresses); never hang
    return ring[(want & (VR_WORDS - 1u)) * 64u + lane];
  }
  __device__ __forceinline__ void init(const lds_u32* r, volatile lds_u32* p, volatile lds_u32* c, uint32_t l) {
    ring = r; prog = p; cons = c; lane = l; stuck = false;
    lo = fetch(0); hi = fetch(1); pre = fetch(2);
    idx = 2; bp = 0; pidx = 0; av = 1;
  }
  __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(hi, lo, bp); }
  __device__ __forceinline__ void advance(uint32_t k) {
    bp += k;
    const bool cross = bp >= 32u;
    lo = cross ? hi : lo; hi = cross ? pre : hi;
    bp = cross ? bp - 32u : bp; idx += cross ? 1u : 0u;
  }
  // The word wanted next is read SPECULATIVELY together with the partner's progress count (progress first: LDS reads
  // complete in order), both one quad pair before they are looked at; settle() -- called before the word can move into
  // the window -- repeats the read in the rare case the partner had not got that far.  No lane polls in the steady
  // state, so the 64 chains of the wavefront do not drag each other through a wait loop.
  __device__ __forceinline__ void prefetch() { cons[lane] = idx; pidx = idx; av = prog[lane]; pre = ring[(idx & (VR_WORDS - 1u)) * 64u + lane]; }
  __device__ __forceinline__ void settle() {
    if (pidx >= av) {
      pre = fetch(pidx); av = pidx + 1u;
    }
  }
};

// The partner's work for one lane's code-block.  VLC: flatten<false>'s string, word by word (rev_read :308-400: bytes from
// cb[lcup - 3] downwards, 7 bits when the byte above is > 0x8F and the low 7 bits are ones, every raw byte OR-ed in with all
// its bits, the 4 (3) initial bits from the high nibble of cb[lcup - 2], zeros beyond the segment).  MEL: mel_read :93-157
// (forwards from cb[lcup - scup], 7 bits after an 0xFF, MSB first, the last byte's low nibble forced to ones, ones beyond
// the segment) feeding the run-length decoder of mel_producer.
struct __attribute__((aligned(4))) B16 { uint32_t d[4]; };
__device__ __forceinline__ B16 load_b16_unaligned(const uint8_t* p) { B16 v; __builtin_memcpy(&v, p, 16); return v; }

template <int ROLE>               // 1: the VLC words, 2: the MEL events (one partner wavefront each)
__device__ __forceinline__ void raw_partner(const uint8_t* __restrict__ cb, uint32_t lcup, uint32_t scup, uint32_t room_below,
                                            const B16& v0, const B16& v1, uint32_t ev_words,
                                            lds_u32* s_ev, volatile lds_u32* s_eprog, const volatile lds_u32* s_econs,
                                            lds_u32* s_vr, volatile lds_u32* s_vprog, const volatile lds_u32* s_vcons,
                                            const volatile lds_u32* s_done, uint32_t lane)
{
  // Bytes come 16 at a time, one such chunk requested ahead of the one being worked on (a lane's loads are its own:
  // nothing to coalesce, so what counts is round trips -- 4 bytes per trip kept the partner behind its chain).
  // ---- VLC state ----
  const uint32_t vreal = scup - 2u, vcount = vreal + 1u;     // real bytes; + the zero byte that keeps a stray bit of the last one
  const uint32_t d0 = cb[lcup - 2];
  uint64_t vacc = d0 >> 4;
  uint32_t vnb = 4u - ((((d0 >> 4) & 7u) == 7u) ? 1u : 0u);
  uint32_t vk = 0, vwr = 0, vp1 = d0 | 0xFu;
  bool vp1_short = false;
  // chunk c of the backward string = raw bytes 16c .. 16c+15: byte 16c in bits 31..24 of d[3], byte 16c+15 in bits 7..0 of d[0]
  auto vload = [&](uint32_t k) -> B16 {
    B16 z; z.d[0] = z.d[1] = z.d[2] = z.d[3] = 0u;
    if (k >= vcount) return z;
    const int off = (int)lcup - 18 - (int)k;                 // address of byte k+15
    if (off >= -(int)room_below) return load_b16_unaligned(cb + off);   // (bytes below the segment do not count: masked by k+j < vreal)
    for (int j = 0; j < 16; ++j) {                           // the first bytes of the buffer: never read below it
      const int o = off + 15 - j;                            // address of byte k+j
      if (o >= 0) z.d[3 - (j >> 2)] |= (uint32_t)cb[o] << (24 - 8 * (j & 3));
    }
    return z;
  };
  B16 vcur = v0, vnext = v1;
  // ---- MEL state ----
  const uint32_t mcount = scup - 1u, moff = lcup - scup;
  auto mload = [&](uint32_t k) -> B16 {                      // raw bytes k .. k+15, byte k in bits 7..0 of d[0]
    if (k < mcount) return load_b16_unaligned(cb + moff + k);
    B16 z; z.d[0] = z.d[1] = z.d[2] = z.d[3] = 0xFFFFFFFFu; return z;
  };
  uint64_t win = 0;                                          // MSB first: the next bit is bit 63
  uint32_t n = 0, mk = 0, mprev = 0, k = 0, nev = 0, ewr = 0;
  B16 mcur, mnext;
  if (ROLE == 2) { mcur = mload(0); mnext = mload(16); }
  uint64_t ev = 0;
  // The partner works in BURSTS: it wakes up when some lane's chain has come within the low mark of what was produced,
  // fills every lane up to the high mark, and sleeps again.  A partner that tops up one word whenever one was taken runs all
  // the time -- and it shares its SIMD with a chain wavefront, whose every instruction then queues behind a partner
  // instruction that has just been issued.
  bool burst = true;
  for (;;) {
    if (s_done[lane] != 0u) break;
    if (!burst) {
      const bool low = ROLE == 1 ? vwr < s_vcons[lane] + VR_LOW : (ewr < ev_words && ewr < s_econs[lane] + EV_LOW);
      if (__ballot(low) == 0ull) { __builtin_amdgcn_s_sleep(32); continue; }
      burst = true;
    }
    bool did = false;
    if (ROLE == 1 && vwr < s_vcons[lane] + VR_WORDS) {       // ---- four more VLC bytes ----
      did = true;
      // the dword's four bytes at once (the byte-wise rule is in flatten<false>): W holds raw byte vk + j in byte j
      uint32_t W = __builtin_bswap32(vcur.d[3]);
      vcur.d[3] = vcur.d[2]; vcur.d[2] = vcur.d[1]; vcur.d[1] = vcur.d[0];
      const uint32_t left = vk < vreal ? vreal - vk : 0u;                      // real bytes from vk on
      W &= left >= 4u ? 0xFFFFFFFFu : (1u << (8u * left)) - 1u;               // beyond them: zeros
      const uint32_t P = (W << 8) | vp1;                                       // byte j: the raw byte before byte j
      const uint32_t G = ((P & 0x7F7F7F7Fu) + 0x70707070u) & P & 0x80808080u;  // ... is > 0x8F
      const uint32_t L7 = ((W & 0x7F7F7F7Fu) + 0x01010101u) & 0x80808080u;     // the low seven bits of byte j are ones
      const uint32_t S = G & L7;                                               // byte j carries 7 bits
      const uint32_t Sprev = (S << 8) | (vp1_short ? 0x80u : 0u);             // byte j-1 carried 7 bits ...
      const uint32_t stray = ((Sprev & P) >> 7) & 0x01010101u;                 // ... and its MSB was set: OR-ed onto byte j's LSB
      uint32_t x = (W | stray) & ~S;
      x = (S & 0x00800000u) ? (x & 0x007FFFFFu) | ((x >> 1) & 0xFF800000u) : x;     // close the gaps above 7-bit bytes
      x = (S & 0x00008000u) ? (x & 0x00007FFFu) | ((x >> 1) & 0xFFFF8000u) : x;
      x = (S & 0x00000080u) ? (x & 0x0000007Fu) | ((x >> 1) & 0xFFFFFF80u) : x;
      vacc |= (uint64_t)x << vnb;
      vnb += 32u - (uint32_t)__popc(S);
      vp1 = W >> 24; vp1_short = (S >> 31) != 0u;
      vk += 4u;
      if ((vk & 15u) == 0u) { vcur = vnext; vnext = vload(vk + 16u); }
      if (vnb >= 32u) {
        s_vr[(vwr & (VR_WORDS - 1u)) * 64u + lane] = (uint32_t)vacc; vacc >>= 32; vnb -= 32u; ++vwr;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        s_vprog[lane] = vwr;
      }
    }
    if (ROLE == 2 && ewr < ev_words && ewr < s_econs[lane] + EV_AHEAD) {  // ---- MEL: bytes in, events out ----
      did = true;
      if (n <= 32u) {
        uint32_t w = mcur.d[0];
        mcur.d[0] = mcur.d[1]; mcur.d[1] = mcur.d[2]; mcur.d[2] = mcur.d[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t b = w & 0xFFu; w >>= 8;
          const uint32_t kk = mk + (uint32_t)j;
          if (kk == mcount - 1u) b |= 0xFu;                  // the last MEL byte shares its low nibble with VLC (:116)
          const uint32_t nbits = kk < mcount ? 8u - (mprev == 0xFFu ? 1u : 0u) : 8u;
          const uint32_t val = kk < mcount ? b & ((1u << nbits) - 1u) : 0xFFu;
          win |= (uint64_t)val << (64u - n - nbits);
          n += nbits; mprev = kk < mcount ? b : 0u;
        }
        mk += 4u;
        if ((mk & 15u) == 0u) { mcur = mnext; mnext = mload(mk + 16u); }
      }
      // (a bounded number of runs per pass: a lane that needs many events must not keep the other 63 lanes' VLC words waiting)
      for (int it = 0; it < MEL_RUNS_PER_PASS && nev <= 31u && n >= 6u; ++it) mel_run(win, n, k, ev, nev);
      if (nev >= 32u) {
        s_ev[(ewr & (EV_RING - 1u)) * 64u + lane] = (uint32_t)ev; ev >>= 32; nev -= 32u; ++ewr;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        s_eprog[lane] = ewr;
      }
    }
    if (__ballot(did) == 0ull) burst = false;                // every lane is at the high mark
  }
}

template <int CH>                 // CH chain wavefronts + 2 CH partner wavefronts per workgroup, 64 code-blocks per three of them
__global__ __launch_bounds__(192 * CH) void ht_dec_step1_raw_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    uint32_t* __restrict__ quads, uint8_t* __restrict__ block_status)
{
  __shared__ uint16_t s_vlc[2048];
  __shared__ uint16_t s_uvlc0[320];
  __shared__ uint32_t s_ev_all[CH][EV_RING * 64];
  __shared__ uint32_t s_vr_all[CH][VR_WORDS * 64];
  __shared__ uint32_t s_ctl_all[CH][5][64];            // MEL words done / taken, VLC words done / wanted next, block done
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_vlc[i] = (&ojphgpu::g_dec_vlc[0][0])[i];
  for (int i = threadIdx.x; i < 320; i += blockDim.x) s_uvlc0[i] = ojphgpu::g_dec_uvlc0[i];
  for (int i = threadIdx.x; i < CH * 5 * 64; i += blockDim.x) (&s_ctl_all[0][0][0])[i] = 0;
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool chain = wv < (uint32_t)CH;
  const uint32_t set = wv % (uint32_t)CH;               // wavefronts CH .. 2CH-1: the VLC words, 2CH .. 3CH-1: the MEL events of the same blocks
  lds_u32* s_ev = (lds_u32*)s_ev_all[set];
  lds_u32* s_vr = (lds_u32*)s_vr_all[set];
  volatile lds_u32* s_eprog = (volatile lds_u32*)s_ctl_all[set][0];
  volatile lds_u32* s_econs = (volatile lds_u32*)s_ctl_all[set][1];
  volatile lds_u32* s_vprog = (volatile lds_u32*)s_ctl_all[set][2];
  volatile lds_u32* s_vcons = (volatile lds_u32*)s_ctl_all[set][3];
  volatile lds_u32* s_done = (volatile lds_u32*)s_ctl_all[set][4];
  if (chain) __builtin_amdgcn_s_setprio(3);
  const uint32_t bi = (blockIdx.x * (uint32_t)CH + set) * 64u + lane;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (d.reversible & 4u) return;                          // 64-bit sample path: ht_dec_step1_kernel<CH, true> on the prep strings
  if (d.w == 0 || d.h == 0 || d.len1 == 0 || d.num_passes == 0) { if (chain) block_status[bi] = 0; return; }   // not coded: zero block
  const uint8_t* cb = data + d.data_off;
  const uint32_t room = d.data_off > 0xFFFFu ? 0xFFFFu : (uint32_t)d.data_off;
  // The VLC partner asks for the first 32 bytes below the segment's end before anybody knows where the segment begins
  // (scup sits in its last two bytes): they travel together with the two bytes check_block() reads, one round trip to
  // memory instead of two in front of the first VLC word.  What lies below the VLC part is masked when it is used.
  B16 v0, v1;
  v0.d[0] = v0.d[1] = v0.d[2] = v0.d[3] = 0u; v1 = v0;
  if (!chain && wv < 2u * (uint32_t)CH) {
    const int o0 = (int)d.len1 - 18, o1 = o0 - 16;
    if (o0 >= -(int)room) v0 = load_b16_unaligned(cb + o0);
    else for (int j = 0; j < 16; ++j) { const int o = o0 + 15 - j; if (o >= 0) v0.d[3 - (j >> 2)] |= (uint32_t)cb[o] << (24 - 8 * (j & 3)); }
    if (o1 >= -(int)room) v1 = load_b16_unaligned(cb + o1);
    else for (int j = 0; j < 16; ++j) { const int o = o1 + 15 - j; if (o >= 0) v1.d[3 - (j >> 2)] |= (uint32_t)cb[o] << (24 - 8 * (j & 3)); }
  }
  const uint32_t scup = check_block(d, cb);
  if (scup == 0) { if (chain) block_status[bi] = 1; return; }
  const uint32_t QW = ((uint32_t)d.w + 1) >> 1, QH = ((uint32_t)d.h + 1) >> 1;
  const uint32_t evw = ev_words_of(QW, QH);
  if (!chain) {
    if (wv < 2u * (uint32_t)CH) raw_partner<1>(cb, d.len1, scup, room, v0, v1, evw, s_ev, s_eprog, s_econs, s_vr, s_vprog, s_vcons, s_done, lane);
    else raw_partner<2>(cb, d.len1, scup, room, v0, v1, evw, s_ev, s_eprog, s_econs, s_vr, s_vprog, s_vcons, s_done, lane);
    return;
  }
  uint32_t* rec = quads + d.scratch_cap;
  RingRd vlc; vlc.init(s_vr, s_vprog, s_vcons, lane);
  EvRd mel; mel.init(s_ev, s_eprog, s_econs, evw, lane);
  if (__all(QW <= 32)) step1_rows<1>(vlc, mel, rec, QW, QH, s_vlc, s_uvlc0);
  else if (__all(QW <= 64)) step1_rows<2>(vlc, mel, rec, QW, QH, s_vlc, s_uvlc0);
  else step1_rows<0>(vlc, mel, rec, QW, QH, s_vlc, s_uvlc0);
  s_done[lane] = 1u;
  block_status[bi] = (mel.stuck || vlc.stuck) ? 1 : 0;
}

template <int CH, bool W64 = false>   // CH chain wavefronts + CH partner wavefronts per workgroup, 64 code-blocks per pair of them
__global__ __launch_bounds__(128 * CH) void ht_dec_step1_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    const uint32_t* __restrict__ aux, uint32_t* __restrict__ quads, uint8_t* __restrict__ block_status)
{
  __shared__ uint16_t s_vlc[2048];
  __shared__ uint16_t s_uvlc0[320];
  __shared__ uint32_t s_ev_all[CH][EV_RING * 64];
  __shared__ uint32_t s_prog_all[CH][64];
  __shared__ uint32_t s_done_all[CH][64];
  __shared__ uint32_t s_cons_all[CH][64];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_vlc[i] = (&ojphgpu::g_dec_vlc[0][0])[i];
  for (int i = threadIdx.x; i < 320; i += blockDim.x) s_uvlc0[i] = ojphgpu::g_dec_uvlc0[i];
  if (threadIdx.x < 64 * CH) { (&s_prog_all[0][0])[threadIdx.x] = 0; (&s_done_all[0][0])[threadIdx.x] = 0; (&s_cons_all[0][0])[threadIdx.x] = 0; }
  __syncthreads();

  // wavefronts 0 .. CH-1: the chains of 64 code-blocks each; wavefronts CH .. 2CH-1: the MEL events of the same blocks
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool chain = wv < (uint32_t)CH;
  const uint32_t set = chain ? wv : wv - (uint32_t)CH;
  uint32_t* s_ev = s_ev_all[set];
  uint32_t* s_prog = s_prog_all[set];
  volatile lds_u32* s_done = (volatile lds_u32*)s_done_all[set];
  volatile lds_u32* s_cons = (volatile lds_u32*)s_cons_all[set];
  // the chain wavefronts are bound by the latency of their serial chains: when anything shares their SIMD
  // (the partner, another stream, the tail of the prep launch) they should win the issue arbitration
  if (chain) __builtin_amdgcn_s_setprio(3);
  const uint32_t bi = (blockIdx.x * (uint32_t)CH + set) * 64u + lane;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (((d.reversible & 4u) != 0) != W64) return;           // the other instantiation's blocks (64-bit sample path or not)
  if (d.w == 0 || d.h == 0 || d.len1 == 0 || d.num_passes == 0) { if (chain) block_status[bi] = 0; return; }   // not coded: zero block
  const uint8_t* cb = data + d.data_off;
  const uint32_t scup = check_block(d, cb);
  if (scup == 0) { if (chain) block_status[bi] = 1; return; }
  const uint32_t QW = ((uint32_t)d.w + 1) >> 1, QH = ((uint32_t)d.h + 1) >> 1;
  const uint32_t evw = ev_words_of(QW, QH);
  if (!chain) {
    mel_producer(aux + d.reserved + vlc_words(scup), mel_words(scup), evw, (lds_u32*)s_ev, (volatile lds_u32*)s_prog, s_cons, s_done, lane);
    return;
  }
  uint32_t* rec = quads + d.scratch_cap;          // pair p of this block: rec + 128 p (interleaved with the wavefront's other 63 blocks)

  FlatRd vlc; vlc.init(aux + d.reserved, vlc_words(scup));
  EvRd mel; mel.init((const lds_u32*)s_ev, (volatile lds_u32*)s_prog, s_cons, evw, lane);

  // every block of this wavefront at most 64 samples wide (the usual case): the significance of the
  // sample row above lives in one 64-bit mask per lane instead of being re-read from the records
  if (__all(QW <= 32)) step1_rows<1, false, W64>(vlc, mel, rec, QW, QH, s_vlc, s_uvlc0);
  else if (__all(QW <= 64)) step1_rows<2, false, W64>(vlc, mel, rec, QW, QH, s_vlc, s_uvlc0);
  else step1_rows<0, false, W64>(vlc, mel, rec, QW, QH, s_vlc, s_uvlc0);
  s_done[lane] = 1u;                               // the partner stops producing events for this block
  block_status[bi] = mel.stuck ? 1 : 0;
}

// -------------------------------------------------------------------------------------------------
// step 2: one wavefront = one code-block, one lane = one sample column
// -------------------------------------------------------------------------------------------------

#ifndef S2_ABL
#define S2_ABL 0                  // attribution experiments on step 2's row loop (never in the product; results are wrong): 1 the sample stores only happen
#endif                            // for a value that never occurs, 2 no prefix sum (a product stands in), 4 no ring reads, 8 no neighbour exponents, 16 no un-stuffing
// de-quantise transfer of one sign-magnitude word (ojph_codestream_gen.cpp:124-168)
__device__ __forceinline__ uint32_t dequantise(uint32_t val, bool rev, uint32_t shift, float delta)
{
  const uint32_t mag = val & 0x7FFFFFFFu;
  if (rev) { const uint32_t iv = mag >> shift, sg = (uint32_t)((int)val >> 31); return (iv ^ sg) - sg; }   // -iv for a set sign
  // (the product of two non-negative floats has a clear sign bit: OR-ing the sample's sign in negates it, -0.0f for a zero
  // magnitude included -- as "-fv" does)
  return __float_as_uint(__fmul_rn((float)mag, delta)) | (val & 0x80000000u);
}

// does the block carry SigProp / MagRef passes that will be decoded (block_decoder32.cpp:752-789)?
__device__ __forceinline__ bool needs_refinement(const ojphgpu_cb_desc& d)
{
  return d.num_passes > 1 && d.num_passes <= 3 && d.len2 > 0 && (d.missing_msbs < 29 || (d.reversible & 4u));   // (:783-789: 32-bit path only)
}

// TX / WD: what the host knows about EVERY block of the launch (0 = nothing, decided per block at run time).
// TX 1: reversible, no refinement passes; TX 2: irreversible, no refinement passes.  WD 1: no block wider than 64
// samples.  The flags are wave-uniform either way; as template constants they take ~20 scalar tests and branches
// out of every quad row.
// Per-block state a sliced step 2 keeps between two slices of a block (in LDS, 20 words per block): words 0..15 the
// bottom-row exponents of the 64 columns (a byte each), 16 the MagSgn bits decoded so far, 17 a restart point of the
// un-stuffer (bytes consumed, a multiple of 256), 18 the un-stuffed bits that point corresponds to, 19 != 0: the block failed.
constexpr uint32_t S2_STATE_WORDS = 24;       // (words 0..19 as above; 20: the length of the block's MagSgn part, kept from the prepare call)

// -DFUSED_TIMELINE (tools/fused_timeline.py; never in the product build): every wavefront of the fused launch leaves, per
// run, 12 words in g_tl -- role, where it ran, when it started / ended (s_memrealtime, 100 MHz), and for a worker how
// long it waited for chains and when it finished each slice of its blocks
#ifdef FUSED_TIMELINE
__device__ uint32_t g_tl[8192 * 12];
__device__ __forceinline__ uint32_t tl_now() { return (uint32_t)__builtin_amdgcn_s_memrealtime(); }
#endif
constexpr uint32_t TICKET_STRIDE = 32;        // words between the fused launch's ticket counters: a cache line each
constexpr uint32_t TICKET_CU_WORDS = 8 * 256; // a counter per compute unit: 8 XCDs x (SE_ID, SH_ID, CU_ID of HW_REG_HW_ID)
constexpr uint32_t TICKET_WORDS = TICKET_CU_WORDS + 2 * TICKET_STRIDE;   // those + the 64-bit counters of the step-1 and the worker numbers

// fused launch: where (in words of the record scratch) the 16-bit records of block `bi`, quad row 0, first quarter, are -- its group of
// 64 blocks starts where the 32-bit layout's does (scratch_cap = that + 2 (bi % 64), ojphgpu_ht_decode_layout)
__device__ __forceinline__ uint32_t rec16_base(const ojphgpu_cb_desc& d, uint32_t bi)
{
  const uint32_t l = bi & 63u;
  return d.scratch_cap - 2u * l + l * 4u;
}

// the fused launch's records of quad rows [qy_begin, min(qy_end, QH)) of block bi: one 16-bit record per quad, lane l = the
// quad of column l (16-bit layout [row][quarter][block], flush_row16).  Eight agent-scope loads in flight, no wait here.
__device__ __forceinline__ void load_slice_records(const uint32_t* __restrict__ quads, const ojphgpu_cb_desc& d, uint32_t bi,
                                                   uint32_t qy_begin, uint32_t qy_end, int lane, uint32_t* ents)
{
  const uint32_t QH = ((uint32_t)d.h + 1u) >> 1;
  const uint16_t* r16 = reinterpret_cast<const uint16_t*>(quads + rec16_base(d, bi) + (size_t)qy_begin * (64u * REC16_ROW_WORDS) + ((uint32_t)lane >> 4) * 256u)
                        + (((uint32_t)lane >> 1) & 7u);
  constexpr uint32_t row_step = 2u * 64u * REC16_ROW_WORDS;     // (in 16-bit units)
  // (no conditions: the layout has all 32 quads of every row; what an idle lane brings is not looked at)
  const uint32_t last = (qy_end < QH ? qy_end : QH) - qy_begin - 1u;      // (a short slice loads its last row again: never beyond what the chain has written)
#pragma unroll
  for (uint32_t i = 0; i < S2_ROWS; ++i)
    ents[i] = (uint32_t)__hip_atomic_load(r16 + (i < last ? i : last) * row_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Step 2 of ONE code-block by one wavefront: quad rows [qy_begin, qy_end) -- the whole block for the plain kernel;
// SLICED: a slice of rows, state from / to `state`, per-quad records read with agent scope (the fused kernel).
template <int TX, int WD, bool SLICED, bool KEEP = false>
__device__ __forceinline__ void step2_block(const ojphgpu_cb_desc& d, uint32_t bi, const uint8_t* __restrict__ data,
                                            const uint32_t* __restrict__ quads, uint32_t* __restrict__ coef,
                                            uint8_t* __restrict__ block_status, uint32_t* ring, uint8_t* s_exp_w,
                                            int lane, uint32_t qy_begin, uint32_t qy_end, uint32_t* state, bool prepare = false)
{
  // KEEP (a ring per block): the slices are preceded by ONE call with `prepare` set, made before the worker waits for
  // anything -- the first MagSgn bytes do not depend on the chain, so the block's ring is cleared and filled and its state
  // set up while the chains work on their first rows; every slice, the first included, then goes on from that state.
  const uint32_t W = d.w, H = d.h, pitch = d.pitch;
  if (W == 0 || H == 0) return;
  uint32_t* dst = coef + d.coef_off;
  const bool rev = TX == 1 ? true : TX == 2 ? false : (d.reversible & 1u) != 0;
  const uint32_t K = d.K_max;
  const bool raw_out = TX ? false : needs_refinement(d);     // SigProp / MagRef follow: keep sign-magnitude words

  auto zero_block = [&]() {                                  // mem_clear path, ojph_codeblock.cpp:247
    for (uint32_t y = 0; y < H; ++y)
      for (uint32_t x = lane; x < W; x += 64) dst[(size_t)y * pitch + x] = 0u;
  };
  static_assert(!SLICED || WD == 1, "slices are for blocks of at most 64 columns");
  if (d.len1 == 0 || d.num_passes == 0) {
    if ((!SLICED || qy_begin == 0) && !prepare) zero_block();
    return;
  }
  const uint32_t missing_msbs = d.missing_msbs;
  const uint32_t p = 30 - missing_msbs;
  const uint8_t* cb = data + d.data_off;
  const uint32_t lcup = d.len1;
  const uint32_t QW = (W + 1) >> 1, QH = (H + 1) >> 1;
  const uint32_t PW = (QW + 1) >> 1;
  const uint32_t* rec = quads + d.scratch_cap;
  const uint32_t qy_last = qy_end < QH ? qy_end : QH;
  // (SLICED: agent-scope loads.  Plain loads were tried -- "the first touch of a line misses and fetches what the chain
  // wrote" -- and returned stale data on the first run over a scratch area: an agent-scope store that has completed is
  // visible to agent-scope loads, not necessarily to a plain load through another XCD's L2.)
  auto rec_at = [&](uint32_t qy_, uint32_t qx_) -> uint32_t {
    const uint32_t* q = rec + (size_t)(qy_ * PW + (qx_ >> 1)) * REC_STRIDE + (qx_ & 1u);
    return SLICED ? ld_agent(q) : *q;
  };
  // SLICED: everything the slice needs from memory is requested at once -- the block's verdict, the length bytes and
  // the records of ALL its quad rows (they are complete: the chain has published them) -- one round trip per slice where
  // a record fetched one row ahead made it one per row (an agent-scope load takes longer than a row's arithmetic).
  // KEEP: what a slice would have to fetch again and again does not change between the slices of a block and is decided by
  // the prepare call: the verdict of check_block -- the very test step 1 applies, on the same bytes -- and the length of the
  // MagSgn part.  Both wait in the block's LDS state (word 19: 0 go on, 1 the block is zero and done, 2 refused by
  // check_block, its first slice still has to zero it; word 20: the length), so that a slice starts without a round trip to
  // memory for the status byte and the two length bytes -- a worker wavefront runs its ~35 slices one after the other, and
  // with nothing prefetched every one of them began with two dependent round trips (measured with the row loop emptied:
  // 0.155 of the 0.204 ms the workers take alone).
  uint32_t st = 0, ms_len = 0;
  if (KEEP && prepare) {
    const uint32_t scup0 = check_block(d, cb);
    if (scup0 == 0u) { if (lane == 0) state[19] = 2u; wave_sync(); return; }
    ms_len = lcup - scup0;
  } else if (KEEP) {
    const uint32_t verdict = rdfirst(state[19]);
    if (verdict == 2u && qy_begin == 0) { zero_block(); if (lane == 0) state[19] = 1u; wave_sync(); }
    if (verdict != 0u) return;
    ms_len = rdfirst(state[20]);
  } else {
    // (what every lane reads from one address is kept as what it is, a scalar: comparisons and sums of such values then run
    // on the scalar unit, beside the vector instructions of another wavefront)
    st = rdfirst(SLICED ? (uint32_t)__hip_atomic_load(block_status + bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)block_status[bi]);
    const uint32_t len_b1 = rdfirst(cb[lcup - 1]), len_b2 = rdfirst(cb[lcup >= 2u ? lcup - 2u : 0u]);   // (a one-byte segment has failed in step 1; its bytes are not used)
    ms_len = lcup - ((len_b1 << 4) + (len_b2 & 0xFu));
  }
  uint32_t ents[SLICED ? S2_ROWS : 1u];
  if (SLICED && !(KEEP && prepare)) {
    load_slice_records(quads, d, bi, qy_begin, qy_end, lane, ents);
  }
  if (st != 0) {
    if (!SLICED || qy_begin == 0) zero_block();
    return;
  }
  const bool wide = WD == 1 ? false : W > 64;

  if (!KEEP || prepare)
    for (uint32_t i = lane; i < RING_ALLOC; i += 64) ring[i] = 0;
  if (wide) for (uint32_t i = lane; i < 2 * EXP_BYTES / 4; i += 64) reinterpret_cast<uint32_t*>(s_exp_w)[i] = 0;
  wave_sync();

  // un-stuffs the next 256 MagSgn bytes into the ring ("after 0xFF only 7 bits", :609-653)
  uint32_t dst_bits = 0, src_pos = 0, mpos = 0;     // bits un-stuffed, bytes consumed, bits decoded (wave-uniform)
  uint32_t hs[5] = { 0, 0, 0, 0, 0 }, hd[5] = { 0, 0, 0, 0, 0 };   // SLICED: the chunks un-stuffed last -- where they began (bytes, bits)
  auto unstuff_chunk = [&]() {
    if (SLICED && !KEEP) {
#pragma unroll
      for (int i = 4; i > 0; --i) { hs[i] = hs[i - 1]; hd[i] = hd[i - 1]; }
      hs[0] = src_pos; hd[0] = dst_bits;
    }
    const uint32_t wb = (dst_bits + 31u) >> 5;               // words above the current partial word are stale
    {
      const uint32_t z0 = (wb + (uint32_t)lane) & RING_MASK;
      ring[z0] = 0;
      if (z0 < 2u) ring[z0 + RING_WORDS] = 0;                // (the copies, see RING_ALLOC)
      if (lane < 2) { const uint32_t z1 = (wb + 64u + (uint32_t)lane) & RING_MASK; ring[z1] = 0; if (z1 < 2u) ring[z1 + RING_WORDS] = 0; }
    }
    wave_sync();
    const uint32_t i0 = src_pos + 4u * (uint32_t)lane;
    uint32_t val = 0, nb = 0;
    // the lane's four bytes (i0 is a multiple of 4; the bytes behind the MagSgn part -- MEL, VLC, the next block, or the slack
    // every data buffer ends with -- are read along and masked) and the four before them: the neighbour lane's, by a DPP
    // move; lane 0 fetches its own
    const uint32_t word = i0 < ms_len ? load_u32_unaligned(cb + i0) : 0u;
    uint32_t pw = from_prev(word);
    if (lane == 0) pw = (i0 && i0 < ms_len) ? load_u32_unaligned(cb + i0 - 4) : 0u;
    // (both loads are awaited HERE on every path: a load left pending behind a skipped branch makes the compiler wait for
    // "everything outstanding" -- the sample stores included -- at the next write of its register, in the middle of the row loop)
    asm volatile("" :: "v"(word), "v"(pw));
    if (i0 < ms_len) {
      const uint32_t cnt = min(4u, ms_len - i0);
      const uint32_t valid = cnt == 4u ? 0xFFFFFFFFu : (1u << (8u * cnt)) - 1u;
      // all four bytes at once: byte k of P / P2 is the raw byte k-1 / k-2 of the stream
      const uint32_t P = (word << 8) | (pw >> 24), P2 = (word << 16) | (pw >> 16);
      auto is_ff = [](uint32_t x) { return ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u; };   // 0x80 in every byte that is 0xFF
      const uint32_t S = is_ff(P) & valid;                          // 0x80 in byte k: byte k follows an 0xFF and carries 7 bits
      const uint32_t stray = (is_ff(P2) >> 7) & (P >> 7) & 0x01010101u;     // MSB of a 7-bit byte, OR-ed onto the next byte's LSB (see flatten)
      uint32_t x = ((word | stray) & valid) & ~S;                 // the bits that count, still in byte positions
      // close the gaps: a 7-bit byte hands every bit above it one position down (top byte: nothing above)
      x = (S & 0x00800000u) ? (x & 0x007FFFFFu) | ((x >> 1) & 0xFF800000u) : x;
      x = (S & 0x00008000u) ? (x & 0x00007FFFu) | ((x >> 1) & 0xFFFF8000u) : x;
      x = (S & 0x00000080u) ? (x & 0x0000007Fu) | ((x >> 1) & 0xFFFFFF80u) : x;
      val = x;
      nb = 8u * cnt - (uint32_t)__popc(S);
    }
    const uint32_t incl = wave_incl_scan(nb);
    const uint32_t pos = dst_bits + incl - nb;
    if (nb) {
      const uint32_t w = pos >> 5, sh = pos & 31;
      const uint32_t wa = w & RING_MASK, wb2 = (w + 1) & RING_MASK;
      atomicOr(&ring[wa], val << sh);
      if (wa < 2u) atomicOr(&ring[wa + RING_WORDS], val << sh);
      if (sh + nb > 32) { atomicOr(&ring[wb2], val >> (32 - sh)); if (wb2 < 2u) atomicOr(&ring[wb2 + RING_WORDS], val >> (32 - sh)); }
    }
    dst_bits += rdlane(incl, 63);
    src_pos += 256;
    wave_sync();
  };

  const uint32_t mmsbp2 = missing_msbs + 2;
  const uint32_t shift = 31 - K;
  const float delta = d.delta;
  const uint32_t half = (uint32_t)lane & 1u;
  bool bad = false;
  uint32_t e_prev = 0;                                   // exponent of this column's bottom sample, row above
  if (KEEP && prepare) {                                 // fill the ring as the first row would, and leave the state of "nothing decoded yet"
    while (src_pos < ms_len && dst_bits < ROW_BITS_MAX + 64u) unstuff_chunk();
    if (lane < 16) state[lane] = 0u;
    if (lane == 0) { state[16] = 0u; state[17] = src_pos; state[18] = dst_bits; state[19] = 0u; state[20] = ms_len; }
    wave_sync();
    return;
  }
  if (SLICED && (KEEP || qy_begin > 0)) {                // take over where the previous slice's worker stopped
    if (!KEEP && rdfirst(state[19]) != 0u) return;
    e_prev = (state[(uint32_t)lane >> 2] >> (8u * ((uint32_t)lane & 3u))) & 0xFFu;
    mpos = rdfirst(state[16]); src_pos = rdfirst(state[17]); dst_bits = rdfirst(state[18]);
    hs[0] = src_pos; hd[0] = dst_bits;
  }
  uint32_t ent_next = 0;
  if (!SLICED) {
    ent_next = (uint32_t)lane < W && qy_begin < qy_last ? rec_at(qy_begin, (uint32_t)lane >> 1) : 0u;      // records are fetched one step ahead
    asm volatile("" : "+v"(ent_next));                // the first record is awaited here, not inside the loop (see the note at the stores)
  }
  for (uint32_t qy = qy_begin; qy < qy_last && !bad; ++qy) {
    const uint8_t* vexp = s_exp_w + (qy & 1) * EXP_BYTES;   // wide blocks: exponents of the sample row above (+1 offset)
    uint8_t* vnew = s_exp_w + ((qy & 1) ^ 1) * EXP_BYTES;
    for (uint32_t c0 = 0; c0 < (WD == 1 ? 1u : W); c0 += 64) {   // (WD 1: one pass, col = lane -- known at compile time)
      while (!(S2_ABL & 16) && src_pos < ms_len && (int32_t)(dst_bits - mpos) < (int32_t)(ROW_BITS_MAX + 64u)) unstuff_chunk();   // (a resumed slice starts below mpos)
      const bool exhausted = src_pos >= ms_len;            // then bits at and beyond dst_bits read as 1 (:609-632)
      const uint32_t col = c0 + (uint32_t)lane;
      const bool act = col < W;
      const uint32_t qx = col >> 1;
      const uint32_t ent = SLICED ? ents[0] : ent_next;
      if (SLICED) {
#pragma unroll
        for (uint32_t i = 0; i + 1 < (SLICED ? S2_ROWS : 1u); ++i) ents[i] = ents[i + 1];
      } else {
        uint32_t nc0 = c0 + 64, nqy = qy;
        if (nc0 >= W) { nc0 = 0; nqy = qy + 1; }
        const uint32_t ncol = nc0 + (uint32_t)lane;
        ent_next = (nqy < qy_last && ncol < W) ? rec_at(nqy, ncol >> 1) : 0u;      // (never beyond the slice: those records may not exist yet)
      }
      // SLICED: ent = packed9 | u << 9 (ht_tables.h: dec_vlc32), 0 for an idle lane
      const uint32_t inf = act ? (ent & 0xFFFFu) : 0u;
      uint32_t U_q = SLICED ? inf >> 9 : ent >> 16;
      if (qy > 0) {
        uint32_t gamma = SLICED ? inf & 0x100u : inf & 0xF0u;
        if (!SLICED) gamma &= gamma - 0x10u;                                            // :1218
        uint32_t em;                      // max exponent over columns 2qx-1 .. 2qx+2 of the sample row above
        if (S2_ABL & 8) em = e_prev;
        else if (!wide) {
          // even lane 2k: max(e[2k-1], e[2k]); odd lane 2k+1: max(e[2k+1], e[2k+2]); then the pair's two halves together
          const uint32_t e_nx = from_next(e_prev), e_pv = from_prev(e_prev);   // both moves run with every lane enabled
          const uint32_t hm = max(e_prev, half ? e_nx : e_pv);
          em = max(hm, from_pair(hm));
        } else {
          const uint32_t b = 2 * qx;
          em = act ? max(max((uint32_t)vexp[b], (uint32_t)vexp[b + 1]), max((uint32_t)vexp[b + 2], (uint32_t)vexp[b + 3])) : 0u;
        }
        U_q += gamma ? max(em, 1u) : 1u;                                                // :1219-1223
      }
      // (an idle lane's record is 0: its U_q is 0 or 1, never above missing_msbs + 2 >= 2 -- no "act &&" needed)
      if (__ballot(U_q > mmsbp2) != 0ull) { bad = true; break; }                        // :1114,:1224
      // this lane's samples: n0 = (col, 2qy), n1 = (col, 2qy+1); quad bits 2*half and 2*half+1
      // sg*: the sample is significant, ek* / eb*: its e_k / e_1 bit
      const uint32_t sel = SLICED ? inf >> (4u * half) : inf >> (2u * half);
      const uint32_t st0 = sel & 3u, st1 = (sel >> 2) & 3u;                  // (SLICED: the samples' two-bit states)
      const bool sg0 = SLICED ? st0 != 0u : (sel & 0x10u) != 0u, sg1 = SLICED ? st1 != 0u : (sel & 0x20u) != 0u;
      const uint32_t ek0 = SLICED ? st0 >> 1 : (sel >> 12) & 1u, ek1 = SLICED ? st1 >> 1 : (sel >> 13) & 1u;
      const uint32_t eb0 = SLICED ? st0 & (st0 >> 1) : (sel >> 8) & 1u, eb1 = SLICED ? st1 & (st1 >> 1) : (sel >> 9) & 1u;
      const uint32_t m0 = sg0 ? U_q - ek0 : 0u;
      const uint32_t m1 = sg1 ? U_q - ek1 : 0u;
      const uint32_t tot = m0 + m1;
      const uint32_t incl = (S2_ABL & 2) ? tot * ((uint32_t)lane + 1u) : wave_incl_scan(tot);
      const uint32_t at = mpos + incl - tot;
      mpos += (S2_ABL & 2) ? 64u : rdlane(incl, 63);
      const uint32_t wi = at >> 5, sh = at & 31;
      const uint32_t* rw = ring + (wi & RING_MASK);
      const uint32_t w0 = (S2_ABL & 4) ? at : rw[0], w1 = (S2_ABL & 4) ? at * 3u : rw[1], w2 = (S2_ABL & 4) ? at * 5u : rw[2];   // (rw[1], rw[2] may be the copies behind the ring)
      uint64_t win = (uint64_t)__funnelshift_r(w0, w1, sh) | ((uint64_t)__funnelshift_r(w1, w2, sh) << 32);
      if (exhausted) {
        if (at >= dst_bits) win = ~0ull;
        else if (at + 64 > dst_bits) win |= ~0ull << (dst_bits - at);
      }
      // both samples without branches: m0 / m1 are 0 for an insignificant sample, its value is masked away at the end
      uint32_t out0, out1, v1;
      {
        const uint32_t ms_val = (uint32_t)win;
        uint32_t v_n = ms_val & ((1u << m0) - 1u);                                      // :1127-1133
        v_n |= eb0 << m0;
        v_n |= 1u;
        const uint32_t val = ((ms_val << 31) | ((v_n + 2u) << (p - 1))) & (sg0 ? 0xFFFFFFFFu : 0u);
        out0 = raw_out ? val : dequantise(val, rev, shift, delta);
      }
      {
        const uint32_t ms_val = __builtin_amdgcn_alignbit((uint32_t)(win >> 32), (uint32_t)win, m0);   // (uint32_t)(win >> m0): m0 <= U_q <= 31
        uint32_t v_n = ms_val & ((1u << m1) - 1u);
        v_n |= eb1 << m1;
        v_n |= 1u;
        const uint32_t keep = sg1 ? 0xFFFFFFFFu : 0u;
        const uint32_t val = ((ms_val << 31) | ((v_n + 2u) << (p - 1))) & keep;
        v1 = v_n & keep;
        out1 = raw_out ? val : dequantise(val, rev, shift, delta);
      }
      const uint32_t e_new = v1 ? 31u - (uint32_t)__clz((int)v1) : 0u;
      if (!wide) e_prev = e_new;
      else if (act) vnew[col + 1] = (uint8_t)e_new;
      // The next step's record (requested at the top of this step) is taken into its register HERE, before this
      // step's stores are issued: loads and stores share one in-order counter on gfx9, so a wait placed after the
      // stores (where the compiler puts it: at the loop top) would also wait for the stores to reach L2.
      if (!SLICED) asm volatile("" : "+v"(ent_next));
      if (act && (!(S2_ABL & 1) || (out0 ^ out1) == 0x9E3779B9u)) {
        const uint32_t y = 2 * qy;
        // (a scalar row address + the lane's 32-bit byte offset: the store takes them as they are, no 64-bit vector add)
        char* rowp = reinterpret_cast<char*>(dst + (size_t)y * pitch);
        char* rowp1 = rowp + (size_t)pitch * 4u;
        uint32_t coff = col * 4u;
        asm volatile("" : "+v"(coff));                 // (kept a 32-bit offset HERE: hoisted out of the loop it becomes a 64-bit vector add per store)
        *reinterpret_cast<uint32_t*>(rowp + coff) = out0;
        if (y + 1 < H) { asm volatile("" : "+v"(coff)); *reinterpret_cast<uint32_t*>(rowp1 + coff) = out1; }
      }
    }
    if (wide) wave_sync();
  }
  if (bad) { zero_block(); if (lane == 0) block_status[bi] = 1; }
  if (SLICED && qy_last < QH) {                          // hand over to the next slice
    if (bad) { if (lane == 0) state[19] = 1u; wave_sync(); return; }
    // the un-stuffer restarts at the latest chunk that begins at or below the first bit still to be decoded
    // (KEEP: the block has a ring of its own, the next slice goes on exactly where this one stopped)
    uint32_t rs = hs[4], rd = hd[4];
#pragma unroll
    for (int i = 3; i >= 0; --i) if (hd[i] <= mpos) { rs = hs[i]; rd = hd[i]; }
    if (KEEP) { rs = src_pos; rd = dst_bits; }
    const uint32_t e4 = e_prev | (from_next(e_prev) << 8);                    // four columns' exponents into one word
    const uint32_t e8 = e4 | ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e4, 0x102, 0xF, 0xF, false) << 16);   // row_shl:2
    if (((uint32_t)lane & 3u) == 0) state[(uint32_t)lane >> 2] = e8;
    if (lane == 0) { state[16] = mpos; state[17] = rs; state[18] = rd; state[19] = 0u; }
    wave_sync();
  }
}

template <int TX, int WD>
__global__ __launch_bounds__(64 * WAVES) void ht_dec_step2_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    const uint32_t* __restrict__ quads, uint32_t* __restrict__ coef, uint8_t* __restrict__ block_status)
{
  __shared__ uint32_t s_ring[WAVES][RING_ALLOC];
  __shared__ uint8_t s_exp[WAVES][2][EXP_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and the compiler knows it
  // Workgroup k of a launch runs on XCD k % 8, each XCD with its own L2.  The per-quad records of 64
  // consecutive blocks share their cache lines (pair-major interleave, see ojphgpu_ht_decode_layout), so
  // consecutive blocks are given to ONE XCD: XCD x works through the x-th contiguous eighth of the blocks.
  const uint32_t q8 = gridDim.x >> 3, r8 = gridDim.x & 7u, xcd = blockIdx.x & 7u;
  const uint32_t wg = xcd * q8 + (xcd < r8 ? xcd : r8) + (blockIdx.x >> 3);
  const uint32_t bi = wg * WAVES + wave;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (d.reversible & 4u) return;                          // 64-bit sample path: ht_dec64_step2_kernel
  step2_block<TX, WD, false>(d, bi, data, quads, coef, block_status, s_ring[wave], &s_exp[wave][0][0], lane, 0u, 0xFFFFFFFFu, nullptr);
}


// -------------------------------------------------------------------------------------------------
// step 2 with TWO code-blocks of at most 32 columns, or FOUR of at most 16, to a wavefront (the IMF profiles' 32 x 32 blocks)
// -------------------------------------------------------------------------------------------------
// step2_block maps the columns of ONE block onto the lanes: with blocks of 32 columns half of the wavefront idles through
// every row.  Here the wavefront is cut into NB segments of LPB = 64 / NB lanes, segment k holding the columns of block
// bA + k: one pass over the quad rows decodes them all.  What is wave-uniform in step2_block -- the MagSgn positions, the
// un-stuffer's state, the verdict -- is uniform per SEGMENT here (vector registers whose LPB lanes agree), the prefix sums stop
// at the segment's end, each segment has its ring.  Launches whose blocks are ALL at most 32 (16) columns wide, without
// refinement passes, take this kernel (ht_decode_step2_launch, kinds bits 6 and 7 clear).  Per block the arithmetic is
// step2_block's: ojph_block_decoder32.cpp:1091-1316.

// inclusive prefix sum inside each segment of LPB lanes (32: lanes 0..31, 32..63; 16: the four DPP rows)
template <int LPB>
__device__ __forceinline__ uint32_t seg_incl_scan(uint32_t v)
{
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
  if (LPB == 32) x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3: the second 16 lanes of each half
  return (uint32_t)x;
}
// the value lane LPB - 1 of the lane's own segment holds (a segment's total after seg_incl_scan)
template <int LPB>
__device__ __forceinline__ uint32_t seg_last(uint32_t v, uint32_t seg)
{
  if (LPB == 32) { const uint32_t a = rdlane(v, 31), b = rdlane(v, 63); return seg ? b : a; }
  const uint32_t a = rdlane(v, 15), b = rdlane(v, 31), c = rdlane(v, 47), d = rdlane(v, 63);
  return seg & 2u ? (seg & 1u ? d : c) : (seg & 1u ? b : a);
}
// the part of a ballot that belongs to the lane's segment
template <int LPB>
__device__ __forceinline__ uint32_t seg_bits(uint64_t m, uint32_t seg)
{
  return (uint32_t)(m >> (seg * (uint32_t)LPB)) & (LPB == 32 ? 0xFFFFFFFFu : 0xFFFFu);
}

template <int TX, int NB>
__device__ __forceinline__ void step2_multi(const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, uint32_t bA,
                                            const uint8_t* __restrict__ data, const uint32_t* __restrict__ quads,
                                            uint32_t* __restrict__ coef, uint8_t* __restrict__ block_status, uint32_t* rings, int lane)
{
  static_assert(TX == 1 || TX == 2, "one wavelet per launch, no refinement passes");
  static_assert(NB == 2 || NB == 4, "two blocks of 32 columns or four of 16");
  constexpr int LPB = 64 / NB;
  constexpr uint32_t ROW_BITS = (uint32_t)LPB * 2u * 32u;        // a quad row of LPB columns consumes at most this many bits
  const uint32_t seg = (uint32_t)lane / (uint32_t)LPB;           // the lane's segment: block bA + seg
  const uint32_t ll = (uint32_t)lane & (uint32_t)(LPB - 1);      // = the lane's column in its block
  const uint32_t bi = bA + seg;
  const bool have = bi < n;
  // the blocks' descriptors arrive as scalars; every lane keeps its segment's
  uint32_t W, H, pitch, lcup, npass, missing_msbs, K, rec_off, flags;
  float delta; uint64_t coef_off, data_off;
  {
    const ojphgpu_cb_desc d0 = blocks[bA], d1 = blocks[bA + 1u < n ? bA + 1u : bA];
#define DSEL2(f) (seg & 1u ? d1.f : d0.f)
    if (NB == 2) {
      W = DSEL2(w); H = DSEL2(h); pitch = DSEL2(pitch); lcup = DSEL2(len1); npass = DSEL2(num_passes); missing_msbs = DSEL2(missing_msbs);
      K = DSEL2(K_max); rec_off = DSEL2(scratch_cap); flags = DSEL2(reversible); delta = DSEL2(delta); coef_off = DSEL2(coef_off); data_off = DSEL2(data_off);
    } else {
      const ojphgpu_cb_desc d2 = blocks[bA + 2u < n ? bA + 2u : bA], d3 = blocks[bA + 3u < n ? bA + 3u : bA];
#define DSEL4(f) (seg & 2u ? (seg & 1u ? d3.f : d2.f) : DSEL2(f))
      W = DSEL4(w); H = DSEL4(h); pitch = DSEL4(pitch); lcup = DSEL4(len1); npass = DSEL4(num_passes); missing_msbs = DSEL4(missing_msbs);
      K = DSEL4(K_max); rec_off = DSEL4(scratch_cap); flags = DSEL4(reversible); delta = DSEL4(delta); coef_off = DSEL4(coef_off); data_off = DSEL4(data_off);
#undef DSEL4
    }
#undef DSEL2
  }
  const bool rev = TX == 1;
  const uint32_t QW = (W + 1u) >> 1, QH = (H + 1u) >> 1, PW = (QW + 1u) >> 1;
  uint32_t* ring = rings + seg * RING_ALLOC;
  uint32_t* dst = coef + coef_off;
  const uint8_t* cb = data + data_off;
  const uint32_t* rec = quads + rec_off;
  const bool exists = have && W != 0u && H != 0u && W <= (uint32_t)LPB && !(flags & 4u);
  const bool coded = exists && lcup != 0u && npass != 0u;
  const uint32_t p = 30u - missing_msbs, mmsbp2 = missing_msbs + 2u, shift = 31u - K;

  auto seg_max = [&](uint32_t v) -> uint32_t {                   // the largest value any segment holds (v uniform per segment)
    uint32_t m = max(rdlane(v, 0), rdlane(v, LPB));
    if (NB == 4) m = max(m, max(rdlane(v, 2 * LPB), rdlane(v, 3 * LPB)));
    return m;
  };
  auto zero_block = [&](bool which) {                            // (mem_clear, ojph_codeblock.cpp:247) by the lanes of the block's segment
    const uint32_t hmax = seg_max(which ? H : 0u);               // (every segment walks the tallest block's rows)
    for (uint32_t y = 0; y < hmax; ++y)
      if (which && y < H && ll < W) dst[(size_t)y * pitch + ll] = 0u;
  };

  // the verdict of step 1 and the length of the MagSgn part (a one-byte segment has failed in step 1; its bytes are not used)
  uint32_t st = 0, ms_len = 0;
  if (coded) {
    st = block_status[bi];
    const uint32_t b1 = cb[lcup - 1u], b2 = cb[lcup >= 2u ? lcup - 2u : 0u];
    ms_len = lcup - ((b1 << 4) + (b2 & 0xFu));
  }
  bool go = coded && st == 0u;                                   // this segment decodes
  {
    const bool z = exists && !go;                                // not coded, or failed in step 1: the block is zero
    if (__ballot(z) != 0ull) zero_block(z);
  }
  if (__ballot(go) == 0ull) return;
  for (uint32_t i = ll; i < RING_ALLOC; i += (uint32_t)LPB) ring[i] = 0;
  wave_sync();

  // ---- un-stuffs the next 4 LPB MagSgn bytes of the segments whose lanes say `need` into their rings (step2_block's rule, :609-653) ----
  uint32_t dst_bits = 0, src_pos = 0, mpos = 0;                  // per segment: bits un-stuffed, bytes consumed, bits decoded
  auto unstuff = [&](bool need) {
    {
      const uint32_t wb = (dst_bits + 31u) >> 5;                 // words above the current partial word are stale
      const uint32_t z0 = (wb + ll) & RING_MASK;
      if (need) { ring[z0] = 0; if (z0 < 2u) ring[z0 + RING_WORDS] = 0; }
      if (need && ll < 2u) { const uint32_t z1 = (wb + (uint32_t)LPB + ll) & RING_MASK; ring[z1] = 0; if (z1 < 2u) ring[z1 + RING_WORDS] = 0; }
    }
    wave_sync();
    const uint32_t i0 = src_pos + 4u * ll;
    const bool in = need && i0 < ms_len;
    const uint32_t word = in ? load_u32_unaligned(cb + i0) : 0u;
    uint32_t pw = from_prev(word);
    if (ll == 0u) pw = (in && i0) ? load_u32_unaligned(cb + i0 - 4) : 0u;
    asm volatile("" :: "v"(word), "v"(pw));                      // (both loads awaited here on every path, see step2_block)
    uint32_t val = 0, nb = 0;
    if (in) {
      const uint32_t cnt = min(4u, ms_len - i0);
      const uint32_t valid = cnt == 4u ? 0xFFFFFFFFu : (1u << (8u * cnt)) - 1u;
      const uint32_t P = (word << 8) | (pw >> 24), P2 = (word << 16) | (pw >> 16);
      auto is_ff = [](uint32_t x) { return ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u; };
      const uint32_t S = is_ff(P) & valid;
      const uint32_t stray = (is_ff(P2) >> 7) & (P >> 7) & 0x01010101u;
      uint32_t x = ((word | stray) & valid) & ~S;
      x = (S & 0x00800000u) ? (x & 0x007FFFFFu) | ((x >> 1) & 0xFF800000u) : x;
      x = (S & 0x00008000u) ? (x & 0x00007FFFu) | ((x >> 1) & 0xFFFF8000u) : x;
      x = (S & 0x00000080u) ? (x & 0x0000007Fu) | ((x >> 1) & 0xFFFFFF80u) : x;
      val = x;
      nb = 8u * cnt - (uint32_t)__popc(S);
    }
    const uint32_t incl = seg_incl_scan<LPB>(nb);
    const uint32_t pos = dst_bits + incl - nb;
    if (nb) {
      const uint32_t w = pos >> 5, sh = pos & 31u;
      const uint32_t wa = w & RING_MASK, wb2 = (w + 1u) & RING_MASK;
      atomicOr(&ring[wa], val << sh);
      if (wa < 2u) atomicOr(&ring[wa + RING_WORDS], val << sh);
      if (sh + nb > 32u) { atomicOr(&ring[wb2], val >> (32u - sh)); if (wb2 < 2u) atomicOr(&ring[wb2 + RING_WORDS], val >> (32u - sh)); }
    }
    const uint32_t total = seg_last<LPB>(incl, seg);
    if (need) { dst_bits += total; src_pos += 4u * (uint32_t)LPB; }
    wave_sync();
  };

  const uint32_t half = ll & 1u;
  const bool edgeL = ll == 0u, edgeR = ll == (uint32_t)(LPB - 1);
  const bool col_in = ll < W;
  auto rec_at = [&](uint32_t qy_) -> uint32_t {
    return *(rec + (size_t)(qy_ * PW + (ll >> 2)) * REC_STRIDE + ((ll >> 1) & 1u));
  };
  uint32_t e_prev = 0;                                           // exponent of this column's bottom sample, row above
  bool failed = false;                                           // this segment met a row it cannot decode (:1114, :1224)
  uint32_t ent_next = (go && col_in) ? rec_at(0) : 0u;           // records are fetched one step ahead
  asm volatile("" : "+v"(ent_next));
  const uint32_t qy_end = seg_max(go ? QH : 0u);
  for (uint32_t qy = 0; qy < qy_end; ++qy) {
    bool rowon = go && qy < QH;
    if (__ballot(rowon) == 0ull) break;
    for (;;) {
      const bool need = rowon && src_pos < ms_len && (int32_t)(dst_bits - mpos) < (int32_t)(ROW_BITS + 64u);
      if (__ballot(need) == 0ull) break;
      unstuff(need);
    }
    const bool exhausted = src_pos >= ms_len;                    // then bits at and beyond dst_bits read as 1 (:609-632)
    const bool act = rowon && col_in;
    const uint32_t ent = ent_next;
    ent_next = (go && col_in && qy + 1u < QH) ? rec_at(qy + 1u) : 0u;
    const uint32_t inf = act ? (ent & 0xFFFFu) : 0u;
    uint32_t U_q = act ? ent >> 16 : 0u;
    if (qy > 0) {
      uint32_t gamma = inf & 0xF0u;
      gamma &= gamma - 0x10u;                                                            // :1218
      // even lane 2k: max(e[2k-1], e[2k]); odd lane 2k+1: max(e[2k+1], e[2k+2]); then the pair's two halves together --
      // the columns beside a block's first and last one do not exist (another block's lanes sit there)
      const uint32_t e_nx = edgeR ? 0u : from_next(e_prev), e_pv = edgeL ? 0u : from_prev(e_prev);
      const uint32_t hm = max(e_prev, half ? e_nx : e_pv);
      const uint32_t em = max(hm, from_pair(hm));
      U_q += gamma ? max(em, 1u) : 1u;                                                   // :1219-1223
      if (!act) U_q = 0u;
    }
    if (seg_bits<LPB>(__ballot(act && U_q > mmsbp2), seg) != 0u) { failed = true; go = false; rowon = false; }   // :1114, :1224
    const bool actr = rowon && col_in;
    const uint32_t sel = inf >> (2u * half);
    const bool sg0 = actr && (sel & 0x10u) != 0u, sg1 = actr && (sel & 0x20u) != 0u;
    const uint32_t ek0 = (sel >> 12) & 1u, ek1 = (sel >> 13) & 1u;
    const uint32_t eb0 = (sel >> 8) & 1u, eb1 = (sel >> 9) & 1u;
    const uint32_t m0 = sg0 ? U_q - ek0 : 0u;
    const uint32_t m1 = sg1 ? U_q - ek1 : 0u;
    const uint32_t tot = m0 + m1;
    const uint32_t incl = seg_incl_scan<LPB>(tot);
    const uint32_t at = mpos + incl - tot;
    mpos += seg_last<LPB>(incl, seg);
    const uint32_t wi = at >> 5, sh = at & 31u;
    const uint32_t* rw = ring + (wi & RING_MASK);
    const uint32_t w0 = rw[0], w1 = rw[1], w2 = rw[2];           // (rw[1], rw[2] may be the copies behind the ring)
    uint64_t win = (uint64_t)__funnelshift_r(w0, w1, sh) | ((uint64_t)__funnelshift_r(w1, w2, sh) << 32);
    if (exhausted) {
      if (at >= dst_bits) win = ~0ull;
      else if (at + 64u > dst_bits) win |= ~0ull << (dst_bits - at);
    }
    uint32_t out0, out1, v1;
    {
      const uint32_t ms_val = (uint32_t)win;
      uint32_t v_n = ms_val & ((1u << m0) - 1u);                                         // :1127-1133
      v_n |= eb0 << m0;
      v_n |= 1u;
      const uint32_t val = ((ms_val << 31) | ((v_n + 2u) << (p - 1u))) & (sg0 ? 0xFFFFFFFFu : 0u);
      out0 = dequantise(val, rev, shift, delta);
    }
    {
      const uint32_t ms_val = __builtin_amdgcn_alignbit((uint32_t)(win >> 32), (uint32_t)win, m0);   // (uint32_t)(win >> m0): m0 <= U_q <= 31
      uint32_t v_n = ms_val & ((1u << m1) - 1u);
      v_n |= eb1 << m1;
      v_n |= 1u;
      const uint32_t keep = sg1 ? 0xFFFFFFFFu : 0u;
      const uint32_t val = ((ms_val << 31) | ((v_n + 2u) << (p - 1u))) & keep;
      v1 = v_n & keep;
      out1 = dequantise(val, rev, shift, delta);
    }
    e_prev = v1 ? 31u - (uint32_t)__clz((int)v1) : 0u;
    asm volatile("" : "+v"(ent_next));                           // the next record is taken into its register before this row's stores (see step2_block)
    if (actr) {
      const uint32_t y = 2u * qy;
      uint32_t* o = dst + (size_t)y * pitch + ll;
      o[0] = out0;
      if (y + 1u < H) o[pitch] = out1;
    }
  }
  if (__ballot(failed) != 0ull) {
    zero_block(failed);
    if (failed && ll == 0u) block_status[bi] = 1;
  }
}

template <int TX, int NB>
__global__ __launch_bounds__(64 * WAVES) void ht_dec_step2_multi_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    const uint32_t* __restrict__ quads, uint32_t* __restrict__ coef, uint8_t* __restrict__ block_status)
{
  __shared__ uint32_t s_ring[WAVES][NB * RING_ALLOC];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // consecutive blocks to ONE XCD, as in ht_dec_step2_kernel (their records share cache lines)
  const uint32_t q8 = gridDim.x >> 3, r8 = gridDim.x & 7u, xcd = blockIdx.x & 7u;
  const uint32_t wg = xcd * q8 + (xcd < r8 ? xcd : r8) + (blockIdx.x >> 3);
  const uint32_t bA = (uint32_t)NB * (wg * WAVES + wave);
  if (bA >= n) return;
  step2_multi<TX, NB>(blocks, n, bA, data, quads, coef, block_status, s_ring[wave], lane);
}


// -------------------------------------------------------------------------------------------------
// step 1 and step 2 in ONE launch: chains first, step-2 workers behind them, slice by slice
// -------------------------------------------------------------------------------------------------
// Step 1 is a latency floor: the serial chain of a code-block takes 0.2 ms however many blocks there are, and while the
// 388 chain wavefronts of an 8K frame run, the rest of the chip has nothing to do -- step 2 of a block needs that block's
// records.  But it needs them ROW BY ROW.  The fused kernel puts both into one launch:
//   * n1 workgroups are step 1 as in ht_dec_step1_raw_kernel (chain + two partner wavefronts per 64 blocks); a chain
//     wavefront stores its records with agent scope (16 bits per quad, a row's worth staged in LDS: flush_row16) and
//     publishes, at the end of every slice of quad rows, how many rows of its 64 blocks are complete (flag per chain
//     wavefront; s_waitcnt vmcnt(0) first: the chain has no loads, its stores are all vmcnt counts).  The chain wavefronts
//     run at priority 3, their partners at 2: a SIMD serves its wavefronts oldest first, and a partner behind three older
//     worker wavefronts kept its chain waiting for VLC words (the workgroups that happened to come second on their CU ran a
//     third slower than the rest, and the launch ends with its slowest chain);
//   * the other workgroups are persistent step-2 worker wavefronts: a wavefront owns `per_wave` blocks (as few as the
//     chip's wavefront slots allow: 5 at 8K), `nwaves` apart in the block order so that every wavefront gets the same mix
//     of bands, and takes them slice by slice -- slice 0 of each of its blocks, then slice 1, ... -- waiting (bounded)
//     until the chain wavefront of a block has published the slice's rows.  A slice is S2_ROWS quad rows of a block, the
//     last S2_ROWS rows of the tallest block as 4 + 2 + 2 (SliceSched), decoded by step2_block<SLICED>; what a block's
//     next slice needs stays in the wavefront's LDS: 80 bytes of state (bottom-row exponents, MagSgn position, where the
//     un-stuffer stands) and, with at most S2_RINGS blocks per wavefront, the block's own ring of un-stuffed MagSgn bits,
//     cleared and filled BEFORE the first wait (otherwise one ring per wavefront and a restart of the un-stuffer at the
//     latest 256-byte boundary per slice).  The workers take priorities 0 and 1 in turns, the younger the wavefront the
//     larger its share of turns at 1: left to the oldest-first rule, slot 0 of a SIMD was through 0.06 ms before slot 5.
// Who plays which role is decided by TICKETS, not by blockIdx.  A workgroup asks where it runs (HW_REG_XCC_ID, HW_REG_HW_ID)
// and marks its COMPUTE UNIT with the run's epoch.  The first workgroup of the run on a CU takes a step-1 number while there
// are any (n1 of them; the launch is not used with more step-1 workgroups than CUs); everyone else is a worker and takes its
// number from the workers' counter.  Step-1 numbers are therefore taken as CUs receive their first workgroup -- one chain
// workgroup to a CU -- and a worker can only be waiting for a workgroup that any CU the run has not touched yet is enough
// to start, whatever order the dispatchers hand workgroups out in and whatever else shares the chip (other streams'
// launches, a second decoder object's fused launch): no deadlock, no dependence on dispatch order or on which XCD a
// workgroup lands on.
// (Measured forms, chains alone / whole launch, 8K frame: roles by blockIdx 0.21 / 0.31 ms.  ONE counter and the first n1
// to come play step 1, the first form of the tickets: 0.29 / 0.36 -- whose atomic arrives first is decided by distance to the
// counter's memory channel, the first 97 of 509 workgroups sat on four of the eight XCDs, 48 of them on one
// (tools/micro/xcc_probe.hip), two chain workgroups to a CU.  A counter per XCD with the XCD's share of the numbers: still
// 0.29 / 0.36 -- an XCD's 64 workgroups start within microseconds of each other, the first 13 to arrive are a random 13, on
// about ten CUs two chain workgroups share the SIMDs and the launch ends with its slowest chain; and a small launch whose
// workgroups miss an XCD -- workgroup i is NOT always on XCD i % 8: after other streams had been busy the same five-workgroup
// launch started elsewhere -- left that XCD's numbers untaken.  First on its CU, as here: 0.22 / 0.31.)
// Nothing is cleared between runs: the CU words hold the epoch of the last run that came by, the two counters
// (epoch << 32) | count -- epoch must GROW from run to run on a scratch.
// The waits are bounded all the same, by TIME (s_memrealtime, 100 MHz; two seconds unless the host says otherwise): a wait
// that runs out does not fail its block -- the worker writes the run's epoch into the RETRY word behind the block status
// array, and the host, when it collects the verdicts of the run, decodes the frame again through the separate step 1 / step 2
// launches (ojphgpu_codec.cpp: fused_retry).  The same goes for a chain that gives up on its partner, and it is what would
// happen on a device whose XCDs do not each start their share of the workgroups (a step-1 number nobody takes).
// The XCDs' L2s are not coherent with each other: everything exchanged inside the launch (records, flags, block status)
// is accessed with agent scope.  Flags carry the run's epoch, so nothing has to be cleared between runs.  Blocks wider
// than 64 samples and blocks with refinement passes keep the separate launches, and so do frames where the one launch
// does not pay (ht_decode_fused_pays).

// waits until the chain wavefront behind `flag` has published `rows` quad rows of this run; returns how many it has
// published by then (0 = the wait ran out: `ticks` of the 100 MHz clock)
__device__ __forceinline__ uint32_t wait_rows(const uint32_t* flag, uint32_t epoch, uint32_t rows, uint32_t ticks)
{
  uint64_t t0 = 0;
  for (uint32_t spins = 0;; ++spins) {
    const uint32_t v = ld_agent(flag);
    if ((v >> 16) == (epoch & 0xFFFFu) && (v & 0xFFFFu) >= rows) return v & 0xFFFFu;
    __builtin_amdgcn_s_sleep(12);
    if ((spins & 63u) == 63u) {                               // the clock is looked at every 64th poll
      const uint64_t now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      else if (now - t0 > (uint64_t)ticks) return 0u;
    }
  }
}

// "decode this run again through the separate launches": the RETRY word behind the block status array (what whoever collects
// the run reads), and -- host_retry, a word of host memory the device can write, or null -- the same where the host sees it
// WITHOUT a copy: a caller that never collects its runs (ojphgpu_decoder_run_device and device-side consumers) is told by
// its next call (OJPHGPU_E_UNCOLLECTED).  Only ever executed when a wait has run out.
__device__ __forceinline__ void ask_for_repeat(uint32_t* retry, uint32_t* host_retry, uint32_t epoch)
{
  st_agent(retry, epoch);
  if (host_retry) __hip_atomic_store(host_retry, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifndef PRIO_PERIOD
#define PRIO_PERIOD 6u
#endif
// NR: un-stuffing rings per worker wavefront -- 1: one ring, every slice of a block re-un-stuffs from the latest chunk
// boundary below its first bit; > 1: a ring per block (per_wave <= NR), a slice goes on where the one before stopped.
template <int TX, int CH, int WGW, int NR>            // WGW wavefronts per workgroup: 3 CH of them work in the step-1 role, all in the worker role
__global__ __launch_bounds__(64 * WGW) __attribute__((amdgpu_waves_per_eu(6))) void ht_dec_fused_kernel(   // (two workgroups to a CU: six wavefronts per SIMD, 80 registers)
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    uint32_t* __restrict__ quads, uint32_t* __restrict__ coef, uint8_t* __restrict__ block_status,
    uint32_t* __restrict__ fstate, uint32_t n1, uint32_t per_wave, uint32_t max_qh, uint32_t epoch, uint32_t dbg,
    uint32_t ticket_off, uint32_t wait_ticks, uint32_t* __restrict__ host_retry)
{
  // one LDS area, carved per role: the step-1 role's tables, event strings, VLC rings and mailboxes -- or the workers'
  // un-stuffing rings and block states
  constexpr uint32_t EV_OFF = 2048 + 160, VR_OFF = EV_OFF + CH * EV_RING * 64, CTL_OFF = VR_OFF + CH * VR_WORDS * 64;
  constexpr uint32_t REC_OFF = CTL_OFF + CH * 5 * 64;                       // a row of pair words per chain wavefront (flush_row16)
  constexpr uint32_t CHAIN_WORDS = REC_OFF + CH * REC16_ROW_WORDS * 64;
  constexpr uint32_t WORKER_WORDS = NR * RING_ALLOC + S2_MAX_PER_WAVE * S2_STATE_WORDS;
  constexpr uint32_t LDS_WORDS = CHAIN_WORDS > WGW * WORKER_WORDS ? CHAIN_WORDS : WGW * WORKER_WORDS;
  __shared__ __attribute__((aligned(16))) uint32_t s_mem[LDS_WORDS];
  uint32_t* const s_vlc = s_mem;                            // dec_vlc32: 2 x 1024 entries
  uint16_t* const s_uvlc0 = reinterpret_cast<uint16_t*>(s_mem + 2048);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const SliceSched sched(max_qh);
  // the workgroup's number in START order (see above)
  uint32_t wgid = blockIdx.x;
  if (!(dbg & 8u)) {                                        // (dbg 8, timing experiment: roles by workgroup index)
    if (threadIdx.x == 0) {
      uint32_t xcc, hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      uint32_t* tk = fstate + ticket_off;
      // the run's number on a counter that is never cleared: the counter holds (epoch << 32) | count -- raised to this
      // run's epoch first (a no-op for all but the first to come), then counted on
      auto take = [&](uint32_t* c) -> uint32_t {
        unsigned long long* c64 = reinterpret_cast<unsigned long long*>(c);
        (void)__hip_atomic_fetch_max(c64, (unsigned long long)epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (uint32_t)__hip_atomic_fetch_add(c64, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      // the first workgroup of this run on its compute unit (CU_ID, SH_ID, SE_ID within the XCD) finds another run's epoch there
      const bool first = __hip_atomic_exchange(tk + (xcc & 7u) * 256u + ((hw >> 8) & 0xFFu), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch;
      uint32_t role = 0xFFFFFFFFu;
      if (first) {                                          // a step-1 workgroup, while there are step-1 numbers left
        const uint32_t t = take(tk + TICKET_CU_WORDS);
        if (t < n1) role = t;
      }
      if (role == 0xFFFFFFFFu) role = n1 + take(tk + TICKET_CU_WORDS + TICKET_STRIDE);   // a worker
      s_mem[0] = role;
    }
    __syncthreads();
    wgid = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_mem[0]);
    __syncthreads();
  }
  uint32_t* const retry = reinterpret_cast<uint32_t*>(block_status + ((n + 3u) & ~3u));   // != the run's epoch: nothing to repeat
#ifdef FUSED_TIMELINE
  uint32_t tl_where;
  {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    tl_where = (xcc << 16) | (hw & 0xFFFFu);
  }
  const uint32_t tl_t0 = tl_now();
  if (threadIdx.x == 0) g_tl_flags = fstate;
#endif

  if (wgid >= n1) {                                         // ---- a step-2 worker wavefront: `per_wave` blocks, slice by slice ----
    if (dbg & 1u) return;                                   // (timing experiment: the chains alone)
    uint32_t* wlds = s_mem + wv * WORKER_WORDS;
    const uint32_t wave_no = (wgid - n1) * (uint32_t)WGW + wv;
    // Its blocks are `nwaves` apart in the block order, not neighbours: the order goes resolution by resolution and band by
    // band, and the bands differ in how much they code (8K frame: five consecutive blocks of the first half of the order keep
    // a wavefront busy for 0.22 ms, five of the HH band of the top level for 0.16) -- every wavefront gets the same mix.
    const uint32_t nwaves = (n + per_wave - 1u) / per_wave;
    if (wave_no >= nwaves) return;
    uint32_t nb = 0;
    while (nb < per_wave && wave_no + nb * nwaves < n) ++nb;
    if (NR > 1)                                             // what does not depend on the chains, before the first wait (step2_block)
      for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t bi = wave_no + k * nwaves;
        const ojphgpu_cb_desc d = blocks[bi];
        step2_block<TX, 1, true, (NR > 1)>(d, bi, data, quads, coef, block_status, wlds + k * RING_ALLOC, nullptr, (int)lane, 0u, 0u,
                                           wlds + NR * RING_ALLOC + k * S2_STATE_WORDS, true);
      }
#ifdef FUSED_TIMELINE
    uint32_t* tl = g_tl + (size_t)((n1 * (uint32_t)WGW + wave_no) & 8191u) * 12u;
    uint32_t tl_wait = 0;
    if (lane == 0) { tl[0] = 2u; tl[1] = tl_where; tl[2] = tl_now(); }
#endif
#ifndef PRIO_PERIOD
#define PRIO_PERIOD 6u
#endif
    uint32_t wave_slot, prio_it = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(wave_slot));     // WAVE_ID: the wavefront's slot on its SIMD = its age there
    for (uint32_t sl = 0; sl < sched.n; ++sl) {
      uint32_t q0, q1;
      sched.bounds(sl, q0, q1);
      // the progress flags of the blocks' chain wavefronts, one per lane: one round trip for all of them
      const uint32_t flv = (lane < nb && !(dbg & 2u)) ? ld_agent(fstate + ((wave_no + lane * nwaves) >> 6)) : 0u;
      for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t bi = wave_no + k * nwaves;
        const ojphgpu_cb_desc d = blocks[bi];
        if (d.w == 0 || d.h == 0) continue;
        const uint32_t QH = ((uint32_t)d.h + 1) >> 1;
        if (q0 >= QH) continue;
        uint32_t* ring = wlds + (NR > 1 ? k : 0u) * RING_ALLOC;
        uint32_t* state = wlds + NR * RING_ALLOC + k * S2_STATE_WORDS;
        const uint32_t need = q1 < QH ? q1 : QH;
        bool there = true;
        const uint32_t fl = rdlane(flv, (int)k);
        if (!(dbg & 2u) && d.len1 != 0 && d.num_passes != 0 && !((fl >> 16) == (epoch & 0xFFFFu) && (fl & 0xFFFFu) >= need)) {   // (dbg 2: the workers alone, over the records of the run before)
#ifdef FUSED_TIMELINE
          const uint32_t tl_w0 = tl_now();
#endif
          const uint32_t seen_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)wait_rows(fstate + (bi >> 6), epoch, need, wait_ticks));
#ifdef FUSED_TIMELINE
          tl_wait += tl_now() - tl_w0;
#endif
          there = seen_rows != 0u;
        }
        if ((dbg & 4u) && sl == 1u && bi % 61u == 7u) there = false;       // (test switch: this wait "ran out")
        if (!there) {
          // the chains are resident (tickets), so this is a chip held up for seconds by something else: the run is
          // marked for a repeat through the separate launches and this wavefront stops (what it leaves undecoded is
          // decoded by the repeat)
          if (lane == 0) ask_for_repeat(retry, host_retry, epoch);
          return;
        }
        // The SIMD serves its wavefronts oldest first: of the six workers of a SIMD the one in slot 0 was through at 0.27 ms,
        // the one in slot 5 at 0.33, alone on its SIMD for the last stretch.  Two priority levels below the partners', taken
        // in turns -- the younger the wavefront, the larger its share of turns at the upper one -- even that out.
        if ((prio_it++ % PRIO_PERIOD) < wave_slot) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        step2_block<TX, 1, true, (NR > 1)>(d, bi, data, quads, coef, block_status, ring, nullptr, (int)lane, q0, q1, state);
      }
#ifdef FUSED_TIMELINE
      if (lane == 0 && sl < 7u) tl[4 + sl] = tl_now();
#endif
    }
#ifdef FUSED_TIMELINE
    if (lane == 0) { tl[3] = tl_now(); tl[11] = tl_wait; }
#endif
    return;
  }

  // ---- step 1: chains and their partners (ht_dec_step1_raw_kernel) ----
  if (dbg & 2u) return;                                     // (timing experiment: the workers alone)
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_vlc[i] = (&ojphgpu::g_dec_vlc32[0][0])[i];
  for (int i = threadIdx.x; i < 320; i += blockDim.x) s_uvlc0[i] = ojphgpu::g_dec_uvlc0[i];
  for (int i = threadIdx.x; i < CH * 5 * 64; i += blockDim.x) s_mem[CTL_OFF + i] = 0;
  __syncthreads();
  if (wv >= 3u * (uint32_t)CH) return;                      // (wavefronts the step-1 role has no use for)
  const bool chain = wv < (uint32_t)CH;
  const uint32_t set = wv % (uint32_t)CH;
  lds_u32* s_ev = (lds_u32*)(s_mem + EV_OFF + set * EV_RING * 64);
  lds_u32* s_vr = (lds_u32*)(s_mem + VR_OFF + set * VR_WORDS * 64);
  uint32_t* const ctl = s_mem + CTL_OFF + set * 5 * 64;
  volatile lds_u32* s_eprog = (volatile lds_u32*)(ctl);
  volatile lds_u32* s_econs = (volatile lds_u32*)(ctl + 64);
  volatile lds_u32* s_vprog = (volatile lds_u32*)(ctl + 128);
  volatile lds_u32* s_vcons = (volatile lds_u32*)(ctl + 192);
  volatile lds_u32* s_done = (volatile lds_u32*)(ctl + 256);
#ifndef PARTNER_PRIO
#define PARTNER_PRIO 2
#endif
  if (chain) __builtin_amdgcn_s_setprio(3);
  else if (PARTNER_PRIO) __builtin_amdgcn_s_setprio(PARTNER_PRIO);
  const uint32_t cw = wgid * (uint32_t)CH + set;            // the chain wavefront's number = its blocks' number / 64
  const uint32_t bi = cw * 64u + lane;
  uint32_t* flag = fstate + cw;
  // (no lane leaves early: the wavefront publishes "all rows done" at the end whatever its blocks are)
  bool alive = bi < n;
  ojphgpu_cb_desc d = blocks[alive ? bi : 0];
  if (alive && (d.w == 0 || d.h == 0 || d.len1 == 0 || d.num_passes == 0)) {      // not coded: zero block
    if (chain) __hip_atomic_store(block_status + bi, (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    alive = false;
  }
  const uint8_t* cb = data + d.data_off;
  const uint32_t room = d.data_off > 0xFFFFu ? 0xFFFFu : (uint32_t)d.data_off;
  B16 v0, v1;
  v0.d[0] = v0.d[1] = v0.d[2] = v0.d[3] = 0u; v1 = v0;
  if (alive && !chain && wv < 2u * (uint32_t)CH) {
    const int o0 = (int)d.len1 - 18, o1 = o0 - 16;
    if (o0 >= -(int)room) v0 = load_b16_unaligned(cb + o0);
    else for (int jj = 0; jj < 16; ++jj) { const int o = o0 + 15 - jj; if (o >= 0) v0.d[3 - (jj >> 2)] |= (uint32_t)cb[o] << (24 - 8 * (jj & 3)); }
    if (o1 >= -(int)room) v1 = load_b16_unaligned(cb + o1);
    else for (int jj = 0; jj < 16; ++jj) { const int o = o1 + 15 - jj; if (o >= 0) v1.d[3 - (jj >> 2)] |= (uint32_t)cb[o] << (24 - 8 * (jj & 3)); }
  }
  uint32_t scup = 0;
  if (alive) {
    scup = check_block(d, cb);
    if (scup == 0) { if (chain) __hip_atomic_store(block_status + bi, (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); alive = false; }
  }
  if (alive) {
    const uint32_t QW = ((uint32_t)d.w + 1) >> 1, QH = ((uint32_t)d.h + 1) >> 1;
    const uint32_t evw = ev_words_of(QW, QH);
    if (!chain) {
      if (wv < 2u * (uint32_t)CH) raw_partner<1>(cb, d.len1, scup, room, v0, v1, evw, s_ev, s_eprog, s_econs, s_vr, s_vprog, s_vcons, s_done, lane);
      else raw_partner<2>(cb, d.len1, scup, room, v0, v1, evw, s_ev, s_eprog, s_econs, s_vr, s_vprog, s_vcons, s_done, lane);
    } else {
      __hip_atomic_store(block_status + bi, (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (what the last run left is not this run's)
      uint32_t* rec = quads + rec16_base(d, bi);
      RingRd vlc; vlc.init(s_vr, s_vprog, s_vcons, lane);
      EvRd mel; mel.init(s_ev, s_eprog, s_econs, evw, lane);
      step1_rows<1, true>(vlc, mel, rec, QW, QH, s_vlc, s_uvlc0, flag, epoch, (lds_u32*)(s_mem + REC_OFF + set * REC16_ROW_WORDS * 64), sched);
      s_done[lane] = 1u;
      if (mel.stuck || vlc.stuck) ask_for_repeat(retry, host_retry, epoch);   // (gave up on its partner: not a verdict on the block -- repeat the run)
    }
  }
  if (chain) publish_rows(flag, epoch, 0xFFFFu, lane);     // every row of every block of this wavefront is complete
#ifdef FUSED_TIMELINE
  if (lane == 0) {
    uint32_t* tl = g_tl + (size_t)((wgid * (uint32_t)WGW + wv) & 8191u) * 12u;
    tl[0] = chain ? 1u : 3u; tl[1] = tl_where; tl[2] = tl_t0; tl[3] = tl_now();
  }
#endif
}


// -------------------------------------------------------------------------------------------------
// refinement: SigProp + MagRef passes (block_decoder32.cpp:1318-1609), one wavefront = one code-block
// -------------------------------------------------------------------------------------------------
// Only foreign codestreams carry these passes (the reference encoder emits the cleanup pass alone).
// Both are serial scans -- a sample's membership in SigProp depends on what its neighbours just
// became -- so one lane walks them; the wavefront's job is to make that walk cheap: all lanes bring
// the block (sign-magnitude words left by step 2), its cleanup significance and the refinement
// bytes into LDS, lane 0 runs the two passes out of LDS, and all lanes de-quantise and store the
// block in coalesced rows.  Significance: one 16-bit word per 4-row stripe and group of 4 columns,
// column-major, bit 4*c + r (:1331-1362).
constexpr int RWAVES = 2;
constexpr uint32_t SIG_ENTRIES = 1040;        // (stripes + 1) * (groups + 2) 16-bit words, w * h <= 4096
constexpr uint32_t PREV_ENTRIES = 264;

struct RefineLds {
  uint32_t smp[4096];
  uint16_t sigma[SIG_ENTRIES];
  uint16_t prev_row[PREV_ENTRIES];
  uint8_t  bytes[2048];
};

struct FwdBits {            // SigProp: forward, LSB first, after 0xFF only 7 bits, exhausted -> zeros (frwd_read<0> :609-655)
  const uint8_t* d; int i, n_bytes; uint64_t win; uint32_t n, unstuff;
  __device__ __forceinline__ void init(const uint8_t* p, int len) { d = p; i = 0; n_bytes = len; win = 0; n = 0; unstuff = 0; }
  __device__ __forceinline__ uint32_t bit() {
    if (n == 0) {
      const uint32_t b = i < n_bytes ? d[i] : 0u; ++i;
      win |= (uint64_t)b;                               // all 8 bits are OR-ed in, also when only 7 count
      n = 8u - unstuff; unstuff = (b == 0xFFu);
    }
    const uint32_t r = (uint32_t)win & 1u; win >>= 1; --n;
    return r;
  }
};

struct BwdBits {            // MagRef: backward from the end, VLC stuffing rule, starts with unstuff = true (rev_read_mrp :453-545)
  const uint8_t* d; int i; uint64_t win; uint32_t n, unstuff;
  __device__ __forceinline__ void init(const uint8_t* p, int len) { d = p; i = len - 1; win = 0; n = 0; unstuff = 1; }
  __device__ __forceinline__ uint32_t bit() {
    if (n == 0) {
      const uint32_t b = i >= 0 ? d[i] : 0u; --i;
      win |= (uint64_t)b;
      n = 8u - ((unstuff && (b & 0x7Fu) == 0x7Fu) ? 1u : 0u); unstuff = b > 0x8Fu;
    }
    const uint32_t r = (uint32_t)win & 1u; win >>= 1; --n;
    return r;
  }
};

// sigma of a block from the records step 1 left in the quad scratch (32-bit records, pair-major: rec + (qy PW + qx / 2)
// REC_STRIDE + (qx & 1), rho in bits 4..7): the reference fills sigma from the quads' rho bits (:1321-1351), NOT from "the
// sample is not zero" -- on a damaged VLC segment a quad may call samples of its second column / second row significant
// where an odd-sized block has no such column / row, and those bits count as neighbours in SigProp and take a bit each in
// MagRef (their samples do not exist and are never written).  Lane l takes quad columns l, l + 64, ...
__device__ __forceinline__ void sigma_from_records(const uint32_t* __restrict__ rec, uint32_t QW, uint32_t QH, uint16_t* sigma,
                                                   uint32_t mstr, int lane)
{
  const uint32_t PW = (QW + 1u) >> 1;
  for (uint32_t qy = 0; qy < QH; ++qy)
    for (uint32_t qx = (uint32_t)lane; qx < QW; qx += 64u) {
      const uint32_t rho = (rec[(size_t)(qy * PW + (qx >> 1)) * REC_STRIDE + (qx & 1u)] >> 4) & 0xFu;
      if (rho == 0u) continue;
      // the quad's samples (x, y) = (2 qx + (n >> 1), 2 qy + (n & 1)): columns x & 3 in {0, 1} or {2, 3} of group qx / 2,
      // rows y & 3 in {0, 1} or {2, 3} of stripe qy / 2
      const uint32_t cb = 2u * (qx & 1u), rb = 2u * (qy & 1u);
      const uint32_t bits = ((rho & 1u) << (4u * cb + rb)) | (((rho >> 1) & 1u) << (4u * cb + rb + 1u)) |
                            (((rho >> 2) & 1u) << (4u * cb + 4u + rb)) | (((rho >> 3) & 1u) << (4u * cb + 4u + rb + 1u));
      const uint32_t e = (qy >> 1) * mstr + (qx >> 1);
      atomicOr(reinterpret_cast<uint32_t*>(sigma) + (e >> 1), bits << (16u * (e & 1u)));
    }
}

__global__ __launch_bounds__(64 * RWAVES) void ht_dec_refine_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    const uint32_t* __restrict__ quads, uint32_t* __restrict__ coef, const uint8_t* __restrict__ block_status)
{
  __shared__ RefineLds s_wave[RWAVES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t bi = blockIdx.x * RWAVES + wave;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (!needs_refinement(d) || d.w == 0 || d.h == 0 || d.len1 == 0 || block_status[bi] != 0 || (d.reversible & 4u)) return;   // (64-bit: ht_dec64_refine_kernel)
  RefineLds& L = s_wave[wave];
  const uint32_t W = d.w, H = d.h, pitch = d.pitch;
  uint32_t* plane = coef + d.coef_off;
  const bool rev = (d.reversible & 1u) != 0, causal = (d.reversible & 2u) != 0;
  const uint32_t p = 30u - d.missing_msbs;
  const int ngroups = (int)((W + 3) >> 2), mstr = ngroups + 2;

  for (uint32_t i = lane; i < SIG_ENTRIES / 2; i += 64) reinterpret_cast<uint32_t*>(L.sigma)[i] = 0;
  for (uint32_t i = lane; i < PREV_ENTRIES / 2; i += 64) reinterpret_cast<uint32_t*>(L.prev_row)[i] = 0;
  wave_sync();
  for (uint32_t y = 0; y < H; ++y)
    for (uint32_t x = lane; x < W; x += 64) L.smp[y * W + x] = plane[(size_t)y * pitch + x];
  sigma_from_records(quads + d.scratch_cap, (W + 1u) >> 1, (H + 1u) >> 1, L.sigma, (uint32_t)mstr, lane);
  const uint8_t* seg = data + d.data_off + d.len1;
  const int len2 = (int)d.len2;                              // < 2047 (ojph_precinct.cpp:509)
  for (int i = lane; i < len2 && i < 2048; i += 64) L.bytes[i] = seg[i];
  wave_sync();

  if (lane == 0) {
    // ---- significance propagation (:1364-1558) ----
    FwdBits spp; spp.init(L.bytes, len2 < 2048 ? len2 : 2048);
    for (int y = 0; y < (int)H; y += 4) {
      uint32_t pattern = 0xFFFFu;
      if ((int)H - y < 4) { pattern = 0x7777u; if ((int)H - y < 3) { pattern = 0x3333u; if ((int)H - y < 2) pattern = 0x1111u; } }
      uint32_t prev = 0;
      const uint16_t* cur_sig = L.sigma + (y >> 2) * mstr;
      const uint16_t* nxt_sig = cur_sig + mstr;
      for (int x = 0, g = 0; x < (int)W; x += 4, ++g) {
        int s = x + 4 - (int)W; if (s < 0) s = 0;
        pattern >>= s * 4;
        const uint32_t ps = L.prev_row[g] | ((uint32_t)L.prev_row[g + 1] << 16);
        const uint32_t ns = nxt_sig[g] | ((uint32_t)nxt_sig[g + 1] << 16);
        uint32_t u = (ps & 0x88888888u) >> 3;                // the row on top
        if (!causal) u |= (ns & 0x11111111u) << 3;           // the row below
        const uint32_t cs = cur_sig[g] | ((uint32_t)cur_sig[g + 1] << 16);
        uint32_t mbr = cs | ((cs & 0x77777777u) << 1) | ((cs & 0xEEEEEEEEu) >> 1) | u;
        uint32_t t = mbr;
        mbr |= (t << 4) | (t >> 4) | (prev >> 12);
        mbr &= pattern; mbr &= ~cs;
        uint32_t new_sig = mbr;
        if (new_sig) {
          const uint32_t inv_sig = ~cs & pattern;
          for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) {
              const uint32_t b = 1u << (4 * c + r);
              if (!(new_sig & b)) continue;
              new_sig &= ~b;
              if (spp.bit()) {
                const uint32_t grow = r == 0 ? 0x33u : (r == 1 ? 0x76u : (r == 2 ? 0xECu : 0xC8u));
                new_sig |= (grow << (4 * c)) & inv_sig;
              }
            }
          new_sig &= 0xFFFFu;
          for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r)
              if (new_sig & (1u << (4 * c + r)))
                L.smp[(uint32_t)(y + r) * W + (uint32_t)(x + c)] = (spp.bit() << 31) | (3u << (p - 2));
        }
        new_sig |= cs;
        L.prev_row[g] = (uint16_t)new_sig;
        t = new_sig;
        new_sig |= ((t & 0x7777u) << 1) | ((t & 0xEEEEu) >> 1);
        prev = (new_sig | u) & 0xF000u;
      }
    }
    // ---- magnitude refinement (:1561-1609) ----
    if (d.num_passes > 2) {
      BwdBits mrp; mrp.init(L.bytes, len2 < 2048 ? len2 : 2048);
      const uint32_t half = 1u << (p - 2);
      for (int y = 0; y < (int)H; y += 4)
        for (int x = 0; x < 4 * ngroups; ++x) {
          const uint32_t nib = ((uint32_t)L.sigma[(y >> 2) * mstr + (x >> 2)] >> (4 * (x & 3))) & 0xFu;
          for (int r = 0; r < 4; ++r)
            if (nib & (1u << r)) {
              const uint32_t sym = mrp.bit();           // (a flagged sample outside the block takes its bit too: :1583-1606)
              if (x < (int)W && y + r < (int)H) L.smp[(uint32_t)(y + r) * W + (uint32_t)x] ^= ((1u - sym) << (p - 1)) | half;
            }
        }
    }
  }
  wave_sync();
  const uint32_t shift = 31 - d.K_max;
  const float delta = d.delta;
  for (uint32_t y = 0; y < H; ++y)
    for (uint32_t x = lane; x < W; x += 64) plane[(size_t)y * pitch + x] = dequantise(L.smp[y * W + x], rev, shift, delta);
}

// -------------------------------------------------------------------------------------------------
// 64-bit sample path: ojph_decode_codeblock64 (block_decoder64.cpp:766-1660), transfer gen_rev_tx_from_cb64
// (ojph_codestream_gen.cpp:140-153)
// -------------------------------------------------------------------------------------------------
// Components that need more than 32 bits of precision (param_qcd::propose_precision, ojph_params.cpp:1684-1706 -- samples
// deeper than about 26 bits, reversible) are decoded by the reference's 64-bit function: the same algorithm with 64-bit
// MagSgn values, a U-VLC extension for u > 32, and byte readers of its own -- its VLC and MagSgn readers take ONE byte at a
// time and MASK the bit a stuffed byte may not carry (rev_read8 :305-327, frwd_read8 :626-640), where the 32-bit function's
// four-byte readers OR the byte in whole; after a masked byte the "previous byte" the stuffing rule looks at is the masked
// value.  The same on conforming streams, different on corrupt ones, so the un-stuffers below restate THOSE rules: whether
// byte k carries 7 bits depends on the parity of the run of 0xFF bytes in front of it (a lane looks back over that run;
// runs are short).  This path is rare by nature and built from the simple forms of the stages: a prep launch un-stuffs
// all THREE segments of a block into flat bit strings in d_aux (VLC | MEL | MagSgn), step 1 is the lane-per-block chain
// on those strings (ht_dec_step1_kernel<CH, true>: FlatRd + the MEL partner wavefront), step 2 reads the MagSgn bits
// from the flat string -- one wavefront per block, one lane per column, any width, 64-bit samples out.
__host__ __device__ __forceinline__ uint32_t ms_words64(uint32_t len1) { return (len1 * 8u + 31u) / 32u + 6u; }   // data + 5 words of ones

// 256 bytes per round, 4 per lane, bits placed by a wavefront prefix sum: KIND 0 = the VLC segment backwards (rev_init8 /
// rev_read8), 1 = the MagSgn segment forwards (frwd_read8<0xFF>); `out` gets the flat LSB-first string and pad words (VLC:
// zeros, MagSgn: ones -- what the readers feed when the segment is exhausted)
template <int KIND>
__device__ uint32_t flatten64(const uint8_t* __restrict__ cb, uint32_t lcup, uint32_t scup, uint32_t* __restrict__ out, uint32_t nominal,
                              uint32_t* lds, int lane)
{
  const uint32_t count = KIND == 0 ? scup - 2u : lcup - scup;
  for (int i = lane; i < 68; i += 64) lds[i] = 0;
  wave_sync();
  uint32_t cursor = 0, wpos = 0;
  bool unstuff0 = false;                                   // state in front of byte 0
  if (KIND == 0) {                                         // rev_init8 (:343-361): the half byte, masked
    uint32_t v = cb[lcup - 2] >> 4;
    const uint32_t t = (v & 7u) == 7u ? 1u : 0u;
    v &= 0xFu >> t;
    if (lane == 0) lds[0] = v;
    cursor = 4u - t; unstuff0 = v > 8u;
    wave_sync();
  }
  auto raw = [&](uint32_t k) -> uint32_t { return KIND == 0 ? (uint32_t)cb[lcup - 3u - k] : (uint32_t)cb[k]; };
  for (uint32_t base = 0; base < count; base += 256) {
    const uint32_t k0 = base + 4u * (uint32_t)lane;
    uint32_t val = 0, nb = 0;
    if (k0 < count) {
      // the state in front of byte k0, from the run of 0xFF bytes that ends at k0 - 1
      uint32_t r = 0;
      while (r < k0 && raw(k0 - 1u - r) == 0xFFu) ++r;
      bool unstuff;
      if (KIND == 0) {
        // VLC: a byte is short when the (masked) byte before it is > 0x8F and its own low 7 bits are ones; a short 0xFF
        // becomes 0x7F.  In front of the run: c = the byte before it (never 0xFF), or the half byte
        const bool u_first = r == k0 ? unstuff0 : raw(k0 - 1u - r) > 0x8Fu;     // is the first 0xFF of the run short?
        unstuff = r == 0 ? u_first : ((r & 1u) ? !u_first : u_first);            // after FF_r: not short -> 0xFF > 0x8F
      } else
        unstuff = (r & 1u) != 0u;                          // MagSgn: byte k is short when an odd number of 0xFF precede it
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t k = k0 + j;
        if (k < count) {
          uint32_t b = raw(k);
          const uint32_t t = KIND == 0 ? ((unstuff && (b & 0x7Fu) == 0x7Fu) ? 1u : 0u) : (unstuff ? 1u : 0u);
          b &= 0xFFu >> t;
          val |= b << nb; nb += 8u - t;
          unstuff = KIND == 0 ? b > 0x8Fu : b == 0xFFu;
        }
      }
    }
    const uint32_t incl = wave_incl_scan(nb);
    or_bits(lds, cursor + incl - nb, val, nb);
    const uint32_t T = cursor + rdlane(incl, 63);
    wave_sync();
    const uint32_t nfull = T >> 5;
    for (uint32_t i = lane; i < nfull; i += 64) out[wpos + i] = lds[i];
    const uint32_t carry = lds[nfull];
    wave_sync();
    for (int i = lane; i < 68; i += 64) lds[i] = 0;
    wave_sync();
    if (lane == 0) lds[0] = carry;
    wave_sync();
    wpos += nfull; cursor = T & 31u;
  }
  const uint32_t total = wpos * 32u + cursor;
  const uint32_t fill = KIND == 0 ? 0u : 0xFFFFFFFFu;
  if (lane == 0 && cursor) { uint32_t w = lds[0]; if (KIND == 1) w |= 0xFFFFFFFFu << cursor; out[wpos] = w; }
  if (cursor) wpos++;
  for (uint32_t i = wpos + (uint32_t)lane; i < nominal; i += 64) out[i] = fill;      // (see flatten)
  wave_sync();
  return total;
}

__global__ __launch_bounds__(64 * WAVES) void ht_dec64_prep_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data, uint32_t* __restrict__ aux)
{
  __shared__ uint32_t s_buf[WAVES][68];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t bi = blockIdx.x * WAVES + wave;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (!(d.reversible & 4u) || d.w == 0 || d.h == 0 || d.len1 == 0 || d.num_passes == 0) return;
  const uint8_t* cb = data + d.data_off;
  const uint32_t scup = check_block(d, cb);
  if (scup == 0) return;
  uint32_t* out = aux + d.reserved;
  flatten64<0>(cb, d.len1, scup, out, vlc_words(scup), s_buf[wave], lane);
  flatten<true>(cb, d.len1, scup, out + vlc_words(scup), s_buf[wave], lane);      // (the MEL reader is the 32-bit function's, :93-157)
  flatten64<1>(cb, d.len1, scup, out + vlc_words(scup) + mel_words(scup), ms_words64(d.len1), s_buf[wave], lane);
}

// step 2 of one block of 64-bit samples: the wavefront walks the quad rows, 64 columns at a time
constexpr int W64_WAVES = 2;
constexpr uint32_t W64_EXP = 1024 + 8;
__global__ __launch_bounds__(64 * W64_WAVES) void ht_dec64_step2_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data, const uint32_t* __restrict__ aux,
    const uint32_t* __restrict__ quads, uint32_t* __restrict__ coef, uint8_t* __restrict__ block_status)
{
  __shared__ uint8_t s_exp[W64_WAVES][2][W64_EXP];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t bi = blockIdx.x * W64_WAVES + wave;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (!(d.reversible & 4u)) return;
  const uint32_t W = d.w, H = d.h, pitch = d.pitch;
  if (W == 0 || H == 0) return;
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(coef + d.coef_off);
  auto zero_block = [&]() {
    for (uint32_t y = 0; y < H; ++y)
      for (uint32_t x = lane; x < W; x += 64) dst[(size_t)y * pitch + x] = 0ull;
  };
  if (d.len1 == 0 || d.num_passes == 0 || block_status[bi] != 0) { zero_block(); return; }
  const uint32_t missing_msbs = d.missing_msbs, p = 62u - missing_msbs, mmsbp2 = missing_msbs + 2u;
  const uint8_t* cb = data + d.data_off;
  const uint32_t lcup = d.len1;
  const uint32_t scup = ((uint32_t)cb[lcup - 1] << 4) + (cb[lcup - 2] & 0xFu);
  const uint32_t ms_len = lcup - scup;
  const uint32_t* ms = aux + d.reserved + vlc_words(scup) + mel_words(scup);
  const uint32_t ms_last = ms_words64(lcup) - 1u;          // (holds ones whatever the segment: flatten64 fills up to there)
  (void)ms_len;
  const uint32_t QW = (W + 1) >> 1, QH = (H + 1) >> 1, PW = (QW + 1) >> 1;
  const uint32_t* rec = quads + d.scratch_cap;
  const bool raw_out = needs_refinement(d);
  const uint32_t shift = 63u - d.K_max;
  uint8_t* ex = &s_exp[wave][0][0];
  for (uint32_t i = lane; i < 2 * W64_EXP / 4; i += 64) reinterpret_cast<uint32_t*>(ex)[i] = 0;
  wave_sync();
  uint32_t mpos = 0;
  bool bad = false;
  for (uint32_t qy = 0; qy < QH && !bad; ++qy) {
    const uint8_t* vexp = ex + (qy & 1) * W64_EXP;         // exponents of the sample row above (+1 offset)
    uint8_t* vnew = ex + ((qy & 1) ^ 1) * W64_EXP;
    for (uint32_t c0 = 0; c0 < W; c0 += 64) {
      const uint32_t col = c0 + (uint32_t)lane;
      const bool act = col < W;
      const uint32_t qx = col >> 1, half = (uint32_t)lane & 1u;
      const uint32_t ent = act ? rec[(size_t)(qy * PW + (qx >> 1)) * REC_STRIDE + (qx & 1u)] : 0u;
      const uint32_t inf = ent & 0xFFFFu;
      uint32_t U_q = ent >> 16;
      if (qy > 0) {
        uint32_t gamma = inf & 0xF0u; gamma &= gamma - 0x10u;                           // :1266
        const uint32_t b = 2 * qx;
        const uint32_t em = act ? max(max((uint32_t)vexp[b], (uint32_t)vexp[b + 1]), max((uint32_t)vexp[b + 2], (uint32_t)vexp[b + 3])) : 0u;
        U_q += gamma ? max(em, 1u) : 1u;                                                // :1267-1270
      }
      if (__ballot(act && U_q > mmsbp2) != 0ull) { bad = true; break; }                 // :1162, :1271
      const uint32_t sel = inf >> (2u * half);
      const uint32_t m0 = (sel & 0x10u) ? U_q - ((sel >> 12) & 1u) : 0u;
      const uint32_t m1 = (sel & 0x20u) ? U_q - ((sel >> 13) & 1u) : 0u;
      const uint32_t tot = m0 + m1;
      const uint32_t incl = wave_incl_scan(tot);
      const uint32_t at = mpos + incl - tot;
      mpos += rdlane(incl, 63);
      // 128 bits from `at` (words beyond the string read as the all-ones pad)
      const uint32_t wi = at >> 5, sh = at & 31u;
      uint32_t w[5];
#pragma unroll
      for (uint32_t i = 0; i < 5; ++i) w[i] = ms[min(wi + i, ms_last)];
      const uint64_t lo = (uint64_t)__funnelshift_r(w[0], w[1], sh) | ((uint64_t)__funnelshift_r(w[1], w[2], sh) << 32);
      const uint64_t hi = (uint64_t)__funnelshift_r(w[2], w[3], sh) | ((uint64_t)__funnelshift_r(w[3], w[4], sh) << 32);
      auto one = [&](uint64_t ms_val, uint32_t m, uint32_t e1, bool on, uint64_t& v_keep) -> uint64_t {   // :1166-1182
        uint64_t v_n = ms_val & ((m < 64u ? (1ull << m) : 0ull) - 1ull);
        v_n |= (uint64_t)e1 << (m & 63u);
        v_n |= 1ull;
        v_keep = on ? v_n : 0ull;
        return on ? ((ms_val << 63) | ((v_n + 2ull) << (p - 1u))) : 0ull;
      };
      uint64_t v0k, v1k;
      const uint64_t val0 = one(lo, m0, (sel >> 8) & 1u, (sel & 0x10u) != 0u, v0k);
      const uint64_t ms1 = m0 == 0u ? lo : (m0 >= 64u ? hi : ((lo >> m0) | (hi << (64u - m0))));
      const uint64_t val1 = one(ms1, m1, (sel >> 9) & 1u, (sel & 0x20u) != 0u, v1k);
      (void)v0k;
      if (act) vnew[col + 1] = (uint8_t)(v1k ? 63u - (uint32_t)__clzll((long long)v1k) : 0u);
      if (act) {
        auto xfer = [&](uint64_t v) -> unsigned long long {                              // gen_rev_tx_from_cb64
          if (raw_out) return v;
          const long long mag = (long long)((v & 0x7FFFFFFFFFFFFFFFull) >> shift);
          return (unsigned long long)((v >> 63) ? -mag : mag);
        };
        const uint32_t y = 2 * qy;
        dst[(size_t)y * pitch + col] = xfer(val0);
        if (y + 1 < H) dst[(size_t)(y + 1) * pitch + col] = xfer(val1);
      }
    }
    wave_sync();
  }
  if (bad) { zero_block(); if (lane == 0) block_status[bi] = 1; }
}

// SigProp + MagRef of a block of 64-bit samples (block_decoder64.cpp:1360-1657): ht_dec_refine_kernel with 64-bit words
struct RefineLds64 {
  uint64_t smp[4096];
  uint16_t sigma[SIG_ENTRIES];
  uint16_t prev_row[PREV_ENTRIES];
  uint8_t  bytes[2048];
};

__global__ __launch_bounds__(64) void ht_dec64_refine_kernel(
    const ojphgpu_cb_desc* __restrict__ blocks, uint32_t n, const uint8_t* __restrict__ data,
    const uint32_t* __restrict__ quads, uint32_t* __restrict__ coef, const uint8_t* __restrict__ block_status)
{
  __shared__ RefineLds64 L;
  const int lane = threadIdx.x & 63;
  const uint32_t bi = blockIdx.x;
  if (bi >= n) return;
  const ojphgpu_cb_desc d = blocks[bi];
  if (!(d.reversible & 4u) || !needs_refinement(d) || d.w == 0 || d.h == 0 || d.len1 == 0 || block_status[bi] != 0) return;
  const uint32_t W = d.w, H = d.h, pitch = d.pitch;
  unsigned long long* plane = reinterpret_cast<unsigned long long*>(coef + d.coef_off);
  const bool causal = (d.reversible & 2u) != 0;
  const uint32_t p = 62u - d.missing_msbs;
  const int ngroups = (int)((W + 3) >> 2), mstr = ngroups + 2;
  for (uint32_t i = lane; i < SIG_ENTRIES / 2; i += 64) reinterpret_cast<uint32_t*>(L.sigma)[i] = 0;
  for (uint32_t i = lane; i < PREV_ENTRIES / 2; i += 64) reinterpret_cast<uint32_t*>(L.prev_row)[i] = 0;
  wave_sync();
  for (uint32_t y = 0; y < H; ++y)
    for (uint32_t x = lane; x < W; x += 64) L.smp[y * W + x] = plane[(size_t)y * pitch + x];
  sigma_from_records(quads + d.scratch_cap, (W + 1u) >> 1, (H + 1u) >> 1, L.sigma, (uint32_t)mstr, lane);   // (block_decoder64.cpp:1363-1393)
  const uint8_t* seg = data + d.data_off + d.len1;
  const int len2 = (int)d.len2;
  for (int i = lane; i < len2 && i < 2048; i += 64) L.bytes[i] = seg[i];
  wave_sync();
  if (lane == 0) {
    FwdBits spp; spp.init(L.bytes, len2 < 2048 ? len2 : 2048);
    for (int y = 0; y < (int)H; y += 4) {
      uint32_t pattern = 0xFFFFu;
      if ((int)H - y < 4) { pattern = 0x7777u; if ((int)H - y < 3) { pattern = 0x3333u; if ((int)H - y < 2) pattern = 0x1111u; } }
      uint32_t prev = 0;
      const uint16_t* cur_sig = L.sigma + (y >> 2) * mstr;
      const uint16_t* nxt_sig = cur_sig + mstr;
      for (int x = 0, g = 0; x < (int)W; x += 4, ++g) {
        int sft = x + 4 - (int)W; if (sft < 0) sft = 0;
        pattern >>= sft * 4;
        const uint32_t ps = L.prev_row[g] | ((uint32_t)L.prev_row[g + 1] << 16);
        const uint32_t ns = nxt_sig[g] | ((uint32_t)nxt_sig[g + 1] << 16);
        uint32_t u = (ps & 0x88888888u) >> 3;
        if (!causal) u |= (ns & 0x11111111u) << 3;
        const uint32_t cs = cur_sig[g] | ((uint32_t)cur_sig[g + 1] << 16);
        uint32_t mbr = cs | ((cs & 0x77777777u) << 1) | ((cs & 0xEEEEEEEEu) >> 1) | u;
        uint32_t t = mbr;
        mbr |= (t << 4) | (t >> 4) | (prev >> 12);
        mbr &= pattern; mbr &= ~cs;
        uint32_t new_sig = mbr;
        if (new_sig) {
          const uint32_t inv_sig = ~cs & pattern;
          for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) {
              const uint32_t b = 1u << (4 * c + r);
              if (!(new_sig & b)) continue;
              new_sig &= ~b;
              if (spp.bit()) {
                const uint32_t grow = r == 0 ? 0x33u : (r == 1 ? 0x76u : (r == 2 ? 0xECu : 0xC8u));
                new_sig |= (grow << (4 * c)) & inv_sig;
              }
            }
          new_sig &= 0xFFFFu;
          for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r)
              if (new_sig & (1u << (4 * c + r)))
                L.smp[(uint32_t)(y + r) * W + (uint32_t)(x + c)] = ((uint64_t)spp.bit() << 63) | (3ull << (p - 2));
        }
        new_sig |= cs;
        L.prev_row[g] = (uint16_t)new_sig;
        t = new_sig;
        new_sig |= ((t & 0x7777u) << 1) | ((t & 0xEEEEu) >> 1);
        prev = (new_sig | u) & 0xF000u;
      }
    }
    if (d.num_passes > 2) {
      BwdBits mrp; mrp.init(L.bytes, len2 < 2048 ? len2 : 2048);
      const uint64_t half = 1ull << (p - 2);
      for (int y = 0; y < (int)H; y += 4)
        for (int x = 0; x < 4 * ngroups; ++x) {
          const uint32_t nib = ((uint32_t)L.sigma[(y >> 2) * mstr + (x >> 2)] >> (4 * (x & 3))) & 0xFu;
          for (int r = 0; r < 4; ++r)
            if (nib & (1u << r)) {
              const uint32_t sym = mrp.bit();           // (also for a flagged sample outside the block)
              if (x < (int)W && y + r < (int)H) L.smp[(uint32_t)(y + r) * W + (uint32_t)x] ^= ((uint64_t)(1u - sym) << (p - 1)) | half;
            }
        }
    }
  }
  wave_sync();
  const uint32_t shift = 63u - d.K_max;
  for (uint32_t y = 0; y < H; ++y)
    for (uint32_t x = lane; x < W; x += 64) {
      const uint64_t v = L.smp[y * W + x];
      const long long mag = (long long)((v & 0x7FFFFFFFFFFFFFFFull) >> shift);
      plane[(size_t)y * pitch + x] = (unsigned long long)((v >> 63) ? -mag : mag);
    }
}

}  // namespace

extern "C" uint32_t ojphgpu_ht_decode_aux_words(uint32_t len1) { return aux_words(len1); }

extern "C" int ojphgpu_ht_decode_layout(ojphgpu_cb_desc* h, uint32_t n, uint64_t* quad_elems, uint64_t* aux_elems)
{
  if ((!h && n) || !quad_elems || !aux_elems) return OJPHGPU_E_INVALID;
  uint64_t q = 0, a = 0;
  for (uint32_t g = 0; g < n; g += 64) {                      // the 64 blocks one step-1 wavefront advances together
    const uint32_t m = n - g < 64u ? n - g : 64u;
    uint64_t pairs = 1, qh_max = 0;
    bool narrow = true;
    for (uint32_t l = 0; l < m; ++l) {
      const uint64_t qw = ((uint64_t)h[g + l].w + 1) >> 1, qh = ((uint64_t)h[g + l].h + 1) >> 1;
      pairs = std::max<uint64_t>(pairs, ((qw + 1) >> 1) * qh);
      qh_max = std::max(qh_max, qh);
      narrow = narrow && h[g + l].w <= 64u;
    }
    for (uint32_t l = 0; l < m; ++l) {
      if (q + 2 * l > 0xFFFFFFFFull || a > 0xFFFFFFFFull) return OJPHGPU_E_INVALID;
      h[g + l].scratch_cap = (uint32_t)(q + 2 * l);
      h[g + l].reserved = (uint32_t)a;
      a += aux_words(h[g + l].len1);
      if (h[g + l].reversible & 4u) a += ms_words64(h[g + l].len1);      // 64-bit sample path: the flat MagSgn string as well
    }
    // the fused launch's 16-bit records of the same group share the area (rec16_base): whole slices of S2_ROWS quad rows,
    // 64 blocks, 64 bytes per block and row
    const uint64_t slices = (qh_max + S2_ROWS - 1) / S2_ROWS;
    q += std::max<uint64_t>((uint64_t)REC_STRIDE * pairs, narrow ? slices * 64u * S2_ROWS * REC16_ROW_WORDS : 0u);
  }
  if (q > 0xFFFFFFFFull || a > 0xFFFFFFFFull) return OJPHGPU_E_INVALID;
  *quad_elems = q; *aux_elems = a;
  return OJPHGPU_OK;
}

namespace ojphgpu {
// OJPHGPU_DEC_PREP=1: the two-launch form (prep writes the flat VLC / MEL strings, step 1 reads them) for A/B runs
bool dec_uses_prep()
{
  static const bool v = [] { const char* e = getenv("OJPHGPU_DEC_PREP"); return e && atoi(e) != 0; }();
  return v;
}
}

extern "C" int ojphgpu_ht_decode_prep(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                                       const uint8_t* d_data, uint32_t* d_aux)
{
  if (n == 0) return OJPHGPU_OK;
  if (!d_blocks || !d_data || !d_aux) return OJPHGPU_E_INVALID;
  if (!ojphgpu::dec_uses_prep()) return OJPHGPU_OK;            // step 1 reads the raw bytes itself
  hipLaunchKernelGGL(ht_dec_prep_kernel, dim3((n + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, (hipStream_t)stream,
                     d_blocks, n, d_data, d_aux);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

extern "C" int ojphgpu_ht_decode_step1(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                                        const uint8_t* d_data, const uint32_t* d_aux, uint32_t* d_quad_scratch,
                                        uint8_t* d_block_status)
{
  if (n == 0) return OJPHGPU_OK;
  if (ojphgpu::ensure_tables() != 0) return OJPHGPU_E_HIP;
  if (!d_blocks || !d_data || !d_aux || !d_quad_scratch || !d_block_status) return OJPHGPU_E_INVALID;
  static const int ch = [] { const char* e = getenv("OJPHGPU_S1_CH"); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 || v == 4 ? v : 4; }();
  const uint32_t sets = (n + 63) / 64;
  if (!ojphgpu::dec_uses_prep()) {
    if (ch == 4) hipLaunchKernelGGL(ht_dec_step1_raw_kernel<4>, dim3((sets + 3) / 4), dim3(768), 0, (hipStream_t)stream, d_blocks, n, d_data, d_quad_scratch, d_block_status);
    else if (ch == 2) hipLaunchKernelGGL(ht_dec_step1_raw_kernel<2>, dim3((sets + 1) / 2), dim3(384), 0, (hipStream_t)stream, d_blocks, n, d_data, d_quad_scratch, d_block_status);
    else hipLaunchKernelGGL(ht_dec_step1_raw_kernel<1>, dim3(sets), dim3(192), 0, (hipStream_t)stream, d_blocks, n, d_data, d_quad_scratch, d_block_status);
    return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
  }
  if (ch == 4) hipLaunchKernelGGL(ht_dec_step1_kernel<4>, dim3((sets + 3) / 4), dim3(512), 0, (hipStream_t)stream, d_blocks, n, d_data, d_aux, d_quad_scratch, d_block_status);
  else if (ch == 2) hipLaunchKernelGGL(ht_dec_step1_kernel<2>, dim3((sets + 1) / 2), dim3(256), 0, (hipStream_t)stream, d_blocks, n, d_data, d_aux, d_quad_scratch, d_block_status);
  else hipLaunchKernelGGL(ht_dec_step1_kernel<1>, dim3(sets), dim3(128), 0, (hipStream_t)stream, d_blocks, n, d_data, d_aux, d_quad_scratch, d_block_status);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

namespace ojphgpu {
// kinds: what the caller knows about the blocks of the range (0 = nothing): bit 0 blocks of at most 64 columns occur,
// bit 1 wider ones, bit 2 reversible ones, bit 3 irreversible ones, bit 4 blocks with SigProp / MagRef passes, bit 6
// blocks of more than 32 columns occur, bit 7 blocks of more than 16, bit 8 "bits 6 and 7 are filled in" -- only with bit 8
// SET does a clear bit 6 / 7 mean that every block is at most 32 / 16 columns wide (opt-in: a caller that builds `kinds`
// from bits 0..4 alone gets one block per wavefront, never the paired kernels, which leave wider blocks alone)
int ht_decode_step2_launch(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n, const uint8_t* d_data,
                           const uint32_t* d_quad_scratch, void* d_coef, uint8_t* d_block_status, int kinds)
{
  if (n == 0) return OJPHGPU_OK;
  if (!d_blocks || !d_data || !d_coef || !d_block_status || !d_quad_scratch) return OJPHGPU_E_INVALID;
  const int tx = (kinds & 16) ? 0 : (kinds & 12) == 4 ? 1 : (kinds & 12) == 8 ? 2 : 0;
  const int wd = (kinds & 3) == 1 ? 1 : 0;
  // two (four) blocks to a wavefront where every block of the range is at most 32 (16) columns wide (kinds bit 6 (7) clear;
  // OJPHGPU_DEC_DUAL=0: never, =2: two at most)
  static const int multi = [] { const char* e = getenv("OJPHGPU_DEC_DUAL"); return e ? atoi(e) : 4; }();
  if (multi && (kinds & 256) && !(kinds & 64) && wd && (tx == 1 || tx == 2)) {
    const uint32_t nb = (!(kinds & 128) && multi >= 4) ? 4u : 2u;
    const dim3 g2(((n + nb - 1) / nb + WAVES - 1) / WAVES), wg2(64 * WAVES);
#define MULTI_LAUNCH(T, NB) hipLaunchKernelGGL((ht_dec_step2_multi_kernel<T, NB>), g2, wg2, 0, (hipStream_t)stream, d_blocks, n, d_data, d_quad_scratch, (uint32_t*)d_coef, d_block_status)
    if (nb == 4) { if (tx == 1) MULTI_LAUNCH(1, 4); else MULTI_LAUNCH(2, 4); }
    else         { if (tx == 1) MULTI_LAUNCH(1, 2); else MULTI_LAUNCH(2, 2); }
#undef MULTI_LAUNCH
    return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
  }
  const dim3 grid((n + WAVES - 1) / WAVES), wg(64 * WAVES);
#define STEP2_LAUNCH(T, W) hipLaunchKernelGGL((ht_dec_step2_kernel<T, W>), grid, wg, 0, (hipStream_t)stream, d_blocks, n, d_data, \
                                              d_quad_scratch, (uint32_t*)d_coef, d_block_status)
  if (wd) { if (tx == 1) STEP2_LAUNCH(1, 1); else if (tx == 2) STEP2_LAUNCH(2, 1); else STEP2_LAUNCH(0, 1); }
  else    { if (tx == 1) STEP2_LAUNCH(1, 0); else if (tx == 2) STEP2_LAUNCH(2, 0); else STEP2_LAUNCH(0, 0); }
#undef STEP2_LAUNCH
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}
}  // namespace ojphgpu

namespace ojphgpu { bool dec_uses_prep(); }
namespace ojphgpu {
// words of the scratch the fused launch needs for n blocks (flags of the chain wavefronts, then the per-block state)
uint64_t ht_decode_fused_state_words(uint32_t n) { return (uint64_t)(((n + 63u) / 64u + 8u + 63u) & ~63u) + TICKET_WORDS; }   // one flag per chain wavefront + the ticket counters
static int dec_fuse_mode()                                // OJPHGPU_DEC_FUSED: 0 never, 1 (default) where it pays, 2 wherever it can
{
  static const int v = [] { const char* e = getenv("OJPHGPU_DEC_FUSED"); return e ? atoi(e) : 1; }();
  return dec_uses_prep() ? 0 : v;
}
bool dec_fuses() { return dec_fuse_mode() != 0; }

// How the fused launch deals n blocks out: workgroups of `wgw` wavefronts, `ch` chains (+ 2 ch partners) in the step-1
// role; every worker wavefront should be resident while the chains run, so the blocks go `per_wave` consecutive ones to
// a wavefront, as few as the chip's wavefront slots allow.  (More blocks than the chip holds at S2_MAX_PER_WAVE: the
// surplus workgroups start when others end and find their rows complete -- slower, never stuck.)
// Shape: workgroups of 12 wavefronts, 4 chains (+ 8 partners) in the step-1 role, two per CU (OJPHGPU_FUSED_SHAPE=0:
// 8 wavefronts, 2 chains, < 40 KB of LDS, four per CU = all 32 wavefront slots of a CU in use -- measured slower, 0.43
// against 0.39 ms for the 8K frame: the chains lose more issue slots to eight wavefronts per SIMD than the workers gain).
struct FusedShape { uint32_t shape, ch, wgw, n1, per_wave, wwgs; };
static FusedShape fused_shape(uint32_t n, uint32_t cus)
{
  static const uint32_t shape = [] { const char* e = getenv("OJPHGPU_FUSED_SHAPE"); return e ? (uint32_t)atoi(e) : 1u; }();
  if (cus == 0) cus = 256;
  FusedShape f;
  f.shape = shape;
  f.ch = shape == 1 ? 4u : 2u; f.wgw = shape == 1 ? 12u : 8u;
  const uint32_t wg_slots = cus * (shape == 1 ? 2u : 4u);
  f.n1 = (n + 64u * f.ch - 1u) / (64u * f.ch);
  const uint32_t waves = (wg_slots > f.n1 ? wg_slots - f.n1 : 1u) * f.wgw;
  uint32_t per_wave = (n + waves - 1u) / waves;
  f.per_wave = per_wave < 1u ? 1u : per_wave > S2_MAX_PER_WAVE ? S2_MAX_PER_WAVE : per_wave;
  f.wwgs = ((n + f.per_wave - 1u) / f.per_wave + f.wgw - 1u) / f.wgw;
  return f;
}
// compute units of a device (the decoder objects ask once, for THEIR device, and hand the number to the calls below)
uint32_t device_cus(int device)
{
  int c = 0;
  if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c <= 0) { (void)hipGetLastError(); c = 256; }
  return (uint32_t)c;
}
// workgroups of the fused launch over n blocks
uint32_t ht_decode_fused_grid(uint32_t n, uint32_t cus) { const FusedShape f = fused_shape(n, cus); return f.n1 + f.wwgs; }

// Does ONE launch pay for these blocks?  Measured (profiles/r03_b_block_sizes*.txt, the C5 / C4 bench lines): it does
// where step 1 is a long latency floor beside an idle chip -- blocks of 64 rows -- and every worker wavefront can be
// resident with a ring per block (<= 5 blocks each: ~25 000 blocks on 256 CUs).  Blocks of 32 rows or fewer have chains
// half as long, and more blocks than that make workers that start when others end: there the separate launches (raw
// step 1, then step 2 beside the small synthesis levels) are faster -- 8K frame in 32x32 blocks 0.67 against 0.80 ms,
// eight 4K frames per step 1.24 against 1.45 ms.  OJPHGPU_DEC_FUSED=2 fuses wherever the launch is able to.
bool ht_decode_fused_pays(uint32_t n, uint32_t max_h, uint32_t cus)
{
  if (dec_fuse_mode() == 0) return false;
  if (fused_shape(n, cus).n1 > (cus ? cus : 256u)) return false;   // (a step-1 workgroup per CU at most: the tickets)
  if (dec_fuse_mode() >= 2) return true;
  return max_h > 32u && fused_shape(n, cus).per_wave <= (uint32_t)S2_RINGS;
}
// step 1 + step 2 of n blocks, all of them at most 64 samples wide, of one wavelet (kinds as in ht_decode_step2_launch)
// and without refinement passes; max_h = the tallest block; epoch: a number that differs from run to run on this scratch;
// d_state: ht_decode_fused_state_words(n) words, zeroed once, and epoch > 0 growing from launch to launch on it (the
// ticket counters carry it); d_blocks: the array ojphgpu_ht_decode_layout laid out, from its first element (the
// 16-bit records of a block are found from its position among the 64 of its chain wavefront); d_block_status: n bytes + the 4-byte RETRY word behind
// them at the next multiple of 4 (== epoch after the run: a wait ran out, decode the blocks again by the separate launches);
// d_host_retry: null, or the device address of a word of mapped host memory that receives the same epoch (ask_for_repeat)
int ht_decode_fused_launch(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n, const uint8_t* d_data, uint32_t* d_quad_scratch,
                           void* d_coef, uint8_t* d_block_status, uint32_t* d_state, uint32_t epoch, uint32_t max_h, int kinds,
                           uint32_t cus, uint32_t* d_host_retry)
{
  if (n == 0) return OJPHGPU_OK;
  if (ensure_tables() != 0) return OJPHGPU_E_HIP;
  if (!d_blocks || !d_data || !d_coef || !d_block_status || !d_quad_scratch || !d_state) return OJPHGPU_E_INVALID;
  const int tx = (kinds & 16) ? 0 : (kinds & 12) == 4 ? 1 : (kinds & 12) == 8 ? 2 : 0;
  if (tx == 0 || (kinds & 3) != 1 || max_h == 0) return OJPHGPU_E_INVALID;
  const uint32_t max_qh = (max_h + 1u) >> 1;
  const FusedShape f = fused_shape(n, cus);
  const uint32_t shape = f.shape, n1 = f.n1, per_wave = f.per_wave, wwgs = f.wwgs, wgw = f.wgw;
  static const uint32_t dbg = [] { const char* e = getenv("OJPHGPU_FUSED_DBG"); return e ? (uint32_t)atoi(e) : 0u; }();
  // how long a worker waits for a chain before it asks for the repeat (OJPHGPU_FUSED_WAIT_MS; ticks of the 100 MHz clock)
  static const uint32_t wait_ticks = [] { const char* e = getenv("OJPHGPU_FUSED_WAIT_MS"); const long ms = e ? atol(e) : 2000; return (uint32_t)((ms < 1 ? 1 : ms > 40000 ? 40000 : ms) * 100000l); }();
  const uint32_t ticket_off = (uint32_t)ht_decode_fused_state_words(n) - TICKET_WORDS;   // the ticket counters: behind the flags
  const dim3 grid(n1 + wwgs), wg(64 * wgw);
#define FUSED_LAUNCH(T, C, W, R) hipLaunchKernelGGL((ht_dec_fused_kernel<T, C, W, R>), grid, wg, 0, (hipStream_t)stream, d_blocks, n, d_data, d_quad_scratch, \
                                                 (uint32_t*)d_coef, d_block_status, d_state, n1, per_wave, max_qh, epoch, dbg, ticket_off, wait_ticks, d_host_retry)
  // a ring per block where the twelve wavefronts' rings fit the LDS the step-1 role needs anyway (OJPHGPU_FUSED_RINGS=1: never)
  static const bool rings = [] { const char* e = getenv("OJPHGPU_FUSED_RINGS"); return !e || atoi(e) != 1; }();
  if (shape == 1 && rings && per_wave <= (uint32_t)S2_RINGS) { if (tx == 1) FUSED_LAUNCH(1, 4, 12, S2_RINGS); else FUSED_LAUNCH(2, 4, 12, S2_RINGS); }
  else if (shape == 1) { if (tx == 1) FUSED_LAUNCH(1, 4, 12, 1); else FUSED_LAUNCH(2, 4, 12, 1); }
  else                 { if (tx == 1) FUSED_LAUNCH(1, 2, 8, 1); else FUSED_LAUNCH(2, 2, 8, 1); }
#undef FUSED_LAUNCH
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}
}  // namespace ojphgpu

namespace ojphgpu {
// words of d_aux a block of the 64-bit sample path needs beyond ojphgpu_ht_decode_aux_words(len1)
uint32_t ht_decode64_extra_aux_words(uint32_t len1) { return ms_words64(len1); }

// the blocks of [d_blocks, d_blocks + n) that are on the 64-bit sample path (cb_desc.reversible bit 2; the others are
// skipped): prep (three flat strings per block in d_aux), step 1 on them, step 2, and -- refine != 0 -- SigProp / MagRef
int ht_decode64_launch(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n, const uint8_t* d_data, uint32_t* d_aux,
                       uint32_t* d_quad_scratch, void* d_coef, uint8_t* d_block_status, int refine)
{
  if (n == 0) return OJPHGPU_OK;
  if (ensure_tables() != 0) return OJPHGPU_E_HIP;
  if (!d_blocks || !d_data || !d_aux || !d_quad_scratch || !d_coef || !d_block_status) return OJPHGPU_E_INVALID;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ht_dec64_prep_kernel, dim3((n + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, s, d_blocks, n, d_data, d_aux);
  const uint32_t sets = (n + 63) / 64;
  hipLaunchKernelGGL((ht_dec_step1_kernel<4, true>), dim3((sets + 3) / 4), dim3(512), 0, s, d_blocks, n, d_data, (const uint32_t*)d_aux,
                     d_quad_scratch, d_block_status);
  hipLaunchKernelGGL(ht_dec64_step2_kernel, dim3((n + W64_WAVES - 1) / W64_WAVES), dim3(64 * W64_WAVES), 0, s, d_blocks, n, d_data,
                     (const uint32_t*)d_aux, (const uint32_t*)d_quad_scratch, (uint32_t*)d_coef, d_block_status);
  if (refine)
    hipLaunchKernelGGL(ht_dec64_refine_kernel, dim3(n), dim3(64), 0, s, d_blocks, n, d_data, (const uint32_t*)d_quad_scratch, (uint32_t*)d_coef,
                       (const uint8_t*)d_block_status);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}
}  // namespace ojphgpu

extern "C" int ojphgpu_ht_decode_step2(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                                        const uint8_t* d_data, const uint32_t* d_quad_scratch, void* d_coef,
                                        uint8_t* d_block_status)
{
  return ojphgpu::ht_decode_step2_launch(stream, d_blocks, n, d_data, d_quad_scratch, d_coef, d_block_status, 0);
}

extern "C" int ojphgpu_ht_decode_refine(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                                         const uint8_t* d_data, const uint32_t* d_quad_scratch, void* d_coef,
                                         const uint8_t* d_block_status)
{
  if (n == 0) return OJPHGPU_OK;
  if (!d_blocks || !d_data || !d_quad_scratch || !d_coef || !d_block_status) return OJPHGPU_E_INVALID;
  hipLaunchKernelGGL(ht_dec_refine_kernel, dim3((n + RWAVES - 1) / RWAVES), dim3(64 * RWAVES), 0, (hipStream_t)stream,
                     d_blocks, n, d_data, d_quad_scratch, (uint32_t*)d_coef, d_block_status);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

extern "C" int ojphgpu_ht_decode(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                                  const uint8_t* d_data, void* d_coef, uint32_t* d_quad_scratch,
                                  uint32_t* d_aux, uint8_t* d_block_status)
{
  int rc = ojphgpu_ht_decode_prep(stream, d_blocks, n, d_data, d_aux);
  if (rc == OJPHGPU_OK) rc = ojphgpu_ht_decode_step1(stream, d_blocks, n, d_data, d_aux, d_quad_scratch, d_block_status);
  if (rc == OJPHGPU_OK) rc = ojphgpu_ht_decode_step2(stream, d_blocks, n, d_data, d_quad_scratch, d_coef, d_block_status);
  if (rc == OJPHGPU_OK) rc = ojphgpu_ht_decode_refine(stream, d_blocks, n, d_data, d_quad_scratch, d_coef, d_block_status);
  // blocks on the 64-bit sample path (descriptor flag; every launch above skipped them)
  if (rc == OJPHGPU_OK) rc = ojphgpu::ht_decode64_launch(stream, d_blocks, n, d_data, d_aux, d_quad_scratch, d_coef, d_block_status, 1);
  return rc;
}


namespace ojphgpu {
int upload_dec_tables(const HtTables& t)
{
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_dec_vlc), t.dec_vlc, sizeof(t.dec_vlc)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_dec_vlc32), t.dec_vlc32, sizeof(t.dec_vlc32)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_dec_uvlc0), t.dec_uvlc0, sizeof(t.dec_uvlc0)) != hipSuccess) return -1;
  return 0;
}
}

#ifdef FUSED_TIMELINE
extern "C" int ojphgpu_debug_fused_timeline(uint32_t* out, uint32_t words, int clear)
{
  if (words > 8192u * 12u) words = 8192u * 12u;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), words * 4u) != hipSuccess) return -1;
  if (out && words == 8192u * 12u && hipMemcpyFromSymbol(out + words, HIP_SYMBOL(g_tl_pub), sizeof(uint32_t) * 1024u * 8u) != hipSuccess) return -1;   // (a caller with room for them)
  if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tl)) != hipSuccess || hipMemset(p, 0, sizeof(uint32_t) * 8192u * 12u) != hipSuccess) return -1; }
  return 0;
}
#endif
