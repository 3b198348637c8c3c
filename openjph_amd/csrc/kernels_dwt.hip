// openjph_amd/csrc/kernels_dwt.hip -- 5/3 (reversible, int32) and 9/7 (irreversible, fp32) DWT
// for gfx950, one launch per decomposition level over every tile-component of a frame.
//
// What the reference does (one image line per call, rolling line buffers):
//   vertical lifting state machine   resolution::push_line / pull_line  ojph_resolution.cpp:547-949
//   per-line horizontal lifting      gen_rev_horz_ana/syn, gen_irv_horz_ana/syn
//                                                                        ojph_transform.cpp:336-852
//   lifting steps / K               param_atk::init_rev53 / init_irv97   ojph_params.cpp:2870-2896
//
// MI355X design (bandwidth-bound, no LDS, no MFMA):
//   * a wavefront owns a strip of 64 column PAIRS (even/odd sample of the same lifting site) and
//     walks down the rows.  Every row is fetched as one coalesced 512-byte segment (8 B / lane).
//   * vertical lifting is a software pipeline in registers: the lane keeps the 4-step lifting
//     state of its two columns (x, a, b, c) and emits one low and one high row per iteration.
//   * horizontal lifting of the emitted rows happens in the same registers: the neighbour
//     column pair is the neighbour lane, fetched with a DPP wave shift (wave_shl:1 / wave_shr:1,
//     a plain VALU move -- no LDS crossbar).  Strips overlap by 2 pairs on each side (halo
//     recomputation) so no inter-wave exchange is needed.
//   * the rows of iteration t+1 are requested before iteration t computes and stores, so the
//     HBM latency of the row stream overlaps the lifting arithmetic of the previous row pair.
//   * the number of row pairs a wavefront walks (its vertical chunk) is chosen per launch
//     (pick_row_pairs): at most 20, so that a large plane takes two or three rounds of workgroups
//     whose reads and writes mix at HBM, down to 8 (analysis) / 4 (synthesis) at the small levels,
//     where the kernel is otherwise bound by the length of the serial walk.
//   * the first analysis level can read the int32 image planes directly (level shift / int ->
//     float conversion of ojph_colour.cpp:238-436 applied in the load), and the last synthesis
//     level can write them (float -> int with rounding + clamp), which removes one full
//     read+write pass over the frame in each direction when no colour transform is used.
//   * the four sub-band rows are written as coalesced 256-byte segments (4 B / lane).
//   Analysis is vertical-then-horizontal, synthesis horizontal-then-vertical, exactly the
//   reference's order -- for 5/3 the rounding makes the order observable.
//   Boundary rule = the reference's: a missing neighbour is replaced by the other neighbour
//   (whole-sample symmetric extension); a 1-sample-long dimension is passed through (even
//   origin) or doubled / halved (odd origin) without the K scaling.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include "../../include/ojphgpu.h"

namespace {

constexpr int HALO = 2;             // column pairs recomputed on each side of a strip
constexpr int VALID = 64 - 2 * HALO; // 60 pairs = 120 columns produced per wavefront
constexpr int MAX_ROW_PAIRS = 20;   // row pairs produced per strip and launch chunk (40 image rows), see pick_row_pairs
constexpr int MIN_ROW_PAIRS = 8, MIN_ROW_PAIRS_INV = 4;

template <bool REV> struct Wv;

template <> struct Wv<true> {       // reversible 5/3 (ojph_params.cpp:2883-2896)
  typedef int T;
  static constexpr bool REV = true, STEPS4 = false, HAS_K = false;
  static constexpr int WARM = 1;    // row pairs a vertical chunk has to start early for its pipeline to be warm (two lifting steps)
  static __device__ __forceinline__ T hK_lo(T v) { return v; }
  static __device__ __forceinline__ T hK_hi(T v) { return v; }
  static __device__ __forceinline__ T a0(T h, T p, T q) { return h - ((p + q) >> 1); }     // predict
  static __device__ __forceinline__ T a1(T l, T p, T q) { return l + ((p + q + 2) >> 2); } // update
  static __device__ __forceinline__ T a2(T h, T, T) { return h; }
  static __device__ __forceinline__ T a3(T l, T, T) { return l; }
  static __device__ __forceinline__ T s0(T l, T p, T q) { return l - ((p + q + 2) >> 2); }
  static __device__ __forceinline__ T s1(T h, T p, T q) { return h + ((p + q) >> 1); }
  static __device__ __forceinline__ T s2(T l, T, T) { return l; }
  static __device__ __forceinline__ T s3(T h, T, T) { return h; }
  static __device__ __forceinline__ T mulK(T v) { return v; }
  static __device__ __forceinline__ T mulKinv(T v) { return v; }
  static __device__ __forceinline__ T dbl(T v) { return v << 1; }
  static __device__ __forceinline__ T halve(T v) { return v >> 1; }
};

template <> struct Wv<false> {      // irreversible 9/7 (ojph_params.cpp:2870-2881)
  typedef float T;
  static constexpr bool REV = false, STEPS4 = true, HAS_K = true;
  static constexpr int WARM = 2;
  // the horizontal analysis scales low by 1 / K, high by K (ojph_transform.cpp:763-775)
  static __device__ __forceinline__ T hK_lo(T v) { return mulKinv(v); }
  static __device__ __forceinline__ T hK_hi(T v) { return mulK(v); }
  // fp32 "add, mul, add" with no contraction: the generic reference build is the bit-exact pin
  static __device__ __forceinline__ T lift(T v, float c, T p, T q) { return __fadd_rn(v, __fmul_rn(c, __fadd_rn(p, q))); }
  static __device__ __forceinline__ T unlift(T v, float c, T p, T q) { return __fsub_rn(v, __fmul_rn(c, __fadd_rn(p, q))); }
  static __device__ __forceinline__ T a0(T h, T p, T q) { return lift(h, (float)-1.586134342059924, p, q); }
  static __device__ __forceinline__ T a1(T l, T p, T q) { return lift(l, (float)-0.052980118572961, p, q); }
  static __device__ __forceinline__ T a2(T h, T p, T q) { return lift(h, (float)0.882911075530934, p, q); }
  static __device__ __forceinline__ T a3(T l, T p, T q) { return lift(l, (float)0.443506852043971, p, q); }
  static __device__ __forceinline__ T s0(T l, T p, T q) { return unlift(l, (float)0.443506852043971, p, q); }
  static __device__ __forceinline__ T s1(T h, T p, T q) { return unlift(h, (float)0.882911075530934, p, q); }
  static __device__ __forceinline__ T s2(T l, T p, T q) { return unlift(l, (float)-0.052980118572961, p, q); }
  static __device__ __forceinline__ T s3(T h, T p, T q) { return unlift(h, (float)-1.586134342059924, p, q); }
  static __device__ __forceinline__ T mulK(T v) { return __fmul_rn(v, (float)1.230174104914001); }
  static __device__ __forceinline__ T mulKinv(T v) { return __fmul_rn(v, __fdiv_rn(1.0f, (float)1.230174104914001)); }
  static __device__ __forceinline__ T dbl(T v) { return __fmul_rn(v, 2.0f); }
  static __device__ __forceinline__ T halve(T v) { return __fmul_rn(v, 0.5f); }
};

// ANY lifting kernel an ATK marker segment describes with up to four steps, and the 5/3 on 64-bit samples: the same
// register pipeline with the steps as launch parameters.  (Part 2 codestreams and components deeper than 26 bits used to
// take one element-wise launch per lifting step and direction -- kernels_lift.hip, 2 N + 2 passes over the plane per level;
// those kernels remain for kernels of more than four steps; levels that transform one direction only are the kernels' MODE.)
// NS = number of steps (compile time: a slot without a step is no instruction at all, not an "add zero" -- which would turn
// a -0.0f into +0.0f); the steps sit in the order the direction applies them: analysis step NS-1 first, updating the odd
// (high-pass) samples; synthesis step 0 first, updating the even ones -- application index i alternates exactly as the
// pipeline's slots a0..a3 / s0..s3 do (param_atk, ojph_params.cpp:2654-2896; gen_rev_vert_step32 / 64, gen_irv_vert_step,
// the horizontal functions, ojph_transform.cpp:209-852).
template <typename TT, int NS> struct WvGen {
  typedef TT T;
  static constexpr bool REV = !std::is_floating_point<TT>::value, STEPS4 = NS > 2, HAS_K = std::is_floating_point<TT>::value;
  static constexpr int WARM = NS > 2 ? 2 : 1;
  int a[4], b[4], e[4]; float A[4]; float K, Kinv; int hswap;   // hswap: the horizontal analysis of an odd number of steps scales the OTHER sub-sequence by 1 / K (:765-777)
  template <int I, bool ADD> __device__ __forceinline__ T step(T v, T p, T q) const {
    if constexpr (I >= NS) return v;
    else if constexpr (REV) {                               // (b + a (l + r)) >> e, the shift counted modulo the width (see kernels_lift.hip)
      const T d = (T)(((T)b[I] + (T)a[I] * (T)(p + q)) >> (e[I] & (int)(8 * sizeof(T) - 1)));
      return ADD ? (T)(v + d) : (T)(v - d);
    } else {
      const float m = __fmul_rn(A[I], __fadd_rn(p, q));
      return ADD ? __fadd_rn(v, m) : __fsub_rn(v, m);
    }
  }
  __device__ __forceinline__ T a0(T h, T p, T q) const { return step<0, true>(h, p, q); }
  __device__ __forceinline__ T a1(T l, T p, T q) const { return step<1, true>(l, p, q); }
  __device__ __forceinline__ T a2(T h, T p, T q) const { return step<2, true>(h, p, q); }
  __device__ __forceinline__ T a3(T l, T p, T q) const { return step<3, true>(l, p, q); }
  __device__ __forceinline__ T s0(T l, T p, T q) const { return step<0, false>(l, p, q); }
  __device__ __forceinline__ T s1(T h, T p, T q) const { return step<1, false>(h, p, q); }
  __device__ __forceinline__ T s2(T l, T p, T q) const { return step<2, false>(l, p, q); }
  __device__ __forceinline__ T s3(T h, T p, T q) const { return step<3, false>(h, p, q); }
  __device__ __forceinline__ T mulK(T v) const { if constexpr (HAS_K) return __fmul_rn(v, K); else return v; }
  __device__ __forceinline__ T mulKinv(T v) const { if constexpr (HAS_K) return __fmul_rn(v, Kinv); else return v; }
  __device__ __forceinline__ T hK_lo(T v) const { if constexpr (HAS_K) return __fmul_rn(v, hswap ? K : Kinv); else return v; }
  __device__ __forceinline__ T hK_hi(T v) const { if constexpr (HAS_K) return __fmul_rn(v, hswap ? Kinv : K); else return v; }
  __device__ __forceinline__ T dbl(T v) const { if constexpr (HAS_K) return __fmul_rn(v, 2.0f); else return (T)(v * 2); }
  __device__ __forceinline__ T halve(T v) const { if constexpr (HAS_K) return __fmul_rn(v, 0.5f); else return (T)(v >> 1); }
};

// value of the same register in lane+1 / lane-1 (DPP wave shifts; the end lanes keep their own
// value, they are halo lanes whose results are never stored)
__device__ __forceinline__ int lane_next(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xF, 0xF, false); }   // wave_shl:1
__device__ __forceinline__ int lane_prev(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false); }   // wave_shr:1
__device__ __forceinline__ float lane_next(float v) { return __int_as_float(lane_next(__float_as_int(v))); }
__device__ __forceinline__ float lane_prev(float v) { return __int_as_float(lane_prev(__float_as_int(v))); }
__device__ __forceinline__ long long lane_next(long long v) { return (long long)(((unsigned long long)(unsigned)lane_next((int)(v >> 32)) << 32) | (unsigned)lane_next((int)v)); }
__device__ __forceinline__ long long lane_prev(long long v) { return (long long)(((unsigned long long)(unsigned)lane_prev((int)(v >> 32)) << 32) | (unsigned)lane_prev((int)v)); }

// neighbour selection with the "missing -> use the other one" rule
// CHK = false: the caller knows every neighbour exists (interior of the plane) and the select folds away
template <bool CHK = true, typename T>
__device__ __forceinline__ T pick(bool e, T v, T other) { return CHK ? (e ? v : other) : v; }

template <typename T> struct Pair { T l, h; };

// geometry of one plane as seen by one lane
struct Geo {
  int w, h, ox, oy;      // ox/oy = 1 when the plane origin is at an odd coordinate
  int j;                 // column-pair index of this lane in "u = x + ox" space
  bool eL, eH, eLn, eHp; // existence of own low/high column and of the neighbours' (j+1 low, j-1 high)
  bool store;            // lane is in the valid (non-halo) zone of its strip
  int xc;                // first column of the lane's 2-sample row load, clamped into the row
  bool from_y, from_x;   // clamped load: the low sample arrives in .y / the high sample arrives in .x
  bool inner;            // wave-uniform: every lane of the strip has all its horizontal neighbours
};

__device__ __forceinline__ bool col_exists(int x, int w) { return x >= 0 && x < w; }

template <class WP, bool CHK = true>
__device__ __forceinline__ void horz_analysis(const WP& w, typename WP::T& vl, typename WP::T& vh, const Geo& g)
{
  typedef typename WP::T T;
  if (CHK && g.w == 1) { if (g.ox) vh = w.dbl(vh); return; }   // ojph_transform.cpp:405-410, :777-782
  T nl = lane_next(vl);
  vh = w.a0(vh, pick<CHK>(g.eL, vl, nl), pick<CHK>(g.eLn, nl, vl));
  T ph = lane_prev(vh);
  vl = w.a1(vl, pick<CHK>(g.eHp, ph, vh), pick<CHK>(g.eH, vh, ph));
  if (WP::STEPS4) {
    nl = lane_next(vl);
    vh = w.a2(vh, pick<CHK>(g.eL, vl, nl), pick<CHK>(g.eLn, nl, vl));
    ph = lane_prev(vh);
    vl = w.a3(vl, pick<CHK>(g.eHp, ph, vh), pick<CHK>(g.eH, vh, ph));
  }
  if (WP::HAS_K) { vl = w.hK_lo(vl); vh = w.hK_hi(vh); }          // ojph_transform.cpp:763-775
}

template <class WP, bool CHK = true>
__device__ __forceinline__ void horz_synthesis(const WP& w, typename WP::T& vl, typename WP::T& vh, const Geo& g)
{
  typedef typename WP::T T;
  if (CHK && g.w == 1) { if (g.ox) vh = w.halve(vh); return; } // ojph_transform.cpp:583-588, :844-849
  if (WP::HAS_K) { vl = w.mulK(vl); vh = w.mulKinv(vh); }   // :797-809
  T ph = lane_prev(vh);
  vl = w.s0(vl, pick<CHK>(g.eHp, ph, vh), pick<CHK>(g.eH, vh, ph));
  T nl = lane_next(vl);
  vh = w.s1(vh, pick<CHK>(g.eL, vl, nl), pick<CHK>(g.eLn, nl, vl));
  if (WP::STEPS4) {
    ph = lane_prev(vh);
    vl = w.s2(vl, pick<CHK>(g.eHp, ph, vh), pick<CHK>(g.eH, vh, ph));
    nl = lane_next(vl);
    vh = w.s3(vh, pick<CHK>(g.eL, vl, nl), pick<CHK>(g.eLn, nl, vl));
  }
}

__device__ __forceinline__ Geo make_geo(const ojphgpu_dwt_desc& d, int strip_x, int lane, bool plain_x = false, bool plain_y = false)
{
  Geo g;
  g.w = (int)d.w; g.h = (int)d.h; g.ox = (d.x_even || plain_x) ? 0 : 1; g.oy = (d.y_even || plain_y) ? 0 : 1;
  g.j = strip_x * VALID - HALO + lane;
  int xl = 2 * g.j - g.ox, xh = xl + 1;
  g.eL = col_exists(xl, g.w); g.eH = col_exists(xh, g.w);
  g.eLn = col_exists(xl + 2, g.w); g.eHp = col_exists(xh - 2, g.w);
  g.store = lane >= HALO && lane < 64 - HALO;
  g.xc = max(min(xl, g.w - 2), 0);
  g.from_y = xl > g.xc; g.from_x = xl < g.xc;
  g.inner = __all(g.eL && g.eH && g.eLn && g.eHp) != 0;
  return g;
}

// sample conversion between the int32 image planes and the working type, fused into the first
// analysis / last synthesis level (ojph_colour.cpp:238-275 rev, :388-436 to float, :316-386 to int)
struct Conv { int bit_depth, is_signed; };

template <bool REV> struct Cv;
template <> struct Cv<true> {
  static __device__ __forceinline__ int from_image(int v, const Conv& c) { return v - (c.is_signed ? 0 : (1 << (c.bit_depth - 1))); }
  static __device__ __forceinline__ int to_image(int v, const Conv& c) { return v + (c.is_signed ? 0 : (1 << (c.bit_depth - 1))); }
};
template <> struct Cv<false> {
  static __device__ __forceinline__ float from_image(int v, const Conv& c) {
    const float mul = __uint_as_float((uint32_t)(127 - c.bit_depth) << 23);            // 2^-B
    const int half = c.is_signed ? 0 : (1 << (c.bit_depth - 1));
    return __fmul_rn((float)(v - half), mul);
  }
  static __device__ __forceinline__ int to_image(float f, const Conv& c) {
    const int neg_limit = (int)0x80000000 >> (32 - c.bit_depth);
    const float mul = __uint_as_float((uint32_t)(127 + c.bit_depth) << 23);            // 2^B
    const float up = -(float)neg_limit, low = (float)neg_limit;
    const int s_up = 0x7FFFFFFF >> (32 - c.bit_depth), s_low = neg_limit;
    const int half = c.is_signed ? 0 : (1 << (c.bit_depth - 1));
    const float t = __fmul_rn(f, mul);
    int v = (int)__fadd_rn(t, t >= 0.0f ? 0.5f : -0.5f);    // ojph_round: truncation of t +- 0.5
    v = t >= low ? v : s_low;
    v = t < up ? v : s_up;
    return v + half;
  }
};

// two-sample accesses that only promise the alignment of one sample
template <typename E> struct Vec2 { typedef E type __attribute__((ext_vector_type(2), aligned(sizeof(E) < 4 ? sizeof(E) : 4))); };

// element type of the plane a kernel variant reads / writes: IMG = 0 the arena (T), 32 an int32
// image, 16 / 8 a 16- / 8-bit image (two's complement for signed components, else unsigned)
template <int IMG, typename T> struct ImgElem { typedef T type; };
template <typename T> struct ImgElem<32, T> { typedef int type; };
template <typename T> struct ImgElem<16, T> { typedef short type; };
template <typename T> struct ImgElem<8, T> { typedef signed char type; };

// Row loads are UNCONDITIONAL: every lane fetches two adjacent samples from a column clamped into
// the row (and the callers clamp the row into the plane), and what a lane fetched is interpreted
// only when it is consumed one iteration later (unpack).  No branch around a load and no use of the
// loaded registers next to it means no wait at the load: the fetch of row pair t+1 is in flight
// while pair t is lifted.  Samples that do not exist are never read by the lifting steps (pick()).
// (What a load returns stays in the registers it arrived in until unpack(): two 16- or 8-bit samples as the ONE register
// the load filled, two 32-bit samples as two.  A field extracted next to the load -- "r.y = v >> 16", or a copy that merges
// the two arms of the w == 1 case -- is a use of the loaded register and puts a wait right behind the load.)
template <typename E, bool SMALL = (sizeof(E) < 4)> struct Raw;
template <typename E> struct Raw<E, true> { uint32_t p; };      // both samples as loaded: the first in the low bits (w == 1: the one sample, zero-extended)
template <typename E> struct Raw<E, false> { E x, y; };         // (w == 1: both are the one sample)
// W1: the plane is one column wide (a property of the plane, so of the whole workgroup: the kernels run their pipeline loop
// in a W1 = true or a W1 = false instantiation, and the loads of the latter have no branch around them at all)
template <typename E, bool W1>
__device__ __forceinline__ Raw<E> load_raw(const void* __restrict__ rowp, const Geo& g)
{
  const E* row = (const E*)rowp;
  Raw<E> r;
  if constexpr (sizeof(E) < 4) {
    typedef typename std::conditional<sizeof(E) == 2, unsigned short, unsigned char>::type UE;
    typedef typename std::conditional<sizeof(E) == 2, uint32_t, unsigned short>::type PK __attribute__((aligned(sizeof(E))));
    if constexpr (W1) r.p = (uint32_t)*reinterpret_cast<const UE*>(row);  // the only sample is column 0
    else r.p = (uint32_t)*reinterpret_cast<const PK*>(row + g.xc);
  } else {
    if constexpr (W1) { r.x = row[0]; r.y = r.x; }
    else { const typename Vec2<E>::type v = *reinterpret_cast<const typename Vec2<E>::type*>(row + g.xc); r.x = v.x; r.y = v.y; }
  }
  return r;
}

// the integer sample a raw container element holds
template <int IMG, typename E>
__device__ __forceinline__ int sample_of(E v, const Conv& cv)
{
  if (IMG == 16) return cv.is_signed ? (int)v : (int)(unsigned short)v;     // unsigned components occupy the full container
  if (IMG == 8) return cv.is_signed ? (int)v : (int)(unsigned char)v;
  return (int)v;
}

// IMG: the row came from an image plane and is converted here
template <class WP, int IMG, typename E>
__device__ __forceinline__ Pair<typename WP::T> unpack(const Raw<E>& v, const Geo& g, const Conv& cv)
{
  Pair<typename WP::T> p;
  E vx, vy;
  if constexpr (sizeof(E) < 4) {
    vx = (E)(v.p & ((1u << (8 * sizeof(E))) - 1u));
    vy = g.w == 1 ? vx : (E)(v.p >> (8 * sizeof(E)));
  } else { vx = v.x; vy = v.y; }
  const auto a = g.from_y ? vy : vx, b = g.from_x ? vx : vy;
  if constexpr (IMG != 0) { p.l = Cv<WP::REV>::from_image(sample_of<IMG>(a, cv), cv); p.h = Cv<WP::REV>::from_image(sample_of<IMG>(b, cv), cv); }
  else { p.l = (typename WP::T)a; p.h = (typename WP::T)b; }
  return p;
}

// Forward / inverse component transform of one sample triple (ojph_colour.cpp:443-571: gen_rct_forward /
// _backward, gen_ict_forward / _backward), on values that went through from_image / are about to go through
// to_image -- the same operations in the same order as the stand-alone conversion kernels (kernels_convert.hip)
constexpr float CT_ALPHA_RF = 0.299f, CT_ALPHA_GF = 0.587f, CT_ALPHA_BF = 0.114f;
template <bool REV> struct Ct;
template <> struct Ct<true> {
  static __device__ __forceinline__ void fwd(int r, int g, int b, int& y, int& cb, int& cr) { y = (r + (g << 1) + b) >> 2; cb = b - g; cr = r - g; }
  static __device__ __forceinline__ void inv(int y, int cb, int cr, int& r, int& g, int& b) { g = y - ((cb + cr) >> 2); r = cr + g; b = cb + g; }
};
template <> struct Ct<false> {
  static __device__ __forceinline__ void fwd(float r, float g, float b, float& y, float& cb, float& cr) {
    const float beta_cb = (float)(0.5 / (1 - (double)CT_ALPHA_BF)), beta_cr = (float)(0.5 / (1 - (double)CT_ALPHA_RF));
    y = __fadd_rn(__fadd_rn(__fmul_rn(CT_ALPHA_RF, r), __fmul_rn(CT_ALPHA_GF, g)), __fmul_rn(CT_ALPHA_BF, b));
    cb = __fmul_rn(beta_cb, __fsub_rn(b, y)); cr = __fmul_rn(beta_cr, __fsub_rn(r, y));
  }
  static __device__ __forceinline__ void inv(float y, float cb, float cr, float& r, float& g, float& b) {
    const float g_cb2g = (float)(2.0 * (double)CT_ALPHA_BF * (1.0 - (double)CT_ALPHA_BF) / (double)CT_ALPHA_GF);
    const float g_cr2g = (float)(2.0 * (double)CT_ALPHA_RF * (1.0 - (double)CT_ALPHA_RF) / (double)CT_ALPHA_GF);
    const float g_cb2b = (float)(2.0 * (1.0 - (double)CT_ALPHA_BF)), g_cr2r = (float)(2.0 * (1.0 - (double)CT_ALPHA_RF));
    g = __fsub_rn(__fsub_rn(y, __fmul_rn(g_cr2g, cr)), __fmul_rn(g_cb2g, cb));
    r = __fadd_rn(y, __fmul_rn(g_cr2r, cr));
    b = __fadd_rn(y, __fmul_rn(g_cb2b, cb));
  }
};

// NC = 3: the rows of the three colour planes become the rows of Y, Cb, Cr
template <class WP, int IMG, int NC, typename E>
__device__ __forceinline__ void unpack_all(const Raw<E>* v, const Geo& g, const Conv* cv, Pair<typename WP::T>* out)
{
#pragma unroll
  for (int k = 0; k < NC; ++k) out[k] = unpack<WP, IMG>(v[k], g, cv[k]);
  if constexpr (NC == 3) {
    const Pair<typename WP::T> r = out[0], gg = out[1], b = out[2];
    Ct<WP::REV>::fwd(r.l, gg.l, b.l, out[0].l, out[1].l, out[2].l);
    Ct<WP::REV>::fwd(r.h, gg.h, b.h, out[0].h, out[1].h, out[2].h);
  }
}

// Memory order inside one iteration of the forward kernel: consume what the previous iteration fetched,
// THEN issue the stores of the previous iteration's results, THEN the fetches of the next iteration,
// then compute.  gfx9 counts loads and stores in one in-order counter; with the stores placed just
// before the fetches, the wait for those fetches one iteration later also covers the stores and
// nothing is ever waited for right after it was issued.  arrived() pins that wait to the top of the
// iteration on every control path (otherwise a path that skips a fetched row leaves the wait to the
// next write of the same registers -- which comes right after the stores).
template <typename E>
__device__ __forceinline__ void arrived(const Raw<E>& a, const Raw<E>& b)
{
  if constexpr (sizeof(E) < 4) asm volatile("" :: "v"(a.p), "v"(b.p));
  else asm volatile("" :: "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
}

// ---------------------------------------------------------------------------------------------
// forward: plane (or image plane, IMG) -> LL, HL, LH, HH
// ---------------------------------------------------------------------------------------------
// NC = 1: one plane per wavefront strip.  NC = 3: the three colour planes of a tile at once -- the rows of R, G, B
// are read once, turned into Y, Cb, Cr rows in registers (RCT / ICT) and run through three vertical pipelines, so a
// colour-transformed frame needs no conversion pass over HBM (descs[3 z .. 3 z + 2] = the planes' descriptors, which
// share their geometry).
// Which (strip group, vertical chunk) a workgroup takes.  The dispatcher deals workgroups to the eight XCDs in turn (linear id
// mod 8), each XCD with an L2 of its own; with the grid's natural order the chunk below a chunk sits on the NEXT XCD, and the
// rows the two share (the vertical halo: two row pairs on either side for the synthesis, a fifth of a 20-pair chunk) come in
// through two L2s.  XCD_REMAP (bit 16 of the row_pairs argument): the workgroups an XCD receives walk a contiguous band of
// chunk rows instead -- a permutation of the plane's workgroups, so whatever the dispatcher really does, every chunk is taken
// exactly once.
constexpr int XCD_REMAP = 1 << 16;
__device__ __forceinline__ void dwt_block_coords(int row_pairs_arg, int& bx, int& by)
{
  bx = (int)blockIdx.x; by = (int)blockIdx.y;
  const uint32_t T = gridDim.x * gridDim.y;
  if (!(row_pairs_arg & XCD_REMAP) || T < 16u) return;
  const uint32_t L = blockIdx.x + gridDim.x * blockIdx.y;
  const uint32_t o = (T * blockIdx.z) & 7u;                  // the plane's first workgroup is number T z of the launch
  const uint32_t r = (L + o) & 7u, q = T >> 3, rem = T & 7u;  // r: the XCD this workgroup is (most likely) on
  uint32_t start = 0;
  for (uint32_t k = 0; k < r; ++k) start += q + ((((k - o) & 7u) < rem) ? 1u : 0u);   // workgroups of this plane on the XCDs before r
  const uint32_t V = start + (L >> 3);
  bx = (int)(V % gridDim.x); by = (int)(V / gridDim.x);
}

// U = row pairs per trip of the pipeline loop (1 or 2).  A trip ends in ONE wait for everything it has in flight -- loads and
// stores share vmcnt and are not counted down in order against each other, so a wait for the rows of the next pair drains
// the stores as well -- and with U = 2 a wavefront has twice the bytes under way per wait: 4 row loads and 8 sub-band row
// stores (forward), 8 and 4 (inverse).  Same arithmetic in the same order; only when rows are requested and stored changes.
// MODE (the levels of a DFS decomposition that transform ONE direction, resolution::push_line's HORZ_TRX / VERT_TRX,
// ojph_resolution.cpp:290-300, :556-600; general lifting policies only): 0 = both directions; 1 = along the rows only -- every
// row is lifted horizontally and goes, whole, to row y of LL | HL; 2 = along the columns only -- the lane's two columns are
// plain neighbours (no horizontal step, no x parity), a low row goes to LL, a high row to LH, at its own columns.
template <class WP, int IMG, int NC, int U = 1, int MODE = 0>
__global__ __launch_bounds__(256) void dwt_forward_kernel(const ojphgpu_dwt_desc* __restrict__ descs,
                                                          uint32_t* __restrict__ base32,
                                                          const void* __restrict__ image, Conv cv, int row_pairs_arg, const WP w)
{
  typedef typename WP::T T;
  constexpr bool REV = WP::REV;
  (void)REV;
  typedef typename ImgElem<IMG, T>::type E;                        // element type of the source rows
  typedef Raw<E> RawRow;
  // a DWT launch is short and the next stage waits for it: when it shares the SIMDs with the long
  // block-coder launch of the side stream, its wavefronts go first
  __builtin_amdgcn_s_setprio(2);
  const ojphgpu_dwt_desc d = descs[blockIdx.z * NC];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and the compiler knows it
  int bx, by;
  dwt_block_coords(row_pairs_arg, bx, by);
  const int row_pairs = row_pairs_arg & 0xFFFF;
  const int strip_x = bx * 4 + wave;
  if (d.w == 0 || d.h == 0) return;
  static_assert(MODE == 0 || (IMG == 0 && NC == 1), "one-direction levels exist below the top level of general-lifting components only");
  const Geo g = make_geo(d, strip_x, lane, MODE == 2, MODE == 1);
  const int npx = (g.w + g.ox + 1) >> 1, npy = (g.h + g.oy + 1) >> 1;
  if (strip_x * VALID >= npx) return;
  const int i0 = by * row_pairs;
  if (i0 >= npy) return;
  const int i1 = min(i0 + row_pairs, npy);

  const char* src[NC]; T* ll[NC]; T* hl[NC]; T* lh[NC]; T* hh[NC];
  Conv cvs[NC];                                            // every plane's own sample format (the colour planes of a damaged SIZ may differ: ojph_tile.cpp:332-437)
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const ojphgpu_dwt_desc dk = k ? descs[blockIdx.z * NC + k] : d;
    cvs[k] = cv;
    if (IMG && dk.reserved) { cvs[k].bit_depth = (int)(dk.reserved & 0xFFu); cvs[k].is_signed = (int)((dk.reserved >> 8) & 1u); }
    // (offsets count 32-bit arena elements whatever T is: a 64-bit plane starts on an even element)
    src[k] = IMG ? (const char*)image + dk.src_off * sizeof(E) : (const char*)(base32 + dk.src_off);
    ll[k] = (T*)(base32 + dk.ll_off); hl[k] = (T*)(base32 + dk.hl_off); lh[k] = (T*)(base32 + dk.lh_off); hh[k] = (T*)(base32 + dk.hh_off);
  }
  const size_t sp = (size_t)d.src_pitch * sizeof(E);
  const int h = g.h, oy = g.oy;
  auto exL = [&](int t) { int y = 2 * t - oy; return y >= 0 && y < h; };
  auto exH = [&](int t) { int y = 2 * t + 1 - oy; return y >= 0 && y < h; };
  auto ldrow_w = [&](int y, RawRow* r, auto w1) {          // image row y of every plane, clamped into the plane
    const size_t o = (size_t)min(max(y, 0), h - 1) * sp;
#pragma unroll
    for (int k = 0; k < NC; ++k) r[k] = load_raw<E, decltype(w1)::value>(src[k] + o, g);
  };
  auto put = [&](int k, int t, bool low_row, T vl, T vh) {   // one transformed row of plane k -> its two sub-bands
    if (!g.store) return;
    const int r = low_row ? t - oy : t;                    // row index inside the sub-band
    if constexpr (MODE == 2) {                             // both columns into the same band, at their own positions
      T* band = low_row ? ll[k] : lh[k];
      const uint32_t bp = low_row ? d.ll_pitch : d.lh_pitch;
      if (g.eL) band[(size_t)r * bp + 2 * g.j] = vl;
      if (g.eH) band[(size_t)r * bp + 2 * g.j + 1] = vh;
    } else {
      T* lo = low_row ? ll[k] : lh[k]; T* hi = low_row ? hl[k] : hh[k];
      const uint32_t lop = low_row ? d.ll_pitch : d.lh_pitch, hip = low_row ? d.hl_pitch : d.hh_pitch;
      if (g.eL) lo[(size_t)r * lop + (g.j - g.ox)] = vl;
      if (g.eH) hi[(size_t)r * hip + g.j] = vh;
    }
  };

  if constexpr (MODE == 1) {                               // rows only: no vertical state, two rows per trip
    const int y1 = min(2 * i1, h);
    auto rows = [&](auto w1) {
      for (int y = 2 * i0; y < y1; y += 2) {
        RawRow ra[NC], rb[NC]; Pair<T> xa[NC], xb[NC];
        ldrow_w(y, ra, w1); ldrow_w(y + 1, rb, w1);
        unpack_all<WP, IMG, NC>(ra, g, cvs, xa); unpack_all<WP, IMG, NC>(rb, g, cvs, xb);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          horz_analysis<WP>(w, xa[k].l, xa[k].h, g); horz_analysis<WP>(w, xb[k].l, xb[k].h, g);
          put(k, y, true, xa[k].l, xa[k].h);
          if (y + 1 < y1) put(k, y + 1, true, xb[k].l, xb[k].h);
        }
      }
    };
    if (g.w == 1) rows(std::true_type()); else rows(std::false_type());
    return;
  }

  if (h == 1) {                                            // ojph_resolution.cpp:604-634, :688-708
    if (i0 > 0) return;
    RawRow r0[NC]; Pair<T> x[NC];
    if (g.w == 1) ldrow_w(0, r0, std::true_type()); else ldrow_w(0, r0, std::false_type());
    unpack_all<WP, IMG, NC>(r0, g, cvs, x);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (oy != 0) { x[k].l = w.dbl(x[k].l); x[k].h = w.dbl(x[k].h); }
      if constexpr (MODE != 2) horz_analysis<WP>(w, x[k].l, x[k].h, g);
      put(k, 0, oy == 0, x[k].l, x[k].h);
    }
    return;
  }

  // vertical software pipeline over row pairs t; see file header
  auto pipeline = [&](auto w1) {
  auto ldrow = [&](int y, RawRow* r) { ldrow_w(y, r, w1); };
  const int t0 = max(i0 - WP::WARM, 0);
  Pair<T> xl[NC], a[NC], ap[NC], b[NC], bp[NC], c[NC], cp[NC];   // x[2t], a[t], a[t-1], b[t], b[t-1], c[t-1], c[t-2]
  Pair<T> out_lo[U][NC], out_hi[U][NC];  // transformed rows of pair out_t, stored one trip later
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    xl[k].l = xl[k].h = 0; a[k] = ap[k] = b[k] = bp[k] = c[k] = cp[k] = xl[k];
#pragma unroll
    for (int u = 0; u < U; ++u) out_lo[u][k] = out_hi[u][k] = xl[k];
  }
  int out_t[U]; bool has_lo[U], has_hi[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { out_t[u] = 0; has_lo[u] = has_hi[u] = false; }
  RawRow rh[U][NC], rn[U][NC];
  ldrow(2 * t0 - oy, rh[0]);
  unpack_all<WP, IMG, NC>(rh[0], g, cvs, xl);
#pragma unroll
  for (int u = 0; u < U; ++u) { ldrow(2 * (t0 + u) + 1 - oy, rh[u]); ldrow(2 * (t0 + u) + 2 - oy, rn[u]); }   // rows of the first trip
  for (int t = t0; t <= i1; t += U) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < NC; ++k) arrived(rh[u][k], rn[u][k]);
    Pair<T> xh[U][NC], xn[U][NC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unpack_all<WP, IMG, NC>(rh[u], g, cvs, xh[u]);
      unpack_all<WP, IMG, NC>(rn[u], g, cvs, xn[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (has_lo[u]) put(k, out_t[u], true, out_lo[u][k].l, out_lo[u][k].h);
        if (has_hi[u]) put(k, out_t[u], false, out_hi[u][k].l, out_hi[u][k].h);
      }
      has_lo[u] = has_hi[u] = false;
    }
    if (t + U <= i1) {                                     // request the rows of the next trip now
#pragma unroll
      for (int u = 0; u < U; ++u) { ldrow(2 * (t + U + u) + 1 - oy, rh[u]); ldrow(2 * (t + U + u) + 2 - oy, rn[u]); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tt = t + u;
      if (u != 0 && tt > i1) break;
      // interior of the plane (every row tt-2 .. tt+1 exists, no strip edge): the selects fold away
      auto lift = [&](auto chk) {
        constexpr bool CHK = decltype(chk)::value;
        const bool eLt = exL(tt), eHt = exH(tt), eLn = exL(tt + 1);
        const bool eHp = exH(tt - 1), eLp = exL(tt - 1), eHpp = exH(tt - 2);
        const bool emit = tt - 1 >= i0 && tt - 1 < i1;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          // a[t]
          ap[k] = a[k];
          a[k].l = w.a0(xh[u][k].l, pick<CHK>(eLt, xl[k].l, xn[u][k].l), pick<CHK>(eLn, xn[u][k].l, xl[k].l));
          a[k].h = w.a0(xh[u][k].h, pick<CHK>(eLt, xl[k].h, xn[u][k].h), pick<CHK>(eLn, xn[u][k].h, xl[k].h));
          // b[t]
          const Pair<T> bo = b[k];                // b[t-1]
          b[k].l = w.a1(xl[k].l, pick<CHK>(eHp, ap[k].l, a[k].l), pick<CHK>(eHt, a[k].l, ap[k].l));
          b[k].h = w.a1(xl[k].h, pick<CHK>(eHp, ap[k].h, a[k].h), pick<CHK>(eHt, a[k].h, ap[k].h));
          bp[k] = bo;
          // c[t-1]
          cp[k] = c[k];
          c[k].l = w.a2(ap[k].l, pick<CHK>(eLp, bp[k].l, b[k].l), pick<CHK>(eLt, b[k].l, bp[k].l));
          c[k].h = w.a2(ap[k].h, pick<CHK>(eLp, bp[k].h, b[k].h), pick<CHK>(eLt, b[k].h, bp[k].h));
          // d[t-1]
          const T dl = w.a3(bp[k].l, pick<CHK>(eHpp, cp[k].l, c[k].l), pick<CHK>(eHp, c[k].l, cp[k].l));
          const T dh = w.a3(bp[k].h, pick<CHK>(eHpp, cp[k].h, c[k].h), pick<CHK>(eHp, c[k].h, cp[k].h));
          if (emit) {
            if (!CHK || eLp) {                                   // ojph_resolution.cpp:674-675
              out_lo[u][k].l = w.mulKinv(dl); out_lo[u][k].h = w.mulKinv(dh);
              if constexpr (MODE != 2) horz_analysis<WP, CHK>(w, out_lo[u][k].l, out_lo[u][k].h, g);
            }
            if (!CHK || eHp) {                                   // :663-664
              out_hi[u][k].l = w.mulK(c[k].l); out_hi[u][k].h = w.mulK(c[k].h);
              if constexpr (MODE != 2) horz_analysis<WP, CHK>(w, out_hi[u][k].l, out_hi[u][k].h, g);
            }
          }
        }
        if (emit) { out_t[u] = tt - 1; has_lo[u] = !CHK || eLp; has_hi[u] = !CHK || eHp; }
      };
      if (g.inner && 2 * (tt - 2) - oy >= 0 && 2 * (tt + 1) - oy < h) lift(std::false_type());
      else lift(std::true_type());
#pragma unroll
      for (int k = 0; k < NC; ++k) xl[k] = xn[u][k];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (has_lo[u]) put(k, out_t[u], true, out_lo[u][k].l, out_lo[u][k].h);
      if (has_hi[u]) put(k, out_t[u], false, out_hi[u][k].l, out_hi[u][k].h);
    }
  };
  if (g.w == 1) pipeline(std::true_type()); else pipeline(std::false_type());
}

struct __attribute__((aligned(4))) I2 { int x, y; };        // 8-byte access that only promises dword alignment

// one reconstructed row of a plane: a[0] = the lane's low (even-site) sample, a[1] the high one, as image integers
// (IMG) or working values
// A reconstructed sample may lie just outside its nominal range (the reference's float -> integer conversion
// rounds 127.6 to 128 after its range test, ojph_colour.cpp:316-386: 256 for an 8-bit component).  In int32 samples
// that value is handed over as it is, like the reference's line_buf; a narrower container SATURATES at its own
// range instead of wrapping, as the reference's image writers do when they pack samples into 8 / 16 bits.
template <int IMG>
__device__ __forceinline__ int fit_container(int v, const Conv& cv)
{
  if (IMG == 32 || IMG == 0) return v;
  const int hi = cv.is_signed ? (1 << (IMG - 1)) - 1 : (1 << IMG) - 1, lo = cv.is_signed ? -(1 << (IMG - 1)) : 0;
  return min(max(v, lo), hi);
}

template <bool REV, int IMG>
__device__ __forceinline__ void store_image_pair(void* __restrict__ rowp, const Geo& g, int a, int b, const Conv& cv)
{
  a = fit_container<IMG>(a, cv); b = fit_container<IMG>(b, cv);
  const int xl = 2 * g.j - g.ox;
  typedef typename ImgElem<IMG, int>::type E;
  E* row = (E*)rowp;
  if (IMG == 32) {
    if (g.eL && g.eH) { I2 v; v.x = a; v.y = b; *reinterpret_cast<I2*>(row + xl) = v; }
    else if (g.eL) row[xl] = (E)a;
    else if (g.eH) row[xl + 1] = (E)b;
  } else {
    if (g.eL && g.eH) { typename Vec2<E>::type v; v.x = (E)a; v.y = (E)b; *reinterpret_cast<typename Vec2<E>::type*>(row + xl) = v; }
    else if (g.eL) row[xl] = (E)a;
    else if (g.eH) row[xl + 1] = (E)b;
  }
}

template <class WP, int IMG>
__device__ __forceinline__ void store_pair(void* __restrict__ rowp, const Geo& g, typename WP::T l, typename WP::T h,
                                           const Conv& cv)
{
  typedef typename WP::T T;
  constexpr bool REV = WP::REV;
  if (!g.store) return;
  const int xl = 2 * g.j - g.ox;
  if constexpr (IMG != 0) store_image_pair<REV, IMG>(rowp, g, Cv<REV>::to_image(l, cv), Cv<REV>::to_image(h, cv), cv);
  else {
    T* row = (T*)rowp;
    if (g.ox == 0 && g.eL && g.eH) {
      typedef T V2 __attribute__((ext_vector_type(2)));
      V2 v; v.x = l; v.y = h;
      *reinterpret_cast<V2*>(row + xl) = v;
    } else {
      if (g.eL) row[xl] = l;
      if (g.eH) row[xl + 1] = h;
    }
  }
}

// one reconstructed row of all NC planes; NC = 3: Y, Cb, Cr -> R, G, B on the way out
template <class WP, int IMG, int NC>
__device__ __forceinline__ void store_rows(char* const* dst, size_t off, const Geo& g, const Pair<typename WP::T>* v, const Conv* cv)
{
  typedef typename WP::T T;
  constexpr bool REV = WP::REV;
  if constexpr (NC == 3) {
    if (!g.store) return;
    T rl, gl, bl, rh, gh, bh;
    Ct<REV>::inv(v[0].l, v[1].l, v[2].l, rl, gl, bl);
    Ct<REV>::inv(v[0].h, v[1].h, v[2].h, rh, gh, bh);
    store_image_pair<REV, IMG>(dst[0] + off, g, Cv<REV>::to_image(rl, cv[0]), Cv<REV>::to_image(rh, cv[0]), cv[0]);
    store_image_pair<REV, IMG>(dst[1] + off, g, Cv<REV>::to_image(gl, cv[1]), Cv<REV>::to_image(gh, cv[1]), cv[1]);
    store_image_pair<REV, IMG>(dst[2] + off, g, Cv<REV>::to_image(bl, cv[2]), Cv<REV>::to_image(bh, cv[2]), cv[2]);
  } else {
#pragma unroll
    for (int k = 0; k < NC; ++k) store_pair<WP, IMG>(dst[k] + off, g, v[k].l, v[k].h, cv[k]);
  }
}

// ---------------------------------------------------------------------------------------------
// inverse: LL, HL, LH, HH -> plane (or image plane, IMG); NC as in the forward kernel
// ---------------------------------------------------------------------------------------------
template <class WP, int IMG, int NC, int U = 1, int MODE = 0>
__global__ __launch_bounds__(256) void dwt_inverse_kernel(const ojphgpu_dwt_desc* __restrict__ descs,
                                                          uint32_t* __restrict__ base32,
                                                          void* __restrict__ image, Conv cv, int row_pairs_arg, const WP w)
{
  typedef typename WP::T T;
  constexpr bool REV = WP::REV;
  (void)REV;
  // a DWT launch is short and the next stage waits for it: when it shares the SIMDs with the long
  // block-coder launch of the side stream, its wavefronts go first
  __builtin_amdgcn_s_setprio(2);
  const ojphgpu_dwt_desc d = descs[blockIdx.z * NC];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and the compiler knows it
  int bx, by;
  dwt_block_coords(row_pairs_arg, bx, by);
  const int row_pairs = row_pairs_arg & 0xFFFF;
  const int strip_x = bx * 4 + wave;
  if (d.w == 0 || d.h == 0) return;
  static_assert(MODE == 0 || (IMG == 0 && NC == 1), "one-direction levels exist below the top level of general-lifting components only");
  const Geo g = make_geo(d, strip_x, lane, MODE == 2, MODE == 1);
  const int npx = (g.w + g.ox + 1) >> 1, npy = (g.h + g.oy + 1) >> 1;
  if (strip_x * VALID >= npx) return;
  const int i0 = by * row_pairs;
  if (i0 >= npy) return;
  const int i1 = min(i0 + row_pairs, npy);

  typedef typename ImgElem<IMG, T>::type E;                        // element type of the destination rows
  char* dst[NC]; const T* ll[NC]; const T* hl[NC]; const T* lh[NC]; const T* hh[NC];
  Conv cvs[NC];                                            // every plane's own sample format (ojph_tile.cpp:439-518 converts component by component)
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const ojphgpu_dwt_desc dk = k ? descs[blockIdx.z * NC + k] : d;
    cvs[k] = cv;
    if (IMG && dk.reserved) { cvs[k].bit_depth = (int)(dk.reserved & 0xFFu); cvs[k].is_signed = (int)((dk.reserved >> 8) & 1u); }
    dst[k] = IMG ? (char*)image + dk.src_off * sizeof(E) : (char*)(base32 + dk.src_off);
    ll[k] = (const T*)(base32 + dk.ll_off); hl[k] = (const T*)(base32 + dk.hl_off); lh[k] = (const T*)(base32 + dk.lh_off); hh[k] = (const T*)(base32 + dk.hh_off);
  }
  const size_t dp = (size_t)d.src_pitch * sizeof(E);
  const int h = g.h, oy = g.oy;
  auto exL = [&](int t) { int y = 2 * t - oy; return y >= 0 && y < h; };
  auto exH = [&](int t) { int y = 2 * t + 1 - oy; return y >= 0 && y < h; };
  // raw sub-band samples of the lane's column pair in the low (LL|HL) or high (LH|HH) row of pair t, every plane
  auto fetch = [&](int t, bool low_row, bool ex, Pair<T>* p) {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      p[k].l = p[k].h = 0;
      if (!ex) continue;
      const int r = low_row ? t - oy : t;
      if constexpr (MODE == 2) {                            // both columns from the same band
        const T* band = low_row ? ll[k] : lh[k];
        const uint32_t bp = low_row ? d.ll_pitch : d.lh_pitch;
        if (g.eL) p[k].l = band[(size_t)r * bp + 2 * g.j];
        if (g.eH) p[k].h = band[(size_t)r * bp + 2 * g.j + 1];
      } else {
        const T* lo = low_row ? ll[k] : lh[k]; const T* hi = low_row ? hl[k] : hh[k];
        const uint32_t lop = low_row ? d.ll_pitch : d.lh_pitch, hip = low_row ? d.hl_pitch : d.hh_pitch;
        if (g.eL) p[k].l = lo[(size_t)r * lop + (g.j - g.ox)];
        if (g.eH) p[k].h = hi[(size_t)r * hip + g.j];
      }
    }
  };

  if constexpr (MODE == 1) {                               // rows only: row y of LL | HL -> row y of the plane, two rows per trip
    const int y1 = min(2 * i1, h);
    for (int y = 2 * i0; y < y1; y += 2) {
      Pair<T> xa[NC], xb[NC];
      fetch(y, true, true, xa); fetch(y + 1, true, y + 1 < y1, xb);
#pragma unroll
      for (int k = 0; k < NC; ++k) { horz_synthesis<WP>(w, xa[k].l, xa[k].h, g); horz_synthesis<WP>(w, xb[k].l, xb[k].h, g); }
      store_rows<WP, IMG, NC>(dst, (size_t)y * dp, g, xa, cvs);
      if (y + 1 < y1) store_rows<WP, IMG, NC>(dst, (size_t)(y + 1) * dp, g, xb, cvs);
    }
    return;
  }

  if (h == 1) {                                            // ojph_resolution.cpp:794-829, :900-923
    if (i0 > 0) return;
    Pair<T> x[NC];
    fetch(0, oy == 0, true, x);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if constexpr (MODE != 2) horz_synthesis<WP>(w, x[k].l, x[k].h, g);
      if (oy != 0) { x[k].l = w.halve(x[k].l); x[k].h = w.halve(x[k].h); }
    }
    store_rows<WP, IMG, NC>(dst, 0, g, x, cvs);
    return;
  }

  // The pipeline loop, in two instantiations.  W1 = false (every plane but the one-column ones): the sub-band rows are fetched
  // UNCONDITIONALLY, from a row and a column clamped into the band (every band of a plane of at least 2 x 2 samples has
  // samples), and what does not exist becomes zero when the fetched registers are consumed a trip later -- no branch and no
  // instruction touches a loaded register next to its load, so nothing waits there.  (With the loads under "if (exists)" the
  // compiler merged them with the zero they replace and, depending on the instantiation, put a wait behind every one of
  // them: eight round trips per trip where one is needed.)  W1 = true: the conditional loads, as before.
  auto pipeline = [&](auto w1) {
  constexpr bool W1 = decltype(w1)::value;
  const int nlc = ((g.ox + g.w + 1) >> 1) - ((g.ox + 1) >> 1), nlr = ((oy + h + 1) >> 1) - ((oy + 1) >> 1);   // low columns / rows of the plane
  const int col_l = MODE == 2 ? min(max(2 * g.j, 0), g.w - 1) : min(max(g.j - g.ox, 0), max(nlc - 1, 0));
  const int col_h = MODE == 2 ? min(max(2 * g.j + 1, 0), g.w - 1) : min(max(g.j, 0), max(g.w - nlc - 1, 0));
  auto fetch_any = [&](int t, bool low_row, Pair<T>* p) {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const T* lo = low_row ? ll[k] : lh[k]; const T* hi = MODE == 2 ? lo : (low_row ? hl[k] : hh[k]);
      const uint32_t lop = low_row ? d.ll_pitch : d.lh_pitch, hip = MODE == 2 ? lop : (low_row ? d.hl_pitch : d.hh_pitch);
      const int rows = low_row ? nlr : h - nlr;
      const int r = min(max(low_row ? t - oy : t, 0), rows - 1);
      p[k].l = lo[(size_t)r * lop + col_l];
      p[k].h = hi[(size_t)r * hip + col_h];
    }
  };
  auto request = [&](int t, Pair<T>* plo, Pair<T>* phi) {
    if constexpr (W1) { fetch(t, true, exL(t), plo); fetch(t, false, exH(t), phi); }
    else { fetch_any(t, true, plo); fetch_any(t, false, phi); }
  };
  auto existing = [&](int t, bool low_row, const Pair<T>& v) {   // a fetched pair as the lifting steps take it
    if constexpr (W1) return v;
    else { const bool ex = low_row ? exL(t) : exH(t); Pair<T> o; o.l = (ex && g.eL) ? v.l : (T)0; o.h = (ex && g.eH) ? v.h : (T)0; return o; }
  };
  const int t0 = max(i0 - WP::WARM, 0);
  Pair<T> z; z.l = z.h = 0;
  Pair<T> c[NC], cp[NC], b[NC], bp[NC], a[NC], ap[NC], xL[NC], xLp[NC];
  // c[t], c[t-1], b[t], b[t-1], a[t-1], a[t-2], xL[t-1], xL[t-2]
#pragma unroll
  for (int k = 0; k < NC; ++k) c[k] = cp[k] = b[k] = bp[k] = a[k] = ap[k] = xL[k] = xLp[k] = z;
  Pair<T> nlo[U][NC], nhi[U][NC];
#pragma unroll
  for (int u = 0; u < U; ++u) request(t0 + u, nlo[u], nhi[u]);      // sub-band rows of the first trip
  for (int t = t0; t <= i1 + 1; t += U) {
    Pair<T> in_lo[U][NC], in_hi[U][NC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < NC; ++k) { in_lo[u][k] = existing(t + u, true, nlo[u][k]); in_hi[u][k] = existing(t + u, false, nhi[u][k]); }
    if (t + U <= i1 + 1) {                                  // request the next trip's rows now
#pragma unroll
      for (int u = 0; u < U; ++u) request(t + U + u, nlo[u], nhi[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tt = t + u;
      if (u != 0 && tt > i1 + 1) break;
      // interior of the plane (every row tt-2 .. tt exists, no strip edge): the selects fold away
      auto lift = [&](auto chk) {
        constexpr bool CHK = decltype(chk)::value;
        const bool eLt = exL(tt), eHt = exH(tt), eLp = exL(tt - 1), eHp = exH(tt - 1);
        const bool eLpp = exL(tt - 2), eHpp = exH(tt - 2);
        Pair<T> xh[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          Pair<T> dd = in_lo[u][k], cc = in_hi[u][k];
          cp[k] = c[k]; c[k] = z;
          if (!CHK || eLt) { if constexpr (MODE != 2) horz_synthesis<WP, CHK>(w, dd.l, dd.h, g); dd.l = w.mulK(dd.l); dd.h = w.mulK(dd.h); } else dd = z;   // :855-856
          if (!CHK || eHt) { if constexpr (MODE != 2) horz_synthesis<WP, CHK>(w, cc.l, cc.h, g); c[k].l = w.mulKinv(cc.l); c[k].h = w.mulKinv(cc.h); }    // :871-872
          // b[t]
          bp[k] = b[k];
          b[k].l = w.s0(dd.l, pick<CHK>(eHp, cp[k].l, c[k].l), pick<CHK>(eHt, c[k].l, cp[k].l));
          b[k].h = w.s0(dd.h, pick<CHK>(eHp, cp[k].h, c[k].h), pick<CHK>(eHt, c[k].h, cp[k].h));
          // a[t-1]
          ap[k] = a[k];
          a[k].l = w.s1(cp[k].l, pick<CHK>(eLp, bp[k].l, b[k].l), pick<CHK>(eLt, b[k].l, bp[k].l));
          a[k].h = w.s1(cp[k].h, pick<CHK>(eLp, bp[k].h, b[k].h), pick<CHK>(eLt, b[k].h, bp[k].h));
          // xL[t-1]
          xLp[k] = xL[k];
          xL[k].l = w.s2(bp[k].l, pick<CHK>(eHpp, ap[k].l, a[k].l), pick<CHK>(eHp, a[k].l, ap[k].l));
          xL[k].h = w.s2(bp[k].h, pick<CHK>(eHpp, ap[k].h, a[k].h), pick<CHK>(eHp, a[k].h, ap[k].h));
          // xH[t-2]
          xh[k].l = w.s3(ap[k].l, pick<CHK>(eLpp, xLp[k].l, xL[k].l), pick<CHK>(eLp, xL[k].l, xLp[k].l));
          xh[k].h = w.s3(ap[k].h, pick<CHK>(eLpp, xLp[k].h, xL[k].h), pick<CHK>(eLp, xL[k].h, xLp[k].h));
        }
        if (tt - 2 >= i0 && tt - 2 < i1 && (!CHK || eHpp))
          store_rows<WP, IMG, NC>(dst, (size_t)(2 * (tt - 2) + 1 - oy) * dp, g, xh, cvs);
        if (tt - 1 >= i0 && tt - 1 < i1 && (!CHK || eLp))
          store_rows<WP, IMG, NC>(dst, (size_t)(2 * (tt - 1) - oy) * dp, g, xL, cvs);
      };
      if (g.inner && 2 * (tt - 2) - oy >= 0 && 2 * tt + 1 - oy < h) lift(std::false_type());
      else lift(std::true_type());
    }
  }
  };
  if (g.w == 1) pipeline(std::true_type()); else pipeline(std::false_type());
}

// Vertical chunk of a strip per workgroup; the result is wave-uniform per launch.  Tall chunks
// re-read fewer halo rows, but a launch whose wavefronts all fit on the chip at once runs them in
// lockstep -- everybody loads, then everybody stores -- and HBM sees bursts of one direction.  With
// chunks of at most 20 row pairs a large plane takes 2-3 rounds of workgroups that drift apart, reads
// and writes mix, and the level-1 launches gain 8-18 % (8K frame, 16K tiled image, batches of 4K
// frames; A/B of 8..96 row pairs in one box visit: 20 is the best, below 16 the extra wavefronts of
// the lower levels take issue slots from the block coder running beside them).
int pick_row_pairs(uint32_t n, uint32_t max_w, uint32_t max_h, bool synthesis)
{
  const uint32_t npx = (max_w + 2) >> 1, npy = (max_h + 2) >> 1;
  const uint64_t strips = (uint64_t)((npx + VALID - 1) / VALID) * n;
  const uint64_t want_waves = 4096;
  uint64_t chunks = (want_waves + strips - 1) / strips;
  if (chunks < 1) chunks = 1;
  uint64_t rp = (npy + chunks - 1) / chunks;
  // the small levels are bound by the length of a wavefront's serial walk (a memory round trip per row pair): the synthesis
  // launches, which run with little beside them, go down to 4 row pairs per chunk (8K frame: all levels 0.2535 -> 0.2465 ms,
  // 4K RGB: 0.122 -> 0.115); the analysis launches share the chip with the block coder of the top resolution and keep 8 --
  // more, shorter workgroups there measured no gain (profiles/r05_a_small_levels.txt).  OJPHGPU_DWT_RP_MIN sets both.
  static const uint64_t rp_env = [] { const char* e = getenv("OJPHGPU_DWT_RP_MIN"); const int v = e ? atoi(e) : 0; return (uint64_t)(v >= 2 && v <= 20 ? v : 0); }();
  const uint64_t rp_min = rp_env ? rp_env : synthesis ? (uint64_t)MIN_ROW_PAIRS_INV : (uint64_t)MIN_ROW_PAIRS;
  rp = rp_min >= 4 ? (rp + 3) & ~3ull : (rp + 1) & ~1ull;
  if (rp < rp_min) rp = rp_min;
  if (rp > (uint64_t)MAX_ROW_PAIRS) rp = MAX_ROW_PAIRS;
  return (int)rp;
}

// the row_pairs argument of a launch: the height, and whether the workgroups take their chunks XCD by XCD (dwt_block_coords;
// OJPHGPU_DWT_XCD=0 keeps the grid's own order)
int row_pairs_arg(int rp)
{
  static const bool xcd = [] { const char* e = getenv("OJPHGPU_DWT_XCD"); return !e || atoi(e) != 0; }();
  return rp | (xcd ? XCD_REMAP : 0);
}

dim3 dwt_grid(uint32_t n, uint32_t max_w, uint32_t max_h, int rp)
{
  uint32_t npx = (max_w + 2) >> 1, npy = (max_h + 2) >> 1;
  uint32_t sx = (npx + VALID - 1) / VALID;
  return dim3((sx + 3) / 4, (npy + rp - 1) / rp, n);
}

// workgroups of `fn` (256 threads, no dynamic LDS) the device holds at once
int resident_workgroups(const void* fn)
{
  static std::mutex mu; static std::map<const void*, int> known;
  std::lock_guard<std::mutex> lock(mu);
  auto it = known.find(fn);
  if (it != known.end()) return it->second;
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  return known[fn] = per_cu * cus;
}

// row pairs per chunk for a latency-bound launch (the colour-fused top level): the height whose workgroups need the fewest
// (rounds x rows walked per workgroup); a chunk walks its row pairs plus about three of halo / warm-up
int fit_rounds(const void* fn, uint32_t planes, uint32_t max_w, uint32_t max_h)
{
  const uint64_t cap = (uint64_t)resident_workgroups(fn);
  int best = 8; uint64_t best_cost = ~0ull;
  for (int rp = 4; rp <= 24; rp += 2) {
    const dim3 g = dwt_grid(planes, max_w, max_h, rp);
    const uint64_t wgs = (uint64_t)g.x * g.y * g.z, rounds = (wgs + cap - 1) / cap;
    const uint64_t cost = rounds * (uint64_t)(rp + 3);
    if (cost < best_cost) { best_cost = cost; best = rp; }
  }
  return best;
}

// container: 0 = no image (arena planes only), 32 = int32 image samples, 16 / 8 = 16- / 8-bit image samples;
// nc = 3: the descriptors come in triples (the colour planes of a tile), see the kernels
#ifndef DWT_TRIP_DEFAULT
#define DWT_TRIP_DEFAULT 2
#endif
template <bool FWD>
int launch(void* stream, int reversible, const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
           void* d_base, void* d_image, Conv cv, int container = 32, int nc = 1)
{
  if (n == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  if (!d_descs || !d_base || (nc != 1 && nc != 3) || n % (uint32_t)nc || (nc == 3 && !d_image)) return OJPHGPU_E_INVALID;
  int rp = pick_row_pairs(n, max_w, max_h, !FWD);
  {
    // the synthesis re-reads two sub-band row pairs above and below each vertical chunk (the inverse top level moves 1.28 x
    // its algorithmic bytes at 20 row pairs per chunk): longer chunks there trade that against the burstiness shorter
    // chunks were introduced for -- OJPHGPU_DWT_RP_INV / OJPHGPU_DWT_RP_FWD override the cap of the large launches
    static const int rp_inv = [] { const char* e = getenv("OJPHGPU_DWT_RP_INV"); const int v = e ? atoi(e) : 0; return v >= 4 && v <= 256 ? v : 0; }();
    static const int rp_fwd = [] { const char* e = getenv("OJPHGPU_DWT_RP_FWD"); const int v = e ? atoi(e) : 0; return v >= 4 && v <= 256 ? v : 0; }();
    const int cap = FWD ? rp_fwd : rp_inv;
    if (cap && rp == MAX_ROW_PAIRS) rp = cap;
  }
  // a colour wavefront carries three pipelines: a third of the wavefronts of the plain kernel, each three times as long, at
  // 94-139 registers (3-5 wavefronts per SIMD) -- a launch bound by latency, not by HBM, whose duration is (rounds of
  // workgroups the chip needs) x (rows a workgroup walks).  The chunk height is chosen per launch so that the workgroups
  // fill whole rounds (fit_rounds; the extra halo rows of short chunks are cheap here: the image side is 1-2 bytes per
  // sample).  4K RGB frame, top level: forward 0.085 -> 0.071 ms (8 row pairs per chunk were 1.2 rounds of its 1 024
  // resident workgroups, 12 are 0.8), profiles/r05_a_colour_chunks.txt.  OJPHGPU_DWT_RP_COLOUR fixes the height.
  static const int rp3 = [] { const char* e = getenv("OJPHGPU_DWT_RP_COLOUR"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 64 ? v : 0; }();
  if (nc == 3) rp = rp3 ? rp3 : 8;
  dim3 grid = dwt_grid(n / (uint32_t)nc, max_w, max_h, rp);
  hipStream_t s = (hipStream_t)stream;
  // two row pairs per trip of the pipeline loop for the one-plane launches (see the kernels' U; OJPHGPU_DWT_TRIP=1: one)
  static const int trip = [] { const char* e = getenv("OJPHGPU_DWT_TRIP"); const int v = e ? atoi(e) : DWT_TRIP_DEFAULT; return v == 2 ? 2 : 1; }();
#define OJPH_LAUNCH(K, REV, IMG, NC, TP) do { auto fn = K<Wv<REV>, IMG, NC, 1>; \
    if (NC == 1 && trip == 2) fn = K<Wv<REV>, IMG, NC, (NC == 1 ? 2 : 1)>; \
    if (NC == 3 && !rp3) { rp = fit_rounds((const void*)fn, n / 3u, max_w, max_h); grid = dwt_grid(n / 3u, max_w, max_h, rp); } \
    hipLaunchKernelGGL(fn, grid, dim3(256), 0, s, d_descs, (uint32_t*)d_base, d_image, cv, row_pairs_arg(rp), Wv<REV>()); } while (0)
#define OJPH_LAUNCH_NC(K, REV, IMG, TP) do { if (nc == 3) OJPH_LAUNCH(K, REV, IMG, 3, TP); else OJPH_LAUNCH(K, REV, IMG, 1, TP); } while (0)
#define OJPH_LAUNCH_IMG(K, REV, TP) do { if (!d_image) OJPH_LAUNCH(K, REV, 0, 1, TP); else if (container == 16) OJPH_LAUNCH_NC(K, REV, 16, TP); \
                                         else if (container == 8) OJPH_LAUNCH_NC(K, REV, 8, TP); else OJPH_LAUNCH_NC(K, REV, 32, TP); } while (0)
  if (FWD) { if (reversible) OJPH_LAUNCH_IMG(dwt_forward_kernel, true, int); else OJPH_LAUNCH_IMG(dwt_forward_kernel, false, float); }
  else { if (reversible) OJPH_LAUNCH_IMG(dwt_inverse_kernel, true, int); else OJPH_LAUNCH_IMG(dwt_inverse_kernel, false, float); }
#undef OJPH_LAUNCH_IMG
#undef OJPH_LAUNCH_NC
#undef OJPH_LAUNCH
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

// a general lifting kernel (ojphgpu_lift: steps in synthesis order) as the pipeline's policy for one direction of use
template <typename TT, int NS>
WvGen<TT, NS> make_policy(const ojphgpu_lift* k, bool synthesis)
{
  WvGen<TT, NS> w;
  memset(&w, 0, sizeof(w));
  for (int i = 0; i < NS; ++i) {
    const ojphgpu_lift_step& st = k->steps[synthesis ? i : NS - 1 - i];      // application order: analysis runs the steps backwards
    w.a[i] = st.a; w.b[i] = st.b; w.e[i] = st.e; w.A[i] = st.A;
  }
  w.K = k->K; w.Kinv = 1.0f / k->K;                                          // (fp32, as gen_irv_horz_ana computes it: host code, no contraction)
  w.hswap = (!synthesis && (NS & 1)) ? 1 : 0;
  return w;
}

template <typename TT, int NS>
int launch_general(hipStream_t s, const ojphgpu_lift* k, const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
                   void* d_base, bool synthesis)
{
  const int rp = pick_row_pairs(n, max_w, max_h, synthesis);
  const dim3 grid = dwt_grid(n, max_w, max_h, rp);
  const WvGen<TT, NS> w = make_policy<TT, NS>(k, synthesis);
  // (two row pairs per trip, like the 5/3 and 9/7 launches; 64-bit samples keep one: twice the registers per sample)
  constexpr int U = sizeof(TT) > 4 ? 1 : DWT_TRIP_DEFAULT;
#define OJPH_GENERAL(M, UU) do { \
    if (synthesis) hipLaunchKernelGGL((dwt_inverse_kernel<WvGen<TT, NS>, 0, 1, UU, M>), grid, dim3(256), 0, s, d_descs, (uint32_t*)d_base, (void*)nullptr, Conv{ 0, 0 }, row_pairs_arg(rp), w); \
    else hipLaunchKernelGGL((dwt_forward_kernel<WvGen<TT, NS>, 0, 1, UU, M>), grid, dim3(256), 0, s, d_descs, (uint32_t*)d_base, (const void*)nullptr, Conv{ 0, 0 }, row_pairs_arg(rp), w); } while (0)
  if (k->horz && k->vert) OJPH_GENERAL(0, U);
  else if (k->horz) OJPH_GENERAL(1, 1);                    // (a DFS level that transforms the rows only / the columns only)
  else OJPH_GENERAL(2, 1);
#undef OJPH_GENERAL
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

// the same with an image side (IMG = container bits): both directions only
template <typename TT, int NS, int IMG>
int launch_general_image(hipStream_t s, const ojphgpu_lift* k, const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
                         void* d_base, void* d_image, Conv cv, bool synthesis)
{
  const int rp = pick_row_pairs(n, max_w, max_h, synthesis);
  const dim3 grid = dwt_grid(n, max_w, max_h, rp);
  const WvGen<TT, NS> w = make_policy<TT, NS>(k, synthesis);
  constexpr int U = DWT_TRIP_DEFAULT;
  if (synthesis) hipLaunchKernelGGL((dwt_inverse_kernel<WvGen<TT, NS>, IMG, 1, U, 0>), grid, dim3(256), 0, s, d_descs, (uint32_t*)d_base, d_image, cv, row_pairs_arg(rp), w);
  else hipLaunchKernelGGL((dwt_forward_kernel<WvGen<TT, NS>, IMG, 1, U, 0>), grid, dim3(256), 0, s, d_descs, (uint32_t*)d_base, (const void*)d_image, cv, row_pairs_arg(rp), w);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}
template <typename TT, int IMG>
int launch_general_image_steps(hipStream_t s, const ojphgpu_lift* k, const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
                               void* d_base, void* d_image, Conv cv, bool synthesis)
{
  switch (k->num_steps) {
    case 1: return launch_general_image<TT, 1, IMG>(s, k, d_descs, n, max_w, max_h, d_base, d_image, cv, synthesis);
    case 2: return launch_general_image<TT, 2, IMG>(s, k, d_descs, n, max_w, max_h, d_base, d_image, cv, synthesis);
    case 3: return launch_general_image<TT, 3, IMG>(s, k, d_descs, n, max_w, max_h, d_base, d_image, cv, synthesis);
    case 4: return launch_general_image<TT, 4, IMG>(s, k, d_descs, n, max_w, max_h, d_base, d_image, cv, synthesis);
  }
  return OJPHGPU_E_INVALID;
}
int general_image(void* stream, const ojphgpu_lift* k, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                  uint32_t max_w, uint32_t max_h, void* d_image, void* d_base, int container_bits, bool synthesis)
{
  if (!k || !params || !d_descs || !d_base || !d_image || !k->horz || !k->vert || k->num_steps < 1 || k->num_steps > 4 || (k->elem != 0 && k->elem != 2))
    return OJPHGPU_E_INVALID;
  if (params->bit_depth == 0 || params->bit_depth > (uint32_t)(container_bits == 32 ? 31 : container_bits)) return OJPHGPU_E_INVALID;
  if (n == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  hipStream_t s = (hipStream_t)stream;
  const Conv cv{ (int)params->bit_depth, (int)params->is_signed };
#define OJPH_GI(TT) (container_bits == 16 ? launch_general_image_steps<TT, 16>(s, k, d_descs, n, max_w, max_h, d_base, d_image, cv, synthesis) : \
                     container_bits == 8 ? launch_general_image_steps<TT, 8>(s, k, d_descs, n, max_w, max_h, d_base, d_image, cv, synthesis) : \
                     container_bits == 32 ? launch_general_image_steps<TT, 32>(s, k, d_descs, n, max_w, max_h, d_base, d_image, cv, synthesis) : OJPHGPU_E_INVALID)
  return k->elem == 0 ? OJPH_GI(int) : OJPH_GI(float);
#undef OJPH_GI
}

template <typename TT>
int launch_general_steps(hipStream_t s, const ojphgpu_lift* k, const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
                         void* d_base, bool synthesis)
{
  switch (k->num_steps) {
    case 1: return launch_general<TT, 1>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
    case 2: return launch_general<TT, 2>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
    case 3: return launch_general<TT, 3>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
    case 4: return launch_general<TT, 4>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
  }
  return OJPHGPU_E_INVALID;
}

}  // namespace

namespace ojphgpu {
// One level of a GENERAL lifting kernel (kernels_lift.hip's ojphgpu_dwt_forward / _inverse_general) through the register
// pipeline of this file: both directions transformed, one to four lifting steps, int32 / int64 / float planes -- one
// launch that reads the plane once and writes its four sub-bands once (or the reverse), where the element-wise form takes
// 2 N + 2 launches, each a full pass.  -> OJPHGPU_E_INVALID when the kernel does not fit (the caller keeps the other form).
bool dwt_general_pipeline_fits(const ojphgpu_lift* k)
{
  return k && (k->horz || k->vert) && k->num_steps >= 1 && k->num_steps <= 4 && k->elem <= 2;
}
int dwt_general_pipeline(void* stream, const ojphgpu_lift* k, const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
                         void* d_base, bool synthesis)
{
  if (!dwt_general_pipeline_fits(k) || !d_descs || !d_base) return OJPHGPU_E_INVALID;
  if (n == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  hipStream_t s = (hipStream_t)stream;
  if (k->elem == 0) return launch_general_steps<int>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
  if (k->elem == 1) return launch_general_steps<long long>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
  return launch_general_steps<float>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
}
}  // namespace ojphgpu

extern "C" int ojphgpu_dwt_forward_general_image(void* stream, const ojphgpu_lift* kernel, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                                 uint32_t n, uint32_t max_w, uint32_t max_h, const void* d_image, void* d_base, int container_bits)
{
  return general_image(stream, kernel, params, d_descs, n, max_w, max_h, const_cast<void*>(d_image), d_base, container_bits, false);
}
extern "C" int ojphgpu_dwt_inverse_general_image(void* stream, const ojphgpu_lift* kernel, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                                 uint32_t n, uint32_t max_w, uint32_t max_h, void* d_image, void* d_base, int container_bits)
{
  return general_image(stream, kernel, params, d_descs, n, max_w, max_h, d_image, d_base, container_bits, true);
}

extern "C" int ojphgpu_dwt_forward(void* stream, int reversible, const ojphgpu_dwt_desc* d_descs,
                                    uint32_t n, uint32_t max_w, uint32_t max_h, void* d_base)
{
  return launch<true>(stream, reversible, d_descs, n, max_w, max_h, d_base, nullptr, Conv{ 0, 0 });
}

extern "C" int ojphgpu_dwt_inverse(void* stream, int reversible, const ojphgpu_dwt_desc* d_descs,
                                    uint32_t n, uint32_t max_w, uint32_t max_h, void* d_base)
{
  return launch<false>(stream, reversible, d_descs, n, max_w, max_h, d_base, nullptr, Conv{ 0, 0 });
}

extern "C" int ojphgpu_dwt_forward_image(void* stream, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                          uint32_t n, uint32_t max_w, uint32_t max_h, const int32_t* d_image, void* d_base)
{
  if (!params || !d_image || params->color_transform || params->bit_depth == 0 || params->bit_depth > 31) return OJPHGPU_E_INVALID;
  return launch<true>(stream, (int)params->reversible, d_descs, n, max_w, max_h, d_base, const_cast<int32_t*>(d_image),
                      Conv{ (int)params->bit_depth, (int)params->is_signed });
}

extern "C" int ojphgpu_dwt_inverse_image(void* stream, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                          uint32_t n, uint32_t max_w, uint32_t max_h, int32_t* d_image, void* d_base)
{
  if (!params || !d_image || params->color_transform || params->bit_depth == 0 || params->bit_depth > 31) return OJPHGPU_E_INVALID;
  return launch<false>(stream, (int)params->reversible, d_descs, n, max_w, max_h, d_base, d_image,
                       Conv{ (int)params->bit_depth, (int)params->is_signed });
}

// the same with the image samples in 16-bit containers (int16 for signed components, else uint16)
extern "C" int ojphgpu_dwt_forward_image16(void* stream, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                            uint32_t n, uint32_t max_w, uint32_t max_h, const uint16_t* d_image, void* d_base)
{
  if (!params || !d_image || params->color_transform || params->bit_depth == 0 || params->bit_depth > 16) return OJPHGPU_E_INVALID;
  return launch<true>(stream, (int)params->reversible, d_descs, n, max_w, max_h, d_base, const_cast<uint16_t*>(d_image),
                      Conv{ (int)params->bit_depth, (int)params->is_signed }, 16);
}

extern "C" int ojphgpu_dwt_inverse_image16(void* stream, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                            uint32_t n, uint32_t max_w, uint32_t max_h, uint16_t* d_image, void* d_base)
{
  if (!params || !d_image || params->color_transform || params->bit_depth == 0 || params->bit_depth > 16) return OJPHGPU_E_INVALID;
  return launch<false>(stream, (int)params->reversible, d_descs, n, max_w, max_h, d_base, d_image,
                       Conv{ (int)params->bit_depth, (int)params->is_signed }, 16);
}

// the general form: container_bits = 32 | 16 | 8; colour != 0: the descriptors come in triples -- the three
// colour planes of a tile -- and the component transform (RCT for the 5/3, ICT for the 9/7, ojph_colour.cpp:443-571)
// is applied in the loads of the first analysis level / the stores of the last synthesis level
extern "C" int ojphgpu_dwt_forward_image_ex(void* stream, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                             uint32_t n, uint32_t max_w, uint32_t max_h, const void* d_image, void* d_base,
                                             int container_bits, int colour)
{
  if (!params || !d_image || params->bit_depth == 0 || params->bit_depth > (uint32_t)(container_bits == 32 ? 31 : container_bits)) return OJPHGPU_E_INVALID;
  if (container_bits != 32 && container_bits != 16 && container_bits != 8) return OJPHGPU_E_INVALID;
  return launch<true>(stream, (int)params->reversible, d_descs, n, max_w, max_h, d_base, const_cast<void*>(d_image),
                      Conv{ (int)params->bit_depth, (int)params->is_signed }, container_bits, colour ? 3 : 1);
}

extern "C" int ojphgpu_dwt_inverse_image_ex(void* stream, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                             uint32_t n, uint32_t max_w, uint32_t max_h, void* d_image, void* d_base,
                                             int container_bits, int colour)
{
  if (!params || !d_image || params->bit_depth == 0 || params->bit_depth > (uint32_t)(container_bits == 32 ? 31 : container_bits)) return OJPHGPU_E_INVALID;
  if (container_bits != 32 && container_bits != 16 && container_bits != 8) return OJPHGPU_E_INVALID;
  return launch<false>(stream, (int)params->reversible, d_descs, n, max_w, max_h, d_base, d_image,
                       Conv{ (int)params->bit_depth, (int)params->is_signed }, container_bits, colour ? 3 : 1);
}
