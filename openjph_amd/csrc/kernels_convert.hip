// openjph_amd/csrc/kernels_convert.hip -- sample conversion (level shift / int<->float) and
// colour transforms (RCT / ICT) between image-sized int32 component planes and the
// tile-component planes of the coefficient arena.
//
// Reference (per image line, ojph_tile.cpp:332-518 calling ojph_colour.cpp):
//   gen_rev_convert            ojph_colour.cpp:238-275   v + shift
//   gen_irv_convert_to_float   ojph_colour.cpp:388-436   (v - half) * 2^-B
//   gen_irv_convert_to_integer ojph_colour.cpp:316-386   round(t * 2^B) clamped, + half
//   gen_rct_forward/backward   ojph_colour.cpp:443-543
//   gen_ict_forward/backward   ojph_colour.cpp:545-567   constants :221-231
// Pure element-wise, HBM-bound: 16-byte vector accesses when rows are 16-byte aligned.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ojphgpu.h"

namespace {

struct ConvParams {
  uint32_t img_w, img_h, num_comps, bit_depth, is_signed, reversible, color;
  uint32_t nlt3;       // type 3 non-linearity of a signed component (gen_rev_convert_nlt_type3, ojph_colour.cpp:273-311)
  uint32_t wide;       // the component is on the 64-bit sample path: its plane holds int64 samples (two arena elements each)
};

constexpr float ALPHA_RF = 0.299f, ALPHA_GF = 0.587f, ALPHA_BF = 0.114f;

__device__ __forceinline__ float to_float(int v, const ConvParams& p)
{
  const float mul = (float)(1.0 / (double)(1ULL << p.bit_depth));
  const int half = p.is_signed ? 0 : (int)(1u << (p.bit_depth - 1));
  return __fmul_rn((float)(v - half), mul);
}

__device__ __forceinline__ int to_int(float f, const ConvParams& p)
{
  const int neg_limit = (int)0x80000000 >> (32 - p.bit_depth);
  const float mul = (float)(1ull << p.bit_depth);
  const float up = -(float)neg_limit, low = (float)neg_limit;
  const int s_up = 0x7FFFFFFF >> (32 - p.bit_depth), s_low = (int)0x80000000 >> (32 - p.bit_depth);
  const int half = p.is_signed ? 0 : (int)(1u << (p.bit_depth - 1));
  float t = __fmul_rn(f, mul);
  int v = (int)__fadd_rn(t, t >= 0.0f ? 0.5f : -0.5f);    // ojph_round: truncation of t +- 0.5
  v = t >= low ? v : s_low;
  v = t < up ? v : s_up;
  return v + half;
}

// the component's own bit depth / signedness / kind of conversion when its descriptor carries one
__device__ __forceinline__ ConvParams with_fmt(ConvParams p, const ojphgpu_convert_desc& d)
{
  if (d.fmt) { p.bit_depth = d.fmt & 0xFFu; p.is_signed = (d.fmt >> 8) & 1u; }
  if (d.fmt & 0x200u) p.reversible = (d.fmt >> 10) & 1u;     // the component's own wavelet (COC)
  p.nlt3 = (d.fmt >> 11) & 1u;
  p.wide = (d.fmt >> 12) & 1u;
  return p;
}

// 64-bit sample path (reversible only).  gen_rev_convert / gen_rev_convert_nlt_type3 with a 32-bit source and a 64-bit
// destination line, and back (ojph_colour.cpp:250-268, :288-311): the level shift in 64-bit arithmetic, the way back
// truncated to 32 bits.  Through the colour transform the reference's lines stay 32 bits wide on the image side
// (ojph_tile.cpp:312-322: the level shift wraps like any si32 sum) and gen_rct_forward / _backward widen / narrow
// themselves (:467-489, :517-541).
__device__ __forceinline__ long long* plane64(uint32_t* arena, const ojphgpu_convert_desc& d) { return reinterpret_cast<long long*>(arena + d.plane_off); }
__device__ __forceinline__ const long long* plane64(const uint32_t* arena, const ojphgpu_convert_desc& d) { return reinterpret_cast<const long long*>(arena + d.plane_off); }
__device__ __forceinline__ long long half64(const ConvParams& p) { return 1ll << (p.bit_depth - 1); }
__device__ __forceinline__ int level_shift32(const ConvParams& p) { return p.is_signed ? 0 : (int)(unsigned)(1ull << (p.bit_depth - 1)); }   // (wraps above 32 bits, like the (si32) cast)
__device__ __forceinline__ int nlt3_map32(int v, const ConvParams& p)
{
  return (p.nlt3 && v < 0) ? (int)(0u - (unsigned)v - (unsigned)(half64(p) + 1)) : v;
}

// type 3 non-linearity: negative values v <-> -v - (2^(B-1) + 1); its own inverse, applied to the integer
// sample on the way in and on the way out (ojph_tile.cpp:352-354, :446-448; ojph_colour.cpp:344-352, :406-412)
__device__ __forceinline__ int nlt3_map(int v, const ConvParams& p)
{
  return (p.nlt3 && v < 0) ? -v - (int)((1u << (p.bit_depth - 1)) + 1u) : v;
}

// S: container of the image samples -- int (32-bit), short (16-bit) or signed char (8-bit): two's complement for
// signed components, the full unsigned range of the container otherwise
template <typename S> __device__ __forceinline__ int sample_in(S v, const ConvParams&) { return (int)v; }
template <> __device__ __forceinline__ int sample_in<short>(short v, const ConvParams& p) { return p.is_signed ? (int)v : (int)(unsigned short)v; }
template <> __device__ __forceinline__ int sample_in<signed char>(signed char v, const ConvParams& p) { return p.is_signed ? (int)v : (int)(unsigned char)v; }

template <typename S>
__global__ __launch_bounds__(256) void convert_forward_kernel(ConvParams p, const ojphgpu_convert_desc* __restrict__ descs,
                                                              const S* __restrict__ image, uint32_t* __restrict__ arena)
{
  const uint32_t tile = blockIdx.z;
  const uint32_t nc = p.num_comps;
  const ojphgpu_convert_desc d0 = descs[tile * nc];
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  auto at = [&](const ojphgpu_convert_desc& d) { return d.img_off + (size_t)(d.src_y0 + y) * d.img_pitch + d.src_x0 + x; };
  const ConvParams pc = p;                        // launch-wide parameters; p becomes the component's
  uint32_t c_first = 0;                           // components past the colour-transformed triple go the plain way
  if (pc.color && (c_first = 3, x < d0.w && y < d0.h)) {   // the first three components share geometry and sample format
    // (the three components share their geometry; each has its own sample format -- the reference converts component by
    // component, ojph_tile.cpp:332-437, and a codestream whose SIZ gives them different depths is read that way)
    const ojphgpu_convert_desc d1 = descs[tile * nc + 1], d2 = descs[tile * nc + 2];
    p = with_fmt(pc, d0);
    const ConvParams p1 = with_fmt(pc, d1), p2 = with_fmt(pc, d2);
    int r = nlt3_map(sample_in<S>(image[at(d0)], p), p), g = nlt3_map(sample_in<S>(image[at(d1)], p1), p1), b = nlt3_map(sample_in<S>(image[at(d2)], p2), p2);
    if (p.reversible && p.wide) {
      const long long rr = (int)((unsigned)r - (unsigned)level_shift32(p)), gg = (int)((unsigned)g - (unsigned)level_shift32(p1)),
                      bb = (int)((unsigned)b - (unsigned)level_shift32(p2));
      plane64(arena, d0)[(size_t)y * d0.pitch + x] = (rr + (gg << 1) + bb) >> 2;
      plane64(arena, d1)[(size_t)y * d1.pitch + x] = bb - gg;
      plane64(arena, d2)[(size_t)y * d2.pitch + x] = rr - gg;
    } else if (p.reversible) {
      r -= level_shift32(p); g -= level_shift32(p1); b -= level_shift32(p2);
      int yy = (r + (g << 1) + b) >> 2, cb = b - g, cr = r - g;
      arena[d0.plane_off + (size_t)y * d0.pitch + x] = (uint32_t)yy;
      arena[d1.plane_off + (size_t)y * d1.pitch + x] = (uint32_t)cb;
      arena[d2.plane_off + (size_t)y * d2.pitch + x] = (uint32_t)cr;
    } else {
      const float beta_cb = (float)(0.5 / (1 - (double)ALPHA_BF)), beta_cr = (float)(0.5 / (1 - (double)ALPHA_RF));
      float rf = to_float(r, p), gf = to_float(g, p1), bf = to_float(b, p2);
      float yy = __fadd_rn(__fadd_rn(__fmul_rn(ALPHA_RF, rf), __fmul_rn(ALPHA_GF, gf)), __fmul_rn(ALPHA_BF, bf));
      float cb = __fmul_rn(beta_cb, __fsub_rn(bf, yy)), cr = __fmul_rn(beta_cr, __fsub_rn(rf, yy));
      arena[d0.plane_off + (size_t)y * d0.pitch + x] = __float_as_uint(yy);
      arena[d1.plane_off + (size_t)y * d1.pitch + x] = __float_as_uint(cb);
      arena[d2.plane_off + (size_t)y * d2.pitch + x] = __float_as_uint(cr);
    }
  }
  for (uint32_t c = c_first; c < nc; ++c) {
    const ojphgpu_convert_desc d = descs[tile * nc + c];
    if (x >= d.w || y >= d.h) continue;           // sub-sampled components are smaller
    p = with_fmt(pc, d);
    if (p.reversible && p.wide) {
      const long long v = sample_in<S>(image[at(d)], p);
      plane64(arena, d)[(size_t)y * d.pitch + x] = (p.nlt3 && p.is_signed) ? (v >= 0 ? v : -v - (half64(p) + 1)) : v - (p.is_signed ? 0 : half64(p));
      continue;
    }
    int v = nlt3_map(sample_in<S>(image[at(d)], p), p);
    uint32_t o;
    if (p.reversible) o = (uint32_t)v - (uint32_t)level_shift32(p);
    else o = __float_as_uint(to_float(v, p));
    arena[d.plane_off + (size_t)y * d.pitch + x] = o;
  }
}

// a narrower container saturates at its own range (see fit_container in kernels_dwt.hip)
template <typename S> __device__ __forceinline__ S sample_out(int v, const ConvParams&) { return (S)v; }
template <> __device__ __forceinline__ short sample_out<short>(int v, const ConvParams& p)
{ return (short)(p.is_signed ? min(max(v, -32768), 32767) : min(max(v, 0), 65535)); }
template <> __device__ __forceinline__ signed char sample_out<signed char>(int v, const ConvParams& p)
{ return (signed char)(p.is_signed ? min(max(v, -128), 127) : min(max(v, 0), 255)); }

template <typename S>
__global__ __launch_bounds__(256) void convert_inverse_kernel(ConvParams p, const ojphgpu_convert_desc* __restrict__ descs,
                                                              S* __restrict__ image, const uint32_t* __restrict__ arena)
{
  const uint32_t tile = blockIdx.z;
  const uint32_t nc = p.num_comps;
  const ojphgpu_convert_desc d0 = descs[tile * nc];
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  auto at = [&](const ojphgpu_convert_desc& d) { return d.img_off + (size_t)(d.src_y0 + y) * d.img_pitch + d.src_x0 + x; };
  const ConvParams pc = p;                        // launch-wide parameters; p becomes the component's
  uint32_t c_first = 0;
  if (pc.color && (c_first = 3, x < d0.w && y < d0.h)) {   // the first three components share geometry and sample format
    const ojphgpu_convert_desc d1 = descs[tile * nc + 1], d2 = descs[tile * nc + 2];
    p = with_fmt(pc, d0);                                    // (each component leaves in its own sample format: ojph_tile.cpp:439-518)
    const ConvParams p1 = with_fmt(pc, d1), p2 = with_fmt(pc, d2);
    if (p.reversible && p.wide) {
      const long long yy = plane64(arena, d0)[(size_t)y * d0.pitch + x], cb = plane64(arena, d1)[(size_t)y * d1.pitch + x],
                      cr = plane64(arena, d2)[(size_t)y * d2.pitch + x];
      const long long g = yy - ((cb + cr) >> 2);
      image[at(d0)] = sample_out<S>(nlt3_map32((int)((unsigned)(int)(cr + g) + (unsigned)level_shift32(p)), p), p);
      image[at(d1)] = sample_out<S>(nlt3_map32((int)((unsigned)(int)g + (unsigned)level_shift32(p1)), p1), p1);
      image[at(d2)] = sample_out<S>(nlt3_map32((int)((unsigned)(int)(cb + g) + (unsigned)level_shift32(p2)), p2), p2);
    }
    uint32_t a = arena[d0.plane_off + (size_t)y * d0.pitch + x];
    uint32_t b = arena[d1.plane_off + (size_t)y * d1.pitch + x];
    uint32_t c = arena[d2.plane_off + (size_t)y * d2.pitch + x];
    if (p.reversible && p.wide) {
    } else if (p.reversible) {
      int yy = (int)a, cb = (int)b, cr = (int)c;
      int g = yy - ((cb + cr) >> 2);
      image[at(d0)] = sample_out<S>(nlt3_map(cr + g + level_shift32(p), p), p); image[at(d1)] = sample_out<S>(nlt3_map(g + level_shift32(p1), p1), p1);
      image[at(d2)] = sample_out<S>(nlt3_map(cb + g + level_shift32(p2), p2), p2);
    } else {
      const float g_cb2g = (float)(2.0 * (double)ALPHA_BF * (1.0 - (double)ALPHA_BF) / (double)ALPHA_GF);
      const float g_cr2g = (float)(2.0 * (double)ALPHA_RF * (1.0 - (double)ALPHA_RF) / (double)ALPHA_GF);
      const float g_cb2b = (float)(2.0 * (1.0 - (double)ALPHA_BF));
      const float g_cr2r = (float)(2.0 * (1.0 - (double)ALPHA_RF));
      float yy = __uint_as_float(a), cb = __uint_as_float(b), cr = __uint_as_float(c);
      float g = __fsub_rn(__fsub_rn(yy, __fmul_rn(g_cr2g, cr)), __fmul_rn(g_cb2g, cb));
      float r = __fadd_rn(yy, __fmul_rn(g_cr2r, cr));
      float bb = __fadd_rn(yy, __fmul_rn(g_cb2b, cb));
      image[at(d0)] = sample_out<S>(nlt3_map(to_int(r, p), p), p); image[at(d1)] = sample_out<S>(nlt3_map(to_int(g, p1), p1), p1); image[at(d2)] = sample_out<S>(nlt3_map(to_int(bb, p2), p2), p2);
    }
  }
  for (uint32_t c = c_first; c < nc; ++c) {
    const ojphgpu_convert_desc d = descs[tile * nc + c];
    if (x >= d.w || y >= d.h) continue;
    p = with_fmt(pc, d);
    if (p.reversible && p.wide) {
      const long long v = plane64(arena, d)[(size_t)y * d.pitch + x];
      image[at(d)] = sample_out<S>((int)((p.nlt3 && p.is_signed) ? (v >= 0 ? v : -v - (half64(p) + 1)) : v + (p.is_signed ? 0 : half64(p))), p);
      continue;
    }
    uint32_t a = arena[d.plane_off + (size_t)y * d.pitch + x];
    int v;
    if (p.reversible) v = (int)(a + (uint32_t)level_shift32(p));
    else v = to_int(__uint_as_float(a), p);
    image[at(d)] = sample_out<S>(nlt3_map(v, p), p);
  }
}

ConvParams make(const ojphgpu_params* q)
{
  ConvParams p;
  p.img_w = q->width; p.img_h = q->height; p.num_comps = q->num_comps; p.bit_depth = q->bit_depth;
  p.is_signed = q->is_signed; p.reversible = q->reversible; p.color = q->color_transform; p.nlt3 = 0; p.wide = 0;
  return p;
}

}  // namespace

extern "C" int ojphgpu_convert_forward(void* stream, const ojphgpu_params* params,
                                        const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                                        uint32_t max_w, uint32_t max_h, const int32_t* d_image, void* d_arena)
{
  if (!params || !d_descs || !d_image || !d_arena) return OJPHGPU_E_INVALID;
  if (n_tiles == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  dim3 grid((max_w + 255) / 256, max_h, n_tiles);
  hipLaunchKernelGGL(convert_forward_kernel<int>, grid, dim3(256), 0, (hipStream_t)stream, make(params), d_descs,
                     d_image, (uint32_t*)d_arena);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

extern "C" int ojphgpu_convert_forward16(void* stream, const ojphgpu_params* params,
                                          const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                                          uint32_t max_w, uint32_t max_h, const uint16_t* d_image, void* d_arena)
{
  if (!params || !d_descs || !d_image || !d_arena || params->bit_depth > 16) return OJPHGPU_E_INVALID;
  if (n_tiles == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  dim3 grid((max_w + 255) / 256, max_h, n_tiles);
  hipLaunchKernelGGL(convert_forward_kernel<short>, grid, dim3(256), 0, (hipStream_t)stream, make(params), d_descs,
                     (const short*)d_image, (uint32_t*)d_arena);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

extern "C" int ojphgpu_convert_inverse(void* stream, const ojphgpu_params* params,
                                        const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                                        uint32_t max_w, uint32_t max_h, int32_t* d_image, const void* d_arena)
{
  if (!params || !d_descs || !d_image || !d_arena) return OJPHGPU_E_INVALID;
  if (n_tiles == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  dim3 grid((max_w + 255) / 256, max_h, n_tiles);
  hipLaunchKernelGGL(convert_inverse_kernel<int>, grid, dim3(256), 0, (hipStream_t)stream, make(params), d_descs,
                     d_image, (const uint32_t*)d_arena);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

extern "C" int ojphgpu_convert_inverse16(void* stream, const ojphgpu_params* params,
                                          const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                                          uint32_t max_w, uint32_t max_h, uint16_t* d_image, const void* d_arena)
{
  if (!params || !d_descs || !d_image || !d_arena || params->bit_depth > 16) return OJPHGPU_E_INVALID;
  if (n_tiles == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  dim3 grid((max_w + 255) / 256, max_h, n_tiles);
  hipLaunchKernelGGL(convert_inverse_kernel<short>, grid, dim3(256), 0, (hipStream_t)stream, make(params), d_descs,
                     (short*)d_image, (const uint32_t*)d_arena);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

// the general form: container_bits = 32 | 16 | 8
extern "C" int ojphgpu_convert_forward_ex(void* stream, const ojphgpu_params* params, const ojphgpu_convert_desc* d_descs,
                                           uint32_t n_tiles, uint32_t max_w, uint32_t max_h, const void* d_image, void* d_arena,
                                           int container_bits)
{
  if (container_bits == 32) return ojphgpu_convert_forward(stream, params, d_descs, n_tiles, max_w, max_h, (const int32_t*)d_image, d_arena);
  if (container_bits == 16) return ojphgpu_convert_forward16(stream, params, d_descs, n_tiles, max_w, max_h, (const uint16_t*)d_image, d_arena);
  if (container_bits != 8 || !params || !d_descs || !d_image || !d_arena || params->bit_depth > 8) return OJPHGPU_E_INVALID;
  if (n_tiles == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  dim3 grid((max_w + 255) / 256, max_h, n_tiles);
  hipLaunchKernelGGL(convert_forward_kernel<signed char>, grid, dim3(256), 0, (hipStream_t)stream, make(params), d_descs,
                     (const signed char*)d_image, (uint32_t*)d_arena);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

extern "C" int ojphgpu_convert_inverse_ex(void* stream, const ojphgpu_params* params, const ojphgpu_convert_desc* d_descs,
                                           uint32_t n_tiles, uint32_t max_w, uint32_t max_h, void* d_image, const void* d_arena,
                                           int container_bits)
{
  if (container_bits == 32) return ojphgpu_convert_inverse(stream, params, d_descs, n_tiles, max_w, max_h, (int32_t*)d_image, d_arena);
  if (container_bits == 16) return ojphgpu_convert_inverse16(stream, params, d_descs, n_tiles, max_w, max_h, (uint16_t*)d_image, d_arena);
  if (container_bits != 8 || !params || !d_descs || !d_image || !d_arena || params->bit_depth > 8) return OJPHGPU_E_INVALID;
  if (n_tiles == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  dim3 grid((max_w + 255) / 256, max_h, n_tiles);
  hipLaunchKernelGGL(convert_inverse_kernel<signed char>, grid, dim3(256), 0, (hipStream_t)stream, make(params), d_descs,
                     (signed char*)d_image, (const uint32_t*)d_arena);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}
