// openjph_amd/csrc/kernels_pixels.hip -- pixel-interleaved samples (the order of .ppm files and of capture / display
// buffers: R G B R G B ..., 8 bits or 16 bits little / big endian) <-> the planar sample containers the codec works
// on ([C][H][W], 8 / 16 / 32 bits per sample).
//
// In the reference this is the job of the image readers / writers, sample by sample on the host: ppm_in::read
// (src/apps/others/ojph_img_io.cpp:338-375: byte swap of 16-bit samples, one component picked out of the
// interleaved line per call) and ppm_out::write with its converters (:539-556, :99-226: clamp, byte swap, interleave).  Here the file's bytes go
// over PCIe as they are and one launch turns them into planes (and back), so that the host never touches a sample.
// Pure data movement, HBM-bound: a thread takes FOUR consecutive pixels -- 4 x C consecutive samples of the
// interleaved side (contiguous bytes), four consecutive samples of every plane (one 4- / 8- / 16-byte store).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ojphgpu.h"

namespace {

template <typename S> __device__ __forceinline__ uint32_t load_sample(const S* p, size_t i, bool swap)
{
  uint32_t v = p[i];
  if (sizeof(S) == 2 && swap) v = ((v & 0xFFu) << 8) | (v >> 8);
  return v;
}
template <typename S> __device__ __forceinline__ void store_sample(S* p, size_t i, uint32_t v, bool swap)
{
  if (sizeof(S) == 2 && swap) v = ((v & 0xFFu) << 8) | ((v >> 8) & 0xFFu);
  p[i] = (S)v;
}

// S: sample type of the interleaved side (uint8_t / uint16_t); D: container of the planar side
template <typename S, typename D, int NC>
__global__ __launch_bounds__(256) void unpack_kernel(const S* __restrict__ src, D* __restrict__ dst, uint64_t npix, uint32_t nc_rt, bool swap)
{
  const uint32_t nc = NC ? (uint32_t)NC : nc_rt;
  const uint64_t p0 = 4ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
  if (p0 >= npix) return;
  const uint32_t n = (uint32_t)(npix - p0 < 4 ? npix - p0 : 4);
  if (NC && n == 4) {                                 // the four pixels as one block of 4 x NC samples, every plane's four as one store
    S v[4 * (NC ? NC : 1)];
    __builtin_memcpy(v, src + (size_t)p0 * NC, sizeof(v));
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)NC; ++c) {
      D o[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        uint32_t x = v[k * NC + c];
        if (sizeof(S) == 2 && swap) x = ((x & 0xFFu) << 8) | (x >> 8);
        o[k] = (D)x;
      }
      __builtin_memcpy(dst + (size_t)c * npix + p0, o, sizeof(o));
    }
    return;
  }
  for (uint32_t c = 0; c < nc; ++c) {
    D* plane = dst + (size_t)c * npix + p0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
      if (k < n) plane[k] = (D)load_sample(src, (size_t)(p0 + k) * nc + c, swap);
  }
}

// planar -> interleaved; values are clamped to [0, 2^bits - 1] the way the reference's writers do (ojph_img_io.cpp:99-226)
template <typename S, typename D, int NC>
__global__ __launch_bounds__(256) void pack_kernel(const D* __restrict__ src, S* __restrict__ dst, uint64_t npix, uint32_t nc_rt, bool swap, uint32_t maxv)
{
  const uint32_t nc = NC ? (uint32_t)NC : nc_rt;
  const uint64_t p0 = 4ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
  if (p0 >= npix) return;
  const uint32_t n = (uint32_t)(npix - p0 < 4 ? npix - p0 : 4);
  if (NC && n == 4) {
    S v[4 * (NC ? NC : 1)];
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)NC; ++c) {
      D in[4];
      __builtin_memcpy(in, src + (size_t)c * npix + p0, sizeof(in));
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        int64_t x = (int64_t)in[k];
        x = x < 0 ? 0 : (x > (int64_t)maxv ? (int64_t)maxv : x);
        uint32_t y = (uint32_t)x;
        if (sizeof(S) == 2 && swap) y = ((y & 0xFFu) << 8) | ((y >> 8) & 0xFFu);
        v[k * NC + c] = (S)y;
      }
    }
    __builtin_memcpy(dst + (size_t)p0 * NC, v, sizeof(v));
    return;
  }
  for (uint32_t c = 0; c < nc; ++c) {
    const D* plane = src + (size_t)c * npix + p0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
      if (k < n) {
        int64_t v = (int64_t)plane[k];
        v = v < 0 ? 0 : (v > (int64_t)maxv ? (int64_t)maxv : v);
        store_sample(dst, (size_t)(p0 + k) * nc + c, (uint32_t)v, swap);
      }
  }
}

template <typename S, typename D>
int launch_unpack(hipStream_t st, const void* src, void* dst, uint64_t npix, uint32_t nc, bool swap)
{
  const dim3 grid((unsigned)((npix + 1023) / 1024)), wg(256);
  if (nc == 1) hipLaunchKernelGGL((unpack_kernel<S, D, 1>), grid, wg, 0, st, (const S*)src, (D*)dst, npix, nc, swap);
  else if (nc == 3) hipLaunchKernelGGL((unpack_kernel<S, D, 3>), grid, wg, 0, st, (const S*)src, (D*)dst, npix, nc, swap);
  else if (nc == 4) hipLaunchKernelGGL((unpack_kernel<S, D, 4>), grid, wg, 0, st, (const S*)src, (D*)dst, npix, nc, swap);
  else hipLaunchKernelGGL((unpack_kernel<S, D, 0>), grid, wg, 0, st, (const S*)src, (D*)dst, npix, nc, swap);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

template <typename S, typename D>
int launch_pack(hipStream_t st, const void* src, void* dst, uint64_t npix, uint32_t nc, bool swap, uint32_t maxv)
{
  const dim3 grid((unsigned)((npix + 1023) / 1024)), wg(256);
  if (nc == 1) hipLaunchKernelGGL((pack_kernel<S, D, 1>), grid, wg, 0, st, (const D*)src, (S*)dst, npix, nc, swap, maxv);
  else if (nc == 3) hipLaunchKernelGGL((pack_kernel<S, D, 3>), grid, wg, 0, st, (const D*)src, (S*)dst, npix, nc, swap, maxv);
  else if (nc == 4) hipLaunchKernelGGL((pack_kernel<S, D, 4>), grid, wg, 0, st, (const D*)src, (S*)dst, npix, nc, swap, maxv);
  else hipLaunchKernelGGL((pack_kernel<S, D, 0>), grid, wg, 0, st, (const D*)src, (S*)dst, npix, nc, swap, maxv);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

// ---- bit-packed samples (10 / 12 / 14 bits each, consecutive in a little-endian bit string: sample i occupies bits
// [i * BITS, (i + 1) * BITS)) <-> 16- or 32-bit containers.  A thread takes 32 samples = BITS dwords.
template <int BITS, typename D>
__global__ __launch_bounds__(256) void unpack_bits_kernel(const uint32_t* __restrict__ src, D* __restrict__ dst, uint64_t n)
{
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t s0 = 32ull * g;
  if (s0 >= n) return;
  uint32_t w[BITS + 1];
#pragma unroll
  for (int i = 0; i < BITS; ++i) w[i] = src[g * BITS + i];
  w[BITS] = 0;
  D o[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int pos = j * BITS, wi = pos >> 5, sh = pos & 31;
    const uint32_t v = sh + BITS <= 32 ? w[wi] >> sh : __funnelshift_r(w[wi], w[wi + 1], sh);
    o[j] = (D)(v & ((1u << BITS) - 1u));
  }
  if (s0 + 32 <= n) __builtin_memcpy(dst + s0, o, sizeof(o));
  else for (int j = 0; j < 32; ++j) if (s0 + j < n) dst[s0 + j] = o[j];
}

template <int BITS, typename D>
__global__ __launch_bounds__(256) void pack_bits_kernel(const D* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n)
{
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t s0 = 32ull * g;
  if (s0 >= n) return;
  D in[32];
  if (s0 + 32 <= n) __builtin_memcpy(in, src + s0, sizeof(in));
  else for (int j = 0; j < 32; ++j) in[j] = s0 + j < n ? src[s0 + j] : (D)0;
  uint32_t w[BITS + 1];
#pragma unroll
  for (int i = 0; i <= BITS; ++i) w[i] = 0;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    int64_t x = (int64_t)in[j];
    x = x < 0 ? 0 : (x > (int64_t)((1u << BITS) - 1u) ? (int64_t)((1u << BITS) - 1u) : x);     // clamp, as the reference's writers do
    const uint32_t v = (uint32_t)x;
    const int pos = j * BITS, wi = pos >> 5, sh = pos & 31;
    w[wi] |= v << sh;
    if (sh + BITS > 32) w[wi + 1] |= v >> (32 - sh);
  }
#pragma unroll
  for (int i = 0; i < BITS; ++i) dst[g * BITS + i] = w[i];     // (the buffer is padded to whole groups of 32 samples)
}

template <typename D>
int launch_bits(hipStream_t st, bool unpack, const void* src, void* dst, uint64_t n, int bits)
{
  const dim3 grid((unsigned)((n + 32ull * 256 - 1) / (32ull * 256))), wg(256);
#define BITS_CASE(B) case B: if (unpack) hipLaunchKernelGGL((unpack_bits_kernel<B, D>), grid, wg, 0, st, (const uint32_t*)src, (D*)dst, n); \
                             else hipLaunchKernelGGL((pack_bits_kernel<B, D>), grid, wg, 0, st, (const D*)src, (uint32_t*)dst, n); break;
  switch (bits) { BITS_CASE(10) BITS_CASE(12) BITS_CASE(14) default: return OJPHGPU_E_INVALID; }
#undef BITS_CASE
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

bool bad_args(const void* a, const void* b, uint32_t w, uint32_t h, uint32_t nc, int pixel_bits, int container_bits)
{
  return !a || !b || w == 0 || h == 0 || nc == 0 || nc > 16384 || (pixel_bits != 8 && pixel_bits != 16) ||
         (container_bits != 8 && container_bits != 16 && container_bits != 32) || (pixel_bits == 16 && container_bits == 8);
}

}  // namespace

extern "C" int ojphgpu_unpack_pixels(void* stream, const void* d_pixels, void* d_planes, uint32_t width, uint32_t height,
                                      uint32_t num_comps, int pixel_bits, int big_endian, int container_bits)
{
  if (bad_args(d_pixels, d_planes, width, height, num_comps, pixel_bits, container_bits)) return OJPHGPU_E_INVALID;
  const uint64_t npix = (uint64_t)width * height;
  hipStream_t st = (hipStream_t)stream;
  const bool swap = big_endian != 0;
  if (pixel_bits == 8) {
    if (container_bits == 8) return launch_unpack<uint8_t, uint8_t>(st, d_pixels, d_planes, npix, num_comps, false);
    if (container_bits == 16) return launch_unpack<uint8_t, uint16_t>(st, d_pixels, d_planes, npix, num_comps, false);
    return launch_unpack<uint8_t, int32_t>(st, d_pixels, d_planes, npix, num_comps, false);
  }
  if (container_bits == 16) return launch_unpack<uint16_t, uint16_t>(st, d_pixels, d_planes, npix, num_comps, swap);
  return launch_unpack<uint16_t, int32_t>(st, d_pixels, d_planes, npix, num_comps, swap);
}

extern "C" int ojphgpu_pack_pixels(void* stream, const void* d_planes, void* d_pixels, uint32_t width, uint32_t height,
                                    uint32_t num_comps, int container_bits, int pixel_bits, int big_endian, uint32_t bit_depth)
{
  if (bad_args(d_planes, d_pixels, width, height, num_comps, pixel_bits, container_bits) || bit_depth == 0 || bit_depth > (uint32_t)pixel_bits)
    return OJPHGPU_E_INVALID;
  const uint64_t npix = (uint64_t)width * height;
  hipStream_t st = (hipStream_t)stream;
  const bool swap = big_endian != 0;
  const uint32_t maxv = (1u << bit_depth) - 1u;
  if (pixel_bits == 8) {
    if (container_bits == 8) return launch_pack<uint8_t, uint8_t>(st, d_planes, d_pixels, npix, num_comps, false, maxv);
    if (container_bits == 16) return launch_pack<uint8_t, uint16_t>(st, d_planes, d_pixels, npix, num_comps, false, maxv);
    return launch_pack<uint8_t, int32_t>(st, d_planes, d_pixels, npix, num_comps, false, maxv);
  }
  if (container_bits == 16) return launch_pack<uint16_t, uint16_t>(st, d_planes, d_pixels, npix, num_comps, swap, maxv);
  return launch_pack<uint16_t, int32_t>(st, d_planes, d_pixels, npix, num_comps, swap, maxv);
}

// d_packed: num_samples samples of `bits` (10, 12, 14) bits each, one little-endian bit string, padded to a multiple of
// 32 samples (4 * bits bytes); d_samples: 16- or 32-bit containers
extern "C" int ojphgpu_unpack_bits(void* stream, const void* d_packed, void* d_samples, uint64_t num_samples, int bits, int container_bits)
{
  if (!d_packed || !d_samples || num_samples == 0 || ((uintptr_t)d_packed & 3u)) return OJPHGPU_E_INVALID;
  if (container_bits == 16) return launch_bits<uint16_t>((hipStream_t)stream, true, d_packed, d_samples, num_samples, bits);
  if (container_bits == 32) return launch_bits<int32_t>((hipStream_t)stream, true, d_packed, d_samples, num_samples, bits);
  return OJPHGPU_E_INVALID;
}

extern "C" int ojphgpu_pack_bits(void* stream, const void* d_samples, void* d_packed, uint64_t num_samples, int container_bits, int bits)
{
  if (!d_packed || !d_samples || num_samples == 0 || ((uintptr_t)d_packed & 3u)) return OJPHGPU_E_INVALID;
  if (container_bits == 16) return launch_bits<uint16_t>((hipStream_t)stream, false, d_samples, d_packed, num_samples, bits);
  if (container_bits == 32) return launch_bits<int32_t>((hipStream_t)stream, false, d_samples, d_packed, num_samples, bits);
  return OJPHGPU_E_INVALID;
}
