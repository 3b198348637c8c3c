// access-pattern microbenchmark: which walk over a 7680x4320x3 fp32 plane set reaches copy bandwidth?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)
constexpr int W = 7680, H = 4320, C = 3;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void lin_copy(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

// column strip walk: VW floats per lane, wave covers 64*VW columns, walks R rows, prefetch distance PF
template <int VW, int PF, bool SPLIT>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ s, float* __restrict__ d, int R)
{
  typedef float V __attribute__((ext_vector_type(VW)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x * 4 + wave;
  const int x = (strip * 64 + lane) * VW;
  if (x >= W) return;
  const int y0 = blockIdx.y * R, y1 = min(y0 + R, H);
  const size_t plane = (size_t)blockIdx.z * W * H;
  const float* sp = s + plane + x;
  float* dp = d + plane;
  V buf[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) buf[i] = (y0 + i < y1) ? *(const V*)(sp + (size_t)(y0 + i) * W) : V(0);
  for (int y = y0; y < y1; y += PF) {
    V cur[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) cur[i] = buf[i];
#pragma unroll
    for (int i = 0; i < PF; ++i) buf[i] = (y + PF + i < y1) ? *(const V*)(sp + (size_t)(y + PF + i) * W) : V(0);
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int yy = y + i;
      if (yy >= y1) break;
      if (!SPLIT) *(V*)(dp + (size_t)yy * W + x) = cur[i];
      else {
        // de-interleave like the DWT: even/odd columns -> left/right half, even/odd rows -> top/bottom half
        typedef float Hh __attribute__((ext_vector_type(VW / 2)));
        Hh e, o;
        if constexpr (VW == 2) { e = cur[i].x; o = cur[i].y; }
        else { e.x = cur[i].x; e.y = cur[i].z; o.x = cur[i].y; o.y = cur[i].w; }
        const size_t ry = (size_t)((yy >> 1) + (yy & 1) * (H / 2)) * W;
        *(Hh*)(dp + ry + x / 2) = e;
        *(Hh*)(dp + ry + W / 2 + x / 2) = o;
      }
    }
  }
}

// tile: a 256-thread block loads TR rows x 1024 floats (4 KB per row) at once, then stores
template <int TR>
__global__ __launch_bounds__(256) void tile(const float* __restrict__ s, float* __restrict__ d)
{
  const int x = blockIdx.x * 1024 + threadIdx.x * 4;
  if (x >= W) return;
  const int y0 = blockIdx.y * TR;
  const size_t plane = (size_t)blockIdx.z * W * H;
  f4 v[TR];
#pragma unroll
  for (int i = 0; i < TR; ++i) v[i] = (y0 + i < H) ? *(const f4*)(s + plane + (size_t)(y0 + i) * W + x) : f4(0);
#pragma unroll
  for (int i = 0; i < TR; ++i) if (y0 + i < H) *(f4*)(d + plane + (size_t)(y0 + i) * W + x) = v[i];
}

template <typename F> float timeit(F f, int reps = 10)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main()
{
  const size_t n = (size_t)W * H * C;
  float *s, *d; CK(hipMalloc(&s, n * 4)); CK(hipMalloc(&d, n * 4)); CK(hipMemset(s, 1, n * 4));
  const double gb = 2.0 * n * 4 / 1e9;
  auto rep = [&](const char* name, float ms) { printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, gb / ms * 1e3); };
  rep("linear float4 copy (4096 blocks)", timeit([&] { hipLaunchKernelGGL(lin_copy, dim3(4096), dim3(256), 0, 0, (const f4*)s, (f4*)d, n / 4); }));
  rep("linear float4 copy (16384 blocks)", timeit([&] { hipLaunchKernelGGL(lin_copy, dim3(16384), dim3(256), 0, 0, (const f4*)s, (f4*)d, n / 4); }));
  for (int R : {32, 64, 128, 256}) {
    char nm[96];
    auto g2 = dim3((W / 2 / 64 + 3) / 4, (H + R - 1) / R, C), g4 = dim3((W / 4 / 64 + 3) / 4, (H + R - 1) / R, C);
    snprintf(nm, 96, "walk 8B/lane pf1 R=%d", R); rep(nm, timeit([&] { hipLaunchKernelGGL((walk<2, 1, false>), g2, dim3(256), 0, 0, s, d, R); }));
    snprintf(nm, 96, "walk 8B/lane pf2 R=%d", R); rep(nm, timeit([&] { hipLaunchKernelGGL((walk<2, 2, false>), g2, dim3(256), 0, 0, s, d, R); }));
    snprintf(nm, 96, "walk 8B/lane pf4 R=%d", R); rep(nm, timeit([&] { hipLaunchKernelGGL((walk<2, 4, false>), g2, dim3(256), 0, 0, s, d, R); }));
    snprintf(nm, 96, "walk 8B/lane pf2 split R=%d", R); rep(nm, timeit([&] { hipLaunchKernelGGL((walk<2, 2, true>), g2, dim3(256), 0, 0, s, d, R); }));
    snprintf(nm, 96, "walk 16B/lane pf1 R=%d", R); rep(nm, timeit([&] { hipLaunchKernelGGL((walk<4, 1, false>), g4, dim3(256), 0, 0, s, d, R); }));
    snprintf(nm, 96, "walk 16B/lane pf2 R=%d", R); rep(nm, timeit([&] { hipLaunchKernelGGL((walk<4, 2, false>), g4, dim3(256), 0, 0, s, d, R); }));
    snprintf(nm, 96, "walk 16B/lane pf4 R=%d", R); rep(nm, timeit([&] { hipLaunchKernelGGL((walk<4, 4, false>), g4, dim3(256), 0, 0, s, d, R); }));
    snprintf(nm, 96, "walk 16B/lane pf2 split R=%d", R); rep(nm, timeit([&] { hipLaunchKernelGGL((walk<4, 2, true>), g4, dim3(256), 0, 0, s, d, R); }));
  }
  rep("tile 8 rows x 4KB", timeit([&] { hipLaunchKernelGGL((tile<8>), dim3((W + 1023) / 1024, (H + 7) / 8, C), dim3(256), 0, 0, s, d); }));
  rep("tile 16 rows x 4KB", timeit([&] { hipLaunchKernelGGL((tile<16>), dim3((W + 1023) / 1024, (H + 15) / 16, C), dim3(256), 0, 0, s, d); }));
  return 0;
}
