// What the host link of this box carries in the two directions at once, by who does the copying: the SDMA engines
// (hipMemcpyAsync) or a copy kernel on mapped pinned memory, in every pairing.  The frame pipes (ojphgpu_pipe.cpp) upload with
// SDMA and download with a kernel; this probe says whether another pairing would carry more.
//   hipcc -O2 --offload-arch=gfx950 -o link_duplex link_duplex.hip && ./link_duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum How { NONE, SDMA, KERNEL };
static const char* name(How h) { return h == NONE ? "-" : h == SDMA ? "sdma" : "kernel"; }

int main(int argc, char** argv)
{
  const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 256) << 20;
  const int reps = 10;
  void *h_up, *h_down, *d_up, *d_down;
  CK(hipHostMalloc(&h_up, bytes, hipHostMallocMapped));
  CK(hipHostMalloc(&h_down, bytes, hipHostMallocMapped));
  CK(hipMalloc(&d_up, bytes));
  CK(hipMalloc(&d_down, bytes));
  memset(h_up, 1, bytes); memset(h_down, 2, bytes);
  CK(hipMemset(d_down, 3, bytes));
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s_up, s_down;
  CK(hipStreamCreateWithPriority(&s_up, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s_down, hipStreamNonBlocking, hi));
  void *hd_up, *hd_down;                         // the device's view of the pinned buffers
  CK(hipHostGetDevicePointer(&hd_up, h_up, 0));
  CK(hipHostGetDevicePointer(&hd_down, h_down, 0));

  auto go = [&](How up, How down, int wgs) {
    double best = 1e9;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      if (up == SDMA) CK(hipMemcpyAsync(d_up, h_up, bytes, hipMemcpyHostToDevice, s_up));
      if (up == KERNEL) copy_kernel<<<wgs, 256, 0, s_up>>>((const uint4*)hd_up, (uint4*)d_up, bytes / 16);
      if (down == SDMA) CK(hipMemcpyAsync(h_down, d_down, bytes, hipMemcpyDeviceToHost, s_down));
      if (down == KERNEL) copy_kernel<<<wgs, 256, 0, s_down>>>((const uint4*)d_down, (uint4*)hd_down, bytes / 16);
      CK(hipDeviceSynchronize());
      const double dt = now() - t0;
      if (r >= 2 && dt < best) best = dt;
    }
    const double gb = bytes / 1e9, total = ((up != NONE) + (down != NONE)) * gb;
    printf("up %-6s down %-6s wgs %4d : %7.3f ms  %6.1f GB/s per direction, %6.1f GB/s together\n", name(up), name(down), wgs,
           best * 1e3, gb / best, total / best);
  };
  printf("%zu MiB each way, best of %d\n", bytes >> 20, reps);
  go(SDMA, NONE, 0); go(NONE, SDMA, 0); go(SDMA, SDMA, 0);
  for (int wgs : { 8, 32, 128, 512 }) {
    go(KERNEL, NONE, wgs); go(NONE, KERNEL, wgs);
    go(SDMA, KERNEL, wgs); go(KERNEL, SDMA, wgs); go(KERNEL, KERNEL, wgs);
  }
  return 0;
}
