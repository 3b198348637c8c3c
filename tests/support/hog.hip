// tests/support/hog.hip -- test infrastructure, not part of the product: a kernel that occupies every wavefront slot and
// most of the LDS of every compute unit for a given time, so that the GPU tests can run the codec's launches on a chip that
// is held by somebody else (tests/test_gpu_contention.py).  Built by __graft_entry__.build() into tests/support/libhog.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void hog_kernel(uint32_t ticks, uint32_t* sink)
{
  extern __shared__ uint32_t s_fill[];
  s_fill[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  uint32_t acc = s_fill[(threadIdx.x * 7u) & 1023u];
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)ticks) {
#pragma unroll
    for (int i = 0; i < 64; ++i) acc = acc * 1664525u + 1013904223u;     // keeps the VALU pipes busy between two looks at the clock
  }
  if (acc == 0x12345678u && sink) *sink = acc;
}

// holds the chip for `ms` milliseconds from the moment its workgroups are resident: `wgs_per_cu` workgroups of 16
// wavefronts and `lds_kb` KB of LDS per compute unit (2 x 16 wavefronts = every slot of a CU)
extern "C" int ojph_test_hog(void* stream, int ms, int wgs_per_cu, int lds_kb)
{
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (ms < 1 || ms > 2000 || wgs_per_cu < 1 || wgs_per_cu > 8 || lds_kb < 4 || lds_kb > 64) return -2;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr = true; }
  hipLaunchKernelGGL(hog_kernel, dim3(cus * wgs_per_cu), dim3(1024), (size_t)lds_kb * 1024, (hipStream_t)stream,
                     (uint32_t)ms * 100000u, (uint32_t*)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
