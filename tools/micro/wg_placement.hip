#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(64) void k(uint32_t* out, int spin)
{
  __shared__ uint32_t lds[1312];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  uint32_t acc = lds[(threadIdx.x * 7) & 63];
  long long t0 = clock64();
  while (clock64() - t0 < spin) acc = acc * 1664525u + 1013904223u;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = (xcc & 0xF) | (acc & 0x80000000u ? 0 : 0); }
}
int main(int argc, char** argv)
{
  int n = argc > 1 ? atoi(argv[1]) : 386;
  uint32_t* d; hipMalloc(&d, n * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d, 20000);
    hipDeviceSynchronize();
  }
  std::vector<uint32_t> h(n * 2); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<uint32_t, int> per_simd, per_cu;
  for (int i = 0; i < n; ++i) {
    uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
    uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    uint32_t cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[cuid]++; per_simd[(cuid << 2) | simd]++;
    if (i < 24) printf("wg %d: xcc %u se %u sh %u cu %u simd %u wave %u\n", i, xcc, se, sh, cu, simd, hw & 0xF);
  }
  std::map<int, int> hs, hc;
  for (auto& kv : per_simd) hs[kv.second]++;
  for (auto& kv : per_cu) hc[kv.second]++;
  printf("n=%d distinct CUs %zu distinct SIMDs %zu\n", n, per_cu.size(), per_simd.size());
  for (auto& kv : hc) printf("  CUs with %d waves: %d\n", kv.first, kv.second);
  for (auto& kv : hs) printf("  SIMDs with %d waves: %d\n", kv.first, kv.second);
  return 0;
}
