// which CUs does a stream created with hipExtStreamCreateWithCUMask use?  (mask bit i -> ?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(64) void k(uint32_t* out, int spin)
{
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = clock64(); uint32_t acc = threadIdx.x;
  while (clock64() - t0 < spin) acc = acc * 1664525u + 1013904223u;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = (xcc & 0xF) | (acc == 12345u ? 16 : 0); }
}
static void run(const char* name, const std::vector<uint32_t>& mask)
{
  hipStream_t s; 
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int n = 2048; uint32_t* d; (void)hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, s, d, 20000); (void)hipStreamSynchronize(s);
  std::vector<uint32_t> h(n * 2); (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<uint32_t, std::set<uint32_t>> per_xcc;
  for (int i = 0; i < n; ++i) { uint32_t hw = h[2 * i], x = h[2 * i + 1] & 0xF; per_xcc[x].insert(((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 0xF)); }
  size_t total = 0; printf("%s:", name);
  for (auto& kv : per_xcc) { printf(" xcc%u=%zu", kv.first, kv.second.size()); total += kv.second.size(); }
  printf("  total CUs %zu\n", total);
  (void)hipFree(d); (void)hipStreamDestroy(s);
}
int main()
{
  run("all 256 bits", std::vector<uint32_t>(8, 0xFFFFFFFFu));
  run("first 32 bits", { 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0 });
  run("every 8th bit", std::vector<uint32_t>(8, 0x01010101u));
  run("all but every 8th", std::vector<uint32_t>(8, 0xFEFEFEFEu));
  run("low half (128 bits)", { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0 });
  return 0;
}
