// Which XCD does workgroup i of a launch run on (HW_REG_XCC_ID), and in which order do workgroups START?
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/xcc_probe tools/micro/xcc_probe.hip && tools/micro/xcc_probe [grid] [threads]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void probe(unsigned* out, unsigned* counter, int spin)
{
  __shared__ unsigned t;
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    t = atomicAdd(counter, 1u);
    out[3 * blockIdx.x] = xcc; out[3 * blockIdx.x + 1] = t; out[3 * blockIdx.x + 2] = hw;
  }
  __syncthreads();
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
int main(int argc, char** argv)
{
  const int grid = argc > 1 ? atoi(argv[1]) : 509, threads = argc > 2 ? atoi(argv[2]) : 768;
  unsigned *d_out, *d_cnt;
  hipMalloc(&d_out, grid * 12); hipMalloc(&d_cnt, 4); hipMemset(d_cnt, 0, 4);
  hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), 0, 0, d_out, d_cnt, 2000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(3 * grid);
  hipMemcpy(h.data(), d_out, grid * 12, hipMemcpyDeviceToHost);
  int hist[16] = { 0 }, rr = 0;
  for (int i = 0; i < grid; ++i) { hist[h[3 * i] & 15]++; if ((h[3 * i] & 15) == (unsigned)(i % 8)) ++rr; }
  printf("grid %d x %d threads: raw XCC_ID of workgroup 0: 0x%x\nworkgroups per XCC_ID & 15:", grid, threads, h[0]);
  for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
  printf("\nworkgroups with XCC_ID == blockIdx %% 8: %d of %d\n", rr, grid);
  printf("first 32 workgroups (blockIdx: xcc, start order, cu from HW_ID bits 8-11, se 13-15):\n");
  for (int i = 0; i < 32 && i < grid; ++i) printf("  %d: xcc %u order %u cu %u se %u\n", i, h[3 * i] & 15, h[3 * i + 1], (h[3 * i + 2] >> 8) & 15, (h[3 * i + 2] >> 13) & 7);
  // start order: which XCDs do the first 97 starters sit on, and how many share a (xcc, se, cu)?
  std::vector<int> by_order(grid);
  for (int i = 0; i < grid; ++i) by_order[h[3 * i + 1]] = i;
  int first[16] = { 0 };
  for (int k = 0; k < 97 && k < grid; ++k) first[h[3 * by_order[k]] & 15]++;
  printf("XCC_ID of the first 97 workgroups to take a ticket:");
  for (int i = 0; i < 8; ++i) printf(" %d", first[i]);
  printf("\n");
  return 0;
}
