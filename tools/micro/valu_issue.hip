// tools/micro/valu_issue.hip -- how many cycles does a SIMD of gfx950 spend per wave64 VALU instruction?
//
// The block coder is bound by VALU issue; bench.py's `roofline_valu` prices its launches with
// this probe's answer.  MI355X_MICROARCH.md lists 2 cycles for v_fma_f32 (SIMD-32); the question is what the INTEGER and
// bit-manipulation instructions the coder consists of cost, and what the packed / DPP / cross-lane forms cost.
//
// Method: one workgroup per CU-slot of W waves per SIMD (4 W waves), every wave runs `iters` x 64 copies of one
// instruction over 8 independent accumulator registers (no dependency stalls: consecutive copies use different
// registers), timed with s_memtime around the loop (shader-clock ticks) and with HIP events around the launch.
//   cycles per instruction per SIMD = ticks of the slowest wave / (W x instructions per wave)   at W waves per SIMD
// A dependent chain (1 accumulator) gives the latency.  Output: one line per instruction, W in {1, 2, 4} inside one
// workgroup (1024 threads at most), and the whole chip at 8 waves per SIMD from HIP events.
//
// build:  hipcc --offload-arch=gfx950 -O2 -o tools/micro/valu_issue tools/micro/valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// OPSTR uses %0 = the accumulator (read + written), %1 / %2 = two other VGPR sources
#define KERNEL(NAME, OPSTR, DEP)                                                                              \
  __global__ __launch_bounds__(1024) void NAME(unsigned long long* ticks, unsigned* sink, int iters)         \
  {                                                                                                           \
    __shared__ unsigned lds_probe[2048];                                                                      \
    lds_probe[threadIdx.x] = (threadIdx.x * 4u) & 0xFFCu; lds_probe[threadIdx.x + 1024] = 0;                                  \
    unsigned a[8];                                                                                            \
    for (int i = 0; i < 8; ++i) a[i] = ((threadIdx.x * 2654435761u + i) & 0xFFCu);                            \
    unsigned b = (threadIdx.x * 4u) + 4u, c = (threadIdx.x & 7u) + 1u;                                       \
    if (iters < 0) sink[1] = lds_probe[b & 2047];                                               \
    __syncthreads();                                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < iters; ++it) {                                                                      \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                         \
        asm volatile(OPSTR : "+v"(a[DEP ? 0 : 0]) : "v"(b), "v"(c));                                          \
        asm volatile(OPSTR : "+v"(a[DEP ? 0 : 1]) : "v"(b), "v"(c));                                          \
        asm volatile(OPSTR : "+v"(a[DEP ? 0 : 2]) : "v"(b), "v"(c));                                          \
        asm volatile(OPSTR : "+v"(a[DEP ? 0 : 3]) : "v"(b), "v"(c));                                          \
        asm volatile(OPSTR : "+v"(a[DEP ? 0 : 4]) : "v"(b), "v"(c));                                          \
        asm volatile(OPSTR : "+v"(a[DEP ? 0 : 5]) : "v"(b), "v"(c));                                          \
        asm volatile(OPSTR : "+v"(a[DEP ? 0 : 6]) : "v"(b), "v"(c));                                          \
        asm volatile(OPSTR : "+v"(a[DEP ? 0 : 7]) : "v"(b), "v"(c));                                          \
      }                                                                                                       \
    }                                                                                                         \
    asm volatile("s_nop 0" ::: "memory");                                                                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    unsigned s = 0;                                                                                           \
    for (int i = 0; i < 8; ++i) s ^= a[i];                                                                    \
    if (s == 0x12345678u) sink[0] = s;                                                                        \
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;       \
  }

KERNEL(k_add_u32, "v_add_u32 %0, %0, %1", 0)
KERNEL(k_add_u32_dep, "v_add_u32 %0, %0, %1", 1)
KERNEL(k_and_b32, "v_and_b32 %0, %0, %1", 0)
KERNEL(k_or_b32, "v_or_b32 %0, %0, %1", 0)
KERNEL(k_xor_b32, "v_xor_b32 %0, %0, %1", 0)
KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1", 0)
KERNEL(k_mov_b32, "v_mov_b32 %0, %1", 0)
KERNEL(k_not_b32, "v_not_b32 %0, %0", 0)
KERNEL(k_min_u32, "v_min_u32 %0, %0, %1", 0)
KERNEL(k_lshrrev, "v_lshrrev_b32 %0, %2, %0", 0)
KERNEL(k_lshl_const, "v_lshlrev_b32 %0, 3, %0", 0)
KERNEL(k_and_lit, "v_and_b32 %0, 0x12345, %0", 0)
KERNEL(k_add_lit, "v_add_u32 %0, 0x12345, %0", 0)
KERNEL(k_bfi, "v_bfi_b32 %0, %0, %1, %2", 0)
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1", 0)
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 1", 0)
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2", 0)
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2", 0)
KERNEL(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0", 0)
KERNEL(k_max_f32, "v_max_f32 %0, %0, %1", 0)
KERNEL(k_cndmask_sgpr, "v_cndmask_b32 %0, %0, %1, s[20:21]", 0)
KERNEL(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc", 0)
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %1", 0)
KERNEL(k_ds_read, "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)", 0)
KERNEL(k_ds_or, "ds_or_b32 %1, %0", 0)
KERNEL(k_ds_write_b8, "ds_write_b8 %1, %0", 0)
KERNEL(k_lshlrev, "v_lshlrev_b32 %0, %2, %0", 0)
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, %2, %1", 0)
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2", 0)
KERNEL(k_bfe, "v_bfe_u32 %0, %0, %2, %2", 0)
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2", 0)
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0", 0)
KERNEL(k_max_u32, "v_max_u32 %0, %0, %1", 0)
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", 0)
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1", 0)
KERNEL(k_cmp_sgpr, "v_cmp_lt_u32 s[20:21], %0, %1", 0)
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", 0)
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1", 0)
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1", 0)
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1", 0)
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2", 0)
KERNEL(k_fma_f32_dep, "v_fma_f32 %0, %0, %1, %2", 1)
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0", 0)
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0", 0)
KERNEL(k_frexp_exp, "v_frexp_exp_i32_f32 %0, %0", 0)
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1", 0)
KERNEL(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1", 0)
KERNEL(k_pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, %2, %0", 0)
KERNEL(k_mov_dpp_row, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", 0)
KERNEL(k_mov_dpp_wave, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf", 0)
KERNEL(k_add_dpp_row, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", 0)
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0", 0)
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", 0)
KERNEL(k_sad, "v_sad_u32 %0, %0, %1, %2", 0)
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2", 0)
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2", 0)
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3", 0)
KERNEL(k_readfirstlane, "v_readfirstlane_b32 s20, %0", 0)
KERNEL(k_bpermute, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)", 0)
KERNEL(k_salu_add, "s_add_u32 s20, s20, s21", 0)
KERNEL(k_salu_bfe, "s_bfe_u32 s20, s20, s21", 0)
KERNEL(k_salu_ff1, "s_ff1_i32_b64 s20, s[22:23]", 0)

// the same with 64-bit accumulators (register pairs): packed fp32 and 64-bit shifts
#define KERNEL64(NAME, OPSTR)                                                                                 \
  __global__ __launch_bounds__(1024) void NAME(unsigned long long* ticks, unsigned* sink, int iters)         \
  {                                                                                                           \
    unsigned long long a[8];                                                                                  \
    for (int i = 0; i < 8; ++i) a[i] = 0x3f8000003f800000ull + threadIdx.x + i;                               \
    unsigned long long b = 0x3f8000013f800001ull; unsigned c = (threadIdx.x & 7u) + 1u;                       \
    __syncthreads();                                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < iters; ++it) {                                                                      \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(OPSTR : "+v"(a[j]) : "v"(b), "v"(c));      \
      }                                                                                                       \
    }                                                                                                         \
    asm volatile("s_nop 0" ::: "memory");                                                                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    unsigned long long s = 0;                                                                                 \
    for (int i = 0; i < 8; ++i) s ^= a[i];                                                                    \
    if (s == 0x12345678ull) sink[0] = (unsigned)s;                                                            \
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;       \
  }
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, %2, %0")
KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, %2, %0")

struct Entry { const char* name; void (*fn)(unsigned long long*, unsigned*, int); bool wide; };

int main(int argc, char** argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  printf("# device %s, %d CUs, clock %d kHz; %d x 64 instructions per wave; ticks = s_memtime\n", prop.name, prop.multiProcessorCount, prop.clockRate, iters);
  unsigned long long* d_ticks; unsigned* d_sink;
  const int maxw = 32 * 1024;
  hipMalloc(&d_ticks, sizeof(unsigned long long) * maxw); hipMalloc(&d_sink, 64);
  std::vector<Entry> es = {
#define E(n) { #n, n, false }
    E(k_add_u32), E(k_add_u32_dep), E(k_and_b32), E(k_or_b32), E(k_xor_b32), E(k_sub_u32), E(k_mov_b32), E(k_not_b32), E(k_min_u32), E(k_lshrrev), E(k_lshl_const), E(k_and_lit), E(k_add_lit), E(k_bfi), E(k_lshl_add), E(k_add_lshl), E(k_mad_u24), E(k_or3), E(k_cvt_u32_f32), E(k_max_f32), E(k_cndmask_sgpr), E(k_cmp_cndmask), E(k_add_co), E(k_ds_read), E(k_ds_or), E(k_ds_write_b8), E(k_lshlrev), E(k_lshl_or), E(k_add3), E(k_bfe), E(k_alignbit), E(k_ffbh), E(k_max_u32),
    E(k_cndmask), E(k_cmp), E(k_cmp_sgpr), E(k_mul_lo), E(k_mul_u24), E(k_add_f32), E(k_mul_f32), E(k_fma_f32), E(k_fma_f32_dep), E(k_cvt_i32_f32),
    E(k_cvt_f32_u32), E(k_frexp_exp), E(k_pk_add_u16), E(k_pk_max_u16), E(k_pk_lshlrev_b16), E(k_mov_dpp_row), E(k_mov_dpp_wave), E(k_add_dpp_row),
    E(k_mbcnt), E(k_perm), E(k_sad), E(k_xad), E(k_and_or), E(k_readlane), E(k_readfirstlane), E(k_bpermute), E(k_salu_add), E(k_salu_bfe), E(k_salu_ff1), E(k_pk_mul_f32), E(k_pk_add_f32), E(k_lshlrev_b64), E(k_lshrrev_b64),
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-18s %s\n", "instruction", "cycles per instruction per SIMD at 1 / 2 / 4 waves per SIMD (s_memtime, one workgroup)   | whole chip, 8 waves per SIMD: ns per instruction per SIMD (HIP events)");
  for (const Entry& e : es) {
    printf("%-18s", e.name + 2);
    for (int W : { 1, 2, 4 }) {
      const int waves = 4 * W;                                   // one workgroup = one CU's worth
      hipLaunchKernelGGL(e.fn, dim3(1), dim3(64 * waves), 0, 0, d_ticks, d_sink, iters);   // warm
      hipLaunchKernelGGL(e.fn, dim3(1), dim3(64 * waves), 0, 0, d_ticks, d_sink, iters);
      hipDeviceSynchronize();
      std::vector<unsigned long long> t(waves);
      hipMemcpy(t.data(), d_ticks, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
      const double worst = (double)*std::max_element(t.begin(), t.end());
      printf(" %6.2f", worst / ((double)W * iters * 64.0));
    }
    {                                                            // whole chip: 256 CUs x 8 waves per SIMD
      const int wgs = 2 * prop.multiProcessorCount, waves = 16;  // two 1024-thread workgroups per CU = 8 waves per SIMD
      hipLaunchKernelGGL(e.fn, dim3(wgs), dim3(64 * waves), 0, 0, d_ticks, d_sink, iters);
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(wgs), dim3(64 * waves), 0, 0, d_ticks, d_sink, iters * 4);
      hipEventRecord(e1); hipEventSynchronize(e1);
      if (hipGetLastError() != hipSuccess) { printf("   | launch failed\n"); continue; }
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      const double per_simd = 8.0 * iters * 4 * 64.0;            // instructions each SIMD issued
      printf("   | %.3f ns (= %.2f cycles at 2.4 GHz)", ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    }
    printf("\n");
  }
  return 0;
}
