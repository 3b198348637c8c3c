// openjph_amd/csrc/kernels_lift.hip -- the wavelet transform in its GENERAL form: any lifting kernel an ATK marker
// segment can describe, levels that transform one direction only (DFS marker segment), 32-bit integer, 64-bit integer
// (the reference's 64-bit sample path) or float samples.
//
// Reference: every wavelet of the reference runs through one code path -- a list of lifting steps in synthesis order
// (param_atk, ojph_params.cpp:2654-2896: 5/3 = {(a 1, b 2, e 2), (a -1, b 1, e 1)}, 9/7 = four float steps and K) applied by
// gen_rev_vert_step32 / 64 (ojph_transform.cpp:209-333), gen_rev_horz_ana32 / 64 (:336-512), gen_rev_horz_syn32 / 64
// (:514-688), gen_irv_vert_step / _times_K / _horz_ana / _horz_syn (:691-852) under resolution::push_line / pull_line
// (ojph_resolution.cpp:547-949), which also holds the one-direction levels (transform_flags, :290-300, :725-949).
//
// The two wavelets every ordinary codestream uses have kernels of their own (kernels_dwt.hip: register pipelines, one
// launch per level, bound by HBM).  This file is the path of everything else -- rare by nature (Part-2 codestreams,
// samples deeper than 26 bits) -- and it is built for being obviously the reference's arithmetic rather than for the last
// byte per second: a level is a short sequence of element-wise launches over the plane, every one of them coalesced
// (a thread owns one sample of a row, neighbouring threads neighbouring samples):
//   analysis : [vertical: one launch per lifting step, steps N-1 .. 0, the first one updating the high-pass rows; then the
//              irreversible K scaling of the rows / the doubling of a one-row plane at an odd coordinate]
//              [horizontal: the same along the rows] [de-interleave into LL / HL / LH / HH]
//   synthesis: [interleave] [horizontal: K scaling / halving, then steps 0 .. N-1, step 0 updating the low-pass samples]
//              [vertical: the same]
// A step updates the samples of one sub-sequence from their two neighbours of the other one, so the threads of a launch
// never read what another thread of the same launch writes; a missing neighbour at the border is replaced by the one
// that exists (the reference's lp[-1] = lp[0], lp[w] = lp[w-1], :373-374, and "sp1 = sig->active ? sig : ssp[i]",
// ojph_resolution.cpp:575-578).  Samples are transformed in place, interleaved, in the plane of the resolution: the
// arithmetic is the reference's, value for value -- (b + a (l + r)) >> e added or subtracted (its a = +-1 special cases
// compute the same number), x +- A (l + r) with an fp32 add, multiply, add and no contraction, K applied where the
// reference applies it (including the sub-sequence its horizontal analysis scales by 1 / K after an odd number of
// steps: the one its swapped pointers end up on, :765-777).
// Traffic: (2 N + 2) passes over the plane per level instead of one -- accepted for this path, see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>
#include "../../include/ojphgpu.h"

namespace ojphgpu {
// kernels_dwt.hip: the same level in ONE launch of the register pipeline, for kernels of up to four steps that transform
// both directions
bool dwt_general_pipeline_fits(const ojphgpu_lift* k);
int dwt_general_pipeline(void* stream, const ojphgpu_lift* k, const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
                         void* d_base, bool synthesis);
}

namespace {

struct LiftArgs {
  const ojphgpu_dwt_desc* descs; uint32_t n;          // the planes of the batch
  uint32_t* base;
  int dir;            // 0: along the rows (horizontal), 1: along the columns (vertical)
  int tgt_high;       // the step updates the high-pass samples (odd canvas coordinates)
  int synthesis;
  int a, b, e; float A;
};

template <typename T> __device__ __forceinline__ T* plane(uint32_t* base, uint64_t off) { return reinterpret_cast<T*>(base + off); }

// one lifting step over every plane of the batch
// (the grid's y and z are capped at 65535: rows and planes beyond that are walked by the same workgroups)
#define LIFT_FOR_EACH_PLANE_ROW(q, d, x, y)                                                   \
  for (uint32_t z_ = blockIdx.z; z_ < (q).n; z_ += gridDim.z)                                  \
    for (uint32_t y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x; y < (q).descs[z_].h; y += gridDim.y) \
      if (const ojphgpu_dwt_desc& d = (q).descs[z_]; x < d.w)

template <typename T>
__device__ __forceinline__ void lift_step_sample(const LiftArgs& q, const ojphgpu_dwt_desc& d, uint32_t x, uint32_t y)
{
  const uint32_t n = q.dir ? d.h : d.w, pos = q.dir ? y : x;
  if (n <= 1) return;
  const bool even = (q.dir ? d.y_even : d.x_even) != 0;
  const bool high = ((pos & 1u) == 0u) != even;
  if ((int)high != q.tgt_high) return;
  const uint32_t l0 = pos == 0 ? 1u : pos - 1u, r0 = pos + 1u >= n ? pos - 1u : pos + 1u;
  T* p = plane<T>(q.base, d.src_off);
  const size_t stride = q.dir ? d.src_pitch : 1u, row = q.dir ? x : (size_t)y * d.src_pitch;
  const T lv = p[row + (size_t)l0 * stride], rv = p[row + (size_t)r0 * stride];
  T& t = p[row + (size_t)pos * stride];
  if constexpr (std::is_floating_point<T>::value) {                       // float
    const float m = __fmul_rn(q.A, __fadd_rn(lv, rv));
    t = q.synthesis ? __fsub_rn(t, m) : __fadd_rn(t, m);
  } else {
    // (the shift counts modulo the width of T, as the reference's shifts do on the hosts it runs on and the oracle restates:
    // an Eatk byte may say anything up to 255)
    const T v = (T)(((T)q.b + (T)q.a * (T)(lv + rv)) >> (q.e & (int)(8 * sizeof(T) - 1)));
    t = q.synthesis ? (T)(t - v) : (T)(t + v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void lift_step_kernel(LiftArgs q)
{
  LIFT_FOR_EACH_PLANE_ROW(q, d, x, y) lift_step_sample<T>(q, d, x, y);
}

// what surrounds the steps of one direction: the irreversible K scaling (analysis: after the steps, synthesis: before
// them) and the one-sample sequence at an odd coordinate (doubled / halved); lp_is_high: the horizontal analysis of the
// reference scales the sub-sequence its `lp` pointer ends on by 1 / K -- the high-pass one after an odd number of steps
struct ScaleArgs { const ojphgpu_dwt_desc* descs; uint32_t n; uint32_t* base; int dir, synthesis, lp_is_high; float K, Kinv; };

template <typename T>
__device__ __forceinline__ void lift_scale_sample(const ScaleArgs& q, const ojphgpu_dwt_desc& d, uint32_t x, uint32_t y)
{
  const uint32_t n = q.dir ? d.h : d.w, pos = q.dir ? y : x;
  const bool even = (q.dir ? d.y_even : d.x_even) != 0;
  T* p = plane<T>(q.base, d.src_off);
  T& t = p[(size_t)y * d.src_pitch + x];
  if (n == 1) {
    if (even) return;
    if constexpr (std::is_floating_point<T>::value) t = __fmul_rn(t, q.synthesis ? 0.5f : 2.0f);
    else t = q.synthesis ? (T)(t >> 1) : (T)(t * 2);
    return;
  }
  if constexpr (std::is_floating_point<T>::value) {
    const bool high = ((pos & 1u) == 0u) != even;
    // synthesis: low x K, high x 1 / K (:799-810, ojph_resolution.cpp:855-870); analysis: "lp" x 1 / K, "hp" x K
    const bool times_K = q.synthesis ? !high : (high != (q.lp_is_high != 0));
    t = __fmul_rn(t, times_K ? q.K : q.Kinv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void lift_scale_kernel(ScaleArgs q)
{
  LIFT_FOR_EACH_PLANE_ROW(q, d, x, y) lift_scale_sample<T>(q, d, x, y);
}

// plane <-> sub-bands.  horz / vert = 0: the level does not transform that direction, all its samples are "low" there
struct SplitArgs { const ojphgpu_dwt_desc* descs; uint32_t n; uint32_t* base; int horz, vert; };

template <typename T, bool JOIN>
__device__ __forceinline__ void lift_split_sample(const SplitArgs& q, const ojphgpu_dwt_desc& d, uint32_t x, uint32_t y)
{
  const bool xh = q.horz && (((x & 1u) == 0u) != (d.x_even != 0));
  const bool yh = q.vert && (((y & 1u) == 0u) != (d.y_even != 0));
  // index inside the sub-sequence: every pair of positions (2k, 2k + 1) holds one sample of each kind, whatever the parity
  const uint32_t bx = q.horz ? x >> 1 : x;
  const uint32_t by = q.vert ? y >> 1 : y;
  const uint64_t off = yh ? (xh ? d.hh_off : d.lh_off) : (xh ? d.hl_off : d.ll_off);
  const uint32_t pitch = yh ? (xh ? d.hh_pitch : d.lh_pitch) : (xh ? d.hl_pitch : d.ll_pitch);
  T* band = plane<T>(q.base, off);
  T* src = plane<T>(q.base, d.src_off);
  if (JOIN) src[(size_t)y * d.src_pitch + x] = band[(size_t)by * pitch + bx];
  else band[(size_t)by * pitch + bx] = src[(size_t)y * d.src_pitch + x];
}

template <typename T, bool JOIN>
__global__ __launch_bounds__(256) void lift_split_kernel(SplitArgs q)
{
  LIFT_FOR_EACH_PLANE_ROW(q, d, x, y) lift_split_sample<T, JOIN>(q, d, x, y);
}

template <typename T>
int level(hipStream_t s, const ojphgpu_lift* k, const ojphgpu_dwt_desc* descs, uint32_t n, uint32_t max_w, uint32_t max_h,
          void* base, bool synthesis)
{
  const dim3 grid((max_w + 255) / 256, max_h < 65535u ? max_h : 65535u, n < 65535u ? n : 65535u), wg(256);   // (the kernels walk what a grid cannot hold)
  const float K = k->K, Kinv = 1.0f / k->K;                    // (1.0f / K in fp32, as gen_irv_horz_ana computes it: host code, no contraction)
  auto steps = [&](int dir) {
    for (uint32_t i = 0; i < k->num_steps; ++i) {
      const uint32_t j = synthesis ? i : k->num_steps - 1u - i;
      LiftArgs a{ descs, n, (uint32_t*)base, dir, synthesis ? (int)(i & 1u) : (int)!(i & 1u), synthesis ? 1 : 0,
                  k->steps[j].a, k->steps[j].b, k->steps[j].e, k->steps[j].A };
      hipLaunchKernelGGL(lift_step_kernel<T>, grid, wg, 0, s, a);
    }
  };
  auto scale = [&](int dir) {
    ScaleArgs a{ descs, n, (uint32_t*)base, dir, synthesis ? 1 : 0, (dir == 0 && !synthesis) ? (int)(k->num_steps & 1u) : 0, K, Kinv };
    hipLaunchKernelGGL(lift_scale_kernel<T>, grid, wg, 0, s, a);
  };
  SplitArgs sp{ descs, n, (uint32_t*)base, k->horz ? 1 : 0, k->vert ? 1 : 0 };
  if (!synthesis) {
    if (k->vert) { steps(1); scale(1); }
    if (k->horz) { steps(0); scale(0); }
    hipLaunchKernelGGL((lift_split_kernel<T, false>), grid, wg, 0, s, sp);
  } else {
    hipLaunchKernelGGL((lift_split_kernel<T, true>), grid, wg, 0, s, sp);
    if (k->horz) { scale(0); steps(0); }
    if (k->vert) { scale(1); steps(1); }
  }
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

int run(void* stream, const ojphgpu_lift* k, const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
        void* d_base, bool synthesis)
{
  if (!k || !d_descs || !d_base || k->num_steps > OJPHGPU_MAX_LIFT_STEPS || k->elem > 2) return OJPHGPU_E_INVALID;
  if (n == 0 || max_w == 0 || max_h == 0) return OJPHGPU_OK;
  for (uint32_t i = 0; i < k->num_steps; ++i) if (k->elem != 2 && (k->steps[i].e < 0 || k->steps[i].e > 255)) return OJPHGPU_E_INVALID;   // (an Eatk byte; counted modulo the sample width)
  // the usual Part-2 kernels and the 5/3 on 64-bit samples go through the register pipeline (one launch per level -- also the
  // levels of a DFS decomposition that transform one direction only); kernels of more than four steps take the launches below;
  // OJPHGPU_LIFT_ELEMENTWISE=1 keeps every level on the element-wise launches below (A/B runs, and the pin of the two forms
  // against each other in tests/test_gpu_wide.py)
  static const bool elementwise = [] { const char* e = getenv("OJPHGPU_LIFT_ELEMENTWISE"); return e && atoi(e) != 0; }();
  if (!elementwise && ojphgpu::dwt_general_pipeline_fits(k)) return ojphgpu::dwt_general_pipeline(stream, k, d_descs, n, max_w, max_h, d_base, synthesis);
  hipStream_t s = (hipStream_t)stream;
  if (k->elem == 0) return level<int>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
  if (k->elem == 1) return level<long long>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
  return level<float>(s, k, d_descs, n, max_w, max_h, d_base, synthesis);
}

}  // namespace

extern "C" int ojphgpu_dwt_forward_general(void* stream, const ojphgpu_lift* kernel, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                                            uint32_t max_w, uint32_t max_h, void* d_base)
{
  return run(stream, kernel, d_descs, n, max_w, max_h, d_base, false);
}

extern "C" int ojphgpu_dwt_inverse_general(void* stream, const ojphgpu_lift* kernel, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                                            uint32_t max_w, uint32_t max_h, void* d_base)
{
  return run(stream, kernel, d_descs, n, max_w, max_h, d_base, true);
}
