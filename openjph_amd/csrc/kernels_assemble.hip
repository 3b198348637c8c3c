// openjph_amd/csrc/kernels_assemble.hip -- codestream assembly on the device.
//
// In the reference the packet writer copies every code-block's bytes from the elastic allocator's
// chunks into the output file behind its packet header (precinct::write, ojph_precinct.cpp:281-324:
// `file->write(cb->next_coded->buf ...)` per block).  Here the block coder leaves the blocks in HBM in
// the order their wavefronts finished; the host codes the packet headers from the block LENGTHS alone
// (ojph_t2.cpp, a layout: blob + placement jobs), and this kernel lays the codestream out in HBM --
// markers / headers from the blob, code-block bytes from the coder's output -- so that ONE device-to-host
// copy delivers the finished codestream and no coded byte passes through a host memcpy.
//
// One wavefront per placement job (a code-block is 1-4 KB: 64 lanes x 16 bytes cover 1 KB per step).
// Source and destination are byte-aligned arbitrarily: the destination is brought to 16-byte alignment
// with a bytewise head, then every lane stores 16 aligned bytes assembled from five aligned source
// dwords with v_alignbyte_b32; a bytewise tail finishes.  HBM traffic = 2 x codestream bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "ojph_plan.h"

namespace {

constexpr int WAVES = 4;

__global__ __launch_bounds__(WAVES * 64) void assemble_codestream(const ojphgpu::T2Job* __restrict__ jobs, uint32_t njobs,
                                                                   const uint8_t* __restrict__ blob, const uint8_t* __restrict__ data,
                                                                   uint8_t* __restrict__ out)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t j = blockIdx.x * WAVES + wave;
  if (j >= njobs) return;
  const ojphgpu::T2Job job = jobs[j];
  const uint8_t* src = (job.blob ? blob : data) + job.src;
  uint8_t* dst = out + job.dst;
  uint32_t n = job.n;
  // head: up to 15 bytes until dst is 16-byte aligned
  uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
  if (head > n) head = n;
  if (lane < head) dst[lane] = src[lane];
  src += head; dst += head; n -= head;
  // body: 16 bytes per lane and step
  const uint32_t sh = (uint32_t)((uintptr_t)src & 3u);
  const uint32_t* s4 = (const uint32_t*)(src - sh);               // aligned dword holding src[0]
  uint4* d16 = (uint4*)dst;
  const uint32_t n16 = n >> 4;
  for (uint32_t i = lane; i < n16; i += 64) {
    const uint32_t* q = s4 + 4 * i;
    const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
    uint4 v;
    if (sh == 0) { v.x = w0; v.y = w1; v.z = w2; v.w = w3; }
    else {
      const uint32_t w4 = q[4];                                   // within the source: src + 16 i + 16 <= end, and sh > 0
      v.x = __builtin_amdgcn_alignbyte(w1, w0, sh); v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
      v.z = __builtin_amdgcn_alignbyte(w3, w2, sh); v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
    }
    d16[i] = v;
  }
  // tail: up to 15 bytes
  const uint32_t done = n16 << 4, tail = n - done;
  if (lane < tail) dst[done + lane] = src[done + lane];
}

// Device -> pinned host memory by a kernel instead of hipMemcpyAsync: on this platform the runtime puts
// host-to-device and device-to-host copies of different streams on the SAME SDMA engine, one after the
// other (measured, profiles/r02_c_pipeline_timeline.txt: the next frame's 3.5 ms upload waited for the
// previous frame's 1.3 ms download), which cost the encode pipeline a third of its frame rate (5.6 -> 3.95 ms
// per 8K frame).  Stores from a kernel go over PCIe as posted writes beside the SDMA upload.  The grid is
// small: the copy is PCIe-bound (55 GB/s alone, ~47 GB/s beside an upload) and 8 workgroups walking
// contiguous segments were the fastest of 8 / 16 / 32 / 64 / 128 / 512 (profiles/r02_d_copy_modes.txt);
// the CUs stay with the next frame's kernels.
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_to_host_kernel(v4u* __restrict__ dst, const v4u* __restrict__ src, size_t n16, size_t seg,
                                                            uint8_t* __restrict__ dst_tail, const uint8_t* __restrict__ src_tail, uint32_t tail)
{
  // every workgroup walks ONE contiguous segment front to back, 16 KB (4 x 16 bytes per lane) per step: the
  // host sees a few sequential write streams of full-size PCIe payloads
  const size_t lo = (size_t)blockIdx.x * seg, hi = lo + seg < n16 ? lo + seg : n16;
  size_t i = lo + threadIdx.x;
  for (; i + 768 < hi; i += 1024) {
    const v4u a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    __builtin_nontemporal_store(a, &dst[i]); __builtin_nontemporal_store(b, &dst[i + 256]);
    __builtin_nontemporal_store(c, &dst[i + 512]); __builtin_nontemporal_store(d, &dst[i + 768]);
  }
  for (; i < hi; i += 256) { const v4u a = src[i]; __builtin_nontemporal_store(a, &dst[i]); }
  if (blockIdx.x == 0 && threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// byte counter + overflow flag of the block coder (device words) -> two words of pinned host memory
__global__ void publish_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t n)
{
  if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
}

}  // namespace

namespace ojphgpu {

// src: device memory, 16-byte aligned; d_dst: the DEVICE address of pinned host memory (hipHostGetDevicePointer), 16-byte aligned
int copy_to_host_launch(void* stream, void* d_dst, const void* src, size_t bytes)
{
  if (bytes == 0) return OJPHGPU_OK;
  if (!d_dst || !src || (((uintptr_t)d_dst | (uintptr_t)src) & 15u)) return OJPHGPU_E_INVALID;
  const size_t n16 = bytes >> 4; const uint32_t tail = (uint32_t)(bytes & 15u);
  static const unsigned max_blocks = [] { const char* e = getenv("OJPHGPU_COPY_WGS"); const long v = e ? atol(e) : 0; return v > 0 && v <= 4096 ? (unsigned)v : 8u; }();
  size_t seg = (n16 + max_blocks - 1) / max_blocks;
  seg = (seg + 1023) & ~(size_t)1023;                        // whole 16 KB steps
  const unsigned blocks = (unsigned)std::max<size_t>(1, (n16 + seg - 1) / seg);
  hipLaunchKernelGGL(copy_to_host_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (v4u*)d_dst, (const v4u*)src, n16, seg,
                     (uint8_t*)d_dst + (n16 << 4), (const uint8_t*)src + (n16 << 4), tail);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

int publish_words_launch(void* stream, uint32_t* d_dst, const uint32_t* src, uint32_t n)
{
  if (!d_dst || !src || n > 64) return OJPHGPU_E_INVALID;
  hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_dst, src, n);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

int assemble_launch(void* stream, const T2Job* d_jobs, uint32_t njobs, const uint8_t* d_blob, const uint8_t* d_data, uint8_t* d_out)
{
  if (njobs == 0) return OJPHGPU_OK;
  if (!d_jobs || !d_blob || !d_out) return OJPHGPU_E_INVALID;
  hipLaunchKernelGGL(assemble_codestream, dim3((njobs + WAVES - 1) / WAVES), dim3(WAVES * 64), 0, (hipStream_t)stream,
                     d_jobs, njobs, d_blob, d_data, d_out);
  return hipGetLastError() == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

}  // namespace ojphgpu
