// openjph_amd/csrc/ojphgpu_pipe.cpp -- frame pipelines: what ojph::codestream's exchange() ... flush() /
// read_headers() ... pull() contract (ojph_codestream_local.cpp:1148-1270, :912-1146) becomes when the hot
// path runs on a GPU behind PCIe and frames follow each other (video, image sequences, a tiled image cut
// into independent codestreams).
//
// One frame alone is bound by the PCIe copies either side of 0.6 ms of kernels; a SEQUENCE is not, if the
// stages of consecutive frames overlap.  A pipe keeps `depth` frames in flight, each in a slot of its own:
//
//   encoder   caller fills slot n+1's pinned frame  |  H2D of n+1  |  kernels of n  |  D2H of n-1's block
//             lengths -> packet headers on host threads -> placement kernel -> D2H of n-1's finished codestream
//   decoder   host threads parse n+1's packet headers  |  H2D of n+1's bytes + descriptors  |  kernels of n  |
//             D2H of n-1's frame into pinned memory the caller reads rows from
//
// on separate HIP streams (copy-in, compute, copy-out) with events between them.  The coded bytes never
// pass through a host memcpy: the encoder's codestream is assembled in HBM (kernels_assemble.hip) from the
// layout the host Tier-2 computes out of the block LENGTHS, and arrives in pinned memory ready to be
// written to a file; the decoder uploads the codestream as it is and addresses the blocks inside it.
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>

#include "ojphgpu_objects.h"
#include "ojph_pool.h"

namespace ojphgpu {
int assemble_launch(void* stream, const T2Job* d_jobs, uint32_t njobs, const uint8_t* d_blob, const uint8_t* d_data, uint8_t* d_out);
int copy_to_host_launch(void* stream, void* d_dst, const void* src, size_t bytes);
int publish_words_launch(void* stream, uint32_t* d_dst, const uint32_t* src, uint32_t n);
}

namespace {

// Pinned host memory, mapped into the device's address space and coherent: uploads from it go through
// hipMemcpyAsync (SDMA); what comes BACK is written into it by kernels (`d` = its device address), see
// kernels_assemble.hip for why.
struct Pinned {
  uint8_t* p = nullptr; uint8_t* d = nullptr; size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    release();
    void* q = nullptr; void* dq = nullptr;
    if (hipHostMalloc(&q, bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (hipHostGetDevicePointer(&dq, q, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(q); return -1; }
    p = (uint8_t*)q; d = (uint8_t*)dq; cap = bytes;
    return 0;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; d = nullptr; cap = 0; }
};

struct Grow {                       // device buffer that grows (never shrinks)
  DeviceBuf b; size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    b.release();
    if (b.alloc(bytes + 64)) { cap = 0; return -1; }
    cap = bytes;
    return 0;
  }
};

enum SlotState { FREE = 0, ACQUIRED, SUBMITTED, DONE, HELD };

// Which engine moves which direction (the two directions of one pipe must not share the SDMA engine, see
// kernels_assemble.hip): 0 = upload by hipMemcpyAsync (SDMA), download by a copy kernel; 1 = upload by a copy
// kernel reading the pinned memory, download by hipMemcpyAsync; 2 = both by hipMemcpyAsync (the runtime decides).
// The stream of the device -> host copies: those are kernels (see kernels_assemble.hip) and must get their few
// workgroups onto the chip while the block coder of the next frames keeps every CU full -- highest priority.
hipError_t create_copy_out_stream(hipStream_t* s)
{
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (getenv("OJPHGPU_COPY_PRIO_OFF")) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}

int copy_mode(const char* name, int dflt)
{
  const char* e = getenv(name);
  const int v = e ? atoi(e) : dflt;
  return v >= 0 && v <= 2 ? v : dflt;
}

int upload(int mode, hipStream_t st, void* d_dst, const Pinned& src, size_t src_off, size_t bytes);
int download(int mode, hipStream_t st, const Pinned& dst, const void* d_src, size_t bytes);

int upload(int mode, hipStream_t st, void* d_dst, const Pinned& src, size_t src_off, size_t bytes)
{
  if (bytes == 0) return OJPHGPU_OK;
  if (mode == 1 && ((src_off | (uintptr_t)d_dst) & 15u) == 0) return copy_to_host_launch(st, d_dst, src.d + src_off, bytes);   // the kernel copies either way
  return hipMemcpyAsync(d_dst, src.p + src_off, bytes, hipMemcpyHostToDevice, st) == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

int download(int mode, hipStream_t st, const Pinned& dst, const void* d_src, size_t bytes)
{
  if (bytes == 0) return OJPHGPU_OK;
  if (mode == 0) return copy_to_host_launch(st, dst.d, d_src, bytes);
  return hipMemcpyAsync(dst.p, d_src, bytes, hipMemcpyDeviceToHost, st) == hipSuccess ? OJPHGPU_OK : OJPHGPU_E_HIP;
}

double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// =============================================================================================
// encoder pipe
// =============================================================================================
struct EncSlot {
  SlotState state = FREE;
  Pinned h_in, h_res, h_lay, h_cs;
  DeviceBuf image, out, counters, pixels;           // pixels: the frame as it was handed over, when it comes pixel-interleaved
  Grow cs;
  hipEvent_t ev_in = nullptr, ev_kern = nullptr, ev_done = nullptr;
  int rc = 0; size_t cs_len = 0;
  double t_submit = 0, t_done = 0, t_t2 = 0;
};

struct ojphgpu_enc_pipe {
  const ojphgpu_plan* handle = nullptr; const Plan* P = nullptr;
  int device = 0, container = 16;
  int mode = copy_mode("OJPHGPU_ENC_COPY_MODE", 0);
  uint32_t depth = 0;
  ojphgpu_encoder* enc = nullptr;
  hipStream_t s_h2d = nullptr, s_comp = nullptr, s_d2h = nullptr;
  std::vector<EncSlot> slots;
  size_t frame_bytes = 0, res_bytes = 0;
  int pixel_bits = 0, big_endian = 0;               // != 0: frames are handed over pixel-interleaved (ojphgpu_enc_pipe_set_pixels)
  int packed_bits = 0;                              // != 0: ... as bit-packed planes (ojphgpu_enc_pipe_set_packed)
  size_t in_bytes = 0;                              // what _acquire hands out: frame_bytes, or the interleaved frame
  uint64_t n_acq = 0, n_sub = 0, n_col = 0;
  std::mutex mu; std::condition_variable cv_work, cv_done;
  std::deque<uint32_t> work; bool stop = false;
  std::vector<std::thread> finishers;
  double sum_t2 = 0, sum_latency = 0; uint64_t n_done = 0;
};

static void enc_finish_frame(ojphgpu_enc_pipe* p, EncSlot& s)
{
  const Plan& P = *p->P;
  ojphgpu_encoder* e = p->enc;
  auto fail = [&](int rc) { s.rc = rc; };
  if (hipSetDevice(p->device) != hipSuccess) return fail(OJPHGPU_E_HIP);
  if (hipEventSynchronize(s.ev_kern) != hipSuccess) return fail(OJPHGPU_E_HIP);
  const double t0 = now_ms();
  const size_t nb = e->block_ids.size();
  const ojphgpu_cb_result* res = (const ojphgpu_cb_result*)s.h_res.p;
  const uint32_t* cnt = (const uint32_t*)(s.h_res.p + nb * sizeof(ojphgpu_cb_result));
  if (cnt[1]) return fail(OJPHGPU_E_OVERFLOW);
  int rc = no_throw([&]() -> int {
    std::vector<ojphgpu_coded_block> cb(P.blocks.size(), ojphgpu_coded_block{ 0, 0, 0, 0, 0 });
    for (size_t i = 0; i < nb; ++i) {
      const ojphgpu_cb_result& r = res[i];
      ojphgpu_coded_block& c = cb[e->block_ids[i]];
      c.offset = r.offset; c.len1 = r.length; c.len2 = 0;
      c.missing_msbs = r.length ? P.bands[P.blocks[e->block_ids[i]].band].K_max - 1 : 0;      // ojph_codeblock.cpp:148
      c.num_passes = r.length ? 1 : 0;
    }
    T2Layout L;
    int r2 = t2_layout_codestream(P, cb.data(), L);
    if (r2) return r2;
    s.t_t2 = now_ms() - t0;
    // the layout goes into the slot's pinned staging -- jobs, then the blob (64-byte aligned) -- where the
    // placement kernel reads it in place; the codestream is laid out in HBM and written to the slot's pinned
    // output by a copy kernel
    const size_t jbytes = L.jobs.size() * sizeof(T2Job), boff = (jbytes + 63) & ~(size_t)63;
    const size_t lbytes = boff + L.blob.size() + 16;
    if (s.h_lay.reserve(lbytes + lbytes / 4)) return OJPHGPU_E_NOMEM;
    if (s.cs.reserve((size_t)L.total + (size_t)L.total / 8 + 64) || s.h_cs.reserve((size_t)L.total + (size_t)L.total / 8 + 64)) return OJPHGPU_E_NOMEM;
    memcpy(s.h_lay.p, L.jobs.data(), jbytes);
    memcpy(s.h_lay.p + boff, L.blob.data(), L.blob.size());
    r2 = assemble_launch(p->s_d2h, (const T2Job*)s.h_lay.d, (uint32_t)L.jobs.size(), s.h_lay.d + boff,
                         (const uint8_t*)s.out.p, (uint8_t*)s.cs.b.p);
    if (r2) return r2;
    r2 = download(p->mode, p->s_d2h, s.h_cs, s.cs.b.p, (size_t)L.total);
    if (r2) return r2;
    HIPCHK(hipEventRecord(s.ev_done, p->s_d2h));
    HIPCHK(hipEventSynchronize(s.ev_done));
    s.cs_len = (size_t)L.total;
    return OJPHGPU_OK;
  });
  if (rc) {
    // uploads, kernels or the copy-out of this slot may already be enqueued: nothing of the slot may be rewritten or
    // re-reserved (hipHostFree / hipFree in reserve()) by the next _acquire while they are in flight
    hipStreamSynchronize(p->s_h2d); hipStreamSynchronize(p->s_comp); hipStreamSynchronize(p->s_d2h);
    fail(rc);
  }
}

static void enc_finisher(ojphgpu_enc_pipe* p)
{
  for (;;) {
    uint32_t si;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_work.wait(lk, [&] { return p->stop || !p->work.empty(); });
      if (p->work.empty()) return;                   // stop requested and nothing left
      si = p->work.front(); p->work.pop_front();
    }
    EncSlot& s = p->slots[si];
    enc_finish_frame(p, s);
    {
      std::lock_guard<std::mutex> lk(p->mu);
      s.t_done = now_ms();
      s.state = DONE;
      p->sum_t2 += s.t_t2; p->sum_latency += s.t_done - s.t_submit; p->n_done++;
    }
    p->cv_done.notify_all();
  }
}

extern "C" void ojphgpu_enc_pipe_destroy(ojphgpu_enc_pipe* p)
{
  if (!p) return;
  (void)hipSetDevice(p->device);
  { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
  p->cv_work.notify_all();
  for (std::thread& t : p->finishers) t.join();
  for (hipStream_t s : { p->s_h2d, p->s_comp, p->s_d2h }) if (s) (void)hipStreamSynchronize(s);
  if (p->enc) ojphgpu_encoder_destroy(p->enc);
  for (EncSlot& s : p->slots) {
    s.h_in.release(); s.h_res.release(); s.h_lay.release(); s.h_cs.release();
    for (DeviceBuf* b : { &s.image, &s.out, &s.counters, &s.cs.b, &s.pixels }) b->release();
    for (hipEvent_t ev : { s.ev_in, s.ev_kern, s.ev_done }) if (ev) (void)hipEventDestroy(ev);
  }
  for (hipStream_t s : { p->s_h2d, p->s_comp, p->s_d2h }) if (s) (void)hipStreamDestroy(s);
  delete p;
}

extern "C" int ojphgpu_enc_pipe_create(const ojphgpu_plan* plan, int device, uint32_t depth, int container_bits,
                                        uint32_t host_threads, ojphgpu_enc_pipe** out)
{
  if (!plan || !out || depth < 2 || depth > 16 || (container_bits != 8 && container_bits != 16 && container_bits != 32)) return OJPHGPU_E_INVALID;
  *out = nullptr;
  return no_throw([&]() -> int {
    HIPCHK(hipSetDevice(device));
    ojphgpu_enc_pipe* p = new (std::nothrow) ojphgpu_enc_pipe();
    if (!p) return OJPHGPU_E_NOMEM;
    struct Owner { ojphgpu_enc_pipe* p; ~Owner() { if (p) ojphgpu_enc_pipe_destroy(p); } } owner{ p };
    const Plan& P = plan->plan;
    p->handle = plan; p->P = &P; p->device = device; p->container = container_bits; p->depth = depth;
    if (container_bits != 32) for (const CompGeo& g : P.comps) if (g.bit_depth > (uint32_t)container_bits) return OJPHGPU_E_INVALID;
    for (hipStream_t* s : { &p->s_h2d, &p->s_comp }) HIPCHK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    HIPCHK(create_copy_out_stream(&p->s_d2h));
    int rc = ojphgpu_encoder_create(plan, device, p->s_comp, &p->enc);
    if (rc) return rc;
    (void)ojphgpu_encoder_set_timing(p->enc, 0);
    ojphgpu_encoder* e = p->enc;
    const size_t nb = e->block_ids.size();
    p->frame_bytes = (size_t)P.frame_elems * (size_t)(container_bits / 8);
    p->in_bytes = p->frame_bytes;
    p->res_bytes = nb * sizeof(ojphgpu_cb_result) + 16;
    // the codestream of a frame: sized from the samples (1 byte each is generous for natural content), grown when a frame needs more
    const size_t cs_guess = std::min<size_t>((size_t)e->out_cap, (size_t)P.frame_elems + (1u << 20));
    p->slots.resize(depth);
    for (EncSlot& s : p->slots) {
      if (s.h_in.reserve(p->frame_bytes + 64) || s.h_res.reserve(p->res_bytes + 64) || s.h_cs.reserve(cs_guess)) return OJPHGPU_E_NOMEM;
      if (s.image.alloc(p->frame_bytes + 64) || s.out.alloc((size_t)e->out_cap + 64) || s.counters.alloc(e->counters_bytes) ||
          s.cs.reserve(cs_guess)) return OJPHGPU_E_NOMEM;
      const size_t lay_guess = nb * sizeof(T2Job) + nb * 8 + (1u << 16);
      if (s.h_lay.reserve(lay_guess)) return OJPHGPU_E_NOMEM;
      for (hipEvent_t* ev : { &s.ev_in, &s.ev_kern, &s.ev_done }) HIPCHK(hipEventCreateWithFlags(ev, hipEventDisableTiming | hipEventReleaseToSystem));
    }
    const uint32_t nthreads = host_threads ? std::min<uint32_t>(host_threads, 16) : 2;
    for (uint32_t i = 0; i < nthreads; ++i) p->finishers.emplace_back(enc_finisher, p);
    owner.p = nullptr;
    *out = p;
    return OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_enc_pipe_acquire(ojphgpu_enc_pipe* p, void** h_frame, size_t* bytes)
{
  if (!p || !h_frame) return OJPHGPU_E_INVALID;
  EncSlot& s = p->slots[p->n_acq % p->depth];
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (s.state == ACQUIRED) { *h_frame = s.h_in.p; if (bytes) *bytes = p->in_bytes; return OJPHGPU_OK; }   // asked twice
    if (s.state != FREE) return OJPHGPU_E_AGAIN;      // every slot is in flight: collect a codestream first
    s.state = ACQUIRED;
  }
  *h_frame = s.h_in.p;
  if (bytes) *bytes = p->in_bytes;
  return OJPHGPU_OK;
}

// the conditions of the pixel-interleaved hand-over: one size for all components, unsigned, depths that fit
static int pixels_fit(const Plan& P, int pixel_bits, int container_bits)
{
  if (pixel_bits != 8 && pixel_bits != 16) return OJPHGPU_E_INVALID;
  if (pixel_bits > container_bits) return OJPHGPU_E_INVALID;
  if (P.frame_elems != (uint64_t)P.p.width * P.p.height * P.p.num_comps) return OJPHGPU_E_INVALID;   // sub-sampled components
  for (const CompGeo& g : P.comps) if (g.is_signed || g.bit_depth > (uint32_t)pixel_bits) return OJPHGPU_E_INVALID;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_enc_pipe_set_pixels(ojphgpu_enc_pipe* p, int pixel_bits, int big_endian)
{
  if (!p || p->n_acq != 0 || p->slots[0].state != FREE || p->packed_bits) return OJPHGPU_E_INVALID;
  return no_throw([&]() -> int {
    if (pixel_bits == 0) { p->pixel_bits = 0; p->in_bytes = p->frame_bytes; return OJPHGPU_OK; }
    const Plan& P = *p->P;
    const int rc = pixels_fit(P, pixel_bits, p->container);
    if (rc) return rc;
    HIPCHK(hipSetDevice(p->device));
    const size_t nbytes = (size_t)P.frame_elems * (size_t)(pixel_bits / 8);
    for (EncSlot& s : p->slots) {
      if (s.h_in.reserve(nbytes + 64)) return OJPHGPU_E_NOMEM;
      if (!s.pixels.p && s.pixels.alloc(nbytes + 64)) return OJPHGPU_E_NOMEM;
    }
    p->pixel_bits = pixel_bits; p->big_endian = big_endian ? 1 : 0; p->in_bytes = nbytes;
    return OJPHGPU_OK;
  });
}

static size_t packed_bytes(uint64_t samples, int bits) { return (size_t)((samples + 31) / 32) * 4u * (size_t)bits; }

static int packed_fit(const Plan& P, int bits, int container_bits)
{
  if ((bits != 10 && bits != 12 && bits != 14) || (container_bits != 16 && container_bits != 32)) return OJPHGPU_E_INVALID;
  for (const CompGeo& g : P.comps) if (g.is_signed || g.bit_depth > (uint32_t)bits) return OJPHGPU_E_INVALID;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_enc_pipe_set_packed(ojphgpu_enc_pipe* p, int bits)
{
  if (!p || p->n_acq != 0 || p->slots[0].state != FREE || p->pixel_bits) return OJPHGPU_E_INVALID;
  return no_throw([&]() -> int {
    if (bits == 0) { p->packed_bits = 0; p->in_bytes = p->frame_bytes; return OJPHGPU_OK; }
    const Plan& P = *p->P;
    const int rc = packed_fit(P, bits, p->container);
    if (rc) return rc;
    HIPCHK(hipSetDevice(p->device));
    const size_t nbytes = packed_bytes(P.frame_elems, bits);
    for (EncSlot& s : p->slots) {
      if (s.h_in.reserve(nbytes + 64)) return OJPHGPU_E_NOMEM;
      if (!s.pixels.p && s.pixels.alloc(nbytes + 64)) return OJPHGPU_E_NOMEM;
    }
    p->packed_bits = bits; p->in_bytes = nbytes;
    return OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_enc_pipe_submit(ojphgpu_enc_pipe* p)
{
  if (!p) return OJPHGPU_E_INVALID;
  const uint32_t si = (uint32_t)(p->n_acq % p->depth);
  EncSlot& s = p->slots[si];
  { std::lock_guard<std::mutex> lk(p->mu); if (s.state != ACQUIRED) return OJPHGPU_E_INVALID; }
  HIPCHK(hipSetDevice(p->device));
  ojphgpu_encoder* e = p->enc;
  s.rc = 0; s.cs_len = 0; s.t_submit = now_ms();
  { const int r0 = upload(p->mode, p->s_h2d, (p->pixel_bits || p->packed_bits) ? s.pixels.p : s.image.p, s.h_in, 0, p->in_bytes); if (r0) return r0; }
  HIPCHK(hipEventRecord(s.ev_in, p->s_h2d));
  HIPCHK(hipStreamWaitEvent(p->s_comp, s.ev_in, 0));
  if (p->pixel_bits) {                               // the file's / capture buffer's bytes -> planes, on the device
    const Plan& P = *p->P;
    const int r0 = ojphgpu_unpack_pixels(p->s_comp, s.pixels.p, s.image.p, P.p.width, P.p.height, P.p.num_comps, p->pixel_bits,
                                         p->big_endian, p->container);
    if (r0) return r0;
  }
  if (p->packed_bits) {
    const int r0 = ojphgpu_unpack_bits(p->s_comp, s.pixels.p, s.image.p, p->P->frame_elems, p->packed_bits, p->container);
    if (r0) return r0;
  }
  // the block coder writes its per-block {offset, length} records straight into the slot's pinned memory
  // (8 bytes per block, posted PCIe writes): all the host needs to code the packet headers
  const size_t nb = e->block_ids.size();
  e->o_out = s.out.p; e->o_results = s.h_res.d; e->o_counters = s.counters.p;
  int rc = ojphgpu_encoder_run_container(e, s.image.p, p->container);
  if (rc) return rc;
  rc = publish_words_launch(p->s_comp, (uint32_t*)(s.h_res.d + nb * sizeof(ojphgpu_cb_result)), (const uint32_t*)s.counters.p, 2);
  if (rc) return rc;
  HIPCHK(hipEventRecord(s.ev_kern, p->s_comp));
  {
    std::lock_guard<std::mutex> lk(p->mu);
    s.state = SUBMITTED;
    p->work.push_back(si);
    p->n_acq++; p->n_sub++;
  }
  p->cv_work.notify_one();
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_enc_pipe_collect(ojphgpu_enc_pipe* p, const uint8_t** h_codestream, size_t* len)
{
  if (!p || !h_codestream || !len) return OJPHGPU_E_INVALID;
  std::unique_lock<std::mutex> lk(p->mu);
  if (p->n_col >= p->n_sub) return OJPHGPU_E_INVALID;            // nothing in flight
  if (p->n_col > 0) {                                              // the codestream handed out last time is released now
    EncSlot& prev = p->slots[(p->n_col - 1) % p->depth];
    if (prev.state == HELD) prev.state = FREE;
  }
  EncSlot& s = p->slots[p->n_col % p->depth];
  p->cv_done.wait(lk, [&] { return s.state == DONE; });
  p->n_col++;
  if (s.rc) { s.state = FREE; return s.rc; }
  s.state = HELD;
  *h_codestream = s.h_cs.p; *len = s.cs_len;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_enc_pipe_stats(ojphgpu_enc_pipe* p, double out[4])
{
  if (!p || !out) return OJPHGPU_E_INVALID;
  std::lock_guard<std::mutex> lk(p->mu);
  out[0] = (double)p->n_done;
  out[1] = p->n_done ? p->sum_t2 / (double)p->n_done : 0.0;          // host Tier-2 (layout) per frame, ms
  out[2] = p->n_done ? p->sum_latency / (double)p->n_done : 0.0;     // submit -> codestream in pinned memory, ms
  out[3] = (double)pool_threads();
  return OJPHGPU_OK;
}

// =============================================================================================
// decoder pipe
// =============================================================================================
struct DecSlot {
  SlotState state = FREE;
  Pinned h_cs, h_descs, h_img, h_status;
  Grow data;
  DeviceBuf image, cb_descs, status, pixels;
  hipEvent_t ev_in = nullptr, ev_kern = nullptr, ev_done = nullptr;
  size_t cs_len = 0;
  int rc = 0; uint32_t failed = 0;
  double t_submit = 0, t_done = 0, t_parse = 0;
};

struct ojphgpu_dec_pipe {
  ojphgpu_plan* first = nullptr;                     // parsed from the first codestream: the geometry of every frame
  const Plan* P = nullptr;
  int device = 0, container = 16, resilient = 0;
  int mode = copy_mode("OJPHGPU_DEC_COPY_MODE", 0);
  uint32_t depth = 0;
  // Consecutive frames go to different decoder objects (own scratch, own compute stream): step 1 of the block decoder
  // is a 0.2 ms chain that leaves most of the chip idle whatever the frame size, so for frames below ~8K one compute
  // stream running frame after frame sets the pace (4K 8-bit: 0.82 ms per frame against 0.44 ms of PCIe); with two
  // objects frame n+1's chains run beside frame n's step 2 and synthesis.  The objects run without their side
  // stream (streams share 4 hardware queues; the frames overlap each other instead).
  static constexpr uint32_t MAX_OBJECTS = 4;
  uint32_t nobj = 0;
  ojphgpu_decoder* decs[MAX_OBJECTS] = {};
  hipStream_t s_comps[MAX_OBJECTS] = {};
  std::mutex enqueue_mus[MAX_OBJECTS];
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;
  std::vector<DecSlot> slots;
  size_t frame_bytes = 0;
  int pixel_bits = 0, big_endian = 0;               // != 0: frames come back pixel-interleaved (ojphgpu_dec_pipe_set_pixels)
  int packed_bits = 0;                              // != 0: ... bit-packed (ojphgpu_dec_pipe_set_packed)
  size_t out_bytes = 0;
  uint64_t n_acq = 0, n_sub = 0, n_col = 0;
  std::mutex mu; std::condition_variable cv_work, cv_done;
  std::deque<uint32_t> work; bool stop = false;
  std::vector<std::thread> workers;
  double sum_parse = 0, sum_latency = 0; uint64_t n_done = 0;
};

static void dec_process_frame(ojphgpu_dec_pipe* p, DecSlot& s)
{
  const Plan& P = *p->P;
  const uint32_t k = (uint32_t)(&s - p->slots.data()) % p->nobj;
  ojphgpu_decoder* d = p->decs[k];
  hipStream_t s_comp = p->s_comps[k];
  auto fail = [&](int rc) { s.rc = rc; };
  if (hipSetDevice(p->device) != hipSuccess) return fail(OJPHGPU_E_HIP);
  const double t0 = now_ms();
  ojphgpu_plan* q = nullptr;
  int rc = ojphgpu_t2_parse(s.h_cs.p, s.cs_len, p->resilient, &q);
  if (rc) return fail(rc);
  struct Hold { ojphgpu_plan* q; ~Hold() { ojphgpu_plan_destroy(q); } } hold{ q };
  rc = no_throw([&]() -> int {
    const Plan& Q = q->plan;
    int r2 = ojphgpu_same_frame_geometry(P, Q, true);
    if (r2) return r2;
    const size_t nb = d->block_ids.size();
    ojphgpu_cb_desc* bd = (ojphgpu_cb_desc*)s.h_descs.p;
    DecFrameInfo fi;
    ojphgpu_decoder_fill_descs(P, Q, d->block_ids, 0, 0, bd, fi);
    uint64_t nquads = 0, naux = 0;
    if (ojphgpu_ht_decode_layout(bd, (uint32_t)nb, &nquads, &naux) != OJPHGPU_OK) return OJPHGPU_E_INVALID;
    if ((naux + 16) * 4 > d->aux.n || (nquads + 16) * 4 > d->quads.n) return OJPHGPU_E_INVALID;     // sized for the worst case at create
    s.t_parse = now_ms() - t0;
    if (fi.first + fi.len > s.cs_len) return OJPHGPU_E_CODESTREAM;
    if (s.data.reserve((size_t)fi.data_bytes() + (size_t)fi.len / 4 + 128)) return OJPHGPU_E_NOMEM;
    if ((r2 = upload(p->mode, p->s_h2d, s.data.b.p, s.h_cs, (size_t)fi.first, (size_t)fi.len)) != 0) return r2;
    if ((r2 = ojphgpu_decoder_upload_pads(p->s_h2d, (uint8_t*)s.data.b.p, s.h_cs.p, s.cs_len, fi.pads)) != 0) return r2;   // (damaged codestreams only)
    if ((r2 = upload(p->mode, p->s_h2d, s.cb_descs.p, s.h_descs, 0, nb * sizeof(ojphgpu_cb_desc))) != 0) return r2;
    HIPCHK(hipEventRecord(s.ev_in, p->s_h2d));
    // kernels of the frame on the object's compute stream, then the downloads; `separate`: the repeat a fused launch asked for
    const size_t st_bytes = ((nb + 3) & ~(size_t)3) + 4;    // the status bytes + the RETRY word behind them
    uint32_t epoch = 0; bool was_fused = false;
    auto run_and_fetch = [&](bool separate) -> int {
      int r3;
      {
        // a decoder object is shared by the frames in flight on it: what a run reads is set and enqueued under a lock
        std::lock_guard<std::mutex> lk(p->enqueue_mus[k]);
        HIPCHK(hipStreamWaitEvent(s_comp, s.ev_in, 0));
        d->o_cb_descs = s.cb_descs.p; d->o_data = s.data.b.p; d->o_status = s.status.p;
        d->any_refine = fi.any_refine; d->kinds = fi.kinds; d->max_len1 = fi.max_len1;
        d->force_separate = separate;
        r3 = ojphgpu_decoder_run_container(d, s.image.p, p->container);
        d->force_separate = false;
        if (r3) return r3;
        epoch = d->fused_epoch; was_fused = d->last_fused;
        if (separate) d->fused_retries++;
        if (p->pixel_bits) {                           // planes -> the pixel order of the file / display buffer
          uint32_t depth = 0;
          for (const CompGeo& g : P.comps) depth = std::max(depth, g.bit_depth);
          r3 = ojphgpu_pack_pixels(s_comp, s.image.p, s.pixels.p, P.p.width, P.p.height, P.p.num_comps, p->container, p->pixel_bits,
                                   p->big_endian, depth);
          if (r3) return r3;
        }
        if (p->packed_bits && (r3 = ojphgpu_pack_bits(s_comp, s.image.p, s.pixels.p, P.frame_elems, p->container, p->packed_bits)) != 0) return r3;
        HIPCHK(hipEventRecord(s.ev_kern, s_comp));
      }
      HIPCHK(hipStreamWaitEvent(p->s_d2h, s.ev_kern, 0));
      if ((r3 = download(p->mode, p->s_d2h, s.h_img, (p->pixel_bits || p->packed_bits) ? s.pixels.p : s.image.p, p->out_bytes)) != 0) return r3;       // beside the next frame's upload
      if ((r3 = download(p->mode, p->s_d2h, s.h_status, s.status.p, st_bytes)) != 0) return r3;
      HIPCHK(hipEventRecord(s.ev_done, p->s_d2h));
      HIPCHK(hipEventSynchronize(s.ev_done));
      return OJPHGPU_OK;
    };
    if ((r2 = run_and_fetch(false)) != 0) return r2;
    // the fused block-decoder launch of this run gave up waiting (the chip was held up for seconds by other work) and marked
    // the run instead of failing blocks: once more through the separate launches (see ojphgpu_decoder_failed_blocks)
    if (was_fused) {
      const bool again = ojphgpu_fused_retry_wanted(s.h_status.p, (uint32_t)nb, epoch);
      { std::lock_guard<std::mutex> lk(p->enqueue_mus[k]); d->fused_outcome(again); }
      if (again && (r2 = run_and_fetch(true)) != 0) return r2;
    }
    uint32_t failed = 0;
    for (size_t i = 0; i < nb; ++i) failed += s.h_status.p[i] != 0;
    s.failed = failed;
    return OJPHGPU_OK;
  });
  if (rc) {
    // the slot goes DONE -> FREE next: whatever was enqueued for it must have finished before _acquire rewrites or
    // re-reserves its buffers, and the shared decoder object must not keep pointing at them
    hipStreamSynchronize(p->s_h2d); hipStreamSynchronize(s_comp); hipStreamSynchronize(p->s_d2h);
    std::lock_guard<std::mutex> lk(p->enqueue_mus[k]);
    if (d->o_cb_descs == s.cb_descs.p) { d->o_cb_descs = nullptr; d->o_data = nullptr; d->o_status = nullptr; }
    fail(rc);
  }
}

static void dec_worker(ojphgpu_dec_pipe* p)
{
  for (;;) {
    uint32_t si;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_work.wait(lk, [&] { return p->stop || !p->work.empty(); });
      if (p->work.empty()) return;
      si = p->work.front(); p->work.pop_front();
    }
    DecSlot& s = p->slots[si];
    dec_process_frame(p, s);
    {
      std::lock_guard<std::mutex> lk(p->mu);
      s.t_done = now_ms();
      s.state = DONE;
      p->sum_parse += s.t_parse; p->sum_latency += s.t_done - s.t_submit; p->n_done++;
    }
    p->cv_done.notify_all();
  }
}

extern "C" void ojphgpu_dec_pipe_destroy(ojphgpu_dec_pipe* p)
{
  if (!p) return;
  (void)hipSetDevice(p->device);
  { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
  p->cv_work.notify_all();
  for (std::thread& t : p->workers) t.join();
  for (hipStream_t s : { p->s_h2d, p->s_comps[0], p->s_comps[1], p->s_comps[2], p->s_comps[3], p->s_d2h }) if (s) (void)hipStreamSynchronize(s);
  for (ojphgpu_decoder* d : p->decs) if (d) ojphgpu_decoder_destroy(d);
  for (DecSlot& s : p->slots) {
    s.h_cs.release(); s.h_descs.release(); s.h_img.release(); s.h_status.release();
    for (DeviceBuf* b : { &s.data.b, &s.image, &s.cb_descs, &s.status, &s.pixels }) b->release();
    for (hipEvent_t ev : { s.ev_in, s.ev_kern, s.ev_done }) if (ev) (void)hipEventDestroy(ev);
  }
  for (hipStream_t s : { p->s_h2d, p->s_comps[0], p->s_comps[1], p->s_comps[2], p->s_comps[3], p->s_d2h }) if (s) (void)hipStreamDestroy(s);
  if (p->first) ojphgpu_plan_destroy(p->first);
  delete p;
}

extern "C" int ojphgpu_dec_pipe_create(const uint8_t* h_codestream, size_t len, int resilient, int device, uint32_t depth,
                                        int container_bits, uint32_t host_threads, ojphgpu_dec_pipe** out)
{
  if (!h_codestream || !out || depth < 2 || depth > 16 || (container_bits != 8 && container_bits != 16 && container_bits != 32)) return OJPHGPU_E_INVALID;
  *out = nullptr;
  return no_throw([&]() -> int {
    HIPCHK(hipSetDevice(device));
    ojphgpu_dec_pipe* p = new (std::nothrow) ojphgpu_dec_pipe();
    if (!p) return OJPHGPU_E_NOMEM;
    struct Owner { ojphgpu_dec_pipe* p; ~Owner() { if (p) ojphgpu_dec_pipe_destroy(p); } } owner{ p };
    int rc = ojphgpu_t2_parse(h_codestream, len, resilient, &p->first);
    if (rc) return rc;
    const Plan& P = p->first->plan;
    p->P = &P; p->device = device; p->container = container_bits; p->depth = depth; p->resilient = resilient;
    if (container_bits != 32) for (const CompGeo& g : P.comps) if (g.bit_depth > (uint32_t)container_bits) return OJPHGPU_E_INVALID;
    HIPCHK(hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking));
    HIPCHK(create_copy_out_stream(&p->s_d2h));
    {
      const char* e = getenv("OJPHGPU_DEC_PIPE_OBJECTS"); const long v = e ? atol(e) : 0;
      p->nobj = v >= 1 && v <= (long)ojphgpu_dec_pipe::MAX_OBJECTS ? (uint32_t)v : 2u;
      p->nobj = std::min(p->nobj, depth);
    }
    for (uint32_t k = 0; k < p->nobj; ++k) {
      HIPCHK(hipStreamCreateWithFlags(&p->s_comps[k], hipStreamNonBlocking));
      rc = ojphgpu_decoder_create(p->first, device, p->s_comps[k], &p->decs[k]);
      if (rc) return rc;
      ojphgpu_decoder* dk = p->decs[k];
      (void)ojphgpu_decoder_set_timing(dk, 0);
      if (p->nobj > 1 && dk->side) {                     // no fork inside a frame: the frames overlap each other
        (void)hipStreamDestroy(dk->side); dk->side = nullptr; dk->n_low = 0;
      }
      // the flat VLC / MEL strings of any later frame fit: the worst case of every block (Lcup <= 4079 + slack)
      uint64_t worst = (uint64_t)dk->block_ids.size() * ojphgpu_ht_decode_aux_words(4079) + 64;
      if (P.any_wide)                                      // 64-bit sample path: the flat MagSgn strings, bounded by what a block of that size can code
        for (uint32_t id : dk->block_ids) {
          const Block& k = P.blocks[id]; const Band& B = P.bands[k.band];
          if (is_wide(P, B.comp)) worst += ojphgpu::ht_decode64_extra_aux_words(block_scratch_bytes(k.r.w, k.r.h, 62));
        }
      dk->aux.release();
      if (dk->aux.alloc((size_t)worst * 4 + 64)) return OJPHGPU_E_NOMEM;
    }
    ojphgpu_decoder* d = p->decs[0];
    const size_t nb = d->block_ids.size();
    p->frame_bytes = (size_t)P.frame_elems * (size_t)(container_bits / 8);
    p->out_bytes = p->frame_bytes;
    p->slots.resize(depth);
    for (DecSlot& s : p->slots) {
      if (s.h_cs.reserve(len + len / 4 + (1u << 16)) || s.h_descs.reserve(nb * sizeof(ojphgpu_cb_desc) + 64) ||
          s.h_img.reserve(p->frame_bytes + 64) || s.h_status.reserve(nb + 64)) return OJPHGPU_E_NOMEM;
      if (s.data.reserve(len + len / 4 + 64) || s.image.alloc((size_t)P.frame_elems * 4 + 64) || s.cb_descs.alloc(nb * sizeof(ojphgpu_cb_desc) + 64) ||
          s.status.alloc(nb + 64)) return OJPHGPU_E_NOMEM;
      HIPCHK(hipMemset(s.status.p, 0, nb + 64));
      for (hipEvent_t* ev : { &s.ev_in, &s.ev_kern, &s.ev_done }) HIPCHK(hipEventCreateWithFlags(ev, hipEventDisableTiming | hipEventReleaseToSystem));
    }
    const uint32_t nthreads = host_threads ? std::min<uint32_t>(host_threads, 16) : 2;
    for (uint32_t i = 0; i < nthreads; ++i) p->workers.emplace_back(dec_worker, p);
    owner.p = nullptr;
    *out = p;
    return OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_dec_pipe_acquire(ojphgpu_dec_pipe* p, size_t len, uint8_t** h_codestream)
{
  if (!p || !h_codestream || len == 0) return OJPHGPU_E_INVALID;
  DecSlot& s = p->slots[p->n_acq % p->depth];
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (s.state != FREE && s.state != ACQUIRED) return OJPHGPU_E_AGAIN;     // every slot is in flight: collect a frame first
    s.state = ACQUIRED;
  }
  if (hipSetDevice(p->device) != hipSuccess) return OJPHGPU_E_HIP;
  if (s.h_cs.reserve(len + len / 4 + 64)) { std::lock_guard<std::mutex> lk(p->mu); s.state = FREE; return OJPHGPU_E_NOMEM; }
  s.cs_len = len;
  *h_codestream = s.h_cs.p;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_dec_pipe_submit(ojphgpu_dec_pipe* p)
{
  if (!p) return OJPHGPU_E_INVALID;
  const uint32_t si = (uint32_t)(p->n_acq % p->depth);
  DecSlot& s = p->slots[si];
  std::lock_guard<std::mutex> lk(p->mu);
  if (s.state != ACQUIRED) return OJPHGPU_E_INVALID;
  s.rc = 0; s.failed = 0; s.t_submit = now_ms();
  s.state = SUBMITTED;
  p->work.push_back(si);
  p->n_acq++; p->n_sub++;
  p->cv_work.notify_one();
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_dec_pipe_collect(ojphgpu_dec_pipe* p, const void** h_frame, size_t* bytes, uint32_t* failed_blocks)
{
  if (!p || !h_frame) return OJPHGPU_E_INVALID;
  std::unique_lock<std::mutex> lk(p->mu);
  if (p->n_col >= p->n_sub) return OJPHGPU_E_INVALID;
  if (p->n_col > 0) {
    DecSlot& prev = p->slots[(p->n_col - 1) % p->depth];
    if (prev.state == HELD) prev.state = FREE;
  }
  DecSlot& s = p->slots[p->n_col % p->depth];
  p->cv_done.wait(lk, [&] { return s.state == DONE; });
  p->n_col++;
  if (s.rc) { s.state = FREE; return s.rc; }
  s.state = HELD;
  *h_frame = s.h_img.p;
  if (bytes) *bytes = p->out_bytes;
  if (failed_blocks) *failed_blocks = s.failed;
  return (s.failed && !p->resilient) ? OJPHGPU_E_BLOCK : OJPHGPU_OK;
}

extern "C" int ojphgpu_dec_pipe_set_pixels(ojphgpu_dec_pipe* p, int pixel_bits, int big_endian)
{
  if (!p || p->n_sub != 0 || p->packed_bits) return OJPHGPU_E_INVALID;
  return no_throw([&]() -> int {
    if (pixel_bits == 0) { p->pixel_bits = 0; p->out_bytes = p->frame_bytes; return OJPHGPU_OK; }
    const Plan& P = *p->P;
    const int rc = pixels_fit(P, pixel_bits, p->container);
    if (rc) return rc;
    for (const CompGeo& g : P.comps) if (g.bit_depth != P.comps[0].bit_depth) return OJPHGPU_E_INVALID;   // one clamp range per frame
    HIPCHK(hipSetDevice(p->device));
    const size_t nbytes = (size_t)P.frame_elems * (size_t)(pixel_bits / 8);
    for (DecSlot& s : p->slots) {
      if (s.h_img.reserve(nbytes + 64)) return OJPHGPU_E_NOMEM;
      if (!s.pixels.p && s.pixels.alloc(nbytes + 64)) return OJPHGPU_E_NOMEM;
    }
    p->pixel_bits = pixel_bits; p->big_endian = big_endian ? 1 : 0; p->out_bytes = nbytes;
    return OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_dec_pipe_set_packed(ojphgpu_dec_pipe* p, int bits)
{
  if (!p || p->n_sub != 0 || p->pixel_bits) return OJPHGPU_E_INVALID;
  return no_throw([&]() -> int {
    if (bits == 0) { p->packed_bits = 0; p->out_bytes = p->frame_bytes; return OJPHGPU_OK; }
    const Plan& P = *p->P;
    const int rc = packed_fit(P, bits, p->container);
    if (rc) return rc;
    HIPCHK(hipSetDevice(p->device));
    const size_t nbytes = packed_bytes(P.frame_elems, bits);
    for (DecSlot& s : p->slots) {
      if (s.h_img.reserve(nbytes + 64)) return OJPHGPU_E_NOMEM;
      if (!s.pixels.p && s.pixels.alloc(nbytes + 64)) return OJPHGPU_E_NOMEM;
    }
    p->packed_bits = bits; p->out_bytes = nbytes;
    return OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_dec_pipe_plan(ojphgpu_dec_pipe* p, const ojphgpu_plan** plan)
{
  if (!p || !plan) return OJPHGPU_E_INVALID;
  *plan = p->first;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_dec_pipe_fused_retries(ojphgpu_dec_pipe* p, uint32_t* count)
{
  if (!p || !count) return OJPHGPU_E_INVALID;
  uint32_t n = 0;
  for (uint32_t k = 0; k < p->nobj; ++k) {
    std::lock_guard<std::mutex> lk(p->enqueue_mus[k]);
    n += p->decs[k]->fused_retries;
  }
  *count = n;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_dec_pipe_stats(ojphgpu_dec_pipe* p, double out[4])
{
  if (!p || !out) return OJPHGPU_E_INVALID;
  std::lock_guard<std::mutex> lk(p->mu);
  out[0] = (double)p->n_done;
  out[1] = p->n_done ? p->sum_parse / (double)p->n_done : 0.0;       // host parse + descriptor fill per frame, ms
  out[2] = p->n_done ? p->sum_latency / (double)p->n_done : 0.0;     // submit -> frame in pinned memory, ms
  out[3] = (double)p->workers.size();
  return OJPHGPU_OK;
}
