// openjph_amd/csrc/ojphgpu_multi.cpp -- ONE frame coded by SEVERAL GPUs of the node, from one process (C ABI section 8).
//
// Tiles are independent in JPEG 2000 -- own transform, quantisation, code-blocks, packets and tile-parts (reference:
// local::codestream::pre_alloc / finalize_alloc give every tile its own object tree, ojph_codestream_local.cpp:113-180, and
// codestream::flush writes them one after the other, tile::flush ojph_tile.cpp:584-610) -- so a tiled frame shards by
// contiguous runs of tiles with no exchange between the devices while they code.  What the reference does in one thread
// over all tiles is done here by one host thread + one codec object per device over its run of tiles:
//
//   encode   every worker: its tiles' rectangles of the frame host -> its GPU, the kernels (ojphgpu_encoder_create_tiles),
//            the tile-parts laid out in HBM (ojphgpu_encoder_finish_tiles_device: only block lengths visit the host);
//            then -- the one point where the devices meet -- the caller's thread makes the main header from everybody's
//            Psot lengths and hands every worker the offset of its run (a prefix sum), and every worker copies its
//            tile-parts from its GPU straight to THAT place of the caller's output buffer.  No device-to-device traffic,
//            no staging copy, no collective: SURVEY.md section 8(e), second option.
//   decode   every worker parses nothing (the codestream was parsed once), uploads the byte range its tiles' blocks live in,
//            runs the kernels and copies its tiles' rectangles of the frame to the caller's image.
//
// The caller's buffers should be pinned (hipHostMalloc / hipHostRegister) for the copies to run at link speed; pageable
// memory works.  The same device may be named more than once (two workers then share it: how the one-GPU test box
// exercises this path).  The multi-process form of the same sharding (one process per GPU, torch.distributed / RCCL for
// the final gather) is openjph_amd/shard.py.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "ojph_plan.h"

using namespace ojphgpu;

namespace {

#define HIPCHK(x) do { if ((x) != hipSuccess) return OJPHGPU_E_HIP; } while (0)

struct Range { uint32_t first, count; };

// contiguous, balanced runs: the first num % n workers take one tile more (openjph_amd/shard.py tile_range)
Range run_of(uint32_t num_tiles, uint32_t k, uint32_t n)
{
  const uint32_t base = num_tiles / n, extra = num_tiles % n;
  return Range{ k * base + (k < extra ? k : extra), base + (k < extra ? 1u : 0u) };
}

// the rectangle of tile t's component c inside the component's plane of the frame (reconstructed resolution)
struct Piece { uint64_t off; uint32_t pitch, w, h; };
Piece piece_of(const Plan& P, uint32_t t, uint32_t c)
{
  const TileComp& tc = P.tcomps[P.tiles[t].comps[c]];
  const Resolution& R = P.ress[tc.res[P.recon_decomps(c)]];
  const CompGeo& g = P.comps[c];
  return Piece{ g.frame_off + (uint64_t)(R.r.y0 - g.y0) * g.w + (R.r.x0 - g.x0), g.w, R.r.w, R.r.h };
}

// The rectangles of component c that a run of tiles covers, merged: tiles of one tile row side by side into one rectangle,
// rectangles of equal extent below each other into one -- a run of whole tile rows is ONE rectangle (and, spanning the
// plane's width, one contiguous copy) where tile by tile it was thousands of 4 KB rows.
struct Rect2 { uint64_t off; uint32_t x, y, w, h; };
std::vector<Rect2> rects_of(const Plan& P, Range tiles, uint32_t c)
{
  std::vector<Rect2> r;
  const CompGeo& g = P.comps[c];
  for (uint32_t t = tiles.first; t < tiles.first + tiles.count; ++t) {
    const Piece q = piece_of(P, t, c);
    if (q.w == 0 || q.h == 0) continue;
    const uint64_t rel = q.off - g.frame_off;
    Rect2 n{ q.off, (uint32_t)(rel % g.w), (uint32_t)(rel / g.w), q.w, q.h };
    if (!r.empty() && r.back().y == n.y && r.back().h == n.h && r.back().x + r.back().w == n.x) r.back().w += n.w;   // next tile of the row
    else r.push_back(n);
  }
  std::vector<Rect2> m;
  for (const Rect2& n : r) {
    if (!m.empty() && m.back().x == n.x && m.back().w == n.w && m.back().y + m.back().h == n.y) m.back().h += n.h;    // next tile row
    else m.push_back(n);
  }
  return m;
}

// host <-> device copy of those rectangles (esz bytes per sample); to_device: host -> device
int copy_rects(const Plan& P, Range tiles, uint32_t nc, uint8_t* dev, uint8_t* host, uint32_t esz, bool to_device, hipStream_t s)
{
  for (uint32_t c = 0; c < nc; ++c) {
    const size_t pitch = (size_t)P.comps[c].w * esz;
    for (const Rect2& q : rects_of(P, tiles, c)) {
      uint8_t* d = dev + q.off * esz; uint8_t* h = host + q.off * esz;
      const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
      if ((size_t)q.w * esz == pitch) HIPCHK(hipMemcpyAsync(to_device ? d : h, to_device ? h : d, pitch * q.h, kind, s));
      else HIPCHK(hipMemcpy2DAsync(to_device ? d : h, pitch, to_device ? h : d, pitch, (size_t)q.w * esz, q.h, kind, s));
    }
  }
  return OJPHGPU_OK;
}

template <typename F>
int run_workers(size_t n, F f)
{
  std::vector<int> rc(n, 0);
  std::vector<std::thread> th;
  // (a thread that cannot be created must not take the process down across the C ABI: what was started is joined, the
  // call fails)
  bool spawned = true;
  try {
    th.reserve(n);
    for (size_t k = 1; k < n; ++k) th.emplace_back([&, k] { rc[k] = no_throw([&] { return f(k); }); });
  } catch (...) { spawned = false; }
  if (spawned) rc[0] = no_throw([&] { return f(0); });
  for (std::thread& t : th) t.join();
  if (!spawned) return OJPHGPU_E_NOMEM;
  for (int r : rc) if (r) return r;
  return OJPHGPU_OK;
}

}  // namespace

struct ojphgpu_multi_encoder {
  const ojphgpu_plan* plan = nullptr;
  struct Worker {
    int device = 0; Range tiles{ 0, 0 };
    hipStream_t stream = nullptr;
    ojphgpu_encoder* enc = nullptr;
    void* d_image = nullptr; void* d_out = nullptr; size_t out_cap = 0;
    size_t len = 0;                                   // bytes of this run's tile-parts (last encode)
  };
  std::vector<Worker> w;
  std::vector<uint32_t> psot;                         // Psot of every tile-part of the frame (last encode)
};

extern "C" void ojphgpu_multi_encoder_destroy(ojphgpu_multi_encoder* m)
{
  if (!m) return;
  for (auto& k : m->w) {
    (void)hipSetDevice(k.device);
    if (k.stream) (void)hipStreamSynchronize(k.stream);
    if (k.enc) ojphgpu_encoder_destroy(k.enc);
    if (k.d_image) (void)hipFree(k.d_image);
    if (k.d_out) (void)hipFree(k.d_out);
    if (k.stream) (void)hipStreamDestroy(k.stream);
  }
  delete m;
}

extern "C" int ojphgpu_multi_encoder_create(const ojphgpu_plan* plan, const int* devices, uint32_t num_devices,
                                             ojphgpu_multi_encoder** out)
{
  if (!plan || !devices || !out || num_devices == 0 || num_devices > 64) return OJPHGPU_E_INVALID;
  *out = nullptr;
  const Plan& P = plan->plan;
  const uint32_t nt = (uint32_t)P.tiles.size();
  return no_throw([&]() -> int {
    ojphgpu_multi_encoder* m = new (std::nothrow) ojphgpu_multi_encoder();
    if (!m) return OJPHGPU_E_NOMEM;
    struct Owner { ojphgpu_multi_encoder* p; ~Owner() { if (p) ojphgpu_multi_encoder_destroy(p); } } owner{ m };
    m->plan = plan;
    const uint32_t n = num_devices < nt ? num_devices : nt;              // a single-tile frame does not shard: one worker
    m->w.resize(n);
    m->psot.assign((size_t)nt * P.parts_per_tile, 0);
    for (uint32_t k = 0; k < n; ++k) {
      auto& W = m->w[k];
      W.device = devices[k]; W.tiles = run_of(nt, k, n);
      HIPCHK(hipSetDevice(W.device));
      HIPCHK(hipStreamCreateWithFlags(&W.stream, hipStreamNonBlocking));
      int rc = ojphgpu_encoder_create_tiles(plan, W.device, W.stream, W.tiles.first, W.tiles.count, &W.enc);
      if (rc) return rc;
      HIPCHK(hipMalloc(&W.d_image, (size_t)P.frame_elems * 4 + 64));
      // the run's tile-parts: what its blocks can code at most, plus markers and packet headers
      uint64_t bound = 1u << 16;
      for (const Block& b : P.blocks) {
        const Band& B = P.bands[b.band];
        if (B.tile >= W.tiles.first && B.tile - W.tiles.first < W.tiles.count) bound += block_scratch_bytes(b.r.w, b.r.h, B.K_max) + 8;
      }
      bound += (uint64_t)W.tiles.count * P.parts_per_tile * 16;
      for (uint32_t t = W.tiles.first; t < W.tiles.first + W.tiles.count; ++t) bound += (uint64_t)P.tiles[t].packets.size() * 8;   // (an empty packet is a byte, SOP / EPH six more)
      W.out_cap = (size_t)bound;
      HIPCHK(hipMalloc(&W.d_out, W.out_cap));
    }
    owner.p = nullptr;
    *out = m;
    return OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_multi_encode(ojphgpu_multi_encoder* m, const int32_t* h_image, uint8_t* h_out, size_t cap, size_t* out_len)
{
  return ojphgpu_multi_encode_container(m, h_image, 32, h_out, cap, out_len);
}

extern "C" int ojphgpu_multi_encode_container(ojphgpu_multi_encoder* m, const void* h_image, int container_bits, uint8_t* h_out, size_t cap,
                                               size_t* out_len)
{
  if (!m || !h_image || !out_len || (container_bits != 32 && container_bits != 16 && container_bits != 8)) return OJPHGPU_E_INVALID;
  const Plan& P = m->plan->plan;
  const uint32_t esz = (uint32_t)container_bits / 8u;
  const uint32_t ppt = P.parts_per_tile, nc = P.p.num_comps;
  // 1. every device: its tiles in, kernels, tile-parts laid out in its HBM
  int rc = run_workers(m->w.size(), [&](size_t k) -> int {
    auto& W = m->w[k];
    HIPCHK(hipSetDevice(W.device));
    int r = copy_rects(P, W.tiles, nc, (uint8_t*)W.d_image, (uint8_t*)const_cast<void*>(h_image), esz, true, W.stream);
    if (r) return r;
    r = container_bits == 16 ? ojphgpu_encoder_run_device16(W.enc, (const uint16_t*)W.d_image)
      : container_bits == 8 ? ojphgpu_encoder_run_device8(W.enc, (const uint8_t*)W.d_image) : ojphgpu_encoder_run_device(W.enc, (const int32_t*)W.d_image);
    if (r) return r;
    r = ojphgpu_encoder_finish_tiles_device(W.enc, (uint8_t*)W.d_out, W.out_cap, &W.len, m->psot.data() + (size_t)W.tiles.first * ppt);
    if (r == OJPHGPU_E_OVERFLOW && W.len > W.out_cap) {
      // the bound of _create was short (it is an estimate; E_OVERFLOW of THIS entry point means "the caller's buffer", which
      // a worker's staging area is not): W.len holds what the run needs -- a larger area, and the assembly once more
      void* bigger = nullptr;
      if (hipMalloc(&bigger, W.len + 64) != hipSuccess) { (void)hipGetLastError(); return OJPHGPU_E_NOMEM; }
      (void)hipFree(W.d_out); W.d_out = bigger; W.out_cap = W.len + 64;
      r = ojphgpu_encoder_finish_tiles_device(W.enc, (uint8_t*)W.d_out, W.out_cap, &W.len, m->psot.data() + (size_t)W.tiles.first * ppt);
    }
    return r == OJPHGPU_E_OVERFLOW ? OJPHGPU_E_INVALID : r;
  });
  if (rc) return rc;
  // 2. the one meeting point: main header from everybody's Psot, a prefix sum of the runs' lengths
  size_t hdr_len = 0;
  rc = ojphgpu_t2_write_main_header(m->plan, m->psot.data(), nullptr, 0, &hdr_len);
  if (rc != OJPHGPU_OK && rc != OJPHGPU_E_OVERFLOW) return rc;
  size_t total = hdr_len + 2;
  for (auto& W : m->w) total += W.len;
  *out_len = total;
  if (!h_out || cap < total) return OJPHGPU_E_OVERFLOW;
  rc = ojphgpu_t2_write_main_header(m->plan, m->psot.data(), h_out, cap, &hdr_len);
  if (rc) return rc;
  std::vector<size_t> at(m->w.size());
  size_t pos = hdr_len;
  for (size_t k = 0; k < m->w.size(); ++k) { at[k] = pos; pos += m->w[k].len; }
  h_out[pos] = 0xFF; h_out[pos + 1] = 0xD9;                              // EOC
  // 3. every device: its tile-parts straight to their place in the caller's buffer
  return run_workers(m->w.size(), [&](size_t k) -> int {
    auto& W = m->w[k];
    HIPCHK(hipSetDevice(W.device));
    if (W.len) HIPCHK(hipMemcpyAsync(h_out + at[k], W.d_out, W.len, hipMemcpyDeviceToHost, W.stream));
    HIPCHK(hipStreamSynchronize(W.stream));
    return OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_multi_encoder_workers(const ojphgpu_multi_encoder* m, uint32_t* num_workers, uint32_t* tiles_per_worker, uint32_t cap)
{
  if (!m || !num_workers) return OJPHGPU_E_INVALID;
  *num_workers = (uint32_t)m->w.size();
  if (tiles_per_worker) for (uint32_t k = 0; k < m->w.size() && k < cap; ++k) tiles_per_worker[k] = m->w[k].tiles.count;
  return OJPHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
struct ojphgpu_multi_decoder {
  ojphgpu_plan* plan = nullptr;                       // parsed from the codestream the decoder was made for
  int resilient = 0;
  struct Worker {
    int device = 0; Range tiles{ 0, 0 };
    hipStream_t stream = nullptr;
    ojphgpu_decoder* dec = nullptr;
    void* d_image = nullptr;
    uint32_t failed = 0;
  };
  std::vector<Worker> w;
};

extern "C" void ojphgpu_multi_decoder_destroy(ojphgpu_multi_decoder* m)
{
  if (!m) return;
  for (auto& k : m->w) {
    (void)hipSetDevice(k.device);
    if (k.stream) (void)hipStreamSynchronize(k.stream);
    if (k.dec) ojphgpu_decoder_destroy(k.dec);
    if (k.d_image) (void)hipFree(k.d_image);
    if (k.stream) (void)hipStreamDestroy(k.stream);
  }
  if (m->plan) ojphgpu_plan_destroy(m->plan);
  delete m;
}

extern "C" int ojphgpu_multi_decoder_create(const uint8_t* h_codestream, size_t len, int resilient, uint32_t skipped_res_for_data,
                                             uint32_t skipped_res_for_recon, const int* devices, uint32_t num_devices,
                                             ojphgpu_multi_decoder** out)
{
  if (!h_codestream || !devices || !out || num_devices == 0 || num_devices > 64) return OJPHGPU_E_INVALID;
  *out = nullptr;
  return no_throw([&]() -> int {
    ojphgpu_multi_decoder* m = new (std::nothrow) ojphgpu_multi_decoder();
    if (!m) return OJPHGPU_E_NOMEM;
    struct Owner { ojphgpu_multi_decoder* p; ~Owner() { if (p) ojphgpu_multi_decoder_destroy(p); } } owner{ m };
    m->resilient = resilient;
    int rc = ojphgpu_t2_parse(h_codestream, len, resilient, &m->plan);
    if (rc) return rc;
    if (skipped_res_for_data || skipped_res_for_recon) {
      rc = ojphgpu_plan_restrict_resolution(m->plan, skipped_res_for_data, skipped_res_for_recon);
      if (rc) return rc;
    }
    const Plan& P = m->plan->plan;
    const uint32_t nt = (uint32_t)P.tiles.size();
    const uint32_t n = num_devices < nt ? num_devices : nt;
    m->w.resize(n);
    for (uint32_t k = 0; k < n; ++k) {
      auto& W = m->w[k];
      W.device = devices[k]; W.tiles = run_of(nt, k, n);
      HIPCHK(hipSetDevice(W.device));
      HIPCHK(hipStreamCreateWithFlags(&W.stream, hipStreamNonBlocking));
      rc = ojphgpu_decoder_create_tiles(m->plan, W.device, W.stream, W.tiles.first, W.tiles.count, &W.dec);
      if (rc) return rc;
      HIPCHK(hipMalloc(&W.d_image, (size_t)P.frame_elems * 4 + 64));
    }
    owner.p = nullptr;
    *out = m;
    return OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_multi_decoder_plan(ojphgpu_multi_decoder* m, const ojphgpu_plan** plan)
{
  if (!m || !plan) return OJPHGPU_E_INVALID;
  *plan = m->plan;
  return OJPHGPU_OK;
}

// h_codestream: the codestream the decoder was created for (its block data is uploaded from here, each device its range)
extern "C" int ojphgpu_multi_decode(ojphgpu_multi_decoder* m, const uint8_t* h_codestream, size_t len, int32_t* h_image, uint32_t* failed_blocks)
{
  return ojphgpu_multi_decode_container(m, h_codestream, len, h_image, 32, failed_blocks);
}

extern "C" int ojphgpu_multi_decode_container(ojphgpu_multi_decoder* m, const uint8_t* h_codestream, size_t len, void* h_image, int container_bits,
                                               uint32_t* failed_blocks)
{
  if (!m || !h_codestream || !h_image || (container_bits != 32 && container_bits != 16 && container_bits != 8)) return OJPHGPU_E_INVALID;
  const Plan& P = m->plan->plan;
  const uint32_t esz = (uint32_t)container_bits / 8u;
  const uint32_t nc = P.p.num_comps;
  int rc = run_workers(m->w.size(), [&](size_t k) -> int {
    auto& W = m->w[k];
    HIPCHK(hipSetDevice(W.device));
    int r = ojphgpu_decoder_upload(W.dec, h_codestream, len);
    if (r) return r;
    r = container_bits == 16 ? ojphgpu_decoder_run_device16(W.dec, (uint16_t*)W.d_image)
      : container_bits == 8 ? ojphgpu_decoder_run_device8(W.dec, (uint8_t*)W.d_image) : ojphgpu_decoder_run_device(W.dec, (int32_t*)W.d_image);
    if (r) return r;
    r = ojphgpu_decoder_failed_blocks(W.dec, &W.failed);            // collects the run (see its comment): before the image is read
    if (r) return r;
    r = copy_rects(P, W.tiles, nc, (uint8_t*)W.d_image, (uint8_t*)h_image, esz, false, W.stream);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(W.stream));
    return OJPHGPU_OK;
  });
  if (rc) return rc;
  uint32_t failed = 0;
  for (auto& W : m->w) failed += W.failed;
  if (failed_blocks) *failed_blocks = failed;
  return (failed && !m->resilient) ? OJPHGPU_E_BLOCK : OJPHGPU_OK;
}

extern "C" int ojphgpu_host_register(void* h_ptr, size_t bytes)
{
  if (!h_ptr || !bytes) return OJPHGPU_E_INVALID;
  if (hipHostRegister(h_ptr, bytes, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return OJPHGPU_E_HIP; }
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_host_unregister(void* h_ptr)
{
  if (!h_ptr) return OJPHGPU_E_INVALID;
  if (hipHostUnregister(h_ptr) != hipSuccess) { (void)hipGetLastError(); return OJPHGPU_E_HIP; }
  return OJPHGPU_OK;
}
