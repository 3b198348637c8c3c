// openjph_amd/csrc/ojphgpu_codec.cpp -- whole-frame encoder / decoder objects of the C ABI.
//
// These are what an ojph::codestream-compatible facade calls where the reference runs its
// per-line object tree: ojphgpu_encoder_* replaces tile::push -> resolution::push_line ->
// subband::push_line -> codeblock::push/encode (ojph_tile.cpp:332, ojph_resolution.cpp:547,
// ojph_subband.cpp:292, ojph_codeblock.cpp:115-175) followed by codestream::flush
// (ojph_codestream_local.cpp:1148); ojphgpu_decoder_* replaces codestream::read
// (ojph_codestream_local.cpp:912) and the pull chain tile::pull -> resolution::pull_line ->
// subband::pull_line -> codeblock::decode/pull_line (ojph_tile.cpp:425, ojph_resolution.cpp:713,
// ojph_subband.cpp:336, ojph_codeblock.cpp:190-266).
//
// Layout in HBM (one arena of 32-bit elements per codec object, planned once per frame shape):
//   [ tile-component planes per resolution | sub-band planes ]   see ojph_plan.cpp (alloc)
// plus descriptor tables, the per-block scratch slots and the compacted code-block bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ht_tables.h"
#include "ojph_plan.h"

using namespace ojphgpu;

namespace ojphgpu {

int ensure_tables()
{
  static std::mutex mu;
  static bool done[64] = { false };
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::lock_guard<std::mutex> lock(mu);
  if (done[dev]) return 0;
  static HtTables tables;
  static bool built = false;
  if (!built) { build_ht_tables(tables); built = true; }
  if (upload_enc_tables(tables) != 0 || upload_dec_tables(tables) != 0) return -1;
  done[dev] = true;
  return 0;
}

}  // namespace ojphgpu

#include "ojphgpu_objects.h"

extern "C" void ojphgpu_encoder_destroy(ojphgpu_encoder* e)
{
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (DeviceBuf* b : { &e->arena, &e->image, &e->dwt_descs, &e->img_descs, &e->cb_descs, &e->conv_descs, &e->scratch, &e->out,
                        &e->results, &e->counters, &e->regions }) b->release();
  e->h_out.release(); e->h_res.release();
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  e->timer.destroy();
  delete e;
}

extern "C" int ojphgpu_encoder_create(const ojphgpu_plan* plan, int device, void* stream, ojphgpu_encoder** out)
{
  if (!plan) return OJPHGPU_E_INVALID;
  return ojphgpu_encoder_create_tiles(plan, device, stream, 0, (uint32_t)plan->plan.tiles.size(), out);
}

static int encoder_create(const ojphgpu_plan* plan, int device, void* stream, uint32_t tile_first, uint32_t tile_count,
                          uint32_t nframes, ojphgpu_encoder** out);

extern "C" int ojphgpu_encoder_create_tiles(const ojphgpu_plan* plan, int device, void* stream, uint32_t tile_first,
                                             uint32_t tile_count, ojphgpu_encoder** out)
{
  return no_throw([&] { return encoder_create(plan, device, stream, tile_first, tile_count, 1, out); });
}

extern "C" int ojphgpu_encoder_create_batch(const ojphgpu_plan* plan, int device, void* stream, uint32_t num_frames,
                                             ojphgpu_encoder** out)
{
  if (!plan) return OJPHGPU_E_INVALID;
  return no_throw([&] { return encoder_create(plan, device, stream, 0, (uint32_t)plan->plan.tiles.size(), num_frames, out); });
}

static int encoder_create(const ojphgpu_plan* plan, int device, void* stream, uint32_t tile_first, uint32_t tile_count,
                          uint32_t nframes, ojphgpu_encoder** out)
{
  if (!plan || !out || nframes == 0) return OJPHGPU_E_INVALID;
  *out = nullptr;
  if ((uint64_t)tile_first + tile_count > plan->plan.tiles.size()) return OJPHGPU_E_INVALID;
  if (plan->plan.skip_read || plan->plan.skip_recon) return OJPHGPU_E_INVALID;     // a decoding-only restriction
  HIPCHK(hipSetDevice(device));
  if (ensure_tables() != 0) return OJPHGPU_E_HIP;
  ojphgpu_encoder* e = new (std::nothrow) ojphgpu_encoder();
  if (!e) return OJPHGPU_E_NOMEM;
  struct Owner { ojphgpu_encoder* p; ~Owner() { if (p) ojphgpu_encoder_destroy(p); } } owner{ e };   // also when a container throws
  const Plan& P = plan->plan;
  e->handle = plan; e->P = &P; e->device = device; e->stream = (hipStream_t)stream;
  auto bail = [&](int rc) { return rc; };

  e->tiles = TileRange{ tile_first, tile_count };
  e->nframes = nframes;
  const TileRange tr = e->tiles;
  const uint64_t frame_elems = P.frame_elems;
  std::vector<ojphgpu_dwt_desc> dd; build_level_batches(P, tr, dd, e->batches);
  std::vector<ojphgpu_dwt_desc> idd;
  build_image_level_descs(P, tr, dd, e->batches, idd);
  replicate_levels(dd, idd, e->batches, nframes, P.arena_elems, frame_elems);
  std::vector<ojphgpu_convert_desc> cd; e->need_convert = build_convert_descs(P, tr, cd, e->conv_max_w, e->conv_max_h);
  replicate_converts(cd, nframes, P.arena_elems, P.frame_elems);
  e->block_ids = blocks_of_tiles(P, tr);
  if (nframes == 1 && max_recon_decomps(P) >= 2 && getenv("OJPHGPU_NO_OVERLAP") == nullptr) {
    auto top = [&](uint32_t id) { const Band& B = P.bands[P.blocks[id].band]; return B.res == P.style(B.comp).L; };
    auto mid = std::stable_partition(e->block_ids.begin(), e->block_ids.end(), top);
    e->n_top = (uint32_t)(mid - e->block_ids.begin());
    {
      // How many of the top resolution's blocks the side stream's launch takes; the rest join the main stream's launch behind
      // the lower levels.  The main stream waits for the side stream at the end of the encode, and a wait on another queue
      // that is not yet satisfied when it is reached costs ~20 us on top of what is waited for (kernel trace of a step:
      // the next launch started 22 us after the side stream's had ended) -- so the branches are cut for the SIDE one to
      // end first, by a model of what each costs on this device (MI355X, profiles/r06_*): the block coder 3.8 us per
      // million samples of the top resolution and 1.6 x that below it (more bytes per sample, shorter launches), a lower
      // analysis level its bytes at 3 TB/s or a 12 us launch floor.  8K 4:4:4: 90 % (step 0.957 -> 0.938 ms); a 4K frame's
      // main branch is the longer one already: 100 %.  OJPHGPU_ENC_TOP_SHARE=<percent> overrides the model.
      static const int share_env = [] { const char* v = getenv("OJPHGPU_ENC_TOP_SHARE"); const int x = v ? atoi(v) : 0; return x < 10 ? 0 : x > 100 ? 100 : x; }();
      double s_top = 0, s_low = 0;                         // millions of samples
      for (size_t i = 0; i < e->block_ids.size(); ++i) {
        const Block& k = P.blocks[e->block_ids[i]];
        (i < e->n_top ? s_top : s_low) += (double)k.r.w * k.r.h * 1e-6;
      }
      double dwt_us = 0, level_bytes = 8.0 * (s_top + s_low) * 1e6 / 4.0;
      for (uint32_t l = 2; l <= max_recon_decomps(P); ++l, level_bytes /= 4.0) dwt_us += std::max(12.0, level_bytes / 3.0e6);
      const double us_per_ms = 3.8, margin_us = 30.0;
      const double move = (s_top + margin_us / us_per_ms - 1.6 * s_low - dwt_us / us_per_ms) / 2.0;   // top samples the main branch should take
      int share = share_env;
      if (!share) share = s_top > 0 && move > 0 ? (int)(100.0 * (1.0 - move / s_top)) : 100;
      share = share < 50 ? 50 : share > 100 ? 100 : share;
      e->n_top = (uint32_t)((uint64_t)e->n_top * (uint32_t)share / 100u);
    }
    if (e->n_top == 0 || e->n_top == e->block_ids.size()) e->n_top = 0;
    else {
      // the side stream carries the long launch of the pair (the top resolution's blocks); the main stream's branch -- four
      // small transform launches, then the other blocks -- is the one that ends last when both share the chip evenly:
      // OJPHGPU_ENC_SIDE_PRIO=low lets the dispatcher prefer the main branch (high: the opposite; unset: equal)
      int prio_lo = 0, prio_hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      const char* pe = getenv("OJPHGPU_ENC_SIDE_PRIO");
      const int prio = pe && pe[0] == 'l' ? prio_lo : pe && pe[0] == 'h' ? prio_hi : (prio_lo + prio_hi) / 2;
      if (hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, pe ? prio : 0) != hipSuccess ||
          hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) return bail(OJPHGPU_E_HIP);
    }
  }
  std::vector<ojphgpu_cb_desc> bd(e->block_ids.size());
  uint64_t scratch_bytes = 0, samples = 0;
  bool over32_top = false, over32_rest = false, over128_top = false, over128_rest = false;
  for (size_t i = 0; i < bd.size(); ++i) {
    const Block& k = P.blocks[e->block_ids[i]]; const Band& B = P.bands[k.band];
    samples += (uint64_t)k.r.w * k.r.h;
    ojphgpu_cb_desc& d = bd[i]; memset(&d, 0, sizeof(d));
    const bool wide = is_wide(P, B.comp);                  // 64-bit samples: two arena elements each
    d.coef_off = B.plane_off + ((uint64_t)k.r.y0 * B.pitch + k.r.x0) * (wide ? 2u : 1u); d.pitch = B.pitch;
    d.w = (uint16_t)k.r.w; d.h = (uint16_t)k.r.h; d.K_max = (uint8_t)B.K_max; d.reversible = (uint8_t)((P.style(B.comp).rev ? 1 : 0) | (wide ? 4 : 0));
    d.missing_msbs = (uint8_t)(B.K_max - 1); d.num_passes = 1; d.delta = B.delta;
    d.data_off = scratch_bytes; d.scratch_cap = block_scratch_bytes(k.r.w, k.r.h, B.K_max);
    scratch_bytes += d.scratch_cap;
    (i < e->n_top ? e->widths_top : e->widths_rest) |= wide ? 32 : ((k.r.w > 64 ? 2 : 1) | ((d.reversible & 1) ? 4 : 8));   // which kernel variants the range needs
    if (!wide && k.r.w > 32 && k.r.w <= 64) (i < e->n_top ? over32_top : over32_rest) = true;
    if (!wide && k.r.w > 128) (i < e->n_top ? over128_top : over128_rest) = true;
  }
  if (!over32_top) e->widths_top |= 16;                    // every block of the range at most 32 samples wide (e.g. the IMF profile's 32 x 32)
  if (!over32_rest) e->widths_rest |= 16;
  if (!over128_top) e->widths_top |= 64;                   // no block wider than 128 samples: 128 x 32 blocks take the narrow kernel, too
  if (!over128_rest) e->widths_rest |= 64;
  if (nframes > 1) {                                      // replicate the block descriptors, frame-major
    const size_t nb = bd.size();
    bd.resize(nb * nframes);
    for (uint32_t f = 1; f < nframes; ++f)
      for (size_t i = 0; i < nb; ++i) {
        ojphgpu_cb_desc d = bd[i];
        d.coef_off += (uint64_t)f * P.arena_elems; d.data_off += (uint64_t)f * scratch_bytes;
        bd[f * nb + i] = d;
      }
    scratch_bytes *= nframes; samples *= nframes;
  }
  // The compacted output: scratch_bytes is a true upper bound of what the blocks can produce (K_max + 2 bits per
  // sample plus stuffing); samples * 3 bytes covers every bit depth up to 20 on any content and keeps the buffer of
  // the usual frames small -- deeper samples get the true bound, so that incompressible content cannot overflow.
  // The byte cursor is 32 bits wide: a batch whose bound exceeds 4 GiB is clamped there and a frame batch that
  // really produces more reports OJPHGPU_E_OVERFLOW (code fewer frames per batch).
  uint32_t kmax_all = 0;
  for (const Band& B : P.bands) kmax_all = std::max(kmax_all, B.K_max);
  uint64_t cap = kmax_all > 20 ? scratch_bytes : std::min<uint64_t>(scratch_bytes, samples * 3 + (1u << 20));
  cap = std::min<uint64_t>(cap, 0xFFFFFF00ull);
  // ... split into regions (see ojphgpu_objects.h): a region holds the sum of its own blocks' shares of that bound
  {
    const char* ev = getenv("OJPHGPU_ENC_REGIONS"); const long v = ev ? atol(ev) : 16;
    uint32_t R = (v >= 0 && v <= 64 && (v & (v - 1)) == 0) ? (uint32_t)v : 16u;
    if (bd.size() < 4 * (size_t)R) R = 0;                   // a handful of blocks: one cursor
    std::vector<uint64_t> rc(R ? R : 1, 0);
    if (R) for (size_t i = 0; i < bd.size(); ++i) {
      const uint64_t own = kmax_all > 20 ? bd[i].scratch_cap : std::min<uint64_t>(bd[i].scratch_cap, (uint64_t)bd[i].w * bd[i].h * 3 + 64);
      rc[(i < e->n_top ? i : i - e->n_top) & (R - 1)] += own;   // the kernel sees the index inside its launch
    }
    uint64_t total = 0;
    e->h_regions.assign(2 * (size_t)R, 0);
    for (uint32_t r = 0; r < R; ++r) {
      const uint64_t c = (rc[r] + (1u << 16) + 255) & ~(uint64_t)255;
      if (total + c > 0xFFFFFF00ull) { R = 0; break; }      // the byte offsets are 32 bits wide: one clamped cursor, as before
      e->h_regions[2 * r] = (uint32_t)total; e->h_regions[2 * r + 1] = (uint32_t)c;
      total += c;
    }
    e->nreg = R;
    if (R) cap = total; else e->h_regions.clear();
    e->counters_bytes = R ? (size_t)R * 128 : 16;
  }
  e->out_cap = (uint32_t)cap;

  if (e->arena.alloc(P.arena_elems * 4 * nframes) || e->dwt_descs.alloc(dd.size() * sizeof(dd[0])) ||
      e->img_descs.alloc(idd.size() * sizeof(dd[0])) || e->cb_descs.alloc(bd.size() * sizeof(bd[0])) || e->conv_descs.alloc(cd.size() * sizeof(cd[0])) ||
      e->scratch.alloc(scratch_bytes) || e->out.alloc(cap) ||
      e->results.alloc(bd.size() * sizeof(ojphgpu_cb_result)) || e->counters.alloc(e->counters_bytes) ||
      (e->nreg && e->regions.alloc(e->h_regions.size() * 4)))
    return bail(OJPHGPU_E_NOMEM);
  if (e->nreg && hipMemcpy(e->regions.p, e->h_regions.data(), e->h_regions.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (hipMemset(e->arena.p, 0, P.arena_elems * 4 * nframes) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (!dd.empty() && hipMemcpy(e->dwt_descs.p, dd.data(), dd.size() * sizeof(dd[0]), hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (!idd.empty() && hipMemcpy(e->img_descs.p, idd.data(), idd.size() * sizeof(idd[0]), hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (!bd.empty() && hipMemcpy(e->cb_descs.p, bd.data(), bd.size() * sizeof(bd[0]), hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (!cd.empty() && hipMemcpy(e->conv_descs.p, cd.data(), cd.size() * sizeof(cd[0]), hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  e->h_results.resize(bd.size());
  if (e->timer.init() != 0) return bail(OJPHGPU_E_HIP);
  owner.p = nullptr;
  *out = e;
  return OJPHGPU_OK;
}

int ojphgpu_encoder_run_container(ojphgpu_encoder* e, const void* d_image, int container);

extern "C" int ojphgpu_encoder_run_device(ojphgpu_encoder* e, const int32_t* d_image) { return ojphgpu_encoder_run_container(e, d_image, 32); }
extern "C" int ojphgpu_encoder_run_device16(ojphgpu_encoder* e, const uint16_t* d_image) { return ojphgpu_encoder_run_container(e, d_image, 16); }
extern "C" int ojphgpu_encoder_run_device8(ojphgpu_encoder* e, const uint8_t* d_image) { return ojphgpu_encoder_run_container(e, d_image, 8); }

// container: 32 = int32 samples, 16 / 8 = 16- / 8-bit samples (two's complement for signed components, else unsigned)
int ojphgpu_encoder_run_container(ojphgpu_encoder* e, const void* d_image, int container)
{
  if (!e || !d_image) return OJPHGPU_E_INVALID;
  const Plan& P = *e->P;
  if (container != 32 && container != 16 && container != 8) return OJPHGPU_E_INVALID;
  if (container != 32) for (const CompGeo& g : P.comps) if (g.bit_depth > (uint32_t)container) return OJPHGPU_E_INVALID;
  hipStream_t s = e->stream;
  Spans& T = e->timer;
  uint8_t* const d_out = (uint8_t*)(e->o_out ? e->o_out : e->out.p);          // a pipeline slot's buffers, or the object's own
  ojphgpu_cb_result* const res = (ojphgpu_cb_result*)(e->o_results ? e->o_results : e->results.p);
  uint32_t* const cnt = (uint32_t*)(e->o_counters ? e->o_counters : e->counters.p);
  HIPCHK(hipMemsetAsync(cnt, 0, e->counters_bytes, s));
  T.start(s);
  int rc = OJPHGPU_OK;
  if (e->need_convert) {
    const int sp = T.begin(SP_CONVERT, s);
    rc = ojphgpu_convert_forward_ex(s, &P.p, (const ojphgpu_convert_desc*)e->conv_descs.p, e->tiles.count * e->nframes,
                                    e->conv_max_w, e->conv_max_h, d_image, e->arena.p, container);
    T.end(sp, s);
  }
  if (rc) return rc;
  const ojphgpu_cb_desc* cbd = (const ojphgpu_cb_desc*)e->cb_descs.p;
  bool forked = false;
  auto fork_top = [&]() -> int {                            // the top resolution's blocks are ready to be coded
    forked = true;
    HIPCHK(hipEventRecord(e->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
    const int sh = T.begin(SP_HT_ENC, e->side);
    int r2 = ojphgpu::ht_encode_launch(e->side, cbd, e->n_top, e->arena.p, (uint8_t*)e->scratch.p, d_out,
                                       e->out_cap, res, cnt, cnt + 1, e->widths_top, (const uint32_t*)e->regions.p, e->nreg);
    if (r2) return r2;
    T.end(sh, e->side);
    HIPCHK(hipEventRecord(e->ev_join, e->side));
    return OJPHGPU_OK;
  };
  for (size_t i = 0; i < e->batches.size(); ++i) {
    const LevelBatch& b = e->batches[i];
    const int sp = T.begin(SP_DWT, s);
    if (b.img_first >= 0) {                                 // level shift / int->float applied in the loads
      ojphgpu_params pp = P.p; pp.reversible = b.rev ? 1 : 0;
      const ojphgpu_dwt_desc* idesc = (const ojphgpu_dwt_desc*)e->img_descs.p + b.img_first;
      rc = b.general ? ojphgpu_dwt_forward_general_image(s, &b.k, &pp, idesc, b.count, b.max_w, b.max_h, d_image, e->arena.p, container)
                     : ojphgpu_dwt_forward_image_ex(s, &pp, idesc, b.count, b.max_w, b.max_h, d_image, e->arena.p, container, b.nc == 3);
    } else if (b.general)                                   // 64-bit samples, Part-2 wavelets / decompositions: the general lifting kernels
      rc = ojphgpu_dwt_forward_general(s, &b.k, (const ojphgpu_dwt_desc*)e->dwt_descs.p + b.first, b.count, b.max_w, b.max_h, e->arena.p);
    else
      rc = ojphgpu_dwt_forward(s, b.rev ? 1 : 0, (const ojphgpu_dwt_desc*)e->dwt_descs.p + b.first, b.count,
                               b.max_w, b.max_h, e->arena.p);
    if (rc) return rc;
    T.end(sp, s);
    const bool top_done = b.depth == 0 && (i + 1 == e->batches.size() || e->batches[i + 1].depth != 0);
    if (e->n_top && top_done && (rc = fork_top()) != 0) return rc;
  }
  if (e->n_top && !forked && (rc = fork_top()) != 0) return rc;
  const uint32_t nb_all = (uint32_t)e->block_ids.size() * e->nframes;
  const int sh = T.begin(SP_HT_ENC, s);
  rc = ojphgpu::ht_encode_launch(s, cbd + e->n_top, nb_all - e->n_top, e->arena.p, (uint8_t*)e->scratch.p,
                                 d_out, e->out_cap, res + e->n_top, cnt, cnt + 1, e->widths_rest, (const uint32_t*)e->regions.p, e->nreg);
  if (rc) return rc;
  T.end(sh, s);
  if (e->n_top) HIPCHK(hipStreamWaitEvent(s, e->ev_join, 0));     // join
  T.finish(s);
  e->ran = true; e->fetched = false;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_encoder_set_timing(ojphgpu_encoder* e, int per_launch)
{
  if (!e) return OJPHGPU_E_INVALID;
  e->timer.detail = per_launch != 0;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_encoder_coded_bytes(ojphgpu_encoder* e, uint64_t* bytes)
{
  if (!e || !bytes || !e->ran) return OJPHGPU_E_INVALID;
  e->h_cursors.assign(e->counters_bytes / 4, 0);
  HIPCHK(hipMemcpyAsync(e->h_cursors.data(), e->counters.p, e->counters_bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  uint64_t total = e->h_cursors[0];
  for (uint32_t r = 1; r < e->nreg; ++r) total += e->h_cursors[32 * (size_t)r];
  *bytes = total;
  return e->h_cursors[1] ? OJPHGPU_E_OVERFLOW : OJPHGPU_OK;
}

// D2H of the block bytes + lengths of the last run (once per run); fills the plan-order coded-block
// table of frame `frame`
static int encoder_fetch(ojphgpu_encoder* e, uint32_t frame, std::vector<ojphgpu_coded_block>& cb)
{
  const Plan& P = *e->P;
  if (frame >= e->nframes) return OJPHGPU_E_INVALID;
  if (!e->fetched) {
    int rc = ojphgpu_encoder_coded_bytes(e, &e->nbytes);
    if (rc) return rc;
    const size_t rbytes = e->h_results.size() * sizeof(ojphgpu_cb_result);
    if (e->h_out.reserve(std::max<size_t>((size_t)e->nbytes + 16, (size_t)e->out_cap / 4)) || e->h_res.reserve(rbytes + 16))
      return OJPHGPU_E_NOMEM;
    if (rbytes) HIPCHK(hipMemcpyAsync(e->h_res.p, e->results.p, rbytes, hipMemcpyDeviceToHost, e->stream));
    // the used part of every region goes to a compact host buffer; the blocks' offsets are moved along below
    std::vector<uint64_t> hpos(e->nreg ? e->nreg : 1, 0);
    if (e->nreg) {
      uint64_t at = 0;
      for (uint32_t r = 0; r < e->nreg; ++r) {
        const uint32_t used = e->h_cursors[32 * (size_t)r];
        hpos[r] = at;
        if (used) HIPCHK(hipMemcpyAsync(e->h_out.p + at, (const uint8_t*)e->out.p + e->h_regions[2 * r], used, hipMemcpyDeviceToHost, e->stream));
        at += used;
      }
    } else if (e->nbytes) HIPCHK(hipMemcpyAsync(e->h_out.p, e->out.p, (size_t)e->nbytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (rbytes) memcpy(e->h_results.data(), e->h_res.p, rbytes);
    if (e->nreg)
      for (size_t i = 0; i < e->h_results.size(); ++i) {
        const uint32_t r = (uint32_t)(i < e->n_top ? i : i - e->n_top) & (e->nreg - 1);
        if (e->h_results[i].length) e->h_results[i].offset = (uint32_t)(e->h_results[i].offset - e->h_regions[2 * r] + hpos[r]);
      }
    e->fetched = true;
  }
  cb.assign(P.blocks.size(), ojphgpu_coded_block{ 0, 0, 0, 0, 0 });
  const size_t nb = e->block_ids.size();
  for (size_t i = 0; i < nb; ++i) {
    const ojphgpu_cb_result& r = e->h_results[(size_t)frame * nb + i];
    ojphgpu_coded_block& c = cb[e->block_ids[i]];
    c.offset = r.offset; c.len1 = r.length; c.len2 = 0;
    c.missing_msbs = r.length ? P.bands[P.blocks[e->block_ids[i]].band].K_max - 1 : 0;      // ojph_codeblock.cpp:148
    c.num_passes = r.length ? 1 : 0;
  }
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_encoder_finish(ojphgpu_encoder* e, uint8_t* h_out, size_t cap, size_t* out_len)
{
  if (!e || !out_len || !e->ran) return OJPHGPU_E_INVALID;
  return ojphgpu_encoder_finish_frame(e, 0, h_out, cap, out_len);
}

extern "C" int ojphgpu_encoder_finish_frame(ojphgpu_encoder* e, uint32_t frame, uint8_t* h_out, size_t cap, size_t* out_len)
{
  if (!e || !out_len || !e->ran) return OJPHGPU_E_INVALID;
  if (e->tiles.first != 0 || e->tiles.count != e->P->tiles.size()) return OJPHGPU_E_INVALID;   // use _finish_tiles
  return no_throw([&]() -> int {
  std::vector<ojphgpu_coded_block> cb;
  const bool tm = getenv("OJPHGPU_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t0 = now();
  int rc = encoder_fetch(e, frame, cb);
  if (rc) return rc;
  auto t1 = now();
  rc = ojphgpu_t2_write(e->handle, e->h_out.p, cb.data(), h_out, cap, out_len);
  if (tm) fprintf(stderr, "ojphgpu: finish frame %u: D2H %.2f ms, Tier-2 %.2f ms\n", frame,
                  std::chrono::duration<double, std::milli>(t1 - t0).count(),
                  std::chrono::duration<double, std::milli>(now() - t1).count());
  return rc;
  });
}

extern "C" int ojphgpu_encoder_finish_tiles(ojphgpu_encoder* e, uint8_t* h_out, size_t cap, size_t* out_len,
                                             uint32_t* tile_part_len)
{
  if (!e || !out_len || !e->ran) return OJPHGPU_E_INVALID;
  return no_throw([&]() -> int {
    std::vector<ojphgpu_coded_block> cb;
    int rc = encoder_fetch(e, 0, cb);
    if (rc) return rc;
    return ojphgpu_t2_write_tiles(e->handle, e->h_out.p, cb.data(), e->tiles.first, e->tiles.count, h_out, cap,
                                  out_len, tile_part_len);
  });
}

namespace ojphgpu {
int assemble_launch(void* stream, const T2Job* d_jobs, uint32_t njobs, const uint8_t* d_blob, const uint8_t* d_data, uint8_t* d_out);
}

// The tile-parts of the range assembled in HBM (kernels_assemble.hip) instead of on the host: only the block
// lengths come to the host, the layout goes back, and the bytes stay on the device -- where the final
// codestream gather of a multi-GPU encode picks them up (RCCL send over xGMI, openjph_amd/shard.py).
extern "C" int ojphgpu_encoder_finish_tiles_device(ojphgpu_encoder* e, uint8_t* d_out, size_t cap, size_t* out_len,
                                                    uint32_t* tile_part_len)
{
  if (!e || !out_len || !e->ran) return OJPHGPU_E_INVALID;
  return no_throw([&]() -> int {
    const Plan& P = *e->P;
    uint64_t nbytes = 0;
    int rc = ojphgpu_encoder_coded_bytes(e, &nbytes);
    if (rc) return rc;
    const size_t nb = e->block_ids.size();
    const size_t rbytes = e->h_results.size() * sizeof(ojphgpu_cb_result);
    if (rbytes) HIPCHK(hipMemcpyAsync(e->h_results.data(), e->results.p, rbytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->fetched = false;                                   // h_results holds device offsets again
    std::vector<ojphgpu_coded_block> cb(P.blocks.size(), ojphgpu_coded_block{ 0, 0, 0, 0, 0 });
    for (size_t i = 0; i < nb; ++i) {
      const ojphgpu_cb_result& r = e->h_results[i];
      ojphgpu_coded_block& c = cb[e->block_ids[i]];
      c.offset = r.offset; c.len1 = r.length; c.len2 = 0;
      c.missing_msbs = r.length ? P.bands[P.blocks[e->block_ids[i]].band].K_max - 1 : 0;
      c.num_passes = r.length ? 1 : 0;
    }
    T2Layout L;
    rc = t2_layout_tiles(P, cb.data(), e->tiles.first, (size_t)e->tiles.first + e->tiles.count, L, tile_part_len);
    if (rc) return rc;
    *out_len = (size_t)L.total;
    if (!d_out || cap < L.total) return OJPHGPU_E_OVERFLOW;
    const size_t jbytes = L.jobs.size() * sizeof(T2Job), boff = (jbytes + 63) & ~(size_t)63;
    DeviceBuf lay;
    if (lay.alloc(boff + L.blob.size() + 64)) return OJPHGPU_E_NOMEM;
    struct Free { DeviceBuf& b; ~Free() { b.release(); } } fr{ lay };
    if (jbytes) HIPCHK(hipMemcpyAsync(lay.p, L.jobs.data(), jbytes, hipMemcpyHostToDevice, e->stream));
    if (!L.blob.empty()) HIPCHK(hipMemcpyAsync((uint8_t*)lay.p + boff, L.blob.data(), L.blob.size(), hipMemcpyHostToDevice, e->stream));
    rc = assemble_launch(e->stream, (const T2Job*)lay.p, (uint32_t)L.jobs.size(), (const uint8_t*)lay.p + boff,
                         (const uint8_t*)(e->o_out ? e->o_out : e->out.p), d_out);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));            // the layout staging goes away with this call
    return OJPHGPU_OK;
  });
}

static int encode_host(ojphgpu_encoder* e, const void* h_image, int container, uint8_t* h_out, size_t cap, size_t* out_len)
{
  if (!e || !h_image) return OJPHGPU_E_INVALID;
  const Plan& P = *e->P;
  const size_t bytes = (size_t)P.frame_elems * 4 * e->nframes;          // sized for the wider container, used by both
  if (!e->image.p && e->image.alloc(bytes)) return OJPHGPU_E_NOMEM;
  HIPCHK(hipMemcpyAsync(e->image.p, h_image, bytes / (32 / container), hipMemcpyHostToDevice, e->stream));
  int rc = ojphgpu_encoder_run_container(e, e->image.p, container);
  if (rc) return rc;
  return ojphgpu_encoder_finish(e, h_out, cap, out_len);
}

extern "C" int ojphgpu_encode(ojphgpu_encoder* e, const int32_t* h_image, uint8_t* h_out, size_t cap, size_t* out_len)
{
  return encode_host(e, h_image, 32, h_out, cap, out_len);
}

extern "C" int ojphgpu_encode16(ojphgpu_encoder* e, const uint16_t* h_image, uint8_t* h_out, size_t cap, size_t* out_len)
{
  return encode_host(e, h_image, 16, h_out, cap, out_len);
}

extern "C" int ojphgpu_encoder_timing(ojphgpu_encoder* e, float out[4])
{
  if (!e || !out || !e->ran) return OJPHGPU_E_INVALID;
  if (e->timer.read(SP_CONVERT, &out[0]) || e->timer.read(SP_DWT, &out[1]) || e->timer.read(SP_HT_ENC, &out[2]) ||
      e->timer.read(-1, &out[3])) return OJPHGPU_E_HIP;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_encoder_ht_timing(ojphgpu_encoder* e, float* out, uint32_t cap, uint32_t* n, uint32_t* n_top)
{
  if (!e || !out || !n || !e->ran) return OJPHGPU_E_INVALID;
  int k = e->timer.read_each(SP_HT_ENC, out, cap);
  if (k < 0) return OJPHGPU_E_HIP;
  *n = (uint32_t)k;
  if (n_top) *n_top = e->n_top;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_encoder_level_timing(ojphgpu_encoder* e, float* out, uint32_t cap, uint32_t* n)
{
  if (!e || !out || !n || !e->ran) return OJPHGPU_E_INVALID;
  int k = e->timer.read_each(SP_DWT, out, cap);
  if (k < 0) return OJPHGPU_E_HIP;
  *n = (uint32_t)k;
  return OJPHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" void ojphgpu_decoder_destroy(ojphgpu_decoder* d)
{
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (DeviceBuf* b : { &d->arena, &d->image, &d->dwt_descs, &d->img_descs, &d->cb_descs, &d->conv_descs, &d->data, &d->status, &d->quads, &d->aux,
                       &d->fstate })
    b->release();
  if (d->h_retry) (void)hipHostFree(d->h_retry);
  if (d->side) (void)hipStreamDestroy(d->side);
  if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
  if (d->ev_join) (void)hipEventDestroy(d->ev_join);
  d->timer.destroy();
  delete d;
}

extern "C" int ojphgpu_decoder_create(const ojphgpu_plan* plan, int device, void* stream, ojphgpu_decoder** out)
{
  if (!plan) return OJPHGPU_E_INVALID;
  return ojphgpu_decoder_create_tiles(plan, device, stream, 0, (uint32_t)plan->plan.tiles.size(), out);
}

static int decoder_create(const ojphgpu_plan* const* plans, uint32_t nframes, int device, void* stream, uint32_t tile_first,
                          uint32_t tile_count, ojphgpu_decoder** out);

extern "C" int ojphgpu_decoder_create_tiles(const ojphgpu_plan* plan, int device, void* stream, uint32_t tile_first,
                                             uint32_t tile_count, ojphgpu_decoder** out)
{
  return no_throw([&] { return decoder_create(&plan, 1, device, stream, tile_first, tile_count, out); });
}

extern "C" int ojphgpu_decoder_create_batch(const ojphgpu_plan* const* plans, uint32_t num_frames, int device, void* stream,
                                             ojphgpu_decoder** out)
{
  if (!plans || num_frames == 0 || !plans[0]) return OJPHGPU_E_INVALID;
  return no_throw([&] { return decoder_create(plans, num_frames, device, stream, 0, (uint32_t)plans[0]->plan.tiles.size(), out); });
}

// Two parsed codestreams describe frames one decoder object can take turns on: same frame format, same
// place for every code-block.  Quantisation may differ (K_max / delta are per frame).
int ojphgpu_same_frame_geometry(const Plan& P, const Plan& Q, bool compare_blocks)
{
  if (Q.coded.size() != Q.blocks.size()) return OJPHGPU_E_INVALID;    // plans must come from ojphgpu_t2_parse
  if (Q.skip_read != P.skip_read || Q.skip_recon != P.skip_recon) return OJPHGPU_E_INVALID;
  if (Q.blocks.size() != P.blocks.size() || Q.arena_elems != P.arena_elems || Q.p.width != P.p.width ||
      Q.p.height != P.p.height || Q.p.num_comps != P.p.num_comps || Q.p.bit_depth != P.p.bit_depth ||
      Q.p.is_signed != P.p.is_signed || Q.p.reversible != P.p.reversible || Q.p.num_decomps != P.p.num_decomps ||
      Q.p.color_transform != P.p.color_transform || Q.p.tile_w != P.p.tile_w || Q.p.tile_h != P.p.tile_h ||
      Q.p.block_w != P.p.block_w || Q.p.block_h != P.p.block_h ||
      Q.p.image_x0 != P.p.image_x0 || Q.p.image_y0 != P.p.image_y0 || Q.p.tile_x0 != P.p.tile_x0 || Q.p.tile_y0 != P.p.tile_y0 ||
      memcmp(Q.p.comp_dx, P.p.comp_dx, sizeof(P.p.comp_dx)) != 0 || memcmp(Q.p.comp_dy, P.p.comp_dy, sizeof(P.p.comp_dy)) != 0 ||
      memcmp(Q.p.comp_depth, P.p.comp_depth, sizeof(P.p.comp_depth)) != 0 || memcmp(Q.p.comp_sign, P.p.comp_sign, sizeof(P.p.comp_sign)) != 0 ||
      Q.nlt3 != P.nlt3 || Q.wide != P.wide || Q.bands.size() != P.bands.size() || memcmp(Q.p.coc, P.p.coc, sizeof(P.p.coc)) != 0)
    return OJPHGPU_E_INVALID;
  if (!compare_blocks) return OJPHGPU_OK;
  // the launches are laid out from the first frame's geometry: every block must sit where that frame has it
  // (precinct sizes move the code-block grid)
  for (size_t i = 0; i < P.blocks.size(); ++i) {
    const Block& a = P.blocks[i]; const Block& b = Q.blocks[i];
    if (a.band != b.band || a.r.x0 != b.r.x0 || a.r.y0 != b.r.y0 || a.r.w != b.r.w || a.r.h != b.r.h) return OJPHGPU_E_INVALID;
  }
  for (size_t i = 0; i < P.bands.size(); ++i)
    if (P.bands[i].plane_off != Q.bands[i].plane_off || P.bands[i].pitch != Q.bands[i].pitch) return OJPHGPU_E_INVALID;
  return OJPHGPU_OK;
}

// Block descriptors of one frame: geometry from P (the decoder's plan), what the packet headers and the
// QCD / QCC of THIS frame's codestream say from Q.  arena_off = element offset of the frame's arena,
// data_base = where the frame's byte range [fi.first, fi.first + fi.len) of the codestream will sit in the
// device data buffer.  scratch_cap / reserved are left to ojphgpu_ht_decode_layout.
void ojphgpu_decoder_fill_descs(const Plan& P, const Plan& Q, const std::vector<uint32_t>& ids, uint64_t arena_off,
                                uint64_t data_base, ojphgpu_cb_desc* bd, DecFrameInfo& fi)
{
  uint64_t max_off = 0, min_off = ~0ull;
  fi.any_refine = false; fi.max_len1 = 0; fi.kinds = 0; fi.pads.clear(); fi.pad_len = 0;
  // blocks whose tile-part ended before their bytes did: the reference decodes what there is with zeros behind it
  // (bb_read_chunk, ojph_bitbuffer_read.h:134-150).  `coded` calls them not coded; here they get what the packet header
  // said and a place of their own behind the uploaded byte range (PadCopy), where the upload puts bytes + zeros.
  std::vector<std::pair<uint32_t, const Plan::PaddedBlock*>> padded_idx;   // by block id, first entry of an id wins (as a linear search would)
  padded_idx.reserve(Q.padded.size());
  for (const Plan::PaddedBlock& pb : Q.padded) padded_idx.emplace_back(pb.block, &pb);
  std::stable_sort(padded_idx.begin(), padded_idx.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  auto padded_of = [&](uint32_t id) -> const Plan::PaddedBlock* {
    auto it = std::lower_bound(padded_idx.begin(), padded_idx.end(), id, [](const auto& a, uint32_t v) { return a.first < v; });
    return it != padded_idx.end() && it->first == id ? it->second : nullptr;
  };
  std::vector<std::pair<size_t, const Plan::PaddedBlock*>> padded_at;
  for (size_t i = 0; i < ids.size(); ++i) {
    const Block& k = P.blocks[ids[i]]; const Band& B = Q.bands[k.band];
    const Plan::PaddedBlock* pb = Q.padded.empty() ? nullptr : padded_of(ids[i]);
    const CodedBlock& c = pb ? pb->hdr : Q.coded[ids[i]];   // this frame's own K_max / delta
    if (pb) padded_at.emplace_back(i, pb);
    ojphgpu_cb_desc& o = bd[i]; memset(&o, 0, sizeof(o));
    const bool wide = is_wide(Q, B.comp);                  // (same_frame_geometry: the same components as in P)
    o.coef_off = arena_off + P.bands[k.band].plane_off + ((uint64_t)k.r.y0 * B.pitch + k.r.x0) * (wide ? 2u : 1u); o.pitch = B.pitch;
    o.w = (uint16_t)k.r.w; o.h = (uint16_t)k.r.h; o.K_max = (uint8_t)B.K_max;
    o.reversible = (uint8_t)((Q.style(B.comp).rev ? 1u : 0u) | (Q.style(B.comp).causal ? 2u : 0u) | (wide ? 4u : 0u));   // bit 1: vertically causal, bit 2: 64-bit samples
    o.missing_msbs = (uint8_t)std::min<uint32_t>(c.missing_msbs, 255); o.num_passes = (uint8_t)c.num_passes;
    if (c.num_passes > 1 && c.len2 > 0) { fi.any_refine = true; fi.kinds |= 16; }
    fi.kinds |= wide ? 32 : ((k.r.w > 64 ? 2 : 1) | ((o.reversible & 1u) ? 4 : 8) | (k.r.w > 32 ? 64 : 0) | (k.r.w > 16 ? 128 : 0) | 256);   // 256: the width bits 6 / 7 are filled in
    o.delta = B.delta; o.len1 = c.len1; o.len2 = c.len2; o.data_off = c.offset;
    fi.max_len1 = std::max(fi.max_len1, c.len1);
    if ((c.len1 + c.len2) && !pb) {
      max_off = std::max<uint64_t>(max_off, c.offset + c.len1 + c.len2);
      min_off = std::min<uint64_t>(min_off, c.offset);
    }
  }
  if (min_off > max_off) min_off = max_off = 0;
  min_off &= ~(uint64_t)15;                                             // only this byte range of the codestream is uploaded
  for (size_t i = 0; i < ids.size(); ++i) {
    ojphgpu_cb_desc& o = bd[i];
    if (o.len1 + o.len2) o.data_off = o.data_off - min_off + data_base; else o.data_off = 0;
  }
  fi.first = min_off; fi.len = max_off - min_off;
  uint64_t at = (fi.len + 63) & ~(uint64_t)63;                          // padded blocks: behind the range, 64 bytes apart at least
  for (const auto& pp : padded_at) {
    ojphgpu_cb_desc& o = bd[pp.first]; const Plan::PaddedBlock& pb = *pp.second;
    const uint32_t total = pb.hdr.len1 + pb.hdr.len2;
    if (total == 0) continue;
    at += 64;                                                           // (room below the block: the VLC partner's 16-byte loads)
    fi.pads.push_back(PadCopy{ pb.hdr.offset, at, std::min(pb.got, total), total });
    o.data_off = data_base + at;
    at = (at + total + 63) & ~(uint64_t)63;
  }
  if (!fi.pads.empty()) at += 64;                                       // (ojphgpu_decoder_upload_pads zeroes a 64-byte margin behind the last block too)
  fi.pad_len = at - ((fi.len + 63) & ~(uint64_t)63);
}

int ojphgpu_decoder_upload_pads(hipStream_t s, uint8_t* d_frame_data, const uint8_t* h_codestream, size_t cs_len, const std::vector<PadCopy>& pads)
{
  for (const PadCopy& c : pads) {
    if (c.src + c.got > cs_len) return OJPHGPU_E_INVALID;
    HIPCHK(hipMemsetAsync(d_frame_data + c.dst - 64, 0, (size_t)c.total + 128, s));     // zeros: the padding, and a margin either side
    if (c.got) HIPCHK(hipMemcpyAsync(d_frame_data + c.dst, h_codestream + c.src, c.got, hipMemcpyHostToDevice, s));
  }
  return OJPHGPU_OK;
}

static int decoder_create(const ojphgpu_plan* const* plans, uint32_t nframes, int device, void* stream, uint32_t tile_first,
                          uint32_t tile_count, ojphgpu_decoder** out)
{
  if (!plans || !plans[0] || !out || nframes == 0) return OJPHGPU_E_INVALID;
  *out = nullptr;
  const ojphgpu_plan* plan = plans[0];
  const Plan& P = plan->plan;
  if ((uint64_t)tile_first + tile_count > P.tiles.size()) return OJPHGPU_E_INVALID;
  for (uint32_t f = 0; f < nframes; ++f) {                              // every frame of a batch has the same geometry
    if (!plans[f]) return OJPHGPU_E_INVALID;
    const int rc = ojphgpu_same_frame_geometry(P, plans[f]->plan, f != 0);
    if (rc) return rc;
  }
  HIPCHK(hipSetDevice(device));
  if (ensure_tables() != 0) return OJPHGPU_E_HIP;
  ojphgpu_decoder* d = new (std::nothrow) ojphgpu_decoder();
  if (!d) return OJPHGPU_E_NOMEM;
  struct Owner { ojphgpu_decoder* p; ~Owner() { if (p) ojphgpu_decoder_destroy(p); } } owner{ d };   // also when a container throws
  d->P = &P; d->device = device; d->stream = (hipStream_t)stream;
  d->cus = ojphgpu::device_cus(device);
  auto bail = [&](int rc) { return rc; };

  d->tiles = TileRange{ tile_first, tile_count };
  d->nframes = nframes;
  const TileRange tr = d->tiles;
  const uint64_t frame_elems = P.frame_elems;
  std::vector<ojphgpu_dwt_desc> dd; build_level_batches(P, tr, dd, d->batches);
  std::vector<ojphgpu_dwt_desc> idd;
  build_image_level_descs(P, tr, dd, d->batches, idd);
  replicate_levels(dd, idd, d->batches, nframes, P.arena_elems, frame_elems);
  std::reverse(d->batches.begin(), d->batches.end());                 // synthesis: lowest resolution first
  std::vector<ojphgpu_convert_desc> cd; d->need_convert = build_convert_descs(P, tr, cd, d->conv_max_w, d->conv_max_h);
  replicate_converts(cd, nframes, P.arena_elems, P.frame_elems);
  std::vector<uint32_t> ids = blocks_of_tiles(P, tr);
  // Overlap of the lower synthesis levels with the block decoder: the blocks below the top
  // resolution come first in descriptor order; once step 2 has produced them, the small, latency-bound
  // launches of levels L .. 2 run on a second stream while step 2 works through the top resolution's
  // blocks (3/4 of the samples) on the main one.  Prep and step 1 stay single launches over all blocks:
  // step 1 costs one serial chain however few blocks it is given, splitting it would pay that twice
  // (measured: 0.96 vs 0.91 ms at 8K with every stage split in two).
  if (nframes == 1 && max_recon_decomps(P) >= 2 && d->batches.size() >= 2 && d->batches.front().depth > 0 && !P.any_wide &&
      getenv("OJPHGPU_NO_OVERLAP") == nullptr) {
    auto low = [&](uint32_t id) { const Band& B = P.bands[P.blocks[id].band]; return B.res < P.recon_decomps(B.comp); };
    auto mid = std::stable_partition(ids.begin(), ids.end(), low);
    d->n_low = (uint32_t)(mid - ids.begin());
    int prio_lo = 0, prio_hi = 0;                         // the short launches of the side stream go first at the dispatcher
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (d->n_low == 0 || d->n_low == ids.size()) d->n_low = 0;
    else if (hipStreamCreateWithPriority(&d->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
             hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming) != hipSuccess) return bail(OJPHGPU_E_HIP);
  }
  d->block_ids = ids;
  d->nblocks = (uint32_t)(ids.size() * nframes);
  std::vector<ojphgpu_cb_desc> bd(ids.size() * nframes);
  uint64_t nquads = 0, naux = 0, data_total = 0;
  d->f_first.assign(nframes, 0); d->f_len.assign(nframes, 0); d->f_base.assign(nframes, 0); d->f_pads.assign(nframes, {});
  for (uint32_t f = 0; f < nframes; ++f) {
    DecFrameInfo fi;
    ojphgpu_decoder_fill_descs(P, plans[f]->plan, ids, (uint64_t)f * P.arena_elems, data_total, bd.data() + (size_t)f * ids.size(), fi);
    d->any_refine |= fi.any_refine; d->kinds |= fi.kinds; d->max_len1 = std::max(d->max_len1, fi.max_len1);
    d->f_first[f] = (size_t)fi.first; d->f_len[f] = (size_t)fi.len; d->f_base[f] = (size_t)data_total;
    d->f_pads[f] = fi.pads;
    data_total += fi.data_bytes();
  }
  d->data_first = d->f_first[0];
  d->data_len = (size_t)data_total;
  // scratch of step 1's records and of the flat VLC / MEL strings (prep and step 1 are single launches
  // over all descriptors)
  if (ojphgpu_ht_decode_layout(bd.data(), (uint32_t)bd.size(), &nquads, &naux) != OJPHGPU_OK) return bail(OJPHGPU_E_INVALID);
  if (nquads >= 0xFFFFFFFFull || naux >= 0xFFFFFFFFull) return bail(OJPHGPU_E_INVALID);
  if (d->quads.alloc((size_t)nquads * 4 + 64) || d->aux.alloc((size_t)naux * 4 + 64)) return bail(OJPHGPU_E_NOMEM);
  for (const ojphgpu_cb_desc& b : bd) d->max_block_h = std::max<uint32_t>(d->max_block_h, b.h);
  if (ojphgpu::dec_fuses()) {                              // flags and per-block state of the fused step 1 + step 2 launch
    const size_t fw = (size_t)ojphgpu::ht_decode_fused_state_words((uint32_t)bd.size());
    if (d->fstate.alloc(fw * 4)) return bail(OJPHGPU_E_NOMEM);
    if (hipMemset(d->fstate.p, 0, fw * 4) != hipSuccess) return bail(OJPHGPU_E_HIP);
    // (a word the launch can reach in host memory: see ojphgpu_decoder::h_retry; without it the notice is simply not given)
    void* hp = nullptr; void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
      d->h_retry = (uint32_t*)hp; d->d_h_retry = (uint32_t*)dp; *d->h_retry = 0;
    } else { (void)hipGetLastError(); if (hp) (void)hipHostFree(hp); }
  }
  if (d->arena.alloc(P.arena_elems * 4 * nframes) || d->dwt_descs.alloc(dd.size() * sizeof(dd[0])) ||
      d->img_descs.alloc(idd.size() * sizeof(dd[0])) || d->cb_descs.alloc(bd.size() * sizeof(bd[0])) || d->conv_descs.alloc(cd.size() * sizeof(cd[0])) ||
      d->data.alloc(d->data_len + 128) || d->status.alloc(bd.size() + 16))
    return bail(OJPHGPU_E_NOMEM);
  if (hipMemset(d->status.p, 0, bd.size() + 16) != hipSuccess) return bail(OJPHGPU_E_HIP);      // (the RETRY word behind the status bytes)
  if (hipMemset(d->arena.p, 0, P.arena_elems * 4 * nframes) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (!dd.empty() && hipMemcpy(d->dwt_descs.p, dd.data(), dd.size() * sizeof(dd[0]), hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (!idd.empty() && hipMemcpy(d->img_descs.p, idd.data(), idd.size() * sizeof(idd[0]), hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (!bd.empty() && hipMemcpy(d->cb_descs.p, bd.data(), bd.size() * sizeof(bd[0]), hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (!cd.empty() && hipMemcpy(d->conv_descs.p, cd.data(), cd.size() * sizeof(cd[0]), hipMemcpyHostToDevice) != hipSuccess) return bail(OJPHGPU_E_HIP);
  if (d->timer.init() != 0) return bail(OJPHGPU_E_HIP);
  owner.p = nullptr;
  *out = d;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_decoder_upload(ojphgpu_decoder* d, const uint8_t* h_codestream, size_t len)
{
  return ojphgpu_decoder_upload_frame(d, 0, h_codestream, len);
}

extern "C" int ojphgpu_decoder_upload_frame(ojphgpu_decoder* d, uint32_t frame, const uint8_t* h_codestream, size_t len)
{
  if (!d || !h_codestream || frame >= d->nframes || len < d->f_first[frame] + d->f_len[frame]) return OJPHGPU_E_INVALID;
  if (d->f_len[frame])
    HIPCHK(hipMemcpyAsync((uint8_t*)d->data.p + d->f_base[frame], h_codestream + d->f_first[frame], d->f_len[frame],
                          hipMemcpyHostToDevice, d->stream));
  return ojphgpu_decoder_upload_pads(d->stream, (uint8_t*)d->data.p + d->f_base[frame], h_codestream, len, d->f_pads[frame]);
}

// prep + step 1 of every block (one launch each: step 1 costs one serial chain however few blocks it gets)
static int decode_chains(ojphgpu_decoder* d, hipStream_t s)
{
  if (d->nblocks == 0 || (d->kinds & 3) == 0) return OJPHGPU_OK;     // (nothing but blocks of the 64-bit sample path: decode_samples)
  Spans& T = d->timer;
  const ojphgpu_cb_desc* cbd = (const ojphgpu_cb_desc*)(d->o_cb_descs ? d->o_cb_descs : d->cb_descs.p);
  uint8_t* status = (uint8_t*)(d->o_status ? d->o_status : d->status.p);
  const uint8_t* data = (const uint8_t*)(d->o_data ? d->o_data : d->data.p);
  int sp = T.begin(SP_PREP, s);
  int rc = ojphgpu_ht_decode_prep(s, cbd, d->nblocks, data, (uint32_t*)d->aux.p);
  if (rc) return rc;
  T.end(sp, s);
  sp = T.begin(SP_STEP1, s);
  rc = ojphgpu_ht_decode_step1(s, cbd, d->nblocks, data, (const uint32_t*)d->aux.p, (uint32_t*)d->quads.p, status);
  if (rc) return rc;
  T.end(sp, s);
  return OJPHGPU_OK;
}

// step 2 (+ the refinement passes) over descriptors [first, first + count) on stream s
static int decode_samples(ojphgpu_decoder* d, hipStream_t s, uint32_t first, uint32_t count)
{
  if (count == 0) return OJPHGPU_OK;
  Spans& T = d->timer;
  const ojphgpu_cb_desc* cbd = (const ojphgpu_cb_desc*)(d->o_cb_descs ? d->o_cb_descs : d->cb_descs.p) + first;
  uint8_t* status = (uint8_t*)(d->o_status ? d->o_status : d->status.p) + first;
  const uint8_t* data = (const uint8_t*)(d->o_data ? d->o_data : d->data.p);
  int sp = T.begin(SP_STEP2, s);
  int rc = (d->kinds & 3) ? ojphgpu::ht_decode_step2_launch(s, cbd, count, data, (const uint32_t*)d->quads.p, d->arena.p, status, d->kinds & ~32) : OJPHGPU_OK;
  if (rc) return rc;
  if (d->kinds & 32) {                                      // the blocks on the 64-bit sample path: all their launches (the others skip them)
    rc = ojphgpu::ht_decode64_launch(s, cbd, count, data, (uint32_t*)d->aux.p, (uint32_t*)d->quads.p, d->arena.p, status, d->any_refine ? 1 : 0);
    if (rc) return rc;
  }
  T.end(sp, s);
  if (d->any_refine && (d->kinds & 3)) {
    sp = T.begin(SP_REFINE, s);
    rc = ojphgpu_ht_decode_refine(s, cbd, count, data, (const uint32_t*)d->quads.p, d->arena.p, status);
    if (rc) return rc;
    T.end(sp, s);
  }
  return OJPHGPU_OK;
}

int ojphgpu_decoder_run_container(ojphgpu_decoder* d, void* d_image, int container);
static int decode_host(ojphgpu_decoder* d, const uint8_t* h_codestream, size_t len, void* h_image, int container);

extern "C" int ojphgpu_decoder_run_device(ojphgpu_decoder* d, int32_t* d_image) { return ojphgpu_decoder_run_container(d, d_image, 32); }
extern "C" int ojphgpu_decoder_run_device16(ojphgpu_decoder* d, uint16_t* d_image) { return ojphgpu_decoder_run_container(d, d_image, 16); }
extern "C" int ojphgpu_decoder_run_device8(ojphgpu_decoder* d, uint8_t* d_image) { return ojphgpu_decoder_run_container(d, d_image, 8); }

int ojphgpu_decoder_run_container(ojphgpu_decoder* d, void* d_image, int container)
{
  if (!d || !d_image) return OJPHGPU_E_INVALID;
  const Plan& P = *d->P;
  if (container != 32 && container != 16 && container != 8) return OJPHGPU_E_INVALID;
  if (container != 32) for (const CompGeo& g : P.comps) if (g.bit_depth > (uint32_t)container) return OJPHGPU_E_INVALID;
  hipStream_t s = d->stream;
  Spans& T = d->timer;
  // A run before this one asked to be decoded again and nobody collected it (ojphgpu_decoder_failed_blocks does): its
  // frame was handed out partly decoded.  The launch writes its epoch into h_retry only when its wait has run out (about two
  // seconds after it started), so the notice may arrive more than one call later: epochs only grow, and every epoch found
  // there that has not been acknowledged yet is reported once.  (A frame pipeline's runs -- o_status set -- are always
  // collected by the pipeline.)
  if (!d->o_status && !d->force_separate && d->h_retry) {
    const uint32_t gave_up = *(volatile uint32_t*)d->h_retry;
    if ((int32_t)(gave_up - d->retry_acked) > 0) {
      d->retry_acked = gave_up; d->uncollected = false;
      return OJPHGPU_E_UNCOLLECTED;
    }
  }
  T.start(s);
  // One launch for step 1 and step 2 (chains first, step-2 workers behind them slice by slice, kernels_ht_dec.hip) when
  // every block is at most 64 samples wide, of one wavelet and without refinement passes -- and where it pays: blocks
  // of 64 rows, few enough for resident workers (ht_decode_fused_pays); the synthesis levels follow on the same stream.  Otherwise: the separate launches, with the lower synthesis levels beside step 2 of the top resolution.
  const bool fused = d->fstate.p && !d->force_separate && !d->fused_off && !d->any_refine && (d->kinds & (3 | 32)) == 1 && ((d->kinds & 12) == 4 || (d->kinds & 12) == 8) &&
                     d->nblocks > 0 && ojphgpu::ht_decode_fused_pays(d->nblocks, d->max_block_h, d->cus);
  d->last_fused = fused; d->last_image = d_image; d->last_container = container;
  const uint32_t n_low = fused ? 0u : d->n_low;
  int rc;
  if (fused) {
    const ojphgpu_cb_desc* cbd = (const ojphgpu_cb_desc*)(d->o_cb_descs ? d->o_cb_descs : d->cb_descs.p);
    uint8_t* status = (uint8_t*)(d->o_status ? d->o_status : d->status.p);
    const uint8_t* data = (const uint8_t*)(d->o_data ? d->o_data : d->data.p);
    const int sp = T.begin(SP_STEP2, s);
    rc = ojphgpu::ht_decode_fused_launch(s, cbd, d->nblocks, data, (uint32_t*)d->quads.p, d->arena.p, status, (uint32_t*)d->fstate.p,
                                         ++d->fused_epoch, d->max_block_h, d->kinds, d->cus, d->d_h_retry);
    if (rc) return rc;
    d->uncollected = true;
    T.end(sp, s);
  } else {
    rc = decode_chains(d, s);
    if (rc) return rc;
    rc = decode_samples(d, s, 0, n_low ? n_low : d->nblocks);
    if (rc) return rc;
  }
  if (n_low) {                                              // fork: the lower synthesis levels on the side stream
    HIPCHK(hipEventRecord(d->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(d->side, d->ev_fork, 0));
  }
  bool joined = false;
  auto finish_blocks = [&]() -> int {                       // meanwhile, on the main stream: the top resolution's blocks
    joined = true;
    HIPCHK(hipEventRecord(d->ev_join, d->side));
    int r2 = decode_samples(d, s, n_low, d->nblocks - n_low);
    if (r2) return r2;
    HIPCHK(hipStreamWaitEvent(s, d->ev_join, 0));           // join before the top synthesis level
    return OJPHGPU_OK;
  };
  for (const LevelBatch& b : d->batches) {
    const bool top = b.depth == 0;                          // the last level of its components (there may be two such launches)
    hipStream_t ls = (n_low && !top) ? d->side : s;
    if (top && n_low && !joined && (rc = finish_blocks()) != 0) return rc;
    const int sp = T.begin(SP_DWT, ls);
    if (b.img_first >= 0) {                                 // float->int / level shift applied in the stores
      ojphgpu_params pp = P.p; pp.reversible = b.rev ? 1 : 0;
      const ojphgpu_dwt_desc* idesc = (const ojphgpu_dwt_desc*)d->img_descs.p + b.img_first;
      rc = b.general ? ojphgpu_dwt_inverse_general_image(ls, &b.k, &pp, idesc, b.count, b.max_w, b.max_h, d_image, d->arena.p, container)
                     : ojphgpu_dwt_inverse_image_ex(ls, &pp, idesc, b.count, b.max_w, b.max_h, d_image, d->arena.p, container, b.nc == 3);
    } else if (b.general)
      rc = ojphgpu_dwt_inverse_general(ls, &b.k, (const ojphgpu_dwt_desc*)d->dwt_descs.p + b.first, b.count, b.max_w, b.max_h, d->arena.p);
    else
      rc = ojphgpu_dwt_inverse(ls, b.rev ? 1 : 0, (const ojphgpu_dwt_desc*)d->dwt_descs.p + b.first, b.count,
                               b.max_w, b.max_h, d->arena.p);
    if (rc) return rc;
    T.end(sp, ls);
  }
  if (n_low && !joined && (rc = finish_blocks()) != 0) return rc;
  if (d->need_convert) {
    const int sp = T.begin(SP_CONVERT, s);
    rc = ojphgpu_convert_inverse_ex(s, &P.p, (const ojphgpu_convert_desc*)d->conv_descs.p, d->tiles.count * d->nframes,
                                    d->conv_max_w, d->conv_max_h, d_image, d->arena.p, container);
    if (rc) return rc;
    T.end(sp, s);
  }
  T.finish(s);
  d->ran = true;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_decoder_set_timing(ojphgpu_decoder* d, int per_launch)
{
  if (!d) return OJPHGPU_E_INVALID;
  d->timer.detail = per_launch != 0;
  return OJPHGPU_OK;
}

// the run that has just been enqueued again, through the separate step 1 / step 2 launches (a fused launch asked for it)
int ojphgpu_decoder_repeat_separate(ojphgpu_decoder* d)
{
  if (!d || !d->last_image) return OJPHGPU_E_INVALID;
  d->force_separate = true;
  const int rc = ojphgpu_decoder_run_container(d, d->last_image, d->last_container);
  d->force_separate = false;
  d->fused_retries++;
  return rc;
}

extern "C" int ojphgpu_decoder_failed_blocks(ojphgpu_decoder* d, uint32_t* count)
{
  if (!d || !count || !d->ran) return OJPHGPU_E_INVALID;
  const size_t tail = ((size_t)d->nblocks + 3u) & ~(size_t)3u;
  std::vector<uint8_t> st(tail + 4);
  d->uncollected = false;
  const uint8_t* status = (const uint8_t*)(d->o_status ? d->o_status : d->status.p);
  for (int pass = 0; pass < 2; ++pass) {
    HIPCHK(hipMemcpyAsync(st.data(), status, st.size(), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    // a fused launch whose workers gave up waiting for their chains (the chip was held up for seconds by other work): its
    // blocks have no verdict yet -- the frame is decoded again through the separate launches, into the same image
    if (pass == 0 && d->last_fused) {
      const bool again = ojphgpu_fused_retry_wanted(st.data(), d->nblocks, d->fused_epoch);
      d->fused_outcome(again);
      if (again) {
        const int rc = ojphgpu_decoder_repeat_separate(d);
        if (rc) return rc;
        continue;
      }
    }
    break;
  }
  // (the stream is drained: every give-up of the runs enqueued so far has reached h_retry; the run collected here has been
  // repeated above if it was one of them, earlier uncollected ones are beyond repair -- ojphgpu_decoder_giveup_epoch tells)
  if (d->h_retry) d->retry_acked = *(volatile uint32_t*)d->h_retry;
  uint32_t n = 0;
  for (uint32_t i = 0; i < d->nblocks; ++i) n += st[i] != 0;
  *count = n;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_decoder_giveup_epoch(ojphgpu_decoder* d, uint32_t* last_giveup, uint32_t* current)
{
  if (!d || !last_giveup || !current) return OJPHGPU_E_INVALID;
  HIPCHK(hipStreamSynchronize(d->stream));
  *last_giveup = d->h_retry ? *(volatile uint32_t*)d->h_retry : 0u;
  *current = d->fused_epoch;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_decoder_fused_retries(ojphgpu_decoder* d, uint32_t* count)
{
  if (!d || !count) return OJPHGPU_E_INVALID;
  *count = d->fused_retries;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_decode(ojphgpu_decoder* d, const uint8_t* h_codestream, size_t len, int32_t* h_image)
{
  return decode_host(d, h_codestream, len, h_image, 32);
}

extern "C" int ojphgpu_decode16(ojphgpu_decoder* d, const uint8_t* h_codestream, size_t len, uint16_t* h_image)
{
  return decode_host(d, h_codestream, len, h_image, 16);
}

static int decode_host(ojphgpu_decoder* d, const uint8_t* h_codestream, size_t len, void* h_image, int container)
{
  if (!d || !h_image) return OJPHGPU_E_INVALID;
  const Plan& P = *d->P;
  if (d->nframes != 1) return OJPHGPU_E_INVALID;           // batches: upload_frame + run_device
  const size_t bytes = (size_t)P.frame_elems * 4;
  if (!d->image.p && d->image.alloc(bytes)) return OJPHGPU_E_NOMEM;
  int rc = ojphgpu_decoder_upload(d, h_codestream, len);
  if (rc) return rc;
  rc = ojphgpu_decoder_run_container(d, d->image.p, container);
  if (rc) return rc;
  uint32_t failed = 0;
  rc = ojphgpu_decoder_failed_blocks(d, &failed);          // first: it may repeat the run (see there), and the image is read after that
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h_image, d->image.p, bytes / (32 / container), hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  return failed ? OJPHGPU_E_BLOCK : OJPHGPU_OK;
}

extern "C" int ojphgpu_decoder_timing(ojphgpu_decoder* d, float out[4])
{
  if (!d || !out || !d->ran) return OJPHGPU_E_INVALID;
  float a = 0, b = 0, c = 0, r = 0;
  if (d->timer.read(SP_PREP, &a) || d->timer.read(SP_STEP1, &b) || d->timer.read(SP_STEP2, &c) || d->timer.read(SP_REFINE, &r) ||
      d->timer.read(SP_DWT, &out[1]) || d->timer.read(SP_CONVERT, &out[2]) || d->timer.read(-1, &out[3])) return OJPHGPU_E_HIP;
  out[0] = a + b + c + r;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_decoder_ht_timing(ojphgpu_decoder* d, float out[3])
{
  if (!d || !out || !d->ran) return OJPHGPU_E_INVALID;
  if (d->timer.read(SP_PREP, &out[0]) || d->timer.read(SP_STEP1, &out[1]) || d->timer.read(SP_STEP2, &out[2])) return OJPHGPU_E_HIP;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_decoder_level_timing(ojphgpu_decoder* d, float* out, uint32_t cap, uint32_t* n)
{
  if (!d || !out || !n || !d->ran) return OJPHGPU_E_INVALID;
  int k = d->timer.read_each(SP_DWT, out, cap);
  if (k < 0) return OJPHGPU_E_HIP;
  *n = (uint32_t)k;
  return OJPHGPU_OK;
}
