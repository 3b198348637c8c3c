// openjph_amd/csrc/ojphgpu_objects.h -- internals shared by the whole-frame codec objects
// (ojphgpu_codec.cpp) and the frame pipelines built on them (ojphgpu_pipe.cpp): device / pinned host
// buffers, launch batching, per-run event timing, and the encoder / decoder objects themselves.
#ifndef OJPHGPU_OBJECTS_H
#define OJPHGPU_OBJECTS_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ht_tables.h"
#include "ojph_plan.h"

using namespace ojphgpu;

namespace {

#define HIPCHK(x) do { if ((x) != hipSuccess) return OJPHGPU_E_HIP; } while (0)

struct DeviceBuf {
  void* p = nullptr; size_t n = 0;
  int alloc(size_t bytes) { n = bytes; return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? 0 : -1; }
  void release() { if (p) (void)hipFree(p); p = nullptr; }
};

// pinned host staging buffer (grows, never shrinks): D2H lands here at PCIe speed and without the
// page-fault cost of a fresh pageable allocation
struct HostBuf {
  uint8_t* p = nullptr; size_t cap = 0; bool pinned = false; unsigned uses = 0;
  // the first run of a codec object gets plain memory (pinning costs more than one pageable copy
  // saves); an object that is run again is a long-lived one and gets a pinned buffer
  int reserve(size_t bytes) {
    const bool want_pin = ++uses >= 2;
    if (bytes <= cap && (pinned || !want_pin)) return 0;
    release();
    void* q = nullptr;
    if (want_pin && hipHostMalloc(&q, bytes, hipHostMallocDefault) == hipSuccess) { p = (uint8_t*)q; pinned = true; }
    else { (void)hipGetLastError(); p = (uint8_t*)malloc(bytes); pinned = false; }
    cap = p ? bytes : 0;
    return p ? 0 : -1;
  }
  void release() { if (p) { if (pinned) (void)hipHostFree(p); else free(p); } p = nullptr; cap = 0; }
};

// One DWT launch: the levels `depth` steps below the top of their component, of the components with
// the same wavelet.  Without COCs that is one batch per resolution; with them a component may have
// fewer levels than another, or the other wavelet, and a depth splits in two.
// nc = 3: the batch holds the top levels of the three colour components of every tile, as triples, and the
// component transform is applied inside the DWT kernel (kernels_dwt.hip); group: 0 all components, 1 the colour
// components (0..2), 2 the others -- how a depth is split when the colour transform is fused
// general: levels of components that need the general lifting kernels (a Part-2 wavelet or decomposition, or 64-bit samples;
// kernels_dwt.hip's WvGen pipeline or kernels_lift.hip) -- k describes the level; components with equal k share a batch
struct LevelBatch { uint32_t first, count, max_w, max_h, depth; bool rev; int img_first; int nc; int group; bool general; ojphgpu_lift k;
                    std::vector<uint32_t> comps; };        // comps: the components of a general-lifting batch, in descriptor order

// DWT descriptors grouped so that one launch handles every tile-component
struct TileRange { uint32_t first, count; bool has(uint32_t t) const { return t >= first && t - first < count; } };

inline bool in_group(uint32_t comp, int group) { return group == 0 || (group == 1) == (comp < 3); }

// The component transform is applied inside the top DWT level of the three colour components (no conversion pass)
// when every one of them has such a level and no non-linearity sits between the samples and the transform
inline bool is_wide(const Plan& P, uint32_t comp) { return comp < P.wide.size() && P.wide[comp] != 0; }
// samples deeper than 26 bits are converted by the conversion kernels, not inside the top DWT level (whose conversion
// arithmetic was built and tested for the depths the 32-bit path had until round 4)
inline bool deep(const Plan& P, uint32_t comp) { return P.comps[comp].bit_depth > 26 || P.general(comp); }

// A component of the general lifting kernels (an ATK marker segment's wavelet) whose top level can take the conversion from /
// to the image samples itself, like the 5/3 and 9/7 launches do (kernels_dwt.hip: WvGen with an image side): 32-bit working
// samples, at most 26 bits deep, a top level that transforms both directions with at most four steps, no colour transform
// over it.  All or nothing per codestream (general_fused): the conversion kernels then have nothing left to do for them.
inline bool general_image_fusable(const Plan& P, uint32_t c)
{
  if (!P.general(c) || is_wide(P, c) || P.comps[c].bit_depth > 26 || P.recon_decomps(c) == 0) return false;
  if (P.p.color_transform && c < 3) return false;
  const ojphgpu_lift k = P.lift_of(c, P.skip_recon + 1);
  return k.horz && k.vert && k.num_steps >= 1 && k.num_steps <= 4 && k.elem != 1;
}
inline bool general_fused(const Plan& P);

inline bool colour_fused(const Plan& P)
{
  if (!P.p.color_transform || P.any_nlt3 || P.p.num_comps < 3) return false;
  for (uint32_t c = 0; c < 3; ++c) if (P.recon_decomps(c) == 0 || deep(P, c)) return false;
  const char* off = getenv("OJPHGPU_NO_COLOUR_FUSION");          // test switch: "1" keeps the stand-alone conversion kernels
  return !(off && off[0] && off[0] != '0');
}

inline bool general_fused(const Plan& P)
{
  const char* off = getenv("OJPHGPU_NO_GENERAL_FUSION");         // test switch: "1" keeps the stand-alone conversion kernels
  if ((off && off[0] && off[0] != '0') || P.any_nlt3 || (P.p.color_transform && !colour_fused(P))) return false;
  bool any = false;
  for (uint32_t c = 0; c < P.p.num_comps; ++c) {
    if (!P.general(c) || P.recon_decomps(c) == 0) continue;
    if (!general_image_fusable(P, c)) return false;
    any = true;
  }
  return any;
}

// gen_comp < 0: the levels of the components the two built-in wavelets' kernels transform; >= 0: of that component alone
template <typename F>
void for_levels_of(const Plan& P, TileRange tr, uint32_t depth, bool rev, int group, int gen_comp, F f)
{
  for (const ojphgpu_level_info& lv : P.levels) {
    if (!tr.has(lv.tile) || P.style(lv.comp).rev != rev || !in_group(lv.comp, group)) continue;
    if (gen_comp < 0 ? P.general(lv.comp) : (int)lv.comp != gen_comp) continue;
    const uint32_t L = P.recon_decomps(lv.comp);            // reduced-resolution decoding stops below the top levels
    if (L > depth && lv.res == L - depth) f(lv);
  }
}

uint32_t max_recon_decomps(const Plan& P)
{
  uint32_t m = 0;
  for (uint32_t c = 0; c < P.p.num_comps; ++c) m = std::max(m, P.recon_decomps(c));
  return m;
}

inline void push_level_desc(const ojphgpu_level_info& lv, std::vector<ojphgpu_dwt_desc>& descs, LevelBatch& b)
{
  ojphgpu_dwt_desc d; memset(&d, 0, sizeof(d));
  d.src_off = lv.src_off; d.ll_off = lv.ll_off; d.hl_off = lv.hl_off; d.lh_off = lv.lh_off; d.hh_off = lv.hh_off;
  d.src_pitch = lv.src_pitch; d.ll_pitch = lv.ll_pitch; d.hl_pitch = lv.hl_pitch; d.lh_pitch = lv.lh_pitch;
  d.hh_pitch = lv.hh_pitch; d.w = lv.w; d.h = lv.h; d.x_even = lv.x_even; d.y_even = lv.y_even;
  descs.push_back(d);
  b.count++; b.max_w = std::max(b.max_w, lv.w); b.max_h = std::max(b.max_h, lv.h);
}

void build_level_batches(const Plan& P, TileRange tr, std::vector<ojphgpu_dwt_desc>& descs, std::vector<LevelBatch>& batches)
{
  descs.clear(); batches.clear();
  const uint32_t depths = max_recon_decomps(P);
  const bool fused = colour_fused(P);
  ojphgpu_lift none; memset(&none, 0, sizeof(none));
  for (uint32_t depth = 0; depth < depths; ++depth) {
    for (int rev = 0; rev < 2; ++rev)
    for (int group = (fused && depth == 0) ? 1 : 0; group <= ((fused && depth == 0) ? 2 : 0); ++group) {
      LevelBatch b{ (uint32_t)descs.size(), 0, 0, 0, depth, rev != 0, -1, group == 1 ? 3 : 1, group, false, none };
      for_levels_of(P, tr, depth, rev != 0, group, -1, [&](const ojphgpu_level_info& lv) { push_level_desc(lv, descs, b); });
      if (b.count) batches.push_back(b);
    }
    for (uint32_t c = 0; c < P.p.num_comps; ++c) {            // components of the general lifting kernels: a batch each
      if (!P.general(c)) continue;
      const uint32_t L = P.recon_decomps(c);
      if (L <= depth) continue;
      // the level `depth` steps below the component's reconstructed top is decomposition level (skipped + depth + 1)
      LevelBatch b{ (uint32_t)descs.size(), 0, 0, 0, depth, P.style(c).rev, -1, 1, 0, true, P.lift_of(c, P.skip_recon + depth + 1) };
      for_levels_of(P, tr, depth, P.style(c).rev, 0, (int)c, [&](const ojphgpu_level_info& lv) { push_level_desc(lv, descs, b); });
      if (!b.count) continue;
      b.comps.push_back(c);
      // components that share a kernel and a kind of level (the usual case: one ATK for the whole codestream) share the launch
      LevelBatch* prev = batches.empty() ? nullptr : &batches.back();
      if (prev && prev->general && prev->depth == depth && prev->first + prev->count == b.first && memcmp(&prev->k, &b.k, sizeof(b.k)) == 0) {
        prev->count += b.count; prev->max_w = std::max(prev->max_w, b.max_w); prev->max_h = std::max(prev->max_h, b.max_h);
        prev->comps.push_back(c);
      } else batches.push_back(b);
    }
  }
}

// Descriptors of the top DWT level of every component with the un-decomposed plane addressed inside
// the image-sized component planes (for ojphgpu_dwt_forward_image / _inverse_image); batch.img_first
// points at them.  None when the fused path does not apply (colour transform).
void build_image_level_descs(const Plan& P, TileRange tr, const std::vector<ojphgpu_dwt_desc>& descs, std::vector<LevelBatch>& batches,
                             std::vector<ojphgpu_dwt_desc>& out)
{
  out.clear();
  if (P.any_nlt3 || (P.p.color_transform && !colour_fused(P))) return;   // those conversions live in the conversion kernels
  const bool gfused = general_fused(P);
  auto image_desc = [&](const ojphgpu_level_info& lv, ojphgpu_dwt_desc d) {
    const TileComp& tc = P.tcomps[P.tiles[lv.tile].comps[lv.comp]];
    const CompGeo& g = P.comps[lv.comp];
    const Rect& rr = P.ress[tc.res[lv.res]].r;              // the tile-component at the reconstructed resolution
    d.src_off = g.frame_off + (uint64_t)(rr.y0 - g.y0) * g.w + (rr.x0 - g.x0);
    d.src_pitch = g.w;
    d.reserved = g.bit_depth | (g.is_signed ? 0x100u : 0u);   // the component's sample format for the fused conversion
    return d;
  };
  for (LevelBatch& b : batches) {
    if (b.depth == 0 && b.count && b.general && gfused) {    // the general lifting kernels' top level, conversion included
      b.img_first = (int)out.size();
      size_t k = 0;
      for (uint32_t c : b.comps)
        for_levels_of(P, tr, 0, P.style(c).rev, 0, (int)c, [&](const ojphgpu_level_info& lv) { out.push_back(image_desc(lv, descs[b.first + k++])); });
      continue;
    }
    if (b.depth != 0 || b.count == 0 || b.general) continue;
    bool all_shallow = true;                                  // (a batch with a deep component keeps the conversion kernels)
    for_levels_of(P, tr, 0, b.rev, b.group, -1, [&](const ojphgpu_level_info& lv) { all_shallow = all_shallow && !deep(P, lv.comp); });
    if (!all_shallow) continue;
    b.img_first = (int)out.size();
    size_t k = 0;
    for_levels_of(P, tr, 0, b.rev, b.group, -1, [&](const ojphgpu_level_info& lv) {
      ojphgpu_dwt_desc d = descs[b.first + k++];
      const TileComp& tc = P.tcomps[P.tiles[lv.tile].comps[lv.comp]];
      const CompGeo& g = P.comps[lv.comp];
      const Rect& rr = P.ress[tc.res[lv.res]].r;              // the tile-component at the reconstructed resolution
      d.src_off = g.frame_off + (uint64_t)(rr.y0 - g.y0) * g.w + (rr.x0 - g.x0);
      d.src_pitch = g.w;
      d.reserved = g.bit_depth | (g.is_signed ? 0x100u : 0u);   // the component's sample format for the fused conversion
      out.push_back(d);
    });
  }
}

// One descriptor per tile and component, in that order; a component whose conversion is fused into
// its top DWT level (no colour transform, at least one level) gets an empty one.  Returns whether
// any component is left for the conversion kernels.
bool build_convert_descs(const Plan& P, TileRange tr, std::vector<ojphgpu_convert_desc>& descs, uint32_t& max_w, uint32_t& max_h)
{
  descs.clear(); max_w = max_h = 0;
  bool any = false;
  for (const Tile& t : P.tiles) {
    if (!tr.has(t.idx)) continue;
    for (uint32_t c = 0; c < P.p.num_comps; ++c) {
      const uint32_t L = P.recon_decomps(c);
      const TileComp& tc = P.tcomps[t.comps[c]];
      const Resolution& R = P.ress[tc.res[L]];
      ojphgpu_convert_desc d; memset(&d, 0, sizeof(d));
      if (L == 0) { const Band& B = P.bands[(size_t)R.band[0]]; d.plane_off = B.plane_off; d.pitch = B.pitch; }
      else { d.plane_off = R.plane_off; d.pitch = R.pitch; }
      const CompGeo& g = P.comps[c];
      d.src_x0 = R.r.x0 - g.x0; d.src_y0 = R.r.y0 - g.y0;
      d.img_pitch = g.w; d.img_off = g.frame_off;
      d.fmt = g.bit_depth | (g.is_signed ? 0x100u : 0u) | 0x200u | (P.style(c).rev ? 0x400u : 0u) | (P.nlt3[c] ? 0x800u : 0u) |   // 0x200: bit 10 says which conversion
              (is_wide(P, c) ? 0x1000u : 0u);
      // (a batch of the top DWT level converts its components itself only when none of them is deep: the batch of the
      // component's wavelet, see build_image_level_descs)
      bool batch_deep = false;
      const int grp = colour_fused(P) ? (c < 3 ? 1 : 2) : 0;
      for (uint32_t o = 0; o < P.p.num_comps; ++o)
        batch_deep = batch_deep || (P.style(o).rev == P.style(c).rev && in_group(o, grp) && !P.general(o) && deep(P, o) && P.recon_decomps(o) > 0);
      if ((P.p.color_transform && !colour_fused(P)) || P.any_nlt3 || L == 0 || (P.general(c) && !general_fused(P)) || batch_deep) { d.w = R.r.w; d.h = R.r.h; any |= d.w && d.h; }
      descs.push_back(d);
      max_w = std::max(max_w, d.w); max_h = std::max(max_h, d.h);
    }
  }
  return any;
}

// Frame batches: the same plan applied to `nframes` independent frames in one set of launches
// (config C5: a batch of independent 4K frames).  Frame f lives f * arena_elems further in the arena
// and f * frame_elems further in the image buffer; descriptors are simply replicated, batch by batch.
void replicate_levels(std::vector<ojphgpu_dwt_desc>& descs, std::vector<ojphgpu_dwt_desc>& img_descs, std::vector<LevelBatch>& batches,
                      uint32_t nframes, uint64_t arena_elems, uint64_t frame_elems)
{
  if (nframes <= 1) return;
  std::vector<ojphgpu_dwt_desc> out, iout; std::vector<LevelBatch> nb;
  for (const LevelBatch& b : batches) {
    LevelBatch n = b;
    n.first = (uint32_t)out.size(); n.count = b.count * nframes;
    if (b.img_first >= 0) n.img_first = (int)iout.size();
    for (uint32_t f = 0; f < nframes; ++f)
      for (uint32_t i = 0; i < b.count; ++i) {
        const uint64_t o = (uint64_t)f * arena_elems;
        ojphgpu_dwt_desc d = descs[b.first + i];
        d.src_off += o; d.ll_off += o; d.hl_off += o; d.lh_off += o; d.hh_off += o;
        out.push_back(d);
        if (b.img_first >= 0) {
          d = img_descs[(size_t)b.img_first + i];
          d.src_off += (uint64_t)f * frame_elems; d.ll_off += o; d.hl_off += o; d.lh_off += o; d.hh_off += o;
          iout.push_back(d);
        }
      }
    nb.push_back(n);
  }
  descs.swap(out); img_descs.swap(iout); batches.swap(nb);
}

void replicate_converts(std::vector<ojphgpu_convert_desc>& descs, uint32_t nframes, uint64_t arena_elems, uint64_t frame_elems)
{
  if (nframes <= 1 || descs.empty()) return;
  const size_t n = descs.size();
  for (uint32_t f = 1; f < nframes; ++f)
    for (size_t i = 0; i < n; ++i) {
      ojphgpu_convert_desc d = descs[i];
      d.plane_off += (uint64_t)f * arena_elems; d.img_off += (uint64_t)f * frame_elems;
      descs.push_back(d);
    }
}

// plan-order indices of the code-blocks that belong to the tile range
std::vector<uint32_t> blocks_of_tiles(const Plan& P, TileRange tr)
{
  std::vector<uint32_t> ids;
  for (size_t i = 0; i < P.blocks.size(); ++i) {
    const Band& B = P.bands[P.blocks[i].band];
    if (tr.has(B.tile) && B.res <= P.top_read_res(B.comp)) ids.push_back((uint32_t)i);   // resolutions above are not decoded: their bands stay zero
  }
  return ids;
}

// Timing of one run_device: every launch (or group of launches) is bracketed by a pair of HIP events
// on the stream it is issued on -- launches of one run may sit on two streams -- and tagged with a
// kind; per-kind sums and the wall time of the whole run are read back afterwards.
struct Spans {
  struct Span { hipEvent_t a, b; int kind; };
  std::vector<Span> pool;            // events are created once and reused
  size_t used = 0;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool ok = false;
  bool detail = true;                // per-launch spans on; off = only the wall time of the run (two events)
  int init() { ok = hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess; return ok ? 0 : -1; }
  void destroy() {
    for (Span& x : pool) { (void)hipEventDestroy(x.a); (void)hipEventDestroy(x.b); }
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    pool.clear();
  }
  void start(hipStream_t s) { used = 0; if (ok) (void)hipEventRecord(t0, s); }
  void finish(hipStream_t s) { if (ok) (void)hipEventRecord(t1, s); }
  int begin(int kind, hipStream_t s) {
    if (!ok || !detail) return -1;
    if (used == pool.size()) {
      Span x{ nullptr, nullptr, kind };
      if (hipEventCreate(&x.a) != hipSuccess || hipEventCreate(&x.b) != hipSuccess) { ok = false; return -1; }
      pool.push_back(x);
    }
    pool[used].kind = kind;
    (void)hipEventRecord(pool[used].a, s);
    return (int)used++;
  }
  void end(int id, hipStream_t s) { if (ok && id >= 0) (void)hipEventRecord(pool[(size_t)id].b, s); }
  // sum of the spans of `kind`; kind < 0: wall time of the run
  int read(int kind, float* out) {
    if (!ok || hipEventSynchronize(t1) != hipSuccess) return -1;
    if (kind < 0) return hipEventElapsedTime(out, t0, t1) == hipSuccess ? 0 : -1;
    float sum = 0;
    for (size_t i = 0; i < used; ++i)
      if (pool[i].kind == kind) { float ms = 0; if (hipEventElapsedTime(&ms, pool[i].a, pool[i].b) != hipSuccess) return -1; sum += ms; }
    *out = sum;
    return 0;
  }
  // the individual spans of `kind`, in issue order
  int read_each(int kind, float* out, uint32_t cap) {
    if (!ok || hipEventSynchronize(t1) != hipSuccess) return -1;
    int n = 0;
    for (size_t i = 0; i < used; ++i)
      if (pool[i].kind == kind && (uint32_t)n < cap) { if (hipEventElapsedTime(&out[n], pool[i].a, pool[i].b) != hipSuccess) return -1; ++n; }
    return n;
  }
};
enum { SP_CONVERT = 0, SP_DWT = 1, SP_HT_ENC = 2, SP_PREP = 3, SP_STEP1 = 4, SP_STEP2 = 5, SP_REFINE = 6 };

}  // namespace

struct ojphgpu_encoder {
  const ojphgpu_plan* handle = nullptr;
  const Plan* P = nullptr;
  int device = 0; hipStream_t stream = nullptr;
  DeviceBuf arena, image, dwt_descs, img_descs, cb_descs, conv_descs, scratch, out, results, counters;
  bool need_convert = false;                       // some component is not converted inside its top DWT level
  std::vector<LevelBatch> batches;
  uint32_t conv_max_w = 0, conv_max_h = 0, out_cap = 0;
  TileRange tiles{ 0, 0 };
  uint32_t nframes = 1;                            // frames coded per run_device (batch)
  bool fetched = false;                            // results / bytes of the last run are on the host
  uint64_t nbytes = 0;
  std::vector<uint32_t> block_ids;                 // plan-order index of each block this encoder codes (per frame)
  std::vector<ojphgpu_cb_result> h_results;
  HostBuf h_out, h_res;
  Spans timer;
  bool ran = false;
  // overlap of the block coder with the lower DWT levels: the blocks of the top resolution (3/4 of
  // the samples) only need the first DWT level, so they are coded on a second stream while the
  // small, latency-bound launches of levels 2..L run on the main one
  uint32_t n_top = 0;                              // descriptors [0, n_top) = blocks of the top resolution
  int widths_top = 0, widths_rest = 0;             // which block encoder kernels each range needs (bit 0 narrow, bit 1 wide, bit 2 reversible, bit 3 irreversible)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // where a run writes its products: the object's own buffers (null), or -- for a frame pipeline that keeps
  // several frames in flight -- the buffers of the frame's slot (ojphgpu_pipe.cpp)
  void* o_out = nullptr; void* o_results = nullptr; void* o_counters = nullptr;
  // The compacted output is split into nreg regions with a cursor each (block i allocates in region i % nreg), so
  // that the blocks' allocation atomics do not all queue on one cache line (claim_output, kernels_ht_enc.hip).
  // h_regions[2r] = first byte, [2r + 1] = capacity; counters: word 32 r = cursor of region r, word 1 = status.
  uint32_t nreg = 0;
  std::vector<uint32_t> h_regions, h_cursors;
  DeviceBuf regions;
  size_t counters_bytes = 16;
};
// the device part of an encode: d_image holds the frame in `container`-bit elements (32 / 16)
int ojphgpu_encoder_run_container(ojphgpu_encoder* e, const void* d_image, int container);

// A block the reference decodes from bytes the codestream does not hold (Plan::padded: its tile-part ended early and
// bb_read_chunk, ojph_bitbuffer_read.h:134-150, handed over zeros for the rest): `got` bytes at `src` of the codestream,
// then zeros up to `total`, placed at `dst` -- an offset into the frame's part of the device data buffer, BEHIND the byte
// range that is uploaded as it is.
struct PadCopy { uint64_t src, dst; uint32_t got, total; };
struct ojphgpu_decoder {
  const Plan* P = nullptr;
  int device = 0; hipStream_t stream = nullptr;
  DeviceBuf arena, image, dwt_descs, img_descs, cb_descs, conv_descs, data, status, quads, aux;
  DeviceBuf fstate; uint32_t fused_epoch = 0, max_block_h = 0;     // the fused step 1 + step 2 launch: its flags / per-block state, run counter
  uint32_t cus = 256;                               // compute units of `device` (the fused launch is shaped for them)
  // A fused launch whose workers gave up waiting (a chip held up for seconds by other work) marks the run in the RETRY word
  // behind the block status array; whoever collects the run's verdicts (ojphgpu_decoder_failed_blocks, the decoder pipe)
  // then repeats the run through the separate launches: last_* is what that repeat needs.
  bool last_fused = false, force_separate = false;
  void* last_image = nullptr; int last_container = 0;
  uint32_t fused_retries = 0;                       // runs repeated that way so far
  // A caller of ojphgpu_decoder_run_device that never collects its runs would never learn that one asked for a repeat: the
  // fused launch also writes the run's epoch into h_retry -- a word of mapped host memory (d_h_retry: its device address) --
  // and the next run of this object finds it there (OJPHGPU_E_UNCOLLECTED).  uncollected: a fused run has been enqueued
  // and nobody has read its verdicts yet.
  uint32_t* h_retry = nullptr; uint32_t* d_h_retry = nullptr; bool uncollected = false;
  uint32_t retry_acked = 0;                         // the newest epoch found in h_retry that has been reported or collected
  // eight repeats in a row and the object stops using the one launch: a wait that runs out costs seconds, and a chip (or a
  // device layout) on which it keeps running out is better served by the separate launches than by trying again
  uint32_t fused_strikes = 0; bool fused_off = false;
  void fused_outcome(bool repeated) { if (!repeated) fused_strikes = 0; else if (++fused_strikes >= 8u) fused_off = true; }
  // descriptors [0, n_low) = blocks below the top resolution (0 = no overlap of the lower synthesis
  // levels with step 2, see decoder_create)
  uint32_t n_low = 0;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool need_convert = false;                       // some component is not converted inside its top DWT level
  TileRange tiles{ 0, 0 };
  uint32_t nframes = 1;
  bool any_refine = false;                         // some block carries SigProp / MagRef passes
  int kinds = 0;                                   // block kinds of the frame(s), for ht_decode_step2_launch
  std::vector<size_t> f_first, f_len, f_base;      // per frame: codestream byte range uploaded, its place in `data`
  std::vector<std::vector<PadCopy>> f_pads;        // per frame: blocks uploaded with zeros behind their bytes (PadCopy)
  uint32_t nblocks = 0;                            // code-blocks of the tile range (all frames)
  std::vector<LevelBatch> batches;
  uint32_t conv_max_w = 0, conv_max_h = 0, max_len1 = 0;
  size_t data_first = 0, data_len = 0;              // byte range of the codestream holding this range's block data
  Spans timer;
  bool ran = false;
  std::vector<uint32_t> block_ids;                 // plan-order index of each block descriptor (per frame)
  // what a run reads: the object's own buffers (null), or those of a frame pipeline's slot
  const void* o_cb_descs = nullptr; const void* o_data = nullptr; void* o_status = nullptr;
};
struct DecFrameInfo {
  uint64_t first = 0, len = 0; bool any_refine = false; uint32_t max_len1 = 0; int kinds = 0;   // kinds: see ht_decode_step2_launch; bit 5: blocks on the 64-bit sample path
  std::vector<PadCopy> pads; uint64_t pad_len = 0;         // padded blocks: their copies, the bytes they take behind `len` (rounded up to 64)
  uint64_t data_bytes() const { return ((len + 63) & ~(uint64_t)63) + pad_len; }
};
// enqueues the copies of a frame's padded blocks (d_frame_data = where the frame's byte range starts in device memory)
int ojphgpu_decoder_upload_pads(hipStream_t s, uint8_t* d_frame_data, const uint8_t* h_codestream, size_t cs_len, const std::vector<PadCopy>& pads);
int  ojphgpu_same_frame_geometry(const Plan& P, const Plan& Q, bool compare_blocks);
void ojphgpu_decoder_fill_descs(const Plan& P, const Plan& Q, const std::vector<uint32_t>& ids, uint64_t arena_off,
                                uint64_t data_base, ojphgpu_cb_desc* bd, DecFrameInfo& fi);
int  ojphgpu_decoder_run_container(ojphgpu_decoder* d, void* d_image, int container);
// after a run has completed, with the status bytes + the RETRY word behind them on the host (nblocks bytes, then the word at
// the next multiple of 4): did the fused launch of that run (epoch) ask for a repeat?
inline bool ojphgpu_fused_retry_wanted(const uint8_t* h_status, uint32_t nblocks, uint32_t epoch)
{
  uint32_t w; memcpy(&w, h_status + ((nblocks + 3u) & ~3u), 4);
  return w == epoch;
}


#endif
