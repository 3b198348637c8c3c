// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" facade over the *real* reference (aous72/OpenJPH 0.31.0), linked against objects
// compiled from /root/reference by oracle/Makefile.  It lets the Python tests / bench drive the
// reference through ctypes:
//   * whole-codestream encode/decode through the public ojph::codestream API
//     (reference: src/core/openjph/ojph_codestream.h:88-383), and
//   * single code-block HT encode / decode through the internal kernels
//     (reference: src/core/coding/ojph_block_encoder.h:52, ojph_block_decoder.h:53).
// Nothing in the product (openjph_amd/) may link or load this file.

#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "ojph_arch.h"
#include "ojph_mem.h"
#include "ojph_file.h"
#include "ojph_params.h"
#include "ojph_codestream.h"
#include "ojph_message.h"
#include "ojph_block_encoder.h"
#include "ojph_block_decoder.h"

namespace ojph { namespace local {
  // reference: src/core/coding/ojph_block_encoder.cpp:258 (+ _avx2 / _avx512 variants)
  bool initialize_block_encoder_tables();
#ifndef OJPH_DISABLE_SIMD
  bool initialize_block_encoder_tables_avx2();
  bool initialize_block_encoder_tables_avx512();
#endif
}}

extern "C" {

struct ref_params {
  uint32_t width, height, num_comps;
  uint32_t bit_depth, is_signed;
  uint32_t reversible, num_decomps, block_w, block_h;
  uint32_t color_transform;      // 0/1
  uint32_t tile_w, tile_h;       // 0 => single tile
  uint32_t prog_order;           // 0 LRCP 1 RLCP 2 RPCL 3 PCRL 4 CPRL
  uint32_t planar;               // 0 interleaved lines, 1 planar
  float    qstep;                // irreversible base step; <=0 => library default
  uint32_t precinct_w, precinct_h; // 0 => default (32768); applied to all resolutions
  uint32_t tlm;                  // request TLM marker
  uint8_t  precinct_exps[36];    // per resolution (0 = lowest): PPx | PPy << 4; all 0 => precinct_w/h for every resolution
  uint32_t image_x0, image_y0;   // image offset; width/height are the image SIZE (extent = offset + size)
  uint32_t tile_x0, tile_y0;     // tile offset
  uint8_t  comp_dx[16], comp_dy[16];   // sub-sampling of component c < 16; 0 => 1
  uint32_t tilepart_div;         // bit 0: tile-parts at resolutions, bit 1: at components
  uint8_t  comp_depth[16], comp_sign[16];   // per component where it differs: depth (0 = bit_depth), 1 unsigned / 2 signed (0 = is_signed)
  uint32_t qfactor;              // 0 = not set
  // COC marker segments (param_cod's comp_idx setters).  Entry k describes the k-th component that gets
  // one, in creation order; mask says which setters are called (1 decompositions, 2 block size,
  // 4 precincts, 8 reversible) -- what is not set keeps the library's COC defaults
  struct { uint8_t comp, mask, reversible, num_decomps, log_bw, log_bh, pad[2]; uint8_t precinct_exps[36]; } coc[16];
  uint32_t num_coc;
  // param_nlt::set_nonlinear_transform calls, in order: component (65535 = ALL_COMPS) and type
  struct { uint16_t comp; uint8_t type, pad; } nlt[17];
  uint32_t num_nlt;
  // param_qcd::set_qfactor(comp, ctype, qfactor) calls, in order
  struct { uint8_t comp, ctype, qfactor, pad; } cqf[16];
  uint32_t num_cqf;
};

static const char* po_names[5] = { "LRCP", "RLCP", "RPCL", "PCRL", "CPRL" };

int ref_simd_level(void)
{
#ifdef OJPH_DISABLE_SIMD
  return -1;
#else
  return ojph::get_cpu_ext_level();
#endif
}

// planes: num_comps pointers to int32 rows (width*height each).  Returns codestream length or
// a negative number on error / insufficient capacity (-needed).
long ref_encode_ex(const ref_params* p, const int32_t* const* planes, uint8_t* out, long out_cap,
                   const char* profile, const char* com);

long ref_encode(const ref_params* p, const int32_t* const* planes, uint8_t* out, long out_cap)
{
  return ref_encode_ex(p, planes, out, out_cap, NULL, NULL);
}

// profile: NULL / "IMF" / "BROADCAST" (codestream::set_profile); com: NULL or a user COM string
long ref_encode_ex(const ref_params* p, const int32_t* const* planes, uint8_t* out, long out_cap,
                   const char* profile, const char* com)
{
  try {
    ojph::set_message_level(getenv("REF_SHIM_VERBOSE") ? ojph::OJPH_MSG_ALL_MSG : ojph::OJPH_MSG_NO_MSG);
    ojph::codestream cs;
    ojph::param_siz siz = cs.access_siz();
    siz.set_image_extent(ojph::point(p->image_x0 + p->width, p->image_y0 + p->height));
    siz.set_num_components(p->num_comps);
    for (uint32_t c = 0; c < p->num_comps; ++c) {
      ojph::point ds(c < 16 && p->comp_dx[c] ? p->comp_dx[c] : 1, c < 16 && p->comp_dy[c] ? p->comp_dy[c] : 1);
      const uint32_t bd = c < 16 && p->comp_depth[c] ? p->comp_depth[c] : p->bit_depth;
      const bool sg = c < 16 && p->comp_sign[c] ? p->comp_sign[c] == 2 : p->is_signed != 0;
      siz.set_component(c, ds, bd, sg);
    }
    siz.set_image_offset(ojph::point(p->image_x0, p->image_y0));
    siz.set_tile_offset(ojph::point(p->tile_x0, p->tile_y0));
    if (p->tile_w && p->tile_h)
      siz.set_tile_size(ojph::size(p->tile_w, p->tile_h));
    ojph::param_cod cod = cs.access_cod();
    cod.set_num_decomposition(p->num_decomps);
    cod.set_block_dims(p->block_w, p->block_h);
    cod.set_progression_order(po_names[p->prog_order % 5]);
    cod.set_color_transform(p->color_transform != 0);
    cod.set_reversible(p->reversible != 0);
    bool per_res = false;
    for (uint32_t i = 0; i <= p->num_decomps && i < 36; ++i) per_res |= p->precinct_exps[i] != 0;
    if (per_res) {
      std::vector<ojph::size> ps;
      for (uint32_t i = 0; i <= p->num_decomps; ++i)
        ps.push_back(ojph::size(1u << (p->precinct_exps[i] & 15), 1u << (p->precinct_exps[i] >> 4)));
      cod.set_precinct_size((int)ps.size(), ps.data());
    } else if (p->precinct_w && p->precinct_h) {
      std::vector<ojph::size> ps(p->num_decomps + 1, ojph::size(p->precinct_w, p->precinct_h));
      cod.set_precinct_size((int)ps.size(), ps.data());
    }
    for (uint32_t k = 0; k < p->num_coc && k < 16; ++k) {
      const auto& q = p->coc[k];
      if (q.mask & 1) cod.set_num_decomposition(q.comp, q.num_decomps);
      if (q.mask & 2) cod.set_block_dims(q.comp, 1u << q.log_bw, 1u << q.log_bh);
      if (q.mask & 4) {
        std::vector<ojph::size> ps;                          // pad[0] = number of sizes given (the library repeats the last one)
        for (uint32_t i = 0; i < q.pad[0]; ++i) ps.push_back(ojph::size(1u << (q.precinct_exps[i] & 15), 1u << (q.precinct_exps[i] >> 4)));
        cod.set_precinct_size(q.comp, (int)ps.size(), ps.data());
      }
      if (q.mask & 8) cod.set_reversible(q.comp, q.reversible != 0);
    }
    if (p->qstep > 0.0f && (!p->reversible || p->num_coc))
      cs.access_qcd().set_irrev_quant(p->qstep);
    if (p->qfactor) cs.access_qcd().set_qfactor((ojph::ui8)p->qfactor);
    for (uint32_t k = 0; k < p->num_cqf && k < 16; ++k)
      cs.access_qcd().set_qfactor(p->cqf[k].comp, ojph::param_qcd::ui8_2_comp_type(p->cqf[k].ctype), p->cqf[k].qfactor);
    for (uint32_t k = 0; k < p->num_nlt && k < 17; ++k)
      cs.access_nlt().set_nonlinear_transform(p->nlt[k].comp, p->nlt[k].type);
    cs.set_planar(p->planar != 0);
    if (p->tlm) cs.request_tlm_marker(true);
    if (p->tilepart_div) cs.set_tilepart_divisions((p->tilepart_div & 1) != 0, (p->tilepart_div & 2) != 0);

    if (profile && profile[0]) cs.set_profile(profile);
    ojph::mem_outfile mf;
    mf.open();
    ojph::comment_exchange ce;
    if (com) ce.set_string(com);
    cs.write_headers(&mf, com ? &ce : NULL, com ? 1 : 0);

    // the library says which component's line it wants next (planar, interleaved, sub-sampled alike)
    std::vector<uint32_t> row(p->num_comps, 0), cw(p->num_comps), ch(p->num_comps);
    uint64_t lines = 0;
    for (uint32_t c = 0; c < p->num_comps; ++c) {
      cw[c] = siz.get_recon_width(c); ch[c] = siz.get_recon_height(c); lines += ch[c];
    }
    // (interleaved exchange asks for every component on every line of component 0,
    // ojph_codestream_local.cpp:1208-1219; with components of different height the library would be
    // handed a row it cannot place and spin in exchange() -- such a request is refused here)
    ojph::ui32 next_comp = 0;
    ojph::line_buf* line = cs.exchange(NULL, next_comp);
    for (uint64_t i = 0; i < lines; ++i) {
      const uint32_t c = next_comp;
      if (line == NULL || c >= p->num_comps || row[c] >= ch[c]) { cs.close(); return 0; }
      memcpy(line->i32, planes[c] + (size_t)row[c] * cw[c], sizeof(int32_t) * cw[c]);
      row[c]++;
      line = cs.exchange(line, next_comp);
    }
    cs.flush();
    long len = (long)mf.tell();
    if (len > out_cap) { cs.close(); return -len; }
    memcpy(out, mf.get_data(), (size_t)len);
    cs.close();
    return len;
  } catch (const std::exception&) {
    return 0;
  }
}

// Decodes to int32 planes (caller allocates num_comps * width * height).  Fills info[0..5] =
// width,height,num_comps,bit_depth,is_signed,reversible.  Returns 0 on success.
int ref_decode_skip(const uint8_t* data, long len, int32_t* const* planes, uint32_t* info, int resilient,
                    uint32_t skip_read, uint32_t skip_recon);

int ref_decode(const uint8_t* data, long len, int32_t* const* planes, uint32_t* info, int resilient)
{
  return ref_decode_skip(data, len, planes, info, resilient, 0, 0);
}

int ref_decode_skip(const uint8_t* data, long len, int32_t* const* planes, uint32_t* info, int resilient,
                    uint32_t skip_read, uint32_t skip_recon)
{
  try {
    ojph::set_message_level(getenv("REF_SHIM_VERBOSE") ? ojph::OJPH_MSG_ALL_MSG : ojph::OJPH_MSG_NO_MSG);
    ojph::mem_infile in;
    in.open(data, (size_t)len);
    ojph::codestream cs;
    if (resilient) cs.enable_resilience();
    cs.read_headers(&in);
    if (skip_read || skip_recon) cs.restrict_input_resolution(skip_read, skip_recon);
    ojph::param_siz siz = cs.access_siz();
    uint32_t nc = siz.get_num_components();
    uint32_t w = siz.get_recon_width(0), h = siz.get_recon_height(0);
    if (info) {
      info[0] = w; info[1] = h; info[2] = nc; info[3] = siz.get_bit_depth(0);
      info[4] = siz.is_signed(0) ? 1 : 0; info[5] = cs.access_cod().is_reversible() ? 1 : 0;
    }
    if (info) {                      // per component (up to 16): recon width, height from info[8] on
      for (uint32_t c = 0; c < nc && c < 16; ++c) { info[8 + 2 * c] = siz.get_recon_width(c); info[9 + 2 * c] = siz.get_recon_height(c); }
    }
    if (planes == NULL) { cs.close(); return 0; }
    cs.create();
    std::vector<uint32_t> row(nc, 0), cw(nc), chh(nc);
    uint64_t lines = 0;
    for (uint32_t c = 0; c < nc; ++c) { cw[c] = siz.get_recon_width(c); chh[c] = siz.get_recon_height(c); lines += chh[c]; }
    for (uint64_t i = 0; i < lines; ++i) {
      ojph::ui32 cn;
      ojph::line_buf* l = cs.pull(cn);
      if (l == NULL || cn >= nc || row[cn] >= chh[cn]) { cs.close(); return -3; }
      memcpy(planes[cn] + (size_t)row[cn] * cw[cn], l->i32, sizeof(int32_t) * cw[cn]);
      row[cn]++;
    }
    cs.close();
    return 0;
  } catch (const std::exception&) {
    return -1;
  } catch (const char*) {
    return -2;
  }
}

// variant: 0 = generic C++, 1 = avx2, 2 = avx512 (SIMD build only)
long ref_encode_block32(int variant, uint32_t* buf, uint32_t missing_msbs, uint32_t width,
                        uint32_t height, uint32_t stride, uint8_t* out, long out_cap)
{
  try {
    static bool init = false;
    if (!init) {
      ojph::local::initialize_block_encoder_tables();
#ifndef OJPH_DISABLE_SIMD
      ojph::local::initialize_block_encoder_tables_avx2();
      ojph::local::initialize_block_encoder_tables_avx512();
#endif
      init = true;
    }
    ojph::mem_elastic_allocator elastic(1048576);
    ojph::coded_lists* coded = NULL;
    ojph::ui32 lengths[2] = { 0, 0 };
    switch (variant) {
#ifndef OJPH_DISABLE_SIMD
      case 1:
        ojph::local::ojph_encode_codeblock_avx2(buf, missing_msbs, 1, width, height, stride,
                                                lengths, &elastic, coded);
        break;
      case 2:
        ojph::local::ojph_encode_codeblock_avx512(buf, missing_msbs, 1, width, height, stride,
                                                  lengths, &elastic, coded);
        break;
#endif
      default:
        ojph::local::ojph_encode_codeblock32(buf, missing_msbs, 1, width, height, stride,
                                             lengths, &elastic, coded);
    }
    if ((long)lengths[0] > out_cap) return -(long)lengths[0];
    memcpy(out, coded->buf, lengths[0]);
    return (long)lengths[0];
  } catch (const std::exception&) {
    return 0;
  }
}

// coded must have 8 readable bytes before and 16 after (ojph_codeblock.h:123-124).
int ref_decode_block32(int variant, const uint8_t* coded, uint32_t len1, uint32_t len2,
                       uint32_t missing_msbs, uint32_t num_passes, uint32_t width,
                       uint32_t height, uint32_t stride, uint32_t* out, int stripe_causal)
{
  std::vector<uint8_t> padded((size_t)len1 + len2 + 64, 0);
  memcpy(padded.data() + 16, coded, (size_t)len1 + len2);
  bool ok;
  try {
#ifndef OJPH_DISABLE_SIMD
    if (variant == 1)
      ok = ojph::local::ojph_decode_codeblock_avx2(padded.data() + 16, out, missing_msbs, num_passes,
                                                   len1, len2, width, height, stride,
                                                   stripe_causal != 0);
    else
#endif
      ok = ojph::local::ojph_decode_codeblock32(padded.data() + 16, out, missing_msbs, num_passes,
                                                len1, len2, width, height, stride,
                                                stripe_causal != 0);
  } catch (const std::exception&) { return -1; }
  (void)variant;
  return ok ? 0 : 1;
}

// the 64-bit sample path (ojph_block_encoder.cpp:1026, ojph_block_decoder64.cpp:766); generic C++ only in the reference
long ref_encode_block64(uint64_t* buf, uint32_t missing_msbs, uint32_t width, uint32_t height, uint32_t stride,
                        uint8_t* out, long out_cap)
{
  try {
    static bool init = false;
    if (!init) { ojph::local::initialize_block_encoder_tables(); init = true; }
    ojph::mem_elastic_allocator elastic(1048576);
    ojph::coded_lists* coded = NULL;
    ojph::ui32 lengths[2] = { 0, 0 };
    ojph::local::ojph_encode_codeblock64(buf, missing_msbs, 1, width, height, stride, lengths, &elastic, coded);
    if ((long)lengths[0] > out_cap) return -(long)lengths[0];
    memcpy(out, coded->buf, lengths[0]);
    return (long)lengths[0];
  } catch (const std::exception&) {
    return 0;
  }
}

int ref_decode_block64(const uint8_t* coded, uint32_t len1, uint32_t len2, uint32_t missing_msbs, uint32_t num_passes,
                       uint32_t width, uint32_t height, uint32_t stride, uint64_t* out, int stripe_causal)
{
  std::vector<uint8_t> padded((size_t)len1 + len2 + 64, 0);
  memcpy(padded.data() + 16, coded, (size_t)len1 + len2);
  bool ok;
  try {
    ok = ojph::local::ojph_decode_codeblock64(padded.data() + 16, out, missing_msbs, num_passes, len1, len2, width, height,
                                              stride, stripe_causal != 0);
  } catch (const std::exception&) { return -1; }
  return ok ? 0 : 1;
}

// 0: the library's messages are swallowed (the block-level entry points do not set a level of their own)
void ref_set_verbose(int on) { ojph::set_message_level(on ? ojph::OJPH_MSG_ALL_MSG : ojph::OJPH_MSG_NO_MSG); }

} // extern "C"
