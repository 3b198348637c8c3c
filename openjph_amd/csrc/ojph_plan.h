// openjph_amd/csrc/ojph_plan.h -- host-side codestream geometry ("plan") for the GPU hot path.
//
// The reference builds a tree of objects tile -> tile_comp -> resolution -> subband -> codeblock
// (+ precinct) and streams image lines through it.  Here the same geometry is flattened once
// per frame shape into tables (bands, code-blocks, DWT levels, precincts, packet order) that are
// uploaded to HBM and drive a handful of large kernels.  Rules restated from:
//   tile / tile-comp rectangles      src/core/codestream/ojph_codestream_local.cpp:113-220,
//                                    ojph_tile.cpp:253-289
//   resolution / sub-band rectangles ojph_resolution.cpp:302-330
//   code-block grid                  ojph_subband.cpp:133-206
//   precincts + block index map      ojph_resolution.cpp:401-441, ojph_subband.cpp:224-276
//   quantisation (K_max, delta)      ojph_params.cpp:1495-1760, ojph_subband.cpp:153-164
#ifndef OJPH_PLAN_H
#define OJPH_PLAN_H

#include <algorithm>
#include <new>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/ojphgpu.h"

namespace ojphgpu {

struct Rect { uint32_t x0, y0, w, h; };

struct Band {
  uint32_t tile, comp, res, band;
  Rect r;                       // band coordinates
  uint32_t K_max;
  float delta, delta_inv;
  uint32_t xcb, ycb;            // log2 of the code-block size in this band (xcb', ycb')
  uint32_t nbx, nby, first_block;
  uint64_t plane_off;           // elements
  uint32_t pitch;
  bool empty;
};

struct Block {
  uint32_t band;
  uint32_t bx, by;              // position in the band's block grid
  Rect r;                       // relative to the band origin
};

struct Precinct {
  uint32_t tile, comp, res;
  uint32_t img_x, img_y;        // precinct::img_point (ojph_resolution.cpp:428-433)
  Rect cb[4];                   // per band: rectangle in the band's block grid
};

struct Resolution {
  uint32_t tile, comp, res;
  uint32_t kind = 1;            // r > 0: what the level that splits this resolution transforms -- 1 both directions, 2 horizontal
                                // only (bands: HL), 3 vertical only (LH), 0 nothing (no bands): DFS marker segment
  Rect r;
  uint32_t log_ppw, log_pph;
  uint32_t npw, nph;            // precinct grid
  uint32_t first_precinct;
  int band[4];                  // band table indices (-1 = absent)
  uint64_t plane_off;           // raw (un-coded) plane of this resolution, elements
  uint32_t pitch;
};

struct TileComp {
  uint32_t tile, comp;
  Rect r;
  std::vector<uint32_t> res;    // resolution table indices, res[0] = lowest
};

struct Tile {
  uint32_t idx;
  Rect r;
  std::vector<uint32_t> comps;  // tile-comp indices
  std::vector<uint32_t> packets; // precinct indices in progression order
};

struct CodedBlock {             // filled by the codestream parser
  uint64_t offset; uint32_t len1, len2, missing_msbs, num_passes;
};

struct CompGeo {                // one image component on its own (sub-sampled) grid
  uint32_t dx, dy;              // sub-sampling factors (XRsiz, YRsiz)
  uint32_t x0, y0, w, h;        // ceil(image offset / d) and the size up to ceil(image extent / d)
  uint64_t frame_off;           // element offset of the component's plane in a frame (image buffer)
  uint32_t bit_depth; bool is_signed;
};

struct CodStyle {                // the COD, or the COC of one component, resolved (ojph_params_local.h:344-383)
  uint32_t L = 5;               // decompositions
  uint32_t lbw = 6, lbh = 6;    // log2 of the nominal code-block size
  bool rev = false;
  bool causal = false;          // vertically causal code-block style (foreign codestreams; set by the parser)
  uint32_t wavelet = 0;         // >= 2: the ATK marker segment with this index is the wavelet (rev follows it); 0: the Part-1 one `rev` names
  int dfs = -1;                 // >= 0 (COC only): the DFS marker segment with this index defines the decomposition; L is the COD's
  bool has_prec = false;        // Scod / Scoc bit 0: precinct sizes are listed
  uint8_t pexp[36] = { 0 };     // PPx | PPy << 4 per resolution when has_prec
  uint32_t rank = 0;            // COC: creation order (1..); 0 = this is the COD
  uint32_t lpw(uint32_t r) const { return has_prec ? (pexp[r] & 15u) : 15u; }
  uint32_t lph(uint32_t r) const { return has_prec ? (pexp[r] >> 4) : 15u; }
};

struct QuantSet {               // contents of a QCD / QCC marker segment (ojph_params_local.h:690-830)
  uint8_t sqcd = 0; uint32_t guard_bits = 0;
  std::vector<uint8_t> q8;      // reversible: exponent bytes as written
  std::vector<uint16_t> q16;    // irreversible: exponent << 11 | mantissa
  bool present = false;         // QCC: the component has its own marker segment
};

struct AtkDef { uint32_t index; bool rev; uint32_t coeff_type; float K; std::vector<ojphgpu_lift_step> steps; };   // an ATK marker segment
struct DfsDef { uint32_t index; std::vector<uint8_t> types; };                // a DFS marker segment: types[d - 1] of decomposition level d

struct NltSeg { uint16_t comp; uint8_t bd, type; };   // an NLT marker segment as written (Cnlt 65535 = all components)

struct Plan {
  ojphgpu_params p;
  std::vector<NltSeg> nlt;       // the main header's NLT segments, in writing order
  std::vector<uint8_t> nlt3;     // per component: the type 3 non-linearity applies (signed component, type 3 in force)
  bool any_nlt3 = false;
  std::vector<CompGeo> comps;
  uint64_t frame_elems;         // elements of one frame = sum of the component planes
  // reduced-resolution decoding (codestream::restrict_input_resolution): the top skip_read
  // resolutions are not decoded (their blocks count as empty), the top skip_recon are not
  // synthesised; comps / frame_elems then describe the smaller reconstructed frame
  uint32_t skip_read = 0, skip_recon = 0;
  // set by the codestream parser before build_plan: what only the reference's WRITER checks (param_cod::check_validity and
  // friends, called from codestream::write_headers alone, ojph_codestream_local.cpp:571-576) does not stop a codestream from being read
  bool parsed = false;
  bool no_packets = false;      // parser: a progression order byte above 4 -- tile::parse_tile_header reads no packet at all (ojph_tile.cpp:900-901)
  // parser: blocks the reference keeps although their tile-part did not hold all their bytes -- it pads them with zeros
  // (bb_read_chunk, ojph_bitbuffer_read.h:134-150).  Their bytes are not in the codestream: in `coded` they are not coded,
  // here is what the packet header said (block = plan order index, got = bytes the codestream does hold at offset)
  struct PaddedBlock { uint32_t block, got; CodedBlock hdr; };
  std::vector<PaddedBlock> padded;
  // tile-part divisions after the progression order had its say (ojph_codestream_local.cpp:582-620):
  // bit 0 = a tile-part per resolution, bit 1 = per component
  uint32_t tilepart_div = 0, parts_per_tile = 1;
  struct Comment { uint16_t rcom; std::vector<uint8_t> data; };
  std::vector<Comment> comments;   // user COM segments of the main header
  uint32_t ntx, nty;
  CodStyle cod;                  // the main header's COD
  std::vector<CodStyle> coc;     // per component; .rank != 0 = the component has a COC of its own
  const CodStyle& style(uint32_t comp) const { return comp < coc.size() && coc[comp].rank ? coc[comp] : cod; }
  std::vector<AtkDef> atks;      // Part 2: the main header's ATK / DFS marker segments
  std::vector<DfsDef> dfss;
  const AtkDef* atk_of(uint32_t comp) const {
    const uint32_t w = style(comp).wavelet;
    if (w >= 2) for (const AtkDef& a : atks) if (a.index == w) return &a;
    return nullptr;
  }
  // what decomposition level d (1 = the first one applied to the tile-component) of a component transforms
  // (param_dfs::get_dwt_type, ojph_params.cpp:2539-2547): 1 both directions, 2 horizontal, 3 vertical, 0 nothing
  uint32_t level_kind(uint32_t comp, uint32_t d) const {
    const int k = style(comp).dfs;
    if (k < 0) return 1;
    for (const DfsDef& f : dfss) if ((int)f.index == k && !f.types.empty()) return f.types[std::min<size_t>(d, f.types.size()) - 1];
    return 1;
  }
  // the component needs the general lifting kernels (kernels_lift.hip): a Part-2 wavelet or decomposition, or 64-bit samples
  bool general(uint32_t comp) const { return style(comp).wavelet >= 2 || style(comp).dfs >= 0 || (comp < wide.size() && wide[comp]); }
  uint32_t max_decomps = 0;      // over the components
  // decompositions of a component that are synthesised / the top resolution whose blocks are decoded
  uint32_t recon_decomps(uint32_t comp) const { return style(comp).L - skip_recon; }
  // (resolution::skipped_res_for_read counts from the component's own top, ojph_resolution.cpp:254-255)
  uint32_t top_read_res(uint32_t comp) const { return style(comp).L - skip_read; }
  // RC tile-part divisions number the parts c + r * num_comps; a component with fewer decompositions
  // than the largest has no part (c, r > its own), the number stays unused (ojph_tile.cpp:637-652)
  bool part_exists(uint32_t k) const { return tilepart_div != 3 || k / p.num_comps <= style(k % p.num_comps).L; }
  QuantSet qcd;                  // the main header's QCD
  std::vector<QuantSet> qcc;     // per component; .present = the component has a QCC of its own
  std::vector<uint32_t> qcc_order; // components with a QCC in the order the segments are written
  const QuantSet& quant(uint32_t comp) const { return comp < qcc.size() && qcc[comp].present ? qcc[comp] : qcd; }
  std::vector<Tile> tiles;
  std::vector<TileComp> tcomps;
  std::vector<Resolution> ress;
  std::vector<Band> bands;
  std::vector<Block> blocks;
  std::vector<Precinct> precincts;
  std::vector<ojphgpu_level_info> levels;   // DWT levels, highest resolution first per tile-comp
  std::vector<CodedBlock> coded;            // only after parse
  uint64_t arena_elems;
  // components on the reference's 64-bit sample path (more than 32 bits of precision, param_qcd::propose_precision
  // ojph_params.cpp:1684-1706): int64 planes, the 64-bit block coder
  std::vector<uint8_t> wide; bool any_wide = false;
  // the lifting kernel of a component's decomposition level d as the general kernels take it (elem: int32 / int64 / float)
  ojphgpu_lift lift_of(uint32_t comp, uint32_t d) const;
  uint32_t max_block_bytes;
  std::string error;
};

// builds everything from p (p.tile_w/h == 0 -> single tile). Returns 0 or OJPHGPU_E_INVALID.
int build_plan(const ojphgpu_params& p, Plan& plan);
// derives the QCD / QCC contents (ojph_params.cpp:1359-1613); false + plan.error when the
// parameters cannot be quantised (qfactor on an unsupported sampling format)
bool derive_quant(Plan& plan);
// param_nlt::check_validity / get_nonlinear_transform (ojph_params.cpp:2087-2208): the NLT segments to
// write and the components the type 3 non-linearity applies to; parsed = the plan comes from a codestream
bool derive_nlt(Plan& plan, bool parsed);
bool derive_precision(Plan& plan);     // which components take the 64-bit sample path; false + plan.error: cannot be coded here
void assign_planes(Plan& plan);        // arena places of the resolution / band planes, the DWT levels
uint32_t band_Kmax(const Plan& plan, uint32_t comp, uint32_t res, uint32_t band);
float band_delta(const Plan& plan, uint32_t comp, uint32_t res, uint32_t band);   // get_irrev_delta (:1650)
// worst-case coded size of a block of w*h samples with K_max magnitude bits
uint32_t block_scratch_bytes(uint32_t w, uint32_t h, uint32_t K_max);

// Tier-2 writer as a layout (ojph_t2.cpp): `blob` holds every byte of the output that is not a code-block
// byte (markers, packet headers) in codestream order; a job places n bytes at output position dst, taken
// from the blob (blob = 1, src = offset in it) or from the code-block data (blob = 0, src = the block's
// offset in it).  Executed by memcpy on the host (t2_place_host) or by a kernel on the device (the frame
// pipeline), so that the coded bytes need not pass through a host copy.
struct T2Job { uint64_t dst, src; uint32_t n, blob; };
struct T2Layout { std::vector<uint8_t> blob; std::vector<T2Job> jobs; uint64_t total = 0; };
int t2_layout_tiles(const Plan& P, const ojphgpu_coded_block* cb, size_t t0, size_t t1, T2Layout& L, uint32_t* len_out);
int t2_layout_codestream(const Plan& P, const ojphgpu_coded_block* cb, T2Layout& L);
void t2_place_host(const T2Layout& L, const uint8_t* data, uint8_t* out);

// Nothing may leave the C ABI as a C++ exception (a codestream from anywhere can ask for tables the
// host cannot hold): entry points that build containers run their bodies through this.
template <typename F>
int no_throw(F f)
{
  try { return f(); }
  catch (const std::bad_alloc&) { return OJPHGPU_E_NOMEM; }
  catch (...) { return OJPHGPU_E_INVALID; }
}

}  // namespace ojphgpu

struct ojphgpu_plan { ojphgpu::Plan plan; };

#endif
