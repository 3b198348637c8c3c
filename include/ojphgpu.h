/* include/ojphgpu.h -- C ABI of the MI355X-native HTJ2K hot path (libojphgpu.so).
 *
 * This is the drop-in boundary for the hot path of aous72/OpenJPH 0.31.0: 5/3 + 9/7 DWT,
 * quantise transfer and the HT block coder.  In the reference that path sits behind three
 * per-line / per-block function-pointer tables
 *     struct codeblock_fun            src/core/codestream/ojph_codeblock_fun.h:93-119
 *     rev_/irv_ vert_step, horz_ana.. src/core/transform/ojph_transform.h:61-96
 *     colour / convert pointers       src/core/transform/ojph_colour.h:53-103
 * which are called once per image line or per code-block from resolution::push_line/pull_line
 * (ojph_resolution.cpp:547,713), codeblock::push/encode/decode/pull_line (ojph_codeblock.cpp:
 * 115-266) and tile::push/pull (ojph_tile.cpp:332-518).  A device round trip per line is not
 * workable, so the same data contracts are exposed here *batched*: one call transforms every
 * plane / codes every code-block of a frame, driven by descriptor tables that the host side
 * (the plan) derives with the reference's geometry rules.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * OJPHGPU_E_* code and never throws.  Pointers named d_* are device (HBM) pointers, h_* are
 * host pointers.  `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 * device entry points are asynchronous on `stream`.
 */
#ifndef OJPHGPU_H
#define OJPHGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OJPHGPU_OK              0
#define OJPHGPU_E_INVALID      -1   /* bad argument / unsupported parameter combination      */
#define OJPHGPU_E_NOMEM        -2
#define OJPHGPU_E_HIP          -3   /* a HIP runtime call failed (no device, launch failure)  */
#define OJPHGPU_E_CODESTREAM   -4   /* malformed codestream (== the reference's OJPH_ERROR)   */
#define OJPHGPU_E_OVERFLOW     -5   /* output buffer too small; *out_len holds the need       */
#define OJPHGPU_E_BLOCK        -6   /* a code-block failed to decode (non-resilient mode)     */
#define OJPHGPU_E_AGAIN        -7   /* a frame pipeline has no free slot: collect a result first */
#define OJPHGPU_E_UNCOLLECTED  -8   /* ojphgpu_decoder_run_device*: the run BEFORE this one asked to be decoded again (see
                                     * ojphgpu_decoder_failed_blocks) and was never collected -- its frame was incomplete */

/* ------------------------------------------------------------------------------------------ *
 * 1. Codestream parameters: what ojph::param_siz / param_cod / param_qcd setters carry
 *    (src/core/openjph/ojph_params.h:68-240).
 * ------------------------------------------------------------------------------------------ */
#define OJPHGPU_MAX_SUBSAMPLED_COMPS 16
#define OJPHGPU_MAX_COC_COMPS 16
/* Part 2 (ITU-T T.801) wavelets, as far as the reference reads them (it never writes them): an ATK marker segment = a
 * lifting kernel (param_atk, ojph_params_local.h:1105-1230, read :2770-2866: whole-sample symmetric kernels with one
 * coefficient per step, even-indexed first), a DFS marker segment = which directions every decomposition level
 * transforms (param_dfs, :1030-1100, read ojph_params.cpp:2596-2644).  A lifting step, in synthesis order:
 * reversible x -+= (b + a (l + r)) >> e, irreversible x -+= A (l + r). */
#define OJPHGPU_MAX_LIFT_STEPS 16
#define OJPHGPU_MAX_ATK 4
#define OJPHGPU_MAX_DFS 4
typedef struct ojphgpu_lift_step { int32_t a, b, e; float A; } ojphgpu_lift_step;
typedef struct ojphgpu_atk {
  uint8_t  index;                  /* Satk & 0xFF: 2..255 (0 and 1 name the Part-1 wavelets); 0 = entry unused       */
  uint8_t  reversible, num_steps;
  uint8_t  coeff_type;             /* how the coefficients are written: 0 8-bit, 1 16-bit integers, 2 float, 3 double */
  float    K;                      /* irreversible: the scaling factor                                                */
  ojphgpu_lift_step steps[OJPHGPU_MAX_LIFT_STEPS];
} ojphgpu_atk;
typedef struct ojphgpu_dfs {
  uint8_t  used, index;            /* Sdfs: 0..15                                                                     */
  uint8_t  num_levels, reserved;   /* Ids; levels beyond it repeat the last one (param_dfs::get_dwt_type :2539-2547)  */
  uint8_t  types[32];              /* per decomposition level, 1 = the first one applied to the image:
                                      0 none, 1 both directions, 2 horizontal only, 3 vertical only                   */
} ojphgpu_dfs;
/* A COC marker segment: what param_cod's comp_idx setters build (ojph_params.h:146-151,
 * ojph_params.cpp:255-282).  The reference starts a new COC from the SPcod defaults -- 5
 * decompositions, 64x64 blocks, wavelet_trans 0 (the 9/7), no precincts (ojph_params_local.h:344-353)
 * -- not from the COD; the facade does the same. */
typedef struct ojphgpu_coc {
  uint8_t  rank;                   /* 0: the component follows the COD.  k >= 1: it has a COC, the k-th one
                                      created (COCs are written in creation order, ojph_params.cpp:1081-1094) */
  uint8_t  reversible;             /* param_cod::set_reversible(comp_idx, ..)                       */
  uint8_t  num_decomps;            /* param_cod::set_num_decomposition(comp_idx, ..)                */
  uint8_t  log_block_w, log_block_h; /* param_cod::set_block_dims(comp_idx, ..), log2 (2..10)        */
  uint8_t  has_precincts;          /* param_cod::set_precinct_size(comp_idx, ..): precinct_exps[0..num_decomps] */
  uint8_t  reserved[2];            /* are PPx | PPy << 4 per resolution; 0 = 32768 x 32768 everywhere.
                                      reserved[0]: bit 0 vertically causal (parser); bit 7: the decomposition is defined
                                      by the DFS marker segment with the index in bits 1..4 -- the number of
                                      decompositions is then the COD's (ojph_params_local.h:503-516, :613-618);
                                      reserved[1]: the wavelet is the ATK marker segment with this index (2..255), 0 =
                                      `reversible` names the Part-1 wavelet */
  uint8_t  precinct_exps[36];
} ojphgpu_coc;
typedef struct ojphgpu_params {
  uint32_t width, height;        /* param_siz::set_image_extent                               */
  uint32_t num_comps;            /* param_siz::set_num_components  (all comps share the below) */
  uint32_t bit_depth;            /* param_siz::set_component(.., bit_depth, ..)               */
  uint32_t is_signed;
  uint32_t reversible;           /* param_cod::set_reversible                                 */
  uint32_t num_decomps;          /* param_cod::set_num_decomposition                          */
  uint32_t block_w, block_h;     /* param_cod::set_block_dims                                 */
  uint32_t color_transform;      /* param_cod::set_color_transform                            */
  uint32_t tile_w, tile_h;       /* param_siz::set_tile_size; 0 = one tile                    */
  uint32_t prog_order;           /* 0 LRCP 1 RLCP 2 RPCL 3 PCRL 4 CPRL (set_progression_order) */
  float    qstep;                /* param_qcd::set_irrev_quant; <= 0 selects 2^-min(16,B)     */
  uint32_t precinct_w, precinct_h; /* 0 = 32768 (no explicit precincts)                        */
  uint32_t tlm;                  /* codestream::request_tlm_marker                            */
  uint32_t reserved[4];          /* [0] bit 0: vertically causal code-block style (set by the parser)
                                    [1] codestream::set_tilepart_divisions: bit 0 = a tile-part per
                                        resolution, bit 1 = per component (what the progression
                                        order cannot honour is dropped as write_headers drops it)
                                    [2] param_qcd::set_qfactor: 1..100, 0 = not set            */
  uint8_t  precinct_exps[36];    /* param_cod::set_precinct_size with a list: per resolution (0 =
                                    lowest) PPx | PPy << 4; all zero = precinct_w/h everywhere  */
  /* reference grid (ojph_params.h:68-112).  width/height stay the image SIZE: the extent the
     reference's set_image_extent takes is image_x0 + width, image_y0 + height.                 */
  uint32_t image_x0, image_y0;   /* param_siz::set_image_offset                                 */
  uint32_t tile_x0, tile_y0;     /* param_siz::set_tile_offset (<= image offset)                */
  uint8_t  comp_dx[OJPHGPU_MAX_SUBSAMPLED_COMPS];   /* param_siz::set_component downsampling of   */
  uint8_t  comp_dy[OJPHGPU_MAX_SUBSAMPLED_COMPS];   /* component c < 16; 0 = 1; later ones are 1  */
  uint8_t  comp_depth[OJPHGPU_MAX_SUBSAMPLED_COMPS]; /* set_component bit depth of component c < 16 */
                                                     /* when it differs from bit_depth; 0 = same   */
  uint8_t  comp_sign[OJPHGPU_MAX_SUBSAMPLED_COMPS];  /* 0 = is_signed, 1 = unsigned, 2 = signed    */
  ojphgpu_coc coc[OJPHGPU_MAX_COC_COMPS];            /* per-component coding style of component c < 16 */
  /* NLT marker segments: param_nlt::set_nonlinear_transform (ojph_params.h:299-345).  Values: 0 = not
     set, 1 = type 0 (no non-linearity) set, 4 = type 3 (binary complement <-> sign magnitude, the only
     other type the reference supports) set.  nlt_default = the ALL_COMPS entry; nlt_comp[c] / nlt_rank[c]
     = component c < 16 and the order its entry was created in (1..); nlt_bd_* = BDnlt as read by the
     parser (0 when the plan was not parsed). */
  uint8_t  nlt_default, nlt_bd_default;
  uint8_t  nlt_comp[OJPHGPU_MAX_COC_COMPS], nlt_rank[OJPHGPU_MAX_COC_COMPS], nlt_bd[OJPHGPU_MAX_COC_COMPS];
  uint8_t  nlt_reserved[2];
  /* param_qcd::set_qfactor(comp_idx, ctype, qfactor) (ojph_params.h:251-254): a QCC for component
     c < 16 with its own quality factor (1..100; 0 = not set), the visual weights of ctype (0 Y, 1 Cb,
     2 Cr) and its place in the creation order (QCCs are written in that order, then the ones the
     library adds, ojph_params.cpp:1822-1834) */
  uint8_t  qcc_qfactor[OJPHGPU_MAX_COC_COMPS], qcc_ctype[OJPHGPU_MAX_COC_COMPS], qcc_rank[OJPHGPU_MAX_COC_COMPS];
  /* Part 2: the COD's wavelet when it is an ATK marker segment (its index, 2..255; 0 = `reversible` decides), and the
     ATK / DFS marker segments of the main header (see ojphgpu_atk / ojphgpu_dfs) */
  uint8_t  wavelet, part2_reserved[3];
  ojphgpu_atk atk[OJPHGPU_MAX_ATK];
  ojphgpu_dfs dfs[OJPHGPU_MAX_DFS];
} ojphgpu_params;

/* ------------------------------------------------------------------------------------------ *
 * 2. The plan: host-side geometry (tile -> tile-comp -> resolution -> subband -> code-block,
 *    precincts, K_max / delta per sub-band) derived with the reference's rules
 *    (ojph_tile.cpp:191-329, ojph_resolution.cpp:240-470, ojph_subband.cpp:117-276,
 *    ojph_params.cpp:1495-1760).  Host only; needs no GPU.
 * ------------------------------------------------------------------------------------------ */
typedef struct ojphgpu_plan ojphgpu_plan;

typedef struct ojphgpu_band_info {   /* one sub-band of one tile-component resolution */
  uint32_t tile, comp, res, band;    /* band: 0 LL 1 HL 2 LH 3 HH */
  uint32_t x0, y0, w, h;             /* band rectangle in band coordinates */
  uint32_t K_max;
  float    delta, delta_inv;         /* irreversible: step / 2^(31-K_max) and its inverse */
  uint32_t nbx, nby, first_block;    /* code-block grid and index of its first block */
  uint64_t plane_off;                /* element (4 B) offset of the band plane in the coefficient arena */
  uint32_t pitch;                    /* elements */
  uint32_t reserved;
} ojphgpu_band_info;

typedef struct ojphgpu_block_info {  /* one code-block */
  uint32_t band;                     /* index into the band table */
  uint32_t x0, y0, w, h;             /* rectangle relative to the band origin */
  uint32_t K_max;
} ojphgpu_block_info;

typedef struct ojphgpu_level_info {  /* one DWT level of one tile-component: res -> res-1 + 3 bands */
  uint32_t tile, comp, res;          /* res = resolution being analysed (>= 1) */
  uint32_t w, h, x_even, y_even;     /* input plane geometry */
  uint64_t src_off; uint32_t src_pitch;
  uint64_t ll_off;  uint32_t ll_pitch;
  uint64_t hl_off;  uint32_t hl_pitch;
  uint64_t lh_off;  uint32_t lh_pitch;
  uint64_t hh_off;  uint32_t hh_pitch;
  uint32_t kind;                     /* 1 both directions (always, without a DFS marker segment), 2 horizontal only (ll +
                                        hl), 3 vertical only (ll + lh), 0 no transform (ll = the plane) */
} ojphgpu_level_info;

typedef struct ojphgpu_coded_block { /* what the packet headers say about one code-block */
  uint64_t offset;                   /* byte offset of the block's data (encode: in the block-data
                                        buffer; decode: in the codestream) */
  uint32_t len1, len2;               /* pass_length[0], pass_length[1] (0,0 = not included) */
  uint32_t missing_msbs, num_passes;
} ojphgpu_coded_block;

int  ojphgpu_plan_create(const ojphgpu_params* params, ojphgpu_plan** out);
void ojphgpu_plan_destroy(ojphgpu_plan* plan);
int  ojphgpu_plan_params(const ojphgpu_plan* plan, ojphgpu_params* out);
/* counts: out[0]=tiles [1]=bands [2]=blocks [3]=dwt levels [4]=coefficient arena elements
 *         [5]=max coded bytes of one block [6]=precincts [7]=tile-comps */
int  ojphgpu_plan_counts(const ojphgpu_plan* plan, uint64_t out[8]);
/* user COM marker segments written after the library's own one (the `comments` argument of
 * codestream::write_headers, ojph_codestream_local.cpp:686-703): n segments, segment i = len[i] bytes
 * at data[i] with registration value rcom[i] (1 = Latin text, 0 = binary).  Replaces earlier ones. */
int  ojphgpu_plan_set_comments(ojphgpu_plan* plan, const uint8_t* const* data, const uint16_t* len,
                               const uint16_t* rcom, uint32_t n);
/* tile-parts every tile is written in (1 without tile-part divisions) */
int  ojphgpu_plan_tile_parts(const ojphgpu_plan* plan, uint32_t* parts_per_tile);
int  ojphgpu_plan_bands(const ojphgpu_plan* plan, ojphgpu_band_info* out, size_t n);
int  ojphgpu_plan_blocks(const ojphgpu_plan* plan, ojphgpu_block_info* out, size_t n);
int  ojphgpu_plan_levels(const ojphgpu_plan* plan, ojphgpu_level_info* out, size_t n);
/* element offset + pitch of tile-component (tile, comp)'s full-resolution plane in the arena;
 * rect = x0, y0, w, h of the tile-component on the component's own (sub-sampled) grid */
int  ojphgpu_plan_comp_plane(const ojphgpu_plan* plan, uint32_t tile, uint32_t comp,
                             uint64_t* off, uint32_t* pitch, uint32_t rect[4]);
/* The image buffer every codec call takes ("frame"): the int32 planes of the components one after
 * the other, component c being out[2] x out[3] samples (param_siz::get_recon_width / _height),
 * tightly packed, starting out[4] | out[5] << 32 elements into the frame.  out[0], out[1] = the
 * component's origin on its own grid (ceil(image offset / sub-sampling)); out[6], out[7] = its
 * sub-sampling factors.  comp == num_comps: out[4] | out[5] << 32 = elements of one whole frame. */
int  ojphgpu_plan_comp_info(const ojphgpu_plan* plan, uint32_t comp, uint32_t out[8]);
/* bit depth and signedness of a component (param_siz::get_bit_depth / is_signed) */
int  ojphgpu_plan_comp_format(const ojphgpu_plan* plan, uint32_t comp, uint32_t* bit_depth, uint32_t* is_signed);
/* coding style of a component, from its COC or the COD (param_cod's comp_idx getters,
 * ojph_params.cpp:374-399): out[0] decompositions, [1] reversible, [2] [3] log2 code-block width /
 * height, [4] 1 = the component has a COC, [5] decompositions left after restrict_resolution,
 * [6] 1 = the type 3 non-linearity applies to the component (NLT marker segment, signed component) */
int  ojphgpu_plan_comp_style(const ojphgpu_plan* plan, uint32_t comp, uint32_t out[8]);
/* the wavelet of decomposition level `level` (1 = the first one applied to the tile-component) of a component, as the
 * general lifting kernels take it: the steps of its ATK marker segment (or of the Part-1 wavelet its COD / COC names),
 * the directions its DFS marker segment has the level transform, elem by the component's sample path
 * (cod.access_atk() + param_dfs::get_dwt_type as resolution::finalize_alloc combines them, ojph_resolution.cpp:264-300) */
struct ojphgpu_lift;
int  ojphgpu_plan_comp_lift(const ojphgpu_plan* plan, uint32_t comp, uint32_t level, struct ojphgpu_lift* out);

/* ------------------------------------------------------------------------------------------ *
 * 3. Tier-2 on the host: marker segments + packet headers (tag trees, pass lengths) around the
 *    code-block bytes.  Replaces local::codestream::write_headers/flush
 *    (ojph_codestream_local.cpp:556-712,1148), tile::flush (ojph_tile.cpp:584-774),
 *    precinct::prepare_precinct/write/parse (ojph_precinct.cpp:94-573) and
 *    read_headers/read (ojph_codestream_local.cpp:769-1146).
 * ------------------------------------------------------------------------------------------ */
/* blocks[i] describes code-block i of the plan (plan order); len1 == 0 -> block not coded.
 * Writes a complete codestream (SOC .. EOC). */
int  ojphgpu_t2_write(const ojphgpu_plan* plan, const uint8_t* h_block_data,
                      const ojphgpu_coded_block* blocks, uint8_t* h_out, size_t cap,
                      size_t* out_len);
/* The same in pieces, for tile-sharded encoding (tiles are independent: own DWT, blocks, packets
 * and tile-parts, ojph_tile.cpp:584-774): the tile-parts (SOT .. last packet) of tiles
 * [tile_first, tile_first + tile_count) -- tile_part_len[i * parts_per_tile + k] receives Psot of
 * tile-part k of tile tile_first + i --
 * and the main header (SOC .. last main-header marker; needs every tile's Psot only when a TLM
 * marker was requested).  codestream = main header | tile-parts in tile order | EOC (0xFFD9). */
int  ojphgpu_t2_write_tiles(const ojphgpu_plan* plan, const uint8_t* h_block_data,
                            const ojphgpu_coded_block* blocks, uint32_t tile_first,
                            uint32_t tile_count, uint8_t* h_out, size_t cap, size_t* out_len,
                            uint32_t* tile_part_len);
int  ojphgpu_t2_write_main_header(const ojphgpu_plan* plan, const uint32_t* tile_part_len,
                                  uint8_t* h_out, size_t cap, size_t* out_len);
/* Parses main header + all tile-parts; creates the plan the codestream implies. */
int  ojphgpu_t2_parse(const uint8_t* h_codestream, size_t len, int resilient, ojphgpu_plan** out);
/* read_headers + restrict_input_resolution + read in the reference's order: the same as ojphgpu_t2_parse followed by
 * ojphgpu_plan_restrict_resolution on an undamaged codestream; on a damaged or truncated one the tile-parts are read the way
 * the reference reads them under the restriction (resolution-major progressions stop at the highest resolution wanted, packets
 * of unwanted resolutions are stepped over: ojph_tile.cpp:806-848, ojph_precinct.cpp:531-541). */
int  ojphgpu_t2_parse_restricted(const uint8_t* h_codestream, size_t len, int resilient, uint32_t skipped_res_for_data,
                                 uint32_t skipped_res_for_recon, ojphgpu_plan** out);
/* after ojphgpu_t2_parse: per-block coded info (offsets are into the parsed codestream) */
int  ojphgpu_plan_coded_blocks(const ojphgpu_plan* plan, ojphgpu_coded_block* out, size_t n);
/* after ojphgpu_t2_parse of a DAMAGED codestream: the blocks the reference would decode from bytes the codestream does not
 * hold -- a packet header promising more bytes than its tile-part has left makes bb_read_chunk (ojph_bitbuffer_read.h:134-150)
 * pad the block with zeros.  hdr = what the packet header said, got = how many of hdr.len1 + hdr.len2 bytes exist at
 * hdr.offset.  ojphgpu_plan_coded_blocks reports these blocks as not coded and the device decoder leaves them zero (what the
 * reference does with them when its block decoder refuses the padded bytes; the cases it does not are listed here).
 * *count receives how many there are (out == NULL: only that); OJPHGPU_E_OVERFLOW when cap is too small. */
typedef struct ojphgpu_padded_block { uint32_t block, got; ojphgpu_coded_block hdr; } ojphgpu_padded_block;
int  ojphgpu_plan_padded_blocks(const ojphgpu_plan* plan, ojphgpu_padded_block* out, size_t cap, size_t* count);
/* codestream::restrict_input_resolution (ojph_codestream_local.cpp:883-900) on a parsed plan, before
 * a decoder is created from it: the top `skipped_res_for_data` resolutions are not decoded and the
 * top `skipped_res_for_recon` (<= the former) are not synthesised.  The frame of the decoder -- and
 * what ojphgpu_plan_comp_info reports -- shrinks to ceil(size / 2^skipped_res_for_recon). */
int  ojphgpu_plan_restrict_resolution(ojphgpu_plan* plan, uint32_t skipped_res_for_data,
                                      uint32_t skipped_res_for_recon);

/* ------------------------------------------------------------------------------------------ *
 * 4. Batched device stages.  Descriptor arrays live in device memory.
 * ------------------------------------------------------------------------------------------ */
typedef struct ojphgpu_dwt_desc {    /* one plane, one DWT level (32-bit elements everywhere) */
  uint64_t src_off, ll_off, hl_off, lh_off, hh_off;   /* element offsets from `base` */
  uint32_t src_pitch, ll_pitch, hl_pitch, lh_pitch, hh_pitch;
  uint32_t w, h;                                       /* size of the un-decomposed plane */
  uint32_t x_even, y_even;                             /* origin parity: 1 = even coordinate */
  uint32_t reserved;
} ojphgpu_dwt_desc;

/* K1-K4 (ojph_transform.cpp:209-852 driven by ojph_resolution.cpp:547-949).  `reversible`
 * selects the 5/3 integer or the 9/7 float lifting; `base` is int32 or float accordingly.
 * max_w / max_h = largest plane in the batch (sizes the launch grid). */
int ojphgpu_dwt_forward(void* stream, int reversible, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                        uint32_t max_w, uint32_t max_h, void* d_base);
int ojphgpu_dwt_inverse(void* stream, int reversible, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                        uint32_t max_w, uint32_t max_h, void* d_base);

/* The transform in its GENERAL form (kernels_lift.hip): any lifting kernel an ATK marker segment can describe
 * (param_atk, ojph_params.cpp:2654-2896), levels that transform one direction only (DFS marker segment:
 * resolution::pull_line's HORZ_TRX / VERT_TRX, ojph_resolution.cpp:290-300, :725-949), and 64-bit integer samples
 * (gen_rev_vert_step64 / gen_rev_horz_ana64 / _syn64, ojph_transform.cpp:261,415,593 -- the reference's sample path
 * for more than 32 bits of precision).  Steps are listed in synthesis order; a reversible step is
 * x -+= (b + a (l + r)) >> e, an irreversible one x -+= A (l + r) with K scaling around the steps.
 * elem: 0 = int32, 1 = int64 planes (an element offset of a descriptor still counts 32-bit elements, a pitch counts
 * samples), 2 = float.  horz / vert = 0: the level leaves that direction alone -- all its samples are "low" there, only
 * ll and hl (vert = 0) or ll and lh (horz = 0) are read / written.  The planes the descriptors name as src are
 * transformed IN PLACE before they are split into the bands (forward) / after they are joined (inverse). */
typedef struct ojphgpu_lift {
  uint32_t num_steps, elem, horz, vert;
  float    K;
  ojphgpu_lift_step steps[OJPHGPU_MAX_LIFT_STEPS];
} ojphgpu_lift;
int ojphgpu_dwt_forward_general(void* stream, const ojphgpu_lift* kernel, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                                uint32_t max_w, uint32_t max_h, void* d_base);
int ojphgpu_dwt_inverse_general(void* stream, const ojphgpu_lift* kernel, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                                uint32_t max_w, uint32_t max_h, void* d_base);
/* the top level of general-lifting components with the conversion from / to the image samples fused in, like
 * ojphgpu_dwt_forward_image_ex / _inverse_image_ex do it for the 5/3 and the 9/7 (rev_convert / irv_convert_to_float / ..._to_integer,
 * ojph_colour.cpp:238-436, applied in the level's loads / stores): int32 or float working samples (kernel->elem 0 / 2), one to
 * four lifting steps, both directions; d_descs address the image planes as for the _image_ex calls (reserved = the plane's own
 * bit depth | signed << 8, or 0 for params'); container_bits 8 / 16 / 32.  No colour transform. */
int ojphgpu_dwt_forward_general_image(void* stream, const ojphgpu_lift* kernel, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                      uint32_t n, uint32_t max_w, uint32_t max_h, const void* d_image, void* d_base, int container_bits);
int ojphgpu_dwt_inverse_general_image(void* stream, const ojphgpu_lift* kernel, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs,
                                      uint32_t n, uint32_t max_w, uint32_t max_h, void* d_image, void* d_base, int container_bits);

/* The same transforms with the sample conversion of the adjacent stage fused in (no colour
 * transform): the first analysis level reads the int32 image planes directly -- level shift
 * (gen_rev_convert, ojph_colour.cpp:238) or int -> float (gen_irv_convert_to_float, :388) applied
 * in the load -- and the last synthesis level writes them (gen_irv_convert_to_integer, :316).
 * For these two calls desc.src_off / src_pitch address the plane inside d_image (elements),
 * everything else inside d_base.  Saves one read + one write of the whole frame per direction. */
int ojphgpu_dwt_forward_image(void* stream, const ojphgpu_params* params,
                              const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w,
                              uint32_t max_h, const int32_t* d_image, void* d_base);
/* the general form of the four calls around it: container_bits = 32 | 16 | 8 (two's complement for signed
 * components, else the full unsigned range of the container; bit depths up to the container's width, 31 for
 * 32).  colour != 0: the descriptors come in triples -- the planes of the three colour components of a tile,
 * which share their geometry -- and the component transform (RCT for the 5/3, ICT for the 9/7:
 * gen_rct_forward / _backward, gen_ict_forward / _backward, ojph_colour.cpp:443-571) is applied in the same
 * loads / stores, so that a colour-transformed frame needs no conversion pass over HBM either. */
int ojphgpu_dwt_forward_image_ex(void* stream, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                                 uint32_t max_w, uint32_t max_h, const void* d_image, void* d_base, int container_bits, int colour);
int ojphgpu_dwt_inverse_image_ex(void* stream, const ojphgpu_params* params, const ojphgpu_dwt_desc* d_descs, uint32_t n,
                                 uint32_t max_w, uint32_t max_h, void* d_image, void* d_base, int container_bits, int colour);
/* the same with the image samples in 16-bit containers: int16 (two's complement) for signed
 * components, uint16 otherwise; bit depths up to 16.  Halves the HBM traffic of the image side of
 * the top level (and the PCIe traffic of whoever fills / drains the image buffer). */
int ojphgpu_dwt_forward_image16(void* stream, const ojphgpu_params* params,
                                const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
                                const uint16_t* d_image, void* d_base);
int ojphgpu_dwt_inverse_image16(void* stream, const ojphgpu_params* params,
                                const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w, uint32_t max_h,
                                uint16_t* d_image, void* d_base);
int ojphgpu_dwt_inverse_image(void* stream, const ojphgpu_params* params,
                              const ojphgpu_dwt_desc* d_descs, uint32_t n, uint32_t max_w,
                              uint32_t max_h, int32_t* d_image, void* d_base);

typedef struct ojphgpu_cb_desc {     /* one code-block */
  uint64_t coef_off;                 /* element offset of the block's first sample in `d_coef` */
  uint32_t pitch;                    /* elements */
  uint16_t w, h;
  uint8_t  K_max, reversible, missing_msbs, num_passes;   /* last two: decode only; reversible bit 1
                                        (decode): vertically causal code-block style; bit 2: the block is on the
                                        64-bit sample path (int64 samples at coef_off, the 64-bit block coder) */
  float    delta;                    /* irreversible: encode uses 1/delta, decode uses delta */
  uint32_t len1, len2;               /* decode: pass lengths */
  uint64_t data_off;                 /* decode: byte offset of the coded bytes in `d_data`;
                                        encode: byte offset of this block's scratch slot */
  uint32_t scratch_cap;              /* encode: bytes available at data_off; decode: offset of the
                                        block's first per-quad record in d_quad_scratch (elements),
                                        see ojphgpu_ht_decode_layout */
  uint32_t reserved;                 /* decode: offset of the block's area in d_aux (elements) */
} ojphgpu_cb_desc;

typedef struct ojphgpu_cb_result {   /* encode result per block */
  uint32_t offset;                   /* byte offset of the block in the compacted output */
  uint32_t length;                   /* pass_length[0]; 0 = block has no significant sample */
} ojphgpu_cb_result;

/* K5/K6 + K8: quantise transfer (ojph_codestream_gen.cpp:59-121) fused into the HT cleanup
 * encoder (ojph_block_encoder.cpp:542-1017).  One wavefront per code-block.  d_cursor is a
 * device uint32 (zeroed by the caller) that ends up holding the number of bytes written to
 * d_out; d_status (device uint32, zeroed) receives a non-zero value on scratch/out overflow. */
int ojphgpu_ht_encode(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                      const void* d_coef, uint8_t* d_scratch, uint8_t* d_out, uint32_t out_cap,
                      ojphgpu_cb_result* d_results, uint32_t* d_cursor, uint32_t* d_status);

/* K9 + K7: HT decoder (ojph_block_decoder32.cpp:742-1613) with the dequantise transfer
 * (ojph_codestream_gen.cpp:124-168) fused.  Three launches: prep (one wavefront per block:
 * un-stuffs the VLC and MEL segments into flat bit strings in d_aux), step 1 (the serial MEL /
 * VLC / U-VLC chains, one lane per block -> one 32-bit record per quad in d_quad_scratch), step 2
 * (MagSgn -> de-quantised samples, one wavefront per block, one lane per sample column).
 * The scratch of the two intermediate products is laid out by ojphgpu_ht_decode_layout, which fills
 * blocks[i].scratch_cap and blocks[i].reserved of the HOST copy of the descriptors before they are
 * uploaded:
 *  - per-quad records: step 1 advances 64 blocks per wavefront (descriptors 64g .. 64g+63 of a
 *    launch), one quad PAIR per lane and iteration, so the records of such a group are interleaved
 *    pair-major -- pair p of the group's lane l sits at group base + 2 * (64 p + l) -- and every store
 *    of the wavefront is one contiguous 512-byte segment.  blocks[i].scratch_cap = offset (uint32
 *    elements) of the block's pair 0; pair p = quad row * ceil(QW / 2) + quad pair follows 128 p
 *    elements further.  A launch over a sub-range must start at a multiple of 64 descriptors of the
 *    array that was laid out.
 *  - blocks[i].reserved = offset (uint32 elements) of the block's flat VLC / MEL strings inside
 *    d_aux, ojphgpu_ht_decode_aux_words(len1) elements long.
 * d_block_status[i] = 0 ok / non-zero failed (block zeroed), mirroring the bool of decode_cb32. */
uint32_t ojphgpu_ht_decode_aux_words(uint32_t len1);
int ojphgpu_ht_decode_layout(ojphgpu_cb_desc* h_blocks, uint32_t n, uint64_t* quad_elems, uint64_t* aux_elems);
int ojphgpu_ht_decode(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                      const uint8_t* d_data, void* d_coef, uint32_t* d_quad_scratch,
                      uint32_t* d_aux, uint8_t* d_block_status);
/* the three launches one by one (ojphgpu_ht_decode == prep, step1, step2 in this order) */
int ojphgpu_ht_decode_prep(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                           const uint8_t* d_data, uint32_t* d_aux);
int ojphgpu_ht_decode_step1(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                            const uint8_t* d_data, const uint32_t* d_aux, uint32_t* d_quad_scratch,
                            uint8_t* d_block_status);
int ojphgpu_ht_decode_step2(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                            const uint8_t* d_data, const uint32_t* d_quad_scratch, void* d_coef,
                            uint8_t* d_block_status);
/* fourth launch, only needed when some block has num_passes > 1 (foreign codestreams): SigProp +
 * MagRef passes (ojph_block_decoder32.cpp:1318-1609) over the blocks that carry them; step 2 left
 * those blocks as sign-magnitude words, this launch refines and de-quantises them.  Bit 1 of
 * blocks[i].reversible selects the vertically (stripe) causal mode of the code-block style.
 * d_quad_scratch: the records ojphgpu_ht_decode_step1 wrote -- the significance the passes start from
 * is the quads' rho bits, as in the reference (:1321-1351), not "the decoded sample is not zero". */
int ojphgpu_ht_decode_refine(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n,
                             const uint8_t* d_data, const uint32_t* d_quad_scratch, void* d_coef,
                             const uint8_t* d_block_status);

typedef struct ojphgpu_convert_desc { /* one tile-component */
  uint64_t plane_off;                /* element offset in the arena */
  uint32_t pitch, w, h;
  uint32_t src_x0, src_y0;           /* position inside the component's image plane */
  uint32_t img_pitch;                /* width of that plane */
  uint64_t img_off;                  /* element offset of that plane in the image buffer (a frame
                                        batch adds the frame's offset) */
  uint32_t fmt;                      /* bit depth | signed << 8 of the component; 0 = from params;
                                        | 0x200 when bit 10 says which conversion the component takes
                                        (1 = reversible level shift, 0 = to float): components with a COC;
                                        | 0x800: NLT type 3 (negative v <-> -v - 2^(B-1) - 1, ojph_colour.cpp:273-311);
                                        | 0x1000: the component is on the 64-bit sample path, its plane holds int64
                                        samples (gen_rev_convert 32 <-> 64 bits, gen_rct_* with 64-bit Y Cb Cr,
                                        ojph_colour.cpp:250-268, :467-489, :517-541) */
  uint32_t reserved;
} ojphgpu_convert_desc;

/* K10/K11: level shift / int<->float conversion and RCT / ICT (ojph_colour.cpp:238-571 as
 * driven by tile::push/pull, ojph_tile.cpp:332-518).  Image samples are int32 planes
 * (component-major, see ojphgpu_plan_comp_info), like the i32 line_bufs the reference exchanges.
 * With the colour transform the first three components share their geometry. */
int ojphgpu_convert_forward(void* stream, const ojphgpu_params* params,
                            const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                            uint32_t max_w, uint32_t max_h, const int32_t* d_image, void* d_arena);
int ojphgpu_convert_inverse(void* stream, const ojphgpu_params* params,
                            const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                            uint32_t max_w, uint32_t max_h, int32_t* d_image, const void* d_arena);
/* container_bits = 32 | 16 | 8 */
int ojphgpu_convert_forward_ex(void* stream, const ojphgpu_params* params, const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                               uint32_t max_w, uint32_t max_h, const void* d_image, void* d_arena, int container_bits);
int ojphgpu_convert_inverse_ex(void* stream, const ojphgpu_params* params, const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                               uint32_t max_w, uint32_t max_h, void* d_image, const void* d_arena, int container_bits);
/* 16-bit sample containers (see ojphgpu_dwt_forward_image16) */
int ojphgpu_convert_forward16(void* stream, const ojphgpu_params* params,
                              const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                              uint32_t max_w, uint32_t max_h, const uint16_t* d_image, void* d_arena);
int ojphgpu_convert_inverse16(void* stream, const ojphgpu_params* params,
                              const ojphgpu_convert_desc* d_descs, uint32_t n_tiles,
                              uint32_t max_w, uint32_t max_h, uint16_t* d_image, const void* d_arena);

/* ------------------------------------------------------------------------------------------ *
 * 5. Whole-frame codec objects: what an ojph::codestream-compatible facade calls from
 *    flush() (encode) and create()/pull() (decode).
 * ------------------------------------------------------------------------------------------ */
typedef struct ojphgpu_encoder ojphgpu_encoder;
typedef struct ojphgpu_decoder ojphgpu_decoder;

int  ojphgpu_encoder_create(const ojphgpu_plan* plan, int device, void* stream, ojphgpu_encoder** out);
void ojphgpu_encoder_destroy(ojphgpu_encoder* enc);
/* An encoder that codes only tiles [tile_first, tile_first + tile_count) of the plan's frame: the
 * unit of multi-GPU sharding (one rank = one contiguous run of tiles).  d_image still addresses
 * the whole frame; only the rows/columns of the range's tiles are read. */
int  ojphgpu_encoder_create_tiles(const ojphgpu_plan* plan, int device, void* stream,
                                  uint32_t tile_first, uint32_t tile_count, ojphgpu_encoder** out);
/* D2H + host Tier-2 of the range: its tile-parts only (see ojphgpu_t2_write_tiles) */
int  ojphgpu_encoder_finish_tiles(ojphgpu_encoder* enc, uint8_t* h_out, size_t cap, size_t* out_len,
                                  uint32_t* tile_part_len);
/* the same with the tile-parts left in DEVICE memory (assembled there by a placement kernel; only the block
 * lengths visit the host): d_out receives *out_len bytes.  What a rank hands to the final codestream gather
 * of a multi-GPU encode (RCCL over xGMI) without a host round trip.  OJPHGPU_E_OVERFLOW + *out_len = need. */
int  ojphgpu_encoder_finish_tiles_device(ojphgpu_encoder* enc, uint8_t* d_out, size_t cap, size_t* out_len,
                                         uint32_t* tile_part_len);
/* An encoder for a BATCH of num_frames independent frames of the plan's shape (video: BASELINE
 * config "512 independent 4K frames"): one set of launches codes all of them, which is what fills
 * the GPU when a single frame does not.  d_image / h_image then hold the frames back to back
 * ([frame][comp][y][x]); every frame gets its own codestream from ojphgpu_encoder_finish_frame. */
int  ojphgpu_encoder_create_batch(const ojphgpu_plan* plan, int device, void* stream,
                                  uint32_t num_frames, ojphgpu_encoder** out);
int  ojphgpu_encoder_finish_frame(ojphgpu_encoder* enc, uint32_t frame, uint8_t* h_out, size_t cap,
                                  size_t* out_len);
/* device part only: d_image (int32 planes, resident in HBM) -> coded block bytes in HBM */
int  ojphgpu_encoder_run_device(ojphgpu_encoder* enc, const int32_t* d_image);
/* the frame in 16-bit containers (same plane layout, 2-byte elements; int16 for signed components,
 * uint16 otherwise; every component at most 16 bits deep) */
int  ojphgpu_encoder_run_device16(ojphgpu_encoder* enc, const uint16_t* d_image);
/* 8-bit containers (int8 for signed components, uint8 otherwise; every component at most 8 bits deep): how 8-bit
 * frames exist in files and capture buffers -- a quarter of the upload of int32 samples */
int  ojphgpu_encoder_run_device8(ojphgpu_encoder* enc, const uint8_t* d_image);
int  ojphgpu_encode16(ojphgpu_encoder* enc, const uint16_t* h_image, uint8_t* h_out, size_t cap, size_t* out_len);
/* D2H of block bytes + lengths, then host Tier-2 -> complete codestream */
int  ojphgpu_encoder_finish(ojphgpu_encoder* enc, uint8_t* h_out, size_t cap, size_t* out_len);
/* convenience: H2D + run_device + finish */
int  ojphgpu_encode(ojphgpu_encoder* enc, const int32_t* h_image, uint8_t* h_out, size_t cap,
                    size_t* out_len);
/* total coded bytes produced by the last run_device (synchronises the stream) */
int  ojphgpu_encoder_coded_bytes(ojphgpu_encoder* enc, uint64_t* bytes);

/* plan must come from ojphgpu_t2_parse over (h_codestream, len) */
int  ojphgpu_decoder_create(const ojphgpu_plan* plan, int device, void* stream, ojphgpu_decoder** out);
/* decodes only tiles [tile_first, tile_first + tile_count): writes their region of d_image */
int  ojphgpu_decoder_create_tiles(const ojphgpu_plan* plan, int device, void* stream,
                                  uint32_t tile_first, uint32_t tile_count, ojphgpu_decoder** out);
/* a decoder for a batch: plans[f] = ojphgpu_t2_parse of frame f's codestream; all frames must have
 * the same geometry.  Upload every frame's bytes with ojphgpu_decoder_upload_frame, then
 * ojphgpu_decoder_run_device writes d_image = [frame][comp][y][x]. */
int  ojphgpu_decoder_create_batch(const ojphgpu_plan* const* plans, uint32_t num_frames, int device,
                                  void* stream, ojphgpu_decoder** out);
int  ojphgpu_decoder_upload_frame(ojphgpu_decoder* dec, uint32_t frame, const uint8_t* h_codestream,
                                  size_t len);
void ojphgpu_decoder_destroy(ojphgpu_decoder* dec);
/* H2D of the codestream bytes */
int  ojphgpu_decoder_upload(ojphgpu_decoder* dec, const uint8_t* h_codestream, size_t len);
/* device part only: coded bytes in HBM -> d_image (int32 planes) */
int  ojphgpu_decoder_run_device(ojphgpu_decoder* dec, int32_t* d_image);
int  ojphgpu_decoder_run_device16(ojphgpu_decoder* dec, uint16_t* d_image);     /* 16-bit containers */
int  ojphgpu_decoder_run_device8(ojphgpu_decoder* dec, uint8_t* d_image);       /* 8-bit containers */
int  ojphgpu_decode16(ojphgpu_decoder* dec, const uint8_t* h_codestream, size_t len, uint16_t* h_image);
/* number of code-blocks that failed to decode in the last run (synchronises).  This call COLLECTS the run: when the
 * one-launch block decoder of that run gave up waiting for its own step-1 part (a chip held up for seconds by other
 * work; it marks the run instead of failing blocks), the frame is decoded again here through the separate launches,
 * into the same d_image -- so read d_image after this call, not before.  The reference decides per block at
 * codeblock::decode (ojph_codeblock.cpp:190-224); this is where its verdicts become visible.
 * A caller that consumes d_image on the device and never collects is told by a LATER ojphgpu_decoder_run_device* call,
 * which returns OJPHGPU_E_UNCOLLECTED (once per give-up, nothing enqueued) when a run before it had asked for the repeat.
 * The launch only says so when its wait has run out (about two seconds), so the notice may come more than one call
 * after the run it is about. */
int  ojphgpu_decoder_failed_blocks(ojphgpu_decoder* dec, uint32_t* count);
/* how many runs of this decoder were repeated that way (0 in any normal process) */
int  ojphgpu_decoder_fused_retries(ojphgpu_decoder* dec, uint32_t* count);
/* synchronises the decoder's stream; *current = the number of one-launch (fused) block-decoder runs enqueued so far,
 * *last_giveup = the number of the newest of them whose wait ran out (0 = none ever did).  A caller that times or pipelines
 * uncollected runs brackets them with two calls: no run in between gave up iff last_giveup <= the first call's *current. */
int  ojphgpu_decoder_giveup_epoch(ojphgpu_decoder* dec, uint32_t* last_giveup, uint32_t* current);
int  ojphgpu_decode(ojphgpu_decoder* dec, const uint8_t* h_codestream, size_t len,
                    int32_t* h_image);

/* timing of the last run_device, in milliseconds, measured with HIP events on the codec's own
 * stream: out[0]=convert+colour [1]=DWT [2]=HT block coder [3]=total */
/* per_launch = 1 (default): every launch is bracketed by HIP events so that the per-stage sums below
 * are available; 0: only the wall time of the run is recorded (out[3]), the other entries read 0 --
 * the event pairs cost a few microseconds each, which shows at these kernel durations */
int  ojphgpu_encoder_set_timing(ojphgpu_encoder* enc, int per_launch);
int  ojphgpu_decoder_set_timing(ojphgpu_decoder* dec, int per_launch);
int  ojphgpu_encoder_timing(ojphgpu_encoder* enc, float out[4]);
int  ojphgpu_decoder_timing(ojphgpu_decoder* dec, float out[4]);
/* the block decoder's three launches of the last run_device: out[0]=prep [1]=step 1 [2]=step 2 (ms) */
int  ojphgpu_decoder_ht_timing(ojphgpu_decoder* dec, float out[3]);
/* the block encoder's launches of the last run_device, in issue order: when the top resolution's
 * blocks are coded on the side stream (single frames with >= 2 decompositions) there are two --
 * [0] = those blocks (*n_top of them, concurrent with the lower DWT levels), [1] = the rest */
int  ojphgpu_encoder_ht_timing(ojphgpu_encoder* enc, float* out, uint32_t cap, uint32_t* n, uint32_t* n_top);
/* duration (ms) of every DWT level launch of the last run_device, in launch order (encode:
 * highest resolution first; decode: lowest first); *n = number of levels written */
int  ojphgpu_encoder_level_timing(ojphgpu_encoder* enc, float* out, uint32_t cap, uint32_t* n);
int  ojphgpu_decoder_level_timing(ojphgpu_decoder* dec, float* out, uint32_t cap, uint32_t* n);

/* ------------------------------------------------------------------------------------------ *
 * 6. Frame pipelines: sequences of frames of one shape with PCIe copies, kernels and host Tier-2 of
 *    consecutive frames overlapped (pinned staging, separate copy-in / compute / copy-out streams).
 *    In the reference one codestream object is re-used across the frames of a sequence through
 *    codestream::restart() (ojph_codestream.h:204, ojph_codestream_local.cpp:78-110): exchange() hands out
 *    line buffers, flush() writes the file (:1148-1270); read_headers() / create() / pull() read one
 *    (:769-1146, :1227-1270).  The pipe keeps that contract per frame -- the caller writes samples into
 *    memory the library hands out and receives a finished codestream, or the other way round -- and
 *    keeps `depth` (2..16) frames in flight.  One caller thread per pipe; results come back in
 *    submission order.  container_bits = 8 | 16 | 32 (see ojphgpu_encoder_run_device16 / 8); host_threads =
 *    threads that do a frame's host part (packet headers / parsing), 0 = 2.
 * ------------------------------------------------------------------------------------------ */
typedef struct ojphgpu_enc_pipe ojphgpu_enc_pipe;
typedef struct ojphgpu_dec_pipe ojphgpu_dec_pipe;

int  ojphgpu_enc_pipe_create(const ojphgpu_plan* plan, int device, uint32_t depth, int container_bits,
                             uint32_t host_threads, ojphgpu_enc_pipe** out);
void ojphgpu_enc_pipe_destroy(ojphgpu_enc_pipe* pipe);
/* pinned host memory for the next frame -- the frame layout of ojphgpu_plan_comp_info in container_bits-bit
 * elements -- which the caller fills (what exchange() hands out line by line).  OJPHGPU_E_AGAIN when all
 * `depth` slots are in flight: collect first. */
int  ojphgpu_enc_pipe_acquire(ojphgpu_enc_pipe* pipe, void** h_frame, size_t* bytes);
/* the acquired frame is complete (flush()): queues H2D -> kernels -> D2H of the block lengths -> packet
 * headers (host threads) -> codestream assembly on the device -> D2H, and returns at once */
int  ojphgpu_enc_pipe_submit(ojphgpu_enc_pipe* pipe);
/* the oldest submitted frame's codestream (SOC .. EOC, byte-identical to ojphgpu_encode's); blocks until it
 * is complete.  *h_codestream is pinned memory of the pipe, valid until the next _collect / _destroy. */
int  ojphgpu_enc_pipe_collect(ojphgpu_enc_pipe* pipe, const uint8_t** h_codestream, size_t* len);
/* out[0] frames completed, [1] mean host Tier-2 time per frame (ms), [2] mean submit -> codestream latency
 * (ms), [3] host threads coding the packet headers of one frame */
int  ojphgpu_enc_pipe_stats(ojphgpu_enc_pipe* pipe, double out[4]);

/* The frames of this pipe are handed over PIXEL-INTERLEAVED (R G B R G B ..., the order of .ppm files and of capture
 * buffers; ppm_in::read, src/apps/others/ojph_img_io.cpp:338-375) instead of as planes: pixel_bits = 8 or 16 per
 * sample, 16-bit samples big endian (as in the files) when big_endian != 0.  _acquire then hands out width * height *
 * components * pixel_bits / 8 bytes, and a launch on the device turns them into the planar containers
 * (ojphgpu_unpack_pixels).  Components must have one size and be unsigned, pixel_bits must hold the bit depth and
 * must not exceed container_bits.  Call before the first _acquire; pixel_bits = 0 switches back to planes. */
int  ojphgpu_enc_pipe_set_pixels(ojphgpu_enc_pipe* pipe, int pixel_bits, int big_endian);
/* ... or as planes of bit-packed samples (ojphgpu_unpack_bits: bits = 10, 12, 14; the whole frame one bit string in
 * the plane order of the planar layout): _acquire hands out ceil(samples * bits / 8) bytes rounded up to whole groups
 * of 32 samples.  Unsigned components whose depth fits `bits`, 16- or 32-bit containers.  bits = 0 switches back. */
int  ojphgpu_enc_pipe_set_packed(ojphgpu_enc_pipe* pipe, int bits);

/* the first codestream of the sequence fixes the frame geometry (it is only parsed, not decoded); every
 * submitted codestream must describe the same frame format and code-block grid (quantisation may differ) */
int  ojphgpu_dec_pipe_create(const uint8_t* h_codestream, size_t len, int resilient, int device, uint32_t depth,
                             int container_bits, uint32_t host_threads, ojphgpu_dec_pipe** out);
void ojphgpu_dec_pipe_destroy(ojphgpu_dec_pipe* pipe);
int  ojphgpu_dec_pipe_plan(ojphgpu_dec_pipe* pipe, const ojphgpu_plan** plan);   /* owned by the pipe */
/* pinned host memory for the next codestream of `len` bytes (read the file straight into it) */
int  ojphgpu_dec_pipe_acquire(ojphgpu_dec_pipe* pipe, size_t len, uint8_t** h_codestream);
/* queues parse (host threads) -> H2D of the code-block bytes + descriptors -> kernels -> D2H of the frame */
int  ojphgpu_dec_pipe_submit(ojphgpu_dec_pipe* pipe);
/* the oldest submitted frame: container_bits-bit samples in the layout of ojphgpu_plan_comp_info, in
 * pinned memory valid until the next _collect / _destroy.  OJPHGPU_E_BLOCK when code-blocks failed and the
 * pipe is not resilient (the frame is still handed out, failed blocks zeroed, *failed_blocks counts them). */
int  ojphgpu_dec_pipe_collect(ojphgpu_dec_pipe* pipe, const void** h_frame, size_t* bytes, uint32_t* failed_blocks);
/* out[0] frames completed, [1] mean host parse time per frame (ms), [2] mean submit -> frame latency (ms), [3] host threads */
int  ojphgpu_dec_pipe_stats(ojphgpu_dec_pipe* pipe, double out[4]);
/* frames this pipe decoded twice because the one-launch block decoder asked for it (ojphgpu_decoder_failed_blocks) */
int  ojphgpu_dec_pipe_fused_retries(ojphgpu_dec_pipe* pipe, uint32_t* count);
/* decoded frames come back pixel-interleaved, clamped to [0, 2^depth - 1] -- ONE depth for the frame: the components
 * must share their bit depth (a .ppm has one maxval), otherwise OJPHGPU_E_INVALID -- (ppm_out::write and its converters,
 * ojph_img_io.cpp:99-226, :539-556); same conditions as ojphgpu_enc_pipe_set_pixels; call before the first _submit */
int  ojphgpu_dec_pipe_set_pixels(ojphgpu_dec_pipe* pipe, int pixel_bits, int big_endian);
int  ojphgpu_dec_pipe_set_packed(ojphgpu_dec_pipe* pipe, int bits);     /* decoded frames come back bit-packed, clamped to [0, 2^bits - 1] */

/* ---------------------------------------------------------------------------------------------
 * 7. Pixel-interleaved samples <-> planar containers on the device (kernels_pixels.hip)
 * ------------------------------------------------------------------------------------------- */
/* d_pixels: width * height pixels of num_comps samples each, pixel_bits = 8 or 16 bits per sample (16-bit samples
 * byte-swapped when big_endian != 0); d_planes: num_comps planes of width * height samples in container_bits-bit
 * elements (8, 16, 32; not narrower than pixel_bits).  Replaces the per-sample loops of the reference's image
 * readers (ppm_in::read, ojph_img_io.cpp:338-375). */
int  ojphgpu_unpack_pixels(void* stream, const void* d_pixels, void* d_planes, uint32_t width, uint32_t height,
                           uint32_t num_comps, int pixel_bits, int big_endian, int container_bits);
/* the way back, values clamped to [0, 2^bit_depth - 1] (gen_cvrt_32b*_to_*, ojph_img_io.cpp:99-226) */
int  ojphgpu_pack_pixels(void* stream, const void* d_planes, void* d_pixels, uint32_t width, uint32_t height,
                         uint32_t num_comps, int container_bits, int pixel_bits, int big_endian, uint32_t bit_depth);
/* Bit-packed samples: 10, 12 or 14 bits each, consecutive in one little-endian bit string (sample i = bits
 * [i * bits, (i + 1) * bits)), the buffer padded to a multiple of 32 samples (4 * bits bytes, 4-byte aligned) <-> 16- or
 * 32-bit containers.  A 12-bit frame crosses PCIe in 1.5 instead of 2 bytes per sample this way (the link is what
 * bounds the frame pipelines).  Packing clamps to [0, 2^bits - 1].  No counterpart in the reference, whose files hold
 * such samples in 16-bit words. */
int  ojphgpu_unpack_bits(void* stream, const void* d_packed, void* d_samples, uint64_t num_samples, int bits, int container_bits);
int  ojphgpu_pack_bits(void* stream, const void* d_samples, void* d_packed, uint64_t num_samples, int container_bits, int bits);

const char* ojphgpu_version(void);

/* ---------------------------------------------------------------------------------------------
 * 8. One frame over several GPUs of the node, from one process (ojphgpu_multi.cpp)
 * ---------------------------------------------------------------------------------------------
 * Tiles are independent (every tile has an object tree of its own in the reference, ojph_codestream_local.cpp:113-180, and
 * codestream::flush writes them one after the other, ojph_tile.cpp:584-610): a tiled frame shards by contiguous runs of
 * tiles, device k of n taking tiles [k * T / n ...) -- the first T % n runs one tile longer -- with one host thread and one
 * codec object per device and NO exchange between the devices while they code.  The only meeting point is the caller's
 * thread: main header from everybody's Psot lengths, a prefix sum of the runs' lengths, and every device copies its
 * tile-parts straight to its place in the caller's buffer (encode) / its tiles' rectangles to the caller's image (decode).
 * A frame of one tile uses the first device only.  h_image: the frame as int32 planes (ojphgpu_plan_comp_info); host
 * buffers should be pinned for the copies to run at link speed.  The same device may be listed more than once. */
typedef struct ojphgpu_multi_encoder ojphgpu_multi_encoder;
typedef struct ojphgpu_multi_decoder ojphgpu_multi_decoder;
int  ojphgpu_multi_encoder_create(const ojphgpu_plan* plan, const int* devices, uint32_t num_devices, ojphgpu_multi_encoder** out);
void ojphgpu_multi_encoder_destroy(ojphgpu_multi_encoder* enc);
/* whole codestream (SOC .. EOC) into h_out; OJPHGPU_E_OVERFLOW with *out_len = the bytes needed when cap is too small */
int  ojphgpu_multi_encode(ojphgpu_multi_encoder* enc, const int32_t* h_image, uint8_t* h_out, size_t cap, size_t* out_len);
/* the same with the frame's samples in container_bits-bit containers (32 | 16 | 8, as ojphgpu_encoder_run_device16 / 8):
 * half / a quarter of the bytes over every device's link */
int  ojphgpu_multi_encode_container(ojphgpu_multi_encoder* enc, const void* h_image, int container_bits, uint8_t* h_out, size_t cap,
                                    size_t* out_len);
/* how the frame was dealt out: workers (<= num_devices, <= tiles) and the tiles of each */
int  ojphgpu_multi_encoder_workers(const ojphgpu_multi_encoder* enc, uint32_t* num_workers, uint32_t* tiles_per_worker, uint32_t cap);
/* parses the codestream once (codestream::read_headers + read, with restrict_input_resolution when the skips are not 0) */
int  ojphgpu_multi_decoder_create(const uint8_t* h_codestream, size_t len, int resilient, uint32_t skipped_res_for_data,
                                  uint32_t skipped_res_for_recon, const int* devices, uint32_t num_devices, ojphgpu_multi_decoder** out);
void ojphgpu_multi_decoder_destroy(ojphgpu_multi_decoder* dec);
int  ojphgpu_multi_decoder_plan(ojphgpu_multi_decoder* dec, const ojphgpu_plan** plan);      /* owned by the decoder */
/* the codestream the decoder was created for -> h_image; OJPHGPU_E_BLOCK when blocks failed and the decoder is not resilient */
int  ojphgpu_multi_decode(ojphgpu_multi_decoder* dec, const uint8_t* h_codestream, size_t len, int32_t* h_image, uint32_t* failed_blocks);
/* Host memory the caller got elsewhere (a shared-memory segment several processes write one codestream into, a file
 * mapping) made known to the HIP runtime this library runs on, so that copies from / to it run at link speed
 * (hipHostRegister / hipHostUnregister).  openjph_amd/shard.py HostGather: every rank of a node copies its tile-parts over
 * its own GPU's link straight to their place in ONE segment -- the gather of SURVEY 8(e) without a receiving GPU. */
int  ojphgpu_host_register(void* h_ptr, size_t bytes);
int  ojphgpu_host_unregister(void* h_ptr);
int  ojphgpu_multi_decode_container(ojphgpu_multi_decoder* dec, const uint8_t* h_codestream, size_t len, void* h_image, int container_bits,
                                    uint32_t* failed_blocks);

#ifdef __cplusplus
}
#endif
#endif /* OJPHGPU_H */
