// include/ojph_gpu_codestream.h -- ojph::codestream-compatible C++ facade over the C ABI
// (include/ojphgpu.h).  Same class names, method names, argument meaning and error behaviour as
// the public interface of aous72/OpenJPH 0.31.0 for the calls ojph_compress / ojph_expand make:
//   ojph::codestream      src/core/openjph/ojph_codestream.h:88-383
//   ojph::param_siz/cod/qcd, comment_exchange   src/core/openjph/ojph_params.h:68-361
//   ojph::line_buf        src/core/openjph/ojph_mem.h:160-192
//   ojph::outfile_base / j2c_outfile / mem_outfile / infile_base / j2c_infile / mem_infile
//                         src/core/openjph/ojph_file.h:74-349
// so that a program written against the reference compiles against this header and links
// libopenjph_gpu.so instead.  The implementation is new: where the reference streams lines
// through a tile / resolution / sub-band / code-block object tree, this facade collects the frame
// in one host buffer and hands it to the batched GPU path at flush() (encode) or on the
// first pull() (decode).  Errors are reported as the reference reports OJPH_ERROR:
// std::runtime_error("ojph error") after the message went to stderr.
//
// SURVEY.md section 8(f) N4 through this interface: components of up to 32 bits (the reference's 64-bit sample path, taken by
// components that need more than 32 bits of precision inside the codec) are written and read -- the lines exchange() / pull()
// hand out stay si32 (LFT_32BIT | LFT_INTEGER) exactly as the reference's do (ojph_codestream_local.cpp:178, :279: its
// application-side lines are si32 at every bit depth; 64-bit lines exist only inside its tile components); codestreams with
// Part-2 wavelets (DFS / ATK marker segments), SigProp / MagRef passes and the vertically causal block style are READ (the
// reference has no writer for them: param_cod has no DFS / ATK setter, ojph_params.h:68-361), and
// param_cod::get_block_vertical_causality() reports what the COD / COC said.  Also: per-component coding styles (COC:
// param_cod's comp_idx setters), per-component quantisation (QCC: set_qfactor(comp, ..), set_irrev_quant(comp, ..)), NLT type
// 3, sub-sampling, components of different bit depth / signedness, image and tile offsets, tile-part divisions, user COM
// markers, qfactor, the IMF / BROADCAST profile checks, reduced-resolution and resilient decoding.  What the GPU path cannot
// take fails loudly in write_headers() / read_headers() / the setter (std::runtime_error after the message, like OJPH_ERROR).
// A restart()ed object codes a sequence of frames through a frame pipeline that outlives restart()
// (pinned frame / codestream buffers, device objects made once per frame format).
#ifndef OJPH_GPU_CODESTREAM_H
#define OJPH_GPU_CODESTREAM_H

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace ojph {

typedef uint8_t ui8;   typedef int8_t si8;
typedef uint16_t ui16; typedef int16_t si16;
typedef uint32_t ui32; typedef int32_t si32;
typedef uint64_t ui64; typedef int64_t si64;

struct size { explicit size(ui32 w = 0, ui32 h = 0) : w(w), h(h) {} ui32 w, h; ui64 area() const { return (ui64)w * h; } };
struct point { explicit point(ui32 x = 0, ui32 y = 0) : x(x), y(y) {} ui32 x, y; };

// one image line handed to / received from the codestream (ojph_mem.h:160-192)
class line_buf {
public:
  enum : ui32 { LFT_UNDEFINED = 0x00, LFT_BYTE = 0x01, LFT_16BIT = 0x02, LFT_32BIT = 0x04, LFT_64BIT = 0x08,
                LFT_INTEGER = 0x10, LFT_SIZE_MASK = 0x0F };
  line_buf() : size(0), pre_size(0), flags(LFT_UNDEFINED), i32(nullptr) {}
  size_t size;
  ui32 pre_size;
  ui32 flags;
  union { si32* i32; si64* i64; float* f32; void* p; };
};

// ---- files (ojph_file.h) ---------------------------------------------------------------------
class outfile_base {
public:
  enum seek : int { OJPH_SEEK_SET = SEEK_SET, OJPH_SEEK_CUR = SEEK_CUR, OJPH_SEEK_END = SEEK_END };
  virtual ~outfile_base() {}
  virtual size_t write(const void* ptr, size_t size) = 0;
  virtual si64 tell() { return 0; }
  virtual int seek(si64, enum outfile_base::seek) { return -1; }
  virtual void flush() {}
  virtual void close() {}
};

class j2c_outfile : public outfile_base {
public:
  j2c_outfile() : fh(nullptr) {}
  ~j2c_outfile() override { if (fh) fclose(fh); }
  void open(const char* filename);
  size_t write(const void* ptr, size_t size) override;
  si64 tell() override;
  void flush() override;
  void close() override;
private:
  FILE* fh;
};

class mem_outfile : public outfile_base {
public:
  mem_outfile() : pos(0), is_open(false) {}
  void open(size_t initial_size = 65536, bool clear_mem = false);
  size_t write(const void* ptr, size_t size) override;
  si64 tell() override { return (si64)pos; }
  int seek(si64 offset, enum outfile_base::seek origin) override;
  void close() override { is_open = false; }
  const ui8* get_data() { return buf.data(); }
  const ui8* get_data() const { return buf.data(); }
  size_t get_used_size() const { return buf.size(); }
  void write_to_file(const char* file_name) const;
private:
  std::vector<ui8> buf; size_t pos; bool is_open;
};

class infile_base {
public:
  enum seek : int { OJPH_SEEK_SET = SEEK_SET, OJPH_SEEK_CUR = SEEK_CUR, OJPH_SEEK_END = SEEK_END };
  virtual ~infile_base() {}
  virtual size_t read(void* ptr, size_t size) = 0;   // returns the number of bytes read
  virtual int seek(si64 offset, enum infile_base::seek origin) = 0;
  virtual si64 tell() = 0;
  virtual bool eof() = 0;
  virtual void close() {}
};

class j2c_infile : public infile_base {
public:
  j2c_infile() : fh(nullptr) {}
  ~j2c_infile() override { if (fh) fclose(fh); }
  void open(const char* filename);
  size_t read(void* ptr, size_t size) override;
  int seek(si64 offset, enum infile_base::seek origin) override;
  si64 tell() override;
  bool eof() override { return feof(fh) != 0; }
  void close() override;
private:
  FILE* fh;
};

class mem_infile : public infile_base {
public:
  mem_infile() : data(nullptr), cur(nullptr), sz(0) {}
  void open(const ui8* data, size_t size) { this->data = cur = data; sz = size; }
  size_t read(void* ptr, size_t size) override;
  int seek(si64 offset, enum infile_base::seek origin) override;
  si64 tell() override { return cur - data; }
  bool eof() override { return cur >= data + sz; }
  void close() override { data = cur = nullptr; sz = 0; }
private:
  const ui8* data; const ui8* cur; size_t sz;
};

// ---- parameters (ojph_params.h) -----------------------------------------------------------------
namespace local { struct codestream_state; }

class param_siz {
public:
  explicit param_siz(local::codestream_state* s) : state(s) {}
  void set_image_extent(point extent);
  void set_tile_size(size s);
  void set_image_offset(point offset);
  void set_tile_offset(point offset);
  void set_num_components(ui32 num_comps);
  void set_component(ui32 comp_num, const point& downsampling, ui32 bit_depth, bool is_signed);
  point get_image_extent() const;
  point get_image_offset() const;
  size get_tile_size() const;
  point get_tile_offset() const;
  ui32 get_num_components() const;
  ui32 get_bit_depth(ui32 comp_num) const;
  bool is_signed(ui32 comp_num) const;
  point get_downsampling(ui32 comp_num) const;
  ui32 get_recon_width(ui32 comp_num) const;
  ui32 get_recon_height(ui32 comp_num) const;
private:
  local::codestream_state* state;
};

class param_cod {
public:
  explicit param_cod(local::codestream_state* s) : state(s) {}
  void set_num_decomposition(ui32 num_decompositions);
  void set_block_dims(ui32 width, ui32 height);
  void set_precinct_size(int num_levels, size* precinct_size);
  void set_progression_order(const char* name);
  void set_color_transform(bool color_transform);
  void set_reversible(bool reversible);
  ui32 get_num_decompositions() const;
  size get_block_dims() const;
  size get_log_block_dims() const;
  bool is_reversible() const;
  size get_precinct_size(ui32 level_num) const;
  size get_log_precinct_size(ui32 level_num) const;
  int get_progression_order() const;
  const char* get_progression_order_as_string() const;
  int get_num_layers() const { return 1; }
  bool is_using_color_transform() const;
  bool packets_may_use_sop() const { return false; }
  bool packets_use_eph() const { return false; }
  bool get_block_vertical_causality() const;         // the COD's code-block style bit 3, as parsed (ojph_params.cpp:368-370)
  // COC marker segments (ojph_params.h:146-158): the first call for a component creates its COC from
  // the SPcod defaults (5 decompositions, 64x64 blocks, wavelet_trans 0 = the 9/7, 32768x32768
  // precincts; ojph_params_local.h:344-353), not from the COD; later calls modify it.  Components 0..15.
  void set_num_decomposition(ui32 comp_idx, ui32 num_decompositions);
  void set_block_dims(ui32 comp_idx, ui32 width, ui32 height);
  void set_precinct_size(ui32 comp_idx, int num_levels, size* precinct_size);
  void set_reversible(ui32 comp_idx, bool reversible);
  ui32 get_num_decompositions(ui32 comp_idx) const;
  size get_block_dims(ui32 comp_idx) const;
  size get_log_block_dims(ui32 comp_idx) const;
  bool is_reversible(ui32 comp_idx) const;
  size get_precinct_size(ui32 comp_idx, ui32 level_num) const;
  size get_log_precinct_size(ui32 comp_idx, ui32 level_num) const;
  bool get_block_vertical_causality(ui32 comp_idx) const;   // the component's COC if it has one, else the COD (ojph_params.cpp:396-399)
private:
  local::codestream_state* state;
};

class param_qcd {
public:
  explicit param_qcd(local::codestream_state* s) : state(s) {}
  void set_irrev_quant(float delta);
  void set_qfactor(ui8 qfactor);              // 1..100: visually weighted step sizes, a QCC per component
  enum comp_type : ui8 { OJPH_COMP_Y = 0, OJPH_COMP_CB = 1, OJPH_COMP_CR = 2, OJPH_COMP_UNDEFINED = 0xFF };
  static comp_type ui8_2_comp_type(ui8 c) { return c <= OJPH_COMP_CR ? static_cast<comp_type>(c) : OJPH_COMP_UNDEFINED; }
  // a QCC for one component (0..15) with its own quality factor and visual weights (ojph_params.h:251-254)
  void set_qfactor(ui32 comp_idx, comp_type ctype, ui8 qfactor);
  // as in the reference (ojph_params.cpp:2011-2018: get_qcc never returns NULL), this reaches the
  // component's QCC only when set_qfactor(comp_idx, ..) made one -- where the quality factor then
  // wins -- and otherwise sets the base step of the QCD
  void set_irrev_quant(ui32 comp_idx, float delta);
private:
  local::codestream_state* state;
};

// NLT marker segments (ojph_params.h:299-345): type 0 (none) and type 3 (binary complement <-> sign
// magnitude for signed components) are what the reference supports; entries for components 0..15
class param_nlt {
public:
  enum special_comp_num : ui16 { ALL_COMPS = 65535 };
  enum nonlinearity : ui8 { OJPH_NLT_NO_NLT = 0, OJPH_NLT_GAMMA_STYLE_NLT = 1, OJPH_NLT_LUT_STYLE_NLT = 2,
                            OJPH_NLT_BINARY_COMPLEMENT_NLT = 3, OJPH_NLT_UNDEFINED = 255 };
  explicit param_nlt(local::codestream_state* s) : state(s) {}
  void set_nonlinear_transform(ui32 comp_num, ui8 nl_type);
  bool get_nonlinear_transform(ui32 comp_num, ui8& bit_depth, bool& is_signed, ui8& nl_type) const;
private:
  local::codestream_state* state;
};

class comment_exchange {
public:
  comment_exchange() : data(nullptr), len(0), Rcom(0) {}
  void set_string(const char* str);
  void set_data(const char* data, ui16 len);
private:
  friend class codestream;
  const char* data; ui16 len; ui16 Rcom;
};

// ---- the codestream (ojph_codestream.h:88-383) -------------------------------------------------
class codestream {
public:
  codestream();
  ~codestream();
  void restart();

  void set_planar(bool planar);
  void set_profile(const char* s);
  void set_tilepart_divisions(bool at_resolutions, bool at_components);
  bool is_tilepart_division_at_resolutions();
  bool is_tilepart_division_at_components();
  void request_tlm_marker(bool needed);
  bool is_tlm_requested();

  void write_headers(outfile_base* file, const comment_exchange* comments = nullptr, ui32 num_comments = 0);
  line_buf* exchange(line_buf* line, ui32& next_component);
  void flush();

  void enable_resilience();
  void read_headers(infile_base* file);
  void restrict_input_resolution(ui32 skipped_res_for_data, ui32 skipped_res_for_recon);
  void create();
  line_buf* pull(ui32& comp_num);

  void close();

  param_siz access_siz();
  param_cod access_cod();
  param_qcd access_qcd();
  param_nlt access_nlt();
  bool is_planar() const;

  // GPU-path extras (not in the reference): device ordinal used by flush() / create()
  void set_device(int device);
  // several GPUs of the node for ONE frame: a tiled frame's tiles are dealt out to them in contiguous runs, one host
  // thread and codec object per device (include/ojphgpu.h section 8; tiles are independent in the reference too,
  // ojph_codestream_local.cpp:113-180); frames of one tile use the first device
  void set_devices(const int* devices, ui32 num_devices);
  // restart()ed sequences keep their frames in pinned slots; by default these hold int32 samples like the reference's
  // line_buf.  true: the narrowest container the bit depth fits (8 / 16 bits: half or a quarter of the PCIe traffic) --
  // samples outside the bit depth's range handed to exchange() then saturate, and so do the values pull() returns
  void set_narrow_sample_containers(bool narrow);
  // A restart()ed object codes a sequence of frames through a frame pipeline (pinned slots, persistent device objects;
  // int32 slots unless set_narrow_sample_containers).  By default flush() returns with the codestream in the file, as
  // the reference's does.  enable_frame_pipelining(n >= 2): flush() only queues the frame -- upload, kernels, Tier-2 and
  // download of frame k run while the application fills frame k + 1 -- and its codestream is written to the outfile
  // given to ITS write_headers() when a later write_headers() needs the slot or names the same outfile object, at
  // drain(), at close() of that outfile (a closed file holds its codestream: the object may be opened again for the next
  // frame right away), or when the codestream object is destroyed.  To keep frames in flight across frames, give every
  // frame an outfile object of its own and close() it after drain().
  void enable_frame_pipelining(ui32 frames_in_flight = 4);
  void drain();

private:
  codestream(const codestream&) = delete;
  codestream& operator=(const codestream&) = delete;
  local::codestream_state* state;
};

}  // namespace ojph
#endif
