// openjph_amd/csrc/ojph_facade.cpp -- implementation of include/ojph_gpu_codestream.h: the
// ojph::codestream-compatible C++ facade, built on the C ABI of libojphgpu.so.
//
// Behaviour mirrored from the reference (cited per function); the mechanism is different: the
// frame is collected in one pinned host buffer (exchange() hands out pointers straight into it, so
// the application writes its samples in place and nothing is copied), flush() runs the batched GPU
// encoder and writes the codestream, read_headers() slurps the codestream and parses it on the
// host, the first pull() runs the batched GPU decoder into the frame buffer and every pull()
// returns a pointer to one of its rows.
#include "../../include/ojph_gpu_codestream.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <stdexcept>

#include "../../include/ojphgpu.h"

namespace ojph {

namespace {

// OJPH_ERROR of the reference (ojph_message.cpp:156-171): message to stderr, then
// std::runtime_error("ojph error")
[[noreturn]] void ojph_error(ui32 code, const char* fmt, ...)
{
  char msg[512];
  va_list ap; va_start(ap, fmt); vsnprintf(msg, sizeof(msg), fmt, ap); va_end(ap);
  fprintf(stderr, "ojph error 0x%08X: %s\n", code, msg);
  throw std::runtime_error("ojph error");
}

const char* const PROG_NAMES[5] = { "LRCP", "RLCP", "RPCL", "PCRL", "CPRL" };

ui32 log2_exact(ui32 v) { ui32 l = 0; while ((1u << l) < v) ++l; return l; }

}  // namespace

// ---- files ---------------------------------------------------------------------------------------
void j2c_outfile::open(const char* filename)
{
  fh = fopen(filename, "wb");
  if (!fh) ojph_error(0x00060001, "failed to open %s for writing", filename);        // ojph_file.cpp:63
}
size_t j2c_outfile::write(const void* ptr, size_t size) { return fwrite(ptr, 1, size, fh); }
si64 j2c_outfile::tell() { return (si64)ftello(fh); }
void j2c_outfile::flush() { fflush(fh); }
void j2c_outfile::close() { if (fh) fclose(fh); fh = nullptr; }

void mem_outfile::open(size_t initial_size, bool) { buf.clear(); buf.reserve(initial_size); pos = 0; is_open = true; }
size_t mem_outfile::write(const void* ptr, size_t size)
{
  if (pos + size > buf.size()) buf.resize(pos + size);
  memcpy(buf.data() + pos, ptr, size);
  pos += size;
  return size;
}
int mem_outfile::seek(si64 offset, enum outfile_base::seek origin)
{
  si64 np = origin == OJPH_SEEK_SET ? offset : origin == OJPH_SEEK_CUR ? (si64)pos + offset : (si64)buf.size() + offset;
  if (np < 0 || (size_t)np > buf.size()) return -1;
  pos = (size_t)np;
  return 0;
}
void mem_outfile::write_to_file(const char* file_name) const
{
  FILE* f = fopen(file_name, "wb");
  if (!f) ojph_error(0x00060003, "failed to open %s for writing", file_name);
  if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); ojph_error(0x00060004, "failed writing to %s", file_name); }
  fclose(f);
}

void j2c_infile::open(const char* filename)
{
  fh = fopen(filename, "rb");
  if (!fh) ojph_error(0x00060002, "failed to open %s for reading", filename);        // ojph_file.cpp:208
}
size_t j2c_infile::read(void* ptr, size_t size) { return fread(ptr, 1, size, fh); }
int j2c_infile::seek(si64 offset, enum infile_base::seek origin) { return fseeko(fh, (off_t)offset, origin); }
si64 j2c_infile::tell() { return (si64)ftello(fh); }
void j2c_infile::close() { if (fh) fclose(fh); fh = nullptr; }

size_t mem_infile::read(void* ptr, size_t size)
{
  size_t avail = (size_t)(data + sz - cur);
  size_t n = size < avail ? size : avail;
  memcpy(ptr, cur, n); cur += n;
  return n;
}
int mem_infile::seek(si64 offset, enum infile_base::seek origin)
{
  const ui8* np = origin == OJPH_SEEK_SET ? data + offset : origin == OJPH_SEEK_CUR ? cur + offset : data + sz + offset;
  if (np < data || np > data + sz) return -1;
  cur = np;
  return 0;
}

// ---- state ---------------------------------------------------------------------------------------
namespace local {

struct comp_info { point ds; ui32 bit_depth; bool is_signed; bool set; };

struct codestream_state {
  ojphgpu_params p;
  std::vector<comp_info> comps;
  point image_offset, tile_offset;
  int planar = -1;                    // -1: not chosen (ojph_codestream_local.cpp:89)
  ui32 skip_recon = 0;                // resolutions left out of the reconstruction (restrict_input_resolution)
  bool resilient = false;
  bool headers_written = false, headers_read = false, decoded = false;
  std::string profile;
  int device = 0;
  // more than one device: a tiled frame is coded by all of them, each a contiguous run of tiles (include/ojphgpu.h section 8)
  std::vector<int> devices;
  ojphgpu_multi_encoder* menc = nullptr; ojphgpu_multi_decoder* mdec = nullptr;
  ui32 skip_data = 0;
  outfile_base* outfile = nullptr;
  infile_base* infile = nullptr;
  ojphgpu_plan* plan = nullptr;
  ojphgpu_encoder* enc = nullptr;
  ojphgpu_decoder* dec = nullptr;
  // An object that is restart()ed codes a SEQUENCE of frames (ojph_codestream.h:204): from the second frame on it
  // works through a frame pipeline (include/ojphgpu.h section 6) that outlives restart() -- pinned frame and
  // codestream buffers, device arena, descriptor tables and streams are made once per frame format, the
  // application's lines land directly in the pinned memory the upload reads.
  bool sequence = false;              // restart() has been called
  ojphgpu_enc_pipe* epipe = nullptr; ojphgpu_plan* epipe_plan = nullptr; ojphgpu_params epipe_params;
  std::vector<std::vector<ui8>> epipe_comments;
  ojphgpu_dec_pipe* dpipe = nullptr; bool dpipe_resilient = false;
  bool frame_of_pipe = false;         // `frame` is memory of a pipe slot: not ours to free
  // Frames of a sequence cross PCIe in the narrowest container their samples fit (8 / 16 / 32 bits, include/ojphgpu.h
  // section 6): the lines exchange() / pull() hand out are int32 staging rows (the reference's line_buf contract), a row
  // is narrowed into the pinned slot when the application hands it back, widened when it is pulled.
  int pipe_bits = 32;                 // container of the pipe's slots
  ui8* slot = nullptr;                // != nullptr: the frame lives in a slot of pipe_bits-bit samples, rows are staged
  std::vector<std::vector<si32>> stage;
  // enable_frame_pipelining(): flush() only queues the frame; its codestream is written to ITS outfile when a later
  // flush() needs the slot, at drain(), or when the object goes away
  struct Pending { outfile_base* file; bool close_after; };
  std::vector<Pending> pending;
  ui32 pipelining = 0;                // 0: flush() returns with the codestream written (the reference's contract)
  bool restricted = false;            // restrict_input_resolution was called for this frame
  si32* frame = nullptr; bool frame_pinned = false; size_t frame_elems = 0;
  std::vector<ui32> cw, ch; std::vector<size_t> coff;   // component planes inside the frame (ojphgpu_plan_comp_info)
  std::vector<si32> spare;                              // a line nobody reads (interleaved exchange past a short component)
  std::vector<ui8> stream;            // the whole codestream (decode)
  std::vector<line_buf> lines;
  ui32 cur_comp = 0, cur_line = 0;
  bool exhausted = false;

  codestream_state() { reset_params(); }
  void reset_params()
  {
    memset(&p, 0, sizeof(p));
    p.reversible = 0; p.num_decomps = 5; p.block_w = 64; p.block_h = 64;      // param_cod defaults (ojph_params_local.h:560-575)
    p.prog_order = 2; p.qstep = -1.0f;                                      // RPCL; qstep chosen from the bit depth
    comps.clear(); image_offset = point(0, 0); tile_offset = point(0, 0); skip_recon = 0;
  }
  void release_pipes()
  {
    try { drain(); } catch (...) {}
    if (epipe) { ojphgpu_enc_pipe_destroy(epipe); epipe = nullptr; }
    if (epipe_plan) { ojphgpu_plan_destroy(epipe_plan); epipe_plan = nullptr; }
    if (dpipe) { ojphgpu_dec_pipe_destroy(dpipe); dpipe = nullptr; }
  }
  void release()                       // what one frame owned; the pipes stay (see `sequence`)
  {
    if (enc) { ojphgpu_encoder_destroy(enc); enc = nullptr; }
    if (dec) { ojphgpu_decoder_destroy(dec); dec = nullptr; }
    if (menc) { ojphgpu_multi_encoder_destroy(menc); menc = nullptr; }
    if (mdec) { ojphgpu_multi_decoder_destroy(mdec); mdec = nullptr; }
    if (plan && plan != epipe_plan) ojphgpu_plan_destroy(plan);
    plan = nullptr;
    if (frame && !frame_of_pipe) { if (frame_pinned) (void)hipHostFree(frame); else free(frame); }
    frame = nullptr; frame_of_pipe = false; restricted = false; slot = nullptr;
    frame_elems = 0; stream.clear(); lines.clear();
    headers_written = headers_read = decoded = false; exhausted = false; cur_comp = cur_line = 0;
    outfile = nullptr; infile = nullptr;
  }
  void alloc_frame(si32* pipe_frame = nullptr)
  {
    cw.assign(p.num_comps, 0); ch.assign(p.num_comps, 0); coff.assign(p.num_comps, 0);
    ui32 info[8], widest = 0;
    for (ui32 c = 0; c < p.num_comps; ++c) {
      ojphgpu_plan_comp_info(plan, c, info);
      cw[c] = info[2]; ch[c] = info[3]; coff[c] = (size_t)info[4] | ((size_t)info[5] << 32);
      widest = cw[c] > widest ? cw[c] : widest;
    }
    ojphgpu_plan_comp_info(plan, p.num_comps, info);
    frame_elems = (size_t)info[4] | ((size_t)info[5] << 32);
    spare.assign(widest, 0);
    // one frame per codestream object: pinning 400 MB costs more than the pageable copy it would
    // speed up (measured on the MI355X host: both ~57 GB/s), so the frame is plain memory; set
    // OJPH_GPU_PIN=1 for long-lived objects that restart() and reuse their buffers
    void* ptr = nullptr;
    if (pipe_frame && pipe_bits != 32) {                  // rows are staged (one int32 row per component)
      slot = (ui8*)pipe_frame; frame_of_pipe = true; frame_pinned = true;
      stage.resize(p.num_comps);
      for (ui32 c = 0; c < p.num_comps; ++c) stage[c].assign(cw[c] + 8, 0);
      lines.assign(p.num_comps, line_buf());
      for (ui32 c = 0; c < p.num_comps; ++c) {
        lines[c].size = cw[c]; lines[c].pre_size = 0; lines[c].flags = line_buf::LFT_32BIT | line_buf::LFT_INTEGER;
      }
      return;
    }
    if (pipe_frame) { frame = pipe_frame; frame_of_pipe = true; frame_pinned = true; }
    else if (getenv("OJPH_GPU_PIN") && hipHostMalloc(&ptr, frame_elems * sizeof(si32), hipHostMallocDefault) == hipSuccess) { frame = (si32*)ptr; frame_pinned = true; }
    else { (void)hipGetLastError(); frame = (si32*)malloc(frame_elems * sizeof(si32)); frame_pinned = false; }
    if (!frame) ojph_error(0x00030F01, "cannot allocate the %zu-sample frame buffer", frame_elems);
    lines.assign(p.num_comps, line_buf());
    for (ui32 c = 0; c < p.num_comps; ++c) {
      lines[c].size = cw[c]; lines[c].pre_size = 0; lines[c].flags = line_buf::LFT_32BIT | line_buf::LFT_INTEGER;
    }
  }
  si32* row(ui32 comp, ui32 line)
  {
    if (line >= ch[comp]) return spare.data();
    return slot ? stage[comp].data() : frame + coff[comp] + (size_t)line * cw[comp];
  }
  // a staged row <-> its place in the slot (pipe_bits-bit samples, planes as ojphgpu_plan_comp_info lays them out)
  void commit_row(ui32 comp, ui32 line)
  {
    if (!slot || line >= ch[comp]) return;
    const si32* sp = stage[comp].data();
    const size_t at = coff[comp] + (size_t)line * cw[comp];
    const ui32 n = cw[comp];
    // a sample outside the container's range saturates (it cannot be carried; the int32 slots of the default setting carry it)
    const bool sg = comps[comp].is_signed;
    const si32 lo = sg ? -(1 << (pipe_bits - 1)) : 0, hi = sg ? (1 << (pipe_bits - 1)) - 1 : (1 << pipe_bits) - 1;
    if (pipe_bits == 8) { ui8* dp = slot + at; for (ui32 x = 0; x < n; ++x) { const si32 v = sp[x] < lo ? lo : (sp[x] > hi ? hi : sp[x]); dp[x] = (ui8)v; } }
    else { ui16* dp = (ui16*)slot + at; for (ui32 x = 0; x < n; ++x) { const si32 v = sp[x] < lo ? lo : (sp[x] > hi ? hi : sp[x]); dp[x] = (ui16)v; } }
  }
  void fetch_row(ui32 comp, ui32 line)
  {
    if (!slot || line >= ch[comp]) return;
    si32* dp = stage[comp].data();
    const size_t at = coff[comp] + (size_t)line * cw[comp];
    const ui32 n = cw[comp];
    const bool sg = comps[comp].is_signed;
    if (pipe_bits == 8) { const ui8* sp = slot + at; if (sg) for (ui32 x = 0; x < n; ++x) dp[x] = (si8)sp[x]; else for (ui32 x = 0; x < n; ++x) dp[x] = sp[x]; }
    else { const ui16* sp = (const ui16*)slot + at; if (sg) for (ui32 x = 0; x < n; ++x) dp[x] = (si16)sp[x]; else for (ui32 x = 0; x < n; ++x) dp[x] = sp[x]; }
  }
  // The slots of a sequence's pipes hold int32 samples unless the application asked for narrow ones
  // (codestream::set_narrow_sample_containers): only int32 carries every value the reference's si32 line_buf does --
  // an out-of-range sample handed to exchange(), the 256 a lossy decode of an 8-bit component can come back with.
  bool narrow = false;
  int container_for_frame() const                        // the slot container of the frame's samples
  {
    if (!narrow) return 32;
    ui32 deepest = 0;
    for (ui32 c = 0; c < p.num_comps && c < comps.size(); ++c) deepest = comps[c].bit_depth > deepest ? comps[c].bit_depth : deepest;
    return deepest <= 8 ? 8 : deepest <= 16 ? 16 : 32;
  }
  // the oldest queued frame: its codestream is collected from the pipe and written to its file
  void write_oldest()
  {
    if (pending.empty() || !epipe) return;
    const Pending pd = pending.front();
    pending.erase(pending.begin());
    const ui8* cs = nullptr; size_t n = 0;
    const int rc = ojphgpu_enc_pipe_collect(epipe, &cs, &n);
    if (rc) ojph_error(0x00030F0B, "GPU encode failed (status %d)", rc);
    if (pd.file->write(cs, n) != n) ojph_error(0x00030071, "Error writing to file");
    if (pd.close_after) pd.file->close();
  }
  void drain() { while (!pending.empty()) write_oldest(); }
  // what param_siz::get_recon_width / _height report (ojph_params.cpp:330-346), also before the plan exists
  ui32 recon_w(ui32 c) const
  {
    const ui64 d = (ui64)(c < comps.size() && comps[c].ds.x ? comps[c].ds.x : 1) << skip_recon;
    const ui64 x1 = (ui64)image_offset.x + p.width;
    return (ui32)((x1 + d - 1) / d - ((ui64)image_offset.x + d - 1) / d);
  }
  ui32 recon_h(ui32 c) const
  {
    const ui64 d = (ui64)(c < comps.size() && comps[c].ds.y ? comps[c].ds.y : 1) << skip_recon;
    const ui64 y1 = (ui64)image_offset.y + p.height;
    return (ui32)((y1 + d - 1) / d - ((ui64)image_offset.y + d - 1) / d);
  }
};

}  // namespace local

using local::codestream_state;

// ---- param_siz -----------------------------------------------------------------------------------
// the state keeps the image SIZE in p.width / p.height and the offset next to it: the extent the
// reference's setter takes is their sum, whichever of the two setters is called first
void param_siz::set_image_extent(point extent)
{
  state->p.width = extent.x > state->image_offset.x ? extent.x - state->image_offset.x : 0;
  state->p.height = extent.y > state->image_offset.y ? extent.y - state->image_offset.y : 0;
}
void param_siz::set_tile_size(size s) { state->p.tile_w = s.w; state->p.tile_h = s.h; }
void param_siz::set_image_offset(point offset)
{
  const point extent = get_image_extent();
  state->image_offset = offset;
  set_image_extent(extent);
}
void param_siz::set_tile_offset(point offset) { state->tile_offset = offset; }
void param_siz::set_num_components(ui32 num_comps)
{
  state->p.num_comps = num_comps;
  state->comps.assign(num_comps, local::comp_info{ point(1, 1), 8, false, false });
}
void param_siz::set_component(ui32 comp_num, const point& downsampling, ui32 bit_depth, bool is_signed)
{
  if (comp_num >= state->comps.size())
    ojph_error(0x00050001, "component number %u is larger than the number of components", comp_num);   // ojph_params.cpp:110
  state->comps[comp_num] = local::comp_info{ downsampling, bit_depth, is_signed, true };
}
point param_siz::get_image_extent() const { return point(state->p.width + state->image_offset.x, state->p.height + state->image_offset.y); }
point param_siz::get_image_offset() const { return state->image_offset; }
size param_siz::get_tile_size() const
{
  const point e = get_image_extent();                       // not set: what write_headers would choose (ojph_codestream_local.cpp:562-570)
  return size(state->p.tile_w ? state->p.tile_w : e.x + state->image_offset.x, state->p.tile_h ? state->p.tile_h : e.y + state->image_offset.y);
}
point param_siz::get_tile_offset() const { return state->tile_offset; }
ui32 param_siz::get_num_components() const { return state->p.num_comps; }
ui32 param_siz::get_bit_depth(ui32 c) const { return c < state->comps.size() ? state->comps[c].bit_depth : state->p.bit_depth; }
bool param_siz::is_signed(ui32 c) const { return c < state->comps.size() ? state->comps[c].is_signed : state->p.is_signed != 0; }
point param_siz::get_downsampling(ui32 c) const { return c < state->comps.size() ? state->comps[c].ds : point(1, 1); }
ui32 param_siz::get_recon_width(ui32 c) const { return state->recon_w(c); }
ui32 param_siz::get_recon_height(ui32 c) const { return state->recon_h(c); }

// ---- param_cod / param_qcd -----------------------------------------------------------------------
void param_cod::set_num_decomposition(ui32 n)
{
  if (n > 32) ojph_error(0x00050002, "maximum number of decompositions cannot exceed 32");           // ojph_params.cpp:254
  state->p.num_decomps = n;
}
void param_cod::set_block_dims(ui32 width, ui32 height)
{
  const ui32 lw = log2_exact(width), lh = log2_exact(height);
  if (width == 0 || width != (1u << lw) || height == 0 || height != (1u << lh) || lw < 2 || lh < 2 || lw + lh > 12)
    ojph_error(0x00050011, "incorrect code block dimensions");                                       // ojph_params.cpp:265
  state->p.block_w = width; state->p.block_h = height;
}
void param_cod::set_precinct_size(int num_levels, size* precinct_size)
{
  memset(state->p.precinct_exps, 0, sizeof(state->p.precinct_exps));
  if (num_levels == 0 || precinct_size == nullptr) { state->p.precinct_w = state->p.precinct_h = 0; return; }
  for (ui32 i = 0; i < 33; ++i) {                          // entry i = resolution i (0 = lowest), the last one repeats (ojph_params.cpp:195-211)
    const size t = precinct_size[(int)i < num_levels ? i : num_levels - 1];
    if (t.w == 0 || t.h == 0) ojph_error(0x00050021, "precinct width or height cannot be 0");
    const ui32 px = log2_exact(t.w), py = log2_exact(t.h);
    if (t.w != (1u << px) || t.h != (1u << py)) ojph_error(0x00050022, "precinct width and height should be a power of 2");
    if (px > 15 || py > 15) ojph_error(0x00050023, "precinct size is too large");
    state->p.precinct_exps[i] = (ui8)(px | (py << 4));
  }
  state->p.precinct_w = precinct_size[0].w; state->p.precinct_h = precinct_size[0].h;
}
void param_cod::set_progression_order(const char* name)
{
  for (ui32 i = 0; i < 5; ++i)
    if (strncasecmp(name, PROG_NAMES[i], 4) == 0 && strlen(name) == 4) { state->p.prog_order = i; return; }
  ojph_error(0x00050031, "unknown progression order");                                                // ojph_params.cpp:318
}
void param_cod::set_color_transform(bool ct) { state->p.color_transform = ct; }
void param_cod::set_reversible(bool rev) { state->p.reversible = rev; }
ui32 param_cod::get_num_decompositions() const { return state->p.num_decomps; }
size param_cod::get_block_dims() const { return size(state->p.block_w, state->p.block_h); }
size param_cod::get_log_block_dims() const { return size(log2_exact(state->p.block_w), log2_exact(state->p.block_h)); }
bool param_cod::is_reversible() const { return state->p.reversible != 0; }
size param_cod::get_precinct_size(ui32 level_num) const
{
  if (level_num < 36 && state->p.precinct_exps[level_num])
    return size(1u << (state->p.precinct_exps[level_num] & 15), 1u << (state->p.precinct_exps[level_num] >> 4));
  return size(state->p.precinct_w ? state->p.precinct_w : 32768, state->p.precinct_h ? state->p.precinct_h : 32768);
}
size param_cod::get_log_precinct_size(ui32 l) const { size s = get_precinct_size(l); return size(log2_exact(s.w), log2_exact(s.h)); }
int param_cod::get_progression_order() const { return (int)state->p.prog_order; }
const char* param_cod::get_progression_order_as_string() const { return PROG_NAMES[state->p.prog_order % 5]; }
bool param_cod::is_using_color_transform() const { return state->p.color_transform != 0; }

// COC interface (ojph_params.cpp:255-282, :374-399; get_or_add_coc :1341-1348)
static ojphgpu_coc& coc_of(local::codestream_state* state, ui32 comp_idx)
{
  if (comp_idx >= OJPHGPU_MAX_COC_COMPS)
    ojph_error(0x00050091, "per-component coding styles (COC) are supported for components 0..%d", OJPHGPU_MAX_COC_COMPS - 1);
  ojphgpu_coc& k = state->p.coc[comp_idx];
  if (k.rank == 0) {
    ui32 made = 0;
    for (const ojphgpu_coc& o : state->p.coc) made = std::max<ui32>(made, o.rank);
    memset(&k, 0, sizeof(k));
    k.rank = (ui8)(made + 1); k.num_decomps = 5; k.log_block_w = 6; k.log_block_h = 6; k.reversible = 0;
  }
  return k;
}
static const ojphgpu_coc* coc_if(const local::codestream_state* state, ui32 comp_idx)
{
  return comp_idx < OJPHGPU_MAX_COC_COMPS && state->p.coc[comp_idx].rank ? &state->p.coc[comp_idx] : nullptr;
}
void param_cod::set_num_decomposition(ui32 comp_idx, ui32 n)
{
  ojphgpu_coc& k = coc_of(state, comp_idx);
  if (n > 32) ojph_error(0x00050001, "maximum number of decompositions cannot exceed 32");
  k.num_decomps = (ui8)n;
}
void param_cod::set_block_dims(ui32 comp_idx, ui32 width, ui32 height)
{
  ojphgpu_coc& k = coc_of(state, comp_idx);
  const ui32 lw = log2_exact(width), lh = log2_exact(height);
  if (width == 0 || width != (1u << lw) || height == 0 || height != (1u << lh) || lw < 2 || lh < 2 || lw + lh > 12)
    ojph_error(0x00050011, "incorrect code block dimensions");
  k.log_block_w = (ui8)lw; k.log_block_h = (ui8)lh;
}
void param_cod::set_precinct_size(ui32 comp_idx, int num_levels, size* precinct_size)
{
  ojphgpu_coc& k = coc_of(state, comp_idx);
  memset(k.precinct_exps, 0, sizeof(k.precinct_exps));
  if (num_levels == 0 || precinct_size == nullptr) { k.has_precincts = 0; return; }
  k.has_precincts = 1;
  for (ui32 i = 0; i <= k.num_decomps; ++i) {              // uses the decompositions set so far (ojph_params.cpp:195)
    const size t = precinct_size[(int)i < num_levels ? i : num_levels - 1];
    if (t.w == 0 || t.h == 0) ojph_error(0x00050021, "precinct width or height cannot be 0");
    const ui32 px = log2_exact(t.w), py = log2_exact(t.h);
    if (t.w != (1u << px) || t.h != (1u << py)) ojph_error(0x00050022, "precinct width and height should be a power of 2");
    if (px > 15 || py > 15) ojph_error(0x00050023, "precinct size is too large");
    if (i > 0 && (px == 0 || py == 0)) ojph_error(0x00050024, "precinct size is too small");
    k.precinct_exps[i] = (ui8)(px | (py << 4));
  }
}
void param_cod::set_reversible(ui32 comp_idx, bool reversible) { coc_of(state, comp_idx).reversible = reversible ? 1 : 0; }
// (ojph_params.cpp:368-370, :396-399) bit 3 of the code-block style byte of the COD / of the component's COC, kept by the
// parser in reserved[0] bit 0 (include/ojphgpu.h); a codestream this library writes never sets it
bool param_cod::get_block_vertical_causality() const { return (state->p.reserved[0] & 1u) != 0; }
bool param_cod::get_block_vertical_causality(ui32 c) const
{
  const ojphgpu_coc* k = coc_if(state, c);
  return k ? (k->reserved[0] & 1u) != 0 : get_block_vertical_causality();
}
ui32 param_cod::get_num_decompositions(ui32 c) const { const ojphgpu_coc* k = coc_if(state, c); return k ? k->num_decomps : get_num_decompositions(); }
size param_cod::get_log_block_dims(ui32 c) const { const ojphgpu_coc* k = coc_if(state, c); return k ? size(k->log_block_w, k->log_block_h) : get_log_block_dims(); }
size param_cod::get_block_dims(ui32 c) const { const size l = get_log_block_dims(c); return size(1u << l.w, 1u << l.h); }
bool param_cod::is_reversible(ui32 c) const { const ojphgpu_coc* k = coc_if(state, c); return k ? k->reversible != 0 : is_reversible(); }
size param_cod::get_log_precinct_size(ui32 c, ui32 level_num) const
{
  const ojphgpu_coc* k = coc_if(state, c);
  if (!k) return get_log_precinct_size(level_num);
  if (!k->has_precincts || level_num >= 36) return size(15, 15);
  return size(k->precinct_exps[level_num] & 15u, k->precinct_exps[level_num] >> 4);
}
size param_cod::get_precinct_size(ui32 c, ui32 level_num) const { const size l = get_log_precinct_size(c, level_num); return size(1u << l.w, 1u << l.h); }

// ---- param_nlt (ojph_params.cpp:441-458, :2176-2208) ---------------------------------------------
void param_nlt::set_nonlinear_transform(ui32 comp_num, ui8 nl_type)
{
  if (nl_type != OJPH_NLT_NO_NLT && nl_type != OJPH_NLT_BINARY_COMPLEMENT_NLT)
    ojph_error(0x00050171, "Nonliearities other than type 0 (No Nonlinearity) or type  3 (Binary Binary Complement to Sign Magnitude Conversion) are not supported yet");
  ojphgpu_params& p = state->p;
  if (comp_num == ALL_COMPS) { p.nlt_default = (ui8)(nl_type + 1); return; }
  if (comp_num >= OJPHGPU_MAX_COC_COMPS) ojph_error(0x00050172, "NLT entries are supported for components 0..%d on the GPU path", OJPHGPU_MAX_COC_COMPS - 1);
  if (p.nlt_comp[comp_num] == 0) {
    ui32 made = 0;
    for (ui8 r : p.nlt_rank) made = std::max<ui32>(made, r);
    p.nlt_rank[comp_num] = (ui8)(made + 1);
  }
  p.nlt_comp[comp_num] = (ui8)(nl_type + 1);
}
bool param_nlt::get_nonlinear_transform(ui32 comp_num, ui8& bit_depth, bool& is_signed, ui8& nl_type) const
{
  const ojphgpu_params& p = state->p;
  const bool own = comp_num < OJPHGPU_MAX_COC_COMPS && p.nlt_comp[comp_num] != 0;
  if (!own && p.nlt_default == 0) return false;
  const ui8 bd = own ? p.nlt_bd[comp_num] : p.nlt_bd_default;
  if (p.nlt_reserved[0]) { bit_depth = (ui8)((bd & 0x7F) + 1); is_signed = (bd & 0x80) != 0; }   // as read from the codestream
  else {                                                                                          // before write_headers fills BDnlt
    const ui32 c = comp_num < state->comps.size() ? comp_num : 0;
    bit_depth = (ui8)(c < state->comps.size() ? state->comps[c].bit_depth : p.bit_depth);
    is_signed = c < state->comps.size() ? state->comps[c].is_signed : p.is_signed != 0;
  }
  nl_type = (ui8)((own ? p.nlt_comp[comp_num] : p.nlt_default) - 1);
  return true;
}

void param_qcd::set_irrev_quant(float delta) { state->p.qstep = delta; }
void param_qcd::set_irrev_quant(ui32 comp_idx, float delta)
{
  if (comp_idx < OJPHGPU_MAX_COC_COMPS && state->p.qcc_qfactor[comp_idx]) return;   // that QCC's quality factor takes precedence (ojph_params.cpp:1456-1459)
  state->p.qstep = delta;
}
void param_qcd::set_qfactor(ui32 comp_idx, comp_type ctype, ui8 qfactor)
{
  if (qfactor < 1 || qfactor > 100) ojph_error(0x00050191, "Qfactor must be between 1 and 100, but was set to %i.", (int)qfactor);   // ojph_params.cpp:2025
  if (ctype > OJPH_COMP_CR) ojph_error(0x00050192, "the component type must be Y, Cb or Cr");
  if (comp_idx >= OJPHGPU_MAX_COC_COMPS) ojph_error(0x00050193, "per-component quality factors are supported for components 0..%d on the GPU path", OJPHGPU_MAX_COC_COMPS - 1);
  ojphgpu_params& p = state->p;
  if (p.qcc_qfactor[comp_idx] == 0) {
    ui32 made = 0;
    for (ui8 r : p.qcc_rank) made = std::max<ui32>(made, r);
    p.qcc_rank[comp_idx] = (ui8)(made + 1);
  }
  p.qcc_qfactor[comp_idx] = qfactor; p.qcc_ctype[comp_idx] = (ui8)ctype;
}
void param_qcd::set_qfactor(ui8 qfactor)
{
  if (qfactor < 1 || qfactor > 100) ojph_error(0x00050181, "Qfactor must be between 1 and 100, but was set to %i.", (int)qfactor);   // ojph_params.cpp:1487
  state->p.reserved[2] = qfactor;
}

void comment_exchange::set_string(const char* str) { data = str; len = (ui16)strlen(str); Rcom = 1; }
void comment_exchange::set_data(const char* d, ui16 l) { data = d; len = l; Rcom = 0; }

// ---- codestream ----------------------------------------------------------------------------------
// The HIP runtime takes ~0.1 s to come up in a fresh process; the first codestream object starts
// that in the background so it overlaps with the caller reading its input file and setting
// parameters (every HIP call made later simply waits for the initialisation to finish).
static void warm_up_gpu_runtime()
{
  static std::once_flag once;
  std::call_once(once, [] { std::thread([] { (void)hipFree(nullptr); }).detach(); });
}

codestream::codestream() : state(new codestream_state()) { warm_up_gpu_runtime(); }
codestream::~codestream() { state->release(); state->release_pipes(); delete state; }
// (ojph_codestream_local.cpp:78-110) the object forgets the frame it coded but keeps what a next frame of the
// same format can use again
void codestream::restart()
{
  state->release(); state->reset_params(); state->planar = -1; state->resilient = false; state->profile.clear();
  state->sequence = true;
}

void codestream::set_planar(bool planar) { state->planar = planar ? 1 : 0; }
bool codestream::is_planar() const { return state->planar == 1; }
void codestream::set_profile(const char* s)
{
  if (strcmp(s, "IMF") != 0 && strcmp(s, "BROADCAST") != 0) ojph_error(0x000300A1, "unknown or unsupported profile");   // ojph_codestream_local.cpp:1103
  state->profile = s;
}
void codestream::set_tilepart_divisions(bool at_resolutions, bool at_components)
{
  state->p.reserved[1] = (at_resolutions ? 1u : 0u) | (at_components ? 2u : 0u);
}
bool codestream::is_tilepart_division_at_resolutions() { return (state->p.reserved[1] & 1u) != 0; }
bool codestream::is_tilepart_division_at_components() { return (state->p.reserved[1] & 2u) != 0; }
void codestream::request_tlm_marker(bool needed) { state->p.tlm = needed; }
bool codestream::is_tlm_requested() { return state->p.tlm != 0; }
void codestream::set_device(int device) { state->device = device; state->devices.clear(); }
void codestream::set_devices(const int* devices, ui32 num_devices)
{
  state->devices.assign(devices, devices + (devices ? num_devices : 0));
  if (!state->devices.empty()) state->device = state->devices[0];
}
void codestream::set_narrow_sample_containers(bool narrow) { state->narrow = narrow; }
void codestream::enable_frame_pipelining(ui32 frames_in_flight)
{
  state->drain();
  state->pipelining = frames_in_flight < 2 ? 0u : frames_in_flight > 16 ? 16u : frames_in_flight;
}
void codestream::drain() { state->drain(); }
void codestream::enable_resilience() { state->resilient = true; }
param_siz codestream::access_siz() { return param_siz(state); }
param_cod codestream::access_cod() { return param_cod(state); }
param_qcd codestream::access_qcd() { return param_qcd(state); }
param_nlt codestream::access_nlt() { return param_nlt(state); }

// IMF / BROADCAST profile rules (ojph_codestream_local.cpp:293-553): the profile only constrains the
// parameters; it also asks for a TLM marker and one tile-part per component
static void check_profile(codestream_state& S)
{
  const ojphgpu_params& p = S.p;
  const bool imf = S.profile == "IMF";
  const ui32 eb = imf ? 0x000300C0 : 0x000300B0;                   // the reference's error code families
  const char* nm = imf ? "IMF" : "broadcast";
  const ui32 ex = p.width + S.image_offset.x, ey = p.height + S.image_offset.y;
  const ui32 L = p.num_decomps;
  if (imf && !(ex <= 8192 && ey <= 6224))
    ojph_error(p.reversible ? 0x000300C1 : 0x000300C2, "Image dimensions do not meet any of the %s IMF profiles", p.reversible ? "lossless" : "lossy");
  if (S.image_offset.x || S.image_offset.y) ojph_error(eb + (imf ? 3 : 1), "For %s profile, image offset (XOsiz, YOsiz) has to be 0.", nm);
  if (S.tile_offset.x || S.tile_offset.y) ojph_error(eb + (imf ? 4 : 2), "For %s profile, tile offset (XTOsiz, YTOsiz) has to be 0.", nm);
  if (p.num_comps > (imf ? 3u : 4u)) ojph_error(eb + (imf ? 5 : 3), "For %s profile, the number of components has to be less or equal to %d", nm, imf ? 3 : 4);
  bool ds1 = true, ds2 = true, bd_ok = true;
  for (ui32 c = 0; c < p.num_comps; ++c) {
    const local::comp_info& ci = S.comps[c];
    ds1 &= ci.ds.y == 1 && ci.ds.x == 1;
    ds2 &= ci.ds.y == 1 && ci.ds.x == ((c == 1 || c == 2) ? 2u : 1u);
    bd_ok &= ci.bit_depth >= 8 && ci.bit_depth <= (imf ? 16u : 12u) && !ci.is_signed;
  }
  if (!ds1 && !ds2)
    ojph_error(eb + (imf ? 6 : 4), "For %s profile, either no component downsampling is used, or the x-dimension of the 2nd and 3rd "
               "components is downsampled by 2.", nm);
  if (!bd_ok) ojph_error(eb + (imf ? 7 : 5), "For %s profile, compnent bit_depth has to be between 8 and %d bits inclusively, and the samples must be unsigned", nm, imf ? 16 : 12);
  auto lg = [](ui32 v) { ui32 k = 0; while ((1u << k) < v) ++k; return k; };
  const ui32 lbw = lg(p.block_w), lbh = lg(p.block_h);
  if (imf) { if (lbw != 5 || lbh != 5) ojph_error(0x000300C8, "For IMF profile, codeblock dimensions are restricted. Use \"-block_size {32,32}\" at the commandline"); }
  else {
    if (L == 0 || L > 5) ojph_error(0x000300B6, "For broadcast profile, number of decompositions has to be between1 and 5 inclusively.");
    if (lbw < 5 || lbw > 7) ojph_error(0x000300B7, "For broadcast profile, codeblock dimensions are restricted such that codeblock width has to be either 32, 64, or 128.");
    if (lbh < 5 || lbh > 7) ojph_error(0x000300B8, "For broadcast profile, codeblock dimensions are restricted such that codeblock height has to be either 32, 64, or 128.");
  }
  // precincts: {128,128} for the lowest resolution, {256,256} above -- the reference's loop keeps only
  // the verdict of the LAST resolution when there is more than one (:380-383, :518-522)
  auto pp = [&](ui32 r, ui32& w, ui32& h) {
    bool per_res = false;
    for (ui32 i = 0; i <= L && i < 36; ++i) per_res |= p.precinct_exps[i] != 0;
    if (per_res) { w = p.precinct_exps[r] & 15u; h = p.precinct_exps[r] >> 4; }
    else if (p.precinct_w && p.precinct_h) { w = lg(p.precinct_w); h = lg(p.precinct_h); }
    else { w = h = 15; }
  };
  ui32 w, h; pp(0, w, h);
  bool pz = w == 7 && h == 7;
  for (ui32 i = 1; i <= L; ++i) { pp(i, w, h); pz = w == 8 && h == 8; }
  if (!pz) ojph_error(imf ? 0x000300C9 : 0x000300B9, "For %s profile, precinct sizes are restricted. Use \"-precincts {128,128},{256,256}\" at the commandline", nm);
  if (p.prog_order != 4) ojph_error(imf ? 0x000300CA : 0x000300BA, "For %s profile, the CPRL progression order must be used. Use \"-prog_order CPRL\".", nm);
  const ui32 tw = p.tile_w ? p.tile_w : ex + S.image_offset.x, th = p.tile_h ? p.tile_h : ey + S.image_offset.y;
  const ui32 tiles = ((ex + tw - 1) / tw) * ((ey + th - 1) / th);
  if (imf) {
    const bool rev = p.reversible != 0;
    bool k2 = ex <= 2048 && ey <= 1556 && L <= 5, k4 = ex <= 4096 && ey <= 3112 && L <= 6, k8 = L <= 7;
    if (L == 0 || (!k2 && !k4 && !k8))
      ojph_error(0x000300CB, "Number of decompositions does not match the IMF profile dictated by wavelet reversibility and image dimensions.");
    if (tiles > 1) {
      if (!rev) ojph_error(0x000300CC, "Lossy IMF profile must have one tile.");
      k2 &= tw == 1024 && th == 1024 && ((tw >= 1024 && L <= 4) || (tw >= 2048 && L <= 5));
      k4 &= ((tw == 1024 && th == 1024) || (tw == 2048 && th == 2048)) && ((tw >= 1024 && L <= 4) || (tw >= 2048 && L <= 5) || (tw >= 4096 && L <= 6));
      k8 &= ((tw == 1024 && th == 1024) || (tw == 2048 && th == 2048) || (tw == 4096 && th == 4096)) &&
            ((tw >= 1024 && L <= 4) || (tw >= 2048 && L <= 5) || (tw >= 4096 && L <= 6) || (tw >= 8192 && L <= 7));
      if (!k2 && !k4 && !k8)
        ojph_error(0x000300CD, "Number of decompositions does not match the IMF profile dictated by wavelet reversibility and image dimensions and tiles.");
    }
  } else if (tiles != 1 && tiles != 4) ojph_error(0x000300BB, "The broadcast profile can only have 1 or 4 tiles");
  S.p.tlm = 1;                                                     // need_tlm = true; tile-parts at components only
  S.p.reserved[1] = 2;
}

// (ojph_codestream_local.cpp:556-712) validation of the parameter set; the marker segments themselves
// are written by flush() together with the tile-parts
void codestream::write_headers(outfile_base* file, const comment_exchange* comments, ui32 num_comments)
{
  codestream_state& S = *state;
  if (S.headers_written) ojph_error(0x00030F02, "write_headers called twice");
  ojphgpu_params& p = S.p;
  if (p.num_comps == 0 || p.width == 0 || p.height == 0) ojph_error(0x00040001, "image extent / components have not been set");
  if (S.tile_offset.x > S.image_offset.x || S.tile_offset.y > S.image_offset.y)
    ojph_error(0x00040002, "Tile offset has to be smaller than the image offset");                     // ojph_params_local.h:240
  p.image_x0 = S.image_offset.x; p.image_y0 = S.image_offset.y; p.tile_x0 = S.tile_offset.x; p.tile_y0 = S.tile_offset.y;
  memset(p.comp_dx, 0, sizeof(p.comp_dx)); memset(p.comp_dy, 0, sizeof(p.comp_dy));
  memset(p.comp_depth, 0, sizeof(p.comp_depth)); memset(p.comp_sign, 0, sizeof(p.comp_sign));
  for (ui32 c = 0; c < p.num_comps; ++c) {
    const local::comp_info& ci = S.comps[c];
    if (!ci.set) ojph_error(0x00040002, "component %u has not been configured", c);
    if (ci.ds.x == 0 || ci.ds.y == 0 || ci.ds.x > 255 || ci.ds.y > 255) ojph_error(0x00030F04, "component sub-sampling factors must be 1..255");
    if ((ci.ds.x != 1 || ci.ds.y != 1) && c >= OJPHGPU_MAX_SUBSAMPLED_COMPS)
      ojph_error(0x00030F04, "sub-sampling is available for the first %d components on the GPU path", OJPHGPU_MAX_SUBSAMPLED_COMPS);
    if (c < OJPHGPU_MAX_SUBSAMPLED_COMPS) { p.comp_dx[c] = (uint8_t)ci.ds.x; p.comp_dy[c] = (uint8_t)ci.ds.y; }
    if (ci.bit_depth != S.comps[0].bit_depth || ci.is_signed != S.comps[0].is_signed) {
      if (c >= OJPHGPU_MAX_SUBSAMPLED_COMPS)
        ojph_error(0x00030F05, "a bit depth / signedness of its own is available for the first %d components on the GPU path", OJPHGPU_MAX_SUBSAMPLED_COMPS);
      p.comp_depth[c] = (uint8_t)ci.bit_depth; p.comp_sign[c] = ci.is_signed ? 2 : 1;
    }
  }
  p.bit_depth = S.comps[0].bit_depth; p.is_signed = S.comps[0].is_signed;
  if (p.color_transform && p.num_comps < 3)
    ojph_error(0x00040013, "color transform can only be employed when the image has 3 or more color components");   // ojph_params.cpp:560
  if (S.planar == -1) S.planar = p.color_transform ? 1 : 0;        // not chosen: the reference's rule (ojph_codestream_local.cpp:622-623)
  if (S.planar == 1 && p.color_transform)
    ojph_error(0x00030021, "the planar interface option cannot be used when colour transform is employed");          // :630
  if (!S.profile.empty()) check_profile(S);
  int rc = ojphgpu_plan_create(&p, &S.plan);
  if (rc) ojph_error(0x00030F07, "parameters rejected by the GPU path (status %d)", rc);
  if (comments != nullptr && num_comments != 0) {                  // ojph_codestream_local.cpp:686-703
    std::vector<const uint8_t*> d(num_comments); std::vector<uint16_t> l(num_comments), r(num_comments);
    for (ui32 i = 0; i < num_comments; ++i) { d[i] = (const uint8_t*)comments[i].data; l[i] = comments[i].len; r[i] = comments[i].Rcom; }
    if (ojphgpu_plan_set_comments(S.plan, d.data(), l.data(), r.data(), num_comments) != OJPHGPU_OK)
      ojph_error(0x00030F06, "COM marker segments rejected");
  }
  std::vector<std::vector<ui8>> cmts;
  for (ui32 i = 0; comments != nullptr && i < num_comments; ++i) {
    std::vector<ui8> c((const ui8*)comments[i].data, (const ui8*)comments[i].data + comments[i].len);
    c.push_back((ui8)comments[i].Rcom); c.push_back((ui8)(comments[i].Rcom >> 8));
    cmts.push_back(c);
  }
  if (S.sequence) {                                                // a frame of a sequence: through the pipeline
    if (S.epipe && (memcmp(&S.epipe_params, &p, sizeof(p)) != 0 || S.epipe_comments != cmts)) {   // another frame format
      S.drain();
      ojphgpu_enc_pipe_destroy(S.epipe); S.epipe = nullptr;
      ojphgpu_plan_destroy(S.epipe_plan); S.epipe_plan = nullptr;
    }
    if (S.epipe && S.pipe_bits != S.container_for_frame()) { S.drain(); ojphgpu_enc_pipe_destroy(S.epipe); S.epipe = nullptr; ojphgpu_plan_destroy(S.epipe_plan); S.epipe_plan = nullptr; }
    if (!S.epipe) {
      S.drain();
      S.pipe_bits = S.container_for_frame();
      rc = ojphgpu_enc_pipe_create(S.plan, S.device, S.pipelining > 2 ? S.pipelining : 4, S.pipe_bits, 0, &S.epipe);
      if (rc) ojph_error(0x00030F08, "cannot create the GPU encoder (status %d): no GPU?", rc);
      S.epipe_plan = S.plan; S.epipe_params = p; S.epipe_comments = cmts;
    } else { ojphgpu_plan_destroy(S.plan); S.plan = S.epipe_plan; }
    void* slot = nullptr; size_t bytes = 0;
    rc = ojphgpu_enc_pipe_acquire(S.epipe, &slot, &bytes);
    while (rc == OJPHGPU_E_AGAIN && !S.pending.empty()) {   // every slot holds a queued frame: the oldest one is written out now
      S.write_oldest();
      rc = ojphgpu_enc_pipe_acquire(S.epipe, &slot, &bytes);
    }
    if (rc) ojph_error(0x00030F08, "the frame pipeline has no free slot (status %d)", rc);
    S.alloc_frame((si32*)slot);
  } else if (S.devices.size() > 1) {                               // a tiled frame over several GPUs
    rc = ojphgpu_multi_encoder_create(S.plan, S.devices.data(), (uint32_t)S.devices.size(), &S.menc);
    if (rc) ojph_error(0x00030F08, "cannot create the GPU encoders (status %d): fewer GPUs than asked for?", rc);
    S.alloc_frame();
  } else {
    rc = ojphgpu_encoder_create(S.plan, S.device, nullptr, &S.enc);
    if (rc) ojph_error(0x00030F08, "cannot create the GPU encoder (status %d): no GPU?", rc);
    S.alloc_frame();
  }
  // a frame still queued for this very file object (enable_frame_pipelining, and the application re-opened the object for
  // the next frame): its codestream is written before the object is used again
  for (const codestream_state::Pending& pd : S.pending) if (pd.file == file) { S.drain(); break; }
  S.outfile = file;
  S.headers_written = true;
  S.cur_comp = 0; S.cur_line = 0; S.exhausted = false;
}

// (ojph_codestream_local.cpp:1176-1224) same protocol: NULL in -> first line out; after the last line
// NULL comes back and next_component is 0
line_buf* codestream::exchange(line_buf* line, ui32& next_component)
{
  codestream_state& S = *state;
  if (!S.headers_written) ojph_error(0x00030F09, "exchange called before write_headers");
  if (line) {                                   // the samples are in place (the line points into the frame), or in the staging row
    if (S.exhausted) { next_component = 0; return nullptr; }
    S.commit_row(S.cur_comp, S.cur_line);
    if (S.planar) {                             // one component at a time, each with its own height (:1195-1207)
      if (++S.cur_line >= S.ch[S.cur_comp]) { S.cur_line = 0; if (++S.cur_comp >= S.p.num_comps) { S.exhausted = true; next_component = 0; return nullptr; } }
    } else {                                    // every component for every line of component 0 (:1208-1219)
      if (++S.cur_comp >= S.p.num_comps) { S.cur_comp = 0; if (++S.cur_line >= S.ch[0]) { S.exhausted = true; next_component = 0; return nullptr; } }
    }
  }
  next_component = S.cur_comp;
  line_buf* l = &S.lines[S.cur_comp];
  l->i32 = S.row(S.cur_comp, S.cur_line);
  return l;
}

// (ojph_codestream_local.cpp:1148-1165)
void codestream::flush()
{
  codestream_state& S = *state;
  if (!S.headers_written) ojph_error(0x00030F0A, "flush called before write_headers");
  if (S.epipe && S.frame_of_pipe) {            // the frame sits in the pipe's pinned slot already: upload, code, assemble, download
    int rc = ojphgpu_enc_pipe_submit(S.epipe);
    if (rc) ojph_error(0x00030F0B, "GPU encode failed (status %d)", rc);
    S.pending.push_back(codestream_state::Pending{ S.outfile, false });
    if (!S.pipelining) S.drain();                // the reference's contract: the codestream is in the file when flush() returns (:1163)
    return;
  }
  if (S.menc) {
    size_t cap = S.frame_elems * 5 + (1u << 20), len = 0;                // (more than 5 bytes per sample: the second round below)
    for (int round = 0; round < 2; ++round) {
      std::unique_ptr<ui8[]> out(new ui8[cap]);
      const int rc = ojphgpu_multi_encode(S.menc, S.frame, out.get(), cap, &len);
      if (rc == OJPHGPU_E_OVERFLOW && round == 0) { cap = len + 16; continue; }
      if (rc) ojph_error(0x00030F0B, "GPU encode failed (status %d)", rc);
      if (S.outfile->write(out.get(), len) != len) ojph_error(0x00030071, "Error writing to file");
      return;
    }
  }
  size_t len = 0;
  int rc = ojphgpu_encode(S.enc, S.frame, nullptr, 0, &len);            // runs the GPU path; reports the codestream size
  if (rc != OJPHGPU_E_OVERFLOW && rc != OJPHGPU_OK) ojph_error(0x00030F0B, "GPU encode failed (status %d)", rc);
  std::unique_ptr<ui8[]> out(new ui8[len + 16]);
  rc = ojphgpu_encoder_finish(S.enc, out.get(), len + 16, &len);         // host Tier-2 only (block bytes are already here)
  if (rc) ojph_error(0x00030F0B, "GPU encode failed (status %d)", rc);
  if (S.outfile->write(out.get(), len) != len) ojph_error(0x00030071, "Error writing to file");        // :1163
}

// (ojph_codestream_local.cpp:769-910 + read() :912-1146)
void codestream::read_headers(infile_base* file)
{
  codestream_state& S = *state;
  S.infile = file;
  S.stream.clear();
  ui8 tmp[65536];
  for (;;) { size_t n = file->read(tmp, sizeof(tmp)); S.stream.insert(S.stream.end(), tmp, tmp + n); if (n < sizeof(tmp)) break; }
  int rc = ojphgpu_t2_parse(S.stream.data(), S.stream.size(), S.resilient ? 1 : 0, &S.plan);
  if (rc == OJPHGPU_E_CODESTREAM) ojph_error(0x00030051, "error reading the codestream headers / tile-parts");
  if (rc) ojph_error(0x00030F0C, "codestream not supported by the GPU path (status %d)", rc);
  ojphgpu_plan_params(S.plan, &S.p);
  S.comps.assign(S.p.num_comps, local::comp_info{ point(1, 1), S.p.bit_depth, S.p.is_signed != 0, true });
  for (ui32 c = 0; c < S.p.num_comps && c < OJPHGPU_MAX_SUBSAMPLED_COMPS; ++c) {
    S.comps[c].ds = point(S.p.comp_dx[c] ? S.p.comp_dx[c] : 1, S.p.comp_dy[c] ? S.p.comp_dy[c] : 1);
    uint32_t bd = 0, sg = 0;
    ojphgpu_plan_comp_format(S.plan, c, &bd, &sg);
    S.comps[c].bit_depth = bd; S.comps[c].is_signed = sg != 0;
  }
  S.image_offset = point(S.p.image_x0, S.p.image_y0); S.tile_offset = point(S.p.tile_x0, S.p.tile_y0);
  if (S.planar == -1) S.planar = S.p.color_transform ? 0 : 1;                                         // :879
  S.headers_read = true;
}

// (ojph_codestream_local.cpp:883-900)
void codestream::restrict_input_resolution(ui32 skipped_res_for_data, ui32 skipped_res_for_recon)
{
  codestream_state& S = *state;
  if (!S.headers_read) ojph_error(0x00030F0D, "restrict_input_resolution called before read_headers");
  if (skipped_res_for_data < skipped_res_for_recon)
    ojph_error(0x000300A1, "skipped_resolution for data %d must be equal or smaller than skipped_resolution for reconstruction %d",
               skipped_res_for_data, skipped_res_for_recon);
  if (skipped_res_for_data > S.p.num_decomps)
    ojph_error(0x000300A2, "skipped_resolution for data %d must be smaller than the number of decomposition levels %d",
               skipped_res_for_data, S.p.num_decomps);
  if (ojphgpu_plan_restrict_resolution(S.plan, skipped_res_for_data, skipped_res_for_recon) != OJPHGPU_OK)
    ojph_error(0x00030F0D, "the GPU path rejected the resolution restriction");
  S.skip_recon = skipped_res_for_recon; S.skip_data = skipped_res_for_data; S.restricted = true;
}

void codestream::create()
{
  codestream_state& S = *state;
  if (!S.headers_read) ojph_error(0x00030F0E, "create called before read_headers");
  if (S.sequence && !S.restricted) {            // a frame of a sequence: through the pipeline (whole-resolution decoding)
    if (S.dpipe && (S.dpipe_resilient != S.resilient || S.pipe_bits != S.container_for_frame())) { ojphgpu_dec_pipe_destroy(S.dpipe); S.dpipe = nullptr; }
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (!S.dpipe) {
        S.pipe_bits = S.container_for_frame();
        int rc = ojphgpu_dec_pipe_create(S.stream.data(), S.stream.size(), S.resilient ? 1 : 0, S.device, 2, S.pipe_bits, 0, &S.dpipe);
        if (rc) ojph_error(0x00030F0F, "cannot create the GPU decoder (status %d): no GPU?", rc);
        S.dpipe_resilient = S.resilient;
      }
      ui8* slot = nullptr;
      int rc = ojphgpu_dec_pipe_acquire(S.dpipe, S.stream.size(), &slot);
      if (rc == OJPHGPU_OK) { memcpy(slot, S.stream.data(), S.stream.size()); rc = ojphgpu_dec_pipe_submit(S.dpipe); }
      const void* frame = nullptr; size_t bytes = 0; ui32 failed = 0;
      if (rc == OJPHGPU_OK) rc = ojphgpu_dec_pipe_collect(S.dpipe, &frame, &bytes, &failed);
      if (rc == OJPHGPU_E_INVALID && attempt == 0) { ojphgpu_dec_pipe_destroy(S.dpipe); S.dpipe = nullptr; continue; }   // another frame format: a new pipe
      if (rc == OJPHGPU_E_BLOCK) { if (!S.resilient) ojph_error(0x000300A1, "Error decoding a codeblock"); }
      else if (rc) ojph_error(0x00030F12, "GPU decode failed (status %d)", rc);
      S.alloc_frame((si32*)const_cast<void*>(frame));
      S.decoded = true;
      break;
    }
    S.cur_comp = 0; S.cur_line = 0; S.exhausted = false;
    return;
  }
  int rc;
  if (S.devices.size() > 1)
    rc = ojphgpu_multi_decoder_create(S.stream.data(), S.stream.size(), S.resilient ? 1 : 0, S.restricted ? S.skip_data : 0,
                                      S.restricted ? S.skip_recon : 0, S.devices.data(), (uint32_t)S.devices.size(), &S.mdec);
  else
    rc = ojphgpu_decoder_create(S.plan, S.device, nullptr, &S.dec);
  if (rc) ojph_error(0x00030F0F, "cannot create the GPU decoder (status %d): no GPU?", rc);
  S.alloc_frame();
  S.cur_comp = 0; S.cur_line = 0; S.exhausted = false; S.decoded = false;
}

// (ojph_codestream_local.cpp:1227-1273)
line_buf* codestream::pull(ui32& comp_num)
{
  codestream_state& S = *state;
  if (!S.dec && !S.mdec && !S.decoded) ojph_error(0x00030F10, "pull called before create");
  if (!S.decoded) {
    int rc = S.mdec ? ojphgpu_multi_decode(S.mdec, S.stream.data(), S.stream.size(), S.frame, nullptr)
                    : ojphgpu_decode(S.dec, S.stream.data(), S.stream.size(), S.frame);
    if (rc == OJPHGPU_E_BLOCK) { if (!S.resilient) ojph_error(0x000300A1, "Error decoding a codeblock"); }   // ojph_codeblock.cpp:214-224
    else if (rc) ojph_error(0x00030F12, "GPU decode failed (status %d)", rc);
    S.decoded = true;
  }
  if (S.exhausted) { comp_num = 0; return nullptr; }
  comp_num = S.cur_comp;
  line_buf* l = &S.lines[S.cur_comp];
  S.fetch_row(S.cur_comp, S.cur_line);
  l->i32 = S.row(S.cur_comp, S.cur_line);
  if (S.planar) {
    if (++S.cur_line >= S.ch[S.cur_comp]) { S.cur_line = 0; if (++S.cur_comp >= S.p.num_comps) S.exhausted = true; }
  } else {
    if (++S.cur_comp >= S.p.num_comps) { S.cur_comp = 0; if (++S.cur_line >= S.ch[0]) S.exhausted = true; }
  }
  return l;
}

void codestream::close()
{
  codestream_state& S = *state;
  if (S.infile) S.infile->close();
  if (S.outfile) {
    // a frame still queued for this file (enable_frame_pipelining): the application may open the same file object for
    // the next frame right after this call, so the queue is written out now -- the codestream is in the file that is closed
    for (const codestream_state::Pending& pd : S.pending) if (pd.file == S.outfile) { S.drain(); break; }
    S.outfile->close();
  }
  S.infile = nullptr; S.outfile = nullptr;
}

}  // namespace ojph
