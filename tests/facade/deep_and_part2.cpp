// tests/facade/deep_and_part2.cpp -- SURVEY.md section 8(f) N4 through the ojph::codestream-compatible facade (the N3
// boundary): what ojph_compress / ojph_expand do with the reference's classes (src/apps/ojph_expand/ojph_expand.cpp:389-425,
// src/apps/ojph_compress/ojph_compress.cpp), on streams only the deep-sample and Part-2 paths take.
//   deep_and_part2 write <out.j2c> <w> <h> <bit depth> <signed 0|1>
//       a one-component frame of that depth (more than 26 bits: the reference's 64-bit sample path) goes in through
//       exchange() -- whose lines must be si32 lines like the reference's (ojph_codestream_local.cpp:279) -- is written
//       reversibly, read back through read_headers / create / pull and compared sample for sample
//   deep_and_part2 read <in.j2c> <out.bin> [resilient]
//       any codestream (the Python side hands over Part-2 streams with DFS / ATK marker segments, deep ones, one with the
//       vertically causal block style): prints what the parameter getters report, writes the pulled planes as int32
// Exit code 0 and "all checks passed" = every check of the mode held.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/ojph_gpu_codestream.h"

namespace {
int failures = 0;
void expect(bool ok, const char* what) { if (!ok) { ++failures; fprintf(stderr, "FAILED: %s\n", what); } }

// the sample the frame holds at (x, y): uses the whole range of the bit depth, top bits included
ojph::si32 sample_of(unsigned x, unsigned y, unsigned depth, bool is_signed)
{
  const unsigned long long range = depth >= 32 ? 0xFFFFFFFFull : ((1ull << depth) - 1ull);
  unsigned long long v = ((unsigned long long)x * 2654435761ull + (unsigned long long)y * 40503ull * 65537ull + ((x ^ y) & 15u)) & range;
  if ((x + y) % 23u == 0u) v = range;                       // the largest value ...
  if ((x + 2 * y) % 29u == 0u) v = 0;                       // ... and the smallest
  long long s = (long long)v;
  if (is_signed) s -= 1ll << (depth - 1);
  return (ojph::si32)(ojph::ui32)(unsigned long long)s;     // the si32 container (an unsigned 32-bit sample keeps its bits)
}

int do_write(const char* path, unsigned w, unsigned h, unsigned depth, bool is_signed)
{
  {
    ojph::codestream cs;
    ojph::param_siz siz = cs.access_siz();
    siz.set_image_extent(ojph::point(w, h));
    siz.set_num_components(1);
    siz.set_component(0, ojph::point(1, 1), depth, is_signed);
    siz.set_image_offset(ojph::point(0, 0));
    siz.set_tile_size(ojph::size(0, 0));
    siz.set_tile_offset(ojph::point(0, 0));
    ojph::param_cod cod = cs.access_cod();
    cod.set_num_decomposition(4);
    cod.set_block_dims(64, 64);
    cod.set_color_transform(false);
    cod.set_reversible(true);
    cs.set_planar(true);
    ojph::j2c_outfile out;
    out.open(path);
    cs.write_headers(&out);
    ojph::ui32 next = 0;
    ojph::line_buf* line = cs.exchange(nullptr, next);
    for (unsigned y = 0; y < h; ++y) {
      expect(line != nullptr && next == 0, "exchange hands out component 0's lines");
      if (!line) return 1;
      expect((line->flags & ojph::line_buf::LFT_SIZE_MASK) == ojph::line_buf::LFT_32BIT && (line->flags & ojph::line_buf::LFT_INTEGER) != 0,
             "the application's lines are si32 at every bit depth (ojph_codestream_local.cpp:279)");
      expect(line->size >= w, "a line holds the component's width");
      for (unsigned x = 0; x < w; ++x) line->i32[x] = sample_of(x, y, depth, is_signed);
      line = cs.exchange(line, next);
    }
    expect(line == nullptr, "no line after the last one");
    cs.flush();
    cs.close();
  }
  {
    ojph::codestream cs;
    ojph::j2c_infile in;
    in.open(path);
    cs.read_headers(&in);
    ojph::param_siz siz = cs.access_siz();
    expect(siz.get_num_components() == 1 && siz.get_bit_depth(0) == depth && siz.is_signed(0) == is_signed, "SIZ says what was written");
    expect(siz.get_recon_width(0) == w && siz.get_recon_height(0) == h, "the frame's size");
    expect(cs.access_cod().is_reversible() && cs.access_cod().get_num_decompositions() == 4, "COD says what was written");
    expect(!cs.access_cod().get_block_vertical_causality(), "a stream of this library is not vertically causal");
    cs.set_planar(true);
    cs.create();
    unsigned long long wrong = 0;
    for (unsigned y = 0; y < h; ++y) {
      ojph::ui32 comp = 9;
      ojph::line_buf* line = cs.pull(comp);
      expect(line != nullptr && comp == 0, "pull hands out component 0's lines");
      if (!line) return 1;
      expect((line->flags & ojph::line_buf::LFT_SIZE_MASK) == ojph::line_buf::LFT_32BIT && (line->flags & ojph::line_buf::LFT_INTEGER) != 0, "pulled lines are si32 lines");
      for (unsigned x = 0; x < w; ++x) wrong += line->i32[x] != sample_of(x, y, depth, is_signed);
    }
    expect(wrong == 0, "every sample of the deep frame comes back");
    if (wrong) fprintf(stderr, "%llu samples differ\n", wrong);
    cs.close();
  }
  return failures;
}

int do_read(const char* path, const char* dump, bool resilient)
{
  ojph::codestream cs;
  if (resilient) cs.enable_resilience();
  ojph::j2c_infile in;
  in.open(path);
  cs.read_headers(&in);
  ojph::param_siz siz = cs.access_siz();
  ojph::param_cod cod = cs.access_cod();
  const unsigned nc = siz.get_num_components();
  printf("components %u causal %d\n", nc, cod.get_block_vertical_causality() ? 1 : 0);
  for (unsigned c = 0; c < nc; ++c)
    printf("comp %u: %u x %u depth %u signed %d decomps %u reversible %d causal %d\n", c, siz.get_recon_width(c), siz.get_recon_height(c),
           siz.get_bit_depth(c), siz.is_signed(c) ? 1 : 0, cod.get_num_decompositions(c), cod.is_reversible(c) ? 1 : 0,
           cod.get_block_vertical_causality(c) ? 1 : 0);
  cs.set_planar(true);
  cs.create();
  FILE* f = fopen(dump, "wb");
  if (!f) { fprintf(stderr, "cannot open %s\n", dump); return 1; }
  for (unsigned c = 0; c < nc; ++c)
    for (unsigned y = 0; y < siz.get_recon_height(c); ++y) {
      ojph::ui32 comp = 0;
      ojph::line_buf* line = cs.pull(comp);
      expect(line != nullptr && comp == c, "planar pull order");
      if (!line) { fclose(f); return 1; }
      fwrite(line->i32, sizeof(ojph::si32), siz.get_recon_width(c), f);
    }
  fclose(f);
  cs.close();
  return failures;
}
}  // namespace

int main(int argc, char** argv)
{
  try {
    int rc = 1;
    if (argc >= 7 && !strcmp(argv[1], "write")) rc = do_write(argv[2], (unsigned)atoi(argv[3]), (unsigned)atoi(argv[4]), (unsigned)atoi(argv[5]), atoi(argv[6]) != 0);
    else if (argc >= 4 && !strcmp(argv[1], "read")) rc = do_read(argv[2], argv[3], argc >= 5 && atoi(argv[4]) != 0);
    else { fprintf(stderr, "usage: deep_and_part2 write out.j2c w h depth signed | read in.j2c out.bin [resilient]\n"); return 2; }
    if (rc == 0) printf("all checks passed\n");
    return rc ? 1 : 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
}
