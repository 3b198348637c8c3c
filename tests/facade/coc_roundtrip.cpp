// tests/facade/coc_roundtrip.cpp -- drives the ojph::codestream-compatible facade through the scenario
// of the reference's tests/test_mixed_coc.cpp (a 4-component 64x64 image coded with the 9/7, its last
// component switched to the 5/3 through param_cod::set_reversible(comp_idx, true)), plus a second
// pass with a component of fewer decompositions and its own block size.  Checks what that test checks:
// encoding succeeds, read_headers reports the per-component settings, the reversible component comes
// back sample for sample.  Writes the codestream to argv[1] so that the Python side can compare its
// bytes with the oracle-built ones.  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/ojph_gpu_codestream.h"

namespace {

int failures = 0;
void expect(bool ok, const char* what) { if (!ok) { ++failures; fprintf(stderr, "FAILED: %s\n", what); } }

struct Scenario {
  unsigned w, h, comps, depth;
  unsigned coc_comp;          // the component that gets a COC
  bool coc_reversible; int coc_decomps, coc_block;   // < 0: setter not called
};

int sample_of(const Scenario& s, unsigned c, unsigned x, unsigned y)
{
  if (c == s.coc_comp) return (int)((x + y * s.w) % (1u << s.depth));   // a ramp: any mismatch after the round trip shows
  return (int)(((x * 3 + y * 5 + c * 17) >> 1) % (1u << s.depth));
}

void run(const Scenario& s, const std::string& path)
{
  {
    ojph::codestream cs;
    ojph::param_siz siz = cs.access_siz();
    siz.set_image_extent(ojph::point(s.w, s.h));
    siz.set_num_components(s.comps);
    for (unsigned c = 0; c < s.comps; ++c) siz.set_component(c, ojph::point(1, 1), s.depth, false);
    siz.set_image_offset(ojph::point(0, 0));
    siz.set_tile_size(ojph::size(s.w, s.h));
    siz.set_tile_offset(ojph::point(0, 0));
    ojph::param_cod cod = cs.access_cod();
    cod.set_num_decomposition(5);
    cod.set_block_dims(64, 64);
    cod.set_color_transform(false);
    cod.set_reversible(false);
    if (s.coc_decomps >= 0) cod.set_num_decomposition(s.coc_comp, (unsigned)s.coc_decomps);
    if (s.coc_block > 0) cod.set_block_dims(s.coc_comp, (unsigned)s.coc_block, (unsigned)s.coc_block);
    cod.set_reversible(s.coc_comp, s.coc_reversible);
    cs.access_qcd().set_irrev_quant(0.01f);
    cs.set_planar(true);
    ojph::j2c_outfile out;
    out.open(path.c_str());
    cs.write_headers(&out);
    ojph::ui32 next = 0;
    ojph::line_buf* line = cs.exchange(nullptr, next);
    for (unsigned c = 0; c < s.comps; ++c)
      for (unsigned y = 0; y < s.h; ++y) {
        expect(next == c && line != nullptr, "exchange asks for the components in planar order");
        for (unsigned x = 0; x < s.w; ++x) line->i32[x] = sample_of(s, c, x, y);
        line = cs.exchange(line, next);
      }
    cs.flush();
    cs.close();
  }
  {
    ojph::codestream cs;
    ojph::j2c_infile in;
    in.open(path.c_str());
    cs.read_headers(&in);
    ojph::param_siz siz = cs.access_siz();
    expect(siz.get_num_components() == s.comps, "component count survives");
    for (unsigned c = 0; c < s.comps; ++c) expect(siz.get_bit_depth(c) == s.depth, "bit depth survives");
    ojph::param_cod cod = cs.access_cod();
    expect(!cod.is_reversible(), "the COD stays irreversible");
    for (unsigned c = 0; c < s.comps; ++c) {
      const bool own = c == s.coc_comp;
      expect(cod.is_reversible(c) == (own ? s.coc_reversible : false), "per-component wavelet as set");
      expect(cod.get_num_decompositions(c) == (own && s.coc_decomps >= 0 ? (unsigned)s.coc_decomps : 5u), "per-component decompositions as set");
      expect(cod.get_block_dims(c).w == (own && s.coc_block > 0 ? (unsigned)s.coc_block : 64u), "per-component block size as set");
    }
    cs.restrict_input_resolution(0, 0);
    cs.set_planar(true);
    cs.create();
    for (unsigned c = 0; c < s.comps; ++c) {
      const unsigned cw = siz.get_recon_width(c), ch = siz.get_recon_height(c);
      expect(cw == s.w && ch == s.h, "reconstructed size");
      for (unsigned y = 0; y < ch; ++y) {
        ojph::ui32 got = 0;
        ojph::line_buf* line = cs.pull(got);
        expect(got == c && line != nullptr, "pull delivers the components in planar order");
        if (!line) return;
        for (unsigned x = 0; x < cw; ++x) {
          const int want = sample_of(s, c, x, y), have = line->i32[x];
          if (c == s.coc_comp && s.coc_reversible) { if (have != want) { expect(false, "reversible component is exact"); return; } }
          else if (have < want - 8 || have > want + 8) { expect(false, "irreversible component within the quantisation error"); return; }
        }
      }
    }
    cs.close();
  }
}

}  // namespace

int main(int argc, char** argv)
{
  const std::string base = argc > 1 ? argv[1] : "coc_roundtrip";
  try {
    run(Scenario{ 64, 64, 4, 8, 3, true, -1, -1 }, base + "_mixed.j2c");        // the reference's test
    run(Scenario{ 150, 100, 3, 10, 1, true, 2, 32 }, base + "_short.j2c");      // fewer decompositions, 32x32 blocks
    run(Scenario{ 150, 100, 3, 8, 0, false, 0, -1 }, base + "_flat.j2c");       // a component without any decomposition
  } catch (const std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 2;
  }
  if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
  printf("coc_roundtrip: all checks passed\n");
  return 0;
}
