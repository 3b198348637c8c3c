// tests/facade/multi_device.cpp -- ONE tiled frame coded by several GPUs through the ojph::codestream-compatible facade
// (codestream::set_devices; include/ojphgpu.h section 8).  The test box has one GPU, so the same device is listed
// twice / three times: the tile runs, worker threads and the placement of the tile-parts are the code a multi-GPU
// node runs.  Checks: the codestream equals the single-device one byte for byte (tiles are independent:
// ojph_codestream_local.cpp:113-180, ojph_tile.cpp:584-610); decoding with several devices gives the frame back.
// Exit code 0 = all checks passed.
#include <cstdio>
#include <stdexcept>
#include <vector>
#include "../../include/ojph_gpu_codestream.h"

namespace {
int failures = 0;
void expect(bool ok, const char* what) { if (!ok) { ++failures; fprintf(stderr, "FAILED: %s\n", what); } }

const unsigned W = 700, H = 500, NC = 3, DEPTH = 10;
int sample_of(unsigned c, unsigned x, unsigned y) { return (int)(((x * 3 + y * (5 + c) + ((x * y) >> 4)) >> 1) % (1u << DEPTH)); }

std::vector<ojph::ui8> encode(const std::vector<int>& devices)
{
  ojph::codestream cs;
  if (!devices.empty()) cs.set_devices(devices.data(), (ojph::ui32)devices.size());
  ojph::param_siz siz = cs.access_siz();
  siz.set_image_extent(ojph::point(W, H));
  siz.set_num_components(NC);
  for (unsigned c = 0; c < NC; ++c) siz.set_component(c, ojph::point(1, 1), DEPTH, false);
  siz.set_tile_size(ojph::size(256, 128));
  ojph::param_cod cod = cs.access_cod();
  cod.set_num_decomposition(4);
  cod.set_reversible(true);
  cod.set_color_transform(true);
  cs.set_planar(false);
  cs.request_tlm_marker(true);
  ojph::mem_outfile out;
  out.open();
  cs.write_headers(&out);
  ojph::ui32 next = 0;
  ojph::line_buf* line = cs.exchange(nullptr, next);
  for (unsigned y = 0; y < H; ++y)
    for (unsigned c = 0; c < NC; ++c) {
      for (unsigned x = 0; x < W; ++x) line->i32[x] = sample_of(c, x, y);
      line = cs.exchange(line, next);
    }
  cs.flush();
  std::vector<ojph::ui8> bytes(out.get_data(), out.get_data() + out.get_used_size());
  cs.close();
  return bytes;
}

void decode_and_check(const std::vector<ojph::ui8>& data, const std::vector<int>& devices)
{
  ojph::codestream cs;
  if (!devices.empty()) cs.set_devices(devices.data(), (ojph::ui32)devices.size());
  ojph::mem_infile in;
  in.open(data.data(), data.size());
  cs.read_headers(&in);
  cs.set_planar(true);
  cs.create();
  for (unsigned c = 0; c < NC; ++c)
    for (unsigned y = 0; y < H; ++y) {
      ojph::ui32 got = 0;
      ojph::line_buf* line = cs.pull(got);
      if (!line || got != c) { expect(false, "pull order"); return; }
      for (unsigned x = 0; x < W; ++x) if (line->i32[x] != sample_of(c, x, y)) { expect(false, "lossless frame"); return; }
    }
  cs.close();
}
}  // namespace

int main()
{
  try {
    const std::vector<ojph::ui8> one = encode({});
    expect(one.size() > 1000, "single-device encode");
    expect(encode({ 0, 0 }) == one, "two workers write the single-device codestream");
    expect(encode({ 0, 0, 0 }) == one, "three workers write the single-device codestream");
    decode_and_check(one, {});
    decode_and_check(one, { 0, 0 });
    decode_and_check(one, { 0, 0, 0 });
  } catch (const std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 2;
  }
  if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
  printf("multi_device: all checks passed\n");
  return 0;
}
