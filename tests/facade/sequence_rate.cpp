// tests/facade/sequence_rate.cpp -- what an application gets from ONE ojph::codestream object coding a sequence of 8K
// 4:4:4 12-bit frames through restart() (BASELINE config 3 as a sequence): frames per second and Gsamples/s with the
// reference's flush() contract (the codestream is in the file when flush() returns) and with the GPU path's
// enable_frame_pipelining() (flush() queues; upload, kernels, Tier-2 and download overlap the application's next frame).
// The application side is a single thread that copies prepared int32 rows into the lines exchange() hands out -- what a
// file reader does; its share of the time is printed separately (it is the same for the reference library).
// usage: facade_sequence_rate [frames] [width height]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/ojph_gpu_codestream.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
  const unsigned frames = argc > 1 ? (unsigned)atoi(argv[1]) : 16;
  const unsigned w = argc > 3 ? (unsigned)atoi(argv[2]) : 7680, h = argc > 3 ? (unsigned)atoi(argv[3]) : 4320, nc = 3, depth = 12;
  std::vector<std::vector<ojph::si32>> rows(64, std::vector<ojph::si32>(w));      // 64 different prepared rows
  unsigned seed = 12345;
  for (auto& r : rows) for (unsigned x = 0; x < w; ++x) { seed = seed * 1664525u + 1013904223u; r[x] = (int)(2048 + 900 * ((x / 197) & 1) + ((seed >> 20) & 127)) & 4095; }
  for (int mode = 0; mode < 2; ++mode) {
    try {
      ojph::codestream cs;
      if (mode) cs.enable_frame_pipelining(4);
      std::vector<ojph::mem_outfile> outs(frames + 2);
      double t_fill = 0, t_flush = 0, t0 = 0;
      size_t bytes = 0;
      for (unsigned f = 0; f < frames + 2; ++f) {
        if (f == 2) { cs.drain(); t0 = now_s(); t_fill = t_flush = 0; }             // two frames of warm-up (runtime start, pipe creation)
        ojph::param_siz siz = cs.access_siz();
        siz.set_image_extent(ojph::point(w, h)); siz.set_num_components(nc);
        for (unsigned c = 0; c < nc; ++c) siz.set_component(c, ojph::point(1, 1), depth, false);
        ojph::param_cod cod = cs.access_cod();
        cod.set_num_decomposition(5); cod.set_block_dims(64, 64); cod.set_color_transform(false); cod.set_reversible(false);
        cs.access_qcd().set_irrev_quant(0.001f);
        cs.set_planar(true);
        outs[f].open();
        const double a = now_s();
        cs.write_headers(&outs[f]);
        ojph::ui32 next = 0;
        ojph::line_buf* line = cs.exchange(nullptr, next);
        for (unsigned c = 0; c < nc; ++c)
          for (unsigned y = 0; y < h; ++y) {
            memcpy(line->i32, rows[(y + 7 * c + f) & 63].data(), (size_t)w * 4);
            line = cs.exchange(line, next);
          }
        const double b = now_s();
        cs.flush();
        cs.close();
        const double c2 = now_s();
        t_fill += b - a; t_flush += c2 - b;
        cs.restart();
      }
      cs.drain();
      const double wall = now_s() - t0;
      for (unsigned f = 2; f < frames + 2; ++f) bytes += outs[f].get_used_size();
      const double samples = (double)w * h * nc * frames;
      printf("%-28s %u frames %ux%ux%u: %.2f ms per frame = %.2f Gsamples/s | application rows %.2f ms, flush()+close() %.2f ms per frame "
             "(= %.2f Gsamples/s behind the application) | %.3f bytes per sample\n",
             mode ? "enable_frame_pipelining(4)" : "flush() writes (reference)", frames, w, h, nc, wall * 1e3 / frames, samples / wall / 1e9,
             t_fill * 1e3 / frames, t_flush * 1e3 / frames, samples / (wall - t_fill > 1e-9 ? wall - t_fill : 1e-9) / 1e9, (double)bytes / samples);
    } catch (const std::exception& e) { fprintf(stderr, "exception: %s\n", e.what()); return 2; }
  }
  return 0;
}
