// openjph_amd/apps/ojph_expand.cpp -- command-line decoder on the GPU path, option-compatible with
// the reference's ojph_expand (src/apps/ojph_expand/ojph_expand.cpp:75-438): -i, -o, -skip_res
// ({n} or {for_data,for_recon}), -resilient.  Output: .pgm (1 component), .ppm (3 components), .yuv /
// .raw (planar).  Prints "Elapsed time = ..." (:204).
#include <chrono>
#include "ojph_app_common.h"
#include "../../include/ojph_gpu_codestream.h"

int main(int argc, char** argv) {
  Args a(argc, argv);
  const char* in = a.get("-i"); const char* out = a.get("-o");
  if (!in || !out) { printf("ojph_expand (GPU path) -i in.j2c -o out.{pgm,ppm,yuv,raw} [-skip_res n] [-resilient true] [-device n | -devices n,n,...]\n"); return -1; }
  try {
    const auto t0 = std::chrono::steady_clock::now();
    const bool verbose = getenv("OJPH_APP_TIMING") != nullptr;       // phase times on stderr
    auto lap = [&](const char* what) {
      if (verbose) fprintf(stderr, "ojph_expand: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    ojph::codestream cs;
    if (a.get("-device")) cs.set_device(atoi(a.get("-device")));
    if (a.get("-devices")) {
      std::vector<int> devs;
      for (long v : Args::numbers(a.get("-devices"))) devs.push_back((int)v);
      if (!devs.empty()) cs.set_devices(devs.data(), (ojph::ui32)devs.size());
    }
    if (Args::to_bool(a.get("-resilient"))) cs.enable_resilience();
    ojph::j2c_infile file;
    file.open(in);
    cs.read_headers(&file);
    lap("file read + headers parsed");
    auto sk = Args::numbers(a.get("-skip_res"));
    if (!sk.empty()) cs.restrict_input_resolution((ojph::ui32)sk[0], (ojph::ui32)(sk.size() > 1 ? sk[1] : sk[0]));
    ojph::param_siz siz = cs.access_siz();
    Image img;
    img.num_comps = siz.get_num_components();
    img.width = siz.get_recon_width(0); img.height = siz.get_recon_height(0);
    img.bit_depth = siz.get_bit_depth(0); img.is_signed = siz.is_signed(0);
    std::vector<unsigned> cw, ch; size_t total_lines = 0;
    for (unsigned c = 0; c < img.num_comps; ++c) {
      cw.push_back(siz.get_recon_width(c)); ch.push_back(siz.get_recon_height(c)); total_lines += ch.back();
      img.depth.push_back(siz.get_bit_depth(c)); img.sgn.push_back(siz.is_signed(c));
    }
    img.layout_sizes(cw, ch);
    const std::string outs(out);
    const bool pnm = ends_with(outs, ".pgm") || ends_with(outs, ".ppm");
    if (ends_with(outs, ".pgm") && img.num_comps != 1) throw std::runtime_error("a .pgm output needs a 1-component codestream");
    if (ends_with(outs, ".ppm") && img.num_comps != 3) throw std::runtime_error("a .ppm output needs a 3-component codestream");
    if (!pnm && !ends_with(outs, ".yuv") && !ends_with(outs, ".raw")) throw std::runtime_error("unknown output file extension (pgm, ppm, yuv, raw)");
    cs.set_planar(!pnm || img.num_comps == 1);
    cs.create();
    lap("decoder created");
    std::vector<unsigned> row(img.num_comps, 0);
    ojph::ui32 comp = 0;
    for (size_t i = 0; i < total_lines; ++i) {
      ojph::line_buf* line = cs.pull(comp);
      if (i == 0) lap("first line (frame decoded)");
      if (!line) break;
      if (row[comp] < img.ch[comp])
        memcpy(img.plane(comp) + (size_t)row[comp] * img.cw[comp], line->i32, img.cw[comp] * sizeof(int));
      row[comp]++;
    }
    cs.close();
    lap("all lines pulled");
    if (pnm) write_pnm(out, img); else write_raw(out, img);
    lap("output file written");
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("Elapsed time = %f\n", dt);
  } catch (const std::exception& e) {
    const char* w = e.what();
    if (w && strncmp(w, "ojph error", 10) != 0) printf("%s\n", w);
    return -1;
  }
  return 0;
}
