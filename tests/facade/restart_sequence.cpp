// tests/facade/restart_sequence.cpp -- one ojph::codestream object coding a SEQUENCE of frames through
// restart() (ojph_codestream.h:204; the reference re-uses its allocations across frames the same way,
// ojph_codestream_local.cpp:78-110).  From the second frame on the facade works through the frame pipelines
// (pinned slots, persistent device objects).  Checks: every frame's codestream equals the one a FRESH object
// produces for the same frame; decoding the sequence with one restarted object gives every frame back; a frame
// of another format in the middle of the sequence (the pipe must be rebuilt) and a reduced-resolution decode in
// the middle (takes the one-shot path) are handled.  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/ojph_gpu_codestream.h"

namespace {

int failures = 0;
void expect(bool ok, const char* what) { if (!ok) { ++failures; fprintf(stderr, "FAILED: %s\n", what); } }

struct Format { unsigned w, h, comps, depth; bool reversible, color; };

int sample_of(const Format& f, unsigned frame, unsigned c, unsigned x, unsigned y)
{
  return (int)(((x * (3 + frame) + y * (5 + c) + frame * 31 + ((x * y) >> 4)) >> 1) % (1u << f.depth));
}

void configure(ojph::codestream& cs, const Format& f)
{
  ojph::param_siz siz = cs.access_siz();
  siz.set_image_extent(ojph::point(f.w, f.h));
  siz.set_num_components(f.comps);
  for (unsigned c = 0; c < f.comps; ++c) siz.set_component(c, ojph::point(1, 1), f.depth, false);
  ojph::param_cod cod = cs.access_cod();
  cod.set_num_decomposition(4);
  cod.set_block_dims(64, 64);
  cod.set_color_transform(f.color);
  cod.set_reversible(f.reversible);
  if (!f.reversible) cs.access_qcd().set_irrev_quant(0.005f);
  cs.set_planar(!f.color);
}

void encode_frame(ojph::codestream& cs, const Format& f, unsigned frame, ojph::mem_outfile& out)
{
  configure(cs, f);
  out.open();
  cs.write_headers(&out);
  ojph::ui32 next = 0;
  ojph::line_buf* line = cs.exchange(nullptr, next);
  if (f.color) {
    for (unsigned y = 0; y < f.h; ++y)
      for (unsigned c = 0; c < f.comps; ++c) {
        expect(next == c && line != nullptr, "exchange order (interleaved)");
        for (unsigned x = 0; x < f.w; ++x) line->i32[x] = sample_of(f, frame, c, x, y);
        line = cs.exchange(line, next);
      }
  } else {
    for (unsigned c = 0; c < f.comps; ++c)
      for (unsigned y = 0; y < f.h; ++y) {
        expect(next == c && line != nullptr, "exchange order (planar)");
        for (unsigned x = 0; x < f.w; ++x) line->i32[x] = sample_of(f, frame, c, x, y);
        line = cs.exchange(line, next);
      }
  }
  cs.flush();
  cs.close();
}

std::vector<ojph::ui8> bytes_of(const ojph::mem_outfile& o) { return std::vector<ojph::ui8>(o.get_data(), o.get_data() + o.get_used_size()); }

void decode_and_check(ojph::codestream& cs, const Format& f, unsigned frame, const std::vector<ojph::ui8>& data, unsigned skip)
{
  ojph::mem_infile in;
  in.open(data.data(), data.size());
  cs.read_headers(&in);
  if (skip) cs.restrict_input_resolution(skip, skip);
  cs.set_planar(true);
  cs.create();
  ojph::param_siz siz = cs.access_siz();
  for (unsigned c = 0; c < f.comps; ++c) {
    const unsigned cw = siz.get_recon_width(c), ch = siz.get_recon_height(c);
    expect(cw == ((f.w + (1u << skip) - 1) >> skip) && ch == ((f.h + (1u << skip) - 1) >> skip), "reconstructed size");
    for (unsigned y = 0; y < ch; ++y) {
      ojph::ui32 got = 0;
      ojph::line_buf* line = cs.pull(got);
      expect(got == c && line != nullptr, "pull order");
      if (!line) return;
      if (skip) continue;
      for (unsigned x = 0; x < cw; ++x) {
        const int want = sample_of(f, frame, c, x, y), have = line->i32[x];
        if (f.reversible ? have != want : (have < want - 12 || have > want + 12)) { expect(false, f.reversible ? "lossless frame" : "lossy frame within the quantisation error"); return; }
      }
    }
  }
  cs.close();
}

}  // namespace

int main()
{
  try {
    const Format A{ 300, 200, 3, 8, true, true }, B{ 257, 131, 1, 12, false, false };
    const Format seq[] = { A, A, A, B, B, A, A };                          // the format changes twice
    const unsigned n = sizeof(seq) / sizeof(seq[0]);
    std::vector<std::vector<ojph::ui8>> fresh(n), restarted(n);
    for (unsigned i = 0; i < n; ++i) {                                     // a fresh object per frame: the one-shot path
      ojph::codestream cs; ojph::mem_outfile out;
      encode_frame(cs, seq[i], i, out);
      fresh[i] = bytes_of(out);
    }
    {
      ojph::codestream cs;                                                 // ONE object for the whole sequence
      for (unsigned i = 0; i < n; ++i) {
        ojph::mem_outfile out;
        encode_frame(cs, seq[i], i, out);
        restarted[i] = bytes_of(out);
        cs.restart();
      }
    }
    for (unsigned i = 0; i < n; ++i) expect(fresh[i] == restarted[i], "a restarted object writes the codestream a fresh one writes");
    {
      // frame pipelining (GPU-path extension): flush() queues, the codestreams reach their files later, in order
      ojph::codestream cs;
      cs.enable_frame_pipelining(3);
      std::vector<ojph::mem_outfile> outs(n);                               // the files of queued frames stay alive
      for (unsigned i = 0; i < n; ++i) {
        encode_frame(cs, seq[i], i, outs[i]);                              // ends with flush() + close()
        cs.restart();
      }
      cs.drain();
      for (unsigned i = 0; i < n; ++i) expect(fresh[i] == bytes_of(outs[i]), "a pipelined sequence writes the codestreams a fresh object writes");
    }
    {
      // the reference's idiom with ONE outfile object for every frame: open, write_headers, ..., flush, close, open again.
      // A queued frame must be in the file that close() closes, not in the next one
      ojph::codestream cs;
      cs.enable_frame_pipelining(3);
      ojph::mem_outfile out;
      for (unsigned i = 0; i < n; ++i) {
        encode_frame(cs, seq[i], i, out);                                  // opens `out`, ends with flush() + close()
        expect(fresh[i] == bytes_of(out), "one outfile object reused from frame to frame holds every frame's own codestream");
        cs.restart();
      }
    }
    {
      // narrow slots (opt-in): the same codestreams; and with the default int32 slots a sample outside the bit depth's range
      // is coded as it was given (the reference's si32 line_buf carries it), which a fresh object's one-shot path does too
      ojph::codestream cs;
      cs.set_narrow_sample_containers(true);
      for (unsigned i = 0; i < n; ++i) {
        ojph::mem_outfile out;
        encode_frame(cs, seq[i], i, out);
        expect(fresh[i] == bytes_of(out), "narrow slots: the codestream a fresh object writes");
        cs.restart();
      }
    }
    {
      const Format C{ 96, 64, 1, 8, true, false };
      auto encode_out_of_range = [&](ojph::codestream& cs, ojph::mem_outfile& out) {
        configure(cs, C);
        out.open();
        cs.write_headers(&out);
        ojph::ui32 next = 0;
        ojph::line_buf* line = cs.exchange(nullptr, next);
        for (unsigned y = 0; y < C.h; ++y) {
          for (unsigned x = 0; x < C.w; ++x) line->i32[x] = (int)((x * 5 + y * 3) % 300) - 20;    // beyond [0, 255] on both sides
          line = cs.exchange(line, next);
        }
        cs.flush(); cs.close();
      };
      ojph::codestream one; ojph::mem_outfile ref_out;
      encode_out_of_range(one, ref_out);
      ojph::codestream cs;
      for (int k = 0; k < 3; ++k) {
        ojph::mem_outfile out;
        encode_out_of_range(cs, out);
        expect(bytes_of(ref_out) == bytes_of(out), "out-of-range samples are coded as given, also from the second frame of a sequence on");
        cs.restart();
      }
    }
    {
      ojph::codestream cs;
      for (unsigned i = 0; i < n; ++i) {
        decode_and_check(cs, seq[i], i, fresh[i], i == 2 ? 1u : 0u);        // frame 2 at half resolution
        cs.restart();
      }
    }
    {
      ojph::codestream cs;
      cs.set_narrow_sample_containers(true);
      for (unsigned i = 0; i < n; ++i) {
        decode_and_check(cs, seq[i], i, fresh[i], 0u);
        cs.restart();
      }
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 2;
  }
  if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
  printf("restart_sequence: all checks passed\n");
  return 0;
}
