// openjph_amd/apps/ojph_app_common.h -- small helpers shared by ojph_compress / ojph_expand:
// command-line parsing in the style of the reference's CLI ("-name value", lists as {a,b},{c,d})
// and PGM / PPM / raw-planar (.yuv / .raw) image files.  Own code; the option NAMES follow
// src/apps/ojph_compress/ojph_compress.cpp:380-438 and src/apps/ojph_expand/ojph_expand.cpp.
#ifndef OJPH_APP_COMMON_H
#define OJPH_APP_COMMON_H
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

struct Args {
  std::vector<std::string> v;
  Args(int argc, char** argv) { for (int i = 1; i < argc; ++i) v.push_back(argv[i]); }
  const char* get(const char* name) const {
    for (size_t i = 0; i + 1 < v.size(); ++i) if (v[i] == name) return v[i + 1].c_str();
    return nullptr;
  }
  bool has(const char* name) const { for (auto& s : v) if (s == name) return true; return false; }
  static bool to_bool(const char* s) { return s && (!strcmp(s, "true") || !strcmp(s, "1")); }
  // "{a,b}" or "{a,b},{c,d}" or "a,b" -> flat list of numbers
  static std::vector<long> numbers(const char* s) {
    std::vector<long> out;
    if (!s) return out;
    const char* p = s;
    while (*p) {
      if ((*p >= '0' && *p <= '9') || *p == '-') { char* e; out.push_back(strtol(p, &e, 10)); p = e; }
      else ++p;
    }
    return out;
  }
  static std::vector<bool> bools(const char* s) {
    std::vector<bool> out;
    if (!s) return out;
    std::string t(s); size_t pos = 0;
    while (pos < t.size()) {
      size_t e = t.find(',', pos); if (e == std::string::npos) e = t.size();
      std::string w = t.substr(pos, e - pos);
      out.push_back(w == "true" || w == "1");
      pos = e + 1;
    }
    return out;
  }
};

inline bool ends_with(const std::string& s, const char* suf) {
  size_t n = strlen(suf);
  if (s.size() < n) return false;
  for (size_t i = 0; i < n; ++i) if (tolower(s[s.size() - n + i]) != suf[i]) return false;
  return true;
}

struct Image {                       // planar int32 samples; sub-sampled components have their own size
  unsigned width = 0, height = 0, num_comps = 0, bit_depth = 8; bool is_signed = false;
  std::vector<unsigned> depth; std::vector<bool> sgn;   // per component where given (else bit_depth / is_signed)
  unsigned bd(unsigned c) const { return c < depth.size() ? depth[c] : bit_depth; }
  bool sg(unsigned c) const { return c < sgn.size() ? (bool)sgn[c] : is_signed; }
  std::vector<unsigned> cw, ch;      // per component (empty until layout() is called)
  std::vector<size_t> off;
  std::vector<int> data;
  // component c is ceil(width / dx[c]) x ceil(height / dy[c]) unless explicit sizes are given
  void layout(const std::vector<unsigned>& dx = {}, const std::vector<unsigned>& dy = {}) {
    cw.assign(num_comps, width); ch.assign(num_comps, height); off.assign(num_comps, 0);
    size_t total = 0;
    for (unsigned c = 0; c < num_comps; ++c) {
      if (c < dx.size() && dx[c] > 1) cw[c] = (width + dx[c] - 1) / dx[c];
      if (c < dy.size() && dy[c] > 1) ch[c] = (height + dy[c] - 1) / dy[c];
      off[c] = total; total += (size_t)cw[c] * ch[c];
    }
    data.resize(total);
  }
  void layout_sizes(const std::vector<unsigned>& w, const std::vector<unsigned>& h) {
    cw = w; ch = h; off.assign(num_comps, 0);
    size_t total = 0;
    for (unsigned c = 0; c < num_comps; ++c) { off[c] = total; total += (size_t)cw[c] * ch[c]; }
    data.resize(total);
  }
  size_t samples() const { return data.size(); }
  int* plane(unsigned c) { return data.data() + off[c]; }
};

inline int pnm_token(FILE* f) {
  int c = fgetc(f);
  for (;;) {
    while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = fgetc(f);
    if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
    break;
  }
  int v = 0;
  while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = fgetc(f); }
  return v;                              // consumes exactly one whitespace character after the number
}

inline void read_pnm(const char* name, Image& img) {
  FILE* f = fopen(name, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + name);
  char m0 = (char)fgetc(f), m1 = (char)fgetc(f);
  if (m0 != 'P' || (m1 != '5' && m1 != '6')) { fclose(f); throw std::runtime_error("only binary PGM (P5) / PPM (P6) are supported"); }
  img.num_comps = m1 == '6' ? 3 : 1;
  img.width = (unsigned)pnm_token(f); img.height = (unsigned)pnm_token(f);
  int maxv = pnm_token(f);
  img.bit_depth = 1; while ((1 << img.bit_depth) <= maxv) ++img.bit_depth;
  img.is_signed = false;
  const size_t n = (size_t)img.width * img.height, bps = maxv > 255 ? 2 : 1;
  std::vector<unsigned char> raw(n * img.num_comps * bps);
  if (fread(raw.data(), 1, raw.size(), f) != raw.size()) { fclose(f); throw std::runtime_error("short PNM file"); }
  fclose(f);
  img.layout();
  for (size_t i = 0; i < n; ++i)
    for (unsigned c = 0; c < img.num_comps; ++c) {
      const unsigned char* p = raw.data() + (i * img.num_comps + c) * bps;
      img.plane(c)[i] = bps == 2 ? (p[0] << 8) | p[1] : p[0];            // PNM is big-endian
    }
}

inline void write_pnm(const char* name, Image& img) {
  if (img.num_comps != 1 && img.num_comps != 3) throw std::runtime_error("PGM / PPM need 1 or 3 components");
  for (unsigned c = 0; c < img.num_comps; ++c)
    if (img.cw[c] != img.width || img.ch[c] != img.height) throw std::runtime_error("PGM / PPM cannot hold sub-sampled components; write a .yuv");
  FILE* f = fopen(name, "wb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + name);
  const int maxv = (1 << img.bit_depth) - 1; const size_t bps = maxv > 255 ? 2 : 1;
  fprintf(f, "P%c\n%u %u\n%d\n", img.num_comps == 3 ? '6' : '5', img.width, img.height, maxv);
  const size_t n = (size_t)img.width * img.height;
  std::vector<unsigned char> raw(n * img.num_comps * bps);
  for (size_t i = 0; i < n; ++i)
    for (unsigned c = 0; c < img.num_comps; ++c) {
      int v = img.plane(c)[i]; v = v < 0 ? 0 : (v > maxv ? maxv : v);
      unsigned char* p = raw.data() + (i * img.num_comps + c) * bps;
      if (bps == 2) { p[0] = (unsigned char)(v >> 8); p[1] = (unsigned char)v; } else p[0] = (unsigned char)v;
    }
  fwrite(raw.data(), 1, raw.size(), f);
  fclose(f);
}

// planar raw (.yuv / .raw), little-endian, (bit depth + 7) / 8 bytes per sample: 1 and 2 as the reference's yuv reader /
// writer, 3 and 4 as its .raw reader / writer for deep samples (raw_in::read, ojph_img_io.cpp:1540-1617; raw_out::write,
// :1679-1810 -- signed samples extend from the container's top bit on the way in; on the way out 3- and 4-byte samples are
// limited to lower <= v <= upper with upper = 2^(depth-1) for signed, 2^depth for unsigned samples, the reference's bounds)
inline size_t raw_bytes_per_sample(unsigned bit_depth) { return bit_depth > 24 ? 4 : bit_depth > 16 ? 3 : bit_depth > 8 ? 2 : 1; }

inline void read_raw(const char* name, Image& img) {
  FILE* f = fopen(name, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + name);
  // img.layout() was called: planes follow each other, each with its own sample size
  for (unsigned c = 0; c < img.num_comps; ++c) {
    const size_t n = (size_t)img.cw[c] * img.ch[c], bps = raw_bytes_per_sample(img.bd(c));
    std::vector<unsigned char> raw(n * bps);
    if (fread(raw.data(), 1, raw.size(), f) != raw.size()) { fclose(f); throw std::runtime_error("short raw file"); }
    int* dst = img.plane(c);
    for (size_t i = 0; i < n; ++i) {
      const unsigned char* q = raw.data() + i * bps;
      unsigned u = q[0];
      for (size_t k = 1; k < bps; ++k) u |= (unsigned)q[k] << (8 * k);
      int v = (int)u;
      if (img.sg(c)) {
        const int sh = bps > 2 ? 32 - 8 * (int)bps : 32 - (int)img.bd(c);   // (1- and 2-byte samples: from the bit depth, as before)
        v = (int)(u << sh) >> sh;
      }
      dst[i] = v;
    }
  }
  fclose(f);
}

inline void write_raw(const char* name, Image& img) {
  FILE* f = fopen(name, "wb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + name);
  for (unsigned c = 0; c < img.num_comps; ++c) {
    const size_t n = (size_t)img.cw[c] * img.ch[c], bps = raw_bytes_per_sample(img.bd(c));
    std::vector<unsigned char> raw(n * bps);
    const int* src = img.plane(c);
    const long long upper = img.sg(c) ? 1ll << (img.bd(c) - 1) : 1ll << img.bd(c), lower = img.sg(c) ? -(1ll << (img.bd(c) - 1)) : 0ll;
    for (size_t i = 0; i < n; ++i) {
      long long v = img.sg(c) ? (long long)src[i] : (long long)(unsigned)src[i];
      if (bps > 2) { v = v < upper ? v : upper; v = v >= lower ? v : lower; }
      else v = src[i];
      for (size_t k = 0; k < bps; ++k) raw[i * bps + k] = (unsigned char)(v >> (8 * k));
    }
    fwrite(raw.data(), 1, raw.size(), f);
  }
  fclose(f);
}
#endif
