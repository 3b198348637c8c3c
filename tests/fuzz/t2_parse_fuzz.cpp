// tests/fuzz/t2_parse_fuzz.cpp -- mutation fuzzing of the codestream parser under ASan / UBSan.
//
// Built by tests/test_parser_fuzz.py from ojph_plan.cpp + ojph_t2.cpp (host-only sources, no HIP) with
// g++ -fsanitize=address,undefined.  Every input lives in a heap buffer of EXACTLY its own length, so a
// read one byte past the codestream aborts the process.  The reference's counterpart is its fuzz
// target (/root/reference/fuzzing/ojph_expand_fuzz_target.cpp): whatever the bytes, read_headers()
// either succeeds or reports an error.
//
//   t2_parse_fuzz <iterations> <seed file> [<seed file> ...]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ojphgpu.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }
static uint32_t below(uint32_t n) { return n ? rnd() % n : 0; }

static unsigned long parsed = 0, refused = 0;

static void run(const std::vector<uint8_t>& v)
{
  uint8_t* exact = (uint8_t*)malloc(v.size() ? v.size() : 1);     // exact size: ASan guards both ends
  if (!v.empty()) memcpy(exact, v.data(), v.size());
  for (int resilient = 0; resilient < 2; ++resilient) {
    ojphgpu_plan* plan = nullptr;
    int rc = ojphgpu_t2_parse(exact, v.size(), resilient, &plan);
    if (rc == OJPHGPU_OK && plan) {
      ++parsed;
      uint64_t counts[8];
      ojphgpu_plan_counts(plan, counts);
      std::vector<ojphgpu_coded_block> cb((size_t)counts[2]);
      if (!cb.empty() && ojphgpu_plan_coded_blocks(plan, cb.data(), cb.size()) == OJPHGPU_OK)
        for (const ojphgpu_coded_block& k : cb)
          if ((uint64_t)k.len1 + k.len2 && k.offset + k.len1 + k.len2 > v.size()) { fprintf(stderr, "block outside the buffer\n"); abort(); }
      ojphgpu_plan_destroy(plan);
    } else {
      ++refused;
      if (plan) { fprintf(stderr, "plan returned with an error status\n"); abort(); }
    }
  }
  free(exact);
}

int main(int argc, char** argv)
{
  if (argc < 3) return 2;
  const long iters = atol(argv[1]);
  std::vector<std::vector<uint8_t>> seeds;
  for (int i = 2; i < argc; ++i) {
    FILE* f = fopen(argv[i], "rb");
    if (!f) return 2;
    std::vector<uint8_t> v; uint8_t buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    seeds.push_back(v);
  }
  // crafted cases: marker segments shorter than their fixed fields (QCD Lqcd = 2 / 3, COD Lcod = 2, SIZ Lsiz
  // = 2 / 40), segment lengths pointing past the file, a lone SOC
  static const uint8_t crafted[][12] = {
    { 0xFF, 0x4F, 0xFF, 0x5C, 0x00, 0x02, 0x00, 0x00 }, { 0xFF, 0x4F, 0xFF, 0x5C, 0x00, 0x03, 0x00, 0x00 },
    { 0xFF, 0x4F, 0xFF, 0x52, 0x00, 0x02, 0x00, 0x00 }, { 0xFF, 0x4F, 0xFF, 0x51, 0x00, 0x02, 0x00, 0x00 },
    { 0xFF, 0x4F, 0xFF, 0x51, 0x00, 0x28, 0x40, 0x00 }, { 0xFF, 0x4F, 0xFF, 0x5D, 0x00, 0x03, 0x00, 0x00 },
    { 0xFF, 0x4F, 0xFF, 0x53, 0x00, 0x09, 0x00, 0x00 }, { 0xFF, 0x4F, 0xFF, 0x76, 0x00, 0x06, 0x00, 0x00 },
    { 0xFF, 0x4F, 0xFF, 0x5C, 0xFF, 0xFF, 0x00, 0x00 }, { 0xFF, 0x4F },
  };
  for (const auto& c : crafted)
    for (size_t n = 2; n <= 8; ++n) run(std::vector<uint8_t>(c, c + n));
  // every seed with one marker segment length rewritten to each small value
  for (const std::vector<uint8_t>& s : seeds) {
    run(s);
    size_t pos = 2;
    while (pos + 4 <= s.size() && !(s[pos] == 0xFF && s[pos + 1] == 0x90)) {
      const size_t L = ((size_t)s[pos + 2] << 8) | s[pos + 3];
      for (uint32_t newL = 0; newL < 48; ++newL) {
        std::vector<uint8_t> v = s; v[pos + 2] = 0; v[pos + 3] = (uint8_t)newL;
        run(v);
        v.resize(pos + 2 + (newL < 2 ? 2 : newL));          // ... and the file ending right behind the shortened segment
        run(v);
      }
      pos += 2 + L;
    }
  }
  for (long it = 0; it < iters; ++it) {
    std::vector<uint8_t> v = seeds[below((uint32_t)seeds.size())];
    size_t hdr_end = v.size();
    for (size_t i = 0; i + 1 < v.size(); ++i) if (v[i] == 0xFF && v[i + 1] == 0x90) { hdr_end = i; break; }
    const uint32_t nmut = 1 + below(4);
    for (uint32_t k = 0; k < nmut && v.size() >= 4; ++k) {
      size_t pos = below(10) < 7 ? 2 + below((uint32_t)hdr_end + 38) : below((uint32_t)v.size());
      if (pos >= v.size()) pos = v.size() - 1;
      const uint32_t mode = below(100);
      if (mode < 45) v[pos] ^= (uint8_t)(1u << below(8));
      else if (mode < 65) v[pos] = (uint8_t)below(256);
      else if (mode < 75) v.erase(v.begin() + (long)pos, v.begin() + (long)std::min(v.size(), pos + 1 + below(8)));
      else if (mode < 85) { uint32_t n = 1 + below(8); for (uint32_t j = 0; j < n; ++j) v.insert(v.begin() + (long)pos, (uint8_t)below(256)); }
      else if (mode < 92) { v[pos] = 0; if (pos + 1 < v.size()) v[pos + 1] = (uint8_t)below(4); }    // tiny lengths / exponents
      else v.resize(pos < 4 ? 4 : pos);
    }
    run(v);
  }
  printf("parsed %lu refused %lu\n", parsed, refused);
  return 0;
}
