#include "ojphgpu.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <dirent.h>
#include <string>
int main(int argc, char** argv) {
  int n = 0, ok = 0, rewritten = 0;
  for (int a = 1; a < argc; ++a) {
    DIR* d = opendir(argv[a]); if (!d) continue;
    while (dirent* e = readdir(d)) {
      std::string p = std::string(argv[a]) + "/" + e->d_name;
      FILE* f = fopen(p.c_str(), "rb"); if (!f) continue;
      std::vector<uint8_t> b; uint8_t tmp[65536]; size_t k;
      while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) b.insert(b.end(), tmp, tmp + k);
      fclose(f);
      if (b.size() < 4) continue;
      for (int res = 0; res < 4; ++res) {                 // 2, 3: restricted reading (one resolution dropped)
        ojphgpu_plan* plan = nullptr;
        int rc = res < 2 ? ojphgpu_t2_parse(b.data(), b.size(), res, &plan) : ojphgpu_t2_parse_restricted(b.data(), b.size(), res & 1, 1, 1, &plan);
        ++n;
        if (rc == 0 && plan) {
          ++ok;
          uint64_t c[8]; ojphgpu_plan_counts(plan, c);
          std::vector<ojphgpu_coded_block> cb(c[2]); ojphgpu_plan_coded_blocks(plan, cb.data(), cb.size());
          for (auto& k2 : cb) if (k2.len1 && k2.offset + k2.len1 + k2.len2 > b.size()) { printf("OUT OF RANGE block in %s\n", p.c_str()); break; }
          size_t cnt = 0; ojphgpu_plan_padded_blocks(plan, nullptr, 0, &cnt);
          // the writer on what was read (a transcoder's use): whatever it makes of a damaged plan, it stays inside its buffers
          size_t need = 0; std::vector<uint8_t> outb(b.size() + 65536);
          int wrc = ojphgpu_t2_write(plan, b.data(), cb.data(), outb.data(), outb.size(), &need);
          if (wrc == 0) ++rewritten;
          ojphgpu_plan_destroy(plan);
        }
      }
    }
    closedir(d);
  }
  printf("%d parses, %d plans, %d written again\n", n, ok, rewritten);
  return 0;
}
