// tests/host/pool_check.cpp -- the host thread pool's contract (openjph_amd/csrc/ojph_pool.h): every item runs exactly
// once, several callers may be inside parallel_for at the same time, and a body that throws does not take the
// process down or leave workers calling through a dead function object: the first exception comes back to the caller
// after every item has run.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>
#include "../../openjph_amd/csrc/ojph_pool.h"

int main()
{
  using namespace ojphgpu;
  std::vector<std::atomic<int>> hit(10000);
  parallel_for(hit.size(), [&](size_t i) { hit[i]++; });
  for (auto& h : hit) if (h.load() != 1) { printf("FAIL: an item ran %d times\n", h.load()); return 1; }
  int caught = 0;
  std::atomic<int> ran{ 0 };
  for (int round = 0; round < 50; ++round) {
    try {
      parallel_for(64, [&](size_t i) { ran++; if (i % 7 == 3) throw std::runtime_error("body failed"); });
    } catch (const std::runtime_error&) { caught++; }
  }
  if (caught != 50 || ran.load() != 50 * 64) { printf("FAIL: caught %d, ran %d\n", caught, ran.load()); return 1; }
  // two callers at once, one of them throwing
  std::atomic<int> a{ 0 }, b{ 0 }; int bad = 0;
  std::thread t1([&] { for (int r = 0; r < 100; ++r) parallel_for(32, [&](size_t) { a++; }); });
  std::thread t2([&] { for (int r = 0; r < 100; ++r) { try { parallel_for(32, [&](size_t i) { b++; if (i == 5) throw 1; }); } catch (int) { bad++; } } });
  t1.join(); t2.join();
  if (a.load() != 3200 || b.load() != 3200 || bad != 100) { printf("FAIL: %d %d %d\n", a.load(), b.load(), bad); return 1; }
  printf("OK threads=%u\n", pool_threads());
  return 0;
}
