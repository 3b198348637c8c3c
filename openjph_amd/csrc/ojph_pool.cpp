// openjph_amd/csrc/ojph_pool.cpp -- see ojph_pool.h
#include "ojph_pool.h"

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace ojphgpu {

namespace {

struct Batch {
  size_t n;
  const std::function<void(size_t)>* fn;
  std::atomic<size_t> next{ 0 }, done{ 0 };
  std::mutex mu; std::condition_variable cv;
  std::exception_ptr error;                              // the first exception a body threw (under mu)
};

class Pool {
public:
  explicit Pool(unsigned workers) {
    for (unsigned i = 0; i < workers; ++i) threads_.emplace_back([this] { work(); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  unsigned workers() const { return (unsigned)threads_.size(); }
  void run(size_t n, const std::function<void(size_t)>& fn) {
    auto b = std::make_shared<Batch>();
    b->n = n; b->fn = &fn;
    { std::lock_guard<std::mutex> lk(mu_); queue_.push_back(b); }
    cv_.notify_all();
    drain(*b);                                           // the caller works on its own batch
    std::unique_lock<std::mutex> lk(b->mu);
    b->cv.wait(lk, [&] { return b->done.load() == n; });   // every item is counted, also one whose body threw: `fn` outlives all calls
    if (b->error) std::rethrow_exception(b->error);        // on the calling thread, after the last body has returned
  }

private:
  static void drain(Batch& b) {
    for (;;) {
      const size_t i = b.next.fetch_add(1);
      if (i >= b.n) return;
      try { (*b.fn)(i); }
      catch (...) { std::lock_guard<std::mutex> lk(b.mu); if (!b.error) b.error = std::current_exception(); }
      if (b.done.fetch_add(1) + 1 == b.n) { std::lock_guard<std::mutex> lk(b.mu); b.cv.notify_all(); }
    }
  }
  void work() {
    for (;;) {
      std::shared_ptr<Batch> b;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] {
          while (!queue_.empty() && queue_.front()->next.load() >= queue_.front()->n) queue_.pop_front();   // handed out completely
          return stop_ || !queue_.empty();
        });
        if (stop_) return;
        b = queue_.front();
      }
      drain(*b);
    }
  }
  std::vector<std::thread> threads_;
  std::deque<std::shared_ptr<Batch>> queue_;
  std::mutex mu_; std::condition_variable cv_;
  bool stop_ = false;
};

unsigned configured_threads()
{
  unsigned n = std::thread::hardware_concurrency();
  n = n ? (n < 8 ? n : 8) : 1;
  if (const char* e = getenv("OJPHGPU_T2_THREADS")) { const long v = atol(e); if (v >= 1 && v <= 256) n = (unsigned)v; }
  return n;
}

Pool* the_pool()
{
  // created on first use and never destroyed: worker threads must not be joined from a static
  // destructor that may run while another thread is still inside the library
  static Pool* p = new Pool(configured_threads() - 1);
  return p;
}

}  // namespace

unsigned pool_threads() { return the_pool()->workers() + 1; }

void parallel_for(size_t n, const std::function<void(size_t)>& fn)
{
  if (n == 0) return;
  Pool* p = the_pool();
  if (n == 1 || p->workers() == 0) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  p->run(n, fn);
}

}  // namespace ojphgpu
