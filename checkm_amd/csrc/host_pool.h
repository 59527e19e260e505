// host_pool.h -- the small thread pool every worker owns (plain C++: tests/native/stress_pool.cpp runs it under ThreadSanitizer).
#pragma once
#include <atomic>
#include <algorithm>
#include <condition_variable>
#include <cstddef>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ckm {

// A few host threads for the per-pair / per-sequence glue between the kernel stages (logs of rescale factors, region
// scans over the decoding terms, segment clustering, bit scores): the device idles while that glue runs.
class HostPool {
 public:
  explicit HostPool(int nthreads) {
    for (int i = 1; i < nthreads; ++i) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  // f(lo, hi) over [0, n) in chunks; the caller works too; returns when every chunk is done and no thread is still inside
  void run(size_t n, size_t chunk, const std::function<void(size_t, size_t)> &f) {
    if (!n) return;
    if (th_.empty() || n <= chunk) { f(0, n); return; }
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &f; n_ = n; chunk_ = chunk; next_.store(0); err_ = nullptr; open_ = true; ++active_;
    }
    cv_.notify_all();
    work(f, n, chunk);
    std::unique_lock<std::mutex> g(m_);
    open_ = false;                                    // late wakers must not join a job whose chunks are all handed out
    --active_;
    done_.wait(g, [this] { return active_ == 0; });   // every helper has left work(): the fields may change again
    job_ = nullptr;
    if (err_) std::rethrow_exception(err_);
  }
 private:
  // the job's description travels by value: a helper never reads fields the next run() may be rewriting
  void work(const std::function<void(size_t, size_t)> &f, size_t n, size_t chunk) {
    for (;;) {
      const size_t lo = next_.fetch_add(chunk);
      if (lo >= n) return;
      try { f(lo, std::min(n, lo + chunk)); } catch (...) { std::lock_guard<std::mutex> g(m_); if (!err_) err_ = std::current_exception(); }
    }
  }
  void loop() {
    for (;;) {
      const std::function<void(size_t, size_t)> *f; size_t n, chunk;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return stop_ || (open_ && job_ && next_.load() < n_); });
        if (stop_) return;
        f = job_; n = n_; chunk = chunk_; ++active_;
      }
      work(*f, n, chunk);
      std::lock_guard<std::mutex> g(m_);
      if (--active_ == 0) done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_; std::condition_variable cv_, done_;
  const std::function<void(size_t, size_t)> *job_ = nullptr;
  size_t n_ = 0, chunk_ = 1; std::atomic<size_t> next_{0};
  int active_ = 0; bool open_ = false, stop_ = false; std::exception_ptr err_;
};

}  // namespace ckm
