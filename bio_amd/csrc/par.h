// par.h -- one pool of host worker threads for the short parallel loops of the library and the
// driver (record scans, packing, copies into pinned memory).  A run hands out 16 MiB of
// inflated BAM at a time, so these loops are a millisecond each and there are thousands of
// them: starting threads per loop (as std::thread in a parallel_for does) costs more than the
// loops.  Workers are started on first use and live until the process ends.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace bp_par {

class Pool {
 public:
  static Pool& Get() {
    static Pool* p = new Pool();  // never destroyed: its threads may outlive static destructors
    return *p;
  }

  // f(i) for every i in [0, n), on at most `threads` threads of which the caller is one.
  // Returns when all calls have returned.  Callers on different threads may overlap.
  // An exception thrown by f (on any thread) stops the hand-out of further indices and is
  // rethrown on the caller once every helper has left the job: the job lives in this frame.
  template <class F>
  void Run(size_t n, int threads, F&& f) {
    if (n == 0) return;
    if (threads <= 1 || n < 2) {
      for (size_t i = 0; i < n; i++) f(i);
      return;
    }
    struct Ctx { F* f; } ctx{&f};
    Job job;
    job.n = n;
    job.arg = &ctx;
    job.call = [](void* a, size_t i) { (*static_cast<Ctx*>(a)->f)(i); };
    job.mu = &mu_;
    job.max_helpers = (int)std::min<size_t>((size_t)threads - 1, n - 1);
    {
      std::lock_guard<std::mutex> lk(mu_);
      Grow(job.max_helpers);
      jobs_.push_back(&job);
    }
    cv_.notify_all();
    Work(&job);
    {
      // no helper may still hold the job when it leaves this frame
      std::unique_lock<std::mutex> lk(mu_);
      for (size_t k = 0; k < jobs_.size(); k++)
        if (jobs_[k] == &job) { jobs_.erase(jobs_.begin() + (ptrdiff_t)k); break; }
      done_cv_.wait(lk, [&] { return job.helpers == 0; });
    }
    if (job.error) std::rethrow_exception(job.error);
  }

 private:
  struct Job {
    size_t n = 0;
    std::atomic<size_t> next{0};
    void* arg = nullptr;
    void (*call)(void*, size_t) = nullptr;
    int max_helpers = 0;
    int joined = 0;   // helpers that ever took the job (under mu_)
    int helpers = 0;  // helpers inside Work() now (under mu_)
    std::exception_ptr error;  // the first exception f threw (under *mu)
    std::mutex* mu = nullptr;
  };
  static constexpr int kMaxWorkers = 63;

  static void Work(Job* j) {
    for (;;) {
      const size_t i = j->next.fetch_add(1, std::memory_order_relaxed);
      if (i >= j->n) return;
      try {
        j->call(j->arg, i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(*j->mu);
        if (!j->error) j->error = std::current_exception();
        j->next.store(j->n, std::memory_order_relaxed);  // hand out nothing more
        return;
      }
    }
  }
  void Grow(int want) {  // under mu_
    want = std::min(want, kMaxWorkers);
    while (n_workers_ < want) {
      n_workers_++;
      std::thread([this] { Loop(); }).detach();
    }
  }
  void Loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      // the newest loop first: the short loops of the record pipeline wait on their helpers,
      // while a batch being inflated is a chunk ahead of its reader anyway
      Job* mine = nullptr;
      for (size_t k = jobs_.size(); k-- > 0;) {
        Job* j = jobs_[k];
        if (j->joined < j->max_helpers && j->next.load(std::memory_order_relaxed) < j->n) { mine = j; break; }
      }
      if (!mine) {
        cv_.wait(lk);
        continue;
      }
      mine->joined++;
      mine->helpers++;
      lk.unlock();
      Work(mine);
      lk.lock();
      if (--mine->helpers == 0) done_cv_.notify_all();
    }
  }

  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<Job*> jobs_;
  int n_workers_ = 0;
};

template <class F>
inline void parallel_for(size_t n, int threads, F&& f) {
  Pool::Get().Run(n, threads, static_cast<F&&>(f));
}

}  // namespace bp_par
