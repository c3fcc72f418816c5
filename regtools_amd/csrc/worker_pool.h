// worker_pool.h -- host threads that outlive one parallel phase.  A phase of the text loaders is a millisecond or two of work per thread; starting
// 31 std::threads for it costs about as much again (they are created one after the other), and a loader has six to eight such phases.  The pool's
// threads are started once per load and woken per phase; tasks are handed out through one atomic counter, the caller works too.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rgx {

class WorkerPool {
  public:
    explicit WorkerPool(size_t threads) {
        for (size_t k = 1; k < threads; ++k) workers_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; ++phase_; }
        wake_.notify_all();
        for (auto &t : workers_) t.join();
    }
    WorkerPool(const WorkerPool &) = delete;
    WorkerPool &operator=(const WorkerPool &) = delete;
    size_t threads() const { return workers_.size() + 1; }
    // f(k) for k in [0, n), each exactly once, on the pool's threads and the caller; returns when all are done.  An exception thrown by a task
    // (bad_alloc from a task's vectors, say) is caught where it happens, the phase still runs to its end -- the workers hold pointers into the
    // caller's frame until then -- and the first one is thrown again here.  A run() from inside a task of the same pool (or from a second thread
    // while one is in flight) does not queue behind itself: it runs its tasks on the calling thread.
    void run(size_t n, const std::function<void(size_t)> &f) {
        if (n == 0) return;
        if (n == 1 || workers_.empty()) { serial(n, f); return; }
        {
            std::unique_lock<std::mutex> g(m_);
            if (in_run_) { g.unlock(); serial(n, f); return; }
            in_run_ = true; error_ = nullptr;
            f_ = &f; n_ = n; next_.store(0, std::memory_order_relaxed); pending_ = workers_.size(); ++phase_;
        }
        wake_.notify_all();
        work();
        std::exception_ptr e;
        {
            std::unique_lock<std::mutex> g(m_);
            done_.wait(g, [this] { return pending_ == 0; });
            f_ = nullptr; in_run_ = false; e = error_; error_ = nullptr;
        }
        if (e) std::rethrow_exception(e);
    }

  private:
    // (the same contract without helpers: every task runs, the first exception is thrown at the end)
    static void serial(size_t n, const std::function<void(size_t)> &f) {
        std::exception_ptr e;
        for (size_t k = 0; k < n; ++k) { try { f(k); } catch (...) { if (!e) e = std::current_exception(); } }
        if (e) std::rethrow_exception(e);
    }
    void work() {
        for (size_t k; (k = next_.fetch_add(1, std::memory_order_relaxed)) < n_;) {
            try { (*f_)(k); }
            catch (...) { std::lock_guard<std::mutex> g(m_); if (!error_) error_ = std::current_exception(); }      // (the other tasks still run: the counter drains)
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return phase_ != seen; });
                seen = phase_;
                if (stop_) return;
            }
            work();
            { std::lock_guard<std::mutex> g(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(size_t)> *f_ = nullptr;
    size_t n_ = 0, pending_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t phase_ = 0;
    bool stop_ = false, in_run_ = false;
    std::exception_ptr error_;
};

// std::sort's result on [b, e), by the pool: runs sorted side by side, then merged pairwise (a stable merge of sorted runs of distinct keys --
// callers whose keys can tie and who care about the order of ties put the tie-breaker into `less`)
template <class It, class Less>
void parallel_sort(WorkerPool &pool, It b, It e, Less less) {
    const size_t n = (size_t)(e - b);
    const size_t runs = n < (1u << 14) ? 1 : std::min<size_t>(pool.threads(), 16);
    if (runs <= 1) { std::sort(b, e, less); return; }
    pool.run(runs, [&](size_t r) { std::sort(b + (long)(n * r / runs), b + (long)(n * (r + 1) / runs), less); });
    for (size_t w = 1; w < runs; w *= 2) {
        std::vector<size_t> starts;
        for (size_t r = 0; r + w < runs; r += 2 * w) starts.push_back(r);
        pool.run(starts.size(), [&](size_t q) {
            const size_t r = starts[q];
            std::inplace_merge(b + (long)(n * r / runs), b + (long)(n * (r + w) / runs), b + (long)(n * std::min(runs, r + 2 * w) / runs), less);
        });
    }
}


// Things whose teardown is page-table or allocator work the caller need not wait for (unmapping a 500 MB file costs 8 ms, handing back the
// tables of a 250 k-transcript annotation 4 ms): handed to ONE background thread of the process, in order.  drain() = wait until everything
// handed over so far is gone (a context being destroyed; the process ending -- registered with atexit when the thread starts).  Round 5: what is
// handed over is HOST teardown only (munmap, vectors, strings): a hipFree on this thread synchronised the device in the middle of the caller's next call,
// and its order against the HIP runtime's own exit handlers was an assumption.
class Reaper {
  public:
    static Reaper &get() { static Reaper *r = new Reaper(); return *r; }      // (never destroyed: no static-destruction order to get wrong)
    void later(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> g(m_);
            if (!started_) { started_ = true; std::thread([this] { loop(); }).detach(); atexit([] { Reaper::get().drain(); }); }
            q_.push_back(std::move(f)); ++queued_;
        }
        wake_.notify_one();
    }
    void drain() {
        std::unique_lock<std::mutex> g(m_);
        const uint64_t want = queued_;
        idle_.wait(g, [&] { return done_ >= want; });
    }

  private:
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return !q_.empty(); });
                f = std::move(q_.front()); q_.erase(q_.begin());
            }
            try { f(); } catch (...) {}
            { std::lock_guard<std::mutex> g(m_); ++done_; }
            idle_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable wake_, idle_;
    std::vector<std::function<void()>> q_;
    uint64_t queued_ = 0, done_ = 0;
    bool started_ = false;
};

}  // namespace rgx
