// Host-side worker pool of the library (C++14, header only, no GPU code).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace qadc {

// A few persistent host threads for the heap replay of a batch's queries (queries are independent; the caller still
// drives the library from one thread).  Spawning threads per batch costs ~50 us each — as much as replaying a dozen
// queries — so the workers are started once and woken per batch; the calling thread takes tasks too.
class WorkerPool {
  public:
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto& t : th_) t.join();
    }
    // runs f(t) for t = 0 .. tasks-1 on up to `threads` threads (this one included)
    template <typename F>
    void run(int tasks, int threads, F&& f) {
        threads = std::max(1, std::min(threads, tasks));
        if (threads == 1) {
            for (int t = 0; t < tasks; ++t) f(t);
            return;
        }
        while ((int)th_.size() < threads - 1) th_.emplace_back([this] { loop(); });
        std::function<void(int)> fn = f;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            tasks_ = tasks;
            next_.store(0);
            helpers_ = threads - 1;
            active_ = 0;
            ++gen_;
        }
        cv_work_.notify_all();
        for (int t; (t = next_.fetch_add(1)) < tasks;) fn(t);
        std::unique_lock<std::mutex> lk(m_);
        helpers_ = 0;                                           // no worker starts on this job any more
        cv_done_.wait(lk, [this] { return active_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || (gen_ != seen && helpers_ > 0); });
            if (stop_) return;
            seen = gen_;
            --helpers_;
            ++active_;
            std::function<void(int)>* fn = job_;
            const int tasks = tasks_;
            lk.unlock();
            for (int t; (t = next_.fetch_add(1)) < tasks;) (*fn)(t);
            lk.lock();
            if (--active_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::function<void(int)>* job_ = nullptr;
    std::atomic<int> next_{0};
    int tasks_ = 0, helpers_ = 0, active_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

}  // namespace qadc
