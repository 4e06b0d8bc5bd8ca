// hostpool.h -- worker threads for the host-side Scene build (triangle hierarchy, edge list, edge hierarchies).
//
// The build is a tree of short jobs (tens of microseconds to a millisecond); std::async(std::launch::async) starts an OS
// thread per job, ~50 us each and started one after the other: several milliseconds per Scene went into thread creation.
// This pool keeps its threads; run() returns a handle whose wait() HELPS -- it executes queued jobs while its own is not
// finished -- so jobs may spawn and wait for jobs (the recursive builders do) without starving the pool.
// The reference builds its structures with Thrust on the CPU thread pool it starts per call (src/parallel.cpp:228-255).
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include <pthread.h>
#include <sched.h>

namespace hostpool {

class Pool {
public:
    struct State {
        std::function<void()> fn;
        std::atomic<bool> done{false};
        std::exception_ptr error;
    };
    class Job {
    public:
        Job() = default;
        Job(Pool *p, std::shared_ptr<State> s) : pool_(p), st_(std::move(s)) {}
        Job(Job &&) = default;
        Job &operator=(Job &&) = default;
        ~Job() { if (st_) { try { wait(); } catch (...) {} } }       // a job never outlives what it refers to
        void wait() {
            if (!st_) return;
            std::shared_ptr<State> s = std::move(st_);
            pool_->help_until(*s);
            if (s->error) std::rethrow_exception(s->error);
        }
    private:
        Pool *pool_ = nullptr;
        std::shared_ptr<State> st_;
    };

    static Pool &get() { static Pool *p = new Pool(); return *p; }      // never destroyed: jobs may run during exit

    template <class F> Job run(F f) {
        auto s = std::make_shared<State>();
        s->fn = std::move(f);
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(s);
            queued_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_one();
        return Job(this, std::move(s));
    }
    int threads() const { return (int)workers_.size(); }

private:
    Pool() {
        unsigned hw = std::thread::hardware_concurrency();
        int cores = (int)(hw ? hw : 8);
        // one process per GPU: the ranks of a node share its cores (torch.distributed.run exports LOCAL_WORLD_SIZE), so a
        // rank's polling workers take their share of them, at most 15 -- 8 ranks x 31 pollers would occupy a 256-thread host
        int ranks = 1;
        if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, std::atoi(e));
        int n = std::max(2, cores / ranks) - 1;
        n = std::max(1, std::min(ranks > 1 ? 15 : 31, n));
        if (const char *e = std::getenv("RDR_POOL_THREADS")) n = std::max(1, std::min(63, std::atoi(e)));
        // RDR_POOL_PIN=k: workers restricted to the k-aligned block of k CPUs the creating thread runs on (phases hand their
        // data from thread to thread: cores that share a last-level cache pass it on without leaving the cache)
        int pin = 0;
        if (const char *e = std::getenv("RDR_POOL_PIN")) pin = std::atoi(e);
        cpu_set_t set;
        CPU_ZERO(&set);
        if (pin > 0) {
            const int here = sched_getcpu();
            if (here >= 0) { const int base = here - here % pin; for (int c = base; c < base + pin; ++c) CPU_SET(c, &set); } else pin = 0;
        }
        for (int i = 0; i < n; ++i) {
            workers_.emplace_back([this] { loop(); });
            if (pin > 0) (void)pthread_setaffinity_np(workers_.back().native_handle(), sizeof(set), &set);
            workers_.back().detach();
        }
    }
    static void execute(State &s) {
        try { s.fn(); } catch (...) { s.error = std::current_exception(); }
        s.fn = nullptr;
        s.done.store(true, std::memory_order_release);
    }
    bool try_one() {
        std::shared_ptr<State> s;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (q_.empty()) return false;
            s = std::move(q_.back());          // newest first: the job a waiter needs was queued last
            q_.pop_back();
            queued_.fetch_sub(1, std::memory_order_release);
        }
        execute(*s);
        return true;
    }
    void help_until(State &want) {
        // the jobs of a build last tens of microseconds: a waiter polls (and helps) instead of sleeping
        int idle = 0;
        while (!want.done.load(std::memory_order_acquire)) {
            if (queued_.load(std::memory_order_acquire) > 0 && try_one()) { idle = 0; continue; }
            if (++idle < 20000) { relax(); continue; }
            std::this_thread::yield();
        }
    }
    void loop() {
        for (;;) {
            std::shared_ptr<State> s;
            // A build is a burst of short jobs a few microseconds apart: after a job a worker polls for the next one for a
            // while (~100 us) before it goes to sleep on the condition variable -- waking a sleeper costs more than most jobs.
            for (int spin = 0; spin < 4000 && queued_.load(std::memory_order_acquire) == 0; ++spin) relax();
            {
                std::unique_lock<std::mutex> lk(m_);
                if (q_.empty()) {
                    sleepers_.fetch_add(1, std::memory_order_acq_rel);
                    cv_.wait(lk, [&] { return !q_.empty(); });
                    sleepers_.fetch_sub(1, std::memory_order_acq_rel);
                }
                s = std::move(q_.front());
                q_.pop_front();
                queued_.fetch_sub(1, std::memory_order_release);
            }
            execute(*s);
        }
    }
    static void relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    std::mutex m_;
    std::atomic<int> queued_{0}, sleepers_{0};
    std::condition_variable cv_;
    std::deque<std::shared_ptr<State>> q_;
    std::vector<std::thread> workers_;
};

template <class F> inline Pool::Job run(F f) { return Pool::get().run(std::move(f)); }

// f(begin, end) over [0, n) in up to 32 chunks of at least `grain` items; the caller takes the first chunk.
template <class F> inline void parallel_chunks(int n, int grain, F f) {
    int chunks = (n + grain - 1) / grain;
    if (chunks > 32) chunks = 32;
    if (chunks <= 1) { if (n > 0) f(0, n); return; }
    std::vector<Pool::Job> jobs;
    jobs.reserve((size_t)chunks - 1);
    for (int c = 1; c < chunks; ++c) {
        int b = (int)((long long)n * c / chunks), e = (int)((long long)n * (c + 1) / chunks);
        jobs.push_back(run([=] { f(b, e); }));
    }
    f(0, (int)((long long)n / chunks));
    for (auto &j : jobs) j.wait();
}

} // namespace hostpool
