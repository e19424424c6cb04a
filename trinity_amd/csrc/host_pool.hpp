// host_pool.hpp — a small persistent pool of host threads for the planner (planner.hpp): tri_batch_create lowers, classifies and cuts a
// batch's queries in parallel, and a batch is compiled per step by callers that do not keep batches around, so the threads outlive
// the call.  Host-only C++17, no HIP; new code, no reference source.
//
// What the measurements on the micro-VMs this engine runs in said (tools/plan_probe.py): starting a std::thread costs
// 4 - 5 ms; a futex wake-up puts the woken thread on the WAKER's CPU (the guest exposes no cache topology, so the scheduler does not look
// for an idle core), i.e. seven woken workers ran one after the other on one core; sched_yield() sleeps for milliseconds.  Hence: workers
// are pinned to distinct CPUs of the process's affinity mask (a rank's own slice of it under a one-process-per-GPU launcher, else the CPUs
// next to the creating thread's: pin_candidates), they keep POLLING for `hot_us` microseconds after their last job before they sleep on a condition variable
// (a caller that compiles a batch per step finds them awake: a job is picked up in about a microsecond; from sleep it takes 100 - 300 us),
// and nobody yields.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <thread>
#include <vector>

// The CPUs this process may USE at once: the affinity mask's size, capped by the cgroup's CPU bandwidth quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) —
// a container that shows 256 CPUs may be allowed 16 of them per period, and a process whose polling threads use more is THROTTLED: every thread stopped until the
// period's end (measured on the GPU box, quota 16: two pools of 15 pollers -> 86 of 250 periods throttled, one tri_batch_create in fifty took 50 - 60 ms).
inline unsigned host_cpu_budget() {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (!sched_getaffinity(0, sizeof allowed, &allowed))
                n = std::max(1, CPU_COUNT(&allowed));
        auto quota = [](const char *path, const char *period_path) -> double {
                FILE *f = fopen(path, "r");
                if (!f)
                        return 0.0;
                char a[64] = {0}, b[64] = {0};
                const int got = fscanf(f, "%63s %63s", a, b);
                fclose(f);
                if (got < 1 || a[0] == 'm' || a[0] == '-') // ("max" / -1: no quota)
                        return 0.0;
                double q = atof(a), p = got >= 2 ? atof(b) : 0.0;
                if (period_path) {
                        FILE *g = fopen(period_path, "r");
                        if (g) {
                                if (fscanf(g, "%63s", b) == 1)
                                        p = atof(b);
                                fclose(g);
                        }
                }
                return q > 0 && p > 0 ? q / p : 0.0;
        };
        double q = quota("/sys/fs/cgroup/cpu.max", nullptr);
        if (q <= 0)
                q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
        if (q > 0)
                n = std::min<unsigned>(n, std::max(1u, (unsigned)q));
        // one process per GPU (LOCAL_WORLD_SIZE ranks on this node share the mask and the quota): a rank's share
        const char *lw = getenv("LOCAL_WORLD_SIZE");
        const long w = lw ? strtol(lw, nullptr, 10) : 0;
        if (w > 1)
                n = std::max(1u, n / (unsigned)w);
        return n;
}

class HostPool {
      public:
        // `threads` includes the calling thread: threads - 1 workers are started (fewer when thread creation fails — the pool then
        // simply has fewer hands; run() still completes on the caller alone)
        // `part` of `nparts`: a process that keeps several pools (tri_dev's planner contexts: two batches compiled side by side) gives each its own
        // stretch of the candidate CPUs — a rank's slice is cut into nparts contiguous pieces; without a slice pool `part` starts `part * threads`
        // CPUs past the creating thread's — so that no two pools' pollers share a CPU.
        // `anchor` (>= 0): the candidate index the FIRST pool of the process started from (anchor()): the pools are created by different threads, on
        // whatever CPUs those happen to run — each starting from its own creator's CPU, two pools' stretches overlapped now and then, and two pollers on
        // one CPU cost a planning pass its time slice (measured: one create in a hundred took 10 - 20 ms, a single pass of it 17 ms).
        // `spread`: a worker's affinity is the pool's whole STRETCH of CPUs (its own CPU first in line, twice as many CPUs as workers where the mask has
        // them) instead of the one CPU — on a large shared host a worker pinned to ONE CPU cannot be moved when something else is given that CPU, and a
        // worker parked with a fragment in hand costs the planning pass a time slice (measured on a 256-CPU box, load 20: one create in a hundred took
        // 10 - 58 ms, all of it one pass waiting for its last fragment); within a stretch the scheduler finds it another CPU.
        explicit HostPool(unsigned threads, bool pin = true, unsigned hot_us = 3000, unsigned part = 0, unsigned nparts = 1, int anchor = -1, bool spread = false) : hot_us_(hot_us) {
                std::vector<int> cpus;
                int base = 0;
                if (pin) {
                        cpus = pin_candidates(&base);
                        if (base >= 0 && anchor >= 0)
                                base = anchor;
                        anchor_ = base;
                        if (nparts > 1 && part < nparts) {
                                if (base < 0 && cpus.size() >= nparts) { // (a rank's slice: this pool's piece of it)
                                        const size_t lo = cpus.size() * part / nparts, hi = cpus.size() * (part + 1) / nparts;
                                        cpus = std::vector<int>(cpus.begin() + (long)lo, cpus.begin() + (long)hi);
                                } else if (base >= 0)
                                        base += (int)(part * threads * (spread ? 2u : 1u));
                        }
                        if (threads > cpus.size() + 1 && !cpus.empty()) // (a rank's slice may be narrower than the threads asked for: no two pollers on one CPU)
                                threads = (unsigned)cpus.size() + 1;
                        if (base < 0 && cpus.size() < 2) // a one-CPU slice: the unpinned caller would share that CPU with a polling worker — the caller plans alone
                                threads = 1;
                }
                for (unsigned i = 1; i < threads; ++i) {
                        try {
                                workers_.emplace_back([this] { loop(); });
                        } catch (...) {
                                break;
                        }
                        if (cpus.size() > 1 || (base < 0 && !cpus.empty())) { // (a rank's slice is honoured even when it is a single CPU)
                                const int cpu = cpus[(size_t)(base + (int)i) % cpus.size()];
                                cpu_set_t s;
                                CPU_ZERO(&s);
                                CPU_SET(cpu, &s);
                                if (spread) { // (the stretch: this pool's CPUs — a rank's whole piece of its slice, else 2 x threads CPUs from the pool's start)
                                        const size_t span = base < 0 ? cpus.size() : std::min<size_t>(cpus.size(), 2u * threads);
                                        for (size_t k = 0; k < span; ++k)
                                                CPU_SET(cpus[(size_t)(std::max(base, 0) + (int)k) % cpus.size()], &s);
                                }
                                pthread_setaffinity_np(workers_.back().native_handle(), sizeof s, &s); // (best effort)
                                pinned_.push_back(cpu);
                        }
                }
        }
        // The CPUs a pool of this process pins its workers to, and where among them it starts (*base: the worker i goes to
        // cpus[(base + i) % cpus.size()]).  One process per GPU is the deployment (torch.distributed.run / any launcher that exports LOCAL_RANK and
        // LOCAL_WORLD_SIZE): the ranks of a node then take DISJOINT contiguous slices of the affinity mask — slice r of LOCAL_WORLD_SIZE — whatever
        // CPU each rank's creating thread happens to run on (eight ranks started side by side land within a few CPUs of each other: their fifteen
        // spinning workers each would otherwise pile onto the same cores).  Without those variables — or with values that do not parse as
        // 0 <= LOCAL_RANK < LOCAL_WORLD_SIZE <= the mask's CPUs (strtol: a non-numeric value reads as 0 / fails the range check) — the CPUs
        // next to the creating thread's, as for a single process.
        static std::vector<int> pin_candidates(int *base) {
                std::vector<int> cpus;
                cpu_set_t allowed;
                CPU_ZERO(&allowed);
                if (!sched_getaffinity(0, sizeof allowed, &allowed))
                        for (int c = 0; c < CPU_SETSIZE; ++c)
                                if (CPU_ISSET(c, &allowed))
                                        cpus.push_back(c);
                *base = 0;
                const char *lr = getenv("LOCAL_RANK"), *lw = getenv("LOCAL_WORLD_SIZE");
                const long r = lr ? strtol(lr, nullptr, 10) : -1, w = lw ? strtol(lw, nullptr, 10) : 0;
                if (w > 1 && r >= 0 && r < w && cpus.size() >= (size_t)w) {
                        const size_t lo = cpus.size() * (size_t)r / (size_t)w, hi = cpus.size() * (size_t)(r + 1) / (size_t)w;
                        cpus = std::vector<int>(cpus.begin() + (long)lo, cpus.begin() + (long)hi);
                        *base = -1; // (worker 1 takes the slice's first CPU)
                        return cpus;
                }
                const int here = sched_getcpu();
                for (size_t i = 0; i < cpus.size(); ++i)
                        if (cpus[i] == here)
                                *base = (int)i;
                return cpus;
        }
        const std::vector<int> &pinned_cpus() const { return pinned_; } // (worker i + 1's CPU)
        int anchor() const { return anchor_; }                          // where this pool's stretch of the candidates starts (-1: a rank's slice): hand it to the process's next pool
        ~HostPool() {
                {
                        std::lock_guard<std::mutex> g(m_);
                        stop_ = true;
                        ++gen_;
                        gen_hint_.store(gen_, std::memory_order_release);
                }
                cv_.notify_all();
                for (auto &t : workers_)
                        t.join();
        }
        HostPool(const HostPool &) = delete;
        HostPool &operator=(const HostPool &) = delete;
        unsigned size() const { return (unsigned)workers_.size() + 1; }

        // fn(k) for every k in [0, n), dealt out dynamically over the workers and the caller; returns when all are done.  fn must not
        // throw (the planner's bodies catch and record).  Not re-entrant: one run() at a time per pool (one tri_dev per host thread).
        void run(unsigned n, const std::function<void(unsigned)> &fn) {
                if (!n)
                        return;
                if (n == 1 || workers_.empty()) {
                        for (unsigned k = 0; k < n; ++k)
                                fn(k);
                        return;
                }
                uint64_t gen;
                bool wake;
                {
                        std::lock_guard<std::mutex> g(m_);
                        gen = ++gen_;
                        // a job is taken by a compare-and-swap on (generation, next index): a worker that wakes up late for an earlier
                        // generation can neither take nor skip a job of this one.  The ticket is CLOSED (index 2^32 - 1: beyond any n) in the
                        // new generation BEFORE fn_ / n_ / left_ change: a straggler still inside work(gen - 1) that holds the old
                        // (generation, n_old) and then reads the NEW, larger n_ can no longer win its CAS on the old word — it would have
                        // run fn(n_old) of the new job, a second time, and taken left_ to zero one job early.
                        state_.store(gen << 32 | 0xffffffffull, std::memory_order_seq_cst);
                        fn_ = &fn;
                        n_.store(n, std::memory_order_relaxed);
                        left_.store(n, std::memory_order_relaxed);
                        state_.store(gen << 32, std::memory_order_release);
                        gen_hint_.store(gen, std::memory_order_release); // (the polling workers see this)
                        wake = sleepers_ != 0;
                }
                if (wake)
                        cv_.notify_all();
                static const bool dbg = getenv("TRINITY_DEBUG_POOL") != nullptr; // (stderr: a run() of more than 3 ms, job by job — who took it, when, on which CPU)
                const auto t_run = std::chrono::steady_clock::now();
                if (dbg) {
                        dbg_jobs_.assign(n, DbgJob{});
                        dbg_t0_ = t_run;
                        dbg_on_.store(true, std::memory_order_release);
                }
                work(gen);
                // (the caller waits for the stragglers by polling, and on the clock should a worker have been descheduled)
                for (uint32_t spin = 0; left_.load(std::memory_order_acquire); ++spin)
                        if (spin > (1u << 22)) {
                                std::unique_lock<std::mutex> g(m_);
                                done_cv_.wait_for(g, std::chrono::microseconds(200), [&] { return !left_.load(std::memory_order_acquire); });
                        }
                if (dbg) {
                        dbg_on_.store(false, std::memory_order_release);
                        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count();
                        if (ms > 3.0) {
                                fprintf(stderr, "[tri pool] run of %u jobs took %.3f ms (woke sleepers: %d):", n, ms, (int)wake);
                                for (unsigned k = 0; k < n; ++k)
                                        fprintf(stderr, " [%u: %.3f-%.3f ms tid %d cpu %d->%d]", k, dbg_jobs_[k].t0, dbg_jobs_[k].t1, dbg_jobs_[k].tid, dbg_jobs_[k].c0, dbg_jobs_[k].c1);
                                fprintf(stderr, "\n");
                        }
                }
        }

      private:
        void work(const uint64_t gen) {
                uint64_t cur = state_.load(std::memory_order_acquire);
                for (;;) {
                        if ((cur >> 32) != (gen & 0xffffffffull) || (uint32_t)cur >= n_.load(std::memory_order_relaxed))
                                return;
                        if (!state_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel, std::memory_order_acquire))
                                continue;
                        const unsigned job = (unsigned)(uint32_t)cur;
                        const bool dbg = dbg_on_.load(std::memory_order_acquire) && job < dbg_jobs_.size();
                        if (dbg) {
                                dbg_jobs_[job].t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0_).count();
                                dbg_jobs_[job].c0 = sched_getcpu();
                                dbg_jobs_[job].tid = (int)(uintptr_t)pthread_self() & 0xffff;
                        }
                        (*fn_)(job);
                        if (dbg) {
                                dbg_jobs_[job].t1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0_).count();
                                dbg_jobs_[job].c1 = sched_getcpu();
                        }
                        if (left_.fetch_sub(1, std::memory_order_acq_rel) == 1)
                                done_cv_.notify_all();
                        cur = state_.load(std::memory_order_acquire);
                }
        }
        void loop() {
                using clk = std::chrono::steady_clock;
                uint64_t seen = 0;
                for (;;) {
                        // poll for the next generation while the pool is hot, then sleep
                        const auto t0 = clk::now();
                        uint64_t g = gen_hint_.load(std::memory_order_acquire);
                        for (uint32_t spin = 0; g == seen; ++spin) {
                                if (!(spin & 1023u) && std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count() >= (long)hot_us_)
                                        break;
#if defined(__x86_64__) || defined(__i386__)
                                __builtin_ia32_pause();
#endif
                                g = gen_hint_.load(std::memory_order_acquire);
                        }
                        if (g == seen) {
                                std::unique_lock<std::mutex> lk(m_);
                                ++sleepers_;
                                cv_.wait(lk, [&] { return gen_ != seen; });
                                --sleepers_;
                                g = gen_;
                        }
                        seen = g;
                        if (stop_flag())
                                return;
                        work(seen);
                }
        }
        bool stop_flag() {
                std::lock_guard<std::mutex> g(m_);
                return stop_;
        }
        struct DbgJob {
                double t0 = 0, t1 = 0;
                int tid = 0, c0 = -1, c1 = -1;
        };
        std::vector<DbgJob> dbg_jobs_; // (TRINITY_DEBUG_POOL)
        std::chrono::steady_clock::time_point dbg_t0_;
        std::atomic<bool> dbg_on_{false};
        std::vector<std::thread> workers_;
        std::vector<int> pinned_;
        int anchor_ = -1;
        std::mutex m_;
        std::condition_variable cv_, done_cv_;
        const std::function<void(unsigned)> *fn_ = nullptr; // (written under m_ before state_ is published; read only after a successful CAS)
        std::atomic<uint32_t> n_{0};
        std::atomic<uint64_t> state_{0}; // generation << 32 | next job
        std::atomic<unsigned> left_{0};
        std::atomic<uint64_t> gen_hint_{0};
        uint64_t gen_ = 0;
        unsigned sleepers_ = 0; // (under m_)
        unsigned hot_us_;
        bool stop_ = false;
};
