// Small re-entrant worker pool for the host-side octree stage.  parallel_for may be called from
// several host threads at once (the reference runs the left and right extractor on two threads,
// src/Frame.cc:127-130); the caller takes part in its own job, so a pool of zero workers still works.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace ft {

class ThreadPool {
public:
    explicit ThreadPool(int nworkers) {
        for (int i = 0; i < nworkers; i++) workers_.emplace_back([this, i] { run(i + 1); });
    }
    ~ThreadPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int size() const { return (int)workers_.size() + 1; }  // workers + the calling thread

    // fn(index, worker_id) for index in [0, n); worker_id in [0, size()) - 0 is the calling thread.
    // NOTE: with concurrent callers worker_id 0 is shared by all callers, so per-worker scratch must
    // be owned by the caller's job (see Job::callerScratch use in extractor.cpp).
    void parallel_for(int n, const std::function<void(int, int)> &fn) {
        if (n <= 0) return;
        auto job = std::make_shared<Job>();
        job->n = n;
        job->fn = &fn;
        if (!workers_.empty() && n > 1) {
            {
                std::lock_guard<std::mutex> lk(m_);
                jobs_.push_back(job);
            }
            cv_.notify_all();
        }
        work(*job, 0);
        // wait for indices still running on workers
        std::unique_lock<std::mutex> lk(job->dm);
        job->dcv.wait(lk, [&] { return job->done.load() >= job->n; });
    }

private:
    struct Job {
        int n = 0;
        const std::function<void(int, int)> *fn = nullptr;
        std::atomic<int> next{0};
        std::atomic<int> done{0};
        std::mutex dm;
        std::condition_variable dcv;
    };
    void work(Job &j, int wid) {
        for (;;) {
            const int i = j.next.fetch_add(1);
            if (i >= j.n) break;
            (*j.fn)(i, wid);
            if (j.done.fetch_add(1) + 1 >= j.n) {
                std::lock_guard<std::mutex> lk(j.dm);
                j.dcv.notify_all();
            }
        }
    }
    void run(int wid) {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
                if (stop_) return;
                job = jobs_.front();
                if (job->next.load() >= job->n) {  // exhausted: retire it and look again
                    jobs_.pop_front();
                    continue;
                }
            }
            work(*job, wid);
        }
    }
    std::vector<std::thread> workers_;
    std::deque<std::shared_ptr<Job>> jobs_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
};

}  // namespace ft
