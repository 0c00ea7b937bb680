// Rendezvous of the batch provers (proto.cpp): N host threads run the unchanged single-proof code; the jobs they submit
// at the same protocol step are merged and executed by ONE call of `run` (one GPU launch), by whichever thread arrives
// last; everybody gets the step's return code.  Pure host code with no HIP dependency, so the CPU tests run it under
// ThreadSanitizer / AddressSanitizer (host_batcher_san.cpp, `make san`).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <vector>

namespace bp {

template <class Job, class Out>
struct StepRendezvous {
    // run(all jobs of the step in submission order, one result slot per job) -> 0 on success
    typedef std::function<int(const std::vector<const Job*>&, Out*)> RunFn;
    explicit StepRendezvous(RunFn fn) : run(std::move(fn)) {}

    RunFn run;
    std::mutex mu;
    std::condition_variable cv;
    int active = 0, waiting = 0;  // threads in the batch / threads blocked in submit()
    uint64_t gen = 0;             // step counter: a waiter leaves when it changes
    int rc = 0;
    std::vector<const std::vector<Job>*> pj;
    std::vector<Out*> po;

    void flush_locked() {  // mu held; every active thread is waiting
        try {
            std::vector<const Job*> all;
            for (auto* v : pj)
                for (auto& j : *v) all.push_back(&j);
            const size_t total = all.size();
            std::vector<Out> res(total);
            rc = total ? run(all, res.data()) : 0;
            if (rc == 0) {
                size_t o = 0;
                for (size_t i = 0; i < pj.size(); i++)
                    for (size_t j = 0; j < pj[i]->size(); j++) po[i][j] = res[o++];
            }
        } catch (...) {  // host allocation failure: every proof of this step fails, none is left waiting
            rc = -5;     // BP_ENOMEM
        }
        pj.clear();
        po.clear();
        waiting = 0;
        gen++;
        cv.notify_all();
    }
    int submit(const std::vector<Job>& jobs, Out* out) {
        std::unique_lock<std::mutex> lk(mu);
        pj.push_back(&jobs);
        po.push_back(out);
        waiting++;
        if (waiting == active) {
            flush_locked();
            return rc;
        }
        const uint64_t my = gen;
        cv.wait(lk, [&] { return gen != my; });
        return rc;
    }
    void leave() {  // a proof finished (or failed): the others must not wait for it
        std::unique_lock<std::mutex> lk(mu);
        active--;
        if (waiting > 0 && waiting == active) flush_locked();
    }
};

}  // namespace bp
