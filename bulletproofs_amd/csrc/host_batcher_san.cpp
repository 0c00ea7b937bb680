// CPU-only harness around the batch provers' rendezvous (fbbatcher.h), built with -fsanitize=thread and with
// -fsanitize=address,undefined by `make san` and run by tests/test_host_sanitizers.py.  It drives StepRendezvous the way
// prove_batch_impl (proto.cpp) does: N threads, each submitting one or two jobs per step for several steps, some threads
// leaving early (a proof that fails), a step whose run() reports an error, a step with no jobs at all.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "fbbatcher.h"

using bp::StepRendezvous;

struct Job {
    int v;
};

static int check(bool ok, const char* what) {
    if (!ok) fprintf(stderr, "FAIL: %s\n", what);
    return ok ? 0 : 1;
}

int main() {
    int fails = 0;
    for (int round = 0; round < 20; round++) {
        const int N = 3 + round * 3, STEPS = 12;
        std::atomic<int> launches{0}, bad{0};
        StepRendezvous<Job, long> rv([&](const std::vector<const Job*>& all, long* res) {
            launches++;
            bool poison = false;
            for (size_t i = 0; i < all.size(); i++) {
                res[i] = 3L * all[i]->v + 1;
                poison = poison || all[i]->v < 0;
            }
            return poison ? -1 : 0;
        });
        rv.active = N;
        rv.pj.reserve(N);
        rv.po.reserve(N);
        std::vector<std::thread> thr;
        for (int t = 0; t < N; t++) {
            thr.emplace_back([&, t] {
                for (int s = 0; s < STEPS; s++) {
                    if (t % 5 == 4 && s == 3 + t % 4) break;  // this "proof" fails on its own and leaves early
                    std::vector<Job> jobs;
                    if (!(s == 7))  // step 7: nobody submits a job, the flush must still release everyone
                        for (int j = 0; j < 1 + (t + s) % 2; j++) jobs.push_back(Job{1000 * t + 10 * s + j});
                    if (s == 9 && t == 0) jobs.push_back(Job{-1});  // step 9 fails as a whole
                    std::vector<long> out(jobs.size() + 1, -7);
                    int rc = rv.submit(jobs, out.data());
                    if (s == 9) {
                        if (rc != -1) bad++;
                    } else {
                        if (rc != 0) bad++;
                        for (size_t j = 0; j < jobs.size(); j++)
                            if (out[j] != 3L * jobs[j].v + 1) bad++;
                    }
                }
                rv.leave();
            });
        }
        for (auto& t : thr) t.join();
        fails += check(bad == 0, "results / return codes");
        fails += check(launches == STEPS - 1, "one run() per step with jobs (step 7 has none)");
        fails += check(rv.active == 0 && rv.waiting == 0, "everybody left");
    }
    if (fails == 0) printf("host_batcher_san OK\n");
    return fails ? 1 : 0;
}
