// CPU-only harness around the single-proof calls' host worker pool (hostpool.h), built with -fsanitize=thread and with
// -fsanitize=address,undefined by `make san` and run by tests/test_host_sanitizers.py.  It drives host_blocks / host_blocks_n
// the way proto.cpp does: passes of many sizes back to back (workers still spinning), passes after a pause (workers asleep on
// the condition variable), block sums added under a lock, and several caller threads at once (one owns the pool, the others'
// passes run inline).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include "hostpool.h"

static int check(bool ok, const char* what) {
    if (!ok) fprintf(stderr, "FAIL: %s\n", what);
    return ok ? 0 : 1;
}

// sum_{i<n} (i * 2654435761 mod 2^32) and a written-once array, through host_blocks
static int one_pass(size_t n, bool allow, size_t nthr_forced) {
    std::vector<uint32_t> out(n, 0);
    std::mutex mu;
    uint64_t total = 0;
    auto fn = [&](size_t lo, size_t hi) {
        uint64_t s = 0;
        for (size_t i = lo; i < hi; i++) {
            out[i] += (uint32_t)(i * 2654435761u);
            s += out[i];
        }
        std::lock_guard<std::mutex> lk(mu);
        total += s;
    };
    if (nthr_forced)
        bp::host_blocks_n(n, nthr_forced, fn);
    else
        bp::host_blocks(n, allow, fn);
    uint64_t want = 0;
    int fails = 0;
    for (size_t i = 0; i < n; i++) {
        want += (uint32_t)(i * 2654435761u);
        if (out[i] != (uint32_t)(i * 2654435761u)) fails = 1;  // every index exactly once
    }
    return fails + (total == want ? 0 : 1);
}

int main() {
    int fails = 0;
    const size_t sizes[] = {0, 1, 2, 3, 7, 16, 511, 512, 513, 1023, 1024, 1025, 4096, 5000};
    for (int round = 0; round < 40; round++)
        for (size_t n : sizes) {
            fails += check(one_pass(n, true, 0) == 0, "host_blocks (workers allowed)");
            fails += check(one_pass(n, false, 0) == 0, "host_blocks (inline)");
            if (n) fails += check(one_pass(n, true, 4) == 0, "host_blocks_n, four blocks");
            if (n) fails += check(one_pass(n, true, 2) == 0, "host_blocks_n, two blocks");
        }
    // workers asleep between passes
    for (int i = 0; i < 5; i++) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        fails += check(one_pass(2048, true, 0) == 0, "pass after a pause");
    }
    // several callers at once
    std::atomic<int> bad{0};
    std::vector<std::thread> thr;
    for (int t = 0; t < 6; t++)
        thr.emplace_back([&bad, t] {
            for (int i = 0; i < 200; i++)
                if (one_pass(1024 + 37 * (size_t)((t + i) % 11), true, 0) != 0) bad++;
        });
    for (auto& t : thr) t.join();
    fails += check(bad.load() == 0, "concurrent callers");
    if (fails) return 1;
    printf("host_pool_san OK\n");
    return 0;
}
