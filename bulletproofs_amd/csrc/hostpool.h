// Host worker pool of the protocol layer's single-proof calls (proto.cpp): block-parallel passes over index ranges between two
// GPU calls.  Pure C++ (no HIP): built on its own under -fsanitize=thread / address by `make san` (host_pool_san.cpp).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>

namespace bp {

// fn(lo, hi) over [0, n) on up to four host threads: the per-round loops of the single-proof provers are 4 nm independent
// scalar products - ~110 us at nm = 1024, ~250 us at nm = 4096 - between two GPU calls that take 65-85 us.  Starting threads
// per round costs more than it saves below nm = 4096 (C2 2.33 -> 2.52 ms with two std::threads per pass), so three workers are
// started once per process and kept: they spin for a new pass for 300 us after the last one (a proof's rounds come every
// 100-200 us), then sleep on a condition variable.  One user at a time (another caller's pass runs inline); not under a batch
// prover's rendezvous, whose callers are already one thread per proof.  The pool is never destroyed (detached threads).
struct HostPool {
    static const int NW = 3;
    std::mutex mu, use;
    std::condition_variable cv;
    std::atomic<uint64_t> gen{0};
    std::atomic<int> pending{0};
    std::function<void(size_t, size_t)> fn;
    size_t n = 0, nthr = 0;
    HostPool() {
        for (int w = 0; w < NW; w++) std::thread([this, w] { worker(w); }).detach();
    }
    static void relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    void worker(int w) {
        uint64_t seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            while (gen.load(std::memory_order_acquire) == seen) {
                relax();
                if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen; });
                }
            }
            seen = gen.load(std::memory_order_acquire);
            const size_t me = (size_t)w + 1;
            if (me < nthr) {
                const size_t per = n / nthr;
                fn(me * per, me == nthr - 1 ? n : (me + 1) * per);
            }
            pending.fetch_sub(1, std::memory_order_release);
        }
    }
    template <class F>
    void run(size_t n_, size_t nthr_, F f) {
        fn = f;
        n = n_;
        nthr = nthr_;
        pending.store(NW, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            gen.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
        f((size_t)0, n_ / nthr_);
        while (pending.load(std::memory_order_acquire) != 0) relax();
    }
};
template <class F>
void host_blocks_n(size_t n, size_t nthr, F fn) {
    if (nthr > n) nthr = n;
    if (nthr > 1) {
        static HostPool* pool = new HostPool();  // (thread-safe initialisation; leaked on purpose)
        std::unique_lock<std::mutex> lk(pool->use, std::try_to_lock);
        if (lk.owns_lock()) {
            pool->run(n, nthr, fn);
            return;
        }
    }
    fn((size_t)0, n);
}
template <class F>
void host_blocks(size_t n, bool allow_threads, F fn) {
    host_blocks_n(n, !allow_threads ? 1 : n >= 1024 ? 4 : n >= 512 ? 2 : 1, fn);
}

}  // namespace bp
