// Host-only ThreadSanitizer driver of the SSP_HOST staging pool (speech_signal_processing_amd/csrc/staging.hpp, StagePoolT): several threads
// take and give back slots of ONE pool — what the pool's mutex is there for (a ctx is not thread-safe, but two host-pointer calls that
// do overlap on one ctx must never share a staging buffer) — and of two pools side by side (two contexts, two threads: the supported
// use).  A slot handed to two holders at once, a torn `busy` flag or an unlocked access is a data race TSan reports and a failed
// ownership check here.  Built and run by tests/test_host.py::test_staging_pool_under_thread_sanitizer (g++ -fsanitize=thread):
// no HIP runtime, the buffer type is malloc-backed.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/ssp.h"

namespace ssp {
struct FakeBuf {
    void* p = nullptr;
    size_t bytes = 0;
    std::atomic<int> owner{-1};   // test-side: which thread holds the slot this buffer backs
    ~FakeBuf() { free(p); }
    int alloc(size_t n) {
        free(p);
        p = malloc(n ? n : 16);
        bytes = p ? n : 0;
        return p ? SSP_OK : SSP_ERR_NOMEM;
    }
};
}  // namespace ssp
#define SSP_STAGING_NO_HIP 1
#define SSP_STAGING_PART 1
#include "../../speech_signal_processing_amd/csrc/staging.hpp"

using Pool = ssp::StagePoolT<ssp::FakeBuf>;
static std::atomic<long> g_errors{0};

static void worker(Pool* pool, int tid, int iters, unsigned seed) {
    for (int i = 0; i < iters; ++i) {
        seed = seed * 1664525u + 1013904223u;
        const size_t n = 1 + (seed >> 8) % (i % 7 == 0 ? (size_t)(96u << 20) : (size_t)(1u << 16));   // (now and then above KEEP_MAX: no slot)
        int rc = 0;
        const int s = pool->take(n, &rc);
        if (s < 0) continue;   // all slots taken, too large, or an allocation failed: the caller gets a buffer of its own
        ssp::FakeBuf& b = pool->slot[s];
        int expect = -1;
        if (!b.owner.compare_exchange_strong(expect, tid)) g_errors++;          // somebody else holds it
        if (b.bytes < n) g_errors++;
        memset(b.p, tid, n < 256 ? n : 256);                                     // (a second holder would race on these bytes: TSan's to find)
        if (((volatile unsigned char*)b.p)[0] != (unsigned char)tid) g_errors++;
        b.owner.store(-1);
        pool->give_back(s);
    }
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 8, iters = argc > 2 ? atoi(argv[2]) : 20000;
    {   // one pool shared by every thread
        Pool pool;
        std::vector<std::thread> t;
        for (int i = 0; i < threads; ++i) t.emplace_back(worker, &pool, i, iters, 17u * (unsigned)i + 1u);
        for (auto& x : t) x.join();
    }
    {   // two pools (two contexts), one thread each, and a third thread hopping between them
        Pool a, b;
        std::thread t0(worker, &a, 0, iters, 3u), t1(worker, &b, 1, iters, 5u);
        std::thread t2([&] { for (int k = 0; k < 50; ++k) { worker(&a, 2, iters / 100, 7u + k); worker(&b, 2, iters / 100, 9u + k); } });
        t0.join(); t1.join(); t2.join();
    }
    printf("stagepool_threads: %d threads x %d takes, ownership errors %ld\n", threads, iters, g_errors.load());
    return g_errors.load() == 0 ? 0 : 1;
}
