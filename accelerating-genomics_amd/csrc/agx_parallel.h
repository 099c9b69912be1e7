// Host-side helper: split [0, n) over a few std::threads (the packers are memory-bound byte
// shuffling; 8 threads are plenty).  fn(begin, end, thread_index).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

inline int agx_host_threads()
{
    static const int v = [] {
        const char *e = getenv("AGX_HOST_THREADS");
        int n = e ? atoi(e) : 0;
        if (n <= 0) n = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        return n;
    }();
    return v;
}

template <typename F>
void agx_parallel_for(int64_t n, int64_t min_per_thread, F fn)
{
    int nt = (int)std::min<int64_t>(agx_host_threads(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
    if (nt <= 1) {
        fn((int64_t)0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    const int64_t chunk = (n + nt - 1) / nt;
    for (int t = 1; t < nt; ++t) {
        const int64_t b = std::min(n, t * chunk), e = std::min(n, b + chunk);
        th.emplace_back([=] { fn(b, e, t); });
    }
    fn((int64_t)0, std::min(n, chunk), 0);
    for (auto &t : th) t.join();
}
