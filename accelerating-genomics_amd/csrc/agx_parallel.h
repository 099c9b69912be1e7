// Host-side helper: split [0, n) over a persistent pool of worker threads shared by every context of
// the process (the planners are short bursts of memory-bound work; spawning std::threads per call cost
// more than the work itself on small batches, and one pool bounds the host threads no matter how many
// devices plan at once).  fn(begin, end, part_index), part_index < agx_host_threads().
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>

// Number of parts a parallel region may be split into (pool workers + the calling thread).
int agx_host_threads();
// Runs task(part) for part = 0 .. parts-1; part 0 on the calling thread.  Reentrant from several
// caller threads (device threads of the multi-device entry points share the pool).
void agx_pool_run(int parts, const std::function<void(int)> &task);

template <typename F>
void agx_parallel_for(int64_t n, int64_t min_per_thread, F fn)
{
    const int nt = (int)std::min<int64_t>(agx_host_threads(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
    if (nt <= 1) {
        fn((int64_t)0, n, 0);
        return;
    }
    const int64_t chunk = (n + nt - 1) / nt;
    agx_pool_run(nt, [&](int t) {
        const int64_t b = std::min(n, t * chunk), e = std::min(n, b + chunk);
        if (b < e) fn(b, e, t);
    });
}
