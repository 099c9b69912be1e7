// Host-side helper: split [0, n) over a persistent pool of worker threads shared by every context of
// the process (the planners are short bursts of memory-bound work; spawning std::threads per call cost
// more than the work itself on small batches, and one pool bounds the host threads no matter how many
// devices plan at once).  fn(begin, end, part_index), part_index < agx_host_threads().
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <thread>
#include <vector>

// Number of parts a parallel region may be split into (pool workers + the calling thread).
int agx_host_threads();
// Runs task(part) for part = 0 .. parts-1; part 0 on the calling thread.  Reentrant from several
// caller threads (device threads of the multi-device entry points share the pool).
void agx_pool_run(int parts, const std::function<void(int)> &task);

// A process that drives n devices at once lets the pool grow to 16 workers per device (at most the cores it may run on).
void agx_pool_reserve(int n_devices);

// One host thread per shard of a multi-device call, shard 0 on the caller.  shard(k) must not throw (the entry points
// catch inside it).  When a thread cannot be started the shards without one run on the caller, after its own: the call
// gets slower, it does not terminate with joinable threads behind it.
template <typename F>
void agx_fan_out(int n, F shard)
{
    if (n <= 1) {
        if (n == 1) shard(0);
        return;
    }
    agx_pool_reserve(n);
    std::vector<std::thread> th;
    int started = 1;
    try {
        th.reserve((size_t)n - 1);
        for (int k = 1; k < n; ++k) {
            th.emplace_back(shard, k);
            started = k + 1;
        }
    } catch (...) {
    }
    shard(0);
    for (int k = started; k < n; ++k) shard(k);
    for (auto &t : th) t.join();
}

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

// Stable counting sort of src into dst by key(e) in [0, n_keys), threaded: every part counts its
// contiguous chunk, the offsets are laid out key-major / part-minor, every part scatters its chunk in order.
template <typename T, typename KeyFn>
void counting_sort(const std::vector<T> &src, std::vector<T> &dst, size_t n_keys, KeyFn key)
{
    const size_t n = src.size();
    dst.resize(n);
    const int parts = (int)std::min<int64_t>(agx_host_threads(), std::max<int64_t>(1, (int64_t)n / 32768));
    std::vector<std::vector<uint32_t>> cnt((size_t)parts);
    const size_t chunk = (n + parts - 1) / (size_t)parts;
    agx_pool_run(parts, [&](int t) {
        std::vector<uint32_t> &c = cnt[(size_t)t];
        c.assign(n_keys, 0);
        const size_t b = std::min(n, (size_t)t * chunk), e = std::min(n, b + chunk);
        for (size_t i = b; i < e; ++i) ++c[key(src[i])];
    });
    uint32_t run = 0;
    for (size_t k = 0; k < n_keys; ++k)
        for (int t = 0; t < parts; ++t) {
            const uint32_t c = cnt[(size_t)t][k];
            cnt[(size_t)t][k] = run;
            run += c;
        }
    agx_pool_run(parts, [&](int t) {
        std::vector<uint32_t> &c = cnt[(size_t)t];
        const size_t b = std::min(n, (size_t)t * chunk), e = std::min(n, b + chunk);
        for (size_t i = b; i < e; ++i) dst[c[key(src[i])]++] = src[i];
    });
}

