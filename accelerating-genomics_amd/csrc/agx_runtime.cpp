// Context, error text, stopwatch, buffer pools and the host thread pool of the C-ABI
// (include/agx.h, "runtime" section).
#include <immintrin.h>
#include <sched.h>
#include <condition_variable>
#include <deque>
#include <thread>

#include "agx_internal.h"
#include "agx_parallel.h"

static thread_local char g_err[512] = "";

extern "C" void agx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------ host thread pool

namespace {

struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    std::atomic<int> pending{0}; // jobs.size(), readable without the lock
    std::vector<std::thread> workers;
    bool stop = false;
    int parts = 1;

    static unsigned cores()
    {
        // the cores this process may run on (a GPU box gives one rank its share), not the machine's count
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
        // a container's CPU quota (cgroup v2 cpu.max "quota period"): the GPU boxes give a lease 16 cores per GPU of a
        // 256-core host this way, with every core in the affinity mask
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            long period = 0;
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long quota = atol(q);
                if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, (quota + period / 2) / period));
            }
            fclose(f);
        }
        return n;
    }
    Pool()
    {
        // One planner splits its passes into at most 16 parts (counting sorts and the final log10 loop stop scaling
        // about there), never more than an eighth of a large machine so that 8 ranks do not oversubscribe it;
        // AGX_HOST_THREADS (tuning build) overrides.  The WORKERS behind the parts grow with the devices a process
        // drives (reserve(): 16 per device, at most the cores), so that the planners of 8 device threads do not
        // queue behind 15 workers.
        const char *e = agx_tune("AGX_HOST_THREADS");
        int n = e ? atoi(e) : 0;
        const unsigned hw = cores();
        if (n <= 0) n = (int)std::min(16u, std::max(hw >= 32 ? hw / 8 : hw, 1u));
        parts = n;
        grow(n - 1);
    }
    void grow(int want_workers)
    {
        std::lock_guard<std::mutex> l(grow_mu);
        try {
            while ((int)workers.size() < want_workers) workers.emplace_back([this] { loop(); });
        } catch (...) {
            if (workers.empty()) parts = 1; // fewer threads than wanted is fine; none at all: everything runs on the caller
        }
    }
    void reserve(int n_devices)
    {
        if (agx_tune("AGX_HOST_THREADS")) return;
        const int64_t want = std::min<int64_t>((int64_t)cores() - 1, (int64_t)parts * std::max(1, n_devices) - 1);
        if (want > (int64_t)workers_hint.load(std::memory_order_relaxed)) {
            grow((int)want);
            workers_hint.store((int)want, std::memory_order_relaxed);
        }
    }
    std::mutex grow_mu;
    std::atomic<int> workers_hint{0};
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
    void loop()
    {
        // A planner is a dozen short parallel regions back to back: a worker that has just finished a job
        // polls for the next one for a few tens of microseconds before it sleeps on the condition variable
        // (waking a sleeper costs about as much as a region of a 65 536-pair batch).
        int spin = 0;
        for (;;) {
            std::function<void()> job;
            if (pending.load(std::memory_order_acquire) > 0 || spin == 0) {
                std::unique_lock<std::mutex> l(mu);
                if (spin == 0) cv.wait(l, [this] { return stop || !jobs.empty(); });
                if (jobs.empty()) {
                    if (stop) return;
                } else {
                    job = std::move(jobs.front());
                    jobs.pop_front();
                    pending.fetch_sub(1, std::memory_order_release);
                }
            }
            if (job) {
                job();
                spin = 4000;
            } else if (spin > 0) {
                --spin;
                __builtin_ia32_pause();
            }
        }
    }
};

Pool &pool()
{
    static Pool p;
    return p;
}

// The text readers' own workers (round 3).  A command line parses chunk k + 1 while chunk k is scored: with one queue the
// scorer's short parts -- a planner pass, a 1 MB slice of a staging copy -- waited behind the parser's long ones (9 MB of
// pread and scanning each), and scoring a 143 MB chunk took 20 ms inside the program where the same call alone takes 4.
Pool &reader_pool()
{
    static Pool p;
    return p;
}

void run_on(Pool &p, int parts, const std::function<void(int)> &task);

} // namespace

int agx_host_threads() { return pool().parts; }
void agx_pool_reserve(int n_devices) { pool().reserve(n_devices); }

// the pool for the C readers (agx_text.c)
extern "C" int agx_host_threads_c(void) { return agx_host_threads(); }
extern "C" void agx_pool_run_c(int parts, void (*task)(int, void *), void *arg)
{
    run_on(reader_pool(), parts, [&](int t) { task(t, arg); });
}

void agx_pool_run(int parts, const std::function<void(int)> &task) { run_on(pool(), parts, task); }

namespace {
void run_on(Pool &p, int parts, const std::function<void(int)> &task)
{
    if (parts <= 1) {
        task(0);
        return;
    }
    struct Sync {
        std::mutex mu;
        std::condition_variable cv;
        int left;
        std::exception_ptr err;
    } sync;
    sync.left = parts - 1;
    {
        // The jobs hold references to this frame (sync, task): they are built aside, where an allocation may fail with
        // nothing queued yet, and enter the queue in ONE insertion at its end -- which either happens entirely or not at
        // all (std::function moves do not throw) -- so no worker can hold a reference to a frame that is unwinding.
        std::vector<std::function<void()>> mine_jobs;
        mine_jobs.reserve((size_t)parts - 1);
        for (int t = 1; t < parts; ++t)
            mine_jobs.emplace_back([&sync, &task, t] {
                std::exception_ptr e;
                try {
                    task(t);
                } catch (...) {
                    e = std::current_exception();
                }
                std::lock_guard<std::mutex> g(sync.mu);
                if (e && !sync.err) sync.err = e;
                if (--sync.left == 0) sync.cv.notify_one();
            });
        std::lock_guard<std::mutex> l(p.mu);
        p.jobs.insert(p.jobs.end(), std::make_move_iterator(mine_jobs.begin()), std::make_move_iterator(mine_jobs.end()));
        p.pending.fetch_add(parts - 1, std::memory_order_release);
    }
    p.cv.notify_all();
    std::exception_ptr mine;
    try {
        task(0);
    } catch (...) {
        mine = std::current_exception();
    }
    // help with queued work instead of sleeping: several callers may share the pool, and a pool with
    // fewer workers than parts must still drain
    for (;;) {
        std::function<void()> job;
        {
            std::lock_guard<std::mutex> l(p.mu);
            if (p.jobs.empty()) break;
            job = std::move(p.jobs.front());
            p.jobs.pop_front();
            p.pending.fetch_sub(1, std::memory_order_release);
        }
        job();
    }
    std::unique_lock<std::mutex> g(sync.mu);
    sync.cv.wait(g, [&sync] { return sync.left == 0; });
    if (mine) std::rethrow_exception(mine);
    if (sync.err) std::rethrow_exception(sync.err);
}
} // namespace

// ------------------------------------------------------------------ staging copies

// memcpy for staging a pageable source into page-locked memory that only the DMA engine will read: streaming (non-temporal)
// stores, so that the destination lines are not first read into the cache (a third of the memory traffic of a plain copy of
// cold data) and the ring of staging blocks does not push the planner's arrays out of it.
__attribute__((target("avx2"))) static void stream_copy_avx2(uint8_t *dst, const uint8_t *src, size_t n)
{
    size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
    if (head > n) head = n;
    memcpy(dst, src, head);
    dst += head, src += head, n -= head;
    const size_t blocks = n / 128;
    for (size_t k = 0; k < blocks; ++k) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src)), b = _mm256_loadu_si256((const __m256i *)(src + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + 64)), d = _mm256_loadu_si256((const __m256i *)(src + 96));
        _mm256_stream_si256((__m256i *)(dst), a);
        _mm256_stream_si256((__m256i *)(dst + 32), b);
        _mm256_stream_si256((__m256i *)(dst + 64), c);
        _mm256_stream_si256((__m256i *)(dst + 96), d);
        src += 128, dst += 128;
    }
    _mm_sfence();
    memcpy(dst, src, n - blocks * 128);
}

void agx_stream_copy(void *dst, const void *src, size_t n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) stream_copy_avx2((uint8_t *)dst, (const uint8_t *)src, n);
    else memcpy(dst, src, n);
}

// ------------------------------------------------------------------ pooled buffers

namespace {

constexpr size_t kPoolMaxBlocks = 512; // (a one-shot call in pieces keeps 8-16 batches alive: 48 blocks overflowed into hipFree, which waits for the running fill)
constexpr size_t kDevPoolMaxBytes = (size_t)24 << 30; // of 288 GB HBM
constexpr size_t kPinPoolMaxBytes = (size_t)4 << 30;

size_t round_block(size_t n)
{
    // small blocks to 4 KB; from 1 MB on to an eighth of the size's power of two (at least 1 MB), so that requests a
    // few per cent apart -- the pieces of a one-shot call -- share a size class: with 1 MB steps a later, slightly
    // larger piece found every block of its class taken by smaller ones and fell through to hipMalloc, which waits
    // for the running fills (9 ms inside a 13 ms call)
    size_t g = 4096;
    if (n >= ((size_t)1 << 20)) {
        size_t p2 = (size_t)1 << 20;
        while (p2 * 2 <= n) p2 *= 2;
        g = std::max<size_t>((size_t)1 << 20, p2 / 8);
    }
    return (n + g - 1) / g * g;
}

// best fit among the free blocks: at least n bytes, at most twice that (+1 MiB)
bool pool_take(std::vector<PoolBlock> &fl, size_t &cached, size_t n, PoolBlock *out)
{
    int best = -1;
    for (int k = 0; k < (int)fl.size(); ++k)
        if (fl[k].bytes >= n && fl[k].bytes <= 2 * n + ((size_t)1 << 20) && (best < 0 || fl[k].bytes < fl[best].bytes)) best = k;
    if (best < 0) return false;
    *out = fl[best];
    cached -= fl[best].bytes;
    fl.erase(fl.begin() + best);
    return true;
}

} // namespace

int DevBuf::alloc(agx_ctx *c, size_t n)
{
    release();
    if (!c) {
        agx_set_error("device allocation without a context");
        return AGX_E_ARG;
    }
    if (n == 0) n = 16;
    PoolBlock b{nullptr, 0};
    bool hit;
    {
        std::lock_guard<std::mutex> l(c->pool_mu);
        hit = pool_take(c->free_dev, c->cached_dev, n, &b);
    }
    if (!hit) {
        b.bytes = round_block(n);
        if (agx_tune("AGX_TRACE_POOL")) fprintf(stderr, "[pool] device miss: %zu bytes asked, hipMalloc(%zu)\n", n, b.bytes);
        hipError_t e = hipMalloc(&b.p, b.bytes);
        if (e != hipSuccess) { // give the cache back and try once more
            std::vector<PoolBlock> drop;
            {
                std::lock_guard<std::mutex> l(c->pool_mu);
                drop.swap(c->free_dev);
                c->cached_dev = 0;
            }
            for (auto &d : drop) (void)hipFree(d.p);
            e = hipMalloc(&b.p, b.bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            agx_set_error("hipMalloc(%zu) -> %s", b.bytes, hipGetErrorString(e));
            return AGX_E_NOMEM;
        }
    }
    ctx = c;
    p = b.p;
    block = b.bytes;
    bytes = n;
    return AGX_OK;
}

void DevBuf::release()
{
    if (p && ctx) {
        void *drop = nullptr;
        {
            std::lock_guard<std::mutex> l(ctx->pool_mu);
            if (ctx->free_dev.size() < kPoolMaxBlocks && ctx->cached_dev + block <= kDevPoolMaxBytes) {
                ctx->free_dev.push_back(PoolBlock{p, block});
                ctx->cached_dev += block;
            } else
                drop = p;
        }
        if (drop) (void)hipFree(drop);
    }
    p = nullptr;
    bytes = block = 0;
    ctx = nullptr;
}

int PinBuf::alloc(agx_ctx *c, size_t n)
{
    release();
    if (!c) {
        agx_set_error("pinned allocation without a context");
        return AGX_E_ARG;
    }
    if (n == 0) n = 16;
    PoolBlock b{nullptr, 0};
    bool hit;
    {
        std::lock_guard<std::mutex> l(c->pool_mu);
        hit = pool_take(c->free_pin, c->cached_pin, n, &b);
    }
    if (!hit) {
        b.bytes = round_block(n);
        if (agx_tune("AGX_TRACE_POOL")) fprintf(stderr, "[pool] pinned miss: %zu bytes asked, hipHostMalloc(%zu)\n", n, b.bytes);
        hipError_t e = hipHostMalloc(&b.p, b.bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            agx_set_error("hipHostMalloc(%zu) -> %s", b.bytes, hipGetErrorString(e));
            return AGX_E_NOMEM;
        }
    }
    ctx = c;
    p = b.p;
    block = b.bytes;
    bytes = n;
    return AGX_OK;
}

void PinBuf::release()
{
    if (p && ctx) {
        void *drop = nullptr;
        {
            std::lock_guard<std::mutex> l(ctx->pool_mu);
            if (ctx->free_pin.size() < kPoolMaxBlocks && ctx->cached_pin + block <= kPinPoolMaxBytes) {
                ctx->free_pin.push_back(PoolBlock{p, block});
                ctx->cached_pin += block;
            } else
                drop = p;
        }
        if (drop) (void)hipHostFree(drop);
    }
    p = nullptr;
    bytes = block = 0;
    ctx = nullptr;
}

// blocks handed out by agx_host_alloc (a handful per process)
static std::mutex g_pinned_mu;
static std::vector<PoolBlock> g_pinned;

bool agx_is_pinned_host(const void *p, size_t bytes)
{
    if (!p) return false;
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> l(g_pinned_mu);
    for (const PoolBlock &b : g_pinned)
        if (a >= (uintptr_t)b.p && a + bytes <= (uintptr_t)b.p + b.bytes) return true;
    return false;
}

// ------------------------------------------------------------------ launch fan-out

int FanOut::begin()
{
    // AGX_FANOUT=0 (tuning build): every launch on the context's stream
    static const bool off = [] {
        const char *e = agx_tune("AGX_FANOUT");
        return e && e[0] == '0';
    }();
    if (off) n = 1;
    if (n <= 1) return AGX_OK;
    const int rc = agx_ctx_prepare_fanout(c); // normally done at batch creation already
    if (rc) return rc;
    AGX_HIP(hipEventRecord(c->fork, c->stream));
    return AGX_OK;
}

// The side streams and their events: made when a batch with several launches is CREATED, so that the first
// launch does not pay for them (hipvers' timed window is launch -> scores of a fresh process).
int agx_ctx_prepare_fanout(agx_ctx *c)
{
    if (c->fork) return AGX_OK;
    AGX_HIP(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    for (int k = 0; k < kAuxStreams; ++k) {
        AGX_HIP(hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking));
        AGX_HIP(hipEventCreateWithFlags(&c->join[k], hipEventDisableTiming));
    }
    return AGX_OK;
}

int agx_ctx_prepare_plan(agx_ctx *c)
{
    if (c->plan) return AGX_OK;
    AGX_HIP(hipStreamCreateWithPriority(&c->plan, hipStreamNonBlocking, c->prio_hi));
    AGX_HIP(hipEventCreateWithFlags(&c->plan_done, hipEventDisableTiming));
    return AGX_OK;
}

hipStream_t FanOut::stream(int k)
{
    if (n <= 1) return c->stream;
    const int lane = k % (kAuxStreams + 1);
    if (lane == 0) return c->stream;
    if (!used[lane - 1]) {
        (void)hipStreamWaitEvent(c->aux[lane - 1], c->fork, 0);
        used[lane - 1] = true;
    }
    return c->aux[lane - 1];
}

int FanOut::end()
{
    for (int k = 0; k < kAuxStreams; ++k)
        if (used[k]) {
            AGX_HIP(hipEventRecord(c->join[k], c->aux[k]));
            AGX_HIP(hipStreamWaitEvent(c->stream, c->join[k], 0));
        }
    return AGX_OK;
}

// ------------------------------------------------------------------ contexts

void agx_ctx_retain(agx_ctx *c)
{
    if (c) c->refs.fetch_add(1, std::memory_order_relaxed);
}

void agx_ctx_release(agx_ctx *c)
{
    if (!c) return;
    if (c->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &b : c->free_dev) (void)hipFree(b.p);
    for (auto &b : c->free_pin) (void)hipHostFree(b.p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (int k = 0; k < kAuxStreams; ++k) {
        if (c->join[k]) (void)hipEventDestroy(c->join[k]);
        if (c->aux[k]) (void)hipStreamDestroy(c->aux[k]);
    }
    if (c->copy) (void)hipStreamDestroy(c->copy);
    if (c->plan) (void)hipStreamDestroy(c->plan);
    if (c->plan_done) (void)hipEventDestroy(c->plan_done);
    if (c->sw_seg_first) (void)hipFree(c->sw_seg_first);
    if (c->sw_segs) (void)hipFree(c->sw_segs);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int agx_shared_ctx(int device, int slot, agx_ctx **out, std::mutex **busy)
{
    struct Entry {
        int device, slot;
        agx_ctx *ctx;
        std::mutex *busy;
    };
    static std::mutex mu;
    static std::vector<Entry> table; // never freed: process lifetime
    std::lock_guard<std::mutex> l(mu);
    for (const Entry &e : table)
        if (e.device == device && e.slot == slot) {
            *out = e.ctx;
            if (busy) *busy = e.busy;
            return AGX_OK;
        }
    agx_ctx *c = nullptr;
    const int rc = agx_ctx_create(device, &c);
    if (rc) return rc;
    table.push_back(Entry{device, slot, c, new std::mutex()});
    *out = c;
    if (busy) *busy = table.back().busy;
    return AGX_OK;
}

extern "C" {

const char *agx_version(void)
{
#ifdef AGX_TUNING
    return "agx 0.2 (gfx950, tuning build)";
#else
    return "agx 0.2 (gfx950)";
#endif
}
const char *agx_last_error(void) { return g_err; }

int agx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int agx_device_name(int device, char *buf, size_t buf_len)
{
    if (!buf || buf_len == 0) {
        agx_set_error("agx_device_name: null buffer");
        return AGX_E_ARG;
    }
    buf[0] = 0;
    if (device < 0 || device >= agx_device_count()) {
        agx_set_error("device %d out of range", device);
        return AGX_E_NODEVICE;
    }
    hipDeviceProp_t prop;
    AGX_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buf_len, "%s", prop.name);
    return AGX_OK;
}

int agx_ctx_create(int device, agx_ctx **out)
{
    if (!out) {
        agx_set_error("agx_ctx_create: out is NULL");
        return AGX_E_ARG;
    }
    *out = nullptr;
    AGX_GUARD_BEGIN
    int n = agx_device_count();
    if (n <= 0) {
        agx_set_error("no HIP device is visible (this library has no CPU fallback)");
        return AGX_E_NODEVICE;
    }
    if (device < 0 || device >= n) {
        agx_set_error("device %d out of range [0,%d)", device, n);
        return AGX_E_NODEVICE;
    }
    AGX_HIP(hipSetDevice(device));
    agx_ctx *c = new agx_ctx();
    c->device = device;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) c->n_cu = cus;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) c->own_stream = true;
    // The copy stream (uploads, the pack kernel) and the planning stream get the highest priority the device offers:
    // their short kernels run beside the previous batch's fill, whose thousands of queued waves would otherwise keep
    // them waiting for slots (a 0.16 ms pack kernel took 0.5 ms beside a fill).
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (agx_tune("AGX_NO_STREAM_PRIO")) prio_hi = 0; // A/B: every stream at the default priority
    c->prio_hi = prio_hi;
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->copy, hipStreamNonBlocking, prio_hi);
    // The planning stream is made here too, right behind the other two: the runtime multiplexes streams onto a few
    // hardware queues in the order they are created, and a context's three streams must not share one -- made lazily,
    // after another context's streams, `plan` landed on the queue of `copy` and every pack kernel waited for the next
    // piece's upload (profiles/r03n_timeline.log).
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->plan, hipStreamNonBlocking, prio_hi);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->plan_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        agx_set_error("context setup on device %d -> %s", device, hipGetErrorString(e));
        agx_ctx_release(c);
        return AGX_E_HIP;
    }
    // The first sizeable copy in each direction brings up a DMA engine queue: 7.5-9 ms for the first 100 KB
    // device-to-host copy of a process (tools/first_call_costs.py; a 256-byte copy takes another path and warms
    // nothing).  Pay that here, beside the rest of the HIP start-up, not in the first agx_*_batch_scores /
    // results: the hipvers window of a fresh process is launch -> scores.
    {
        constexpr size_t kWarm = (size_t)1 << 20;
        DevBuf d;
        PinBuf h;
        if (d.alloc(c, kWarm) == AGX_OK && h.alloc(c, kWarm) == AGX_OK) {
            memset(h.p, 0, kWarm);
            (void)hipMemcpyAsync(d.p, h.p, kWarm, hipMemcpyHostToDevice, c->copy);
            (void)hipStreamSynchronize(c->copy);
            (void)hipMemcpyAsync(h.p, d.p, kWarm, hipMemcpyDeviceToHost, c->stream);
            (void)hipStreamSynchronize(c->stream);
            (void)hipMemcpyAsync(d.p, h.p, kWarm, hipMemcpyHostToDevice, c->stream);
            (void)hipMemcpyAsync(h.p, d.p, kWarm, hipMemcpyDeviceToHost, c->copy);
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->copy);
        }
        d.release();
        h.release();
        (void)hipGetLastError();
    }
    *out = c;
    return AGX_OK;
    AGX_GUARD_END("agx_ctx_create")
}

void agx_ctx_destroy(agx_ctx *c) { agx_ctx_release(c); }

int agx_ctx_device(const agx_ctx *c) { return c ? c->device : -1; }
void *agx_ctx_stream(const agx_ctx *c) { return c ? (void *)c->stream : nullptr; }

int agx_ctx_set_stream(agx_ctx *c, void *s)
{
    int rc = agx_bind(c);
    if (rc) return rc;
    if (c->own_stream && c->stream) {
        AGX_HIP(hipStreamSynchronize(c->stream));
        (void)hipStreamDestroy(c->stream);
    }
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return AGX_OK;
}

int agx_ctx_set_option(agx_ctx *c, int key, int64_t value)
{
    if (!c) {
        agx_set_error("agx_ctx_set_option: null context");
        return AGX_E_ARG;
    }
    switch (key) {
    case AGX_OPT_SW_KERNEL:
        if (value < AGX_SW_KERNEL_AUTO || value > AGX_SW_KERNEL_PACKED_BIASED) break;
        c->opt_sw_kernel = (int)value;
        return AGX_OK;
    case AGX_OPT_SW_PLANNER:
        if (value < AGX_SW_PLANNER_AUTO || value > AGX_SW_PLANNER_DEVICE) break;
        c->opt_sw_planner = (int)value;
        return AGX_OK;
    case AGX_OPT_PHMM_TRAINS:
        if (value < AGX_PHMM_TRAINS_AUTO || value > AGX_PHMM_TRAINS_ON) break;
        c->opt_phmm_trains = (int)value;
        return AGX_OK;
    default: break;
    }
    agx_set_error("agx_ctx_set_option: unknown key %d or bad value %lld", key, (long long)value);
    return AGX_E_ARG;
}

int agx_ctx_sync(agx_ctx *c)
{
    int rc = agx_bind(c);
    if (rc) return rc;
    AGX_HIP(hipStreamSynchronize(c->stream));
    return AGX_OK;
}

int agx_ctx_timer_start(agx_ctx *c)
{
    int rc = agx_bind(c);
    if (rc) return rc;
    AGX_HIP(hipEventRecord(c->ev0, c->stream));
    return AGX_OK;
}

int agx_ctx_timer_stop(agx_ctx *c, float *ms)
{
    int rc = agx_bind(c);
    if (rc) return rc;
    AGX_HIP(hipEventRecord(c->ev1, c->stream));
    AGX_HIP(hipEventSynchronize(c->ev1));
    float t = 0;
    AGX_HIP(hipEventElapsedTime(&t, c->ev0, c->ev1));
    if (ms) *ms = t;
    return AGX_OK;
}

int agx_ctx_timer_mark(agx_ctx *c)
{
    int rc = agx_bind(c);
    if (rc) return rc;
    AGX_HIP(hipEventRecord(c->ev1, c->stream));
    return AGX_OK;
}

int agx_ctx_timer_elapsed(agx_ctx *c, float *ms)
{
    int rc = agx_bind(c);
    if (rc) return rc;
    AGX_HIP(hipEventSynchronize(c->ev1));
    float t = 0;
    AGX_HIP(hipEventElapsedTime(&t, c->ev0, c->ev1));
    if (ms) *ms = t;
    return AGX_OK;
}

int agx_warmup_devices(const int *devices, int n_devices)
{
    AGX_GUARD_BEGIN
    const int avail = agx_device_count();
    if (avail <= 0) {
        agx_set_error("no HIP device is visible (this library has no CPU fallback)");
        return AGX_E_NODEVICE;
    }
    if (!devices && (n_devices <= 0 || n_devices > avail)) n_devices = avail;
    std::vector<int> slot_of((size_t)avail, 0);
    for (int k = 0; k < n_devices; ++k) {
        const int dev = devices ? devices[k] : k;
        if (dev < 0 || dev >= avail) {
            agx_set_error("agx_warmup_devices: device %d out of range [0,%d)", dev, avail);
            return AGX_E_NODEVICE;
        }
        agx_ctx *c = nullptr;
        const int rc = agx_shared_ctx(dev, slot_of[(size_t)dev]++, &c); // shards sharing a device have contexts of their own
        if (rc) return rc;
    }
    return AGX_OK;
    AGX_GUARD_END("agx_warmup_devices")
}

void *agx_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        agx_set_error("agx_host_alloc(%zu): no pinned memory (is a HIP device visible?)", bytes);
        return nullptr;
    }
    try {
        std::lock_guard<std::mutex> l(g_pinned_mu);
        g_pinned.push_back(PoolBlock{p, bytes});
    } catch (...) {
        (void)hipHostFree(p);
        agx_set_error("agx_host_alloc: out of host memory");
        return nullptr;
    }
    return p;
}

void agx_host_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> l(g_pinned_mu);
        for (size_t k = 0; k < g_pinned.size(); ++k)
            if (g_pinned[k].p == p) {
                g_pinned[k] = g_pinned.back();
                g_pinned.pop_back();
                break;
            }
    }
    (void)hipHostFree(p);
}

} // extern "C"
