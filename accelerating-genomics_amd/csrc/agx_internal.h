// Internal to libagx.so: context object, error plumbing, pooled device / pinned buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/agx.h"

extern "C" void agx_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define AGX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            agx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return e_ == hipErrorOutOfMemory ? AGX_E_NOMEM : AGX_E_HIP;                        \
        }                                                                                      \
    } while (0)

// Every extern "C" entry point that can allocate runs its body through this: a C caller must never see
// a C++ exception (std::bad_alloc from a std::vector, std::system_error from a thread).
#define AGX_GUARD_BEGIN try {
#define AGX_GUARD_END(fn_name)                                     \
    }                                                              \
    catch (const std::bad_alloc &)                                 \
    {                                                              \
        agx_set_error("%s: out of host memory", fn_name);          \
        return AGX_E_NOMEM;                                        \
    }                                                              \
    catch (const std::exception &ex_)                              \
    {                                                              \
        agx_set_error("%s: %s", fn_name, ex_.what());              \
        return AGX_E_NOMEM;                                        \
    }

// Experiment and test knobs (AGX_SW_FORCE_C, AGX_*_TAIL_BETA, AGX_FANOUT, ...) exist only in the tuning
// build (libagx_tuning.so, -DAGX_TUNING; tools/ and the calibration scripts load that one).  The shipped
// libagx.so reads no environment variable: it behaves the same in every process.
static inline const char *agx_tune(const char *name)
{
#ifdef AGX_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

constexpr int kAuxStreams = 3;

struct PoolBlock {
    void *p;
    size_t bytes;
};

struct agx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cu = 0;
    // Uploads run on their own stream so that building batch k+1 overlaps the fill of batch k.
    hipStream_t copy = nullptr;
    // Side streams for batches that need several kernel launches (one per lane-tiling class):
    // the launches are independent, so they are spread over the main stream and these, forked
    // and joined with events -- the tail of one class overlaps the head of the next.
    hipStream_t aux[kAuxStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[kAuxStreams] = {nullptr, nullptr, nullptr};
    // Batches keep their context alive: agx_ctx_destroy() only drops the creator's reference, the
    // streams and the pools go when the last batch has been destroyed too (any destroy order is fine).
    std::atomic<int> refs{1};
    // Free device / pinned-host blocks kept for the next batch: hipMalloc / hipFree / hipHostMalloc cost
    // 0.1 .. several ms each and hipFree synchronises the device.
    std::mutex pool_mu;
    std::vector<PoolBlock> free_dev, free_pin;
    size_t cached_dev = 0, cached_pin = 0;
    // Device-side planning (agx_sw_plan_kernel.hip): its own stream, so that the planning kernels run beside the
    // upload of the sequences on `copy`, an event the pack kernel waits for, and the tiling table of the packed
    // biased fill on the device (made on first use, freed with the context).
    hipStream_t plan = nullptr;
    int prio_hi = 0; // highest stream priority of the device (copy and planning streams)
    hipEvent_t plan_done = nullptr;
    void *sw_seg_first = nullptr, *sw_segs = nullptr;
    // options (agx_ctx_set_option)
    int opt_sw_kernel = 0;
    int opt_sw_planner = 0;
    int opt_phmm_trains = 0;
};

int agx_ctx_prepare_fanout(agx_ctx *c); // side streams + events for batches of several launches
int agx_ctx_prepare_plan(agx_ctx *c);   // the planning stream and its event
void agx_ctx_retain(agx_ctx *c);
void agx_ctx_release(agx_ctx *c); // frees everything when the last reference goes

// Usage: FanOut f(ctx, n_launches); f.begin(); ... launch k on f.stream(k) ...; f.end();
// Everything is ordered after prior work on ctx->stream and before later work on it.
struct FanOut {
    agx_ctx *c;
    int n;
    bool used[kAuxStreams] = {false, false, false};
    FanOut(agx_ctx *ctx, int n_launches) : c(ctx), n(n_launches) {}
    int begin();
    hipStream_t stream(int k);
    int end();
};

// Device buffer drawn from its context's pool (freed explicitly so error paths stay simple C-style).
struct DevBuf {
    agx_ctx *ctx = nullptr;
    void *p = nullptr;
    size_t bytes = 0; // requested size; the block behind it may be larger
    size_t block = 0;
    int alloc(agx_ctx *c, size_t n);
    void release();
};

// Pinned host staging buffer drawn from the context's pool.
struct PinBuf {
    agx_ctx *ctx = nullptr;
    void *p = nullptr;
    size_t bytes = 0, block = 0;
    int alloc(agx_ctx *c, size_t n);
    void release();
};

static inline int agx_bind(const agx_ctx *c)
{
    if (!c) {
        agx_set_error("null context");
        return AGX_E_ARG;
    }
    AGX_HIP(hipSetDevice(c->device));
    return AGX_OK;
}

// memcpy with streaming stores, for staging copies into page-locked memory (agx_runtime.cpp)
void agx_stream_copy(void *dst, const void *src, size_t n);

// true when [p, p + bytes) lies in a block handed out by agx_host_alloc (page-locked: the device can DMA into or
// out of it directly).  A registry of our own rather than hipPointerGetAttributes: asked about an ordinary
// malloc'ed pointer, the runtime's first answer in a process took 7.5 ms (tools/first_call_costs.py) -- inside
// the launch -> scores window of a fresh process.  Memory pinned by other means is treated as pageable (staged).
bool agx_is_pinned_host(const void *p, size_t bytes);

// device -> page-locked host memory by a kernel on stream s (agx_copy_kernel.hip); both pointers 16-byte aligned
int agx_copy_out_launch(const void *src, void *dst, size_t bytes, hipStream_t s);
void agx_copy_preload();

// Process-wide contexts for the entry points that take device ordinals instead of a context
// (agx_*_devices, agx_*_multi, agx_pairHMM): created on first use, kept until the process ends.
// slot distinguishes several shards mapped onto the same device.
// *busy (may be NULL) receives the context's own mutex: a shard holds it while it uses the context, so that two
// host threads calling agx_*_devices at once take turns per (device, slot) instead of interleaving on its streams.
int agx_shared_ctx(int device, int slot, agx_ctx **out, std::mutex **busy = nullptr);
