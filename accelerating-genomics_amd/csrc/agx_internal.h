// Internal to libagx.so: context object, error plumbing, device buffer helper.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

constexpr int kAuxStreams = 3;

struct agx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cu = 0;
    // Side streams for batches that need several kernel launches (one per lane-tiling class):
    // the launches are independent, so they are spread over the main stream and these, forked
    // and joined with events -- the tail of one class overlaps the head of the next.
    hipStream_t aux[kAuxStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[kAuxStreams] = {nullptr, nullptr, nullptr};
};

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

// RAII-less device buffer (freed explicitly so error paths stay simple C-style).
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n)
    {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            agx_set_error("hipMalloc(%zu) -> %s", n, hipGetErrorString(e));
            return AGX_E_NOMEM;
        }
        bytes = n;
        return AGX_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
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
