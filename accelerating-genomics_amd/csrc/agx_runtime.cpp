// Context, error text and stopwatch of the C-ABI (include/agx.h, "runtime" section).
#include "agx_internal.h"

static thread_local char g_err[512] = "";

extern "C" void agx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int FanOut::begin()
{
    // AGX_FANOUT=0: every launch on the context's stream (experiments)
    static const bool off = [] {
        const char *e = getenv("AGX_FANOUT");
        return e && e[0] == '0';
    }();
    if (off) n = 1;
    if (n <= 1) return AGX_OK;
    if (!c->fork) {
        AGX_HIP(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
        for (int k = 0; k < kAuxStreams; ++k) {
            AGX_HIP(hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking));
            AGX_HIP(hipEventCreateWithFlags(&c->join[k], hipEventDisableTiming));
        }
    }
    AGX_HIP(hipEventRecord(c->fork, c->stream));
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

extern "C" {

const char *agx_version(void) { return "agx 0.1 (gfx950)"; }
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
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        agx_set_error("context setup on device %d -> %s", device, hipGetErrorString(e));
        agx_ctx_destroy(c);
        return AGX_E_HIP;
    }
    c->own_stream = true;
    *out = c;
    return AGX_OK;
}

void agx_ctx_destroy(agx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (int k = 0; k < kAuxStreams; ++k) {
        if (c->join[k]) (void)hipEventDestroy(c->join[k]);
        if (c->aux[k]) (void)hipStreamDestroy(c->aux[k]);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

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

} // extern "C"
