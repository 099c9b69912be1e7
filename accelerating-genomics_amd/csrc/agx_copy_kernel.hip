// Results to page-locked host memory by a kernel instead of the copy engine: consecutive stores over PCIe.
// Behind a fill on the same stream this ends 6-7 us earlier than a hipMemcpyAsync of the same 256-512 KB (config 2:
// launch -> scores 0.1948 -> 0.1881 ms, tools/window_sw.py) -- the copy engine's start-up is the difference.
#include "agx_internal.h"

namespace {

template <typename W>
__global__ void __launch_bounds__(256) copy_out(const W *__restrict__ src, W *__restrict__ dst, size_t n_words, size_t bytes)
{
    const size_t k = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (k < n_words) dst[k] = src[k];
    if (k == 0) // the last bytes % sizeof(W)
        for (size_t i = n_words * sizeof(W); i < bytes; ++i)
            reinterpret_cast<unsigned char *>(dst)[i] = reinterpret_cast<const unsigned char *>(src)[i];
}

template <typename W>
int launch(const void *src, void *dst, size_t bytes, hipStream_t s)
{
    const size_t n = bytes / sizeof(W);
    hipLaunchKernelGGL(copy_out<W>, dim3((unsigned)((std::max<size_t>(n, 1) + 255) / 256)), dim3(256), 0, s, (const W *)src, (W *)dst, n, bytes);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace

// src: device memory, dst: page-locked host memory; the widest word both pointers are aligned to is used
int agx_copy_out_launch(const void *src, void *dst, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return 0;
    const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst);
    if ((a & 15u) == 0) return launch<uint4>(src, dst, bytes, s);
    if ((a & 7u) == 0) return launch<uint2>(src, dst, bytes, s);
    if ((a & 3u) == 0) return launch<uint32_t>(src, dst, bytes, s);
    return launch<unsigned char>(src, dst, bytes, s);
}

void agx_copy_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&copy_out<uint4>));
}
