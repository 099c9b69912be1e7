// The last line of pairHMM() -- log10(sum) - log10(C), antidiagsPairHMM.c:242 -- for the FLOAT modes, on the device.
//
// The double modes keep it on the host: their results are compared bit for bit with the reference's "%.17g"
// output, and only the host libm rounds like the reference's.  The float modes promise 1e-6 relative; a double
// log10 on the device is good to an ulp of double, and it takes 0.1 ms of host time per 65 536 pairs out of the
// launch -> results-on-host window (bench.py's pairhmm leg: 0.55 -> 0.45 ms per step).
// Pairs the rescue pass (or the striped kernel) recomputed in double arrive with their sum negated: they are
// scaled by DBL_MAX/16, the others by FLT_MAX/16.  sums[] itself is left alone (results may be fetched twice).
#include "agx_phmm.h"

namespace {

__global__ void __launch_bounds__(256) phmm_finish_f32(const double *__restrict__ sums, double *__restrict__ logs, uint32_t n,
                                                       double log_c64, double log_c32, unsigned long long *__restrict__ n_rescued,
                                                       unsigned long long *__restrict__ n_rescued_host)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    // the rescue counter goes to page-locked host memory from here: a D2H copy of its own for eight bytes cost
    // about as much as the copy of the results
    if (k == 0) {
        n_rescued_host[0] = n_rescued[0]; // pairs the rescue pass recomputed
        n_rescued_host[1] = n_rescued[1]; // pairs the packed fill found below the float range
        // ... and are reset here, by their only reader: the next launch of the batch then needs no memset in front of its
        // fill (round 3: a 4 us fill kernel of the runtime's plus its gap, inside every launch -> results window)
        n_rescued[0] = 0;
        n_rescued[1] = 0;
    }
    if (k >= n) return;
    double v = sums[k];
    double c = log_c32;
    if (v < 0.0) { // (-0.0 is not below zero: an all-zero sum stays a float-scaled zero, log10 -> -inf either way)
        v = -v;
        c = log_c64;
    }
    logs[k] = log10(v) - c;
}

} // namespace

int agx_phmm_finish_launch(const double *sums, double *logs, uint32_t n, double log_c64, double log_c32,
                           unsigned long long *n_rescued, unsigned long long *n_rescued_host, hipStream_t s)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(phmm_finish_f32, dim3((n + 255) / 256), dim3(256), 0, s, sums, logs, n, log_c64, log_c32, n_rescued, n_rescued_host);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Loads this file's code object now: the first launch of a kernel otherwise pays for it (1-2 ms in a fresh process --
// inside hipvers' launch -> scores window).  Called when a batch that will use these kernels is created.
void agx_phmm_finish_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&phmm_finish_f32));
}
