#include "agx_sw_kernel.inc"

template <int C>
int launch(const SwParams &prm, const uint32_t *img, const SwGroup *groups, const SwWave *waves, uint32_t n_waves,
           int32_t *scores, hipStream_t s)
{
    const uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(sw_fill<C>, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores, (const int16_t *)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace

int agx_sw_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup *groups,
                        const SwWave *waves, uint32_t n_waves, int32_t *scores, hipStream_t s)
{
    if (n_waves == 0) return 0;
    switch (cols_per_lane) {
#define AGX_SW_CASE(CC) \
    case CC: return launch<CC>(prm, img, groups, waves, n_waves, scores, s);
        AGX_SW_FOR_EACH_CLASS(AGX_SW_CASE)
#undef AGX_SW_CASE
    default: return -2;
    }
}

// Loads this file's code object now: the first launch of a kernel otherwise pays for it (1-2 ms in a fresh process --
// inside hipvers' launch -> scores window).  Called when a batch that will use these kernels is created.
void agx_sw_i32_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&sw_fill<38>));
}
