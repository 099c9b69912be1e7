// Very wide classes of the int32 kernel (80/120/160 columns per lane: up to 10 240 columns, the
// line limit of hipvers.cpp:40).  One wave per SIMD, state spills into the accumulator half of the
// register file; correctness path for rare very long pairs, built in its own translation unit.
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

int agx_sw_wide_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup *groups,
                        const SwWave *waves, uint32_t n_waves, int32_t *scores, hipStream_t s)
{
    if (n_waves == 0) return 0;
    switch (cols_per_lane) {
#define AGX_SW_CASE(CC) \
    case CC: return launch<CC>(prm, img, groups, waves, n_waves, scores, s);
        AGX_SW_FOR_EACH_WIDE_CLASS(AGX_SW_CASE)
#undef AGX_SW_CASE
    default: return -2;
    }
}
