// The packed float PairHMM fill with read trains (TRAIN in agx_phmm_pk_kernel.inc): fast cell only, every width.
#include "agx_phmm_pk_kernel.inc"

int agx_phmm_pk_train_launch_class(int cols_per_lane, bool all_groups_16, const uint32_t *img, const PhGroup2 *groups, const PhTab *tabs,
                                   const PhWave *waves, uint32_t n_waves, const void *lut, const void *lut_mis, double *sums,
                                   const PhUnderflow &uf, size_t lds_bytes, hipStream_t s)
{
    if (n_waves == 0) return 0;
    switch (cols_per_lane) {
#define AGX_PH_PK_CASE(CC) \
    case CC: \
        return all_groups_16 ? launch<CC, true, true, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s) \
                             : launch<CC, false, true, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s);
        AGX_PH_FOR_EACH_PK_CLASS(AGX_PH_PK_CASE)
#undef AGX_PH_PK_CASE
    default: return -2;
    }
}

void agx_phmm_pk_train_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&phmm_fill_pk<kPkThreeWaveWidth, true, true, true>));
}
