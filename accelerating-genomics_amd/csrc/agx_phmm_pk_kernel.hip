// The packed float PairHMM fill (agx_phmm_pk_kernel.inc): the builds without read trains, and the kernel that makes the fast
// cell's table rows once per batch.
#include "agx_phmm_pk_kernel.inc"

namespace {

// every read's rows, once per batch: one workgroup of 64 threads per read
__global__ void __launch_bounds__(64) phmm_pk_rows(const uint32_t *__restrict__ img, const PhTab *__restrict__ reads, uint32_t n_reads,
                                                   const float *__restrict__ lut, const float *__restrict__ lut_mis, float4 *__restrict__ rows,
                                                   uint32_t rows_base_dw)
{
    const uint32_t r = blockIdx.x;
    if (r >= n_reads) return;
    const PhTab tb = reads[r];
    const unsigned char *rp = reinterpret_cast<const unsigned char *>(img + tb.read_dw);
    const uint32_t trk = ((tb.R + 3u) >> 2) * 4u;
    float4 *dst = rows + 2 * (size_t)((tb.read_dw - rows_base_dw) / 5u * 4u);
    for (uint32_t i = threadIdx.x; i < tb.R; i += 64) {
        float4 a, b;
        pk_fast_row(rp, trk, (int)tb.R, (int)i, lut, lut_mis, a, b);
        dst[2 * i] = a;
        dst[2 * i + 1] = b;
    }
}

} // namespace

int agx_phmm_pk_launch_class(int cols_per_lane, bool all_groups_16, bool fast, const uint32_t *img, const PhGroup2 *groups, const PhTab *tabs,
                             const PhWave *waves, uint32_t n_waves, const void *lut, const void *lut_mis, double *sums,
                             const PhUnderflow &uf, size_t lds_bytes, hipStream_t s)
{
    if (n_waves == 0) return 0;
    switch (cols_per_lane) {
#define AGX_PH_PK_CASE(CC) \
    case CC: \
        return all_groups_16 ? (fast ? launch<CC, true, true, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s) \
                                       : launch<CC, true, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s)) \
                             : (fast ? launch<CC, false, true, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s) \
                                       : launch<CC, false, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s));
        AGX_PH_FOR_EACH_PK_CLASS(AGX_PH_PK_CASE)
#undef AGX_PH_PK_CASE
    default: return -2;
    }
}

int agx_phmm_pk_rows_launch(const uint32_t *img, const PhTab *reads, uint32_t n_reads, const void *lut, const void *lut_mis, void *rows,
                            uint32_t rows_base_dw, hipStream_t s)
{
    if (n_reads == 0) return 0;
    hipLaunchKernelGGL(phmm_pk_rows, dim3(n_reads), dim3(64), 0, s, img, reads, n_reads, (const float *)lut, (const float *)lut_mis, (float4 *)rows,
                       rows_base_dw);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Loads this file's code object now: the first launch of a kernel otherwise pays for it (1-2 ms in a fresh process --
// inside hipvers' launch -> scores window).  Called when a batch that will use these kernels is created.
void agx_phmm_pk_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&phmm_fill_pk_w3<kPkThreeWaveWidth, true, true, false>));
}
