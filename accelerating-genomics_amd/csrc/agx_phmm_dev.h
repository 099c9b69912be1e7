// Device helpers shared by the PairHMM kernels (included by .hip files only).
#pragma once
#include "agx_phmm.h"

namespace agx_ph {

// DPP wave_shr:1: lane i receives lane i-1's v; lane 0 (always a group's first lane) overrides it.
__device__ __forceinline__ int shr1i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ float shr1(float v) { return __int_as_float(shr1i(__float_as_int(v))); }
__device__ __forceinline__ double shr1(double v)
{
    return __hiloint2double(shr1i(__double2hiint(v)), shr1i(__double2loint(v)));
}

// value of lane l (wave-uniform l) in every lane
__device__ __forceinline__ double lane_value(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <bool FMA> __device__ __forceinline__ double mad(double a, double b, double c)
{
    if constexpr (FMA) return __builtin_fma(a, b, c);
    return a * b + c; // two roundings (the including file is built with contraction off)
}
template <bool FMA> __device__ __forceinline__ float mad(float a, float b, float c)
{
    if constexpr (FMA) return __builtin_fmaf(a, b, c);
    return a * b + c;
}

// The wave's read tables -> LDS.  Table k holds `rows` rows in column-major order
// {Qr[rows], Qi[rows], Qd[rows], Qg[rows], (Qmis[rows]), base[rows]}: G-1 neutral rows
// (Qi = Qd = 0, Qg = 1), the read, a neutral tail.  lut = pow(10, -(c-33)/10) per quality byte
// as the host libm computed it (antidiagsPairHMM.c:104-107); PROBS: the tracks already hold
// probabilities (pairHMM() seam).
template <typename T, bool PROBS>
__device__ __forceinline__ void build_read_tables(unsigned char *lds, const uint32_t *__restrict__ img,
                                                  const PhTab *__restrict__ tabs, uint32_t first_tab, uint32_t n_tabs,
                                                  uint32_t rows, int G, int lane, const T *__restrict__ lut,
                                                  const T *__restrict__ lut_mis)
{
    const bool mis_col = lut_mis != nullptr; // wave-uniform: a fifth table column holds the mismatch prior
    const uint32_t ncol = mis_col ? 5u : 4u;
    const size_t tab_bytes = ph_tab_bytes(sizeof(T) == 8, mis_col, rows);
    for (uint32_t k = 0; k < n_tabs; ++k) {
        const PhTab tb = tabs[first_tab + k];
        T *tq = reinterpret_cast<T *>(lds + k * tab_bytes);
        unsigned char *tc = reinterpret_cast<unsigned char *>(tq + ncol * rows);
        const unsigned char *rp = reinterpret_cast<const unsigned char *>(img + tb.read_dw);
        const uint32_t trk = ((tb.R + 3u) >> 2) * 4u; // bytes per track
        for (uint32_t r = lane; r < rows; r += 64) {
            const int i = (int)r - (G - 1);
            T vr = 0, vi = 0, vd = 0, vg = 1, vm = 0; // neutral row
            unsigned char c = 0;
            if (i >= 0 && i < (int)tb.R) {
                if constexpr (PROBS) { // tracks are probabilities, bases follow them
                    const double *q = reinterpret_cast<const double *>(rp);
                    vr = (T)q[i];
                    vi = (T)q[tb.R + i];
                    vd = (T)q[2 * tb.R + i];
                    vg = (T)q[3 * tb.R + i];
                    c = reinterpret_cast<const unsigned char *>(q + 4 * tb.R)[i];
                } else {
                    c = rp[i];
                    vr = lut[rp[trk + i]];
                    if (mis_col) vm = lut_mis[rp[trk + i]]; // Qr/3 (AGX_PHMM_GATK_PRIOR)
                    vi = lut[rp[2 * trk + i]];
                    vd = lut[rp[3 * trk + i]];
                    vg = lut[rp[4 * trk + i]];
                }
            }
            tq[r] = vr;
            tq[rows + r] = vi;
            tq[2 * rows + r] = vd;
            tq[3 * rows + r] = vg;
            if (mis_col) tq[4 * rows + r] = vm;
            tc[r] = c;
        }
    }
}

} // namespace agx_ph
