// Smith-Waterman fill, packed variant, second formulation ("biased"): the schedule and data layout of
// agx_sw_pk_kernel.hip -- two alignment pairs per lane group, pair A in the low and pair B in the
// high 16 bits of every state register -- with the cell rewritten for the issue rates measured on
// gfx950 (tools/valu_microbench2.hip, profiles/r02_valu_microbench2.log):
//
//   * v_add_u32 / v_sub_u32 / v_xor_b32 with VGPR-only operands issue in about 2.4 cycles per wave64,
//     every v_pk_* instruction in 4.2.  All four additions of the recurrence therefore run as plain
//     32-bit adds on both halves at once.  No carry or borrow may cross bit 16, so every value is kept
//     as an UNSIGNED half with a bias B added: stored = true + B >= 0 always, constants are subtracted
//     (never added as two's complement), and the vertical gap state is kept clamped at zero
//     (P~ = max(P, 0): a negative P never reaches H -- H >= 0 -- and its successors P - 1, P - 2, ... are
//     negative too, so max(H_up + gf, P~_up + ge, 0) = max(P_new, 0) exactly).  That clamp is also what
//     delivers the zero floor of antidiagonalSmithWaterman.c:333: H = max(P~, Q, H_diag + s) >= 0.
//   * gfx950 has a packed three-input maximum, v_pk_maximum3_f16.  With B >= 1024 + |gf| + delta and all
//     values below 0x7c00 every stored half is the bit pattern of a positive NORMAL half-precision number,
//     and for those the floating-point order is the integer order: the instruction is an exact unsigned
//     max3 here.  It folds the clamp into the gap maximum and the two maxima of :333 into one.
//
//   per two cells:  e' = max3(z_up, e - |ge|, B)         v_sub_u32, v_pk_maximum3_f16     (:313, clamped)
//                   f  = max(z_left, f - |ge|)           v_sub_u32, v_pk_max_u16          (:321)
//                   m  = min(x ^ y, delta)               v_xor_b32, v_pk_min_u16          (:332, match test)
//                   u  = (z_diag + hd) - m               v_add_u32, v_sub_u32             (:332)
//                   H' = max3(e', f, u)                  v_pk_maximum3_f16                (:333)
//                   z  = H' - |gf|                       v_sub_u32
//                   best = max3(best, z, z_next)         half a v_pk_maximum3_f16         (:335)
//   = 6 full-rate + 4.5 packed instructions against 1 + 11 in agx_sw_pk_kernel.hip.
//
// The host picks this kernel when the scoring and the longest shorter side keep every stored half in
// [0x0400, 0x7c00) (agx_sw.cpp; always true for the reference's +1/-1/-3/-1 up to 2560 columns);
// scores are bit-identical to the other kernels and to the reference.
#include "agx_sw.h"

namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 as_v(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t umax2(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_v(a), as_v(b))); }
__device__ __forceinline__ uint32_t umin2(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_min(as_v(a), as_v(b))); }
// exact unsigned max3 per half for patterns of positive normal half-precision numbers (see above)
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a wave-uniform constant forced into a VGPR: with an SGPR or literal operand v_add/v_sub_u32 fall back
// to the 4-cycle rate ("v_subrev_u32 SGPR constant" in the microbenchmark)
__device__ __forceinline__ uint32_t in_vgpr(uint32_t s)
{
    uint32_t r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(s));
    return r;
}

__device__ __forceinline__ uint32_t shr1u(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xf, 0xf, false); // wave_shr:1
}

template <int C>
__device__ __forceinline__ void pk2_body(const SwParams &prm, const uint32_t *__restrict__ img, const SwGroup2 *__restrict__ groups,
                                         const SwWave w, int32_t *__restrict__ scores)
{
    static_assert(C % 2 == 0, "the running maximum takes two columns per instruction");
    constexpr int XW = (C + 3) / 4; // dwords holding this lane's C symbols
    const int sh_sym = prm.shift;              // symbols live as byte << shift
    const uint32_t row_pad = 0x100u << sh_sym; // never equals (byte << shift)
    const uint32_t ge = in_vgpr(prm.age2), gf = in_vgpr(prm.agf2), hd = in_vgpr(prm.hd2); // |ge|, |gf|, match + |gf|
    const uint32_t bias = prm.bias2, delta = prm.delta2;
    const uint32_t z0 = prm.bias2 - prm.agf2; // H = 0 as the state both gap recurrences read (z = H + gf)
    const int lane = threadIdx.x & 63;
    const int G = w.G;
    const int grp = lane / G;
    const int gl = lane - grp * G;
    const bool active = grp < (int)w.n_groups;
    const bool start = gl == 0;
    const bool feeder = active && start;

    SwGroup2 g;
#pragma unroll
    for (int k = 0; k < 2; ++k) g.x_dw[k] = g.y_dw[k] = g.lx_ly[k] = g.out[k] = 0;
    if (active) g = groups[w.first_group + grp];
    const int lyA = (int)(g.lx_ly[0] >> 16), lyB = (int)(g.lx_ly[1] >> 16);
    const int nqA = (lyA + 3) >> 2, nqB = (lyB + 3) >> 2;

    // this lane's C symbols of both short sequences -> one register per column: (a << shift) | (b << shift) << 16
    uint32_t xq[C];
    {
        const uint32_t o = (uint32_t)gl * C, d0 = o >> 2, sh = o & 3u;
        uint32_t ra[XW + 1], rb[XW + 1];
#pragma unroll
        for (int k = 0; k <= XW; ++k) {
            ra[k] = active ? img[g.x_dw[0] + d0 + k] : 0u;
            rb[k] = active ? img[g.x_dw[1] + d0 + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < XW; ++k) {
            const uint32_t a = __builtin_amdgcn_alignbyte(ra[k + 1], ra[k], sh);
            const uint32_t b = __builtin_amdgcn_alignbyte(rb[k + 1], rb[k], sh);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * k + i < C)
                    xq[4 * k + i] = ((((a >> (8 * i)) & 0xffu) << sh_sym) | ((((b >> (8 * i)) & 0xffu) << sh_sym) << 16));
        }
    }

    const uint32_t *ypA = img + g.y_dw[0], *ypB = img + g.y_dw[1];
    auto quadA = [&](int q) -> uint32_t { return (feeder && q < nqA) ? ypA[q] : 0u; };
    auto quadB = [&](int q) -> uint32_t { return (feeder && q < nqB) ? ypB[q] : 0u; };

    // state per owned column, both pairs packed, biased: z = H + gf + B and e = max(P, 0) + B
    uint32_t z[C], e[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        z[j] = z0;
        e[j] = bias;
    }
    // the horizontal gap state needs no clamp: Q >= z_left >= gf always; "no gap open yet" is Q = gf,
    // whose successor gf + ge loses against every z_left
    uint32_t z_last = z0, f_last = z0, diag_in = z0, best = z0;
    uint32_t yc_prev = row_pad | (row_pad << 16);

    uint32_t a0 = quadA(0), a1 = quadA(1), a2 = quadA(2);
    uint32_t b0 = quadB(0), b1 = quadB(1), b2 = quadB(2);
    const int steps = (int)w.steps;
    uint32_t rowsA = 0, rowsB = 0;
    int t = 0;

    auto step = [&]() __attribute__((always_inline)) {
        const uint32_t fa = (t < lyA) ? ((rowsA & 0xffu) << sh_sym) : row_pad;
        const uint32_t fb = (t < lyB) ? ((rowsB & 0xffu) << sh_sym) : row_pad;
        rowsA >>= 8;
        rowsB >>= 8;
        const uint32_t fresh = fa | (fb << 16);
        uint32_t zl = shr1u(0, z_last);
        uint32_t fl = shr1u(0, f_last);
        uint32_t yc = shr1u(fresh, yc_prev);
        if (start) { // column 0: H = 0, Q = -inf (antidiagonalSmithWaterman.c:299-306)
            zl = z0;
            fl = z0;
            yc = fresh;
        }
        uint32_t zd = diag_in; // H[r-1][first column - 1] + gf
        diag_in = zl;
        uint32_t zleft = zl, f = fl;
#pragma unroll
        for (int j = 0; j < C; j += 2) {
            uint32_t zn[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t up = z[j + k];
                const uint32_t ev = umax3(up, e[j + k] - ge, bias); // reference P, :313, clamped at 0
                f = umax2(zleft, f - ge);                          // reference Q, :321
                const uint32_t m = umin2(xq[j + k] ^ yc, delta);   // 0 on a match, match - mismatch otherwise
                const uint32_t u = (zd + hd) - m;                  // H_diag + match / + mismatch, :332
                const uint32_t v = umax3(ev, f, u);                // :333 (ev >= B carries the zero floor)
                zn[k] = v - gf;
                e[j + k] = ev;
                z[j + k] = zn[k];
                zd = up;
                zleft = zn[k];
            }
            best = umax3(best, zn[0], zn[1]); // :335
        }
        z_last = zleft;
        f_last = f;
        yc_prev = yc;
        ++t;
    };

    const int quads = steps >> 2;
    for (int q = 0; q < quads; ++q) {
        rowsA = a0;
        a0 = a1;
        a1 = a2;
        a2 = quadA(q + 3);
        rowsB = b0;
        b0 = b1;
        b1 = b2;
        b2 = quadB(q + 3);
#pragma unroll
        for (int b = 0; b < 4; ++b) step();
    }
    rowsA = a0;
    rowsB = b0;
#pragma unroll 1
    while (t < steps) step();

    // max over the group's lanes (G need not be a power of two), both halves at once
    for (int o = 1; o < G; o <<= 1) {
        const uint32_t other = (uint32_t)__shfl_down((int)best, o);
        if (gl + o < G) best = umax2(best, other);
    }
    if (feeder) {
        const int off = (int)(z0 & 0xffffu); // stored value of H = 0
        scores[g.out[0]] = (int)(best & 0xffffu) - off;
        scores[g.out[1]] = (int)(best >> 16) - off; // a group without a second pair points this at the spare slot
    }
}

template <int C>
__global__ void __launch_bounds__(256) sw_fill_pk2(const SwParams prm, const uint32_t *__restrict__ img,
                                                   const SwGroup2 *__restrict__ groups,
                                                   const SwWave *__restrict__ waves, uint32_t n_waves,
                                                   int32_t *__restrict__ scores)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= n_waves) return;
    pk2_body<C>(prm, img, groups, waves[wave], scores);
}

// Mixed batches: ONE launch for every lane-tiling class.  Each wavefront reads its class (columns per lane)
// from its record and runs that class's fill; the kernel is allocated the registers of the widest class
// (100 VGPRs, five waves per SIMD -- the fill is bound by VALU issue, not by occupancy).  Against one launch
// per class this (a) lets the planner use every width, so padding shrinks, (b) dispatches the waves of ALL
// classes longest first, (c) has no stream fork/join and no per-launch ramp.
__global__ void __launch_bounds__(256) sw_fill_pk2_any(const SwParams prm, const uint32_t *__restrict__ img,
                                                       const SwGroup2 *__restrict__ groups,
                                                       const SwWave *__restrict__ waves, uint32_t n_waves,
                                                       int32_t *__restrict__ scores)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= n_waves) return;
    const SwWave w = waves[wave];
    switch (__builtin_amdgcn_readfirstlane(w.reserved)) { // columns per lane of this wave
#define AGX_SW_CASE(CC) \
    case CC: pk2_body<CC>(prm, img, groups, w, scores); break;
        AGX_SW_FOR_EACH_CLASS(AGX_SW_CASE)
#undef AGX_SW_CASE
    default: break;
    }
}

template <int C>
int launch(const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves, uint32_t n_waves,
           int32_t *scores, hipStream_t s)
{
    const uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(sw_fill_pk2<C>, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace

int agx_sw_pk2_launch_any(const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves, uint32_t n_waves,
                          int32_t *scores, hipStream_t s)
{
    if (n_waves == 0) return 0;
    const uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(sw_fill_pk2_any, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int agx_sw_pk2_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups,
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
