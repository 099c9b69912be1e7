// Smith-Waterman fill, packed variant: the same wavefront schedule as agx_sw_kernel.hip, but
// every group of G lanes carries TWO alignment pairs at once -- pair A in the low 16 bits and
// pair B in the high 16 bits of each state register -- and the cell recurrence runs on packed
// 16-bit VALU instructions (v_pk_add_i16 / v_pk_max_i16 / v_pk_min_u16 / v_pk_sub_i16).
//
// Why: on gfx950 v_max_i32 / v_max3_i32 / v_cndmask issue at about half the rate of v_add_u32,
// and one packed instruction costs the same issue slot as one of those slow ones while doing two
// cells (tools/valu_microbench.hip, profiles/r01_valu_microbench.log).  12 packed instructions per
// 2 cells replace 2 x 10.5 scalar ones.
//
// Exactness: scores are bounded by the shorter length (<= 2560 < 32767) and the gap states by
// -4 and the -inf stand-in, so int16 lanes hold every value of the reference recurrence
// (antidiagonalSmithWaterman.c:313,:321,:332-335) without wrap-around; results are bit-identical
// to the int32 kernel.  Symbols are compared as (byte << shift): x ^ y is then 0 on a match and
// >= 2^shift >= delta otherwise, so min(x ^ y, delta) is 0 / delta = match - mismatch (2 with the
// reference's +1 / -1).  Scoring constants arrive as kernel arguments (SGPR operands).
#include "agx_sw.h"

namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr short kNeg16 = -16000; // -inf stand-in; never decremented more than once before a max

__device__ __forceinline__ s16x2 splat(short v) { return s16x2{v, v}; }
__device__ __forceinline__ s16x2 as_s(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

__device__ __forceinline__ uint32_t shr1u(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xf, 0xf, false); // wave_shr:1
}

template <int C>
__global__ void __launch_bounds__(256) sw_fill_pk(const SwParams prm, const uint32_t *__restrict__ img,
                                                  const SwGroup2 *__restrict__ groups,
                                                  const SwWave *__restrict__ waves, uint32_t n_waves,
                                                  int32_t *__restrict__ scores)
{
    constexpr int XW = (C + 3) / 4; // dwords holding this lane's C symbols
    const int sh_sym = prm.shift;                         // symbols live as byte << shift
    const uint32_t row_pad = 0x100u << sh_sym;            // never equals (byte << shift)
    const s16x2 ge = as_s(prm.ge2), gf = as_s(prm.gf2), hd = as_s(prm.hd2);
    const u16x2 delta = __builtin_bit_cast(u16x2, prm.delta2);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= n_waves) return;
    const int lane = threadIdx.x & 63;
    const SwWave w = waves[wave];
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

    // this lane's C symbols of both short sequences -> one register per column: (a << 1) | (b << 1) << 16
    // (lane gl owns bytes [gl*C, gl*C + C) of each short sequence; C need not be a multiple of 4)
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

    // state per owned column: z = H - 4 and e = reference P, both pairs packed
    s16x2 z[C], e[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        z[j] = gf;
        e[j] = splat(kNeg16);
    }
    s16x2 z_last = gf, f_last = splat(kNeg16), diag_in = gf, best = gf;
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
        s16x2 zl = as_s(shr1u(0, as_u(z_last)));
        s16x2 fl = as_s(shr1u(0, as_u(f_last)));
        uint32_t yc = shr1u(fresh, yc_prev);
        if (start) { // column 0: H = 0, Q = -inf (antidiagonalSmithWaterman.c:299-306)
            zl = gf;
            fl = splat(kNeg16);
            yc = fresh;
        }
        s16x2 zd = diag_in; // H[r-1][first column - 1] - 4
        diag_in = zl;
        s16x2 zleft = zl, f = fl;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const s16x2 up = z[j];
            const s16x2 ev = pmax(up, e[j] + ge); // reference P, :313
            f = pmax(zleft, f + ge);              // reference Q, :321
            const u16x2 d = __builtin_bit_cast(u16x2, xq[j] ^ yc);
            const u16x2 m2 = __builtin_elementwise_min(d, delta); // 0 on a match, match - mismatch otherwise
            // H_diag + match >= 1 as an unsigned value; the saturating subtract of 0 / delta yields
            // max(H_diag + match / + mismatch, 0): the diagonal move (:332) and the zero floor of :333 at once
            const u16x2 hd1 = __builtin_bit_cast(u16x2, zd + hd);
            const s16x2 s = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(hd1, m2));
            const s16x2 v = pmax(pmax(ev, f), s); // :333 (e, f may be negative, s carries the floor)
            const s16x2 zn = v + gf;
            e[j] = ev;
            z[j] = zn;
            zd = up;
            zleft = zn;
            best = pmax(best, zn); // :335
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
        const s16x2 other = as_s((uint32_t)__shfl_down((int)as_u(best), o));
        if (gl + o < G) best = pmax(best, other);
    }
    if (feeder) {
        scores[g.out[0]] = (int)best[0] - prm.gf;
        if (g.out[1] < prm.n_out) scores[g.out[1]] = (int)best[1] - prm.gf; // a group without a second pair points this at the spare slot
    }
}

template <int C>
int launch(const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves, uint32_t n_waves,
           int32_t *scores, hipStream_t s)
{
    const uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(sw_fill_pk<C>, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace

int agx_sw_pk_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups,
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
