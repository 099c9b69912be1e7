// PairHMM forward recurrence, packed float variant (AGX_PHMM_F32_FMA): the schedule of
// agx_phmm_kernel.hip -- haplotype columns across a group of G lanes, read rows streaming skewed,
// neighbours by DPP, per-read probability table in LDS -- but every lane group carries TWO
// haplotypes of the same read, one in each half of a float2, and the cell runs on packed fp32
// instructions with FMA contraction:
//     t = gm * (X_d + Y_d)            v_pk_add_f32, v_pk_mul_f32
//     M = prior * fma(mm, M_d, t)     v_pk_fma_f32, v_pk_mul_f32     (antidiagsPairHMM.c:184)
//     X = fma(M_up, Qi, X_up * Qg)    v_pk_mul_f32, v_pk_fma_f32     (:189)
//     Y = fma(Y_left, Qg, M_left*Qd)  v_pk_mul_f32, v_pk_fma_f32     (:194; the product is off the column chain)
// = 8 packed instructions + 2 compares + 2 selects per 2 cells, against 11 + 2 scalar ones per cell
// in the order-exact float kernel; v_pk_*_f32 issue at 75 T elements/s on this chip where
// v_mul/v_add_f32 reach 65 and v_fma_f32 42 (tools/valu_microbench.hip), and
// tools/phmm_mix_microbench.hip measured the two cell loops at 0.17 vs 0.32 ps/cell.
//
// FAST (the default; the host asks for it when (a) no read has a gap-continuation quality of Phred 0, i.e. 1 - Qg > 0
// everywhere, (b) haplotypes hold only A, C, G, T and reads only A, C, G, T, N) changes two things:
//   * X and Y are stored multiplied by the NEXT row's gm = 1 - Qg, so the diagonal term gm * (X_d + Y_d) is a bare
//     sum and one multiplication per two cells is gone.  Every use of X and Y is a product with a row constant, so the
//     factor folds into the per-read table: X' = fma(M_up, Qi gm+, X'_up * (Qg gm+ / gm)), Y' = fma(Y'_left, Qg,
//     M_left * (Qd gm+)); gm+ = 1 behind the last row, so the final sum reads the true X; the row-0 state Y = init
//     is scaled by the first row's gm.
//   * the match test is a table lookup: a row carries T, the byte 0x3f at the position of its base's two-bit code
//     ((b >> 1) & 3; everywhere for N), each column one v_perm_b32 selector {0x0c, 0x0c, 4 + code_b, code_a}.
//     v_perm_b32(0, T, sel) is 0x3f000000 = 0.5f where haplotype a matches, v_perm_b32(T', 0, sel) with
//     T' = (T & 0x01010101) << 7 is 0x00800000 = 2^-126 where b does, and prior = fma({0.5, 2^-126} or 0,
//     {2 (pm - pq), 2^126 (pm - pq)}, pq) -- three instructions where two compares and two selects stood.
//   * M is stored times the row's D = Qd gm+, the factor the Y chain multiplies it with (1 in the last row, whose Y
//     nobody reads, so the final sum sees the true M): Y' = fma(Y'_left, Qg, M_left) needs no product, the factor
//     folds into the priors and, through the previous row's D, out of the other two uses of M (row constants
//     mm / D-, Qi gm+ / D-).
// = 9 packed/perm instructions per 2 cells against 12.
//
// Numerics: float with the initial constant FLT_MAX/16 like AGX_PHMM_F32, but contracted -- not
// bit-identical to the oracle's float restatement; the bar is BASELINE config 3's 1e-6 relative on
// the log10 likelihood (tests/test_phmm_gpu.py).  Pairs whose float sum underflows are recomputed
// by the double kernel (its own plan, RESCUE mode) exactly as for AGX_PHMM_F32.
#include "agx_phmm.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kPkThreeWaveWidth = 19;

__device__ __forceinline__ int shr1i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); } // wave_shr:1
__device__ __forceinline__ f2 shr1(f2 v)
{
    return f2{__int_as_float(shr1i(__float_as_int(v.x))), __int_as_float(shr1i(__float_as_int(v.y)))};
}
// DPP row_shr:1 with bound_ctrl: lane i of every 16-lane row receives lane i-1's v, a row's first lane 0
__device__ __forceinline__ int rshr1i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ f2 rshr1(f2 v)
{
    return f2{__int_as_float(rshr1i(__float_as_int(v.x))), __int_as_float(rshr1i(__float_as_int(v.y)))};
}
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// One row of the fast cell's table for read position i (0 <= i < R): {pq D, 2 (pm - pq) D, (1-(Qi+Qd)) / D-, Qi gm+ / D- |
// Qg gm+ / gm, -, Qg, T} (D = Qd gm+ of this row, D- of the previous one; see phmm_fill_pk_body).  rp: the read's five
// byte tracks, trk bytes each.  Used by the fill (a wave deriving its own rows) and by phmm_pk_rows (once per batch).
__device__ __forceinline__ void pk_fast_row(const unsigned char *__restrict__ rp, uint32_t trk, int R, int i, const float *__restrict__ lut,
                                            const float *__restrict__ lut_mis, float4 &ra, float4 &rb)
{
    const uint32_t c = rp[i];
    const float vr = lut[rp[trk + i]];
    const float vm = lut_mis ? lut_mis[rp[trk + i]] : vr; // Qr/3 with AGX_PHMM_GATK_PRIOR, else Qr itself
    const float vi = lut[rp[2 * trk + i]], vd = lut[rp[3 * trk + i]], vg = lut[rp[4 * trk + i]];
    const float pm = 1 - vr;                       // p(): match or N
    const float pq = c == (uint32_t)'N' ? pm : vm; //      mismatch
    auto gm_of = [&](int k) -> float { // 1 - Qg of read row k, extended to both sides
        if (k >= R) return 1.f;
        return 1 - lut[rp[4 * trk + (k < 0 ? 0 : k)]];
    };
    const double g = gm_of(i), gp = gm_of(i + 1);
    auto d_of = [&](int k) -> double { // D of read row k; 1 outside the read and in its last row
        if (k < 0 || k >= R - 1) return 1.0;
        return (double)lut[rp[3 * trk + k]] * (double)gm_of(k + 1);
    };
    const double dcur = d_of(i), dprev = d_of(i - 1);
    const uint32_t tbl = c == (uint32_t)'N' ? 0x3f3f3f3fu : 0x3fu << (8u * ((c >> 1) & 3u));
    ra = float4{(float)((double)pq * dcur), (float)(2 * ((double)pm - (double)pq) * dcur), (float)((1 - ((double)vi + (double)vd)) / dprev),
                (float)((double)vi * gp / dprev)};
    rb = float4{(float)((double)vg * gp / g), 0.f, vg, __uint_as_float(tbl)};
}

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

// ROW16: every wave of the launch has groups of exactly 16 lanes (uniform batches such as H = 300 in 16 x 19):
// the groups coincide with the DPP rows, and the row shift's zero fill is the column-0 boundary.
template <int C, bool ROW16, bool FAST>
__device__ __forceinline__ void phmm_fill_pk_body(const uint32_t *__restrict__ img, const PhGroup2 *__restrict__ groups,
                                                  const PhTab *__restrict__ tabs, const PhWave *__restrict__ waves,
                                                  uint32_t n_waves, const float *__restrict__ lut,
                                                  const float *__restrict__ lut_mis, double *__restrict__ sums, const PhUnderflow uf)
{
    constexpr int HW = (C + 3) / 4; // dwords holding this lane's C haplotype bases
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t wave = blockIdx.x; // one wavefront per workgroup
    if (wave >= n_waves) return;
    const int lane = threadIdx.x;
    const PhWave w = waves[wave];
    const int G = w.G;
    const int grp = lane / G;
    const int gl = lane - grp * G;
    const bool active = grp < (int)w.n_groups;
    const bool start = gl == 0;

    PhGroup2 g;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        g.hap_dw[k] = g.H[k] = g.out[k] = 0;
        g.init32[k] = 0;
    }
    g.R_tab = 0;
    if (active) g = groups[w.first_group + grp];
    const int R = (int)(g.R_tab & 0xffffu);
    const int HA = (int)g.H[0], HB = (int)g.H[1];

    // ---- read tables -> LDS.  One 32-byte row per read position holds what the cell loop consumes,
    // already derived: {1-Qr, mismatch prior, 1-(Qi+Qd), 1-Qg | Qi, Qd, Qg, base} -- the same float
    // operations p()/mm() prescribe (:111-117), done once per row here instead of once per lane and
    // step; a lane fetches its row with two ds_read_b128 from one address.  Neutral rows (before
    // the read and behind it): priors irrelevant, mm = 1, 1-Qg = 0, Qi = Qd = 0, Qg = 1.
    // FAST: {pq D, 2 (pm - pq) D, (1-(Qi+Qd)) / D-, Qi gm+ / D- | Qg gm+ / gm, -, Qg, T} (D = Qd gm+ of this row, D- of the
    // previous one; see below), gm of a row before the read =
    // the first row's, behind it = 1, and both priors of a neutral row 0 (neutral rows then keep the scaled state as it is).
    const uint32_t rows = w.steps + (uint32_t)G - 1u;
    const size_t tab_bytes = ph_pk_tab_bytes(rows);
    for (uint32_t k = 0; k < w.n_tabs; ++k) {
        const PhTab tb = tabs[w.first_tab + k];
        float4 *tr = reinterpret_cast<float4 *>(lds + k * tab_bytes);
        const unsigned char *rp = reinterpret_cast<const unsigned char *>(img + tb.read_dw);
        const uint32_t trk = ((tb.R + 3u) >> 2) * 4u; // bytes per track
        const float4 *pre = (FAST && uf.pk_rows) ? reinterpret_cast<const float4 *>(uf.pk_rows) + 2 * (size_t)((tb.read_dw - uf.rows_base_dw) / 5u * 4u) : nullptr;
        for (uint32_t r = lane; r < rows; r += 64) {
            const int i = (int)r - (G - 1);
            float vr = 0, vi = 0, vd = 0, vg = 1, vm = 0; // neutral row
            uint32_t c = 0;
            if (i >= 0 && i < (int)tb.R) {
                c = rp[i];
                vr = lut[rp[trk + i]];
                vm = lut_mis ? lut_mis[rp[trk + i]] : vr; // Qr/3 with AGX_PHMM_GATK_PRIOR, else Qr itself
                vi = lut[rp[2 * trk + i]];
                vd = lut[rp[3 * trk + i]];
                vg = lut[rp[4 * trk + i]];
            }
            const float pm = 1 - vr;                          // p(): match or N
            const float pq = c == (uint32_t)'N' ? pm : vm;    //      mismatch
            if constexpr (FAST) {
                // (a neutral row: both priors zero, every factor one -- the scaled state passes through it unchanged; the plain
                // cell's neutral rows keep M = 0 through gm = 0, which this cell no longer multiplies with)
                float4 fa = float4{0.f, 0.f, 1.f, 0.f}, fb = float4{1.f, 0.f, 1.f, 0.f};
                if (i >= 0 && i < (int)tb.R) {
                    if (pre) { // made once per batch (phmm_pk_rows)
                        fa = pre[2 * i];
                        fb = pre[2 * i + 1];
                    } else
                        pk_fast_row(rp, trk, (int)tb.R, i, lut, lut_mis, fa, fb);
                }
                tr[2 * r] = fa;
                tr[2 * r + 1] = fb;
            } else {
                tr[2 * r] = float4{pm, pq, 1 - (vi + vd), 1 - vg}; // mm() (:115-117)
                tr[2 * r + 1] = float4{vi, vd, vg, __uint_as_float(c)};
            }
        }
    }
    __syncthreads();

    const uint32_t tabi = g.R_tab >> 16;
    const float4 *trow = reinterpret_cast<const float4 *>(lds + tabi * tab_bytes) + 2 * (G - 1 - gl);

    // this lane's C bases of both haplotypes, one register per column: a | b << 16;  FAST: the v_perm_b32 selector
    // {0x0c, 0x0c, 4 + code_b, code_a}, 0x0c (the constant 0: matches nothing) for the padding behind a haplotype
    uint32_t hq[C];
    unsigned long long na = 0, nb = 0; // haplotype 'N' matches every read base (:111-113); rare
    {
        const uint32_t o = (uint32_t)gl * C, d0 = o >> 2, sh = o & 3u;
        uint32_t ra[HW + 1], rb[HW + 1];
#pragma unroll
        for (int k = 0; k <= HW; ++k) {
            ra[k] = active ? img[g.hap_dw[0] + d0 + k] : 0u;
            rb[k] = active ? img[g.hap_dw[1] + d0 + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < HW; ++k) {
            const uint32_t a = __builtin_amdgcn_alignbyte(ra[k + 1], ra[k], sh);
            const uint32_t b = __builtin_amdgcn_alignbyte(rb[k + 1], rb[k], sh);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * k + i < C) {
                    const uint32_t ca = (a >> (8 * i)) & 0xffu, cb = (b >> (8 * i)) & 0xffu;
                    if constexpr (FAST)
                        hq[4 * k + i] = 0x0c0cu | ((cb ? 4u + ((cb >> 1) & 3u) : 0x0cu) << 16) | ((ca ? (ca >> 1) & 3u : 0x0cu) << 24);
                    else
                        hq[4 * k + i] = ca | (cb << 16);
                    na |= (ca == (uint32_t)'N' ? 1ull : 0ull) << (4 * k + i);
                    nb |= (cb == (uint32_t)'N' ? 1ull : 0ull) << (4 * k + i);
                }
        }
    }

    f2 init = f2{g.init32[0], g.init32[1]};
    if constexpr (FAST) { // row 0's Y as the first read row consumes it: times that row's gm
        if (active) {
            const PhTab tb = tabs[w.first_tab + tabi];
            const unsigned char *rp = reinterpret_cast<const unsigned char *>(img + tb.read_dw);
            init *= 1 - lut[rp[4u * (((tb.R + 3u) >> 2) * 4u)]];
        }
    }
    f2 M[C], X[C], Y[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        M[j] = splat(0.f);
        X[j] = splat(0.f);
        Y[j] = init;
    }
    f2 pM = splat(0.f), pX = splat(0.f), pY = init;
    // every lane sums its own columns of the last row; the lanes are combined in double after the loop
    // (a float running sum over thousands of columns would lose the 1e-6 the mode promises)
    double part_a = 0, part_b = 0;
    const int steps = (int)w.steps;
    const int col0 = gl * C;

    auto fill = [&](auto hapn_tag) {
        constexpr bool HAPN = decltype(hapn_tag)::value;
        // (two steps per trip: the values a step hands to the next one -- pM, pX, pY -- then change registers by
        // renaming instead of six v_mov per step)
        auto one_step = [&](int t) __attribute__((always_inline)) {
            const float4 ra = trow[2 * t], rb = trow[2 * t + 1];
            const float pm = ra.x, pq = ra.y;
            const uint32_t rc = __float_as_uint(rb.w);
            // plain: {pm, pq, mm, gm | Qi, Qd, Qg, base};  FAST: {pq D, 2 (pm - pq) D, mm / D-, Qi gm+ / D- | Qg gm+ / gm, -, Qg, T}
            const f2 mm = splat(ra.z), gm = splat(ra.w);
            const f2 qi = splat(FAST ? ra.w : rb.x), qd = splat(rb.y), qg = splat(rb.z);
            const f2 qx = splat(FAST ? rb.x : rb.z); // what X_up is multiplied with
            const f2 pq2 = splat(ra.x), dp2 = f2{ra.y, ra.y * 0x1p125f}; // FAST: prior = fma(flag, dp2, pq2)
            const uint32_t t80 = (rc & 0x01010101u) << 7;

            f2 lM, lX, lY; // left neighbours; column 0 of rows >= 1 is all zeros (:168-178)
            if constexpr (ROW16) {
                lM = rshr1(M[C - 1]);
                lX = rshr1(X[C - 1]);
                lY = rshr1(Y[C - 1]);
            } else {
                lM = shr1(M[C - 1]);
                lX = shr1(X[C - 1]);
                lY = shr1(Y[C - 1]);
                if (start) {
                    lM = splat(0.f);
                    lX = splat(0.f);
                    lY = splat(0.f);
                }
            }
            const f2 dM0 = pM, dX0 = pX, dY0 = pY;
            pM = lM;
            pX = lX;
            pY = lY;
            // pass A, right to left: M and X in place
#pragma unroll
            for (int j = C - 1; j >= 0; --j) {
                f2 prior;
                if constexpr (FAST) {
                    const f2 flag = f2{__uint_as_float(__builtin_amdgcn_perm(0u, rc, hq[j])), __uint_as_float(__builtin_amdgcn_perm(t80, 0u, hq[j]))};
                    prior = fma2(flag, dp2, pq2);
                } else {
                    bool ma = (hq[j] & 0xffffu) == rc, mb = (hq[j] >> 16) == rc;
                    if constexpr (HAPN) {
                        ma = ma || ((na >> j) & 1ull);
                        mb = mb || ((nb >> j) & 1ull);
                    }
                    prior = f2{ma ? pm : pq, mb ? pm : pq};
                }
                const f2 dM = j ? M[j > 0 ? j - 1 : 0] : dM0;
                const f2 dX = j ? X[j > 0 ? j - 1 : 0] : dX0;
                const f2 dY = j ? Y[j > 0 ? j - 1 : 0] : dY0;
                const f2 x = fma2(M[j], qi, X[j] * qx);
                const f2 m = prior * fma2(mm, dM, FAST ? dX + dY : gm * (dX + dY));
                X[j] = x;
                M[j] = m;
            }
            // pass B, left to right: Y needs the new M and Y of column j-1.  The products M * Qd do not depend on
            // the chain, so only one fused multiply-add per column sits on it (mul + fma, both on the chain, took
            // twice the latency per column and half of the loop's hazard nops).
            f2 cY = lY;
            if constexpr (FAST) { // M is stored times Qd gm+ already: one fused multiply-add per column, nothing else
                f2 cM = lM;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const f2 y = fma2(cY, qg, cM);
                    cM = M[j];
                    cY = y;
                    Y[j] = y;
                }
            } else {
                f2 a[3] = {lM * qd, M[0] * qd, C > 1 ? M[C > 1 ? 1 : 0] * qd : splat(0.f)}; // products run three columns ahead of the chain
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const f2 y = fma2(cY, qg, a[j % 3]);
                    if (j + 3 < C + 1) a[j % 3] = M[j + 2 < C ? j + 2 : C - 1] * qd;
                    cY = y;
                    Y[j] = y;
                }
            }
            if (t - gl + 1 == R) { // last read row: likelihood (:206-212)
                // This block runs once per lane position (G times per wave, a few lanes each), so it is
                // kept short: four float accumulators per half (at most 8 terms each: the rounding stays
                // below 3e-7 of the lane's sum), one conversion to double per half.  Lanes whose columns
                // all lie inside both haplotypes -- all of a step's active lanes share gl -- skip the masks.
                f2 a0 = splat(0.f), a1 = splat(0.f), a2 = splat(0.f), a3 = splat(0.f);
                if (col0 + C <= HA && col0 + C <= HB) {
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        const f2 v = M[j] + X[j];
                        if ((j & 3) == 0) a0 += v;
                        if ((j & 3) == 1) a1 += v;
                        if ((j & 3) == 2) a2 += v;
                        if ((j & 3) == 3) a3 += v;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        f2 v = M[j] + X[j];
                        if (col0 + j >= HA) v.x = 0.f;
                        if (col0 + j >= HB) v.y = 0.f;
                        if ((j & 3) == 0) a0 += v;
                        if ((j & 3) == 1) a1 += v;
                        if ((j & 3) == 2) a2 += v;
                        if ((j & 3) == 3) a3 += v;
                    }
                }
                const f2 tot = (a0 + a1) + (a2 + a3);
                part_a = (double)tot.x;
                part_b = (double)tot.y;
            }
        };
        int t = 0;
        // (only the builds with registers to spare: the plain cell at widths 19, 31, 32 sits at its cap and spills 40-100
        // values when two steps are in flight; width 19's fast cell -- config 3, capped at 168 registers for three waves
        // per SIMD -- spilled five and ran no faster than one step per trip without spills: 0.321-0.327 against 0.326 ms)
        if constexpr (FAST && ROW16 && C != kPkThreeWaveWidth) {
            for (; t + 1 < steps; t += 2) {
                one_step(t);
                one_step(t + 1);
            }
            if (t < steps) one_step(t);
        } else
            for (; t < steps; ++t) one_step(t);
    };
    if (!FAST && __any((na | nb) != 0))
        fill(std::true_type{});
    else
        fill(std::false_type{});

    for (int dlt = 1; dlt < G; dlt <<= 1) { // inclusive scan over the group's lanes
        const double va = __shfl_up(part_a, dlt), vb = __shfl_up(part_b, dlt);
        if (gl >= dlt) {
            part_a += va;
            part_b += vb;
        }
    }
    unsigned under = 0;
    if (active && gl == G - 1) {
        if (uf.guard_k2 != 0.0f) { // accuracy guard: a likelihood this close to 1 for a read this long goes to the double pass
            // (the hardware's own v_sqrt_f32 / v_exp_f32: three instructions; the library calls cost 2 % of the launch)
            const float near_one = (float)uf.guard_c * __builtin_amdgcn_exp2f(-uf.guard_k2 * __builtin_amdgcn_sqrtf((float)(g.R_tab & 0xffffu) + 8.0f));
            if ((float)part_a > near_one) part_a = 0.0;
            if ((float)part_b > near_one) part_b = 0.0;
        }
        sums[g.out[0]] = part_a;
        sums[g.out[1]] = part_b; // a group without a second haplotype points this at the spare slot
        // the census the host reads with the results: any pair below the float range means the double rescue plan
        // has to run (it is launched only then -- agx_phmm_batch_results)
        under = (unsigned)(!(part_a >= uf.below) && g.out[0] < uf.n_pairs) + (unsigned)(!(part_b >= uf.below) && g.out[1] < uf.n_pairs);
        if (under) atomicAdd(uf.count, (unsigned long long)under);
    }
    if (uf.logs_host) {
        // Bound results (agx_phmm_batch_bind_results): the last line of pairHMM() -- log10(sum) - log10(C), :242 -- here, and
        // straight into the caller's page-locked array.  The wave's results move to its first lanes (lane i takes group i's
        // values from that group's last lane) so that neighbouring haplotypes leave as adjacent stores: one PCIe write per
        // wave in a batch planned in output order.  A pair that went to the rescue plan sets the flag: the
        // host then takes the results the long way (rescue pass, log10 kernel).
        // lane j takes value j & 1 of group j >> 1 (from that group's last lane): ONE log10 per lane, and neighbouring
        // haplotypes leave as adjacent 8-byte stores of adjacent lanes
        const int src = (((lane >> 1) + 1) * G - 1) & 63;
        const double pa = __shfl(part_a, src), pb = __shfl(part_b, src);
        const uint32_t oa = (uint32_t)__shfl((int)g.out[0], src), ob = (uint32_t)__shfl((int)g.out[1], src);
        const double v = (lane & 1) ? pb : pa;
        const uint32_t o = (lane & 1) ? ob : oa;
        if ((lane >> 1) < (int)w.n_groups && o < uf.n_pairs) uf.logs_host[o] = log10(v) - uf.log_c32;
        if (under) *uf.flag_host = 1u;
    }
}

template <int C, bool ROW16, bool FAST>
__global__ void __launch_bounds__(64) phmm_fill_pk(const uint32_t *__restrict__ img, const PhGroup2 *__restrict__ groups,
                                                   const PhTab *__restrict__ tabs, const PhWave *__restrict__ waves,
                                                   uint32_t n_waves, const float *__restrict__ lut,
                                                   const float *__restrict__ lut_mis, double *__restrict__ sums, const PhUnderflow uf)
{
    phmm_fill_pk_body<C, ROW16, FAST>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf);
}

// Same fill asked to fit three waves per SIMD (168 VGPRs).  Width 19 -- the tiling of H = 300 -- needs
// 172 left alone and drops to two waves; the two spilled values are touched once per row.
template <int C, bool ROW16, bool FAST>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
phmm_fill_pk_w3(const uint32_t *__restrict__ img, const PhGroup2 *__restrict__ groups, const PhTab *__restrict__ tabs,
                const PhWave *__restrict__ waves, uint32_t n_waves, const float *__restrict__ lut,
                const float *__restrict__ lut_mis, double *__restrict__ sums, const PhUnderflow uf)
{
    phmm_fill_pk_body<C, ROW16, FAST>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf);
}

// ... and two waves per SIMD (256 VGPRs) for widths 31 and 32 (H = 500 in 16 x 32), 257 left alone.
template <int C, bool ROW16, bool FAST>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
phmm_fill_pk_w2(const uint32_t *__restrict__ img, const PhGroup2 *__restrict__ groups, const PhTab *__restrict__ tabs,
                const PhWave *__restrict__ waves, uint32_t n_waves, const float *__restrict__ lut,
                const float *__restrict__ lut_mis, double *__restrict__ sums, const PhUnderflow uf)
{
    phmm_fill_pk_body<C, ROW16, FAST>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf);
}

template <int C, bool ROW16, bool FAST>
int launch(const uint32_t *img, const PhGroup2 *groups, const PhTab *tabs, const PhWave *waves, uint32_t n_waves,
           const void *lut, const void *lut_mis, double *sums, const PhUnderflow &uf, size_t lds, hipStream_t s)
{
    void (*k)(const uint32_t *, const PhGroup2 *, const PhTab *, const PhWave *, uint32_t, const float *, const float *, double *, PhUnderflow);
    if constexpr (C == kPkThreeWaveWidth)
        k = phmm_fill_pk_w3<C, ROW16, FAST>;
    else if constexpr (C > 30)
        k = phmm_fill_pk_w2<C, ROW16, FAST>;
    else
        k = phmm_fill_pk<C, ROW16, FAST>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -1;
    }
    hipLaunchKernelGGL(k, dim3(n_waves), dim3(64), lds, s, img, groups, tabs, waves, n_waves, (const float *)lut,
                       (const float *)lut_mis, sums, uf);
    return hipGetLastError() == hipSuccess ? 0 : -1;
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
        return all_groups_16 ? (fast ? launch<CC, true, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s) \
                                       : launch<CC, true, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s)) \
                             : (fast ? launch<CC, false, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s) \
                                       : launch<CC, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, uf, lds_bytes, s));
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
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&phmm_fill_pk_w3<kPkThreeWaveWidth, true, true>));
}
