// Smith-Waterman fill in 32-BIT state with the DNA-coded match (round 3): BASELINE config 2 as worded ("int32 affine-gap").
//
// agx_sw_kernel.inc's int32 cell spends three of its 8.5 instructions on the diagonal term -- byte compare, select, add.
// The packed biased fill (agx_sw_pk2_kernel.hip) gets the same term from ONE v_perm_b32 table lookup fused into a
// v_add3_u32, on an image the pack kernel has coded (sw_pack_dna: x as selector bytes, right-aligned; y as shift counts;
// final newlines stripped and put back at the end).  This kernel runs that image and that plan -- the packed plan's lane
// groups of two pairs -- in plain signed 32-bit integers, ONE PAIR AT A TIME: a group fills its first pair, then its
// second, each in full int32 state (z = H + gf, e = clamped vertical gap per column, rising offsets as in the int32 kernel).
//
//   per cell: e' = max3(z_up, e, floor)          v_max3_i32
//             f  = max(z_left, f) (+ ge)         v_max_i32, v_add_u32
//             m  = M_row[code_x]                 v_perm_b32
//             u  = z_diag + (mismatch + |gf| [+ |ge|]) + m      v_add3_u32
//             H' = max3(e', f, u);  z = H' + c   v_max3_i32, v_add_u32
//             best = max3(best, z_j, z_j+1)      half a v_max3_i32
//   7.5 instructions per cell against 8.5; the stripped sentinel also takes config 2's 151 columns to 150 = 4 x 38 lanes'
//   worth without the padding of 4 x 40.
//
// A wave whose pairs did not all pass the pack kernel's DNA test keeps bytes (bit 16 of its class word clear) and runs
// the general 32-bit cell on them.  Scores are bit-identical to every other kernel (tests/test_sw_gpu.py).
#include <type_traits>

#include "agx_sw.h"

namespace {

constexpr uint32_t kRowPadSym = 0x100u; // never equals a byte

__device__ __forceinline__ int shr1(int old, int v)
{
    // DPP wave_shr:1 -- lane i receives lane i-1's v (a group's first lane substitutes the boundary)
    return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);
}

// one pair of a lane group (half h of its SwGroup2 record) in 32-bit state; returns this lane's running maximum at the
// offset of the last step (the caller reduces over the group's lanes and takes the offset off)
template <int C, bool FAST>
__device__ __forceinline__ int i32d_half(const SwParams &prm, const uint32_t *__restrict__ img, uint32_t x_dw, uint32_t y_dw, uint32_t lx_ly,
                                         int steps, int G, int gl, bool active, bool start)
{
    constexpr int XW = (C + 3) / 4;
    const int ge = prm.ge, gf = prm.gf, age = -ge;
    const int s_match = prm.hd, s_mis = prm.hd - prm.delta;
    const int ly = (int)(lx_ly >> 16);
    const int lx = (int)(lx_ly & (FAST ? 0xfffu : 0x7fffu));
    const int nyq = (ly + 3) >> 2;
    const bool feeder = active && start;

    // this lane's C columns: FAST -> one v_perm_b32 selector per column {0x0c, 0x0c, 0x0c, code} (the image holds code
    // 0..3 for a group's first pair, 4 + code for its second, 0x0c for padding: the row table sits in BOTH halves of the
    // permute's 64-bit source, so either picks its byte); general -> the symbol itself
    uint32_t xq[C];
    {
        const uint32_t o = (uint32_t)gl * C, d0 = o >> 2, sh = o & 3u;
        const uint32_t fill = (FAST && !(active && x_dw)) ? 0x0c0c0c0cu : 0u; // a vacant half / an idle lane: all padding
        uint32_t raw[XW + 1];
#pragma unroll
        for (int k = 0; k <= XW; ++k) raw[k] = (active ? img[x_dw + d0 + k] : 0u) | fill;
#pragma unroll
        for (int k = 0; k < XW; ++k) {
            const uint32_t a = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * k + i < C) {
                    uint32_t v = (a >> (8 * i)) & 0xffu;
                    if constexpr (FAST) v |= 0x0c0c0c00u;
                    asm volatile("" : "+v"(v)); // opaque: else the compiler re-extracts the byte in every step
                    xq[4 * k + i] = v;
                }
        }
    }
    const uint32_t *yp = img + y_dw;
    const uint32_t no_row = FAST ? 0x1f1f1f1fu : 0u;
    auto row_quad = [&](int q) -> uint32_t { return (feeder && q < nyq) ? yp[q] : no_row; };

    int floor_t = -age;
    int zb = gf - age;
    int z[C], e[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        z[j] = gf - age;
        e[j] = -2 * age;
    }
    int z_last = gf - age, f_last = gf - age, diag_in = gf - 2 * age, best = gf - age;
    int row_prev = FAST ? 0 : (int)kRowPadSym; // FAST: the row table M (byte c = delta where the row symbol is x's code c); general: the row symbol
    const uint32_t kv = (uint32_t)(prm.delta & 0xff) << 24;

    // FAST: the step after which the group's last lane holds H[lx'][ly'] in its last column (bit 13 of the record: both
    // sequences ended with the sentinel the pack kernel stripped)
    const bool both_nl = FAST && active && gl == G - 1 && ((lx_ly >> 13) & 1u) && lx > 0 && ly > 0;
    const int cap_t = both_nl ? ly + G - 2 : -1;
    int corner = gf - age;

    uint32_t q0 = row_quad(0), q1 = row_quad(1), q2 = row_quad(2);
    uint32_t rows = 0;
    int t = 0;

    auto step = [&]() __attribute__((always_inline)) {
        int fresh;
        if constexpr (FAST)
            fresh = (int)(kv >> (rows & 0xffu)); // shift count 31 (padding, a symbol x lacks): nothing matches
        else
            fresh = (t < ly) ? (int)(rows & 0xffu) : (int)kRowPadSym;
        rows >>= 8;
        int zl = shr1(zb, z_last);
        int fl = shr1(zb, f_last);
        int yc = shr1(fresh, row_prev);
        if (start) {
            zl = zb;
            fl = zb;
            yc = fresh;
        }
        best += age;
        int zd = diag_in;
        diag_in = zl;
        int zleft = zl, f = fl;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int up = z[j];
            const int ev = max(max(up, e[j]), floor_t); // reference P, :313, clamped at 0
            f = max(zleft, f);                          // reference Q, :321
            if (j) f += ge;
            const int lag = j ? 0 : age;
            int s;
            if constexpr (FAST)
                s = zd + (s_mis + lag) + (int)__builtin_amdgcn_perm((uint32_t)yc, (uint32_t)yc, xq[j]); // mismatch, plus delta on a match
            else
                s = zd + ((int)xq[j] == yc ? s_match + lag : s_mis + lag); // H_diag + match / + mismatch, :332
            const int v = max(max(ev, f), s);                                // :333
            const int zn = v + (gf + age);
            e[j] = ev;
            z[j] = zn;
            zd = up;
            zleft = zn;
        }
#pragma unroll
        for (int j = 0; j < C; j += 2) best = j + 1 < C ? max(max(best, z[j]), z[j + 1]) : max(best, z[j]); // :335
        if constexpr (FAST) corner = t == cap_t ? zleft : corner;
        z_last = zleft;
        f_last = f;
        row_prev = yc;
        floor_t += age;
        zb += age;
        ++t;
    };

    const int quads = steps >> 2;
    for (int q = 0; q < quads; ++q) {
        rows = q0;
        q0 = q1;
        q1 = q2;
        q2 = row_quad(q + 3);
#pragma unroll
        for (int b = 0; b < 4; ++b) step();
    }
    rows = q0;
#pragma unroll 1
    while (t < steps) step();

    if constexpr (FAST) {
        // the stripped sentinels: when both sequences ended with one, the two newlines align behind the corner cell; the
        // corner was taken at offset r(cap_t), the maximum stands at r(steps - 1)
        const int match = prm.hd + prm.gf; // hd = match - gf
        if (both_nl) best = max(best, corner + match + (steps - 1 - cap_t) * age);
    }
    return best;
}

template <int C>
__device__ __forceinline__ void i32d_body(const SwParams &prm, const uint32_t *__restrict__ img, const SwGroup2 *__restrict__ groups,
                                          const SwWave w, int32_t *__restrict__ scores)
{
    const int lane = threadIdx.x & 63;
    const int G = w.G;
    const int grp = lane / G;
    const int gl = lane - grp * G;
    const bool active = grp < (int)w.n_groups;
    const bool start = gl == 0;
    SwGroup2 g;
#pragma unroll
    for (int k = 0; k < 2; ++k) g.x_dw[k] = g.y_dw[k] = g.lx_ly[k] = g.out[k] = 0;
    if (active) g = groups[w.first_group + grp];
    const int steps = (int)w.steps;
    int b0, b1;
    if (__builtin_amdgcn_readfirstlane(w.reserved >> 16) & 1u) { // every pair of the wave is DNA-coded (set by the pack kernel)
        b0 = i32d_half<C, true>(prm, img, g.x_dw[0], g.y_dw[0], g.lx_ly[0], steps, G, gl, active, start);
        b1 = i32d_half<C, true>(prm, img, g.x_dw[1], g.y_dw[1], g.lx_ly[1], steps, G, gl, active, start);
    } else {
        b0 = i32d_half<C, false>(prm, img, g.x_dw[0], g.y_dw[0], g.lx_ly[0], steps, G, gl, active, start);
        b1 = i32d_half<C, false>(prm, img, g.x_dw[1], g.y_dw[1], g.lx_ly[1], steps, G, gl, active, start);
    }
    // max over the group's lanes (G need not be a power of two)
    for (int o = 1; o < G; o <<= 1) {
        const int o0 = __shfl_down(b0, o), o1 = __shfl_down(b1, o);
        if (gl + o < G) {
            b0 = max(b0, o0);
            b1 = max(b1, o1);
        }
    }
    const int off = prm.gf + (steps - 1) * -prm.ge; // z of the last step stands r(steps - 1) above H + gf
    // the wave's results move to its first lanes and leave as 8-byte stores of adjacent lanes (see agx_sw_pk2_kernel.hip)
    const int src = (lane * G) & 63;
    const int sa = __shfl(b0 - off, src), sb = __shfl(b1 - off, src);
    const uint32_t oa = (uint32_t)__shfl((int)g.out[0], src), ob = (uint32_t)__shfl((int)g.out[1], src);
    if (lane < (int)w.n_groups) {
        if (ob == oa + 1u && !(oa & 1u) && ob < prm.n_out)
            *reinterpret_cast<int2 *>(scores + oa) = make_int2(sa, sb);
        else {
            scores[oa] = sa;
            if (ob < prm.n_out) scores[ob] = sb;
        }
    }
}

template <int C>
__global__ void __launch_bounds__(256) sw_fill_i32d(const SwParams prm, const uint32_t *__restrict__ img, const SwGroup2 *__restrict__ groups,
                                                    const SwWave *__restrict__ waves, uint32_t n_waves, int32_t *__restrict__ scores)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= n_waves) return;
    i32d_body<C>(prm, img, groups, waves[wave], scores);
}

// mixed batches: ONE launch, every wave reads its class from its record
__global__ void __launch_bounds__(256) sw_fill_i32d_any(const SwParams prm, const uint32_t *__restrict__ img, const SwGroup2 *__restrict__ groups,
                                                        const SwWave *__restrict__ waves, uint32_t n_waves, int32_t *__restrict__ scores)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= n_waves) return;
    const SwWave w = waves[wave];
    switch (__builtin_amdgcn_readfirstlane(w.reserved) & 0xffffu) {
#define AGX_SW_CASE(CC) \
    case CC: i32d_body<CC>(prm, img, groups, w, scores); break;
        AGX_SW_FOR_EACH_CLASS(AGX_SW_CASE)
#undef AGX_SW_CASE
    default: break;
    }
}

} // namespace

int agx_sw_i32d_launch_any(const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves, uint32_t n_waves, int32_t *scores,
                           hipStream_t s)
{
    if (n_waves == 0) return 0;
    hipLaunchKernelGGL(sw_fill_i32d_any, dim3((n_waves + 3) / 4), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int agx_sw_i32d_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves,
                             uint32_t n_waves, int32_t *scores, hipStream_t s)
{
    if (n_waves == 0) return 0;
    const uint32_t blocks = (n_waves + 3) / 4;
    switch (cols_per_lane) {
#define AGX_SW_CASE(CC)                                                                                               \
    case CC:                                                                                                          \
        hipLaunchKernelGGL(sw_fill_i32d<CC>, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores); \
        return hipGetLastError() == hipSuccess ? 0 : -1;
        AGX_SW_FOR_EACH_CLASS(AGX_SW_CASE)
#undef AGX_SW_CASE
    default: return -2;
    }
}

void agx_sw_i32d_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&sw_fill_i32d_any));
}
