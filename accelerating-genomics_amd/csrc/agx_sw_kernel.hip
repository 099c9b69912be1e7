// Smith-Waterman affine-gap score-only fill for gfx950 (wave64).
//
// What it replaces: the per-pair anti-diagonal sweep of
// smithWaterman/antidiagonalSmithWaterman.c:254-347 (P/Q/D recurrences at :313,:321,:332-335,
// boundaries at :290-306).  Same scores, different schedule:
//
//   * A pair is owned by a GROUP of G consecutive lanes of one wavefront (G is chosen on
//     the host per pair, 1..64; floor(64/G) pairs share a wave).  Lane g of the group owns
//     C consecutive columns (C = template parameter) of the shorter sequence and keeps their
//     H (reference D) and E (reference P, vertical gap) cells in VGPRs for the whole fill.
//   * The longer sequence streams through the group one row per step, skewed by one step per
//     lane: at step t lane g fills row t-g of its C columns -- the cells of C consecutive
//     anti-diagonals, i.e. the reference's anti-diagonal wavefront tiled C cells deep.
//   * The left-neighbour dependency (H and F=reference Q of the last column of lane g-1, and
//     the row symbol) moves one lane to the right per step with DPP wave_shr:1; the first
//     lane of every group substitutes the matrix boundary (H=0, F=-inf) and the next row symbol.
//   * Only anti-diagonals d-1 and d-2 are live in the reference (m_get/m_set, :96-184); here
//     that state is h[]/e[] + the three shifted registers.  Nothing is spilled to LDS or HBM.
//
// Padding is by never-matching symbols instead of masks: columns beyond the short sequence
// hold symbol 0x00 (rejected in real input, AGX_E_SYMBOL) and rows before/after the long
// sequence hold 0x100.  With zero boundaries, local alignment scores of padded cells are
// bounded by the real cells they derive from, so max over everything == max over the real
// matrix (tests/test_oracle_sw.py::test_negative_infinity_is_equivalent...).  -inf is a large
// negative constant: P/Q never carry it past the boundary cell.
//
// Roofline: HBM traffic is one read of both sequences + one int32 per pair (~0.0135 B/cell at
// 150x150); the kernel is bound by VALU issue (about 11 integer ops per cell), see DESIGN.md.
#include "agx_sw.h"

#ifndef AGX_SW_UNROLL4
#define AGX_SW_UNROLL4 1
#endif
#ifndef AGX_DPP_MOV
#define AGX_DPP_MOV 0
#endif

namespace {

constexpr int kNegInf = -(1 << 20);
constexpr uint32_t kRowPad = 0x100u; // never equals a byte

__device__ __forceinline__ int shr1(int old, int v)
{
    // DPP wave_shr:1 -- lane i receives lane i-1's v.  Lane 0 has no source; it is always the
    // first lane of a group and substitutes the matrix boundary, so its value is never used.
#if AGX_DPP_MOV
    (void)old;
    return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, false);
#else
    return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);
#endif
}

template <int C>
__global__ void __launch_bounds__(256) sw_fill(const SwParams prm, const uint32_t *__restrict__ img,
                                               const SwGroup *__restrict__ groups,
                                               const SwWave *__restrict__ waves, uint32_t n_waves,
                                               int32_t *__restrict__ scores)
{
    constexpr int XW = (C + 3) / 4; // dwords holding this lane's C symbols
    const int ge = prm.ge, gf = prm.gf, s_match = prm.hd, s_mis = prm.hd - prm.delta;
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

    SwGroup g;
    g.x_dw = g.y_dw = g.lx_ly = g.out = 0;
    if (active) g = groups[w.first_group + grp];
    const int ly = (int)(g.lx_ly >> 16);
    const int nyq = (ly + 3) >> 2;

    // lane gl owns bytes [gl*C, gl*C + C) of the short sequence: fetch the covering dwords and
    // byte-align them (C need not be a multiple of 4; the block is padded so the extra dword exists)
    uint32_t xw[XW];
    {
        const uint32_t o = (uint32_t)gl * C, d0 = o >> 2, sh = o & 3u;
        uint32_t raw[XW + 1];
#pragma unroll
        for (int k = 0; k <= XW; ++k) raw[k] = active ? img[g.x_dw + d0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < XW; ++k) xw[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh);
    }

    const uint32_t *yp = img + g.y_dw;
    auto row_quad = [&](int q) -> uint32_t { return (feeder && q < nyq) ? yp[q] : 0u; };

    // State per owned column: z = H - 4 (what both gap recurrences consume, so it is stored
    // instead of H) and e = reference P.  Two VGPRs per column.
    int z[C], e[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        z[j] = gf;
        e[j] = kNegInf;
    }
    int z_last = gf, f_last = kNegInf, diag_in = gf, best = gf;
    int yc_prev = (int)kRowPad;

    uint32_t q0 = row_quad(0), q1 = row_quad(1), q2 = row_quad(2);
    const int steps = (int)w.steps;
    uint32_t rows = 0;
    int t = 0;

    // one row of the lane's C columns
    auto step = [&]() __attribute__((always_inline)) {
        const int fresh = (t < ly) ? (int)(rows & 0xffu) : (int)kRowPad;
        rows >>= 8;
        int zl = shr1(gf, z_last);
        int fl = shr1(kNegInf, f_last);
        int yc = shr1(fresh, yc_prev);
        if (start) { // column 0: H = 0, Q = -inf (antidiagonalSmithWaterman.c:299-306)
            zl = gf;
            fl = kNegInf;
            yc = fresh;
        }
        int zd = diag_in; // H[r-1][first column - 1] - 4
        diag_in = zl;
        int zleft = zl, f = fl;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int xs = (int)((xw[j >> 2] >> (8 * (j & 3))) & 0xffu);
            const int up = z[j];
            const int ev = max(up, e[j] + ge);              // reference P, :313
            f = max(zleft, f + ge);                         // reference Q, :321
            const int s = zd + (xs == yc ? s_match : s_mis); // H_diag + match / + mismatch, :332
            const int v = max(max(ev, f), max(s, 0));       // :333
            const int zn = v + gf;
            e[j] = ev;
            z[j] = zn;
            zd = up;
            zleft = zn;
            best = max(best, zn); // :335
        }
        z_last = zleft;
        f_last = f;
        yc_prev = yc;
        ++t;
    };

    const int quads = steps >> 2;
    for (int q = 0; q < quads; ++q) { // four rows per packed dword of the long sequence
        rows = q0;
        q0 = q1;
        q1 = q2;
        q2 = row_quad(q + 3);
#if AGX_SW_UNROLL4
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int b = 0; b < 4; ++b) step();
    }
    rows = q0;
#pragma unroll 1
    while (t < steps) step(); // 0..3 remaining rows

    best -= gf;

    // max over the group's lanes (G need not be a power of two)
    for (int o = 1; o < G; o <<= 1) {
        const int other = __shfl_down(best, o);
        if (gl + o < G) best = max(best, other);
    }
    if (feeder) scores[g.out] = best;
}

template <int C>
int launch(const SwParams &prm, const uint32_t *img, const SwGroup *groups, const SwWave *waves, uint32_t n_waves,
           int32_t *scores, hipStream_t s)
{
    const uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(sw_fill<C>, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
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
