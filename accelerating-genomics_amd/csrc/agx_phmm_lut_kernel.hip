// PairHMM forward recurrence in double with the match/mismatch prior LOOKED UP instead of selected
// (gfx950, wave64; built with -ffp-contract=off like agx_phmm_kernel.hip: in AGX_PHMM_F64 every product and sum
// rounds exactly once, in the reference's order -- raw sums stay bit-identical to pairHMM/antidiagsPairHMM.c:120-267).
//
// agx_phmm_kernel.hip spends 3 of its 14 VALU instructions per cell on p() (antidiagsPairHMM.c:111-113): a byte
// compare and two v_cndmask_b32 (a double is two registers).  On plain DNA -- reads of A, C, G, T, N, haplotypes of
// A, C, G, T, checked by the host while it copies the tracks -- the prior of a cell depends on the read row and on
// one of FOUR haplotype letters, so the per-read LDS table ("query profile") carries it ready-made:
//
//     row r of a table = { Qi, Qd, Qg, prior[A], prior[C], prior[T], prior[G] }        (7 doubles = 56 bytes)
//     prior[c] = (read base == c or read base == 'N') ? 1 - Qr : Qr  (Qr/3 with AGX_PHMM_GATK_PRIOR)
//     a table = one neutral row, the read's R rows, one neutral row: a lane clamps its row index (v_med3_i32) instead
//     of walking G-1 stored neutral rows at either end, and every table is as long as ITS read (PhTab.R carries the
//     table's offset / 16 in its upper half) -- mixed regions fit more tables into a wave's 20 KB
//
// and a cell fetches its prior with ONE VALU instruction (v_add_u32_sdwa: row offset + the column's letter code * 8,
// the codes sit pre-scaled in the bytes of the lane's haplotype registers) and one ds_read_b64, which issues on the
// LDS port, not on the VALU: 12 instead of 14 VALU instructions per cell (9 instead of 11 with explicit fma).
// The 56-byte row stride spreads the 16 lanes of a group (consecutive rows) over 32 bank pairs.
// 1 - Qr is formed once per table row by the same subtraction the reference does per cell: same rounding.
//
// Everything else -- lane groups, skew, DPP neighbours, neutral rows, the likelihood summed down the lanes in column
// order -- is agx_phmm_kernel.hip's schedule (see there).  Neutral rows carry prior 0 for every letter: M = 0 * (...)
// keeps the row-0 state exactly as prior 1 or Qr did (mm = 1, 1 - Qg = 0, M_diag = 0).
#include "agx_phmm_dev.h"

#pragma clang fp contract(off)

namespace {

using agx_ph::mad;
using agx_ph::shr1;

__device__ __forceinline__ int rshr1i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ double rshr1(double v)
{
    return __hiloint2double(rshr1i(__double2hiint(v)), rshr1i(__double2loint(v)));
}

constexpr uint32_t kRow = AGX_PH_LUT_ROW_BYTES; // 56

// LDS by byte offset: the written-out add below hands over plain numbers, and `lds + number` would cost an instruction
// per use (the array's own offset is a link-time constant the compiler cannot fold through the assembly)
typedef __attribute__((address_space(3))) const double lds_cdouble;
typedef __attribute__((address_space(3))) unsigned char lds_byte;
__device__ __forceinline__ double lds_double(uint32_t off) { return *reinterpret_cast<lds_cdouble *>(off); }

// base + byte `b` of packed, in one instruction.  Written out because the compiler, left alone, extracts the C bytes
// into C registers of their own before the loop (24 VGPRs more at 32 columns per lane: spills inside the cell loop).
__device__ __forceinline__ uint32_t add_byte(uint32_t base, uint32_t packed, int b)
{
    uint32_t r;
    switch (b & 3) {
    case 0: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(base), "v"(packed)); break;
    case 1: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(base), "v"(packed)); break;
    case 2: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(base), "v"(packed)); break;
    default: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(base), "v"(packed)); break;
    }
    return r;
}

// PHASED (round 3, builds for 16-lane groups): the likelihood sum travels down the lanes only while some lane is at its read's last
// row -- a window of about G steps; the loops before and behind it carry neither the sum's lane shift nor the compare and the
// summing block (agx_phmm_pk_kernel.inc: the same three loops).
// STREAM (round 3): a read whose table would take more than a wave's LDS share (ph_lut_is_ring: more than 365 rows) keeps a ring
// of 256 rows: rows 0 ... 255 at the start, and in step 64 k + 62 the 64 rows of half k + 3 replace those of half k - 1, which no
// lane can reach any more (a lane is at most 63 rows behind the wave's first).  One wave per workgroup: LDS operations stay in
// program order, no barrier.  Tables of shorter reads in the same launch are whole, as without STREAM.
template <int C, bool FMA, bool ROW16, bool PHASED, bool STREAM>
__device__ __forceinline__ void phmm_lut_body(const uint32_t *__restrict__ img, const PhGroup *__restrict__ groups,
                                              const PhTab *__restrict__ tabs, const PhWave *__restrict__ waves, uint32_t n_waves,
                                              const double *__restrict__ lut, const double *__restrict__ lut_mis,
                                              double *__restrict__ sums)
{
    constexpr int HW = (C + 3) / 4; // dwords holding this lane's C haplotype letters
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

    PhGroup g;
    g.hap_dw = g.H = g.R_tab = g.out = 0;
    g.init64 = 0;
    g.init32 = 0;
    if (active) g = groups[w.first_group + grp];
    const int R = (int)(g.R_tab & 0xffffu);
    const int H = (int)g.H;

    // ---- read tables -> LDS
    uint32_t my_tab = 0; // byte offset of this group's table
    uint32_t my_mask = 0xffffffffu; // STREAM: 255 when this group's table is a ring
    // table row `row` (0 and R + 1: neutral) of the read behind tb, written to slot `slot` of its table
    auto build_row = [&](const PhTab tb, uint32_t tab_off, uint32_t row, uint32_t slot) __attribute__((always_inline)) {
        double *tq = reinterpret_cast<double *>(lds + tab_off);
        const unsigned char *rp = reinterpret_cast<const unsigned char *>(img + tb.read_dw);
        const uint32_t trk = ((tb.R + 3u) >> 2) * 4u; // bytes per track
        const int i = (int)row - 1;
        double vi = 0, vd = 0, vg = 1, e0 = 0, e1 = 0, e2 = 0, e3 = 0; // neutral row
        if (i >= 0 && i < (int)tb.R) {
            const uint32_t c = rp[i];
            const double qr = lut[rp[trk + i]];
            const double pm = 1 - qr;                                  // p(): match or N (:111-113)
            const double pq = lut_mis ? lut_mis[rp[trk + i]] : qr;     //      mismatch (Qr/3: AGX_PHMM_GATK_PRIOR)
            vi = lut[rp[2 * trk + i]];
            vd = lut[rp[3 * trk + i]];
            vg = lut[rp[4 * trk + i]];
            const bool any = c == (uint32_t)'N';
            const uint32_t rc = (c >> 1) & 3u; // A 0, C 1, T 2, G 3
            e0 = any || rc == 0u ? pm : pq;
            e1 = any || rc == 1u ? pm : pq;
            e2 = any || rc == 2u ? pm : pq;
            e3 = any || rc == 3u ? pm : pq;
        }
        double *out = tq + 7u * slot;
        out[0] = vi;
        out[1] = vd;
        out[2] = vg;
        out[3] = e0;
        out[4] = e1;
        out[5] = e2;
        out[6] = e3;
    };
    bool any_ring = false; // wave-uniform
    for (uint32_t k = 0; k < w.n_tabs; ++k) {
        PhTab tb = tabs[w.first_tab + k];
        const uint32_t tab_off = (tb.R >> 16) * 16u;
        tb.R &= 0xffffu;
        const bool ring = STREAM && ph_lut_is_ring(tb.R + 2u);
        any_ring = any_ring || ring;
        if (k == (g.R_tab >> 16)) {
            my_tab = tab_off;
            if (ring) my_mask = AGX_PH_LUT_RING_ROWS - 1u;
        }
        const uint32_t rows_now = ring ? AGX_PH_LUT_RING_ROWS : tb.R + 2u;
        for (uint32_t r = lane; r < rows_now; r += 64) build_row(tb, tab_off, r, r);
    }
    __syncthreads();

    // lane gl owns haplotype bytes [gl*C, gl*C + C): fetch the covering dwords, byte-align them (C need not be a
    // multiple of 4; every haplotype is followed by zero slack) and turn every letter into its table offset:
    // ((b >> 1) & 3) * 8 = (b << 2) & 0x18.  Padding bytes (0x00) read letter A's prior: their columns never flow left.
    uint32_t cw[HW];
    {
        const uint32_t o = (uint32_t)gl * C, d0 = o >> 2, sh = o & 3u;
        uint32_t raw[HW + 1];
#pragma unroll
        for (int k = 0; k <= HW; ++k) raw[k] = active ? img[g.hap_dw + d0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < HW; ++k) cw[k] = (__builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh) << 2) & 0x18181818u;
    }

    const double init = g.init64;
    double M[C], X[C], Y[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        M[j] = 0;
        X[j] = 0;
        Y[j] = init;
    }
    double pM = 0, pX = 0, pY = init; // what arrived from the left one step ago = diagonal neighbour
    double acc_prev = 0, result = 0;  // the sum runs down the lanes in column order (reference order, :206-212)
    const int steps = (int)w.steps;
    const int col0 = gl * C;
    // lane gl sits on read row t - gl: table row clamp(t - gl, -1, R) + 1
    int trow = 1 - gl;

    my_tab += (uint32_t)reinterpret_cast<uintptr_t>((lds_byte *)lds); // from here on an LDS address
    auto one_step = [&](int t, auto sum_tag) __attribute__((always_inline)) {
        constexpr bool SUM = decltype(sum_tag)::value;
        if constexpr (STREAM) {
            if (any_ring && (t & 63) == 62) { // (scalar condition) half (t >> 6) + 3 of every ring: one row per lane
                const uint32_t row = ((uint32_t)(t >> 6) + 3u) * 64u + (uint32_t)lane;
                for (uint32_t k = 0; k < w.n_tabs; ++k) {
                    PhTab tb = tabs[w.first_tab + k];
                    const uint32_t tab_off = (tb.R >> 16) * 16u;
                    tb.R &= 0xffffu;
                    if (ph_lut_is_ring(tb.R + 2u) && row <= tb.R + 1u) build_row(tb, tab_off, row, row & (AGX_PH_LUT_RING_ROWS - 1u));
                }
            }
        }
        uint32_t rowidx = (uint32_t)min(max(trow, 0), R + 1); // v_med3_i32
        if constexpr (STREAM) rowidx &= my_mask;
        const uint32_t rowoff = my_tab + rowidx * kRow;
        const double q_i = lds_double(rowoff), q_d = lds_double(rowoff + 8), q_g = lds_double(rowoff + 16);
        const double mm = 1 - (q_i + q_d); // mm() (:115-117)
        const double gm = 1 - q_g;
        const uint32_t priors = rowoff + 24u; // (added here: behind the written-out add the compiler would not fold it into the read's offset)

        double lM, lX, lY, acc = 0; // left neighbours; column 0 of rows >= 1 is all zeros (:168-178)
        if constexpr (ROW16) {
            lM = rshr1(M[C - 1]);
            lX = rshr1(X[C - 1]);
            lY = rshr1(Y[C - 1]);
            if constexpr (SUM) acc = rshr1(acc_prev);
        } else {
            lM = shr1(M[C - 1]);
            lX = shr1(X[C - 1]);
            lY = shr1(Y[C - 1]);
            if constexpr (SUM) acc = shr1(acc_prev);
            if (start) {
                lM = 0;
                lX = 0;
                lY = 0;
                acc = 0;
            }
        }
        const double dM0 = pM, dX0 = pX, dY0 = pY;
        pM = lM;
        pX = lX;
        pY = lY;
        // pass A, right to left: M and X in place -- M[i][j] needs row i-1 of column j-1, which this order has not
        // overwritten yet
#pragma unroll
        for (int j = C - 1; j >= 0; --j) {
            const double prior = lds_double(add_byte(priors, cw[j >> 2], j));
            const double dM = j ? M[j > 0 ? j - 1 : 0] : dM0;
            const double dX = j ? X[j > 0 ? j - 1 : 0] : dX0;
            const double dY = j ? Y[j > 0 ? j - 1 : 0] : dY0;
            const double x = mad<FMA>(M[j], q_i, X[j] * q_g);          // :189
            const double m = prior * mad<FMA>(mm, dM, gm * (dX + dY)); // :184
            X[j] = x;
            M[j] = m;
        }
        // pass B, left to right: Y[i][j] needs the new M and Y of column j-1 (:194)
        double cM = lM, cY = lY;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double y = mad<FMA>(cM, q_d, cY * q_g);
            cM = M[j];
            cY = y;
            Y[j] = y;
        }
        if constexpr (SUM) {
            if (t - gl + 1 == R) { // last read row: likelihood (:206-212), columns in order
                if (col0 + C <= H) {
#pragma unroll
                    for (int j = 0; j < C; ++j) acc += (M[j] + X[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < C; ++j)
                        if (col0 + j < H) acc += (M[j] + X[j]);
                }
                if (gl == G - 1) result = acc;
            }
            acc_prev = acc;
        }
        ++trow;
    };
    if constexpr (PHASED) {
        // wave-uniform window of the steps in which some lane is at its read's last row (inactive lanes: R = 0, none)
        int lo = active ? R + gl - 1 : 0x7fffffff, hi = active ? R + gl - 1 : -1;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, __shfl_xor(lo, d));
            hi = max(hi, __shfl_xor(hi, d));
        }
        lo = min(__builtin_amdgcn_readfirstlane(lo), steps);
        hi = min(__builtin_amdgcn_readfirstlane(hi) + 1, steps);
        int t = 0;
        for (; t < lo; ++t) one_step(t, std::false_type{});
        for (; t < hi; ++t) one_step(t, std::true_type{});
        for (; t < steps; ++t) one_step(t, std::false_type{});
    } else
        for (int t = 0; t < steps; ++t) one_step(t, std::true_type{});
    if (active && gl == G - 1) sums[g.out] = result;
}

template <int C, bool FMA, bool ROW16, bool PHASED, bool STREAM>
__global__ void __launch_bounds__(64) phmm_fill_lut(const uint32_t *__restrict__ img, const PhGroup *__restrict__ groups,
                                                    const PhTab *__restrict__ tabs, const PhWave *__restrict__ waves,
                                                    uint32_t n_waves, const double *__restrict__ lut,
                                                    const double *__restrict__ lut_mis, double *__restrict__ sums)
{
    phmm_lut_body<C, FMA, ROW16, PHASED, STREAM>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums);
}

// the widest classes asked to fit two waves per SIMD (256 VGPRs), as phmm_fill_w2 in agx_phmm_kernel.hip
template <int C, bool FMA, bool ROW16, bool PHASED, bool STREAM>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
phmm_fill_lut_w2(const uint32_t *__restrict__ img, const PhGroup *__restrict__ groups, const PhTab *__restrict__ tabs,
                 const PhWave *__restrict__ waves, uint32_t n_waves, const double *__restrict__ lut,
                 const double *__restrict__ lut_mis, double *__restrict__ sums)
{
    phmm_lut_body<C, FMA, ROW16, PHASED, STREAM>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums);
}

template <int C, bool FMA, bool ROW16, bool PHASED, bool STREAM = false>
int launch(const uint32_t *img, const PhGroup *groups, const PhTab *tabs, const PhWave *waves, uint32_t n_waves, const void *lut,
           const void *lut_mis, double *sums, size_t lds, hipStream_t s)
{
    void (*k)(const uint32_t *, const PhGroup *, const PhTab *, const PhWave *, uint32_t, const double *, const double *, double *);
    if constexpr (C >= AGX_PH_LUT_W2_FROM || (PHASED && C >= 22)) // (the three-loop builds from 22 columns on: 218-276 registers left alone)
        k = phmm_fill_lut_w2<C, FMA, ROW16, PHASED, STREAM>;
    else
        k = phmm_fill_lut<C, FMA, ROW16, PHASED, STREAM>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -1;
    }
    hipLaunchKernelGGL(k, dim3(n_waves), dim3(64), lds, s, img, groups, tabs, waves, n_waves, (const double *)lut,
                       (const double *)lut_mis, sums);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int C>
int launch_mode(bool fma, bool all_g16, bool phased, bool stream, const uint32_t *img, const PhGroup *groups, const PhTab *tabs, const PhWave *waves,
                uint32_t n_waves, const void *lut, const void *lut_mis, double *sums, size_t lds, hipStream_t s)
{
    if constexpr (C > 32)
        return -2; // double classes end at 32 columns per lane
    else {
        if (stream) { // (long reads: the builds for groups of any width serve 16-lane groups too)
            if (fma) return launch<C, true, false, false, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, lds, s);
            return launch<C, false, false, false, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, lds, s);
        }
        if (fma) {
            if (all_g16) return launch<C, true, true, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, lds, s);
            return launch<C, true, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, lds, s);
        }
        // Three loops where they were measured to pay (tools/lut_loops.sh, profiles/r03ap_lut_loops.log): config 5's tiling, 16 lanes
        // x 32 columns in the reference's operation order, 1.610 -> 1.589 ms; with explicit fma the same build LOSES 3 %
        // (1.214 -> 1.254 ms), at 30 columns 0.7 % / 3 %, narrower widths do not move.
        if constexpr (C == 32) {
            if (all_g16 && phased) return launch<C, false, true, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, lds, s);
        }
        if (all_g16) return launch<C, false, true, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, lds, s);
        return launch<C, false, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, lds, s);
    }
}

} // namespace

int agx_phmm_lut_launch_class(bool fma, int cols_per_lane, bool all_groups_16, const uint32_t *img, const PhGroup *groups,
                              const PhTab *tabs, const PhWave *waves, uint32_t n_waves, const void *lut, const void *lut_mis,
                              double *sums, size_t lds_bytes, bool phased, bool stream, hipStream_t s)
{
    if (n_waves == 0) return 0;
#define AGX_PH_CASE(CC) \
    case CC: return launch_mode<CC>(fma, all_groups_16, phased, stream, img, groups, tabs, waves, n_waves, lut, lut_mis, sums, lds_bytes, s);
    switch (cols_per_lane) {
        AGX_PH_FOR_EACH_CLASS(AGX_PH_CASE)
    default: return -2;
    }
#undef AGX_PH_CASE
}

// Loads this file's code object when a batch that will use it is created (see agx_phmm_scalar_preload).
void agx_phmm_lut_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&phmm_fill_lut<16, false, false, false, false>));
}
