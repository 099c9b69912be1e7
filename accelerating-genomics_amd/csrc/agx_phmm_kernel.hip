// PairHMM forward recurrence for gfx950 (wave64).  Built with -ffp-contract=off: in the
// reference-order modes every product and sum below rounds exactly once, like the
// reference compiled for baseline x86-64; the FMA mode contracts explicitly.
//
// What it replaces: pairHMM() of pairHMM/antidiagsPairHMM.c:120-267 (same arithmetic as
// pairHMM/pairHMMmatrix.c:41-66):
//     M[i][j] = p(R[i-1],H[j-1],Qr[i-1]) * (mm(Qi,Qd)*M[i-1][j-1] + (1-Qg)*(X[i-1][j-1]+Y[i-1][j-1]))
//     X[i][j] = M[i-1][j]*Qi + X[i-1][j]*Qg            Y[i][j] = M[i][j-1]*Qd + Y[i][j-1]*Qg
//     sum = SUM_j (M[R][j] + X[R][j])     (row 0: M=X=0, Y=MAX/16/H; column 0 of rows>=1: 0)
//
// Schedule:
//   * One (read, haplotype) pair per GROUP of G lanes (G chosen on the host, 1..64); lane g owns
//     C haplotype columns (template parameter) and keeps their M/X/Y of the previous read row in
//     VGPRs.  Read rows stream through the group skewed one step per lane -- the reference's
//     anti-diagonal wavefront tiled C cells deep.  Left/diagonal neighbours of a lane's first
//     column arrive from lane g-1 by DPP wave_shr:1; Y's in-row dependency is in-lane.
//   * The "query profile" lives in LDS: per read one table {Qr,Qi,Qd,Qg as probabilities (94-entry
//     pow(10,-q/10) LUT computed by the host libm, antidiagsPairHMM.c:104-107), base}, built once
//     per wave and shared by all its groups that use the read.  Lane g reads row t-g each step.
//   * No masks in the cell loop.  Rows a lane visits before its first / after its last real row
//     are NEUTRAL table rows (Qi=Qd=0, Qg=1, so mm=1 and 1-Qg=0): they reproduce the row-0
//     state (M=0, X=0, Y=init) exactly, so a lane that starts late finds the boundary it needs.
//     Columns beyond H compute garbage that never flows left, and are skipped in the final sum.
//   * The sum over the last row runs down the lanes in column order (one add per column, the
//     reference's order), so in AGX_PHMM_F64 the raw sum is bit-identical to the reference.
//
// Bound: VALU issue (11 flops/cell in reference order, 8 with FMA); HBM traffic is the inputs
// once per wave plus one double per pair.  See DESIGN.md.
#include "agx_phmm.h"

#include <type_traits>


#pragma clang fp contract(off)

namespace {

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

// DPP row_shr:1 with bound_ctrl: lane i of every 16-lane row receives lane i-1's v, a row's first lane 0
__device__ __forceinline__ int rshr1i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ double rshr1(double v)
{
    return __hiloint2double(rshr1i(__double2hiint(v)), rshr1i(__double2loint(v)));
}

template <bool FMA> __device__ __forceinline__ double mad(double a, double b, double c)
{
    if constexpr (FMA) return __builtin_fma(a, b, c);
    return a * b + c; // two roundings (file is built with contraction off)
}
template <bool FMA> __device__ __forceinline__ float mad(float a, float b, float c)
{
    if constexpr (FMA) return __builtin_fmaf(a, b, c);
    return a * b + c;
}

// ROW16: every wave of the launch has groups of exactly 16 lanes (uniform batches such as H = 500 in
// 16 x 32): the groups coincide with the DPP rows, and the row shift's zero fill is the column-0 boundary.
template <typename T, int C, bool FMA, bool RESCUE, bool PROBS, bool ROW16>
__device__ __forceinline__ void phmm_fill_body(const uint32_t *__restrict__ img, const PhGroup *__restrict__ groups,
                                               const PhTab *__restrict__ tabs, const PhWave *__restrict__ waves,
                                               uint32_t n_waves, const T *__restrict__ lut, const T *__restrict__ lut_mis,
                                               double *__restrict__ sums, double rescue_below,
                                               unsigned long long *__restrict__ n_rescued)
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

    PhGroup g;
    g.hap_dw = g.H = g.R_tab = g.out = 0;
    g.init64 = 0;
    g.init32 = 0;
    if (active) g = groups[w.first_group + grp];
    const int R = (int)(g.R_tab & 0xffffu);
    const int H = (int)g.H;

    bool wanted = active;
    if constexpr (RESCUE) {
        // second pass over an fp32 result: only pairs that underflowed are recomputed in double
        const double prev = active ? sums[g.out] : 1.0;
        wanted = active && !(prev >= rescue_below);
        if (!__any(wanted)) return;
    }

    // ---- read tables -> LDS
    const uint32_t rows = w.steps + (uint32_t)G - 1u;
    const bool mis_col = lut_mis != nullptr; // wave-uniform: a fifth table column holds the mismatch prior
    const uint32_t ncol = mis_col ? 5u : 4u;
    const size_t tab_bytes = ph_tab_bytes(sizeof(T) == 8, mis_col, rows);
    for (uint32_t k = 0; k < w.n_tabs; ++k) {
        const PhTab tb = tabs[w.first_tab + k];
        T *tq = reinterpret_cast<T *>(lds + k * tab_bytes);
        unsigned char *tc = reinterpret_cast<unsigned char *>(tq + ncol * rows);
        const unsigned char *rp = reinterpret_cast<const unsigned char *>(img + tb.read_dw);
        const uint32_t trk = ((tb.R + 3u) >> 2) * 4u; // bytes per track
        for (uint32_t r = lane; r < rows; r += 64) {
            const int i = (int)r - (G - 1);
            T vr = 0, vi = 0, vd = 0, vg = 1, vm = 0; // neutral row
            unsigned char c = 0;
            if (i >= 0 && i < (int)tb.R) {
                if constexpr (PROBS) { // pairHMM() seam: tracks are probabilities, bases follow them
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
    __syncthreads();

    const uint32_t mis_off = mis_col ? 4u * rows : 0u; // branch-free: without the column, re-read Qr
    const uint32_t tabi = g.R_tab >> 16;
    const T *tq = reinterpret_cast<const T *>(lds + tabi * tab_bytes) + (G - 1 - gl);
    const unsigned char *tc = reinterpret_cast<const unsigned char *>(lds + tabi * tab_bytes + ncol * rows * sizeof(T)) + (G - 1 - gl);

    // lane gl owns haplotype bytes [gl*C, gl*C + C): fetch the covering dwords and byte-align them
    // (C need not be a multiple of 4; every haplotype is followed by zero slack)
    uint32_t hw[HW];
    {
        const uint32_t o = (uint32_t)gl * C, d0 = o >> 2, sh = o & 3u;
        uint32_t raw[HW + 1];
#pragma unroll
        for (int k = 0; k <= HW; ++k) raw[k] = active ? img[g.hap_dw + d0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < HW; ++k) hw[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh);
    }
    // haplotype 'N' matches every read base (p(), :111-113).  It is rare, so the cell loop exists
    // twice: without the test when no lane of the wave holds an 'N', with a per-column bit otherwise.
    unsigned long long nmask = 0;
#pragma unroll
    for (int j = 0; j < C; ++j)
        nmask |= (((hw[j >> 2] >> (8 * (j & 3))) & 0xffu) == (uint32_t)'N' ? 1ull : 0ull) << j;

    const T init = sizeof(T) == 8 ? (T)g.init64 : (T)g.init32;
    T M[C], X[C], Y[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        M[j] = 0;
        X[j] = 0;
        Y[j] = init;
    }
    T pM = 0, pX = 0, pY = init; // what arrived from the left one step ago = diagonal neighbour
    // double: the sum runs down the lanes in column order (reference order).  float: every lane sums
    // its columns of the last row in double and the lanes are combined after the loop -- a float
    // running sum over thousands of columns would lose the 1e-6 the mode promises.
    constexpr bool CHAIN = sizeof(T) == 8;
    T acc_prev = 0;
    double result = 0, part = 0;
    const int steps = (int)w.steps;
    const int col0 = gl * C;

    auto fill = [&](auto hapn_tag) {
        constexpr bool HAPN = decltype(hapn_tag)::value;
        for (int t = 0; t < steps; ++t) {
            const T q_r = tq[t], q_i = tq[rows + t], q_d = tq[2 * rows + t], q_g = tq[3 * rows + t];
            const T q_m = tq[mis_off + t]; // mismatch prior: the reference's is Qr itself (mis_off = 0)
            const uint32_t rc = tc[t];
            const T pm = 1 - q_r;                        // p(): match or N (:111-113)
            const T pq = rc == (uint32_t)'N' ? pm : q_m; //      mismatch
            const T mm = 1 - (q_i + q_d);                // mm() (:115-117)
            const T gm = 1 - q_g;

            T lM, lX, lY, acc = 0; // left neighbours; column 0 of rows >= 1 is all zeros (:168-178)
            if constexpr (ROW16) {
                lM = rshr1(M[C - 1]);
                lX = rshr1(X[C - 1]);
                lY = rshr1(Y[C - 1]);
                if constexpr (CHAIN) acc = rshr1(acc_prev);
            } else {
                lM = shr1(M[C - 1]);
                lX = shr1(X[C - 1]);
                lY = shr1(Y[C - 1]);
                if constexpr (CHAIN) acc = shr1(acc_prev);
                if (start) {
                    lM = 0;
                    lX = 0;
                    lY = 0;
                    acc = 0;
                }
            }
            const T dM0 = pM, dX0 = pX, dY0 = pY;
            pM = lM;
            pX = lX;
            pY = lY;
            // pass A, right to left: M and X in place -- M[i][j] needs row i-1 of column j-1, which
            // this order has not overwritten yet, so no value has to be copied aside.
#pragma unroll
            for (int j = C - 1; j >= 0; --j) {
                const uint32_t hc = (hw[j >> 2] >> (8 * (j & 3))) & 0xffu;
                bool match = hc == rc;
                if constexpr (HAPN) match = match || ((nmask >> j) & 1ull);
                const T prior = match ? pm : pq;
                const T dM = j ? M[j > 0 ? j - 1 : 0] : dM0;
                const T dX = j ? X[j > 0 ? j - 1 : 0] : dX0;
                const T dY = j ? Y[j > 0 ? j - 1 : 0] : dY0;
                const T x = mad<FMA>(M[j], q_i, X[j] * q_g);           // :189
                const T m = prior * mad<FMA>(mm, dM, gm * (dX + dY)); // :184
                X[j] = x;
                M[j] = m;
            }
            // pass B, left to right: Y[i][j] needs the new M and Y of column j-1 (:194)
            T cM = lM, cY = lY;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const T y = mad<FMA>(cM, q_d, cY * q_g);
                cM = M[j];
                cY = y;
                Y[j] = y;
            }
            if (t - gl + 1 == R) { // last read row: likelihood (:206-212), columns in order
                if constexpr (CHAIN) {
                    // (the lanes active in this block share gl: the test is uniform in practice and
                    // interior lanes skip the per-column masks)
                    if (col0 + C <= H) {
#pragma unroll
                        for (int j = 0; j < C; ++j) acc += (M[j] + X[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < C; ++j)
                            if (col0 + j < H) acc += (M[j] + X[j]);
                    }
                    if (gl == G - 1) result = acc;
                } else {
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        if (col0 + j < H) part += (double)(M[j] + X[j]);
                        // keep the conversions next to their adds: hoisted together they would hold
                        // 2C extra registers live and cost a wave of occupancy
                        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (CHAIN) acc_prev = acc;
        }
    };
    if (__any(nmask != 0))
        fill(std::true_type{});
    else
        fill(std::false_type{});

    if constexpr (!CHAIN) { // inclusive scan over the group's lanes: its last lane ends with the total
        for (int dlt = 1; dlt < G; dlt <<= 1) {
            const double v = __shfl_up(part, dlt);
            if (gl >= dlt) part += v;
        }
        result = part;
    }
    if (wanted && gl == G - 1) {
        // the rescue pass stores its (double-scaled) sum negated so the host can tell the scalings apart
        sums[g.out] = RESCUE ? -(double)result : (double)result;
        if constexpr (RESCUE) atomicAdd(n_rescued, 1ull);
    }
}

template <typename T, int C, bool FMA, bool RESCUE, bool PROBS, bool ROW16>
__global__ void __launch_bounds__(64) phmm_fill(const uint32_t *__restrict__ img, const PhGroup *__restrict__ groups,
                                                const PhTab *__restrict__ tabs, const PhWave *__restrict__ waves,
                                                uint32_t n_waves, const T *__restrict__ lut, const T *__restrict__ lut_mis,
                                                double *__restrict__ sums, double rescue_below,
                                                unsigned long long *__restrict__ n_rescued)
{
    phmm_fill_body<T, C, FMA, RESCUE, PROBS, ROW16>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued);
}

// The same fill asked to fit two waves per SIMD (256 VGPRs).  In double, 32 columns per lane -- the
// width that tiles H = 500 over 16 lanes, four pairs per wave -- need 267 registers left alone and drop
// to one wave; with the limit the allocator spills 22 values, all outside the cell loop.
template <typename T, int C, bool FMA, bool RESCUE, bool PROBS, bool ROW16>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
phmm_fill_w2(const uint32_t *__restrict__ img, const PhGroup *__restrict__ groups, const PhTab *__restrict__ tabs,
             const PhWave *__restrict__ waves, uint32_t n_waves, const T *__restrict__ lut, const T *__restrict__ lut_mis,
             double *__restrict__ sums, double rescue_below, unsigned long long *__restrict__ n_rescued)
{
    phmm_fill_body<T, C, FMA, RESCUE, PROBS, ROW16>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued);
}

template <typename T, int C, bool FMA, bool RESCUE, bool PROBS, bool ROW16 = false>
int launch(const uint32_t *img, const PhGroup *groups, const PhTab *tabs, const PhWave *waves, uint32_t n_waves,
           const void *lut, const void *lut_mis, double *sums, double rescue_below, unsigned long long *n_rescued, size_t lds,
           hipStream_t s)
{
    void (*k)(const uint32_t *, const PhGroup *, const PhTab *, const PhWave *, uint32_t, const T *, const T *, double *, double,
              unsigned long long *);
    if constexpr (sizeof(T) == 8 && C == 32)
        k = phmm_fill_w2<T, C, FMA, RESCUE, PROBS, ROW16>;
    else
        k = phmm_fill<T, C, FMA, RESCUE, PROBS, ROW16>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -1;
    }
    hipLaunchKernelGGL(k, dim3(n_waves), dim3(64), lds, s, img, groups, tabs, waves, n_waves, (const T *)lut, (const T *)lut_mis, sums,
                       rescue_below, n_rescued);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int C>
int launch_mode(int mode, bool all_g16, const uint32_t *img, const PhGroup *groups, const PhTab *tabs, const PhWave *waves,
                uint32_t n_waves, const void *lut, const void *lut_mis, double *sums, double rescue_below,
                unsigned long long *n_rescued, size_t lds, hipStream_t s)
{
    if constexpr (C > 32) { // wider than 32 columns only exists in float (VGPR budget)
        if (mode == 2) return launch<float, C, false, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
        // the double rescue pass of a float batch reuses the float batch's records (rare, may spill)
        if (mode == 3) return launch<double, C, false, true, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
        return -2;
    } else
    switch (mode) {
    case 0:
        if (all_g16) return launch<double, C, false, false, false, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
        return launch<double, C, false, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
    case 1:
        if (all_g16) return launch<double, C, true, false, false, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
        return launch<double, C, true, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
    case 2: return launch<float, C, false, false, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
    case 3: return launch<double, C, false, true, false>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
    case 4: return launch<double, C, false, false, true>(img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds, s);
    default: return -2;
    }
}

} // namespace

int agx_phmm_launch_class(int mode, int cols_per_lane, bool all_groups_16, const uint32_t *img, const PhGroup *groups, const PhTab *tabs,
                          const PhWave *waves, uint32_t n_waves, const void *lut, const void *lut_mis, double *sums,
                          double rescue_below, unsigned long long *n_rescued, size_t lds_bytes, hipStream_t s)
{
    if (n_waves == 0) return 0;
#define AGX_PH_CASE(CC) \
    case CC: return launch_mode<CC>(mode, all_groups_16, img, groups, tabs, waves, n_waves, lut, lut_mis, sums, rescue_below, n_rescued, lds_bytes, s);
    switch (cols_per_lane) {
        AGX_PH_FOR_EACH_CLASS(AGX_PH_CASE)
    default: return -2;
    }
#undef AGX_PH_CASE
}

// Loads this file's code object now: the first launch of a kernel otherwise pays for it (1-2 ms in a fresh process --
// inside hipvers' launch -> scores window).  Called when a batch that will use these kernels is created.
void agx_phmm_scalar_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&phmm_fill_w2<double, 32, false, false, false, true>));
}
