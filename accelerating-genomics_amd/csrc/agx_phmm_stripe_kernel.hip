// PairHMM forward recurrence for haplotypes wider than one wave can hold in registers
// (more than 64 lanes x the widest class): the haplotype is cut into STRIPES of 64*C columns that
// one wavefront fills one after the other.  Built with -ffp-contract=off (see agx_phmm_kernel.hip,
// whose recurrence, padding rules and reference citations apply unchanged).
//
// What differs from phmm_fill:
//   * One pair per wavefront (G = 64), double arithmetic only; a workgroup walks the plan's pairs
//     grid-stride so that its boundary scratch is reused.  Two waves per SIMD are requested
//     (amdgpu_waves_per_eu): the stripe bookkeeping would otherwise push the kernel past 256 VGPRs
//     and halve the occupancy, which costs more than the few spills outside the cell loop.
//   * Between stripes the last column of stripe s (M, X, Y of every read row) is the column-0
//     input of stripe s+1.  Lane 63 stores its three values every step to a per-workgroup
//     scratch in HBM (24 B/step, indexed by the step that produced them); the next stripe loads
//     them 64 rows at a time, one row per lane, and hands lane 0 the row it needs with
//     v_readlane -- no per-step global load sits on the recurrence's critical path.
//   * The likelihood sum keeps running down the lanes and across stripes in column order, so the
//     AGX_PHMM_F64 result stays bit-identical to the reference for any haplotype length.
//
// Bound: VALU issue, like phmm_fill; the scratch adds 48 B of HBM traffic per read row and stripe.
#include "agx_phmm_dev.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace {

using namespace agx_ph;

template <int C, bool FMA, bool PROBS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
phmm_fill_striped(const uint32_t *__restrict__ img, const PhGroup *__restrict__ groups, const PhTab *__restrict__ tabs,
                  const PhWave *__restrict__ waves, uint32_t n_waves, const double *__restrict__ lut,
                  const double *__restrict__ lut_mis, int mis_div, double *__restrict__ sums, double *__restrict__ scratch,
                  uint32_t scratch_rows, int negate)
{
    constexpr int G = 64;
    constexpr int HW = (C + 3) / 4;
    constexpr int STRIPE = G * C;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    const bool start = lane == 0;
    // two boundary buffers of {M, X, Y}[scratch_rows], ping-ponged between stripes
    double *const bnd = scratch + (size_t)blockIdx.x * 6u * scratch_rows;

    for (uint32_t wave = blockIdx.x; wave < n_waves; wave += gridDim.x) {
        const PhWave w = waves[wave];
        const PhGroup g = groups[w.first_group];
        const int R = (int)(g.R_tab & 0xffffu);
        const int H = (int)g.H;
        const int steps = (int)w.steps; // R + 63
        const uint32_t rows = w.steps + (uint32_t)G - 1u;
        const bool mis_col = lut_mis != nullptr;
        const uint32_t ncol = mis_col ? 5u : 4u;

        __syncthreads(); // the previous pair is done with the table
        build_read_tables<double, PROBS>(lds, img, tabs, w.first_tab, 1u, rows, G, lane, lut, lut_mis);
        __syncthreads();

        const uint32_t mis_off = mis_col ? 4u * rows : 0u;
        const double *tq = reinterpret_cast<const double *>(lds) + (G - 1 - lane);
        const unsigned char *tc = lds + ncol * rows * sizeof(double) + (G - 1 - lane);
        const double init = g.init64;
        const int n_stripes = (H + STRIPE - 1) / STRIPE;
        double carry = 0, result = 0; // likelihood sum over the columns of earlier stripes

        for (int s = 0; s < n_stripes; ++s) {
            const double *in = bnd + (size_t)(s & 1) * 3u * scratch_rows;
            double *out = bnd + (size_t)((s + 1) & 1) * 3u * scratch_rows;
            const int col0 = s * STRIPE + lane * C;

            uint32_t hw[HW];
            {
                const uint32_t o = (uint32_t)col0, d0 = o >> 2, sh = o & 3u;
                uint32_t raw[HW + 1];
#pragma unroll
                for (int k = 0; k <= HW; ++k) raw[k] = img[g.hap_dw + d0 + k];
#pragma unroll
                for (int k = 0; k < HW; ++k) hw[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh);
            }
            unsigned long long nmask = 0;
#pragma unroll
            for (int j = 0; j < C; ++j)
                nmask |= (((hw[j >> 2] >> (8 * (j & 3))) & 0xffu) == (uint32_t)'N' ? 1ull : 0ull) << j;

            double M[C], X[C], Y[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                M[j] = 0;
                X[j] = 0;
                Y[j] = init;
            }
            double pM = 0, pX = 0, pY = init;
            double acc_prev = 0;
            double inM = 0, inX = 0, inY = 0; // this lane's row of the current 64-row boundary block

            auto fill = [&](auto hapn_tag) {
                constexpr bool HAPN = decltype(hapn_tag)::value;
                for (int t = 0; t < steps; ++t) {
                    const int k = t & 63;
                    if (k == 0 && s > 0) {
                        // rows t .. t+63 of the previous stripe's last column; lane 63 produced row r at step r + 63
                        const uint32_t idx = (uint32_t)t + 63u + (uint32_t)lane;
                        inM = in[idx];
                        inX = in[scratch_rows + idx];
                        inY = in[2u * scratch_rows + idx];
                    }
                    const double q_r = tq[t], q_i = tq[rows + t], q_d = tq[2 * rows + t], q_g = tq[3 * rows + t];
                    const double q_m = mis_div ? q_r / 3.0 : tq[mis_off + t]; // (the host's table: d[c] / 3.0, agx_phmm.cpp build_lut)
                    const uint32_t rc = tc[t];
                    const double pm = 1 - q_r;
                    const double pq = rc == (uint32_t)'N' ? pm : q_m;
                    const double mm = 1 - (q_i + q_d);
                    const double gm = 1 - q_g;

                    double lM = shr1(M[C - 1]), lX = shr1(X[C - 1]), lY = shr1(Y[C - 1]);
                    double acc = shr1(acc_prev);
                    const double bM = lane_value(inM, k), bX = lane_value(inX, k), bY = lane_value(inY, k);
                    if (start) { // column 0 of the matrix (zeros, :168-178) or the previous stripe's last column
                        lM = bM;
                        lX = bX;
                        lY = bY;
                        acc = carry;
                    }
                    const double dM0 = pM, dX0 = pX, dY0 = pY;
                    pM = lM;
                    pX = lX;
                    pY = lY;
#pragma unroll
                    for (int j = C - 1; j >= 0; --j) {
                        const uint32_t hc = (hw[j >> 2] >> (8 * (j & 3))) & 0xffu;
                        bool match = hc == rc;
                        if constexpr (HAPN) match = match || ((nmask >> j) & 1ull);
                        const double prior = match ? pm : pq;
                        const double dM = j ? M[j > 0 ? j - 1 : 0] : dM0;
                        const double dX = j ? X[j > 0 ? j - 1 : 0] : dX0;
                        const double dY = j ? Y[j > 0 ? j - 1 : 0] : dY0;
                        const double x = mad<FMA>(M[j], q_i, X[j] * q_g);
                        const double m = prior * mad<FMA>(mm, dM, gm * (dX + dY));
                        X[j] = x;
                        M[j] = m;
                    }
                    double cM = lM, cY = lY;
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        const double y = mad<FMA>(cM, q_d, cY * q_g);
                        cM = M[j];
                        cY = y;
                        Y[j] = y;
                    }
                    if (lane == G - 1 && s + 1 < n_stripes) { // boundary for the next stripe
                        out[t] = M[C - 1];
                        out[scratch_rows + t] = X[C - 1];
                        out[2u * scratch_rows + t] = Y[C - 1];
                    }
                    if (t - lane + 1 == R) {
                        if (col0 + C <= H) { // interior lane: no per-column masks
#pragma unroll
                            for (int j = 0; j < C; ++j) acc += (M[j] + X[j]);
                        } else {
#pragma unroll
                            for (int j = 0; j < C; ++j)
                                if (col0 + j < H) acc += (M[j] + X[j]);
                        }
                        if (lane == G - 1) result = acc;
                    }
                    acc_prev = acc;
                }
            };
            if (__any(nmask != 0))
                fill(std::true_type{});
            else
                fill(std::false_type{});

            carry = lane_value(result, G - 1);
            __syncthreads(); // lane 63's stores are visible to the whole wave before the next stripe loads them
        }
        if (lane == G - 1) sums[g.out] = negate ? -result : result;
    }
}

template <int C, bool FMA, bool PROBS>
int launch(const uint32_t *img, const PhGroup *groups, const PhTab *tabs, const PhWave *waves, uint32_t n_waves, uint32_t grid,
           const double *lut, const double *lut_mis, int mis_div, double *sums, double *scratch, uint32_t scratch_rows, int negate,
           size_t lds, hipStream_t s)
{
    auto k = phmm_fill_striped<C, FMA, PROBS>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -1;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, s, img, groups, tabs, waves, n_waves, lut, lut_mis, mis_div, sums, scratch,
                       scratch_rows, negate);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int C>
int launch_mode(int mode, const uint32_t *img, const PhGroup *groups, const PhTab *tabs, const PhWave *waves, uint32_t n_waves,
                uint32_t grid, const double *l, const double *lm, int mis_div, double *sums, double *scratch, uint32_t scratch_rows,
                int negate, size_t lds_bytes, hipStream_t s)
{
    switch (mode) {
    case 0: return launch<C, false, false>(img, groups, tabs, waves, n_waves, grid, l, lm, mis_div, sums, scratch, scratch_rows, negate, lds_bytes, s);
    case 1: return launch<C, true, false>(img, groups, tabs, waves, n_waves, grid, l, lm, mis_div, sums, scratch, scratch_rows, negate, lds_bytes, s);
    case 4: return launch<C, false, true>(img, groups, tabs, waves, n_waves, grid, l, lm, mis_div, sums, scratch, scratch_rows, negate, lds_bytes, s);
    default: return -2;
    }
}

} // namespace

int agx_phmm_stripe_launch(int mode, int cols_per_lane, const uint32_t *img, const PhGroup *groups, const PhTab *tabs, const PhWave *waves,
                           uint32_t n_waves, uint32_t grid, const void *lut, const void *lut_mis, int mis_div, double *sums, double *scratch,
                           uint32_t scratch_rows, int negate, size_t lds_bytes, hipStream_t s)
{
    if (n_waves == 0) return 0;
    const double *l = (const double *)lut, *lm = (const double *)lut_mis;
#define AGX_PH_CASE(CC) \
    case CC: return launch_mode<CC>(mode, img, groups, tabs, waves, n_waves, grid, l, lm, mis_div, sums, scratch, scratch_rows, negate, lds_bytes, s);
    switch (cols_per_lane) {
        AGX_PH_FOR_EACH_STRIPE_CLASS(AGX_PH_CASE)
    default: return -2;
    }
#undef AGX_PH_CASE
}
