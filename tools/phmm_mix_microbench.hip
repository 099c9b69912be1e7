// Does a packed-FMA float PairHMM cell (two pairs per lane) beat the scalar reference-order cell?
// Synthetic cell loops with the real instruction mixes, C columns in registers, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/phmm_mix_microbench.hip -o /tmp/phmm_mix && /tmp/phmm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int C>
__global__ void __launch_bounds__(64) scalar_cell(int steps, const unsigned *sym, float *out)
{
    float M[C], X[C], Y[C];
    unsigned hw[C];
    for (int j = 0; j < C; ++j) { M[j] = 0.f; X[j] = 0.f; Y[j] = 1e30f; hw[j] = sym[(threadIdx.x * C + j) & 1023]; }
    float pM = 0, pX = 0, pY = 1e30f;
    for (int t = 0; t < steps; ++t) {
        const unsigned rc = sym[(t + threadIdx.x) & 1023];
        const float q_r = 1e-3f + 1e-6f * t, q_i = 2e-5f, q_d = 3e-5f, q_g = 0.1f;
        const float pm = 1 - q_r, pq = q_r, mm = 1 - (q_i + q_d), gm = 1 - q_g;
        const float dM0 = pM, dX0 = pX, dY0 = pY;
        pM = M[C - 1]; pX = X[C - 1]; pY = Y[C - 1];
#pragma unroll
        for (int j = C - 1; j >= 0; --j) {
            const float prior = hw[j] == rc ? pm : pq;
            const float dM = j ? M[j ? j - 1 : 0] : dM0, dX = j ? X[j ? j - 1 : 0] : dX0, dY = j ? Y[j ? j - 1 : 0] : dY0;
            const float x = M[j] * q_i + X[j] * q_g;
            const float m = prior * (mm * dM + gm * (dX + dY));
            X[j] = x; M[j] = m;
        }
        float cM = pM * 0.5f, cY = pY * 0.5f;
#pragma unroll
        for (int j = 0; j < C; ++j) { const float y = cM * q_d + cY * q_g; cM = M[j]; cY = y; Y[j] = y; }
    }
    float s = 0;
    for (int j = 0; j < C; ++j) s += M[j] + X[j] + Y[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int C>
__global__ void __launch_bounds__(64) packed_cell(int steps, const unsigned *sym, float *out)
{
    f2 M[C], X[C], Y[C];
    unsigned ha[C], hb[C];
    for (int j = 0; j < C; ++j) { M[j] = f2{0.f, 0.f}; X[j] = f2{0.f, 0.f}; Y[j] = f2{1e30f, 1e30f}; ha[j] = sym[(threadIdx.x * C + j) & 1023]; hb[j] = sym[(threadIdx.x * C + j + 77) & 1023]; }
    f2 pM = f2{0, 0}, pX = f2{0, 0}, pY = f2{1e30f, 1e30f};
    for (int t = 0; t < steps; ++t) {
        const unsigned rc = sym[(t + threadIdx.x) & 1023];
        const float q_r = 1e-3f + 1e-6f * t, q_i = 2e-5f, q_d = 3e-5f, q_g = 0.1f;
        const float pm = 1 - q_r, pq = q_r, mm = 1 - (q_i + q_d), gm = 1 - q_g;
        const f2 qi2 = f2{q_i, q_i}, qg2 = f2{q_g, q_g}, qd2 = f2{q_d, q_d}, mm2 = f2{mm, mm}, gm2 = f2{gm, gm};
        const f2 dM0 = pM, dX0 = pX, dY0 = pY;
        pM = M[C - 1]; pX = X[C - 1]; pY = Y[C - 1];
#pragma unroll
        for (int j = C - 1; j >= 0; --j) {
            f2 prior;
            prior.x = ha[j] == rc ? pm : pq;
            prior.y = hb[j] == rc ? pm : pq;
            const f2 dM = j ? M[j ? j - 1 : 0] : dM0, dX = j ? X[j ? j - 1 : 0] : dX0, dY = j ? Y[j ? j - 1 : 0] : dY0;
            const f2 x = __builtin_elementwise_fma(M[j], qi2, X[j] * qg2);
            const f2 m = prior * __builtin_elementwise_fma(mm2, dM, gm2 * (dX + dY));
            X[j] = x; M[j] = m;
        }
        f2 cM = pM * 0.5f, cY = pY * 0.5f;
#pragma unroll
        for (int j = 0; j < C; ++j) { const f2 y = __builtin_elementwise_fma(cM, qd2, cY * qg2); cM = M[j]; cY = y; Y[j] = y; }
    }
    f2 s = f2{0, 0};
    for (int j = 0; j < C; ++j) s += M[j] + X[j] + Y[j];
    out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
}

template <typename K> void run(const char *name, K k, int C, int cells_per_lane_step, int waves)
{
    unsigned *sym; float *out;
    CHECK(hipMalloc(&sym, 4096)); CHECK(hipMalloc(&out, 4 * 64 * waves));
    unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = "ACGT"[(i * 7 + i / 3) & 3];
    CHECK(hipMemcpy(sym, h, 4096, hipMemcpyHostToDevice));
    const int steps = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, 10, sym, out); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, steps, sym, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double cells = (double)waves * 64 * steps * cells_per_lane_step;
    printf("%-28s C=%2d waves %6d: %.3f ms  %.3f ps/cell  (%.2f T cells/s)\n", name, C, waves, best, best * 1e9 / cells, cells / best / 1e9);
    CHECK(hipFree(sym)); CHECK(hipFree(out));
}

int main()
{
    for (int waves : {2048, 8192}) {
        run("scalar order-exact", scalar_cell<38>, 38, 38, waves);
        run("scalar order-exact", scalar_cell<30>, 30, 30, waves);
        run("packed FMA (2 pairs/lane)", packed_cell<30>, 30, 60, waves);
        run("packed FMA (2 pairs/lane)", packed_cell<24>, 24, 48, waves);
        run("packed FMA (2 pairs/lane)", packed_cell<20>, 20, 40, waves);
    }
    return 0;
}
