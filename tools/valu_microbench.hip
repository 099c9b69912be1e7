// VALU issue-rate microbenchmark for gfx950: how many shader cycles one SIMD spends per
// wave64 instruction, for the opcodes the SW / PairHMM fills are made of, at 1..8 waves/SIMD.
// This is the measured "VALU roofline" DESIGN.md prices the kernels against.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o gpurun_out/valu_microbench && ./gpurun_out/valu_microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// 8 independent accumulators, one instruction each; the block is repeated 4x per loop trip.
#define OP8_3(ins)                                                                                                   \
    asm volatile(ins " %0, %0, %8\n\t" ins " %1, %1, %8\n\t" ins " %2, %2, %8\n\t" ins " %3, %3, %8\n\t" ins       \
                     " %4, %4, %8\n\t" ins " %5, %5, %8\n\t" ins " %6, %6, %8\n\t" ins " %7, %7, %8"                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                    \
                 : "v"(b))
#define OP8_4(ins)                                                                                                   \
    asm volatile(ins " %0, %0, %8, %8\n\t" ins " %1, %1, %8, %8\n\t" ins " %2, %2, %8, %8\n\t" ins                  \
                     " %3, %3, %8, %8\n\t" ins " %4, %4, %8, %8\n\t" ins " %5, %5, %8, %8\n\t" ins                  \
                     " %6, %6, %8, %8\n\t" ins " %7, %7, %8, %8"                                                    \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                    \
                 : "v"(b))

template <typename T, int OP>
__global__ void __launch_bounds__(256) bench(int iters, T *out, unsigned long long *cyc)
{
    T a0 = (T)threadIdx.x, a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3, a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6,
      a7 = a0 + (T)7;
    T b = (T)(blockIdx.x & 3) + (T)1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (OP == 0) OP8_3("v_add_u32");
            if constexpr (OP == 1) OP8_3("v_max_i32");
            if constexpr (OP == 2) OP8_4("v_max3_i32");
            if constexpr (OP == 3) OP8_3("v_pk_max_i16");
            if constexpr (OP == 4) OP8_3("v_pk_add_i16");
            if constexpr (OP == 5) OP8_3("v_add_f32");
            if constexpr (OP == 6) OP8_3("v_mul_f32");
            if constexpr (OP == 7) OP8_4("v_fma_f32");
            if constexpr (OP == 8) OP8_4("v_pk_fma_f32");
            if constexpr (OP == 9) OP8_3("v_pk_mul_f32");
            if constexpr (OP == 10) OP8_3("v_pk_add_f32");
            if constexpr (OP == 11) OP8_3("v_add_f64");
            if constexpr (OP == 12) OP8_3("v_mul_f64");
            if constexpr (OP == 13) OP8_4("v_fma_f64");
            if constexpr (OP == 14) OP8_3("v_xor_b32");
            if constexpr (OP == 15) OP8_3("v_pk_sub_u16");
            if constexpr (OP == 16) OP8_4("v_pk_mad_i16");
            if constexpr (OP == 17) OP8_3("v_pk_min_u16");
            if constexpr (OP == 18) {
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b));
            }
            if constexpr (OP == 19) { // compare (SDWA byte select) + cndmask pairs, as in the SW cell
                asm volatile("v_cmp_eq_u32_sdwa s[20:21], %0, %8 src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                             "v_cndmask_b32_e64 %1, 3, 5, s[20:21]\n\t"
                             "v_cmp_eq_u32_sdwa s[22:23], %2, %8 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                             "v_cndmask_b32_e64 %3, 3, 5, s[22:23]\n\t"
                             "v_cmp_eq_u32_sdwa s[24:25], %4, %8 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
                             "v_cndmask_b32_e64 %5, 3, 5, s[24:25]\n\t"
                             "v_cmp_eq_u32_sdwa s[26:27], %6, %8 src0_sel:BYTE_3 src1_sel:DWORD\n\t"
                             "v_cndmask_b32_e64 %7, 3, 5, s[26:27]"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b)
                             : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            }
            if constexpr (OP == 20) { // dependent chain: each instruction consumes the previous result
                asm volatile("v_max_i32 %0, %0, %8\n\tv_add_u32 %0, %0, %8\n\tv_max_i32 %0, %0, %8\n\tv_add_u32 %0, %0, %8\n\t"
                             "v_max_i32 %0, %0, %8\n\tv_add_u32 %0, %0, %8\n\tv_max_i32 %0, %0, %8\n\tv_add_u32 %0, %0, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename T, int OP>
void run(const char *name, int elems_per_instr)
{
    const int iters = 20000;
    T *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, sizeof(T) * 256 * 256 * 8));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8));
    printf("%-26s", name);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = 256 * w; // 256 CUs x w blocks of 4 waves = w waves per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((bench<T, OP>), dim3(blocks), dim3(256), 0, 0, 100, out, cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((bench<T, OP>), dim3(blocks), dim3(256), 0, 0, iters, out, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= blocks;
        const double instr_per_wave = (double)iters * 32;
        // s_memtime ticks per instruction of one wave, divided by the waves sharing the SIMD
        const double cyc_per_instr_simd = mean / instr_per_wave / w;
        const double wall_rate = (double)blocks * 4 * instr_per_wave * 64 * elems_per_instr / (ms * 1e-3) / 1e12;
        printf(" | w=%d: %5.2f tick/instr/SIMD %6.2f Tlane-op/s", w, cyc_per_instr_simd, wall_rate);
    }
    printf("\n");
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    printf("tick = s_memtime unit; 'Tlane-op/s' counts packed halves as separate lane-ops\n");
    run<int, 0>("v_add_u32", 1);
    run<int, 1>("v_max_i32", 1);
    run<int, 2>("v_max3_i32", 1);
    run<int, 14>("v_xor_b32", 1);
    run<int, 3>("v_pk_max_i16", 2);
    run<int, 4>("v_pk_add_i16", 2);
    run<int, 15>("v_pk_sub_u16", 2);
    run<int, 17>("v_pk_min_u16", 2);
    run<int, 16>("v_pk_mad_i16", 2);
    run<int, 18>("v_mov_b32_dpp wave_shr:1", 1);
    run<int, 19>("v_cmp_sdwa+v_cndmask", 1);
    run<int, 20>("dependent max/add chain", 1);
    run<float, 5>("v_add_f32", 1);
    run<float, 6>("v_mul_f32", 1);
    run<float, 7>("v_fma_f32", 1);
    run<double, 8>("v_pk_fma_f32", 2);
    run<double, 9>("v_pk_mul_f32", 2);
    run<double, 10>("v_pk_add_f32", 2);
    run<double, 11>("v_add_f64", 1);
    run<double, 12>("v_mul_f64", 1);
    run<double, 13>("v_fma_f64", 1);
    return 0;
}
