// Second VALU issue-rate microbenchmark for gfx950 (round 2): the opcodes considered for the
// re-formulated Smith-Waterman cell (biased unsigned halves, 32-bit adds, v_pk_maximum3_f16 used as an
// integer max3 on positive normal half patterns) and for mask-based selects, at 1..8 waves/SIMD, plus two
// instruction streams shaped like the old and the new cell.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench2.hip -o gpurun_out/valu_microbench2 && ./gpurun_out/valu_microbench2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

#define OP8_3(ins)                                                                                             \
    asm volatile(ins " %0, %0, %8\n\t" ins " %1, %1, %8\n\t" ins " %2, %2, %8\n\t" ins " %3, %3, %8\n\t" ins \
                     " %4, %4, %8\n\t" ins " %5, %5, %8\n\t" ins " %6, %6, %8\n\t" ins " %7, %7, %8"          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                 : "v"(b))
#define OP8_4(ins)                                                                                    \
    asm volatile(ins " %0, %0, %8, %8\n\t" ins " %1, %1, %8, %8\n\t" ins " %2, %2, %8, %8\n\t" ins   \
                     " %3, %3, %8, %8\n\t" ins " %4, %4, %8, %8\n\t" ins " %5, %5, %8, %8\n\t" ins   \
                     " %6, %6, %8, %8\n\t" ins " %7, %7, %8, %8"                                     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                 : "v"(b))
// b as the first source (v_sub_u32 a = b - a style orderings do not matter for timing)
#define OP8_4S(ins)                                                                                       \
    asm volatile(ins " %0, %0, %8, %9\n\t" ins " %1, %1, %8, %9\n\t" ins " %2, %2, %8, %9\n\t" ins       \
                     " %3, %3, %8, %9\n\t" ins " %4, %4, %8, %9\n\t" ins " %5, %5, %8, %9\n\t" ins       \
                     " %6, %6, %8, %9\n\t" ins " %7, %7, %8, %9"                                         \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)         \
                 : "v"(b), "s"(sc))

// 32 instructions per trip in every variant
template <int OP>
__global__ void __launch_bounds__(256) bench(int iters, unsigned *out, unsigned long long *cyc, unsigned sc)
{
    unsigned a0 = 0x08000800u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = 0x08010801u + (blockIdx.x & 3);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP < 100) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (OP == 0) OP8_3("v_sub_u32");
                if constexpr (OP == 1) OP8_3("v_and_b32");
                if constexpr (OP == 2) OP8_3("v_pk_max_u16");
                if constexpr (OP == 3) OP8_4("v_pk_maximum3_f16");
                if constexpr (OP == 4) OP8_3("v_pk_max_f16");
                if constexpr (OP == 5) OP8_4("v_perm_b32");
                if constexpr (OP == 6) OP8_4("v_bfe_i32");
                if constexpr (OP == 7) OP8_4("v_bfi_b32");
                if constexpr (OP == 8) OP8_3("v_lshlrev_b32");
                if constexpr (OP == 9) OP8_4("v_add3_u32");
                if constexpr (OP == 10) OP8_3("v_min_u32");
                if constexpr (OP == 11) OP8_3("v_or_b32");
                if constexpr (OP == 12) OP8_4("v_pk_minimum3_f16");
                if constexpr (OP == 13) OP8_4("v_maximum3_f32");
                if constexpr (OP == 14) OP8_4S("v_pk_maximum3_f16"); // one SGPR operand, as the clamp constant would be
                if constexpr (OP == 15) OP8_4("v_and_or_b32");
                if constexpr (OP == 16) OP8_4("v_xad_u32");
                if constexpr (OP == 17) OP8_4("v_max3_u16");
                if constexpr (OP == 18) OP8_4("v_lshl_add_u32");
                if constexpr (OP == 19) OP8_4("v_alignbit_b32");
            }
        }
        if constexpr (OP == 100) { // the round-1 packed cell, two columns: 12 instructions each, dependencies as in the kernel
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile(
                    "v_pk_add_u16 %4, %0, %8\n\t"  // e + ge
                    "v_pk_max_i16 %0, %1, %4\n\t"  // ev
                    "v_pk_add_u16 %5, %2, %8\n\t"  // f + ge
                    "v_pk_max_i16 %2, %3, %5\n\t"  // f
                    "v_xor_b32 %6, %7, %8\n\t"     // d
                    "v_pk_min_u16 %6, %6, %8\n\t"  // m2
                    "v_pk_add_u16 %4, %1, %8\n\t"  // hd1
                    "v_pk_sub_u16 %4, %4, %6 clamp\n\t"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(b));
        }
        if constexpr (OP == 101) { // a stream with the old cell's class mix: 11 packed + 1 xor per 12, x 8 = 96 ... scaled to 32: 29 pk + 3 xor
#pragma unroll
            for (int r = 0; r < 1; ++r) {
                asm volatile(
                    "v_pk_add_u16 %0, %0, %8\n\t" "v_pk_max_i16 %1, %1, %8\n\t" "v_pk_add_u16 %2, %2, %8\n\t" "v_pk_max_i16 %3, %3, %8\n\t"
                    "v_xor_b32 %4, %4, %8\n\t" "v_pk_min_u16 %5, %5, %8\n\t" "v_pk_add_u16 %6, %6, %8\n\t" "v_pk_sub_u16 %7, %7, %8 clamp\n\t"
                    "v_pk_max_i16 %0, %0, %8\n\t" "v_pk_max_i16 %1, %1, %8\n\t" "v_pk_add_u16 %2, %2, %8\n\t" "v_pk_max_i16 %3, %3, %8\n\t"
                    "v_pk_add_u16 %4, %4, %8\n\t" "v_pk_max_i16 %5, %5, %8\n\t" "v_pk_add_u16 %6, %6, %8\n\t" "v_pk_max_i16 %7, %7, %8\n\t"
                    "v_xor_b32 %0, %0, %8\n\t" "v_pk_min_u16 %1, %1, %8\n\t" "v_pk_add_u16 %2, %2, %8\n\t" "v_pk_sub_u16 %3, %3, %8 clamp\n\t"
                    "v_pk_max_i16 %4, %4, %8\n\t" "v_pk_max_i16 %5, %5, %8\n\t" "v_pk_add_u16 %6, %6, %8\n\t" "v_pk_max_i16 %7, %7, %8\n\t"
                    "v_pk_add_u16 %0, %0, %8\n\t" "v_pk_max_i16 %1, %1, %8\n\t" "v_pk_add_u16 %2, %2, %8\n\t" "v_pk_max_i16 %3, %3, %8\n\t"
                    "v_xor_b32 %4, %4, %8\n\t" "v_pk_min_u16 %5, %5, %8\n\t" "v_pk_add_u16 %6, %6, %8\n\t" "v_pk_sub_u16 %7, %7, %8 clamp"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(b));
            }
        }
        if constexpr (OP == 102) { // the new cell's class mix, independent: per 11: 6 fast (sub/add/xor u32) + 5 slow (2 max3_f16, 2 pk_max_u16, 1 pk_min) -> 32 = 17 fast + 15 slow
            asm volatile(
                "v_sub_u32 %0, %0, %8\n\t" "v_pk_maximum3_f16 %1, %1, %8, %8\n\t" "v_sub_u32 %2, %2, %8\n\t" "v_pk_max_u16 %3, %3, %8\n\t"
                "v_xor_b32 %4, %4, %8\n\t" "v_pk_min_u16 %5, %5, %8\n\t" "v_add_u32 %6, %6, %8\n\t" "v_sub_u32 %7, %7, %8\n\t"
                "v_pk_maximum3_f16 %0, %0, %8, %8\n\t" "v_sub_u32 %1, %1, %8\n\t" "v_pk_max_u16 %2, %2, %8\n\t"
                "v_sub_u32 %3, %3, %8\n\t" "v_pk_maximum3_f16 %4, %4, %8, %8\n\t" "v_sub_u32 %5, %5, %8\n\t" "v_pk_max_u16 %6, %6, %8\n\t"
                "v_xor_b32 %7, %7, %8\n\t" "v_pk_min_u16 %0, %0, %8\n\t" "v_add_u32 %1, %1, %8\n\t" "v_sub_u32 %2, %2, %8\n\t"
                "v_pk_maximum3_f16 %3, %3, %8, %8\n\t" "v_sub_u32 %4, %4, %8\n\t" "v_pk_max_u16 %5, %5, %8\n\t"
                "v_sub_u32 %6, %6, %8\n\t" "v_pk_maximum3_f16 %7, %7, %8, %8\n\t" "v_sub_u32 %0, %0, %8\n\t" "v_pk_max_u16 %1, %1, %8\n\t"
                "v_xor_b32 %2, %2, %8\n\t" "v_pk_min_u16 %3, %3, %8\n\t" "v_add_u32 %4, %4, %8\n\t" "v_sub_u32 %5, %5, %8\n\t"
                "v_pk_maximum3_f16 %6, %6, %8, %8\n\t" "v_sub_u32 %7, %7, %8"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(b));
        }
        if constexpr (OP == 104) { // 4 fast then 4 slow
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile(
                    "v_sub_u32 %0, %0, %8\n\t" "v_sub_u32 %1, %1, %8\n\t" "v_sub_u32 %2, %2, %8\n\t" "v_sub_u32 %3, %3, %8\n\t"
                    "v_pk_max_u16 %4, %4, %8\n\t" "v_pk_max_u16 %5, %5, %8\n\t" "v_pk_max_u16 %6, %6, %8\n\t" "v_pk_max_u16 %7, %7, %8"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(b));
        }
        if constexpr (OP == 105) { // 16 fast then 16 slow
            asm volatile(
                "v_sub_u32 %0, %0, %8\n\t" "v_sub_u32 %1, %1, %8\n\t" "v_sub_u32 %2, %2, %8\n\t" "v_sub_u32 %3, %3, %8\n\t"
                "v_sub_u32 %4, %4, %8\n\t" "v_sub_u32 %5, %5, %8\n\t" "v_sub_u32 %6, %6, %8\n\t" "v_sub_u32 %7, %7, %8\n\t"
                "v_sub_u32 %0, %0, %8\n\t" "v_sub_u32 %1, %1, %8\n\t" "v_sub_u32 %2, %2, %8\n\t" "v_sub_u32 %3, %3, %8\n\t"
                "v_sub_u32 %4, %4, %8\n\t" "v_sub_u32 %5, %5, %8\n\t" "v_sub_u32 %6, %6, %8\n\t" "v_sub_u32 %7, %7, %8\n\t"
                "v_pk_max_u16 %0, %0, %8\n\t" "v_pk_max_u16 %1, %1, %8\n\t" "v_pk_max_u16 %2, %2, %8\n\t" "v_pk_max_u16 %3, %3, %8\n\t"
                "v_pk_max_u16 %4, %4, %8\n\t" "v_pk_max_u16 %5, %5, %8\n\t" "v_pk_max_u16 %6, %6, %8\n\t" "v_pk_max_u16 %7, %7, %8\n\t"
                "v_pk_max_u16 %0, %0, %8\n\t" "v_pk_max_u16 %1, %1, %8\n\t" "v_pk_max_u16 %2, %2, %8\n\t" "v_pk_max_u16 %3, %3, %8\n\t"
                "v_pk_max_u16 %4, %4, %8\n\t" "v_pk_max_u16 %5, %5, %8\n\t" "v_pk_max_u16 %6, %6, %8\n\t" "v_pk_max_u16 %7, %7, %8"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(b));
        }
        if constexpr (OP == 106) { // 2 fast then 2 slow
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile(
                    "v_sub_u32 %0, %0, %8\n\t" "v_sub_u32 %1, %1, %8\n\t" "v_pk_max_u16 %2, %2, %8\n\t" "v_pk_max_u16 %3, %3, %8\n\t"
                    "v_sub_u32 %4, %4, %8\n\t" "v_sub_u32 %5, %5, %8\n\t" "v_pk_max_u16 %6, %6, %8\n\t" "v_pk_max_u16 %7, %7, %8"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(b));
        }
        if constexpr (OP == 107) { // fast ops with distinct opcodes alternating (sub/xor/add/and)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile(
                    "v_sub_u32 %0, %0, %8\n\t" "v_xor_b32 %1, %1, %8\n\t" "v_add_u32 %2, %2, %8\n\t" "v_and_b32 %3, %3, %8\n\t"
                    "v_sub_u32 %4, %4, %8\n\t" "v_xor_b32 %5, %5, %8\n\t" "v_add_u32 %6, %6, %8\n\t" "v_or_b32 %7, %7, %8"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(b));
        }
        if constexpr (OP == 108) { // fast ops reading two different VGPR sources (a_k, a_k+1) -- bank effects?
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile(
                    "v_sub_u32 %0, %1, %8\n\t" "v_sub_u32 %1, %2, %8\n\t" "v_sub_u32 %2, %3, %8\n\t" "v_sub_u32 %3, %4, %8\n\t"
                    "v_sub_u32 %4, %5, %8\n\t" "v_sub_u32 %5, %6, %8\n\t" "v_sub_u32 %6, %7, %8\n\t" "v_sub_u32 %7, %0, %8"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(b));
        }
        if constexpr (OP == 109) { // v_sub_u32 with an SGPR constant
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile(
                    "v_subrev_u32 %0, %8, %0\n\t" "v_subrev_u32 %1, %8, %1\n\t" "v_subrev_u32 %2, %8, %2\n\t" "v_subrev_u32 %3, %8, %3\n\t"
                    "v_subrev_u32 %4, %8, %4\n\t" "v_subrev_u32 %5, %8, %5\n\t" "v_subrev_u32 %6, %8, %6\n\t" "v_subrev_u32 %7, %8, %7"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "s"(sc));
        }
        if constexpr (OP == 103) { // alternating fast / slow, independent
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile(
                    "v_sub_u32 %0, %0, %8\n\t" "v_pk_max_u16 %1, %1, %8\n\t" "v_sub_u32 %2, %2, %8\n\t" "v_pk_max_u16 %3, %3, %8\n\t"
                    "v_sub_u32 %4, %4, %8\n\t" "v_pk_max_u16 %5, %5, %8\n\t" "v_sub_u32 %6, %6, %8\n\t" "v_pk_max_u16 %7, %7, %8"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(b));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name)
{
    const int iters = 20000;
    unsigned *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, sizeof(unsigned) * 256 * 256 * 8));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8));
    printf("%-34s", name);
    for (int w : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * w; // 256 CUs x w blocks of 4 waves = w waves per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((bench<OP>), dim3(blocks), dim3(256), 0, 0, 100, out, cyc, 0x08000800u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((bench<OP>), dim3(blocks), dim3(256), 0, 0, iters, out, cyc, 0x08000800u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_wave = (double)iters * 32;
        // wall-clock nanoseconds one SIMD spends per wave64 instruction, and the same in 2.4 GHz cycles
        const double ns = (ms * 1e6) / (instr_per_wave * w);
        printf(" | w=%d: %5.2f cyc/instr", w, ns * 2.4);
    }
    printf("\n");
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d kHz; cycles = wall time x 2.4 GHz per wave64 instruction per SIMD\n", p.name, p.multiProcessorCount, p.clockRate);
    run<0>("v_sub_u32");
    run<1>("v_and_b32");
    run<11>("v_or_b32");
    run<8>("v_lshlrev_b32");
    run<10>("v_min_u32");
    run<9>("v_add3_u32");
    run<18>("v_lshl_add_u32");
    run<16>("v_xad_u32");
    run<15>("v_and_or_b32");
    run<19>("v_alignbit_b32");
    run<2>("v_pk_max_u16");
    run<4>("v_pk_max_f16");
    run<3>("v_pk_maximum3_f16");
    run<14>("v_pk_maximum3_f16 (1 SGPR src)");
    run<12>("v_pk_minimum3_f16");
    run<13>("v_maximum3_f32");
    run<17>("v_max3_u16");
    run<5>("v_perm_b32");
    run<6>("v_bfe_i32");
    run<7>("v_bfi_b32");
    run<100>("old SW cell (dependent, 8 of 12)");
    run<101>("old SW cell class mix (indep.)");
    run<102>("new SW cell class mix (indep.)");
    run<103>("alternating sub_u32 / pk_max_u16");
    run<106>("2 sub_u32 then 2 pk_max_u16");
    run<104>("4 sub_u32 then 4 pk_max_u16");
    run<105>("16 sub_u32 then 16 pk_max_u16");
    run<107>("fast mix sub/xor/add/and/or");
    run<108>("v_sub_u32 dst != src");
    run<109>("v_subrev_u32 SGPR constant");
    return 0;
}
