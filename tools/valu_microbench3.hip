// Third VALU microbenchmark (round 2): WHY the full-rate class (v_add/sub/xor_u32, 2.4 cycles per wave64 in a
// pure stream) buys nothing inside the SW cell, where it alternates with packed instructions.
//   M1  wave-parity split: even waves issue only v_sub_u32, odd waves only v_pk_max_u16 (same SIMD)
//   M2  every wave alternates the two classes on disjoint accumulators, 1024-thread workgroups (4 waves per
//       SIMD), with / without an s_barrier per trip that keeps the SIMD's waves in phase
//   M3  every wave alternates the two classes on the SAME accumulator (a dependent chain crossing classes)
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench3.hip -o /tmp/vm3 && /tmp/vm3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
#define F8 "v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t"
#define S8 "v_pk_max_u16 %0, %0, %8\n\tv_pk_max_u16 %1, %1, %8\n\tv_pk_max_u16 %2, %2, %8\n\tv_pk_max_u16 %3, %3, %8\n\tv_pk_max_u16 %4, %4, %8\n\tv_pk_max_u16 %5, %5, %8\n\tv_pk_max_u16 %6, %6, %8\n\tv_pk_max_u16 %7, %7, %8\n\t"
#define ALT8 "v_sub_u32 %0, %0, %8\n\tv_pk_max_u16 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_pk_max_u16 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_pk_max_u16 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_pk_max_u16 %7, %7, %8\n\t"
// chain crossing classes: each accumulator sees sub, pk_max, sub, pk_max ...
#define DEP8A "v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_pk_max_u16 %4, %4, %8\n\tv_pk_max_u16 %5, %5, %8\n\tv_pk_max_u16 %6, %6, %8\n\tv_pk_max_u16 %7, %7, %8\n\t"
#define DEP8B "v_pk_max_u16 %0, %0, %8\n\tv_pk_max_u16 %1, %1, %8\n\tv_pk_max_u16 %2, %2, %8\n\tv_pk_max_u16 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t"
// same instruction order as DEP8A/DEP8B but the classes never share an accumulator
#define IND8A "v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_pk_max_u16 %4, %4, %8\n\tv_pk_max_u16 %5, %5, %8\n\tv_pk_max_u16 %6, %6, %8\n\tv_pk_max_u16 %7, %7, %8\n\t"

template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) bench(int iters, unsigned *out)
{
    unsigned a0 = 0x08000800u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = 0x00010001u + (blockIdx.x & 3);
    const int wv = threadIdx.x >> 6;
    // waves of a workgroup are dealt to the SIMDs round-robin: waves wv and wv + 4 share a SIMD
    const bool fast_wave = ((wv >> 2) & 1) == 0;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 1) {
            if (fast_wave) asm volatile(F8 F8 F8 F8 REGS);
            else asm volatile(S8 S8 S8 S8 REGS);
        }
        if constexpr (MODE == 2) asm volatile(ALT8 ALT8 ALT8 ALT8 REGS);
        if constexpr (MODE == 3) {
            asm volatile(ALT8 ALT8 ALT8 ALT8 REGS);
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (MODE == 4) asm volatile(DEP8A DEP8B DEP8A DEP8B REGS);
        if constexpr (MODE == 5) asm volatile(IND8A IND8A IND8A IND8A REGS);
        if constexpr (MODE == 6) asm volatile(F8 F8 F8 F8 REGS);
        if constexpr (MODE == 7) asm volatile(S8 S8 S8 S8 REGS);
        if constexpr (MODE == 8) { // all waves: fast block, barrier, slow block, barrier (phases aligned chip-wide per workgroup)
            asm volatile(F8 F8 REGS);
            __builtin_amdgcn_s_barrier();
            asm volatile(S8 S8 REGS);
            __builtin_amdgcn_s_barrier();
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE, int THREADS>
void run(const char *name)
{
    const int iters = 20000;
    unsigned *out;
    CHECK(hipMalloc(&out, sizeof(unsigned) * 1024 * 256 * 2));
    printf("%-58s", name);
    for (int wps : {2, 4, 8}) { // waves per SIMD
        const int waves_per_block = THREADS / 64;
        const int blocks = 256 * wps * 4 / waves_per_block;
        if (blocks * THREADS > 1024 * 256 * 2 || blocks < 256) { printf(" | w=%d:   n/a        ", wps); continue; }
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((bench<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, 100, out);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((bench<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ns = (ms * 1e6) / ((double)iters * 32 * wps);
        printf(" | w=%d: %5.2f cyc/instr", wps, ns * 2.4);
    }
    printf("\n");
    CHECK(hipFree(out));
}

int main()
{
    printf("cycles = wall time x 2.4 GHz per wave64 instruction per SIMD (averaged over all waves of the SIMD)\n");
    run<6, 512>("pure v_sub_u32, 512-thread groups");
    run<7, 512>("pure v_pk_max_u16, 512-thread groups");
    run<1, 512>("M1 parity split (waves 0-3 fast, 4-7 slow), 512 threads");
    run<2, 1024>("M2 alternating, disjoint accumulators, 1024 threads");
    run<3, 1024>("M2 same + s_barrier per 32 instructions");
    run<8, 1024>("M2' 16 fast | barrier | 16 slow | barrier, 1024 threads");
    run<5, 256>("M3 ref: 4 fast 4 slow, disjoint accumulators");
    run<4, 256>("M3 4 fast 4 slow, chains crossing classes");
    return 0;
}
