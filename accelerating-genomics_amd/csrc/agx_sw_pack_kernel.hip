// Builds the packed Smith-Waterman image on the device from the caller's raw `bases` array.
//
// The host no longer shuffles bytes (round 1 memcpy'd every sequence into a padded host image and
// uploaded that with a pageable hipMemcpy: 5 ms of 6 for BASELINE config 2).  It uploads `bases` and
// `off` exactly as the caller holds them plus the group records its planner wrote, and this kernel
// copies every pair's shorter sequence (zero-padded to its lane group's G*C columns + one spare word)
// and longer sequence (zero-padded to 4 bytes) to the image offsets the records name -- the layout
// csrc/agx_sw.h describes, which the fill kernels read unchanged.  It also performs the input check the
// packer used to do: byte 0x00 (the padding symbol) in a real sequence, or a byte outside the
// substitution matrix's alphabet, is reported through `flag` ([0] = count, [1] = smallest pair index).
//
// HBM-bound and tiny next to the fill: one read of the sequences, one write of the image
// (config 2: 20 MB + 22 MB, about 15 us).
#include "agx_sw.h"

namespace {

// one byte of a sequence as the image holds it; idx beyond the sequence = padding
template <bool MAT>
__device__ __forceinline__ uint32_t fetch(const uint8_t *__restrict__ seq, uint32_t idx, uint32_t len,
                                          const uint8_t *__restrict__ code, bool &bad)
{
    if (idx >= len) return 0u;
    uint32_t b = seq[idx];
    if constexpr (MAT) {
        const uint32_t c = code[b];
        bad |= c == 0xffu;
        b = (c + 1u) & 0xffu; // symbol numbers 1..n, 0 stays the padding symbol
    } else
        bad |= b == 0u;
    return b;
}

template <bool MAT>
__device__ __forceinline__ void copy_seq(uint32_t *__restrict__ dst, uint32_t n_dw, const uint8_t *__restrict__ seq, uint32_t len,
                                         const uint8_t *__restrict__ code, int lane, bool &bad)
{
    for (uint32_t i = lane; i < n_dw; i += 64) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) v |= fetch<MAT>(seq, 4 * i + k, len, code, bad) << (8 * k);
        dst[i] = v;
    }
}

template <bool MAT, int SLOTS>
__global__ void __launch_bounds__(256) sw_pack(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ off, uint64_t base,
                                               const uint32_t *__restrict__ groups, uint32_t n_groups, uint32_t n_pairs,
                                               uint32_t *__restrict__ img, const uint8_t *__restrict__ code,
                                               uint32_t *__restrict__ flag)
{
    __shared__ uint8_t lcode[MAT ? 256 : 1];
    if constexpr (MAT) {
        lcode[threadIdx.x] = code[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t total = n_groups * (uint32_t)SLOTS;
    for (uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 6); slot < total; slot += n_waves) {
        const uint32_t g = slot / SLOTS, h = slot % SLOTS;
        // SwGroup: {x_dw, y_dw, lx_ly, out};  SwGroup2: {x_dw[2], y_dw[2], lx_ly[2], out[2]}
        const uint32_t *rec = groups + (size_t)g * (4 * SLOTS);
        const uint32_t x_dw = rec[h], y_dw = rec[SLOTS + h], ll = rec[2 * SLOTS + h], out = rec[3 * SLOTS + h];
        if (out >= n_pairs) continue; // vacant half of a packed group: points at the zero block
        const uint32_t lx = ll & 0x7fffu, xsec = (ll >> 15) & 1u, ly = ll >> 16;
        const uint8_t *x = raw + (off[2 * (size_t)out + xsec] - base);
        const uint8_t *y = raw + (off[2 * (size_t)out + (xsec ^ 1u)] - base);
        bool bad = false;
        copy_seq<MAT>(img + x_dw, y_dw - x_dw, x, lx, lcode, lane, bad);
        copy_seq<MAT>(img + y_dw, (ly + 3u) >> 2, y, ly, lcode, lane, bad);
        if (__any(bad) && lane == 0) {
            atomicAdd(&flag[0], 1u);
            atomicMin(&flag[1], out);
        }
    }
}

} // namespace

int agx_sw_pack_launch(bool matrix, int slots, const uint8_t *raw, const uint64_t *off, uint64_t base, const void *groups,
                       uint32_t n_groups, uint32_t n_pairs, uint32_t *img, const uint8_t *code, uint32_t *flag, int n_cu,
                       hipStream_t s)
{
    if (n_groups == 0) return 0;
    const uint64_t total = (uint64_t)n_groups * (uint64_t)slots;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 3) / 4, (uint64_t)std::max(n_cu, 1) * 16u);
    const uint32_t *g = (const uint32_t *)groups;
    if (matrix)
        hipLaunchKernelGGL((sw_pack<true, 1>), dim3(blocks), dim3(256), 0, s, raw, off, base, g, n_groups, n_pairs, img, code, flag);
    else if (slots == 2)
        hipLaunchKernelGGL((sw_pack<false, 2>), dim3(blocks), dim3(256), 0, s, raw, off, base, g, n_groups, n_pairs, img, code, flag);
    else
        hipLaunchKernelGGL((sw_pack<false, 1>), dim3(blocks), dim3(256), 0, s, raw, off, base, g, n_groups, n_pairs, img, code, flag);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
