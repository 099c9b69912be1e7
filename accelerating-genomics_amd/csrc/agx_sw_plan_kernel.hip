// The Smith-Waterman planner's O(pairs) passes on the device (round 3).
//
// What agx_sw.cpp's host planner does for a mixed batch -- tile every pair by table lookup, sort the pairs by
// (class, lanes per group descending, longer side descending, file order), cut the sorted list into groups and
// wavefronts, give every pair its place in the image, write the group and wave records, order the waves longest
// first -- as kernels behind the upload of len[], so that the host is left with one reduction pass over len[] / off[]
// and the batch-level rules.  The records are the ones the host planner writes, byte for byte (same keys, stable
// sorts): tests/test_sw_gpu.py compares the two planners through agx_sw_batch_info and the scores.
//
//   sw_plan_keys     pair -> sort key (class << 22 | (64 - G) << 16 | longest - ly), value = pair number
//   [radix sort]     rocprim::radix_sort_pairs, 28 bits, stable (pairs with an empty side sort last)
//   sw_plan_words    sorted entry -> words of its image block; [exclusive scan] -> image offsets
//   sw_plan_records  sorted entry -> its half of a SwGroup2 / its SwGroup, the SwWave of the wave it opens,
//                    that wave's dispatch key (longest first), padded cells
//   [radix sort]     of the waves' keys, stable; sw_plan_gather lays the wave records out in that order
//
// The host gives the (class, G) buckets' extents -- it counts them in its reduction pass -- so nothing here needs a
// round trip: every launch size is known before the first kernel runs.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "agx_sw.h"

namespace {

__constant__ int d_classes[kSwNumClasses] = {4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 80, 120, 160};

__global__ void __launch_bounds__(256) sw_plan_keys(const uint32_t *__restrict__ len, uint32_t n_pairs, const uint32_t *__restrict__ seg_first,
                                                    const uint32_t *__restrict__ segs, uint32_t longest, uint32_t *__restrict__ keys,
                                                    uint32_t *__restrict__ vals)
{
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n_pairs; p += gridDim.x * 256u) {
        const uint2 l = reinterpret_cast<const uint2 *>(len)[p];
        uint32_t key = kSwPlanEmptyKey;
        if (l.x && l.y) {
            const uint32_t lx = min(l.x, l.y), ly = max(l.x, l.y);
            uint32_t k = seg_first[lx];
            const uint32_t end = seg_first[lx + 1];
            while (k + 1 < end && (segs[k + 1] & 0xffffu) <= ly) ++k; // the lower envelope's segment that covers ly
            const uint32_t s = segs[k], cls = (s >> 16) & 0xffu, G = s >> 24;
            key = ((cls * 64u + 64u - G) << 16) | (longest - ly);
        }
        keys[p] = key;
        vals[p] = p;
    }
}

__device__ __forceinline__ void decode_key(uint32_t key, uint32_t longest, uint32_t &bucket, uint32_t &C, uint32_t &G, uint32_t &ly)
{
    bucket = key >> 16;
    C = (uint32_t)d_classes[bucket >> 6];
    G = 64u - (bucket & 63u);
    ly = longest - (key & 0xffffu);
}

__global__ void __launch_bounds__(256) sw_plan_words(const uint32_t *__restrict__ keys, uint32_t n_fill, uint32_t longest, uint32_t *__restrict__ words)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_fill; i += gridDim.x * 256u) {
        uint32_t b, C, G, ly;
        decode_key(keys[i], longest, b, C, G, ly);
        words[i] = (G * C + 3u) / 4u + 1u + (ly + 3u) / 4u; // [x block: G * C bytes + a spare word][y block]
    }
}

// bucket table rows (host-made, one per (class, G) id): first entry, entries, first group, groups, first wave
template <int SLOTS>
__global__ void __launch_bounds__(256) sw_plan_records(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ scan,
                                                       const uint32_t *__restrict__ len, const uint32_t *__restrict__ buckets, uint32_t n_fill,
                                                       uint32_t n_pairs, uint32_t longest, uint32_t img0, uint32_t *__restrict__ groups,
                                                       SwWave *__restrict__ waves, uint32_t *__restrict__ wave_keys, uint32_t *__restrict__ wave_ids,
                                                       unsigned long long *__restrict__ padded)
{
    unsigned long long my_padded = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_fill; i += gridDim.x * 256u) {
        uint32_t b, C, G, ly;
        decode_key(keys[i], longest, b, C, G, ly);
        const uint32_t *bk = buckets + 5u * b;
        const uint32_t first = bk[0], count = bk[1], group0 = bk[2], n_groups = bk[3], wave0 = bk[4];
        const uint32_t j = i - first, g = group0 + j / SLOTS, h = j % SLOTS;
        const uint32_t pair = vals[i];
        const uint2 l = reinterpret_cast<const uint2 *>(len)[pair];
        const uint32_t second_short = l.y < l.x ? 1u : 0u; // ties keep file order (antidiagonalSmithWaterman.c:229-244)
        const uint32_t lx = second_short ? l.y : l.x;
        const uint32_t x_dw = img0 + scan[i], y_dw = x_dw + (G * C + 3u) / 4u + 1u;
        const uint32_t ll = lx | (second_short << 15) | (ly << 16);
        uint32_t *rec = groups + (size_t)g * (4u * SLOTS); // SwGroup {x, y, ll, out} / SwGroup2 {x[2], y[2], ll[2], out[2]}
        rec[h] = x_dw;
        rec[SLOTS + h] = y_dw;
        rec[2 * SLOTS + h] = ll;
        rec[3 * SLOTS + h] = pair;
        if (SLOTS == 2 && h == 0 && j + 1 == count) { // the group's second half is vacant: zero block, spare score slot
            rec[1] = 0;
            rec[3] = 0;
            rec[5] = 0;
            rec[7] = n_pairs;
        }
        const uint32_t per_wave = 64u / G;
        if (j % (per_wave * SLOTS) == 0) { // this entry opens a wave: rows are sorted long first, so its ly is the wave's longest
            const uint32_t wl = j / (per_wave * SLOTS), w = wave0 + wl;
            SwWave wv;
            wv.first_group = group0 + wl * per_wave;
            wv.n_groups = (uint16_t)min(per_wave, n_groups - wl * per_wave);
            wv.G = (uint16_t)G;
            wv.steps = ly + G - 1u;
            wv.reserved = C;
            waves[w] = wv;
            wave_keys[w] = kSwPlanWaveKeyMax - wv.steps * C; // ascending = longest (steps x columns) first
            wave_ids[w] = w;
            my_padded += (unsigned long long)wv.steps * 64u * C * SLOTS;
        }
    }
    // one atomic per wavefront
#pragma unroll
    for (int d = 32; d; d >>= 1) my_padded += __shfl_xor(my_padded, d);
    if ((threadIdx.x & 63) == 0 && my_padded) atomicAdd(padded, my_padded);
}

__global__ void __launch_bounds__(256) sw_plan_gather(const SwWave *__restrict__ in, const uint32_t *__restrict__ order, uint32_t n_waves, SwWave *__restrict__ out)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_waves; i += gridDim.x * 256u)
        reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(in)[order[i]];
}

inline uint32_t blocks_for(uint64_t n, int n_cu) { return (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)std::max(n_cu, 1) * 16u); }

} // namespace

size_t agx_sw_plan_temp_bytes(uint32_t n_pairs, uint32_t n_waves)
{
    size_t a = 0, b = 0, c = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (size_t)n_pairs, 0u, 28u, (hipStream_t) nullptr);
    (void)rocprim::exclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)n_pairs, rocprim::plus<uint32_t>(),
                                  (hipStream_t) nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, c, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (size_t)std::max(n_waves, 1u), 0u, 24u, (hipStream_t) nullptr);
    return std::max(a, std::max(b, c)) + 256;
}

int agx_sw_plan_launch(const SwPlanArgs &a, hipStream_t s)
{
    static_assert(sizeof(SwWave) == 16, "sw_plan_gather moves wave records as 16-byte words");
    const uint32_t n = a.n_pairs;
    hipLaunchKernelGGL(sw_plan_keys, dim3(blocks_for(n, a.n_cu)), dim3(256), 0, s, a.len, n, a.seg_first, a.segs, a.longest, a.keys_a, a.vals_a);
    size_t tb = a.temp_bytes;
    if (rocprim::radix_sort_pairs(a.temp, tb, a.keys_a, a.keys_b, a.vals_a, a.vals_b, (size_t)n, 0u, 28u, s) != hipSuccess) return -1;
    if (a.n_fill == 0) return hipGetLastError() == hipSuccess ? 0 : -1;
    // keys_a / vals_a are free again: words and their scan live there
    uint32_t *words = a.keys_a, *scan = a.vals_a;
    hipLaunchKernelGGL(sw_plan_words, dim3(blocks_for(a.n_fill, a.n_cu)), dim3(256), 0, s, a.keys_b, a.n_fill, a.longest, words);
    tb = a.temp_bytes;
    if (rocprim::exclusive_scan(a.temp, tb, words, scan, 0u, (size_t)a.n_fill, rocprim::plus<uint32_t>(), s) != hipSuccess) return -1;
    if (a.slots == 2)
        hipLaunchKernelGGL((sw_plan_records<2>), dim3(blocks_for(a.n_fill, a.n_cu)), dim3(256), 0, s, a.keys_b, a.vals_b, scan, a.len, a.buckets, a.n_fill, n,
                           a.longest, a.img0, a.groups, a.waves_tmp, a.wave_keys_a, a.wave_ids_a, a.padded);
    else
        hipLaunchKernelGGL((sw_plan_records<1>), dim3(blocks_for(a.n_fill, a.n_cu)), dim3(256), 0, s, a.keys_b, a.vals_b, scan, a.len, a.buckets, a.n_fill, n,
                           a.longest, a.img0, a.groups, a.waves_tmp, a.wave_keys_a, a.wave_ids_a, a.padded);
    tb = a.temp_bytes;
    if (rocprim::radix_sort_pairs(a.temp, tb, a.wave_keys_a, a.wave_keys_b, a.wave_ids_a, a.wave_ids_b, (size_t)a.n_waves, 0u, 24u, s) != hipSuccess)
        return -1;
    hipLaunchKernelGGL(sw_plan_gather, dim3(blocks_for(a.n_waves, a.n_cu)), dim3(256), 0, s, a.waves_tmp, a.wave_ids_b, a.n_waves, a.waves);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

void agx_sw_plan_preload()
{
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&sw_plan_keys));
}
