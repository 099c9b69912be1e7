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

// ---------------------------------------------------------------------------------------------------------------
// The same for the biased packed fill (agx_sw_pk2_kernel.hip), plus its DNA test.  One pack wavefront per FILL
// wavefront: it looks at every pair of that wave and, when all of them qualify, writes CODES instead of bytes
// (and says so in bit 16 of the wave record's class word):
//   * a pair qualifies when its shorter sequence x -- without a final newline -- holds at most four distinct
//     symbols, and a final newline of either sequence could align with nothing but the other's final newline
//     (no newline inside the other sequence);
//   * x becomes v_perm_b32 selector bytes: code 0..3 (its symbols in ascending byte order; 4 + code in the group's
//     second pair), RIGHT-aligned in the group's G * C columns, 0x0c in the padding columns before it;
//   * y becomes shift counts 8 * (3 - code), or 31 for a symbol x does not contain and for the padding of the last
//     quad (pk2_fill's fast_head turns them into the row tables);
//   * final newlines are stripped and the record's lengths rewritten (lx' | both-had-one << 13 |
//     second-is-shorter << 15 | ly' << 16) -- the fill adds the sentinel match back (see pk2_fill).
// A wave with a pair that does not qualify (an N in a read, protein letters, ...) keeps the byte image and runs
// the general cell.
__device__ __forceinline__ uint32_t dna_code(uint32_t b, uint32_t syms, int nsym)
{
    uint32_t c = 4;
#pragma unroll
    for (int k = 3; k >= 0; --k)
        if (k < nsym && b == ((syms >> (8 * k)) & 0xffu)) c = (uint32_t)k;
    return c;
}

// calls f(byte) for the n bytes at p, fetched as aligned words (p has any alignment; the words that hold the first
// and the last byte lie inside the upload: it starts on a word and ends with a spare one)
template <class F>
__device__ __forceinline__ void for_each_byte(const uint8_t *p, uint32_t n, F f)
{
    const uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u), total = lead + n;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - lead);
    for (uint32_t w = 0; 4u * w < total; ++w) {
        const uint32_t v = q[w];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t idx = 4u * w + k;
            if (idx >= lead && idx < total) f((v >> (8 * k)) & 0xffu);
        }
    }
}

constexpr uint32_t kDnaLongSide = 4096; // a longer y is scanned by a whole wavefront instead of one lane

// One workgroup (4 wavefronts) per fill wavefront.  Pass 1, two THREADS per pair, one on each sequence: the distinct
// symbols of x' (kept in the order met -- the codes only have to agree between x and y), the sentinels, the byte-0
// check; the verdict of the whole fill wave is the AND over its pairs.  Pass 2, one wavefront per pair in turn, lanes across the bytes:
// the image, coded or as bytes.  (The first version ran both passes pair by pair on one wavefront with five
// reduction rounds per pair: 321 us for config 2 against 45 us of the plain pack; profiles/r02b.)
__global__ void __launch_bounds__(256) sw_pack_dna(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ off, uint64_t base,
                                                   uint32_t *__restrict__ groups, uint32_t *__restrict__ waves, uint32_t n_fill_waves,
                                                   uint32_t n_pairs, uint32_t *__restrict__ img, uint32_t *__restrict__ flag)
{
    // per pair of the fill wave (at most 64 groups x 2): {symbols of x', packed}, {verdict and lengths}
    __shared__ uint32_t s_syms[128];
    __shared__ uint32_t s_info[128];
    __shared__ uint32_t s_yinfo[128];
    __shared__ uint32_t s_not_dna;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (uint32_t fw = blockIdx.x; fw < n_fill_waves; fw += gridDim.x) {
        // SwWave: {first_group, n_groups | G << 16, steps, class word}
        const uint32_t first_group = waves[4 * (size_t)fw], n_groups = waves[4 * (size_t)fw + 1] & 0xffffu;
        const uint32_t gc = (waves[4 * (size_t)fw + 1] >> 16) * (waves[4 * (size_t)fw + 3] & 0xffffu); // columns of a lane group
        const uint32_t n_slots = n_groups * 2u;
        if (threadIdx.x == 0) s_not_dna = 0;
        __syncthreads();
        // ---- pass 1: two threads per pair, one on each sequence
        {
            const uint32_t sl = threadIdx.x >> 1, which = threadIdx.x & 1u;
            if (sl < n_slots) {
                const uint32_t *rec = groups + (size_t)(first_group + sl / 2u) * 8u;
                const uint32_t h = sl & 1u, ll = rec[4 + h], out = rec[6 + h];
                if (out >= n_pairs) { // vacant half: nothing to read, fits either kind of wave
                    if (which == 0) {
                        s_syms[sl] = 0;
                        s_info[sl] = 0x80000000u;
                    } else
                        s_yinfo[sl] = 0;
                } else {
                    const uint32_t lx = ll & 0x7fffu, xsec = (ll >> 15) & 1u, ly = ll >> 16;
                    const uint8_t *x = raw + (off[2 * (size_t)out + xsec] - base);
                    const uint8_t *y = raw + (off[2 * (size_t)out + (xsec ^ 1u)] - base);
                    const uint32_t xnl = lx && x[lx - 1] == '\n', ynl = ly && y[ly - 1] == '\n';
                    const uint32_t lxs = lx - xnl, lys = ly - ynl;
                    bool bad = false, has_nl = false;
                    if (which == 0) {
                        uint32_t s0 = 0x100, s1 = 0x100, s2 = 0x100, s3 = 0x100, nsym = 0; // 0x100: no byte
                        for_each_byte(x, lxs, [&](uint32_t b) {
                            bad |= b == 0u;
                            has_nl |= b == '\n';
                            if (b != s0 && b != s1 && b != s2 && b != s3) {
                                if (nsym == 0) s0 = b;
                                if (nsym == 1) s1 = b;
                                if (nsym == 2) s2 = b;
                                if (nsym == 3) s3 = b;
                                nsym = min(nsym + 1u, 5u); // 5 = "more than four"
                            }
                        });
                        s_syms[sl] = (s0 & 0xffu) | ((s1 & 0xffu) << 8) | ((s2 & 0xffu) << 16) | ((s3 & 0xffu) << 24);
                        s_info[sl] = nsym | (xnl << 4) | (ynl << 5) | ((uint32_t)(lys > kDnaLongSide) << 6) | ((uint32_t)has_nl << 7) |
                                     ((uint32_t)bad << 8) | ((uint32_t)(lx >= 4096u) << 9);
                    } else {
                        if (lys <= kDnaLongSide)
                            for_each_byte(y, lys, [&](uint32_t b) {
                                bad |= b == 0u;
                                has_nl |= b == '\n';
                            });
                        s_yinfo[sl] = (uint32_t)has_nl | ((uint32_t)bad << 1);
                    }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < n_slots) {
            const uint32_t sl = threadIdx.x, info = s_info[sl], yinfo = s_yinfo[sl];
            if (!(info & 0x80000000u)) {
                const uint32_t nsym = info & 0xfu, xnl = (info >> 4) & 1u, ynl = (info >> 5) & 1u, x_has_nl = (info >> 7) & 1u;
                if (((info >> 8) | (yinfo >> 1)) & 1u) {
                    const uint32_t out = groups[(size_t)(first_group + sl / 2u) * 8u + 6 + (sl & 1u)];
                    atomicAdd(&flag[0], 1u);
                    atomicMin(&flag[1], out);
                }
                const bool ok = nsym <= 4 && !(xnl && (yinfo & 1u)) && !(ynl && x_has_nl) && !((info >> 9) & 1u);
                if (!ok) s_not_dna = 1;
            }
        }
        __syncthreads();
        // long second sequences: their scan, a wavefront each
        for (uint32_t sl = wid; sl < n_slots; sl += 4) {
            const uint32_t info = s_info[sl];
            if ((info & 0x80000000u) || !((info >> 6) & 1u)) continue;
            const uint32_t *rec = groups + (size_t)(first_group + sl / 2u) * 8u;
            const uint32_t h = sl & 1u, ll = rec[4 + h], out = rec[6 + h];
            const uint32_t xsec = (ll >> 15) & 1u, ly = ll >> 16, ynl = (info >> 5) & 1u, xnl = (info >> 4) & 1u;
            const uint8_t *y = raw + (off[2 * (size_t)out + (xsec ^ 1u)] - base);
            bool bad = false, y_has_nl = false;
            for (uint32_t i = lane; i < ly - ynl; i += 64) {
                const uint32_t b = y[i];
                bad |= b == 0u;
                y_has_nl |= b == '\n';
            }
            if (__any(bad) && lane == 0) {
                atomicAdd(&flag[0], 1u);
                atomicMin(&flag[1], out);
            }
            if (xnl && __any(y_has_nl) && lane == 0) s_not_dna = 1;
        }
        __syncthreads();
        const bool all_dna = s_not_dna == 0;
        // ---- pass 2: write the image, coded or as bytes
        for (uint32_t sl = wid; sl < n_slots; sl += 4) {
            const uint32_t info = s_info[sl];
            if (info & 0x80000000u) continue;
            uint32_t *rec = groups + (size_t)(first_group + sl / 2u) * 8u;
            const uint32_t h = sl & 1u, x_dw = rec[h], y_dw = rec[2 + h], ll = rec[4 + h], out = rec[6 + h];
            const uint32_t lx = ll & 0x7fffu, xsec = (ll >> 15) & 1u, ly = ll >> 16;
            const uint8_t *x = raw + (off[2 * (size_t)out + xsec] - base);
            const uint8_t *y = raw + (off[2 * (size_t)out + (xsec ^ 1u)] - base);
            if (all_dna) {
                const uint32_t syms = s_syms[sl];
                const int nsym = (int)(info & 0xfu);
                const uint32_t xnl = (info >> 4) & 1u, ynl = (info >> 5) & 1u, lxs = lx - xnl, lys = ly - ynl;
                const uint32_t lead = gc - lxs; // right-aligned: x'[k] sits in column lead + k
                for (uint32_t i = lane; i < y_dw - x_dw; i += 64) {
                    uint32_t v = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t col = 4 * i + k;
                        const uint32_t c = (col >= lead && col < gc) ? dna_code(x[col - lead], syms, nsym) + 4u * h : 0x0cu;
                        v |= c << (8 * k);
                    }
                    img[x_dw + i] = v;
                }
                for (uint32_t i = lane; i < (ly + 3u) >> 2; i += 64) {
                    uint32_t v = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        uint32_t sc = 31u;
                        if (4 * i + k < lys) {
                            const uint32_t c = dna_code(y[4 * i + k], syms, nsym);
                            if (c < 4u) sc = 8u * (3u - c);
                        }
                        v |= sc << (8 * k);
                    }
                    img[y_dw + i] = v;
                }
                if (lane == 0) rec[4 + h] = lxs | ((xnl & ynl) << 13) | (xsec << 15) | (lys << 16);
            } else {
                bool bad = false; // (already reported in pass 1)
                copy_seq<false>(img + x_dw, y_dw - x_dw, x, lx, nullptr, lane, bad);
                copy_seq<false>(img + y_dw, (ly + 3u) >> 2, y, ly, nullptr, lane, bad);
            }
        }
        if (all_dna && threadIdx.x == 0) waves[4 * (size_t)fw + 3] |= 1u << 16;
        __syncthreads(); // the shared verdict is reset for the next fill wave
    }
}

} // namespace

int agx_sw_pack_dna_launch(const uint8_t *raw, const uint64_t *off, uint64_t base, void *groups, void *waves, uint32_t n_waves,
                           uint32_t n_pairs, uint32_t *img, uint32_t *flag, int n_cu, hipStream_t s)
{
    if (n_waves == 0) return 0;
    static_assert(sizeof(SwWave) == 16 && sizeof(SwGroup2) == 32, "sw_pack_dna reads the records as words");
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)n_waves, (uint64_t)std::max(n_cu, 1) * 32u);
    hipLaunchKernelGGL(sw_pack_dna, dim3(blocks), dim3(256), 0, s, raw, off, base, (uint32_t *)groups, (uint32_t *)waves, n_waves, n_pairs,
                       img, flag);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

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

// Loads this file's code object now: the first launch of a kernel otherwise pays for it (1-2 ms in a fresh process --
// inside hipvers' launch -> scores window).  Called when a batch that will use these kernels is created.
void agx_sw_pack_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&sw_pack_dna));
}
