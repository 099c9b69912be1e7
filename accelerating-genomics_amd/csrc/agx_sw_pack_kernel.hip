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
// The same for the biased packed fill (agx_sw_pk2_kernel.hip), plus its DNA test.  One workgroup per FILL wavefront:
// it looks at every pair of that wave and, when all of them qualify, writes CODES instead of bytes (and says so in
// bit 16 of the wave record's class word):
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
//
// Round 3: both passes work on 16-BYTE CHUNKS spread over all 256 threads.  The sequences of the fill wave (up to 128
// pairs x 2) are laid end to end as a list of chunks (an exclusive scan of their chunk counts in LDS, a thread finds
// its chunk's sequence by bisection), so every thread has one 16-byte load in flight per trip whatever the lengths
// are.  Pass 1 reads aligned 16-byte pieces of the raw sequences: the distinct symbols of x' go into a 256-bit set
// per pair in LDS (a thread adds the at most four symbols of its piece), the byte-0 and newline tests run on the same
// bytes.  Pass 2 produces 16 bytes of the image per trip from one unaligned 16-byte load.  (Round 2 scanned a whole
// sequence per thread -- two threads per pair, forty dependent loads each -- and wrote the image from four byte loads
// per word: 101 us for config 2 with a third of the VALU busy, 3.4x the batch's bytes moved.)
__device__ __forceinline__ uint32_t dna_code(uint32_t b, uint32_t syms, int nsym)
{
    uint32_t c = 4;
#pragma unroll
    for (int k = 3; k >= 0; --k)
        if (k < nsym && b == ((syms >> (8 * k)) & 0xffu)) c = (uint32_t)k;
    return c;
}

struct alignas(16) Quad {
    uint32_t w[4];
};
struct __attribute__((packed, aligned(1))) QuadU {
    uint32_t w[4];
};

constexpr int kPackEntries = 256; // (pair slot, which sequence) of one fill wave: 64 groups x 2 pairs x 2 sequences

// exclusive scan of one value per thread over the 256 threads of the workgroup; s_pre[0..256] receives it (s_pre[256] = total)
__device__ __forceinline__ void block_scan256(uint32_t v, uint32_t *s_pre, uint32_t *s_wsum)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wid) before += s_wsum[w];
    s_pre[threadIdx.x] = before + incl - v;
    if (threadIdx.x == 255) s_pre[256] = before + incl;
    __syncthreads();
}

// the entry e with s_pre[e] <= c < s_pre[e + 1] (entries with no chunks are skipped by construction)
__device__ __forceinline__ uint32_t find_entry(const uint32_t *s_pre, uint32_t c)
{
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = kPackEntries / 2; step; step >>= 1)
        if (s_pre[lo + step] <= c) lo += step;
    return lo;
}

__global__ void __launch_bounds__(256) sw_pack_dna(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ off, uint64_t base,
                                                   uint32_t *__restrict__ groups, uint32_t *__restrict__ waves, uint32_t n_fill_waves,
                                                   uint32_t n_pairs, uint32_t *__restrict__ img, uint32_t *__restrict__ flag)
{
    __shared__ uint32_t s_bits[128][8];        // distinct symbols of x' per pair slot
    __shared__ uint64_t s_src[kPackEntries];   // where the entry's sequence starts in raw
    __shared__ uint32_t s_len[kPackEntries];   // its length without a final newline
    __shared__ uint32_t s_flags[kPackEntries]; // 1 = holds byte 0, 2 = holds a newline (before the final one), 4 = a piece with more than four symbols
    __shared__ uint32_t s_pre[kPackEntries + 1];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_info[128], s_syms[128];
    __shared__ uint32_t s_dst[kPackEntries], s_wlen[kPackEntries];
    __shared__ uint32_t s_not_dna;
    const uint32_t e = threadIdx.x, sl = e >> 1, which = e & 1u; // this thread's entry: which = 0 the shorter sequence x, 1 = y
    for (uint32_t i = threadIdx.x; i < 128 * 8; i += 256) (&s_bits[0][0])[i] = 0;
    for (uint32_t fw = blockIdx.x; fw < n_fill_waves; fw += gridDim.x) {
        // SwWave: {first_group, n_groups | G << 16, steps, class word}
        const uint32_t first_group = waves[4 * (size_t)fw], n_groups = waves[4 * (size_t)fw + 1] & 0xffffu;
        const uint32_t gc = (waves[4 * (size_t)fw + 1] >> 16) * (waves[4 * (size_t)fw + 3] & 0xffffu); // columns of a lane group
        const uint32_t n_slots = n_groups * 2u;
        if (threadIdx.x == 0) s_not_dna = 0;
        // ---- this thread's entry: where its sequence lies, how long it is without a final newline
        uint32_t *rec = groups + (size_t)(first_group + (sl >> 1)) * 8u; // SwGroup2: {x_dw[2], y_dw[2], lx_ly[2], out[2]}
        const uint32_t h = sl & 1u;
        uint32_t len = 0, nl = 0, x_dw = 0, y_dw = 0, lx = 0, ly = 0, xsec = 0, out = n_pairs;
        uint64_t src = 0;
        if (sl < n_slots) {
            x_dw = rec[h], y_dw = rec[2 + h];
            const uint32_t ll = rec[4 + h];
            out = rec[6 + h];
            if (out < n_pairs) {
                lx = ll & 0x7fffu, xsec = (ll >> 15) & 1u, ly = ll >> 16;
                src = off[2 * (size_t)out + (which ? xsec ^ 1u : xsec)] - base;
                len = which ? ly : lx;
                nl = len && raw[src + len - 1] == '\n';
            }
        }
        s_src[e] = src;
        s_len[e] = len - nl;
        s_flags[e] = 0;
        // ---- pass 1: aligned 16-byte pieces of every sequence
        {
            const uint32_t span = (uint32_t)(src & 15u) + (len - nl);
            block_scan256(len - nl ? (span + 15u) >> 4 : 0u, s_pre, s_wsum);
        }
        const uint32_t total1 = s_pre[kPackEntries];
        for (uint32_t c = threadIdx.x; c < total1; c += 256) {
            const uint32_t ce = find_entry(s_pre, c), k = c - s_pre[ce];
            const uint64_t s0 = s_src[ce];
            const uint32_t n = s_len[ce];
            const uint64_t at = (s0 & ~(uint64_t)15) + 16u * k;
            const Quad q = *reinterpret_cast<const Quad *>(raw + at);
            // bytes [lo, hi) of the piece belong to the sequence
            const uint32_t lo = k == 0 ? (uint32_t)(s0 & 15u) : 0u;
            const uint64_t end = s0 + n;
            const uint32_t hi = end - at < 16u ? (uint32_t)(end - at) : 16u;
            uint32_t fl = 0;
            if (ce & 1u) { // y: the two tests
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    const uint32_t b = (q.w[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    const bool in = j >= lo && j < hi;
                    fl |= (in && b == 0u) ? 1u : 0u;
                    fl |= (in && b == '\n') ? 2u : 0u;
                }
            } else { // x: the tests and the piece's distinct symbols
                uint32_t t0 = 0x100, t1 = 0x100, t2 = 0x100, t3 = 0x100, nsym = 0; // 0x100: no byte
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    const uint32_t b = (q.w[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    const bool in = j >= lo && j < hi;
                    fl |= (in && b == 0u) ? 1u : 0u;
                    fl |= (in && b == '\n') ? 2u : 0u;
                    if (in && b != t0 && b != t1 && b != t2 && b != t3) {
                        if (nsym == 0) t0 = b;
                        if (nsym == 1) t1 = b;
                        if (nsym == 2) t2 = b;
                        if (nsym == 3) t3 = b;
                        nsym = min(nsym + 1u, 5u);
                    }
                }
                if (nsym > 4) fl |= 4u;
                uint32_t *bits = s_bits[ce >> 1];
                if (t0 < 0x100) atomicOr(&bits[t0 >> 5], 1u << (t0 & 31u));
                if (t1 < 0x100) atomicOr(&bits[t1 >> 5], 1u << (t1 & 31u));
                if (t2 < 0x100) atomicOr(&bits[t2 >> 5], 1u << (t2 & 31u));
                if (t3 < 0x100) atomicOr(&bits[t3 >> 5], 1u << (t3 & 31u));
            }
            if (fl) atomicOr(&s_flags[ce], fl);
        }
        __syncthreads();
        // ---- the verdict: one thread per pair slot
        {
            const uint32_t my_nl = nl, other_nl = __shfl_xor(nl, 1); // the partner entry (x <-> y) is the neighbouring lane
            if (which == 0 && sl < n_slots) {
                uint32_t info = 0x80000000u, syms = 0; // vacant half: nothing to read, fits either kind of wave
                if (out < n_pairs) {
                    const uint32_t fx = s_flags[e], fy = s_flags[e + 1];
                    uint32_t nsym = 0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        uint32_t m = s_bits[sl][w];
                        s_bits[sl][w] = 0; // ready for the next fill wave
                        while (m) {
                            const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
                            m &= m - 1u;
                            if (nsym < 4) syms |= (32u * (uint32_t)w + bit) << (8 * nsym);
                            ++nsym;
                        }
                    }
                    if (nsym > 4 || (fx & 4u)) nsym = 5;
                    const uint32_t xnl = my_nl, ynl = other_nl;
                    info = nsym | (xnl << 4) | (ynl << 5);
                    if ((fx | fy) & 1u) {
                        atomicAdd(&flag[0], 1u);
                        atomicMin(&flag[1], out);
                    }
                    const bool x_has_nl = (fx >> 1) & 1u, y_has_nl = (fy >> 1) & 1u;
                    const bool ok = nsym <= 4 && !(xnl && y_has_nl) && !(ynl && x_has_nl) && lx < 4096u;
                    if (!ok) s_not_dna = 1;
                }
                s_info[sl] = info;
                s_syms[sl] = syms;
            }
        }
        __syncthreads();
        const bool all_dna = s_not_dna == 0;
        // ---- pass 2: 16 bytes of the image per trip.  An entry's block: x -> [x_dw, y_dw), y -> (ly + 3) / 4 words.
        const uint32_t ndw = out < n_pairs ? (which ? (ly + 3u) >> 2 : y_dw - x_dw) : 0u;
        const uint32_t dst = which ? y_dw : x_dw;
        // (s_len is reused for the block's destination, s_flags for its size in words: pass 1 is over)
        // what pass 2 needs of another thread's entry travels through LDS: destination, words, length as written
        // (s_flags is reused for the block's size in words: its pass-1 contents were read before the barrier above)
        s_flags[e] = ndw;
        s_dst[e] = dst;
        s_wlen[e] = all_dna ? len - nl : len;
        block_scan256((ndw + 3u) >> 2, s_pre, s_wsum);
        const uint32_t total2 = s_pre[kPackEntries];
        for (uint32_t c = threadIdx.x; c < total2; c += 256) {
            const uint32_t ce = find_entry(s_pre, c), k = c - s_pre[ce];
            const uint32_t csl = ce >> 1, cy = ce & 1u, ch = csl & 1u;
            const uint32_t n = s_wlen[ce], nw = s_flags[ce];
            const uint64_t s0 = s_src[ce];
            // column 16 k of the block is byte `first` of the sequence (x of a coded wave is right-aligned in gc columns)
            const int32_t first = (int32_t)(16u * k) - ((all_dna && !cy) ? (int32_t)(gc - n) : 0);
            Quad q = {{0, 0, 0, 0}};
            if (first + 16 > 0 && first < (int32_t)n) {
                const QuadU u = *reinterpret_cast<const QuadU *>(raw + (int64_t)s0 + first);
                q.w[0] = u.w[0], q.w[1] = u.w[1], q.w[2] = u.w[2], q.w[3] = u.w[3];
            }
            uint32_t o[4];
            if (all_dna) {
                const uint32_t syms = s_syms[csl];
                const int nsym = (int)(s_info[csl] & 0xfu);
#pragma unroll
                for (uint32_t w = 0; w < 4; ++w) {
                    uint32_t v = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const int32_t idx = first + (int32_t)(4 * w + j);
                        const bool in = idx >= 0 && idx < (int32_t)n;
                        const uint32_t code = dna_code((q.w[w] >> (8 * j)) & 0xffu, syms, nsym);
                        const uint32_t byte = cy ? ((in && code < 4u) ? 8u * (3u - code) : 31u) : (in ? code + 4u * ch : 0x0cu);
                        v |= byte << (8 * j);
                    }
                    o[w] = v;
                }
            } else {
#pragma unroll
                for (uint32_t w = 0; w < 4; ++w) {
                    uint32_t v = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const int32_t idx = first + (int32_t)(4 * w + j);
                        if (idx >= 0 && idx < (int32_t)n) v |= q.w[w] & (0xffu << (8 * j));
                    }
                    o[w] = v;
                }
            }
            uint32_t *d = img + s_dst[ce] + 4u * k;
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w)
                if (4u * k + w < nw) d[w] = o[w];
        }
        if (all_dna) {
            if (which == 0 && sl < n_slots && out < n_pairs) {
                const uint32_t info = s_info[sl], xnl = (info >> 4) & 1u, ynl = (info >> 5) & 1u;
                rec[4 + h] = (lx - xnl) | ((xnl & ynl) << 13) | (xsec << 15) | ((ly - ynl) << 16);
            }
            if (threadIdx.x == 0) waves[4 * (size_t)fw + 3] |= 1u << 16;
        }
        __syncthreads(); // the shared state is rewritten for the next fill wave
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
