// Smith-Waterman fill, packed variant, second formulation ("biased"): the schedule and data layout of
// agx_sw_pk_kernel.hip -- two alignment pairs per lane group, pair A in the low and pair B in the
// high 16 bits of every state register -- with the cell rewritten around what gfx950 issues cheaply
// (tools/valu_microbench2.hip, valu_microbench3.hip: every VALU instruction of this mix costs one 4.2-cycle
// slot per wave64, so the instruction COUNT is what matters):
//
//   * Every value is kept as an UNSIGNED half with a bias B added: stored = true + B >= 0 always, constants
//     are subtracted (never added as two's complement), so one 32-bit add serves both halves and no carry or
//     borrow crosses bit 16.  The vertical gap state is kept clamped at zero (P~ = max(P, 0): a negative P
//     never reaches H -- H >= 0 -- and its successors P - 1, P - 2, ... are negative too, so
//     max(H_up + gf, P~_up + ge, 0) = max(P_new, 0) exactly).  That clamp is also what delivers the zero
//     floor of antidiagonalSmithWaterman.c:333: H = max(P~, Q, H_diag + s) >= 0.
//   * gfx950 has a packed three-input maximum, v_pk_maximum3_f16.  With B >= 1024 + |gf| + delta and all
//     values below 0x7c00 every stored half is the bit pattern of a positive NORMAL half-precision number,
//     and for those the floating-point order is the integer order: the instruction is an exact unsigned
//     max3 here.  It folds the clamp into the gap maximum and the two maxima of :333 into one.
//
//   general plain cell, per two cells (10.5 instructions):
//                   e' = max3(z_up, e - |ge|, B)         v_sub_u32, v_pk_maximum3_f16     (:313, clamped)
//                   f  = max(z_left, f - |ge|)           v_sub_u32, v_pk_max_u16          (:321)
//                   m  = min(x ^ y, delta)               v_xor_b32, v_pk_min_u16          (:332, match test)
//                   u  = (z_diag + hd) - m               v_add_u32, v_sub_u32             (:332)
//                   H' = max3(e', f, u)                  v_pk_maximum3_f16                (:333)
//                   z  = H' - |gf|                       v_sub_u32
//                   best = max3(best, z, z_next)         half a v_pk_maximum3_f16         (:335)
//   DNA-coded rising cell (7.5): the match term is one v_perm_b32 table lookup for both pairs, fused with the
//   diagonal add into v_add3_u32 (FAST, below), and stored values rise by |ge| per step so that the vertical gap
//   needs no subtraction (RISE, below).  12 in agx_sw_pk_kernel.hip.
//
// The host picks this kernel when the scoring and the longest shorter side keep every stored half in
// [0x0400, 0x7c00) (agx_sw.cpp; always true for the reference's +1/-1/-3/-1 up to 2560 columns);
// scores are bit-identical to the other kernels and to the reference.
#include "agx_sw.h"
#include <type_traits>

namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 as_v(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t umax2(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_v(a), as_v(b))); }
__device__ __forceinline__ uint32_t umin2(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_min(as_v(a), as_v(b))); }
// exact unsigned max3 per half for patterns of positive normal half-precision numbers (see above): v_pk_maximum3_f16.
// Through the builtin rather than inline assembly: after every asm block the compiler's hazard pass pads with an
// s_nop (17 a step at 38 columns).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c)
{
    const f16x2 x = __builtin_bit_cast(f16x2, a), y = __builtin_bit_cast(f16x2, b), z = __builtin_bit_cast(f16x2, c);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}
// a wave-uniform constant forced into a VGPR: with an SGPR or literal operand v_add/v_sub_u32 fall back
// to the 4-cycle rate ("v_subrev_u32 SGPR constant" in the microbenchmark)
__device__ __forceinline__ uint32_t in_vgpr(uint32_t s)
{
    uint32_t r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(s));
    return r;
}

// ---- the head of a step: this row's symbols and the four values a lane takes over from its left neighbour ----
// A lane group's first lane (`start`) takes fresh values -- the row symbol / table, H = 0 and Q = -inf of column 0
// (antidiagonalSmithWaterman.c:299-306) -- every other lane what its left neighbour held one step ago.  Written
// out: v_cndmask_b32 with a DPP wave_shr:1 source does the shift and the choice in ONE instruction (the compiler's
// own lowering was v_mov_dpp + v_cndmask per value, plus a compare against the row count per pair, a mask and a
// shift for the symbol: 35 instructions a step next to the 361 of the cells at C = 38; these blocks have 6 and 5).
// The DPP reads come at least three instructions after anything inside the block wrote a register (the gfx9 rule
// is two wait states between a VALU write and a DPP read of the same register); what they read from outside was
// written before the block began.
#define AGX_DPP_TAKE " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"

// FAST: the image holds y as SHIFT COUNTS: s = 8 * (3 - code), or 31 for "matches nothing" (a symbol x does not
// contain, and every row beyond the sequence).  (delta << 24) >> s is the row's table: delta in the byte of the code
// that matches, 0 elsewhere; delta < 128, so s = 31 leaves nothing.  The byte of the quad is picked by SDWA.
#define AGX_FAST_HEAD(BYTE)                                                                                                  \
    asm("v_lshrrev_b32_sdwa %4, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" BYTE " src1_sel:DWORD\n"               \
        "v_lshrrev_b32_sdwa %5, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" BYTE " src1_sel:DWORD\n"               \
        "s_mov_b64 vcc, %9\n"                                                                                                \
        "v_cndmask_b32_dpp %0, %10, %12, vcc" AGX_DPP_TAKE "v_cndmask_b32_dpp %1, %11, %12, vcc" AGX_DPP_TAKE                \
        "v_cndmask_b32_dpp %2, %2, %4, vcc" AGX_DPP_TAKE "v_cndmask_b32_dpp %3, %3, %5, vcc" AGX_DPP_TAKE                    \
        : "=&v"(zl), "=&v"(fl), "+v"(ta), "+v"(tb), "=&v"(ma), "=&v"(mb)                                                     \
        : "v"(rowsA), "v"(rowsB), "v"(kv), "s"(start_mask), "v"(z_last), "v"(f_last), "v"(z0v)                               \
        : "vcc")

template <int K>
__device__ __forceinline__ void fast_head(uint32_t &zl, uint32_t &fl, uint32_t &ta, uint32_t &tb, uint32_t rowsA, uint32_t rowsB,
                                          uint32_t kv, uint64_t start_mask, uint32_t z_last, uint32_t f_last, uint32_t z0v)
{
    uint32_t ma, mb;
    if constexpr (K == 1)
        AGX_FAST_HEAD("BYTE_1");
    else if constexpr (K == 2)
        AGX_FAST_HEAD("BYTE_2");
    else if constexpr (K == 3)
        AGX_FAST_HEAD("BYTE_3");
    else
        AGX_FAST_HEAD("BYTE_0");
}

// general: the image holds the symbols themselves (zero beyond a sequence -- byte 0 is no symbol, and padding
// COLUMNS carry 0x100 << shift, which is no byte either); byte k of the two quads -> {a, 0, b, 0} << shift
__device__ __forceinline__ void bytes_head(uint32_t &zl, uint32_t &fl, uint32_t &yc, uint32_t rowsA, uint32_t rowsB, uint32_t sel,
                                           uint32_t sh_sym, uint64_t start_mask, uint32_t z_last, uint32_t f_last, uint32_t z0v)
{
    uint32_t fresh;
    asm("v_perm_b32 %3, %5, %4, %6\n"
        "v_lshlrev_b32 %3, %7, %3\n"
        "s_mov_b64 vcc, %8\n"
        "v_cndmask_b32_dpp %0, %9, %11, vcc" AGX_DPP_TAKE "v_cndmask_b32_dpp %1, %10, %11, vcc" AGX_DPP_TAKE
        "v_cndmask_b32_dpp %2, %2, %3, vcc" AGX_DPP_TAKE
        : "=&v"(zl), "=&v"(fl), "+v"(yc), "=&v"(fresh)
        : "v"(rowsA), "v"(rowsB), "s"(sel), "v"(sh_sym), "s"(start_mask), "v"(z_last), "v"(f_last), "v"(z0v)
        : "vcc");
}

// FAST = the wave's pairs all passed the pack kernel's DNA test (agx_sw_pack_kernel.hip, sw_pack_dna): at most four
// distinct symbols in the shorter sequence, a trailing newline sentinel at most at the very end of either.  The
// image then holds CODES, the sentinels stripped:
//   x  as v_perm_b32 selector bytes -- code 0..3 for pair A, 4 + code for pair B, 0x0c (the constant 0) for a padding
//      column -- RIGHT-aligned in the group's G * C columns: padding columns on the left behave exactly like column
//      0 (H = 0, Q = -inf), and the last symbol always sits in the last column of the group's last lane;
//   y  as shift counts (fast_head).
// The row table M (byte c = delta if the row symbol is x's code c, else 0) travels down the lanes, and ONE
// v_perm_b32 per column yields the match bonus of BOTH pairs (selector byte 0 picks M_A[code], byte 2 picks
// M_B[code]) where the general cell spends v_xor_b32 + v_pk_min_u16: 9.5 instead of 10.5 instructions per two
// cells.  The stripped sentinels are put back at the end: a final newline aligns with nothing but the other
// sequence's final newline, so the score is max(best, H[lx'][ly'] + match) when both had one
// (antidiagonalSmithWaterman.c:229-247 keeps the newline as a symbol), best otherwise.  H[lx'][ly']
// is what the group's last lane holds in its last column after step ly' - 1 + (G - 1).
//
// RISE = "rising offsets": every stored value additionally carries an offset that grows by |ge| per step, the same in
// all lanes: r(t) = (t + 2) |ge|, on z of step t; r(t - 1) on e, f and H of step t.  The vertical gap then needs no
// subtraction at all -- P_new + r(t) = max(z_up + r(t), (P + r(t-1))) since r(t) - |ge| = r(t - 1) -- the horizontal gap
// subtracts after its maximum instead of before, the diagonal is unchanged (z_diag carries r(t - 1), which is H's
// offset), and z = H - (|gf| - |ge|).  One v_sub_u32 less per two cells; per STEP the floor, the column-0 value
// (wave-uniform: a scalar add) and the running maximum rise by |ge|.  What a lane takes over from its left
// neighbour was made one step earlier and is |ge| behind: in the lane's first column that lag cancels the horizontal
// gap's subtraction (max(z_left + |ge|, f_left + |ge|) - |ge|), and the diagonal adds |ge| through its constant.
// The host asks for this variant when B + the largest score + (steps + 2) |ge| stays
// below 0x7c00 (agx_sw.cpp) -- rows up to about 27 000 with the reference's scores; beyond, the plain cell.
//
// KC = column classes of the rising cell (0: plain cell, 1: rising, 4: rising with classes).  With KC = 4 column j of a
// lane additionally carries (j mod 4) |ge|: from one column to the next the offset rises by |ge|, which is exactly what
// the horizontal gap subtracts -- f = max(z_left, f) with no subtraction, except where the class wraps (every fourth
// column: minus 4 |ge|) -- and the diagonal adds |ge| through its constant (minus 3 |ge| at a wrap: the host asks for
// this variant only when mismatch + |gf| >= 3 |ge|, so that constant is not negative).  The floor and the running
// maximum exist once per class (each rises by |ge| per step; a class's maximum takes its columns two at a time), what a
// lane hands to its right neighbour loses the last column's class offset on arrival, and the classes' maxima are
// brought to one offset at the end.  Per column (two cells) 6 + 1/4 + 1/2 instructions and ten per step, against
// 7 + 1/2 and two.
template <int C, bool FAST, int KC>
__device__ __forceinline__ void pk2_fill(const SwParams &prm, const uint32_t *__restrict__ img, const SwGroup2 &g, const SwWave &w,
                                         int32_t *__restrict__ scores, int lane, int G, int gl, bool active, bool start, bool feeder)
{
    constexpr int XW = (C + 3) / 4; // dwords holding this lane's C symbols
    constexpr bool RISE = KC > 0;
    constexpr int NK = KC > 1 ? KC : 1;               // floors / running maxima kept
    constexpr int kEnd = KC > 1 ? (C - 1) % KC : 0;   // class of the lane's last column
    const uint32_t sh_sym = prm.shift;         // general: symbols live as byte << shift
    const uint32_t col_pad = 0x100u << sh_sym; // never equals (byte << shift)
    const uint32_t ge = in_vgpr(prm.age2), gf = in_vgpr(RISE ? prm.agf2 - prm.age2 : prm.agf2); // |ge|; |gf| (RISE: |gf| - |ge|)
    const uint32_t hd = in_vgpr(FAST ? prm.hd2 - prm.delta2 : prm.hd2);                // mismatch + |gf| / match + |gf|
    const uint32_t bias = prm.bias2, delta = prm.delta2;
    const uint32_t z0 = prm.bias2 - prm.agf2; // H = 0 as the state both gap recurrences read (z = H + gf), both halves
    const uint32_t hd0 = in_vgpr((FAST ? prm.hd2 - prm.delta2 : prm.hd2) + (RISE ? prm.age2 : 0u)); // first column's diagonal; KC > 1: every non-wrapping one's
    const uint32_t hdw = in_vgpr((FAST ? prm.hd2 - prm.delta2 : prm.hd2) + prm.age2 - (uint32_t)NK * prm.age2); // KC > 1: a wrapping column's diagonal
    const uint32_t ge_wrap = in_vgpr((uint32_t)NK * prm.age2), c_end = in_vgpr((uint32_t)kEnd * prm.age2);
    uint32_t zb = (RISE ? z0 + prm.age2 : z0) + (uint32_t)kEnd * prm.age2; // column 0 as the first lane takes it over: z0 + r(t - 1) (+ what every lane takes off on arrival)
    uint32_t floorv[NK];                                          // P~ >= 0 at H's offset: B + r(t - 1) (+ class); in VGPRs: as an
#pragma unroll                                                    // SGPR operand it drew an s_nop after every group of four
    for (int k = 0; k < NK; ++k) floorv[k] = (RISE ? bias + prm.age2 : bias) + (uint32_t)k * prm.age2;
    const uint32_t z_init = RISE ? z0 + prm.age2 : z0;            // H = 0 one step before the first: z0 + r(-1)
    const uint32_t kv = in_vgpr((prm.delta2 & 0xffu) << 24); // FAST: the table source
    const uint64_t start_mask = __ballot(start);
    const uint32_t lx_mask = FAST ? 0xfffu : 0x7fffu;
    const int lxA = (int)(g.lx_ly[0] & lx_mask), lxB = (int)(g.lx_ly[1] & lx_mask);
    const int lyA = (int)(g.lx_ly[0] >> 16), lyB = (int)(g.lx_ly[1] >> 16);
    const int nqA = (lyA + 3) >> 2, nqB = (lyB + 3) >> 2;

    // this lane's C symbols of both short sequences -> one register per column:
    //   general: (a << shift) | (b << shift) << 16;   FAST: the v_perm_b32 selector {sel_a, 0x0c, sel_b, 0x0c}
    uint32_t xq[C];
    {
        const uint32_t o = (uint32_t)gl * C, d0 = o >> 2, sh = o & 3u;
        // FAST: a vacant half (its record points at the zero block, offset 0) and idle lanes read zeros: all padding
        const uint32_t fillA = (FAST && !(active && g.x_dw[0])) ? 0x0c0c0c0cu : 0u;
        const uint32_t fillB = (FAST && !(active && g.x_dw[1])) ? 0x0c0c0c0cu : 0u;
        uint32_t ra[XW + 1], rb[XW + 1];
#pragma unroll
        for (int k = 0; k <= XW; ++k) {
            ra[k] = (active ? img[g.x_dw[0] + d0 + k] : 0u) | fillA;
            rb[k] = (active ? img[g.x_dw[1] + d0 + k] : 0u) | fillB;
        }
#pragma unroll
        for (int k = 0; k < XW; ++k) {
            const uint32_t a = __builtin_amdgcn_alignbyte(ra[k + 1], ra[k], sh);
            const uint32_t b = __builtin_amdgcn_alignbyte(rb[k + 1], rb[k], sh);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * k + i < C) {
                    const uint32_t ca = (a >> (8 * i)) & 0xffu, cb = (b >> (8 * i)) & 0xffu;
                    uint32_t v;
                    if constexpr (FAST)
                        v = ca | 0x0c00u | (cb << 16) | 0x0c000000u;
                    else {
                        const int col = (int)o + 4 * k + i;
                        v = (col < lxA ? ca << sh_sym : col_pad) | ((col < lxB ? cb << sh_sym : col_pad) << 16);
                    }
                    // opaque to the compiler: left visible, it keeps the length tests as lane masks and rebuilds
                    // every register in every step to save registers
                    asm volatile("" : "+v"(v));
                    xq[4 * k + i] = v;
                }
        }
    }

    const uint32_t *ypA = img + g.y_dw[0], *ypB = img + g.y_dw[1];
    const uint32_t no_row = FAST ? 0x1f1f1f1fu : 0u; // rows beyond the sequence match nothing
    auto quadA = [&](int q) -> uint32_t { return (feeder && q < nqA) ? ypA[q] : no_row; };
    auto quadB = [&](int q) -> uint32_t { return (feeder && q < nqB) ? ypB[q] : no_row; };

    // state per owned column, both pairs packed, biased: z = H + gf + B and e = max(P, 0) + B
    uint32_t z[C], e[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        z[j] = z_init + (uint32_t)(j % NK) * prm.age2;
        e[j] = bias + (uint32_t)(j % NK) * prm.age2; // anything up to the first floor
    }
    // the horizontal gap state needs no clamp: Q >= z_left >= gf always; "no gap open yet" is Q = gf,
    // whose successor gf + ge loses against every z_left
    uint32_t z_last = z_init + (uint32_t)kEnd * prm.age2, f_last = z_last, diag_in = z0;
    uint32_t best[NK]; // :335, one per class
#pragma unroll
    for (int k = 0; k < NK; ++k) best[k] = z_init + (uint32_t)k * prm.age2;
    uint32_t yc = 0;         // general: the row symbols of both pairs
    uint32_t ta = 0, tb = 0; // FAST: the row tables of pair A / pair B

    // FAST: the step after which the group's last lane holds H[lx'][ly'] in its last column (bit 13 of the record:
    // both sequences ended with the sentinel); a stripped side that is empty leaves the corner at H = 0
    const bool last = active && gl == G - 1;
    const bool nlA = FAST && last && ((g.lx_ly[0] >> 13) & 1u), nlB = FAST && last && ((g.lx_ly[1] >> 13) & 1u);
    const int capA_t = (nlA && lxA > 0 && lyA > 0) ? lyA + G - 2 : -1, capB_t = (nlB && lxB > 0 && lyB > 0) ? lyB + G - 2 : -1;
    uint32_t cornerA = z_init + (uint32_t)kEnd * prm.age2, cornerB = cornerA; // (taken from the last column: its class offset comes off at the end)

    uint32_t a0 = quadA(0), a1 = quadA(1), a2 = quadA(2);
    uint32_t b0 = quadB(0), b1 = quadB(1), b2 = quadB(2);
    const int steps = (int)w.steps;
    uint32_t rowsA = 0, rowsB = 0;
    int t = 0;

    // K = which byte of the quads this step reads (the tail loop shifts the quads instead: K = 0)
    auto step = [&](auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        uint32_t zl, fl;
        if constexpr (FAST)
            fast_head<K>(zl, fl, ta, tb, rowsA, rowsB, kv, start_mask, z_last, f_last, zb);
        else
            bytes_head(zl, fl, yc, rowsA, rowsB, 0x0c040c00u + 0x00010001u * K, sh_sym, start_mask, z_last, f_last, zb);
        if constexpr (KC > 1 && kEnd > 0) { // what the left neighbour's last column carried for its class comes off
            zl -= c_end;
            fl -= c_end;
        }
        if constexpr (RISE) {
#pragma unroll
            for (int k = 0; k < NK; ++k) best[k] += ge;
        }
        uint32_t zd = diag_in; // H[r-1][first column - 1] + gf
        diag_in = zl;
        uint32_t zleft = zl, f = fl;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool wrap = KC > 1 && j > 0 && j % NK == 0; // compile-time after unrolling
            const uint32_t up = z[j];
            uint32_t ev; // reference P, :313, clamped at 0;  reference Q, :321
            if constexpr (RISE) {
                ev = umax3(up, e[j], floorv[j % NK]);
                f = umax2(zleft, f);
                if (KC == 1 && j > 0) f -= ge; // (first column: see above)
                if (wrap) f -= ge_wrap;
            } else {
                ev = umax3(up, e[j] - ge, bias);
                f = umax2(zleft, f - ge);
            }
            const uint32_t hdc = KC > 1 ? (wrap ? hdw : hd0) : (j ? hd : hd0);
            uint32_t u;                                        // H_diag + match / + mismatch, :332
            if constexpr (FAST)
                u = (zd + hdc) + __builtin_amdgcn_perm(tb, ta, xq[j]); // mismatch, plus delta on a match
            else
                u = (zd + hdc) - umin2(xq[j] ^ yc, delta); // match, minus delta on a mismatch
            const uint32_t v = umax3(ev, f, u);                // :333 (ev >= B carries the zero floor)
            const uint32_t zn = v - gf;
            e[j] = ev;
            z[j] = zn;
            zd = up;
            zleft = zn;
        }
        // :335 -- every class's maximum takes that class's columns two at a time
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int j = k; j < C; j += 2 * NK) {
                if (j + NK < C)
                    best[k] = umax3(best[k], z[j], z[j + NK]);
                else
                    best[k] = umax2(best[k], z[j]);
            }
        }
        if constexpr (FAST) {
            cornerA = t == capA_t ? zleft : cornerA;
            cornerB = t == capB_t ? zleft : cornerB;
        }
        z_last = zleft;
        f_last = f;
        if constexpr (RISE) {
#pragma unroll
            for (int k = 0; k < NK; ++k) floorv[k] += prm.age2;
            zb += prm.age2;
        }
        ++t;
    };

    const int quads = steps >> 2;
    for (int q = 0; q < quads; ++q) {
        rowsA = a0;
        a0 = a1;
        a1 = a2;
        a2 = quadA(q + 3);
        rowsB = b0;
        b0 = b1;
        b1 = b2;
        b2 = quadB(q + 3);
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
    }
    rowsA = a0;
    rowsB = b0;
#pragma unroll 1
    while (t < steps) {
        step(std::integral_constant<int, 0>{});
        rowsA >>= 8;
        rowsB >>= 8;
    }

    if constexpr (FAST) {
        // the stripped sentinels: when both sequences ended with one, the two newlines align after the corner cell.
        // H_corner + match as a z value is z_corner + match.
        const uint32_t match2 = prm.hd2 - prm.agf2; // (match + |gf|) - |gf| in both halves
        const uint32_t ge1 = RISE ? prm.age2 & 0xffffu : 0u; // the corner was taken at offset r(cap_t), best stands at r(steps - 1)
        const uint32_t off_end = (uint32_t)kEnd * (prm.age2 & 0xffffu);
        uint32_t cand = best[0];
        if (nlA) cand = (cand & 0xffff0000u) | ((cornerA + match2 + (uint32_t)(steps - 1 - capA_t) * ge1 - off_end) & 0xffffu);
        if (nlB) cand = (cand & 0xffffu) | ((cornerB + match2 + (((uint32_t)(steps - 1 - capB_t) * ge1 - off_end) << 16)) & 0xffff0000u);
        best[0] = umax2(best[0], cand);
    }
#pragma unroll
    for (int k = 1; k < NK; ++k) best[0] = umax2(best[0], best[k] - (uint32_t)k * prm.age2); // the classes at one offset
    uint32_t bestv = best[0];
    // max over the group's lanes (G need not be a power of two), both halves at once
    for (int o = 1; o < G; o <<= 1) {
        const uint32_t other = (uint32_t)__shfl_down((int)bestv, o);
        if (gl + o < G) bestv = umax2(bestv, other);
    }
    {
        const int off = (int)(z0 & 0xffffu) + (RISE ? (steps + 1) * (int)(prm.age2 & 0xffffu) : 0); // stored value of H = 0, at r(steps - 1)
        // The wave's results move to its first lanes -- lane i takes group i's two scores from that group's first lane --
        // and go out from there: neighbouring pairs as ONE 8-byte store per group, so that in a batch planned in file order
        // adjacent lanes write adjacent bytes and the wave's scores leave as one request (one PCIe write when the scores
        // array is the caller's page-locked one, agx_sw_batch_bind_scores; the spare slot n_pairs a vacant half points at
        // does not exist there).
        const int src = (lane * G) & 63;
        const int sa = __shfl((int)(bestv & 0xffffu) - off, src), sb = __shfl((int)(bestv >> 16) - off, src);
        const uint32_t oa = (uint32_t)__shfl((int)g.out[0], src), ob = (uint32_t)__shfl((int)g.out[1], src);
        if (lane < (int)w.n_groups) {
            if (ob == oa + 1u && !(oa & 1u) && ob < prm.n_out)
                *reinterpret_cast<int2 *>(scores + oa) = make_int2(sa, sb);
            else {
                scores[oa] = sa;
                if (ob < prm.n_out) scores[ob] = sb;
            }
        }
    }
}

template <int C, int KC>
__device__ __forceinline__ void pk2_body(const SwParams &prm, const uint32_t *__restrict__ img, const SwGroup2 *__restrict__ groups,
                                         const SwWave w, int32_t *__restrict__ scores)
{
    static_assert(C % 2 == 0, "the running maximum takes two columns per instruction");
    // column classes cost ten instructions a step and save three quarters of one per column: narrow lanes do without
    constexpr int KCC = KC > 1 && C < 14 ? 1 : KC;
    const int lane = threadIdx.x & 63;
    const int G = w.G;
    const int grp = lane / G;
    const int gl = lane - grp * G;
    const bool active = grp < (int)w.n_groups;
    const bool start = gl == 0;
    const bool feeder = active && start;

    SwGroup2 g;
#pragma unroll
    for (int k = 0; k < 2; ++k) g.x_dw[k] = g.y_dw[k] = g.lx_ly[k] = g.out[k] = 0;
    if (active) g = groups[w.first_group + grp];
    // bit 16 of the wave record's class word: set by the pack kernel when every pair of the wave is DNA-coded
    if (__builtin_amdgcn_readfirstlane(w.reserved >> 16) & 1u)
        pk2_fill<C, true, KCC>(prm, img, g, w, scores, lane, G, gl, active, start, feeder);
    else
        pk2_fill<C, false, KCC>(prm, img, g, w, scores, lane, G, gl, active, start, feeder);
}

template <int C, int KC>
__global__ void __launch_bounds__(256) sw_fill_pk2(const SwParams prm, const uint32_t *__restrict__ img,
                                                   const SwGroup2 *__restrict__ groups,
                                                   const SwWave *__restrict__ waves, uint32_t n_waves,
                                                   int32_t *__restrict__ scores)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= n_waves) return;
    pk2_body<C, KC>(prm, img, groups, waves[wave], scores);
}

// Mixed batches: ONE launch for every lane-tiling class.  Each wavefront reads its class (columns per lane)
// from its record and runs that class's fill; the kernel is allocated the registers of the widest class
// (202-206 VGPRs as its code objects state them -- tools/kernel_resources.py --, two waves per SIMD: the fill is bound by
// VALU issue, not by occupancy).  Against one launch
// per class this (a) lets the planner use every width, so padding shrinks, (b) dispatches the waves of ALL
// classes longest first, (c) has no stream fork/join and no per-launch ramp.
template <int KC>
__global__ void __launch_bounds__(256) sw_fill_pk2_any(const SwParams prm, const uint32_t *__restrict__ img,
                                                       const SwGroup2 *__restrict__ groups,
                                                       const SwWave *__restrict__ waves, uint32_t n_waves,
                                                       int32_t *__restrict__ scores)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= n_waves) return;
    const SwWave w = waves[wave];
    switch (__builtin_amdgcn_readfirstlane(w.reserved) & 0xffffu) { // columns per lane of this wave
#define AGX_SW_CASE(CC) \
    case CC: pk2_body<CC, KC>(prm, img, groups, w, scores); break;
        AGX_SW_FOR_EACH_CLASS(AGX_SW_CASE)
#undef AGX_SW_CASE
    default: break;
    }
}

template <int C, int KC>
int launch(const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves, uint32_t n_waves,
           int32_t *scores, hipStream_t s)
{
    const uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL((sw_fill_pk2<C, KC>), dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace

int agx_sw_pk2_launch_any(int rising, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves, uint32_t n_waves,
                          int32_t *scores, hipStream_t s)
{
    if (n_waves == 0) return 0;
    const uint32_t blocks = (n_waves + 3) / 4;
    if (rising == 4)
        hipLaunchKernelGGL(sw_fill_pk2_any<4>, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
    else if (rising)
        hipLaunchKernelGGL(sw_fill_pk2_any<1>, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
    else
        hipLaunchKernelGGL(sw_fill_pk2_any<0>, dim3(blocks), dim3(256), 0, s, prm, img, groups, waves, n_waves, scores);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int agx_sw_pk2_launch_class(int cols_per_lane, int rising, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups,
                            const SwWave *waves, uint32_t n_waves, int32_t *scores, hipStream_t s)
{
    if (n_waves == 0) return 0;
    switch (cols_per_lane) {
#define AGX_SW_CASE(CC) \
    case CC: return rising == 4 ? launch<CC, 4>(prm, img, groups, waves, n_waves, scores, s) : rising ? launch<CC, 1>(prm, img, groups, waves, n_waves, scores, s) : launch<CC, 0>(prm, img, groups, waves, n_waves, scores, s);
        AGX_SW_FOR_EACH_CLASS(AGX_SW_CASE)
#undef AGX_SW_CASE
    default: return -2;
    }
}

// Loads this file's code object now: the first launch of a kernel otherwise pays for it (1-2 ms in a fresh process --
// inside hipvers' launch -> scores window).  Called when a batch that will use these kernels is created.
void agx_sw_pk2_preload()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&sw_fill_pk2_any<4>));
}
