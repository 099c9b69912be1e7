// Host side of the Smith-Waterman path: validation, lane-tiling choice, packing into the
// device image, launches, multi-device sharding (include/agx.h, "Smith-Waterman" section).
#include "agx_sw.h"

#include <algorithm>
#include <chrono>
#include <string>
#include <thread>

#include "agx_internal.h"
#include "agx_parallel.h"

namespace {

struct Tiling {
    int cls; // index into kSwClasses
    int G;
};

// AGX_SW_KERNEL=i32 selects the scalar int32 kernel, pk1 the first packed int16 formulation
// (agx_sw_pk_kernel.hip); default is the biased packed one (agx_sw_pk2_kernel.hip, two pairs per lane
// group) whenever its value range allows.  Scores are identical.
int kernel_choice() // 0 = int32, 1 = packed (signed halves), 2 = packed (biased unsigned halves)
{
    static const int v = [] {
        const char *e = getenv("AGX_SW_KERNEL");
        if (e && strcmp(e, "i32") == 0) return 0;
        if (e && strcmp(e, "pk1") == 0) return 1;
        return 2;
    }();
    return v;
}
bool use_packed_kernel() { return kernel_choice() != 0; }

// Lane time a pair costs under tiling (class ci, G): steps * C * 64 / floor(64 / G) padded
// cells (the lanes of a wave that cannot host another group are charged to the pair), weighted
// by the measured per-cell cost of the class.
// beta: lanes' worth of extra weight on a wave's own duration (steps * C), which favours spreading long
// pairs over more lanes.  0 in the throughput regime; the planner raises it for batches whose waves
// would fill the chip a little more than once (see create_batch).  AGX_SW_TAIL_BETA overrides (experiments).
inline double tail_beta_override()
{
    static const double v = [] {
        const char *e = getenv("AGX_SW_TAIL_BETA");
        return e ? atof(e) : -1.0;
    }();
    return v;
}

inline double tiling_cost(bool packed, int ly, int ci, int G, double beta)
{
    const double wgt = packed ? kSwPkClassCost[ci] : kSwClassCost[ci];
    return (double)(ly + G - 1) * kSwClasses[ci] * ((64.0 / (double)(64 / G)) * wgt + beta);
}

// Tuning knob for experiments (not part of the ABI): AGX_SW_MAX_C caps the columns per lane.
int max_cols_per_lane()
{
    static const int v = [] {
        const char *e = getenv("AGX_SW_MAX_C");
        const int n = e ? atoi(e) : 0;
        return n >= 4 ? n : AGX_SW_MAX_COLS_PER_LANE;
    }();
    return v;
}

// AGX_SW_FORCE_C pins the class (calibration runs only; pairs that do not fit 64 lanes fail).
int force_cols_per_lane()
{
    static const int v = [] {
        const char *e = getenv("AGX_SW_FORCE_C");
        return e ? atoi(e) : 0;
    }();
    return v;
}

// allowed: bit ci set = class ci may be used; *cost_out: the lane time estimate of the choice
Tiling choose_tiling(bool packed, int lx, int ly, uint32_t allowed = ~0u, double *cost_out = nullptr, double beta = 0.0)
{
    Tiling best{-1, 0};
    double best_cost = 0;
    for (int ci = 0; ci < kSwNumClasses; ++ci) {
        if (!((allowed >> ci) & 1u)) continue;
        const int C = kSwClasses[ci];
        const int G = (lx + C - 1) / C;
        if (G > 64) continue;
        if (C > max_cols_per_lane() && best.cls >= 0) continue;
        if (force_cols_per_lane() && C != force_cols_per_lane()) continue;
        if ((packed ? kSwPkClassCost[ci] : kSwClassCost[ci]) == 0) continue; // class not built for this kernel
        const double c = tiling_cost(packed, ly, ci, G, beta);
        if (best.cls < 0 || c < best_cost || (c == best_cost && C > kSwClasses[best.cls])) {
            best = Tiling{ci, G};
            best_cost = c;
        }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}

// AGX_SW_MAX_CLASSES: upper bound on the kernel classes a mixed batch may spread over (default 6)
int max_classes()
{
    static const int v = [] {
        const char *e = getenv("AGX_SW_MAX_CLASSES");
        const int n = e ? atoi(e) : 0;
        return n > 0 ? n : 6;
    }();
    return v;
}

// Uniform batches (most pairs share one shape, e.g. fixed-length reads): every wave of that shape
// costs the same, so the launch lasts ceil(waves / SIMDs) wave-times -- the tiling is chosen for the
// whole shape with that quantisation instead of pair by pair.
Tiling choose_tiling_uniform(bool packed, int lx, int ly, int64_t count, int n_simd)
{
    Tiling best{-1, 0};
    double best_cost = 0;
    const int slots = packed ? 2 : 1;
    for (int ci = 0; ci < kSwNumClasses; ++ci) {
        const int C = kSwClasses[ci];
        const int G = (lx + C - 1) / C;
        if (G > 64) continue;
        if (C > max_cols_per_lane() && best.cls >= 0) continue;
        if (force_cols_per_lane() && C != force_cols_per_lane()) continue;
        const int64_t per_wave = (int64_t)(64 / G) * slots;
        const int64_t waves = (count + per_wave - 1) / per_wave;
        const int64_t rounds = (waves + n_simd - 1) / n_simd;
        const double wgt = packed ? kSwPkClassCost[ci] : kSwClassCost[ci];
        if (wgt == 0) continue;
        const double c = (double)rounds * (ly + G - 1) * C * wgt;
        if (best.cls < 0 || c < best_cost) {
            best = Tiling{ci, G};
            best_cost = c;
        }
    }
    return best;
}

struct PairPlan {
    uint32_t pair;
    uint16_t lx;
    uint32_t ly;
    uint8_t cls;
    uint8_t G;
    uint8_t x_is_second; // 1 = sequence 2p+1 is the shorter one
};

struct ClassLaunch {
    int C = 0;
    uint32_t first_wave = 0, n_waves = 0;
};

} // namespace

struct agx_sw_batch {
    agx_ctx *ctx = nullptr;
    bool packed = false;
    bool biased = false; // packed batches: the biased formulation (agx_sw_pk2_kernel.hip)
    SwParams prm{};
    int64_t n_pairs = 0;
    DevBuf img, groups, waves, scores;
    DevBuf table; // substitution-matrix mode: kSwMatDim^2 int16 entries
    bool matrix = false;
    std::vector<ClassLaunch> launches;
    agx_sw_info info{};
};

namespace {
int create_batch(agx_ctx *ctx, const agx_sw_scoring *scoring, const agx_sw_matrix *matrix, const uint8_t *bases,
                 const uint64_t *off, const uint32_t *len, int64_t n_pairs, agx_sw_batch **out);
}

extern "C" {

void agx_sw_batch_destroy(agx_sw_batch *b)
{
    if (!b) return;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    b->img.release();
    b->groups.release();
    b->waves.release();
    b->scores.release();
    b->table.release();
    delete b;
}

int agx_sw_batch_create(agx_ctx *ctx, const uint8_t *bases, const uint64_t *off, const uint32_t *len, int64_t n_pairs,
                        agx_sw_batch **out)
{
    return agx_sw_batch_create_scored(ctx, nullptr, bases, off, len, n_pairs, out);
}

int agx_sw_batch_create_scored(agx_ctx *ctx, const agx_sw_scoring *scoring, const uint8_t *bases, const uint64_t *off,
                               const uint32_t *len, int64_t n_pairs, agx_sw_batch **out)
{
    return create_batch(ctx, scoring, nullptr, bases, off, len, n_pairs, out);
}

int agx_sw_batch_create_matrix(agx_ctx *ctx, const agx_sw_matrix *matrix, const uint8_t *bases, const uint64_t *off,
                               const uint32_t *len, int64_t n_pairs, agx_sw_batch **out)
{
    if (!matrix) {
        agx_set_error("agx_sw_batch_create_matrix: matrix is NULL");
        return AGX_E_ARG;
    }
    return create_batch(ctx, nullptr, matrix, bases, off, len, n_pairs, out);
}

} // extern "C"

namespace {

int create_batch(agx_ctx *ctx, const agx_sw_scoring *scoring, const agx_sw_matrix *matrix, const uint8_t *bases,
                 const uint64_t *off, const uint32_t *len, int64_t n_pairs, agx_sw_batch **out)
{
    if (!out) {
        agx_set_error("agx_sw_batch_create: out is NULL");
        return AGX_E_ARG;
    }
    *out = nullptr;
    // ctx == NULL: plan only (no device needed) -- the batch answers agx_sw_batch_info() and nothing else
    int rc = ctx ? agx_bind(ctx) : AGX_OK;
    if (rc) return rc;
    const int n_cu = ctx ? ctx->n_cu : 256;
    if (n_pairs < 0 || (n_pairs > 0 && (!off || !len))) {
        agx_set_error("agx_sw_batch_create: bad arguments (n_pairs=%lld)", (long long)n_pairs);
        return AGX_E_ARG;
    }
    if (n_pairs > 0x7fffffffLL / 2) {
        agx_set_error("agx_sw_batch_create: more than 2^30 pairs in one batch");
        return AGX_E_LIMIT;
    }

    // ---- scoring -> kernel constants
    const agx_sw_scoring ref_scoring = AGX_SW_SCORING_REFERENCE;
    agx_sw_scoring sc = scoring ? *scoring : ref_scoring;
    std::vector<int16_t> table; // matrix mode: [kSwMatDim][kSwMatDim], row/column 0 = padding
    if (matrix) {
        const int n = matrix->n_symbols;
        if (n < 1 || n > AGX_SW_MATRIX_MAX_SYMBOLS) {
            agx_set_error("substitution matrix: %d symbols, supported 1..%d", n, AGX_SW_MATRIX_MAX_SYMBOLS);
            return AGX_E_ARG;
        }
        int lo = 0;
        for (int a = 0; a < n; ++a)
            for (int c = 0; c < n; ++c) {
                if (matrix->score[a][c] != matrix->score[c][a]) {
                    agx_set_error("substitution matrix is not symmetric at (%d, %d)", a, c);
                    return AGX_E_ARG;
                }
                lo = std::min(lo, (int)matrix->score[a][c]);
            }
        for (int k = 0; k < 256; ++k)
            if (matrix->code[k] != 0xff && matrix->code[k] >= n) {
                agx_set_error("substitution matrix: code[%d] = %d is not a symbol number below %d", k, matrix->code[k], n);
                return AGX_E_ARG;
            }
        // the generic range check below sees a match/mismatch pair that always passes
        sc = agx_sw_scoring{1, 0, matrix->gap_open, matrix->gap_extend};
        const int gf = matrix->gap_open + matrix->gap_extend;
        // padding cells score the matrix minimum (<= 0): they cannot raise a local-alignment maximum
        table.assign((size_t)kSwMatDim * kSwMatDim, (int16_t)(lo - gf));
        for (int a = 0; a < n; ++a)
            for (int c = 0; c < n; ++c) table[(size_t)(a + 1) * kSwMatDim + (c + 1)] = (int16_t)(matrix->score[a][c] - gf);
    }
    // mismatch <= 0: padding relies on never-matching symbols not raising a score
    if (sc.match < 1 || sc.match > 12 || sc.mismatch > 0 || sc.mismatch < sc.match - 128 || sc.gap_open > 0 ||
        sc.gap_open < -1000 || sc.gap_extend > 0 || sc.gap_extend < -1000) {
        agx_set_error("scoring {match %d, mismatch %d, open %d, extend %d} outside the supported range", sc.match,
                      sc.mismatch, sc.gap_open, sc.gap_extend);
        return AGX_E_LIMIT;
    }
    SwParams prm{};
    prm.ge = sc.gap_extend;
    prm.gf = sc.gap_open + sc.gap_extend;
    prm.hd = sc.match - prm.gf;
    prm.delta = sc.match - sc.mismatch;
    prm.shift = 0;
    while ((1 << prm.shift) < prm.delta) ++prm.shift;
    auto twice = [](int v) { return (uint32_t)(uint16_t)(int16_t)v * 0x10001u; };
    prm.ge2 = twice(prm.ge);
    prm.gf2 = twice(prm.gf);
    prm.hd2 = twice(prm.hd);
    prm.delta2 = twice(prm.delta);
    // the packed int16 kernel covers shorter sides up to 64 x 40 columns; one longer pair moves the
    // whole batch to the int32 kernel, which also has the wide classes (up to 64 x 160)
    bool packed = use_packed_kernel() && !matrix; // the matrix lookup exists in the int32 kernel only
    uint32_t longest_short = 0;
    if (packed)
        for (int64_t p = 0; p < n_pairs; ++p) {
            const uint32_t sh = std::min(len[2 * p], len[2 * p + 1]);
            longest_short = std::max(longest_short, sh);
            if (sh > (uint32_t)kSwPackedMaxShort) {
                packed = false;
                break;
            }
        }
    // Biased formulation: every stored half must be the pattern of a positive normal half-precision number,
    // [0x0400, 0x7c00).  Smallest: B - max(|gf| + |ge|, delta); largest: B + (longest shorter side + 1) * match + |gf|.
    prm.age2 = twice(-prm.ge);
    prm.agf2 = twice(-prm.gf);
    const int bias = 0x0400 + std::max(-prm.gf - prm.ge, prm.delta);
    prm.bias2 = twice(bias);
    const bool biased = packed && kernel_choice() == 2 &&
                        (int64_t)bias + ((int64_t)longest_short + 1) * sc.match - prm.gf < 0x7c00;
    const uint32_t max_short = matrix ? (uint32_t)kSwPackedMaxShort : AGX_SW_MAX_SHORT_LEN; // no wide classes in matrix mode

    const bool trace = getenv("AGX_TRACE_CREATE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    if (n_pairs > 0 && !bases) {
        for (int64_t p = 0; p < 2 * n_pairs; ++p)
            if (len[p]) {
                agx_set_error("agx_sw_batch_create: bases is NULL");
                return AGX_E_ARG;
            }
    }

    // ---- plan every pair (threads over pairs): validate, orient, choose the lane tiling
    const double beta0 = tail_beta_override() >= 0 ? tail_beta_override() : 0.0;
    std::vector<PairPlan> all((size_t)n_pairs);
    struct Worker {
        int rc = AGX_OK;
        int64_t bad_pair = -1;
        int64_t cells = 0;
        double waves = 0;
        double class_work[sizeof(kSwClasses) / sizeof(kSwClasses[0])] = {};
    };
    std::vector<Worker> wk((size_t)agx_host_threads());
    agx_parallel_for(n_pairs, 4096, [&](int64_t lo, int64_t hi, int tid) {
        Worker &me = wk[(size_t)tid];
        for (int64_t p = lo; p < hi; ++p) {
            PairPlan &pp = all[(size_t)p];
            pp = PairPlan{};
            pp.pair = (uint32_t)p;
            pp.cls = 255; // 255 = nothing to fill
            const uint32_t la = len[2 * p], lb = len[2 * p + 1];
            me.cells += (int64_t)la * lb;
            if (la == 0 || lb == 0) continue; // no interior cell: score stays 0
            const bool second_short = lb < la; // ties keep file order (antidiagonalSmithWaterman.c:229-244)
            const uint32_t lx = second_short ? lb : la, ly = second_short ? la : lb;
            int rc = AGX_OK;
            Tiling tl{-1, 0};
            bool bad_symbol;
            if (matrix) { // any byte outside the alphabet maps to 0xff
                const uint8_t *q = bases + off[2 * p], *r = bases + off[2 * p + 1];
                uint8_t bad = 0;
                for (uint32_t k = 0; k < la; ++k) bad |= (uint8_t)(matrix->code[q[k]] == 0xff);
                for (uint32_t k = 0; k < lb; ++k) bad |= (uint8_t)(matrix->code[r[k]] == 0xff);
                bad_symbol = bad != 0;
            } else
                bad_symbol = memchr(bases + off[2 * p], 0, la) || memchr(bases + off[2 * p + 1], 0, lb);
            if (lx > max_short || ly > 0xffffu) rc = AGX_E_LIMIT;
            else if (bad_symbol) rc = AGX_E_SYMBOL;
            else {
                double cost = 0;
                tl = choose_tiling(packed, (int)lx, (int)ly, ~0u, &cost, beta0);
                if (tl.cls < 0 || (matrix && kSwClasses[tl.cls] > 40)) rc = AGX_E_LIMIT; // no wide classes in matrix mode
                else {
                    me.class_work[tl.cls] += cost;
                    me.waves += (double)tl.G / 64.0 / (packed ? 2 : 1);
                }
            }
            if (rc != AGX_OK) {
                if (me.rc == AGX_OK) {
                    me.rc = rc;
                    me.bad_pair = p;
                }
                continue;
            }
            pp.lx = (uint16_t)lx;
            pp.ly = ly;
            pp.cls = (uint8_t)tl.cls;
            pp.G = (uint8_t)tl.G;
            pp.x_is_second = second_short ? 1 : 0;
        }
    });
    int64_t cells = 0;
    for (const Worker &w : wk) cells += w.cells;
    for (const Worker &w : wk)
        if (w.rc != AGX_OK) {
            const int64_t p = w.bad_pair;
            if (w.rc == AGX_E_SYMBOL && matrix)
                agx_set_error("pair %lld contains a byte outside the substitution matrix's alphabet", (long long)p);
            else if (w.rc == AGX_E_SYMBOL)
                agx_set_error("pair %lld contains byte 0x00, which is reserved as the padding symbol", (long long)p);
            else
                agx_set_error("pair %lld: lengths %u x %u exceed the supported %u x 65535 (shorter x longer)", (long long)p,
                              len[2 * p], len[2 * p + 1], max_short);
            return w.rc;
        }
    // Tail regime: when the planned waves fill the chip's resident capacity (about 5 per SIMD for this
    // kernel) less than 1.6 times, a launch lasts as long as its longest waves -- alone on their SIMDs
    // in a small batch, or stranded in a mostly empty second filling.  Such a batch is re-tiled with a
    // term on a wave's own duration (3 lanes' worth): long pairs spread over more lanes, waves get
    // shorter and more numerous.  Mixed 32..512 batches (tools/sw_tail_beta_sweep.py,
    // sw_tail_rule_check.py): 8192 pairs 1.24 -> 2.6 TCUPS, 16 384 2.46 -> 2.9, 131 072 4.15 -> 4.61;
    // beyond 1.6 fillings the term costs 1-3 % and is left out.  (Uniform batches are re-tiled below
    // with their own wave-count model.)
    double beta_used = beta0;
    if (tail_beta_override() < 0 && n_cu > 0) {
        double waves_est = 0;
        for (const Worker &w : wk) waves_est += w.waves;
        const double fill = waves_est / (5.0 * 4.0 * n_cu);
        if (fill < 1.6) {
            // the emptier the chip, the more a wave's own duration counts (sweep of 2048 ... 131 072 pairs)
            beta_used = fill < 0.1 ? 10.0 : fill < 0.4 ? 6.0 : 3.0;
            for (Worker &w : wk) {
                w.waves = 0;
                for (double &c : w.class_work) c = 0;
            }
            agx_parallel_for(n_pairs, 4096, [&](int64_t lo, int64_t hi, int tid) {
                Worker &me = wk[(size_t)tid];
                for (int64_t p = lo; p < hi; ++p) {
                    PairPlan &pp = all[(size_t)p];
                    if (pp.cls == 255) continue;
                    double cost = 0;
                    const Tiling tl = choose_tiling(packed, (int)pp.lx, (int)pp.ly, ~0u, &cost, beta_used);
                    if (tl.cls >= 0 && !(matrix && kSwClasses[tl.cls] > 40)) {
                        pp.cls = (uint8_t)tl.cls;
                        pp.G = (uint8_t)tl.G;
                    }
                    me.class_work[pp.cls] += cost;
                    me.waves += (double)pp.G / 64.0 / (packed ? 2 : 1);
                }
            });
        }
    }
    // Every class is its own launch and the measured cost curve is flat over many widths: a mixed batch
    // keeps the classes that carry most of the work -- about one per 4096 wavefronts, at most 6
    // (tools/sw_mixed_sweep.py: 16384 pairs of 32..512 went from 0.63 to 2.5 TCUPS, 65536 from 2.1 to 3.9) -- and
    // re-tiles the other pairs among them (a pair no kept class can span keeps its own).
    {
        double work[sizeof(kSwClasses) / sizeof(kSwClasses[0])] = {};
        double waves_est = 0;
        for (const Worker &w : wk) {
            waves_est += w.waves;
            for (int c = 0; c < kSwNumClasses; ++c) work[c] += w.class_work[c];
        }
        static const double per_class = [] {
            const char *e = getenv("AGX_SW_WAVES_PER_CLASS");
            return e && atof(e) > 0 ? atof(e) : 4096.0;
        }();
        const int k_max = std::min(max_classes(), 1 + (int)(waves_est / per_class));
        int used = 0;
        for (int c = 0; c < kSwNumClasses; ++c) used += work[c] > 0;
        if (used > k_max) {
            int order[sizeof(kSwClasses) / sizeof(kSwClasses[0])];
            for (int c = 0; c < kSwNumClasses; ++c) order[c] = c;
            std::sort(order, order + kSwNumClasses, [&](int x, int y) { return work[x] > work[y]; });
            uint32_t keep = 0;
            for (int k = 0; k < k_max; ++k) keep |= 1u << order[k];
            agx_parallel_for(n_pairs, 4096, [&](int64_t lo, int64_t hi, int) {
                for (int64_t p = lo; p < hi; ++p) {
                    PairPlan &pp = all[(size_t)p];
                    if (pp.cls == 255 || ((keep >> pp.cls) & 1u)) continue;
                    const Tiling tl = choose_tiling(packed, (int)pp.lx, (int)pp.ly, keep, nullptr, beta_used);
                    if (tl.cls >= 0) {
                        pp.cls = (uint8_t)tl.cls;
                        pp.G = (uint8_t)tl.G;
                    }
                }
            });
        }
    }
    // dominant shape?  (sampled first, counted only if the sample says so)
    if (n_pairs >= 1024 && n_cu > 0) {
        const size_t stride = (size_t)n_pairs / 512;
        uint32_t cand = 0;
        int votes = 0;
        for (size_t k = 0; k < 512; ++k) { // Boyer-Moore majority vote over a sample
            const PairPlan &pp = all[k * stride];
            const uint32_t key = (uint32_t)pp.lx << 16 | (pp.ly & 0xffffu);
            if (pp.cls == 255) continue;
            if (votes == 0) {
                cand = key;
                votes = 1;
            } else
                votes += key == cand ? 1 : -1;
        }
        int64_t count = 0;
        if (votes > 0)
            for (const PairPlan &pp : all)
                if (pp.cls != 255 && ((uint32_t)pp.lx << 16 | (pp.ly & 0xffffu)) == cand) ++count;
        if (count * 2 >= n_pairs) {
            const Tiling tl = choose_tiling_uniform(packed, (int)(cand >> 16), (int)(cand & 0xffffu), count, 4 * n_cu);
            if (tl.cls >= 0)
                for (PairPlan &pp : all)
                    if (pp.cls != 255 && ((uint32_t)pp.lx << 16 | (pp.ly & 0xffffu)) == cand) {
                        pp.cls = (uint8_t)tl.cls;
                        pp.G = (uint8_t)tl.G;
                    }
        }
    }
    const double t_plan = now();

    // ---- order: class, then lanes per group (wide first), then long rows first, then file order;
    // waves end up homogeneous and the longest waves of a launch are dispatched first.
    // Two stable counting passes (LSD): by ly descending, then by (class, G descending).
    std::vector<PairPlan> plan;
    {
        uint32_t max_ly = 0;
        size_t n_fill = 0;
        for (const PairPlan &pp : all)
            if (pp.cls != 255) {
                max_ly = std::max(max_ly, pp.ly);
                ++n_fill;
            }
        std::vector<uint32_t> cnt((size_t)max_ly + 2, 0);
        for (const PairPlan &pp : all)
            if (pp.cls != 255) ++cnt[(size_t)(max_ly - pp.ly) + 1];
        for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
        std::vector<PairPlan> tmp(n_fill);
        for (const PairPlan &pp : all)
            if (pp.cls != 255) tmp[cnt[(size_t)(max_ly - pp.ly)]++] = pp;
        std::vector<uint32_t> cnt2((size_t)kSwNumClasses * 64 + 1, 0);
        auto bucket = [](const PairPlan &pp) { return (size_t)pp.cls * 64 + (size_t)(64 - pp.G); };
        for (const PairPlan &pp : tmp) ++cnt2[bucket(pp) + 1];
        for (size_t k = 1; k < cnt2.size(); ++k) cnt2[k] += cnt2[k - 1];
        plan.resize(n_fill);
        for (const PairPlan &pp : tmp) plan[cnt2[bucket(pp)]++] = pp;
    }
    std::vector<PairPlan>().swap(all);
    const double t_sort = now();

    // ---- form waves and lay out the image (offsets only), then copy the bytes with threads.
    // Packed kernel: a group carries up to two pairs (slots); the int32 kernel one.
    const int slots = packed ? 2 : 1;
    struct Slot {
        int32_t plan[2]; // indices into plan[], -1 = empty second slot
    };
    std::vector<Slot> gslots;
    gslots.reserve(plan.size() / slots + 16);
    std::vector<SwWave> waves;
    std::vector<ClassLaunch> launches;
    int64_t padded = 0;
    // word 0.. of the image: a zero block any empty slot points at (x of up to 64*40 bytes)
    size_t img_dw = packed ? (size_t)kSwPackedMaxShort / 4 + 1 : 0;
    std::vector<uint32_t> x_dw(plan.size()), y_dw(plan.size());
    size_t i = 0;
    while (i < plan.size()) {
        const int cls = plan[i].cls;
        ClassLaunch cl;
        cl.C = kSwClasses[cls];
        cl.first_wave = (uint32_t)waves.size();
        while (i < plan.size() && plan[i].cls == cls) {
            const int G = plan[i].G;
            const int per_wave = 64 / G;
            SwWave w{};
            w.first_group = (uint32_t)gslots.size();
            w.G = (uint16_t)G;
            int n = 0, max_ly = 0;
            while (i < plan.size() && plan[i].cls == cls && plan[i].G == G && n < per_wave) {
                Slot sl{{-1, -1}};
                for (int k = 0; k < slots && i < plan.size() && plan[i].cls == cls && plan[i].G == G; ++k, ++i) {
                    const PairPlan &pp = plan[i];
                    const size_t xdw = ((size_t)G * cl.C + 3) / 4 + 1, ydw = ((size_t)pp.ly + 3) / 4;
                    if (img_dw + xdw + ydw > 0xffffffffull) {
                        agx_set_error("packed image exceeds 16 GiB; split the batch");
                        return AGX_E_LIMIT;
                    }
                    x_dw[i] = (uint32_t)img_dw;
                    img_dw += xdw;
                    y_dw[i] = (uint32_t)img_dw;
                    img_dw += ydw;
                    sl.plan[k] = (int32_t)i;
                    max_ly = std::max(max_ly, (int)pp.ly);
                }
                gslots.push_back(sl);
                ++n;
            }
            w.n_groups = (uint16_t)n;
            w.steps = (uint32_t)(max_ly + G - 1);
            padded += (int64_t)w.steps * 64 * cl.C * slots;
            waves.push_back(w);
        }
        cl.n_waves = (uint32_t)waves.size() - cl.first_wave;
        // dispatch order = longest waves first (a wave lasts steps x C; C is the class's): the buckets were
        // filled widest group first, which leaves narrow groups with long rows for the end of the launch
        static const bool sort_waves = [] {
            const char *e = getenv("AGX_SW_SORT_WAVES"); // experiment knob: 0 keeps the bucket order
            return !(e && e[0] == '0');
        }();
        if (sort_waves)
            std::stable_sort(waves.begin() + cl.first_wave, waves.end(),
                             [](const SwWave &a, const SwWave &b) { return a.steps > b.steps; });
        launches.push_back(cl);
    }
    // group records
    std::vector<SwGroup> groups1(packed ? 0 : gslots.size());
    std::vector<SwGroup2> groups2(packed ? gslots.size() : 0);
    for (size_t k = 0; k < gslots.size(); ++k) {
        for (int h = 0; h < slots; ++h) {
            const int32_t pi = gslots[k].plan[h];
            uint32_t xd = 0, yd = 0, ll = 0, out = (uint32_t)n_pairs; // empty slot: zero block, spare score
            if (pi >= 0) {
                const PairPlan &pp = plan[(size_t)pi];
                xd = x_dw[(size_t)pi];
                yd = y_dw[(size_t)pi];
                ll = (uint32_t)pp.lx | (pp.ly << 16);
                out = pp.pair;
            }
            if (packed) {
                groups2[k].x_dw[h] = xd;
                groups2[k].y_dw[h] = yd;
                groups2[k].lx_ly[h] = ll;
                groups2[k].out[h] = out;
            } else {
                groups1[k] = SwGroup{xd, yd, ll, out};
            }
        }
    }
    const void *groups_data = packed ? (const void *)groups2.data() : (const void *)groups1.data();
    const size_t groups_bytes = packed ? groups2.size() * sizeof(SwGroup2) : groups1.size() * sizeof(SwGroup);

    struct ImgBuf { // uninitialised storage: every byte is written below (data or zero padding)
        uint32_t *p = nullptr;
        size_t n = 0;
        ~ImgBuf() { free(p); }
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
        const uint32_t *data() const { return p; }
    } img;
    img.n = img_dw;
    img.p = (uint32_t *)malloc(std::max<size_t>(img_dw, 4) * 4);
    if (!img.p) {
        agx_set_error("agx_sw_batch_create: out of host memory for the packed image");
        return AGX_E_NOMEM;
    }
    if (packed) memset(img.p, 0, (size_t)kSwPackedMaxShort + 4);
    agx_parallel_for((int64_t)plan.size(), 2048, [&](int64_t lo, int64_t hi, int) {
        for (int64_t k = lo; k < hi; ++k) {
            const PairPlan &pp = plan[(size_t)k];
            const uint64_t ox = off[2 * (uint64_t)pp.pair + pp.x_is_second];
            const uint64_t oy = off[2 * (uint64_t)pp.pair + (pp.x_is_second ^ 1)];
            uint8_t *x = (uint8_t *)(img.p + x_dw[(size_t)k]), *y = (uint8_t *)(img.p + y_dw[(size_t)k]);
            const size_t xb = (size_t)(y_dw[(size_t)k] - x_dw[(size_t)k]) * 4, yb = (((size_t)pp.ly + 3) / 4) * 4;
            if (matrix) { // symbol numbers 1..n; 0 stays the padding symbol
                for (uint32_t c = 0; c < pp.lx; ++c) x[c] = (uint8_t)(matrix->code[bases[ox + c]] + 1);
                for (uint32_t c = 0; c < pp.ly; ++c) y[c] = (uint8_t)(matrix->code[bases[oy + c]] + 1);
            } else {
                memcpy(x, bases + ox, pp.lx);
                memcpy(y, bases + oy, pp.ly);
            }
            memset(x + pp.lx, 0, xb - pp.lx);
            memset(y + pp.ly, 0, yb - pp.ly);
        }
    });

    const double t_pack = now();
    // ---- device image
    agx_sw_batch *b = new agx_sw_batch();
    b->ctx = ctx;
    b->n_pairs = n_pairs;
    b->launches = launches;
    b->info.n_pairs = n_pairs;
    b->info.cells = cells;
    b->info.padded_cells = padded;
    b->packed = packed;
    b->biased = biased;
    b->matrix = matrix != nullptr;
    b->prm = prm;
    b->info.input_bytes = (int64_t)(img.size() * 4 + groups_bytes + waves.size() * sizeof(SwWave));
    b->info.n_launches = (int32_t)launches.size();
    b->info.n_waves = (int32_t)waves.size();
    if (!ctx) { // planning only
        *out = b;
        return AGX_OK;
    }
    rc = b->img.alloc(img.size() * 4);
    if (!rc) rc = b->groups.alloc(groups_bytes);
    if (!rc) rc = b->waves.alloc(waves.size() * sizeof(SwWave));
    if (!rc && matrix) rc = b->table.alloc(table.size() * sizeof(int16_t));
    if (!rc) rc = b->scores.alloc(((size_t)n_pairs + 1) * sizeof(int32_t)); // +1: spare slot of empty packed halves
    if (rc) {
        agx_sw_batch_destroy(b);
        return rc;
    }
    hipError_t e = hipSuccess;
    if (!img.empty()) e = hipMemcpy(b->img.p, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && groups_bytes)
        e = hipMemcpy(b->groups.p, groups_data, groups_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && !waves.empty())
        e = hipMemcpy(b->waves.p, waves.data(), waves.size() * sizeof(SwWave), hipMemcpyHostToDevice);
    if (e == hipSuccess && matrix) e = hipMemcpy(b->table.p, table.data(), table.size() * sizeof(int16_t), hipMemcpyHostToDevice);
    // pairs with an empty side are never touched by a kernel: their score is this zero
    if (e == hipSuccess) e = hipMemset(b->scores.p, 0, b->scores.bytes);
    if (e != hipSuccess) {
        agx_set_error("agx_sw_batch_create: upload -> %s", hipGetErrorString(e));
        agx_sw_batch_destroy(b);
        return AGX_E_HIP;
    }
    if (trace)
        fprintf(stderr, "[agx_sw_batch_create] %lld pairs: plan %.2f ms, sort %.2f ms, pack %.2f ms, alloc+H2D %.2f ms (%.1f MB)\n",
                (long long)n_pairs, t_plan - t_begin, t_sort - t_plan, t_pack - t_sort, now() - t_pack, img.size() * 4 / 1e6);
    *out = b;
    return AGX_OK;
}

} // namespace

extern "C" {

int agx_sw_batch_launch(agx_sw_batch *b)
{
    if (!b) {
        agx_set_error("agx_sw_batch_launch: null batch");
        return AGX_E_ARG;
    }
    if (!b->ctx) {
        agx_set_error("this batch was planned without a context (no device): it cannot be launched");
        return AGX_E_NODEVICE;
    }
    int rc = agx_bind(b->ctx);
    if (rc) return rc;
    FanOut fan(b->ctx, (int)b->launches.size());
    rc = fan.begin();
    if (rc) return rc;
    int k = 0;
    // widest class first: its waves have the longest rows-times-columns chain, so they should not be the tail
    for (auto it = b->launches.rbegin(); it != b->launches.rend(); ++it) {
        const ClassLaunch &cl = *it;
        hipStream_t st = fan.stream(k++);
        const int r = b->matrix
                          ? agx_sw_mat_launch_class(cl.C, b->prm, (const uint32_t *)b->img.p, (const SwGroup *)b->groups.p,
                                                    (const SwWave *)b->waves.p + cl.first_wave, cl.n_waves,
                                                    (int32_t *)b->scores.p, (const int16_t *)b->table.p, st)
                          : b->packed
                          ? (b->biased ? agx_sw_pk2_launch_class : agx_sw_pk_launch_class)(cl.C, b->prm, (const uint32_t *)b->img.p, (const SwGroup2 *)b->groups.p,
                                                   (const SwWave *)b->waves.p + cl.first_wave, cl.n_waves,
                                                   (int32_t *)b->scores.p, st)
                          : (cl.C > 40 ? agx_sw_wide_launch_class : agx_sw_launch_class)(
                                cl.C, b->prm, (const uint32_t *)b->img.p, (const SwGroup *)b->groups.p,
                                (const SwWave *)b->waves.p + cl.first_wave, cl.n_waves, (int32_t *)b->scores.p, st);
        if (r) {
            agx_set_error("sw_fill<%d> launch failed: %s", cl.C, hipGetErrorString(hipGetLastError()));
            return AGX_E_HIP;
        }
    }
    return fan.end();
}

int agx_sw_batch_scores(agx_sw_batch *b, int32_t *scores)
{
    if (!b || (!scores && b->n_pairs)) {
        agx_set_error("agx_sw_batch_scores: null argument");
        return AGX_E_ARG;
    }
    if (!b->ctx) {
        agx_set_error("this batch was planned without a context (no device): it has no scores");
        return AGX_E_NODEVICE;
    }
    int rc = agx_bind(b->ctx);
    if (rc) return rc;
    AGX_HIP(hipStreamSynchronize(b->ctx->stream));
    if (b->n_pairs)
        AGX_HIP(hipMemcpy(scores, b->scores.p, (size_t)b->n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost));
    return AGX_OK;
}

int agx_sw_batch_info(const agx_sw_batch *b, agx_sw_info *info)
{
    if (!b || !info) {
        agx_set_error("agx_sw_batch_info: null argument");
        return AGX_E_ARG;
    }
    *info = b->info;
    return AGX_OK;
}

int agx_sw_score(agx_ctx *ctx, const uint8_t *bases, const uint64_t *off, const uint32_t *len, int64_t n_pairs,
                 int32_t *scores)
{
    agx_sw_batch *b = nullptr;
    int rc = agx_sw_batch_create(ctx, bases, off, len, n_pairs, &b);
    if (rc) return rc;
    const bool trace = getenv("AGX_TRACE_CREATE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    rc = agx_sw_batch_launch(b);
    const double t1 = now();
    if (!rc) rc = agx_sw_batch_scores(b, scores);
    const double t2 = now();
    agx_sw_batch_destroy(b);
    if (trace) fprintf(stderr, "[agx_sw_score] launch %.2f ms, wait+D2H %.2f ms, destroy %.2f ms\n", t1 - t0, t2 - t1, now() - t2);
    return rc;
}

int agx_sw_score_multi(int n_devices, const uint8_t *bases, const uint64_t *off, const uint32_t *len, int64_t n_pairs,
                       int32_t *scores)
{
    const int avail = agx_device_count();
    if (avail <= 0) {
        agx_set_error("no HIP device is visible (this library has no CPU fallback)");
        return AGX_E_NODEVICE;
    }
    // AGX_MULTI_OVERSUBSCRIBE=1 (tests on a one-GPU box): keep the requested shard count, shard k runs on device k % avail
    const bool oversub = getenv("AGX_MULTI_OVERSUBSCRIBE") != nullptr && n_devices > 0 && n_devices <= 64;
    if (n_devices <= 0 || (n_devices > avail && !oversub)) n_devices = avail;
    if (n_pairs < 0 || (n_pairs > 0 && (!off || !len || !scores))) {
        agx_set_error("agx_sw_score_multi: bad arguments");
        return AGX_E_ARG;
    }
    // contiguous shards balanced by cells; results land in disjoint slices
    std::vector<int64_t> cut(n_devices + 1, n_pairs);
    {
        double total = 0;
        for (int64_t p = 0; p < n_pairs; ++p) total += (double)len[2 * p] * len[2 * p + 1] + 1.0;
        double acc = 0;
        int d = 1;
        cut[0] = 0;
        for (int64_t p = 0; p < n_pairs && d < n_devices; ++p) {
            acc += (double)len[2 * p] * len[2 * p + 1] + 1.0;
            while (d < n_devices && acc >= total * d / n_devices) cut[d++] = p + 1;
        }
    }
    std::vector<int> rcs(n_devices, AGX_OK);
    std::vector<std::string> errs(n_devices);
    std::vector<std::thread> th;
    for (int d = 0; d < n_devices; ++d) {
        th.emplace_back([&, d]() {
            const int64_t lo = cut[d], hi = cut[d + 1];
            if (hi <= lo) return;
            agx_ctx *c = nullptr;
            int rc = agx_ctx_create(d % avail, &c);
            if (!rc) rc = agx_sw_score(c, bases, off + 2 * lo, len + 2 * lo, hi - lo, scores + lo);
            if (rc) errs[d] = agx_last_error();
            agx_ctx_destroy(c);
            rcs[d] = rc;
        });
    }
    for (auto &t : th) t.join();
    for (int d = 0; d < n_devices; ++d)
        if (rcs[d]) {
            agx_set_error("device %d: %s", d, errs[d].c_str());
            return rcs[d];
        }
    return AGX_OK;
}

} // extern "C"
