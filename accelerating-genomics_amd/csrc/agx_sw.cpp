// Host side of the Smith-Waterman path: validation, lane-tiling choice, wave formation, uploads,
// launches, multi-device sharding (include/agx.h, "Smith-Waterman" section).
//
// What the host does per batch (agx_sw_batch_create), every step threaded over the process's pool:
//   A  one pass over len[]: orientation (shorter sequence across the lanes), limits, cell count
//   B  lane tiling per pair from a per-length lookup table (the lower envelope of the class cost lines),
//      then the batch-level rules: tail regime, class consolidation, dominant shape
//   C  two stable counting passes: long rows first, then (class, lanes per group)
//   D  waves, group records and image offsets in closed form per (class, G) bucket
// while the caller's `bases` and `off` arrays travel to the device as they are; a device kernel
// (agx_sw_pack_kernel.hip) then builds the padded image and checks the symbols.  No byte of a sequence
// is touched by the host.
#include "agx_sw.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "agx_internal.h"
#include "agx_parallel.h"

namespace {

struct Tiling {
    int cls; // index into kSwClasses
    int G;
};

// Tuning build only (see agx_tune): AGX_SW_TAIL_BETA, AGX_SW_MAX_C, AGX_SW_FORCE_C, AGX_SW_MAX_CLASSES,
// AGX_SW_WAVES_PER_CLASS, AGX_SW_SORT_WAVES, AGX_SW_KERNEL.  The shipped library runs on the defaults.
inline double tail_beta_override()
{
    static const double v = [] {
        const char *e = agx_tune("AGX_SW_TAIL_BETA");
        return e ? atof(e) : -1.0;
    }();
    return v;
}
int max_cols_per_lane()
{
    static const int v = [] {
        const char *e = agx_tune("AGX_SW_MAX_C");
        const int n = e ? atoi(e) : 0;
        return n >= 4 ? n : AGX_SW_MAX_COLS_PER_LANE;
    }();
    return v;
}
int force_cols_per_lane()
{
    static const int v = [] {
        const char *e = agx_tune("AGX_SW_FORCE_C");
        return e ? atoi(e) : 0;
    }();
    return v;
}
int max_classes_override() // 0 = none
{
    static const int v = [] {
        const char *e = agx_tune("AGX_SW_MAX_CLASSES");
        const int n = e ? atoi(e) : 0;
        return n > 0 ? n : 0;
    }();
    return v;
}
int tuned_kernel() // 0 = no override
{
    static const int v = [] {
        const char *e = agx_tune("AGX_SW_KERNEL");
        if (e && strcmp(e, "i32") == 0) return AGX_SW_KERNEL_INT32;
        if (e && strcmp(e, "pk1") == 0) return AGX_SW_KERNEL_PACKED_SIGNED;
        if (e && strcmp(e, "pk2") == 0) return AGX_SW_KERNEL_PACKED_BIASED;
        return 0;
    }();
    return v;
}

// per-class cost table of the kernel family a batch runs (relative lane time per padded cell)
inline const double *class_costs(int family) // 0 int32, 1 packed signed, 2 packed biased, 3 int32 on the packed plan's coded image
{
    return family == 0 ? kSwClassCost : family == 1 ? kSwPkClassCost : family == 3 ? kSwI32dClassCost : kSwPk2ClassCost;
}

// Lane time a pair costs under tiling (class ci, G): steps * C * 64 / floor(64 / G) padded cells (the
// lanes of a wave that cannot host another group are charged to the pair), weighted by the measured
// per-cell cost of the class.  beta: lanes' worth of extra weight on a wave's own duration (steps * C),
// which favours spreading long pairs over more lanes; 0 in the throughput regime.
inline double tiling_slope(const double *costs, int ci, int G, double beta)
{
    return kSwClasses[ci] * ((64.0 / (double)(64 / G)) * costs[ci] + beta);
}

// The choice for one shorter length lx as a function of the longer length ly: cost_i(ly) = (ly + G_i - 1) * w_i
// is a line per class, the best class is their lower envelope -- a handful of segments.
struct TilingSeg {
    uint32_t ly_from; // the segment covers [ly_from, next segment's ly_from)
    uint8_t cls, G;
    double w;
};
struct TilingTable {
    std::vector<uint32_t> first; // index of lx's first segment; first[lx + 1] ends it (no segment = no class spans lx)
    std::vector<TilingSeg> segs;
    inline bool pick(uint32_t lx, uint32_t ly, Tiling *t, double *cost) const
    {
        const uint32_t a = first[lx], b = first[lx + 1];
        if (a == b) return false;
        uint32_t k = a;
        while (k + 1 < b && segs[k + 1].ly_from <= ly) ++k;
        const TilingSeg &s = segs[k];
        *t = Tiling{s.cls, s.G};
        if (cost) *cost = (double)(ly + s.G - 1) * s.w;
        return true;
    }
};

// present[lx] != 0: some pair of the batch has that shorter length (only those rows are built)
void build_tiling_table(TilingTable &tt, const std::vector<uint8_t> &present, const double *costs, uint32_t allowed, double beta)
{
    const uint32_t n_lx = (uint32_t)present.size();
    std::vector<std::vector<TilingSeg>> rows(n_lx);
    agx_parallel_for((int64_t)n_lx, 64, [&](int64_t lo, int64_t hi, int) {
        for (int64_t lx = std::max<int64_t>(lo, 1); lx < hi; ++lx) {
            if (!present[(size_t)lx]) continue;
            struct Line {
                int ci, G;
                double w;
            } ln[kSwNumClasses];
            int n = 0;
            for (int ci = 0; ci < kSwNumClasses; ++ci) {
                if (!((allowed >> ci) & 1u)) continue;
                const int C = kSwClasses[ci];
                const int G = ((int)lx + C - 1) / C;
                if (G > 64) continue;
                if (C > max_cols_per_lane() && n > 0) continue;
                if (force_cols_per_lane() && C != force_cols_per_lane()) continue;
                if (costs[ci] == 0) continue; // class not built for this kernel
                ln[n++] = Line{ci, G, tiling_slope(costs, ci, G, beta)};
            }
            if (n == 0) continue;
            auto cost_at = [&](int k, double ly) { return (ly + ln[k].G - 1) * ln[k].w; };
            // start at ly = lx (the longer side is never shorter); ties go to the wider class
            int cur = 0;
            for (int k = 1; k < n; ++k)
                if (cost_at(k, (double)lx) <= cost_at(cur, (double)lx)) cur = k;
            uint32_t from = (uint32_t)lx;
            std::vector<TilingSeg> &row = rows[(size_t)lx];
            for (;;) {
                row.push_back(TilingSeg{from, (uint8_t)ln[cur].ci, (uint8_t)ln[cur].G, ln[cur].w});
                // the next line to undercut the current one: smaller slope, first ly where it is strictly cheaper
                int nxt = -1;
                double nxt_at = 0;
                for (int k = 0; k < n; ++k) {
                    if (!(ln[k].w < ln[cur].w)) continue;
                    const double x = ((ln[k].G - 1) * ln[k].w - (ln[cur].G - 1) * ln[cur].w) / (ln[cur].w - ln[k].w);
                    double at = std::floor(x) + 1;
                    if (at <= (double)from) at = (double)from + 1;
                    if (nxt < 0 || at < nxt_at || (at == nxt_at && ln[k].w < ln[nxt].w)) {
                        nxt = k;
                        nxt_at = at;
                    }
                }
                if (nxt < 0 || nxt_at > 65535.0) break;
                cur = nxt;
                from = (uint32_t)nxt_at;
            }
        }
    });
    tt.first.assign((size_t)n_lx + 1, 0);
    tt.segs.clear();
    for (uint32_t lx = 0; lx < n_lx; ++lx) {
        tt.first[lx] = (uint32_t)tt.segs.size();
        tt.segs.insert(tt.segs.end(), rows[lx].begin(), rows[lx].end());
    }
    tt.first[n_lx] = (uint32_t)tt.segs.size();
}

// Uniform batches (most pairs share one shape, e.g. fixed-length reads): every wave of that shape
// costs the same, so the launch lasts ceil(waves / SIMDs) wave-times -- the tiling is chosen for the
// whole shape with that quantisation instead of pair by pair.
Tiling choose_tiling_uniform(const double *costs, int slots, int lx, int ly, int64_t count, int n_simd)
{
    Tiling best{-1, 0};
    double best_cost = 0;
    for (int ci = 0; ci < kSwNumClasses; ++ci) {
        const int C = kSwClasses[ci];
        const int G = (lx + C - 1) / C;
        if (G > 64) continue;
        if (C > max_cols_per_lane() && best.cls >= 0) continue;
        if (force_cols_per_lane() && C != force_cols_per_lane()) continue;
        const int64_t per_wave = (int64_t)(64 / G) * slots;
        const int64_t waves = (count + per_wave - 1) / per_wave;
        const int64_t rounds = (waves + n_simd - 1) / n_simd;
        const double wgt = costs[ci];
        if (wgt == 0) continue;
        const double c = (double)rounds * (ly + G - 1) * C * wgt;
        if (best.cls < 0 || c < best_cost) {
            best = Tiling{ci, G};
            best_cost = c;
        }
    }
    return best;
}

// agx_sw_score sends a large batch through in pieces (upload of piece k + 1 beside the fill of piece k); tuning build:
// AGX_SW_PIECE_MB, AGX_SW_PIECE_MIN_PAIRS
inline uint64_t piece_bytes()
{
    static const uint64_t v = [] {
        const char *e = agx_tune("AGX_SW_PIECE_MB");
        return (uint64_t)(e && atoi(e) > 0 ? atoi(e) : 32) << 20;
    }();
    return v;
}
inline int64_t piece_min_pairs()
{
    static const int64_t v = [] {
        const char *e = agx_tune("AGX_SW_PIECE_MIN_PAIRS");
        return (int64_t)(e && atoi(e) > 0 ? atoi(e) : 32768);
    }();
    return v;
}
constexpr size_t kRawPad = 64;       // bytes in front of and behind the uploaded sequences (16-byte aligned pieces, agx_sw_pack_kernel.hip)
constexpr uint8_t kClsEmpty = 255;   // an empty side: nothing to fill
constexpr uint8_t kClsUntiled = 254; // pass A done, no tiling yet
struct PairPlan {                    // 12 bytes: the sorts move these
    uint32_t pair;
    uint32_t ly;
    uint16_t lxo; // lx | (1 = sequence 2p+1 is the shorter one) << 15, as the group records carry it
    uint8_t cls;
    uint8_t G;
    uint32_t lx() const { return lxo & 0x7fffu; }
};

struct ClassLaunch {
    int C = 0;
    uint32_t first_wave = 0, n_waves = 0;
};

// one (class, G) run of the sorted plan: entry j of the run is slot j % slots of its group j / slots,
// a wave takes 64 / G groups
struct Bucket {
    size_t first = 0, count = 0;     // entries of plan[]
    size_t group0 = 0, n_groups = 0; // group records
    size_t wave0 = 0, n_waves = 0;
    int cls = 0, G = 0;
};

inline double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

struct agx_sw_batch {
    agx_ctx *ctx = nullptr; // retained
    int family = 0;         // 0 int32, 1 packed signed, 2 packed biased, 3 int32 on the packed plan's coded image
    SwParams prm{};
    int64_t n_pairs = 0;
    DevBuf img, groups, waves, scores;
    DevBuf table; // substitution-matrix mode: kSwMatDim^2 int16 entries
    int rising = 0; // biased packed fill: 1 = stored values rise by |ge| per step, 4 = and by |ge| per column in classes of four (agx_sw_pk2_kernel.hip, KC)
    PinBuf out_stage; // page-locked landing block of the scores, taken at create: agx_sw_batch_scores allocates nothing
                      // (a first hipHostMalloc costs milliseconds, and hipvers' timed window is launch -> scores)
    bool matrix = false;
    std::vector<ClassLaunch> launches;
    agx_sw_info info{};
    // A batch created without the closing wait (the pieces of agx_sw_score): its upload, planning and pack kernels may
    // still be running; the temporaries they use, the event a launch has to wait for and the symbol check's verdict
    // are held here until finish_create().
    struct SwPending *pending = nullptr;
    // agx_sw_batch_bind_scores: a page-locked array of the caller's that launches write their scores into themselves
    // (only a batch planned in file order does: its waves then write consecutive bytes)
    bool file_order = false;
    int32_t *bound = nullptr;
};

namespace {
int create_batch(agx_ctx *ctx, const agx_sw_scoring *scoring, const agx_sw_matrix *matrix, const uint8_t *bases,
                 const uint64_t *off, const uint32_t *len, int64_t n_pairs, agx_sw_batch **out, bool defer = false);
int finish_create(agx_sw_batch *b);
void drop_pending(agx_sw_batch *b);
}

extern "C" {

void agx_sw_batch_destroy(agx_sw_batch *b)
{
    if (!b) return;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    drop_pending(b);
    b->img.release();
    b->groups.release();
    b->waves.release();
    b->scores.release();
    b->table.release();
    b->out_stage.release();
    agx_ctx_release(b->ctx); // the batch's own reference: a context outlives its batches
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
    AGX_GUARD_BEGIN
    return create_batch(ctx, scoring, nullptr, bases, off, len, n_pairs, out);
    AGX_GUARD_END("agx_sw_batch_create")
}

int agx_sw_batch_create_matrix(agx_ctx *ctx, const agx_sw_matrix *matrix, const uint8_t *bases, const uint64_t *off,
                               const uint32_t *len, int64_t n_pairs, agx_sw_batch **out)
{
    if (!matrix) {
        agx_set_error("agx_sw_batch_create_matrix: matrix is NULL");
        return AGX_E_ARG;
    }
    AGX_GUARD_BEGIN
    return create_batch(ctx, nullptr, matrix, bases, off, len, n_pairs, out);
    AGX_GUARD_END("agx_sw_batch_create_matrix")
}

} // extern "C"

namespace {

// Batches of at least this many pairs are candidates for the device planner: below it the host's threaded passes
// take well under a millisecond and the batch is in the tail regime anyway.
constexpr int64_t kDevPlanMinPairs = 49152;

// The full tiling table of a kernel family (every class allowed, no tail term, every shorter length up to the
// family's limit): what a device-planned batch is tiled with.  Made once per process.
const TilingTable &full_tiling_table(int family)
{
    static TilingTable tabs[4];
    static std::once_flag once[4];
    std::call_once(once[family], [family] {
        const std::vector<uint8_t> present((size_t)(family == 0 ? AGX_SW_MAX_SHORT_LEN : kSwPackedMaxShort) + 1, 1);
        build_tiling_table(tabs[family], present, class_costs(family), ~0u, 0.0);
    });
    return tabs[family];
}

// temporaries of a device-planned batch (released when create_batch leaves)
struct DevPlan {
    PinBuf h_len, h_buckets, h_padded;
    DevBuf d_len, d_buckets, d_padded, keys_a, keys_b, vals_a, vals_b, wkeys_a, wkeys_b, wids_a, wids_b, waves_tmp, temp;
    std::vector<uint32_t> hist; // pairs per (class, G) bucket
    size_t img_dw = 0;
    bool started = false;
    hipEvent_t uploaded = nullptr; // recorded on the copy stream behind this batch's sequences
    hipEvent_t done = nullptr; // recorded behind this batch's planning kernels (its own: two creates may be in flight on one context)
    void release()
    {
        if (done) (void)hipEventDestroy(done);
        if (uploaded) (void)hipEventDestroy(uploaded);
        done = uploaded = nullptr;
        for (PinBuf *x : {&h_len, &h_buckets, &h_padded}) x->release();
        for (DevBuf *x : {&d_len, &d_buckets, &d_padded, &keys_a, &keys_b, &vals_a, &vals_b, &wkeys_a, &wkeys_b, &wids_a, &wids_b, &waves_tmp, &temp})
            x->release();
    }
};

} // namespace

// everything a create leaves behind while its device work is still in flight
struct SwPending {
    agx_ctx *ctx = nullptr;
    DevBuf d_raw, d_off, d_code, d_flag;
    PinBuf h_groups, h_waves, h_flag, h_dense, h_dense_off;
    DevPlan dp;
    hipStream_t tail = nullptr; // the stream the pack kernel and the verdict's copy were queued on
    hipEvent_t ready = nullptr; // recorded behind them: what a launch waits for
    bool device_plan = false, matrix = false;
    ~SwPending()
    {
        // kernels may still read the temporaries when an error path (or a destroy without finish) gets here
        if (ctx && dp.started && ctx->plan) (void)hipStreamSynchronize(ctx->plan);
        if (tail) (void)hipStreamSynchronize(tail);
        if (ready) (void)hipEventDestroy(ready);
        dp.release();
        for (DevBuf *x : {&d_raw, &d_off, &d_code, &d_flag}) x->release();
        for (PinBuf *x : {&h_groups, &h_waves, &h_flag, &h_dense, &h_dense_off}) x->release();
    }
};

namespace {

void drop_pending(agx_sw_batch *b)
{
    delete b->pending;
    b->pending = nullptr;
}

// the closing wait of a deferred create: the symbol check's verdict, the device planner's padded-cell count
int finish_create(agx_sw_batch *b)
{
    SwPending *p = b->pending;
    if (!p) return AGX_OK;
    const hipError_t e = hipStreamSynchronize(p->tail);
    int rc = AGX_OK;
    if (e != hipSuccess) {
        agx_set_error("agx_sw_batch_create: upload -> %s", hipGetErrorString(e));
        rc = AGX_E_HIP;
    } else {
        if (p->device_plan) b->info.padded_cells = (int64_t) * (const unsigned long long *)p->dp.h_padded.p;
        const uint32_t *flag = (const uint32_t *)p->h_flag.p;
        if (flag && flag[0]) {
            if (p->matrix)
                agx_set_error("pair %u contains a byte outside the substitution matrix's alphabet", flag[1]);
            else
                agx_set_error("pair %u contains byte 0x00, which is reserved as the padding symbol", flag[1]);
            rc = AGX_E_SYMBOL;
        }
    }
    drop_pending(b);
    return rc;
}

// the family-2 tiling table on the device, made on the context's first device-planned batch
int ensure_device_table(agx_ctx *ctx, hipStream_t s)
{
    if (ctx->sw_seg_first) return AGX_OK;
    const TilingTable &tt = full_tiling_table(2);
    std::vector<uint32_t> segs(tt.segs.size());
    for (size_t k = 0; k < segs.size(); ++k) segs[k] = (tt.segs[k].ly_from & 0xffffu) | ((uint32_t)tt.segs[k].cls << 16) | ((uint32_t)tt.segs[k].G << 24);
    void *d_first = nullptr, *d_segs = nullptr;
    AGX_HIP(hipMalloc(&d_first, tt.first.size() * sizeof(uint32_t)));
    if (hipMalloc(&d_segs, std::max<size_t>(segs.size(), 1) * sizeof(uint32_t)) != hipSuccess) {
        (void)hipFree(d_first);
        agx_set_error("device tiling table: out of device memory");
        return AGX_E_NOMEM;
    }
    // (pageable sources: these two copies return when the data has left the host buffers)
    hipError_t e = hipMemcpyAsync(d_first, tt.first.data(), tt.first.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && !segs.empty()) e = hipMemcpyAsync(d_segs, segs.data(), segs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipFree(d_first);
        (void)hipFree(d_segs);
        agx_set_error("device tiling table: upload -> %s", hipGetErrorString(e));
        return AGX_E_HIP;
    }
    ctx->sw_seg_first = d_first;
    ctx->sw_segs = d_segs;
    return AGX_OK;
}

// Allocates the batch's record arrays and the planner's temporaries, uploads len[] and the bucket table and queues the
// planning kernels on the context's planning stream (the pack kernel follows them there).
int launch_device_plan(agx_ctx *ctx, DevPlan &dp, agx_sw_batch *b, uint32_t n_pairs, uint32_t n_fill, uint32_t longest_long, int slots, uint32_t img0,
                       size_t n_groups, size_t n_waves)
{
    hipStream_t ps = ctx->plan;
    agx_sw_plan_preload();
    int rc;
    {
        static std::mutex once_per_context; // the table is made by whichever create comes first
        std::lock_guard<std::mutex> l(once_per_context);
        rc = ensure_device_table(ctx, ps);
    }
    if (!rc && (hipEventCreateWithFlags(&dp.done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&dp.uploaded, hipEventDisableTiming) != hipSuccess)) {
        agx_set_error("device planner: cannot create an event");
        rc = AGX_E_HIP;
    }
    const size_t pw = (size_t)n_pairs * sizeof(uint32_t), ww = std::max<size_t>(n_waves, 1) * sizeof(uint32_t);
    const size_t temp_bytes = agx_sw_plan_temp_bytes(n_pairs, (uint32_t)n_waves);
    if (!rc) rc = b->groups.alloc(ctx, std::max<size_t>(n_groups, 1) * sizeof(SwGroup2));
    if (!rc) rc = b->waves.alloc(ctx, std::max<size_t>(n_waves, 1) * sizeof(SwWave));
    if (!rc) rc = dp.d_len.alloc(ctx, 2 * pw);
    if (!rc) rc = dp.d_buckets.alloc(ctx, dp.h_buckets.bytes);
    if (!rc) rc = dp.d_padded.alloc(ctx, 64);
    if (!rc) rc = dp.h_padded.alloc(ctx, 64);
    for (DevBuf *x : {&dp.keys_a, &dp.keys_b, &dp.vals_a, &dp.vals_b})
        if (!rc) rc = x->alloc(ctx, pw);
    for (DevBuf *x : {&dp.wkeys_a, &dp.wkeys_b, &dp.wids_a, &dp.wids_b})
        if (!rc) rc = x->alloc(ctx, ww);
    if (!rc) rc = dp.waves_tmp.alloc(ctx, std::max<size_t>(n_waves, 1) * sizeof(SwWave));
    if (!rc) rc = dp.temp.alloc(ctx, temp_bytes);
    if (rc) return rc;
    dp.started = true;
    AGX_HIP(hipMemcpyAsync(dp.d_len.p, dp.h_len.p, 2 * pw, hipMemcpyHostToDevice, ps));
    AGX_HIP(hipMemcpyAsync(dp.d_buckets.p, dp.h_buckets.p, dp.h_buckets.bytes, hipMemcpyHostToDevice, ps));
    AGX_HIP(hipMemsetAsync(dp.d_padded.p, 0, 8, ps));
    SwPlanArgs a{};
    a.len = (const uint32_t *)dp.d_len.p;
    a.n_pairs = n_pairs;
    a.n_fill = n_fill;
    a.seg_first = (const uint32_t *)ctx->sw_seg_first;
    a.segs = (const uint32_t *)ctx->sw_segs;
    a.longest = longest_long;
    a.buckets = (const uint32_t *)dp.d_buckets.p;
    a.img0 = img0;
    a.slots = slots;
    a.n_waves = (uint32_t)n_waves;
    a.keys_a = (uint32_t *)dp.keys_a.p, a.keys_b = (uint32_t *)dp.keys_b.p, a.vals_a = (uint32_t *)dp.vals_a.p, a.vals_b = (uint32_t *)dp.vals_b.p;
    a.wave_keys_a = (uint32_t *)dp.wkeys_a.p, a.wave_keys_b = (uint32_t *)dp.wkeys_b.p, a.wave_ids_a = (uint32_t *)dp.wids_a.p, a.wave_ids_b = (uint32_t *)dp.wids_b.p;
    a.waves_tmp = (SwWave *)dp.waves_tmp.p;
    a.waves = (SwWave *)b->waves.p;
    a.groups = (uint32_t *)b->groups.p;
    a.padded = (unsigned long long *)dp.d_padded.p;
    a.temp = dp.temp.p;
    a.temp_bytes = temp_bytes;
    a.n_cu = ctx->n_cu;
    if (agx_sw_plan_launch(a, ps)) {
        agx_set_error("device planner: launch failed: %s", hipGetErrorString(hipGetLastError()));
        return AGX_E_HIP;
    }
    AGX_HIP(hipMemcpyAsync(dp.h_padded.p, dp.d_padded.p, 8, hipMemcpyDeviceToHost, ps));
    AGX_HIP(hipEventRecord(dp.done, ps));
    return AGX_OK;
}

int create_batch(agx_ctx *ctx, const agx_sw_scoring *scoring, const agx_sw_matrix *matrix, const uint8_t *bases,
                 const uint64_t *off, const uint32_t *len, int64_t n_pairs, agx_sw_batch **out, bool defer)
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
    prm.age2 = twice(-prm.ge);
    prm.agf2 = twice(-prm.gf);
    const int bias = 0x0400 + std::max(-prm.gf - prm.ge, prm.delta);
    prm.bias2 = twice(bias);

    const bool trace = agx_tune("AGX_TRACE_CREATE") != nullptr;
    const double t_begin = now_ms();
    if (n_pairs > 0 && !bases) {
        for (int64_t p = 0; p < 2 * n_pairs; ++p)
            if (len[p]) {
                agx_set_error("agx_sw_batch_create: bases is NULL");
                return AGX_E_ARG;
            }
    }

    // ---- where the per-pair passes of the planner will run.  A large batch for the packed biased fill is a candidate
    // for the device planner (agx_sw_plan_kernel.hip): pass A then only reduces -- no per-pair record is written on the
    // host -- and the batch-level rules decide below whether the candidate stands (a uniform batch, one in the tail
    // regime or with a dominant shape is planned on the host as before).
    const int want = tuned_kernel() ? tuned_kernel() : ctx ? ctx->opt_sw_kernel : AGX_SW_KERNEL_AUTO;
    const int planner_opt = ctx ? ctx->opt_sw_planner : AGX_SW_PLANNER_HOST;
    bool dev_candidate = ctx && !matrix && planner_opt != AGX_SW_PLANNER_HOST && n_pairs > 0 &&
                         (planner_opt == AGX_SW_PLANNER_DEVICE || n_pairs >= kDevPlanMinPairs) &&
                         (want == AGX_SW_KERNEL_AUTO || want == AGX_SW_KERNEL_PACKED_BIASED) && tail_beta_override() < 0 &&
                         !max_classes_override() && !agx_tune("AGX_SW_SORT_WAVES") && !agx_tune("AGX_SW_ONE_LAUNCH") &&
                         !agx_tune("AGX_SW_WAVES_PER_CLASS") && !agx_tune("AGX_SW_HOST_PLAN");

    // ---- pass A (threads over pairs): orient, check the limits, count cells, extent of `bases`
    std::vector<PairPlan> all;
    if (!dev_candidate) all.resize((size_t)n_pairs);
    PairPlan *allp = dev_candidate ? nullptr : all.data();
    const uint32_t hard_max_short = matrix ? (uint32_t)kSwPackedMaxShort : AGX_SW_MAX_SHORT_LEN; // no wide classes in matrix mode
    struct Worker {
        int rc = AGX_OK;
        int64_t bad_pair = -1;
        int64_t cells = 0, votes = 0;
        uint32_t longest_short = 0, longest_long = 0;
        uint32_t shape0 = 0xffffffffu; // first (lx, ly) this part saw
        bool mixed = false;            // ... and whether it saw another one
        int64_t n_fill = 0;
        uint64_t lo = ~0ull, hi = 0, sum_len = 0;
        double waves = 0;
        double class_work[kSwNumClasses] = {};
        std::vector<uint8_t> present; // [lx] != 0: this part saw that shorter length
    };
    std::vector<Worker> wk((size_t)agx_host_threads());
    agx_parallel_for(n_pairs, 8192, [&](int64_t lo, int64_t hi, int tid) {
        Worker &me = wk[(size_t)tid];
        me.present.assign((size_t)hard_max_short + 1, 0);
        for (int64_t p = lo; p < hi; ++p) {
            PairPlan scratch;
            PairPlan &pp = allp ? allp[(size_t)p] : scratch;
            pp = PairPlan{};
            pp.pair = (uint32_t)p;
            pp.cls = kClsEmpty;
            const uint32_t la = len[2 * p], lb = len[2 * p + 1];
            me.cells += (int64_t)la * lb;
            me.sum_len += (uint64_t)la + lb;
            if (la) {
                me.lo = std::min(me.lo, off[2 * p]);
                me.hi = std::max(me.hi, off[2 * p] + la);
            }
            if (lb) {
                me.lo = std::min(me.lo, off[2 * p + 1]);
                me.hi = std::max(me.hi, off[2 * p + 1] + lb);
            }
            if (la == 0 || lb == 0) continue; // no interior cell: score stays 0
            const bool second_short = lb < la; // ties keep file order (antidiagonalSmithWaterman.c:229-244)
            const uint32_t lx = second_short ? lb : la, ly = second_short ? la : lb;
            if (lx > hard_max_short || ly > 0xffffu) {
                if (me.rc == AGX_OK) {
                    me.rc = AGX_E_LIMIT;
                    me.bad_pair = p;
                }
                continue;
            }
            me.longest_short = std::max(me.longest_short, lx);
            me.longest_long = std::max(me.longest_long, ly);
            me.present[lx] = 1;
            const uint32_t key = lx << 16 | ly;
            if (me.shape0 == 0xffffffffu) me.shape0 = key;
            else if (key != me.shape0) me.mixed = true;
            ++me.n_fill;
            pp.lxo = (uint16_t)(lx | (second_short ? 0x8000u : 0u));
            pp.ly = ly;
            pp.cls = kClsUntiled;
        }
    });
    int64_t cells = 0;
    uint32_t longest_short = 0, longest_long = 0;
    uint64_t ext_lo = ~0ull, ext_hi = 0, sum_len = 0;
    for (const Worker &w : wk) {
        cells += w.cells;
        longest_short = std::max(longest_short, w.longest_short);
        longest_long = std::max(longest_long, w.longest_long);
        ext_lo = std::min(ext_lo, w.lo);
        ext_hi = std::max(ext_hi, w.hi);
        sum_len += w.sum_len;
    }
    for (const Worker &w : wk)
        if (w.rc != AGX_OK) {
            const int64_t p = w.bad_pair;
            agx_set_error("pair %lld: lengths %u x %u exceed the supported %u x 65535 (shorter x longer)", (long long)p,
                          len[2 * p], len[2 * p + 1], hard_max_short);
            return w.rc;
        }
    if (ext_hi < ext_lo) ext_lo = ext_hi = 0; // no byte at all
    std::vector<uint8_t> present((size_t)longest_short + 1, 0);
    for (const Worker &w : wk)
        for (size_t lx = 0; lx < present.size() && lx < w.present.size(); ++lx) present[lx] |= w.present[lx];
    const double t_pass_a = now_ms();

    // ---- kernel family.  The packed int16 kernels cover shorter sides up to 64 x 40 columns; one longer
    // pair moves the whole batch to the int32 kernel, which also has the wide classes (up to 64 x 160).
    // Biased formulation: every stored half must be the pattern of a positive normal half-precision number,
    // [0x0400, 0x7c00): smallest B - max(|gf| + |ge|, delta), largest B + (longest shorter side + 1) * match + |gf|.
    int family = want == AGX_SW_KERNEL_INT32 ? 0 : want == AGX_SW_KERNEL_PACKED_SIGNED ? 1 : 2;
    if (matrix || longest_short > (uint32_t)kSwPackedMaxShort) family = 0; // the matrix lookup exists in the int32 kernel only
    // The int32 kernel asked for (BASELINE config 2 as worded) runs the packed plan and its DNA-coded image in 32-bit state,
    // one pair of a lane group at a time (agx_sw_i32d_kernel.hip: 7.5 instead of 8.5 instructions per cell), wherever the
    // coded match exists: delta and mismatch + |gf| must be bytes, the shorter sides within the packed plan's 2560 columns.
    if (family == 0 && !matrix && longest_short <= (uint32_t)kSwPackedMaxShort && prm.delta < 128 && prm.hd >= prm.delta && !agx_tune("AGX_SW_I32_CLASSIC")) family = 3;
    if (family == 2 && !((int64_t)bias + ((int64_t)longest_short + 1) * sc.match - prm.gf < 0x7c00)) family = 1;
    // ... and its rising-offset variant adds (steps + 2) |ge| on top, steps <= longest longer side + 63
    int rising = family == 2 &&
                 (int64_t)bias + ((int64_t)longest_short + 1) * sc.match - prm.gf + ((int64_t)longest_long + 66 + 3) * -(int64_t)prm.ge < 0x7c00;
    // ... with column classes when the wrapping column's diagonal constant, mismatch + |gf| - 3 |ge|, is not negative
    if (rising && prm.hd - prm.delta + 3 * prm.ge >= 0) rising = 4;
    if (const char *e = agx_tune("AGX_SW_RISE")) rising = e[0] == '0' ? 0 : e[0] == '1' && rising ? 1 : rising;
    const bool packed = family != 0;
    const double *costs = class_costs(family);
    const int slots = packed ? 2 : 1;

    // plan-only batches check the symbols on the host; with a device the pack kernel does it
    if (!ctx && n_pairs > 0) {
        std::vector<int64_t> bad((size_t)agx_host_threads(), -1);
        agx_parallel_for(n_pairs, 4096, [&](int64_t lo, int64_t hi, int tid) {
            for (int64_t p = lo; p < hi && bad[(size_t)tid] < 0; ++p) {
                const uint32_t la = len[2 * p], lb = len[2 * p + 1];
                if (la == 0 || lb == 0) continue;
                const uint8_t *q = bases + off[2 * p], *r = bases + off[2 * p + 1];
                bool hit = false;
                if (matrix) {
                    for (uint32_t k = 0; k < la; ++k) hit |= matrix->code[q[k]] == 0xff;
                    for (uint32_t k = 0; k < lb; ++k) hit |= matrix->code[r[k]] == 0xff;
                } else
                    hit = memchr(q, 0, la) || memchr(r, 0, lb);
                if (hit) bad[(size_t)tid] = p;
            }
        });
        int64_t first_bad = -1;
        for (int64_t v : bad)
            if (v >= 0 && (first_bad < 0 || v < first_bad)) first_bad = v;
        if (first_bad >= 0) {
            if (matrix)
                agx_set_error("pair %lld contains a byte outside the substitution matrix's alphabet", (long long)first_bad);
            else
                agx_set_error("pair %lld contains byte 0x00, which is reserved as the padding symbol", (long long)first_bad);
            return AGX_E_SYMBOL;
        }
    }

    agx_sw_batch *b = new agx_sw_batch();
    struct Guard { // error paths: free whatever the batch holds
        agx_sw_batch *&b;
        ~Guard()
        {
            if (b) agx_sw_batch_destroy(b);
        }
    } guard{b};
    agx_ctx_retain(ctx);
    b->ctx = ctx;
    b->n_pairs = n_pairs;
    b->family = family;
    b->rising = rising;
    if (ctx) { // the code objects this batch will launch from, loaded now rather than inside its first launch
        agx_sw_pack_preload();
        agx_copy_preload();
        if (family == 2) agx_sw_pk2_preload();
        if (family == 3) agx_sw_i32d_preload();
        if (family == 0 && !matrix) agx_sw_i32_preload();
    }
    b->matrix = matrix != nullptr;
    b->prm = prm;
    b->prm.n_out = (uint32_t)n_pairs + 1u;

    // ---- the caller's arrays start travelling now, while the plan is made: a helper thread drives the
    // copies (a pageable source makes hipMemcpyAsync block while the runtime stages it)
    std::unique_ptr<SwPending> tmp(new SwPending()); // released when this function leaves, or kept by the batch (defer)
    tmp->ctx = ctx;
    tmp->matrix = matrix != nullptr;
    DevBuf &d_raw = tmp->d_raw, &d_off = tmp->d_off, &d_code = tmp->d_code, &d_flag = tmp->d_flag;
    PinBuf &h_groups = tmp->h_groups, &h_waves = tmp->h_waves, &h_flag = tmp->h_flag, &h_dense = tmp->h_dense, &h_dense_off = tmp->h_dense_off;
    // `bases` is normally dense; a caller whose sequences are islands in a much larger array gets a dense
    // copy (pinned, its own offsets) instead of an upload of the gaps
    const bool dense_copy = (ext_hi - ext_lo) > 4 * sum_len + ((uint64_t)64 << 20);
    const uint64_t raw_bytes = dense_copy ? sum_len : ext_hi - ext_lo;
    const uint64_t raw_base = dense_copy ? 0 : ext_lo;
    int up_rc = AGX_OK;
    char up_err[300] = "";
    auto upload_inputs = [&]() {
        try {
            if (hipSetDevice(ctx->device) != hipSuccess) {
                up_rc = AGX_E_HIP;
                snprintf(up_err, sizeof up_err, "hipSetDevice failed on the upload thread");
                return;
            }
            int r = d_raw.alloc(ctx, (size_t)raw_bytes + 2 * kRawPad); // the pack kernels read 16-byte pieces that may begin before / end behind the sequences
            if (!r) r = d_off.alloc(ctx, (size_t)n_pairs * 2 * sizeof(uint64_t));
            if (!r) r = d_flag.alloc(ctx, 2 * sizeof(uint32_t));
            if (!r && matrix) r = d_code.alloc(ctx, 256);
            const uint8_t *src = bases + ext_lo;
            const uint64_t *src_off = off;
            if (!r && dense_copy) {
                r = h_dense.alloc(ctx, (size_t)sum_len + 16);
                if (!r) r = h_dense_off.alloc(ctx, (size_t)n_pairs * 2 * sizeof(uint64_t));
                if (!r) {
                    uint64_t *o = (uint64_t *)h_dense_off.p, at = 0;
                    for (int64_t k = 0; k < 2 * n_pairs; ++k) {
                        o[k] = at;
                        if (len[k]) memcpy((uint8_t *)h_dense.p + at, bases + off[k], len[k]);
                        at += len[k];
                    }
                    src = (const uint8_t *)h_dense.p;
                    src_off = o;
                }
            }
            if (r) {
                up_rc = r;
                snprintf(up_err, sizeof up_err, "%s", agx_last_error());
                return;
            }
            hipError_t e = hipSuccess;
            // A pageable source: the runtime's own staging moved fresh pages at 4-5 GB/s (25-35 ms per 143 MB chunk
            // of the command line).  Staged here instead: slices are copied by a few threads into a ring of pinned
            // blocks, each slice's DMA runs while the next is being copied.  Pinned sources go down in one DMA.
            constexpr size_t kSlice = (size_t)16 << 20;
            constexpr int kRing = 3;
            // (known page-locked = allocated by agx_host_alloc; anything else is treated as pageable)
            const bool pageable = raw_bytes > 2 * kSlice && !dense_copy && !agx_is_pinned_host(src, (size_t)raw_bytes);
            if (pageable) {
                PinBuf ring[kRing];
                hipEvent_t done[kRing] = {nullptr, nullptr, nullptr};
                int r = AGX_OK;
                for (int k = 0; k < kRing && !r; ++k) {
                    r = ring[k].alloc(ctx, kSlice);
                    if (!r && hipEventCreateWithFlags(&done[k], hipEventDisableTiming) != hipSuccess) r = AGX_E_HIP;
                }
                size_t at = 0;
                for (int k = 0; !r && at < raw_bytes; ++k, at += kSlice) {
                    const int slot = k % kRing;
                    const size_t n = std::min(kSlice, (size_t)raw_bytes - at);
                    if (k >= kRing && hipEventSynchronize(done[slot]) != hipSuccess) r = AGX_E_HIP;
                    const int parts = std::max(1, agx_host_threads()); // every pool thread: 4 (round 2) and 8 copied a 16 MB slice in 0.4 ms, above its 0.29 ms DMA
                    const size_t per = (n + parts - 1) / parts;
                    agx_pool_run(parts, [&](int t) {
                        const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
                        if (lo < hi) agx_stream_copy((uint8_t *)ring[slot].p + lo, src + at + lo, hi - lo);
                    });
                    if (!r && (hipMemcpyAsync((uint8_t *)d_raw.p + kRawPad + at, ring[slot].p, n, hipMemcpyHostToDevice, ctx->copy) != hipSuccess ||
                               hipEventRecord(done[slot], ctx->copy) != hipSuccess))
                        r = AGX_E_HIP;
                }
                for (int k = 0; k < kRing; ++k) {
                    if (done[k]) {
                        (void)hipEventSynchronize(done[k]); // the ring goes back to the pool: its DMAs must be over
                        (void)hipEventDestroy(done[k]);
                    }
                    ring[k].release();
                }
                if (r) e = hipErrorUnknown;
            } else if (raw_bytes)
                e = hipMemcpyAsync((uint8_t *)d_raw.p + kRawPad, src, (size_t)raw_bytes, hipMemcpyHostToDevice, ctx->copy);
            if (e == hipSuccess)
                e = hipMemcpyAsync(d_off.p, src_off, (size_t)n_pairs * 2 * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->copy);
            if (e == hipSuccess && matrix) e = hipMemcpyAsync(d_code.p, matrix->code, 256, hipMemcpyHostToDevice, ctx->copy);
            if (e == hipSuccess) e = hipMemsetAsync(d_flag.p, 0, 4, ctx->copy);                    // [0] offending pairs
            if (e == hipSuccess) e = hipMemsetAsync((char *)d_flag.p + 4, 0xff, 4, ctx->copy);     // [1] smallest of them
            if (e != hipSuccess) {
                up_rc = AGX_E_HIP;
                snprintf(up_err, sizeof up_err, "upload of the sequences -> %s", hipGetErrorString(e));
            }
        } catch (const std::exception &ex) {
            up_rc = AGX_E_NOMEM;
            snprintf(up_err, sizeof up_err, "upload thread: %s", ex.what());
        }
    };
    std::thread uploader;
    struct Joiner {
        std::thread &t;
        ~Joiner()
        {
            if (t.joinable()) t.join();
        }
    } joiner{uploader};
    if (ctx && n_pairs > 0) {
        // page-locked arrays (agx_host_alloc) go down without the runtime staging anything: hipMemcpyAsync returns at once,
        // and the helper thread -- 50 us to start and join, a tenth of a config-2-sized call -- is not needed
        const bool queued_at_once = !dense_copy && !matrix && agx_is_pinned_host(bases + ext_lo, (size_t)raw_bytes) &&
                                    agx_is_pinned_host(off, (size_t)n_pairs * 2 * sizeof(uint64_t));
        if (queued_at_once) upload_inputs();
        else uploader = std::thread(upload_inputs);
    }

    // one shape in the whole batch?
    int64_t n_fill = 0;
    uint32_t shape = 0xffffffffu;
    bool one_shape = true;
    for (const Worker &w : wk) {
        n_fill += w.n_fill;
        if (w.mixed) one_shape = false;
        if (w.shape0 != 0xffffffffu) {
            if (shape == 0xffffffffu) shape = w.shape0;
            else if (shape != w.shape0) one_shape = false;
        }
    }
    const bool uniform = one_shape && n_fill == n_pairs && n_pairs >= 1024 && n_cu > 0;

    // ---- device planner: the candidate's verdict.  Pass A2 (threads over pairs) tiles every pair by lookup in the
    // full table and only COUNTS: pairs per (class, G) bucket -- from which every group, wave and record offset
    // follows without a scan -- image words, estimated waves, the votes of the sampled dominant shape; it also
    // copies len[] into page-locked memory for its upload.  The rules that need another tiling (tail regime,
    // class consolidation, dominant shape) send the batch to the host planner.
    bool device_plan = false;
    DevPlan &dp = tmp->dp;
    if (dev_candidate && family == 2 && !uniform && n_cu > 0 && n_fill > 0 && longest_short <= (uint32_t)kSwPackedMaxShort) {
        const TilingTable &tt = full_tiling_table(family);
        rc = dp.h_len.alloc(ctx, (size_t)n_pairs * 2 * sizeof(uint32_t));
        if (rc) return rc;
        uint32_t cand = 0;
        int votes = 0;
        {
            const size_t stride = std::max<size_t>(1, (size_t)n_pairs / 512);
            for (size_t k = 0; k < 512 && k * stride < (size_t)n_pairs; ++k) { // Boyer-Moore majority vote over a sample
                const uint32_t la = len[2 * k * stride], lb = len[2 * k * stride + 1];
                if (la == 0 || lb == 0) continue;
                const uint32_t sh = std::min(la, lb) << 16 | (std::max(la, lb) & 0xffffu);
                if (votes == 0) {
                    cand = sh;
                    votes = 1;
                } else
                    votes += sh == cand ? 1 : -1;
            }
        }
        struct Count {
            std::vector<uint32_t> hist;
            uint64_t words = 0;
            double waves = 0;
            int64_t votes = 0;
            bool untiled = false;
        };
        std::vector<Count> cnt((size_t)agx_host_threads());
        uint32_t *pinned_len = (uint32_t *)dp.h_len.p;
        agx_parallel_for(n_pairs, 8192, [&](int64_t lo, int64_t hi, int tid) {
            Count &me = cnt[(size_t)tid];
            me.hist.assign((size_t)kSwPlanBuckets, 0);
            memcpy(pinned_len + 2 * lo, len + 2 * lo, (size_t)(hi - lo) * 2 * sizeof(uint32_t));
            for (int64_t p = lo; p < hi; ++p) {
                const uint32_t la = len[2 * p], lb = len[2 * p + 1];
                if (la == 0 || lb == 0) continue;
                const uint32_t lx = std::min(la, lb), ly = std::max(la, lb);
                Tiling tl;
                if (!tt.pick(lx, ly, &tl, nullptr)) {
                    me.untiled = true;
                    continue;
                }
                ++me.hist[(size_t)tl.cls * 64 + (size_t)(64 - tl.G)];
                me.words += ((uint64_t)tl.G * kSwClasses[tl.cls] + 3) / 4 + 1 + ((uint64_t)ly + 3) / 4;
                me.waves += (double)tl.G / 64.0 / slots;
                me.votes += (lx << 16 | (ly & 0xffffu)) == cand;
            }
        });
        dp.hist.assign((size_t)kSwPlanBuckets, 0);
        uint64_t words = 0;
        double waves_est = 0;
        int64_t cand_count = 0;
        bool untiled = false;
        for (const Count &c : cnt) {
            if (c.hist.empty()) continue;
            for (int k = 0; k < kSwPlanBuckets; ++k) dp.hist[(size_t)k] += c.hist[(size_t)k];
            words += c.words;
            waves_est += c.waves;
            cand_count += c.votes;
            untiled |= c.untiled;
        }
        const double fill = waves_est / (5.0 * 4.0 * n_cu);
        const size_t img_dw_dev = (size_t)kSwPackedMaxShort / 4 + 1 + words;
        device_plan = !untiled && fill >= 0.48 && waves_est >= 2048.0 && !(votes > 0 && cand_count * 2 >= n_pairs) && img_dw_dev <= 0xffffffffull;
        dp.img_dw = img_dw_dev;
        if (trace)
            fprintf(stderr, "[agx_sw_batch_create] device planner verdict %d: untiled %d, waves %.0f (fill %.2f), sampled shape votes %d -> %lld of %lld pairs, image %zu words\n",
                    (int)device_plan, (int)untiled, waves_est, fill, votes, (long long)cand_count, (long long)n_pairs, img_dw_dev);
    } else if (trace)
        fprintf(stderr, "[agx_sw_batch_create] no device planner: candidate %d, family %d, uniform %d, n_fill %lld, longest shorter side %u\n",
                (int)dev_candidate, family, (int)uniform, (long long)n_fill, longest_short);
    if (dev_candidate && !device_plan) { // the host planner after all: its per-pair records, as pass A would have written them
        all.resize((size_t)n_pairs);
        allp = all.data();
        agx_parallel_for(n_pairs, 8192, [&](int64_t lo, int64_t hi, int) {
            for (int64_t p = lo; p < hi; ++p) {
                PairPlan &pp = allp[(size_t)p];
                pp = PairPlan{};
                pp.pair = (uint32_t)p;
                pp.cls = kClsEmpty;
                const uint32_t la = len[2 * p], lb = len[2 * p + 1];
                if (la == 0 || lb == 0) continue;
                const bool second_short = lb < la;
                pp.lxo = (uint16_t)((second_short ? lb : la) | (second_short ? 0x8000u : 0u));
                pp.ly = second_short ? la : lb;
                pp.cls = kClsUntiled;
            }
        });
        dp.h_len.release();
    }

    // ---- uniform batches (fixed-length reads: BASELINE config 2): one shape, so one tiling -- chosen with the
    // wave-count quantisation below -- and nothing to sort: file order is already the order passes B and C
    // would produce.
    std::vector<PairPlan> plan;
    bool planned = false;
    if (uniform) {
        const Tiling tl = choose_tiling_uniform(costs, slots, (int)(shape >> 16), (int)(shape & 0xffffu), n_pairs, 4 * n_cu);
        if (tl.cls >= 0 && !(matrix && kSwClasses[tl.cls] > 40)) {
            agx_parallel_for(n_pairs, 32768, [&](int64_t lo, int64_t hi, int) {
                for (int64_t p = lo; p < hi; ++p) {
                    all[(size_t)p].cls = (uint8_t)tl.cls;
                    all[(size_t)p].G = (uint8_t)tl.G;
                }
            });
            plan.swap(all);
            planned = true;
            b->file_order = true;
        }
    }
    double t_plan = now_ms(), t_sort = t_plan;

    // ---- pass B: lane tiling per pair, then the batch-level rules.  Host-only work from here to the records.
    if (!planned && !device_plan) {
    TilingTable tt;
    double beta_used = tail_beta_override() >= 0 ? tail_beta_override() : 0.0;
    auto tile_all = [&](uint32_t allowed, double beta, bool only_outside) {
        build_tiling_table(tt, present, costs, allowed, beta);
        for (Worker &w : wk) {
            w.waves = 0;
            w.rc = AGX_OK;
            for (double &c : w.class_work) c = 0;
        }
        agx_parallel_for(n_pairs, 8192, [&](int64_t lo, int64_t hi, int tid) {
            Worker &me = wk[(size_t)tid];
            for (int64_t p = lo; p < hi; ++p) {
                PairPlan &pp = all[(size_t)p];
                if (pp.cls == kClsEmpty) continue;
                Tiling tl;
                double cost = 0;
                const bool keep_own = only_outside && pp.cls < kSwNumClasses && ((allowed >> pp.cls) & 1u);
                if (!keep_own && tt.pick(pp.lx(), pp.ly, &tl, &cost) && !(matrix && kSwClasses[tl.cls] > 40)) {
                    pp.cls = (uint8_t)tl.cls;
                    pp.G = (uint8_t)tl.G;
                } else if (pp.cls == kClsUntiled) { // no class spans this pair (cannot happen within the limits)
                    if (me.rc == AGX_OK) {
                        me.rc = AGX_E_LIMIT;
                        me.bad_pair = p;
                    }
                    continue;
                } else // keeps the class it has
                    cost = (double)(pp.ly + pp.G - 1) * tiling_slope(costs, pp.cls, pp.G, beta);
                me.class_work[pp.cls] += cost;
                me.waves += (double)pp.G / 64.0 / slots;
            }
        });
    };
    tile_all(~0u, beta_used, false);
    for (const Worker &w : wk)
        if (w.rc != AGX_OK) {
            agx_set_error("pair %lld: no lane tiling fits its %u columns", (long long)w.bad_pair, all[(size_t)w.bad_pair].lx());
            return w.rc;
        }
    // Tail regime: when the planned waves fill the chip's resident capacity (about 5 per SIMD for this
    // kernel) less than 1.6 times, a launch lasts as long as its longest waves -- alone on their SIMDs
    // in a small batch, or stranded in a mostly empty second filling.  Such a batch is re-tiled with a
    // term on a wave's own duration (3 lanes' worth; more the emptier the chip): long pairs spread over
    // more lanes, waves get shorter and more numerous (tools/sw_tail_beta_sweep.py, sw_tail_rule_check.py:
    // mixed 32..512 pairs, 8192 pairs 1.24 -> 2.6 TCUPS, 16 384 2.46 -> 2.9, 131 072 4.15 -> 4.61).  Beyond
    // 1.6 fillings the term costs 1-3 % and is left out.  (Uniform batches are re-tiled below with their
    // own wave-count model.)
    if (tail_beta_override() < 0 && n_cu > 0) {
        double waves_est = 0;
        for (const Worker &w : wk) waves_est += w.waves;
        const double fill = waves_est / (5.0 * 4.0 * n_cu);
        // (the biased packed fill runs two waves per SIMD, and with its round-2b cell the term pays up to 1.2 of ITS
        // fillings = 0.48 of these: 45 056 mixed pairs 6.5 -> 7.2 TCUPS, 49 152 7.15 -> 7.26, but 57 344 7.74 -> 7.55 and
        // 65 536 8.07 -> 7.71 -- tools/sw_tail_rule_check.py)
        if (fill < (family >= 2 ? 0.48 : 1.6)) {
            beta_used = fill < 0.1 ? 10.0 : fill < 0.4 ? 6.0 : 3.0;
            tile_all(~0u, beta_used, false);
        }
    }
    // Every class is its own launch and the measured cost curve is flat over many widths: a mixed batch
    // keeps the classes that carry most of the work -- about one per 4096 wavefronts, at most 6
    // (tools/sw_mixed_sweep.py: 16384 pairs of 32..512 went from 0.63 to 2.5 TCUPS, 65536 from 2.1 to 3.9) --
    // and re-tiles the other pairs among them (a pair no kept class can span keeps its own).
    {
        double work[kSwNumClasses] = {};
        double waves_est = 0;
        for (const Worker &w : wk) {
            waves_est += w.waves;
            for (int c = 0; c < kSwNumClasses; ++c) work[c] += w.class_work[c];
        }
        static const double per_class = [] {
            const char *e = agx_tune("AGX_SW_WAVES_PER_CLASS");
            return e && atof(e) > 0 ? atof(e) : 4096.0;
        }();
        // The biased packed kernel runs every class in ONE launch (sw_fill_pk2_any), so from about two wavefronts
        // per SIMD on it keeps them all: padding shrinks (useful cells 0.908 -> 0.932 on config 4's per-GPU shard)
        // and nothing is forked or joined: 131 072 mixed pairs 5.62 -> 6.02 TCUPS, 262 144 6.05 -> 6.39, 65 536
        // 5.48 -> 5.81.  Smaller batches are in the tail regime, where the launch lasts as long as its longest
        // waves and the few-classes rule still wins (16 384 pairs: 3.58 against 3.02 TCUPS; tools/sw_mixed_check.py).
        const int k_max = max_classes_override()                  ? std::min(max_classes_override(), 1 + (int)(waves_est / per_class))
                          : family >= 2 && waves_est >= 2048.0 ? kSwNumClasses
                                                                  : std::min(6, 1 + (int)(waves_est / per_class));
        int used = 0;
        for (int c = 0; c < kSwNumClasses; ++c) used += work[c] > 0;
        if (used > k_max) {
            int order[kSwNumClasses];
            for (int c = 0; c < kSwNumClasses; ++c) order[c] = c;
            std::sort(order, order + kSwNumClasses, [&](int x, int y) { return work[x] > work[y]; });
            uint32_t keep = 0;
            for (int k = 0; k < k_max; ++k) keep |= 1u << order[k];
            tile_all(keep, beta_used, true);
        }
    }
    // dominant shape?  (sampled first, counted only if the sample says so)
    if (n_pairs >= 1024 && n_cu > 0) {
        const size_t stride = (size_t)n_pairs / 512;
        uint32_t cand = 0;
        int votes = 0;
        auto shape = [](const PairPlan &pp) { return pp.lx() << 16 | (pp.ly & 0xffffu); };
        for (size_t k = 0; k < 512; ++k) { // Boyer-Moore majority vote over a sample
            const PairPlan &pp = all[k * stride];
            if (pp.cls == kClsEmpty) continue;
            if (votes == 0) {
                cand = shape(pp);
                votes = 1;
            } else
                votes += shape(pp) == cand ? 1 : -1;
        }
        if (votes > 0) {
            for (Worker &w : wk) w.votes = 0;
            agx_parallel_for(n_pairs, 16384, [&](int64_t lo, int64_t hi, int tid) {
                int64_t c = 0;
                for (int64_t p = lo; p < hi; ++p) c += all[(size_t)p].cls != kClsEmpty && shape(all[(size_t)p]) == cand;
                wk[(size_t)tid].votes = c;
            });
            int64_t count = 0;
            for (const Worker &w : wk) count += w.votes;
            if (count * 2 >= n_pairs) {
                const Tiling tl = choose_tiling_uniform(costs, slots, (int)(cand >> 16), (int)(cand & 0xffffu), count, 4 * n_cu);
                if (tl.cls >= 0 && !(matrix && kSwClasses[tl.cls] > 40))
                    agx_parallel_for(n_pairs, 16384, [&](int64_t lo, int64_t hi, int) {
                        for (int64_t p = lo; p < hi; ++p) {
                            PairPlan &pp = all[(size_t)p];
                            if (pp.cls != kClsEmpty && shape(pp) == cand) {
                                pp.cls = (uint8_t)tl.cls;
                                pp.G = (uint8_t)tl.G;
                            }
                        }
                    });
            }
        }
    }
    t_plan = now_ms();

    // ---- pass C: order = class, then lanes per group (wide first), then long rows first, then file
    // order; waves end up homogeneous and the longest waves of a launch are dispatched first.
    // Two stable counting passes (LSD): by ly descending, then by (class, G descending); pairs with an
    // empty side sort into a trailing bucket and are dropped.
    {
        std::vector<PairPlan> tmp;
        counting_sort(all, tmp, (size_t)longest_long + 2, [&](const PairPlan &pp) { return (size_t)(longest_long + 1 - pp.ly); });
        std::vector<PairPlan>().swap(all);
        const size_t n_buckets = (size_t)kSwNumClasses * 64;
        counting_sort(tmp, plan, n_buckets + 1, [&](const PairPlan &pp) {
            return pp.cls == kClsEmpty ? n_buckets : (size_t)pp.cls * 64 + (size_t)(64 - pp.G);
        });
        while (!plan.empty() && plan.back().cls == kClsEmpty) plan.pop_back();
    }
    t_sort = now_ms();
    } // !planned && !device_plan

    // what the rest of the function needs of a plan, whoever made it
    size_t n_groups = 0, n_waves_total = 0, groups_bytes = 0, waves_bytes = 0;
    const size_t img0 = packed ? (size_t)kSwPackedMaxShort / 4 + 1 : 0;
    size_t img_dw = img0;
    std::vector<SwWave> waves;
    std::vector<ClassLaunch> launches;
    int64_t padded = 0;
    double t_waves = t_sort, t_records = t_sort;
    std::vector<uint8_t> groups_host; // plan-only: no pinned memory without a device
    if (device_plan) {
        // ---- the buckets' extents from their counts (ascending bucket id = the sort order), then everything per pair on
        // the device, on the planning stream, beside the upload of the sequences
        {
            static std::mutex once_per_context;
            std::lock_guard<std::mutex> l(once_per_context);
            rc = agx_ctx_prepare_plan(ctx);
        }
        if (!rc) rc = dp.h_buckets.alloc(ctx, (size_t)kSwPlanBuckets * 5 * sizeof(uint32_t));
        if (rc) return rc;
        uint32_t *bt = (uint32_t *)dp.h_buckets.p;
        size_t entries = 0, n_waves = 0;
        uint32_t class_mask = 0;
        for (int k = 0; k < kSwPlanBuckets; ++k) {
            const size_t count = dp.hist[(size_t)k];
            const int G = 64 - (k & 63);
            const size_t ng = (count + slots - 1) / slots, per_wave = (size_t)(64 / G), nw = (ng + per_wave - 1) / per_wave;
            bt[5 * k + 0] = (uint32_t)entries;
            bt[5 * k + 1] = (uint32_t)count;
            bt[5 * k + 2] = (uint32_t)n_groups;
            bt[5 * k + 3] = (uint32_t)ng;
            bt[5 * k + 4] = (uint32_t)n_waves;
            entries += count;
            n_groups += ng;
            n_waves += nw;
            if (count) class_mask |= 1u << (k >> 6);
        }
        n_waves_total = n_waves;
        img_dw = dp.img_dw;
        groups_bytes = n_groups * sizeof(SwGroup2);
        waves_bytes = n_waves * sizeof(SwWave);
        ClassLaunch cl; // several classes: ONE launch, the class read per wave (sw_fill_pk2_any); else that class's own fill
        cl.C = (class_mask & (class_mask - 1)) ? 0 : kSwClasses[__builtin_ctz(class_mask)];
        cl.first_wave = 0;
        cl.n_waves = (uint32_t)n_waves;
        launches.assign(1, cl);
        rc = launch_device_plan(ctx, dp, b, (uint32_t)n_pairs, (uint32_t)entries, longest_long, slots, (uint32_t)img0, n_groups, n_waves);
        if (!rc) rc = h_flag.alloc(ctx, 2 * sizeof(uint32_t));
        if (rc) return rc;
        t_waves = t_records = now_ms();
    } else {
    // ---- pass D: every (class, G) bucket is regular, so waves, records and offsets need no scan but the
    // prefix sum of the image words.
    std::vector<Bucket> bk;
    for (size_t i = 0; i < plan.size();) {
        Bucket q;
        q.first = i;
        q.cls = plan[i].cls;
        q.G = plan[i].G;
        size_t lo = i, hi = plan.size(); // the run's end by bisection (equal keys are contiguous)
        while (lo + 1 < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            if (plan[mid].cls == q.cls && plan[mid].G == q.G) lo = mid;
            else hi = mid;
        }
        q.count = lo + 1 - i;
        i = lo + 1;
        bk.push_back(q);
    }
    size_t n_waves = 0;
    for (Bucket &q : bk) {
        q.group0 = n_groups;
        q.n_groups = (q.count + slots - 1) / slots;
        n_groups += q.n_groups;
        q.wave0 = n_waves;
        const size_t per_wave = (size_t)(64 / q.G);
        q.n_waves = (q.n_groups + per_wave - 1) / per_wave;
        n_waves += q.n_waves;
    }
    waves.assign(n_waves, SwWave{});
    for (const Bucket &q : bk) {
        if (launches.empty() || launches.back().C != kSwClasses[q.cls]) {
            ClassLaunch cl;
            cl.C = kSwClasses[q.cls];
            cl.first_wave = (uint32_t)q.wave0;
            launches.push_back(cl);
        }
        launches.back().n_waves += (uint32_t)q.n_waves;
        const size_t per_wave = (size_t)(64 / q.G);
        for (size_t wl = 0; wl < q.n_waves; ++wl) {
            SwWave w{};
            w.first_group = (uint32_t)(q.group0 + wl * per_wave);
            w.n_groups = (uint16_t)std::min(per_wave, q.n_groups - wl * per_wave);
            w.G = (uint16_t)q.G;
            w.steps = plan[q.first + wl * per_wave * slots].ly + (uint32_t)q.G - 1u; // rows are sorted long first
            w.reserved = (uint32_t)kSwClasses[q.cls];
            waves[q.wave0 + wl] = w;
            padded += (int64_t)w.steps * 64 * kSwClasses[q.cls] * slots;
        }
    }
    // dispatch order = longest waves first (a wave lasts steps x C; C is the class's): the buckets were
    // filled widest group first, which leaves narrow groups with long rows for the end of the launch
    static const bool sort_waves = [] {
        const char *e = agx_tune("AGX_SW_SORT_WAVES");
        return !(e && e[0] == '0');
    }();
    // A mixed batch of the biased packed kernel is ONE launch (sw_fill_pk2_any): all its waves in one list,
    // longest first across the classes.
    static const bool one_launch_ok = [] {
        const char *e = agx_tune("AGX_SW_ONE_LAUNCH");
        return !(e && e[0] == '0');
    }();
    if (family >= 2 && launches.size() > 1 && one_launch_ok) {
        if (sort_waves)
            std::stable_sort(waves.begin(), waves.end(), [](const SwWave &a, const SwWave &b) {
                return (uint64_t)a.steps * a.reserved > (uint64_t)b.steps * b.reserved;
            });
        ClassLaunch all_classes;
        all_classes.C = 0; // 0 = every class, read per wave
        all_classes.first_wave = 0;
        all_classes.n_waves = (uint32_t)waves.size();
        launches.assign(1, all_classes);
    } else if (sort_waves && !launches.empty())
        agx_pool_run((int)launches.size(), [&](int k) {
            const ClassLaunch &cl = launches[(size_t)k];
            std::stable_sort(waves.begin() + cl.first_wave, waves.begin() + cl.first_wave + cl.n_waves,
                             [](const SwWave &a, const SwWave &b) { return a.steps > b.steps; });
        });
    // image offsets: [x block][y block] per entry in plan order, after the zero block vacant slots point at
    std::vector<uint32_t> x_dw(plan.size()), y_dw(plan.size());
    img_dw = img0;
    {
        const int parts = (int)std::min<int64_t>(agx_host_threads(), std::max<int64_t>(1, (int64_t)plan.size() / 16384));
        std::vector<size_t> part_sum((size_t)parts + 1, 0);
        const size_t chunk = (plan.size() + parts - 1) / (size_t)parts;
        auto words = [&](const PairPlan &pp, size_t *xw) {
            *xw = ((size_t)pp.G * kSwClasses[pp.cls] + 3) / 4 + 1;
            return *xw + ((size_t)pp.ly + 3) / 4;
        };
        agx_pool_run(parts, [&](int t) {
            const size_t lo = std::min(plan.size(), (size_t)t * chunk), hi = std::min(plan.size(), lo + chunk);
            size_t s = 0, xw;
            for (size_t i = lo; i < hi; ++i) s += words(plan[i], &xw);
            part_sum[(size_t)t + 1] = s;
        });
        for (int t = 0; t < parts; ++t) part_sum[(size_t)t + 1] += part_sum[(size_t)t];
        img_dw = img0 + part_sum[(size_t)parts];
        if (img_dw > 0xffffffffull) {
            agx_set_error("packed image exceeds 16 GiB; split the batch");
            return AGX_E_LIMIT;
        }
        agx_pool_run(parts, [&](int t) {
            const size_t lo = std::min(plan.size(), (size_t)t * chunk), hi = std::min(plan.size(), lo + chunk);
            size_t at = img0 + part_sum[(size_t)t], xw;
            for (size_t i = lo; i < hi; ++i) {
                const size_t tot = words(plan[i], &xw);
                x_dw[i] = (uint32_t)at;
                y_dw[i] = (uint32_t)(at + xw);
                at += tot;
            }
        });
    }
    t_waves = now_ms();

    // ---- group records, written straight into pinned staging when there is a device
    groups_bytes = n_groups * (packed ? sizeof(SwGroup2) : sizeof(SwGroup));
    waves_bytes = waves.size() * sizeof(SwWave);
    n_waves_total = waves.size();
    void *groups_data = nullptr;
    if (ctx) {
        rc = h_groups.alloc(ctx, groups_bytes);
        if (!rc) rc = h_waves.alloc(ctx, waves_bytes);
        if (!rc) rc = h_flag.alloc(ctx, 2 * sizeof(uint32_t));
        if (rc) return rc;
        groups_data = h_groups.p;
        if (waves_bytes) memcpy(h_waves.p, waves.data(), waves_bytes);
    } else {
        groups_host.resize(groups_bytes);
        groups_data = groups_host.data();
    }
    for (const Bucket &q : bk)
        agx_parallel_for((int64_t)q.n_groups, 8192, [&](int64_t a, int64_t z, int) {
            for (int64_t g = a; g < z; ++g)
                for (int h = 0; h < slots; ++h) {
                    const size_t j = (size_t)g * slots + h;
                    uint32_t xd = 0, yd = 0, ll = 0, outi = (uint32_t)n_pairs; // vacant slot: zero block, spare score
                    if (j < q.count) {
                        const PairPlan &pp = plan[q.first + j];
                        xd = x_dw[q.first + j];
                        yd = y_dw[q.first + j];
                        ll = (uint32_t)pp.lxo | (pp.ly << 16);
                        outi = pp.pair;
                    }
                    if (packed) {
                        SwGroup2 &r = reinterpret_cast<SwGroup2 *>(groups_data)[q.group0 + (size_t)g];
                        r.x_dw[h] = xd;
                        r.y_dw[h] = yd;
                        r.lx_ly[h] = ll;
                        r.out[h] = outi;
                    } else
                        reinterpret_cast<SwGroup *>(groups_data)[q.group0 + (size_t)g] = SwGroup{xd, yd, ll, outi};
                }
        });
    t_records = now_ms();

    } // host-made plan
    b->launches = launches;
    b->info.n_pairs = n_pairs;
    b->info.cells = cells;
    b->info.padded_cells = padded;
    b->info.input_bytes = (int64_t)(img_dw * 4 + groups_bytes + waves_bytes);
    b->info.n_launches = (int32_t)launches.size();
    b->info.n_waves = (int32_t)n_waves_total;
    b->info.planned_on_device = device_plan ? 1 : 0;
    if (!ctx) { // planning only
        if (trace)
            fprintf(stderr, "[agx_sw_batch_create, plan only] %lld pairs: pass A %.2f ms | tiling %.2f, sort %.2f, waves %.2f, records %.2f\n",
                    (long long)n_pairs, t_pass_a - t_begin, t_plan - t_pass_a, t_sort - t_plan, t_waves - t_sort, t_records - t_waves);
        *out = b;
        b = nullptr;
        return AGX_OK;
    }

    // ---- device image: records up, image built and checked by the pack kernel, all on the copy stream
    if (uploader.joinable()) uploader.join();
    if (up_rc) {
        agx_set_error("%s", up_err);
        return up_rc;
    }
    const double t_joined = now_ms();
    rc = b->img.alloc(ctx, std::max<size_t>(img_dw, 4) * 4);
    if (!rc && !device_plan) rc = b->groups.alloc(ctx, groups_bytes); // (the device planner has written its own already)
    if (!rc && !device_plan) rc = b->waves.alloc(ctx, waves_bytes);
    if (!rc && matrix) rc = b->table.alloc(ctx, table.size() * sizeof(int16_t));
    if (!rc) rc = b->scores.alloc(ctx, ((size_t)n_pairs + 1) * sizeof(int32_t)); // +1: spare slot of vacant packed halves
    if (!rc) rc = b->out_stage.alloc(ctx, ((size_t)n_pairs + 1) * sizeof(int32_t));
    if (!rc && launches.size() > 1) rc = agx_ctx_prepare_fanout(ctx);
    if (rc) return rc;
    hipStream_t cs = ctx->copy;
    hipError_t e = hipSuccess;
    // A device-planned batch packs on the PLANNING stream, behind its records, once its sequences have arrived (an event
    // on the copy stream): the copy stream then carries nothing but uploads, and the next piece of a one-shot call
    // (agx_sw_score: two creator threads) uploads right behind this one instead of behind this one's pack kernel.
    hipStream_t ts = device_plan ? ctx->plan : cs;
    if (device_plan) {
        e = hipEventRecord(dp.uploaded, cs); // the uploader thread has queued everything (joined above)
        if (e == hipSuccess) e = hipStreamWaitEvent(ts, dp.uploaded, 0);
    }
    if (e == hipSuccess && !device_plan && groups_bytes) e = hipMemcpyAsync(b->groups.p, h_groups.p, groups_bytes, hipMemcpyHostToDevice, cs);
    if (e == hipSuccess && !device_plan && waves_bytes) e = hipMemcpyAsync(b->waves.p, h_waves.p, waves_bytes, hipMemcpyHostToDevice, cs);
    if (e == hipSuccess && matrix)
        e = hipMemcpyAsync(b->table.p, table.data(), table.size() * sizeof(int16_t), hipMemcpyHostToDevice, ts);
    // pairs with an empty side are never touched by a kernel: their score is this zero
    if (e == hipSuccess) e = hipMemsetAsync(b->scores.p, 0, b->scores.bytes, ts);
    if (e == hipSuccess && packed) e = hipMemsetAsync(b->img.p, 0, (size_t)kSwPackedMaxShort + 4, ts);
    uint32_t *flag = (uint32_t *)h_flag.p;
    flag[0] = 0;
    flag[1] = 0xffffffffu;
    if (e == hipSuccess && n_groups) {
        const char *dna_knob = agx_tune("AGX_SW_DNA");
        // the DNA-coded cell adds (mismatch + |gf|) and a table byte: both must be non-negative bytes
        const bool dna = family >= 2 && prm.delta < 128 && prm.hd >= prm.delta && !(dna_knob && dna_knob[0] == '0');
        const int pr = dna // the biased packed fill has a DNA-coded cell: its pack kernel decides per wavefront
                           ? agx_sw_pack_dna_launch((const uint8_t *)d_raw.p + kRawPad, (const uint64_t *)d_off.p, raw_base, b->groups.p, b->waves.p,
                                                    (uint32_t)n_waves_total, (uint32_t)n_pairs, (uint32_t *)b->img.p, (uint32_t *)d_flag.p,
                                                    n_cu, ts)
                           : agx_sw_pack_launch(matrix != nullptr, slots, (const uint8_t *)d_raw.p + kRawPad, (const uint64_t *)d_off.p, raw_base,
                                                b->groups.p, (uint32_t)n_groups, (uint32_t)n_pairs, (uint32_t *)b->img.p,
                                                (const uint8_t *)d_code.p, (uint32_t *)d_flag.p, n_cu, ts);
        if (pr) {
            agx_set_error("sw_pack launch failed: %s", hipGetErrorString(hipGetLastError()));
            return AGX_E_HIP;
        }
        e = hipMemcpyAsync(flag, d_flag.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ts);
    }
    if (e == hipSuccess && defer) { // the caller finishes later (finish_create): everything stays queued
        e = hipEventCreateWithFlags(&tmp->ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(tmp->ready, ts);
        if (e == hipSuccess) {
            tmp->tail = ts;
            tmp->device_plan = device_plan;
            b->pending = tmp.release();
            *out = b;
            b = nullptr;
            return AGX_OK;
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ts); // blocking by contract; the staging buffers are free again
    if (e != hipSuccess) {
        agx_set_error("agx_sw_batch_create: upload -> %s", hipGetErrorString(e));
        return AGX_E_HIP;
    }
    if (device_plan) b->info.padded_cells = (int64_t) * (const unsigned long long *)dp.h_padded.p; // (the planning stream ended before the pack kernel began)
    if (flag[0]) {
        if (matrix)
            agx_set_error("pair %u contains a byte outside the substitution matrix's alphabet", flag[1]);
        else
            agx_set_error("pair %u contains byte 0x00, which is reserved as the padding symbol", flag[1]);
        return AGX_E_SYMBOL;
    }
    if (trace)
        fprintf(stderr,
                "[agx_sw_batch_create] %lld pairs%s: pass A %.2f ms | tiling %.2f, sort %.2f, waves %.2f, records %.2f | waited %.2f ms more "
                "for the upload of %.1f MB | device pack + sync %.2f ms\n",
                (long long)n_pairs, device_plan ? " (planned on the device)" : "", t_pass_a - t_begin, t_plan - t_pass_a, t_sort - t_plan, t_waves - t_sort, t_records - t_waves,
                t_joined - t_records, raw_bytes / 1e6, now_ms() - t_joined);
    *out = b;
    b = nullptr;
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
    // a batch whose create was not finished: its fill waits for the pack kernel on the device, not on the host
    if (b->pending && b->pending->ready) AGX_HIP(hipStreamWaitEvent(b->ctx->stream, b->pending->ready, 0));
    SwParams prm = b->prm;
    if (b->bound) prm.n_out = (uint32_t)b->n_pairs; // the caller's array has no spare slot
    FanOut fan(b->ctx, (int)b->launches.size());
    rc = fan.begin();
    if (rc) return rc;
    int k = 0;
    // widest class first: its waves have the longest rows-times-columns chain, so they should not be the tail
    for (auto it = b->launches.rbegin(); it != b->launches.rend(); ++it) {
        const ClassLaunch &cl = *it;
        hipStream_t st = fan.stream(k++);
        const uint32_t *img = (const uint32_t *)b->img.p;
        const SwWave *wv = (const SwWave *)b->waves.p + cl.first_wave;
        int32_t *scores = b->bound ? b->bound : (int32_t *)b->scores.p;
        int r;
        if (b->matrix)
            r = agx_sw_mat_launch_class(cl.C, prm, img, (const SwGroup *)b->groups.p, wv, cl.n_waves, scores,
                                        (const int16_t *)b->table.p, st);
        else if (b->family == 3)
            r = cl.C == 0 ? agx_sw_i32d_launch_any(prm, img, (const SwGroup2 *)b->groups.p, wv, cl.n_waves, scores, st)
                          : agx_sw_i32d_launch_class(cl.C, prm, img, (const SwGroup2 *)b->groups.p, wv, cl.n_waves, scores, st);
        else if (b->family == 2 && cl.C == 0)
            r = agx_sw_pk2_launch_any(b->rising, prm, img, (const SwGroup2 *)b->groups.p, wv, cl.n_waves, scores, st);
        else if (b->family == 2)
            r = agx_sw_pk2_launch_class(cl.C, b->rising, prm, img, (const SwGroup2 *)b->groups.p, wv, cl.n_waves, scores, st);
        else if (b->family == 1)
            r = agx_sw_pk_launch_class(cl.C, prm, img, (const SwGroup2 *)b->groups.p, wv, cl.n_waves, scores, st);
        else
            r = (cl.C > 40 ? agx_sw_wide_launch_class : agx_sw_launch_class)(cl.C, prm, img, (const SwGroup *)b->groups.p, wv,
                                                                             cl.n_waves, scores, st);
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
    if (b->pending) { // (the pieces of agx_sw_score finish before they fetch; kept for safety)
        rc = finish_create(b);
        if (rc) return rc;
    }
    if (b->n_pairs == 0) {
        AGX_HIP(hipStreamSynchronize(b->ctx->stream));
        return AGX_OK;
    }
    const size_t bytes = (size_t)b->n_pairs * sizeof(int32_t);
    if (b->bound && scores == b->bound) { // the launches wrote them there themselves: nothing to copy
        AGX_HIP(hipStreamSynchronize(b->ctx->stream));
        return AGX_OK;
    }
    if (b->bound) { // bound elsewhere: the device array was not written by the last launch
        AGX_HIP(hipStreamSynchronize(b->ctx->stream));
        memcpy(scores, b->bound, bytes);
        return AGX_OK;
    }
    // one copy kernel right behind the last fill on the launch stream: straight into the caller's array when that is
    // page-locked (agx_host_alloc), else into pinned staging and a host copy from there
    int32_t *dst = agx_is_pinned_host(scores, bytes) ? scores : (int32_t *)b->out_stage.p;
    if (agx_copy_out_launch(b->scores.p, dst, bytes, b->ctx->stream)) {
        agx_set_error("agx_sw_batch_scores: copy kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        return AGX_E_HIP;
    }
    AGX_HIP(hipStreamSynchronize(b->ctx->stream));
    if (dst != scores) memcpy(scores, dst, bytes);
    return AGX_OK;
}

int agx_sw_batch_bind_scores(agx_sw_batch *b, int32_t *scores)
{
    if (!b) {
        agx_set_error("agx_sw_batch_bind_scores: null batch");
        return AGX_E_ARG;
    }
    if (!b->ctx) {
        agx_set_error("this batch was planned without a context (no device)");
        return AGX_E_NODEVICE;
    }
    if (scores && !agx_is_pinned_host(scores, (size_t)std::max<int64_t>(b->n_pairs, 1) * sizeof(int32_t))) {
        agx_set_error("agx_sw_batch_bind_scores: the array is not page-locked memory of agx_host_alloc (or too short for %lld scores)", (long long)b->n_pairs);
        return AGX_E_ARG;
    }
    const int rc = agx_bind(b->ctx);
    if (rc) return rc;
    AGX_HIP(hipStreamSynchronize(b->ctx->stream)); // launches in flight still write the old destination
    // only a batch whose records are in file order takes the binding (a sorted batch's waves would scatter 4-byte
    // writes over PCIe: measured slower than the copy kernel behind the fill); the call is a hint otherwise
    b->bound = (scores && b->file_order && b->n_pairs > 0 && !b->matrix) ? scores : nullptr;
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
    AGX_GUARD_BEGIN
    // A batch of 64 MB and more goes through in up to eight contiguous pieces of at least 32 MB of sequence: piece k + 1 is
    // uploaded, planned and packed (copy and planning streams) while piece k is being filled (launch stream), so the call
    // lasts about as long as the upload plus the last piece's fill instead of upload + fill (1 048 576 mixed pairs, 573 MB,
    // from page-locked memory: 25.2 -> 12.7 ms).  Scores are fetched at the end, piece by piece, into
    // the caller's array.
    int pieces = 1;
    if (ctx && n_pairs >= 2 * piece_min_pairs() && len) {
        std::vector<uint64_t> part((size_t)agx_host_threads(), 0);
        agx_parallel_for(n_pairs, 65536, [&](int64_t lo, int64_t hi, int t) {
            uint64_t s = 0;
            for (int64_t p = 2 * lo; p < 2 * hi; ++p) s += len[p];
            part[(size_t)t] = s;
        });
        uint64_t bytes = 0;
        for (uint64_t v : part) bytes += v;
        // at most eight pieces (more cost a pageable source's staging more than they hide), none below 32 MB / 32 768 pairs
        // (tools/piece_size_sweep.sh: config 4's 72 MB shard 2.96 -> 2.31 ms in two pieces, 2.73 in four)
        pieces = (int)std::min<uint64_t>({(uint64_t)8, bytes / piece_bytes(), (uint64_t)(n_pairs / piece_min_pairs())});
        if (pieces < 2) pieces = 1;
    }
    if (pieces == 1) {
        agx_sw_batch *b = nullptr;
        int rc = agx_sw_batch_create(ctx, bases, off, len, n_pairs, &b);
        if (rc) return rc;
        rc = agx_sw_batch_launch(b);
        if (!rc) rc = agx_sw_batch_scores(b, scores);
        agx_sw_batch_destroy(b); // its buffers return to the context's pools for the next call
        return rc;
    }
    std::vector<agx_sw_batch *> bs((size_t)pieces, nullptr);
    struct Cleanup {
        std::vector<agx_sw_batch *> &v;
        ~Cleanup()
        {
            for (agx_sw_batch *b : v) agx_sw_batch_destroy(b);
        }
    } cleanup{bs};
    auto cut = [&](int k) { return n_pairs * k / pieces; };
    // Every piece is created WITHOUT its closing wait and launched behind an event: the copy stream carries the pieces'
    // uploads back to back, the planning stream their planning and pack kernels, the launch stream their fills, and
    // the host is ahead of all three.  The verdicts of the symbol checks are collected afterwards, piece by piece (the
    // first failing piece holds the smallest offending pair).
    auto renumber = [&](int rc, int64_t lo) { // the messages name pair numbers: of the whole batch, not of the piece
        if (rc == AGX_E_SYMBOL || rc == AGX_E_LIMIT) {
            unsigned long long p = 0;
            char rest[400] = "";
            if (sscanf(agx_last_error(), "pair %llu%399[^\n]", &p, rest) >= 1) agx_set_error("pair %llu%s", p + (unsigned long long)lo, rest);
        }
        return rc;
    };
    const bool trace = agx_tune("AGX_TRACE_CREATE") != nullptr;
    const double t_begin = now_ms();
    for (int k = 0; k < pieces; ++k) {
        const int64_t lo = cut(k), hi = cut(k + 1);
        const double ta = now_ms();
        int rc = renumber(create_batch(ctx, nullptr, nullptr, bases, off + 2 * lo, len + 2 * lo, hi - lo, &bs[(size_t)k], true), lo);
        if (!rc) rc = agx_sw_batch_launch(bs[(size_t)k]);
        if (rc) return rc;
        if (trace) fprintf(stderr, "[agx_sw_score] piece %d of %d queued in %.2f ms (at %.2f)\n", k, pieces, now_ms() - ta, now_ms() - t_begin);
    }
    for (int k = 0; k < pieces; ++k) {
        const int rc = renumber(finish_create(bs[(size_t)k]), cut(k));
        if (rc) return rc;
        if (trace) fprintf(stderr, "[agx_sw_score] piece %d finished at %.2f ms\n", k, now_ms() - t_begin);
    }
    for (int k = 0; k < pieces; ++k) {
        const int rc = agx_sw_batch_scores(bs[(size_t)k], scores + cut(k));
        if (rc) return rc;
        agx_sw_batch_destroy(bs[(size_t)k]);
        bs[(size_t)k] = nullptr;
    }
    return AGX_OK;
    AGX_GUARD_END("agx_sw_score")
}

int agx_sw_shard_cuts(const uint32_t *len, int64_t n_pairs, int n_shards, int64_t *cut)
{
    if (n_pairs < 0 || n_shards < 1 || !cut || (n_pairs > 0 && !len)) {
        agx_set_error("agx_sw_shard_cuts: bad arguments");
        return AGX_E_ARG;
    }
    // contiguous shards balanced by cells
    for (int d = 0; d <= n_shards; ++d) cut[d] = n_pairs;
    cut[0] = 0;
    double total = 0;
    for (int64_t p = 0; p < n_pairs; ++p) total += (double)len[2 * p] * len[2 * p + 1] + 1.0;
    double acc = 0;
    int d = 1;
    for (int64_t p = 0; p < n_pairs && d < n_shards; ++p) {
        acc += (double)len[2 * p] * len[2 * p + 1] + 1.0;
        while (d < n_shards && acc >= total * d / n_shards) cut[d++] = p + 1;
    }
    return AGX_OK;
}

int agx_sw_score_devices(const int *devices, int n_devices, const uint8_t *bases, const uint64_t *off, const uint32_t *len,
                         int64_t n_pairs, int32_t *scores)
{
    AGX_GUARD_BEGIN
    const int avail = agx_device_count();
    if (avail <= 0) {
        agx_set_error("no HIP device is visible (this library has no CPU fallback)");
        return AGX_E_NODEVICE;
    }
    if (!devices || n_devices < 1 || n_devices > 1024 || n_pairs < 0 || (n_pairs > 0 && (!off || !len || !scores))) {
        agx_set_error("agx_sw_score_devices: bad arguments");
        return AGX_E_ARG;
    }
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= avail) {
            agx_set_error("agx_sw_score_devices: device %d out of range [0,%d)", devices[k], avail);
            return AGX_E_NODEVICE;
        }
    // results land in disjoint slices of the caller's array: no exchange step
    std::vector<int64_t> cut((size_t)n_devices + 1);
    int rc = agx_sw_shard_cuts(len, n_pairs, n_devices, cut.data());
    if (rc) return rc;
    std::vector<int> rcs((size_t)n_devices, AGX_OK), slot((size_t)n_devices, 0);
    for (int k = 0; k < n_devices; ++k) // shards sharing a device get contexts of their own
        for (int j = 0; j < k; ++j) slot[(size_t)k] += devices[j] == devices[k];
    std::vector<std::string> errs((size_t)n_devices);
    auto shard = [&](int k) {
        const int64_t lo = cut[(size_t)k], hi = cut[(size_t)k + 1];
        if (hi <= lo) return;
        int r;
        try {
            agx_ctx *c = nullptr;
            std::mutex *busy = nullptr;
            r = agx_shared_ctx(devices[k], slot[(size_t)k], &c, &busy); // created once per process: pools stay warm
            if (!r) {
                std::lock_guard<std::mutex> turn(*busy); // concurrent callers take turns on this (device, slot)
                r = agx_sw_score(c, bases, off + 2 * lo, len + 2 * lo, hi - lo, scores + lo);
            }
        } catch (const std::exception &ex) {
            agx_set_error("shard %d: %s", k, ex.what());
            r = AGX_E_NOMEM;
        }
        if (r) errs[(size_t)k] = agx_last_error();
        rcs[(size_t)k] = r;
    };
    agx_fan_out(n_devices, shard);
    for (int k = 0; k < n_devices; ++k)
        if (rcs[(size_t)k]) {
            agx_set_error("device %d: %s", devices[k], errs[(size_t)k].c_str());
            return rcs[(size_t)k];
        }
    return AGX_OK;
    AGX_GUARD_END("agx_sw_score_devices")
}

int agx_sw_score_multi(int n_devices, const uint8_t *bases, const uint64_t *off, const uint32_t *len, int64_t n_pairs,
                       int32_t *scores)
{
    const int avail = agx_device_count();
    if (avail <= 0) {
        agx_set_error("no HIP device is visible (this library has no CPU fallback)");
        return AGX_E_NODEVICE;
    }
    if (n_devices <= 0 || n_devices > avail) n_devices = avail;
    n_devices = std::min(n_devices, 1024);
    int devs[1024];
    for (int k = 0; k < n_devices; ++k) devs[k] = k;
    return agx_sw_score_devices(devs, n_devices, bases, off, len, n_pairs, scores);
}

} // extern "C"
