// Host side of the Smith-Waterman path: validation, lane-tiling choice, packing into the
// device image, launches, multi-device sharding (include/agx.h, "Smith-Waterman" section).
#include "agx_sw.h"

#include <algorithm>
#include <string>
#include <thread>

#include "agx_internal.h"

namespace {

struct Tiling {
    int cls; // index into kSwClasses
    int G;
};

// Lane time a pair costs under tiling (class ci, G): steps * C * 64 / floor(64 / G) padded
// cells (the lanes of a wave that cannot host another group are charged to the pair), weighted
// by the measured per-cell cost of the class.
inline double tiling_cost(int ly, int ci, int G)
{
    return (double)(ly + G - 1) * kSwClasses[ci] * (64.0 / (double)(64 / G)) * kSwClassCost[ci];
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

Tiling choose_tiling(int lx, int ly)
{
    Tiling best{-1, 0};
    double best_cost = 0;
    for (int ci = 0; ci < kSwNumClasses; ++ci) {
        const int C = kSwClasses[ci];
        const int G = (lx + C - 1) / C;
        if (G > 64) continue;
        if (C > max_cols_per_lane() && best.cls >= 0) continue;
        if (force_cols_per_lane() && C != force_cols_per_lane()) continue;
        const double c = tiling_cost(ly, ci, G);
        if (best.cls < 0 || c < best_cost || (c == best_cost && C > kSwClasses[best.cls])) {
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
    int64_t n_pairs = 0;
    DevBuf img, groups, waves, scores;
    std::vector<ClassLaunch> launches;
    agx_sw_info info{};
};

extern "C" {

void agx_sw_batch_destroy(agx_sw_batch *b)
{
    if (!b) return;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    b->img.release();
    b->groups.release();
    b->waves.release();
    b->scores.release();
    delete b;
}

int agx_sw_batch_create(agx_ctx *ctx, const uint8_t *bases, const uint64_t *off, const uint32_t *len, int64_t n_pairs,
                        agx_sw_batch **out)
{
    if (!out) {
        agx_set_error("agx_sw_batch_create: out is NULL");
        return AGX_E_ARG;
    }
    *out = nullptr;
    int rc = agx_bind(ctx);
    if (rc) return rc;
    if (n_pairs < 0 || (n_pairs > 0 && (!off || !len))) {
        agx_set_error("agx_sw_batch_create: bad arguments (n_pairs=%lld)", (long long)n_pairs);
        return AGX_E_ARG;
    }
    if (n_pairs > 0x7fffffffLL / 2) {
        agx_set_error("agx_sw_batch_create: more than 2^30 pairs in one batch");
        return AGX_E_LIMIT;
    }

    // ---- plan every pair
    std::vector<PairPlan> plan;
    plan.reserve((size_t)n_pairs);
    int64_t cells = 0;
    for (int64_t p = 0; p < n_pairs; ++p) {
        const uint32_t la = len[2 * p], lb = len[2 * p + 1];
        if ((la || lb) && !bases) {
            agx_set_error("agx_sw_batch_create: bases is NULL");
            return AGX_E_ARG;
        }
        cells += (int64_t)la * lb;
        if (la == 0 || lb == 0) continue; // no interior cell: score stays 0
        const bool second_short = lb < la; // ties keep file order (antidiagonalSmithWaterman.c:229-244)
        const uint32_t lx = second_short ? lb : la, ly = second_short ? la : lb;
        if (lx > AGX_SW_MAX_SHORT_LEN || ly > 0xffffu) {
            agx_set_error("pair %lld: lengths %u x %u exceed the supported %d x 65535", (long long)p, la, lb,
                          AGX_SW_MAX_SHORT_LEN);
            return AGX_E_LIMIT;
        }
        if (memchr(bases + off[2 * p], 0, la) || memchr(bases + off[2 * p + 1], 0, lb)) {
            agx_set_error("pair %lld contains byte 0x00, which is reserved as the padding symbol", (long long)p);
            return AGX_E_SYMBOL;
        }
        const Tiling tl = choose_tiling((int)lx, (int)ly);
        if (tl.cls < 0) {
            agx_set_error("pair %lld: no lane tiling fits %u columns", (long long)p, lx);
            return AGX_E_LIMIT;
        }
        PairPlan pp{};
        pp.pair = (uint32_t)p;
        pp.lx = (uint16_t)lx;
        pp.ly = ly;
        pp.cls = (uint8_t)tl.cls;
        pp.G = (uint8_t)tl.G;
        pp.x_is_second = second_short ? 1 : 0;
        plan.push_back(pp);
    }
    // class, then lanes per group, then long rows first: waves end up homogeneous and the
    // longest waves of a launch are dispatched first.
    std::sort(plan.begin(), plan.end(), [](const PairPlan &a, const PairPlan &b) {
        if (a.cls != b.cls) return a.cls < b.cls;
        if (a.G != b.G) return a.G > b.G;
        if (a.ly != b.ly) return a.ly > b.ly;
        return a.pair < b.pair;
    });

    // ---- form waves and pack the image
    std::vector<SwGroup> groups(plan.size());
    std::vector<SwWave> waves;
    std::vector<uint32_t> img;
    img.reserve((size_t)(cells ? 1024 : 16));
    std::vector<ClassLaunch> launches;
    int64_t padded = 0;
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
            w.first_group = (uint32_t)i;
            w.G = (uint16_t)G;
            int n = 0, max_ly = 0;
            while (i < plan.size() && plan[i].cls == cls && plan[i].G == G && n < per_wave) {
                const PairPlan &pp = plan[i];
                const uint64_t ox = off[2 * (uint64_t)pp.pair + pp.x_is_second];
                const uint64_t oy = off[2 * (uint64_t)pp.pair + (pp.x_is_second ^ 1)];
                SwGroup g;
                const size_t xdw = (size_t)G * cl.C / 4, ydw = ((size_t)pp.ly + 3) / 4;
                if (img.size() + xdw + ydw > 0xffffffffull) {
                    agx_set_error("packed image exceeds 16 GiB; split the batch");
                    return AGX_E_LIMIT;
                }
                g.x_dw = (uint32_t)img.size();
                img.resize(img.size() + xdw, 0u);
                memcpy((uint8_t *)&img[g.x_dw], bases + ox, pp.lx);
                g.y_dw = (uint32_t)img.size();
                img.resize(img.size() + ydw, 0u);
                memcpy((uint8_t *)&img[g.y_dw], bases + oy, pp.ly);
                g.lx_ly = (uint32_t)pp.lx | (pp.ly << 16);
                g.out = pp.pair;
                groups[i] = g;
                max_ly = std::max(max_ly, (int)pp.ly);
                ++n;
                ++i;
            }
            w.n_groups = (uint16_t)n;
            w.steps = (uint32_t)(max_ly + G - 1);
            padded += (int64_t)w.steps * 64 * cl.C;
            waves.push_back(w);
        }
        cl.n_waves = (uint32_t)waves.size() - cl.first_wave;
        launches.push_back(cl);
    }

    // ---- device image
    agx_sw_batch *b = new agx_sw_batch();
    b->ctx = ctx;
    b->n_pairs = n_pairs;
    b->launches = launches;
    b->info.n_pairs = n_pairs;
    b->info.cells = cells;
    b->info.padded_cells = padded;
    b->info.input_bytes = (int64_t)(img.size() * 4 + groups.size() * sizeof(SwGroup) + waves.size() * sizeof(SwWave));
    b->info.n_launches = (int32_t)launches.size();
    b->info.n_waves = (int32_t)waves.size();
    rc = b->img.alloc(img.size() * 4);
    if (!rc) rc = b->groups.alloc(groups.size() * sizeof(SwGroup));
    if (!rc) rc = b->waves.alloc(waves.size() * sizeof(SwWave));
    if (!rc) rc = b->scores.alloc((size_t)n_pairs * sizeof(int32_t));
    if (rc) {
        agx_sw_batch_destroy(b);
        return rc;
    }
    hipError_t e = hipSuccess;
    if (!img.empty()) e = hipMemcpy(b->img.p, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && !groups.empty())
        e = hipMemcpy(b->groups.p, groups.data(), groups.size() * sizeof(SwGroup), hipMemcpyHostToDevice);
    if (e == hipSuccess && !waves.empty())
        e = hipMemcpy(b->waves.p, waves.data(), waves.size() * sizeof(SwWave), hipMemcpyHostToDevice);
    // pairs with an empty side are never touched by a kernel: their score is this zero
    if (e == hipSuccess) e = hipMemset(b->scores.p, 0, b->scores.bytes);
    if (e != hipSuccess) {
        agx_set_error("agx_sw_batch_create: upload -> %s", hipGetErrorString(e));
        agx_sw_batch_destroy(b);
        return AGX_E_HIP;
    }
    *out = b;
    return AGX_OK;
}

int agx_sw_batch_launch(agx_sw_batch *b)
{
    if (!b) {
        agx_set_error("agx_sw_batch_launch: null batch");
        return AGX_E_ARG;
    }
    int rc = agx_bind(b->ctx);
    if (rc) return rc;
    for (const ClassLaunch &cl : b->launches) {
        const int r = agx_sw_launch_class(cl.C, (const uint32_t *)b->img.p, (const SwGroup *)b->groups.p,
                                          (const SwWave *)b->waves.p + cl.first_wave, cl.n_waves,
                                          (int32_t *)b->scores.p, b->ctx->stream);
        if (r) {
            agx_set_error("sw_fill<%d> launch failed: %s", cl.C, hipGetErrorString(hipGetLastError()));
            return AGX_E_HIP;
        }
    }
    return AGX_OK;
}

int agx_sw_batch_scores(agx_sw_batch *b, int32_t *scores)
{
    if (!b || (!scores && b->n_pairs)) {
        agx_set_error("agx_sw_batch_scores: null argument");
        return AGX_E_ARG;
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
    rc = agx_sw_batch_launch(b);
    if (!rc) rc = agx_sw_batch_scores(b, scores);
    agx_sw_batch_destroy(b);
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
    if (n_devices <= 0 || n_devices > avail) n_devices = avail;
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
            int rc = agx_ctx_create(d, &c);
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
