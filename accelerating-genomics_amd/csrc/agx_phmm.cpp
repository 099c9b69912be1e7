// Host side of the PairHMM path: validation, lane-tiling choice, packing, launches, the final
// log10 (host libm, as the reference does), multi-device sharding, and the pairHMM() seam
// (include/agx.h, "PairHMM" section).
#include "agx_phmm.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

#include "agx_internal.h"
#include "agx_parallel.h"

namespace {

// LDS bytes a wave may spend on several read tables (8 waves/CU fit 160 KiB); AGX_PHMM_TAB_BUDGET overrides (experiments)
size_t tab_budget()
{
    static const size_t v = [] {
        const char *e = agx_tune("AGX_PHMM_TAB_BUDGET");
        const long n = e ? atol(e) : 0;
        return n > 0 ? (size_t)n : (size_t)20 * 1024;
    }();
    return v;
}
constexpr uint32_t kHapSlack = 44;       // zero bytes after every haplotype: any tiling reads in bounds

struct Plan {
    uint32_t out;
    uint32_t read, hap;
    uint32_t R, H;
    uint32_t th; // the haplotype length the pair is tiled for (packed kernel: the longer one of its lane group)
    uint8_t cls;
    uint8_t G;
};

struct ClassLaunch {
    int C = 0;
    bool all_g16 = true; // every wave of the class has groups of exactly 16 lanes
    uint32_t first_wave = 0, n_waves = 0;
    size_t lds = 0;        // dynamic LDS of the fill in the batch's precision
    size_t lds_rescue = 0; // same tables with double rows (F32 rescue pass)
};

// Tuning knob for experiments (not part of the ABI): AGX_PHMM_MAX_C caps the columns per lane.
int max_cols_per_lane()
{
    static const int v = [] {
        const char *e = agx_tune("AGX_PHMM_MAX_C");
        const int n = e ? atoi(e) : 0;
        return n >= 4 ? n : kPhClasses[kPhNumClasses - 1];
    }();
    return v;
}

// AGX_PHMM_FORCE_C pins the class (calibration runs only); a class table without that width
// ignores it (e.g. the double rescue plan of a packed float batch forced to an odd width).
int force_cols_per_lane_raw()
{
    static const int v = [] {
        const char *e = agx_tune("AGX_PHMM_FORCE_C");
        return e ? atoi(e) : 0;
    }();
    return v;
}

constexpr int stripe_cols() { return AGX_PH_STRIPE_COLS; }

// kind 0..2 = rows of kPhClassCost over kPhClasses; kind 3 = the packed float kernel's own class table
struct ClassTable {
    const int *C;
    const double *cost;
    int n;
};
ClassTable class_table(int kind)
{
    if (kind == 3) return ClassTable{kPhPkClasses, kPhPkClassCost, kPhPkNumClasses};
    return ClassTable{kPhClasses, kPhClassCost[kind], kPhNumClasses};
}

int force_cols_per_lane(const ClassTable &ct)
{
    const int f = force_cols_per_lane_raw();
    for (int ci = 0; f && ci < ct.n; ++ci)
        if (ct.C[ci] == f && ct.cost[ci] != 0) return f;
    return 0;
}

// AGX_PHMM_TAIL_BETA overrides the planner's tail term (experiments); negative = unset
double tail_beta_override()
{
    static const double v = [] {
        const char *e = agx_tune("AGX_PHMM_TAIL_BETA");
        return e ? atof(e) : -1.0;
    }();
    return v;
}

// allowed: bit ci set = class ci may be used; *cost_out: the lane time estimate of the choice;
// beta: lanes' worth of extra weight on a wave's own duration (steps * C), see make_plan
void choose_tiling(int precision, uint32_t R, uint32_t H, uint64_t allowed, uint8_t *cls, uint8_t *G_out, double *cost_out,
                   double beta)
{
    const ClassTable ct = class_table(precision);
    int best = -1, bestG = 0;
    double best_cost = 0;
    for (int ci = 0; ci < ct.n; ++ci) {
        if (!((allowed >> ci) & 1u)) continue;
        const int C = ct.C[ci];
        const int G = (int)((H + C - 1) / C);
        if (G > 64) continue;
        if (C > max_cols_per_lane() && best >= 0) continue;
        if (force_cols_per_lane(ct) && C != force_cols_per_lane(ct)) continue;
        const double wgt = ct.cost[ci];
        if (wgt == 0) continue; // class not built for this arithmetic
        const double cost = (double)(R + G - 1) * C * ((64.0 / (double)(64 / G)) * wgt + beta);
        if (best < 0 || cost < best_cost) {
            best = ci;
            bestG = G;
            best_cost = cost;
        }
    }
    *cls = (uint8_t)(best < 0 ? 255 : best);
    *G_out = (uint8_t)bestG;
    if (cost_out) *cost_out = best_cost;
}

// AGX_PHMM_MAX_CLASSES: upper bound on the kernel classes a mixed batch may spread over (default 6)
int max_classes()
{
    static const int v = [] {
        const char *e = agx_tune("AGX_PHMM_MAX_CLASSES");
        const int n = e ? atoi(e) : 0;
        return n > 0 ? n : 6;
    }();
    return v;
}

// Uniform batches (most pairs share one (R, H) shape): the launch lasts ceil(waves / SIMDs)
// wave-times, so the tiling of that shape is chosen with that quantisation.
// pair_reads: the plan will pair reads into trains (packed float fill): a wave then lasts 2 (R + 1) + G - 2 steps and there are
// half as many -- the quantisation is that of the paired waves (H = 270: 9 lanes x 30 columns are 2341 paired waves, three
// rounds of which the last is a seventh full, 0.363 ms; 16 x 17 are 4096, four full rounds, 0.286 ms)
void choose_tiling_uniform(int precision, uint32_t R, uint32_t H, int64_t count, int n_simd, bool pair_reads, uint8_t *cls, uint8_t *G_out)
{
    const ClassTable ct = class_table(precision);
    int best = -1, bestG = 0;
    double best_cost = 0;
    for (int ci = 0; ci < ct.n; ++ci) {
        const int C = ct.C[ci];
        const int G = (int)((H + C - 1) / C);
        if (G > 64) continue;
        if (C > max_cols_per_lane() && best >= 0) continue;
        if (force_cols_per_lane(ct) && C != force_cols_per_lane(ct)) continue;
        const double wgt = ct.cost[ci];
        if (wgt == 0) continue;
        const int64_t per_wave = 64 / G;
        int64_t waves = (count + per_wave - 1) / per_wave;
        uint32_t steps = R + (uint32_t)G - 1u;
        if (pair_reads && C <= 30) { // (widths 31 and 32 do not pair: see partner_of)
            waves = (waves + 1) / 2;
            steps = 2u * (R + 1u) + (uint32_t)G - 2u;
        }
        const int64_t rounds = (waves + n_simd - 1) / n_simd;
        const double cost = (double)rounds * steps * C * wgt;
        if (best < 0 || cost < best_cost) {
            best = ci;
            bestG = G;
            best_cost = cost;
        }
    }
    *cls = (uint8_t)(best < 0 ? 255 : best);
    *G_out = (uint8_t)bestG;
}

// 256-entry quality LUT exactly as partition_read() computes it (antidiagsPairHMM.c:104-107)
// mis: the mismatch prior per quality byte -- Qr (reference) or Qr/3 (AGX_PHMM_GATK_PRIOR)
void build_lut(bool gatk_prior, double *d, float *f, double *mis_d, float *mis_f)
{
    for (int c = 0; c < 256; ++c) {
        // the reference holds the quality bytes in plain `char`, signed on x86-64 (:100-107): a byte of 0x80 and above
        // is a negative number there (200 -> -56), and so it is here
        d[c] = pow(10.0, -((c < 128 ? c : c - 256) - 33.0) * 0.1);
        f[c] = (float)d[c];
        mis_d[c] = gatk_prior ? d[c] / 3.0 : d[c];
        mis_f[c] = gatk_prior ? f[c] / 3.0f : f[c];
    }
}

// ---- a plan: lane tilings, waves and records for one kernel family
struct PlanOut {
    std::vector<PhGroup> groups1;
    std::vector<PhGroup2> groups2;
    std::vector<PhTab> tabs;
    std::vector<PhWave> waves;
    std::vector<ClassLaunch> launches;
    int64_t padded = 0;
    // where the padded cells go (AGX_TRACE_CREATE): [0] useful R x H, [1] columns beyond H (vacant halves included), [2] the G - 1
    // steps of skew, [3] steps beyond a group's own R + G - 1 (the wave steps its longest read), [4] lanes without a group
    int64_t waste[5] = {0, 0, 0, 0, 0};
    bool trains = false; // packed plan: groups may carry a second read, every table has two PhTab entries (agx_phmm.h)
    bool file_order = false; // one (R, H) shape: the records are in output order (agx_phmm_batch_bind_results)
};

// What a plan is made from: the pairs with work in output order (region, read, haplotype) and where the image holds
// every read and haplotype.  A packed float batch keeps it: its double rescue plan is made when a fill first counts a
// pair below the float range (config 3: never), not at creation.
struct PlanSeed {
    std::vector<Plan> gen0;
    std::vector<uint32_t> read_dw, hap_dw;
    int64_t n_pairs = 0;
    int n_cu = 256;
    bool gatk_prior = false;
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Cuts [0, n) into at most `parts` pieces of about equal size whose inner boundaries satisfy is_cut(i) (i starts a new
// run): every O(pairs) pass of the planner that works run by run is threaded over such pieces.
template <typename F>
std::vector<size_t> cut_at_runs(size_t n, int parts, F is_cut)
{
    std::vector<size_t> cut{0};
    for (int k = 1; k < parts; ++k) {
        size_t i = std::max(cut.back(), n * (size_t)k / (size_t)parts);
        while (i < n && (i == 0 || !is_cut(i))) ++i;
        if (i > cut.back() && i < n) cut.push_back(i);
    }
    cut.push_back(n);
    return cut;
}

// kind = row of kPhClassCost (0 f64, 1 f64 FMA, 2 f32, 3 packed f32 FMA); slots = pairs per group; lut_rows: the read
// tables have the rows of agx_phmm_lut_kernel.hip; `gen` (a copy of seed.gen0, or seed.gen0 itself when nobody needs
// it afterwards) is consumed.
// trains: 0 = read trains where they pay (packed plans of enough waves), 1 = never, 2 = wherever two reads can share a group
int make_plan(const PlanSeed &seed, std::vector<Plan> gen, int kind, int slots, bool rows_f64, bool lut_rows, bool trace, int trains, PlanOut &po)
{
    const double tm0 = now_ms();
    double tm1 = tm0, tm2 = tm0, tm3 = tm0, tm4 = tm0;
    const ClassTable ct = class_table(kind);
    const bool gatk_prior = seed.gatk_prior;
    const int n_cu = seed.n_cu;
    const std::vector<uint32_t> &read_dw = seed.read_dw, &hap_dw = seed.hap_dw;
    const uint32_t vacant_out = (uint32_t)seed.n_pairs;
    auto tab_bytes = [&](bool f64, uint32_t rows) {
        return slots == 2 ? ph_pk_tab_bytes(rows) : lut_rows ? ph_lut_tab_bytes(rows) : ph_tab_bytes(f64, gatk_prior, rows);
    };
    const int64_t n = (int64_t)gen.size();
    auto key_of = [](const Plan &p) { return p.R << 16 | p.th; };
    // Two haplotypes share a lane group in the packed kernel: within every read's run, order the
    // haplotypes by length and tile neighbours for the longer of the two, so that partners get the
    // same class (mixed-length regions would otherwise leave most second slots vacant).
    // (fixed-length input -- BASELINE configs 3 and 5 -- needs neither this pairing nor the sorts below: every
    // key is equal, the enumeration order already is the order they would produce)
    const int nthr = agx_host_threads();
    struct Range {
        uint32_t r_min = 0xffffffffu, r_max = 0, h_min = 0xffffffffu, h_max = 0;
        char pad[48]; // one cache line per part
    };
    std::vector<Range> ranges((size_t)nthr);
    agx_parallel_for(n, 16384, [&](int64_t lo, int64_t hi, int t) {
        Range r;
        for (int64_t k = lo; k < hi; ++k) {
            Plan &p = gen[(size_t)k];
            p.th = p.H;
            r.r_min = std::min(r.r_min, p.R), r.r_max = std::max(r.r_max, p.R);
            r.h_min = std::min(r.h_min, p.H), r.h_max = std::max(r.h_max, p.H);
        }
        ranges[(size_t)t] = r;
    });
    Range all;
    for (const Range &r : ranges) {
        all.r_min = std::min(all.r_min, r.r_min), all.r_max = std::max(all.r_max, r.r_max);
        all.h_min = std::min(all.h_min, r.h_min), all.h_max = std::max(all.h_max, r.h_max);
    }
    const bool one_shape = n > 0 && all.r_min == all.r_max && all.h_min == all.h_max;
    const int run_parts = (int)std::min<int64_t>(nthr, std::max<int64_t>(1, n / 8192));
    if (slots == 2 && !one_shape) {
        const std::vector<size_t> cut = cut_at_runs((size_t)n, run_parts, [&](size_t i) { return gen[i].read != gen[i - 1].read; });
        agx_pool_run((int)cut.size() - 1, [&](int t) {
            size_t a = cut[(size_t)t];
            const size_t end = cut[(size_t)t + 1];
            while (a < end) {
                size_t z = a;
                while (z < end && gen[z].read == gen[a].read) ++z;
                std::stable_sort(gen.begin() + (ptrdiff_t)a, gen.begin() + (ptrdiff_t)z, [](const Plan &x, const Plan &y) { return x.H > y.H; });
                for (size_t k = a; k < z; ++k) gen[k].th = gen[a + ((k - a) & ~(size_t)1)].H; // the pair's longer one
                a = z;
            }
        });
    }
    // Tiling per (R, H) shape.  The per-class cost curve is flat over many widths, and every class
    // is its own launch: a mixed batch first chooses freely, then keeps the few classes that carry
    // most of the work and re-tiles the rest among them (a shape no kept class can span keeps its own).
    // The shapes of a batch live in a window of read and haplotype lengths: they are counted in a dense table over
    // that window (threads add to it directly) when it has at most 2^20 cells, else through a hash map.
    const uint64_t win_r = n ? (uint64_t)all.r_max - all.r_min + 1 : 0, win_h = n ? (uint64_t)all.h_max - all.h_min + 1 : 0;
    // (... and no more than eight cells per pair beyond 65 536: a small batch with a wide window goes through the map)
    const bool dense = win_r * win_h <= std::min<uint64_t>((uint64_t)1 << 20, std::max<uint64_t>(65536, 8 * (uint64_t)n));
    std::unordered_map<uint32_t, uint32_t> sparse_slot; // key -> slot (batches whose window is too wide)
    auto slot_of = [&](uint32_t key) -> size_t {
        if (dense) return (size_t)((key >> 16) - all.r_min) * (size_t)win_h + ((key & 0xffffu) - all.h_min);
        return sparse_slot.find(key)->second;
    };
    std::vector<uint32_t> slot_count;        // pairs per slot
    std::vector<uint16_t> slot_tile;         // cls << 8 | G
    std::vector<uint32_t> shape_key, shape_slot; // the distinct shapes
    if (one_shape) {
        slot_count.assign(1, (uint32_t)n);
        shape_key.push_back(key_of(gen[0]));
        shape_slot.push_back(0);
    } else if (dense) {
        slot_count.assign((size_t)(win_r * win_h), 0);
        agx_parallel_for(n, 16384, [&](int64_t lo, int64_t hi, int) {
            uint32_t last = 0, run = 0; // consecutive pairs often share their shape
            for (int64_t k = lo; k < hi; ++k) {
                const uint32_t key = key_of(gen[(size_t)k]);
                if (run && key == last) {
                    ++run;
                    continue;
                }
                if (run) __atomic_fetch_add(&slot_count[slot_of(last)], run, __ATOMIC_RELAXED);
                last = key;
                run = 1;
            }
            if (run) __atomic_fetch_add(&slot_count[slot_of(last)], run, __ATOMIC_RELAXED);
        });
        for (size_t sl = 0; sl < slot_count.size(); ++sl)
            if (slot_count[sl]) {
                shape_key.push_back((uint32_t)(all.r_min + sl / win_h) << 16 | (uint32_t)(all.h_min + sl % win_h));
                shape_slot.push_back((uint32_t)sl);
            }
    } else {
        for (const Plan &p : gen) {
            const auto it = sparse_slot.emplace(key_of(p), (uint32_t)slot_count.size());
            if (it.second) {
                slot_count.push_back(0);
                shape_key.push_back(key_of(p));
                shape_slot.push_back(it.first->second);
            }
            ++slot_count[it.first->second];
        }
    }
    slot_tile.assign(slot_count.size(), (uint16_t)0xff00);
    const int64_t n_shapes = (int64_t)shape_key.size();
    const uint64_t all_classes = ~0ull;
    std::vector<double> class_work((size_t)ct.n, 0.0);
    double waves_est = 0;
    auto tile_all = [&](double beta) {
        std::vector<std::vector<double>> work((size_t)nthr, std::vector<double>((size_t)ct.n + 1, 0.0)); // [ct.n]: waves
        agx_parallel_for(n_shapes, 512, [&](int64_t lo, int64_t hi, int t) {
            std::vector<double> &wk = work[(size_t)t];
            for (int64_t k = lo; k < hi; ++k) {
                const uint32_t key = shape_key[(size_t)k], cnt = slot_count[shape_slot[(size_t)k]];
                uint8_t c = 255, G = 0;
                double cost = 0;
                choose_tiling(kind, key >> 16, key & 0xffffu, all_classes, &c, &G, &cost, beta);
                slot_tile[shape_slot[(size_t)k]] = (uint16_t)(c << 8 | G);
                if (c < ct.n) {
                    wk[c] += cost * cnt;
                    wk[(size_t)ct.n] += (double)cnt * G / 64.0 / slots;
                }
            }
        });
        std::fill(class_work.begin(), class_work.end(), 0.0);
        waves_est = 0;
        for (const auto &wk : work) {
            for (int c = 0; c < ct.n; ++c) class_work[(size_t)c] += wk[(size_t)c];
            waves_est += wk[(size_t)ct.n];
        }
    };
    double beta = tail_beta_override() >= 0 ? tail_beta_override() : 0.0;
    tm1 = now_ms();
    tile_all(beta);
    // Tail regime (as in the SW planner): a batch whose waves fill the chip's resident capacity (3 waves
    // per SIMD for the packed kernel, 2 for the others) less than 1.6 times lasts as long as its longest
    // waves; it is re-tiled with 2 lanes' worth of extra cost on a wave's own duration, which spreads
    // pairs over more lanes.  Mixed regions (tools/phmm_tail_beta_sweep.py): 2048 pairs +27 % packed /
    // +35 % double, 16 384 pairs +30 % / +3 %; beyond 1.6 fillings the term costs a few percent.
    if (tail_beta_override() < 0 && n_cu > 0 && waves_est / ((slots == 2 ? 3.0 : 2.0) * 4.0 * n_cu) < 1.6) {
        beta = 2.0;
        tile_all(beta);
    }
    {
        // small batches afford fewer launches: about one class per 16384 wavefronts of work
        // (tools/phmm_classes_sweep.py: the count matters little, fewer is never worse by more than 3 %)
        const int k_max = std::min(max_classes(), 1 + (int)(waves_est / 16384.0));
        int used = 0;
        for (double wk : class_work) used += wk > 0;
        if (used > k_max) {
            std::vector<int> order((size_t)ct.n);
            for (int k = 0; k < ct.n; ++k) order[(size_t)k] = k;
            std::sort(order.begin(), order.end(), [&](int x, int y) { return class_work[(size_t)x] > class_work[(size_t)y]; });
            uint64_t keep = 0;
            for (int k = 0; k < k_max; ++k) keep |= 1ull << order[(size_t)k];
            agx_parallel_for(n_shapes, 512, [&](int64_t lo, int64_t hi, int) {
                for (int64_t k = lo; k < hi; ++k) {
                    uint16_t &v = slot_tile[shape_slot[(size_t)k]];
                    if ((v >> 8) < 64 && ((keep >> (v >> 8)) & 1u)) continue;
                    uint8_t c = 255, G = 0;
                    choose_tiling(kind, shape_key[(size_t)k] >> 16, shape_key[(size_t)k] & 0xffffu, keep, &c, &G, nullptr, beta);
                    if (c < ct.n) v = (uint16_t)(c << 8 | G);
                }
            });
        }
    }
    tm2 = now_ms();
    {
        std::atomic<int64_t> unfit{-1};
        agx_parallel_for(n, 16384, [&](int64_t lo, int64_t hi, int) {
            uint32_t last_key = 0;
            uint16_t last_v = 0;
            bool have = false;
            for (int64_t gi = lo; gi < hi; ++gi) {
                Plan &p = gen[(size_t)gi];
                const uint32_t key = key_of(p);
                if (!have || key != last_key) {
                    last_v = slot_tile[one_shape ? 0 : slot_of(key)];
                    last_key = key;
                    have = true;
                }
                p.cls = (uint8_t)(last_v >> 8);
                p.G = (uint8_t)(last_v & 0xff);
                if (p.cls >= ct.n) {
                    int64_t none = -1;
                    unfit.compare_exchange_strong(none, gi);
                }
            }
        });
        if (unfit.load() >= 0) {
            const Plan &p = gen[(size_t)unfit.load()];
            agx_set_error("pair (read %u, hap %u): no lane tiling fits %u columns", p.read, p.hap, p.H);
            return AGX_E_LIMIT;
        }
    }
    // dominant (R, H) shape?
    if (n >= 1024 && n_cu > 0) {
        const size_t stride = (size_t)n / 512;
        uint32_t cand = 0;
        int votes = 0;
        for (size_t k = 0; k < 512; ++k) {
            const uint32_t key = key_of(gen[k * stride]);
            if (votes == 0) {
                cand = key;
                votes = 1;
            } else
                votes += key == cand ? 1 : -1;
        }
        std::atomic<int64_t> count{0};
        if (one_shape)
            count = n;
        else
            agx_parallel_for(n, 16384, [&](int64_t lo, int64_t hi, int) {
                int64_t c = 0;
                for (int64_t k = lo; k < hi; ++k) c += key_of(gen[(size_t)k]) == cand;
                count.fetch_add(c, std::memory_order_relaxed);
            });
        if (votes > 0 && count.load() * 2 >= n) {
            uint8_t c = 255, G = 0;
            const bool pair_reads = slots == 2 && trains != 1 && (trains == 2 || waves_est >= 16.0 * n_cu);
            choose_tiling_uniform(kind, cand >> 16, cand & 0xffffu, (count.load() + slots - 1) / slots, 4 * n_cu, pair_reads, &c, &G);
            if (c < ct.n)
                agx_parallel_for(n, 16384, [&](int64_t lo, int64_t hi, int) {
                    for (int64_t k = lo; k < hi; ++k)
                        if (key_of(gen[(size_t)k]) == cand) {
                            gen[(size_t)k].cls = c;
                            gen[(size_t)k].G = G;
                        }
                });
        }
    }
    tm3 = now_ms();
    // order: class, lanes per group (wide first), then long reads first, read, haplotype -- a wave's
    // groups then have similar row counts, and haplotypes of one read stay adjacent (one LDS table).
    // `gen` is (read, haplotype)-ordered: two stable counting passes, by read length, then by class.
    std::vector<Plan> plan;
    po.file_order = one_shape;
    if (one_shape)
        plan.swap(gen); // one (R, H) shape: one class, one G, equal keys throughout
    else {
        const uint32_t max_r = all.r_max;
        std::vector<Plan> tmp;
        counting_sort(gen, tmp, (size_t)max_r + 1, [&](const Plan &p) { return (size_t)(max_r - p.R); });
        counting_sort(tmp, plan, (size_t)ct.n * 64 + 1, [](const Plan &p) { return (size_t)p.cls * 64 + (size_t)(64 - p.G); });
    }
    std::vector<Plan>().swap(gen);
    tm4 = now_ms();
    // Waves and records.  The greedy filling below runs on pieces of `plan` that start where the read, the class or
    // the group width changes (a wave never spans such a piece's end; a packed group's partner is the next haplotype of
    // the SAME read, so no group does either); the pieces' records are then laid end to end.
    // Read trains (packed float fill, fast cell): two consecutive reads of the sorted order whose runs in this bucket hold
    // the same haplotypes -- reads of one region -- share their lane groups: the second enters as the first leaves, the skew
    // is paid once (K (R + 1) + G - 1 steps for K = 2 reads instead of K (R + G - 1)).  Measured envelope, tools/train_emulation.py:
    // +5.4 % on config 3, +5.8 % at four times its size; trains of four +4 %, of eight a loss (their tables outgrow the LDS
    // share of a wave).  Only where the paired waves still fill the chip: from two waves per SIMD after pairing.
    bool use_trains = slots == 2 && trains != 1 && (trains == 2 || waves_est >= 16.0 * n_cu);
    // the run of entries of plan[a]'s read in its (class, G) bucket ends at run_end; returns how many entries on its partner
    // run -- the next read, same bucket, same haplotypes -- starts (0: none)
    auto partner_of = [&](size_t a, const size_t end, size_t &run_end) -> size_t {
        const uint32_t rd = plan[a].read;
        const int cls = plan[a].cls, G = plan[a].G;
        size_t a_end = a;
        while (a_end < end && plan[a_end].read == rd && plan[a_end].cls == cls && plan[a_end].G == G) ++a_end;
        run_end = a_end;
        if (a_end >= end || plan[a_end].cls != cls || plan[a_end].G != G) return 0;
        // (widths 31 and 32 sit at their 256-register cap: the train build spills 15-25 values and gains nothing --
        // config 5's packed float shard -1.5 %, four times its size +1.6 %, tools/trains_shapes.py)
        if (trains != 2 && ct.C[cls] > 30) return 0;
        const uint32_t rd2 = plan[a_end].read;
        size_t b_end = a_end;
        while (b_end < end && plan[b_end].read == rd2 && plan[b_end].cls == cls && plan[b_end].G == G) ++b_end;
        if (b_end - a_end != a_end - a) return 0;
        for (size_t k = 0; k < a_end - a; ++k)
            if (plan[a + k].hap != plan[a_end + k].hap) return 0;
        // the tables a wave of such groups needs must fit its LDS share: a run that does not fill a wave shares it with
        // its neighbours' tables (mixed regions), and a train's table is twice a read's
        const size_t run_groups = (a_end - a + 1) / 2, per_wave = (size_t)(64 / G);
        const size_t tables = trains == 2 ? 1 : (per_wave + run_groups - 1) / run_groups + (run_groups % per_wave ? 1 : 0); // (forced: one table must fit)
        if (tables * ph_pk_tab_bytes(plan[a].R + 1u + plan[a_end].R + 2u * ((uint32_t)G - 1u)) > tab_budget()) return 0;
        return a_end - a;
    };
    auto fill_waves = [&](size_t i, const size_t end, PlanOut &o) {
        size_t run_end = i, run_delta = 0; // the read run i is in ends at run_end; its partner run starts run_delta entries on (0: none)
        auto find_partner = [&](size_t a) { run_delta = partner_of(a, end, run_end); };
        while (i < end) {
            const int cls = plan[i].cls;
            ClassLaunch cl;
            cl.C = ct.C[cls];
            cl.first_wave = (uint32_t)o.waves.size();
            while (i < end && plan[i].cls == cls) {
                const int G = plan[i].G;
                const int per_wave = 64 / G;
                PhWave w{};
                w.first_group = (uint32_t)(slots == 2 ? o.groups2.size() : o.groups1.size());
                w.first_tab = (uint32_t)o.tabs.size();
                w.G = (uint16_t)G;
                int cnt = 0;
                uint32_t steps = 0, ntabs = 0, last_read = 0xffffffffu;
                size_t lut_bytes = 0; // lut_rows: every table is as long as its own read (+ a neutral row at either end)
                while (i < end && plan[i].cls == cls && plan[i].G == G && cnt < per_wave) {
                    const Plan &p = plan[i];
                    if (use_trains && i >= run_end) find_partner(i);
                    const size_t tr = use_trains ? run_delta : 0; // this entry's second read sits tr entries on
                    const uint32_t R2 = tr ? plan[i + tr].R : 0u;
                    const uint32_t nsteps = std::max(steps, p.R + (tr ? R2 + 1u : 0u) + (uint32_t)G - 1u);
                    const uint32_t ntabs_new = ntabs + (p.read != last_read ? 1u : 0u);
                    if (lut_rows) {
                        if (cnt > 0 && p.read != last_read && lut_bytes + ph_lut_lds_bytes(p.R + 2u) > tab_budget()) break;
                    } else if (cnt > 0 && ntabs_new > 1 && tab_bytes(rows_f64, nsteps + G - 1) * ntabs_new > tab_budget())
                        break;
                    if (p.read != last_read) {
                        // (lut_rows: the table's offset / 16 rides in the upper half of R, agx_phmm_lut_kernel.hip)
                        o.tabs.push_back(PhTab{read_dw[p.read], lut_rows ? p.R | (uint32_t)(lut_bytes / 16) << 16 : p.R});
                        if (use_trains) o.tabs.push_back(tr ? PhTab{read_dw[plan[i + tr].read], R2} : PhTab{0u, 0u});
                        if (lut_rows) lut_bytes += ph_lut_lds_bytes(p.R + 2u);
                        last_read = p.read;
                    }
                    ntabs = ntabs_new;
                    steps = nsteps;
                    if (slots == 2) {
                        PhGroup2 g{};
                        g.R_tab = p.R | ((ntabs - 1) << 16);
                        g.hap_dw[0] = hap_dw[p.hap];
                        g.H[0] = p.H;
                        g.out[0] = p.out;
                        g.init32[0] = FLT_MAX / 16 / (float)p.H;
                        // second slot: the next haplotype of the same read in this class, else vacant
                        g.hap_dw[1] = 0;
                        g.H[1] = 0;
                        g.out[1] = vacant_out;
                        g.init32[1] = 0;
                        g.R2 = R2;
                        g.out2[0] = tr ? plan[i + tr].out : vacant_out;
                        g.out2[1] = vacant_out;
                        if (i + 1 < end && plan[i + 1].cls == cls && plan[i + 1].G == G && plan[i + 1].read == p.read) {
                            const Plan &q = plan[i + 1];
                            g.hap_dw[1] = hap_dw[q.hap];
                            g.H[1] = q.H;
                            g.out[1] = q.out;
                            g.init32[1] = FLT_MAX / 16 / (float)q.H;
                            if (tr) g.out2[1] = plan[i + 1 + tr].out;
                            ++i;
                        }
                        o.groups2.push_back(g);
                    } else {
                        PhGroup g{};
                        g.hap_dw = hap_dw[p.hap];
                        g.H = p.H;
                        g.R_tab = p.R | ((ntabs - 1) << 16);
                        g.out = p.out;
                        g.init64 = DBL_MAX / 16 / (double)p.H;
                        g.init32 = FLT_MAX / 16 / (float)p.H;
                        o.groups1.push_back(g);
                    }
                    ++cnt;
                    ++i;
                    if (use_trains && run_delta && i == run_end) { // the partner run rode along: step over it
                        i += run_delta;
                        run_end = i;
                        run_delta = 0;
                    }
                }
                w.n_groups = (uint16_t)cnt;
                w.n_tabs = (uint16_t)ntabs;
                w.steps = steps;
                if (trace) {
                    const int64_t cols = (int64_t)G * cl.C;
                    for (int k = 0; k < cnt; ++k) {
                        uint32_t R = 0, H[2] = {0, 0}, extra = 0; // extra: a train's reset row
                        if (slots == 2) {
                            const PhGroup2 &g = o.groups2[w.first_group + (size_t)k];
                            R = (g.R_tab & 0xffffu) + g.R2, H[0] = g.H[0], H[1] = g.H[1], extra = g.R2 ? 1u : 0u;
                        } else {
                            const PhGroup &g = o.groups1[w.first_group + (size_t)k];
                            R = g.R_tab & 0xffffu, H[0] = g.H;
                        }
                        for (int h = 0; h < slots; ++h) {
                            o.waste[0] += (int64_t)R * H[h];
                            o.waste[1] += (int64_t)R * (cols - H[h]);
                            o.waste[2] += (int64_t)(G - 1 + (int)extra) * cols;
                            o.waste[3] += (int64_t)(steps - (R + extra + (uint32_t)G - 1u)) * cols;
                        }
                    }
                    o.waste[4] += (int64_t)steps * (64 - (int64_t)cnt * G) * cl.C * slots;
                }
                if (G != 16) cl.all_g16 = false;
                cl.lds = std::max(cl.lds, lut_rows ? lut_bytes : tab_bytes(rows_f64, steps + G - 1) * ntabs);
                if (kind == 2) cl.lds_rescue = std::max(cl.lds_rescue, ph_tab_bytes(true, gatk_prior, steps + G - 1) * ntabs); // (only the float fill's records are reused for a double pass)
                o.padded += (int64_t)steps * 64 * cl.C * slots;
                o.waves.push_back(w);
            }
            cl.n_waves = (uint32_t)o.waves.size() - cl.first_wave;
            o.launches.push_back(cl);
        }
    };
    const std::vector<size_t> cut = cut_at_runs(plan.size(), run_parts, [&](size_t i) {
        return plan[i].read != plan[i - 1].read || plan[i].cls != plan[i - 1].cls || plan[i].G != plan[i - 1].G;
    });
    const int pieces = (int)cut.size() - 1;
    auto build = [&](PlanOut &po) {
    if (pieces <= 1)
        fill_waves(0, plan.size(), po);
    else {
        std::vector<PlanOut> part((size_t)pieces);
        agx_pool_run(pieces, [&](int t) {
            PlanOut &o = part[(size_t)t];
            const size_t span = cut[(size_t)t + 1] - cut[(size_t)t];
            (slots == 2 ? o.groups2.reserve(span / 2 + 64) : o.groups1.reserve(span));
            o.waves.reserve(span / 4 + 16);
            o.tabs.reserve(span / 8 + 16);
            fill_waves(cut[(size_t)t], cut[(size_t)t + 1], o);
        });
        size_t n_groups = 0, n_tabs = 0, n_waves = 0;
        for (const PlanOut &o : part) n_groups += o.groups1.size() + o.groups2.size(), n_tabs += o.tabs.size(), n_waves += o.waves.size();
        (slots == 2 ? po.groups2.resize(n_groups) : po.groups1.resize(n_groups));
        po.tabs.resize(n_tabs);
        po.waves.resize(n_waves);
        std::vector<size_t> g0((size_t)pieces), t0((size_t)pieces), w0((size_t)pieces);
        size_t ga = 0, ta = 0, wa = 0;
        for (int t = 0; t < pieces; ++t) {
            const PlanOut &o = part[(size_t)t];
            g0[(size_t)t] = ga, t0[(size_t)t] = ta, w0[(size_t)t] = wa;
            ga += o.groups1.size() + o.groups2.size(), ta += o.tabs.size(), wa += o.waves.size();
            po.padded += o.padded;
            for (int k = 0; k < 5; ++k) po.waste[k] += o.waste[k];
            for (ClassLaunch cl : o.launches) { // a class that runs on into the next piece is one launch
                cl.first_wave += (uint32_t)w0[(size_t)t];
                if (!po.launches.empty() && po.launches.back().C == cl.C) {
                    ClassLaunch &m = po.launches.back();
                    m.n_waves += cl.n_waves;
                    m.all_g16 = m.all_g16 && cl.all_g16;
                    m.lds = std::max(m.lds, cl.lds);
                    m.lds_rescue = std::max(m.lds_rescue, cl.lds_rescue);
                } else
                    po.launches.push_back(cl);
            }
        }
        agx_pool_run(pieces, [&](int t) {
            const PlanOut &o = part[(size_t)t];
            if (slots == 2)
                std::copy(o.groups2.begin(), o.groups2.end(), po.groups2.begin() + (ptrdiff_t)g0[(size_t)t]);
            else
                std::copy(o.groups1.begin(), o.groups1.end(), po.groups1.begin() + (ptrdiff_t)g0[(size_t)t]);
            std::copy(o.tabs.begin(), o.tabs.end(), po.tabs.begin() + (ptrdiff_t)t0[(size_t)t]);
            for (size_t k = 0; k < o.waves.size(); ++k) {
                PhWave w = o.waves[k];
                w.first_group += (uint32_t)g0[(size_t)t];
                w.first_tab += (uint32_t)t0[(size_t)t];
                po.waves[w0[(size_t)t] + k] = w;
            }
        });
    }
    };
    // (one shape throughout: every run pairs like the first, and pairing only removes steps -- one plan is made, whichever)
    size_t probe_end = 0;
    const bool sure = use_trains && one_shape && !plan.empty() && partner_of(0, plan.size(), probe_end) != 0;
    if (use_trains && one_shape && !sure && trains != 2) use_trains = false;
    if (use_trains) {
        // Trains double a table: where a wave's groups come from several reads (mixed regions: a read's haplotype pairs do
        // not fill a wave) fewer tables fit the wave's LDS share and lanes stay empty -- on the reference's corpus shape
        // useful cells 0.85 -> 0.77.  Both plans are made (the wave filling is a tenth of the planner) and the one with
        // fewer padded cells stays; forced trains (AGX_PHMM_TRAINS_ON) skip the comparison.
        PlanOut with;
        build(with);
        with.trains = true;
        bool keep = trains == 2 || sure;
        if (!keep) {
            use_trains = false;
            build(po);
            keep = (double)with.padded < 0.985 * (double)po.padded;
            if (trace)
                fprintf(stderr, "[phmm make_plan kind %d] read trains: %zu waves / %.4g padded cells with, %zu / %.4g without -> %s\n", kind, with.waves.size(),
                        (double)with.padded, po.waves.size(), (double)po.padded, keep ? "with" : "without");
        }
        if (keep) {
            with.file_order = po.file_order;
            po = std::move(with);
        }
    } else
        build(po);
    for (const ClassLaunch &cl : po.launches) {
        // dispatch order = longest waves first (a wave lasts steps x C): the buckets were filled widest
        // group first, which leaves narrow groups with long reads for the end of the launch
        if (!one_shape)
            std::stable_sort(po.waves.begin() + cl.first_wave, po.waves.begin() + cl.first_wave + cl.n_waves,
                             [](const PhWave &a, const PhWave &b) { return a.steps > b.steps; });
        if (std::max(cl.lds, cl.lds_rescue) > 160 * 1024) {
            agx_set_error("a read table of %zu bytes does not fit the 160 KiB LDS", cl.lds);
            return AGX_E_LIMIT;
        }
    }
    if (trace)
        fprintf(stderr, "[phmm make_plan kind %d] pairing+shapes %.2f, tiling %.2f, assign %.2f, sort %.2f, waves+records %.2f ms (%d pieces)\n",
                kind, tm1 - tm0, tm2 - tm1, tm3 - tm2, tm4 - tm3, now_ms() - tm4, pieces);
    if (trace && po.padded > 0) {
        const double t = (double)po.padded;
        fprintf(stderr, "[phmm make_plan kind %d] %zu waves, padded cells %.4g: useful %.3f, columns beyond H %.3f, skew %.3f, steps beyond the group's read %.3f, "
                        "lanes without a group %.3f; read trains %d; classes",
                kind, po.waves.size(), t, po.waste[0] / t, po.waste[1] / t, po.waste[2] / t, po.waste[3] / t, po.waste[4] / t, (int)po.trains);
        for (const ClassLaunch &cl : po.launches) fprintf(stderr, " %dx%u", cl.C, cl.n_waves);
        fprintf(stderr, "\n");
    }
    return AGX_OK;
}

} // namespace

// accuracy guard of the packed float fill (PhUnderflow, agx_phmm.h): |log10 L| below this times sqrt(R) -> double.
// 1.24 x the worst error found over 3.6 million pairs (reads of 10 ... 2000 bases, three error rates; tools/phmm_f32_guard_cal.py)
constexpr double kGuardRef = 0.18, kGuardGatk = 0.40;

// agx_phmm_forward sends a batch of more than two of these through in pieces of whole regions (see there)
constexpr double kPhmmPieceCells = 4.0e9;

struct agx_phmm_batch {
    agx_ctx *ctx = nullptr; // retained
    int precision = AGX_PHMM_F64;
    bool probs = false; // read tracks are probabilities (pairHMM() seam), not Phred characters
    bool gatk_prior = false;
    int64_t n_pairs = 0;
    bool packed = false;          // main plan uses PhGroup2 records (AGX_PHMM_F32_FMA)
    bool rescue_pending = false;  // packed batches: the rescue plan has not run for the last launch (it runs from
                                  // agx_phmm_batch_results, and only when the fill counted a pair below the float range)
    bool trains = false;          // ... with read trains (PlanOut::trains)
    bool fast = false;            // ... and its fill runs the fast cell (plain DNA, no Phred-0 gap-continuation quality in the batch)
    bool lut_ring = false;        // ... and some read's table is a ring (ph_lut_is_ring): the STREAM builds
    bool lut_prior = false;       // double modes on plain DNA: priors looked up in the read tables (agx_phmm_lut_kernel.hip)
    bool separate_rescue = false; // the double rescue pass has its own plan (packed batches), made on first use from:
    std::unique_ptr<PlanSeed> rescue_seed;
    bool rescue_broken = false; // making that plan failed (out of memory): the batch's float results cannot be completed
    DevBuf img, sums, lut, counter;
    bool counters_dirty = false; // a launch has counted into `counter` and nobody has taken (and reset) the counts yet
    PinBuf out_stage; // page-locked landing block of the results, taken at create (agx_phmm_batch_results allocates nothing)
    struct DevPlan {
        DevBuf groups, tabs, waves;
        std::vector<ClassLaunch> launches;
    } main, rescue, stripe; // stripe: pairs whose haplotype no class spans, one per wavefront
    DevBuf stripe_scratch;    // 6 * stripe_rows doubles per workgroup of the striped launch
    uint32_t stripe_rows = 0, stripe_grid = 0;
    bool stripe_mis_div = false; // the striped launch derives the GATK mismatch prior itself (see create)
    agx_phmm_info info{};
    // agx_phmm_batch_bind_results: a page-locked array of the caller's that a packed float fill in output order writes its
    // log10 likelihoods into itself; bound_flag (in out_stage) says whether a pair went to the rescue plan instead
    bool file_order = false;
    double *bound = nullptr;
    PinBuf bound_flag;
    // the fast cell's table rows of every read, made once at creation (phmm_pk_rows); rows_base_dw: image word of the first read
    DevBuf pk_rows;
    uint32_t rows_base_dw = 0;
};

namespace {

// Shared by the byte-track API and the probability-track seam.  When prob[] is non-NULL it holds
// the four probability tracks of every read (Qr,Qi,Qd,Qg concatenated per read, read_off units).
int create_batch(agx_ctx *ctx, const agx_phmm_desc *d, const double *const prob[4], int precision, agx_phmm_batch **out)
{
    if (!out) {
        agx_set_error("agx_phmm_batch_create: out is NULL");
        return AGX_E_ARG;
    }
    *out = nullptr;
    // ctx == NULL: plan only (no device needed) -- the batch answers agx_phmm_batch_info() and nothing else
    int rc = ctx ? agx_bind(ctx) : AGX_OK;
    if (rc) return rc;
    const int n_cu = ctx ? ctx->n_cu : 256;
    const bool gatk_prior = (precision & AGX_PHMM_GATK_PRIOR) != 0;
    precision &= ~AGX_PHMM_GATK_PRIOR;
    if (!d || precision < AGX_PHMM_F64 || precision > AGX_PHMM_F32_FMA) {
        agx_set_error("agx_phmm_batch_create: bad descriptor or precision %d", precision);
        return AGX_E_ARG;
    }
    if (d->n_regions && (!d->region_read || !d->region_hap || !d->read_off || !d->hap_off)) {
        agx_set_error("agx_phmm_batch_create: NULL offset table");
        return AGX_E_ARG;
    }
    const bool f64 = precision == AGX_PHMM_F64 || precision == AGX_PHMM_F64_FMA;
    const bool packed = precision == AGX_PHMM_F32_FMA;
    const bool probs = prob != nullptr;
    if (probs && precision != AGX_PHMM_F64) {
        agx_set_error("probability tracks are only supported with AGX_PHMM_F64");
        return AGX_E_ARG;
    }

    const bool trace = agx_tune("AGX_TRACE_CREATE") != nullptr;
    auto now = [] { return now_ms(); };
    const double t_begin = now();
    // ---- enumerate the pairs in output order: region, read, haplotype (threads over regions)
    std::unique_ptr<PlanSeed> seed_holder(new PlanSeed());
    PlanSeed &seed = *seed_holder;
    seed.n_cu = n_cu;
    seed.gatk_prior = gatk_prior;
    std::vector<Plan> &gen0 = seed.gen0;
    int64_t n_pairs = 0, cells = 0;
    // A descriptor whose regions name only part of its reads and haplotypes -- a shard of agx_phmm_forward_devices, a
    // piece of agx_phmm_forward: they keep the caller's absolute indices -- is narrowed to that part first: the image
    // holds what the regions use, not every read and haplotype of the caller (each of config 5's eight shards carried
    // the whole batch's 33 MB before).  r_base / h_base: what error messages add to name the caller's numbers.
    agx_phmm_desc view;
    std::vector<uint32_t> view_rr, view_rh;
    uint32_t r_base = 0, h_base = 0;
    {
        const uint32_t ng = d->n_regions;
        uint32_t rmin = d->n_reads, rmax = 0, hmin = d->n_haps, hmax = 0;
        for (uint32_t g = 0; g < ng; ++g) {
            const uint32_t r0 = d->region_read[g], r1 = d->region_read[g + 1];
            const uint32_t h0 = d->region_hap[g], h1 = d->region_hap[g + 1];
            if (r1 < r0 || h1 < h0 || r1 > d->n_reads || h1 > d->n_haps) {
                agx_set_error("region %u: read/haplotype ranges out of order or out of bounds", g);
                return AGX_E_ARG;
            }
            rmin = std::min(rmin, r0), rmax = std::max(rmax, r1);
            hmin = std::min(hmin, h0), hmax = std::max(hmax, h1);
        }
        if (ng && (rmin > 0 || rmax < d->n_reads || hmin > 0 || hmax < d->n_haps)) {
            view = *d;
            view_rr.assign(d->region_read, d->region_read + ng + 1);
            view_rh.assign(d->region_hap, d->region_hap + ng + 1);
            for (uint32_t &v : view_rr) v -= rmin;
            for (uint32_t &v : view_rh) v -= hmin;
            view.region_read = view_rr.data();
            view.region_hap = view_rh.data();
            view.read_off = d->read_off + rmin; // offsets into the tracks stay absolute
            view.hap_off = d->hap_off + hmin;
            view.n_reads = rmax - rmin;
            view.n_haps = hmax - hmin;
            r_base = rmin, h_base = hmin;
            d = &view;
        }
    }
    {
        const uint32_t ng = d->n_regions;
        // lengths are checked once per read / haplotype, not once per pair
        for (uint32_t r = 0; d->read_off && r < d->n_reads; ++r)
            if (d->read_off[r + 1] - d->read_off[r] > AGX_PHMM_MAX_READ_LEN) {
                agx_set_error("read %u: %llu bases exceed the supported %d", r + r_base, (unsigned long long)(d->read_off[r + 1] - d->read_off[r]),
                              AGX_PHMM_MAX_READ_LEN);
                return AGX_E_LIMIT;
            }
        for (uint32_t h = 0; d->hap_off && h < d->n_haps; ++h)
            if (d->hap_off[h + 1] - d->hap_off[h] > AGX_PHMM_MAX_HAP_LEN) {
                agx_set_error("haplotype %u: %llu bases exceed the supported %d", h + h_base, (unsigned long long)(d->hap_off[h + 1] - d->hap_off[h]),
                              AGX_PHMM_MAX_HAP_LEN);
                return AGX_E_LIMIT;
            }
        // per region: output offset (every pair has an output slot) and offset into gen0 (pairs with work)
        std::vector<int64_t> out0((size_t)ng + 1, 0), fill0((size_t)ng + 1, 0);
        for (uint32_t g = 0; g < ng; ++g) {
            const uint32_t r0 = d->region_read[g], r1 = d->region_read[g + 1], h0 = d->region_hap[g], h1 = d->region_hap[g + 1];
            int64_t nr = 0, nh = 0;
            for (uint32_t r = r0; r < r1; ++r) nr += d->read_off[r + 1] != d->read_off[r];
            for (uint32_t h = h0; h < h1; ++h) nh += d->hap_off[h + 1] != d->hap_off[h];
            out0[g + 1] = out0[g] + (int64_t)(r1 - r0) * (h1 - h0);
            fill0[g + 1] = fill0[g] + nr * nh;
            cells += (int64_t)(d->read_off[r1] - d->read_off[r0]) * (int64_t)(d->hap_off[h1] - d->hap_off[h0]);
            if (out0[g + 1] > 0x7ffffff0LL) {
                agx_set_error("more than 2^31 pairs in one batch");
                return AGX_E_LIMIT;
            }
        }
        n_pairs = out0[ng];
        seed.n_pairs = n_pairs;
        gen0.resize((size_t)fill0[ng]);
        // (grain: about 16384 pairs per part -- smaller parts cost more in hand-over than they save)
        const int64_t region_grain = std::max<int64_t>(1, 16384 / std::max<int64_t>(1, n_pairs / std::max<uint32_t>(ng, 1)));
        agx_parallel_for((int64_t)ng, region_grain, [&](int64_t ga, int64_t gz, int) {
            for (int64_t g = ga; g < gz; ++g) {
                const uint32_t r0 = d->region_read[g], r1 = d->region_read[g + 1], h0 = d->region_hap[g], h1 = d->region_hap[g + 1];
                size_t at = (size_t)fill0[(size_t)g];
                int64_t o = out0[(size_t)g];
                for (uint32_t r = r0; r < r1; ++r) {
                    const uint32_t R = (uint32_t)(d->read_off[r + 1] - d->read_off[r]);
                    for (uint32_t h = h0; h < h1; ++h, ++o) {
                        const uint32_t H = (uint32_t)(d->hap_off[h + 1] - d->hap_off[h]);
                        if (R == 0 || H == 0) continue; // sum over an empty row/all-zero row is 0 (log10 -> -inf)
                        Plan p{};
                        p.out = (uint32_t)o;
                        p.read = r;
                        p.hap = h;
                        p.R = R;
                        p.H = H;
                        gen0[at++] = p;
                    }
                }
            }
        });
    }

    // ---- haplotypes wider than 64 lanes x the widest class of this arithmetic go to the striped kernel
    std::vector<Plan> gen_long;
    {
        const ClassTable ct = class_table(precision);
        uint32_t span = 0;
        for (int ci = 0; ci < ct.n; ++ci)
            if (ct.cost[ci] != 0) span = std::max(span, 64u * (uint32_t)ct.C[ci]);
        uint64_t longest = 0;
        for (uint32_t h = 0; d->hap_off && h < d->n_haps; ++h) longest = std::max<uint64_t>(longest, d->hap_off[h + 1] - d->hap_off[h]);
        if (longest > span) {
            size_t keep = 0;
            for (const Plan &p : gen0) {
                if (p.H > span)
                    gen_long.push_back(p);
                else
                    gen0[keep++] = p;
            }
            gen0.resize(keep);
        }
    }

    // ---- image: every read and haplotype once, shared by all plans of the batch.  Offsets come from two
    // prefix sums, the bytes are copied by threads (the pairHMM() seam's probability tracks excepted: one pair);
    // with a device the image is built straight in pinned staging memory.
    const size_t zero_dw = (64 * 32 + 8) / 4; // words 0..: an all-zero haplotype block for vacant packed slots (64 lanes x 32 columns)
    const uint32_t n_reads = d->read_off ? d->n_reads : 0, n_haps = d->hap_off ? d->n_haps : 0;
    std::vector<uint32_t> &read_dw = seed.read_dw, &hap_dw = seed.hap_dw;
    read_dw.assign(n_reads, 0xffffffffu);
    hap_dw.assign(n_haps, 0xffffffffu);
    size_t img_dw = zero_dw;
    const size_t stripe_bytes = 64u * (size_t)stripe_cols();
    auto hap_block_dw = [&](size_t H) { // zero slack: any class tiling, and whole stripes of the striped kernel, read in bounds
        const size_t bytes = std::max(H + kHapSlack, (H + stripe_bytes - 1) / stripe_bytes * stripe_bytes + 8);
        return (bytes + 3) / 4;
    };
    for (uint32_t r = 0; r < n_reads; ++r) {
        const size_t R = (size_t)(d->read_off[r + 1] - d->read_off[r]), trk = (R + 3) / 4;
        if (probs && (img_dw & 1)) ++img_dw; // doubles need 8-byte alignment
        read_dw[r] = (uint32_t)img_dw;
        img_dw += probs ? R * 8 + trk : 5 * trk;
        if (img_dw > 0xfffffff0ull) break;
    }
    const size_t reads_end_dw = img_dw; // the reads' tracks lie in [zero_dw, reads_end_dw), 5 x ceil(R / 4) words each
    for (uint32_t h = 0; h < n_haps && img_dw <= 0xfffffff0ull; ++h) {
        hap_dw[h] = (uint32_t)img_dw;
        img_dw += hap_block_dw((size_t)(d->hap_off[h + 1] - d->hap_off[h]));
    }
    if (img_dw > 0xfffffff0ull) {
        agx_set_error("packed image exceeds 16 GiB; split the batch");
        return AGX_E_LIMIT;
    }
    PinBuf img_pin;
    std::vector<uint32_t> img_heap;
    struct ImgGuard {
        PinBuf &p;
        ~ImgGuard() { p.release(); }
    } img_guard{img_pin};
    uint32_t *img = nullptr;
    if (ctx) {
        rc = img_pin.alloc(ctx, img_dw * 4);
        if (rc) return rc;
        img = (uint32_t *)img_pin.p;
    } else {
        img_heap.resize(img_dw);
        img = img_heap.data();
    }
    memset(img, 0, zero_dw * 4);
    const bool have_tracks = d->read_bases && d->hap_bases && (probs || (d->q_base && d->q_ins && d->q_del && d->q_gcp));
    if (!gen0.empty() && !have_tracks) {
        agx_set_error("agx_phmm_batch_create: NULL track");
        return AGX_E_ARG;
    }
    // packed float fill: its fast cell (agx_phmm_pk_kernel.hip) divides by 1 - Qg and codes the bases in two bits, so a
    // gap-continuation quality of Phred 0 or below, a read base outside ACGTN or a haplotype base outside ACGT anywhere
    // in the batch keeps the plain cell
    std::atomic<bool> not_fast{false}, not_dna{false}; // not_dna: a read base outside ACGTN or a haplotype base outside ACGT
    std::atomic<bool> wild{false};                     // a quality byte below '!' (as a signed char): no read trains
    static const auto dna_table = [] {
        std::array<uint8_t, 256> t{};
        for (const char *p = "ACGT"; *p; ++p) t[(uint8_t)*p] = 3; // bit 0: allowed in a read, bit 1: in a haplotype
        t[(uint8_t)'N'] = 1;
        return t;
    }();
    if (have_tracks) {
        agx_parallel_for((int64_t)n_reads, 2048, [&](int64_t ra, int64_t rz, int) {
            bool zero = false, other = false, low = false;
            for (int64_t r = ra; r < rz && !probs && precision != AGX_PHMM_F32; ++r)
                for (uint64_t k = d->read_off[r]; k < d->read_off[r + 1]; ++k) {
                    zero |= packed && (int8_t)d->q_gcp[k] <= (int8_t)'!'; // signed as the reference's char: 0x80.. are "probabilities" above 1
                    other |= !(dna_table[d->read_bases[k]] & 1);
                    low |= packed && ((int8_t)d->q_base[k] < (int8_t)'!' || (int8_t)d->q_ins[k] < (int8_t)'!' || (int8_t)d->q_del[k] < (int8_t)'!');
                }
            if (zero || other) not_fast.store(true, std::memory_order_relaxed);
            if (low) wild.store(true, std::memory_order_relaxed);
            if (other) not_dna.store(true, std::memory_order_relaxed);
            for (int64_t r = ra; r < rz; ++r) {
                const uint64_t o = d->read_off[r];
                const size_t R = (size_t)(d->read_off[r + 1] - o), trk = (R + 3) / 4;
                if (probs) {
                    double *q = reinterpret_cast<double *>(img + read_dw[(size_t)r]);
                    for (int k = 0; k < 4; ++k) memcpy(q + (size_t)k * R, prob[k] + o, R * sizeof(double));
                    uint8_t *c = reinterpret_cast<uint8_t *>(q + 4 * R);
                    memcpy(c, d->read_bases + o, R);
                    memset(c + R, 0, trk * 4 - R);
                } else {
                    uint8_t *p = reinterpret_cast<uint8_t *>(img + read_dw[(size_t)r]);
                    const uint8_t *src[5] = {d->read_bases, d->q_base, d->q_ins, d->q_del, d->q_gcp};
                    for (int k = 0; k < 5; ++k) {
                        memcpy(p + (size_t)k * trk * 4, src[k] + o, R);
                        memset(p + (size_t)k * trk * 4 + R, 0, trk * 4 - R);
                    }
                }
            }
        });
        agx_parallel_for((int64_t)n_haps, 2048, [&](int64_t ha, int64_t hz, int) {
            for (int64_t h = ha; h < hz; ++h) {
                const uint64_t o = d->hap_off[h];
                const size_t H = (size_t)(d->hap_off[h + 1] - o);
                uint8_t *p = reinterpret_cast<uint8_t *>(img + hap_dw[(size_t)h]);
                memcpy(p, d->hap_bases + o, H);
                if (!probs && precision != AGX_PHMM_F32) {
                    bool other = false;
                    for (size_t k = 0; k < H; ++k) other |= !(dna_table[d->hap_bases[o + k]] & 2);
                    if (other) {
                        not_fast.store(true, std::memory_order_relaxed);
                        not_dna.store(true, std::memory_order_relaxed);
                    }
                }
                memset(p + H, 0, hap_block_dw(H) * 4 - H);
            }
        });
    }
    if (trace) fprintf(stderr, "[phmm create] enumerate %.2f ms\n", now() - t_begin);
    // A packed float batch cannot reuse its records for the double rescue pass: that pass has a plan of its own, made
    // from the kept seed when a fill first counts a pair below the float range (ensure_rescue_plan).
    // double modes on plain DNA run the kernel whose read tables carry the priors (other rows, other LDS sizes)
    // (... while the longest read's table stays within a wave's LDS share when two waves per SIMD are resident -- 20 KB, 363
    // rows of 56 bytes: a launch reserves its longest table for every wave, so beyond that fewer waves fit a CU and the
    // 33-byte rows of phmm_fill win -- 7 % at reads of 420 bases, 19 % at 500, 20 % at 600 -- although that kernel selects
    // its priors (tools/lut_vs_select_long_reads.py, profiles/r03as_lut_vs_select.log; the limit was 40 KB until round 3);
    // a 4096-row read fits the 160 KB only there)
    uint64_t longest_read = 0;
    for (uint32_t r = 0; r < n_reads; ++r) longest_read = std::max<uint64_t>(longest_read, d->read_off[r + 1] - d->read_off[r]);
    // (round 3, second step: such reads keep a 256-row ring in LDS, refilled as the wave advances -- the looked-up-prior kernel
    // for every read length; AGX_PHMM_NO_RING in the tuning build restores the 20 KB rule)
    const bool lut_ring = ph_lut_is_ring((uint32_t)longest_read + 2u);
    const bool lut_prior = f64 && !probs && have_tracks && !not_dna.load() && (!lut_ring || !agx_tune("AGX_PHMM_NO_RING")) && !agx_tune("AGX_PHMM_NO_LUT");
    const size_t n_work = gen0.size(); // pairs with work (every pair of the batch, unless a read or haplotype is empty)
    PlanOut pmain;
    // read trains: the fast cell only, and only on qualities that are probabilities (a byte below '!' is a "probability"
    // above 1: a first read whose state ran to infinity there would hand NaN to the second through the reset row's 0 x inf)
    const int trains_opt = ctx ? ctx->opt_phmm_trains : AGX_PHMM_TRAINS_AUTO;
    const bool may_train = packed && !not_fast.load() && !wild.load() && !agx_tune("AGX_PHMM_PLAIN_CELL") && !agx_tune("AGX_PHMM_NO_TRAINS");
    rc = packed ? make_plan(seed, gen0, 3, 2, false, false, trace, may_train ? (trains_opt == AGX_PHMM_TRAINS_ON ? 2 : trains_opt == AGX_PHMM_TRAINS_OFF ? 1 : 0) : 1, pmain)
                : make_plan(seed, std::move(gen0), precision, 1, f64, lut_prior, trace, 1, pmain);
    if (rc) return rc;
    // striped plan: one pair per wavefront, every pair its own read table
    PlanOut pstripe;
    uint32_t stripe_steps = 0;
    size_t stripe_lds = 0;
    for (const Plan &p : gen_long) {
        PhWave w{};
        w.first_group = (uint32_t)pstripe.groups1.size();
        w.first_tab = (uint32_t)pstripe.tabs.size();
        w.n_groups = 1;
        w.n_tabs = 1;
        w.G = 64;
        w.steps = p.R + 63u;
        pstripe.tabs.push_back(PhTab{read_dw[p.read], p.R});
        PhGroup g{};
        g.hap_dw = hap_dw[p.hap];
        g.H = p.H;
        g.R_tab = p.R;
        g.out = p.out;
        g.init64 = DBL_MAX / 16 / (double)p.H;
        g.init32 = FLT_MAX / 16 / (float)p.H;
        pstripe.groups1.push_back(g);
        pstripe.waves.push_back(w);
        stripe_steps = std::max(stripe_steps, w.steps);
        stripe_lds = std::max(stripe_lds, ph_tab_bytes(true, gatk_prior, w.steps + 63u));
        const int64_t n_stripes = (p.H + 64 * stripe_cols() - 1) / (64 * stripe_cols());
        pstripe.padded += n_stripes * w.steps * 64 * stripe_cols();
    }
    // The GATK prior's fifth table column (Qr / 3) does not fit beside the others for reads beyond 3 870 bases (41 bytes x
    // (R + 126) rows > 160 KiB): the striped launch then keeps four columns and divides in its step head -- the same IEEE
    // division the host's table was made with, so the results do not change.
    bool stripe_mis_div = false;
    if (gatk_prior && stripe_lds > 160 * 1024) {
        stripe_mis_div = true;
        stripe_lds = ph_tab_bytes(true, false, stripe_steps + 63u);
    }
    if (stripe_lds > 160 * 1024) {
        agx_set_error("a read table of %zu bytes does not fit the 160 KiB LDS", stripe_lds);
        return AGX_E_LIMIT;
    }
    const double t_plan = now();
    const int64_t padded = pmain.padded + pstripe.padded;

    const double t_pack = now();
    // ---- upload
    agx_phmm_batch *b = new agx_phmm_batch();
    struct Guard { // error paths: free whatever the batch holds
        agx_phmm_batch *&b;
        ~Guard()
        {
            if (b) agx_phmm_batch_destroy(b);
        }
    } guard{b};
    agx_ctx_retain(ctx);
    b->ctx = ctx;
    b->precision = precision;
    b->probs = probs;
    b->gatk_prior = gatk_prior;
    b->packed = packed;
    b->file_order = packed && pmain.file_order && pstripe.waves.empty() && (int64_t)n_work == n_pairs;
    b->fast = packed && !not_fast.load() && !agx_tune("AGX_PHMM_PLAIN_CELL");
    b->trains = packed && pmain.trains;
    b->lut_prior = lut_prior;
    b->lut_ring = lut_prior && lut_ring;
    // the code objects this batch will launch from, loaded now rather than inside its first launch
    if (ctx) {
        if (packed) agx_phmm_pk_preload();
        if (packed && pmain.trains) agx_phmm_pk_train_preload();
        agx_phmm_scalar_preload();
        if (lut_prior) agx_phmm_lut_preload();
        agx_copy_preload();
        if (precision == AGX_PHMM_F32 || precision == AGX_PHMM_F32_FMA) agx_phmm_finish_preload();
    }
    b->separate_rescue = packed;
    b->n_pairs = n_pairs;
    b->main.launches = pmain.launches;
    if (packed) b->rescue_seed = std::move(seed_holder); // (`seed` stays valid: the batch owns it now)
    const size_t groups_bytes = packed ? pmain.groups2.size() * sizeof(PhGroup2) : pmain.groups1.size() * sizeof(PhGroup);
    b->info.n_pairs = n_pairs;
    b->info.cells = cells;
    b->info.padded_cells = padded;
    b->info.input_bytes = (int64_t)(img_dw * 4 + groups_bytes + pmain.tabs.size() * sizeof(PhTab) +
                                    pmain.waves.size() * sizeof(PhWave));
    const bool two_pass = precision == AGX_PHMM_F32 || precision == AGX_PHMM_F32_FMA;
    // (a packed batch's rescue plan is not counted: it is launched only when a fill underflowed)
    b->info.n_launches = (int32_t)(pmain.launches.size() + (!packed && two_pass ? pmain.launches.size() : 0));
    b->info.n_waves = (int32_t)(pmain.waves.size() + pstripe.waves.size());
    if (!pstripe.waves.empty()) {
        ClassLaunch cl;
        cl.C = stripe_cols();
        cl.n_waves = (uint32_t)pstripe.waves.size();
        cl.lds = cl.lds_rescue = stripe_lds;
        b->stripe.launches.push_back(cl);
        b->info.n_launches += 1;
        b->stripe_mis_div = stripe_mis_div;
        b->stripe_rows = (stripe_steps + 128u + 63u) & ~63u; // a 64-row block may start at the last step, 63 rows ahead
        b->stripe_grid = std::min<uint32_t>(cl.n_waves, (uint32_t)n_cu * 8u);
    }
    if (!ctx) { // planning only
        *out = b;
        b = nullptr;
        return AGX_OK;
    }
    // Everything goes through one pinned staging block and the context's copy stream: one DMA per array,
    // one synchronisation at the end (create is blocking by contract).
    struct Piece {
        DevBuf *dst;
        const void *src;
        size_t bytes;
    };
    struct Lut {
        double d[256];
        float f[256];
        double mis_d[256];
        float mis_f[256];
    } lut; // layout the launch code relies on: [lut_d][lut_f][mis_d][mis_f]
    build_lut(gatk_prior, lut.d, lut.f, lut.mis_d, lut.mis_f);
    std::vector<Piece> pieces;
    pieces.push_back(Piece{&b->img, nullptr, img_dw * 4}); // already in pinned memory (img_pin)
    pieces.push_back(Piece{&b->main.groups, packed ? (const void *)pmain.groups2.data() : (const void *)pmain.groups1.data(), groups_bytes});
    pieces.push_back(Piece{&b->main.tabs, pmain.tabs.data(), pmain.tabs.size() * sizeof(PhTab)});
    pieces.push_back(Piece{&b->main.waves, pmain.waves.data(), pmain.waves.size() * sizeof(PhWave)});
    if (!pstripe.waves.empty()) {
        pieces.push_back(Piece{&b->stripe.groups, pstripe.groups1.data(), pstripe.groups1.size() * sizeof(PhGroup)});
        pieces.push_back(Piece{&b->stripe.tabs, pstripe.tabs.data(), pstripe.tabs.size() * sizeof(PhTab)});
        pieces.push_back(Piece{&b->stripe.waves, pstripe.waves.data(), pstripe.waves.size() * sizeof(PhWave)});
    }
    pieces.push_back(Piece{&b->lut, &lut, sizeof lut});
    size_t stage_bytes = 0;
    for (const Piece &pc : pieces)
        if (pc.src) stage_bytes += (pc.bytes + 255) & ~(size_t)255;
    PinBuf stage;
    struct StageGuard {
        PinBuf &s;
        ~StageGuard() { s.release(); }
    } stage_guard{stage};
    rc = stage.alloc(ctx, stage_bytes);
    for (const Piece &pc : pieces)
        if (!rc) rc = pc.dst->alloc(ctx, pc.bytes);
    if (!rc && !pstripe.waves.empty()) rc = b->stripe_scratch.alloc(ctx, (size_t)b->stripe_grid * 6u * b->stripe_rows * sizeof(double));
    if (!rc) rc = b->sums.alloc(ctx, ((size_t)n_pairs + 1) * sizeof(double)); // +1: spare slot of vacant packed halves
    if (!rc) rc = b->counter.alloc(ctx, 2 * sizeof(unsigned long long)); // [0] rescued, [1] below the float range
    if (!rc) rc = b->out_stage.alloc(ctx, 2 * (size_t)n_pairs * sizeof(double) + 16);
    if (!rc && b->info.n_launches > 1) rc = agx_ctx_prepare_fanout(ctx);
    if (rc) return rc;
    hipStream_t cs = ctx->copy;
    hipError_t e = hipSuccess;
    {
        size_t at = 0;
        for (const Piece &pc : pieces) {
            if (!pc.src) { // the image: built in its own pinned block
                if (e == hipSuccess && pc.bytes) e = hipMemcpyAsync(pc.dst->p, img, pc.bytes, hipMemcpyHostToDevice, cs);
                continue;
            }
            if (pc.bytes) memcpy((char *)stage.p + at, pc.src, pc.bytes);
            if (e == hipSuccess && pc.bytes) e = hipMemcpyAsync(pc.dst->p, (char *)stage.p + at, pc.bytes, hipMemcpyHostToDevice, cs);
            at += (pc.bytes + 255) & ~(size_t)255;
        }
    }
    if (e == hipSuccess && !pstripe.waves.empty()) e = hipMemsetAsync(b->stripe_scratch.p, 0, b->stripe_scratch.bytes, cs);
    if (e == hipSuccess) e = hipMemsetAsync(b->sums.p, 0, b->sums.bytes, cs); // degenerate pairs keep sum 0
    if (e == hipSuccess) e = hipMemsetAsync(b->counter.p, 0, b->counter.bytes, cs);
    // The fast cell's table rows -- two float4 of derived constants per read position, several double divisions each --
    // are the same in every wave that uses a read and in every launch of the batch: made here, once (phmm_pk_rows), the
    // waves copy them into LDS instead of deriving them (3 % of config 3's launch).
    PinBuf h_reads;
    struct ReadsGuard {
        PinBuf &p;
        ~ReadsGuard() { p.release(); }
    } reads_guard{h_reads};
    if (e == hipSuccess && b->fast && !probs && n_reads && !agx_tune("AGX_PHMM_NO_ROWS")) {
        const size_t n_rows = (reads_end_dw - zero_dw) / 5 * 4;
        rc = b->pk_rows.alloc(ctx, std::max<size_t>(n_rows, 1) * 32);
        DevBuf d_reads;
        if (!rc) rc = h_reads.alloc(ctx, (size_t)n_reads * sizeof(PhTab));
        if (!rc) rc = d_reads.alloc(ctx, (size_t)n_reads * sizeof(PhTab));
        if (rc) {
            d_reads.release();
            return rc;
        }
        PhTab *hr = (PhTab *)h_reads.p;
        for (uint32_t r = 0; r < n_reads; ++r) hr[r] = PhTab{read_dw[r], (uint32_t)(d->read_off[r + 1] - d->read_off[r])};
        b->rows_base_dw = (uint32_t)zero_dw;
        e = hipMemcpyAsync(d_reads.p, h_reads.p, (size_t)n_reads * sizeof(PhTab), hipMemcpyHostToDevice, cs);
        const void *lut_f = (const char *)b->lut.p + 256 * sizeof(double);
        const void *mis_f = (const char *)b->lut.p + 256 * (2 * sizeof(double) + sizeof(float));
        if (e == hipSuccess && agx_phmm_pk_rows_launch((const uint32_t *)b->img.p, (const PhTab *)d_reads.p, n_reads, lut_f, gatk_prior ? mis_f : nullptr,
                                                       b->pk_rows.p, b->rows_base_dw, cs))
            e = hipErrorLaunchFailure;
        if (e == hipSuccess) e = hipStreamSynchronize(cs);
        d_reads.release();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(cs);
    if (e != hipSuccess) {
        agx_set_error("agx_phmm_batch_create: upload -> %s", hipGetErrorString(e));
        return AGX_E_HIP;
    }
    if (trace)
        fprintf(stderr, "[agx_phmm_batch_create] %lld pairs: plan+order %.2f ms, waves+pack %.2f ms, alloc+H2D %.2f ms (%.1f MB)\n",
                (long long)n_pairs, t_plan - t_begin, t_pack - t_plan, now() - t_pack, img_dw * 4 / 1e6);
    *out = b;
    b = nullptr;
    return AGX_OK;
}

} // namespace

extern "C" {

void agx_phmm_batch_destroy(agx_phmm_batch *b)
{
    if (!b) return;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    b->img.release();
    b->stripe_scratch.release();
    for (agx_phmm_batch::DevPlan *pl : {&b->main, &b->rescue, &b->stripe}) {
        pl->groups.release();
        pl->tabs.release();
        pl->waves.release();
    }
    b->sums.release();
    b->out_stage.release();
    b->bound_flag.release();
    b->pk_rows.release();
    b->lut.release();
    b->counter.release();
    agx_ctx_release(b->ctx); // the batch's own reference: a context outlives its batches
    delete b;
}

int agx_phmm_batch_create(agx_ctx *ctx, const agx_phmm_desc *d, int precision, agx_phmm_batch **out)
{
    AGX_GUARD_BEGIN
    return create_batch(ctx, d, nullptr, precision, out);
    AGX_GUARD_END("agx_phmm_batch_create")
}

int agx_phmm_batch_launch(agx_phmm_batch *b)
{
    if (!b) {
        agx_set_error("agx_phmm_batch_launch: null batch");
        return AGX_E_ARG;
    }
    if (!b->ctx) {
        agx_set_error("this batch was planned without a context (no device): it cannot be launched");
        return AGX_E_NODEVICE;
    }
    int rc = agx_bind(b->ctx);
    if (rc) return rc;
    hipStream_t s = b->ctx->stream;
    const void *lut_d = b->lut.p;
    const void *lut_f = (const char *)b->lut.p + 256 * sizeof(double);
    const void *mis_d = (const char *)b->lut.p + 256 * (sizeof(double) + sizeof(float));
    const void *mis_f = (const char *)b->lut.p + 256 * (2 * sizeof(double) + sizeof(float));
    const bool f32_family = b->precision == AGX_PHMM_F32 || b->precision == AGX_PHMM_F32_FMA;
    // The two counters are zero at creation and are reset by whoever reads them (the log10 kernel of agx_phmm_batch_results; a
    // bound batch whose flag stayed clear has counted nothing): only a launch whose predecessor's counts nobody took resets
    // them itself.
    if (f32_family && b->counters_dirty) AGX_HIP(hipMemsetAsync(b->counter.p, 0, 2 * sizeof(unsigned long long), s));
    b->counters_dirty = f32_family;
    b->rescue_pending = b->separate_rescue;
    if (b->bound) *(volatile unsigned *)b->bound_flag.p = 0; // set by the fill when a pair goes to the rescue plan
    const void *mis_for_d = b->gatk_prior ? mis_d : nullptr, *mis_for_f = b->gatk_prior ? mis_f : nullptr;
    auto launch_scalar = [&](const agx_phmm_batch::DevPlan &pl, const ClassLaunch &cl, int mode, hipStream_t st) -> int {
        const bool f64 = mode != 2;
        const size_t lds = mode == 3 && !b->separate_rescue ? cl.lds_rescue : cl.lds; // same records, wider table rows
        const int r = agx_phmm_launch_class(mode, cl.C, cl.all_g16, (const uint32_t *)b->img.p, (const PhGroup *)pl.groups.p,
                                            (const PhTab *)pl.tabs.p, (const PhWave *)pl.waves.p + cl.first_wave, cl.n_waves,
                                            f64 ? lut_d : lut_f, f64 ? mis_for_d : mis_for_f, (double *)b->sums.p,
                                            (double)AGX_PHMM_F32_RESCUE, (unsigned long long *)b->counter.p, lds, st);
        if (r) {
            agx_set_error("phmm_fill<C=%d, mode %d> launch failed: %s", cl.C, mode, hipGetErrorString(hipGetLastError()));
            return AGX_E_HIP;
        }
        return AGX_OK;
    };
    // the counter reset above is ordered before the fork; different classes run side by side
    {
        FanOut fan(b->ctx, (int)(b->main.launches.size() + b->stripe.launches.size()));
        rc = fan.begin();
        if (rc) return rc;
        int k = 0;
        for (const ClassLaunch &cl : b->stripe.launches) { // longest-running waves first
            const bool fma = b->precision == AGX_PHMM_F64_FMA || b->precision == AGX_PHMM_F32_FMA;
            const int mode = b->probs ? 4 : fma ? 1 : 0;
            // a float batch's long pairs are computed in double: stored negated like its rescued pairs
            const int r = agx_phmm_stripe_launch(mode, cl.C, (const uint32_t *)b->img.p, (const PhGroup *)b->stripe.groups.p,
                                                 (const PhTab *)b->stripe.tabs.p, (const PhWave *)b->stripe.waves.p, cl.n_waves,
                                                 b->stripe_grid, lut_d, b->stripe_mis_div ? nullptr : mis_for_d, b->stripe_mis_div ? 1 : 0, (double *)b->sums.p,
                                                 (double *)b->stripe_scratch.p, b->stripe_rows, f32_family ? 1 : 0, cl.lds,
                                                 fan.stream(k++));
            if (r) {
                agx_set_error("phmm_fill_striped launch failed: %s", hipGetErrorString(hipGetLastError()));
                return AGX_E_HIP;
            }
        }
        for (const ClassLaunch &cl : b->main.launches) {
            hipStream_t st = fan.stream(k++);
            if (b->packed) {
                const PhUnderflow uf = PhUnderflow{(double)AGX_PHMM_F32_RESCUE, (uint32_t)b->n_pairs, (unsigned long long *)b->counter.p + 1,
                                                                   (double)(FLT_MAX / 16), (float)((b->gatk_prior ? kGuardGatk : kGuardRef) * 3.3219280948873623),
                                                                   b->bound, b->bound ? (unsigned *)b->bound_flag.p : nullptr, log10((double)(FLT_MAX / 16)),
                                                                   b->pk_rows.p, b->rows_base_dw};
                const int r = b->trains ? agx_phmm_pk_train_launch_class(cl.C, cl.all_g16, (const uint32_t *)b->img.p, (const PhGroup2 *)b->main.groups.p,
                                                                         (const PhTab *)b->main.tabs.p, (const PhWave *)b->main.waves.p + cl.first_wave,
                                                                         cl.n_waves, lut_f, mis_for_f, (double *)b->sums.p, uf, cl.lds, st)
                                        : agx_phmm_pk_launch_class(cl.C, cl.all_g16, b->fast, (const uint32_t *)b->img.p, (const PhGroup2 *)b->main.groups.p,
                                                                   (const PhTab *)b->main.tabs.p, (const PhWave *)b->main.waves.p + cl.first_wave,
                                                                   cl.n_waves, lut_f, mis_for_f, (double *)b->sums.p, uf, cl.lds, st);
                if (r) {
                    agx_set_error("phmm_fill_pk<C=%d> launch failed: %s", cl.C, hipGetErrorString(hipGetLastError()));
                    return AGX_E_HIP;
                }
            } else if (b->lut_prior) {
                const int r = agx_phmm_lut_launch_class(b->precision == AGX_PHMM_F64_FMA, cl.C, cl.all_g16, (const uint32_t *)b->img.p,
                                                        (const PhGroup *)b->main.groups.p, (const PhTab *)b->main.tabs.p,
                                                        (const PhWave *)b->main.waves.p + cl.first_wave, cl.n_waves, lut_d, mis_for_d,
                                                        (double *)b->sums.p, cl.lds, !agx_tune("AGX_PHMM_LUT_ONE_LOOP"), b->lut_ring, st);
                if (r) {
                    agx_set_error("phmm_fill_lut<C=%d> launch failed: %s", cl.C, hipGetErrorString(hipGetLastError()));
                    return AGX_E_HIP;
                }
            } else {
                int mode = b->precision;
                if (b->probs) mode = 4;
                rc = launch_scalar(b->main, cl, mode, st);
                // AGX_PHMM_F32: the double recomputation of underflowed pairs reuses the class's records
                if (!rc && b->precision == AGX_PHMM_F32) rc = launch_scalar(b->main, cl, 3, st);
                if (rc) return rc;
            }
        }
        rc = fan.end();
        if (rc) return rc;
    }
    return AGX_OK;
}

// The double rescue plan of a packed float batch (its own records over the same pairs): every wave looks at its pairs'
// float sums and recomputes those below the float range.  Run from agx_phmm_batch_results, when the fill counted any.
static int launch_rescue_plan(agx_phmm_batch *b)
{
    if (b->rescue_broken) {
        agx_set_error("agx_phmm_batch_results: the double rescue pass of this batch could not be planned earlier; create the batch again");
        return AGX_E_NOMEM;
    }
    if (b->rescue_seed) { // first underflow of this batch: plan the double pass over the same pairs and upload its records
        struct Broken { // any way out of this block but its end leaves the batch without the plan it needs
            agx_phmm_batch *b;
            bool done = false;
            ~Broken() { b->rescue_broken = !done; }
        } broken{b};
        PlanOut pr;
        const bool trace = agx_tune("AGX_TRACE_CREATE") != nullptr;
        std::unique_ptr<PlanSeed> seed = std::move(b->rescue_seed);
        int rc = make_plan(*seed, std::move(seed->gen0), AGX_PHMM_F64, 1, true, false, trace, 1, pr);
        if (rc) return rc;
        struct Piece {
            DevBuf *dst;
            const void *src;
            size_t bytes;
        } pieces[3] = {{&b->rescue.groups, pr.groups1.data(), pr.groups1.size() * sizeof(PhGroup)},
                       {&b->rescue.tabs, pr.tabs.data(), pr.tabs.size() * sizeof(PhTab)},
                       {&b->rescue.waves, pr.waves.data(), pr.waves.size() * sizeof(PhWave)}};
        size_t stage_bytes = 0;
        for (const Piece &pc : pieces) stage_bytes += (pc.bytes + 255) & ~(size_t)255;
        PinBuf stage;
        struct StageGuard {
            PinBuf &s;
            ~StageGuard() { s.release(); }
        } stage_guard{stage};
        rc = stage.alloc(b->ctx, stage_bytes);
        for (const Piece &pc : pieces)
            if (!rc) rc = pc.dst->alloc(b->ctx, pc.bytes);
        if (!rc && pr.launches.size() > 1) rc = agx_ctx_prepare_fanout(b->ctx);
        if (rc) return rc;
        hipStream_t cs = b->ctx->copy;
        size_t at = 0;
        for (const Piece &pc : pieces) {
            if (pc.bytes) {
                memcpy((char *)stage.p + at, pc.src, pc.bytes);
                AGX_HIP(hipMemcpyAsync(pc.dst->p, (char *)stage.p + at, pc.bytes, hipMemcpyHostToDevice, cs));
            }
            at += (pc.bytes + 255) & ~(size_t)255;
        }
        AGX_HIP(hipStreamSynchronize(cs));
        b->rescue.launches = pr.launches;
        broken.done = true;
    }
    const void *lut_d = b->lut.p;
    const void *mis_d = (const char *)b->lut.p + 256 * (sizeof(double) + sizeof(float));
    FanOut fan(b->ctx, (int)b->rescue.launches.size());
    int rc = fan.begin();
    if (rc) return rc;
    int k = 0;
    for (const ClassLaunch &cl : b->rescue.launches) {
        const int r = agx_phmm_launch_class(3, cl.C, cl.all_g16, (const uint32_t *)b->img.p, (const PhGroup *)b->rescue.groups.p,
                                            (const PhTab *)b->rescue.tabs.p, (const PhWave *)b->rescue.waves.p + cl.first_wave, cl.n_waves,
                                            lut_d, b->gatk_prior ? mis_d : nullptr, (double *)b->sums.p, (double)AGX_PHMM_F32_RESCUE,
                                            (unsigned long long *)b->counter.p, cl.lds, fan.stream(k++));
        if (r) {
            agx_set_error("phmm_fill<C=%d, rescue> launch failed: %s", cl.C, hipGetErrorString(hipGetLastError()));
            return AGX_E_HIP;
        }
    }
    return fan.end();
}

int agx_phmm_batch_results(agx_phmm_batch *b, double *log10_lik, double *raw_sum)
{
    AGX_GUARD_BEGIN
    if (!b || (!log10_lik && b->n_pairs)) {
        agx_set_error("agx_phmm_batch_results: null argument");
        return AGX_E_ARG;
    }
    if (!b->ctx) {
        agx_set_error("this batch was planned without a context (no device): it has no results");
        return AGX_E_NODEVICE;
    }
    int rc = agx_bind(b->ctx);
    if (rc) return rc;
    if (b->bound && log10_lik == b->bound && !raw_sum) {
        // bound results: the fill wrote them into this array itself; only a pair sent to the rescue plan (underflow,
        // accuracy guard) makes the long way below necessary
        AGX_HIP(hipStreamSynchronize(b->ctx->stream));
        if (*(volatile unsigned *)b->bound_flag.p == 0) {
            b->rescue_pending = false;
            b->info.n_rescued = 0;
            b->counters_dirty = false; // no pair was counted: the counters are still zero
            return AGX_OK;
        }
    }
    // through pinned staging on the launch stream: the DMAs queue right behind the last kernel
    const size_t sum_bytes = (size_t)b->n_pairs * sizeof(double);
    const bool f32 = b->precision == AGX_PHMM_F32 || b->precision == AGX_PHMM_F32_FMA;
    // antidiagsPairHMM.c:242 -- both logarithms in double: by the host libm for the double modes (bit-identical to
    // the reference's output), by the device for the float modes (agx_phmm_finish_kernel.hip)
    const double c64 = log10(DBL_MAX / 16), c32 = log10((double)(FLT_MAX / 16));
    const bool want_sums = raw_sum != nullptr || !f32;
    char *const stage = (char *)b->out_stage.p; // [two counters][logs (float modes)][sums (when wanted)]
    volatile unsigned long long *host_counter = (volatile unsigned long long *)stage; // written by the log10 kernel itself
    double *s = nullptr, *dev_logs = nullptr;                                         // (float modes; the double modes have no rescue pass)
    hipStream_t st = b->ctx->stream;
    for (int round = 0; round < 2; ++round) {
        host_counter[0] = host_counter[1] = 0;
        char *at = stage + 16;
        if (f32 && b->n_pairs) {
            // the log10 kernel stores straight into page-locked host memory -- the caller's array when that is
            // page-locked (agx_host_alloc), else the staging block: consecutive 8-byte stores, no D2H copy behind it
            dev_logs = agx_is_pinned_host(log10_lik, sum_bytes) ? log10_lik : (double *)at;
            if (agx_phmm_finish_launch((const double *)b->sums.p, dev_logs, (uint32_t)b->n_pairs, c64, c32,
                                       (unsigned long long *)b->counter.p, (unsigned long long *)stage, st)) {
                agx_set_error("phmm_finish launch failed: %s", hipGetErrorString(hipGetLastError()));
                return AGX_E_HIP;
            }
            at += sum_bytes;
        }
        if (want_sums) {
            s = (double *)at;
            if (b->n_pairs && agx_copy_out_launch(b->sums.p, s, sum_bytes, st)) {
                agx_set_error("agx_phmm_batch_results: copy kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
                return AGX_E_HIP;
            }
            at += sum_bytes;
        }
        AGX_HIP(hipStreamSynchronize(st));
        // a packed float batch launches its double rescue plan only now, and only when the fill counted a pair below
        // the float range (config 3: never) -- then the results are taken a second time
        if (!(b->rescue_pending && host_counter[1] != 0)) break;
        b->rescue_pending = false;
        rc = launch_rescue_plan(b);
        if (rc) return rc;
    }
    b->rescue_pending = false;
    if (f32 && b->n_pairs) {
        if (b->counters_dirty) b->info.n_rescued = (int64_t)host_counter[0]; // (a second fetch without a launch in between finds them reset: the figure stays)
        b->counters_dirty = false;
    } else
        b->info.n_rescued = (int64_t)host_counter[0];
    if (f32) {
        if (b->n_pairs && dev_logs != log10_lik) memcpy(log10_lik, dev_logs, sum_bytes);
    } else
        agx_parallel_for(b->n_pairs, 2048, [&](int64_t lo, int64_t hi, int) {
            for (int64_t k = lo; k < hi; ++k) log10_lik[k] = log10(s[k]) - c64;
        });
    if (raw_sum && b->n_pairs) {
        if (f32) // the rescue pass stores its double-scaled sum negated so the two scalings stay apart: callers get |sum|
            for (int64_t k = 0; k < b->n_pairs; ++k) raw_sum[k] = s[k] < 0 ? -s[k] : s[k];
        else
            memcpy(raw_sum, s, sum_bytes);
    }
    return AGX_OK;
    AGX_GUARD_END("agx_phmm_batch_results")
}

int agx_phmm_batch_bind_results(agx_phmm_batch *b, double *log10_lik)
{
    if (!b) {
        agx_set_error("agx_phmm_batch_bind_results: null batch");
        return AGX_E_ARG;
    }
    if (!b->ctx) {
        agx_set_error("this batch was planned without a context (no device)");
        return AGX_E_NODEVICE;
    }
    if (log10_lik && !agx_is_pinned_host(log10_lik, (size_t)std::max<int64_t>(b->n_pairs, 1) * sizeof(double))) {
        agx_set_error("agx_phmm_batch_bind_results: the array is not page-locked memory of agx_host_alloc (or too short for %lld results)", (long long)b->n_pairs);
        return AGX_E_ARG;
    }
    int rc = agx_bind(b->ctx);
    if (rc) return rc;
    AGX_HIP(hipStreamSynchronize(b->ctx->stream)); // launches in flight still write the old destination
    // taken by packed float batches whose records are in output order (one (R, H) shape, every pair with work); a hint otherwise
    const bool take = log10_lik && b->packed && b->file_order && b->n_pairs > 0;
    if (take && !b->bound_flag.p) {
        rc = b->bound_flag.alloc(b->ctx, 64);
        if (rc) return rc;
    }
    b->bound = take ? log10_lik : nullptr;
    return AGX_OK;
}

int agx_phmm_batch_info(const agx_phmm_batch *b, agx_phmm_info *info)
{
    if (!b || !info) {
        agx_set_error("agx_phmm_batch_info: null argument");
        return AGX_E_ARG;
    }
    *info = b->info;
    return AGX_OK;
}

int agx_phmm_forward(agx_ctx *ctx, const agx_phmm_desc *d, int precision, double *log10_lik)
{
    AGX_GUARD_BEGIN
    // A large batch goes through in pieces of whole regions (cut by cells, like the shards of agx_phmm_forward_devices):
    // piece k + 1 is enumerated, planned and uploaded while piece k is being filled, so the host's 3-4 ms for config 5's
    // 262 144 pairs hide behind the 12 ms of fills instead of standing in front of them.
    int pieces = 1;
    if (ctx && d && d->n_regions >= 2 && d->region_read && d->region_hap && d->read_off && d->hap_off) {
        double cells = 0;
        for (uint32_t g = 0; g < d->n_regions; ++g) {
            const uint32_t r0 = d->region_read[g], r1 = d->region_read[g + 1], h0 = d->region_hap[g], h1 = d->region_hap[g + 1];
            if (r1 < r0 || h1 < h0 || r1 > d->n_reads || h1 > d->n_haps) {
                cells = 0; // malformed: let the one-batch path report it
                break;
            }
            cells += (double)(d->read_off[r1] - d->read_off[r0]) * (double)(d->hap_off[h1] - d->hap_off[h0]);
        }
        pieces = (int)std::min<double>({8.0, cells / kPhmmPieceCells, (double)d->n_regions});
        if (pieces < 2) pieces = 1;
    }
    if (pieces == 1) {
        agx_phmm_batch *b = nullptr;
        int rc = agx_phmm_batch_create(ctx, d, precision, &b);
        if (rc) return rc;
        rc = agx_phmm_batch_launch(b);
        if (!rc) rc = agx_phmm_batch_results(b, log10_lik, nullptr);
        agx_phmm_batch_destroy(b);
        return rc;
    }
    std::vector<uint32_t> cut((size_t)pieces + 1);
    int rc = agx_phmm_shard_cuts(d, pieces, cut.data());
    if (rc) return rc;
    std::vector<agx_phmm_batch *> bs((size_t)pieces, nullptr);
    struct Cleanup {
        std::vector<agx_phmm_batch *> &v;
        ~Cleanup()
        {
            for (agx_phmm_batch *b : v) agx_phmm_batch_destroy(b);
        }
    } cleanup{bs};
    std::vector<int64_t> first_out((size_t)pieces + 1, 0);
    for (int k = 0; k < pieces; ++k) {
        int64_t n = 0;
        for (uint32_t g = cut[(size_t)k]; g < cut[(size_t)k + 1]; ++g)
            n += (int64_t)(d->region_read[g + 1] - d->region_read[g]) * (d->region_hap[g + 1] - d->region_hap[g]);
        first_out[(size_t)k + 1] = first_out[(size_t)k] + n;
        if (cut[(size_t)k + 1] <= cut[(size_t)k]) continue;
        agx_phmm_desc sub = *d;
        sub.region_read = d->region_read + cut[(size_t)k]; // absolute read / haplotype indices stay valid
        sub.region_hap = d->region_hap + cut[(size_t)k];
        sub.n_regions = cut[(size_t)k + 1] - cut[(size_t)k];
        rc = agx_phmm_batch_create(ctx, &sub, precision, &bs[(size_t)k]);
        if (!rc) rc = agx_phmm_batch_launch(bs[(size_t)k]);
        if (rc) return rc;
    }
    for (int k = 0; k < pieces; ++k) {
        if (!bs[(size_t)k]) continue;
        rc = agx_phmm_batch_results(bs[(size_t)k], log10_lik + first_out[(size_t)k], nullptr);
        if (rc) return rc;
        agx_phmm_batch_destroy(bs[(size_t)k]);
        bs[(size_t)k] = nullptr;
    }
    return AGX_OK;
    AGX_GUARD_END("agx_phmm_forward")
}

int agx_phmm_shard_cuts(const agx_phmm_desc *d, int n_shards, uint32_t *cut)
{
    if (!d || n_shards < 1 || !cut || (d->n_regions && (!d->region_read || !d->region_hap || !d->read_off || !d->hap_off))) {
        agx_set_error("agx_phmm_shard_cuts: bad arguments");
        return AGX_E_ARG;
    }
    // whole regions stay together; contiguous shards balanced by cells
    const uint32_t ng = d->n_regions;
    for (int k = 0; k <= n_shards; ++k) cut[k] = ng;
    cut[0] = 0;
    double total = 0;
    for (uint32_t g = 0; g < ng; ++g) {
        const uint32_t r0 = d->region_read[g], r1 = d->region_read[g + 1], h0 = d->region_hap[g], h1 = d->region_hap[g + 1];
        if (r1 < r0 || h1 < h0 || r1 > d->n_reads || h1 > d->n_haps) {
            agx_set_error("region %u: ranges out of order or out of bounds", g);
            return AGX_E_ARG;
        }
        total += (double)(d->read_off[r1] - d->read_off[r0]) * (double)(d->hap_off[h1] - d->hap_off[h0]) + 1.0;
    }
    double acc = 0;
    int k = 1;
    for (uint32_t g = 0; g < ng && k < n_shards; ++g) {
        const uint32_t r0 = d->region_read[g], r1 = d->region_read[g + 1], h0 = d->region_hap[g], h1 = d->region_hap[g + 1];
        acc += (double)(d->read_off[r1] - d->read_off[r0]) * (double)(d->hap_off[h1] - d->hap_off[h0]) + 1.0;
        while (k < n_shards && acc >= total * k / n_shards) cut[k++] = g + 1;
    }
    return AGX_OK;
}

int agx_phmm_forward_devices(const int *devices, int n_devices, const agx_phmm_desc *d, int precision, double *log10_lik)
{
    AGX_GUARD_BEGIN
    const int avail = agx_device_count();
    if (avail <= 0) {
        agx_set_error("no HIP device is visible (this library has no CPU fallback)");
        return AGX_E_NODEVICE;
    }
    if (!devices || n_devices < 1 || n_devices > 1024 || !d ||
        (d->n_regions && (!d->region_read || !d->region_hap || !d->read_off || !d->hap_off || !log10_lik))) {
        agx_set_error("agx_phmm_forward_devices: bad arguments");
        return AGX_E_ARG;
    }
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= avail) {
            agx_set_error("agx_phmm_forward_devices: device %d out of range [0,%d)", devices[k], avail);
            return AGX_E_NODEVICE;
        }
    std::vector<uint32_t> cut((size_t)n_devices + 1);
    int rc = agx_phmm_shard_cuts(d, n_devices, cut.data());
    if (rc) return rc;
    const uint32_t ng = d->n_regions;
    std::vector<int64_t> first_out((size_t)ng + 1, 0);
    for (uint32_t g = 0; g < ng; ++g)
        first_out[g + 1] = first_out[g] + (int64_t)(d->region_read[g + 1] - d->region_read[g]) * (d->region_hap[g + 1] - d->region_hap[g]);
    std::vector<int> rcs((size_t)n_devices, AGX_OK), slot((size_t)n_devices, 0);
    for (int k = 0; k < n_devices; ++k) // shards sharing a device get contexts of their own
        for (int j = 0; j < k; ++j) slot[(size_t)k] += devices[j] == devices[k];
    std::vector<std::string> errs((size_t)n_devices);
    auto shard = [&](int k) {
        const uint32_t lo = cut[(size_t)k], hi = cut[(size_t)k + 1];
        if (hi <= lo) return;
        int r;
        try {
            agx_phmm_desc sub = *d;
            sub.region_read = d->region_read + lo; // absolute read/hap indices stay valid
            sub.region_hap = d->region_hap + lo;
            sub.n_regions = hi - lo;
            agx_ctx *c = nullptr;
            std::mutex *busy = nullptr;
            r = agx_shared_ctx(devices[k], slot[(size_t)k], &c, &busy); // created once per process: pools stay warm
            if (!r) {
                std::lock_guard<std::mutex> turn(*busy); // concurrent callers take turns on this (device, slot)
                r = agx_phmm_forward(c, &sub, precision, log10_lik + first_out[lo]);
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
    AGX_GUARD_END("agx_phmm_forward_devices")
}

int agx_phmm_forward_multi(int n_devices, const agx_phmm_desc *d, int precision, double *log10_lik)
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
    return agx_phmm_forward_devices(devs, n_devices, d, precision, log10_lik);
}

void agx_pairHMM(double *likelihood, double *M, double *X, double *Y, char *R, char *H, int read_len, int haplotype_len,
                 double *Qr, double *Qi, double *Qd, double *Qg)
{
    (void)M;
    (void)X;
    (void)Y; // the reference's rolling anti-diagonal scratch (antidiagsPairHMM.c:452-455): unused here
    static std::mutex mu;
    agx_ctx *ctx = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!likelihood) return;
    *likelihood = NAN;
    if (read_len < 0 || haplotype_len < 0 || !R || !H || !Qr || !Qi || !Qd || !Qg) {
        agx_set_error("agx_pairHMM: bad arguments");
        return;
    }
    try {
    if (agx_shared_ctx(0, 0, &ctx) != AGX_OK) return; // process-wide, created on first use (this function's own mutex covers its use)
    const uint64_t roff[2] = {0, (uint64_t)read_len}, hoff[2] = {0, (uint64_t)haplotype_len};
    const uint32_t reg[2] = {0, 1};
    agx_phmm_desc d{};
    d.read_bases = reinterpret_cast<const uint8_t *>(R);
    d.read_off = roff;
    d.n_reads = 1;
    d.hap_bases = reinterpret_cast<const uint8_t *>(H);
    d.hap_off = hoff;
    d.n_haps = 1;
    d.region_read = reg;
    d.region_hap = reg;
    d.n_regions = 1;
    const double *prob[4] = {Qr, Qi, Qd, Qg};
    agx_phmm_batch *b = nullptr;
    if (create_batch(ctx, &d, prob, AGX_PHMM_F64, &b) != AGX_OK) return;
    double v = NAN;
    if (agx_phmm_batch_launch(b) == AGX_OK && agx_phmm_batch_results(b, &v, nullptr) == AGX_OK) *likelihood = v;
    agx_phmm_batch_destroy(b);
    } catch (const std::exception &ex) {
        agx_set_error("agx_pairHMM: %s", ex.what());
    }
}

} // extern "C"
