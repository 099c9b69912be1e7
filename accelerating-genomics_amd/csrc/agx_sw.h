// Device work records of the Smith-Waterman fill (shared by scheduler and kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

// One alignment pair = one group of G lanes.  Offsets are in 4-byte words into the packed
// image: x (the shorter sequence, zero-padded to G*C bytes rounded up to 4, plus one spare word),
// y (the longer, padded to 4).
struct SwGroup {
    uint32_t x_dw;
    uint32_t y_dw;
    uint32_t lx_ly; // lx | (1 = the pair's SECOND sequence is the shorter one) << 15 | ly << 16; the fills read ly only
    uint32_t out;   // index into scores[] = the caller's pair number
};

// Scoring, as the kernels consume it (built by the host from agx_sw_scoring).  With z = H + gf the
// state both gap recurrences read, a cell is  e = max(z_up, e + ge),  f = max(z_left, f + ge),
// s = H_diag + match - {0, delta},  H = max(e, f, s, 0),  z = H + gf.
struct SwParams {
    int32_t ge;      // gap extend (<= 0): every further gap cell
    int32_t gf;      // first gap cell = open + extend (<= 0)
    int32_t hd;      // match - gf: turns z_diag into H_diag + match
    int32_t delta;   // match - mismatch (> 0)
    int32_t shift;   // packed kernel: symbols are compared as byte << shift, 2^shift >= delta
    // the same, replicated into both 16-bit halves for the packed kernels
    uint32_t ge2, gf2, hd2, delta2;
    // biased packed kernel (agx_sw_pk2_kernel.hip): |ge|, |gf| and the bias B added to every stored half
    uint32_t age2, agf2, bias2;
    // slots of the scores array a launch may write (packed kernels: a vacant half points at slot n_pairs, which exists in
    // the device array but not in a caller's page-locked one -- agx_sw_batch_bind_scores)
    uint32_t n_out;
};

// Packed kernel: one group of G lanes carries two pairs (index 0 = low 16 bits, 1 = high 16 bits
// of every state register).  A group without a second pair points [1] at an all-zero sequence
// of length 0 and at the spare score slot scores[n_pairs].
struct SwGroup2 {
    uint32_t x_dw[2];
    uint32_t y_dw[2];
    uint32_t lx_ly[2];
    uint32_t out[2];
};

// One wavefront: n_groups groups of G lanes, all stepping `steps` rows
// (= max(ly) + G - 1 over its groups).
struct SwWave {
    uint32_t first_group;
    uint16_t n_groups;
    uint16_t G;
    uint32_t steps;
    uint32_t reserved; // class word: bits 0..15 columns per lane of this wave's class (read by the one-launch kernel of
                       // mixed batches); bit 16 set on the device by sw_pack_dna: every pair of the wave is DNA-coded
};

// Column-per-lane classes the kernels are instantiated for (any width works: a lane's symbols are
// byte-aligned on load), every even width so that common lengths tile with little padding.
#define AGX_SW_FOR_EACH_CLASS(X) \
    X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)
// wide classes: int32 kernel only, for shorter sides beyond 64 x 40 columns (agx_sw_wide_kernel.hip)
#define AGX_SW_FOR_EACH_WIDE_CLASS(X) X(80) X(120) X(160)
static const int kSwClasses[] = {4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 80, 120, 160};
static const int kSwPackedMaxShort = 64 * 40; // the packed kernel has no wide classes
static const int kSwNumClasses = sizeof(kSwClasses) / sizeof(kSwClasses[0]);
// Measured lane time per padded cell of each class, relative to the widest one (MI355X,
// tools/calibrate_classes.py, profiles/r01_calibration*.log): narrow classes amortise the
// per-step work (DPP shifts, row symbol, loop control) over fewer cells.
static const double kSwClassCost[] = {1.373, 1.250, 1.178, 1.138, 1.112, 1.080, 1.051, 1.033, 1.025, 1.022, 1.014, 1.014, 1.011, 1.007, 1.007, 1.004, 1.004, 1.004, 1.000, 1.6, 2.2, 4.0};
// same for the packed int16 kernel
static const double kSwPkClassCost[] = {1.543, 1.358, 1.278, 1.210, 1.173, 1.136, 1.111, 1.086, 1.068, 1.037, 1.025, 1.025, 1.019, 1.012, 1.006, 1.006, 1.006, 1.000, 1.000, 0, 0, 0}; // 0 = not built
// and for its biased formulation (agx_sw_pk2_kernel.hip, the DNA-coded rising cell with column classes; tools/cal_sw_pk.py,
// profiles/r02h_cal_sw_pk2.log: 0.0951 ps per padded cell at 40 columns)
static const double kSwPk2ClassCost[] = {1.851, 1.581, 1.450, 1.367, 1.318, 1.263, 1.196, 1.184, 1.145, 1.097, 1.095, 1.056, 1.049, 1.053, 1.025, 1.023, 1.022, 1.016, 1.000, 0, 0, 0};

// the 32-bit fill on the packed plan's coded image (agx_sw_i32d_kernel.hip): the int32 kernel's per-class costs, no wide classes
static const double kSwI32dClassCost[] = {1.373, 1.250, 1.178, 1.138, 1.112, 1.080, 1.051, 1.033, 1.025, 1.022, 1.014, 1.014, 1.011, 1.007, 1.007, 1.004, 1.004, 1.004, 1.000, 0, 0, 0};

// substitution-matrix mode: symbol numbers 1..32 in the image, 0 = padding; the device table is
// kSwMatDim x kSwMatDim int16 entries score - (gap_open + gap_extend)
constexpr int kSwMatDim = 33;
int agx_sw_mat_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup *groups,
                            const SwWave *waves, uint32_t n_waves, int32_t *scores, const int16_t *table, hipStream_t s);
// Builds the image the records describe from the caller's raw arrays, on the device (agx_sw_pack_kernel.hip).
// raw = bases[base ..), off = the caller's offsets (absolute), groups = SwGroup (slots 1) or SwGroup2 (slots 2)
// records, code = substitution-matrix byte map or NULL, flag[2] = {offending pairs, smallest of them}.
int agx_sw_pack_launch(bool matrix, int slots, const uint8_t *raw, const uint64_t *off, uint64_t base, const void *groups,
                       uint32_t n_groups, uint32_t n_pairs, uint32_t *img, const uint8_t *code, uint32_t *flag, int n_cu,
                       hipStream_t s);
// The biased packed fill's variant: one pack wavefront per fill wavefront, DNA-codes the waves whose pairs all qualify
// (rewrites their group records' lengths and sets bit 16 of the wave records' class word).
int agx_sw_pack_dna_launch(const uint8_t *raw, const uint64_t *off, uint64_t base, void *groups, void *waves, uint32_t n_waves,
                           uint32_t n_pairs, uint32_t *img, uint32_t *flag, int n_cu, hipStream_t s);
int agx_sw_pk_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups,
                           const SwWave *waves, uint32_t n_waves, int32_t *scores, hipStream_t s);
int agx_sw_pk2_launch_class(int cols_per_lane, int rising, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups,
                            const SwWave *waves, uint32_t n_waves, int32_t *scores, hipStream_t s);
// every class of a mixed batch in one launch: waves[].reserved holds each wave's columns per lane
void agx_sw_pk2_preload();
void agx_sw_pack_preload();
void agx_sw_i32_preload();
int agx_sw_pk2_launch_any(int rising, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves, uint32_t n_waves,
                          int32_t *scores, hipStream_t s);
// the 32-bit fill with the DNA-coded match: the packed plan's records and image, one pair of a lane group at a time
int agx_sw_i32d_launch_any(const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves, uint32_t n_waves, int32_t *scores,
                           hipStream_t s);
int agx_sw_i32d_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves,
                             uint32_t n_waves, int32_t *scores, hipStream_t s);
void agx_sw_i32d_preload();
int agx_sw_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup *groups,
                        const SwWave *waves, uint32_t n_waves, int32_t *scores, hipStream_t s);
int agx_sw_wide_launch_class(int cols_per_lane, const SwParams &prm, const uint32_t *img, const SwGroup *groups,
                             const SwWave *waves, uint32_t n_waves, int32_t *scores, hipStream_t s);

// ---- device-side planning (agx_sw_plan_kernel.hip): the O(pairs) passes of the planner as kernels
constexpr uint32_t kSwPlanEmptyKey = 1u << 27;           // sort key of a pair with an empty side: behind every bucket
constexpr uint32_t kSwPlanWaveKeyMax = (1u << 24) - 1u;  // wave dispatch key = this - steps x columns per lane
constexpr int kSwPlanBuckets = kSwNumClasses * 64;       // bucket id = class index * 64 + (64 - lanes per group)
struct SwPlanArgs {
    const uint32_t *len;        // the caller's len[], on the device
    uint32_t n_pairs, n_fill;   // pairs / pairs with work
    const uint32_t *seg_first;  // tiling table: segments of shorter length lx are [seg_first[lx], seg_first[lx + 1])
    const uint32_t *segs;       // ly_from | class index << 16 | lanes per group << 24
    uint32_t longest;           // longest longer side of the batch
    const uint32_t *buckets;    // kSwPlanBuckets x {first entry, entries, first group, groups, first wave}
    uint32_t img0;              // image word the first block starts at
    int slots;                  // pairs per lane group (2 = packed kernels)
    uint32_t n_waves;
    uint32_t *keys_a, *keys_b, *vals_a, *vals_b;                       // n_pairs words each
    uint32_t *wave_keys_a, *wave_keys_b, *wave_ids_a, *wave_ids_b;   // n_waves words each
    SwWave *waves_tmp, *waves;                                        // n_waves records each; `waves` receives the dispatch order
    uint32_t *groups;                                                 // SwGroup / SwGroup2 records
    unsigned long long *padded;                                       // += padded cells
    void *temp;
    size_t temp_bytes;
    int n_cu;
};
size_t agx_sw_plan_temp_bytes(uint32_t n_pairs, uint32_t n_waves);
int agx_sw_plan_launch(const SwPlanArgs &a, hipStream_t s);
void agx_sw_plan_preload();
