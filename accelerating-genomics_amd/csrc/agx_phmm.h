// Device work records of the PairHMM forward fill (shared by scheduler and kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// One (read, haplotype) pair = one group of G lanes; lane g owns haplotype columns
// [g*C+1, (g+1)*C].  hap_dw: 4-byte-word offset of the haplotype bases in the packed image
// (zero-padded to G*C bytes).  tab: which of the wave's LDS read tables this pair uses.
struct PhGroup {
    uint32_t hap_dw;
    uint32_t H;
    uint32_t R_tab; // R | tab << 16
    uint32_t out;   // index into sums[]
    double init64;  // Y[0][j] = DBL_MAX/16/H, divided on the host (antidiagsPairHMM.c:135)
    float init32;   // FLT_MAX/16/H for the float fill
    uint32_t reserved;
};

// Packed float kernel (AGX_PHMM_F32_FMA): one group of G lanes carries two haplotypes of the same
// read ([0] in the x half, [1] in the y half of every float2).  A group without a second haplotype
// points [1] at an empty haplotype (H = 0) and at the spare slot sums[n_pairs].
// Read trains (round 3): the group fills a second read of R2 bases (0: none) behind the first without draining; its sums
// go to out2[] (the spare slot when there is none).
struct PhGroup2 {
    uint32_t hap_dw[2];
    uint32_t H[2];
    uint32_t out[2];
    float init32[2];
    uint32_t R_tab; // R | tab << 16
    uint32_t R2;
    uint32_t out2[2];
};

// One read table to build in LDS: the read's five tracks start at read_dw, each padded to 4 bytes.
// (A launch with read trains has TWO entries per table: the first read, then the second or {0, 0}.)
struct PhTab {
    uint32_t read_dw;
    uint32_t R;
};

// One wavefront: n_groups groups of G lanes sharing n_tabs read tables, stepping `steps` rows
// (>= max(R) + G - 1).  Every table has steps + G - 1 rows: G-1 neutral rows, the read, neutral tail.
struct PhWave {
    uint32_t first_group;
    uint32_t first_tab;
    uint16_t n_groups;
    uint16_t n_tabs;
    uint16_t G;
    uint16_t reserved;
    uint32_t steps;
};

#define AGX_PH_FOR_EACH_CLASS(X) \
    X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)
// classes of the packed float kernel: seven registers per column (six of state, one of symbols); widths
// 31 and 32 are built for two waves per SIMD like width 32 of the double kernel (257 -> 256 VGPRs)
#define AGX_PH_FOR_EACH_PK_CLASS(X) \
    X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) \
    X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
// (every width: with two haplotypes per group and at most 32 columns, odd widths often tile a
// haplotype over all 64 lanes where the even ones leave lanes idle, e.g. H = 300 = 16 lanes x 19)
static const int kPhPkClasses[] = {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32};
static const int kPhPkNumClasses = sizeof(kPhPkClasses) / sizeof(kPhPkClasses[0]);
// measured for every width (profiles/r01_calibration.log, "packed float kernel, third calibration")
static const double kPhPkClassCost[] = {1.834, 1.541, 1.454, 1.332, 1.293, 1.220, 1.195, 1.141, 1.137, 1.098, 1.093, 1.083, 1.073, 1.059, 1.063, 1.073, 1.068, 1.044, 1.034, 1.049, 1.034, 1.024, 1.024, 1.029, 1.020, 1.005, 1.000, 1.015, 1.005};
static const int kPhClasses[] = {4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40};
static const int kPhNumClasses = sizeof(kPhClasses) / sizeof(kPhClasses[0]);
// Measured lane time per padded cell of each class relative to the best one of its arithmetic
// (MI355X, tools/calibrate_classes.py, profiles/r01_calibration*.log); 0 = class not built
// for that arithmetic (too many VGPRs).  Rows: f64 reference order, f64 FMA, f32.  (Width 32 in double
// runs the two-waves-per-SIMD build, tools/cal_f64_wide.py.)
static const double kPhClassCost[3][19] = {
    // (the two double rows: round 2c, measured with the looked-up-prior fill, profiles/r02w_cal_f64_lut.log; before, with
    // phmm_fill: 1.656 1.400 1.307 1.244 1.218 1.173 1.124 1.133 1.109 1.073 1.073 1.053 1.051 1.000 1.011 and
    // 1.752 1.449 1.330 1.227 1.193 1.174 1.124 1.127 1.108 1.071 1.071 1.050 1.032 1.024 1.000)
    {1.569, 1.330, 1.232, 1.176, 1.144, 1.101, 1.096, 1.071, 1.058, 1.060, 1.035, 1.053, 1.005, 1.000, 1.015, 0, 0, 0, 0},
    {1.801, 1.457, 1.331, 1.258, 1.185, 1.136, 1.089, 1.099, 1.073, 1.086, 1.050, 1.066, 1.007, 1.017, 1.000, 0, 0, 0, 0},
    {1.872, 1.500, 1.346, 1.248, 1.184, 1.158, 1.132, 1.109, 1.090, 1.075, 1.090, 1.068, 1.045, 1.023, 1.011, 1.000, 1.071, 1.056, 1.045},
};

// bytes of LDS one table row takes: Qr, Qi, Qd, Qg (+ a separate mismatch prior when it is not
// Qr itself, AGX_PHMM_GATK_PRIOR) + the read base
__host__ __device__ static inline size_t ph_row_bytes(bool f64, bool mis_col) { return (f64 ? 8u : 4u) * (mis_col ? 5u : 4u) + 1u; }
__host__ __device__ static inline size_t ph_tab_bytes(bool f64, bool mis_col, uint32_t rows)
{
    return (ph_row_bytes(f64, mis_col) * rows + 15u) & ~(size_t)15u;
}

// the double kernel with looked-up priors (agx_phmm_lut_kernel.hip): one row of {Qi, Qd, Qg, prior[A], prior[C], prior[T], prior[G]}
// per read position
#define AGX_PH_LUT_ROW_BYTES 56u
#define AGX_PH_LUT_W2_FROM 32 /* widths from here on are built for two waves per SIMD */
__host__ __device__ static inline size_t ph_lut_tab_bytes(uint32_t rows) { return ((size_t)rows * AGX_PH_LUT_ROW_BYTES + 15u) & ~(size_t)15u; }
// Reads whose table would exceed a wave's LDS share at two waves per SIMD (20 KB = 365 rows) keep a RING of 256 rows instead
// (round 3, STREAM in agx_phmm_lut_kernel.hip): a wave only ever needs the rows behind its current step, the ring is refilled
// 64 rows at a time.  rows = the read's R + 2.
#define AGX_PH_LUT_FULL_ROWS 365u
#define AGX_PH_LUT_RING_ROWS 256u
__host__ __device__ static inline bool ph_lut_is_ring(uint32_t rows) { return rows > AGX_PH_LUT_FULL_ROWS; }
__host__ __device__ static inline size_t ph_lut_lds_bytes(uint32_t rows) { return ph_lut_tab_bytes(ph_lut_is_ring(rows) ? AGX_PH_LUT_RING_ROWS : rows); }
// the packed float kernel's tables: one 32-byte row of derived values per read position
__host__ __device__ static inline size_t ph_pk_tab_bytes(uint32_t rows) { return (size_t)rows * 32u; }

// mode: 0 = f64 reference order, 1 = f64 with FMA contraction, 2 = f32, 3 = f64 rescue pass
// over an f32 result (only groups whose sums[out] < rescue_below are recomputed), 4 = f64
// reference order with probability tracks instead of Phred characters (pairHMM() seam).
// The packed float fill counts the pairs whose sum came out below the float range (`below`; indices >= n_pairs are the
// spare slot of vacant halves): the double rescue plan is launched only when that count is not zero.
// Accuracy guard (round 3): the float recurrences lose about 1.1e-7 sqrt(R) of log10 L (2.7e-7 sqrt(R) with the GATK prior)
// at worst -- tools/phmm_f32_guard_cal.py, profiles/r03h_f32_guard_cal.log -- which matters where |log10 L| is small: a
// pair whose |log10 L| comes out below guard_k sqrt(R + 8) is handed to the double rescue plan like an underflowed one (its
// sum is stored as 0).  guard_c = the float scaling constant FLT_MAX / 16, guard_k2 = guard_k log2(10); guard_k2 = 0: off.
struct PhUnderflow {
    double below;
    uint32_t n_pairs;
    unsigned long long *count;
    double guard_c;
    float guard_k2;
    // agx_phmm_batch_bind_results: where the fill itself writes log10(sum) - log10(FLT_MAX / 16) (page-locked host memory,
    // n_pairs doubles; NULL: nowhere), and a word it sets when a pair went to the rescue plan instead
    double *logs_host;
    unsigned *flag_host;
    double log_c32;
    // the fast cell's table rows made once per batch (phmm_pk_rows: two float4 per read position, the read whose tracks
    // start at image word d at row (d - rows_base_dw) / 5 * 4); NULL: every wave derives its rows itself
    const void *pk_rows;
    uint32_t rows_base_dw;
};
// makes those rows for every read of `reads` (one PhTab per read); lut / lut_mis as the packed fill takes them
int agx_phmm_pk_rows_launch(const uint32_t *img, const PhTab *reads, uint32_t n_reads, const void *lut, const void *lut_mis, void *rows,
                            uint32_t rows_base_dw, hipStream_t s);
int agx_phmm_pk_launch_class(int cols_per_lane, bool all_groups_16, bool fast, const uint32_t *img, const PhGroup2 *groups, const PhTab *tabs,
                             const PhWave *waves, uint32_t n_waves, const void *lut, const void *lut_mis, double *sums,
                             const PhUnderflow &uf, size_t lds_bytes, hipStream_t s);
// the fast cell with read trains (two PhTab entries per table; agx_phmm_pk_train_kernel.hip)
int agx_phmm_pk_train_launch_class(int cols_per_lane, bool all_groups_16, const uint32_t *img, const PhGroup2 *groups, const PhTab *tabs,
                                   const PhWave *waves, uint32_t n_waves, const void *lut, const void *lut_mis, double *sums,
                                   const PhUnderflow &uf, size_t lds_bytes, hipStream_t s);
void agx_phmm_pk_train_preload();
// Haplotypes no class can span (more than 64 lanes x the widest class of the batch's arithmetic) run
// one pair per wavefront in stripes of 64 x AGX_PH_STRIPE_COLS = 1536 columns, always in double (mode 0, 1
// or 4).  grid workgroups walk the n_waves pairs; scratch holds 6 * scratch_rows doubles per workgroup.
// 24 columns: the widest tiling that keeps two waves per SIMD (<= 256 VGPRs) without spills in the
// cell loop; measured against 26/28/30 in profiles/r01_calibration.log ("striped kernel").
#define AGX_PH_STRIPE_COLS 24
#define AGX_PH_FOR_EACH_STRIPE_CLASS(X) X(24)
// mis_div (with lut_mis = NULL): the mismatch prior is Qr / 3, divided in the step head (GATK prior on reads whose five-column
// table would not fit the LDS)
int agx_phmm_stripe_launch(int mode, int cols_per_lane, const uint32_t *img, const PhGroup *groups, const PhTab *tabs, const PhWave *waves,
                           uint32_t n_waves, uint32_t grid, const void *lut, const void *lut_mis, int mis_div, double *sums, double *scratch,
                           uint32_t scratch_rows, int negate, size_t lds_bytes, hipStream_t s);
// float modes: log10(sum) - log10(C) per pair on the device (a negated sum = recomputed in double, scaled by DBL_MAX/16)
// AGX_PHMM_F64 / F64_FMA on plain DNA (reads of ACGTN, haplotypes of ACGT): the prior comes from the read's LDS table
int agx_phmm_lut_launch_class(bool fma, int cols_per_lane, bool all_groups_16, const uint32_t *img, const PhGroup *groups,
                              const PhTab *tabs, const PhWave *waves, uint32_t n_waves, const void *lut, const void *lut_mis,
                              double *sums, size_t lds_bytes, bool phased, bool stream, hipStream_t s);
void agx_phmm_lut_preload();
void agx_phmm_pk_preload();
void agx_phmm_scalar_preload();
void agx_phmm_finish_preload();
// (copies the two counters to the host and RESETS them)
int agx_phmm_finish_launch(const double *sums, double *logs, uint32_t n, double log_c64, double log_c32,
                           unsigned long long *n_rescued, unsigned long long *n_rescued_host, hipStream_t s);
int agx_phmm_launch_class(int mode, int cols_per_lane, bool all_groups_16, const uint32_t *img, const PhGroup *groups, const PhTab *tabs,
                          const PhWave *waves, uint32_t n_waves, const void *lut, const void *lut_mis, double *sums,
                          double rescue_below,
                          unsigned long long *n_rescued, size_t lds_bytes, hipStream_t s);
