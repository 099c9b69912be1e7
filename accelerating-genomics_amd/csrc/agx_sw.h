// Device work records of the Smith-Waterman fill (shared by scheduler and kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// One alignment pair = one group of G lanes.  Offsets are in 4-byte words into the packed
// image: x (the shorter sequence, zero-padded to G*C bytes), y (the longer, padded to 4).
struct SwGroup {
    uint32_t x_dw;
    uint32_t y_dw;
    uint32_t lx_ly; // lx | ly << 16
    uint32_t out;   // index into scores[]
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
    uint32_t reserved;
};

// Column-per-lane classes the kernel is instantiated for.
static const int kSwClasses[] = {4, 8, 12, 16, 20, 24, 28, 32, 36, 40};
static const int kSwNumClasses = sizeof(kSwClasses) / sizeof(kSwClasses[0]);
// Measured lane time per padded cell of each class, relative to the widest one (MI355X,
// tools/calibrate_classes.py, profiles/r01_calibration.log): narrow classes amortise the
// per-step work (DPP shifts, row symbol, loop control) over fewer cells.
static const double kSwClassCost[] = {1.365, 1.186, 1.117, 1.069, 1.026, 1.015, 1.011, 1.007, 1.004, 1.0};
// same for the packed int16 kernel (profiles/r01_calibration_pk.log)
static const double kSwPkClassCost[] = {1.537, 1.280, 1.171, 1.120, 1.063, 1.046, 1.017, 1.017, 1.006, 1.0};

int agx_sw_pk_launch_class(int cols_per_lane, const uint32_t *img, const SwGroup2 *groups, const SwWave *waves,
                           uint32_t n_waves, int32_t *scores, hipStream_t s);
int agx_sw_launch_class(int cols_per_lane, const uint32_t *img, const SwGroup *groups, const SwWave *waves,
                        uint32_t n_waves, int32_t *scores, hipStream_t s);
