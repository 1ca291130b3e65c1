// timg_amd/csrc/resample_plan.h
//
// Host-side resampling plan for the HIP scaler: the coefficient tables that
// make the device kernels bit-compatible with timg's STB scaler back-end
// (src/image-scaler.cc:75-98 -> third_party/stb/stb_image_resize2.h).
//
// The plan is O(W+H) work done once per (in_w,in_h,out_w,out_h) geometry in
// ImageScaler::Create's GPU twin; kernels only consume flat arrays.
#ifndef TIMG_AMD_RESAMPLE_PLAN_H
#define TIMG_AMD_RESAMPLE_PLAN_H

#include <cstdint>
#include <vector>

namespace timg_amd {

enum FilterKind : int {
    kFilterDefault  = 0,
    kFilterBox      = 1,  // stb "trapezoid"
    kFilterTriangle = 2,
    kFilterMitchell = 5,
    kFilterPoint    = 6,
};

// Horizontal taps of one output column: input pixels [n0, n0+count) weighted
// by coeff[coeff_offset .. +count).  The position of a tap inside this window
// (even/odd) selects its accumulation chain -- see hgather in the kernels.
struct HTaps {
    int32_t n0;
    int32_t count;
};

// Vertical contributions of one output row, in accumulation order.
struct VRun {
    int32_t first;  // index into v_rows / v_coeff
    int32_t count;
};

struct ResamplePlan {
    int in_w = 0, in_h = 0, out_w = 0, out_h = 0;
    int in_fmt = 0;

    bool identity       = false;  // 1:1 both axes: plain copy (+ swizzle)
    bool vertical_first = false;  // order of the two separable passes
    bool h_sequential   = false;  // <=3 taps per column: single chain

    int h_filter = 0, v_filter = 0;
    int h_width  = 0;   // floats per column in h_coeff ("widest")
    int v_is_gather = 0;  // 1 up / 2 down / 0 scatter (informational)
    int v_widest    = 0;

    // first/last input column any horizontal tap can touch
    int x_lo = 0, x_hi = 0;

    std::vector<HTaps> h_taps;     // out_w
    std::vector<float> h_coeff;    // out_w * h_width
    std::vector<VRun> v_runs;      // out_h
    std::vector<int32_t> v_rows;   // input row of each contribution
    std::vector<float> v_coeff;    // its weight

    // Largest number of output rows any single input row contributes to, and
    // the longest vertical run.  Decide streaming-kernel applicability.
    int max_active_rows = 0;
    int max_v_count     = 0;
    int max_h_count     = 0;
};

// Returns false for degenerate geometry.  `filter` is a FilterKind:
// kFilterDefault reproduces the reference's choice per axis.
bool BuildResamplePlan(int in_w, int in_h, int in_fmt, int out_w, int out_h,
                       int filter, ResamplePlan *plan);

}  // namespace timg_amd
#endif
