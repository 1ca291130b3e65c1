// timg_amd/csrc/debug_api.hip -- host-only introspection used by the CPU test
// suite (no GPU needed): dumps the resample plan in a normalised form.  Not
// part of the public ABI (not declared in include/timg_hip.h).
#include <cstring>

#include "resample_plan.h"

extern "C" int timg_hip_debug_plan_dump(int sw, int sh, int in_fmt, int dw, int dh, int filter,
                                        int *header, int *h_taps, float *h_coeff,
                                        int h_coeff_cap, int *v_cnt, int *v_rows,
                                        float *v_coeff, int v_cap) {
    timg_amd::ResamplePlan p;
    if (!timg_amd::BuildResamplePlan(sw, sh, in_fmt, dw, dh, filter, &p)) return -1;
    header[0] = p.vertical_first;
    header[1] = p.identity;
    header[2] = p.h_sequential;
    header[3] = p.h_width;
    header[4] = p.max_active_rows;
    header[5] = p.max_v_count;
    header[6] = p.max_h_count;
    header[7] = p.v_is_gather;
    if ((long)dw * p.h_width > h_coeff_cap) return -1;
    if ((long)p.v_rows.size() > v_cap) return -1;
    for (int x = 0; x < dw; ++x) {
        h_taps[2 * x]     = p.h_taps[x].n0;
        h_taps[2 * x + 1] = p.h_taps[x].count;
    }
    memcpy(h_coeff, p.h_coeff.data(), p.h_coeff.size() * sizeof(float));
    for (int y = 0; y < dh; ++y) v_cnt[y] = p.v_runs[y].count;
    memcpy(v_rows, p.v_rows.data(), p.v_rows.size() * sizeof(int));
    memcpy(v_coeff, p.v_coeff.data(), p.v_coeff.size() * sizeof(float));
    return (int)p.v_rows.size();
}
