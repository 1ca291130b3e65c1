// timg_amd/csrc/cu_mask.h -- the CU mask of timg_hip_stream_create (capi.hip) as host code of its own, exported through
// the test-only debug library (tests/test_cu_mask.py, CPU).
//
// The mask as the driver reads it on a multi-XCD device (the kernel driver's own description of its mask walk): bit n of
// the mask is a CU of XCD n % kXcds, and an XCD's bits go round its kEngines shader engines -- its bit j is CU j / kEngines
// of engine j % kEngines.  So "the last r bits of every XCD" with r a multiple of kEngines takes r / kEngines CUs from
// every engine of every XCD.  Measured on the metric step (profiles/r6/partitioned_streams.txt): reserving 4 / 8 / 12 / 16
// CUs an XCD 1.23 / 1.28 / 1.16 / 1.18 ms against 1.32 on one stream; 3, 5, 9 ... (engines left with unequal CU counts:
// a launch runs at the pace of its smallest engine) 1.25-1.33; the same bits read XCD-major (a whole XCD masked) 1.42.
#ifndef TIMG_AMD_CU_MASK_H_
#define TIMG_AMD_CU_MASK_H_

#include <stdint.h>

namespace timg_amd {

constexpr int kXcds = 8, kEngines = 4, kCuMaskWords = 32;

// Fills mask[0 .. (cu_count + 31) / 32) and returns the CUs of every XCD that are kept free (the request rounded down to a
// multiple of kEngines; 0: every bit set), or -1 when the device's shape is not the one the layout above describes or the
// reserve would leave an XCD without CUs.
inline int BuildCuMask(int cu_count, int reserved_cus_per_xcd, uint32_t mask[kCuMaskWords]) {
    for (int i = 0; i < kCuMaskWords; ++i) mask[i] = 0;
    if (cu_count <= 0 || cu_count > 32 * kCuMaskWords || reserved_cus_per_xcd < 0) return -1;
    const int per_xcd  = cu_count / kXcds;
    const int reserved = reserved_cus_per_xcd / kEngines * kEngines;
    if (reserved > 0 && (cu_count != per_xcd * kXcds || per_xcd % kEngines != 0 || reserved >= per_xcd)) return -1;
    for (int n = 0; n < cu_count; ++n)
        if (reserved == 0 || n / kXcds < per_xcd - reserved) mask[n >> 5] |= 1u << (n & 31);
    return reserved;
}

}  // namespace timg_amd

#endif  // TIMG_AMD_CU_MASK_H_
