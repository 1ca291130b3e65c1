// timg_amd/csrc/wave_ops.h -- wave64 scans and reductions on the DPP data path.
//
// __shfl* compiles to ds_bpermute_b32: an LDS-crossbar round trip (~100+ cycles) per step,
// six dependent ones for a scan.  The row_shr / row_bcast forms of DPP do the same job
// as operand modifiers of ordinary VALU instructions (gfx9: rows of 16 lanes, row_bcast:15
// and :31 carry a row's last lane into the following rows).
#ifndef TIMG_AMD_WAVE_OPS_H_
#define TIMG_AMD_WAVE_OPS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace timg_amd {

// value of lane `l` (wave-uniform) as a scalar
__device__ __forceinline__ uint32_t ReadLane(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

// shifted-in / masked-out lanes contribute 0
#define TIMG_DPP0(v, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, 0xf, false))
// ... or the lane's own value (for idempotent operators)
#define TIMG_DPPV(v, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), ctrl, rows, 0xf, false))

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t WaveInclusiveAdd(uint32_t v) {
    v += TIMG_DPP0(v, 0x111, 0xf);  // row_shr:1
    v += TIMG_DPP0(v, 0x112, 0xf);  // row_shr:2
    v += TIMG_DPP0(v, 0x114, 0xf);  // row_shr:4
    v += TIMG_DPP0(v, 0x118, 0xf);  // row_shr:8   -> prefix inside every row of 16
    v += TIMG_DPP0(v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    v += TIMG_DPP0(v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
    return v;
}

// sum over the wave, as a scalar
__device__ __forceinline__ uint32_t WaveSum(uint32_t v) { return ReadLane(WaveInclusiveAdd(v), 63); }

// maximum over the wave, as a scalar
__device__ __forceinline__ uint32_t WaveMaxU32(uint32_t v) {
    // (0 is the identity of an unsigned maximum: with 'old' = 0 the compiler folds each exchange into the v_max_u32
    // that consumes it -- with 'old' = v it emitted a copy, a v_mov_dpp and the maximum, four instructions a stage)
    v = max(v, TIMG_DPP0(v, 0x111, 0xf));
    v = max(v, TIMG_DPP0(v, 0x112, 0xf));
    v = max(v, TIMG_DPP0(v, 0x114, 0xf));
    v = max(v, TIMG_DPP0(v, 0x118, 0xf));
    v = max(v, TIMG_DPP0(v, 0x142, 0xa));
    v = max(v, TIMG_DPP0(v, 0x143, 0xc));
    return ReadLane(v, 63);
}

// inclusive prefix maximum over the 64 lanes
__device__ __forceinline__ uint32_t WaveInclusiveMaxU32(uint32_t v) {
    v = max(v, TIMG_DPP0(v, 0x111, 0xf));
    v = max(v, TIMG_DPP0(v, 0x112, 0xf));
    v = max(v, TIMG_DPP0(v, 0x114, 0xf));
    v = max(v, TIMG_DPP0(v, 0x118, 0xf));
    v = max(v, TIMG_DPP0(v, 0x142, 0xa));
    v = max(v, TIMG_DPP0(v, 0x143, 0xc));
    return v;
}

// the value of the lane below (lane 0: 0)
__device__ __forceinline__ uint32_t WaveShr1(uint32_t v) { return TIMG_DPP0(v, 0x138, 0xf); }  // wave_shr:1

// exclusive prefix maximum over the 64 lanes (lane 0: 0)
__device__ __forceinline__ uint32_t WaveExclusiveMaxU32(uint32_t v) {
    v = max(v, TIMG_DPP0(v, 0x111, 0xf));
    v = max(v, TIMG_DPP0(v, 0x112, 0xf));
    v = max(v, TIMG_DPP0(v, 0x114, 0xf));
    v = max(v, TIMG_DPP0(v, 0x118, 0xf));
    v = max(v, TIMG_DPP0(v, 0x142, 0xa));
    v = max(v, TIMG_DPP0(v, 0x143, 0xc));
    return TIMG_DPP0(v, 0x138, 0xf);  // wave_shr:1
}

// bitwise OR over the wave, as a scalar
__device__ __forceinline__ uint32_t WaveOr(uint32_t v) {
    v |= TIMG_DPP0(v, 0x111, 0xf);
    v |= TIMG_DPP0(v, 0x112, 0xf);
    v |= TIMG_DPP0(v, 0x114, 0xf);
    v |= TIMG_DPP0(v, 0x118, 0xf);
    v |= TIMG_DPP0(v, 0x142, 0xa);
    v |= TIMG_DPP0(v, 0x143, 0xc);
    return ReadLane(v, 63);
}

#undef TIMG_DPP0
#undef TIMG_DPPV

}  // namespace timg_amd

#endif  // TIMG_AMD_WAVE_OPS_H_
