// Microbenchmark (round 5): what ONE source row of the matrix scale kernel costs a SIMD in issue time, with no memory
// behind it -- the kernel's own instruction mix in the kernel's own order, at the kernel's occupancy (four waves per
// SIMD: 256-lane workgroups, four per CU), pieces added one at a time:
//   A  24 v_pk_fma_f32 (acc = acc * keep + prod, keep an SGPR pair)                      -- the sums
//   B  A + 12 v_mfma_f32_4x4x1 (zero accumulator)                                        -- + the products
//   C  B + 12 v_cvt_f32_ubyteN + 6 v_pk_mul_f32                                          -- + the decode
//   D  C + 4 v_writelane_b32 + s_nop 3 + 2 v_min3_u32                                    -- + the A operand, alpha minimum
//   E  D + 4 v_mov_b32                                                                   -- + the ring read-out (round 5)
//   R  D without the cvt, S  D without any decode (round 6: what typed buffer loads would leave on the VALU)
//   F  12 cvt alone      G  12 mfma alone      H  6 pk_mul alone     I  D with plain v_mul (12) instead of pk_mul (6)
// Output: nominal-clock cycles per "row" and per SIMD (4 waves share it), i.e. time * 2.4 GHz / (rows per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
Mix(float *out, unsigned q0, unsigned q1, unsigned q2, unsigned q3, float w, int iters) {
    f2 acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = f2{threadIdx.x * 0.001f + i, 1.0f};
    unsigned q[4] = {q0 + threadIdx.x, q1, q2, q3};
    const f2 keep = {1.0f, 1.0f};
    const f2 k255 = {1.0f / 255.0f, 1.0f / 255.0f};
    int wa = 0;
    unsigned amin = 0xffffffffu;
    const f4 zero = {0, 0, 0, 0};
    const int ws = __builtin_amdgcn_readfirstlane(__float_as_int(w));
    constexpr bool kSums = MODE <= 4 || MODE >= 8, kMfma = (MODE >= 1 && MODE <= 4) || MODE == 6 || MODE >= 8,
                   kDecode = (MODE >= 2 && MODE <= 4) || (MODE >= 8 && MODE != 13), kLane = (MODE >= 3 && MODE <= 4) || MODE >= 8,
                   kMov = MODE == 4, kCvtOnly = MODE == 5, kMulOnly = MODE == 7, kPlainMul = MODE == 8,
                   kMfmaScale = MODE == 9, kPermFma = MODE == 10, kCvtNoScale = MODE == 11,
                   kTyped = MODE == 12;  // (round 6) the samples arrive as float(u8) from a typed load: no cvt, the scale stays
    int ca = __float_as_int(1.0f / 255.0f);  // (A operand of the scaling MFMA: the same constant in lanes 0..3)
    for (int it = 0; it < iters; ++it) {
        unsigned qq[4] = {q[0], q[1], q[2], q[3]};
        if (kMov) {
            asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "=&v"(qq[0]), "=&v"(qq[1]), "=&v"(qq[2]), "=&v"(qq[3]) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
        }
        if (kLane) {
            asm volatile("v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %1, 1\n\tv_writelane_b32 %0, %1, 2\n\t"
                         "v_writelane_b32 %0, %1, 3\n\ts_nop 3" : "+v"(wa) : "s"(ws));
            asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(amin) : "v"(qq[0]), "v"(qq[1]));
            asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(amin) : "v"(qq[2]), "v"(qq[3]));
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float d[6] = {1.0f, 2.0f, 3.0f, 4.0f, 5.0f, 6.0f};
            if (kDecode || kCvtOnly || kMulOnly) {
                const unsigned pa = qq[half * 2], pb = qq[half * 2 + 1];
                if (kPermFma) {
                    unsigned x[6];
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[0]) : "v"(pa), "v"(0x4B000000u), "s"(0x030c0c04u));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[1]) : "v"(pa), "v"(0x4B000000u), "s"(0x030c0c05u));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[2]) : "v"(pa), "v"(0x4B000000u), "s"(0x030c0c06u));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[3]) : "v"(pb), "v"(0x4B000000u), "s"(0x030c0c04u));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[4]) : "v"(pb), "v"(0x4B000000u), "s"(0x030c0c05u));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[5]) : "v"(pb), "v"(0x4B000000u), "s"(0x030c0c06u));
                    const f2 off = {-8388608.0f / 255.0f, -8388608.0f / 255.0f};
#pragma unroll
                    for (int i = 0; i < 6; i += 2) {
                        f2 m = {__uint_as_float(x[i]), __uint_as_float(x[i + 1])};
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(m) : "v"(k255), "v"(off));
                        d[i] = m.x;
                        d[i + 1] = m.y;
                    }
                } else if (kTyped) {
                    d[0] = __uint_as_float(pa); d[1] = __uint_as_float(pb); d[2] = __uint_as_float(pa ^ 1u);
                    d[3] = __uint_as_float(pb ^ 1u); d[4] = __uint_as_float(pa ^ 2u); d[5] = __uint_as_float(pb ^ 2u);
                } else if (!kMulOnly) {
                    asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(d[0]) : "v"(pa));
                    asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(d[1]) : "v"(pa));
                    asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(d[2]) : "v"(pa));
                    asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(d[3]) : "v"(pb));
                    asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(d[4]) : "v"(pb));
                    asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(d[5]) : "v"(pb));
                }
                if (kMfmaScale) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        f4 r = __builtin_amdgcn_mfma_f32_4x4x1f32(__int_as_float(ca), d[i], zero, 4, 0, 0);
                        d[i] = r.x;
                    }
                } else if (!kCvtOnly && !kPermFma && !kCvtNoScale) {
                    if (kPlainMul) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(k255.x));
                    } else {
#pragma unroll
                        for (int i = 0; i < 6; i += 2) {
                            f2 m = {d[i], d[i + 1]};
                            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(m) : "v"(k255));
                            d[i] = m.x;
                            d[i + 1] = m.y;
                        }
                    }
                }
                if (kCvtOnly || kMulOnly) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) asm volatile("" : : "v"(d[i]));
                }
            }
            f4 prod[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                prod[i] = f4{d[i], d[i], d[i], d[i]};
                if (kMfma) prod[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(__int_as_float(wa), d[i], zero, 4, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kSums) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const f2 lo = {prod[i].x, prod[i].y}, hi = {prod[i].z, prod[i].w};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[half * 12 + 2 * i]) : "s"(keep), "v"(lo));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[half * 12 + 2 * i + 1]) : "s"(keep), "v"(hi));
                }
            } else if (kMfma) {
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("" : : "v"(prod[i]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = __uint_as_float(amin) + __int_as_float(wa);
#pragma unroll
    for (int i = 0; i < 24; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The decode as ONE instruction per byte: v_mul_f32 with an SDWA byte select reads the byte as the bits of a float, i.e. the
// denormal u * 2^-149 (denormals are on: .amdhsa_float_denorm_mode_32 3); times K = (1/255) * 2^149 the exact product is
// u * (1/255), rounded once -- bit for bit RN(float(u) * (1.0f/255.0f)), what v_cvt_f32_ubyte + v_mul_f32 compute.
//   PIPE 0: decode, products, sums per half row in the kernel's order
//   PIPE 1: the NEXT half row's decode interleaved with this half row's sums (one multiply behind every second sum)
template <int PIPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
Mix2(float *out, unsigned q0, unsigned q1, unsigned q2, unsigned q3, float w, int iters) {
    f2 acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = f2{threadIdx.x * 0.001f + i, 1.0f};
    unsigned q[4] = {q0 + threadIdx.x, q1, q2, q3};
    const f2 keep = {1.0f, 1.0f};
    const float kscale = __builtin_ldexpf(1.0f / 255.0f, 149);
    int wa = 0;
    unsigned amin = 0xffffffffu;
    const f4 zero = {0, 0, 0, 0};
    const int ws = __builtin_amdgcn_readfirstlane(__float_as_int(w));
#define SDWA_MUL(D, Q, B) asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD" : "=v"(D) : "v"(Q), "v"(kscale))
    float dn[6];
    if (PIPE) {
        SDWA_MUL(dn[0], q[0], 0); SDWA_MUL(dn[1], q[0], 1); SDWA_MUL(dn[2], q[0], 2);
        SDWA_MUL(dn[3], q[1], 0); SDWA_MUL(dn[4], q[1], 1); SDWA_MUL(dn[5], q[1], 2);
    }
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %1, 1\n\tv_writelane_b32 %0, %1, 2\n\t"
                     "v_writelane_b32 %0, %1, 3\n\ts_nop 3" : "+v"(wa) : "s"(ws));
        asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(amin) : "v"(q[0]), "v"(q[1]));
        asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(amin) : "v"(q[2]), "v"(q[3]));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float d[6];
            if (PIPE) {
#pragma unroll
                for (int i = 0; i < 6; ++i) d[i] = dn[i];
            } else {
                const unsigned pa = q[half * 2], pb = q[half * 2 + 1];
                SDWA_MUL(d[0], pa, 0); SDWA_MUL(d[1], pa, 1); SDWA_MUL(d[2], pa, 2);
                SDWA_MUL(d[3], pb, 0); SDWA_MUL(d[4], pb, 1); SDWA_MUL(d[5], pb, 2);
            }
            f4 prod[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) prod[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(__int_as_float(wa), d[i], zero, 4, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned na = q[(half * 2 + 2) & 3], nb = q[(half * 2 + 3) & 3];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f2 lo = {prod[i].x, prod[i].y}, hi = {prod[i].z, prod[i].w};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[half * 12 + 2 * i]) : "s"(keep), "v"(lo));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[half * 12 + 2 * i + 1]) : "s"(keep), "v"(hi));
                if (PIPE) {
                    if (i == 0) SDWA_MUL(dn[0], na, 0);
                    if (i == 1) SDWA_MUL(dn[1], na, 1);
                    if (i == 2) SDWA_MUL(dn[2], na, 2);
                    if (i == 3) SDWA_MUL(dn[3], nb, 0);
                    if (i == 4) SDWA_MUL(dn[4], nb, 1);
                    if (i == 5) SDWA_MUL(dn[5], nb, 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = __uint_as_float(amin) + __int_as_float(wa) + dn[0];
#pragma unroll
    for (int i = 0; i < 24; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// (the identity itself, all 256 bytes: the SDWA product against cvt + mul)
__global__ void SdwaCheck(unsigned *bad) {
    const unsigned u = threadIdx.x & 255u, word = u * 0x01010101u;
    const float kscale = __builtin_ldexpf(1.0f / 255.0f, 149);
    float a, b, c, ref;
    SDWA_MUL(a, word, 0); SDWA_MUL(b, word, 1); SDWA_MUL(c, word, 2);
    asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(ref) : "v"(word));
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ref) : "v"(1.0f / 255.0f));
    if (__float_as_uint(a) != __float_as_uint(ref) || __float_as_uint(b) != __float_as_uint(ref) || __float_as_uint(c) != __float_as_uint(ref))
        atomicAdd(bad, 1u);
}

// (round 6) the same row WITH its loads, from an L1/L2-resident strip (4 KB a workgroup row, re-read every iteration), two
// rows in flight behind hand-placed waits:
//   TYPED 0: one global_load_dwordx4 a lane, 12 cvt + 6 pk_mul, 2 v_min3_u32            -- the kernel's row today
//   TYPED 1: four buffer_load_format_xyzw (8_8_8_8 USCALED: float(u8) in the registers), 6 pk_mul IN PLACE, 2 v_min3_f32
//   TYPED 2: as 1 without the multiplies (what an exact UNORM would leave; it is not exact: typed_load.hip)
typedef int i4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int TYPED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
Mix3(float *out, const unsigned char *src, float w, int iters) {
    f2 acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = f2{threadIdx.x * 0.001f + i, 1.0f};
    const f2 keep = {1.0f, 1.0f};
    const f2 k255 = {1.0f / 255.0f, 1.0f / 255.0f};
    int wa = 0;
    unsigned amin = 0xffffffffu;
    float fmin = 1e30f;
    const f4 zero = {0, 0, 0, 0};
    const int ws = __builtin_amdgcn_readfirstlane(__float_as_int(w));
    const unsigned char *strip = src + (size_t)blockIdx.x * 8192;
    const unsigned long long a = reinterpret_cast<unsigned long long>(strip);
    i4v rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    rs.z = 8192;
    rs.w = (int)((4u | (5u << 3) | (6u << 6) | (7u << 9)) | (2u << 12) | (10u << 15));
    const unsigned voff = (threadIdx.x >> 6) * 1024 + (threadIdx.x & 63) * 4;
    const unsigned char *lane_ptr = strip + threadIdx.x * 16;
    u4v qa, qb;
    f4 ta[4], tb[4];
    auto issue = [&](u4v &q, f4 (&t)[4], unsigned soff) __attribute__((always_inline)) {
        if (TYPED == 0) {
            const unsigned char *p = lane_ptr + soff;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q) : "v"(p) : "memory");
        } else {
            asm volatile("buffer_load_format_xyzw %0, %4, %5, %6 offen\n\t"
                         "buffer_load_format_xyzw %1, %4, %5, %6 offen offset:256\n\t"
                         "buffer_load_format_xyzw %2, %4, %5, %6 offen offset:512\n\t"
                         "buffer_load_format_xyzw %3, %4, %5, %6 offen offset:768"
                         : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : "v"(voff), "s"(rs), "s"(soff) : "memory");
        }
    };
    auto row = [&](u4v &q, f4 (&t)[4]) __attribute__((always_inline)) {
        if (TYPED == 0) asm volatile("s_waitcnt vmcnt(1)" : "+v"(q) : : "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : : "memory");
        asm volatile("v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %1, 1\n\tv_writelane_b32 %0, %1, 2\n\t"
                     "v_writelane_b32 %0, %1, 3\n\ts_nop 3" : "+v"(wa) : "s"(ws));
        if (TYPED == 0) {
            asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(amin) : "v"(q.x), "v"(q.y));
            asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(amin) : "v"(q.z), "v"(q.w));
        } else {
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(fmin) : "v"(t[0].w), "v"(t[1].w));
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(fmin) : "v"(t[2].w), "v"(t[3].w));
        }
        const unsigned qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float d[6];
            if (TYPED == 0) {
                const unsigned pa = qq[half * 2], pb = qq[half * 2 + 1];
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(d[0]) : "v"(pa));
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(d[1]) : "v"(pa));
                asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(d[2]) : "v"(pa));
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(d[3]) : "v"(pb));
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(d[4]) : "v"(pb));
                asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(d[5]) : "v"(pb));
#pragma unroll
                for (int i = 0; i < 6; i += 2) {
                    f2 m = {d[i], d[i + 1]};
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(m) : "v"(k255));
                    d[i] = m.x;
                    d[i + 1] = m.y;
                }
            } else {
                // (x, y) of a pixel are an aligned register pair of its load; z of the two pixels is not a pair: one
                // plain multiply each -- 2 packed + 2 plain multiplies per two pixels
                f4 &p0 = t[half * 2], &p1 = t[half * 2 + 1];
                if (TYPED == 1) {
                    f2 m0 = {p0.x, p0.y}, m1 = {p1.x, p1.y};
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(m0) : "v"(k255));
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(m1) : "v"(k255));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(p0.z) : "v"(k255.x));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(p1.z) : "v"(k255.x));
                    p0.x = m0.x; p0.y = m0.y; p1.x = m1.x; p1.y = m1.y;
                }
                d[0] = p0.x; d[1] = p0.y; d[2] = p0.z; d[3] = p1.x; d[4] = p1.y; d[5] = p1.z;
            }
            f4 prod[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) prod[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(__int_as_float(wa), d[i], zero, 4, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f2 lo = {prod[i].x, prod[i].y}, hi = {prod[i].z, prod[i].w};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[half * 12 + 2 * i]) : "s"(keep), "v"(lo));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[half * 12 + 2 * i + 1]) : "s"(keep), "v"(hi));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    issue(qa, ta, 0);
    issue(qb, tb, 4096);
    for (int it = 0; it < iters; it += 2) {
        row(qa, ta);
        issue(qa, ta, 0);
        row(qb, tb);
        issue(qb, tb, 4096);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qa), "+v"(qb), "+v"(ta[0]), "+v"(ta[1]), "+v"(ta[2]), "+v"(ta[3]), "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]) : : "memory");
    float s = __uint_as_float(amin) + __int_as_float(wa) + fmin;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int TYPED>
void Run3(const char *name) {
    float *out;
    unsigned char *src;
    const int blocks = 256 * 4, iters = 20000;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&src, (size_t)blocks * 8192);
    hipMemset(src, 0x40, (size_t)blocks * 8192);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    Mix3<TYPED><<<blocks, 256>>>(out, src, 0.25f, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    Mix3<TYPED><<<blocks, 256>>>(out, src, 0.25f, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.3f ms: %6.1f nominal cycles per row and wave (one of four waves on its SIMD)\n", name, ms,
           ms * 1e-3 * 2.4e9 / (4.0 * iters));
    hipFree(out);
    hipFree(src);
}

template <int PIPE>
void Run2(const char *name) {
    float *out;
    const int blocks = 256 * 4, iters = 20000;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    Mix2<PIPE><<<blocks, 256>>>(out, 0x01020304u, 0x05060708u, 0x090a0b0cu, 0x0d0e0f10u, 0.25f, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    Mix2<PIPE><<<blocks, 256>>>(out, 0x01020304u, 0x05060708u, 0x090a0b0cu, 0x0d0e0f10u, 0.25f, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.3f ms: %6.1f nominal cycles per row and wave (one of four waves on its SIMD)\n", name, ms,
           ms * 1e-3 * 2.4e9 / (4.0 * iters));
    hipFree(out);
}

template <int MODE>
void Run(const char *name) {
    float *out;
    const int blocks = 256 * 4, iters = 20000;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    Mix<MODE><<<blocks, 256>>>(out, 0x01020304u, 0x05060708u, 0x090a0b0cu, 0x0d0e0f10u, 0.25f, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    Mix<MODE><<<blocks, 256>>>(out, 0x01020304u, 0x05060708u, 0x090a0b0cu, 0x0d0e0f10u, 0.25f, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // every SIMD holds 4 waves, each doing `iters` rows: cycles per row and SIMD = time * clock / (4 * iters)
    printf("%-46s %8.3f ms: %6.1f nominal cycles per row and wave (one of four waves on its SIMD)\n", name, ms,
           ms * 1e-3 * 2.4e9 / (4.0 * iters));
    hipFree(out);
}

int main() {
    Run<0>("A 24 pk_fma");
    Run<1>("B A + 12 mfma");
    Run<2>("C B + 12 cvt + 6 pk_mul");
    Run<3>("D C + 4 writelane + nop + 2 min3");
    Run<4>("E D + 4 v_mov");
    Run<5>("F 12 cvt");
    Run<6>("G 12 mfma");
    Run<7>("H 6 pk_mul");
    Run<8>("I D with 12 v_mul for 6 pk_mul");
    Run<9>("J D with the scale as 12 more mfma");
    Run<10>("K D with 12 v_perm + 6 pk_fma as the decode");
    Run<11>("L D without the scale (12 cvt only)");
    Run<12>("R D without the 12 cvt (typed load, USCALED)");
    Run<13>("S D without any decode (typed load, UNORM)");
    Run3<0>("T0 D with its load (dwordx4 from L1/L2)");
    Run3<1>("T1 typed xyzw USCALED loads + 4 pk_mul + 4 mul");
    Run3<2>("T2 typed loads, no multiplies (UNORM: inexact)");
    Run2<0>("P D with the decode as 12 v_mul_f32_sdwa");
    Run2<1>("Q P, next half row's decode between the sums");
    {
        unsigned *bad, h = 0;
        hipMalloc(&bad, 4);
        hipMemset(bad, 0, 4);
        SdwaCheck<<<1, 256>>>(bad);
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("v_mul_f32_sdwa(byte, 2^149/255) == v_mul_f32(v_cvt_f32_ubyte(byte), 1/255) for all 256 bytes: %s (%u differ)\n", h ? "NO" : "yes", h);
    }
    return 0;
}
