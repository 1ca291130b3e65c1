// Microbenchmark: VALU issue rate of plain vs packed fp32 mul/add, byte->float decode variants on gfx950.
// columns: time, SIMD cycles per wave-instruction assuming the NOMINAL 2.4 GHz (the clock under these
// loads is lower: read ratios between rows, not absolute cycles), lane-ops per nominal clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) Rate(float *out, float w0, float w1, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // plain: 16 mul + 16 add, separately rounded
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float t;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(w0));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(t), "v"(w1));
            }
        } else if (MODE == 1) {  // packed: 8 pk_mul + 8 pk_add  (same 32 lane-ops)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float2v v = {a[i], a[i + 1]}, t, ww0 = {w0, w0}, ww1 = {w1, w1};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(ww0));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(t), "v"(ww1));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else if (MODE == 2) {  // fma: 16 v_fma
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(w0), "v"(w1));
        } else if (MODE == 3) {  // cvt ubyte
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                unsigned u = __float_as_uint(a[i]);
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u));
            }
        } else if (MODE == 5) {  // v_perm_b32 (byte -> mantissa of 2^23)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                unsigned u = __float_as_uint(a[i]), r;
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(u), "v"(0x4B000000u), "s"(0x030c0c05u));
                a[i] = __uint_as_float(r);
            }
        } else if (MODE == 6) {  // decode as perm + fma: (2^23 + b) * k - 2^23 * k == b * k, one rounding
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                unsigned u = __float_as_uint(a[i]), r;
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(u), "v"(0x4B000000u), "s"(0x030c0c05u));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(r), "s"(w0), "v"(w1));
            }
        } else if (MODE == 7) {  // decode as cvt_ubyte + mul
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                unsigned u = __float_as_uint(a[i]);
                float t;
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(t) : "v"(u));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(t), "v"(w0));
            }
        } else if (MODE == 8) {  // decode two channels: 2 perm + 1 pk_fma
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                unsigned u = __float_as_uint(a[i]), r0, r1;
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r0) : "v"(u), "v"(0x4B000000u), "s"(0x030c0c05u));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r1) : "v"(u), "v"(0x4B000000u), "s"(0x030c0c06u));
                float2v v = {__uint_as_float(r0), __uint_as_float(r1)}, ww0 = {w0, w0}, ww1 = {w1, w1};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(ww0), "v"(ww1));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else if (MODE == 9) {  // plain v_mul only (issue rate of an independent stream)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(w0));
        } else if (MODE == 10) {  // v_pk_mul only
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float2v v = {a[i], a[i + 1]}, ww0 = {w0, w0};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(ww0));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else if (MODE == 11) {  // v_or_b32_sdwa (byte -> mantissa of 2^23)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                unsigned u = __float_as_uint(a[i]), r;
                asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(0x4B000000u), "v"(u));
                a[i] = __uint_as_float(r);
            }
        } else if (MODE == 12) {  // decode two channels: 2 sdwa or + 1 pk_fma
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                unsigned u = __float_as_uint(a[i]), r0, r1;
                asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r0) : "v"(0x4B000000u), "v"(u));
                asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r1) : "v"(0x4B000000u), "v"(u));
                float2v v = {__uint_as_float(r0), __uint_as_float(r1)}, ww0 = {w0, w0}, ww1 = {w1, w1};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(ww0), "v"(ww1));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else if (MODE == 13) {  // decode two channels as the kernel does: 2 cvt_ubyte + 1 pk_mul
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                unsigned u = __float_as_uint(a[i]);
                float2v v, ww0 = {w0, w0};
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(v.x) : "v"(u));
                asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(v.y) : "v"(u));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(ww0));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else if (MODE == 14 || MODE == 15 || MODE == 16) {
            // the scale kernel's vertical mix per 4 pixels x 3 channels: 12 MFMA (4x4x1, zero accumulator) and the 48 sums
            // as 24 v_pk_fma_f32 (14) / 48 v_fma_f32 (15); 16: the 24 v_pk_fma_f32 alone
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 zero = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                f4 p = {a[i], a[i], a[i], a[i]};
                if (MODE != 16) p = __builtin_amdgcn_mfma_f32_4x4x1f32(w0, a[i], zero, 4, 0, 0);
                asm volatile("" : "+v"(p));
                if (MODE == 15) {
                    float t0 = a[i], t1 = a[(i + 1) & 15], t2 = a[(i + 2) & 15], t3 = a[(i + 3) & 15];
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t0) : "v"(w1), "v"(p.x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t1) : "v"(w1), "v"(p.y));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t2) : "v"(w1), "v"(p.z));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t3) : "v"(w1), "v"(p.w));
                    a[i] = t0 + t1 * 0.0f + t2 * 0.0f + t3 * 0.0f;
                } else {
                    float2v lo = {p.x, p.y}, hi = {p.z, p.w}, ww1 = {w1, w1};
                    float2v v0 = {a[i], a[(i + 1) & 15]}, v1 = {a[(i + 2) & 15], a[(i + 3) & 15]};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(ww1), "v"(lo));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(ww1), "v"(hi));
                    a[i] = v0.x + v0.y * 0.0f + v1.x * 0.0f + v1.y * 0.0f;
                }
            }
        } else if (MODE == 4) {  // pk_fma
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float2v v = {a[i], a[i + 1]}, ww0 = {w0, w0}, ww1 = {w1, w1};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(ww0), "v"(ww1));
                a[i] = v.x; a[i + 1] = v.y;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void Run(const char *name, int instr_per_iter, int laneops_per_iter) {
    float *out;
    const int blocks = 256 * 8, iters = 20000;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    Rate<MODE><<<blocks, 256>>>(out, 1.0001f, 0.5f, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    Rate<MODE><<<blocks, 256>>>(out, 1.0001f, 0.5f, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0;
    const double instr = waves * iters * instr_per_iter;
    const double simd_cycles = ms * 1e-3 * 2.4e9;  // nominal clock
    printf("%-10s %.3f ms: %.2f cycles/wave-instr/SIMD (at 2.4 GHz), %.1f lane-ops/clk/CU\n", name, ms,
           simd_cycles / (instr / 1024.0), waves * iters * laneops_per_iter * 64.0 / (simd_cycles * 256.0));
    hipFree(out);
}

int main() {
    Run<0>("plain", 32, 32);
    Run<1>("packed", 16, 32);
    Run<2>("fma", 16, 16);
    Run<3>("cvt_ubyte", 16, 16);
    Run<4>("pk_fma", 8, 16);
    Run<5>("perm", 16, 16);
    Run<6>("perm+fma", 32, 16);
    Run<7>("cvt+mul", 32, 16);
    Run<8>("2perm+pkfma", 24, 16);
    Run<9>("mul", 16, 16);
    Run<10>("pk_mul", 8, 16);
    Run<11>("sdwa_or", 16, 16);
    Run<12>("2sdwa_or+pkfma", 24, 16);
    Run<13>("2cvt+pkmul", 24, 16);
    Run<14>("12mfma+24pkfma(+48 glue)", 84, 96);
    Run<15>("12mfma+48fma(+48 glue)", 108, 96);
    Run<16>("24pkfma(+48 glue) no mfma", 72, 96);
    return 0;
}
