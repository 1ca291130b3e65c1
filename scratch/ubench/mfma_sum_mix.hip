// Microbenchmark: the scale kernel's vertical mix -- 12 x v_mfma_f32_4x4x1 (zero accumulator: rounded products) and
// the 48 sums acc = acc * keep + prod -- with the sums as 24 v_pk_fma_f32 or as 48 v_fma_f32, with and without the
// MFMAs, 4 waves per SIMD (256 threads x 4 workgroups per CU).  Cycles per iteration at the NOMINAL 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) Mix(float *out, float w0, float keep, int iters) {
    f4 acc[12];
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        acc[i] = f4{0, 0, 0, 0};
        x[i]   = threadIdx.x * 0.001f + i;
    }
    const f4 zero = {0, 0, 0, 0};
    const f2 k2   = {keep, keep};
    for (int it = 0; it < iters; ++it) {
        f4 p[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (MODE & 1) p[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(w0, x[i], zero, 4, 0, 0);
            else p[i] = f4{x[i], x[i], x[i], x[i]};
            asm volatile("" : "+v"(p[i]));
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (MODE & 2) {  // 48 plain fma
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(keep), "v"(p[i].x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].y) : "v"(keep), "v"(p[i].y));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].z) : "v"(keep), "v"(p[i].z));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].w) : "v"(keep), "v"(p[i].w));
            } else {  // 24 packed fma
                f2 lo = {acc[i].x, acc[i].y}, hi = {acc[i].z, acc[i].w};
                f2 pl = {p[i].x, p[i].y}, ph = {p[i].z, p[i].w};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(lo) : "v"(k2), "v"(pl));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(hi) : "v"(k2), "v"(ph));
                acc[i] = f4{lo.x, lo.y, hi.x, hi.y};
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void Run(const char *name) {
    float *out;
    const int blocks = 256 * 4, iters = 20000;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    Mix<MODE><<<blocks, 256>>>(out, 1.0001f, 0.999f, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    Mix<MODE><<<blocks, 256>>>(out, 1.0001f, 0.999f, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // 4 waves per SIMD, each `iters` iterations: SIMD cycles per (wave, iteration)
    printf("%-34s %.3f ms: %.1f SIMD cycles per wave-iteration (4 waves/SIMD, nominal 2.4 GHz)\n", name, ms,
           ms * 1e-3 * 2.4e9 / (4.0 * iters));
    (void)hipFree(out);
}

int main() {
    Run<0>("24 pk_fma");
    Run<1>("12 mfma + 24 pk_fma");
    Run<2>("48 fma");
    Run<3>("12 mfma + 48 fma");
    return 0;
}
