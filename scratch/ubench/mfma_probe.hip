// Probe: layout and rounding of v_mfma_f32_4x4x1_16b_f32 on gfx950.
//  * which lane/register receives A_i * B_j
//  * is D = A*B + 0 the correctly rounded fp32 product (compare with v_mul_f32), incl. tiny operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void Probe(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    float4v c = {0.f, 0.f, 0.f, 0.f};
    float4v r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    d[l * 4 + 0] = r.x; d[l * 4 + 1] = r.y; d[l * 4 + 2] = r.z; d[l * 4 + 3] = r.w;
}

__global__ void ProbeBcast(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    float4v c = {0.f, 0.f, 0.f, 0.f};
    // cbsz = 4, abid = 0: every block takes its A values from block 0 (lanes 0..3)?
    float av = l < 4 ? a[l] : 1e30f;
    int wa = 0;
    asm volatile("v_writelane_b32 %0, %1, 0" : "+v"(wa) : "s"(__float_as_int(2.0f)));
    asm volatile("v_writelane_b32 %0, %1, 1" : "+v"(wa) : "s"(__float_as_int(3.0f)));
    asm volatile("v_writelane_b32 %0, %1, 2" : "+v"(wa) : "s"(__float_as_int(5.0f)));
    asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(wa) : "s"(__float_as_int(7.0f)));
    float4v r = __builtin_amdgcn_mfma_f32_4x4x1f32(av, b[l], c, 4, 0, 0);
    float4v r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(__int_as_float(wa), b[l], c, 4, 0, 0);
    d[l * 8 + 0] = r.x; d[l * 8 + 1] = r.y; d[l * 8 + 2] = r.z; d[l * 8 + 3] = r.w;
    d[l * 8 + 4] = r2.x; d[l * 8 + 5] = r2.y; d[l * 8 + 6] = r2.z; d[l * 8 + 7] = r2.w;
}

__global__ void Rounding(const float *a, const float *b, float *mf, float *vm, int n) {
    const int l = threadIdx.x;
    for (int it = 0; it < n; ++it) {
        const float av = a[it * 64 + l], bv = b[it * 64 + l];
        float4v c = {0.f, 0.f, 0.f, 0.f};
        float4v r = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c, 0, 0, 0);
        // lane l = 4*blk + j holds A_i * B_j in register i (if the layout is as assumed)
        for (int i = 0; i < 4; ++i) {
            const float ai = __shfl(av, (l & ~3) + i);
            float p;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(ai), "v"(bv));
            mf[(it * 64 + l) * 4 + i] = i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w;
            vm[(it * 64 + l) * 4 + i] = p;
        }
    }
}

int main() {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1000.f + l; hb[l] = 1.f + l * 0.001f; }
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    Probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 3, 4, 5, 63}) {
        printf("lane %2d:", l);
        for (int v = 0; v < 4; ++v) {
            // find (i, j) with ha[i] * hb[j] == hd
            int fi = -1, fj = -1;
            for (int i = 0; i < 64 && fi < 0; ++i) for (int j = 0; j < 64; ++j)
                if (ha[i] * hb[j] == hd[l * 4 + v]) { fi = i; fj = j; break; }
            printf("  reg%d = A[lane %d] * B[lane %d]", v, fi, fj);
        }
        printf("\n");
    }
    {
        float *d2; hipMalloc(&d2, 64 * 8 * 4);
        ProbeBcast<<<1, 64>>>(a, b, d2);
        float h2[512]; hipMemcpy(h2, d2, 2048, hipMemcpyDeviceToHost);
        for (int l : {0, 5, 18, 63}) {
            printf("bcast lane %2d: r = %g %g %g %g (A[0..3]*B[l] = %g %g %g %g)  r2/B = %g %g %g %g\n", l, h2[l*8], h2[l*8+1], h2[l*8+2], h2[l*8+3],
                   ha[0]*hb[l], ha[1]*hb[l], ha[2]*hb[l], ha[3]*hb[l], h2[l*8+4]/hb[l], h2[l*8+5]/hb[l], h2[l*8+6]/hb[l], h2[l*8+7]/hb[l]);
        }
    }
    // rounding: random mantissas, operands like the scaler's (bytes/255 times filter weights), tiny ones
    const int n = 4096;
    float *ra = new float[n * 64], *rb = new float[n * 64];
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < n * 64; ++i) {
        const int kind = (i / 64) % 4;
        if (kind == 0) { ra[i] = (float)((int)(rnd() >> 8) - (1 << 23)) / 65536.0f; rb[i] = (float)(rnd() >> 8) / 16777216.0f; }
        else if (kind == 1) { ra[i] = ((rnd() >> 9) / 8388608.0f - 0.5f) * 0.4f; rb[i] = (float)(rnd() & 255) * (1.0f / 255.0f); }
        else if (kind == 2) { uint32_t u = (rnd() & 0x007fffffu) | ((rnd() % 60 + 1) << 23); memcpy(&ra[i], &u, 4); rb[i] = (float)(rnd() & 255) * (1.0f / 255.0f); }
        else { uint32_t u = rnd() & 0x807fffffu; memcpy(&ra[i], &u, 4); rb[i] = 1.0f + (rnd() & 3); }  // denormal A
    }
    float *da, *db, *dm, *dv;
    hipMalloc(&da, n * 256); hipMalloc(&db, n * 256); hipMalloc(&dm, n * 1024); hipMalloc(&dv, n * 1024);
    hipMemcpy(da, ra, n * 256, hipMemcpyHostToDevice); hipMemcpy(db, rb, n * 256, hipMemcpyHostToDevice);
    Rounding<<<1, 64>>>(da, db, dm, dv, n);
    float *hm = new float[n * 256], *hv = new float[n * 256];
    hipMemcpy(hm, dm, n * 1024, hipMemcpyDeviceToHost); hipMemcpy(hv, dv, n * 1024, hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0};
    for (int i = 0; i < n * 256; ++i) {
        const int kind = (i / 256) % 4;
        ++tot[kind];
        if (memcmp(&hm[i], &hv[i], 4) != 0) {
            if (bad[kind]++ < 3) printf("kind %d mismatch: mfma %a  v_mul %a\n", kind, hm[i], hv[i]);
        }
    }
    for (int k = 0; k < 4; ++k) printf("kind %d: %ld of %ld products differ from v_mul_f32\n", k, bad[k], tot[k]);
    return 0;
}
