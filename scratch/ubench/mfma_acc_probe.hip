// Probe: is v_mfma_f32_4x4x1's A*B + C a fused multiply-add (one rounding) or round(round(A*B) + C)?
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scratch/ubench/mfma_acc_probe.hip -o mfma_acc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void Acc(const float *a, const float *b, const float *c, float *mf, float *sep, float *fused, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float av = a[i], bv = b[i], cv = c[i];
    float4v cc = {cv, cv, cv, cv};
    float4v r = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, cc, 0, 0, 0);
    mf[i]    = r[threadIdx.x & 3];  // A[4b + (l&3)] * B[l]: the lane's own product
    sep[i]   = av * bv + cv;        // (-ffp-contract=off: two roundings)
    fused[i] = __builtin_fmaf(av, bv, cv);
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), c(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    for (int i = 0; i < n; ++i) {
        a[i] = ((rnd() >> 9) / 8388608.0f - 0.3f) * 0.4f;
        b[i] = (float)(rnd() & 255) * (1.0f / 255.0f);
        c[i] = ((rnd() >> 9) / 8388608.0f) * ((i & 1) ? 1.0f : 0.01f);
    }
    float *da, *db, *dc, *dm, *ds, *df;
    for (float **p : {&da, &db, &dc, &dm, &ds, &df}) hipMalloc(p, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    Acc<<<n / 256, 256>>>(da, db, dc, dm, ds, df, n);
    std::vector<float> m(n), sp(n), fu(n);
    hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(sp.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(fu.data(), df, n * 4, hipMemcpyDeviceToHost);
    long dsep = 0, dfus = 0, sepfus = 0;
    for (int i = 0; i < n; ++i) {
        dsep += m[i] != sp[i];
        dfus += m[i] != fu[i];
        sepfus += sp[i] != fu[i];
    }
    printf("mfma(a,b,c) over %d triples: differs from round(round(a*b)+c) in %ld, from fma(a,b,c) in %ld (the two differ in %ld)\n", n, dsep, dfus, sepfus);
    return 0;
}
