// Do 16-byte global loads work at 4-byte aligned addresses on gfx950 (SH_MEM alignment mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void K(const unsigned char *src, uint4 *dst, int off) {
    const uint4 v = *reinterpret_cast<const uint4 *>(src + off + threadIdx.x * 20);
    dst[threadIdx.x] = v;
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i * 2654435761u;
    unsigned char *d; uint4 *o;
    hipMalloc(&d, 16384); hipMalloc(&o, 64 * 16);
    hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
    for (int off : {0, 4, 8, 12}) {
        K<<<1, 64>>>(d, o, off);
        hipError_t e = hipDeviceSynchronize();
        uint4 r[64];
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; ++t) {
            const unsigned *w = h.data() + (off + t * 20) / 4;
            if (r[t].x != w[0] || r[t].y != w[1] || r[t].z != w[2] || r[t].w != w[3]) ++bad;
        }
        printf("offset %2d: %s, %d mismatching lanes\n", off, hipGetErrorString(e), bad);
    }
    return 0;
}
