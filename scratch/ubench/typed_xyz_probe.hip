#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f3 __attribute__((ext_vector_type(3)));
typedef int i4 __attribute__((ext_vector_type(4)));
__global__ void K(const uint8_t *src, float *out) {
    const uint64_t a = (uint64_t)src;
    i4 rs; rs.x = (int)(uint32_t)a; rs.y = (int)(uint32_t)(a >> 32) & 0xffff; rs.z = 4096;
    rs.w = (int)((4u | (5u << 3) | (6u << 6) | (7u << 9)) | (2u << 12) | (10u << 15));
    unsigned o0 = threadIdx.x * 4, o1 = o0 + 256, o2 = o0 + 512, o3 = o0 + 768;
    float r[16];
    // four xyz loads into v[a:a+2] with a poisoned 4th register each
    asm volatile(
        "v_mov_b32 v43, 0x42f60000\n\tv_mov_b32 v47, 0x42f60000\n\tv_mov_b32 v51, 0x42f60000\n\tv_mov_b32 v55, 0x42f60000\n\t"
        "buffer_load_format_xyz v[40:42], %16, %20, 0 offen\n\t"
        "buffer_load_format_xyz v[44:46], %17, %20, 0 offen\n\t"
        "buffer_load_format_xyz v[48:50], %18, %20, 0 offen\n\t"
        "buffer_load_format_xyz v[52:54], %19, %20, 0 offen\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43\n\t"
        "v_mov_b32 %4, v44\n\tv_mov_b32 %5, v45\n\tv_mov_b32 %6, v46\n\tv_mov_b32 %7, v47\n\t"
        "v_mov_b32 %8, v48\n\tv_mov_b32 %9, v49\n\tv_mov_b32 %10, v50\n\tv_mov_b32 %11, v51\n\t"
        "v_mov_b32 %12, v52\n\tv_mov_b32 %13, v53\n\tv_mov_b32 %14, v54\n\tv_mov_b32 %15, v55"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
          "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
        : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(rs)
        : "memory", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = r[i];
}
int main() {
    std::vector<uint8_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (uint8_t)((i * 7 + (i >> 8)) & 255);
    uint8_t *d; float *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 64 * 16 * 4);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    K<<<1, 64>>>(d, o);
    std::vector<float> g(64 * 16);
    hipMemcpy(g.data(), o, g.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0, poison_lost = 0;
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            for (int c = 0; c < 3; ++c) bad += g[l * 16 + k * 4 + c] != (float)h[l * 4 + k * 256 + c];
            poison_lost += g[l * 16 + k * 4 + 3] != 123.0f;
        }
    printf("xyz loads: %d wrong components of 768, %d of 256 poisoned fourth registers overwritten (lane 0: %g %g %g | %g ; expect %d %d %d | 123)\n",
           bad, poison_lost, g[0], g[1], g[2], g[3], h[0], h[1], h[2]);
    return 0;
}
