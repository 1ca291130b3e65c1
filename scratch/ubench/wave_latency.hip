// scratch/ubench/wave_latency.hip -- what ONE wave alone on its SIMD pays per instruction on gfx950: dependent chains
// against independent streams of the instruction kinds the sixel diffusion's step is made of (s_memtime around
// 512 instructions).  The diffusion is a serial chain executed by a single wave per SIMD: these numbers, not the
// many-wave issue rates of valu_rate.hip, price its step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP512(x) REP8(REP64(x))

#define BENCH(NAME, N, BODY)                                                         \
    __global__ void __launch_bounds__(64) NAME(uint64_t *out, uint32_t *sink, uint32_t seed) { \
        extern __shared__ uint32_t lds[];                                             \
        for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (i * 4 + 4) & 0xfffc;  \
        __syncthreads();                                                              \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 5, d = seed * 7, e = (threadIdx.x * 4) & 0xfffc; \
        uint32_t m = 0x00030003u;                                                     \
        uint32_t *gp = sink + 64 + threadIdx.x;                                       \
        uint32_t *gq = sink + 1024 + (threadIdx.x >> 1) * 1024;  /* 32 distinct pages/lines: a row per lane pair */ \
        uint32_t eq = ((threadIdx.x >> 1) * 997u) & 0x7fffu;      /* scattered bytes of a 32 KB table */ \
        uint64_t wide = threadIdx.x;                                                  \
        uint4 quad = make_uint4(0, 0, 0, 0);                                          \
        uint64_t t0, t1;                                                              \
        for (int rep = 0; rep < 3; ++rep) {                                           \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));  \
            asm volatile(BODY : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), [wide] "+v"(wide), [quad] "=&v"(quad) : [m] "v"(m), [gp] "v"(gp), [sp] "s"(sink), [gq] "v"(gq), [eq] "v"(eq) : "vcc", "s20", "s21", "memory"); \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));  \
        }                                                                             \
        if (threadIdx.x == 0) out[0] = (t1 - t0);                                     \
        sink[threadIdx.x] = a + b + c + d + e + (uint32_t)wide + quad.x + quad.w;                                        \
    }

// dependent chains
BENCH(dep_pk_add,     512, REP512("v_pk_add_i16 %0, %0, %[m] clamp\n\t"))
BENCH(ind4_pk_add,    512, REP64(REP8("v_pk_add_i16 %0, %0, %[m] clamp\n\tv_pk_add_i16 %1, %1, %[m] clamp\n\tv_pk_add_i16 %2, %2, %[m] clamp\n\tv_pk_add_i16 %3, %3, %[m] clamp\n\t")) )
BENCH(ind2_pk_add,    512, REP64(REP8("v_pk_add_i16 %0, %0, %[m] clamp\n\tv_pk_add_i16 %1, %1, %[m] clamp\n\t")))
BENCH(dep_add_u32,    512, REP512("v_add_u32 %0, %0, %[m]\n\t"))
BENCH(dep_and,        512, REP512("v_and_b32 %0, %0, %[m]\n\t"))
BENCH(dep_pk_mad,     512, REP512("v_pk_mad_u16 %0, %0, %[m], %0\n\t"))
BENCH(dep_perm,       512, REP512("v_perm_b32 %0, %0, %0, %[m]\n\t"))
BENCH(dep_dot2,       512, REP512("v_dot2_u32_u16 %0, %0, %[m], 0\n\t"))
BENCH(dep_dpp,        512, REP512("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"))
BENCH(dep_dpp_quad,   512, REP512("v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"))
BENCH(dep_add_then_dpp, 512, REP64(REP8("v_add_u32 %0, %0, %[m]\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t")))
BENCH(dep_cmp_cnd,    512, REP64(REP8("v_cmp_gt_u32 vcc, %[m], %0\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t")))
BENCH(dep_valu_salu,  512, REP64(REP8("v_add_u32 %0, %0, %[m]\n\ts_add_u32 s20, s20, 1\n\t")))
BENCH(ind_salu,       512, REP512("s_add_u32 s20, s20, 1\n\t"))
BENCH(dep_valu_2salu, 512, REP64(REP8("v_add_u32 %0, %0, %[m]\n\ts_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\t")))
// LDS: dependent byte reads (address = previous result), and the same with 4 independent VALU ops in the shadow
BENCH(dep_lds_u8,     64, REP64("ds_read_b32 %4, %4\n\ts_waitcnt lgkmcnt(0)\n\t"))
BENCH(dep_lds_shadow, 64, REP64("ds_read_b32 %4, %4\n\tv_add_u32 %0, %0, %[m]\n\tv_add_u32 %1, %1, %[m]\n\tv_add_u32 %2, %2, %[m]\n\tv_add_u32 %3, %3, %[m]\n\ts_waitcnt lgkmcnt(0)\n\t"))
BENCH(lds_2reads,     64, REP64("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:32768\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %4, 0xfffc, %4\n\t"))

// shader clock of a kernel that keeps `blocks` single-wave workgroups spinning for ~ms: s_memtime (shader clocks) against
// s_memrealtime (100 MHz)
__global__ void __launch_bounds__(64) ClockProbe(uint64_t *out, int iters) {
    uint64_t c0, c1, r0, r1;
    uint32_t a = threadIdx.x;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
    for (int i = 0; i < iters; ++i) asm volatile(REP64("v_add_u32 %0, %0, 1\n\t") : "+v"(a));
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = a; }
}


// issue cost of instructions whose results nobody waits for inside the timed stretch (the harness drains them after it)
BENCH(iss_gload,      64, REP64("global_load_dword %1, %[gp], off\n\t"))
BENCH(iss_gload_s,    64, REP64("global_load_ubyte %1, %4, %[sp]\n\t"))
BENCH(iss_gstore,     64, REP64("global_store_dword %[gp], %0, off\n\t"))
BENCH(iss_gstore_s,   64, REP64("global_store_dword %4, %0, %[sp]\n\t"))
BENCH(iss_ds_read,    64, REP64("ds_read_b32 %1, %4\n\t"))
BENCH(iss_ds_read2,   64, REP64("ds_read2_b32 %[wide], %4 offset0:1 offset1:2\n\t"))
BENCH(iss_ds_write16, 64, REP64("ds_write_b16 %4, %0 offset:8\n\t"))
BENCH(iss_ds_write32, 64, REP64("ds_write_b32 %4, %0 offset:8\n\t"))
BENCH(iss_lshl_add64, 512, REP512("v_lshl_add_u64 %[wide], %[wide], 2, %[wide]\n\t"))
BENCH(iss_med3,       512, REP512("v_med3_i32 %0, %0, 0, %[m]\n\t"))
BENCH(iss_readfirst,  512, REP512("v_readfirstlane_b32 s20, %0\n\t"))
BENCH(iss_saveexec,   512, REP64(REP8("s_and_saveexec_b64 s[20:21], vcc\n\ts_cbranch_execz 1f\n\tv_add_u32 %0, %0, %[m]\n\t1: s_or_b64 exec, exec, s[20:21]\n\t")))
BENCH(iss_br_nottaken,512, REP64(REP8("s_cmp_eq_u32 s20, 12345\n\ts_cbranch_scc1 2f\n\t")) "2:\n\t")
BENCH(iss_br_taken,   64, REP64("s_branch 3f\n\tv_add_u32 %0, %0, %[m]\n\t3:\n\t"))
BENCH(iss_cmp_cnd_nop,1024, REP64(REP8("v_cmp_gt_u32 vcc, %[m], %1\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t")))
BENCH(iss_sleep1,     64, REP64("s_sleep 1\n\t"))

BENCH(iss_gload_rows,  64, REP64("global_load_dword %1, %[gq], off\n\t"))
BENCH(iss_gload4_rows, 64, REP64("global_load_dwordx4 %[quad], %[gq], off\n\t"))
BENCH(iss_gbyte_scat,  64, REP64("global_load_ubyte %1, %[eq], %[sp]\n\t"))
BENCH(iss_gstore_rows, 64, REP64("global_store_dword %[gq], %0, off\n\t"))
BENCH(gload_rows_32apart, 32, REP8(REP8("v_add_u32 %0, %0, %[m]\n\t") REP8("v_add_u32 %0, %0, %[m]\n\t") "global_load_dword %1, %[gq], off\n\tglobal_load_ubyte %2, %[eq], %[sp]\n\t") )

struct Case { const char *name; void (*fn)(uint64_t *, uint32_t *, uint32_t); int n; };
#define C(NAME, N) {#NAME, NAME, N}
int main() {
    Case cases[] = {C(dep_pk_add, 512), C(ind2_pk_add, 1024), C(ind4_pk_add, 2048), C(dep_add_u32, 512), C(dep_and, 512), C(dep_pk_mad, 512),
                    C(dep_perm, 512), C(dep_dot2, 512), C(dep_dpp, 512), C(dep_dpp_quad, 512), C(dep_add_then_dpp, 1024),
                    C(dep_cmp_cnd, 1024), C(dep_valu_salu, 1024), C(ind_salu, 512), C(dep_valu_2salu, 1536),
                    C(dep_lds_u8, 64), C(dep_lds_shadow, 64), C(lds_2reads, 64),
                    C(iss_gload, 64), C(iss_gload_s, 64), C(iss_gstore, 64), C(iss_gstore_s, 64), C(iss_ds_read, 64), C(iss_ds_read2, 64),
                    C(iss_ds_write16, 64), C(iss_ds_write32, 64), C(iss_lshl_add64, 512), C(iss_med3, 512), C(iss_readfirst, 512),
                    C(iss_saveexec, 2048), C(iss_br_nottaken, 1024), C(iss_br_taken, 64), C(iss_cmp_cnd_nop, 1536), C(iss_sleep1, 64),
                    C(iss_gload_rows, 64), C(iss_gload4_rows, 64), C(iss_gbyte_scat, 64), C(iss_gstore_rows, 64), C(gload_rows_32apart, 8 * 18)};
    uint64_t *out; uint32_t *sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 1 << 20);
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int blocks : {1, 256}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(ClockProbe, dim3(blocks), dim3(64), 0, 0, out, 4000);
            uint64_t v[3]; hipMemcpy(v, out, 24, hipMemcpyDeviceToHost);
            printf("clock probe, %4d one-wave workgroups: %llu shader clocks in %llu x 10 ns = %.0f MHz; %.2f clocks per v_add\n", blocks,
                   (unsigned long long)v[0], (unsigned long long)v[1], v[0] / (v[1] * 0.01), (double)v[0] / (4000.0 * 64));
        }
    }
    for (auto &c : cases) {
        hipFuncSetAttribute((const void *)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        uint64_t best = ~0ull;
        for (int r = 0; r < 5; ++r) {
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 65536, 0, out, sink, 12345u + r);
            uint64_t v; hipMemcpy(&v, out, 8, hipMemcpyDeviceToHost);
            if (v < best) best = v;
        }
        // s_memtime counts at 100 MHz on gfx9-family parts: report raw ticks and ns per instruction
        printf("%-18s %6llu clocks / %4d instr = %6.2f shader clocks (s_memtime) per instr\n", c.name, (unsigned long long)best, c.n, (double)best / c.n);
    }
    return 0;
}
