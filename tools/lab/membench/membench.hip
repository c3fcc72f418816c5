// membench.hip -- cost model of the vector-memory pipeline for the lane-per-member access pattern (round 4 lab, not product code).
// 256 x W single-wave workgroups; every wave issues ITER vector-memory instructions of one kind and pattern; reports ns per wave-instruction per CU
// (= time x CUs x waves-in-flight-normalised) so that "what does a sparse / scattered / coalesced dwordx4 cost the CU" can be read off.
//   hipcc --offload-arch=gfx950 -O3 -o membench membench.hip && ./membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum Op { LOAD = 0, STORE = 1, ST_LD_FAR = 2, ST_LD_NEAR = 3, LD_ST_DEFER = 4 };
enum Pat { COAL = 0, GROUP16 = 1, SCATTER = 2, SCATTER_UNAL = 3 };

// lds_pad forces W waves per CU
template <int OP, int PAT>
__global__ __launch_bounds__(64) void k_mem(uint8_t *buf, uint64_t lane_mask, uint32_t iters, uint32_t wait_every, uint32_t region /* bytes per lane */, uint32_t *sink) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const bool on = (lane_mask >> lane) & 1;
    uint8_t *base = buf + (size_t)wave * 64 * region;
    uint8_t *p;
    uint32_t step = 16, wrap = region;
    if (PAT == COAL) { p = base + 16 * lane; step = 1024; wrap = 64 * region; }
    else if (PAT == GROUP16) { p = base + (size_t)(lane >> 4) * 16 * region + 16 * (lane & 15); step = 256; wrap = 16 * region; }
    else { p = base + (size_t)lane * region + (PAT == SCATTER_UNAL ? 5 : 0); step = 16; wrap = region - 32; }
    u32x4 acc = {lane, 1, 2, 3};
    uint32_t off = 0;
    // U operations are issued back to back (each load into registers of its own, live until the wait), then ONE wait: U = 1 measures a
    // dependent chain (latency / waves), U = 8 the pipeline's throughput
#define MB_BODY(U)                                                                                                                   \
    for (uint32_t i = 0; i < iters; i += U) {                                                                                        \
        u32x4 ld[U];                                                                                                                 \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                                              \
            ld[u] = u32x4{0, 0, 0, 0};                                                                                               \
            if (on) {                                                                                                                \
                if (OP == LOAD) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ld[u]) : "v"(p + off) : "memory"); }        \
                else if (OP == STORE) { asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p + off), "v"(acc) : "memory"); }     \
                else if (OP == ST_LD_FAR) {                                                                                          \
                    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p + off), "v"(acc) : "memory");                           \
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ld[u]) : "v"(p + ((off + wrap / 2) % wrap)) : "memory");  \
                } else if (OP == ST_LD_NEAR) {                                                                                       \
                    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p + off), "v"(acc) : "memory");                           \
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ld[u]) : "v"(p + (off >= 208 ? off - 208 : 0)) : "memory"); \
                }                                                                                                                    \
            }                                                                                                                        \
            off += step; if (off >= wrap) off = 0;                                                                                   \
        }                                                                                                                            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                             \
        _Pragma("unroll") for (int u = 0; u < U; ++u) { asm volatile("" : "+v"(ld[u])); acc[1] ^= ld[u][0]; }                        \
    }
    if (wait_every == 1) { MB_BODY(1) } else if (wait_every == 4) { MB_BODY(4) } else { MB_BODY(8) }
#undef MB_BODY
    if (acc[0] == 0x12345678 && acc[1] == 77) sink[0] = acc[2];
    if (lds[0] == 12345) sink[1] = 1;
}

template <int OP, int PAT>
static void run(const char *name, uint8_t *buf, uint32_t *sink, int waves_per_cu, uint64_t mask, uint32_t iters, uint32_t wait_every, uint32_t region) {
    const int cus = 256;
    const uint32_t lds_bytes = (160 * 1024 / waves_per_cu) & ~255u;
    hipFuncSetAttribute((const void *)k_mem<OP, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mem<OP, PAT>), dim3(cus * waves_per_cu), dim3(64), lds_bytes, 0, buf, mask, iters, wait_every, region, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const int ninstr = (OP >= ST_LD_FAR) ? 2 : 1;
    const double ns_per_instr_cu = best * 1e6 / ((double)iters * waves_per_cu * ninstr);   // CU-time per wave-instruction
    const double ns_per_iter_wave = best * 1e6 / iters;                                        // latency of one iteration of one wave
    printf("%-28s waves/CU %2d lanes %2d wait_every %3u region %6u: %8.3f ms  %7.1f ns/instr/CU  %8.1f ns/iter/wave  %7.1f GB/s\n", name, waves_per_cu, __builtin_popcountll(mask),
           wait_every, region, best, ns_per_instr_cu, ns_per_iter_wave, (double)iters * cus * waves_per_cu * ninstr * __builtin_popcountll(mask) * 16 / best / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)256 * 12 * 64 * 65536;      // 12.9 GB: a 64 KiB region per lane at 12 waves per CU
    uint8_t *buf; uint32_t *sink;
    if (hipMalloc(&buf, bytes + 4096) != hipSuccess) { fprintf(stderr, "alloc\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, bytes);
    const uint64_t all = ~0ull, m16 = 0xffffull, m11 = 0x0000092492492491ull & 0x7ffffffffffull, m4 = 0x0001000100010001ull, m1 = 1ull;
    const uint32_t it = 2000;
    for (int w : {12, 3}) {
        printf("== %d waves per CU\n", w);
        for (uint32_t we : {8u, 1u}) {
            for (uint32_t region : {65536u, 1024u}) {
                for (uint64_t m : {all, m16, m11, m4, m1}) {
                    run<LOAD, SCATTER>("load scatter", buf, sink, w, m, it, we, region);
                    run<STORE, SCATTER>("store scatter", buf, sink, w, m, it, we, region);
                }
                run<LOAD, SCATTER_UNAL>("load scatter unaligned", buf, sink, w, all, it, we, region);
                run<STORE, SCATTER_UNAL>("store scatter unaligned", buf, sink, w, all, it, we, region);
                run<LOAD, GROUP16>("load 4x16 groups", buf, sink, w, all, it, we, region);
                run<STORE, GROUP16>("store 4x16 groups", buf, sink, w, all, it, we, region);
                run<LOAD, GROUP16>("load 1x16 group", buf, sink, w, m16, it, we, region);
                run<STORE, GROUP16>("store 1x16 group", buf, sink, w, m16, it, we, region);
                run<LOAD, COAL>("load coalesced", buf, sink, w, all, it, we, region);
                run<STORE, COAL>("store coalesced", buf, sink, w, all, it, we, region);
                run<ST_LD_FAR, SCATTER>("store+load far", buf, sink, w, all, it, we, region);
                run<ST_LD_NEAR, SCATTER>("store+load near (RAW)", buf, sink, w, all, it, we, region);
                run<ST_LD_FAR, SCATTER>("store+load far 11 lanes", buf, sink, w, m11, it, we, region);
                run<ST_LD_NEAR, SCATTER>("store+load near 11 lanes", buf, sink, w, m11, it, we, region);
            }
        }
    }
    return 0;
}
