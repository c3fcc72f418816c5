// micro-benchmark: per-lane private-region copies with 16-byte accesses at different byte alignments.
// Question: does gfx950's L1 (TCP) split byte-unaligned dwordx4 accesses into several accesses?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((vector_size(16)));
typedef u32x4 u32x4_u __attribute__((aligned(1), may_alias));
__global__ void k_copy(uint8_t *buf, uint32_t region, uint32_t iters, uint32_t off_ld, uint32_t off_st, uint32_t lanes) {
    uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= lanes) return;
    uint8_t *base = buf + (size_t)l * region;
    for (uint32_t i = 0; i < iters; ++i) {
        u32x4 v = *(const u32x4_u *)(base + 256 + (size_t)i * 16 - 224 + off_ld);
        *(u32x4_u *)(base + 256 + (size_t)i * 16 + off_st) = v;
    }
}
int main() {
    const uint32_t region = 66048, lanes = 256 * 448, iters = 4000;
    uint8_t *buf; hipMalloc(&buf, (size_t)region * lanes + 4096);
    hipMemset(buf, 1, (size_t)region * lanes + 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t cfg[][2] = {{0, 0}, {0, 0}, {4, 4}, {1, 1}, {3, 7}, {0, 1}, {1, 0}, {8, 8}};
    for (auto &c : cfg) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_copy, dim3(lanes / 64), dim3(64), 0, 0, buf, region, iters, c[0], c[1], lanes);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("ld+%u st+%u : %.3f ms  %.1f GB/s (r+w)\n", c[0], c[1], ms, 2.0 * iters * 16.0 * lanes / ms / 1e6);
    }
    return 0;
}
