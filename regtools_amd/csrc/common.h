// common.h -- shared definitions for the regtools_amd HIP path.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RGX_HD __host__ __device__ __forceinline__
#define RGX_D __device__ __forceinline__
#define RGX_COLD __host__ __device__ __attribute__((noinline))      // rare and heavy: out of line, so that the hot loop around the call keeps its registers
#else
#define RGX_COLD __attribute__((noinline))
#define RGX_HD inline
#define RGX_D inline
#endif

namespace rgx {

// Unaligned little-endian loads. gfx950 global memory handles any byte alignment in hardware
// (unaligned access mode is on under HSA); the aligned(1)/may_alias typedefs make that one load/store.
typedef uint64_t u64_unaligned __attribute__((aligned(1), may_alias));
typedef uint32_t u32_unaligned __attribute__((aligned(1), may_alias));
typedef uint16_t u16_unaligned __attribute__((aligned(1), may_alias));
RGX_HD uint32_t ld32(const uint8_t *p) { return *(const u32_unaligned *)p; }
RGX_HD uint64_t ld64(const uint8_t *p) { return *(const u64_unaligned *)p; }
RGX_HD uint16_t ld16(const uint8_t *p) { return *(const u16_unaligned *)p; }
RGX_HD void st64(uint8_t *p, uint64_t v) { *(u64_unaligned *)p = v; }
RGX_HD void st32(uint8_t *p, uint32_t v) { *(u32_unaligned *)p = v; }
RGX_HD void st16(uint8_t *p, uint16_t v) { *(u16_unaligned *)p = v; }
// 16-byte unaligned access (one global_load/store_dwordx4 on gfx950)
typedef uint32_t u32x4 __attribute__((vector_size(16)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(1), may_alias));
RGX_HD u32x4 ld128(const uint8_t *p) { return *(const u32x4_unaligned *)p; }
RGX_HD void st128(uint8_t *p, u32x4 v) { *(u32x4_unaligned *)p = v; }

// smallest of three unsigned words (one v_min3_u32 on gfx950)
RGX_HD uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t m = a < b ? a : b;
    return m < c ? m : c;
}

constexpr uint32_t kBgzfMaxBlock = 0x10000;  // htslib/bgzf.h:42 BGZF_MAX_BLOCK_SIZE

// one BGZF member as the device sees it
struct Member {
    uint64_t cpos;   // offset of the raw-DEFLATE payload in the compressed buffer (member start + 18)
    uint64_t upos;   // offset of the member's first inflated byte in the arena
    uint32_t clen;   // payload bytes (member length - 26)
    uint32_t isize;  // ISIZE footer (layout hint; the kernel verifies the real length against it)
};

}  // namespace rgx
