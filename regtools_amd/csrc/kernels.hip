// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the junctions-extract hot path.
//
// Everything here is integer/byte work bounded by HBM traffic or (for DEFLATE) by per-lane latency; there is
// deliberately no MFMA.  Wave width is 64 everywhere (ballot masks are 64-bit).
#include "kernels.h"

#include <mutex>
#include <type_traits>

#include "bam_core.h"
#include "inflate_core.h"
#include "inflate_coop.h"
#include "inflate_ring.h"
#include "inflate_wave.h"

namespace rgx {

// =====================================================================================================
// wave helpers
// =====================================================================================================
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}

// =====================================================================================================
// a1. BGZF inflate: ONE LANE PER MEMBER (64 members per wavefront, one wave per workgroup)
// =====================================================================================================
// Per-lane Huffman symbol lists, lane-interleaved so that lane L only ever touches LDS bank L % 32, bit-packed:
//   ll_sym : 288 x 9 bits, sorted by (code length, symbol): the first kHotSyms entries -- the SHORTEST, i.e. most frequent,
//            codes -- in LDS dwords 0..kLdsD-1, the long-code tail in the global scratch (one dependent L2 read per rare symbol)
//   d_sym  :  32 x 5 bits -> LDS dwords kLdsD..kLdsD+4                    + 1 pad dword
// What the split buys is occupancy: LDS per wave = kLdsDwordsPerLane * 256 B decides how many waves (each 64 members) a CU of
// 160 KiB holds, and a lane's trip is a chain of dependent ALU/LDS steps plus one memory round trip -- the other waves of
// the SIMD are what fills those gaps (measured with the output stage in place: 5 -> 7 waves per CU is 1.3x).
// The canonical codes themselves (bounds, lengths, list offsets) are 2 x 16 REGISTER words (inflate_core.h Code); the
// code-length scratch that only a block header needs is global too (40 dwords per lane, before the cold symbols).
#ifndef RGX_LAB_HOT_SYMS
#define RGX_LAB_HOT_SYMS 160
#endif
constexpr uint32_t kHotSyms = RGX_LAB_HOT_SYMS;
constexpr uint32_t kLdsLL = 0, kLdsD = (kHotSyms * 9 + 31) / 32, kLdsDwordsPerLane = kLdsD + 6;
constexpr uint32_t kInflateLdsBytes = kLdsDwordsPerLane * 64 * 4;
// (round 4: the cold end of the list is 16 bits per symbol -- ONE request per lookup, no read-modify-write when the list is built)
constexpr uint32_t kScratchLenWords = 40, kScratchColdWords = (288 - kHotSyms + 1) / 2;
constexpr uint32_t kScratchWordsPerLane = kScratchLenWords + kScratchColdWords;

struct LdsTab {
    static constexpr bool kIsHandle = true;     // (pointers into LDS and scratch: block_header_call copies the handle, not tables)
    uint32_t *base;       // LDS, already offset by lane
    uint32_t *scratch;    // global, already offset by global lane
    uint32_t stride;      // lanes in the grid (dword stride of the scratch)
    __device__ __forceinline__ uint32_t rd(uint32_t dw) const { return base[dw * 64]; }
    __device__ __forceinline__ void wr(uint32_t dw, uint32_t v) { base[dw * 64] = v; }
    __device__ __forceinline__ uint32_t grd(uint32_t dw) const { return scratch[(size_t)dw * stride]; }
    __device__ __forceinline__ void gwr(uint32_t dw, uint32_t v) { scratch[(size_t)dw * stride] = v; }
    __device__ __forceinline__ uint32_t get_bits(uint32_t dw0, uint32_t bit, uint32_t width) const {
        const uint32_t dw = dw0 + (bit >> 5), sh = bit & 31;
        const uint64_t w = (uint64_t)rd(dw) | (uint64_t)rd(dw + 1) << 32;
        return (uint32_t)(w >> sh) & ((1u << width) - 1u);
    }
    __device__ __forceinline__ void set_bits(uint32_t dw0, uint32_t bit, uint32_t width, uint32_t v) {
        const uint32_t dw = dw0 + (bit >> 5), sh = bit & 31;
        uint64_t w = (uint64_t)rd(dw) | (uint64_t)rd(dw + 1) << 32;
        const uint64_t m = (uint64_t)((1u << width) - 1u) << sh;
        w = (w & ~m) | ((uint64_t)v << sh);
        wr(dw, (uint32_t)w);
        if (sh + width > 32) wr(dw + 1, (uint32_t)(w >> 32));
    }
    // cold symbols: two per dword of the lane's scratch column
    __device__ __forceinline__ uint32_t get_cold(uint32_t c) const { return (grd(kScratchLenWords + (c >> 1)) >> (16u * (c & 1u))) & 0x1ffu; }
    __device__ __forceinline__ void set_cold(uint32_t c, uint32_t v) {
        ((uint16_t *)(scratch + (size_t)(kScratchLenWords + (c >> 1)) * stride))[c & 1u] = (uint16_t)v;
    }
    __device__ __forceinline__ uint32_t get_ll_sym(uint32_t i) const { return i < kHotSyms ? get_bits(kLdsLL, i * 9, 9) : get_cold(i - kHotSyms); }
    __device__ __forceinline__ void set_ll_sym(uint32_t i, uint32_t v) { if (i < kHotSyms) set_bits(kLdsLL, i * 9, 9, v); else set_cold(i - kHotSyms, v); }
    __device__ __forceinline__ uint32_t get_d_sym(uint32_t i) const { return get_bits(kLdsD, i * 5, 5); }
    __device__ __forceinline__ void set_d_sym(uint32_t i, uint32_t v) { set_bits(kLdsD, i * 5, 5, v); }
    __device__ __forceinline__ uint32_t get_len_word(uint32_t w) const { return grd(w); }
    __device__ __forceinline__ void set_len_word(uint32_t w, uint32_t v) { gwr(w, v); }
    __device__ __forceinline__ void clear_syms() {}
};

// PROBE = false: the pipeline's launch -- member m goes to its planned place arena + upos (planned from the ISIZE footers) and must
// inflate to exactly ISIZE bytes.  PROBE = true: the repair launch for files whose footers lie (bgzf.c:292-316 never reads ISIZE: a
// block is as long as zlib says): every member into its own 64 KiB slot, true length (or ~0 = does not inflate) to sizes[m].
// PIECE: the same code under a second symbol -- the launches of rgx_extract_mem's overlapped upload cover a third of a file each and
// run side by side; a profiler's per-kernel statistics keep them apart from the whole-range launches the roofline is quoted on.
template <bool PROBE, bool PIECE = false, int LITS = 1 /* literals per trip (inflate_core.h): 4 for payloads that are mostly literals */>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_inflate(const uint8_t *__restrict__ comp, const Member *__restrict__ members,
                                                uint32_t n_members, uint8_t *__restrict__ arena, uint64_t upos_bias, uint32_t *len_scratch,
                                                uint32_t *status, uint32_t ignore_below, uint32_t index_bias, uint8_t *bad) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t m = blockIdx.x * 64 + threadIdx.x;
    if (m >= n_members) return;
    Member mb = members[m];
    LdsTab T{lds + threadIdx.x, len_scratch + m, gridDim.x * 64};
    uint32_t out_len = 0;
    if (PROBE) {
        if (mb.isize == 0xffffffffu) { status[m] = 0xffffffffu; return; }   // k_member_link: BSIZE runs past the end of the file (or is < 26): the read fails upstream
        const int st = inflate_raw(comp + mb.cpos, mb.clen, arena + (uint64_t)m * kBgzfMaxBlock, kBgzfMaxBlock, &out_len, T);
        status[m] = st == INF_OK ? out_len : 0xffffffffu;
        return;
    }
    // a member whose claimed size is no BGZF block size owns no bytes of the arena (k_member_compact): it must not write any, whatever
    // range the host asked for.  ~0 = the member runs past the end of the file (k_member_link).
    int st;
    if (mb.isize > kBgzfMaxBlock) st = mb.isize == 0xffffffffu ? INF_IN_OVERRUN : INF_OUT_OVERFLOW;
    else {
        st = inflate_raw<LITS>(comp + mb.cpos, mb.clen, arena + (mb.upos - upos_bias), mb.isize, &out_len, T);
        if (st == INF_OK && out_len != mb.isize) st = INF_SIZE_MISMATCH;
    }
    if (st != INF_OK) {
        // members in front of a seek target are inflated for the header only: their failures do not end the record stream and are
        // reported apart (the header read and the footer check still want to know)
        const uint32_t mi = m + index_bias;       // (index in the caller's member range: a range may be launched in several pieces)
        if (bad) bad[mi] = 1;
        uint32_t *slot = mi >= ignore_below ? status : status + kStatusEarly;
        uint32_t prev = atomicMin(&slot[0], mi);
        if (mi < prev) slot[1] = (uint32_t)st;    // best effort: status of (one of) the earliest bad members
    }
}

// ---- round 2: the same decoder with its output in LDS (inflate_ring.h) --------------------------------------------------------------
// LDS of one wave (= one workgroup): [symbol lists, kLdsDwordsPerLane x 64 dwords][rings, kRingLaneDw x 64 dwords][flush work list, 128 x 4].
// Everything is dword-interleaved across the lanes (dword d of lane L at d * 64 + L), so a lane only ever touches bank L % 32 and
// consecutive dwords of a lane are one ds_read2st64_b32 / ds_write2st64_b32 apart.  159 dwords per lane = 40 704 B per wave: four waves per CU.
constexpr uint32_t kRingLdsDwords = (kLdsDwordsPerLane + kRingLaneDw + 8) * 64;      // (flush list: 128 items of 4 dwords)
constexpr uint32_t kRingLdsBytes = kRingLdsDwords * 4;

struct LdsRing {
    uint32_t *base;       // LDS, already offset by lane
    __device__ __forceinline__ uint32_t rd(uint32_t dw) const { return base[dw * 64]; }
    __device__ __forceinline__ void rd4(uint32_t dw, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d) const {
        const uint32_t *q = base + dw * 64;
        a = q[0]; b = q[64]; c = q[128]; d = q[192];
    }
    __device__ __forceinline__ void wr4(uint32_t dw, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        uint32_t *q = base + dw * 64;
        q[0] = a; q[64] = b; q[128] = c; q[192] = d;
    }
    __device__ __forceinline__ void wr8(uint32_t byte_pos, uint32_t b) { ((uint8_t *)(base + (byte_pos >> 2) * 64))[byte_pos & 3u] = (uint8_t)b; }
};

// The wave as a flushing machine.  A flush round: every lane with complete lines appends one 16-byte work item per line {line base
// address, position, ring line, member} to a list in LDS (slots from ballots: no atomics); then lane j stores piece j & 7 (16 bytes) of
// item 8 * step + (j >> 3): one store instruction = eight whole, aligned 128-byte lines, and only as many instructions as there are lines.
// Four steps are read before the first is stored, so a round waits on LDS twice, not per line.  (A member's ring lives in ONE bank, so
// the eight lanes of a line take turns on it: 32 LDS cycles per eight lines -- LDS time well spent: what HBM sees is each output byte
// once, in full lines.)
struct WaveCoop {
    uint32_t *ring_wave;  // LDS: dword d of lane L's ring at ring_wave[d * 64 + L]
    uint32_t *list;       // LDS: 128 work items of 4 dwords
    uint32_t lane;
    __device__ __forceinline__ bool any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0; }
    __device__ __forceinline__ void flush_lines(const LdsRing &R, RingOut &O) const {
        if (O.fl & 127u) {                                        // the partial line in front of the member's first boundary: this lane alone
            const uint32_t c = (O.fl + 127u) & ~127u;
            if (O.hd >= c) { O.slow_flush(R, O.fl, c); O.fl = c; }
        }
        const uint32_t nl = (O.fl & 127u) ? 0u : (O.hd - O.fl) >> 7;            // 0..2 (head - flushed <= kRingFill)
        const uint64_t m0 = __builtin_amdgcn_ballot_w64(nl > 0), m1 = __builtin_amdgcn_ballot_w64(nl > 1);
        const uint32_t c0 = (uint32_t)__popcll(m0), total = c0 + (uint32_t)__popcll(m1);
        if (!total) return;
        const uint64_t lt = (1ull << lane) - 1ull;
        {
            const uint64_t a = (uint64_t)(uintptr_t)O.lbase;
            uint32_t line = (O.fl % kRingBytes) >> 7;
            if (nl > 0) { const u32x4 e = {(uint32_t)a, (uint32_t)(a >> 32), O.fl, line << 8 | lane}; *(u32x4 *)(list + 4 * (uint32_t)__popcll(m0 & lt)) = e; }
            if (nl > 1) {
                if (++line >= 3) line = 0;
                const u32x4 e = {(uint32_t)a, (uint32_t)(a >> 32), O.fl + 128, line << 8 | lane}; *(u32x4 *)(list + 4 * (c0 + (uint32_t)__popcll(m1 & lt))) = e;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t piece = lane & 7u, sub = lane >> 3;
        for (uint32_t base = 0; base < total; base += 32) {
            u32x4 e[4], v[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const uint32_t k = base + 8 * s4 + sub;
                e[s4] = *(const u32x4 *)(list + 4 * (k < total ? k : total - 1));
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const uint32_t *q = ring_wave + ((e[s4][3] >> 8) * 32 + piece * 4) * 64 + (e[s4][3] & 0xffu);
                v[s4] = u32x4{q[0], q[64], q[128], q[192]};
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (base + 8 * s4 + sub < total) {
                    uint8_t *dst = (uint8_t *)(uintptr_t)((uint64_t)e[s4][0] | (uint64_t)e[s4][1] << 32) + e[s4][2] + 16 * piece;
                    *(u32x4 *)dst = v[s4];
                }
            }
        }
        O.fl += 128 * nl;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();                             // (the list is rewritten by the next round)
    }
};

template <bool PROBE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_inflate_ring(const uint8_t *__restrict__ comp, const Member *__restrict__ members,
                                                uint32_t n_members, uint8_t *arena, uint64_t upos_bias, uint32_t *len_scratch,
                                                uint32_t *status, uint32_t ignore_below, uint32_t index_bias, uint8_t *bad) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t m = blockIdx.x * 64 + lane;
    const bool have = m < n_members;
    Member mb = members[have ? m : n_members - 1];
    LdsTab T{lds + lane, len_scratch + (have ? m : 0), gridDim.x * 64};
    LdsRing R{lds + kLdsDwordsPerLane * 64 + lane};
    WaveCoop C{lds + kLdsDwordsPerLane * 64, lds + (kLdsDwordsPerLane + kRingLaneDw) * 64, lane};
    uint32_t out_len = 0;
    if (PROBE) {
        const bool run = have && mb.isize != 0xffffffffu;         // ~0: BSIZE runs past the end of the file (or is < 26): the read fails upstream
        const int st = inflate_ring(comp + mb.cpos, mb.clen, arena + (uint64_t)(have ? m : 0) * kBgzfMaxBlock, kBgzfMaxBlock, &out_len, T, R, C, run);
        if (have) status[m] = run && st == INF_OK ? out_len : 0xffffffffu;
        return;
    }
    // a member whose claimed size is no BGZF block size owns no bytes of the arena (k_member_compact): it must not write any, whatever
    // range the host asked for.  ~0 = the member runs past the end of the file (k_member_link).
    const bool run = have && mb.isize <= kBgzfMaxBlock;
    int st = inflate_ring(comp + mb.cpos, mb.clen, arena + (run ? mb.upos - upos_bias : 0), run ? mb.isize : 0, &out_len, T, R, C, run);
    if (!have) return;
    if (!run) st = mb.isize == 0xffffffffu ? INF_IN_OVERRUN : INF_OUT_OVERFLOW;
    else if (st == INF_OK && out_len != mb.isize) st = INF_SIZE_MISMATCH;
    if (st != INF_OK) {
        const uint32_t mi = m + index_bias;
        if (bad) bad[mi] = 1;
        uint32_t *slot = mi >= ignore_below ? status : status + kStatusEarly;
        uint32_t prev = atomicMin(&slot[0], mi);
        if (mi < prev) slot[1] = (uint32_t)st;
    }
}

// ---- round 3: long copies moved by the wave (inflate_coop.h) ---------------------------------------------------------------------------
// The wave as a copying machine.  A round serves up to four pending long copies, sixteen lanes each: the owners' lane numbers come from a
// ballot (scalar bit scans), every lane pulls its group's copy -- destination of the first whole chunk (relative to the wave's lowest output
// address: the members of a wave lie next to each other in the arena), distance, chunk count -- with two ds_bpermute, loads 16 bytes from
// (chunk j) - distance and, after the trip's one wait, stores them to chunk j: aligned, and 256 consecutive bytes per group.  Up to
// four rounds are in flight per trip (one per four lanes that start a long match in the same trip; the bench payload averages two);
// a trip with more finishes the earlier ones on the spot.
struct WaveCopy {                     // (kCoopRounds = 4 rounds in flight: d0..d3 / t0..t3)
    uint8_t *wave_base;
    uint32_t lane;
    u32x4 d0, d1, d2, d3;
    uint32_t t0, t1, t2, t3;          // destination (relative) of what d* holds, ~0 = nothing
    __device__ __forceinline__ bool any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0; }
    __device__ __forceinline__ void stores() {
        if (t0 != 0xffffffffu) *(u32x4 *)(wave_base + t0) = d0;
        if (t1 != 0xffffffffu) *(u32x4 *)(wave_base + t1) = d1;
        if (t2 != 0xffffffffu) *(u32x4 *)(wave_base + t2) = d2;
        if (t3 != 0xffffffffu) *(u32x4 *)(wave_base + t3) = d3;
        t0 = t1 = t2 = t3 = 0xffffffffu;
    }
    __device__ __forceinline__ void begin(bool want, uint8_t *out, uint32_t o_body, uint32_t dist, uint32_t nb) {
        uint64_t mask = __builtin_amdgcn_ballot_w64(want);
        if (!mask) return;
        const uint32_t rel = (uint32_t)(out - wave_base) + o_body, packed = dist | nb << 16;
        const uint32_t g = lane >> 4, j = lane & 15u;
        for (;;) {
#define RGX_ROUND(D, Tg)                                                                                                         \
            if (mask) {                                                                                                          \
                const int l0 = __builtin_ctzll(mask); mask &= mask - 1;                                                          \
                const int l1 = mask ? __builtin_ctzll(mask) : -1; mask &= mask - 1;                                              \
                const int l2 = mask ? __builtin_ctzll(mask) : -1; mask &= mask - 1;                                              \
                const int l3 = mask ? __builtin_ctzll(mask) : -1; mask &= mask - 1;                                              \
                const int lg = g == 0 ? l0 : g == 1 ? l1 : g == 2 ? l2 : l3;                                                     \
                const uint32_t dd = (uint32_t)__builtin_amdgcn_ds_bpermute((lg < 0 ? 0 : lg) << 2, (int)rel);                    \
                const uint32_t pp = (uint32_t)__builtin_amdgcn_ds_bpermute((lg < 0 ? 0 : lg) << 2, (int)packed);                 \
                if (lg >= 0 && j < (pp >> 16)) { Tg = dd + 16u * j; D = ld128(wave_base + Tg - (pp & 0xffffu)); }                 \
            }
            RGX_ROUND(d0, t0) RGX_ROUND(d1, t1) RGX_ROUND(d2, t2) RGX_ROUND(d3, t3)
#undef RGX_ROUND
            if (!mask) break;
            __builtin_amdgcn_s_waitcnt(0x0F70);                   // more than four rounds in one trip: finish these now
            stores();
        }
    }
    __device__ __forceinline__ void end() { stores(); }
};

// Lane assignment: per group of kSortGroup consecutive members, their indices sorted by compressed length (bitonic sort in LDS, one
// workgroup per group).  A group spans at most kSortGroup x 64 KiB = 64 MiB of the arena: offsets relative to its first member fit 32 bits.
constexpr uint32_t kSortGroup = kInflateSortGroup;
__global__ __launch_bounds__(256) void k_member_sort(const Member *__restrict__ members, uint32_t n_members, uint32_t *perm) {
    __shared__ uint32_t key[kSortGroup];
    const uint32_t g0 = blockIdx.x * kSortGroup;
    for (uint32_t t = threadIdx.x; t < kSortGroup; t += 256) {
        const uint32_t i = g0 + t;
        key[t] = i < n_members ? (min(members[i].clen, 0x1fffffu) << 10 | t) : 0xffffffffu;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= kSortGroup; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < kSortGroup; t += 256) {
                const uint32_t x = t ^ j;
                if (x > t) {
                    const uint32_t a = key[t], b = key[x];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { key[t] = b; key[x] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t t = threadIdx.x; t < kSortGroup; t += 256) if (g0 + t < n_members) perm[g0 + t] = g0 + (key[t] & 1023u);
}

// What k_inflate_coop asks of a member list the PIPELINE did not make (rgx_k_inflate / rgx_k_inflate_form): its wave copies address the arena with
// 32-bit offsets from the first member of a group of kSortGroup, so upos must not decrease along the list and a group (plus one member's
// 64 KiB) must span less than 4 GiB.  A list that does not is refused: status[0] = the first offending member, status[1] = INF_OUT_OVERFLOW.
__global__ void k_members_check(const Member *__restrict__ members, uint32_t n_members, uint32_t *status, uint32_t *veto) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_members) return;
    const uint32_t first = (m / kSortGroup) * kSortGroup;
    const bool bad = (m > 0 && members[m].upos < members[m - 1].upos) || members[m].upos < members[first].upos ||
                     members[m].upos - members[first].upos > 0xfffe0000ull;
    if (bad) { *veto = 1; const uint32_t prev = atomicMin(&status[0], m); if (m < prev) status[1] = (uint32_t)INF_OUT_OVERFLOW; }
}

// early tail: this wave's bytes (and its members' verdicts) are out -- count it in its part of the launch (kernels.h InflateGate)
__device__ __forceinline__ void gate_wave_done(const InflateGate &gate, uint32_t lane) {
    if (!gate.done) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) {
        const uint32_t b = blockIdx.x;
        uint32_t part = 0;
#pragma unroll
        for (uint32_t k = 0; k + 1 < kGateParts; ++k) part += b >= gate.part_start[k] ? 1u : 0u;
        __hip_atomic_fetch_add(gate.done + part, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_wait_done(const uint32_t *done, uint32_t expected, uint32_t *timed_out) {
    uint32_t spins = 0;
    while (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expected) {
        if (++spins > 120000u) { *timed_out = 1; break; }
        for (int k = 0; k < 5; ++k) __builtin_amdgcn_s_sleep(127);
    }
}
void launch_wait_done(const uint32_t *done, uint32_t expected, uint32_t *timed_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_wait_done, dim3(1), dim3(1), 0, stream, done, expected, timed_out);
}

// one lane per member like k_inflate; no lane leaves before the wave is done (the lanes without a member serve the others' copies)
template <bool PROBE, bool PIECE = false, int WIN = 0 /* bit reader: 0 = 8 bytes a refill, 1 = 16-byte window */>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_inflate_coop(const uint8_t *__restrict__ comp, const Member *__restrict__ members,
                                                uint32_t n_members, uint8_t *arena, uint64_t upos_bias, uint32_t *len_scratch,
                                                uint32_t *status, uint32_t ignore_below, uint32_t index_bias, uint8_t *bad, uint32_t pairs,
                                                const uint32_t *__restrict__ perm, const uint32_t *veto, InflateGate gate) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // (stage entry point only: k_members_check found the list's layout unfit for 32-bit offsets -- nothing may be written)
    if (veto && *veto) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t slot = blockIdx.x * 64 + lane;
    const bool have = slot < n_members;
    // which member this lane decodes: the members of a group of kSortGroup come sorted by compressed length (k_member_sort), so that the 64
    // lanes of a wave need about the same number of trips (a wave runs as long as its longest lane: 93.5 -> 98.6 % of the lanes busy)
    const uint32_t m = have ? (perm ? perm[slot] : slot) : n_members - 1;
    const Member mb = members[m];
    if (PIECE && gate.flags) {
        // the upload chunk with the last byte any lane of this wave reads (payload + footer + the bit reader's look-ahead); one lane polls,
        // seldom (every waiting wave's poll is a request to the fabric the running waves share), an acquire then drops what this CU's L1 may
        // hold.  A chunk that never comes (the host gave up): after ~2 s the wave's members are reported as not inflated.
        uint64_t end_b = mb.cpos + mb.clen + 24;
        end_b = end_b > gate.lo ? end_b - gate.lo : 0;
        uint32_t need = (uint32_t)min((uint64_t)(gate.n_chunks - 1), end_b / gate.chunk_bytes);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) need = max(need, (uint32_t)__shfl_xor((int)need, d, 64));
        bool arrived = true;
        if (lane == 0) {
            uint32_t spins = 0;
            while (__hip_atomic_load(gate.flags + need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != gate.epoch) {
                if (++spins > 100000u) { arrived = false; break; }
                for (int k = 0; k < 5; ++k) __builtin_amdgcn_s_sleep(127);
            }
        }
        arrived = __builtin_amdgcn_ballot_w64(!arrived) == 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // (the members of the last chunks are what the call waits for -- their own chain behind the last arrival: their waves issue ahead of the
        //  waves of earlier chunks that share their SIMD)
        if (!arrived) {
            if (have) {
                const uint32_t mi = m + index_bias;
                if (bad) bad[mi] = 1;
                uint32_t *sl = mi >= ignore_below ? status : status + kStatusEarly;
                const uint32_t prev = atomicMin(&sl[0], mi);
                if (mi < prev) sl[1] = (uint32_t)INF_IN_OVERRUN;
            }
            gate_wave_done(gate, lane);
            return;
        }
    }
    LdsTab T{lds + lane, len_scratch + (have ? slot : 0), gridDim.x * 64};
    // the lowest output address the wave's lanes can have: the first member of the wave's group (upos grows with the member index; PROBE: slot m)
    const uint32_t first = perm ? (blockIdx.x * 64 / kSortGroup) * kSortGroup : blockIdx.x * 64;
    const uint64_t base_off = PROBE ? (uint64_t)first * kBgzfMaxBlock : members[first].upos - upos_bias;
    WaveCopy C;
    C.wave_base = arena + base_off; C.lane = lane;
    C.d0 = C.d1 = C.d2 = C.d3 = u32x4{0, 0, 0, 0};
    C.t0 = C.t1 = C.t2 = C.t3 = 0xffffffffu;
    uint32_t out_len = 0;
    if (PROBE) {
        const bool run = have && mb.isize != 0xffffffffu;         // ~0: BSIZE runs past the end of the file (or is < 26): the read fails upstream
        const int st = inflate_coop<BitReader>(comp + mb.cpos, mb.clen, arena + (uint64_t)(have ? m : first) * kBgzfMaxBlock, kBgzfMaxBlock, &out_len, T, C, run, pairs);
        if (have) status[m] = run && st == INF_OK ? out_len : 0xffffffffu;
        return;
    }
    // a member whose claimed size is no BGZF block size owns no bytes of the arena (k_member_compact): it must not write any, whatever
    // range the host asked for.  ~0 = the member runs past the end of the file (k_member_link).
    const bool run = have && mb.isize <= kBgzfMaxBlock;
    typedef typename std::conditional<WIN == 1, BitReaderWin, BitReader>::type BR;
    int st = inflate_coop<BR>(comp + mb.cpos, mb.clen, run ? arena + (mb.upos - upos_bias) : C.wave_base, run ? mb.isize : 0, &out_len, T, C, run, pairs);
    if (have) {
        if (!run) st = mb.isize == 0xffffffffu ? INF_IN_OVERRUN : INF_OUT_OVERFLOW;
        else if (st == INF_OK && out_len != mb.isize) st = INF_SIZE_MISMATCH;
        if (st != INF_OK) {
            const uint32_t mi = m + index_bias;
            if (bad) bad[mi] = 1;
            uint32_t *sl = mi >= ignore_below ? status : status + kStatusEarly;
            uint32_t prev = atomicMin(&sl[0], mi);
            if (mi < prev) sl[1] = (uint32_t)st;
        }
    }
    if (PIECE) gate_wave_done(gate, lane);
}

// (the code-length / cold-symbol scratch of every lane, + the lane assignment of k_inflate_coop behind it)
static size_t inflate_scratch_words(uint32_t n_members) { return (size_t)((n_members + 63) / 64) * 64 * kScratchWordsPerLane; }
size_t inflate_scratch_bytes(uint32_t n_members) { return (inflate_scratch_words(n_members) + ((size_t)n_members + 63) / 64 * 64 + kSortGroup) * 4; }

// ---- the small-input form: one member per wave, the whole member in LDS (inflate_wave.h) ------------------------------------------------
struct DevWave {
    uint32_t lane;
    template <class F> __device__ __forceinline__ void lanes(F f) const { f(lane); }
    __device__ __forceinline__ void sync() const {                       // one wave per workgroup: orders the lanes' LDS traffic (and the compiler)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};
static_assert(sizeof(WaveShared) <= 80 * 1024, "two wave-per-member workgroups per CU");

__global__ __launch_bounds__(64) void k_inflate_wave(const uint8_t *__restrict__ comp, const Member *__restrict__ members, uint32_t n_members, uint8_t *arena,
                                                     uint64_t upos_bias, uint32_t *status, uint32_t ignore_below, uint32_t index_bias, uint8_t *bad) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    WaveShared &S = *reinterpret_cast<WaveShared *>(lds);
    const uint32_t m = blockIdx.x;
    if (m >= n_members) return;
    const Member mb = members[m];
    DevWave W{threadIdx.x};
    int st;
    uint32_t out_len = 0;
    // (the same refusals as k_inflate: a member whose claimed size is no BGZF block size owns no bytes of the arena)
    if (mb.isize > kBgzfMaxBlock) st = mb.isize == 0xffffffffu ? INF_IN_OVERRUN : INF_OUT_OVERFLOW;
    else {
        st = inflate_wave(W, S, comp + mb.cpos, mb.clen, arena + (mb.upos - upos_bias), mb.isize, &out_len);
        if (st == INF_OK && out_len != mb.isize) st = INF_SIZE_MISMATCH;
    }
    if (st != INF_OK && threadIdx.x == 0) {
        const uint32_t mi = m + index_bias;
        if (bad) bad[mi] = 1;
        uint32_t *slot = mi >= ignore_below ? status : status + kStatusEarly;
        uint32_t prev = atomicMin(&slot[0], mi);
        if (mi < prev) slot[1] = (uint32_t)st;
    }
}

// Four forms of the decoder (rgx_k_inflate_form / REGTOOLS_AMD_INFLATE = lane | wave | ring | coop):
//   1 lane  k_inflate       (round 1: one member per lane, every lane copies its own matches, 12 waves per CU)
//   2 wave  k_inflate_wave  (round 2: one member per WAVE, member in LDS: small inputs -- 1.17 ms per 512 members against the lane forms'
//                            7.7-9.7 ms per launch whatever its size; wins up to ~4,000 members, tools/inflate_forms.py)
//   3 ring  k_inflate_ring  (round 2: lane form with a per-lane LDS window; 0.5x the HBM traffic, 4 waves per CU, 1.9x slower: DESIGN.md 5.1)
//   4 coop  k_inflate_coop  (round 3: lane form, long matches moved by the wave in aligned 256-byte pieces: inflate_coop.h)
// The pipeline (form 0) takes the wave form up to kWaveFormMaxMembers members and kDefaultLaneForm beyond.
constexpr uint32_t kWaveFormMaxMembers = 2048;
constexpr int kDefaultLaneForm = 4;
constexpr int kDefaultWindow = 1;     // k_inflate_coop's bit reader (its WIN parameter)
static int inflate_form_env() {
    static const int f = [] {
        const char *e = getenv("REGTOOLS_AMD_INFLATE");
        return !e ? 0 : !strcmp(e, "lane") ? 1 : !strcmp(e, "wave") ? 2 : !strcmp(e, "ring") ? 3 : !strcmp(e, "coop") ? 4 : 0;
    }();
    return f;
}
// (the attribute belongs to the current device's copy of the function: once per device, and the shard threads of rgx_extract_multi
// may get here together)
// A configuration step that failed (hipFuncSetAttribute on this device's copy of a kernel) must not pass silently: the launch behind it
// would fail or run with too little LDS, the arena would stay untouched and the stages behind it would read it.  The error is kept per host
// thread until the caller's next checked HIP call (api_internal.h HIP_TRY -> pending_launch_error) turns it into RGX_ERR_DEVICE.
static thread_local hipError_t tl_launch_error = hipSuccess;
hipError_t pending_launch_error() {
    hipError_t e = tl_launch_error; tl_launch_error = hipSuccess;
    const hipError_t last = hipGetLastError();                    // (a launch whose configuration was refused: sticky until read)
    return e != hipSuccess ? e : last;
}
static void inflate_attrs() {
    static std::mutex mu;
    static bool done_dev[64] = {};
    static hipError_t err_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    bool &done = done_dev[dev];
    if (done) { if (err_dev[dev] != hipSuccess) tl_launch_error = err_dev[dev]; return; }
    hipError_t first = hipSuccess;
    auto set = [&](const void *fn, uint32_t bytes) { const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (e != hipSuccess && first == hipSuccess) first = e; };
    set((const void *)k_inflate<false, false>, kInflateLdsBytes);
    set((const void *)k_inflate<false, true>, kInflateLdsBytes);
    set((const void *)k_inflate<true>, kInflateLdsBytes);
    set((const void *)k_inflate_coop<false, false, 0>, kInflateLdsBytes);
    set((const void *)k_inflate<false, false, 4>, kInflateLdsBytes);
    set((const void *)k_inflate<false, true, 4>, kInflateLdsBytes);
    set((const void *)k_inflate_coop<false, true, 0>, kInflateLdsBytes);
    set((const void *)k_inflate_coop<false, false, 1>, kInflateLdsBytes);
    set((const void *)k_inflate_coop<false, true, 1>, kInflateLdsBytes);
    set((const void *)k_inflate_coop<true>, kInflateLdsBytes);
    set((const void *)k_inflate_wave, (uint32_t)sizeof(WaveShared));
    set((const void *)k_inflate_ring<false>, kRingLdsBytes);
    set((const void *)k_inflate_ring<true>, kRingLdsBytes);
    err_dev[dev] = first; done = true;
    if (first != hipSuccess) tl_launch_error = first;
}
// The gate's flag word, written ON THE DEVICE (a one-lane kernel queued on the copy stream behind the chunk's copy): a word the copy engine
// wrote was seen by the polling waves only hundreds of microseconds later (their system-scope loads are served by an L2 that the engine's
// write does not reach; measured: 604 ms per step); a store from a wave is coherent with them.
__global__ void k_gate_set(uint32_t *flag, uint32_t epoch) { __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
void launch_gate_set(uint32_t *flag, uint32_t epoch, hipStream_t stream) { hipLaunchKernelGGL(k_gate_set, dim3(1), dim3(1), 0, stream, flag, epoch); }
bool inflate_takes_coop(uint32_t n_members) {
    const int f = inflate_form_env();
    return f ? f == 4 : n_members > kWaveFormMaxMembers && kDefaultLaneForm == 4;
}
void launch_inflate(const uint8_t *comp, const Member *members, uint32_t n_members, uint8_t *arena, uint64_t upos_bias, uint32_t *len_scratch,
                    uint32_t *status, hipStream_t stream, uint32_t ignore_below, uint32_t index_bias, bool piece, int form, uint8_t *bad, int plan, bool check_layout, InflateGate gate) {
    if (!n_members) return;
    inflate_attrs();
    // k_inflate_coop's mode word (inflate_coop.h): bit 0 = a literal and the symbol behind it per trip, bits 0 + 1 = exact bit counts and a
    // match behind a literal pair in the same trip, bit 2 = runs (distance <= 16) written from registers, 64 bytes a trip (round 4).
    // Same box, interleaved, ms (tools/lab/runs_ab.sh, profiles/r04_inflate_modes.txt): bench payload mode 1 15.2-15.3 / 3 14.2-15.1 / 7 14.1-15.0;
    // random bases + qualities 1: 48.2-48.4 / 3: 45.4-47.4; long reads 0: 121.2 / 1: 123.9-124.7 / 4: 79.8-79.9 / 5: 80.4-80.6 / 7: 80.5-80.9.
    const uint32_t two = (plan & 1) ? 7u : 4u;
    if (!form) form = inflate_form_env();
    if (!form) form = n_members <= kWaveFormMaxMembers ? 2 : (plan & 2) ? 1 : kDefaultLaneForm;
    const uint32_t blocks = (n_members + 63) / 64;
    switch (form) {
    case 2:
        hipLaunchKernelGGL(k_inflate_wave, dim3(n_members), dim3(64), (uint32_t)sizeof(WaveShared), stream, comp, members, n_members, arena, upos_bias, status, ignore_below, index_bias, bad);
        break;
    case 3:
        hipLaunchKernelGGL(k_inflate_ring<false>, dim3(blocks), dim3(64), kRingLdsBytes, stream, comp, members, n_members, arena, upos_bias, len_scratch, status, ignore_below, index_bias, bad);
        break;
    case 4: {
        // plan bit 0 set (payloads that compress up to 32 x): literal pairs and lanes sorted by compressed length; the windowed bit reader always
        // (round 4: its loads are no longer waited for on the spot -- long reads 117.5 ms with it against 120.7 without, kernels.h)
        const bool sort = (plan & 1) != 0;
        const int win = kDefaultWindow;
        uint32_t *perm = sort ? len_scratch + inflate_scratch_words(n_members) : nullptr;
        if (perm) hipLaunchKernelGGL(k_member_sort, dim3((n_members + kSortGroup - 1) / kSortGroup), dim3(256), 0, stream, members, n_members, perm);
        // (the veto word: behind the lane assignment in the scratch; only the stage entry point asks for the check)
        uint32_t *veto = check_layout ? len_scratch + inflate_scratch_words(n_members) + ((size_t)n_members + 63) / 64 * 64 + kSortGroup - 1 : nullptr;
        if (check_layout) {
            (void)hipMemsetAsync(veto, 0, 4, stream);
            hipLaunchKernelGGL(k_members_check, dim3((n_members + 255) / 256), dim3(256), 0, stream, members, n_members, status, veto);
        }
#define RGX_COOP(PIECE_, WIN_) hipLaunchKernelGGL((k_inflate_coop<false, PIECE_, WIN_>), dim3(blocks), dim3(64), kInflateLdsBytes, stream, comp, members, n_members, arena, upos_bias, len_scratch, status, ignore_below, index_bias, bad, two, perm, veto, gate)
        if (piece) { if (win) RGX_COOP(true, 1); else RGX_COOP(true, 0); }
        else { if (win) RGX_COOP(false, 1); else RGX_COOP(false, 0); }
#undef RGX_COOP
        break;
    }
    default: {
        // plan bit 1 = a payload of mostly literals (< 8 x): up to four of them per trip
        const bool lits4 = form == 5 || (plan & 2) != 0;          // (form 5: the stage entry point's way to ask for it)
#define RGX_LANE(PIECE_, LITS_) hipLaunchKernelGGL((k_inflate<false, PIECE_, LITS_>), dim3(blocks), dim3(64), kInflateLdsBytes, stream, comp, members, n_members, arena, upos_bias, len_scratch, status, ignore_below, index_bias, bad)
        if (piece) { if (lits4) RGX_LANE(true, 4); else RGX_LANE(true, 1); }
        else { if (lits4) RGX_LANE(false, 4); else RGX_LANE(false, 1); }
#undef RGX_LANE
    }
    }
}
void launch_inflate_probe(const uint8_t *comp, const Member *members, uint32_t n_members, uint8_t *slots, uint32_t *len_scratch, uint32_t *sizes,
                          hipStream_t stream) {
    if (!n_members) return;
    inflate_attrs();
    const int form = inflate_form_env() ? inflate_form_env() : kDefaultLaneForm;
    const uint32_t blocks = (n_members + 63) / 64;
    if (form == 3) hipLaunchKernelGGL(k_inflate_ring<true>, dim3(blocks), dim3(64), kRingLdsBytes, stream, comp, members, n_members, slots, (uint64_t)0, len_scratch, sizes, 0u, 0u, (uint8_t *)nullptr);
    else if (form == 4) hipLaunchKernelGGL(k_inflate_coop<true>, dim3(blocks), dim3(64), kInflateLdsBytes, stream, comp, members, n_members, slots, (uint64_t)0, len_scratch, sizes, 0u, 0u, (uint8_t *)nullptr, 1u, (const uint32_t *)nullptr, (const uint32_t *)nullptr, InflateGate());
    else hipLaunchKernelGGL(k_inflate<true>, dim3(blocks), dim3(64), kInflateLdsBytes, stream, comp, members, n_members, slots, (uint64_t)0, len_scratch, sizes, 0u, 0u, (uint8_t *)nullptr);
}

// =====================================================================================================
// a2. record framing: speculative per-segment chains, verified exactly
// =====================================================================================================
// Walk the block_size chain from `o` until it leaves [.., seg_end). Returns the first record start >= seg_end
// (or kChainEnd when the chain hits an unreadable record / the end of the stream) and counts the records started.
// cp (optional): cp[j] = offset of record 8j from seg_begin -- k_decode_seg's lanes re-walk eight records each from there instead of
// one lane re-walking all of them.
__device__ __forceinline__ uint64_t walk_chain(const uint8_t *arena, uint64_t o, uint64_t seg_end, uint64_t lim, uint32_t &cnt, uint16_t *cp = nullptr,
                                               uint64_t seg_begin = 0, bool lite = false) {
    cnt = 0;
    while (o < seg_end) {
        if (o + 36 > lim) return kChainEnd;             // bam_read1: short read of the fixed part
        RecHead h;
        if (lite) {                                     // block_size alone: ONE request per record (the rest of the test: the decode pass, SegGeom::lite_walk)
            h.block_len = (int32_t)ld32(arena + o);
            if (h.block_len < 32) return kChainEnd;
        } else {
            rec_head(arena + o, h);
            if (!rec_sane(h)) return kChainEnd;         // sam.c:421-423 -> iteration ends
        }
        uint64_t nxt = o + 4 + (uint64_t)(uint32_t)h.block_len;
        if (nxt > lim) return kChainEnd;                // truncated record body
        if (cp && (cnt & 7u) == 0) cp[cnt >> 3] = (uint16_t)(o - seg_begin);
        ++cnt;
        o = nxt;
    }
    return o;
}

// the bytes segment s covers: [a, b), where its chain's readable bytes end (lim), and whether s is the first segment of its chain
__device__ __forceinline__ void seg_of(const SegGeom &g, uint32_t s, uint64_t &a, uint64_t &b, uint64_t &lim, bool &first) {
    if (!g.chunks) {
        a = g.pos0 + (uint64_t)s * g.seg_bytes; b = a + g.seg_bytes; lim = g.lim; first = s == 0;
        if (b > lim) b = lim;
        return;
    }
    uint32_t lo = 0, hi = g.n_chunks;                    // the last chunk whose seg_base is <= s (few chunks, the table stays in the caches)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g.chunks[mid].seg_base <= s) lo = mid; else hi = mid; }
    const SegChunk c = g.chunks[lo];
    a = c.a + (uint64_t)(s - c.seg_base) * g.seg_bytes; b = a + g.seg_bytes; lim = c.dlim; first = s == c.seg_base;
    if (b > c.b) b = c.b;
}

__global__ void k_seg_walk(const uint8_t *__restrict__ arena, SegGeom g, uint32_t n_seg, int32_t n_ref,
                           uint64_t *seg_start, uint64_t *seg_exit, uint32_t *seg_cnt, uint16_t *seg_cp, uint32_t s_begin) {
    uint32_t s = s_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    uint16_t *cp = g.seg_bytes == kSegBytes ? seg_cp + (size_t)s * kSegCpSlots : nullptr;     // (checkpoints: 16-bit offsets, 16 KiB segments only)
    uint64_t a, b, lim; bool first;
    seg_of(g, s, a, b, lim, first);
    uint64_t o = a, ex = kChainEnd;
    uint32_t cnt = 0;
    if (!first) {
        // guess: first offset in the segment where three chained records all look like records AND the block_size chain from
        // there leaves the segment through readable records.  A false start almost never survives that (a 16 KiB walk is dozens
        // of records), so wrong exits -- the expensive kind of misprediction, see k_seg_verify -- are rare.
        // (the search loop and the walk are kept apart so that the lanes of a wave walk their segments together)
        o = kChainEnd;
        uint64_t c = a;
        for (;;) {
            // Coarse filter first, 16 candidate offsets per 20-byte load: the word at a record start is a block_size, i.e. in
            // [32, 2^27] (rec_sane / rec_plausible).  Sequence and quality bytes (long reads: kilobytes of them before the next
            // record) fail it without another memory access; only survivors get the full test below.
            uint64_t cand = kChainEnd;
            while (c < b) {
                const u32x4 v = ld128(arena + c);
                const uint32_t w[5] = {v[0], v[1], v[2], v[3], ld32(arena + c + 16)};
                uint32_t mask = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t bl = (i & 3) ? __builtin_amdgcn_alignbyte(w[(i >> 2) + 1], w[i >> 2], (uint32_t)(i & 3)) : w[i >> 2];
                    mask |= (bl - 32u <= (1u << 27) - 32u ? 1u : 0u) << i;
                }
                const uint64_t room = b - c;
                if (room < 16) mask &= (1u << (uint32_t)room) - 1u;
                bool found = false;
                while (mask) {
                    const uint32_t i = (uint32_t)__builtin_ctz(mask);
                    mask &= mask - 1;
                    const uint64_t q = c + i;
                    if (!rec_plausible(arena, q, lim, n_ref)) continue;
                    const uint64_t c2 = q + 4 + (uint64_t)ld32(arena + q);
                    if (c2 < lim) {
                        if (!rec_plausible(arena, c2, lim, n_ref)) continue;
                        const uint64_t c3 = c2 + 4 + (uint64_t)ld32(arena + c2);
                        if (c3 < lim && !rec_plausible(arena, c3, lim, n_ref)) continue;
                    }
                    cand = q; found = true; break;
                }
                if (found) break;
                c += 16;
            }
            if (cand == kChainEnd) break;
            ex = walk_chain(arena, cand, b, lim, cnt, cp, a, g.lite_walk != 0);
            if (ex != kChainEnd) { o = cand; break; }
            c = cand + 1;
        }
        if (o == kChainEnd) { ex = kChainUnknown; o = b; cnt = 0; }  // nothing usable: a placeholder that claims nothing (see k_seg_verify)
    } else ex = walk_chain(arena, o, b, lim, cnt, cp, a, g.lite_walk != 0);
    seg_start[s] = o; seg_exit[s] = ex; seg_cnt[s] = cnt;
}

// is segment s (> 0) framed the way its left neighbour's exit says it must be?
__device__ __forceinline__ bool seg_consistent(uint64_t b /* the segment's end */, uint64_t expect, uint64_t st, uint64_t ex, uint32_t cnt) {
    return expect >= b ? (st == b && ex == expect && cnt == 0) : st == expect;   // expect >= b (incl. kChainEnd): no record starts here
}

// One sweep.  Segment states: a GUESS (start/exit from a walk that left the segment through readable records), or a PLACEHOLDER
// (the search found nothing; exit = kChainUnknown: typical for the segments inside a record that is longer than a segment).
//  * a placeholder adopts its left neighbour's exit as soon as that exit is a real value -- it has no opinion to defend, and runs of
//    placeholders (long reads) resolve left to right, all runs of the file in parallel;
//  * a guess that disagrees with its left neighbour's exit is re-walked from that exit only once the neighbour itself agrees with
//    ITS neighbour: a wrong exit (a guess that landed on a false chain which never re-joins the true one) is then repaired where it
//    happened instead of being copied one segment to the right per sweep with the repair trailing it to the end of the file.
// Sweeps needed = longest run of consecutive placeholders / disagreeing guesses (+1 to see "no change").  status[0] = leftmost segment
// that is not final yet, so the loop ends only when the whole chain agrees -- which, segment 0 being exact, means it is exact.
// status[1] = leftmost segment whose chain ends (kChainEnd): when that one lies in the exact prefix the stream really ends there
// (truncated / unreadable record, sam.c:421-423) and the host empties everything to its right in one launch.
__global__ void k_seg_verify(const uint8_t *__restrict__ arena, SegGeom g, uint32_t n_seg,
                             const uint64_t *__restrict__ st_in, const uint64_t *__restrict__ ex_in, const uint32_t *__restrict__ cnt_in,
                             uint64_t *st_out, uint64_t *ex_out, uint32_t *cnt_out, uint32_t *status, uint16_t *seg_cp) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    uint64_t st = st_in[s], ex = ex_in[s];
    uint32_t cnt = cnt_in[s];
    uint64_t a, b, lim; bool first;
    seg_of(g, s, a, b, lim, first);
    if (ex == kChainEnd && (first || ex_in[s - 1] != kChainEnd)) atomicMin(status + 1, s);
    if (!first) {                                                               // (a chain's first segment starts at an exact offset)
        const uint64_t expect = ex_in[s - 1];
        if (expect == kChainUnknown) atomicMin(status, s);                      // the left neighbour knows nothing yet: wait
        else if (!seg_consistent(b, expect, st, ex, cnt)) {
            atomicMin(status, s);
            const bool placeholder = ex == kChainUnknown;
            uint64_t a1, b1, lim1; bool first1;
            seg_of(g, s - 1, a1, b1, lim1, first1);
            const bool left_settled = first1 || (ex_in[s - 2] != kChainUnknown && seg_consistent(b1, ex_in[s - 2], st_in[s - 1], expect, cnt_in[s - 1]));
            if (placeholder || left_settled) {
                if (expect >= b) { st = b; ex = expect; cnt = 0; }
                else { st = expect; ex = walk_chain(arena, st, b, lim, cnt, g.seg_bytes == kSegBytes ? seg_cp + (size_t)s * kSegCpSlots : nullptr, a, g.lite_walk != 0); }
            }
        }
    }
    st_out[s] = st; ex_out[s] = ex; cnt_out[s] = cnt;
}

// the record chain ended inside segment `last`: no record starts to its right
__global__ void k_seg_truncate(SegGeom g, uint32_t n_seg, uint32_t last, uint64_t *st, uint64_t *ex, uint32_t *cnt) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg || s <= last) return;
    uint64_t a, b, lim; bool first;
    seg_of(g, s, a, b, lim, first);
    st[s] = b; ex[s] = kChainEnd; cnt[s] = 0;
}

void launch_seg_walk(const uint8_t *arena, SegGeom g, uint32_t n_seg, int32_t n_ref, uint64_t *seg_start,
                     uint64_t *seg_exit, uint32_t *seg_cnt, uint16_t *seg_cp, hipStream_t stream, uint32_t s_begin) {
    if (n_seg <= s_begin) return;
    hipLaunchKernelGGL(k_seg_walk, dim3((n_seg - s_begin + 63) / 64), dim3(64), 0, stream, arena, g, n_seg, n_ref, seg_start, seg_exit, seg_cnt, seg_cp, s_begin);
}
void launch_seg_verify(const uint8_t *arena, SegGeom g, uint32_t n_seg, const uint64_t *seg_start_in,
                       const uint64_t *seg_exit_in, const uint32_t *seg_cnt_in, uint64_t *seg_start_out, uint64_t *seg_exit_out,
                       uint32_t *seg_cnt_out, uint32_t *status, uint16_t *seg_cp, hipStream_t stream) {
    if (!n_seg) return;
    hipLaunchKernelGGL(k_seg_verify, dim3((n_seg + 63) / 64), dim3(64), 0, stream, arena, g, n_seg, seg_start_in, seg_exit_in,
                       seg_cnt_in, seg_start_out, seg_exit_out, seg_cnt_out, status, seg_cp);
}
void launch_seg_truncate(SegGeom g, uint32_t n_seg, uint32_t last, uint64_t *seg_start, uint64_t *seg_exit, uint32_t *seg_cnt, hipStream_t stream) {
    if (!n_seg) return;
    hipLaunchKernelGGL(k_seg_truncate, dim3((n_seg + 255) / 256), dim3(256), 0, stream, g, n_seg, last, seg_start, seg_exit, seg_cnt);
}

// =====================================================================================================
// a2/a3/a5/a6. decode to SoA (one lane per record) + per-read junction-event count
// =====================================================================================================
// ONE WAVE PER SEGMENT.  The segment's bytes are loaded once, coalesced (16 B per lane per
// instruction), into LDS; lane 0 re-walks the (already verified) record chain there at LDS latency and leaves the
// record offsets in LDS; then the 64 lanes decode the records in parallel from LDS and write the SoA rows coalesced.
// Scattered 4-byte global reads of the record heads (one 64 B HBM sector per field group) become a streaming read.
constexpr uint32_t kSegTail = 1024;                      // bytes past the segment end kept in LDS (record bodies)
constexpr uint32_t kSegMaxRecs = kSegBytes / 36 + 2;     // a record is at least 36 bytes
// STAGED = false is the form for long records (kilobytes of sequence and quality per record, two or three records per segment):
// nothing is staged, heads / CIGAR / aux are read where they lie, and the 95 % of the stream that is sequence and quality never
// leaves HBM.  The host picks it when the mean record is longer than kSparseRecordBytes.
template <bool STAGED>
__global__ __launch_bounds__(64) void k_decode_seg(const uint8_t *__restrict__ arena, SegGeom g, uint32_t n_seg,
                                                   const uint64_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_base,
                                                   const uint32_t *__restrict__ seg_cnt, ExtractCfg cfg, ReadSoA soa, uint32_t *seg_iter,
                                                   uint32_t *seg_long, const uint16_t *__restrict__ seg_cp, uint32_t s_begin) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_buf[];          // kSegBytes + kSegTail + 48 when STAGED, nothing otherwise
    __shared__ uint32_t s_off[kSegMaxRecs];
    const uint32_t s = s_begin + blockIdx.x, lane = threadIdx.x;
    const uint32_t cnt = seg_cnt[s];
    if (cnt == 0) { if (lane == 0) { seg_iter[s] = 0; seg_long[s] = 0; } return; }
    uint64_t a, seg_b, seg_lim; bool seg_first;
    seg_of(g, s, a, seg_b, seg_lim, seg_first);
    // window [w0, w1): 16-byte aligned start at/below the segment begin, end = segment end + tail, clipped to the arena
    const uint64_t w0 = a & ~15ull;
    uint64_t w1 = a + kSegBytes + kSegTail;
    if (w1 > g.data_end) w1 = g.data_end;
    if constexpr (STAGED) {
        // all loads of the window in flight at once (18 x 1 KiB per wave), then the LDS stores: one HBM latency, not eighteen
        constexpr int kChunks = (kSegBytes + kSegTail + 16 + 1023) / 1024;
        u32x4 r[kChunks];
#pragma unroll
        for (int j = 0; j < kChunks; ++j) {
            const uint64_t o = w0 + (uint64_t)j * 1024 + (uint64_t)lane * 16;
            if (o < w1) r[j] = ld128(arena + o);               // reads < 16 B past lim: the arena is padded
        }
#pragma unroll
        for (int j = 0; j < kChunks; ++j) {
            const uint64_t o = w0 + (uint64_t)j * 1024 + (uint64_t)lane * 16;
            if (o < w1) *(u32x4 *)(s_buf + (o - w0)) = r[j];
        }
    }
    __syncthreads();
    // unaligned 32-bit read from the LDS window: two aligned dword reads + a byte funnel shift (v_alignbyte_b32)
    const uint32_t *s_w = (const uint32_t *)s_buf;
    auto lds32 = [&](uint32_t off) -> uint32_t {
        if constexpr (STAGED) return __builtin_amdgcn_alignbyte(s_w[(off >> 2) + 1], s_w[off >> 2], off & 3u);
        else return ld32(arena + w0 + off);
    };
    // chain walk inside LDS: lane j re-walks records 8j .. 8j+7 from the checkpoint the framing left (a segment holds at most
    // kSegMaxRecs = 457 records = 58 checkpoints)
    if (lane * 8 < cnt) {
        uint64_t o = a + seg_cp[(size_t)s * kSegCpSlots + lane];
        const uint32_t k_end = min(cnt, lane * 8 + 8);
        for (uint32_t k = lane * 8; k < k_end; ++k) {
            const uint32_t ro = (uint32_t)(o - w0);
            s_off[k] = ro;
            o += 4 + (uint64_t)((STAGED && o + 4 <= w1) ? lds32(ro) : ld32(arena + o));
        }
    }
    __syncthreads();
    const uint32_t base = seg_base[s];
    uint32_t n_iter = 0, n_long = 0;
    uint32_t first_stop = 0xffffffffu, last_in = 0;
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint32_t ro = s_off[k];
        const uint64_t o = w0 + ro;
        // the 36-byte record head always lies inside the window (tail >= 36); CIGAR / aux of the spliced minority are read from the arena
        RecHead h;
        {
            const uint32_t sh = ro & 3u, wi = ro >> 2;
            uint32_t x[8];
            if constexpr (STAGED) {
                uint32_t d[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) d[q] = s_w[wi + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
            } else {
                (void)sh; (void)wi;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = ld32(arena + o + 4 * q);
            }
            h.block_len = (int32_t)x[0]; h.tid = (int32_t)x[1]; h.pos = (int32_t)x[2];
            h.l_qname = x[3] & 0xff; h.n_cigar = x[4] & 0xffff; h.flag = x[4] >> 16;
            h.l_qseq = (int32_t)x[5]; h.mtid = (int32_t)x[6];
            h.aux_off = (int64_t)h.l_qname + 4 * (int64_t)h.n_cigar + (((int64_t)h.l_qseq + 1) >> 1) + h.l_qseq;
        }
        // (the framing only looked at block_size: the call starts over with the full walk.  Until then the record is inert: its CIGAR count and its
        //  aux offset are not to be believed -- a negative l_seq puts the aux walk in front of the arena, 65,535 operations run past its end)
        if (cfg.insane_out && !rec_sane(h)) { cfg.insane_out[0] = 1; h.n_cigar = 0; }
        // body (CIGAR, aux) from the LDS window when the whole record is inside it, else straight from the arena
        const uint64_t rec_end = o + 4 + (uint64_t)(uint32_t)h.block_len;
        const bool in_win = STAGED && rec_end <= w1;
        const uint32_t cig_ro = ro + 36 + h.l_qname;
        const uint8_t *g_data = arena + o + 36;
        auto cigar_at = [&](uint32_t q) -> uint32_t { return in_win ? lds32(cig_ro + 4 * q) : ld32(g_data + h.l_qname + 4 * (size_t)q); };
        const uint32_t i = base + k;
        soa.tid[i] = h.tid; soa.pos[i] = h.pos;
        soa.flag_nc[i] = h.flag << 16 | h.n_cigar;
        soa.cig_off[i] = o + 36 + h.l_qname;
        if (soa.rec_off) soa.rec_off[i] = o;
        bool in_region = true;
        if (cfg.region_tid != -2) {
            in_region = h.tid == cfg.region_tid && h.pos < cfg.region_end;
            if (cfg.stop_out) {                                // hts_itr_next's end rule (hts.c:1946-1950), see ExtractCfg
                if (i >= cfg.stop_index) in_region = false;
                else if (!in_region) first_stop = min(first_stop, i);
            }
            if (in_region) {                                   // bam_endpos (sam.c:336-342)
                int32_t endpos = h.pos + 1;
                if (!(h.flag & 4) && h.n_cigar > 0) {
                    int32_t l = 0;
                    for (uint32_t q = 0; q < h.n_cigar; ++q) l += (int32_t)cig_ref_len(cigar_at(q));
                    endpos = h.pos + l;
                }
                in_region = endpos > cfg.region_beg;
            }
        }
        n_iter += in_region ? 1u : 0u;
        if (in_region) last_in = i + 1;
        uint32_t nev = 0;
        char strand = '?';
        uint8_t odd = 0;                                     // bit 7 of the row's strand byte: k_collect_odd_aux (identify)
        if (in_region && h.n_cigar > 1 && h.tid >= 0 && h.tid < cfg.n_ref) {
            if (cfg.bc0 && cfg.abort_out) {                   // -b: the barcode tag is asked for first (the walk is strand_from_tag's; what it finds is the barcode kernels' business)
                const int64_t l_data = (int64_t)h.block_len - 32;
                bool unknown = false;
                if (in_win) { const uint8_t *body = s_buf + ro + 36; (void)strand_from_tag(body + h.aux_off, body + l_data, cfg.bc0, cfg.bc1, &unknown); }
                else (void)strand_from_tag(g_data + h.aux_off, g_data + l_data, cfg.bc0, cfg.bc1, &unknown);
                if (unknown) atomicMin(cfg.abort_out, i);
            }
            bool has_n = false;
            for (uint32_t q = 0; q < h.n_cigar; ++q) {
                const uint32_t c = cigar_at(q);
                if (cig_is_N(c)) { const uint32_t len = c >> 4; nev += !(len < cfg.min_intron || len > cfg.max_intron); has_n = true; }
            }
            // (the tag is asked for by every junction the walk reaches, also one junction_qc then drops: any N operation, cc:415 / :447 / :467 / :490)
            if (nev || (has_n && (cfg.abort_out || cfg.odd_count) && cfg.strandness == 0)) {
                if (cfg.strandness == 0) {
                    const int64_t l_data = (int64_t)h.block_len - 32;
                    bool unknown = false;
                    // two call sites on purpose: the LDS one compiles to ds_read_u8, the other to global loads (no flat/generic pointer)
                    if (in_win) { const uint8_t *body = s_buf + ro + 36; strand = strand_from_tag(body + h.aux_off, body + l_data, cfg.tag0, cfg.tag1, &unknown); }
                    else strand = strand_from_tag(g_data + h.aux_off, g_data + l_data, cfg.tag0, cfg.tag1, &unknown);
                    if (unknown && cfg.abort_out) atomicMin(cfg.abort_out, i);
                    if (unknown && cfg.odd_count) { atomicAdd(cfg.odd_count, 1u); odd = 0x80; }
                } else strand = strand_from_flag(h.flag, cfg.strandness);
                if (nev) n_long += h.n_cigar > cfg.long_threshold ? 1u : 0u;
            }
        }
        soa.strand[i] = (uint8_t)strand | odd;
        soa.n_ev[i] = nev;
    }
    // per-segment totals instead of global atomics: one hot address serialises at ~90 atomics/us
    n_iter = wave_incl_scan(n_iter); n_long = wave_incl_scan(n_long);
    if (lane == 63) { seg_iter[s] = n_iter; seg_long[s] = n_long; }
    if (cfg.stop_out) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { first_stop = min(first_stop, (uint32_t)__shfl_xor((int)first_stop, d, 64)); last_in = max(last_in, (uint32_t)__shfl_xor((int)last_in, d, 64)); }
        // (segments are launched roughly in order: after the first few, the values in memory already beat most candidates)
        if (lane == 0) {
            if (first_stop < *(volatile uint32_t *)&cfg.stop_out[0]) atomicMin(&cfg.stop_out[0], first_stop);
            if (last_in > *(volatile uint32_t *)&cfg.stop_out[1]) atomicMax(&cfg.stop_out[1], last_in);
        }
    }
}

// Long records (k_decode_seg<false>'s workload: two or three records per 16 KiB segment), one LANE per segment.  A wave per segment is a workgroup
// per two records there -- 4 M workgroups of which one lane works, each a chain of dependent gathers (checkpoint, the records' lengths, head, CIGAR, aux):
// 10.6 ms for config 5's 10 M records.  Here a lane follows its own segment's records from the verified start; sixty-four chains per wave.
// Same rows, same counts: the per-record part restates k_decode_seg's with every read from the arena.
__global__ __launch_bounds__(64) void k_decode_sparse(const uint8_t *__restrict__ arena, uint32_t n_seg, const uint64_t *__restrict__ seg_start,
                                                      const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_cnt, ExtractCfg cfg, ReadSoA soa,
                                                      uint32_t *seg_iter, uint32_t *seg_long, uint32_t s_begin) {
    const uint32_t s = s_begin + blockIdx.x * 64 + threadIdx.x;
    uint32_t first_stop = 0xffffffffu, last_in = 0;
    if (s < n_seg) {
        const uint32_t cnt = seg_cnt[s];
        uint32_t n_iter = 0, n_long = 0;
        if (cnt) {
            uint64_t o = seg_start[s];
            const uint32_t base = seg_base[s];
            for (uint32_t k = 0; k < cnt; ++k) {
                RecHead h;
                rec_head(arena + o, h);
                if (cfg.insane_out && !rec_sane(h)) { cfg.insane_out[0] = 1; h.n_cigar = 0; }       // (inert until the full walk, as in k_decode_seg)
                const uint8_t *g_data = arena + o + 36;
                auto cigar_at = [&](uint32_t q) -> uint32_t { return ld32(g_data + h.l_qname + 4 * (size_t)q); };
                const uint32_t i = base + k;
                soa.tid[i] = h.tid; soa.pos[i] = h.pos;
                soa.flag_nc[i] = h.flag << 16 | h.n_cigar;
                soa.cig_off[i] = o + 36 + h.l_qname;
                if (soa.rec_off) soa.rec_off[i] = o;
                bool in_region = true;
                if (cfg.region_tid != -2) {
                    in_region = h.tid == cfg.region_tid && h.pos < cfg.region_end;
                    if (cfg.stop_out) {                                // hts_itr_next's end rule (hts.c:1946-1950), see ExtractCfg
                        if (i >= cfg.stop_index) in_region = false;
                        else if (!in_region) first_stop = min(first_stop, i);
                    }
                    if (in_region) {                                   // bam_endpos (sam.c:336-342)
                        int32_t endpos = h.pos + 1;
                        if (!(h.flag & 4) && h.n_cigar > 0) {
                            int32_t l = 0;
                            for (uint32_t q = 0; q < h.n_cigar; ++q) l += (int32_t)cig_ref_len(cigar_at(q));
                            endpos = h.pos + l;
                        }
                        in_region = endpos > cfg.region_beg;
                    }
                }
                n_iter += in_region ? 1u : 0u;
                if (in_region) last_in = i + 1;
                uint32_t nev = 0;
                char strand = '?';
                uint8_t odd = 0;
                if (in_region && h.n_cigar > 1 && h.tid >= 0 && h.tid < cfg.n_ref) {
                    if (cfg.bc0 && cfg.abort_out) {
                        bool unknown = false;
                        (void)strand_from_tag(g_data + h.aux_off, g_data + ((int64_t)h.block_len - 32), cfg.bc0, cfg.bc1, &unknown);
                        if (unknown) atomicMin(cfg.abort_out, i);
                    }
                    bool has_n = false;
                    for (uint32_t q = 0; q < h.n_cigar; ++q) {
                        const uint32_t c = cigar_at(q);
                        if (cig_is_N(c)) { const uint32_t len = c >> 4; nev += !(len < cfg.min_intron || len > cfg.max_intron); has_n = true; }
                    }
                    if (nev || (has_n && (cfg.abort_out || cfg.odd_count) && cfg.strandness == 0)) {
                        if (cfg.strandness == 0) {
                            const int64_t l_data = (int64_t)h.block_len - 32;
                            bool unknown = false;
                            strand = strand_from_tag(g_data + h.aux_off, g_data + l_data, cfg.tag0, cfg.tag1, &unknown);
                            if (unknown && cfg.abort_out) atomicMin(cfg.abort_out, i);
                            if (unknown && cfg.odd_count) { atomicAdd(cfg.odd_count, 1u); odd = 0x80; }
                        } else strand = strand_from_flag(h.flag, cfg.strandness);
                        if (nev) n_long += h.n_cigar > cfg.long_threshold ? 1u : 0u;
                    }
                }
                soa.strand[i] = (uint8_t)strand | odd;
                soa.n_ev[i] = nev;
                o += 4 + (uint64_t)(uint32_t)h.block_len;
            }
        }
        seg_iter[s] = n_iter; seg_long[s] = n_long;
    }
    if (cfg.stop_out) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { first_stop = min(first_stop, (uint32_t)__shfl_xor((int)first_stop, d, 64)); last_in = max(last_in, (uint32_t)__shfl_xor((int)last_in, d, 64)); }
        if (threadIdx.x == 0) {
            if (first_stop < *(volatile uint32_t *)&cfg.stop_out[0]) atomicMin(&cfg.stop_out[0], first_stop);
            if (last_in > *(volatile uint32_t *)&cfg.stop_out[1]) atomicMax(&cfg.stop_out[1], last_in);
        }
    }
}

// list of the reads that go to the wave-per-read kernel, built without atomics: one wave per segment, positions from the
// exclusive scan of the per-segment counts
__global__ __launch_bounds__(64) void k_long_fill(uint32_t n_seg, const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_cnt,
                                                  const uint32_t *__restrict__ seg_long_base, ExtractCfg cfg, ReadSoA soa, uint32_t *long_list) {
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    const uint32_t cnt = seg_cnt[s], base = seg_base[s];
    uint32_t out = seg_long_base[s];
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
        const uint32_t k = k0 + lane;
        bool is_long = false;
        if (k < cnt) is_long = soa.n_ev[base + k] != 0 && (soa.flag_nc[base + k] & 0xffff) > cfg.long_threshold;
        const uint64_t m = __ballot(is_long);
        if (is_long) long_list[out + (uint32_t)__popcll(m & lanemask_lt())] = base + k;
        out += (uint32_t)__popcll(m);
    }
}

const DecodeKnobs &decode_knobs() {
    static const DecodeKnobs k = [] {
        DecodeKnobs v;
        if (const char *e = getenv("REGTOOLS_AMD_DECODE")) {
            v.wave_form = !strncmp(e, "wave", 4);
            if (const char *c = strchr(e, ',')) v.seg_bytes = atoi(c + 1);
        }
        return v;
    }();
    return k;
}
void launch_decode_seg(const uint8_t *arena, SegGeom g, uint32_t n_seg, const uint64_t *seg_start, const uint32_t *seg_base,
                       const uint32_t *seg_cnt, ExtractCfg cfg, ReadSoA soa, uint32_t *seg_iter, uint32_t *seg_long, const uint16_t *seg_cp, bool staged, hipStream_t stream,
                       uint32_t s_begin) {
    if (n_seg <= s_begin) return;
    const uint32_t n = n_seg - s_begin;
    if (staged && g.seg_bytes == kSegBytes) hipLaunchKernelGGL(k_decode_seg<true>, dim3(n), dim3(64), kSegBytes + kSegTail + 48, stream, arena, g, n_seg, seg_start, seg_base, seg_cnt, cfg, soa, seg_iter, seg_long, seg_cp, s_begin);
    else {
        static const bool wave_form = decode_knobs().wave_form;     // (tests: the workgroup-per-segment form it replaced)
        if (wave_form && g.seg_bytes == kSegBytes) hipLaunchKernelGGL(k_decode_seg<false>, dim3(n), dim3(64), 0, stream, arena, g, n_seg, seg_start, seg_base, seg_cnt, cfg, soa, seg_iter, seg_long, seg_cp, s_begin);
        else hipLaunchKernelGGL(k_decode_sparse, dim3((n + 63) / 64), dim3(64), 0, stream, arena, n_seg, seg_start, seg_base, seg_cnt, cfg, soa, seg_iter, seg_long, s_begin);
    }
}
// identify -s XS: the reads whose strand tag lies behind an aux field of unknown type (marked by the decode kernels), with the span the iterator tests
__global__ void k_collect_odd_aux(const uint8_t *__restrict__ arena, ReadSoA soa, uint32_t n_rec, uint32_t cap, uint32_t *count, int32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec || !(soa.strand[i] & 0x80u)) return;
    const uint32_t fnc = soa.flag_nc[i];
    const int32_t pos = soa.pos[i];
    const uint32_t k = atomicAdd(count, 1u);
    if (k < cap) { out[3 * (size_t)k] = soa.tid[i]; out[3 * (size_t)k + 1] = pos; out[3 * (size_t)k + 2] = rec_endpos(arena + soa.cig_off[i], fnc & 0xffffu, fnc >> 16, pos); }
}
void launch_collect_odd_aux(const uint8_t *arena, ReadSoA soa, uint32_t n_rec, uint32_t cap, uint32_t *count, int32_t *out, hipStream_t stream) {
    if (n_rec) hipLaunchKernelGGL(k_collect_odd_aux, dim3((n_rec + 255) / 256), dim3(256), 0, stream, arena, soa, n_rec, cap, count, out);
}
void launch_long_fill(uint32_t n_seg, const uint32_t *seg_base, const uint32_t *seg_cnt, const uint32_t *seg_long_base, ExtractCfg cfg, ReadSoA soa,
                      uint32_t *long_list, hipStream_t stream) {
    if (n_seg) hipLaunchKernelGGL(k_long_fill, dim3(n_seg), dim3(64), 0, stream, n_seg, seg_base, seg_cnt, seg_long_base, cfg, soa, long_list);
}


// =====================================================================================================
// a4. CIGAR scan + junction emit
// =====================================================================================================
__device__ __forceinline__ void put_event(const EventSoA &ev, uint32_t slot, int32_t tid, uint32_t start, uint32_t end, uint32_t ts,
                                          uint32_t te, char strand, uint32_t rpos, uint32_t rend, uint32_t read) {
    if (ev.read) ev.read[slot] = read;
    ev.tid[slot] = (uint32_t)tid; ev.start[slot] = start;
    ev.ilen_cls[slot] = (end - start) << 2 | strand_class(strand);
    ev.ts[slot] = ts; ev.te[slot] = te; ev.strand[slot] = (uint8_t)strand;
    if (ev.rpos) { ev.rpos[slot] = rpos; ev.rend[slot] = rend; }
}

// short reads: one lane per read, the serial state machine (config 2: 1 or 3 ops)
// (row_begin, part_totals, n_parts: the early tail emits the rows of one part of the file at a time; ev_base then counts from the part's first row, and the
//  events of the parts in front of it -- their totals sit in device memory, nobody waited for them -- are added here)
__global__ void k_emit_short(const uint8_t *__restrict__ arena, uint32_t n_rec, ExtractCfg cfg, ReadSoA soa,
                             const uint32_t *__restrict__ ev_base, EventSoA ev, uint32_t row_begin, const uint32_t *__restrict__ part_totals, uint32_t n_parts, uint32_t slot_cap) {
    uint32_t i = row_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    uint32_t nev = soa.n_ev[i];
    if (!nev) return;
    uint32_t fnc = soa.flag_nc[i], n_cigar = fnc & 0xffff;
    if (n_cigar > cfg.long_threshold) return;
    const uint8_t *cig = arena + soa.cig_off[i];
    int32_t tid = soa.tid[i];
    char strand = (char)soa.strand[i];
    uint32_t slot = ev_base[i];
    for (uint32_t k = 0; k < n_parts; ++k) slot += part_totals[k];
    const int32_t pos = soa.pos[i];
    const uint32_t rend = ev.rpos ? (uint32_t)rec_endpos(cig, n_cigar, fnc >> 16, pos) : 0u;     // bam_endpos of the supporting read
    char carried = 0;
    cigar_walk(pos, cig, n_cigar, [&](uint32_t s, uint32_t e, uint32_t ts, uint32_t te) {
        char st = strand;
        if (cfg.fa_data) {
            // set_junction_strand (junctions_extractor.cc:345-359): motif first, the tag/flag rule only when the motif says '?'.
            // The strand is decided (and carried to the read's next junction) BEFORE junction_qc.
            const FaContig fc = cfg.fa_tab[tid];
            if (!fc.present) { cfg.fa_missing[0] = 1u + (uint32_t)tid; }
            else { const char m = strand_from_motif(cfg.fa_data, fc, s, e, carried); if (m != '?') st = m; }
            carried = st;
        }
        if (intron_ok(s, e, cfg.min_intron, cfg.max_intron)) { if (slot < slot_cap) put_event(ev, slot, tid, s, e, ts, te, st, (uint32_t)pos, rend, i); ++slot; }   // (slot_cap: a block sized before the count was known)
    });
}

// long reads: one WAVE per read. Ops are staged through LDS in tiles of 64, classified with ballots and
// positioned with a wave prefix sum of reference advance (parallel form):
//   for the N op at index i:  start = R[i], end = R[i+1],
//                             thick_start = R[pb+1]  (pb = last breaker {N,D,X,I,S} before i, or read start),
//                             thick_end   = R[nb]    (nb = first breaker after i, or read end).
__global__ __launch_bounds__(256) void k_emit_long(const uint8_t *__restrict__ arena, const uint32_t *__restrict__ long_list,
                                                   uint32_t n_long, ExtractCfg cfg, ReadSoA soa,
                                                   const uint32_t *__restrict__ ev_base, EventSoA ev) {
    __shared__ uint32_t s_ops[4][64];
    __shared__ uint32_t s_R[4][65];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    // (round 4: a read is three dependent round trips -- its index, its row, its CIGAR -- in front of ~200 instructions: the NEXT read's index and row are
    //  requested while this one's CIGAR is in flight and its junctions are worked out, so that an iteration waits for one round trip, not three)
    const uint32_t stride = gridDim.x * 4;
    uint32_t w = blockIdx.x * 4 + wave;
    if (w >= n_long) return;
    uint32_t nx_i = long_list[w];
    uint32_t nn_i = w + stride < n_long ? long_list[w + stride] : 0u;          // (the index of the read after the next: its row's loads must not wait for it)
    uint32_t nx_fnc = soa.flag_nc[nx_i], nx_slot = ev_base[nx_i];
    uint64_t nx_cig = soa.cig_off[nx_i];
    int32_t nx_tid = soa.tid[nx_i], nx_pos = soa.pos[nx_i];
    uint8_t nx_strand = soa.strand[nx_i];
    for (; w < n_long; w += stride) {
        const uint32_t i = nx_i, fnc = nx_fnc;
        const uint32_t n_cigar = fnc & 0xffff;
        const uint8_t *cig = arena + nx_cig;
        const int32_t tid = nx_tid;
        const char strand = (char)nx_strand;
        uint32_t slot = nx_slot;
        const uint32_t rpos = (uint32_t)nx_pos;
        // this read's first CIGAR tile: requested now, looked at below
        const uint32_t c_first = lane < n_cigar ? ld32(cig + 4 * (size_t)lane) : 0x5u /* 0H: inert */;
        // ... the next read's row behind it (its index came with the last iteration), and the index of the one after
        if (w + stride < n_long) {
            nx_i = nn_i;
            nx_fnc = soa.flag_nc[nx_i]; nx_slot = ev_base[nx_i]; nx_cig = soa.cig_off[nx_i]; nx_tid = soa.tid[nx_i]; nx_pos = soa.pos[nx_i]; nx_strand = soa.strand[nx_i];
            if (w + 2 * stride < n_long) nn_i = long_list[w + 2 * stride];
        }
        uint32_t rend = 0;
        if (ev.rpos) {                               // bam_endpos (sam.c:336-342): pos + reference span, or pos+1 for unmapped-flagged reads
            uint32_t span = 0;
            for (uint32_t t0 = lane; t0 < n_cigar; t0 += 64) span += cig_ref_len(ld32(cig + 4 * (size_t)t0));
            span = __shfl(wave_incl_scan(span), 63, 64);
            rend = ((fnc >> 16) & 4u) ? rpos + 1 : rpos + span;
        }
        uint32_t refpos = rpos;                      // R at the start of the tile
        uint32_t ts_carry = refpos;                  // R[pb+1] for an N with no breaker earlier in the tile
        // junction opened in an earlier tile and still waiting for its right anchor's end
        bool pend = false; uint32_t p_start = 0, p_end = 0, p_ts = 0; char p_strand = strand;
        char carried = 0;                            // intron-motif rule: strand of the read's previous junction
        for (uint32_t t0 = 0; t0 < n_cigar; t0 += 64) {
            const uint32_t k = t0 + lane;
            const uint32_t c = t0 == 0 ? c_first : k < n_cigar ? ld32(cig + 4 * (size_t)k) : 0x5u /* 0H: inert */;
            s_ops[wave][lane] = c;
            const uint32_t adv = cig_advances_junction_state(c) ? (c >> 4) : 0u;
            const uint32_t incl = wave_incl_scan(adv);
            const uint32_t R_before = refpos + incl - adv, R_after = refpos + incl;
            s_R[wave][lane] = R_before;
            if (lane == 63) s_R[wave][64] = R_after;
            const bool brk = k < n_cigar && cig_is_breaker(c);
            const bool isN = k < n_cigar && cig_is_N(c);
            const uint64_t B = __ballot(brk);
            __builtin_amdgcn_wave_barrier();
            // close the pending junction at this tile's first breaker
            if (pend && B) {
                const uint32_t nb = (uint32_t)__ffsll((unsigned long long)B) - 1;
                const uint32_t te = s_R[wave][nb];
                if (lane == 0 && intron_ok(p_start, p_end, cfg.min_intron, cfg.max_intron)) put_event(ev, slot, tid, p_start, p_end, p_ts, te, p_strand, rpos, rend, i);
                if (intron_ok(p_start, p_end, cfg.min_intron, cfg.max_intron)) ++slot;
                pend = false;
            }
            // per-lane junction geometry
            const uint64_t below = B & lanemask_lt();
            const uint64_t above = lane == 63 ? 0ull : (B >> (lane + 1)) << (lane + 1);
            const uint32_t ts = below ? s_R[wave][(63 - (uint32_t)__clzll((long long)below)) + 1] : ts_carry;
            const bool closed = above != 0;
            const uint32_t te = closed ? s_R[wave][(uint32_t)__ffsll((unsigned long long)above) - 1] : 0u;
            const bool ok = isN && intron_ok(R_before, R_after, cfg.min_intron, cfg.max_intron);
            // strand of this lane's junction: per-read tag/flag rule, or the intron-motif rule with its carried-over state --
            // a two-state chain over the N ops in CIGAR order, resolved lane by lane (a long read has a few dozen at most)
            char my_strand = strand;
            if (cfg.fa_data) {
                const FaContig fc = cfg.fa_tab[tid];
                char sA = strand, sB = strand;           // outcome if the carried strand is not '-' / is '-'
                if (isN) {
                    if (!fc.present) cfg.fa_missing[0] = 1u + (uint32_t)tid;
                    else {
                        const char mA = strand_from_motif(cfg.fa_data, fc, R_before, R_after, 0), mB = strand_from_motif(cfg.fa_data, fc, R_before, R_after, '-');
                        if (mA != '?') sA = mA;
                        if (mB != '?') sB = mB;
                    }
                }
                uint64_t nm = __ballot(isN);
                while (nm) {
                    const uint32_t l = (uint32_t)__ffsll((unsigned long long)nm) - 1; nm &= nm - 1;
                    const char a = (char)__shfl((int)sA, l, 64), b = (char)__shfl((int)sB, l, 64);
                    const char s = carried == '-' ? b : a;
                    if (lane == l) my_strand = s;
                    carried = s;
                }
            }
            // events keep CIGAR order: rank among the qc-passing N lanes that close inside this tile
            const uint64_t emit_mask = __ballot(ok && closed);
            if (ok && closed) put_event(ev, slot + (uint32_t)__popcll(emit_mask & lanemask_lt()), tid, R_before, R_after, ts, te, my_strand, rpos, rend, i);
            slot += (uint32_t)__popcll(emit_mask);
            // the last breaker of the tile: if it is an N it stays open into the next tile
            if (B) {
                const uint32_t lastb = 63 - (uint32_t)__clzll((long long)B);
                const uint32_t lc = s_ops[wave][lastb];
                ts_carry = s_R[wave][lastb + 1];
                if (cig_is_N(lc)) {
                    pend = true;
                    p_start = s_R[wave][lastb]; p_end = s_R[wave][lastb + 1];
                    const uint64_t bl = B & ((1ull << lastb) - 1ull);
                    p_ts = __shfl(ts, lastb, 64);
                    p_strand = (char)__shfl((int)my_strand, lastb, 64);
                    (void)bl;
                }
            }
            refpos = __shfl(R_after, 63, 64);
            __builtin_amdgcn_wave_barrier();
        }
        if (pend && lane == 0 && intron_ok(p_start, p_end, cfg.min_intron, cfg.max_intron)) put_event(ev, slot, tid, p_start, p_end, p_ts, refpos, p_strand, rpos, rend, i);
    }
}

void launch_emit_short(const uint8_t *arena, uint32_t n_rec, ExtractCfg cfg, ReadSoA soa, const uint32_t *ev_base, EventSoA ev,
                       hipStream_t stream, uint32_t row_begin, const uint32_t *part_totals, uint32_t n_parts, uint32_t slot_cap) {
    if (n_rec <= row_begin) return;
    hipLaunchKernelGGL(k_emit_short, dim3((n_rec - row_begin + 255) / 256), dim3(256), 0, stream, arena, n_rec, cfg, soa, ev_base, ev, row_begin, part_totals, n_parts, slot_cap);
}
void launch_emit_long(const uint8_t *arena, const uint32_t *long_list, uint32_t n_long, ExtractCfg cfg,
                      ReadSoA soa, const uint32_t *ev_base, EventSoA ev, hipStream_t stream) {
    if (!n_long) return;
    uint32_t blocks = (n_long + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;       // grid-stride: 8 workgroups per CU
    hipLaunchKernelGGL(k_emit_long, dim3(blocks), dim3(256), 0, stream, arena, long_list, n_long, cfg, soa, ev_base, ev);
}

// =====================================================================================================
// primitives: exclusive scan (3 kernels), stable 8-bit LSD radix pass on a permutation
// =====================================================================================================
constexpr uint32_t kScanTile = 4096;   // 256 threads x 16 items

__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t &block_total) {
    const uint32_t incl = wave_incl_scan(v);
    const uint32_t w = threadIdx.x >> 6;
    if (lane_id() == 63) s_wave[w] = incl;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) off += (k < w) ? s_wave[k] : 0u;
    block_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return off + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t *__restrict__ in, uint32_t n, uint32_t *tile_sum) {
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 16;
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) sum += (base + k < n) ? in[base + k] : 0u;
    uint32_t tot; block_excl_scan_256(sum, s_wave, tot);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_tiles(uint32_t *tile_sum, uint32_t n_tiles, uint32_t *total) {
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sum[i] : 0u;
        uint32_t tot; const uint32_t ex = block_excl_scan_256(v, s_wave, tot);
        if (i < n_tiles) tile_sum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *__restrict__ in, uint32_t *out, uint32_t n, const uint32_t *__restrict__ tile_sum) {
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 16;
    uint32_t v[16], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) { v[k] = (base + k < n) ? in[base + k] : 0u; sum += v[k]; }
    uint32_t tot; uint32_t run = block_excl_scan_256(sum, s_wave, tot) + tile_sum[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

size_t scan_tmp_words(uint32_t n) { return (size_t)(n + kScanTile - 1) / kScanTile + 1; }

void launch_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total, uint32_t *tmp, hipStream_t stream) {
    if (!n) { if (total) (void)hipMemsetAsync(total, 0, 4, stream); return; }
    const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_reduce, dim3(tiles), dim3(256), 0, stream, in, n, tmp);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(256), 0, stream, tmp, tiles, total);
    hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(256), 0, stream, in, out, n, tmp);
}

// ---- radix pass ------------------------------------------------------------------------------------------
// one wave per workgroup, 64 x rows keys per tile.  Round 4: 8 rows for sorts of up to two million keys -- the partial rows of the group-by, the unique rows
// of the output order: a few hundred thousand keys in 2,048-key tiles are 150 workgroups on 256 CUs, each a chain of 32 dependent rows (a pass 41 us); in
// 512-key tiles four times as many workgroups walk a quarter of the rows each
constexpr uint32_t kRadixRowsLarge = 32, kRadixRowsSmall = 8, kRadixSmallMax = 2u << 20;
static inline uint32_t radix_rows(uint32_t n) {
    return n <= kRadixSmallMax ? kRadixRowsSmall : kRadixRowsLarge;
}

__device__ __forceinline__ uint32_t radix_digit(const uint32_t *__restrict__ word, const uint32_t *__restrict__ perm_in, uint32_t i,
                                                uint32_t shift, uint32_t mask, uint32_t &src) {
    src = perm_in ? perm_in[i] : i;
    return (word[src] >> shift) & mask;
}

// lanes holding the same digit as me (8 ballots)
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
    uint64_t m = __ballot(valid);
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) {
        const uint64_t bal = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return m;
}

// KEYED = the key of position i is word[i] itself (it travels with the permutation: launch_radix_pass_keyed) instead of word[perm_in[i]]
template <bool KEYED>
__global__ __launch_bounds__(64) void k_radix_hist(const uint32_t *__restrict__ word, uint32_t shift, uint32_t mask,
                                                   const uint32_t *__restrict__ perm_in, uint32_t n, uint32_t n_tiles, uint32_t *hist, uint32_t rows) {
    __shared__ uint32_t s_cnt[256];
    for (uint32_t k = threadIdx.x; k < 256; k += 64) s_cnt[k] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * 64 * rows;
    for (uint32_t r = 0; r < rows; ++r) {
        const uint32_t i = base + r * 64 + threadIdx.x;
        if (i < n) { uint32_t src; atomicAdd(&s_cnt[KEYED ? ((word[i] >> shift) & mask) : radix_digit(word, perm_in, i, shift, mask, src)], 1u); }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 256; k += 64) hist[(size_t)k * n_tiles + blockIdx.x] = s_cnt[k];   // digit-major
}

template <bool KEYED>
__global__ __launch_bounds__(64) void k_radix_scatter(const uint32_t *__restrict__ word, uint32_t shift, uint32_t mask,
                                                      const uint32_t *__restrict__ perm_in, uint32_t *perm_out, uint32_t n,
                                                      uint32_t n_tiles, const uint32_t *__restrict__ hist_scan, uint32_t *key_out,
                                                      const uint32_t *__restrict__ digit_total, uint32_t rows) {
    __shared__ uint32_t s_base[256];
    {
        // where digit d starts in the output = the totals of the digits below it: 256 numbers, scanned by every workgroup for itself
        // (4 per lane + a wave scan) instead of one more launch; + this tile's rank inside the digit (k_radix_offsets)
        const uint32_t k0 = threadIdx.x * 4;
        const uint32_t t0 = digit_total[k0], t1 = digit_total[k0 + 1], t2 = digit_total[k0 + 2], t3 = digit_total[k0 + 3];
        const uint32_t incl = wave_incl_scan(t0 + t1 + t2 + t3);
        uint32_t run = incl - (t0 + t1 + t2 + t3);
        s_base[k0] = run + hist_scan[(size_t)k0 * n_tiles + blockIdx.x]; run += t0;
        s_base[k0 + 1] = run + hist_scan[(size_t)(k0 + 1) * n_tiles + blockIdx.x]; run += t1;
        s_base[k0 + 2] = run + hist_scan[(size_t)(k0 + 2) * n_tiles + blockIdx.x]; run += t2;
        s_base[k0 + 3] = run + hist_scan[(size_t)(k0 + 3) * n_tiles + blockIdx.x];
    }
    __syncthreads();
    const uint32_t base = blockIdx.x * 64 * rows;
    for (uint32_t r = 0; r < rows; ++r) {
        const uint32_t i = base + r * 64 + threadIdx.x;
        const bool valid = i < n;
        uint32_t src = 0, key = 0;
        uint32_t d = 0;
        if (KEYED) { if (valid) { key = word[i]; src = perm_in ? perm_in[i] : i; d = (key >> shift) & mask; } }
        else d = valid ? radix_digit(word, perm_in, i, shift, mask, src) : 0u;
        const uint64_t m = match_digit(d, valid);
        if (valid) {
            const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
            const uint32_t leader = (uint32_t)__ffsll((unsigned long long)m) - 1;
            uint32_t b = 0;
            if (lane_id() == leader) { b = s_base[d]; s_base[d] = b + (uint32_t)__popcll(m); }
            b = __shfl(b, leader, 64);
            perm_out[b + rank] = src;
            if (KEYED) key_out[b + rank] = key;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// one workgroup per digit: exclusive scan of that digit's per-tile counts (in place) + the digit's total
__global__ __launch_bounds__(256) void k_radix_offsets(uint32_t *hist, uint32_t n_tiles, uint32_t *digit_total) {
    __shared__ uint32_t s_wave[4];
    uint32_t *row = hist + (size_t)blockIdx.x * n_tiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += 256 * 8) {
        const uint32_t i0 = base + threadIdx.x * 8;
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = i0 + k < n_tiles ? row[i0 + k] : 0u; sum += v[k]; }
        uint32_t tot; uint32_t ex = carry + block_excl_scan_256(sum, s_wave, tot);
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (i0 + k < n_tiles) row[i0 + k] = ex; ex += v[k]; }
        carry += tot;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

// an upper bound that is MONOTONE in n: a buffer sized for n keys is reused for sorts of fewer keys (the unique rows of a merge), and up to
// kRadixSmallMax keys take the small tiles, which need four times the histogram words per key of the large ones
size_t radix_tmp_words(uint32_t n) {
    const size_t tile = 64 * (size_t)radix_rows(n), tiles = ((size_t)n + tile - 1) / tile;
    const size_t n_small = std::min<size_t>(n, kRadixSmallMax), tile_small = 64 * (size_t)radix_rows((uint32_t)n_small);
    const size_t tiles_small = (n_small + tile_small - 1) / tile_small;
    return 256 * std::max(tiles, tiles_small) + 256 + 4;
}

// a pass = three launches: per-tile digit counts, per-digit scan over the tiles, scatter
void launch_radix_pass(const uint32_t *word, uint32_t shift, uint32_t bits, const uint32_t *perm_in, uint32_t *perm_out, uint32_t n,
                       uint32_t *tmp, hipStream_t stream) {
    if (!n) return;
    const uint32_t rows = radix_rows(n), tiles = (n + 64 * rows - 1) / (64 * rows);
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t *hist = tmp, *totals = tmp + (size_t)256 * tiles;
    hipLaunchKernelGGL(k_radix_hist<false>, dim3(tiles), dim3(64), 0, stream, word, shift, mask, perm_in, n, tiles, hist, rows);
    hipLaunchKernelGGL(k_radix_offsets, dim3(256), dim3(256), 0, stream, hist, tiles, totals);
    hipLaunchKernelGGL(k_radix_scatter<false>, dim3(tiles), dim3(64), 0, stream, word, shift, mask, perm_in, perm_out, n, tiles, hist, (uint32_t *)nullptr, totals, rows);
}
void launch_radix_pass_keyed(const uint32_t *key_in, uint32_t *key_out, uint32_t shift, uint32_t bits, const uint32_t *perm_in, uint32_t *perm_out,
                             uint32_t n, uint32_t *tmp, hipStream_t stream) {
    if (!n) return;
    const uint32_t rows = radix_rows(n), tiles = (n + 64 * rows - 1) / (64 * rows);
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t *hist = tmp, *totals = tmp + (size_t)256 * tiles;
    hipLaunchKernelGGL(k_radix_hist<true>, dim3(tiles), dim3(64), 0, stream, key_in, shift, mask, perm_in, n, tiles, hist, rows);
    hipLaunchKernelGGL(k_radix_offsets, dim3(256), dim3(256), 0, stream, hist, tiles, totals);
    hipLaunchKernelGGL(k_radix_scatter<true>, dim3(tiles), dim3(64), 0, stream, key_in, shift, mask, perm_in, perm_out, n, tiles, hist, key_out, totals, rows);
}


// =====================================================================================================
// a1 (container). BGZF member discovery on the device
// =====================================================================================================
// check_header (bgzf.c:348-355): ID1 ID2 CM, FLG&4, XLEN == 6, 'B' 'C', SLEN == 2  (bytes 0,1,2,3,10..15)
__device__ __forceinline__ bool bgzf_magic_at(const uint8_t *__restrict__ p) {
    if (p[0] != 31 || p[1] != 139) return false;
    return p[2] == 8 && (p[3] & 4) && p[10] == 6 && p[11] == 0 && p[12] == 'B' && p[13] == 'C' && p[14] == 2 && p[15] == 0;
}

// one thread tests 16 consecutive offsets; a 256-thread workgroup covers one 4 KiB tile
__device__ __forceinline__ uint32_t magic_mask16(const uint8_t *__restrict__ bam, uint64_t len, uint64_t base) {
    uint32_t m = 0;
    if (base + 16 + 18 <= len) {
        // cheap prefilter on the two ID bytes using two aligned-agnostic 16-byte loads
        u32x4 a = ld128(bam + base), b = ld128(bam + base + 16);
        uint8_t w[32];
        *(u32x4 *)w = a; *(u32x4 *)(w + 16) = b;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (w[k] == 31 && w[k + 1] == 139) { if (bgzf_magic_at(bam + base + k)) m |= 1u << k; }
    } else {
        for (int k = 0; k < 16; ++k) if (base + k + 18 <= len && bgzf_magic_at(bam + base + k)) m |= 1u << k;
    }
    return m;
}

__global__ __launch_bounds__(256) void k_magic_count(const uint8_t *__restrict__ bam, uint64_t len, uint32_t *tile_cnt) {
    __shared__ uint32_t s_wave[4];
    const uint64_t base = (uint64_t)blockIdx.x * kMagicTile + threadIdx.x * 16;
    const uint32_t c = (uint32_t)__popc(magic_mask16(bam, len, base));
    uint32_t tot; block_excl_scan_256(c, s_wave, tot);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_magic_fill(const uint8_t *__restrict__ bam, uint64_t len, const uint32_t *__restrict__ tile_base, uint64_t *cand) {
    __shared__ uint32_t s_wave[4];
    const uint64_t base = (uint64_t)blockIdx.x * kMagicTile + threadIdx.x * 16;
    uint32_t m = magic_mask16(bam, len, base);
    uint32_t tot; uint32_t slot = tile_base[blockIdx.x] + block_excl_scan_256((uint32_t)__popc(m), s_wave, tot);
    while (m) { const uint32_t k = (uint32_t)__ffs((int)m) - 1; m &= m - 1; cand[slot++] = base + k; }
}

__global__ void k_member_link(const uint8_t *__restrict__ bam, uint64_t len, const uint64_t *__restrict__ cand, uint32_t n, uint32_t *next,
                              uint32_t *isize, uint32_t *reach, uint64_t root2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t off = cand[i];
    const uint64_t blen = (uint64_t)ld16(bam + off + 16) + 1;       // bgzf.c:525
    uint32_t nx = n, isz = 0xffffffffu;                              // malformed member: chain ends, never inflated
    if (blen >= 26 && off + blen <= len) {
        isz = ld32(bam + off + blen - 4);
        if (isz == 0xffffffffu) isz = 0xfffffffeu;                       // ~0 is the mark of an unusable member; as a footer value both just mean "too big"
        const uint64_t want = off + blen;
        uint32_t lo = i + 1, hi = n;                                 // candidates are sorted by offset
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cand[mid] < want) lo = mid + 1; else hi = mid; }
        if (lo < n && cand[lo] == want) nx = lo;
    } else nx = 0xffffffffu;                                         // marks "this candidate itself is unusable"
    next[i] = nx; isize[i] = isz;
    // chain roots: the start of the file, and (root2 != ~0) the compressed offset a seek lands on -- bgzf_seek does not care whether
    // the members in front of it still chain up
    reach[i] = ((i == 0 && off == 0) || off == root2) ? 1u : 0u;
}

__global__ void k_member_jump(uint32_t n, const uint32_t *__restrict__ next_in, uint32_t *next_out, uint32_t *reach) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t nx = next_in[i];
    if (nx < n) {
        if (reach[i]) reach[nx] = 1u;                                // 0 -> 1 only: benign race
        next_out[i] = next_in[nx];
    } else next_out[i] = nx;
}

__global__ void k_member_compact(const uint8_t *__restrict__ bam, uint64_t len, const uint64_t *__restrict__ cand, const uint32_t *__restrict__ isize,
                                 const uint32_t *__restrict__ reach, const uint32_t *__restrict__ rank, uint32_t n, Member *members, uint32_t *isize_compact) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !reach[i]) return;
    const uint64_t off = cand[i];
    const uint32_t blen = (uint32_t)ld16(bam + off + 16) + 1;
    const uint32_t r = rank[i];
    // inflate_block (bgzf.c:292-316) hands zlib block_length - 16 bytes from offset 18: the payload AND the footer (a stream that ends late
    // eats CRC bytes instead of failing, e.g. when BSIZE is one short).  Here: everything up to the member's end, clipped so that the
    // decoder's 16-byte look-ahead stays inside the documented bam_len + 8 readable bytes.
    Member m; m.cpos = off + 18; m.upos = 0; m.isize = isize[i];
    m.clen = blen >= 26 ? blen - 18 : 0;
    if (m.cpos + m.clen + 8 > len) m.clen = len > m.cpos + 8 ? (uint32_t)(len - 8 - m.cpos) : 0;
    members[r] = m;
    isize_compact[r] = (isize[i] <= kBgzfMaxBlock) ? isize[i] : 0u;  // oversized/unusable members hold no bytes in the arena
}

// one block (the member count lives on the device; no host round trip): 256 threads x 8 consecutive members per trip
__global__ __launch_bounds__(256) void k_member_upos(Member *members, const uint32_t *__restrict__ isz, const uint32_t *__restrict__ n_ptr, uint64_t *total) {
    __shared__ uint32_t s_wave[4];
    const uint32_t n = *n_ptr;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += 2048) {
        const uint32_t i0 = base + threadIdx.x * 8;
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = i0 + k < n ? isz[i0 + k] : 0u; sum += v[k]; }     // 2048 * 65536 < 2^32
        uint32_t tot; uint32_t ex = block_excl_scan_256(sum, s_wave, tot);
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (i0 + k < n) members[i0 + k].upos = carry + ex; ex += v[k]; }
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void k_member_query(const Member *__restrict__ members, const uint32_t *__restrict__ n_ptr, const uint64_t *__restrict__ q_coff, uint32_t n_q,
                               uint32_t *q_index, uint64_t *q_upos) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_q) return;
    const uint32_t n = *n_ptr;
    const uint64_t want = q_coff[k] + 18;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (members[mid].cpos < want) lo = mid + 1; else hi = mid; }
    const bool hit = lo < n && members[lo].cpos == want;
    q_index[k] = hit ? lo : n;
    q_upos[k] = hit ? members[lo].upos : ~0ull;
}

// the true lengths a probe launch found replace the ISIZE footers (members and the compact copy the offsets are scanned from)
__global__ void k_member_fix(Member *members, uint32_t *isize_compact, const uint32_t *__restrict__ n_ptr, const uint32_t *__restrict__ fix) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    const uint32_t sz = fix[i];
    members[i].isize = sz;
    isize_compact[i] = sz <= kBgzfMaxBlock ? sz : 0u;
}
__global__ void k_member_stop(const Member *__restrict__ members, const uint32_t *__restrict__ n_ptr, const uint32_t *__restrict__ from_ptr, uint32_t *stop) {
    const uint32_t n = *n_ptr;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *from_ptr) return;
    if (i < n && (members[i].isize == 0 || members[i].isize > kBgzfMaxBlock)) atomicMin(stop, i);
}

void launch_magic_count(const uint8_t *bam, uint64_t len, uint32_t n_tiles, uint32_t *tile_cnt, hipStream_t stream) {
    if (n_tiles) hipLaunchKernelGGL(k_magic_count, dim3(n_tiles), dim3(256), 0, stream, bam, len, tile_cnt);
}
void launch_magic_fill(const uint8_t *bam, uint64_t len, uint32_t n_tiles, const uint32_t *tile_base, uint64_t *cand, hipStream_t stream) {
    if (n_tiles) hipLaunchKernelGGL(k_magic_fill, dim3(n_tiles), dim3(256), 0, stream, bam, len, tile_base, cand);
}
void launch_member_link(const uint8_t *bam, uint64_t len, const uint64_t *cand, uint32_t n, uint32_t *next, uint32_t *isize, uint32_t *reach,
                        uint64_t root2, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_member_link, dim3((n + 255) / 256), dim3(256), 0, stream, bam, len, cand, n, next, isize, reach, root2);
}
void launch_member_jump(uint32_t n, const uint32_t *next_in, uint32_t *next_out, uint32_t *reach, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_member_jump, dim3((n + 255) / 256), dim3(256), 0, stream, n, next_in, next_out, reach);
}
void launch_member_compact(const uint8_t *bam, uint64_t len, const uint64_t *cand, const uint32_t *isize, const uint32_t *reach, const uint32_t *rank, uint32_t n,
                           Member *members, uint32_t *isize_compact, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_member_compact, dim3((n + 255) / 256), dim3(256), 0, stream, bam, len, cand, isize, reach, rank, n, members, isize_compact);
}
void launch_member_upos(Member *members, const uint32_t *isize_compact, const uint32_t *n_members, uint64_t *total, hipStream_t stream) {
    hipLaunchKernelGGL(k_member_upos, dim3(1), dim3(256), 0, stream, members, isize_compact, n_members, total);
}
void launch_member_query(const Member *members, const uint32_t *n_members, const uint64_t *q_coff, uint32_t n_q, uint32_t *q_index, uint64_t *q_upos,
                         hipStream_t stream) {
    if (n_q) hipLaunchKernelGGL(k_member_query, dim3((n_q + 63) / 64), dim3(64), 0, stream, members, n_members, q_coff, n_q, q_index, q_upos);
}
void launch_member_fix(Member *members, uint32_t *isize_compact, uint32_t max_members, const uint32_t *n_members, const uint32_t *fix, hipStream_t stream) {
    if (max_members) hipLaunchKernelGGL(k_member_fix, dim3((max_members + 255) / 256), dim3(256), 0, stream, members, isize_compact, n_members, fix);
}
void launch_member_stop(const Member *members, uint32_t max_members, const uint32_t *n_members, const uint32_t *from, uint32_t *stop, hipStream_t stream) {
    if (max_members) hipLaunchKernelGGL(k_member_stop, dim3((max_members + 255) / 256), dim3(256), 0, stream, members, n_members, from, stop);
}

// =====================================================================================================
// a7. group-by: head flags, segmented reduce, first-seen naming
// =====================================================================================================
__global__ void k_heads(EventSoA ev, const uint32_t *__restrict__ perm, uint32_t n, uint32_t *head) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = 1;
    if (i > 0) {
        const uint32_t a = perm[i], b = perm[i - 1];
        h = (ev.tid[a] != ev.tid[b]) || (ev.start[a] != ev.start[b]) || (ev.ilen_cls[a] != ev.ilen_cls[b]);
    }
    head[i] = h;
}

// seg_excl = exclusive scan of head; the row of sorted position i is seg_excl[i] + head[i] - 1.
// ts_min must be pre-filled with 0xffffffff and te_max with 0.
__global__ void k_reduce(EventSoA ev, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ head,
                         const uint32_t *__restrict__ seg_excl, uint32_t n, UniqueSoA u, uint32_t *head_pos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    uint32_t row = 0, ts = 0xffffffffu, te = 0, hd = 0;
    if (valid) {
        const uint32_t e = perm[i];
        hd = head[i];
        row = seg_excl[i] + hd - 1;
        ts = ev.ts[e]; te = ev.te[e];
        if (hd) {
            head_pos[row] = i;
            u.tid[row] = ev.tid[e]; u.start[row] = ev.start[e]; u.end[row] = ev.start[e] + (ev.ilen_cls[e] >> 2);
            u.first_seen[row] = e;          // stable sort: the first event of the run is the earliest read
        }
    }
    // wave-segmented min/max: runs are contiguous in lane order (keys are sorted)
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t r2 = __shfl_down(row, d, 64), ts2 = __shfl_down(ts, d, 64), te2 = __shfl_down(te, d, 64);
        const bool v2 = __shfl_down((uint32_t)valid, d, 64) != 0;
        if (lane + d < 64 && v2 && r2 == row) { ts = min(ts, ts2); te = max(te, te2); }
    }
    // the first lane of each run in this wave publishes
    const uint32_t prev_row = __shfl_up(row, 1, 64);
    const bool run_head = valid && (lane == 0 || prev_row != row);
    if (run_head) { atomicMin(&u.ts_min[row], ts); atomicMax(&u.te_max[row], te); }
}

// ---- round 4: pre-aggregation -------------------------------------------------------------------------------------------------------
// The events of one junction come close together in file order (the reads that support it start within a read length of each other), so
// most of the group-by can happen before anything is sorted: a workgroup takes kAggTile consecutive events, groups equal keys in an LDS hash
// table (open addressing; a slot holds the tile-local index of the event that claimed it, keys are compared in the staged key columns) and
// writes ONE partial row per distinct key of the tile -- key, number of events, min thick_start, max thick_end, first and last event index.
// The radix sort, the head flags and the reduce then run on the partial rows (bench file 7.5 M events -> ~0.4 M rows, long reads 125 M ->
// a few M) and combine them with sum / min / max, which do not care in which order the tiles appended their rows.  Exact for any input:
// events that do not repeat inside a tile simply stay rows of one.
constexpr uint32_t kAggEmpty = 0xffffffffu;
template <uint32_t kAggTile>
__global__ __launch_bounds__(256) void k_preagg(EventSoA ev, uint32_t n, PartialSoA p, uint32_t *p_total) {
    constexpr uint32_t kAggSlots = 2 * kAggTile;
    __shared__ uint32_t s_tid[kAggTile], s_start[kAggTile], s_ilen[kAggTile], s_slot[kAggSlots];
    __shared__ uint32_t s_cnt[kAggTile], s_ts[kAggTile], s_te[kAggTile], s_first[kAggTile], s_last[kAggTile];
    __shared__ uint32_t s_wave[4], s_base;
    const uint32_t base = blockIdx.x * kAggTile, t = threadIdx.x;
#pragma unroll
    for (uint32_t k = 0; k < kAggTile / 256; ++k) {
        const uint32_t i = t + 256 * k, e = base + i;
        if (e < n) { s_tid[i] = ev.tid[e]; s_start[i] = ev.start[e]; s_ilen[i] = ev.ilen_cls[e]; }
        s_cnt[i] = 0; s_ts[i] = 0xffffffffu; s_te[i] = 0; s_first[i] = 0xffffffffu; s_last[i] = 0;
        s_slot[i] = kAggEmpty; s_slot[i + kAggTile] = kAggEmpty;
    }
    __syncthreads();
    uint32_t mine = 0;                                        // bit k: my event k claimed a slot (it writes the key's row)
#pragma unroll
    for (uint32_t k = 0; k < kAggTile / 256; ++k) {
        const uint32_t i = t + 256 * k, e = base + i;
        if (e < n) {
            const uint32_t kt = s_tid[i], ks = s_start[i], kl = s_ilen[i];
            uint32_t h = ks * 0x9E3779B1u ^ kl * 0x85EBCA6Bu ^ kt * 0xC2B2AE35u;
            h = (h ^ h >> 15) & (kAggSlots - 1);
            uint32_t leader;
            for (;;) {
                const uint32_t prev = atomicCAS(&s_slot[h], kAggEmpty, i);
                if (prev == kAggEmpty) { leader = i; mine |= 1u << k; break; }
                if (s_tid[prev] == kt && s_start[prev] == ks && s_ilen[prev] == kl) { leader = prev; break; }
                h = (h + 1) & (kAggSlots - 1);
            }
            atomicAdd(&s_cnt[leader], 1u);
            atomicMin(&s_ts[leader], ev.ts[e]); atomicMax(&s_te[leader], ev.te[e]);
            atomicMin(&s_first[leader], i); atomicMax(&s_last[leader], i);
        }
    }
    __syncthreads();
    uint32_t tot;
    uint32_t at = block_excl_scan_256((uint32_t)__popc(mine), s_wave, tot);
    if (t == 0) s_base = tot ? atomicAdd(p_total, tot) : 0u;
    __syncthreads();
    at += s_base;
#pragma unroll
    for (uint32_t k = 0; k < kAggTile / 256; ++k) {
        if (mine >> k & 1u) {
            const uint32_t i = t + 256 * k;
            p.tid[at] = s_tid[i]; p.start[at] = s_start[i]; p.ilen_cls[at] = s_ilen[i];
            p.count[at] = s_cnt[i]; p.ts[at] = s_ts[i]; p.te[at] = s_te[i]; p.first[at] = base + s_first[i]; p.last[at] = base + s_last[i];
            ++at;
        }
    }
}

// k_reduce for partial rows: per key the sum of the counts, min / max of the thick bounds, the earliest first and the latest last event.
// u.count / te_max / last_seen must be pre-filled with 0, u.ts_min / first_seen with 0xffffffff.
__global__ void k_reduce_partials(PartialSoA p, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ head,
                                  const uint32_t *__restrict__ seg_excl, uint32_t n, UniqueSoA u) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    uint32_t row = 0, ts = 0xffffffffu, te = 0, cnt = 0, first = 0xffffffffu, last = 0;
    if (valid) {
        const uint32_t e = perm[i];
        const uint32_t hd = head[i];
        row = seg_excl[i] + hd - 1;
        ts = p.ts[e]; te = p.te[e]; cnt = p.count[e]; first = p.first[e]; last = p.last[e];
        if (hd) { u.tid[row] = p.tid[e]; u.start[row] = p.start[e]; u.end[row] = p.start[e] + (p.ilen_cls[e] >> 2); }
    }
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t r2 = __shfl_down(row, d, 64), ts2 = __shfl_down(ts, d, 64), te2 = __shfl_down(te, d, 64), c2 = __shfl_down(cnt, d, 64);
        const uint32_t f2 = __shfl_down(first, d, 64), l2 = __shfl_down(last, d, 64);
        const bool v2 = __shfl_down((uint32_t)valid, d, 64) != 0;
        if (lane + d < 64 && v2 && r2 == row) { ts = min(ts, ts2); te = max(te, te2); cnt += c2; first = min(first, f2); last = max(last, l2); }
    }
    const uint32_t prev_row = __shfl_up(row, 1, 64);
    const bool run_head = valid && (lane == 0 || prev_row != row);
    if (run_head) {
        atomicMin(&u.ts_min[row], ts); atomicMax(&u.te_max[row], te); atomicAdd(&u.count[row], cnt);
        atomicMin(&u.first_seen[row], first); atomicMax(&u.last_seen[row], last);
    }
}
__global__ void k_reduce_finish_partials(const uint8_t *__restrict__ ev_strand, uint32_t n_unique, UniqueSoA u, uint32_t *first_flag) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_unique) return;
    u.strand[r] = ev_strand[u.last_seen[r]];          // newest read's strand wins (junctions_extractor.cc:233)
    first_flag[u.first_seen[r]] = 1;
}

__global__ void k_reduce_finish(EventSoA ev, const uint32_t *__restrict__ perm, uint32_t n, uint32_t n_unique,
                                const uint32_t *__restrict__ head_pos, UniqueSoA u, uint32_t *first_flag) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_unique) return;
    const uint32_t hp = head_pos[r], nx = (r + 1 < n_unique) ? head_pos[r + 1] : n;
    u.count[r] = nx - hp;
    const uint32_t last = perm[nx - 1];
    u.last_seen[r] = last;
    u.strand[r] = ev.strand[last];          // newest read's strand wins (junctions_extractor.cc:233)
    first_flag[u.first_seen[r]] = 1;
}

__global__ void k_name_rank(uint32_t n_unique, const uint32_t *__restrict__ flag_scan, UniqueSoA u) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_unique) u.name_rank[r] = flag_scan[u.first_seen[r]] + 1;   // 1-based rank of first occurrence
}

__global__ void k_gather_u32(uint32_t n, const uint32_t *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t *out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = table[idx[r]];
}

// the unique rows in final order as ten contiguous u32 columns (tid,start,end,ts,te,count,name_rank,first_seen,last_seen,strand):
// one block, one copy to the host
__global__ void k_rows_out(UniqueSoA u, const uint32_t *__restrict__ order, uint32_t n, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = order[i];
    const size_t N = n;
    out[i] = u.tid[s]; out[N + i] = u.start[s]; out[2 * N + i] = u.end[s]; out[3 * N + i] = u.ts_min[s]; out[4 * N + i] = u.te_max[s];
    out[5 * N + i] = u.count[s]; out[6 * N + i] = u.name_rank[s]; out[7 * N + i] = u.first_seen[s]; out[8 * N + i] = u.last_seen[s];
    out[9 * N + i] = u.strand[s];
}

// one row of the host table block (layout: kernels.h table_block_rows)
__device__ __forceinline__ void table_row(uint8_t *out, size_t m, uint32_t i, uint32_t tid, uint32_t start, uint32_t end, uint32_t ts, uint32_t te, uint32_t count,
                                          uint64_t name_index, uint64_t first, uint64_t last, uint32_t strand, uint32_t min_anchor) {
    uint64_t *q8 = (uint64_t *)out;
    q8[i] = name_index; q8[m + i] = first; q8[2 * m + i] = last;
    uint32_t *q4 = (uint32_t *)(out + m * 24);
    q4[i] = tid; q4[m + i] = start; q4[2 * m + i] = end; q4[3 * m + i] = ts; q4[4 * m + i] = te; q4[5 * m + i] = count;
    uint8_t *q1 = out + m * 48;
    q1[i] = (uint8_t)strand;
    q1[m + i] = (uint32_t)(start - ts) >= min_anchor;       // OR over reads of (start - thick_start >= a) == the test on the minimum
    q1[2 * m + i] = (uint32_t)(te - end) >= min_anchor;
}
__global__ void k_rows_table(UniqueSoA u, const uint32_t *__restrict__ order, uint32_t n, uint32_t min_anchor, uint8_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = order[i];
    table_row(out, table_block_rows(n), i, u.tid[s], u.start[s], u.end[s], u.ts_min[s], u.te_max[s], u.count[s], u.name_rank[s], u.first_seen[s], u.last_seen[s], u.strand[s], min_anchor);
}

__global__ void k_fill_u32(uint32_t *p, uint32_t v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

void launch_heads(EventSoA ev, const uint32_t *perm, uint32_t n, uint32_t *head, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_heads, dim3((n + 255) / 256), dim3(256), 0, stream, ev, perm, n, head);
}
void launch_reduce(EventSoA ev, const uint32_t *perm, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, UniqueSoA u,
                   uint32_t *head_pos, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_reduce, dim3((n + 255) / 256), dim3(256), 0, stream, ev, perm, head, seg_excl, n, u, head_pos);
}
void launch_reduce_finish(EventSoA ev, const uint32_t *perm, uint32_t n, uint32_t n_unique, const uint32_t *head_pos, UniqueSoA u,
                          uint32_t *first_flag, hipStream_t stream) {
    if (n_unique) hipLaunchKernelGGL(k_reduce_finish, dim3((n_unique + 255) / 256), dim3(256), 0, stream, ev, perm, n, n_unique, head_pos, u, first_flag);
}
void launch_name_rank(uint32_t n_unique, const uint32_t *flag_scan, UniqueSoA u, hipStream_t stream) {
    if (n_unique) hipLaunchKernelGGL(k_name_rank, dim3((n_unique + 255) / 256), dim3(256), 0, stream, n_unique, flag_scan, u);
}
void launch_gather_u32(uint32_t n, const uint32_t *table, const uint32_t *idx, uint32_t *out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_gather_u32, dim3((n + 255) / 256), dim3(256), 0, stream, n, table, idx, out);
}
void launch_rows_out(UniqueSoA u, const uint32_t *order, uint32_t n, uint32_t *out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_rows_out, dim3((n + 255) / 256), dim3(256), 0, stream, u, order, n, out);
}
void launch_rows_table(UniqueSoA u, const uint32_t *order, uint32_t n, uint32_t min_anchor, uint8_t *out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_rows_table, dim3((n + 255) / 256), dim3(256), 0, stream, u, order, n, min_anchor, out);
}
void launch_preagg(EventSoA ev, uint32_t n, PartialSoA p, uint32_t *p_total, hipStream_t stream) {
    if (!n) return;
    // (measured, round 4: 2048 events per tile = 80 KB of LDS, two workgroups per CU: reduce 1.52 ms on the bench file against 1.40-1.42 with 1024)
    hipLaunchKernelGGL(k_preagg<1024>, dim3((n + 1023) / 1024), dim3(256), 0, stream, ev, n, p, p_total);
}
void launch_reduce_partials(PartialSoA p, const uint32_t *perm, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, UniqueSoA u, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_reduce_partials, dim3((n + 255) / 256), dim3(256), 0, stream, p, perm, head, seg_excl, n, u);
}
void launch_reduce_finish_partials(const uint8_t *ev_strand, uint32_t n_unique, UniqueSoA u, uint32_t *first_flag, hipStream_t stream) {
    if (n_unique) hipLaunchKernelGGL(k_reduce_finish_partials, dim3((n_unique + 255) / 256), dim3(256), 0, stream, ev_strand, n_unique, u, first_flag);
}
void launch_fill_u32(uint32_t *p, uint32_t v, size_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, v, n);
}

}  // namespace rgx
