// inflate_coop.h -- raw DEFLATE (RFC 1951), one BGZF member per GPU LANE, long LZ77 copies moved by the WAVE.
//
// Replaces, for the device path, bgzf.c:292-316 inflate_block (zlib inflate, windowBits -15) of /root/reference/src/utils/htslib.
// Round 3 form of inflate_core.h's decoder.  Same per-lane Huffman machinery (register-resident canonical codes, symbol lists behind
// `Tab`, one symbol per trip, output through the 16-byte register stage OutStage); what changed is who moves a long match.
//
// inflate_raw has every lane copy its own matches, 16 bytes per load and per store: 64 lanes = 64 different cache lines per memory
// instruction, and what the memory system charges for is the NUMBER of requests in flight (profiles/r02_pmc_traffic.json: 4.4x the
// algorithmic traffic, L1 waiting on misses 73 % of the time).  On BAM payloads most output bytes come from a minority of long matches
// (bench file: 25 % of the matches are 129-258 bytes long and carry 92 % of the bytes, tools/lab/trip_stats.cpp).  Here a copy of more
// than kLaneCopyMax bytes is handed to the wave: per round, four such copies are served by sixteen lanes each -- lane j of a group loads
// 16 bytes at (destination chunk j) - distance and stores them to destination chunk j, ALIGNED, so a 258-byte match is two or three
// full lines read and written by one load and one store instruction instead of 2 x 17 single-line requests spread over three trips.
// The owner lane only completes the chunk its stage holds (the copy's first 1..15 bytes) and keeps the copy's last partial chunk as its new
// stage: the invariant "every output byte reaches memory once, in an aligned 16-byte store" survives.  Shorter copies stay per-lane.
//
// Round 4: the block header is a CALL (block_header_cold below): inlined, its tables' registers were the symbol loop's too and the allocator
// spilled copy registers inside the trip -- a load waited for on the spot, then stored to scratch.  With the header out of line the loop needs
// 167 registers and spills none, and its only wait for memory is the one spelled out at (R).  Two further designs were built on top of this,
// exact, measured and not kept (tools/lab/r4_deferred_cold_and_group_head_tail.patch): cold symbol-list entries read without
// blocking the trip, and the head / tail pieces of a long copy loaded by the group's spare lanes and handed over through LDS.
//
// The per-lane part is plain C++ over two accessors -- Tab (symbol lists) and Coop (the wave) -- so that tests/hostemu runs it on the
// host against zlib with a one-lane "wave" (HostCopy below).
#pragma once
#include "inflate_core.h"

namespace rgx {

#ifndef RGX_LAB_LANE_COPY_MAX
#define RGX_LAB_LANE_COPY_MAX 64
#endif
constexpr uint32_t kLaneCopyMax = RGX_LAB_LANE_COPY_MAX;     // bytes a lane copies itself per trip (4 chunk registers); longer pieces go to the wave
constexpr uint32_t kCoopCopyMax = 258;    // a whole match: head (< 16) + at most 16 aligned chunks + tail (< 16)

// one-lane "wave" of the host build: the body chunks are copied on the spot
struct HostCopy {
    RGX_HD bool any(bool b) const { return b; }
    RGX_HD void begin(bool want, uint8_t *out, uint32_t o_body, uint32_t dist, uint32_t nb) {
        if (!want) return;
        for (uint32_t j = 0; j < nb; ++j) st128(out + o_body + 16 * j, ld128(out + o_body + 16 * j - dist));
    }
    RGX_HD void end() {}
};

#ifdef RGX_LAB_TRIPS
static __device__ uint32_t rgx_lab_trips_total;      // (lab builds: wave trips of a launch)
#endif
#ifdef RGX_HOST_TRIPS
static uint64_t rgx_host_trips;                      // (lab tools on the host: trips of the one-lane wave)
#endif
// Every lane of the wave must call this together (Coop::any / begin / end are wave-wide); `active` = false: the lane has no member and
// only serves the others' copies.  Returns an InflateStatus; *out_len = bytes produced (all of them in memory on return, also after an error).
template <class BR, class Tab, class Coop>
RGX_HD int inflate_coop(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len, Tab &T, Coop &C, bool active,
                        uint32_t mode /* same for every lane of the wave.  bit 0: a literal and the symbol behind it in one trip; bit 1 (with bit 0): a
                                         second literal may have a MATCH or the end of the block behind it in the same trip; bit 2: a match at a
                                         distance of 16 or less is a run written from registers, 64 bytes a trip (expand_run, inflate_core.h) */, uint32_t *in_used = nullptr) {
    const bool pairs = (mode & 1u) != 0, runs = (mode & 3u) == 3u, regruns = (mode & 4u) != 0;
#ifdef RGX_LAB_TRIPS
    uint32_t rgx_lab_trips = 0;
#endif
    BR br;                                                     // BitReader, or BitReaderWin where trips are memory-bound (inflate_core.h)
    br.p = in; br.in = in; br.in_len = in_len; br.buf = 0; br.cnt = 0; br.idle();
    if (active) br.init(in, in_len);
    OutStage S; S.init(out, out_cap);
    uint32_t o = 0;
    int status = INF_OK;
    uint32_t last = 0;
    bool in_symbols = false, done = false, fin = !active;
    uint32_t pend_len = 0, pend_dist = 0;
    Code LL, DD;
#pragma unroll
    for (int k = 0; k < 16; ++k) { LL.c[k] = 0; DD.c[k] = 0; }
    u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;   // copy registers: loaded (A) and consumed (C) under the same predicate; never re-initialised

    for (;;) {
#ifdef RGX_LAB_TRIPS
        ++rgx_lab_trips;
#endif
#ifdef RGX_HOST_TRIPS
        ++rgx_host_trips;
#endif
        // ---- A: loads of the pending copy -----------------------------------------------------------------------
        const bool copying = !fin && pend_len != 0;
        uint32_t n = 0, head = 0, nb = 0, tail = 0;
        bool coop = false, rl = false;
        if (copying) {
            n = pend_len < kCoopCopyMax ? pend_len : kCoopCopyMax;
            // a run: the period is in the 16 bytes at (o - distance), the trip writes 64 bytes of it instead of doubling 1, 2, 4 ... bytes a trip
            rl = regruns && pend_dist <= 16 && pend_len > pend_dist;
            if (rl) n = n < kLaneCopyMax ? n : kLaneCopyMax;
            else if (pend_dist < n) n = pend_dist;                  // only bytes that are already produced (any distance >= 1)
            coop = n > kLaneCopyMax;
            if (pend_dist < n + 16) S.flush_partial(o);             // ... and the last < 16 of those may still be in the stage
            const uint8_t *s = out + o - pend_dist;
            if (!coop) {
                v0 = ld128(s);
                if (!rl) {
                    if (n > 16) v1 = ld128(s + 16);
                    if (n > 32) v2 = ld128(s + 32);
                    if (n > 48) v3 = ld128(s + 48);
                }
            } else {
                const uint32_t k = (S.a + o) & 15u;
                head = k ? 16u - k : 0u;                            // completes the chunk the stage holds (this lane)
                nb = (n - head) >> 4;                               // whole destination chunks: the wave's
                tail = (n - head) & 15u;                            // the copy's last partial chunk: the new stage (this lane)
                if (head) v0 = ld128(s);
                if (tail) v1 = ld128(s + head + 16 * nb);
            }
        }
        C.begin(coop, out, o + head, pend_dist, nb);
        // ---- B: next symbol (only when the copy, if any, ends with this batch) -------------------------------------
        uint32_t lit = 256, lit2 = 256, new_len = 0, new_dist = 0;
        if (!fin && pend_len == n && !done) do {
            if (in_symbols) {
                // Up to three code lookups, ONE non-literal behind them.  The first symbol of a trip is covered by the >= 48 bits R leaves in the
                // buffer (15 + 5 + 15 + 13).  mode bit 0: a literal and the symbol behind it in one trip when 48 bits are still there (pays where
                // literals are frequent -- random bases and qualities: 12 % off the kernel -- and costs where the trips are run-length copies
                // that decode nothing -- long reads: +11 %: the host decides per launch from the file's compression ratio, launch_inflate).
                // mode bits 0 + 1 (round 4): symbols behind the first are taken when every bit of them is in the buffer, counted exactly
                // instead of by the worst case, and a second literal may have a MATCH (or the end of the block) behind it: on BAM payloads
                // two literals between two matches is the usual shape (bench file: two thirds of the literal runs are pairs), and that was
                // two trips -- the pair, then the match alone, a trip that moves nothing (2,030 -> 1,628 trips per member).  A third
                // literal, a code that is no code, bits that are not there yet: left where they are for the next trip, which reports errors.
                uint32_t sym = 0x7fffffffu, l = 0, q = 0;
                uint64_t sb; uint32_t sc;                         // the bit buffer in front of the symbol being looked at
                for (;;) {
                    sb = br.buf; sc = br.cnt;
                    const uint32_t idx = code_lookup(LL, rev15(br.peek(15)), l);
                    if (l == 0 || idx >= 288) { if (q == 0) { status = INF_BAD_CODE; fin = true; } break; }
                    const uint32_t s1 = T.get_ll_sym(idx);
                    if (s1 >= 256) { sym = s1; break; }
                    if (q == 2) break;                             // (a third literal stays)
                    br.drop(l);
                    if (q == 0) lit = s1; else lit2 = s1;
                    ++q;
                    if (!(runs ? (q < 3 && br.cnt >= 15) : (pairs && q < 2 && br.cnt >= 48))) break;
                }
                if (sym == 0x7fffffffu) break;                     // literals only (or an error)
                bool fits = true;                                  // (q > 0: the symbol is taken if all of it is in the buffer)
                br.drop(l);
                if (sym == 256) {
                    if (q == 0 || sc >= l) {
                        in_symbols = false;
                        if (br.overran()) { status = INF_IN_OVERRUN; fin = true; break; }
                        if (last) done = true;
                    } else fits = false;
                } else {
                    const uint32_t c = sym - 257;
                    if (c > 28) { if (q == 0) { status = INF_BAD_CODE; fin = true; break; } fits = false; }
                    else {
                        uint32_t e = 0;
                        if (c < 8) new_len = 3 + c;
                        else if (c == 28) new_len = 258;
                        else { e = (c >> 2) - 1; new_len = ((4 + (c & 3)) << e) + 3 + br.bits(e); }
                        uint32_t dl;
                        const uint32_t didx = code_lookup(DD, rev15(br.peek(15)), dl);
                        const uint32_t dsym = T.get_d_sym(didx & 31u);
                        if (dl == 0 || didx >= 32 || dsym > 29) { if (q == 0) { status = INF_BAD_CODE; fin = true; break; } fits = false; }
                        else {
                            br.drop(dl);
                            uint32_t de = 0;
                            if (dsym < 4) new_dist = 1 + dsym;
                            else { de = (dsym >> 1) - 1; new_dist = ((2 + (dsym & 1)) << de) + 1 + br.bits(de); }
                            fits = q == 0 || sc >= l + e + dl + de;
                        }
                    }
                }
                if (!fits) { br.buf = sb; br.cnt = sc; new_len = 0; new_dist = 0; }
            } else if (!copying) {
                // block header (rare, heavy): only with no copy in flight, so that it may write output itself
                const int r = block_header_call(br, T, LL, DD, in, in_len, out, o, out_cap, last, status, S);
                if (status != INF_OK) { fin = true; break; }
                if (r) in_symbols = true;
                else if (last) { if (br.overran()) { status = INF_IN_OVERRUN; fin = true; break; } done = true; }
            }
        } while (0);
        // ---- R: the trip's one memory wait (inflate_core.h on why it is spelled out): fold in the word prefetched a trip ago, prefetch the next
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), expcnt/lgkmcnt untouched
#endif
        C.end();                                                    // the wave's stores of this trip's long copies
        if (!fin) br.ensure(48);
        // ---- C: output -------------------------------------------------------------------------------------------------
        if (copying) {                                              // (an error in B leaves the copy of this trip to be finished: its loads are done)
            if (rl) expand_run(v0, v1, v2, v3, pend_dist);
            if (!coop) {
#define RGX_PUT(J, V)                                                                                                            \
                if (n > 16u * (J)) {                                                                                                 \
                    const uint32_t nbytes = n - 16u * (J) < 16u ? n - 16u * (J) : 16u;                                               \
                    S.put_chunk(o + 16u * (J), (uint64_t)(V)[0] | (uint64_t)(V)[1] << 32, (uint64_t)(V)[2] | (uint64_t)(V)[3] << 32, nbytes); \
                }
                RGX_PUT(0, v0) RGX_PUT(1, v1) RGX_PUT(2, v2) RGX_PUT(3, v3)
#undef RGX_PUT
            } else {
                if (head) S.put_chunk(o, (uint64_t)v0[0] | (uint64_t)v0[1] << 32, (uint64_t)v0[2] | (uint64_t)v0[3] << 32, head);   // full: goes to memory
                uint64_t tl = (uint64_t)v1[0] | (uint64_t)v1[1] << 32, th = (uint64_t)v1[2] | (uint64_t)v1[3] << 32;
                const uint32_t mb = 8 * tail;                       // keep the low `tail` bytes (0..15)
                tl &= mb >= 64 ? ~0ull : ((1ull << (mb & 63)) - 1);
                th &= mb <= 64 ? 0ull : ((1ull << ((mb - 64) & 63)) - 1);
                S.lo = tl; S.hi = th;
            }
            o += n; pend_len -= n;
            // an overlapping copy is periodic with period pend_dist, so 2 * pend_dist is as good a distance for the rest (a run written from
            // registers: the largest period x 2^k that the 64 bytes just written hold)
            if (rl) {
#pragma unroll
                for (int k = 0; k < 6; ++k) if (2 * pend_dist <= n) pend_dist += pend_dist;       // (n: the bytes this trip wrote -- all of them periods)
            } else if (n == pend_dist) pend_dist += pend_dist;
        }
        // (an error met while decoding the trip's second symbol: the literal in front of it is not written -- what a failed member left
        //  behind is never read, the stream ends where the member starts)
        if (!fin) {
            if (lit < 256) {
                if (o >= out_cap) { status = INF_OUT_OVERFLOW; fin = true; }
                else { S.put_byte(o, lit); ++o; }
            }
            if (!fin && lit2 < 256) {
                if (o >= out_cap) { status = INF_OUT_OVERFLOW; fin = true; }
                else { S.put_byte(o, lit2); ++o; }
            }
        }
        if (!fin) {
            if (new_len) {
                if (new_dist > o) { status = INF_BAD_DIST; fin = true; }
                else if (o + new_len > out_cap) { status = INF_OUT_OVERFLOW; fin = true; }
                else { pend_len = new_len; pend_dist = new_dist; }
            }
            if (done && pend_len == 0) fin = true;
        }
        if (!C.any(!fin)) break;
    }
#ifdef RGX_LAB_TRIPS
    if (threadIdx.x == 0) atomicAdd(&rgx_lab_trips_total, rgx_lab_trips);
#endif
    if (active) S.flush_partial(o);                                 // the tail chunk (also on errors: what was produced is in memory)
    if (in_used) *in_used = (uint32_t)(((uint64_t)(br.p - br.in) * 8 - br.cnt + 7) / 8);
    *out_len = o;
    return status;
}

}  // namespace rgx
