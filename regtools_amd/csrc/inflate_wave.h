// inflate_wave.h -- raw DEFLATE (RFC 1951) for one BGZF member per WAVEFRONT: the small-input form of the decoder.
//
// Replaces, for the device path, bgzf.c:292-316 inflate_block (zlib inflate, windowBits -15) of /root/reference/src/utils/htslib.
// k_inflate gives every member one LANE: 196,608 members in flight fill the chip, but ONE member takes that lane ~8 ms (9,000 symbol trips of
// ~2,000 cycles), so a launch costs 8 ms however few members it has -- a 70-member file, a small -r query, the tail of anything.  Here a
// member gets a whole wave and the whole member lives in LDS:
//   * the 64 lanes run the SAME decoder state (bit buffer, position) redundantly -- no divergence, table reads are broadcasts;
//   * Huffman decode is one lookup in a direct table indexed by the next 10 (literal/length) or 8 (distance) bits, built per block in LDS;
//     codes longer than that take the canonical bit-by-bit walk (RFC 1951 3.2.2) over the sorted symbol list;
//   * a match is copied by all lanes at once inside the 64 KiB output window in LDS (lane i moves byte i; an overlapping match is its own
//     period, out[o + i] = out[o - dist + i % dist]); the window goes to HBM once, in 16-byte stores of consecutive lanes.
// ~150-500 cycles per symbol instead of ~2,000 per lane trip: a member in 1-2 ms.  Two workgroups per CU (69 KB of LDS each), so it only
// pays below a few thousand members; the host picks (api_front.cpp).
// The algorithm is plain C++ over a `Wave` policy (lane loops, shared memory) so that tests/hostemu runs it on the host against zlib.
#pragma once
#include "inflate_core.h"

namespace rgx {

constexpr uint32_t kWvLLBits = 10, kWvDBits = 8;
constexpr uint32_t kWvWindow = kBgzfMaxBlock;          // a BGZF member inflates to at most 64 KiB (bgzf.h:42)

// what a wave keeps in LDS (host: a plain struct)
struct WaveShared {
    uint8_t  window[kWvWindow];                        // the member's output
    uint16_t ll_fast[1u << kWvLLBits];                 // (symbol << 4 | code length), 0 = the code is longer than the index
    uint16_t d_fast[1u << kWvDBits];
    uint16_t ll_sorted[288], d_sorted[32];             // symbols ordered by (code length, symbol): the canonical walk's list
    uint16_t ll_count[16], d_count[16];                // codes per length
    uint8_t  lens[320];                                // code lengths of the block being set up
};

// ---- the host's one-thread "wave": lane loops run 0..63 in turn ------------------------------------------------------------------------
struct HostWave {
    template <class F> RGX_HD void lanes(F f) const { for (uint32_t l = 0; l < 64; ++l) f(l); }
    RGX_HD void sync() const {}
};

// wave-uniform LSB-first bit reader (every lane holds the same state); reads never go more than 8 bytes past the payload
struct WvBits {
    const uint8_t *p, *in; uint32_t in_len; uint64_t buf; uint32_t cnt;
    RGX_HD void init(const uint8_t *i, uint32_t n) { in = i; in_len = n; p = i; buf = 0; cnt = 0; }
    RGX_HD void refill() {                              // afterwards cnt >= 56
        if ((size_t)(p - in) > (size_t)in_len + 8) p = in + in_len + 8;
        buf |= ld64(p) << cnt;
        p += (63u - cnt) >> 3;
        cnt |= 56u;
    }
    RGX_HD void need(uint32_t n) { if (cnt < n) refill(); }
    RGX_HD uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    RGX_HD void drop(uint32_t n) { buf >>= n; cnt -= n; }
    RGX_HD uint32_t bits(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    RGX_HD bool overran() const { return (uint64_t)(p - in) * 8 > (uint64_t)in_len * 8 + cnt; }
};

// The canonical walk (RFC 1951 3.2.2): codes of one length are consecutive integers, shorter codes come first.  Reads the code bit by bit
// (MSB of the code first, as DEFLATE packs Huffman codes) from the low end of the bit buffer.  Returns the symbol or -1.
RGX_HD int wv_walk(WvBits &br, const uint16_t *count, const uint16_t *sorted) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        code |= (int)br.bits(1);
        const int n = count[len];
        if (code - n < first) return sorted[index + (code - first)];
        index += n; first += n; first <<= 1; code <<= 1;
    }
    return -1;
}

// Build the decode tables of one code from n code lengths in S.lens[base ..): counts, sorted list, direct table.
// Returns INF_OK, INF_OVERSUBSCRIBED or INF_INCOMPLETE (zlib inftrees.c: an incomplete set is legal only when empty or a single 1-bit code).
template <class Wave>
RGX_HD int wv_build(const Wave &W, WaveShared &S, uint32_t base, uint32_t n, bool dist) {
    uint16_t *count = dist ? S.d_count : S.ll_count, *sorted = dist ? S.d_sorted : S.ll_sorted, *fast = dist ? S.d_fast : S.ll_fast;
    const uint32_t fbits = dist ? kWvDBits : kWvLLBits;
    uint32_t cnt[16], offs[16];
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    for (uint32_t s = 0; s < n; ++s) ++cnt[S.lens[base + s]];                    // (every lane the same: ~300 steps per block)
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left = left * 2 - (int)cnt[l]; if (left < 0) return INF_OVERSUBSCRIBED; }
    const uint32_t used = n - cnt[0];
    if (left > 0 && !(used == 0 || (used == 1 && cnt[1] == 1))) return INF_INCOMPLETE;
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + cnt[l];
    W.lanes([&](uint32_t lane) { if (lane < 16) count[lane] = (uint16_t)(lane ? cnt[lane] : 0); });
    for (uint32_t s = 0; s < n; ++s) { const uint32_t l = S.lens[base + s]; if (l) sorted[offs[l]++] = (uint16_t)s; }   // (same value from every lane)
    W.sync();
    // direct table: entry e = the code that the low bits of e start with, if it is at most fbits long.  Lane-parallel over the entries: each
    // walks its own index canonically (MSB-first code = bit-reversed low bits of e).
    uint32_t first_of[16], index_of[16];
    { uint32_t first = 0, index = 0; for (int l = 1; l <= 15; ++l) { first_of[l] = first; index_of[l] = index; index += cnt[l]; first = (first + cnt[l]) << 1; } }
    W.lanes([&](uint32_t lane) {
        for (uint32_t e = lane; e < (1u << fbits); e += 64) {
            uint32_t code = 0, entry = 0;
            for (uint32_t l = 1; l <= fbits; ++l) {
                code = code << 1 | ((e >> (l - 1)) & 1u);
                if (code - first_of[l] < cnt[l] && code >= first_of[l]) { entry = (uint32_t)sorted[index_of[l] + (code - first_of[l])] << 4 | l; break; }
            }
            fast[e] = (uint16_t)entry;
        }
    });
    W.sync();
    return INF_OK;
}

// Inflate one raw-DEFLATE stream into `out` (capacity out_cap <= 64 KiB).  Every lane of the wave calls this with the same arguments.
// Returns an InflateStatus; *out_len = bytes produced (all of them are in memory on return, also after an error).
template <class Wave>
RGX_HD int inflate_wave(const Wave &W, WaveShared &S, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len) {
    WvBits br; br.init(in, in_len);
    uint32_t o = 0;
    int status = INF_OK;
    if (out_cap > kWvWindow) out_cap = kWvWindow;
    for (uint32_t last = 0; !last && status == INF_OK;) {
        if (br.overran()) { status = INF_IN_OVERRUN; break; }
        br.need(3);
        last = br.bits(1);
        const uint32_t btype = br.bits(2);
        if (btype == 3) { status = INF_BAD_BTYPE; break; }
        if (btype == 0) {
            br.drop(br.cnt & 7);
            br.need(32);
            const uint32_t len = br.bits(16), nlen = br.bits(16);
            if ((len ^ 0xffff) != nlen) { status = INF_BAD_STORED; break; }
            if (o + len > out_cap) { status = INF_OUT_OVERFLOW; break; }
            const uint8_t *src = br.p - (br.cnt >> 3);                          // first raw byte (whole bytes still in the bit buffer included)
            if ((uint64_t)(src - in) + len > in_len) { status = INF_IN_OVERRUN; break; }
            W.lanes([&](uint32_t lane) { for (uint32_t i = lane; i < len; i += 64) S.window[o + i] = src[i]; });
            W.sync();
            o += len;
            br.p = src + len; br.buf = 0; br.cnt = 0;
            continue;
        }
        if (btype == 1) {
            // fixed code (RFC 1951 3.2.6): lengths 8 x144, 9 x112, 7 x24, 8 x8; 30 distance codes of length 5 (+2 that never occur)
            W.lanes([&](uint32_t lane) { for (uint32_t s = lane; s < 320; s += 64) S.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5); });
            W.sync();
            wv_build(W, S, 0, 288, false);
            wv_build(W, S, 288, 32, true);
        } else {
            br.need(14);
            const uint32_t hlit = br.bits(5) + 257, hdist = br.bits(5) + 1, hclen = br.bits(4) + 4;
            if (hlit > 286 || hdist > 30) { status = INF_BAD_HEADER; break; }
            // the code-length code: 19 lengths of 3 bits in the order of RFC 1951 3.2.7, decoded with its own small canonical walk
            const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl_len[19];
            for (int i = 0; i < 19; ++i) cl_len[i] = 0;
            for (uint32_t i = 0; i < hclen; ++i) { br.need(3); cl_len[ord[i]] = (uint8_t)br.bits(3); }
            uint16_t cl_count[16], cl_sorted[19];
            { uint32_t c[8]; for (int l = 0; l < 8; ++l) c[l] = 0;
              for (int s = 0; s < 19; ++s) ++c[cl_len[s]];
              int left = 1; for (int l = 1; l <= 7; ++l) left = left * 2 - (int)c[l];
              if (left != 0) { status = left < 0 ? INF_OVERSUBSCRIBED : INF_INCOMPLETE; break; }
              uint32_t off[8]; off[1] = 0; for (int l = 1; l < 7; ++l) off[l + 1] = off[l] + c[l];
              for (int l = 0; l < 16; ++l) cl_count[l] = (uint16_t)(l >= 1 && l <= 7 ? c[l] : 0);
              for (int s = 0; s < 19; ++s) if (cl_len[s]) cl_sorted[off[cl_len[s]]++] = (uint16_t)s; }
            const uint32_t n = hlit + hdist;
            uint32_t i = 0, prev = 0;
            while (i < n && status == INF_OK) {
                br.need(24);
                const int sym = wv_walk(br, cl_count, cl_sorted);
                if (sym < 0) { status = INF_BAD_CODE; break; }
                uint32_t rep = 1, val = (uint32_t)sym;
                if (sym >= 16) {
                    if (sym == 16) { if (i == 0) { status = INF_BAD_REPEAT; break; } val = prev; rep = 3 + br.bits(2); }
                    else if (sym == 17) { val = 0; rep = 3 + br.bits(3); }
                    else { val = 0; rep = 11 + br.bits(7); }
                    if (i + rep > n) { status = INF_BAD_REPEAT; break; }
                }
                prev = val;
                for (uint32_t k = 0; k < rep; ++k, ++i) S.lens[i < hlit ? i : 288 + (i - hlit)] = (uint8_t)val;      // (same value from every lane)
            }
            if (status != INF_OK) break;
            if (S.lens[256] == 0) { status = INF_NO_EOB; break; }
            W.sync();
            status = wv_build(W, S, 0, hlit, false);
            if (status != INF_OK) break;
            status = wv_build(W, S, 288, hdist, true);
            if (status != INF_OK) break;
        }
        // ---- symbols ---------------------------------------------------------------------------------------------------------------
        for (;;) {
            br.need(48);                                                        // a whole symbol: 15 + 5 + 15 + 13 bits
            uint32_t e = S.ll_fast[br.peek(kWvLLBits)];
            int sym;
            if (e) { sym = (int)(e >> 4); br.drop(e & 15u); }
            else { sym = wv_walk(br, S.ll_count, S.ll_sorted); if (sym < 0) { status = INF_BAD_CODE; break; } }
            if (sym < 256) {
                if (o >= out_cap) { status = INF_OUT_OVERFLOW; break; }
                S.window[o++] = (uint8_t)sym;                                   // (same byte from every lane)
                continue;
            }
            if (sym == 256) { if (br.overran()) status = INF_IN_OVERRUN; break; }
            const uint32_t c = (uint32_t)sym - 257;
            if (c > 28) { status = INF_BAD_CODE; break; }
            uint32_t len;
            if (c < 8) len = 3 + c;
            else if (c == 28) len = 258;
            else { const uint32_t x = (c >> 2) - 1; len = ((4 + (c & 3)) << x) + 3 + br.bits(x); }
            uint32_t d = S.d_fast[br.peek(kWvDBits)];
            int dsym;
            if (d) { dsym = (int)(d >> 4); br.drop(d & 15u); }
            else { dsym = wv_walk(br, S.d_count, S.d_sorted); if (dsym < 0) { status = INF_BAD_CODE; break; } }
            if (dsym > 29) { status = INF_BAD_CODE; break; }
            uint32_t dist;
            if (dsym < 4) dist = 1 + (uint32_t)dsym;
            else { const uint32_t x = ((uint32_t)dsym >> 1) - 1; dist = ((2 + ((uint32_t)dsym & 1)) << x) + 1 + br.bits(x); }
            if (dist > o) { status = INF_BAD_DIST; break; }
            if (o + len > out_cap) { status = INF_OUT_OVERFLOW; break; }
            // all lanes copy: byte i of the match is byte i % dist of the `dist` bytes in front of it (an overlapping match repeats itself)
            const uint32_t s0 = o - dist;
            W.sync();                                                           // (the literals written above are in the window)
            W.lanes([&](uint32_t lane) {
                for (uint32_t i = lane; i < len; i += 64) S.window[o + i] = S.window[s0 + (dist >= len ? i : i % dist)];
            });
            W.sync();
            o += len;
        }
    }
    // the window goes out once
    W.sync();
    W.lanes([&](uint32_t lane) {
        uint32_t i = lane * 16;
        for (; i + 16 <= o; i += 64 * 16) st128(out + i, *(const u32x4 *)(S.window + i));
        if (i < o) for (uint32_t k = i; k < o; ++k) out[k] = S.window[k];       // (the one lane whose piece is cut by the end)
    });
    *out_len = o;
    return status;
}

}  // namespace rgx
