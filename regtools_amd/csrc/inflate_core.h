// inflate_core.h -- raw DEFLATE (RFC 1951) decoder for one BGZF member, written to run as ONE GPU LANE.
//
// Replaces, for the device path, bgzf.c:292-316 inflate_block (zlib inflate with windowBits -15) of
// /root/reference/src/utils/htslib.  The decoder is a straight-line per-lane state machine so that a
// 64-lane wavefront inflates 64 members at once; the only per-member tables that do not fit in
// registers (the canonical symbol lists and the code-length scratch) live behind the `Tab` accessor,
// which on the device is lane-interleaved LDS (bank = lane % 32, conflict-free) and on the host a plain
// array -- the same code is compiled by g++ for the CPU-side unit tests against zlib.
//
// Huffman decode is table-free in the hot loop: each canonical code is 16 sorted register words (bound, length and
// list offset packed together); 15 compare+select steps on the next 15 bits yield the code's length AND its index in
// the sorted symbol list with no memory access, then ONE symbol-list lookup.  Length/distance bases are arithmetic.
#pragma once
#include "common.h"

namespace rgx {

enum InflateStatus : int {
    INF_OK = 0,
    INF_BAD_BTYPE = 1, INF_BAD_STORED = 2, INF_BAD_HEADER = 3, INF_OVERSUBSCRIBED = 4, INF_INCOMPLETE = 5,
    INF_BAD_REPEAT = 6, INF_NO_EOB = 7, INF_BAD_CODE = 8, INF_BAD_DIST = 9, INF_OUT_OVERFLOW = 10,
    INF_IN_OVERRUN = 11, INF_SIZE_MISMATCH = 12,
};

// ---- host-side table storage (unit tests) ---------------------------------------------------------
// The `Tab` accessor hides where the per-member tables live.  Device (kernels.hip, LdsTab): the two canonical symbol
// lists sit bit-packed in lane-interleaved LDS (288 x 9 bits + 32 x 5 bits = 344 B per lane -> 22 KB per wave -> 7 waves
// per CU) and the code-length scratch (needed only while a block header is parsed) in a global, lane-interleaved buffer.
struct HostTab {
    uint16_t ll_sym[288]; uint8_t d_sym[32]; uint32_t len_words[40];
    static constexpr bool kIsHandle = false;               // the tables themselves (the device's Tab is a handle: pointers into LDS and scratch)
    RGX_HD uint32_t get_ll_sym(uint32_t i) const { return ll_sym[i]; }
    RGX_HD void set_ll_sym(uint32_t i, uint32_t v) { ll_sym[i] = (uint16_t)v; }
    RGX_HD uint32_t get_d_sym(uint32_t i) const { return d_sym[i]; }
    RGX_HD void set_d_sym(uint32_t i, uint32_t v) { d_sym[i] = (uint8_t)v; }
    RGX_HD uint32_t get_len_word(uint32_t w) const { return len_words[w]; }     // 8 code lengths (4 bits each) per word
    RGX_HD void set_len_word(uint32_t w, uint32_t v) { len_words[w] = v; }
    RGX_HD void clear_syms() {}
};
constexpr uint32_t kLenWordsLL = 36, kLenWordsD = 4;   // literal/length lengths in words 0..35, distance lengths in 36..39

// One canonical Huffman code, entirely in registers: 16 words sorted ascending,
//   c[k] = lim[k] << 13 | (k+1) << 9 | offs[k+1]        k = 0..14   (codes of length k+1)
//   c[15] = lim[15] << 13                                 (length field 0 = "no such code")
// lim[k] = exclusive upper bound, left-justified to 15 bits, of all codes of length <= k (lim[0] = 0);
// offs[k+1] = number of symbols with length <= k = index of the first length-(k+1) symbol in the sorted list.
// For the next 15 bits v (MSB-first): the last entry with lim <= v names the code's length and where it sits.
struct Code { uint32_t c[16]; };

// returns the index into the sorted symbol list; len = 0 when v starts no code
RGX_HD uint32_t code_lookup(const Code &C, uint32_t v15, uint32_t &len) {
    const uint32_t key = v15 << 13 | 0x1fffu;
    // The words ascend (bounds never fall, lengths rise), so "the last word <= key" is the LARGEST word <= key, and that is key minus the
    // smallest of the sixteen unsigned differences key - c[k]: a word above key wraps to a difference larger than any real one (c[0] <= 0x1fff
    // <= key always gives a real one).  Sixteen independent subtractions and eight three-way minima -- no compare, hence no condition
    // register between a compare and its select (gfx950 pads every such pair with two wait states: round 6).
    uint32_t d = key - C.c[0];
#pragma unroll
    for (int k = 1; k < 16; k += 2) d = min3u(d, key - C.c[k], k + 1 < 16 ? key - C.c[k + 1] : 0xffffffffu);
    const uint32_t m = key - d;
    len = (m >> 9) & 15u;
    return (m & 0x1ffu) + ((v15 - (m >> 13)) >> (15u - len));
}

// Build one canonical code from n code lengths stored 8 per word starting at word w0.
// kind 0 = literal/length (symbol list via set_ll_sym), 1 = distance (set_d_sym).
// Returns INF_OK, INF_OVERSUBSCRIBED or INF_INCOMPLETE (zlib inftrees.c: an incomplete set is legal only when it is
// empty or a single 1-bit code).
template <class Tab>
RGX_HD int build_code(Tab &T, uint32_t w0, uint32_t n, int kind, Code &C) {
    uint32_t count[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (uint32_t w = 0; w * 8 < n; ++w) {
        uint32_t word = T.get_len_word(w0 + w);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t l = (w * 8 + (uint32_t)j < n) ? (word & 15u) : 0u;
            word >>= 4;
#pragma unroll
            for (int k = 1; k < 16; ++k) count[k] += (l == (uint32_t)k) ? 1u : 0u;
        }
    }
    int left = 1;
    uint32_t code = 0, offs = 0;
    uint32_t offs_of[16];
    offs_of[0] = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left = left * 2 - (int)count[l];                 // (may go negative: over-subscribed, reported below)
        C.c[l - 1] = (code << (15 - l)) << 13 | (uint32_t)l << 9 | offs;   // lim[l-1] = first code of length l, left-justified to 15 bits
        offs_of[l] = offs;
        code += count[l]; offs += count[l];
        code <<= 1;
    }
    C.c[15] = (code >> 1) << 13;   // lim[15]: bound of all codes (15 bits wide already)
    if (left < 0) return INF_OVERSUBSCRIBED;
    if (left > 0 && !(offs == 0 || (offs == 1 && count[1] == 1))) return INF_INCOMPLETE;
    for (uint32_t w = 0; w * 8 < n; ++w) {
        uint32_t word = T.get_len_word(w0 + w);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = w * 8 + (uint32_t)j;
            const uint32_t l = (i < n) ? (word & 15u) : 0u;
            word >>= 4;
            if (l) {
                uint32_t slot = 0;
#pragma unroll
                for (int k = 1; k < 16; ++k) if (l == (uint32_t)k) { slot = offs_of[k]; offs_of[k] = slot + 1; }
                if (kind == 0) T.set_ll_sym(slot, i); else T.set_d_sym(slot, i);
            }
        }
    }
    return INF_OK;
}

// LSB-first bit reader with a one-word look-ahead: the load for the NEXT 32 bits is issued when the current
// word is merged, so its latency hides behind the decode of the following symbols (a lane has no other
// wave to hide behind: 3 waves per CU).  Reads may run up to 8 bytes past the payload (the caller pads the
// buffer; in a BGZF file the footer and the next member follow anyway); consuming bits past the end is
// detected by overran().
struct BitReader {
    const uint8_t *p;      // address of the first payload byte that is not in `buf` yet (`next` was loaded from here)
    const uint8_t *in;     // payload start
    uint32_t in_len;
    uint64_t buf;          // LSB-first bit buffer; bits above cnt are already the right ones (they are OR-ed in again later)
    uint32_t cnt;          // valid bits in buf
    uint64_t next;         // the 8 bytes at p, loaded one refill ahead of their use
    RGX_HD void init(const uint8_t *i, uint32_t n) { in = i; in_len = n; p = i; buf = 0; cnt = 0; next = ld64(p); }
    // Afterwards cnt >= 56: a whole symbol trip (15 + 5 + 15 + 13 bits) never needs memory in the middle.  Whole bytes
    // only: (63 - cnt) >> 3 of them fit, and cnt + 8 * that == cnt | 56.  The load issued here is consumed by the NEXT
    // refill, so the one wait it needs sits a full trip after its issue.
    RGX_HD void refill() {
        buf |= next << cnt;
        p += (63u - cnt) >> 3;
        cnt |= 56u;
        // a corrupt stream may ask for bits that do not exist: never read more than 16 bytes past the payload (the loads stay inside
        // the caller's buffer + its 8-byte slack); overran() is true from here on and the run ends at the next check or at out_cap
        if ((size_t)(p - in) > (size_t)in_len + 8) p = in + in_len + 8;
        next = ld64(p);
    }
    RGX_HD void ensure(uint32_t n) { if (cnt < n) refill(); }
    RGX_HD uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    RGX_HD void drop(uint32_t n) { buf >>= n; cnt -= n; }
    RGX_HD uint32_t bits(uint32_t n) { uint32_t v = peek(n); drop(n); return v; }
    // bits consumed so far = 8 * (p - in) - cnt
    RGX_HD bool overran() const { return (uint64_t)(p - in) * 8 > (uint64_t)in_len * 8 + cnt; }
    // byte-aligned raw access for stored blocks: address of the next unconsumed byte once cnt == 0
    RGX_HD const uint8_t *byte_ptr() const { return p; }
    RGX_HD void restart_at(const uint8_t *q) { p = q; buf = 0; cnt = 0; next = ld64(p); }
    RGX_HD void idle() { next = 0; }
};

// The same reader with a 16-byte look-ahead window in registers: one 16-byte load serves the refills of the next 8 consumed bytes, where
// BitReader loads the 8 bytes at p again at every refill.  On payloads that compress 20x and more a trip consumes about one byte, so that is
// 8x fewer bit-stream requests -- and the bit-stream lines (one per lane, 196 k of them) no longer sit in the L2 being touched every trip,
// crowding out the output lines (FETCH_SIZE of the bench file's launch 32 -> 15 GB).  Costs ~12 VALU instructions per refill: pays where trips
// are memory-bound (k_inflate_coop: -6 %), loses where they are instruction-bound (random bases and qualities: +23 %, which keep BitReader).
struct BitReaderWin {
    const uint8_t *p, *in;
    uint32_t in_len;
    uint64_t buf;
    uint32_t cnt;
    u32x4 win;             // the 16 bytes at p - off, kept as loaded: the load lands in these registers and nothing touches them before the next refill
                           // (as two 64-bit words put together on arrival, every window load was waited for on the spot -- a second memory
                           //  round trip in the middle of most wave trips)
    uint32_t off;          // <= 8 between refills, so the 8 bytes at p are always inside the window
    RGX_HD void load_window() {
        // a corrupt stream may ask for bits that do not exist: never read more than 16 bytes past the payload
        const uint8_t *wp = (size_t)(p - in) > (size_t)in_len ? in + in_len : p;
        off = (uint32_t)(p - wp);
        win = ld128(wp);
    }
    RGX_HD void init(const uint8_t *i, uint32_t n) { in = i; in_len = n; p = i; buf = 0; cnt = 0; load_window(); }
    RGX_HD void refill() {
        const uint32_t sh = 8 * off;
        const uint64_t w0 = (uint64_t)win[0] | (uint64_t)win[1] << 32, w1 = (uint64_t)win[2] | (uint64_t)win[3] << 32;
        const uint64_t next = (sh >= 64 ? 0 : w0 >> sh) | (sh == 0 ? 0 : w1 << ((64 - sh) & 63));      // the 8 bytes at p
        buf |= next << cnt;
#if defined(__HIP_DEVICE_COMPILE__)
        // (the window's new load must not be scheduled in front of these last uses of its old bytes: it would then need registers of its own,
        //  and the copy into `win` a wait)
        asm volatile("" : "+v"(buf) : : "memory");
#endif
        const uint32_t adv = (63u - cnt) >> 3;
        p += adv; off += adv;
        cnt |= 56u;
        // overran() is true from here on when p left the payload, and the run ends at the next check or at out_cap
        if ((size_t)(p - in) > (size_t)in_len + 8) { off -= (uint32_t)((size_t)(p - in) - ((size_t)in_len + 8)); p = in + in_len + 8; }
        if (off > 8) load_window();
    }
    RGX_HD void ensure(uint32_t n) { if (cnt < n) refill(); }
    RGX_HD uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    RGX_HD void drop(uint32_t n) { buf >>= n; cnt -= n; }
    RGX_HD uint32_t bits(uint32_t n) { uint32_t v = peek(n); drop(n); return v; }
    RGX_HD bool overran() const { return (uint64_t)(p - in) * 8 > (uint64_t)in_len * 8 + cnt; }
    RGX_HD const uint8_t *byte_ptr() const { return p; }
    RGX_HD void restart_at(const uint8_t *q) { p = q; buf = 0; cnt = 0; load_window(); }
    RGX_HD void idle() { win = u32x4{0, 0, 0, 0}; off = 0; }
};

RGX_HD uint32_t rev15(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> 17;
#else
    uint32_t r = 0;
    for (int i = 0; i < 15; ++i) r |= ((v >> i) & 1u) << (14 - i);
    return r;
#endif
}

// ---- output staging -------------------------------------------------------------------------------------------------
// Every output byte goes through a 16-byte register chunk aligned on the DESTINATION ADDRESS and reaches memory exactly once,
// as part of one aligned 16-byte store (the chunks cut by the two ends of the member are written byte-exactly).  Why: with one
// lane per member a wave's stores land in 64 different cache lines; what the memory system then pays for is the NUMBER of
// write requests -- a literal stored as a single byte or a 10-byte match stored as an unaligned, line-straddling chunk cost a
// full request each, and tens of thousands of lanes per L2 each dribbling partial writes into their own line evict one another
// (measured: one store per 16 output bytes instead of one per symbol takes a third off the kernel on real-looking payloads).
// Invariant: bytes [0, k) of (lo, hi) are the chunk's valid bytes, k = (a + o) & 15; bytes >= k are zero.
RGX_HD uint64_t shl64(uint64_t x, uint32_t s) { return s >= 64 ? 0 : x << s; }
RGX_HD uint64_t shr64(uint64_t x, uint32_t s) { return s >= 64 ? 0 : x >> s; }

// ---- runs from registers (round 4) -------------------------------------------------------------------------------------------------
// A match whose distance d is at most 16 repeats the d bytes in front of it: given the 16 bytes at (o - d) -- of which the first d are
// output already written, the rest anything -- the next 64 bytes of output are E[j] = P[j mod d], built here without touching memory again.
// (The doubling rule of the symbol loop gets there in log2 steps, one trip -- one memory round trip and one partial-chunk store -- each:
// long reads spend half their trips in such runs.)  x0 = E[0..16): P doubled up four times; chunk c = E[16c..16c+16): with r = 16c mod d,
// bytes [0, d - r) come from x0 >> 8r and the rest from x0 << 8(d - r).
RGX_HD void shl128(uint64_t &lo, uint64_t &hi, uint32_t s) {          // s in bits, any value (>= 128 gives 0)
    const uint64_t nh = shl64(hi, s) | shr64(lo, 64u - s) | shl64(lo, s - 64u);      // (out-of-range counts, also the wrapped ones, give 0)
    lo = shl64(lo, s); hi = nh;
}
RGX_HD void shr128(uint64_t &lo, uint64_t &hi, uint32_t s) {
    const uint64_t nl = shr64(lo, s) | shl64(hi, 64u - s) | shr64(hi, s - 64u);
    hi = shr64(hi, s); lo = nl;
}
RGX_HD void expand_run(u32x4 &v0, u32x4 &v1, u32x4 &v2, u32x4 &v3, uint32_t d /* 1..16 */) {
    uint64_t xl = (uint64_t)v0[0] | (uint64_t)v0[1] << 32, xh = (uint64_t)v0[2] | (uint64_t)v0[3] << 32;
    { const uint32_t mb = 8 * d; xl &= mb >= 64 ? ~0ull : ((1ull << (mb & 63)) - 1); xh &= mb >= 128 ? ~0ull : (mb <= 64 ? 0ull : ((1ull << ((mb - 64) & 63)) - 1)); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { uint64_t tl = xl, th = xh; shl128(tl, th, (8u * d) << k); xl |= tl; xh |= th; }
    const uint32_t r1 = (uint32_t)(0x0123456702410100ull >> (4 * (d - 1))) & 15u;     // 16 mod d, d = 1..16
    uint32_t r = 0;
    u32x4 *dst[3] = {&v1, &v2, &v3};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        r += r1; if (r >= d) r -= d;                                                 // 16 (c + 1) mod d
        uint64_t al = xl, ah = xh, bl = xl, bh = xh;
        shr128(al, ah, 8 * r); shl128(bl, bh, 8 * (d - r));
        const uint32_t mb = 8 * (d - r);                                              // 8..128 bits kept of a
        al &= mb >= 64 ? ~0ull : ((1ull << (mb & 63)) - 1); ah &= mb >= 128 ? ~0ull : (mb <= 64 ? 0ull : ((1ull << ((mb - 64) & 63)) - 1));
        al |= bl; ah |= bh;
        *dst[c] = u32x4{(uint32_t)al, (uint32_t)(al >> 32), (uint32_t)ah, (uint32_t)(ah >> 32)};
    }
    v0 = u32x4{(uint32_t)xl, (uint32_t)(xl >> 32), (uint32_t)xh, (uint32_t)(xh >> 32)};
}

// the two chunks cut by a member's ends, byte by byte (twice per member): a CALL -- inlined, this loop stood at every one of the eighteen places of the
// symbol loop that can complete a chunk, and its bookkeeping took scalar registers the loop needed (61 of them spilled to lanes of a vector register)
static RGX_COLD void store_chunk_edge(uint8_t *out, uint32_t cap, int32_t cb, uint64_t l, uint64_t h) {
#pragma nounroll
    for (int i = 0; i < 16; ++i) {
        const int32_t m = cb + i;
        if (m >= 0 && (uint32_t)m < cap) out[m] = (uint8_t)((i < 8 ? l >> (8 * i) : h >> (8 * (i - 8))) & 0xff);
    }
}

struct OutStage {
    uint8_t *out; uint32_t cap, a;
    uint64_t lo, hi;
    RGX_HD void init(uint8_t *o_, uint32_t cap_) { out = o_; cap = cap_; a = (uint32_t)((uintptr_t)o_ & 15u); lo = 0; hi = 0; }
    // write the chunk that holds member offset o (any offset inside it); bytes outside [0, cap) are never touched
    RGX_HD void store_chunk(uint32_t o, uint64_t l, uint64_t h) const {
        const int32_t cb = (int32_t)((a + o) & ~15u) - (int32_t)a;        // member offset of the chunk's byte 0 (< 0 for the head chunk)
        if (cb >= 0 && (uint32_t)cb + 16 <= cap) {
#ifdef RGX_LAB_DROP_STAGE
            if ((((a + o) >> 4) & 3u) != 3u) return;                       // (lab, WRONG OUTPUT: three of four stage stores dropped -- what whole-sector stores could save at most)
#endif
            u32x4 v = {(uint32_t)l, (uint32_t)(l >> 32), (uint32_t)h, (uint32_t)(h >> 32)};
            *(u32x4 *)(out + cb) = v;                                      // 16-byte aligned by construction
        } else {
            store_chunk_edge(out, cap, cb, l, h);                          // (only the two chunks cut by the member's ends get here)
        }
    }
    // make memory agree with the stage up to o (a copy is about to read bytes that are still in registers)
    RGX_HD void flush_partial(uint32_t o) const { if ((a + o) & 15u) store_chunk(o, lo, hi); }
    // The two appenders are written without data-dependent branches (selects and clamped shifts only): in a 64-lane wave every lane
    // is at a different byte phase k, so an `if (k < 8)` costs both sides for everybody.
    RGX_HD void put_byte(uint32_t o, uint32_t b) {
        const uint32_t k = (a + o) & 15u, s = 8 * k;
        lo |= shl64((uint64_t)b, s);                        // k < 8, else shifted out
        hi |= shl64((uint64_t)b, s - 64u);                  // k >= 8 (s - 64 wraps to a huge count for k < 8: shifted out)
        if (k == 15) { store_chunk(o, lo, hi); lo = 0; hi = 0; }
    }
    // append nb (1..16) bytes held in the low end of (dl, dh); whatever lies above them is ignored
    RGX_HD void put_chunk(uint32_t o, uint64_t dl, uint64_t dh, uint32_t nb) {
        const uint32_t mb = 8 * nb;                                        // keep the low nb bytes
        dl &= mb >= 64 ? ~0ull : ((1ull << (mb & 63)) - 1);
        dh &= mb >= 128 ? ~0ull : (mb <= 64 ? 0ull : ((1ull << ((mb - 64) & 63)) - 1));
        const uint32_t k = (a + o) & 15u, s = 8 * k, r = s & 63u;
        const bool q = s >= 64;
        // (dh:dl) << s as four 64-bit words x0..x3: x0,x1 complete this chunk, x2,x3 spill into the next
        const uint64_t a0 = dl << r, a1 = (dh << r) | shr64(dl, 64 - r), a2 = shr64(dh, 64 - r);
        const uint64_t cl = lo | (q ? 0 : a0), ch = hi | (q ? a0 : a1), rl = q ? a1 : a2, rh = q ? a2 : 0;
        const bool full = k + nb >= 16;
        if (full) store_chunk(o, cl, ch);
        lo = full ? rl : cl; hi = full ? rh : ch;
    }
    // after bytes were written straight to memory (stored blocks): pick the current chunk's valid bytes up again
    RGX_HD void resync(uint32_t o) {
        const uint32_t k = (a + o) & 15u;
        lo = 0; hi = 0;
        if (k) {
            const uint8_t *c = out + o - k;                               // aligned; may start before `out`, inside the same 16-byte block
            const u32x4 v = *(const u32x4 *)c;
            lo = (uint64_t)v[0] | (uint64_t)v[1] << 32; hi = (uint64_t)v[2] | (uint64_t)v[3] << 32;
            if (k < 8) { lo &= (1ull << (8 * k)) - 1; hi = 0; } else hi &= shl64(1ull, 8 * (k - 8)) - 1;
        }
    }
};

// The tables of a fixed (btype 1) or dynamic (btype 2) block, bit reader positioned right after the 3 header bits.
// Returns 1 = symbols follow, 0 = status says why not.
template <class BR, class Tab>
RGX_HD int build_block_codes(BR &br, Tab &T, Code &LL, Code &DD, uint32_t btype, int &status) {
    T.clear_syms();
    if (btype == 1) {
        // fixed code (RFC 1951 3.2.6): lengths 8 x144, 9 x112, 7 x24, 8 x8; 32 distance codes of length 5 (30,31 rejected at decode)
        for (uint32_t w = 0; w < 36; ++w) T.set_len_word(w, w < 18 ? 0x88888888u : w < 32 ? 0x99999999u : w < 35 ? 0x77777777u : 0x88888888u);
        for (uint32_t w = 0; w < 4; ++w) T.set_len_word(kLenWordsLL + w, 0x55555555u);
        build_code(T, 0, 288, 0, LL);
        build_code(T, kLenWordsLL, 32, 1, DD);
        return 1;
    }
    br.ensure(16);
    const uint32_t hlit = br.bits(5) + 257, hdist = br.bits(5) + 1, hclen = br.bits(4) + 4;
    if (hlit > 286 || hdist > 30) { status = INF_BAD_HEADER; return 0; }
    // code-length code: 19 lengths of 3 bits, kept in a register (3 bits each)
    uint64_t cl_lens = 0;
    {
        const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (uint32_t i = 0; i < hclen; ++i) {
            br.ensure(8);
            const uint64_t l = br.bits(3);
            cl_lens |= l << (3 * ord[i]);
        }
    }
    // canonical data for the CL code, entirely in registers: 7 bounds, 19 symbols of 5 bits in 2 regs
    uint32_t cl_count[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) cl_count[l] = 0;
    for (uint32_t sy = 0; sy < 19; ++sy) {
        const uint32_t l = (uint32_t)(cl_lens >> (3 * sy)) & 7;
#pragma unroll
        for (int k = 0; k < 8; ++k) cl_count[k] += (l == (uint32_t)k) ? 1u : 0u;
    }
    uint32_t cl_lim[8], cl_base[8], cl_off[8];
    {
        int left = 1; uint32_t code = 0, offs = 0;
        cl_lim[0] = 0; cl_base[0] = 0; cl_off[0] = 0;
#pragma unroll
        for (int l = 1; l <= 7; ++l) {
            left = left * 2 - (int)cl_count[l];
            cl_base[l] = (offs - code) & 0xff; cl_off[l] = offs;
            code += cl_count[l]; offs += cl_count[l];
            cl_lim[l] = code << (7 - l);
            code <<= 1;
        }
        if (left != 0) { status = left < 0 ? INF_OVERSUBSCRIBED : INF_INCOMPLETE; return 0; }
    }
    uint64_t cl_sym_lo = 0, cl_sym_hi = 0;  // symbol list, 5 bits per slot, slots 0..11 in lo, 12..18 in hi
    for (uint32_t sy = 0; sy < 19; ++sy) {
        const uint32_t l = (uint32_t)(cl_lens >> (3 * sy)) & 7;
        if (l) {
            uint32_t slot = 0;
#pragma unroll
            for (int k = 1; k < 8; ++k) if (l == (uint32_t)k) { slot = cl_off[k]; cl_off[k] = slot + 1; }
            if (slot < 12) cl_sym_lo |= (uint64_t)sy << (5 * slot); else cl_sym_hi |= (uint64_t)sy << (5 * (slot - 12));
        }
    }
    // read hlit + hdist code lengths; they are buffered 8 per word (literal/length words 0.., distance words 36..)
    const uint32_t n = hlit + hdist;
    uint32_t i = 0, prev = 0;
    uint32_t acc = 0, acc_n = 0, acc_w = 0;       // 4-bit lengths being packed, how many, destination word
    uint32_t eob_len = 0;
    while (i < n) {
        br.ensure(16);                                  // 7 (code) + 7 (repeat count)
        const uint32_t v7 = rev15(br.peek(7)) >> 8;   // 7 bits MSB-first
        uint32_t l = 1;
#pragma unroll
        for (int k = 1; k <= 7; ++k) l += (v7 >= cl_lim[k]) ? 1u : 0u;
        if (l > 7) { status = INF_BAD_CODE; return 0; }
        uint32_t base = 0;
#pragma unroll
        for (int k = 1; k <= 7; ++k) if (l == (uint32_t)k) base = cl_base[k];
        const uint32_t slot = (base + (v7 >> (7 - l))) & 0xff;
        const uint32_t sym = slot < 12 ? (uint32_t)(cl_sym_lo >> (5 * slot)) & 31 : (uint32_t)(cl_sym_hi >> (5 * (slot - 12))) & 31;
        br.drop(l);
        uint32_t rep = 1, val = sym;
        if (sym >= 16) {
            if (sym == 16) { if (i == 0) { status = INF_BAD_REPEAT; return 0; } val = prev; rep = 3 + br.bits(2); }
            else if (sym == 17) { val = 0; rep = 3 + br.bits(3); }
            else { val = 0; rep = 11 + br.bits(7); }
            if (i + rep > n) { status = INF_BAD_REPEAT; return 0; }
        }
        prev = val;
        for (uint32_t k = 0; k < rep; ++k) {
            if (i == 256) eob_len = val;
            acc |= val << (4 * acc_n); ++acc_n; ++i;
            if (acc_n == 8 || i == hlit || i == n) {           // word full, or end of the literal/length or distance run
                T.set_len_word(acc_w, acc);
                acc = 0; acc_n = 0;
                acc_w = (i == hlit) ? kLenWordsLL : acc_w + 1;   // the distance lengths restart on their own word
            }
        }
    }
    if (eob_len == 0) { status = INF_NO_EOB; return 0; }
    status = build_code(T, 0, hlit, 0, LL);
    if (status != INF_OK) return 0;
    status = build_code(T, kLenWordsLL, hdist, 1, DD);
    return status == INF_OK ? 1 : 0;
}

// ---- block header ------------------------------------------------------------------------------------------------
// Parses one DEFLATE block header.  Dynamic/fixed blocks: builds the two canonical codes (returns 1 = symbols follow).
// Stored blocks: copies the raw bytes and returns 0 (= another header follows, or the stream ends if *last).
template <class BR, class Tab>
RGX_HD int block_header(BR &br, Tab &T, Code &LL, Code &DD, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t &o,
                        uint32_t out_cap, uint32_t &last, int &status, OutStage &S) {
    if (br.overran()) { status = INF_IN_OVERRUN; return 0; }     // a run of empty stored blocks must not walk off the input
    br.ensure(32);
    last = br.bits(1);
    const uint32_t btype = br.bits(2);
    if (btype == 0) {
        // stored: skip to byte boundary, LEN, NLEN, raw bytes
        br.drop(br.cnt & 7);
        br.ensure(32);
        uint32_t len = br.bits(16);
        const uint32_t nlen = br.bits(16);
        if ((len ^ 0xffff) != nlen) { status = INF_BAD_STORED; return 0; }
        if (o + len > out_cap) { status = INF_OUT_OVERFLOW; return 0; }
        S.flush_partial(o);                                    // raw bytes go straight to memory; the stage is picked up again below
        while (len && br.cnt >= 8) { out[o++] = (uint8_t)br.bits(8); --len; }     // bytes still in the bit buffer
        if (len) {
            const uint8_t *src = br.byte_ptr();   // cnt == 0 here: the prefetched word starts at the next payload byte
            if ((uint64_t)(src - in) + len > in_len) { status = INF_IN_OVERRUN; return 0; }
            uint32_t k = 0;
            for (; k + 16 <= len; k += 16) st128(out + o + k, ld128(src + k));
            for (; k < len; ++k) out[o + k] = src[k];
            o += len; br.restart_at(src + len);
        }
        S.resync(o);
        return 0;
    }
    if (btype == 3) { status = INF_BAD_BTYPE; return 0; }
    return build_block_codes(br, T, LL, DD, btype, status);
}

// The block header (once or a few times per member, thousands of instructions, its own long-lived tables) as a CALL.  The loop's state goes in
// and out by value, so that nothing of it has its address taken where it matters.
template <class BR> struct HeaderState { BR br; Code LL, DD; OutStage S; uint32_t o, last; int status; };
template <class BR, class Tab>
RGX_COLD int block_header_cold(HeaderState<BR> &h, Tab &T, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap) {
    return block_header(h.br, T, h.LL, h.DD, in, in_len, out, h.o, out_cap, h.last, h.status, h.S);
}
// (callers: the symbol loops of inflate_raw below and of inflate_coop, inflate_coop.h)
template <class BR, class Tab>
RGX_HD int block_header_call(BR &br, Tab &T, Code &LL, Code &DD, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t &o, uint32_t out_cap,
                             uint32_t &last, int &status, OutStage &S) {
    HeaderState<BR> h{br, LL, DD, S, o, last, status};
    int r;
    if constexpr (Tab::kIsHandle) { Tab Tc = T; r = block_header_cold(h, Tc, in, in_len, out, out_cap); }      // (a handle: the copy's address escapes, not T's)
    else r = block_header_cold(h, T, in, in_len, out, out_cap);
    // (pointers that went through memory come back without their address space: rebuilt from the arguments)
    { const size_t pofs = (size_t)(h.br.p - h.br.in); br = h.br; br.in = in; br.p = in + pofs; }
    LL = h.LL; DD = h.DD; S = h.S; S.out = out; o = h.o; last = h.last; status = h.status;
    return r;
}

constexpr uint32_t kCopyBatch = 128;   // bytes moved per memory round trip (8 independent 16-byte loads in flight)

// Inflate one raw-DEFLATE stream. Returns an InflateStatus; *out_len = bytes produced.
//
// The symbol loop is a per-lane state machine whose every trip does AT MOST ONE dependent memory round trip:
//   A. issue the loads of this lane's pending LZ77 copy (up to kCopyBatch bytes whose sources are already written),
//   B. while they are in flight, decode the lane's next symbol if the copy ends with this batch (bit buffer only),
//   R. wait once, top the bit buffer up from the word prefetched a trip ago and prefetch the next one,
//   C. push the copied bytes, then the decoded literal, through the output stage (OutStage), or arm the next copy.
// In a 64-lane wavefront every lane is at a different point of a different member; a lane in the middle of a long
// match therefore no longer stalls the 63 others for a whole copy loop -- each trip costs one round trip for all.
// LITS: literals a trip may take (1, or up to 4: a literal whose successors are literals too and still lie in the bit buffer are decoded in the same
// trip -- the trip's memory wait and output logic are paid once for them.  For payloads that are mostly literals: random bases and qualities;
// the host picks per launch, inflate_plan_for).  The symbols decoded and their order are the same for every LITS.
template <int LITS = 1, class Tab>
RGX_HD int inflate_raw(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len, Tab &T, uint32_t *in_used = nullptr) {
    BitReader br; br.init(in, in_len);
    OutStage S; S.init(out, out_cap);
    uint32_t o = 0;
    int status = INF_OK;
    uint32_t last = 0;
    bool in_symbols = false, done = false;
    uint32_t pend_len = 0, pend_dist = 0;
    Code LL, DD;
#pragma unroll
    for (int k = 0; k < 16; ++k) { LL.c[k] = 0; DD.c[k] = 0; }

    u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0, v6 = v0, v7 = v0;   // copy registers: a chunk is loaded (A) and consumed (C)
                                                                                             // under the same predicate; never re-initialised
    for (;;) {
        // ---- A: loads of the pending copy -----------------------------------------------------------------------
        const bool copying = pend_len != 0;
        uint32_t n = 0;
        if (copying) {
            n = pend_len < kCopyBatch ? pend_len : kCopyBatch;
            if (pend_dist < n) n = pend_dist;                       // only bytes that are already produced (any distance >= 1)
            if (pend_dist < n + 16) S.flush_partial(o);             // ... and the last < 16 of those may still be in the stage
            const uint8_t *s = out + o - pend_dist;
            v0 = ld128(s);
            if (n > 16) v1 = ld128(s + 16);
            if (n > 32) v2 = ld128(s + 32);
            if (n > 48) v3 = ld128(s + 48);
            if (n > 64) v4 = ld128(s + 64);
            if (n > 80) v5 = ld128(s + 80);
            if (n > 96) v6 = ld128(s + 96);
            if (n > 112) v7 = ld128(s + 112);
        }
        // ---- B: next symbol (only when the copy, if any, ends with this batch) -------------------------------------
        uint32_t lit = 256, new_len = 0, new_dist = 0;
        uint32_t more[LITS > 1 ? LITS - 1 : 1];                 // the literals behind `lit` taken in this trip (256 = none)
#pragma unroll
        for (int q = 0; q < (LITS > 1 ? LITS - 1 : 1); ++q) more[q] = 256;
        if (pend_len == n && !done) {
            if (in_symbols) {
                const uint32_t v = rev15(br.peek(15));         // >= 48 valid bits here: the whole trip is fed from the buffer
                uint32_t l;
                const uint32_t idx = code_lookup(LL, v, l);
                if (l == 0 || idx >= 288) { status = INF_BAD_CODE; break; }
                const uint32_t sym = T.get_ll_sym(idx);
                br.drop(l);
                if (sym < 256) {
                    lit = sym;
                    if (LITS > 1) {
                        bool run = true;                        // (no break: the lanes of a wave leave the unrolled steps together)
#pragma unroll
                        for (int q = 0; q < LITS - 1; ++q) {
                            // a code is at most 15 bits: with 15 in the buffer the next symbol can be looked at; anything but a literal stays where it is
                            // and is the next trip's symbol (also a code that is no code: that trip reports it)
                            run = run && br.cnt >= 15;
                            uint32_t l2 = 0, idx2 = 0, sym2 = 256;
                            if (run) { idx2 = code_lookup(LL, rev15(br.peek(15)), l2); run = l2 != 0 && idx2 < 288; }
                            if (run) { sym2 = T.get_ll_sym(idx2); run = sym2 < 256; }
                            if (run) { br.drop(l2); more[q] = sym2; }
                        }
                    }
                } else if (sym == 256) {
                    in_symbols = false;
                    if (br.overran()) { status = INF_IN_OVERRUN; break; }
                    if (last) done = true;
                } else {
                    const uint32_t c = sym - 257;
                    if (c > 28) { status = INF_BAD_CODE; break; }
                    if (c < 8) new_len = 3 + c;
                    else if (c == 28) new_len = 258;
                    else { const uint32_t e = (c >> 2) - 1; new_len = ((4 + (c & 3)) << e) + 3 + br.bits(e); }
                    const uint32_t dv = rev15(br.peek(15));
                    uint32_t dl;
                    const uint32_t didx = code_lookup(DD, dv, dl);
                    if (dl == 0 || didx >= 32) { status = INF_BAD_CODE; break; }
                    const uint32_t dsym = T.get_d_sym(didx);
                    br.drop(dl);
                    if (dsym > 29) { status = INF_BAD_CODE; break; }
                    if (dsym < 4) new_dist = 1 + dsym;
                    else { const uint32_t e = (dsym >> 1) - 1; new_dist = ((2 + (dsym & 1)) << e) + 1 + br.bits(e); }
                }
            } else if (!copying) {
                // block header (rare, heavy): only with no copy in flight, so that it may write output itself
                const int r = block_header_call(br, T, LL, DD, in, in_len, out, o, out_cap, last, status, S);      // (out of line: round 4, inflate_coop.h)
                if (status != INF_OK) break;
                if (r) in_symbols = true;
                else if (last) { if (br.overran()) { status = INF_IN_OVERRUN; break; } done = true; }
            }
        }
        // ---- R: the trip's one memory wait: fold in the word prefetched a trip ago, prefetch the next ----------------------
        // The explicit vmcnt(0) tells the compiler's wait-count pass that no load is pending past this point on ANY path (it cannot
        // see that the predicates of a chunk's load and of its use are the same); without it the pass drains the counter at the
        // top of every trip -- which, the counter being shared, also waits for the stores issued a few instructions earlier.
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), expcnt/lgkmcnt untouched
#endif
        br.ensure(48);
        // ---- C: output -------------------------------------------------------------------------------------------------
        if (copying) {
#define RGX_PUT(J, V)                                                                                                            \
            if (n > 16u * (J)) {                                                                                                 \
                const uint32_t nb = n - 16u * (J) < 16u ? n - 16u * (J) : 16u;                                                   \
                S.put_chunk(o + 16u * (J), (uint64_t)(V)[0] | (uint64_t)(V)[1] << 32, (uint64_t)(V)[2] | (uint64_t)(V)[3] << 32, nb); \
            }
            RGX_PUT(0, v0) RGX_PUT(1, v1) RGX_PUT(2, v2) RGX_PUT(3, v3) RGX_PUT(4, v4) RGX_PUT(5, v5) RGX_PUT(6, v6) RGX_PUT(7, v7)
#undef RGX_PUT
            o += n; pend_len -= n;
            // an overlapping copy is periodic with period pend_dist, so 2 * pend_dist is as good a distance for the rest: a run
            // (distance 1) grows 1, 2, 4, ... bytes per trip and is at full 128-byte batches after eight
            if (n == pend_dist) pend_dist += pend_dist;
        }
        if (lit < 256) {
            if (o >= out_cap) { status = INF_OUT_OVERFLOW; break; }
            S.put_byte(o, lit); ++o;
            if (LITS > 1) {
                bool over = false;
#pragma unroll
                for (int q = 0; q < LITS - 1; ++q)
                    if (more[q] < 256 && !over) { if (o >= out_cap) over = true; else { S.put_byte(o, more[q]); ++o; } }
                if (over) { status = INF_OUT_OVERFLOW; break; }
            }
        } else if (new_len) {
            if (new_dist > o) { status = INF_BAD_DIST; break; }
            if (o + new_len > out_cap) { status = INF_OUT_OVERFLOW; break; }
            pend_len = new_len; pend_dist = new_dist;
        }
        if (done && pend_len == 0) break;
    }
    S.flush_partial(o);                                             // the tail chunk (also on errors: what was produced is in memory)
    if (in_used) *in_used = (uint32_t)(((uint64_t)(br.p - br.in) * 8 - br.cnt + 7) / 8);   // whole bytes of the stream that were consumed
    *out_len = o;
    return status;
}

}  // namespace rgx
