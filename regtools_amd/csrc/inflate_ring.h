// inflate_ring.h -- raw DEFLATE (RFC 1951) for one BGZF member per GPU LANE, output through a per-lane window in LDS.
//
// Replaces, for the device path, bgzf.c:292-316 inflate_block (zlib inflate, windowBits -15) of /root/reference/src/utils/htslib.
// Same per-lane Huffman machinery as inflate_core.h (register-resident canonical codes, one symbol per trip); what changed is
// where the bytes go.  Round 1's decoder wrote every member straight to HBM, one lane per member: 64 lanes = 64 different
// cache lines per store and per LZ77 source load, 4.4x the algorithmic HBM traffic and a memory round trip in every trip of the
// symbol loop.  Here every lane owns a RING of the last 384 output bytes in LDS:
//   * literals and matches are written to the ring (dword-interleaved across the lanes: lane L only ever touches bank L % 32);
//   * a match whose distance is <= kNearMax is copied ring -> ring (LDS latency instead of an HBM/L2 round trip, no global request);
//   * farther matches and stored blocks read their source from global memory (it has been flushed by then) into the ring;
//   * the ring drains to HBM in whole, aligned 128-byte lines written COOPERATIVELY by the wave (8 lanes per line, 8 lines per
//     store instruction): HBM sees every output byte exactly once, as a full line.
// The per-lane part (this file) is plain C++ over three small accessors -- Tab (Huffman symbol lists), Ring (the window) and
// Coop (the wave) -- so that tests/hostemu can run it on the host against zlib with a one-lane "wave".
#pragma once
#include "inflate_core.h"

namespace rgx {

constexpr uint32_t kRingBytes = 384, kRingDw = kRingBytes / 4, kRingChunks = kRingBytes / 16;
constexpr uint32_t kRingMirrorDw = 4;                         // ring dwords 96..99 repeat dwords 0..3 (source reads run up to 4 dwords past a chunk start)
constexpr uint32_t kRingLaneDw = kRingDw + kRingMirrorDw;
constexpr uint32_t kRingSlack = 16;                           // a chunk store may leave up to 15 bytes of garbage AHEAD of the head
constexpr uint32_t kRingFill = kRingBytes - kRingSlack;       // head - flushed never exceeds this
constexpr uint32_t kNearMax = kRingBytes - 2 * kRingSlack;    // 352: match distances served from the ring
constexpr uint32_t kRingBatch = 128;                          // bytes copied per trip (8 chunks of 16, cut on the destination)
constexpr uint32_t kFlushAt = 256;                            // a lane holding this many unflushed bytes asks the wave for a flush round
constexpr uint32_t kArenaFrontPad = 128;                      // far copies read up to 15 bytes in front of their source: the arena starts this far into its allocation

RGX_HD uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t r) {       // bytes r..r+3 of the 8-byte pair (hi:lo), r in 0..3
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, r);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * r));
#endif
}

// ---- host-side ring (unit tests): the lane's 100 dwords as a plain array -------------------------------------------------------
struct HostRing {
    uint32_t w[kRingLaneDw];
    RGX_HD uint32_t rd(uint32_t dw) const { return w[dw]; }
    RGX_HD void rd4(uint32_t dw, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d) const { a = w[dw]; b = w[dw + 1]; c = w[dw + 2]; d = w[dw + 3]; }
    RGX_HD void wr4(uint32_t dw, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { w[dw] = a; w[dw + 1] = b; w[dw + 2] = c; w[dw + 3] = d; }
    RGX_HD void wr8(uint32_t byte_pos, uint32_t b) { w[byte_pos >> 2] = (w[byte_pos >> 2] & ~(0xffu << (8 * (byte_pos & 3)))) | b << (8 * (byte_pos & 3)); }
};

// Where a lane's member goes.  Positions are counted in "P space": P = (destination address & 127) + bytes produced, so that
// P % 128 == 0 is a line boundary of the destination and lbase + P is the destination address; ring position = P % 384.
struct RingOut {
    uint8_t *lbase;       // destination address of P == 0 (128-byte aligned; up to 127 bytes in front of the member)
    uint32_t hd, fl;      // head (next byte to produce) and flushed-up-to, in P space; fl <= hd <= fl + kRingFill
    uint32_t hp;          // hd % kRingBytes
    RGX_HD void init(uint8_t *out) {
        const uint32_t p0 = (uint32_t)((uintptr_t)out & 127u);
        lbase = out - p0; hd = p0; fl = p0; hp = p0;
    }
    // this lane alone writes [lo, hi) of its member to memory (the partial lines at the two ends of a member; < 128 bytes)
    template <class Ring>
    RGX_HD void slow_flush(const Ring &R, uint32_t lo, uint32_t hi) const {
        uint32_t rp = lo % kRingBytes;
        for (uint32_t p = lo; p < hi;) {
            const uint32_t w = R.rd(rp >> 2);
            if ((p & 3u) == 0 && p + 4 <= hi) { *(uint32_t *)(lbase + p) = w; p += 4; rp += 4; }
            else { lbase[p] = (uint8_t)(w >> (8 * (p & 3u))); ++p; ++rp; }
            if (rp >= kRingBytes) rp -= kRingBytes;
        }
    }
};

// one-lane "wave" of the host build: every complete line goes out at once
struct HostCoop {
    RGX_HD bool any(bool b) const { return b; }
    template <class Ring>
    RGX_HD void flush_lines(const Ring &R, RingOut &O) const {
        if (O.fl & 127u) {
            const uint32_t c = (O.fl + 127u) & ~127u;
            if (O.hd < c) return;
            O.slow_flush(R, O.fl, c); O.fl = c;
        }
        while (O.fl + 128 <= O.hd) {
            const uint32_t d0 = (O.fl % kRingBytes) >> 2;
            for (uint32_t i = 0; i < 32; ++i) *(uint32_t *)(O.lbase + O.fl + 4 * i) = R.rd(d0 + i);
            O.fl += 128;
        }
    }
};

// Inflate one raw-DEFLATE stream into `out` (capacity out_cap).  Returns an InflateStatus; *out_len = bytes produced (all of them are
// in memory on return, also after an error).  `active` = false: the lane has no member and only takes part in the wave's flush rounds.
// Every lane of the wave must call this together (Coop::any / Coop::flush_lines are wave-wide).
//
// One trip of the loop =  A. start the loads of the pending copy (ring reads, or global loads for far / stored sources)
//                         B. decode the next symbol from the bit buffer while they are in flight (only if the copy ends this trip)
//                         R. the trip's one wait on global memory; top the bit buffer up, prefetch the next word
//                         C. write the copied bytes and the literal to the ring, or arm the next copy
//                         F. (wave-wide, only when some lane asks) flush the complete lines of all 64 rings
template <class Tab, class Ring, class Coop>
RGX_HD int inflate_ring(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len, Tab &T, Ring &R, Coop &C, bool active,
                        uint32_t *in_used = nullptr) {
    BitReader br;
    br.p = in; br.in = in; br.in_len = in_len; br.buf = 0; br.cnt = 0; br.next = 0;
    if (active) br.init(in, in_len);
    RingOut O; O.init(out);
    uint32_t o = 0;
    int status = INF_OK;
    uint32_t last = 0;
    bool in_symbols = false, done = false, stored = false;
    bool fin = !active, tail_done = !active;
    uint32_t pend_len = 0, pend_dist = 0;
    const uint8_t *stored_src = in;
    Code LL, DD;
#pragma unroll
    for (int k = 0; k < 16; ++k) { LL.c[k] = 0; DD.c[k] = 0; }
    uint32_t S[37];                                   // source dwords of the copy in flight: loaded in A, consumed in C under the same predicates
#pragma unroll
    for (int k = 0; k < 37; ++k) S[k] = 0;

    for (;;) {
        // ---- the lane's half-trip in front of the wait: A (loads of the pending copy) and B (next symbol) -----------------------------
        const uint32_t held = O.hd - O.fl;
        bool want_flush = !fin && held >= kFlushAt;
        bool live = !fin && held + 2 <= kRingFill;                    // (no room even for a literal: wait for the flush round)
        const uint32_t ph = O.hp & 15u;
        uint32_t n = 0, r = 0;
        uint32_t lit = 256, new_len = 0, new_dist = 0;
        uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;                       // chunk 0 of the destination as it is (its first ph bytes stay)
        if (live) do {
            const bool copying = pend_len != 0;
            if (copying) {
                n = pend_len < kRingBatch - ph ? pend_len : kRingBatch - ph;       // chunks 0..7 of the destination
                const uint32_t room = kRingFill - 1 - held;                        // (keeps one byte for this trip's literal)
                if (room < n) n = room;
                bool near = false;
                if (!stored) {
                    if (pend_dist < n) n = pend_dist;                              // only bytes that exist already (any distance >= 1)
                    near = pend_dist <= kNearMax;
                    if (!near) {                                                   // far: the source must have reached memory
                        const uint32_t src = O.hd - pend_dist;
                        const uint32_t avail = O.fl > src ? O.fl - src : 0;
                        if (avail < n) n = avail;
                    }
                }
                if (n == 0) { want_flush = true; live = false; break; }
                R.rd4((O.hp >> 4) * 4u, e0, e1, e2, e3);
                if (near) {
                    // dest dword D (chunk-aligned numbering) <- bytes 4D - dist .. +3 = align_bytes(S[j+1], S[j], r) of source dwords q+j
                    const uint32_t dq = (pend_dist + 3u) >> 2;
                    r = (4u - (pend_dist & 3u)) & 3u;
                    uint32_t q = (O.hp >> 4) * 4u + kRingDw - dq;
                    if (q >= kRingDw) q -= kRingDw;
                    S[0] = R.rd(q);
#define RGX_NEAR(Cn)                                                                                                             \
                    if (16u * (Cn) < ph + n) {                                                                                       \
                        uint32_t qc = q + 4u * (Cn); if (qc >= kRingDw) qc -= kRingDw;                                               \
                        R.rd4(qc + 1, S[4 * (Cn) + 1], S[4 * (Cn) + 2], S[4 * (Cn) + 3], S[4 * (Cn) + 4]);                           \
                    }
                    RGX_NEAR(0) RGX_NEAR(1) RGX_NEAR(2) RGX_NEAR(3) RGX_NEAR(4) RGX_NEAR(5) RGX_NEAR(6) RGX_NEAR(7)
#undef RGX_NEAR
                } else {
                    // the source stream is loaded from `ph` bytes in front of its first byte, so that dword j of the load IS destination dword j
                    const uint8_t *s = (stored ? stored_src : O.lbase + (O.hd - pend_dist)) - ph;
#define RGX_FAR(Cn)                                                                                                              \
                    if (16u * (Cn) < ph + n) {                                                                                       \
                        const u32x4 v = ld128(s + 16 * (Cn));                                                                        \
                        S[4 * (Cn)] = v[0]; S[4 * (Cn) + 1] = v[1]; S[4 * (Cn) + 2] = v[2]; S[4 * (Cn) + 3] = v[3];                  \
                    }
                    RGX_FAR(0) RGX_FAR(1) RGX_FAR(2) RGX_FAR(3) RGX_FAR(4) RGX_FAR(5) RGX_FAR(6) RGX_FAR(7)
#undef RGX_FAR
                }
            }
            // ---- B: next symbol (only when the copy, if any, ends with this batch) --------------------------------------------
            if (pend_len == n && !done) {
                if (in_symbols) {
                    const uint32_t v = rev15(br.peek(15));
                    uint32_t l;
                    const uint32_t idx = code_lookup(LL, v, l);
                    if (l == 0 || idx >= 288) { status = INF_BAD_CODE; fin = true; break; }
                    const uint32_t sym = T.get_ll_sym(idx);
                    br.drop(l);
                    if (sym < 256) lit = sym;
                    else if (sym == 256) {
                        in_symbols = false;
                        if (br.overran()) { status = INF_IN_OVERRUN; fin = true; break; }
                        if (last) done = true;
                    } else {
                        const uint32_t c = sym - 257;
                        if (c > 28) { status = INF_BAD_CODE; fin = true; break; }
                        if (c < 8) new_len = 3 + c;
                        else if (c == 28) new_len = 258;
                        else { const uint32_t e = (c >> 2) - 1; new_len = ((4 + (c & 3)) << e) + 3 + br.bits(e); }
                        const uint32_t dv = rev15(br.peek(15));
                        uint32_t dl;
                        const uint32_t didx = code_lookup(DD, dv, dl);
                        if (dl == 0 || didx >= 32) { status = INF_BAD_CODE; fin = true; break; }
                        const uint32_t dsym = T.get_d_sym(didx);
                        br.drop(dl);
                        if (dsym > 29) { status = INF_BAD_CODE; fin = true; break; }
                        if (dsym < 4) new_dist = 1 + dsym;
                        else { const uint32_t e = (dsym >> 1) - 1; new_dist = ((2 + (dsym & 1)) << e) + 1 + br.bits(e); }
                    }
                } else if (!copying) {
                    // block header (rare, heavy): only with no copy in flight
                    if (br.overran()) { status = INF_IN_OVERRUN; fin = true; break; }
                    br.ensure(32);
                    last = br.bits(1);
                    const uint32_t btype = br.bits(2);
                    if (btype == 0) {
                        // stored: LEN, NLEN at the next byte boundary, then raw bytes -- copied like a match whose source is the input
                        br.drop(br.cnt & 7);
                        br.ensure(32);
                        const uint32_t len = br.bits(16), nlen = br.bits(16);
                        if ((len ^ 0xffff) != nlen) { status = INF_BAD_STORED; fin = true; break; }
                        if (o + len > out_cap) { status = INF_OUT_OVERFLOW; fin = true; break; }
                        const uint8_t *src = br.p - (br.cnt >> 3);                  // first raw byte (whole bytes still in the bit buffer included)
                        if ((uint64_t)(src - in) + len > in_len) { status = INF_IN_OVERRUN; fin = true; break; }
                        br.restart_at(src + len);
                        if (len) { stored = true; stored_src = src; pend_len = len; pend_dist = 0; }
                        else if (last) { if (br.overran()) { status = INF_IN_OVERRUN; fin = true; break; } done = true; }
                    } else if (btype == 3) { status = INF_BAD_BTYPE; fin = true; break; }
                    else {
                        if (!build_block_codes(br, T, LL, DD, btype, status)) { fin = true; break; }
                        in_symbols = true;
                    }
                }
            }
        } while (0);
        if (fin) live = false;
        // ---- R: the trip's one wait on global memory (every lane: the explicit vmcnt(0) keeps the compiler's wait-count pass from draining
        //         the counter at the top of every trip, inflate_core.h) ... and F right behind it: the stores of a flush round then have a whole
        //         trip to be acknowledged before the next wait sees them ------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
        if (fin && !tail_done && O.hd - O.fl >= 128) want_flush = true;         // whole lines still go out cooperatively
        if (C.any(want_flush)) C.flush_lines(R, O);
        if (fin && !tail_done) { O.slow_flush(R, O.fl, O.hd); O.fl = O.hd; tail_done = true; }
        // ---- C: output ---------------------------------------------------------------------------------------------------------------
        if (live) {
            br.ensure(48);
            if (n) {
                uint32_t ci = O.hp >> 4;                                       // ring chunk of the head
                {   // chunk 0: its first ph bytes are older output and stay
                    uint32_t v0 = align_bytes(S[1], S[0], r), v1 = align_bytes(S[2], S[1], r), v2 = align_bytes(S[3], S[2], r), v3 = align_bytes(S[4], S[3], r);
#define RGX_KEEP(J, E, V)                                                                                                        \
                    {                                                                                                                \
                        const uint32_t kb = ph > 4u * (J) ? ph - 4u * (J) : 0u;                                                      \
                        const uint32_t m = kb >= 4u ? 0xffffffffu : (1u << (8u * kb)) - 1u;                                          \
                        V = (E & m) | (V & ~m);                                                                                      \
                    }
                    RGX_KEEP(0, e0, v0) RGX_KEEP(1, e1, v1) RGX_KEEP(2, e2, v2) RGX_KEEP(3, e3, v3)
#undef RGX_KEEP
                    R.wr4(ci * 4u, v0, v1, v2, v3);
                    if (ci == 0) R.wr4(kRingDw, v0, v1, v2, v3);                  // (the mirror of ring dwords 0..3)
                }
#define RGX_PUT(Cn)                                                                                                              \
                if (16u * (Cn) < ph + n) {                                                                                           \
                    ++ci; if (ci >= kRingChunks) ci -= kRingChunks;                                                                  \
                    const uint32_t w0 = align_bytes(S[4 * (Cn) + 1], S[4 * (Cn)], r), w1 = align_bytes(S[4 * (Cn) + 2], S[4 * (Cn) + 1], r),     \
                                   w2 = align_bytes(S[4 * (Cn) + 3], S[4 * (Cn) + 2], r), w3 = align_bytes(S[4 * (Cn) + 4], S[4 * (Cn) + 3], r); \
                    R.wr4(ci * 4u, w0, w1, w2, w3);                                                                                  \
                    if (ci == 0) R.wr4(kRingDw, w0, w1, w2, w3);                                                                     \
                }
                RGX_PUT(1) RGX_PUT(2) RGX_PUT(3) RGX_PUT(4) RGX_PUT(5) RGX_PUT(6) RGX_PUT(7)
#undef RGX_PUT
                o += n; O.hd += n; O.hp += n; if (O.hp >= kRingBytes) O.hp -= kRingBytes;
                pend_len -= n;
                if (stored) {
                    stored_src += n;
                    if (pend_len == 0) { stored = false; if (last) { if (br.overran()) { status = INF_IN_OVERRUN; fin = true; } else done = true; } }
                } else if (n == pend_dist) pend_dist += pend_dist;        // an overlapping copy is periodic: 2 * dist is as good a distance for the rest
            }
            if (lit < 256) {
                if (o >= out_cap) { status = INF_OUT_OVERFLOW; fin = true; }
                else {
                    R.wr8(O.hp, lit);
                    if (O.hp < 16) R.wr8(O.hp + kRingBytes, lit);
                    ++o; ++O.hd; ++O.hp; if (O.hp >= kRingBytes) O.hp = 0;
                }
            } else if (new_len) {
                if (new_dist > o) { status = INF_BAD_DIST; fin = true; }
                else if (o + new_len > out_cap) { status = INF_OUT_OVERFLOW; fin = true; }
                else { pend_len = new_len; pend_dist = new_dist; }
            }
            if (done && pend_len == 0) fin = true;
        }
        if (!C.any(!tail_done)) break;
    }
    if (in_used) *in_used = (uint32_t)(((uint64_t)(br.p - br.in) * 8 - br.cnt + 7) / 8);
    *out_len = o;
    return status;
}

}  // namespace rgx
