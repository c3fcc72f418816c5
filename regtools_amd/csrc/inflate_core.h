// inflate_core.h -- raw DEFLATE (RFC 1951) decoder for one BGZF member, written to run as ONE GPU LANE.
//
// Replaces, for the device path, bgzf.c:292-316 inflate_block (zlib inflate with windowBits -15) of
// /root/reference/src/utils/htslib.  The decoder is a straight-line per-lane state machine so that a
// 64-lane wavefront inflates 64 members at once; the only per-member tables that do not fit in
// registers (the canonical symbol lists and the code-length scratch) live behind the `Tab` accessor,
// which on the device is lane-interleaved LDS (bank = lane % 32, conflict-free) and on the host a plain
// array -- the same code is compiled by g++ for the CPU-side unit tests against zlib.
//
// Huffman decode is table-free in the hot loop: the 15 left-justified canonical upper bounds of each
// code live in registers; a symbol's length is 1 + #(bounds <= next-15-bits), found with 14 compares
// and no memory access, then ONE symbol-list lookup.  Length/distance bases are computed arithmetically.
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
struct HostTab {
    uint16_t ll_sym[288]; uint8_t d_sym[32]; uint16_t ll_base[16]; uint16_t d_base[16]; uint8_t lens[320];
    RGX_HD uint32_t get_ll_sym(uint32_t i) const { return ll_sym[i]; }
    RGX_HD void set_ll_sym(uint32_t i, uint32_t v) { ll_sym[i] = (uint16_t)v; }
    RGX_HD uint32_t get_d_sym(uint32_t i) const { return d_sym[i]; }
    RGX_HD void set_d_sym(uint32_t i, uint32_t v) { d_sym[i] = (uint8_t)v; }
    RGX_HD uint32_t get_ll_base(uint32_t l) const { return ll_base[l]; }
    RGX_HD void set_ll_base(uint32_t l, uint32_t v) { ll_base[l] = (uint16_t)v; }
    RGX_HD uint32_t get_d_base(uint32_t l) const { return d_base[l]; }
    RGX_HD void set_d_base(uint32_t l, uint32_t v) { d_base[l] = (uint16_t)v; }
    RGX_HD uint32_t get_len(uint32_t i) const { return lens[i]; }
    RGX_HD void set_len(uint32_t i, uint32_t v) { lens[i] = (uint8_t)v; }
};

// 15 canonical upper bounds (left-justified to 15 bits, exclusive, cumulative over lengths)
struct Bounds { uint32_t lim[16]; };

// LSB-first bit reader with a one-word look-ahead: the load for the NEXT 32 bits is issued when the current
// word is merged, so its latency hides behind the decode of the following symbols (a lane has no other
// wave to hide behind: 3 waves per CU).  Reads may run up to 8 bytes past the payload (the caller pads the
// buffer; in a BGZF file the footer and the next member follow anyway); consuming bits past the end is
// detected by overran().
struct BitReader {
    const uint8_t *p;      // address of the word held in `next`
    const uint8_t *in;     // payload start
    uint32_t in_len;
    uint64_t buf;          // LSB-first bit buffer
    uint32_t cnt;          // valid bits in buf
    uint32_t next;         // prefetched word at p
    RGX_HD void init(const uint8_t *i, uint32_t n) { in = i; in_len = n; p = i; buf = 0; cnt = 0; next = ld32(p); }
    // afterwards cnt >= 33: enough for a 15-bit code + 13 extra bits
    RGX_HD void refill() {
        if (cnt <= 32) {
            buf |= (uint64_t)next << cnt; cnt += 32; p += 4;
            next = ld32(p);
        }
    }
    RGX_HD uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    RGX_HD void drop(uint32_t n) { buf >>= n; cnt -= n; }
    RGX_HD uint32_t bits(uint32_t n) { uint32_t v = peek(n); drop(n); return v; }
    // bits consumed so far = 8 * (p - in) - cnt
    RGX_HD bool overran() const { return (uint64_t)(p - in) * 8 > (uint64_t)in_len * 8 + cnt; }
    // byte-aligned raw access for stored blocks: address of the next unconsumed byte once cnt == 0
    RGX_HD const uint8_t *byte_ptr() const { return p; }
    RGX_HD void restart_at(const uint8_t *q) { p = q; buf = 0; cnt = 0; next = ld32(p); }
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

// length (1..15) of the code at the head of v15 (MSB-first, left-justified); 16 = invalid
RGX_HD uint32_t code_len(const Bounds &b, uint32_t v15) {
    uint32_t l = 1;
#pragma unroll
    for (int k = 1; k <= 15; ++k) l += (v15 >= b.lim[k]) ? 1u : 0u;
    return l;
}

// Build canonical decode data from code lengths lens[off .. off+n) held in Tab.
// kind 0 = literal/length (symbol list via set_ll_sym / base via set_ll_base), 1 = distance.
// Returns INF_OK, INF_OVERSUBSCRIBED or INF_INCOMPLETE (zlib's rules: inftrees.c -- an incomplete set is
// only legal for a distance code with a single length-1 code or no codes at all).
template <class Tab>
RGX_HD int build_code(Tab &T, uint32_t off, uint32_t n, int kind, Bounds &B) {
    uint32_t count[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t l = T.get_len(off + i);
        // count[l]++ without dynamic register indexing
#pragma unroll
        for (int k = 0; k < 16; ++k) count[k] += (l == (uint32_t)k) ? 1u : 0u;
    }
    int left = 1;
    uint32_t code = 0, offs = 0;
    uint32_t offs_of[16];
    B.lim[0] = 0;
    offs_of[0] = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left <<= 1; left -= (int)count[l];
        uint32_t base = (offs - code) & 0xffff;     // symbol index = base + code (mod 2^16)
        if (kind == 0) T.set_ll_base((uint32_t)l, base); else T.set_d_base((uint32_t)l, base);
        offs_of[l] = offs;
        code += count[l]; offs += count[l];
        B.lim[l] = code << (15 - l);
        code <<= 1;
    }
    if (left < 0) return INF_OVERSUBSCRIBED;
    // zlib inftrees.c: an incomplete set is legal only when it is empty or a single 1-bit code
    if (left > 0 && !(offs == 0 || (offs == 1 && count[1] == 1))) return INF_INCOMPLETE;
    // symbol lists, in (length, symbol) order
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t l = T.get_len(off + i);
        if (l) {
            uint32_t slot = 0;
#pragma unroll
            for (int k = 1; k < 16; ++k) if (l == (uint32_t)k) { slot = offs_of[k]; offs_of[k] = slot + 1; }
            if (kind == 0) T.set_ll_sym(slot, i); else T.set_d_sym(slot, i);
        }
    }
    return INF_OK;
}

// Copy a match inside the output (out[o .. o+len) = out[o-dist ..]), byte-exact LZ77 semantics.
// The chain load -> store -> dependent load is what a lone lane spends its life on, so copies move 16 bytes per
// memory round trip and, when the distance allows, 64 bytes per round trip (four independent loads in flight).
// `slack` = bytes this lane may scribble past o+len inside its own member (they are rewritten by later symbols
// before anything reads them); with less than 16 bytes of slack the exact tail path is used.
RGX_HD void lz_copy(uint8_t *out, uint32_t o, uint32_t dist, uint32_t len, uint32_t slack) {
    uint8_t *d = out + o;
    const uint8_t *s = d - dist;
    if (dist < 16) {
        // grow the period to D = k*dist >= 16 by writing the first D bytes narrowly, then fall into the wide path
        uint32_t D = dist;
        while (D < 16) D += dist;
        uint32_t n0 = len < D ? len : D;
        if (dist >= 8) {
            uint32_t n = n0;
            while (n >= 8) { st64(d, ld64(s)); d += 8; s += 8; n -= 8; }
            while (n) { *d++ = *s++; --n; }
        } else {
            uint64_t pat = 0;
            for (uint32_t k = 0; k < dist; ++k) pat |= (uint64_t)s[k] << (8 * k);
            const uint32_t sh = 8 * (dist - 1);
            for (uint32_t n = n0; n; --n) { uint8_t b = (uint8_t)pat; *d++ = b; pat = (pat >> 8) | ((uint64_t)b << sh); }
        }
        len -= n0;
        if (!len) return;
        dist = D; s = d - dist;
    }
    if (dist >= 64) {
        while (len >= 64) {
            u32x4 a = ld128(s), b = ld128(s + 16), c = ld128(s + 32), e = ld128(s + 48);
            st128(d, a); st128(d + 16, b); st128(d + 32, c); st128(d + 48, e);
            d += 64; s += 64; len -= 64;
        }
    }
    if (slack >= 16) {
        // whole 16-byte chunks, overshooting by at most 15 bytes
        for (uint32_t n = 0; n < len; n += 16) { st128(d + n, ld128(s + n)); }
        return;
    }
    while (len >= 16) { st128(d, ld128(s)); d += 16; s += 16; len -= 16; }
    if (len) {
        // exact tail from two 8-byte loads (no dependent byte loop)
        if (len >= 8) { st64(d, ld64(s)); d += 8; s += 8; len -= 8; }
        if (len) {
            uint64_t v = 0;
            for (uint32_t k = 0; k < len; ++k) v |= (uint64_t)s[k] << (8 * k);
            if (len & 4) { st32(d, (uint32_t)v); d += 4; v >>= 32; }
            if (len & 2) { st16(d, (uint16_t)v); d += 2; v >>= 16; }
            if (len & 1) *d = (uint8_t)v;
        }
    }
}

// Inflate one raw-DEFLATE stream. Returns an InflateStatus; *out_len = bytes produced.
template <class Tab>
RGX_HD int inflate_raw(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len, Tab &T) {
    BitReader br; br.init(in, in_len);
    uint32_t o = 0;
    int status = INF_OK;
    uint32_t last = 0;
    while (!last && status == INF_OK) {
        br.refill();
        last = br.bits(1);
        uint32_t btype = br.bits(2);
        if (btype == 0) {
            // stored: skip to byte boundary, LEN, NLEN, raw bytes
            br.drop(br.cnt & 7);
            br.refill();
            uint32_t len = br.bits(16);
            br.refill();
            uint32_t nlen = br.bits(16);
            if ((len ^ 0xffff) != nlen) { status = INF_BAD_STORED; break; }
            if (o + len > out_cap) { status = INF_OUT_OVERFLOW; break; }
            // bytes still buffered come first
            while (len && br.cnt >= 8) { out[o++] = (uint8_t)br.bits(8); --len; }
            if (len) {
                const uint8_t *src = br.byte_ptr();   // cnt == 0 here: the prefetched word starts at the next payload byte
                if ((uint64_t)(src - in) + len > in_len) { status = INF_IN_OVERRUN; break; }
                for (uint32_t k = 0; k < len; ++k) out[o + k] = src[k];
                o += len; br.restart_at(src + len);
            }
            continue;
        }
        if (btype == 3) { status = INF_BAD_BTYPE; break; }

        Bounds LL, DD;
        if (btype == 1) {
            // fixed code (RFC 1951 3.2.6): 288 literal/length lengths, 30 distance codes of length 5
            for (uint32_t i = 0; i < 288; ++i) T.set_len(i, i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
            for (uint32_t i = 0; i < 32; ++i) T.set_len(288 + i, 5);   // 30,31 never legal: rejected at decode
            build_code(T, 0, 288, 0, LL);
            build_code(T, 288, 32, 1, DD);
        } else {
            br.refill();
            uint32_t hlit = br.bits(5) + 257, hdist = br.bits(5) + 1, hclen = br.bits(4) + 4;
            if (hlit > 286 || hdist > 30) { status = INF_BAD_HEADER; break; }
            // code-length code: 19 lengths of 3 bits, kept in a register (3 bits each)
            uint64_t cl_lens = 0;
            {
                const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                for (uint32_t i = 0; i < hclen; ++i) {
                    br.refill();
                    uint64_t l = br.bits(3);
                    cl_lens |= l << (3 * ord[i]);
                }
            }
            // canonical data for the CL code, entirely in registers: 7 bounds, 19 symbols of 5 bits in 2 regs
            uint32_t cl_count[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) cl_count[l] = 0;
            for (uint32_t s = 0; s < 19; ++s) {
                uint32_t l = (uint32_t)(cl_lens >> (3 * s)) & 7;
#pragma unroll
                for (int k = 0; k < 8; ++k) cl_count[k] += (l == (uint32_t)k) ? 1u : 0u;
            }
            uint32_t cl_lim[8], cl_base[8], cl_off[8];
            {
                int left = 1; uint32_t code = 0, offs = 0;
                cl_lim[0] = 0; cl_base[0] = 0; cl_off[0] = 0;
#pragma unroll
                for (int l = 1; l <= 7; ++l) {
                    left <<= 1; left -= (int)cl_count[l];
                    cl_base[l] = (offs - code) & 0xff; cl_off[l] = offs;
                    code += cl_count[l]; offs += cl_count[l];
                    cl_lim[l] = code << (7 - l);
                    code <<= 1;
                }
                if (left != 0) { status = left < 0 ? INF_OVERSUBSCRIBED : INF_INCOMPLETE; break; }
            }
            uint64_t cl_sym_lo = 0, cl_sym_hi = 0;  // symbol list, 5 bits per slot, slots 0..11 in lo, 12..18 in hi
            for (uint32_t s = 0; s < 19; ++s) {
                uint32_t l = (uint32_t)(cl_lens >> (3 * s)) & 7;
                if (l) {
                    uint32_t slot = 0;
#pragma unroll
                    for (int k = 1; k < 8; ++k) if (l == (uint32_t)k) { slot = cl_off[k]; cl_off[k] = slot + 1; }
                    if (slot < 12) cl_sym_lo |= (uint64_t)s << (5 * slot); else cl_sym_hi |= (uint64_t)s << (5 * (slot - 12));
                }
            }
            // read hlit + hdist code lengths
            uint32_t n = hlit + hdist, i = 0, prev = 0;
            while (i < n) {
                br.refill();
                uint32_t v7 = rev15(br.peek(7)) >> 8;   // 7 bits MSB-first
                uint32_t l = 1;
#pragma unroll
                for (int k = 1; k <= 7; ++k) l += (v7 >= cl_lim[k]) ? 1u : 0u;
                if (l > 7) { status = INF_BAD_CODE; break; }
                uint32_t base = 0;
#pragma unroll
                for (int k = 1; k <= 7; ++k) if (l == (uint32_t)k) base = cl_base[k];
                uint32_t slot = (base + (v7 >> (7 - l))) & 0xff;
                uint32_t sym = slot < 12 ? (uint32_t)(cl_sym_lo >> (5 * slot)) & 31 : (uint32_t)(cl_sym_hi >> (5 * (slot - 12))) & 31;
                br.drop(l);
                if (sym < 16) { T.set_len(i++, sym); prev = sym; }
                else {
                    uint32_t rep, val;
                    if (sym == 16) { if (i == 0) { status = INF_BAD_REPEAT; break; } val = prev; rep = 3 + br.bits(2); }
                    else if (sym == 17) { val = 0; rep = 3 + br.bits(3); }
                    else { val = 0; rep = 11 + br.bits(7); }
                    if (i + rep > n) { status = INF_BAD_REPEAT; break; }
                    for (uint32_t k = 0; k < rep; ++k) T.set_len(i++, val);
                    prev = val;
                }
            }
            if (status != INF_OK) break;
            if (T.get_len(256) == 0) { status = INF_NO_EOB; break; }
            // distance lengths sit right after the hlit literal/length lengths
            status = build_code(T, 0, hlit, 0, LL);
            if (status != INF_OK) break;
            status = build_code(T, hlit, hdist, 1, DD);
            if (status != INF_OK) break;
        }

        // ---- symbol loop -------------------------------------------------------------------------------
        for (;;) {
            br.refill();                                   // >= 32 bits: 15 (code) + 5 (extra) fit
            uint32_t v = rev15(br.peek(15));
            uint32_t l = code_len(LL, v);
            if (l > 15) { status = INF_BAD_CODE; break; }
            uint32_t idx = (T.get_ll_base(l) + (v >> (15 - l))) & 0xffff;
            if (idx >= 288) { status = INF_BAD_CODE; break; }
            uint32_t sym = T.get_ll_sym(idx);
            br.drop(l);
            if (sym < 256) {
                if (o >= out_cap) { status = INF_OUT_OVERFLOW; break; }
                out[o++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256) break;
            uint32_t c = sym - 257;
            if (c > 28) { status = INF_BAD_CODE; break; }
            uint32_t len;
            if (c < 8) len = 3 + c;
            else if (c == 28) len = 258;
            else { uint32_t e = (c >> 2) - 1; len = ((4 + (c & 3)) << e) + 3 + br.bits(e); }
            br.refill();                                   // 15 (code) + 13 (extra)
            uint32_t dv = rev15(br.peek(15));
            uint32_t dl = code_len(DD, dv);
            if (dl > 15) { status = INF_BAD_CODE; break; }
            uint32_t didx = (T.get_d_base(dl) + (dv >> (15 - dl))) & 0xffff;
            if (didx >= 32) { status = INF_BAD_CODE; break; }
            uint32_t dsym = T.get_d_sym(didx);
            br.drop(dl);
            if (dsym > 29) { status = INF_BAD_CODE; break; }
            uint32_t dist;
            if (dsym < 4) dist = 1 + dsym;
            else { uint32_t e = (dsym >> 1) - 1; dist = ((2 + (dsym & 1)) << e) + 1 + br.bits(e); }
            if (dist > o) { status = INF_BAD_DIST; break; }
            if (o + len > out_cap) { status = INF_OUT_OVERFLOW; break; }
            lz_copy(out, o, dist, len, out_cap - (o + len));
            o += len;
        }
        if (status == INF_OK && br.overran()) status = INF_IN_OVERRUN;
    }
    *out_len = o;
    return status;
}

}  // namespace rgx
