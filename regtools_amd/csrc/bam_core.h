// bam_core.h -- per-record / per-read device cores of the junctions-extract hot path (host-compilable for
// the CPU unit tests).  file:line citations are relative to /root/reference.
#pragma once
#include "common.h"

namespace rgx {

// ---- BAM record framing (src/utils/htslib/sam.c:399-433 bam_read1) ------------------------------------
struct RecHead {
    int32_t  block_len;   // bytes after the length word
    int32_t  tid, pos;
    uint32_t l_qname, n_cigar, flag;
    int32_t  l_qseq;
    int32_t  mtid;
    int64_t  aux_off;     // offset of the aux area from the start of the variable data
};

// Reads the fixed part at arena offset o (caller guarantees o+36 <= lim).
RGX_HD void rec_head(const uint8_t *a, RecHead &h) {
    h.block_len = (int32_t)ld32(a);
    h.tid = (int32_t)ld32(a + 4); h.pos = (int32_t)ld32(a + 8);
    uint32_t x2 = ld32(a + 12), x3 = ld32(a + 16);
    h.l_qname = x2 & 0xff; h.n_cigar = x3 & 0xffff; h.flag = x3 >> 16;
    h.l_qseq = (int32_t)ld32(a + 20);
    h.mtid = (int32_t)ld32(a + 24);
    h.aux_off = (int64_t)h.l_qname + 4 * (int64_t)h.n_cigar + (((int64_t)h.l_qseq + 1) >> 1) + h.l_qseq;
}

// bam_read1's own acceptance test (sam.c:421-423): anything else ends iteration silently.
RGX_HD bool rec_sane(const RecHead &h) {
    int64_t l_data = (int64_t)h.block_len - 32;
    if (l_data < 0 || h.l_qseq < 0 || h.l_qname < 1) return false;
    return h.aux_off <= l_data;
}

// Speculation filter used ONLY to guess where a segment's first record starts; every guess is verified
// against the exact chain afterwards, so this predicate affects speed, never results.
RGX_HD bool rec_plausible(const uint8_t *arena, uint64_t o, uint64_t lim, int32_t n_ref) {
    if (o + 36 > lim) return false;
    RecHead h; rec_head(arena + o, h);
    if (!rec_sane(h)) return false;
    if (h.block_len > (1 << 27)) return false;
    if (h.tid < -1 || h.tid >= n_ref || h.mtid < -1 || h.mtid >= n_ref) return false;
    if (h.pos < -1 || h.pos > (1 << 29)) return false;       // a .bai cannot index beyond 2^29 (hts.c:1517)
    uint64_t q_end = o + 36 + h.l_qname - 1;
    if (q_end < lim && arena[q_end] != 0) return false;     // qname is NUL-terminated
    return true;
}

// ---- strand rules -----------------------------------------------------------------------------------------
// src/junctions/junctions_extractor.cc:297-322 set_junction_strand_flag (strandness 1 = RF, 2 = FR; 3 acts as b=2)
RGX_HD char strand_from_flag(uint32_t flag, int strandness) {
    int rev = (flag >> 4) & 1, mrev = (flag >> 5) & 1, r1 = (flag >> 6) & 1, r2 = (flag >> 7) & 1;
    int nb = (strandness - 1) == 0 ? 1 : 0;        // !bool_strandness
    int f = nb ^ r1 ^ rev, s = nb ^ r2 ^ mrev;
    return f != s ? '?' : (f ? '+' : '-');
}

// junctions_extractor.cc:283-294 + sam.c:1254-1266 bam_aux_get / :1233-1252 skip_aux / :1301-1307 bam_aux2A.
// First tag equal to `tag`: type 'A' with a non-NUL value gives that char, anything else '?'.
// *unknown (optional) is set when a tag of a type skip_aux does not know stands in front of the one looked for: upstream abort()s there (sam.c:1248).
RGX_HD char strand_from_tag(const uint8_t *aux, const uint8_t *end, uint8_t t0, uint8_t t1, bool *unknown = nullptr) {
    const uint8_t *s = aux;
    while (s + 3 <= end) {
        bool hit = s[0] == t0 && s[1] == t1;
        s += 2;
        if (hit) return (s[0] == 'A' && s + 2 <= end && s[1] != 0) ? (char)s[1] : '?';
        uint8_t t = *s++;
        uint32_t sz;
        switch (t) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'd': sz = 8; break;
            case 'Z': case 'H': { while (s < end && *s) ++s; sz = 1; break; }
            case 'B': {
                if (s + 5 > end) return '?';
                uint8_t st = *s++; uint32_t n = (uint32_t)s[0] | (uint32_t)s[1] << 8 | (uint32_t)s[2] << 16 | (uint32_t)s[3] << 24; s += 4;   // bytes: s may point into LDS
                uint32_t es = (st == 'c' || st == 'C' || st == 'A') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : (st == 'd') ? 8 : 0;
                if ((uint64_t)es * n > (uint64_t)(end - s)) return '?';
                sz = es * n; break;
            }
            default: if (unknown) *unknown = true; return '?';   // upstream abort()s on an unknown type (the caller is told; the strand reads "not found")
        }
        s += sz;
    }
    return '?';
}

// junctions_extractor.cc:362-374 set_junction_barcode: bam_aux_get(tag) (sam.c:1254-1266) + bam_aux2Z (sam.c:1309-1315).
// 1 = found, the value is aux[*val, *val + *len) (its NUL excluded); 0 = no such tag (the barcode is then "?"); -1 = the tag is there
// but is not a Z/H string (upstream builds a std::string from NULL there and dies).
RGX_HD int aux_find_string(const uint8_t *aux, const uint8_t *end, uint8_t t0, uint8_t t1, uint32_t *val, uint32_t *len) {
    const uint8_t *s = aux;
    while (s + 3 <= end) {
        const bool hit = s[0] == t0 && s[1] == t1;
        s += 2;
        const uint8_t t = *s++;
        if (hit) {
            if (t != 'Z' && t != 'H') return -1;
            const uint8_t *e = s;
            while (e < end && *e) ++e;
            *val = (uint32_t)(s - aux); *len = (uint32_t)(e - s);
            return 1;
        }
        uint32_t sz;
        switch (t) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'd': sz = 8; break;
            case 'Z': case 'H': { while (s < end && *s) ++s; sz = 1; break; }
            case 'B': {
                if (s + 5 > end) return 0;
                uint8_t st = *s++; uint32_t n = (uint32_t)s[0] | (uint32_t)s[1] << 8 | (uint32_t)s[2] << 16 | (uint32_t)s[3] << 24; s += 4;
                uint32_t es = (st == 'c' || st == 'C' || st == 'A') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : (st == 'd') ? 8 : 0;
                if ((uint64_t)es * n > (uint64_t)(end - s)) return 0;
                sz = es * n; break;
            }
            default: return 0;
        }
        s += sz;
    }
    return 0;
}
// grouping key of a barcode string: 64 bits, two independently seeded 32-bit lanes; equal strings are CHECKED byte for byte afterwards
// (k_bc_heads), the hash only brings them together
RGX_HD void barcode_hash(const uint8_t *s, uint32_t len, uint32_t *lo, uint32_t *hi) {
    uint32_t a = 0x811c9dc5u ^ len, b = 0x9747b28cu + len * 0x85ebca6bu;
    for (uint32_t i = 0; i < len; ++i) {
        a = (a ^ s[i]) * 0x01000193u;
        b = (b + s[i]) * 0xcc9e2d51u; b = (b << 15 | b >> 17) * 0x1b873593u;
    }
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15;
    b ^= b >> 13; b *= 0xc2b2ae35u; b ^= b >> 16;
    *lo = a; *hi = b;
}

// ---- intron-motif strand rule (junctions_extractor.cc:325-342, :564-584; faidx.c:341-413) --------------------------------
// FASTA file bytes live in HBM; one descriptor per BAM contig (matched by name on the host).
struct FaContig { int64_t offset, len; int32_t line_blen, line_len; int32_t present; int32_t pad; };

// fai_fetch("chr:beg1-end1") for a two-base window: returns the number of bases (0..2) and the bases in b[0..1]
RGX_HD int fa_fetch2(const uint8_t *fa, const FaContig &c, uint32_t beg1, uint32_t end1, uint8_t *b) {
    int64_t beg = beg1, end = end1;
    if (beg > 0) --beg;
    if (beg >= c.len) beg = c.len;
    if (end >= c.len) end = c.len;
    if (beg > end) beg = end;
    int n = 0;
    if (c.line_blen <= 0) return 0;
    for (int64_t p = beg; p < end && n < 2; ++p) b[n++] = fa[c.offset + p / c.line_blen * c.line_len + p % c.line_blen];
    return n;
}
RGX_HD uint8_t base_comp(uint8_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }   // utils/common.h:59-83

// strand of the junction [start,end) from its splice-site motif; `carried` is the strand of the previous junction of the same
// read (upstream reuses the Junction object, cc:576-579).  '?' when the motif is not canonical.
RGX_HD char strand_from_motif(const uint8_t *fa, const FaContig &c, uint32_t start, uint32_t end, char carried) {
    uint8_t s1[2], s2[2];
    const int l1 = fa_fetch2(fa, c, start + 1u, start + 2u, s1);
    const int l2 = fa_fetch2(fa, c, end + 1u - 2u, end + 1u - 1u, s2);
    if (l1 != 2 || l2 != 2) return '?';
    uint8_t d0, d1, a0, a1;                       // motif = d0 d1 '-' a0 a1
    if (carried == '-') { d0 = base_comp(s2[1]); d1 = base_comp(s2[0]); a0 = base_comp(s1[1]); a1 = base_comp(s1[0]); }
    else { d0 = s1[0]; d1 = s1[1]; a0 = s2[0]; a1 = s2[1]; }
    const uint32_t m = (uint32_t)d0 << 24 | (uint32_t)d1 << 16 | (uint32_t)a0 << 8 | a1;
    if (m == 0x47544147u /*GT-AG*/ || m == 0x47434147u /*GC-AG*/ || m == 0x41544143u /*AT-AC*/) return '+';
    if (m == 0x43544143u /*CT-AC*/ || m == 0x43544743u /*CT-GC*/ || m == 0x47544154u /*GT-AT*/) return '-';
    return '?';
}

// key class of a strand char (junctions_extractor.cc:186-193)
RGX_HD uint32_t strand_class(char c) { return c == '+' ? 0u : c == '-' ? 1u : 2u; }

// ---- CIGAR (htslib/sam.h:75-104) ----------------------------------------------------------------------------
// op classes: N; M,= extend an anchor; D,X break it and advance the reference; I,S break it; H,P,B,10-15 are inert.
RGX_HD uint32_t cig_ref_len(uint32_t c) {          // bam_cigar_type bit 1 (0x3C1A7)
    uint32_t op = c & 0xf;
    return ((0x3C1A7u >> (op << 1)) & 2u) ? (c >> 4) : 0u;
}
RGX_HD bool cig_is_N(uint32_t c) { return (c & 0xf) == 3; }
RGX_HD bool cig_is_breaker(uint32_t c) { uint32_t op = c & 0xf; return op == 3 || op == 2 || op == 8 || op == 1 || op == 4; }
RGX_HD bool cig_advances_junction_state(uint32_t c) {   // ref-advancing for the junction machine: M,=,D,X,N
    uint32_t op = c & 0xf; return op == 0 || op == 7 || op == 2 || op == 8 || op == 3;
}

// bam_endpos (sam.c:336-342)
RGX_HD int32_t rec_endpos(const uint8_t *cig, uint32_t n_cigar, uint32_t flag, int32_t pos) {
    if (!(flag & 4) && n_cigar > 0) {
        int32_t l = 0;
        for (uint32_t k = 0; k < n_cigar; ++k) l += (int32_t)cig_ref_len(ld32(cig + 4 * (size_t)k));
        return pos + l;
    }
    return pos + 1;
}

// The serial state machine of junctions_extractor.cc:377-497.  `emit(start,end,ts,te)` is
// called once per N op, in order.  Used by the lane-per-read kernel and by the unit tests; the
// wave-per-read kernel computes the same four numbers with prefix sums (see cigar_scan in kernels.hip).
template <class Emit>
RGX_HD void cigar_walk(int32_t pos, const uint8_t *cig, uint32_t n_cigar, Emit &&emit) {
    uint32_t start = (uint32_t)pos, ts = (uint32_t)pos, end = 0, te = 0;
    bool started = false;
    for (uint32_t i = 0; i < n_cigar; ++i) {
        uint32_t c = ld32(cig + 4 * (size_t)i), op = c & 0xf, len = c >> 4;
        if (op == 3) {
            if (!started) { end = start + len; te = end; started = true; }
            else { emit(start, end, ts, te); ts = end; start = te; end = start + len; te = end; }
        } else if (op == 0 || op == 7) {
            if (!started) start += len; else te += len;
        } else if (op == 2 || op == 8) {
            if (!started) { start += len; ts = start; }
            else { emit(start, end, ts, te); start = te + len; ts = start; }
            started = false;
        } else if (op == 1 || op == 4) {
            if (!started) ts = start;
            else { emit(start, end, ts, te); start = te; ts = start; }
            started = false;
        }
    }
    if (started) emit(start, end, ts, te);
}

// junction_qc (junctions_extractor.cc:160-170), the intron-length half: unsigned compare
RGX_HD bool intron_ok(uint32_t start, uint32_t end, uint32_t min_intron, uint32_t max_intron) {
    uint32_t l = end - start;
    return !(l < min_intron || l > max_intron);
}

}  // namespace rgx
