// synth_bam.cpp -- deterministic synthetic BAM + BAI writer (tooling, NOT the hot path). See synth_bam.h.
//
// Everything about a read is a pure function of (seed, read id), so generation is embarrassingly
// parallel: (1) every read's (tid,pos) is computed and the reads are sorted by coordinate with a
// bucketed parallel sort, (2) record sizes are prefix-summed into stream offsets, (3) each thread owns a
// contiguous range of 0xff00-byte BGZF members, regenerates exactly the records that intersect it,
// deflates them, and indexes the records that start inside it, (4) the per-thread index fragments are
// stitched into a BAI (bins + chunks, 16 KiB linear index, pseudo-bin 37450, n_no_coor) as the SAM
// spec section 5 and /root/reference/src/utils/htslib/hts.c:1517-1567 (the loader) define it.
#include "synth_bam.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t kBlock = 0xff00;  // inflated bytes per BGZF member (htslib/bgzf.h:41)

inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
struct Rng {  // counter-based: stream (seed, id), draw k
    uint64_t s; uint64_t k = 0;
    Rng(uint64_t seed, uint64_t stream, uint64_t id) : s(mix64(seed ^ mix64(stream * 0x51ED2701u + 17)) ^ mix64(id)) {}
    uint64_t next() { return mix64(s + (++k) * 0xD1342543DE82EF95ull); }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct Contig { std::string name; uint32_t len; };

const Contig kHuman[] = {
    {"chr1", 248956422}, {"chr2", 242193529}, {"chr3", 198295559}, {"chr4", 190214555}, {"chr5", 181538259},
    {"chr6", 170805979}, {"chr7", 159345973}, {"chr8", 145138636}, {"chr9", 138394717}, {"chr10", 133797422},
    {"chr11", 135086622}, {"chr12", 133275309}, {"chr13", 114364328}, {"chr14", 107043718}, {"chr15", 101991189},
    {"chr16", 90338345}, {"chr17", 83257441}, {"chr18", 80373285}, {"chr19", 58617616}, {"chr20", 64444167},
    {"chr21", 46709983}, {"chr22", 50818468}, {"chrX", 156040895}};
const Contig kFuzz[] = {{"1", 600000}, {"10", 300000}, {"2", 500000}, {"MT", 16569}};

struct Intron { int32_t tid; uint32_t donor, len; char strand; };

// gene model shared by the BAM's intron table and rgx_synth_annotation
struct Gene { int32_t tid; char strand; std::vector<uint32_t> es, ee; /* 1-based inclusive exons */ std::vector<std::vector<uint16_t>> tx; };

constexpr uint32_t kLocusExons = 26;
struct Locus { int32_t tid; char strand; uint32_t start[kLocusExons], len[kLocusExons]; };   // LONG shape, see make_long

struct Ctx {
    rgx_synth_params p;
    std::vector<Gene> genes;
    std::vector<Locus> loci;
    std::vector<Contig> contigs;
    std::vector<uint64_t> contig_off;  // linear genome offsets
    uint64_t genome_len = 0;
    std::vector<Intron> introns;
    std::vector<uint64_t> intron_cum;  // cumulative spliced-read counts per intron (SHORT shape)
    uint64_t n_spliced = 0;
};

// one fully specified read
struct Read {
    int32_t tid = -1, pos = -1;
    uint16_t flag = 0;
    uint8_t mapq = 60;
    uint32_t l_qseq = 0;
    std::vector<uint32_t> cigar;
    uint8_t aux[64]; uint32_t l_aux = 0;
};

inline uint32_t cig(uint32_t len, uint32_t op) { return len << 4 | op; }
enum { M = 0, I = 1, D = 2, N = 3, S = 4, H = 5, P = 6, EQ = 7, X = 8 };

void pick_locus(const Ctx &c, Rng &r, uint32_t span, int32_t &tid, uint32_t &pos) {
    // length-weighted contig, then uniform position that leaves `span` bases of room
    for (int tries = 0; tries < 64; ++tries) {
        double uu = r.unit();
        if (c.p.n_slices > 1) uu = ((double)c.p.slice_index + uu) / (double)c.p.n_slices;   // coordinate window of this slice
        uint64_t g = (uint64_t)(uu * (double)c.genome_len);
        size_t t = std::upper_bound(c.contig_off.begin(), c.contig_off.end(), g) - c.contig_off.begin() - 1;
        if (t >= c.contigs.size()) t = c.contigs.size() - 1;
        uint32_t len = c.contigs[t].len;
        if (len <= span + 2) continue;
        tid = (int32_t)t; pos = 1 + (uint32_t)(r.unit() * (double)(len - span - 2));
        return;
    }
    tid = 0; pos = 1;
}

void put_aux(Read &rd, const char *tag, char type, uint8_t v) {
    rd.aux[rd.l_aux++] = (uint8_t)tag[0]; rd.aux[rd.l_aux++] = (uint8_t)tag[1];
    rd.aux[rd.l_aux++] = (uint8_t)type; rd.aux[rd.l_aux++] = v;
}

void build_gene(const Ctx &c, uint64_t g, Gene &G) {
    Rng r(c.p.seed, 7, g);
    G.strand = (r.next() & 1) ? '+' : '-';
    const uint32_t ne = 4 + r.below(9);                 // 4..12 exons
    std::vector<uint32_t> el(ne), il(ne - 1);
    uint64_t span = 0;
    for (uint32_t e = 0; e < ne; ++e) { el[e] = 80 + r.below(321); span += el[e]; }
    for (uint32_t i = 0; i + 1 < ne; ++i) { il[i] = (uint32_t)(70.0 * std::pow(50000.0 / 70.0, r.unit())); if (il[i] < 70) il[i] = 70; span += il[i]; }
    uint32_t pos;
    pick_locus(c, r, (uint32_t)std::min<uint64_t>(span + 400, 0x7fffffffu), G.tid, pos);
    pos += 150;
    G.es.resize(ne); G.ee.resize(ne);
    for (uint32_t e = 0; e < ne; ++e) { G.es[e] = pos; G.ee[e] = pos + el[e] - 1; pos += el[e] + (e + 1 < ne ? il[e] : 0); }
    G.tx.assign(4, {});
    for (uint32_t t = 0; t < 4; ++t) {
        for (uint32_t e = 0; e < ne; ++e)
            if (t == 0 || e == 0 || e + 1 == ne || r.below(4) != 0) G.tx[t].push_back((uint16_t)e);
    }
}

// ---- shape 0: config 2/3 ------------------------------------------------------------------------
void make_short(const Ctx &c, uint64_t rid, Read &rd) {
    static const uint16_t flags[4] = {99, 147, 83, 163};
    Rng r(c.p.seed, 1, rid);
    rd.flag = flags[r.below(4)];
    rd.l_qseq = 101;
    rd.cigar.clear(); rd.l_aux = 0;
    char xs;
    if (rid < c.n_spliced) {
        size_t k = std::upper_bound(c.intron_cum.begin(), c.intron_cum.end(), rid) - c.intron_cum.begin();
        const Intron &in = c.introns[k];
        uint32_t a = 1 + r.below(100);
        rd.tid = in.tid; rd.pos = (int32_t)(in.donor - a);
        rd.cigar = {cig(a, M), cig(in.len, N), cig(101 - a, M)};
        xs = in.strand;
    } else {
        uint32_t pos;
        pick_locus(c, r, 101, rd.tid, pos);
        rd.pos = (int32_t)pos;
        rd.cigar = {cig(101, M)};
        xs = (r.next() & 1) ? '+' : '-';
    }
    put_aux(rd, "NH", 'C', 1); put_aux(rd, "XS", 'A', (uint8_t)xs); put_aux(rd, "NM", 'C', 0);
}

// ---- shape 1: config 5 (long reads) ---------------------------------------------------------------
// A locus is a fixed chain of kLocusExons exons; a read covers 6-21 consecutive exons of one locus end to end, so the junctions
// of a locus repeat across the ~40 reads drawn from it (a long-read transcriptome, not 125 M unrelated introns).  What varies per
// read is which exons it spans and how each exon's bases are spelled (M/=/X runs with small I/D edits, soft clips).
void build_locus(const Ctx &c, uint64_t k, Locus &L) {
    Rng r(c.p.seed, 6, k);
    L.strand = (r.next() & 1) ? '+' : '-';
    uint32_t il[kLocusExons];
    uint64_t span = 0;
    for (uint32_t e = 0; e < kLocusExons; ++e) {
        L.len[e] = 170 + r.below(281);                // 170..450
        il[e] = (uint32_t)(70.0 * std::pow(20000.0 / 70.0, r.unit()));
        if (il[e] < 70) il[e] = 70;
        span += L.len[e] + (e + 1 < kLocusExons ? il[e] : 0);
    }
    uint32_t pos;
    pick_locus(c, r, (uint32_t)span + 64, L.tid, pos);
    for (uint32_t e = 0; e < kLocusExons; ++e) { L.start[e] = pos; pos += L.len[e] + il[e]; }
}

void make_long(const Ctx &c, uint64_t rid, Read &rd) {
    Rng r(c.p.seed, 2, rid);
    rd.flag = (r.next() & 1) ? 16 : 0;
    rd.cigar.clear(); rd.l_aux = 0;
    const Locus &L = c.loci[r.below((uint32_t)c.loci.size())];
    const uint32_t n_introns = 5 + r.below(16);           // 5..20 N ops
    const uint32_t first = r.below(kLocusExons - n_introns);
    uint32_t spare = 64 - (2 * n_introns + 1);            // ops left after one op per exon and the N ops
    uint32_t qlen = 0;
    std::vector<uint32_t> &cg = rd.cigar;
    if (r.below(4) == 0 && spare) { const uint32_t sc = 1 + r.below(30); cg.push_back(cig(sc, S)); qlen += sc; --spare; }
    const bool tail_clip = r.below(6) == 0 && spare;
    if (tail_clip) --spare;
    for (uint32_t b = 0; b <= n_introns; ++b) {
        const uint32_t e = first + b;
        uint32_t left = L.len[e];                         // reference bases of this exon still to spell
        // optional edit in the middle of the exon: costs two ops (the edit and the second run)
        if (spare >= 2 && r.below(3) == 0) {
            const uint32_t head = 20 + r.below(left - 60), k = r.below(3), el = 1 + r.below(4);
            cg.push_back(cig(head, r.below(8) == 7 ? EQ : M)); qlen += head; left -= head;
            if (k == 0) { cg.push_back(cig(el, I)); qlen += el; }
            else if (k == 1) { cg.push_back(cig(el, D)); left -= el; }
            else { cg.push_back(cig(el, X)); qlen += el; left -= el; }
            spare -= 2;
        }
        cg.push_back(cig(left, r.below(8) == 7 ? EQ : M)); qlen += left;
        if (b < n_introns) cg.push_back(cig(L.start[e + 1] - (L.start[e] + L.len[e]), N));
    }
    if (tail_clip) { const uint32_t sc = 1 + r.below(30); cg.push_back(cig(sc, S)); qlen += sc; }
    rd.l_qseq = qlen;
    rd.tid = L.tid; rd.pos = (int32_t)L.start[first];
    const char other = L.strand == '+' ? '-' : '+';
    put_aux(rd, "NH", 'C', 1); put_aux(rd, "ts", 'A', (uint8_t)(r.below(50) == 0 ? other : L.strand));
    put_aux(rd, "XS", 'A', (uint8_t)(r.below(50) == 0 ? other : L.strand));
}

// ---- shape 2: fuzz (tests) -------------------------------------------------------------------------
void make_fuzz(const Ctx &c, uint64_t rid, Read &rd) {
    Rng r(c.p.seed, 3, rid);
    rd.cigar.clear(); rd.l_aux = 0;
    rd.flag = (uint16_t)(r.next() & 0xfff);
    rd.mapq = (uint8_t)r.below(61);
    // the last ~1 % of ids are coordinate-less reads (tid = -1)
    bool nocoor = (rid % 97) == 0;
    uint32_t style = r.below(10);
    uint32_t n_ops = style < 3 ? 1 : style < 5 ? 3 : 1 + r.below(12);
    static const uint32_t ops[] = {M, M, M, N, N, I, D, S, H, P, EQ, X, M, N, 9 /* B */};
    uint32_t qlen = 0, rspan = 0;
    for (uint32_t k = 0; k < n_ops; ++k) {
        uint32_t op = ops[r.below(sizeof ops / sizeof ops[0])];
        if (style < 3) op = M;
        if (style >= 3 && style < 5) op = (k == 1) ? N : M;
        uint32_t len = (op == N) ? (r.below(8) == 0 ? 60 + r.below(20) : 70 + r.below(3000)) : (op == M || op == EQ) ? 1 + r.below(40) : 1 + r.below(5);
        if (r.below(50) == 0) len = 0;
        rd.cigar.push_back(cig(len, op));
        if (op == M || op == I || op == S || op == EQ || op == X) qlen += len;
        if (op == M || op == D || op == N || op == EQ || op == X) rspan += len;
    }
    if (r.below(40) == 0) rd.cigar.clear();  // mapped-looking read without CIGAR
    rd.l_qseq = qlen;
    if (nocoor) { rd.tid = -1; rd.pos = -1; rd.flag |= 4; if (rd.cigar.size() > 1) rd.cigar.resize(1); }  // tid<0 with n_cigar>1 is UB upstream
    else {
        uint32_t pos;
        // cluster reads so junction keys repeat
        Rng rl(c.p.seed, 4, rid % 61);
        pick_locus(c, rl, 20000, rd.tid, pos);
        pos += r.below(24);
        rd.pos = (int32_t)pos;
        if ((uint64_t)pos + rspan + 2 >= c.contigs[rd.tid].len) rd.pos = 1;
    }
    // aux zoo: the strand tag may be missing, of a wrong type, '.', NUL, or hidden behind other tags
    uint32_t a = r.below(12);
    auto putZ = [&](const char *tag, const char *z) { rd.aux[rd.l_aux++] = tag[0]; rd.aux[rd.l_aux++] = tag[1]; rd.aux[rd.l_aux++] = 'Z'; size_t n = strlen(z) + 1; memcpy(rd.aux + rd.l_aux, z, n); rd.l_aux += (uint32_t)n; };
    auto putI = [&](const char *tag, char t, uint32_t v, int nb) { rd.aux[rd.l_aux++] = tag[0]; rd.aux[rd.l_aux++] = tag[1]; rd.aux[rd.l_aux++] = (uint8_t)t; memcpy(rd.aux + rd.l_aux, &v, (size_t)nb); rd.l_aux += (uint32_t)nb; };
    if (a & 1) putI("NH", 'C', 1, 1);
    if (a == 2) putZ("MD", "10A5");
    if (a == 3) { rd.aux[rd.l_aux++] = 'Z'; rd.aux[rd.l_aux++] = 'B'; rd.aux[rd.l_aux++] = 'B'; rd.aux[rd.l_aux++] = 'S'; uint32_t n = 3; memcpy(rd.aux + rd.l_aux, &n, 4); rd.l_aux += 4; memset(rd.aux + rd.l_aux, 7, 6); rd.l_aux += 6; }
    if (a == 4) putI("AS", 'i', 77, 4);
    if (a == 5) putI("XT", 's', 9, 2);
    switch (r.below(9)) {
        case 0: put_aux(rd, "XS", 'A', '+'); break;
        case 1: put_aux(rd, "XS", 'A', '-'); break;
        case 2: put_aux(rd, "XS", 'A', '.'); break;
        case 3: putZ("XS", "+"); break;
        case 4: put_aux(rd, "XS", 'A', 0); break;
        case 5: put_aux(rd, "XS", 'A', '?'); break;
        case 6: put_aux(rd, "ts", 'A', '-'); put_aux(rd, "XS", 'A', '+'); break;
        case 7: break;  // no tag
        default: put_aux(rd, "XS", 'A', (r.next() & 1) ? '+' : '-'); break;
    }
    if (a == 7) putI("NM", 'C', 0, 1);
}

void make_read(const Ctx &c, uint64_t rid, Read &rd) {
    switch (c.p.shape) {
        case RGX_SHAPE_LONG: make_long(c, rid, rd); break;
        case RGX_SHAPE_FUZZ: make_fuzz(c, rid, rd); break;
        default: make_short(c, rid, rd); break;
    }
}

inline uint32_t record_size(const Read &rd, uint32_t l_qname) {
    return 4 + 32 + l_qname + 4 * (uint32_t)rd.cigar.size() + (rd.l_qseq + 1) / 2 + rd.l_qseq + rd.l_aux;
}

inline int32_t ref_span(const Read &rd) {
    int32_t l = 0;
    for (uint32_t cg : rd.cigar) { uint32_t op = cg & 0xf; if ((0x3C1A7 >> (op << 1)) & 2) l += (int32_t)(cg >> 4); }
    return l;
}
inline int32_t end_pos(const Read &rd) {  // sam.c:336-342 bam_endpos
    if (!(rd.flag & 4) && !rd.cigar.empty()) return rd.pos + ref_span(rd);
    return rd.pos + 1;
}

inline int reg2bin(int64_t beg, int64_t end) {  // SAM spec 5.3, min_shift 14, depth 5
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
void put64(uint8_t *p, uint64_t v) { put32(p, (uint32_t)v); put32(p + 4, (uint32_t)(v >> 32)); }

// serialise one record (sam.c:399-433 is the reader this must satisfy)
uint32_t write_record(const Ctx &c, const Read &rd, uint64_t sorted_idx, uint64_t rid, uint8_t *out) {
    char qname[32];
    int lq = snprintf(qname, sizeof qname, "q%014llu", (unsigned long long)sorted_idx) + 1;
    uint32_t sz = record_size(rd, (uint32_t)lq);
    put32(out, sz - 4);
    put32(out + 4, (uint32_t)rd.tid); put32(out + 8, (uint32_t)rd.pos);
    int bin = rd.pos < 0 ? 4680 : reg2bin(rd.pos, end_pos(rd));
    put32(out + 12, (uint32_t)bin << 16 | (uint32_t)rd.mapq << 8 | (uint32_t)lq);
    put32(out + 16, (uint32_t)rd.flag << 16 | (uint32_t)rd.cigar.size());
    put32(out + 20, rd.l_qseq);
    put32(out + 24, (uint32_t)rd.tid); put32(out + 28, (uint32_t)(rd.pos < 0 ? -1 : rd.pos + 150)); put32(out + 32, 0);
    uint8_t *p = out + 36;
    memcpy(p, qname, (size_t)lq); p += lq;
    for (uint32_t cg : rd.cigar) { put32(p, cg); p += 4; }
    uint32_t nseq = (rd.l_qseq + 1) / 2;
    if (!c.p.realistic_payload) { memset(p, 0x11, nseq); p += nseq; memset(p, 0xff, rd.l_qseq); p += rd.l_qseq; }
    else {
        Rng r(c.p.seed, 9, rid);
        static const uint8_t nib[4] = {1, 2, 4, 8};
        for (uint32_t i = 0; i < nseq; ++i) { uint64_t v = r.next(); p[i] = (uint8_t)(nib[v & 3] << 4 | nib[(v >> 2) & 3]); }
        p += nseq;
        static const uint8_t qb[8] = {2, 11, 25, 37, 37, 37, 37, 25};
        for (uint32_t i = 0; i < rd.l_qseq; i += 4) { uint64_t v = r.next(); uint8_t q = qb[v & 7]; for (uint32_t j = i; j < i + 4 && j < rd.l_qseq; ++j) p[j] = q; }
        p += rd.l_qseq;
    }
    memcpy(p, rd.aux, rd.l_aux); p += rd.l_aux;
    return sz;
}

// BGZF member around a raw-deflate payload (bgzf.c:63 g_magic, :525 block_length, footer CRC32+ISIZE)
size_t bgzf_compress(const uint8_t *src, uint32_t n, int level, std::vector<uint8_t> &dst) {
    size_t base = dst.size();
    dst.resize(base + 18 + compressBound(n) + 16 + 8);
    static const uint8_t magic[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(&dst[base], magic, 16);
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef *)src; zs.avail_in = n;
    zs.next_out = &dst[base + 18]; zs.avail_out = (uInt)(dst.size() - base - 18 - 8);
    deflate(&zs, Z_FINISH);
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    size_t total = 18 + clen + 8;
    if (total > 65536) {  // incompressible: store (cannot happen at 0xff00 with deflate's stored-block fallback, kept for safety)
        fprintf(stderr, "synth_bam: BGZF member overflow\n"); abort();
    }
    dst[base + 16] = (uint8_t)((total - 1) & 0xff); dst[base + 17] = (uint8_t)((total - 1) >> 8);
    put32(&dst[base + 18 + clen], (uint32_t)crc32(crc32(0, nullptr, 0), src, n));
    put32(&dst[base + 18 + clen + 4], n);
    dst.resize(base + total);
    return total;
}

struct ChunkRun { int32_t tid; uint32_t bin; uint64_t beg, end; };           // stream offsets
struct LinHit { int32_t tid; uint32_t win; uint64_t off; };
struct RefMeta { uint64_t beg = UINT64_MAX, end = 0, n_mapped = 0, n_unmapped = 0; };

struct ThreadOut {
    std::vector<uint8_t> comp;            // compressed members, in order
    std::vector<uint32_t> member_len;     // compressed length of each member
    std::vector<ChunkRun> runs;
    std::vector<LinHit> lin;
    std::map<int32_t, RefMeta> meta;
    uint64_t n_no_coor = 0, cigar_ops = 0;
};

void index_record(ThreadOut &to, int32_t tid, int32_t pos, int32_t endp, uint16_t flag, uint64_t off, uint64_t off_end) {
    if (tid < 0) { to.n_no_coor++; return; }
    uint32_t bin = (uint32_t)reg2bin(pos, endp);
    if (!to.runs.empty() && to.runs.back().tid == tid && to.runs.back().bin == bin && to.runs.back().end == off) to.runs.back().end = off_end;
    else to.runs.push_back({tid, bin, off, off_end});
    int64_t b = pos < 0 ? 0 : pos, e = endp <= pos ? (int64_t)pos + 1 : endp;
    for (int64_t w = b >> 14; w <= (e - 1) >> 14; ++w) {
        if (!to.lin.empty() && to.lin.back().tid == tid && to.lin.back().win >= (uint32_t)w) continue;  // only first touch per window matters (sorted input)
        to.lin.push_back({tid, (uint32_t)w, off});
    }
    RefMeta &m = to.meta[tid];
    if (off < m.beg) m.beg = off;
    if (off_end > m.end) m.end = off_end;
    if (flag & 4) m.n_unmapped++; else m.n_mapped++;
}

struct VoffMap {  // stream offset -> virtual offset
    uint64_t rec_base_stream = 0;        // stream offset (within record area) -> member index via division
    std::vector<uint64_t> coff;          // compressed offset of each record-area member
    uint64_t eof_coff = 0;               // compressed offset just past the last record member
    uint64_t total = 0;                  // total record-area bytes
    uint64_t operator()(uint64_t s) const {
        if (s >= total) return eof_coff << 16;
        uint64_t k = s / kBlock; return coff[k] << 16 | (s % kBlock);
    }
};

template <class Map>
std::vector<uint8_t> build_bai(int n_ref, std::vector<ThreadOut> &outs, const Map &vm) {
    struct RefIdx { std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> lin; RefMeta meta; bool has = false; };
    std::vector<RefIdx> refs((size_t)n_ref);
    uint64_t n_no_coor = 0;
    for (auto &to : outs) {
        n_no_coor += to.n_no_coor;
        for (auto &r : to.runs) {
            if (r.tid >= n_ref) continue;
            auto &v = refs[(size_t)r.tid].bins[r.bin];
            uint64_t b = vm(r.beg), e = vm(r.end);
            if (!v.empty() && v.back().second == b) v.back().second = e; else v.push_back({b, e});
        }
        for (auto &l : to.lin) {
            if (l.tid >= n_ref) continue;
            auto &lv = refs[(size_t)l.tid].lin;
            if (lv.size() <= l.win) lv.resize((size_t)l.win + 1, 0);
            uint64_t v = vm(l.off);
            if (lv[l.win] == 0 || v < lv[l.win]) lv[l.win] = v;
        }
        for (auto &kv : to.meta) {
            if (kv.first >= n_ref) continue;
            RefMeta &m = refs[(size_t)kv.first].meta; refs[(size_t)kv.first].has = true;
            m.beg = std::min(m.beg, kv.second.beg); m.end = std::max(m.end, kv.second.end);
            m.n_mapped += kv.second.n_mapped; m.n_unmapped += kv.second.n_unmapped;
        }
    }
    std::vector<uint8_t> out;
    auto w32 = [&](uint32_t v) { size_t n = out.size(); out.resize(n + 4); put32(&out[n], v); };
    auto w64 = [&](uint64_t v) { size_t n = out.size(); out.resize(n + 8); put64(&out[n], v); };
    out.insert(out.end(), {'B', 'A', 'I', 1});
    w32((uint32_t)n_ref);
    for (auto &r : refs) {
        w32((uint32_t)(r.bins.size() + (r.has ? 1 : 0)));
        for (auto &b : r.bins) {
            w32(b.first); w32((uint32_t)b.second.size());
            for (auto &c : b.second) { w64(c.first); w64(c.second); }
        }
        if (r.has) { w32(37450); w32(2); w64(vm(r.meta.beg)); w64(vm(r.meta.end)); w64(r.meta.n_mapped); w64(r.meta.n_unmapped); }
        // linear index: zeros are forward-filled by the loader (hts.c:1559-1560); write them the way samtools does
        for (size_t i = 1; i < r.lin.size(); ++i) if (r.lin[i] == 0) r.lin[i] = r.lin[i - 1];
        w32((uint32_t)r.lin.size());
        for (uint64_t v : r.lin) w64(v);
    }
    w64(n_no_coor);
    return out;
}

std::vector<uint8_t> header_bytes(const Ctx &c) {
    std::string text = "@HD\tVN:1.4\tSO:coordinate\n";
    for (auto &ct : c.contigs) text += "@SQ\tSN:" + ct.name + "\tLN:" + std::to_string(ct.len) + "\n";
    text += "@PG\tID:regtools_amd_synth\tPN:regtools_amd_synth\n";
    std::vector<uint8_t> h;
    auto w32 = [&](uint32_t v) { size_t n = h.size(); h.resize(n + 4); put32(&h[n], v); };
    h.insert(h.end(), {'B', 'A', 'M', 1});
    w32((uint32_t)text.size()); h.insert(h.end(), text.begin(), text.end());
    w32((uint32_t)c.contigs.size());
    for (auto &ct : c.contigs) { w32((uint32_t)ct.name.size() + 1); h.insert(h.end(), ct.name.begin(), ct.name.end()); h.push_back(0); w32(ct.len); }
    return h;
}

void setup(Ctx &c) {
    if (c.p.shape == RGX_SHAPE_FUZZ) c.contigs.assign(std::begin(kFuzz), std::end(kFuzz));
    else c.contigs.assign(std::begin(kHuman), std::end(kHuman));
    c.contig_off.clear(); c.genome_len = 0;
    for (auto &ct : c.contigs) { c.contig_off.push_back(c.genome_len); c.genome_len += ct.len; }
    if (c.p.shape == RGX_SHAPE_LONG) {
        uint64_t nl = c.p.n_introns ? c.p.n_introns : std::min<uint64_t>(20000, std::max<uint64_t>(64, c.p.n_reads / 40));   // <= 5x10^5 distinct junctions
        c.loci.resize(nl);
        for (uint64_t k = 0; k < nl; ++k) build_locus(c, k, c.loci[k]);
    }
    if (c.p.shape != RGX_SHAPE_SHORT) return;
    double frac = c.p.spliced_frac > 0 ? c.p.spliced_frac : 0.15;
    c.n_spliced = (uint64_t)((double)c.p.n_reads * frac);
    uint64_t ni = c.p.n_introns ? c.p.n_introns : 300000;
    if (ni > c.n_spliced) ni = c.n_spliced;
    c.introns.resize(ni);
    if (c.p.n_genes) {
        c.genes.resize(c.p.n_genes);
        for (uint64_t g = 0; g < c.p.n_genes; ++g) build_gene(c, g, c.genes[g]);
    }
    for (uint64_t k = 0; k < ni && c.p.n_genes; ++k) {
        Rng r(c.p.seed, 8, k);
        const Gene &G = c.genes[r.below(c.p.n_genes)];
        const uint32_t ne = (uint32_t)G.es.size();
        uint32_t a = r.below(ne - 1), b = a + 1;
        if (r.below(5) == 0 && a + 2 < ne) b = a + 2;            // exon skipping
        uint32_t donor = G.ee[a], acc = G.es[b] - 1;             // 0-based intron [donor, acc)
        if (r.below(10) == 0) donor += 1 + r.below(6);           // novel donor
        c.introns[k] = {G.tid, donor, acc - donor, G.strand};
    }
    for (uint64_t k = 0; k < ni && !c.p.n_genes; ++k) {
        Rng r(c.p.seed, 5, k);
        uint32_t len = (uint32_t)(70.0 * std::pow(500000.0 / 70.0, r.unit()));
        if (len < 70) len = 70;
        if (len > 500000) len = 500000;
        int32_t tid; uint32_t pos;
        pick_locus(c, r, len + 400, tid, pos);
        c.introns[k] = {tid, pos + 150, len, (r.next() & 1) ? '+' : '-'};
    }
    // Zipf(1.0) expected counts, deterministic rounding; leftovers go to the head of the table
    c.intron_cum.assign(ni, 0);
    if (ni) {
        double H = 0; for (uint64_t k = 1; k <= ni; ++k) H += 1.0 / (double)k;
        uint64_t used = 0; std::vector<uint64_t> cnt(ni);
        for (uint64_t k = 0; k < ni; ++k) { cnt[k] = (uint64_t)((double)c.n_spliced / H / (double)(k + 1)); used += cnt[k]; }
        for (uint64_t k = 0; used < c.n_spliced; k = (k + 1) % ni) { cnt[k]++; used++; }
        uint64_t acc = 0; for (uint64_t k = 0; k < ni; ++k) { acc += cnt[k]; c.intron_cum[k] = acc; }
    }
}

template <class F> void parallel_for(int threads, uint64_t n, F f) {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([=] { uint64_t a = n * (uint64_t)t / (uint64_t)threads, b = n * (uint64_t)(t + 1) / (uint64_t)threads; f(t, a, b); });
    for (auto &x : th) x.join();
}

int generate(const rgx_synth_params *pp, rgx_synth_result *res) {
    Ctx c; c.p = *pp;
    if (c.p.level <= 0) c.p.level = 6;
    int T = c.p.threads > 0 ? c.p.threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    const uint64_t n = c.p.n_reads;
    if (n >= (1ull << 32)) return 2;
    setup(c);

    // (1) coordinates + sort ---------------------------------------------------------------------
    struct KeyId { uint64_t key; uint32_t rid; };
    std::vector<KeyId> keys(n);
    const uint64_t unmapped_key = (uint64_t)c.contigs.size() << 40;
    parallel_for(T, n, [&](int, uint64_t a, uint64_t b) {
        Read rd;
        for (uint64_t i = a; i < b; ++i) {
            make_read(c, i, rd);
            keys[i].key = rd.tid < 0 ? unmapped_key : ((uint64_t)rd.tid << 40 | (uint64_t)(uint32_t)(rd.pos + 1));
            keys[i].rid = (uint32_t)i;
        }
    });
    {   // bucket by (tid, pos >> 20), then sort buckets in parallel
        auto bucket_of = [&](uint64_t k) { return (size_t)((k >> 40) * 512 + ((k & 0xffffffffffull) >> 20)); };
        size_t nb = (c.contigs.size() + 1) * 512;
        std::vector<uint64_t> cnt(nb + 1, 0);
        for (uint64_t i = 0; i < n; ++i) cnt[bucket_of(keys[i].key) + 1]++;
        for (size_t b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
        std::vector<KeyId> tmp(n);
        std::vector<uint64_t> cur(cnt.begin(), cnt.end() - 1);
        for (uint64_t i = 0; i < n; ++i) tmp[cur[bucket_of(keys[i].key)]++] = keys[i];
        keys.swap(tmp);
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&] {
            for (size_t b; (b = next.fetch_add(1)) < nb;)
                std::sort(keys.begin() + (ptrdiff_t)cnt[b], keys.begin() + (ptrdiff_t)cnt[b + 1],
                          [](const KeyId &x, const KeyId &y) { return x.key != y.key ? x.key < y.key : x.rid < y.rid; });
        });
        for (auto &x : th) x.join();
    }

    // (2) stream offsets -----------------------------------------------------------------------------
    std::vector<uint64_t> soff(n + 1, 0);
    parallel_for(T, n, [&](int, uint64_t a, uint64_t b) {
        Read rd;
        for (uint64_t i = a; i < b; ++i) { make_read(c, keys[i].rid, rd); soff[i + 1] = record_size(rd, 16); }
    });
    for (uint64_t i = 0; i < n; ++i) soff[i + 1] += soff[i];
    const uint64_t total = soff[n];
    const uint64_t n_members = (total + kBlock - 1) / kBlock;

    // (3) per-thread member ranges ------------------------------------------------------------------
    std::vector<ThreadOut> outs((size_t)T);
    parallel_for(T, n_members, [&](int t, uint64_t k0, uint64_t k1) {
        if (k0 >= k1) return;
        ThreadOut &to = outs[(size_t)t];
        const uint64_t B0 = k0 * kBlock, B1 = std::min<uint64_t>(k1 * kBlock, total);
        uint64_t i = (uint64_t)(std::upper_bound(soff.begin(), soff.end(), B0) - soff.begin()) - 1;
        std::vector<uint8_t> blockbuf(kBlock), rec(1 << 16);
        uint32_t fill = 0;
        Read rd;
        for (; i < n && soff[i] < B1; ++i) {
            make_read(c, keys[i].rid, rd);
            uint32_t sz = record_size(rd, 16);
            if (sz > rec.size()) rec.resize(sz);
            write_record(c, rd, i, keys[i].rid, rec.data());
            if (soff[i] >= B0) { index_record(to, rd.tid, rd.pos, end_pos(rd), rd.flag, soff[i], soff[i] + sz); to.cigar_ops += rd.cigar.size(); }
            uint64_t lo = std::max<uint64_t>(soff[i], B0), hi = std::min<uint64_t>(soff[i] + sz, B1);
            uint64_t p = lo;
            while (p < hi) {
                uint32_t take = (uint32_t)std::min<uint64_t>(hi - p, kBlock - fill);
                memcpy(&blockbuf[fill], rec.data() + (p - soff[i]), take);
                fill += take; p += take;
                if (fill == kBlock) { to.member_len.push_back((uint32_t)bgzf_compress(blockbuf.data(), fill, c.p.level, to.comp)); fill = 0; }
            }
        }
        if (fill) to.member_len.push_back((uint32_t)bgzf_compress(blockbuf.data(), fill, c.p.level, to.comp));
    });

    // (4) stitch: header member(s) | record members | EOF marker ------------------------------------------
    std::vector<uint8_t> hdr = header_bytes(c), hdr_comp;
    for (size_t p = 0; p < hdr.size(); p += kBlock) bgzf_compress(hdr.data() + p, (uint32_t)std::min<size_t>(kBlock, hdr.size() - p), c.p.level, hdr_comp);
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t bam_len = hdr_comp.size() + sizeof eof;
    for (auto &to : outs) bam_len += to.comp.size();
    uint8_t *bam = (uint8_t *)malloc(bam_len);
    if (!bam) return 3;
    memcpy(bam, hdr_comp.data(), hdr_comp.size());
    VoffMap vm; vm.total = total; vm.coff.reserve(n_members);
    size_t w = hdr_comp.size();
    uint64_t cigar_ops = 0;
    for (auto &to : outs) {
        size_t q = w;
        for (uint32_t l : to.member_len) { vm.coff.push_back(q); q += l; }
        memcpy(bam + w, to.comp.data(), to.comp.size()); w += to.comp.size();
        std::vector<uint8_t>().swap(to.comp);
        cigar_ops += to.cigar_ops;
    }
    vm.eof_coff = w;
    memcpy(bam + w, eof, sizeof eof); w += sizeof eof;

    std::vector<uint8_t> bai = build_bai((int)c.contigs.size(), outs, vm);
    res->bam = bam; res->bam_len = bam_len;
    res->bai = (uint8_t *)malloc(bai.size()); memcpy(res->bai, bai.data(), bai.size()); res->bai_len = bai.size();
    res->n_reads = n; res->n_spliced = c.n_spliced; res->n_blocks = n_members + (hdr.size() + kBlock - 1) / kBlock + 1;
    res->inflated_bytes = total + hdr.size(); res->cigar_ops = cigar_ops;
    return 0;
}

bool write_file(const std::string &path, const uint8_t *d, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(d, 1, n, f) == n;
    return fclose(f) == 0 && ok;
}

}  // namespace

extern "C" int rgx_synth_generate(const rgx_synth_params *p, rgx_synth_result *out) {
    memset(out, 0, sizeof *out);
    return generate(p, out);
}

extern "C" void rgx_synth_free(rgx_synth_result *r) { free(r->bam); free(r->bai); memset(r, 0, sizeof *r); }

extern "C" int rgx_synth_write(const rgx_synth_params *p, const char *path, rgx_synth_result *stats) {
    rgx_synth_result r;
    int rc = rgx_synth_generate(p, &r);
    if (rc) return rc;
    bool ok = write_file(path, r.bam, r.bam_len) && write_file(std::string(path) + ".bai", r.bai, r.bai_len);
    if (stats) { *stats = r; stats->bam = nullptr; stats->bai = nullptr; }
    free(r.bam); free(r.bai);
    return ok ? 0 : 4;
}

// ---- config 4 companions: GTF + VCF (+ FASTA) --------------------------------------------------------------------------------
namespace {
inline char synth_base(uint64_t seed, int32_t tid, uint32_t pos0) {
    const uint64_t h = mix64(mix64(seed * 0x2545F4914F6CDD1Dull + 99) ^ ((uint64_t)(uint32_t)tid << 40) ^ (uint64_t)(pos0 >> 5));
    return "ACGT"[(h >> (2 * (pos0 & 31))) & 3];
}
}

extern "C" int rgx_synth_annotation(const rgx_synth_params *pp, uint32_t n_variants, const char *gtf_path, const char *vcf_path, const char *fasta_path) {
    Ctx c; c.p = *pp; c.p.shape = RGX_SHAPE_SHORT; c.p.n_slices = 0; c.p.n_reads = 0;
    if (!c.p.n_genes) return 2;
    int T = c.p.threads > 0 ? c.p.threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    setup(c);
    if (gtf_path) {
        FILE *f = fopen(gtf_path, "w");
        if (!f) return 4;
        fprintf(f, "#synthetic annotation: seed %llu, %u genes x 4 transcripts\n", (unsigned long long)c.p.seed, c.p.n_genes);
        for (uint32_t g = 0; g < c.p.n_genes; ++g) {
            const Gene &G = c.genes[g];
            const char *chrom = c.contigs[(size_t)G.tid].name.c_str();
            for (size_t t = 0; t < G.tx.size(); ++t)
                for (uint16_t e : G.tx[t])
                    fprintf(f, "%s\tsynth\texon\t%u\t%u\t.\t%c\t.\tgene_id \"G%06u\"; transcript_id \"G%06u.T%zu\"; gene_name \"GENE%u\";\n", chrom, G.es[e], G.ee[e],
                            G.strand, g, g, t, g);
        }
        if (fclose(f) != 0) return 4;
    }
    if (vcf_path) {
        std::vector<uint64_t> v(n_variants);
        for (uint32_t i = 0; i < n_variants; ++i) {
            Rng r(c.p.seed, 9, i);
            int32_t tid; uint32_t pos1;
            if (r.below(20) == 0) {
                const Gene &G = c.genes[r.below(c.p.n_genes)];
                const uint32_t e = r.below((uint32_t)G.es.size());
                const uint32_t edge = (r.next() & 1) ? G.es[e] : G.ee[e];
                tid = G.tid; pos1 = edge + r.below(7) - 3;
            } else { pick_locus(c, r, 2, tid, pos1); }
            v[i] = (uint64_t)(uint32_t)tid << 32 | pos1;
        }
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        FILE *f = fopen(vcf_path, "w");
        if (!f) return 4;
        fputs("##fileformat=VCFv4.1\n", f);
        for (auto &ct : c.contigs) fprintf(f, "##contig=<ID=%s,length=%u>\n", ct.name.c_str(), ct.len);
        fputs("##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Depth\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n", f);
        for (uint64_t x : v) {
            const int32_t tid = (int32_t)(x >> 32); const uint32_t pos1 = (uint32_t)x;
            const char ref = synth_base(c.p.seed, tid, pos1 - 1);
            const char alt = ref == 'A' ? 'G' : ref == 'G' ? 'A' : ref == 'C' ? 'T' : 'C';
            fprintf(f, "%s\t%u\t.\t%c\t%c\t50\tPASS\tDP=%u\n", c.contigs[(size_t)tid].name.c_str(), pos1, ref, alt, 10 + (uint32_t)(x % 50));
        }
        if (fclose(f) != 0) return 4;
    }
    if (fasta_path) {
        FILE *f = fopen(fasta_path, "wb");
        FILE *fi = fopen((std::string(fasta_path) + ".fai").c_str(), "w");
        if (!f || !fi) { if (f) fclose(f); if (fi) fclose(fi); return 4; }
        uint64_t off = 0;
        std::vector<char> buf;
        for (size_t t = 0; t < c.contigs.size(); ++t) {
            const uint32_t len = c.contigs[t].len;
            off += (uint64_t)fprintf(f, ">%s\n", c.contigs[t].name.c_str());
            fprintf(fi, "%s\t%u\t%llu\t60\t61\n", c.contigs[t].name.c_str(), len, (unsigned long long)off);
            const uint64_t lines = ((uint64_t)len + 59) / 60, bytes = (uint64_t)len + lines;
            buf.resize(bytes);
            parallel_for(T, lines, [&](int, uint64_t a, uint64_t b) {
                for (uint64_t l = a; l < b; ++l) {
                    char *o = &buf[l * 61];
                    const uint32_t p0 = (uint32_t)(l * 60), n = std::min<uint32_t>(60, len - p0);
                    for (uint32_t k = 0; k < n; ++k) o[k] = synth_base(c.p.seed, (int32_t)t, p0 + k);
                    o[n] = '\n';
                }
            });
            if (fwrite(buf.data(), 1, bytes, f) != bytes) { fclose(f); fclose(fi); return 4; }
            off += bytes;
        }
        if (fclose(f) != 0 || fclose(fi) != 0) return 4;
    }
    return 0;
}

// ---- index an existing BAM (zlib inflate; tooling only) -----------------------------------------------
extern "C" int rgx_synth_index(const char *bam_path) {
    FILE *f = fopen(bam_path, "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END); long fl = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> file((size_t)fl);
    if (fl && fread(file.data(), 1, (size_t)fl, f) != (size_t)fl) { fclose(f); return 1; }
    fclose(f);
    std::vector<uint8_t> data; std::vector<uint64_t> coff, uoff;
    size_t off = 0;
    while (off + 18 <= file.size()) {
        size_t bl = (size_t)(file[off + 16] | file[off + 17] << 8) + 1;
        if (off + bl > file.size()) break;
        size_t base = data.size(); data.resize(base + 65536);
        z_stream zs; memset(&zs, 0, sizeof zs);
        zs.next_in = &file[off + 18]; zs.avail_in = (uInt)(bl - 18); zs.next_out = &data[base]; zs.avail_out = 65536;
        inflateInit2(&zs, -15); int zr = inflate(&zs, Z_FINISH); inflateEnd(&zs);
        if (zr != Z_STREAM_END) return 2;
        coff.push_back(off); uoff.push_back(base);
        data.resize(base + zs.total_out);
        off += bl;
    }
    auto voff = [&](uint64_t s) -> uint64_t {
        // first member whose range contains s (skipping empty members); s == end maps past the last data member
        size_t k = (size_t)(std::upper_bound(uoff.begin(), uoff.end(), s) - uoff.begin()) - 1;
        if (s >= data.size()) { // end of data: offset of the first member at/after the end
            size_t j = 0; while (j < uoff.size() && !(uoff[j] >= data.size())) ++j;
            return j < coff.size() ? coff[j] << 16 : (uint64_t)file.size() << 16;
        }
        return coff[k] << 16 | (s - uoff[k]);
    };
    auto r32 = [&](size_t p) { return (uint32_t)data[p] | (uint32_t)data[p + 1] << 8 | (uint32_t)data[p + 2] << 16 | (uint32_t)data[p + 3] << 24; };
    if (data.size() < 12 || memcmp(data.data(), "BAM\1", 4)) return 3;
    size_t p = 8 + r32(4);
    int n_ref = (int)r32(p); p += 4;
    for (int i = 0; i < n_ref; ++i) { uint32_t ln = r32(p); p += 4 + ln + 4; }
    std::vector<ThreadOut> outs(1);
    ThreadOut &to = outs[0];
    while (p + 36 <= data.size()) {
        uint32_t bs = r32(p);
        if (p + 4 + bs > data.size()) break;
        int32_t tid = (int32_t)r32(p + 4), pos = (int32_t)r32(p + 8);
        uint32_t x2 = r32(p + 12), x3 = r32(p + 16);
        Read rd; rd.tid = tid; rd.pos = pos; rd.flag = (uint16_t)(x3 >> 16);
        uint32_t nc = x3 & 0xffff, lq = x2 & 0xff;
        for (uint32_t k = 0; k < nc; ++k) rd.cigar.push_back(r32(p + 36 + lq + 4 * k));
        index_record(to, tid, pos, end_pos(rd), rd.flag, p, p + 4 + bs);
        p += 4 + bs;
    }
    std::vector<uint8_t> bai = build_bai(n_ref, outs, voff);
    return write_file(std::string(bam_path) + ".bai", bai.data(), bai.size()) ? 0 : 4;
}
