// cse_core.h -- per-variant / per-junction cores of `cis-splice-effects identify` (device + host compilable).
// Restated from the behavioural spec; file:line citations are relative to /root/reference/src.
#pragma once
#include "common.h"

namespace rgx {

// flat GTF model as the kernels see it (built on the host by GtfModel)
struct GtfView {
    const uint8_t  *tx_strand;    // '+' / '-'
    const uint32_t *tx_exon_off;  // first exon of transcript t in es/ee
    const uint32_t *tx_n_exons;
    const uint32_t *es, *ee;      // exon start/end exactly as written in the GTF (1-based inclusive), strand-sorted per transcript
    const uint64_t *bin_key;      // sorted (chrom index << 32 | UCSC bin); ties keep transcript-id order
    const uint32_t *bin_tx;       // transcript of each bin_key entry
    uint32_t n_bin;
    // optional direct index into the table above: entries of (contig c, bin b) are [bin_start[c * bin_stride + b], bin_start[c * bin_stride + b + 1]);
    // b < bin_stride (= the largest bin any transcript has, + 1).  nullptr: binary search (annotations with absurdly many contigs x bins)
    const uint32_t *bin_start;
    uint32_t bin_stride;
    uint32_t keep_single;         // `junctions annotate -S` (junctions_annotator.cc:392-393): single-exon transcripts take part in the junction scan
};

// bedFile.h:49-63: 7 levels, offsets with the upstream 32678 typo, first shift 14, next shift 3
RGX_HD uint32_t bin_offset(int lvl) {
    return lvl == 0 ? 32678u + 4096u + 512u + 64u + 8u + 1u : lvl == 1 ? 4681u : lvl == 2 ? 585u : lvl == 3 ? 73u : lvl == 4 ? 9u : lvl == 5 ? 1u : 0u;
}
// bedFile.h:339-354 getBin
RGX_HD uint32_t ucsc_bin(uint32_t start, uint32_t end) {
    --end; start >>= 14; end >>= 14;
    for (int i = 0; i < 7; ++i) { if (start == end) return bin_offset(i) + start; start >>= 3; end >>= 3; }
    return 0;
}

RGX_HD uint32_t bin_lower_bound(const GtfView &g, uint64_t key) {
    uint32_t lo = 0, hi = g.n_bin;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (g.bin_key[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// entries of the (contig, bin) table for bins b0 .. b1 of contig c: [lo, hi)
RGX_HD void bin_range(const GtfView &g, uint32_t c, uint32_t b0, uint32_t b1, uint32_t &lo, uint32_t &hi) {
    if (g.bin_start) {
        if (b0 >= g.bin_stride) { lo = hi = 0; return; }
        if (b1 >= g.bin_stride) b1 = g.bin_stride - 1;
        lo = g.bin_start[(size_t)c * g.bin_stride + b0]; hi = g.bin_start[(size_t)c * g.bin_stride + b1 + 1];
        return;
    }
    lo = bin_lower_bound(g, (uint64_t)c << 32 | b0);
    uint32_t a = lo, b = g.n_bin;
    const uint64_t k1 = (uint64_t)c << 32 | b1;
    while (a < b) { const uint32_t mid = (a + b) >> 1; if (g.bin_key[mid] <= k1) a = mid + 1; else b = mid; }
    hi = a;
}

enum : uint32_t { ANN_NONE = 0, ANN_EXONIC = 1, ANN_INTRONIC = 2, ANN_SPL_EXONIC = 3, ANN_SPL_INTRONIC = 4 };

struct VariantOpts { uint32_t intronic_min, exonic_min; int all_intronic, all_exonic, skip_single; };

RGX_HD uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// variants/variants_annotator.cc:169-239 set_variant_cis_effect_limits_{ps,ns}
RGX_HD void cis_limits(char strand, const uint32_t *s, const uint32_t *e, uint32_t n, uint32_t ann, uint32_t i, uint32_t &ces, uint32_t &cee) {
    const bool cassette = ann == ANN_EXONIC || ann == ANN_SPL_EXONIC || ann == ANN_SPL_INTRONIC;
    if (strand == '+') {
        if (cassette) {
            const uint32_t a = i != 0 ? s[i - 1] : s[0]; if (a < ces) ces = a;
            const uint32_t b = i != n - 1 ? e[i + 1] : e[n - 1]; if (b > cee) cee = b;
        } else if (ann == ANN_INTRONIC) { if (e[i] < ces) ces = e[i]; if (s[i + 1] > cee) cee = s[i + 1]; }
    } else {
        if (cassette) {
            const uint32_t b = i != 0 ? e[i - 1] : e[0]; if (b > cee) cee = b;
            const uint32_t a = i != n - 1 ? s[i + 1] : s[n - 1]; if (a < ces) ces = a;
        } else if (ann == ANN_INTRONIC) { if (s[i] > cee) cee = s[i]; if (e[i + 1] < ces) ces = e[i + 1]; }
    }
}

// variants_annotator.cc:263-431 get_variant_overlaps_spliceregion_{ps,ns}: v = variant.end (1-based position); uint32 wraps as upstream
RGX_HD uint32_t variant_vs_transcript(char strand, const uint32_t *s, const uint32_t *e, uint32_t n, uint32_t v, const VariantOpts &o, uint32_t &dist,
                                      uint32_t &ces, uint32_t &cee) {
    const uint32_t I = o.intronic_min, E = o.exonic_min;
#define RGX_HIT(A, D) do { dist = (D); cis_limits(strand, s, e, n, (A), i, ces, cee); return (A); } while (0)
    if (strand == '+') {
        if (s[0] > v || e[n - 1] < v) return ANN_NONE;
        for (uint32_t i = 0; i < n; ++i) {
            if (o.all_exonic && v >= s[i] && v <= e[i]) RGX_HIT(ANN_EXONIC, umin32(v - s[i], e[i] - v));
            if (o.all_intronic && i != n - 1 && v > e[i] && v < s[i + 1]) RGX_HIT(ANN_INTRONIC, umin32(v - e[i], s[i + 1] - v));
            if ((uint32_t)(s[i] - I) > v) return ANN_NONE;
            if (i != 0 && v >= s[i] && v <= e[i] && v <= (uint32_t)(s[i] + E)) RGX_HIT(ANN_SPL_EXONIC, umin32(v - s[i], e[i] - v));
            if (v < s[i] && v >= (uint32_t)(s[i] - I) && i != 0 && v > e[i - 1]) RGX_HIT(ANN_SPL_INTRONIC, umin32(v - e[i - 1], s[i] - v));
            if (i != n - 1 && v <= e[i] && v >= s[i] && v >= (uint32_t)(e[i] - E)) RGX_HIT(ANN_SPL_EXONIC, umin32(v - s[i], e[i] - v));
            if (v > e[i] && v <= (uint32_t)(e[i] + I) && i != n - 1 && v < s[i + 1]) RGX_HIT(ANN_SPL_INTRONIC, umin32(v - e[i], s[i + 1] - v));
        }
    } else {
        if (s[n - 1] > v || e[0] < v) return ANN_NONE;
        for (uint32_t i = 0; i < n; ++i) {
            if (o.all_exonic && v >= s[i] && v <= e[i]) RGX_HIT(ANN_EXONIC, umin32(v - s[i], e[i] - v));
            if (o.all_intronic && i != n - 1 && v < s[i] && v > e[i + 1]) RGX_HIT(ANN_INTRONIC, umin32(v - e[i + 1], s[i] - v));
            if ((uint32_t)(e[i] + I) < v) return ANN_NONE;
            if (i != n - 1 && v >= s[i] && v <= e[i] && v <= (uint32_t)(s[i] + E)) RGX_HIT(ANN_SPL_EXONIC, umin32(v - s[i], e[i] - v));
            if (v < s[i] && v >= (uint32_t)(s[i] - I) && i != n - 1 && v > e[i + 1]) RGX_HIT(ANN_SPL_INTRONIC, umin32(v - e[i + 1], s[i] - v));
            if (i != 0 && v <= e[i] && v >= s[i] && v >= (uint32_t)(e[i] - E)) RGX_HIT(ANN_SPL_EXONIC, umin32(v - s[i], e[i] - v));
            if (v > e[i] && v <= (uint32_t)(e[i] + I) && i != 0 && v < s[i - 1]) RGX_HIT(ANN_SPL_INTRONIC, umin32(v - e[i], s[i - 1] - v));
        }
    }
#undef RGX_HIT
    return ANN_NONE;
}

// One variant against every candidate transcript, in the reference's visitation order (level fine->coarse, bin ascending,
// transcript id ascending) -- variants_annotator.cc:455-518.  `hit(t, ann, dist)` is called per hit, in order.
// last (optional): what upstream's `variant.score` holds when the walk is over -- the distance of the LAST transcript looked at if that one was a hit,
// "-1" (0xffffffff here) if it was not (get_variant_overlaps_spliceregion_* reset the field on entry, :265 / :349); identify / associate print it
// with every splice-relevant variant (cis_splice_effects_identifier.cc:275).
template <class Hit>
RGX_HD void variant_scan(const GtfView &g, int32_t chrom, uint32_t pos0, const VariantOpts &o, uint32_t &ces, uint32_t &cee, uint32_t &exon_visits, Hit &&hit,
                         uint32_t *last = nullptr) {
    ces = 0xffffffffu; cee = 0; exon_visits = 0;
    if (last) *last = 0xffffffffu;
    if (chrom < 0) return;
    uint32_t sb = (uint32_t)(pos0 - o.intronic_min) >> 14, eb = (uint32_t)(pos0 + o.intronic_min) >> 14;
    for (int lvl = 0; lvl < 7; ++lvl) {
        const uint32_t off = bin_offset(lvl);
        if (sb <= eb) {
            uint32_t j0, j1;
            bin_range(g, (uint32_t)chrom, sb + off, eb + off, j0, j1);
            for (uint32_t j = j0; j < j1; ++j) {
                const uint32_t t = g.bin_tx[j], n = g.tx_n_exons[t];
                if (o.skip_single && n == 1) continue;
                exon_visits += n;
                uint32_t dist = 0;
                const uint32_t ann = variant_vs_transcript((char)g.tx_strand[t], g.es + g.tx_exon_off[t], g.ee + g.tx_exon_off[t], n, pos0 + 1, o, dist, ces, cee);
                if (ann != ANN_NONE) hit(t, ann, dist);
                if (last) *last = ann != ANN_NONE ? dist : 0xffffffffu;
            }
        }
        sb >>= 3; eb >>= 3;
    }
}

// ---- junction annotation (junctions/junctions_annotator.cc:128-363) ---------------------------------------------------------
struct JunctionFlags { uint32_t known_donor, known_acceptor, known_junction; };
enum : uint32_t { ITEM_TX = 0, ITEM_EXON = 1, ITEM_DONOR = 2, ITEM_ACCEPTOR = 3 };

RGX_HD bool anchor_not_N(const JunctionFlags &f) { return f.known_junction || f.known_donor || f.known_acceptor; }   // annotate_anchor :295-308

// overlap_ps / overlap_ns (:128-201, :228-292); js = junction.start, je = junction.end (= Junction.end + 1).
// item(kind, a, b) reports skipped exons / donors / acceptors (duplicates allowed; the caller makes them unique).
template <class Item>
RGX_HD bool junction_vs_transcript(char strand, const uint32_t *s, const uint32_t *e, uint32_t n, uint32_t js, uint32_t je, JunctionFlags &f, Item &&item, bool keep_single = false) {
    // skip_single_exon_genes_ (:131, :231): true unless `junctions annotate -S`; identify / associate never clear it (junctions_annotator.h:209-214).
    // A single exon can only end up a known donor / acceptor (every "skipped" test wants a neighbour); upstream's unchecked exons[i + 1] is "no match".
    if (n == 1 && !keep_single) return false;
    bool started = false;
    if (strand == '+') {
        if (s[0] > je || e[n - 1] < js) return false;
        for (uint32_t i = 0; i < n; ++i) {
            if (s[i] > je) break;
            if (e[i] == js && i + 1 < n && s[i + 1] == je) { f.known_acceptor = f.known_donor = f.known_junction = 1; }   // upstream reads exons[i+1] unchecked: "no match" 
            else {
                if (!started && e[i] >= js) started = true;
                if (started) {
                    if (s[i] > js && e[i] < je && i > 0 && i < n - 1) item(ITEM_EXON, s[i], e[i]);
                    if (e[i] > js && e[i] < je && i < n - 1) item(ITEM_DONOR, e[i], 0u);
                    if (s[i] < je && s[i] > js && i > 0) item(ITEM_ACCEPTOR, s[i], 0u);
                    if (e[i] == js) f.known_donor = 1;
                    if (s[i] == je) f.known_acceptor = 1;
                }
            }
        }
    } else {
        if (e[0] < js || s[n - 1] > je) return false;
        for (uint32_t i = 0; i < n; ++i) {
            if (e[i] < js) break;
            if (s[i] == je && i + 1 < n && e[i + 1] == js) { f.known_acceptor = f.known_donor = f.known_junction = 1; }
            else {
                if (!started && s[i] <= je) started = true;
                if (started) {
                    if (s[i] > js && e[i] < je && i > 0 && i < n - 1) item(ITEM_EXON, s[i], e[i]);
                    if (e[i] > js && e[i] < je && i < n - 1) item(ITEM_ACCEPTOR, e[i], 0u);
                    if (s[i] < je && s[i] > js) item(ITEM_DONOR, s[i], 0u);
                    if (e[i] == js) f.known_acceptor = 1;
                    if (s[i] == je) f.known_donor = 1;
                }
            }
        }
    }
    return anchor_not_N(f);     // order-dependent on purpose: flags accumulate over the transcripts visited so far
}

// annotate_junction_with_gtf (:344-363) + check_for_overlap (:313-340): transcripts visited in (level, bin, id) order
template <class Item>
RGX_HD void junction_scan(const GtfView &g, int32_t chrom, uint32_t js, uint32_t je, char strand, JunctionFlags &f, uint32_t &exon_visits, Item &&item) {
    f.known_donor = f.known_acceptor = f.known_junction = 0; exon_visits = 0;
    if (chrom < 0 || (strand != '+' && strand != '-')) return;     // '?' matches no transcript (:322-323)
    uint32_t sb = js >> 14, eb = (uint32_t)(je - 1) >> 14;
    for (int lvl = 0; lvl < 7; ++lvl) {
        const uint32_t off = bin_offset(lvl);
        if (sb <= eb) {
            uint32_t j0, j1;
            bin_range(g, (uint32_t)chrom, sb + off, eb + off, j0, j1);
            for (uint32_t j = j0; j < j1; ++j) {
                const uint32_t t = g.bin_tx[j];
                if ((char)g.tx_strand[t] != strand) continue;
                exon_visits += g.tx_n_exons[t];
                if (junction_vs_transcript(strand, g.es + g.tx_exon_off[t], g.ee + g.tx_exon_off[t], g.tx_n_exons[t], js, je, f, item, g.keep_single != 0)) item(ITEM_TX, t, 0u);
            }
        }
        sb >>= 3; eb >>= 3;
    }
}

}  // namespace rgx
