/* oracle/oracle_cse.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of `regtools cis-splice-effects identify`, plain C.
 * Citations are relative to /root/reference/src.  Pinned against the reference's 2 x 3 identify goldens and
 * against oracle/_ref on synthetic GTF/VCF/BAM/FASTA inputs (tests/golden/make_golden_cse.py).
 */
#define _POSIX_C_SOURCE 200809L
#include <ctype.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "oracle.h"
#include "oracle_internal.h"

/* ==================================================================================================
 * GTF model (gtf/gtf_parser.cc:63-263)
 * ================================================================================================ */
typedef struct {
    char *id, *chrom, *gene_name, *gene_id;
    char strand;
    uint32_t n_exons, cap;
    uint32_t *es, *ee;          /* exon start/end as written in the GTF (1-based inclusive), strand-sorted after load */
    uint32_t bin;
} gtf_tx;

typedef struct { gtf_tx *tx; size_t n, cap; } gtf_model;

static char *xstrndup(const char *s, size_t n) { char *r = (char *)malloc(n + 1); memcpy(r, s, n); r[n] = 0; return r; }

/* lineFileUtilities.h:24-33 Tokenize: std::getline semantics (no trailing empty field) */
static int tokenize(const char *s, size_t len, char delim, const char **beg, size_t *flen, int max) {
    int n = 0; size_t i = 0;
    if (len == 0) return 0;
    for (;;) {
        size_t j = i;
        while (j < len && s[j] != delim) ++j;
        if (n < max) { beg[n] = s + i; flen[n] = j - i; }
        ++n;
        if (j >= len) break;
        i = j + 1;
        if (i >= len) break;          /* "a\t" -> one field */
    }
    return n;
}

/* gtf_parser.cc:89-104 parse_attribute */
static char *gtf_attr(const char *attrs, size_t alen, const char *key) {
    size_t i = 0, klen = strlen(key);
    while (i < alen) {
        size_t j = i;
        while (j < alen && attrs[j] != ';') ++j;
        const char *p = attrs + i; size_t l = j - i;
        if (l && p[0] == ' ') { ++p; --l; }
        /* first token up to ' ', second token up to next ' ' */
        size_t a = 0; while (a < l && p[a] != ' ') ++a;
        if (a == klen && !memcmp(p, key, klen) && l > 0) {
            size_t b = a < l ? a + 1 : l, c = b;
            while (c < l && p[c] != ' ') ++c;
            const char *v = p + b; size_t vl = c - b;
            if (vl >= 1 && v[0] == '"' && v[vl - 1] == '"') { if (vl >= 2) { ++v; vl -= 2; } else { vl = 0; } }   /* common.h:85-92 unquote */
            return xstrndup(v, vl);
        }
        if (j >= alen) break;
        i = j + 1;
    }
    return strdup("NA");
}

static int tx_cmp_id(const void *a, const void *b) { return strcmp(((const gtf_tx *)a)->id, ((const gtf_tx *)b)->id); }

static gtf_tx *gtf_find_or_add(gtf_model *m, const char *id) {
    for (size_t i = 0; i < m->n; ++i) if (!strcmp(m->tx[i].id, id)) return &m->tx[i];   /* small inputs only: this is the oracle */
    if (m->n == m->cap) { m->cap = m->cap ? m->cap * 2 : 256; m->tx = (gtf_tx *)realloc(m->tx, m->cap * sizeof(gtf_tx)); }
    gtf_tx *t = &m->tx[m->n++];
    memset(t, 0, sizeof *t);
    t->id = strdup(id);
    return t;
}

/* returns 0 ok, 1 = the reference would exit(1) / throw */
static int gtf_load(const char *path, gtf_model *m, char *err, size_t errlen) {
    memset(m, 0, sizeof *m);
    size_t len; uint8_t *d = orc_slurp(path, &len);
    if (!d) { snprintf(err, errlen, "\nUnable to open GTF file."); return 1; }
    /* a fast id index for big files */
    size_t pos = 0;
    while (pos < len) {
        size_t e = pos; while (e < len && d[e] != '\n') ++e;
        const char *line = (const char *)d + pos; size_t ll = e - pos;
        pos = e + 1;
        if (ll == 0) { free(d); snprintf(err, errlen, "basic_string::at"); return 1; }      /* line.at(0) throws (gtf_parser.cc:230) */
        if (line[0] == '#') continue;
        const char *f[16]; size_t fl[16];
        int nf = tokenize(line, ll, '\t', f, fl, 16);
        if (nf != 9) { free(d); snprintf(err, errlen, "Expected 9 fields in GTF line."); return 1; }
        if (!(fl[2] == 4 && !memcmp(f[2], "exon", 4))) continue;
        char *tid = gtf_attr(f[8], fl[8], "transcript_id");
        if (!strcmp(tid, "NA")) { free(tid); continue; }
        gtf_tx *t = gtf_find_or_add(m, tid);
        free(tid);
        if (!t->chrom) {          /* first exon line of the transcript fixes gene name/id (gtf_parser.cc:266-273) */
            t->gene_name = gtf_attr(f[8], fl[8], "gene_name");
            t->gene_id = gtf_attr(f[8], fl[8], "gene_id");
        }
        char *cs = xstrndup(f[3], fl[3]), *ce = xstrndup(f[4], fl[4]);
        uint32_t st = (uint32_t)atol(cs), en = (uint32_t)atol(ce);
        free(cs); free(ce);
        if (t->n_exons == t->cap) { t->cap = t->cap ? t->cap * 2 : 8; t->es = (uint32_t *)realloc(t->es, t->cap * 4); t->ee = (uint32_t *)realloc(t->ee, t->cap * 4); }
        if (t->n_exons == 0) { t->chrom = xstrndup(f[0], fl[0]); t->strand = fl[6] == 1 ? f[6][0] : '?'; }
        t->es[t->n_exons] = st; t->ee[t->n_exons] = en; t->n_exons++;
    }
    free(d);
    qsort(m->tx, m->n, sizeof(gtf_tx), tx_cmp_id);          /* std::map<string,Transcript> order */
    for (size_t i = 0; i < m->n; ++i) {
        gtf_tx *t = &m->tx[i];
        if (t->strand != '+' && t->strand != '-') { snprintf(err, errlen, "Undefined strand for exon "); return 1; }   /* gtf_parser.cc:193-197 exit(1) */
        /* sort_exons_within_transcripts: '+' ascending start, '-' descending start (insertion sort: stable) */
        for (uint32_t a = 1; a < t->n_exons; ++a) {
            uint32_t s = t->es[a], e = t->ee[a]; int b = (int)a - 1;
            while (b >= 0 && (t->strand == '+' ? t->es[b] > s : t->es[b] < s)) { t->es[b + 1] = t->es[b]; t->ee[b + 1] = t->ee[b]; --b; }
            t->es[b + 1] = s; t->ee[b + 1] = e;
        }
        t->bin = orc_get_bin(t->es[0], t->ee[t->n_exons - 1]);      /* gtf_parser.cc:154-160 */
    }
    return 0;
}

static void gtf_free(gtf_model *m) {
    for (size_t i = 0; i < m->n; ++i) { free(m->tx[i].id); free(m->tx[i].chrom); free(m->tx[i].gene_name); free(m->tx[i].gene_id); free(m->tx[i].es); free(m->tx[i].ee); }
    free(m->tx);
}

static const uint32_t kBinOff[7] = { 32678 + 4096 + 512 + 64 + 8 + 1, 4096 + 512 + 64 + 8 + 1, 512 + 64 + 8 + 1, 64 + 8 + 1, 8 + 1, 1, 0 };

/* ==================================================================================================
 * variants (variants/variants_annotator.cc:169-518)
 * ================================================================================================ */
typedef struct {
    uint32_t intronic_min, exonic_min;
    int all_intronic, all_exonic, skip_single;
} va_opts;

enum { ANN_NONE = 0, ANN_EXONIC, ANN_INTRONIC, ANN_SPL_EXONIC, ANN_SPL_INTRONIC };
static const char *kAnn[] = { "non_splice_region", "exonic", "intronic", "splicing_exonic", "splicing_intronic" };

typedef struct { uint32_t ces, cee; } cis_lim;

static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

/* set_variant_cis_effect_limits_{ps,ns} :169-239 */
static void cis_limits(const gtf_tx *t, int ann, uint32_t i, cis_lim *c) {
    const uint32_t *s = t->es, *e = t->ee; uint32_t n = t->n_exons;
    if (t->strand == '+') {
        if (ann == ANN_EXONIC || ann == ANN_SPL_EXONIC || ann == ANN_SPL_INTRONIC) {
            uint32_t a = i != 0 ? s[i - 1] : s[0]; if (a < c->ces) c->ces = a;
            uint32_t b = i != n - 1 ? e[i + 1] : e[n - 1]; if (b > c->cee) c->cee = b;
        } else if (ann == ANN_INTRONIC) {
            if (e[i] < c->ces) c->ces = e[i];
            if (s[i + 1] > c->cee) c->cee = s[i + 1];
        }
    } else {
        if (ann == ANN_EXONIC || ann == ANN_SPL_EXONIC || ann == ANN_SPL_INTRONIC) {
            uint32_t b = i != 0 ? e[i - 1] : e[0]; if (b > c->cee) c->cee = b;
            uint32_t a = i != n - 1 ? s[i + 1] : s[n - 1]; if (a < c->ces) c->ces = a;
        } else if (ann == ANN_INTRONIC) {
            if (s[i] > c->cee) c->cee = s[i];
            if (e[i + 1] < c->ces) c->ces = e[i + 1];
        }
    }
}

/* get_variant_overlaps_spliceregion_{ps,ns} :263-431.  vend = variant.end (1-based position). Returns the annotation,
 * *dist the score, and updates the running cis window. All arithmetic is uint32 (wraps as upstream). */
static int variant_vs_transcript(const gtf_tx *t, uint32_t vend, const va_opts *o, uint32_t *dist, cis_lim *c) {
    const uint32_t *s = t->es, *e = t->ee; uint32_t n = t->n_exons;
    const uint32_t I = o->intronic_min, E = o->exonic_min;
    if (t->strand == '+') {
        if (s[0] > vend || e[n - 1] < vend) return ANN_NONE;
        for (uint32_t i = 0; i < n; ++i) {
            if (o->all_exonic && vend >= s[i] && vend <= e[i]) { *dist = umin(vend - s[i], e[i] - vend); cis_limits(t, ANN_EXONIC, i, c); return ANN_EXONIC; }
            if (o->all_intronic && i != n - 1 && vend > e[i] && vend < s[i + 1]) { *dist = umin(vend - e[i], s[i + 1] - vend); cis_limits(t, ANN_INTRONIC, i, c); return ANN_INTRONIC; }
            if ((uint32_t)(s[i] - I) > vend) return ANN_NONE;
            if (i != 0 && vend >= s[i] && vend <= e[i] && vend <= (uint32_t)(s[i] + E)) { *dist = umin(vend - s[i], e[i] - vend); cis_limits(t, ANN_SPL_EXONIC, i, c); return ANN_SPL_EXONIC; }
            if (vend < s[i] && vend >= (uint32_t)(s[i] - I) && i != 0 && vend > e[i - 1]) { *dist = umin(vend - e[i - 1], s[i] - vend); cis_limits(t, ANN_SPL_INTRONIC, i, c); return ANN_SPL_INTRONIC; }
            if (i != n - 1 && vend <= e[i] && vend >= s[i] && vend >= (uint32_t)(e[i] - E)) { *dist = umin(vend - s[i], e[i] - vend); cis_limits(t, ANN_SPL_EXONIC, i, c); return ANN_SPL_EXONIC; }
            if (vend > e[i] && vend <= (uint32_t)(e[i] + I) && i != n - 1 && vend < s[i + 1]) { *dist = umin(vend - e[i], s[i + 1] - vend); cis_limits(t, ANN_SPL_INTRONIC, i, c); return ANN_SPL_INTRONIC; }
        }
    } else {
        if (s[n - 1] > vend || e[0] < vend) return ANN_NONE;
        for (uint32_t i = 0; i < n; ++i) {
            if (o->all_exonic && vend >= s[i] && vend <= e[i]) { *dist = umin(vend - s[i], e[i] - vend); cis_limits(t, ANN_EXONIC, i, c); return ANN_EXONIC; }
            if (o->all_intronic && i != n - 1 && vend < s[i] && vend > e[i + 1]) { *dist = umin(vend - e[i + 1], s[i] - vend); cis_limits(t, ANN_INTRONIC, i, c); return ANN_INTRONIC; }
            if ((uint32_t)(e[i] + I) < vend) return ANN_NONE;
            if (i != n - 1 && vend >= s[i] && vend <= e[i] && vend <= (uint32_t)(s[i] + E)) { *dist = umin(vend - s[i], e[i] - vend); cis_limits(t, ANN_SPL_EXONIC, i, c); return ANN_SPL_EXONIC; }
            if (vend < s[i] && vend >= (uint32_t)(s[i] - I) && i != n - 1 && vend > e[i + 1]) { *dist = umin(vend - e[i + 1], s[i] - vend); cis_limits(t, ANN_SPL_INTRONIC, i, c); return ANN_SPL_INTRONIC; }
            if (i != 0 && vend <= e[i] && vend >= s[i] && vend >= (uint32_t)(e[i] - E)) { *dist = umin(vend - s[i], e[i] - vend); cis_limits(t, ANN_SPL_EXONIC, i, c); return ANN_SPL_EXONIC; }
            if (vend > e[i] && vend <= (uint32_t)(e[i] + I) && i != 0 && vend < s[i - 1]) { *dist = umin(vend - e[i], s[i - 1] - vend); cis_limits(t, ANN_SPL_INTRONIC, i, c); return ANN_SPL_INTRONIC; }
        }
    }
    return ANN_NONE;
}

typedef struct { char *buf; size_t n, cap; } sbuf;
static void sb_add(sbuf *b, const char *s) { size_t l = strlen(s); if (b->n + l + 1 > b->cap) { b->cap = (b->n + l + 1) * 2; b->buf = (char *)realloc(b->buf, b->cap); } memcpy(b->buf + b->n, s, l + 1); b->n += l; }

typedef struct {
    char *chrom; uint32_t start, end;       /* (pos0, pos0+1) */
    uint32_t ces, cee;
    sbuf genes, transcripts, distances, annotations;   /* "NA" when empty */
    int relevant;
} ann_variant;

/* annotate_record_with_transcripts :455-518 */
static void annotate_variant(const gtf_model *m, const va_opts *o, const char *chrom, uint32_t pos0, ann_variant *v) {
    memset(v, 0, sizeof *v);
    v->chrom = strdup(chrom); v->start = pos0; v->end = pos0 + 1;
    cis_lim c = { UINT_MAX, 0 };
    uint32_t sb = (uint32_t)(pos0 - o->intronic_min) >> 14, eb = (uint32_t)(pos0 + o->intronic_min) >> 14;
    const char **seen_genes = NULL; size_t n_seen = 0;
    for (int lvl = 0; lvl < 7; ++lvl) {
        for (uint32_t b = sb + kBinOff[lvl]; b <= eb + kBinOff[lvl]; ++b) {
            for (size_t k = 0; k < m->n; ++k) {            /* transcripts of (chrom, bin) in transcript-id order */
                const gtf_tx *t = &m->tx[k];
                if (t->bin != b || strcmp(t->chrom, chrom)) continue;
                if (o->skip_single && t->n_exons == 1) continue;
                uint32_t dist = 0;
                int ann = variant_vs_transcript(t, v->end, o, &dist, &c);
                if (ann == ANN_NONE) continue;
                char num[16]; snprintf(num, sizeof num, "%u", dist);
                int first = v->transcripts.n == 0;
                int gene_seen = 0;
                for (size_t g = 0; g < n_seen; ++g) if (!strcmp(seen_genes[g], t->gene_name)) gene_seen = 1;
                if (!gene_seen) { if (!first) sb_add(&v->genes, ","); sb_add(&v->genes, t->gene_name); seen_genes = (const char **)realloc(seen_genes, (n_seen + 1) * sizeof(char *)); seen_genes[n_seen++] = t->gene_name; }
                if (!first) { sb_add(&v->distances, ","); sb_add(&v->transcripts, ","); sb_add(&v->annotations, ","); }
                sb_add(&v->distances, num); sb_add(&v->transcripts, t->id); sb_add(&v->annotations, kAnn[ann]);
            }
            if (b == UINT_MAX) break;
        }
        sb >>= 3; eb >>= 3;
    }
    free(seen_genes);
    v->relevant = v->transcripts.n != 0;
    if (!v->relevant) { sb_add(&v->genes, "NA"); sb_add(&v->transcripts, "NA"); sb_add(&v->distances, "NA"); sb_add(&v->annotations, "NA"); }
    v->ces = c.ces; v->cee = c.cee;
}
static void variant_free(ann_variant *v) { free(v->chrom); free(v->genes.buf); free(v->transcripts.buf); free(v->distances.buf); free(v->annotations.buf); }

/* ==================================================================================================
 * junction annotation (junctions/junctions_annotator.cc:128-363)
 * ================================================================================================ */
typedef struct {
    int known_donor, known_acceptor, known_junction;
    char anchor[4];
    uint32_t *acc, *don; size_t n_acc, n_don;            /* unique coordinates */
    uint64_t *exo; size_t n_exo;                          /* unique (start<<32|end) */
    const gtf_tx **txs; size_t n_tx;                      /* transcripts_overlap (unique) */
} jann;

static void uadd32(uint32_t **a, size_t *n, uint32_t v) { for (size_t i = 0; i < *n; ++i) if ((*a)[i] == v) return; *a = (uint32_t *)realloc(*a, (*n + 1) * 4); (*a)[(*n)++] = v; }
static void uadd64(uint64_t **a, size_t *n, uint64_t v) { for (size_t i = 0; i < *n; ++i) if ((*a)[i] == v) return; *a = (uint64_t *)realloc(*a, (*n + 1) * 8); (*a)[(*n)++] = v; }

static void set_anchor(jann *j) {   /* annotate_anchor :295-308 */
    const char *a = "N";
    if (j->known_junction) a = "DA";
    else if (j->known_donor) a = j->known_acceptor ? "NDA" : "D";
    else if (j->known_acceptor) a = "A";
    strcpy(j->anchor, a);
}

/* overlap_ps / overlap_ns :128-201, :228-292; js = junction.start, je = junction.end (= Junction.end + 1) */
/* skip_single_exon_genes_ (junctions_annotator.cc:131, :231): cleared only by `junctions annotate -S` (:392-393); identify / associate keep it (h:209-214) */
static int g_keep_single_exon = 0;
static int overlap(const gtf_tx *t, uint32_t js, uint32_t je, jann *j) {
    const uint32_t *s = t->es, *e = t->ee; uint32_t n = t->n_exons;
    if (n == 1 && !g_keep_single_exon) return 0;
    int junction_start = 0;
    if (t->strand == '+') {
        if (s[0] > je || e[n - 1] < js) return 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (s[i] > je) break;
            if (e[i] == js && i + 1 < n && s[i + 1] == je) { j->known_acceptor = j->known_donor = j->known_junction = 1; }   /* exons[i+1] read past the end upstream when i is last: "no match" */
            else {
                if (!junction_start && e[i] >= js) junction_start = 1;
                if (junction_start) {
                    if (s[i] > js && e[i] < je && i > 0 && i < n - 1) uadd64(&j->exo, &j->n_exo, (uint64_t)s[i] << 32 | e[i]);
                    if (e[i] > js && e[i] < je && i < n - 1) uadd32(&j->don, &j->n_don, e[i]);
                    if (s[i] < je && s[i] > js && i > 0) uadd32(&j->acc, &j->n_acc, s[i]);
                    if (e[i] == js) j->known_donor = 1;
                    if (s[i] == je) j->known_acceptor = 1;
                }
            }
        }
    } else {
        if (e[0] < js || s[n - 1] > je) return 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (e[i] < js) break;
            if (s[i] == je && i + 1 < n && e[i + 1] == js) { j->known_acceptor = j->known_donor = j->known_junction = 1; }
            else {
                if (!junction_start && s[i] <= je) junction_start = 1;
                if (junction_start) {
                    if (s[i] > js && e[i] < je && i > 0 && i < n - 1) uadd64(&j->exo, &j->n_exo, (uint64_t)s[i] << 32 | e[i]);
                    if (e[i] > js && e[i] < je && i < n - 1) uadd32(&j->acc, &j->n_acc, e[i]);
                    if (s[i] < je && s[i] > js) uadd32(&j->don, &j->n_don, s[i]);
                    if (e[i] == js) j->known_acceptor = 1;
                    if (s[i] == je) j->known_donor = 1;
                }
            }
        }
    }
    set_anchor(j);
    return strcmp(j->anchor, "N") != 0;
}

/* annotate_junction_with_gtf :344-363 + check_for_overlap :313-340 */
static void annotate_junction(const gtf_model *m, const char *chrom, uint32_t js, uint32_t je, char strand, jann *j) {
    memset(j, 0, sizeof *j);
    strcpy(j->anchor, "N");
    uint32_t sb = js >> 14, eb = (uint32_t)(je - 1) >> 14;
    for (int lvl = 0; lvl < 7; ++lvl) {
        for (uint32_t b = sb + kBinOff[lvl]; b <= eb + kBinOff[lvl]; ++b) {
            for (size_t k = 0; k < m->n; ++k) {
                const gtf_tx *t = &m->tx[k];
                if (t->bin != b || strcmp(t->chrom, chrom)) continue;
                if (strand != t->strand) continue;           /* '?' and friends never match a transcript */
                if (overlap(t, js, je, j)) {
                    int dup = 0; for (size_t q = 0; q < j->n_tx; ++q) if (j->txs[q] == t) dup = 1;
                    if (!dup) { j->txs = (const gtf_tx **)realloc(j->txs, (j->n_tx + 1) * sizeof(void *)); j->txs[j->n_tx++] = t; }
                }
            }
        }
        sb >>= 3; eb >>= 3;
    }
}
static void jann_free(jann *j) { free(j->acc); free(j->don); free(j->exo); free(j->txs); }

/* ==================================================================================================
 * VCF text (vcf.c:1782-1958 reads CHROM and POS; everything else passes through)
 * ================================================================================================ */
typedef struct { char *text; size_t len; size_t *line_off; size_t n_lines; } textfile;

static int text_load(const char *path, textfile *t) {
    memset(t, 0, sizeof *t);
    gzFile g = gzopen(path, "rb");                /* plain or gzip/bgzip, as hts_open accepts */
    if (!g) return 1;
    size_t cap = 1 << 16;
    t->text = (char *)malloc(cap);
    for (;;) {
        if (t->len + 65536 > cap) { cap *= 2; t->text = (char *)realloc(t->text, cap); }
        int r = gzread(g, t->text + t->len, 65536);
        if (r <= 0) break;
        t->len += (size_t)r;
    }
    gzclose(g);
    size_t capl = 1024; t->line_off = (size_t *)malloc(capl * sizeof(size_t));
    size_t p = 0;
    while (p < t->len) {
        if (t->n_lines + 2 > capl) { capl *= 2; t->line_off = (size_t *)realloc(t->line_off, capl * sizeof(size_t)); }
        t->line_off[t->n_lines++] = p;
        while (p < t->len && t->text[p] != '\n') ++p;
        ++p;
    }
    t->line_off[t->n_lines] = t->len + (t->len && t->text[t->len - 1] != '\n' ? 1 : 0);
    return 0;
}

/* htslib's header normaliser (vcf.c:116 bcf_hdr_sync / :821 bcf_hdr_write): the PASS filter right after ##fileformat unless the header
 * declares it, the four appended INFO lines right before #CHROM */
static void vcf_header_line(FILE *fv, const textfile *vcf, size_t li, const char *line, size_t ll, int *header_done) {
    if (ll >= 6 && !memcmp(line, "#CHROM", 6)) {
        fputs("##INFO=<ID=genes,Number=1,Type=String,Description=\"The Variant falls in the splice region of these genes\">\n", fv);
        fputs("##INFO=<ID=transcripts,Number=1,Type=String,Description=\"The Variant falls in the splice region of these transcripts\">\n", fv);
        fputs("##INFO=<ID=distances,Number=1,Type=String,Description=\"Vector of Min(Distance from start/end of exon in the transcript.)\">\n", fv);
        fputs("##INFO=<ID=annotations,Number=1,Type=String,Description=\"Does the variant fall in exonic/intronic splicing related space in the transcript.\">\n", fv);
    }
    fwrite(line, 1, ll, fv); fputc('\n', fv);
    if (!*header_done && ll >= 12 && !memcmp(line, "##fileformat", 12)) {
        int declared = 0;
        for (size_t k = li + 1; k < vcf->n_lines && vcf->text[vcf->line_off[k]] == '#'; ++k)
            if (!strncmp(vcf->text + vcf->line_off[k], "##FILTER=<ID=PASS", 17)) declared = 1;
        if (!declared) fputs("##FILTER=<ID=PASS,Description=\"All filters passed\">\n", fv);
        *header_done = 1;
    }
}
/* bcf_update_info_string x4 + vcf_format (vcf.c:2783, :2069): the record text with the four tags appended to INFO (column 8) */
static void vcf_record(FILE *fv, const char *line, size_t ll, const char **f, const size_t *fl, int nf, const ann_variant *v) {
    for (int k = 0; k < nf && k < 16; ++k) {
        if (k) fputc('\t', fv);
        if (k == 7) {
            if (!(fl[7] == 1 && f[7][0] == '.')) { fwrite(f[7], 1, fl[7], fv); fputc(';', fv); }
            fprintf(fv, "genes=%s;transcripts=%s;distances=%s;annotations=%s", v->genes.buf, v->transcripts.buf, v->distances.buf, v->annotations.buf);
        } else fwrite(f[k], 1, fl[k], fv);
    }
    if (nf > 16) { const char *rest = f[15] + fl[15]; fwrite(rest, 1, (size_t)(line + ll - rest), fv); }
    fputc('\n', fv);
}

/* ==================================================================================================
 * the driver (cis-splice-effects/cis_splice_effects_identifier.cc:222-312)
 * ================================================================================================ */
typedef struct {
    char *chrom; uint32_t start, end, ts, te, count; char strand;   /* first-inserted row for this (chrom,start,end) */
    ann_variant **vars; size_t n_vars;
    const char *strand_s, *color; int nblocks;                      /* associate: strand string, colour and block count of the BED row */
} cse_junction;

static int cj_cmp(const void *a, const void *b) {     /* AnnotatedJunction operator< (junctions_annotator.h:169-177) on (chrom, start, end+1) */
    const cse_junction *x = (const cse_junction *)a, *y = (const cse_junction *)b;
    int c = strcmp(x->chrom, y->chrom); if (c) return c;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (x->end != y->end) return x->end < y->end ? -1 : 1;
    return 0;
}
static int var_cmp(const void *a, const void *b) {    /* AnnotatedVariant operator< (variants_annotator.h:64-72) */
    const ann_variant *x = *(ann_variant *const *)a, *y = *(ann_variant *const *)b;
    int c = strcmp(x->chrom, y->chrom); if (c) return c;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (x->end != y->end) return x->end < y->end ? -1 : 1;
    return 0;
}

/* ==================================================================================================
 * BED12 junction rows as bedtools' BedFile hands them out (bedFile.cpp:103-260, bedFile.h:565-780) and
 * JunctionsAnnotator::adjust_junction_ends moves them (junctions_annotator.cc:66-81)
 * ================================================================================================ */
typedef struct { char *chrom, *name, *score, *strand, *color; uint32_t start, end, ts, te; int nblocks; } bed_junc;

static int bed_is_header(const char *s, size_t n) {
    return (n >= 1 && s[0] == '#') || (n >= 7 && !memcmp(s, "browser", 7)) || (n >= 5 && !memcmp(s, "track", 5));
}
static int all_digits(const char *s, size_t n) { for (size_t i = 0; i < n; ++i) if (s[i] < '0' || s[i] > '9') return 0; return 1; }

/* returns 0 ok, 1 = the reference exits with status 1 (message in err) */
static int bed_load(const char *path, bed_junc **out, size_t *n_out, char *err, size_t errlen) {
    textfile t;
    *out = NULL; *n_out = 0;
    if (text_load(path, &t)) { snprintf(err, errlen, "Error: The requested file (%s) could not be opened. Exiting!\n", path); return 1; }
    size_t li = 0, bed_type = 0;
    int rc = 0;
    /* GetHeader: leading lines that start with #, browser or track */
    for (; li < t.n_lines; ++li) {
        const char *l = t.text + t.line_off[li]; size_t ll = t.line_off[li + 1] - t.line_off[li]; if (ll) --ll;
        if (!bed_is_header(l, ll)) break;
    }
    for (int first = 1; li < t.n_lines; ++li, first = 0) {
        const char *l = t.text + t.line_off[li]; size_t ll = t.line_off[li + 1] - t.line_off[li]; if (ll) --ll;
        if (ll && l[ll - 1] == '\r') --ll;
        const char *f[32]; size_t fl[32];
        int nf = ll ? tokenize(l, ll, '\t', f, fl, 32) : 0;
        if (first) bed_type = (size_t)nf;
        if (nf == 0) break;                                                   /* BED_BLANK: get_single_junction() returns false */
        if (bed_is_header(f[0], fl[0])) break;                                /* BED_HEADER after the first data line: iteration ends too */
        if (nf < 3) { snprintf(err, errlen, "It looks as though you have less than 3 columns at line: %zu.  Are you sure your files are tab-delimited?\n", li + 1); rc = 1; break; }
        if (!(all_digits(f[1], fl[1]) && all_digits(f[2], fl[2]))) { snprintf(err, errlen, "Unexpected file format.  Please use tab-delimited BED, GFF, or VCF.\n"); rc = 1; break; }
        if ((size_t)nf != bed_type) { snprintf(err, errlen, "Differing number of BED fields encountered at line: %zu.  Exiting...\n", li + 1); rc = 1; break; }
        char *a = xstrndup(f[1], fl[1]), *b = xstrndup(f[2], fl[2]);
        uint32_t start = (uint32_t)atoi(a), end = (uint32_t)atoi(b);
        free(a); free(b);
        if (start == end) { --start; ++end; }                                 /* zero-length feature (bedFile.h:708-712) */
        if (start > end) { snprintf(err, errlen, "Error: malformed BED entry at line %zu. Start was greater than end. Exiting.\n", li + 1); rc = 1; break; }
        if (nf != 12 || fl[10] == 0) {                                        /* adjust_junction_ends */
            char *c = xstrndup(f[0], fl[0]);
            snprintf(err, errlen, "BED line not in BED12 format. start: %s:%u\n", c, start);
            free(c); rc = 1; break;
        }
        bed_junc j; memset(&j, 0, sizeof j);
        j.chrom = xstrndup(f[0], fl[0]); j.name = xstrndup(f[3], fl[3]); j.score = xstrndup(f[4], fl[4]); j.strand = xstrndup(f[5], fl[5]);
        j.color = xstrndup(f[8], fl[8]);
        char *nb = xstrndup(f[9], fl[9]); j.nblocks = atoi(nb); free(nb);
        const char *bs[8]; size_t bl[8];
        int nbs = tokenize(f[10], fl[10], ',', bs, bl, 8);
        char *b0 = xstrndup(bs[0], bl[0]), *b1 = nbs > 1 ? xstrndup(bs[1], bl[1]) : xstrndup("0", 1);
        j.ts = start; j.te = end;
        j.start = start + (uint32_t)atoi(b0); j.end = end - (uint32_t)(atoi(b1) - 1);
        free(b0); free(b1);
        *out = (bed_junc *)realloc(*out, (*n_out + 1) * sizeof j); (*out)[(*n_out)++] = j;
    }
    free(t.text); free(t.line_off);
    return rc;
}
static void bed_free(bed_junc *b, size_t n) { for (size_t i = 0; i < n; ++i) { free(b[i].chrom); free(b[i].name); free(b[i].score); free(b[i].strand); free(b[i].color); } free(b); }

/* get_splice_site (junctions_annotator.cc:94-114); je = AnnotatedJunction.end.  0 ok, 1 = fai_fetch failed */
static int splice_site(fasta *fa, const char *chrom, uint32_t js, uint32_t je, const char *strand, char *site, size_t cap, char *err, size_t errlen) {
    char s1[8] = "", s2[8] = "";
    int l1 = fa ? fasta_fetch(fa, chrom, (int64_t)js + 1, (int64_t)js + 2, s1, 4) : -1;
    if (l1 < 0) { snprintf(err, errlen, "Unable to extract FASTA sequence for position %s:%u-%u\n\n", chrom, js + 1, js + 2); return 1; }
    int l2 = fasta_fetch(fa, chrom, (int64_t)je - 2, (int64_t)je - 1, s2, 4);
    if (l2 < 0) { snprintf(err, errlen, "Unable to extract FASTA sequence for position %s:%u-%u\n\n", chrom, je - 2, je - 1); return 1; }
    s1[l1] = 0; s2[l2] = 0;
    if (!strcmp(strand, "-")) { orc_rev_comp(s1, l1); orc_rev_comp(s2, l2); snprintf(site, cap, "%s-%s", s2, s1); }
    else snprintf(site, cap, "%s-%s", s1, s2);
    return 0;
}

/* AnnotatedJunction::print (junctions_annotator.h:84-126) up to the transcripts column; the caller ends the line */
static void print_annotated_junction(FILE *fo, const gtf_model *gm, const char *chrom, uint32_t js, uint32_t je, const char *name, const char *score,
                                     const char *strand, const char *site) {
    jann a;
    annotate_junction(gm, chrom, js, je, strand[0] && !strand[1] ? strand[0] : '?', &a);
    fprintf(fo, "%s\t%u\t%u\t%s\t%s\t%s\t%s\t%zu\t%zu\t%zu\t%s\t%d\t%d\t%d", chrom, js, je, name, score, strand, site, a.n_acc, a.n_exo, a.n_don,
            a.anchor, a.known_donor, a.known_acceptor, a.known_junction);
    if (a.n_tx) {
        /* set< vector<string> > genes_overlap: unique (name,id) pairs in lexicographic order; set<string> transcripts */
        const gtf_tx **g = (const gtf_tx **)malloc(a.n_tx * sizeof(void *)); size_t ng = 0;
        for (size_t q = 0; q < a.n_tx; ++q) {
            int dup = 0;
            for (size_t w = 0; w < ng; ++w) if (!strcmp(g[w]->gene_name, a.txs[q]->gene_name) && !strcmp(g[w]->gene_id, a.txs[q]->gene_id)) dup = 1;
            if (!dup) g[ng++] = a.txs[q];
        }
        for (size_t x = 1; x < ng; ++x) { const gtf_tx *t = g[x]; size_t y = x; while (y > 0 && (strcmp(g[y - 1]->gene_name, t->gene_name) > 0 || (!strcmp(g[y - 1]->gene_name, t->gene_name) && strcmp(g[y - 1]->gene_id, t->gene_id) > 0))) { g[y] = g[y - 1]; --y; } g[y] = t; }
        fputc('\t', fo);
        for (size_t x = 0; x < ng; ++x) fprintf(fo, "%s%s", x ? "," : "", g[x]->gene_name);
        fputc('\t', fo);
        for (size_t x = 0; x < ng; ++x) fprintf(fo, "%s%s", x ? "," : "", g[x]->gene_id);
        free(g);
        const gtf_tx **tt = (const gtf_tx **)malloc(a.n_tx * sizeof(void *));
        memcpy(tt, a.txs, a.n_tx * sizeof(void *));
        for (size_t x = 1; x < a.n_tx; ++x) { const gtf_tx *t = tt[x]; size_t y = x; while (y > 0 && strcmp(tt[y - 1]->id, t->id) > 0) { tt[y] = tt[y - 1]; --y; } tt[y] = t; }
        fputc('\t', fo);
        for (size_t x = 0; x < a.n_tx; ++x) fprintf(fo, "%s%s", x ? "," : "", tt[x]->id);
        free(tt);
    } else fputs("\tNA\tNA\tNA", fo);
    jann_free(&a);
}

void orc_cse_default_params(orc_cse_params *p) {
    /* cis_splice_effects_identifier.h:101-117 */
    memset(p, 0, sizeof *p);
    p->intronic_min = 2; p->exonic_min = 3; p->skip_single = 1; p->strandness = -1;
    p->strand_tag[0] = 'X'; p->strand_tag[1] = 'S'; p->min_anchor = 8; p->min_intron = 70; p->max_intron = 500000;
}

int orc_identify(const orc_cse_params *p, char *err, size_t errlen) {
    gtf_model gm;
    if (gtf_load(p->gtf, &gm, err, errlen)) return 1;
    textfile vcf;
    if (text_load(p->vcf, &vcf)) { gtf_free(&gm); snprintf(err, errlen, "Unable to open file.\n\n"); return 1; }
    va_opts vo = { p->intronic_min, p->exonic_min, p->all_intronic, p->all_exonic, p->skip_single };
    FILE *fv = p->out_vcf ? fopen(p->out_vcf, "w") : NULL;
    cse_junction *cj = NULL; size_t n_cj = 0, cap_cj = 0;
    ann_variant **all_vars = NULL; size_t n_all = 0;
    int rc = 0;
    bed_junc *bed = NULL; size_t n_bed = 0;
    if (p->bed && bed_load(p->bed, &bed, &n_bed, err, errlen)) rc = 1;
    int header_done = 0;
    for (size_t li = 0; li < vcf.n_lines && !rc; ++li) {
        const char *line = vcf.text + vcf.line_off[li];
        size_t ll = vcf.line_off[li + 1] - vcf.line_off[li]; if (ll && line[ll - 1] == '\n') --ll; else if (ll) --ll;
        if (ll && line[ll - 1] == '\r') --ll;
        if (ll == 0) continue;
        if (line[0] == '#') { if (fv) vcf_header_line(fv, &vcf, li, line, ll, &header_done); continue; }
        const char *f[16]; size_t fl[16];
        int nf = tokenize(line, ll, '\t', f, fl, 16);
        if (nf < 2) continue;
        char *chrom = xstrndup(f[0], fl[0]);
        char *ps = xstrndup(f[1], fl[1]);
        uint32_t pos0 = (uint32_t)(atoi(ps) - 1);
        free(ps);
        ann_variant *v = (ann_variant *)malloc(sizeof *v);
        annotate_variant(&gm, &vo, chrom, pos0, v);
        free(chrom);
        all_vars = (ann_variant **)realloc(all_vars, (n_all + 1) * sizeof(void *)); all_vars[n_all++] = v;
        if (!v->relevant) continue;
        if (fv) vcf_record(fv, line, ll, f, fl, nf, v);
        if (p->bed) {
            /* associate (associator.cc:261-272): every BED junction of the variant's contig that starts or ends inside the cis window */
            for (size_t i = 0; i < n_bed; ++i) {
                const bed_junc *j = &bed[i];
                const uint32_t jend = j->end - 1;                       /* Junction.end (parse_BED_to_junctions :221) */
                if (strcmp(j->chrom, v->chrom)) continue;
                if (!((j->start >= v->ces && j->start <= v->cee) || (jend <= v->cee && jend >= v->ces))) continue;
                cse_junction key; key.chrom = j->chrom; key.start = j->start; key.end = jend;
                size_t q = 0;
                for (; q < n_cj; ++q) if (!cj_cmp(&cj[q], &key)) break;
                if (q == n_cj) {
                    if (n_cj == cap_cj) { cap_cj = cap_cj ? cap_cj * 2 : 64; cj = (cse_junction *)realloc(cj, cap_cj * sizeof *cj); }
                    cse_junction *n = &cj[n_cj++];
                    n->chrom = strdup(j->chrom); n->start = j->start; n->end = jend; n->ts = j->ts; n->te = j->te;
                    n->count = (uint32_t)atoi(j->score); n->strand = j->strand[0] && !j->strand[1] ? j->strand[0] : '?';
                    n->strand_s = j->strand; n->color = j->color; n->nblocks = j->nblocks; n->vars = NULL; n->n_vars = 0;
                }
                cse_junction *n = &cj[q];
                int dup = 0; for (size_t w = 0; w < n->n_vars; ++w) if (!var_cmp(&n->vars[w], &v)) dup = 1;
                if (!dup) { n->vars = (ann_variant **)realloc(n->vars, (n->n_vars + 1) * sizeof(void *)); n->vars[n->n_vars++] = v; }
            }
            continue;
        }
        /* region of the extraction (:270-274), uint32 arithmetic */
        char region[512];
        uint32_t rs = p->window ? (uint32_t)(v->start - p->window) : v->ces, re = p->window ? (uint32_t)(v->end + p->window) : v->cee;
        snprintf(region, sizeof region, "%s:%u-%u", v->chrom, rs, re);
        orc_params ep; orc_default_params(&ep);
        ep.bam = p->bam; ep.region = region; ep.strandness = p->strandness; ep.strand_tag[0] = p->strand_tag[0]; ep.strand_tag[1] = p->strand_tag[1];
        ep.min_anchor = p->min_anchor; ep.min_intron = p->min_anchor /* ctor quirk junctions_extractor.h:200 */; ep.max_intron = p->max_intron;
        ep.fasta = (p->override_motif || p->strandness == 3) ? p->fasta : NULL;
        orc_table *t = NULL;
        if (orc_extract(&ep, &t, err, errlen)) { rc = 1; break; }
        for (size_t i = 0; i < t->n; ++i) {
            const orc_junction *j = &t->rows[i];
            if (!((j->start >= v->ces && j->start <= v->cee) || (j->end <= v->cee && j->end >= v->ces))) continue;
            cse_junction key; key.chrom = t->ref_name[j->tid]; key.start = j->start; key.end = j->end;
            size_t q = 0;
            for (; q < n_cj; ++q) if (!cj_cmp(&cj[q], &key)) break;
            if (q == n_cj) {
                if (n_cj == cap_cj) { cap_cj = cap_cj ? cap_cj * 2 : 64; cj = (cse_junction *)realloc(cj, cap_cj * sizeof *cj); }
                cse_junction *n = &cj[n_cj++];
                n->chrom = strdup(key.chrom); n->start = j->start; n->end = j->end; n->ts = j->thick_start; n->te = j->thick_end;
                n->count = j->read_count; n->strand = j->strand; n->vars = NULL; n->n_vars = 0;
                n->strand_s = NULL; n->color = "255,0,0"; n->nblocks = 2;
            }
            cse_junction *n = &cj[q];
            int dup = 0; for (size_t w = 0; w < n->n_vars; ++w) if (!var_cmp(&n->vars[w], &v)) dup = 1;
            if (!dup) { n->vars = (ann_variant **)realloc(n->vars, (n->n_vars + 1) * sizeof(void *)); n->vars[n->n_vars++] = v; }
        }
        orc_table_free(t);
    }
    if (fv) fclose(fv);
    if (!rc) {
        /* annotate_junctions :222-246 */
        fasta *fa = fasta_load(p->fasta);
        FILE *fo = p->out_tsv ? fopen(p->out_tsv, "w") : stdout;
        FILE *fj = p->out_bed ? fopen(p->out_bed, "w") : NULL;
        qsort(cj, n_cj, sizeof *cj, cj_cmp);
        fputs("chrom\tstart\tend\tname\tscore\tstrand\tsplice_site\tacceptors_skipped\texons_skipped\tdonors_skipped\tanchor\tknown_donor\tknown_acceptor\tknown_junction\tgene_names\tgene_ids\ttranscripts\tvariant_info\n", fo);
        for (size_t i = 0; i < n_cj && !rc; ++i) {
            cse_junction *j = &cj[i];
            uint32_t js = j->start, je = j->end + 1;
            /* get_splice_site :94-114 */
            char s1[8] = "", s2[8] = "", site[24];
            int l1 = fa ? fasta_fetch(fa, j->chrom, (int64_t)js + 1, (int64_t)js + 2, s1, 4) : -1;
            int l2 = fa ? fasta_fetch(fa, j->chrom, (int64_t)je - 2, (int64_t)je - 1, s2, 4) : -1;
            if (l1 < 0 || l2 < 0) { snprintf(err, errlen, "Unable to extract FASTA sequence for position\n\n"); rc = 1; break; }
            s1[l1] = 0; s2[l2] = 0;
            char strand1[2] = { j->strand, 0 };
            const char *strand = j->strand_s ? j->strand_s : strand1;
            if (!strcmp(strand, "-")) { orc_rev_comp(s1, l1); orc_rev_comp(s2, l2); snprintf(site, sizeof site, "%s-%s", s2, s1); }
            else snprintf(site, sizeof site, "%s-%s", s1, s2);
            char name[32]; snprintf(name, sizeof name, "JUNC%08zu", i + 1);
            if (fj) fprintf(fj, "%s\t%u\t%u\t%s\t%u\t%s\t%u\t%u\t%s\t%d\t%u,%u\t0,%u\n", j->chrom, j->ts, j->te, name, j->count, strand, j->ts, j->te, j->color, j->nblocks,
                            (uint32_t)(j->start - j->ts), (uint32_t)(j->te - j->end), (uint32_t)(j->end - j->ts));
            char score[16]; snprintf(score, sizeof score, "%u", j->count);
            print_annotated_junction(fo, &gm, j->chrom, js, je, name, score, strand, site);
            /* variant_set_to_string (variants_annotator.h:227-235): set order, "chrom:start-end" with int fields */
            qsort(j->vars, j->n_vars, sizeof(void *), var_cmp);
            fputc('\t', fo);
            for (size_t w = 0; w < j->n_vars; ++w) fprintf(fo, "%s%s:%d-%d", w ? "," : "", j->vars[w]->chrom, (int)j->vars[w]->start, (int)j->vars[w]->end);
            fputc('\n', fo);
        }
        if (fo != stdout) fclose(fo);
        if (fj) fclose(fj);
        fasta_free(fa);
    }
    for (size_t i = 0; i < n_cj; ++i) { free(cj[i].chrom); free(cj[i].vars); }
    free(cj);
    bed_free(bed, n_bed);
    for (size_t i = 0; i < n_all; ++i) { variant_free(all_vars[i]); free(all_vars[i]); }
    free(all_vars);
    free(vcf.text); free(vcf.line_off);
    gtf_free(&gm);
    return rc;
}

/* ==================================================================================================
 * `junctions annotate` (junctions_main.cc:62-93)
 * ================================================================================================ */
int orc_junctions_annotate_opts(const char *bed, const char *fasta_path, const char *gtf, const char *out, int include_single_exon, char *err, size_t errlen) {
    g_keep_single_exon = include_single_exon;
    const int rc = orc_junctions_annotate(bed, fasta_path, gtf, out, err, errlen);
    g_keep_single_exon = 0;
    return rc;
}
int orc_junctions_annotate(const char *bed, const char *fasta_path, const char *gtf, const char *out, char *err, size_t errlen) {
    gtf_model gm;
    if (gtf_load(gtf, &gm, err, errlen)) return 1;
    FILE *fo = out ? fopen(out, "w") : stdout;
    if (!fo) { gtf_free(&gm); snprintf(err, errlen, "Unable to open %s", out); return 1; }
    fputs("chrom\tstart\tend\tname\tscore\tstrand\tsplice_site\tacceptors_skipped\texons_skipped\tdonors_skipped\tanchor\tknown_donor\tknown_acceptor\tknown_junction\tgene_names\tgene_ids\ttranscripts\n", fo);
    bed_junc *b = NULL; size_t nb = 0;
    char berr[512] = "";
    const int brc = bed_load(bed, &b, &nb, berr, sizeof berr);        /* rows before a bad line are still annotated and printed */
    fasta *fa = fasta_load(fasta_path);
    int rc = 0;
    for (size_t i = 0; i < nb && !rc; ++i) {
        char site[24];
        if (splice_site(fa, b[i].chrom, b[i].start, b[i].end, b[i].strand, site, sizeof site, err, errlen)) { rc = 1; break; }
        print_annotated_junction(fo, &gm, b[i].chrom, b[i].start, b[i].end, b[i].name, b[i].score, b[i].strand, site);
        fputc('\n', fo);
    }
    if (!rc && brc) { snprintf(err, errlen, "%s", berr); rc = 1; }
    if (fo != stdout) fclose(fo);
    fasta_free(fa); bed_free(b, nb); gtf_free(&gm);
    return rc;
}

/* ==================================================================================================
 * `variants annotate` (variants_annotator.cc:541-550): every record, the four tags appended to INFO
 * ================================================================================================ */
int orc_variants_annotate(const orc_cse_params *p, char *err, size_t errlen) {
    gtf_model gm;
    if (gtf_load(p->gtf, &gm, err, errlen)) return 1;
    textfile vcf;
    if (text_load(p->vcf, &vcf)) { gtf_free(&gm); snprintf(err, errlen, "Unable to open file.\n\n"); return 1; }
    va_opts vo = { p->intronic_min, p->exonic_min, p->all_intronic, p->all_exonic, p->skip_single };
    FILE *fv = p->out_vcf ? fopen(p->out_vcf, "w") : stdout;
    if (!fv) { gtf_free(&gm); free(vcf.text); free(vcf.line_off); snprintf(err, errlen, "Unable to open output VCF file.\n\n"); return 1; }
    int header_done = 0;
    for (size_t li = 0; li < vcf.n_lines; ++li) {
        const char *line = vcf.text + vcf.line_off[li];
        size_t ll = vcf.line_off[li + 1] - vcf.line_off[li]; if (ll) --ll;
        if (ll && line[ll - 1] == '\r') --ll;
        if (ll == 0) continue;
        if (line[0] == '#') { vcf_header_line(fv, &vcf, li, line, ll, &header_done); continue; }
        const char *f[16]; size_t fl[16];
        int nf = tokenize(line, ll, '\t', f, fl, 16);
        if (nf < 2) continue;
        char *chrom = xstrndup(f[0], fl[0]), *ps = xstrndup(f[1], fl[1]);
        ann_variant v;
        annotate_variant(&gm, &vo, chrom, (uint32_t)(atoi(ps) - 1), &v);
        vcf_record(fv, line, ll, f, fl, nf, &v);
        variant_free(&v); free(chrom); free(ps);
    }
    if (fv != stdout) fclose(fv);
    free(vcf.text); free(vcf.line_off);
    gtf_free(&gm);
    return 0;
}
