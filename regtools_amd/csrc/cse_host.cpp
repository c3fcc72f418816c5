// cse_host.cpp -- see cse_host.h.  Citations relative to /root/reference/src.
#include "cse_host.h"
#include "worker_pool.h"

#include <chrono>
#include <functional>
#include <limits.h>
#include "host_io.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <set>
#include <thread>
#include <numeric>

#include "cse_core.h"
#include "inflate_core.h"

namespace rgx {

static bool slurp(const std::string &path, std::string &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    bool ok = n == 0 || fread(&out[0], 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// gtf_parser.cc:89-104 parse_attribute + utils/common.h:85-92 unquote
static std::string gtf_attr(const char *attrs, size_t alen, const char *key) {
    const size_t klen = strlen(key);
    size_t i = 0;
    while (i < alen) {
        size_t j = i;
        while (j < alen && attrs[j] != ';') ++j;
        const char *p = attrs + i; size_t l = j - i;
        if (l && p[0] == ' ') { ++p; --l; }
        size_t a = 0; while (a < l && p[a] != ' ') ++a;
        if (l > 0 && a == klen && !memcmp(p, key, klen)) {
            size_t b = a < l ? a + 1 : l, c = b;
            while (c < l && p[c] != ' ') ++c;
            const char *v = p + b; size_t vl = c - b;
            if (vl >= 1 && v[0] == '"' && v[vl - 1] == '"') { if (vl >= 2) { ++v; vl -= 2; } else vl = 0; }
            return std::string(v, vl);
        }
        if (j >= alen) break;
        i = j + 1;
    }
    return "NA";
}

// value of `key` in a GTF attribute column as a view into the text (no allocation); found = false -> "NA" upstream
static bool gtf_attr_view(const char *attrs, size_t alen, const char *key, size_t klen, const char *&v, size_t &vl) {
    size_t i = 0;
    while (i < alen) {
        size_t j = i;
        while (j < alen && attrs[j] != ';') ++j;
        const char *p = attrs + i; size_t l = j - i;
        if (l && p[0] == ' ') { ++p; --l; }
        size_t a = 0; while (a < l && p[a] != ' ') ++a;
        if (l > 0 && a == klen && !memcmp(p, key, klen)) {
            size_t b = a < l ? a + 1 : l, c = b;
            while (c < l && p[c] != ' ') ++c;
            v = p + b; vl = c - b;
            if (vl >= 1 && v[0] == '"' && v[vl - 1] == '"') { if (vl >= 2) { ++v; vl -= 2; } else vl = 0; }
            return true;
        }
        if (j >= alen) break;
        i = j + 1;
    }
    return false;
}

// atol on a field that is not NUL-terminated (leading blanks, optional sign, digits)
static long field_atol(const char *s, size_t n) {
    size_t i = 0;
    while (i < n && (s[i] == ' ' || (s[i] >= 9 && s[i] <= 13))) ++i;
    bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; ++i; }
    unsigned long v = 0;                                 // atol == strtol: saturates (stdlib), never overflows
    const unsigned long lim = neg ? (unsigned long)LONG_MAX + 1ul : (unsigned long)LONG_MAX;
    bool sat = false;
    while (i < n && s[i] >= '0' && s[i] <= '9') {
        const unsigned long d = (unsigned long)(s[i] - '0');
        if (v > (lim - d) / 10) sat = true; else v = v * 10 + d;
        ++i;
    }
    if (sat) return neg ? LONG_MIN : LONG_MAX;
    return neg ? (long)(0ul - v) : (long)v;
}

std::string GtfModel::load(const std::string &path) {
    std::shared_ptr<FileBytes> file_h = std::make_shared<FileBytes>();
    FileBytes &file = *file_h;
    if (!file.open(path)) return "\nUnable to open GTF file.";      // (the scan threads fault the mapping in, each its own range)
    const char *text = (const char *)file.data();
    const size_t text_len = file.size();
    const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    auto lap = [&](const char *what) { if (trace) { const double t = now(); fprintf(stderr, "[rgx trace] gtf: %-22s +%8.3f ms\n", what, t - t_last); t_last = t; } };
    lap("map + populate");
    typedef std::pair<const char *, size_t> View;                                  // a piece of the mapped text
    struct SvHash { size_t operator()(const View &k) const { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < k.second; ++i) h = (h ^ (uint8_t)k.first[i]) * 1099511628211ull; return (size_t)h; } };
    struct SvEq { bool operator()(const View &a, const View &b) const { return a.second == b.second && !memcmp(a.first, b.first, a.second); } };
    // ---- pass 1 (threads): every line -> at most one exon record, filed under the part's own transcript list (first appearance order) ------
    struct LocalTx { View id, attrs, chrom; uint64_t hash; uint8_t strand; uint32_t n; uint32_t global; uint32_t fill; int32_t fill_chrom; };
    struct Part { BigVec<LocalTx> tx; std::vector<View> chroms; BigVec<uint32_t> r_tx, r_s, r_e; size_t err_pos = SIZE_MAX; const char *err = nullptr; };
    const unsigned thread_cap = 32u;
    const unsigned hw = usable_threads(thread_cap);
    // (four parts per thread, handed out as threads come free: on a host whose CPU quota is shared with the rest of the call, equal parts do not take equal time)
    size_t n_parts = text_len < (1u << 22) ? 1 : std::min<size_t>((size_t)hw * 4, text_len >> 20);
    if (const char *e = getenv("REGTOOLS_AMD_GTF_PARTS")) { const long v = atol(e); if (v >= 1 && v <= 256) n_parts = (size_t)v; }   // (tests: the threaded path on small files)
    std::vector<size_t> cut(n_parts + 1, text_len);
    cut[0] = 0;
    for (size_t k = 1; k < n_parts; ++k) {                                   // cut points just after a newline
        size_t q = text_len * k / n_parts;
        const char *nl = (const char *)memchr(text + q, '\n', text_len - q);
        cut[k] = nl ? (size_t)(nl - text) + 1 : text_len;
    }
    std::vector<Part> parts(n_parts);
    std::vector<double> scan_ms(n_parts, 0);
    auto scan = [&](size_t k) {
        Part &P = parts[k];
        const double t_scan = trace ? now() : 0;
        struct Stamp { std::vector<double> &v; size_t k; double t0; bool on; std::function<double()> clk; ~Stamp() { if (on) v[k] = clk() - t0; } } stamp{scan_ms, k, t_scan, trace, now};
        std::unordered_map<View, uint32_t, SvHash, SvEq> by_id;
        const char *last_tv = nullptr; size_t last_tl = 0; uint32_t last_k = 0;  // exon lines of a transcript are usually adjacent
        const char *last_attrs = nullptr; size_t last_plen = 0;
        size_t pos = cut[k];
        const size_t lim = cut[k + 1];
        P.r_tx.reserve((lim - pos) / 96 + 16); P.r_s.reserve((lim - pos) / 96 + 16); P.r_e.reserve((lim - pos) / 96 + 16);
        while (pos < lim) {
            const char *nl = (const char *)memchr(text + pos, '\n', lim - pos);
            const size_t e = nl ? (size_t)(nl - text) : lim;
            const char *line = text + pos; const size_t ll = e - pos;
            const size_t line_pos = pos;
            pos = e + 1;
            if (ll == 0) { P.err_pos = line_pos; P.err = "basic_string::at"; return; }          // line.at(0) throws (gtf_parser.cc:230)
            if (line[0] == '#') continue;
            // Tokenize on tabs (std::getline semantics: no empty field after a trailing tab); exactly 9 fields or the run dies
            const char *fb[10]; size_t fl[10]; size_t nf = 0;
            {
                // (the first eight fields are a few bytes each: a byte loop; the attribute column is one memchr for a tab that should not be there)
                size_t i = 0, f0 = 0;
                for (; i < ll && nf < 8; ++i) if (line[i] == '\t') { fb[nf] = line + f0; fl[nf] = i - f0; ++nf; f0 = i + 1; }
                if (nf == 8 && f0 < ll) {
                    const char *t = (const char *)memchr(line + f0, '\t', ll - f0);
                    if (!t) { fb[8] = line + f0; fl[8] = ll - f0; nf = 9; }
                    else {                                                          // a tenth field (or a trailing tab, which getline does not count)
                        nf = 9; fb[8] = line + f0; fl[8] = (size_t)(t - line) - f0;
                        if ((size_t)(t - line) + 1 < ll) nf = 10;
                    }
                } else if (nf < 8) { if (f0 < ll || nf == 0) ++nf; }              // the last field of a short line (none after a trailing tab)
            }
            if (nf != 9) { P.err_pos = line_pos; P.err = "Expected 9 fields in GTF line."; return; }   // gtf_parser.cc:67-70
            if (!(fl[2] == 4 && !memcmp(fb[2], "exon", 4))) continue;
            const char *tv; size_t tl;
            uint32_t t;
            // exon lines of a transcript are usually adjacent and start their attribute column with the same bytes: when this line's column
            // repeats the last one's up to and including the delimiter that ended the transcript_id value there, the search below would walk
            // the same bytes to the same answer
            if (last_plen && fl[8] >= last_plen && !memcmp(fb[8], last_attrs, last_plen)) { tv = last_tv; tl = last_tl; t = last_k; }
            else {
            if (!gtf_attr_view(fb[8], fl[8], "transcript_id", 13, tv, tl) || (tl == 2 && !memcmp(tv, "NA", 2))) continue;   // gtf_parser.cc:118
            {   // where the value's token ended: behind an optional closing quote; the byte there (a blank or the ';') is part of the prefix
                const char *tok_end = tv + tl;
                if (tok_end < fb[8] + fl[8] && *tok_end == '"' && tv > fb[8] && tv[-1] == '"') ++tok_end;
                last_attrs = fb[8];
                last_plen = tok_end < fb[8] + fl[8] && (*tok_end == ' ' || *tok_end == ';') ? (size_t)(tok_end - fb[8]) + 1 : 0;
            }
            if (last_tv && tl == last_tl && !memcmp(tv, last_tv, tl)) t = last_k;
            else if (auto it = by_id.find(View(tv, tl)); it != by_id.end()) t = it->second;
            else {
                t = (uint32_t)P.tx.size(); by_id.emplace(View(tv, tl), t);
                int32_t lc = -1;                                                         // the part's own contig list (the text is at hand here, not in the merge)
                for (size_t q = P.chroms.size(); q-- > 0;) if (P.chroms[q].second == fl[0] && !memcmp(P.chroms[q].first, fb[0], fl[0])) { lc = (int32_t)q; break; }
                if (lc < 0) { lc = (int32_t)P.chroms.size(); P.chroms.push_back(View(fb[0], fl[0])); }
                P.tx.push_back(LocalTx{View(tv, tl), View(fb[8], fl[8]), View(fb[0], fl[0]), 0 /* set by the merge: this entry opened its transcript */, fl[6] == 1 ? (uint8_t)fb[6][0] : (uint8_t)'?', 0, 0, 0, lc});   // first exon line seen wins (gtf_parser.cc:266-273)
            }
            }
            last_tv = tv; last_tl = tl; last_k = t;
            ++P.tx[t].n;
            P.r_tx.push_back(t); P.r_s.push_back((uint32_t)field_atol(fb[3], fl[3])); P.r_e.push_back((uint32_t)field_atol(fb[4], fl[4]));
        }
    };
    WorkerPool pool(n_parts > 1 ? hw : 1);
    auto run_parallel = [&](size_t n, const std::function<void(size_t)> &f) { pool.run(n, f); };
    run_parallel(n_parts, scan);
    if (trace) { double lo = 1e9, hi = 0, sum = 0; for (double v : scan_ms) { lo = std::min(lo, v); hi = std::max(hi, v); sum += v; } fprintf(stderr, "[rgx trace] gtf: scan threads: %zu parts, min %.3f avg %.3f max %.3f ms\n", n_parts, lo, sum / (double)n_parts, hi); }
    lap("scan (threads)");
    // the first bad line in file order ends the run, as upstream
    for (size_t k = 0; k < n_parts; ++k) if (parts[k].err) return parts[k].err;
    // ---- merge: every part's transcript list, sorted together by id (threads; equal ids keep file order = part order).  A run of equal ids is one
    //      transcript, its first member the first exon line the file has for it (whose attributes and strand win, gtf_parser.cc:266-273), its
    //      position among the runs the transcript's place in std::map<string,Transcript> order (ids compared as unsigned bytes, like std::string):
    //      the global transcript number IS that rank, so exons are filed straight into the final order ------------------------------------------
    // (the id's first sixteen bytes travel with the entry, big-endian: the sort and the grouping below rarely leave this array)
    struct Ref { uint64_t p0, p1; const char *id; uint32_t len, part, idx; };
    BigVec<Ref> refs;
    {
        std::vector<size_t> base(n_parts + 1, 0);
        for (size_t k = 0; k < n_parts; ++k) base[k + 1] = base[k] + parts[k].tx.size();
        refs.resize(base[n_parts]);
        run_parallel(n_parts, [&](size_t k) {
            for (size_t i = 0; i < parts[k].tx.size(); ++i) {
                const View &v = parts[k].tx[i].id;
                uint8_t pre[16] = {0};
                memcpy(pre, v.first, std::min<size_t>(16, v.second));
                uint64_t p0, p1; memcpy(&p0, pre, 8); memcpy(&p1, pre + 8, 8);
                refs[base[k] + i] = Ref{__builtin_bswap64(p0), __builtin_bswap64(p1), v.first, (uint32_t)v.second, (uint32_t)k, (uint32_t)i};
            }
        });
    }
    // ids as unsigned bytes, a proper prefix first (std::string's order); equal ids in part order.  The zero padding of a short id's key can only
    // tie with real NUL bytes of a longer one, and then the lengths decide the same way the bytes would.
    auto id_cmp = [](const Ref &a, const Ref &b) -> int {
        if (a.p0 != b.p0) return a.p0 < b.p0 ? -1 : 1;
        if (a.p1 != b.p1) return a.p1 < b.p1 ? -1 : 1;
        if (a.len > 16 && b.len > 16) { const int c = memcmp(a.id + 16, b.id + 16, std::min(a.len, b.len) - 16); if (c) return c; }
        return a.len < b.len ? -1 : a.len > b.len ? 1 : 0;
    };
    auto ref_less = [&](const Ref &a, const Ref &b) { const int c = id_cmp(a, b); return c ? c < 0 : a.part < b.part; };   // (one entry per part and id)
    const size_t n_refs = refs.size();
    {
        const size_t n = n_refs;
        const size_t runs = n < (1u << 15) ? 1 : std::min<size_t>(hw, 16);
        run_parallel(runs, [&](size_t r) { std::sort(refs.begin() + (long)(n * r / runs), refs.begin() + (long)(n * (r + 1) / runs), ref_less); });
        for (size_t w = 1; w < runs; w *= 2) {
            std::vector<size_t> starts;
            for (size_t r = 0; r + w < runs; r += 2 * w) starts.push_back(r);
            run_parallel(starts.size(), [&](size_t q) {
                const size_t r = starts[q];
                std::inplace_merge(refs.begin() + (long)(n * r / runs), refs.begin() + (long)(n * (r + w) / runs), refs.begin() + (long)(n * std::min(runs, r + 2 * w) / runs), ref_less);
            });
        }
    }
    lap("sort ids (threads)");
    struct Tmp { View id, attrs; int32_t chrom; uint8_t strand; uint32_t n, first_part, first_idx; };
    // rank[i] = the transcript entry i belongs to (one pass over the sorted ids), then every thread fills the transcripts that open in its range
    BigVec<uint32_t> rank(n_refs);
    {
        const size_t fp = n_refs < (1u << 12) ? 1 : hw;
        run_parallel(fp, [&](size_t b) { for (size_t i = n_refs * b / fp; i < n_refs * (b + 1) / fp; ++i) rank[i] = i && id_cmp(refs[i], refs[i - 1]) != 0; });
        uint32_t r = 0;
        for (size_t i = 0; i < n_refs; ++i) { r += rank[i]; rank[i] = r; }
    }
    lap("ranks");
    BigVec<Tmp> tmp(n_refs ? (size_t)rank[n_refs - 1] + 1 : 0);               // in final (id) order
    const size_t merge_parts = n_refs < (1u << 12) ? 1 : hw;
    run_parallel(merge_parts, [&](size_t b) {
        size_t i = n_refs * b / merge_parts;
        const size_t lim = n_refs * (b + 1) / merge_parts;
        while (i < lim && i && rank[i] == rank[i - 1]) ++i;                        // (a transcript that opened in the range before this one)
        while (i < lim) {
            LocalTx &f = parts[refs[i].part].tx[refs[i].idx];
            Tmp &x = tmp[rank[i]];
            x.id = f.id; x.attrs = f.attrs; x.chrom = -1; x.strand = f.strand; x.n = 0; x.first_part = refs[i].part; x.first_idx = refs[i].idx;
            f.hash = 1;                                                            // (the hash slot now says: this entry opened its transcript)
            size_t j = i;
            for (; j < n_refs && rank[j] == rank[i]; ++j) {
                LocalTx &l = parts[refs[j].part].tx[refs[j].idx];
                if (j != i) l.hash = 0;
                l.global = rank[i]; l.fill = x.n;                                  // this part's exons follow those of the parts before it
                x.n += l.n;
            }
            i = j;
        }
    });
    lap("group (threads)");
    // the contig table, in the order in which the file's transcripts first name a contig (file order = part order, then list order)
    {
        for (Part &P : parts) {
            std::vector<int32_t> global(P.chroms.size(), -2);                      // -2: no transcript of this part that opens here has named it yet
            for (LocalTx &l : P.tx) {
                if (!l.hash) continue;
                int32_t &gi = global[(size_t)l.fill_chrom];
                if (gi == -2) {
                    std::string cn(P.chroms[(size_t)l.fill_chrom].first, P.chroms[(size_t)l.fill_chrom].second);
                    auto ci = chrom_index.find(cn);
                    if (ci == chrom_index.end()) { ci = chrom_index.emplace(cn, (int32_t)chroms.size()).first; chroms.push_back(cn); }
                    gi = ci->second;
                }
                l.fill_chrom = gi;
            }
        }
    }
    run_parallel(merge_parts, [&](size_t b) {
        for (size_t t = tmp.size() * b / merge_parts; t < tmp.size() * (b + 1) / merge_parts; ++t) tmp[t].chrom = parts[tmp[t].first_part].tx[tmp[t].first_idx].fill_chrom;
    });
    lap("merge");
    const size_t n_tx = tmp.size();
    BigVec<uint32_t> goff(n_tx + 1, 0);
    for (size_t k = 0; k < n_tx; ++k) goff[k + 1] = goff[k] + tmp[k].n;
    // exons grouped by transcript, file order kept inside a group: every part files its own lines (threads)
    BigVec<uint32_t> gs(goff[n_tx]), ge(goff[n_tx]);
    lap("allocate exon lists");
    run_parallel(n_parts, [&](size_t k) {
        Part &P = parts[k];
        for (LocalTx &l : P.tx) l.fill += goff[l.global];
        for (size_t i = 0; i < P.r_tx.size(); ++i) { const uint32_t q = P.tx[P.r_tx[i]].fill++; gs[q] = P.r_s[i]; ge[q] = P.r_e[i]; }
    });
    lap("file exons (threads)");
    BigVec<uint32_t> order(n_tx);
    std::iota(order.begin(), order.end(), 0u);
    for (uint32_t k : order) if (tmp[k].strand != '+' && tmp[k].strand != '-') return "Undefined strand for exon ";   // gtf_parser.cc:193-197 exit(1): the first in map order
    // the transcript tables, in that order (threads over ranges of it)
    tx_id.resize(n_tx); tx_gene_name.resize(n_tx); tx_gene_id.resize(n_tx); tx_chrom.resize(n_tx); tx_strand.resize(n_tx);
    tx_exon_off.resize(n_tx); tx_n_exons.resize(n_tx); tx_bin.resize(n_tx);
    es.resize(goff[n_tx]); ee.resize(goff[n_tx]);
    { uint32_t o = 0; for (size_t i = 0; i < n_tx; ++i) { tx_exon_off[i] = o; o += tmp[order[i]].n; } }
    lap("allocate tables");
    const size_t build_parts = n_tx < (1u << 12) ? 1 : (size_t)hw * 4;
    run_parallel(build_parts, [&](size_t b) {
        std::vector<uint32_t> idx;
        for (size_t i = n_tx * b / build_parts; i < n_tx * (b + 1) / build_parts; ++i) {
            const uint32_t k = order[i];
            const Tmp &t = tmp[k];
            const uint32_t *ts = gs.data() + goff[k], *te = ge.data() + goff[k];
            idx.resize(t.n);
            std::iota(idx.begin(), idx.end(), 0u);
            // sort_exons_within_transcripts: '+' ascending start, '-' descending start (stable)
            if (t.n <= 24) {                                                       // (an insertion sort is stable and asks for no buffer)
                const bool asc = t.strand == '+';
                for (uint32_t q = 1; q < t.n; ++q) {
                    const uint32_t v = idx[q], key = ts[v];
                    uint32_t r = q;
                    while (r > 0 && (asc ? ts[idx[r - 1]] > key : ts[idx[r - 1]] < key)) { idx[r] = idx[r - 1]; --r; }
                    idx[r] = v;
                }
            } else if (t.strand == '+') std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t c) { return ts[a] < ts[c]; });
            else std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t c) { return ts[a] > ts[c]; });
            tx_id[i].assign(t.id.first, t.id.second);
            tx_gene_name[i] = gtf_attr(t.attrs.first, t.attrs.second, "gene_name");
            tx_gene_id[i] = gtf_attr(t.attrs.first, t.attrs.second, "gene_id");
            tx_chrom[i] = t.chrom; tx_strand[i] = t.strand; tx_n_exons[i] = t.n;
            const uint32_t o = tx_exon_off[i];
            for (uint32_t q = 0; q < t.n; ++q) { es[o + q] = ts[idx[q]]; ee[o + q] = te[idx[q]]; }
            tx_bin[i] = ucsc_bin(es[o], ee[o + t.n - 1]);                                                      // gtf_parser.cc:154-160
        }
    });
    lap("tables (threads)");
    // chr -> bin -> [transcript ids ascending]: a stable counting sort over (contig, bin) -- bins are < 70,200 for coordinates below 2^29
    {
        uint32_t max_bin = 0;
        for (uint32_t t = 0; t < n_tx; ++t) max_bin = std::max(max_bin, tx_bin[t]);
        const uint64_t buckets = (uint64_t)chroms.size() * ((uint64_t)max_bin + 1);
        bin_key.resize(n_tx); bin_tx.resize(n_tx);
        if (buckets <= (1u << 23)) {
            const uint64_t nb = (uint64_t)max_bin + 1;
            BigVec<uint32_t> head((size_t)buckets + 1, 0);
            for (uint32_t t = 0; t < n_tx; ++t) ++head[(size_t)((uint64_t)(uint32_t)tx_chrom[t] * nb + tx_bin[t]) + 1];
            for (size_t k = 0; k < (size_t)buckets; ++k) head[k + 1] += head[k];
            bin_start.assign(head.begin(), head.end()); bin_stride = (uint32_t)nb;     // the kernels' direct index (cse_core.h bin_range)
            for (uint32_t t = 0; t < n_tx; ++t) {
                const uint32_t q = head[(size_t)((uint64_t)(uint32_t)tx_chrom[t] * nb + tx_bin[t])]++;
                bin_key[q] = (uint64_t)(uint32_t)tx_chrom[t] << 32 | tx_bin[t]; bin_tx[q] = t;
            }
        } else {
            std::vector<std::pair<uint64_t, uint32_t>> kv(n_tx);                   // (key, transcript): plain pair order = stable by transcript
            for (uint32_t t = 0; t < n_tx; ++t) kv[t] = {(uint64_t)(uint32_t)tx_chrom[t] << 32 | tx_bin[t], t};
            std::sort(kv.begin(), kv.end());
            for (size_t i = 0; i < n_tx; ++i) { bin_key[i] = kv[i].first; bin_tx[i] = kv[i].second; }
        }
    }
    lap("bins");
    // What is left is teardown: the parts' record lists and transcript lists, the merge's arrays, the mapping of the text -- page-table work the caller
    // (identify's GTF thread, on the call's critical path once the extraction is done) need not wait for: kept with the model (cse_host.h load_scratch).
    {
        auto keep = [this](auto &obj) { typedef typename std::remove_reference<decltype(obj)>::type T; load_scratch.push_back(std::shared_ptr<void>(new T(std::move(obj)), [](void *q) { delete (T *)q; })); };
        keep(parts); keep(refs); keep(rank); keep(tmp); keep(goff); keep(gs); keep(ge); keep(order);
        load_scratch.push_back(file_h);
    }
    lap("teardown kept for later");
    return "";
}

// gzip / bgzip input (hts_open accepts both, hts.c:204-260): host_io's gunzip_all, i.e. the product's own decoder compiled for the host.
// A stream that starts with the BCF magic is a BCF file (bcf_hdr_read vcf.c:788-818, bcf_read1_core :899-926).
// bcf_hdr_append x 4 (variants_annotator.cc:136-151)
static void declare_annotation_keys(VcfDictionary &h) {
    h.declare("##INFO=<ID=genes,Number=1,Type=String,Description=\"The Variant falls in the splice region of these genes\">");
    h.declare("##INFO=<ID=transcripts,Number=1,Type=String,Description=\"The Variant falls in the splice region of these transcripts\">");
    h.declare("##INFO=<ID=distances,Number=1,Type=String,Description=\"Vector of Min(Distance from start/end of exon in the transcript.)\">");
    h.declare("##INFO=<ID=annotations,Number=1,Type=String,Description=\"Does the variant fall in exonic/intronic splicing related space in the transcript.\">");
}

std::string VcfText::load(const std::string &path, bool annotating) {
    if (!slurp(path, text)) return "Unable to open file.\n\n";
    if (text.size() >= 2 && (uint8_t)text[0] == 0x1f && (uint8_t)text[1] == 0x8b) {
        std::string plain;
        std::string e = gunzip_all((const uint8_t *)text.data(), text.size(), plain);
        if (!e.empty()) return e;
        text.swap(plain);
    }
    if (text.size() >= 3 && !memcmp(text.data(), "BCF", 3)) {
        bcf = true;
        if (text.size() < 9 || memcmp(text.data(), "BCF\2\2", 5)) return "Unable to read header.\n\n";       // "only BCFv2.2 is supported"
        uint32_t l_text; memcpy(&l_text, text.data() + 5, 4);
        if (text.size() - 9 < l_text) return "Unable to read header.\n\n";
        hdr.ingest(std::string(text.data() + 9, strnlen(text.data() + 9, l_text)));
        if (!hdr.failure().empty()) { death = hdr.failure_aborts() ? 2 : 1; return hdr.failure() + "\n"; }
        size_t o = 9 + (size_t)l_text;
        while (text.size() - o >= 32) {
            uint32_t x[4]; memcpy(x, text.data() + o, 16);
            if (x[0] < 24 || text.size() - o - 8 < (size_t)x[0] + x[1]) break;                     // a record cut short: the read fails, the loop ends
            const int32_t rid = (int32_t)x[2];
            recs.push_back({o, hdr.contig_name(rid), x[3]});
            o += 8 + (size_t)x[0] + x[1];
        }
        return "";
    }
    if (text.size() < 16 || memcmp(text.data(), "##fileformat=VCF", 16)) return "Unable to read header.\n\n";      // hts.c:248: nothing else is taken for a VCF
    line_off.reserve(text.size() / 32 + 16);
    size_t p = 0;
    while (p < text.size()) {
        line_off.push_back(p);
        const char *nl = (const char *)memchr(text.data() + p, '\n', text.size() - p);
        p = nl ? (size_t)(nl - text.data()) + 1 : text.size() + 1;
    }
    line_off.push_back(text.size() + (text.empty() || text.back() == '\n' ? 0 : 1));
    const size_t n_lines_ = line_off.size() - 1;
    // the header: every line up to the first one that does not start with "##" (vcf_hdr_read vcf.c:1242-1286; empty lines are skipped, a line
    // that does not start with '#' in there means there is no sample line)
    size_t first_rec_line = n_lines_;
    {
        std::string htxt; bool closed = false;
        for (size_t i = 0; i < n_lines_; ++i) {
            const char *l; size_t n; line(i, l, n);
            if (!n) continue;
            if (l[0] != '#') { fputs("[E::vcf_hdr_read] no sample line\n", stderr); return "Unable to read header.\n\n"; }      // vcf.c:1249-1256
            htxt.append(l, n); htxt += '\n';
            if (n < 2 || l[1] != '#') { closed = true; first_rec_line = i + 1; break; }
        }
        (void)closed;                                      // (a file that ends inside its "##" lines is a header without samples and no record, vcf.c:1247-1287)
        hdr.ingest(htxt);
        if (!hdr.failure().empty()) { death = hdr.failure_aborts() ? 2 : 1; return hdr.failure() + "\n"; }
        // a tabix index next to the file: the sequence names it lists that the header does not declare join the header as ##contig lines
        // (vcf_hdr_read, vcf.c:1289-1309) -- they are written out with it, and a record on such a contig draws no warning.  An index that
        // cannot be read is no index (tbx_index_load returns NULL).
        std::string tbi_path;
        std::vector<uint8_t> raw;
        const bool have_tbi = find_tbi(path, tbi_path);
        if (have_tbi) {                                            // hts_idx_load2 (hts.c:2046-2054), as for a BAM's index
            struct stat sm, si;
            if (!stat(path.c_str(), &sm) && !stat(tbi_path.c_str(), &si) && si.st_mtime < sm.st_mtime)
                fprintf(stderr, "Warning: The index file is older than the data file: %s\n", tbi_path.c_str());
        }
        if (have_tbi && read_file(tbi_path, raw)) {
            std::string idx;
            const uint8_t *d = raw.data(); size_t n = raw.size();
            if (n >= 2 && d[0] == 0x1f && d[1] == 0x8b) { if (gunzip_all(d, n, idx).empty()) { d = (const uint8_t *)idx.data(); n = idx.size(); } else n = 0; }
            if (n >= 36 && !memcmp(d, "TBI\1", 4)) {
                int32_t n_ref, l_nm; memcpy(&n_ref, d + 4, 4); memcpy(&l_nm, d + 32, 4);
                // the whole index must load, not only its names (tbx_index_load -> hts_idx_load_core, hts.c:1517-1567): per sequence the bins
                // (number, chunk count, chunks; a bin number listed twice is an error) and the linear index, every read complete
                bool body_ok = n_ref >= 0 && l_nm >= 0 && (size_t)l_nm <= n - 36;
                if (body_ok) {
                    size_t o = 36 + (size_t)l_nm;
                    auto rd32 = [&](uint32_t &v) { if (n - o < 4) return false; memcpy(&v, d + o, 4); o += 4; return true; };
                    for (int32_t k = 0; k < n_ref && body_ok; ++k) {
                        uint32_t nb = 0, ni = 0;
                        if (!rd32(nb)) { body_ok = false; break; }
                        std::set<uint32_t> seen;
                        for (uint32_t b = 0; b < nb && body_ok; ++b) {
                            uint32_t key = 0, nc = 0;
                            if (!rd32(key) || !seen.insert(key).second || !rd32(nc) || (n - o) / 16 < nc) { body_ok = false; break; }
                            o += (size_t)nc * 16;
                        }
                        if (!body_ok || !rd32(ni) || (n - o) / 8 < ni) { body_ok = false; break; }
                        o += (size_t)ni * 8;
                    }
                }
                if (body_ok) {
                    const char *q = (const char *)d + 36, *e = q + l_nm;
                    for (int32_t k = 0; k < n_ref && q < e; ++k) {
                        const size_t ln = strnlen(q, (size_t)(e - q));
                        const std::string name(q, ln);
                        if (!hdr.knows_contig(name)) hdr.declare("##contig=<ID=" + name + ">");
                        q += ln + 1;
                    }
                }
            }
        }
    }
    // CHROM and POS of every record line, by several threads over ranges of lines (file order kept: the ranges are concatenated in order).  Every line
    // is also read the way vcf_parse reads its NAMES (read_text_record, names_only): what it says on stderr is kept with the record's index, and the
    // read loop ends at the first record it refuses (sample columns that do not match the header, vcf.c:1551-1556, 1760-1766) -- or dies in
    // (vcf.c:1612-1613, 1638-1639).  A thread has its own copy of the header: a name the header does not declare draws its line once per FILE upstream
    // (the name joins the header there), so a later range's line for a name an earlier range has met is dropped when the ranges are put together.
    const size_t T = n_lines_ < (1u << 16) ? 1 : usable_threads(16);
    struct Range { std::vector<Rec> recs; std::vector<std::pair<size_t, std::string>> notes; bool stopped = false; size_t first_id = SIZE_MAX; std::string fatal; bool aborts = false; };
    std::vector<Range> part(T);
    auto scan = [&](size_t t) {
        Range &R = part[t];
        std::vector<Rec> &out = R.recs;
        const size_t a = std::max(first_rec_line, n_lines_ * t / T), b = n_lines_ * (t + 1) / T;
        if (b > a) out.reserve(b - a);
        VcfDictionary h = hdr;
        if (annotating) declare_annotation_keys(h);
        std::vector<std::string> said;
        h.notes_to(&said);
        VcfRecord probe;
        for (size_t i = a; i < b; ++i) {
            const char *l; size_t n; line(i, l, n);
            const ReadResult got = read_text_record(h, l, n, probe, /*names_only=*/true);
            for (std::string &m : said) R.notes.emplace_back(out.size(), std::move(m));
            said.clear();
            if (got != ReadResult::kOk) { R.stopped = true; if (got == ReadResult::kFatal) { R.fatal = h.failure(); R.aborts = h.failure_aborts(); } break; }
            // Every line behind the header is a record upstream (vcf_read, vcf.c:1958-1964: hts_getline + vcf_parse, no look at what the line
            // is): a blank line, a '#' line or any text without a tab is CHROM = the whole line with everything else left as bcf_clear1
            // left it (POS 1, no ID / REF / ALT / QUAL / FILTER / INFO)
            const char *t1 = (const char *)memchr(l, '\t', n);
            if (!t1) { out.push_back({i, std::string(l, n), 0u}); continue; }
            const char *t2 = (const char *)memchr(t1 + 1, '\t', (size_t)(l + n - t1 - 1));
            std::string ps(t1 + 1, t2 ? (size_t)(t2 - t1 - 1) : (size_t)(l + n - t1 - 1));
            if (t2 && R.first_id == SIZE_MAX) R.first_id = out.size();
            out.push_back({i, std::string(l, (size_t)(t1 - l)), (uint32_t)atoi(ps.c_str()) - 1u});
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < T; ++t) pool.emplace_back(scan, t);
        scan(0);
        for (auto &th : pool) th.join();
    }
    first_with_id = SIZE_MAX;
    {
        size_t total = 0; for (auto &R : part) total += R.recs.size();
        recs.reserve(total);
        std::set<std::string> warned;                                      // the "[W::" lines earlier ranges have drawn
        for (size_t t = 0; t < T; ++t) {
            Range &R = part[t];
            const size_t base = recs.size();
            if (first_with_id == SIZE_MAX && R.first_id != SIZE_MAX) first_with_id = base + R.first_id;
            std::vector<const std::string *> fresh;
            for (auto &m : R.notes) {
                const bool once = m.second.compare(0, 4, "[W::") == 0;
                if (once && T > 1 && warned.count(m.second)) continue;
                if (once && T > 1) fresh.push_back(&m.second);
                notes.emplace_back(base + m.first, m.second);
            }
            for (const std::string *m : fresh) warned.insert(*m);
            if (T == 1) recs.swap(R.recs); else for (auto &r : R.recs) recs.push_back(std::move(r));
            if (R.stopped) { fatal = R.fatal; fatal_aborts = R.aborts; break; }
        }
    }
    return "";
}

ReadResult VcfText::typed(size_t i, VcfDictionary &h, VcfRecord &r) const {
    if (bcf) return read_bcf_record((const uint8_t *)text.data() + recs[i].line, text.size() - recs[i].line, r) ? ReadResult::kOk : ReadResult::kRefused;
    const char *l; size_t n; line(recs[i].line, l, n);
    return read_text_record(h, l, n, r);
}

std::string write_annotated_vcf_records(FILE *fv, const VcfText &vcf, const std::vector<size_t> &todo, const std::function<VcfAnnot(size_t)> &annot, bool print_notes) {
    VcfDictionary hdr = vcf.hdr;
    declare_annotation_keys(hdr);
    { std::string h; hdr.render(h); fwrite(h.data(), 1, h.size(), fv); }
    // records are independent: ranges of them are re-serialised by several threads, each with its own copy of the dictionary (a name the
    // header does not declare joins the copy; what is printed is the name), and written in order
    const size_t T = todo.size() < 4096 ? 1 : usable_threads(16);
    std::vector<std::string> outs(T), fatal(T);
    auto work = [&](size_t t) {
        VcfDictionary h = hdr;
        h.silence();                                                       // (what reading a record says is in vcf.notes already)
        VcfRecord rec;
        std::string &o = outs[t];
        static const std::string kNA = "NA";
        for (size_t k = todo.size() * t / T; k < todo.size() * (t + 1) / T; ++k) {
            const size_t ri = todo[k];
            if (print_notes && T == 1) vcf.flush_notes(ri + 1);
            const ReadResult got = vcf.typed(ri, h, rec);
            if (got == ReadResult::kFatal) { fatal[t] = h.failure(); return; }
            if (got != ReadResult::kOk) continue;
            rec.id_seen_before = vcf.first_with_id < ri;
            const VcfAnnot a = annot(ri);
            set_info_text(h, rec, "genes", a.genes ? *a.genes : kNA);
            set_info_text(h, rec, "transcripts", a.transcripts ? *a.transcripts : kNA);
            set_info_text(h, rec, "distances", a.distances ? *a.distances : kNA);
            set_info_text(h, rec, "annotations", a.annotations ? *a.annotations : kNA);
            write_text_record(h, rec, o);
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < T; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    if (print_notes && T > 1) vcf.flush_notes(SIZE_MAX);
    for (size_t t = 0; t < T; ++t) {
        fwrite(outs[t].data(), 1, outs[t].size(), fv);
        if (!fatal[t].empty()) return fatal[t];
    }
    if (print_notes) vcf.flush_notes(SIZE_MAX);
    return vcf.fatal;                                                      // (the record behind the last one: load() has met it)
}

void VcfText::line(size_t i, const char *&p, size_t &len) const {
    p = text.data() + line_off[i];
    size_t end = line_off[i + 1];
    len = end - line_off[i];
    if (len) --len;                                   // the '\n' (or the virtual one after an unterminated last line)
    if (len > 1 && p[len - 1] == '\r') --len;          // kseq.h:143
}

Fasta::~Fasta() { unmap(); }
void Fasta::unmap() { if (data && size) munmap(const_cast<char *>(data), size); data = nullptr; size = 0; }

bool Fasta::load(const std::string &path) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return false; }
    size = (size_t)st.st_size;
    if (size) {
        void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { size = 0; return false; }
        data = static_cast<const char *>(m);
    } else close(fd);
    std::string fai;
    if (slurp(path + ".fai", fai)) {
        size_t p = 0;
        while (p < fai.size()) {
            size_t e = fai.find('\n', p); if (e == std::string::npos) e = fai.size();
            std::string line = fai.substr(p, e - p); p = e + 1;
            size_t t = line.find('\t');
            if (t == std::string::npos) continue;
            Seq s; s.name = line.substr(0, t);
            long long len, off; int lb, ll;
            if (sscanf(line.c_str() + t + 1, "%lld\t%lld\t%d\t%d", &len, &off, &lb, &ll) != 4) continue;
            s.len = len; s.offset = off; s.line_blen = lb; s.line_len = ll;
            seqs.push_back(s);
        }
    } else {
        size_t i = 0, n = size;
        while (i < n) {
            if (data[i] != '>') { while (i < n && data[i] != '\n') ++i; ++i; continue; }
            size_t j = i + 1; while (j < n && !isspace((unsigned char)data[j])) ++j;
            Seq s; s.name.assign(data + i + 1, j - i - 1);
            while (j < n && data[j] != '\n') ++j;
            ++j;
            s.offset = (int64_t)j; s.len = 0; s.line_blen = 0; s.line_len = 0;
            while (j < n && data[j] != '>') {
                size_t k = j, bases = 0;
                while (k < n && data[k] != '\n') { if (isgraph((unsigned char)data[k])) ++bases; ++k; }
                const size_t ll = k - j + (k < n ? 1 : 0);
                if (!s.line_len) { s.line_len = (int)ll; s.line_blen = (int)bases; }
                s.len += (int64_t)bases;
                j = k + 1;
            }
            seqs.push_back(s);
            i = j;
        }
    }
    return true;
}

bool Fasta::fetch(const std::string &name, int64_t beg1, int64_t end1, std::string &out) const {
    const Seq *s = nullptr;
    for (const Seq &q : seqs) if (q.name == name) s = &q;     // khash: a later duplicate replaces the earlier one
    out.clear();
    if (!s) return false;
    int64_t beg = beg1, end = end1;
    if (beg > 0) --beg;
    if (beg >= s->len) beg = s->len;
    if (end >= s->len) end = s->len;
    if (beg > end) beg = end;
    if (s->line_blen <= 0) return true;
    size_t p = (size_t)(s->offset + beg / s->line_blen * s->line_len + beg % s->line_blen);
    while (p < size && (int64_t)out.size() < end - beg) { const int c = (unsigned char)data[p++]; if (isgraph(c)) out.push_back((char)c); }
    return true;
}

static bool bed_header(const char *s, size_t n) {
    return (n >= 1 && s[0] == '#') || (n >= 7 && !memcmp(s, "browser", 7)) || (n >= 5 && !memcmp(s, "track", 5));
}

std::string BedJunctions::load(const std::string &path) {
    std::string text;
    if (!slurp(path, text)) return "Error: The requested file (" + path + ") could not be opened. Exiting!\n";
    std::vector<std::pair<size_t, size_t>> lines;                       // (offset, length without the newline)
    for (size_t p = 0; p < text.size();) {
        size_t e = text.find('\n', p); if (e == std::string::npos) e = text.size();
        lines.push_back({p, e - p});
        p = e + 1;
    }
    size_t li = 0;
    while (li < lines.size() && bed_header(text.data() + lines[li].first, lines[li].second)) ++li;    // GetHeader
    size_t bed_type = 0;
    for (bool first = true; li < lines.size(); ++li, first = false) {
        const char *l = text.data() + lines[li].first; size_t ll = lines[li].second;
        if (ll && l[ll - 1] == '\r') --ll;
        std::vector<std::pair<const char *, size_t>> f;                 // std::getline-style split: no empty field after a trailing tab
        for (size_t i = 0; i < ll;) { size_t j = i; while (j < ll && l[j] != '\t') ++j; f.push_back({l + i, j - i}); i = j + 1; }
        if (first) bed_type = f.size();
        if (f.empty() || bed_header(f[0].first, f[0].second)) break;    // BED_BLANK / BED_HEADER: get_single_junction() stops
        const std::string line_no = std::to_string(li + 1);
        if (f.size() < 3) return "It looks as though you have less than 3 columns at line: " + line_no + ".  Are you sure your files are tab-delimited?\n";
        auto digits = [](const std::pair<const char *, size_t> &x) { for (size_t i = 0; i < x.second; ++i) if (x.first[i] < '0' || x.first[i] > '9') return false; return true; };
        if (!digits(f[1]) || !digits(f[2])) return "Unexpected file format.  Please use tab-delimited BED, GFF, or VCF.\n";
        if (f.size() != bed_type) return "Differing number of BED fields encountered at line: " + line_no + ".  Exiting...\n";
        auto str = [&](size_t k) { return std::string(f[k].first, f[k].second); };
        uint32_t s0 = (uint32_t)atoi(str(1).c_str()), e0 = (uint32_t)atoi(str(2).c_str());
        if (s0 == e0) { --s0; ++e0; }                                   // zero-length feature (bedFile.h:708-712)
        if (s0 > e0) return "Error: malformed BED entry at line " + line_no + ". Start was greater than end. Exiting.\n";
        if (f.size() != 12 || f[10].second == 0) return "BED line not in BED12 format. start: " + str(0) + ":" + std::to_string(s0) + "\n";
        const std::string bs = str(10);
        const size_t comma = bs.find(',');
        const int b0 = atoi(bs.c_str()), b1 = comma == std::string::npos ? 0 : atoi(bs.c_str() + comma + 1);
        chrom.push_back(str(0)); name.push_back(str(3)); score.push_back(str(4)); strand.push_back(str(5)); color.push_back(str(8));
        nblocks.push_back(atoi(str(9).c_str()));
        ts.push_back(s0); te.push_back(e0);
        start.push_back(s0 + (uint32_t)b0); end.push_back(e0 - (uint32_t)(b1 - 1));
    }
    return "";
}

std::string rev_comp(const std::string &s) {
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
    return r;
}

}  // namespace rgx
