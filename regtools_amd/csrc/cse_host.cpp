// cse_host.cpp -- see cse_host.h.  Citations relative to /root/reference/src.
#include "cse_host.h"
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

// lineFileUtilities.h:24-33 Tokenize (std::getline: consecutive delimiters give empty fields, no trailing empty field)
static void tokenize(const char *s, size_t len, char delim, std::vector<std::pair<const char *, size_t>> &out) {
    out.clear();
    if (!len) return;
    size_t i = 0;
    for (;;) {
        size_t j = i;
        while (j < len && s[j] != delim) ++j;
        out.push_back({s + i, j - i});
        if (j >= len) break;
        i = j + 1;
        if (i >= len) break;
    }
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
    FileBytes file;
    if (!file.open(path, /*populate=*/true)) return "\nUnable to open GTF file.";
    const char *text = (const char *)file.data();
    const size_t text_len = file.size();
    // ---- pass 1 (threads): every line -> at most one exon record; the pieces are views into the mapped text ---------------------
    struct Rec { const char *tid; const char *attrs; const char *chrom; uint32_t tid_len, attrs_len, chrom_len, s, e; uint8_t strand; };
    struct Part { std::vector<Rec> recs; size_t err_pos = SIZE_MAX; const char *err = nullptr; };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_parts = text_len < (1u << 22) ? 1 : std::min<size_t>(hw ? hw : 4, 16);
    std::vector<size_t> cut(n_parts + 1, text_len);
    cut[0] = 0;
    for (size_t k = 1; k < n_parts; ++k) {                                   // cut points just after a newline
        size_t q = text_len * k / n_parts;
        const char *nl = (const char *)memchr(text + q, '\n', text_len - q);
        cut[k] = nl ? (size_t)(nl - text) + 1 : text_len;
    }
    std::vector<Part> parts(n_parts);
    auto scan = [&](size_t k) {
        Part &P = parts[k];
        size_t pos = cut[k];
        const size_t lim = cut[k + 1];
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
            for (size_t i = 0;;) {
                const char *t = (const char *)memchr(line + i, '\t', ll - i);
                const size_t j = t ? (size_t)(t - line) : ll;
                if (nf < 10) { fb[nf] = line + i; fl[nf] = j - i; }
                ++nf;
                if (j >= ll) break;
                i = j + 1;
                if (i >= ll) break;
            }
            if (nf != 9) { P.err_pos = line_pos; P.err = "Expected 9 fields in GTF line."; return; }   // gtf_parser.cc:67-70
            if (!(fl[2] == 4 && !memcmp(fb[2], "exon", 4))) continue;
            const char *tv; size_t tl;
            if (!gtf_attr_view(fb[8], fl[8], "transcript_id", 13, tv, tl) || (tl == 2 && !memcmp(tv, "NA", 2))) continue;   // gtf_parser.cc:118
            P.recs.push_back(Rec{tv, fb[8], fb[0], (uint32_t)tl, (uint32_t)fl[8], (uint32_t)fl[0], (uint32_t)field_atol(fb[3], fl[3]), (uint32_t)field_atol(fb[4], fl[4]),
                                 fl[6] == 1 ? (uint8_t)fb[6][0] : (uint8_t)'?'});
        }
    };
    if (n_parts == 1) scan(0);
    else {
        std::vector<std::thread> th;
        for (size_t k = 0; k < n_parts; ++k) th.emplace_back(scan, k);
        for (auto &t : th) t.join();
    }
    // ---- pass 2 (serial, file order): group exon records by transcript; the first bad line in file order ends the run, as upstream ---
    struct Tmp { std::string id, gene_name, gene_id; int32_t chrom; uint8_t strand; uint32_t n = 0; };
    std::vector<Tmp> tmp;
    std::vector<uint32_t> ex_tx, ex_s, ex_e;                                       // exon lines in file order
    const char *last_tv = nullptr; size_t last_tl = 0; uint32_t last_k = 0;        // exon lines of a transcript are usually adjacent
    struct SvHash { size_t operator()(const std::pair<const char *, size_t> &k) const { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < k.second; ++i) h = (h ^ (uint8_t)k.first[i]) * 1099511628211ull; return (size_t)h; } };
    struct SvEq { bool operator()(const std::pair<const char *, size_t> &a, const std::pair<const char *, size_t> &b) const { return a.second == b.second && !memcmp(a.first, b.first, a.second); } };
    std::unordered_map<std::pair<const char *, size_t>, uint32_t, SvHash, SvEq> by_id;     // keys are views into the mapped text
    by_id.reserve(1 << 16);
    { size_t total = 0; for (auto &P : parts) total += P.recs.size(); ex_tx.reserve(total); ex_s.reserve(total); ex_e.reserve(total); }
    for (size_t k = 0; k < n_parts; ++k) {
        const Part &P = parts[k];
        for (const Rec &r : P.recs) {
            uint32_t t;
            if (last_tv && r.tid_len == last_tl && !memcmp(r.tid, last_tv, last_tl)) t = last_k;
            else if (auto it = by_id.find({r.tid, r.tid_len}); it != by_id.end()) t = it->second;
            else {
                t = (uint32_t)tmp.size(); by_id.emplace(std::make_pair(r.tid, (size_t)r.tid_len), t);
                Tmp x; x.id.assign(r.tid, r.tid_len);
                x.gene_name = gtf_attr(r.attrs, r.attrs_len, "gene_name");          // first exon line seen wins (gtf_parser.cc:266-273)
                x.gene_id = gtf_attr(r.attrs, r.attrs_len, "gene_id");
                std::string cn(r.chrom, r.chrom_len);
                auto ci = chrom_index.find(cn);
                if (ci == chrom_index.end()) { ci = chrom_index.emplace(cn, (int32_t)chroms.size()).first; chroms.push_back(cn); }
                x.chrom = ci->second;
                x.strand = r.strand;
                tmp.push_back(std::move(x));
            }
            last_tv = r.tid; last_tl = r.tid_len; last_k = t;
            ++tmp[t].n;
            ex_tx.push_back(t); ex_s.push_back(r.s); ex_e.push_back(r.e);
        }
        if (P.err) return P.err;                                                   // everything before it was consumed, nothing after it matters
    }
    // exons grouped by transcript, file order kept inside a group (one counting pass instead of 250 k small vectors)
    std::vector<uint32_t> goff(tmp.size() + 1, 0), gs(ex_tx.size()), ge(ex_tx.size());
    for (size_t k = 0; k < tmp.size(); ++k) goff[k + 1] = goff[k] + tmp[k].n;
    { std::vector<uint32_t> fill(goff.begin(), goff.end() - 1);
      for (size_t i = 0; i < ex_tx.size(); ++i) { const uint32_t q = fill[ex_tx[i]]++; gs[q] = ex_s[i]; ge[q] = ex_e[i]; } }
    std::vector<uint32_t> order(tmp.size());
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return tmp[a].id < tmp[b].id; });   // std::map<string,Transcript>
    es.reserve(gs.size()); ee.reserve(gs.size());
    std::vector<uint32_t> idx;
    for (uint32_t k : order) {
        Tmp &t = tmp[k];
        if (t.strand != '+' && t.strand != '-') return "Undefined strand for exon ";                          // gtf_parser.cc:193-197 exit(1)
        const uint32_t *ts = gs.data() + goff[k], *te = ge.data() + goff[k];
        idx.resize(t.n);
        std::iota(idx.begin(), idx.end(), 0u);
        // sort_exons_within_transcripts: '+' ascending start, '-' descending start (stable)
        if (t.strand == '+') std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return ts[a] < ts[b]; });
        else std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return ts[a] > ts[b]; });
        tx_id.push_back(t.id); tx_gene_name.push_back(t.gene_name); tx_gene_id.push_back(t.gene_id);
        tx_chrom.push_back(t.chrom); tx_strand.push_back(t.strand);
        tx_exon_off.push_back((uint32_t)es.size()); tx_n_exons.push_back((uint32_t)idx.size());
        for (uint32_t i : idx) { es.push_back(ts[i]); ee.push_back(te[i]); }
        tx_bin.push_back(ucsc_bin(es[tx_exon_off.back()], ee.back()));                                         // gtf_parser.cc:154-160
    }
    // chr -> bin -> [transcript ids ascending]
    std::vector<uint32_t> bt(tx_id.size());
    std::iota(bt.begin(), bt.end(), 0u);
    std::stable_sort(bt.begin(), bt.end(), [&](uint32_t a, uint32_t b) {
        const uint64_t ka = (uint64_t)(uint32_t)tx_chrom[a] << 32 | tx_bin[a], kb = (uint64_t)(uint32_t)tx_chrom[b] << 32 | tx_bin[b];
        return ka < kb;
    });
    for (uint32_t t : bt) { bin_key.push_back((uint64_t)(uint32_t)tx_chrom[t] << 32 | tx_bin[t]); bin_tx.push_back(t); }
    return "";
}

// gzip / bgzip input (hts_open accepts both, hts.c:204-260): host_io's gunzip_all, i.e. the product's own decoder compiled for the host
std::string VcfText::load(const std::string &path) {
    if (!slurp(path, text)) return "Unable to open file.\n\n";
    if (text.size() >= 2 && (uint8_t)text[0] == 0x1f && (uint8_t)text[1] == 0x8b) {
        std::string plain;
        std::string e = gunzip_all((const uint8_t *)text.data(), text.size(), plain);
        if (!e.empty()) return e;
        if (plain.size() >= 3 && !memcmp(plain.data(), "BCF", 3)) return "regtools_amd: BCF input is not supported on this path\n\n";
        text.swap(plain);
    }
    size_t p = 0;
    while (p < text.size()) {
        line_off.push_back(p);
        size_t e = text.find('\n', p); if (e == std::string::npos) e = text.size();
        p = e + 1;
    }
    line_off.push_back(text.size() + (text.empty() || text.back() == '\n' ? 0 : 1));
    for (size_t i = 0; i + 1 < line_off.size(); ++i) {
        const char *l; size_t n; line(i, l, n);
        if (!n || l[0] == '#') continue;
        const char *t1 = (const char *)memchr(l, '\t', n);
        if (!t1) continue;
        const char *t2 = (const char *)memchr(t1 + 1, '\t', (size_t)(l + n - t1 - 1));
        std::string ps(t1 + 1, t2 ? (size_t)(t2 - t1 - 1) : (size_t)(l + n - t1 - 1));
        recs.push_back({i, std::string(l, (size_t)(t1 - l)), (uint32_t)(atoi(ps.c_str()) - 1)});
    }
    return "";
}

void VcfText::line(size_t i, const char *&p, size_t &len) const {
    p = text.data() + line_off[i];
    size_t end = line_off[i + 1];
    len = end - line_off[i];
    if (len) --len;                                   // the '\n' (or the virtual one after an unterminated last line)
    if (len && p[len - 1] == '\r') --len;
}

Fasta::~Fasta() { if (data && size) munmap(const_cast<char *>(data), size); }

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
