// vcf_model.cpp -- see vcf_model.h.  Host code only; the line numbers cite /root/reference/src/utils/htslib/vcf.c.
#include "vcf_model.h"

#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace rgx {

static const uint32_t kFloatMissing = 0x7F800001u, kFloatVectorEnd = 0x7F800002u;
static const int32_t kI32Missing = INT32_MIN, kI32End = INT32_MIN + 1;
static const int kTypeShift[16] = {0, 0, 1, 2, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// =====================================================================================================
// header
// =====================================================================================================
static bool escaped(const char *min, const char *s) {           // an odd run of backslashes in front of s
    int n = 0;
    while (--s >= min && *s == '\\') ++n;
    return n % 2;
}

// vcf.c:262-342.  Works on the whole header text (a key runs to the next '=' wherever that is, as upstream's does).
bool VcfHdr::parse_line(const char *line, size_t &len, Line &out) const {
    out = Line();
    const char *p = line;
    if (p[0] != '#' || p[1] != '#') { len = 0; return false; }
    p += 2;
    const char *q = p;
    while (*q && *q != '=') ++q;
    if (*q != '=' || q == p) { len = (size_t)(q - line) + 1; return false; }
    out.key.assign(p, (size_t)(q - p));
    p = ++q;
    if (*p != '<') {                                            // ##key=value
        while (*q && *q != '\n') ++q;
        out.value.assign(p, (size_t)(q - p));
        len = (size_t)(q - line) + 1;
        return true;
    }
    out.structured = true;
    int open = 1;
    while (*q && *q != '\n' && open) {
        p = ++q;                                                // past '<' or ','
        if (*q && (isalpha((unsigned char)*q) || *q == '_')) { ++q; while (*q && (isalnum((unsigned char)*q) || *q == '_' || *q == '.')) ++q; }
        if (*q != '=' || q == p) {
            while (*q && *q != '\n') ++q;
            fprintf(stderr, "Could not parse the header line: \"%.*s\"\n", (int)(q - line), line);
            len = (size_t)(q - line) + 1;
            return false;
        }
        std::string key(p, (size_t)(q - p));
        p = ++q;
        const bool quoted = *p == '"';
        if (quoted) { ++p; ++q; }
        for (; *q; ++q) {
            if (quoted) { if (*q == '"' && !escaped(p, q)) break; }
            else {
                if (*q == '<') ++open;
                if (*q == '>') --open;
                if (!open) break;
                if (*q == ',' && open == 1) break;
            }
        }
        std::string val(p, (size_t)(q - p));
        out.kv.emplace_back(std::move(key), quoted ? "\"" + val + "\"" : val);
        if (quoted && *q) ++q;
        if (*q == '>') { --open; ++q; }
    }
    while (*q == ' ') ++q;
    len = (size_t)(q - line) + 1;
    return true;
}

// bcf_hdr_set_idx (vcf.c:344-364)
bool VcfHdr::set_idx(std::vector<std::string> &names, int &id, const std::string &tag) {
    if (id == -1) id = (int)names.size();
    else if (id < (int)names.size() && !names[(size_t)id].empty()) { error = "Conflicting IDX=" + std::to_string(id) + " lines in the header dictionary, the new tag is " + tag; return false; }
    if (id >= (int)names.size()) names.resize((size_t)id + 1);
    names[(size_t)id] = tag;
    return true;
}

static int find_key_nocase(const VcfHdr::Line &l, const char *k) {
    for (size_t i = 0; i < l.kv.size(); ++i) if (!strcasecmp(k, l.kv[i].first.c_str())) return (int)i;
    return -1;
}

// bcf_hdr_register_hrec (vcf.c:366-488): 1 = the dictionaries changed, 0 = nothing registered
int VcfHdr::register_line(Line &l) {
    if (l.key == "contig") {
        l.type = HL_CTG;
        int i = find_key_nocase(l, "length"), dummy;
        if (i >= 0 && sscanf(l.kv[(size_t)i].second.c_str(), "%d", &dummy) != 1) return 0;
        i = find_key_nocase(l, "ID");
        if (i < 0) return 0;
        const std::string name = l.kv[(size_t)i].second;
        if (contigs.count(name)) return 0;
        int idx = find_key_nocase(l, "IDX");
        const bool had_idx = idx != -1;
        if (had_idx) {
            char *e; idx = (int)strtol(l.kv[(size_t)idx].second.c_str(), &e, 10);
            if (*e) return 0;
        }
        if (!set_idx(contig_name, idx, name)) return 0;
        contigs[name] = idx;
        if (!had_idx) l.kv.emplace_back("IDX", std::to_string(idx));
        return 1;
    }
    if (l.key == "INFO") l.type = HL_INFO;
    else if (l.key == "FILTER") l.type = HL_FLT;
    else if (l.key == "FORMAT") l.type = HL_FMT;
    else if (!l.kv.empty()) { l.type = HL_STR; return 1; }
    else return 0;
    const std::string *id = nullptr;
    int vtype = -1, idx = -1;
    for (auto &kv : l.kv) {
        if (kv.first == "ID") id = &kv.second;
        else if (kv.first == "IDX") { char *e; idx = (int)strtol(kv.second.c_str(), &e, 10); if (*e) return 0; }
        else if (kv.first == "Type") {
            if (kv.second == "Integer") vtype = HT_INT;
            else if (kv.second == "Float") vtype = HT_REAL;
            else if (kv.second == "Flag") vtype = HT_FLAG;
            else vtype = HT_STR;                                // String, Character, anything else ("assuming String")
        }
    }
    if (!id) return 0;
    const std::string name = *id;
    auto it = tags.find(name);
    if (it != tags.end()) {
        if (it->second.has[l.type]) return 0;                   // declared before: the later line is dropped
        it->second.has[l.type] = true; it->second.vtype[l.type] = vtype;
        if (idx == -1) l.kv.emplace_back("IDX", std::to_string(it->second.id));
        return 1;
    }
    const bool had_idx = idx != -1;
    if (!set_idx(tag_name, idx, name)) return 0;
    Tag t; t.id = idx; t.has[l.type] = true; t.vtype[l.type] = vtype;
    tags[name] = t;
    if (!had_idx) l.kv.emplace_back("IDX", std::to_string(idx));
    return 1;
}

// bcf_hdr_add_hrec (vcf.c:490-525)
int VcfHdr::add(Line &&l) {
    l.type = HL_GEN;
    if (!register_line(l)) {
        if (l.type != HL_GEN) return 0;
        for (const Line &o : lines) {
            if (o.type != HL_GEN || o.key != l.key) continue;
            if (l.key == "fileformat" || o.value == l.value) return 0;
        }
    }
    const bool gen = l.type == HL_GEN;
    lines.push_back(std::move(l));
    return gen ? 0 : 1;
}

bool VcfHdr::append(const std::string &line) {
    Line l; size_t len;
    const std::string z = line + std::string(2, '\0');
    if (!parse_line(z.c_str(), len, l)) return false;
    add(std::move(l));
    return true;
}

// bcf_hdr_parse (vcf.c:588-613) + bcf_hdr_parse_sample_line (:94-114)
void VcfHdr::parse(const std::string &text_in) {
    const std::string text = text_in + std::string(4, '\0');
    const char *p = text.c_str();
    Line l; size_t len;
    const bool first = parse_line(p, len, l);
    if (!first || strcasecmp(l.key.c_str(), "fileformat")) fprintf(stderr, "[W::bcf_hdr_parse] The first line should be ##fileformat; is the VCF/BCF header broken?\n");
    if (first) add(std::move(l));
    append("##FILTER=<ID=PASS,Description=\"All filters passed\">");          // PASS is always entry 0 of the dictionary, line 2 of the output
    while (p < text.c_str() + text_in.size() && parse_line(p, len, l)) { add(std::move(l)); p += len; }
    if (p > text.c_str() + text_in.size()) p = text.c_str() + text_in.size();
    int field = 0;
    for (const char *a = p, *q = p;; ++q) {
        if (*q != '\t' && *q != 0 && *q != '\n') continue;
        if (++field > 9) {
            std::string s(a, (size_t)(q - a));
            if (s.empty()) { error = "Empty sample name: trailing spaces/tabs in the header line?"; return; }
            for (auto &o : samples) if (o == s) { error = "Duplicated sample name '" + s + "'"; return; }
            samples.push_back(std::move(s));
        }
        if (*q == 0 || *q == '\n') break;
        a = q + 1;
    }
}

// bcf_hdr_fmt_text (vcf.c:1334-1376), text flavour: the IDX keys stay inside
void VcfHdr::format(std::string &out) const {
    for (const Line &l : lines) {
        out += "##"; out += l.key; out += '=';
        if (!l.structured) { out += l.value; out += '\n'; continue; }
        out += '<';
        bool any = false;
        for (auto &kv : l.kv) {
            if (kv.first == "IDX") continue;
            if (any) out += ',';
            out += kv.first; out += '='; out += kv.second;
            any = true;
        }
        out += ">\n";
    }
    out += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
    if (!samples.empty()) { out += "\tFORMAT"; for (auto &s : samples) { out += '\t'; out += s; } }
    out += '\n';
}

int VcfHdr::contig_or_add(const std::string &name) {            // vcf.c:1797-1814
    auto it = contigs.find(name);
    if (it != contigs.end()) return it->second;
    fprintf(stderr, "[W::vcf_parse] contig '%s' is not defined in the header. (Quick workaround: index the file with tabix.)\n", name.c_str());
    append("##contig=<ID=" + name + ">");
    it = contigs.find(name);
    return it == contigs.end() ? -1 : it->second;
}

const VcfHdr::Tag &VcfHdr::tag_or_add(const std::string &name, int hl) {    // vcf.c:1852-1866 (FILTER), 1889-1901 (INFO), 1562-1573 (FORMAT)
    static const Tag none;
    auto it = tags.find(name);
    if (hl == HL_FLT ? it != tags.end() : (it != tags.end() && it->second.has[hl])) return it->second;
    if (hl == HL_FLT) {
        fprintf(stderr, "[W::vcf_parse] FILTER '%s' is not defined in the header\n", name.c_str());
        append("##FILTER=<ID=" + name + ",Description=\"Dummy\">");
    } else {
        fprintf(stderr, "[W::%s] %s '%s' is not defined in the header, assuming Type=String\n", hl == HL_INFO ? "vcf_parse" : "_vcf_parse_format", hl == HL_INFO ? "INFO" : "FORMAT", name.c_str());
        append(std::string(hl == HL_INFO ? "##INFO=<ID=" : "##FORMAT=<ID=") + name + ",Number=1,Type=String,Description=\"Dummy\">");
    }
    it = tags.find(name);
    return it == tags.end() ? none : it->second;
}

// =====================================================================================================
// typed values
// =====================================================================================================
static void put_raw(std::string &s, const void *p, size_t n) { s.append((const char *)p, n); }

static VcfRec::Typed enc_int1(int32_t x) {                      // bcf_enc_int1 (vcf.h:852-873)
    VcfRec::Typed t; t.n = 1;
    if (x == kI32End) { t.type = VT_INT8; t.data.push_back((char)(INT8_MIN + 1)); }
    else if (x == kI32Missing) { t.type = VT_INT8; t.data.push_back((char)INT8_MIN); }
    else if (x <= INT8_MAX && x > INT8_MIN) { t.type = VT_INT8; t.data.push_back((char)x); }
    else if (x <= INT16_MAX && x > INT16_MIN) { t.type = VT_INT16; const int16_t z = (int16_t)x; put_raw(t.data, &z, 2); }
    else { t.type = VT_INT32; put_raw(t.data, &x, 4); }
    return t;
}

// bcf_enc_vint (vcf.c:1416-1453): the narrowest type that holds every value; per = values per sample (FORMAT) or all of them (INFO)
static VcfRec::Typed enc_vint(const std::vector<int32_t> &a, int per) {
    VcfRec::Typed t;
    const int n = (int)a.size();
    if (n == 0) return t;
    if (n == 1) return enc_int1(a[0]);
    int32_t mx = INT32_MIN + 1, mn = INT32_MAX;
    for (int32_t v : a) { if (v == kI32Missing || v == kI32End) continue; if (mx < v) mx = v; if (mn > v) mn = v; }
    t.n = per <= 0 ? n : per;
    if (mx <= INT8_MAX && mn > INT8_MIN + 1) {
        t.type = VT_INT8;
        for (int32_t v : a) t.data.push_back(v == kI32End ? (char)(INT8_MIN + 1) : v == kI32Missing ? (char)INT8_MIN : (char)v);
    } else if (mx <= INT16_MAX && mn > INT16_MIN + 1) {
        t.type = VT_INT16;
        for (int32_t v : a) { const int16_t x = v == kI32End ? (int16_t)(INT16_MIN + 1) : v == kI32Missing ? (int16_t)INT16_MIN : (int16_t)v; put_raw(t.data, &x, 2); }
    } else {
        t.type = VT_INT32;
        for (int32_t v : a) put_raw(t.data, &v, 4);
    }
    return t;
}

static VcfRec::Typed enc_chars(const char *p, size_t n) { VcfRec::Typed t; t.type = VT_CHAR; t.n = (int)n; t.data.assign(p, n); return t; }

static void put_int(std::string &s, int v) { char b[16]; snprintf(b, sizeof b, "%d", v); s += b; }
static void put_float(std::string &s, float f) { char b[48]; snprintf(b, sizeof b, "%g", (double)f); s += b; }

// bcf_fmt_array (vcf.c:1467-1504): n values of `type` at p
static void fmt_array(std::string &s, int n, int type, const uint8_t *p) {
    if (n == 0) { s += '.'; return; }
    if (type == VT_CHAR) {
        for (int j = 0; j < n && p[j]; ++j) s += p[j] == 0x07 ? '.' : (char)p[j];
        return;
    }
    for (int j = 0; j < n; ++j) {
        bool missing = false, end = false; int32_t iv = 0; float fv = 0;
        if (type == VT_INT8) { const int8_t v = (int8_t)p[j]; missing = v == INT8_MIN; end = v == INT8_MIN + 1; iv = v; }
        else if (type == VT_INT16) { int16_t v; memcpy(&v, p + 2 * j, 2); missing = v == INT16_MIN; end = v == INT16_MIN + 1; iv = v; }
        else if (type == VT_INT32) { int32_t v; memcpy(&v, p + 4 * j, 4); missing = v == kI32Missing; end = v == kI32End; iv = v; }
        else if (type == VT_FLOAT) { uint32_t u; memcpy(&u, p + 4 * j, 4); missing = u == kFloatMissing; end = u == kFloatVectorEnd; memcpy(&fv, &u, 4); }
        else return;                                            // (upstream: "todo: type" and exit)
        if (end) break;
        if (j) s += ',';
        if (missing) s += '.';
        else if (type == VT_FLOAT) put_float(s, fv);
        else put_int(s, iv);
    }
}

// =====================================================================================================
// text record -> typed record
// =====================================================================================================
namespace {
struct FmtAux { int key, vtype; bool is_gt; int max_m = 0, max_l = 0, max_g = 0, size = 0; };
}

// _vcf_parse_format (vcf.c:1535-1780).  b = the line as a NUL-terminated mutable buffer, [p, q) the FORMAT column, end = b's end.
static int parse_format(VcfHdr &h, VcfRec &v, char *p, char *q, char *end) {
    const int n_hdr_samples = (int)h.samples.size();
    if (!n_hdr_samples) return 0;
    if (q >= end) { fprintf(stderr, "[vcf_parse] Error: FORMAT column with no sample columns\n"); return -1; }
    std::vector<FmtAux> fmt;
    for (char *t = p;;) {                                       // the keys, ':' separated (empty ones included)
        char *e = t; while (*e && *e != ':') ++e;
        const std::string name(t, (size_t)(e - t));
        const VcfHdr::Tag &tag = h.tag_or_add(name, HL_FMT);
        FmtAux f; f.key = tag.id; f.is_gt = name == "GT"; f.vtype = tag.vtype[HL_FMT];
        fmt.push_back(f);
        if (!*e) break;
        t = e + 1;
    }
    const int n_fmt = (int)fmt.size();
    // widths: values per field, characters per field, alleles per genotype, over the samples
    char *r = q + 1;
    int m = 1, l = 1, g = 1;
    v.n_sample = 0;
    while (r < end) {
        int j = 0;
        for (;;) {
            if (*r == '\t') *r = 0;
            if (*r == ':' || !*r) {
                if (fmt[(size_t)j].max_m < m) fmt[(size_t)j].max_m = m;
                if (fmt[(size_t)j].max_l < l - 1) fmt[(size_t)j].max_l = l - 1;
                if (fmt[(size_t)j].is_gt && fmt[(size_t)j].max_g < g) fmt[(size_t)j].max_g = g;
                l = 0; m = g = 1;
                if (*r == ':') { if (++j >= n_fmt) { h.error = "Incorrect number of FORMAT fields"; return -1; } }
                else break;
            } else if (*r == ',') ++m;
            else if (fmt[(size_t)j].is_gt && (*r == '|' || *r == '/')) ++g;
            if (r >= end) break;
            ++r; ++l;
        }
        ++v.n_sample;
        if (v.n_sample == n_hdr_samples) break;
        ++r;
    }
    // One block for all fields, each field's samples back to back, fields 8-byte aligned -- as upstream lays them out: its width count
    // misses a character for the FIRST field of every sample but the first (l restarts at 0 there, at 1 elsewhere), so a longer string
    // in a later sample runs over into the next field's bytes; keeping the layout keeps what is printed then (writes past the block's
    // end are dropped here).
    std::string mem;
    std::vector<size_t> off(fmt.size());
    for (size_t j = 0; j < fmt.size(); ++j) {
        FmtAux &f = fmt[j];
        if (!f.max_m) f.max_m = 1;
        if (f.vtype == HT_STR) f.size = f.is_gt ? f.max_g << 2 : f.max_l;
        else if (f.vtype == HT_REAL || f.vtype == HT_INT) f.size = f.max_m << 2;
        else { h.error = "the format type is currently not supported"; return -1; }       // (upstream aborts on Flag in FORMAT)
        mem.resize((mem.size() + 7) & ~(size_t)7, '\0');
        off[j] = mem.size();
        mem.resize(mem.size() + (size_t)v.n_sample * (size_t)f.size, '\0');
    }
    auto put8 = [&](size_t o, char c) { if (o < mem.size()) mem[o] = c; };
    auto put32 = [&](size_t o, uint32_t x) { if (o + 4 <= mem.size()) memcpy(&mem[o], &x, 4); };
    auto fill_missing = [&](size_t j, int sm) {
        const FmtAux &z = fmt[j];
        const size_t o = off[j] + (size_t)z.size * (size_t)sm;
        if (z.vtype == HT_STR && !z.is_gt) { if (z.size) put8(o, '.'); for (int k = 1; k < z.size; ++k) put8(o + (size_t)k, 0); return; }
        if (z.vtype == HT_REAL) { put32(o, kFloatMissing); for (int k = 1; k < z.size >> 2; ++k) put32(o + 4 * (size_t)k, kFloatVectorEnd); return; }
        put32(o, (uint32_t)kI32Missing); for (int k = 1; k < z.size >> 2; ++k) put32(o + 4 * (size_t)k, (uint32_t)kI32End);
    };
    char *t = q + 1;
    int sm = 0;
    while (t < end) {
        if (sm == n_hdr_samples) break;
        size_t j = 0;
        while (*t) {
            const FmtAux &z = fmt[j];
            const size_t o = off[j] + (size_t)z.size * (size_t)sm;
            const int cap = z.size >> 2;
            int k = 0;
            if (z.vtype == HT_STR && z.is_gt) {
                uint32_t phased = 0;
                for (;; ++t) {
                    if (*t == '.') { ++t; put32(o + 4 * (size_t)k++, phased); }
                    else put32(o + 4 * (size_t)k++, (uint32_t)(((uint64_t)(strtol(t, &t, 10) + 1) << 1) | phased));
                    phased = *t == '|';
                    if (*t == ':' || *t == 0) break;
                }
                for (; k < cap; ++k) put32(o + 4 * (size_t)k, (uint32_t)kI32End);
            } else if (z.vtype == HT_STR) {
                for (; *t != ':' && *t; ++t) put8(o + (size_t)k++, *t);
                for (; k < z.size; ++k) put8(o + (size_t)k, 0);
            } else if (z.vtype == HT_INT) {
                for (;; ++t) {
                    if (*t == '.') { put32(o + 4 * (size_t)k++, (uint32_t)kI32Missing); ++t; }
                    else put32(o + 4 * (size_t)k++, (uint32_t)(int32_t)strtol(t, &t, 10));
                    if (*t == ':' || *t == 0) break;
                }
                for (; k < cap; ++k) put32(o + 4 * (size_t)k, (uint32_t)kI32End);
            } else {
                for (;; ++t) {
                    if (*t == '.' && !isdigit((unsigned char)t[1])) { put32(o + 4 * (size_t)k++, kFloatMissing); ++t; }
                    else { const float f = (float)strtod(t, &t); uint32_t bits; memcpy(&bits, &f, 4); put32(o + 4 * (size_t)k++, bits); }
                    if (*t == ':' || *t == 0) break;
                }
                for (; k < cap; ++k) put32(o + 4 * (size_t)k, kFloatVectorEnd);
            }
            if (*t == 0) { for (++j; j < fmt.size(); ++j) fill_missing(j, sm); break; }
            if (*t == ':') { if (j + 1 < fmt.size()) ++j; }
            ++t;
        }
        ++sm; ++t;
    }
    for (size_t j = 0; j < fmt.size(); ++j) {
        const FmtAux &z = fmt[j];
        const std::string bytes = mem.substr(off[j], (size_t)v.n_sample * (size_t)z.size);
        VcfRec::Fmt f; f.key = z.key;
        if (z.vtype == HT_STR && !z.is_gt) { f.v.type = VT_CHAR; f.v.n = z.size; f.v.data = bytes; }
        else if (z.vtype == HT_INT || z.is_gt) {
            std::vector<int32_t> a((size_t)(z.size >> 2) * (size_t)v.n_sample);
            if (!a.empty()) memcpy(a.data(), bytes.data(), a.size() * 4);
            f.v = enc_vint(a, z.size >> 2);
        } else { f.v.type = VT_FLOAT; f.v.n = z.size >> 2; f.v.data = bytes; }
        v.fmt.push_back(std::move(f));
    }
    if (v.n_sample != n_hdr_samples) {
        fprintf(stderr, "[vcf_parse] Number of columns does not match the number of samples (%d vs %d).\n", v.n_sample, n_hdr_samples);
        return -1;
    }
    return 0;
}

// vcf_parse (vcf.c:1782-1956)
int vcf_parse_line(VcfHdr &h, const char *line, size_t len, VcfRec &v) {
    v = VcfRec();
    std::string buf(line, len);
    buf.append(2, '\0');
    char *b = &buf[0], *end = b + len;
    int i = 0;
    for (char *p = b; p <= end; ++i) {
        char *q = p; while (q < end && *q != '\t') ++q;
        *q = 0;
        const bool dot = !strcmp(p, ".");
        if (i == 0) v.rid = h.contig_or_add(p);
        else if (i == 1) v.pos = (int32_t)((uint32_t)atoi(p) - 1u);
        else if (i == 2) { v.have_shared = true; v.id = dot ? enc_chars(p, 0) : enc_chars(p, (size_t)(q - p)); }
        else if (i == 3) v.alleles.push_back(enc_chars(p, (size_t)(q - p)));
        else if (i == 4) {
            if (!dot) for (char *t = p, *r = p;; ++r) { if (*r == ',' || *r == 0) { v.alleles.push_back(enc_chars(t, (size_t)(r - t))); t = r + 1; } if (r == q) break; }
        } else if (i == 5) { if (!dot) { const float f = (float)atof(p); memcpy(&v.qual_bits, &f, 4); } }
        else if (i == 6) {
            if (!dot) {
                if (q > p && q[-1] == ';') q[-1] = 0;
                for (char *t = p;;) {
                    char *e = t; while (*e && *e != ';') ++e;
                    const bool last = !*e;
                    *e = 0;
                    v.flt.push_back(h.tag_or_add(t, HL_FLT).id);
                    if (last) break;
                    t = e + 1;
                }
            }
        } else if (i == 7) {
            if (!dot) {
                if (q > p && q[-1] == ';') q[-1] = 0;
                char *key = p;
                for (char *r = p;; ++r) {
                    if (*r != ';' && *r != '=' && *r != 0) continue;
                    char *val = nullptr, *ve = r;
                    int c = *r; *r = 0;
                    if (c == '=') { val = r + 1; for (ve = val; *ve != ';' && *ve != 0; ++ve) {} c = *ve; *ve = 0; }
                    if (!*key) { if (c == 0) break; r = ve; key = r + 1; continue; }          // ";;"
                    const VcfHdr::Tag &tag = h.tag_or_add(key, HL_INFO);
                    const int y = tag.vtype[HL_INFO];
                    VcfRec::Info inf; inf.key = tag.id;
                    if (!val) { /* a flag: no value */ }
                    else if (y == HT_FLAG || y == HT_STR) inf.v = enc_chars(val, (size_t)(ve - val));
                    else if (y == HT_INT || y == HT_REAL) {
                        int n_val = 1;
                        for (char *t = val; *t; ++t) if (*t == ',') ++n_val;
                        if (y == HT_INT) {
                            std::vector<int32_t> z((size_t)n_val);
                            char *t = val, *te;
                            for (int k = 0; k < n_val; ++k, ++t) {
                                z[(size_t)k] = (int32_t)strtol(t, &te, 10);
                                if (te == t) { z[(size_t)k] = kI32Missing; while (*te && *te != ',') ++te; }
                                t = te;
                                if (!*t) { for (++k; k < n_val; ++k) z[(size_t)k] = kI32Missing; break; }       // (upstream would read past the value's end here)
                            }
                            inf.v = enc_vint(z, -1);
                        } else {
                            inf.v.type = VT_FLOAT; inf.v.n = n_val;
                            char *t = val, *te;
                            for (int k = 0; k < n_val; ++k, ++t) {
                                float f = (float)strtod(t, &te);
                                uint32_t bits; memcpy(&bits, &f, 4);
                                if (te == t) { bits = kFloatMissing; while (*te && *te != ',') ++te; }
                                put_raw(inf.v.data, &bits, 4);
                                t = te;
                                if (!*t) { for (++k; k < n_val; ++k) put_raw(inf.v.data, &kFloatMissing, 4); break; }
                            }
                        }
                    }
                    v.info.push_back(std::move(inf));
                    if (c == 0) break;
                    r = ve; key = r + 1;
                }
            }
        } else if (i == 8) return parse_format(h, v, p, q, end);
        p = q + 1;
    }
    return 0;
}

// =====================================================================================================
// BCF record -> typed record (bcf_read1_core vcf.c:899-926, bcf_unpack :2000-2066)
// =====================================================================================================
namespace {
struct Cur {
    const uint8_t *p, *e; bool bad = false;
    bool need(size_t n) { if ((size_t)(e - p) < n) { bad = true; return false; } return true; }
    int32_t int_of(int type) {
        if (type == VT_INT8) { if (!need(1)) return 0; return (int8_t)*p++; }
        if (type == VT_INT16) { if (!need(2)) return 0; int16_t v; memcpy(&v, p, 2); p += 2; return v; }
        if (!need(4)) return 0;
        int32_t v; memcpy(&v, p, 4); p += 4; return v;
    }
    int32_t typed_int() { if (!need(1)) return 0; const int t = *p & 0xf; ++p; return int_of(t); }
    int32_t size(int &type) { if (!need(1)) { type = 0; return 0; } type = *p & 0xf; if ((*p >> 4) != 15) return *p++ >> 4; ++p; return typed_int(); }
    VcfRec::Typed typed(int per_mult = 1) {
        VcfRec::Typed t; t.n = size(t.type);
        if (t.n < 0) { bad = true; t.n = 0; }
        const size_t bytes = ((size_t)t.n << kTypeShift[t.type & 15]) * (size_t)per_mult;
        if (!need(bytes)) { t.n = 0; return t; }
        t.data.assign((const char *)p, bytes); p += bytes;
        return t;
    }
};
}

size_t bcf_parse_record(const uint8_t *p, size_t avail, VcfRec &v) {
    v = VcfRec();
    if (avail < 32) return 0;
    uint32_t x[8]; memcpy(x, p, 32);
    if (x[0] < 24) return 0;
    const size_t l_shared = x[0] - 24, l_indiv = x[1];
    if (avail - 32 < l_shared || avail - 32 - l_shared < l_indiv) return 0;
    v.rid = (int32_t)x[2]; v.pos = (int32_t)x[3]; v.qual_bits = x[5];
    const uint32_t n_allele = x[6] >> 16, n_info = x[6] & 0xffff;
    uint32_t n_fmt = x[7] >> 24; v.n_sample = (int)(x[7] & 0xffffff);
    if ((!l_indiv || !v.n_sample) && n_fmt) n_fmt = 0;
    Cur c{p + 32, p + 32 + l_shared};
    if (l_shared) {
        v.have_shared = true;
        v.id = c.typed();
        for (uint32_t i = 0; i < n_allele; ++i) v.alleles.push_back(c.typed());
        if (c.need(1)) {
            if (*c.p >> 4) { int type; const int32_t n = c.size(type); for (int32_t i = 0; i < n && !c.bad; ++i) v.flt.push_back(c.int_of(type)); }
            else ++c.p;
        }
        for (uint32_t i = 0; i < n_info && !c.bad; ++i) { VcfRec::Info inf; inf.key = c.typed_int(); inf.v = c.typed(); v.info.push_back(std::move(inf)); }
    }
    Cur d{p + 32 + l_shared, p + 32 + l_shared + l_indiv};
    if (v.n_sample) for (uint32_t i = 0; i < n_fmt && !d.bad; ++i) { VcfRec::Fmt f; f.key = d.typed_int(); f.v = d.typed(v.n_sample); v.fmt.push_back(std::move(f)); }
    if (c.bad || d.bad) return 0;
    return 32 + l_shared + l_indiv;
}

// =====================================================================================================
// bcf_update_info_string (vcf.c:2783-2868, values = one C string)
// =====================================================================================================
bool vcf_update_info_string(const VcfHdr &h, VcfRec &r, const std::string &key, const std::string &value) {
    auto it = h.tags.find(key);
    if (it == h.tags.end() || !it->second.has[HL_INFO]) return false;
    const int id = it->second.id;
    VcfRec::Typed t = enc_chars(value.data(), value.size());
    for (auto &inf : r.info) if (inf.key == id) { inf.v = std::move(t); return true; }
    VcfRec::Info inf; inf.key = id; inf.v = std::move(t);
    r.info.push_back(std::move(inf));
    return true;
}

// =====================================================================================================
// vcf_format (vcf.c:2069-2164) behind bcf_write's sample-count check (:1201-1209)
// =====================================================================================================
static const std::string &name_of(const std::vector<std::string> &names, int id) { static const std::string none; return id >= 0 && (size_t)id < names.size() ? names[(size_t)id] : none; }

static void format_gt(std::string &s, const VcfRec::Typed &f, int isample) {       // bcf_format_gt (vcf.h:803-823)
    const int w = 1 << kTypeShift[f.type & 15];
    const uint8_t *base = (const uint8_t *)f.data.data() + (size_t)isample * (size_t)f.n * (size_t)w;
    int i = 0;
    for (; i < f.n; ++i) {
        int32_t v;
        if (f.type == VT_INT8) { v = (int8_t)base[i]; if (v == INT8_MIN + 1) break; }
        else if (f.type == VT_INT16) { int16_t x; memcpy(&x, base + 2 * i, 2); v = x; if (v == INT16_MIN + 1) break; }
        else { memcpy(&v, base + 4 * i, 4); if (v == kI32End) break; }
        if (i) s += "/|"[v & 1];
        if (!(v >> 1)) s += '.'; else put_int(s, (v >> 1) - 1);
    }
    if (i == 0) s += '.';
}

bool vcf_format_line(const VcfHdr &h, const VcfRec &v, std::string &s) {
    if ((int)h.samples.size() != v.n_sample) {
        fprintf(stderr, "[bcf_write] Broken VCF record, the number of columns at %s:%d does not match the number of samples (%d vs %d).\n",
                name_of(h.contig_name, v.rid).c_str(), (int32_t)((uint32_t)v.pos + 1u), v.n_sample, (int)h.samples.size());
        return false;
    }
    s += name_of(h.contig_name, v.rid);
    s += '\t'; put_int(s, (int32_t)((uint32_t)v.pos + 1u));
    s += '\t';
    if (v.have_shared) fmt_array(s, v.id.n, VT_CHAR, (const uint8_t *)v.id.data.data()); else if (!v.id_buffer_used) s += '.';
    s += '\t';
    if (!v.alleles.empty()) fmt_array(s, v.alleles[0].n, VT_CHAR, (const uint8_t *)v.alleles[0].data.data()); else s += '.';
    s += '\t';
    if (v.alleles.size() > 1) for (size_t i = 1; i < v.alleles.size(); ++i) { if (i > 1) s += ','; fmt_array(s, v.alleles[i].n, VT_CHAR, (const uint8_t *)v.alleles[i].data.data()); }
    else s += '.';
    s += '\t';
    if (v.qual_bits == kFloatMissing) s += '.'; else { float f; memcpy(&f, &v.qual_bits, 4); put_float(s, f); }
    s += '\t';
    if (!v.flt.empty()) for (size_t i = 0; i < v.flt.size(); ++i) { if (i) s += ';'; s += name_of(h.tag_name, v.flt[i]); }
    else s += '.';
    s += '\t';
    if (!v.info.empty()) {
        bool first = true;
        for (const VcfRec::Info &z : v.info) {
            if (!first) s += ';';
            first = false;
            s += name_of(h.tag_name, z.key);
            if (z.v.n <= 0) continue;
            s += '=';
            const uint8_t *d = (const uint8_t *)z.v.data.data();
            if (z.v.n == 1) {                                                    // a lone value: no end-of-vector test, no 0x07 -> '.' mapping
                if (z.v.type == VT_CHAR) s += (char)d[0];
                else if (z.v.type == VT_INT8) { const int8_t x = (int8_t)d[0]; if (x == INT8_MIN) s += '.'; else put_int(s, x); }
                else if (z.v.type == VT_INT16) { int16_t x; memcpy(&x, d, 2); if (x == INT16_MIN) s += '.'; else put_int(s, x); }
                else if (z.v.type == VT_INT32) { int32_t x; memcpy(&x, d, 4); if (x == kI32Missing) s += '.'; else put_int(s, x); }
                else if (z.v.type == VT_FLOAT) { uint32_t u; memcpy(&u, d, 4); if (u == kFloatMissing) s += '.'; else { float f; memcpy(&f, &u, 4); put_float(s, f); } }
            } else fmt_array(s, z.v.n, z.v.type, d);
        }
    } else s += '.';
    if (v.n_sample) {
        if (!v.fmt.empty()) {
            int gt_i = -1;
            for (size_t i = 0; i < v.fmt.size(); ++i) {
                s += i ? ':' : '\t';
                const std::string &k = name_of(h.tag_name, v.fmt[i].key);
                s += k;
                if (k == "GT") gt_i = (int)i;
            }
            for (int j = 0; j < v.n_sample; ++j) {
                s += '\t';
                for (size_t i = 0; i < v.fmt.size(); ++i) {
                    const VcfRec::Typed &f = v.fmt[i].v;
                    if (i) s += ':';
                    if (gt_i == (int)i && (f.type == VT_INT8 || f.type == VT_INT16 || f.type == VT_INT32)) format_gt(s, f, j);
                    else fmt_array(s, f.n, f.type, (const uint8_t *)f.data.data() + (size_t)j * ((size_t)f.n << kTypeShift[f.type & 15]));
                }
            }
        } else for (int j = 0; j <= v.n_sample; ++j) s += "\t.";
    }
    s += '\n';
    return true;
}

}  // namespace rgx
