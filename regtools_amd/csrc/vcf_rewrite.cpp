// vcf_rewrite.cpp -- see vcf_rewrite.h.  Written from the rules of DESIGN.md section 7.1 (H* header, R* record, S* samples, P* printing); the
// rule a piece of code implements is named next to it.  Host code only.
#include "vcf_rewrite.h"
#include <atomic>

#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace rgx {

namespace {

// reserved codes (BCF2 specification, "missing" and "end of vector")
constexpr uint32_t kRealMissing = 0x7F800001u, kRealPad = 0x7F800002u;
constexpr int32_t kMissing32 = INT32_MIN, kPad32 = INT32_MIN + 1;
inline int32_t missing_of(int width) { return width == 1 ? INT8_MIN : width == 2 ? INT16_MIN : INT32_MIN; }
inline int32_t pad_of(int width) { return missing_of(width) + 1; }

// a stretch of bytes inside somebody else's buffer
struct Span {
    const char *p = nullptr;
    size_t n = 0;
    bool is(const char *lit) const { return n == strlen(lit) && !memcmp(p, lit, n); }
    std::string str() const { return std::string(p, n); }
    Span until_nul() const { const void *z = memchr(p, 0, n); return Span{p, z ? (size_t)((const char *)z - p) : n}; }      // R2: a column ends at a NUL byte
};

// cut `s` at every byte for which `is_sep` holds; n separators give n + 1 pieces
template <class F>
void cut(Span s, F is_sep, std::vector<Span> &out) {
    out.clear();
    size_t from = 0;
    for (size_t i = 0; i <= s.n; ++i)
        if (i == s.n || is_sep(s.p[i])) { out.push_back(Span{s.p + from, i - from}); from = i + 1; }
}

void append_int(std::string &s, long v) { char b[24]; s.append(b, (size_t)snprintf(b, sizeof b, "%ld", v)); }
void append_real(std::string &s, uint32_t bits) { float f; memcpy(&f, &bits, 4); char b[48]; s.append(b, (size_t)snprintf(b, sizeof b, "%g", (double)f)); }
uint32_t real_bits(double d) { const float f = (float)d; uint32_t u; memcpy(&u, &f, 4); return u; }

}  // namespace

// =================================================================================================================================================
// header
// =================================================================================================================================================

// H2: one "##" line, read by a small automaton over the bytes (a NUL, or the end of `s`, ends whatever is being read).
//   tag       bytes up to the first '=' (line ends do not stop it); empty or unterminated: the line does not scan
//   plain     what follows the '=' when that is not '<': up to the end of the line
//   attribute name = letter or '_' then letters, digits, '_', '.', followed by '='; anything else: the line does not scan (with a message)
//   value     quoted -- up to the first '"' that no odd run of backslashes precedes -- or bare: up to a ',' at depth 1 or to the '>' that
//             takes the depth to 0 ('<' inside a bare value adds a level)
//   depth     the '>' behind a value is counted AGAIN when it follows directly, so after a bare last value the depth is -1, after a quoted one 0;
//             attributes go on while the depth is not 0 and the line has not ended, one separator byte (whatever it is) skipped in front of each
//   tail      blanks are skipped; the next line starts one byte further on
int VcfDictionary::scan_line(const std::string &s, size_t from, Entry &e, size_t &next) {
    const size_t end = s.size();
    auto at = [&](size_t i) -> unsigned char { return i < end ? (unsigned char)s[i] : 0; };
    e = Entry();
    if (at(from) != '#' || at(from + 1) != '#') { next = from; return 0; }
    enum State { kTag, kPlain, kGap, kName, kQuoted, kBare, kBehindValue, kTail, kBroken } st = kTag;
    size_t i = from + 2, mark = i;
    int depth = 1;
    Attr cur;
    for (;;) {
        const unsigned char c = at(i);
        switch (st) {
        case kTag:
            if (c == '=') {
                if (i == mark) { next = i + 1; return 0; }
                e.tag.assign(s, mark, i - mark);
                ++i;
                if (at(i) == '<') { e.angle = true; st = kGap; } else { mark = i; st = kPlain; }
            } else if (!c) { next = i + 1; return 0; }
            else ++i;
            break;
        case kPlain:
            if (!c || c == '\n') { e.plain.assign(s, mark, i - mark); next = i + 1; return 1; }
            ++i;
            break;
        case kGap:                                                  // on '<' or on the byte between two attributes
            ++i; mark = i; st = kName;
            break;
        case kName: {
            const bool first = i == mark;
            if (first ? (isalpha(c) || c == '_') : (isalnum(c) || c == '_' || c == '.')) { ++i; break; }
            if (c != '=' || first) { st = kBroken; break; }
            cur.name.assign(s, mark, i - mark);
            ++i;
            if (at(i) == '"') { ++i; mark = i; st = kQuoted; } else { mark = i; st = kBare; }
            break;
        }
        case kQuoted: {
            bool closes = false;
            if (c == '"') { size_t k = i, slashes = 0; while (k > mark && s[k - 1] == '\\') { --k; ++slashes; } closes = slashes % 2 == 0; }
            if (!c || closes) {
                cur.text = "\"" + s.substr(mark, i - mark) + "\"";
                if (c) ++i;
                st = kBehindValue;
            } else ++i;
            break;
        }
        case kBare:
            if (c == '<') ++depth;
            if (c == '>') --depth;
            if (!c || depth == 0 || (c == ',' && depth == 1)) { cur.text.assign(s, mark, i - mark); st = kBehindValue; }
            else ++i;
            break;
        case kBehindValue:
            e.attrs.push_back(std::move(cur)); cur = Attr();
            if (c == '>') { --depth; ++i; }
            st = (at(i) && at(i) != '\n' && depth != 0) ? kGap : kTail;
            break;
        case kTail:
            while (at(i) == ' ') ++i;
            next = i + 1;
            return 1;
        case kBroken:
            while (at(i) && at(i) != '\n') ++i;
            next = i + 1;                                                  // (the caller says so: unscannable())
            return -1;
        }
    }
}

// H5: a name takes the number it asks for (IDX=) or the next free one; a number that already names something else is fatal
bool VcfDictionary::claim(std::vector<std::string> &names, int &number, const std::string &name) {
    if (number < 0) number = (int)names.size();
    else if ((size_t)number < names.size() && !names[(size_t)number].empty()) {
        // exit(1), vcf.c:349-353 (the message carries __FILE__ upstream: the build's path; the file's name here)
        fail("[vcf.c:351 bcf_hdr_set_idx] Conflicting IDX=" + std::to_string(number) + " lines in the header dictionary, the new tag is " + name);
        return false;
    }
    if ((size_t)number >= names.size()) names.resize((size_t)number + 1);
    names[(size_t)number] = name;
    return true;
}

namespace {
// the text of attribute `name` (nullptr = absent); the contig rule matches names without regard to case, the id rule exactly
template <class Attrs> const std::string *attr_text(const Attrs &attrs, const char *name, bool any_case) {
    for (auto &a : attrs) if (any_case ? !strcasecmp(a.name.c_str(), name) : a.name == name) return &a.text;
    return nullptr;
}
bool whole_int(const std::string &t, int &v) { char *e; v = (int)strtol(t.c_str(), &e, 10); return !*e; }
}  // namespace

// H4: a contig line needs an ID that is new; a length, when given, must start with a number
bool VcfDictionary::admit_contig(const Entry &e) {
    if (const std::string *len = attr_text(e.attrs, "length", true)) { int dummy; if (sscanf(len->c_str(), "%d", &dummy) != 1) return false; }
    const std::string *id = attr_text(e.attrs, "ID", true);
    if (!id || contig_number_.count(*id)) return false;
    int number = -1;
    if (const std::string *idx = attr_text(e.attrs, "IDX", true)) if (!whole_int(*idx, number)) {
        note("[vcf.c:398 bcf_hdr_register_hrec] Error parsing the IDX tag, skipping.\n");
        return false;
    }
    if (!claim(contig_names_, number, *id)) return false;
    contig_number_[*id] = number;
    return true;
}

// H3: a FILTER / INFO / FORMAT line needs an ID that this role has not declared yet; one name has one number in all three roles
bool VcfDictionary::admit_id(const Entry &e, Role role) {
    const std::string *id = nullptr;
    int number = -1, kind = kUndeclared;
    bool counted_per_genotype = false;
    for (const Attr &a : e.attrs) {                                          // (in the line's order: what is said about it comes out in that order, vcf.c:421-461)
        if (a.name == "ID") id = &a.text;
        else if (a.name == "IDX") {
            if (!whole_int(a.text, number)) { note("[vcf.c:431 bcf_hdr_register_hrec] Error parsing the IDX tag, skipping.\n"); return false; }
        } else if (a.name == "Type") {
            if (a.text == "Integer") kind = kInteger;
            else if (a.text == "Float") kind = kReal;
            else if (a.text == "Flag") kind = kFlag;
            else {
                kind = kText;
                if (a.text != "String" && a.text != "Character") note("[E::bcf_hdr_register_hrec] The type \"" + a.text + "\" is not supported, assuming \"String\"\n");
            }
        } else if (a.name == "Number") counted_per_genotype = a.text == "G";
    }
    if (!id) return false;
    if (role == kFormat && *id == "PL" && !(ids_.count(*id) && ids_[*id].has[kFormat])) pl_is_per_genotype_ = counted_per_genotype;
    auto known = ids_.find(*id);
    if (known != ids_.end()) {
        if (known->second.has[role]) return false;
        known->second.has[role] = true; known->second.kind[role] = kind;
        return true;
    }
    if (!claim(id_names_, number, *id)) return false;
    Id fresh; fresh.number = number; fresh.has[role] = true; fresh.kind[role] = kind;
    ids_[*id] = fresh;
    return true;
}

// H3-H7: which lines are kept
void VcfDictionary::admit(Entry &&e) {
    bool keep;
    if (e.tag == "contig") { e.cls = Entry::kContigDecl; keep = admit_contig(e); }
    else if (e.tag == "INFO") { e.cls = Entry::kInfoDecl; keep = admit_id(e, kInfo); }
    else if (e.tag == "FILTER") { e.cls = Entry::kFilterDecl; keep = admit_id(e, kFilter); }
    else if (e.tag == "FORMAT") { e.cls = Entry::kFormatDecl; keep = admit_id(e, kFormat); }
    else if (!e.attrs.empty()) { e.cls = Entry::kStructured; keep = true; }                        // H6: ##ALT, ##PEDIGREE ...: always, repeats too
    else {                                                                                         // H7: ##tag=text
        keep = true;
        for (const Entry &o : entries_)
            if (o.cls == Entry::kGeneric && o.tag == e.tag && (e.tag == "fileformat" || o.plain == e.plain)) { keep = false; break; }
    }
    if (keep) entries_.push_back(std::move(e));
}

// vcf.c:309
void VcfDictionary::unscannable(const std::string &s, size_t from, size_t next) {
    note("Could not parse the header line: \"" + s.substr(from, next - 1 - from) + "\"\n");
}

bool VcfDictionary::declare(const std::string &line) {
    Entry e; size_t next;
    const int scanned = scan_line(line, 0, e, next);
    if (scanned < 0) unscannable(line, 0, next);
    if (scanned <= 0) return false;
    admit(std::move(e));
    return true;
}

// H9: the line header reading stopped at is the column line; its fields from the tenth on name the samples
void VcfDictionary::read_column_line(const std::string &s, size_t from) {
    if (from > s.size()) from = s.size();
    size_t stop = from;
    while (stop < s.size() && s[stop] && s[stop] != '\n') ++stop;
    std::vector<Span> cols;
    cut(Span{s.data() + from, stop - from}, [](char c) { return c == '\t'; }, cols);
    for (size_t k = 9; k < cols.size(); ++k) {
        std::string name = cols[k].str();
        // both end in abort() (vcf.c:64-69, 79-83)
        if (name.empty()) { fail("[E::bcf_hdr_add_sample] Empty sample name: trailing spaces/tabs in the header line?", true); return; }
        for (auto &have : samples_) if (have == name) { fail("[E::bcf_hdr_add_sample] Duplicated sample name '" + name + "'", true); return; }
        samples_.push_back(std::move(name));
    }
}

// H1, H8
void VcfDictionary::ingest(const std::string &text) {
    Entry e; size_t next;
    const int first_scan = scan_line(text, 0, e, next);
    const bool first_scans = first_scan > 0;
    if (first_scan < 0) unscannable(text, 0, next);
    if (!first_scans || strcasecmp(e.tag.c_str(), "fileformat")) fprintf(stderr, "[W::bcf_hdr_parse] The first line should be ##fileformat; is the VCF/BCF header broken?\n");
    if (first_scans) admit(std::move(e));
    declare("##FILTER=<ID=PASS,Description=\"All filters passed\">");          // id number 0, line 2 of what is printed
    size_t at = 0;                                                             // (the first line is met again here, as a repeat)
    while (failure_.empty() && at < text.size()) {
        const int scanned = scan_line(text, at, e, next);
        if (scanned < 0) unscannable(text, at, next);
        if (scanned <= 0) break;
        admit(std::move(e)); at = next;
    }
    if (!failure_.empty()) return;
    read_column_line(text, at);
    if (!failure_.empty()) return;
    // bcf_hdr_check_sanity (vcf.c:564-586): once per process.  (Its GL half looks the name up among the SAMPLES -- vcf.c:577 passes the wrong
    // dictionary -- and so never fires for a FORMAT id; it is left out.)
    static std::atomic<bool> pl_warned{false};
    const Id *pl = find_id("PL");
    if (pl && pl->has[kFormat] && !pl_is_per_genotype_ && !pl_warned.exchange(true)) note("[W::bcf_hdr_check_sanity] PL should be declared as Number=G\n");
}

// P1
void VcfDictionary::render(std::string &out) const {
    for (const Entry &e : entries_) {
        out += "##"; out += e.tag; out += '=';
        if (!e.angle) { out += e.plain; out += '\n'; continue; }
        out += '<';
        const char *sep = "";
        for (const Attr &a : e.attrs) {
            if (a.name == "IDX") continue;
            out += sep; out += a.name; out += '='; out += a.text;
            sep = ",";
        }
        out += ">\n";
    }
    out += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
    if (!samples_.empty()) { out += "\tFORMAT"; for (auto &s : samples_) { out += '\t'; out += s; } }
    out += '\n';
}

void VcfDictionary::note(const std::string &line) {
    if (quiet_) return;
    if (sink_) sink_->push_back(line);
    else fputs(line.c_str(), stderr);
}

int VcfDictionary::contig_for(const std::string &name) {
    auto it = contig_number_.find(name);
    if (it == contig_number_.end()) {
        note("[W::vcf_parse] contig '" + name + "' is not defined in the header. (Quick workaround: index the file with tabix.)\n");
        declare("##contig=<ID=" + name + ">");
        it = contig_number_.find(name);
    }
    return it == contig_number_.end() ? -1 : it->second;
}

VcfDictionary::Id VcfDictionary::id_for(const std::string &name, Role role) {
    const Id *have = find_id(name);
    if (have && (role == kFilter || have->has[role])) return *have;            // R6: a FILTER name may be an id of any role
    if (role == kFilter) {
        note("[W::vcf_parse] FILTER '" + name + "' is not defined in the header\n");
        declare("##FILTER=<ID=" + name + ",Description=\"Dummy\">");
    } else {
        note(std::string(role == kInfo ? "[W::vcf_parse] INFO '" : "[W::_vcf_parse_format] FORMAT '") + name + "' is not defined in the header, assuming Type=String\n");
        declare(std::string(role == kInfo ? "##INFO=<ID=" : "##FORMAT=<ID=") + name + ",Number=1,Type=String,Description=\"Dummy\">");
    }
    have = find_id(name);
    return have ? *have : Id();
}

// =================================================================================================================================================
// values
// =================================================================================================================================================
namespace {

VcfValue text_value(Span s) { VcfValue v; v.store = VcfValue::kBytes; v.count = (int)s.n; v.bytes.assign(s.p, s.n); return v; }

// R9: integers read from text (32-bit, the two reserved codes included) take the narrowest width that holds all of them.  One value alone:
// anything in (-128, 127] is one byte, in (-32768, 32767] two; several: the largest must fit and the smallest must lie ABOVE the width's
// two reserved codes.  The reserved codes themselves fit everywhere and become the width's own.
void settle_width(VcfValue &v) {
    const size_t n = v.ints.size();
    if (!n) { v.store = VcfValue::kNone; v.count = 0; return; }
    v.store = VcfValue::kInts;
    int width;
    if (n == 1) {
        const int32_t x = v.ints[0];
        width = (x == kPad32 || x == kMissing32) ? 1 : (x <= INT8_MAX && x > INT8_MIN) ? 1 : (x <= INT16_MAX && x > INT16_MIN) ? 2 : 4;
    } else {
        int32_t hi = INT32_MIN + 1, lo = INT32_MAX;
        for (int32_t x : v.ints) if (x != kMissing32 && x != kPad32) { if (x > hi) hi = x; if (x < lo) lo = x; }
        width = (hi <= INT8_MAX && lo > INT8_MIN + 1) ? 1 : (hi <= INT16_MAX && lo > INT16_MIN + 1) ? 2 : 4;
    }
    v.width = (uint8_t)width;
    for (int32_t &x : v.ints) x = x == kMissing32 ? missing_of(width) : x == kPad32 ? pad_of(width) : x;
}

// R8: the numbers of an INFO value.  Slots = commas + 1.  A cursor reads a number (C's strtol / strtod: blanks, sign, as many digits as fit the
// syntax); where none starts the slot is missing and the cursor moves up to the next comma; at the end of the text the slots left over are
// missing; otherwise ONE byte is stepped over -- whatever it is -- and the next slot is read there.
template <class Read, class Put>
void read_info_numbers(const std::string &text, Read read, Put put) {
    size_t slots = 1;
    for (char c : text) slots += c == ',';
    const char *cur = text.c_str();
    size_t k = 0;
    for (; k < slots; ++k) {
        const char *stop = cur;
        const bool got = read(cur, stop);
        put(got);
        if (!got) while (*stop && *stop != ',') ++stop;
        cur = stop;
        if (!*cur) { ++k; break; }
        ++cur;
    }
    for (; k < slots; ++k) put(false);
}

VcfValue info_integers(const std::string &text) {
    VcfValue v;
    long got_v = 0;
    read_info_numbers(text, [&](const char *at, const char *&stop) { char *e; got_v = strtol(at, &e, 10); stop = e; return e != at; },
                      [&](bool got) { v.ints.push_back(got ? (int32_t)got_v : kMissing32); });
    v.count = (int)v.ints.size();
    settle_width(v);
    return v;
}

VcfValue info_reals(const std::string &text) {
    VcfValue v; v.store = VcfValue::kReals;
    double got_v = 0;
    read_info_numbers(text, [&](const char *at, const char *&stop) { char *e; got_v = strtod(at, &e); stop = e; return e != at; },
                      [&](bool got) { v.reals.push_back(got ? real_bits(got_v) : kRealMissing); });
    v.count = (int)v.reals.size();
    return v;
}

}  // namespace

// =================================================================================================================================================
// sample columns (S1-S8)
// =================================================================================================================================================
namespace {

// S4: the samples' values are laid on one tape of bytes -- a block per FORMAT key in key order, each starting at a multiple of 8, a block
// being n_samples slots of the key's width -- and a value is written from its slot's first byte on WITHOUT being cut to the slot: what does
// not fit runs on along the tape, a later write covers an earlier one, bytes past the tape's end are dropped.  (With S3's first-key rule that is how the
// last letter of a sample's text can turn up inside the next key's first number.)
class Tape {
  public:
    size_t add_block(size_t bytes) { const size_t at = (mem_.size() + 7) & ~(size_t)7; mem_.resize(at + bytes, '\0'); return at; }
    void byte(size_t at, char c) { if (at < mem_.size()) mem_[at] = c; }
    void word(size_t at, uint32_t w) { if (at < mem_.size()) memcpy(&mem_[at], &w, std::min<size_t>(4, mem_.size() - at)); }   // (a number that straddles the end leaves its first bytes)
    const char *at(size_t o) const { return mem_.data() + o; }
  private:
    std::string mem_;
};

struct Column {                       // one FORMAT key
    enum Shape { kText, kGenotype, kInts, kReals } shape = kText;
    int key = -1;
    long chars = 0, commas1 = 0, alleles = 0;      // S3: the widest piece seen: bytes; commas + 1; '/' and '|' + 1
    size_t slot = 0, origin = 0;                   // bytes per sample; the block's place on the tape
};

// S5: how a piece of text becomes a slot's bytes.  All three number readers share one walk: read a value at the cursor; at the piece's end stop,
// else step over ONE byte (whatever it is) and read again; the slot's remaining places get the "no more values" code.
template <class ReadOne>
void write_numbers(Tape &tape, size_t at, size_t places, const std::string &piece, uint32_t pad, ReadOne read_one) {
    const char *cur = piece.c_str();
    size_t k = 0;
    for (;;) {
        tape.word(at + 4 * k++, read_one(cur));
        if (!*cur) break;
        ++cur;
    }
    for (; k < places; ++k) tape.word(at + 4 * k, pad);
}

void write_piece(Tape &tape, const Column &col, size_t sample, const std::string &piece) {
    const size_t at = col.origin + col.slot * sample, places = col.slot / 4;
    switch (col.shape) {
    case Column::kText: {
        size_t k = 0;
        for (; k < piece.size(); ++k) tape.byte(at + k, piece[k]);
        for (; k < col.slot; ++k) tape.byte(at + k, 0);
        break;
    }
    case Column::kGenotype: {                       // allele a is stored as (a + 1) * 2, '.' as 0; + 1 when a '|' stands in front of it
        uint32_t bar = 0;
        write_numbers(tape, at, places, piece, (uint32_t)kPad32, [&](const char *&cur) {
            uint32_t code;
            if (*cur == '.') { ++cur; code = bar; }
            else { char *e; const long a = strtol(cur, &e, 10); cur = e; code = (uint32_t)(((uint64_t)(a + 1) << 1) | bar); }
            bar = *cur == '|';
            return code;
        });
        break;
    }
    case Column::kInts:
        write_numbers(tape, at, places, piece, (uint32_t)kPad32, [](const char *&cur) {
            if (*cur == '.') { ++cur; return (uint32_t)kMissing32; }
            char *e; const long x = strtol(cur, &e, 10); cur = e;
            return (uint32_t)(int32_t)x;
        });
        break;
    case Column::kReals:
        write_numbers(tape, at, places, piece, kRealPad, [](const char *&cur) {
            if (*cur == '.' && !isdigit((unsigned char)cur[1])) { ++cur; return kRealMissing; }
            char *e; const double x = strtod(cur, &e); cur = e;
            return real_bits(x);
        });
        break;
    }
}

// S6: a key the sample's text does not reach
void write_absent(Tape &tape, const Column &col, size_t sample) {
    const size_t at = col.origin + col.slot * sample;
    if (col.shape == Column::kText) { for (size_t k = 0; k < col.slot; ++k) tape.byte(at + k, k ? 0 : '.'); return; }
    const uint32_t first = col.shape == Column::kReals ? kRealMissing : (uint32_t)kMissing32, rest = col.shape == Column::kReals ? kRealPad : (uint32_t)kPad32;
    tape.word(at, first);                                                              // (also into a slot of no width: S4)
    for (size_t k = 1; k < col.slot / 4; ++k) tape.word(at + 4 * k, rest);
}

// "CHROM:POS" the way the reference's messages name a record
std::string place_of(const VcfDictionary &dict, const VcfRecord &rec) { return dict.contig_name(rec.contig) + ":" + std::to_string((long long)rec.pos0 + 1); }

// The messages below carry __FILE__ and __LINE__ upstream (vcf.c:1553, 1764): the path is the build's, so only the file's name is repeated here.
ReadResult read_samples(VcfDictionary &dict, Span format, Span samples_region, bool have_region, VcfRecord &rec, bool names_only) {
    const size_t want = dict.n_samples();
    if (!want) return ReadResult::kOk;                                               // S1: a header without samples: the columns are not looked at
    if (!have_region) {
        dict.note("[vcf.c:1553 _vcf_parse_format] Error: FORMAT column with no sample columns starting at " + place_of(dict, rec) + "\n");
        return ReadResult::kRefused;
    }
    // S1: the keys
    std::vector<Span> names;
    cut(format.until_nul(), [](char c) { return c == ':'; }, names);
    std::vector<Column> cols(names.size());
    std::vector<int> declared(names.size());
    for (size_t j = 0; j < names.size(); ++j) {
        const std::string name = names[j].str();
        const VcfDictionary::Id id = dict.id_for(name, VcfDictionary::kFormat);
        cols[j].key = id.number;
        declared[j] = id.kind[VcfDictionary::kFormat];
        cols[j].shape = declared[j] == VcfDictionary::kInteger ? Column::kInts : declared[j] == VcfDictionary::kReal ? Column::kReals
                        : name == "GT" ? Column::kGenotype : Column::kText;
    }
    // S2: the samples: the region behind FORMAT cut at tabs (and NUL bytes), the header's count of them at most
    std::vector<Span> all, pieces;
    cut(samples_region, [](char c) { return c == '\t' || c == 0; }, all);
    if (!all.empty() && all.back().n == 0) all.pop_back();                            // nothing behind the last tab is no sample
    const size_t have = all.size() < want ? all.size() : want;
    // S3: widths
    for (size_t s = 0; s < have; ++s) {
        cut(all[s], [](char c) { return c == ':'; }, pieces);
        if (pieces.size() > cols.size()) { dict.fail("Incorrect number of FORMAT fields at " + place_of(dict, rec)); return ReadResult::kFatal; }   // exit(1), vcf.c:1612
        for (size_t j = 0; j < pieces.size(); ++j) {
            Column &c = cols[j];
            long commas1 = 1, alleles = 1;
            for (size_t k = 0; k < pieces[j].n; ++k) { const char ch = pieces[j].p[k]; commas1 += ch == ','; alleles += ch == '/' || ch == '|'; }
            const long chars = (long)pieces[j].n - (s > 0 && j == 0 ? 1 : 0);         // the first key's text counts one short in every sample but the first
            if (chars > c.chars) c.chars = chars;
            if (commas1 > c.commas1) c.commas1 = commas1;
            if (alleles > c.alleles) c.alleles = alleles;
        }
    }
    Tape tape;
    for (size_t j = 0; j < cols.size(); ++j) {
        Column &c = cols[j];
        if (declared[j] != VcfDictionary::kInteger && declared[j] != VcfDictionary::kReal && declared[j] != VcfDictionary::kText) {
            // (abort(), vcf.c:1638-1639; the number is the declaration's type code: a Flag is 0)
            // (a declaration without a Type reads as 15 there)
            dict.fail("[E::_vcf_parse_format] the format type " + std::to_string(declared[j] < 0 ? 15 : declared[j]) + " currently not supported", true);
            return ReadResult::kFatal;
        }
        if (names_only) continue;
        c.slot = c.shape == Column::kText ? (size_t)c.chars : 4 * (size_t)(c.shape == Column::kGenotype ? c.alleles : c.commas1 ? c.commas1 : 1);
        c.origin = tape.add_block(c.slot * have);
    }
    const auto counted = [&]() {
        if (have == want) return ReadResult::kOk;
        dict.note("[vcf.c:1764 _vcf_parse_format] Number of columns at " + place_of(dict, rec) + " does not match the number of samples (" + std::to_string(have) + " vs " +
                  std::to_string(want) + ").\n");
        return ReadResult::kRefused;
    };
    if (names_only) return counted();
    // S5-S7: the values, sample by sample, key by key
    for (size_t s = 0; s < have; ++s) {
        if (all[s].n == 0) continue;                                                  // S7: an empty column writes nothing: its slots stay zero bytes
        cut(all[s], [](char c) { return c == ':'; }, pieces);
        size_t n = pieces.size();
        const bool open_end = n > 1 && pieces[n - 1].n == 0;                          // S7: "...:" -- the keys from the one behind the last colon on stay zero bytes
        if (open_end) --n;
        for (size_t j = 0; j < n && j < cols.size(); ++j) write_piece(tape, cols[j], s, pieces[j].str());
        if (!open_end) for (size_t j = n; j < cols.size(); ++j) write_absent(tape, cols[j], s);
    }
    // S8: read the blocks back
    rec.n_samples = (int)have;
    for (const Column &c : cols) {
        VcfRecord::Tagged f; f.key = c.key;
        if (c.shape == Column::kText) { f.v.store = VcfValue::kBytes; f.v.count = (int)c.slot; f.v.bytes.assign(tape.at(c.origin), c.slot * have); }
        else if (c.shape == Column::kReals) {
            f.v.store = VcfValue::kReals; f.v.count = (int)(c.slot / 4); f.v.reals.resize(c.slot / 4 * have);
            if (!f.v.reals.empty()) memcpy(f.v.reals.data(), tape.at(c.origin), f.v.reals.size() * 4);
        } else {
            f.v.ints.resize(c.slot / 4 * have);
            if (!f.v.ints.empty()) memcpy(f.v.ints.data(), tape.at(c.origin), f.v.ints.size() * 4);
            settle_width(f.v);
            f.v.count = (int)(c.slot / 4);
        }
        rec.fields.push_back(std::move(f));
    }
    return counted();
}

}  // namespace

// =================================================================================================================================================
// a text line -> record (R1-R11)
// =================================================================================================================================================
ReadResult read_text_record(VcfDictionary &dict, const char *line, size_t len, VcfRecord &rec, bool names_only) {
    rec = VcfRecord();
    std::vector<Span> col;
    cut(Span{line, len}, [](char c) { return c == '\t'; }, col);
    // R1
    rec.contig = dict.contig_for(col[0].until_nul().str());
    // R2
    if (col.size() > 1) rec.pos0 = (int32_t)((uint32_t)atoi(col[1].until_nul().str().c_str()) - 1u);
    if (col.size() > 2) {                                                             // R3
        rec.past_pos = true;
        if (!col[2].until_nul().is(".")) rec.id = col[2].str();                       // (kept whole: printing stops at a NUL byte, P2)
    }
    if (col.size() > 3 && !names_only) rec.alleles.push_back(col[3].str());           // R4
    if (col.size() > 4 && !names_only && !col[4].until_nul().is(".")) {                              // R4: ALT is cut at commas -- and at NUL bytes
        std::vector<Span> alts;
        cut(col[4], [](char c) { return c == ',' || c == 0; }, alts);
        for (Span a : alts) rec.alleles.push_back(a.str());
    }
    if (col.size() > 5 && !names_only) {                                              // R5
        const std::string q = col[5].until_nul().str();
        if (q != ".") rec.qual = real_bits(atof(q.c_str()));
    }
    // R6, R7: lists separated by ';', one ';' at the very end of the column is not a separator
    auto list_of = [](Span c) { Span s = c.until_nul(); if (s.n == c.n && s.n && s.p[s.n - 1] == ';') --s.n; return s; };
    if (col.size() > 6 && !col[6].until_nul().is(".")) {
        std::vector<Span> names;
        cut(list_of(col[6]), [](char c) { return c == ';'; }, names);
        for (Span n : names) rec.filters.push_back(dict.id_for(n.str(), VcfDictionary::kFilter).number);
    }
    if (col.size() > 7 && !col[7].until_nul().is(".")) {
        std::vector<Span> items;
        cut(list_of(col[7]), [](char c) { return c == ';'; }, items);
        for (Span it : items) {
            const char *eq = (const char *)memchr(it.p, '=', it.n);
            const Span key{it.p, eq ? (size_t)(eq - it.p) : it.n};
            if (!key.n) continue;                                                     // ";;", "=x"
            const VcfDictionary::Id id = dict.id_for(key.str(), VcfDictionary::kInfo);
            if (names_only) continue;
            VcfRecord::Tagged t; t.key = id.number;
            if (eq) {                                                                 // R7: without '=' there is no value, whatever the type
                const Span val{eq + 1, it.n - key.n - 1};
                const int kind = id.kind[VcfDictionary::kInfo];
                if (kind == VcfDictionary::kFlag || kind == VcfDictionary::kText) t.v = text_value(val);
                else if (kind == VcfDictionary::kInteger) t.v = info_integers(val.str());
                else if (kind == VcfDictionary::kReal) t.v = info_reals(val.str());
            }
            rec.info.push_back(std::move(t));
        }
    }
    if (col.size() > 8) {                                                             // R11
        const bool have_region = col.size() > 9;
        const Span region = have_region ? Span{col[9].p, (size_t)(line + len - col[9].p)} : Span{};
        return read_samples(dict, col[8], region, have_region, rec, names_only);
    }
    return ReadResult::kOk;
}

// =================================================================================================================================================
// a BCF record -> record (BCF2.2: the hts-specs document; R10)
// =================================================================================================================================================
namespace {

class Bytes {
  public:
    Bytes(const uint8_t *p, size_t n) : p_(p), left_(n) {}
    bool ok() const { return ok_; }
    bool empty() const { return left_ == 0; }
    uint8_t peek() const { return left_ ? *p_ : 0; }
    const uint8_t *take(size_t n) { if (n > left_) { ok_ = false; left_ = 0; return nullptr; } const uint8_t *q = p_; p_ += n; left_ -= n; return q; }
    // an integer of `type` (1, 2: that many bytes; anything else: four)
    int32_t integer(int type) {
        const size_t w = type == 1 ? 1 : type == 2 ? 2 : 4;
        const uint8_t *q = take(w);
        if (!q) return 0;
        if (w == 1) return (int8_t)q[0];
        if (w == 2) { int16_t v; memcpy(&v, q, 2); return v; }
        int32_t v; memcpy(&v, q, 4); return v;
    }
    // a descriptor byte: type in the low nibble, count in the high one; count 15 = a typed integer follows with the count
    void descriptor(int &type, int32_t &count) {
        const uint8_t *d = take(1);
        if (!d) { type = 0; count = 0; return; }
        type = *d & 15; count = *d >> 4;
        if (count == 15) { const uint8_t *t = take(1); count = t ? integer(*t & 15) : 0; }
        if (count < 0) { ok_ = false; count = 0; }
    }
    int32_t typed_integer() { const uint8_t *t = take(1); return t ? integer(*t & 15) : 0; }
    // `count` x `repeat` elements of `type`
    void vector(int type, int32_t count, size_t repeat, VcfValue &v) {
        const size_t elem = type == 2 ? 2 : (type == 3 || type == 5) ? 4 : 1, n = (size_t)count * repeat;
        const uint8_t *q = take(n * elem);
        v = VcfValue(); v.count = count;
        if (!q) { v.count = 0; return; }
        if (type == 1 || type == 2 || type == 3) {
            v.store = VcfValue::kInts; v.width = (uint8_t)elem; v.ints.resize(n);
            for (size_t i = 0; i < n; ++i) { if (elem == 1) v.ints[i] = (int8_t)q[i]; else if (elem == 2) { int16_t x; memcpy(&x, q + 2 * i, 2); v.ints[i] = x; } else memcpy(&v.ints[i], q + 4 * i, 4); }
        } else if (type == 5) { v.store = VcfValue::kReals; v.reals.resize(n); if (n) memcpy(v.reals.data(), q, 4 * n); }
        else { v.store = type == 7 ? VcfValue::kBytes : VcfValue::kOpaque; v.bytes.assign((const char *)q, n); }
    }
    void value(VcfValue &v, size_t repeat = 1) { int type; int32_t count; descriptor(type, count); vector(type, count, repeat, v); }
    // ID and alleles: `count` bytes are their text whatever the descriptor says the elements are (P2 prints them up to a NUL)
    std::string text() {
        int type; int32_t count; descriptor(type, count);
        const size_t elem = type == 2 ? 2 : (type == 3 || type == 5) ? 4 : 1;
        const uint8_t *q = take((size_t)count * elem);
        return q ? std::string((const char *)q, (size_t)count) : std::string();
    }
  private:
    const uint8_t *p_; size_t left_; bool ok_ = true;
};


}  // namespace

size_t read_bcf_record(const uint8_t *p, size_t avail, VcfRecord &rec) {
    rec = VcfRecord();
    if (avail < 32) return 0;
    uint32_t head[8]; memcpy(head, p, 32);
    if (head[0] < 24) return 0;                                                       // the first length counts the 24 fixed bytes behind the two lengths
    const size_t shared = head[0] - 24, indiv = head[1];
    if (avail - 32 < shared || avail - 32 - shared < indiv) return 0;
    rec.contig = (int32_t)head[2]; rec.pos0 = (int32_t)head[3]; rec.qual = head[5];
    const uint32_t n_alleles = head[6] >> 16, n_info = head[6] & 0xffffu;
    uint32_t n_fields = head[7] >> 24;
    rec.n_samples = (int)(head[7] & 0xffffffu);
    if (!indiv || !rec.n_samples) n_fields = 0;
    Bytes sh(p + 32, shared);
    if (shared) {
        rec.past_pos = true;
        rec.id = sh.text();
        for (uint32_t a = 0; a < n_alleles; ++a) rec.alleles.push_back(sh.text());
        if (!sh.empty()) {
            if (sh.peek() >> 4) { int type; int32_t count; sh.descriptor(type, count); for (int32_t k = 0; k < count && sh.ok(); ++k) rec.filters.push_back(sh.integer(type)); }
            else sh.take(1);
        } else sh.take(1);
        for (uint32_t k = 0; k < n_info && sh.ok(); ++k) { VcfRecord::Tagged t; t.key = sh.typed_integer(); sh.value(t.v); rec.info.push_back(std::move(t)); }
    }
    Bytes in(p + 32 + shared, indiv);
    for (uint32_t k = 0; k < n_fields && in.ok(); ++k) { VcfRecord::Tagged t; t.key = in.typed_integer(); in.value(t.v, (size_t)rec.n_samples); rec.fields.push_back(std::move(t)); }
    if (!sh.ok() || !in.ok()) return 0;
    return 32 + shared + indiv;
}

// R12
bool set_info_text(const VcfDictionary &dict, VcfRecord &rec, const std::string &key, const std::string &value) {
    const VcfDictionary::Id *id = dict.find_id(key);
    if (!id || !id->has[VcfDictionary::kInfo]) return false;
    VcfValue v = text_value(Span{value.data(), value.size()});
    for (auto &t : rec.info) if (t.key == id->number) { t.v = std::move(v); return true; }
    VcfRecord::Tagged t; t.key = id->number; t.v = std::move(v);
    rec.info.push_back(std::move(t));
    return true;
}

// =================================================================================================================================================
// record -> text (P2-P7)
// =================================================================================================================================================
namespace {

// P2: text: up to a NUL byte, the byte 0x07 reads '.'; nothing at all reads '.'
void print_text(std::string &out, const char *p, size_t n) {
    if (!n) { out += '.'; return; }
    for (size_t k = 0; k < n && p[k]; ++k) out += p[k] == 0x07 ? '.' : p[k];
}

// P2: `count` values from element `from` on: commas between them, '.' for a missing one, the first "no more values" code ends the list (the
// comma in front of it included: it was never written); no values at all read '.'
void print_list(std::string &out, const VcfValue &v, size_t from, int count) {
    if (count <= 0) { out += '.'; return; }
    switch (v.store) {
    case VcfValue::kBytes: print_text(out, v.bytes.data() + from, (size_t)count); return;
    case VcfValue::kInts:
        for (int k = 0; k < count; ++k) {
            const int32_t x = v.ints[from + (size_t)k];
            if (x == pad_of(v.width)) return;
            if (k) out += ',';
            if (x == missing_of(v.width)) out += '.'; else append_int(out, x);
        }
        return;
    case VcfValue::kReals:
        for (int k = 0; k < count; ++k) {
            const uint32_t x = v.reals[from + (size_t)k];
            if (x == kRealPad) return;
            if (k) out += ',';
            if (x == kRealMissing) out += '.'; else append_real(out, x);
        }
        return;
    default: return;                                                                  // a storage class nothing prints
    }
}

// P5: one value on its own (INFO only): no "no more values" test -- that code is a number like any other here -- and no 0x07 rule
void print_alone(std::string &out, const VcfValue &v) {
    if (v.store == VcfValue::kBytes) out += v.bytes[0];
    else if (v.store == VcfValue::kInts) { if (v.ints[0] == missing_of(v.width)) out += '.'; else append_int(out, v.ints[0]); }
    else if (v.store == VcfValue::kReals) { if (v.reals[0] == kRealMissing) out += '.'; else append_real(out, v.reals[0]); }
}

// P6: a genotype: codes up to the first "no more values"; '/' or '|' by a code's low bit (not in front of the first); the rest of the
// code, halved, is 0 for '.' and allele + 1 otherwise; nothing printed at all reads '.'
void print_genotype(std::string &out, const VcfValue &v, size_t from, int count) {
    int k = 0;
    for (; k < count; ++k) {
        const int32_t code = v.ints[from + (size_t)k];
        if (code == pad_of(v.width)) break;
        if (k) out += (code & 1) ? '|' : '/';
        if ((code >> 1) == 0) out += '.'; else append_int(out, (long)(code >> 1) - 1);
    }
    if (!k) out += '.';
}

}  // namespace

bool write_text_record(const VcfDictionary &dict, const VcfRecord &rec, std::string &out) {
    const long pos = (int32_t)((uint32_t)rec.pos0 + 1u);
    if ((int)dict.n_samples() != rec.n_samples) {                                     // P7
        fprintf(stderr, "[vcf.c:1207 bcf_write] Broken VCF record, the number of columns at %s:%d does not match the number of samples (%d vs %d).\n",
                dict.contig_name(rec.contig).c_str(), (int)pos, rec.n_samples, (int)dict.n_samples());
        return false;
    }
    out += dict.contig_name(rec.contig);
    out += '\t'; append_int(out, pos);
    out += '\t';
    if (rec.past_pos) print_text(out, rec.id.data(), rec.id.size());                   // P3
    else if (!rec.id_seen_before) out += '.';
    out += '\t';
    if (rec.alleles.empty()) out += '.'; else print_text(out, rec.alleles[0].data(), rec.alleles[0].size());
    out += '\t';
    if (rec.alleles.size() < 2) out += '.';
    else for (size_t a = 1; a < rec.alleles.size(); ++a) { if (a > 1) out += ','; print_text(out, rec.alleles[a].data(), rec.alleles[a].size()); }
    out += '\t';
    if (rec.qual == kRealMissing) out += '.'; else append_real(out, rec.qual);
    out += '\t';
    if (rec.filters.empty()) out += '.';
    else for (size_t k = 0; k < rec.filters.size(); ++k) { if (k) out += ';'; out += dict.id_name(rec.filters[k]); }
    out += '\t';
    if (rec.info.empty()) out += '.';                                                 // P4
    else {
        const char *sep = "";
        for (const VcfRecord::Tagged &t : rec.info) {
            out += sep; sep = ";";
            out += dict.id_name(t.key);
            if (t.v.count <= 0) continue;                                             // a flag, an empty text: the key alone
            out += '=';
            if (t.v.count == 1) print_alone(out, t.v); else print_list(out, t.v, 0, t.v.count);
        }
    }
    if (rec.n_samples) {                                                              // P6
        if (rec.fields.empty()) { for (int s = 0; s <= rec.n_samples; ++s) out += "\t."; }
        else {
            int genotype_at = -1;                                                     // the LAST key that is called GT
            for (size_t j = 0; j < rec.fields.size(); ++j) {
                out += j ? ':' : '\t';
                const std::string &name = dict.id_name(rec.fields[j].key);
                out += name;
                if (name == "GT") genotype_at = (int)j;
            }
            for (int s = 0; s < rec.n_samples; ++s) {
                out += '\t';
                for (size_t j = 0; j < rec.fields.size(); ++j) {
                    const VcfValue &v = rec.fields[j].v;
                    if (j) out += ':';
                    const size_t from = (size_t)s * (size_t)(v.count > 0 ? v.count : 0);
                    if ((int)j == genotype_at && v.store == VcfValue::kInts) print_genotype(out, v, from, v.count);
                    else print_list(out, v, from, v.count);
                }
            }
        }
    }
    out += '\n';
    return true;
}

}  // namespace rgx
