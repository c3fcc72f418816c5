// vcf_rewrite.h -- the annotated-VCF output (`cis-splice-effects identify -v`, `variants annotate -o`): read a VCF / BCF record, add the four
// INFO strings, print it the way the reference's output reads.  Host code; nothing here runs on the device.
//
// The reference does not copy its input lines: every record goes through htslib's typed form and comes back out as text
// (variants_annotator.cc:118-154, 521-537: bcf_hdr_read, bcf_hdr_append x 4, bcf_hdr_write; bcf_read, bcf_update_info_string x 4, bcf_write).
// What a user sees of that round trip is a NORMALISATION of the text -- numbers re-printed, header lines de-duplicated, sample fields padded --
// and DESIGN.md section 7.1 states it as rules H1-H9 (header), R1-R12 (record -> values), S1-S8 (sample columns), P1-P7 (values -> text),
// each pinned by outputs of the real reference (tests/golden/vcf_writer, tests/golden/annot_ref).  This module is written from those rules:
//
//   VcfDictionary        the header: an ordered list of entries + two name registries (FILTER/INFO/FORMAT ids, contigs)     rules H*
//   VcfRecord/VcfValue   one record as typed vectors (what BCF stores, so a BCF record loads without conversion)          rules R*, S*
//   read_text_record / read_bcf_record / set_info_text / write_text_record                                                 rules R*, S*, P*
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace rgx {

// ---- the header ---------------------------------------------------------------------------------------------------------------------------
class VcfDictionary {
  public:
    enum Role { kFilter = 0, kInfo = 1, kFormat = 2 };
    enum Kind { kUndeclared = -1, kFlag = 0, kInteger = 1, kReal = 2, kText = 3 };        // the Type= of an INFO / FORMAT declaration
    struct Id { int number = -1; bool has[3] = {false, false, false}; int kind[3] = {kUndeclared, kUndeclared, kUndeclared}; };

    // text = the "##" lines and the column line, each ending in '\n' (VCF), or the header block of a BCF stream.  Afterwards failure() is
    // empty or the message the reference stops with.
    void ingest(const std::string &text);
    // one more "##..." line, under the same admission rules (H3-H7).  false = the line does not scan.
    bool declare(const std::string &line);
    void render(std::string &out) const;
    const std::string &failure() const { return failure_; }
    bool failure_aborts() const { return failure_aborts_; }              // the reference ends in abort() there, not in exit(1)
    void fail(const std::string &why, bool aborts = false) { failure_ = why; failure_aborts_ = aborts; }
    // What the reference's parser writes to stderr about a record (a name the header does not declare, sample columns that do not fit) goes to
    // stderr here too -- or into `sink`, one line per entry, for a caller that prints it where the reference would; quiet = nowhere.
    void notes_to(std::vector<std::string> *sink) { sink_ = sink; quiet_ = false; }
    void silence() { sink_ = nullptr; quiet_ = true; }
    void note(const std::string &line);

    size_t n_samples() const { return samples_.size(); }
    bool knows_contig(const std::string &name) const { return contig_number_.count(name) != 0; }
    // "" for a number nothing was registered under
    const std::string &contig_name(int number) const { return name_in(contig_names_, number); }
    const std::string &id_name(int number) const { return name_in(id_names_, number); }
    const Id *find_id(const std::string &name) const { auto it = ids_.find(name); return it == ids_.end() ? nullptr : &it->second; }
    // Names a record uses without a declaration join the dictionary with a warning (R1, R6, R7, S1) -- the header has been printed by then, so
    // only the numbering sees them.  -1 / an Id without a number when even the made-up declaration does not scan.
    int contig_for(const std::string &name);
    Id id_for(const std::string &name, Role role);

  private:
    struct Attr { std::string name, text; };                      // text keeps its quotes
    struct Entry {
        enum Class { kGeneric, kFilterDecl, kInfoDecl, kFormatDecl, kContigDecl, kStructured };
        std::string tag, plain;                                   // ##tag=plain
        bool angle = false;                                       // ##tag=<attrs>
        std::vector<Attr> attrs;
        Class cls = kGeneric;
    };
    static int scan_line(const std::string &s, size_t from, Entry &e, size_t &next);      // 1: scanned, 0: not a "##" line, -1: one that does not scan
    void unscannable(const std::string &s, size_t from, size_t next);
    void admit(Entry &&e);
    bool admit_contig(const Entry &e);
    bool admit_id(const Entry &e, Role role);
    bool claim(std::vector<std::string> &names, int &number, const std::string &name);
    void read_column_line(const std::string &s, size_t from);
    static const std::string &name_in(const std::vector<std::string> &names, int number) {
        static const std::string none;
        return number >= 0 && (size_t)number < names.size() ? names[(size_t)number] : none;
    }

    std::vector<Entry> entries_;                                  // output order
    std::unordered_map<std::string, Id> ids_;
    std::vector<std::string> id_names_;
    std::unordered_map<std::string, int> contig_number_;
    std::vector<std::string> contig_names_;
    std::vector<std::string> samples_;
    std::string failure_;
    bool failure_aborts_ = false;
    bool pl_is_per_genotype_ = false;                             // the FORMAT declaration of PL says Number=G
    std::vector<std::string> *sink_ = nullptr;
    bool quiet_ = false;
};

// ---- one record ------------------------------------------------------------------------------------------------------------------------------
// A vector of values of one storage class.  Integers keep the byte width they are stored with: the two reserved codes of a width (its most
// negative number = "missing", the next one = "no more values") are what P2 / P6 test for.
struct VcfValue {
    enum Store : uint8_t { kNone, kBytes, kInts, kReals, kOpaque };
    Store store = kNone;
    uint8_t width = 0;                  // kInts: 1, 2 or 4
    int count = 0;                      // values (INFO: all of them; a sample field: per sample)
    std::string bytes;                  // kBytes / kOpaque
    std::vector<int32_t> ints;          // kInts, sign-extended
    std::vector<uint32_t> reals;        // kReals, IEEE single bit patterns
};

struct VcfRecord {
    int contig = -1;
    int32_t pos0 = 0;
    uint32_t qual = 0x7F800001u;        // bit pattern; this one = missing
    bool past_pos = false;              // the line had more than CHROM and POS (R3)
    bool id_seen_before = false;        // an earlier record of the same file had an ID column (set by the caller; P3)
    std::string id;                     // "" = none
    std::vector<std::string> alleles;   // REF first
    std::vector<int> filters;           // id numbers
    struct Tagged { int key = -1; VcfValue v; };
    std::vector<Tagged> info, fields;   // fields: the sample columns' FORMAT keys, values of all samples back to back
    int n_samples = 0;
};

enum class ReadResult { kOk, kRefused /* the reference's reader stops at this record */, kFatal /* dict.failure() says why */ };

// one text line without its '\n'.  names_only: what the reference's parser would SAY about the line and whether it reads it (the names the line uses, the
// shape of its sample columns) without converting a value -- rec is not usable afterwards.
ReadResult read_text_record(VcfDictionary &dict, const char *line, size_t len, VcfRecord &rec, bool names_only = false);
// p = the record's first byte (its two length words), avail = bytes readable.  Returns the record's size, 0 when it is cut short or damaged.
size_t read_bcf_record(const uint8_t *p, size_t avail, VcfRecord &rec);
// R12: the INFO entry `key` becomes the text `value` where it stands, or is appended.  false = the header declares no such INFO id.
bool set_info_text(const VcfDictionary &dict, VcfRecord &rec, const std::string &key, const std::string &value);
// appends the record's line and '\n'.  false = not printed (its sample count is not the header's, P7).
bool write_text_record(const VcfDictionary &dict, const VcfRecord &rec, std::string &out);

}  // namespace rgx
