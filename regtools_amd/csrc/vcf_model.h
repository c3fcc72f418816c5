// vcf_model.h -- the VCF/BCF header dictionary, the typed record and the text formatter behind the annotated-VCF output
// (`cis-splice-effects identify -v`, `variants annotate -o`).
//
// The reference writes that file through htslib: bcf_hdr_read -> bcf_hdr_append x4 -> bcf_hdr_write, then per record
// bcf_read -> bcf_update_info_string x4 -> bcf_write (variants_annotator.cc:118-154, 521-537).  Every record is therefore
// PARSED into BCF's typed form and RE-SERIALISED (vcf.c:1782 vcf_parse, :2069 vcf_format): floats come back as "%g" of a
// 32-bit float, integers as plain decimals, trailing FORMAT fields are filled in, header lines are de-duplicated and re-formatted.
// This file restates that model (host code, plain C++; nothing here runs on the device):
//   VcfHdr   -- bcf_hdr_parse / bcf_hdr_parse_line / bcf_hdr_register_hrec / bcf_hdr_fmt_text (vcf.c:262-620, 1334-1376)
//   VcfRec   -- one record in BCF's typed layout, from a text line (vcf.c:1535-1956) or from a BCF record (vcf.c:899-927, 1966-2068)
//   update_info_string / format -- bcf_update_info (vcf.c:2783-2868), vcf_format (vcf.c:2069-2164)
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace rgx {

enum { VT_NULL = 0, VT_INT8 = 1, VT_INT16 = 2, VT_INT32 = 3, VT_FLOAT = 5, VT_CHAR = 7 };     // BCF_BT_*
enum { HT_FLAG = 0, HT_INT = 1, HT_REAL = 2, HT_STR = 3 };                                  // BCF_HT_*
enum { HL_FLT = 0, HL_INFO = 1, HL_FMT = 2, HL_CTG = 3, HL_STR = 4, HL_GEN = 5 };           // BCF_HL_*

struct VcfHdr {
    struct Line {                       // bcf_hrec_t
        std::string key, value;         // generic line: ##key=value
        bool structured = false;        // ##key=<k=v,...>
        std::vector<std::pair<std::string, std::string>> kv;   // values keep their quotes
        int type = HL_GEN;
    };
    struct Tag { int id = -1; bool has[3] = {false, false, false}; int vtype[3] = {-1, -1, -1}; };   // FILTER / INFO / FORMAT entry of one ID
    std::vector<Line> lines;            // in output order
    std::unordered_map<std::string, Tag> tags;   std::vector<std::string> tag_name;      // BCF_DT_ID
    std::unordered_map<std::string, int> contigs; std::vector<std::string> contig_name;  // BCF_DT_CTG
    std::vector<std::string> samples;
    std::string error;                  // set when upstream would have stopped (conflicting IDX, duplicated sample)

    // text = the header lines, each ending in '\n', the #CHROM line last (vcf_hdr_read) or the BCF header block
    void parse(const std::string &text);
    bool append(const std::string &line);            // bcf_hdr_append
    void format(std::string &out) const;             // bcf_hdr_fmt_text(is_bcf = 0)
    int tag_id(const std::string &name) const { auto it = tags.find(name); return it == tags.end() ? -1 : it->second.id; }
    // what vcf_parse does with names the header does not declare: a dummy line joins the dictionary (the header was written before)
    int contig_or_add(const std::string &name);
    const Tag &tag_or_add(const std::string &name, int hl);

  private:
    bool parse_line(const char *p, size_t &len, Line &out) const;
    int add(Line &&l);                               // bcf_hdr_add_hrec: 1 = dictionary changed
    int register_line(Line &l);
    bool set_idx(std::vector<std::string> &names, int &id, const std::string &tag);
};

struct VcfRec {
    struct Typed { int type = VT_NULL; int n = 0; std::string data; };       // n values of `type` (INFO: the whole vector; FORMAT: per sample)
    int32_t rid = 0, pos = 0;
    uint32_t qual_bits = 0x7F800001u;   // bcf_float_missing
    bool have_shared = false;           // false = only CHROM / POS were present (bcf_unpack leaves everything at its cleared state)
    bool id_buffer_used = false;        // an earlier record of the file had an ID column (set by the writer): see VcfText::first_with_id
    Typed id;
    std::vector<Typed> alleles;
    std::vector<int32_t> flt;
    struct Info { int key; Typed v; };
    std::vector<Info> info;
    struct Fmt { int key; Typed v; };   // v.n values per sample, n_sample samples back to back
    std::vector<Fmt> fmt;
    int n_sample = 0;
};

// One text record line (no '\n').  Returns 0, or -1 where vcf_parse returns an error (the reference's read loop ends there).
int vcf_parse_line(VcfHdr &h, const char *line, size_t len, VcfRec &r);
// One BCF record: `p` points at its 32-byte fixed part, `avail` bytes are readable.  Returns the record's total length, 0 when truncated.
size_t bcf_parse_record(const uint8_t *p, size_t avail, VcfRec &r);
// bcf_update_info_string: replaces the first INFO entry of that key or appends one.  false = the header has no such INFO tag.
bool vcf_update_info_string(const VcfHdr &h, VcfRec &r, const std::string &key, const std::string &value);
// vcf_format: appends the line including '\n'.  false = bcf_write refuses the record (its sample count is not the header's).
bool vcf_format_line(const VcfHdr &h, const VcfRec &r, std::string &out);

}  // namespace rgx
