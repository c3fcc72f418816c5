// cse_host.h -- host-side text containers of `cis-splice-effects identify`: GTF -> flat arrays, VCF
// lines, FASTA random access.  Parsing only; every interval computation happens in the kernels (cse_core.h).
#pragma once
#include <stdint.h>

#include <stdio.h>

#include <functional>
#include <memory>
#include <string>
#include "bigvec.h"
#include "vcf_rewrite.h"
#include <unordered_map>
#include <vector>

namespace rgx {

// gtf/gtf_parser.cc:63-263
struct GtfModel {
    std::vector<std::string> chroms;                      // GTF contig names, first-seen order
    std::unordered_map<std::string, int32_t> chrom_index;
    // transcripts in std::map<string,...> order (ascending id)
    // (BigVec: large blocks on huge pages, integers not zeroed by resize() -- the threads that fill the tables take the page faults)
    BigVec<std::string> tx_id, tx_gene_name, tx_gene_id;
    BigVec<int32_t>  tx_chrom;
    BigVec<uint8_t>  tx_strand;
    BigVec<uint32_t> tx_exon_off, tx_n_exons, tx_bin;
    BigVec<uint32_t> es, ee;                               // strand-sorted per transcript
    BigVec<uint64_t> bin_key;                              // (chrom << 32 | bin), sorted, ties in transcript order
    BigVec<uint32_t> bin_tx;
    BigVec<uint32_t> bin_start;                            // contigs x bin_stride + 1 entries: where (contig, bin)'s entries begin (empty: too many to index)
    uint32_t bin_stride = 0;
    // What load() built the tables from -- the parts' record lists, the merge's arrays, the mapping of the text: taking them down is page-table work
    // (13 ms for a GENCODE-scale file) that load()'s caller need not wait for.  They go with the model, or when release_load_scratch() says so
    // (rgx_gtf_load: a model that lives long); `identify` hands the whole model to the process's background thread once its outputs are written.
    std::vector<std::shared_ptr<void>> load_scratch;
    void release_load_scratch() { load_scratch.clear(); }
    // returns "" on success, else the message the reference would die with
    std::string load(const std::string &path);
    int32_t chrom_of(const std::string &name) const { auto it = chrom_index.find(name); return it == chrom_index.end() ? -1 : it->second; }
};

// vcf.c:1782-1958: only CHROM and POS are consumed; the text is kept for the -v pass-through
struct VcfText {
    std::string text;                  // the (inflated) file: VCF text, or a BCF stream
    std::vector<size_t> line_off;      // text: n_lines + 1 entries
    struct Rec { size_t line; std::string chrom; uint32_t pos0; };      // line = line index (text) / byte offset of the record (BCF)
    std::vector<Rec> recs;             // what the reference's read loop hands out, in file order (it ends at the first record vcf_parse refuses)
    bool bcf = false;
    size_t first_with_id = (size_t)-1; // first record that carries an ID column: bcf_unpack gives the reader's record its ID buffer there, and a
                                       // later record WITHOUT the column prints that buffer, emptied, where an earlier one prints "." (vcf.c:2012-2018, :2075)
    VcfDictionary hdr;                 // the header as the reference's reader holds it (vcf_rewrite.h)
    // What the reference's parser says on stderr about the records it reads, each line with the index of the record that draws it (recs.size() = the
    // record the read loop ends at): names the header does not declare (once per name, vcf.c:1829-1838, 1877-1887, 1908-1919, 1561-1571), sample columns
    // that do not fit.  load() finds them -- it looks at every record's names -- and the caller prints them where the reference would (flush_notes).
    std::vector<std::pair<size_t, std::string>> notes;
    mutable size_t notes_printed = 0;
    // the first `upto` records' lines that are not out yet (one thread at a time)
    void flush_notes(size_t upto) const { for (; notes_printed < notes.size() && notes[notes_printed].first < upto; ++notes_printed) fputs(notes[notes_printed].second.c_str(), stderr); }
    // the record behind the last one ends the PROCESS upstream (exit(1) in the middle of vcf_parse, or abort()): its message, without the '\n'
    std::string fatal;
    bool fatal_aborts = false;
    // load()'s message is what htslib printed before it ended the process itself: 1 = with exit(1), 2 = with abort() (0: the tool's own error path)
    int death = 0;
    // annotating = the four INFO lines of the annotated VCF are declared before the records are read (variants_annotator.cc:130-154; `identify`
    // without -v does not do that, identifier.cc:263-264): a record that carries one of those keys draws no line then
    std::string load(const std::string &path, bool annotating = true);
    size_t n_lines() const { return line_off.empty() ? 0 : line_off.size() - 1; }
    void line(size_t i, const char *&p, size_t &len) const;
    // record i as typed values (h = a private copy of hdr: names the header does not declare join it)
    ReadResult typed(size_t i, VcfDictionary &h, VcfRecord &r) const;
};

// The annotated VCF: header with the four INFO lines appended, then records `todo` (indices into vcf.recs, ascending), each with the four
// INFO values annot(i) supplies (all nullptr = "NA") -- variants_annotator.cc:130-154, 521-533 through htslib's typed round trip.
// Returns "" or the message upstream stops with.
struct VcfAnnot { const std::string *genes, *transcripts, *distances, *annotations; };
// print_notes: vcf.notes go to stderr from here, a record's lines in front of what writing it says (the caller prints none of them itself).
std::string write_annotated_vcf_records(FILE *fv, const VcfText &vcf, const std::vector<size_t> &todo, const std::function<VcfAnnot(size_t)> &annot, bool print_notes = true);

// faidx.c:288-413 (uncompressed FASTA + .fai, the index is built in memory when the file is missing)
struct Fasta {
    struct Seq { std::string name; int64_t len, offset; int line_blen, line_len; };
    const char *data = nullptr;   // the file, mmap'd read-only (lookups touch a few pages of a multi-GB genome)
    size_t size = 0;
    void unmap();
    std::vector<Seq> seqs;
    Fasta() = default;
    Fasta(const Fasta &) = delete;
    Fasta &operator=(const Fasta &) = delete;
    ~Fasta();
    bool load(const std::string &path);
    // fai_fetch("name:beg1-end1"): returns false when the contig is unknown
    bool fetch(const std::string &name, int64_t beg1, int64_t end1, std::string &out) const;
};

std::string rev_comp(const std::string &s);   // utils/common.h:59-83

// BED12 junction rows the way bedtools' BedFile::GetNextBed hands them out (bedFile.cpp:103-260, bedFile.h:565-780) after
// JunctionsAnnotator::adjust_junction_ends (junctions_annotator.cc:66-81): start/end are the intron, ts/te the BED start/end.
struct BedJunctions {
    std::vector<std::string> chrom, name, score, strand, color;
    std::vector<uint32_t> start, end, ts, te;
    std::vector<int32_t> nblocks;
    size_t n() const { return chrom.size(); }
    // "" on success.  On a malformed line the rows before it are kept and the message is what the reference dies with
    // (exit status 1 either way).
    std::string load(const std::string &path);
};

}  // namespace rgx
