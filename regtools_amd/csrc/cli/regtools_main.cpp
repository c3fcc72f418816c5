// regtools_main.cpp -- host CLI in front of libregtools_amd.so: `regtools-amd junctions extract ...`.
//
// Keeps the reference's sub-command surface for the accelerated path: same flags, defaults, stderr echo and exit
// codes as /root/reference/src/junctions/junctions_extractor.cc:42-143 (parse_options/usage),
// src/junctions/junctions_main.cc:45-107 (dispatch, exception -> exit code) and src/regtools.cc:36-74
// (banner, top-level usage).  Everything data-parallel happens behind the C ABI (include/regtools_amd.h).
#include <errno.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <memory>
#include <vector>

#include "regtools_amd.h"

namespace {

// An empty line in a GTF: upstream's loader calls line.at(0) on it outside any try block (gtf_parser.cc:230), the std::out_of_range is nobody's to catch,
// the process prints libstdc++'s terminate message and aborts (status 134).  The library reports the line as an error; the tool then does what upstream
// does -- the same call, uncaught (the handlers around it take std::runtime_error only) -- so the message and the status are the reference's.
// (`cis-splice-effects identify / associate` catch std::exception around everything, cis_splice_effects_main.cc:35-51, :55-71 (std::logic_error): there the
// same exception's
// what() is the message and the status is 1 -- caught = true)
void die_as_upstream_on_empty_gtf_line(const char *err, bool caught = false) {
    if (strcmp(err, "basic_string::at")) return;
    if (!caught) { std::cerr.flush(); fflush(nullptr); (void)std::string().at(0); }
    try { (void)std::string().at(0); } catch (const std::out_of_range &e) { throw std::runtime_error(e.what()); }
}

// htslib ends the process itself on two kinds of VCF record (vcf.c:1610-1614 exit(1); :1638-1639 abort()), past every handler of the tool: the
// library reports them (RGX_ERR_EXIT / RGX_ERR_ABORT with what htslib printed) and the tool goes the same way.
void die_where_upstreams_library_does(int rc, const char *err) {
    if (rc != RGX_ERR_EXIT && rc != RGX_ERR_ABORT) return;
    // (what htslib printed; the library's own words -- a read whose aux fields bam_aux_get abort()s on, sam.c:1233-1252 -- are not upstream's: nothing is
    // printed there)
    std::cerr.flush(); if (strncmp(err, "regtools_amd:", 13)) fputs(err, stderr); fflush(nullptr);
    if (rc == RGX_ERR_ABORT) abort();
    exit(1);
}

struct HelpRequested { std::string text; };

struct ExtractOptions {
    std::string bam = "NA", ref = "NA", output = "NA", barcodes = "NA", region = ".", tag = "XS";
    uint32_t min_anchor = 8, min_intron = 70, max_intron = 500000;
    int strandness = -1;
    int device = 0;
};

void extract_usage(std::ostream &out) {
    out << "Usage:\t\tregtools junctions extract [options] indexed_alignments.bam\n"
        << "Options:\n"
        << "\t\t-a INT\tMinimum anchor length. Junctions which satisfy a minimum \n\t\t\t anchor length on both sides are reported. [8]\n"
        << "\t\t-m INT\tMinimum intron length. [70]\n"
        << "\t\t-M INT\tMaximum intron length. [500000]\n"
        << "\t\t-o FILE\tThe file to write output to. [STDOUT]\n"
        << "\t\t-r STR\tThe region to identify junctions \n\t\t\t in \"chr:start-end\" format. Entire BAM by default.\n"
        << "\t\t-s INT\tStrandness mode \n\t\t\t XS, use XS tags provided by aligner; RF, first-strand; FR, second-strand. REQUIRED\n"
        << "\t\t-t STR\tTag used in bam to label strand. [XS]\n"
        << "\t\t-b STR\tThe file containing the barcodes of interest for single cell data.\n\n";
}

// junctions_extractor.cc:42-122
ExtractOptions parse_extract(int argc, char **argv) {
    ExtractOptions o;
    optind = 1;
    int c;
    while ((c = getopt(argc, argv, "ha:m:M:o:r:t:s:b:")) != -1) {
        switch (c) {
            case 'h': { std::ostringstream ss; extract_usage(ss); throw HelpRequested{ss.str()}; }
            case 'a': o.min_anchor = (uint32_t)atoi(optarg); break;
            case 'm': o.min_intron = (uint32_t)atoi(optarg); break;
            case 'M': o.max_intron = (uint32_t)atoi(optarg); break;
            case 'o': o.output = optarg; break;
            case 'r': o.region = optarg; break;
            case 't': o.tag = optarg; break;
            case 's': {
                std::string s = optarg;
                if (s == "XS") o.strandness = 0; else if (s == "RF") o.strandness = 1; else if (s == "FR") o.strandness = 2;
                else if (s == "intron-motif") o.strandness = 3;
                else throw std::runtime_error("Unrecognized strandness argument!\n\n");
                break;
            }
            case 'b': o.barcodes = optarg; break;
            default: extract_usage(std::cerr); throw std::runtime_error("Error parsing inputs!(1)\n\n");
        }
    }
    if (argc - optind >= 1) o.bam = argv[optind++];
    if (argc - optind >= 1) o.ref = argv[optind++];
    if (optind < argc || o.bam == "NA") { extract_usage(std::cerr); throw std::runtime_error("Error parsing inputs!(2)\n\n"); }
    if (o.strandness == -1) { extract_usage(std::cerr); throw std::runtime_error("Please supply strandness mode with '-s' option!\n\n"); }
    if (o.strandness == 3 && o.ref == "NA") { extract_usage(std::cerr); throw std::runtime_error("Strandness mode 'intron-motif' requires a fasta file!\n\n"); }
    std::cerr << "Minimum junction anchor length: " << o.min_anchor << "\nMinimum intron length: " << o.min_intron
              << "\nMaximum intron length: " << o.max_intron << "\nAlignment: " << o.bam << "\nOutput file: " << o.output << "\n";
    if (o.barcodes != "NA") std::cerr << "Barcode file: " << o.barcodes << "\n";
    std::cerr << std::endl;
    return o;
}

// junctions_main.cc:45-59
double wall_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

int junctions_extract(int argc, char **argv) {
    try {
        const double t_start = wall_ms();
        ExtractOptions o = parse_extract(argc, argv);
        char err[512] = {0};
        rgx_ctx *ctx = nullptr;
        if (const char *d = getenv("REGTOOLS_AMD_DEVICE")) o.device = atoi(d);
        // REGTOOLS_AMD_DEVICES=0,1,2,...: the file is sharded over these GPUs (rgx_extract_multi: a host thread per device, one RCCL
        // all-gather of the shards' rows, merge on the first); the output does not depend on the list
        std::vector<int> devices;
        if (const char *d = getenv("REGTOOLS_AMD_DEVICES")) {
            for (const char *q = d; *q;) { char *e; long v = strtol(q, &e, 10); if (e == q) break; devices.push_back((int)v); q = *e == ',' ? e + 1 : e;
                if (*e && *e != ',') break; }
        }
        if (devices.size() == 1) o.device = devices[0];
        if (devices.size() <= 1 && rgx_ctx_create(o.device, &ctx, err, sizeof err) != RGX_OK) throw std::runtime_error(err);
        const double t_ctx = wall_ms();
        rgx_extract_params p;
        rgx_extract_params_default(&p);
        p.region = o.region.c_str(); p.strandness = o.strandness;
        p.strand_tag[0] = o.tag.size() > 0 ? o.tag[0] : 0; p.strand_tag[1] = o.tag.size() > 1 ? o.tag[1] : 0;
        p.min_anchor = o.min_anchor; p.min_intron = o.min_intron; p.max_intron = o.max_intron;
        p.fasta_path = o.ref == "NA" ? nullptr : o.ref.c_str();
        p.barcodes = o.barcodes != "NA";                                   // -b (junctions_extractor.cc:82-84, :393-395)
        rgx_junction_table *t = nullptr;
        int rc = devices.size() > 1 ? rgx_extract_multi(devices.data(), (int)devices.size(), o.bam.c_str(), &p, &t, err, sizeof err)
                                    : rgx_extract(ctx, o.bam.c_str(), &p, &t, err, sizeof err);
        // (an aux field of unknown type in front of a spliced read's strand tag: upstream's bam_aux_get abort()s, sam.c:1248 -- nothing printed, SIGABRT)
        if (rc == RGX_ERR_ABORT) { std::cerr.flush(); fflush(nullptr); abort(); }
        if (rc != RGX_OK) { if (ctx) rgx_ctx_destroy(ctx); throw std::runtime_error(err); }
        const double t_extract = wall_ms();
        // one formatting pass: a row is a contig name + at most 160 bytes of numbers (Junction::print, junctions_extractor.h:90-98)
        size_t max_name = 0;
        for (int32_t i = 0; i < t->n_ref; ++i) max_name = std::max(max_name, strlen(t->ref_name[i]));
        // (uninitialised on purpose: a zero-filled vector of that size -- 51 MB for 300 k rows, of which 30 are written -- was 10 of the process's 250 ms)
        size_t text_cap = (size_t)t->n * (max_name + 160) + 1;
        std::unique_ptr<char[]> text(new char[text_cap]);
        size_t n = rgx_table_format_bed12(t, 1, text.get(), text_cap);
        if (n > text_cap) { text_cap = n + 1; text.reset(new char[text_cap]); n = rgx_table_format_bed12(t, 1, text.get(), text_cap); }
        const double t_format = wall_ms();
        FILE *f = o.output == "NA" ? stdout : fopen(o.output.c_str(), "w");
        // (an output file that cannot be opened is skipped as upstream's ofstream would; a write that comes up SHORT -- full disk, closed pipe --
        //  is an error here: the process ends with _exit below, nothing later could report it)
        bool short_write = false;
        if (f) { short_write = fwrite(text.get(), 1, n, f) != n; if (f != stdout) short_write |= fclose(f) != 0; else short_write |= fflush(f) != 0; }
        const double t_write = wall_ms();
        if (p.barcodes) {                                                  // print_all_junctions: an unopenable file is skipped silently (cc:255-256, :272)
            if (FILE *b = fopen(o.barcodes.c_str(), "w")) {
                const size_t nb = rgx_table_format_barcodes(t, 1, nullptr, 0);
                std::vector<char> bt(nb + 1);
                rgx_table_format_barcodes(t, 1, bt.data(), nb);
                short_write |= fwrite(bt.data(), 1, nb, b) != nb; short_write |= fclose(b) != 0;
            }
        }
        if (short_write) { fprintf(stderr, "regtools-amd: writing the output failed (%s)\n", strerror(errno)); fflush(stderr); _exit(1); }
        if (getenv("REGTOOLS_AMD_STATS"))
            fprintf(stderr, "[regtools_amd] records=%llu events=%llu junctions=%llu inflate=%.3fms records=%.3fms scan=%.3fms reduce=%.3fms total=%.3fms\n",
                    (unsigned long long)t->n_records, (unsigned long long)t->n_events, (unsigned long long)t->n, t->ms_inflate, t->ms_records,
                    t->ms_scan, t->ms_reduce, t->ms_total),
            fprintf(stderr, "[regtools_amd] process: options %.1f ms, context %.1f ms, extract %.1f ms (pipeline %.1f), format %.1f ms, write %.1f ms\n",
                    0.0, t_ctx - t_start, t_extract - t_ctx, t->ms_total, t_format - t_extract, t_write - t_format);
        // the outputs are on disk: leave without handing gigabytes of device memory back one buffer at a time and without the runtime's
        // orderly shutdown (both happen anyway when the process ends; ~0.1 s of a 0.3 s run)
        fflush(stdout); fflush(stderr);
        _exit(0);
    } catch (const HelpRequested &h) {
        std::cerr << h.text << std::endl;
        return 0;
    } catch (const std::runtime_error &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}

// ---- junctions annotate (junctions_annotator.cc:385-437, junctions_main.cc:62-93) ------------------------------------------------
void annotate_usage(std::ostream &out) {
    out << "Usage:\t\tregtools junctions annotate [options] junctions.bed ref.fa annotations.gtf\n"
        << "Options:\t-S include single exon genes\n"
        << "\t\t-o FILE\tThe file to write output to. [STDOUT]\n\n";
}

rgx_ctx *open_ctx() {
    char err[512] = {0};
    rgx_ctx *ctx = nullptr;
    int dev = 0; if (const char *d = getenv("REGTOOLS_AMD_DEVICE")) dev = atoi(d);
    if (rgx_ctx_create(dev, &ctx, err, sizeof err) != RGX_OK) throw std::runtime_error(err);
    return ctx;
}

int junctions_annotate(int argc, char **argv) {
    try {
        std::string out = "NA";
        bool skip_single = true;
        optind = 1;
        int c;
        while ((c = getopt(argc, argv, "So:h")) != -1) {
            switch (c) {
                case 'S': skip_single = false; break;
                case 'o': out = optarg; break;
                case 'h': { std::ostringstream ss; annotate_usage(ss); throw HelpRequested{ss.str()}; }
                default: annotate_usage(std::cerr); throw std::runtime_error("Error parsing inputs!(1)\n\n");
            }
        }
        std::string bed, ref = "NA", gtf;
        if (argc - optind >= 3) { bed = argv[optind++]; ref = argv[optind++]; gtf = argv[optind++]; }
        if (optind < argc || ref == "NA" || bed.empty() || gtf.empty()) { annotate_usage(std::cerr); throw std::runtime_error("Error parsing inputs!(2)\n\n"); }
        std::cerr << "Reference: " << ref << "\nGTF: " << gtf << "\nJunctions: " << bed << "\n" << (skip_single ? "Skipping single exon genes.\n" : "");
        if (out != "NA") std::cerr << "Output file: " << out << "\n";
        std::cerr << "\n";
        rgx_ctx *ctx = open_ctx();
        char err[512] = {0};
        uint64_t n = 0;
        int rc = rgx_junctions_annotate_opts(ctx, bed.c_str(), ref.c_str(), gtf.c_str(), out == "NA" ? nullptr : out.c_str(), (skip_single ? 0 :
            RGX_ANNOTATE_SINGLE_EXON) | RGX_ANNOTATE_ECHO, &n, err, sizeof err);
        rgx_ctx_destroy(ctx);
        if (rc != RGX_OK) { die_as_upstream_on_empty_gtf_line(err); throw std::runtime_error(err); }
        std::cerr << "\nAnnotated " << n << " lines.\n";
    } catch (const HelpRequested &h) {
        std::cerr << h.text << std::endl;
        return 0;
    } catch (const std::runtime_error &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}

int junctions_usage(std::ostream &out) {
    out << "Usage:\t\tregtools junctions <command> [options]\n"
        << "Command:\textract\t\tIdentify exon-exon junctions from alignments.\n"
        << "\t\tannotate\tAnnotate the junctions.\n\n";
    return 0;
}

// junctions_main.cc:96-107
int junctions_main(int argc, char **argv) {
    if (argc > 1) {
        std::string sub = argv[1];
        if (sub == "extract") return junctions_extract(argc - 1, argv + 1);
        if (sub == "annotate") return junctions_annotate(argc - 1, argv + 1);
    }
    return junctions_usage(std::cout);
}

// ---- cis-splice-effects identify (cis_splice_effects_identifier.cc:112-219, cis_splice_effects_main.cc:35-93) -------------
void space_options(std::ostream &out);

void window_options(std::ostream &out) {
    out << "\t\t-a INT\tMinimum anchor length. Junctions which satisfy a minimum \n\t\t\t anchor length on both sides are reported. [8]\n"
        << "\t\t-m INT\tMinimum intron length. [70]\n\t\t-M INT\tMaximum intron length. [500000]\n"
        << "\t\t-w INT\tWindow size in b.p to identify splicing events in.\n\t\t\t The tool identifies events in variant.start +/- w basepairs.\n"
        << "\t\t\t Default behaviour is to look at the window between previous and next exons.\n";
}

void identify_usage(std::ostream &out, bool associate = false) {
    out << "Usage:\t\tregtools cis-splice-effects " << (associate ? "associate [options] variants.vcf junctions.bed" :
        "identify [options] variants.vcf alignments.bam") << " ref.fa annotations.gtf\n"
        << "Options:\n"
        << "\t\t-o STR\tOutput file containing the aberrant splice junctions with annotations. [STDOUT]\n"
        << "\t\t-v STR\tOutput file containing variants annotated as splice relevant (VCF format).\n"
        << "\t\t-j STR\tOutput file containing the aberrant junctions in BED12 format.\n";
    if (!associate)
        out << "\t\t-s INT\tStrandness mode \n\t\t\t XS, use XS tags provided by aligner; RF, first-strand; FR, second-strand. intron-motif, infer strand using canonical intron motifs. REQUIRED\n"
            << "\t\t-C\tOverride strand assignments by inferring based on canonical motifs. Does not need to be specified if passing '-s intron-motif'.\n"
            << "\t\t-t STR\tTag used in bam to label strand. [XS]\n";
    window_options(out);
    space_options(out);
    if (!associate)
        out << "\t\t-b STR\tThe file containing the barcodes of interest for single cell data.\n"
            << "\t\t-C\tTells cis-splice-effects identify that you want intron-motif method to take priority when assigning strand. i.e. decide strandedness based on the fasta rather than what is encoded in the alignment file.\n";
    out << "\n";
}

bool file_exists(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (!f) return false; fclose(f); return true; }

// associate = true: `cis-splice-effects associate` (cis_splice_effects_associator.cc:104-180): no -s/-t/-b/-C, the second positional is a BED12
int cse_identify(int argc, char **argv, bool associate = false) {
    try {
        rgx_identify_params p;
        rgx_identify_params_default(&p);
        std::string out_tsv = "NA", out_vcf = "NA", out_bed = "NA", tag = "XS", barcodes = "NA";
        optind = 1;
        int c;
        while ((c = getopt(argc, argv, associate ? "o:w:v:j:e:Ei:ISha:m:M:" : "o:w:v:j:e:Ei:ISht:s:a:m:M:b:C")) != -1) {
            switch (c) {
                case 'o': out_tsv = optarg; break;
                case 'w': p.window = (uint32_t)atoi(optarg); break;
                case 'v': out_vcf = optarg; break;
                case 'j': out_bed = optarg; break;
                case 'i': p.intronic_min = (uint32_t)atoi(optarg); break;
                case 'e': p.exonic_min = (uint32_t)atoi(optarg); break;
                case 'I': p.all_intronic = 1; break;
                case 'E': p.all_exonic = 1; break;
                case 'S': p.skip_single = 0; break;
                case 'h': { std::ostringstream ss; identify_usage(ss, associate); throw HelpRequested{ss.str()}; }
                case 's': {
                    std::string s = optarg;
                    if (s == "XS") p.strandness = 0; else if (s == "RF") p.strandness = 1; else if (s == "FR") p.strandness = 2;
                    else if (s == "intron-motif") p.strandness = 3;
                    else throw std::runtime_error("Unrecognized strandness argument!\n\n");
                    break;
                }
                case 't': tag = optarg; break;
                case 'a': p.min_anchor = (uint32_t)atoi(optarg); break;
                case 'm': p.min_intron = (uint32_t)atoi(optarg); break;
                case 'M': p.max_intron = (uint32_t)atoi(optarg); break;
                case 'b': barcodes = optarg; break;
                case 'C': p.override_motif = 1; break;
                default: identify_usage(std::cerr, associate); throw std::runtime_error("Error parsing inputs!(1)\n\n");
            }
        }
        std::string vcf = "NA", bam = "NA", ref = "NA", gtf = "NA";
        if (argc - optind >= 4) { vcf = argv[optind++]; bam = argv[optind++]; ref = argv[optind++]; gtf = argv[optind++]; }
        if (optind < argc || vcf == "NA" || bam == "NA" || ref == "NA" || gtf == "NA") { identify_usage(std::cerr, associate);
            throw std::runtime_error("Error parsing inputs!(2)\n\n"); }
        if (associate) p.strandness = 0;
        if (p.strandness == -1) { identify_usage(std::cerr); throw std::runtime_error("Please supply strand specificity with '-s' option!\n\n"); }
        if (!file_exists(vcf) || !file_exists(bam) || !file_exists(ref) ||
            !file_exists(gtf)) throw std::runtime_error("Please make sure input files exist.\n\n");
        // the echo of parse_options (identifier.cc:203-218, associator.cc:156-171), then what identify() / associate() write while they work (p.echo)
        std::cerr << "Variant file: " << vcf << (associate ? "\nJunctions BED file: " :
            "\nAlignment file: ") << bam << "\nReference fasta file: " << ref << "\nAnnotation file: " << gtf << "\n";
        if (p.window != 0) std::cerr << "Window size: " << p.window << "\n";
        if (out_tsv != "NA") std::cerr << "Output file: " << out_tsv << "\n";
        if (out_bed != "NA") std::cerr << "Output junctions BED file: " << out_bed << "\n";
        if (out_vcf != "NA") std::cerr << "Annotated variants file: " << out_vcf << "\n";
        std::cerr << "\n";
        p.echo = 1;
        if (associate) p.bed_path = bam.c_str();
        p.vcf_path = vcf.c_str(); p.bam_path = bam.c_str(); p.fasta_path = ref.c_str(); p.gtf_path = gtf.c_str();
        p.out_tsv = out_tsv == "NA" ? nullptr : out_tsv.c_str(); p.out_vcf = out_vcf == "NA" ? nullptr : out_vcf.c_str(); p.out_bed = out_bed == "NA" ?
            nullptr : out_bed.c_str();
        p.strand_tag[0] = tag.size() > 0 ? tag[0] : 0; p.strand_tag[1] = tag.size() > 1 ? tag[1] : 0;
        char err[512] = {0};
        rgx_ctx *ctx = nullptr;
        int dev = 0; if (const char *d = getenv("REGTOOLS_AMD_DEVICE")) dev = atoi(d);
        // REGTOOLS_AMD_DEVICES=0,1,...: identify's extraction is sharded over these GPUs (rgx_identify_multi); the outputs do not depend on the list
        std::vector<int> devices;
        if (const char *d = getenv("REGTOOLS_AMD_DEVICES")) {
            for (const char *q = d; *q;) { char *e; long v = strtol(q, &e, 10); if (e == q) break; devices.push_back((int)v); q = *e == ',' ? e + 1 : e;
                if (*e && *e != ',') break; }
        }
        if (devices.size() == 1) dev = devices[0];
        const bool multi = !associate && devices.size() > 1;
        if (!multi && rgx_ctx_create(dev, &ctx, err, sizeof err) != RGX_OK) throw std::runtime_error(err);
        rgx_identify_stats st;
        int rc = associate ? rgx_associate(ctx, &p, &st, err, sizeof err)
                           : multi ? rgx_identify_multi(devices.data(), (int)devices.size(), &p, &st, err, sizeof err) : rgx_identify(ctx, &p, &st, err,
                               sizeof err);
        if (ctx) rgx_ctx_destroy(ctx);
        die_where_upstreams_library_does(rc, err);
        if (rc != RGX_OK) { die_as_upstream_on_empty_gtf_line(err, /*caught=*/true); throw std::runtime_error(err); }
        if (barcodes != "NA") {
            // identify's extractor is built without a barcode file (identifier.cc:288 -> junctions_extractor.h:197-205), so every junction's map
            // is empty and print_barcodes (identifier.cc:239-241) writes "0\t" per junction; an unopenable file ends the run (set_ostream :90-95)
            FILE *b = fopen(barcodes.c_str(), "w");
            if (!b) throw std::runtime_error("Unable to open " + barcodes);
            for (uint64_t i = 0; i < st.n_junctions; ++i) fputs("0\t\n", b);
            fclose(b);
        }
        if (getenv("REGTOOLS_AMD_STATS"))
            fprintf(stderr,
                "[regtools_amd] variants=%llu relevant=%llu windows=%llu pairs=%llu junctions=%llu total=%.3fms (gtf %.3f variants %.3f extract %.3f join %.3f annotate %.3f output %.3f)\n",
                    (unsigned long long)st.n_variants, (unsigned long long)st.n_relevant, (unsigned long long)st.n_windows, (unsigned long long)st.n_pairs,
                    (unsigned long long)st.n_junctions, st.ms_total, st.ms_gtf, st.ms_variants, st.ms_extract, st.ms_join, st.ms_annotate, st.ms_output);
    } catch (const HelpRequested &h) {
        std::cerr << h.text << std::endl;
        return 0;
    } catch (const std::runtime_error &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}

// ---- variants annotate (variants_annotator.cc:48-110, variants_main.cc) -------------------------------------------------------------
void space_options(std::ostream &out) {
    out << "\t\t-e INT\tMaximum distance from the start/end of an exon \n\t\t\t to annotate a variant as relevant to splicing, the variant \n\t\t\t is in exonic space, i.e a coding variant. [3]\n"
        << "\t\t-i INT\tMaximum distance from the start/end of an exon \n\t\t\t to annotate a variant as relevant to splicing, the variant \n\t\t\t is in intronic space. [2]\n"
        << "\t\t-I\tAnnotate variants in intronic space within a transcript(not to be used with -i).\n"
        << "\t\t-E\tAnnotate variants in exonic space within a transcript(not to be used with -e).\n"
        << "\t\t-S\tDon't skip single exon transcripts.\n";
}

void variants_usage(std::ostream &out) {
    out << "Usage:\t\tregtools variants annotate [options] variants.vcf annotations.gtf\n"
        << "Options:\n"
        << "\t\t-o FILE\tThe file to write output to. [STDOUT]\n";
    space_options(out);
    out << "\n";
}

int variants_annotate(int argc, char **argv) {
    try {
        rgx_identify_params p;
        rgx_identify_params_default(&p);
        std::string out = "NA";
        optind = 1;
        int c;
        while ((c = getopt(argc, argv, "e:Ei:ISho:")) != -1) {
            switch (c) {
                case 'i': p.intronic_min = (uint32_t)atoi(optarg); break;
                case 'e': p.exonic_min = (uint32_t)atoi(optarg); break;
                case 'I': p.all_intronic = 1; break;
                case 'E': p.all_exonic = 1; break;
                case 'S': p.skip_single = 0; break;
                case 'o': out = optarg; break;
                case 'h': { std::ostringstream ss; variants_usage(ss); throw HelpRequested{ss.str()}; }
                default: variants_usage(std::cout); throw std::runtime_error("Error parsing inputs!(1)\n\n");
            }
        }
        std::string vcf = "NA", gtf = "NA";
        if (argc - optind >= 2) { vcf = argv[optind++]; gtf = argv[optind++]; }
        if (optind < argc || vcf == "NA" || gtf == "NA") { variants_usage(std::cout); throw std::runtime_error("Error parsing inputs!(2)\n\n"); }
        p.vcf_path = vcf.c_str(); p.gtf_path = gtf.c_str(); p.out_vcf = out == "NA" ? nullptr : out.c_str();
        // variants_annotator.cc:93-108
        std::cerr << "Variant file: " << vcf << "\nGTF file: " << gtf << "\nOutput vcf file: " << out << "\n";
        if (!p.all_intronic) std::cerr << "Intronic min distance: " << p.intronic_min << "\n";
        if (!p.all_exonic) std::cerr << "Exonic min distance: " << p.exonic_min << "\n";
        if (!p.skip_single) std::cerr << "Not skipping single exon genes.\n";
        if (out != "NA") std::cerr << "Output file: " << out << "\n";
        std::cerr << "\n";
        rgx_ctx *ctx = open_ctx();
        char err[512] = {0};
        int rc = rgx_variants_annotate(ctx, &p, nullptr, err, sizeof err);
        rgx_ctx_destroy(ctx);
        die_where_upstreams_library_does(rc, err);
        if (rc != RGX_OK) { die_as_upstream_on_empty_gtf_line(err); throw std::runtime_error(err); }
    } catch (const HelpRequested &h) {
        std::cerr << h.text << std::endl;
        return 0;
    } catch (const std::runtime_error &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}

int variants_main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "annotate") return variants_annotate(argc - 1, argv + 1);
    std::cout << "Usage:\t\tregtools variants <command> [options]\nCommand:\tannotate\t\tAnnotate variants with splicing information.\n\n";
    return 0;
}

int cse_main(int argc, char **argv) {
    if (argc > 1) {
        std::string sub = argv[1];
        if (sub == "identify") return cse_identify(argc - 1, argv + 1);
        if (sub == "associate") return cse_identify(argc - 1, argv + 1, true);
    }
    std::cout << "Usage:\t\tregtools cis-splice-effects <command> [options]\nCommand:\tidentify\t\tIdentify cis splicing effects.\n\t\tassociate\tAssociate extracted junctions with variants\n\n";
    return 0;
}

}  // namespace

// regtools.cc:36-74: the banner and the top-level usage are the reference's bytes (pinned against the reference's own main() in tests/test_cli_contract.py);
// the library's own version string is rgx_version() (REGTOOLS_AMD_TRACE prints it)
int main(int argc, char **argv) {
    // this process makes one library call: the context does without streams of its own (api.cpp ensure_upload_streams)
    setenv("REGTOOLS_AMD_ONE_SHOT", "1", 0);
    std::cerr << "\nProgram:\tregtools\nVersion:\t1.0.0" << std::endl;
    if (getenv("REGTOOLS_AMD_TRACE")) std::cerr << "[rgx trace] " << rgx_version() << std::endl;
    if (argc > 1) {
        std::string sub = argv[1];
        if (sub == "junctions") return junctions_main(argc - 1, argv + 1);
        if (sub == "cis-splice-effects") return cse_main(argc - 1, argv + 1);
        if (sub == "variants") return variants_main(argc - 1, argv + 1);
        // listed by the usage text below, as upstream's; not part of this build (out of scope)
        if (sub == "cis-ase") {
            std::cerr << "regtools-amd: the cis-ase commands are not part of the MI355X build; use the reference binary for them\n";
            return 1;
        }
    }
    std::cerr << "Usage:\t\tregtools <command> [options]\n"
              << "Command:\tjunctions\t\tTools that operate on feature junctions (e.g. exon-exon junctions from RNA-seq).\n"
              << "\t\tcis-ase\t\t\tTools related to allele specific expression in cis.\n"
              << "\t\tcis-splice-effects\tTools related to splicing effects of variants.\n"
              << "\t\tvariants\t\tTools that operate on variants.\n\n";
    return 0;
}
