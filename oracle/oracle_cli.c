/* oracle/oracle_cli.c -- TEST INFRASTRUCTURE ONLY.
 * Command-line face of the CPU restatement, flag-compatible with `regtools junctions extract`
 * (junctions_extractor.cc:42-122), plus `time` used by bench.py's cpu_baseline leg. */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* `regtools cis-splice-effects identify` option surface (cis_splice_effects_identifier.cc:112-219); with associate = 1 the
 * `cis-splice-effects associate` one (cis_splice_effects_associator.cc:104-180): no -s/-t/-C, second positional is a BED */
static int identify_main(int argc, char **argv, int associate) {
    orc_cse_params p; orc_cse_default_params(&p);
    int c;
    optind = 2;
    while ((c = getopt(argc, argv, associate ? "o:w:v:j:e:Ei:ISa:m:M:" : "o:w:v:j:e:Ei:ISt:s:a:m:M:C")) != -1) {
        switch (c) {
            case 'o': p.out_tsv = optarg; break;
            case 'w': p.window = (uint32_t)atoi(optarg); break;
            case 'v': p.out_vcf = optarg; break;
            case 'j': p.out_bed = optarg; break;
            case 'i': p.intronic_min = (uint32_t)atoi(optarg); break;
            case 'e': p.exonic_min = (uint32_t)atoi(optarg); break;
            case 'I': p.all_intronic = 1; break;
            case 'E': p.all_exonic = 1; break;
            case 'S': p.skip_single = 0; break;
            case 't': p.strand_tag[0] = optarg[0]; p.strand_tag[1] = optarg[0] ? optarg[1] : 0; break;
            case 's':
                if (!strcmp(optarg, "XS")) p.strandness = 0; else if (!strcmp(optarg, "RF")) p.strandness = 1;
                else if (!strcmp(optarg, "FR")) p.strandness = 2; else if (!strcmp(optarg, "intron-motif")) p.strandness = 3;
                else { fprintf(stderr, "Unrecognized strandness argument!\n\n"); return 1; }
                break;
            case 'a': p.min_anchor = (uint32_t)atoi(optarg); break;
            case 'm': p.min_intron = (uint32_t)atoi(optarg); break;
            case 'M': p.max_intron = (uint32_t)atoi(optarg); break;
            case 'C': p.override_motif = 1; break;
            default: fprintf(stderr, "Error parsing inputs!(1)\n\n"); return 1;
        }
    }
    if (argc - optind >= 4) { p.vcf = argv[optind++]; p.bam = argv[optind++]; p.fasta = argv[optind++]; p.gtf = argv[optind++]; }
    if (optind < argc || !p.vcf) { fprintf(stderr, "Error parsing inputs!(2)\n\n"); return 1; }
    if (associate) { p.bed = p.bam; p.strandness = 0; }
    if (p.strandness == -1) { fprintf(stderr, "Please supply strand specificity with '-s' option!\n\n"); return 1; }
    const char *files[4] = {p.vcf, p.bam, p.fasta, p.gtf};
    for (int k = 0; k < 4; ++k) if (access(files[k], F_OK)) { fprintf(stderr, "Please make sure input files exist.\n\n"); return 1; }
    char err[256] = "";
    if (orc_identify(&p, err, sizeof err)) { fputs(err, stderr); return 1; }
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "identify")) return identify_main(argc, argv, 0);
    if (argc >= 2 && !strcmp(argv[1], "associate")) return identify_main(argc, argv, 1);
    if (argc >= 2 && !strcmp(argv[1], "junctions-annotate")) {          /* junctions_annotator.cc:385-428 */
        const char *out = NULL; int c, keep_single = 0; optind = 2;
        while ((c = getopt(argc, argv, "So:")) != -1) {
            if (c == 'o') out = optarg;
            else if (c == 'S') keep_single = 1;
            else { fprintf(stderr, "Error parsing inputs!(1)\n\n"); return 1; }
        }
        if (argc - optind != 3) { fprintf(stderr, "Error parsing inputs!(2)\n\n"); return 1; }
        char err[512] = "";
        if (orc_junctions_annotate_opts(argv[optind], argv[optind + 1], argv[optind + 2], out, keep_single, err, sizeof err)) { fputs(err, stderr); return 1; }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "variants-annotate")) {           /* variants_annotator.cc:48-110 */
        orc_cse_params p; orc_cse_default_params(&p);
        int c; optind = 2;
        while ((c = getopt(argc, argv, "e:Ei:ISo:")) != -1) {
            switch (c) {
                case 'i': p.intronic_min = (uint32_t)atoi(optarg); break;
                case 'e': p.exonic_min = (uint32_t)atoi(optarg); break;
                case 'I': p.all_intronic = 1; break;
                case 'E': p.all_exonic = 1; break;
                case 'S': p.skip_single = 0; break;
                case 'o': p.out_vcf = optarg; break;
                default: fprintf(stderr, "Error parsing inputs!(1)\n\n"); return 1;
            }
        }
        if (argc - optind < 2) { fprintf(stderr, "Error parsing inputs!(2)\n\n"); return 1; }
        p.vcf = argv[optind]; p.gtf = argv[optind + 1];
        char err[512] = "";
        if (orc_variants_annotate(&p, err, sizeof err)) { fputs(err, stderr); return 1; }
        return 0;
    }
    if (argc < 2 || (strcmp(argv[1], "extract") && strcmp(argv[1], "time"))) {
        fprintf(stderr, "usage: oracle_cli {extract|time} [-a N -m N -M N -o FILE -r REGION -t TAG -s XS|RF|FR|intron-motif] in.bam [ref.fa]\n");
        return 1;
    }
    int timing = !strcmp(argv[1], "time");
    orc_params p; orc_default_params(&p);
    const char *outfile = NULL, *bcfile = NULL;
    int c;
    optind = 2;
    while ((c = getopt(argc, argv, "a:m:M:o:r:t:s:b:")) != -1) {
        switch (c) {
            case 'a': p.min_anchor = (uint32_t)atoi(optarg); break;
            case 'm': p.min_intron = (uint32_t)atoi(optarg); break;
            case 'M': p.max_intron = (uint32_t)atoi(optarg); break;
            case 'o': outfile = optarg; break;
            case 'b': bcfile = optarg; p.barcodes = 1; break;
            case 'r': p.region = optarg; break;
            case 't': p.strand_tag[0] = optarg[0]; p.strand_tag[1] = optarg[0] ? optarg[1] : 0; break;
            case 's':
                if (!strcmp(optarg, "XS")) p.strandness = 0; else if (!strcmp(optarg, "RF")) p.strandness = 1;
                else if (!strcmp(optarg, "FR")) p.strandness = 2; else if (!strcmp(optarg, "intron-motif")) p.strandness = 3;
                else { fprintf(stderr, "Unrecognized strandness argument!\n\n"); return 1; }
                break;
            default: fprintf(stderr, "Error parsing inputs!(1)\n\n"); return 1;
        }
    }
    if (argc - optind >= 1) p.bam = argv[optind++];
    if (argc - optind >= 1) p.fasta = argv[optind++];
    if (optind < argc || !p.bam) { fprintf(stderr, "Error parsing inputs!(2)\n\n"); return 1; }
    if (p.strandness == -1) { fprintf(stderr, "Please supply strandness mode with '-s' option!\n\n"); return 1; }
    if (p.strandness == 3 && !p.fasta) { fprintf(stderr, "Strandness mode 'intron-motif' requires a fasta file!\n\n"); return 1; }

    char err[256]; orc_table *t = NULL;
    double t0 = now();
    if (orc_extract(&p, &t, err, sizeof err)) { if (!strcmp(err, "abort()\n")) abort(); fputs(err, stderr); return 1; }
    double t1 = now();
    if (timing) {
        printf("{\"records\": %llu, \"events\": %llu, \"junctions\": %zu, \"inflated_bytes\": %llu, \"seconds\": %.6f}\n",
               (unsigned long long)t->n_records_total, (unsigned long long)t->n_events, t->n,
               (unsigned long long)t->inflated_bytes, t1 - t0);
    } else {
        FILE *out = outfile ? fopen(outfile, "w") : stdout;
        if (!out) { perror("open output"); return 1; }
        orc_print_bed12(t, out, 1);
        if (outfile) fclose(out);
        if (bcfile) {
            FILE *b = fopen(bcfile, "w");
            if (b) { orc_print_barcodes(t, b, 1); fclose(b); }      /* an unopenable -b file is silently skipped (cc:255-256, :272) */
        }
    }
    orc_table_free(t);
    return 0;
}
