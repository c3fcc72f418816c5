// synth_bam.h -- deterministic synthetic BAM + BAI writer (tooling, NOT the hot path).
//
// Produces the input shapes of SURVEY.md section 8(d): coordinate-sorted BAM, BGZF members cut at
// 0xff00 inflated bytes, zlib level configurable, plus a spec-conformant .bai (the reference refuses
// to run without one: /root/reference/src/junctions/junctions_extractor.cc:508-512).
// Lives in its own shared object (libregtools_synth.so) because it needs zlib's *deflate*; the product
// library (libregtools_amd.so) never links zlib.
#ifndef REGTOOLS_SYNTH_BAM_H
#define REGTOOLS_SYNTH_BAM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RGX_SHAPE_SHORT = 0, // config 2/3: 101 bp, 15 % `aMbNcM`, introns from a Zipf(1.0)-weighted table
    RGX_SHAPE_LONG  = 1, // config 5: l_qseq 1000-10000, n_cigar <= 64, 5-20 N ops, M/I/D/S/=/X mix; reads span 6-21 exons of a locus table
    RGX_SHAPE_FUZZ  = 2, // tests: tiny contigs named 1,10,2,MT, every op code, odd tags/flags, unmapped tail
};

typedef struct {
    int      shape;
    uint64_t n_reads;
    uint64_t seed;
    int      level;        // zlib level (6 = the named shape)
    int      threads;      // 0 = hardware concurrency
    uint32_t n_introns;    // size of the intron table (0 = 300000, clamped to the number of spliced reads)
    double   spliced_frac; // 0 = 0.15
    int      realistic_payload; // 0 = seq 0x11 / qual 0xff (the named shape); 1 = random bases + binned quals
    int      slice_index;  // this file holds the slice_index-th of n_slices equal coordinate windows of the genome
    int      n_slices;     // 0/1 = whole genome (multi-GPU weak scaling: rank r generates slice r of the big BAM)
    uint32_t n_genes;      // 0 = introns at independent random loci; >0 = SHORT-shape introns come from a gene model of this many
                           //     genes (the one rgx_synth_annotation writes for the same seed), 10 % with a novel donor
} rgx_synth_params;

typedef struct {
    uint8_t *bam;  size_t bam_len;
    uint8_t *bai;  size_t bai_len;
    uint64_t n_reads, n_spliced, n_blocks, inflated_bytes, cigar_ops;
} rgx_synth_result;

// Generate into memory (malloc'd; free with rgx_synth_free). Returns 0 on success.
int  rgx_synth_generate(const rgx_synth_params *p, rgx_synth_result *out);
void rgx_synth_free(rgx_synth_result *r);

// Generate straight to <path> and <path>.bai.
int  rgx_synth_write(const rgx_synth_params *p, const char *path, rgx_synth_result *stats);

// config 4 companions of a SHORT-shape BAM generated with the same (seed, n_genes): a GTF of n_genes genes x 4 transcripts
// (4-12 exons per gene, alternative transcripts skip internal exons), a sorted VCF of n_variants SNVs (5 % within 3 bp of an
// exon edge, the rest uniform) and, when fasta_path is not NULL, the genome the REF alleles come from (+ .fai).
int  rgx_synth_annotation(const rgx_synth_params *p, uint32_t n_variants, const char *gtf_path, const char *vcf_path, const char *fasta_path);

// Build <path>.bai for an existing coordinate-sorted BAM (used for hand-made test BAMs).
int  rgx_synth_index(const char *bam_path);

#ifdef __cplusplus
}
#endif
#endif
