// synth_cli.cpp -- command-line face of the synthetic BAM writer (tooling).
#include "synth_bam.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char **argv) {
    if (argc >= 3 && !strcmp(argv[1], "index")) return rgx_synth_index(argv[2]);
    if (argc >= 6 && !strcmp(argv[1], "annotation")) {   // annotation PREFIX N_GENES N_VARIANTS SEED [--fasta]
        rgx_synth_params p; memset(&p, 0, sizeof p);
        p.n_genes = (uint32_t)atoi(argv[3]); p.seed = strtoull(argv[5], nullptr, 10);
        const std::string pre = argv[2];
        const bool fa = argc >= 7 && !strcmp(argv[6], "--fasta");
        return rgx_synth_annotation(&p, (uint32_t)atoi(argv[4]), (pre + ".gtf").c_str(), (pre + ".vcf").c_str(), fa ? (pre + ".fa").c_str() : nullptr);
    }
    if (argc < 4 || strcmp(argv[1], "write")) {
        fprintf(stderr, "usage: synth_bam write OUT.bam N_READS [--shape short|long|fuzz] [--seed S] [--level L] [--threads T] [--introns K] [--realistic] [--genes G]\n"
                        "       synth_bam annotation PREFIX N_GENES N_VARIANTS SEED [--fasta]\n"
                        "       synth_bam index IN.bam\n");
        return 1;
    }
    rgx_synth_params p; memset(&p, 0, sizeof p);
    p.n_reads = strtoull(argv[3], nullptr, 10); p.seed = 1; p.level = 6;
    for (int i = 4; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--shape" && i + 1 < argc) { std::string v = argv[++i]; p.shape = v == "long" ? RGX_SHAPE_LONG : v == "fuzz" ? RGX_SHAPE_FUZZ : RGX_SHAPE_SHORT; }
        else if (a == "--seed" && i + 1 < argc) p.seed = strtoull(argv[++i], nullptr, 10);
        else if (a == "--level" && i + 1 < argc) p.level = atoi(argv[++i]);
        else if (a == "--threads" && i + 1 < argc) p.threads = atoi(argv[++i]);
        else if (a == "--introns" && i + 1 < argc) p.n_introns = (uint32_t)atoi(argv[++i]);
        else if (a == "--realistic") p.realistic_payload = 1;
        else if (a == "--genes" && i + 1 < argc) p.n_genes = (uint32_t)atoi(argv[++i]);
        else if (a == "--slice" && i + 2 < argc) { p.slice_index = atoi(argv[++i]); p.n_slices = atoi(argv[++i]); }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
    }
    rgx_synth_result st;
    int rc = rgx_synth_write(&p, argv[2], &st);
    if (rc) { fprintf(stderr, "synth_bam: failed (%d)\n", rc); return rc; }
    printf("{\"reads\": %llu, \"spliced\": %llu, \"members\": %llu, \"inflated_bytes\": %llu, \"bam_bytes\": %zu, \"cigar_ops\": %llu}\n",
           (unsigned long long)st.n_reads, (unsigned long long)st.n_spliced, (unsigned long long)st.n_blocks,
           (unsigned long long)st.inflated_bytes, st.bam_len, (unsigned long long)st.cigar_ops);
    return 0;
}
