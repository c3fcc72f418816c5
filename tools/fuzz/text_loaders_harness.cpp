// ASan/UBSan harness for the host text loaders of regtools_amd (GTF, VCF incl. gz, BED12, FASTA + .fai): argv[1] = kind, argv[2] = file
#include "cse_host.h"
#include "host_io.h"
#include <cstdio>
#include <cstring>
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::string k = argv[1], path = argv[2], msg;
    if (k == "gtf") { rgx::GtfModel g; msg = g.load(path); printf("gtf %zu tx %zu exons: %s\n", g.tx_id.size(), g.es.size(), msg.c_str()); }
    else if (k == "vcf") { rgx::VcfText v; msg = v.load(path); printf("vcf %zu recs: %s\n", v.recs.size(), msg.c_str()); }
    else if (k == "bed") { rgx::BedJunctions b; msg = b.load(path); printf("bed %zu rows: %s\n", b.n(), msg.c_str()); }
    else if (k == "fa") { rgx::Fasta f; bool ok = f.load(path); std::string s; if (ok) for (auto &q : f.seqs) { f.fetch(q.name, 1, 50, s); f.fetch(q.name, q.len - 3, q.len + 10, s); } printf("fa %d %zu seqs\n", (int)ok, f.seqs.size()); }
    return 0;
}
