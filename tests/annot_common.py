"""Shared by test_oracle_annot.py / test_gpu_annot.py: the fixtures of SURVEY 8(f) rows f2/f3 (`junctions annotate`,
`variants annotate`, `cis-splice-effects associate`) -- the data files the reference's integration tests hold (tests/golden/annot_ref)
and the reference's outputs on the synthetic quartets (tests/golden/annot, made by make_golden_annot.py)."""
import json
import os
import subprocess

import cases
import cse_synth

ANNOT = os.path.join(cases.GOLD, "annot")
REF = os.path.join(cases.GOLD, "annot_ref")
CSE_REF = os.path.join(cases.GOLD, "cse_ref")
MANIFEST = json.load(open(os.path.join(ANNOT, "manifest.json")))
# the reference's own goldens: test_variants_annotate.py:39-118
VA_REF = [([], "test1.vcf", "default"), (["-e", "6", "-i", "6", "-S"], "test1.vcf", "e6-i6-S"), (["-E"], "test2.vcf", "E"), (["-I"], "test2.vcf", "I"),
          (["-E", "-i", "6"], "test2.vcf", "E-i6"), (["-e", "6", "-I"], "test2.vcf", "e6-I")]
_quartets = {}


def ref_vcf(name):
    return os.path.join(CSE_REF if name == "test1.vcf" else REF, name)


def quartet(seed, n_genes, tmp, oracle_cli):
    """quartet + the BED12 of `junctions extract -s XS` on its BAM (made with the pinned oracle)"""
    if seed not in _quartets:
        q = cse_synth.build(os.path.join(str(tmp), "s%d" % seed), seed=seed, n_genes=n_genes)
        q["bed"] = os.path.join(str(tmp), "s%d.bed" % seed)
        subprocess.run([oracle_cli, "extract", "-s", "XS", "-o", q["bed"], q["bam"]], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        _quartets[seed] = q
    return _quartets[seed]


def read(path):
    return open(path, "rb").read()


def malformed_gtfs(tmp):
    """(path, exit status of the reference, header-only?) -- verified against the real reference (oracle/_ref) in the dev container.  An empty line
    makes the reference die with an uncaught std::out_of_range (status 134); here that is an ordinary error (status 1)."""
    g = read(os.path.join(REF, "test_ensemble_chr22.gtf")).decode().splitlines()
    first_exon = next(i for i, l in enumerate(g) if "\texon\t" in l)
    out = []

    def put(name, lines, rc):
        p = os.path.join(str(tmp), name)
        open(p, "w").write("\n".join(lines) + "\n")
        out.append((p, rc))
    put("empty_line.gtf", g[:20] + [""] + g[20:], 1)
    put("eight_fields.gtf", g[:20] + ["\t".join(g[20].split("\t")[:8])] + g[21:], 1)
    put("comment_inside.gtf", g[:20] + ["#comment"] + g[20:], 0)
    f = g[first_exon].split("\t"); f[6] = "."
    put("no_strand_first_exon.gtf", g[:first_exon] + ["\t".join(f)] + g[first_exon + 1:], 1)
    f = g[first_exon + 1].split("\t"); f[6] = "."
    put("no_strand_later_exon.gtf", g[:first_exon + 1] + ["\t".join(f)] + g[first_exon + 2:], 0)
    return out
