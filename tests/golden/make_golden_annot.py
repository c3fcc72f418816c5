#!/usr/bin/env python
"""Golden outputs of the REAL reference (oracle/_ref) for `junctions annotate`, `variants annotate` and `cis-splice-effects associate`
on the deterministic synthetic quartets of tests/cse_synth.py (the BED12 input is the reference's own `junctions extract -s XS` of the
quartet's BAM; tests regenerate it with the pinned oracle).  Also copies the data files the reference's integration tests hold for
these commands into tests/golden/annot_ref/.  Dev container only."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cse_synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
DATA = "/root/reference/tests/integration-test/data"
VA_ARGS = [[], ["-E"], ["-I"], ["-e", "6", "-i", "6"], ["-S"], ["-E", "-I"]]
AS_ARGS = [[], ["-E"], ["-I", "-e", "6"], ["-S"], ["-w", "100"], ["-i", "10", "-e", "10"]]


def main():
    ref_dir = os.path.join(HERE, "annot_ref")
    os.makedirs(ref_dir, exist_ok=True)
    for f in ["bed/test_hcc1395_junctions.bed", "gtf/test_ensemble_chr22.gtf", "cis-splice-effects-associate/junctions_extract.bed", "vcf/test2.vcf",
              "junctions-annotate/expected-annotate.out"] + ["variants-annotate/" + x for x in sorted(os.listdir(os.path.join(DATA, "variants-annotate")))]:
        dst = os.path.join(ref_dir, os.path.basename(f))
        shutil.copyfile(os.path.join(DATA, f), dst)
    out = os.path.join(HERE, "annot")
    os.makedirs(out, exist_ok=True)
    cases = []
    with tempfile.TemporaryDirectory() as td:
        for seed in (1, 2, 3):
            q = cse_synth.build(os.path.join(td, "s%d" % seed), seed=seed, n_genes=10 + 2 * seed)
            bed = os.path.join(td, "s%d.bed" % seed)
            subprocess.run([REF, "junctions", "extract", "-s", "XS", "-o", bed, q["bam"]], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            name = "ja_s%d" % seed
            r = subprocess.run([REF, "junctions", "annotate", "-o", os.path.join(out, name + ".tsv"), bed, q["fasta"], q["gtf"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            cases.append(dict(name=name, cmd="junctions-annotate", seed=seed, n_genes=10 + 2 * seed, args=[], rc=r.returncode))
            # -S (junctions_annotator.cc:392-393): single-exon transcripts take part (the synthetic annotations have some, made of an exon that
            # ends where annotated junctions start)
            name = "ja_s%d_S" % seed
            r = subprocess.run([REF, "junctions", "annotate", "-S", "-o", os.path.join(out, name + ".tsv"), bed, q["fasta"], q["gtf"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            cases.append(dict(name=name, cmd="junctions-annotate", seed=seed, n_genes=10 + 2 * seed, args=["-S"], rc=r.returncode))
            for k, a in enumerate(VA_ARGS):
                name = "va_s%d_%d" % (seed, k)
                r = subprocess.run([REF, "variants", "annotate"] + a + ["-o", os.path.join(out, name + ".vcf"), q["vcf"], q["gtf"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                cases.append(dict(name=name, cmd="variants-annotate", seed=seed, n_genes=10 + 2 * seed, args=a, rc=r.returncode))
            for k, a in enumerate(AS_ARGS):
                name = "as_s%d_%d" % (seed, k)
                files = {x: os.path.join(out, "%s.%s" % (name, x)) for x in ("tsv", "vcf", "bed")}
                r = subprocess.run([REF, "cis-splice-effects", "associate"] + a + ["-o", files["tsv"], "-v", files["vcf"], "-j", files["bed"], q["vcf"], bed, q["fasta"], q["gtf"]],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                rows = open(files["tsv"]).read().count("\n") - 1 if os.path.exists(files["tsv"]) else -1
                cases.append(dict(name=name, cmd="associate", seed=seed, n_genes=10 + 2 * seed, args=a, rc=r.returncode, rows=rows))
                print(name, a, "rc", r.returncode, "rows", rows)
    json.dump(cases, open(os.path.join(out, "manifest.json"), "w"), indent=1)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
