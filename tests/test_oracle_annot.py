"""Pins the oracle's restatements of `junctions annotate`, `variants annotate` and `cis-splice-effects associate` (oracle/oracle_cse.c):
the reference's own goldens for the three commands and 39 outputs of the real reference on synthetic quartets."""
import os
import subprocess

import pytest

import annot_common as ac


@pytest.fixture(scope="module")
def work(tmp_path_factory):
    return tmp_path_factory.mktemp("annot")


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode


def test_reference_junctions_annotate_golden(work, oracle_cli):
    out = os.path.join(str(work), "ja.out")
    assert run([oracle_cli, "junctions-annotate", "-o", out, os.path.join(ac.REF, "test_hcc1395_junctions.bed"), os.path.join(ac.CSE_REF, "test_chr22.fa"),
                os.path.join(ac.REF, "test_ensemble_chr22.gtf")]) == 0
    assert ac.read(out) == ac.read(os.path.join(ac.REF, "expected-annotate.out"))


@pytest.mark.parametrize("args,vcf,name", ac.VA_REF, ids=[x[2] for x in ac.VA_REF])
def test_reference_variants_annotate_goldens(args, vcf, name, work, oracle_cli):
    out = os.path.join(str(work), "va_%s.out" % name)
    assert run([oracle_cli, "variants-annotate"] + args + ["-o", out, ac.ref_vcf(vcf), os.path.join(ac.CSE_REF, "test_ensemble_chr22.2.gtf")]) == 0
    assert ac.read(out) == ac.read(os.path.join(ac.REF, "expected-annotate-%s.out" % name))


def test_reference_associate_golden(work, oracle_cli):
    """test_cis_splice_effects_associate.py:35-54: output identical to the identify -s XS goldens"""
    pre = os.path.join(str(work), "as_ref")
    assert run([oracle_cli, "associate", "-o", pre + ".tsv", "-v", pre + ".vcf", "-j", pre + ".bed", os.path.join(ac.CSE_REF, "test1.vcf"),
                os.path.join(ac.REF, "junctions_extract.bed"), os.path.join(ac.CSE_REF, "test_chr22.fa"), os.path.join(ac.CSE_REF, "test_ensemble_chr22.2.gtf")]) == 0
    for ext, gold in (("tsv", "annotatedjunctions"), ("vcf", "annotatedvariants"), ("bed", "junctions")):
        assert ac.read(pre + "." + ext) == ac.read(os.path.join(ac.CSE_REF, "expected-cis-splice-effects-identify-default-%s.out" % gold)), ext


@pytest.mark.parametrize("case", ac.MANIFEST, ids=[c["name"] for c in ac.MANIFEST])
def test_oracle_equals_reference(case, work, oracle_cli):
    q = ac.quartet(case["seed"], case["n_genes"], work, oracle_cli)
    pre = os.path.join(str(work), case["name"])
    if case["cmd"] == "junctions-annotate":
        assert run([oracle_cli, "junctions-annotate"] + case["args"] + ["-o", pre + ".tsv", q["bed"], q["fasta"], q["gtf"]]) == case["rc"]      # ([] or ["-S"])
        exts = ["tsv"]
    elif case["cmd"] == "variants-annotate":
        assert run([oracle_cli, "variants-annotate"] + case["args"] + ["-o", pre + ".vcf", q["vcf"], q["gtf"]]) == case["rc"]
        exts = ["vcf"]
    else:
        assert run([oracle_cli, "associate"] + case["args"] + ["-o", pre + ".tsv", "-v", pre + ".vcf", "-j", pre + ".bed", q["vcf"], q["bed"], q["fasta"], q["gtf"]]) == case["rc"]
        exts = ["tsv", "vcf", "bed"]
    for ext in exts:
        assert ac.read(pre + "." + ext) == ac.read(os.path.join(ac.ANNOT, "%s.%s" % (case["name"], ext))), ext


def test_bed_reader_quirks(work, oracle_cli):
    """bedtools' reader as the reference drives it: leading header lines are skipped, a later header or blank line silently ends the input,
    a row that is not BED12 ends the run with status 1 after the rows before it were written."""
    rows = ac.read(os.path.join(ac.REF, "test_hcc1395_junctions.bed")).decode().splitlines()
    fa, gtf = os.path.join(ac.CSE_REF, "test_chr22.fa"), os.path.join(ac.REF, "test_ensemble_chr22.gtf")
    exp = ac.read(os.path.join(ac.REF, "expected-annotate.out")).decode().splitlines()

    def annotate(lines):
        p = os.path.join(str(work), "q.bed")
        open(p, "w").write("\n".join(lines) + "\n")
        out = os.path.join(str(work), "q.out")
        rc = run([oracle_cli, "junctions-annotate", "-o", out, p, fa, gtf])
        return rc, ac.read(out).decode().splitlines()

    assert annotate(["track name=x", "#comment"] + rows) == (0, exp)
    assert annotate(rows[:3] + ["#late header"] + rows[3:]) == (0, exp[:4])
    assert annotate(rows[:5] + [""] + rows[5:]) == (0, exp[:6])
    rc, got = annotate(rows[:2] + ["\t".join(rows[2].split("\t")[:6])] + rows[3:])
    assert rc == 1 and got == exp[:3]


def test_malformed_gtf_exit_codes(work, oracle_cli):
    bed, fa = os.path.join(ac.REF, "test_hcc1395_junctions.bed"), os.path.join(ac.CSE_REF, "test_chr22.fa")
    for path, rc in ac.malformed_gtfs(work):
        out = os.path.join(str(work), "mg.out")
        assert run([oracle_cli, "junctions-annotate", "-o", out, bed, fa, path]) == rc, path
        if rc == 0:
            assert ac.read(out) == ac.read(os.path.join(ac.REF, "expected-annotate.out")), path
