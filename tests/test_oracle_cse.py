"""Pins the oracle's `cis-splice-effects identify` restatement (oracle/oracle_cse.c): the reference's own 2 x 3 goldens and
42 outputs of the real reference on synthetic GTF/VCF/FASTA/BAM quartets (tests/golden/cse, made by make_golden_cse.py)."""
import json
import os
import subprocess

import pytest

import cases
import cse_synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSE = os.path.join(cases.GOLD, "cse")
MANIFEST = json.load(open(os.path.join(CSE, "manifest.json")))
_quartets = {}


def quartet(seed, n_genes, tmp):
    if seed not in _quartets:
        _quartets[seed] = cse_synth.build(os.path.join(str(tmp), "s%d" % seed), seed=seed, n_genes=n_genes)
    return _quartets[seed]


@pytest.fixture(scope="module")
def work(tmp_path_factory):
    return tmp_path_factory.mktemp("cse")


def run_identify(exe_args, q, out_prefix):
    files = {x: "%s.%s" % (out_prefix, x) for x in ("tsv", "vcf", "bed")}
    r = subprocess.run(exe_args + ["-o", files["tsv"], "-v", files["vcf"], "-j", files["bed"], q["vcf"], q["bam"], q["fasta"], q["gtf"]],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, files


REF_GOLD = os.path.join(cases.GOLD, "cse_ref")


@pytest.mark.parametrize("strand,name", [("XS", "default"), ("RF", "default-stranded")])
def test_reference_identify_goldens(strand, name, work, oracle_cli):
    q = dict(vcf=os.path.join(REF_GOLD, "test1.vcf"), bam=os.path.join(REF_GOLD, "test_hcc1395.2.bam"),
             fasta=os.path.join(REF_GOLD, "test_chr22.fa"), gtf=os.path.join(REF_GOLD, "test_ensemble_chr22.2.gtf"))
    rc, files = run_identify([oracle_cli, "identify", "-s", strand], q, os.path.join(str(work), "ref_" + strand))
    assert rc == 0
    for ext, gold in (("tsv", "annotatedjunctions"), ("vcf", "annotatedvariants"), ("bed", "junctions")):
        exp = open(os.path.join(REF_GOLD, "expected-cis-splice-effects-identify-%s-%s.out" % (name, gold)), "rb").read()
        assert open(files[ext], "rb").read() == exp, ext


@pytest.mark.parametrize("case", MANIFEST, ids=[c["name"] for c in MANIFEST])
def test_oracle_equals_reference(case, work, oracle_cli):
    q = quartet(case["seed"], case["n_genes"], work)
    rc, files = run_identify([oracle_cli, "identify"] + case["args"], q, os.path.join(str(work), case["name"]))
    assert rc == case["rc"]
    for ext in ("tsv", "vcf", "bed"):
        assert open(files[ext], "rb").read() == open(os.path.join(CSE, "%s.%s" % (case["name"], ext)), "rb").read(), ext
