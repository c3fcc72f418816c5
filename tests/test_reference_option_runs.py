"""`cis-splice-effects identify` / `associate` on the REFERENCE'S OWN data files under every option combination its integration tests run them with
(tests/integration-test/test_cis_splice_effects_identify.py:100-348, test_cis_splice_effects_associate.py:80-200 -- upstream asserts only the exit status there;
its two golden triplets cover `-s XS` and `-s RF` with nothing else set).  tests/golden/cse_ref_opts/ holds the three output files of the real reference for each
(make_golden_cse_ref_opts.py): the oracle must print them here, the product on the GPU box.  With them `junctions annotate` (upstream: one golden) on both junction
files and both annotations the reference's tests hold, with and without -S.

The same for `junctions extract` on the two BAMs of real aligner output the reference's tests hold, under options its six goldens do not reach (FR, intron-motif with the
chr22 genome -- which ends the run with status 1 on the BAM of contig 1 --, another strand tag, anchor and intron bounds at their edges, regions of every form):
tests/golden/extract_ref_opts/ (make_golden_extract_ref_opts.py), 50 runs of the real reference."""
import json
import os
import subprocess

import pytest

import cases

OPTS = os.path.join(cases.GOLD, "cse_ref_opts")
CSE_REF = os.path.join(cases.GOLD, "cse_ref")
BED = os.path.join(cases.GOLD, "annot_ref", "junctions_extract.bed")
MANIFEST = json.load(open(os.path.join(OPTS, "manifest.json")))
EXTRACT = os.path.join(cases.GOLD, "extract_ref_opts")
EXTRACT_MANIFEST = json.load(open(os.path.join(EXTRACT, "manifest.json")))
QUARTET = [os.path.join(CSE_REF, x) for x in ("test1.vcf", "test_hcc1395.2.bam", "test_chr22.fa", "test_ensemble_chr22.2.gtf")]


def inputs(case):
    if case["cmd"] == "junctions-annotate":                             # (bed, genome, annotation)
        return [os.path.join(cases.GOLD, case["bed"]), QUARTET[2], os.path.join(cases.GOLD, case["gtf"])]
    return QUARTET if case["cmd"] == "identify" else [QUARTET[0], BED, QUARTET[2], QUARTET[3]]


def outputs(pre, case):
    return ["-o", pre + ".tsv"] + ([] if case["cmd"] == "junctions-annotate" else ["-v", pre + ".vcf", "-j", pre + ".bed"])


def same_files(pre, case):
    for ext in ("tsv",) if case["cmd"] == "junctions-annotate" else ("tsv", "vcf", "bed"):
        assert open("%s.%s" % (pre, ext), "rb").read() == open(os.path.join(OPTS, "%s.%s" % (case["name"], ext)), "rb").read(), (case["name"], ext)


def test_every_upstream_combination_is_among_the_cases():
    # test_cis_splice_effects_identify.py: -e 6 -i 6 -S, -E, -I, -E -i 6, -e 6 -I, -a 30, -m 8039 -M 8039, -w 5 (each with -s XS)
    have = {tuple(c["args"]) for c in MANIFEST if c["cmd"] == "identify"}
    for opts in (["-e", "6", "-i", "6", "-S"], ["-E"], ["-I"], ["-E", "-i", "6"], ["-e", "6", "-I"], ["-a", "30"], ["-m", "8039", "-M", "8039"], ["-w", "5"]):
        assert tuple(opts + ["-s", "XS"]) in have, opts
    # the option runs do not all print the same thing (the switches reach the outputs on this data)
    assert len({open(os.path.join(OPTS, c["name"] + ".vcf"), "rb").read() for c in MANIFEST if c["cmd"] != "junctions-annotate"}) >= 5
    assert len({open(os.path.join(OPTS, c["name"] + ".tsv"), "rb").read() for c in MANIFEST}) >= 3


@pytest.mark.parametrize("case", MANIFEST, ids=[c["name"] for c in MANIFEST])
def test_oracle_prints_what_the_reference_prints(case, tmp_path, oracle_cli):
    pre = str(tmp_path / case["name"])
    r = subprocess.run([oracle_cli, case["cmd"]] + case["args"] + outputs(pre, case) + inputs(case), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == case["rc"], r.stderr
    same_files(pre, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", MANIFEST, ids=[c["name"] for c in MANIFEST])
def test_product_prints_what_the_reference_prints(gpu_ctx, case, tmp_path):
    import regtools_amd
    pre = str(tmp_path / case["name"])
    mirror = {"identify": regtools_amd.CisSpliceEffectsIdentifier, "associate": regtools_amd.CisSpliceEffectsAssociator, "junctions-annotate": regtools_amd.JunctionsAnnotator}
    obj = mirror[case["cmd"]](ctx=gpu_ctx)
    obj.parse_options(case["args"] + outputs(pre, case) + inputs(case))
    getattr(obj, "annotate" if case["cmd"] == "junctions-annotate" else case["cmd"])()
    assert case["rc"] == 0
    same_files(pre, case)


def extract_argv(case):
    return case["args"] + [os.path.join(cases.GOLD, case["bam"])] + ([os.path.join(cases.GOLD, case["fasta"])] if case["fasta"] else [])


def test_the_extract_runs_are_not_all_alike():
    assert len({open(os.path.join(EXTRACT, c["name"] + ".out"), "rb").read() for c in EXTRACT_MANIFEST}) >= 15
    assert sorted({c["rc"] for c in EXTRACT_MANIFEST}) == [0, 1]


@pytest.mark.parametrize("case", EXTRACT_MANIFEST, ids=[c["name"] for c in EXTRACT_MANIFEST])
def test_oracle_extracts_what_the_reference_extracts(case):
    from conftest import run_oracle
    rc, out, _ = run_oracle(extract_argv(case))
    assert rc == case["rc"]
    assert out == open(os.path.join(EXTRACT, case["name"] + ".out"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("case", EXTRACT_MANIFEST, ids=[c["name"] for c in EXTRACT_MANIFEST])
def test_product_extracts_what_the_reference_extracts(gpu_ctx, case):
    import regtools_amd
    je = regtools_amd.JunctionsExtractor(ctx=gpu_ctx)
    try:
        je.parse_options(extract_argv(case))
        je.identify_junctions_from_BAM()
        rc, out = 0, je.bed12()
    except regtools_amd.RegtoolsError as e:
        rc, out = (0 if e.code == 0 else 1), b""
    assert rc == case["rc"]
    assert out == open(os.path.join(EXTRACT, case["name"] + ".out"), "rb").read()
