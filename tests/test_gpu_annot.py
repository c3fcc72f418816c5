"""GPU parity of SURVEY 8(f) rows f2/f3 through the C ABI: `junctions annotate`, `variants annotate`, `cis-splice-effects associate`
against the reference's own goldens and 39 outputs of the real reference on synthetic quartets.  Byte-exact on every output file."""
import os
import subprocess

import pytest

import annot_common as ac

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def work(tmp_path_factory):
    return tmp_path_factory.mktemp("annot_gpu")


def run_mirror(obj, argv, method):
    import regtools_amd
    try:
        obj.parse_options(argv)
        getattr(obj, method)()
    except regtools_amd.RegtoolsError as e:
        return (0 if e.code == 0 else 1), str(e)
    return 0, ""


def test_reference_junctions_annotate_golden(gpu_ctx, work):
    import regtools_amd
    out = os.path.join(str(work), "ja.out")
    ja = regtools_amd.JunctionsAnnotator(ctx=gpu_ctx)
    rc, msg = run_mirror(ja, ["-o", out, os.path.join(ac.REF, "test_hcc1395_junctions.bed"), os.path.join(ac.CSE_REF, "test_chr22.fa"),
                              os.path.join(ac.REF, "test_ensemble_chr22.gtf")], "annotate")
    assert rc == 0, msg
    assert ac.read(out) == ac.read(os.path.join(ac.REF, "expected-annotate.out"))
    assert ja.n_rows == 41


@pytest.mark.parametrize("args,vcf,name", ac.VA_REF, ids=[x[2] for x in ac.VA_REF])
def test_reference_variants_annotate_goldens(gpu_ctx, args, vcf, name, work):
    import regtools_amd
    out = os.path.join(str(work), "va_%s.out" % name)
    rc, msg = run_mirror(regtools_amd.VariantsAnnotator(ctx=gpu_ctx), args + ["-o", out, ac.ref_vcf(vcf), os.path.join(ac.CSE_REF, "test_ensemble_chr22.2.gtf")], "annotate_vcf")
    assert rc == 0, msg
    assert ac.read(out) == ac.read(os.path.join(ac.REF, "expected-annotate-%s.out" % name))


def test_reference_associate_golden(gpu_ctx, work):
    import regtools_amd
    pre = os.path.join(str(work), "as_ref")
    ca = regtools_amd.CisSpliceEffectsAssociator(ctx=gpu_ctx)
    rc, msg = run_mirror(ca, ["-o", pre + ".tsv", "-v", pre + ".vcf", "-j", pre + ".bed", os.path.join(ac.CSE_REF, "test1.vcf"), os.path.join(ac.REF, "junctions_extract.bed"),
                              os.path.join(ac.CSE_REF, "test_chr22.fa"), os.path.join(ac.CSE_REF, "test_ensemble_chr22.2.gtf")], "associate")
    assert rc == 0, msg
    for ext, gold in (("tsv", "annotatedjunctions"), ("vcf", "annotatedvariants"), ("bed", "junctions")):
        assert ac.read(pre + "." + ext) == ac.read(os.path.join(ac.CSE_REF, "expected-cis-splice-effects-identify-default-%s.out" % gold)), ext
    assert ca.stats["n_junctions"] == 1


@pytest.mark.parametrize("case", ac.MANIFEST, ids=[c["name"] for c in ac.MANIFEST])
def test_equals_reference_outputs(gpu_ctx, case, work, oracle_cli):
    import regtools_amd
    q = ac.quartet(case["seed"], case["n_genes"], work, oracle_cli)
    pre = os.path.join(str(work), case["name"])
    if case["cmd"] == "junctions-annotate":
        rc, msg = run_mirror(regtools_amd.JunctionsAnnotator(ctx=gpu_ctx), case["args"] + ["-o", pre + ".tsv", q["bed"], q["fasta"], q["gtf"]], "annotate")      # ([] or ["-S"])
        exts = ["tsv"]
    elif case["cmd"] == "variants-annotate":
        rc, msg = run_mirror(regtools_amd.VariantsAnnotator(ctx=gpu_ctx), case["args"] + ["-o", pre + ".vcf", q["vcf"], q["gtf"]], "annotate_vcf")
        exts = ["vcf"]
    else:
        rc, msg = run_mirror(regtools_amd.CisSpliceEffectsAssociator(ctx=gpu_ctx),
                             case["args"] + ["-o", pre + ".tsv", "-v", pre + ".vcf", "-j", pre + ".bed", q["vcf"], q["bed"], q["fasta"], q["gtf"]], "associate")
        exts = ["tsv", "vcf", "bed"]
    assert rc == case["rc"], msg
    for ext in exts:
        assert ac.read(pre + "." + ext) == ac.read(os.path.join(ac.ANNOT, "%s.%s" % (case["name"], ext))), ext


def test_bed_reader_quirks_and_cli(gpu_ctx, work):
    """Same quirk table as tests/test_oracle_annot.py (verified there against the reference), through the product's CLI binary."""
    rows = ac.read(os.path.join(ac.REF, "test_hcc1395_junctions.bed")).decode().splitlines()
    fa, gtf = os.path.join(ac.CSE_REF, "test_chr22.fa"), os.path.join(ac.REF, "test_ensemble_chr22.gtf")
    exp = ac.read(os.path.join(ac.REF, "expected-annotate.out")).decode().splitlines()
    exe = os.path.join(ROOT, "bin", "regtools-amd")

    def annotate(lines):
        p = os.path.join(str(work), "q.bed")
        open(p, "w").write("\n".join(lines) + "\n")
        out = os.path.join(str(work), "q.out")
        r = subprocess.run([exe, "junctions", "annotate", "-o", out, p, fa, gtf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return r.returncode, ac.read(out).decode().splitlines()

    assert annotate(["track name=x", "#comment"] + rows) == (0, exp)
    assert annotate(rows[:3] + ["#late header"] + rows[3:]) == (0, exp[:4])
    assert annotate(rows[:5] + [""] + rows[5:]) == (0, exp[:6])
    rc, got = annotate(rows[:2] + ["\t".join(rows[2].split("\t")[:6])] + rows[3:])
    assert rc == 1 and got == exp[:3]
    # the other two commands through the binary
    pre = os.path.join(str(work), "cli")
    r = subprocess.run([exe, "variants", "annotate", "-o", pre + ".vcf", os.path.join(ac.CSE_REF, "test1.vcf"), os.path.join(ac.CSE_REF, "test_ensemble_chr22.2.gtf")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and ac.read(pre + ".vcf") == ac.read(os.path.join(ac.REF, "expected-annotate-default.out"))
    r = subprocess.run([exe, "cis-splice-effects", "associate", "-o", pre + ".tsv", os.path.join(ac.CSE_REF, "test1.vcf"), os.path.join(ac.REF, "junctions_extract.bed"),
                        os.path.join(ac.CSE_REF, "test_chr22.fa"), os.path.join(ac.CSE_REF, "test_ensemble_chr22.2.gtf")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and ac.read(pre + ".tsv") == ac.read(os.path.join(ac.CSE_REF, "expected-cis-splice-effects-identify-default-annotatedjunctions.out"))
    for sub in (["junctions", "annotate", "-h"], ["variants", "annotate", "-h"], ["cis-splice-effects", "associate", "-h"]):
        assert subprocess.run([exe] + sub, stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 0


def test_malformed_gtf_exit_codes(gpu_ctx, work):
    import regtools_amd
    bed, fa = os.path.join(ac.REF, "test_hcc1395_junctions.bed"), os.path.join(ac.CSE_REF, "test_chr22.fa")
    for path, rc in ac.malformed_gtfs(work):
        out = os.path.join(str(work), "mg.out")
        got, msg = run_mirror(regtools_amd.JunctionsAnnotator(ctx=gpu_ctx), ["-o", out, bed, fa, path], "annotate")
        assert got == rc, (path, msg)
        if rc == 0:
            assert ac.read(out) == ac.read(os.path.join(ac.REF, "expected-annotate.out")), path


# -- the annotated VCF through htslib's typed round trip (regtools_amd/csrc/vcf_rewrite.cpp): "%g" floats, FORMAT fill-in, header
#    de-duplication, undeclared tags, gzip and BCF input -- against the real reference's output for the same input and GTF
import json  # noqa: E402

import vcf_cases  # noqa: E402

VCF_GOLD = os.path.join(ROOT, "tests", "golden", "vcf_writer")
VCF_NAMES = sorted(json.load(open(os.path.join(VCF_GOLD, "manifest.json"))))


@pytest.fixture(scope="module")
def vcf_inputs(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("vcf_in"))
    gtf = os.path.join(d, "near.gtf")
    open(gtf, "w").write(vcf_cases.GTF_NEAR)
    return vcf_cases.build(d), gtf, d


@pytest.mark.parametrize("name", VCF_NAMES)
def test_annotated_vcf_equals_the_reference(gpu_ctx, vcf_inputs, name):
    import regtools_amd
    inputs, gtf, d = vcf_inputs
    src, out = os.path.join(d, name + ".vcf"), os.path.join(d, name + ".out.vcf")
    open(src, "wb").write(inputs[name])
    for suffix, blob in vcf_cases.companions().get(name, {}).items():
        open(os.path.join(d, name + suffix), "wb").write(blob)
    rc, msg = run_mirror(regtools_amd.VariantsAnnotator(ctx=gpu_ctx), ["-o", out, src, gtf], "annotate_vcf")
    assert rc == 0, msg
    assert ac.read(out) == ac.read(os.path.join(VCF_GOLD, name + ".near.vcf"))


def test_a_genome_rewritten_between_two_calls_is_read_again(gpu_ctx, work):
    """The context keeps the last FASTA mapped from call to call (api_ctx.cpp host_fasta): a file that changed under the same name is a new file."""
    import regtools_amd
    src = ac.read(os.path.join(ac.CSE_REF, "test_chr22.fa")).decode()
    fa = os.path.join(str(work), "again.fa")
    bed, gtf = os.path.join(ac.REF, "test_hcc1395_junctions.bed"), os.path.join(ac.REF, "test_ensemble_chr22.gtf")
    out = os.path.join(str(work), "again.out")

    def sites(text):
        open(fa, "w").write(text)
        if os.path.exists(fa + ".fai"):
            os.remove(fa + ".fai")
        ja = regtools_amd.JunctionsAnnotator(ctx=gpu_ctx)
        rc, msg = run_mirror(ja, ["-o", out, bed, fa, gtf], "annotate")
        assert rc == 0, msg
        return [l.split("\t")[6] for l in ac.read(out).decode().splitlines()[1:]]

    first = sites(src)
    assert first == [l.split("\t")[6] for l in ac.read(os.path.join(ac.REF, "expected-annotate.out")).decode().splitlines()[1:]]
    head, body = src.split("\n", 1)
    swapped = head + "\n" + body.translate(str.maketrans("ACGTacgt", "CATGcatg"))          # same size, same name, other bases
    second = sites(swapped)
    assert second == [s.translate(str.maketrans("ACGT", "CATG")) for s in first] and second != first
    assert sites(src) == first
