"""`cis-splice-effects identify -s XS` and a spliced read whose strand tag lies behind an aux field of unknown type (tests/odd_aux_cases.py): upstream extracts
every splice-relevant variant's window on its own, and the first window that READS such a read ends the process inside bam_aux_get (abort(), sam.c:1233-1252)
behind that variant's echo; a read no window reads ends nothing.  The product extracts once: the decode kernels mark such reads, the host asks which window
reads one (cse_api.cpp).  Status, streams and the files of the runs that complete are the real reference's (tests/golden/cli/cli_odd_aux_streams.json)."""
import json
import os
import subprocess

import pytest

import odd_aux_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "regtools-amd")
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
GOLD = os.path.join(ROOT, "tests", "golden", "cli", "cli_odd_aux_streams.json")


def _run(exe, argv, td):
    r = subprocess.run([exe] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    fix = lambda b: b.replace(ROOT.encode(), b"@ROOT@").replace(td.encode(), b"@TMP@").decode("latin-1")
    return r.returncode, fix(r.stdout), fix(r.stderr)


def _check(exe, tmp_path, need_all):
    gold = json.load(open(GOLD))
    td = str(tmp_path)
    cs, digest = odd_aux_cases.cases(td)
    assert digest == gold["inputs_sha256"], "the generated inputs are not the ones the golden was made from"
    seen = set()
    for argv, files in cs:
        want = gold[" ".join(os.path.basename(a) for a in argv)]
        rc, out, err = _run(exe, argv, td)
        assert (rc, out) == (want["rc"], want["stdout"]), (argv, rc)
        assert err == want["stderr"], (argv, err[-500:])
        got_files = {os.path.basename(f): open(f, "rb").read().decode("latin-1") for f in files if os.path.exists(f)}
        assert got_files == want["files"], argv
        seen.add(rc)
    if need_all:
        assert seen == {0, -6}


def test_golden_is_what_the_reference_does(built, tmp_path):
    if not os.path.exists(REF):
        pytest.skip("the reference binary is only built in the dev container")
    _check(REF, tmp_path, True)


@pytest.mark.gpu
def test_product_ends_where_the_reference_does(built, tmp_path):
    _check(EXE, tmp_path, True)


@pytest.mark.gpu
def test_library_reports_the_abort_over_shards(built, gpu_ctx, tmp_path):
    """the same through the C ABI: RGX_ERR_ABORT, also when the extraction is sharded over a device list (every shard's marked reads reach the window test)"""
    import regtools_amd
    from regtools_amd import cse
    td = str(tmp_path)
    q, bams, _ = odd_aux_cases.build(td)
    for name, dies in (("mid", True), ("clear", False)):
        for devices in (None, [0, 0, 0]):
            ci = cse.CisSpliceEffectsIdentifier(ctx=gpu_ctx) if devices is None else cse.CisSpliceEffectsIdentifier(devices=devices)
            ci.parse_options(["-s", "XS", "-o", os.path.join(td, "o.tsv"), "-j", os.path.join(td, "o.bed"), q["vcf"], bams[name], q["fasta"], q["gtf"]])
            try:
                ci.identify()
                died = False
            except regtools_amd.RegtoolsError as e:
                assert e.code == 9, e                       # RGX_ERR_ABORT
                died = True
            assert died == dies, (name, devices)
