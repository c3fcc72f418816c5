"""What reading a VCF SAYS and how it ends (CPU; regtools_amd/csrc/vcf_rewrite.cpp + cse_host.cpp in tests/hostemu against the real reference's stderr,
status and output file: tests/golden/vcf_writer/diagnostics.json from tests/golden/make_golden_vcf_diag.py).  The reference reads every record through
htslib's vcf_parse, which warns once per name the header does not declare, ends the read loop at a record whose sample columns do not fit, and ends
the PROCESS itself -- exit(1) or abort(), past regtools' handlers -- on a sample with more fields than FORMAT has keys or a FORMAT key that is a
Flag; reading the header has messages and two aborts of its own.  One process per case: "PL should be declared as Number=G" is said once per process."""
import json
import os
import subprocess
import sys

import pytest

import vcf_diag_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "vcf_writer", "diagnostics.json")))

RUNNER = """
import ctypes, sys
lib = ctypes.CDLL(sys.argv[1])
err = ctypes.create_string_buffer(512)
rc = lib.emu_vcf_rewrite(sys.argv[2].encode(), sys.argv[3].encode(), err, 512)
sys.stdout.write("%d\\n%s" % (rc, err.value.decode("latin1")))
"""


def test_every_case_has_its_expectation():
    assert sorted(GOLD) == sorted(vcf_diag_cases.CASES)
    assert {g["rc"] for g in GOLD.values()} == {0, 1, -6}            # a complete run, exit(1), abort()


@pytest.mark.parametrize("name", sorted(vcf_diag_cases.CASES))
def test_says_and_ends_as_the_reference(built, tmp_path, name):
    gold = GOLD[name]
    src, dst = str(tmp_path / "in.vcf"), str(tmp_path / "out.vcf")
    open(src, "w").write(vcf_diag_cases.CASES[name])
    r = subprocess.run([sys.executable, "-c", RUNNER, os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"), src, dst], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0, r.stderr
    rc, _, msg = r.stdout.decode("latin1").partition("\n")
    rc = int(rc)
    said = [l for l in r.stderr.decode("latin1").split("\n") if l] + [l for l in msg.split("\n") if l]      # (the tool prints the call's message last)
    assert said == gold["stderr"]
    if gold["rc"] == 0:
        assert rc == 0
        assert open(dst, "rb").read().decode("latin1") == gold["out"]
    elif gold["rc"] == 1:
        # through regtools' own handler ("Unable to read header."), or htslib's exit(1) while it reads the header (4) / a record (2)
        assert rc == (1 if gold["stderr"][-1] == "Unable to read header." else 4 if name == "conflicting_idx" else 2)
    else:
        assert gold["rc"] == -6 and rc == (5 if "sample" in name else 3)      # abort() from the header's sample line / from a record
