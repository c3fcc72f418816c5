"""Region queries on a BAM with one out-of-order record: the reference's iterator ends at the first record it reads whose tid is not the
region's or whose pos is not below its end (hts_itr_next, hts.c:1946-1950).  Expected outputs: the real reference's
(tests/golden/unsorted, made by make_golden_unsorted.py).  This pins the ORACLE's restatement of the iterator (chunk list, seeks, end rule).
The product reads the region's member range [lo, hi) in one piece and keeps what overlaps: the same rows for every file whose records are in
index order, not for these (known deviations).  tools/fuzz/gpu_corrupt_bam.py meets the same difference on files whose
index went stale ("record" damage with -r)."""
import os

import pytest

import unsorted_cases as uc
from conftest import ROOT, run_oracle
from regtools_amd import synth

GOLD = os.path.join(ROOT, "tests", "golden", "unsorted")
CASES = [(k, r) for k in uc.KINDS for r in uc.REGIONS]


def expected(kind, region):
    return open(os.path.join(GOLD, uc.golden_name(kind, region)), "rb").read()


@pytest.fixture(scope="module")
def bams(tmp_path_factory):
    d = tmp_path_factory.mktemp("unsorted")
    return {k: uc.build(str(d / (k + ".bam")), k, synth.index) for k in uc.KINDS}


@pytest.mark.parametrize("kind,region", CASES, ids=["%s-%s" % c for c in CASES])
def test_oracle_equals_reference(bams, kind, region):
    rc, out, err = run_oracle(["-s", "XS", "-r", region, bams[kind]])
    assert rc == 0, err
    assert out == expected(kind, region)
