"""Region queries with an index that does not describe its file (tests/stale_cases.py): the ORACLE's restatement of hts_itr_query /
hts_itr_next against the real reference, where that is built (oracle/_ref, the dev container and -- it travels -- the GPU box).  The product is
held to the oracle on the same files in tests/test_gpu_region_iter.py."""
import os
import subprocess

import pytest

import stale_cases as sc
from conftest import ROOT, run_oracle
from regtools_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")


@pytest.mark.skipif(not os.path.exists(REF), reason="the real reference is not built here")
@pytest.mark.parametrize("shape,n,seed", sc.SHAPES)
def test_oracle_equals_reference_on_stale_indexes(tmp_path, shape, n, seed):
    base = str(tmp_path / "base.bam")
    synth.write(base, n, shape=shape, seed=seed)
    bam, bai = open(base, "rb").read(), open(base + ".bai", "rb").read()
    path, bed = str(tmp_path / "case.bam"), str(tmp_path / "ref.bed")
    compared = 0
    for case, data, regions in sc.variants(bam, seed, shape):
        open(path, "wb").write(data)
        open(path + ".bai", "wb").write(bai)
        for region in regions:
            args = ["-s", "XS", "-r", region]
            try:
                rr = subprocess.run([REF, "junctions", "extract"] + args + ["-o", bed, path], capture_output=True, timeout=20)
            except subprocess.TimeoutExpired:
                continue
            if rr.returncode not in (0, 1):
                continue                                     # the reference itself died on this file: nothing to compare with
            rc, out, _ = run_oracle(args + [path])
            assert (rc != 0) == (rr.returncode != 0), (case, region)
            if rc == 0:
                assert out == open(bed, "rb").read(), (case, region)
            compared += 1
    assert compared >= sc.N_VARIANTS


@pytest.mark.skipif(not os.path.exists(REF), reason="the real reference is not built here")
def test_oracle_equals_reference_with_an_empty_member_mid_file(tmp_path):
    base = str(tmp_path / "base.bam")
    synth.write(base, 30000, shape="short", seed=9)
    path, bed = str(tmp_path / "case.bam"), str(tmp_path / "ref.bed")
    compared = 0
    for case, data in sc.empty_member_variants(open(base, "rb").read()):
        open(path, "wb").write(data)
        if os.path.exists(path + ".bai"):
            os.remove(path + ".bai")
        try:
            synth.index(path)
        except RuntimeError:
            continue
        for region in sc.EMPTY_REGIONS:
            args = ["-s", "XS", "-r", region]
            rr = subprocess.run([REF, "junctions", "extract"] + args + ["-o", bed, path], capture_output=True, timeout=60)
            if rr.returncode not in (0, 1):
                continue
            rc, out, _ = run_oracle(args + [path])
            assert (rc != 0) == (rr.returncode != 0), (case, region)
            if rc == 0:
                assert out == open(bed, "rb").read(), (case, region)
            compared += 1
    assert compared >= 4
