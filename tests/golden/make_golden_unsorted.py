#!/usr/bin/env python
"""Expected outputs of the REAL reference (oracle/_ref/regtools_ref junctions extract -r) for tests/unsorted_cases.py.  Dev container only."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import unsorted_cases  # noqa: E402
from regtools_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
out = os.path.join(HERE, "unsorted")
os.makedirs(out, exist_ok=True)
with tempfile.TemporaryDirectory() as td:
    for kind in unsorted_cases.KINDS:
        bam = unsorted_cases.build(os.path.join(td, kind + ".bam"), kind, synth.index)
        for region in unsorted_cases.REGIONS:
            dst = os.path.join(out, unsorted_cases.golden_name(kind, region))
            r = subprocess.run([REF, "junctions", "extract", "-s", "XS", "-r", region, "-o", dst, bam], capture_output=True)
            assert r.returncode == 0, r.stderr[-300:]
            print(kind, region, open(dst).read().count("\n"))
