"""Regenerates tests/golden/barcodes/: the BAMs of tests/barcode_cases.py and what the REAL reference (oracle/_ref/regtools_ref, built from
/root/reference by oracle/Makefile) prints for `junctions extract -b` on them.  Run in the dev container only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import barcode_cases as bc  # noqa: E402
from regtools_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
for name in bc.CASES:
    bam = os.path.join(bc.GOLD, name + ".bam")
    bc.build(name, bam)
    synth.index(bam)
    for args in bc.ARGS[name]:
        stem = os.path.join(bc.GOLD, "%s.%s" % (name, bc.arg_tag(args)))
        r = subprocess.run([REF, "junctions", "extract"] + args + ["-o", stem + ".bed", "-b", stem + ".barcodes", bam], capture_output=True)
        assert r.returncode == 0, r.stderr[-300:]
        print(name, args, sum(1 for _ in open(stem + ".bed")), "rows")
