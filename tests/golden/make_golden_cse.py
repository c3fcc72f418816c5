#!/usr/bin/env python
"""Golden outputs of the REAL reference (oracle/_ref) for `cis-splice-effects identify` on the deterministic synthetic
quartets of tests/cse_synth.py.  Stores only the argument lists and the reference's three output files per case
(the inputs are regenerated from the seed at test time).  Dev container only."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cse_synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
ARGSETS = [["-s", "XS"], ["-s", "RF"], ["-s", "FR", "-w", "60"], ["-s", "XS", "-E"], ["-s", "XS", "-I"], ["-s", "XS", "-e", "6", "-i", "6"],
           ["-s", "XS", "-a", "20"], ["-s", "XS", "-M", "1500"], ["-s", "XS", "-S"], ["-s", "XS", "-C"], ["-s", "intron-motif"],
           ["-s", "XS", "-E", "-I", "-w", "500"], ["-s", "RF", "-C", "-i", "10"], ["-s", "XS", "-m", "4000", "-a", "3"]]


def main():
    out = os.path.join(HERE, "cse")
    os.makedirs(out, exist_ok=True)
    cases = []
    with tempfile.TemporaryDirectory() as td:
        for seed in (1, 2, 3):
            q = cse_synth.build(os.path.join(td, "s%d" % seed), seed=seed, n_genes=10 + 2 * seed)
            for k, a in enumerate(ARGSETS):
                name = "cse_s%d_%02d" % (seed, k)
                files = {x: os.path.join(out, "%s.%s" % (name, x)) for x in ("tsv", "vcf", "bed")}
                r = subprocess.run([REF, "cis-splice-effects", "identify"] + a + ["-o", files["tsv"], "-v", files["vcf"], "-j", files["bed"],
                                                                               q["vcf"], q["bam"], q["fasta"], q["gtf"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                rows = open(files["tsv"]).read().count("\n") - 1 if os.path.exists(files["tsv"]) else -1
                cases.append(dict(name=name, seed=seed, n_genes=10 + 2 * seed, args=a, rc=r.returncode, rows=rows))
                print(name, a, "rc", r.returncode, "junction rows", rows)
    json.dump(cases, open(os.path.join(out, "manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
