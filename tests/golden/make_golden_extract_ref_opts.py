#!/usr/bin/env python
"""`junctions extract` on the two BAMs of real aligner output the reference's tests hold (test_hcc1395.bam, test_hcc1395.2.bam), under option combinations its six goldens
(tests/integration-test/test_junctions_extract.py:34-85) do not reach: the other strand rules (FR, intron-motif with the chr22 genome), another strand tag, anchor / intron bounds at
their edges, regions of every form.  Outputs of the REAL reference (oracle/_ref) into tests/golden/extract_ref_opts/.  Dev container only."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
OUT = os.path.join(HERE, "extract_ref_opts")
BAMS = {"hcc": "test_hcc1395.bam", "hcc2": os.path.join("cse_ref", "test_hcc1395.2.bam")}
FASTA = os.path.join("cse_ref", "test_chr22.fa")
OPTS = [["-s", "FR"], ["-s", "FR", "-a", "30"], ["-s", "XS", "-t", "NH"], ["-s", "XS", "-t", "XS"], ["-s", "XS", "-a", "0"], ["-s", "XS", "-a", "1", "-m", "0"], ["-s", "RF", "-a", "50"],
        ["-s", "RF", "-M", "1000"], ["-s", "XS", "-m", "1000", "-M", "10000"], ["-s", "XS", "-m", "8040"], ["-s", "XS", "-M", "8038"], ["-s", "FR", "-m", "8039", "-M", "8039"],
        ["-s", "XS", "-r", "1"], ["-s", "XS", "-r", "22"], ["-s", "FR", "-r", "1:22000000-23000000"], ["-s", "XS", "-r", "1:22405013"], ["-s", "XS", "-r", "1:22,405,013-22,413,052"],
        ["-s", "RF", "-r", "1:22413052-22413053"], ["-s", "XS", "-r", "22:1-30000000"], ["-s", "XS", "-r", "22:29000000"], ["-s", "XS", "-r", "2"], ["-s", "XS", "-r", "X:1-2"],
        ["-s", "intron-motif"], ["-s", "intron-motif", "-r", "22"], ["-s", "intron-motif", "-a", "0", "-m", "0"]]


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = []
    for key, bam in BAMS.items():
        for k, opts in enumerate(OPTS):
            motif = "intron-motif" in opts
            name = "%s_%02d" % (key, k)
            out = os.path.join(OUT, name + ".out")
            r = subprocess.run([REF, "junctions", "extract"] + opts + ["-o", out, os.path.join(HERE, bam)] + ([os.path.join(HERE, FASTA)] if motif else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if not os.path.exists(out):
                open(out, "w").close()
            cases.append(dict(name=name, bam=bam, args=opts, fasta=FASTA if motif else None, rc=r.returncode, rows=open(out).read().count("\n")))
            print(cases[-1])
    json.dump(cases, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print(len(cases), "cases,", len({open(os.path.join(OUT, c["name"] + ".out"), "rb").read() for c in cases}), "distinct outputs")


if __name__ == "__main__":
    main()
