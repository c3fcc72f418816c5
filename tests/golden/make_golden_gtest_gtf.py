#!/usr/bin/env python
"""The GtfParser gtest's transcripts (tests/lib/gtf/test_gtf_parser.cc:122-279: three exons of ENST00000263253 handed over out of order, on '+' and on '-')
as `junctions annotate` sees them: whether the exons were sorted the way sort_exons_within_transcripts sorts them decides which junctions are known,
which exons are skipped and where donors and acceptors are.  Inputs (the gtest's GTF lines, a small genome of contig 22, BED12 rows around the two introns)
and the outputs of the REAL reference (oracle/_ref) go to tests/golden/gtest_gtf/.  Dev container only."""
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gtest_gtf_known_answers as K  # noqa: E402  (the gtest's lines)

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
OUT = os.path.join(HERE, "gtest_gtf")


def bed_rows(strand):
    rows, k = [], 0
    # every pairing of coordinates around the exon ends 9800 / 10000 and the exon starts 9900 / 10100 (the off-by-one forms included), one intron that skips
    # the middle exon, one that lies inside an exon, one outside the transcript
    pairs = [(a + da, b + db) for a, b in ((9800, 9900), (10000, 10100), (9800, 10100)) for da in (-1, 0, 1) for db in (-1, 0, 1)]
    pairs += [(9720, 9780), (9850, 10050), (9000, 9500), (10300, 10900), (9750, 10150)]
    for js, je in pairs:
        for s in (strand, "-" if strand == "+" else "+"):
            k += 1
            a = 20
            rows.append("22\t%d\t%d\tJUNC%08d\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n" % (js - a, je + a, k, k % 7 + 1, s, js - a, je + a, a, a, je - js + a))
    return "".join(rows)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = random.Random(22)
    seq = "".join(rng.choice("ACGT") for _ in range(16000))
    with open(os.path.join(OUT, "genome.fa"), "w") as f:
        f.write(">22\n")
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60] + "\n")
    with open(os.path.join(OUT, "genome.fa.fai"), "w") as f:
        f.write("22\t%d\t4\t60\t61\n" % len(seq))
    for name, text, strand in (("ps", K.SORT_PS, "+"), ("ns", K.SORT_NS, "-")):
        open(os.path.join(OUT, name + ".gtf"), "w").write(text)
        open(os.path.join(OUT, name + ".bed"), "w").write(bed_rows(strand))
        for flag in ([], ["-S"]):
            out = os.path.join(OUT, "%s%s.out" % (name, "_S" if flag else ""))
            r = subprocess.run([REF, "junctions", "annotate"] + flag + ["-o", out, os.path.join(OUT, name + ".bed"), os.path.join(OUT, "genome.fa"), os.path.join(OUT, name + ".gtf")],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            print(name, flag, "rc", r.returncode, open(out).read().count("\n"), "lines")
            assert r.returncode == 0


if __name__ == "__main__":
    main()
