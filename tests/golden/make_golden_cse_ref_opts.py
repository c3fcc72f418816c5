#!/usr/bin/env python
"""The option combinations the reference's integration tests run `cis-splice-effects identify` / `associate` with on ITS OWN data files
(tests/integration-test/test_cis_splice_effects_identify.py:100-348, test_cis_splice_effects_associate.py:80-200: upstream only asserts the exit status of
these; its two golden triplets cover `-s XS` and `-s RF` alone): the three output files of the REAL reference (oracle/_ref) for each, into
tests/golden/cse_ref_opts/.  The inputs are the data files already held under tests/golden/cse_ref/ and annot_ref/.  Dev container only."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
CSE_REF = os.path.join(HERE, "cse_ref")
OUT = os.path.join(HERE, "cse_ref_opts")

# (name, options, where upstream runs it)
IDENTIFY = [("e6_i6_S", ["-e", "6", "-i", "6", "-S"], ":100-122"), ("E", ["-E"], ":124-148"), ("I", ["-I"], ":150-174"), ("E_i6", ["-E", "-i", "6"], ":176-201"),
            ("e6_I", ["-e", "6", "-I"], ":203-229"), ("a30", ["-a", "30"], ":233-285"), ("m8039_M8039", ["-m", "8039", "-M", "8039"], ":287-305"), ("w5", ["-w", "5"], ":334-348"),
            # the same switches where they change something on this data: a window that reaches the junction, every intronic variant with a wide window
            ("w20000", ["-w", "20000"], "(-w, wide)"), ("I_w20000", ["-I", "-w", "20000"], "(-I -w, wide)"), ("E_I_S_a30_RF", ["-E", "-I", "-S", "-a", "30"], "(all of them)")]
ASSOCIATE = [("e6_i6_S", ["-e", "6", "-i", "6", "-S"]), ("E", ["-E"]), ("I", ["-I"]), ("E_i6", ["-E", "-i", "6"]), ("e6_I", ["-e", "6", "-I"]), ("w20000", ["-w", "20000"])]


def main():
    os.makedirs(OUT, exist_ok=True)
    quartet = [os.path.join(CSE_REF, x) for x in ("test1.vcf", "test_hcc1395.2.bam", "test_chr22.fa", "test_ensemble_chr22.2.gtf")]
    bed = os.path.join(HERE, "annot_ref", "junctions_extract.bed")
    cases = []
    for name, opts, where in IDENTIFY:
        for strand in ("XS", "RF") if name != "E_I_S_a30_RF" else ("RF",):
            n = "id_%s_%s" % (strand, name) if name != "E_I_S_a30_RF" else "id_" + name
            files = {x: os.path.join(OUT, "%s.%s" % (n, x)) for x in ("tsv", "vcf", "bed")}
            r = subprocess.run([REF, "cis-splice-effects", "identify"] + opts + ["-s", strand, "-o", files["tsv"], "-v", files["vcf"], "-j", files["bed"]] + quartet,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            cases.append(dict(name=n, cmd="identify", args=opts + ["-s", strand], rc=r.returncode, upstream=where, rows=open(files["tsv"]).read().count("\n") - 1))
            print(cases[-1])
    for name, opts in ASSOCIATE:
        n = "as_" + name
        files = {x: os.path.join(OUT, "%s.%s" % (n, x)) for x in ("tsv", "vcf", "bed")}
        r = subprocess.run([REF, "cis-splice-effects", "associate"] + opts + ["-o", files["tsv"], "-v", files["vcf"], "-j", files["bed"], quartet[0], bed, quartet[2], quartet[3]],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        cases.append(dict(name=n, cmd="associate", args=opts, rc=r.returncode, rows=open(files["tsv"]).read().count("\n") - 1))
        print(cases[-1])
    # `junctions annotate` (upstream: one golden, test_junctions_annotate.py:34-42): both junction files and both annotations the reference's tests hold, with and without -S
    annot = os.path.join(HERE, "annot_ref")
    for bk, b in (("hccj", os.path.join(annot, "test_hcc1395_junctions.bed")), ("jex", bed)):
        for gk, g in (("gtf1", os.path.join(annot, "test_ensemble_chr22.gtf")), ("gtf2", quartet[3])):
            for opts in ([], ["-S"]):
                n = "ja_%s_%s%s" % (bk, gk, "_S" if opts else "")
                out = os.path.join(OUT, n + ".tsv")
                r = subprocess.run([REF, "junctions", "annotate"] + opts + ["-o", out, b, quartet[2], g], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                cases.append(dict(name=n, cmd="junctions-annotate", args=opts, bed=os.path.relpath(b, HERE), gtf=os.path.relpath(g, HERE), rc=r.returncode, rows=open(out).read().count("\n") - 1))
                print(cases[-1])
    json.dump(cases, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
