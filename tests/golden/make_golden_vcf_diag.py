#!/usr/bin/env python
"""What the REAL reference (oracle/_ref/regtools_ref variants annotate -o) says and writes for the inputs of tests/vcf_diag_cases.py, and how it ends:
tests/golden/vcf_writer/diagnostics.json = {name: {"rc": status (negative: the signal), "stderr": [htslib's lines], "out": the output file when rc is 0}}.
htslib puts __FILE__ into some messages: the build's path is cut down to the file's name.  Dev container only."""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vcf_cases  # noqa: E402
import vcf_diag_cases  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")


def main():
    gold = {}
    with tempfile.TemporaryDirectory() as td:
        gtf = os.path.join(td, "far.gtf")
        open(gtf, "w").write(vcf_cases.GTF_FAR)
        for name, text in sorted(vcf_diag_cases.CASES.items()):
            src, dst = os.path.join(td, name + ".vcf"), os.path.join(td, name + ".out.vcf")
            open(src, "w").write(text)
            r = subprocess.run([REF, "variants", "annotate", "-o", dst, src, gtf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            lines = r.stderr.decode("latin1").split("\n")
            start = max(k for k, l in enumerate(lines) if l.startswith("Output file")) + 2       # behind the option echo and its blank line
            said = [re.sub(r"\[[^\] ]*/(vcf\.c:\d+ )", r"[\1", l) for l in lines[start:] if l]
            gold[name] = {"rc": r.returncode, "stderr": said}
            if r.returncode == 0: gold[name]["out"] = open(dst, "rb").read().decode("latin1")
            print(name, r.returncode, said)
    json.dump(gold, open(os.path.join(HERE, "vcf_writer", "diagnostics.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
