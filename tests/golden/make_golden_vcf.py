#!/usr/bin/env python
"""Expected outputs of the REAL reference (oracle/_ref/regtools_ref variants annotate -o) for the inputs of tests/vcf_cases.py, twice:
with a GTF that is nowhere near a variant (every record gets NA: what tests/test_vcf_writer.py checks on the CPU) and with one whose
exons put some variants into splice regions (the GPU test through rgx_variants_annotate).  Dev container only."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vcf_cases  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")


def main():
    out = os.path.join(HERE, "vcf_writer")
    os.makedirs(out, exist_ok=True)
    manifest = {}
    with tempfile.TemporaryDirectory() as td:
        gtfs = {}
        for kind, text in (("far", vcf_cases.GTF_FAR), ("near", vcf_cases.GTF_NEAR)):
            gtfs[kind] = os.path.join(td, kind + ".gtf")
            open(gtfs[kind], "w").write(text)
        for name, data in sorted(vcf_cases.build(td).items()):
            src = os.path.join(td, name + ".vcf")
            open(src, "wb").write(data)
            for suffix, blob in vcf_cases.companions().get(name, {}).items():
                open(os.path.join(td, name + suffix), "wb").write(blob)
            manifest[name] = {}
            for kind in ("far", "near"):
                dst = os.path.join(out, "%s.%s.vcf" % (name, kind))
                r = subprocess.run([REF, "variants", "annotate", "-o", dst, src, gtfs[kind]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                manifest[name][kind] = r.returncode
                print(name, kind, "rc", r.returncode, os.path.getsize(dst) if os.path.exists(dst) else -1)
    json.dump(manifest, open(os.path.join(out, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
