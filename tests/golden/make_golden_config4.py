#!/usr/bin/env python
"""SHA-256 digests of the REAL reference's three `cis-splice-effects identify` outputs (oracle/_ref) on configs[3]'s quartet at a TENTH of its
size -- 5 M reads, 6 250 genes (25 000 transcripts), 50 000 variants, the generator and seed bench.py uses (synth.write / synth.annotation, seed 4)
-- so that the GPU suite can check byte identity on a workload of that shape without the reference (which needs ~6 minutes for the full size,
~30 s for this one).  Only the digests and the counts are stored (tests/golden/cse/config4_reduced.json); the inputs are regenerated from the
seed at test time.  Dev container only (needs oracle/_ref):   python tests/golden/make_golden_config4.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from regtools_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
READS, GENES, VARIANTS, SEED = 5_000_000, 6_250, 50_000, 4


def main():
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        pre = os.path.join(td, "c4r")
        st = synth.write(pre + ".bam", READS, shape="short", seed=SEED, n_genes=GENES)
        ann = synth.annotation(pre, GENES, VARIANTS, seed=SEED, fasta=True)
        files = {x: pre + ".ref." + x for x in ("tsv", "vcf", "bed")}
        t = time.time()
        r = subprocess.run([REF, "cis-splice-effects", "identify", "-s", "XS", "-o", files["tsv"], "-v", files["vcf"], "-j", files["bed"],
                            ann["vcf"], pre + ".bam", ann["fasta"], ann["gtf"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        out = dict(reads=st["n_reads"], genes=GENES, variants=VARIANTS, seed=SEED, args=["-s", "XS"], reference_seconds=round(dt, 1),
                   inputs_sha256={k: hashlib.sha256(open(p, "rb").read()).hexdigest() for k, p in (("bam", pre + ".bam"), ("vcf", ann["vcf"]), ("gtf", ann["gtf"]), ("fasta", ann["fasta"]))},
                   outputs={x: dict(sha256=hashlib.sha256(open(p, "rb").read()).hexdigest(), bytes=os.path.getsize(p), lines=open(p, "rb").read().count(b"\n")) for x, p in files.items()})
        json.dump(out, open(os.path.join(HERE, "cse", "config4_reduced.json"), "w"), indent=1)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
