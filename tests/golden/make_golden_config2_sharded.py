#!/usr/bin/env python
"""SHA-256 of the REAL reference's BED12 (oracle/_ref) on configs[2]'s workload at a TENTH of its size: eight coordinate slices of 5 M reads
(SURVEY 8d "Config 3": the generator and seed bench.py's ranks use -- synth.generate(reads, seed=1, slice_index=k, n_slices=8)) joined into
ONE 40 M-read BAM (tests/slices.py: byte-level concatenation of the slices' record members + the merged index).  The GPU suite regenerates
the slices from the seed, extracts them as eight shards, merges, and compares digests (tests/test_gpu_config2_sharded.py); only the digest
and the counts are stored (tests/golden/config2_sharded.json).  Dev container only (needs oracle/_ref):
    python tests/golden/make_golden_config2_sharded.py"""
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
from regtools_amd import synth  # noqa: E402
import slices  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
READS_PER_SLICE, N_SLICES, SEED = 5_000_000, 8, 1


def main():
    parts = [synth.generate(READS_PER_SLICE, shape="short", seed=SEED, slice_index=k, n_slices=N_SLICES) for k in range(N_SLICES)]
    bam = slices.concat_slices([p[0] for p in parts])
    bai = slices.merge_bai([p[0] for p in parts], [p[1] for p in parts])
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        path = os.path.join(td, "c2s.bam")
        open(path, "wb").write(bam)
        open(path + ".bai", "wb").write(bai)
        t = time.time()
        r = subprocess.run([REF, "junctions", "extract", "-s", "XS", "-o", path + ".bed", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        bed = open(path + ".bed", "rb").read()
    out = dict(reads_per_slice=READS_PER_SLICE, n_slices=N_SLICES, seed=SEED, args=["-s", "XS"], reference_seconds=round(dt, 1),
               reads=sum(p[2]["n_reads"] for p in parts), bam_sha256=hashlib.sha256(bam).hexdigest(), bai_sha256=hashlib.sha256(bai).hexdigest(),
               bed12=dict(sha256=hashlib.sha256(bed).hexdigest(), bytes=len(bed), lines=bed.count(b"\n")))
    json.dump(out, open(os.path.join(HERE, "config2_sharded.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
