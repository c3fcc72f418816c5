"""A coordinate-sorted BAM in which ONE record is out of order (it claims another contig, or a position behind the region's end): the
reference's region iterator ends at the first record it reads whose tid is not the region's or whose pos is not below the region's end
(hts_itr_next, hts.c:1946-1950), whatever follows.  Inputs for tests/test_gpu_parity.py / tests/test_oracle_unsorted.py; the expected outputs
under tests/golden/unsorted/ are the REAL reference's (tests/golden/make_golden_unsorted.py)."""
import random
import struct

import bamio

REGIONS = ["chrA", "chrA:1-9000", "chrA:6000-12000", "chrB", "chrA:1-5000"]
KINDS = ["tid", "pos"]


def build(path, kind, index):
    rnd = random.Random(7)
    recs = []
    for tid in (0, 1):
        pos = 1000
        for k in range(500):
            pos += rnd.randrange(1, 40)
            recs.append(bamio.record(tid, pos, "%dM%dN%dM" % (20 + rnd.randrange(10), 200 + rnd.randrange(300), 25), qname="q%05d" % k, aux=bamio.tagA("XS", "+-"[k % 2])))
    r = bytearray(recs[230])
    if kind == "tid":
        struct.pack_into("<i", r, 4, 1)            # "I am on chrB"
    else:
        struct.pack_into("<i", r, 8, 9500)         # behind chrA:1-9000's end, inside chrA:6000-12000
    recs[230] = bytes(r)
    bamio.write_bam(path, [("chrA", 1000000), ("chrB", 1000000)], recs, block=4000)
    index(path)
    return path


def golden_name(kind, region):
    return "%s.%s.bed" % (kind, region.replace(":", "_"))
