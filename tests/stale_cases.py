"""BAM files whose index no longer describes them (TEST INFRASTRUCTURE): one member's payload is changed and re-compressed to another size
after the file was indexed, so every later offset of the index points somewhere else -- and files with an empty BGZF member in the middle,
indexed afterwards.  Region queries on them exercise the reference iterator's chunk-by-chunk seeks and its end rule (hts_itr_next,
hts.c:1924-1965).  Used by tests/test_stale_index.py (oracle against the real reference) and tests/test_gpu_region_iter.py (product against
the oracle)."""
import random
import struct
import zlib

import bamio

SHAPES = [("short", 20000, 3), ("fuzz", 8000, 4), ("long", 150, 5)]
N_VARIANTS = 24


def regions_of(shape):
    return ["1", "10:1000-200000", "2", "MT:1-5000"] if shape == "fuzz" else ["chr1", "chr2:1-90000000", "chr1:5000000-60000000", "chr3"]


def stale_variant(bam, rng):
    b = bytearray(bam)
    members = list(bamio.bgzf_members(bam))
    mi = rng.randrange(1, max(2, len(members) - 1))
    coff, payload, _ = members[mi]
    raw = bytearray(zlib.decompress(payload, -15))
    how = rng.choice(["field", "grow", "shrink"])
    if how == "field" and len(raw) > 64:
        k = rng.randrange(len(raw) - 40)
        raw[k:k + 4] = struct.pack("<I", rng.choice([0, 31, 32, 33, 1 << 27, 0x7fffffff, 0xffffffff, rng.randrange(1 << 16)]))
        newm = bamio.bgzf_member(bytes(raw))
    elif how == "grow":
        newm = bamio.bgzf_member(bytes(raw), level=0 if len(raw) < 65000 else 1)
    else:
        newm = bamio.bgzf_member(bytes(raw[:max(1, len(raw) - rng.randrange(1, 400))]))
    bl = struct.unpack_from("<H", b, coff + 16)[0] + 1
    b[coff:coff + bl] = newm
    return bytes(b)


def variants(bam, seed, shape):
    """yields (case number, file bytes, [two regions])"""
    rng = random.Random(1000 + seed)
    for case in range(N_VARIANTS):
        yield case, stale_variant(bam, rng), rng.sample(regions_of(shape), 2)


EMPTY_REGIONS = ["chr1", "chr2", "chr5:1-80000000", "chr9"]


def empty_member_variants(bam):
    """yields (case number, file bytes with an empty BGZF member in front of a random member); to be indexed AFTERWARDS"""
    members = list(bamio.bgzf_members(bam))
    rng = random.Random(77)
    for case in range(6):
        mi = rng.randrange(2, len(members) - 1)
        b = bytearray(bam)
        b[members[mi][0]:members[mi][0]] = bamio.EOF_MARKER
        yield case, bytes(b)
