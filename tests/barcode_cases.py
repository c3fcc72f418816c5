"""Inputs for the `junctions extract -b` parity tests (TEST INFRASTRUCTURE ONLY): small hand-made BAMs whose reads carry CB:Z cell
barcodes, committed under tests/golden/barcodes/ next to what the real reference printed for them (make_golden_barcodes.py)."""
import os
import random
import struct

import bamio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "barcodes")

# name -> (builder args, argument lists run through the reference)
CASES = {
    "pools": dict(n=24000, seed=5, loci=26, pools=(3, 40, 3000), absent=0.03, empty=0.02),
    "few": dict(n=3000, seed=6, loci=9, pools=(1, 2, 14), absent=0.0, empty=0.0),
    "rehash": dict(n=30000, seed=7, loci=3, pools=(12, 13, 6000), absent=0.01, empty=0.0),
    "qmark": dict(n=2000, seed=8, loci=5, pools=(2, 2, 2), absent=0.3, empty=0.1, literal_qmark=True),
}
ARGS = {
    "pools": [["-s", "XS"], ["-s", "RF", "-a", "4"], ["-s", "XS", "-r", "chrZ:20000-70000"], ["-s", "XS", "-m", "200", "-M", "380"]],
    "few": [["-s", "XS"], ["-s", "FR"]],
    "rehash": [["-s", "XS"]],
    "qmark": [["-s", "XS"], ["-s", "XS", "-t", "NH"]],
}


def arg_tag(args):
    return "_".join(a.strip("-").replace(":", "-") for a in args)


def tagB(tag, vals):
    return tag.encode() + b"Bs" + struct.pack("<i", len(vals)) + b"".join(struct.pack("<h", v) for v in vals)


def build(name, path):
    """Writes <path> (coordinate-sorted BAM); the caller indexes it."""
    c = CASES[name]
    rng = random.Random(c["seed"])
    alphabet = "ACGT"
    bcs = ["".join(rng.choice(alphabet) for _ in range(rng.choice([16, 16, 16, 7, 9, 24, 1, 40]))) + "-1" for _ in range(max(c["pools"]))]
    if c.get("literal_qmark"):
        bcs[0] = "?"
    recs = []
    for k in range(c["n"]):
        j = rng.randrange(c["loci"])
        pos = 1000 + j * 5000 + (0 if j % 3 else rng.randrange(3))
        flag = rng.choice([0, 16, 99, 147, 83, 163])
        aux = b""
        if rng.random() < 0.5:
            aux += tagB("ZB", [1, 2, 3])                          # tags of every width before CB: the aux walk must skip them
        aux += bamio.tagA("XS", "+-"[j & 1]) + b"NHC" + bytes([1 + j % 3])
        r = rng.random()
        if r < c["absent"]:
            pass
        elif r < c["absent"] + c["empty"]:
            aux += bamio.tagZ("CB", "")
        else:
            bc = rng.choice(bcs[:c["pools"][j % 3]])
            aux += (b"CBH" + bc.encode() + b"\0") if rng.random() < 0.02 else bamio.tagZ("CB", bc)
        if rng.random() < 0.3:
            aux += bamio.tagZ("UB", "ACGTACGTAC")
        cig = "%dM%dN%dM" % (20, 100 + j * 13, 30)
        if k % 50 == 0:
            cig = "20M200N30M300N25M"
        elif k % 41 == 0:
            cig = "3M%dN40M" % (100 + j * 13)                     # left anchor too short: counted, maybe never printed
        recs.append((pos, k, bamio.record(0, pos, cig, flag=flag, qname="q%05d" % k, aux=aux)))
    recs.sort(key=lambda t: (t[0], t[1]))
    bamio.write_bam(path, [("chrZ", 10000000)], [r for _, _, r in recs])
