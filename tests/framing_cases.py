"""Inputs that stress the speculative record framing (TEST INFRASTRUCTURE): byte strings that look like BAM record heads are
planted inside `B` aux arrays, so a segment's first plausible offset is often not a record start.

variants
  huge      decoy block_size 100,000,000: its chain runs past the end of the stream
  to_end    decoy block_size patched so that its chain ends EXACTLY at the end of the stream: a readable chain with a wrong exit
  insane    like to_end, plus one real record in the middle whose l_read_name is 0: bam_read1 gives up there
            (htslib sam.c:421-423) and the reference silently reports only what came before
"""
import os
import random
import shutil
import struct

import bamio

CONTIGS = [("chrA", 5000000), ("chrB", 3000000)]
MAGIC = 123456789


def build_big_header(path, seed=9):
    """6,000 contigs: the BAM header alone spans five BGZF members, tids need 13 bits, output order is by contig NAME string."""
    rnd = random.Random(seed)
    contigs = [("s%d" % i, 200000 + 7 * i) for i in range(6000)]
    recs = []
    for tid in (3, 17, 170, 1700, 4000, 5999):
        pos = 100
        for k in range(60):
            pos += rnd.randint(1, 400)
            cigar = "%dM%dN%dM" % (rnd.randint(8, 60), rnd.choice([90, 500, 1200]), rnd.randint(8, 60)) if k % 3 else "50M"
            recs.append(bamio.record(tid, pos, cigar, flag=rnd.choice([0, 16]), qname="h%d_%d" % (tid, k), aux=bamio.tagA("XS", "+-"[k & 1])))
    from regtools_amd import synth
    bamio.write_bam(path, contigs, recs)
    synth.index(path)
    return path


def build_ultralong(path, seed=13):
    """Reads of 60-250 kb (records of up to 375 KB: dozens of 16 KiB framing segments in which no record starts) between ordinary ones."""
    rnd = random.Random(seed)
    contigs = [("chrL", 40000000)]
    recs, pos = [], 1000
    for k in range(90):
        pos += rnd.randint(50, 3000)
        if k % 3 == 0:
            blocks = rnd.randint(3, 9)
            ops = []
            for b in range(blocks):
                ops.append((rnd.randint(8000, 30000), 0))
                if b + 1 < blocks:
                    ops.append((rnd.choice([120, 900, 15000]), 3))
            if rnd.random() < 0.5:
                ops.insert(1, (rnd.randint(1, 40), 1))
            recs.append(bamio.record(0, pos, ops, flag=rnd.choice([0, 16]), qname="ul%d" % k, aux=bamio.tagA("XS", "+-"[k & 1])))
        else:
            recs.append(bamio.record(0, pos, "%dM%dN%dM" % (rnd.randint(8, 70), rnd.choice([100, 2000]), rnd.randint(8, 70)), qname="s%d" % k, aux=bamio.tagA("XS", "+-"[k & 1])))
    # one read with 12,001 CIGAR operations (6,000 introns): the wave-per-read emitter walks it in LDS-sized pieces
    ops = []
    for i in range(6000):
        ops += [(rnd.randint(9, 30), 0), (rnd.choice([80, 100, 700]), 3)]
    ops.append((25, 0))
    pos += 5000
    recs.append(bamio.record(0, pos, ops, flag=16, qname="manyops", aux=bamio.tagA("XS", "-")))
    recs.append(bamio.record(0, pos + 10, "30M100N30M", qname="after", aux=bamio.tagA("XS", "-")))
    from regtools_amd import synth
    bamio.write_bam(path, contigs, recs)
    synth.index(path)
    return path


def build(variant, path, seed=5, n_records=4000):
    if variant == "big_header":
        return build_big_header(path)
    if variant == "ultralong":
        return build_ultralong(path)
    rnd = random.Random(seed)
    first = 100000000 if variant == "huge" else MAGIC
    decoy = struct.pack("<iiiIIiiii", first, 0, 5, 1, 0, 0, -1, -1, 0) + b"\0"
    recs, pos = [], 100
    for i in range(n_records):
        tid = 0 if i < n_records * 2 // 3 else 1
        if i == n_records * 2 // 3:
            pos = 100
        pos += rnd.randint(1, 300)
        aux = b""
        if rnd.random() < 0.5:
            body = bytes(rnd.randrange(1, 250) for _ in range(rnd.randint(0, 40))) + decoy + bytes(rnd.randrange(1, 250) for _ in range(rnd.randint(0, 9)))
            aux += b"ZBBC" + struct.pack("<i", len(body)) + body
        aux += bamio.tagA("XS", "+-"[i & 1])
        cigar = "%dM%dN%dM" % (rnd.randint(8, 60), rnd.randint(70, 5000), rnd.randint(8, 60)) if rnd.random() < 0.4 else "%dM" % rnd.randint(30, 150)
        recs.append(bamio.record(tid, pos, cigar, flag=rnd.choice([0, 16, 99, 147]), qname="q%d" % i, aux=aux))
    hdr = bamio.header_bytes(CONTIGS)
    stream = bytearray(b"".join(recs))
    from regtools_amd import synth
    bamio.write_bam(path, CONTIGS, [bytes(stream)])
    synth.index(path)                                   # index of the clean layout; only `-r` would look inside it
    if variant == "huge":
        return path
    lim = len(hdr) + len(stream)
    key, k = struct.pack("<i", MAGIC), 0
    while True:
        k = stream.find(key, k)
        if k < 0:
            break
        struct.pack_into("<i", stream, k, lim - (len(hdr) + k) - 4)
        k += 4
    if variant == "insane":
        off = sum(len(r) for r in recs[: n_records // 2])
        stream[off + 12] = 0                            # l_read_name = 0 in the middle record
    bai = path + ".bai.keep"
    shutil.copy(path + ".bai", bai)
    bamio.write_bam(path, CONTIGS, [bytes(stream)])
    os.replace(bai, path + ".bai")
    return path
