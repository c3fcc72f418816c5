#!/usr/bin/env python
"""Regenerates tests/golden/: runs the REAL reference (oracle/_ref/regtools_ref, compiled from /root/reference by
oracle/Makefile) on hand-made and synthetic inputs and stores inputs + expected outputs as fixtures.

Only runs in the dev container (needs /root/reference through oracle/_ref).  The committed fixtures are data:
small BAM/BAI inputs, generator specs (shape/reads/seed -- the generator is deterministic) and the reference's
stdout for each argument list.  manifest.json lists every case.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bamio  # noqa: E402
from regtools_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
cases = []


def run_ref(bam, args, after=()):
    r = subprocess.run([REF, "junctions", "extract"] + args + [bam] + list(after), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout


def add_case(name, args, bam=None, synth_spec=None, tmp_bam=None, framing=None, cse=None, after=()):
    """cse: the BAM (and the FASTA named by the "<fasta>" placeholder in `after`) of a deterministic tests/cse_synth.py quartet."""
    path = tmp_bam if tmp_bam else os.path.join(HERE, bam)
    rc, out = run_ref(path, args, [cse["_fasta"] if a == "<fasta>" else a for a in after])
    exp = "%s.expected" % name
    with open(os.path.join(HERE, "expected", exp), "wb") as f:
        f.write(out)
    c = dict(name=name, args=args, bam=bam, synth=synth_spec, framing=framing, expected=exp, rc=rc, rows=out.count(b"\n"))
    if cse:
        c["cse"] = {k: v for k, v in cse.items() if not k.startswith("_")}
    if after:
        c["after"] = list(after)
    cases.append(c)
    print("%-44s rc=%d rows=%d" % (name, rc, out.count(b"\n")))


def main():
    os.makedirs(os.path.join(HERE, "expected"), exist_ok=True)
    xs_plus = bamio.tagA("XS", "+")

    # (i) the CIGAR edge table of SURVEY.md 9.7: one read each, XS:A:+, contig chrT
    cigars = ["20M100N20M100N20M", "5S20M100N20M", "10M1I10M100N20M", "20M100N10M1D10M100N20M", "20M100N20M5S", "20M100N100N20M",
              "100N20M", "20M100N", "20M2P100N20M", "5H20M100N20M5H", "10=1X100N20M", "10M1X9M100N20M", "20M100N9M1X10M", "20M69N20M",
              "20M70N20M", "20M500000N20M", "20M500001N20M", "7M100N20M", "8M100N20M", "20M100N7M", "20M100N8M",
              "20M100N20M1I20M100N20M", "20M100N5M100N20M", "20M1D100N20M", "20M100N1D20M", "20M100N1I20M", "3S", "20M",
              "10M1B10M100N20M", "4M0N4M100N30M", "30M100N30M0I30M100N30M", "20M100N20M3D20M100N20M3X100N20M"]
    recs = [bamio.record(0, 1000000 * (i + 1), c, qname="e%02d" % i, aux=xs_plus) for i, c in enumerate(cigars)]
    bamio.write_bam(os.path.join(HERE, "edge_cigars.bam"), [("chrT", 60000000)], recs)
    synth.index(os.path.join(HERE, "edge_cigars.bam"))
    add_case("edge_cigars.XS", ["-s", "XS"], bam="edge_cigars.bam")
    add_case("edge_cigars.XS.a0", ["-s", "XS", "-a", "0"], bam="edge_cigars.bam")
    add_case("edge_cigars.XS.a1.m0.M4000000000", ["-s", "XS", "-a", "1", "-m", "0", "-M", "4000000000"], bam="edge_cigars.bam")
    add_case("edge_cigars.RF", ["-s", "RF"], bam="edge_cigars.bam")

    # (ii) strand rules: 16 flag nibbles x same junction; tag variants; class-2 collisions ('?' then '.')
    recs = []
    for k in range(16):
        recs.append(bamio.record(0, 1000 + 1000 * k, "20M100N20M", flag=k << 4, qname="f%02d" % k, aux=xs_plus))
    tagcases = [bamio.tagA("XS", "+"), bamio.tagA("XS", "-"), bamio.tagA("XS", "."), bamio.tagZ("XS", "+"), bamio.tagA("XS", 0), b"",
                bamio.tagA("NH", "1") + bamio.tagA("XS", "-"), bamio.tagZ("MD", "20") + bamio.tagA("XS", "+"),
                b"XSB" + b"C" + (3).to_bytes(4, "little") + b"\1\2\3" + bamio.tagA("XS", "-"),       # first XS is type B -> '?'
                b"ZBBS" + (2).to_bytes(4, "little") + b"\0\0\0\0" + bamio.tagA("XS", "-"),           # B array before the tag
                bamio.tagA("ts", "-") + bamio.tagA("XS", "+"), b"ASi" + (5).to_bytes(4, "little") + bamio.tagA("XS", "-"),
                b"XFf" + b"\0\0\x80\x3f" + b"XDd" + b"\0" * 8 + b"XSs\x01\x00" + bamio.tagA("XS", "+")]  # XS:s first -> '?'
    for k, aux in enumerate(tagcases):
        recs.append(bamio.record(0, 50000 + 1000 * k, "25M200N25M", qname="t%02d" % k, aux=aux))
    # same junction, strand '?' (no tag) then '.', then '?' again on another junction
    recs.append(bamio.record(0, 90000, "30M300N30M", qname="c0"))
    recs.append(bamio.record(0, 90000, "30M300N30M", qname="c1", aux=bamio.tagA("XS", ".")))
    recs.append(bamio.record(0, 91000, "30M300N30M", qname="c2", aux=bamio.tagA("XS", ".")))
    recs.append(bamio.record(0, 91000, "30M300N30M", qname="c3"))
    recs.append(bamio.record(0, 92000, "30M300N30M", qname="c4", aux=bamio.tagA("XS", "+")))
    recs.append(bamio.record(0, 92000, "30M300N30M", qname="c5", aux=bamio.tagA("XS", "-")))
    recs.append(bamio.record(0, 92005, "25M300N30M", qname="c6", aux=bamio.tagA("XS", "+")))
    bamio.write_bam(os.path.join(HERE, "strand.bam"), [("chrS", 1000000)], recs)
    synth.index(os.path.join(HERE, "strand.bam"))
    for s in ("XS", "RF", "FR"):
        add_case("strand.%s" % s, ["-s", s], bam="strand.bam")
    add_case("strand.XS.tag_ts", ["-s", "XS", "-t", "ts"], bam="strand.bam")
    add_case("strand.XS.tag_short", ["-s", "XS", "-t", "X"], bam="strand.bam")

    # (iii) contig names 1,10,2,MT: output order is by name STRING; (vi) records straddle 300-byte members;
    #       unmapped reads (tid -1) at the end; secondary/duplicate/qc-fail flags are NOT filtered
    contigs = [("1", 500000), ("10", 500000), ("2", 500000), ("MT", 16569)]
    recs = []
    for tid in range(4):
        for k in range(40):
            pos = 100 + 97 * k
            cg = "%dM%dN%dM" % (10 + k % 17, 100 + 13 * (k % 5), 12 + k % 9)
            recs.append(bamio.record(tid, pos, cg, flag=[0, 256, 512, 1024, 2048, 16][k % 6], qname="m%d_%d" % (tid, k), aux=bamio.tagA("XS", "+-"[k % 2])))
    recs.append(bamio.record(-1, -1, "10M", flag=4, qname="u0"))
    recs.append(bamio.record(-1, -1, [], flag=4, qname="u1"))
    bamio.write_bam(os.path.join(HERE, "contigs.bam"), contigs, recs, block=300)
    synth.index(os.path.join(HERE, "contigs.bam"))
    add_case("contigs.XS", ["-s", "XS"], bam="contigs.bam")
    for reg in ("10", "2:1000-2000", "MT:1-100000", "1:3990-3990", "1:3,000-4,000", "2:500", "10:1-2"):
        add_case("contigs.XS.r_%s" % reg.replace(":", "_").replace(",", ""), ["-s", "XS", "-r", reg], bam="contigs.bam")
    add_case("contigs.XS.r_missing", ["-s", "XS", "-r", "chrNope:1-100"], bam="contigs.bam")
    add_case("contigs.XS.r_reversed", ["-s", "XS", "-r", "1:2000-1000"], bam="contigs.bam")

    # (vi) an empty BGZF member in the middle of the record stream ends iteration there (htslib 1.2.1, bgzf.c:548-578)
    stream = b"".join(recs[:120])
    data = bytearray()
    data += bamio.bgzf_member(bamio.header_bytes(contigs))
    half = len(b"".join(recs[:60]))
    data += bamio.bgzf_member(stream[:half]) + bamio.EOF_MARKER + bamio.bgzf_member(stream[half:]) + bamio.EOF_MARKER
    with open(os.path.join(HERE, "empty_member.bam"), "wb") as f:
        f.write(bytes(data))
    synth.index(os.path.join(HERE, "empty_member.bam"))
    add_case("empty_member.XS", ["-s", "XS"], bam="empty_member.bam")

    # truncated file: cut in the middle of the last data member (no EOF marker)
    whole = open(os.path.join(HERE, "contigs.bam"), "rb").read()
    with open(os.path.join(HERE, "truncated.bam"), "wb") as f:
        f.write(whole[: len(whole) - 28 - 40])
    with open(os.path.join(HERE, "truncated.bam.bai"), "wb") as f:
        f.write(open(os.path.join(HERE, "contigs.bam.bai"), "rb").read())
    add_case("truncated.XS", ["-s", "XS"], bam="truncated.bam")

    # (iv) deterministic synthetic inputs: only the spec and the expected output are stored
    import tempfile
    specs = [("short", 30000, 21), ("short", 30000, 22), ("long", 400, 23), ("fuzz", 20000, 24), ("fuzz", 20000, 25)]
    argsets = [["-s", "XS"], ["-s", "RF"], ["-s", "FR", "-a", "20"], ["-s", "XS", "-m", "200", "-M", "3000"]]
    with tempfile.TemporaryDirectory() as td:
        for shape, n, seed in specs:
            p = os.path.join(td, "%s_%d.bam" % (shape, seed))
            synth.write(p, n, shape=shape, seed=seed)
            spec = dict(shape=shape, n_reads=n, seed=seed)
            for a in argsets:
                add_case("synth_%s_%d.%s" % (shape, seed, "_".join(x.strip("-") for x in a)), a, synth_spec=spec, tmp_bam=p)
            if shape == "fuzz":
                for reg in ("1:5000-60000", "MT", "2:100000-100500"):
                    add_case("synth_%s_%d.XS.r_%s" % (shape, seed, reg.replace(":", "_")), ["-s", "XS", "-r", reg], synth_spec=spec, tmp_bam=p)
            if shape == "short":
                add_case("synth_%s_%d.XS.r_chr7" % (shape, seed), ["-s", "XS", "-r", "chr7:1000000-80000000"], synth_spec=spec, tmp_bam=p)

        # (v) record-framing stress: decoy record heads inside aux arrays, and a chain that ends in the middle of the file
        import framing_cases
        for variant in ("huge", "to_end", "insane"):
            p = framing_cases.build(variant, os.path.join(td, "framing_%s.bam" % variant))
            add_case("framing_%s.XS" % variant, ["-s", "XS"], framing=variant, tmp_bam=p)
            add_case("framing_%s.RF.a20" % variant, ["-s", "RF", "-a", "20"], framing=variant, tmp_bam=p)
        # (vi) a header of 6,000 contigs (five BGZF members of header before the first record)
        p = framing_cases.build("big_header", os.path.join(td, "framing_big_header.bam"))
        add_case("big_header.XS", ["-s", "XS"], framing="big_header", tmp_bam=p)
        add_case("big_header.XS.r_s1700", ["-s", "XS", "-r", "s1700"], framing="big_header", tmp_bam=p)
        add_case("big_header.XS.r_s5999_1-20000", ["-s", "XS", "-r", "s5999:1-20000"], framing="big_header", tmp_bam=p)
        # (vii) reads of up to 250 kb: records far longer than a BGZF member or a framing segment
        p = framing_cases.build("ultralong", os.path.join(td, "framing_ultralong.bam"))
        add_case("ultralong.XS", ["-s", "XS"], framing="ultralong", tmp_bam=p)
        add_case("ultralong.RF.a0.M100000", ["-s", "RF", "-a", "0", "-M", "100000"], framing="ultralong", tmp_bam=p)
        add_case("ultralong.XS.r_chrL_100000-150000", ["-s", "XS", "-r", "chrL:100000-150000"], framing="ultralong", tmp_bam=p)

    # (viii) region "*" (hts_itr_querys -> HTS_IDX_NOCOOR, hts.c:1903-1904, :1733-1741): the records behind the last reference's reads;
    #        an index whose last reference has no pseudo-bin and no unplaced count gives no iterator (exit 1)
    add_case("contigs.XS.r_star", ["-s", "XS", "-r", "*"], bam="contigs.bam")
    add_case("strand.XS.r_star", ["-s", "XS", "-r", "*"], bam="strand.bam")
    add_case("hcc1395.XS.r_star", ["-s", "XS", "-r", "*"], bam="test_hcc1395.bam")

    # (ix) a contig name of 1,000 characters (the BAM header puts no limit on l_name; Junction::print writes it through a std::string)
    long_name = "contig_" + "N" * 993
    recs = [bamio.record(0, 1000 + 37 * k, "%dM%dN%dM" % (20 + k % 5, 100 + 7 * (k % 3), 25), qname="l%02d" % k, aux=xs_plus) for k in range(12)]
    bamio.write_bam(os.path.join(HERE, "longname.bam"), [(long_name, 500000), ("short", 1000)], recs)
    synth.index(os.path.join(HERE, "longname.bam"))
    add_case("longname.XS", ["-s", "XS"], bam="longname.bam")
    add_case("longname.XS.a30", ["-s", "XS", "-a", "30"], bam="longname.bam")

    # (x) `junctions extract ... <bam> <fasta>`: the intron-motif strand rule with its carried-over state (junctions_extractor.cc:325-359,
    #     :564-584), on the deterministic quartets of tests/cse_synth.py (genes on both strands, reads across several junctions)
    import cse_synth
    with tempfile.TemporaryDirectory() as td:
        for seed, n_genes in ((5, 14), (8, 10)):
            q = cse_synth.build(os.path.join(td, "q%d" % seed), seed=seed, n_genes=n_genes)
            spec = dict(seed=seed, n_genes=n_genes, _fasta=q["fasta"])
            for a in (["-s", "intron-motif"], ["-s", "XS"], ["-s", "RF", "-a", "3"], ["-s", "FR", "-m", "200"]):
                add_case("motif_q%d.%s" % (seed, "_".join(x.strip("-") for x in a)), a, cse=spec, tmp_bam=q["bam"], after=["<fasta>"])

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("%d cases" % len(cases))


if __name__ == "__main__":
    main()
