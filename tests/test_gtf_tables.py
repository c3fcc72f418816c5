"""GtfModel::load (csrc/cse_host.cpp; SURVEY 8a row a12: gtf/gtf_parser.cc:63-263) on the host: its tables against a plain restatement of what
the reference's parser keeps, and the threaded path (text cut into parts, the parts' transcript lists merged by one sort) against the
one-part path -- on GTFs whose transcripts straddle the cuts, repeat far apart, and come in an order that is not the id order."""
import ctypes
import os
import random

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def emu(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
    lib.emu_gtf_dump.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    return lib


def ucsc_bin(start, end):                                           # bedFile.h:339-354 (start, end as the parser passes them)
    offsets = [32678 + 4096 + 512 + 64 + 8 + 1, 4681, 585, 73, 9, 1, 0]    # bedFile.h:49-63: seven levels, upstream's 32678 as it stands
    s, e = start >> 14, (end - 1) >> 14
    for i in range(7):
        if s == e:
            return offsets[i] + s
        s >>= 3
        e >>= 3
    return 0


def attr(col, key):
    for field in col.split(";"):
        if field.startswith(" "):
            field = field[1:]
        parts = field.split(" ")
        if field and parts[0] == key:
            v = parts[1] if len(parts) > 1 else ""
            if len(v) >= 1 and v[0] == '"' and v[-1] == '"':
                v = v[1:-1] if len(v) >= 2 else ""
            return v
    return "NA"


def restate(text):
    tx, order_chroms = {}, []
    for line in text.split("\n")[:-1]:
        if line.startswith("#"):
            continue
        f = line.split("\t")
        if len(f) > 1 and f[-1] == "":                              # std::getline: no empty field behind a trailing tab
            f.pop()
        assert len(f) == 9
        if f[2] != "exon":
            continue
        tid = attr(f[8], "transcript_id")
        if tid == "NA":
            continue
        if tid not in tx:
            if f[0] not in order_chroms:
                order_chroms.append(f[0])
            tx[tid] = dict(chrom=f[0], strand=f[6], gn=attr(f[8], "gene_name"), gi=attr(f[8], "gene_id"), ex=[])
        tx[tid]["ex"].append((int(f[3]), int(f[4])))
    out = ["chrom %d %s" % (i, c) for i, c in enumerate(order_chroms)]
    ids = sorted(tx, key=lambda s: s.encode())
    bins = []
    for t, tid in enumerate(ids):
        x = tx[tid]
        ex = sorted(x["ex"], key=lambda e: e[0], reverse=x["strand"] == "-")       # stable, by start only
        b = ucsc_bin(ex[0][0], ex[-1][1])
        ci = order_chroms.index(x["chrom"])
        out.append("tx %s %s %s %d %s %d" % (tid, x["gn"], x["gi"], ci, x["strand"], b) + "".join(" %d-%d" % e for e in ex))
        bins.append((ci, b, t))
    out += ["bin %d %d %d" % k for k in sorted(bins)]
    out.append("bin_start ok")
    return "\n".join(out) + "\n"


def make_gtf(seed, n_tx, shuffle):
    rnd = random.Random(seed)
    chroms = ["chr%s" % c for c in ("2", "10", "1", "X")]
    lines = ["##description: test", "#!genome-build none"]
    blocks = []
    for t in range(n_tx):
        chrom = chroms[t * len(chroms) // n_tx]
        strand = rnd.choice("+-")
        tid = "T%s%d" % (rnd.choice(["", "x", "A", "~", "RANSCRIPT_WITH_A_LONG_NAME_", "RANSCRIPT_WITH_A"]), rnd.randrange(10 ** rnd.randrange(1, 7)))
        gene = "G%d" % (t // 3)
        pos = rnd.randrange(1000, 200_000_000)
        rows = ["%s\tsrc\ttranscript\t%d\t%d\t.\t%s\t.\tgene_id \"%s\"; transcript_id \"%s\";" % (chrom, pos, pos + 10, strand, gene, tid)]
        for e in range(rnd.randrange(1, 9)):
            s = pos + rnd.randrange(0, 50000)
            name = "N%d_%d" % (t, e)                            # gene_name differs per exon line: the FIRST exon line's wins
            style = rnd.randrange(6)                                # how the id is written changes from line to line; the id does not
            tcol = ['transcript_id "%s";' % tid, "transcript_id %s;" % tid, 'transcript_id "%s" ;' % tid, 'transcript_id "%s"' % tid,
                    'transcript_id "%s"; ' % tid, 'transcript_id "%s";' % tid][style]
            rest = "" if style == 3 else ' gene_name "%s"; exon_number %d;' % (name, e)
            front = "" if style == 5 else 'gene_id "%s"; ' % gene
            rows.append("%s\tsrc\texon\t%d\t%d\t.\t%s\t.\t%s%s%s%s"
                        % (chrom, s, s + rnd.randrange(1, 3000), strand, front, tcol, rest, "\t" if rnd.random() < 0.05 else ""))
        if rnd.random() < 0.1:
            rows.append("%s\tsrc\tCDS\t%d\t%d\t.\t%s\t0\tgene_id \"%s\";" % (chrom, pos, pos + 5, strand, gene))
        blocks.append(rows)
    if shuffle:                                                     # exon lines of one transcript far apart (and duplicate ids merge)
        flat = [r for b in blocks for r in b]
        by_chrom = {}
        for r in flat:
            by_chrom.setdefault(r.split("\t")[0], []).append(r)
        flat = []
        for c in chroms:
            rows = by_chrom.get(c, [])
            rnd.shuffle(rows)
            flat += rows
        lines += flat
    else:
        for b in blocks:
            lines += b
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed,n_tx,shuffle", [(1, 300, False), (2, 300, True), (3, 2500, False), (4, 40, True), (5, 6000, True), (6, 40000, False)])   # (the last two: the merge and the id sort in several ranges)
def test_tables_equal_restatement_for_every_part_count(emu, tmp_path, seed, n_tx, shuffle):
    text = make_gtf(seed, n_tx, shuffle)
    # (duplicate transcript ids on two contigs would make 'first line wins' depend on nothing else: keep them, they are legal input)
    gtf, out = str(tmp_path / "t.gtf"), str(tmp_path / "t.dump")
    open(gtf, "w").write(text)
    want = restate(text)
    err = ctypes.create_string_buffer(256)
    for parts in (1, 2, 7, 32):
        os.environ["REGTOOLS_AMD_GTF_PARTS"] = str(parts)
        try:
            assert emu.emu_gtf_dump(gtf.encode(), out.encode(), err, 256) == 0, err.value
        finally:
            del os.environ["REGTOOLS_AMD_GTF_PARTS"]
        assert open(out).read() == want, parts


BAD = ["chr1\tonly\tthree", "chr1\ts\texon\t1\t2\t.\t+\t.\tgene_id \"g\";\textra", "chr1\ts\texon\t1\t2\t.\t+\t.\t", "chr1\ts\texon\t1\t2\t.\t+\t.",
       "chr1\ts\texon\t1\t2\t.\t+\t.\t\tx", "nothing", "\t"]


@pytest.mark.parametrize("bad", range(len(BAD)))
def test_first_bad_line_in_file_order_ends_the_run(emu, tmp_path, bad):
    text = make_gtf(5, 200, False).split("\n")
    text.insert(150, BAD[bad])
    text.insert(400, "")
    gtf = str(tmp_path / "bad.gtf")
    open(gtf, "w").write("\n".join(text))
    err = ctypes.create_string_buffer(256)
    for parts in (1, 5):
        os.environ["REGTOOLS_AMD_GTF_PARTS"] = str(parts)
        try:
            assert emu.emu_gtf_dump(gtf.encode(), str(tmp_path / "x").encode(), err, 256) == 1
        finally:
            del os.environ["REGTOOLS_AMD_GTF_PARTS"]
        assert err.value == b"Expected 9 fields in GTF line."


def test_nine_fields_with_an_empty_attribute_column_is_a_line(emu, tmp_path):
    text = make_gtf(6, 50, False).split("\n")
    text.insert(20, "chr1\ts\texon\t1\t2\t.\t+\t.\t\t")             # eight fields, an empty ninth, a trailing tab: no transcript_id, skipped
    gtf, out = str(tmp_path / "t.gtf"), str(tmp_path / "t.dump")
    open(gtf, "w").write("\n".join(text))
    err = ctypes.create_string_buffer(256)
    assert emu.emu_gtf_dump(gtf.encode(), out.encode(), err, 256) == 0, err.value
    del text[20]
    assert open(out).read() == restate("\n".join(text))
