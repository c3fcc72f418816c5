"""Inputs for the annotated-VCF writer tests: VCF text (hand-made edge cases and seeded random records), gzip, BCF (tests/bcfio.py).
`build()` returns {name: bytes}.  The expected outputs under tests/golden/vcf_writer/ are the REAL reference's
(tests/golden/make_golden_vcf.py); this module is data, not product code."""
import gzip
import random

import bcfio
from bcfio import CHAR, FLOAT, FLOAT_END, FLOAT_MISSING, I8_END, I8_MISSING, I32_MISSING, INT8, INT16, INT32, record

HDR = '''##fileformat=VCFv4.2
##FILTER=<ID=q10,Description="Quality below 10">
##INFO=<ID=DP,Number=1,Type=Integer,Description="Total Depth">
##INFO=<ID=AF,Number=A,Type=Float,Description="Allele Frequency">
##INFO=<ID=DB,Number=0,Type=Flag,Description="dbSNP">
##INFO=<ID=STR,Number=1,Type=String,Description="a string">
##INFO=<ID=MQ,Number=.,Type=Float,Description="q">
##INFO=<ID=AC,Number=A,Type=Integer,Description="ac">
##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth">
##FORMAT=<ID=GQ,Number=1,Type=Float,Description="q">
##FORMAT=<ID=PL,Number=G,Type=Integer,Description="pl">
##FORMAT=<ID=FT,Number=1,Type=String,Description="ft">
##contig=<ID=1,length=249250621>
##contig=<ID=2>
##source=foo
'''
S0 = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
S2 = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\n"

# a transcript on contig 1 whose exons put positions 90..130 into splice regions; FAR has nothing near any variant
GTF_NEAR = ('1\tx\texon\t80\t110\t.\t+\t.\tgene_id "g1"; transcript_id "t1"; gene_name "n1";\n'
            '1\tx\texon\t300\t400\t.\t+\t.\tgene_id "g1"; transcript_id "t1"; gene_name "n1";\n'
            '1\tx\texon\t60\t104\t.\t-\t.\tgene_id "g2"; transcript_id "t2"; gene_name "n2";\n'
            '1\tx\texon\t10\t20\t.\t-\t.\tgene_id "g2"; transcript_id "t2"; gene_name "n2";\n')
GTF_FAR = 'zz\tx\texon\t100\t200\t.\t+\t.\tgene_id "g"; transcript_id "t"; gene_name "n";\n'


def _text_cases():
    c = {}
    c["floats"] = HDR + S0 + "\n".join([
        "1\t100\trs1\tA\tG\t30.00\tPASS\tDP=10;AF=0.500;DB",
        "1\t101\t.\tA\tG,T\t1e3\tq10\tAF=1.234e-01,0.3333333333;MQ=59.97,60",
        "1\t102\t.\tA\t.\t.\t.\t.",
        "1\t0103\t.\tAC\tA\t0.000012345\tPASS;\tDP=+5;AC=007,-3;STR=hello world;",
        "2\t104\ta;b\tN\t<DEL>\t29.5\tq10;PASS\tDP=abc;AF=.;MQ=nan,inf,-inf,1e-50",
        "3\t105\t.\tA\tC\t5\tnewfilt\tNEWKEY=1,2;NEWFLAG;DP=3;;AF=0.1",
        "1\t106\t.\tA\tC\t5\tPASS\tDP=2147483648;AC=127,128,-127,-128,32767,32768,-32768,-32767",
        "1\t107\t.\tA\tC\t5\tPASS\tDP=;STR=;AF=;DB=x",
        "1\t108\t.\tA\tC\t5\tPASS\tgenes=old;DP=1",
        "1\t109\t.\tA\tC\t1234567.89\tPASS\tAF=123456789,0.1234567,100000,1000000,0.0001,0.00001",
        "1\t110\t.\tA\tC\t5\tPASS\tDP=5x,6;AC=1e5,7;AC=-2147483648;DP=-2147483647",
    ]) + "\n"
    c["samples"] = HDR + S2 + "\n".join([
        "1\t100\t.\tA\tG\t30\tPASS\tDP=10\tGT:DP:GQ:PL\t0/1:10:99.5:0,10,100\t1|1:.:.:.",
        "1\t101\t.\tA\tG\t30\tPASS\tDP=10\tGT:DP:GQ\t0/1\t./.:5",
        "1\t102\t.\tA\tG,T\t30\tPASS\tDP=10\tGT:PL:FT\t1/2:1,2,3,4,5,6:PASS\t0:7,8:longer_string",
        "1\t103\t.\tA\tG\t30\tPASS\tDP=10\tGT:NEWF:DP\t0|1:abc:300\t.:.:40000",
        "1\t104\t.\tA\tG\t30\tPASS\tDP=10\tDP:GQ\t1:0.123456789\t2:1e-3",
        "1\t105\t.\tA\tG\t30\tPASS\tDP=10",                                                  # no sample columns: bcf_write refuses it
        "1\t106\t.\tA\tG\t30\tPASS\tDP=10\tGT\t0/0\t0/1\textra",
        "1\t107\t.\tA\tG\t30\tPASS\tDP=10\tFT:GT\tabc:0/1\tdefgh:1/1",                       # upstream's width count is one short here: 'h' lands in A's genotype
    ]) + "\n"
    c["stop_samples"] = HDR + S2 + "\n".join([
        "1\t100\t.\tA\tG\t30\tPASS\tDP=10\tGT\t0/1\t1/1",
        "1\t101\t.\tA\tG\t30\tPASS\tDP=10\tGT\t0/1",                                          # one sample short: the read loop ends here
        "1\t102\t.\tA\tG\t30\tPASS\tDP=10\tGT\t0/1\t1/1",
    ]) + "\n"
    c["hdr_dups"] = ('''##fileformat=VCFv4.1
##source=a
##source=a
##source=b
##INFO=<ID=DP,Number=1,Type=Integer,Description="d1">
##INFO=<ID=DP,Number=1,Type=Float,Description="d2">
##FORMAT=<ID=DP,Number=1,Type=Integer,Description="fmt dp">
##FILTER=<ID=PASS,Description="custom pass">
##contig=<ID=1,length=abc>
##contig=<ID=2,length=5,assembly=b37>
##contig=<ID=2,length=6>
##contig=<length=6>
##ALT=<ID=DEL,Description="Deletion">
##PEDIGREE=<Name_0=G0-ID,Name_1=G1-ID>
##INFO=<ID=genes,Number=1,Type=String,Description="prior">
##INFO=<ID=X,Number=1,Type=String,Description="quoted \\"inner\\" text, with comma",Source="s",Version="3">
##INFO=<ID=Y,Number=1,Type=Whatever,Description=unquoted>
##fileformat=VCFv4.2
##weird=<a=b>   
''') + S0 + "1\t5\t.\tA\tC\t.\t.\tDP=3.7;X=q;Y=2;genes=zzz\n2\t6\t.\tA\tC\t.\tPASS\t.\n"
    c["crlf"] = (HDR + S0).replace("\n", "\r\n") + "1\t100\t.\tA\tG\t30\tPASS\tDP=10\r\n1\t101\t.\tA\tG\t30\tPASS\tDP=11\r\n"
    c["badline"] = HDR + "##thisisbad\n##INFO=<ID=ZZ,Number=1,Type=Integer,Description=\"after\">\n" + S0 + "1\t100\t.\tA\tG\t30\tPASS\tZZ=10\n"
    c["badstruct"] = HDR + "##INFO=<ID=ZZ,9Number=1>\n##INFO=<ID=QQ,Number=1,Type=Integer,Description=\"after\">\n" + S0 + "1\t100\t.\tA\tG\t30\tPASS\tQQ=10\n"
    c["emptyhdrlines"] = "##fileformat=VCFv4.2\n\n##source=x\n\n" + S0 + "1\t100\t.\tA\tG\t30\tPASS\t.\n"
    c["shortcols"] = HDR + S0 + "1\t100\t.\tA\tG\t30\tPASS\n1\t101\t.\tA\tG\t30\n"
    return c


def _random_vcf(seed, n_rec=25):
    rng = random.Random(seed)

    def num_int():
        c = rng.random()
        if c < 0.1: return "."
        if c < 0.2: return rng.choice(["+", "-", "0", "00"]) + str(rng.randrange(1000))
        if c < 0.3: return str(rng.choice([127, 128, -127, -128, -126, 32767, 32768, -32767, -32768, -32766, 2147483647, -2147483648, -2147483647, 2147483648, 99999999999]))
        return str(rng.randrange(-50, 70000))

    def num_float():
        c = rng.random()
        if c < 0.1: return "."
        if c < 0.2: return rng.choice(["nan", "inf", "-inf", "NaN", "1e400", "1e-400", ".5", "5.", "-.5e1", "0x1p3"])
        if c < 0.5: return "%.*e" % (rng.randrange(0, 12), rng.uniform(-1, 1) * 10 ** rng.randrange(-8, 9))
        if c < 0.8: return "%.*f" % (rng.randrange(0, 10), rng.uniform(-1000, 1000))
        return repr(rng.uniform(0, 1))

    def strv(): return "".join(rng.choice("abcXYZ019_.|/-+%") for _ in range(rng.randrange(1, 8)))
    def vec(f): return ",".join(f() for _ in range(rng.choice([1, 1, 1, 2, 3, 5, 17])))

    def info():
        if rng.random() < 0.05: return "."
        items = []
        for _ in range(rng.randrange(1, 7)):
            k = rng.choice(["DP", "AF", "DB", "STR", "MQ", "AC", "UNDEF1", "UNDEF2"])
            c = rng.random()
            if k == "DB" or (k.startswith("UNDEF") and c < 0.3): items.append(k)
            elif k in ("DP", "AC"): items.append(k + "=" + vec(num_int))
            elif k in ("AF", "MQ"): items.append(k + "=" + vec(num_float))
            else: items.append(k + "=" + strv())
        s = ";".join(items)
        return s + ";" if rng.random() < 0.1 else s

    def gt():
        s = ""
        for i in range(rng.choice([1, 2, 2, 2, 3])):
            if i: s += rng.choice("/|")
            s += rng.choice([".", "0", "1", "2", "10"])
        return s

    def sample(keys, full):
        out = []
        for k in keys:
            if k == "GT": out.append(gt())
            elif k == "DP": out.append(num_int())
            elif k == "PL": out.append(vec(num_int))
            elif k == "GQ": out.append(vec(num_float) if rng.random() < 0.3 else num_float())
            else: out.append(strv().replace(":", "_"))
        if not full and len(out) > 1: out = out[:rng.randrange(1, len(out) + 1)]
        return ":".join(out)

    ns = rng.choice([0, 1, 2, 3])
    lines = []
    for _ in range(n_rec):
        f = [rng.choice(["1", "1", "2", "3", "chrUn"]), str(rng.randrange(60, 140)), rng.choice([".", "rs%d" % rng.randrange(99), "a;b"]), rng.choice(["A", "AC", "N"]),
             rng.choice([".", "G", "G,T", "<DEL>", "G,<INS>,*"]), rng.choice([".", num_float(), "30", "29.5", "1e2"]).replace("nan", "5").replace("NaN", "5"),
             rng.choice([".", "PASS", "q10", "q10;PASS", "undefF", "PASS;"]), info()]
        if ns and rng.random() < 0.95:
            keys = ["GT"] + rng.sample(["DP", "GQ", "PL", "FT", "UNDEFF"], rng.randrange(0, 5))
            if rng.random() < 0.1: keys = keys[1:] or ["DP"]
            f.append(":".join(keys))
            for _s in range(ns): f.append(sample(keys, rng.random() < 0.7))
        lines.append("\t".join(f))
    head = HDR + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO" + ("\tFORMAT" + "".join("\tS%d" % i for i in range(ns)) if ns else "") + "\n"
    return head + "\n".join(lines) + "\n"


BCF_HDR = '''##fileformat=VCFv4.2
##FILTER=<ID=PASS,Description="All filters passed">
##FILTER=<ID=q10,Description="Quality below 10">
##INFO=<ID=DP,Number=1,Type=Integer,Description="Total Depth">
##INFO=<ID=AF,Number=A,Type=Float,Description="Allele Frequency">
##INFO=<ID=DB,Number=0,Type=Flag,Description="dbSNP">
##INFO=<ID=STR,Number=1,Type=String,Description="a string">
##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth">
##FORMAT=<ID=GQ,Number=1,Type=Float,Description="q">
##FORMAT=<ID=FT,Number=1,Type=String,Description="ft">
##contig=<ID=1,length=249250621>
##contig=<ID=2>
#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB
'''


def _bcf_records():
    # dictionary ids in order of appearance: PASS 0, q10 1, DP 2, AF 3, DB 4, STR 5, GT 6 (FORMAT DP = 2), GQ 7, FT 8
    return [
        record(0, 107, b"rs1", [b"A", b"G"], 30.0, [0], [(2, INT8, [10]), (3, FLOAT, [0.5]), (4, INT8, None)], [(6, INT8, 2, [2, 4, 4, 5]), (2, INT8, 1, [10, I8_MISSING])], 2),
        record(1, 100, b"", [b"A", b"G", b"T"], None, [], [(3, FLOAT, [0.1234567, ('bits', FLOAT_MISSING)]), (5, CHAR, b"hello"), (2, INT16, [300])],
               [(6, INT8, 2, [2, I8_END, 0, 3]), (7, FLOAT, 2, [99.5, ('bits', FLOAT_END), ('bits', FLOAT_MISSING), 1.0]), (8, CHAR, 4, b"PASSab\0\0")], 2),
        record(0, 109, b"x", [b"AC"], 1e-5, [1, 0], [(2, INT32, [100000, I32_MISSING, 7]), (5, CHAR, b"q"), (2, INT8, [I8_MISSING])], [], 2),
        record(0, 110, b"y", [b"A", b"C"], 3.0, [0], [(2, INT8, list(range(20)))], [(2, INT16, 1, [1000, 2000])], 2),
        record(0, 111, b"z", [b"A", b"C"], 3.0, [0], [], [], 0),           # no samples in a file that has two: not written
    ]


def _line_cases():
    """Lines behind the header that are no ordinary records: upstream's read loop parses every one of them (vcf_read, vcf.c:1958-1964) -- a
    blank line, a '#' line, text without a tab, a record cut short behind POS; the ID such a record prints depends on whether an earlier
    record of the file had an ID column (round 3)."""
    r = ["1\t100\t.\tA\tG\t30\tPASS\tDP=10", "1\t1005\t.\tC\tT\t40\tPASS\tDP=11", "1\t2100\trs1\tG\tA\t50\tPASS\tDP=12"]
    h = HDR + S0
    return {
        "lines_blank_mid": h + r[0] + "\n\n" + r[1] + "\n" + r[2] + "\n",
        "lines_blank_end": h + "\n".join(r) + "\n\n",
        "lines_blank_first": h + "\n\n" + r[0] + "\n",
        "lines_hash_mid": h + r[0] + "\n#comment line\n" + r[1] + "\n##INFO=<ID=XX,Number=1,Type=Integer,Description=\"x\">\n" + r[2] + "\n",
        "lines_header_again": h + r[0] + "\n" + S0 + r[1] + "\n",
        "lines_notab": h + r[0] + "\nhello world\n   \n" + r[1] + "\n",
        "lines_short": h + "1\t50\n" + r[0] + "\n1\t95\n" + r[2] + "\n1\t70\n\n",
    }


def _tbi(names, cut_body=False, bins=None):
    """a tabix index that lists these sequence names and no bins (BGZF-compressed, as tabix writes it).  cut_body: the file ends behind the
    names (hts_idx_load_core fails on the first n_bin it cannot read: no index); bins: per sequence, a list of bin numbers, one chunk each
    (a number listed twice makes the load fail: "duplicate bin")."""
    import struct
    import bamio
    nm = b"".join(n.encode() + b"\0" for n in names)
    body = b"TBI\1" + struct.pack("<iiiiiii", len(names), 2, 1, 2, 0, ord("#"), 0) + struct.pack("<i", len(nm)) + nm
    for k, _ in enumerate(names):
        if cut_body:
            break
        bl = bins[k] if bins else []
        body += struct.pack("<i", len(bl))
        for b in bl:
            body += struct.pack("<Ii", b, 1) + struct.pack("<QQ", 0, 1 << 16)
        body += struct.pack("<i", 0)
    return bamio.bgzf_member(body) + bamio.EOF_MARKER


_TBI_VCF = HDR + S0 + "1\t100\t.\tA\tG\t30\tPASS\tDP=10\n7\t1005\t.\tC\tT\t40\tPASS\tDP=11\n8\t5\t.\tC\tT\t40\tPASS\tDP=11\n"


def companions():
    """{case name: {file suffix replacing / following ".vcf": bytes}}: files the test harness writes next to NAME.vcf.  A .tbi's sequence names
    that the header does not declare join the header (vcf_hdr_read, vcf.c:1289-1309); the index is looked for as NAME.vcf.tbi, then NAME.tbi;
    one that cannot be read is no index."""
    return {"tbi_contigs": {".vcf.tbi": _tbi(["1", "7", "chrUn", "2"])},
            "tbi_stem": {".tbi": _tbi(["zz", "7"])},
            "tbi_broken": {".vcf.tbi": b"TBI\1 this is not a tabix index"},
            # an index whose body cannot be loaded is no index at all, names or not (tbx_index_load -> hts_idx_load_core, hts.c:1517-1567)
            "tbi_cut_body": {".vcf.tbi": _tbi(["1", "7", "chrUn"], cut_body=True)},
            "tbi_dup_bin": {".vcf.tbi": _tbi(["1", "chrUn"], bins=[[4681, 4681], []])},
            "tbi_with_bins": {".vcf.tbi": _tbi(["1", "chrUn"], bins=[[4681, 4682], [585]])}}


def build(tmpdir):
    """{name: bytes}; tmpdir is scratch for the BCF writer"""
    import os
    out = {k: v.encode() for k, v in _text_cases().items()}
    out.update({k: v.encode() for k, v in _line_cases().items()})
    for k in companions():
        out[k] = _TBI_VCF.encode()
    for seed in range(1, 9):
        out["random%d" % seed] = _random_vcf(seed).encode()
    out["floats_gz"] = gzip.compress(out["floats"], mtime=0)
    p = os.path.join(tmpdir, "x.bcf")
    bcfio.write_bcf(p, BCF_HDR, _bcf_records()); out["typed_bcf"] = open(p, "rb").read()
    bcfio.write_bcf(p, BCF_HDR, _bcf_records(), compress=False); out["typed_bcf_raw"] = open(p, "rb").read()
    return out


def simple_vcf_to_bcf(vcf_text, path):
    """BCF form of a VCF whose records carry at most `DP=<int>` in INFO (tests/cse_synth.py writes those): header text as is, typed
    records by hand.  -> path"""
    lines = vcf_text.splitlines()
    hdr = [l for l in lines if l.startswith("#")]
    contigs = [l.split("ID=")[1].split(",")[0].rstrip(">") for l in hdr if l.startswith("##contig")]
    ids = ["PASS"] + [l.split("ID=")[1].split(",")[0] for l in hdr if l.startswith(("##INFO", "##FILTER", "##FORMAT")) and "ID=PASS" not in l]
    recs = []
    for l in lines:
        if not l or l.startswith("#"):
            continue
        f = l.split("\t")
        info = []
        if f[7] != ".":
            k, v = f[7].split("=")
            v = int(v)
            info.append((ids.index(k), INT8 if -120 <= v <= 127 else INT16, [v]))
        recs.append(record(contigs.index(f[0]), int(f[1]) - 1, b"" if f[2] == "." else f[2].encode(), [f[3].encode()] + ([] if f[4] == "." else [a.encode() for a in f[4].split(",")]),
                           None if f[5] == "." else float(f[5]), [] if f[6] == "." else [ids.index(x) for x in f[6].split(";")], info))
    bcfio.write_bcf(path, "\n".join(hdr) + "\n", recs)
    return path
