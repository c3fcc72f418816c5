"""The known answers the reference's own GtfParser unit test holds (tests/lib/gtf/test_gtf_parser.cc), by name, on the product's GTF
loader (GtfModel::load, csrc/cse_host.cpp: on the host through tests/hostemu, on the GPU box through rgx_gtf_load):

  ParseExonLineTest / AddExonToTranscriptTest (:44-119)  the EP300 exon line: gene_name EP300, gene_id ENSG00000100393, bin 37359, the bin's list
  ParseAttributeTest (:74-83)                            tss_id / ccds_id / gene_source values; "NA" for a key that is not there
  SortExonTranscriptPsTest (:122-199)                    exons 10100, 9900, 9700 of a '+' transcript come out ascending
  SortExonTranscriptNsTest (:202-279)                    exons 9900, 9700, 10100 of a '-' transcript come out descending

The sort tests' transcripts are also annotated against: tests/golden/gtest_gtf/ holds the gtest's lines as two GTFs, BED12 rows around their introns and what the REAL
reference's `junctions annotate` makes of them (make_golden_gtest_gtf.py) -- which junction is known and which exon skipped depends on the exons' order -- for the
oracle here and the product on the GPU box.

The vectors are the gtest's data (its line text, coordinates, expected values), not its code."""
import ctypes as C
import os

import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "gtest_gtf")
ANNOTATE = [(n, f) for n in ("ps", "ns") for f in ([], ["-S"])]

# the attribute column of test_gtf_parser.cc:45-52 with the fields the three sort tests vary
def column(ccds, exon_id, exon_number):
    return ('ccds_id "%s"; exon_id "%s"; exon_number "%s"; gene_biotype "protein_coding"; gene_id "ENSG00000100393"; gene_name "EP300"; '
            'gene_source "ensembl_havana"; p_id "P5137"; tag "CCDS"; transcript_id "ENST00000263253"; transcript_name "EP300-001"; '
            'transcript_source "ensembl_havana"; tss_id "TSS138009"' % (ccds, exon_id, exon_number))


def line(start, end, strand, col):
    return "22\tprotein_coding\texon\t%d\t%d\t.\t%s\t.\t%s\n" % (start, end, strand, col)


EP300 = line(12791, 14103, "+", column("CCDS14010", "ENSE00001343011", "1"))                                   # :45-52, :87-103
SORT_PS = (line(10100, 10200, "+", column("CCDS14010", "ENSE00001343011", "3")) + line(9900, 10000, "+", column("CCDS14011", "ENSE00001343012", "2"))
           + line(9700, 9800, "+", column("CCDS14012", "ENSE00001343013", "1")))                               # :123-172, added in this order :192-194
SORT_NS = (line(9900, 10000, "-", column("CCDS14011", "ENSE00001343012", "2")) + line(9700, 9800, "-", column("CCDS14012", "ENSE00001343013", "1"))
           + line(10100, 10200, "-", column("CCDS14010", "ENSE00001343011", "3")))                             # :203-252, added in this order :272-274


@pytest.fixture(scope="module")
def emu(built):
    lib = C.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
    lib.emu_gtf_dump.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    return lib


def dump(emu, tmp_path, text, parts=None):
    gtf, out = str(tmp_path / "k.gtf"), str(tmp_path / "k.txt")
    open(gtf, "w").write(text)
    err = C.create_string_buffer(256)
    if parts:
        os.environ["REGTOOLS_AMD_GTF_PARTS"] = str(parts)
    try:
        assert emu.emu_gtf_dump(gtf.encode(), out.encode(), err, 256) == 0, err.value
    finally:
        os.environ.pop("REGTOOLS_AMD_GTF_PARTS", None)
    return open(out).read().split("\n")[:-1]


def test_add_exon_to_transcript(emu, tmp_path):
    rows = dump(emu, tmp_path, EP300)
    # get_gene_from_transcript -> {EP300, ENSG00000100393} (:104-107); bin_from_transcript -> 37359 (:113-114); transcripts_from_bin("22", 37359) -> that one (:115-118)
    assert rows == ["chrom 0 22", "tx ENST00000263253 EP300 ENSG00000100393 0 + 37359 12791-14103", "bin 0 37359 0", "bin_start ok"]


def test_parse_attribute(emu, tmp_path):
    # parse_attribute (:74-83): the value between the quotes of the named field, "NA" when the column has no such field -- seen through the two
    # attributes the tables keep: a column that names neither gene gives NA, NA
    rows = dump(emu, tmp_path, line(12791, 14103, "+", 'ccds_id "CCDS14010"; gene_source "ensembl_havana"; transcript_id "T1"; tss_id "TSS138009"'))
    assert rows[1] == "tx T1 NA NA 0 + 37359 12791-14103"
    # ... and one whose gene fields sit where tss_id / ccds_id sit in the gtest's vector (last field without a closing ';', first field)
    rows = dump(emu, tmp_path, line(12791, 14103, "+", 'gene_id "CCDS14010"; gene_source "ensembl_havana"; transcript_id "T1"; gene_name "TSS138009"'))
    assert rows[1] == "tx T1 TSS138009 CCDS14010 0 + 37359 12791-14103"


@pytest.mark.parametrize("parts", [None, 3])
def test_sort_exons_within_transcripts(emu, tmp_path, parts):
    # '+': ascending start (:187-190, :198); '-': descending start (:267-270, :278); the threaded loader (a cut between the lines) files them the same way
    rows = dump(emu, tmp_path, SORT_PS, parts)
    assert rows[1].split(" ")[7:] == ["9700-9800", "9900-10000", "10100-10200"] and rows[1].split(" ")[5] == "+"
    rows = dump(emu, tmp_path, SORT_NS, parts)
    assert rows[1].split(" ")[7:] == ["10100-10200", "9900-10000", "9700-9800"] and rows[1].split(" ")[5] == "-"


@pytest.mark.gpu
def test_product_loads_the_gtest_lines(gpu_ctx, tmp_path):
    from regtools_amd import _ffi
    L = _ffi.lib()
    L.rgx_gtf_transcript_id.restype = C.c_char_p
    for text, n_exons in ((EP300, 1), (SORT_PS, 3), (SORT_NS, 3)):
        gtf = str(tmp_path / "k.gtf")
        open(gtf, "w").write(text)
        g = C.c_void_p()
        err = C.create_string_buffer(256)
        assert L.rgx_gtf_load(gpu_ctx._h, gtf.encode(), C.byref(g), err, 256) == 0, err.value
        n_tx, n_ex, n_ch = C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.rgx_gtf_info(g, C.byref(n_tx), C.byref(n_ex), C.byref(n_ch))
        assert (n_tx.value, n_ex.value, n_ch.value) == (1, n_exons, 1)
        assert L.rgx_gtf_transcript_id(g, 0) == b"ENST00000263253"
        b = C.c_uint32()
        assert L.rgx_gtf_transcript_bin(g, b"ENST00000263253", C.byref(b)) == 0
        # (exon 12791-14103 -> 37359, :113-114; the sort tests' transcripts span 9700-10200 of the same 16 kb window)
        assert b.value == 37359
        assert L.rgx_gtf_transcript_bin(g, b"ENSTfake", C.byref(b)) != 0                                       # (:108-111: nothing is known of it)
        L.rgx_gtf_free(g)


def test_the_committed_gtfs_are_the_gtest_lines():
    assert open(os.path.join(GOLD, "ps.gtf")).read() == SORT_PS and open(os.path.join(GOLD, "ns.gtf")).read() == SORT_NS


@pytest.mark.parametrize("name,flag", ANNOTATE, ids=["%s%s" % (n, "_S" if f else "") for n, f in ANNOTATE])
def test_oracle_annotates_against_the_sorted_exons_as_the_reference(name, flag, tmp_path, oracle_cli):
    import subprocess
    out = str(tmp_path / "o.out")
    r = subprocess.run([oracle_cli, "junctions-annotate"] + flag + ["-o", out, os.path.join(GOLD, name + ".bed"), os.path.join(GOLD, "genome.fa"), os.path.join(GOLD, name + ".gtf")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    got = open(out, "rb").read()
    assert got == open(os.path.join(GOLD, "%s%s.out" % (name, "_S" if flag else "")), "rb").read()
    assert {row.split(b"\t")[10] for row in got.split(b"\n")[1:-1]} == {b"A", b"D", b"DA", b"NDA", b"N"}     # (every anchor class is among the rows)


@pytest.mark.gpu
@pytest.mark.parametrize("name,flag", ANNOTATE, ids=["%s%s" % (n, "_S" if f else "") for n, f in ANNOTATE])
def test_product_annotates_against_the_sorted_exons_as_the_reference(gpu_ctx, name, flag, tmp_path):
    import regtools_amd
    out = str(tmp_path / "p.out")
    ja = regtools_amd.JunctionsAnnotator(ctx=gpu_ctx)
    ja.parse_options(flag + ["-o", out, os.path.join(GOLD, name + ".bed"), os.path.join(GOLD, "genome.fa"), os.path.join(GOLD, name + ".gtf")])
    ja.annotate()
    assert open(out, "rb").read() == open(os.path.join(GOLD, "%s%s.out" % (name, "_S" if flag else "")), "rb").read()
