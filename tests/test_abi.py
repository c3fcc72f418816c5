"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/regtools_amd.h
declares, refuses to run without a GPU (no CPU fallback), and its host-only entry points (row packing, shard
merge, BED12 formatting) behave.  No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    from regtools_amd import _ffi
    header = open(os.path.join(ROOT, "include", "regtools_amd.h")).read()
    declared = set(re.findall(r"\b(rgx_[a-z0-9_]+)\s*\(", header))
    declared -= {"rgx_ctx", "rgx_member"}
    L = _ffi.lib()
    for sym in sorted(declared):
        assert hasattr(L, sym), "libregtools_amd.so does not export %s" % sym
    assert declared == set(_ffi.EXPORTS), (declared ^ set(_ffi.EXPORTS))
    assert b"gfx950" in L.rgx_version()


def test_default_params_match_reference_defaults(built):
    from regtools_amd import _ffi
    p = _ffi.ExtractParams()
    _ffi.lib().rgx_extract_params_default(C.byref(p))
    # JunctionsExtractor default ctor, junctions_extractor.h:185-198
    assert (p.region, p.strandness, bytes(p.strand_tag), p.min_anchor, p.min_intron, p.max_intron, p.fasta_path) == (b".", -1, b"XS", 8, 70, 500000, None)


def test_no_gpu_means_loud_failure_not_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    import regtools_amd
    with pytest.raises(regtools_amd.RegtoolsError) as e:
        regtools_amd.Context(0)
    assert e.value.code == 4 and "no CPU fallback" in str(e.value)
    # the CLI maps it to the reference's exit code for runtime errors (junctions_main.cc:51-57)
    import subprocess
    r = subprocess.run([os.path.join(ROOT, "bin", "regtools-amd"), "junctions", "extract", "-s", "XS",
                        os.path.join(ROOT, "tests", "golden", "strand.bam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stdout == b""


def test_cli_option_errors_and_help(built):
    import subprocess
    exe = os.path.join(ROOT, "bin", "regtools-amd")
    run = lambda *a: subprocess.run([exe] + list(a), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run("junctions", "extract", "-h").returncode == 0                        # test_help
    assert run("junctions", "extract", "-s", "XS").returncode == 1                   # test_no_bam
    assert run("junctions", "extract", "x.bam").returncode == 1                      # test_no_strandness
    assert run("junctions", "extract", "-s", "bogus", "x.bam").returncode == 1
    assert run("junctions", "extract", "-s", "intron-motif", "x.bam").returncode == 1  # needs a FASTA
    assert run().returncode == 0 and run("junctions").returncode == 0


def test_python_option_parser_mirrors_reference(built):
    import regtools_amd
    je = regtools_amd.JunctionsExtractor()
    je.parse_options(["-a", "12", "-m", "100", "-M", "900", "-r", "1:5-9", "-t", "ts", "-s", "FR", "-o", "o.bed", "in.bam"])
    assert (je.min_anchor_length_, je.min_intron_length_, je.max_intron_length_, je.region_, je.strand_tag_, je.strandness_, je.get_bam()) == (12, 100, 900, "1:5-9", "ts", 2, "in.bam")
    for bad in (["-s", "XS"], ["in.bam"], ["-s", "nope", "in.bam"], ["-s", "intron-motif", "in.bam"], ["-s", "XS", "a.bam", "b.fa", "c"]):
        with pytest.raises(regtools_amd.RegtoolsError):
            regtools_amd.JunctionsExtractor().parse_options(bad)


def _table_from_rows(rows, names=("chrA", "chrB")):
    """rows: (tid,start,end,ts,te,count,first,last,strand) -> JunctionTable* through rgx_table_unpack."""
    import struct
    from regtools_amd import _ffi
    L = _ffi.lib()
    # a names-only table: unpack needs a contig table to copy
    raw = b"".join(struct.pack("<12I", r[0] & 0xffffffff, r[1], r[2], r[3], r[4], r[5], r[6] & 0xffffffff, r[6] >> 32, r[7] & 0xffffffff, r[7] >> 32, ord(r[8]), 0) for r in rows)
    proto = _ffi.JunctionTable()
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    lens = (C.c_uint32 * len(names))(*([1000000] * len(names)))
    proto.n_ref, proto.ref_name, proto.ref_len = len(names), arr, lens
    t = C.POINTER(_ffi.JunctionTable)()
    buf = (C.c_uint8 * max(1, len(raw))).from_buffer_copy(raw or b"\0")
    assert L.rgx_table_unpack(buf, len(rows), C.byref(proto), C.byref(t)) == 0
    return t


def test_pack_unpack_roundtrip_and_merge_semantics(built):
    from regtools_amd import _ffi, distributed
    L = _ffi.lib()
    # shard 0 then shard 1 (file order). Same key in both shards: counts add, thick bounds widen, the name comes
    # from the earliest first_seen, the strand from the latest last_seen (class 2: '?' then '.').
    s0 = [(0, 100, 200, 90, 230, 3, 0, 7, "+"), (0, 300, 400, 280, 410, 1, 5, 5, "?"), (1, 50, 150, 20, 160, 2, 9, 11, "-")]
    s1 = [(0, 100, 200, 80, 220, 2, 3, 4, "+"), (0, 300, 400, 290, 450, 4, 0, 9, "."), (0, 100, 200, 95, 205, 1, 6, 6, "-")]
    t0, t1 = _table_from_rows(s0), _table_from_rows(s1)
    p0, n0 = distributed.pack_table(t0)
    assert n0 == 3 and len(p0) == 3 * 48
    m = distributed.merge_packed([distributed.pack_table(t0), distributed.pack_table(t1)], t0, 8)
    bed = m.bed12(only_anchored=False).decode().splitlines()
    # rows sorted by (chrom, thick_start, thick_end, name)
    assert bed == [
        "chrA\t80\t230\tJUNC00000001\t5\t+\t80\t230\t255,0,0\t2\t20,30\t0,120",
        "chrA\t95\t205\tJUNC00000004\t1\t-\t95\t205\t255,0,0\t2\t5,5\t0,105",
        "chrA\t280\t450\tJUNC00000002\t5\t.\t280\t450\t255,0,0\t2\t20,50\t0,120",
        "chrB\t20\t160\tJUNC00000003\t2\t-\t20\t160\t255,0,0\t2\t30,10\t0,130",
    ]
    # print_all_junctions filter: both anchors >= 8
    assert len(m.bed12(only_anchored=True).decode().splitlines()) == 3
    for t in (t0, t1):
        L.rgx_table_free(t)


def test_bed12_rows_carry_contig_names_of_any_length(built):
    """The BAM header puts no limit on l_name; Junction::print (junctions_extractor.h:90-98) writes the name through a std::string.
    A fixed line buffer here once truncated / over-read rows of contigs with names beyond ~370 characters."""
    from regtools_amd import _ffi
    L = _ffi.lib()
    names = ("c" * 5000, "x" * 511, "y" * 512, "z" * 513)
    rows = [(k, 100 + k, 300 + k, 90, 330 + k, 2 + k, k, k, "+-?."[k]) for k in range(4)]
    t = _table_from_rows(rows, names=names)
    n = L.rgx_table_format_bed12(t, 0, None, 0)
    buf = C.create_string_buffer(n + 1)
    assert L.rgx_table_format_bed12(t, 0, buf, n) == n
    # _table_from_rows leaves name_index 0 and the anchor flags unset: the row text is what this checks
    exp = "".join("%s\t90\t%d\tJUNC00000000\t%d\t%s\t90\t%d\t255,0,0\t2\t%d,30\t0,%d\n" % (names[k], 330 + k, 2 + k, "+-?."[k], 330 + k, 10 + k, 210 + k) for k in range(4))
    got = buf.raw[:n].decode()
    assert sorted(got.splitlines()) == sorted(exp.splitlines())
    L.rgx_table_free(t)


def test_python_mirror_reads_numbers_like_atoi(built):
    """junctions_extractor.cc:62-70 reads -a / -m / -M with atoi: "12x" is 12, "foo" is 0, never an exception."""
    import regtools_amd
    je = regtools_amd.JunctionsExtractor()
    je.parse_options(["-a", "12x", "-m", "foo", "-M", " 77", "-s", "XS", "in.bam"])
    assert (je.min_anchor_length_, je.min_intron_length_, je.max_intron_length_) == (12, 0, 77)
