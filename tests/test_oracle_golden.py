"""Pins the oracle (oracle/oracle.c, the CPU restatement) against (a) the reference's own six
junctions-extract golden files, (b) the reference's gtest known answers, (c) outputs of the real reference
(oracle/_ref) on hand-made and synthetic inputs stored under tests/golden/ by make_golden.py.  CPU only."""
import ctypes
import os

import pytest

import cases
from conftest import run_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def synth_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("synth")


@pytest.mark.parametrize("args,golden", cases.REF_GOLDENS, ids=[g for _, g in cases.REF_GOLDENS])
def test_reference_integration_goldens(args, golden):
    rc, out, _ = run_oracle(args + [os.path.join(cases.GOLD, "test_hcc1395.bam")])
    assert rc == 0
    assert out == open(os.path.join(cases.GOLD, "junctions-extract", golden), "rb").read()


@pytest.mark.parametrize("case", cases.MANIFEST, ids=[c["name"] for c in cases.MANIFEST])
def test_oracle_equals_reference_outputs(case, synth_dir):
    rc, out, _ = run_oracle(cases.case_argv(case, synth_dir))
    assert rc == case["rc"]
    assert out == cases.expected(case)


def test_reference_exit_codes(tmp_path):
    # tests/integration-test/test_junctions_extract.py:87-109
    assert run_oracle(["-s", "XS"])[0] == 1                                   # no bam
    assert run_oracle(["-s", "XS", "does_not_exist.bam"])[0] == 1             # missing bam
    assert run_oracle([os.path.join(cases.GOLD, "test_hcc1395.bam")])[0] == 1  # no -s
    # BAM without an index is fatal even for whole-file runs (junctions_extractor.cc:508-512)
    p = tmp_path / "noidx.bam"
    p.write_bytes(open(os.path.join(cases.GOLD, "strand.bam"), "rb").read())
    rc, _, err = run_oracle(["-s", "XS", str(p)])
    assert rc == 1 and b"Unable to open BAM/SAM index" in err


@pytest.fixture(scope="module")
def lib(built):
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    L.orc_get_bin.restype = ctypes.c_uint32
    L.orc_get_bin.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    L.orc_strand_from_flag.restype = ctypes.c_char
    L.orc_strand_from_flag.argtypes = [ctypes.c_uint32, ctypes.c_int]
    return L


def test_gtf_bin_known_answer(lib):
    # tests/lib/gtf/test_gtf_parser.cc:113-118 pins the 32678 typo: exon 12791-14103 -> bin 37359
    assert lib.orc_get_bin(12791, 14103) == 37359
    assert lib.orc_get_bin(0, 1 << 14) == 37359            # finest level, window 0
    assert lib.orc_get_bin((1 << 14) - 1, (1 << 14) + 1) == 4681   # crosses a 16 kb boundary -> level 1


def test_flag_strand_table(lib):
    # SURVEY.md 9.5, oracle-verified: RF over k = flag >> 4
    rf = "+??-?+-??-+?-??+"
    for k in range(16):
        assert lib.orc_strand_from_flag(k << 4, 1).decode() == rf[k]
        fr = {"+": "-", "-": "+", "?": "?"}[rf[k]]
        assert lib.orc_strand_from_flag(k << 4, 2).decode() == fr


def test_cigar_state_machine_known_answers(lib):
    class Cand(ctypes.Structure):
        _fields_ = [("start", ctypes.c_uint32), ("end", ctypes.c_uint32), ("ts", ctypes.c_uint32), ("te", ctypes.c_uint32)]
    import bamio

    def walk(pos, cigar):
        ops = bamio.parse_cigar(cigar)
        arr = (ctypes.c_uint32 * len(ops))(*[l << 4 | o for l, o in ops])
        out = (Cand * 16)()
        n = lib.orc_cigar_walk(pos, arr, len(ops), out, 16)
        return [(out[i].start, out[i].end, out[i].ts, out[i].te) for i in range(n)]
    # SURVEY.md 9.7 rows 0, 3, 5, 10 (start, end, thick_start, thick_end)
    assert walk(1000000, "20M100N20M100N20M") == [(1000020, 1000120, 1000000, 1000140), (1000140, 1000240, 1000120, 1000260)]
    assert walk(4000000, "20M100N10M1D10M100N20M") == [(4000020, 4000120, 4000000, 4000130), (4000141, 4000241, 4000131, 4000261)]
    assert walk(6000000, "20M100N100N20M") == [(6000020, 6000120, 6000000, 6000120), (6000120, 6000220, 6000120, 6000240)]
    assert walk(11000000, "10=1X100N20M") == [(11000011, 11000111, 11000011, 11000131)]
    assert walk(5, "3S") == [] and walk(5, "20M") == []
