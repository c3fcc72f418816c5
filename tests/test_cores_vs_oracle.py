"""The per-alignment device cores (bam_core.h / cse_core.h, compiled for the host by tests/hostemu) against the oracle's
restatements on generated inputs: the CIGAR state machine, the flag strand rule, the UCSC bin with the reference's offset
typo, and bam_endpos.  No GPU needed; hypothesis drives the inputs."""
import ctypes as C
import os

import pytest
from hypothesis import given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Cand(C.Structure):
    _fields_ = [("start", C.c_uint32), ("end", C.c_uint32), ("thick_start", C.c_uint32), ("thick_end", C.c_uint32)]


@pytest.fixture(scope="module")
def libs(built):
    emu = C.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    emu.emu_cigar_walk.argtypes = [C.c_int32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(Cand), C.c_int]
    orc.orc_cigar_walk.argtypes = [C.c_int32, C.POINTER(C.c_uint32), C.c_int, C.POINTER(Cand), C.c_int]
    emu.emu_ucsc_bin.restype = C.c_uint32
    orc.orc_get_bin.restype = C.c_uint32
    orc.orc_strand_from_flag.restype = C.c_char
    emu.emu_rec_endpos.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_int32]
    emu.emu_rec_endpos.restype = C.c_int32
    return emu, orc


# op codes 0..9 are MIDNSHP=XB; 10..15 never appear in valid files but must not derail anything
cigar_op = st.tuples(st.sampled_from([0, 0, 0, 3, 3, 1, 2, 4, 5, 6, 7, 8, 9, 12]),
                     st.one_of(st.integers(0, 300), st.integers(0, (1 << 28) - 1)))
cigars = st.lists(cigar_op, min_size=0, max_size=70)


@settings(max_examples=1500, deadline=None)
@given(pos=st.integers(-1, (1 << 29) - 1), ops=cigars)
def test_cigar_state_machine_equals_oracle(libs, pos, ops):
    emu, orc = libs
    n = len(ops)
    arr = (C.c_uint32 * max(n, 1))(*[(l << 4) | o for o, l in ops])
    a, b = (Cand * 80)(), (Cand * 80)()
    na = emu.emu_cigar_walk(pos, arr, n, a, 80)
    nb = orc.orc_cigar_walk(pos, arr, n, b, 80)
    assert na == nb
    for i in range(min(na, 80)):
        assert (a[i].start, a[i].end, a[i].thick_start, a[i].thick_end) == (b[i].start, b[i].end, b[i].thick_start, b[i].thick_end), (i, ops)


def test_flag_strand_rule_equals_oracle_for_every_flag(libs):
    emu, orc = libs
    for strandness in (1, 2, 3):
        for flag in range(1 << 12):
            assert chr(emu.emu_strand_from_flag(flag, strandness)) == orc.orc_strand_from_flag(flag, strandness).decode(), (flag, strandness)


@settings(max_examples=3000, deadline=None)
@given(start=st.integers(0, (1 << 32) - 1), length=st.integers(0, 1 << 30))
def test_ucsc_bin_equals_oracle(libs, start, length):
    emu, orc = libs
    end = min(start + length, (1 << 32) - 1)
    assert emu.emu_ucsc_bin(start, end) == orc.orc_get_bin(start, end)


@settings(max_examples=1500, deadline=None)
@given(pos=st.integers(0, (1 << 29) - 1), flag=st.integers(0, 0xfff), ops=st.lists(st.tuples(st.integers(0, 9), st.integers(0, 100000)), max_size=64))
def test_endpos_is_bam_endpos(libs, pos, flag, ops):
    """sam.c:336-342: pos + reference length of the CIGAR (M, D, N, =, X), or pos + 1 for unmapped reads and empty CIGARs"""
    emu, _ = libs
    n = len(ops)
    arr = (C.c_uint32 * max(n, 1))(*[(l << 4) | o for o, l in ops])
    exp = pos + 1
    if not (flag & 4) and n > 0:
        exp = pos + sum(l for o, l in ops if o in (0, 2, 3, 7, 8))
    assert emu.emu_rec_endpos(arr, n, flag, pos) == ((exp + (1 << 31)) % (1 << 32)) - (1 << 31)
