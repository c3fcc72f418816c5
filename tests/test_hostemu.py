"""CPU unit tests of the per-lane device cores compiled for the host (tests/hostemu): the DEFLATE decoder
against zlib, on real BGZF members and on adversarial streams.  No GPU needed."""
import ctypes
import os
import random
import zlib

import pytest

import bamio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(built):
    return ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))


_phase = [0]


def inflate(emu, payload, cap=65536):
    """Both decoders on the same payload: round 1's lane decoder (inflate_core.h) and the LDS-window decoder (inflate_ring.h, one-lane
    wave, destination phase cycling through all 128 line offsets); they must agree, and the common answer is returned."""
    out = ctypes.create_string_buffer(cap + 64)
    n = ctypes.c_uint32(0)
    buf = ctypes.create_string_buffer(payload + b"\0" * 16, len(payload) + 16)
    st = emu.emu_inflate(buf, len(payload), out, cap, ctypes.byref(n))
    got = out.raw[:n.value]
    out2 = ctypes.create_string_buffer(cap + 64)
    n2 = ctypes.c_uint32(0)
    _phase[0] = (_phase[0] + 37) % 128
    st2 = emu.emu_inflate_ring(buf, len(payload), out2, cap, ctypes.byref(n2), _phase[0])
    assert st2 not in (-100, -101), "the ring decoder wrote outside its member (phase %d)" % _phase[0]
    assert (st2 == 0) == (st == 0), (st, st2, _phase[0])
    if st == 0:
        assert out2.raw[:n2.value] == got, "ring decoder differs at phase %d" % _phase[0]
    # ... and the wave-per-member decoder (inflate_wave.h): direct tables + canonical walk, whole member in (emulated) LDS
    out3 = ctypes.create_string_buffer(cap + 64)
    n3 = ctypes.c_uint32(0)
    st3 = emu.emu_inflate_wave(buf, len(payload), out3, cap, ctypes.byref(n3), _phase[0])
    assert st3 not in (-100, -101), "the wave decoder wrote outside its member"
    assert (st3 == 0) == (st == 0), (st, st3)
    if st == 0:
        assert out3.raw[:n3.value] == got, "wave decoder differs"
    # ... and the lane decoder taking up to four literals per trip (same symbols in the same order, same status)
    out5 = ctypes.create_string_buffer(cap + 64)
    n5 = ctypes.c_uint32(0)
    st5 = emu.emu_inflate_lits(buf, len(payload), out5, cap, ctypes.byref(n5))
    assert st5 == st and n5.value == n.value and out5.raw[:n5.value] == got, (st, st5, n.value, n5.value)
    # ... and round 3's decoder (inflate_coop.h): long matches split into head / wave-copied aligned body / tail
    # with and without the second symbol of a trip (bit 0), plain / windowed bit reader (bit 1), literal pair + match per trip (bit 2), runs written from registers (bit 3)
    for pairs in (0, 1, 2, 3, 5, 7, 8, 10, 13, 15):
        out4 = ctypes.create_string_buffer(cap + 64)
        n4 = ctypes.c_uint32(0)
        st4 = emu.emu_inflate_coop(buf, len(payload), out4, cap, ctypes.byref(n4), _phase[0], pairs)
        assert st4 not in (-100, -101), "the coop decoder wrote outside its member (phase %d)" % _phase[0]
        assert (st4 == 0) == (st == 0), (st, st4, _phase[0], pairs)
        if st == 0:
            assert out4.raw[:n4.value] == got, "coop decoder differs at phase %d (pairs %d)" % (_phase[0], pairs)
    return st, got


def test_members_of_synthetic_bams(emu, tmp_path):
    from regtools_amd import synth
    for shape, n in (("short", 6000), ("fuzz", 4000), ("long", 60)):
        bam, _, _ = synth.generate(n, shape=shape, seed=5)
        k = 0
        for _, payload, isize in bamio.bgzf_members(bam):
            st, got = inflate(emu, payload)
            assert st == 0 and got == zlib.decompress(payload, -15) and len(got) == isize
            k += 1
        assert k > 2


def test_realistic_payload_members(emu):
    from regtools_amd import synth
    bam, _, _ = synth.generate(3000, shape="short", seed=9, realistic=True)
    for _, payload, isize in bamio.bgzf_members(bam):
        st, got = inflate(emu, payload)
        assert st == 0 and got == zlib.decompress(payload, -15)


@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
def test_block_types_and_strategies(emu, strategy):
    rnd = random.Random(strategy + 17)
    for trial in range(60):
        n = rnd.choice([0, 1, 2, 7, 8, 15, 16, 17, 63, 64, 65, 300, 5000, 65280, 65536])
        kind = rnd.randrange(5)
        if kind == 0: data = bytes(rnd.getrandbits(8) for _ in range(n))                # incompressible -> stored blocks
        elif kind == 1: data = bytes(rnd.choice(b"ACGT") for _ in range(n))
        elif kind == 2: data = (b"abcdefghij" * 7000)[:n]                                  # period 10
        elif kind == 3: data = bytes([rnd.randrange(2)]) * n                               # distance-1 runs
        else: data = bytes((i * 7) % 251 for i in range(n))
        for lvl in (0, 1, 6, 9):
            c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strategy)
            p = c.compress(data) + c.flush()
            st, got = inflate(emu, p)
            assert st == 0 and got == data, (trial, n, kind, lvl)


def test_every_short_distance_and_length(emu):
    # matches with distance 1..40 and every tail length: exercises the period-widening and exact-tail paths
    for dist in list(range(1, 41)) + [63, 64, 65, 127, 128, 129, 300]:
        seed = bytes((i * 37 + 11) % 256 for i in range(dist))
        for total in (dist + 3, dist + 15, dist + 16, dist + 17, dist + 64, dist + 257, dist + 258, dist + 700):
            data = (seed * (total // dist + 2))[:total]
            c = zlib.compressobj(9, zlib.DEFLATED, -15, 9)
            p = c.compress(data) + c.flush()
            st, got = inflate(emu, p, cap=len(data))     # cap == exact size: no slack at the end
            assert st == 0 and got == data, (dist, total)


def test_output_stage_respects_every_alignment_and_both_member_ends(emu):
    """The decoder writes aligned 16-byte chunks cut on the DESTINATION address; the chunks cut by the two ends of the member
    must be written byte-exactly (a neighbouring member's lane owns the bytes on the other side)."""
    rnd = random.Random(11)
    payloads = []
    for size in (1, 2, 15, 16, 17, 31, 33, 100, 1000, 5000):
        data = bytes(rnd.choice(b"ACGT") if rnd.random() < 0.7 else rnd.randrange(256) for _ in range(size))
        payloads.append((data, zlib.compress(data, 6)[2:-4]))
    stored = bytes(rnd.randrange(256) for _ in range(300))
    c = zlib.compressobj(0, zlib.DEFLATED, -15)
    payloads.append((stored, c.compress(stored) + c.flush()))
    c = zlib.compressobj(6, zlib.DEFLATED, -15)                      # deflate, sync flush (empty stored block), deflate
    mixed = c.compress(b"abcabcabcabc" * 9) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(b"xyzzy" * 31) + c.flush()
    payloads.append((b"abcabcabcabc" * 9 + b"xyzzy" * 31, mixed))
    raw = ctypes.create_string_buffer(8192 + 256)
    base = ctypes.addressof(raw)
    for data, p in payloads:
        src = ctypes.create_string_buffer(p + b"\0" * 16, len(p) + 16)
        for mis in range(16):
            off = (-base) % 16 + 64 + mis
            ctypes.memset(base, 0xA5, len(raw))
            n = ctypes.c_uint32(0)
            st = emu.emu_inflate(src, len(p), ctypes.c_void_p(base + off), len(data), ctypes.byref(n))
            got = raw.raw
            assert st == 0 and n.value == len(data), (len(data), mis, st)
            assert got[off:off + len(data)] == data, (len(data), mis)
            assert got[:off] == b"\xa5" * off and got[off + len(data):] == b"\xa5" * (len(raw) - off - len(data)), (len(data), mis)


def test_output_capacity_is_respected(emu):
    data = b"xyz" * 1000
    p = zlib.compress(data, 6)[2:-4]
    st, got = inflate(emu, p, cap=100)
    assert st != 0 and len(got) <= 100


def test_corrupt_streams_never_crash_and_agree_with_zlib_when_valid(emu):
    rnd = random.Random(3)
    for trial in range(1500):
        data = bytes(rnd.choice(b"ACGTN") for _ in range(rnd.choice([50, 2000])))
        p = bytearray(zlib.compress(data, 6)[2:-4])
        k = rnd.randrange(len(p))
        p[k] ^= 1 << rnd.randrange(8)
        try:
            exp = zlib.decompress(bytes(p), -15)
        except zlib.error:
            exp = None
        st, got = inflate(emu, bytes(p))
        if exp is not None and len(exp) <= 65536:
            assert st == 0 and got == exp
    # truncated payloads
    p = zlib.compress(b"hello world " * 500, 6)[2:-4]
    for cut in range(0, len(p), 7):
        st, got = inflate(emu, p[:cut])
        assert st != 0


def test_reads_stay_within_16_bytes_of_the_payload_even_on_corrupt_streams(emu):
    """The decoder prefetches its bit stream; the C ABI promises the caller that nothing beyond bam_len + 8 is touched (the payload of the
    last member ends 8 bytes before that).  The payload is placed so that the 17th byte after it is an unmapped page: an over-read would
    kill the test process."""
    import mmap
    libc = ctypes.CDLL(None, use_errno=True)
    page = mmap.PAGESIZE
    m = mmap.mmap(-1, 3 * page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert libc.mprotect(ctypes.c_void_p(base + 2 * page), page, 0) == 0          # PROT_NONE guard page
    rnd = random.Random(17)
    out = ctypes.create_string_buffer(65536 + 64)
    n = ctypes.c_uint32(0)
    try:
        for trial in range(600):
            data = bytes(rnd.choice(b"ACGTN!#") for _ in range(rnd.choice([40, 900, 5000])))
            p = bytearray(zlib.compress(data, 6)[2:-4])
            mode = trial % 3
            if mode == 1:
                k = rnd.randrange(len(p)); p[k] ^= 1 << rnd.randrange(8)
            elif mode == 2:
                p = p[: rnd.randrange(1, len(p))]
            start = 2 * page - 16 - len(p)
            m[start:start + len(p)] = bytes(p)
            m[2 * page - 16:2 * page] = bytes(rnd.randrange(256) for _ in range(16))
            st = emu.emu_inflate(ctypes.c_void_p(base + start), len(p), out, 65536, ctypes.byref(n))
            if mode == 0:
                assert st == 0 and out.raw[:n.value] == data
    finally:
        libc.mprotect(ctypes.c_void_p(base + 2 * page), page, 3)


def test_ring_decoder_every_destination_phase_and_window_edge(emu):
    """inflate_ring.h: matches at the distances around the ring's near limit (352) and its size (384), lengths around the 128-byte batch,
    at every destination phase of a 128-byte line; guard bytes around the member must stay untouched (emu_inflate_ring checks them)."""
    rnd = random.Random(23)
    for dist in (1, 3, 4, 5, 15, 16, 17, 127, 128, 129, 220, 228, 351, 352, 353, 367, 368, 369, 383, 384, 385, 400, 1000, 32768):
        seed = bytes(rnd.randrange(256) for _ in range(dist))
        for total in (dist + 3, dist + 130, dist + 258, dist + 700):
            data = (seed * (total // dist + 2))[:total]
            c = zlib.compressobj(9, zlib.DEFLATED, -15, 9)
            p = c.compress(data) + c.flush()
            for phase in (0, 1, 15, 16, 17, 63, 64, 100, 127, rnd.randrange(128)):
                out = ctypes.create_string_buffer(len(data) + 64)
                n = ctypes.c_uint32(0)
                st = emu.emu_inflate_ring(p, len(p), out, len(data), ctypes.byref(n), phase)
                assert st == 0 and out.raw[:n.value] == data, (dist, total, phase, st)


def test_coop_decoder_every_destination_phase_and_copy_split(emu):
    """inflate_coop.h: matches around the lane / wave hand-over (64 bytes), around whole-chunk counts (head + 16 k + tail) and the longest
    match, distances from runs to the window's end, at every destination phase of a 16-byte chunk; guard bytes around the member must stay
    untouched (emu_inflate_coop checks them)."""
    rnd = random.Random(29)
    for dist in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 63, 64, 65, 66, 79, 80, 81, 127, 128, 129, 221, 257, 258, 259, 300, 1000, 32768):
        seed = bytes(rnd.randrange(256) for _ in range(dist))
        for total in (dist + 3, dist + 17, dist + 64, dist + 65, dist + 80, dist + 130, dist + 258, dist + 259, dist + 700):
            data = (seed * (total // dist + 2))[:total]
            c = zlib.compressobj(9, zlib.DEFLATED, -15, 9)
            p = c.compress(data) + c.flush()
            for phase in list(range(16)) + [31, 100, 127]:
                out = ctypes.create_string_buffer(len(data) + 64)
                n = ctypes.c_uint32(0)
                st = emu.emu_inflate_coop(p, len(p), out, len(data), ctypes.byref(n), phase, (phase & 3) | (4 if phase & 5 == 5 or phase > 16 else 0) | (8 if phase & 2 or dist <= 16 and phase & 1 else 0))
                assert st == 0 and out.raw[:n.value] == data, (dist, total, phase, st)


def test_parallel_member_scan_equals_the_serial_walk(emu):
    """scan_members_parallel (the member list rgx_extract_mem launches the inflate from while the file is still uploading): for a well-formed
    file exactly the BSIZE chain from offset 0, for any thread count; for a file that is not (cut inside a member, bytes appended, a member's
    header damaged, an oversized ISIZE) it must say no -- the device's member discovery then decides what the reference would do."""
    from regtools_amd import synth
    emu.emu_scan_members.restype = ctypes.c_long
    emu.emu_scan_members.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]

    def scan(data, threads):
        cap = 200000
        arr = (ctypes.c_uint64 * (4 * cap))()
        total = ctypes.c_uint64(0)
        n = emu.emu_scan_members(data, len(data), threads, arr, cap, ctypes.byref(total))
        return n, [tuple(arr[4 * k: 4 * k + 4]) for k in range(max(n, 0))], total.value
    for shape, n_reads in (("short", 400000), ("fuzz", 30000), ("long", 3000)):
        bam, _, _ = synth.generate(n_reads, shape=shape, seed=3)
        want, up = [], 0
        for off, payload, isize in bamio.bgzf_members(bam):                    # (the empty EOF member included)
            want.append((off + 18, up, len(payload) + 8, isize)); up += isize
        for threads in (1, 2, 3, 7, 24):
            n, got, total = scan(bam, threads)
            assert n == len(got) and total == up
            assert [(c, u, i) for c, u, _, i in got] == [(c, u, i) for c, u, _, i in want]
            assert all(cl <= len(bam) - 8 - c for c, _, cl, _ in got)
        assert scan(bam[:-40], 4)[0] == -1 and scan(bam[:len(bam) // 2], 4)[0] == -1          # cut inside a member
        assert scan(bam + b"\0" * 100, 4)[0] == -1                                             # bytes behind the last member
        k = got[len(got) // 2][0] - 18
        assert scan(bam[:k] + b"\x1e" + bam[k + 1:], 4)[0] == -1                               # a header that is not one
        big = got[len(got) // 3]
        foot = big[0] + big[2] - 4                                                              # ISIZE footer of that member
        assert scan(bam[:foot] + (70000).to_bytes(4, "little") + bam[foot + 4:], 4)[0] == -1   # claims more than a BGZF block holds


@pytest.mark.parametrize("threads", [1, 2, 5, 16])
def test_worker_pool_parallel_sort_and_huge_page_vectors(emu, threads):
    """worker_pool.h (every task once; parallel_sort == std::sort, 5 .. 300,000 keys with many ties) and bigvec.h (2 MiB-aligned blocks, growth keeps
    the contents, class types are constructed)."""
    for seed, n in ((1, 5), (2, 1000), (3, 16384), (4, 100_001), (5, 300_000)):
        assert emu.emu_pool_selftest(seed, n, threads) == 0, (seed, n, threads)
