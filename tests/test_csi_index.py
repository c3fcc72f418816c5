"""The index may be a .csi and either index may be BGZF-compressed (hts_idx_load, hts.c:2031-2042; hts_idx_load_local reads through
bgzf_open, hts.c:1569-1618).  CPU half: the oracle, the real reference where it is built (oracle/_ref) and the product's host-side
index normalisation agree on every form; GPU half: the product through the C-ABI."""
import ctypes
import os
import shutil
import subprocess

import pytest

import bamio
import csi_common
from conftest import ROOT, run_oracle
from regtools_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
REGION = {"short": "chr2:1000-90000000", "fuzz": "10:1000-200000"}
FORMS = ("bai", "csi", "csi_plain", "csi_aux", "bai_bgzf", "both")


def make_forms(tmp_path, shape="short", n=3000, seed=5):
    """One BAM, one directory per index form."""
    src = str(tmp_path / "src.bam")
    synth.write(src, n, shape=shape, seed=seed)
    bai = open(src + ".bai", "rb").read()
    out = {}
    for form in FORMS:
        d = tmp_path / form
        d.mkdir()
        bam = str(d / "x.bam")
        shutil.copy(src, bam)
        if form in ("bai", "both"):
            open(bam + ".bai", "wb").write(bai)
        if form == "bai_bgzf":
            open(bam + ".bai", "wb").write(b"".join(bamio.bgzf_member(bai[i:i + 0xff00]) for i in range(0, len(bai), 0xff00)) + bamio.EOF_MARKER)
        if form in ("csi", "both"):
            open(bam + ".csi", "wb").write(csi_common.csi_bytes(bai))
        if form == "csi_plain":
            open(bam + ".csi", "wb").write(csi_common.csi_bytes(bai, compress=False))
        if form == "csi_aux":
            open(bam + ".csi", "wb").write(csi_common.csi_bytes(bai, aux=b"\x01\x02\x03\x04\x05\x06\x07\x08"))
        out[form] = bam
    return out


@pytest.mark.parametrize("shape", ["short", "fuzz"])
def test_oracle_reads_every_index_form(tmp_path, shape):
    forms = make_forms(tmp_path, shape)
    want = None
    for form, bam in forms.items():
        for region in (".", None):
            args = ["-s", "XS", "-o", str(tmp_path / "o.bed")] + (["-r", REGION[shape]] if region is None else []) + [bam]
            rc, _, err = run_oracle(args)
            assert rc == 0, (form, err)
            got = (region, open(tmp_path / "o.bed").read())
            want = want or {}
            assert want.setdefault(region, got[1]) == got[1], form
    assert want["."].count("\n") > 10


@pytest.mark.skipif(not os.path.exists(REF), reason="the real reference is only built where /root/reference exists")
@pytest.mark.parametrize("shape", ["short", "fuzz"])
def test_oracle_equals_reference_on_every_index_form(tmp_path, shape):
    forms = make_forms(tmp_path, shape, n=2000, seed=9)
    for form, bam in forms.items():
        for extra in ([], ["-r", REGION[shape]]):
            r = subprocess.run([REF, "junctions", "extract", "-s", "XS", "-o", str(tmp_path / "r.bed")] + extra + [bam], capture_output=True)
            rc, _, err = run_oracle(["-s", "XS", "-o", str(tmp_path / "o.bed")] + extra + [bam])
            assert (r.returncode != 0) == (rc != 0), (form, r.stderr, err)
            assert open(tmp_path / "r.bed").read() == open(tmp_path / "o.bed").read(), form


@pytest.mark.skipif(not os.path.exists(REF), reason="the real reference is only built where /root/reference exists")
def test_a_broken_csi_is_not_rescued_by_the_bai_next_to_it(tmp_path):
    forms = make_forms(tmp_path)
    bam = forms["both"]
    open(bam + ".csi", "wb").write(b"CSI\1garbage")
    r = subprocess.run([REF, "junctions", "extract", "-s", "XS", "-o", str(tmp_path / "r.bed"), bam], capture_output=True)
    rc, _, err = run_oracle(["-s", "XS", "-o", str(tmp_path / "o.bed"), bam])
    assert r.returncode != 0 and rc != 0
    assert b"Unable to open BAM/SAM index" in err


def test_host_normalisation_keeps_what_the_pipeline_reads(tmp_path):
    emu = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
    emu.emu_index_summary.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_uint64)]

    def summary(blob, targets):
        out = (ctypes.c_uint64 * 5)()
        t = (ctypes.c_uint64 * len(targets))(*targets)
        got = (ctypes.c_uint64 * len(targets))()
        ok = emu.emu_index_summary(blob, len(blob), out, t, len(targets), got)
        return ok, list(out), list(got)

    for shape in ("short", "fuzz", "long"):
        bam, bai, _ = synth.generate(1500, shape=shape, seed=3)
        targets = [(len(bam) * k // 7) << 16 for k in range(1, 7)]
        ok, ref, ref_got = summary(bai, targets)
        assert ok and ref[1] == 1 and ref[4] > 0
        refs, _ = csi_common.parse_bai(bai)
        starts = {c[0] for bins, lin in refs for b, chunks in bins if b != csi_common.META_BAI for c in chunks} | {v for _, lin in refs for v in lin if v}
        assert all(g in starts or g == 2 ** 64 - 1 for g in ref_got)
        for blob in (csi_common.csi_bytes(bai), csi_common.csi_bytes(bai, compress=False), csi_common.csi_bytes(bai, aux=b"x" * 13),
                     b"".join(bamio.bgzf_member(bai[i:i + 0xff00]) for i in range(0, len(bai), 0xff00)) + bamio.EOF_MARKER):
            ok, got, anchors = summary(blob, targets)
            assert ok and got[:4] == ref[:4]
            # a CSI lists a subset of the BAI's record starts (chunk begins + the bins' lower bounds): still record starts, never earlier
            assert all(a in starts or a == 2 ** 64 - 1 for a in anchors)
            assert all(a >= b for a, b in zip(anchors, ref_got))
    for junk in (b"", b"CSI\1", b"CSI\1" + b"\xff" * 40, b"BAI", b"\x1f\x8b\x08\x04" + b"\0" * 30, csi_common.csi_bytes(bai)[:40]):
        assert summary(junk, [])[0] == 0 or junk[:4] == b"BAI\1"


# ---- GPU half: the product through the C-ABI ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["short", "fuzz"])
def test_product_reads_every_index_form(gpu_ctx, tmp_path, shape):
    from test_gpu_parity import gpu_extract
    from regtools_amd import distributed
    forms = make_forms(tmp_path, shape, n=40000, seed=21)
    for form, bam in forms.items():
        for args in (["-s", "XS"], ["-s", "RF", "-r", REGION[shape]]):
            rc, out, je = gpu_extract(gpu_ctx, bam, args)
            orc, exp, _ = run_oracle(args + [bam])
            assert rc == orc == 0 and out == exp, (form, args)
        # shard cut points come from the index's record starts: the merged shards must equal the single pass with a CSI's sparser anchors too
        _, single, _ = gpu_extract(gpu_ctx, bam, ["-s", "XS"])
        parts, keep, recs = [], [], 0
        for g in range(3):
            rc, _, je = gpu_extract(gpu_ctx, bam, ["-s", "XS"], shard=g, n_shards=3)
            assert rc == 0
            keep.append(je); parts.append(distributed.pack_table(je.table)); recs += je.stats["n_records"]
        assert recs == 40000 and distributed.merge_packed(parts, keep[0].table, 8).bed12() == single, form


@pytest.mark.gpu
def test_product_index_errors(gpu_ctx, tmp_path):
    import regtools_amd
    forms = make_forms(tmp_path, n=500)
    for blob in (b"CSI\1garbage", b"", csi_common.csi_bytes(open(forms["bai"] + ".bai", "rb").read())[:60]):
        open(forms["both"] + ".csi", "wb").write(blob)             # the .bai next to it must not rescue the run (hts.c:2031-2042)
        je = regtools_amd.JunctionsExtractor(bam=forms["both"], strandness=0, ctx=gpu_ctx)
        with pytest.raises(regtools_amd.RegtoolsError) as e:
            je.identify_junctions_from_BAM()
        assert str(e.value) == "Unable to open BAM/SAM index. Make sure alignments are indexed\n\n"


@pytest.mark.gpu
def test_identify_with_a_csi_index(gpu_ctx, tmp_path):
    """cis-splice-effects identify opens the same index (identifier.cc:288-290 -> junctions_extractor.cc:508-512)."""
    from test_gpu_cse import gpu_identify, REF_GOLD
    src = os.path.join(REF_GOLD, "test_hcc1395.2.bam")
    bam = str(tmp_path / "x.bam")
    shutil.copy(src, bam)
    shutil.copy(src + ".bai", bam + ".bai")
    csi_common.bai_to_csi(bam)
    q = dict(vcf=os.path.join(REF_GOLD, "test1.vcf"), bam=bam, fasta=os.path.join(REF_GOLD, "test_chr22.fa"), gtf=os.path.join(REF_GOLD, "test_ensemble_chr22.2.gtf"))
    rc, files, ci, msg = gpu_identify(gpu_ctx, ["-s", "RF"], q, str(tmp_path / "csi"))
    assert rc == 0, msg
    for ext, gold in (("tsv", "annotatedjunctions"), ("vcf", "annotatedvariants"), ("bed", "junctions")):
        exp = open(os.path.join(REF_GOLD, "expected-cis-splice-effects-identify-default-stranded-%s.out" % gold), "rb").read()
        assert open(files[ext], "rb").read() == exp, ext


def test_region_span_covers_every_overlapping_record():
    """bai_region_span (the member range of a -r query): every record that overlaps the region starts inside [lo, hi), both ends are
    record boundaries; the same through a .csi of the BAI's geometry and of other geometries."""
    import random
    import struct
    emu = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
    emu.emu_region_span.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    emu.emu_host_header.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    rng = random.Random(3)
    for shape, n in (("short", 60000), ("fuzz", 20000), ("long", 800)):
        bam, bai, _ = synth.generate(n, shape=shape, seed=13)
        name = ctypes.create_string_buffer(64)
        n_ref = emu.emu_host_header(bam, len(bam), name, 64)
        # records with their virtual offsets
        recs, upos = [], 0
        starts = []                                    # (inflated offset of member start, compressed offset)
        for coff, payload, isz in bamio.bgzf_members(bam):
            starts.append((upos, coff)); upos += isz
        inflated = bamio.inflate_all(bam)
        contigs, _ = bamio.split_records(inflated)
        assert n_ref == len(contigs) and name.value.decode() == contigs[0][0]
        q = 12 + struct.unpack_from("<i", inflated, 4)[0]
        for _ in contigs:
            q += 8 + struct.unpack_from("<i", inflated, q)[0]
        import bisect
        keys = [u for u, _ in starts]
        def voff(u):
            k = bisect.bisect_right(keys, u) - 1
            return starts[k][1] << 16 | (u - starts[k][0])
        while q + 4 <= len(inflated):
            bl = struct.unpack_from("<i", inflated, q)[0]
            tid, pos = struct.unpack_from("<ii", inflated, q + 4)
            l_qname = inflated[q + 12]; n_cig = struct.unpack_from("<H", inflated, q + 16)[0]; flag = struct.unpack_from("<H", inflated, q + 18)[0]
            ref = 0
            for k in range(n_cig):
                c = struct.unpack_from("<I", inflated, q + 36 + l_qname + 4 * k)[0]
                if (c & 15) in (0, 2, 3, 7, 8): ref += c >> 4
            endpos = pos + ref if (not flag & 4 and n_cig) else pos + 1
            recs.append((tid, pos, endpos, voff(q)))
            q += 4 + bl
        boundaries = {r[3] for r in recs} | {voff(q)}
        for _ in range(150):
            tid = rng.randrange(len(contigs))
            L = contigs[tid][1]
            beg = rng.randrange(0, max(1, L)); end = beg + rng.choice([1, 50, 16384, 100000, 5_000_000, L])
            lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
            r = emu.emu_region_span(bai, len(bai), tid, beg, end, ctypes.byref(lo), ctypes.byref(hi))
            inside = [v for t, p0, p1, v in recs if t == tid and p0 < end and p1 > beg]
            assert r >= 0
            if r == 0:
                assert not inside
            else:
                assert all(lo.value <= v < hi.value for v in inside), (shape, tid, beg, end)
                assert lo.value in boundaries and (hi.value in boundaries or hi.value >= max(boundaries))
        # a .csi in the default geometry (min_shift 14, depth 5 = the BAI's) gives the same kind of span
        csi = csi_common.csi_bytes(bai)
        for _ in range(60):
            tid = rng.randrange(len(contigs)); L = contigs[tid][1]
            beg = rng.randrange(0, max(1, L)); end = beg + rng.choice([1, 16384, 100000, L])
            r = emu.emu_region_span(csi, len(csi), tid, beg, end, ctypes.byref(lo), ctypes.byref(hi))
            inside = [v for t, p0, p1, v in recs if t == tid and p0 < end and p1 > beg]
            assert r >= 0 and (r == 1 or not inside)
            if r == 1:
                assert all(lo.value <= v < hi.value for v in inside) and lo.value in boundaries
        # a .csi of any other geometry (min_shift / depth as `samtools index -c -m` takes them; contigs longer than 2^29 need one): the span
        # comes from the real bins and their loffs (hts_itr_query, hts.c:1708-1800) and is as narrow as a .bai's
        for min_shift, depth in ((12, 6), (10, 7), (15, 5), (14, 6)):
            csi = csi_common.csi_from_bam(bam, min_shift, depth)
            narrow = narrow_bai = 0
            lo2, hi2 = ctypes.c_uint64(), ctypes.c_uint64()
            for _ in range(40):
                tid = rng.randrange(len(contigs)); L = contigs[tid][1]
                beg = rng.randrange(0, max(1, L)); end = beg + rng.choice([1, 4096, 100000, L])
                r = emu.emu_region_span(csi, len(csi), tid, beg, end, ctypes.byref(lo), ctypes.byref(hi))
                inside = [v for t, p0, p1, v in recs if t == tid and p0 < end and p1 > beg]
                assert r >= 0 and (r == 1 or not inside), (min_shift, depth, tid, beg, end)
                if r == 1:
                    assert all(lo.value <= v < hi.value for v in inside) and lo.value in boundaries
                    narrow += (hi.value >> 16) - (lo.value >> 16) < len(bam) // 4
                else:
                    narrow += 1
                if emu.emu_region_span(bai, len(bai), tid, beg, end, ctypes.byref(lo2), ctypes.byref(hi2)) == 1:
                    narrow_bai += (hi2.value >> 16) - (lo2.value >> 16) < len(bam) // 4
                else:
                    narrow_bai += 1
            assert narrow >= 3 and narrow >= narrow_bai - 2, (min_shift, depth, narrow, narrow_bai)       # small regions read a small part of the file, as with a .bai
        gz = b"".join(bamio.bgzf_member(bai[i:i + 0xff00]) for i in range(0, len(bai), 0xff00)) + bamio.EOF_MARKER
        assert emu.emu_region_span(gz, len(gz), 0, 100, 5000, ctypes.byref(lo), ctypes.byref(hi)) >= 0


# ---- a .csi of another geometry (min_shift / depth): what a genome with contigs beyond 2^29 needs -----------------------------------------
GEOMETRIES = ((12, 6), (10, 7), (15, 5), (14, 6))


def make_geometry(tmp_path, min_shift, depth, shape="short", n=3000, seed=5):
    d = tmp_path / ("g%d_%d" % (min_shift, depth))
    d.mkdir()
    bam = str(d / "x.bam")
    synth.write(bam, n, shape=shape, seed=seed)
    os.remove(bam + ".bai")
    open(bam + ".csi", "wb").write(csi_common.csi_from_bam(open(bam, "rb").read(), min_shift, depth))
    return bam


def small_regions(bed_text, k=3):
    """regions around a few junction rows of a whole-file run, plus a whole contig and an empty stretch"""
    rows = [l.split("\t") for l in bed_text.splitlines()]
    out = ["%s:%d-%d" % (r[0], max(1, int(r[1]) - 50), int(r[2]) + 50) for r in rows[::max(1, len(rows) // k)][:k]]
    return out + [rows[0][0], "%s:1-2" % rows[-1][0]]


@pytest.mark.skipif(not os.path.exists(REF), reason="the real reference is only built where /root/reference exists")
@pytest.mark.parametrize("geo", GEOMETRIES, ids=["m%d_d%d" % g for g in GEOMETRIES])
def test_oracle_equals_reference_on_other_csi_geometries(tmp_path, geo):
    bam = make_geometry(tmp_path, *geo, n=20000, seed=9)
    rc, _, err = run_oracle(["-s", "XS", "-o", str(tmp_path / "w.bed"), bam])
    assert rc == 0, err
    for region in small_regions(open(tmp_path / "w.bed").read()):
        r = subprocess.run([REF, "junctions", "extract", "-s", "XS", "-o", str(tmp_path / "r.bed"), "-r", region, bam], capture_output=True)
        rc, _, err = run_oracle(["-s", "XS", "-o", str(tmp_path / "o.bed"), "-r", region, bam])
        assert r.returncode == rc == 0, (region, r.stderr, err)
        assert open(tmp_path / "r.bed").read() == open(tmp_path / "o.bed").read(), region


@pytest.mark.gpu
@pytest.mark.parametrize("geo", GEOMETRIES, ids=["m%d_d%d" % g for g in GEOMETRIES])
def test_product_region_queries_through_other_csi_geometries(gpu_ctx, tmp_path, geo):
    """-r through a .csi of another geometry: equal to the oracle (pinned to the reference above), and the member range comes from the
    index -- a small region inflates a small part of the file, as with a .bai"""
    from test_gpu_parity import gpu_extract
    bam = make_geometry(tmp_path, *geo, n=200000, seed=21)
    rc, whole, je = gpu_extract(gpu_ctx, bam, ["-s", "XS"])
    assert rc == 0 and whole.count(b"\n") > 50
    all_members = je.stats["n_members"]
    narrow = 0
    for region in small_regions(whole.decode()):
        rc, out, je = gpu_extract(gpu_ctx, bam, ["-s", "RF", "-r", region])
        orc, exp, _ = run_oracle(["-s", "RF", "-r", region, bam])
        assert rc == orc == 0 and out == exp, region
        narrow += je.stats["n_members"] * 4 < all_members
    assert narrow >= 3, (geo, narrow, all_members)
