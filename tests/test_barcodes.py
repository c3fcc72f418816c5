"""`junctions extract -b`: per-junction cell-barcode counts (junctions_extractor.cc:362-374, :204-217; Junction::print_barcodes h:99-111).
The second output file lists every junction's distinct barcodes in the iteration order of a std::unordered_map, so the container's
layout is part of the parity contract: the oracle restates libstdc++'s and is pinned here against the real container, against what the
real reference printed (tests/golden/barcodes/, make_golden_barcodes.py) and -- where it is built -- the real reference itself."""
import ctypes
import os
import random
import subprocess

import pytest

import barcode_cases as bc
import bamio
from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
ORACLE = os.path.join(ROOT, "oracle", "oracle_cli")
ALL = [(n, a) for n in bc.CASES for a in bc.ARGS[n]]
IDS = ["%s-%s" % (n, bc.arg_tag(a)) for n, a in ALL]


def golden(name, args, ext):
    return open(os.path.join(bc.GOLD, "%s.%s.%s" % (name, bc.arg_tag(args), ext)), "rb").read()


def test_committed_bams_are_what_the_builder_makes(tmp_path):
    for name in bc.CASES:
        p = str(tmp_path / (name + ".bam"))
        bc.build(name, p)
        assert bamio.inflate_all(p) == bamio.inflate_all(os.path.join(bc.GOLD, name + ".bam")), name


@pytest.mark.parametrize("name,args", ALL, ids=IDS)
def test_oracle_equals_reference_outputs(tmp_path, name, args):
    bed, bcf = str(tmp_path / "o.bed"), str(tmp_path / "o.bc")
    r = subprocess.run([ORACLE, "extract"] + args + ["-o", bed, "-b", bcf, os.path.join(bc.GOLD, name + ".bam")], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert open(bed, "rb").read() == golden(name, args, "bed")
    got, want = open(bcf, "rb").read(), golden(name, args, "barcodes")
    assert got == want
    # the goldens are worth something: some junction went through several rehashes, and lines match rows
    assert want.count(b"\n") == golden(name, args, "bed").count(b"\n")
    if name == "rehash":
        assert max(int(l.split(b"\t")[0]) for l in want.splitlines()) > 2500


def test_restated_container_equals_the_real_unordered_map():
    emu = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    SZ = ctypes.c_size_t
    emu.emu_umap_order.restype = SZ
    orc.orc_umap_order.restype = SZ
    rng = random.Random(11)
    seen_buckets = set()
    for n_distinct, n in ((1, 5), (2, 9), (11, 40), (13, 13), (14, 60), (29, 100), (30, 31), (59, 400), (60, 60), (127, 128), (200, 1500), (400, 3000), (600, 2500), (900, 5000), (3000, 3500), (4200, 9000)):
        pool = ["".join(rng.choice("ACGTN-1?") for _ in range(rng.choice([0, 1, 7, 8, 9, 15, 16, 17, 18, 33]))) for _ in range(n_distinct // 8)]
        pool += ["".join(rng.choice("ACGT") for _ in range(16)) + "-1" for _ in range(n_distinct - len(pool))]
        keys = [rng.choice(pool) for _ in range(n)]
        arr = (ctypes.c_char_p * n)(*[k.encode() for k in keys])
        o1, c1, o2, c2 = (SZ * n)(), (ctypes.c_int * n)(), (SZ * n)(), (ctypes.c_int * n)()
        nb = SZ()
        k1 = emu.emu_umap_order(arr, SZ(n), o1, c1, ctypes.byref(nb))
        k2 = orc.orc_umap_order(arr, SZ(n), o2, c2)
        assert k1 == k2 == len(set(keys))
        assert list(o1)[:k1] == list(o2)[:k2] and list(c1)[:k1] == list(c2)[:k2], (n_distinct, n)
        seen_buckets.add(nb.value)
    assert {13, 29, 59, 127, 257, 541, 1109, 5087} <= seen_buckets, seen_buckets          # the growth sequence the oracle tabulates


@pytest.mark.skipif(not os.path.exists(REF), reason="the real reference is only built where /root/reference exists")
def test_identify_b_prints_an_empty_map_per_junction(tmp_path):
    # identify's extractor is constructed without a barcode file (identifier.cc:288): "0\t" per junction
    g = os.path.join(ROOT, "tests", "golden", "cse_ref")
    out = {k: str(tmp_path / k) for k in ("tsv", "bed", "bc")}
    r = subprocess.run([REF, "cis-splice-effects", "identify", "-s", "RF", "-o", out["tsv"], "-j", out["bed"], "-b", out["bc"], os.path.join(g, "test1.vcf"),
                        os.path.join(g, "test_hcc1395.2.bam"), os.path.join(g, "test_chr22.fa"), os.path.join(g, "test_ensemble_chr22.2.gtf")], capture_output=True)
    assert r.returncode == 0
    n = open(out["bed"]).read().count("\n")
    assert n > 0 and open(out["bc"]).read() == "0\t\n" * n


# ---- GPU half: the product through the C-ABI ---------------------------------------------------------------------------------------------
def gpu_extract_b(ctx, bam, args, tmp_path):
    import regtools_amd
    bed, bcf = str(tmp_path / "g.bed"), str(tmp_path / "g.bc")
    je = regtools_amd.JunctionsExtractor(ctx=ctx)
    je.parse_options(list(args) + ["-o", bed, "-b", bcf, bam])
    je.identify_junctions_from_BAM()
    je.print_all_junctions()
    return open(bed, "rb").read(), open(bcf, "rb").read(), je


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", ALL, ids=IDS)
def test_product_equals_reference_outputs(gpu_ctx, tmp_path, name, args):
    bed, bcs, je = gpu_extract_b(gpu_ctx, os.path.join(bc.GOLD, name + ".bam"), args, tmp_path)
    assert bed == golden(name, args, "bed")
    assert bcs == golden(name, args, "barcodes")
    rows, maps = je.get_all_junctions(), je.get_barcodes()
    assert len(rows) == len(maps)
    for j, m in zip(rows, maps):
        assert sum(c for _, c in m) == j.read_count and len({b for b, _ in m}) == len(m)
    # a binding refills Junction::barcodes in first-seen order (bc_insert_rank): the REAL container must then iterate in the listed order
    emu = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
    emu.emu_umap_order.restype = ctypes.c_size_t
    for listed, ins in zip(maps, je.get_barcodes(insertion_order=True)):
        if len(listed) < 2 or b"" in [b for b, _ in ins]:
            continue
        n = len(ins)
        arr = (ctypes.c_char_p * n)(*[b for b, _ in ins])
        order, counts, nb = (ctypes.c_size_t * n)(), (ctypes.c_int * n)(), ctypes.c_size_t()
        assert emu.emu_umap_order(arr, ctypes.c_size_t(n), order, counts, ctypes.byref(nb)) == n
        assert [ins[order[k]][0] for k in range(n)] == [b for b, _ in listed]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n", [("short", 120000), ("fuzz", 60000), ("long", 1500)])
def test_product_equals_oracle_without_any_tag(gpu_ctx, tmp_path, shape, n):
    # no CB anywhere: every junction's map is {"?": read_count}
    from regtools_amd import synth
    bam = str(tmp_path / "x.bam")
    synth.write(bam, n, shape=shape, seed=31)
    bed, bcs, je = gpu_extract_b(gpu_ctx, bam, ["-s", "XS"], tmp_path)
    r = subprocess.run([ORACLE, "extract", "-s", "XS", "-o", str(tmp_path / "o.bed"), "-b", str(tmp_path / "o.bc"), bam], capture_output=True)
    assert r.returncode == 0
    assert bed == open(tmp_path / "o.bed", "rb").read() and bcs == open(tmp_path / "o.bc", "rb").read()
    assert bcs.count(b"\n") == bed.count(b"\n") > 10 and all(l.startswith(b"1\t?:") for l in bcs.splitlines())


@pytest.mark.gpu
def test_product_barcode_errors_and_cli(gpu_ctx, tmp_path):
    import regtools_amd
    from regtools_amd import synth
    # a CB tag that is not a string: the reference builds a std::string from NULL and dies; the product reports it
    bam = str(tmp_path / "bad.bam")
    recs = [bamio.record(0, 100 + k, "20M200N30M", qname="r%d" % k, aux=bamio.tagA("XS", "+") + (b"CBC\x07" if k == 5 else bamio.tagZ("CB", "AAAC-1"))) for k in range(10)]
    bamio.write_bam(bam, [("chrZ", 100000)], recs)
    synth.index(bam)
    je = regtools_amd.JunctionsExtractor(bam=bam, strandness=0, ctx=gpu_ctx, output_barcodes_file=str(tmp_path / "x.bc"))
    with pytest.raises(regtools_amd.RegtoolsError):
        je.identify_junctions_from_BAM()
    je = regtools_amd.JunctionsExtractor(bam=bam, strandness=0, ctx=gpu_ctx)          # without -b the tag is never looked at
    je.identify_junctions_from_BAM()
    assert je.get_barcodes() == [[] for _ in je.get_all_junctions()] and je.barcodes_text(False) == b"0\t\n" * len(je.get_all_junctions())
    good = os.path.join(bc.GOLD, "few.bam")
    # a different tag name (barcode_tag_, junctions_extractor.h:182): UB is on ~30 % of the reads
    je = regtools_amd.JunctionsExtractor(bam=good, strandness=0, ctx=gpu_ctx, output_barcodes_file="x", barcode_tag="UB")
    je.identify_junctions_from_BAM()
    assert all({b for b, _ in m} <= {b"ACGTACGTAC", b"?"} for m in je.get_barcodes())
    # the command-line face
    exe = os.path.join(ROOT, "bin", "regtools-amd")
    args = bc.ARGS["pools"][1]
    r = subprocess.run([exe, "junctions", "extract"] + args + ["-o", str(tmp_path / "c.bed"), "-b", str(tmp_path / "c.bc"), os.path.join(bc.GOLD, "pools.bam")], capture_output=True)
    assert r.returncode == 0 and b"Barcode file: " in r.stderr
    assert open(tmp_path / "c.bed", "rb").read() == golden("pools", args, "bed") and open(tmp_path / "c.bc", "rb").read() == golden("pools", args, "barcodes")
    g = os.path.join(ROOT, "tests", "golden", "cse_ref")
    r = subprocess.run([exe, "cis-splice-effects", "identify", "-s", "RF", "-o", str(tmp_path / "i.tsv"), "-j", str(tmp_path / "i.bed"), "-b", str(tmp_path / "i.bc"),
                        os.path.join(g, "test1.vcf"), os.path.join(g, "test_hcc1395.2.bam"), os.path.join(g, "test_chr22.fa"), os.path.join(g, "test_ensemble_chr22.2.gtf")], capture_output=True)
    assert r.returncode == 0
    n = open(tmp_path / "i.bed").read().count("\n")
    assert n > 0 and open(tmp_path / "i.bc").read() == "0\t\n" * n


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", ALL, ids=IDS)
def test_barcodes_across_shards_equal_the_reference(gpu_ctx, tmp_path, name, args):
    """-b with the file cut into shards (rgx_extract_multi; rgx_table_merge_barcodes): a junction's barcode map only depends on the order in
    which distinct barcodes first reach it, shard order is file order -- the two output files must be the reference's for any shard count."""
    import regtools_amd
    je = regtools_amd.JunctionsExtractor(ctx=gpu_ctx)
    je.parse_options(list(args) + ["-b", str(tmp_path / "x.bc"), os.path.join(bc.GOLD, name + ".bam")])
    kw = dict(strandness=je.strandness_, strand_tag=je.strand_tag_, min_anchor_length=je.min_anchor_length_, min_intron_length=je.min_intron_length_,
              max_intron_length=je.max_intron_length_, region=je.region_, output_barcodes_file="x", barcode_tag=je.barcode_tag_)
    for devs in ([0, 0], [0, 0, 0, 0, 0]):
        m = regtools_amd.extract_multi(devs, bam=os.path.join(bc.GOLD, name + ".bam"), **kw)
        assert m.bed12() == golden(name, args, "bed"), devs
        assert m.barcodes_text() == golden(name, args, "barcodes"), devs


@pytest.mark.gpu
def test_barcodes_across_shards_of_a_larger_file(gpu_ctx, tmp_path):
    """Shard cuts that really split junctions' supporting reads: a synthetic file with CB tags from a pool, 1 / 2 / 7 shards, host merge
    (rgx_table_merge over shard tables) and rgx_extract_multi; all equal the single-shard output, which equals the oracle's."""
    import regtools_amd
    from regtools_amd import distributed
    rnd = random.Random(5)
    pool = ["".join(rnd.choice("ACGT") for _ in range(16)) + "-1" for _ in range(400)]
    recs = []
    for k in range(60000):
        j = rnd.randrange(40)
        aux = bamio.tagA("XS", "+-"[j % 2]) + (bamio.tagZ("CB", rnd.choice(pool)) if rnd.random() < 0.9 else b"")
        recs.append((1000 + 50 * j + rnd.randrange(30), bamio.record(0, 1000 + 50 * j + rnd.randrange(30), "%dM%dN%dM" % (20 + rnd.randrange(10), 300 + 10 * j, 25), qname="q%05d" % k, aux=aux)))
    recs.sort(key=lambda r: r[0])
    bam = str(tmp_path / "cb.bam")
    bamio.write_bam(bam, [("chrC", 1000000)], [r for _, r in recs], block=3000)
    from regtools_amd import synth
    synth.index(bam)
    bed, bcs, _ = gpu_extract_b(gpu_ctx, bam, ["-s", "XS"], tmp_path)
    r = subprocess.run([ORACLE, "extract", "-s", "XS", "-o", str(tmp_path / "o.bed"), "-b", str(tmp_path / "o.bc"), bam], capture_output=True)
    assert r.returncode == 0 and bed == open(tmp_path / "o.bed", "rb").read() and bcs == open(tmp_path / "o.bc", "rb").read()
    assert max(int(l.split(b"\t")[0]) for l in bcs.splitlines()) > 30
    for devs in ([0, 0], [0] * 7):
        m = regtools_amd.extract_multi(devs, bam=bam, strandness=0, output_barcodes_file="x")
        assert m.bed12() == bed and m.barcodes_text() == bcs, devs
    # the host merge of shard tables (what a multi-process driver without a device merge would call)
    parts = []
    for g in range(3):
        je = regtools_amd.JunctionsExtractor(bam=bam, strandness=0, ctx=gpu_ctx, output_barcodes_file="x", shard=g, n_shards=3)
        je.identify_junctions_from_BAM()
        parts.append(je)
    L = regtools_amd._ffi.lib()
    ptrs = (ctypes.POINTER(regtools_amd._ffi.JunctionTable) * 3)(*[p.table for p in parts])
    out = ctypes.POINTER(regtools_amd._ffi.JunctionTable)()
    err = ctypes.create_string_buffer(256)
    assert L.rgx_table_merge(ptrs, 3, 8, ctypes.byref(out), err, len(err)) == 0, err.value
    m = distributed.MergedTable(out)
    assert m.bed12() == bed and m.barcodes_text() == bcs
    # the same over the wire format of the one-process-per-GPU driver: every shard's packed rows and its barcode block as bytes
    # (rgx_table_pack / rgx_table_pack_barcodes), unpacked and merged where they arrive (distributed.merge_packed)
    wire = [(distributed.pack_table(p.table), distributed.pack_barcodes(p.table)) for p in parts]
    assert all(len(b) >= 40 for _, b in wire) and any(len(b) > 1000 for _, b in wire)
    m2 = distributed.merge_packed([w for w, _ in wire], parts[0].table, 8, None, [b for _, b in wire])
    assert m2.bed12() == bed and m2.barcodes_text() == bcs
    # a damaged block is refused, not trusted
    bad = bytearray(wire[1][1]); bad[30] ^= 0x40
    with pytest.raises(RuntimeError):
        distributed.merge_packed([w for w, _ in wire], parts[0].table, 8, None, [wire[0][1], bytes(bad), wire[2][1]])
    # ... and through a process group (nccl = RCCL, one rank): rows merged on the device, barcode lists gathered behind them
    import socket
    import torch.distributed as dist
    je1 = regtools_amd.JunctionsExtractor(bam=bam, strandness=0, ctx=gpu_ctx, output_barcodes_file="x")
    je1.identify_junctions_from_BAM()
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        m3 = distributed.gather_and_merge(je1, min_anchor=8)
        assert m3.bed12() == bed and m3.barcodes_text() == bcs
    finally:
        dist.destroy_process_group()
