"""GPU parity tests proper: the HIP path, called through the C ABI, against the reference's golden files, the
reference outputs stored by tests/golden/make_golden.py, and the CPU oracle on seeded inputs.  Bit-exact: every
comparison is on BED12 bytes.  Run with `-m gpu` on an MI355X."""
import ctypes as C
import json
import os
import subprocess
import zlib

import pytest

import bamio
import cases
from conftest import run_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def synth_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("synth")


def gpu_extract(ctx, bam, args, after=(), **kw):
    """(rc, bed12 bytes) the way `regtools junctions extract <args> bam [after]` would produce them."""
    import regtools_amd
    je = regtools_amd.JunctionsExtractor(ctx=ctx, **kw)
    try:
        je.parse_options(list(args) + [bam] + list(after))
        je.identify_junctions_from_BAM()
    except regtools_amd.RegtoolsError as e:
        return (0 if e.code == 0 else 1), b"", je
    return 0, je.bed12(), je


@pytest.mark.parametrize("args,golden", cases.REF_GOLDENS, ids=[g for _, g in cases.REF_GOLDENS])
def test_reference_integration_goldens(gpu_ctx, args, golden):
    rc, out, _ = gpu_extract(gpu_ctx, os.path.join(cases.GOLD, "test_hcc1395.bam"), args)
    assert rc == 0
    assert out == open(os.path.join(cases.GOLD, "junctions-extract", golden), "rb").read()


@pytest.mark.parametrize("case", cases.MANIFEST, ids=[c["name"] for c in cases.MANIFEST])
def test_equals_reference_outputs(gpu_ctx, case, synth_dir):
    argv = cases.case_argv(case, synth_dir)
    rc, out, _ = gpu_extract(gpu_ctx, argv[len(case["args"])], case["args"], after=argv[len(case["args"]) + 1:])
    assert rc == case["rc"]
    assert out == cases.expected(case)


def test_error_contract(gpu_ctx, tmp_path):
    import regtools_amd
    je = regtools_amd.JunctionsExtractor(bam="does_not_exist.bam", strandness=0, ctx=gpu_ctx)
    with pytest.raises(regtools_amd.RegtoolsError) as e:
        je.identify_junctions_from_BAM()
    # junctions_extractor.cc:505, behind the line htslib writes first (hts.c:408-411; the reference's stderr: tests/golden/cli/cli_vcf_notes_streams.json)
    assert str(e.value) == "[E::hts_open_format] fail to open file 'does_not_exist.bam'\nUnable to open BAM/SAM file.\n\n"
    p = tmp_path / "noidx.bam"
    p.write_bytes(open(os.path.join(cases.GOLD, "strand.bam"), "rb").read())
    je = regtools_amd.JunctionsExtractor(bam=str(p), strandness=0, ctx=gpu_ctx)
    with pytest.raises(regtools_amd.RegtoolsError) as e:
        je.identify_junctions_from_BAM()
    assert str(e.value) == "Unable to open BAM/SAM index. Make sure alignments are indexed\n\n"   # cc:510-511
    je = regtools_amd.JunctionsExtractor(bam=os.path.join(cases.GOLD, "contigs.bam"), region="nope:1-5", strandness=0, ctx=gpu_ctx)
    with pytest.raises(regtools_amd.RegtoolsError) as e:
        je.identify_junctions_from_BAM()
    assert str(e.value) == "Unable to iterate to region within BAM.\n\n"             # cc:521
    p = tmp_path / "garbage.bam"
    p.write_bytes(b"not a bam at all" * 10)
    (tmp_path / "garbage.bam.bai").write_bytes(open(os.path.join(cases.GOLD, "strand.bam.bai"), "rb").read())
    je = regtools_amd.JunctionsExtractor(bam=str(p), strandness=0, ctx=gpu_ctx)
    with pytest.raises(regtools_amd.RegtoolsError):
        je.identify_junctions_from_BAM()


@pytest.mark.parametrize("shape,n,seed", [("short", 200000, 101), ("short", 1000000, 102), ("long", 3000, 103), ("fuzz", 150000, 104), ("fuzz", 150000, 105)])
def test_against_oracle_on_seeded_inputs(gpu_ctx, synth_dir, shape, n, seed):
    from regtools_amd import synth
    p = os.path.join(str(synth_dir), "o_%s_%d.bam" % (shape, seed))
    synth.write(p, n, shape=shape, seed=seed)
    for args in (["-s", "XS"], ["-s", "RF", "-a", "3"], ["-s", "FR", "-m", "90", "-M", "20000"]):
        rc, out, je = gpu_extract(gpu_ctx, p, args)
        orc, exp, _ = run_oracle(args + [p])
        assert rc == orc == 0
        assert out == exp, (shape, seed, args)
        assert je.stats["n_records"] == n


def test_realistic_payload_and_every_zlib_level(gpu_ctx, synth_dir):
    # different DEFLATE shapes (stored blocks at level 0, fixed/dynamic mixes) must decode identically
    from regtools_amd import synth
    for level, realistic in ((1, True), (6, True), (9, False), (1, False)):
        p = os.path.join(str(synth_dir), "lvl%d_%d.bam" % (level, realistic))
        synth.write(p, 60000, shape="short", seed=7, level=level, realistic=realistic)
        rc, out, _ = gpu_extract(gpu_ctx, p, ["-s", "XS"])
        assert rc == 0 and out == run_oracle(["-s", "XS", p])[1]


def test_stored_deflate_blocks(gpu_ctx, tmp_path):
    # level-0 members are pure stored blocks (BTYPE 00): hand-made file
    recs = [bamio.record(0, 100 + 50 * k, "%dM%dN%dM" % (20 + k % 5, 100 + k, 30), qname="s%03d" % k, aux=bamio.tagA("XS", "+")) for k in range(500)]
    p = str(tmp_path / "stored.bam")
    bamio.write_bam(p, [("chrZ", 1000000)], recs, level=0, block=3000)
    from regtools_amd import synth
    synth.index(p)
    rc, out, _ = gpu_extract(gpu_ctx, p, ["-s", "XS"])
    assert rc == 0 and out == run_oracle(["-s", "XS", p])[1] and out.count(b"\n") > 100


def test_shard_merge_is_independent_of_shard_count(gpu_ctx, synth_dir):
    # member-range shards cut at index offsets; merged output must not depend on G
    from regtools_amd import synth, distributed
    for shape, n, seed in (("short", 300000, 201), ("fuzz", 100000, 202)):
        p = os.path.join(str(synth_dir), "shard_%s.bam" % shape)
        synth.write(p, n, shape=shape, seed=seed)
        _, single, je1 = gpu_extract(gpu_ctx, p, ["-s", "XS"])
        for G in (2, 3, 8):
            parts, recs, keep = [], 0, []
            for g in range(G):
                rc, _, je = gpu_extract(gpu_ctx, p, ["-s", "XS"], shard=g, n_shards=G)
                assert rc == 0
                keep.append(je)
                parts.append(distributed.pack_table(je.table))
                recs += je.stats["n_records"]
            assert recs == n
            merged = distributed.merge_packed(parts, keep[0].table, 8)
            assert merged.bed12() == single, (shape, G)
            # the same merge with the packed rows resident in HBM, as the RCCL all-gather leaves them (rgx_table_merge_device)
            import torch
            stride = max(1, max(k for _, k in parts))
            big = torch.zeros(G * stride * distributed.ROW, dtype=torch.uint8, device="cuda")
            for g, (b, k) in enumerate(parts):
                if k:
                    big[g * stride * distributed.ROW: g * stride * distributed.ROW + len(b)].copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
            torch.cuda.synchronize()
            dmerged = distributed.merge_device(gpu_ctx, big.data_ptr(), stride, [k for _, k in parts], keep[0].table, 8)
            assert dmerged.bed12() == single, (shape, G, "device merge")
            assert dmerged.bed12(False) == merged.bed12(False), (shape, G, "device merge, all rows")


def test_cli_binary_writes_identical_file(gpu_ctx, tmp_path):
    exe = os.path.join(ROOT, "bin", "regtools-amd")
    out = str(tmp_path / "o.bed")
    bam = os.path.join(cases.GOLD, "test_hcc1395.bam")
    r = subprocess.run([exe, "junctions", "extract", "-s", "RF", "-a", "30", "-o", out, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0
    assert open(out, "rb").read() == open(os.path.join(cases.GOLD, "junctions-extract", "expected-stranded-a30.out"), "rb").read()
    r = subprocess.run([exe, "junctions", "extract", "-s", "XS", "-r", "1:22405013-22405020", bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == open(os.path.join(cases.GOLD, "junctions-extract", "expected-r1:22405013-22405020.out"), "rb").read()
    assert subprocess.run([exe, "junctions", "extract", "-s", "XS", "missing.bam"], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 1


FORMS = [(1, "lane"), (2, "wave"), (3, "ring"), (4, "coop"), (5, "lane4")]       # rgx_k_inflate_form: k_inflate, k_inflate_wave, k_inflate_ring, k_inflate_coop, k_inflate<.., 4 literals per trip>


@pytest.mark.parametrize("form", [f for f, _ in FORMS], ids=[n for _, n in FORMS])
def test_inflate_kernel_against_zlib(gpu_ctx, form):
    # the DEFLATE kernels in isolation, through the C-ABI stage entry point
    import torch
    from regtools_amd import _ffi, synth
    bam, _, _ = synth.generate(40000, shape="short", seed=31, realistic=True)
    members, upos, expect = [], 0, []
    for off, payload, isize in bamio.bgzf_members(bam):
        members.append((off + 18, upos, len(payload), isize))
        expect.append(zlib.decompress(payload, -15))
        upos += isize
    arr = (_ffi.Member * len(members))(*[_ffi.Member(*m) for m in members])
    d_comp = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda")
    d_comp[: len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8))
    d_mem = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    d_arena = torch.zeros(upos + 512, dtype=torch.uint8, device="cuda")           # (the ring form reads up to 15 bytes in front of a member)
    d_status = torch.tensor([0xffffffff, 0], dtype=torch.int64).to(torch.uint32).cuda() if hasattr(torch, "uint32") else None
    if d_status is None:
        pytest.skip("torch without uint32")
    torch.cuda.synchronize()
    rc = _ffi.lib().rgx_k_inflate_form(form, d_comp.data_ptr(), d_mem.data_ptr(), len(members), d_arena.data_ptr() + 256, d_status.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0
    assert d_status.cpu().tolist()[0] == 0xffffffff
    assert bytes(d_arena[256:256 + upos].cpu().numpy().tobytes()) == b"".join(expect)


def test_coop_form_refuses_a_member_list_out_of_arena_order(gpu_ctx):
    """k_inflate_coop addresses the arena with 32-bit offsets from a group's first member: a caller's own list whose members do not lie in the
    arena in list order is refused by the stage entry point (status = first offending member, INF_OUT_OVERFLOW) and nothing is written;
    the lane form takes the same list."""
    import torch
    from regtools_amd import _ffi, synth
    bam, _, _ = synth.generate(3000, shape="short", seed=5)
    members, upos, expect = [], 0, []
    for off, payload, isize in bamio.bgzf_members(bam):
        members.append([off + 18, upos, len(payload), isize])
        expect.append(zlib.decompress(payload, -15))
        upos += isize
    assert len(members) >= 3
    # swap the places of the first two members in the arena (the list keeps its order)
    members[0][1], members[1][1] = members[1][3], 0
    want = expect[1] + expect[0] + b"".join(expect[2:])
    arr = (_ffi.Member * len(members))(*[_ffi.Member(*m) for m in members])
    d_comp = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda")
    d_comp[: len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8))
    d_mem = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    for form, ok in ((4, False), (1, True)):
        d_arena = torch.zeros(upos + 512, dtype=torch.uint8, device="cuda")
        d_status = torch.tensor([0xffffffff, 0], dtype=torch.int64).to(torch.uint32).cuda()
        torch.cuda.synchronize()
        assert _ffi.lib().rgx_k_inflate_form(form, d_comp.data_ptr(), d_mem.data_ptr(), len(members), d_arena.data_ptr() + 256, d_status.data_ptr(), None) == 0
        torch.cuda.synchronize()
        st = d_status.cpu().tolist()
        got = bytes(d_arena[256:256 + upos].cpu().numpy().tobytes())
        if ok:
            assert st[0] == 0xffffffff and got == want
        else:
            assert st == [1, 10] and got == bytes(upos), st


def test_multi_million_read_properties(gpu_ctx, synth_dir):
    """Size-independent properties at a multi-million-read scale (the 50M-read identity check lives in bench.py)."""
    from regtools_amd import synth
    n = 5_000_000
    bam, bai, st = synth.generate(n, shape="short", seed=1)
    import regtools_amd
    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=gpu_ctx)
    je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
    rows = je.get_all_junctions()
    assert je.stats["n_records"] == n
    assert sum(j.read_count for j in rows) == je.stats["n_events"] == st["n_spliced"]     # every spliced read has one N in range
    assert sorted(int(j.name[4:]) for j in rows) == list(range(1, len(rows) + 1))         # names are a permutation of 1..J
    keys = [(j.chrom, j.thick_start, j.thick_end, j.name) for j in rows]
    assert keys == sorted(keys)                                                           # compare_junctions order
    assert all(j.thick_start <= j.start < j.end <= j.thick_end for j in rows)
    first = je.bed12()
    je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)                          # idempotent on a warm workspace
    assert je.bed12() == first


def test_extract_with_fasta_strand_rule(gpu_ctx, tmp_path):
    """`junctions extract ... ref.fa`: intron-motif first, tag/flag rule when the motif is not canonical (junctions_extractor.cc:345-359)."""
    import cse_synth
    q = cse_synth.build(str(tmp_path / "q"), seed=5, n_genes=14)
    for args in (["-s", "XS"], ["-s", "intron-motif"], ["-s", "RF", "-a", "3"]):
        import regtools_amd
        je = regtools_amd.JunctionsExtractor(ctx=gpu_ctx)
        je.parse_options(args + [q["bam"], q["fasta"]])
        je.identify_junctions_from_BAM()
        rc, exp, _ = run_oracle(args + [q["bam"], q["fasta"]])
        assert rc == 0 and je.bed12() == exp and exp.count(b"\n") > 20, args
    # long reads through the wave-per-read kernel with the carried-over strand state
    from regtools_amd import synth
    p = str(tmp_path / "long.bam")
    synth.write(p, 400, shape="long", seed=9)
    # a FASTA for the human contig names would be 3 GB; a missing contig must fail like the reference (exit 1)
    je = regtools_amd.JunctionsExtractor(ctx=gpu_ctx)
    je.parse_options(["-s", "XS", p, q["fasta"]])
    with pytest.raises(regtools_amd.RegtoolsError):
        je.identify_junctions_from_BAM()


@pytest.mark.parametrize("variant,max_sweeps", [("huge", 3), ("to_end", 24), ("insane", 24)])
def test_record_framing_converges_quickly_on_decoys(gpu_ctx, synth_dir, variant, max_sweeps):
    """The speculative framing is exact whatever it guessed; what decoys may cost is verification sweeps.  A wrong exit must be
    repaired where it happened (sweeps ~ longest run of mispredicted segments), not chased to the end of the file, and a chain
    that really ends in the middle of the file must not be walked one segment per sweep."""
    case = [c for c in cases.MANIFEST if c["name"] == "framing_%s.XS" % variant][0]
    rc, out, je = gpu_extract(gpu_ctx, cases.case_bam(case, synth_dir), case["args"])
    assert rc == 0 and out == cases.expected(case)
    assert 1 <= je.stats["framing_sweeps"] <= max_sweeps, je.stats


def test_rccl_gather_and_device_merge_single_rank(gpu_ctx, synth_dir):
    """The bench's N > 1 path (one all-gather over the nccl = RCCL backend, then rgx_table_merge_device) with a process group of one:
    the merged table of a single shard is the shard's own table."""
    import socket
    import torch.distributed as dist
    from regtools_amd import synth, distributed
    p = os.path.join(str(synth_dir), "rccl1.bam")
    synth.write(p, 200000, shape="short", seed=77)
    rc, single, je = gpu_extract(gpu_ctx, p, ["-s", "XS"])
    assert rc == 0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        merged = distributed.gather_and_merge(je, min_anchor=8)
        assert merged.bed12() == single and merged.n == je.table.contents.n
    finally:
        dist.destroy_process_group()


def test_record_framing_runs_of_startless_segments_resolve_in_parallel(gpu_ctx, synth_dir):
    """Thirty reads of up to 250 kb (up to 23 framing segments in which no record starts) and one read with 12,001 CIGAR operations.
    The runs of start-less segments must resolve side by side (sweeps ~ longest run), not one after the other (sweeps ~ their sum,
    about 300 here)."""
    case = [c for c in cases.MANIFEST if c["name"] == "ultralong.XS"][0]
    rc, out, je = gpu_extract(gpu_ctx, cases.case_bam(case, synth_dir), case["args"])
    assert rc == 0 and out == cases.expected(case)
    assert 1 <= je.stats["framing_sweeps"] <= 48, je.stats


@pytest.mark.parametrize("form", [f for f, _ in FORMS], ids=[n for _, n in FORMS])
def test_inflate_kernel_on_adversarial_members(gpu_ctx, form):
    """The DEFLATE kernel alone on members built to hit its corners: runs (distance 1..15 with distance doubling), maximum-length matches,
    incompressible bytes (stored blocks), fixed-Huffman blocks, several blocks per member (Z_FULL_FLUSH), one-byte and empty-ish members,
    every zlib level and strategy, and 64 lanes of a wave that each see a different kind of stream.  Compared with zlib byte for byte;
    the arena is checked beyond every member's end (nothing may be written there)."""
    import random
    import torch
    from regtools_amd import _ffi
    rnd = random.Random(99)
    datas = []
    for k in range(260):
        kind = k % 13
        n = rnd.choice([1, 2, 15, 16, 17, 255, 256, 4000, 30000, 65280])
        if kind == 0: d = bytes([rnd.randrange(256)]) * n
        elif kind == 1: d = (bytes(rnd.randrange(256) for _ in range(rnd.randint(2, 15))) * (n // 2 + 1))[:n]
        elif kind == 2: d = bytes(rnd.randrange(256) for _ in range(n))
        elif kind == 3: d = bytes(rnd.choice(b"ACGT") for _ in range(n))
        elif kind == 4: d = (bytes(rnd.randrange(256) for _ in range(300)) * (n // 300 + 1))[:n]
        elif kind == 5: d = b"".join(bytes([rnd.randrange(256)]) * rnd.randint(1, 600) for _ in range(n // 100 + 1))[:n]
        elif kind == 6: d = bytes((i * 7 + (i >> 8)) & 0xff for i in range(n))
        elif kind == 7: d = (b"\x11" * 51 + b"\xff" * 101 + bytes(rnd.randrange(256) for _ in range(12))) * (n // 164 + 1)
        elif kind == 8: d = bytes(rnd.choice(b"ab") for _ in range(n))
        elif kind == 9: d = b"\0" * n
        elif kind == 10: d = bytes(rnd.randrange(4) for _ in range(n))
        elif kind == 11: d = (bytes(rnd.randrange(256) for _ in range(16)) * (n // 16 + 1))[:n]
        else: d = bytes(rnd.randrange(256) if rnd.random() < 0.1 else 65 for _ in range(n))
        datas.append(d[:n] if len(d) >= n else d)
    members, expect, blob, upos = [], [], bytearray(), 0
    for k, d in enumerate(datas):
        level = [0, 1, 6, 9][k % 4]
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][k % 5]
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        if k % 7 == 3 and len(d) > 10:
            payload = c.compress(d[: len(d) // 3]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(d[len(d) // 3:]) + c.flush()
        else:
            payload = c.compress(d) + c.flush()
        if len(payload) > 65000:
            continue
        members.append((len(blob), upos, len(payload), len(d)))
        blob += payload + bytes(rnd.randrange(256) for _ in range(rnd.randint(8, 40)))     # footer-like bytes between payloads
        expect.append(d)
        upos += len(d) + (k % 3) * 5                                                             # gaps: bytes no member owns
    total = upos + 64
    arr = (_ffi.Member * len(members))(*[_ffi.Member(*m) for m in members])
    d_comp = torch.zeros(len(blob) + 64, dtype=torch.uint8, device="cuda")
    d_comp[: len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    d_mem = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    d_arena = torch.full((256 + total + 256,), 0xA5, dtype=torch.uint8, device="cuda")
    d_status = torch.tensor([0xffffffff, 0], dtype=torch.int64).to(torch.uint32).cuda()
    torch.cuda.synchronize()
    rc = _ffi.lib().rgx_k_inflate_form(form, d_comp.data_ptr(), d_mem.data_ptr(), len(members), d_arena.data_ptr() + 256, d_status.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0 and d_status.cpu().tolist()[0] == 0xffffffff
    got = d_arena.cpu().numpy().tobytes()
    want = bytearray(b"\xa5" * (256 + total + 256))
    for (cpos, up, clen, isz), d in zip(members, expect):
        want[256 + up: 256 + up + isz] = d
    assert got == bytes(want)


def test_device_side_row_packing_equals_host_packing(gpu_ctx, synth_dir):
    """rgx_last_table_pack_device: the packed rows of the last extraction, written in HBM, are the bytes rgx_table_pack makes on the host;
    a table that is not the context's last result is refused (the caller then packs on the host)."""
    import torch
    from regtools_amd import _ffi, synth, distributed
    L = _ffi.lib()
    p1 = os.path.join(str(synth_dir), "pack1.bam"); synth.write(p1, 120000, shape="short", seed=81)
    p2 = os.path.join(str(synth_dir), "pack2.bam"); synth.write(p2, 50000, shape="fuzz", seed=82)
    rc, _, je1 = gpu_extract(gpu_ctx, p1, ["-s", "XS"])
    assert rc == 0
    host_bytes, n = distributed.pack_table(je1.table)
    dev = torch.zeros((n + 5) * distributed.ROW, dtype=torch.uint8, device="cuda")
    err = C.create_string_buffer(256)
    assert L.rgx_last_table_pack_device(gpu_ctx._h, je1.table, C.c_void_p(dev.data_ptr()), n + 5, err, len(err)) == 0, err.value
    assert dev[: n * distributed.ROW].cpu().numpy().tobytes() == host_bytes and n > 1000
    assert L.rgx_last_table_pack_device(gpu_ctx._h, je1.table, C.c_void_p(dev.data_ptr()), n - 1, err, len(err)) != 0      # too small
    rc, _, je2 = gpu_extract(gpu_ctx, p2, ["-s", "XS"])
    assert rc == 0
    assert L.rgx_last_table_pack_device(gpu_ctx._h, je1.table, C.c_void_p(dev.data_ptr()), n + 5, err, len(err)) != 0      # stale table
    assert b"not the result of the last extraction" in err.value


def test_full_size_properties(gpu_ctx):
    """BASELINE configs[1] at its real size (50 M reads; the oracle would need minutes here, bench.py does that comparison against the
    reference itself): properties that do not depend on the size -- determinism, conservation of events, output order, naming, and
    independence of the shard count."""
    from regtools_amd import synth, distributed
    bam, bai, st = synth.generate(50_000_000, shape="short", seed=1)
    import regtools_amd
    def run(**kw):
        je = regtools_amd.JunctionsExtractor(strandness=0, ctx=gpu_ctx, **kw)
        je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
        return je
    a, b = run(), run()
    bed = a.bed12()
    assert bed == b.bed12() and a.stats["n_records"] == st["n_reads"] == 50_000_000
    rows = a.get_all_junctions()
    assert sum(j.read_count for j in rows) == a.stats["n_events"] > 5_000_000
    assert sorted(int(j.name[4:]) for j in rows) == list(range(1, len(rows) + 1))             # first-seen names: a permutation of 1..n
    keys = [(j.chrom, j.thick_start, j.thick_end, j.name) for j in rows]
    assert keys == sorted(keys)                                                               # compare_junctions (h:117-140)
    assert len({(j.chrom, j.start, j.end, j.strand in "+-" and j.strand) for j in rows}) == len(rows)
    parts, keep, recs = [], [], 0
    for g in range(2):
        je = run(shard=g, n_shards=2)
        keep.append(je); parts.append(distributed.pack_table(je.table)); recs += je.stats["n_records"]
    assert recs == 50_000_000
    assert distributed.merge_packed(parts, keep[0].table, 8).bed12() == bed


def test_region_queries_read_only_the_members_the_index_names(gpu_ctx, synth_dir):
    """-r: the member range comes from the index's bins + linear index (hts_itr_query, hts.c:1733-1800); the rows must be the oracle's
    for any region, and a small region must not inflate the whole file."""
    import random
    from regtools_amd import synth
    p = os.path.join(str(synth_dir), "region_idx.bam")
    synth.write(p, 400000, shape="short", seed=77)
    _, _, whole = gpu_extract(gpu_ctx, p, ["-s", "XS"])
    rng = random.Random(5)
    regions = ["chr1:1-50000", "chr1:100000-100001", "chr2:5000000-9000000", "chr7", "chrX:1-1000", "chr1:200000000-249000000", "chr22:1-60000000",
               "chr3:16384-16385", "chr3:16383-32769", "chrY:10000000-10000100"]
    for _ in range(12):
        c = rng.choice(["chr1", "chr2", "chr5", "chr11", "chr19", "chrX"])
        b = rng.randrange(1, 150_000_000); regions.append("%s:%d-%d" % (c, b, b + rng.choice([1, 100, 20000, 3_000_000, 80_000_000])))
    small = 0
    for reg in regions:
        for args in (["-s", "XS", "-r", reg], ["-s", "RF", "-a", "5", "-r", reg]):
            rc, out, je = gpu_extract(gpu_ctx, p, args)
            orc, exp, _ = run_oracle(args + [p])
            assert rc == orc and out == exp, (reg, args)
        small += rc == 0 and je.stats["inflated_bytes"] * 4 < whole.stats["inflated_bytes"]
    assert small >= 12, small
    # the same through a .csi (no bin geometry this path reads: whole-file inflate, same rows) and with the file already in HBM
    import shutil, csi_common, torch
    q = os.path.join(str(synth_dir), "region_csi.bam")
    shutil.copy(p, q); shutil.copy(p + ".bai", q + ".bai"); csi_common.bai_to_csi(q)
    raw, bai = open(p, "rb").read(), open(p + ".bai", "rb").read()
    d = torch.zeros(len(raw) + 64, dtype=torch.uint8, device="cuda"); d[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8)); torch.cuda.synchronize()
    import regtools_amd
    for reg in regions[:6]:
        exp = run_oracle(["-s", "XS", "-r", reg, p])[1]
        assert gpu_extract(gpu_ctx, q, ["-s", "XS", "-r", reg])[1] == exp
        je = regtools_amd.JunctionsExtractor(strandness=0, region=reg, ctx=gpu_ctx)
        je.identify_junctions_from_BAM(bai_bytes=bai, device_ptr=d.data_ptr(), device_len=len(raw))
        assert je.bed12() == exp


def test_host_bytes_overlapped_upload_equals_device_resident_path(gpu_ctx):
    """rgx_extract_mem on a file large enough for the chunked upload (SURVEY 8d's timed region: host bytes -> table): members found by the
    host scan, one inflate launch per upload chunk on the side streams.  Same bytes as with the file resident in HBM (device member
    discovery, one launch), from page-locked and from pageable memory, whole file / region / shards; a file the host scan does not vouch
    for (cut inside a member) takes the device discovery after the upload and still equals the device-resident run."""
    import torch
    import regtools_amd
    from regtools_amd import synth, distributed
    bam, bai, st = synth.generate(3_000_000, shape="short", seed=11)
    assert len(bam) > (24 << 20)                                                    # at least two upload chunks
    d = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda"); d[:len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8)); torch.cuda.synchronize()
    pin = regtools_amd.PinnedBuffer(bam)

    def run(how, n=len(bam), **kw):
        je = regtools_amd.JunctionsExtractor(strandness=0, ctx=gpu_ctx, **kw)
        if how == "device": je.identify_junctions_from_BAM(bai_bytes=bai, device_ptr=d.data_ptr(), device_len=n)
        elif how == "pinned": je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=n)
        else: je.identify_junctions_from_BAM(bam_bytes=bam[:n], bai_bytes=bai)
        return je
    ref = run("device")
    assert ref.stats["n_records"] == 3_000_000 and ref.stats["n_events"] == st["n_spliced"]
    for how in ("pinned", "pageable", "pinned"):
        je = run(how)
        assert je.bed12() == ref.bed12() and je.stats["n_records"] == 3_000_000 and je.stats["inflated_bytes"] == ref.stats["inflated_bytes"], how
    for reg in ("chr2", "chr1:1000000-90000000", "chrX:1-500"):
        assert run("pinned", region=reg).bed12() == run("device", region=reg).bed12(), reg
    parts, keep = [], []
    for g in range(3):
        je = run("pinned", shard=g, n_shards=3)
        keep.append(je); parts.append(distributed.pack_table(je.table))
    assert distributed.merge_packed(parts, keep[0].table, 8).bed12() == ref.bed12()
    cut = len(bam) - 70000                                                          # inside a member: the record stream ends early (bgzf.c:421-546)
    a, b = run("device", n=cut), run("pinned", n=cut)
    assert a.bed12() == b.bed12() and a.stats["n_records"] == b.stats["n_records"] < 3_000_000
    pin.close()


def test_multi_device_host_equals_single_gpu(gpu_ctx, synth_dir):
    """rgx_extract_multi (the C++ host of junctions_main.cc:45-59 on a multi-GPU node): a host thread per device, one RCCL
    all-gather of the shards' packed rows, device merge.  With N >= 2 visible GPUs the collective runs on all of them; on a one-GPU box
    the same device is listed several times (the shards take turns, the exchange is a device copy) -- every other line is the same.
    The bytes must be the single-GPU bytes for any device list."""
    import torch
    import regtools_amd
    from regtools_amd import synth
    p = os.path.join(str(synth_dir), "multi.bam")
    synth.write(p, 600000, shape="short", seed=31)
    _, single, je = gpu_extract(gpu_ctx, p, ["-s", "XS"])
    n_gpu = torch.cuda.device_count()
    lists = [[0], [0, 0], [0, 0, 0, 0, 0]]
    if n_gpu >= 2:
        lists += [list(range(n_gpu)), list(range(n_gpu - 1, -1, -1))[:2]]
    from regtools_amd import _ffi
    kind = lambda: _ffi.lib().rgx_multi_exchange_kind().decode()
    for devs in lists:
        m = regtools_amd.extract_multi(devs, bam=p, strandness=0)
        assert m.bed12() == single, devs
        assert m.table.contents.n_records == je.stats["n_records"] and m.table.contents.n_events == je.stats["n_events"], devs
        # how the rows travelled: nothing to move, device copies when a device is listed twice, RCCL between distinct devices
        want = "none" if len(devs) == 1 else "device copies" if len(set(devs)) < len(devs) else "rccl grouped send/recv, %d ranks" % len(devs)
        assert kind().startswith(want), (devs, kind())
    if n_gpu >= 2:
        # the fall-back a node without a working RCCL takes: the same rows as peer copies, and the call says so
        os.environ["REGTOOLS_AMD_RCCL"] = "off"
        try:
            assert regtools_amd.extract_multi(list(range(n_gpu)), bam=p, strandness=0).bed12() == single
            assert kind().startswith("hipMemcpyPeerAsync") and "REGTOOLS_AMD_RCCL=off" in kind(), kind()
        finally:
            del os.environ["REGTOOLS_AMD_RCCL"]
    for args, kw in ((["-s", "RF", "-a", "20"], dict(strandness=1, min_anchor_length=20)), (["-s", "XS", "-r", "chr3"], dict(strandness=0, region="chr3"))):
        exp = gpu_extract(gpu_ctx, p, args)[1]
        assert regtools_amd.extract_multi(lists[-1], bam=p, **kw).bed12() == exp, args
    os.environ["REGTOOLS_AMD_RCCL"] = "selftest"           # one rank through ncclCommInitAll / ncclAllGather (librccl.so.1 loaded at run time) + device merge
    try:
        assert regtools_amd.extract_multi([0], bam=p, strandness=0).bed12() == single
    finally:
        del os.environ["REGTOOLS_AMD_RCCL"]
    raw, bai = open(p, "rb").read(), open(p + ".bai", "rb").read()
    assert regtools_amd.extract_multi([0, 0, 0], bam_bytes=raw, bai_bytes=bai, strandness=0).bed12() == single
    # the CLI: REGTOOLS_AMD_DEVICES shards the file, the output file is the same
    out = os.path.join(str(synth_dir), "multi.bed")
    env = dict(os.environ, REGTOOLS_AMD_DEVICES=",".join(str(d) for d in lists[-1]))
    r = subprocess.run([os.path.join(ROOT, "bin", "regtools-amd"), "junctions", "extract", "-s", "XS", "-o", out, p], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and open(out, "rb").read() == single
    # -b across shards (rgx_table_merge_barcodes; tests/test_barcodes.py has the files with real CB tags): no tag here, every map is {"?": count}
    jb = regtools_amd.JunctionsExtractor(bam=p, strandness=0, ctx=gpu_ctx, output_barcodes_file="x.tsv")
    jb.identify_junctions_from_BAM()
    mb = regtools_amd.extract_multi([0, 0, 0], bam=p, strandness=0, output_barcodes_file="x.tsv")
    assert mb.bed12() == single and mb.barcodes_text() == jb.barcodes_text() and mb.barcodes_text().count(b"\n") == single.count(b"\n")


def test_full_size_long_read_properties(gpu_ctx):
    """BASELINE configs[4] at its real size (10 M long reads, l_qseq 1000-10000, n_cigar <= 64, 5-20 N ops: 65 GB inflated; the reference
    needs minutes for it, bench.py compares a 200 k-read sample): the size-independent properties -- determinism, conservation of
    events through the wave-per-read emit kernel, output order, naming, and independence of the shard count."""
    from regtools_amd import synth, distributed
    bam, bai, st = synth.generate(10_000_000, shape="long", seed=1)
    import regtools_amd
    pin = regtools_amd.PinnedBuffer(bam)

    def run(**kw):
        je = regtools_amd.JunctionsExtractor(strandness=0, ctx=gpu_ctx, **kw)
        je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
        return je
    a, b = run(), run()
    bed = a.bed12()
    assert bed == b.bed12() and a.stats["n_records"] == st["n_reads"] == 10_000_000
    rows = a.get_all_junctions()
    assert sum(j.read_count for j in rows) == a.stats["n_events"] > 50_000_000             # 5-20 junction events per read
    assert sorted(int(j.name[4:]) for j in rows) == list(range(1, len(rows) + 1))             # first-seen names: a permutation of 1..n
    keys = [(j.chrom, j.thick_start, j.thick_end, j.name) for j in rows]
    assert keys == sorted(keys)                                                               # compare_junctions (h:117-140)
    assert all(j.thick_start <= j.start < j.end <= j.thick_end for j in rows)
    parts, keep, recs = [], [], 0
    for g in range(2):
        je = run(shard=g, n_shards=2)
        keep.append(je); parts.append(distributed.pack_table(je.table)); recs += je.stats["n_records"]
    assert recs == 10_000_000
    assert distributed.merge_packed(parts, keep[0].table, 8).bed12() == bed
    pin.close()


def test_bench_multi_rank_path_on_one_gpu(gpu_ctx, tmp_path):
    """bench.py's N > 1 path (what the driver launches on a multi-GPU node) with two ranks on THIS GPU over gloo (BENCH_BACKEND / BENCH_DEVICE):
    every rank extracts its coordinate slice, the packed rows are gathered and merged -- the table rank 0 reports must be the merge of the two
    slices' tables made here, one after the other.  And bench.py's C++ host mode (--multi-host cpp, rgx_extract_multi on the device list
    [0, 0]): the table of one file of twice the reads, sharded, must be that file's single-GPU table."""
    import json
    import sys
    from regtools_amd import synth
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bed2 = str(tmp_path / "ranks2.bed")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29617",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "200000", "--no-cpu-baseline", "--no-extras", "--dump-bed", bed2],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["multi_gpu"]["per_rank"][1]["reads"] == 200000 and line["multi_gpu"]["merge_ms"] > 0
    ck = line["multi_gpu"]["checks"]                 # the line's own assertions (bench.py): records, an independent host merge, supporting reads
    assert ck["records_conserved"] and ck["bed12_equals_independent_merge"] and ck["counts_conserved"] and ck["merged_rows"] == line["junction_rows"] == ck["independent_host_merge_rows"]
    assert sum(r["n_records"] for r in line["multi_gpu"]["per_rank"]) == 400000
    # round 6: every rank reports its host link with all ranks copying at once and its NUMA binding; the sustained pass (two files in flight per rank,
    # the collective and the merge under the next file's upload) ran and produced the timed step's table
    assert all(r["upload_GBps_all_ranks_at_once"] > 0 and isinstance(r["host_binding"], str) for r in line["multi_gpu"]["per_rank"])
    assert line["value_sustained"] > 0 and line["sustained"]["table_identical_to_the_timed_step"] and line["sustained"]["in_flight"] == 2
    # the same two slices, one rank each, merged here
    import regtools_amd
    from regtools_amd import distributed
    parts, keep = [], []
    for rk in range(2):
        bam, bai, _ = synth.generate(200000, shape="short", seed=1, slice_index=rk, n_slices=2)
        je = regtools_amd.JunctionsExtractor(strandness=0, ctx=gpu_ctx)
        je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
        keep.append(je); parts.append(distributed.pack_table(je.table))
    want = distributed.merge_packed(parts, keep[0].table, 8).bed12()
    assert open(bed2, "rb").read() == want
    # ... and bench.py's C++ host mode: one file of 2 x 200000 reads over the device list [0, 0] == the single-GPU table of that file
    bedc = str(tmp_path / "cpp.bed")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--multi-host", "cpp", "--steps", "1", "--warmup", "1", "--reads", "200000", "--dump-bed", bedc],
                       env=os.environ, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    linec = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert linec["multi_gpu"]["checks"]["bed12_equals_single_device"] and linec["multi_gpu"]["exchange"].startswith("device copies"), linec["multi_gpu"]
    bam, bai, _ = synth.generate(400000, shape="short", seed=1)
    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=gpu_ctx)
    je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
    assert open(bedc, "rb").read() == je.bed12()


def test_long_record_decode_lane_form_equals_the_wave_form(gpu_ctx, synth_dir):
    """k_decode_sparse (long records: one lane follows a segment's two or three records) against k_decode_seg<false> (a workgroup per segment,
    REGTOOLS_AMD_DECODE=wave), through the CLI: whole file, a region query (the end rule's stop / last-in reduction), intron limits; and
    against the oracle."""
    from regtools_amd import synth
    p = os.path.join(str(synth_dir), "sparse_long.bam")
    synth.write(p, 30000, shape="long", seed=77)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "regtools-amd")
    for args in (["-s", "XS"], ["-s", "RF", "-m", "200", "-M", "20000"], ["-s", "XS", "-r", "chr2:1000000-90000000"], ["-s", "FR", "-r", "chr1"]):
        outs = []
        for form, seg in (("lane", "16384"), ("wave", "16384"), ("lane", "131072")):      # (the last: the segment size files of long records get, api_records.cpp seg_bytes)
            o = p + "." + form + seg + ".bed"
            r = subprocess.run([exe, "junctions", "extract"] + args + ["-o", o, p], env=dict(os.environ, REGTOOLS_AMD_DECODE=form + "," + seg),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr
            outs.append(open(o, "rb").read())
        assert outs[0] == outs[1] == outs[2], args
        assert outs[0] == run_oracle(args + [p])[1], args
    assert len(outs[0]) > 1000


def test_arena_placement_trials_leave_the_results_alone(gpu_ctx):
    """Opt-in since round 6 (REGTOOLS_AMD_ARENA=5): a context that is not one-shot times its first large DEFLATE launch into fresh arenas, one at a time, and
    keeps a faster one (rgx_ctx_arena_trials).  The call that calibrates, the calls behind it (on the kept arena, possibly another one than the
    first call's data lies in) and a context with the default (no trials) must print the same bytes; the calibration happens once."""
    import subprocess
    import sys
    code = ("import sys, json, regtools_amd; from regtools_amd import synth\n"
            "bam, bai, st = synth.generate(10_000_000, shape='short', seed=21)\n"
            "ctx = regtools_amd.Context(0); beds, trials = [], []\n"
            "for k in range(3):\n"
            "    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)\n"
            "    je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)\n"
            "    assert je.stats['n_records'] == st['n_reads']\n"
            "    beds.append(je.bed12()); trials.append(ctx.arena_trials())\n"
            "ctx.close()\n"
            "assert beds[0] == beds[1] == beds[2] and len(beds[0]) > 1000\n"
            "sys.stderr.write('TRIALS ' + json.dumps(trials) + '\\n')\n"
            "sys.stdout.buffer.write(beds[0])\n")
    outs = {}
    for knob in ("5", None):
        env = dict(os.environ, PYTHONPATH=ROOT)
        env.pop("REGTOOLS_AMD_ARENA", None)
        if knob: env["REGTOOLS_AMD_ARENA"] = knob
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-8000:]
        trials = json.loads([l for l in r.stderr.decode().splitlines() if l.startswith("TRIALS ")][-1][7:])
        if knob:
            # the call's arena + one to five challengers (the first that wins ends the trials: never two challengers' memory at once), once per context
            assert 2 <= len(trials[0]) <= 6 and all(t > 0 for t in trials[0]), trials
            assert trials[0] == trials[1] == trials[2]
        else:
            assert trials == [[], [], []], trials                                      # the default: no trials
        outs[knob] = r.stdout
    assert outs["5"] == outs[None] and len(outs[None]) > 1000


def test_arena_made_of_pieces_grows_and_shrinks_like_a_block(gpu_ctx):
    """The arena's device memory is created in pieces and mapped side by side into one reserved address range (DevBuf::map_pieces; 512 MiB pieces
    by default, so only files of millions of reads take that form).  With 2 MiB pieces every file beyond a few thousand reads does: one context sees files of
    growing and shrinking size (the range is unmapped, released and reserved anew as the arena grows) and must print what a context with a hipMalloc arena prints."""
    import sys
    code = ("import sys, hashlib, regtools_amd; from regtools_amd import synth\n"
            "ctx = regtools_amd.Context(0)\n"
            "for n, seed in ((2_000, 3), (300_000, 4), (1_500_000, 5), (300_000, 4), (2_500_000, 6)):\n"
            "    bam, bai, st = synth.generate(n, shape='short', seed=seed)\n"
            "    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)\n"
            "    je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)\n"
            "    assert je.stats['n_records'] == st['n_reads']\n"
            "    print(n, len(je.bed12()), hashlib.sha256(je.bed12()).hexdigest())\n"
            "ctx.close()\n")
    outs = []
    for knob in ("0,2", "0,64", "0,0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, REGTOOLS_AMD_ARENA=knob, PYTHONPATH=ROOT), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (knob, r.stderr.decode()[-2000:])
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] and outs[0].count(b"\n") == 5, outs
