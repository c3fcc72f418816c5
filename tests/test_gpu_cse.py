"""GPU parity of `cis-splice-effects identify` through the C ABI: the reference's 2 x 3 goldens, the 42
outputs of the real reference on synthetic quartets (tests/golden/cse), and the stage entry points against the gtest known
answers.  Bit-exact on all three output files."""
import ctypes as C
import json
import os
import subprocess

import pytest

import cases
import cse_synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSE = os.path.join(cases.GOLD, "cse")
REF_GOLD = os.path.join(cases.GOLD, "cse_ref")
MANIFEST = json.load(open(os.path.join(CSE, "manifest.json")))
_quartets = {}


@pytest.fixture(scope="module")
def work(tmp_path_factory):
    return tmp_path_factory.mktemp("cse_gpu")


def quartet(seed, n_genes, tmp):
    if seed not in _quartets:
        _quartets[seed] = cse_synth.build(os.path.join(str(tmp), "s%d" % seed), seed=seed, n_genes=n_genes)
    return _quartets[seed]


def gpu_identify(ctx, args, q, prefix):
    import regtools_amd
    files = {x: "%s.%s" % (prefix, x) for x in ("tsv", "vcf", "bed")}
    ci = regtools_amd.CisSpliceEffectsIdentifier(ctx=ctx)
    try:
        ci.parse_options(list(args) + ["-o", files["tsv"], "-v", files["vcf"], "-j", files["bed"], q["vcf"], q["bam"], q["fasta"], q["gtf"]])
        ci.identify()
    except regtools_amd.RegtoolsError as e:
        return (0 if e.code == 0 else 1), files, ci, str(e)
    return 0, files, ci, ""


def uses_motif(args):
    return "-C" in args or "intron-motif" in args


@pytest.mark.parametrize("strand,name", [("XS", "default"), ("RF", "default-stranded")])
def test_reference_identify_goldens(gpu_ctx, strand, name, work):
    q = dict(vcf=os.path.join(REF_GOLD, "test1.vcf"), bam=os.path.join(REF_GOLD, "test_hcc1395.2.bam"),
             fasta=os.path.join(REF_GOLD, "test_chr22.fa"), gtf=os.path.join(REF_GOLD, "test_ensemble_chr22.2.gtf"))
    rc, files, ci, msg = gpu_identify(gpu_ctx, ["-s", strand], q, os.path.join(str(work), "ref_" + strand))
    assert rc == 0, msg
    for ext, gold in (("tsv", "annotatedjunctions"), ("vcf", "annotatedvariants"), ("bed", "junctions")):
        exp = open(os.path.join(REF_GOLD, "expected-cis-splice-effects-identify-%s-%s.out" % (name, gold)), "rb").read()
        assert open(files[ext], "rb").read() == exp, ext
    assert ci.stats["n_variants"] == 20 and ci.stats["n_relevant"] == 10 and ci.stats["n_junctions"] == 1


@pytest.mark.parametrize("case", MANIFEST, ids=[c["name"] for c in MANIFEST])
def test_equals_reference(gpu_ctx, case, work):
    q = quartet(case["seed"], case["n_genes"], work)
    rc, files, ci, msg = gpu_identify(gpu_ctx, case["args"], q, os.path.join(str(work), case["name"]))
    assert rc == case["rc"], msg
    for ext in ("tsv", "vcf", "bed"):
        assert open(files[ext], "rb").read() == open(os.path.join(CSE, "%s.%s" % (case["name"], ext)), "rb").read(), ext


def test_gtf_bin_known_answer_and_stage_entry_points(gpu_ctx, work):
    from regtools_amd import _ffi
    L = _ffi.lib()
    # tests/lib/gtf/test_gtf_parser.cc:113-118: a single exon 12791-14103 lands in bin 37359 (the 32678 offset typo)
    gtf = os.path.join(str(work), "one.gtf")
    open(gtf, "w").write('22\tsrc\texon\t12791\t14103\t.\t+\t.\tgene_id "G"; gene_name "N"; transcript_id "ENST1";\n'
                         '22\tsrc\texon\t15000\t15100\t.\t+\t.\tgene_id "G"; gene_name "N"; transcript_id "ENST1";\n'
                         '22\tsrc\texon\t14000\t14103\t.\t-\t.\tgene_id "H"; transcript_id "ENST2";\n22\tsrc\texon\t12791\t12900\t.\t-\t.\tgene_id "H"; transcript_id "ENST2";\n')
    g = C.c_void_p()
    err = C.create_string_buffer(256)
    assert L.rgx_gtf_load(gpu_ctx._h, gtf.encode(), C.byref(g), err, 256) == 0, err.value
    n_tx, n_ex, n_ch = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L.rgx_gtf_info(g, C.byref(n_tx), C.byref(n_ex), C.byref(n_ch))
    assert (n_tx.value, n_ex.value, n_ch.value) == (2, 4, 1)
    b = C.c_uint32()
    assert L.rgx_gtf_transcript_bin(g, b"ENST2", C.byref(b)) == 0 and b.value == 37359     # negative strand: (start of last exon, end of first) reversed pair
    # a10: SNV one base into the intron after exon 1 of ENST1 -> splicing_intronic, distance 1, window = [exon0 start, exon1 end]
    chroms = (C.c_char_p * 3)(b"22", b"22", b"chrNope")
    pos0 = (C.c_uint32 * 3)(14104 - 1, 13000 - 1, 5)
    hits = C.POINTER(_ffi.VariantHits)()
    assert L.rgx_variant_windows(gpu_ctx._h, g, 3, chroms, pos0, 2, 3, 0, 0, 1, C.byref(hits), err, 256) == 0, err.value
    h = hits.contents
    assert [h.hit_off[i] for i in range(4)] == [0, 1, 1, 1]
    assert (h.hit_annotation[0], h.hit_distance[0], L.rgx_gtf_transcript_id(g, h.hit_transcript[0])) == (4, 1, b"ENST1")
    assert (h.cis_start[0], h.cis_end[0]) == (12791, 15100) and (h.cis_start[1], h.cis_end[1]) == (0xffffffff, 0)
    L.rgx_variant_hits_free(hits)
    # a11: the annotated junction 14103 -> 15000 on '+' is a known junction (DA) of ENST1; on '-' its start coincides with the end of
    # ENST2's first (genomically last) exon, which upstream calls a known ACCEPTOR on the negative strand (junctions_annotator.cc:283-285)
    jc = (C.c_char_p * 2)(b"22", b"22")
    js = (C.c_uint32 * 2)(14103, 14103); je = (C.c_uint32 * 2)(15000, 15000)
    ann = C.POINTER(_ffi.JunctionAnnot)()
    assert L.rgx_annotate_junctions(gpu_ctx._h, g, 2, jc, js, je, b"+-", C.byref(ann), err, 256) == 0, err.value
    a = ann.contents
    assert a.flags[0] == 7 and a.tx_off[1] - a.tx_off[0] == 1 and L.rgx_gtf_transcript_id(g, a.tx[0]) == b"ENST1"
    assert a.flags[1] == 2 and a.tx_off[2] - a.tx_off[1] == 1 and L.rgx_gtf_transcript_id(g, a.tx[a.tx_off[1]]) == b"ENST2"
    L.rgx_junction_annot_free(ann)
    L.rgx_gtf_free(g)


def test_cli_identify_and_errors(gpu_ctx, work):
    exe = os.path.join(ROOT, "bin", "regtools-amd")
    q = dict(vcf=os.path.join(REF_GOLD, "test1.vcf"), bam=os.path.join(REF_GOLD, "test_hcc1395.2.bam"),
             fasta=os.path.join(REF_GOLD, "test_chr22.fa"), gtf=os.path.join(REF_GOLD, "test_ensemble_chr22.2.gtf"))
    o = os.path.join(str(work), "cli")
    r = subprocess.run([exe, "cis-splice-effects", "identify", "-s", "RF", "-o", o + ".tsv", "-v", o + ".vcf", "-j", o + ".bed", q["vcf"], q["bam"], q["fasta"], q["gtf"]],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert open(o + ".tsv", "rb").read() == open(os.path.join(REF_GOLD, "expected-cis-splice-effects-identify-default-stranded-annotatedjunctions.out"), "rb").read()
    run = lambda *a: subprocess.run([exe, "cis-splice-effects", "identify"] + list(a), stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode
    # tests/integration-test/test_cis_splice_effects_identify.py:80-348 only pin exit codes for these
    assert run("-h") == 0
    assert run("-s", "XS", q["vcf"], q["bam"], q["fasta"]) == 1                     # missing gtf
    assert run(q["vcf"], q["bam"], q["fasta"], q["gtf"]) == 1                        # no -s
    assert run("-s", "XS", "nope.vcf", q["bam"], q["fasta"], q["gtf"]) == 1         # file_qc
    assert run("-s", "bogus", q["vcf"], q["bam"], q["fasta"], q["gtf"]) == 1


def test_window_join_equals_one_extraction_per_window(gpu_ctx, work):
    """rgx_window_join (row a9 on its own): for each window the rows equal what `junctions extract -r chr:beg-end` prints for it
    (window-restricted counts, thick bounds and names), while the BAM is inflated and scanned once for all windows."""
    import random
    from regtools_amd import _ffi, synth
    from conftest import run_oracle
    L = _ffi.lib()
    for shape, n, seed, contigs in (("short", 150000, 31, [("chr1", 248956422), ("chr7", 159345973), ("chrX", 156040895)]),
                                    ("fuzz", 60000, 32, [("1", 600000), ("10", 300000), ("2", 500000), ("MT", 16569)])):
        bam = os.path.join(str(work), "wj_%s.bam" % shape)
        synth.write(bam, n, shape=shape, seed=seed)
        rnd = random.Random(seed)
        wins = []
        for k in range(14):
            name, ln = rnd.choice(contigs)
            beg = rnd.randrange(0, ln - 10)
            span = rnd.choice([500, 20000, 2000000, ln])
            wins.append((name, beg, min(ln, beg + span)))
        wins.append((contigs[0][0], 0, contigs[0][1]))
        wins.append(wins[3])                                        # the same window twice: two independent result groups
        p = _ffi.ExtractParams(); L.rgx_extract_params_default(C.byref(p)); p.strandness = 0
        W = len(wins)
        chrom = (C.c_char_p * W)(*[w[0].encode() for w in wins])
        beg = (C.c_int32 * W)(*[w[1] for w in wins]); end = (C.c_int32 * W)(*[w[2] for w in wins])
        out = C.POINTER(_ffi.WindowRows)(); err = C.create_string_buffer(512)
        rc = L.rgx_window_join(gpu_ctx._h, bam.encode(), C.byref(p), W, chrom, beg, end, C.byref(out), err, len(err))
        assert rc == 0, err.value
        r = out.contents
        got = {w: [] for w in range(W)}
        for i in range(r.n):
            s, e, ts, te = r.start[i], r.end[i], r.thick_start[i], r.thick_end[i]
            if s - ts >= 8 and te - e >= 8:                         # print_all_junctions keeps anchored rows only
                got[r.window[i]].append("%s\t%d\t%d\tJUNC%08d\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n" % (
                    wins[r.window[i]][0], ts, te, r.name_index[i], r.read_count[i], r.strand[i].decode(), ts, te, s - ts, te - e, e - ts))
        nonempty = 0
        for w, (name, b0, e0) in enumerate(wins):
            rc, exp, _ = run_oracle(["-s", "XS", "-r", "%s:%d-%d" % (name, b0 + 1, e0), bam])
            assert rc == 0
            assert "".join(got[w]).encode() == exp, (shape, w, wins[w])
            nonempty += bool(exp)
        assert nonempty >= 4
        L.rgx_window_rows_free(out)
    # an unknown contig is the reference's region error
    chrom = (C.c_char_p * 1)(b"chrNope"); beg = (C.c_int32 * 1)(0); end = (C.c_int32 * 1)(10)
    out = C.POINTER(_ffi.WindowRows)()
    assert L.rgx_window_join(gpu_ctx._h, bam.encode(), C.byref(p), 1, chrom, beg, end, C.byref(out), err, len(err)) != 0
    assert b"Unable to iterate to region" in err.value


def test_window_join_batches_give_the_same_files(gpu_ctx, work):
    """The (window, event) pairs are materialised in batches of whole windows (REGTOOLS_AMD_PAIR_BATCH pairs per batch, 2^26 by default);
    with a batch of 200 pairs every case below runs through dozens of batches and must still write the reference's three files."""
    import sys
    env = dict(os.environ, REGTOOLS_AMD_PAIR_BATCH="200")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_cse as t, pytest; "
            "sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', '-k', 'test_equals_reference and (s1_00 or s2_03 or s3_07 or s1_11)', t.__file__]))"
            % (ROOT, os.path.join(ROOT, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]


def test_compressed_vcf_inputs(gpu_ctx, work):
    """hts_open takes gzip and bgzip VCFs (the reference's outputs for a .vcf.gz equal those for the plain file -- checked against
    oracle/_ref); here the members are inflated with the product's own decoder compiled for the host."""
    import gzip
    import bamio
    import regtools_amd
    src = open(os.path.join(REF_GOLD, "test1.vcf"), "rb").read()
    forms = {"gzip": gzip.compress(src, 6),
             "bgzip": b"".join(bamio.bgzf_member(src[k:k + 700]) for k in range(0, len(src), 700)) + bamio.EOF_MARKER,
             "two_members": gzip.compress(src[:1000]) + gzip.compress(src[1000:])}
    for name, blob in forms.items():
        vcf = os.path.join(str(work), "t1_%s.vcf.gz" % name)
        open(vcf, "wb").write(blob)
        q = dict(vcf=vcf, bam=os.path.join(REF_GOLD, "test_hcc1395.2.bam"), fasta=os.path.join(REF_GOLD, "test_chr22.fa"), gtf=os.path.join(REF_GOLD, "test_ensemble_chr22.2.gtf"))
        rc, files, ci, msg = gpu_identify(gpu_ctx, ["-s", "XS"], q, os.path.join(str(work), "gz_" + name))
        assert rc == 0, msg
        for ext, gold in (("tsv", "annotatedjunctions"), ("vcf", "annotatedvariants"), ("bed", "junctions")):
            assert open(files[ext], "rb").read() == open(os.path.join(REF_GOLD, "expected-cis-splice-effects-identify-default-%s.out" % gold), "rb").read(), (name, ext)
    # a truncated gzip stream is an error, not a crash
    bad = os.path.join(str(work), "bad.vcf.gz")
    open(bad, "wb").write(forms["gzip"][: len(forms["gzip"]) // 2])
    q["vcf"] = bad
    rc, files, ci, msg = gpu_identify(gpu_ctx, ["-s", "XS"], q, os.path.join(str(work), "gz_bad"))
    assert rc == 1


def test_bcf_input(gpu_ctx, work):
    """The variants as a BCF (bcf_hdr_read / bcf_read, vcf.c:788-818, 899-926): the reference's three outputs for the .bcf are the ones for the
    .vcf (checked against oracle/_ref when the fixtures were made), so the product's must be the stored goldens too -- including the -v file,
    which is re-serialised from the typed records."""
    import vcf_cases
    done = 0
    for case in MANIFEST:
        if case["rc"] != 0 or uses_motif(case["args"]) or done >= 3:
            continue
        q = dict(quartet(case["seed"], case["n_genes"], work))
        q["vcf"] = vcf_cases.simple_vcf_to_bcf(open(q["vcf"]).read(), os.path.join(str(work), "s%d.bcf" % case["seed"]))
        rc, files, ci, msg = gpu_identify(gpu_ctx, case["args"], q, os.path.join(str(work), "bcf_" + case["name"]))
        assert rc == 0, msg
        for ext in ("tsv", "vcf", "bed"):
            assert open(files[ext], "rb").read() == open(os.path.join(CSE, "%s.%s" % (case["name"], ext)), "rb").read(), (case["name"], ext)
        done += 1
    assert done == 3


def test_junction_scan_wave_form_equals_the_lane_form(gpu_ctx, work):
    """k_junction_scan_wave (one wave per junction, candidate transcripts dealt to the lanes, two wave scans for the order-dependent parts) against
    the one-lane-per-junction kernel it replaced (REGTOOLS_AMD_JSCAN=lane), on an annotation dense enough that junctions see several turns of 64
    candidate transcripts: `identify` and `junctions annotate`, every output file."""
    from regtools_amd import synth
    exe = os.path.join(ROOT, "bin", "regtools-amd")
    pre = os.path.join(str(work), "dense")
    synth.write(pre + ".bam", 400_000, shape="short", seed=11, n_genes=30_000)
    ann = synth.annotation(pre, 30_000, 20_000, seed=11, fasta=True)
    outs = {}
    for form in ("wave", "lane"):
        env = dict(os.environ)
        if form == "lane":
            env["REGTOOLS_AMD_JSCAN"] = "lane"
        o = pre + "." + form
        r = subprocess.run([exe, "cis-splice-effects", "identify", "-s", "XS", "-o", o + ".tsv", "-v", o + ".vcf", "-j", o + ".bed", ann["vcf"], pre + ".bam", ann["fasta"], ann["gtf"]],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([exe, "junctions", "annotate", "-o", o + ".ja", o + ".bed", ann["fasta"], ann["gtf"]], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        outs[form] = [open(o + e, "rb").read() for e in (".tsv", ".vcf", ".bed", ".ja")]
    assert outs["wave"] == outs["lane"]
    rows = outs["wave"][0].decode().splitlines()
    assert len(rows) > 200
    # (the test is about junctions with many candidates: some row lists more than 64 transcripts' worth of skipped elements or transcripts)
    assert max(len(r.split("\t")[16].split(",")) for r in rows[1:]) >= 3


def test_identify_over_several_device_listings_gives_the_same_files(gpu_ctx, work):
    """rgx_identify_multi: the extraction sharded over the listed devices, the events gathered on the first, the rest as on one device.
    One GPU is visible here, so the list names it two and five times (its shards take turns on it): the shard cuts, the per-shard extraction, the
    gather in file order and the join over the gathered events are the real ones, the copies are device-local.  Through the Python mirror and the
    CLI (REGTOOLS_AMD_DEVICES), with -s XS and with the window option."""
    import regtools_amd
    from regtools_amd import synth
    from regtools_amd.cse import CisSpliceEffectsIdentifier
    pre = os.path.join(str(work), "multi")
    st = synth.write(pre + ".bam", 1_500_000, shape="short", seed=23, n_genes=6_000)
    assert os.path.getsize(pre + ".bam") > (8 << 20)                  # (smaller files are not sharded)
    ann = synth.annotation(pre, 6_000, 30_000, seed=23, fasta=True)

    def run(tag, devices, extra):
        ci = CisSpliceEffectsIdentifier(ctx=gpu_ctx if not devices else None, devices=devices)
        o = pre + "." + tag
        ci.parse_options(["-s", "XS"] + extra + ["-o", o + ".tsv", "-v", o + ".vcf", "-j", o + ".bed", ann["vcf"], pre + ".bam", ann["fasta"], ann["gtf"]])
        ci.identify()
        return [open(o + e, "rb").read() for e in (".tsv", ".vcf", ".bed")], ci.stats

    for extra in ([], ["-w", "5000"]):
        one, s1 = run("one", None, extra)
        assert len(one[0].splitlines()) > 500
        for devices in ([0, 0], [0, 0, 0, 0, 0]):
            got, sm = run("m%d" % len(devices), devices, extra)
            assert got == one, (devices, extra)
            assert sm["n_records"] == s1["n_records"] == st["n_reads"] and sm["n_events"] == s1["n_events"]
    exe = os.path.join(ROOT, "bin", "regtools-amd")
    o = pre + ".cli"
    r = subprocess.run([exe, "cis-splice-effects", "identify", "-s", "XS", "-o", o + ".tsv", "-v", o + ".vcf", "-j", o + ".bed", ann["vcf"], pre + ".bam", ann["fasta"], ann["gtf"]],
                       env=dict(os.environ, REGTOOLS_AMD_DEVICES="0,0,0", REGTOOLS_AMD_TRACE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count(b"launch inflate") == 3, r.stderr[-3000:]             # (three shards were extracted, not one file)
    base, _ = run("one", None, [])
    assert [open(o + e, "rb").read() for e in (".tsv", ".vcf", ".bed")] == base
