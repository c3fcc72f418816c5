"""the host CLI keeps the reference's subcommands, option letters, exit codes AND streams: which text goes to
stdout and which to stderr (junctions_main.cc:35-41,96-107; variants_main.cc:33,64; variants_annotator.cc:80,92;
junctions_annotator.h:235; cis_splice_effects_main.cc:73-93).  Every argument list here ends before any device work, so the test
runs without a GPU; where the real reference binary is present (dev container, oracle/_ref) exit code, stdout and stderr are
compared with it directly, and everywhere with the copies of its output committed under tests/golden/cli (made by
tests/golden/make_golden_cli.py from that binary).  Since round 6 that binary is the reference's own main() (src/regtools.cc compiled with the version.h
cmake would generate: oracle/Makefile), so the version banner and the top-level usage text are compared byte for byte as well.  The cases need no device:
they carry the gpu marker too, so that the driver's GPU run sees them next to the three CLI tests that do device work."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "regtools-amd")
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
VCF, FA, GTF = (os.path.join(GOLD, "cse_ref", x) for x in ("test1.vcf", "test_chr22.fa", "test_ensemble_chr22.2.gtf"))
BAM = os.path.join(GOLD, "cse_ref", "test_hcc1395.2.bam")
BED = os.path.join(GOLD, "annot_ref", "junctions_extract.bed")

# (argv, exit code of the reference)   -- test_regtools_main.py, test_junctions_main.py, test_junctions_extract.py:87-109,
# test_cis_splice_effects_identify.py, test_variants_main.py, test_cis_splice_effects_associate.py:56-60
CASES = [
    ([], 0), (["-h"], 0), (["nonsense"], 0),
    (["junctions"], 0), (["junctions", "-h"], 0), (["junctions", "extract", "-h"], 0), (["junctions", "annotate", "-h"], 0),
    (["junctions", "extract", "-s", "XS", "-o", "/dev/null"], 1),                       # no BAM
    (["junctions", "extract", "-o", "/dev/null", BAM], 1),                              # no strandness
    (["junctions", "extract", "-s", "sideways", BAM], 1),
    (["junctions", "extract", "-s", "intron-motif", BAM], 1),                           # needs a FASTA
    (["junctions", "extract", "-Q", "-s", "XS", BAM], 1),                               # unknown option
    (["junctions", "annotate"], 1), (["junctions", "annotate", BED, FA], 1),
    (["variants"], 0), (["variants", "-h"], 0), (["variants", "annotate", "-h"], 0), (["variants", "annotate"], 1), (["variants", "annotate", VCF], 1),
    (["cis-splice-effects"], 0), (["cis-splice-effects", "-h"], 0), (["cis-splice-effects", "identify", "-h"], 0), (["cis-splice-effects", "associate", "-h"], 0),
    (["cis-splice-effects", "identify", VCF, BAM, FA, GTF], 1),                         # no -s
    (["cis-splice-effects", "identify", "-s", "XS", VCF, BAM, FA], 1),
    (["cis-splice-effects", "identify", "-s", "XS", VCF, BAM, FA, "/no/such.gtf"], 1),
    (["cis-splice-effects", "identify", "-s", "up", VCF, BAM, FA, GTF], 1),
    (["cis-splice-effects", "associate", VCF, BED, FA], 1),
    (["cis-splice-effects", "associate", VCF, "/no/such.bed", FA, GTF], 1),
    (["cis-splice-effects", "associate", "-s", "XS", VCF, BED, FA, GTF], 1),            # -s is not an option of associate
    # round 5: more ways of ending in a usage text, for the streams
    (["junctions", "nonsense"], 0), (["variants", "nonsense"], 0), (["cis-splice-effects", "nonsense"], 0),
    (["junctions", "annotate", "-Q", BED, FA, GTF], 1), (["junctions", "annotate", BED, FA, GTF, "extra"], 1),
    (["variants", "annotate", "-Q", VCF, GTF], 1),
    (["cis-splice-effects", "identify", "-Q", "-s", "XS", VCF, BAM, FA, GTF], 1),
    (["cis-splice-effects", "identify", "-s", "XS", VCF, BAM, FA, GTF, "extra"], 1),
    (["cis-splice-effects", "identify", "-s", "XS"], 1),
    (["cis-splice-effects", "associate", VCF, BED, FA, GTF, "extra"], 1),
    (["cis-splice-effects", "associate"], 1),
    (["junctions", "extract", "-s", "XS", BAM, FA, "extra"], 1),
]
TOP_LEVEL = 0                # (round 5: the first three lists ended in a stand-in's usage text and were compared by exit code only)


def abort_cases(bad_gtf):
    """argument lists that end in GtfParser::load meeting an empty line (gtf_parser.cc:230: line.at(0), uncaught -> SIGABRT)"""
    return [["junctions", "annotate", BED, FA, bad_gtf], ["variants", "annotate", VCF, bad_gtf],
            ["cis-splice-effects", "identify", "-s", "XS", VCF, BAM, FA, bad_gtf], ["cis-splice-effects", "associate", VCF, BED, FA, bad_gtf]]


def valid_cases(td):
    """runs that go all the way: every byte the reference writes to stderr on the way (option echo, "exonic_min_distance_ is 3", a block per
    splice-relevant variant, "Annotated n lines.") and the exit status; outputs go to files under td"""
    o = lambda n: os.path.join(td, n)
    return [["junctions", "extract", "-s", "XS", "-o", o("je.bed"), BAM],
            ["junctions", "extract", "-s", "RF", "-a", "6", "-m", "50", "-M", "100000", "-r", "22:1-1000000", "-o", o("je2.bed"), BAM],
            ["junctions", "annotate", "-o", o("ja.tsv"), BED, FA, GTF], ["junctions", "annotate", "-S", "-o", o("ja2.tsv"), BED, FA, GTF],
            ["variants", "annotate", "-o", o("va.vcf"), VCF, GTF], ["variants", "annotate", "-E", "-I", "-S", "-o", o("va2.vcf"), VCF, GTF],
            ["variants", "annotate", "-e", "5", "-i", "4", "-o", o("va3.vcf"), VCF, GTF],
            ["cis-splice-effects", "identify", "-s", "XS", "-o", o("ci.tsv"), "-v", o("ci.vcf"), "-j", o("ci.bed"), VCF, BAM, FA, GTF],
            ["cis-splice-effects", "identify", "-s", "RF", "-w", "100", "-o", o("ci2.tsv"), VCF, BAM, FA, GTF],
            ["cis-splice-effects", "identify", "-s", "XS", "-E", "-I", "-o", o("ci3.tsv"), VCF, BAM, FA, GTF],
            ["cis-splice-effects", "associate", "-o", o("ca.tsv"), "-v", o("ca.vcf"), "-j", o("ca.bed"), VCF, BED, FA, GTF],
            ["cis-splice-effects", "associate", "-w", "500", "-o", o("ca2.tsv"), VCF, BED, FA, GTF]]


def vcf_note_cases(td):
    """The test VCF with records that make htslib talk or end the process (vcf.c: undeclared names once per name, sample columns that do not fit,
    exit(1) on a sample with too many fields, abort() on a Flag in FORMAT), through the tool: (argument list, expected status)."""
    lines = open(VCF).read().split("\n")
    first = next(k for k, l in enumerate(lines) if l and not l.startswith("#"))
    head, recs = lines[:first], [l for l in lines[first:] if l]

    def variant(name, edit, extra_header=()):
        r = [l.split("\t") for l in recs]
        edit(r)
        path = os.path.join(td, name + ".vcf")
        open(path, "w").write("\n".join(head[:-1] + list(extra_header) + head[-1:] + ["\t".join(x) for x in r]) + "\n")
        return path

    def names(r):                              # undeclared names on records of both kinds (splice relevant or not), a repeat, the four annotation keys
        r[0][7] += ";NEW1=5"; r[0][6] = "lowq"; r[2][7] += ";NEW1=6;NEW2"; r[4][7] = "genes=G;" + r[4][7]; r[7][0] = "23"; r[9][8] += ":ZZ"; r[9][9] += ":1"; r[9][10] += ":2"
        r[12][7] += ";NEW3=x"; r[12][6] = "lowq;q10"

    def short(r): names(r); r[10] = r[10][:10]                      # one sample column short: the read loop ends there
    def many(r): names(r); r[10][9] += ":7:8"                       # more fields than FORMAT has keys: exit(1) inside htslib
    def many_early(r): r[0][9] += ":7:8"
    def flag(r): names(r); r[10][8] += ":FL"; r[10][9] += ":1"; r[10][10] += ":1"     # a Flag among the FORMAT keys: abort() inside htslib

    o = lambda n: os.path.join(td, n)
    fl = ['##FORMAT=<ID=FL,Number=0,Type=Flag,Description="a flag">']
    v_names, v_short, v_many, v_early, v_flag = (variant("names", names), variant("short", short), variant("many", many), variant("many_early", many_early),
                                                 variant("flag", flag, fl))
    ident = ["cis-splice-effects", "identify", "-s", "XS"]
    return [(["variants", "annotate", "-o", o("n1.vcf"), v_names, GTF], 0), (["variants", "annotate", "-o", o("n2.vcf"), v_short, GTF], 0),
            (["variants", "annotate", "-o", o("n3.vcf"), v_many, GTF], 1), (["variants", "annotate", "-o", o("n4.vcf"), v_flag, GTF], -6),
            (ident + ["-o", o("i1.tsv"), "-v", o("i1.vcf"), v_names, BAM, FA, GTF], 0), (ident + ["-o", o("i2.tsv"), v_names, BAM, FA, GTF], 0),
            (ident + ["-o", o("i3.tsv"), "-v", o("i3.vcf"), v_short, BAM, FA, GTF], 0), (ident + ["-o", o("i4.tsv"), "-v", o("i4.vcf"), v_many, BAM, FA, GTF], 1),
            (ident + ["-o", o("i5.tsv"), v_flag, BAM, FA, GTF], -6), (ident + ["-o", o("i6.tsv"), v_early, BAM, FA, GTF], 1),
            (["cis-splice-effects", "associate", "-o", o("a1.tsv"), "-v", o("a1.vcf"), v_names, BED, FA, GTF], 0),
            (["cis-splice-effects", "associate", "-o", o("a2.tsv"), v_many, BED, FA, GTF], 1)]


def bam_note_cases(td):
    """what htslib says when a BAM and its index are opened and a region string is read (sam.c:122-127 no EOF member; hts.c:2046-2054 an index older than its
    file; hts.c:1865-1872 the numbers of -r; hts.c:408-411 a file that is not there): (argument list, expected status)"""
    import shutil
    o = lambda n: os.path.join(td, n)
    old, cut = o("old_index.bam"), o("no_eof.bam")
    shutil.copy(BAM, old); shutil.copy(BAM + ".bai", old + ".bai")
    os.utime(old + ".bai", (1500000000, 1500000000))
    data = open(BAM, "rb").read()
    assert data[-28:-16] == bytes.fromhex("1f8b08040000000000ff0600")
    open(cut, "wb").write(data[:-28]); shutil.copy(BAM + ".bai", cut + ".bai")
    os.utime(cut, (1500000000, 1500000000))                                    # (the copy's index is not the older one)
    import vcf_cases                                                           # a VCF with a tabix index next to it, the index the older file
    with_tbi = o("with_tbi.vcf")
    open(with_tbi, "wb").write(vcf_cases.build(td)["tbi_with_bins"])
    for suffix, blob in vcf_cases.companions()["tbi_with_bins"].items(): open(with_tbi[:-4] + suffix, "wb").write(blob)
    for suffix in vcf_cases.companions()["tbi_with_bins"]: os.utime(with_tbi[:-4] + suffix, (1500000000, 1500000000))
    je = ["junctions", "extract", "-s", "XS"]
    ident = ["cis-splice-effects", "identify", "-s", "XS"]
    return [(["variants", "annotate", "-o", o("b0.vcf"), with_tbi, GTF], 0), (je + ["-o", o("b1.bed"), cut], 0), (je + ["-o", o("b2.bed"), old], 0), (je + ["-o", o("b3.bed"), "-r", "22:1-100000.5", BAM], 0),
            (je + ["-o", o("b4.bed"), "-r", "22:1-100000x", BAM], 0), (je + ["-o", o("b5.bed"), "-r", "22:1.55e1-1e5", BAM], 0),
            (je + ["-o", o("b6.bed"), "-r", "22:1,000-200,000", BAM], 0), (je + ["-o", o("b7.bed"), o("not_there.bam")], 1),
            (ident + ["-o", o("b8.tsv"), VCF, cut, FA, GTF], 0), (ident + ["-o", o("b9.tsv"), "-v", o("b9.vcf"), VCF, old, FA, GTF], 0)]


def normalise_streams(b, td):
    """paths of this checkout and of the temporary directory; htslib's __FILE__ in front of "vcf.c:<line>" (the build's path upstream, nothing here)"""
    import re
    return re.sub(rb"\[[^\] ]*/(vcf\.c:\d+ )", rb"[\1", b.replace(ROOT.encode(), b"@ROOT@").replace(td.encode(), b"@TMP@"))


def case_id(argv):
    return " ".join(os.path.basename(a) for a in argv) or "(none)"


def run_full(exe, argv):
    r = subprocess.run([exe] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return r.returncode, r.stdout.replace(ROOT.encode(), b"@ROOT@"), r.stderr.replace(ROOT.encode(), b"@ROOT@")


def run(exe, argv):
    return run_full(exe, argv)[0]


@pytest.mark.parametrize("argv,ref_rc", CASES, ids=[case_id(c[0]) for c in CASES])
def test_exit_codes_match_the_reference(built, argv, ref_rc):
    if os.path.exists(REF):
        assert run(REF, argv) == ref_rc, "the recorded reference exit code is stale"
    assert run(EXE, argv) == ref_rc


# (twice: once in the CPU suite, once -- the same check, marked gpu -- in the suite the driver runs on the GPU box)
@pytest.mark.parametrize("suite", ["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("k", range(TOP_LEVEL, len(CASES)), ids=[case_id(c[0]) for c in CASES[TOP_LEVEL:]])
def test_stdout_and_stderr_bytes_match_the_reference(built, k, suite):
    argv, ref_rc = CASES[k]
    gold = json.load(open(os.path.join(GOLD, "cli", "cli_streams.json")))[case_id(argv)]
    want_out, want_err = gold["stdout"].encode("latin-1"), gold["stderr"].encode("latin-1")
    if os.path.exists(REF):
        rc, out, err = run_full(REF, argv)
        assert (rc, out, err) == (ref_rc, want_out, want_err), "tests/golden/cli is stale: run tests/golden/make_golden_cli.py"
    rc, out, err = run_full(EXE, argv)
    assert rc == ref_rc
    assert out == want_out
    assert err == want_err


@pytest.mark.gpu
def test_empty_gtf_line_ends_the_process_as_upstream(built, tmp_path):
    """An empty line in a GTF: upstream's loader calls line.at(0) outside any try block; `junctions annotate` and `variants annotate` die of SIGABRT behind
    libstdc++'s terminate message (status 134 in a shell), the two `cis-splice-effects` commands catch std::exception and print its what() (status 1).
    The tool does the same call when the library reports such a line: same status, same streams.
    (The context is made before the annotation is read, hence the gpu marker.)"""
    gold = json.load(open(os.path.join(GOLD, "cli", "cli_abort_streams.json")))
    g = open(GTF).read().splitlines()
    bad = os.path.join(str(tmp_path), "empty_line.gtf")
    open(bad, "w").write("\n".join(g[:20] + [""] + g[20:]) + "\n")
    for argv in abort_cases(bad):
        want = gold[case_id(argv)]
        r = subprocess.run([EXE] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        fix = lambda b: b.replace(ROOT.encode(), b"@ROOT@").replace(str(tmp_path).encode(), b"@TMP@")
        assert r.returncode == want["rc"] and want["rc"] in (-6, 1), (argv, r.returncode, r.stderr[-300:])
        assert fix(r.stdout) == want["stdout"].encode("latin-1"), argv
        assert fix(r.stderr) == want["stderr"].encode("latin-1"), argv


@pytest.mark.gpu
def test_stderr_of_runs_that_go_all_the_way(built, tmp_path):
    """The reference talks on stderr while it works: the option echo, "exonic_min_distance_ is 3" from the annotator's constructor, for every
    splice-relevant variant "Variant <BED fields>" and "Variant region is <region>", "Annotated n lines." -- byte for byte (round 6:
    rgx_identify_params.echo, the tool's echo blocks)."""
    gold = json.load(open(os.path.join(GOLD, "cli", "cli_valid_streams.json")))
    td = str(tmp_path)
    for argv in valid_cases(td):
        want = gold[case_id(argv)]
        r = subprocess.run([EXE] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        fix = lambda b: b.replace(ROOT.encode(), b"@ROOT@").replace(td.encode(), b"@TMP@")
        assert r.returncode == want["rc"] == 0, (argv, r.returncode, r.stderr[-300:])
        assert fix(r.stdout) == want["stdout"].encode("latin-1"), argv
        assert fix(r.stderr) == want["stderr"].encode("latin-1"), (argv, fix(r.stderr)[-400:])


@pytest.mark.gpu
def test_what_htslib_says_about_a_vcf_and_where_it_ends_the_process(built, tmp_path):
    """htslib talks while the reference reads a VCF -- a line for every name the header does not declare, once per name, between the "Variant" blocks of
    `cis-splice-effects`; a line for the record whose sample columns do not fit, where reading stops -- and on two kinds of record it ends the process
    itself, past regtools' handlers: exit(1) behind "Incorrect number of FORMAT fields at ...", abort() behind "the format type 0 currently not supported".
    The same for a BAM: no EOF member, an index older than its file (both said again for every splice-relevant variant of `identify`, which opens the BAM per
    variant upstream), numbers in -r that lose digits or drag letters along, a file that is not there.
    Streams and status are the reference's (tests/golden/cli/cli_vcf_notes_streams.json), and so are the files of the runs that complete."""
    gold = json.load(open(os.path.join(GOLD, "cli", "cli_vcf_notes_streams.json")))
    td = str(tmp_path)
    for argv, status in vcf_note_cases(td) + bam_note_cases(td):
        want = gold[case_id(argv)]
        r = subprocess.run([EXE] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == want["rc"] == status, (argv, r.returncode, r.stderr[-300:])
        assert normalise_streams(r.stdout, td) == want["stdout"].encode("latin-1"), argv
        assert normalise_streams(r.stderr, td) == want["stderr"].encode("latin-1"), (argv, normalise_streams(r.stderr, td)[-600:])
        if status == 0:
            for path, text in want["files"].items():
                assert open(os.path.join(td, path), "rb").read() == text.encode("latin-1"), (argv, path)
