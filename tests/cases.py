"""Shared helpers: the golden manifest and how a case's input BAM is materialised."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))

REF_GOLDENS = [  # the reference's own integration goldens (tests/integration-test/test_junctions_extract.py:34-85)
    (["-s", "XS"], "expected-a.out"),
    (["-s", "XS", "-a", "30"], "expected-a30.out"),
    (["-s", "RF"], "expected-stranded-a.out"),
    (["-s", "RF", "-a", "30"], "expected-stranded-a30.out"),
    (["-s", "XS", "-m", "8039", "-M", "8039"], "expected-i8039-I8039.out"),
    (["-s", "XS", "-r", "1:22405013-22405020"], "expected-r1:22405013-22405020.out"),
]

_synth_cache = {}


def case_bam(case, tmpdir):
    """Path of the case's BAM: a committed fixture, or a deterministic synthetic file generated on demand."""
    if case["bam"]:
        return os.path.join(GOLD, case["bam"])
    if case.get("cse"):
        import cse_synth
        key = ("cse", case["cse"]["seed"], case["cse"]["n_genes"])
        if key not in _synth_cache:
            _synth_cache[key] = cse_synth.build(os.path.join(str(tmpdir), "cse_q%d_%d" % key[1:]), seed=key[1], n_genes=key[2])
        return _synth_cache[key]["bam"]
    if case.get("framing"):
        import framing_cases
        key = ("framing", case["framing"])
        if key not in _synth_cache:
            _synth_cache[key] = framing_cases.build(case["framing"], os.path.join(str(tmpdir), "framing_%s.bam" % case["framing"]))
        return _synth_cache[key]
    from regtools_amd import synth
    s = case["synth"]
    key = (s["shape"], s["n_reads"], s["seed"])
    if key not in _synth_cache:
        p = os.path.join(str(tmpdir), "synth_%s_%d_%d.bam" % key)
        synth.write(p, s["n_reads"], shape=s["shape"], seed=s["seed"])
        _synth_cache[key] = p
    return _synth_cache[key]


def case_argv(case, tmpdir):
    """The full argument list of the case: options, the BAM, then whatever follows it (the FASTA of a `cse` quartet)."""
    bam = case_bam(case, tmpdir)
    after = []
    for a in case.get("after", []):
        after.append(_synth_cache[("cse", case["cse"]["seed"], case["cse"]["n_genes"])]["fasta"] if a == "<fasta>" else a)
    return list(case["args"]) + [bam] + after


def expected(case):
    return open(os.path.join(GOLD, "expected", case["expected"]), "rb").read()
