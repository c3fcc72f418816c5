"""Record what the REAL reference (oracle/_ref/regtools_ref = /root/reference/src compiled in place) writes to stdout and stderr for
the argument lists of tests/test_cli_contract.py -> tests/golden/cli/cli_streams.json.  Run in the dev container:
    python tests/golden/make_golden_cli.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import test_cli_contract as t  # noqa: E402

out = {}
for argv, rc in t.CASES[t.TOP_LEVEL:]:
    r = t.run_full(t.REF, argv)
    assert r[0] == rc, (argv, r)
    out[t.case_id(argv)] = {"rc": r[0], "stdout": r[1].decode("latin-1"), "stderr": r[2].decode("latin-1")}
os.makedirs(os.path.join(HERE, "cli"), exist_ok=True)
json.dump(out, open(os.path.join(HERE, "cli", "cli_streams.json"), "w"), indent=1, sort_keys=True)
print("%d argument lists recorded" % len(out))

# round 6: a GTF with an empty line -- the reference dies of an uncaught std::out_of_range (SIGABRT, libstdc++'s terminate message on stderr)
import subprocess, tempfile
ab = {}
with tempfile.TemporaryDirectory() as td:
    g = open(t.GTF).read().splitlines()
    bad = os.path.join(td, "empty_line.gtf")
    open(bad, "w").write("\n".join(g[:20] + [""] + g[20:]) + "\n")
    for argv in t.abort_cases(bad):
        r = subprocess.run([t.REF] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        fix = lambda b: b.replace(t.ROOT.encode(), b"@ROOT@").replace(td.encode(), b"@TMP@")
        ab[t.case_id(argv)] = {"rc": r.returncode, "stdout": fix(r.stdout).decode("latin-1"), "stderr": fix(r.stderr).decode("latin-1")}
        assert r.returncode in (-6, 1), (argv, r.returncode)
json.dump(ab, open(os.path.join(HERE, "cli", "cli_abort_streams.json"), "w"), indent=1, sort_keys=True)
print("%d aborting argument lists recorded" % len(ab))

# round 6: runs that go all the way -- the whole of stderr
va = {}
with tempfile.TemporaryDirectory() as td:
    for argv in t.valid_cases(td):
        r = subprocess.run([t.REF] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        fix = lambda b: b.replace(t.ROOT.encode(), b"@ROOT@").replace(td.encode(), b"@TMP@")
        assert r.returncode == 0, (argv, r.returncode, r.stderr[-300:])
        va[t.case_id(argv)] = {"rc": 0, "stdout": fix(r.stdout).decode("latin-1"), "stderr": fix(r.stderr).decode("latin-1")}
json.dump(va, open(os.path.join(HERE, "cli", "cli_valid_streams.json"), "w"), indent=1, sort_keys=True)
print("%d complete runs recorded" % len(va))

# round 6: what htslib says about a VCF and where it ends the process (tests/test_cli_contract.py vcf_note_cases)
vn = {}
with tempfile.TemporaryDirectory() as td:
    for argv, status in t.vcf_note_cases(td) + t.bam_note_cases(td):
        r = subprocess.run([t.REF] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == status, (argv, r.returncode, r.stderr[-300:])
        outs = [a for a in argv if a.startswith(td) and os.path.basename(a)[0] in "niab" and os.path.basename(a)[1].isdigit()]
        vn[t.case_id(argv)] = {"rc": r.returncode, "stdout": t.normalise_streams(r.stdout, td).decode("latin-1"), "stderr": t.normalise_streams(r.stderr, td).decode("latin-1"),
                               "files": {os.path.basename(a): open(a, "rb").read().decode("latin-1") for a in outs} if status == 0 else {}}
json.dump(vn, open(os.path.join(HERE, "cli", "cli_vcf_notes_streams.json"), "w"), indent=1, sort_keys=True)
print("%d runs over talkative VCFs recorded" % len(vn))
