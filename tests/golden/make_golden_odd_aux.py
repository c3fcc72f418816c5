"""Dev container only: runs the REAL reference (oracle/_ref/regtools_ref) on tests/odd_aux_cases.py's inputs and stores its status, streams and output files
-> tests/golden/cli/cli_odd_aux_streams.json.   python tests/golden/make_golden_odd_aux.py"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import odd_aux_cases
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
gold = {}
with tempfile.TemporaryDirectory() as td:
    cs, digest = odd_aux_cases.cases(td)
    gold["inputs_sha256"] = digest
    for argv, files in cs:
        r = subprocess.run([REF] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        fix = lambda b: b.replace(ROOT.encode(), b"@ROOT@").replace(td.encode(), b"@TMP@").decode("latin-1")
        key = " ".join(os.path.basename(a) for a in argv)
        gold[key] = {"rc": r.returncode, "stdout": fix(r.stdout), "stderr": fix(r.stderr),
                     "files": {os.path.basename(f): open(f, "rb").read().decode("latin-1") for f in files if os.path.exists(f)}}
        print(key, r.returncode, len(r.stderr), sorted(gold[key]["files"]))
json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "cli", "cli_odd_aux_streams.json"), "w"), indent=0, sort_keys=True)
