"""`cis-splice-effects associate`, `junctions annotate`, `variants annotate` over random synthetic quartets (tests/cse_synth.py) and option
sets.  Without `gpu`: the oracle judged by the real reference (dev container); with `gpu`: the product (C ABI) judged by the oracle."""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cse_synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref"); ORC = os.path.join(ROOT, "oracle", "oracle_cli")
if GPU:
    import regtools_amd
    from regtools_amd import cse
    ctx = regtools_amd.Context(0)
bad = n = 0


def outputs(files):
    return [open(f, "rb").read() if os.path.exists(f) else None for f in files]


with tempfile.TemporaryDirectory() as td:
    for case in range(n_cases):
        seed = rng.randrange(1000, 100000)
        q = cse_synth.build(os.path.join(td, "s%d" % case), seed=seed, n_genes=rng.choice([3, 8, 20]), reads_per_junction=rng.choice([1, 4]))
        bed = os.path.join(td, "j%d.bed" % case)
        subprocess.run([ORC, "extract", "-s", rng.choice(["XS", "RF"]), "-a", str(rng.choice([1, 8])), "-o", bed, q["bam"]], check=True, capture_output=True)
        opt = []
        for flag, vals in (("-w", [1, 100, 5000, 200000]), ("-e", [0, 1, 3, 10]), ("-i", [0, 2, 50])):
            if rng.random() < 0.3: opt += [flag, str(rng.choice(vals))]
        for flag in ("-E", "-I", "-S"):
            if rng.random() < 0.2: opt.append(flag)
        va_opt = [x for x in opt if x not in ("-w",)]
        if "-w" in opt:
            k = opt.index("-w"); va_opt = opt[:k] + opt[k + 2:]
        jobs = [("associate", opt, 3), ("variants", va_opt, 1), ("junctions", [], 1)]
        for what, o, n_files in jobs:
            res = {}
            for who in ("oracle", "other"):
                files = [os.path.join(td, "%s_%s_%d.%d" % (who, what, case, k)) for k in range(n_files)]
                for f in files:
                    if os.path.exists(f): os.remove(f)
                if what == "associate":
                    tail = o + ["-o", files[0], "-v", files[1], "-j", files[2], q["vcf"], bed, q["fasta"], q["gtf"]]
                    ocmd, rcmd = [ORC, "associate"], [REF, "cis-splice-effects", "associate"]
                elif what == "variants":
                    tail = o + ["-o", files[0], q["vcf"], q["gtf"]]
                    ocmd, rcmd = [ORC, "variants-annotate"], [REF, "variants", "annotate"]
                else:
                    tail = ["-o", files[0], bed, q["fasta"], q["gtf"]]
                    ocmd, rcmd = [ORC, "junctions-annotate"], [REF, "junctions", "annotate"]
                if who == "oracle":
                    rc = subprocess.run(ocmd + tail, capture_output=True).returncode
                elif GPU:
                    obj = {"associate": cse.CisSpliceEffectsAssociator, "variants": cse.VariantsAnnotator, "junctions": cse.JunctionsAnnotator}[what](ctx=ctx)
                    try:
                        obj.parse_options(tail)
                        {"associate": lambda: obj.associate(), "variants": lambda: obj.annotate_vcf(), "junctions": lambda: obj.annotate()}[what]()
                        rc = 0
                    except regtools_amd.RegtoolsError as e:
                        rc = 0 if e.code == 0 else 1
                else:
                    rc = subprocess.run(rcmd + tail, capture_output=True).returncode
                    if rc not in (0, 1): rc = None
                res[who] = (rc, outputs(files) if rc == 0 else None)
            if res["other"][0] is None:
                continue
            n += 1
            if res["oracle"] != res["other"]:
                bad += 1
                print("DISAGREE case %d seed %d %s opts %s rc %s vs %s" % (case, seed, what, o, res["oracle"][0], res["other"][0]), flush=True)
print("runs %d disagreements %d" % (n, bad))
