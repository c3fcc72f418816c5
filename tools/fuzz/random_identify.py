"""`cis-splice-effects identify` over many synthetic GTF/VCF/FASTA/BAM quartets (tests/cse_synth.py) and random option sets.
Without `gpu`: the oracle judged by the real reference (dev container); with `gpu`: the product (C ABI) judged by the oracle.  All three
output files are compared byte for byte."""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cse_synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref"); ORC = os.path.join(ROOT, "oracle", "oracle_cli")
if GPU:
    import regtools_amd
    ctx = regtools_amd.Context(0)
bad = 0
with tempfile.TemporaryDirectory() as td:
    for case in range(n_cases):
        seed = rng.randrange(1000, 100000)
        q = cse_synth.build(os.path.join(td, "s%d" % case), seed=seed, n_genes=rng.choice([3, 8, 20]), reads_per_junction=rng.choice([1, 4, 9]))
        if os.environ.get("FUZZ_DAMAGE"):
            # the record stream ends somewhere: a member that does not inflate, or the file cut short (upstream reads every window through the index on its own)
            import bamio
            raw = bytearray(open(q["bam"], "rb").read())
            mem = list(bamio.bgzf_members(bytes(raw)))
            if rng.random() < 0.7 and len(mem) > 3:
                for _ in range(rng.choice([1, 1, 2])):
                    coff, payload, _isz = mem[rng.randrange(1, len(mem) - 1)]
                    for j in range(rng.randrange(0, max(1, len(payload) - 24)), min(len(payload), rng.randrange(8, 64) + 40)):
                        raw[coff + 18 + j] ^= rng.randrange(1, 256)
            else:
                raw = raw[: rng.randrange(len(raw) // 4, len(raw))]
            open(q["bam"], "wb").write(bytes(raw))
        args = ["-s", rng.choice(["XS", "RF", "FR", "intron-motif"])]
        for flag, vals in (("-w", [1, 100, 5000, 200000]), ("-e", [0, 1, 3, 10]), ("-i", [0, 2, 50]), ("-a", [1, 8, 20]), ("-M", [500, 500000])):
            if rng.random() < 0.3: args += [flag, str(rng.choice(vals))]
        for flag in ("-E", "-I", "-S", "-C"):
            if rng.random() < 0.2: args.append(flag)
        outs = {}
        for who in ("a", "b"):
            files = [os.path.join(td, "%s_%d.%s" % (who, case, x)) for x in ("tsv", "vcf", "bed")]
            tail = ["-o", files[0], "-v", files[1], "-j", files[2], q["vcf"], q["bam"], q["fasta"], q["gtf"]]
            if who == "a":
                rc = subprocess.run([ORC, "identify"] + args + tail, capture_output=True).returncode
            elif GPU:
                ci = regtools_amd.CisSpliceEffectsIdentifier(ctx=ctx)
                try:
                    ci.parse_options(args + tail); ci.identify(); rc = 0
                except regtools_amd.RegtoolsError as e:
                    rc = 0 if e.code == 0 else 1
            else:
                rc = subprocess.run([REF, "cis-splice-effects", "identify"] + args + tail, capture_output=True).returncode
                if rc not in (0, 1): rc = None
            outs[who] = (rc, [open(f, "rb").read() if os.path.exists(f) else None for f in files] if rc == 0 else None)
        if outs["b"][0] is None:
            continue
        if outs["a"] != outs["b"]:
            bad += 1
            print("DISAGREE case %d seed %d args %s rc %s vs %s" % (case, seed, args, outs["a"][0], outs["b"][0]), flush=True)
print("cases %d disagreements %d" % (n_cases, bad))
