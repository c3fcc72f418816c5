"""Differential run of the oracle against the real reference (dev container only: needs oracle/_ref/regtools_ref) on random small BAMs:
random CIGARs over all ten ops, flags, strand tags of every type, positions at the contig ends, huge / tiny introns, several contigs
whose names sort differently from their order, random -a/-m/-M/-s/-t/-r."""
import os, random, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamio
from regtools_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref"); ORC = os.path.join(ROOT, "oracle", "oracle_cli")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"          # judge the PRODUCT (through the C ABI) by the oracle instead of the oracle by the reference
if GPU:
    import regtools_amd
    ctx = regtools_amd.Context(0)
bad = rows_total = nonempty = 0
with tempfile.TemporaryDirectory() as td:
    for case in range(n_cases):
        contigs = [(n, rng.choice([5000, 200000, 3000000])) for n in rng.sample(["chr1", "chr10", "chr2", "1", "X", "MT", "chrUn_1", "b", "a"], rng.randrange(1, 5))]
        recs = []
        with_bc = rng.random() < 0.35                      # -b: cell barcodes (junctions_extractor.cc:362-374)
        bcs = ["".join(rng.choice("ACGT") for _ in range(rng.choice([1, 8, 16]))) + "-1" for _ in range(rng.choice([1, 3, 40]))] + ["?", ""]
        for tid, (name, L) in enumerate(contigs):
            pos = 0
            for k in range(rng.randrange(0, 120)):
                pos += rng.choice([0, 0, 1, 7, 300, 5000])
                if pos >= L: break
                nops = rng.choice([0, 1, 1, 2, 3, 3, 3, 5, 9, 17, 40])
                ops = []
                for _ in range(nops):
                    op = rng.choice("MMMMNNNIDSHP=XB")
                    ln = rng.choice([0, 1, 2, 7, 8, 9, 30, 69, 70, 71, 500, 499999, 500000, 500001, 2000000]) if op == "N" else rng.choice([0, 1, 5, 8, 20, 76])
                    ops.append("%d%s" % (ln, op))
                cig = "".join(ops) or "*"
                flag = rng.choice([0, 16, 4, 99, 147, 83, 163, 256, 2048, 1 | 64 | 16, 1 | 128 | 32, 0x400])
                aux = b""
                r = rng.random()
                if r < 0.5: aux += bamio.tagA("XS", rng.choice("+-?.*"))
                elif r < 0.6: aux += b"XSC\x2b"
                elif r < 0.7: aux += bamio.tagZ("XS", "+")
                elif r < 0.75: aux += b"XSA\x00"
                if rng.random() < 0.3: aux = bamio.tagZ("RG", "g1") + aux + b"NMi" + struct.pack("<i", 3)
                if rng.random() < 0.2: aux += bamio.tagA("ZS", rng.choice("+-"))
                if with_bc and rng.random() < 0.85: aux += bamio.tagZ("CB", rng.choice(bcs))
                try:
                    recs.append(bamio.record(tid, pos, cig if cig != "*" else "", flag=flag, qname="r%d" % len(recs), aux=aux))
                except Exception:
                    pass
        if rng.random() < 0.3:
            recs.append(bamio.record(-1, -1, "", flag=4, qname="unmapped"))
        p = os.path.join(td, "c.bam")
        try:
            bamio.write_bam(p, contigs, recs, block=rng.choice([0xff00, 700, 3000]))
            synth.index(p)
        except Exception as e:
            continue
        name0 = contigs[0][0]
        args = ["-s", rng.choice(["XS", "RF", "FR"])]
        if rng.random() < 0.5: args += ["-a", str(rng.choice([0, 1, 8, 9, 30]))]
        if rng.random() < 0.4: args += ["-m", str(rng.choice([0, 1, 70, 500]))]
        if rng.random() < 0.4: args += ["-M", str(rng.choice([1, 69, 500000, 3000000]))]
        if rng.random() < 0.2: args += ["-t", "ZS"]
        if rng.random() < 0.4: args += ["-r", rng.choice([name0, "%s:%d-%d" % (name0, rng.randrange(1, 3000), rng.randrange(1, 9000)), "%s:100" % name0, "nope", name0 + ":5-2"])]
        bc_o, bc_r = os.path.join(td, "o.bc"), os.path.join(td, "r.bc")
        for f in (bc_o, bc_r):
            if os.path.exists(f): os.remove(f)
        o = subprocess.run([ORC, "extract"] + args + (["-b", bc_o] if with_bc else []) + ["-o", os.path.join(td, "o.bed"), p], capture_output=True)
        if GPU:
            je = regtools_amd.JunctionsExtractor(ctx=ctx)
            try:
                je.parse_options(args + (["-b", bc_r] if with_bc else []) + [p]); je.identify_junctions_from_BAM(); rc, out = 0, je.bed12()
                if with_bc: open(bc_r, "wb").write(je.barcodes_text(True))
            except regtools_amd.RegtoolsError:
                rc, out = 1, b""
            open(os.path.join(td, "r.bed"), "wb").write(out)
            r = subprocess.CompletedProcess([], rc)
        else:
            r = subprocess.run([REF, "junctions", "extract"] + args + (["-b", bc_r] if with_bc else []) + ["-o", os.path.join(td, "r.bed"), p], capture_output=True)
            if r.returncode not in (0, 1):
                continue
        same = (r.returncode != 0) == (o.returncode != 0) and (r.returncode != 0 or open(os.path.join(td, "r.bed"), "rb").read() == open(os.path.join(td, "o.bed"), "rb").read())
        if same and with_bc and o.returncode == 0:
            same = open(bc_o, "rb").read() == open(bc_r, "rb").read()
        if o.returncode == 0:
            k = sum(1 for _ in open(os.path.join(td, "o.bed"))); rows_total += k; nonempty += k > 0
        if not same:
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", "dfuzz_case_%d.bam" % case) if GPU else "/tmp/dfuzz/case_%d.bam" % case
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            import shutil; shutil.copy(p, keep); shutil.copy(p + ".bai", keep + ".bai")
            print("DISAGREE case %d args %s ref rc %d oracle rc %d -> %s" % (case, args, r.returncode, o.returncode, keep), flush=True)
print("cases %d disagreements %d (non-empty outputs %d, rows %d)" % (n_cases, bad, nonempty, rows_total))
