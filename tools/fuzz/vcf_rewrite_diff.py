"""Differential run of the annotated-VCF writer (regtools_amd/csrc/vcf_rewrite.cpp through tests/hostemu's emu_vcf_rewrite): what it writes
against what the REAL reference writes (oracle/_ref/regtools_ref variants annotate -o, with a GTF that is nowhere near a variant) on
  well  N SEED   random well-formed files (tests/vcf_cases._random_vcf, fresh seeds) and header variations
  mut   N SEED   the inputs of tests/vcf_cases.py with a few bytes flipped / deleted / doubled / inserted (text inputs only; no NUL bytes)
or against another build of the writer (an older libhostemu.so) with every kind of mutation, NUL bytes and BCF included:
  lib   N SEED PATH
Dev container only (the reference binary does not travel).  Prints the first differences and a count."""
import ctypes, os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vcf_cases

REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
mode, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
new = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
other = ctypes.CDLL(sys.argv[4]) if mode == "lib" else None
rng = random.Random(seed)


def run_lib(lib, src, dst):
    if os.path.exists(dst): os.remove(dst)
    err = ctypes.create_string_buffer(512)
    rc = lib.emu_vcf_rewrite(src.encode(), dst.encode(), err, 512)
    return rc, open(dst, "rb").read() if os.path.exists(dst) else b""


def run_ref(src, dst, gtf):
    if os.path.exists(dst): os.remove(dst)
    try:
        r = subprocess.run([REF, "variants", "annotate", "-o", dst, src, gtf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=20)
    except subprocess.TimeoutExpired:
        return None, b""
    return r.returncode, open(dst, "rb").read() if os.path.exists(dst) else b""


HDR_EXTRAS = ['##ALT=<ID=DEL,Description="Deletion">', '##ALT=<ID=DEL,Description="Deletion">', '##source=foo', '##source=bar', '##INFO=<ID=DP,Number=1,Type=Float,Description="again">',
              '##FORMAT=<ID=AF,Number=A,Type=Float,Description="fmt af">', '##FILTER=<ID=DP,Description="a filter named like an INFO">', '##contig=<ID=chrUn,length=77>',
              '##contig=<ID=1>', '##INFO=<ID=NT,Number=1,Description="no type">', '##INFO=<ID=Q1,Number=1,Type=String,Description="a \\"quoted\\" word">',
              '##INFO=<ID=B1,Number=1,Type=String,Description=bare<nested,comma>end>', '##weird=<a=b>', '##weird=<a="b">  ', '##x=<>', '##x=<>', '##fileformat=VCFv4.3',
              '##FORMAT=<ID=GT,Number=1,Type=Integer,Description="int genotype">', '##FORMAT=<ID=FL,Number=0,Type=Flag,Description="flag in format">',
              '##INFO=<ID=I1,Number=1,Type=Integer,Description="with idx",IDX=40>', '##INFO=<ID=I2,Number=1,Type=Integer,Description="bad idx",IDX=4x>', '##FILTER=<ID=q10,Description="again">']


def well_formed(k):
    text = vcf_cases._random_vcf(1000 + seed * 100000 + k, n_rec=rng.choice([5, 25]))
    lines = text.split("\n")
    if rng.random() < 0.7:
        for _ in range(rng.randrange(1, 5)):
            lines.insert(rng.randrange(1, 15), rng.choice(HDR_EXTRAS))
    return "\n".join(lines).encode()


def mutate(b, allow_nul):
    b = bytearray(b)
    for _ in range(rng.choice([1, 1, 2, 4])):
        if not b: break
        k, p = rng.random(), rng.randrange(len(b))
        pool = [9, 10, ord(":"), ord(";"), ord(","), ord("="), ord("."), ord("<"), ord(">"), ord('"'), ord("|"), ord("/"), ord(" "), ord("-"), ord("1"), ord("e"), ord("\\")]
        if allow_nul: pool += [0, 7, 13, 255, rng.randrange(256)]
        if k < 0.5: b[p] = rng.choice(pool)
        elif k < 0.65: del b[p:p + rng.randrange(1, 30)]
        elif k < 0.8: b[p:p] = b[p:p + rng.randrange(1, 30)]
        else: b[p:p] = bytes(rng.choice([b"\t", b"\t\t", b":", b";;", b"=", b",,", b"\n", b"\t.\t", b":.", b"./."]))
    return bytes(b)


bad = 0
with tempfile.TemporaryDirectory() as td:
    inputs = vcf_cases.build(td)
    texts = sorted(k for k in inputs if not k.endswith("_gz") and "bcf" not in k and not k.startswith("tbi"))
    src, d1, d2, gtf = (os.path.join(td, x) for x in ("in.vcf", "new.vcf", "other.vcf", "far.gtf"))
    open(gtf, "w").write(vcf_cases.GTF_FAR)
    for k in range(n):
        if mode == "well": data = well_formed(k)
        elif mode == "mut": data = mutate(inputs[rng.choice(texts)], False)
        else: data = mutate(inputs[rng.choice(sorted(inputs) + ["typed_bcf_raw"] * 4)], True)
        open(src, "wb").write(data)
        rc_new, out_new = run_lib(new, src, d1)
        rc_o, out_o = run_lib(other, src, d2) if other else run_ref(src, d2, gtf)
        if rc_o is None or (other is None and rc_o not in (0, 1)):
            continue                                                   # the reference hung or died of a signal: nothing to compare with
        if (rc_new != 0) != (rc_o != 0) or (rc_new == 0 and out_new != out_o):
            bad += 1
            if bad <= 40:                                              # (the first five inputs are kept, the first forty differences shown)
                keep = os.path.join(tempfile.gettempdir(), "vcf_diff_%s_%d_%d.vcf" % (mode, seed, k)) if bad <= 5 or os.environ.get("KEEP_ALL") else None
                if keep: open(keep, "wb").write(data)
                print("DIFF case", k, "rc", rc_new, rc_o, "kept", keep)
                a, b = out_new.split(b"\n"), out_o.split(b"\n")
                for i in range(max(len(a), len(b))):
                    x, y = (a[i] if i < len(a) else None), (b[i] if i < len(b) else None)
                    if x != y:
                        print("  new  :", x); print("  other:", y); break
print("runs", n, "differences", bad)
