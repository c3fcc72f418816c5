"""ASan/UBSan run of the annotated-VCF writer (vcf_rewrite.cpp + cse_host.cpp in tests/hostemu, built with -fsanitize=address,undefined)
on mutated VCF / gzip / BCF inputs: byte flips, deleted and duplicated spans, truncation.  No reference involved: the run looks for
memory errors only.  Build line and usage: tools/fuzz/README.md.  Run as
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) python tools/fuzz/asan_vcf_writer.py /tmp/libhostemu_asan.so 3000"""
import ctypes, os, random, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vcf_cases

lib = ctypes.CDLL(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
with tempfile.TemporaryDirectory() as td:
    inputs = vcf_cases.build(td)
    names = sorted(inputs)
    src, dst = os.path.join(td, "in.vcf"), os.path.join(td, "out.vcf")
    err = ctypes.create_string_buffer(512)
    rcs = {}
    for it in range(n):
        # the BGZF container hides a BCF's bytes from the mutations: the raw form is drawn more often
        b = bytearray(inputs[rng.choice(names + ["typed_bcf_raw"] * 6)])
        for _ in range(rng.choice([1, 1, 2, 4, 8])):
            if not b: break
            k = rng.random()
            p = rng.randrange(len(b))
            if k < 0.5: b[p] = rng.choice([0, 9, 10, 13, ord(":"), ord(";"), ord(","), ord("="), ord("."), ord("<"), ord(">"), ord('"'), 255, rng.randrange(256)])
            elif k < 0.65: del b[p:p + rng.randrange(1, 40)]
            elif k < 0.8: b[p:p] = b[p:p + rng.randrange(1, 40)]
            elif k < 0.9: b = b[:p]
            else: b[p:p] = bytes(rng.choice([b"\t", b"\t\t", b":", b";;", b"=", b",,", b"\n", b"\t.\t"]))
        open(src, "wb").write(bytes(b))
        rc = lib.emu_vcf_rewrite(src.encode(), dst.encode(), err, 512)
        rcs[rc] = rcs.get(rc, 0) + 1
    print("runs", n, "exit classes", rcs)
