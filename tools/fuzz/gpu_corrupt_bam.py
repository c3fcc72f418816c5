"""Whole-pipeline robustness on an MI355X: mutated BAM files (bit flips in compressed payloads, BSIZE / ISIZE / gzip header bytes, record
fields inside re-compressed members, truncation) through the C-ABI, each compared with the oracle (exit status and BED12 bytes).
Run under `timeout`; prints one line per disagreement."""
import os, random, struct, subprocess, sys, tempfile, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamio
import regtools_amd
from regtools_amd import synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CPU = len(sys.argv) > 3 and sys.argv[3] == "cpu"      # no GPU: the ORACLE judged by the real reference on the same damaged files
ctx = None if CPU else regtools_amd.Context(0)
ORACLE = os.path.join(ROOT, "oracle", "oracle_cli")
REFBIN = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
bad = 0
with tempfile.TemporaryDirectory() as td:
    bases = []
    for shape, n, seed in (("short", 20000, 3), ("fuzz", 8000, 4), ("long", 150, 5)):
        p = os.path.join(td, "base_%s.bam" % shape)
        synth.write(p, n, shape=shape, seed=seed)
        bases.append((open(p, "rb").read(), open(p + ".bai", "rb").read()))
    for case in range(n_cases):
        bam, bai = rng.choice(bases)
        b = bytearray(bam)
        members = list(bamio.bgzf_members(bam))
        kind = rng.choice(["payload", "payload", "bsize", "isize", "magic", "truncate", "record", "crc", "dup_eof"])
        mi = rng.randrange(1 if len(members) > 2 else 0, len(members))          # usually not the header member
        coff, payload, isz = members[mi]
        if kind == "payload":
            for _ in range(rng.choice([1, 1, 4])):
                b[coff + 18 + rng.randrange(max(1, len(payload)))] ^= 1 << rng.randrange(8)
        elif kind == "bsize":
            struct.pack_into("<H", b, coff + 16, rng.choice([0, 1, 25, 26, 27, rng.randrange(65536), struct.unpack_from("<H", b, coff + 16)[0] ^ 1]))
        elif kind == "isize":
            struct.pack_into("<I", b, coff + 18 + len(payload) + 4, rng.choice([0, 1, (isz - 1) & 0xffffffff, isz + 1, 65536, 65537, 0xffffffff]))
        elif kind == "magic":
            hb = rng.choice([0, 1, 2, 3, 10, 12, 13, 14])
            b[coff + hb] ^= rng.choice([1, 0x80, 0xff])
            if hb > 2:
                kind = "bgzf_extra"      # gzip magic intact, BGZF extra field not: upstream falls back to plain-gzip decoding: crash check only
        elif kind == "truncate":
            b = b[:rng.randrange(30, len(b))]
        elif kind == "crc":
            b[coff + 18 + len(payload) + rng.randrange(4)] ^= 0xff               # the reference never checks the CRC
        elif kind == "dup_eof":
            b[coff:coff] = bamio.EOF_MARKER                                      # an empty member in the middle ends the stream
        elif kind == "record":
            raw = bytearray(zlib.decompress(payload, -15))
            if len(raw) > 64:
                k = rng.randrange(len(raw) - 40)
                raw[k:k + 4] = struct.pack("<I", rng.choice([0, 31, 32, 33, 1 << 27, (1 << 27) + 1, 0x7fffffff, 0xffffffff, rng.randrange(1 << 16)]))
                newm = bamio.bgzf_member(bytes(raw))
                bl = struct.unpack_from("<H", b, coff + 16)[0] + 1
                b[coff:coff + bl] = newm
        path = os.path.join(td, "case.bam")
        open(path, "wb").write(bytes(b)); open(path + ".bai", "wb").write(bai)
        args = rng.choice([["-s", "XS"], ["-s", "RF", "-a", "3"], ["-s", "XS", "-r", rng.choice(["chr1", "1", "chr2:1-90000000", "10:1000-200000"])]]
                          + ([["-s", "RF", "-b", os.path.join(td, "bc.txt")]] if os.environ.get("FUZZ_BARCODES") else []))   # (-b: set_junction_barcode's walk over damaged aux fields)
        orc = subprocess.run([ORACLE, "extract"] + args + [path], capture_output=True)
        if CPU and kind == "bgzf_extra":
            continue
        if CPU:
            bed = os.path.join(td, "ref.bed")
            try:
                rr = subprocess.run([REFBIN, "junctions", "extract"] + args + ["-o", bed, path], capture_output=True, timeout=20)
            except subprocess.TimeoutExpired:
                print("note: case %d kind %s args %s: the reference does not finish in 20 s (oracle rc %d)" % (case, kind, args, orc.returncode), flush=True)
                continue
            if rr.returncode in (0, 1) and ((rr.returncode != 0) != (orc.returncode != 0) or (rr.returncode == 0 and open(bed, "rb").read() != orc.stdout)):
                bad += 1
                keep = "/tmp/dfuzz/corrupt_%d.bam" % case
                os.makedirs("/tmp/dfuzz", exist_ok=True)
                open(keep, "wb").write(bytes(b)); open(keep + ".bai", "wb").write(bai)
                print("DISAGREE case %d kind %s member %d args %s: reference rc %d, oracle rc %d rows %d -> %s" % (case, kind, mi, args, rr.returncode, orc.returncode, orc.stdout.count(b"\n"), keep), flush=True)
            continue
        if os.environ.get("FUZZ_VERBOSE"):                                  # a crash of the process must leave its input behind
            keep = os.path.join(ROOT, "gpurun_out", "last_case.bam")
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            open(keep, "wb").write(bytes(b)); open(keep + ".bai", "wb").write(bai)
            open(os.path.join(ROOT, "gpurun_out", "last_case.txt"), "w").write("case %d kind %s member %d args %s\n" % (case, kind, mi, args))
        je = regtools_amd.JunctionsExtractor(ctx=ctx)
        try:
            je.parse_options(args + [path]); je.identify_junctions_from_BAM(); rc, out = 0, je.bed12()
        except regtools_amd.RegtoolsError as e:
            rc, out = 1, b""
        if "-r" not in args and rc == 0 and rng.random() < 0.3:
            # the same file cut into shards (multi-GPU path): the merged shards must give the same bytes
            from regtools_amd import distributed
            G = rng.choice([2, 3, 5])
            parts, keep = [], []
            try:
                for g in range(G):
                    js = regtools_amd.JunctionsExtractor(ctx=ctx, shard=g, n_shards=G)
                    js.parse_options(args + [path]); js.identify_junctions_from_BAM()
                    keep.append(js); parts.append(distributed.pack_table(js.table))
                merged = distributed.merge_packed(parts, keep[0].table, int(args[args.index("-a") + 1]) if "-a" in args else 8,
                                                  [k.stats["stream_ended"] for k in keep]).bed12()
            except regtools_amd.RegtoolsError:
                merged = None
            if merged != out:
                bad += 1
                print("DISAGREE case %d kind %s member %d args %s: %d shards merged differ from the single pass" % (case, kind, mi, args, G), flush=True)
        if "-r" in args and kind == "record":
            # (a re-compressed member moves every later byte: the index no longer belongs to the file; let the real reference judge, where it is built)
            ref = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
            if not os.path.exists(ref):
                continue
            bed = os.path.join(td, "ref.bed")
            try:
                rr = subprocess.run([ref, "junctions", "extract"] + args + ["-o", bed, path], capture_output=True, timeout=20)
            except subprocess.TimeoutExpired:
                continue
            if rr.returncode not in (0, 1):
                continue                                               # the reference itself died (abort / segfault): nothing to compare with
            orc = subprocess.CompletedProcess([], rr.returncode, open(bed, "rb").read() if rr.returncode == 0 else b"", b"")
        if kind == "bgzf_extra":
            continue
        if (rc != 0) != (orc.returncode != 0) or (rc == 0 and out != orc.stdout):
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_case_%d.bam" % case)
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            open(keep, "wb").write(bytes(b)); open(keep + ".bai", "wb").write(bai)
            print("DISAGREE case %d kind %s member %d args %s: gpu rc %d rows %d, oracle rc %d rows %d" % (case, kind, mi, args, rc, out.count(b"\n"), orc.returncode, orc.stdout.count(b"\n")), flush=True)
print("cases %d, disagreements %d" % (n_cases, bad))
