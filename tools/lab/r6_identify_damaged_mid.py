"""GPU-box script (round 6, last session): `identify` on a damaged mid-size quartet (2 M reads, 8,000 transcripts, 20,000 variants; a member in the middle wrecked) --
the windows read one by one (window_join_by_seeks) against the oracle, and what that costs next to the undamaged file's one pass."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamio
from regtools_amd import synth
EXE = os.path.join(ROOT, "bin", "regtools-amd"); ORC = os.path.join(ROOT, "oracle", "oracle_cli")
with tempfile.TemporaryDirectory() as td:
    pre = os.path.join(td, "m")
    synth.write(pre + ".bam", 2_000_000, shape="short", seed=7, n_genes=2000)
    ann = synth.annotation(pre, 2000, 20000, seed=7, fasta=True)
    raw = bytearray(open(pre + ".bam", "rb").read())
    mem = list(bamio.bgzf_members(bytes(raw)))
    coff = mem[len(mem) // 2][0]
    for j in range(40, 80):
        raw[coff + 18 + j] ^= 0xff
    open(pre + ".dam.bam", "wb").write(bytes(raw))
    import shutil
    shutil.copy(pre + ".bam.bai", pre + ".dam.bam.bai")
    res = {}
    for who, exe, sub in (("oracle", ORC, ["identify"]), ("tool", EXE, ["cis-splice-effects", "identify"])):
        for name in ("", ".dam"):
            out = [os.path.join(td, "%s%s.%s" % (who, name, x)) for x in ("tsv", "bed")]
            t = time.time()
            r = subprocess.run([exe] + sub + ["-s", "XS", "-o", out[0], "-j", out[1], ann["vcf"], pre + name + ".bam", ann["fasta"], ann["gtf"]], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, env=dict(os.environ, REGTOOLS_AMD_STATS="1"))
            res[who + name] = (r.returncode, [open(f, "rb").read() if os.path.exists(f) else None for f in out])
            print("%-12s rc %d  %.2f s  tsv %d lines  %s" % (who + name, r.returncode, time.time() - t, res[who + name][1][0].count(b"\n") if res[who + name][1][0] else -1,
                  [l for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("[regtools_amd]")][-1:]), flush=True)
    print("undamaged: tool == oracle", res["tool"] == res["oracle"], "  damaged: tool == oracle", res["tool.dam"] == res["oracle.dam"],
          "  damage changes the output", res["oracle"] != res["oracle.dam"])
    assert res["tool"] == res["oracle"] and res["tool.dam"] == res["oracle.dam"]
