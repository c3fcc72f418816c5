"""Soak: one context, files of changing size in rotation (the arena is unmapped, released and mapped anew whenever it grows), device memory watched.
   python tools/soak/arena_soak.py [rounds]   (REGTOOLS_AMD_ARENA=0,2 makes every arena piecewise)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import regtools_amd
from regtools_amd import synth

import torch
def free_mib():
    return torch.cuda.mem_get_info(0)[0] >> 20

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
files = [synth.generate(n, shape="short", seed=s) for n, s in ((3_000, 1), (400_000, 2), (2_500_000, 3), (12_000_000, 4))]
want = [None] * len(files)
ctx = regtools_amd.Context(0)
marks = []
for r in range(rounds):
    for k in ([0, 1, 2, 3] if r % 2 == 0 else [3, 1, 2, 0]):
        bam, bai, st = files[k]
        je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
        je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
        h = hashlib.sha256(je.bed12()).hexdigest()
        assert je.stats["n_records"] == st["n_reads"]
        if want[k] is None: want[k] = h
        assert want[k] == h, (r, k)
    if r in (1, rounds // 2, rounds - 1): marks.append((r, free_mib()))
print("free MiB after rounds", marks, "arena trials", ctx.arena_trials())
ctx.close()
print("free MiB after close", free_mib())
assert abs(marks[0][1] - marks[-1][1]) < 256, marks
