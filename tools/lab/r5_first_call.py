"""tools/lab/r5_first_call.py: what a context's first calls cost (the arena's allocation, the placement trials), wall ms of calls 1..4 on the 50 M-read bench file from page-locked memory"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import regtools_amd
from regtools_amd import synth
bam, bai, st = synth.generate(50_000_000, shape="short", seed=1, threads=32)
pin = regtools_amd.PinnedBuffer(bam)
ctx = regtools_amd.Context(0)
je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
ts = []
for k in range(4):
    t = time.time(); je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam)); ts.append(round(1e3 * (time.time() - t), 1))
print(os.environ.get("REGTOOLS_AMD_ARENA", "default"), "calls ms", ts, "trials", [round(x, 2) for x in ctx.arena_trials()])
