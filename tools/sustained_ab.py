#!/usr/bin/env python3
"""Files back to back on one device: one call after the other on one context against rgx_pipeline with 1, 2, 3 files in flight (csrc/pipeline.cpp).
   python tools/sustained_ab.py [--reads N] [--files F] [--realistic]"""
import argparse, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")     # (as bench.py: a hardware queue per stream, read by the HIP runtime when it starts)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import regtools_amd  # noqa: E402
from regtools_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=50_000_000)
ap.add_argument("--files", type=int, default=12)
ap.add_argument("--realistic", action="store_true")
ap.add_argument("--depths", default="1,2,3")
a = ap.parse_args()
bam, bai, st = synth.generate(a.reads, shape="short", seed=1, realistic=a.realistic)
pin = regtools_amd.PinnedBuffer(bam)
ctx = regtools_amd.Context(0)
je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
for _ in range(3):
    je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
want = je.bed12()
torch.cuda.synchronize(); t = time.time()
for _ in range(a.files):
    je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
print(json.dumps({"mode": "one context, one call after the other", "ms_per_file": round(1e3 * (time.time() - t) / a.files, 3)}), flush=True)
for depth in [int(x) for x in a.depths.split(",")]:
    pl = regtools_amd.Pipeline(0, depth)

    def run(nf):
        tk = [pl.submit(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam), strandness=0) for _ in range(min(depth, nf))]
        out = None
        for k in range(nf):
            out = pl.wait(tk[k])
            if len(tk) < nf:
                tk.append(pl.submit(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam), strandness=0))
        return out
    run(2 * depth)
    torch.cuda.synchronize(); t = time.time()
    last = run(a.files)
    dt = time.time() - t
    print(json.dumps({"mode": "pipeline", "in_flight": depth, "ms_per_file": round(1e3 * dt / a.files, 3), "alignments_per_s": round(st["n_reads"] * a.files / dt), "same_table": last.bed12() == want}), flush=True)
    pl.close()
