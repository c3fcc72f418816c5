"""Soak: files of four sizes in rotation through rgx_pipeline (page-locked inputs: the arrival-gated launch), every table hashed against a sequential call's,
every file's time watched -- a gated wave that waits out its 2 s time-out, or a call that fails over to another path, shows as a slow file.
   GPU_MAX_HW_QUEUES=32 python tools/soak/pipeline_soak.py [files_per_depth] [depths]     (default 240 files, depths 2,3)"""
import hashlib, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import regtools_amd
from regtools_amd import synth

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 240
depths = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,3").split(",")]
files = []
ctx = regtools_amd.Context(0)
for n, s in ((400_000, 2), (2_500_000, 3), (12_000_000, 4), (30_000_000, 5)):
    bam, bai, st = synth.generate(n, shape="short", seed=s)
    pin = regtools_amd.PinnedBuffer(bam)
    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
    je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
    assert je.stats["n_records"] == st["n_reads"]
    files.append(dict(pin=pin, n=len(bam), bai=bai, want=hashlib.sha256(je.bed12()).hexdigest(), reads=st["n_reads"]))
ctx.close()
for depth in depths:
    pl = regtools_amd.Pipeline(0, depth)
    order = [(k * 7 + k // 5) % 4 for k in range(n_files)]
    tk, done_at, t0 = [], [], time.time()
    for k in range(min(depth, n_files)):
        f = files[order[k]]; tk.append(pl.submit(bai_bytes=f["bai"], host_ptr=f["pin"].ptr, host_len=f["n"], strandness=0))
    for k in range(n_files):
        je = pl.wait(tk[k]); done_at.append(time.time())
        f = files[order[k]]
        assert je.stats["n_records"] == f["reads"] and hashlib.sha256(je.bed12()).hexdigest() == f["want"], (depth, k)
        if len(tk) < n_files:
            g = files[order[len(tk)]]; tk.append(pl.submit(bai_bytes=g["bai"], host_ptr=g["pin"].ptr, host_len=g["n"], strandness=0))
    gaps = [1e3 * (b - a) for a, b in zip([t0] + done_at[:-1], done_at)]
    reads = sum(files[o]["reads"] for o in order)
    print(json.dumps({"in_flight": depth, "GPU_MAX_HW_QUEUES": os.environ["GPU_MAX_HW_QUEUES"], "files": n_files, "every_table_identical": True,
                      "alignments_per_s": round(reads / (done_at[-1] - t0)), "ms_between_results": {"median": round(sorted(gaps)[len(gaps) // 2], 2),
                      "max": round(max(gaps), 2), "over_100": sum(g > 100 for g in gaps)}}), flush=True)
    pl.close()
