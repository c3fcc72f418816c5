"""How long the cross-shard merge takes at 2/4/8 shards of the bench workload, measured on ONE GPU: the shards' tables are produced one after
the other, packed, laid out in HBM the way the all-gather leaves them, and merged (rgx_table_merge_device).  REGTOOLS_AMD_TRACE=1 prints
the stage split."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import regtools_amd
from regtools_amd import synth, distributed as rd

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = regtools_amd.Context(0)
for world in (2, 4, 8):
    parts, keep = [], []
    for r in range(world):
        bam, bai, st = synth.generate(reads, shape="short", seed=1, slice_index=r, n_slices=world)
        je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
        je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
        parts.append(rd.pack_table(je.table)); keep.append(je)
    stride = max(k for _, k in parts)
    big = torch.zeros(world * stride * rd.ROW, dtype=torch.uint8, device="cuda")
    for g, (b, k) in enumerate(parts):
        big[g * stride * rd.ROW: g * stride * rd.ROW + len(b)].copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
    torch.cuda.synchronize()
    for it in range(3):
        t0 = time.time()
        m = rd.merge_device(ctx, big.data_ptr(), stride, [k for _, k in parts], keep[0].table, 8)
        t1 = time.time()
        print("world %d: merge %.2f ms, %d rows in, %d rows out" % (world, (t1 - t0) * 1e3, sum(k for _, k in parts), m.n), flush=True)
    del m
