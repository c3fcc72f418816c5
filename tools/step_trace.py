"""One bench-sized extraction with REGTOOLS_AMD_TRACE=1: where the host-visible time of a step goes (stage marks of prepare_events /
reduce / table fill)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import regtools_amd
from regtools_amd import synth
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
bam, bai, st = synth.generate(reads, shape="short", seed=1)
d = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda"); d[:len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8)); torch.cuda.synchronize()
ctx = regtools_amd.Context(0); je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
for it in range(4):
    if it == 3: os.environ["REGTOOLS_AMD_TRACE"] = "1"
    t0 = time.time(); je.identify_junctions_from_BAM(bai_bytes=bai, device_ptr=d.data_ptr(), device_len=len(bam)); t1 = time.time()
    print("step %d: %.2f ms" % (it, (t1 - t0) * 1e3), je.stats["ms_inflate"], je.stats["ms_records"], je.stats["ms_scan"], je.stats["ms_reduce"], flush=True)
