import sys, time, os, socket
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
import regtools_amd
from regtools_amd import synth, distributed as rd
bam, bai, st = synth.generate(50_000_000, shape="short", seed=1)
d = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda"); d[:len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8)); torch.cuda.synchronize()
ctx = regtools_amd.Context(0); je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
for it in range(4):
    t0 = time.time(); je.identify_junctions_from_BAM(bai_bytes=bai, device_ptr=d.data_ptr(), device_len=len(bam)); t1 = time.time()
    m = rd.gather_and_merge(je, min_anchor=8); torch.cuda.synchronize(); t2 = time.time()
    print("extract %.1f ms, gather+merge %.1f ms (rows %d)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, m.n))
dist.destroy_process_group()
