#!/usr/bin/env python3
"""Time the lane forms of the DEFLATE kernel (rgx_k_inflate_form 1 = k_inflate, 4 = k_inflate_coop, optionally 3 = ring) on one synthetic
file's whole member list, and check that they leave the same bytes in the arena (and zlib's, on a sample of members).
   python tools/inflate_bench.py [--reads N] [--realistic] [--shape long] [--forms 1,4]"""
import argparse
import ctypes as C
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from regtools_amd import _ffi, synth  # noqa: E402


def member_table(bam):
    """(cpos, upos, clen, isize) per BGZF member, vectorised enough for 170 k members"""
    out, off, upos, n = [], 0, 0, len(bam)
    mv = memoryview(bam)
    while off + 18 <= n:
        bl = (mv[off + 16] | mv[off + 17] << 8) + 1
        isz = int.from_bytes(mv[off + bl - 4: off + bl], "little")
        out.append((off + 18, upos, bl - 26, isz)); upos += isz; off += bl
    return out, upos


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--realistic", action="store_true")
    ap.add_argument("--shape", default="short")
    ap.add_argument("--forms", default="1,4")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--bam", default=None, help="cache file: read when it exists, else generated and written there")
    ap.add_argument("--prefix", type=float, default=100.0, help="percent of the member list to launch (occupancy sweeps)")
    a = ap.parse_args()
    if a.bam and os.path.exists(a.bam):
        bam = open(a.bam, "rb").read(); st = {"n_reads": a.reads}
    else:
        bam, _, st = synth.generate(a.reads, shape=a.shape, seed=1, realistic=a.realistic)
        if a.bam:
            open(a.bam, "wb").write(bam)
    members, upos = member_table(bam)
    if a.prefix < 100.0:
        members = members[: max(64, int(len(members) * a.prefix / 100.0))]
        upos = members[-1][1] + members[-1][3]
    arr = np.array(members, dtype=np.uint64)
    packed = np.zeros((len(members), 3), dtype=np.uint64)
    packed[:, 0] = arr[:, 0]; packed[:, 1] = arr[:, 1]; packed[:, 2] = arr[:, 2] | (arr[:, 3] << np.uint64(32))
    d_mem = torch.from_numpy(packed.view(np.uint8).reshape(-1)).cuda()
    d_comp = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda")
    d_comp[: len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8))
    d_status = torch.tensor([0xffffffff, 0], dtype=torch.int64).to(torch.uint32).cuda()
    L = _ffi.lib()
    alg = sum(m[2] for m in members) + upos
    sums = {}
    for form in [int(f) for f in a.forms.split(",")]:
        d_arena = torch.zeros(upos + 1024, dtype=torch.uint8, device="cuda")
        times = []
        for rep in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            rc = L.rgx_k_inflate_form(form, d_comp.data_ptr(), d_mem.data_ptr(), len(members), d_arena.data_ptr() + 256, d_status.data_ptr(), None)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0
            times.append(e0.elapsed_time(e1))
        ok = d_status.cpu().tolist()[0] == 0xffffffff
        # checksum of the arena (int64 sums over 8-byte words, position-weighted would be better but this catches any byte loss) + a zlib sample
        words = d_arena[256: 256 + (upos // 8) * 8].view(torch.int64)
        sums[form] = (int(words.sum().item()), int((words[::97] * 3).sum().item()), int(d_arena[:256].sum().item()), int(d_arena[256 + upos:].sum().item()))
        bad = 0
        for k in range(0, len(members), max(1, len(members) // 40)):
            cpos, up, clen, isz = members[k]
            want = zlib.decompress(bytes(bam[cpos: cpos + clen]), -15)
            got = d_arena[256 + up: 256 + up + isz].cpu().numpy().tobytes()
            bad += want != got
        print(json.dumps({"form": form, "members": len(members), "ms_min": round(min(times), 3), "ms_all": [round(t, 3) for t in times], "status_ok": ok,
                          "GBps_alg": round(alg / min(times) / 1e6, 1), "frac_of_8TBps": round(alg / min(times) / 1e6 / 8000, 4),
                          "zlib_sample_mismatches": bad, "guard_bytes_touched": sums[form][2] + sums[form][3]}), flush=True)
        del d_arena
    vals = list(sums.values())
    print(json.dumps({"arena_checksums_equal": all(v[:2] == vals[0][:2] for v in vals), "workload": {"reads": st["n_reads"], "realistic": a.realistic, "shape": a.shape, "C": len(bam), "U": upos}}))


if __name__ == "__main__":
    main()
