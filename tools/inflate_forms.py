#!/usr/bin/env python3
"""Time the forms of the DEFLATE kernel (rgx_k_inflate_form: 1 = one member per lane, 2 = one member per wave, 3 = lane + LDS window) on
prefixes of a synthetic file's member list: where does the wave form stop paying?   python tools/inflate_forms.py [--realistic]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from regtools_amd import _ffi, synth  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bamio  # noqa: E402


def main():
    realistic = "--realistic" in sys.argv
    bam, _, _ = synth.generate(3_000_000, shape="short", seed=7, realistic=realistic)
    members, upos = [], 0
    for off, payload, isize in bamio.bgzf_members(bam):
        members.append((off + 18, upos, len(payload), isize)); upos += isize
    arr = (_ffi.Member * len(members))(*[_ffi.Member(*m) for m in members])
    d_comp = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda"); d_comp[: len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8))
    d_mem = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    d_arena = torch.zeros(upos + 512, dtype=torch.uint8, device="cuda")
    d_status = torch.tensor([0xffffffff, 0], dtype=torch.int64).to(torch.uint32).cuda()
    L = _ffi.lib()
    ref = None
    for n in (16, 64, 256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, len(members)):
        n = min(n, len(members))
        row = {"members": n}
        for form in (1, 2, 3):
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                L.rgx_k_inflate_form(form, d_comp.data_ptr(), d_mem.data_ptr(), n, d_arena.data_ptr() + 256, d_status.data_ptr(), None)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            row["ms_form%d" % form] = round(best, 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
