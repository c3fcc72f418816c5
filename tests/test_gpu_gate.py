"""The arrival-gated inflate launch of the overlapped upload (rgx_extract_mem on page-locked host bytes, round 4): ONE k_inflate_coop launch whose waves wait
for the upload chunk their members lie in.  Files of the sizes the rest of the suite uses would not take that path (it starts at 8 MB and 2048
members), so a child process lowers the thresholds (REGTOOLS_AMD_OVERLAP_MIN, REGTOOLS_AMD_INFLATE=coop) and runs synthetic files of every shape,
a truncated file and the reference's golden BAM through it; the parent compares with the oracle."""
import json
import os
import subprocess
import sys

import pytest

import cases
from conftest import ROOT, run_oracle

pytestmark = pytest.mark.gpu

CHILD = r"""
import json, sys
import regtools_amd
from regtools_amd import extractor
jobs = json.load(open(sys.argv[1]))
ctx = regtools_amd.Context(0)
out = []
for j in jobs:
    bam = open(j["bam"], "rb").read(); bai = open(j["bam"] + ".bai", "rb").read()
    pin = regtools_amd.PinnedBuffer(bam)
    je = regtools_amd.JunctionsExtractor(ctx=ctx, **j["kw"])
    res = []
    for rep in range(2):                       # twice: the flags keep the earlier call's epoch
        try:
            je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
            res.append(dict(rc=0, bed=je.bed12().decode("latin1"), n_records=je.stats["n_records"]))
        except extractor.RegtoolsError as e:
            res.append(dict(rc=e.code if hasattr(e, "code") else 1, bed="", n_records=0))
    out.append(res)
    pin.close()
json.dump(out, open(sys.argv[2], "w"))
"""


@pytest.mark.parametrize("chunks", ["2", "7", "16", "16-early-tail"])
def test_gated_launch_equals_the_oracle(gpu_ctx, tmp_path, chunks):
    from regtools_amd import synth
    jobs = []
    early = chunks.endswith("early-tail")
    # (the early tail cuts the member list at multiples of 1024 members: its variant needs a file of a few thousand)
    for shape, n, seed in (("short", 1200000 if early else 120000, 3), ("fuzz", 20000, 4), ("long", 2000, 5)):
        p = str(tmp_path / ("%s.bam" % shape))
        synth.write(p, n, shape=shape, seed=seed)
        jobs.append(dict(bam=p, kw=dict(strandness=0), args=["-s", "XS"]))
        jobs.append(dict(bam=p, kw=dict(strandness=1, min_anchor_length=12), args=["-s", "RF", "-a", "12"]))
    # a file cut in the middle of a member: the stream ends there, as for the sequential reader
    raw = open(jobs[0]["bam"], "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(raw[: len(raw) * 2 // 3])
    open(cut + ".bai", "wb").write(open(jobs[0]["bam"] + ".bai", "rb").read())
    jobs.append(dict(bam=cut, kw=dict(strandness=0), args=["-s", "XS"]))
    jobs.append(dict(bam=os.path.join(cases.GOLD, "test_hcc1395.bam"), kw=dict(strandness=1, min_anchor_length=30), args=["-s", "RF", "-a", "30"]))
    jf, of = str(tmp_path / "jobs.json"), str(tmp_path / "out.json")
    json.dump(jobs, open(jf, "w"))
    # "16-early-tail": the members of the last upload chunks as a second launch, the front part of the arena framed and decoded under it (round 4;
    # REGTOOLS_AMD_EARLY_TAIL_MIN lets files of a few hundred members take that path)
    env = dict(os.environ, REGTOOLS_AMD_OVERLAP_MIN="0", REGTOOLS_AMD_INFLATE="coop", REGTOOLS_AMD_GATE_CHUNKS=chunks.split("-")[0], REGTOOLS_AMD_TRACE="1", PYTHONPATH=ROOT)
    if early:
        env["REGTOOLS_AMD_EARLY_TAIL_MIN"] = "1"
    else:
        env["REGTOOLS_AMD_EARLY_TAIL"] = "0"
    r = subprocess.run([sys.executable, "-c", CHILD, jf, of], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"gated" in r.stderr, "the calls did not take the gated launch:\n" + r.stderr.decode()[-2000:]
    assert (b"early tail" in r.stderr) == early, "early tail taken / not taken against the test's intent:\n" + r.stderr.decode()[-2000:]
    got = json.load(open(of))
    for j, res in zip(jobs, got):
        rc, exp, _ = run_oracle(j["args"] + [j["bam"]])
        for one in res:
            assert (one["rc"] == 0) == (rc == 0), (j["bam"], one["rc"], rc)
            if rc == 0:
                assert one["bed"].encode("latin1") == exp, (j["bam"], j["args"])
