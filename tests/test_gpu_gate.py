"""The arrival-gated inflate launch of the overlapped upload (rgx_extract_mem on page-locked host bytes, round 4): ONE k_inflate_coop launch whose waves wait
for the upload chunk their members lie in.  Files of the sizes the rest of the suite uses would not take that path (it starts at 8 MB and 2048
members), so a child process lowers the thresholds (REGTOOLS_AMD_OVERLAP="min_bytes,chunks[,early_min_members]", REGTOOLS_AMD_INFLATE=coop) and runs synthetic files of every shape,
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


@pytest.mark.parametrize("chunks", ["2", "7", "16", "16-early-tail", "16-early-tail-7cuts"])
def test_gated_launch_equals_the_oracle(gpu_ctx, tmp_path, chunks):
    from regtools_amd import synth
    jobs = []
    early = "early-tail" in chunks
    seven = chunks.endswith("7cuts")           # (round 5: up to seven cuts; the parts are multiples of 1,024 members, so the file has eleven thousand)
    # (the early tail cuts the member list at multiples of 1024 members: its variant needs a file of a few thousand)
    for shape, n, seed in (("short", 3200000 if seven else 1200000 if early else 120000, 3), ("fuzz", 20000, 4), ("long", 2000, 5)):
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
    # the third number of REGTOOLS_AMD_OVERLAP lets files of a few hundred members take that path)
    env = dict(os.environ, REGTOOLS_AMD_OVERLAP="0," + chunks.split("-")[0] + (",1" if early else ""), REGTOOLS_AMD_INFLATE="coop", REGTOOLS_AMD_TRACE="1", PYTHONPATH=ROOT)
    if not early:
        env["REGTOOLS_AMD_EARLY_TAIL"] = "0"
    if seven:
        env["REGTOOLS_AMD_EARLY_TAIL"] = "2,4,6,8,10,12,14"
    r = subprocess.run([sys.executable, "-c", CHILD, jf, of], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"gated" in r.stderr, "the calls did not take the gated launch:\n" + r.stderr.decode()[-2000:]
    if seven:
        assert b"early tail: part 5" in r.stderr, "fewer parts than the test means to run:\n" + r.stderr.decode()[-2000:]
    assert (b"early tail" in r.stderr) == early, "early tail taken / not taken against the test's intent:\n" + r.stderr.decode()[-2000:]
    got = json.load(open(of))
    for j, res in zip(jobs, got):
        rc, exp, _ = run_oracle(j["args"] + [j["bam"]])
        for one in res:
            assert (one["rc"] == 0) == (rc == 0), (j["bam"], one["rc"], rc)
            if rc == 0:
                assert one["bed"].encode("latin1") == exp, (j["bam"], j["args"])


def test_early_tail_with_records_longer_than_its_margin(gpu_ctx, tmp_path):
    """The early tail frames and decodes the front of the arena while the members behind it still inflate; its margin is one member (64 KiB).
    A record that STARTS in the prefix and ends more than a member behind it (a CIGAR of tens of thousands of operations) must not be decoded
    from bytes that are not there yet: the call has to notice (the verified chain's exit from the prefix) and take the one-pass order.  A
    short-read file with one ~190 KB record laid across every boundary an early part can end on (multiples of 1,024 members: every member of
    the file inflates to 0xff00 bytes, so those boundaries are known arena offsets)."""
    import struct
    import bamio
    from regtools_amd import synth
    src = str(tmp_path / "short.bam")
    synth.write(src, 1_200_000, shape="short", seed=9)
    inflated = bamio.inflate_all(src)
    contigs, _ = bamio.split_records(inflated[: 1 << 16] + b"")       # (header only: the record list of the cut-off copy is not used)
    hdr_len = len(bamio.header_bytes(contigs))
    # upstream's header text may differ from bamio's: measure the header the file has
    l_text = struct.unpack_from("<i", inflated, 4)[0]
    q = 12 + l_text
    for _ in range(struct.unpack_from("<i", inflated, 8 + l_text)[0]):
        q += 8 + struct.unpack_from("<i", inflated, q)[0]
    hdr_len = q
    n_ops = 10000
    cigar = [(8, 0), (75, 3)] * n_ops + [(8, 0)]
    block, targets = 0xff00, []
    n_hdr_members = (hdr_len + block - 1) // block
    out, off, k = [], hdr_len, 1
    # bamio.write_bam cuts the header into its own members and the record stream every 0xff00 bytes: member 1024 k starts at this stream offset
    boundary = lambda kk: (1024 * kk - n_hdr_members) * block
    pos, n_long, prev = hdr_len, 0, None
    while pos + 4 <= len(inflated):
        bs = struct.unpack_from("<i", inflated, pos)[0]
        rec = inflated[pos: pos + 4 + bs]
        stream_off = off - hdr_len
        if prev is not None and stream_off >= boundary(k) - 150_000:
            tid, p0 = struct.unpack_from("<ii", prev, 4)
            long_rec = bamio.record(tid, p0, cigar, flag=99, qname="long%04d" % k, aux=bamio.tagA("XS", "+"))
            assert len(long_rec) > 150_000 + 16384          # starts in the prefix (a member and a segment in front of the boundary), ends behind the boundary
            out.append(long_rec); off += len(long_rec); n_long += 1; k += 1
        out.append(rec); off += len(rec); prev = rec
        pos += 4 + bs
    assert n_long >= 3, n_long
    p = str(tmp_path / "longtail.bam")
    bamio.write_bam(p, contigs, out, level=1)
    synth.index(p)
    jobs = [dict(bam=p, kw=dict(strandness=0), args=["-s", "XS"]), dict(bam=src, kw=dict(strandness=0), args=["-s", "XS"])]
    jf, of = str(tmp_path / "jobs.json"), str(tmp_path / "out.json")
    json.dump(jobs, open(jf, "w"))
    env = dict(os.environ, REGTOOLS_AMD_OVERLAP="0,16,1", REGTOOLS_AMD_INFLATE="coop", REGTOOLS_AMD_TRACE="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD, jf, of], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"early tail: part" in r.stderr, r.stderr.decode()[-2000:]
    assert b"behind the inflated part" in r.stderr, "no long record lay across an early part's end: the test's file misses its purpose\n" + r.stderr.decode()[-3000:]
    got = json.load(open(of))
    for j, res in zip(jobs, got):
        rc, exp, _ = run_oracle(j["args"] + [j["bam"]])
        assert rc == 0
        for one in res:
            assert one["rc"] == 0 and one["bed"].encode("latin1") == exp, j["bam"]
