"""`cis-splice-effects identify` on a BAM whose record stream ENDS somewhere (a member that does not inflate, a file cut short): upstream reads every variant's
window through the index on its own (cis_splice_effects_identifier.cc:288-290), so the windows BEHIND the damage still see their reads, and a window that runs
into the damage keeps what it read before.  One pass over the file stops at the damage; for such files the product reads every window by its own region
extraction (cse_api.cpp window_join_by_seeks).  CPU: the oracle against the real reference; GPU: the tool's two output files against the oracle's, also with the
extraction sharded."""
import os
import shutil
import subprocess

import pytest

import bamio
import cse_synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_cli")
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
EXE = os.path.join(ROOT, "bin", "regtools-amd")
KINDS = ["member_2", "member_mid", "member_late", "cut_mid", "two_members"]


def build(td, kind):
    q = cse_synth.build(os.path.join(td, "q"), seed=5, n_genes=10)
    bam = open(q["bam"], "rb").read()
    mem = list(bamio.bgzf_members(bam))
    b = bytearray(bam)

    def wreck(k):
        coff = mem[k][0]
        for j in range(40, 60):
            b[coff + 18 + j] ^= 0xff                      # the member's DEFLATE stream no longer inflates

    if kind == "member_2":
        wreck(2)
    elif kind == "member_mid":
        wreck(len(mem) // 2)
    elif kind == "member_late":
        wreck(len(mem) - 3)
    elif kind == "two_members":
        wreck(3); wreck(len(mem) - 4)
    elif kind == "cut_mid":
        b = b[: mem[len(mem) // 2][0] + 200]              # the file ends inside a member
    p = os.path.join(td, kind + ".bam")
    open(p, "wb").write(bytes(b))
    shutil.copy(q["bam"] + ".bai", p + ".bai")
    return q, p


def run(exe, q, bam, td, tag, extra=(), env=None):
    tsv, bed = os.path.join(td, tag + ".tsv"), os.path.join(td, tag + ".bed")
    sub = ["cis-splice-effects", "identify"] if exe != ORACLE else ["identify"]
    r = subprocess.run([exe] + sub + ["-s", "XS"] + list(extra) + ["-o", tsv, "-j", bed, q["vcf"], bam, q["fasta"], q["gtf"]], stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL, timeout=300, env=env)
    return r.returncode, open(tsv, "rb").read() if os.path.exists(tsv) else None, open(bed, "rb").read() if os.path.exists(bed) else None


@pytest.mark.parametrize("kind", KINDS)
def test_oracle_follows_the_reference(built, tmp_path, kind):
    td = str(tmp_path)
    q, bam = build(td, kind)
    got = run(ORACLE, q, bam, td, "oracle")
    assert got[0] == 0 and got[1] is not None
    whole = run(ORACLE, q, q["bam"], td, "whole")
    assert whole[1] != got[1], "the damage changes nothing: the case is idle"
    if os.path.exists(REF):
        assert got == run(REF, q, bam, td, "ref")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_product_reads_every_window_as_upstream_does(built, tmp_path, kind):
    td = str(tmp_path)
    q, bam = build(td, kind)
    want = run(ORACLE, q, bam, td, "oracle")
    assert run(EXE, q, bam, td, "tool") == want
    assert run(EXE, q, bam, td, "tool_w", extra=["-w", "300"]) == run(ORACLE, q, bam, td, "oracle_w", extra=["-w", "300"])
    assert run(EXE, q, bam, td, "sharded", env=dict(os.environ, REGTOOLS_AMD_DEVICES="0,0,0")) == want
