"""BASELINE configs[2] -- "synthetic 400M-read BAM coordinate-sharded across 8xMI355X, RCCL junction-count reduce" (SURVEY 8d "Config 3", 8e;
the caller replaced: junctions_main.cc:45-59) -- in the GPU suite (round 5).  One GPU is what the suite has, so the eight shards take turns
on it; every other line of the multi-GPU paths is the one an 8-GPU node runs:
  * at a TENTH of the size (8 slices x 5 M reads, the generator and seed of bench.py's ranks) the merged table of the eight shards must have
    the SHA-256 the REAL reference produced on the 40 M-read file the slices make when they are joined (tests/golden/config2_sharded.json,
    made by tests/golden/make_golden_config2_sharded.py from oracle/_ref) -- through every merge path: the host merge of eight extractions,
    the device merge (rgx_table_merge_device), rgx_extract_multi over the device list [0]*8, and one device reading the joined file;
  * at the FULL size (8 x 50 M = 400 M reads, 4.3 GB of BGZF, 88.5 GB inflated) (a) rgx_extract_multi_mem with the device list [0]*8 on the
    joined file and (b) bench.py --gpus 8 under torchrun (eight ranks over gloo on this GPU: every rank extracts its slice, all-gather of
    packed rows, merge) must both give the bytes ONE device gives for the joined file, all 400 M records decoded, names a permutation of
    1..n, rows in compare_junctions' order, supporting reads conserved."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

import cases
import slices

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(cases.GOLD, "config2_sharded.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def make_slices(reads, n, seed):
    from regtools_amd import synth
    parts = [synth.generate(reads, shape="short", seed=seed, slice_index=k, n_slices=n) for k in range(n)]
    assert all(p[2]["n_reads"] == reads for p in parts)
    return parts


def table_properties(rows, n_events):
    assert sum(j.read_count for j in rows) == n_events                                        # supporting reads conserved
    assert sorted(int(j.name[4:]) for j in rows) == list(range(1, len(rows) + 1))             # first-seen names: a permutation of 1..n
    keys = [(j.chrom, j.thick_start, j.thick_end, j.name) for j in rows]
    assert keys == sorted(keys)                                                               # compare_junctions (junctions_extractor.h:117-140)
    assert len({(j.chrom, j.start, j.end, j.strand in "+-" and j.strand) for j in rows}) == len(rows)


def test_a_tenth_of_config2_eight_shards_have_the_reference_digest(gpu_ctx):
    import torch
    import regtools_amd
    from regtools_amd import distributed
    n, reads = GOLD["n_slices"], GOLD["reads_per_slice"]
    parts = make_slices(reads, n, GOLD["seed"])
    packed, keep, n_rec, n_ev = [], [], 0, 0
    for bam, bai, _ in parts:                                       # what every rank of an 8-GPU job does with its slice
        je = regtools_amd.JunctionsExtractor(strandness=0, ctx=gpu_ctx)
        je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
        keep.append(je); packed.append(distributed.pack_table(je.table)); n_rec += je.stats["n_records"]; n_ev += je.stats["n_events"]
    assert n_rec == GOLD["reads"] == n * reads
    merged = distributed.merge_packed(packed, keep[0].table, 8)
    bed = merged.bed12()
    assert (len(bed), bed.count(b"\n")) == (GOLD["bed12"]["bytes"], GOLD["bed12"]["lines"])
    assert sha(bed) == GOLD["bed12"]["sha256"]                      # == the real reference on the joined 40 M-read file
    # the same rows resident in HBM, as the RCCL all-gather leaves them
    stride = max(k for _, k in packed)
    big = torch.zeros(n * stride * distributed.ROW, dtype=torch.uint8, device="cuda")
    for g, (b, k) in enumerate(packed):
        big[g * stride * distributed.ROW: g * stride * distributed.ROW + len(b)].copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
    torch.cuda.synchronize()
    assert sha(distributed.merge_device(gpu_ctx, big.data_ptr(), stride, [k for _, k in packed], keep[0].table, 8).bed12()) == GOLD["bed12"]["sha256"]
    del big, keep, merged
    # the joined file: one device, and the C++ host over eight shards of it (member ranges cut at index record starts)
    bam = slices.concat_slices([p[0] for p in parts])
    bai = slices.merge_bai([p[0] for p in parts], [p[1] for p in parts])
    assert sha(bai) == GOLD["bai_sha256"]
    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=gpu_ctx)
    je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
    assert je.stats["n_records"] == GOLD["reads"] and je.stats["n_events"] == n_ev and sha(je.bed12()) == GOLD["bed12"]["sha256"]
    m = regtools_amd.extract_multi([0] * n, bam_bytes=bam, bai_bytes=bai, strandness=0)
    assert m.table.contents.n_records == GOLD["reads"] and sha(m.bed12()) == GOLD["bed12"]["sha256"]


def test_config2_at_full_size_400M_reads_in_eight_shards(gpu_ctx, tmp_path):
    import regtools_amd
    n, reads, seed = 8, 50_000_000, 1
    parts = make_slices(reads, n, seed)
    bam = slices.concat_slices([p[0] for p in parts])
    bai = slices.merge_bai([p[0] for p in parts], [p[1] for p in parts])
    del parts
    pin = regtools_amd.PinnedBuffer(bam)
    n_bytes = len(bam)
    del bam
    # ONE device reads the joined 400 M-read file (88.5 GB inflated, resident in its 288 GB) -- in a context of its own, closed before the
    # eight shards' contexts are made: the session's context keeps the workspaces of the tests before it, and ~110 GB more would not fit
    # next to 8 x 14 GB
    big_ctx = regtools_amd.Context(0)
    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=big_ctx)
    je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=n_bytes)
    single = je.bed12()
    assert je.stats["n_records"] == n * reads
    table_properties(je.get_all_junctions(), je.stats["n_events"])
    n_events = je.stats["n_events"]
    del je
    big_ctx.close()
    # (b) bench.py's N = 8 path (first: the eight contexts rgx_extract_multi makes stay with this process, and eight rank processes of 15 GB each want the room)
    # bench.py's N = 8 path: eight ranks (gloo, all on this GPU), each with its own slice; the line checks itself (records conserved, the
    # collective's table == an independent host merge of the ranks' tables, counts conserved) and the table it dumps must be `single`
    bed8 = str(tmp_path / "ranks8.bed")
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", "29633",
                        os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--reads", str(reads), "--seed", str(seed),
                        "--no-cpu-baseline", "--no-extras", "--dump-bed", bed8], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n and sum(x["n_records"] for x in line["multi_gpu"]["per_rank"]) == n * reads
    ck = line["multi_gpu"]["checks"]
    assert ck["records_conserved"] and ck["bed12_equals_independent_merge"] and ck["counts_conserved"] and ck["supporting_reads"] == n_events
    assert open(bed8, "rb").read() == single
    # (a) the C++ host: eight shards of that file on the device list [0]*8 (rgx_extract_multi_mem: a thread and a context per listed device)
    m = regtools_amd.extract_multi([0] * n, bai_bytes=bai, host_ptr=pin.ptr, host_len=n_bytes, strandness=0)
    assert m.table.contents.n_records == n * reads and m.table.contents.n_events == n_events
    assert m.bed12() == single
    del m
    pin.close()
