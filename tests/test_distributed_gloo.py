"""The N > 1 path on CPU: two processes, gloo backend.  Each rank holds one half of a BAM (cut at a record
boundary, i.e. what a coordinate shard is), produces its partial junction table -- here with the CPU oracle,
since no GPU is available in this tier -- and the product's exchange code (regtools_amd.distributed: pack rows,
ONE all-gather, C-ABI merge) must reproduce the single-process result bit for bit."""
import ctypes as C
import os
import struct
import subprocess
import sys

import pytest

import bamio
from conftest import run_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, os, struct, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
from regtools_amd import _ffi, distributed
from test_distributed_gloo import oracle_partial_table
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
parts_dir, strand = sys.argv[1], sys.argv[2]
tab = oracle_partial_table(os.path.join(parts_dir, "part%d.bam" % rank), strand)
merged = distributed.gather_and_merge(tab, min_anchor=8)
open(os.path.join(parts_dir, "merged_rank%d.bed" % rank), "wb").write(merged.bed12())
dist.barrier(); dist.destroy_process_group()
'''


class OrcJunction(C.Structure):
    _fields_ = [("tid", C.c_int32), ("start", C.c_uint32), ("end", C.c_uint32), ("thick_start", C.c_uint32), ("thick_end", C.c_uint32),
                ("read_count", C.c_uint32), ("name_index", C.c_uint64), ("strand", C.c_char), ("left_ok", C.c_uint8), ("right_ok", C.c_uint8),
                ("first_seen", C.c_uint64), ("last_seen", C.c_uint64), ("barcodes", C.c_void_p)]


class OrcTable(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("ref_name", C.POINTER(C.c_char_p)), ("ref_len", C.POINTER(C.c_uint32)), ("n", C.c_size_t),
                ("rows", C.POINTER(OrcJunction)), ("n_records", C.c_uint64), ("n_records_total", C.c_uint64), ("n_events", C.c_uint64),
                ("inflated_bytes", C.c_uint64)]


class OrcParams(C.Structure):
    _fields_ = [("bam", C.c_char_p), ("region", C.c_char_p), ("strandness", C.c_int), ("strand_tag", C.c_char * 2), ("min_anchor", C.c_uint32),
                ("min_intron", C.c_uint32), ("max_intron", C.c_uint32), ("fasta", C.c_char_p), ("barcodes", C.c_int)]


def oracle_partial_table(bam_path, strand):
    """Runs the oracle on one shard file and re-expresses its rows as a product JunctionTable* (via the packed-row
    wire format), carrying first_seen/last_seen exactly as a GPU shard would."""
    from regtools_amd import _ffi
    O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    p = OrcParams()
    O.orc_default_params(C.byref(p))
    p.bam = bam_path.encode(); p.strandness = {"XS": 0, "RF": 1, "FR": 2}[strand]
    t = C.POINTER(OrcTable)()
    err = C.create_string_buffer(256)
    assert O.orc_extract(C.byref(p), C.byref(t), err, 256) == 0, err.value
    tt = t.contents
    raw = b"".join(struct.pack("<12I", r.tid & 0xffffffff, r.start, r.end, r.thick_start, r.thick_end, r.read_count,
                               r.first_seen & 0xffffffff, r.first_seen >> 32, r.last_seen & 0xffffffff, r.last_seen >> 32, ord(r.strand), 0)
                   for r in (tt.rows[i] for i in range(tt.n)))
    proto = _ffi.JunctionTable()
    proto.n_ref, proto.ref_name, proto.ref_len = tt.n_ref, tt.ref_name, tt.ref_len
    out = C.POINTER(_ffi.JunctionTable)()
    buf = (C.c_uint8 * max(1, len(raw))).from_buffer_copy(raw or b"\0")
    assert _ffi.lib().rgx_table_unpack(buf, tt.n, C.byref(proto), C.byref(out)) == 0
    return out


@pytest.mark.parametrize("shape,n,seed,strand", [("fuzz", 30000, 41, "XS"), ("short", 40000, 42, "RF"), ("fuzz", 20000, 43, "FR")])
def test_two_rank_gloo_merge_equals_single_process(tmp_path, shape, n, seed, strand):
    from regtools_amd import synth
    whole = str(tmp_path / "whole.bam")
    synth.write(whole, n, shape=shape, seed=seed)
    contigs, recs = bamio.split_records(bamio.inflate_all(whole))
    cut = len(recs) // 2 + 7
    for r, part in enumerate((recs[:cut], recs[cut:])):
        p = str(tmp_path / ("part%d.bam" % r))
        bamio.write_bam(p, contigs, part)
        synth.index(p)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + seed), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29500 + seed), str(script), str(tmp_path), strand], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    rc, expect, _ = run_oracle(["-s", strand, whole])
    assert rc == 0 and expect.count(b"\n") > 50
    for rank in (0, 1):
        assert open(str(tmp_path / ("merged_rank%d.bed" % rank)), "rb").read() == expect


# ---- -b through the process group: every rank's barcode block travels behind its packed rows --------------------------------------------------
BC_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
from regtools_amd import distributed
from test_distributed_gloo import shard_with_barcodes
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
tab, _, _ = shard_with_barcodes(rank)
merged = distributed.gather_and_merge(tab, min_anchor=8)
open(os.path.join(sys.argv[1], "bc_rank%d.txt" % rank), "wb").write(merged.bed12(False) + b"--\n" + merged.barcodes_text(False))
dist.barrier(); dist.destroy_process_group()
'''


def shard_with_barcodes(rank):
    """A hand-made shard table (three junctions, the middle one only in shard 1's file range ... both shards see the outer two) with its
    barcode lists: (JunctionTable*, packed rows, barcode block).  What a GPU rank holds after `junctions extract -b` on its shard."""
    from regtools_amd import _ffi
    L = _ffi.lib()
    keys = [(0, 100, 200), (0, 300, 450)] if rank == 0 else [(0, 100, 200), (0, 250, 280), (0, 300, 450)]
    lists = ([[("AAAC-1", 3, 0), ("GGTT-1", 1, 1)], [("CCCC-1", 2, 0)]] if rank == 0 else
             [[("GGTT-1", 4, 0), ("TTTT-1", 1, 1)], [("ACGT-1", 1, 0)], [("CCCC-1", 1, 1), ("AAAA-1", 5, 0)]])
    rows = b"".join(struct.pack("<12I", tid, s, e, s - 20, e + 30, sum(c for _, c, _ in bl), 10 * rank + i, 0, 10 * rank + i + 5, 0, ord("+"), 0)
                    for i, ((tid, s, e), bl) in enumerate(zip(keys, lists)))
    proto = _ffi.JunctionTable()
    names = (C.c_char_p * 1)(b"chrB"); lens = (C.c_uint32 * 1)(100000)
    proto.n_ref, proto.ref_name, proto.ref_len = 1, names, lens
    t = C.POINTER(_ffi.JunctionTable)()
    assert L.rgx_table_unpack((C.c_uint8 * len(rows)).from_buffer_copy(rows), len(keys), C.byref(proto), C.byref(t)) == 0
    flat = [x for bl in lists for x in bl]
    row_begin, str_begin, o, so = [0], [0], 0, 0
    for bl in lists:
        o += len(bl); row_begin.append(o)
    for sname, _, _ in flat:
        so += len(sname); str_begin.append(so)
    block = (struct.pack("<3Q", len(keys), len(flat), so) + struct.pack("<%dQ" % len(row_begin), *row_begin) + struct.pack("<%dQ" % len(str_begin), *str_begin) +
             struct.pack("<%dI" % len(flat), *[c for _, c, _ in flat]) + struct.pack("<%dI" % len(flat), *[r for _, _, r in flat]) + "".join(s for s, _, _ in flat).encode())
    assert L.rgx_table_unpack_barcodes(t, (C.c_uint8 * len(block)).from_buffer_copy(block), len(block)) == 0
    return t, (rows, len(keys)), block


def test_two_rank_gloo_merge_carries_barcodes(tmp_path):
    from regtools_amd import distributed
    script = tmp_path / "bc_worker.py"
    script.write_text(BC_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", str(script), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    # what the same two shards give when merged in one process (the semantics of that merge against the reference: tests/test_barcodes.py on the GPU)
    shards = [shard_with_barcodes(k) for k in (0, 1)]
    m = distributed.merge_packed([s[1] for s in shards], shards[0][0], 8, None, [s[2] for s in shards])
    expect = m.bed12(False) + b"--\n" + m.barcodes_text(False)
    # counts of a barcode seen by both shards add up; a junction keeps the barcodes of every shard
    text = m.barcodes_text(False).decode()
    assert "GGTT-1:5" in text and "AAAC-1:3" in text and "TTTT-1:1" in text and "ACGT-1:1" in text and "CCCC-1:3" in text and "AAAA-1:5" in text
    for rank in (0, 1):
        assert open(str(tmp_path / ("bc_rank%d.txt" % rank)), "rb").read() == expect
    # the block is validated where it arrives
    bad = bytearray(shards[1][2]); bad[24] = 7
    with pytest.raises(RuntimeError):
        distributed.merge_packed([s[1] for s in shards], shards[0][0], 8, None, [shards[0][2], bytes(bad)])
