"""rgx_table_merge_device on more rows than the small-tile radix sort takes (round 5, advisor finding of round 4): N > 2 M packed rows whose
unique keys number U <= 2 M.  The key sort of the N rows runs in 2,048-key tiles, the sorts of the U unique rows in 512-key tiles -- four
times the histogram words per key -- out of ONE scratch area sized for N: radix_tmp_words must bound every sort it is used for.  The rows are
synthetic (48-byte packed rows, rgx_table_pack's layout) and the device merge must print what the host merge (rgx_table_merge, its own code
path) prints for the same parts."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def make_part(g, lo, n, rng):
    k = np.arange(lo, lo + n, dtype=np.int64)
    rows = np.zeros((n, 12), dtype=np.uint32)
    start = 1000 + 37 * k
    end = start + 100 + (k % 50)
    rows[:, 0] = k % 23
    rows[:, 1] = start
    rows[:, 2] = end
    rows[:, 3] = start - 8 - (k * 3 + g * 5) % 20
    rows[:, 4] = end + 8 + (k * 7 + g * 11) % 20
    rows[:, 5] = 1 + (k * 7 + g) % 9
    first = rng.permutation(n).astype(np.int64) * 3 + 1                 # record index of the key's first event in this shard: unique
    last = first + 1 + (k % 5)
    rows[:, 6] = first & 0xffffffff
    rows[:, 8] = last & 0xffffffff
    rows[:, 10] = np.where(k % 3 == 0, ord("-"), ord("+"))                     # (the strand class is part of the key: a function of the key alone)
    rank = np.empty(n, dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(1, n + 1)
    rows[:, 11] = rank                                                  # name_index: first-seen rank inside the shard
    # a shard's rows come in its print order (chrom string rank, thick_start, thick_end, name): any fixed order will do for a merge
    return rows


def test_device_merge_of_more_than_two_million_rows(gpu_ctx):
    import torch
    import regtools_amd
    from regtools_amd import distributed
    # a table to take the contig names from (23 contigs)
    je = regtools_amd.JunctionsExtractor(bam=os.path.join(cases.GOLD, "test_hcc1395.bam"), strandness=1, ctx=gpu_ctx)
    je.identify_junctions_from_BAM()
    assert je.table.contents.n_ref >= 23
    rng = np.random.default_rng(5)
    G, per, step = 4, 600_000, 150_000                                   # N = 2.4 M rows, U = 1.05 M unique keys
    parts = [make_part(g, g * step, per, rng) for g in range(G)]
    packed = [(p.tobytes(), per) for p in parts]
    assert G * per > 2 * 1024 * 1024 and 3 * step + per <= 2 * 1024 * 1024
    host = distributed.merge_packed(packed, je.table, 8)
    assert host.n == 3 * step + per
    big = torch.zeros(G * per * distributed.ROW, dtype=torch.uint8, device="cuda")
    for g, (b, k) in enumerate(packed):
        big[g * per * distributed.ROW: (g + 1) * per * distributed.ROW].copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
    torch.cuda.synchronize()
    dev = distributed.merge_device(gpu_ctx, big.data_ptr(), per, [per] * G, je.table, 8)
    assert dev.n == host.n
    assert dev.bed12(False) == host.bed12(False)
    assert dev.bed12() == host.bed12()
    # ... and a second shape: U just above a quarter of N (where the old sizing was first too small), N barely above the tile switch
    G2, per2, step2 = 3, 700_000, 30_000
    parts2 = [make_part(g, g * step2, per2, rng) for g in range(G2)]
    packed2 = [(p.tobytes(), per2) for p in parts2]
    host2 = distributed.merge_packed(packed2, je.table, 8)
    big2 = torch.zeros(G2 * per2 * distributed.ROW, dtype=torch.uint8, device="cuda")
    for g, (b, k) in enumerate(packed2):
        big2[g * per2 * distributed.ROW: (g + 1) * per2 * distributed.ROW].copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
    torch.cuda.synchronize()
    dev2 = distributed.merge_device(gpu_ctx, big2.data_ptr(), per2, [per2] * G2, je.table, 8)
    assert dev2.n == host2.n == 2 * step2 + per2 and dev2.bed12(False) == host2.bed12(False)
