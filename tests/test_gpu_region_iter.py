"""The reference's region iterator on the GPU path (hts_itr_query / hts_itr_next, hts.c:1733-1800, :1924-1965): the index's chunks, one seek
each, and the end rule -- the first record read that lies on another contig or at / behind the region's end finishes the iteration.
 * files with one out-of-order record: the product against the REAL reference's outputs (tests/golden/unsorted, the same ten files that
   pin the oracle in tests/test_unsorted_region.py);
 * files whose index went stale (a member re-compressed to another size after indexing: every later offset of the index points somewhere
   else) and files with an empty member between two chunks: the product against the oracle, which restates the iterator and is pinned to the
   real reference on such files (tools/fuzz/gpu_corrupt_bam.py cpu mode)."""
import os

import pytest

import stale_cases as sc
import unsorted_cases as uc
from conftest import ROOT, run_oracle
from regtools_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "unsorted")
CASES = [(k, r) for k in uc.KINDS for r in uc.REGIONS]


def gpu_extract(ctx, bam, args):
    import regtools_amd
    je = regtools_amd.JunctionsExtractor(ctx=ctx)
    try:
        je.parse_options(list(args) + [bam])
        je.identify_junctions_from_BAM()
    except regtools_amd.RegtoolsError:
        return 1, b"", je
    return 0, je.bed12(), je


def gpu_extract_sharded(ctx, bam, args, n_shards):
    """The same query with the iterator's chunk list dealt to n_shards shards (api_front.cpp: runs of the list, in order) and the shard tables merged:
    what several ranks / devices would do with a region query.  (0, bed12) or (1, b"")."""
    import regtools_amd
    from regtools_amd import distributed
    keep, parts = [], []
    for g in range(n_shards):
        je = regtools_amd.JunctionsExtractor(ctx=ctx, shard=g, n_shards=n_shards)
        try:
            je.parse_options(list(args) + [bam])
            je.identify_junctions_from_BAM()
        except regtools_amd.RegtoolsError:
            return 1, b""
        keep.append(je)
        parts.append(distributed.pack_table(je.table))
    return 0, distributed.merge_packed(parts, keep[0].table, 8).bed12()


@pytest.fixture(scope="module")
def bams(tmp_path_factory):
    d = tmp_path_factory.mktemp("unsorted_gpu")
    return {k: uc.build(str(d / (k + ".bam")), k, synth.index) for k in uc.KINDS}


@pytest.mark.parametrize("kind,region", CASES, ids=["%s-%s" % c for c in CASES])
def test_out_of_order_record_equals_reference(gpu_ctx, bams, kind, region):
    rc, out, _ = gpu_extract(gpu_ctx, bams[kind], ["-s", "XS", "-r", region])
    assert rc == 0
    want = open(os.path.join(GOLD, uc.golden_name(kind, region)), "rb").read()
    assert out == want
    for n in (2, 5):
        assert gpu_extract_sharded(gpu_ctx, bams[kind], ["-s", "XS", "-r", region], n) == (0, want), n


@pytest.mark.parametrize("shape,n,seed", sc.SHAPES)
def test_stale_index_region_queries_equal_oracle(gpu_ctx, tmp_path, shape, n, seed):
    base = str(tmp_path / "base.bam")
    synth.write(base, n, shape=shape, seed=seed)
    bam, bai = open(base, "rb").read(), open(base + ".bai", "rb").read()
    path = str(tmp_path / "case.bam")
    checked = 0
    for case, data, regions in sc.variants(bam, seed, shape):
        open(path, "wb").write(data)
        open(path + ".bai", "wb").write(bai)
        for region in regions:
            args = ["-s", "XS", "-r", region]
            orc_rc, orc_out, _ = run_oracle(args + [path])
            rc, out, _ = gpu_extract(gpu_ctx, path, args)
            assert (rc != 0) == (orc_rc != 0), (case, region)
            if rc == 0:
                assert out == orc_out, (case, region)
                assert gpu_extract_sharded(gpu_ctx, path, args, 3) == (0, orc_out), (case, region)
            checked += 1
    assert checked == 2 * sc.N_VARIANTS


def test_empty_member_between_chunks_ends_one_chunk_only(gpu_ctx, tmp_path):
    """An empty BGZF member reads as the end of the file for the reader that runs into it (bgzf.c:548-578) -- a later chunk is a seek past
    it.  The index is made AFTER the member went in, so it describes the file."""
    base = str(tmp_path / "base.bam")
    synth.write(base, 30000, shape="short", seed=9)
    path = str(tmp_path / "case.bam")
    for case, data in sc.empty_member_variants(open(base, "rb").read()):
        open(path, "wb").write(data)
        if os.path.exists(path + ".bai"):
            os.remove(path + ".bai")
        try:
            synth.index(path)
        except RuntimeError:
            continue                                              # (the indexer itself stops at the empty member: nothing to query)
        for region in sc.EMPTY_REGIONS:
            args = ["-s", "XS", "-r", region]
            orc_rc, orc_out, _ = run_oracle(args + [path])
            rc, out, _ = gpu_extract(gpu_ctx, path, args)
            assert (rc != 0) == (orc_rc != 0), (case, region)
            if rc == 0:
                assert out == orc_out, (case, region)
                assert gpu_extract_sharded(gpu_ctx, path, args, 2) == (0, orc_out), (case, region)
