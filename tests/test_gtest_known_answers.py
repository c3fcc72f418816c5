"""The known answers the reference's own unit test holds for JunctionsExtractor (/root/reference/tests/lib/junctions/test_junctions_extractor.cc:75-141):
AddJunction -- five junction events in this order: (10000,10200) seen three times with thick bounds 9900-10300, 9500-10200, 9950-10700 on '+', then
(8000,8500) 7000-10000 on '+' and the same junction on '-' -- must print three rows, sorted, with names in order of first occurrence (JUNC00000002 and
..3 for the two strands of the second junction, ..1 with count 3 and the min / max thick bounds for the first); JunctionName -- names start at
JUNC00000001; PrintJunction -- the BED12 block arithmetic.  The gtest calls add_junction directly; here five reads produce exactly those events through
the CIGAR walk (a CIGAR that ENDS in N gives the event whose thick_end equals its end: SURVEY 9.3), in the test's order (the file is not sorted: whole-file
iteration is file order).  The expected text is the gtest's, typed out.  CPU: the oracle and, where present, the real reference; GPU: the product."""
import os
import subprocess

import pytest

import bamio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_cli")
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")


def bed12(chrom, ts, te, name, count, strand, start, end):
    # Junction::print as the gtest spells it (test_junctions_extractor.cc:93-98, 118-138)
    return "%s\t%d\t%d\t%s\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n" % (chrom, ts, te, name, count, strand, ts, te, start - ts, te - end, end - ts)


EXPECTED_ADD_JUNCTION = (bed12("chr1", 7000, 10000, "JUNC00000002", 1, "+", 8000, 8500) +
                         bed12("chr1", 7000, 10000, "JUNC00000003", 1, "-", 8000, 8500) +
                         bed12("chr1", 9500, 10700, "JUNC00000001", 3, "+", 10000, 10200)).encode()
# the row the gtest's PrintJunction expects, name and count as the extractor would give a lone read
EXPECTED_ONE = bed12("chr1", 9500, 10700, "JUNC00000001", 1, "+", 10000, 10200).encode()


def write_events(path, events):
    recs = []
    for k, (start, end, ts, te, strand) in enumerate(events):
        cigar = "%dM%dN" % (start - ts, end - start) + ("%dM" % (te - end) if te > end else "")
        recs.append(bamio.record(0, ts, cigar, qname="r%d" % k, aux=bamio.tagA("XS", strand)))
    bamio.write_bam(path, [("chr1", 1000000)], recs)
    from regtools_amd import synth
    synth.index(path)
    return path


ADD_JUNCTION = [(10000, 10200, 9900, 10300, "+"), (10000, 10200, 9500, 10200, "+"), (10000, 10200, 9950, 10700, "+"),
                (8000, 8500, 7000, 10000, "+"), (8000, 8500, 7000, 10000, "-")]


def test_oracle_and_reference_print_the_gtest_text(built, tmp_path):
    p = write_events(os.path.join(str(tmp_path), "add.bam"), ADD_JUNCTION)
    q = write_events(os.path.join(str(tmp_path), "one.bam"), [(10000, 10200, 9500, 10700, "+")])
    for exe in [ORACLE] + ([REF] if os.path.exists(REF) else []):
        cmd = [exe] + (["junctions"] if exe == REF else []) + ["extract", "-s", "XS"]
        assert subprocess.run(cmd + [p], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout == EXPECTED_ADD_JUNCTION, exe
        assert subprocess.run(cmd + [q], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout == EXPECTED_ONE, exe


@pytest.mark.gpu
def test_product_prints_the_gtest_text(gpu_ctx, tmp_path):
    import regtools_amd
    p = write_events(os.path.join(str(tmp_path), "add.bam"), ADD_JUNCTION)
    je = regtools_amd.JunctionsExtractor(bam=p, strandness=0, ctx=gpu_ctx)
    je.identify_junctions_from_BAM()
    assert je.bed12() == EXPECTED_ADD_JUNCTION
    rows = je.get_all_junctions()
    assert [(j.name, j.read_count, j.strand, j.thick_start, j.thick_end) for j in rows] == [
        ("JUNC00000002", 1, "+", 7000, 10000), ("JUNC00000003", 1, "-", 7000, 10000), ("JUNC00000001", 3, "+", 9500, 10700)]
    q = write_events(os.path.join(str(tmp_path), "one.bam"), [(10000, 10200, 9500, 10700, "+")])
    je = regtools_amd.JunctionsExtractor(bam=q, strandness=0, ctx=gpu_ctx)
    je.identify_junctions_from_BAM()
    assert je.bed12() == EXPECTED_ONE                                  # JunctionName: the first name is JUNC00000001; PrintJunction: the block arithmetic
