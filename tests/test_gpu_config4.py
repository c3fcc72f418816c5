"""BASELINE configs[3] -- `cis-splice-effects identify` on a 50 M-read BAM + 500 k-variant VCF + GENCODE-scale GTF -- in the GPU suite (round 4).
The reference needs ~6 minutes for that quartet, so byte identity is checked twice removed:
  * at a TENTH of the size (same generator, same seed) the three outputs must have the SHA-256 digests the REAL reference produced in the dev
    container (tests/golden/cse/config4_reduced.json, made by tests/golden/make_golden_config4.py from oracle/_ref; 48 s there);
  * at the full size the size-independent properties must hold: every record decoded, one cis window per splice-relevant variant, the junction
    table unique and in the reference's set order, TSV / BED row for row the same junctions, every variant line carrying its four tags, and
    the same bytes from a second call.
bench.py (identify_config4) times the same workload and compares a 2 M-read sample with the reference run beside it."""
import hashlib
import json
import os

import pytest

import cases

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(cases.GOLD, "cse", "config4_reduced.json")))


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def quartet(tmp, reads, genes, variants, seed):
    from regtools_amd import synth
    pre = os.path.join(str(tmp), "c4")
    st = synth.write(pre + ".bam", reads, shape="short", seed=seed, n_genes=genes)
    ann = synth.annotation(pre, genes, variants, seed=seed, fasta=True)
    return pre, st, ann


def identify(ctx, pre, ann, tag):
    from regtools_amd.cse import CisSpliceEffectsIdentifier
    ci = CisSpliceEffectsIdentifier(ctx=ctx)
    ci.parse_options(["-s", "XS", "-o", pre + tag + ".tsv", "-v", pre + tag + ".vcf", "-j", pre + tag + ".bed", ann["vcf"], pre + ".bam", ann["fasta"], ann["gtf"]])
    ci.identify()
    return dict(ci.stats)


def test_a_tenth_of_config4_has_the_reference_digests(gpu_ctx, tmp_path):
    pre, st, ann = quartet(tmp_path, GOLD["reads"], GOLD["genes"], GOLD["variants"], GOLD["seed"])
    # the generator is deterministic: the text inputs are the bytes the reference saw (the BAM's members may be cut differently by another
    # thread count; its records are the same)
    assert sha(ann["vcf"]) == GOLD["inputs_sha256"]["vcf"] and sha(ann["gtf"]) == GOLD["inputs_sha256"]["gtf"] and sha(ann["fasta"]) == GOLD["inputs_sha256"]["fasta"]
    S = identify(gpu_ctx, pre, ann, ".gpu")
    assert S["n_records"] == GOLD["reads"]
    for ext in ("tsv", "vcf", "bed"):
        p = pre + ".gpu." + ext
        assert os.path.getsize(p) == GOLD["outputs"][ext]["bytes"], ext
        assert sha(p) == GOLD["outputs"][ext]["sha256"], ext


def test_config4_at_full_size(gpu_ctx, tmp_path):
    reads, genes, variants = 50_000_000, 62_500, 500_000
    pre, st, ann = quartet(tmp_path, reads, genes, variants, 4)
    S = identify(gpu_ctx, pre, ann, ".a")
    assert S["n_records"] == st["n_reads"] == reads and S["n_variants"] > 0.99 * variants
    assert S["n_relevant"] == S["n_windows"] > 0 and S["n_pairs"] >= S["n_junctions"] > 0
    tsv = open(pre + ".a.tsv", "rb").read().split(b"\n")
    assert tsv[0].startswith(b"chrom\tstart\tend\tname\tscore\tstrand\tsplice_site") and tsv[-1] == b""
    rows = [l.split(b"\t") for l in tsv[1:-1]]
    assert len(rows) == S["n_junctions"]
    keys = [(r[0], int(r[1]), int(r[2])) for r in rows]
    assert len(set(keys)) == len(keys) and keys == sorted(keys)             # std::set<Junction> order, first insert wins (identifier.cc:292-299)
    assert all(len(r) == 18 and r[17] for r in rows)                        # every junction names the variant(s) whose window holds it
    bed = [l.split(b"\t") for l in open(pre + ".a.bed", "rb").read().split(b"\n")[:-1]]
    assert len(bed) == len(rows)
    for b, r in zip(bed[:5000] + bed[-5000:], rows[:5000] + rows[-5000:]):  # BED12 block geometry of the same junction (the TSV's end is one past the intron's last base)
        sizes = [int(x) for x in b[10].split(b",")[:2]]
        assert b[0] == r[0] and int(b[1]) + sizes[0] == int(r[1]) and int(b[2]) - sizes[1] + 1 == int(r[2]) and b[3] == r[3] and b[4] == r[4] and b[5] == r[5]
    recs = [l for l in open(pre + ".a.vcf", "rb").read().split(b"\n") if l and not l.startswith(b"#")]
    assert len(recs) == S["n_relevant"]
    assert all(all(t in l for t in (b"genes=", b"transcripts=", b"distances=", b"annotations=")) for l in recs)
    S2 = identify(gpu_ctx, pre, ann, ".b")
    assert S2["n_junctions"] == S["n_junctions"]
    for ext in ("tsv", "vcf", "bed"):
        assert open(pre + ".a." + ext, "rb").read() == open(pre + ".b." + ext, "rb").read(), ext
