"""the host CLI keeps the reference's subcommands, option letters and exit codes.  Every argument list here
ends before any device work, so the test runs without a GPU; where the real reference binary is present (dev container,
oracle/_ref) the exit codes are compared with it directly, elsewhere with the codes recorded from it below."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "regtools-amd")
REF = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
VCF, FA, GTF = (os.path.join(GOLD, "cse_ref", x) for x in ("test1.vcf", "test_chr22.fa", "test_ensemble_chr22.2.gtf"))
BAM = os.path.join(GOLD, "cse_ref", "test_hcc1395.2.bam")
BED = os.path.join(GOLD, "annot_ref", "junctions_extract.bed")

# (argv, exit code of the reference)   -- test_regtools_main.py, test_junctions_main.py, test_junctions_extract.py:87-109,
# test_cis_splice_effects_identify.py, test_variants_main.py, test_cis_splice_effects_associate.py:56-60
CASES = [
    ([], 0), (["-h"], 0), (["nonsense"], 0),
    (["junctions"], 0), (["junctions", "-h"], 0), (["junctions", "extract", "-h"], 0), (["junctions", "annotate", "-h"], 0),
    (["junctions", "extract", "-s", "XS", "-o", "/dev/null"], 1),                       # no BAM
    (["junctions", "extract", "-o", "/dev/null", BAM], 1),                              # no strandness
    (["junctions", "extract", "-s", "sideways", BAM], 1),
    (["junctions", "extract", "-s", "intron-motif", BAM], 1),                           # needs a FASTA
    (["junctions", "extract", "-Q", "-s", "XS", BAM], 1),                               # unknown option
    (["junctions", "annotate"], 1), (["junctions", "annotate", BED, FA], 1),
    (["variants"], 0), (["variants", "-h"], 0), (["variants", "annotate", "-h"], 0), (["variants", "annotate"], 1), (["variants", "annotate", VCF], 1),
    (["cis-splice-effects"], 0), (["cis-splice-effects", "-h"], 0), (["cis-splice-effects", "identify", "-h"], 0), (["cis-splice-effects", "associate", "-h"], 0),
    (["cis-splice-effects", "identify", VCF, BAM, FA, GTF], 1),                         # no -s
    (["cis-splice-effects", "identify", "-s", "XS", VCF, BAM, FA], 1),
    (["cis-splice-effects", "identify", "-s", "XS", VCF, BAM, FA, "/no/such.gtf"], 1),
    (["cis-splice-effects", "identify", "-s", "up", VCF, BAM, FA, GTF], 1),
    (["cis-splice-effects", "associate", VCF, BED, FA], 1),
    (["cis-splice-effects", "associate", VCF, "/no/such.bed", FA, GTF], 1),
    (["cis-splice-effects", "associate", "-s", "XS", VCF, BED, FA, GTF], 1),            # -s is not an option of associate
]


def run(exe, argv):
    return subprocess.run([exe] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120).returncode


@pytest.mark.parametrize("argv,ref_rc", CASES, ids=[" ".join(os.path.basename(a) for a in c[0]) or "(none)" for c in CASES])
def test_exit_codes_match_the_reference(built, argv, ref_rc):
    if os.path.exists(REF):
        assert run(REF, argv) == ref_rc, "the recorded reference exit code is stale"
    assert run(EXE, argv) == ref_rc
