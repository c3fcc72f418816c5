import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import regtools_amd
args = sys.argv[2:]
je = regtools_amd.JunctionsExtractor(ctx=regtools_amd.Context(0))
try:
    je.parse_options(args + [sys.argv[1]]); je.identify_junctions_from_BAM(); print("rc 0 rows", je.bed12().count(b"\n"))
except regtools_amd.RegtoolsError as e:
    print("rc 1", e)
