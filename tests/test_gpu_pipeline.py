"""Several files in flight on one device (rgx_pipeline_create / rgx_extract_submit / rgx_extract_wait, csrc/pipeline.cpp): N interleaved files must be N
sequential rgx_extract_mem calls byte for byte -- different shapes and strand rules, a damaged file and an unreadable one in the middle, tickets waited
for out of order -- and the sequential calls are checked against the oracle."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_cli")
SNAME = {0: "XS", 1: "RF", 2: "FR"}


def _files(tmp_path):
    from regtools_amd import synth
    out = []
    for k, (shape, n, strand) in enumerate([("short", 400_000, 0), ("fuzz", 20_000, 1), ("long", 1_500, 0), ("short", 150_000, 2), ("fuzz", 8_000, 0),
                                            ("short", 900_000, 1), ("long", 700, 2), ("short", 60_000, 0)]):
        bam, bai, st = synth.generate(n, shape=shape, seed=40 + k)
        out.append(dict(bam=bam, bai=bai, strand=strand, kind="ok"))
    # a file cut in the middle of a member (the stream ends there: a shorter table, not an error) and one that is no BAM at all (an error)
    cut = dict(out[0]); cut["bam"] = out[0]["bam"][: len(out[0]["bam"]) * 2 // 3]; cut["kind"] = "cut"
    junk = dict(bam=b"this is not a BAM file, not even a gzip stream" * 10, bai=out[1]["bai"], strand=0, kind="junk")
    out.insert(3, cut); out.insert(6, junk)
    return out


def _sequential(ctx, f):
    import regtools_amd
    je = regtools_amd.JunctionsExtractor(strandness=f["strand"], ctx=ctx)
    try:
        je.identify_junctions_from_BAM(bam_bytes=f["bam"], bai_bytes=f["bai"])
    except regtools_amd.RegtoolsError as e:
        return ("error", e.code, str(e))
    return ("ok", je.bed12(False), je.stats["n_records"])


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_interleaved_files_equal_sequential_calls(gpu_ctx, tmp_path, depth):
    import regtools_amd
    if depth > 2 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 4 * depth:
        pytest.skip("three files in flight want a hardware queue per stream (run by test_launches_at_once_with_a_hardware_queue_per_stream)")
    files = _files(tmp_path)
    want = [_sequential(gpu_ctx, f) for f in files]
    assert [w[0] for w, f in zip(want, files) if f["kind"] == "junk"] == ["error"]
    # the sequential calls against the oracle (whole files only: what a cut file yields is pinned elsewhere, tests/test_gpu_parity.py)
    for k, (w, f) in enumerate(zip(want, files)):
        if f["kind"] != "ok" or k % 3:
            continue
        p = os.path.join(str(tmp_path), "f%d.bam" % k)
        open(p, "wb").write(f["bam"]); open(p + ".bai", "wb").write(f["bai"])
        exp = subprocess.run([ORACLE, "extract", "-s", SNAME[f["strand"]], p], stdout=subprocess.PIPE, check=True).stdout
        je = regtools_amd.JunctionsExtractor(strandness=f["strand"], ctx=gpu_ctx)
        je.identify_junctions_from_BAM(bam_bytes=f["bam"], bai_bytes=f["bai"])
        assert je.bed12() == exp
    pl = regtools_amd.Pipeline(0, depth)
    try:
        for rnd in range(2):                                              # the second round meets warm contexts
            tickets = [pl.submit(bam_bytes=f["bam"], bai_bytes=f["bai"], strandness=f["strand"]) for f in files]
            order = list(range(len(files)))
            if rnd:
                order = order[::-1]                                           # waited for last to first
            got = [None] * len(files)
            for k in order:
                try:
                    je = pl.wait(tickets[k])
                    got[k] = ("ok", je.bed12(False), je.stats["n_records"])
                except regtools_amd.RegtoolsError as e:
                    got[k] = ("error", e.code, str(e))
            assert got == want
        with pytest.raises(regtools_amd.RegtoolsError):
            pl._open[12345] = (None, None, None); pl.wait(12345)              # a ticket nobody was given
    finally:
        pl.close()


def test_pipeline_destroy_with_files_in_flight(gpu_ctx, tmp_path):
    """Tables nobody waited for are released, queued files run to their end (their buffers were promised to the pipeline)."""
    import regtools_amd
    files = [f for f in _files(tmp_path) if f["kind"] == "ok"][:4]
    pl = regtools_amd.Pipeline(0, 2)
    for f in files:
        pl.submit(bam_bytes=f["bam"], bai_bytes=f["bai"], strandness=f["strand"])
    pl.close()
    assert _sequential(gpu_ctx, files[0])[0] == "ok"                          # the device is still usable


def test_more_than_two_files_in_flight_need_more_hardware_queues(gpu_ctx):
    """Three and more contexts on the runtime's default four hardware queues put a file's gated waves and the kernels that release them into one queue
    (every such wave then waits out its time-out): refused unless GPU_MAX_HW_QUEUES says there is a hardware queue per stream (four per file in flight)."""
    import regtools_amd
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 12:
        regtools_amd.Pipeline(0, 3).close()
    else:
        with pytest.raises(regtools_amd.RegtoolsError) as e:
            regtools_amd.Pipeline(0, 3)
        assert "GPU_MAX_HW_QUEUES" in str(e.value)


def test_launches_at_once_with_a_hardware_queue_per_stream():
    """GPU_MAX_HW_QUEUES >= 16 when HIP starts: the pipeline's DEFLATE launches no longer take the chip in turns (api_internal.h LinkTurn::chip_in_turns) and
    three files may be in flight.  The runtime reads the variable once, so the same tests run again in a process of their own."""
    import sys
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 16:
        pytest.skip("this process already runs that way")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "interleaved_files_equal_sequential_calls or more_than_two_files"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-3000:]
