#!/usr/bin/env python3
"""Cold-process wall time of `regtools-amd junctions extract -s XS -o out.bed FILE` next to the reference's on the same file (both from the
page cache), with the process's own breakdown (REGTOOLS_AMD_STATS).   python tools/cli_wall.py [--reads N] [--bam FILE]"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measure(bam, runs=5, with_reference=True, pause_s=0.5):
    """wall_s = the MEDIAN of `runs` cold processes (round 3 quoted the best of three; on the driver's box the three differed 4 x: a process started right
    behind another finds the driver still taking the other's 13 GB of HBM apart -- hence the pause between runs, which is not timed)"""
    out = {"runs": []}
    cli = os.path.join(ROOT, "bin", "regtools-amd")
    bed = bam + ".cli.bed"
    for k in range(runs):
        if k:
            time.sleep(pause_s)
        t0 = time.time()
        r = subprocess.run([cli, "junctions", "extract", "-s", "XS", "-o", bed, bam], env=dict(os.environ, REGTOOLS_AMD_STATS="1"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        dt = time.time() - t0
        line = [l for l in r.stderr.decode().splitlines() if l.startswith("[regtools_amd] process:")]
        out["runs"].append({"wall_s": round(dt, 4), "rc": r.returncode, "breakdown": line[0][len("[regtools_amd] process: "):] if line else None})
    walls = sorted(x["wall_s"] for x in out["runs"])
    out["wall_s"] = walls[len(walls) // 2]
    out["best_wall_s"] = walls[0]
    ref = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
    if with_reference and os.path.exists(ref):
        rbed = bam + ".ref.bed"
        t0 = time.time()
        rr = subprocess.run([ref, "junctions", "extract", "-s", "XS", "-o", rbed, bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out["reference_wall_s"] = round(time.time() - t0, 3)
        out["identical_file"] = bool(rr.returncode == 0 and open(rbed, "rb").read() == open(bed, "rb").read())
        out["process_ratio"] = round(out["reference_wall_s"] / out["wall_s"], 1)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--bam", default=None)
    ap.add_argument("--no-reference", action="store_true")
    a = ap.parse_args()
    bam = a.bam or "/tmp/cli_%d.bam" % a.reads
    if not os.path.exists(bam):
        subprocess.run([os.path.join(ROOT, "bin", "synth_bam"), "write", bam, str(a.reads), "--seed", "1"], check=True, stdout=subprocess.DEVNULL)
    print(json.dumps(measure(bam, with_reference=not a.no_reference), indent=1))
