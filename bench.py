#!/usr/bin/env python
"""bench.py -- `junctions extract` hot path on N MI355X GPUs (one process per GPU).

A "step" is one complete pass of the hot path over one synthetic BAM (SURVEY.md 8d, config 2 shape):
BGZF file bytes in page-locked host memory -> chunked upload, overlapped with -> inflate -> record framing -> SoA
decode -> CIGAR scan/emit -> radix sort + segmented reduce -> sorted junction table on the host.  (`value` is timed
on that region; `value_device_resident` is the same pass with the file already in HBM, reported beside it.)  N > 1 is weak scaling: every rank
holds its own coordinate slice (same read count) of one big coordinate-sorted BAM, the per-rank tables are
exchanged with one RCCL all-gather of packed rows and merged.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel: the
DEFLATE kernel, timed with HIP events on the pipeline's own stream) and `cpu_baseline` (the real
reference built into oracle/_ref when present, else the oracle port; rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A hardware queue per stream for the pass with two files in flight (two contexts x four streams; the HIP runtime reads this once, when it starts -- hence
# here, in front of the first import of torch): on the default four queues the streams of the two files share queues pairwise and a kernel behind another
# stream's launch waits for it -- 22.1-22.8 ms per file against 20.2-21.5 on sixteen (profiles/r06_pipeline_hw_queues_ab.txt).  The timed step (`value`) does
# not move with it (24.7-25.6 ms on either).  Thirty-two rather than sixteen: a rank of an N > 1 job has torch's and RCCL's streams beside the pipeline's ten, and streams that
# share a queue again are what made three files in flight on sixteen queues take 90 ms per file (two in flight: 19.7-20.1 ms on thirty-two, 19.8-20.3 on sixteen).
# An environment that sets the variable is left alone.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# the whole-file launch of the DEFLATE kernel the pipeline runs above 2048 members (kernels.hip launch_inflate; REGTOOLS_AMD_INFLATE overrides)
INFLATE_KERNEL = {"lane": "rgx::k_inflate<false, false, 1>", "ring": "rgx::k_inflate_ring<false>", "wave": "rgx::k_inflate_wave"}.get(
    os.environ.get("REGTOOLS_AMD_INFLATE", ""), "rgx::k_inflate_coop<false, false, 1>")


def inflate_kernel_for(compressed, inflated):
    """kernels.h inflate_plan_for / launch_inflate: k_inflate_coop with the windowed bit reader for every payload class (round 4)"""
    return INFLATE_KERNEL


def committed_traffic(key, kernel, alg_bytes):
    """HBM-side traffic of a workload's DEFLATE launch from the round's committed rocprofv3 PMC passes (profiles/r04_pmc_traffic.json, made by
    tools/pmc_traffic_r4.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of the launch alone on the same file).  A committed measurement, not
    one of this run -- only quoted when kernel and algorithmic bytes are this run's."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))[key]
        if pm.get("kernel") != kernel or abs(pm["algorithmic_bytes"] - alg_bytes) > 1e-3 * alg_bytes:
            return {"traffic": None}
        t = (pm["FETCH_SIZE_KiB"] + pm["WRITE_SIZE_KiB"]) * 1024.0
        return {"traffic": (2 * pm["FETCH_SIZE_KiB"] + pm["WRITE_SIZE_KiB"]) * 1024.0, "traffic_uncorrected": t,
                "traffic_source": "profiles/r04_pmc_traffic.json[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this kernel on this workload, tools/pmc_traffic_r4.sh; "
                                  "a committed measurement, not taken in this run)" % key,
                "traffic_note": "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 = %.1f x the algorithmic bytes (the guide's gfx950 correction); uncorrected %.1f x (the x2 is "
                                "calibrated for wide coalesced streams -- profiles/r03_pmc_tail_kernels.json -- not for this kernel's 16-byte requests of 64 lanes in 64 lines: the truth lies between)"
                                % (((2 * pm["FETCH_SIZE_KiB"] + pm["WRITE_SIZE_KiB"]) * 1024.0) / alg_bytes, t / alg_bytes)}
    except Exception:
        return {"traffic": None}


def live_traffic(bam, n_reads, kernel, alg_bytes, extra_args=()):
    """HBM-side traffic of the DEFLATE launch measured IN THIS RUN: two separate `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE -- they do not fit
    one pass, MI355X_MICROARCH.md PMC slots) over tools/inflate_bench.py, which launches the same kernel symbol on this very file through the stage entry
    point (rgx_k_inflate_form 4) and checks a sample of members against zlib.  Units and correction as the guide's HBM section prescribes: the counters
    are KiB, and on gfx950 FETCH_SIZE tallies 128-byte fabric requests at 64 bytes -- doubled before it is compared with a byte count.  None when
    rocprofv3 is not on the PATH or a pass fails (the caller then quotes the committed measurement and says so)."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    short = kernel.split("::")[-1].split("<")[0]
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            path = os.path.join(td, "w.bam")
            with open(path, "wb") as f:
                f.write(bam)
            vals = {}
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(td, ctr)
                cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable,
                       os.path.join(ROOT, "tools", "inflate_bench.py"), "--reads", str(n_reads), "--forms", "4", "--reps", "2", "--bam", path] + list(extra_args)
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
                got = []
                for fcsv in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(fcsv)):
                        if short in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                            got.append(float(row["Counter_Value"]))
                if r.returncode != 0 or not got:
                    return None
                vals[ctr] = sum(got) / len(got)
        fetch, write = vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
        return {"traffic": 2.0 * fetch + write, "traffic_uncorrected": fetch + write, "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                "traffic_source": "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over tools/inflate_bench.py launching %s on this file "
                                  "(average of its launches)" % kernel,
                "traffic_note": "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 = %.1f x the algorithmic bytes -- gfx950's FETCH_SIZE counts 128-byte requests at 64 bytes "
                                "(MI355X_MICROARCH.md, HBM); uncorrected %.1f x.  The doubling is calibrated on wide coalesced streams (profiles/r03_pmc_tail_kernels.json); this "
                                "kernel's reads are 16-byte requests of 64 lanes in 64 lines, so the truth lies between the two" % ((2 * fetch + write) / alg_bytes, (fetch + write) / alg_bytes)}
    except Exception:
        return None


def cpu_baseline(bam_path, n_reads, n_events):
    """Time the reference (or the oracle port) on the host cores: single thread, because the reference has
    no threads at all."""
    ref = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
    orc = os.path.join(ROOT, "oracle", "oracle_cli")
    out = os.path.join(os.path.dirname(bam_path), "cpu_baseline.bed")
    if os.path.exists(ref):
        t0 = time.time()
        r = subprocess.run([ref, "junctions", "extract", "-s", "XS", "-o", out, bam_path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        if r.returncode == 0 and dt > 0:
            cb = dict(value=n_reads / dt, unit="alignments/s", cores=1, kind="reference", seconds=round(dt, 3),
                      junction_events_per_s=n_events / dt,
                      sample="the full bench workload (%d reads), reference regtools built from /root/reference by oracle/Makefile, 1 thread" % n_reads)
            cb["all_host_cores"] = reference_on_all_cores(ref, bam_path, n_reads)
            return cb, out
    if not os.path.exists(orc):
        subprocess.run(["make", "-s"], cwd=os.path.join(ROOT, "oracle"), check=True)
    t0 = time.time()
    subprocess.run([orc, "extract", "-s", "XS", "-o", out, bam_path], check=True)
    dt = time.time() - t0
    return dict(value=n_reads / dt, unit="alignments/s", cores=1, kind="port", seconds=round(dt, 3), junction_events_per_s=n_events / dt,
                sample="the full bench workload (%d reads), oracle/ C restatement, 1 thread" % n_reads), out


def reference_on_all_cores(ref, bam_path, n_reads):
    """The reference has no threads; the most a node's host cores can do with it is one process per contig (`-r chrN`, each
    seeking through the .bai).  Reported next to the single-thread figure: aggregate alignments/s = all reads / slowest-finish."""
    from concurrent.futures import ThreadPoolExecutor
    contigs = ["chr%d" % i for i in range(1, 23)] + ["chrX"]
    cores = os.cpu_count() or 1
    try:                                   # (a container's CPU quota: the GPU boxes show 256 hardware threads and grant 16 cores)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, -(-int(q) // int(per))))
    except Exception:
        pass
    workers = max(1, min(len(contigs), cores))

    def one(c):
        return subprocess.run([ref, "junctions", "extract", "-s", "XS", "-r", c, "-o", os.devnull, bam_path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode

    t0 = time.time()
    with ThreadPoolExecutor(workers) as ex:
        rcs = list(ex.map(one, contigs))
    dt = time.time() - t0
    if any(rcs) or dt <= 0:
        return None
    return dict(value=n_reads / dt, unit="alignments/s", processes=len(contigs), cores=workers, seconds=round(dt, 3),
                note="one reference process per contig (-r), run %d at a time (the cores the container's CPU quota grants)" % workers)


def run_reference(argv, timeout=600):
    ref = os.path.join(ROOT, "oracle", "_ref", "regtools_ref")
    if not os.path.exists(ref):
        return None, None
    t0 = time.time()
    try:
        rc = subprocess.run([ref] + argv, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout).returncode
    except subprocess.TimeoutExpired:
        return None, float(timeout)
    return rc, time.time() - t0


def extra_extract(regtools_amd, synth, ctx, label, shape, realistic, n_reads, sample_reads, seed, threads, steps=3):
    """One more `junctions extract` workload beside the headline: the same timed region (page-locked host bytes -> table), the DEFLATE kernel's
    roofline on its own algorithmic bytes (C + U of THIS file), and the reference timed on a bounded sample of the same shape
    (its output compared byte for byte with the GPU's on that sample)."""
    import torch
    t_gen = time.time()
    bam, bai, st = synth.generate(n_reads, shape=shape, seed=seed, threads=threads, realistic=realistic)
    t_gen = time.time() - t_gen
    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
    pin = regtools_amd.PinnedBuffer(bam)
    je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
    torch.cuda.synchronize()
    ms = 1e3 * (time.time() - t0) / steps
    n_events = je.stats["n_events"]
    d_bam = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda")
    d_bam[: len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8))
    torch.cuda.synchronize()
    k_ms, stage = [], None
    for _ in range(2):
        je.identify_junctions_from_BAM(bai_bytes=bai, device_ptr=d_bam.data_ptr(), device_len=len(bam))
        k_ms.append(je.stats["ms_inflate"])
        stage = {k: round(je.stats["ms_" + k], 3) for k in ("inflate", "records", "scan", "reduce", "total")}
    s = je.stats
    alg = s["compressed_bytes"] + s["inflated_bytes"]
    out = {"workload": label, "reads": st["n_reads"], "ms": ms, "alignments_per_s": st["n_reads"] / (ms * 1e-3), "junction_events_per_s": n_events / (ms * 1e-3),
           "ms_device_resident": stage["total"], "stage_ms": stage, "input_generation_s": round(t_gen, 2),
           "bytes_per_alignment": {"compressed": s["compressed_bytes"] / st["n_reads"], "inflated": s["inflated_bytes"] / st["n_reads"]},
           "roofline": {"bound": "hbm", "kernel": inflate_kernel_for(s["compressed_bytes"], s["inflated_bytes"]), "kernel_ms": min(k_ms), "algorithmic_bytes": alg,
                        "achieved": alg / (min(k_ms) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (min(k_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    out["roofline"].update(committed_traffic("long10M" if shape == "long" else "realistic" if realistic else "default", out["roofline"]["kernel"], alg))
    del d_bam, bam
    pin.close()
    # the reference on a bounded sample of the same shape (the full file would take it minutes)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "sample.bam")
        ss = synth.write(p, sample_reads, shape=shape, seed=seed + 1, threads=threads, realistic=realistic)
        bed = os.path.join(td, "ref.bed")
        rc, dt = run_reference(["junctions", "extract", "-s", "XS", "-o", bed, p])
        if dt is not None:
            jx = regtools_amd.JunctionsExtractor(bam=p, strandness=0, ctx=ctx)
            jx.identify_junctions_from_BAM()
            out["reference"] = {"sample": "%d reads of the same shape (seed %d), reference regtools, 1 thread" % (ss["n_reads"], seed + 1), "seconds": round(dt, 3),
                                "alignments_per_s": ss["n_reads"] / dt if rc == 0 else None,
                                "bed12_identical_to_gpu": bool(rc == 0 and open(bed, "rb").read() == jx.bed12())}
            if rc == 0 and shape == "short":     # what all the host's cores do with the reference on this payload (one process per contig)
                out["reference"]["all_host_cores"] = reference_on_all_cores(os.path.join(ROOT, "oracle", "_ref", "regtools_ref"), p, ss["n_reads"])
    return out


def extra_identify(regtools_amd, synth, ctx, reads, genes, variants, sample, seed, threads):
    """configs[3]: `cis-splice-effects identify` (50 M-read BAM + 500 k-variant VCF + GENCODE-scale GTF) through rgx_identify; the interval
    kernels' rooflines on SURVEY 8d's algorithmic bytes (B(variant) = 12 + 8 E_v, B(junction) = 20 + 8 E_j, B(window) = 8 + 16 K_w, summed by
    the kernels themselves); the reference on a bounded sample quartet, its three output files compared byte for byte."""
    from regtools_amd.cse import CisSpliceEffectsIdentifier

    def quartet(td, tag, n_reads, n_genes, n_var):
        pre = os.path.join(td, tag)
        st = synth.write(pre + ".bam", n_reads, shape="short", seed=seed, n_genes=n_genes, threads=threads)
        ann = synth.annotation(pre, n_genes, n_var, seed=seed, fasta=True, threads=threads)
        return pre, st, ann

    def run(pre, ann, tag):
        ci = CisSpliceEffectsIdentifier(ctx=ctx)
        ci.parse_options(["-s", "XS", "-o", pre + tag + ".tsv", "-v", pre + tag + ".vcf", "-j", pre + tag + ".bed", ann["vcf"], pre + ".bam", ann["fasta"], ann["gtf"]])
        t = time.time()
        ci.identify()
        return time.time() - t, dict(ci.stats)

    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        t_gen = time.time()
        pre, st, ann = quartet(td, "c4", reads, genes, variants)
        t_gen = time.time() - t_gen
        runs = [run(pre, ann, ".gpu") for _ in range(3)]
        wall, S = min(runs, key=lambda r: r[0])
        b_var = 12 * S["n_variants"] + 8 * S["exon_visits_variants"]
        b_jun = 20 * S["n_junctions"] + 8 * S["exon_visits_junctions"]
        b_win = 8 * S["n_windows"] + 16 * S["n_pairs"]

        def roof(b, ms):
            return {"bound": "hbm", "algorithmic_bytes": b, "kernel_ms": ms, "achieved": (b / (ms * 1e-3) / 1e9) if ms > 0 else None, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": (b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else None}
        out = {"workload": "configs[3]: cis-splice-effects identify -s XS, %d-read BAM + %d-variant VCF + %d-transcript GTF + FASTA" % (st["n_reads"], variants, genes * 4),
               "seconds": wall, "all_seconds": [round(r[0], 4) for r in runs], "first_call_seconds": round(runs[0][0], 4), "input_generation_s": round(t_gen, 2),
               "stage_ms": {k[3:]: round(S[k], 3) for k in ("ms_gtf", "ms_variants", "ms_extract", "ms_join", "ms_annotate", "ms_output", "ms_total")},
               "stage_note": "the GTF and the VCF are parsed on host threads while the device extracts: gtf / variants = what was left of them once the "
                             "extraction was done; seconds = the best of three calls on one context (it keeps the genome mapped: the first call maps it)",
               "counts": {k: S[k] for k in ("n_variants", "n_relevant", "n_windows", "n_pairs", "n_junctions", "n_records", "n_events", "exon_visits_variants", "exon_visits_junctions")},
               "roofline": {"k_variant_scan": roof(b_var, S["ms_k_variant_scan"]), "k_junction_scan": roof(b_jun, S["ms_k_junction_scan"]),
                            "k_window_pairs": roof(b_win, S["ms_k_window_pairs"]),
                            "note": "random gathers over the flat GTF arrays: one lane per variant (k_variant_scan), one wave per junction (k_junction_scan_wave), a wave per small window and sixteen waves per slice of a hot one (k_window_ranges + k_window_pairs_small + k_window_pairs, count and fill: 19 k windows, most with a few dozen candidate events, a few with 600 k -- two binary searches of ~23 dependent loads per window and the hottest window's slices are the launches' time, not the bytes): far below the HBM line by construction (SURVEY 8d)"}}
        spre, sst, sann = quartet(td, "s4", *sample)
        run(spre, sann, ".gpu")
        rc, dt = run_reference(["cis-splice-effects", "identify", "-s", "XS", "-o", spre + ".ref.tsv", "-v", spre + ".ref.vcf", "-j", spre + ".ref.bed",
                                sann["vcf"], spre + ".bam", sann["fasta"], sann["gtf"]], timeout=300)
        if dt is not None:
            same = rc == 0 and all(open(spre + ".gpu." + e, "rb").read() == open(spre + ".ref." + e, "rb").read() for e in ("tsv", "vcf", "bed"))
            out["reference"] = {"sample": "%d reads, %d genes, %d variants (same generator, seed %d), reference regtools, 1 thread" % (sample[0], sample[1], sample[2], seed),
                                "seconds": round(dt, 3), "finished": rc is not None, "outputs_identical_to_gpu": bool(same)}
    return out


def bind_to_gpu_numa(device_index):
    """N > 1 (one process per GPU): this rank's threads -- and with them the first touch of its page-locked file buffer -- go to the CPUs of the NUMA node its
    GPU hangs on (sysfs: the PCI device's numa_node, the node's cpulist), so that eight uploads of 53 GB/s do not cross the sockets.  Returns what it did (for
    the JSON line); anything missing or odd (no sysfs, node -1, a cpuset that shares no CPU with the node) leaves the process as it is."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return "none (the device reports no NUMA node)"
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        mine = os.sched_getaffinity(0) & cpus
        if not mine:
            return "none (node %d shares no CPU with this process's set)" % node
        os.sched_setaffinity(0, mine)
        return "node %d of %s, %d CPUs" % (node, bdf, len(mine))
    except Exception as e:           # (never in the way of the run)
        return "none (%s)" % type(e).__name__


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)     # (three: the context settles -- arena placement, page-locked table block, recycled event block -- over its first calls: 23.6-23.8 ms per step behind three, 23.8-24.3 behind one)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU (config 2: 50M)")
    ap.add_argument("--shape", default="short", choices=["short", "long"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other workloads reported beside the headline (realistic payload, config 5 long reads, config 4 identify)")
    ap.add_argument("--host-only", action="store_true", help="(profiling) only the timed host-bytes pass; the roofline line then quotes the overlapped launches")
    ap.add_argument("--realistic", action="store_true", help="random bases + binned qualities instead of the named constant payload")
    ap.add_argument("--multi-host", default="ranks", choices=["ranks", "cpp"],
                    help="N > 1: 'ranks' = one process per GPU, torch.distributed over RCCL (what torchrun launches; the default); 'cpp' = THIS process drives "
                         "all N devices through rgx_extract_multi (the C++ host of the CLI: a thread per device, one RCCL gather); without torchrun only")
    ap.add_argument("--dump-bed", default=None, help="(tests) rank 0 writes the last step's BED12 here")
    ap.add_argument("--no-sustained", action="store_true", help="skip the pass with two files in flight (value_sustained)")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes (the line then quotes the committed measurement)")
    args = ap.parse_args()
    if args.multi_host == "cpp" and args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return main_cpp_host(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the hot path has no CPU fallback")
    # (BENCH_DEVICE / BENCH_BACKEND: test knobs -- several ranks on ONE GPU with the gloo backend exercise this script's N > 1 path where no
    # multi-GPU node is at hand; the driver's runs use neither)
    device_index = int(os.environ.get("BENCH_DEVICE", local_rank))
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device_index)
    host_binding = "none (one process)"
    if (world > 1 and os.environ.get("BENCH_DEVICE") is None) or os.environ.get("BENCH_NUMA_BIND"):
        host_binding = bind_to_gpu_numa(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    import regtools_amd
    from regtools_amd import _ffi, synth
    from regtools_amd import distributed as rdist

    # ---- synthetic input: this rank's coordinate slice of the job ---------------------------------------------
    t_gen = time.time()
    threads = max(1, (os.cpu_count() or 8) // max(1, world))
    bam, bai, st = synth.generate(args.reads, shape=args.shape, seed=args.seed, threads=threads, realistic=args.realistic,
                                  slice_index=rank, n_slices=world)
    t_gen = time.time() - t_gen
    n_reads = st["n_reads"]
    ctx = regtools_amd.Context(device_index)
    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
    # The timed region: file bytes in page-locked HOST memory -> sorted junction table in host memory.  The upload is part of
    # every step (chunked over a copy stream, the members of the chunks that have arrived inflate meanwhile: rgx_extract_mem).
    pin = regtools_amd.PinnedBuffer(bam)

    merge_ms = []
    launch_ms_in_step = []                 # the arrival-gated DEFLATE launch of every timed step (its own HIP events on the stream it runs on)

    def step():
        je.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
        launch_ms_in_step.append(je.stats.get("ms_inflate_launch", 0.0))
        if world > 1:
            tm = time.time()
            m = rdist.gather_and_merge(je, min_anchor=8)
            merge_ms.append(1e3 * (time.time() - tm))
            return m
        return je.table

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    del merge_ms[:]
    del launch_ms_in_step[:]
    t0 = time.time()
    last = None
    for _ in range(args.steps):
        last = step()
    fence()
    dt = time.time() - t0
    n_events = je.stats["n_events"]
    bed_host_path = je.bed12() if world == 1 else None
    if args.dump_bed and rank == 0:
        with open(args.dump_bed, "wb") as f:
            f.write(je.bed12() if world == 1 else last.bed12())
    # N > 1: what every rank spent where, for a scaling line that checks itself (one small all-gather, outside the timed region)
    rank_report = None
    if world > 1:
        # every rank's host link with ALL ranks copying at once (256 MiB out of page-locked memory, three times, the best): whether eight uploads share
        # one socket's memory or cross the sockets shows here, next to the rank's binding
        probe_src = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True)
        probe_dst = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        h2d = 0.0
        for _ in range(3):
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); probe_dst.copy_(probe_src, non_blocking=True); e1.record(); e1.synchronize()
            h2d = max(h2d, (256 << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del probe_src, probe_dst
        bindings = [None] * world
        dist.all_gather_object(bindings, host_binding)
        mine = torch.tensor([je.stats["ms_inflate"], je.stats["ms_records"], je.stats["ms_scan"], je.stats["ms_reduce"], je.stats["ms_total"],
                             sum(merge_ms) / max(1, len(merge_ms)), float(n_reads), float(je.stats["n_junctions"]), float(je.stats["n_records"]), h2d], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_report = [dict(rank=r, inflate_ms=round(v[0].item(), 3), records_ms=round(v[1].item(), 3), scan_ms=round(v[2].item(), 3), reduce_ms=round(v[3].item(), 3),
                            extract_total_ms=round(v[4].item(), 3), gather_and_merge_ms=round(v[5].item(), 3), reads=int(v[6].item()), rows=int(v[7].item()),
                            n_records=int(v[8].item()), upload_GBps_all_ranks_at_once=round(v[9].item(), 1), host_binding=bindings[r]) for r, v in enumerate(allr)]
        # The N > 1 line checks itself (outside the timed region; the driver runs this path on hardware nobody else has seen it on):
        #  (i) every rank decoded its whole slice: sum of n_records == N x reads;
        #  (ii) the table the collective + device merge produced == an INDEPENDENT merge of the same per-rank tables -- every rank's packed rows
        #       go to rank 0 as plain objects (no RCCL), rgx_table_merge (the host merge, its own code path) merges them, and the two tables
        #       must have the same rows and print the same BED12 bytes; (iii) supporting reads are conserved: sum of the merged counts == sum
        #       of the ranks' counts.
        assert sum(r["n_records"] for r in rank_report) == world * n_reads, ("records lost", [r["n_records"] for r in rank_report], world, n_reads)
        payload, nrows = rdist.pack_table(je.table)
        parts = [None] * world if rank == 0 else None
        dist.gather_object((payload, nrows, int(je.table.contents.stream_ended)), parts, dst=0)
        if rank == 0:
            ref = rdist.merge_packed([(b, n) for b, n, _ in parts], je.table, 8, ended=[bool(e) for _, _, e in parts])
            import numpy as np
            cnt_parts = sum(int(np.frombuffer(b, dtype=np.uint32).reshape(-1, 12)[:, 5].sum()) for b, n, _ in parts if n)
            mt = last.table.contents
            cnt_merged = int(np.ctypeslib.as_array(mt.read_count, shape=(int(mt.n),)).astype(np.int64).sum()) if mt.n else 0
            multi_checks = dict(records_conserved=True, merged_rows=int(last.n), independent_host_merge_rows=int(ref.n),
                                bed12_equals_independent_merge=bool(last.bed12() == ref.bed12()), counts_conserved=bool(cnt_parts == cnt_merged),
                                supporting_reads=cnt_merged)
            assert multi_checks["merged_rows"] == multi_checks["independent_host_merge_rows"] and multi_checks["bed12_equals_independent_merge"], multi_checks
            assert multi_checks["counts_conserved"], (cnt_parts, cnt_merged)

    # Second, untimed-for-the-headline pass with the file ALREADY RESIDENT in HBM: one k_inflate launch over the whole file, which is what
    # the roofline of the dominant kernel is quoted on (HIP events on the pipeline's stream), and the per-stage times.
    d_bam = torch.zeros(len(bam) + 64, dtype=torch.uint8, device="cuda")
    d_bam[: len(bam)].copy_(torch.frombuffer(bytearray(bam), dtype=torch.uint8))
    torch.cuda.synchronize()

    def step_resident():
        je.identify_junctions_from_BAM(bai_bytes=bai, device_ptr=d_bam.data_ptr(), device_len=len(bam))
        if world > 1:
            return rdist.gather_and_merge(je, min_anchor=8)
        return je.table

    if args.host_only:
        step_resident = step
    step_resident()
    inflate_ms, stage_ms = [], dict(inflate=0.0, records=0.0, scan=0.0, reduce=0.0, total=0.0)
    fence()
    t1 = time.time()
    for _ in range(1 if args.host_only else args.steps):
        step_resident()
        s = je.stats
        inflate_ms.append(s["ms_inflate"])
        for k in stage_ms:
            stage_ms[k] += s["ms_" + k] / (1 if args.host_only else args.steps)
    fence()
    dt_res = (time.time() - t1) * (args.steps if args.host_only else 1)
    # Sustained: files back to back with TWO in flight (rgx_pipeline, csrc/pipeline.cpp: file k+1's upload and arrival-gated inflate run under file k's
    # tail; N > 1: under file k's collective and merge as well).  The per-file headline above stays what SURVEY 8d defines; this is what a cohort run sees.
    sustained = None
    # (two more contexts with an arena each: only where the device has the room -- eight test ranks on ONE GPU do not)
    free_b, _ = torch.cuda.mem_get_info()
    need_b = 2 * (int(je.stats["inflated_bytes"] * 1.2) + 2 * len(bam) + (3 << 30))
    if os.environ.get("BENCH_DEVICE") is not None:
        need_b *= world                        # (the test layout: all ranks on one GPU)
    room = free_b > need_b
    if world > 1:
        rt = torch.tensor([1.0 if room else 0.0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(rt, op=dist.ReduceOp.MIN)
        room = rt.item() > 0
    if not args.host_only and not args.no_sustained and not room:
        sustained = {"skipped": "not enough free device memory for two more contexts (%.1f GB free, %.1f GB wanted)" % (free_b / 1e9, need_b / 1e9)}
    if not args.host_only and not args.no_sustained and room:
        n_files = max(8, args.steps)
        pl = regtools_amd.Pipeline(device_index, 2)

        def run_files(nf):
            tickets, out = [pl.submit(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam), strandness=0)], None
            for k in range(nf):
                if k + 1 < nf:
                    tickets.append(pl.submit(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam), strandness=0))
                jk = pl.wait(tickets[k])
                out = rdist.gather_and_merge(jk, min_anchor=8) if world > 1 else jk
            return out

        run_files(2)                                   # (both contexts' first call: their workspaces)
        fence()
        t2 = time.time()
        last_p = run_files(n_files)
        fence()
        dt_sus = time.time() - t2
        same = bool(last_p.bed12() == (je.bed12() if world == 1 else last.bed12()))
        del last_p
        pl.close()
        if world > 1:
            t3 = torch.tensor([dt_sus, 0.0 if same else 1.0], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
            dt_sus, same = float(t3[0].item()), t3[1].item() == 0.0
        hwq = os.environ.get("GPU_MAX_HW_QUEUES", "")
        sustained = {"files": n_files, "in_flight": 2, "ms_per_file": 1e3 * dt_sus / n_files, "table_identical_to_the_timed_step": same,
                     "GPU_MAX_HW_QUEUES": hwq,
                     "is": "files back to back through rgx_extract_submit / rgx_extract_wait, two contexts on the device: the uploads take the link in turns, "
                           + ("the DEFLATE launches go out at once (a hardware queue per stream) and the files' kernels interleave" if hwq.isdigit() and int(hwq) >= 16
                              else "the DEFLATE launches take the chip in turns, file k+1's upload and gated inflate run under file k's tail")
                           + (", all of it under file k's all-gather + merge" if world > 1 else "")}
        assert same, "sustained pass: the table differs from the timed step's"
    if world > 1:
        tt = torch.tensor([dt, dt_res], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_res = float(tt[0].item()), float(tt[1].item())
        cnt = torch.tensor([float(n_reads), float(n_events)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_reads, total_events = cnt[0].item(), cnt[1].item()
    else:
        total_reads, total_events = float(n_reads), float(n_events)

    if rank == 0:
        s = je.stats
        ms_step = 1e3 * dt / args.steps
        aln_per_s = total_reads * args.steps / dt
        # dominant kernel: k_inflate. Algorithmic bytes per launch: every compressed byte read once
        # + every inflated byte written once = (C + U) * alignments in the launch.
        alg_bytes = s["compressed_bytes"] + s["inflated_bytes"]
        k_ms = sum(inflate_ms) / len(inflate_ms)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # HBM-side traffic of the same kernel from rocprofv3 PMC passes (tools/pmc_traffic.sh; separate --pmc runs of this
        # very command).  Only quoted when the committed measurement was taken on this exact workload.
        tr = {"traffic": None}
        if world == 1:
            kname = inflate_kernel_for(s["compressed_bytes"], s["inflated_bytes"])
            tr = (None if args.no_live_traffic else live_traffic(bam, n_reads, kname, alg_bytes, (["--realistic"] if args.realistic else []) + (["--shape", "long"] if args.shape == "long" else []))) \
                or committed_traffic("long10M" if args.shape == "long" else "realistic" if args.realistic else "default", kname, alg_bytes)
        traffic, traffic_note, traffic_source = tr.get("traffic"), tr.get("traffic_note"), tr.get("traffic_source")
        k_in_step = [x for x in launch_ms_in_step if x > 0]
        line = {
            "metric": "alignments/sec + junctions/sec, junctions extract, 1/2/4/8 MI355X",
            "value": aln_per_s, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 integer", "data": "synthetic",
            "config": {"workload": ("configs[1]: synthetic %d-read 101 bp BAM per GPU, ~15%% reads with one N-op, junctions extract -s XS%s" % (n_reads, " (realistic payload)" if args.realistic else "")) if args.shape == "short" else
                                   "configs[4]-shape: synthetic %d long reads per GPU (l_qseq 1000-10000, n_cigar <= 64, 5-20 N ops), junctions extract -s XS" % n_reads,
                       "reads_per_gpu": n_reads, "shape": args.shape, "seed": args.seed, "sharding": "coordinate slice per GPU, all-gather of packed junction rows" if world > 1 else "single GPU", "host_binding_rank0": host_binding,
                       "bgzf_members": s["n_members"], "compressed_bytes_per_gpu": s["compressed_bytes"], "inflated_bytes_per_gpu": s["inflated_bytes"],
                       "bytes_per_alignment": {"compressed": s["compressed_bytes"] / n_reads, "inflated": s["inflated_bytes"] / n_reads}},
            "junction_events_per_s": total_events * args.steps / dt,
            "timed_region": "file bytes in page-locked host memory -> sorted junction table in host memory (SURVEY.md 8d): chunked H2D upload inside every step, overlapped with the inflate",
            "value_device_resident": total_reads * args.steps / dt_res, "ms_per_step_device_resident": 1e3 * dt_res / args.steps,
            "value_sustained": (total_reads * 1e3 / sustained["ms_per_file"]) if sustained and "ms_per_file" in sustained else None, "sustained": sustained,
            "junction_rows": s["n_junctions"] if world == 1 else int(last.n),
            "multi_gpu": None if world == 1 else {"host": "one process per GPU (torchrun), torch.distributed backend %s" % backend, "rccl_ranks": world if backend == "nccl" else 0,
                                                  "exchange": "all_gather_into_tensor of the ranks' packed 48-byte rows into HBM + rgx_table_merge_device on every rank" if backend == "nccl"
                                                              else "all_gather of packed rows on the host + rgx_table_merge (test backend)",
                                                  "merge_ms": round(max(r["gather_and_merge_ms"] for r in rank_report), 3),
                                                  "merge_is": "serial behind the extraction in the timed step; in the sustained pass it runs under the next file's upload",
                                                  "upload_GBps_all_ranks_at_once": [r["upload_GBps_all_ranks_at_once"] for r in rank_report],
                                                  "bound_hint": "host link" if min(r["upload_GBps_all_ranks_at_once"] for r in rank_report) < 35 else "device",
                                                  "checks": multi_checks, "per_rank": rank_report},
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "input_generation_s": round(t_gen, 2),
            "roofline": {"bound": "hbm", "kernel": inflate_kernel_for(s["compressed_bytes"], s["inflated_bytes"]), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_uncorrected": tr.get("traffic_uncorrected"), "traffic_source": traffic_source, "traffic_note": traffic_note,
                         "kernel_ms": k_ms, "kernel_ms_is": "the whole-file launch of the device-resident pass (HIP events on the pipeline's stream): what achieved / frac are quoted on",
                         "kernel_ms_in_step": (sum(k_in_step) / len(k_in_step)) if k_in_step else None,
                         "kernel_ms_in_step_is": "the arrival-gated launch the TIMED step runs (same kernel under its PIECE symbol, HIP events on its own stream): it spans the upload -- "
                                                 "its waves wait for the chunk their members lie in -- so its duration is not a rate of the kernel",
                         "arena_placement_trials_ms": ctx.arena_trials(),
                         "arena_placement_is": "opt-in since round 6 (REGTOOLS_AMD_ARENA=5; [] = the default, no trials): the launch's time depends on where the arena's pages lie (stable per "
                                               "allocation; DESIGN.md 5.5) and a context can time the same launch into fresh allocations, one at a time, and keep a faster one",
                         "algorithmic_bytes": alg_bytes,
                         "note": "DEFLATE is a serial bit stream per member: one lane per member (long matches copied by the wave), bound by per-lane dependent ALU/LDS chains, the L1's rate of scattered per-lane accesses and one memory round trip per symbol trip, far below the HBM line (SURVEY 8d; DESIGN.md 5)",
                         "pipeline_frac": aln_per_s / world * (alg_bytes / n_reads) / (HBM_PEAK_GBS * 1e9)},
        }
        if world == 1 and not args.no_cpu_baseline:
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "bench.bam")
                with open(path, "wb") as f:
                    f.write(bam)
                with open(path + ".bai", "wb") as f:
                    f.write(bai)
                cb, bed = cpu_baseline(path, n_reads, s["n_events"])
                # while we are here: the GPU table must equal the CPU one on the full-size workload
                with open(bed, "rb") as f:
                    ref_bed = f.read()
                    cb["bed12_identical_to_gpu"] = ref_bed == je.bed12() and ref_bed == bed_host_path
                line["cpu_baseline"] = cb
                # the TOOL, not the loop: a cold `regtools-amd junctions extract -s XS -o out.bed bench.bam` process on the same file (page cache),
                # against the cold reference process timed above -- context creation, HBM allocation, file mapping, upload from pageable memory,
                # BED12 text and the write included
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import cli_wall
                    cw = cli_wall.measure(path, runs=5, with_reference=False)
                    cli_bed = open(path + ".cli.bed", "rb").read()
                    line["cli"] = {"command": "bin/regtools-amd junctions extract -s XS -o out.bed bench.bam (cold process, file in the page cache)",
                                   "wall_s": cw["wall_s"], "wall_s_is": "median of 5 cold processes, 0.5 s apart", "best_wall_s": cw["best_wall_s"], "runs": cw["runs"], "reference_wall_s": cb.get("seconds") if cb.get("kind") == "reference" else None,
                                   "identical_file": cli_bed == ref_bed,
                                   "ratio_process": round(cb["seconds"] / cw["wall_s"], 1) if cb.get("kind") == "reference" else None,
                                   "ratio_pipeline": round(cb["seconds"] / (ms_step * 1e-3), 1) if cb.get("kind") == "reference" else None,
                                   "note": "ratio_process = reference process wall / this process wall; ratio_pipeline = reference process wall / one warm in-process step (the headline's region)"}
                except Exception as e:
                    line["cli"] = {"error": repr(e)}
        if world == 1 and not args.no_extras and not args.host_only and args.shape == "short" and not args.realistic:
            try:
                del d_bam
                pin.close()
                which = os.environ.get("BENCH_EXTRAS", "realistic,long,identify").split(",")
                if "realistic" in which:
                  line["realistic"] = extra_extract(regtools_amd, synth, ctx, "configs[1] shape with random bases and binned qualities (payload that does not compress to nothing)",
                                                  "short", True, args.reads, max(1, args.reads // 10), args.seed, threads)
                if "long" in which:
                  line["long10M"] = extra_extract(regtools_amd, synth, ctx, "configs[4]: long reads, l_qseq 1000-10000, n_cigar <= 64, 5-20 N ops", "long", False,
                                                max(1, args.reads // 5), max(1, args.reads // 250), args.seed, threads, steps=2)
                if "identify" in which:
                  line["identify_config4"] = extra_identify(regtools_amd, synth, ctx, args.reads, max(4, args.reads // 800), max(10, args.reads // 100),
                                                          (max(1, args.reads // 25), max(4, args.reads // 20000), max(10, args.reads // 2500)), 4, threads)
            except Exception as e:           # the headline stands on its own
                line["extras_error"] = repr(e)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def main_cpp_host(args):
    """--multi-host cpp: this one process drives N devices through rgx_extract_multi_mem -- the C++ host the CLI uses with REGTOOLS_AMD_DEVICES
    (multi.cpp: the members scanned once, a thread + context per device uploading only its shard's byte range, one RCCL send/recv gather of the
    packed rows to the first device, merge there).  Strong scaling of ONE file of N x --reads reads.  A device list longer than the visible GPUs
    repeats device 0 (the shards then take turns: exercises the code, measures nothing)."""
    import torch
    import regtools_amd
    from regtools_amd import synth
    n = args.gpus
    visible = torch.cuda.device_count()
    devices = list(range(n)) if visible >= n else [0] * n
    bam, bai, st = synth.generate(args.reads * n, shape=args.shape, seed=args.seed, realistic=args.realistic)
    pin = regtools_amd.PinnedBuffer(bam)
    kw = dict(strandness=0)

    def step():
        return regtools_amd.extract_multi(devices, bam_bytes=None, bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam), **kw)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        m = step()
    dt = time.time() - t0
    if args.dump_bed:
        with open(args.dump_bed, "wb") as f:
            f.write(m.bed12())
    tc = m.table.contents
    # the line checks itself (outside the timed region): every record of the file was decoded by some shard, and the merged table is the table
    # ONE device makes of the same file (one more extraction, on the first device, of the whole file)
    from regtools_amd import _ffi
    exchange = _ffi.lib().rgx_multi_exchange_kind().decode()
    assert int(tc.n_records) == st["n_reads"], ("records lost", int(tc.n_records), st["n_reads"])
    ctx1 = regtools_amd.Context(devices[0])
    je1 = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx1)
    je1.identify_junctions_from_BAM(bai_bytes=bai, host_ptr=pin.ptr, host_len=len(bam))
    checks = dict(records_conserved=True, merged_rows=int(tc.n), single_device_rows=int(je1.stats["n_junctions"]), bed12_equals_single_device=bool(m.bed12() == je1.bed12()))
    assert checks["merged_rows"] == checks["single_device_rows"] and checks["bed12_equals_single_device"], checks
    del je1
    ctx1.close()
    line = {"metric": "alignments/sec + junctions/sec, junctions extract, 1/2/4/8 MI355X", "value": st["n_reads"] * args.steps / dt, "unit": "alignments/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8/u32 integer", "data": "synthetic",
            "config": {"workload": "one synthetic %d-read 101 bp BAM sharded by BGZF member range over %d devices (rgx_extract_multi)" % (st["n_reads"], n), "devices": devices,
                       "distinct_devices": len(set(devices)) == n},
            "junction_events_per_s": tc.n_events * args.steps / dt, "junction_rows": int(tc.n),
            "multi_gpu": {"host": "one process, a thread and a context per device (multi.cpp)", "rccl_ranks": n if len(set(devices)) == n else 0,
                          "exchange": exchange + " -> rgx_table_merge_device on the first device", "checks": checks,
                          "max_stage_ms": {"inflate": tc.ms_inflate, "records": tc.ms_records, "scan": tc.ms_scan, "reduce": tc.ms_reduce, "shard_total": tc.ms_total}}}
    print(json.dumps(line), flush=True)
    pin.close()


if __name__ == "__main__":
    main()
