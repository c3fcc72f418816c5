"""Synthetic BAM/BAI inputs of SURVEY.md section 8(d) (tooling; libregtools_synth.so)."""
import ctypes as C

from . import _ffi

SHAPES = {"short": 0, "long": 1, "fuzz": 2}


def _params(n_reads, shape, seed, level, threads, n_introns, realistic, slice_index=0, n_slices=1, n_genes=0):
    p = _ffi.SynthParams()
    p.shape, p.n_reads, p.seed, p.level, p.threads = SHAPES[shape], n_reads, seed, level, threads
    p.n_introns, p.spliced_frac, p.realistic_payload = n_introns, 0.0, 1 if realistic else 0
    p.slice_index, p.n_slices, p.n_genes = slice_index, n_slices, n_genes
    return p


def generate(n_reads, shape="short", seed=1, level=6, threads=0, n_introns=0, realistic=False, slice_index=0, n_slices=1, n_genes=0):
    """Returns (bam_bytes, bai_bytes, stats)."""
    L = _ffi.synth()
    p = _params(n_reads, shape, seed, level, threads, n_introns, realistic, slice_index, n_slices, n_genes)
    r = _ffi.SynthResult()
    rc = L.rgx_synth_generate(C.byref(p), C.byref(r))
    if rc:
        raise RuntimeError("synthetic BAM generation failed (%d)" % rc)
    try:
        # (c_ubyte * n).from_address handles buffers beyond 2 GiB, which C.string_at does not
        bam = bytes((C.c_ubyte * r.bam_len).from_address(r.bam))
        bai = bytes((C.c_ubyte * r.bai_len).from_address(r.bai))
        stats = dict(n_reads=r.n_reads, n_spliced=r.n_spliced, n_members=r.n_blocks, inflated_bytes=r.inflated_bytes,
                     cigar_ops=r.cigar_ops, bam_bytes=r.bam_len)
    finally:
        L.rgx_synth_free(C.byref(r))
    return bam, bai, stats


def write(path, n_reads, shape="short", seed=1, level=6, threads=0, n_introns=0, realistic=False, slice_index=0, n_slices=1, n_genes=0):
    L = _ffi.synth()
    p = _params(n_reads, shape, seed, level, threads, n_introns, realistic, slice_index, n_slices, n_genes)
    r = _ffi.SynthResult()
    rc = L.rgx_synth_write(C.byref(p), path.encode(), C.byref(r))
    if rc:
        raise RuntimeError("synthetic BAM generation failed (%d)" % rc)
    return dict(n_reads=r.n_reads, n_spliced=r.n_spliced, n_members=r.n_blocks, inflated_bytes=r.inflated_bytes,
                cigar_ops=r.cigar_ops, bam_bytes=r.bam_len)


def index(bam_path):
    rc = _ffi.synth().rgx_synth_index(bam_path.encode())
    if rc:
        raise RuntimeError("indexing %s failed (%d)" % (bam_path, rc))


def annotation(prefix, n_genes, n_variants, seed=1, fasta=True, threads=0):
    """Config-4 companions of a short-shape BAM generated with the same (seed, n_genes): PREFIX.gtf, PREFIX.vcf, PREFIX.fa(+.fai)."""
    p = _params(0, "short", seed, 6, threads, 0, False, n_genes=n_genes)
    rc = _ffi.synth().rgx_synth_annotation(C.byref(p), n_variants, (prefix + ".gtf").encode(), (prefix + ".vcf").encode(),
                                           (prefix + ".fa").encode() if fasta else None)
    if rc:
        raise RuntimeError("synthetic annotation generation failed (%d)" % rc)
    return dict(gtf=prefix + ".gtf", vcf=prefix + ".vcf", fasta=prefix + ".fa" if fasta else None)
