"""ctypes bindings of the C ABI declared in include/regtools_amd.h (libregtools_amd.so, HIP/gfx950)
and of the tooling library libregtools_synth.so (synthetic BAM writer).

The product library is loaded lazily and loading failures are NEVER swallowed: there is no Python or CPU
fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libregtools_amd.so")
SYNTH_PATH = os.path.join(_HERE, "libregtools_synth.so")


class ExtractParams(C.Structure):
    _fields_ = [("region", C.c_char_p), ("strandness", C.c_int32), ("strand_tag", C.c_char * 2),
                ("min_anchor", C.c_uint32), ("min_intron", C.c_uint32), ("max_intron", C.c_uint32),
                ("fasta_path", C.c_char_p), ("shard", C.c_int32), ("n_shards", C.c_int32),
                ("barcodes", C.c_int32), ("barcode_tag", C.c_char * 2)]


class JunctionTable(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("ref_name", C.POINTER(C.c_char_p)), ("ref_len", C.POINTER(C.c_uint32)),
                ("n", C.c_uint64), ("tid", C.POINTER(C.c_int32)), ("start", C.POINTER(C.c_uint32)),
                ("end", C.POINTER(C.c_uint32)), ("thick_start", C.POINTER(C.c_uint32)),
                ("thick_end", C.POINTER(C.c_uint32)), ("read_count", C.POINTER(C.c_uint32)),
                ("name_index", C.POINTER(C.c_uint64)), ("strand", C.POINTER(C.c_char)),
                ("left_ok", C.POINTER(C.c_uint8)), ("right_ok", C.POINTER(C.c_uint8)),
                ("n_records", C.c_uint64), ("n_events", C.c_uint64), ("inflated_bytes", C.c_uint64),
                ("compressed_bytes", C.c_uint64), ("n_members", C.c_uint64),
                ("ms_total", C.c_double), ("ms_inflate", C.c_double), ("ms_records", C.c_double),
                ("ms_scan", C.c_double), ("ms_reduce", C.c_double),
                ("first_seen", C.POINTER(C.c_uint64)), ("last_seen", C.POINTER(C.c_uint64)), ("framing_sweeps", C.c_uint64),
                ("bc_row_begin", C.POINTER(C.c_uint64)), ("bc_count", C.POINTER(C.c_uint32)), ("bc_str_begin", C.POINTER(C.c_uint64)),
                ("bc_text", C.POINTER(C.c_char)), ("bc_insert_rank", C.POINTER(C.c_uint32)), ("ms_barcodes", C.c_double), ("stream_ended", C.c_uint64),
                ("ms_inflate_launch", C.c_double)]


class Member(C.Structure):
    _fields_ = [("cpos", C.c_uint64), ("upos", C.c_uint64), ("clen", C.c_uint32), ("isize", C.c_uint32)]


# every symbol include/regtools_amd.h declares (tests check the library exports all of them)
EXPORTS = ["rgx_extract_params_default", "rgx_ctx_create", "rgx_ctx_destroy", "rgx_extract", "rgx_extract_mem",
           "rgx_extract_device", "rgx_table_free", "rgx_table_merge", "rgx_table_pack", "rgx_table_unpack",
           "rgx_table_format_bed12", "rgx_table_format_barcodes", "rgx_version", "rgx_k_inflate",
           "rgx_identify_params_default", "rgx_identify", "rgx_gtf_load", "rgx_gtf_free", "rgx_gtf_info", "rgx_gtf_transcript_bin",
           "rgx_gtf_transcript_id", "rgx_variant_windows", "rgx_variant_hits_free", "rgx_annotate_junctions", "rgx_junction_annot_free",
           "rgx_associate", "rgx_variants_annotate", "rgx_junctions_annotate", "rgx_junctions_annotate_opts", "rgx_table_merge_device", "rgx_window_join", "rgx_window_rows_free", "rgx_last_table_pack_device",
           "rgx_host_alloc", "rgx_host_free", "rgx_extract_multi", "rgx_extract_multi_mem", "rgx_multi_exchange_kind", "rgx_k_inflate_form", "rgx_table_merge_barcodes",
           "rgx_table_pack_barcodes", "rgx_table_unpack_barcodes", "rgx_identify_multi", "rgx_ctx_arena_trials",
           "rgx_pipeline_create", "rgx_pipeline_depth", "rgx_pipeline_ctx", "rgx_extract_submit", "rgx_extract_wait", "rgx_pipeline_destroy"]


class IdentifyParams(C.Structure):
    _fields_ = [("vcf_path", C.c_char_p), ("bam_path", C.c_char_p), ("fasta_path", C.c_char_p), ("gtf_path", C.c_char_p),
                ("out_tsv", C.c_char_p), ("out_vcf", C.c_char_p), ("out_bed", C.c_char_p), ("window", C.c_uint32),
                ("intronic_min", C.c_uint32), ("exonic_min", C.c_uint32), ("all_intronic", C.c_int32), ("all_exonic", C.c_int32),
                ("skip_single", C.c_int32), ("strandness", C.c_int32), ("strand_tag", C.c_char * 2), ("min_anchor", C.c_uint32),
                ("min_intron", C.c_uint32), ("max_intron", C.c_uint32), ("override_motif", C.c_int32), ("bed_path", C.c_char_p), ("echo", C.c_int32)]


class IdentifyStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_variants", "n_relevant", "n_windows", "n_pairs", "n_window_rows", "n_junctions", "n_records", "n_events",
                                          "exon_visits_variants", "exon_visits_junctions")] + \
               [(n, C.c_double) for n in ("ms_total", "ms_gtf", "ms_variants", "ms_extract", "ms_join", "ms_annotate", "ms_output",
                                          "ms_k_variant_scan", "ms_k_junction_scan", "ms_k_window_pairs")]


class VariantHits(C.Structure):
    _fields_ = [("n", C.c_uint64), ("cis_start", C.POINTER(C.c_uint32)), ("cis_end", C.POINTER(C.c_uint32)), ("hit_off", C.POINTER(C.c_uint32)),
                ("hit_transcript", C.POINTER(C.c_uint32)), ("hit_annotation", C.POINTER(C.c_uint32)), ("hit_distance", C.POINTER(C.c_uint32))]


class WindowRows(C.Structure):
    _fields_ = [("n", C.c_uint64)] + [(k, C.POINTER(C.c_uint32)) for k in ("window", "start", "end", "thick_start", "thick_end", "read_count", "name_index")] + \
               [("strand", C.POINTER(C.c_char))]


class JunctionAnnot(C.Structure):
    _fields_ = [("n", C.c_uint64), ("flags", C.POINTER(C.c_uint32)), ("n_acceptors_skipped", C.POINTER(C.c_uint32)),
                ("n_exons_skipped", C.POINTER(C.c_uint32)), ("n_donors_skipped", C.POINTER(C.c_uint32)), ("tx_off", C.POINTER(C.c_uint32)),
                ("tx", C.POINTER(C.c_uint32))]

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so and ask for it by file name, while this library asks for
    the SONAME libamdhip64.so.7: loaded in the wrong order the process ends up with TWO HIP runtimes and the second
    one sees no GPU.  Loading torch's copy first (when torch is installed) makes both resolve to the same object."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libregtools_amd.so (raises if it was not built: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("regtools_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        P = C.POINTER
        L.rgx_version.restype = C.c_char_p
        L.rgx_extract_params_default.argtypes = [P(ExtractParams)]
        L.rgx_ctx_create.argtypes = [C.c_int, P(C.c_void_p), C.c_char_p, C.c_size_t]
        L.rgx_ctx_destroy.argtypes = [C.c_void_p]
        L.rgx_ctx_arena_trials.argtypes = [C.c_void_p, P(C.c_float), C.c_int]
        L.rgx_extract.argtypes = [C.c_void_p, C.c_char_p, P(ExtractParams), P(P(JunctionTable)), C.c_char_p, C.c_size_t]
        L.rgx_extract_mem.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, P(ExtractParams),
                                      P(P(JunctionTable)), C.c_char_p, C.c_size_t]
        L.rgx_extract_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         P(ExtractParams), P(P(JunctionTable)), C.c_char_p, C.c_size_t]
        L.rgx_table_free.argtypes = [P(JunctionTable)]
        L.rgx_pipeline_create.argtypes = [C.c_int, C.c_int, P(C.c_void_p), C.c_char_p, C.c_size_t]
        L.rgx_pipeline_depth.argtypes = [C.c_void_p]
        L.rgx_pipeline_ctx.argtypes = [C.c_void_p, C.c_uint64]
        L.rgx_pipeline_ctx.restype = C.c_void_p
        L.rgx_extract_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, P(ExtractParams), P(C.c_uint64), C.c_char_p, C.c_size_t]
        L.rgx_extract_wait.argtypes = [C.c_void_p, C.c_uint64, P(P(JunctionTable)), C.c_char_p, C.c_size_t]
        L.rgx_pipeline_destroy.argtypes = [C.c_void_p]
        L.rgx_extract_multi.argtypes = [P(C.c_int), C.c_int, C.c_char_p, P(ExtractParams), P(P(JunctionTable)), C.c_char_p, C.c_size_t]
        L.rgx_multi_exchange_kind.restype = C.c_char_p
        L.rgx_multi_exchange_kind.argtypes = []
        L.rgx_extract_multi_mem.argtypes = [P(C.c_int), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, P(ExtractParams), P(P(JunctionTable)), C.c_char_p, C.c_size_t]
        L.rgx_host_alloc.argtypes = [C.c_size_t]
        L.rgx_host_alloc.restype = C.c_void_p
        L.rgx_host_free.argtypes = [C.c_void_p]
        L.rgx_table_merge.argtypes = [P(P(JunctionTable)), C.c_int, C.c_uint32, P(P(JunctionTable)), C.c_char_p, C.c_size_t]
        L.rgx_table_merge_barcodes.argtypes = [P(P(JunctionTable)), C.c_int, P(JunctionTable), C.c_char_p, C.c_size_t]
        L.rgx_table_pack_barcodes.argtypes = [P(JunctionTable), C.c_void_p, C.c_size_t]
        L.rgx_table_pack_barcodes.restype = C.c_size_t
        L.rgx_table_unpack_barcodes.argtypes = [P(JunctionTable), C.c_void_p, C.c_size_t]
        L.rgx_table_pack.argtypes = [P(JunctionTable), C.c_void_p, C.c_size_t]
        L.rgx_table_pack.restype = C.c_size_t
        L.rgx_table_unpack.argtypes = [C.c_void_p, C.c_size_t, P(JunctionTable), P(P(JunctionTable))]
        L.rgx_last_table_pack_device.argtypes = [C.c_void_p, P(JunctionTable), C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.rgx_table_merge_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64), C.c_int, C.c_uint32, P(JunctionTable), P(P(JunctionTable)), C.c_char_p, C.c_size_t]
        L.rgx_table_format_bed12.argtypes = [P(JunctionTable), C.c_int, C.c_char_p, C.c_size_t]
        L.rgx_table_format_bed12.restype = C.c_size_t
        L.rgx_table_format_barcodes.argtypes = [P(JunctionTable), C.c_int, C.c_char_p, C.c_size_t]
        L.rgx_table_format_barcodes.restype = C.c_size_t
        L.rgx_k_inflate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rgx_k_inflate_form.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rgx_identify_params_default.argtypes = [P(IdentifyParams)]
        L.rgx_identify.argtypes = [C.c_void_p, P(IdentifyParams), P(IdentifyStats), C.c_char_p, C.c_size_t]
        L.rgx_associate.argtypes = L.rgx_identify.argtypes
        L.rgx_identify_multi.argtypes = [P(C.c_int), C.c_int, P(IdentifyParams), P(IdentifyStats), C.c_char_p, C.c_size_t]
        L.rgx_variants_annotate.argtypes = L.rgx_identify.argtypes
        L.rgx_junctions_annotate_opts.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, P(C.c_uint64), C.c_char_p, C.c_size_t]
        L.rgx_junctions_annotate.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, P(C.c_uint64), C.c_char_p, C.c_size_t]
        L.rgx_gtf_load.argtypes = [C.c_void_p, C.c_char_p, P(C.c_void_p), C.c_char_p, C.c_size_t]
        L.rgx_gtf_free.argtypes = [C.c_void_p]
        L.rgx_gtf_info.argtypes = [C.c_void_p, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
        L.rgx_gtf_transcript_bin.argtypes = [C.c_void_p, C.c_char_p, P(C.c_uint32)]
        L.rgx_gtf_transcript_id.argtypes = [C.c_void_p, C.c_uint32]
        L.rgx_gtf_transcript_id.restype = C.c_char_p
        L.rgx_variant_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_char_p), P(C.c_uint32), C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                          P(P(VariantHits)), C.c_char_p, C.c_size_t]
        L.rgx_variant_hits_free.argtypes = [P(VariantHits)]
        L.rgx_annotate_junctions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_char_p), P(C.c_uint32), P(C.c_uint32), C.c_char_p,
                                             P(P(JunctionAnnot)), C.c_char_p, C.c_size_t]
        L.rgx_junction_annot_free.argtypes = [P(JunctionAnnot)]
        L.rgx_window_join.argtypes = [C.c_void_p, C.c_char_p, P(ExtractParams), C.c_uint64, P(C.c_char_p), P(C.c_int32), P(C.c_int32), P(P(WindowRows)), C.c_char_p, C.c_size_t]
        L.rgx_window_rows_free.argtypes = [P(WindowRows)]
        _lib = L
    return _lib


class SynthParams(C.Structure):
    _fields_ = [("shape", C.c_int), ("n_reads", C.c_uint64), ("seed", C.c_uint64), ("level", C.c_int),
                ("threads", C.c_int), ("n_introns", C.c_uint32), ("spliced_frac", C.c_double),
                ("realistic_payload", C.c_int), ("slice_index", C.c_int), ("n_slices", C.c_int), ("n_genes", C.c_uint32)]


class SynthResult(C.Structure):
    _fields_ = [("bam", C.c_void_p), ("bam_len", C.c_size_t), ("bai", C.c_void_p), ("bai_len", C.c_size_t),
                ("n_reads", C.c_uint64), ("n_spliced", C.c_uint64), ("n_blocks", C.c_uint64),
                ("inflated_bytes", C.c_uint64), ("cigar_ops", C.c_uint64)]


_synth = None


def synth():
    global _synth
    if _synth is None:
        if not os.path.exists(SYNTH_PATH):
            raise RuntimeError("regtools_amd: %s is missing -- run __graft_entry__.build()" % SYNTH_PATH)
        L = C.CDLL(SYNTH_PATH)
        P = C.POINTER
        L.rgx_synth_generate.argtypes = [P(SynthParams), P(SynthResult)]
        L.rgx_synth_free.argtypes = [P(SynthResult)]
        L.rgx_synth_write.argtypes = [P(SynthParams), C.c_char_p, P(SynthResult)]
        L.rgx_synth_index.argtypes = [C.c_char_p]
        L.rgx_synth_annotation.argtypes = [P(SynthParams), C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p]
        _synth = L
    return _synth


def format_bed12(table_ptr, only_anchored=True):
    """BED12 text of a table in ONE formatting pass (rgx_table_format_bed12 formats every row even for its size query): a row is its contig's
    name and at most 160 bytes of numbers (Junction::print, junctions_extractor.h:90-98), so n x (longest name + 160) bytes hold it; the exact
    two-call protocol only if that bound were ever short."""
    L = lib()
    t = table_ptr.contents
    longest = max([len(t.ref_name[i]) for i in range(t.n_ref)] or [0])
    cap = int(t.n) * (longest + 160) + 1
    buf = C.create_string_buffer(cap)
    n = L.rgx_table_format_bed12(table_ptr, 1 if only_anchored else 0, buf, cap)
    if n > cap:
        buf = C.create_string_buffer(n + 1)
        L.rgx_table_format_bed12(table_ptr, 1 if only_anchored else 0, buf, n)
    return buf.raw[:n]
