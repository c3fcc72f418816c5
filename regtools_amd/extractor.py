"""Host-side mirror of the reference's `JunctionsExtractor` (src/junctions/junctions_extractor.h:149-247)
on top of the C ABI.  Same method names, argument meaning and error behaviour, so the parity tests read like
the reference's own tests (tests/lib/junctions/test_junctions_extractor.cc,
tests/integration-test/test_junctions_extract.py).  All data-parallel work happens in libregtools_amd.so
on the GPU; nothing here touches alignment bytes.
"""
import ctypes as C

from . import _ffi

STRANDNESS = {"XS": 0, "RF": 1, "FR": 2, "intron-motif": 3}


class RegtoolsError(RuntimeError):
    """What the reference throws as std::runtime_error (junctions_main.cc:51-57 turns it into exit code 1)."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class Junction(object):
    """One row of the table -- struct Junction (junctions_extractor.h:39-112)."""
    __slots__ = ("chrom", "start", "end", "thick_start", "thick_end", "name", "read_count", "strand",
                 "has_left_min_anchor", "has_right_min_anchor")

    def bed12(self):
        # Junction::print (junctions_extractor.h:90-98)
        return "%s\t%d\t%d\t%s\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d" % (
            self.chrom, self.thick_start, self.thick_end, self.name, self.read_count, self.strand,
            self.thick_start, self.thick_end, (self.start - self.thick_start) & 0xffffffff,
            (self.thick_end - self.end) & 0xffffffff, (self.end - self.thick_start) & 0xffffffff)


def _atoi(s):
    """atoi as junctions_extractor.cc:62-70 uses it: leading white space, an optional sign, the digits that follow; anything else is 0."""
    import re
    m = re.match(r"\s*([+-]?\d+)", s)
    return int(m.group(1)) if m else 0


class Context(object):
    """One HIP device context (stream + reusable HBM workspace)."""

    def __init__(self, device=0):
        self._lib = _ffi.lib()
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self._lib.rgx_ctx_create(device, C.byref(self._h), err, len(err))
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())

    def arena_trials(self):
        """rgx_ctx_arena_trials: the DEFLATE launch's time into the call's own arena ([0]) and into the challengers the context's last placement
        calibration tried (ms); [] when none has run."""
        buf = (C.c_float * 8)()
        n = self._lib.rgx_ctx_arena_trials(self._h, buf, 8)
        return [round(float(buf[k]), 3) for k in range(min(n, 8))]

    def close(self):
        if self._h:
            self._lib.rgx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedBuffer(object):
    """File bytes in page-locked host memory (rgx_host_alloc): the input form under which rgx_extract_mem's chunked upload overlaps the
    inflate of the chunks that have already arrived."""

    def __init__(self, data):
        self._lib = _ffi.lib()
        self.size = len(data)
        self.ptr = self._lib.rgx_host_alloc(self.size + 64)
        if not self.ptr:
            raise RegtoolsError(1, "regtools_amd: no page-locked memory for %d bytes\n" % self.size)
        C.memmove(self.ptr, data if isinstance(data, bytes) else bytes(data), self.size)

    def close(self):
        if self.ptr:
            self._lib.rgx_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class JunctionsExtractor(object):
    """JunctionsExtractor(bam, region, strandness, strand_tag, min_anchor, min_intron, max_intron, ref)."""

    def __init__(self, bam="NA", region=".", strandness=-1, strand_tag="XS", min_anchor_length=8,
                 min_intron_length=70, max_intron_length=500000, ref="NA", ctx=None, device=0,
                 shard=0, n_shards=1, output_barcodes_file="NA", barcode_tag="CB"):
        self.bam_, self.region_, self.strandness_, self.strand_tag_ = bam, region, strandness, strand_tag
        self.min_anchor_length_, self.min_intron_length_, self.max_intron_length_ = min_anchor_length, min_intron_length, max_intron_length
        self.ref_ = ref
        self.shard, self.n_shards = shard, n_shards
        self.output_barcodes_file_, self.barcode_tag_ = output_barcodes_file, barcode_tag      # junctions_extractor.h:174, :182
        self._ctx, self._device = ctx, device
        self._table = None
        self.stats = {}

    # -- parse_options (junctions_extractor.cc:42-122) ---------------------------------------------------------
    def parse_options(self, argv):
        import getopt
        try:
            opts, args = getopt.getopt(list(argv), "ha:m:M:o:r:t:s:b:")
        except getopt.GetoptError:
            raise RegtoolsError(1, "Error parsing inputs!(1)\n\n")
        self.output_file_ = "NA"
        for k, v in opts:
            if k == "-h":
                raise RegtoolsError(0, "help")
            elif k == "-a": self.min_anchor_length_ = _atoi(v)
            elif k == "-m": self.min_intron_length_ = _atoi(v)
            elif k == "-M": self.max_intron_length_ = _atoi(v)
            elif k == "-o": self.output_file_ = v
            elif k == "-r": self.region_ = v
            elif k == "-t": self.strand_tag_ = v
            elif k == "-s":
                if v not in STRANDNESS:
                    raise RegtoolsError(1, "Unrecognized strandness argument!\n\n")
                self.strandness_ = STRANDNESS[v]
            elif k == "-b": self.output_barcodes_file_ = v
        if len(args) >= 1: self.bam_ = args[0]
        if len(args) >= 2: self.ref_ = args[1]
        if len(args) > 2 or self.bam_ == "NA":
            raise RegtoolsError(1, "Error parsing inputs!(2)\n\n")
        if self.strandness_ == -1:
            raise RegtoolsError(1, "Please supply strandness mode with '-s' option!\n\n")
        if self.strandness_ == 3 and self.ref_ == "NA":
            raise RegtoolsError(1, "Strandness mode 'intron-motif' requires a fasta file!\n\n")
        return 0

    def get_bam(self):
        return self.bam_

    def _params(self):
        p = _ffi.ExtractParams()
        _ffi.lib().rgx_extract_params_default(C.byref(p))
        self._region_b = self.region_.encode()
        p.region = self._region_b
        p.strandness = self.strandness_
        tag = (self.strand_tag_.encode() + b"\0\0")[:2]
        p.strand_tag = tag
        p.min_anchor, p.min_intron, p.max_intron = self.min_anchor_length_ & 0xffffffff, self.min_intron_length_ & 0xffffffff, self.max_intron_length_ & 0xffffffff
        self._fa_b = None if self.ref_ == "NA" else self.ref_.encode()
        p.fasta_path = self._fa_b
        p.shard, p.n_shards = self.shard, self.n_shards
        p.barcodes = 0 if self.output_barcodes_file_ == "NA" else 1
        p.barcode_tag = (self.barcode_tag_.encode() + b"\0\0")[:2]
        return p

    # -- identify_junctions_from_BAM (junctions_extractor.cc:500-535) -------------------------------------------------
    def identify_junctions_from_BAM(self, bam_bytes=None, bai_bytes=None, device_ptr=None, device_len=0, host_ptr=None, host_len=0):
        """bam_bytes: the file as a bytes object; host_ptr/host_len: the file at a host address (page-locked memory from
        regtools_amd.PinnedBuffer lets the upload overlap the inflate, rgx_extract_mem); device_ptr/device_len: the file already in HBM."""
        lib = _ffi.lib()
        if self._ctx is None:
            self._ctx = Context(self._device)
        p = self._params()
        tab = C.POINTER(_ffi.JunctionTable)()
        err = C.create_string_buffer(512)
        if host_ptr is not None:
            rc = lib.rgx_extract_mem(self._ctx._h, C.c_void_p(host_ptr), host_len, bai_bytes, len(bai_bytes), C.byref(p), C.byref(tab), err, len(err))
        elif bam_bytes is None and device_ptr is None:
            rc = lib.rgx_extract(self._ctx._h, self.bam_.encode(), C.byref(p), C.byref(tab), err, len(err))
        elif device_ptr is None:
            rc = lib.rgx_extract_mem(self._ctx._h, bam_bytes, len(bam_bytes), bai_bytes, len(bai_bytes), C.byref(p), C.byref(tab), err, len(err))
        else:
            rc = lib.rgx_extract_device(self._ctx._h, C.c_void_p(device_ptr), device_len, bai_bytes, len(bai_bytes),
                                        C.byref(p), C.byref(tab), err, len(err))
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())
        self._free()
        self._table = tab
        t = tab.contents
        self.stats = dict(n_records=t.n_records, n_events=t.n_events, n_junctions=t.n, inflated_bytes=t.inflated_bytes,
                          compressed_bytes=t.compressed_bytes, n_members=t.n_members, ms_total=t.ms_total, ms_inflate=t.ms_inflate,
                          ms_records=t.ms_records, ms_scan=t.ms_scan, ms_reduce=t.ms_reduce, framing_sweeps=t.framing_sweeps,
                          ms_barcodes=t.ms_barcodes, stream_ended=bool(t.stream_ended), ms_inflate_launch=t.ms_inflate_launch)
        return 0

    def _free(self):
        if self._table:
            _ffi.lib().rgx_table_free(self._table)
            self._table = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    # -- get_all_junctions (junctions_extractor.cc:238-246): every row, sorted ----------------------------------------
    def get_all_junctions(self):
        t = self._table.contents
        out = []
        for i in range(t.n):
            j = Junction()
            j.chrom = t.ref_name[t.tid[i]].decode()
            j.start, j.end, j.thick_start, j.thick_end = t.start[i], t.end[i], t.thick_start[i], t.thick_end[i]
            j.name = "JUNC%08d" % t.name_index[i]
            j.read_count = t.read_count[i]
            j.strand = t.strand[i].decode("latin-1")
            j.has_left_min_anchor, j.has_right_min_anchor = bool(t.left_ok[i]), bool(t.right_ok[i])
            out.append(j)
        return out

    # -- print_all_junctions (junctions_extractor.cc:249-280): BED12 of the rows anchored on both sides ------------------
    def bed12(self, only_anchored=True):
        return _ffi.format_bed12(self._table, only_anchored)

    # -- Junction::print_barcodes per printed row (junctions_extractor.h:99-111, cc:272-273) ---------------------------------
    def barcodes_text(self, only_anchored=True):
        lib = _ffi.lib()
        n = lib.rgx_table_format_barcodes(self._table, 1 if only_anchored else 0, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib.rgx_table_format_barcodes(self._table, 1 if only_anchored else 0, buf, n)
        return buf.raw[:n]

    def get_barcodes(self, insertion_order=False):
        """Per row (get_all_junctions order): [(barcode, count), ...] in the order print_barcodes writes them, or (insertion_order) in the
        order the junction first saw them -- the order to refill a Junction::barcodes map in."""
        t = self._table.contents
        if not t.bc_row_begin:
            return [[] for _ in range(t.n)]
        if insertion_order:
            listed = self.get_barcodes()
            out, k = [], 0
            for m in listed:
                o = [None] * len(m)
                for e in m:
                    o[t.bc_insert_rank[k]] = e
                    k += 1
                out.append(o)
            return out
        text = C.string_at(t.bc_text, t.bc_str_begin[t.bc_row_begin[t.n]]) if t.n else b""
        return [[(text[t.bc_str_begin[k]:t.bc_str_begin[k + 1]], t.bc_count[k]) for k in range(t.bc_row_begin[i], t.bc_row_begin[i + 1])]
                for i in range(t.n)]

    def print_all_junctions(self, out=None):
        data = self.bed12(True)
        if self.output_barcodes_file_ != "NA":
            try:
                with open(self.output_barcodes_file_, "wb") as f:      # an unopenable file is silently skipped upstream (cc:255-256, :272)
                    f.write(self.barcodes_text(True))
            except OSError:
                pass
        target = getattr(self, "output_file_", "NA")
        if target != "NA":
            with open(target, "wb") as f:
                f.write(data)
        elif out is not None:
            out.write(data.decode("latin-1"))
        else:
            import sys
            sys.stdout.write(data.decode("latin-1"))

    @property
    def table(self):
        return self._table


class _BorrowedContext(object):
    """A context the pipeline owns, as far as the merge calls need one (they take ctx._h)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)


class Pipeline(object):
    """rgx_pipeline: several files in flight on one device (depth contexts, file k on context k mod depth).  submit() returns a ticket at once,
    wait(ticket) a JunctionsExtractor holding that file's table -- the same table a sequential identify_junctions_from_BAM gives.  The caller
    keeps the input buffers alive until wait() returns (bytes objects are pinned to the ticket here).
    GPU_MAX_HW_QUEUES in the environment BEFORE the process's first HIP call (importing torch counts): 16 or more (eight per file in flight) lets the files'
    DEFLATE launches go out at once, ~7 % per file; depth > 2 needs 4 x depth."""

    def __init__(self, device=0, depth=2):
        self._lib = _ffi.lib()
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self._lib.rgx_pipeline_create(device, depth, C.byref(self._h), err, len(err))
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())
        self._open = {}

    def submit(self, bam_bytes=None, bai_bytes=None, host_ptr=None, host_len=0, **kw):
        je = JunctionsExtractor(**kw)
        p = je._params()
        ticket = C.c_uint64()
        err = C.create_string_buffer(512)
        if host_ptr is not None:
            rc = self._lib.rgx_extract_submit(self._h, C.c_void_p(host_ptr), host_len, bai_bytes, len(bai_bytes), C.byref(p), C.byref(ticket), err, len(err))
        else:
            rc = self._lib.rgx_extract_submit(self._h, bam_bytes, len(bam_bytes), bai_bytes, len(bai_bytes), C.byref(p), C.byref(ticket), err, len(err))
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())
        self._open[ticket.value] = (je, bam_bytes, bai_bytes)
        return ticket.value

    def wait(self, ticket):
        je, _, _ = self._open.pop(ticket)
        tab = C.POINTER(_ffi.JunctionTable)()
        err = C.create_string_buffer(512)
        rc = self._lib.rgx_extract_wait(self._h, ticket, C.byref(tab), err, len(err))
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())
        je._table = tab
        je._ctx = _BorrowedContext(self._lib.rgx_pipeline_ctx(self._h, ticket))      # (rgx_pipeline_ctx: where this file's rows still lie, for a device-side merge)
        t = tab.contents
        je.stats = dict(n_records=t.n_records, n_events=t.n_events, n_junctions=t.n, ms_total=t.ms_total, stream_ended=bool(t.stream_ended))
        return je

    def close(self):
        if self._h:
            self._lib.rgx_pipeline_destroy(self._h)
            self._h = C.c_void_p()
            self._open = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def extract_multi(devices, bam=None, bam_bytes=None, bai_bytes=None, host_ptr=None, host_len=0, **kw):
    """rgx_extract_multi: `junctions extract` sharded over `devices` (a host thread per device, one RCCL gather of the shards' rows to
    devices[0], merge there).  kw: the JunctionsExtractor constructor's arguments.  host_ptr / host_len: the file in (page-locked) host memory
    instead of bam_bytes.  Returns a distributed.MergedTable-like object."""
    from . import distributed
    lib = _ffi.lib()
    je = JunctionsExtractor(bam=bam or "NA", **kw)
    p = je._params()
    devs = (C.c_int * len(devices))(*devices)
    tab = C.POINTER(_ffi.JunctionTable)()
    err = C.create_string_buffer(512)
    if host_ptr is not None:
        rc = lib.rgx_extract_multi_mem(devs, len(devices), C.c_void_p(host_ptr), host_len, bai_bytes, len(bai_bytes), C.byref(p), C.byref(tab), err, len(err))
    elif bam_bytes is not None:
        rc = lib.rgx_extract_multi_mem(devs, len(devices), bam_bytes, len(bam_bytes), bai_bytes, len(bai_bytes), C.byref(p), C.byref(tab), err, len(err))
    else:
        rc = lib.rgx_extract_multi(devs, len(devices), bam.encode(), C.byref(p), C.byref(tab), err, len(err))
    if rc != 0:
        raise RegtoolsError(rc, err.value.decode())
    return distributed.MergedTable(tab)


def junctions_extract(argv):
    """junctions_extract() of src/junctions/junctions_main.cc:45-59: returns the process exit code."""
    import sys
    je = JunctionsExtractor()
    try:
        je.parse_options(argv)
        je.identify_junctions_from_BAM()
        je.print_all_junctions()
    except RegtoolsError as e:
        if e.code == 0:
            return 0
        sys.stderr.write(str(e))
        return 1
    return 0
