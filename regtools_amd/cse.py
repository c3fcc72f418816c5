"""Host-side mirror of the reference's `CisSpliceEffectsIdentifier` (src/cis-splice-effects/cis_splice_effects_identifier.{h,cc})
over the C ABI: same option letters, defaults, error texts and outputs (-o TSV, -v VCF, -j BED12)."""
import ctypes as C
import getopt

from . import _ffi
from .extractor import Context, RegtoolsError, STRANDNESS


class CisSpliceEffectsIdentifier(object):
    def __init__(self, ctx=None, device=0, devices=None):
        """devices = [d0, d1, ...]: the BAM's extraction is sharded over these GPUs (rgx_identify_multi; a device may be listed twice), everything
        behind it runs on d0.  The outputs do not depend on the list."""
        self._ctx, self._device, self._devices = ctx, device, list(devices) if devices else None
        self.p = _ffi.IdentifyParams()
        _ffi.lib().rgx_identify_params_default(C.byref(self.p))
        self.stats = {}
        self._keep = []

    # cis_splice_effects_identifier.cc:112-219
    def parse_options(self, argv):
        try:
            opts, args = getopt.getopt(list(argv), "o:w:v:j:e:Ei:ISht:s:a:m:M:b:C")
        except getopt.GetoptError:
            raise RegtoolsError(1, "Error parsing inputs!(1)\n\n")
        p = self.p

        def keep(s):
            b = s.encode()
            self._keep.append(b)
            return b
        for k, v in opts:
            if k == "-h": raise RegtoolsError(0, "help")
            elif k == "-o": p.out_tsv = keep(v)
            elif k == "-w": p.window = int(v) & 0xffffffff
            elif k == "-v": p.out_vcf = keep(v)
            elif k == "-j": p.out_bed = keep(v)
            elif k == "-i": p.intronic_min = int(v) & 0xffffffff
            elif k == "-e": p.exonic_min = int(v) & 0xffffffff
            elif k == "-I": p.all_intronic = 1
            elif k == "-E": p.all_exonic = 1
            elif k == "-S": p.skip_single = 0
            elif k == "-s":
                if v not in STRANDNESS:
                    raise RegtoolsError(1, "Unrecognized strandness argument!\n\n")
                p.strandness = STRANDNESS[v]
            elif k == "-t": p.strand_tag = (v.encode() + b"\0\0")[:2]
            elif k == "-a": p.min_anchor = int(v) & 0xffffffff
            elif k == "-m": p.min_intron = int(v) & 0xffffffff
            elif k == "-M": p.max_intron = int(v) & 0xffffffff
            elif k == "-b": raise RegtoolsError(1, "regtools_amd: -b (single-cell barcodes) is outside the accelerated path\n\n")
            elif k == "-C": p.override_motif = 1
        if len(args) != 4:
            raise RegtoolsError(1, "Error parsing inputs!(2)\n\n")
        if p.strandness == -1:
            raise RegtoolsError(1, "Please supply strand specificity with '-s' option!\n\n")
        import os
        if not all(os.path.exists(a) for a in args):
            raise RegtoolsError(1, "Please make sure input files exist.\n\n")
        p.vcf_path, p.bam_path, p.fasta_path, p.gtf_path = [keep(a) for a in args]

    # cis_splice_effects_identifier.cc:256-312
    def identify(self):
        if self._devices and len(self._devices) > 1:
            st = _ffi.IdentifyStats()
            err = C.create_string_buffer(512)
            dev = (C.c_int * len(self._devices))(*self._devices)
            rc = _ffi.lib().rgx_identify_multi(dev, len(self._devices), C.byref(self.p), C.byref(st), err, len(err))
            if rc != 0:
                raise RegtoolsError(rc, err.value.decode())
            self.stats = {n: getattr(st, n) for n, _ in st._fields_}
            return 0
        return self._run(_ffi.lib().rgx_identify)

    def _run(self, fn):
        if self._ctx is None:
            self._ctx = Context(self._device)
        st = _ffi.IdentifyStats()
        err = C.create_string_buffer(512)
        rc = fn(self._ctx._h, C.byref(self.p), C.byref(st), err, len(err))
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())
        self.stats = {n: getattr(st, n) for n, _ in st._fields_}
        return 0


class CisSpliceEffectsAssociator(CisSpliceEffectsIdentifier):
    """Mirror of `CisSpliceEffectsAssociator` (src/cis-splice-effects/cis_splice_effects_associator.{h,cc}): junctions come from a BED12."""

    # cis_splice_effects_associator.cc:104-180
    def parse_options(self, argv):
        try:
            opts, args = getopt.getopt(list(argv), "o:w:v:j:e:Ei:ISha:m:M:")
        except getopt.GetoptError:
            raise RegtoolsError(1, "Error parsing inputs!(1)\n\n")
        if len(args) != 4:
            raise RegtoolsError(1, "Error parsing inputs!(2)\n\n")
        CisSpliceEffectsIdentifier.parse_options(self, ["-s", "XS"] + list(argv))
        self.p.bed_path = self.p.bam_path

    # cis_splice_effects_associator.cc:234-276
    def associate(self):
        return self._run(_ffi.lib().rgx_associate)


class VariantsAnnotator(object):
    """Mirror of `VariantsAnnotator::annotate_vcf` (src/variants/variants_annotator.cc:48-110, 541-550)."""

    def __init__(self, ctx=None, device=0):
        self._ctx, self._device = ctx, device
        self.p = _ffi.IdentifyParams()
        _ffi.lib().rgx_identify_params_default(C.byref(self.p))
        self.stats = {}
        self._keep = []

    def parse_options(self, argv):
        try:
            opts, args = getopt.getopt(list(argv), "e:Ei:ISho:")
        except getopt.GetoptError:
            raise RegtoolsError(1, "Error parsing inputs!(1)\n\n")
        p = self.p
        for k, v in opts:
            if k == "-h": raise RegtoolsError(0, "help")
            elif k == "-i": p.intronic_min = int(v) & 0xffffffff
            elif k == "-e": p.exonic_min = int(v) & 0xffffffff
            elif k == "-I": p.all_intronic = 1
            elif k == "-E": p.all_exonic = 1
            elif k == "-S": p.skip_single = 0
            elif k == "-o":
                self._keep.append(v.encode()); p.out_vcf = self._keep[-1]
        if len(args) < 2:
            raise RegtoolsError(1, "Error parsing inputs!(2)\n\n")
        self._keep += [args[0].encode(), args[1].encode()]
        p.vcf_path, p.gtf_path = self._keep[-2], self._keep[-1]

    def annotate_vcf(self):
        if self._ctx is None:
            self._ctx = Context(self._device)
        st = _ffi.IdentifyStats()
        err = C.create_string_buffer(512)
        rc = _ffi.lib().rgx_variants_annotate(self._ctx._h, C.byref(self.p), C.byref(st), err, len(err))
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())
        self.stats = {n: getattr(st, n) for n, _ in st._fields_}
        return 0


class JunctionsAnnotator(object):
    """Mirror of the `junctions annotate` driver (src/junctions/junctions_main.cc:62-93, junctions_annotator.cc:385-428)."""

    def __init__(self, ctx=None, device=0):
        self._ctx, self._device = ctx, device
        self.bed = self.ref = self.gtf = self.output_file = None
        self.n_rows = 0

    def parse_options(self, argv):
        try:
            opts, args = getopt.getopt(list(argv), "So:h")
        except getopt.GetoptError:
            raise RegtoolsError(1, "Error parsing inputs!(1)\n\n")
        for k, v in opts:
            if k == "-h": raise RegtoolsError(0, "help")
            elif k == "-o": self.output_file = v
            elif k == "-S": self.skip_single_exon_genes = False
        if len(args) != 3:
            raise RegtoolsError(1, "Error parsing inputs!(2)\n\n")
        self.bed, self.ref, self.gtf = args

    def annotate(self):
        if self._ctx is None:
            self._ctx = Context(self._device)
        err = C.create_string_buffer(512)
        n = C.c_uint64(0)
        rc = _ffi.lib().rgx_junctions_annotate_opts(self._ctx._h, self.bed.encode(), self.ref.encode(), self.gtf.encode(),
                                                    self.output_file.encode() if self.output_file else None,
                                                    0 if getattr(self, "skip_single_exon_genes", True) else 1, C.byref(n), err, len(err))
        self.n_rows = n.value
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())
        return 0


def cis_splice_effects_identify(argv):
    """cis_splice_effects_identify() of src/cis-splice-effects/cis_splice_effects_main.cc:35-51 -> process exit code."""
    import sys
    ci = CisSpliceEffectsIdentifier()
    try:
        ci.parse_options(argv)
        ci.identify()
    except RegtoolsError as e:
        if e.code == 0:
            return 0
        sys.stderr.write(str(e))
        return 1
    return 0
