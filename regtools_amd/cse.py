"""Host-side mirror of the reference's `CisSpliceEffectsIdentifier` (src/cis-splice-effects/cis_splice_effects_identifier.{h,cc})
over the C ABI: same option letters, defaults, error texts and outputs (-o TSV, -v VCF, -j BED12)."""
import ctypes as C
import getopt

from . import _ffi
from .extractor import Context, RegtoolsError, STRANDNESS


class CisSpliceEffectsIdentifier(object):
    def __init__(self, ctx=None, device=0):
        self._ctx, self._device = ctx, device
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
        if self._ctx is None:
            self._ctx = Context(self._device)
        st = _ffi.IdentifyStats()
        err = C.create_string_buffer(512)
        rc = _ffi.lib().rgx_identify(self._ctx._h, C.byref(self.p), C.byref(st), err, len(err))
        if rc != 0:
            raise RegtoolsError(rc, err.value.decode())
        self.stats = {n: getattr(st, n) for n, _ in st._fields_}
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
