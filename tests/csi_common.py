"""BAI -> CSI conversion for hand-made test inputs (TEST INFRASTRUCTURE ONLY).

Writes what `samtools index -c` (min_shift 14, depth 5) would for the same BAM: "CSI\\1", min_shift, depth, l_aux = 0, then per
reference the bins with their u64 loffset (the linear-index entry of the bin's first 16 KiB window, 0 for the pseudo-bin; hts.c:1330-1350
update_loff) and no linear index, then n_no_coor; the whole file BGZF-compressed (hts.c:1441-1452 hts_idx_save writes through bgzf)."""
import os
import struct

import bamio

META_BAI = 37450


def parse_bai(d):
    assert d[:4] == b"BAI\1"
    n_ref = struct.unpack_from("<i", d, 4)[0]
    p = 8
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", d, p)[0]; p += 4
        bins = []
        for _ in range(n_bin):
            b, nc = struct.unpack_from("<Ii", d, p); p += 8
            chunks = [struct.unpack_from("<QQ", d, p + 16 * k) for k in range(nc)]
            p += 16 * nc
            bins.append((b, chunks))
        n_intv = struct.unpack_from("<i", d, p)[0]; p += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, d, p)); p += 8 * n_intv
        refs.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", d, p)[0] if p + 8 <= len(d) else None
    return refs, n_no_coor


def _bin_first_window(b):
    for level in range(5, -1, -1):
        t = ((1 << (3 * level)) - 1) // 7
        if b >= t:
            return (b - t) << (3 * (5 - level))
    raise AssertionError


def csi_bytes(bai, compress=True, aux=b""):
    refs, n_no_coor = parse_bai(bai)
    o = bytearray(b"CSI\1" + struct.pack("<iii", 14, 5, len(aux)) + aux + struct.pack("<i", len(refs)))
    for bins, lin in refs:
        o += struct.pack("<i", len(bins))
        for b, chunks in bins:
            if b == META_BAI:
                loff = 0
            else:
                w = _bin_first_window(b)
                loff = lin[min(w, len(lin) - 1)] if lin else 0
            o += struct.pack("<IQi", b, loff, len(chunks))      # with depth 5 the pseudo-bin keeps its number
            for c in chunks:
                o += struct.pack("<QQ", *c)
    if n_no_coor is not None:
        o += struct.pack("<Q", n_no_coor)
    if not compress:
        return bytes(o)
    out = b"".join(bamio.bgzf_member(bytes(o[i:i + 0xff00])) for i in range(0, len(o), 0xff00))
    return out + bamio.EOF_MARKER


def bai_to_csi(bam_path, keep_bai=False, compress=True, aux=b""):
    """<bam>.bai -> <bam>.csi"""
    bai = open(bam_path + ".bai", "rb").read()
    with open(bam_path + ".csi", "wb") as f:
        f.write(csi_bytes(bai, compress, aux))
    if not keep_bai:
        os.remove(bam_path + ".bai")
    return bam_path + ".csi"
