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


# ---- a CSI of ANY geometry, built from the BAM's records (what `samtools index -c -m MIN_SHIFT` writes, up to chunk merging) ----------------
def _ref_len(rec):
    n_cigar = struct.unpack_from("<H", rec, 16)[0]
    l_qname = rec[12]
    flag = struct.unpack_from("<H", rec, 18)[0]
    if flag & 4 or not n_cigar:
        return 1
    ln = 0
    for k in range(n_cigar):
        c = struct.unpack_from("<I", rec, 36 + l_qname + 4 * k)[0]
        if (c & 15) in (0, 2, 3, 7, 8):
            ln += c >> 4
    return max(ln, 1)


def _reg2bin(beg, end, min_shift, depth):
    end -= 1
    s, t = min_shift, ((1 << (3 * depth)) - 1) // 7
    for level in range(depth, 0, -1):
        if beg >> s == end >> s:
            return t + (beg >> s)
        s += 3
        t -= 1 << (3 * (level - 1))
    return 0


def csi_from_bam(bam, min_shift, depth, compress=True):
    """bam: bytes of a coordinate-sorted BAM.  Per reference: bins -> chunks (runs of consecutive records of one bin), each bin's loff = the
    smallest record start among the records overlapping the bin's first window or, when there is none, the next window that has one
    (hts.c:1193-1215 update_loff), the pseudo-bin n_bins + 1 with the reference's file range and read counts."""
    import bisect
    members = list(bamio.bgzf_members(bam))
    upos, acc = [], 0
    for _, _, isz in members:
        upos.append(acc); acc += isz
    inflated = bamio.inflate_all(bam)
    contigs, recs = bamio.split_records(inflated)

    def voff(q):
        if q >= acc:
            return (members[-1][0] + 18 + len(members[-1][1]) + 8) << 16
        m = bisect.bisect_right(upos, q) - 1
        while members[m][2] == 0:
            m += 1
        return members[m][0] << 16 | (q - upos[m])
    q = len(inflated) - sum(len(r) for r in recs)
    per_ref = [dict(bins={}, lin={}, beg=None, end=None, mapped=0, unmapped=0) for _ in contigs]
    n_no_coor = 0
    last = (None, None)
    for r in recs:
        tid, pos = struct.unpack_from("<ii", r, 4)
        u, v = voff(q), voff(q + len(r))
        q += len(r)
        if tid < 0:
            n_no_coor += 1
            continue
        R = per_ref[tid]
        flag = struct.unpack_from("<H", r, 18)[0]
        R["unmapped" if flag & 4 else "mapped"] += 1
        if R["beg"] is None:
            R["beg"] = u
        R["end"] = v
        end = pos + _ref_len(r)
        b = _reg2bin(pos, end, min_shift, depth)
        chunks = R["bins"].setdefault(b, [])
        if last == (tid, b) and chunks and chunks[-1][1] == u:
            chunks[-1][1] = v
        else:
            chunks.append([u, v])
        last = (tid, b)
        for w in range(pos >> min_shift, ((end - 1) >> min_shift) + 1):
            if w not in R["lin"] or R["lin"][w] > u:
                R["lin"][w] = u
    n_bins = ((1 << (3 * depth + 3)) - 1) // 7
    o = bytearray(b"CSI\1" + struct.pack("<iii", min_shift, depth, 0) + struct.pack("<i", len(contigs)))
    for R in per_ref:
        wins = sorted(R["lin"])

        def loff(b):
            level, t = 0, 0
            while level < depth and b >= t + (1 << (3 * level)):
                t += 1 << (3 * level); level += 1
            w0 = (b - t) << (3 * (depth - level))
            k = bisect.bisect_left(wins, w0)
            return R["lin"][wins[k]] if k < len(wins) else 0
        o += struct.pack("<i", len(R["bins"]) + (1 if R["beg"] is not None else 0))
        for b in sorted(R["bins"]):
            o += struct.pack("<IQi", b, loff(b), len(R["bins"][b]))
            for c in R["bins"][b]:
                o += struct.pack("<QQ", *c)
        if R["beg"] is not None:
            o += struct.pack("<IQi", n_bins + 1, 0, 2) + struct.pack("<QQQQ", R["beg"], R["end"], R["mapped"], R["unmapped"])
    o += struct.pack("<Q", n_no_coor)
    if not compress:
        return bytes(o)
    return b"".join(bamio.bgzf_member(bytes(o[i:i + 0xff00])) for i in range(0, len(o), 0xff00)) + bamio.EOF_MARKER
