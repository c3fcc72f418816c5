"""configs[2] helper (TEST INFRASTRUCTURE): the eight coordinate slices of SURVEY 8d's "Config 3" joined into ONE BAM at the byte level.
A slice file is header member(s) | record members | EOF marker (synth_bam.cpp); slice k's reads lie in the k-th coordinate window of the
linear genome, so slice 0 without its EOF marker + the record members of slices 1.. + one EOF marker is a coordinate-sorted BAM whose
record order is (slice, index in slice) = the order the merged shards claim to have."""
import struct
import zlib

EOF_LEN = 28


def header_span(bam):
    """compressed length of the BGZF members that hold the BAM header (the header ends on a member boundary in the generator's files)"""
    off, inflated, need = 0, b"", None
    while True:
        bl = struct.unpack_from("<H", bam, off + 16)[0] + 1
        inflated += zlib.decompress(bam[off + 18: off + bl - 8], -15)
        off += bl
        if need is None and len(inflated) >= 12:
            l_text = struct.unpack_from("<i", inflated, 4)[0]
            if len(inflated) >= 12 + l_text:
                q, n_ref = 12 + l_text, struct.unpack_from("<i", inflated, 8 + l_text)[0]
                ok = True
                for _ in range(n_ref):
                    if len(inflated) < q + 4:
                        ok = False
                        break
                    q += 8 + struct.unpack_from("<i", inflated, q)[0]
                if ok and len(inflated) >= q:
                    need = q
        if need is not None:
            assert len(inflated) == need, "the header does not end on a member boundary"
            return off


def concat_slices(slices):
    out = bytearray(slices[0][:-EOF_LEN])
    for s in slices[1:]:
        out += s[header_span(s):-EOF_LEN]
    out += slices[0][-EOF_LEN:]
    return bytes(out)


def parse_bai(b):
    assert b[:4] == b"BAI\1"
    n_ref = struct.unpack_from("<i", b, 4)[0]
    q, refs = 8, []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", b, q)[0]
        q += 4
        bins = {}
        for _ in range(n_bin):
            bn, n_chunk = struct.unpack_from("<Ii", b, q)
            q += 8
            bins[bn] = [struct.unpack_from("<QQ", b, q + 16 * k) for k in range(n_chunk)]
            q += 16 * n_chunk
        n_intv = struct.unpack_from("<i", b, q)[0]
        q += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, b, q))
        q += 8 * n_intv
        refs.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", b, q)[0] if q + 8 <= len(b) else 0
    return refs, n_no_coor


def merge_bai(slice_bams, slice_bais):
    """The index of concat_slices(slice_bams) from the slices' own indexes: every virtual offset of slice k moves by the compressed bytes in
    front of its record members; the chunk lists of a bin are joined in slice order, the linear index takes the smallest offset per window
    (zeros filled forward, as the indexer writes them), the pseudo-bin 37450 the outermost offsets and the summed counts."""
    META = 37450
    shift, at = [], len(slice_bams[0]) - EOF_LEN
    for k, s in enumerate(slice_bams):
        hs = header_span(s)
        if k == 0:
            shift.append(0)
        else:
            shift.append(at - hs)
            at += len(s) - hs - EOF_LEN
    parsed = [parse_bai(b) for b in slice_bais]
    n_ref = len(parsed[0][0])
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for r in range(n_ref):
        bins, lin, meta = {}, [], None
        for k, (refs, _) in enumerate(parsed):
            d = shift[k] << 16
            b_k, l_k = refs[r]
            for bn in sorted(b_k):
                if bn == META:
                    (beg, end), (nm, nu) = b_k[bn]
                    meta = [beg + d, end + d, nm, nu] if meta is None else [min(meta[0], beg + d), max(meta[1], end + d), meta[2] + nm, meta[3] + nu]
                else:
                    bins.setdefault(bn, []).extend((u + d, v + d) for u, v in b_k[bn])
            if len(l_k) > len(lin):
                lin += [0] * (len(l_k) - len(lin))
            for i, v in enumerate(l_k):
                if v and (lin[i] == 0 or v + d < lin[i]):
                    lin[i] = v + d
        for i in range(1, len(lin)):
            if lin[i] == 0:
                lin[i] = lin[i - 1]
        out += struct.pack("<i", len(bins) + (1 if meta else 0))
        for bn in sorted(bins):
            out += struct.pack("<Ii", bn, len(bins[bn])) + b"".join(struct.pack("<QQ", u, v) for u, v in bins[bn])
        if meta:
            out += struct.pack("<IiQQQQ", META, 2, *meta)
        out += struct.pack("<i", len(lin)) + struct.pack("<%dQ" % len(lin), *lin)
    out += struct.pack("<Q", sum(p[1] for p in parsed))
    return bytes(out)
