"""Tiny pure-python BAM/BGZF reader+writer for hand-made test inputs (TEST INFRASTRUCTURE ONLY)."""
import struct
import zlib

OPS = "MIDNSHP=XB"
EOF_MARKER = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def bgzf_member(data, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8)
    payload = c.compress(data) + c.flush()
    total = 18 + len(payload) + 8
    assert total <= 65536
    return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", total - 1) + payload +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def bgzf_members(path_or_bytes):
    d = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    off = 0
    while off + 18 <= len(d):
        bl = struct.unpack_from("<H", d, off + 16)[0] + 1
        yield off, d[off + 18: off + bl - 8], struct.unpack_from("<I", d, off + bl - 4)[0]
        off += bl


def inflate_all(path_or_bytes):
    return b"".join(zlib.decompress(p, -15) for _, p, _ in bgzf_members(path_or_bytes))


def parse_cigar(s):
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num), OPS.index(ch)))
            num = ""
    return out


def header_bytes(contigs, text=None):
    if text is None:
        text = "@HD\tVN:1.4\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    t = text.encode()
    h = b"BAM\1" + struct.pack("<i", len(t)) + t + struct.pack("<i", len(contigs))
    for name, ln in contigs:
        n = name.encode() + b"\0"
        h += struct.pack("<i", len(n)) + n + struct.pack("<i", ln)
    return h


def record(tid, pos, cigar, flag=0, qname="r", aux=b"", mapq=60, l_seq=None, n_cigar_override=None):
    ops = parse_cigar(cigar) if isinstance(cigar, str) else cigar
    qn = qname.encode() + b"\0"
    if l_seq is None:
        l_seq = sum(l for l, o in ops if o in (0, 1, 4, 7, 8))
    cig = b"".join(struct.pack("<I", l << 4 | o) for l, o in ops)
    seq = b"\x11" * ((l_seq + 1) // 2)
    qual = b"\xff" * l_seq
    nc = len(ops) if n_cigar_override is None else n_cigar_override
    body = struct.pack("<iiIIiiii", tid, pos, (4680 << 16) | (mapq << 8) | len(qn), (flag << 16) | nc, l_seq, -1, -1, 0) + qn + cig + seq + qual + aux
    return struct.pack("<i", len(body)) + body


def tagA(tag, ch):
    return tag.encode() + b"A" + (ch.encode() if isinstance(ch, str) else bytes([ch]))


def tagZ(tag, s):
    return tag.encode() + b"Z" + s.encode() + b"\0"


def write_bam(path, contigs, records, block=0xff00, level=6, extra_members_after_header=(), split_points=()):
    """records: list of bytes (already serialised, coordinate-sorted). Members are cut every `block` bytes of the
    record stream (records may straddle members), plus optional explicit extra members (e.g. an empty one)."""
    out = bytearray()
    hdr = header_bytes(contigs)
    for k in range(0, len(hdr), 0xff00):                 # a header with thousands of contigs spans several members
        out += bgzf_member(hdr[k:k + 0xff00], level)
    for m in extra_members_after_header:
        out += m
    stream = b"".join(records)
    cuts = sorted(set([p for p in split_points if 0 < p < len(stream)]))
    pos = 0
    pieces = []
    for c in cuts + [len(stream)]:
        seg = stream[pos:c]
        for k in range(0, len(seg), block):
            pieces.append(seg[k:k + block])
        pos = c
    for p in pieces:
        out += bgzf_member(p, level)
    out += EOF_MARKER
    with open(path, "wb") as f:
        f.write(out)
    return bytes(out)


def split_records(inflated):
    """-> (header_bytes, [record bytes])"""
    l_text = struct.unpack_from("<i", inflated, 4)[0]
    q = 8 + l_text
    n_ref = struct.unpack_from("<i", inflated, q)[0]
    q += 4
    contigs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", inflated, q)[0]
        name = inflated[q + 4:q + 4 + ln - 1].decode()
        L = struct.unpack_from("<i", inflated, q + 4 + ln)[0]
        contigs.append((name, L))
        q += 8 + ln
    recs = []
    while q + 4 <= len(inflated):
        bs = struct.unpack_from("<i", inflated, q)[0]
        recs.append(inflated[q:q + 4 + bs])
        q += 4 + bs
    return contigs, recs
