"""A minimal BCF2.2 writer for tests (the product and the real reference both READ what this writes; nothing here is product code).
Typed values as the BCF2 specification lays them out: descriptor byte = count << 4 | type, counts >= 15 spill into a typed integer."""
import struct

import bamio

INT8, INT16, INT32, FLOAT, CHAR = 1, 2, 3, 5, 7
I8_MISSING, I8_END = -128, -127
I16_MISSING, I16_END = -32768, -32767
I32_MISSING, I32_END = -2 ** 31, -2 ** 31 + 1
FLOAT_MISSING, FLOAT_END = 0x7F800001, 0x7F800002
_FMT = {INT8: "<b", INT16: "<h", INT32: "<i"}


def typed_int(x):
    t = INT8 if -120 <= x <= 127 else INT16 if -32000 <= x <= 32767 else INT32
    return bytes([1 << 4 | t]) + struct.pack(_FMT[t], x)


def descriptor(n, t):
    return bytes([n << 4 | t]) if n < 15 else bytes([15 << 4 | t]) + typed_int(n)


def typed(t, values):
    """values: ints (INT*), floats or raw 32-bit patterns given as ('bits', x) (FLOAT), or bytes (CHAR)"""
    if t == CHAR:
        return descriptor(len(values), CHAR) + values
    out = descriptor(len(values), t)
    for v in values:
        if t == FLOAT:
            out += struct.pack("<I", v[1]) if isinstance(v, tuple) else struct.pack("<f", v)
        else:
            out += struct.pack(_FMT[t], v)
    return out


def record(rid, pos, ident, alleles, qual, filters, info, fmt=(), n_sample=0, rlen=None):
    """info: [(key_id, type, values)]; fmt: [(key_id, type, per_sample_count, flat values over all samples)]; qual: float or None"""
    shared = typed(CHAR, ident)
    for a in alleles:
        shared += typed(CHAR, a)
    shared += typed(INT8, filters) if filters else b"\x00"
    for key, t, values in info:
        shared += typed_int(key) + (b"\x00" if values is None else typed(t, values))
    indiv = b""
    for key, t, per, values in fmt:
        indiv += typed_int(key)
        if t == CHAR:
            indiv += descriptor(per, CHAR) + values
        else:
            indiv += descriptor(per, t) + typed(t, values)[len(descriptor(len(values), t)):]
    q = struct.pack("<I", FLOAT_MISSING) if qual is None else struct.pack("<f", qual)
    fixed = struct.pack("<ii i", rid, pos, len(alleles[0]) if rlen is None else rlen) + q + struct.pack("<II", len(alleles) << 16 | len(info), len(fmt) << 24 | n_sample)
    return struct.pack("<II", len(shared) + 24, len(indiv)) + fixed + shared + indiv


def write_bcf(path, header_text, records, compress=True, version=b"BCF\x02\x02"):
    text = header_text.encode() + b"\0"
    raw = version + struct.pack("<I", len(text)) + text + b"".join(records)
    if not compress:
        open(path, "wb").write(raw)
        return
    out = b""
    for o in range(0, len(raw), 0xff00):
        out += bamio.bgzf_member(raw[o:o + 0xff00])
    out += bamio.bgzf_member(b"")
    open(path, "wb").write(out)
