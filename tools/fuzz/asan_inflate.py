import ctypes, random, zlib, sys
emu = ctypes.CDLL('/tmp/libhostemu_asan.so')
rng = random.Random(7)
def deflate(data, level):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, rng.choice([1,8,9]), rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE]))
    return c.compress(data) + c.flush()
n_ok = n_bad = 0
for it in range(4000):
    kind = rng.random()
    if kind < 0.3: data = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 3000)))
    elif kind < 0.6: data = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(0, 20000)))
    else: data = (bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40))) * rng.randrange(1, 800))[:65536]
    comp = bytearray(deflate(data, rng.choice([0,1,6,9])))
    corrupt = rng.random() < 0.6
    if corrupt and comp:
        for _ in range(rng.choice([1,1,3])): comp[rng.randrange(len(comp))] ^= 1 << rng.randrange(8)
        if rng.random() < 0.3: comp = comp[:rng.randrange(len(comp)+1)]
    pad = bytes(comp) + b"\0" * 16                       # the documented slack: 16 readable bytes after the payload
    inbuf = ctypes.create_string_buffer(pad, len(pad))
    cap = rng.choice([len(data), len(data), max(0, len(data) - rng.randrange(1, 50)), 65536])
    a = rng.randrange(16)
    outbuf = ctypes.create_string_buffer(cap + a + 16 + 64)    # +64: loads of copy sources may look past the end (documented)
    ol = ctypes.c_uint32()
    st = emu.emu_inflate(inbuf, len(comp), ctypes.byref(outbuf, a), cap, ctypes.byref(ol))
    if st == 0:
        n_ok += 1
        if not corrupt and cap >= len(data): assert outbuf.raw[a:a+ol.value] == data
    else: n_bad += 1
print("ok", n_ok, "rejected", n_bad)
