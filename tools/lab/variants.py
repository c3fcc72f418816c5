#!/usr/bin/env python3
"""Source patches for tools/lab/build.sh.  Every patch must apply exactly once."""
import sys

def sub(s, old, new, count=1):
    assert s.count(old) == count, (old, s.count(old))
    return s.replace(old, new)

def no_copy_loads(s):
    s = sub(s, "            v0 = ld128(s);\n", "            v0 = u32x4{(uint32_t)(uintptr_t)s, 0, 0, 0};\n")
    for k in range(1, 8):
        s = sub(s, "v%d = ld128(s + %d);" % (k, 16 * k), "v%d = v0;" % k)
    return s

def no_copy_stores(s):
    s = sub(s, "            if (slack >= 16) {                                     // whole chunks, scribbling < 16 bytes past the copy\n                st128(d, v0);",
            "            if (slack >= 16) {\n                if (v0[0] == 0x12345u && v7[1] == 77u) st128(d, v0);")
    for k in range(1, 8):
        s = sub(s, "                if (n > %d) st128(d + %d, v%d);\n" % (16 * k, 16 * k, k), "")
    return s

def no_literal_stores(s):
    return sub(s, "            out[o++] = (uint8_t)lit;", "            if (lit == 999u) out[o] = 0; ++o;")

def lds_bytes(nbytes):
    def f(s):
        return s          # applied to kernels.hip by patch_kernels below
    f.kernels = ("constexpr uint32_t kInflateLdsBytes = kLdsDwordsPerLane * 64 * 4;", "constexpr uint32_t kInflateLdsBytes = %d;" % nbytes)
    return f

def hot_loads(s):      # every copy load hits the same (cached) place: what do the loads cost when they never miss?
    return sub(s, "            const uint8_t *s = out + o - pend_dist;\n", "            const uint8_t *s = out + (pend_dist & 15u);\n")

def aligned_loads(s):  # wrong bytes, right addresses rounded down to 16
    return sub(s, "            const uint8_t *s = out + o - pend_dist;\n", "            const uint8_t *s = out + ((o - pend_dist) & ~15u);\n")

def aligned_stores(s):
    return sub(s, "            uint8_t *d = out + o;\n            const uint32_t slack", "            uint8_t *d = out + (o & ~15u);\n            const uint32_t slack")

def batch(nb):
    def f(s):
        return sub(s, "constexpr uint32_t kCopyBatch = 128;", "constexpr uint32_t kCopyBatch = %d;" % nb)
    return f

def lit_every8(s):      # one literal store in eight: what would combining literal writes buy?
    return sub(s, "            out[o++] = (uint8_t)lit;", "            if ((o & 7u) == 0u) out[o] = (uint8_t)lit; ++o;")

def chunk_on_boundary(s):   # copy stores only when the batch crosses a 16-byte boundary of the output (about one store per 16 output bytes)
    s = sub(s, "            if (slack >= 16) {                                     // whole chunks, scribbling < 16 bytes past the copy\n                st128(d, v0);",
            "            if (slack >= 16) {\n                if (((o + n) ^ o) & ~15u) st128(d, v0);")
    return s

VARIANTS = {
    "base": [],
    "noload": [no_copy_loads],
    "noload_nostore": [no_copy_loads, no_copy_stores],
    "decode_only": [no_copy_loads, no_copy_stores, no_literal_stores],
    "lit8": [lit_every8],
    "wc16": [lit_every8, chunk_on_boundary],
    "nolit": [no_literal_stores],
    "nostore": [no_copy_stores, no_literal_stores],
    "hotload": [hot_loads],
    "alignload": [aligned_loads],
    "alignstore": [aligned_stores],
    "alignboth": [aligned_loads, aligned_stores],
}

if __name__ == "__main__":
    if sys.argv[1] == "--list":
        print(" ".join(VARIANTS))
    else:
        v, d = sys.argv[1], sys.argv[2]
        p = d + "/inflate_core.h"
        s = open(p).read()
        for f in VARIANTS[v]:
            s = f(s)
        open(p, "w").write(s)
        pk = d + "/kernels.hip"
        k = open(pk).read()
        for f in VARIANTS[v]:
            if hasattr(f, "kernels"):
                k = sub(k, f.kernels[0], f.kernels[1])
        open(pk, "w").write(k)
