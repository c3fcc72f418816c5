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

VARIANTS = {
    "base": [],
    "noload": [no_copy_loads],
    "noload_nostore": [no_copy_loads, no_copy_stores],
    "decode_only": [no_copy_loads, no_copy_stores, no_literal_stores],
    "occ5": [lds_bytes(32768)],
    "occ3": [lds_bytes(54000)],
    "occ2": [lds_bytes(81000)],
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
